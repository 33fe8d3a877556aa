// eig3.hpp -- symmetric 3x3 eigen-decomposition without a sweep loop (fp64), for the covariance tails of rows G2 and N1.
//
// The round-5 tails ran a cyclic Jacobi of up to 30 sweeps with data-dependent exits per lane (0.075 of HBM, 13 x above the floor of
// k_cov_from_knn).  Only the eigenvector of the SMALLEST eigenvalue is needed for the PLANE regularisation
// (fast_gicp calculate_covariances: U diag(1, 1, 1e-3) V^T == I - 0.999 n n^T, restated in oracle/gicp_oracle.cpp:410), and only the
// eigenVALUES for the point features (RING_ros/util.py:134-158).  Here, straight-line code with selects:
//   stage 1     : the eigenvalue that stands alone, by the trigonometric closed form on the matrix scaled to unit largest entry and shifted
//                 to zero trace (root 2 cos(acos(|det B| / 2) / 3) of  y^3 - 3 y - det(B) = 0), and its eigenvector as the largest cross
//                 product of two rows of A - w I;
//   stage 2     : the other two eigenpairs from the 2x2 restriction of A to that vector's orthogonal complement (cancellation-free closed
//                 form).  The trigonometric form alone is sqrt(eps)-sensitive where two eigenvalues meet -- ring-line neighbourhoods of a
//                 lidar scan have two small eigenvalues 1e-5 of the largest apart -- and left 3e-8 against the oracle's Jacobi there;
//   polish      : one step of Rayleigh-quotient iteration  v <- adj(A - (v^T A v) I) v  on the smallest eigenvector;
//   degenerate  : exactly collinear neighbours (rank 1) give a unit vector orthogonal to the line; the zero matrix and multiples of the
//                 identity give (1, 0, 0), like the Jacobi this replaces.
// Products feeding sums may contract to FMAs here (the library is otherwise built with -ffp-contract=off): nothing is compared bit for bit
// with this solver's output, and an FMA only removes a rounding.
// Compiled for the device by hipcc and for the host by g++ (tests/test_eig3.py checks this very text against LAPACK and the oracle).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define MRS_HD __host__ __device__ __forceinline__
#else
#define MRS_HD inline
#endif

namespace mrs {

struct Sym3 {
    double a00, a01, a02, a11, a12, a22;
};

MRS_HD double eig3_max(double a, double b) { return a > b ? a : b; }

// scale to unit largest |entry|; returns the scale (0 for the zero matrix, entries then stay 0)
MRS_HD double eig3_scale(const double* c, Sym3& s)
{
    double m = eig3_max(eig3_max(fabs(c[0]), fabs(c[1])), eig3_max(fabs(c[2]), fabs(c[4])));
    m = eig3_max(m, eig3_max(fabs(c[5]), fabs(c[8])));
    const double inv = m > 0.0 ? 1.0 / m : 0.0;
    s.a00 = c[0] * inv; s.a01 = c[1] * inv; s.a02 = c[2] * inv;
    s.a11 = c[4] * inv; s.a12 = c[5] * inv; s.a22 = c[8] * inv;
    return m;
}

MRS_HD void eig3_cross(const double* a, const double* b, double* c)
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// Stage 1 + 2 of the decomposition of the SCALED matrix.  The trigonometric eigenvalues are accurate to eps ||A|| only for the eigenvalue that
// stands alone (acos is sqrt(eps)-sensitive where two eigenvalues meet), so, like Eberly's robust 3x3 solver:
//   stage 1: the separated eigenvalue (the smallest when det(B) < 0, plane-like; the largest otherwise, line-like) and its eigenvector e as the
//            largest cross product of two rows of A - w I (well conditioned: the other two eigenvalues are both far from w);
//   stage 2: an orthonormal basis (U, V) of e's complement and the 2x2 restriction [[m00, m01], [m01, m11]] of A to it, whose eigenpairs come
//            from a cancellation-free closed form: the remaining eigenvalues are mid -+ r.
// Everything is accurate to ~eps ||A|| / gap, the bound of a backward-stable solver.
struct Eig3Stages {
    bool sep_is_smallest, isotropic;      // isotropic: zero matrix or a multiple of the identity
    double w_sep, e[3], U[3], V[3], m00, m01, m11;
};

MRS_HD void eig3_stages(const Sym3& s, Eig3Stages& g)
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    const double q = (s.a00 + s.a11 + s.a22) * (1.0 / 3.0);
    const double b00 = s.a00 - q, b11 = s.a11 - q, b22 = s.a22 - q;
    const double off = s.a01 * s.a01 + s.a02 * s.a02 + s.a12 * s.a12;
    const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * off) * (1.0 / 6.0));
    const double ip = p > 0.0 ? 1.0 / p : 0.0;
    const double e00 = b00 * ip, e11 = b11 * ip, e22 = b22 * ip, e01 = s.a01 * ip, e02 = s.a02 * ip, e12 = s.a12 * ip;
    const double det = e00 * (e11 * e22 - e12 * e12) - e01 * (e01 * e22 - e12 * e02) + e02 * (e01 * e12 - e11 * e02);
    double h = 0.5 * det;
    h = h < -1.0 ? -1.0 : (h > 1.0 ? 1.0 : h);
    g.sep_is_smallest = h < 0.0;
    g.isotropic = !(p > 0.0);
    // |h| -> the separated root is 2 cos(acos(|h|) / 3) up to sign: one acos + one cos, both away from their sensitive ends
    const double beta = 2.0 * cos(acos(fabs(h)) * (1.0 / 3.0));
    const double w = g.sep_is_smallest ? q - p * beta : q + p * beta;
    {
        const double r0[3] = {s.a00 - w, s.a01, s.a02}, r1[3] = {s.a01, s.a11 - w, s.a12}, r2[3] = {s.a02, s.a12, s.a22 - w};
        double c01[3], c02[3], c12[3];
        eig3_cross(r0, r1, c01); eig3_cross(r0, r2, c02); eig3_cross(r1, r2, c12);
        const double d01 = c01[0] * c01[0] + c01[1] * c01[1] + c01[2] * c01[2];
        const double d02 = c02[0] * c02[0] + c02[1] * c02[1] + c02[2] * c02[2];
        const double d12 = c12[0] * c12[0] + c12[1] * c12[1] + c12[2] * c12[2];
        const bool use02 = d02 > d01;
        double dm = use02 ? d02 : d01;
        double cx = use02 ? c02[0] : c01[0], cy = use02 ? c02[1] : c01[1], cz = use02 ? c02[2] : c01[2];
        const bool use12 = d12 > dm;
        dm = use12 ? d12 : dm;
        cx = use12 ? c12[0] : cx; cy = use12 ? c12[1] : cy; cz = use12 ? c12[2] : cz;
        // all three eigenvalues equal (zero matrix, multiples of the identity): every cross product vanishes -> e = (1, 0, 0), and the 2x2 stage
        // then returns U = (0, 0, 1) x ... consistently (any basis is an eigenbasis)
        const bool none = !(dm > 0.0);
        cx = none ? 1.0 : cx; cy = none ? 0.0 : cy; cz = none ? 0.0 : cz;
        const double inv = 1.0 / sqrt(cx * cx + cy * cy + cz * cz);
        g.e[0] = cx * inv; g.e[1] = cy * inv; g.e[2] = cz * inv;
    }
    // orthonormal complement of e
    const double* W = g.e;
    const bool first = fabs(W[0]) > fabs(W[1]);
    const double il = 1.0 / sqrt(first ? W[0] * W[0] + W[2] * W[2] : W[1] * W[1] + W[2] * W[2]);
    g.U[0] = first ? -W[2] * il : 0.0;
    g.U[1] = first ? 0.0 : W[2] * il;
    g.U[2] = first ? W[0] * il : -W[1] * il;
    eig3_cross(W, g.U, g.V);
    const double au0 = s.a00 * g.U[0] + s.a01 * g.U[1] + s.a02 * g.U[2];
    const double au1 = s.a01 * g.U[0] + s.a11 * g.U[1] + s.a12 * g.U[2];
    const double au2 = s.a02 * g.U[0] + s.a12 * g.U[1] + s.a22 * g.U[2];
    const double av0 = s.a00 * g.V[0] + s.a01 * g.V[1] + s.a02 * g.V[2];
    const double av1 = s.a01 * g.V[0] + s.a11 * g.V[1] + s.a12 * g.V[2];
    const double av2 = s.a02 * g.V[0] + s.a12 * g.V[1] + s.a22 * g.V[2];
    g.m00 = g.U[0] * au0 + g.U[1] * au1 + g.U[2] * au2;
    g.m01 = g.U[0] * av0 + g.U[1] * av1 + g.U[2] * av2;
    g.m11 = g.V[0] * av0 + g.V[1] * av1 + g.V[2] * av2;
    // the separated eigenvalue as the Rayleigh quotient of its vector (second-order accurate in the vector's error)
    const double ae0 = s.a00 * W[0] + s.a01 * W[1] + s.a02 * W[2];
    const double ae1 = s.a01 * W[0] + s.a11 * W[1] + s.a12 * W[2];
    const double ae2 = s.a02 * W[0] + s.a12 * W[1] + s.a22 * W[2];
    g.w_sep = W[0] * ae0 + W[1] * ae1 + W[2] * ae2;
}

// unit eigenvector of the smallest eigenvalue of the symmetric 3x3 `c` (row-major, 9 doubles) -> n_out[3]
MRS_HD void smallest_eigvec(const double* c, double* n_out)
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    Sym3 s;
    eig3_scale(c, s);
    Eig3Stages g;
    eig3_stages(s, g);
    double v[3];
    {
        // eigenvector of the SMALLER eigenvalue of [[m00, m01], [m01, m11]]: (m01, -d - r) or (d - r, m01), whichever has no cancellation
        const double d = 0.5 * (g.m00 - g.m11), r = sqrt(d * d + g.m01 * g.m01);
        const bool pos = d >= 0.0;
        double x = pos ? g.m01 : d - r, y = pos ? -d - r : g.m01;
        const bool flat = !(r > 0.0);                                 // the 2x2 is a multiple of the identity: any direction, take U
        x = flat ? 1.0 : x; y = flat ? 0.0 : y;
        const double inv = 1.0 / sqrt(x * x + y * y);
        x *= inv; y *= inv;
        v[0] = g.sep_is_smallest ? g.e[0] : x * g.U[0] + y * g.V[0];
        v[1] = g.sep_is_smallest ? g.e[1] : x * g.U[1] + y * g.V[1];
        v[2] = g.sep_is_smallest ? g.e[2] : x * g.U[2] + y * g.V[2];
    }
    // one step of Rayleigh-quotient iteration  v <- adj(A - (v^T A v) I) v  (cubic): polishes the last digits; skipped where adj ~ 0
    // (a double small eigenvalue to ~1e-10 of the largest: every unit vector of that plane is an answer)
    {
        const double av0 = s.a00 * v[0] + s.a01 * v[1] + s.a02 * v[2];
        const double av1 = s.a01 * v[0] + s.a11 * v[1] + s.a12 * v[2];
        const double av2 = s.a02 * v[0] + s.a12 * v[1] + s.a22 * v[2];
        const double mu = v[0] * av0 + v[1] * av1 + v[2] * av2;
        const double r0[3] = {s.a00 - mu, s.a01, s.a02}, r1[3] = {s.a01, s.a11 - mu, s.a12}, r2[3] = {s.a02, s.a12, s.a22 - mu};
        double k0[3], k1[3], k2[3];                                   // rows (= columns) of adj(A - mu I)
        eig3_cross(r1, r2, k0); eig3_cross(r2, r0, k1); eig3_cross(r0, r1, k2);
        const double x = k0[0] * v[0] + k0[1] * v[1] + k0[2] * v[2];
        const double y = k1[0] * v[0] + k1[1] * v[1] + k1[2] * v[2];
        const double z = k2[0] * v[0] + k2[1] * v[1] + k2[2] * v[2];
        const double nn = x * x + y * y + z * z;
        const bool ok = nn > 1e-20;
        const double inv = 1.0 / sqrt(ok ? nn : 1.0);
        v[0] = ok ? x * inv : v[0]; v[1] = ok ? y * inv : v[1]; v[2] = ok ? z * inv : v[2];
    }
    n_out[0] = g.isotropic ? 1.0 : v[0]; n_out[1] = g.isotropic ? 0.0 : v[1]; n_out[2] = g.isotropic ? 0.0 : v[2];
}

// eigenvalues of the symmetric 3x3 `c`, DESCENDING (w[0] >= w[1] >= w[2])
MRS_HD void sym3_eigvals(const double* c, double* w)
{
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
    Sym3 s;
    const double m = eig3_scale(c, s);
    Eig3Stages g;
    eig3_stages(s, g);
    const double mid = 0.5 * (g.m00 + g.m11), d = 0.5 * (g.m00 - g.m11), r = sqrt(d * d + g.m01 * g.m01);
    const double lo = mid - r, hi = mid + r;
    w[0] = (g.sep_is_smallest ? hi : g.w_sep) * m;
    w[1] = (g.sep_is_smallest ? lo : hi) * m;
    w[2] = (g.sep_is_smallest ? g.w_sep : lo) * m;
}

}  // namespace mrs

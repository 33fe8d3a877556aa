// oracle/ref_gm_host/fftw3.h -- TEST INFRASTRUCTURE ONLY.
// Stand-in for the five FFTW names GlobalManager::calcRelOri uses (FFTW is not in this image): an in-place, unnormalised 2-D
// complex DFT in double, rows then columns, every output a plain sum in index order (FFTW_BACKWARD = exponent sign +).
#pragma once
#include <cmath>
#include <cstdlib>
#include <vector>

typedef double fftw_complex[2];
struct ref_fftw_plan_s { int h, w, sign; fftw_complex *in, *out; };
typedef ref_fftw_plan_s* fftw_plan;
enum { FFTW_FORWARD = -1, FFTW_BACKWARD = +1, FFTW_ESTIMATE = 64 };

inline void* fftw_malloc(size_t n) { return std::malloc(n); }
inline void fftw_free(void* p) { std::free(p); }
inline fftw_plan fftw_plan_dft_2d(int h, int w, fftw_complex* in, fftw_complex* out, int sign, unsigned) { return new ref_fftw_plan_s{h, w, sign, in, out}; }
inline void fftw_destroy_plan(fftw_plan p) { delete p; }

inline void ref_dft_line(const std::vector<double>& xr, const std::vector<double>& xi, int n, int sign, std::vector<double>& yr, std::vector<double>& yi)
{
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < n; ++k) {
        double sr = 0.0, si = 0.0;
        for (int j = 0; j < n; ++j) {
            const double ang = sign * 2.0 * pi * (double)((long)k * j % n) / n;
            const double c = std::cos(ang), s = std::sin(ang);
            sr += xr[j] * c - xi[j] * s;
            si += xr[j] * s + xi[j] * c;
        }
        yr[k] = sr; yi[k] = si;
    }
}

inline void fftw_execute(fftw_plan p)
{
    const int h = p->h, w = p->w;
    std::vector<double> re((size_t)h * w), im((size_t)h * w);
    for (int i = 0; i < h * w; ++i) { re[i] = p->in[i][0]; im[i] = p->in[i][1]; }
    std::vector<double> xr(w), xi(w), yr(w), yi(w);
    for (int r = 0; r < h; ++r) {                 // along the rows (width)
        for (int c = 0; c < w; ++c) { xr[c] = re[(size_t)r * w + c]; xi[c] = im[(size_t)r * w + c]; }
        ref_dft_line(xr, xi, w, p->sign, yr, yi);
        for (int c = 0; c < w; ++c) { re[(size_t)r * w + c] = yr[c]; im[(size_t)r * w + c] = yi[c]; }
    }
    xr.resize(h); xi.resize(h); yr.resize(h); yi.resize(h);
    for (int c = 0; c < w; ++c) {                 // along the columns (height)
        for (int r = 0; r < h; ++r) { xr[r] = re[(size_t)r * w + c]; xi[r] = im[(size_t)r * w + c]; }
        ref_dft_line(xr, xi, h, p->sign, yr, yi);
        for (int r = 0; r < h; ++r) { re[(size_t)r * w + c] = yr[r]; im[(size_t)r * w + c] = yi[r]; }
    }
    for (int i = 0; i < h * w; ++i) { p->out[i][0] = re[i]; p->out[i][1] = im[i]; }
}

// eigen_min.hpp -- stand-in for the three fixed-size Eigen types in libgpu.so's signatures, used ONLY where Eigen itself
// is not installed (this build image).  Same template signature as Eigen::Matrix (so that the mangled names of the
// exported functions are the ones the Mapping node links against), same storage (dense float array, column-major unless
// Options has the RowMajor bit), and a user-provided copy constructor like Eigen's, so that the C++ ABI passes the by-value
// arguments the same way (by reference to a temporary).  A maintainer builds libgpu_shim.cpp against the real <Eigen/Core>.
#pragma once
namespace Eigen {
enum { ColMajor = 0, RowMajor = 1 };
template <typename Scalar, int Rows, int Cols, int Options, int MaxRows, int MaxCols>
class Matrix {
public:
    Matrix() { for (int i = 0; i < Rows * Cols; ++i) m_[i] = Scalar(0); }
    Matrix(const Matrix& o) { for (int i = 0; i < Rows * Cols; ++i) m_[i] = o.m_[i]; }
    Matrix& operator=(const Matrix& o) { for (int i = 0; i < Rows * Cols; ++i) m_[i] = o.m_[i]; return *this; }
    Scalar& operator()(int r, int c) { return (Options & RowMajor) ? m_[r * Cols + c] : m_[c * Rows + r]; }
    Scalar operator()(int r, int c) const { return (Options & RowMajor) ? m_[r * Cols + c] : m_[c * Rows + r]; }
    const Scalar* data() const { return m_; }
private:
    alignas((Rows * Cols * sizeof(Scalar)) % 16 == 0 ? 16 : alignof(Scalar)) Scalar m_[Rows * Cols];
};
typedef Matrix<float, 4, 4, 0, 4, 4> Matrix4f;
typedef Matrix<float, 3, 3, 0, 3, 3> Matrix3f;
typedef Matrix<float, 1, 3, 1, 1, 3> RowVector3f;
}  // namespace Eigen

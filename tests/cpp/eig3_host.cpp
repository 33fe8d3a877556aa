// Host build of mr_slam_amd/csrc/eig3.hpp (the text the device kernels compile) behind two C entry points, for tests/test_eig3.py.
#include "../../mr_slam_amd/csrc/eig3.hpp"

extern "C" void eig3_smallest_vec(const double* a /* [n][9] */, int n, double* out /* [n][3] */)
{
    for (int i = 0; i < n; ++i) mrs::smallest_eigvec(a + 9 * (long)i, out + 3 * (long)i);
}

extern "C" void eig3_values_desc(const double* a, int n, double* out /* [n][3] */)
{
    for (int i = 0; i < n; ++i) mrs::sym3_eigvals(a + 9 * (long)i, out + 3 * (long)i);
}

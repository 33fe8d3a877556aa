"""The fp32 fast path of the Cartesian rasterisers is exact: EVERY float in [-1, 1] that the fast path accepts lands in the bin the
reference's double formula gives (generate_bev_cython_binary/src/kernel.cu:52-54).  The device expression consists of correctly
rounded IEEE operations only, so evaluating it on the host (oracle/fastpath_oracle.c, no contraction, no fast-math) reproduces the
device's bits; the walk is exhaustive over the 2.13 G bit patterns, not sampled."""
import pytest


def _eps_stand_alone(bins):          # mr_slam_amd/csrc/bev_cart.hpp eps_for(): k_cart_lds, cart_axis
    return max(2e-4, (bins + 2) * 2e-6)


def _eps_fast(bins):                 # mr_slam_amd/csrc/bev_cart.hpp make_cart(): the fused kernel's fast path
    return (bins + 2) * 4e-7


@pytest.mark.parametrize("bins", [120, 40, 200, 1000])
def test_cart_fast_path_is_exact_for_every_float(oracle, bins):
    bound = bins * 2.0 ** -23 * (1 + 2.0 ** -24)          # two roundings: inv = fl(1 / gap), g = fl(v * inv + inv)
    for eps in (_eps_fast(bins), _eps_stand_alone(bins)):
        accepted, mismatches, worst, bad = oracle.cart_fastpath_check(bins, eps)
        assert mismatches == 0, f"bins={bins} eps={eps}: {mismatches} accepted values land in another bin, e.g. v={bad!r}"
        assert accepted > 2.0e8                             # the fast path is the common case, not an empty set
        assert worst <= bound and eps >= 3.0 * bound, (worst, bound, eps)


def test_cart_fast_path_needs_its_margin(oracle):
    """without a margin the fp32 quotient does cross bin edges: the check is able to fail"""
    accepted, mismatches, _, bad = oracle.cart_fastpath_check(120, 0.0)
    assert mismatches > 0 and bad != 0.0


@pytest.mark.parametrize("bins", [120, 1000])
def test_cart_axis_fast_try_is_exact_for_every_float(oracle, bins):
    """cart_axis() (index kernels, reference-layout kernels, first try of the exact path): (v + 1.0f) * inv, three roundings"""
    accepted, mismatches, worst, bad = oracle.cart_fastpath_check(bins, _eps_stand_alone(bins), axis_form=True)
    assert mismatches == 0, f"bins={bins}: v={bad!r}"
    assert accepted > 2.0e8 and worst <= bins * 3 * 2.0 ** -24 * 1.001 and _eps_stand_alone(bins) >= 5 * worst


@pytest.mark.parametrize("num_height", [20, 1])
def test_polar_height_fast_path_is_exact_for_every_float(oracle, num_height):
    """height layer of the polar rasterisers (kernel.cpp:48,66: float add, float division, floor), every finite float z"""
    eps = max(2e-5, (2 * num_height + 16) * 2e-6)            # bev.hip make_polar()
    accepted, mismatches, bad = oracle.polar_height_fastpath_check(num_height, 1, eps)
    assert mismatches == 0, f"num_height={num_height}: z={bad!r}"
    assert accepted > 1.0e9

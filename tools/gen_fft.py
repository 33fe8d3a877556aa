#!/usr/bin/env python3
"""Generates mr_slam_amd/csrc/fft_codelets.hpp: fully unrolled, in-register complex FFT codelets
(straight-line code, compile-time twiddles, static register indexing) for the correlation kernels.

    python tools/gen_fft.py            # writes the header
    python tools/gen_fft.py --check    # also executes the generated arithmetic in numpy vs numpy.fft

Mixed-radix decimation in time with hard-coded radix 2/3/4/5 butterflies; 60 = 4 * 3 * 5.
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Emitter:
    """Straight-line program over complex values held as v2f = (re, im) register pairs.  Every statement is recorded twice:
    as C text (packed fp32: one v_pk_add / v_pk_mul / v_pk_fma per statement on gfx950) and as a closure over numpy
    float32 pairs (--check executes exactly the emitted arithmetic, fused multiply-adds included)."""

    def __init__(self):
        self.lines = []
        self.prog = []
        self.n = 0

    def tmp(self, expr, fn):
        name = f"t{self.n}"
        self.n += 1
        self.lines.append(f"const v2f {name} = {expr};")
        self.prog.append((name, fn))
        return name


def lit(v):
    return repr(float(np.float32(v))) + "f"


def pair(a, b):
    return f"(v2f){{{lit(a)}, {lit(b)}}}"


def f32(v):
    return np.float32(v)


def fma32(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))      # exact product + one rounding (|values| << 2^100)


def add(E, a, b):
    return E.tmp(f"{a} + {b}", lambda e: (e[a][0] + e[b][0], e[a][1] + e[b][1]))


def sub(E, a, b):
    return E.tmp(f"{a} - {b}", lambda e: (e[a][0] - e[b][0], e[a][1] - e[b][1]))


def fma_c(E, a, c0, c1, b):
    """a * (c0, c1) + b, fused"""
    c0, c1 = f32(c0), f32(c1)
    return E.tmp(f"__builtin_elementwise_fma({a}, {pair(c0, c1)}, {b})", lambda e: (fma32(e[a][0], c0, e[b][0]), fma32(e[a][1], c1, e[b][1])))


def fma_swap_c(E, a, c0, c1, b):
    """(a.im, a.re) * (c0, c1) + b, fused: b + i*a is (c0, c1) = (-1, 1)"""
    c0, c1 = f32(c0), f32(c1)
    return E.tmp(f"__builtin_elementwise_fma({a}.yx, {pair(c0, c1)}, {b})", lambda e: (fma32(e[a][1], c0, e[b][0]), fma32(e[a][0], c1, e[b][1])))


def mul_c(E, a, c0, c1):
    c0, c1 = f32(c0), f32(c1)
    return E.tmp(f"{a} * {pair(c0, c1)}", lambda e: (e[a][0] * c0, e[a][1] * c1))


def mul_swap_c(E, a, c0, c1):
    c0, c1 = f32(c0), f32(c1)
    return E.tmp(f"{a}.yx * {pair(c0, c1)}", lambda e: (e[a][1] * c0, e[a][0] * c1))


def add_i(E, b, a, sign):
    """b + sign * i * a"""
    return fma_swap_c(E, a, -sign, sign, b)


def cmul_const(E, x, wr, wi):
    """x * (wr + i wi) with constant w; trivial cases folded."""
    if abs(wr - 1) < 1e-15 and abs(wi) < 1e-15:
        return x
    if abs(wr + 1) < 1e-15 and abs(wi) < 1e-15:
        return mul_c(E, x, -1.0, -1.0)
    if abs(wr) < 1e-15 and abs(wi - 1) < 1e-15:
        return mul_swap_c(E, x, -1.0, 1.0)
    if abs(wr) < 1e-15 and abs(wi + 1) < 1e-15:
        return mul_swap_c(E, x, 1.0, -1.0)
    # (xr wr - xi wi, xi wr + xr wi) = fma(x, (wr, wr), x.yx * (-wi, wi))
    t = mul_swap_c(E, x, -wi, wi)
    return fma_c(E, x, wr, wr, t)


def butterfly(E, xs, sign):
    """DFT of len(xs) in {2,3,4,5}: X[q] = sum_j xs[j] exp(sign*2*pi*i*j*q/r)."""
    r = len(xs)
    if r == 2:
        return [add(E, xs[0], xs[1]), sub(E, xs[0], xs[1])]
    if r == 4:
        a, b = add(E, xs[0], xs[2]), sub(E, xs[0], xs[2])
        c, d = add(E, xs[1], xs[3]), sub(E, xs[1], xs[3])
        return [add(E, a, c), add_i(E, b, d, sign), sub(E, a, c), add_i(E, b, d, -sign)]
    if r == 3:
        s = add(E, xs[1], xs[2])
        d = sub(E, xs[1], xs[2])
        x0 = add(E, xs[0], s)
        c = math.cos(2 * math.pi / 3)
        sn = sign * math.sin(2 * math.pi / 3)
        m = fma_c(E, s, c, c, xs[0])
        return [x0, fma_swap_c(E, d, -sn, sn, m), fma_swap_c(E, d, sn, -sn, m)]      # m +- i sn d
    if r == 5:
        c1, c2 = math.cos(2 * math.pi / 5), math.cos(4 * math.pi / 5)
        s1, s2 = sign * math.sin(2 * math.pi / 5), sign * math.sin(4 * math.pi / 5)
        a1, b1 = add(E, xs[1], xs[4]), sub(E, xs[1], xs[4])
        a2, b2 = add(E, xs[2], xs[3]), sub(E, xs[2], xs[3])
        x0 = add(E, add(E, xs[0], a1), a2)
        m1 = fma_c(E, a2, c2, c2, fma_c(E, a1, c1, c1, xs[0]))
        m2 = fma_c(E, a2, c1, c1, fma_c(E, a1, c2, c2, xs[0]))
        u = fma_c(E, b2, s2, s2, mul_c(E, b1, s1, s1))       # s1 b1 + s2 b2
        v = fma_c(E, b2, -s1, -s1, mul_c(E, b1, s2, s2))     # s2 b1 - s1 b2
        return [x0, add_i(E, m1, u, 1), add_i(E, m2, v, 1), add_i(E, m2, v, -1), add_i(E, m1, u, -1)]
    raise ValueError(r)


def gen_fft(E, xs, sign):
    """Natural-order DFT of the list of symbols xs."""
    N = len(xs)
    if N == 1:
        return xs
    if N in (2, 3, 4, 5):
        return butterfly(E, xs, sign)
    for r in (4, 2, 3, 5):
        if N % r == 0:
            break
    else:
        raise ValueError(N)
    m = N // r
    subs = [gen_fft(E, xs[j::r], sign) for j in range(r)]
    out = [None] * N
    for k in range(m):
        tw = []
        for j in range(r):
            ang = sign * 2 * math.pi * j * k / N
            tw.append(cmul_const(E, subs[j][k], math.cos(ang), math.sin(ang)))
        res = butterfly(E, tw, sign)
        for q in range(r):
            out[k + m * q] = res[q]
    return out


def emit_codelet(N, sign, name):
    E = Emitter()
    ins = [E.tmp(f"x[{i}]", (lambda i: lambda e: e["x"][i])(i)) for i in range(N)]   # inputs first: the outputs overwrite the array
    out = gen_fft(E, ins, sign)
    body = "\n    ".join(E.lines)
    stores = "\n    ".join(f"x[{i}] = {o};" for i, o in enumerate(out))
    return (f"// {N}-point complex DFT, exponent sign {'+' if sign > 0 else '-'}, unnormalised, in place, natural order\n"
            f"__device__ __forceinline__ void {name}(v2f (&x)[{N}])\n{{\n    {body}\n    {stores}\n}}\n"), E, out


def run_numpy(E, out, x):
    """Execute the generated arithmetic with numpy float32 scalars (for --check)."""
    env = {"x": [(np.float32(v.real), np.float32(v.imag)) for v in x]}
    for name, fn in E.prog:
        env[name] = fn(env)
    return np.array([complex(env[o][0], env[o][1]) for o in out])


def main():
    check = "--check" in sys.argv
    parts = ["// GENERATED by tools/gen_fft.py -- do not edit.  Straight-line in-register FFT codelets; a complex value is one v2f\n"
             "// (re, im) in an aligned VGPR pair, so that every statement is ONE packed fp32 instruction (v_pk_add / v_pk_mul / v_pk_fma).\n"
             "#pragma once\n\ntypedef float v2f __attribute__((ext_vector_type(2)));\n"]
    for sign, name in ((+1, "cfft60_inv"), (-1, "cfft60_fwd")):
        code, E, out = emit_codelet(60, sign, name)
        parts.append(code)
        print(f"{name}: {len(E.lines) - 60} packed statements")
        if check:
            rng = np.random.default_rng(1)
            x = (rng.normal(size=60) + 1j * rng.normal(size=60)).astype(np.complex64)
            got = run_numpy(E, out, x)
            ref = np.fft.ifft(x.astype(np.complex128)) * 60 if sign > 0 else np.fft.fft(x.astype(np.complex128))
            err = np.abs(got - ref).max() / np.abs(ref).max()
            print(f"  check vs numpy.fft: max rel err {err:.2e}")
            assert err < 5e-6
    cs = ", ".join(lit(math.cos(2 * math.pi * k / 120)) for k in range(61))
    sn = ", ".join(lit(math.sin(2 * math.pi * k / 120)) for k in range(61))
    parts.append("// exp(2*pi*i*k/120), k = 0..60 (pre/post-processing twiddles of the 120-point real transforms)\n"
                 f"__device__ constexpr float kCos120[61] = {{{cs}}};\n__device__ constexpr float kSin120[61] = {{{sn}}};\n")
    path = os.path.join(ROOT, "mr_slam_amd", "csrc", "fft_codelets.hpp")
    open(path, "w").write("\n".join(parts))
    print("wrote", path)


if __name__ == "__main__":
    main()

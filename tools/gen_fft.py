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
    def __init__(self):
        self.lines = []
        self.n = 0

    def tmp(self, expr):
        name = f"t{self.n}"
        self.n += 1
        self.lines.append(f"const float {name} = {expr};")
        return name


def lit(v):
    return repr(float(np.float32(v))) + "f"


def cmul_const(E, x, wr, wi):
    """(xr + i xi) * (wr + i wi) with constant w; trivial cases folded."""
    xr, xi = x
    if abs(wr - 1) < 1e-15 and abs(wi) < 1e-15:
        return x
    if abs(wr + 1) < 1e-15 and abs(wi) < 1e-15:
        return (E.tmp(f"-{xr}"), E.tmp(f"-{xi}"))
    if abs(wr) < 1e-15 and abs(wi - 1) < 1e-15:
        return (E.tmp(f"-{xi}"), xr)
    if abs(wr) < 1e-15 and abs(wi + 1) < 1e-15:
        return (xi, E.tmp(f"-{xr}"))
    re = E.tmp(f"__builtin_fmaf({xr}, {lit(wr)}, -({xi} * {lit(wi)}))")
    im = E.tmp(f"__builtin_fmaf({xr}, {lit(wi)}, {xi} * {lit(wr)})")
    return (re, im)


def add(E, a, b):
    return (E.tmp(f"{a[0]} + {b[0]}"), E.tmp(f"{a[1]} + {b[1]}"))


def sub(E, a, b):
    return (E.tmp(f"{a[0]} - {b[0]}"), E.tmp(f"{a[1]} - {b[1]}"))


def mul_i(E, a, sign):
    """a * (sign * i)"""
    if sign > 0:
        return (E.tmp(f"-{a[1]}"), a[0])
    return (a[1], E.tmp(f"-{a[0]}"))


def butterfly(E, xs, sign):
    """DFT of len(xs) in {2,3,4,5}: X[q] = sum_j xs[j] exp(sign*2*pi*i*j*q/r)."""
    r = len(xs)
    if r == 2:
        return [add(E, xs[0], xs[1]), sub(E, xs[0], xs[1])]
    if r == 4:
        a, b = add(E, xs[0], xs[2]), sub(E, xs[0], xs[2])
        c, d = add(E, xs[1], xs[3]), sub(E, xs[1], xs[3])
        di = mul_i(E, d, sign)
        return [add(E, a, c), add(E, b, di), sub(E, a, c), sub(E, b, di)]
    if r == 3:
        s = add(E, xs[1], xs[2])
        d = sub(E, xs[1], xs[2])
        x0 = add(E, xs[0], s)
        c = math.cos(2 * math.pi / 3)
        sn = sign * math.sin(2 * math.pi / 3)
        m = (E.tmp(f"__builtin_fmaf({s[0]}, {lit(c)}, {xs[0][0]})"), E.tmp(f"__builtin_fmaf({s[1]}, {lit(c)}, {xs[0][1]})"))
        # i * sn * d
        rot = (E.tmp(f"{lit(-sn)} * {d[1]}"), E.tmp(f"{lit(sn)} * {d[0]}"))
        return [x0, add(E, m, rot), sub(E, m, rot)]
    if r == 5:
        c1, c2 = math.cos(2 * math.pi / 5), math.cos(4 * math.pi / 5)
        s1, s2 = sign * math.sin(2 * math.pi / 5), sign * math.sin(4 * math.pi / 5)
        a1, b1 = add(E, xs[1], xs[4]), sub(E, xs[1], xs[4])
        a2, b2 = add(E, xs[2], xs[3]), sub(E, xs[2], xs[3])
        x0 = (E.tmp(f"{xs[0][0]} + {a1[0]} + {a2[0]}"), E.tmp(f"{xs[0][1]} + {a1[1]} + {a2[1]}"))
        m1 = (E.tmp(f"__builtin_fmaf({a2[0]}, {lit(c2)}, __builtin_fmaf({a1[0]}, {lit(c1)}, {xs[0][0]}))"),
              E.tmp(f"__builtin_fmaf({a2[1]}, {lit(c2)}, __builtin_fmaf({a1[1]}, {lit(c1)}, {xs[0][1]}))"))
        m2 = (E.tmp(f"__builtin_fmaf({a2[0]}, {lit(c1)}, __builtin_fmaf({a1[0]}, {lit(c2)}, {xs[0][0]}))"),
              E.tmp(f"__builtin_fmaf({a2[1]}, {lit(c1)}, __builtin_fmaf({a1[1]}, {lit(c2)}, {xs[0][1]}))"))
        # i * (s1 b1 + s2 b2)  and  i * (s2 b1 - s1 b2)
        u = (E.tmp(f"__builtin_fmaf({b2[0]}, {lit(s2)}, {b1[0]} * {lit(s1)})"), E.tmp(f"__builtin_fmaf({b2[1]}, {lit(s2)}, {b1[1]} * {lit(s1)})"))
        v = (E.tmp(f"__builtin_fmaf({b2[0]}, {lit(-s1)}, {b1[0]} * {lit(s2)})"), E.tmp(f"__builtin_fmaf({b2[1]}, {lit(-s1)}, {b1[1]} * {lit(s2)})"))
        iu = (E.tmp(f"-{u[1]}"), u[0])
        iv = (E.tmp(f"-{v[1]}"), v[0])
        return [x0, add(E, m1, iu), add(E, m2, iv), sub(E, m2, iv), sub(E, m1, iu)]
    raise ValueError(r)


def gen_fft(E, xs, sign):
    """Natural-order DFT of the list of (re, im) symbols xs."""
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
    xs = [(f"re[{i}]", f"im[{i}]") for i in range(N)]
    # read inputs into temporaries first (the outputs overwrite the arrays)
    ins = [(E.tmp(a), E.tmp(b)) for a, b in xs]
    out = gen_fft(E, ins, sign)
    body = "\n    ".join(E.lines)
    stores = "\n    ".join(f"re[{i}] = {o[0]}; im[{i}] = {o[1]};" for i, o in enumerate(out))
    return (f"// {N}-point complex DFT, exponent sign {'+' if sign > 0 else '-'}, unnormalised, in place, natural order\n"
            f"__device__ __forceinline__ void {name}(float (&re)[{N}], float (&im)[{N}])\n{{\n    {body}\n    {stores}\n}}\n"), E


def run_numpy(E_lines, stores, N, x):
    """Execute the generated arithmetic with numpy float32 scalars (for --check)."""
    env = {"re": [np.float32(v.real) for v in x], "im": [np.float32(v.imag) for v in x],
           "__builtin_fmaf": lambda a, b, c: np.float32(np.float64(a) * np.float64(b) + np.float64(c))}
    for ln in E_lines:
        name, expr = ln[len("const float "):-1].split(" = ", 1)
        expr = expr.replace("f,", ",").replace("f)", ")").replace("f *", " *").replace("f;", ";")
        import re as _re
        expr = _re.sub(r"(\d)f\b", r"\1", expr)
        env[name] = np.float32(eval(expr, {}, env))
    return env


def main():
    check = "--check" in sys.argv
    parts = ["// GENERATED by tools/gen_fft.py -- do not edit.  Straight-line in-register FFT codelets.\n#pragma once\n"]
    for sign, name in ((+1, "cfft60_inv"), (-1, "cfft60_fwd")):
        code, E = emit_codelet(60, sign, name)
        parts.append(code)
        print(f"{name}: {len(E.lines)} statements")
        if check:
            rng = np.random.default_rng(1)
            x = (rng.normal(size=60) + 1j * rng.normal(size=60)).astype(np.complex64)
            out = gen_fft(Emitter(), [(0, 0)] * 0, sign) if False else None
            # re-run generation capturing outputs
            E2 = Emitter()
            ins = [(E2.tmp(f"re[{i}]"), E2.tmp(f"im[{i}]")) for i in range(60)]
            outs = gen_fft(E2, ins, sign)
            env = run_numpy(E2.lines, None, 60, x)
            got = np.array([complex(env[o[0]] if isinstance(o[0], str) else o[0], env[o[1]] if isinstance(o[1], str) else o[1]) for o in outs])
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

"""Generates tests/golden/radon_ring120.npz with the REFERENCE's analytic Radon checker
(torch-radon/src/symbolic.cpp compiled in place -> oracle/_ref/libref_symbolic.so).

  image         : a seeded 30-blob phantom discretised by SymbolicFunction::discretize
  sino_analytic : symbolic_forward at the RING geometry (120 angles linspace(0,2pi,120), 120 det)
  sino_oracle   : oracle/radon_oracle.c on the same image (regression pin of the restatement)
Run in the build container:  python tests/golden/make_golden_radon.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as O  # noqa: E402

rng = np.random.default_rng(42)
f = O.RefSymbolicFunction(120, 120)
for _ in range(30):
    w = rng.uniform(0, 1)
    cx, cy = rng.uniform(-60, 60), rng.uniform(-60, 60)
    rx, ry = rng.uniform(120 / 32, 30), rng.uniform(120 / 32, 30)
    if rng.integers(0, 2) < 0.25:
        f.add_ellipse(w, cx, cy, rx, ry)
    else:
        f.add_gaussian(5 * w, cx, cy, 0.5 / rx ** 2, 0.5 / ry ** 2)
img = f.discretize(120, 120)
ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
ana = f.forward(ang, 120, 1.0)
orc = O.radon_parallel(img, ang, 120, 1.0)
print("rel L2 error vs analytic:", np.linalg.norm(ana - orc) / np.linalg.norm(ana))
np.savez_compressed(os.path.join(HERE, "radon_ring120.npz"), image=img, angles=ang, sino_analytic=ana, sino_oracle=orc)

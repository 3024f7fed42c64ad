#!/usr/bin/env python3
"""One-off fuzz of fsq_phase_correlate against the oracle (TEST TOOL): random shapes (odd, prime, strips, beyond the LDS tiles of
the vector-ALU DFT), upsample factors 1 .. 100, noise / shifted-spot / constant / single-pixel image pairs, uint16 and float64
inputs, batches.  Shifts must be the oracle's exactly (they live on the 1 / upsample_factor grid); error and diffphase within 1e-9
(FFT factorizations round differently) - compared as error^2, which is what the rounding noise of an exact match
perturbs.   usage: python3 tools/fuzz_register.py [seed] [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from fluorosequencingimageanalysis_amd import phase_correlate as pc  # noqa: E402

O.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for t in range(cases):
    big = rng.random() < 0.08
    H = int(rng.integers(1500, 2100)) if big else int(rng.integers(4, 300))
    W = int(rng.integers(4, 64)) if big else int(rng.integers(4, 300))
    if rng.random() < 0.5:
        H, W = W, H
    uf = int(rng.choice([1, 2, 10, 20, 100]))
    kind = int(rng.integers(4))
    if kind == 0:
        ref = rng.integers(0, 4000, (H, W)).astype(np.uint16)
        reg = np.roll(ref, (int(rng.integers(-H // 2, H // 2 + 1)), int(rng.integers(-W // 2, W // 2 + 1))), (0, 1))
        reg = np.minimum(reg.astype(np.int64) + rng.integers(0, 40, (H, W)), 65535).astype(np.uint16)
    elif kind == 1:
        ref = rng.integers(0, 4000, (H, W)).astype(np.uint16); reg = rng.integers(0, 4000, (H, W)).astype(np.uint16)
    elif kind == 2:
        ref = np.zeros((H, W), np.uint16); reg = np.zeros((H, W), np.uint16)
        ref[rng.integers(0, H), rng.integers(0, W)] = 1000; reg[rng.integers(0, H), rng.integers(0, W)] = 700
    else:
        yy, xx = np.mgrid[0:H, 0:W]
        c = rng.uniform(0, [H, W], (2, 2))
        ref = (1000 * np.exp(-((yy - c[0, 0]) ** 2 + (xx - c[0, 1]) ** 2) / 8.0) + rng.integers(0, 20, (H, W))).astype(np.float64)
        reg = (1000 * np.exp(-((yy - c[1, 0]) ** 2 + (xx - c[1, 1]) ** 2) / 8.0) + rng.integers(0, 20, (H, W))).astype(np.float64)
    try:
        r = pc.phase_correlate(ref, reg, upsample_factor=uf)
    except Exception as e:        # noqa: BLE001
        bad += 1
        print("RAISED case %d: %s %s shape %s uf %d kind %d" % (t, type(e).__name__, e, (H, W), uf, kind), flush=True)
        continue
    e = O.phase_correlate(ref, reg, uf)
    ok = float(r[0]) == e[0] and float(r[1]) == e[1] and abs(float(r[2]) ** 2 - e[2] ** 2) < 1e-9 and \
        (abs(float(r[3]) - e[3]) < 1e-9 or abs(abs(float(r[3]) - e[3]) - 2 * np.pi) < 1e-9)
    if not ok:
        bad += 1
        print("DIFF case %d: shape %s uf %d kind %d: %r vs %r" % (t, (H, W), uf, kind, tuple(float(x) for x in r), e), flush=True)
# batches: several pairs of one shape in one call (the batched kernels index per-pair peaks, offsets and partial sums)
nb = 0
for t in range(cases // 3):
    H, W = int(rng.integers(4, 200)), int(rng.integers(4, 200))
    uf = int(rng.choice([1, 10, 20, 100]))
    k = int(rng.integers(2, 9))
    refs = rng.integers(0, 4000, (k, H, W)).astype(np.uint16)
    regs = np.stack([np.roll(refs[i], (int(rng.integers(-H // 2, H // 2 + 1)), int(rng.integers(-W // 2, W // 2 + 1))), (0, 1)) for i in range(k)])
    regs = np.minimum(regs.astype(np.int64) + rng.integers(0, 40, (k, H, W)), 65535).astype(np.uint16)
    if rng.random() < 0.3:
        regs[int(rng.integers(k))] = rng.integers(0, 4000, (H, W))          # one unrelated pair among them
    got = pc.phase_correlate_batch(refs, regs, uf)
    for i in range(k):
        e = O.phase_correlate(refs[i], regs[i], uf)
        r = got[i]
        ok = float(r[0]) == e[0] and float(r[1]) == e[1] and abs(float(r[2]) ** 2 - e[2] ** 2) < 1e-9 and \
            (abs(float(r[3]) - e[3]) < 1e-9 or abs(abs(float(r[3]) - e[3]) - 2 * np.pi) < 1e-9)
        nb += 1
        if not ok:
            bad += 1
            print("DIFF batch case %d pair %d of %d: shape %s uf %d: %r vs %r" % (t, i, k, (H, W), uf, tuple(float(x) for x in r), e), flush=True)
print("registration: %d cases + %d pairs in batches, %d differ from the oracle" % (cases, nb, bad), flush=True)
sys.exit(1 if bad else 0)

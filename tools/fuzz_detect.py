#!/usr/bin/env python3
"""One-off fuzz of the detection stage (fsq_detect behind pflib._psf_candidates) against the oracle (TEST TOOL): random image
shapes from 5x5 up, every median size 1..15 (even ones too: scipy's origin convention), random integer correlation matrices of
1x1 .. 15x15 with negative entries, c_std incl. 0 and negative, noise / sparse / saturated / constant images, batches of several
fields.  The oracle itself was checked against scipy.ndimage.median_filter + scipy.signal.correlate on the same kind of input
(300 random cases, 0 differences).   usage: python3 tools/fuzz_detect.py [seed] [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from fluorosequencingimageanalysis_amd import engine as E, pflib  # noqa: E402

O.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 400
bad = n_cand = 0
for t in range(cases):
    H, W = int(rng.integers(5, 200)), int(rng.integers(5, 200))
    nf = int(rng.integers(1, 5))
    kind = int(rng.integers(5))
    if kind == 0:
        imgs = rng.integers(0, int(rng.integers(2, 65536)), (nf, H, W))
    elif kind == 1:
        imgs = np.full((nf, H, W), int(rng.integers(0, 65536)))
    elif kind == 2:
        imgs = rng.integers(0, 50, (nf, H, W)); imgs[rng.random((nf, H, W)) < 0.02] = 65535
    elif kind == 3:
        imgs = np.minimum(rng.poisson(rng.uniform(1, 3000), (nf, H, W)), 65535)
    else:
        imgs = rng.integers(60000, 65536, (nf, H, W))
    imgs = imgs.astype(np.uint16)
    m = int(rng.integers(1, 16)) if rng.random() < 0.3 else int(rng.integers(1, 10))       # (1 .. FSQ_MAX_KSIZE = 15; the large ones are slow)
    ks = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15]))
    K = rng.integers(-6, 7, (ks, ks)).astype(np.int64) if rng.random() < 0.8 else pflib.default_correlation_matrix
    c = float(rng.choice([0.0, 0.5, 1.0, 2.0, 3.5, -1.0]))
    prm = E.detect_params(m, K, c)
    eng = E.Engine(nf, H, W, fit_workspace=False)
    total = eng.detect(E.to_device_u16(imgs), prm)
    cand, counts, _offs = eng.candidates(total)
    thr = eng.thr.cpu().numpy()
    off = 0
    for f in range(nf):
        hw, cm, th = O.candidates(imgs[f], m, K, c, return_cm=True)
        mine = cand[off:off + len(hw)]
        ok = (int(counts[f]) == len(hw)) and np.array_equal(mine[:, 1:], hw) and (mine[:, 0] == f).all() and \
            np.float64(thr[f]).view(np.uint64) == np.float64(th).view(np.uint64)
        if not ok:
            bad += 1
            print("DIFF case %d field %d: shape %s median %d kernel %d c_std %g: %d vs %d candidates, thr %r vs %r" % (t, f, (H, W), m, ks, c, int(counts[f]), len(hw), float(thr[f]), th), flush=True)
        off += int(counts[f])
        n_cand += len(hw)
print("detection: %d cases, %d candidates, %d fields differ from the oracle" % (cases, n_cand, bad), flush=True)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""One-off fuzz of the LM fit on stand-alone 5x5 ROIs against the oracle (TEST TOOL; results quoted in DESIGN.md): pixel patterns
no spot field produces - uniform noise over every dynamic range, single hot pixels, ramps, checkerboards, saturated plateaus with
a hole, two-level images, near-constant ROIs, Poisson spots of extreme widths - through fsq_fit_rois in the reference-faithful and
the textbook mode, every parameter / exit status / iteration count bit for bit.
usage: python3 tools/fuzz_rois.py [seed] [rois per family]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from fluorosequencingimageanalysis_amd import _native as N, engine as E  # noqa: E402

O.build()
if os.environ.get("FSQ_DEBUG_FORCE_NORM_RECOMPUTE"):       # (the kernels' debug switch: the oracle takes the same branch)
    O.lib().fsq_o_set_force_norm_recompute(1)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(seed)
yy, xx = np.mgrid[0:5, 0:5]
fam = {}
hi = rng.integers(1, 65536, per)
fam["uniform noise"] = (rng.random((per, 5, 5)) * hi[:, None, None]).astype(np.uint16)
a = rng.integers(0, 200, (per, 5, 5)); k = rng.integers(0, 25, per); a.reshape(per, 25)[np.arange(per), k] = rng.integers(200, 65536, per)
fam["hot pixel"] = a.astype(np.uint16)
g = rng.normal(0, 1, (per, 2)) * rng.integers(1, 4000, per)[:, None]
fam["ramps"] = np.clip(rng.integers(0, 30000, per)[:, None, None] + g[:, 0, None, None] * yy + g[:, 1, None, None] * xx + rng.normal(0, 3, (per, 5, 5)), 0, 65535).astype(np.uint16)
lo, hi2 = rng.integers(0, 1000, per), rng.integers(0, 65536, per)
fam["checkerboard"] = np.where(((yy + xx) % 2 == 0)[None], lo[:, None, None], hi2[:, None, None]).astype(np.uint16)
p = np.full((per, 5, 5), 65535, np.int64); kk = rng.integers(0, 25, (per, 3)); 
for j in range(3): p.reshape(per, 25)[np.arange(per), kk[:, j]] = rng.integers(0, 65535, per)
fam["saturated with holes"] = p.astype(np.uint16)
fam["two levels"] = np.where(rng.random((per, 5, 5)) < rng.random(per)[:, None, None], lo[:, None, None], hi2[:, None, None]).astype(np.uint16)
base = rng.integers(0, 65530, per)
fam["nearly constant"] = (base[:, None, None] + rng.integers(0, 3, (per, 5, 5)) * (rng.random(per) < 0.7)[:, None, None]).astype(np.uint16)
cy, cx = rng.uniform(-1, 5, per), rng.uniform(-1, 5, per); sy, sx = rng.uniform(0.2, 6, per), rng.uniform(0.2, 6, per)
amp, off = 10 ** rng.uniform(0, 4.8, per), rng.uniform(0, 3000, per)
m = off[:, None, None] + amp[:, None, None] * np.exp(-((yy - cy[:, None, None]) ** 2 / (2 * sy[:, None, None] ** 2) + (xx - cx[:, None, None]) ** 2 / (2 * sx[:, None, None] ** 2)))
fam["poisson spots of any width"] = np.clip(rng.poisson(m), 0, 65535).astype(np.uint16)
t0 = time.time()
total = bad = 0
for name, rois in fam.items():
    r = rois.reshape(-1, 25)
    for mode in (N.MODE_REF, N.MODE_TEXTBOOK):
        got, _ = E.fit_rois(r, mode)
        ref = O.fit_rois(r, mode, 16)
        same = np.ones(len(r), bool)
        for a_, b_ in zip(("H", "A", "p2", "p3", "sigma_h", "sigma_w", "theta"), range(7)):
            x, y = got[a_], np.ascontiguousarray(ref["p"][:, b_])
            same &= (x.view(np.uint64) == y.view(np.uint64)) | (np.isnan(x) & np.isnan(y))
        same &= (got["status"] == ref["status"]) & (got["niter"] == ref["niter"]) & (got["nfev"] == ref["nfev"])
        total += len(r); bad += int((~same).sum())
        print("%-28s mode %d: %d fits, %d differ; exits %s; slow-queue fits %d" % (name, mode, len(r), int((~same).sum()),
              dict(zip(*np.unique(ref["status"], return_counts=True))), N.lib().fsq_fit_last_slow_count()), flush=True)
print("ROIs: %d fits, %d differ from the oracle (%.0f s)" % (total, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Throughput of fsq_phase_correlate (config 3's registration step): 224 pairs of 512x512 frames (32 fields x 8 cycles,
7 consecutive pairs each), upsample_factor 20, inputs resident in HBM; wall time over the call (it synchronises)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluorosequencingimageanalysis_amd import _native as N  # noqa: E402
from fluorosequencingimageanalysis_amd import synth  # noqa: E402

H = W = 512
stacks = [synth.make_cycle_stack(100 + f, n_cycles=8, shape=(H, W), n_spots=500)[0] for f in range(4)]
ref = np.concatenate([s[:-1] for s in stacks]).astype(np.float64)
reg = np.concatenate([s[1:] for s in stacks]).astype(np.float64)
ref = np.tile(ref, (8, 1, 1))
reg = np.tile(reg, (8, 1, 1))
n = len(ref)
d_ref, d_reg = torch.from_numpy(ref).cuda(), torch.from_numpy(reg).cuda()
out = torch.empty((n, 4), dtype=torch.float64, device="cuda")
L = N.lib()
s = torch.cuda.current_stream().cuda_stream
for uf in (20, 1):
    N.check(L.fsq_phase_correlate(d_ref.data_ptr(), d_reg.data_ptr(), n, H, W, uf, out.data_ptr(), s), "pc")
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        N.check(L.fsq_phase_correlate(d_ref.data_ptr(), d_reg.data_ptr(), n, H, W, uf, out.data_ptr(), s), "pc")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    by = n * 60 * 2 ** 20            # ~60 MiB of algorithmic traffic per 512x512 pair (fp64 complex)
    print(json.dumps({"metric": "registration_pairs_per_sec", "upsample_factor": uf, "value": n / dt, "pairs": n, "ms": dt * 1e3,
                      "roofline": {"bound": "hbm", "achieved": by / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": by / dt / 1e9 / 8000.0}}))

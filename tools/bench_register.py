#!/usr/bin/env python3
"""Throughput of fsq_phase_correlate (config 3's registration step): 224 pairs of 512x512 uint16 frames (32 fields x 8
cycles, 7 consecutive pairs each) resident in HBM, upsample_factor 20 and 1.  bench.py --config 3 reports the same."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluorosequencingimageanalysis_amd import _native as N  # noqa: E402
from fluorosequencingimageanalysis_amd import engine as E, phase_correlate as PC, synth  # noqa: E402

H = W = 512
stacks = [synth.make_cycle_stack(100 + f, n_cycles=8, shape=(H, W), n_spots=500)[0] for f in range(4)]
ref = np.tile(np.concatenate([s[:-1] for s in stacks]), (8, 1, 1))
reg = np.tile(np.concatenate([s[1:] for s in stacks]), (8, 1, 1))
n = len(ref)
for dtype, name in ((N.DTYPE_U16, "u16"), (N.DTYPE_F64, "f64")):
    if dtype == N.DTYPE_U16:
        d_ref, d_reg = E.to_device_u16(ref), E.to_device_u16(reg)
    else:
        d_ref, d_reg = torch.from_numpy(ref.astype(np.float64)).cuda(), torch.from_numpy(reg.astype(np.float64)).cuda()
    for uf in (20, 1):
        R = PC.Registrar(n, H, W, uf, dtype)
        out = R.register(d_ref, d_reg)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 10
        for _ in range(reps):
            R.register(d_ref, d_reg, out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print(json.dumps({"metric": "registration_pairs_per_sec", "input": name, "upsample_factor": uf, "value": n / dt, "pairs": n,
                          "ms": dt * 1e3}))

#!/usr/bin/env python3
"""Throughput of fsq_mexican_hat on the bench workload's peak table: 1 024 fields of 512x512, ~520 kept
peaks each; HIP events around the kernel; algorithmic bytes = (2*9+1)^2 * 2 B read + 8 B written per spot."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluorosequencingimageanalysis_amd import _native as N  # noqa: E402
from fluorosequencingimageanalysis_amd import engine as E  # noqa: E402

n_fields, H, W, per = 1024, 512, 512, 520
rng = np.random.default_rng(0)
img = torch.from_numpy(rng.integers(0, 4000, (n_fields, H, W), dtype=np.int64).astype(np.uint16).view(np.int16)).cuda()
fhw = np.stack([np.repeat(np.arange(n_fields), per), rng.integers(0, H, n_fields * per), rng.integers(0, W, n_fields * per)], axis=1)
d = torch.from_numpy(fhw.astype(np.int32)).cuda()
out = torch.empty(len(fhw), dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
L = N.lib()
for _ in range(2):
    N.check(L.fsq_mexican_hat(img.data_ptr(), n_fields, H, W, d.data_ptr(), len(fhw), 6, 9, out.data_ptr(), s), "mh")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 10
for _ in range(reps):
    N.check(L.fsq_mexican_hat(img.data_ptr(), n_fields, H, W, d.data_ptr(), len(fhw), 6, 9, out.data_ptr(), s), "mh")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
by = len(fhw) * (361 * 2 + 8)
print(json.dumps({"metric": "mexican_hat_spots_per_sec", "value": len(fhw) / (ms * 1e-3), "spots": len(fhw), "ms": ms,
                  "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": by / (ms * 1e-3) / 1e9 / 8000.0}}))

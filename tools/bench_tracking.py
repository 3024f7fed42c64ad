#!/usr/bin/env python3
"""Throughput of the two tracking kernels on config-3-shaped work: 1 024 fields x 8 frames of 512x512 with ~500 spots per
frame (greedy tracking of peak tables), and 1 024 x 450 initial spots followed through 8 frames (luminosity centroid).
Kernel time by HIP events around the C-ABI calls (tables / frames resident in HBM)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluorosequencingimageanalysis_amd import _native as N, engine as E, synth  # noqa: E402

rng = np.random.default_rng(0)
n_fields, F, H, W, nsp = 1024, 8, 512, 512, 500
L = N.lib()
dev = torch.device("cuda")
# ---- greedy tracking: spot tables with sub-pixel drift, 15 % drop-out per frame --------------------------------------
counts = np.zeros((n_fields, F), np.int32)
parts, offs = [], np.zeros((n_fields, F, 2))
for k in range(n_fields):
    base = np.stack([rng.integers(8, H - 8, nsp), rng.integers(8, W - 8, nsp)], 1)
    base = base[np.unique((base[:, 0] // 3) * 1000 + base[:, 1] // 3, return_index=True)[1]]      # >= 3 px apart-ish
    cum = np.zeros(2)
    alive = np.ones(len(base), bool)
    for f in range(F):
        if f:
            step = np.round(rng.uniform(-3, 3, 2) * 20) / 20
            offs[k, f] = step
            cum = cum + step
            alive &= rng.uniform(size=len(base)) > 0.15
        pts = np.rint(base[alive] - cum).astype(np.int32)
        pts = pts[np.unique(pts[:, 0] * 4096 + pts[:, 1], return_index=True)[1]]
        counts[k, f] = len(pts)
        parts.append(pts)
hw = np.ascontiguousarray(np.concatenate(parts), dtype=np.int32)
start = np.concatenate([[0], np.cumsum(counts.sum(axis=1))]).astype(np.int32)
total = int(start[-1])
pair_cap = 8 * int(counts.max())
ws = torch.empty(L.fsq_track_workspace_bytes(n_fields, F, H, W, pair_cap), dtype=torch.uint8, device=dev)
t = lambda a: torch.from_numpy(a).to(dev)      # noqa: E731
d_hw, d_start, d_counts, d_off = t(hw.reshape(-1)), t(start), t(counts.reshape(-1)), t(offs.reshape(-1))
d_prev, d_next = torch.empty(total, dtype=torch.int32, device=dev), torch.empty(total, dtype=torch.int32, device=dev)
d_kept = torch.empty(total, dtype=torch.uint8, device=dev)
d_traces = torch.empty(total * F, dtype=torch.int32, device=dev)
d_nt, d_nd, d_st = (torch.empty(n_fields, dtype=torch.int32, device=dev) for _ in range(3))
s = torch.cuda.current_stream().cuda_stream


def track():
    N.check(L.fsq_greedy_tracking(d_hw.data_ptr(), d_start.data_ptr(), d_counts.data_ptr(), d_off.data_ptr(), n_fields, F, H, W, 2, 0.0,
                                  d_prev.data_ptr(), d_next.data_ptr(), d_kept.data_ptr(), d_traces.data_ptr(), d_nt.data_ptr(),
                                  d_nd.data_ptr(), d_st.data_ptr(), pair_cap, ws.data_ptr(), ws.numel(), s), "track")


track()
torch.cuda.synchronize()
assert int(d_st.abs().max()) == 0, "a field reported an error"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    track()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(json.dumps({"metric": "greedy_tracking_fields_per_sec", "value": n_fields / (ms * 1e-3), "ms_per_call": ms, "fields": n_fields,
                  "frames": F, "spots": total, "traces": int(d_nt.sum()), "spots_per_sec": total / (ms * 1e-3),
                  "note": "includes the 2 GB memset of the per-field bin grids"}))

# ---- luminosity-centroid tracking ---------------------------------------------------------------------------------------
nf2 = 64                                     # 64 fields x 8 frames of real synthetic frames, tiled spots
stacks = np.stack([synth.make_cycle_stack(200 + k, n_cycles=F, shape=(H, W), n_spots=nsp, max_drift=2.0, dropout=0.1)[0] for k in range(4)])
frames = np.tile(stacks, (nf2 // 4, 1, 1, 1))
d_frames = E.to_device_u16(frames)
init = np.stack([rng.integers(8, H - 8, nf2 * 450), rng.integers(8, W - 8, nf2 * 450)], 1).astype(np.int32)
fld = np.repeat(np.arange(nf2), 450).astype(np.int32)
d_init, d_fld = t(init.reshape(-1)), t(fld)
d_out = torch.empty((len(init), F, 2), dtype=torch.int32, device=dev)
d_pres = torch.empty((len(init), F), dtype=torch.uint8, device=dev)
d_err = torch.zeros(1, dtype=torch.int32, device=dev)


def centroid():
    N.check(L.fsq_centroid_tracking(d_frames.data_ptr(), nf2, F, H, W, d_init.data_ptr(), d_fld.data_ptr(), len(init), 3, 3.0, None,
                                    d_out.data_ptr(), d_pres.data_ptr(), d_err.data_ptr(), s), "centroid")


centroid()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    centroid()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
by = len(init) * (F - 1) * (49 + 25) * 2                # window + 5x5 area, uint16
print(json.dumps({"metric": "centroid_tracking_spot_frames_per_sec", "value": len(init) * (F - 1) / (ms * 1e-3), "ms_per_call": ms,
                  "spots": len(init), "frames": F,
                  "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": by / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_spot_frame": (49 + 25) * 2}}))

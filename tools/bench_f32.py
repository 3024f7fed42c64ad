"""The opt-in single-precision solver (FSQ_MODE_TEXTBOOK_F32, csrc/fsq_fit_f32.h) on the bench workload and on BASELINE configs[4]'s
field shape: whole-pipeline fits/s (detect -> fit -> consolidate, fields resident in HBM), the fit kernel alone, and how far its
results are from the fp64 textbook solver (FSQ_MODE_TEXTBOOK, which is the reference with MINPACK's qrsolv, bit for bit) - a
REPORT split by the fp64 solver's exit status, as SURVEY PROBE 9 does; nothing here is asserted.
usage: python3 tools/bench_f32.py [fields=1024] [steps=6]      prints one JSON line per measurement"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from fluorosequencingimageanalysis_amd import _native as N, engine as E, pflib  # noqa: E402

PARAMS = ("H", "A", "p2", "p3", "sigma_h", "sigma_w", "theta")


def agreement(rows32, rows64, keep64):
    p32 = np.stack([rows32[k] for k in PARAMS], 1)
    p64 = np.stack([rows64[k] for k in PARAMS], 1)
    e = (np.abs(p32 - p64) / np.maximum(np.abs(p64), 1e-12))[:, :6].max(1)        # (all but the angle)
    out = {"fits": int(len(e)), "within_1e-4": float((e <= 1e-4).mean()), "within_1e-3": float((e <= 1e-3).mean()),
           "kept": int(keep64.sum()), "kept_within_1e-4": float((e[keep64] <= 1e-4).mean()),
           "kept_within_1e-3": float((e[keep64] <= 1e-3).mean()), "by_fp64_exit_status": {}}
    for s in np.unique(rows64["status"]):
        m = rows64["status"] == s
        out["by_fp64_exit_status"][str(int(s))] = {"fits": int(m.sum()), "within_1e-4": float((e[m] <= 1e-4).mean()),
                                                   "kept": int((m & keep64).sum()),
                                                   "kept_within_1e-4": float((e[m & keep64] <= 1e-4).mean()) if (m & keep64).any() else None}
    return out


def run_engine(d_img, prm, mode, n, H, W):
    """-> (rows of all candidates, mask of the rows the R^2 filter keeps - pflib.py:466, before consolidation)"""
    eng = E.Engine(n, H, W)
    total = eng.run(d_img, prm, 0.7, 4, mode, True)
    torch.cuda.synchronize()
    rows = eng.rows[:total].cpu().numpy().view(N.ROW_DTYPE).reshape(-1)
    return rows, ~(rows["r2"] < 0.7)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda", 0)
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    # ---- the bench workload (configs[1]): n fields of 512 x 512, ~500 spots -----------------------------------------------
    imgs = bench.make_fields(range(n), (512, 512), 500)
    d_img = E.to_device_u16(imgs)
    pipe = E.StreamPipeline(n, 512, 512, depth=4, device=dev, mode=N.MODE_TEXTBOOK_F32)
    pipe.run([(d_img, prm)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    totals = pipe.run([(d_img, prm)] * steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.close()
    print(json.dumps({"metric": "textbook_f32_fits_per_sec", "value": sum(totals) / dt, "fields_per_sec": n * steps / dt,
                      "ms_per_step": dt / steps * 1e3,
                      "note": "detect -> single-precision LM -> consolidate over %d resident 512x512 fields per step, %d steps "
                              "(the headline's workload with FSQ_MODE_TEXTBOOK_F32)" % (n, steps)}), flush=True)
    # the fit kernel alone (kinit + kfit_f32 + kfinish), HIP events on its stream
    eng = E.Engine(n, 512, 512)
    total = eng.detect(d_img, prm)
    eng.fit(d_img, total, N.MODE_TEXTBOOK_F32)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        eng.fit(d_img, total, N.MODE_TEXTBOOK_F32)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"metric": "textbook_f32_fit_only_fits_per_sec", "value": total / (ms * 1e-3), "ms": ms, "fits": total}), flush=True)
    # agreement with the fp64 textbook solver on 64 of those fields
    m = min(64, n)
    r32, _ = run_engine(d_img[:m], prm, N.MODE_TEXTBOOK_F32, m, 512, 512)
    r64, k64 = run_engine(d_img[:m], prm, N.MODE_TEXTBOOK, m, 512, 512)
    print(json.dumps({"metric": "textbook_f32_agreement_512", "value": agreement(r32, r64, k64),
                      "note": "6 parameters (all but the angle), relative, against FSQ_MODE_TEXTBOOK on %d fields" % m}), flush=True)
    del d_img, eng
    torch.cuda.empty_cache()
    # ---- BASELINE configs[4]'s shape: 2048 x 2048 fields with 5000 spots, fp16 pixel loads ---------------------------------
    # (32 fields per step: one block of 16 waves consolidates a field - pflib's dict semantics are sequential per field)
    big = bench.make_fields(range(5000, 5032), (2048, 2048), 5000)
    b16, _scale = E.quantise_f16(big)
    imgs16, fmt = E.as_pixel_fields(b16)
    prm16 = E.detect_params(5, pflib.default_correlation_matrix, 2, fmt)
    d_big = E.to_device_u16(imgs16)
    r32, _ = run_engine(d_big, prm16, N.MODE_TEXTBOOK_F32, len(big), 2048, 2048)
    r64, k64 = run_engine(d_big, prm16, N.MODE_TEXTBOOK, len(big), 2048, 2048)
    print(json.dumps({"metric": "textbook_f32_agreement_cfg4", "value": agreement(r32, r64, k64),
                      "note": "%d fields of 2048x2048 with 5000 spots, fp16 pixel loads, against FSQ_MODE_TEXTBOOK on the same pixels" % len(big)}), flush=True)
    pipe = E.StreamPipeline(len(big), 2048, 2048, depth=4, device=dev, mode=N.MODE_TEXTBOOK_F32)
    pipe.run([(d_big, prm16)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    totals = pipe.run([(d_big, prm16)] * steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.close()
    print(json.dumps({"metric": "textbook_f32_cfg4_fits_per_sec", "value": sum(totals) / dt, "fields_per_sec": len(big) * steps / dt,
                      "note": "%d fields of 2048x2048 / 5000 spots per step, fp16 pixel loads + single-precision LM" % len(big)}), flush=True)
    # the REFERENCE-FAITHFUL solver on the same shape (fp16 pixel loads, fp64 mpfit arithmetic bit for bit): continuous batching
    # through two fit queues, as the headline is measured
    group = E.StreamPipelineGroup(len(big), 2048, 2048, queues=2, depth=6, device=dev)
    group.run([(d_big, prm16)])
    torch.cuda.synchronize()
    rsteps = max(2, steps)
    t0 = time.perf_counter()
    totals = group.run([(d_big, prm16)] * rsteps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    group.close()
    print(json.dumps({"metric": "ref_cfg4_fits_per_sec", "value": sum(totals) / dt, "fields_per_sec": len(big) * rsteps / dt,
                      "ms_per_step": dt / rsteps * 1e3,
                      "note": "BASELINE configs[4]'s shape with the reference-faithful fp64 solver: %d fields of 2048x2048 / 5000 spots per "
                              "step (%d candidates), fp16 pixel loads, detect + fit + consolidate, %d steps through two fit queues"
                              % (len(big), totals[0], rsteps)}), flush=True)


if __name__ == "__main__":
    main()

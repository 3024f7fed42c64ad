#!/usr/bin/env python3
"""bench.py - headline benchmark of the per-field hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): 1 024 synthetic 512x512 uint16 fields per GPU, ~500 spots each
(seeds rank*1024 .. +1023 of fluorosequencingimageanalysis_amd.synth), resident in HBM before the timed
region.  One step = detect -> LM-fit every candidate -> R^2 filter + consolidation over the whole batch
(+ the RCCL gather of the peak tables to rank 0 when N > 1).  value = candidate LM solves per second,
whole job.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_FIT = 2.0e5            # reference-faithful LM solve, fp64 (plus ~4e3 exp, not counted)
PEAK_FP64_VALU_TFLOPS = 78.6    # MI355X fp64 vector peak = 157.3 TFLOP/s fp32 vector / 2 (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def _make(seed_shape):
    from fluorosequencingimageanalysis_amd import synth
    seed, shape, n_spots = seed_shape
    return synth.make_field(seed, shape, n_spots)


def make_fields(seeds, shape, n_spots):
    import multiprocessing as mp
    n = min(16, os.cpu_count() or 1)
    pool = mp.get_context("fork").Pool(n)
    try:
        out = np.stack(pool.map(_make, [(s, shape, n_spots) for s in seeds], chunksize=8))
    finally:
        pool.close()        # let the workers leave on their own: Pool.terminate() SIGTERMs them, and under rocprofv3
        pool.join()         # the profiler's signal handler in a forked worker can block that forever
    return out


def cpu_baseline(imgs, cand, counts, offsets, n_threads):
    """The oracle (C restatement of the reference) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    n_fields = min(64, len(imgs))          # ~270k LM solves: about 25 s of CPU work (1.5 s on 16 threads)
    n = int(offsets[n_fields])
    c = cand[:n]
    rois = np.stack([imgs[f, h - 2:h + 3, w - 2:w + 3] for f, h, w in c]).reshape(-1, 25)
    O.fit_rois(rois[:256], n_threads=n_threads)        # warm-up
    t = time.perf_counter()
    O.fit_rois(rois, mode=0, n_threads=n_threads)
    dt = time.perf_counter() - t
    return {"value": n / dt, "unit": "fits/s", "cores": n_threads, "kind": "port",
            "sample": "LM fits of all %d candidates of the first %d fields of the workload, oracle/fsq_oracle.c "
                      "(reference-faithful fp64) on %d OpenMP threads, %.2f s" % (n, n_fields, n_threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fields", type=int, default=1024, help="fields per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spots", type=int, default=500)
    ap.add_argument("--lanes", type=int, default=2, help="shares of the batch worked side by side on their own streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import _native as N
    from fluorosequencingimageanalysis_amd import distributed as D
    from fluorosequencingimageanalysis_amd import engine as E
    from fluorosequencingimageanalysis_amd import pflib

    # synthetic fields first: the worker pool forks before this process has touched the GPU or opened a communicator
    shape = (a.size, a.size)
    env_rank = int(os.environ.get("RANK", "0"))
    imgs = make_fields(range(env_rank * a.fields, (env_rank + 1) * a.fields), shape, a.spots)

    rank, world, local = D.init_from_env()
    assert rank == env_rank
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU; the HIP path has no CPU fallback"
    local = local % torch.cuda.device_count()      # (more ranks than GPUs only in the gloo rehearsal of the N > 1 path)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    d_img = E.to_device_u16(imgs, dev)
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)

    # The batch is worked by `lanes` engines on their own HIP streams, lane k owning the k-th contiguous share of
    # the fields and starting half a pass after lane k-1, so that the latency-bound tail of one share's LM rounds
    # overlaps the busy early rounds of the other (engine.LanePipeline).  One step = every lane passes once over its
    # share = one pass over the whole batch.  --lanes 1 runs the batch as a single share.
    lanes = max(1, min(a.lanes, a.fields))
    cut = [a.fields * k // lanes for k in range(lanes + 1)]
    engs = [E.Engine(cut[k + 1] - cut[k], a.size, a.size, device=dev) for k in range(lanes)]
    d_share = [d_img[cut[k]:cut[k + 1]] for k in range(lanes)]
    pipe = E.LanePipeline(engs)
    base_ev = torch.cuda.Event(enable_timing=True)
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
          for _ in range(lanes)]
    totals = [[0] * max(a.steps, a.warmup, 1) for _ in range(lanes)]
    import queue
    outbox = queue.Queue()

    def work(timed):
        def f(k, i, eng):
            total = eng.detect(d_share[k], prm)
            if timed:
                ev[k][i][0].record()
            eng.fit(d_share[k], total, N.MODE_REF)
            if timed:
                ev[k][i][1].record()
            eng.consolidate(0.7, 4, True)
            totals[k][i] = total
            if world > 1:        # the one exchange of the path: this share's peak table, handed to the main thread
                nk = eng.nkeep.cpu().numpy()
                offs = eng.offsets.cpu().numpy()
                idx = np.concatenate([np.arange(offs[f_], offs[f_] + max(int(nk[f_]), 0)) for f_ in range(eng.n_fields)])
                kept = eng.rows[:total].index_select(0, eng.keep[:total].index_select(0, torch.from_numpy(idx).to(dev)).long())
                done = torch.cuda.Event()
                done.record()
                outbox.put((k, i, kept, done))
        return f

    def gather_all(n_steps):
        """Main thread: RCCL gather of every (lane, step) table to rank 0, in an order all ranks share."""
        pending = {}
        for i in range(n_steps):
            for k in range(lanes):
                while (k, i) not in pending:
                    kk, ii, kept, done = outbox.get()
                    pending[(kk, ii)] = (kept, done)
                kept, done = pending.pop((k, i))
                torch.cuda.current_stream().wait_event(done)
                kept.record_stream(torch.cuda.current_stream())
                D.gather_tables(kept, 0)

    def run_steps(n_steps, timed, stagger):
        if n_steps <= 0:
            return
        if world > 1:
            import threading
            th = threading.Thread(target=pipe.run, args=(work(timed), n_steps, stagger), daemon=True)
            th.start()
            gather_all(n_steps)
            th.join()
        else:
            pipe.run(work(timed), n_steps, stagger)

    # one untimed pass to page everything in and to learn the stagger (half a pass of one share)
    t = time.perf_counter()
    run_steps(1, False, 0.0)
    torch.cuda.synchronize()
    stagger = 0.5 * (time.perf_counter() - t) if lanes > 1 else 0.0
    run_steps(a.warmup, False, stagger)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    base_ev.record()
    t0 = time.perf_counter()
    run_steps(a.steps, True, stagger)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = sum(totals[k][0] for k in range(lanes))
    tt = torch.tensor([dt, float(total)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, total_all = float(tmax[0]), float(tsum[1])
    else:
        total_all = float(total)
    # LM-fit launches of the timed region: per-launch durations and the time during which at least one was running
    spans = sorted((base_ev.elapsed_time(ev[k][i][0]), base_ev.elapsed_time(ev[k][i][1])) for k in range(lanes) for i in range(a.steps))
    fit_ms = [e - b_ for b_, e in spans]
    busy_ms, cur_b, cur_e = 0.0, None, None
    for b_, e in spans:
        if cur_e is None or b_ > cur_e:
            busy_ms += (cur_e - cur_b) if cur_e is not None else 0.0
            cur_b, cur_e = b_, e
        else:
            cur_e = max(cur_e, e)
    busy_ms += (cur_e - cur_b) if cur_e is not None else 0.0
    if rank == 0:
        fit_avg_ms = float(np.mean(fit_ms))
        fits_per_s = total_all * a.steps / dt
        # `lanes` launches run side by side, each on a share of the CUs: the rate the chip sustains on this kernel is
        # the fits of all launches over the time at least one launch was running (= flops per launch / its duration
        # when lanes == 1)
        achieved = total * a.steps * FLOP_PER_FIT / (busy_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            # measured for one pass over the 1 024-field batch (rocprofv3 --pmc, tools/collect_profiles.sh); a launch
            # covers one lane's share of it
            traffic = json.load(open(tpath)).get("fit_kernel_hbm_bytes_per_1024_field_pass")
            if traffic is not None:
                traffic = traffic * (a.fields / 1024.0) / lanes
        out = {
            "metric": "psf_lm_fits_per_sec", "value": fits_per_s, "unit": "fits/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic %dx%d uint16 fields per GPU, %d spots each, "
                                   "detect + LM-fit every candidate + consolidate (reference-faithful fp64); the batch is worked as %d "
                                   "share(s) on their own HIP streams"
                                   % (a.fields, a.size, a.size, a.spots, lanes),
                       "fields_per_gpu": a.fields, "candidates_per_gpu": int(total),
                       "parallelism": "fields sharded over %d rank(s), RCCL p2p gather of peak tables" % world},
            "fields_per_sec": a.fields * world * a.steps / dt,
            "roofline": {"bound": "valu-fp64",
                         "note": "the LM solve is fp64 vector-ALU work with 98 B of algorithmic I/O per fit: there is no "
                                 "MFMA-shaped contraction and it is not HBM-bound (see the hbm entry), so it is priced against "
                                 "the fp64 VECTOR peak - DESIGN.md 4.2",
                         "kernel": "LM fit = kinit + rounds of (kA_jacobian, kB_step) + kfinish, timed as one unit",
                         "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                         "launch_ms": fit_avg_ms, "fits_per_launch": int(total // lanes), "concurrent_launches": lanes,
                         "launches": lanes * a.steps, "busy_ms": busy_ms, "flop_per_fit": FLOP_PER_FIT,
                         "hbm": {"algorithmic_bytes_per_fit": 98,
                                 "algorithmic_GBps": 98.0 * total * a.steps / (busy_ms * 1e-3) / 1e9,
                                 "traffic_GBps": (traffic * lanes * a.steps / (busy_ms * 1e-3) / 1e9) if traffic else None,
                                 "peak_GBps": PEAK_HBM_GBS}},
        }
        if not a.no_cpu_baseline:
            cand, counts, offsets = engs[0].candidates(totals[0][0])
            n_thr = min(16, len(os.sched_getaffinity(0)))     # the box's CPU share for one GPU
            out["cpu_baseline"] = cpu_baseline(imgs, cand, counts, offsets, n_thr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py - headline benchmark of the per-field hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): 1 024 synthetic 512x512 uint16 fields per GPU, ~500 spots each
(seeds rank*1024 .. +1023 of fluorosequencingimageanalysis_amd.synth), resident in HBM before the timed
region.  One step = detect -> LM-fit every candidate -> R^2 filter + consolidation over the whole batch
(+ the RCCL gather of the peak tables to rank 0 when N > 1).  The K timed steps are K batches streamed through
engine.StreamPipeline (continuous batching of the LM fits; --pipeline lanes runs every step stand-alone).
value = candidate LM solves per second, whole job.  Prints ONE JSON line on rank 0.
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
    ap.add_argument("--fields", type=int, default=1024, help="fields per GPU and step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spots", type=int, default=500)
    ap.add_argument("--pipeline", choices=("stream", "lanes"), default="stream",
                    help="stream: continuous batching of the LM fits across steps (engine.StreamPipeline); "
                         "lanes: every step a stand-alone batch, worked as --lanes shares (engine.LanePipeline)")
    ap.add_argument("--lanes", type=int, default=2, help="lanes pipeline: shares of the batch worked side by side")
    ap.add_argument("--queues", type=int, default=3,
                    help="stream pipeline: fit queues (each with its own streams and host thread) the fields of every step are "
                         "split over; the kernels of different queues overlap each other's ramp-down")
    ap.add_argument("--depth", type=int, default=16, help="stream pipeline: batches in flight at most")
    ap.add_argument("--inject-below", type=int, default=None, help="stream pipeline: submit the next batch once fewer fits are alive")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3),
                    help="2: BASELINE configs[1], detect + fit + consolidate (the headline); 3: the registration step of configs[2] "
                         "(phase correlation of consecutive cycle frames, upsample_factor 20) - pairs/s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the H2D-inclusive and dict-materialising side measurements")
    ap.add_argument("--materialise", action="store_true",
                    help="N > 1: every step's gathered peak RECORDS (378 B per peak, not just the 128 B rows) go to rank 0 and "
                         "rank 0 builds the reference's dicts from them inside the timed region (distributed.py's dict form)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process becomes the launcher of N ranks (one per GPU) and touches
        # neither the GPU nor torch.cuda itself; rank 0's JSON line is relayed and checked against the request
        sys.exit(launch_ranks(a))

    import torch
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import _native as N
    from fluorosequencingimageanalysis_amd import distributed as D
    from fluorosequencingimageanalysis_amd import engine as E
    from fluorosequencingimageanalysis_amd import pflib

    if a.config == 3:
        return bench_registration(a, torch, dist, D, E, N)
    # synthetic fields first: the worker pool forks before this process has touched the GPU or opened a communicator
    shape = (a.size, a.size)
    env_rank = int(os.environ.get("RANK", "0"))
    imgs = make_fields(range(env_rank * a.fields, (env_rank + 1) * a.fields), shape, a.spots)

    rank, world, local = D.init_from_env()
    assert rank == env_rank
    if world != a.gpus:         # never report a line for another job size than the one asked for
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s) (WORLD_SIZE); launch with `python bench.py "
                         "--gpus %d` or torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus, a.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU; the HIP path has no CPU fallback"
    backend = dist.get_backend() if world > 1 else "none"
    if world > torch.cuda.device_count() and backend != "gloo":
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (FSQ_DIST_BACKEND=gloo rehearses the N > 1 path on fewer GPUs)"
                         % (world, torch.cuda.device_count()))
    local = local % torch.cuda.device_count()      # (more ranks than GPUs only in the gloo rehearsal of the N > 1 path)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    d_img = E.to_device_u16(imgs, dev)
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)

    if a.pipeline == "stream":
        res = run_stream(a, torch, dist, D, E, N, d_img, imgs, prm, dev, rank, world)
    else:
        res = run_lanes(a, torch, dist, D, E, N, d_img, prm, dev, rank, world)
    dt, total, kept, busy_ms, fit_launches, how, cand_tables = res

    tt = torch.tensor([dt, float(total), float(kept)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, total_all, kept_all = float(tmax[0]), float(tsum[1]), float(tsum[2])
    else:
        total_all, kept_all = float(total), float(kept)
    if rank == 0:
        fits_per_s = total_all * a.steps / dt
        # the rate the chip sustains on the LM-fit kernels: the fits of all steps over the time those kernels had the
        # stream (stream pipeline: the fit queue's own stream, busy from the first to the last round of the timed
        # region; lanes: the time at least one lane's fit launch was running)
        achieved = total * a.steps * FLOP_PER_FIT / (busy_ms * 1e-3) / 1e12
        # PMC-derived figures (HBM bytes, issue fractions) cannot be collected inside this run (rocprofv3 --pmc serialises the
        # kernels): they come from profiles/fit_counters_latest.json, which records the hash of the kernel sources it was taken on, and are
        # only reported when the sources are still those - otherwise null with stale = true.
        prof, from_profile = {}, {"stale": True}
        ppath = os.path.join(ROOT, "profiles", "fit_counters_latest.json")
        lib_sha = N.source_sha16()
        if os.path.exists(ppath):
            prof = json.load(open(ppath))
            if prof.get("source_sha16") == lib_sha:
                from_profile = dict(prof, stale=False, note="not measured in this run: separate rocprofv3 --pmc passes of the same library "
                                                            "on %s fields per step (tools/collect_profiles.sh)" % prof.get("fields_per_step"))
            else:
                from_profile = {"stale": True, "source_sha16_of_profile": prof.get("source_sha16"), "source_sha16_now": lib_sha}
                prof = {}
        traffic = prof.get("fit_kernel_hbm_bytes_per_1024_field_step")
        if traffic is not None:
            traffic = traffic * (a.fields / 1024.0)
        out = {
            "metric": "psf_lm_fits_per_sec", "value": fits_per_s, "unit": "fits/s", "n_gpus": world,
            "ranks": (dist.get_world_size() if world > 1 else 1), "backend": backend,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic %dx%d uint16 fields per GPU, %d spots each, "
                                   "detect + LM-fit every candidate + consolidate (reference-faithful fp64); %s"
                                   % (a.fields, a.size, a.size, a.spots, how),
                       "fields_per_gpu": a.fields, "candidates_per_gpu": int(total),
                       "parallelism": "fields sharded over %d rank(s), RCCL p2p gather of peak tables" % world},
            "fields_per_sec": a.fields * world * a.steps / dt,
            "peaks_per_sec": kept_all * a.steps / dt,
            "roofline": {"bound": "valu-fp64",
                         "note": "the LM solve is fp64 vector-ALU work with 98 B of algorithmic I/O per fit: there is no "
                                 "MFMA-shaped contraction and it is not HBM-bound (see the hbm entry), so it is priced against "
                                 "the fp64 VECTOR peak - DESIGN.md 4.2",
                         "kernel": "LM fit = kinit + rounds of (kA_jacobian, kB_step) + kfinish, timed as one unit",
                         "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": traffic,
                         "fits_per_step": int(total), "busy_ms": busy_ms, "busy_ms_per_step": busy_ms / a.steps,
                         "launches": fit_launches, "flop_per_fit": FLOP_PER_FIT,
                         "from_profile": from_profile,
                         "hbm": {"algorithmic_bytes_per_fit": 98,
                                 "algorithmic_GBps": 98.0 * total * a.steps / (busy_ms * 1e-3) / 1e9,
                                 "traffic_GBps": (traffic * a.steps / (busy_ms * 1e-3) / 1e9) if traffic else None,
                                 "peak_GBps": PEAK_HBM_GBS}},
        }
        if not a.no_extras and world == 1:
            out["extras"] = extras(a, torch, E, N, pflib, imgs, d_img, prm, dev)
        if not a.no_cpu_baseline:
            cand, counts, offsets = cand_tables()
            n_thr = min(16, len(os.sched_getaffinity(0)))     # the box's CPU share for one GPU
            out["cpu_baseline"] = cpu_baseline(imgs, cand, counts, offsets, n_thr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_rank0_line(line, gpus):
    """The launcher's acceptance test of what rank 0 printed: one JSON object whose n_gpus / ranks equal the request."""
    j = json.loads(line)
    if j.get("n_gpus") != gpus or j.get("ranks", gpus) != gpus:
        raise ValueError("rank 0 reported n_gpus=%r ranks=%r for --gpus %d" % (j.get("n_gpus"), j.get("ranks"), gpus))
    return j


def launch_ranks(a, argv=None, run=None):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks of this script (torch.distributed.run, one
    process per GPU, rendezvous on 127.0.0.1) as CHILD processes - this process has not touched the GPU and never does -,
    relay rank 0's JSON line and return the children's exit code.  A line that does not carry n_gpus == N is an error (the
    reference's own pool never silently shrinks either: pflib.py:1082-1099 starts exactly num_processes workers)."""
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = (run or subprocess.run)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in (p.stdout or "").splitlines() if ln.startswith("{")]
    for ln in (p.stdout or "").splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        print("bench.py: the %d-rank job failed with exit code %d" % (a.gpus, p.returncode), file=sys.stderr)
        return p.returncode or 1
    if len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        return 1
    try:
        check_rank0_line(lines[0], a.gpus)
    except ValueError as e:
        print("bench.py: %s" % e, file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


REG_BYTES_PER_PX = 2 * (2 + 8) + 2 * 4 * 8 + (16 + 16 + 8) + 4 * 8 + 8 + 2 * 8      # = 180
# fp64 real-transform registration, per pixel of a pair: 2 images uint16 -> float64 (2 B in, 8 B out each); two forward
# real-to-complex FFTs of two passes each (8 B in + 8 B out per pass: a half spectrum is 8 B per pixel); cross-power
# (two half spectra in, one out); the inverse transform (two passes); the peak search (8 B); the upsampled DFT's one
# read of both half spectra.  = 180 B per pixel = 45 MiB per 512x512 pair ("halve with R2C").


def measure_registration(torch, E, N, size, spots, steps, warmup, rank=0, barrier=None):
    """configs[2]'s registration step: consecutive cycle frames of channel 0 registered at upsample_factor 20
    (SequenceExperiment.offsets_from_frames, flexlibrary.py:1717-1741): 32 fields x 7 pairs of uint16 frames per step, resident
    in HBM.  -> dict(pairs, wall seconds for `steps` calls, kernel ms per call, algorithmic bytes per call)."""
    from fluorosequencingimageanalysis_amd import phase_correlate as PC
    from fluorosequencingimageanalysis_amd import synth
    H = W = size
    stacks = [synth.make_cycle_stack(100 + 7 * rank + f, n_cycles=8, shape=(H, W), n_spots=spots)[0] for f in range(4)]
    ref = np.tile(np.concatenate([s_[:-1] for s_ in stacks]), (8, 1, 1))
    reg = np.tile(np.concatenate([s_[1:] for s_ in stacks]), (8, 1, 1))
    n = len(ref)
    d_ref, d_reg = E.to_device_u16(ref), E.to_device_u16(reg)
    R = PC.Registrar(n, H, W, 20, N.DTYPE_U16)
    out = R.register(d_ref, d_reg)
    for _ in range(warmup):
        R.register(d_ref, d_reg, out)
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        R.register(d_ref, d_reg, out)
    ev1.record()
    if barrier is not None:
        barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"pairs": n, "seconds": dt, "launch_ms": ev0.elapsed_time(ev1) / steps, "bytes_per_call": REG_BYTES_PER_PX * H * W * n}


def bench_registration(a, torch, dist, D, E, N):
    """python bench.py --config 3: the registration step alone.  value = pairs per second."""
    rank, world, local = D.init_from_env()
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    H = W = a.size
    m = measure_registration(torch, E, N, a.size, a.spots, a.steps, a.warmup, rank, lambda: _barrier(torch, dist, world))
    n, dt = m["pairs"], m["seconds"]
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    if rank == 0:
        ms = m["launch_ms"]
        by = m["bytes_per_call"]
        print(json.dumps({
            "metric": "registration_pairs_per_sec", "value": n * world * a.steps / dt, "unit": "pairs/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2] registration step: %d pairs of consecutive %dx%d uint16 cycle frames per GPU and "
                                   "step (32 fields x 7 pairs), phase correlation at upsample_factor 20" % (n, H, W),
                       "pairs_per_gpu": n, "parallelism": "pairs sharded over %d rank(s), no exchange" % world},
            "roofline": {"bound": "hbm", "kernel": "fsq_phase_correlate as one unit (rocFFT real transforms + cross-power, peak, "
                                                   "upsampled-DFT kernels)", "achieved": by / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                         "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                         "algorithmic_bytes_per_pair": REG_BYTES_PER_PX * H * W, "launch_ms": ms}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _barrier(torch, dist, world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def run_stream(a, torch, dist, D, E, N, d_img, imgs, prm, dev, rank, world):
    """K steps = K batches streamed through engine.StreamPipelineGroup: continuous batching of the LM fits (a step's
    detection, its fits and its consolidation all happen inside the timed region, but the slow fits of one step finish
    inside the round launches of the following steps instead of in hundreds of nearly empty launches of their own), the
    fields of every step split over `--queues` fit queues on their own streams so that their kernels overlap each other's
    ramp-down.  With N > 1 every step's kept tables go to rank 0 (the one exchange of the path), issued by the main thread
    in (step, queue) order while the queues' threads keep working."""
    import queue as _queue
    import threading
    group = E.StreamPipelineGroup(a.fields, a.size, a.size, queues=a.queues, depth=a.depth, inject_below=a.inject_below, device=dev)
    Q = len(group.pipes)
    kept = [0] * Q
    box = _queue.Queue()
    built = [0]
    from fluorosequencingimageanalysis_amd import pflib as pflib_mod

    def on_done(j, k, eng, total):
        kept[k] = eng.nkeep[eng.n_fields]           # (device scalar; read after the run)
        if world > 1:
            if a.materialise:                       # the full peak records (FsqRow + fit_img + sub_img) and the per-field counts
                table = eng.peak_records(d_img[group.cut[k]:group.cut[k + 1]])[0]
                nk = eng.nkeep[:eng.n_fields].clone().reshape(-1, 1)
            else:
                table, nk = eng.kept_table()[0], None
            ev = torch.cuda.Event()
            ev.record()
            box.put((j, k, table, ev, nk))

    def steps(n):
        res, errs = [None], []

        def body():
            try:
                res[0] = group.run([(d_img, prm)] * n, on_done)
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
                box.put(None)
        if world == 1:
            return group.run([(d_img, prm)] * n, on_done)
        th = threading.Thread(target=body, daemon=True)
        th.start()
        ready = {}
        for j in range(n):
            for k in range(Q):
                while (j, k) not in ready:
                    item = box.get()
                    if item is None:
                        raise errs[0]
                    ready[(item[0], item[1])] = item[2:]
                table, ev, nk = ready.pop((j, k))
                torch.cuda.current_stream().wait_event(ev)
                tab, _ = D.gather_tables(table, 0)
                if nk is not None:                  # --materialise: rank 0 turns every gathered table into the reference's dicts
                    cnt, _ = D.gather_tables(nk, 0)
                    if rank == 0:
                        built[0] += len(pflib_mod.records_to_dicts(tab.cpu().numpy(), cnt.cpu().numpy().reshape(-1)))
        th.join()
        if errs:
            raise errs[0]
        return res[0]

    steps(1)                            # page everything in
    steps(a.warmup)
    _barrier(torch, dist, world)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(Q)]
    r0 = [p.queue.rounds for p in group.pipes]
    for k in range(Q):
        ev[k][0].record(group.pipes[k].fit_stream)
    t0 = time.perf_counter()
    totals = steps(a.steps)
    for k in range(Q):
        ev[k][1].record(group.pipes[k].fit_stream)
    _barrier(torch, dist, world)
    dt = time.perf_counter() - t0
    busy_ms = max(ev[k][0].elapsed_time(ev[k][1]) for k in range(Q))        # the queues' streams are busy side by side
    rounds = [group.pipes[k].queue.rounds - r0[k] for k in range(Q)]
    how = ("the steps are streamed through %d fit queue(s) on their own streams, the fields of a step split between them "
           "(continuous batching: at most %d batches in flight per queue, next batch submitted below %d live fits; %s rounds "
           "for %d steps)" % (Q, group.pipes[0].depth, group.pipes[0].inject_below, rounds, a.steps))
    if world > 1:
        how += ("; every step's %s gathered to rank 0 inside the timed region"
                % ("peak records (378 B per peak) and built into the reference's dicts there (--materialise: %d field dicts)" % built[0]
                   if a.materialise else "kept rows (128 B per peak)"))

    def cand_tables():
        e = E.Engine(a.fields, a.size, a.size, device=dev, fit_workspace=False)
        return e.candidates(e.detect(d_img, prm))
    return dt, totals[0], sum(int(x) for x in kept), busy_ms, {"rounds": rounds, "steps": a.steps, "queues": Q}, how, cand_tables


def run_lanes(a, torch, dist, D, E, N, d_img, prm, dev, rank, world):
    """Every step a stand-alone batch, worked by `lanes` engines on their own HIP streams / host threads, lane k owning
    the k-th contiguous share of the fields and starting half a pass after lane k-1 (engine.LanePipeline).  Kept for A/B."""
    assert world == 1, "--pipeline lanes is a single-GPU A/B path"
    lanes = max(1, min(a.lanes, a.fields))
    cut = [a.fields * k // lanes for k in range(lanes + 1)]
    engs = [E.Engine(cut[k + 1] - cut[k], a.size, a.size, device=dev) for k in range(lanes)]
    d_share = [d_img[cut[k]:cut[k + 1]] for k in range(lanes)]
    pipe = E.LanePipeline(engs)
    base_ev = torch.cuda.Event(enable_timing=True)
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
          for _ in range(lanes)]
    totals = [[0] * max(a.steps, a.warmup, 1) for _ in range(lanes)]

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
        return f

    t = time.perf_counter()
    pipe.run(work(False), 1, 0.0)
    torch.cuda.synchronize()
    stagger = 0.5 * (time.perf_counter() - t) if lanes > 1 else 0.0
    if a.warmup > 0:
        pipe.run(work(False), a.warmup, stagger)
    _barrier(torch, dist, world)
    base_ev.record()
    t0 = time.perf_counter()
    pipe.run(work(True), a.steps, stagger)
    _barrier(torch, dist, world)
    dt = time.perf_counter() - t0
    total = sum(totals[k][0] for k in range(lanes))
    spans = sorted((base_ev.elapsed_time(ev[k][i][0]), base_ev.elapsed_time(ev[k][i][1])) for k in range(lanes) for i in range(a.steps))
    busy_ms, cur_b, cur_e = 0.0, None, None
    for b_, e in spans:
        if cur_e is None or b_ > cur_e:
            busy_ms += (cur_e - cur_b) if cur_e is not None else 0.0
            cur_b, cur_e = b_, e
        else:
            cur_e = max(cur_e, e)
    busy_ms += (cur_e - cur_b) if cur_e is not None else 0.0
    kept = sum(int(e.nkeep[e.n_fields]) for e in engs)
    how = "every step a stand-alone batch worked as %d share(s) on their own HIP streams" % lanes
    return (dt, total, kept, busy_ms, {"fit_calls": lanes * a.steps}, how,
            lambda: engs[0].candidates(totals[0][0]))


def extras(a, torch, E, N, pflib, imgs, d_img, prm, dev):
    """Side measurements SURVEY 8d lists next to the headline, taken in the same run: the stream pipeline with every step's
    images uploaded inside the timed region; the dict-materialising pflib surface; the command line end to end (TIFF -> pickle +
    CSV); the registration step of configs[2]; the tracking and photometry kernels; the opt-in single-precision
    solver of configs[4] with its distance from the fp64 textbook solver (tools/bench_f32.py)."""
    import shutil
    import subprocess
    import tempfile
    out = {}
    n = min(a.steps, 6)
    pinned = torch.from_numpy(imgs.view(np.int16)).pin_memory()
    bufs = [torch.empty_like(d_img) for _ in range(3)]
    pipe = E.StreamPipeline(a.fields, a.size, a.size, depth=min(a.depth, 8), inject_below=a.inject_below, device=dev)

    def jobs(k):
        for j in range(k):
            b = bufs[j % 3]
            torch.cuda.current_stream().wait_stream(pipe.fit_stream)   # kinit of the batch that used this buffer
            b.copy_(pinned, non_blocking=True)
            yield b, prm

    pipe.run(jobs(1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(jobs(n))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.close()
    del pipe, bufs, pinned
    out["h2d_inclusive_fields_per_sec"] = a.fields * n / dt
    out["h2d_inclusive_note"] = "%d steps, each step's %d fields copied from pinned host memory (%.0f MiB) on the side stream" % (
        n, a.fields, imgs.nbytes / 2**20)
    # the drop-in surface: host stack in, list of dicts of 12-tuples out (H2D, streamed chunks, D2H, Python objects)
    pflib.find_peptides_batch(imgs[:256])
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        d = pflib.find_peptides_batch(imgs)
        dt = time.perf_counter() - t0
        npk = sum(len(x) for x in d)
        del d
        best = dt if best is None else min(best, dt)
    out["find_peptides_batch_fields_per_sec"] = len(imgs) / best
    out["find_peptides_batch_note"] = ("pflib.find_peptides_batch on %d host fields -> list of dicts of 12-tuples (pinned H2D, chunks "
                                       "of %d fields fitted by stand-alone passes in 3 lanes, D2H, dicts built by a worker thread "
                                       "while the next chunks are fitted; %d peaks)" % (len(imgs), pflib.CHUNK_PIXELS // (a.size * a.size), npk))
    pflib.find_peptides_records(imgs[:256])                 # (builds this path's pipeline, as the 256-field call above did for the lanes)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        rec, counts, _fmt = pflib.find_peptides_records(imgs)
        dt = time.perf_counter() - t0
        del rec
        best = dt if best is None else min(best, dt)
    out["find_peptides_records_fields_per_sec"] = len(imgs) / best
    out["find_peptides_records_note"] = "the same stack through the continuous-batching pipeline, returned as byte tables (no Python objects per peak)"
    # the command line end to end: a directory of 16-bit TIFFs -> pickle + CSV per image
    m = min(256, len(imgs))
    tmp = tempfile.mkdtemp(prefix="fsq_bench_cli_")
    try:
        from PIL import Image
        from fluorosequencingimageanalysis_amd import basic_image_script as cli
        for i in range(m):
            Image.fromarray(imgs[i]).save(os.path.join(tmp, "field%04d.tif" % i), format="TIFF")
        t0 = time.perf_counter()
        res = cli.main(["-L", os.path.join(tmp, "log.txt"), tmp])
        dt = time.perf_counter() - t0
        out["cli_images_per_sec"] = len(res) / dt
        out["cli_note"] = ("basic_image_script over %d synthetic 16-bit TIFFs of %dx%d in one process: TIFF -> PNG conversion, read, "
                           "fit, protocol-0 pickle (%.1f MB per image) + CSV per image; %d images processed"
                           % (m, a.size, a.size, os.path.getsize(next(iter(res.values()))[1]) / 1e6 if res else 0.0, len(res)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    pflib.release_gpu_resources()
    # registration (configs[2]'s step), 10 calls
    r = measure_registration(torch, E, N, a.size, a.spots, 10, 2)
    out["registration_pairs_per_sec"] = r["pairs"] * 10 / r["seconds"]
    out["registration_hbm_frac"] = r["bytes_per_call"] / (r["launch_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS
    out["registration_note"] = "%d pairs of %dx%d uint16 frames per call, upsample_factor 20, %.2f ms of kernel time per call" % (
        r["pairs"], a.size, a.size, r["launch_ms"])
    # tracking and photometry kernels (their own processes: tools/bench_tracking.py, tools/bench_photometry.py)
    torch.cuda.empty_cache()
    for script in ("bench_tracking.py", "bench_photometry.py", "bench_f32.py"):
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], capture_output=True, text=True, timeout=400)
            for line in p.stdout.splitlines():
                if line.startswith("{"):
                    j = json.loads(line)
                    key = {"greedy_tracking_fields_per_sec": "tracking_fields_per_sec",
                           "centroid_tracking_spot_frames_per_sec": "centroid_tracking_spot_frames_per_sec",
                           "mexican_hat_spots_per_sec": "photometry_spots_per_sec"}.get(j["metric"], j["metric"])
                    out[key] = j["value"]
                    if script == "bench_f32.py" and "note" in j:
                        out[key + "_note"] = j["note"]
            if p.returncode != 0:
                out[script + "_error"] = p.stderr[-300:]
        except Exception as e:      # noqa: BLE001 - a side measurement must not take the headline down
            out[script + "_error"] = repr(e)
    return out


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:      # noqa: BLE001
        if isinstance(e, SystemExit) and not e.code:
            raise
        # a failed rank must not leave its peers waiting in a collective: report and leave without the interpreter's
        # orderly shutdown (which would block in the process group's destructor); torchrun then ends the job non-zero
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)

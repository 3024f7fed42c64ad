"""Drop-in for the per-field image path of the reference's `pflib` module, computed on an MI355X.

Same names, keyword arguments, defaults, return shapes and exceptions as the reference functions
(file:line of each is given in its docstring); the arithmetic is done by hand-written HIP kernels
behind the C ABI of include/fsq.h and reproduces the reference's fp64 results bit for bit (see
DESIGN.md).  There is no CPU fallback: without libfsq_hip.so / a GPU every compute entry point raises.

Batch extension (not in the reference): `find_peptides_batch(images, ...)` runs many same-sized
fields in one pass and is what bench.py and the multi-GPU driver use.
"""
import atexit
import collections
import concurrent.futures
import csv
import gc
import logging
import math
import os
import pickle
import threading
import time

import numpy as np

from . import _native as N
from . import engine as _engine

logger = logging.getLogger(__name__)
logger.addHandler(logging.NullHandler())

default_correlation_matrix = _engine.DEFAULT_CORRELATION_MATRIX.copy()      # pflib.py:48-52

#: The reference is Python 2: dict keys use round-half-away-from-zero (pflib.py:515).
PY2_ROUND = True


# ---- GPU resources kept between calls ---------------------------------------------------------------------------------
# Building an Engine / StreamPipeline means allocating its workspaces (GBs for a pipeline); the drop-in functions are called
# image after image, batch after batch, so the objects are kept in a small process-wide LRU cache keyed by the batch shape.
_CACHE = collections.OrderedDict()
_CACHE_LOCK = threading.RLock()
MAX_CACHED_RESOURCES = 4


def _cached(key, factory):
    with _CACHE_LOCK:
        if key in _CACHE:
            _CACHE.move_to_end(key)
            return _CACHE[key]
        while len(_CACHE) >= MAX_CACHED_RESOURCES:
            _, old = _CACHE.popitem(last=False)
            close = getattr(old, "close", None)
            if close is not None:
                close()
        obj = _CACHE[key] = factory()
        return obj


def release_gpu_resources():
    """Free the cached engines / pipelines (they are re-created on demand)."""
    with _CACHE_LOCK:
        while _CACHE:
            _, old = _CACHE.popitem()
            close = getattr(old, "close", None)
            if close is not None:
                close()


def _device_key():
    torch = _engine._torch()
    return torch.cuda.current_device()


def _pixel_max(words, fmt):
    """Largest pixel value of a PIXELS_U32 stack (bounds the detection's integer domain); None for the 16-bit formats."""
    if fmt != N.PIXELS_U32 or not words.size:
        return None
    vmax = int(words.max())
    if vmax >= 2 ** 31:
        raise NotImplementedError("pixel values outside [0, 2^31) are not supported by the GPU path")
    return vmax


def _chunk_params(prm, host_words):
    """The detection parameters of ONE chunk of a PIXELS_U32 stack: a copy of `prm` with pixel_bits from the chunk's own maximum
    (taken by the stager thread on the words it has just copied - the only pass over the pixels the host makes)."""
    vmax = _pixel_max(host_words, N.PIXELS_U32)
    out = N.FsqDetectParams.from_buffer_copy(prm)
    out.pixel_bits = max(1, int(vmax).bit_length())
    return out


def _psf_candidates(image, median_filter_size=5, correlation_matrix=default_correlation_matrix, c_std=2, **kwargs):
    """Candidate pixels for PSF fitting, as a list [(h, w), ...] in raster order.  Reference pflib.py:217-258."""
    img, fmt = _engine.as_pixel_fields(image)
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std, fmt, _pixel_max(img, fmt))  # ValueError as pflib.py:236-239
    if img.ndim != 2:
        raise ValueError("image must be two-dimensional")
    H, W = img.shape
    if H < 5 or W < 5:              # no pixel has a 5x5 neighbourhood: the reference's loop over range(2, H - 2) is empty (pflib.py:252)
        return []
    eng = _cached(("detect", _device_key(), H, W), lambda: _engine.Engine(1, H, W, fit_workspace=False))
    with _CACHE_LOCK:
        total = eng.detect(_engine.to_device_pixels(img, fmt), prm)
        cand, _, _ = eng.candidates(total)
    return [(int(h), int(w)) for _, h, w in cand]


def _2d_gaussian_function(H, A, h_0, w_0, sigma_h, sigma_w, theta, h, w):
    """The circular Gaussian the reference's Monte-Carlo fitter evaluates (pflib.py:93-114: sigma_w and theta are accepted and
    not used there either): A * exp(-((h - h_0)^2 + (w - w_0)^2) / (2 sigma_h^2)) + H.  Host-side NumPy, as in the reference."""
    a = (h - h_0) ** 2
    b = (w - w_0) ** 2
    return A * np.exp(-np.divide(a + b, 2 * sigma_h ** 2)) + H


def _fit_2d_gaussian_monte_carlo(subimage, N_iter=10**3):
    """pflib.py:117-177 draws N_iter parameter sets from numpy's global, unseeded random state and keeps the best: its output
    differs from run to run in the reference itself, so there is nothing to reproduce."""
    raise NotImplementedError("fit_type='monte_carlo' draws from an unseeded RNG in the reference (pflib.py:117-177) and is not reproduced")


def _fit_2d_gaussian(subimage, implementation='agpy'):
    """Fit a 2D Gaussian to a 5x5 pixel area -> (h_0, w_0, H, A, sigma_h, sigma_w, theta, fit_img).
    Reference pflib.py:180-214 (h_0/w_0 in the 5x5 frame, as the reference returns them)."""
    subimage = np.asarray(subimage)
    assert subimage.shape[0] == 5 and subimage.shape[1] == 5
    if implementation != 'agpy':
        raise NotImplementedError("Currently, only agpy is supported.")
    torch = _engine._torch()
    ws = _cached(("roi1", _device_key()), lambda: _Roi1Workspace())
    with _CACHE_LOCK:
        r, fit = ws.fit(subimage)
    return (float(r["p2"]), float(r["p3"]), float(r["H"]), float(r["A"]), float(r["sigma_h"]), float(r["sigma_w"]),
            float(r["theta"]), fit)


class _Roi1Workspace:
    """Device buffers of _fit_2d_gaussian (one 5x5 ROI per call)."""

    def __init__(self):
        torch = _engine._torch()
        self.torch = torch
        self.rows = torch.empty((1, 128), dtype=torch.uint8, device="cuda")
        self.ws = torch.empty(N.lib().fsq_fit_workspace_bytes(1), dtype=torch.uint8, device="cuda")
        self.fit_img = torch.empty((1, 25), dtype=torch.float64, device="cuda")

    def fit(self, subimage):
        torch = self.torch
        d = _engine.to_device_u16(_engine.as_u16_fields(subimage).reshape(1, 25))
        s = torch.cuda.current_stream().cuda_stream
        N.check(N.lib().fsq_fit_rois(d.data_ptr(), 1, N.MODE_REF, self.rows.data_ptr(), self.ws.data_ptr(), self.ws.numel(), s),
                "fsq_fit_rois")
        N.check(N.lib().fsq_fit_images(self.rows.data_ptr(), None, 1, self.fit_img.data_ptr(), s), "fsq_fit_images")
        return self.rows.cpu().numpy().view(N.ROW_DTYPE).reshape(-1)[0], self.fit_img.cpu().numpy().reshape(5, 5)


def illumina_s_n(sub_img):
    """(max(sub_img) - mean(edge)) / std(edge) over the one-pixel boundary.  Reference pflib.py:261-281."""
    sub_img = np.asarray(sub_img)
    if not (len(sub_img.shape) == 2 and sub_img.shape[0] == sub_img.shape[1]):
        raise ValueError("sub_img must be square, but has shape " + str(sub_img))
    n = sub_img.shape[0]
    edge = ([sub_img[h, w] for h in (0, -1) for w in range(n)] +
            [sub_img[h, w] for h in range(1, n - 1) for w in (0, -1)])
    return (np.amax(sub_img) - np.mean(edge)) / np.std(edge)


#: _records_to_dicts converts about this many peaks per interpreter call (see there)
DICT_SLICE_PEAKS = 2048


try:                                    # host-side C extension (csrc/fsq_pyhost.c, built by the Makefile): the same objects, twice as fast
    from . import _fsq_pyhost
except ImportError:                     # (not built: the interpreter does the same work)
    _fsq_pyhost = None


def _records_to_dicts(rows, fit, sub, offs, failed=(), pixel_format=N.PIXELS_U16):
    """Peak records of a batch -> one {(h, w): 12-tuple} per field (AssertionError instances for the fields in `failed`).
    With the records themselves (fit and sub None, engine.peak_record_view) the C builder does it, a few fields per call so
    that other threads get the interpreter in between; _records_to_dicts_py is the same in Python (and what the C builder
    is tested against)."""
    if (fit is not None or _fsq_pyhost is None or not rows.flags.c_contiguous
            or rows.dtype != (_engine.RECORD_DTYPE_U32 if pixel_format == N.PIXELS_U32 else _engine.RECORD_DTYPE)):
        return _records_to_dicts_py(rows, fit, sub, offs, failed, pixel_format)
    was_enabled = gc.isenabled()
    gc.disable()                        # (millions of fresh containers, none of them cyclic)
    try:
        raw = rows.view(np.uint8).reshape(-1, _engine.peak_record_bytes(pixel_format))
        o = np.ascontiguousarray(offs, dtype=np.int64)
        n_fields = len(o) - 1
        out = []
        f0 = 0
        while f0 < n_fields:
            f1 = f0 + 1
            while f1 < n_fields and o[f1 + 1] - o[f0] <= DICT_SLICE_PEAKS:
                f1 += 1
            out.extend(_fsq_pyhost.fields_to_dicts(raw, o, f0, f1, int(pixel_format)))
            f0 = f1
        for f in failed:
            out[f] = AssertionError("field %d: re-keyed peak collides with an existing key (pflib.py:518)" % f)
        return out
    finally:
        if was_enabled:
            gc.enable()


def _records_to_dicts_py(rows, fit, sub, offs, failed=(), pixel_format=N.PIXELS_U16):
    """Peak records of a batch -> one {(h, w): 12-tuple} per field, in the reference's
    dict order (pflib.py:396-407, 475, 514-519); fields listed in `failed` give an AssertionError instance instead.
    Built column-wise: the value types are the reference's (numpy.float64 scalars, a Python float for rmse, 5x5 arrays -
    views of the batch's sub_img / fit_img blocks), the tuples are zipped together and the dicts filled by the interpreter's
    C loops; the cyclic garbage collector is held off meanwhile (millions of fresh containers, none of them cyclic).
    The table is worked off a few fields (DICT_SLICE_PEAKS peaks) at a time: every list() / zip() below is ONE call into the
    interpreter's C code, during which no other thread can take the interpreter - and the thread that drives the GPU pipeline
    needs it for a moment after each of its library calls.  With whole 60 000-peak chunks per call it waited milliseconds every
    time and the GPU ran dry (the dict-building call measured 1 490 fields/s against 2 900 for the same call without dicts).
    rows: FsqRow table with fit / sub as separate float64 / int64 [k, 5, 5] blocks (engine.split_peak_records), or - fit and
    sub None - the records themselves (engine.peak_record_view): the 5x5 blocks are then copied out slice by slice too."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        offs = [int(x) for x in offs]
        n_fields = len(offs) - 1
        out = []
        f0 = 0
        while f0 < n_fields:
            f1 = f0 + 1
            while f1 < n_fields and offs[f1 + 1] - offs[f0] <= DICT_SLICE_PEAKS:
                f1 += 1
            a, b = offs[f0], offs[f1]
            part = rows[a:b]
            cols = [list(np.ascontiguousarray(part[k])) for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")]
            rmse = part["rmse"].tolist()
            r2 = list(np.ascontiguousarray(part["r2"]))
            s_n = list(np.ascontiguousarray(part["s_n"]))
            if fit is None:
                fit_blk = np.ascontiguousarray(part["fit"])
                sub_blk = _engine.pixel_values(np.ascontiguousarray(part["sub"]), pixel_format).reshape(-1, 5, 5)
            else:
                fit_blk, sub_blk = fit[a:b], sub[a:b]
            tuples = list(zip(*cols, list(sub_blk), list(fit_blk), rmse, r2, s_n))
            keys = list(zip(part["key_h"].tolist(), part["key_w"].tolist()))
            for f in range(f0, f1):
                if f in failed:
                    out.append(AssertionError("field %d: re-keyed peak collides with an existing key (pflib.py:518)" % f))
                else:
                    out.append(dict(zip(keys[offs[f] - a:offs[f + 1] - a], tuples[offs[f] - a:offs[f + 1] - a])))
            f0 = f1
        return out
    finally:
        if was_enabled:
            gc.enable()


def _engine_dicts(eng, d_img, pixel_format=N.PIXELS_U16):
    """Consolidated results of an Engine pass -> list of per-field dicts (AssertionError instances for failed fields)."""
    rec, offs = eng.peak_records(d_img)
    nkeep = eng.nkeep.cpu().numpy()
    rows, fit, sub = _engine.split_peak_records(rec.cpu().numpy(), pixel_format)
    failed = set(int(f) for f in np.nonzero(nkeep[:eng.n_fields] < 0)[0])
    return _records_to_dicts(rows, fit, sub, offs.cpu().numpy(), failed)


def _wide_pass(eng, imgs, prm, r_2_threshold, radius, mode):
    """find_peptides of a PIXELS_U32 stack on a caller-owned Engine: one stand-alone pass (detect -> fit -> consolidate on 32-bit
    pixels), the kept table and fit images copied back, sub_img cut from the host copy (pflib.py:443)."""
    if len(imgs) != eng.n_fields or imgs.shape[1:] != (eng.H, eng.W):
        raise ValueError("the Engine was built for %d fields of %d x %d" % (eng.n_fields, eng.H, eng.W))
    d = np.arange(-2, 3)
    d_img = _engine.to_device_pixels(imgs, N.PIXELS_U32)
    eng.run(d_img, prm, r_2_threshold, radius, mode, PY2_ROUND)
    nkeep = eng.nkeep.cpu().numpy()
    table, offs = eng.kept_table()
    fit = eng.fit_images(table).cpu().numpy().reshape(-1, 5, 5)
    rows = table.cpu().numpy().view(N.ROW_DTYPE).reshape(-1)
    sub = imgs[rows["field"][:, None, None], (rows["h"][:, None] + d)[:, :, None], (rows["w"][:, None] + d)[:, None, :]].astype(np.int64)
    failed = set(int(f) for f in np.nonzero(nkeep[:eng.n_fields] < 0)[0])
    return _records_to_dicts(rows, fit, sub, offs.cpu().numpy(), failed)


#: stacks of at most this many pixels - one image, a few small ones - are worked by ONE library call on the calling thread
#: (_small_pass) instead of the _BatchRunner's stager / lanes / builder threads, which only pay off over several chunks
SMALL_PASS_PIXELS = 8 * 512 * 512


def _small_pass(imgs, fmt, prm, r_2_threshold, radius, mode):
    """find_peptides of a small stack: upload, fsq_find_peptides (the whole path in one call, engine.PathRunner), records back,
    dicts - on the calling thread, with a cached runner of the stack's shape (a single image used to set up four
    pinned buffers, three lanes and their threads)."""
    n, H, W = imgs.shape
    wide = fmt == N.PIXELS_U32
    def build():
        r = _engine.PathRunner(n, H, W, record_bytes=_engine.peak_record_bytes(fmt))
        r.lock = threading.Lock()           # (its buffers serve one call at a time; other shapes' runners are not held up)
        return r
    runner = _cached(("small", _device_key(), n, H, W, wide), build)
    with runner.lock:
        rec, offs, nk, _ = runner.run(_engine.to_device_pixels(imgs, fmt), prm, r_2_threshold, radius, mode, PY2_ROUND)
        rec, offs, nk = rec.cpu().numpy(), offs.cpu().numpy(), nk.cpu().numpy()
    failed = set(int(f) for f in np.nonzero(nk[:n] < 0)[0])
    return _records_to_dicts(_engine.peak_record_view(rec, fmt), None, None, offs, failed, fmt)


#: find_peptides_batch streams a stack through the GPU in chunks of about this many pixels (engine.StreamPipeline)
CHUNK_PIXELS = 128 * 512 * 512
#: (kept for callers that tuned it: the largest stack that used to be worked as ONE pass)
MAX_PIXELS_PER_PASS = 1024 * 512 * 512


class _RunnerClosed(Exception):
    pass


# The interpreter's thread switch interval is process-global: the batch runners shorten it while they run (their pipeline
# thread needs the interpreter for a few calls per chunk; while the worker builds dicts it would wait a whole interval - 5 ms by
# default - for each of them).  Runs of different runners may overlap, so the original value is kept once, with a count of
# the runs in progress, and restored when the last of them ends (a per-call save / restore could leave it short).
_SWITCH = {"n": 0, "saved": None, "lock": threading.Lock()}
BATCH_SWITCH_INTERVAL = 1e-4


def _switch_interval_acquire():
    import sys
    with _SWITCH["lock"]:
        if _SWITCH["n"] == 0:
            _SWITCH["saved"] = sys.getswitchinterval()
            sys.setswitchinterval(BATCH_SWITCH_INTERVAL)
        _SWITCH["n"] += 1


def _switch_interval_release():
    import sys
    with _SWITCH["lock"]:
        _SWITCH["n"] -= 1
        if _SWITCH["n"] == 0 and _SWITCH["saved"] is not None:
            sys.setswitchinterval(_SWITCH["saved"])
            _SWITCH["saved"] = None


def _run_cached_runner(key, factory, *args, **kw):
    """runner.run(...) on the cached runner of `key`; a runner another thread's call has just evicted (and closed) is replaced."""
    while True:
        runner = _cached(key, factory)
        try:
            return runner.run(*args, **kw)
        except _RunnerClosed:
            continue


#: stand-alone passes side by side when find_peptides_batch builds dicts (see _BatchRunner)
DICT_LANES = 3
#: ... whose first chunks are this many times smaller than the rest (one entry per group of DICT_LANES chunks)
DICT_RAMP = "2"


class _BatchRunner:
    """find_peptides over a stack of same-shaped fields: chunks of `per` fields are uploaded through pinned staging buffers,
    fitted, and their peak records are copied back and turned into dicts by a worker thread while the GPU works on the following
    chunks.  The object (workspaces, staging buffers) lives in the module cache between calls.
    Two ways through the GPU.  Record tables (raw=True) and the single-precision solver: engine.StreamPipeline - continuous
    batching, the slow fits of one chunk finish inside the round launches of the next; fastest for the GPU, but every chunk
    holds a few fits that run to maxiter, so ALL chunks of a call complete together at the very end.  Dicts: DICT_LANES
    stand-alone passes side by side (each lane its own Engine, stream and thread, a chunk complete when its own slowest fit is):
    ~10 % slower on the GPU, but the chunks complete evenly spaced, and the 0.23 s the interpreter needs to build 490 000
    12-tuples overlap the fitting instead of following it (1 024 fields: 0.60 s -> 0.45 s)."""

    def __init__(self, per, H, W, mode=N.MODE_REF, wide=False):
        """wide: the runner of PIXELS_U32 stacks (uint32 staging buffers, 428-byte records, a fit queue for 32-bit pixels)."""
        torch = _engine._torch()
        self.torch = torch
        self.per, self.H, self.W = int(per), int(H), int(W)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.mode = mode
        self.wide = bool(wide)
        self.rec_bytes = _engine.PEAK_RECORD_BYTES_U32 if self.wide else _engine.PEAK_RECORD_BYTES
        self.pipe = None                        # (both built on first use)
        self.lane_engines = self.lane_streams = None
        # four staging buffers: the stager runs at most two chunks ahead of the pipeline thread (queue of 2), so the buffer of
        # chunk c is written again (chunk c + 4) only after the pipeline thread has recorded the upload event of chunk c
        self.pin = [torch.empty((self.per, H, W), dtype=torch.int32 if self.wide else torch.int16).pin_memory() for _ in range(4)]
        self.pin_ev = [None] * 4
        self.pin_free = [threading.Event() for _ in range(4)]   # set: the chunk staged in the buffer has been handed to the GPU
        for e in self.pin_free:
            e.set()
        self.rec_pin = None                     # pinned landing buffer of the peak records (dict-building calls)
        self.lock = threading.Lock()
        self.closed = False

    def close(self):
        """(called by the cache when it evicts the runner; waits for a call that is using it)"""
        with self.lock:
            if not self.closed:
                self.closed = True
                if self.pipe is not None:
                    self.pipe.close()

    def run(self, words, fmt, prm, r_2_threshold, radius, on_chunk=None, raw=False, device=False):
        """words: uint16[n, H, W] (host).  -> list of n dicts / AssertionError instances.  on_chunk(first, dicts), if given, is
        called (in the worker thread) as soon as the dicts of fields first .. first + len(dicts) - 1 exist.
        raw=True: -> (peak records uint8[k, engine.PEAK_RECORD_BYTES] in field order, int32[n] peaks per field, -1 where the
        re-key assertion fired) instead of dicts - the form the multi-GPU gather ships.  raw=True, device=True: the records stay
        where fsq_find_peptides' kernels wrote them - a uint8 DEVICE tensor (torch) instead of a host array; only the per-field
        counts come to the host (the RCCL gather sends device memory: no D2H -> H2D bounce, distributed.py)."""
        import queue as _queue
        import sys
        torch, per = self.torch, self.per
        n_lanes = 0 if (raw or self.mode == N.MODE_TEXTBOOK_F32) else int(os.environ.get("FSQ_BATCH_LANES", DICT_LANES))
        if self.wide != (fmt == N.PIXELS_U32):
            raise ValueError("this runner was built for %s pixels" % ("32-bit" if self.wide else "16-bit"))
        n = len(words)
        if n_lanes > 0:
            # lanes take chunks of any size up to `per`: the first ones are small so that the first records - and with them the
            # worker that builds the dicts, which is the longest chain of the call - start early
            sizes, left = [], n
            ramp = [int(x) for x in os.environ.get("FSQ_BATCH_RAMP", DICT_RAMP).split(",") if x]
            for want in [max(1, per // dv) for dv in ramp for _ in range(n_lanes)]:
                if left > per:
                    sizes.append(min(want, left))
                    left -= sizes[-1]
            while left > 0:
                sizes.append(min(per, left))
                left -= sizes[-1]
        else:
            sizes = [min(per, n - c * per) for c in range(-(-n // per))]
        first = [0]
        for m in sizes:
            first.append(first[-1] + m)
        n_chunks = len(sizes)
        out = [None] * (n if n_lanes > 0 else n_chunks * per)
        bufs = {}
        pool = concurrent.futures.ThreadPoolExecutor(1)
        futures = []
        staged = _queue.Queue(maxsize=2)

        def run_lanes():
            """Stand-alone passes instead of the continuous-batching pipeline: every lane (its own Engine, stream and thread) takes
            the next staged chunk, runs detect -> fit -> consolidate on it and hands its peak records to the worker.  A chunk is
            complete when its own slowest fit is, so the chunks finish evenly spaced in time and the worker builds the dicts of
            one while the lanes fit the next."""
            errs = []

            def body(k):
                try:
                    torch.cuda.set_device(self.dev)
                    eng = self.lane_engines[k]
                    with torch.cuda.stream(self.lane_streams[k]):
                        while True:
                            item = staged.get()
                            if item is None or isinstance(item, BaseException):
                                staged.put(item)            # (the end marker / the stager's error is for every lane)
                                if isinstance(item, BaseException):
                                    raise item
                                return
                            c, i, prm_c = item
                            d = self.pin[i][:sizes[c]].to(self.dev, non_blocking=True)
                            self.pin_ev[i] = torch.cuda.Event()
                            self.pin_ev[i].record()
                            self.pin_free[i].set()
                            # the whole path of the chunk as one library call (no interpreter between the stages: the
                            # worker that builds the dicts has it); the runner's buffers are re-used, so what the worker
                            # will copy to the host is cloned
                            rec, offs, nk, _ = eng.run(d, prm_c, r_2_threshold, radius, self.mode, PY2_ROUND)
                            # records and counts go to pinned host buffers on this lane's stream right away (the runner's
                            # device buffers are free for its next chunk, stream order); the worker only waits for the event
                            m, kk = sizes[c], int(rec.shape[0])
                            j = self.land_free.get()
                            if self.land_rec[j] is None or self.land_rec[j].shape[0] < kk:
                                self.land_rec[j] = torch.empty((kk + kk // 4 + 1024, self.rec_bytes), dtype=torch.uint8).pin_memory()
                            if self.land_meta[j] is None or self.land_meta[j].shape[0] < 2 * per + 2:
                                self.land_meta[j] = torch.empty(2 * per + 2, dtype=torch.int32).pin_memory()
                            self.land_rec[j][:kk].copy_(rec, non_blocking=True)
                            self.land_meta[j][:2 * m + 1].copy_(torch.cat([nk[:m], offs[:m + 1]]), non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record()
                            futures.append(pool.submit(materialise_landed, c, j, kk, m, ev))
                        self.lane_streams[k].synchronize()
                except BaseException as e:      # noqa: BLE001 - re-raised by the caller
                    errs.append(e)

            ths = [threading.Thread(target=body, args=(k,), daemon=True) for k in range(n_lanes)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            if errs:
                raise errs[0]

        stop = threading.Event()        # set when the call is over (normally or not): the stager leaves at its next wait

        def put_staged(item):
            while not stop.is_set():
                try:
                    staged.put(item, timeout=0.05)
                    return True
                except _queue.Full:
                    pass
            return False

        def stage():        # (own thread) copies the chunks into the pinned staging buffers ahead of the pipeline
            try:
                for c in range(n_chunks):
                    part = words[first[c]:first[c + 1]]
                    i = c % 4
                    while not self.pin_free[i].wait(0.05):  # (the consumer of the chunk last staged here has issued its upload ...)
                        if stop.is_set():                   # (... or the GPU side has failed and nobody ever will)
                            return
                    if stop.is_set():
                        return
                    self.pin_free[i].clear()
                    if self.pin_ev[i] is not None:
                        self.pin_ev[i].synchronize()        # (... and the upload is done)
                    host = self.pin[i].numpy().view(np.uint32 if self.wide else np.uint16)
                    host[:len(part)] = part
                    if len(part) < per and n_lanes == 0:    # (pipeline engines have a fixed field count: the last chunk is filled up with copies of its last field)
                        host[len(part):] = part[-1]
                    prm_c = _chunk_params(prm, host[:len(part)]) if self.wide else prm
                    if not put_staged((c, i, prm_c) if n_lanes > 0 else (i, prm_c)):
                        return
                if n_lanes > 0:
                    put_staged(None)
            except BaseException as e:      # noqa: BLE001 - handed to the pipeline thread
                put_staged(e)

        def jobs():
            for c in range(n_chunks):
                item = staged.get()
                if isinstance(item, BaseException):
                    raise item
                i, prm_c = item
                bufs[c] = self.pin[i].to(self.dev, non_blocking=True)
                self.pin_ev[i] = torch.cuda.Event()
                self.pin_ev[i].record()
                self.pin_free[i].set()
                yield bufs[c], prm_c

        def materialise(c, rec, offs, nk, ev):
            with torch.cuda.device(self.dev):
                ev.synchronize()
                if raw:
                    rec_host = rec if device else rec.cpu().numpy()
                else:       # through a pinned buffer that is re-used (this thread works one chunk at a time): no fresh pages
                    k = rec.shape[0]
                    if self.rec_pin is None or self.rec_pin.shape[0] < k:
                        self.rec_pin = torch.empty((k + k // 4 + 1024, self.rec_bytes), dtype=torch.uint8).pin_memory()
                    self.rec_pin[:k].copy_(rec, non_blocking=True)
                meta = torch.cat([nk.to(torch.int32), offs.to(torch.int32)]).cpu().numpy()    # (synchronises this thread's stream)
                nk, offs = meta[:len(nk)], meta[len(nk):]
            if raw:
                out[c] = (rec_host, np.where(nk < 0, nk, np.diff(offs)).astype(np.int32))
                return
            failed = set(int(f) for f in np.nonzero(nk < 0)[0])
            dicts = _records_to_dicts(_engine.peak_record_view(self.rec_pin[:k].numpy(), fmt), None, None, offs, failed, fmt)
            if n_lanes > 0:
                out[first[c]:first[c + 1]] = dicts
            else:
                out[c * per:(c + 1) * per] = dicts
            if on_chunk is not None:
                on_chunk(first[c], dicts[:sizes[c]])

        def materialise_landed(c, j, kk, m, ev):            # (lanes: the chunk's records are already on their way to land_rec[j])
            try:
                ev.synchronize()
                meta = self.land_meta[j][:2 * m + 1].numpy()
                nk, offs = meta[:m].copy(), meta[m:].copy()
                failed = set(int(f) for f in np.nonzero(nk < 0)[0])
                dicts = _records_to_dicts(_engine.peak_record_view(self.land_rec[j][:kk].numpy(), fmt), None, None, offs, failed, fmt)
            finally:
                self.land_free.put(j)
            out[first[c]:first[c + 1]] = dicts
            if on_chunk is not None:
                on_chunk(first[c], dicts[:sizes[c]])

        def on_done(c, eng, total):                         # (side stream current, the chunk consolidated on it)
            rec, offs = eng.peak_records(bufs.pop(c))
            nk = eng.nkeep[:per].clone()
            ev = torch.cuda.Event()
            ev.record()
            futures.append(pool.submit(materialise, c, rec, offs, nk, ev))

        with self.lock:
            if self.closed:                 # evicted from the cache by another thread between look-up and call: the caller fetches a new one
                pool.shutdown(wait=False)
                raise _RunnerClosed()
            for e in self.pin_free:         # (a call that failed may have left a buffer marked busy)
                e.set()
            self.land_free = _queue.Queue()
            for j in range(n_lanes + 2):
                self.land_free.put(j)
            if n_lanes > 0 and (self.lane_engines is None or len(self.lane_engines) != n_lanes):
                self.lane_engines = [_engine.PathRunner(self.per, self.H, self.W, device=self.dev, record_bytes=self.rec_bytes)
                                     for _ in range(n_lanes)]
                self.lane_streams = [torch.cuda.Stream(device=self.dev) for _ in range(n_lanes)]
                self.land_rec = [None] * (n_lanes + 2)      # pinned landing buffers of the chunks' records / counts
                self.land_meta = [None] * (n_lanes + 2)
            if n_lanes == 0 and self.pipe is None:
                self.pipe = _engine.StreamPipeline(self.per, self.H, self.W, depth=12, device=self.dev, mode=self.mode, wide=self.wide)
            stager = threading.Thread(target=stage, daemon=True)
            # (the pipeline thread needs the interpreter for a few calls per chunk; while the worker builds dicts it would
            # wait a whole switch interval - 5 ms by default - for each of them)
            _switch_interval_acquire()
            try:
                stager.start()
                if n_lanes > 0:
                    run_lanes()
                else:
                    self.pipe.run(jobs(), on_done, r_2_threshold, radius, PY2_ROUND)
            finally:
                _switch_interval_release()
                stop.set()                                  # (after a failure the stager may be waiting for a buffer or for room
                stager.join()                               # in the queue: both waits look at `stop`, so it leaves within 50 ms)
                for e in self.pin_free:
                    e.set()
                pool.shutdown(wait=True)
        for f in futures:
            f.result()                                      # re-raises what the worker raised
        if raw:
            # (the padding fields of the last chunk are cut off: their records sit at the end of that chunk's table)
            counts = np.concatenate([out[c][1] for c in range(n_chunks)])[:n]
            recs = []
            for c in range(n_chunks):
                k = int(np.maximum(out[c][1][:max(0, min(per, n - c * per))], 0).sum())
                recs.append(out[c][0][:k])
            if device:
                with torch.cuda.device(self.dev):
                    for r in recs:          # (allocated on the pipeline's side stream, read by the concatenation on this thread's)
                        r.record_stream(torch.cuda.current_stream(self.dev))
                    return (torch.cat(recs) if recs else torch.zeros((0, self.rec_bytes), dtype=torch.uint8, device=self.dev)), counts
            return np.concatenate(recs) if recs else np.zeros((0, self.rec_bytes), np.uint8), counts
        return out[:n]


SOLVERS = {"reference": N.MODE_REF, "textbook": N.MODE_TEXTBOOK, "textbook_f32": N.MODE_TEXTBOOK_F32}


def _solver_mode(solver):
    try:
        return SOLVERS[solver]
    except KeyError:
        raise ValueError("solver must be one of %s" % ", ".join(sorted(SOLVERS)))


def find_peptides_batch(images, median_filter_size=5, correlation_matrix=default_correlation_matrix,
                        candidate_pixels=None, c_std=2, r_2_threshold=0.7, consolidation_radius=4,
                        fit_type='gauss', N_iter=10**3, engine=None, errors='raise', on_chunk=None, solver='reference'):
    """find_peptides over a stack uint16[n, H, W] -> list of n dicts.

    The stack is streamed through the GPU in chunks of about CHUNK_PIXELS pixels (engine.StreamPipeline: continuous
    batching of the LM fits) by a _BatchRunner that stays alive between calls.  errors='return' puts the AssertionError of a
    field whose re-key collides (pflib.py:518) in that field's place instead of raising it.  on_chunk(first_index, dicts)
    is called as soon as a chunk's dicts exist (from a worker thread).
    solver (an extension; the reference has one solver): 'reference' - mpfit as the reference runs it, bit for bit (default);
    'textbook' - the same with MINPACK's qrsolv; 'textbook_f32' - the opt-in single-precision approximation of
    BASELINE configs[4] (csrc/fsq_fit_f32.h: NOT the reference's numbers - 61 % of its kept fits lie within 1e-4 of the fp64
    solver, 80 % within 1e-3, parity unpinned: configs[4]'s "fp32 LM accumulate" is not met as a parity path)."""
    mode = _solver_mode(solver)
    if consolidation_radius < 2:
        raise ValueError("consolidation_radius must be at least 2")                # pflib.py:431-432
    if fit_type != 'gauss':
        raise NotImplementedError("fit_type='monte_carlo' draws from an unseeded RNG in the reference "
                                  "(pflib.py:117-177) and is not reproduced")
    # (candidate_pixels: "Not yet implemented" in the reference, pflib.py:374 - accepted and ignored there and here)
    imgs, fmt = _engine.as_pixel_fields(images)            # integer dtypes, floats holding integer values, or float16
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std, fmt)   # (PIXELS_U32: pixel_bits per chunk / small pass)
    if imgs.ndim != 3:
        raise ValueError("images must have shape (n, H, W)")
    n, H, W = imgs.shape
    if n == 0:
        return []
    if H < 5 or W < 5:              # (no candidates: pflib.py:252)
        return [{} for _ in range(n)]
    wide = fmt == N.PIXELS_U32                              # pixel values beyond 16 bits: uint32 words, 428-byte records
    if wide and mode == N.MODE_TEXTBOOK_F32:
        raise NotImplementedError("solver='textbook_f32' takes 16-bit pixels only")
    if wide and engine is not None:
        out = _wide_pass(engine, imgs, _chunk_params(prm, imgs), r_2_threshold, consolidation_radius, mode)
    elif engine is not None:                                # (a caller-owned Engine: one stand-alone pass)
        d_img = _engine.to_device_u16(imgs)
        engine.run(d_img, prm, r_2_threshold, consolidation_radius, mode, PY2_ROUND)
        out = _engine_dicts(engine, d_img, fmt)
    elif (n * H * W <= min(SMALL_PASS_PIXELS, CHUNK_PIXELS) and mode != N.MODE_TEXTBOOK_F32
          and "FSQ_BATCH_LANES" not in os.environ):            # (one chunk, and a small one)
        out = _small_pass(imgs, fmt, _chunk_params(prm, imgs) if wide else prm, r_2_threshold, consolidation_radius, mode)
        if on_chunk is not None:
            on_chunk(0, out)
    else:
        n_chunks = max(1, -(-(n * H * W) // CHUNK_PIXELS))
        per = -(-n // n_chunks)
        out = _run_cached_runner(("batch", _device_key(), per, H, W, mode, wide), lambda: _BatchRunner(per, H, W, mode, wide),
                                 imgs, fmt, prm, r_2_threshold, consolidation_radius, on_chunk)
    if errors == 'raise':
        for d in out:
            if isinstance(d, Exception):
                raise d
    return out


def find_peptides_records(images, median_filter_size=5, correlation_matrix=default_correlation_matrix, c_std=2,
                          r_2_threshold=0.7, consolidation_radius=4, solver='reference', device=False, wide=None, **unused):
    """find_peptides over a stack, results as the byte tables the multi-GPU gather ships instead of dicts:
    -> (records uint8[k, engine.peak_record_bytes(pixel format)] of all fields in order, int32[n] peaks per field (-1: the re-key
    assertion of pflib.py:518 fired for that field), pixel format).  records_to_dicts turns them into find_peptides' dicts.
    device=True: the records are returned as a torch uint8 tensor in HBM (never copied to the host).
    wide=True: integer pixels are worked as uint32 (428-byte records) even when every value fits 16 bits - for callers whose
    parts must agree on one record format (the ranks of find_peptides_sharded)."""
    mode = _solver_mode(solver)
    if consolidation_radius < 2:
        raise ValueError("consolidation_radius must be at least 2")
    if unused.get("fit_type", "gauss") != 'gauss':
        raise NotImplementedError("fit_type='monte_carlo' draws from an unseeded RNG in the reference "
                                  "(pflib.py:117-177) and is not reproduced")
    imgs, fmt = _engine.as_pixel_fields(images, wide=bool(wide))
    is_wide = fmt == N.PIXELS_U32
    if is_wide and mode == N.MODE_TEXTBOOK_F32:
        raise NotImplementedError("solver='textbook_f32' takes 16-bit pixels only")
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std, fmt)        # (PIXELS_U32: pixel_bits per chunk)
    if imgs.ndim != 3:
        raise ValueError("images must have shape (n, H, W)")
    n, H, W = imgs.shape
    if n == 0 or H < 5 or W < 5:
        if device:
            torch = _engine._torch()
            return torch.zeros((0, _engine.peak_record_bytes(fmt)), dtype=torch.uint8, device="cuda"), np.zeros(n, np.int32), fmt
        return np.zeros((0, _engine.peak_record_bytes(fmt)), np.uint8), np.zeros(n, np.int32), fmt
    n_chunks = max(1, -(-(n * H * W) // CHUNK_PIXELS))
    per = -(-n // n_chunks)
    rec, counts = _run_cached_runner(("batch", _device_key(), per, H, W, mode, is_wide), lambda: _BatchRunner(per, H, W, mode, is_wide),
                                     imgs, fmt, prm, r_2_threshold, consolidation_radius, raw=True, device=bool(device))
    return rec, counts, fmt


def records_to_dicts(records, counts, pixel_format=N.PIXELS_U16):
    """The inverse packaging of find_peptides_records: -> list of dicts (AssertionError instances for failed fields)."""
    counts = np.asarray(counts).reshape(-1)
    failed = set(int(k) for k in np.nonzero(counts < 0)[0])
    offs = np.concatenate([[0], np.cumsum(np.maximum(counts, 0))])
    return _records_to_dicts(_engine.peak_record_view(records, pixel_format), None, None, offs, failed, pixel_format)


def count_candidates(images, median_filter_size=5, correlation_matrix=default_correlation_matrix, c_std=2, **unused):
    """Number of PSF candidates of every field of a stack (one detection pass, no fits): the weights of the
    longest-processing-time partition (pflib.py:1043-1054)."""
    imgs, fmt = _engine.as_pixel_fields(images)
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std, fmt, _pixel_max(imgs, fmt))
    if imgs.ndim != 3:
        raise ValueError("images must have shape (n, H, W)")
    n, H, W = imgs.shape
    if n == 0 or H < 5 or W < 5:
        return np.zeros(n, np.int64)
    eng = _cached(("count", _device_key(), n, H, W), lambda: _engine.Engine(n, H, W, fit_workspace=False))
    with _CACHE_LOCK:
        eng.detect(_engine.to_device_pixels(imgs, fmt), prm)
        return eng.counts[:n].cpu().numpy().astype(np.int64)


def find_peptides(image, median_filter_size=5, correlation_matrix=default_correlation_matrix,
                  candidate_pixels=None, c_std=2, r_2_threshold=0.7, consolidation_radius=4, fit_type='gauss',
                  N_iter=10**3):
    """Find labeled peptides in a TIRF image and characterise their PSFs.  Reference pflib.py:284-520.

    Returns {(round(h_0), round(w_0)): (h_0, w_0, H, A, sigma_h, sigma_w, theta, sub_img, fit_img, rmse, r_2, s_n)}."""
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("image must be two-dimensional")
    return find_peptides_batch(image[None], median_filter_size, correlation_matrix, candidate_pixels, c_std,
                               r_2_threshold, consolidation_radius, fit_type, N_iter)[0]


# ---- output naming / on-disk formats (reference pflib.py:523-746) --------------------------------
_HASH_DIGITS = "0123456789abcdefghijklmnopqrstuvwxyz"


def _py2_round(x):
    """Python-2 round(): half away from zero."""
    x = float(x)
    return math.floor(x + 0.5) if x >= 0 else math.ceil(x - 0.5)


def _py2_str(x):
    """str() of a float as Python 2 printed it ('%.12g', always with a decimal point or exponent)."""
    if isinstance(x, (float, np.floating)):
        s = "%.12g" % float(x)
        if s in ("inf", "-inf", "nan"):
            return s
        if "." not in s and "e" not in s:
            s += ".0"
        return s
    return str(x)


def _epoch_to_hash(epoch):
    """Unix epoch (rounded to the nearest second) -> base-36 string.  Reference pflib.py:523-543."""
    if epoch <= 0:
        raise ValueError("epoch must be positive.")
    n = int(_py2_round(epoch))
    out = ""
    while n > 0:
        n, d = divmod(n, len(_HASH_DIGITS))
        out = _HASH_DIGITS[d] + out
    return out


def _hash_to_epoch(epoch_hash):
    """Inverse of _epoch_to_hash.  Reference pflib.py:546-566."""
    epoch = 0
    for c in epoch_hash:
        d = _HASH_DIGITS.find(c)
        if d < 0:
            raise ValueError("epoch_hash contains unrecognized character(s).")
        epoch = epoch * len(_HASH_DIGITS) + d
    return epoch


def _psfs_filename(image_path, timestamp_epoch, format_suffix):
    """abspath(image_path) + '_psfs_' + hash + format_suffix.  Reference pflib.py:569-591."""
    if timestamp_epoch is None:
        timestamp_epoch = _py2_round(time.time())
    return os.path.abspath(image_path) + '_psfs_' + _epoch_to_hash(timestamp_epoch) + format_suffix


def _output_path(image_path, timestamp_epoch, output_path, suffix):
    if image_path is None and output_path is None:
        raise ValueError("Either image_path or output_path must be provided.")
    if output_path is None:
        if timestamp_epoch is None:
            timestamp_epoch = _py2_round(time.time())
        output_path = _psfs_filename(os.path.abspath(image_path), timestamp_epoch, suffix)
    return output_path


_PICKLE_LOCK = threading.Lock()
_NUMPY2_MODULES = ("numpy._core", "numpy._core.multiarray", "numpy._core.numeric", "numpy._core._multiarray_umath")


def _py2_pickle_bytes(obj):
    """Protocol-0 pickle of `obj` whose numpy globals carry the module paths of numpy 1.x (`numpy.core.multiarray`), which
    numpy 2 still resolves: the files are read back by the reference's Python 2 (flexlibrary.py:541-547), whose numpy has no
    `numpy._core`.  The C pickler does the work; it renames modules for protocols < 3 through the tables of `_compat_pickle`
    (the mechanism that writes `__builtin__` for `builtins`), which get the numpy entries for the duration of the call."""
    import _compat_pickle
    with _PICKLE_LOCK:
        added = [m for m in _NUMPY2_MODULES if m not in _compat_pickle.REVERSE_IMPORT_MAPPING]
        for m in added:
            _compat_pickle.REVERSE_IMPORT_MAPPING[m] = "numpy.core" + m[len("numpy._core"):]
        try:
            return pickle.dumps(obj, protocol=0, fix_imports=True)
        finally:
            for m in added:
                del _compat_pickle.REVERSE_IMPORT_MAPPING[m]


def save_psfs_pkl(psfs, image_path=None, timestamp_epoch=None, output_path=None):
    """Pickle the PSF dict with protocol 0, as the reference's cPickle.dump does (pflib.py:594-636)."""
    output_path = _output_path(image_path, timestamp_epoch, output_path, '.pkl')
    with open(output_path, 'wb') as f:
        f.write(_py2_pickle_bytes(psfs))
    return output_path


CSV_HEADER = ['Absolute image path', 'PSF center (h) coordinate', 'PSF center (w) coordinate', 'PSF base (H)eight',
              'PSF (A)mplitude', 'PSF width (sigma_h)', 'PSF width (sigma_w)', 'PSF (theta)', 'PSF (rmse)',
              'PSF (r_2)', 'PSF (s_n)']


def save_psfs_csv(psfs, image_path=None, timestamp_epoch=None, output_path=None):
    """Tab-delimited table, one row per PSF, floats printed as Python 2's str() did (pflib.py:639-711)."""
    if image_path is not None:
        image_path = os.path.abspath(image_path)
    output_path = _output_path(image_path, timestamp_epoch, output_path, '.csv')
    with open(output_path, 'w', newline='') as f:
        wr = csv.writer(f, dialect='excel-tab')
        wr.writerow(CSV_HEADER)
        for (h, w), v in psfs.items():
            wr.writerow([image_path] + [_py2_str(x) for x in v[:7]] + [_py2_str(v[9]), _py2_str(v[10]), _py2_str(v[11])])
    return output_path


#: zlib level of the PNG files this module writes (converted images, overlays); 1 = fastest.  The reference shells out to
#: ImageMagick / uses PIL's default: the bytes of those files never were comparable, the pixels are.
PNG_COMPRESS_LEVEL = 1


def convert_image(input_path, output_path=None, output_format='png', convert_command='convert'):
    """Convert an image into the desired format; returns the path of the converted image, None on failure
    (pflib.py:55-90).  The reference shells out to ImageMagick's `convert`; the default command is replaced by an
    in-process PIL conversion (16-bit TIFF -> 16-bit PNG keeps every pixel value), any other `convert_command` is
    run as the reference runs it.  An existing file at the output path is overwritten."""
    log = logging.getLogger()
    if output_path is None:
        output_path = '.'.join((input_path, output_format))
    try:
        if convert_command == 'convert':
            from PIL import Image
            with Image.open(input_path) as im:
                # (PNG: the fastest deflate level - 16-bit camera noise does not compress anyway, 302 KB against 282 KB at the
                # default level for a 512 x 512 field, at a sixth of the time; the pixel values are what matters downstream)
                kw = {"compress_level": PNG_COMPRESS_LEVEL} if output_format.lower() == "png" else {}
                im.save(output_path, format=output_format.upper(), **kw)
        else:
            import subprocess
            p = subprocess.Popen([convert_command, input_path, output_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            stdout, stderr = p.communicate()
            if stdout:
                log.debug(stdout)
            if stderr:
                log.debug(stderr)
            p.wait()
    except Exception as e:      # noqa: BLE001 - as the reference: log, return None
        log.exception(e, exc_info=True)
        output_path = None
    return output_path


def read_image(image_path):
    """-> (converted_path, image array).  Reference pflib.py:714-746: an image is a PNG if and only if its suffix is
    '.png'; for any other image an existing `<path>.png` is used, else the image is converted to `<path>.png` first
    (convert_image) - the PNG is what is read, and what the output file names are derived from."""
    from PIL import Image
    converted_path = image_path = os.path.abspath(image_path)
    if image_path[-4:] != '.png':
        if os.path.exists(image_path + '.png'):
            converted_path += '.png'
        else:
            converted_path = convert_image(image_path)
            if converted_path is None:
                raise IOError("cannot convert %s to PNG" % image_path)
    with Image.open(converted_path) as im:
        return converted_path, np.array(im)


def _rescale_to_uint8(image):
    """skimage.exposure.rescale_intensity(image, out_range=np.uint8) as the reference's scikit-image computes it
    (exposure.py of 0.11-0.12: clip to the image's own [min, max], (image - min) / float(max - min) * 255, cast back to the
    INPUT dtype) followed by the reference's .astype(np.uint8): both casts truncate.  A constant image (max == min) divides by
    zero there (nan / a warning); here it gives zeros.  scikit-image is not installed in the build container: parity unpinned."""
    img = np.asarray(image)
    imin, imax = img.min(), img.max()
    if imax == imin:
        return np.zeros(img.shape, np.uint8)
    scaled = (np.clip(img, imin, imax) - imin) / float(imax - imin) * 255.0
    if np.issubdtype(img.dtype, np.integer):
        scaled = scaled.astype(img.dtype)
    return scaled.astype(np.uint8)


def _intensity_scaling(image, **kwargs):
    """contrast_filter of save_psfs_png (reference pflib.py:767-780): the image stretched linearly into 8 bits."""
    return _rescale_to_uint8(image)


def _histogram_equalization(image, **kwargs):
    """contrast_filter of save_psfs_png (reference pflib.py:749-764): skimage.exposure.equalize_hist (for integer images: one
    histogram bin per value between min and max, the cumulative distribution interpolated at every pixel) then stretched into
    8 bits.  scikit-image is not installed in the build container: parity unpinned."""
    img = np.asarray(image)
    if np.issubdtype(img.dtype, np.integer):
        lo = int(img.min())
        hist = np.bincount((img.ravel().astype(np.int64) - lo))
        centers = np.arange(lo, lo + len(hist))
    else:
        hist, edges = np.histogram(img.ravel(), bins=256)
        centers = (edges[:-1] + edges[1:]) / 2.0
    cdf = hist.cumsum() / float(hist.sum())
    return _rescale_to_uint8(np.interp(img.ravel(), centers, cdf).reshape(img.shape))


def save_psfs_png(psfs, image_path, timestamp_epoch=None, output_path=None, square_size=9, square_color='lightblue',
                  square_colors=None, contrast_filter=_intensity_scaling, contrast_filter_args=None):
    """Highlight the found PSFs with squares on a contrast-stretched 8-bit copy of the image and save it as a PNG; returns the
    path.  Reference pflib.py:783-880: same arguments, same file name (_psfs_filename(image_path, epoch, '.png') unless
    output_path is given), same drawing calls (PIL ImageOps.colorize of the 'L' image, one ImageDraw.rectangle outline per
    peak, vertices (w -/+ radius, h -/+ radius) with radius = (square_size - 1) / 2 in Python-2 integer division), ValueError for
    an even or too small square_size.  Host side only (PIL); the bytes of the PNG depend on the installed PIL / zlib and are
    not pinned against the reference."""
    from PIL import Image, ImageDraw, ImageOps
    image_path = os.path.abspath(image_path)
    if output_path is None:
        if timestamp_epoch is None:
            timestamp_epoch = _py2_round(time.time())
        output_path = _psfs_filename(image_path, timestamp_epoch, '.png')
    converted_path, image = read_image(image_path)
    filtered_image = contrast_filter(image, **(contrast_filter_args or {}))
    pillow_image = Image.fromarray(np.ascontiguousarray(filtered_image, dtype=np.uint8)).convert("L")
    highlighted_image = ImageOps.colorize(pillow_image, (0, 0, 0), (255, 255, 255))
    if square_size % 2 == 0 or square_size < 3:
        raise ValueError("square_size must be an odd integer >= 3")
    radius = (square_size - 1) // 2
    draw = ImageDraw.Draw(highlighted_image)
    for (h, w) in psfs.keys():
        square = ((w - radius, h - radius), (w + radius, h + radius))
        color = square_color if (square_colors is None or (h, w) not in square_colors) else square_colors[(h, w)]
        draw.rectangle(square, fill=None, outline=color)
    highlighted_image.save(output_path, compress_level=PNG_COMPRESS_LEVEL)
    return output_path


#: image_batch / _candidate_counts hold at most about this many pixels of one image shape in host memory before they are
#: worked off (fitted, saved, dropped): a directory tree of any size is processed in bounded memory, results are on disk as
#: soon as their window is done
WINDOW_PIXELS = 8 * CHUNK_PIXELS


# ---- host-side workers for the file layer ------------------------------------------------------------------------------
# Per image the GPU needs a fraction of a millisecond; converting / decoding the image and writing its protocol-0 pickle
# (1 MB of text) and CSV take ~100 ms of host time.  The reference runs whole images in worker processes
# (multiprocessing.Pool, pflib.py:1082-1099); here the GPU work stays in this process and the workers do the file work:
# read_image before, save_psfs_* after.  They are spawned (never forked: this process may have initialised the GPU) and
# import nothing that touches the GPU.
IO_WORKERS = None               # None: the cores this process may run on less two, at most 16; 0: everything in this process
_IO_POOL = {"pool": None, "n": 0}
IO_POOL_MIN_IMAGES = 16         # lists shorter than this are not worth starting the workers for


class _IoPool:
    """n worker processes (`python -m fluorosequencingimageanalysis_amd._io_worker`), each served by one thread of a thread
    pool: submit(name, *args) -> Future of pflib._read_job / _save_job run in a worker."""

    def __init__(self, n):
        self.n = int(n)
        self.threads = concurrent.futures.ThreadPoolExecutor(self.n)
        self.local = threading.local()
        self.procs = []
        self.lock = threading.Lock()
        # all workers are started NOW: they import numpy / PIL (0.3 - 0.7 s on a cold machine) while the caller goes on to its
        # own start-up (the GPU context, the library) instead of each one doing so inside its first job
        self.idle = collections.deque(self._spawn() for _ in range(self.n))

    def _spawn(self):
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        # the workers decode / encode images and pickle: no linear algebra - numpy's BLAS must not start a thread per core in
        # every one of them (on a 256-core host that is thousands of threads at start-up)
        for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            env.setdefault(k, "1")
        w = subprocess.Popen([sys.executable, "-m", "fluorosequencingimageanalysis_amd._io_worker"], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, env=env)
        with self.lock:
            self.procs.append(w)
        return w

    def _worker(self):
        w = getattr(self.local, "proc", None)
        if w is None or w.poll() is not None:
            w = None
            with self.lock:
                while self.idle and w is None:
                    w = self.idle.popleft()
                    if w.poll() is not None:
                        w = None
            if w is None:
                w = self._spawn()
            self.local.proc = w
        return w

    def _call(self, name, args):
        import struct
        w = self._worker()
        blob = pickle.dumps((name, args), protocol=pickle.HIGHEST_PROTOCOL)
        w.stdin.write(struct.pack("<q", len(blob)))
        w.stdin.write(blob)
        w.stdin.flush()
        head = w.stdout.read(8)
        if len(head) < 8:
            raise RuntimeError("an I/O worker process died")
        res = pickle.loads(w.stdout.read(struct.unpack("<q", head)[0]))
        if isinstance(res, tuple) and len(res) == 2 and res[0] == "__worker_error__":
            raise RuntimeError(res[1])
        return res

    def submit(self, name, *args):
        return self.threads.submit(self._call, name, args)

    def shutdown(self):
        self.threads.shutdown(wait=True)
        for w in self.procs:
            try:
                w.stdin.close()
                w.wait(timeout=10)
            except Exception:       # noqa: BLE001
                w.kill()


def _io_pool(n_images, num_processes=None):
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 2
    n = IO_WORKERS if IO_WORKERS is not None else min(16, max(1, cores - 2))      # (the cores this process may use, less the GPU's driver threads)
    if num_processes is not None:
        n = int(num_processes)
    if n <= 0 or n_images < IO_POOL_MIN_IMAGES:
        return None
    with _CACHE_LOCK:
        if _IO_POOL["pool"] is None or _IO_POOL["n"] != n:
            if _IO_POOL["pool"] is not None:
                _IO_POOL["pool"].shutdown()
            _IO_POOL["pool"] = _IoPool(n)
            _IO_POOL["n"] = n
        return _IO_POOL["pool"]


def prestart_io_workers(n_images, num_processes=None):
    """Start the host worker processes of image_batch ahead of it (the command line does, before it initialises anything else)."""
    if num_processes is not None and (num_processes < 1 or round(num_processes) != num_processes):
        return None                 # (parallel_image_batch raises for it)
    return _io_pool(n_images, num_processes)


def _warm_gpu():
    """The first-use costs of the GPU path - HIP context, loading the library and its code objects, the first allocations -
    on a tiny frame.  image_batch runs this on a thread while the first window of images is being read; errors are left to the
    real call to report."""
    try:
        img = np.full((1, 48, 48), 100, np.uint16)
        img[0, 22:27, 22:27] += np.array([[1, 4, 7, 4, 1], [4, 20, 33, 20, 4], [7, 33, 55, 33, 7], [4, 20, 33, 20, 4],
                                          [1, 4, 7, 4, 1]], np.uint16) * 40
        find_peptides_records(img)
    except Exception:       # noqa: BLE001
        pass


_WARM = {"thread": None, "lock": threading.Lock()}


def start_gpu_warmup():
    """Run _warm_gpu on a thread (once at a time).  The command line calls this first thing, so that importing torch (a second
    or more of interpreter time), creating the HIP context and loading the library overlap the workers' start-up and the reading
    of the first images; image_batch waits for it before its first GPU pass."""
    with _WARM["lock"]:
        t = _WARM["thread"]
        if t is None or not t.is_alive():
            t = _WARM["thread"] = threading.Thread(target=_warm_gpu, name="fsq-warm-gpu", daemon=True)
            t.start()
    return t


def _join_gpu_warmup():
    with _WARM["lock"]:
        t, _WARM["thread"] = _WARM["thread"], None
    if t is not None:
        t.join()


def shutdown_io_workers():
    with _CACHE_LOCK:
        if _IO_POOL["pool"] is not None:
            _IO_POOL["pool"].shutdown()
            _IO_POOL["pool"] = None


atexit.register(shutdown_io_workers)


def _portable_error(e):
    """An exception that survives the trip between processes (its class may not pickle)."""
    try:
        pickle.loads(pickle.dumps(e))
        return e
    except Exception:           # noqa: BLE001
        return RuntimeError("%s: %s" % (type(e).__name__, e))


def _read_job(ap):
    """(worker process or inline) -> (path, converted path, uint16 array, None) or (path, None, None, exception)."""
    try:
        converted, img = read_image(ap)
        a = _engine.as_u16_fields(img)
        if a.ndim != 2:
            raise ValueError("image must be two-dimensional")
        return ap, converted, a, None
    except Exception as e:      # noqa: BLE001 - reported per image, like the reference (pflib.py:960-964)
        return ap, None, None, _portable_error(e)


def _save_job(psfs, converted, timestamp_epoch):
    """(worker process or inline) -> ((pkl, csv, png) paths, None) or (None, exception)."""
    try:
        return (save_psfs_pkl(psfs, image_path=converted, timestamp_epoch=timestamp_epoch),
                save_psfs_csv(psfs, image_path=converted, timestamp_epoch=timestamp_epoch),
                save_psfs_png(psfs, image_path=converted, timestamp_epoch=timestamp_epoch)), None
    except Exception as e:      # noqa: BLE001
        return None, _portable_error(e)


def _save_records_job(blob, pixel_format, converted, timestamp_epoch):
    """(worker process or inline) The peak records of ONE image (bytes of uint8[k, engine.PEAK_RECORD_BYTES], as the GPU wrote
    them) -> the reference's dict (built HERE, not in the process that drives the GPU) -> pickle and CSV, then the PNG overlay
    last, so that a reader of the pickles never waits for the cosmetics.  -> ((pkl, csv, png) paths, None) or (None, exception)."""
    try:
        rec = np.frombuffer(blob, dtype=np.uint8).reshape(-1, _engine.PEAK_RECORD_BYTES)
        psfs = records_to_dicts(rec, [len(rec)], pixel_format)[0]
        pkl = save_psfs_pkl(psfs, image_path=converted, timestamp_epoch=timestamp_epoch)
        tab = save_psfs_csv(psfs, image_path=converted, timestamp_epoch=timestamp_epoch)
        png = save_psfs_png(psfs, image_path=converted, timestamp_epoch=timestamp_epoch)
        return (pkl, tab, png), None
    except Exception as e:      # noqa: BLE001
        return None, _portable_error(e)


def _read_all(image_paths, pool):
    """read_image for every path, in order; with a pool a bounded number of reads is in flight ahead of the consumer."""
    if pool is None:
        for ap in image_paths:
            yield _read_job(ap)
        return
    ahead = collections.deque()
    it = iter(image_paths)
    limit = 8 * _IO_POOL["n"]
    while True:
        while len(ahead) < limit:
            try:
                ahead.append(pool.submit("read", next(it)))
            except StopIteration:
                break
        if not ahead:
            return
        yield ahead.popleft().result()


def _windows(image_paths, on_unreadable, pool=None):
    """Read the images of a list in order and yield them in windows of same-shaped images, each window at most
    WINDOW_PIXELS pixels: (shape, [(path, converted_path, uint16 array), ...]).  An image that cannot be read, or is not a
    2-D image with 16-bit values, is reported through on_unreadable(path, exception) and skipped."""
    pending = {}
    for ap, converted, a, err in _read_all(image_paths, pool):
        if err is not None:
            on_unreadable(ap, err)
            continue
        group = pending.setdefault(a.shape, [])
        group.append((ap, converted, a))
        if len(group) * a.size >= WINDOW_PIXELS:
            yield a.shape, pending.pop(a.shape)
    for shape in list(pending):
        yield shape, pending.pop(shape)


def _fit_image_groups(images, find_peptides_parameters):
    """find_peptides for a list of 2-D images of any shapes: same-shaped images go through the GPU together
    (find_peptides_batch); -> list of dicts, with the exception an image raised in its place."""
    out = [None] * len(images)
    groups = {}
    for i, img in enumerate(images):
        try:
            a = _engine.as_u16_fields(img)
            if a.ndim != 2:
                raise ValueError("image must be two-dimensional")
            groups.setdefault(a.shape, []).append((i, a))
        except Exception as e:      # noqa: BLE001 - reported per image like the reference
            out[i] = e
    for shape, members in groups.items():
        try:
            res = find_peptides_batch(np.stack([a for _, a in members]), errors='return', **find_peptides_parameters)
        except Exception as e:      # noqa: BLE001 - parameter errors etc. hit every image of the group
            res = [e] * len(members)
        for (i, _), r in zip(members, res):
            out[i] = r
    return out


def _candidate_counts(image_paths, detect_parameters=None):
    """Number of PSF candidates of every image of a list (None where the image cannot be read) - what
    pflib.parallel_image_batch balances its workers by (pflib.py:1043-1050).  Same-shaped images share a GPU pass; the list
    is worked through in bounded windows (_windows)."""
    log = logging.getLogger()
    prm = _engine.detect_params(**{"median_filter_size": 5, "correlation_matrix": default_correlation_matrix, "c_std": 2,
                                   **(detect_parameters or {})})
    index = {}
    for i, p in enumerate(image_paths):
        index.setdefault(p, []).append(i)
    out = [None] * len(image_paths)

    def unreadable(path, e):        # logged and skipped like pflib.py:1044-1048
        log.exception(e, exc_info=True)

    for (H, W), members in _windows(list(dict.fromkeys(image_paths)), unreadable):
        eng = _cached(("count", _device_key(), len(members), H, W), lambda: _engine.Engine(len(members), H, W, fit_workspace=False))
        with _CACHE_LOCK:
            eng.detect(_engine.to_device_u16(np.stack([a for _, _, a in members])), prm)
            counts = eng.counts.cpu().numpy()
        for k, (p, _, _) in enumerate(members):
            for i in index[p]:
                out[i] = int(counts[k])
    return out


def image_batch(image_paths, find_peptides_parameters=None, timestamp_epoch=None, num_processes=None):
    """Find PSFs in every image of a list and save them as pickle and CSV; per-image failures are logged and
    skipped (pflib.py:883-996).  Same-shaped images share GPU passes; the list is worked through in windows of bounded size
    (_windows), every window's files being written before the next one is fitted; reading / converting the images and writing
    the files is done by host worker processes (num_processes of them; None: IO_WORKERS) while this process drives the GPU.
    Returns {absolute image path: (converted image path, pkl path, csv path, png path)} (png: save_psfs_png)."""
    log = logging.getLogger()
    if timestamp_epoch is None:
        timestamp_epoch = _py2_round(time.time())
    paths = list(dict.fromkeys(os.path.abspath(p) for p in image_paths))       # absolute, duplicates dropped (:947-949)
    if find_peptides_parameters is None:
        find_peptides_parameters = {}
    pool = _io_pool(len(paths), num_processes)
    done, saving = {}, collections.deque()
    if pool is not None and not _CACHE:         # (nothing has used the GPU yet: pay its first-use costs while the images are read)
        start_gpu_warmup()

    def unreadable(path, e):        # the reference swallows and logs every per-image failure (:960-964)
        log.error("cannot read %s", path, exc_info=(type(e), e, e.__traceback__))

    def reap(keep):
        while len(saving) > keep:
            ap, converted, fut = saving.popleft()
            files, err = fut.result() if pool is not None else fut
            if err is not None:
                log.error("cannot save the PSFs of %s", ap, exc_info=(type(err), err, err.__traceback__))
            else:
                done.setdefault(ap, (converted,) + tuple(files))

    # The process that drives the GPU never creates a Python object per peak: a window's images go through the continuous-
    # batching pipeline as one stack and come back as ONE byte table (378 bytes per peak, find_peptides_records); every
    # image's slice of it travels to a host worker as bytes, and the worker builds the reference's dict and writes the files
    # (round 3 built the dicts here and pickled every one of them into a worker's pipe: 57 images/s).
    for shape, members in _windows(paths, unreadable, pool):
        err, rec, counts, fmt = None, None, None, N.PIXELS_U16
        _join_gpu_warmup()
        try:
            _check_find_peptides_parameters(find_peptides_parameters)
            rec, counts, fmt = find_peptides_records(np.stack([a for _, _, a in members]), **find_peptides_parameters)
            offs = np.concatenate([[0], np.cumsum(np.maximum(counts, 0))])
        except Exception as e:      # noqa: BLE001 - parameter errors etc. hit every image of the window
            err = e
        for m, (ap, converted, _) in enumerate(members):
            e = err
            if e is None and counts[m] < 0:
                e = AssertionError("field %d: re-keyed peak collides with an existing key (pflib.py:518)" % m)
            if e is not None:
                log.error("find_peptides failed for %s", ap, exc_info=(type(e), e, e.__traceback__))
                continue
            blob = rec[offs[m]:offs[m + 1]].tobytes()
            if pool is not None:
                saving.append((ap, converted, pool.submit("save_records", blob, int(fmt), converted, timestamp_epoch)))
                reap(8 * _IO_POOL["n"])
            else:
                saving.append((ap, converted, _save_records_job(blob, int(fmt), converted, timestamp_epoch)))
                reap(0)
        del rec, members
    reap(0)
    _join_gpu_warmup()
    return {ap: done[ap] for ap in paths if ap in done}


_FIND_PEPTIDES_KEYWORDS = ("median_filter_size", "correlation_matrix", "candidate_pixels", "c_std", "r_2_threshold",
                           "consolidation_radius", "fit_type", "N_iter")


def _check_find_peptides_parameters(fp):
    """find_peptides(**fp) raises TypeError for a keyword it does not have (the reference's image_batch logs that per image,
    pflib.py:957-964); find_peptides_records swallows unknown keywords, so the check is made here."""
    for k in fp:
        if k not in _FIND_PEPTIDES_KEYWORDS:
            raise TypeError("find_peptides() got an unexpected keyword argument '%s'" % k)


def parallel_image_batch(image_paths, find_peptides_parameters=None, timestamp_epoch=None, num_processes=None):
    """Same contract as pflib.parallel_image_batch (pflib.py:1000-1111).

    The reference balances the images over `num_processes` worker processes by candidate count.  Here the workers are
    the ranks of the torch.distributed job this process belongs to (one process per GPU, `torchrun`): images are
    balanced over them with the same longest-processing-time rule and every rank's results are merged on all ranks
    (fluorosequencingimageanalysis_amd.distributed.image_batch_sharded).  Without a process group one GPU handles all
    images.  num_processes is validated like the reference's (:1060-1061); it sets the number of host worker processes that
    read / convert the images and write the result files (image_batch)."""
    if num_processes is not None and (num_processes < 1 or round(num_processes) != num_processes):
        raise ValueError("Number of processes must be an integer >= 1")
    from . import distributed as _dist
    if _dist.world_size() > 1:
        return _dist.image_batch_sharded(image_paths, find_peptides_parameters, timestamp_epoch)
    return image_batch(image_paths, find_peptides_parameters, timestamp_epoch, num_processes)

"""ctypes wrapper of the CPU oracle (TEST INFRASTRUCTURE - never imported by the product)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class FsqOFit(ctypes.Structure):
    _fields_ = [("p", ctypes.c_double * 7), ("fnorm", ctypes.c_double),
                ("status", ctypes.c_int32), ("niter", ctypes.c_int32),
                ("nfev", ctypes.c_int32), ("pad", ctypes.c_int32)]


class FsqORow(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in
                ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta", "rmse", "r2", "s_n")] + \
               [("h", ctypes.c_int32), ("w", ctypes.c_int32)]


FIT_DTYPE = np.dtype([("p", np.float64, 7), ("fnorm", np.float64), ("status", np.int32),
                      ("niter", np.int32), ("nfev", np.int32), ("pad", np.int32)])
ROW_DTYPE = np.dtype([(k, np.float64) for k in
                      ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta", "rmse", "r2", "s_n")] +
                     [("h", np.int32), ("w", np.int32)])
assert FIT_DTYPE.itemsize == ctypes.sizeof(FsqOFit) and ROW_DTYPE.itemsize == ctypes.sizeof(FsqORow)

DEFAULT_K = np.array([[-5935, -5935, -5935, -5935, -5935],
                      [-5935, 8027, 8027, 8027, -5935],
                      [-5935, 8027, 30742, 8027, -5935],
                      [-5935, 8027, 8027, 8027, -5935],
                      [-5935, -5935, -5935, -5935, -5935]], dtype=np.int64)

_libs = {}


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib(libm=False):
    name = "libfsq_oracle_libm.so" if libm else "libfsq_oracle.so"
    if name not in _libs:
        path = os.path.join(HERE, name)
        if os.environ.get("FSQ_ORACLE_LIB") and not libm:      # (tests/test_sanitizers.py: an ASan / UBSan build of the same sources)
            path = os.environ["FSQ_ORACLE_LIB"]
        elif not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.fsq_o_illumina_s_n.restype = ctypes.c_double
        L.fsq_o_enorm.restype = ctypes.c_double
        L.fsq_o_pairwise_sum.restype = ctypes.c_double
        L.fsq_o_pairwise_sum.argtypes = [ctypes.c_void_p, ctypes.c_long]
        L.fsq_o_numpy_sum.restype = ctypes.c_double
        L.fsq_o_numpy_sum.argtypes = [ctypes.c_void_p, ctypes.c_long]
        _libs[name] = L
    return _libs[name]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _pixels(img):
    """(contiguous array, wide): uint16 frames as they are, anything wider as uint32 (values < 2^31)."""
    img = np.asarray(img)
    if img.dtype.itemsize > 2 and img.dtype.kind in "iu":
        if img.size and (int(img.min()) < 0 or int(img.max()) >= 2 ** 31):
            raise ValueError("pixel values outside [0, 2^31)")
        return np.ascontiguousarray(img, dtype=np.uint32), True
    return np.ascontiguousarray(img, dtype=np.uint16), False


def candidates(img, med_size=5, K=DEFAULT_K, c_std=2.0, libm=False, return_cm=False):
    img, wide = _pixels(img)
    K = np.ascontiguousarray(K, dtype=np.int64)
    H, W = img.shape
    cap = H * W
    hw = np.empty((cap, 2), np.int32)
    cm = np.empty((H, W), np.int64)
    thr = ctypes.c_double()
    n = (lib(libm).fsq_o_candidates_u32 if wide else lib(libm).fsq_o_candidates)(_p(img), H, W, int(med_size), _p(K), K.shape[0], ctypes.c_double(c_std),
                                   _p(hw), cap, _p(cm), ctypes.byref(thr))
    if n < 0:
        raise ValueError("oracle candidates error %d" % n)
    if return_cm:
        return hw[:n].copy(), cm, thr.value
    return hw[:n].copy()


def fit_rois(rois, mode=0, n_threads=1, libm=False):
    rois = np.asarray(rois)
    out = np.zeros(rois.size // 25, FIT_DTYPE)
    if rois.dtype.itemsize > 2 and rois.dtype.kind in "iu":
        rois = np.ascontiguousarray(rois, dtype=np.int64).reshape(-1, 25)
        lib(libm).fsq_o_fit_rois_i64(_p(rois), len(rois), int(mode), int(n_threads), _p(out))
        return out
    rois = np.ascontiguousarray(rois, dtype=np.uint16).reshape(-1, 25)
    lib(libm).fsq_o_fit_rois_u16(_p(rois), len(rois), int(mode), int(n_threads), _p(out))
    return out


def model(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    g = np.empty(25)
    lib().fsq_o_model(_p(p), _p(g))
    return g.reshape(5, 5)


def illumina_s_n(roi, libm=False):
    roi = np.ascontiguousarray(roi, dtype=np.int64)
    return lib(libm).fsq_o_illumina_s_n(_p(roi))


def find_peptides(img, med_size=5, K=DEFAULT_K, c_std=2.0, r2_thr=0.7, radius=4, mode=0, n_threads=1,
                  libm=False):
    """Returns (rows[all candidates], fits, keep_idx, key_hw)."""
    img, wide = _pixels(img)
    K = np.ascontiguousarray(K, dtype=np.int64)
    H, W = img.shape
    cap = H * W
    rows = np.zeros(cap, ROW_DTYPE)
    fits = np.zeros(cap, FIT_DTYPE)
    keep = np.zeros(cap, np.int32)
    key = np.zeros((cap, 2), np.int32)
    nc, nk = ctypes.c_int32(), ctypes.c_int32()
    rc = (lib(libm).fsq_o_find_peptides_u32 if wide else lib(libm).fsq_o_find_peptides)(_p(img), H, W, int(med_size), _p(K), K.shape[0],
                                       ctypes.c_double(c_std), ctypes.c_double(r2_thr), int(radius),
                                       int(mode), int(n_threads), _p(rows), _p(fits), _p(keep), _p(key),
                                       cap, ctypes.byref(nc), ctypes.byref(nk))
    if rc == -4:
        raise AssertionError("pflib.py:518 assert")
    if rc < 0:
        raise ValueError("oracle find_peptides error %d" % rc)
    return rows[:nc.value].copy(), fits[:nc.value].copy(), keep[:nk.value].copy(), key[:nk.value].copy()


def enorm(x, inc=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = (len(x) + inc - 1) // inc
    return lib().fsq_o_enorm(_p(x), n, inc)


def phase_correlate(ref, reg, upsample_factor=1):
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    reg = np.ascontiguousarray(reg, dtype=np.float64)
    if ref.shape != reg.shape or ref.ndim != 2:
        raise ValueError("shape")
    out = np.zeros(4)
    rc = lib().fsq_o_phase_correlate(_p(ref), _p(reg), ref.shape[0], ref.shape[1], int(upsample_factor), _p(out))
    if rc < 0:
        raise ValueError("oracle phase_correlate error %d" % rc)
    return tuple(out)


def mexican_hat(img, hw, brim_size=6, radius=9):
    """Spot.mexican_hat_photometry_metric (flexlibrary.py:172-210) for integer spot centres hw[n, 2]."""
    img, wide = _pixels(img)
    H, W = img.shape
    L = lib()
    f = L.fsq_o_mexican_hat_u32 if wide else L.fsq_o_mexican_hat
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6
    return np.array([f(_p(img), H, W, int(h), int(w), int(brim_size), int(radius)) for h, w in hw])


def gaussian_volume(fit7, scaling=10 ** 6):
    """Spot.gaussian_volume_photometry_metric (flexlibrary.py:212-230): float(scaling) * A * sigma_h * sigma_w."""
    f = np.asarray(fit7, dtype=np.float64)
    return (float(scaling) * f[:, 3]) * f[:, 4] * f[:, 5]


def euclid2(dh, dw):
    """scipy.spatial.distance.euclidean of a 2-vector of differences as OpenBLAS dnrm2 evaluates it (x87 extended)."""
    L = lib()
    L.fsq_o_euclid2.restype = ctypes.c_double
    L.fsq_o_euclid2.argtypes = [ctypes.c_double, ctypes.c_double]
    return L.fsq_o_euclid2(float(dh), float(dw))


def greedy_tracking(frame_hw, offsets, shape, candidate_radius=2, spot_radius=0.0):
    """Experiment.greedy_particle_tracking (flexlibrary.py:680-1027) for one field.
    frame_hw: list (per frame) of int arrays [n_f, 2] (Spot.h, Spot.w); offsets: [(d_h, d_w)] relative to the previous
    frame.  Returns (traces int32[n_traces, n_frames] of global spot numbers / -1, n_discarded, link_prev, link_next, kept)."""
    counts = np.array([len(x) for x in frame_hw], dtype=np.int32)
    hw = (np.concatenate([np.asarray(x, dtype=np.int32).reshape(-1, 2) for x in frame_hw])
          if counts.sum() else np.zeros((0, 2), np.int32))
    hw = np.ascontiguousarray(hw, dtype=np.int32)
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.float64).reshape(-1, 2))
    n = int(counts.sum())
    F = len(counts)
    prev = np.full(n + 1, -1, np.int32)
    nxt = np.full(n + 1, -1, np.int32)
    kept = np.zeros(n + 1, np.uint8)
    traces = np.full((n + 1, F), -1, np.int32)
    nt, nd = ctypes.c_int32(), ctypes.c_int32()
    L = lib()
    L.fsq_o_greedy_tracking.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.fsq_o_greedy_tracking(F, _p(counts), _p(hw), _p(off), int(shape[0]), int(shape[1]), int(candidate_radius),
                                 float(spot_radius), _p(prev), _p(nxt), _p(kept), _p(traces), n + 1,
                                 ctypes.byref(nt), ctypes.byref(nd))
    if rc == -1:
        raise ValueError("The first image's offset must be (0, 0) by definiton.")
    if rc == -2:
        raise AssertionError("bin already filled in frame_bins (flexlibrary.py:851)")
    if rc < 0:
        raise RuntimeError("oracle greedy_tracking error %d" % rc)
    return traces[:nt.value].copy(), nd.value, prev[:n].copy(), nxt[:n].copy(), kept[:n].astype(bool)


def centroid_tracking(frames, init_hw, search_radius=3, s_n_cutoff=3.0, offsets=None, size=5):
    """Experiment.luminosity_centroid_particle_tracking (flexlibrary.py:1262-1317) for one field:
    frames uint16[F, H, W], init_hw int[n, 2], offsets int[F, 2] or None -> (hw int32[n, F, 2], present bool[n, F])."""
    frames, wide = _pixels(frames)
    F, H, W = frames.shape
    hw = np.ascontiguousarray(np.asarray(init_hw, dtype=np.int32).reshape(-1, 2))
    n = len(hw)
    off = None if offsets is None else np.ascontiguousarray(np.asarray(offsets, dtype=np.int64).reshape(F, 2))
    out = np.zeros((n, F, 2), np.int32)
    present = np.zeros((n, F), np.uint8)
    L = lib()
    f = L.fsq_o_centroid_tracking_u32 if wide else L.fsq_o_centroid_tracking
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = f(_p(frames), F, H, W, _p(hw), n, int(size), int(search_radius), float(s_n_cutoff),
                                   None if off is None else _p(off), _p(out), _p(present))
    if rc == -1:
        raise ValueError("cannot convert float NaN to integer")
    if rc < 0:
        raise NotImplementedError("oracle centroid_tracking: only Spot size 5")
    return out, present.astype(bool)

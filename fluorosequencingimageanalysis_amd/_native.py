"""ctypes binding of libfsq_hip.so (C ABI declared in include/fsq.h).

The HIP library IS the product: there is no CPU fallback.  Importing this module without the built
library, or calling into it without a GPU, fails loudly."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSQ_HIP_LIB") or os.path.join(HERE, "csrc", "libfsq_hip.so")   # env: A/B builds

FSQ_OK, FSQ_EINVAL, FSQ_ENOMEM, FSQ_ERANGE, FSQ_EHIP, FSQ_EASSERT, FSQ_ENOTIMPL, FSQ_EAGAIN, FSQ_EINTERNAL = 0, -1, -2, -3, -4, -5, -6, -7, -8
MAX_TICKETS = 32
MODE_REF, MODE_TEXTBOOK, MODE_TEXTBOOK_F32, ENGINE_LANE, ENGINE_QUAD = 0, 1, 2, 0x100, 0x200
PIXELS_U16, PIXELS_F16, PIXELS_F16_FLAG = 0, 1, 0x1000
PIXELS_U32, PIXELS_U32_FLAG = 2, 0x2000        # uint32 pixels (values < 2^31): fsq_detect / fsq_fit_candidates only
DTYPE_F64, DTYPE_U16 = 0, 1

ROW_DTYPE = np.dtype([(k, np.float64) for k in
                      ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta", "rmse", "r2", "s_n", "p2", "p3")] +
                     [(k, np.int32) for k in ("h", "w", "field", "status", "niter", "nfev", "key_h", "key_w")])
assert ROW_DTYPE.itemsize == 128


MAX_KSIZE = 15              # FSQ_MAX_KSIZE of include/fsq.h


class FsqDetectParams(ctypes.Structure):
    _fields_ = [("median_filter_size", ctypes.c_int32), ("ksz", ctypes.c_int32), ("c_std", ctypes.c_double),
                ("K", ctypes.c_int64 * (MAX_KSIZE * MAX_KSIZE)), ("pixel_format", ctypes.c_int32), ("pixel_bits", ctypes.c_int32)]


class NativeLibraryMissing(RuntimeError):
    pass


_lib = None

_SIGS = {
    "fsq_version": (ctypes.c_char_p, []),
    "fsq_last_hip_error": (ctypes.c_char_p, []),
    "fsq_device_count": (ctypes.c_int, []),
    "fsq_detect_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "fsq_detect": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(FsqDetectParams), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "fsq_fit_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64]),
    "fsq_fit_candidates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_void_p]),
    "fsq_fitq_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int64]),
    "fsq_fitq_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "fsq_fitq_submit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "fsq_fitq_advance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_int)]),
    "fsq_fitq_take": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "fsq_fitq_alive": (ctypes.c_int64, [ctypes.c_void_p]),
    "fsq_fitq_rounds": (ctypes.c_int64, [ctypes.c_void_p]),
    "fsq_fitq_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "fsq_fit_rois": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int64, ctypes.c_void_p]),
    "fsq_consolidate_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "fsq_consolidate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "fsq_kept_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "fsq_fit_images": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p]),
    "fsq_find_peptides_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]),
    "fsq_find_peptides": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(FsqDetectParams),
                                         ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                         ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                         ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "fsq_phase_correlate_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                             ctypes.c_void_p]),
    "fsq_phase_correlate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "fsq_mexican_hat": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "fsq_mexican_hat_u32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "fsq_track_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "fsq_greedy_tracking": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_double] + [ctypes.c_void_p] * 7 +
                            [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "fsq_centroid_tracking": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "fsq_centroid_tracking_u32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "fsq_selftest_dnrm2": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "fsq_selftest_division": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    "fsq_selftest_exp": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    "fsq_selftest_rotation": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    "fsq_selftest_square": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
                                           ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p]),
    "fsq_fit_last_slow_count": (ctypes.c_int64, []),
    "fsq_has_ab_engines": (ctypes.c_int, []),
}
EXPORTED = tuple(_SIGS)


def lib():
    """Load libfsq_hip.so (once). Raises NativeLibraryMissing if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C fluorosequencingimageanalysis_amd/csrc). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                f = getattr(L, name)      # AttributeError if the library lacks a declared symbol
            except AttributeError:
                if os.environ.get("FSQ_HIP_LIB") and (name.startswith("fsq_selftest") or name == "fsq_has_ab_engines"):
                    continue              # (an older A/B build without a newer self-test hook)
                raise
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def source_sha16():
    """Hash of the library's sources (csrc/*.hip, csrc/*.h, include/fsq.h): what profiles/fit_counters_latest.json records so
    that counters taken on one version of the kernels are not reported for another (bench.py)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")) + glob.glob(os.path.join(HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "fsq.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def check(rc, what):
    if rc == FSQ_OK:
        return
    if rc == FSQ_EINVAL:
        raise ValueError("%s: invalid argument" % what)
    if rc == FSQ_EASSERT:
        raise AssertionError("%s: re-keyed peak collides with an existing key (pflib.py:518)" % what)
    if rc == FSQ_ENOTIMPL:
        raise NotImplementedError(what)
    if rc == FSQ_ENOMEM:
        raise MemoryError(what)
    if rc == FSQ_EAGAIN:
        raise BlockingIOError("%s: the fit queue has no room right now" % what)
    if rc == FSQ_EHIP:
        raise RuntimeError("%s: HIP error: %s" % (what, lib().fsq_last_hip_error().decode()))
    if rc == FSQ_EINTERNAL:
        raise RuntimeError("%s: the fit queue is empty but a batch is incomplete (engine bug)" % what)
    raise RuntimeError("%s: error %d" % (what, rc))

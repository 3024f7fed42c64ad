"""Host-side driver of the HIP hot path: detect -> fit -> consolidate for a batch of fields that is
resident in HBM.  PyTorch is used only for device memory and streams; all arithmetic happens in
libfsq_hip.so through the C ABI (include/fsq.h).  There is no CPU fallback."""
import ctypes

import numpy as np

from . import _native as N

DEFAULT_CORRELATION_MATRIX = np.array([[-5935, -5935, -5935, -5935, -5935],
                                       [-5935, 8027, 8027, 8027, -5935],
                                       [-5935, 8027, 30742, 8027, -5935],
                                       [-5935, 8027, 8027, 8027, -5935],
                                       [-5935, -5935, -5935, -5935, -5935]])   # values of pflib.py:48-52


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("fluorosequencingimageanalysis_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch


def to_device_u16(images, device=None):
    """uint16 numpy array (any shape; or the 16-bit words of a float16 image, as_pixel_fields) -> int16-typed torch
    tensor on the GPU holding the same bytes."""
    torch = _torch()
    a = np.ascontiguousarray(images, dtype=np.uint16)
    return torch.from_numpy(a.view(np.int16)).to(device or "cuda", non_blocking=False)


def to_device_pixels(words, pixel_format, device=None):
    """The host words of as_pixel_fields -> device tensor of the same bytes (int16-typed for the 16-bit formats, int32-typed
    for PIXELS_U32)."""
    if pixel_format != N.PIXELS_U32:
        return to_device_u16(words, device)
    a = np.ascontiguousarray(words, dtype=np.uint32)
    return _torch().from_numpy(a.view(np.int32)).to(device or "cuda", non_blocking=False)


def as_u16_fields(image):
    """One image / a stack as uint16, the way the reference's `image.astype(np.int64)` (pflib.py:241, 443) reads its input:
    integer pixels as they are, floating-point pixels truncated toward zero.  The GPU path works on 16-bit pixels: values
    outside [0, 65535] (or not finite) raise NotImplementedError."""
    a = np.asarray(image)
    if a.dtype == np.uint16:
        return np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        if a.size and not np.isfinite(a).all():
            raise NotImplementedError("non-finite pixel values are not supported by the GPU path")
        a = a.astype(np.int64)                      # truncation toward zero, as the reference
    elif a.dtype.kind not in "iub":
        raise NotImplementedError("only real-valued pixel data is supported (got %s)" % a.dtype)
    if a.size and (a.min() < 0 or a.max() > 65535):
        raise NotImplementedError("pixel values outside [0, 65535] are not supported by the GPU path")
    return np.ascontiguousarray(a.astype(np.uint16))


def as_pixel_fields(image, wide=False):
    """-> (pixel words to upload, pixel format).  Integer images (and floats, truncated toward zero) whose values fit 16 bits
    become uint16 words; with a value beyond 65 535 - or with wide=True, or handed over as uint32 - uint32 words (PIXELS_U32, values
    below 2^31); a
    float16 image (BASELINE.json configs[4]: pixel values pre-scaled into binary16) is uploaded as it is and truncated toward zero
    by the kernels' loads, which is what the reference's image.astype(np.int64) (pflib.py:241, 443) does with it."""
    a = np.asarray(image)
    if a.dtype == np.float16:
        if a.size and not (np.isfinite(a).all() and (a >= 0).all()):
            raise NotImplementedError("float16 pixels must be finite and non-negative")
        return np.ascontiguousarray(a).view(np.uint16), N.PIXELS_F16
    if a.dtype.kind == "f":
        if a.size and not np.isfinite(a).all():
            raise NotImplementedError("non-finite pixel values are not supported by the GPU path")
        a = a.astype(np.int64)                      # truncation toward zero, as the reference
    if a.dtype == np.uint32:
        # uint32 input is taken as it is - no pass over the data to see whether it would fit 16 bits, no copy (a stack of 1 024
        # fields is 1 GB: each pass costs 0.1 s of a 0.4 s call); values of 2^31 and more are refused where the data is next touched
        # (pflib: the stager's per-chunk maximum / _pixel_max)
        return np.ascontiguousarray(a), N.PIXELS_U32
    if wide and a.dtype.kind in "iub" and a.dtype != np.float16:      # (the caller wants uint32 words whatever the values are)
        if a.size and (int(a.min()) < 0 or int(a.max()) >= 2 ** 31):
            raise NotImplementedError("pixel values outside [0, 2^31) are not supported by the GPU path")
        return np.ascontiguousarray(a.astype(np.uint32)), N.PIXELS_U32
    if a.dtype.kind in "iu" and a.dtype.itemsize > 2 and a.size and int(a.max()) > 65535:
        # beyond 16 bits (round 4): uint32 words, PIXELS_U32 - detection, fits, consolidation and records as for 16-bit pixels
        if int(a.min()) < 0 or int(a.max()) >= 2 ** 31:
            raise NotImplementedError("pixel values outside [0, 2^31) are not supported by the GPU path")
        return np.ascontiguousarray(a.astype(np.uint32)), N.PIXELS_U32
    return as_u16_fields(a), N.PIXELS_U16


def as_integer_fields(image):
    """-> (pixel words, PIXELS_U16 or PIXELS_U32) for the entry points that read VALUES only (photometry, centroid tracking): like
    as_pixel_fields, but a float16 image is truncated to integers like any other float image instead of being uploaded as binary16 words."""
    a = np.asarray(image)
    words, fmt = as_pixel_fields(a.astype(np.float32) if a.dtype == np.float16 else a)
    if fmt == N.PIXELS_U32 and words.size and int(words.max()) >= 2 ** 31:         # (uint32 input is not scanned by as_pixel_fields)
        raise NotImplementedError("pixel values outside [0, 2^31) are not supported by the GPU path")
    return words, fmt


def quantise_f16(images):
    """Integer pixels -> (float16 image, scale): scaled down just enough to fit binary16's range (max 65504), then
    rounded to binary16 ("image pre-scaled to fit fp16")."""
    a = np.asarray(images)
    mx = float(a.max()) if a.size else 0.0
    scale = 1.0 if mx <= 65504.0 else 65504.0 / mx
    return (a.astype(np.float64) * scale).astype(np.float16), scale


def pixel_values(words, pixel_format):
    """The integer pixel values the kernels see for 16-bit words of the given format (host side, for sub_img)."""
    w = np.asarray(words)
    if pixel_format == N.PIXELS_F16:
        return w.view(np.float16).astype(np.int64)
    if pixel_format == N.PIXELS_U32:
        return w.view(np.uint32).astype(np.int64)
    return w.view(np.uint16).astype(np.int64)


def detect_params(median_filter_size, correlation_matrix, c_std, pixel_format=N.PIXELS_U16, pixel_max=None):
    """pixel_max (PIXELS_U32): the largest pixel value of the images - its bit length bounds the integer response's
    exactness domain (FsqDetectParams.pixel_bits); None = 31 bits."""
    K = np.asarray(correlation_matrix)
    if K.ndim != 2 or K.shape[0] != K.shape[1] or K.shape[0] % 2 == 0:
        raise ValueError("correlation_matrix must be square, with an odd number of rows and columns")
    if K.shape[0] > N.MAX_KSIZE or not (1 <= int(median_filter_size) <= N.MAX_KSIZE):
        raise NotImplementedError("correlation_matrix / median_filter_size larger than %d are not supported" % N.MAX_KSIZE)
    p = N.FsqDetectParams()
    p.median_filter_size = int(median_filter_size)
    p.ksz = int(K.shape[0])
    p.c_std = float(c_std)
    p.pixel_format = int(pixel_format)
    p.pixel_bits = max(1, int(pixel_max).bit_length()) if (pixel_format == N.PIXELS_U32 and pixel_max is not None) else 0
    flat = K.astype(np.int64).ravel()
    for i, v in enumerate(flat):
        p.K[i] = int(v)
    return p


PEAK_RECORD_BYTES = 128 + 200 + 50


#: one peak record as a NumPy structure (unaligned: 378 bytes): the FsqRow fields + fit_img + the sub_img pixel words
RECORD_DTYPE = np.dtype({"names": list(N.ROW_DTYPE.names) + ["fit", "sub"],
                         "formats": [N.ROW_DTYPE.fields[k][0] for k in N.ROW_DTYPE.names] + [("<f8", (5, 5)), ("<u2", (5, 5))],
                         "offsets": [N.ROW_DTYPE.fields[k][1] for k in N.ROW_DTYPE.names] + [128, 328],
                         "itemsize": PEAK_RECORD_BYTES})


#: the record of PIXELS_U32 frames (fsq_find_peptides with uint32 pixels): sub_img as 25 uint32 words, 428 bytes
PEAK_RECORD_BYTES_U32 = 128 + 200 + 100
RECORD_DTYPE_U32 = np.dtype({"names": list(N.ROW_DTYPE.names) + ["fit", "sub"],
                             "formats": [N.ROW_DTYPE.fields[k][0] for k in N.ROW_DTYPE.names] + [("<f8", (5, 5)), ("<u4", (5, 5))],
                             "offsets": [N.ROW_DTYPE.fields[k][1] for k in N.ROW_DTYPE.names] + [128, 328],
                             "itemsize": PEAK_RECORD_BYTES_U32})


def peak_record_bytes(pixel_format=N.PIXELS_U16):
    return PEAK_RECORD_BYTES_U32 if pixel_format == N.PIXELS_U32 else PEAK_RECORD_BYTES


def peak_record_view(rec, pixel_format=N.PIXELS_U16):
    """uint8[k, peak_record_bytes(pixel_format)] (host, C-contiguous) -> RECORD_DTYPE[k] (RECORD_DTYPE_U32 for PIXELS_U32)
    view of the same memory (no copy)."""
    if pixel_format == N.PIXELS_U32:
        return np.ascontiguousarray(rec).reshape(-1, PEAK_RECORD_BYTES_U32).view(RECORD_DTYPE_U32).reshape(-1)
    return np.ascontiguousarray(rec).reshape(-1, PEAK_RECORD_BYTES).view(RECORD_DTYPE).reshape(-1)


def split_peak_records(rec, pixel_format=N.PIXELS_U16):
    """uint8[k, PEAK_RECORD_BYTES] (host) -> (rows FsqRow[k], fit_img float64[k, 5, 5], sub_img int64[k, 5, 5])."""
    rec = np.ascontiguousarray(rec).reshape(-1, PEAK_RECORD_BYTES)
    rows = np.ascontiguousarray(rec[:, :128]).view(N.ROW_DTYPE).reshape(-1)
    fit = np.ascontiguousarray(rec[:, 128:328]).view(np.float64).reshape(-1, 5, 5)
    sub = pixel_values(np.ascontiguousarray(rec[:, 328:378]).view(np.uint16), pixel_format).reshape(-1, 5, 5)
    return rows, fit, sub


class PathRunner:
    """fsq_find_peptides - the whole path (detect -> fit -> consolidate -> peak records) of a batch of same-shaped fields as ONE
    library call on the current stream, with its own workspace: no interpreter between the stages (the calling thread holds
    the interpreter lock only to make the call).  Buffers grow on demand (FSQ_ERANGE tells by how much)."""

    def __init__(self, n_fields, H, W, device=None, cand_cap=None, record_cap=None, record_bytes=PEAK_RECORD_BYTES):
        """record_bytes: PEAK_RECORD_BYTES, or PEAK_RECORD_BYTES_U32 for a runner that is handed PIXELS_U32 frames."""
        torch = _torch()
        self.torch = torch
        self.record_bytes = int(record_bytes)
        self.dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.L = N.lib()
        self.n_fields, self.H, self.W = int(n_fields), int(H), int(W)
        self.offsets = torch.zeros(self.n_fields + 1, dtype=torch.int32, device=self.dev)
        self.nkeep = torch.zeros(self.n_fields + 1, dtype=torch.int32, device=self.dev)
        self._size(int(cand_cap or self.n_fields * max(1024, self.H * self.W // 48)), int(record_cap or self.n_fields * max(256, self.H * self.W // 400)))

    def _size(self, cand_cap, record_cap):
        torch = self.torch
        self.cand_cap, self.record_cap = int(cand_cap), int(record_cap)
        nbytes = self.L.fsq_find_peptides_workspace_bytes(self.n_fields, self.H, self.W, self.cand_cap, self.record_cap)
        if nbytes < 0:
            raise ValueError("invalid batch shape")
        self.ws = None
        self.records = None
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.records = torch.empty((self.record_cap, self.record_bytes), dtype=torch.uint8, device=self.dev)

    def run(self, d_img, prm, r2_threshold=0.7, radius=4, mode=N.MODE_REF, py2_round=True):
        """-> (records uint8[k, PEAK_RECORD_BYTES] - a view of this runner's buffer, valid until its next run -, offsets int32[n + 1],
        nkeep int32[n + 1], candidates) on the device; enqueued on the current stream."""
        ncand, nrec = ctypes.c_int64(0), ctypes.c_int64(0)
        n_fields = int(d_img.shape[0])              # (any number of fields up to the capacity the runner was built for)
        if not (1 <= n_fields <= self.n_fields) or tuple(d_img.shape[1:]) != (self.H, self.W):
            raise ValueError("d_img must hold 1 .. %d fields of %d x %d" % (self.n_fields, self.H, self.W))
        if peak_record_bytes(prm.pixel_format) != self.record_bytes or d_img.element_size() != (4 if prm.pixel_format == N.PIXELS_U32 else 2):
            raise ValueError("pixel format %d does not match this runner's record size / the image's word size" % prm.pixel_format)
        while True:
            rc = self.L.fsq_find_peptides(d_img.data_ptr(), n_fields, self.H, self.W, ctypes.byref(prm), float(r2_threshold),
                                          int(radius), 1 if py2_round else 0, int(mode), self.cand_cap, self.records.data_ptr(),
                                          self.record_cap, self.offsets.data_ptr(), self.nkeep.data_ptr(), ctypes.byref(ncand),
                                          ctypes.byref(nrec), self.ws.data_ptr(), self.ws.numel(),
                                          self.torch.cuda.current_stream(self.dev).cuda_stream)
            if rc != N.FSQ_ERANGE:
                break
            self._size(max(self.cand_cap, ncand.value + ncand.value // 8 + 1024), max(self.record_cap, nrec.value + nrec.value // 8 + 256))
        N.check(rc, "fsq_find_peptides")
        return self.records[:nrec.value], self.offsets[:n_fields + 1], self.nkeep[:n_fields + 1], ncand.value


class DeviceBatch:
    """Results of one detect+fit+consolidate pass, still on the GPU."""
    __slots__ = ("n_fields", "H", "W", "cand", "counts", "offsets", "rows", "keep", "nkeep", "total", "thr")


class Engine:
    """Re-usable buffers for a fixed (n_fields, H, W) batch shape on one GPU."""

    def __init__(self, n_fields, H, W, device=None, cand_per_field=None, fit_workspace=True, shared_ws=None):
        """fit_workspace=False: the LM fit runs in a FitQueue, which owns the solver state (StreamPipeline).
        shared_ws: a uint8 device tensor used as the detection / consolidation workspace instead of a private
        one - for engines whose detect and consolidate calls are all enqueued on ONE stream."""
        torch = _torch()
        self.torch = torch
        self.dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.L = N.lib()
        self.n_fields, self.H, self.W = int(n_fields), int(H), int(W)
        ws = self.L.fsq_detect_workspace_bytes(self.n_fields, self.H, self.W)
        ws2 = self.L.fsq_consolidate_workspace_bytes(self.n_fields, self.H, self.W)
        if ws < 0 or ws2 < 0:
            raise ValueError("invalid batch shape")
        if shared_ws is not None:
            if shared_ws.numel() < max(ws, ws2):
                raise ValueError("shared_ws is smaller than workspace_bytes(%d, %d, %d)" % (self.n_fields, self.H, self.W))
            self.ws = shared_ws
        else:
            self.ws = torch.empty(max(ws, ws2), dtype=torch.uint8, device=self.dev)
        self._fit_workspace = bool(fit_workspace)
        self.counts = torch.zeros(self.n_fields + 1, dtype=torch.int32, device=self.dev)
        self.offsets = torch.zeros(self.n_fields + 1, dtype=torch.int32, device=self.dev)
        self.nkeep = torch.zeros(self.n_fields + 1, dtype=torch.int32, device=self.dev)
        self.thr = torch.zeros(self.n_fields, dtype=torch.float64, device=self.dev)
        per = cand_per_field or max(1024, (self.H * self.W) // 32)
        self._alloc_cand(self.n_fields * per)

    def _alloc_cand(self, cap):
        torch = self.torch
        self.cap = int(cap)
        self.cand = torch.empty((self.cap, 3), dtype=torch.int32, device=self.dev)
        self.rows = torch.empty((self.cap, 128), dtype=torch.uint8, device=self.dev)
        self.keep = torch.empty(self.cap, dtype=torch.int32, device=self.dev)
        self.fit_ws = (torch.empty(self.L.fsq_fit_workspace_bytes(self.cap), dtype=torch.uint8, device=self.dev)
                       if self._fit_workspace else None)

    def _stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def detect(self, d_img, prm):
        """Run candidate detection; returns the total number of candidates (host int, synchronises)."""
        rc = self.L.fsq_detect(d_img.data_ptr(), self.n_fields, self.H, self.W, ctypes.byref(prm), self.cand.data_ptr(),
                               self.cap, self.counts.data_ptr(), self.offsets.data_ptr(), self.thr.data_ptr(),
                               self.ws.data_ptr(), self.ws.numel(), self._stream())
        N.check(rc, "fsq_detect")
        total = int(self.counts[self.n_fields].item())
        if total < 0:
            raise NotImplementedError("fsq_detect: the response of a field sums to 2^53 or more, beyond which "
                                      "numpy.mean (pflib.py:250) is no longer the exact integer mean")
        if total > self.cap:                       # candidate buffer too small: grow and redo the pass
            self._alloc_cand(int(total * 1.25) + 1024)
            return self.detect(d_img, prm)
        return total

    @staticmethod
    def workspace_bytes(n_fields, H, W):
        L = N.lib()
        return max(L.fsq_detect_workspace_bytes(n_fields, H, W), L.fsq_consolidate_workspace_bytes(n_fields, H, W))

    def fit(self, d_img, total, mode=N.MODE_REF, pixel_format=N.PIXELS_U16):
        if self.fit_ws is None:
            raise RuntimeError("this Engine was created without a fit workspace (its fits run in a FitQueue)")
        if pixel_format == N.PIXELS_F16:
            mode |= N.PIXELS_F16_FLAG
        elif pixel_format == N.PIXELS_U32:
            mode |= N.PIXELS_U32_FLAG
        rc = self.L.fsq_fit_candidates(d_img.data_ptr(), self.n_fields, self.H, self.W, self.cand.data_ptr(), total,
                                       mode, self.rows.data_ptr(), self.fit_ws.data_ptr(), self.fit_ws.numel(),
                                       self._stream())
        N.check(rc, "fsq_fit_candidates")

    def consolidate(self, r2_threshold, radius, py2_round=True):
        rc = self.L.fsq_consolidate(self.rows.data_ptr(), self.counts.data_ptr(), self.offsets.data_ptr(), self.n_fields,
                                    self.H, self.W, float(r2_threshold), int(radius), 1 if py2_round else 0,
                                    self.keep.data_ptr(), self.nkeep.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                    self._stream())
        N.check(rc, "fsq_consolidate")

    def run(self, d_img, prm, r2_threshold=0.7, radius=4, mode=N.MODE_REF, py2_round=True):
        """detect -> fit -> consolidate on images already in HBM. Returns total candidates."""
        if radius < 2:
            raise ValueError("consolidation_radius must be at least 2")
        total = self.detect(d_img, prm)
        self.fit(d_img, total, mode, prm.pixel_format)
        self.consolidate(r2_threshold, radius, py2_round)
        return total

    # ---- the peak table ---------------------------------------------------------------------
    def kept_table(self):
        """The kept peaks of all fields as one contiguous device table in field order (fsq_kept_rows):
        -> (uint8[k, 128] FsqRow bytes, int32[n_fields + 1] per-field offsets into it).  Reads the kept count
        (synchronises the current stream)."""
        torch = self.torch
        k = int(self.nkeep[self.n_fields].item())
        table = torch.empty((max(k, 0), 128), dtype=torch.uint8, device=self.dev)
        offs = torch.empty(self.n_fields + 1, dtype=torch.int32, device=self.dev)
        rc = self.L.fsq_kept_rows(self.rows.data_ptr(), self.keep.data_ptr(), self.offsets.data_ptr(), self.nkeep.data_ptr(),
                                  self.n_fields, table.data_ptr(), table.shape[0], offs.data_ptr(), self._stream())
        N.check(rc, "fsq_kept_rows")
        return table, offs

    def fit_images(self, table):
        """fit_img of every row of a kept table (gaussfitter.py:253) -> float64[k, 25] on the device."""
        fit = self.torch.empty((table.shape[0], 25), dtype=self.torch.float64, device=self.dev)
        if table.shape[0]:
            N.check(self.L.fsq_fit_images(table.data_ptr(), None, table.shape[0], fit.data_ptr(), self._stream()),
                    "fsq_fit_images")
        return fit

    def peak_records(self, d_img):
        """Everything pflib's 12-tuple holds, per kept peak, as one device byte table uint8[k, PEAK_RECORD_BYTES]:
        the FsqRow (128 B), fit_img float64[25] (200 B), sub_img uint16[25] (50 B) - the unit of the multi-GPU
        gather (d_img int32-typed, PIXELS_U32: uint32[25], 100 B, records of PEAK_RECORD_BYTES_U32).
        -> (records, per-field offsets int32[n_fields + 1])."""
        torch = self.torch
        table, offs = self.kept_table()
        k = table.shape[0]
        fit = self.fit_images(table)
        ints = table[:, 96:128].contiguous().view(torch.int32)            # h, w, field, status, niter, nfev, key_h, key_w
        d = torch.arange(-2, 3, device=self.dev)
        hh = (ints[:, 0].long()[:, None, None] + d[None, :, None]).expand(k, 5, 5)
        ww = (ints[:, 1].long()[:, None, None] + d[None, None, :]).expand(k, 5, 5)
        ff = ints[:, 2].long()[:, None, None].expand(k, 5, 5)
        sub = d_img[ff, hh, ww].contiguous()                              # int16- / int32-typed pixel words (pflib.py:443)
        rec = torch.cat([table, fit.view(torch.uint8).reshape(k, 200), sub.view(torch.uint8).reshape(k, 25 * sub.element_size())], dim=1)
        return rec, offs

    # ---- host-side extraction --------------------------------------------------------------
    def kept_tables(self, total=None):
        """Copy the consolidated tables to the host: list (per field) of (rows, fit_imgs) or None when the
        reference's re-key assertion fired for that field."""
        nkeep = self.nkeep.cpu().numpy()
        table, offs = self.kept_table()
        fit = self.fit_images(table).cpu().numpy().reshape(-1, 5, 5)
        rows = table.cpu().numpy().view(N.ROW_DTYPE).reshape(-1)
        offs = offs.cpu().numpy()
        out = []
        for f in range(self.n_fields):
            if int(nkeep[f]) < 0:
                out.append(None)
                continue
            out.append((rows[offs[f]:offs[f + 1]], fit[offs[f]:offs[f + 1]]))
        return out

    def all_rows(self, total):
        return self.rows[:total].cpu().numpy().view(N.ROW_DTYPE).reshape(-1)

    def candidates(self, total):
        return self.cand[:total].cpu().numpy(), self.counts.cpu().numpy(), self.offsets.cpu().numpy()


class LanePipeline:
    """Several Engines ("lanes"), each on its own HIP stream and host thread, working through successive batches.

    The LM fit of one batch ends in a long tail of rounds with few fits left (a fit may need 200 sequential
    iterations; those rounds are launch/latency bound and leave most CUs idle).  Fits of different batches are
    independent, so a second lane whose batch is offset in time fills those idle CUs with the busy early rounds of
    its own batch - the multi-batch analogue of the reference's multiprocessing.Pool over image partitions
    (pflib.py:1046-1111).  Results are unaffected: every lane runs the same kernels on its own buffers."""

    def __init__(self, engines):
        self.engines = list(engines)
        torch = self.engines[0].torch
        self.streams = [torch.cuda.Stream(device=e.dev) for e in self.engines]

    def run(self, work, n_steps, stagger_s=0.0):
        """Call work(lane, step, engine) for step in range(n_steps) in every lane (lane k starts k*stagger_s late).
        Returns when all lanes are done and their streams are idle; re-raises the first lane error."""
        import threading
        import time
        torch = self.engines[0].torch
        errs = []

        def body(k):
            try:
                torch.cuda.set_device(self.engines[k].dev)
                with torch.cuda.stream(self.streams[k]):
                    if k and stagger_s > 0:
                        time.sleep(k * stagger_s)
                    for i in range(n_steps):
                        work(k, i, self.engines[k])
                    self.streams[k].synchronize()
            except BaseException as e:      # noqa: BLE001 - handed to the caller below
                errs.append(e)

        ths = [threading.Thread(target=body, args=(k,), daemon=True) for k in range(len(self.engines))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]


class FitQueue:
    """The LM-fit engine kept alive across batches (fsq_fitq_* of include/fsq.h): batches are submitted while
    earlier ones are still finishing and every round advances all fits in flight, so the long latency-bound tail
    of one batch rides along in the full launches of the next ones."""

    def __init__(self, pool_slots, queue_cap, mode=N.MODE_REF, device=None, stream=None):
        torch = _torch()
        self.torch = torch
        self.dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.L = N.lib()
        self.stream = stream or torch.cuda.Stream(device=self.dev)
        nbytes = self.L.fsq_fitq_workspace_bytes(int(pool_slots), int(queue_cap))
        if nbytes < 0:
            raise ValueError("invalid fit queue size")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.pool_slots, self.queue_cap = int(pool_slots), int(queue_cap)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.dev):
            N.check(self.L.fsq_fitq_create(ctypes.byref(h), self.ws.data_ptr(), nbytes, self.pool_slots, self.queue_cap,
                                           mode, self.stream.cuda_stream), "fsq_fitq_create")
        self.h = h

    def submit(self, d_img, n_fields, H, W, d_cand, n, d_rows, pixel_format=N.PIXELS_U16):
        """-> ticket, or None when the queue has no room right now (advance and try again)."""
        t = ctypes.c_int(-1)
        rc = self.L.fsq_fitq_submit(self.h, d_img.data_ptr(), int(pixel_format), n_fields, H, W, d_cand.data_ptr(), int(n),
                                    d_rows.data_ptr(), ctypes.byref(t))
        if rc == N.FSQ_EAGAIN:
            return None
        N.check(rc, "fsq_fitq_submit")
        return t.value

    def advance(self, max_rounds=0, alive_below=0):
        """-> (fits alive, batches finished during the call)."""
        a, f = ctypes.c_int64(0), ctypes.c_int(0)
        N.check(self.L.fsq_fitq_advance(self.h, int(max_rounds), int(alive_below), ctypes.byref(a), ctypes.byref(f)),
                "fsq_fitq_advance")
        return a.value, f.value

    def take(self, ticket, consumer_stream):
        """True once the batch's rows are written (consumer_stream then waits for them; the ticket is released)."""
        rc = self.L.fsq_fitq_take(self.h, int(ticket), consumer_stream.cuda_stream)
        if rc < 0:
            N.check(rc, "fsq_fitq_take")
        return rc == 1

    @property
    def alive(self):
        return int(self.L.fsq_fitq_alive(self.h))

    @property
    def rounds(self):
        return int(self.L.fsq_fitq_rounds(self.h))

    def close(self):
        if self.h is not None:
            h, self.h = self.h, None
            N.check(self.L.fsq_fitq_destroy(h), "fsq_fitq_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass


_ALLOC_LOCK = __import__("threading").Lock()


class StreamPipeline:
    """detect -> LM fit -> consolidate over a STREAM of same-shaped batches with continuous batching of the fits.

    One host thread, two HIP streams: the fit queue's rounds on one, every batch's detection, consolidation and
    hand-over on the other.  A new batch is submitted as soon as fewer than `inject_below` fits are alive, so the
    round launches stay full while the slow fits of earlier batches (up to 200 sequential iterations each) finish
    inside them.  Each batch in flight needs its own candidate / row buffers (an Engine without fit workspace); the
    detection / consolidation workspace is shared, its users all being on the side stream.  Results per batch are
    bit-identical to Engine.run (fits are independent; only the order of execution changes).

    The reference's counterpart is the image loop of pflib.image_batch / parallel_image_batch (pflib.py:940-996,
    1082-1099)."""

    def __init__(self, n_fields, H, W, depth=16, cand_per_batch=None, inject_below=None, device=None,
                 mode=N.MODE_REF, cand_per_field=None, wide=False):
        """wide: a pipeline for PIXELS_U32 frames (its fit queue is created for 32-bit pixels and takes nothing else)."""
        torch = _torch()
        self.torch = torch
        self.wide = bool(wide)
        if self.wide and mode == N.MODE_TEXTBOOK_F32:
            raise NotImplementedError("the single-precision solver takes 16-bit pixels only")
        self.dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.n_fields, self.H, self.W = int(n_fields), int(H), int(W)
        self.depth = max(2, min(int(depth), N.MAX_TICKETS))
        self.side = torch.cuda.Stream(device=self.dev)
        self.shared_ws = torch.empty(Engine.workspace_bytes(self.n_fields, self.H, self.W), dtype=torch.uint8, device=self.dev)
        self.engines = [Engine(self.n_fields, self.H, self.W, device=self.dev, cand_per_field=cand_per_field,
                               fit_workspace=False, shared_ws=self.shared_ws) for _ in range(self.depth)]
        self.mode = mode
        self._inject_below_arg = inject_below
        self.fit_stream = torch.cuda.Stream(device=self.dev)       # the fit queue's stream (kept when the queue is rebuilt larger)
        # HIP binds a stream to one of the few hardware queues when the stream is first USED, round robin; two streams on one
        # hardware queue run their kernels one after the other.  Used here, right after its creation, the fit stream gets its
        # hardware queue in creation order (two pipelines side by side: different ones) - bound later, behind the side
        # streams' first detections, the two fit queues of bench.py shared one and lost their overlap (226 against 210 ms).
        with torch.cuda.device(self.dev), torch.cuda.stream(self.fit_stream):
            self._stream_pin = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.queue = None
        self.cand_per_batch = 0
        self.inject_below = int(inject_below or 0)
        # The queue is sized from what the FIRST batch's detection finds (+ 25 %), not from the engines' worst-case candidate
        # capacity (1 per 32 pixels): half the memory on bench.py's batches.  A later batch that does not fit waits until the
        # queue is empty and gets a larger one (run()).  cand_per_batch fixes the size up front instead.
        self.cand_per_batch_fixed = bool(cand_per_batch)
        if cand_per_batch:
            self._make_queue(int(cand_per_batch))

    def _make_queue(self, per):
        """(Re)build the fit queue for batches of up to `per` candidates.  A batch's pool slots are held until its last fit is
        done; the queue itself only ever holds the fits still alive."""
        if self.queue is not None:
            self.queue.close()
        self.cand_per_batch = per = int(per)
        # (the single-precision solver runs no rounds: its queue holds the by-candidate pool only)
        torch = self.torch
        # (one allocation at a time, and not under the caller's side stream)
        with _ALLOC_LOCK, torch.cuda.device(self.dev), torch.cuda.stream(torch.cuda.default_stream(self.dev)):
            self.queue = FitQueue(pool_slots=min((self.depth + 2) * per, (1 << 27) - 1),
                                  queue_cap=64 if self.mode == N.MODE_TEXTBOOK_F32 else 2 * per + per // 2,
                                  mode=self.mode | (N.PIXELS_U32_FLAG if self.wide else 0), device=self.dev, stream=self.fit_stream)
        # (the injection threshold stays a quarter of the engines' candidate capacity, as tuned on bench.py's batches)
        self.inject_below = int(self._inject_below_arg if self._inject_below_arg is not None else min(per, self.engines[0].cap) // 4
                                if self.cand_per_batch_fixed else self.engines[0].cap // 4)

    def run(self, jobs, on_done=None, r2_threshold=0.7, radius=4, py2_round=True):
        """jobs: iterable of (d_img, detect_params).  on_done(job_index, engine, total) is called, in order of
        completion, with the side stream current and the batch consolidated on it: whatever reads the engine's
        buffers must be enqueued on that stream inside the callback (the engine is re-used for a later job).
        Returns the per-job candidate totals."""
        if radius < 2:
            raise ValueError("consolidation_radius must be at least 2")
        torch, q = self.torch, self.queue
        free = list(self.engines)
        inflight, totals = {}, []
        pending, exhausted = None, False
        it = iter(enumerate(jobs))
        with torch.cuda.device(self.dev), torch.cuda.stream(self.side):
            while True:
                if pending is None and not exhausted and free:
                    try:
                        j, (d_img, prm) = next(it)
                    except StopIteration:
                        exhausted = True
                    else:
                        eng = free.pop()
                        total = eng.detect(d_img, prm)          # (synchronises the side stream: the count is needed)
                        totals.append(total)
                        if self.queue is None:
                            self._make_queue(total + total // 4 + 1024)
                            q = self.queue
                        pending = (j, eng, total, d_img, prm.pixel_format)
                if pending is not None and (not inflight or q.alive < self.inject_below):
                    j, eng, total, d_img, fmt = pending
                    t = q.submit(d_img, eng.n_fields, eng.H, eng.W, eng.cand, total, eng.rows, fmt)
                    if t is not None:
                        inflight[t] = pending
                        pending = None
                        continue                                 # detect the job after this one before advancing
                    if not inflight:        # the queue is empty and still too small for this batch: size it for the batch
                        if total <= self.cand_per_batch:
                            raise MemoryError("a batch of %d candidates does not fit the fit queue" % total)
                        self._make_queue(total + total // 4 + 1024)
                        q = self.queue
                        continue
                if not inflight:
                    if pending is None and exhausted:
                        break
                    continue
                q.advance(0, self.inject_below if (pending is not None or not exhausted) else 0)
                for t in sorted(inflight, key=lambda k: inflight[k][0]):
                    if not q.take(t, self.side):
                        continue
                    j, eng, total, d_img, fmt = inflight.pop(t)
                    eng.consolidate(r2_threshold, radius, py2_round)
                    if on_done is not None:
                        on_done(j, eng, total)
                    free.append(eng)
            self.side.synchronize()
        return totals

    def close(self):
        if self.queue is not None:
            self.queue.close()


class StreamPipelineGroup:
    """Several StreamPipelines side by side: the fields of every batch are split into contiguous shares, share k is
    worked by pipeline k (its own fit queue, its own pair of HIP streams, its own host thread).  A round of the LM fit is
    two or three kernels whose last waves run alone while the launch drains; with a second queue on other streams the
    hardware fills that ramp-down with the other queue's kernels (+3.4 % fits/s with two queues on bench.py's batches,
    nothing more with three).  Results per field are those of Engine.run."""

    def __init__(self, n_fields, H, W, queues=2, device=None, **kw):
        torch = _torch()
        self.torch = torch
        self.dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        q = max(1, min(int(queues), int(n_fields)))
        self.cut = [int(n_fields) * k // q for k in range(q + 1)]
        self.pipes = [StreamPipeline(self.cut[k + 1] - self.cut[k], H, W, device=self.dev, **kw) for k in range(q)]

    def run(self, jobs, on_done=None, **kw):
        """jobs: list of (d_img[n_fields, H, W], detect_params).  on_done(job_index, queue_index, engine, total) is called
        in queue k's thread with that pipeline's side stream current (see StreamPipeline.run); the engine holds the fields
        cut[k] .. cut[k+1] of the job.  Returns the candidate totals per job (summed over the shares)."""
        import threading
        jobs = list(jobs)
        totals = [None] * len(self.pipes)
        errs = []

        def body(k):
            try:
                self.torch.cuda.set_device(self.dev)
                lo, hi = self.cut[k], self.cut[k + 1]
                cb = None if on_done is None else (lambda j, eng, total: on_done(j, k, eng, total))
                totals[k] = self.pipes[k].run([(d[lo:hi], prm) for d, prm in jobs], cb, **kw)
            except BaseException as e:      # noqa: BLE001 - handed to the caller below
                errs.append(e)

        ths = [threading.Thread(target=body, args=(k,), daemon=True) for k in range(len(self.pipes))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]
        return [sum(t[j] for t in totals) for j in range(len(jobs))]

    def close(self):
        for p in self.pipes:
            p.close()


def fit_rois(rois, mode=N.MODE_REF):
    """LM-fit stand-alone 5x5 ROIs (uint16[n,5,5]); returns the FsqRow table on the host."""
    torch = _torch()
    r = as_u16_fields(rois).reshape(-1, 25)
    d = to_device_u16(r)
    rows = torch.empty((len(r), 128), dtype=torch.uint8, device=d.device)
    ws = torch.empty(N.lib().fsq_fit_workspace_bytes(len(r)), dtype=torch.uint8, device=d.device)
    rc = N.lib().fsq_fit_rois(d.data_ptr(), len(r), mode, rows.data_ptr(), ws.data_ptr(), ws.numel(),
                              torch.cuda.current_stream().cuda_stream)
    N.check(rc, "fsq_fit_rois")
    return rows.cpu().numpy().view(N.ROW_DTYPE).reshape(-1), rows

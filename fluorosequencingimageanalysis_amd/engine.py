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
    """uint16 numpy array (any shape) -> int16-typed torch tensor on the GPU holding the same bytes."""
    torch = _torch()
    a = np.ascontiguousarray(images, dtype=np.uint16)
    return torch.from_numpy(a.view(np.int16)).to(device or "cuda", non_blocking=False)


def as_u16_fields(image):
    """Validate one image / a stack like the reference's `image.astype(np.int64)` input, as uint16."""
    a = np.asarray(image)
    if a.dtype == np.uint16:
        return np.ascontiguousarray(a)
    if a.dtype.kind not in "iub":
        raise NotImplementedError("only integer pixel data is supported (got %s)" % a.dtype)
    if a.size and (a.min() < 0 or a.max() > 65535):
        raise NotImplementedError("pixel values outside [0, 65535] are not supported by the GPU path")
    return np.ascontiguousarray(a.astype(np.uint16))


def detect_params(median_filter_size, correlation_matrix, c_std):
    K = np.asarray(correlation_matrix)
    if K.ndim != 2 or K.shape[0] != K.shape[1] or K.shape[0] % 2 == 0:
        raise ValueError("correlation_matrix must be square, with an odd number of rows and columns")
    if K.shape[0] > 9 or not (1 <= int(median_filter_size) <= 9):
        raise NotImplementedError("correlation_matrix / median_filter_size larger than 9 are not supported")
    p = N.FsqDetectParams()
    p.median_filter_size = int(median_filter_size)
    p.ksz = int(K.shape[0])
    p.c_std = float(c_std)
    flat = K.astype(np.int64).ravel()
    for i, v in enumerate(flat):
        p.K[i] = int(v)
    return p


class DeviceBatch:
    """Results of one detect+fit+consolidate pass, still on the GPU."""
    __slots__ = ("n_fields", "H", "W", "cand", "counts", "offsets", "rows", "keep", "nkeep", "total", "thr")


class Engine:
    """Re-usable buffers for a fixed (n_fields, H, W) batch shape on one GPU."""

    def __init__(self, n_fields, H, W, device=None, cand_per_field=None):
        torch = _torch()
        self.torch = torch
        self.dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.L = N.lib()
        self.n_fields, self.H, self.W = int(n_fields), int(H), int(W)
        ws = self.L.fsq_detect_workspace_bytes(self.n_fields, self.H, self.W)
        ws2 = self.L.fsq_consolidate_workspace_bytes(self.n_fields, self.H, self.W)
        if ws < 0 or ws2 < 0:
            raise ValueError("invalid batch shape")
        self.ws = torch.empty(max(ws, ws2), dtype=torch.uint8, device=self.dev)
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
        self.fit_ws = torch.empty(self.L.fsq_fit_workspace_bytes(self.cap), dtype=torch.uint8, device=self.dev)

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

    def fit(self, d_img, total, mode=N.MODE_REF):
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
        self.fit(d_img, total, mode)
        self.consolidate(r2_threshold, radius, py2_round)
        return total

    # ---- host-side extraction --------------------------------------------------------------
    def kept_tables(self, total):
        """Copy the consolidated tables to the host: list (per field) of (rows, fit_imgs) or None when the
        reference's re-key assertion fired for that field."""
        torch = self.torch
        nkeep = self.nkeep.cpu().numpy()
        offsets = self.offsets.cpu().numpy()
        keep = self.keep[:max(total, 1)].cpu().numpy()
        idx_parts = [keep[offsets[f]:offsets[f] + max(int(nkeep[f]), 0)] for f in range(self.n_fields)]
        idx = np.concatenate(idx_parts) if idx_parts else np.zeros(0, np.int32)
        out = []
        if len(idx):
            d_idx = torch.from_numpy(idx.astype(np.int64)).to(self.dev)
            rows = self.rows[:total].index_select(0, d_idx).cpu().numpy().view(N.ROW_DTYPE).reshape(-1)
            fit = torch.empty((len(idx), 25), dtype=torch.float64, device=self.dev)
            d_idx32 = d_idx.to(torch.int32)
            rc = self.L.fsq_fit_images(self.rows.data_ptr(), d_idx32.data_ptr(), len(idx), fit.data_ptr(), self._stream())
            N.check(rc, "fsq_fit_images")
            fit = fit.cpu().numpy().reshape(-1, 5, 5)
        else:
            rows = np.zeros(0, N.ROW_DTYPE)
            fit = np.zeros((0, 5, 5))
        pos = 0
        for f in range(self.n_fields):
            k = int(nkeep[f])
            if k < 0:
                out.append(None)
                continue
            out.append((rows[pos:pos + k], fit[pos:pos + k]))
            pos += k
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

"""Drop-in for the reference's `phase_correlate` module (phase_correlate.py:11-196), computed on an MI355X:
rocFFT 2-D transforms + hand-written HIP kernels (cross-power spectrum, complex argmax, upsampled
matrix-multiply DFT) behind `fsq_phase_correlate` of the C ABI (include/fsq.h).  No CPU fallback."""
import numpy as np

from . import _native as N
from . import engine as _engine


class Registrar:
    """Re-usable workspace for registering batches of n pairs of H x W images (fsq_phase_correlate).
    Inputs are device tensors: uint16 camera frames as they sit in HBM (int16-typed torch tensors holding the same
    bytes, engine.to_device_u16) or float64.  register() only enqueues on the current stream."""

    def __init__(self, n_pairs, H, W, upsample_factor=1, dtype=N.DTYPE_U16, device=None):
        torch = _engine._torch()
        self.torch = torch
        self.dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
        self.n, self.H, self.W, self.uf, self.dtype = int(n_pairs), int(H), int(W), int(upsample_factor), int(dtype)
        self.L = N.lib()
        self.ws = None
        self.ws_stream = None

    def _workspace(self, stream):
        if self.ws is None or self.ws_stream != stream:        # (plans and their work areas are per stream)
            nbytes = self.L.fsq_phase_correlate_workspace_bytes(self.n, self.H, self.W, self.uf, self.dtype, stream)
            N.check(min(nbytes, 0), "fsq_phase_correlate_workspace_bytes")
            self.ws = self.torch.empty(nbytes, dtype=self.torch.uint8, device=self.dev)
            self.ws_stream = stream
        return self.ws

    def register(self, d_ref, d_reg, out=None):
        """-> float64[n, 4] device tensor (row_shift, col_shift, error, diffphase), enqueued on the current stream."""
        torch = self.torch
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        ws = self._workspace(stream)
        if out is None:
            out = torch.empty((self.n, 4), dtype=torch.float64, device=self.dev)
        rc = self.L.fsq_phase_correlate(d_ref.data_ptr(), d_reg.data_ptr(), self.dtype, self.n, self.H, self.W, self.uf,
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        N.check(rc, "fsq_phase_correlate")
        return out


def phase_correlate_batch(ref_images, reg_images, upsample_factor=1):
    """Register a stack of pairs [n, H, W] -> float64[n, 4] = (row_shift, col_shift, error, diffphase).
    uint16 stacks are uploaded and transformed as they are; anything else goes through float64 like the reference's
    np.array(image, dtype=np.float64) (phase_correlate.py:63-64)."""
    torch = _engine._torch()
    ref, reg = np.asarray(ref_images), np.asarray(reg_images)
    if ref.shape != reg.shape:
        raise ValueError("Error: images must be same size for phase_correlate")
    if ref.ndim != 3:
        raise ValueError("Error: phase_correlate only supports 2D images")
    n, H, W = ref.shape
    if ref.dtype == np.uint16 and reg.dtype == np.uint16:
        dtype = N.DTYPE_U16
        d_ref, d_reg = _engine.to_device_u16(ref), _engine.to_device_u16(reg)
    else:
        dtype = N.DTYPE_F64
        d_ref = torch.from_numpy(np.ascontiguousarray(ref, dtype=np.float64)).cuda()
        d_reg = torch.from_numpy(np.ascontiguousarray(reg, dtype=np.float64)).cuda()
    return Registrar(n, H, W, upsample_factor, dtype).register(d_ref, d_reg).cpu().numpy()


def phase_correlate(ref_image, reg_image, upsample_factor=1):
    """Efficient subpixel image registration by cross-correlation.  Reference phase_correlate.py:11-134.

    Returns (row_shift, col_shift, error, diffphase): the shift to apply to `reg_image` to bring it into
    registration with `ref_image` (opposite in sign to the shift that was applied to the content)."""
    ref_image = np.asarray(ref_image)
    reg_image = np.asarray(reg_image)
    if ref_image.shape != reg_image.shape:
        raise ValueError("Error: images must be same size for phase_correlate")
    if len(ref_image.shape) != 2:
        raise ValueError("Error: phase_correlate only supports 2D images")
    r = phase_correlate_batch(ref_image[None], reg_image[None], upsample_factor)[0]
    if upsample_factor == 1:        # the reference returns integer pixel shifts in this branch (:75-92)
        return np.int64(r[0]), np.int64(r[1]), np.float64(r[2]), np.float64(r[3])
    return np.float64(r[0]), np.float64(r[1]), np.float64(r[2]), np.float64(r[3])


def offsets_from_frames(alignment_frames, upsample_factor=20):
    """Frame-to-frame alignment of one field's cycle stack: the loop of SequenceExperiment.offsets_from_frames
    (flexlibrary.py:1717-1741) - offsets[0] = (0, 0) and offsets[f + 1] = phase_correlate(frame f, frame f + 1)[:2],
    each relative to the PREVIOUS frame - with all pairs registered in one batched GPU call.
    `alignment_frames`: sequence of 2-D images (or objects with an `.image` attribute, like flexlibrary.Image)."""
    imgs = [np.asarray(getattr(f, "image", f)) for f in alignment_frames]
    if not all(im.dtype == np.uint16 for im in imgs):
        imgs = [np.asarray(im, dtype=np.float64) for im in imgs]
    offsets = [(0, 0) for _ in imgs]
    if len(imgs) < 2:
        return offsets
    for im in imgs:
        if im.ndim != 2:
            raise ValueError("phase_correlate only supports 2D images.")
        if im.shape != imgs[0].shape:
            raise ValueError("phase_correlate requires images of the same shape.")
    out = phase_correlate_batch(np.stack(imgs[:-1]), np.stack(imgs[1:]), upsample_factor)
    for f in range(len(imgs) - 1):
        d_h, d_w = out[f][0], out[f][1]
        if upsample_factor == 1:             # the reference returns numpy ints on this branch (phase_correlate.py:85-92)
            d_h, d_w = np.int64(d_h), np.int64(d_w)
        offsets[f + 1] = (d_h, d_w)
    return offsets

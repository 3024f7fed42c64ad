"""Drop-in for the reference's `phase_correlate` module (phase_correlate.py:11-196), computed on an MI355X:
rocFFT 2-D transforms + hand-written HIP kernels (cross-power spectrum, complex argmax, upsampled
matrix-multiply DFT) behind `fsq_phase_correlate` of the C ABI (include/fsq.h).  No CPU fallback."""
import numpy as np

from . import _native as N
from . import engine as _engine


def phase_correlate_batch(ref_images, reg_images, upsample_factor=1):
    """Register a stack of pairs float64[n, H, W] -> float64[n, 4] = (row_shift, col_shift, error, diffphase)."""
    torch = _engine._torch()
    ref = np.ascontiguousarray(ref_images, dtype=np.float64)
    reg = np.ascontiguousarray(reg_images, dtype=np.float64)
    if ref.shape != reg.shape:
        raise ValueError("Error: images must be same size for phase_correlate")
    if ref.ndim != 3:
        raise ValueError("Error: phase_correlate only supports 2D images")
    n, H, W = ref.shape
    d_ref = torch.from_numpy(ref).cuda()
    d_reg = torch.from_numpy(reg).cuda()
    out = torch.empty((n, 4), dtype=torch.float64, device=d_ref.device)
    rc = N.lib().fsq_phase_correlate(d_ref.data_ptr(), d_reg.data_ptr(), n, H, W, int(upsample_factor), out.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    N.check(rc, "fsq_phase_correlate")
    return out.cpu().numpy()


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
    imgs = [np.asarray(getattr(f, "image", f), dtype=np.float64) for f in alignment_frames]
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

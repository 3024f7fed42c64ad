"""Spot photometry on the peak table - the per-spot metrics of flexlibrary.Spot computed for whole
tables of spots on the GPU.  Names and arguments follow flexlibrary.py:172-230."""

import numpy as np

from . import _native as N
from . import engine as _engine


def mexican_hat_photometry_metric(images, spots, brim_size=6, radius=9):
    """Spot.mexican_hat_photometry_metric (flexlibrary.py:172-210) for many spots at once.

    images: one 2-D integer image or a stack [n_fields, H, W] (values in [0, 2^31); beyond 65 535: fsq_mexican_hat_u32);
    spots:  int array [n, 2] of (h, w) centres for a single image, or [n, 3] of (field, h, w) for a stack.
    Returns float64[n]: sum(crown pixels) - len(crown) * median(brim pixels) over the (2*radius+1)^2 window, clipped at the
    image borders exactly as Spot.image_slice does (return_invalid=True behaviour)."""
    torch = _engine._torch()
    imgs, fmt = _engine.as_integer_fields(images)
    if imgs.ndim == 2:
        imgs = imgs[None]
    if imgs.ndim != 3:
        raise ValueError("images must be 2-D or a [n_fields, H, W] stack")
    n_fields, H, W = imgs.shape
    sp = np.asarray(spots, dtype=np.int64)
    if sp.ndim != 2 or sp.shape[1] not in (2, 3):
        raise ValueError("spots must be [n, 2] (h, w) or [n, 3] (field, h, w)")
    if sp.shape[1] == 2:
        if n_fields != 1:
            raise ValueError("(h, w) spots need a single image; use (field, h, w) for a stack")
        sp = np.concatenate([np.zeros((len(sp), 1), np.int64), sp], axis=1)
    if len(sp) and (sp[:, 0].min() < 0 or sp[:, 0].max() >= n_fields):
        raise ValueError("field index out of range")
    if len(sp) == 0:
        return np.zeros(0)
    d_img = _engine.to_device_pixels(imgs, fmt)
    d_sp = torch.from_numpy(np.ascontiguousarray(sp.astype(np.int32))).to(d_img.device)
    d_out = torch.empty(len(sp), dtype=torch.float64, device=d_img.device)
    rc = (N.lib().fsq_mexican_hat_u32 if fmt == N.PIXELS_U32 else N.lib().fsq_mexican_hat)(d_img.data_ptr(), n_fields, H, W, d_sp.data_ptr(), len(sp), int(brim_size), int(radius),
                                 d_out.data_ptr(), torch.cuda.current_stream(d_img.device).cuda_stream)
    N.check(rc, "fsq_mexican_hat")
    return d_out.cpu().numpy()


def gaussian_volume_photometry_metric(gaussian_fits, scaling=10 ** 6):
    """Spot.gaussian_volume_photometry_metric (flexlibrary.py:212-230): float(scaling) * A * sigma_h * sigma_w for a
    list of pflib.find_peptides tuples (or an [n, >=6] array in the tuple's order)."""
    f = np.array([[float(x) for x in g[:7]] for g in gaussian_fits], dtype=np.float64).reshape(-1, 7)
    return (float(scaling) * f[:, 3]) * f[:, 4] * f[:, 5]

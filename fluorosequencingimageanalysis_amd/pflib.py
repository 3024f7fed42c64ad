"""Drop-in for the per-field image path of the reference's `pflib` module, computed on an MI355X.

Same names, keyword arguments, defaults, return shapes and exceptions as the reference functions
(file:line of each is given in its docstring); the arithmetic is done by hand-written HIP kernels
behind the C ABI of include/fsq.h and reproduces the reference's fp64 results bit for bit (see
DESIGN.md).  There is no CPU fallback: without libfsq_hip.so / a GPU every compute entry point raises.

Batch extension (not in the reference): `find_peptides_batch(images, ...)` runs many same-sized
fields in one pass and is what bench.py and the multi-GPU driver use.
"""
import csv
import logging
import math
import os
import pickle
import time

import numpy as np

from . import _native as N
from . import engine as _engine

logger = logging.getLogger(__name__)
logger.addHandler(logging.NullHandler())

default_correlation_matrix = _engine.DEFAULT_CORRELATION_MATRIX.copy()      # pflib.py:48-52

#: The reference is Python 2: dict keys use round-half-away-from-zero (pflib.py:515).
PY2_ROUND = True


def _psf_candidates(image, median_filter_size=5, correlation_matrix=default_correlation_matrix, c_std=2, **kwargs):
    """Candidate pixels for PSF fitting, as a list [(h, w), ...] in raster order.  Reference pflib.py:217-258."""
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std)      # ValueError as pflib.py:236-239
    img = _engine.as_u16_fields(image)
    if img.ndim != 2:
        raise ValueError("image must be two-dimensional")
    H, W = img.shape
    eng = _engine.Engine(1, H, W)
    total = eng.detect(_engine.to_device_u16(img), prm)
    cand, _, _ = eng.candidates(total)
    return [(int(h), int(w)) for _, h, w in cand]


def _fit_2d_gaussian(subimage, implementation='agpy'):
    """Fit a 2D Gaussian to a 5x5 pixel area -> (h_0, w_0, H, A, sigma_h, sigma_w, theta, fit_img).
    Reference pflib.py:180-214 (h_0/w_0 in the 5x5 frame, as the reference returns them)."""
    subimage = np.asarray(subimage)
    assert subimage.shape[0] == 5 and subimage.shape[1] == 5
    if implementation != 'agpy':
        raise NotImplementedError("Currently, only agpy is supported.")
    rows, d_rows = _engine.fit_rois(subimage.reshape(1, 5, 5))
    torch = _engine._torch()
    fit = torch.empty((1, 25), dtype=torch.float64, device=d_rows.device)
    N.check(N.lib().fsq_fit_images(d_rows.data_ptr(), None, 1, fit.data_ptr(), torch.cuda.current_stream().cuda_stream),
            "fsq_fit_images")
    r = rows[0]
    return (float(r["p2"]), float(r["p3"]), float(r["H"]), float(r["A"]), float(r["sigma_h"]), float(r["sigma_w"]),
            float(r["theta"]), fit.cpu().numpy().reshape(5, 5))


def illumina_s_n(sub_img):
    """(max(sub_img) - mean(edge)) / std(edge) over the one-pixel boundary.  Reference pflib.py:261-281."""
    sub_img = np.asarray(sub_img)
    if not (len(sub_img.shape) == 2 and sub_img.shape[0] == sub_img.shape[1]):
        raise ValueError("sub_img must be square, but has shape " + str(sub_img))
    n = sub_img.shape[0]
    edge = ([sub_img[h, w] for h in (0, -1) for w in range(n)] +
            [sub_img[h, w] for h in range(1, n - 1) for w in (0, -1)])
    return (np.amax(sub_img) - np.mean(edge)) / np.std(edge)


def _table_to_dict(img, rows, fit):
    """FsqRow table -> the reference's {(h, w): 12-tuple} (pflib.py:396-407, 475)."""
    out = {}
    for r, f in zip(rows, fit):
        h, w = int(r["h"]), int(r["w"])
        sub = img[h - 2:h + 3, w - 2:w + 3].astype(np.int64)
        out[(int(r["key_h"]), int(r["key_w"]))] = (
            np.float64(r["h0"]), np.float64(r["w0"]), np.float64(r["H"]), np.float64(r["A"]),
            np.float64(r["sigma_h"]), np.float64(r["sigma_w"]), np.float64(r["theta"]), sub, f.copy(),
            float(r["rmse"]), np.float64(r["r2"]), np.float64(r["s_n"]))
    return out


def find_peptides_batch(images, median_filter_size=5, correlation_matrix=default_correlation_matrix,
                        candidate_pixels=None, c_std=2, r_2_threshold=0.7, consolidation_radius=4,
                        fit_type='gauss', N_iter=10**3, engine=None):
    """find_peptides over a stack uint16[n, H, W] in one GPU pass -> list of n dicts."""
    if consolidation_radius < 2:
        raise ValueError("consolidation_radius must be at least 2")                # pflib.py:431-432
    if fit_type != 'gauss':
        raise NotImplementedError("fit_type='monte_carlo' draws from an unseeded RNG in the reference "
                                  "(pflib.py:117-177) and is not reproduced")
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std)
    imgs = _engine.as_u16_fields(images)
    if imgs.ndim != 3:
        raise ValueError("images must have shape (n, H, W)")
    n, H, W = imgs.shape
    eng = engine or _engine.Engine(n, H, W)
    d_img = _engine.to_device_u16(imgs)
    total = eng.run(d_img, prm, r_2_threshold, consolidation_radius, N.MODE_REF, PY2_ROUND)
    tables = eng.kept_tables(total)
    out = []
    for f, t in enumerate(tables):
        if t is None:
            raise AssertionError("field %d: re-keyed peak collides with an existing key (pflib.py:518)" % f)
        out.append(_table_to_dict(imgs[f], t[0], t[1]))
    return out


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


def save_psfs_pkl(psfs, image_path=None, timestamp_epoch=None, output_path=None):
    """Pickle the PSF dict with protocol 0, as the reference's cPickle.dump does (pflib.py:594-636)."""
    output_path = _output_path(image_path, timestamp_epoch, output_path, '.pkl')
    with open(output_path, 'wb') as f:
        pickle.dump(psfs, f, protocol=0)
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


def read_image(image_path):
    """-> (converted_path, image array).  The reference converts non-PNG files with ImageMagick and reads
    the PNG (pflib.py:714-746); PIL reads 16-bit TIFF/PNG directly, so no conversion is needed and
    converted_path is the image itself (or an existing `<path>.png`, which the reference prefers)."""
    from PIL import Image
    converted_path = image_path = os.path.abspath(image_path)
    if image_path[-4:] != '.png' and os.path.exists(image_path + '.png'):
        converted_path += '.png'
    with Image.open(converted_path) as im:
        return converted_path, np.array(im)


def image_batch(image_paths, find_peptides_parameters=None, timestamp_epoch=None):
    """Fit every image of a list; per-image failures are logged and skipped (pflib.py:883-996).
    Returns {original path: (converted image path, pkl path, csv path, png path)} (png is None here)."""
    if find_peptides_parameters is None:
        find_peptides_parameters = {}
    if timestamp_epoch is None:
        timestamp_epoch = time.time()
    out = {}
    seen = set()
    for p in image_paths:
        ap = os.path.abspath(p)
        if ap in seen:
            continue
        seen.add(ap)
        try:
            converted, img = read_image(ap)
            psfs = find_peptides(img, **find_peptides_parameters)
            pkl = save_psfs_pkl(psfs, image_path=ap, timestamp_epoch=timestamp_epoch)
            tab = save_psfs_csv(psfs, image_path=ap, timestamp_epoch=timestamp_epoch)
            out[p] = (converted, pkl, tab, None)
        except Exception as e:      # the reference swallows and logs every per-image failure
            logger.exception(e, exc_info=True)
            continue
    return out


def parallel_image_batch(image_paths, find_peptides_parameters=None, timestamp_epoch=None, num_processes=None):
    """Same contract as pflib.parallel_image_batch (pflib.py:1000-1111).  The reference balances images
    over worker processes; here one GPU process handles them and `num_processes` is accepted and ignored
    (multi-GPU sharding lives in fluorosequencingimageanalysis_amd.distributed)."""
    return image_batch(image_paths, find_peptides_parameters, timestamp_epoch)

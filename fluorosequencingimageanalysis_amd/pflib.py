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
    img, fmt = _engine.as_pixel_fields(image)
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std, fmt)  # ValueError as pflib.py:236-239
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


def _records_to_dicts(rows, fit, sub, offs, failed=()):
    """Peak records of a batch (engine.split_peak_records) -> one {(h, w): 12-tuple} per field, in the reference's
    dict order (pflib.py:396-407, 475, 514-519); fields listed in `failed` give an AssertionError instance instead."""
    cols = [rows[k].tolist() for k in ("key_h", "key_w")]
    f64 = [[np.float64(x) for x in rows[k]] for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")]
    rmse = rows["rmse"].tolist()
    r2 = [np.float64(x) for x in rows["r2"]]
    s_n = [np.float64(x) for x in rows["s_n"]]
    out = []
    for f in range(len(offs) - 1):
        if f in failed:
            out.append(AssertionError("field %d: re-keyed peak collides with an existing key (pflib.py:518)" % f))
            continue
        d = {}
        for i in range(int(offs[f]), int(offs[f + 1])):
            d[(cols[0][i], cols[1][i])] = (f64[0][i], f64[1][i], f64[2][i], f64[3][i], f64[4][i], f64[5][i], f64[6][i],
                                           sub[i], fit[i], rmse[i], r2[i], s_n[i])
        out.append(d)
    return out


def _engine_dicts(eng, d_img, pixel_format=N.PIXELS_U16):
    """Consolidated results of an Engine pass -> list of per-field dicts (AssertionError instances for failed fields)."""
    rec, offs = eng.peak_records(d_img)
    nkeep = eng.nkeep.cpu().numpy()
    rows, fit, sub = _engine.split_peak_records(rec.cpu().numpy(), pixel_format)
    failed = set(int(f) for f in np.nonzero(nkeep[:eng.n_fields] < 0)[0])
    return _records_to_dicts(rows, fit, sub, offs.cpu().numpy(), failed)


#: find_peptides_batch works stacks larger than this many pixels as a stream of chunks (engine.StreamPipeline)
MAX_PIXELS_PER_PASS = 1024 * 512 * 512


def find_peptides_batch(images, median_filter_size=5, correlation_matrix=default_correlation_matrix,
                        candidate_pixels=None, c_std=2, r_2_threshold=0.7, consolidation_radius=4,
                        fit_type='gauss', N_iter=10**3, engine=None, errors='raise'):
    """find_peptides over a stack uint16[n, H, W] -> list of n dicts.

    One GPU pass for a stack of up to MAX_PIXELS_PER_PASS pixels; larger stacks are cut into equal chunks that are
    streamed through engine.StreamPipeline (continuous batching of the LM fits).  errors='return' puts the
    AssertionError of a field whose re-key collides (pflib.py:518) in that field's place instead of raising it."""
    if consolidation_radius < 2:
        raise ValueError("consolidation_radius must be at least 2")                # pflib.py:431-432
    if fit_type != 'gauss':
        raise NotImplementedError("fit_type='monte_carlo' draws from an unseeded RNG in the reference "
                                  "(pflib.py:117-177) and is not reproduced")
    # (candidate_pixels: "Not yet implemented" in the reference, pflib.py:374 - accepted and ignored there and here)
    imgs, fmt = _engine.as_pixel_fields(images)            # integer dtypes, or float16 (fp16 pixel loads)
    prm = _engine.detect_params(median_filter_size, correlation_matrix, c_std, fmt)
    if imgs.ndim != 3:
        raise ValueError("images must have shape (n, H, W)")
    n, H, W = imgs.shape
    if n == 0:
        return []
    per = max(1, MAX_PIXELS_PER_PASS // (H * W))
    if engine is not None or n <= per:
        eng = engine or _engine.Engine(n, H, W)
        d_img = _engine.to_device_u16(imgs)
        eng.run(d_img, prm, r_2_threshold, consolidation_radius, N.MODE_REF, PY2_ROUND)
        out = _engine_dicts(eng, d_img, fmt)
    else:
        n_chunks = -(-n // per)
        per = -(-n // n_chunks)
        pad = n_chunks * per - n                # the last chunk is filled up with copies of the last field
        pipe = _engine.StreamPipeline(per, H, W, depth=min(8, n_chunks + 1))
        out = [None] * (n_chunks * per)
        bufs = {}

        def jobs():
            for c in range(n_chunks):
                part = imgs[c * per:(c + 1) * per]
                if len(part) < per:
                    part = np.concatenate([part, np.repeat(part[-1:], pad, axis=0)])
                bufs[c] = _engine.to_device_u16(part)
                yield bufs[c], prm

        def on_done(c, eng, total):
            out[c * per:(c + 1) * per] = _engine_dicts(eng, bufs.pop(c), fmt)

        try:
            pipe.run(jobs(), on_done, r_2_threshold, consolidation_radius, PY2_ROUND)
        finally:
            pipe.close()
        out = out[:n]
    if errors == 'raise':
        for d in out:
            if isinstance(d, Exception):
                raise d
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


class _Py2Pickler(pickle._Pickler):
    """Protocol-0 pickler whose numpy globals carry the module paths of numpy 1.x (`numpy.core.multiarray`), which
    numpy 2 still resolves: the files are read back by the reference's Python 2 (flexlibrary.py:541-547), whose
    numpy has no `numpy._core`."""

    def save_global(self, obj, name=None):
        mod = getattr(obj, "__module__", None) or ""
        if mod.startswith("numpy._core"):
            name = name or getattr(obj, "__qualname__", None) or obj.__name__
            self.write(pickle.GLOBAL + ("numpy.core" + mod[len("numpy._core"):]).encode("ascii") + b"\n" +
                       name.encode("ascii") + b"\n")
            self.memoize(obj)
            return
        pickle._Pickler.save_global(self, obj, name)

    dispatch = dict(pickle._Pickler.dispatch)
    import types as _types
    dispatch[_types.FunctionType] = save_global
    dispatch[_types.BuiltinFunctionType] = save_global
    del _types


def save_psfs_pkl(psfs, image_path=None, timestamp_epoch=None, output_path=None):
    """Pickle the PSF dict with protocol 0, as the reference's cPickle.dump does (pflib.py:594-636)."""
    output_path = _output_path(image_path, timestamp_epoch, output_path, '.pkl')
    with open(output_path, 'wb') as f:
        _Py2Pickler(f, protocol=0).dump(psfs)
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
                im.save(output_path, format=output_format.upper())
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


def save_psfs_png(psfs, image_path=None, timestamp_epoch=None, output_path=None, **kwargs):
    """The reference draws the fitted spots over a contrast-stretched copy of the image (pflib.py:749-880).  That
    overlay is cosmetic and outside the hot path; nothing is written and None is returned, which is
    what image_batch's result tuple then carries in its png slot."""
    return None


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
    pflib.parallel_image_batch balances its workers by (pflib.py:1043-1050).  Same-shaped images share a GPU pass."""
    log = logging.getLogger()
    prm = _engine.detect_params(**{"median_filter_size": 5, "correlation_matrix": default_correlation_matrix, "c_std": 2,
                                   **(detect_parameters or {})})
    out = [None] * len(image_paths)
    groups = {}
    for i, p in enumerate(image_paths):
        try:
            a = _engine.as_u16_fields(read_image(p)[1])
            if a.ndim != 2:
                raise ValueError("image must be two-dimensional")
            groups.setdefault(a.shape, []).append((i, a))
        except Exception as e:      # noqa: BLE001 - logged and skipped like pflib.py:1044-1048
            log.exception(e, exc_info=True)
    for (H, W), members in groups.items():
        eng = _engine.Engine(len(members), H, W, fit_workspace=False)
        eng.detect(_engine.to_device_u16(np.stack([a for _, a in members])), prm)
        counts = eng.counts.cpu().numpy()
        for k, (i, _) in enumerate(members):
            out[i] = int(counts[k])
    return out


def image_batch(image_paths, find_peptides_parameters=None, timestamp_epoch=None):
    """Find PSFs in every image of a list and save them as pickle and CSV; per-image failures are logged and
    skipped (pflib.py:883-996).  Same-shaped images share one GPU pass.
    Returns {absolute image path: (converted image path, pkl path, csv path, png path)} (png: save_psfs_png)."""
    log = logging.getLogger()
    if timestamp_epoch is None:
        timestamp_epoch = _py2_round(time.time())
    paths = list(dict.fromkeys(os.path.abspath(p) for p in image_paths))       # absolute, duplicates dropped (:947-949)
    if find_peptides_parameters is None:
        find_peptides_parameters = {}
    read = []
    for ap in paths:
        try:
            converted, img = read_image(ap)
        except Exception as e:      # noqa: BLE001 - the reference swallows and logs every per-image failure (:960-964)
            log.exception(e, exc_info=True)
            continue
        read.append((ap, converted, img))
    results = _fit_image_groups([img for _, _, img in read], find_peptides_parameters)
    processed = {}
    for (ap, converted, img), psfs in zip(read, results):
        if isinstance(psfs, Exception):
            log.error("find_peptides failed for %s", ap, exc_info=(type(psfs), psfs, psfs.__traceback__))
            continue
        try:
            pkl = save_psfs_pkl(psfs, image_path=converted, timestamp_epoch=timestamp_epoch)
            tab = save_psfs_csv(psfs, image_path=converted, timestamp_epoch=timestamp_epoch)
            png = save_psfs_png(psfs, image_path=converted, timestamp_epoch=timestamp_epoch)
        except Exception as e:      # noqa: BLE001
            log.exception(e, exc_info=True)
            continue
        processed.setdefault(ap, (converted, pkl, tab, png))
    return processed


def parallel_image_batch(image_paths, find_peptides_parameters=None, timestamp_epoch=None, num_processes=None):
    """Same contract as pflib.parallel_image_batch (pflib.py:1000-1111).

    The reference balances the images over `num_processes` worker processes by candidate count.  Here the workers are
    the ranks of the torch.distributed job this process belongs to (one process per GPU, `torchrun`): images are
    balanced over them with the same longest-processing-time rule and every rank's results are merged on all ranks
    (fluorosequencingimageanalysis_amd.distributed.image_batch_sharded).  Without a process group one GPU handles all
    images.  num_processes is validated like the reference's (:1060-1061) and otherwise ignored."""
    if num_processes is not None and (num_processes < 1 or round(num_processes) != num_processes):
        raise ValueError("Number of processes must be an integer >= 1")
    from . import distributed as _dist
    if _dist.world_size() > 1:
        return _dist.image_batch_sharded(image_paths, find_peptides_parameters, timestamp_epoch)
    return image_batch(image_paths, find_peptides_parameters, timestamp_epoch)

"""Drop-in for the tracking entry points of the reference's `flexlibrary.Experiment`, computed on
the GPU from the peak tables:

    Experiment.accumulate_offsets        flexlibrary.py:567-593
    Experiment.discard_dropouts          flexlibrary.py:626-678
    Experiment.greedy_particle_tracking  flexlibrary.py:680-1027

Same names, arguments, return shapes and exceptions; `track_fields` is the batch form (many fields per launch) that
works directly on `(h, w)` tables such as the dict keys pflib.find_peptides returns.  The arithmetic is in
csrc/fsq_track.hip (fsq_greedy_tracking of include/fsq.h); there is no CPU fallback."""
import ctypes

import numpy as np

from . import _native as N
from . import engine as _engine
from .pflib import _py2_round


def track_fields(fields, offsets, frame_shape, candidate_radius=2, spot_radius=0, device=None):
    """Track the spots of many independent fields in one launch.

    fields:  list (per field) of lists (per frame) of integer arrays [n, 2] = (Spot.h, Spot.w); every field has the same
             number of frames.
    offsets: list (per field) of lists (per frame) of (d_h, d_w) relative to the previous frame, offsets[k][0] == (0, 0).
    Returns a list (per field) of (traces int32[n_traces, n_frames], n_discarded, prev int32[n], next int32[n],
    kept bool[n]): spot numbers count through the field's frames in order, -1 = no spot.  Raises ValueError /
    AssertionError where the reference does (first offset not (0, 0); two spots of one frame in one bin)."""
    torch = _engine._torch()
    dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
    if int(candidate_radius) != candidate_radius:
        raise TypeError("slice indices must be integers")          # the reference slices with candidate_radius (:905)
    n_fields = len(fields)
    if n_fields == 0:
        return []
    F = len(fields[0])
    H, W = int(frame_shape[0]), int(frame_shape[1])
    counts = np.zeros((n_fields, F), np.int32)
    parts = []
    for k, frames in enumerate(fields):
        if len(frames) != F or len(offsets[k]) != F:
            raise ValueError("every field needs the same number of frames and one offset per frame")
        for f, hw in enumerate(frames):
            a = np.asarray(hw)
            if a.size and not np.array_equal(a, np.rint(a)):
                raise NotImplementedError("Spot.h / Spot.w must be whole numbers (flexlibrary.py:449 makes them so)")
            a = a.astype(np.int32).reshape(-1, 2)
            counts[k, f] = len(a)
            parts.append(a)
    hw = np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros((0, 2), np.int32))
    start = np.concatenate([[0], np.cumsum(counts.sum(axis=1))]).astype(np.int32)
    total = int(start[-1])
    # (no limit on frames or spots since round 4: long time series keep their frame tables in the workspace, large fields read the
    # "has been paired" facts off the links instead of LDS bitmaps; the reference has no limit either)
    off = np.ascontiguousarray(np.array([[(float(o[0]), float(o[1])) for o in offs] for offs in offsets], dtype=np.float64))
    for k in range(n_fields):
        if off[k, 0, 0] != 0 or off[k, 0, 1] != 0:
            raise ValueError("The first image's offset must be (0, 0) by definiton.")           # flexlibrary.py:581-583
    L = N.lib()
    pair_cap = max(4096, 8 * int(counts.max()) if counts.size else 4096)
    while True:         # candidate pairs per frame are bounded only by the data: on overflow the list is doubled and the call repeated
        res = _track_launch(torch, dev, L, hw, start, counts, off, n_fields, F, H, W, candidate_radius, spot_radius, pair_cap, total)
        if not (res[0] == N.FSQ_ERANGE).any() or pair_cap >= (1 << 28):
            break
        pair_cap *= 4
    st, nt, nd, prev, nxt, kept, traces = res
    out = []
    for k in range(n_fields):
        if st[k] == N.FSQ_EASSERT:
            raise AssertionError("field %d: two spots of one frame round to the same bin of frame_bins "
                                 "(flexlibrary.py:851)" % k)
        N.check(int(st[k]), "fsq_greedy_tracking (field %d)" % k)
        a, b = int(start[k]), int(start[k + 1])
        out.append((traces[a:a + int(nt[k])].copy(), int(nd[k]), prev[a:b].copy(), nxt[a:b].copy(), kept[a:b].copy()))
    return out




def _track_launch(torch, dev, L, hw, start, counts, off, n_fields, F, H, W, candidate_radius, spot_radius, pair_cap, total):
    ws_bytes = L.fsq_track_workspace_bytes(n_fields, F, H, W, pair_cap)
    if ws_bytes < 0:
        raise ValueError("invalid tracking shape")
    t = lambda a: torch.from_numpy(a).to(dev)          # noqa: E731
    d_hw, d_start, d_counts, d_off = t(hw.reshape(-1)), t(start), t(counts.reshape(-1)), t(off.reshape(-1))
    d_prev = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    d_next = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    d_kept = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    d_traces = torch.empty(max(total, 1) * F, dtype=torch.int32, device=dev)
    d_nt = torch.empty(n_fields, dtype=torch.int32, device=dev)
    d_nd = torch.empty(n_fields, dtype=torch.int32, device=dev)
    d_st = torch.empty(n_fields, dtype=torch.int32, device=dev)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    rc = L.fsq_greedy_tracking(d_hw.data_ptr(), d_start.data_ptr(), d_counts.data_ptr(), d_off.data_ptr(), n_fields, F, H, W,
                               int(candidate_radius), float(spot_radius), d_prev.data_ptr(), d_next.data_ptr(),
                               d_kept.data_ptr(), d_traces.data_ptr(), d_nt.data_ptr(), d_nd.data_ptr(), d_st.data_ptr(),
                               pair_cap, ws.data_ptr(), ws_bytes, torch.cuda.current_stream(dev).cuda_stream)
    N.check(rc, "fsq_greedy_tracking")
    return (d_st.cpu().numpy(), d_nt.cpu().numpy(), d_nd.cpu().numpy(), d_prev.cpu().numpy(), d_next.cpu().numpy(),
            d_kept.cpu().numpy().astype(bool), d_traces.cpu().numpy().reshape(-1, F))


def centroid_track_fields(frames, init_hw, spot_field=None, search_radius=3, s_n_cutoff=3.0, offsets=None, device=None):
    """Luminosity-centroid tracking of many spots in many fields in one launch (fsq_centroid_tracking).

    frames integer[n_fields, F, H, W] (or [F, H, W] for one field; values below 2^31, beyond 65 535: fsq_centroid_tracking_u32); init_hw int[n, 2]; spot_field int[n] (default: all in
    field 0); offsets whole-pixel (d_h, d_w)[n_fields, F, 2] or None.  -> (hw int32[n, F, 2], present bool[n, F]);
    raises ValueError where the reference does (a search window that sums to zero)."""
    torch = _engine._torch()
    dev = torch.device(device or ("cuda:%d" % torch.cuda.current_device()))
    fr, fmt = _engine.as_integer_fields(frames)
    if fr.ndim == 3:
        fr = fr[None]
    if fr.ndim != 4:
        raise ValueError("frames must have shape (n_fields, F, H, W)")
    n_fields, F, H, W = fr.shape
    hw = np.ascontiguousarray(np.asarray(init_hw, dtype=np.int32).reshape(-1, 2))
    n = len(hw)
    sf = np.zeros(n, np.int32) if spot_field is None else np.ascontiguousarray(spot_field, dtype=np.int32)
    if len(sf) != n or (n and (sf.min() < 0 or sf.max() >= n_fields)):
        raise ValueError("spot_field must name a field for every spot")
    d_off = None
    if offsets is not None:
        off = np.asarray(offsets)
        if not np.array_equal(off, np.rint(off)):
            raise TypeError("slice indices must be integers")       # what the reference's image slicing raises (:1223)
        d_off = torch.from_numpy(np.ascontiguousarray(off.astype(np.int64).reshape(n_fields, F, 2))).to(dev)
    d_fr = _engine.to_device_pixels(fr, fmt, dev)
    d_hw, d_sf = torch.from_numpy(hw).to(dev), torch.from_numpy(sf).to(dev)
    d_out = torch.empty((max(n, 1), F, 2), dtype=torch.int32, device=dev)
    d_pres = torch.empty((max(n, 1), F), dtype=torch.uint8, device=dev)
    d_err = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = (N.lib().fsq_centroid_tracking_u32 if fmt == N.PIXELS_U32 else N.lib().fsq_centroid_tracking)(d_fr.data_ptr(), n_fields, F, H, W, d_hw.data_ptr(), d_sf.data_ptr(), n, int(search_radius),
                                       float(s_n_cutoff), d_off.data_ptr() if d_off is not None else None, d_out.data_ptr(),
                                       d_pres.data_ptr(), d_err.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    N.check(rc, "fsq_centroid_tracking")
    if int(d_err.item()):
        raise ValueError("cannot convert float NaN to integer")     # int(round(nan)) of an all-zero window's centroid
    return d_out[:n].cpu().numpy(), d_pres[:n].cpu().numpy().astype(bool)


class Spot(object):
    """A square of pixels in an image: the reference's Spot as far as tracking needs it (flexlibrary.py:74-112):
    `parent_Image` (anything with an `.image` array), integer centre (h, w), odd size, optional gaussian_fit tuple."""

    def __init__(self, parent_Image, h, w, size, gaussian_fit=None):
        self.parent_Image = parent_Image
        if size % 2 == 0:
            raise AttributeError("Spot.size must be odd.")
        self.size = size
        r, shape = (size - 1) // 2, parent_Image.image.shape
        if not (0 <= h - r and h + r < shape[0] and 0 <= w - r and w + r < shape[1]):
            if (gaussian_fit is None or not (r <= gaussian_fit[0] < shape[0] - r) and (r <= gaussian_fit[1] < shape[1] - r)):
                raise AttributeError("Spot area of size " + str(size) + " at " + str((h, w)) + " with gaussian_fit " +
                                     str(gaussian_fit) + " does not fit into parent_Image.image.shape of " + str(shape))
        self.h, self.w = h, w
        self.gaussian_fit = gaussian_fit

    def image_slice(self, radius=None):
        """The Spot's square of pixels, clipped at the image borders.  flexlibrary.py:113-146."""
        if radius is None:
            radius = (self.size - 1) // 2
        img = self.parent_Image.image
        return img[max(0, self.h - radius):min(img.shape[0], self.h + radius + 1),
                   max(0, self.w - radius):min(img.shape[1], self.w + radius + 1)]

    def valid_slice(self, radius=None):
        """Is the slice of the requested radius contained in the parent image.  flexlibrary.py:148-157."""
        if radius is None:
            radius = (self.size - 1) // 2
        sl = self.image_slice(radius=radius)
        return sl.shape[0] == sl.shape[1] == 2 * radius + 1

    def simple_photometry_metric(self, return_invalid=True):
        """Sum of the Spot's pixels.  flexlibrary.py:159-170."""
        if not return_invalid and not self.valid_slice():
            return None
        return np.sum(self.image_slice())

    def mexican_hat_photometry_metric(self, brim_size=6, radius=9, return_invalid=True):
        """sum(crown) - len(crown) * median(brim) over the clipped (2 radius + 1)^2 window.  flexlibrary.py:172-210
        (one spot through fsq_mexican_hat; photometry.mexican_hat_photometry_metric takes whole tables)."""
        from . import photometry as _ph
        if radius is None:
            radius = (self.size - 1) // 2
        if not return_invalid and not self.valid_slice(radius=radius):
            return None
        return _ph.mexican_hat_photometry_metric(self.parent_Image.image, [(self.h, self.w)], brim_size, radius)[0]

    def gaussian_volume_photometry_metric(self, scaling=10**6, default=0, return_invalid=True):
        """float(scaling) * A * sigma_h * sigma_w of the Gaussian fit.  flexlibrary.py:212-230."""
        if not return_invalid and not self.valid_slice():
            return None
        if self.gaussian_fit is None:
            return default
        return float(scaling) * self.gaussian_fit[3] * self.gaussian_fit[4] * self.gaussian_fit[5]

    def illumina_s_n(self):
        """pflib.illumina_s_n of the Spot's pixels.  flexlibrary.py:319-320."""
        from . import pflib as _pf
        return _pf.illumina_s_n(self.image_slice())


class Image(object):
    """A fluorosequencing image and its Spots: the reference's Image as far as the hot path needs it
    (flexlibrary.py:323-455): `image` (2-D array, or read from metadata['filepath'] with pflib.read_image), `metadata`,
    `spots`, and find_gaussian_psfs, which turns pflib.find_peptides' dict into Spot objects."""

    def __init__(self, image=None, metadata=None, spots=None):
        self.metadata = metadata if metadata is not None else {}
        if image is not None:
            self.image = image
        elif 'filepath' in self.metadata:
            from . import pflib as _pf
            self.image = _pf.read_image(self.metadata['filepath'])[1]
        else:
            raise AttributeError("Image.image must be defined: it was neither passed at initialization nor given a "
                                 "filepath to be read from.")
        self.spots = spots if spots is not None else []

    def _append_spots(self, new_fits, spots_append):
        if not spots_append:
            self.spots = []
        for (h, w), new_fit in new_fits.items():
            self.spots.append(Spot(self, int(_py2_round(h)), int(_py2_round(w)), 5, gaussian_fit=new_fit))
        return len(new_fits)

    def find_gaussian_psfs(self, pflib_args=None, spots_append=True):
        """Apply pflib.find_peptides to self.image and store the PSFs as Spots; returns their number.
        flexlibrary.py:426-455."""
        from . import pflib as _pf
        return self._append_spots(_pf.find_peptides(self.image, **(pflib_args or {})), spots_append)


def find_gaussian_psfs_batch(images, pflib_args=None, spots_append=True):
    """Image.find_gaussian_psfs for a list of same-shaped Images in one GPU pass (pflib.find_peptides_batch);
    returns the numbers of Spots found."""
    from . import pflib as _pf
    images = list(images)
    if not images:
        return []
    tables = _pf.find_peptides_batch(np.stack([np.asarray(im.image) for im in images]), **(pflib_args or {}))
    return [im._append_spots(t, spots_append) for im, t in zip(images, tables)]


class Experiment(object):
    """The static tracking helpers of the reference's Experiment class."""

    @staticmethod
    def easy_load_processed_image(image_filepath, psf_pkl_filepath=None, load_psfs=True):
        """Load a processed image and the PSF pickle pflib wrote for it into an Image with its Spots.
        Reference flexlibrary.py:516-564 (the resume mechanism of the experiment scripts: an image that has a
        `<image>*_psfs_*.pkl` next to it is not fitted again, basic_experiment_script.py:243-247, 377-399).

        image_filepath: path of the PNG image; psf_pkl_filepath: its pickle, or None to take the last of
        sorted(glob(image_filepath + '*_psfs_*.pkl')) - the file names end in the base-36 time stamp of
        pflib._psfs_filename, so that is the most recent one; load_psfs=False: the Image gets no Spots.
        Returns (Image, number of PSFs whose Spot could not be made): every PSF becomes a Spot at the Python-2-rounded
        dict key with size fit_img.shape[0]; a PSF for which Spot.__init__ raises is logged and counted."""
        import glob
        import logging
        import pickle
        from PIL import Image as _PILImage
        logger = logging.getLogger(__name__)
        with _PILImage.open(image_filepath) as f:
            image = np.array(f)
        image_object = Image(image=image, metadata={'filepath': image_filepath}, spots=None)
        discarded_spots = 0
        if load_psfs:
            if psf_pkl_filepath is None:
                pkl_files = sorted(glob.glob(image_filepath + '*_psfs_*.pkl'))
                if len(pkl_files) == 0:
                    raise ValueError("For image_filepath = " + image_filepath + " psf_pkl_filepath passed as None when " +
                                     "no pkl files available.")
                psf_pkl_filepath = pkl_files[-1]
            with open(psf_pkl_filepath, 'rb') as f:
                psfs = pickle.load(f, encoding='latin1')        # (latin1: files written by the reference's Python 2)
            spot_objects = []
            for (h, w), gaussian_fit in psfs.items():
                fit_img = gaussian_fit[8]
                try:
                    spot_objects.append(Spot(parent_Image=image_object, h=int(_py2_round(h)), w=int(_py2_round(w)),
                                             size=fit_img.shape[0], gaussian_fit=gaussian_fit))
                except Exception as e:      # noqa: BLE001 - as the reference: logged, counted, skipped
                    logger.info("flexlibrary.easy_load_processed_image: Ignoring Spot due to Spot.__init__ exception.")
                    logger.exception(e, exc_info=True)
                    discarded_spots += 1
            image_object.spots = spot_objects
        return image_object, discarded_spots

    @staticmethod
    def luminosity_centroid_particle_tracking(frames, initial_spots, search_radius=3, s_n_cutoff=3.0, offsets=None):
        """Follow Spots through frames by the centroid of pixel luminosity.  flexlibrary.py:1262-1317.
        frames: Images (objects with `.image`) of one shape; initial_spots: Spots of frames[0] (size 5).
        Returns one list per spot: its Spot in every frame, or None."""
        frames = list(frames)
        if not all(spot.parent_Image is frames[0] for spot in initial_spots):
            raise ValueError("All initial_spots must be in frames[0].")
        initial_spots = list(initial_spots)
        if not initial_spots:
            return []
        if any(s.size != 5 for s in initial_spots):
            raise NotImplementedError("luminosity-centroid tracking on the GPU handles Spots of size 5")
        stack = np.stack([np.asarray(f.image) for f in frames])
        off = None if offsets is None else np.asarray([(o[0], o[1]) for o in offsets])[None]
        hw, present = centroid_track_fields(stack, [(s.h, s.w) for s in initial_spots], None, search_radius, s_n_cutoff, off)
        tracks = []
        for spot, row, pr in zip(initial_spots, hw, present):
            tr = [spot]
            for f in range(1, len(frames)):
                tr.append(Spot(frames[f], int(row[f][0]), int(row[f][1]), spot.size) if pr[f] else None)
            tracks.append(tr)
        return tracks

    @staticmethod
    def accumulate_offsets(offsets):
        """Offsets relative to the preceding image -> offsets relative to the first.  flexlibrary.py:567-593."""
        if offsets[0] != (0, 0):
            raise ValueError("The first image's offset must be (0, 0) by definiton.")
        return [(sum([o[0] for o in offsets[:f + 1]]), sum([o[1] for o in offsets[:f + 1]])) for f in range(len(offsets))]

    @staticmethod
    def get_cumulative_offset(offsets, f, g=0):
        """Cumulative offset of frame f with respect to frame g.  flexlibrary.py:595-601."""
        cf = Experiment.accumulate_offsets(offsets)[f]
        cg = Experiment.accumulate_offsets(offsets)[g]
        return (cf[0] - cg[0], cf[1] - cg[1])

    @staticmethod
    def round_coordinates(h, w):
        return int(_py2_round(h)), int(_py2_round(w))             # flexlibrary.py:603-605 (Python 2 round)

    @staticmethod
    def apply_offset(coordinates, offset):
        return coordinates[0] + offset[0], coordinates[1] + offset[1]

    @staticmethod
    def unapply_offset(offset_coordinates, offset):
        return offset_coordinates[0] - offset[0], offset_coordinates[1] - offset[1]

    @staticmethod
    def offset_frame_coordinates(offsets, coordinate, f, g):
        """Given a coordinate in frame g, its coordinate in frame f.  flexlibrary.py:619-624."""
        return Experiment.apply_offset(coordinate, Experiment.get_cumulative_offset(offsets=offsets, f=f, g=g))

    @staticmethod
    def discard_dropouts(spots, spot_cumulative_offsets, frame_cumulative_offsets, image_shape, spot_radius=0):
        """Drop the Spots whose position falls outside some frame of the sequence.  flexlibrary.py:626-678.
        (Host arithmetic: a handful of comparisons per spot; the tracking kernel applies the same rule itself.)"""
        filtered, discarded = [], 0
        for i, spot in enumerate(spots):
            oh, ow = Experiment.apply_offset((spot.h, spot.w), spot_cumulative_offsets[i])
            for offset in frame_cumulative_offsets:
                gh, gw = Experiment.unapply_offset((oh, ow), offset)
                if not (spot_radius <= gh < image_shape[0] - 0.5 - spot_radius and
                        spot_radius <= gw < image_shape[1] - 0.5 - spot_radius):
                    discarded += 1
                    break
            else:
                filtered.append(spot)
        return filtered, discarded

    @staticmethod
    def greedy_particle_tracking(frame_spots, frame_shape, candidate_radius=2, offsets=None, spot_radius=0):
        """Track Spots across frames.  flexlibrary.py:680-1027.

        frame_spots: iterable (frames) of iterables of objects with integer `.h` / `.w`; offsets: (delta_h, delta_w) of
        every frame relative to the one before.  Returns (traces, number of discarded spots): one list per tracked spot
        holding its Spot object (or None) for every frame, in the reference's order."""
        frame_spots = [list(fr) for fr in frame_spots]
        if offsets is None:
            raise TypeError("'int' object is not iterable")       # the reference's default branch fails the same way (:787)
        res = track_fields([[np.array([(s.h, s.w) for s in fr]).reshape(-1, 2) for fr in frame_spots]],
                           [list(offsets)], frame_shape, candidate_radius, spot_radius)[0]
        flat = [s for fr in frame_spots for s in fr]
        traces = [[(flat[i] if i >= 0 else None) for i in row] for row in res[0]]
        return traces, res[1]


class SequenceExperiment(Experiment):
    """A sequence of frames of one field: the reference's SequenceExperiment as far as registration and tracking go
    (flexlibrary.py:1680-1810): `peptide_frames` / `alignment_frames` (Images), `offsets`, `spot_traces`."""

    def __init__(self, peptide_frames, alignment_frames=None, offsets=None, spot_traces=None, num_discarded_spots=0,
                 photometry_adjustments=None):
        self.peptide_frames = peptide_frames
        self.alignment_frames = alignment_frames
        self.offsets = offsets
        self.spot_traces = spot_traces
        self.num_discarded_spots = num_discarded_spots
        self.photometry_adjustments = photometry_adjustments

    def offsets_from_frames(self, upsample_factor=20):
        """Frame-to-frame alignment by phase correlation, all pairs in one GPU call.  flexlibrary.py:1717-1741."""
        from . import phase_correlate as _pc
        if self.alignment_frames is None:
            raise AttributeError("Calling offsets_from_frames without alignment_frames defined.")
        self.offsets = _pc.offsets_from_frames(self.alignment_frames, upsample_factor=upsample_factor)
        return self.offsets

    def trace_existing_spots(self, spot_radius=None):
        """greedy_particle_tracking over the Spots the frames already hold.  flexlibrary.py:1770-1809."""
        if spot_radius is not None:
            raise NotImplementedError("spot_radius currently not implemented")
        self.spot_traces, self.num_discarded_spots = Experiment.greedy_particle_tracking(
            frame_spots=[image.spots for image in self.peptide_frames], frame_shape=self.peptide_frames[0].image.shape,
            offsets=self.offsets, spot_radius=0)
        return self.spot_traces

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (container only).

TEST INFRASTRUCTURE.  Imports the reference from /root/reference through
oracle/refload.py (load-time py2->py3 transform, nothing copied to disk) and
records its outputs on seeded synthetic inputs made by
fluorosequencingimageanalysis_amd/synth.py.  The .npz files hold DATA only
(inputs or their seeds + the reference's outputs).

Usage (about 4 minutes on 8 cores):
  NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR" \
      python oracle/gen_golden.py [--only fields|reg|kat|phot|phot_wide|degen|params|tiny|wide|textbook|io|loader|track|track_long|centroid|centroid_wide]

Golden sets: G1 per-ROI fits, G2 candidate lists, G3 full
find_peptides tables, G4 phase_correlate tuples, G5 known-answer tests.
"""
import argparse
import multiprocessing as mp
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

NPY_ENV = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
if __name__ == "__main__" and os.environ.get("NPY_DISABLE_CPU_FEATURES") != NPY_ENV:
    os.environ["NPY_DISABLE_CPU_FEATURES"] = NPY_ENV      # numpy.exp/sin/cos == glibc's
    os.execv(sys.executable, [sys.executable] + sys.argv)

import numpy as np  # noqa: E402

from fluorosequencingimageanalysis_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# name -> (kind, args)   every field is rebuilt from these seeds by tests
FIELDS = {
    "f0_cfg1_512_200": dict(seed=1, shape=(512, 512), n_spots=200, kind="std"),
    "f1_cfg2_512_500": dict(seed=0, shape=(512, 512), n_spots=500, kind="std"),
    "f2_rect_384x640_300": dict(seed=7, shape=(384, 640), n_spots=300, kind="std"),
    "f3_hard_256": dict(seed=11, shape=(256, 256), n_spots=150, kind="hard"),
    "f4_dense_1024_1250": dict(seed=5, shape=(1024, 1024), n_spots=1250, kind="std"),
    "f5_small_96": dict(seed=21, shape=(96, 96), n_spots=12, kind="std"),
}


def build_field(spec):
    if spec["kind"] == "std":
        return synth.make_field(spec["seed"], spec["shape"], spec["n_spots"])
    return synth.make_hard_field(spec["seed"], spec["shape"], spec["n_spots"])


_R = None
TEXTBOOK = bool(int(os.environ.get("FSQ_GOLDEN_TEXTBOOK", "0")))   # run the reference with MINPACK's qrsolv (diagonal copied)


def _ref():
    global _R
    if _R is None:
        from refload import load_reference
        _R = load_reference(textbook_qrsolv=TEXTBOOK)
        holder = []

        class Rec(_R.mp.mpfit):
            def __init__(self, *a, **k):
                _R.mp.mpfit.__init__(self, *a, **k)
                holder.append(self)
        _R.gf.mpfit = Rec
        _R.holder = holder
    return _R


class _NoisyNumpy:
    """numpy proxy whose exp() is perturbed by a seeded +-1 ulp (stability probe)."""

    def __init__(self, seed):
        self._rng = np.random.default_rng(seed)

    def __getattr__(self, k):
        return getattr(np, k)

    def exp(self, x):
        y = np.exp(x)
        d = self._rng.integers(-1, 2, size=np.shape(y))
        return np.where(d > 0, np.nextafter(y, np.inf), np.where(d < 0, np.nextafter(y, -np.inf), y))


def fit_chunk(args):
    """Fit a chunk of ROIs with the reference; returns per-ROI records."""
    rois, noise_seed = args
    R = _ref()
    R.gf.numpy = _NoisyNumpy(noise_seed) if noise_seed is not None else np
    out = []
    for roi in rois:
        del R.holder[:]
        res = R.pf._fit_2d_gaussian(roi.astype(np.int64))
        m = R.holder[-1]
        # returned order (h_0,w_0,H,A,sh,sw,theta,fit) ; mpfit order p = (H,A,p2,p3,s4,s5,th)
        out.append((np.array(m.params, dtype=np.float64), int(m.status), int(m.niter),
                    int(m.nfev), float(m.fnorm), np.asarray(res[7], dtype=np.float64)))
    R.gf.numpy = np
    return out


def run_fits(pool, rois, noise_seed=None, chunk=64):
    chunks = [(rois[i:i + chunk], None if noise_seed is None else noise_seed * 100003 + i)
              for i in range(0, len(rois), chunk)]
    res = []
    for part in pool.imap(fit_chunk, chunks):
        res.extend(part)
    return res


def degenerate_images():
    """Small frames on which the fit leaves its usual path (flat / saturated / dim / pure-noise / hot-pixel content):
    early returns, zero-variance ROIs, non-finite quality metrics."""
    rng = np.random.default_rng(4711)
    out = {}
    out["d0_flat_32"] = np.full((32, 32), 100, np.uint16)
    out["d1_sat_40"] = np.full((40, 40), 65535, np.uint16)
    out["d2_satpart_48"] = np.minimum(synth.make_field(31, (48, 48), 6).astype(np.int64) * 30, 65535).astype(np.uint16)
    out["d3_dim_48"] = (synth.make_field(32, (48, 48), 6) // 40).astype(np.uint16)
    out["d4_noise_40"] = rng.integers(0, 3000, (40, 40)).astype(np.uint16)
    out["d5_noise_lo_36"] = rng.integers(0, 3, (36, 36)).astype(np.uint16)
    hot = np.full((32, 32), 100, np.uint16)
    hot[9, 11] = 60000
    hot[20, 21] = 60000
    hot[20, 22] = 30000
    out["d6_hotpixel_32"] = hot
    out["d7_zero_24"] = np.zeros((24, 24), np.uint16)
    return out


def parameter_cases():
    """find_peptides with other than its default keywords (median window, correlation matrix, c_std, r_2 threshold,
    consolidation radius) through the unmodified reference -> tests/golden/params_*.npz: pins the general detection kernel
    (any odd matrix up to 15 x 15, any median window) and the consolidation at other radii against the reference itself."""
    rng = np.random.default_rng(20240)
    k7 = rng.integers(-4000, 9000, (7, 7))
    k7[3, 3] = 40000
    k3 = np.array([[-1, 2, -1], [2, 12, 2], [-1, 2, -1]])
    k11 = -np.ones((11, 11), dtype=np.int64) * 700
    k11[3:8, 3:8] = 3000
    k11[5, 5] = 25000
    k15 = rng.integers(-900, 400, (15, 15))
    k15[5:10, 5:10] = pf_default = np.array([[-5935, -5935, -5935, -5935, -5935], [-5935, 8027, 8027, 8027, -5935],
                                             [-5935, 8027, 30742, 8027, -5935], [-5935, 8027, 8027, 8027, -5935],
                                             [-5935, -5935, -5935, -5935, -5935]])
    k13 = rng.integers(-300, 300, (13, 13))
    k13[4:9, 4:9] = pf_default // 2
    return {
        "p0_med3_k3_r2": dict(seed=61, shape=(112, 112), n_spots=14, kind="std",
                              params=dict(median_filter_size=3, correlation_matrix=k3, c_std=1.5, r_2_threshold=0.5, consolidation_radius=2)),
        "p1_med7_k7_r6": dict(seed=62, shape=(128, 96), n_spots=14, kind="std",
                              params=dict(median_filter_size=7, correlation_matrix=k7, c_std=2.5, r_2_threshold=0.8, consolidation_radius=6)),
        "p2_med4_k11_r9": dict(seed=63, shape=(120, 120), n_spots=25, kind="hard",
                               params=dict(median_filter_size=4, correlation_matrix=k11, c_std=1.0, r_2_threshold=0.3, consolidation_radius=9)),
        "p3_med9_k5_r3": dict(seed=64, shape=(96, 144), n_spots=30, kind="hard",
                              params=dict(median_filter_size=9, c_std=3, r_2_threshold=0.0, consolidation_radius=3)),
        # the largest windows the GPU path takes (FSQ_MAX_KSIZE = 15; 9 until round 4)
        "p4_med15_k15_r4": dict(seed=65, shape=(104, 120), n_spots=16, kind="std",
                                params=dict(median_filter_size=15, correlation_matrix=k15, c_std=2, r_2_threshold=0.6, consolidation_radius=4)),
        "p5_med11_k13_r5": dict(seed=66, shape=(128, 100), n_spots=30, kind="hard",
                                params=dict(median_filter_size=11, correlation_matrix=k13, c_std=1.5, r_2_threshold=0.4, consolidation_radius=5)),
    }


def tiny_cases():
    """Frames barely larger than - or as small as - one 5 x 5 neighbourhood, some with median windows and correlation matrices
    LARGER than the frame (scipy's 'reflect' indexing wraps more than once, the zero-padded correlation sees mostly padding)
    -> tests/golden/tiny_*.npz."""
    rng = np.random.default_rng(555)

    def spot(shape, c, amp, floor=200, noise=30):
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        g = amp * np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2) / (2 * 1.1 ** 2))
        return np.clip(np.rint(floor + g + rng.normal(0, noise, shape)), 0, 65535).astype(np.uint16)
    k7 = rng.integers(-2000, 3000, (7, 7))
    k7[3, 3] = 30000
    k9 = -np.ones((9, 9), dtype=np.int64) * 500
    k9[3:6, 3:6] = 4000
    return {
        "t0_5x5": dict(image=spot((5, 5), (2.2, 1.9), 4000)),
        "t1_5x9_c1": dict(image=spot((5, 9), (2.0, 4.3), 6000), params=dict(c_std=1)),
        "t2_6x6_med7": dict(image=spot((6, 6), (2.6, 3.1), 5000), params=dict(median_filter_size=7, c_std=1)),
        "t3_7x12_k7": dict(image=spot((7, 12), (3.3, 6.2), 9000), params=dict(correlation_matrix=k7, c_std=1, r_2_threshold=0.2)),
        "t4_9x5_k9_med9": dict(image=spot((9, 5), (4.4, 2.1), 7000),
                               params=dict(median_filter_size=9, correlation_matrix=k9, c_std=0.5, r_2_threshold=0.0, consolidation_radius=2)),
        "t5_8x8_med15_k15": dict(image=spot((8, 8), (3.7, 4.2), 8000),
                                 params=dict(median_filter_size=15, correlation_matrix=np.pad(k7, 4, constant_values=-100), c_std=0.5,
                                             r_2_threshold=0.1, consolidation_radius=3)),
    }


def wide_images():
    """Frames whose pixel values do not fit 16 bits (uint32 arrays; the reference computes on image.astype(np.int64),
    pflib.py:241, 443, whatever integer type it is handed) -> tests/golden/wide_*.npz, the vectors of FSQ_PIXELS_U32."""
    out = {}
    out["w0_22bit_128"] = synth.make_field(41, (128, 128), 14).astype(np.uint32) * 37 + 70000
    out["w1_28bit_96"] = synth.make_field(42, (96, 96), 10).astype(np.uint32) * 4099
    mixed = synth.make_field(43, (96, 96), 10).astype(np.uint32)          # 16-bit background, peaks far beyond it
    out["w2_mixed_96"] = np.where(mixed > 1500, mixed * 53, mixed).astype(np.uint32)
    hard = synth.make_hard_field(44, (112, 112), 40).astype(np.uint32)
    out["w3_hard_20bit_112"] = hard * 16 + (np.arange(112 * 112, dtype=np.uint32).reshape(112, 112) * 2654435761 >> 28)
    return out


def gen_fields(pool, with_stability=("f0_cfg1_512_200", "f1_cfg2_512_500", "f3_hard_256"), fields=None, prefix="field_"):
    R = _ref()
    for name, spec in (fields or FIELDS).items():
        img = spec["image"] if "image" in spec else build_field(spec)
        prm = dict(spec.get("params", {}))                  # find_peptides keywords other than the defaults
        det = {k: prm[k] for k in ("median_filter_size", "correlation_matrix", "c_std") if k in prm}
        cands = R.pf._psf_candidates(img, **det)
        cand = np.array(cands, dtype=np.int32).reshape(-1, 2)
        rois = [img[h - 2:h + 3, w - 2:w + 3] for h, w in cands]
        print(name, img.shape, "candidates", len(cands), flush=True)
        fits = run_fits(pool, rois)
        params = np.array([f[0] for f in fits]).reshape(-1, 7)
        status = np.array([f[1] for f in fits], dtype=np.int32)
        niter = np.array([f[2] for f in fits], dtype=np.int32)
        nfev = np.array([f[3] for f in fits], dtype=np.int32)
        fnorm = np.array([f[4] for f in fits])
        # replay the cached fits through the reference's own find_peptides
        it = iter(fits)
        roi_it = iter(rois)

        def replay(sub, implementation='agpy'):
            f = next(it)
            r = next(roi_it)
            assert np.array_equal(sub, r)
            p = f[0]
            return (p[2], p[3], p[0], p[1], p[4], p[5], p[6], f[5])
        orig = R.pf._fit_2d_gaussian
        R.pf._fit_2d_gaussian = replay
        table_error = 0
        try:
            table = R.pf.find_peptides(img, **prm)
        except AssertionError:          # pflib.py:518: a re-keyed peak lands on an existing key
            table, table_error = {}, 1
        finally:
            R.pf._fit_2d_gaussian = orig
        keys = np.array(list(table.keys()), dtype=np.int32).reshape(-1, 2)
        vals = list(table.values())
        tab7 = np.array([[float(x) for x in v[:7]] for v in vals]).reshape(-1, 7)
        tab_sub = np.array([v[7] for v in vals], dtype=np.int64).reshape(-1, 5, 5)
        tab_fit = np.array([v[8] for v in vals], dtype=np.float64).reshape(-1, 5, 5)
        tab_m = np.array([[float(v[9]), float(v[10]), float(v[11])] for v in vals]).reshape(-1, 3)
        extra = {"prm_" + k: np.asarray(v) for k, v in prm.items()}
        if name in with_stability:
            stable = np.ones(len(fits), dtype=bool)
            for k in (1, 2):
                alt = run_fits(pool, rois, noise_seed=k)
                ap = np.array([f[0] for f in alt]).reshape(-1, 7)
                ast = np.array([f[1] for f in alt], dtype=np.int32)
                rel = np.abs(ap[:, :6] - params[:, :6]) / np.maximum(np.abs(params[:, :6]), 1e-300)
                stable &= (rel.max(axis=1) <= 1e-6) & (ast == status)
            extra["stable"] = stable
            print("   stable fraction", stable.mean(), flush=True)
        np.savez_compressed(
            os.path.join(GOLD, "%s%s.npz" % (prefix, name)),
            seed=spec.get("seed", -1), shape=np.array(img.shape), n_spots=spec.get("n_spots", -1),
            kind=spec.get("kind", "image"), image_crc=np.uint32(zlib.crc32(img.tobytes())),
            image=img if img.size <= 256 * 256 else np.zeros((0, 0), np.uint16), table_error=table_error,
            candidates=cand, params=params, status=status, niter=niter, nfev=nfev, fnorm=fnorm,
            table_keys=keys, table7=tab7, table_sub=tab_sub, table_fit=tab_fit, table_metrics=tab_m,
            **extra)
        print("   peaks", len(keys), "status mix", np.unique(status, return_counts=True), flush=True)


def gen_reg():
    R = _ref()
    rng = np.random.default_rng(99)
    cases = []
    frames, off = synth.make_cycle_stack(3, n_cycles=5, shape=(512, 512), n_spots=500)
    for k in range(1, 5):
        cases.append(("cycle512_%d" % k, frames[k - 1], frames[k]))
    frames, off = synth.make_cycle_stack(4, n_cycles=4, shape=(200, 328), n_spots=150)
    for k in range(1, 4):
        cases.append(("cycle200x328_%d" % k, frames[k - 1], frames[k]))
    # odd sizes, negative / large shifts of smooth random content
    for i, (H, W, dy, dx) in enumerate([(63, 65, -3, 5), (101, 77, 7, -11), (64, 64, 0, 0),
                                        (33, 128, -16, 40), (127, 31, 20, -9), (50, 50, -25, 25),
                                        (97, 97, 12.4, -7.7), (81, 120, -0.35, 0.65), (16, 16, 1, 1)]):
        base = rng.normal(0, 1, (H + 64, W + 96))
        from scipy.ndimage import gaussian_filter, shift as ndshift
        base = gaussian_filter(base, 2.0) * 1000 + 500
        ref = base[32:32 + H, 48:48 + W]
        mov = ndshift(base, (dy, dx), order=3, mode="wrap")[32:32 + H, 48:48 + W]
        ref = np.clip(np.rint(ref), 0, 65535).astype(np.uint16)
        mov = np.clip(np.rint(mov + rng.normal(0, 2, mov.shape)), 0, 65535).astype(np.uint16)
        cases.append(("odd%d_%dx%d" % (i, H, W), ref, mov))
    out = {}
    names = []
    for name, a, b in cases:
        names.append(name)
        out["ref_" + name] = a
        out["reg_" + name] = b
        for uf in (1, 20, 100):
            r = R.pc.phase_correlate(a, b, upsample_factor=uf)
            out["out_%s_uf%d" % (name, uf)] = np.array([float(np.real(x)) for x in r])
            print(name, uf, out["out_%s_uf%d" % (name, uf)], flush=True)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "registration.npz"), **out)


def gen_kat():
    R = _ref()
    out = {}
    # qrfac worked example of the reference docstring (mpfit.py:1711-1742), run through the code
    m = R.mp.mpfit.__new__(R.mp.mpfit)
    m.debug = 0
    m.machar = R.mp.machar(double=1)
    a = np.array([[9., 4.], [2., 8.], [6., 7.]])
    aa, ipvt, rdiag, acnorm = m.qrfac(a.copy(), pivot=0)
    out["qrfac_in"], out["qrfac_a"], out["qrfac_ipvt"] = a, aa, ipvt
    out["qrfac_rdiag"], out["qrfac_acnorm"] = rdiag, acnorm
    rng = np.random.default_rng(5)
    J = rng.normal(0, 1, (25, 7)) * 10 ** rng.uniform(-2, 3, 7)
    aa, ipvt, rdiag, acnorm = m.qrfac(J.copy(), pivot=1)
    out["qrfac25_in"], out["qrfac25_a"], out["qrfac25_ipvt"] = J, aa, ipvt
    out["qrfac25_rdiag"], out["qrfac25_acnorm"] = rdiag, acnorm
    # enorm pins (numpy.dot through this container's OpenBLAS) : contiguous 25/7, strided columns
    vec25 = rng.normal(0, 1, (64, 25)) * 10 ** rng.uniform(-3, 3, (64, 1))
    vec7 = rng.normal(0, 1, (64, 7)) * 10 ** rng.uniform(-3, 3, (64, 1))
    out["enorm25_in"], out["enorm25_out"] = vec25, np.array([m.enorm(v) for v in vec25])
    out["enorm7_in"], out["enorm7_out"] = vec7, np.array([m.enorm(v) for v in vec7])
    out["enorm_col_in"] = J
    out["enorm_col_out"] = np.array([[m.enorm(J[j:, k]) for k in range(7)] for j in range(8)])
    # epoch hash round trips (pflib.py:523-566)
    epochs = np.array([0.4, 1.0, 35.0, 36.0, 1296.0, 1450000000.0, 1791084336.2, 2 ** 40 + 0.5])
    out["epochs"] = epochs
    out["epoch_hashes"] = np.array([R.pf._epoch_to_hash(e) for e in epochs])
    out["hash_epochs"] = np.array([R.pf._hash_to_epoch(h) for h in out["epoch_hashes"]], dtype=np.int64)
    # illumina_s_n (pflib.py:261-281)
    rois = rng.integers(50, 4000, (32, 5, 5)).astype(np.int64)
    out["sn_in"] = rois
    out["sn_out"] = np.array([R.pf.illumina_s_n(r) for r in rois])
    # twodgaussian model values (gaussfitter.py:63-140)
    P = np.column_stack([rng.uniform(0, 200, 64), rng.uniform(100, 3000, 64), rng.uniform(2, 3, 64),
                         rng.uniform(2, 3, 64), rng.uniform(.75, 2, 64), rng.uniform(.75, 2, 64),
                         rng.uniform(0, 360, 64)])
    P[:8, 6] = 0.0
    P[8:12, 6] = 360.0
    out["model_in"] = P
    out["model_out"] = np.array([R.gf.twodgaussian(p, 0, 1, 1)(*np.indices((5, 5))) for p in P])
    np.savez_compressed(os.path.join(GOLD, "kat.npz"), **out)
    print("kat done", flush=True)


def gen_photometry():
    """G6: Spot.mexican_hat_photometry_metric / gaussian_volume_photometry_metric of the reference
    (flexlibrary.py:172-230) on the reference's own peaks of two fields plus spots pushed against the borders.
    The parent image is handed over as int64 (what pflib.find_peptides works on, pflib.py:241): under numpy >= 2
    the reference's python sum() over uint16 scalars would wrap at 65 535, which the original numpy-1 run did not."""
    import refload
    ref = refload.load_flexlibrary()
    fl, pf = ref.fl, ref.pf

    class Parent(object):
        pass
    out = {}
    for name in ("f5_small_96", "f3_hard_256"):
        spec = FIELDS[name]
        img = build_field(spec)
        parent = Parent()
        parent.image = img.astype(np.int64)
        psfs = pf.find_peptides(img)
        hw, fit7, mh, mh2, vol = [], [], [], [], []
        H, W = img.shape
        extra = [(0, 0), (1, W - 2), (H - 1, W - 1), (H // 2, 3), (5, W // 2), (H - 4, 8), (9, 9), (H - 10, W - 10)]
        items = [((int(pf.round(v[0])), int(pf.round(v[1]))), v) for v in psfs.values()]
        items += [(e, None) for e in extra]
        for (h, w), v in items:
            sp = fl.Spot.__new__(fl.Spot)                  # border spots: skip the constructor's fit-inside check
            sp.parent_Image, sp.h, sp.w, sp.size, sp.gaussian_fit = parent, h, w, 5, v
            hw.append((h, w))
            mh.append(float(sp.mexican_hat_photometry_metric()))
            mh2.append(float(sp.mexican_hat_photometry_metric(brim_size=2, radius=4)))
            vol.append(float(sp.gaussian_volume_photometry_metric()) if v is not None else 0.0)
            fit7.append([float(x) for x in v[:7]] if v is not None else [0.0] * 7)
        out["hw_" + name] = np.array(hw, dtype=np.int32)
        out["fit7_" + name] = np.array(fit7)
        out["mexican_hat_b6_r9_" + name] = np.array(mh)
        out["mexican_hat_b2_r4_" + name] = np.array(mh2)
        out["gaussian_volume_" + name] = np.array(vol)
    out["names"] = np.array(["f5_small_96", "f3_hard_256"])
    np.savez_compressed(os.path.join(GOLD, "photometry.npz"), **out)
    print("photometry.npz", {k: v.shape for k, v in out.items()})


def gen_photometry_wide():
    """Spot.mexican_hat_photometry_metric of the reference (flexlibrary.py:172-210) with windows beyond the 31 x 31 the GPU kernel was
    limited to until round 4: (brim, radius) = (6, 16), (10, 40), (3, 150 - larger than the image) on the peaks of one field and on
    spots at and beyond the borders -> tests/golden/photometry_wide.npz."""
    import refload
    ref = refload.load_flexlibrary()
    fl = ref.fl

    class Parent(object):
        pass
    name = "f3_hard_256"
    img = build_field(FIELDS[name])
    parent = Parent()
    parent.image = img.astype(np.int64)         # (see gen_photometry: python sum() over uint16 scalars would wrap under numpy 2)
    g = np.load(os.path.join(GOLD, "field_%s.npz" % name))
    hw = [tuple(int(v) for v in k) for k in g["table_keys"].reshape(-1, 2)][::3]
    H, W = img.shape
    hw += [(0, 0), (1, W - 2), (H - 1, W - 1), (H // 2, 3), (5, W // 2), (H - 4, 8), (40, 40), (H - 41, W - 41), (128, 128)]
    out = {"name": np.array(name), "hw": np.array(hw, dtype=np.int32), "cases": np.array([(6, 16), (10, 40), (3, 150)])}
    for brim, radius in out["cases"]:
        vals = []
        for h, w in hw:
            sp = fl.Spot.__new__(fl.Spot)
            sp.parent_Image, sp.h, sp.w, sp.size, sp.gaussian_fit = parent, h, w, 5, None
            vals.append(float(sp.mexican_hat_photometry_metric(brim_size=int(brim), radius=int(radius))))
        out["mexican_hat_b%d_r%d" % (brim, radius)] = np.array(vals)
        print("photometry_wide", (int(brim), int(radius)), len(vals), "spots", flush=True)
    # ... and on the same field with pixel values beyond 16 bits (x 300: up to 24 bits)
    parent.image = img.astype(np.int64) * 300
    out["pixel_scale"] = np.int64(300)
    for brim, radius in ((6, 9), (10, 40)):
        vals = []
        for h, w in hw:
            sp = fl.Spot.__new__(fl.Spot)
            sp.parent_Image, sp.h, sp.w, sp.size, sp.gaussian_fit = parent, h, w, 5, None
            vals.append(float(sp.mexican_hat_photometry_metric(brim_size=int(brim), radius=int(radius))))
        out["scaled_mexican_hat_b%d_r%d" % (brim, radius)] = np.array(vals)
        print("photometry_wide, pixels x 300", (brim, radius), len(vals), "spots", flush=True)
    np.savez_compressed(os.path.join(GOLD, "photometry_wide.npz"), **out)


def tracking_cases():
    """name -> (frame_hw list of int arrays, offsets list of (d_h, d_w), shape, candidate_radius, spot_radius).
    Two cycle stacks of config 3 (spots = the oracle's find_peptides keys of every frame - the reference takes 87 s per
    512x512 frame for the same table, pinned equal by the field goldens; offsets = the reference's own phase_correlate at
    upsample_factor 20) and hand-made adversarial layouts: exact distance ties, competing ancestors, re-appearing
    spots, drift past the borders, a wider radius, integer offsets."""
    import oracle as O
    R = _ref()
    cases = {}
    for name, seed, shape, n_spots, n_cycles in (("stack256", 30, (256, 256), 150, 8), ("stack512", 3, (512, 512), 500, 6)):
        frames, _ = synth.make_cycle_stack(seed, n_cycles=n_cycles, shape=shape, n_spots=n_spots)
        hw = []
        for fr in frames:
            rows, fits, keep, key = O.find_peptides(fr, n_threads=os.cpu_count())
            hw.append(np.asarray(key, dtype=np.int32).reshape(-1, 2))
        offsets = [(0, 0)]
        for f in range(1, n_cycles):
            d_h, d_w, _, _ = R.pc.phase_correlate(frames[f - 1], frames[f], upsample_factor=20)
            offsets.append((float(d_h), float(d_w)))
        cases[name] = (hw, offsets, shape, 2, 0)
    rng = np.random.default_rng(77)
    # exact ties: lattice spots, integer offsets, descendants at distance exactly 1 from two ancestors
    base = np.array([(h, w) for h in range(6, 60, 6) for w in range(6, 60, 6)], dtype=np.int32)
    f1 = base + np.array([0, 1], np.int32)
    f2 = np.concatenate([base[::2] + np.array([1, 1], np.int32), base[1::2] + np.array([3, 3], np.int32)])
    f3 = base[rng.permutation(len(base))[:50]]
    cases["ties_int"] = ([base, f1, f2, f3], [(0, 0), (0, 0), (1, 0), (-1, 0)], (64, 64), 2, 0)
    # sub-pixel drift on the 1/20 grid with jitter, drop-outs that come back, spots pushed out of the field
    pts = rng.integers(3, 117, (140, 2)).astype(np.int32)
    pts = pts[np.unique(pts[:, 0] * 1000 + pts[:, 1], return_index=True)[1]]
    keep_far = [0]
    for i in range(1, len(pts)):
        if np.abs(pts[keep_far] - pts[i]).max(axis=1).min() > 3:
            keep_far.append(i)
    pts = pts[keep_far]
    offs, frames_hw, cum = [(0, 0)], [pts.copy()], np.zeros(2)
    for f in range(1, 7):
        step = np.round(rng.uniform(-2.5, 2.5, 2) * 20) / 20
        offs.append((float(step[0]), float(step[1])))
        cum = cum + step
        alive = rng.uniform(size=len(pts)) > 0.25
        jit = rng.integers(-1, 2, (len(pts), 2))
        frames_hw.append((np.rint(pts - cum).astype(np.int32) + jit)[alive])
    cases["drift_dropout"] = (frames_hw, offs, (120, 120), 2, 0)
    cases["drift_radius3_edge2"] = (frames_hw, offs, (120, 120), 3, 2)
    # two spots of different frames competing for one bin / one descendant
    cases["compete"] = ([np.array([[10, 10], [20, 20], [30, 30]], np.int32), np.array([[10, 11], [31, 30]], np.int32),
                         np.array([[10, 10], [20, 21], [30, 30], [30, 32]], np.int32), np.array([[11, 10], [20, 20], [30, 31]], np.int32)],
                        [(0, 0), (0.5, 0.5), (-0.5, -0.45), (0.05, 0)], (40, 40), 2, 0)
    cases["empty_frames"] = ([np.zeros((0, 2), np.int32), np.array([[5, 5]], np.int32), np.zeros((0, 2), np.int32),
                              np.array([[5, 6]], np.int32)], [(0, 0), (0, 0), (0, 0), (0, 0)], (12, 12), 2, 0)
    return cases


def tracking_long_cases():
    """Series beyond the limits the GPU tracker had until round 4 (64 frames, 32 768 spots per field): a 90-frame series with
    drift, drop-outs and re-appearing spots, and a two-frame field of 34 000 spots."""
    rng = np.random.default_rng(404)
    cases = {}
    pts = rng.integers(4, 92, (80, 2)).astype(np.int32)
    pts = pts[np.unique(pts[:, 0] * 1000 + pts[:, 1], return_index=True)[1]]
    far = [0]
    for i in range(1, len(pts)):
        if np.abs(pts[far] - pts[i]).max(axis=1).min() > 3:
            far.append(i)
    pts = pts[far]
    offs, frames_hw, cum = [(0, 0)], [pts.copy()], np.zeros(2)
    for f in range(1, 90):
        step = np.round(rng.uniform(-1.2, 1.2, 2) * 20) / 20
        if f % 17 == 0:
            step = -np.round(cum * 20) / 20                 # (the stage comes back to where it started)
        offs.append((float(step[0]), float(step[1])))
        cum = cum + step
        alive = rng.uniform(size=len(pts)) > 0.3
        jit = rng.integers(-1, 2, (len(pts), 2))
        hw = (np.rint(pts - cum).astype(np.int32) + jit)[alive]
        hw = hw[(hw[:, 0] >= 0) & (hw[:, 0] < 96) & (hw[:, 1] >= 0) & (hw[:, 1] < 96)]
        hw = hw[np.unique(hw[:, 0] * 1000 + hw[:, 1], return_index=True)[1]]
        frames_hw.append(hw[rng.permutation(len(hw))])
    cases["long90"] = (frames_hw, offs, (96, 96), 2, 0)
    cases["long90_radius3_edge3"] = (frames_hw[:70], offs[:70], (96, 96), 3, 3)
    big = rng.permutation(1024 * 1024)[:17000]
    a = np.stack([big // 1024, big % 1024], axis=1).astype(np.int32)
    moved = a[rng.uniform(size=len(a)) > 0.1]
    moved = np.clip(moved + rng.integers(-1, 2, moved.shape).astype(np.int32), 0, 1023)       # most spots again, a pixel off at most
    fresh = rng.permutation(1024 * 1024)[:1800]
    b = np.concatenate([moved, np.stack([fresh // 1024, fresh % 1024], axis=1).astype(np.int32)])
    b = b[np.unique(b[:, 0] * 2048 + b[:, 1], return_index=True)[1]]
    cases["many34000"] = ([a, b[rng.permutation(len(b))]], [(0, 0), (0.4, -0.35)], (1024, 1024), 2, 0)
    return cases


def gen_tracking(cases=None, out_name="tracking.npz"):
    """N1 goldens: Experiment.greedy_particle_tracking of the reference itself (flexlibrary.py:680-1027, loaded by
    refload.load_flexlibrary with Python-2 round) on the cases above -> tests/golden/tracking.npz."""
    import refload
    ref = refload.load_flexlibrary(_ref())
    fl = ref.fl

    class S(object):
        __slots__ = ("h", "w", "gid")

    out = {"names": []}
    for name, (frame_hw, offsets, shape, radius, spot_radius) in (cases if cases is not None else tracking_cases()).items():
        gid = 0
        frame_spots = []
        for hw in frame_hw:
            spots = []
            for h, w in hw:
                s = S()
                s.h, s.w, s.gid = int(h), int(w), gid         # Spot.h / Spot.w are python ints (flexlibrary.py:449)
                gid += 1
                spots.append(s)
            frame_spots.append(spots)
        traces, n_disc = fl.Experiment.greedy_particle_tracking(frame_spots, shape, candidate_radius=radius,
                                                               offsets=[tuple(o) for o in offsets], spot_radius=spot_radius)
        t = np.array([[(-1 if s is None else s.gid) for s in tr] for tr in traces], dtype=np.int32).reshape(-1, len(frame_hw))
        out["names"].append(name)
        out[name + "_counts"] = np.array([len(x) for x in frame_hw], dtype=np.int32)
        out[name + "_hw"] = (np.concatenate([np.asarray(x, np.int32).reshape(-1, 2) for x in frame_hw])).astype(np.int32)
        out[name + "_offsets"] = np.asarray(offsets, dtype=np.float64)
        out[name + "_shape"] = np.array(shape)
        out[name + "_radius"] = np.array([radius, spot_radius])
        out[name + "_traces"] = t
        out[name + "_discarded"] = np.int32(n_disc)
        print(name, "spots", gid, "traces", len(t), "discarded", n_disc, "full-length", int((t >= 0).all(axis=1).sum()), flush=True)
    out["names"] = np.array(out["names"])
    np.savez_compressed(os.path.join(GOLD, out_name), **out)


def centroid_cases():
    """name -> (frames uint16[F,H,W], init_hw, offsets int[F,2] or None, search_radius, s_n_cutoff)."""
    import oracle as O
    R = _ref()
    cases = {}
    frames, _ = synth.make_cycle_stack(41, n_cycles=7, shape=(160, 160), n_spots=60, max_drift=2.5, dropout=0.2)
    rows, fits, keep, key = O.find_peptides(frames[0], n_threads=os.cpu_count())
    offs = [(0, 0)]
    for f in range(1, len(frames)):
        d_h, d_w, _, _ = R.pc.phase_correlate(frames[f - 1], frames[f], upsample_factor=1)
        offs.append((int(d_h), int(d_w)))
    cases["stack160_registered"] = (frames, np.asarray(key, np.int32), np.array(offs, np.int64), 3, 3.0)
    cases["stack160_no_offsets_r2"] = (frames, np.asarray(key, np.int32), None, 2, 3.0)
    cases["stack160_strict"] = (frames, np.asarray(key, np.int32), np.array(offs, np.int64), 3, 12.0)
    # spots along the borders and in the corners of a noisy frame, big drifts
    rng = np.random.default_rng(9)
    noisy = rng.integers(90, 400, (5, 48, 48)).astype(np.uint16)
    noisy[:, 20:23, 20:23] += 3000
    edge = np.array([(2, 2), (2, 45), (45, 2), (45, 45), (3, 24), (24, 3), (44, 24), (24, 44), (21, 21), (5, 5), (10, 40)], np.int32)
    cases["borders"] = (noisy, edge, np.array([(0, 0), (2, -3), (-4, 1), (0, 5), (-6, -6)], np.int64), 3, 3.0)
    return cases


def centroid_wide_cases():
    """The centroid-tracking cases with pixel values beyond 16 bits (uint32 frames) -> tests/golden/centroid_tracking_wide.npz."""
    base = centroid_cases()
    frames, init, offs, sr, cut = base["stack160_registered"]
    wide = frames.astype(np.uint32) * 517 + 70000
    cases = {"stack160_wide_registered": (wide, init, offs, sr, cut),
             "stack160_wide_strict_r2": (wide, init, offs, 2, 12.0)}
    frames, init, offs, sr, cut = base["borders"]
    cases["borders_wide"] = (np.minimum(frames.astype(np.int64) * 600000, 2 ** 31 - 1).astype(np.uint32), init, offs, sr, cut)
    return cases


def gen_centroid(cases=None, out_name="centroid_tracking.npz"):
    """N4 goldens: Experiment.luminosity_centroid_particle_tracking of the reference (flexlibrary.py:1262-1317)."""
    import refload
    ref = refload.load_flexlibrary(_ref())
    fl = ref.fl

    class Img(object):
        def __init__(self, a):
            self.image = a
    out = {"names": []}
    for name, (frames, init_hw, offsets, sr, cut) in (cases if cases is not None else centroid_cases()).items():
        imgs = [Img(f) for f in frames]
        spots = [fl.Spot(imgs[0], int(h), int(w), 5) for h, w in init_hw]
        tracks = fl.Experiment.luminosity_centroid_particle_tracking(
            imgs, spots, search_radius=sr, s_n_cutoff=cut,
            offsets=None if offsets is None else [(int(a), int(b)) for a, b in offsets])
        hw = np.array([[(-1, -1) if s is None else (s.h, s.w) for s in tr] for tr in tracks], dtype=np.int32)
        out["names"].append(name)
        out[name + "_frames"] = frames
        out[name + "_init"] = np.asarray(init_hw, np.int32)
        out[name + "_offsets"] = np.zeros((0, 2), np.int64) if offsets is None else np.asarray(offsets, np.int64)
        out[name + "_params"] = np.array([sr, cut])
        out[name + "_hw"] = hw
        print(name, "spots", len(spots), "present fraction", float((hw[:, :, 0] >= 0).mean()), flush=True)
    out["names"] = np.array(out["names"])
    np.savez_compressed(os.path.join(GOLD, out_name), **out)


def gen_io():
    """On-disk formats: the reference's own save_psfs_csv / save_psfs_pkl / _psfs_filename on its own
    find_peptides result for field f5 (pflib.py:569-711).  The CSV text is what the reference writes under THIS
    interpreter (Python 3 prints floats with 17 significant digits where Python 2's str() printed 12): tests compare
    the header, the path column, the row order and the numbers."""
    import json
    import pickle
    import tempfile
    R = _ref()
    spec = FIELDS["f5_small_96"]
    img = build_field(spec)
    psfs = R.pf.find_peptides(img)
    d = tempfile.mkdtemp()
    fake_image = os.path.join(d, "plate 1", "f5_small_96.tif")
    os.makedirs(os.path.dirname(fake_image))
    epoch = 1450000000.4
    csv_path = R.pf.save_psfs_csv(psfs, image_path=fake_image, timestamp_epoch=epoch)
    pkl_path = R.pf.save_psfs_pkl(psfs, image_path=fake_image, timestamp_epoch=epoch)
    text = open(csv_path, newline="").read().replace(d, "<DIR>")
    back = pickle.load(open(pkl_path, "rb"))                 # (a file this script wrote a moment ago)
    assert list(back.keys()) == list(psfs.keys())
    with open(os.path.join(GOLD, "io_f5_small_96_psfs.csv"), "w", newline="") as f:
        f.write(text)
    meta = {"image": "<DIR>/plate 1/f5_small_96.tif", "epoch": epoch,
            "csv_path": csv_path.replace(d, "<DIR>"), "pkl_path": pkl_path.replace(d, "<DIR>"),
            "n_psfs": len(psfs), "keys": [list(map(int, k)) for k in psfs.keys()],
            "filename_kat": [[p, e, sfx, R.pf._psfs_filename(p, e, sfx)] for p, e, sfx in
                             (("/data/run 7/a.tif", 1450000000.4, ".csv"), ("/x/y.png", 36.5, ".pkl"), ("/x/y.png", 1791084336.2, ".png"))]}
    json.dump(meta, open(os.path.join(GOLD, "io_f5_small_96.json"), "w"), indent=1)
    print("io fixtures:", meta["csv_path"], meta["n_psfs"], flush=True)


def loader_inputs(psfs_full):
    """The two PSF dicts of the easy_load_processed_image fixture (shared with tests/test_flexlibrary_loader.py, which rebuilds
    them from the committed field fixture): an older complete one, and a newer one with half the PSFs plus three crafted
    entries that exercise Spot.__init__'s bounds logic (flexlibrary.py:100-112)."""
    items = list(psfs_full.items())
    newer = dict(items[:len(items) // 2])
    sub, fit = np.zeros((5, 5), np.int64), np.zeros((5, 5))
    mk = lambda h0, w0: (np.float64(h0), np.float64(w0), np.float64(100.), np.float64(500.), np.float64(1.), np.float64(1.),  # noqa: E731
                         np.float64(0.), sub, fit, 1.0, np.float64(0.9), np.float64(5.))
    newer[(0, 50)] = mk(0.4, 50.2)          # square off the top edge, fitted centre too close to it: rejected
    newer[(1, 30)] = mk(2.2, 30.1)          # square off the top edge, fitted centre inside the margin: accepted
    newer[(40, 95)] = mk(40.3, 200.0)       # square off the right edge; the `and`/`or` precedence of :104-111 lets it pass
    return dict(items), newer


def gen_loader():
    """Experiment.easy_load_processed_image (flexlibrary.py:516-564) on files the reference's own save_psfs_pkl wrote: the
    latest `<image>*_psfs_*.pkl` wins, every PSF becomes a Spot of size fit_img.shape[0] unless Spot.__init__ raises."""
    import json
    import tempfile
    from PIL import Image as PILImage
    R = _ref()
    from refload import load_flexlibrary
    load_flexlibrary(R)
    R.fl.imread = lambda p: np.array(PILImage.open(p))
    spec = FIELDS["f5_small_96"]
    img = build_field(spec)
    older, newer = loader_inputs(R.pf.find_peptides(img))
    d = tempfile.mkdtemp()
    png = os.path.join(d, "f5_small_96.tif.png")
    PILImage.fromarray(img).save(png)
    R.pf.save_psfs_pkl(older, image_path=png, timestamp_epoch=1450000000.4)
    R.pf.save_psfs_pkl(newer, image_path=png, timestamp_epoch=1450000500.0)
    # (the reference opens the pickle in Python 2's text mode; under Python 3 it has to be binary)
    im, discarded = R.fl.Experiment.easy_load_processed_image(png)
    im_old, disc_old = R.fl.Experiment.easy_load_processed_image(png, psf_pkl_filepath=R.pf._psfs_filename(png, 1450000000.4, ".pkl"))
    im_none, disc_none = R.fl.Experiment.easy_load_processed_image(png, load_psfs=False)
    out = {"epochs": [1450000000.4, 1450000500.0],
           "latest": {"spots": [[int(s.h), int(s.w), int(s.size)] for s in im.spots], "discarded": int(discarded),
                      "fit_h0": [float(s.gaussian_fit[0]) for s in im.spots]},
           "older": {"spots": [[int(s.h), int(s.w), int(s.size)] for s in im_old.spots], "discarded": int(disc_old)},
           "no_psfs": {"spots": len(im_none.spots), "discarded": int(disc_none)},
           "image_shape": list(im.image.shape), "filepath_is_metadata": im.metadata == {"filepath": png}}
    json.dump(out, open(os.path.join(GOLD, "loader_f5_small_96.json"), "w"), indent=1)
    print("loader fixture:", len(out["latest"]["spots"]), "spots,", out["latest"]["discarded"], "discarded;", len(out["older"]["spots"]), "in the older file", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    if a.only in ("", "kat"):
        gen_kat()
    if a.only in ("", "reg"):
        gen_reg()
    if a.only in ("", "phot"):
        gen_photometry()
    if a.only in ("", "phot_wide"):
        gen_photometry_wide()
    if a.only in ("", "fields"):
        with mp.Pool(a.procs) as pool:
            gen_fields(pool)
    if a.only in ("", "track"):
        gen_tracking()
    if a.only in ("", "track_long"):
        gen_tracking(tracking_long_cases(), "tracking_long.npz")
    if a.only in ("", "centroid"):
        gen_centroid()
    if a.only in ("", "centroid_wide"):
        gen_centroid(centroid_wide_cases(), "centroid_tracking_wide.npz")
    if a.only in ("", "loader"):
        gen_loader()
    if a.only in ("", "io"):
        gen_io()
    if a.only in ("", "degen"):
        # degenerate frames through the unmodified reference (statuses 0 / 4 / -16, NaN r_2 passing pflib.py:466)
        with mp.Pool(a.procs) as pool:
            gen_fields(pool, with_stability=(), fields={k: {"image": v} for k, v in degenerate_images().items()},
                       prefix="degen_")
    if a.only in ("", "params"):
        with mp.Pool(a.procs) as pool:
            gen_fields(pool, with_stability=(), fields=parameter_cases(), prefix="params_")
    if a.only in ("", "tiny"):
        with mp.Pool(a.procs) as pool:
            gen_fields(pool, with_stability=(), fields=tiny_cases(), prefix="tiny_")
    if a.only in ("", "wide"):
        with mp.Pool(a.procs) as pool:
            gen_fields(pool, with_stability=(), fields={k: {"image": v} for k, v in wide_images().items()}, prefix="wide_")
    if a.only in ("", "textbook"):
        # the reference with MINPACK's qrsolv (x = numpy.diagonal(r).copy(), refload.load_reference(textbook_qrsolv=True)):
        # pins the oracle's / the GPU's FSQ_MODE_TEXTBOOK.  A fresh interpreter so that every worker loads that variant.
        if not TEXTBOOK:
            import subprocess
            env = dict(os.environ, FSQ_GOLDEN_TEXTBOOK="1")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--only", "textbook", "--procs", str(a.procs)], env=env)
        else:
            with mp.Pool(a.procs) as pool:
                gen_fields(pool, with_stability=(), prefix="textbook_",
                           fields={k: FIELDS[k] for k in ("f5_small_96", "f3_hard_256")})


if __name__ == "__main__":
    main()

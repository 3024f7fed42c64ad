"""CPU tests (-m "not gpu"): the oracle (oracle/fsq_oracle.c) against outputs of the reference itself.

tests/golden/*.npz were produced by oracle/gen_golden.py running the reference's own code
(pflib / gaussfitter / mpfit / phase_correlate) in the build container.  The bar is BIT equality for
candidates, fitted parameters, exit status, iteration / evaluation counts and the consolidated table."""
import os

import numpy as np
import pytest

import oracle as O
from _util import (DEGEN_NAMES, FIELD_NAMES, GOLD, PARAM_NAMES, TEXTBOOK_NAMES, TINY_NAMES, WIDE_NAMES, bits_equal, golden_params, load_field,
                   rois_of)


@pytest.fixture(scope="module", autouse=True)
def _build():
    O.build()


@pytest.mark.parametrize("name", FIELD_NAMES)
def test_candidates_exact(name):
    g, img = load_field(name)
    hw = O.candidates(img)
    assert np.array_equal(hw, g["candidates"])


@pytest.mark.parametrize("name", FIELD_NAMES)
@pytest.mark.parametrize("libm", [False, True])
def test_fits_bit_exact(name, libm):
    """Every LM solve: parameters, status, niter, nfev, fnorm identical to mpfit's (pflib.py:180-214)."""
    g, img = load_field(name)
    f = O.fit_rois(rois_of(img, g["candidates"]), mode=0, n_threads=os.cpu_count(), libm=libm)
    assert bits_equal(f["p"], g["params"]).all()
    assert np.array_equal(f["status"], g["status"])
    assert np.array_equal(f["niter"], g["niter"])
    assert np.array_equal(f["nfev"], g["nfev"])
    assert bits_equal(f["fnorm"], g["fnorm"]).all()


@pytest.mark.parametrize("name", FIELD_NAMES)
def test_find_peptides_table(name):
    """Full pflib.find_peptides (pflib.py:284-520): keys, 7 parameters, rmse, r_2, s_n, fit image."""
    g, img = load_field(name)
    rows, fits, keep, key = O.find_peptides(img, n_threads=os.cpu_count())
    assert np.array_equal(key, g["table_keys"])          # same keys in the same dict order
    r = rows[keep]
    got7 = np.stack([r[k] for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")], axis=1)
    assert bits_equal(got7, g["table7"]).all()
    gotm = np.stack([r["rmse"], r["r2"], r["s_n"]], axis=1)
    assert bits_equal(gotm, g["table_metrics"]).all()
    sub = np.stack([img[h - 2:h + 3, w - 2:w + 3] for h, w in zip(r["h"], r["w"])]).astype(np.int64)
    assert np.array_equal(sub, g["table_sub"])
    fit = np.stack([O.model(p) for p in fits["p"][keep]])
    assert bits_equal(fit, g["table_fit"]).all()


def _check_against_golden(g, img, mode, prm=None):
    """candidates, every LM solve and the consolidated table of one fixture, bit for bit."""
    prm = prm or {}
    det = dict(med_size=prm.get("median_filter_size", 5), K=prm.get("correlation_matrix", O.DEFAULT_K), c_std=prm.get("c_std", 2.0))
    fp = dict(det, r2_thr=prm.get("r_2_threshold", 0.7), radius=prm.get("consolidation_radius", 4))
    assert np.array_equal(O.candidates(img, **det), g["candidates"])
    f = O.fit_rois(rois_of(img, g["candidates"]), mode=mode, n_threads=os.cpu_count())
    assert bits_equal(f["p"], g["params"]).all()
    for k in ("status", "niter", "nfev"):
        assert np.array_equal(f[k], g[k]), k
    assert bits_equal(f["fnorm"], g["fnorm"]).all()
    if int(g["table_error"]):
        with pytest.raises(AssertionError):
            O.find_peptides(img, mode=mode, n_threads=os.cpu_count(), **fp)
        return
    rows, fits, keep, key = O.find_peptides(img, mode=mode, n_threads=os.cpu_count(), **fp)
    assert np.array_equal(key, g["table_keys"].reshape(-1, 2))
    r = rows[keep]
    got7 = np.stack([r[k] for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")], axis=1)
    assert bits_equal(got7, g["table7"].reshape(-1, 7)).all()
    gotm = np.stack([r["rmse"], r["r2"], r["s_n"]], axis=1)
    assert bits_equal(gotm, g["table_metrics"].reshape(-1, 3)).all()      # (bits_equal treats NaN == NaN)


@pytest.mark.parametrize("name", DEGEN_NAMES)
def test_degenerate_frames_match_reference(name):
    """Flat, saturated, dim, pure-noise, hot-pixel and all-zero frames through the unmodified reference
    (oracle/gen_golden.py --only degen): gtol exits (status 4, mpfit.py:1151), zero-variance ROIs whose r_2 is NaN
    and passes the filter (pflib.py:466), consolidation among NaN scores.  (mpfit's status-0 early returns,
    mpfit.py:956-964, cannot be reached through pflib: gaussfitter clips the start into the bounds, gaussfitter.py:202-204;
    none of these frames drives a step non-finite, status -16, mpfit.py:1330-1335.)"""
    g, img = load_field(name, prefix="degen_")
    _check_against_golden(g, img, mode=0)


@pytest.mark.parametrize("name", PARAM_NAMES)
def test_nondefault_keywords_match_reference(name):
    """find_peptides with other median windows (3, 4, 7, 9), correlation matrices (3 x 3, 7 x 7, 11 x 11), c_std, r_2 thresholds
    and consolidation radii (2, 3, 6, 9) through the unmodified reference (oracle/gen_golden.py --only params)."""
    g, img = load_field(name, prefix="params_")
    prm = golden_params(g)
    assert prm and int(g["table_error"]) == 0 and len(g["table_keys"]) > 5
    _check_against_golden(g, img, mode=0, prm=prm)


@pytest.mark.parametrize("name", TINY_NAMES)
def test_tiny_frames_match_reference(name):
    """5 x 5 ... 9 x 5 frames, some smaller than their median window / correlation matrix (scipy's 'reflect' index wraps more than
    once; the correlation sees mostly zero padding), through the unmodified reference (oracle/gen_golden.py --only tiny)."""
    g, img = load_field(name, prefix="tiny_")
    assert min(img.shape) <= 9 and len(g["candidates"]) >= 1 and len(g["table_keys"]) >= 1
    _check_against_golden(g, img, mode=0, prm=golden_params(g))


@pytest.mark.parametrize("name", WIDE_NAMES)
def test_frames_beyond_16_bits_match_reference(name):
    """uint32 frames with 20- to 28-bit pixel values through the unmodified reference (oracle/gen_golden.py --only wide;
    pflib works on image.astype(np.int64), pflib.py:241, 443): the vectors behind FSQ_PIXELS_U32."""
    g, img = load_field(name, prefix="wide_")
    assert img.dtype == np.uint32 and int(img.max()) > 65535
    _check_against_golden(g, img, mode=0)


def test_degenerate_goldens_cover_the_cases():
    st = np.concatenate([np.load(os.path.join(GOLD, "degen_%s.npz" % n))["status"] for n in DEGEN_NAMES])
    assert (st == 4).any() and (st == 2).any() and (st == 5).any()
    m = np.load(os.path.join(GOLD, "degen_d7_zero_24.npz"))["table_metrics"]
    assert np.isnan(m[:, 1]).any()                      # a kept peak whose r_2 is NaN


@pytest.mark.parametrize("name", TEXTBOOK_NAMES)
def test_textbook_mode_matches_patched_reference(name):
    """mode=1 == the reference run with MINPACK's qrsolv (the solution vector a COPY of R's diagonal,
    refload.load_reference(textbook_qrsolv=True)): pins FSQ_MODE_TEXTBOOK."""
    g, img = load_field(name, prefix="textbook_")
    _check_against_golden(g, img, mode=1)


def test_textbook_mode_differs_but_converges():
    """mode=1 (qrsolv diagonal copied, MINPACK behaviour) is a different algorithm: SURVEY fact 3."""
    g, img = load_field("f5_small_96")
    rois = rois_of(img, g["candidates"])
    a = O.fit_rois(rois, mode=0)
    b = O.fit_rois(rois, mode=1)
    assert (b["status"] > 0).all()
    assert not bits_equal(a["p"], b["p"]).all()


def test_kat():
    k = np.load(os.path.join(GOLD, "kat.npz"))
    import ctypes
    L = O.lib()
    # qrfac: docstring example mpfit.py:1711-1742 and a 25x7 pivoted case
    for tag, pivot in (("qrfac", 0), ("qrfac25", 1)):
        a = np.ascontiguousarray(k[tag + "_in"]).copy()
        m, n = a.shape
        ipvt = np.zeros(n, np.int32)
        rdiag = np.zeros(n)
        acnorm = np.zeros(n)
        L.fsq_o_qrfac(a.ctypes.data_as(ctypes.c_void_p), m, n, pivot, ipvt.ctypes.data_as(ctypes.c_void_p),
                      rdiag.ctypes.data_as(ctypes.c_void_p), acnorm.ctypes.data_as(ctypes.c_void_p))
        assert bits_equal(a, k[tag + "_a"]).all()
        assert np.array_equal(ipvt, k[tag + "_ipvt"])
        assert bits_equal(rdiag, k[tag + "_rdiag"]).all() and bits_equal(acnorm, k[tag + "_acnorm"]).all()
    np.testing.assert_allclose(k["qrfac_rdiag"], [-11.0, -7.48166], rtol=1e-5)      # the documented numbers
    # enorm == numpy.dot through OpenBLAS (contiguous 25 / 7, strided Jacobian columns)
    assert all(O.enorm(v) == e for v, e in zip(k["enorm25_in"], k["enorm25_out"]))
    assert all(O.enorm(v) == e for v, e in zip(k["enorm7_in"], k["enorm7_out"]))
    J = np.ascontiguousarray(k["enorm_col_in"])
    for j in range(8):
        for c in range(7):
            assert O.enorm(J[j:, :].ravel()[c:], inc=7) == k["enorm_col_out"][j, c]
    # illumina_s_n, model
    assert all(O.illumina_s_n(r) == e for r, e in zip(k["sn_in"], k["sn_out"]))
    for p, e in zip(k["model_in"], k["model_out"]):
        assert bits_equal(O.model(p), e).all()


def test_numpy_sum_model():
    """numpy.sum of float64 = 8192-element chunks, each pairwise-summed (feeds numpy.std, pflib.py:250)."""
    rng = np.random.default_rng(0)
    import ctypes
    for shp in ((1,), (7,), (8,), (9,), (127,), (129,), (4096,), (8192,), (8193,), (20000,), (97, 101),
                (512, 512), (384, 640), (100000,), (1024, 1024)):
        x = rng.normal(0, 1e6, shp) ** 2
        got = O.lib().fsq_o_numpy_sum(x.ctypes.data_as(ctypes.c_void_p), x.size)
        assert got == float(np.sum(x)), shp


def test_threshold_matches_numpy():
    """mean + c_std*std of the int64 response image, bit for bit (pflib.py:250)."""
    g, img = load_field("f3_hard_256")
    hw, cm, thr = O.candidates(img, return_cm=True)
    assert thr == float(np.mean(cm) + 2 * np.std(cm))
    import scipy.ndimage, scipy.signal
    im = img.astype(np.int64)
    mf = im - np.minimum(scipy.ndimage.median_filter(im, 5), im)
    ref_cm = np.maximum(scipy.signal.correlate(mf, O.DEFAULT_K, mode="same"), 0)
    assert np.array_equal(cm, ref_cm)


def test_registration_golden():
    """phase_correlate.py:11-134: shifts exact on the 1/uf grid; error/diffphase to FFT rounding level."""
    g = np.load(os.path.join(GOLD, "registration.npz"))
    for name in g["names"]:
        if name.startswith("cycle512") and name != "cycle512_1":
            continue        # plain-DFT oracle: one 512x512 pair is enough on CPU
        for uf in (1, 20, 100):
            r = O.phase_correlate(g["ref_" + name], g["reg_" + name], uf)
            e = g["out_%s_uf%d" % (name, uf)]
            assert r[0] == e[0] and r[1] == e[1], (name, uf, r, e)
            assert abs(r[2] - e[2]) < 1e-9 and abs(r[3] - e[3]) < 1e-9


def test_errors():
    img = np.zeros((32, 32), np.uint16)
    with pytest.raises(ValueError):
        O.candidates(img, K=np.ones((4, 4), np.int64))          # pflib.py:236-239
    with pytest.raises(ValueError):
        O.find_peptides(img, radius=1)                          # pflib.py:431-432


def test_photometry_matches_reference():
    """Spot.mexican_hat_photometry_metric / gaussian_volume_photometry_metric (flexlibrary.py:172-230) as recorded from
    the reference on its own peaks and on spots pushed against the image borders (tests/golden/photometry.npz)."""
    g = np.load(os.path.join(GOLD, "photometry.npz"))
    for name in g["names"]:
        name = str(name)
        _, img = load_field(name)
        hw = g["hw_" + name]
        assert np.array_equal(O.mexican_hat(img, hw), g["mexican_hat_b6_r9_" + name])
        assert np.array_equal(O.mexican_hat(img, hw, 2, 4), g["mexican_hat_b2_r4_" + name])
        assert np.array_equal(O.gaussian_volume(g["fit7_" + name]).view(np.uint64), g["gaussian_volume_" + name].view(np.uint64))


def test_photometry_with_large_windows_matches_reference():
    """Mexican-hat windows beyond 31 x 31 - (brim, radius) = (6, 16), (10, 40), (3, 150: larger than the image) - as recorded from
    the reference (tests/golden/photometry_wide.npz, oracle/gen_golden.py --only phot_wide)."""
    g = np.load(os.path.join(GOLD, "photometry_wide.npz"))
    _, img = load_field(str(g["name"]))
    for brim, radius in g["cases"]:
        exp = g["mexican_hat_b%d_r%d" % (brim, radius)]
        got = O.mexican_hat(img, g["hw"], int(brim), int(radius))
        assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.array_equal(got[~np.isnan(exp)], exp[~np.isnan(exp)]), (brim, radius)
    wide = img.astype(np.uint32) * int(g["pixel_scale"])            # pixel values beyond 16 bits (fsq_o_mexican_hat_u32)
    assert int(wide.max()) > 65535
    for brim, radius in ((6, 9), (10, 40)):
        assert np.array_equal(O.mexican_hat(wide, g["hw"], brim, radius), g["scaled_mexican_hat_b%d_r%d" % (brim, radius)])

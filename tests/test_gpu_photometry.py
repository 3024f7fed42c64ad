"""GPU parity tests (-m gpu) of the spot photometry through the C ABI: the reference's recorded
values (tests/golden/photometry.npz), the oracle on random spots incl. clipped windows and other brim/radius, and a
full-size table (every kept peak of 256 fields) checked through the oracle on a sample."""
import os

import numpy as np
import pytest

from _util import GOLD, load_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from fluorosequencingimageanalysis_amd import photometry, pflib, synth
    import oracle as O
    O.build()
    return photometry, pflib, synth, O


def test_golden(env):
    ph, pflib, synth, O = env
    g = np.load(os.path.join(GOLD, "photometry.npz"))
    for name in g["names"]:
        name = str(name)
        _, img = load_field(name)
        hw = g["hw_" + name]
        assert np.array_equal(ph.mexican_hat_photometry_metric(img, hw), g["mexican_hat_b6_r9_" + name])
        assert np.array_equal(ph.mexican_hat_photometry_metric(img, hw, brim_size=2, radius=4), g["mexican_hat_b2_r4_" + name])
        fits = [tuple(r) for r in g["fit7_" + name]]
        assert np.array_equal(ph.gaussian_volume_photometry_metric(fits).view(np.uint64), g["gaussian_volume_" + name].view(np.uint64))


def test_large_windows_equal_reference(env):
    """(brim, radius) = (6, 16), (10, 40), (3, 150) against the reference's recorded values (tests/golden/photometry_wide.npz): the
    any-radius kernel of round 4."""
    ph, pflib, synth, O = env
    g = np.load(os.path.join(GOLD, "photometry_wide.npz"))
    _, img = load_field(str(g["name"]))
    for brim, radius in g["cases"]:
        exp = g["mexican_hat_b%d_r%d" % (brim, radius)]
        got = ph.mexican_hat_photometry_metric(img, g["hw"], brim_size=int(brim), radius=int(radius))
        assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.array_equal(got[~np.isnan(exp)], exp[~np.isnan(exp)]), (brim, radius)


def test_random_spots_and_shapes_vs_oracle(env):
    ph, pflib, synth, O = env
    rng = np.random.default_rng(5)
    img = synth.make_field(3, (120, 200), 40)
    img[:8, :8] = 65535                                   # saturated corner: sums stay exact
    hw = np.stack([rng.integers(-3, 123, 400), rng.integers(-3, 203, 400)], axis=1)
    # (radius > 15: the any-radius kernel of round 4 - windows beyond 31 x 31 are re-read from memory instead of held in registers)
    for brim, radius in ((6, 9), (0, 3), (1, 1), (3, 15), (5, 4), (2, 0), (6, 16), (10, 40), (0, 150)):
        got = ph.mexican_hat_photometry_metric(img, hw, brim_size=brim, radius=radius)
        exp = O.mexican_hat(img, hw, brim, radius)
        assert np.array_equal(np.isnan(got), np.isnan(exp)), (brim, radius)               # empty brim -> nan (numpy.median([]))
        ok = ~np.isnan(exp)
        assert np.array_equal(got[ok].view(np.uint64), exp[ok].view(np.uint64)), (brim, radius)
    with pytest.raises(ValueError):
        ph.mexican_hat_photometry_metric(np.stack([img, img]), hw)
    assert len(ph.mexican_hat_photometry_metric(img, np.zeros((0, 2), int))) == 0


def test_stack_table(env):
    """(field, h, w) table over a stack: the kept peaks of every field, as Image.find_gaussian_psfs would build Spots
    (flexlibrary.py:426-455: h = int(round(h_0)), w = int(round(w_0)))."""
    ph, pflib, synth, O = env
    imgs = np.stack([synth.make_field(900 + i, (128, 128), 30) for i in range(6)])
    tables = pflib.find_peptides_batch(imgs)
    spots = np.array([(f, k[0], k[1]) for f, t in enumerate(tables) for k in t.keys()], dtype=np.int64)
    got = ph.mexican_hat_photometry_metric(imgs, spots)
    for f in range(6):
        m = spots[:, 0] == f
        assert np.array_equal(got[m], O.mexican_hat(imgs[f], spots[m][:, 1:]))

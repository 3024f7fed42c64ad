"""CPU tests of flexlibrary.Experiment.easy_load_processed_image (flexlibrary.py:516-564) against what the REFERENCE's own method
returned for the same files (tests/golden/loader_f5_small_96.json, oracle/gen_golden.py --only loader): the latest
`<image>*_psfs_*.pkl` wins, every PSF becomes a Spot of size fit_img.shape[0], Spot.__init__ failures are counted."""
import json
import os

import numpy as np
import pytest

from _util import GOLD, ROOT, load_field
from test_batch_io import golden_psfs


def _inputs():
    """The fixture's two PSF dicts, rebuilt exactly as oracle/gen_golden.py::loader_inputs builds them."""
    psfs, img = golden_psfs()
    items = list(psfs.items())
    newer = dict(items[:len(items) // 2])
    sub, fit = np.zeros((5, 5), np.int64), np.zeros((5, 5))

    def mk(h0, w0):
        return (np.float64(h0), np.float64(w0), np.float64(100.), np.float64(500.), np.float64(1.), np.float64(1.),
                np.float64(0.), sub, fit, 1.0, np.float64(0.9), np.float64(5.))
    newer[(0, 50)] = mk(0.4, 50.2)
    newer[(1, 30)] = mk(2.2, 30.1)
    newer[(40, 95)] = mk(40.3, 200.0)
    return dict(items), newer, img


def _files(tmp_path):
    from PIL import Image
    from fluorosequencingimageanalysis_amd import pflib
    g = json.load(open(os.path.join(GOLD, "loader_f5_small_96.json")))
    older, newer, img = _inputs()
    png = str(tmp_path / "f5_small_96.tif.png")
    Image.fromarray(img).save(png)
    p_old = pflib.save_psfs_pkl(older, image_path=png, timestamp_epoch=g["epochs"][0])
    p_new = pflib.save_psfs_pkl(newer, image_path=png, timestamp_epoch=g["epochs"][1])
    return g, png, p_old, p_new, img, older, newer


def test_easy_load_equals_reference(tmp_path, caplog):
    from fluorosequencingimageanalysis_amd.flexlibrary import Experiment
    g, png, p_old, p_new, img, older, newer = _files(tmp_path)
    assert sorted([p_old, p_new])[-1] == p_new
    im, discarded = Experiment.easy_load_processed_image(png)
    assert np.array_equal(im.image, img) and list(im.image.shape) == g["image_shape"] and im.metadata == {"filepath": png}
    assert [[s.h, s.w, s.size] for s in im.spots] == g["latest"]["spots"] and discarded == g["latest"]["discarded"] == 1
    assert [float(s.gaussian_fit[0]) for s in im.spots] == g["latest"]["fit_h0"]
    assert all(s.parent_Image is im for s in im.spots)
    # an explicit (older) pickle; no PSFs at all
    im2, d2 = Experiment.easy_load_processed_image(png, psf_pkl_filepath=p_old)
    assert [[s.h, s.w, s.size] for s in im2.spots] == g["older"]["spots"] and d2 == g["older"]["discarded"] == 0
    im3, d3 = Experiment.easy_load_processed_image(png, load_psfs=False)
    assert len(im3.spots) == g["no_psfs"]["spots"] == 0 and d3 == 0
    # the Spots carry the pickled tuples: the table read back equals the table saved
    back = {(s.h, s.w): s.gaussian_fit for s in im2.spots}
    assert list(back) == list(older)
    for k in older:
        assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(back[k], older[k]))


def test_easy_load_errors(tmp_path):
    from PIL import Image
    from fluorosequencingimageanalysis_amd.flexlibrary import Experiment
    png = str(tmp_path / "lonely.png")
    Image.fromarray(np.zeros((16, 16), np.uint16)).save(png)
    with pytest.raises(ValueError):                         # no pickle next to the image (flexlibrary.py:541-546)
        Experiment.easy_load_processed_image(png)
    im, d = Experiment.easy_load_processed_image(png, load_psfs=False)
    assert im.spots == [] and d == 0

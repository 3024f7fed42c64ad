"""GPU tests (-m gpu) of the batch / IO / CLI layer on the real compute path: the basic_image_script command line over a
directory of 16-bit TIFFs (SURVEY.md 8a image_batch row, 8f N2), the chunked find_peptides_batch, candidate counting."""
import os
import pickle
import warnings

import numpy as np
import pytest

from _util import bits_equal, load_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluorosequencingimageanalysis_amd import basic_image_script, pflib, synth
    import oracle as O
    O.build()
    return basic_image_script, pflib, synth, O


def _write_tif(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path, format="TIFF")


def _check_outputs(pflib, O, res_tuple, img, golden=None):
    conv, pkl, tab, png = res_tuple
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = pickle.load(open(pkl, "rb"))
    rows, fits, keep, key = O.find_peptides(img, n_threads=16)
    assert np.array_equal(np.array(list(d.keys()), dtype=np.int32).reshape(-1, 2), key)
    r = rows[keep]
    exp7 = np.stack([r[k] for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")], axis=1)
    got7 = np.array([[float(x) for x in v[:7]] for v in d.values()]).reshape(-1, 7)
    assert bits_equal(got7, exp7).all()
    if golden is not None:                              # the reference's own table for this image
        assert np.array_equal(key, golden["table_keys"]) and bits_equal(got7, golden["table7"]).all()
        assert np.array_equal(np.array([v[7] for v in d.values()]), golden["table_sub"])
        assert bits_equal(np.array([v[8] for v in d.values()]), golden["table_fit"]).all()
    lines = open(tab, newline="").read().split("\r\n")
    assert lines[0].split("\t") == pflib.CSV_HEADER and len(lines) == len(d) + 2
    for line, v in zip(lines[1:-1], d.values()):
        f = line.split("\t")
        assert f[0] == conv
        assert f[1:] == [pflib._py2_str(x) for x in v[:7]] + [pflib._py2_str(v[9]), pflib._py2_str(v[10]), pflib._py2_str(v[11])]


def test_cli_over_a_directory_of_tiffs(env, tmp_path):
    cli, pflib, synth, O = env
    g, f5 = load_field("f5_small_96")
    other = synth.make_field(77, (96, 96), 10)
    small = synth.make_field(78, (64, 80), 5)
    (tmp_path / "a" / "sub").mkdir(parents=True)
    _write_tif(str(tmp_path / "a" / "f5.tif"), f5)
    _write_tif(str(tmp_path / "a" / "sub" / "other.tif"), other)
    _write_tif(str(tmp_path / "a" / "small.tif"), small)
    (tmp_path / "a" / "broken.tif").write_bytes(b"II*\x00garbage")
    log = str(tmp_path / "log.txt")
    res = cli.main(["-L", log, str(tmp_path / "a")])
    assert sorted(res) == sorted(str(tmp_path / "a" / p) for p in ("f5.tif", "sub/other.tif", "small.tif"))
    _check_outputs(pflib, O, res[str(tmp_path / "a" / "f5.tif")], f5, golden=g)
    _check_outputs(pflib, O, res[str(tmp_path / "a" / "sub" / "other.tif")], other)
    _check_outputs(pflib, O, res[str(tmp_path / "a" / "small.tif")], small)
    assert "broken.tif" in open(log).read()
    # non-default parameters reach find_peptides
    res2 = cli.main(["--parameters", "{'c_std': 3, 'r_2_threshold': 0.9}", "-L", log, str(tmp_path / "a" / "sub")])
    conv, pkl, tab, png = res2[str(tmp_path / "a" / "sub" / "other.tif")]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = pickle.load(open(pkl, "rb"))
    rows, fits, keep, key = O.find_peptides(other, c_std=3.0, r2_thr=0.9, n_threads=16)
    assert np.array_equal(np.array(list(d.keys()), dtype=np.int32).reshape(-1, 2), key)


def test_chunked_batch_equals_one_pass(env, monkeypatch):
    """A stack larger than CHUNK_PIXELS goes through engine.StreamPipeline in chunks (the last one padded):
    same dicts as one pass."""
    cli, pflib, synth, O = env
    imgs = np.stack([synth.make_field(600 + i, (96, 96), 8 + i) for i in range(11)])
    one = pflib.find_peptides_batch(imgs)                   # (a small stack: one library call on this thread, pflib._small_pass)
    assert any(k[0] == "small" and k[2:5] == (11, 96, 96) for k in pflib._CACHE)
    monkeypatch.setattr(pflib, "CHUNK_PIXELS", 3 * 96 * 96)
    many = pflib.find_peptides_batch(imgs)                  # (chunks of 3 fields through the runner's lanes)
    assert any(k[0] == "batch" and k[2:5] == (3, 96, 96) for k in pflib._CACHE)
    assert len(one) == len(many) == 11
    for a, b in zip(one, many):
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for x, y in zip(a[k], b[k]))


def test_candidate_counts(env, tmp_path):
    cli, pflib, synth, O = env
    imgs = [synth.make_field(610 + i, shp, 9) for i, shp in enumerate([(96, 96), (64, 80), (96, 96)])]
    paths = []
    for i, im in enumerate(imgs):
        paths.append(str(tmp_path / ("c%d.tif" % i)))
        _write_tif(paths[-1], im)
    paths.insert(1, str(tmp_path / "missing.tif"))
    counts = pflib._candidate_counts(paths, {"c_std": 2})
    assert counts[1] is None
    assert [counts[0], counts[2], counts[3]] == [len(O.candidates(im)) for im in imgs]

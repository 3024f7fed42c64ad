"""CPU tests of the batch / IO / CLI layer around the hot path (SURVEY.md 8a rows image_batch, output formats; 8f N2):
pflib.read_image, save_psfs_csv / save_psfs_pkl / _psfs_filename against files the REFERENCE wrote
(tests/golden/io_f5_small_96*, oracle/gen_golden.py --only io), image_batch's path / grouping / swallow-and-log
semantics (pflib.py:940-996) and the basic_image_script command line (basic_image_script.py:36-124).
The GPU work is stood in for by the reference's recorded table where a test needs PSFs - the compute path itself is
covered by the -m gpu tests (tests/test_gpu_batch_io.py runs this layer on the real thing)."""
import json
import logging
import os
import pickle

import numpy as np
import pytest

from _util import GOLD, load_field


def golden_psfs(name="f5_small_96"):
    """The reference's find_peptides dict for a golden field, rebuilt from the fixture."""
    g, img = load_field(name)
    d = {}
    for k, t7, sub, fit, m in zip(g["table_keys"], g["table7"], g["table_sub"], g["table_fit"], g["table_metrics"]):
        d[(int(k[0]), int(k[1]))] = tuple(np.float64(x) for x in t7) + (sub, fit, float(m[0]), np.float64(m[1]), np.float64(m[2]))
    return d, img


def test_csv_equals_reference_file(tmp_path):
    """save_psfs_csv against the file the reference's save_psfs_csv wrote for the same PSFs (pflib.py:639-711):
    same name, header, path column, row order and line ends; every number equal to the 12 significant digits
    Python 2's str() printed (the fixture was written under Python 3, which prints 17)."""
    from fluorosequencingimageanalysis_amd import pflib
    meta = json.load(open(os.path.join(GOLD, "io_f5_small_96.json")))
    ref_text = open(os.path.join(GOLD, "io_f5_small_96_psfs.csv"), newline="").read().replace("<DIR>", str(tmp_path))
    psfs, _ = golden_psfs()
    assert [list(k) for k in psfs] == meta["keys"]
    image = meta["image"].replace("<DIR>", str(tmp_path))
    os.makedirs(os.path.dirname(image))
    path = pflib.save_psfs_csv(psfs, image_path=image, timestamp_epoch=meta["epoch"])
    assert path == meta["csv_path"].replace("<DIR>", str(tmp_path))
    got = open(path, newline="").read()
    assert got.count("\r\n") == ref_text.count("\r\n") == len(psfs) + 1            # excel-tab dialect line ends
    gl, rl = got.split("\r\n"), ref_text.split("\r\n")
    assert gl[0] == rl[0]
    for a, b in zip(gl[1:-1], rl[1:-1]):
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[0] == fb[0] == image and len(fa) == len(fb) == 11
        for x, y in zip(fa[1:], fb[1:]):
            assert x == pflib._py2_str(float(y))
            assert abs(float(x) - float(y)) <= 1e-11 * abs(float(y))
    assert pflib.save_psfs_pkl(psfs, image_path=image, timestamp_epoch=meta["epoch"]) == meta["pkl_path"].replace("<DIR>", str(tmp_path))
    for p, e, sfx, exp in meta["filename_kat"]:
        assert pflib._psfs_filename(p, e, sfx) == exp


def test_pkl_round_trip_and_py2_module_paths(tmp_path):
    """save_psfs_pkl: protocol 0 like cPickle.dump's default (pflib.py:635); what is read back equals the dict; the
    stream names numpy's globals by their numpy-1 paths (the reference's Python 2 reads these files,
    flexlibrary.py:541-547).  Interop with a real Python 2 cannot be run here: parity unpinned for that."""
    from fluorosequencingimageanalysis_amd import pflib
    psfs, _ = golden_psfs()
    p = pflib.save_psfs_pkl(psfs, output_path=str(tmp_path / "x.pkl"))
    raw = open(p, "rb").read()
    assert raw.startswith(b"(dp") and b"numpy._core" not in raw and b"numpy.core.multiarray" in raw
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        back = pickle.load(open(p, "rb"))
    assert list(back.keys()) == list(psfs.keys())
    for k in psfs:
        for a, b in zip(back[k], psfs[k]):
            assert np.array_equal(np.asarray(a), np.asarray(b)) and type(a) is type(b)
    with pytest.raises(ValueError):
        pflib.save_psfs_pkl(psfs)


def _write_tif(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path, format="TIFF")


def test_read_image_16bit(tmp_path):
    from PIL import Image
    from fluorosequencingimageanalysis_amd import pflib
    _, img = load_field("f5_small_96")
    _write_tif(str(tmp_path / "a.tif"), img)
    # a non-PNG image without a PNG sibling is converted to '<image>.png' first (pflib.py:737-745, convert_image)
    conv, arr = pflib.read_image(str(tmp_path / "a.tif"))
    assert conv == str(tmp_path / "a.tif.png") and os.path.exists(conv)
    assert arr.dtype == np.uint16 and np.array_equal(arr, img)                    # 16-bit values survive the conversion
    # an existing '<image>.png' is what the reference reads instead (pflib.py:737-739)
    Image.fromarray((img // 2).astype(np.uint16)).save(str(tmp_path / "a.tif.png"))
    conv, arr = pflib.read_image(str(tmp_path / "a.tif"))
    assert conv == str(tmp_path / "a.tif.png") and np.array_equal(arr, img // 2)
    conv, arr = pflib.read_image(conv)                                            # a PNG is read as it is
    assert conv == str(tmp_path / "a.tif.png") and np.array_equal(arr, img // 2)
    with pytest.raises(Exception):
        pflib.read_image(str(tmp_path / "missing.tif"))
    assert pflib.convert_image(str(tmp_path / "missing.tif")) is None             # logged, None (pflib.py:87-90)
    out = pflib.convert_image(str(tmp_path / "a.tif"), output_path=str(tmp_path / "b.png"))
    assert out == str(tmp_path / "b.png") and np.array_equal(np.array(Image.open(out)), img)


def records_of(psfs):
    """A find_peptides dict -> the peak records (uint8[k, engine.PEAK_RECORD_BYTES]) whose records_to_dicts is that dict."""
    from fluorosequencingimageanalysis_amd import engine
    rec = np.zeros(len(psfs), engine.RECORD_DTYPE)
    for r, ((kh, kw), v) in zip(rec, psfs.items()):
        for name, x in zip(("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta"), v[:7]):
            r[name] = x
        r["sub"], r["fit"], r["rmse"], r["r2"], r["s_n"] = v[7], v[8], v[9], v[10], v[11]
        r["key_h"], r["key_w"], r["h"], r["w"] = kh, kw, kh, kw
    return rec.view(np.uint8).reshape(len(psfs), engine.PEAK_RECORD_BYTES)


@pytest.fixture
def fake_gpu(monkeypatch):
    """The GPU call of image_batch (find_peptides_records: byte tables, no dicts) stood in for: every field 'finds' the golden
    PSFs (records what it was called with); a field whose first pixel is 7 'fails' the re-key assertion (count -1)."""
    from fluorosequencingimageanalysis_amd import _native, pflib
    psfs, _ = golden_psfs()
    one = records_of(psfs)
    back = pflib.records_to_dicts(one, [len(one)])[0]
    assert list(back) == list(psfs) and all(np.array_equal(np.asarray(a), np.asarray(b)) and type(a) is type(b)
                                            for k in psfs for a, b in zip(back[k], psfs[k]))
    calls = []

    def fake(images, **kw):
        calls.append((np.asarray(images).shape, kw))
        if kw.get("fit_type", "gauss") != "gauss":
            raise NotImplementedError("monte_carlo")
        counts = np.array([-1 if int(im[0, 0]) == 7 else len(one) for im in images], np.int32)
        return np.concatenate([one for c in counts if c >= 0] or [one[:0]]), counts, _native.PIXELS_U16
    monkeypatch.setattr(pflib, "find_peptides_records", fake)
    return calls


def test_image_batch_semantics(tmp_path, fake_gpu, monkeypatch, caplog):
    """pflib.image_batch (pflib.py:940-996): absolute de-duplicated keys, output names derived from the CONVERTED path,
    unreadable / failing images logged and skipped, same-shaped images in one GPU call."""
    from PIL import Image
    from fluorosequencingimageanalysis_amd import pflib
    _, img = load_field("f5_small_96")
    d = tmp_path / "run"
    d.mkdir()
    _write_tif(str(d / "a.tif"), img)
    _write_tif(str(d / "b.tif"), img)
    _write_tif(str(d / "c.tif"), img[:64, :80].copy())
    bad = img.copy()
    bad[0, 0] = 7                                       # the stand-in raises the re-key assertion for this one
    _write_tif(str(d / "d.tif"), bad)
    (d / "corrupt.tif").write_bytes(b"not an image")
    Image.fromarray(img).save(str(d / "b.tif.png"))     # b has a converted sibling
    monkeypatch.chdir(d)
    with caplog.at_level(logging.ERROR):
        res = pflib.image_batch(["a.tif", str(d / "a.tif"), "b.tif", "c.tif", "d.tif", "corrupt.tif", "nope.tif"],
                                find_peptides_parameters={"c_std": 3}, timestamp_epoch=1450000000.4)
    assert sorted(res) == [str(d / "a.tif"), str(d / "b.tif"), str(d / "c.tif")]
    conv, pkl, tab, png = res[str(d / "b.tif")]
    assert conv == str(d / "b.tif.png") and pkl == conv + "_psfs_nzaj5s.pkl" and tab == conv + "_psfs_nzaj5s.csv" and png == conv + "_psfs_nzaj5s.png"
    assert res[str(d / "a.tif")][0] == str(d / "a.tif.png")                      # converted on the way, like the reference
    assert res[str(d / "a.tif")][1] == str(d / "a.tif.png") + "_psfs_nzaj5s.pkl"
    for v in res.values():
        assert os.path.exists(v[1]) and os.path.exists(v[2]) and os.path.exists(v[3])
        assert open(v[2]).read().splitlines()[1].split("\t")[0] == v[0]          # 'Absolute image path' = converted path
    # the overlay (pflib.py:783-880): the image stretched into 8 bits, a light-blue 9 x 9 outline around every peak
    import pickle
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        peaks = pickle.load(open(pkl, "rb"))
    over = np.array(Image.open(png))
    grey = pflib._intensity_scaling(img)
    assert over.shape == img.shape + (3,) and len(peaks) > 3
    outline = np.zeros(img.shape, bool)
    for (h, w) in peaks:
        for hh, ww in [(h - 4, x) for x in range(w - 4, w + 5)] + [(h + 4, x) for x in range(w - 4, w + 5)] + \
                      [(y, w - 4) for y in range(h - 4, h + 5)] + [(y, w + 4) for y in range(h - 4, h + 5)]:
            if 0 <= hh < img.shape[0] and 0 <= ww < img.shape[1]:
                outline[hh, ww] = True
    assert (over[outline] == (173, 216, 230)).all()
    assert (over[~outline] == grey[~outline][:, None]).all()
    shapes = sorted(c[0] for c in fake_gpu)
    assert shapes == [(1, 64, 80), (3, 96, 96)] and all(c[1] == {"c_std": 3} for c in fake_gpu)
    # corrupt and missing: logged by convert_image and again by image_batch (as in the reference); re-key: once
    assert len([r for r in caplog.records if r.levelno >= logging.ERROR]) == 5


def test_parallel_image_batch_validates_num_processes(fake_gpu, tmp_path):
    from fluorosequencingimageanalysis_amd import pflib
    _, img = load_field("f5_small_96")
    _write_tif(str(tmp_path / "a.tif"), img)
    with pytest.raises(ValueError):
        pflib.parallel_image_batch([str(tmp_path / "a.tif")], num_processes=0)           # pflib.py:1060-1061
    with pytest.raises(ValueError):
        pflib.parallel_image_batch([str(tmp_path / "a.tif")], num_processes=2.5)
    res = pflib.parallel_image_batch([str(tmp_path / "a.tif")], num_processes=4, timestamp_epoch=36)
    assert list(res) == [str(tmp_path / "a.tif")] and res[str(tmp_path / "a.tif")][1].endswith("_psfs_10.pkl")


def test_basic_image_script_cli(tmp_path, fake_gpu, monkeypatch):
    """basic_image_script.py:84-124: --parameters through ast.literal_eval, -mc fills fit_type / N_iter without
    overriding, *.tif files of every directory tree, the log file, the call into parallel_image_batch."""
    from fluorosequencingimageanalysis_amd import basic_image_script as cli, pflib
    _, img = load_field("f5_small_96")
    (tmp_path / "p1" / "deep").mkdir(parents=True)
    (tmp_path / "p2").mkdir()
    _write_tif(str(tmp_path / "p1" / "x.tif"), img)
    _write_tif(str(tmp_path / "p1" / "deep" / "y.tif"), img)
    _write_tif(str(tmp_path / "p2" / "z.tif"), img)
    (tmp_path / "p2" / "skip.tiff").write_bytes(b"")
    (tmp_path / "p2" / "note.txt").write_text("x")
    seen = {}
    real = pflib.parallel_image_batch

    def spy(paths, find_peptides_parameters=None, timestamp_epoch=None, num_processes=None):
        seen.update(paths=list(paths), fp=find_peptides_parameters, n=num_processes, t=timestamp_epoch)
        return real(paths, find_peptides_parameters, timestamp_epoch, num_processes)
    monkeypatch.setattr(pflib, "parallel_image_batch", spy)
    log = str(tmp_path / "run.log")
    out = cli.main(["--parameters", "{'median_filter_size': 7, 'c_std': 3}", "-n", "3", "-L", log,
                    str(tmp_path / "p1"), str(tmp_path / "p2")])
    assert sorted(seen["paths"]) == sorted(str(tmp_path / p) for p in ("p1/x.tif", "p1/deep/y.tif", "p2/z.tif"))
    assert seen["fp"] == {"median_filter_size": 7, "c_std": 3} and seen["n"] == 3 and isinstance(seen["t"], float)
    assert sorted(out) == sorted(seen["paths"])
    text = open(log).read()
    assert "basic_image_script starting at" in text and "Will process target images" in text and "x.tif" in text
    # -mc: fit_type / N_iter only where --parameters did not set them; the GPU path refuses monte_carlo per image
    out = cli.main(["-mc", "--N_iter", "50", "--parameters", "{'N_iter': 9}", "-L", log, str(tmp_path / "p2")])
    assert seen["fp"] == {"N_iter": 9, "fit_type": "monte_carlo"} and out == {}
    with pytest.raises(SystemExit):
        cli.main(["-L", log])                           # at least one directory
    with pytest.raises((ValueError, SyntaxError)):
        cli.main(["--parameters", "{'c_std': ", "-L", log, str(tmp_path / "p2")])

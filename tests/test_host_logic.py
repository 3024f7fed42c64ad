"""CPU tests of the host-side logic and of the C ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from _util import GOLD, ROOT


def test_library_exports_every_declared_symbol():
    """libfsq_hip.so loads and exports each function include/fsq.h declares."""
    from fluorosequencingimageanalysis_amd import _native
    hdr = open(os.path.join(ROOT, "include", "fsq.h")).read()
    declared = set(re.findall(r"\b(fsq_[a-z_0-9]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_total")}
    assert declared == set(_native.EXPORTED), declared ^ set(_native.EXPORTED)
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    L = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _native.lib().fsq_version().startswith(b"fsq-hip")


def test_row_struct_layout():
    from fluorosequencingimageanalysis_amd import _native
    assert _native.ROW_DTYPE.itemsize == 128
    assert _native.ROW_DTYPE.fields["h"][1] == 96 and _native.ROW_DTYPE.fields["key_w"][1] == 124


def test_epoch_hash_kat():
    from fluorosequencingimageanalysis_amd import pflib
    k = np.load(os.path.join(GOLD, "kat.npz"))
    for e, h, back in zip(k["epochs"], k["epoch_hashes"], k["hash_epochs"]):
        assert pflib._epoch_to_hash(float(e)) == str(h)
        assert pflib._hash_to_epoch(str(h)) == int(back)
    with pytest.raises(ValueError):
        pflib._epoch_to_hash(0)
    with pytest.raises(ValueError):
        pflib._hash_to_epoch("ab!c")
    assert pflib._psfs_filename("/x/y.tif", 36, ".pkl") == "/x/y.tif_psfs_10.pkl"


def test_py2_str_and_csv(tmp_path):
    from fluorosequencingimageanalysis_amd import pflib
    assert pflib._py2_str(np.float64(2.0)) == "2.0"
    assert pflib._py2_str(0.1 + 0.2) == "0.3"
    assert pflib._py2_str(1234567.8901234567) == "1234567.89012"
    assert pflib._py2_str(1e20) == "1e+20"
    psfs = {(3, 4): (3.25, 4.5, 100.0, 2000.0, 1.1, 1.2, 0.0, np.zeros((5, 5), np.int64), np.zeros((5, 5)), 5.0, 0.9, 12.5)}
    p = pflib.save_psfs_csv(psfs, output_path=str(tmp_path / "a.csv"), image_path="img.tif")
    lines = open(p).read().splitlines()
    assert lines[0].split("\t") == pflib.CSV_HEADER
    assert lines[1].split("\t")[1:] == ["3.25", "4.5", "100.0", "2000.0", "1.1", "1.2", "0.0", "5.0", "0.9", "12.5"]
    q = pflib.save_psfs_pkl(psfs, output_path=str(tmp_path / "a.pkl"))
    import pickle
    assert list(pickle.load(open(q, "rb")).keys()) == [(3, 4)]
    with pytest.raises(ValueError):
        pflib.save_psfs_csv(psfs)


def test_parameter_validation_without_gpu():
    from fluorosequencingimageanalysis_amd import engine, pflib
    with pytest.raises(ValueError):
        engine.detect_params(5, np.ones((4, 4)), 2)
    with pytest.raises(ValueError):
        engine.detect_params(5, np.ones((3, 5)), 2)
    p = engine.detect_params(5, pflib.default_correlation_matrix, 2)
    assert p.ksz == 5 and p.K[12] == 30742 and p.K[0] == -5935
    with pytest.raises(ValueError):
        pflib.find_peptides(np.zeros((16, 16), np.uint16), consolidation_radius=1)
    # floating-point pixels are truncated toward zero like the reference's image.astype(np.int64) (pflib.py:241, 443)
    got = engine.as_u16_fields(np.array([[0.0, 1.9, 65535.99], [2.5, 3.0, 100.2]]))
    assert got.dtype == np.uint16 and got.tolist() == [[0, 1, 65535], [2, 3, 100]]
    assert engine.as_u16_fields(np.array([[7, 9]], np.int32)).tolist() == [[7, 9]]
    for bad in (np.array([[70000]]), np.array([[-1]]), np.array([[65536.0]]), np.array([[-0.5 - 1]]), np.array([[np.nan]]),
                np.array([[np.inf]]), np.array([[1 + 2j]])):
        with pytest.raises(NotImplementedError):
            engine.as_u16_fields(bad)
    assert pflib.illumina_s_n(np.arange(25).reshape(5, 5)) == (24 - np.mean([0, 1, 2, 3, 4, 20, 21, 22, 23, 24, 5, 9, 10, 14, 15, 19])) / np.std([0, 1, 2, 3, 4, 20, 21, 22, 23, 24, 5, 9, 10, 14, 15, 19])
    with pytest.raises(ValueError):
        pflib.illumina_s_n(np.zeros((4, 5)))
    # an image without a single 5x5 neighbourhood has no candidates (the reference's loop over range(2, H - 2), pflib.py:252,
    # is empty): empty results, no GPU involved
    for shape in ((1, 1), (4, 100), (100, 4), (2, 7)):
        tiny = np.ones(shape, np.uint16)
        assert pflib.find_peptides(tiny) == {} and pflib._psf_candidates(tiny) == []
        assert pflib.find_peptides_batch(np.stack([tiny] * 3)) == [{}, {}, {}]
        assert pflib.count_candidates(np.stack([tiny] * 2)).tolist() == [0, 0]
        rec, counts, _ = pflib.find_peptides_records(np.stack([tiny] * 2))
        assert rec.shape == (0, engine.PEAK_RECORD_BYTES) and counts.tolist() == [0, 0] and pflib.records_to_dicts(rec, counts) == [{}, {}]


def test_product_does_not_import_oracle():
    """the product path must not route through oracle/ (only tests/, smoke() and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "fluorosequencingimageanalysis_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "fsq_oracle" not in src and "libfsq_oracle" not in src, f


def test_c_dict_builder_equals_python_builder():
    """csrc/fsq_pyhost.c (peak records -> the reference's dicts) against the Python builder on synthetic records: keys, order,
    value types (numpy.float64 scalars, a Python float for rmse, int64 / float64 5x5 views of one block per field), NaN metrics,
    empty fields, failed fields, float16 pixel words, the 428-byte records of uint32 pixels."""
    from fluorosequencingimageanalysis_amd import _native as N, engine as E, pflib
    if pflib._fsq_pyhost is None:           # (a fresh tree: build it like the library)
        import importlib
        import __graft_entry__ as ge
        ge.build()
        pflib._fsq_pyhost = importlib.import_module("fluorosequencingimageanalysis_amd._fsq_pyhost")
    rng = np.random.default_rng(5)
    n = 5000
    for fmt in (N.PIXELS_U16, N.PIXELS_F16, N.PIXELS_U32):
        rec = rng.integers(0, 255, (n, E.peak_record_bytes(fmt)), dtype=np.uint8)
        v = E.peak_record_view(rec, fmt)
        assert v.dtype.itemsize == (428 if fmt == N.PIXELS_U32 else 378)
        v["key_h"] = rng.integers(0, 512, n)
        v["key_w"] = np.arange(n)
        for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta", "rmse", "r2", "s_n"):
            v[k] = rng.normal(0, 100, n)
        v["r2"][::7] = np.nan
        v["fit"] = rng.normal(0, 1, (n, 5, 5))
        v["sub"] = rng.integers(0, {N.PIXELS_F16: 0x7c00, N.PIXELS_U16: 65536, N.PIXELS_U32: 2 ** 31}[fmt], (n, 5, 5))
        counts = rng.multinomial(n, np.ones(40) / 40)
        counts[3] = 0
        offs = np.concatenate([[0], np.cumsum(counts)])
        failed = {5, 39}
        a = pflib._records_to_dicts(v, None, None, offs, failed, fmt)
        b = pflib._records_to_dicts_py(v, None, None, offs, failed, fmt)
        assert len(a) == len(b) == 40 and a[3] == {} and isinstance(a[5], AssertionError) and isinstance(a[39], AssertionError)
        assert str(a[5]) == str(b[5])
        for x, y in zip(a, b):
            if isinstance(x, AssertionError):
                continue
            assert list(x) == list(y) and all(type(k[0]) is int and type(k[1]) is int for k in x)
            for p, q in zip(x.values(), y.values()):
                assert type(p) is tuple and len(p) == 12
                for i in (0, 1, 2, 3, 4, 5, 6, 9, 10, 11):
                    assert type(p[i]) is type(q[i]) and (p[i] == q[i] or (p[i] != p[i] and q[i] != q[i]))
                for i in (7, 8):
                    assert p[i].dtype == q[i].dtype and p[i].shape == (5, 5) and p[i].flags.writeable and p[i].flags.c_contiguous
                    assert not p[i].flags.owndata and p[i].base is not None and np.array_equal(p[i], q[i])


def test_contrast_filters_and_overlay_arguments(tmp_path):
    """pflib._intensity_scaling / _histogram_equalization / save_psfs_png (reference pflib.py:749-880; scikit-image is absent
    here, so these follow its published arithmetic - parity unpinned): 8-bit results, truncating casts, monotone equalisation,
    per-peak colours, the file name rule and the square_size check."""
    from PIL import Image
    from fluorosequencingimageanalysis_amd import pflib
    img = (np.arange(64 * 48).reshape(64, 48) * 13 % 5000 + 100).astype(np.uint16)
    s = pflib._intensity_scaling(img)
    assert s.dtype == np.uint8 and s.min() == 0 and s.max() == 255
    assert np.array_equal(s, ((img.astype(np.float64) - img.min()) / float(img.max() - img.min()) * 255.0).astype(np.uint16).astype(np.uint8))
    assert (pflib._intensity_scaling(np.full((4, 4), 7, np.uint16)) == 0).all()
    e = pflib._histogram_equalization(img)
    assert e.dtype == np.uint8 and e.max() == 255
    order = np.argsort(img.ravel(), kind="stable")
    assert (np.diff(e.ravel()[order].astype(int)) >= 0).all()                      # monotone in the pixel value
    p = str(tmp_path / "x.png")
    Image.fromarray(img).save(p)
    psfs = {(10, 12): None, (30, 20): None}
    out = pflib.save_psfs_png(psfs, p, timestamp_epoch=1450000000.4, square_size=5, square_color="red",
                              square_colors={(30, 20): "lime"}, contrast_filter=pflib._histogram_equalization)
    assert out == p + "_psfs_nzaj5s.png"
    over = np.array(Image.open(out))
    assert tuple(over[8, 10]) == (255, 0, 0) and tuple(over[32, 22]) == (0, 255, 0) and tuple(over[10, 12]) == (e[10, 12],) * 3
    assert pflib.save_psfs_png({}, p, output_path=str(tmp_path / "y.png")) == str(tmp_path / "y.png")
    with pytest.raises(ValueError):
        pflib.save_psfs_png(psfs, p, square_size=4)


def test_bench_launcher_refuses_a_line_of_another_job_size(monkeypatch, capsys):
    """`python bench.py --gpus N` without a torchrun environment starts N child ranks itself and never
    relays a line whose n_gpus / ranks differ from the request; a failing job gives a non-zero exit code."""
    import argparse
    import json
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake(n_gpus, ranks=None, rc=0, extra=""):
        def run(cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            line = json.dumps({"metric": "psf_lm_fits_per_sec", "n_gpus": n_gpus, "ranks": n_gpus if ranks is None else ranks})
            return subprocess.CompletedProcess(cmd, rc, stdout=extra + line + "\n")
        return run

    a = argparse.Namespace(gpus=4)
    assert bench.launch_ranks(a, ["--gpus", "4", "--steps", "2"], run=fake(4)) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 4
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert bench.launch_ranks(a, [], run=fake(1)) == 1                 # the silent one-rank fallback of round 3
    assert bench.launch_ranks(a, [], run=fake(4, ranks=2)) == 1
    assert bench.launch_ranks(a, [], run=fake(4, rc=3)) == 3
    assert bench.launch_ranks(a, [], run=fake(4, extra="{\"stray\": 1}\n")) == 1
    assert capsys.readouterr().out == ""                               # nothing relayed in any of the refused cases
    with pytest.raises(ValueError):
        bench.check_rank0_line('{"n_gpus": 1}', 2)


def test_bench_with_mismatched_world_size_exits_nonzero():
    """WORLD_SIZE = 1 in the environment and --gpus 2: the bench leaves with an error instead of printing a one-GPU line."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fields", "2", "--size", "32", "--spots", "2",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


class _FakeEvent:
    def __init__(self, *a, **k):
        pass

    def record(self, *a):
        pass

    def synchronize(self):
        pass


class _FakeStream(_FakeEvent):
    cuda_stream = 0


@pytest.fixture
def cpu_batch_runner(monkeypatch):
    """pflib._BatchRunner with the device taken out: real torch CPU tensors for its staging / landing buffers, no-op streams
    and events, and a stand-in for engine.PathRunner / StreamPipeline supplied by the test.  Only the host-side choreography
    (stager, lanes, worker, clean-up) runs - which is what the tests below are about."""
    import contextlib
    import torch
    from fluorosequencingimageanalysis_amd import engine, pflib
    monkeypatch.setattr(engine, "_torch", lambda: torch)
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    monkeypatch.setattr(torch, "device", lambda *a, **k: "cpu")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    monkeypatch.setattr(torch.cuda, "Stream", _FakeStream)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    return pflib


def test_batch_runner_raises_promptly_when_the_gpu_side_fails(cpu_batch_runner, monkeypatch):
    """a failure of the GPU side with five or more chunks still unstaged used to leave the stager
    waiting for a staging buffer nobody would ever release, and the call spinning in its clean-up for good (holding the
    runner's lock).  Every lane fails on its second chunk of a 12-chunk stack: the call must raise that error within
    seconds, leave no stager thread behind, restore the interpreter's switch interval and be usable again."""
    import sys
    import threading
    import time
    import torch
    pflib = cpu_batch_runner
    from fluorosequencingimageanalysis_amd import engine
    calls = []

    class Lane:
        def __init__(self, *a, **k):
            self.fail = True

        def run(self, d, prm, *a):
            calls.append(int(d.shape[0]))
            if self.fail and len(calls) > 3:
                raise MemoryError("stand-in: a batch does not fit the fit queue")
            m = int(d.shape[0])
            z = torch.zeros(m + 1, dtype=torch.int32)
            return torch.zeros((0, engine.PEAK_RECORD_BYTES), dtype=torch.uint8), z, z.clone(), 0

    monkeypatch.setattr(engine, "PathRunner", Lane)
    runner = pflib._BatchRunner(1, 8, 8)
    words = np.zeros((12, 8, 8), np.uint16)
    interval = sys.getswitchinterval()
    before = threading.active_count()
    t0 = time.time()
    with pytest.raises(MemoryError):
        runner.run(words, 0, None, 0.7, 4)
    assert time.time() - t0 < 10.0
    assert abs(sys.getswitchinterval() - interval) < 1e-9
    time.sleep(0.2)
    assert threading.active_count() <= before
    assert runner.lock.acquire(timeout=1.0)             # the lock is free again (an evicting thread would not hang)
    runner.lock.release()
    for e in runner.lane_engines:
        e.fail = False
    assert runner.run(words, 0, None, 0.7, 4) == [{} for _ in range(12)]


def test_batch_runner_pipeline_failure_does_not_hang(cpu_batch_runner, monkeypatch):
    """The same for the continuous-batching path (raw=True, what find_peptides_sharded uses): the pipeline raises after two
    of twelve chunks."""
    import time
    pflib = cpu_batch_runner
    from fluorosequencingimageanalysis_amd import engine

    class Pipe:
        def __init__(self, *a, **k):
            pass

        def run(self, jobs, on_done, *a):
            for c, _ in enumerate(jobs):
                if c == 2:
                    raise RuntimeError("stand-in: FSQ_EINTERNAL")

        def close(self):
            pass

    monkeypatch.setattr(engine, "StreamPipeline", Pipe)
    runner = pflib._BatchRunner(1, 8, 8)
    t0 = time.time()
    with pytest.raises(RuntimeError, match="FSQ_EINTERNAL"):
        runner.run(np.zeros((12, 8, 8), np.uint16), 0, None, 0.7, 4, raw=True)
    assert time.time() - t0 < 10.0


def test_switch_interval_is_restored_after_overlapping_runs():
    """two overlapping runs must leave the interpreter's switch interval as they found it."""
    import sys
    from fluorosequencingimageanalysis_amd import pflib
    before = sys.getswitchinterval()
    pflib._switch_interval_acquire()
    pflib._switch_interval_acquire()
    assert abs(sys.getswitchinterval() - pflib.BATCH_SWITCH_INTERVAL) < 1e-9
    pflib._switch_interval_release()
    assert abs(sys.getswitchinterval() - pflib.BATCH_SWITCH_INTERVAL) < 1e-9
    pflib._switch_interval_release()
    assert abs(sys.getswitchinterval() - before) < 1e-9

"""GPU parity tests (-m gpu) of the registration path against the reference's own outputs
(tests/golden/registration.npz, produced by the reference's phase_correlate on numpy.fft).
Shifts must agree exactly on the 1/upsample grid; error / diffphase to FFT rounding level (1e-9 abs,
different FFT factorizations round differently - SURVEY.md 8c)."""
import os

import numpy as np
import pytest

from _util import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pc():
    import torch
    assert torch.cuda.is_available()
    from fluorosequencingimageanalysis_amd import phase_correlate as m
    return m


def test_golden_pairs(pc):
    g = np.load(os.path.join(GOLD, "registration.npz"))
    for name in g["names"]:
        a, b = g["ref_" + name], g["reg_" + name]
        for uf in (1, 20, 100):
            r = pc.phase_correlate(a, b, upsample_factor=uf)
            e = g["out_%s_uf%d" % (name, uf)]
            assert float(r[0]) == e[0] and float(r[1]) == e[1], (name, uf, r, e)
            assert abs(float(r[2]) - e[2]) < 1e-9 and abs(float(r[3]) - e[3]) < 1e-9, (name, uf, r, e)
    r = pc.phase_correlate(g["ref_odd0_63x65"], g["reg_odd0_63x65"], 1)
    assert isinstance(r[0], np.int64) and isinstance(r[2], np.float64)


def test_batch_equals_single_and_known_shift(pc):
    from fluorosequencingimageanalysis_amd import synth
    frames, off = synth.make_cycle_stack(9, n_cycles=4, shape=(256, 256), n_spots=200)
    out = pc.phase_correlate_batch(frames[:-1].astype(np.float64), frames[1:].astype(np.float64), 20)
    for k in range(3):
        single = pc.phase_correlate(frames[k], frames[k + 1], 20)
        assert float(single[0]) == out[k][0] and float(single[1]) == out[k][1]
        assert abs(float(single[2]) - out[k][2]) < 1e-12 and abs(float(single[3]) - out[k][3]) < 1e-12
        true = off[k + 1] - off[k]          # content moved by +true  => reported shift is -true
        assert abs(out[k][0] + true[0]) < 0.15 and abs(out[k][1] + true[1]) < 0.15


def test_errors(pc):
    with pytest.raises(ValueError):
        pc.phase_correlate(np.zeros((8, 8)), np.zeros((8, 9)))
    with pytest.raises(ValueError):
        pc.phase_correlate(np.zeros((2, 8, 8)), np.zeros((2, 8, 8)))


def test_u16_input_equals_f64_and_full_spectrum_path(pc, monkeypatch):
    """uint16 frames transformed as they sit in HBM give what the float64 copies give, bit for bit (the conversion is
    exact); the older complex-to-complex path (FSQ_REGISTER_Z2Z=1) agrees on the shifts exactly and on error / diffphase to
    rounding level."""
    from fluorosequencingimageanalysis_amd import synth
    frames, off = synth.make_cycle_stack(12, n_cycles=5, shape=(256, 256), n_spots=200)
    odd = frames[:, :201, :77].copy()                   # odd sizes: vector-ALU DFT, Bluestein FFT lengths
    for fr in (frames, odd):
        for uf in (1, 20):
            a = pc.phase_correlate_batch(fr[:-1], fr[1:], uf)
            b = pc.phase_correlate_batch(fr[:-1].astype(np.float64), fr[1:].astype(np.float64), uf)
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
            monkeypatch.setenv("FSQ_REGISTER_Z2Z", "1")
            c = pc.phase_correlate_batch(fr[:-1].astype(np.float64), fr[1:].astype(np.float64), uf)
            monkeypatch.delenv("FSQ_REGISTER_Z2Z")
            assert np.array_equal(a[:, :2], c[:, :2])
            assert np.abs(a[:, 2:] - c[:, 2:]).max() < 1e-9


def test_two_threads_on_their_own_streams(pc):
    """Same-shaped batches registered from two host threads on two streams at once (plans are cached per stream and
    calls serialise on the plan lock while they enqueue): both get the single-threaded result."""
    import threading
    import torch
    from fluorosequencingimageanalysis_amd import _native as N, engine as E, synth
    frames, _ = synth.make_cycle_stack(13, n_cycles=9, shape=(128, 128), n_spots=60)
    d = E.to_device_u16(frames)
    ref = pc.Registrar(8, 128, 128, 20).register(d[:-1], d[1:]).cpu().numpy()
    res, errs = {}, []

    def work(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                R = pc.Registrar(8, 128, 128, 20)
                for _ in range(20):
                    out = R.register(d[:-1], d[1:])
                st.synchronize()
                res[k] = out.cpu().numpy()
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert np.array_equal(res[0], ref) and np.array_equal(res[1], ref)


def _shifted(rng, H, W, dh, dw):
    """Smooth random content and a copy circularly shifted by (dh, dw) pixels (a Fourier-domain shift: any real dh, dw)."""
    base = rng.normal(0, 1, (H, W))
    F = np.fft.fft2(base)
    F *= np.exp(-0.5 * ((np.fft.fftfreq(H)[:, None] * 40) ** 2 + (np.fft.fftfreq(W)[None, :] * 40) ** 2))      # low-pass
    ref = np.fft.ifft2(F).real
    ramp = np.exp(-2j * np.pi * (np.fft.fftfreq(H)[:, None] * dh + np.fft.fftfreq(W)[None, :] * dw))
    return ref, np.fft.ifft2(F * ramp).real


def test_any_shape_registers(pc):
    """phase_correlate.py:137-196 works for any H x W: shapes that fit neither the MFMA tiles (cols % 16, rows % 4) nor the
    LDS of the vector-ALU DFT in one piece (2 048 x 2 050: its twiddle vectors and partial sums are tiled), prime x prime
    (Bluestein FFT lengths): the recovered shift equals the one put in on the 1 / upsample_factor grid; a small prime shape
    equals the oracle; many distinct batch sizes do not pile up plans."""
    import oracle as O
    rng = np.random.default_rng(5)
    for (H, W), (dh, dw) in (((2048, 2050), (7.0, -11.0)), ((2048, 2050), (-3.35, 0.65)), ((1031, 1033), (12.4, -7.7)),
                             ((3001, 40), (-1.25, 2.0))):
        ref, reg = _shifted(rng, H, W, dh, dw)
        r = pc.phase_correlate(ref, reg, upsample_factor=20)
        assert abs(float(r[0]) + dh) < 1e-9 and abs(float(r[1]) + dw) < 1e-9, ((H, W), (dh, dw), r)
        assert float(r[2]) < 1e-3
    a = np.zeros((1, 2048, 2050), np.uint16)
    a[0, 100, 100] = 1000
    assert pc.phase_correlate_batch(a, a, 20)[0][:2].tolist() == [0.0, 0.0]
    assert pc.phase_correlate_batch(a, a, 1)[0][:2].tolist() == [0.0, 0.0]
    ref, reg = _shifted(rng, 61, 67, 2.3, -4.45)
    r, e = pc.phase_correlate(ref, reg, upsample_factor=20), O.phase_correlate(ref, reg, 20)
    # (error = sqrt(|1 - peak^2 / (amp amp)|) of an exact shift is the square root of rounding noise: compare its square)
    assert float(r[0]) == e[0] and float(r[1]) == e[1] and abs(float(r[2]) ** 2 - e[2] ** 2) < 1e-12 and abs(float(r[3]) - e[3]) < 1e-9
    img = np.random.default_rng(0).integers(0, 4000, (40, 32, 48)).astype(np.uint16)
    for n in range(1, 40, 3):                           # 13 batch sizes x 2 plans: beyond the cache bound
        r = pc.phase_correlate_batch(img[:n], np.roll(img[:n], (2, -3), axis=(1, 2)), 1)
        assert (r[:, 0] == -2).all() and (r[:, 1] == 3).all()

#!/usr/bin/env python3
"""Fuzz of the FSQ_PIXELS_U32 path on the GPU against the oracle (TEST TOOL; results quoted in DESIGN.md):
  1. random wide fields (synthetic fields scaled into 17..31 bits, offsets, noise, saturation at 2^31 - 1) with random detection /
     consolidation parameters through pflib.find_peptides_batch -> every field's dict == the oracle's find_peptides; the same
     stack through pflib.find_peptides_records (fit queue for 32-bit pixels, 428-byte records) -> the same dicts;
  2. adversarial stand-alone 5x5 ROIs with 32-bit values (flat, hot pixels, ramps, full-range noise, saturated peaks) laid out as
     a strip image and fitted through fsq_fit_candidates | FSQ_PIXELS_U32_FLAG in both fp64 modes -> parameters, status,
     iteration / evaluation counts and the metrics of every row == the oracle's.
usage: python3 tools/fuzz_wide.py [seed=2031] [shapes=8] [rois=40000]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from fluorosequencingimageanalysis_amd import _native as N, engine as E, pflib, synth  # noqa: E402

O.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2031
n_shapes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_rois = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
rng = np.random.default_rng(seed)
t0 = time.time()
TOP = 2 ** 31 - 1


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def wide_field(H, W):
    img = synth.make_field(int(rng.integers(1 << 30)), (H, W), int(rng.integers(0, max(2, H * W // 300)))).astype(np.int64)
    kind = int(rng.integers(6))
    if kind == 0:
        img = img * int(rng.integers(2, 30000))
    elif kind == 1:
        img = img * int(rng.integers(2, 3000)) + int(rng.integers(0, 1 << 28))
    elif kind == 2:                                         # saturating at the top of the domain
        img = np.minimum(img * int(rng.integers(20000, 400000)), TOP) if rng.random() < 0.3 else np.minimum(img * int(rng.integers(2000, 40000)), TOP)
    elif kind == 3:                                         # pure noise over a random range
        img = rng.integers(0, int(rng.integers(70000, TOP if rng.random() < 0.2 else 1 << 24)), (H, W))
    elif kind == 4:                                         # 16-bit background, a few enormous pixels
        hot = rng.random((H, W)) < 0.002
        img = np.where(hot, rng.integers(1 << 20, TOP if rng.random() < 0.3 else 1 << 27, (H, W)), img)
    else:                                                   # barely beyond 16 bits
        img = img + 65000
    img = np.clip(img, 0, TOP)
    if img.max() <= 65535:
        img[0, 0] = 65536 + int(rng.integers(0, 1000))
    return img.astype(np.uint32)


# ---- 1. fields through find_peptides_batch ---------------------------------------------------------------------------------
n_fields = n_peaks = n_assert = n_refused = 0
for shape_i in range(n_shapes):
    H, W = int(rng.integers(24, 150)), int(rng.integers(24, 150))
    nf = int(rng.integers(3, 14))
    med = int(rng.choice([3, 4, 5, 7]))
    ks = int(rng.choice([3, 5, 5, 7]))
    if ks == 5 and rng.random() < 0.6:
        K = pflib.default_correlation_matrix
    else:
        K = rng.integers(-3000, 6000, (ks, ks)).astype(np.int64)
        K[ks // 2, ks // 2] = int(rng.integers(8000, 40000))
    c_std = float(rng.choice([1.0, 2.0, 3.5]))
    r2 = float(rng.choice([0.3, 0.7, 0.9]))
    rad = int(rng.choice([2, 4, 7]))
    imgs = np.stack([wide_field(H, W) for _ in range(nf)])
    old = pflib.CHUNK_PIXELS
    pflib.CHUNK_PIXELS = int(rng.integers(1, nf + 1)) * H * W          # slices of 1 .. nf fields (the last one padded)
    kw = dict(median_filter_size=med, correlation_matrix=K, c_std=c_std, r_2_threshold=r2, consolidation_radius=rad)
    try:
        got = pflib.find_peptides_batch(imgs, errors='return', **kw)
        # the same stack as record tables (the continuous-batching pipeline, a fit queue for 32-bit pixels) must give the same dicts
        rec, counts, fmt = pflib.find_peptides_records(imgs, **{k: v for k, v in kw.items()})
        again = pflib.records_to_dicts(rec, counts, fmt)
        assert fmt == N.PIXELS_U32 and len(again) == len(got)
        for a, b in zip(got, again):
            assert isinstance(a, Exception) == isinstance(b, Exception)
            if not isinstance(a, Exception):
                assert list(a) == list(b), shape_i
                for k in a:
                    assert all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for x, y in zip(a[k], b[k])), shape_i
    except NotImplementedError:
        # the response image of some field sums to 2^53 or more, where numpy.mean (pflib.py:250) is no longer the exact integer
        # mean: refused by design, for the whole stack.  Field by field then: refused exactly where the oracle refuses
        got, n_ref = [], 0
        for f in range(nf):
            try:
                got.append(pflib.find_peptides_batch(imgs[f:f + 1], errors='return', **kw)[0])
            except NotImplementedError:
                try:
                    O.candidates(imgs[f], med_size=med, K=K, c_std=c_std)
                except ValueError:
                    got.append(None)
                    n_ref += 1
                else:
                    raise AssertionError("shape %d field %d: the GPU path refused what the oracle accepts" % (shape_i, f))
        n_refused += n_ref
    finally:
        pflib.CHUNK_PIXELS = old
    for f, d in enumerate(got):
        if d is None:
            continue
        try:
            rows, fits, keep, key = O.find_peptides(imgs[f], med_size=med, K=K, c_std=c_std, r2_thr=r2, radius=rad, n_threads=16)
        except AssertionError:
            assert isinstance(d, AssertionError), (shape_i, f)
            n_assert += 1
            continue
        assert not isinstance(d, Exception), (shape_i, f, d)
        assert np.array_equal(np.array(list(d.keys()), dtype=np.int32).reshape(-1, 2), key), (shape_i, f)
        vals = list(d.values())
        r = rows[keep]
        t7 = np.array([[float(x) for x in v[:7]] for v in vals]).reshape(-1, 7)
        assert bits_equal(t7, np.stack([r[k] for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")], axis=1)).all(), (shape_i, f)
        m = np.array([[float(v[9]), float(v[10]), float(v[11])] for v in vals]).reshape(-1, 3)
        assert bits_equal(m, np.stack([r["rmse"], r["r2"], r["s_n"]], axis=1)).all(), (shape_i, f)
        for v, h, w in zip(vals, r["h"], r["w"]):
            assert np.array_equal(v[7], imgs[f][h - 2:h + 3, w - 2:w + 3].astype(np.int64)), (shape_i, f)
        n_peaks += len(vals)
    n_fields += nf
    print("shape %d: %dx%d x %d fields ok (%.0f s)" % (shape_i, H, W, nf, time.time() - t0), flush=True)
print("fields: %d checked, %d peaks, %d re-key assertions reproduced, %d fields refused as by the oracle (response sum >= 2^53)"
      % (n_fields, n_peaks, n_assert, n_refused), flush=True)

# ---- 2. adversarial ROIs ----------------------------------------------------------------------------------------------------
torch = E._torch()


def roi_batch(n):
    out = np.empty((n, 25), np.int64)
    kinds = rng.integers(0, 8, n)
    for i, k in enumerate(kinds):
        top = int(rng.integers(70000, TOP))
        if k == 0:
            r = np.full(25, int(rng.integers(0, top)))
        elif k == 1:
            r = rng.integers(0, top, 25)
        elif k == 2:                                        # one hot pixel on a flat or noisy floor
            r = rng.integers(0, int(rng.integers(1, 5000)), 25)
            r[int(rng.integers(25))] = top
        elif k == 3:                                        # a Gaussian spot of random width / amplitude / centre, saturating or not
            yy, xx = np.mgrid[0:5, 0:5]
            s = rng.uniform(0.3, 3.0, 2)
            c = rng.uniform(0.5, 3.5, 2)
            g = np.exp(-((yy - c[0]) ** 2 / (2 * s[0] ** 2) + (xx - c[1]) ** 2 / (2 * s[1] ** 2)))
            r = np.minimum(rng.integers(0, 1 << 16) + g.ravel() * rng.uniform(1e4, 6e9), TOP).astype(np.int64)
        elif k == 4:                                        # ramps
            r = (np.arange(25) * int(rng.integers(1, top // 25 + 1))) if rng.random() < 0.5 else (np.arange(25)[::-1] * int(rng.integers(1, top // 25 + 1)))
        elif k == 5:                                        # two levels
            r = np.where(rng.random(25) < 0.5, int(rng.integers(0, 1000)), top)
        elif k == 6:                                        # everything at the top of the domain but one pixel
            r = np.full(25, TOP)
            r[int(rng.integers(25))] = int(rng.integers(0, TOP))
        else:                                               # 16-bit values (the 32-bit kernels on small numbers)
            r = rng.integers(0, 65536, 25)
        out[i] = r
    return out


def gpu_fit(rois, mode):
    n = len(rois)
    img = np.ascontiguousarray(rois.reshape(n, 5, 5).transpose(1, 0, 2).reshape(5, 5 * n).astype(np.uint32))      # ROI i = columns 5i .. 5i + 4
    d_img = E.to_device_pixels(img[None], N.PIXELS_U32)
    cand = np.zeros((n, 3), np.int32)
    cand[:, 1] = 2
    cand[:, 2] = 5 * np.arange(n) + 2
    d_cand = torch.from_numpy(cand).cuda()
    rows = torch.zeros(n * 128, dtype=torch.uint8, device="cuda")
    ws = torch.zeros(N.lib().fsq_fit_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    rc = N.lib().fsq_fit_candidates(d_img.data_ptr(), 1, 5, 5 * n, d_cand.data_ptr(), n, mode | N.PIXELS_U32_FLAG, rows.data_ptr(),
                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    N.check(rc, "fsq_fit_candidates")
    torch.cuda.synchronize()
    return rows.cpu().numpy().view(N.ROW_DTYPE)


done = 0
while done < n_rois:
    n = min(20000, n_rois - done)
    rois = roi_batch(n)
    for mode in (0, 1):
        got = gpu_fit(rois, mode)
        ref = O.fit_rois(rois, mode=mode, n_threads=16)
        p = np.stack([got[k] for k in ("H", "A", "p2", "p3", "sigma_h", "sigma_w", "theta")], axis=1)
        bad = ~bits_equal(p, ref["p"]).all(axis=1)
        assert not bad.any(), "mode %d: %d of %d fits differ, first ROI %s" % (mode, bad.sum(), n, rois[np.nonzero(bad)[0][0]])
        for k in ("status", "niter", "nfev"):
            assert np.array_equal(got[k], ref[k]), (mode, k)
        exp = np.zeros(1, O.ROW_DTYPE)
        for i in range(0, n, 97):
            roi = np.ascontiguousarray(rois[i])
            O.lib().fsq_o_fit_metrics(roi.ctypes.data_as(ctypes.c_void_p), ref["p"][i].ctypes.data_as(ctypes.c_void_p), 2, 5 * i + 2,
                                      exp.ctypes.data_as(ctypes.c_void_p))
            for k in ("h0", "w0", "rmse", "r2", "s_n"):
                assert bits_equal(got[k][i], exp[k][0]).all(), (mode, k, i)
    done += n
    print("rois: %d done, status mix %s (%.0f s)" % (done, dict(zip(*np.unique(ref["status"], return_counts=True))), time.time() - t0), flush=True)
print("FUZZ WIDE OK seed %d" % seed)

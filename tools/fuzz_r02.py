#!/usr/bin/env python3
"""One-off fuzz of the round-2 paths on the GPU against the oracle (TEST TOOL; results quoted in DESIGN.md):
  1. random small fields (sparse to crowded, dim, saturated, pure noise) streamed through engine.StreamPipelineGroup in
     batches, random detection / consolidation parameters per batch -> every field's kept table == the oracle's;
  2. random spot layouts through the greedy tracker -> traces, links, discarded counts == the oracle's;
  3. random frames through the luminosity-centroid tracker == the oracle's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from fluorosequencingimageanalysis_amd import _native as N, engine as E, flexlibrary as fl, pflib, synth  # noqa: E402

O.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 2026)
t0 = time.time()

# ---- 1. fields through the stream pipeline -------------------------------------------------------------------------
n_fields_checked = n_fits = 0
for shape_i in range(10):
    H, W = int(rng.integers(24, 160)), int(rng.integers(24, 160))
    nf, nb = 12, 5
    med = int(rng.choice([3, 5, 7]))
    c_std = float(rng.choice([1.0, 2.0, 3.5]))
    r2 = float(rng.choice([0.3, 0.7, 0.9]))
    rad = int(rng.choice([2, 4, 7]))
    batches = []
    for b in range(nb):
        fs = []
        for i in range(nf):
            img = synth.make_field(int(rng.integers(1 << 30)), (H, W), int(rng.integers(0, max(2, H * W // 300))))
            mode = int(rng.integers(5))
            if mode == 1:
                img = np.minimum(img.astype(np.int64) * int(rng.integers(5, 40)), 65535).astype(np.uint16)
            elif mode == 2:
                img = (img // 40).astype(np.uint16)
            elif mode == 3:
                img = rng.integers(0, int(rng.integers(2, 5000)), (H, W)).astype(np.uint16)
            fs.append(img)
        batches.append(np.stack(fs))
    fmt = N.PIXELS_U16
    if rng.random() < 0.3:          # fp16 pixel loads: the fields as binary16 (pre-scaled when they exceed its range), the oracle on the truncated values
        halves = [E.quantise_f16(b)[0] for b in batches]
        words = [E.as_pixel_fields(h)[0] for h in halves]
        batches = [E.pixel_values(w, N.PIXELS_F16).astype(np.uint16) for w in words]
        fmt = N.PIXELS_F16
    else:
        words = batches
    prm = E.detect_params(med, pflib.default_correlation_matrix, c_std, fmt)
    d = [E.to_device_u16(w) for w in words]
    group = E.StreamPipelineGroup(nf, H, W, queues=2, depth=4, inject_below=int(rng.choice([0, 500, 1 << 40])))
    got = {}

    def on_done(j, k, eng, total):
        table, offs = eng.kept_table()
        got[(j, k)] = (table.cpu().numpy().view(N.ROW_DTYPE).reshape(-1).copy(), offs.cpu().numpy().copy(), eng.nkeep.cpu().numpy().copy())
    group.run([(x, prm) for x in d], on_done, r2_threshold=r2, radius=rad, py2_round=True)
    cut = group.cut
    group.close()
    # the whole path as one library call (fsq_find_peptides) on one batch, buffers far too small at first, fewer fields than the
    # runner was built for: the same tables
    jb = int(rng.integers(nb))
    m = int(rng.integers(1, nf + 1))
    pr = E.PathRunner(nf, H, W, cand_cap=int(rng.integers(1, 64)), record_cap=int(rng.integers(1, 8)))
    rec, roffs, rnk, _ = pr.run(d[jb][:m], prm, r2, rad, N.MODE_REF, True)
    rec, roffs, rnk = rec.cpu().numpy(), roffs.cpu().numpy(), rnk.cpu().numpy()
    for f in range(m):
        k = 0 if f < cut[1] else 1
        table, offs, nkeep = got[(jb, k)]
        ff = f - cut[k]
        assert rnk[f] == nkeep[ff], ("path runner", shape_i, f)
        if nkeep[ff] >= 0:
            mine = E.peak_record_view(rec[roffs[f]:roffs[f + 1]])
            t = table[offs[ff]:offs[ff + 1]]
            for name in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta", "rmse", "r2", "s_n", "key_h", "key_w"):
                x, y = np.ascontiguousarray(mine[name]), np.ascontiguousarray(t[name])
                assert x.tobytes() == y.tobytes(), ("path runner", shape_i, f, name)
    for j in range(nb):
        for k in range(2):
            table, offs, nkeep = got[(j, k)]
            for f in range(cut[k + 1] - cut[k]):
                img = batches[j][cut[k] + f]
                try:
                    rows, fits, keep, key = O.find_peptides(img, med_size=med, c_std=c_std, r2_thr=r2, radius=rad, n_threads=16)
                except AssertionError:
                    assert nkeep[f] == -1, "oracle raised the re-key assertion, GPU did not"
                    continue
                assert nkeep[f] == len(keep), (shape_i, j, k, f)
                t = table[offs[f]:offs[f + 1]]
                r = rows[keep]
                assert np.array_equal(np.stack([t["key_h"], t["key_w"]], 1), key)
                for a, b in (("h0", "h0"), ("w0", "w0"), ("H", "H"), ("A", "A"), ("sigma_h", "sigma_h"), ("sigma_w", "sigma_w"),
                             ("theta", "theta"), ("rmse", "rmse"), ("r2", "r2"), ("s_n", "s_n")):
                    x, y = t[a], r[b]
                    assert ((x.view(np.uint64) == y.view(np.uint64)) | (np.isnan(x) & np.isnan(y))).all(), (shape_i, j, k, f, a)
                n_fields_checked += 1
                n_fits += len(rows)
print("fields: %d fields, %d LM fits identical to the oracle (%.0f s)" % (n_fields_checked, n_fits, time.time() - t0), flush=True)

# ---- 2. greedy tracking ---------------------------------------------------------------------------------------------
t0 = time.time()
checked = asserted = 0
for rep in range(12):
    F = int(rng.integers(2, 12))
    H, W = int(rng.integers(40, 200)), int(rng.integers(40, 200))
    radius = int(rng.choice([1, 2, 3]))
    spot_radius = float(rng.choice([0, 0, 2]))
    fields, offs = [], []
    for k in range(60):
        nsp = int(rng.integers(0, 120))
        base = np.stack([rng.integers(0, H, nsp), rng.integers(0, W, nsp)], 1)
        frames, o, cum = [], [(0, 0)], np.zeros(2)
        for f in range(F):
            if f:
                step = np.round(rng.uniform(-3, 3, 2) * 20) / 20 if rng.uniform() < 0.7 else rng.integers(-3, 4, 2).astype(float)
                o.append((float(step[0]), float(step[1])))
                cum = cum + step
            pts = np.rint(base - cum).astype(np.int64) + rng.integers(-1, 2, base.shape)
            pts = pts[rng.uniform(size=len(pts)) > 0.25]
            if len(pts):
                pts = pts[np.unique(pts[:, 0] * 100000 + pts[:, 1], return_index=True)[1]]
            frames.append(pts.reshape(-1, 2))
        fields.append(frames)
        offs.append(o)
    # per field: the oracle decides whether the reference would assert; fields that do are run alone to see the GPU raise too
    ok_fields, ok_offs, exp = [], [], []
    for frames, o in zip(fields, offs):
        try:
            exp.append(O.greedy_tracking(frames, o, (H, W), radius, spot_radius))
            ok_fields.append(frames)
            ok_offs.append(o)
        except AssertionError:
            asserted += 1
            try:
                fl.track_fields([frames], [o], (H, W), radius, spot_radius)
                raise SystemExit("GPU did not raise the bin assertion")
            except AssertionError:
                pass
    res = fl.track_fields(ok_fields, ok_offs, (H, W), radius, spot_radius)
    for r, e in zip(res, exp):
        assert np.array_equal(r[0], e[0]) and r[1] == e[1] and np.array_equal(r[2], e[2]) and np.array_equal(r[3], e[3]) and np.array_equal(r[4], e[4])
        checked += 1
print("tracking: %d fields identical to the oracle, %d with the reference's bin assertion raised on both sides (%.0f s)" % (checked, asserted, time.time() - t0), flush=True)

# ---- 3. luminosity-centroid tracking --------------------------------------------------------------------------------
t0 = time.time()
checked = 0
for rep in range(30):
    F, H, W = int(rng.integers(2, 10)), int(rng.integers(16, 120)), int(rng.integers(16, 120))
    nfld = 4
    frames = rng.integers(1, int(rng.integers(50, 4000)), (nfld, F, H, W)).astype(np.uint16)
    for k in range(nfld):
        for s in range(6):
            h, w = rng.integers(3, H - 3), rng.integers(3, W - 3)
            frames[k, :, h - 1:h + 2, w - 1:w + 2] += np.uint16(rng.integers(500, 20000))
    R = int(rng.choice([1, 2, 3, 4]))
    cut = float(rng.choice([0.5, 3.0, 8.0]))
    off = rng.integers(-4, 5, (nfld, F, 2))
    off[:, 0] = 0
    nsp = 40
    init = np.stack([rng.integers(2, H - 2, nsp), rng.integers(2, W - 2, nsp)], 1)
    fld = rng.integers(0, nfld, nsp).astype(np.int32)
    got, pres = fl.centroid_track_fields(frames, init, fld, R, cut, off)
    for k in range(nfld):
        m = fld == k
        e, ep = O.centroid_tracking(frames[k], init[m], R, cut, off[k])
        assert np.array_equal(got[m], e) and np.array_equal(pres[m], ep)
        checked += int(m.sum())
print("centroid tracking: %d spot tracks identical to the oracle (%.0f s)" % (checked, time.time() - t0), flush=True)

"""GPU test (-m gpu) of fsq_consolidate alone on adversarial tables: the reference's sequential dict loops (pflib.py:466, 477-519:
R^2 filter, consolidation, re-key with its assert) restated in Python, against the kernel - which since round 4 lets the waves of a
block take the candidates' turns concurrently wherever the windows do not interact.  The tables are made to interact as much as
possible: up to every pixel a survivor, long chains of overlapping windows, exact R^2 ties, NaN R^2 (passes the filter, loses every
comparison), fitted centres half a pixel off (re-keys, colliding re-keys: the reference's AssertionError), radii 2 .. 9 (windows
larger than the one-pass register window), fields of one block's 8 and 16 waves - in both of the kernel's schedules."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def py2_round(x):
    return int(np.floor(x + 0.5)) if x >= 0 else int(np.ceil(x - 0.5))


def reference_consolidate(h, w, h0, w0, r2, H, W, thr, radius):
    """-> (kept candidate numbers in the reference's dict order, their keys) or None when the assert of pflib.py:518 fires."""
    bins = {}
    for i in range(len(h)):
        if not (r2[i] < thr):                                   # pflib.py:466 (NaN passes)
            bins.setdefault((int(h[i]), int(w[i])), i)
    for (hh, ww), i in list(bins.items()):                      # pflib.py:479-512
        if (hh, ww) not in bins:
            continue
        dead = False
        for hd in range(max(0, hh - radius - 2), min(hh + radius + 3, H)):
            for wd in range(max(0, ww - radius - 2), min(ww + radius + 3, W)):
                if (hd == hh and wd == ww) or (hd, wd) not in bins:
                    continue
                k = bins[(hd, wd)]
                if (h0[i] - h0[k]) ** 2 + (w0[i] - w0[k]) ** 2 > radius ** 2:
                    continue
                if r2[i] > r2[k]:
                    del bins[(hd, wd)]
                else:
                    del bins[(hh, ww)]
                    dead = True
                    break
            if dead:
                break
    for (hh, ww), i in list(bins.items()):                      # pflib.py:514-519
        hr, wr = py2_round(h0[i]), py2_round(w0[i])
        if hr != hh or wr != ww:
            del bins[(hh, ww)]
            if (hr, wr) in bins:
                return None
            bins.setdefault((hr, wr), i)
    return list(bins.values()), list(bins.keys())


def make_field(rng, H, W, kind):
    """Candidate table of one field in raster order: (h, w, h0, w0, r2).  Coordinates are multiples of 1/4 so that numpy's scalar
    x ** 2 (= libm pow, which the kernel restates) and x * x agree exactly and the restatement above needs no libm model."""
    dens = {"sparse": 0.03, "medium": 0.15, "dense": 0.6, "full": 1.0, "chains": 0.0, "rekey": 0.1, "wild": 0.08}[kind]
    hh, ww = np.mgrid[2:H - 2, 2:W - 2]
    hh, ww = hh.ravel(), ww.ravel()
    if kind == "chains":                                        # rows of candidates 3 px apart: every window overlaps the next one's
        m = (hh % 5 == 2) & (ww % 3 == 0)
    else:
        m = rng.random(len(hh)) < dens
    h, w = hh[m], ww[m]
    n = len(h)
    off = rng.integers(-2, 3, (n, 2)) / 4.0                     # centres within +-0.5 of the pixel, on a 1/4 grid (0.5: re-keys)
    if kind == "wild":                                          # centres up to 3 px off their pixel (no real fit does that: the assert's cases)
        off = rng.integers(-12, 13, (n, 2)) / 4.0
    elif kind != "rekey":
        off = np.where(np.abs(off) == 0.5, 0.25, off)
    h0, w0 = h + off[:, 0], w + off[:, 1]
    r2 = rng.integers(0, 8, n) / 8.0 + 0.2                      # few distinct values: exact ties everywhere
    r2[rng.random(n) < 0.05] = np.nan
    r2[rng.random(n) < 0.1] = 0.1                               # below the threshold
    return h.astype(np.int32), w.astype(np.int32), h0, w0, r2


@pytest.mark.parametrize("form", ["components", "blocks"])
@pytest.mark.parametrize("H,W,radius", [(40, 56, 4), (33, 47, 2), (64, 64, 7), (48, 40, 9), (1100, 1000, 4)])
def test_consolidation_equals_the_sequential_reference(H, W, radius, form, monkeypatch):
    """form: the kernel's two schedules (csrc/fsq_consolidate.hip) - connected components of overlapping windows, each walked by one
    wave, and one block per field whose waves take turns by dependency; the library picks one by the batch's shape, here each is forced."""
    import torch
    monkeypatch.setenv("FSQ_CONSOLIDATE_" + form.upper(), "1")
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluorosequencingimageanalysis_amd import _native as N, engine as E
    rng = np.random.default_rng(1000 * H + radius)
    big = H * W > (1 << 20)                                     # (16 waves per field beyond a megapixel)
    kinds = ["sparse", "chains"] if big else ["sparse", "medium", "dense", "full", "chains", "rekey", "rekey", "medium"] + ["wild"] * 6
    fields = [make_field(rng, H, W, k) for k in kinds]
    counts = np.array([len(f[0]) for f in fields] + [0], np.int32)
    offsets = np.concatenate([[0], np.cumsum(counts[:-1])]).astype(np.int32)
    total = int(counts[:-1].sum())
    counts[-1] = total
    rows = np.zeros(total, N.ROW_DTYPE)
    for f, (h, w, h0, w0, r2) in enumerate(fields):
        a = slice(offsets[f], offsets[f] + len(h))
        rows["h"][a], rows["w"][a], rows["h0"][a], rows["w0"][a], rows["r2"][a], rows["field"][a] = h, w, h0, w0, r2, f
    eng = E.Engine(len(fields), H, W, fit_workspace=False, cand_per_field=H * W)
    eng.rows[:total].copy_(torch.from_numpy(rows.view(np.uint8).reshape(total, 128)))
    eng.counts.copy_(torch.from_numpy(counts))
    eng.offsets.copy_(torch.from_numpy(offsets))
    for rep in range(3 if not big else 1):                      # (the schedule of the waves differs from run to run; the result must not)
        eng.consolidate(0.5, radius, True)
        torch.cuda.synchronize()
        nkeep = eng.nkeep.cpu().numpy()
        keep = eng.keep[:max(total, 1)].cpu().numpy()
        out = eng.rows[:total].cpu().numpy().view(N.ROW_DTYPE).reshape(-1)
        n_assert = 0
        for f, (h, w, h0, w0, r2) in enumerate(fields):
            exp = reference_consolidate(h, w, h0, w0, r2, H, W, 0.5, radius)
            if exp is None:
                assert nkeep[f] == -1, "field %d (%s): the reference's assert fires, the kernel reports %d peaks" % (f, kinds[f], nkeep[f])
                n_assert += 1
                continue
            idx, keys = exp
            assert nkeep[f] == len(idx), "field %d (%s): %d kept, the reference keeps %d" % (f, kinds[f], nkeep[f], len(idx))
            got = keep[offsets[f]:offsets[f] + nkeep[f]] - offsets[f]
            assert np.array_equal(got, np.array(idx, dtype=np.int64)), "field %d (%s): kept set / order differs" % (f, kinds[f])
            kk = out[keep[offsets[f]:offsets[f] + nkeep[f]]]
            assert [(int(a), int(b)) for a, b in zip(kk["key_h"], kk["key_w"])] == keys
        if radius == 2:
            assert 1 <= n_assert < 6                             # colliding re-keys, and "wild" fields without one, are among the cases

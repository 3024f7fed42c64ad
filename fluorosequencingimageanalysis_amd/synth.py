"""Seeded synthetic fluorosequencing fields (own code).

The reference ships no example data, so every test, golden vector and bench
input is made here: flat background + circular-Gaussian spots + Poisson shot
noise + Gaussian read noise, rounded and clipped to uint16 - the same kind of
TIRF frame `pflib.find_peptides` (reference pflib.py:284) is written for.

Only numpy is used so the generator runs identically in the build container
and on the GPU box.
"""
import numpy as np

BACKGROUND = 100.0
SIGMA_PX = 1.1
AMP_RANGE = (800.0, 3000.0)
READ_NOISE = 8.0
EDGE_MARGIN = 8


def spot_table(seed, shape=(512, 512), n_spots=200):
    """Return (rows, cols, amplitudes) of the spots of field `seed`."""
    rng = np.random.default_rng(seed)
    H, W = shape
    r = rng.uniform(EDGE_MARGIN, H - EDGE_MARGIN, n_spots)
    c = rng.uniform(EDGE_MARGIN, W - EDGE_MARGIN, n_spots)
    a = rng.uniform(AMP_RANGE[0], AMP_RANGE[1], n_spots)
    return r, c, a


def render(shape, rows, cols, amps, noise_seed, sigma=SIGMA_PX):
    """Render spots onto a noisy background; returns uint16[H, W]."""
    H, W = shape
    img = np.full(shape, BACKGROUND, dtype=np.float64)
    rad = int(np.ceil(5 * sigma))
    for r, c, a in zip(rows, cols, amps):
        r0, c0 = int(round(r)), int(round(c))
        rl, rh = max(0, r0 - rad), min(H, r0 + rad + 1)
        cl, ch = max(0, c0 - rad), min(W, c0 + rad + 1)
        if rl >= rh or cl >= ch:
            continue
        yy, xx = np.mgrid[rl:rh, cl:ch]
        img[rl:rh, cl:ch] += a * np.exp(-((yy - r) ** 2 + (xx - c) ** 2) / (2.0 * sigma * sigma))
    rng = np.random.default_rng([noise_seed, 0x5EED])
    out = rng.poisson(img).astype(np.float64) + rng.normal(0.0, READ_NOISE, shape)
    return np.clip(np.rint(out), 0, 65535).astype(np.uint16)


def make_field(seed, shape=(512, 512), n_spots=200):
    """One synthetic field. cfg1 = make_field(1, (512,512), 200)."""
    r, c, a = spot_table(seed, shape, n_spots)
    return render(shape, r, c, a, seed)


def make_hard_field(seed, shape=(256, 256), n_spots=150):
    """Edge-case field: spots anywhere (also on the borders), overlapping, some saturating."""
    rng = np.random.default_rng([seed, 0xBAD])
    H, W = shape
    r = rng.uniform(0.5, H - 0.5, n_spots)
    c = rng.uniform(0.5, W - 0.5, n_spots)
    a = 10 ** rng.uniform(2.5, 4.9, n_spots)
    return render(shape, r, c, a, seed)


def make_fields(seeds, shape=(512, 512), n_spots=500):
    """Stack of fields uint16[n, H, W] (cfg2: seeds 0..1023, 500 spots)."""
    return np.stack([make_field(s, shape, n_spots) for s in seeds])


def make_cycle_stack(seed, n_cycles=8, shape=(512, 512), n_spots=500,
                     max_drift=3.0, dropout=0.15):
    """cfg3: frames of one field x channel over cycles with cumulative sub-pixel
    drift and per-cycle spot dropout.  Returns (frames uint16[n_cycles,H,W],
    true cumulative offsets float64[n_cycles,2])."""
    rng = np.random.default_rng([seed, 0xC1C1E])
    r, c, a = spot_table(seed, shape, n_spots)
    alive = np.ones(n_spots, dtype=bool)
    off = np.zeros((n_cycles, 2))
    frames = []
    for k in range(n_cycles):
        if k > 0:
            off[k] = off[k - 1] + rng.uniform(-max_drift, max_drift, 2)
            alive &= rng.uniform(size=n_spots) >= dropout
        frames.append(render(shape, r[alive] + off[k, 0], c[alive] + off[k, 1], a[alive],
                             seed * 1000 + k))
    return np.stack(frames), off

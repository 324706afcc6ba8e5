"""Deterministic synthetic 3-D drill-hole point sets.

The reference ships no data (README.md:41-49 names train.txt/test.txt but
neither is in the repository), so every workload here is synthetic:
H = ceil(N/128) holes of 128 samples, collars uniform over a 2 km x 2 km
lease, 2 m down-hole spacing, a smooth anisotropic grade field plus noise.
Seed = 20171027 + N, so each N names exactly one data set.
"""
import numpy as np

SAMPLES_PER_HOLE = 128

# reference defaults: Kernel.cpp:763-773 (ExpAns), :317-320 (bias), GP_Utils.cpp:43 (sn2)
DEFAULT_EXPANS = (np.pi / 3.1, 1.5, np.pi / 3.1, 1.5, np.pi / 3.1, 1.3, 0.9, 0.6)
DEFAULT_BIAS = 0.2
DEFAULT_SN2 = 0.016


def drillholes_raw(N, seed=None):
    """Raw (unstandardised) coordinates in metres and grades. Returns X (N,3), y (N,)."""
    rng = np.random.default_rng(20171027 + N if seed is None else seed)
    H = (N + SAMPLES_PER_HOLE - 1) // SAMPLES_PER_HOLE
    collar = np.column_stack([rng.uniform(0, 2000, H), rng.uniform(0, 2000, H), rng.uniform(900, 1100, H)])
    az = rng.uniform(0, 2 * np.pi, H)
    dip = np.deg2rad(rng.uniform(60, 90, H))
    direction = np.column_stack([np.cos(dip) * np.sin(az), np.cos(dip) * np.cos(az), -np.sin(dip)])
    depth = 2.0 * np.arange(SAMPLES_PER_HOLE)
    X = collar[:, None, :] + depth[None, :, None] * direction[:, None, :]
    X = X.reshape(H * SAMPLES_PER_HOLE, 3) + rng.normal(0, 0.05, (H * SAMPLES_PER_HOLE, 3))
    # smooth field: 8 cosine modes, wavelength ~150 m, anisotropy 3:2:1 in a rotated frame
    frame, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    w = rng.normal(size=(8, 3)) / 150.0 * np.array([1.0 / 3.0, 1.0 / 2.0, 1.0])
    w = w @ frame.T
    a = rng.normal(0, 1, 8) / np.sqrt(8)
    phi = rng.uniform(0, 2 * np.pi, 8)
    g = (a[None, :] * np.cos(X @ w.T + phi[None, :])).sum(1)
    y = np.exp(0.8 * g + rng.normal(0, 0.1, g.shape))
    return np.asfortranarray(X[:N]), np.ascontiguousarray(y[:N])


def symmetric_standardise(X, y):
    """Control::prep_symmetric (Control.cpp:299-324) in train mode: one common centre and
    half-range for the three coordinate columns, separate ones for y.
    Returns Xs, ys, params ((d+1) x 2: offset, scale; row 0 = y) like Control.cpp:304-315."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    lo, hi = X.min(), X.max()
    ylo, yhi = y.min(), y.max()
    params = np.zeros((X.shape[1] + 1, 2))
    params[0] = (0.5 * (yhi + ylo), 0.5 * (yhi - ylo))
    params[1:4] = (0.5 * (hi + lo), 0.5 * (hi - lo))
    for j in range(3, X.shape[1]):   # further columns (rock type): their own range, Control.cpp:311-315;
        cj = X[:, j]                 # lo/hi above are over ALL input columns, as Control.h:51-52 has it
        params[j + 1] = (0.5 * (cj.max() + cj.min()), 0.5 * (cj.max() - cj.min()))
    Xs = (X - params[1:, 0][None, :]) / params[1:, 1][None, :]
    ys = (y - params[0, 0]) / params[0, 1]
    return np.asfortranarray(Xs), np.ascontiguousarray(ys), params


def drillholes(N, seed=None):
    """Standardised training set: X in [-1,1]^3 (col-major), y in [-1,1]."""
    X, y = drillholes_raw(N, seed)
    Xs, ys, _ = symmetric_standardise(X, y)
    return Xs, ys


def rock_codes(X_raw, seed=0):
    """A synthetic rock-type code 1..4 per sample: banded along a tilted direction, with 5 % mislogged."""
    rng = np.random.default_rng(4171 + len(X_raw) + seed)
    t = X_raw @ np.array([0.35, -0.2, 0.9])
    band = np.floor(4.0 * (t - t.min()) / (np.ptp(t) + 1e-12)).clip(0, 3)
    flip = rng.random(len(t)) < 0.05
    band[flip] = rng.integers(0, 4, flip.sum())
    return band + 1.0


def drillholes4(N, seed=None):
    """Standardised 4-column training set: x, y, z + rock-type code."""
    X, y = drillholes_raw(N, seed)
    X4 = np.column_stack([X, rock_codes(X)])
    Xs, ys, _ = symmetric_standardise(X4, y)
    return Xs, ys


def test_points4(M, seed=0):
    """Test locations with a rock-type column drawn from the four standardised codes."""
    P = test_points(M, seed)
    rng = np.random.default_rng(77 + M + seed)
    codes = rng.integers(0, 4, M) * (2.0 / 3.0) - 1.0
    return np.asfortranarray(np.column_stack([P, codes]))


def test_points(M, seed=0):
    """Test locations inside the standardised cube. For M a perfect cube: a regular block model
    (config 5: 100^3); otherwise uniform random points."""
    c = round(M ** (1.0 / 3.0))
    if c ** 3 == M:
        ax = np.linspace(-0.95, 0.95, c)
        gx, gy, gz = np.meshgrid(ax, ax, ax, indexing="ij")
        return np.asfortranarray(np.column_stack([gx.ravel(), gy.ravel(), gz.ravel()]))
    rng = np.random.default_rng(991 + M + seed)
    return np.asfortranarray(rng.uniform(-0.95, 0.95, (M, 3)))

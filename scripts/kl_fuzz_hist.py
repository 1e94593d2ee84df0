"""Random 2048-bin histograms of eight families for the KL sweep fuzz tests (CPU and GPU)."""
import numpy as np


def random_histogram(rng):
    kind = rng.integers(0, 8)
    x = np.arange(2048, dtype=np.float64)
    scale = 10.0 ** rng.uniform(1, 9)
    if kind == 0:      # half-gaussian of random width
        h = np.exp(-0.5 * (x / rng.uniform(20, 900)) ** 2)
    elif kind == 1:    # exponential / laplace tail
        h = np.exp(-x / rng.uniform(5, 600))
    elif kind == 2:    # ReLU-like: spike at 0 plus a tail
        h = np.exp(-x / rng.uniform(30, 400)); h[0] *= rng.uniform(10, 1e4)
    elif kind == 3:    # sparse: most bins empty
        h = np.where(rng.random(2048) < rng.uniform(0.01, 0.3), rng.random(2048), 0.0)
    elif kind == 4:    # uniform with noise
        h = 1.0 + 0.1 * rng.random(2048)
    elif kind == 5:    # bumps
        h = sum(np.exp(-0.5 * ((x - rng.uniform(0, 2047)) / rng.uniform(2, 80)) ** 2) for _ in range(rng.integers(1, 6)))
    elif kind == 6:    # a single outlier bin far out, mass near zero
        h = np.exp(-x / rng.uniform(2, 30)); h[rng.integers(1500, 2048)] += 1.0 / scale * rng.integers(1, 5)
    else:              # tiny counts (0..3 per bin)
        h = rng.integers(0, 4, 2048).astype(np.float64); scale = 1.0
    h = np.rint(h * scale)
    if rng.random() < 0.2:
        h[rng.integers(128, 2048):] = 0          # empty tail
    return h.astype(np.int64)

"""Seeded synthetic inputs shared by the CPU and GPU tests."""
import numpy as np

ESA_CLASSES = np.array([0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 95, 100], dtype=np.uint8)
HSG_CODES = np.array([0, 1, 2, 3, 4, 11, 12, 13, 14, 255], dtype=np.uint8)
# every soil code the path distinguishes + classes that are in no CSV
HSG_NASTY = np.array([0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16, 100, 254, 255], dtype=np.uint8)
ESA_NASTY = np.array([0, 1, 9, 10, 11, 20, 30, 40, 50, 60, 70, 80, 90, 95, 96, 100, 101, 200, 255],
                     dtype=np.uint8)


def make_block(seed, H, W, hsy, hsx, nasty=False, jitter=True):
    """A landcover tile, a coarse soil window and two geotransforms that overlap it.

    The soil window is offset and scaled by non-round amounts so that the
    `round` ties and the edge clamps of src/cn.c:225-229 are exercised.
    """
    rng = np.random.default_rng(seed)
    esa = rng.choice(ESA_NASTY if nasty else ESA_CLASSES, size=(H, W)).astype(np.uint8)
    coarse = rng.choice(HSG_NASTY if nasty else HSG_CODES, size=(hsy, hsx)).astype(np.uint8)
    px = 3.0 / max(W, 1)
    gt = [-111.0, px, 0.0, 39.0, 0.0, -px]
    sx = 3.0 / hsx
    sy = 3.0 / hsy
    if jitter:
        # soil grid starts a bit outside / inside the tile and is slightly finer
        soil_gt = [-111.0 - 0.37 * sx, sx * 1.013, 0.0, 39.0 + 0.61 * sy, 0.0, -sy * 0.987]
    else:
        soil_gt = [-111.0, sx, 0.0, 39.0, 0.0, -sy]
    return esa, gt, coarse, soil_gt


def random_tables(seed, n=9):
    """Reference-format tables with awkward values: negatives, >=255, 254, 0."""
    rng = np.random.default_rng(seed)
    t = rng.integers(-300, 600, size=(n, 256, 5), dtype=np.int32)
    t[rng.random(t.shape) < 0.3] = 255
    t[rng.random(t.shape) < 0.05] = 254
    t[rng.random(t.shape) < 0.05] = 0
    t[rng.random(t.shape) < 0.05] = 256
    return t

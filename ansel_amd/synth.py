"""Seeded synthetic raw input for the export pixelpipe.

No raw file ships with the reference (tests/integration has no images), so every test and
the benchmark run on a synthetic RGGB mosaic: a smooth scene (2-D sinusoids + hard edges
+ low-pass texture) seen through per-channel white-balance attenuation, with
Poisson-Gaussian noise, black level 512, white point 16383 and ~1 % clipped highlights.
"""
import numpy as np

SIZES = {
    "24MP": (6000, 4000),
    "45MP": (8256, 5504),
    "60MP": (9504, 6336),
    "100MP": (11648, 8736),
}

FILTERS_RGGB = 0x94949494  # dcraw filter word, R at (0,0)
BLACK = 512
WHITE = 16383
# as-shot white balance multipliers (R, G, B, G2) of a daylight Sony-like sensor
WB_COEFFS = (2.2900391, 1.0, 1.6503906, 1.0)


def fc(row, col, filters=FILTERS_RGGB):
    """FC(), src/develop/imageop_math.h:190-193 (vectorised)"""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    return (filters >> ((((row << 1) & 14) + (col & 1)) << 1)) & 3


def scene_rgb(width, height, seed=1):
    """linear scene-referred RGB in [0, ~1.3], float32, shape (h, w, 3)"""
    rng = np.random.default_rng(seed)
    y = np.linspace(0.0, 1.0, height, dtype=np.float32)[:, None]
    x = np.linspace(0.0, float(width) / height, width, dtype=np.float32)[None, :]
    base = np.zeros((height, width), dtype=np.float32)
    for k in range(6):
        fx, fy = rng.uniform(1.0, 40.0, 2).astype(np.float32)
        ph = np.float32(rng.uniform(0, 2 * np.pi))
        amp = np.float32(0.5 / (1 + k))
        base += amp * np.sin(np.float32(2 * np.pi) * (fx * x + fy * y) + ph)
    base = (base - base.min()) / (base.max() - base.min())
    # hard edges: a few rectangles and a diagonal
    for _ in range(8):
        x0, x1 = np.sort(rng.integers(0, width, 2))
        y0, y1 = np.sort(rng.integers(0, height, 2))
        base[y0:y1, x0:x1] *= np.float32(rng.uniform(0.3, 1.6))
    # low-pass texture: coarse random field, bilinear-ish upsample by repeat + blur along axes
    cw, ch_ = max(width // 16, 2), max(height // 16, 2)
    coarse = rng.random((ch_, cw), dtype=np.float32)
    tex = np.repeat(np.repeat(coarse, 16, axis=0), 16, axis=1)[:height, :width]
    if tex.shape != (height, width):
        tex = np.pad(tex, ((0, height - tex.shape[0]), (0, width - tex.shape[1])), mode="edge")
    base = base * (np.float32(0.75) + np.float32(0.5) * tex)
    tint = np.stack([
        np.float32(0.9) + np.float32(0.2) * np.sin(np.float32(7.0) * x + np.float32(0.3)) * np.ones_like(y),
        np.ones((height, width), dtype=np.float32),
        np.float32(0.9) + np.float32(0.2) * np.cos(np.float32(5.0) * y) * np.ones_like(x),
    ], axis=-1)
    rgb = base[..., None] * tint
    # ~1 % blown highlights
    thr = np.quantile(base[::8, ::8], 0.99)
    rgb[base > thr] *= np.float32(4.0)
    return rgb.astype(np.float32)


def bayer_mosaic(width, height, seed=1, iso=400.0, filters=FILTERS_RGGB):
    """uint16 CFA (h, w): what iop/basebuffer.c hands to rawprepare"""
    rng = np.random.default_rng(seed + 1000)
    rgb = scene_rgb(width, height, seed)
    rows = np.arange(height)[:, None]
    cols = np.arange(width)[None, :]
    c = fc(rows, cols, filters)
    c3 = np.where(c == 3, 1, c)
    sensor = np.take_along_axis(rgb, c3[..., None], axis=-1)[..., 0]
    wb = np.asarray(WB_COEFFS, dtype=np.float32)
    sensor = sensor / wb[c] * np.float32(0.85)
    a = np.float32(2e-5 * iso / 100.0)
    b = np.float32(2e-7)
    sigma = np.sqrt(np.maximum(a * sensor + b, 0)).astype(np.float32)
    sensor = sensor + sigma * rng.standard_normal(sensor.shape, dtype=np.float32)
    dn = np.rint(sensor * np.float32(WHITE - BLACK) + np.float32(BLACK))
    return np.clip(dn, 0, WHITE).astype(np.uint16)


def rgba_image(width, height, seed=1, lo=0.0, hi=1.2):
    """float4 RGBA test plane for the post-demosaic modules (alpha = 0 like demosaic output)"""
    rgb = scene_rgb(width, height, seed)
    out = np.zeros((height, width, 4), dtype=np.float32)
    out[..., :3] = lo + (hi - lo) * rgb / np.float32(max(rgb.max(), 1e-6))
    return out


def adversarial_rgba(width, height, seed=7):
    """planes for ULP tests: zeros, denormals, negatives, HDR > 1, constant colour, hot pixel"""
    rng = np.random.default_rng(seed)
    img = rng.random((height, width, 4), dtype=np.float32)
    img[..., 3] = 0.0
    h8 = max(height // 8, 1)
    img[0 * h8:1 * h8] = 0.0
    img[1 * h8:2 * h8, :, :3] = 1e-39 * rng.random((min(h8, height - h8), width, 3), dtype=np.float32)
    img[2 * h8:3 * h8, :, :3] *= -0.1
    img[3 * h8:4 * h8, :, :3] *= 64.0
    img[4 * h8:5 * h8, :, :3] = np.float32([0.18, 0.18, 0.18])
    img[5 * h8:6 * h8, :, :3] = np.float32([0.9, 0.05, 0.02])
    if height > 6 * h8 + 2 and width > 4:
        img[6 * h8 + 1, width // 2, :3] = 1000.0
    return img


def _mosaic_from_scene(rgb, rng, iso, filters):
    """the sensor model of bayer_mosaic() on a given scene: CFA sampling, white-balance attenuation, noise, quantisation"""
    height, width = rgb.shape[:2]
    rows = np.arange(height)[:, None]
    cols = np.arange(width)[None, :]
    c = fc(rows, cols, filters)
    c3 = np.where(c == 3, 1, c)
    sensor = np.take_along_axis(rgb, c3[..., None], axis=-1)[..., 0]
    wb = np.asarray(WB_COEFFS, dtype=np.float32)
    sensor = sensor / wb[c] * np.float32(0.85)
    a = np.float32(2e-5 * iso / 100.0)
    b = np.float32(2e-7)
    sigma = np.sqrt(np.maximum(a * sensor + b, 0)).astype(np.float32)
    sensor = sensor + sigma * rng.standard_normal(sensor.shape, dtype=np.float32)
    dn = np.rint(sensor * np.float32(WHITE - BLACK) + np.float32(BLACK))
    return np.clip(dn, 0, WHITE).astype(np.uint16)


def bayer_mosaic_tiled(width, height, seed=1, tile=2048, iso=400.0, filters=FILTERS_RGGB):
    """Large mosaics for the benchmark and the at-size parity tests, in seconds instead of minutes at 100 MP: the
    frame is a grid of tile x tile mosaics (tile is even, so the CFA phase is preserved), EVERY TILE WITH ITS OWN
    SEED -- its own noise realisation over its own variant of the scene (one of the four mirror images of a seeded
    base scene, its own exposure and colour cast), mosaiced after the mirroring -- so no two tiles of a 100 MP frame
    hold the same values (a first version repeated one tile 24 times: addressing at scale, value diversity of 4 MP)."""
    tw, th = min(tile, width), min(tile, height)
    base = scene_rgb(tw, th, seed)
    out = np.empty((height, width), np.uint16)
    k = 0
    for y0 in range(0, height, th):
        for x0 in range(0, width, tw):
            rng = np.random.default_rng((seed + 1000, k))
            v = base[::-1] if k & 1 else base
            v = v[:, ::-1] if k & 2 else v
            gain = (np.float32(rng.uniform(0.5, 1.05)) * (np.float32(1.0) + np.float32(0.12) * rng.standard_normal(3).astype(np.float32)))
            t = _mosaic_from_scene(v * gain.astype(np.float32), rng, iso, filters)
            hh, ww = min(th, height - y0), min(tw, width - x0)
            out[y0:y0 + hh, x0:x0 + ww] = t[:hh, :ww]
            k += 1
    return out

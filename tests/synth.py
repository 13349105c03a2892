"""Seeded synthetic images shared by the tests, the golden generator and bench.py."""
import numpy as np


def blob_image(width, height, seed, n_blobs=None, channels=1, shift=(0, 0), wrap=False):
    """Gaussian blobs over a smooth gradient plus noise, clipped to uint8. `shift`
    translates the blob field so consecutive frames of a sequence can be matched; with
    `wrap` blobs that leave on one side come back on the other, so every frame of a long
    sequence holds the same number of blobs."""
    rng = np.random.RandomState(seed)
    if n_blobs is None:
        n_blobs = max(8, width * height // 300)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    img = 60.0 + 40.0 * (xx / max(width - 1, 1)) + 30.0 * (yy / max(height - 1, 1))
    cx = rng.uniform(0, width, n_blobs) + shift[0]
    cy = rng.uniform(0, height, n_blobs) + shift[1]
    if wrap:
        cx, cy = cx % width, cy % height
    sg = rng.uniform(1.5, 9.0, n_blobs)
    am = rng.uniform(40.0, 200.0, n_blobs) * rng.choice([-1.0, 1.0], n_blobs)
    for i in range(n_blobs):
        r = int(4 * sg[i]) + 1
        x0, x1 = max(0, int(cx[i]) - r), min(width, int(cx[i]) + r + 1)
        y0, y1 = max(0, int(cy[i]) - r), min(height, int(cy[i]) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        sub = np.exp(-((xx[y0:y1, x0:x1] - cx[i]) ** 2 + (yy[y0:y1, x0:x1] - cy[i]) ** 2) / (2 * sg[i] ** 2))
        img[y0:y1, x0:x1] += am[i] * sub
    noise = np.random.RandomState(seed + 7919).normal(0.0, 2.0, (height, width)).astype(np.float32)
    gray = np.clip(img + noise, 0, 255).astype(np.uint8)
    if channels == 1:
        return gray
    out = np.empty((height, width, 3), np.uint8)
    out[..., 0] = np.clip(gray.astype(np.int32) - 10, 0, 255)
    out[..., 1] = gray
    out[..., 2] = np.clip(gray.astype(np.int32) + 10, 0, 255)
    return out


def sift_like_descriptors(n, seed, dim=128):
    """Non-negative vectors shaped like SIFT descriptors: |N(0,1)| -> unit length -> clamp 0.2
    -> unit length -> x512 -> clamp 255 (values stay fractional, as the reference leaves them)."""
    rng = np.random.RandomState(seed)
    d = np.abs(rng.standard_normal((n, dim))).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = np.minimum(d, 0.2)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.minimum(d * 512.0, 255.0).astype(np.float32)


def dense_image(width, height, seed=0, mix=0.5):
    """Alternating bright / dark 5x5 dots on a 12-pixel grid blended with blob_image: the
    first octave yields more detections than max_features = items/100, so the list limit
    (SIFT/sift.cpp:408, 1356) is exercised while the coarser octaves still find keypoints."""
    rng = np.random.RandomState(seed)
    img = np.full((height, width), 128, np.float32)
    period, size, hp = 12, 5, 6
    for y in range(3, height - size, period):
        for x in range(3, width - size, period):
            img[y:y + size, x:x + size] = rng.uniform(215, 255)
            if y + hp + size < height and x + hp + size < width:
                img[y + hp:y + hp + size, x + hp:x + hp + size] = rng.uniform(0, 40)
    img = img * (1 - mix) + mix * blob_image(width, height, seed + 1, n_blobs=width * height // 900).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)

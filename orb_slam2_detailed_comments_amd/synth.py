"""Seeded synthetic 8-bit frames: value-noise background + random
axis-aligned / rotated rectangles + pixel noise.  Datasets (TUM/KITTI/EuRoC) are not in the
image and cannot be fetched, so every test and bench input comes from here.

stream(...)  : the same scene translated by (3t mod 17, 2t mod 11) px with gain 1 +- 0.02.
stereo_pair(): right image = left scene re-rendered with a per-rectangle disparity U[4,60] px
               (rectified by construction).
"""
from __future__ import annotations
import numpy as np

_SEED_BASE = 0x5EED0000


def _value_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-amp, amp, size=(gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


class Scene:
    """A float scene larger than the frame, so translated windows can be cropped from it."""

    def __init__(self, width, height, stream_id=0, margin=64, nrect=None):
        self.w, self.h, self.margin = width, height, margin
        rng = np.random.Generator(np.random.PCG64(_SEED_BASE + stream_id))
        H, W = height + 2 * margin, width + 2 * margin
        bg = np.full((H, W), 110.0, np.float32)
        for cell, amp in ((64, 40.0), (32, 20.0), (16, 10.0)):
            bg += _value_noise(rng, H, W, cell, amp)
        self.bg = bg
        if nrect is None:
            nrect = max(8, int(round(400 * (width * height) / (640.0 * 480.0))))
        self.rects = []
        for _ in range(nrect):
            cx = rng.uniform(0, W); cy = rng.uniform(0, H)
            sx = rng.uniform(6, 60); sy = rng.uniform(6, 60)
            ang = 0.0 if rng.uniform() < 0.5 else rng.uniform(0, np.pi)
            val = rng.uniform(20, 235)
            disp = rng.uniform(4, 60)
            self.rects.append((cx, cy, sx, sy, ang, val, disp))
        self.noise_rng = np.random.Generator(np.random.PCG64(_SEED_BASE + 7919 * (stream_id + 1)))

    def _render(self, shift_by_disparity=False):
        img = self.bg.copy()
        H, W = img.shape
        for (cx, cy, sx, sy, ang, val, disp) in self.rects:
            if shift_by_disparity:
                cx = cx - disp
            r = 0.75 * (sx + sy)
            x0 = int(max(0, np.floor(cx - r))); x1 = int(min(W, np.ceil(cx + r) + 1))
            y0 = int(max(0, np.floor(cy - r))); y1 = int(min(H, np.ceil(cy + r) + 1))
            if x0 >= x1 or y0 >= y1:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
            dx = xx - cx; dy = yy - cy
            ca, sa = np.cos(ang), np.sin(ang)
            u = dx * ca + dy * sa; v = -dx * sa + dy * ca
            m = (np.abs(u) <= sx / 2) & (np.abs(v) <= sy / 2)
            img[y0:y1, x0:x1][m] = val
        return img

    def frame(self, t=0, right=False):
        if not hasattr(self, "_left"):
            self._left = self._render(False)
        if right and not hasattr(self, "_right"):
            self._right = self._render(True)
        base = self._right if right else self._left
        dx, dy = (3 * t) % 17, (2 * t) % 11
        m = self.margin
        win = base[m + dy:m + dy + self.h, m + dx:m + dx + self.w]
        gain = 1.0 + 0.02 * np.sin(0.7 * t)
        rng = np.random.Generator(np.random.PCG64(_SEED_BASE + 104729 * (t + 1) + (1 if right else 0)))
        out = win * gain + rng.normal(0.0, 2.0, size=win.shape).astype(np.float32)
        return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def stream(width, height, nframes, stream_id=0):
    sc = Scene(width, height, stream_id)
    return np.stack([sc.frame(t) for t in range(nframes)])


def stereo_pair(width, height, stream_id=0, t=0):
    sc = Scene(width, height, stream_id)
    return sc.frame(t, right=False), sc.frame(t, right=True)


def degenerate(kind, width, height):
    if kind == "flat":
        return np.full((height, width), 128, np.uint8)
    if kind == "checker":
        yy, xx = np.mgrid[0:height, 0:width]
        return (((yy + xx) & 1) * 255).astype(np.uint8)
    if kind == "square":
        im = np.full((height, width), 30, np.uint8)
        im[height // 2 - 10:height // 2 + 10, width // 2 - 10:width // 2 + 10] = 220
        return im
    raise ValueError(kind)

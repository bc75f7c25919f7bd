"""Synthetic inputs (seeded, no dataset needed): band-limited noise textures and simple stereo
sequences.  Input generation only — not part of the product path and not the oracle.
SURVEY.md §8(d): texture = Gaussian-filtered uniform noise (sigma 1.5), seed 0xD1CE."""
import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates

TEX_SEED = 0xD1CE


def texture(h, w, seed=TEX_SEED, sigma=1.5):
    rng = np.random.default_rng(seed)
    t = gaussian_filter(rng.uniform(0.0, 1.0, (h, w)), sigma)
    # add a coarser octave so all pyramid levels carry gradient
    t += 0.7 * gaussian_filter(rng.uniform(0.0, 1.0, (h, w)), sigma * 4)
    t = (t - t.min()) / (t.max() - t.min())
    return t.astype(np.float32)


def sample(tex, xs, ys):
    """bilinear sample of the float texture at (xs, ys) -> uint8 image"""
    v = map_coordinates(tex, [ys, xs], order=1, mode="reflect")
    return np.ascontiguousarray(np.clip(v * 255.0 + 0.5, 0, 255).astype(np.uint8))


class PlaneSequence:
    """A fronto-parallel textured plane seen by a translating + slowly rotating stereo rig.

    Frame k: left image samples the texture at  R(theta_k) * s_k * (u - c) + c + t_k ; the right image is the left
    one displaced by a constant disparity (plane at constant depth).  Cheap to render, gives sub-pixel
    ground-truth flow for every pixel."""

    def __init__(self, w, h, seed=TEX_SEED, disparity=12.5, margin=160):
        self.w, self.h, self.disp, self.m = w, h, disparity, margin
        self.tex = texture(h + 2 * margin, w + 2 * margin, seed)
        self.u, self.v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))

    def pose(self, k):
        tx = 40.0 * np.sin(0.07 * k) + 1.3 * k * 0.0
        ty = 25.0 * np.sin(0.045 * k + 0.5)
        th = 0.02 * np.sin(0.05 * k)
        s = 1.0 + 0.03 * np.sin(0.03 * k)
        return tx, ty, th, s

    def warp(self, k, x, y):
        tx, ty, th, s = self.pose(k)
        cx, cy = self.w / 2.0, self.h / 2.0
        c, sn = np.cos(th) * s, np.sin(th) * s
        X = c * (x - cx) - sn * (y - cy) + cx + tx + self.m
        Y = sn * (x - cx) + c * (y - cy) + cy + ty + self.m
        return X, Y

    def frame(self, k):
        X, Y = self.warp(k, self.u, self.v)
        left = sample(self.tex, X, Y)
        Xr, Yr = self.warp(k, self.u + self.disp, self.v)
        right = sample(self.tex, Xr, Yr)
        return left, right

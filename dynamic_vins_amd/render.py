"""RoomRenderer: renders the box room of sim.py as textured planes through the distorted pinhole stereo rig, so the
real front end (LK + Shi-Tomasi) runs on images that are consistent with the simulated IMU stream.
Input generation only (torch is used as an array library; runs on the GPU when one is present)."""
import numpy as np
import torch

from . import sim


def _texture(size=2048, seed=sim.TEX_SEED):
    g = torch.Generator().manual_seed(seed)
    tex = torch.zeros(size, size)
    # octaves of band-limited noise (sigma 1.5 texels at the finest level)
    for octave, (scale, amp) in enumerate([(1, 1.0), (4, 0.8), (16, 0.6), (64, 0.4)]):
        n = size // scale
        t = torch.rand(1, 1, n, n, generator=g)
        k = torch.arange(-4, 5, dtype=torch.float32)
        w = torch.exp(-k * k / (2 * 1.5 * 1.5)); w = (w / w.sum()).view(1, 1, 1, -1)
        t = torch.nn.functional.conv2d(torch.nn.functional.pad(t, (4, 4, 0, 0), mode="circular"), w)
        t = torch.nn.functional.conv2d(torch.nn.functional.pad(t, (0, 0, 4, 4), mode="circular"), w.transpose(2, 3))
        if scale > 1:
            t = torch.nn.functional.interpolate(t, size=(size, size), mode="bicubic", align_corners=False)
        t = (t - t.mean()) / (t.std() + 1e-9)
        tex += amp * t[0, 0]
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    return tex


class RoomRenderer:
    def __init__(self, cam, w, h, half=(9.0, 7.0, 3.0), texel=0.012, device=None, seed=sim.TEX_SEED, cam1=None):
        self.dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        self.w, self.h, self.half, self.texel = w, h, half, texel
        self.tex = _texture(2048, seed).to(self.dev)[None, None]
        self.rays = self._rays(cam, w, h)
        self.rays1 = self._rays(cam1, w, h) if (cam1 is not None and cam1 != cam) else self.rays      # the right camera's own intrinsics (un_cam1_pinhole.yaml differs from cam0)

    def _rays(self, cam, w, h):
        # per-pixel undistorted ray (liftProjective: 8 fixed-point iterations, PinholeCamera.cc:450-508)
        u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        mx_d, my_d = (u - cam["cx"]) / cam["fx"], (v - cam["cy"]) / cam["fy"]
        mx, my = mx_d.copy(), my_d.copy()
        k1, k2, p1, p2 = cam["k1"], cam["k2"], cam["p1"], cam["p2"]
        for _ in range(20):
            r2 = mx * mx + my * my
            rad = k1 * r2 + k2 * r2 * r2
            dx = mx * rad + 2 * p1 * mx * my + p2 * (r2 + 2 * mx * mx)
            dy = my * rad + 2 * p2 * mx * my + p1 * (r2 + 2 * my * my)
            mx, my = mx_d - dx, my_d - dy
        rays = np.stack([mx, my, np.ones_like(mx)], -1).reshape(-1, 3)
        return torch.from_numpy(rays).to(self.dev, torch.float64)

    def _render(self, R_wc, p_wc, rays=None):
        """R_wc, p_wc: camera-to-world rotation / camera centre (numpy)"""
        R = torch.from_numpy(np.ascontiguousarray(R_wc)).to(self.dev, torch.float64)
        o = torch.from_numpy(np.ascontiguousarray(p_wc)).to(self.dev, torch.float64)
        d = (self.rays if rays is None else rays) @ R.T                                  # world ray directions
        hx, hy, hz = self.half
        best_t = torch.full((d.shape[0],), float("inf"), dtype=torch.float64, device=self.dev)
        uu = torch.zeros_like(best_t); vv = torch.zeros_like(best_t)
        faces = [(0, hx, 1, 2, 0.0), (0, -hx, 1, 2, 3.3), (1, hy, 0, 2, 7.1), (1, -hy, 0, 2, 11.7), (2, hz, 0, 1, 17.9), (2, -hz, 0, 1, 23.3)]
        for axis, pos, a1, a2, offs in faces:
            denom = d[:, axis]
            t = (pos - o[axis]) / denom
            ok = (t > 1e-6) & (t < best_t)
            hit1 = o[a1] + t * d[:, a1]
            hit2 = o[a2] + t * d[:, a2]
            lim1, lim2 = self.half[a1], self.half[a2]
            ok &= (hit1.abs() <= lim1 + 1e-9) & (hit2.abs() <= lim2 + 1e-9)
            best_t = torch.where(ok, t, best_t)
            uu = torch.where(ok, hit1 / self.texel + offs * 97.0, uu)
            vv = torch.where(ok, hit2 / self.texel + offs * 53.0, vv)
        size = self.tex.shape[-1]
        # wrap into the tile with reflection
        def refl(x):
            x = torch.remainder(x, 2 * size)
            return torch.where(x >= size, 2 * size - 1 - x, x)
        gx = (refl(uu) + 0.5) / size * 2 - 1
        gy = (refl(vv) + 0.5) / size * 2 - 1
        grid = torch.stack([gx, gy], -1).view(1, self.h, self.w, 2).to(torch.float32)
        img = torch.nn.functional.grid_sample(self.tex, grid, mode="bilinear", padding_mode="reflection", align_corners=False)
        return (img[0, 0] * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).contiguous()

    def stereo(self, traj, t, t_ic1=sim.T_IC1):
        R, p = traj.R(t), traj.p(t)
        out = []
        for tic, rays in ((sim.T_IC0, self.rays), (t_ic1, self.rays1)):
            out.append(self._render(R @ sim.R_IC, p + R @ tic, rays))
        return out


class DynRoomRenderer(RoomRenderer):
    """RoomRenderer + rigid textured boxes moving through the room (dynsim.MovingBox): the dynamic variant of SURVEY 8(d).
    Besides the gray images it returns what the perception front end of the reference would deliver for the LEFT camera: the per-pixel
    object id (0 = background, MovingBox.id otherwise; the instance masks of SOLOv2 / the VIODE label image) and the depth map
    (the stereo network's disparity: the source of InstFeat::DetectExtraPoints)."""

    def _render_dyn(self, R_wc, p_wc, boxes, t, want_aux, rays=None):
        rays = self.rays if rays is None else rays
        R = torch.from_numpy(np.ascontiguousarray(R_wc)).to(self.dev, torch.float64)
        o = torch.from_numpy(np.ascontiguousarray(p_wc)).to(self.dev, torch.float64)
        d = rays @ R.T
        best_t = torch.full((d.shape[0],), float("inf"), dtype=torch.float64, device=self.dev)
        uu = torch.zeros_like(best_t); vv = torch.zeros_like(best_t)
        ident = torch.zeros(d.shape[0], dtype=torch.int32, device=self.dev)
        faces = [(0, self.half[0], 1, 2, 0.0), (0, -self.half[0], 1, 2, 3.3), (1, self.half[1], 0, 2, 7.1), (1, -self.half[1], 0, 2, 11.7), (2, self.half[2], 0, 1, 17.9), (2, -self.half[2], 0, 1, 23.3)]
        for axis, pos, a1, a2, offs in faces:
            tt = (pos - o[axis]) / d[:, axis]
            ok = (tt > 1e-6) & (tt < best_t)
            hit1 = o[a1] + tt * d[:, a1]; hit2 = o[a2] + tt * d[:, a2]
            ok &= (hit1.abs() <= self.half[a1] + 1e-9) & (hit2.abs() <= self.half[a2] + 1e-9)
            best_t = torch.where(ok, tt, best_t)
            uu = torch.where(ok, hit1 / self.texel + offs * 97.0, uu)
            vv = torch.where(ok, hit2 / self.texel + offs * 53.0, vv)
        for bi, b in enumerate(boxes):
            Rwo = torch.from_numpy(np.ascontiguousarray(b.R(t))).to(self.dev, torch.float64)
            Pwo = torch.from_numpy(np.ascontiguousarray(b.p(t))).to(self.dev, torch.float64)
            half = torch.from_numpy(np.ascontiguousarray(b.dims / 2)).to(self.dev, torch.float64)
            ob = (o - Pwo) @ Rwo                       # R_wo^T (o - P)
            db = d @ Rwo
            inv = 1.0 / torch.where(db.abs() < 1e-12, torch.full_like(db, 1e-12), db)
            t1 = (-half - ob) * inv; t2 = (half - ob) * inv
            tmin = torch.minimum(t1, t2).max(1).values; tmax = torch.maximum(t1, t2).min(1).values
            ok = (tmax >= tmin) & (tmin > 1e-6) & (tmin < best_t)
            loc = ob + tmin[:, None] * db
            rel = (loc / half).abs()
            ax = rel.argmax(1)
            a1 = torch.where(ax == 0, 1, 0); a2 = torch.where(ax == 2, 1, 2)
            sgn = torch.gather(loc, 1, ax[:, None])[:, 0].sign()
            h1 = torch.gather(loc, 1, a1[:, None])[:, 0]; h2 = torch.gather(loc, 1, a2[:, None])[:, 0]
            offs = 31.0 + 13.0 * bi + 2.0 * ax.to(torch.float64) + (sgn > 0).to(torch.float64)
            best_t = torch.where(ok, tmin, best_t)
            uu = torch.where(ok, h1 / (self.texel * 0.6) + offs * 97.0, uu)          # finer texture on the objects: they are closer and smaller
            vv = torch.where(ok, h2 / (self.texel * 0.6) + offs * 53.0, vv)
            ident = torch.where(ok, torch.full_like(ident, int(b.id)), ident)
        size = self.tex.shape[-1]

        def refl(x):
            x = torch.remainder(x, 2 * size)
            return torch.where(x >= size, 2 * size - 1 - x, x)
        gx = (refl(uu) + 0.5) / size * 2 - 1; gy = (refl(vv) + 0.5) / size * 2 - 1
        grid = torch.stack([gx, gy], -1).view(1, self.h, self.w, 2).to(torch.float32)
        img = torch.nn.functional.grid_sample(self.tex, grid, mode="bilinear", padding_mode="reflection", align_corners=False)
        img = (img[0, 0] * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).contiguous()
        if not want_aux:
            return img, None, None
        depth = (best_t * rays[:, 2]).view(self.h, self.w)          # z in the camera frame: the ray (x, y, 1) scaled by t
        return img, ident.view(self.h, self.w).to(torch.uint8).contiguous(), depth

    def stereo_dynamic(self, traj, t, boxes, t_ic1=sim.T_IC1):
        """-> (left u8, right u8, id map of the left image u8, depth map of the left image f64), all H x W"""
        R, p = traj.R(t), traj.p(t)
        left, ident, depth = self._render_dyn(R @ sim.R_IC, p + R @ sim.T_IC0, boxes, t, True)
        right, _, _ = self._render_dyn(R @ sim.R_IC, p + R @ t_ic1, boxes, t, False, self.rays1)
        return left, right, ident, depth

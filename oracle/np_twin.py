"""float64 NumPy twin of oracle/sas_oracle.c -- TEST INFRASTRUCTURE ONLY.

Same algorithm, written independently with libm/NumPy double
precision and no operation-order contract.  It cross-checks the C oracle: continuous outputs
must agree to ~1e-5, and a pixel may differ more only through a threshold decision
(alpha < 1/255, T <= 1e-4) taken on a value within float32 rounding of the threshold.
Sized for small scenes (<= a few thousand Gaussians, <= 256x256) -- or, with ``crop``, for a window of tiles of a
large frame (the config-3 golden: tests/golden/render_twin_cfg3_crop.npz).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

TILE = 16
NEAR, FAR, EPS2D = 0.01, 1e10, 0.3
ALPHA_THRESHOLD = 1.0 / 255.0


def quat_to_rotmat(q: np.ndarray) -> np.ndarray:
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def sh_basis(dirs: np.ndarray, degree: int) -> np.ndarray:
    """[N,(degree+1)^2] real SH basis in gsplat's sign convention."""
    d = dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    B = [np.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        B += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if degree >= 2:
        z2 = z * z
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        B += [0.5462742152960395 * fS1, -1.092548430592079 * z * y,
              0.9461746957575601 * z2 - 0.3153915652525201,
              -1.092548430592079 * z * x, 0.5462742152960395 * fC1]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        B += [-0.5900435899266435 * fS2, fTmp1B * fS1, fTmp0C * y,
              z * (1.865881662950577 * z2 - 1.119528997770346),
              fTmp0C * x, fTmp1B * fC1, -0.5900435899266435 * fC2]
    return np.stack(B, axis=1)


def project(means, opacities, colors, viewmat, K, W, H, quats=None, scales=None, cov6=None,
            sh_degree=3, group_id=None, group_Rt=None) -> Dict[str, np.ndarray]:
    means = np.asarray(means, np.float64).reshape(-1, 3)
    n = means.shape[0]
    op = np.asarray(opacities, np.float64).reshape(-1)
    V = np.asarray(viewmat, np.float64).reshape(4, 4)
    Kc = np.asarray(K, np.float64).reshape(3, 3)
    fx, fy, cx, cy = Kc[0, 0], Kc[1, 1], Kc[0, 2], Kc[1, 2]
    Rg = None
    if group_id is not None and group_Rt is not None:
        G = np.asarray(group_Rt, np.float64).reshape(-1, 3, 4)[np.asarray(group_id).astype(np.int64)]
        Rg = G[:, :, :3]
        means = np.einsum("nij,nj->ni", Rg, means) + G[:, :, 3]
    if quats is not None:
        R = quat_to_rotmat(np.asarray(quats, np.float64).reshape(-1, 4))
        if Rg is not None:
            R = Rg @ R
        M = R * np.asarray(scales, np.float64).reshape(-1, 1, 3)
        cov = M @ M.transpose(0, 2, 1)
    else:
        c6 = np.asarray(cov6, np.float64).reshape(-1, 6)
        cov = np.empty((n, 3, 3))
        cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2] = c6[:, 0], c6[:, 1], c6[:, 2]
        cov[:, 1, 0], cov[:, 1, 1], cov[:, 1, 2] = c6[:, 1], c6[:, 3], c6[:, 4]
        cov[:, 2, 0], cov[:, 2, 1], cov[:, 2, 2] = c6[:, 2], c6[:, 4], c6[:, 5]
        if Rg is not None:
            cov = Rg @ cov @ Rg.transpose(0, 2, 1)
    Rw, tw = V[:3, :3], V[:3, 3]
    pc = means @ Rw.T + tw
    covc = Rw @ cov @ Rw.T
    x, y, z = pc[:, 0], pc[:, 1], pc[:, 2]
    valid = (z >= NEAR) & (z <= FAR)
    zs = np.where(valid, z, 1.0)
    tan_fovx, tan_fovy = 0.5 * W / fx, 0.5 * H / fy
    lim_x_pos = (W - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (H - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    rz = 1.0 / zs
    tx = zs * np.minimum(lim_x_pos, np.maximum(-lim_x_neg, x * rz))
    ty = zs * np.minimum(lim_y_pos, np.maximum(-lim_y_neg, y * rz))
    J = np.zeros((n, 2, 3))
    J[:, 0, 0] = fx * rz
    J[:, 0, 2] = -fx * tx * rz * rz
    J[:, 1, 1] = fy * rz
    J[:, 1, 2] = -fy * ty * rz * rz
    c2 = J @ covc @ J.transpose(0, 2, 1)
    c00 = c2[:, 0, 0] + EPS2D
    c01 = c2[:, 0, 1]
    c11 = c2[:, 1, 1] + EPS2D
    det = c00 * c11 - c01 * c01
    valid &= det > 0
    dets = np.where(valid, det, 1.0)
    conic = np.stack([c11 / dets, -c01 / dets, c00 / dets], axis=1)
    mx = fx * x * rz + cx
    my = fy * y * rz + cy
    valid &= op >= np.float32(ALPHA_THRESHOLD)
    with np.errstate(invalid="ignore", divide="ignore"):
        extent = np.minimum(3.33, np.sqrt(2.0 * np.log(np.maximum(op, 1e-30) / ALPHA_THRESHOLD)))
    extent = np.where(valid, extent, 0.0)
    b = 0.5 * (c00 + c11)
    v1 = b + np.sqrt(np.maximum(0.01, b * b - det))
    r1 = extent * np.sqrt(np.maximum(v1, 0))
    rx = np.ceil(np.minimum(extent * np.sqrt(np.maximum(c00, 0)), r1))
    ry = np.ceil(np.minimum(extent * np.sqrt(np.maximum(c11, 0)), r1))
    valid &= ~((rx <= 0) & (ry <= 0))
    valid &= ~((mx + rx <= 0) | (mx - rx >= W) | (my + ry <= 0) | (my - ry >= H))
    valid &= (rx > 0) & (ry > 0)
    campos = -Rw.T @ tw
    if sh_degree >= 0:
        kk = (sh_degree + 1) ** 2
        coef = np.asarray(colors, np.float64).reshape(n, kk, 3)
        B = sh_basis(means - campos, sh_degree)
        rgb = np.maximum(np.einsum("nk,nkc->nc", B, coef) + 0.5, 0.0)
    else:
        rgb = np.asarray(colors, np.float64).reshape(n, 3)
    radii = np.stack([np.where(valid, rx, 0), np.where(valid, ry, 0)], axis=1).astype(np.int32)
    return dict(valid=valid, radii=radii, means2d=np.stack([mx, my], 1), depths=z, conics=conic,
                colors=rgb, opacities=op)


def render(means, opacities, colors, viewmat, K, width, height, quats=None, scales=None, cov6=None,
           sh_degree=3, group_id=None, group_Rt=None, background=(0, 0, 0), depth_mode=0,
           depth_key_f32: bool = True, crop=None) -> Dict[str, np.ndarray]:
    """Full frame in float64.  ``depth_key_f32`` sorts by the float32 depth (as every f32
    implementation does) so that near-ties order identically; ties fall back to index.
    ``crop = (tx0, ty0, tx1, ty1)``: only that window of 16-pixel tiles is rendered (from the Gaussians whose tile
    rectangle touches it); the frames returned are the window's pixels, ``n_isect`` its intersections."""
    W, H = int(width), int(height)
    P = project(means, opacities, colors, viewmat, K, W, H, quats, scales, cov6, sh_degree, group_id, group_Rt)
    tw, th = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    vis = np.nonzero(P["valid"])[0]
    m2, rad = P["means2d"], P["radii"].astype(np.float64)
    x0 = np.clip(np.floor((m2[:, 0] - rad[:, 0]) / TILE), 0, tw).astype(int)
    x1 = np.clip(np.ceil((m2[:, 0] + rad[:, 0]) / TILE), 0, tw).astype(int)
    y0 = np.clip(np.floor((m2[:, 1] - rad[:, 1]) / TILE), 0, th).astype(int)
    y1 = np.clip(np.ceil((m2[:, 1] + rad[:, 1]) / TILE), 0, th).astype(int)
    lists = [[] for _ in range(tw * th)]
    cx0, cy0, cx1, cy1 = (0, 0, tw, th) if crop is None else [int(v) for v in crop]
    if crop is not None:
        vis = vis[(x1[vis] > cx0) & (x0[vis] < cx1) & (y1[vis] > cy0) & (y0[vis] < cy1)]
    for i in vis:
        for ty in range(max(y0[i], cy0), min(y1[i], cy1)):
            for tx in range(max(x0[i], cx0), min(x1[i], cx1)):
                lists[ty * tw + tx].append(i)
    dkey = P["depths"].astype(np.float32) if depth_key_f32 else P["depths"]
    bg = np.asarray(background, np.float64)
    rgb = np.zeros((H, W, 3))
    alpha = np.zeros((H, W, 1))
    depth = np.zeros((H, W, 1))
    n_isect = 0
    for t, ids in enumerate(lists):
        ty, tx = divmod(t, tw)
        if not (cx0 <= tx < cx1 and cy0 <= ty < cy1):
            continue
        ys = np.arange(ty * TILE, min((ty + 1) * TILE, H))
        xs = np.arange(tx * TILE, min((tx + 1) * TILE, W))
        px, py = np.meshgrid(xs + 0.5, ys + 0.5)
        T = np.ones_like(px)
        acc = np.zeros(px.shape + (4,))
        done = np.zeros(px.shape, bool)
        ids = np.asarray(ids, dtype=np.int64)
        n_isect += len(ids)
        if len(ids):
            order = np.lexsort((ids, dkey[ids]))
            for g in ids[order]:
                dx, dy = m2[g, 0] - px, m2[g, 1] - py
                a, b, c = P["conics"][g]
                sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
                al = np.minimum(0.999, P["opacities"][g] * np.exp(-sigma))
                use = (~done) & (sigma >= 0) & (al >= ALPHA_THRESHOLD)
                nT = T * (1 - al)
                stop = use & (nT <= 1e-4)
                done |= stop
                use &= ~stop
                w = np.where(use, al * T, 0.0)
                acc[..., :3] += w[..., None] * P["colors"][g]
                acc[..., 3] += w * P["depths"][g]
                T = np.where(use, nT, T)
                if done.all():
                    break
        a_out = 1 - T
        sl = (slice(ys[0], ys[-1] + 1), slice(xs[0], xs[-1] + 1))
        alpha[sl + (0,)] = a_out
        depth[sl + (0,)] = acc[..., 3] / np.maximum(a_out, 1e-10)
        rgb[sl] = np.clip(acc[..., :3] + (1 - a_out)[..., None] * bg, 0, 1)
    if depth_mode == 1:
        depth = np.where(alpha > 0, depth, depth.max())
    if crop is not None:
        win = (slice(cy0 * TILE, min(cy1 * TILE, H)), slice(cx0 * TILE, min(cx1 * TILE, W)))
        rgb, alpha, depth = rgb[win], alpha[win], depth[win]
    return dict(rgb=rgb, alpha=alpha, depth=depth, n_visible=len(vis), n_isect=n_isect, proj=P)

"""NumPy float32 restatement of the single-pass projection's exact tile culling (sim_a_splat_amd/csrc/sas_kernels.hip:
cull_geom / cull_row / tile_reached), operation for operation, for the CPU test of the criterion."""
import numpy as np

F = np.float32


def med3(a, lo, hi):
    """v_med3_f32(a, lo, hi) for lo <= hi: a clamped to [lo, hi]."""
    return np.minimum(np.maximum(a, lo), hi)


def tile_reached(mx, my, ca, cb, cc, thr, tx, ty, px=16.0):
    """All arguments float32 arrays (one entry per (Gaussian, tile) pair; tx, ty tile coordinates).  True = kept."""
    px = F(px)
    ha, hc = F(0.5) * ca, F(0.5) * cc
    nba, nbc = -cb / ca, -cb / cc
    lim = thr + F(0.05)
    ly = (ty.astype(F) * px + F(0.5)) - my
    hy = ly + (px - F(1.0))
    dyc = med3(F(0.0), ly, hy)
    c2 = hc * dyc * dyc
    t = nba * dyc
    bdyc = cb * dyc
    lx = (tx.astype(F) * px + F(0.5)) - mx
    hx = lx + (px - F(1.0))
    dxc = med3(F(0.0), lx, hx)
    dys = med3(nbc * dxc, ly, hy)
    dxs = med3(t, lx, hx)
    t1, t3 = ha * dxc, hc * dys
    # fma_(a, b, c) in the kernel is a fused multiply-add: evaluated here in float64 and rounded once
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)
    s1 = fma(t1, dxc, t3 * dys)
    q1 = fma(cb * dxc, dys, s1)
    m1 = fma(np.full_like(s1, -2e-5), s1, q1)
    s2 = fma(ha * dxs, dxs, c2)
    q2 = fma(bdyc, dxs, s2)
    m2 = fma(np.full_like(s2, -2e-5), s2, q2)
    return ~((m1 > lim) & (m2 > lim))


def kept_pairs(o, opacities, width, height, logf):
    """From an oracle dump `o` (radii, means2d, conics of every Gaussian; sas_oracle dump=True): T3's rectangle pairs and which
    of them `tile_reached` keeps at 16-pixel binning.  Returns (n_rect, keep[bool per pair], rep[Gaussian per pair], tx, ty).
    `logf`: the contract logarithm (oracle.logf)."""
    idx = np.nonzero((o["radii"] > 0).all(axis=1))[0]
    mx, my = o["means2d"][idx, 0], o["means2d"][idx, 1]
    A, B, C = (o["conics"][idx, k] for k in range(3))
    op = np.asarray(opacities, np.float32).reshape(-1)[idx]
    thr = np.array([logf(float(np.float32(255.0) * v)) for v in op], np.float32) + np.float32(1e-3)   # (project_view: lnq + 1e-3)
    thr = np.where(thr < np.float32(85.9), thr, np.float32(np.inf)).astype(np.float32)   # (... and nothing is culled once every pixel passes)
    tw, th = (width + 15) // 16, (height + 15) // 16
    rx, ry = o["radii"][idx, 0].astype(np.float32), o["radii"][idx, 1].astype(np.float32)
    x0 = np.clip(np.floor((mx - rx) / 16), 0, tw).astype(np.int64); x1 = np.clip(np.ceil((mx + rx) / 16), 0, tw).astype(np.int64)
    y0 = np.clip(np.floor((my - ry) / 16), 0, th).astype(np.int64); y1 = np.clip(np.ceil((my + ry) / 16), 0, th).astype(np.int64)
    w, area = x1 - x0, (x1 - x0) * (y1 - y0)
    rep = np.repeat(np.arange(len(idx)), area)
    k = np.arange(int(area.sum())) - (np.cumsum(area) - area)[rep]
    tx, ty = x0[rep] + k % np.maximum(w[rep], 1), y0[rep] + k // np.maximum(w[rep], 1)
    keep = tile_reached(mx[rep], my[rep], A[rep], B[rep], C[rep], thr[rep], tx, ty)
    return int(area.sum()), keep, rep, tx, ty

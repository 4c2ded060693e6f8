"""Rigid/similarity pose algebra of the dynamic scene, pure NumPy float64.

The reference does this with viser.transforms objects inside SplatHandler
(sim_a_splat/splat/splat_handler.py:62-83, :227-314, :316-332).  Here poses are plain (R, t)
pairs and every link pose is obtained by composing similarity transforms, which is the same
algebra written once instead of term by term.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np


def mm3(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """[...,3,3] @ [...,3,3] with the products summed left to right in plain float64 operations -- what a triple loop
    does (``sas_api.cpp`` mul33), independent of the BLAS behind ``np.matmul`` (which may fuse multiply-adds): the
    library's C forms of this module's functions are held to it bit for bit (tests/test_host_logic.py)."""
    return (np.asarray(A, np.float64)[..., :, :, None] * np.asarray(B, np.float64)[..., None, :, :]).sum(axis=-2)


def mv3(A: np.ndarray, v: np.ndarray) -> np.ndarray:
    """[...,3,3] @ [...,3], summed left to right (see ``mm3``)."""
    return (np.asarray(A, np.float64) * np.asarray(v, np.float64)[..., None, :]).sum(axis=-1)


def quat_wxyz_to_matrix(q: Sequence[float]) -> np.ndarray:
    return quats_wxyz_to_matrices(np.asarray(q, dtype=np.float64).reshape(1, 4))[0]   # one arithmetic for both forms


def matrix_to_quat_wxyz(R: np.ndarray) -> np.ndarray:
    R = np.asarray(R, dtype=np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.sqrt((q * q).sum())


def quats_wxyz_to_matrices(q: np.ndarray) -> np.ndarray:
    """[K,4] (any norm) -> [K,3,3]; ``quat_wxyz_to_matrix`` is this function with K = 1, so the scalar and the
    batched form are one arithmetic."""
    q = np.asarray(q, dtype=np.float64).reshape(-1, 4)
    q = q / np.sqrt((q * q).sum(axis=1, keepdims=True))
    qq = q[:, :, None] * q[:, None, :]                   # all pairwise products: qq[:, a, b] = q_a q_b (w x y z = 0 1 2 3)
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (qq[:, 2, 2] + qq[:, 3, 3]); R[:, 0, 1] = 2 * (qq[:, 1, 2] - qq[:, 0, 3]); R[:, 0, 2] = 2 * (qq[:, 1, 3] + qq[:, 0, 2])
    R[:, 1, 0] = 2 * (qq[:, 1, 2] + qq[:, 0, 3]); R[:, 1, 1] = 1 - 2 * (qq[:, 1, 1] + qq[:, 3, 3]); R[:, 1, 2] = 2 * (qq[:, 2, 3] - qq[:, 0, 1])
    R[:, 2, 0] = 2 * (qq[:, 1, 3] - qq[:, 0, 2]); R[:, 2, 1] = 2 * (qq[:, 2, 3] + qq[:, 0, 1]); R[:, 2, 2] = 1 - 2 * (qq[:, 1, 1] + qq[:, 2, 2])
    return R


def matrices_to_quats_wxyz(R: np.ndarray) -> np.ndarray:
    """[K,3,3] -> [K,4] unit quaternions; the branches of ``matrix_to_quat_wxyz`` (trace > 0, else the largest
    diagonal element), evaluated for all K at once."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    K = R.shape[0]
    q = np.empty((K, 4))
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    pos = tr > 0
    if pos.any():
        Rp = R[pos]
        s = np.sqrt(tr[pos] + 1.0) * 2
        qp = np.stack([0.25 * s, (Rp[:, 2, 1] - Rp[:, 1, 2]) / s, (Rp[:, 0, 2] - Rp[:, 2, 0]) / s, (Rp[:, 1, 0] - Rp[:, 0, 1]) / s], axis=1)
        q[pos] = qp / np.sqrt((qp * qp).sum(axis=1, keepdims=True))
    for k in np.nonzero(~pos)[0]:
        q[k] = matrix_to_quat_wxyz(R[k])          # already normalised
    return q


class SO3:
    """The two members of ``viser.transforms.SO3`` the reference's render path touches: ``.wxyz`` and
    ``.as_matrix()``."""

    def __init__(self, wxyz: Sequence[float]):
        self.wxyz = np.asarray(wxyz, dtype=np.float64).reshape(4)

    def as_matrix(self) -> np.ndarray:
        return quat_wxyz_to_matrix(self.wxyz)


class SE3:
    """Stand-in for ``viser.transforms.SE3`` as the reference's camera dictionaries use it
    (examples/demo_pusht_splat.py:54-78: ``tf.SE3(wxyz_xyz=...)``; read back through
    ``.rotation().wxyz`` and ``.translation()``, splat_env_wrapper.py:56-63,153-154).  A real viser
    SE3 works wherever this one does: everything downstream goes through ``pose_wxyz_xyz``."""

    def __init__(self, wxyz_xyz: Sequence[float]):
        v = np.asarray(wxyz_xyz, dtype=np.float64).reshape(7)
        self.wxyz_xyz = v

    def rotation(self) -> SO3:
        return SO3(self.wxyz_xyz[:4])

    def translation(self) -> np.ndarray:
        return self.wxyz_xyz[4:].copy()


def pose_wxyz_xyz(pose) -> Tuple[np.ndarray, np.ndarray]:
    """(wxyz [4], xyz [3]) of a camera / frame pose given as an SE3-like object (``.rotation().wxyz`` and
    ``.translation()``: viser.transforms.SE3 or ``poses.SE3``), a ``(wxyz, xyz)`` pair, or a flat 7-vector."""
    if type(pose) is SE3:                                  # the common case on the per-step path: no attribute probing
        v = pose.wxyz_xyz
        return v[:4], v[4:]
    if hasattr(pose, "rotation") and hasattr(pose, "translation"):
        rot = pose.rotation() if callable(pose.rotation) else pose.rotation
        tr = pose.translation() if callable(pose.translation) else pose.translation
        return np.asarray(rot.wxyz, dtype=np.float64).reshape(4), np.asarray(tr, dtype=np.float64).reshape(3)
    if isinstance(pose, (tuple, list)) and len(pose) == 2:
        return np.asarray(pose[0], dtype=np.float64).reshape(4), np.asarray(pose[1], dtype=np.float64).reshape(3)
    v = np.asarray(pose, dtype=np.float64).reshape(-1)
    if v.shape != (7,):
        raise TypeError("pose must be SE3-like, a (wxyz, xyz) pair or a 7-vector wxyz_xyz")
    return v[:4].copy(), v[4:].copy()


def decompose_icp(icp: np.ndarray, tol: float = 1e-6) -> Tuple[float, np.ndarray, np.ndarray]:
    """Uniform scale, rotation and translation of the saved ICP 4x4 (splat_handler.py:66-83).
    Raises ValueError where the reference asserts (non-uniform scale / shear)."""
    A = np.asarray(icp, dtype=np.float64)[:3, :3]
    G = A.T @ A
    off = G[~np.eye(3, dtype=bool)]
    if np.any(np.abs(off) >= tol):
        raise ValueError("ICP matrix is not a scaled rotation (off-diagonal of R^T R)")
    s2 = float(np.mean(np.diag(G)))
    if np.any(np.abs(np.diag(G) - s2) >= tol):
        raise ValueError("ICP matrix has non-uniform scale")
    s = float(np.sqrt(s2))
    return s, A / s, np.asarray(icp, dtype=np.float64)[:3, 3].copy()


class Sim3:
    """x -> s R x + t."""

    def __init__(self, s: float, R: np.ndarray, t: np.ndarray):
        self.s, self.R, self.t = float(s), np.asarray(R, np.float64), np.asarray(t, np.float64)

    def __matmul__(self, o: "Sim3") -> "Sim3":
        return Sim3(self.s * o.s, self.R @ o.R, self.s * self.R @ o.t + self.t)

    def inv(self) -> "Sim3":
        Rt = self.R.T
        return Sim3(1.0 / self.s, Rt, -(Rt @ self.t) / self.s)


def link_splat_pose(scale: float, Ri: np.ndarray, ti: np.ndarray, Rfk: np.ndarray, tfk: np.ndarray,
                    q_msg: Sequence[float], p_msg: Sequence[float], weld_t=(0.0, 0.0, 0.0)) -> Tuple[np.ndarray, np.ndarray]:
    """Pose (R, t) of one link's splat group for a Drake draw message (splat_handler.py:239-288).

    The Gaussians of a link live in the splat frame, captured at the mask-time joint
    configuration: x_splat = ICP(FK_link(x_link)).  Moving the link to the message pose M gives
    x' = ICP(M(FK^-1(ICP^-1(x_splat)))), i.e. the group transform  ICP * M * FK^-1 * ICP^-1  with
    ICP = (s, Ri, ti) a similarity and FK, M rigid.  Scales cancel, so the result is rigid.
    """
    icp = Sim3(scale, Ri, ti)
    fk = Sim3(1.0, Rfk, tfk)
    msg = Sim3(1.0, quat_wxyz_to_matrix(q_msg), np.asarray(p_msg, np.float64) + np.asarray(weld_t, np.float64))
    g = icp @ msg @ fk.inv() @ icp.inv()
    return g.R, g.t


def link_splat_poses(scale: float, Ri: np.ndarray, ti: np.ndarray, Rfk: np.ndarray, tfk: np.ndarray, q_msg: np.ndarray,
                     p_msg: np.ndarray, weld_t=(0.0, 0.0, 0.0)) -> Tuple[np.ndarray, np.ndarray]:
    """``link_splat_pose`` for K links at once (Rfk [K,3,3], tfk [K,3], q_msg [K,4], p_msg [K,3]) in the
    reference's expanded form (splat_handler.py:265-278):
        R = Ri Rm Rfk^T Ri^T,   t = ti - R ti + s Ri (tm - Rm Rfk^T tfk),   tm = p_msg + weld."""
    Rm = quats_wxyz_to_matrices(q_msg)
    tm = np.asarray(p_msg, np.float64).reshape(-1, 3) + np.asarray(weld_t, np.float64)
    Ri, ti = np.asarray(Ri, np.float64), np.asarray(ti, np.float64)
    RmF = mm3(Rm, np.transpose(np.asarray(Rfk, np.float64), (0, 2, 1)))        # Rm Rfk^T
    R = mm3(mm3(Ri, RmF), Ri.T)
    t = (ti - mv3(R, ti)) + mv3(Ri, scale * (tm - mv3(RmF, np.asarray(tfk, np.float64))))
    return R, t


def attached_frame(scale: float, Ri: np.ndarray, ti: np.ndarray, q_link: Sequence[float], p_link: Sequence[float],
                   local_xyz: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """Camera pose attached to a link (splat_handler.py:316-332).  As in the reference, the local
    offset is ADDED to the link position in world axes (not rotated by the link) and then scaled."""
    p = (np.asarray(p_link, np.float64) + np.asarray(local_xyz, np.float64)) * scale
    return mm3(Ri, quat_wxyz_to_matrix(q_link)), mv3(Ri, p) + np.asarray(ti, np.float64)


def rt_to_row12(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    out = np.empty((3, 4), dtype=np.float32)
    out[:, :3] = R
    out[:, 3] = t
    return out.reshape(12)

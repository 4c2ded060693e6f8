"""Forward kinematics of a URDF's visual meshes at the mask-time joint configuration.

``SplatHandler._add_robot_meshes`` (sim_a_splat/splat/splat_handler.py:147-200) loads the URDF with
urchin, evaluates ``visual_trimesh_fk(cfg=dict(zip(actuated_joint_names, joint_config)))`` and keeps
one SE3 per visual mesh (``fk_tf``, :197); ``draw_handler`` uses them as ``Rfk, tfk`` (:262-278).
This module restates that computation without urchin/trimesh: it parses the URDF XML and returns
``link_pose @ visual.origin [@ diag(mesh scale)]`` for every visual that has a mesh.

Orders follow urchin 0.0.29 / networkx 3.5 (``pixi.lock:268,199``; neither is vendored, restated from
their published source) because the reference indexes both lists by position
(``fk_tf[i]``, ``dict(zip(actuated_joint_names, joint_config))``):

* links: urchin builds a DiGraph with the links as nodes (file order) and one edge child -> parent per
  joint, and walks ``reversed(list(nx.topological_sort(G)))``; networkx emits topological *generations*
  (Kahn's algorithm, nodes of a generation in insertion order), so the order is by height above the
  leaves, not by depth below the base -- it coincides with file order for a serial chain such as the
  xarm6 and differs for branching robots (grippers) and links declared out of order;
* actuated joints: sorted by the number of links on the child's path to the base (``_sort_joints``,
  ``np.argsort`` of small arrays: ties keep file order);
* one pose per mesh of a visual: mesh files are not opened here, so a visual counts as ONE mesh
  (multi-mesh files, e.g. a .dae scene, would need the file; ``meshes_per_visual`` lets a caller say so).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

ACTUATED = ("revolute", "continuous", "prismatic")


def _floats(text: Optional[str], n: int, default: Sequence[float]) -> np.ndarray:
    if text is None:
        return np.asarray(default, dtype=np.float64)
    v = np.asarray([float(x) for x in text.split()], dtype=np.float64)
    if v.shape != (n,):
        raise ValueError(f"expected {n} numbers, got {text!r}")
    return v


def rpy_matrix(rpy) -> np.ndarray:
    """URDF fixed-axis roll/pitch/yaw: R = Rz(yaw) @ Ry(pitch) @ Rx(roll)."""
    r, p, y = (float(a) for a in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin(el: Optional[ET.Element]) -> np.ndarray:
    T = np.eye(4)
    if el is not None:
        T[:3, :3] = rpy_matrix(_floats(el.get("rpy"), 3, (0, 0, 0)))
        T[:3, 3] = _floats(el.get("xyz"), 3, (0, 0, 0))
    return T


def axis_angle_matrix(axis, angle: float) -> np.ndarray:
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * Kx + (1.0 - np.cos(angle)) * (Kx @ Kx)


@dataclass
class Joint:
    name: str
    type: str
    parent: str
    child: str
    origin: np.ndarray
    axis: np.ndarray
    mimic: Optional[Tuple[str, float, float]] = None   # (joint, multiplier, offset)

    def motion(self, q: float) -> np.ndarray:
        T = np.eye(4)
        if self.type in ("revolute", "continuous"):
            T[:3, :3] = axis_angle_matrix(self.axis, q)
        elif self.type == "prismatic":
            T[:3, 3] = self.axis / np.linalg.norm(self.axis) * q
        return T


@dataclass
class Visual:
    origin: np.ndarray
    mesh: Optional[str]
    scale: Optional[np.ndarray]


@dataclass
class Robot:
    links: List[str]
    visuals: Dict[str, List[Visual]]
    joints: List[Joint]
    base: str = ""
    actuated_joints: List[Joint] = field(default_factory=list)
    fk_link_order: List[str] = field(default_factory=list)   # urchin's link_fk / visual_trimesh_fk order

    @property
    def actuated_joint_names(self) -> List[str]:
        return [j.name for j in self.actuated_joints]


def load(urdf: Union[str, Path]) -> Robot:
    """Parse a URDF file (or XML text)."""
    text = str(urdf)
    root = ET.fromstring(text) if text.lstrip().startswith("<") else ET.parse(text).getroot()
    links, visuals = [], {}
    for le in root.findall("link"):
        name = le.get("name")
        links.append(name)
        vs = []
        for ve in le.findall("visual"):
            me = ve.find("geometry/mesh")
            scale = _floats(me.get("scale"), 3, (1, 1, 1)) if me is not None and me.get("scale") else None
            vs.append(Visual(_origin(ve.find("origin")), me.get("filename") if me is not None else None, scale))
        visuals[name] = vs
    joints = []
    for je in root.findall("joint"):
        ax = je.find("axis")
        mm = je.find("mimic")
        joints.append(Joint(je.get("name"), je.get("type"), je.find("parent").get("link"), je.find("child").get("link"),
                            _origin(je.find("origin")), _floats(ax.get("xyz") if ax is not None else None, 3, (1, 0, 0)),
                            (mm.get("joint"), float(mm.get("multiplier", 1.0)), float(mm.get("offset", 0.0))) if mm is not None else None))
    children = {j.child for j in joints}
    bases = [l for l in links if l not in children]
    if len(bases) != 1:
        raise ValueError(f"URDF must have exactly one base link, found {bases}")
    rb = Robot(links, visuals, joints, base=bases[0])
    parent_joint = {j.child: j for j in joints}
    if len(parent_joint) != len(joints):
        raise ValueError("a link is the child of two joints")
    depth: Dict[str, int] = {}
    for l in links:                               # links on the path to the base, the link itself included
        d, cur, seen = 1, l, set()
        while cur != rb.base:
            if cur in seen or cur not in parent_joint:
                raise ValueError("URDF joints do not form a tree rooted at the base link")
            seen.add(cur)
            cur = parent_joint[cur].parent
            d += 1
        depth[l] = d
    # evaluation order for link_fk: any base-outwards order (parents before children)
    rb.joints = [j for _, _, j in sorted(((depth[j.child], k, j) for k, j in enumerate(joints)), key=lambda t: t[:2])]
    # urchin: actuated joints in file order, stably sorted by the child's path length to the base
    rb.actuated_joints = [j for j in rb.joints if j.type in ACTUATED and j.mimic is None]
    # urchin / networkx: reversed topological generations of the child -> parent graph
    indeg = {l: 0 for l in links}
    for j in joints:
        indeg[j.parent] += 1
    zero, topo = [l for l in links if indeg[l] == 0], []
    while zero:
        gen, zero = zero, []
        for l in gen:
            topo.append(l)
            if l in parent_joint:
                p = parent_joint[l].parent
                indeg[p] -= 1
                if indeg[p] == 0:
                    zero.append(p)
    rb.fk_link_order = topo[::-1]
    return rb


def link_fk(robot: Robot, cfg: Union[Dict[str, float], Sequence[float], None] = None) -> Dict[str, np.ndarray]:
    """World pose of every link.  ``cfg``: joint name -> position, or positions in
    ``actuated_joint_names`` order (shorter sequences leave the remaining joints at 0, as
    ``dict(zip(names, joint_config))`` does in the reference)."""
    if cfg is None:
        cfg = {}
    if not isinstance(cfg, dict):
        cfg = dict(zip(robot.actuated_joint_names, [float(x) for x in np.asarray(cfg).reshape(-1)]))
    unknown = set(cfg) - {j.name for j in robot.joints}
    if unknown:
        raise KeyError(f"unknown joints {sorted(unknown)}")
    poses = {robot.base: np.eye(4)}
    for j in robot.joints:                       # base-outwards: the parent is always done
        q = cfg.get(j.name, 0.0)
        if j.mimic is not None:
            q = cfg.get(j.mimic[0], 0.0) * j.mimic[1] + j.mimic[2]
        poses[j.child] = poses[j.parent] @ j.origin @ j.motion(q)
    return poses


def visual_mesh_fk(robot: Robot, cfg=None, meshes_per_visual: Optional[Dict[str, int]] = None) -> List[np.ndarray]:
    """One 4x4 per visual mesh in urchin's order (the ``fk_tf`` list of splat_handler.py:197): links in
    ``robot.fk_link_order``, a link's visuals in file order.  ``meshes_per_visual`` maps a mesh filename
    to the number of meshes the file holds (urchin emits one entry per mesh; default 1)."""
    poses = link_fk(robot, cfg)
    out = []
    for l in robot.fk_link_order or robot.links:
        for v in robot.visuals[l]:
            if v.mesh is None:
                continue
            T = poses[l] @ v.origin
            if v.scale is not None:
                S = np.eye(4)
                S[:3, :3] = np.diag(v.scale)
                T = T @ S
            out.extend([T] * int((meshes_per_visual or {}).get(v.mesh, 1)))
    return out

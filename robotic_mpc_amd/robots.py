"""Kinematic-chain constants for the batched MPC engine.

Replaces the reference's ``UrdfLoader`` (loader.py:5-72, a Pinocchio wrapper) for the
hot path: the engine only needs, per revolute joint, the fixed placement of the joint
frame in its parent and the joint axis, plus the fixed placement of the end-effector
frame (``ee_link`` for ur10, ``tool0`` for ur5 -- loader.py:33-36).  Fixed joints on
the way are folded into the neighbouring placements, exactly what Pinocchio's URDF
parser does when it builds ``oMf``.

The literal numbers below are the ``<origin>``/``<axis>`` attributes of
ur_description/urdf/ur10.urdf and ur5.urdf; they are quoted verbatim (for instance
``1.570796325`` is *not* pi/2) because a 1.8e-9 rad difference is visible at the
1e-12 level the kinematics tests run at.
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np


def rpy_to_matrix(rpy: Sequence[float]) -> np.ndarray:
    """URDF fixed-axis roll-pitch-yaw: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ],
        dtype=np.float64,
    )


@dataclass
class KinematicChain:
    """Six revolute joints + end-effector frame, in the layout the C-ABI expects.

    ``place[i] = [R row-major (9), p (3)]``: placement of joint i's frame in the frame of
    joint i-1 (world for i=0) at q=0; ``place[6]``: end-effector frame in joint 5's frame.
    """

    name: str
    ee_frame: str
    place: np.ndarray = field(default_factory=lambda: np.zeros((7, 12)))
    axis: np.ndarray = field(default_factory=lambda: np.zeros((6, 3)))
    joint_names: Tuple[str, ...] = ()

    def packed(self, t_ee: Sequence[float] = (0.0, 0.0, 0.1)) -> np.ndarray:
        """Flat fp64 vector [place(84); axis(18); t_ee(3)] = 105 doubles (mpcb_problem.robot)."""
        return np.concatenate([self.place.ravel(), self.axis.ravel(), np.asarray(t_ee, dtype=np.float64)])


# (joint name, type, xyz, rpy, axis) from the chain root to the end-effector frame.
_UR10 = [  # ur_description/urdf/ur10.urdf
    ("world_joint", "fixed", (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), None),                          # :286-290
    ("shoulder_pan_joint", "revolute", (0.0, 0.0, 0.1273), (0.0, 0.0, 0.0), (0, 0, 1)),          # :50-54
    ("shoulder_lift_joint", "revolute", (0.0, 0.220941, 0.0), (0.0, 1.570796325, 0.0), (0, 1, 0)),  # :78-82
    ("elbow_joint", "revolute", (0.0, -0.1719, 0.612), (0.0, 0.0, 0.0), (0, 1, 0)),              # :106-110
    ("wrist_1_joint", "revolute", (0.0, 0.0, 0.5723), (0.0, 1.570796325, 0.0), (0, 1, 0)),       # :134-138
    ("wrist_2_joint", "revolute", (0.0, 0.1149, 0.0), (0.0, 0.0, 0.0), (0, 0, 1)),               # :162-166
    ("wrist_3_joint", "revolute", (0.0, 0.0, 0.1157), (0.0, 0.0, 0.0), (0, 1, 0)),               # :190-194
    ("ee_fixed_joint", "fixed", (0.0, 0.0922, 0.0), (-1.5707963267948966, 0.0, 0.0), None),      # :218-221
]
_UR5 = [  # ur_description/urdf/ur5.urdf
    ("base_link-base_link_inertia", "fixed", (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), None),             # :274-283
    ("shoulder_pan_joint", "revolute", (0.0, 0.0, 0.089159), (0.0, 0.0, 3.141592653589793), (0, 0, 1)),  # :284-288
    ("shoulder_lift_joint", "revolute", (0.0, 0.0, 0.0), (1.570796327, 0.0, 0.0), (0, 0, 1)),    # :292-296
    ("elbow_joint", "revolute", (-0.425, 0.0, 0.0), (0.0, 0.0, 0.0), (0, 0, 1)),                 # :300-304
    ("wrist_1_joint", "revolute", (-0.39225, 0.0, 0.10915), (0.0, 0.0, 0.0), (0, 0, 1)),         # :308-312
    ("wrist_2_joint", "revolute", (0.0, -0.09465, -1.941303950897609e-11), (1.570796327, 0.0, 0.0), (0, 0, 1)),  # :316-320
    ("wrist_3_joint", "revolute", (0.0, 0.0823, -1.688001216681175e-11),
     (1.570796326589793, 3.141592653589793, 3.141592653589793), (0, 0, 1)),                      # :324-328
    ("wrist_3-flange", "fixed", (0.0, 0.0, 0.0), (0.0, -1.5707963267948966, -1.5707963267948966), None),  # :346-350
    ("flange-tool0", "fixed", (0.0, 0.0, 0.0), (1.5707963267948966, 0.0, 1.5707963267948966), None),      # :354-359
]

_EE_FRAME = {"ur10": "ee_link", "ur5": "tool0"}  # loader.py:33-36
_BUILTIN = {"ur10": _UR10, "ur5": _UR5}


def _fold(name: str, ee_frame: str, joints) -> KinematicChain:
    """Compose fixed joints into the placements of the six revolute joints."""
    R = np.eye(3)
    p = np.zeros(3)
    place: List[np.ndarray] = []
    axes: List[np.ndarray] = []
    names: List[str] = []
    for jname, jtype, xyz, rpy, axis in joints:
        Rj = rpy_to_matrix(rpy)
        p = p + R @ np.asarray(xyz, dtype=np.float64)
        R = R @ Rj
        if jtype in ("revolute", "continuous"):
            place.append(np.concatenate([R.ravel(), p]))
            a = np.asarray(axis, dtype=np.float64)
            axes.append(a / np.linalg.norm(a))
            names.append(jname)
            R = np.eye(3)
            p = np.zeros(3)
        elif jtype != "fixed":
            raise ValueError(f"joint '{jname}': type '{jtype}' is not supported (6 revolute joints only)")
    if len(place) != 6:
        raise ValueError(f"robot '{name}': expected 6 revolute joints on the chain, found {len(place)}")
    place.append(np.concatenate([R.ravel(), p]))
    return KinematicChain(name=name, ee_frame=ee_frame, place=np.array(place), axis=np.array(axes),
                          joint_names=tuple(names))


def builtin_chain(robot_name: str) -> KinematicChain:
    """Chain for ``robot_name`` in {'ur10','ur5'} (the two robots loader.py:33-36 knows)."""
    if robot_name not in _BUILTIN:
        raise ValueError(f"unknown robot_name '{robot_name}'; known: {sorted(_BUILTIN)} "
                         f"(use chain_from_urdf for another 6-DoF URDF)")
    return _fold(robot_name, _EE_FRAME[robot_name], _BUILTIN[robot_name])


def _floats(s: Optional[str], n: int) -> Tuple[float, ...]:
    if s is None:
        return tuple([0.0] * n)
    v = tuple(float(t) for t in s.split())
    if len(v) != n:
        raise ValueError(f"expected {n} numbers, got '{s}'")
    return v


def chain_from_urdf(path: str, ee_frame: str, name: Optional[str] = None) -> KinematicChain:
    """Tiny URDF reader: walks parent links from ``ee_frame`` up to the root."""
    root = ET.parse(path).getroot()
    by_child = {}
    for j in root.findall("joint"):
        if j.find("parent") is None or j.find("child") is None:
            continue  # <transmission><joint .../> entries
        by_child[j.find("child").get("link")] = j
    chain = []
    link = ee_frame
    while link in by_child:
        j = by_child[link]
        o = j.find("origin")
        xyz = _floats(o.get("xyz") if o is not None else None, 3)
        rpy = _floats(o.get("rpy") if o is not None else None, 3)
        ax = j.find("axis")
        axis = _floats(ax.get("xyz"), 3) if ax is not None else (1.0, 0.0, 0.0)
        chain.append((j.get("name"), j.get("type"), xyz, rpy, axis if j.get("type") != "fixed" else None))
        link = j.find("parent").get("link")
    if not chain:
        raise ValueError(f"frame '{ee_frame}' is not the child link of any joint in {path}")
    chain.reverse()
    return _fold(name or root.get("name", "robot"), ee_frame, chain)

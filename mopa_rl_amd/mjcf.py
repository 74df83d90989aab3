"""MJCF-subset scene compiler: XML model -> flat kinematic/collision arrays.

The reference loads its scene with MuJoCo's own XML compiler
(`mj_loadXML`, reference motion_planners/include/mujoco_wrapper.h:75 and
motion_planners/KinematicPlanner.cpp:72).  MuJoCo is not available to this
build, so this module re-implements the part of the MJCF compile step the
state-validity path needs:

  * ``<include>`` expansion (paths relative to the top-level model file),
  * nested ``<default class=...>`` inheritance + ``childclass`` scoping,
  * the body tree in depth-first document order (MuJoCo's body/geom/joint id
    order), hinge / slide / free joints with their qpos addresses,
  * primitive geoms (plane, sphere, capsule, cylinder, box; ``fromto``),
    ``contype`` / ``conaffinity``, ``<contact><exclude>``,
  * the *static* candidate collision-pair list after MuJoCo's filters
    (same body, same weld group, parent-child weld groups unless the parent
    group is the world, excludes, ``(ct1&ca2)|(ct2&ca1)``) -- see
    SURVEY.md Appendix D [3P].

Everything here is plain Python + numpy; nothing is on the hot path.  The
result is a :class:`CompiledModel`, serialisable to a small JSON file
(``mopa_rl_amd/scenes/*.json``) so that the GPU box -- which has no access to
the reference's asset directory -- loads the same numbers.
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# MuJoCo enum values (mjtGeom / mjtJoint) -- kept so geom/joint type codes mean
# the same thing on both sides of the boundary.
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = range(4)

_GEOM_TYPES = {
    "plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
    "ellipsoid": GEOM_ELLIPSOID, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH,
}
_JNT_TYPES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
GEOM_TYPE_NAMES = {v: k for k, v in _GEOM_TYPES.items()}

SUPPORTED_COLLISION_TYPES = (GEOM_PLANE, GEOM_SPHERE, GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX)


class MjcfError(ValueError):
    pass


def _friction3(txt):
    v = _floats(txt)
    return (v + [1.0, 0.005, 0.0001][len(v):])[:3]


def _floats(s: str) -> List[float]:
    return [float(x) for x in s.replace(",", " ").split()]


def _quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def _quat_rotate(q, v):
    """rotate v by the unit quaternion q (wxyz)"""
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return R @ np.asarray(v, dtype=np.float64)


def _normalize_quat(q):
    q = np.asarray(q, dtype=np.float64)
    n = math.sqrt(float(q @ q))
    if n < 1e-15:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return q / n


def _euler_to_quat(e, seq: str, to_rad: float):
    """MuJoCo `eulerseq`: lower-case = intrinsic (rotating-frame) axes."""
    q = np.array([1.0, 0.0, 0.0, 0.0])
    for ang, ax in zip(e, seq):
        h = 0.5 * ang * to_rad
        r = np.array([math.cos(h), 0.0, 0.0, 0.0])
        r["xyz".index(ax.lower()) + 1] = math.sin(h)
        q = _quat_mul(q, r) if ax.islower() else _quat_mul(r, q)
    return q


def _z_to_vec_quat(v):
    """Quaternion rotating +z onto unit vector ``v`` (MuJoCo's fromto rule)."""
    v = np.asarray(v, dtype=np.float64)
    v = v / np.linalg.norm(v)
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, v)
    s = np.linalg.norm(axis)
    c = float(z @ v)
    if s < 1e-10:
        # parallel / anti-parallel
        return np.array([1.0, 0.0, 0.0, 0.0]) if c > 0 else np.array([0.0, 1.0, 0.0, 0.0])
    axis = axis / s
    ang = math.atan2(s, c)
    return np.array([math.cos(ang / 2), *(axis * math.sin(ang / 2))])


@dataclass
class CompiledModel:
    """Flat arrays describing kinematics + collision geometry of one scene."""
    name: str
    nq: int
    # bodies (index 0 = world)
    body_names: List[str]
    body_parent: np.ndarray      # [nbody] i32
    body_pos: np.ndarray         # [nbody,3]
    body_quat: np.ndarray        # [nbody,4] normalised wxyz
    body_jntadr: np.ndarray      # [nbody] first joint id or -1
    body_jntnum: np.ndarray      # [nbody]
    body_weldid: np.ndarray      # [nbody]
    # joints
    jnt_names: List[str]
    jnt_type: np.ndarray         # [njnt]
    jnt_qposadr: np.ndarray      # [njnt]
    jnt_body: np.ndarray         # [njnt]
    jnt_axis: np.ndarray         # [njnt,3] normalised
    jnt_pos: np.ndarray          # [njnt,3]
    jnt_ref: np.ndarray          # [njnt] qpos0 for hinge/slide
    jnt_limited: np.ndarray      # [njnt]
    jnt_range: np.ndarray        # [njnt,2]
    qpos0: np.ndarray            # [nq]
    # all geoms (MuJoCo id order) -- names only, for name2id parity
    all_geom_names: List[str]
    all_geom_body: np.ndarray    # [ngeom_all]
    # collidable geoms (contype|conaffinity != 0)
    geom_mjid: np.ndarray        # [ngeom] id among all geoms
    geom_type: np.ndarray        # [ngeom]
    geom_body: np.ndarray        # [ngeom]
    geom_size: np.ndarray        # [ngeom,3]
    geom_pos: np.ndarray         # [ngeom,3]
    geom_quat: np.ndarray        # [ngeom,4]
    geom_contype: np.ndarray
    geom_conaffinity: np.ndarray
    geom_margin: np.ndarray
    geom_mesh: List[str]         # mesh asset name or ""
    # candidate pairs after static filters: indices into the collidable table,
    # ordered (type1<=type2, then id) as MuJoCo orders geom1/geom2 of a contact
    pair_geom: np.ndarray        # [npair,2]
    # sites
    site_names: List[str]
    site_body: np.ndarray
    site_pos: np.ndarray
    site_quat: np.ndarray
    # convex hulls of the meshes collidable geoms use (MuJoCo collides mesh geoms through their hull):
    # vertices in the geom frame, already re-centred at the mesh's volume centroid (see _load_mesh)
    geom_dataid: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))      # [ngeom] mesh id or -1
    mesh_vertadr: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))     # [nmesh]
    mesh_vertnum: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))     # [nmesh]
    mesh_vert: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), dtype=np.float64)) # [nmeshvert,3]
    # joint actuators (<actuator><position|motor|velocity joint=...>), in document order = MuJoCo's ctrl order
    act_names: List[str] = field(default_factory=list)
    act_joint: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))        # [nu] joint id
    act_ctrllimited: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))  # [nu]
    act_ctrlrange: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), dtype=np.float64))
    # ---- dynamics (servo dynamics of SURVEY.md 8 f4b; absent in scenes compiled before they were kept) ----
    # body inertials as MuJoCo's compiler derives them (explicit <inertial>, else summed over ALL the body's geoms at
    # their density / mass -- `inertiafromgeom="auto"`), kept as COM + full tensor about the COM in the body frame
    # (MuJoCo stores the same thing as a principal frame `body_iquat` + `body_inertia` diagonal)
    body_mass: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))       # [nbody]
    body_ipos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), dtype=np.float64))  # [nbody,3]
    body_inertia: np.ndarray = field(default_factory=lambda: np.zeros((0, 6), dtype=np.float64))  # [nbody,6] xx yy zz xy xz yz
    jnt_damping: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))     # [njnt]
    jnt_armature: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))    # [njnt]
    jnt_stiffness: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))   # [njnt]
    act_kind: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))          # [nu] 0 motor, 1 position, 2 velocity
    act_gain: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))        # [nu] kp / kv / 1
    act_gear: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))        # [nu] gear[0]
    act_forcelimited: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))  # [nu]
    act_forcerange: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), dtype=np.float64))
    opt: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))             # [4] gravity xyz, timestep
    geom_friction: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.float64))   # [ngeom] sliding friction of the collidable geoms
    geom_solref: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), dtype=np.float64))  # [ngeom,2] (timeconst, dampratio)
    geom_solimp: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), dtype=np.float64))  # [ngeom,3] (d0, dmax, width)
    geom_condim: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int32))         # [ngeom]
    geom_friction3: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), dtype=np.float64))   # [ngeom,3] sliding, torsional, rolling friction
    meta: Dict[str, object] = field(default_factory=dict)

    # ---- name lookups mirroring the mujoco-py calls the reference makes ----
    def geom_name2id(self, name: str) -> int:
        return self.all_geom_names.index(name)

    def body_name2id(self, name: str) -> int:
        return self.body_names.index(name)

    def joint_name2id(self, name: str) -> int:
        return self.jnt_names.index(name)

    def site_name2id(self, name: str) -> int:
        return self.site_names.index(name)

    def get_joint_qpos_addr(self, name: str) -> int:
        return int(self.jnt_qposadr[self.joint_name2id(name)])

    def geoms_of_bodies(self, body_names: List[str]) -> List[int]:
        """MuJoCo geom ids of every geom attached to the named bodies
        (reference env/base.py `static_geom_ids`)."""
        ids = [self.body_name2id(b) for b in body_names]
        return [g for g, b in enumerate(self.all_geom_body) if int(b) in ids]

    # ---- serialisation ----
    _ARRAYS = (
        "body_parent body_pos body_quat body_jntadr body_jntnum body_weldid jnt_type jnt_qposadr jnt_body "
        "jnt_axis jnt_pos jnt_ref jnt_limited jnt_range qpos0 all_geom_body geom_mjid geom_type geom_body "
        "geom_size geom_pos geom_quat geom_contype geom_conaffinity geom_margin pair_geom site_body site_pos "
        "site_quat"
    ).split()
    _OPTIONAL = "geom_dataid mesh_vertadr mesh_vertnum mesh_vert".split()   # absent in scenes without collidable meshes
    _ACT = "act_joint act_ctrllimited act_ctrlrange".split()                  # absent in scenes compiled before actuators were kept
    _DYN = ("body_mass body_ipos body_inertia jnt_damping jnt_armature jnt_stiffness act_kind act_gain act_gear act_forcelimited "
            "act_forcerange opt geom_friction").split()                                   # absent in scenes compiled before round 3
    _CT = "geom_solref geom_solimp geom_condim".split()                                  # absent in scenes compiled before round 4
    _CT6 = "geom_friction3".split()                                                      # absent in scenes compiled before round 6
    _LISTS = "body_names jnt_names all_geom_names geom_mesh site_names".split()

    def to_json(self) -> str:
        d = {"name": self.name, "nq": self.nq, "meta": self.meta}
        for k in self._LISTS:
            d[k] = getattr(self, k)
        d["act_names"] = list(self.act_names)
        for k in self._ARRAYS + ([k for k in self._OPTIONAL] if len(self.mesh_vertnum) else []) + self._ACT + \
                (self._DYN if len(self.body_mass) else []) + (self._CT if len(self.geom_solref) else []) + \
                (self._CT6 if len(self.geom_friction3) else []):
            a = getattr(self, k)
            d[k] = {"dtype": str(a.dtype), "shape": list(a.shape),
                    "data": [float(x).hex() if a.dtype.kind == "f" else int(x) for x in a.ravel()]}
        return json.dumps(d, indent=None, separators=(",", ":"))

    @classmethod
    def from_json(cls, s: str) -> "CompiledModel":
        d = json.loads(s)
        kw = {"name": d["name"], "nq": d["nq"], "meta": d.get("meta", {})}
        for k in cls._LISTS:
            kw[k] = list(d[k])
        kw["act_names"] = list(d.get("act_names", []))
        for k in cls._ARRAYS + [k for k in cls._OPTIONAL + cls._ACT + cls._DYN + cls._CT + cls._CT6 if k in d]:
            e = d[k]
            if e["dtype"].startswith("float"):
                a = np.array([float.fromhex(x) for x in e["data"]], dtype=np.float64)
            else:
                a = np.array(e["data"], dtype=np.int32)
            kw[k] = a.reshape(e["shape"])
        m = cls(**kw)
        if len(m.geom_dataid) == 0:
            m.geom_dataid = np.full(len(m.geom_type), -1, dtype=np.int32)
        return m

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            f.write(self.to_json())

    # what the separation proofs in `meta` (tools/prove_separated_pairs.py) were derived from: kinematic tree, joints and their
    # ranges, collidable geoms, candidate pairs.  The proof keys are only honoured while this hash is the one they were stamped with.
    PROOF_KEYS = ("never_violating_pairs", "never_violating_pairs_thr", "pair_cull_radius", "prune_guard_band", "never_within_margin_pairs",
                  "never_violating_pairs_note")

    def geometry_sha256(self) -> str:
        import hashlib
        h = hashlib.sha256()
        for name in ("body_parent", "body_pos", "body_quat", "body_jntadr", "body_jntnum", "jnt_type", "jnt_qposadr", "jnt_body", "jnt_axis",
                     "jnt_pos", "jnt_ref", "jnt_limited", "jnt_range", "geom_type", "geom_body", "geom_size", "geom_pos", "geom_quat",
                     "geom_margin", "pair_geom", "mesh_vert"):
            a = np.ascontiguousarray(getattr(self, name))
            h.update(name.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
        return h.hexdigest()

    def drop_stale_proofs(self) -> bool:
        """remove the proof keys from `meta` if they were stamped for another geometry (or never stamped); True if anything was dropped"""
        if not any(k in self.meta for k in self.PROOF_KEYS):
            return False
        if self.meta.get("proof_geometry_sha256") == self.geometry_sha256():
            return False
        for k in self.PROOF_KEYS + ("proof_geometry_sha256",):
            self.meta.pop(k, None)
        return True

    @classmethod
    def load(cls, path: str) -> "CompiledModel":
        with open(path) as f:
            m = cls.from_json(f.read())
        if m.drop_stale_proofs():
            import warnings
            warnings.warn(f"{path}: the separation proofs in its meta were made for another geometry and are ignored "
                          "(rerun tools/prove_separated_pairs.py)")
        return m


# --------------------------------------------------------------------------
# XML front end
# --------------------------------------------------------------------------
def _expand_includes(elem: ET.Element, base_dir: str) -> None:
    """Replace every <include file=.../> by the children of the included root.
    MuJoCo resolves include paths relative to the top-level model file."""
    i = 0
    while i < len(elem):
        child = elem[i]
        if child.tag == "include":
            path = os.path.join(base_dir, child.attrib["file"])
            sub = ET.parse(path).getroot()
            _expand_includes(sub, base_dir)
            elem.remove(child)
            for k, sc in enumerate(list(sub)):
                elem.insert(i + k, sc)
            i += len(list(sub))
        else:
            _expand_includes(child, base_dir)
            i += 1


class _Defaults:
    """Default-class tree: class name -> {tag -> attrib dict}, with parents."""

    def __init__(self):
        self.classes: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}
        self.parent: Dict[str, Optional[str]] = {"main": None}

    def add_section(self, elem: ET.Element, cls: str = "main") -> None:
        for ch in elem:
            if ch.tag == "default":
                name = ch.attrib.get("class")
                if name is None:
                    raise MjcfError("nested <default> without class")
                if name not in self.classes:
                    self.classes[name] = {}
                    self.parent[name] = cls
                self.add_section(ch, name)
            else:
                self.classes[cls].setdefault(ch.tag, {}).update(ch.attrib)

    def resolve(self, cls: Optional[str], tag: str) -> Dict[str, str]:
        chain = []
        c = cls or "main"
        if c not in self.classes:
            raise MjcfError(f"unknown default class {c!r}")
        while c is not None:
            chain.append(c)
            c = self.parent[c]
        out: Dict[str, str] = {}
        for c in reversed(chain):
            out.update(self.classes[c].get(tag, {}))
        return out


def _load_mesh(path: str, scale) -> Tuple[np.ndarray, np.ndarray]:
    """Binary-STL mesh -> (convex-hull vertices re-centred at the volume centroid, centroid), both in the mesh frame.
    MuJoCo's compiler does the same two things to a mesh asset that matter for collision: geoms collide through the
    convex hull of the vertices, and the mesh is re-centred at its centre of mass (the geom frame moves with it), which
    is the interior point MPR starts from.  (It also re-orients to the principal axes -- irrelevant to the shape.)"""
    from scipy.spatial import ConvexHull
    raw = open(path, "rb").read()
    ntri = int(np.frombuffer(raw[80:84], dtype="<u4")[0])
    if len(raw) != 84 + 50 * ntri:
        raise MjcfError(f"{path}: only binary STL meshes are supported")
    rec = np.frombuffer(raw[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    tri = rec["v"].astype(np.float64) * np.asarray(scale, dtype=np.float64)
    # volume centroid from signed tetrahedra (closed surface)
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)) / 6.0
    centroid = ((a + b + c) / 4.0 * vol[:, None]).sum(0) / vol.sum()
    verts = np.unique(tri.reshape(-1, 3), axis=0)
    hull = ConvexHull(verts)
    hv = verts[np.sort(hull.vertices)] - centroid
    return hv, centroid



def _read_stl(path: str, scale) -> np.ndarray:
    """Binary-STL triangles [ntri, 3, 3] (float64, scaled), mesh frame."""
    raw = open(path, "rb").read()
    ntri = int(np.frombuffer(raw[80:84], dtype="<u4")[0])
    if len(raw) != 84 + 50 * ntri:
        raise MjcfError(f"{path}: only binary STL meshes are supported")
    rec = np.frombuffer(raw[84:], dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    return rec["v"].astype(np.float64) * np.asarray(scale, dtype=np.float64)


def _mesh_mass_props(tri: np.ndarray) -> Tuple[float, np.ndarray, np.ndarray]:
    """(volume, centroid, unit-density inertia tensor about the centroid) of the closed triangle surface `tri`, by the
    divergence theorem over signed tetrahedra (origin, a, b, c).  MuJoCo's compiler derives a mesh geom's mass and
    inertia from its triangles the same way (mass = density * volume)."""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = det.sum() / 6.0
    if not vol > 0:
        raise MjcfError("mesh with non-positive volume (open or inward-facing surface)")
    centroid = ((a + b + c) * det[:, None]).sum(0) / 24.0 / vol
    # second moments  S = integral of x x^T dV  (canonical tetrahedron formula)
    S = np.zeros((3, 3))
    for p, q in ((a, a), (b, b), (c, c)):
        S += np.einsum("i,ij,ik->jk", det, p, q) * 2.0
    for p, q in ((a, b), (a, c), (b, c)):
        S += np.einsum("i,ij,ik->jk", det, p, q) + np.einsum("i,ij,ik->jk", det, q, p)
    S /= 120.0
    S -= vol * np.outer(centroid, centroid)
    inertia = np.trace(S) * np.eye(3) - S
    return float(vol), centroid, inertia


def _shape_mass_props(gtype: int, size) -> Tuple[float, np.ndarray]:
    """(volume, unit-density inertia diagonal about the centre, own axes) of a primitive geom."""
    r = float(size[0])
    if gtype == GEOM_SPHERE:
        v = 4.0 / 3.0 * math.pi * r ** 3
        return v, np.full(3, 0.4 * v * r * r)
    if gtype == GEOM_CAPSULE:
        l = float(size[1])
        vc, vs = math.pi * r * r * 2.0 * l, 4.0 / 3.0 * math.pi * r ** 3
        ixx = vc * (3.0 * r * r + 4.0 * l * l) / 12.0 + vs * (0.4 * r * r + 0.75 * r * l + l * l)
        return vc + vs, np.array([ixx, ixx, vc * r * r / 2.0 + vs * 0.4 * r * r])
    if gtype == GEOM_CYLINDER:
        l = float(size[1])
        v = math.pi * r * r * 2.0 * l
        ixx = v * (3.0 * r * r + 4.0 * l * l) / 12.0
        return v, np.array([ixx, ixx, v * r * r / 2.0])
    if gtype == GEOM_BOX:
        a, b, c = (float(x) for x in size[:3])
        v = 8.0 * a * b * c
        return v, np.array([v * (b * b + c * c) / 3.0, v * (a * a + c * c) / 3.0, v * (a * a + b * b) / 3.0])
    raise MjcfError(f"no mass properties for geom type {GEOM_TYPE_NAMES.get(gtype, gtype)}")


def _quat_to_mat(q) -> np.ndarray:
    return np.stack([_quat_rotate(q, e) for e in np.eye(3)], axis=1)


class _Builder:
    def __init__(self, xml_path: str):
        self.xml_path = xml_path
        root = ET.parse(xml_path).getroot()
        if root.tag != "mujoco":
            raise MjcfError("top-level element must be <mujoco>")
        _expand_includes(root, os.path.dirname(os.path.abspath(xml_path)))
        self.root = root
        self.model_name = root.attrib.get("model", os.path.basename(xml_path))
        comp: Dict[str, str] = {}
        for c in root.findall("compiler"):
            comp.update(c.attrib)
        self.to_rad = 1.0 if comp.get("angle", "degree") == "radian" else math.pi / 180.0
        self.inertiafromgeom = comp.get("inertiafromgeom", "auto")
        optn: Dict[str, str] = {}
        for o in root.findall("option"):
            optn.update(o.attrib)
        self.gravity = np.array(_floats(optn.get("gravity", "0 0 -9.81")), dtype=np.float64)
        self.timestep = float(optn.get("timestep", "0.002"))
        self.eulerseq = comp.get("eulerseq", "xyz")
        self.defaults = _Defaults()
        for d in root.findall("default"):
            self.defaults.add_section(d)
        # mesh assets: name -> (file path, scale); MuJoCo resolves files against meshdir, itself relative to the model file
        self.mesh_assets: Dict[str, Tuple[str, List[float]]] = {}
        mdir = os.path.join(os.path.dirname(os.path.abspath(xml_path)), comp.get("meshdir", ""))
        for a in root.findall("asset"):
            for me in a.findall("mesh"):
                f = me.attrib.get("file", "")
                name = me.attrib.get("name", os.path.splitext(os.path.basename(f))[0])
                self.mesh_assets[name] = (os.path.join(mdir, f), _floats(me.attrib.get("scale", "1 1 1")))
        # accumulators
        self.bodies: List[dict] = []
        self.joints: List[dict] = []
        self.geoms: List[dict] = []
        self.sites: List[dict] = []
        self.nq = 0

    # orientation of a frame-bearing element
    def _orient(self, at: Dict[str, str]) -> np.ndarray:
        if "quat" in at:
            return _normalize_quat(_floats(at["quat"]))
        if "euler" in at:
            return _normalize_quat(_euler_to_quat(_floats(at["euler"]), self.eulerseq, self.to_rad))
        for k in ("axisangle", "xyaxes", "zaxis"):
            if k in at:
                raise MjcfError(f"orientation attribute {k!r} not supported by this MJCF subset")
        return np.array([1.0, 0.0, 0.0, 0.0])

    def _attrs(self, elem: ET.Element, childclass: Optional[str]) -> Dict[str, str]:
        cls = elem.attrib.get("class", childclass)
        at = dict(self.defaults.resolve(cls, elem.tag))
        at.update(elem.attrib)
        return at

    def build(self) -> CompiledModel:
        world = {"name": "world", "parent": 0, "pos": np.zeros(3), "quat": np.array([1.0, 0, 0, 0]),
                 "jntadr": -1, "jntnum": 0, "inertial": None}
        self.bodies.append(world)
        # world-level elements of every <worldbody> section, in document order
        wbs = self.root.findall("worldbody")
        for wb in wbs:
            self._body_contents(wb, 0, None, recurse=False)
        for wb in wbs:
            for ch in wb:
                if ch.tag == "body":
                    self._body(ch, 0, None)
        return self._finish()

    def _body_contents(self, elem: ET.Element, bid: int, childclass: Optional[str], recurse: bool = True) -> None:
        for ch in elem:
            if ch.tag in ("joint", "freejoint"):
                self._joint(ch, bid, childclass)
            elif ch.tag == "geom":
                self._geom(ch, bid, childclass)
            elif ch.tag == "site":
                self._site(ch, bid, childclass)

    def _body(self, elem: ET.Element, parent: int, childclass: Optional[str]) -> None:
        childclass = elem.attrib.get("childclass", childclass)
        bid = len(self.bodies)
        b = {"name": elem.attrib.get("name", f"body{bid}"), "parent": parent,
             "pos": np.array(_floats(elem.attrib.get("pos", "0 0 0")), dtype=np.float64),
             "quat": self._orient(elem.attrib), "jntadr": -1, "jntnum": 0, "inertial": None}
        ine = elem.find("inertial")
        if ine is not None:
            b["inertial"] = dict(ine.attrib)
        self.bodies.append(b)
        self._body_contents(elem, bid, childclass)
        for ch in elem:
            if ch.tag == "body":
                self._body(ch, bid, childclass)

    def _joint(self, elem: ET.Element, bid: int, childclass: Optional[str]) -> None:
        if elem.tag == "freejoint":
            at = dict(elem.attrib)
            jtype = JNT_FREE
        else:
            at = self._attrs(elem, childclass)
            jtype = _JNT_TYPES[at.get("type", "hinge")]
        jid = len(self.joints)
        axis = np.array(_floats(at.get("axis", "0 0 1")), dtype=np.float64)
        n = np.linalg.norm(axis)
        if jtype in (JNT_HINGE, JNT_SLIDE):
            if n < 1e-12:
                raise MjcfError("zero joint axis")
            axis = axis / n
        ref = float(at.get("ref", "0"))
        rng = _floats(at.get("range", "0 0"))
        if jtype == JNT_HINGE:
            ref *= self.to_rad
            rng = [r * self.to_rad for r in rng]
        limited = at.get("limited", "false") == "true"
        j = {"name": at.get("name", f"joint{jid}"), "type": jtype, "qposadr": self.nq, "body": bid,
             "axis": axis, "pos": np.array(_floats(at.get("pos", "0 0 0")), dtype=np.float64),
             "ref": ref, "limited": int(limited), "range": np.array(rng, dtype=np.float64),
             "damping": float(at.get("damping", "0")), "armature": float(at.get("armature", "0")),
             "stiffness": float(at.get("stiffness", "0"))}
        self.joints.append(j)
        body = self.bodies[bid]
        if body["jntnum"] == 0:
            body["jntadr"] = jid
        body["jntnum"] += 1
        self.nq += {JNT_FREE: 7, JNT_BALL: 4, JNT_SLIDE: 1, JNT_HINGE: 1}[jtype]

    def _geom(self, elem: ET.Element, bid: int, childclass: Optional[str]) -> None:
        at = self._attrs(elem, childclass)
        gtype = _GEOM_TYPES[at.get("type", "sphere")]
        size = _floats(at.get("size", "0 0 0"))
        size = (size + [0.0, 0.0, 0.0])[:3]
        pos = np.array(_floats(at.get("pos", "0 0 0")), dtype=np.float64)
        quat = self._orient(at)
        if "fromto" in at:
            ft = np.array(_floats(at["fromto"]), dtype=np.float64)
            a, b = ft[:3], ft[3:]
            pos = 0.5 * (a + b)
            quat = _z_to_vec_quat(b - a)
            half = 0.5 * float(np.linalg.norm(b - a))
            if gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
                size = [size[0], half, 0.0]
            elif gtype in (GEOM_BOX, GEOM_ELLIPSOID):
                size = [size[0], size[0], half]
            else:
                raise MjcfError("fromto on unsupported geom type")
        # canonicalise unused size slots so equal shapes compare equal
        if gtype == GEOM_SPHERE:
            size = [size[0], 0.0, 0.0]
        elif gtype in (GEOM_CAPSULE, GEOM_CYLINDER):
            size = [size[0], size[1], 0.0]
        _si = _floats(at.get("solimp", "0.9 0.95 0.001"))
        if (len(_si) > 3 and abs(_si[3] - 0.5) > 1e-12) or (len(_si) > 4 and abs(_si[4] - 2.0) > 1e-12):
            # the impedance curve restated by the contact stage is MuJoCo's default one (midpoint 0.5, power 2)
            raise ValueError(f"geom {at.get('name', '')!r}: solimp with midpoint != 0.5 or power != 2 is not supported")
        g = {"name": at.get("name", ""), "type": gtype, "body": bid, "size": np.array(size, dtype=np.float64),
             "pos": pos, "quat": quat, "contype": int(at.get("contype", "1")),
             "conaffinity": int(at.get("conaffinity", "1")), "margin": float(at.get("margin", "0")),
             "mesh": at.get("mesh", ""), "density": float(at.get("density", "1000")),
             "friction": _floats(at.get("friction", "1 0.005 0.0001"))[0],
             "friction3": _friction3(at.get("friction", "1 0.005 0.0001")),    # (sliding, torsional, rolling); missing entries = MuJoCo's defaults
             "solref": (_floats(at.get("solref", "0.02 1")) + [1.0])[:2],
             "solimp": (_floats(at.get("solimp", "0.9 0.95 0.001")) + [0.95, 0.001])[:3],   # (d0, dmax, width); midpoint 0.5, power 2
             "condim": int(at.get("condim", "3")),
             "mass": (float(at["mass"]) if "mass" in at else None)}
        self.geoms.append(g)

    def _site(self, elem: ET.Element, bid: int, childclass: Optional[str]) -> None:
        at = self._attrs(elem, childclass)
        self.sites.append({"name": at.get("name", ""), "body": bid,
                           "pos": np.array(_floats(at.get("pos", "0 0 0")), dtype=np.float64),
                           "quat": self._orient(at)})

    def _inertials(self, geoms: List[dict]):
        """Body mass / COM / inertia tensor about the COM (body frame) by MuJoCo's compile rule: an explicit <inertial>
        wins (unless inertiafromgeom="true"); otherwise every geom of the body contributes density * volume (or its
        `mass`) -- visual mesh geoms included, so bodies that carry a mesh and a collision primitive count both."""
        nb = len(self.bodies)
        mass, ipos, inertia = np.zeros(nb), np.zeros((nb, 3)), np.zeros((nb, 6))
        mesh_cache: Dict[str, Tuple[float, np.ndarray, np.ndarray]] = {}
        for b in range(1, nb):
            ine = self.bodies[b]["inertial"]
            if ine is not None and self.inertiafromgeom != "true":
                m = float(ine["mass"])
                R = _quat_to_mat(self._orient(ine))
                if "fullinertia" in ine:
                    xx, yy, zz, xy, xz, yz = _floats(ine["fullinertia"])
                    T = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
                else:
                    T = np.diag(_floats(ine.get("diaginertia", "0 0 0")))
                T = R @ T @ R.T
                c = np.array(_floats(ine.get("pos", "0 0 0")))
            elif self.inertiafromgeom == "false":
                continue
            else:
                parts = []
                for g in geoms:
                    if g["body"] != b or g["type"] == GEOM_PLANE:
                        continue
                    R = _quat_to_mat(g["quat"])
                    if g["type"] == GEOM_MESH:
                        if g["mesh"] not in mesh_cache:
                            path, scale = self.mesh_assets[g["mesh"]]
                            mesh_cache[g["mesh"]] = _mesh_mass_props(_read_stl(path, scale))
                        vol, cen, Ti = mesh_cache[g["mesh"]]
                        gc = g["pos"] + R @ cen
                        Ti = R @ Ti @ R.T
                    else:
                        vol, d = _shape_mass_props(g["type"], g["size"])
                        gc = g["pos"]
                        Ti = R @ np.diag(d) @ R.T
                    gm = g["mass"] if g["mass"] is not None else g["density"] * vol
                    parts.append((gm, gc, Ti * (gm / vol)))
                m = sum(p[0] for p in parts)
                if not parts or m <= 0.0:
                    continue
                c = sum(p[0] * p[1] for p in parts) / m
                T = np.zeros((3, 3))
                for gm, gc, Ti in parts:
                    d = gc - c
                    T += Ti + gm * (float(d @ d) * np.eye(3) - np.outer(d, d))
            mass[b], ipos[b] = m, c
            inertia[b] = [T[0, 0], T[1, 1], T[2, 2], T[0, 1], T[0, 2], T[1, 2]]
        return mass, ipos, inertia

    # ----------------------------------------------------------------------
    def _finish(self) -> CompiledModel:
        # MuJoCo lists geoms/joints/sites in body-id (DFS) order.  Elements of a
        # body were appended while visiting it, except world-level elements
        # which were all appended first -- i.e. already body-ordered.
        order = sorted(range(len(self.geoms)), key=lambda i: (self.geoms[i]["body"], i))
        geoms = [self.geoms[i] for i in order]
        nb = len(self.bodies)
        parent = np.array([b["parent"] for b in self.bodies], dtype=np.int32)
        jntnum = np.array([b["jntnum"] for b in self.bodies], dtype=np.int32)
        weld = np.zeros(nb, dtype=np.int32)
        for b in range(1, nb):
            weld[b] = b if jntnum[b] > 0 else weld[parent[b]]

        excludes = set()
        names = [b["name"] for b in self.bodies]
        for c in self.root.findall("contact"):
            for ex in c.findall("exclude"):
                b1, b2 = names.index(ex.attrib["body1"]), names.index(ex.attrib["body2"])
                excludes.add((min(b1, b2), max(b1, b2)))

        coll = [i for i, g in enumerate(geoms) if g["contype"] != 0 or g["conaffinity"] != 0]
        pairs: List[Tuple[int, int]] = []
        for a in range(len(coll)):
            for b in range(a + 1, len(coll)):
                g1, g2 = geoms[coll[a]], geoms[coll[b]]
                b1, b2 = g1["body"], g2["body"]
                if b1 == b2:
                    continue
                if not ((g1["contype"] & g2["conaffinity"]) or (g2["contype"] & g1["conaffinity"])):
                    continue
                w1, w2 = int(weld[b1]), int(weld[b2])
                if w1 == w2:
                    continue
                pw1, pw2 = int(weld[parent[w1]]), int(weld[parent[w2]])
                if w1 != 0 and w2 != 0 and (w1 == pw2 or w2 == pw1):
                    continue
                if (min(b1, b2), max(b1, b2)) in excludes:
                    continue
                # MuJoCo orders the two geoms of a contact by type, then id
                if g1["type"] > g2["type"]:
                    pairs.append((b, a))
                else:
                    pairs.append((a, b))

        qpos0 = np.zeros(self.nq)
        for j in self.joints:
            a = j["qposadr"]
            if j["type"] == JNT_FREE:
                body = self.bodies[j["body"]]
                qpos0[a:a + 3] = body["pos"]
                qpos0[a + 3:a + 7] = body["quat"]
            elif j["type"] == JNT_BALL:
                qpos0[a:a + 4] = [1, 0, 0, 0]
            else:
                qpos0[a] = j["ref"]

        def arr(lst, key, dt, shape=None):
            a = np.array([x[key] for x in lst], dtype=dt)
            if shape is not None:
                a = a.reshape(shape)
            return a

        cg = [dict(geoms[i]) for i in coll]
        mesh_ids: Dict[str, int] = {}
        mesh_vertadr, mesh_vertnum, mesh_vert = [], [], []
        geom_dataid = np.full(len(cg), -1, dtype=np.int32)
        for gi, g in enumerate(cg):
            if g["type"] != GEOM_MESH:
                continue
            if g["mesh"] not in self.mesh_assets:
                raise MjcfError(f"geom {g['name']!r}: unknown mesh asset {g['mesh']!r}")
            path, scale = self.mesh_assets[g["mesh"]]
            hv, centroid = _load_mesh(path, scale)
            if g["mesh"] not in mesh_ids:
                mesh_ids[g["mesh"]] = len(mesh_vertadr)
                mesh_vertadr.append(sum(mesh_vertnum))
                mesh_vertnum.append(len(hv))
                mesh_vert.append(hv)
            geom_dataid[gi] = mesh_ids[g["mesh"]]
            # the geom frame follows the re-centred mesh; size = half extents of the hull's AABB (as MuJoCo reports it)
            g["pos"] = g["pos"] + _quat_rotate(g["quat"], centroid)
            g["size"] = 0.5 * (hv.max(0) - hv.min(0))
        body_mass, body_ipos, body_inertia = self._inertials(geoms)
        acts = []
        jnames = [j["name"] for j in self.joints]
        for sec in self.root.findall("actuator"):
            for el in sec:
                at = self._attrs(el, None)
                if "joint" not in at:
                    raise MjcfError(f"actuator <{el.tag}> without joint= is not supported by this MJCF subset")
                if el.tag not in ("motor", "position", "velocity"):
                    raise MjcfError(f"actuator <{el.tag}> is not supported by this MJCF subset")
                acts.append({"name": at.get("name", ""), "joint": jnames.index(at["joint"]),
                             "limited": int(at.get("ctrllimited", "false") == "true"),
                             "range": (_floats(at.get("ctrlrange", "0 0")) + [0.0, 0.0])[:2],
                             "kind": ("motor", "position", "velocity").index(el.tag),
                             "gain": float(at.get({"motor": "_", "position": "kp", "velocity": "kv"}[el.tag], "1")),
                             "gear": _floats(at.get("gear", "1"))[0],
                             "flimited": int(at.get("forcelimited", "false") == "true"),
                             "frange": (_floats(at.get("forcerange", "0 0")) + [0.0, 0.0])[:2]})
        nj = len(self.joints)
        jr = np.zeros((nj, 2))
        for i, j in enumerate(self.joints):
            if len(j["range"]) == 2:
                jr[i] = j["range"]
        return CompiledModel(
            name=self.model_name, nq=self.nq,
            body_names=names, body_parent=parent,
            body_pos=arr(self.bodies, "pos", np.float64, (nb, 3)),
            body_quat=arr(self.bodies, "quat", np.float64, (nb, 4)),
            body_jntadr=arr(self.bodies, "jntadr", np.int32), body_jntnum=jntnum, body_weldid=weld,
            jnt_names=[j["name"] for j in self.joints],
            jnt_type=arr(self.joints, "type", np.int32), jnt_qposadr=arr(self.joints, "qposadr", np.int32),
            jnt_body=arr(self.joints, "body", np.int32),
            jnt_axis=arr(self.joints, "axis", np.float64, (nj, 3)),
            jnt_pos=arr(self.joints, "pos", np.float64, (nj, 3)),
            jnt_ref=arr(self.joints, "ref", np.float64), jnt_limited=arr(self.joints, "limited", np.int32),
            jnt_range=jr, qpos0=qpos0,
            all_geom_names=[g["name"] for g in geoms],
            all_geom_body=arr(geoms, "body", np.int32),
            geom_mjid=np.array(coll, dtype=np.int32),
            geom_type=arr(cg, "type", np.int32), geom_body=arr(cg, "body", np.int32),
            geom_size=arr(cg, "size", np.float64, (len(cg), 3)),
            geom_pos=arr(cg, "pos", np.float64, (len(cg), 3)),
            geom_quat=arr(cg, "quat", np.float64, (len(cg), 4)),
            geom_contype=arr(cg, "contype", np.int32), geom_conaffinity=arr(cg, "conaffinity", np.int32),
            geom_margin=arr(cg, "margin", np.float64),
            geom_mesh=[g["mesh"] for g in cg],
            geom_dataid=geom_dataid, mesh_vertadr=np.array(mesh_vertadr, dtype=np.int32),
            mesh_vertnum=np.array(mesh_vertnum, dtype=np.int32),
            mesh_vert=(np.concatenate(mesh_vert) if mesh_vert else np.zeros((0, 3))),
            pair_geom=np.array(pairs, dtype=np.int32).reshape(-1, 2),
            site_names=[s["name"] for s in self.sites],
            site_body=arr(self.sites, "body", np.int32),
            site_pos=arr(self.sites, "pos", np.float64, (len(self.sites), 3)),
            site_quat=arr(self.sites, "quat", np.float64, (len(self.sites), 4)),
            act_names=[a["name"] for a in acts], act_joint=arr(acts, "joint", np.int32),
            act_ctrllimited=arr(acts, "limited", np.int32),
            act_ctrlrange=np.array([a["range"] for a in acts], dtype=np.float64).reshape(-1, 2),
            body_mass=body_mass, body_ipos=body_ipos, body_inertia=body_inertia,
            jnt_damping=arr(self.joints, "damping", np.float64), jnt_armature=arr(self.joints, "armature", np.float64),
            jnt_stiffness=arr(self.joints, "stiffness", np.float64),
            act_kind=arr(acts, "kind", np.int32), act_gain=arr(acts, "gain", np.float64),
            act_gear=arr(acts, "gear", np.float64),
            act_forcelimited=arr(acts, "flimited", np.int32),
            act_forcerange=np.array([a["frange"] for a in acts], dtype=np.float64).reshape(-1, 2),
            opt=np.array([*self.gravity, self.timestep], dtype=np.float64),
            geom_friction=arr(cg, "friction", np.float64),
            geom_solref=np.array([g["solref"] for g in cg], dtype=np.float64).reshape(-1, 2),
            geom_solimp=np.array([g["solimp"] for g in cg], dtype=np.float64).reshape(-1, 3),
            geom_condim=arr(cg, "condim", np.int32),
            geom_friction3=np.array([g["friction3"] for g in cg], dtype=np.float64).reshape(-1, 3),
            meta={"source": os.path.basename(self.xml_path)},
        )


def compile_mjcf(xml_path: str) -> CompiledModel:
    """Compile an MJCF file (subset, see module docstring) to flat arrays."""
    return _Builder(xml_path).build()


def pair_type_histogram(m: CompiledModel) -> Dict[str, int]:
    h: Dict[str, int] = {}
    for a, b in m.pair_geom:
        k = f"{GEOM_TYPE_NAMES[int(m.geom_type[a])]}-{GEOM_TYPE_NAMES[int(m.geom_type[b])]}"
        h[k] = h.get(k, 0) + 1
    return h

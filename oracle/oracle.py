"""ctypes binding of the CPU oracle (oracle/mopa_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under mopa_rl_amd/.
PARITY UNPINNED vs MuJoCo/OMPL (see mopa_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmopa_oracle.so")
ORC_FAR = 1.0e10


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("mopa_oracle.c", "mopa_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmopa_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [
            C.c_int, C.c_int, _ip, _dp, _dp, _ip, _ip,
            C.c_int, _ip, _ip, _dp, _dp, _dp, _ip, _dp,
            C.c_int, _ip, _ip, _ip, _dp, _dp, _dp,
            C.c_int, _ip, C.c_int, _ip, C.c_int, _ip, C.c_double]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_num_active.argtypes = [C.c_void_p]
        L.orc_num_active.restype = C.c_int
        L.orc_active_idx.argtypes = [C.c_void_p, _ip]
        L.orc_sincos.argtypes = [C.c_double, _dp, _dp]
        L.orc_fk.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_fk_bodies.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_geom_dist.restype = C.c_double
        L.orc_geom_dist.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp]
        L.orc_pair_dist.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_is_valid.restype = C.c_int
        L.orc_is_valid.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_is_valid_batch.argtypes = [C.c_void_p, _dp, _dp, C.c_int64, C.c_int64, _u8p, _dp, C.c_int]
        L.orc_check_motion.restype = C.c_int
        L.orc_check_motion.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, C.POINTER(C.c_int64)]
        L.orc_check_motion_batch.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int64, C.c_int64, C.c_double, _u8p, C.c_int]
        L.orc_rng_u64.restype = C.c_uint64
        L.orc_rng_u64.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_rng_uniform.restype = C.c_double
        L.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_plan.restype = C.c_int
        L.orc_plan.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                               _dp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def sincos(x: float) -> Tuple[float, float]:
    s, c = C.c_double(), C.c_double()
    lib().orc_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def geom_dist(t1, size1, pos1, mat1, t2, size2, pos2, mat2) -> float:
    k1 = [_d(np.asarray(x, dtype=np.float64).ravel()) for x in (size1, pos1, mat1)]
    k2 = [_d(np.asarray(x, dtype=np.float64).ravel()) for x in (size2, pos2, mat2)]
    return lib().orc_geom_dist(int(t1), k1[0][1], k1[1][1], k1[2][1], int(t2), k2[0][1], k2[1][1], k2[2][1])


def rng_uniform(seed: int, stream: int, counter: int) -> float:
    return lib().orc_rng_uniform(seed, stream, counter)


class OracleScene:
    """Oracle-side planner object: compiled model + passive/ignored lists + threshold."""

    def __init__(self, model, passive_joint_idx: Sequence[int], ignored_contacts: Sequence[Tuple[int, int]],
                 contact_threshold: float):
        m = model
        self.model = m
        self._keep = []

        def d(a):
            a, p = _d(a); self._keep.append(a); return p

        def i(a):
            a, p = _i(a); self._keep.append(a); return p

        ign = np.asarray(ignored_contacts, dtype=np.int32).reshape(-1, 2)
        pas = np.asarray(passive_joint_idx, dtype=np.int32)
        self._h = lib().orc_scene_create(
            m.nq, len(m.body_names), i(m.body_parent), d(m.body_pos), d(m.body_quat), i(m.body_jntadr), i(m.body_jntnum),
            len(m.jnt_names), i(m.jnt_type), i(m.jnt_qposadr), d(m.jnt_axis), d(m.jnt_pos), d(m.jnt_ref),
            i(m.jnt_limited), d(m.jnt_range),
            len(m.geom_type), i(m.geom_type), i(m.geom_body), i(m.geom_mjid), d(m.geom_size), d(m.geom_pos), d(m.geom_quat),
            len(m.pair_geom), i(m.pair_geom), len(pas), i(pas), len(ign), i(ign), float(contact_threshold))
        self.nq = m.nq
        self.na = lib().orc_num_active(self._h)
        ai = np.zeros(self.na, dtype=np.int32)
        lib().orc_active_idx(self._h, ai.ctypes.data_as(_ip))
        self.active_idx = ai
        self.ngeom = len(m.geom_type)
        self.npair = len(m.pair_geom)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def fk(self, qpos):
        q, qp = _d(qpos)
        gpos = np.zeros((self.ngeom, 3)); gmat = np.zeros((self.ngeom, 9))
        lib().orc_fk(self._h, qp, gpos.ctypes.data_as(_dp), gmat.ctypes.data_as(_dp))
        return gpos, gmat.reshape(-1, 3, 3)

    def fk_bodies(self, qpos):
        q, qp = _d(qpos)
        nb = len(self.model.body_names)
        xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4))
        lib().orc_fk_bodies(self._h, qp, xpos.ctypes.data_as(_dp), xquat.ctypes.data_as(_dp))
        return xpos, xquat

    def pair_dist(self, qpos):
        q, qp = _d(qpos)
        out = np.zeros(self.npair)
        lib().orc_pair_dist(self._h, qp, out.ctypes.data_as(_dp))
        return out

    def is_valid(self, qpos) -> Tuple[bool, float]:
        q, qp = _d(qpos)
        md = C.c_double()
        v = lib().orc_is_valid(self._h, qp, C.byref(md))
        return bool(v), md.value

    def is_valid_batch(self, q_active, qpos_env, samples_per_env: Optional[int] = None, nthreads: int = 1,
                       want_min_dist: bool = True):
        qa, qap = _d(q_active)
        qe, qep = _d(np.atleast_2d(qpos_env))
        N = qa.shape[0]
        assert qa.shape[1] == self.na and qe.shape[1] == self.nq
        spe = samples_per_env if samples_per_env is not None else max(1, N // qe.shape[0])
        assert (N + spe - 1) // spe <= qe.shape[0]
        valid = np.zeros(N, dtype=np.uint8)
        md = np.zeros(N) if want_min_dist else None
        lib().orc_is_valid_batch(self._h, qap, qep, N, spe, valid.ctypes.data_as(_u8p),
                                 md.ctypes.data_as(_dp) if md is not None else None, nthreads)
        return valid, md

    def check_motion(self, qpos_env, qa, qb, resolution: float = 0.005):
        e, ep = _d(qpos_env); a, ap = _d(qa); b, bp = _d(qb)
        n = C.c_int64(0)
        r = lib().orc_check_motion(self._h, ep, ap, bp, resolution, C.byref(n))
        return bool(r), n.value

    def check_motion_batch(self, qa, qb, qpos_env, samples_per_env: Optional[int] = None,
                           resolution: float = 0.005, nthreads: int = 1):
        a, ap = _d(qa); b, bp = _d(qb)
        qe, qep = _d(np.atleast_2d(qpos_env))
        N = a.shape[0]
        spe = samples_per_env if samples_per_env is not None else max(1, N // qe.shape[0])
        valid = np.zeros(N, dtype=np.uint8)
        lib().orc_check_motion_batch(self._h, ap, bp, qep, N, spe, resolution, valid.ctypes.data_as(_u8p), nthreads)
        return valid

    def plan(self, start, goal, range_: float, resolution: float = 0.005, max_iters: int = 2000,
             max_nodes: int = 4096, seed: int = 0, env_id: int = 0, max_path: int = 512):
        s, sp = _d(start); g, gp = _d(goal)
        path = np.zeros((max_path, self.nq))
        plen = C.c_int(0); nchk = C.c_int64(0); nit = C.c_int(0)
        st = lib().orc_plan(self._h, sp, gp, range_, resolution, max_iters, max_nodes, seed, env_id,
                            path.ctypes.data_as(_dp), max_path, C.byref(plen), C.byref(nchk), C.byref(nit))
        return st, path[:plen.value].copy(), nchk.value, nit.value

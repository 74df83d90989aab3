"""Batched damped-least-squares IK -- mirrors reference env/inverse_kinematics.py:18-135
(`qpos_from_site_pose`: position targets, or position + orientation targets as the MoPA+IK rollouts call it,
rl/mopa_rollouts.py:690-705) on libmopa_hip.so (kernel K5 `k_ik_solve`, one lane per env).

    ik = BatchIK(model, site="grip_site", joint_names=[...])
    res = ik.solve(qpos, target_pos, target_quat, max_steps=100, tol=1e-2)   # IKResult of GPU tensors; qpos updated in place

`qpos_from_site_pose(model, qpos, site, target_pos, joint_names, ...)` is the single-problem form with the reference's
argument names and return type (`IKResult(qpos, err_norm, steps, success)`); the reference passes a live env, here the
compiled model and a qpos vector stand in for it.  There is no CPU fallback.
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from .batch import _ptr, _stream_handle, _torch

IKResult = collections.namedtuple("IKResult", ["qpos", "err_norm", "steps", "success"])


class BatchIK:
    def __init__(self, model, site: str, joint_names: Sequence[str], device: int = -1):
        L = _lib.lib()
        self.model = model
        si = model.site_name2id(site)
        self.site_body = int(model.site_body[si])
        self.site_off = np.asarray(model.site_pos[si], dtype=np.float64)
        self.joint_ids = np.array([model.joint_name2id(j) for j in joint_names], dtype=np.int32)
        keep = []
        d = _lib.MopaIkDesc()
        d.model = _lib.model_struct(model, keep)
        ji, jp = _lib._i(self.joint_ids)
        keep.append(ji)
        d.n_joints, d.joint_ids = len(ji), jp
        d.site_body, d.site_off, d.device = self.site_body, (C.c_double * 3)(*self.site_off), int(device)
        d.site_quat = (C.c_double * 4)(*np.asarray(model.site_quat[si], dtype=np.float64))
        h = C.c_void_p()
        _lib.check(L.mopa_ik_create(C.byref(d), C.byref(h)))
        self._h = h
        self.nq = model.nq

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().mopa_ik_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def site_pose(self, qpos, stream=None):
        """World pose of the site for E states: (pos [E,3], mat [E,3,3]) -- `get_site_xpos / get_site_xmat(ik_target)`."""
        torch = _torch()
        E = qpos.shape[0]
        pos = torch.empty(E, 3, dtype=torch.float64, device=qpos.device)
        mat = torch.empty(E, 3, 3, dtype=torch.float64, device=qpos.device)
        _lib.check(_lib.lib().mopa_ik_site_pose_batch(self._h, E, _ptr(qpos), _ptr(pos), _ptr(mat), _stream_handle(stream)))
        return pos, mat

    def targets(self, site_pos, site_mat, ac, action_range: float, world_lo, world_hi, stream=None):
        """The IK problem of the MoPA + IK action space for E envs in ONE launch (`MoPARolloutRunner._cart2dispalcement`,
        rl/mopa_rollouts.py:87-99,681-696): (target_cart [E,3], target_quat [E,4] wxyz) from the site pose and the policy's action rows
        ac [E, >= 7] (Cartesian displacement + rotation quaternion); world_lo / world_hi: 3 floats each."""
        import ctypes as C
        torch = _torch()
        E = site_pos.shape[0]
        if ac.dim() != 2 or ac.shape[0] != E or ac.shape[1] < 7 or ac.stride(1) != 1 or ac.dtype != torch.float64 or not ac.is_cuda:
            raise _lib.MopaError("ac must be a float64 GPU tensor [E, >= 7] with unit column stride")
        for t, name in ((site_pos, "site_pos"), (site_mat, "site_mat")):
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
                raise _lib.MopaError(f"{name} must be a contiguous float64 GPU tensor")
        cart = torch.empty(E, 3, dtype=torch.float64, device=site_pos.device)
        quat = torch.empty(E, 4, dtype=torch.float64, device=site_pos.device)
        lo = (C.c_double * 3)(*[float(x) for x in world_lo])
        hi = (C.c_double * 3)(*[float(x) for x in world_hi])
        _lib.check(_lib.lib().mopa_ik_targets_batch(self._h, E, _ptr(site_pos), _ptr(site_mat), _ptr(ac), int(ac.stride(0)), float(action_range),
                                                    lo, hi, _ptr(cart), _ptr(quat), _stream_handle(stream)))
        return cart, quat

    def solve(self, qpos, target_pos, target_quat=None, max_steps: int = 100, rot_weight: float = 1.0, tol: float = 1e-14,
              max_update_norm: float = 2.0, progress_thresh: float = 20.0, regularization_strength: float = 3e-2, stream=None) -> IKResult:
        """qpos [E, nq] (updated in place), target_pos [E, 3], target_quat [E, 4] wxyz or None: contiguous float64 GPU tensors.
        Defaults are the reference's (inverse_kinematics.py:24-30); the rollouts call it with max_steps=100, tol=1e-2."""
        torch = _torch()
        checks = [(qpos, "qpos", self.nq), (target_pos, "target_pos", 3)] + ([(target_quat, "target_quat", 4)] if target_quat is not None else [])
        for t, name, cols in checks:
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != cols:
                raise _lib.MopaError(f"{name} must be a contiguous float64 GPU tensor of shape [E, {cols}]")
        E = qpos.shape[0]
        if target_pos.shape[0] != E or (target_quat is not None and target_quat.shape[0] != E):
            raise _lib.MopaError("qpos and targets disagree on E")
        # (the kernel writes all three for every row: no fill launches)
        err = torch.empty(E, dtype=torch.float64, device=qpos.device)
        steps = torch.empty(E, dtype=torch.int32, device=qpos.device)
        succ = torch.empty(E, dtype=torch.uint8, device=qpos.device)
        _lib.check(_lib.lib().mopa_ik_solve_batch(self._h, E, _ptr(qpos), _ptr(target_pos),
                                                  _ptr(target_quat) if target_quat is not None else None, float(rot_weight),
                                                  int(max_steps), float(tol), float(max_update_norm), float(progress_thresh),
                                                  float(regularization_strength), _ptr(err), _ptr(steps), _ptr(succ), _stream_handle(stream)))
        return IKResult(qpos=qpos, err_norm=err, steps=steps, success=succ)


def qpos_from_site_pose(model, qpos, site, target_pos=None, target_quat=None, joint_names=None, max_steps=100, rot_weight=1.0,
                        tol=1e-14, max_update_norm=2.0, progress_thresh=20.0, regularization_threshold=0.1,
                        regularization_strength=3e-2) -> IKResult:
    """Single problem, reference argument names (inverse_kinematics.py:18-31)."""
    torch = _torch()
    if target_pos is None:
        if target_quat is None:
            raise ValueError("At least one of `target_pos` or `target_quat` must be specified")
        raise NotImplementedError("an orientation target without a position target is not served (the reference itself fails on "
                                  "it: a 6-row Jacobian against a 3-vector, inverse_kinematics.py:101-116)")
    if joint_names is None:
        raise NotImplementedError("joint_names=None (all dofs) is not served: pass the movable joints")
    ik = BatchIK(model, site, list(joint_names))
    q = torch.tensor(np.asarray(qpos, dtype=np.float64)[None], device="cuda")
    t = torch.tensor(np.asarray(target_pos, dtype=np.float64)[None], device="cuda")
    tq = torch.tensor(np.asarray(target_quat, dtype=np.float64)[None], device="cuda") if target_quat is not None else None
    r = ik.solve(q, t, tq, max_steps, rot_weight, tol, max_update_norm, progress_thresh, regularization_strength)
    return IKResult(qpos=r.qpos[0].cpu().numpy(), err_norm=float(r.err_norm[0]), steps=int(r.steps[0]), success=bool(r.success[0]))

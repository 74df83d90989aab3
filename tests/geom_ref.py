"""Independent numeric reference for signed distances between convex primitives (test helper).

signed_dist(A, B) = max over unit directions n of  n.(cB - cA) - h_A(n) - h_B(-n)
which is the separation distance when the shapes are disjoint and minus the minimum translation
depth when they overlap (h = support function about the centre).  Evaluated by dense direction
sampling + local refinement; accuracy ~1e-7.  Nothing here shares code with oracle/ or the kernels.
"""
import numpy as np
from scipy.optimize import minimize

PLANE, SPHERE, CAPSULE, CYLINDER, BOX, MESH = 0, 2, 3, 5, 6, 7


def support(t, size, mat, n):
    """h(n) for directions n [k,3] (unit)."""
    ax = mat.T  # rows = local axes in world
    if t == SPHERE:
        return np.full(len(n), size[0])
    if t == CAPSULE:
        return size[0] + size[1] * np.abs(n @ ax[2])
    if t == CYLINDER:
        c = n @ ax[2]
        return size[1] * np.abs(c) + size[0] * np.sqrt(np.maximum(0.0, 1.0 - c * c))
    if t == BOX:
        return np.abs(n @ ax[0]) * size[0] + np.abs(n @ ax[1]) * size[1] + np.abs(n @ ax[2]) * size[2]
    if t == MESH:      # `size` holds the hull vertices [k,3] in the mesh frame
        return ((n @ mat) @ np.asarray(size, float).reshape(-1, 3).T).max(axis=1)
    raise ValueError(t)


def _fib(k):
    i = np.arange(k) + 0.5
    phi = np.arccos(1 - 2 * i / k)
    th = np.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)


_DIRS = _fib(40000)


def signed_dist(t1, s1, p1, m1, t2, s2, p2, m2, refine=40):
    s1, p1, m1, s2, p2, m2 = (np.asarray(x, float) for x in (s1, p1, m1, s2, p2, m2))   # MESH: s2 = vertices
    m1, m2 = m1.reshape(3, 3), m2.reshape(3, 3)
    if t1 == PLANE:
        n = m1[:, 2]
        return float(n @ (p2 - p1) - support(t2, s2, m2, -n[None])[0])
    d = p2 - p1

    def sep(n):
        n = np.atleast_2d(n)
        n = n / np.linalg.norm(n, axis=1, keepdims=True)
        return n @ d - support(t1, s1, m1, n) - support(t2, s2, m2, -n)

    v = sep(_DIRS)
    best = -np.inf
    for i in np.argsort(-v)[:refine]:
        r = minimize(lambda x: -sep(x)[0], _DIRS[i], method="Nelder-Mead",
                     options={"xatol": 1e-10, "fatol": 1e-13, "maxiter": 600})
        best = max(best, -r.fun)
    return float(best)


def rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rand_size(rng, t):
    if t == SPHERE:
        return np.array([rng.uniform(0.02, 0.2), 0, 0])
    if t in (CAPSULE, CYLINDER):
        return np.array([rng.uniform(0.02, 0.15), rng.uniform(0.02, 0.3), 0])
    if t == BOX:
        return rng.uniform(0.01, 0.25, 3)
    return np.zeros(3)

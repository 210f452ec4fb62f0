"""Run the slab path with virtual ranks on one GPU and compare with the single-rank run (subprocess of the test)."""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L

pj.init(0)
lib = L.lib()


def run(nranks, n, Lz_factor, centers, steps, scheme_run, nn=None, LL=None, radius=1.0, dt=None):
    N = 3 if nn is None else len(nn)
    nn = np.array([n, n, n * Lz_factor] if nn is None else nn, dtype=np.int64)
    LL = np.array([4.0, 4.0, 4.0 * Lz_factor] if LL is None else LL, dtype=np.float64)
    if len(centers) == 1:
        kind, params = L.PG_BODY_BALL, np.array(list(centers[0]) + [radius])
    else:
        kind, params = L.PG_BODY_MULTIBALL, np.array([1.0, float(len(centers))] + [v for c in centers for v in c])
    M = int(np.prod(nn + 1))
    x = np.zeros(2 * M)
    keys = np.array([L.PG_KEY[k] for k in (("left", "right", "top", "bottom") if N > 1 else ("top", "bottom"))], dtype=np.int32)
    n_own = np.zeros(nranks, dtype=np.int64)
    nnz = np.zeros(nranks, dtype=np.int64)
    ngh = np.zeros(nranks, dtype=np.int64)
    its = np.zeros(nranks, dtype=np.int64)
    dt = 0.75 * (4.0 / n) ** 2 if dt is None else dt
    L.check(lib.pg_debug_run_virtual_ranks(nranks, int(N), L.iptr(nn), L.dptr(LL), kind, L.dptr(params), len(params), C.c_double(1.0),
                                           C.c_double(1.0), len(keys), keys.ctypes.data_as(L.c_i32_p), C.c_double(dt), 0,
                                           scheme_run, C.c_int64(steps), L.dptr(x), L.iptr(n_own), L.iptr(nnz), L.iptr(ngh),
                                           L.iptr(its)))
    return x, n_own, nnz, ngh, its


def loop_info(nranks):
    """per rank of the last run: (rows of the full system, rows / ghost entries of the system the warm loop iterated on)"""
    res = []
    for r in range(nranks):
        full, rows, nbytes, gh = (C.c_int64() for _ in range(4))
        L.check(lib.pg_debug_virtual_rank_info(r, C.byref(full), C.byref(rows), C.byref(nbytes), C.byref(gh)))
        res.append((full.value, rows.value, gh.value))
    return res


out = {}
for case, (n, zf, centers) in {"sphere": (24, 1, [(2.01, 2.01, 2.01)]),
                               "two_spheres": (16, 2, [(2.01, 2.01, 2.01), (2.01, 2.01, 6.01)])}.items():
    ref, n1, nnz1, _, it1 = run(1, n, zf, centers, 3, 1)
    for nr in (2, 3, 4):
        x, n_own, nnz, ngh, its = run(nr, n, zf, centers, 3, 1)
        err = float(np.linalg.norm(x - ref) / np.linalg.norm(ref))
        out[f"{case}_{nr}"] = {"rel_l2": err, "n_own": n_own.tolist(), "n_total_1": int(n1[0]), "nnz": nnz.tolist(),
                               "nnz_total_1": int(nnz1[0]), "n_ghost": ngh.tolist(), "iters": its.tolist(), "iters_1": int(it1[0]),
                               "loop": loop_info(nr)}
# data that change every step (interface value ramp): the rows alone on their diagonal move in every step on every rank, and
# their change reaches the rows coupled to them through the compact loop's coupling block -- across slab faces by one
# exchange of the deltas
L.check(lib.pg_debug_set_virtual_rank_ramp(C.c_double(0.25)))
ref, n1, nnz1, _, it1 = run(1, 24, 1, [(2.01, 2.01, 2.01)], 4, 1)
for nr in (2, 3):
    x, n_own, nnz, ngh, its = run(nr, 24, 1, [(2.01, 2.01, 2.01)], 4, 1)
    out[f"ramp_sphere_{nr}"] = {"rel_l2": float(np.linalg.norm(x - ref) / np.linalg.norm(ref)), "n_own": n_own.tolist(),
                                "n_total_1": int(n1[0]), "nnz": nnz.tolist(), "nnz_total_1": int(nnz1[0]),
                                "n_ghost": ngh.tolist(), "iters": its.tolist(), "iters_1": int(it1[0]), "loop": loop_info(nr),
                                "ref_max": float(np.max(np.abs(ref)))}
L.check(lib.pg_debug_set_virtual_rank_ramp(C.c_double(0.0)))
# random slab problems: anisotropic 2-D / 3-D grids, a ball anywhere in the box (cut by slab faces, touching borders),
# 2-4 ranks, BE / CN -- against the same problem on one rank
rng = np.random.default_rng(77)
for q in range(8):
    N = int(rng.integers(2, 4))
    nn = [int(v) for v in rng.integers(9, 17 if N == 3 else 33, size=N)]
    LL = [float(v) for v in rng.uniform(1.0, 3.0, size=N)]
    c = [LL[d] * float(rng.uniform(0.2, 0.8)) for d in range(N)]
    rad = float(rng.uniform(0.2, 0.45)) * min(LL)
    sch = int(rng.integers(0, 2))
    dtq = 0.5 * min(LL[d] / nn[d] for d in range(N)) ** 2
    kw = dict(nn=nn, LL=LL, radius=rad, dt=dtq)
    ref, n1, nnz1, _, it1 = run(1, 0, 1, [c], 2, sch, **kw)
    nr = int(rng.integers(2, 5))
    x, n_own, nnz, ngh, its = run(nr, 0, 1, [c], 2, sch, **kw)
    out[f"random{q}_{N}d_{nr}"] = {"rel_l2": float(np.linalg.norm(x - ref) / max(np.linalg.norm(ref), 1e-300)), "n_own": n_own.tolist(),
                                   "n_total_1": int(n1[0]), "nnz": nnz.tolist(), "nnz_total_1": int(nnz1[0]),
                                   "n_ghost": ngh.tolist(), "iters": its.tolist(), "iters_1": int(it1[0]), "random": True}
# GMRES(m): the same slab code with the multi-dot all-reduces of pg_gmres.hip
L.check(lib.pg_debug_set_virtual_rank_method(L.PG_METHOD["gmres"], 6))
ref, n1, nnz1, _, it1 = run(1, 24, 1, [(2.01, 2.01, 2.01)], 2, 1)
for nr in (2, 3):
    x, n_own, nnz, ngh, its = run(nr, 24, 1, [(2.01, 2.01, 2.01)], 2, 1)
    out[f"gmres_sphere_{nr}"] = {"rel_l2": float(np.linalg.norm(x - ref) / np.linalg.norm(ref)), "n_own": n_own.tolist(),
                                 "n_total_1": int(n1[0]), "nnz": nnz.tolist(), "nnz_total_1": int(nnz1[0]),
                                 "n_ghost": ngh.tolist(), "iters": its.tolist(), "iters_1": int(it1[0])}
L.check(lib.pg_debug_set_virtual_rank_method(L.PG_METHOD["bicgstab"], 0))
print(json.dumps(out))

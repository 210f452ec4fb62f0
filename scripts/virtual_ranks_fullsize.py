"""Full-size rehearsal of the weak-scaling bench on ONE GPU: `nranks` virtual ranks (host threads, in-process transport)
run the (n, n, n*nranks) grid with one sphere per slab -- the problem `bench.py --gpus nranks` gives to real ranks.
Catches what small grids cannot: 2M > 2^31 padded unknowns at 8 x 512^3, per-rank sizes, partition balance.

    python scripts/virtual_ranks_fullsize.py [nranks=8] [n=512] [steps=3] [strong [state]]

`strong state`: the slab runs also return their owned planes of the state and the assembled field is compared with the
1-rank state (rel. L2 and max-abs difference in the JSON line).
"""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L

nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pj.init(0)
lib = L.lib()


with_state = len(sys.argv) > 5 and sys.argv[5] == "state"


def run(nr, zf):
    nn = np.array([n, n, n * zf], dtype=np.int64)
    LL = np.array([4.0, 4.0, 4.0 * zf])
    centers = [(2.01, 2.01, 2.01 + 4.0 * s) for s in range(zf)]
    if zf == 1:
        kind, params = L.PG_BODY_BALL, np.array(list(centers[0]) + [1.0])
    else:
        kind, params = L.PG_BODY_MULTIBALL, np.array([1.0, float(zf)] + [v for c in centers for v in c])
    keys = np.array([L.PG_KEY[k] for k in ("left", "right", "top", "bottom")], dtype=np.int32)
    n_own, nnz, ngh, its = (np.zeros(nr, dtype=np.int64) for _ in range(4))
    dt = 0.75 * (4.0 / n) ** 2
    x = np.zeros(2 * int(np.prod(nn + 1))) if with_state else None   # every rank writes its owned planes
    t0 = time.time()
    L.check(lib.pg_debug_run_virtual_ranks(nr, 3, L.iptr(nn), L.dptr(LL), kind, L.dptr(params), len(params), C.c_double(1.0),
                                           C.c_double(1.0), len(keys), keys.ctypes.data_as(L.c_i32_p), C.c_double(dt), 0,
                                           1, C.c_int64(steps), L.dptr(x) if with_state else None, L.iptr(n_own), L.iptr(nnz),
                                           L.iptr(ngh), L.iptr(its)))
    out = dict(n_own=n_own.tolist(), nnz=nnz.tolist(), n_ghost=ngh.tolist(), iters=its.tolist(), wall_s=time.time() - t0)
    # what the warm loop of every rank iterated on: the compact system (rows alone on their diagonal solved in the
    # right-hand-side pass, pg_reduce.hip) when loop_rows < n_own
    loop_rows, loop_bytes, loop_ghosts = [], [], []
    for r in range(nr):
        full, rows, nbytes, gh = (C.c_int64() for _ in range(4))
        L.check(lib.pg_debug_virtual_rank_info(r, C.byref(full), C.byref(rows), C.byref(nbytes), C.byref(gh)))
        loop_rows.append(rows.value); loop_bytes.append(nbytes.value); loop_ghosts.append(gh.value)
    out.update(loop_rows=loop_rows, loop_bytes_per_launch=loop_bytes, loop_ghosts=loop_ghosts,
               rows_alone_on_their_diagonal=[int(a - b) for a, b in zip(n_own.tolist(), loop_rows)])
    if with_state:
        states[nr] = x
    return out


states = {}


strong = len(sys.argv) > 4 and sys.argv[4] == "strong"   # the SAME n^3 sphere problem cut into slabs: every rank exchanges halos
one = run(1, 1)
many = run(nranks, 1 if strong else nranks)
if strong:
    ok = (sum(many["n_own"]) == one["n_own"][0] and sum(many["nnz"]) == one["nnz"][0] and len(set(many["iters"])) == 1
          and abs(many["iters"][0] - one["iters"][0]) <= 2 and all(g > 0 for g in many["n_ghost"]))
    res = {"nranks": nranks, "n": n, "strong": True, "one_rank": one, "virtual": many, "slabs_sum_to_one_rank": ok}
    if with_state:
        x1, xn = states[1], states[nranks]
        d = xn - x1
        res["state_rel_l2"] = float(np.linalg.norm(d) / np.linalg.norm(x1))
        res["state_max_abs_diff"] = float(np.max(np.abs(d)))
        res["state_norm"] = float(np.linalg.norm(x1))
        res["state_max"] = float(np.max(x1))
    print(json.dumps(res))
    sys.exit(0 if ok else 1)
# the spheres sit at 2.01 + 4s: the last bits of (z - c) differ from slab to slab, so a handful of cut cells may differ
close = lambda a, b: abs(a - b) <= 1e-3 * b
ok = (all(close(v, one["n_own"][0]) for v in many["n_own"]) and all(close(v, one["nnz"][0]) for v in many["nnz"])
      and len(set(many["iters"])) == 1 and many["iters"][0] <= 1.3 * one["iters"][0] + 2)
print(json.dumps({"nranks": nranks, "n": n, "one_rank": one, "virtual": many, "slabs_match_one_rank": ok}))
sys.exit(0 if ok else 1)

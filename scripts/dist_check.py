"""One slab-decomposed heat problem through the REAL multi-rank transport (RCCL communicators, halo stream, all-reduced
Krylov scalars), for tests/test_gpu_rccl.py.

    python scripts/dist_check.py OUT.npz [n=48]                         1 rank, plain pg_init            (the reference run)
    PG_TEST_RCCL=1 python scripts/dist_check.py OUT.npz [n=48]          1 rank behind a 1-rank RCCL communicator
    python -m torch.distributed.run --nproc-per-node 2 ... scripts/dist_check.py OUT.npz [n=48]   2 ranks, 2 GPUs

Every rank writes OUT.rank<r>.npz: its owned planes of the state after 1 BE + 8 CN steps (the later ones start from an extrapolation of older states whose fit goes through an all-reduce) (zeros elsewhere), iteration
counts and sizes.  The ranks' arrays add up to the global state.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))

import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L

if world > 1:
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="gloo")       # control plane only: the 128-byte RCCL id travels over it
    box = [pj.get_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    pj.init_distributed(local_rank, rank, world, box[0])
elif os.environ.get("PG_TEST_RCCL"):
    pj.init_distributed(local_rank, 0, 1, pj.get_unique_id())
else:
    pj.init(local_rank)

mesh = pj.Mesh((n, n, n), (4.0, 4.0, 4.0))
cap = pj.Capacity(pj.Sphere((2.01, 2.01, 2.01), 1.3), mesh)          # the ball spans the slab faces: every rank exchanges halos
bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
dt = 0.75 * (4.0 / n) ** 2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 1e30, bcb, pj.Dirichlet(1.0), "CN", save_states=False, max_steps=8, reltol=1e-13)
info = s.system_info(3)
np.savez(f"{out}.rank{rank}.npz", x=s.x, iters=int(s.last_run.total_iters), n_own=int(info.n_own), n_ghost=int(info.n_ghost),
         unconverged=int(s.unconverged), degree=int(s.last_run.poly_degree))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print(f"rank {rank}/{world}: n_own {info.n_own} ghosts {info.n_ghost} iters {s.last_run.total_iters}", flush=True)

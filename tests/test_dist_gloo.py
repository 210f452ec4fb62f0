"""world_size-2 gloo test (CPU) of the N>1 ALGORITHM the HIP path implements: 1-D slab partition of the reduced
system by planes (pg_partition_planes), one ghost chunk per neighbour and SpMV, dot products all-reduced, the collective admission test and the iteration of the
Neumann right preconditioner --
restated here with the oracle's matrix in numpy and torch.distributed(gloo) so that it runs without a GPU.
The device implementation of the same plan is verified on the GPU box by tests/test_gpu_virtual_ranks.py."""
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["PG_ROOT"])
    from oracle import penguin_oracle as po
    from oracle.geometry import Ball
    from penguin.jl_amd import _lib as L

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 10
    mesh = po.Mesh((n, n, n), (4.0, 4.0, 4.0))
    cap = po.make_capacity(Ball((2.01, 2.01, 2.01), 1.0), mesh)
    op = po.make_diffusion_ops(cap)
    ph = po.Phase(cap, op, lambda x, y, z, t: 0.0, lambda x, y, z: 1.0)
    keys = ("left", "right", "top", "bottom")
    bcb = po.BorderConditions({k: po.Dirichlet(1.0) for k in keys})
    M = (n + 1) ** 3
    dt = 0.75 * (4.0 / n) ** 2
    s = po.DiffusionUnsteadyMono(ph, bcb, po.Dirichlet(1.0), dt, np.zeros(2 * M), "BE")
    A, b, idx = po.remove_zero_rows_cols(s.A, s.b)
    d = np.abs(A.diagonal()); ds = 1 / np.sqrt(d)
    import scipy.sparse as sp
    A = (sp.diags(ds) @ A @ sp.diags(ds)).tocsr(); b = ds * b
    # ---- slab partition by planes, weights = active unknowns per plane (host logic of the product) ----
    plane = (n + 1) ** 2
    cell = idx % M
    pl = cell // plane
    w = np.bincount(pl, minlength=n + 1).astype(np.int64)
    bounds = np.zeros(world + 1, dtype=np.int64)
    L.check(L.lib().pg_partition_planes(L.iptr(w), C.c_int64(n + 1), C.c_int32(world), L.iptr(bounds)))
    own = np.flatnonzero((pl >= bounds[rank]) & (pl < bounds[rank + 1]))
    Aloc = A[own, :]
    cols = np.unique(Aloc.indices)
    ghost = np.setdiff1d(cols, own)
    # ghosts must live in the single plane adjacent to the slab (one ghost plane per neighbour)
    assert np.all((pl[ghost] == bounds[rank] - 1) | (pl[ghost] == bounds[rank + 1]))
    nloc = len(own)
    lmap = -np.ones(A.shape[0], dtype=np.int64); lmap[own] = np.arange(nloc); lmap[ghost] = nloc + np.arange(len(ghost))
    Al = sp.csr_matrix((Aloc.data, lmap[Aloc.indices], Aloc.indptr), shape=(nloc, nloc + len(ghost)))

    def halo(v):
        full = torch.zeros(A.shape[0], dtype=torch.float64); full[own] = torch.from_numpy(v[:nloc].copy())
        dist.all_reduce(full)                      # test harness shortcut for the neighbour send/recv
        v[nloc:] = full.numpy()[ghost]

    def dot(a, c):
        t = torch.tensor([float(a @ c)], dtype=torch.float64); dist.all_reduce(t); return float(t)

    bl = b[own]
    x = np.zeros(nloc + len(ghost)); r = bl.copy(); rh = r.copy(); p = np.zeros_like(x); v = np.zeros(nloc)
    rho_old = alpha = omega = 1.0; rho = dot(rh, r); bb = rho; its = 0
    while its < 500:
        its += 1
        beta = (rho / rho_old) * (alpha / omega)
        p[:nloc] = r + beta * (p[:nloc] - omega * v); halo(p)
        v = Al @ p
        alpha = rho / dot(rh, v)
        sv = np.zeros_like(x); sv[:nloc] = r - alpha * v; halo(sv)
        t = Al @ sv
        omega = dot(t, sv[:nloc]) / dot(t, t)
        x[:nloc] += alpha * p[:nloc] + omega * sv[:nloc]
        r = sv[:nloc] - omega * t
        rho_old, rho = rho, dot(rh, r)
        if dot(r, r) <= 1e-26 * bb:
            break
    xs, its_s, _ = po.bicgstab_ref(A, b, reltol=1e-13)
    err = np.linalg.norm(x[:nloc] - xs[own]) / np.linalg.norm(xs)
    assert err < 1e-10, err
    assert its < 100          # (the serial oracle also restarts on breakdown, so counts need not be equal)

    # ---- the Neumann-preconditioned slab iteration (pg_krylov.hip default): admission is a COLLECTIVE decision (max of the
    # per-rank Gershgorin radii, identity-row columns left out; here every column's row length is known globally), each
    # half step exchanges two halos (p then u = 2p - Ap) and all-reduces the same dots ----
    rl = np.diff(A.indptr)
    ident = rl == 1
    Aown = A[own, :].tocsr()
    rad = 0.0
    for q in range(nloc):
        a, e = Aown.indptr[q], Aown.indptr[q + 1]
        if e - a <= 1:
            continue
        cj, vj = Aown.indices[a:e], Aown.data[a:e]
        gi = own[q]
        offm = (cj != gi) & ~ident[cj]
        rad = max(rad, abs(1.0 - vj[cj == gi].sum()) + float(np.sum(np.abs(vj[offm]) * ds[gi] / ds[cj[offm]])))
    tr = torch.tensor([rad], dtype=torch.float64); dist.all_reduce(tr, op=dist.ReduceOp.MAX)
    assert float(tr) < 0.95, float(tr)          # same answer on every rank: the preconditioner is admitted

    def vec(owned):
        o = np.zeros(nloc + len(ghost)); o[:nloc] = owned; halo(o); return o

    # (with T0 = 0 the right-hand side lives on identity rows and (r̂, r) collapses after a step: restart with r̂ := r,
    #  the rule of pg_krylov.hip and of the oracle's bicgstab_ref)
    x2 = np.zeros(nloc); r = bl.copy(); rh = r.copy(); p = np.zeros(nloc); v = np.zeros(nloc)
    rho_old = alpha = omega = 1.0; rho = rhat2 = dot(rh, r); its2 = 0; restart = False
    while its2 < 500:
        its2 += 1
        if restart:
            p = r.copy(); restart = False
        else:
            p = r + (rho / rho_old) * (alpha / omega) * (p - omega * v)
        u = 2.0 * p - Al @ vec(p)
        v = Al @ vec(u)
        den = dot(rh, v)
        force = den == 0.0
        alpha = 0.0 if force else rho / den
        sv = r - alpha * v
        us = 2.0 * sv - Al @ vec(sv)
        t = Al @ vec(us)
        tt = dot(t, t)
        omega = dot(t, sv) / tt if tt != 0.0 else 0.0
        x2 += alpha * u + omega * us
        r = sv - omega * t
        rho_old, rho = rho, dot(rh, r)
        rr = dot(r, r)
        if rr <= 1e-26 * bb:
            break
        if omega == 0.0 or force or rho * rho < 1e-20 * rhat2 * rr:
            rh = r.copy(); rho = rhat2 = rr; alpha = omega = 1.0; restart = True
    err2 = np.linalg.norm(x2 - xs[own]) / np.linalg.norm(xs)
    assert err2 < 1e-10, err2
    assert its2 <= its // 2 + 1, (its2, its)
    print(f"rank {rank}: rows {nloc} ghosts {len(ghost)} iters {its} (serial {its_s}) err {err:.2e}; "
          f"Neumann-preconditioned: iters {its2} err {err2:.2e} gershgorin {float(tr):.3f}", flush=True)
    dist.destroy_process_group()
''')


def test_slab_bicgstab_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PG_ROOT=str(ROOT), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script)], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("Neumann-preconditioned: iters") == 2

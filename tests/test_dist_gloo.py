"""world_size-2 gloo test (CPU) of the N>1 ALGORITHM the HIP path implements: 1-D slab partition of the reduced
system by planes (pg_partition_planes), one ghost chunk per neighbour and SpMV, dot products all-reduced, the collective admission test and the iteration of the
Neumann right preconditioner --
restated here with the oracle's matrix in numpy and torch.distributed(gloo) so that it runs without a GPU.
The LOCAL NUMBERING is the product's own: every rank asks pg_slab_numbering_host (the host code build_numbering ends in,
pg_host_algos.h) for its segment offsets, ghost segments and send chunks, lays its vectors out accordingly, and the halo is
the exchange pg_comm.hip makes -- one contiguous send and one contiguous receive per unknown kind and neighbour, straight
out of / into the vectors (gloo send / recv standing in for ncclSend / ncclRecv).
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
    # ---- the product's local numbering of this slab (pg_slab_numbering_host): active flags of the stored planes in, segment
    # offsets / ghost segments / send chunks out
    K, nplanes = 2, n + 1
    p0, p1 = int(bounds[rank]), int(bounds[rank + 1])
    s0, s1 = max(p0 - 3, 0), min(p1 + 3, nplanes)
    Mloc = (s1 - s0) * plane
    kind = idx // M
    active = np.zeros(K * Mloc, dtype=np.uint8)
    stored = (pl >= s0) & (pl < s1)
    active[kind[stored] * Mloc + (cell[stored] - s0 * plane)] = 1
    out = np.zeros(2 + 10 * K, dtype=np.int64)
    L.check(L.lib().pg_slab_numbering_host(C.c_int32(K), C.c_int64(plane), C.c_int64(nplanes), C.c_int64(p0), C.c_int64(p1),
                                           active.ctypes.data_as(C.POINTER(C.c_uint8)), L.iptr(out)))
    n_own, n_ghost = int(out[0]), int(out[1])
    names = ("cnt_own", "off_own", "cntL", "offL", "cntU", "offU", "sendL_off", "sendL_cnt", "sendU_off", "sendU_cnt")
    nb = {nm: out[2 + q * K: 2 + (q + 1) * K].tolist() for q, nm in enumerate(names)}
    nloc = len(own)
    assert n_own == nloc, (n_own, nloc)
    # owned: the reference's common_idx order restricted to the slab (kind 0 by cell, then kind 1) -- `own` already is
    assert [int(np.count_nonzero(kind[own] == k)) for k in range(K)] == nb["cnt_own"]
    assert nb["off_own"] == [0, nb["cnt_own"][0]]
    # ghosts in the product's order: lower ghost plane kind 0, kind 1, then upper ghost plane kind 0, kind 1
    glist = []
    for side_plane, cnt_key, off_key in ((p0 - 1, "cntL", "offL"), (p1, "cntU", "offU")):
        for k in range(K):
            g = np.flatnonzero((pl == side_plane) & (kind == k)) if 0 <= side_plane < nplanes else np.zeros(0, dtype=np.int64)
            assert len(g) == nb[cnt_key][k], (side_plane, k, len(g), nb[cnt_key][k])
            assert nb[off_key][k] == nloc + len(glist)
            glist.extend(g.tolist())
    ghost = np.array(glist, dtype=np.int64)
    assert len(ghost) == n_ghost
    # every column a row of this slab references is owned or sits in one of those ghost segments (one ghost plane per side)
    assert np.all(np.isin(np.setdiff1d(cols, own), ghost))
    # the send chunks are the owned actives of the first / last owned plane, contiguous at the start / end of each kind
    for k in range(K):
        seg = own[nb["off_own"][k]: nb["off_own"][k] + nb["cnt_own"][k]]
        if p0 > 0:
            assert np.array_equal(seg[: nb["sendL_cnt"][k]], np.flatnonzero((pl == p0) & (kind == k)))
            assert nb["sendL_off"][k] == nb["off_own"][k]
        else:
            assert nb["sendL_cnt"][k] == 0
        if p1 < nplanes:
            assert np.array_equal(own[nb["sendU_off"][k]: nb["sendU_off"][k] + nb["sendU_cnt"][k]], np.flatnonzero((pl == p1 - 1) & (kind == k)))
        else:
            assert nb["sendU_cnt"][k] == 0
    lmap = -np.ones(A.shape[0], dtype=np.int64); lmap[own] = np.arange(nloc); lmap[ghost] = nloc + np.arange(len(ghost))
    Al = sp.csr_matrix((Aloc.data, lmap[Aloc.indices], Aloc.indptr), shape=(nloc, nloc + len(ghost)))

    def halo(v):
        """pg_comm.hip rccl_halo: per kind, send the chunk of owned values the neighbour needs, receive its chunk straight into
        the ghost segment -- contiguous slices of v, nothing packed.  (gloo send / recv are blocking: the lower rank of a
        pair sends first.)"""
        for k in range(K):
            for nbr, s_off, s_cnt, r_off, r_cnt in ((rank - 1, nb["sendL_off"][k], nb["sendL_cnt"][k], nb["offL"][k], nb["cntL"][k]),
                                                    (rank + 1, nb["sendU_off"][k], nb["sendU_cnt"][k], nb["offU"][k], nb["cntU"][k])):
                if nbr < 0 or nbr >= world:
                    continue
                snd = torch.from_numpy(v[s_off: s_off + s_cnt].copy())
                rcv = torch.zeros(r_cnt, dtype=torch.float64)
                if rank < nbr:
                    if s_cnt: dist.send(snd, nbr)
                    if r_cnt: dist.recv(rcv, nbr)
                else:
                    if r_cnt: dist.recv(rcv, nbr)
                    if s_cnt: dist.send(snd, nbr)
                v[r_off: r_off + r_cnt] = rcv.numpy()
    # what I send up is what my upper neighbour's lower ghost segment holds, kind by kind (both sides computed it alone)
    mine = torch.tensor(nb["sendL_cnt"] + nb["sendU_cnt"] + nb["cntL"] + nb["cntU"], dtype=torch.int64)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    for r in range(world - 1):
        lo, up = allc[r].tolist(), allc[r + 1].tolist()
        assert lo[K:2 * K] == up[2 * K:3 * K], (lo, up)      # sendU of r == cntL of r + 1
        assert up[0:K] == lo[3 * K:4 * K], (lo, up)          # sendL of r + 1 == cntU of r
    # a halo of a vector that holds global numbers: every ghost entry receives the number of the unknown it stands for
    probe = np.zeros(nloc + len(ghost)); probe[:nloc] = own.astype(np.float64) + 1.0
    halo(probe)
    assert np.array_equal(probe[nloc:], ghost.astype(np.float64) + 1.0)

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

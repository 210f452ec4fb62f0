#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X:
    time-steps/sec + SpMV GB/s (% HBM roofline), 3D mono diffusion 512^3   (benchmark/Heat3D.jl shape)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 512]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the resident problem: one Crank-Nicolson time step of
solve_DiffusionUnsteadyMono! = right-hand side (1 SpMV + fused vector kernel) + BiCGStab solve to
reltol 1e-12 + un-scaling, all on the GPU through the C ABI.  Inputs (capacities, CSR system, state) are
resident in HBM when the timed region starts.

N > 1 is weak scaling: grid (n, n, n*N), domain (4, 4, 4N), one sphere per slab,
slab-decomposed along z with RCCL halo exchange + dot all-reduce inside libpenguin_hip.so; `value` is the
aggregate n^3-subdomain time-steps per second (N x global steps/s).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the
SpMV, HIP-event timed on the library stream inside the timed region), `step_roofline` (the whole time step's
algorithmic bytes over its wall time) and `cpu_baseline` (the oracle's C
restatement of the same Krylov loop on the same matrix, on the host cores of this box; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=512, help="cells per dimension of one slab (512 = the metric's config)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (SURVEY 8d, config 4): the SAME n^3 problem slab-decomposed over the ranks "
                         "(default: weak scaling, one n^3 sphere problem per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=6,
                    help="time-steps the CPU baseline times per thread count (6 single-thread steps ~ 10 s, + 6 on all cores)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch  # device sync + torch.distributed control plane (rendezvous, barrier, max-over-ranks)
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    import penguin.jl_amd as pj
    from penguin.jl_amd import _lib as L

    # RCCL prints a version banner on STDOUT the first time a communicator is created on a box; stdout is reserved for
    # the one JSON line, so fd 1 points at stderr while the communicators come up
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            box = [pj.get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            pj.init_distributed(local_rank, rank, world, box[0])
            dist.barrier()
        elif os.environ.get("PG_TEST_RCCL"):
            # 1-rank RCCL communicator: exercises ncclCommInitRank / ncclAllReduce on a 1-GPU box
            pj.init_distributed(local_rank, 0, 1, pj.get_unique_id())
        else:
            pj.init(local_rank)
        torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    lib = L.lib()

    # ---------------------------------------------------------------- workload (synthetic, deterministic)
    n, g = args.n, (1 if args.strong else world)
    mesh = pj.Mesh((n, n, n * g), (4.0, 4.0, 4.0 * g), (0.0, 0.0, 0.0))
    body = pj.Sphere((2.01, 2.01, 2.01), 1.0) if g == 1 else pj.MultiSphere([(2.01, 2.01, 2.01 + 4.0 * s) for s in range(g)], 1.0)
    t0 = time.time()
    cap = pj.Capacity(body, mesh)
    cap_ms = cap.kernel_ms
    op = pj.DiffusionOps(cap)
    M = (n + 1) * (n + 1) * (n * g + 1)
    keys = ("left", "right", "top", "bottom")          # benchmark/Heat3D.jl:57-62
    bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in keys})
    bci = pj.Dirichlet(1.0)
    # f = 0 and D = 1 are passed as constants and T0 = zeros as None: the reference's closures / zeros(2M) would
    # be evaluated / allocated over all M = 1.1e9 padded cells on every rank at 8 GPUs
    phase = pj.Phase(cap, op, 0.0, 1.0)
    dt = 0.75 * (4.0 / n) ** 2                         # benchmark/Heat3D.jl:69
    s = pj.DiffusionUnsteadyMono(phase, bcb, bci, dt, None, "BE")   # BE first (Heat3D.jl:72)
    setup_s = time.time() - t0
    opts = L.pg_krylov_opts(L.PG_METHOD["bicgstab"], 1e-12, 0.0, 0, 4, int(os.environ.get("PG_WARM_START", "1")))
    info = L.pg_step_info()
    L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
    CN = L.PG_SCHEME["CN"]                             # then CN (Heat3D.jl:74)
    run = L.pg_run_info()

    def steps(k: int) -> L.pg_run_info:
        r = L.pg_run_info()
        L.check(lib.pg_solver_run(s._h, C.c_double(1e30), C.c_int32(CN), C.byref(opts), C.c_int32(0), C.c_int64(k),
                                  C.c_int32(0), C.byref(r)))
        return r

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        L.check(lib.pg_device_synchronize())

    steps(args.warmup)
    L.check(lib.pg_set_profiling(1))                   # HIP events around every SpMV launch on the library stream
    sync()
    t0 = time.perf_counter()
    run = steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    L.check(lib.pg_set_profiling(0))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    sysinfo = s.system_info(3)                         # the run matrix as the SpMV streams it (preconditioned)
    n_rows, nnz = int(sysinfo.n_own), int(sysinfo.nnz)
    if world > 1:
        tot = torch.tensor([n_rows, nnz], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        n_rows_g, nnz_g = int(tot[0].item()), int(tot[1].item())
    else:
        n_rows_g, nnz_g = n_rows, nnz
    b_csr = 12 * nnz + 20 * n_rows                     # algorithmic bytes of one plain-CSR SpMV (this rank)
    # the kernel in the loop streams the stencil-sliced image of the same matrix: its algorithmic bytes are the
    # bytes of THAT format (records + P/G streams + x and y once) -- pricing it with the CSR figure would credit
    # bytes it never has to move (DESIGN.md "SpMV")
    # + 8 n: every launch timed in the loop is a fused-dot launch that also reads r-hat (SURVEY's CSR figure leaves the
    # dot operand out; the CSR number below is kept as SURVEY defines it)
    b_spmv = int(sysinfo.spmv_bytes) + 8 * n_rows if os.environ.get("PG_SPMV_VARIANT", "70") in ("70", "66") else b_csr
    if bool(sysinfo.neumann_ok) and os.environ.get("PG_POLY", "1") != "0" and b_spmv != b_csr:
        # preconditioned loop: the bracketed launches alternate between v = A u (r-hat) and t = A u_s, whose (t, s) dot
        # reads s as a vector of its own (+8 n on every second launch)
        b_spmv += 4 * n_rows
    spmv_ms = run.spmv_ms_total / max(run.spmv_launches, 1)
    achieved = b_spmv / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0
    iters = run.total_iters / max(run.steps, 1)

    traffic = None
    tf = ROOT / "profiles" / "r01_spmv_traffic.json"
    if tf.exists():
        try:
            traffic = json.loads(tf.read_text()).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "time-steps/sec + SpMV GB/s (% HBM roofline), 3D mono diffusion 512^3",
        "value": g * args.steps / elapsed,
        # weak scaling: every rank advances its own n^3 slab each step, so the whole-job rate is N x the global step rate
        # (n^3-subdomain time-steps per second summed over the slabs); at N = 1 and with --strong it is the plain step rate
        "unit": "time-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if args.strong and world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"3D monophasic unsteady diffusion {n}^3 per GPU (grid {n}x{n}x{n * g}), sphere r=1 per slab, "
                        "Dirichlet(1) interface, Dirichlet(1) on :left/:right/:top/:bottom, BE first solve then CN steps "
                        "(benchmark/Heat3D.jl shape), BiCGStab reltol 1e-12 on the equilibrated, cell-block-preconditioned reduced CSR system",
            "grid": [n, n, n * g],
            "rows_global": n_rows_g, "nnz_global": nnz_g, "rows_rank0": n_rows, "nnz_rank0": nnz,
            "krylov_iters_per_step": iters,
            "spmv_per_step": run.spmv_launches / max(run.steps, 1),
            "parallelism": f"slab-z x{world}, RCCL halo + dot all-reduce",
            "value_is": "global time-steps/s" if g == 1 else f"{n}^3-slab time-steps/s summed over the {world} slabs "
                        f"(= {world} x {args.steps / elapsed:.1f} global steps/s)",
            "setup_s": setup_s, "capacity_kernels_ms": cap_ms,
            # set-up kernels reported separately, over the (3 + 5N) * 8 * M bytes of capacity fields they produce
            "capacity_cells_per_s": M / (cap_ms * 1e-3) / max(world, 1) if cap_ms > 0 else None,
            "capacity_GBs": (3 + 5 * 3) * 8 * M / max(world, 1) / (cap_ms * 1e-3) / 1e9 if cap_ms > 0 else None,
            "device": pj.device_name(),
        },
        "roofline": {
            "kernel": "k_spmv_s (stencil-sliced CSR: uniform / pattern slices + packed irregular rows, fp64)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "bytes_per_launch": b_spmv,
            "csr_bytes_per_launch": b_csr,
            "csr_equivalent_GBs": b_csr / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else 0.0,
            "rows_uniform": int(sysinfo.rows_uniform), "rows_pattern": int(sysinfo.rows_pattern),
            "rows_irregular": int(sysinfo.rows_irregular), "slices": int(sysinfo.spmv_slices),
            "avg_launch_ms": spmv_ms,
            "launches_timed": int(run.spmv_launches),
        },
    }

    # The whole SpMV-bound time step against the same roofline (north star: ">= 60 % of the HBM roofline on the SpMV-bound
    # time-step"): algorithmic bytes of one CN step of THIS loop on this rank -- every kernel's operand vectors once, the
    # matrix in the format that is streamed --
    #   per BiCGStab iteration: 2 SpMV (format bytes, x and y included) + r-hat in both (2 x 8n) + k_bicg_s (r, v, r-hat -> s:
    #   4 x 8n) + k_bicg_xrp (x, p, s, t, v -> x, r, p: 8 x 8n)  = 2 spmv_bytes + 112 n
    #   per step: + the right-hand side's SpMV (spmv_bytes) + k_rhs_init (74 n)
    # divided by the measured wall time of the step (launch gaps, host polls and the per-step kernels included).
    b_fmt = int(sysinfo.spmv_bytes)
    neumann = bool(sysinfo.neumann_ok) and os.environ.get("PG_POLY", "1") != "0"
    if neumann:
        # right-preconditioned iteration (M^-1 = 2I - A): 4 SpMVs (two of them with the 2x - Ax epilogue, no r-hat), the
        # separate (t, s) operand, k_bicg_s (4 vectors) and k_bicg_xrp with u and u_s (10 vectors):
        # 4 spmv_bytes + (2 + 1 + 4 + 10) x 8 n = 4 spmv_bytes + 136 n
        step_bytes = (4.0 * iters + 1.0) * b_fmt + iters * 136.0 * n_rows + 74.0 * n_rows
    else:
        step_bytes = (2.0 * iters + 1.0) * b_fmt + iters * 112.0 * n_rows + 74.0 * n_rows
    step_gbs = step_bytes / (elapsed / args.steps) / 1e9
    out["step_roofline"] = {
        "what": ("one CN time step of the loop on rank 0: (4 iters + 1) SpMV format bytes + iters x 136 n" if neumann else
                 "one CN time step of the loop on rank 0: (2 iters + 1) SpMV format bytes + iters x 112 n") +
                " (BiCGStab vector kernels and the dot operands) + 74 n (right-hand side), over the wall time per step",
        "neumann_preconditioner": neumann, "gershgorin_radius": float(sysinfo.gershgorin),
        "bound": "hbm", "bytes_per_step": step_bytes, "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": step_gbs / HBM_PEAK_GBS,
        "survey_8d_bytes_per_step": iters * (2.0 * b_csr + 168.0 * n_rows) + 48.0 * n_rows + 16.0 * n_rows,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(s, dt, args.cpu_steps, n)
        except Exception as e:   # the GPU metric above must still be reported (e.g. the host lacks memory for the 65 M-entry copy)
            out["cpu_baseline"] = {"value": None, "unit": "time-steps/s", "cores": 1, "kind": "port",
                                   "sample": f"not measured: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(s, dt: float, cpu_steps: int, n: int) -> dict:
    """The oracle's C restatement of the same loop (oracle/krylov_ref.c) on the SAME reduced system, timed on
    this box's host cores.  The reference itself (Julia) cannot run here; kind = "port"."""
    import scipy.sparse as sp

    from oracle import krylov_c

    Ah, bh, idx = s.system(3)          # the preconditioned run system (B^-1 S A S, B^-1 S b) the GPU iterates on
    nrow = Ah.shape[0]
    Ah = Ah[:, :nrow].tocsr()
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))   # a 1-GPU box's CPU share
    res = {}
    for label, nt, reps in (("single_thread", 1, cpu_steps), ("all_cores", ncores, max(cpu_steps, 2))):
        t0 = time.perf_counter()
        its = 0
        for _ in range(reps):
            y = krylov_c.spmv(Ah, bh, nthreads=nt)          # the CN right-hand side's SpMV
            x, it, rn = krylov_c.solve(Ah, bh, "bicgstab", reltol=1e-12, maxiter=10000, nthreads=nt)
            its += it
        el = time.perf_counter() - t0
        res[label] = {"value": reps / el, "cores": nt, "iters_per_step": its / reps, "steps_timed": reps}
    # the same algorithm the GPU loop runs (Neumann right preconditioner), on all host cores: an apples-to-apples figure
    # next to the reference-like plain iteration above
    t0 = time.perf_counter()
    its = 0
    reps = max(cpu_steps, 2)
    for _ in range(reps):
        y = krylov_c.spmv(Ah, bh, nthreads=ncores)
        x, it, rn = krylov_c.solve(Ah, bh, "bicgstab_neumann", reltol=1e-12, maxiter=10000, nthreads=ncores)
        its += it
    el = time.perf_counter() - t0
    res["all_cores_neumann"] = {"value": reps / el, "cores": ncores, "iters_per_step": its / reps, "steps_timed": reps}
    return {
        "value": res["single_thread"]["value"],
        "unit": "time-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{res['single_thread']['steps_timed']} CN step(s) of the same {n}^3 reduced system (n={nrow}, nnz={Ah.nnz}): "
                  "1 SpMV + plain BiCGStab reltol 1e-12 (the iteration IterativeSolvers would run; the GPU loop's Neumann right "
                  "preconditioner is not applied here), oracle/krylov_ref.c, single thread (the reference's Krylov path is "
                  "single-threaded); all-cores OpenMP figure alongside",
        "all_cores": res["all_cores"],
        "all_cores_same_algorithm_as_gpu": res["all_cores_neumann"],
        "host_cores": ncores,
    }


if __name__ == "__main__":
    main()

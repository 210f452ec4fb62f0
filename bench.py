#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X:
    time-steps/sec + SpMV GB/s (% HBM roofline), 3D mono diffusion 512^3   (benchmark/Heat3D.jl shape)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 512] [--strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the resident problem: one Crank-Nicolson time step of
solve_DiffusionUnsteadyMono! = right-hand side (1 SpMV + fused vector kernel) + BiCGStab solve to reltol 1e-12 (tested in
the units of x: ||S r|| <= reltol ||S b||) + recovery of x, all on the GPU through the C ABI.  Inputs (capacities, CSR
system, state) are resident in HBM when the timed region starts.

N > 1: one process per GPU.  Invoked WITHOUT torch.distributed.run (`python bench.py --gpus N`, as the driver may do), the
script starts the N ranks itself as child processes (`python -m torch.distributed.run ...`) before anything touches a
GPU and relays rank 0's JSON line.  Default = weak scaling: grid (n, n, n*N), domain (4, 4, 4N),
one sphere per slab; every rank owns one n^3 sphere problem, the fluid stays away from the slab faces, so NO halo is
exchanged (the matrix has no ghost columns) and the only collective in the loop is the all-reduce of the Krylov scalars;
`value` = n^3-subdomain time-steps per second summed over the ranks.  `--strong`: the SAME n^3 problem cut into N slabs
(balanced by active rows): every SpMV exchanges one ghost plane with each neighbour (RCCL send/recv over xGMI,
overlapped with the interior rows).  No multi-GPU run of either kind has been measured yet (no multi-GPU box was
available to the author): the slab code is verified with virtual ranks on one GPU (tests/test_gpu_config4.py).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the SpMV launch,
HIP-event timed on the library stream inside the timed region), `step_roofline` (the whole time step's algorithmic
bytes over its wall time) and `cpu_baseline` (the oracle's C restatement of the same Krylov loop on the same matrix, on
the host cores of this box; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0   # same guide: what a streaming kernel reaches on this part (the practical ceiling)
PROFILE_SUMMARY = ROOT / "profiles" / "r03_spmv_profile.json"   # scripts/profile_round3.sh: rocprofv3 trace + PMC passes


def fail(msg: str) -> None:
    """A run that is not a valid measurement ends with a message and a non-zero code (not an assert: python -O strips them)."""
    print(f"bench.py: INVALID RUN: {msg}", file=sys.stderr, flush=True)
    raise SystemExit(3)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (nothing in this process
    has touched the GPU) and hand back their exit code; rank 0's JSON line goes to our stdout through the children."""
    port = int(os.environ.get("MASTER_PORT", "29541"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--n", str(args.n)]
    if args.strong:
        cmd.append("--strong")
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env, cwd=str(ROOT)).returncode


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
    ap.add_argument("--cpu-steps", type=int, default=4,
                    help="time-steps the CPU baseline times per variant (4 single-thread plain solves ~ 10 s)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            raise SystemExit(spawn_ranks(args))
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
    warm = int(os.environ.get("PG_WARM_START", "1"))
    opts = L.pg_krylov_opts(L.PG_METHOD["bicgstab"], 1e-12, 0.0, 0, 4, warm, 0, 0)
    info = L.pg_step_info()
    L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
    if not info.converged:
        fail("the first (BE) solve did not converge")
    CN = L.PG_SCHEME["CN"]                             # then CN (Heat3D.jl:74)

    def steps(k: int, o=opts) -> L.pg_run_info:
        r = L.pg_run_info()
        L.check(lib.pg_solver_run(s._h, C.c_double(1e30), C.c_int32(CN), C.byref(o), C.c_int32(0), C.c_int64(k),
                                  C.c_int32(0), C.byref(r)))
        return r

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        L.check(lib.pg_device_synchronize())

    steps(args.warmup)
    L.check(lib.pg_set_profiling(1))                   # HIP events around sampled SpMV launches on the library stream
    sync()
    t0 = time.perf_counter()
    run = steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    L.check(lib.pg_set_profiling(0))
    # a step that did not meet the tolerance is not a valid step: the number below would be meaningless
    if run.unconverged_steps != 0:
        fail(f"{run.unconverged_steps} of {run.steps} timed solves did not converge")
    if run.steps != args.steps:
        fail(f"{run.steps} steps ran instead of {args.steps}")
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    sysinfo = s.system_info(3)                         # the run matrix as the SpMV streams it (preconditioned)
    n_rows, nnz = int(sysinfo.n_own), int(sysinfo.nnz)
    if world > 1:
        tot = torch.tensor([n_rows, nnz, int(sysinfo.n_ghost)], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        n_rows_g, nnz_g, ghosts_g = int(tot[0].item()), int(tot[1].item()), int(tot[2].item())
    else:
        n_rows_g, nnz_g, ghosts_g = n_rows, nnz, int(sysinfo.n_ghost)
    b_csr = 12 * nnz + 20 * n_rows                     # algorithmic bytes of one plain-CSR SpMV (this rank)
    # the kernel in the loop streams the stencil-sliced image of the same matrix: its algorithmic bytes are the bytes of
    # THAT format (unit + slice records, P/G streams, x and y once) -- pricing it with the CSR figure would credit bytes
    # it never has to move (DESIGN.md "SpMV")
    sliced = os.environ.get("PG_SPMV_VARIANT", "70") in ("70", "66")
    b_fmt_full = int(sysinfo.spmv_bytes) if sliced else b_csr
    # the warm loop iterates on the same matrix without the Dirichlet interface unknowns (rows of the identity, solved before
    # the iteration: pg_reduce.hip); the step's first product (right-hand side, start residual) uses the full matrix
    loopinfo = s.system_info(7)
    n_full = n_rows
    reduced = sliced and int(loopinfo.spmv_bytes) != int(sysinfo.spmv_bytes)
    if reduced:
        n_rows = int(loopinfo.rows_matrix)
    b_fmt = int(loopinfo.spmv_bytes) if sliced else b_csr
    m = int(run.poly_degree)                           # products per application of the preconditioned operator (0: plain)
    # launches timed in the loop: LEAN (a factor of the preconditioner polynomial: x in, y out, the matrix) and CLOSING
    # launches (fused dots: + r-hat and the chain's input vector, read as the subtrahend / dot operand: + 16 n)
    lean_ms = run.spmv_lean_ms_total / max(run.spmv_lean_launches, 1)
    dots_ms = run.spmv_ms_total / max(run.spmv_launches, 1)
    b_dots = b_fmt + (16 if m else 12) * n_rows        # plain iteration: r-hat in both launches, s in every second one
    dominant_lean = m >= 2 and run.spmv_lean_launches > 0
    # x-space form of the preconditioned loop (the default, pg_krylov.hip): the m - 1 launches of a chain are HORNER steps
    # u <- tau in + (I - tau A) u -- the matrix, u in, u out and the chain's input vector as a third stream (+ 8 n), except the
    # first, whose u is the input itself; the closing launch is the plain product with its dots: + r-hat (first application of
    # an iteration) or + r-hat and s (second): + 12 n on average.  y-space form: lean launches of two streams (b_fmt), closing
    # launches + 16 n, and a recovery per solve.
    xspace = m >= 2 and int(run.poly_xspace) != 0
    b_chain = (b_fmt + 8.0 * n_rows * (m - 2) / (m - 1)) if xspace else float(b_fmt)
    if xspace:
        b_dots = b_fmt + 12 * n_rows
    k_ms, k_bytes = (lean_ms, b_chain) if dominant_lean else (dots_ms, b_dots)
    achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # Krylov iterations: an iteration that ended at its half step ran one application of the operator, not two
    iters = run.total_iters / max(run.steps, 1)
    iters_eff = (run.total_iters - 0.5 * run.half_exits) / max(run.steps, 1)

    # What the committed rocprofv3 runs say about the same launch (profiles/r03_spmv_profile.json, written by
    # scripts/profile_round3.sh from `rocprofv3 --kernel-trace --stats -- python3 bench.py ...` and the separate --pmc
    # passes): used ONLY when that profile was taken from these very sources (source hash), else reported as stale.
    from penguin.jl_amd.build import source_hash

    src_hash = source_hash()
    prof = {}
    if PROFILE_SUMMARY.exists():
        try:
            prof = json.loads(PROFILE_SUMMARY.read_text())
        except Exception:
            prof = {}
    prof_current = bool(prof) and prof.get("source_hash") == src_hash
    trace_avg_us = prof.get("chain_launch_rocprofv3_stats_average_us") if prof_current else None
    traffic = prof.get("chain_launch_hbm_bytes_per_launch") if prof_current else None

    halo = ghosts_g > 0
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
                        "(benchmark/Heat3D.jl shape), BiCGStab reltol 1e-12 in the units of x (||S r|| <= reltol ||S b||) on the "
                        "equilibrated, cell-block-preconditioned reduced system, warm start",
            "grid": [n, n, n * g],
            "rows_global": n_rows_g, "nnz_global": nnz_g, "rows_rank0": n_rows, "nnz_rank0": nnz,
            "krylov_iters_per_step": iters, "krylov_iters_per_step_counting_half_steps_as_half": iters_eff,
            "polynomial_preconditioner_degree": m,
            "solves_that_needed_a_second_application": int(2 * run.total_iters - run.half_exits - run.steps),
            "products_per_step_in_the_solves": getattr(run, "products", 0) / max(run.steps, 1),
            "start_of_each_solve": "the previous state, extrapolated from older states where a least-squares fit says it pays "
                                   "(PG_GUESS_STATES, pg_solver_guess_info): same stopping test, fewer products",
            "spmv_timed_per_step": (run.spmv_launches + run.spmv_lean_launches) / max(run.steps, 1),
            "timed_window": f"steps {args.warmup + 1}..{args.warmup + args.steps} after the BE solve (iterations per step fall as the field settles)",
            "parallelism": (f"slab-z x{world}: " + ("one ghost plane per neighbour per SpMV (RCCL send/recv, overlapped) + " if halo else
                            "no halo traffic (no row references a ghost column) + ") + "all-reduce of the Krylov scalars") if world > 1
                           else "1 GPU",
            "value_is": "global time-steps/s" if g == 1 else f"{n}^3-slab time-steps/s summed over the {world} slabs "
                        f"(= {world} x {args.steps / elapsed:.1f} global steps/s)",
            "setup_s": setup_s, "capacity_kernels_ms": cap_ms,
            # set-up kernels reported separately, over the (3 + 5N) * 8 * M bytes of capacity fields they produce
            "capacity_cells_per_s": M / (cap_ms * 1e-3) / max(world, 1) if cap_ms > 0 else None,
            "capacity_GBs": (3 + 5 * 3) * 8 * M / max(world, 1) / (cap_ms * 1e-3) / 1e9 if cap_ms > 0 else None,
            "device": pj.device_name(),
            "library_variant": pj.config_string(),     # every PG_* selector this process ran with (pg_config_string)
            "source_hash": src_hash,
        },
        "roofline": {
            "kernel": "k_spmv_s (marching units + stencil slices + packed irregular rows, fp64): " +
                      (("Horner step u <- tau p + (I - tau A) u of the preconditioner polynomial (u in, u out, matrix, + the chain's "
                        "input p as a third stream in m - 2 of the m - 1 launches of a chain: bytes_per_launch is the chain's mean)"
                        if xspace else "lean launch w <- w - tau A w of the preconditioner polynomial (x in, y out, matrix)")
                       if dominant_lean else "launch with fused dots"),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            # `frac` is the LOWER of two measurements of the same launch: live HIP events around whole chains in this run, and
            # the average duration rocprofv3's kernel trace gives that kernel in the committed profile of these sources (the
            # figure a reader can recompute from profiles/).  Without a current profile it is the live figure alone.
            "frac": min(achieved, k_bytes / (trace_avg_us * 1e-6) / 1e9 if (trace_avg_us and dominant_lean) else achieved) / HBM_PEAK_GBS,
            "frac_events": achieved / HBM_PEAK_GBS,
            "frac_kernel_trace": (k_bytes / (trace_avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (trace_avg_us and dominant_lean) else None,
            "kernel_trace_avg_launch_us": trace_avg_us if dominant_lean else None,
            "profile": {"file": str(PROFILE_SUMMARY.relative_to(ROOT)), "source_hash_of_profile": prof.get("source_hash"),
                        "current": prof_current,
                        "note": None if prof_current else "no rocprofv3 profile of these sources is committed: frac = frac_events, traffic = null"},
            "peak_achievable": HBM_ACHIEVABLE_GBS,
            "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
            "traffic": traffic,
            "traffic_over_algorithmic": (traffic / k_bytes) if (traffic and dominant_lean) else None,
            "bytes_per_launch": k_bytes, "form": ("x-space: Horner chains, no recovery" if xspace else "y-space: lean chains + recovery") if m >= 2 else "plain iteration",
            "avg_launch_ms": k_ms,
            "launches_timed": int(run.spmv_lean_launches if dominant_lean else run.spmv_launches),
            # HIP events bracket whole chains of launches back to back, i.e. kernel + the gap to the next dependent kernel;
            # rocprofv3's kernel trace of the same command (profiles/, copied into the traffic JSON) times the kernel alone
            "timing": "HIP events around each chain of m - 1 lean launches on the compute stream, every 3rd bracket",
            "closing_launches": {"what": "last product of a chain: + the chain's input vector and r-hat, fused dots, in-launch scalar phase",
                                 "bytes_per_launch": b_dots, "avg_launch_ms": dots_ms, "launches_timed": int(run.spmv_launches),
                                 "achieved": b_dots / (dots_ms * 1e-3) / 1e9 if dots_ms > 0 else 0.0,
                                 "frac": b_dots / (dots_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dots_ms > 0 else 0.0},
            "csr_bytes_per_launch": 12 * int(loopinfo.nnz) + 20 * n_rows if reduced else b_csr,
            "csr_equivalent_GBs": (12 * int(loopinfo.nnz) + 20 * n_rows if reduced else b_csr) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0,
            "rows_marched": int(loopinfo.rows_marched), "march_units": int(loopinfo.spmv_units),
            "rows_uniform": int(loopinfo.rows_uniform), "rows_pattern": int(loopinfo.rows_pattern),
            "rows_irregular": int(loopinfo.rows_irregular), "slices": int(loopinfo.spmv_slices),
            "rows_in_the_iteration": n_rows, "nnz_in_the_iteration": int(loopinfo.nnz),
            "rows_alone_on_their_diagonal_solved_before_the_iteration": n_full - n_rows,
            "full_matrix": {"rows": n_full, "nnz": nnz, "bytes_per_launch": b_fmt_full, "rows_irregular": int(sysinfo.rows_irregular)},
        },
    }

    # The whole SpMV-bound time step against the same roofline (north star: ">= 60 % of the HBM roofline on the SpMV-bound
    # time-step"): algorithmic bytes of one CN step of THIS loop on this rank -- every kernel's operand vectors once, the
    # matrix in the format that is streamed --
    #   per application of the operator (2 per iteration, 1 in an iteration that ends at its half step):
    #       (m - 1) lean launches (b_fmt) + the closing launch (b_fmt + 16 n);  plain iteration: one launch, b_fmt + 8 n (+ 8 n)
    #   per iteration: k_bicg_s (r, v, r-hat, S -> s: 5 x 8 n) + k_bicg_xrp (y, p, s: read + written, t, v, S: 9 x 8 n)
    #   per step: the right-hand side's SpMV (b_fmt) + the fused right-hand side / solver start (full rows: y-hat, S, constant
    #       part, mass, z, flags, compact index in, b-hat out = 54 n_full; compact rows: r and r-hat out = 16 n; p is not
    #       written, the first iteration reads r-hat for it -- so the first closing launch reads ONE extra vector: - 8 n; y
    #       is not zeroed, its first update assigns) + the recovery x = x0 + q(A) y:
    #       (m - 1) Horner launches (b_fmt + 8 n: the y vector as a third stream) + x += tau_0 u (24 n)
    # divided by the measured wall time of the step (launch gaps, host polls and the per-step kernels included).
    #   x-space form: per application the chain's m - 1 launches (b_chain each) + the closing launch (b_fmt + 8 n, + 8 n more in
    #       the second application of an iteration); k_bicg_xrp additionally reads the two preconditioned vectors and the index
    #       map of the compact system (11.5 x 8 n); the half-step update x += alpha M^-1 p (28 n) once per solve that ends there;
    #       no recovery.
    # degree of the preconditioner polynomial: the LAST solve's in `m`; the window's mean from the products counted (the
    # extrapolated start lowers it from step to step)
    apps = 2.0 * run.total_iters - run.half_exits            # applications of the preconditioned operator in the window
    m_mean = (run.products / apps) if (m >= 2 and apps > 0 and getattr(run, "products", 0) > 0) else float(m)
    # extrapolated start of the quiet steps (pg_solver_guess_info): per older state read, its product in the step's first
    # kernel (full-system length) and the state itself in the solve's first update of x (through the loop system's map); with
    # PG_GUESS_DEFER=0 both in the first kernel, which then also writes the new state out of place (one more write)
    guess_reads = getattr(run, "guess_states_read", 0) / max(run.steps, 1)
    cfgs = pj.config_string()
    guess_on = "guess_states=0" not in cfgs
    if "guess_defer=1" in cfgs:
        guess_bytes = guess_reads * (8.0 * n_full + 8.0 * n_rows)
    else:
        guess_bytes = (16.0 * guess_reads + (8.0 if guess_on else 0.0)) * n_full
    if xspace:
        half_per_step = run.half_exits / max(run.steps, 1)
        mm = m_mean
        b_chain_mean = b_fmt + 8.0 * n_rows * (mm - 2) / (mm - 1)
        # (compact loop system: the start writes r-hat only -- r = p = r-hat are read from it in the first iteration, whose
        #  k_bicg_s therefore reads one vector less)
        start_bytes = (54.0 * n_full + 8.0 * n_rows - 8.0 * n_rows) if reduced else 74.0 * n_full
        step_bytes = (2.0 * iters_eff) * ((mm - 1) * b_chain_mean + b_fmt + 8.0 * n_rows) + (iters - half_per_step) * 8.0 * n_rows \
            + iters * 40.0 * n_rows + (iters - half_per_step) * 92.0 * n_rows + half_per_step * 28.0 * n_rows + b_fmt_full + start_bytes \
            + guess_bytes
    elif m >= 2:
        per_apply = (m - 1) * b_fmt + (b_fmt + 16.0 * n_rows)
        start_bytes = (54.0 * n_full + 16.0 * n_rows - 8.0 * n_rows) if reduced else (74.0 * n_full + 8.0 * n_rows)
        step_bytes = (2.0 * iters_eff) * per_apply + iters * (40.0 + 72.0) * n_rows + b_fmt_full + start_bytes \
            + (m - 1) * (b_fmt + 8.0 * n_rows) + 24.0 * n_rows
    else:
        step_bytes = 2.0 * iters * b_fmt + b_fmt_full + iters * (12.0 + 40.0 + 72.0) * n_rows + 74.0 * n_full
    step_gbs = step_bytes / (elapsed / args.steps) / 1e9
    out["step_roofline"] = {
        "what": "one CN time step of the loop on rank 0: every launch's operand vectors once and the matrix in the streamed "
                "format (formula in bench.py), over the wall time per step",
        "polynomial_preconditioner_degree": m, "polynomial_preconditioner_degree_mean": m_mean,
        "older_states_read_per_step": guess_reads, "gershgorin_radius": float(sysinfo.gershgorin),
        "bound": "hbm", "bytes_per_step": step_bytes, "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": step_gbs / HBM_PEAK_GBS,
        "survey_8d_bytes_per_step": iters * (2.0 * b_csr + 168.0 * n_rows) + 48.0 * n_rows + 16.0 * n_rows,
    }

    # Several ranks, default (weak-scaling) invocation: the headline line above exchanges no halo (one sphere per slab).  The
    # SAME run therefore also times the strong-scaling shape of config 4 -- the one n^3 sphere problem cut into `world` slabs
    # balanced by active rows, every product exchanging one ghost chunk per unknown kind with each neighbour (ncclSend /
    # ncclRecv on the communication stream, overlapped with the interior rows) -- and reports it as a sub-record, so that the
    # first run on a multi-GPU node exercises the halo path without a second invocation.  Same loop, same tolerances.
    if world > 1 and not args.strong:
        try:
            out["strong_scaling"] = strong_record(pj, L, lib, dist, torch, args, world, rank, opts, CN, sync)
        except SystemExit:
            raise
        except Exception as e:
            out["strong_scaling"] = {"value": None, "error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            # the GPU loop's iteration count from a zero initial guess, for the comparison with the (cold-start) CPU figures
            cold = L.pg_krylov_opts(L.PG_METHOD["bicgstab"], 1e-12, 0.0, 0, 4, 0, 0, 0)
            rc = steps(3, cold)
            out["cpu_baseline"] = cpu_baseline(s, args.cpu_steps, n, m, float(sysinfo.gershgorin))
            out["cpu_baseline"]["gpu_cold_start_iters_per_step"] = rc.total_iters / max(rc.steps, 1)
        except Exception as e:   # the GPU metric above must still be reported (e.g. the host lacks memory for the 65 M-entry copy)
            out["cpu_baseline"] = {"value": None, "unit": "time-steps/s", "cores": 1, "kind": "port",
                                   "sample": f"not measured: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def strong_record(pj, L, lib, dist, torch, args, world, rank, opts, CN, sync) -> dict:
    """The n^3 sphere problem slab-decomposed over all ranks (SURVEY 8d config 4, strong scaling): K timed CN steps after the
    BE solve and the warm-up, max over ranks; every SpMV of the loop exchanges halos."""
    n = args.n
    mesh = pj.Mesh((n, n, n), (4.0, 4.0, 4.0), (0.0, 0.0, 0.0))
    cap = pj.Capacity(pj.Sphere((2.01, 2.01, 2.01), 1.0), mesh)
    phase = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
    bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
    dt = 0.75 * (4.0 / n) ** 2
    s = pj.DiffusionUnsteadyMono(phase, bcb, pj.Dirichlet(1.0), dt, None, "BE")
    info = L.pg_step_info()
    L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(info)))
    if not info.converged:
        fail("strong-scaling sub-run: the first (BE) solve did not converge")

    def steps(k):
        r = L.pg_run_info()
        L.check(lib.pg_solver_run(s._h, C.c_double(1e30), C.c_int32(CN), C.byref(opts), C.c_int32(0), C.c_int64(k), C.c_int32(0),
                                  C.byref(r)))
        return r

    steps(args.warmup)
    sync()
    t0 = time.perf_counter()
    run = steps(args.steps)
    sync()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    if run.unconverged_steps != 0 or run.steps != args.steps:
        fail(f"strong-scaling sub-run: {run.unconverged_steps} unconverged solves, {run.steps} of {args.steps} steps")
    full, loop = s.system_info(3), s.system_info(7)
    t = torch.tensor([int(full.n_own), int(full.n_ghost), int(loop.rows_matrix), int(loop.n_ghost_loop)], dtype=torch.int64, device="cuda")
    mx = t.clone()
    dist.all_reduce(t)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return {
        "what": f"strong scaling: the one {n}^3 sphere problem cut into {world} slabs (balanced by active rows); every product of the "
                "loop exchanges one ghost chunk per unknown kind with each neighbour (ncclSend / ncclRecv, overlapped with the "
                "interior rows), the Krylov scalars are all-reduced",
        "value": args.steps / elapsed, "unit": "time-steps/s", "ms_per_step": elapsed / args.steps * 1e3, "scaling": "strong",
        "steps": args.steps, "warmup": args.warmup, "n_gpus": world,
        "rows_global": int(t[0].item()), "ghost_entries_global": int(t[1].item()),
        "loop_rows_global": int(t[2].item()), "loop_ghost_entries_global": int(t[3].item()),
        "rows_max_per_rank": int(mx[0].item()), "loop_is_compact": bool(loop.loop_is_compact),
        "krylov_iters_per_step": run.total_iters / max(run.steps, 1), "polynomial_preconditioner_degree": int(run.poly_degree),
    }


def cpu_baseline(s, cpu_steps: int, n: int, m: int, gersh: float) -> dict:
    """The oracle's C restatement of the same loop (oracle/krylov_ref.c) on the SAME reduced system, timed on this box's
    host cores.  The reference itself (Julia) cannot run here; kind = "port".  Every CPU solve starts from zero (the host
    side has no time loop of its own): `gpu_cold_start_iters_per_step` is the GPU loop's count under the same condition."""
    from oracle import krylov_c
    from penguin.jl_amd import _lib as L

    Ah, bh, idx = s.system(3)          # the preconditioned run system (B^-1 S A S, B^-1 S b) the GPU iterates on
    nrow = Ah.shape[0]
    Ah = Ah[:, :nrow].tocsr()
    wts = np.empty(nrow)
    L.check(L.lib().pg_solver_get_row_scaling(s._h, 1, L.dptr(wts)))   # the weights of the convergence test
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))   # a 1-GPU box's CPU share
    res = {}
    variants = (("single_thread", 1, cpu_steps, 0), ("all_cores", ncores, max(cpu_steps, 2), 0),
                ("all_cores_same_algorithm_as_gpu", ncores, max(cpu_steps, 2), m))
    for label, nt, reps, deg in variants:
        t0 = time.perf_counter()
        its = nmv = 0
        for _ in range(reps):
            krylov_c.spmv(Ah, bh, nthreads=nt)          # the CN right-hand side's SpMV
            x, it, rn, mv = krylov_c.solve_poly(Ah, bh, deg, gersh, weights=wts, reltol=1e-12, maxiter=10000, nthreads=nt)
            its += it
            nmv += mv
        el = time.perf_counter() - t0
        res[label] = {"value": reps / el, "cores": nt, "iters_per_step": its / reps, "products_per_step": nmv / reps,
                      "steps_timed": reps, "polynomial_degree": deg}
    return {
        "value": res["single_thread"]["value"],
        "unit": "time-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{res['single_thread']['steps_timed']} CN step(s) of the same {n}^3 reduced system (n={nrow}, nnz={Ah.nnz}): "
                  "1 SpMV + plain BiCGStab from a zero initial guess to the same test (||S r|| <= 1e-12 ||S b||): the iteration "
                  "IterativeSolvers would run, oracle/krylov_ref.c, single thread (the reference's Krylov path is single-threaded); "
                  "all-cores OpenMP figures alongside, plain and with the GPU loop's polynomial preconditioner",
        "all_cores": res["all_cores"],
        "all_cores_same_algorithm_as_gpu": res["all_cores_same_algorithm_as_gpu"],
        "host_cores": ncores,
    }


if __name__ == "__main__":
    main()

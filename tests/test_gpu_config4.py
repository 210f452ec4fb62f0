"""BASELINE config 4 -- 3-D monophasic unsteady diffusion, 512^3, sphere body (benchmark/Heat3D.jl:55-74) -- under
asserting tests at FULL size.  The oracle's direct solve cannot reach 10.3 M rows, so the checks are the size-independent
ones the domain offers:

  * capacities: volumes tile the ball, cut set == {Γ > 0}, the active-row count the bench reports;
  * the solve exactly as bench.py runs it (BE first solve, then CN steps; BiCGStab reltol 1e-12, warm start, polynomial
    right preconditioner on): discrete maximum principle, Dirichlet interface value, and the TRUE relative residual
    ||b - A x|| / ||b|| of the REFERENCE's raw reduced system (no scaling, no preconditioner: what
    `remove_zero_rows_cols!` + `solve_system!` see, src/solver.jl:59-78,158-188) evaluated on the host;
  * the same run with IterativeSolvers' plain iteration (zero initial guess, no preconditioner) gives the same state
    to the north star's 1e-10;
  * the same problem cut into 8 slabs (the strong-scaling shape of config 4; virtual ranks = the per-rank code on host
    threads sharing the one GPU) reproduces the 1-rank state.
"""
import json
import math
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.common import rel_l2

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

N512_ROWS = 10_294_905          # rows of the reduced system bench.py reports for this configuration
KEYS = ("left", "right", "top", "bottom")   # benchmark/Heat3D.jl:57-62


def _solver(pj, cap, n, M):
    ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)           # f = 0, D = 1 (Heat3D.jl:64-65) as constants
    bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in KEYS})
    dt = 0.75 * (4.0 / n) ** 2                                   # Heat3D.jl:69
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")   # T0 = zeros, BE first (:65-72)
    return s, ph, bcb, dt


def test_full_size_properties_512(pj):
    n = 512
    M = (n + 1) ** 3
    mesh = pj.Mesh((n,) * 3, (4.0,) * 3)
    cap = pj.Capacity(pj.Sphere((2.01,) * 3, 1.0), mesh)
    V, G, ct = cap.V, cap.Γ, cap.cell_types
    assert V.sum() == pytest.approx(4.0 / 3.0 * math.pi, rel=1e-11)             # volumes tile the ball
    assert G.sum() == pytest.approx(4.0 * math.pi, rel=1e-10)                   # interface pieces tile the sphere
    assert np.array_equal(np.flatnonzero(ct == -1), np.flatnonzero(G > 0))      # cut set == {Γ > 0}
    cut = G > 0
    fluid = V > 0
    del ct

    # ---- the bench's run: reltol 1e-12, warm start, preconditioner on (all defaults of the host layer)
    s, ph, bcb, dt = _solver(pj, cap, n, M)
    info = s.system_info(0)
    assert info.n_own == N512_ROWS
    assert info.n_gamma == np.count_nonzero(cut)
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 1e30, bcb, pj.Dirichlet(1.0), "CN", save_states=False, max_steps=3,
                                     reltol=1e-12)
    assert s.unconverged == 0
    assert s.last_run.steps == 3 and s.last_run.unconverged_steps == 0
    assert s.last_run.worst_relres <= 1e-12
    assert bool(s.system_info(3).neumann_ok)                                     # the preconditioned loop did run
    x = s.x
    Tw, Tg = x[:M], x[M:]
    assert Tw[fluid].min() > -1e-9 and Tw[fluid].max() < 1.0 + 1e-9             # discrete maximum principle
    assert np.all(np.abs(Tg[cut] - 1.0) <= 1e-9)                                # Dirichlet interface value
    assert np.all(np.abs(Tw[~fluid & (Tw != 0.0)] - 1.0) <= 1e-9)               # border identity rows hold the border value
    # the field has actually moved: heat has entered through the interface but not yet reached the centre
    c = (n + 1) * (n + 1) * (n // 2) + (n + 1) * (n // 2) + n // 2
    assert abs(Tw[c]) < 1e-6 and Tw[fluid].max() > 0.5

    # ---- true residual of the reference's raw reduced system of the last step, on the host
    A, b, idx = s.system(1)                                                      # raw A (CN), b of the last step, common_idx
    assert A.shape[0] == N512_ROWS and np.all(np.diff(idx) > 0)
    xr = x[idx]
    r = b - A[:, :N512_ROWS] @ xr
    true_relres = float(np.linalg.norm(r) / np.linalg.norm(b))
    assert true_relres <= 1e-11, true_relres
    del A, r

    # ---- IterativeSolvers' plain iteration: zero initial guess, no preconditioner
    s2, ph2, bcb2, _ = _solver(pj, cap, n, M)
    pj.solve_DiffusionUnsteadyMono_b(s2, ph2, dt, 1e30, bcb2, pj.Dirichlet(1.0), "CN", save_states=False, max_steps=3,
                                     reltol=1e-12, warm_start=False, precond=-1)
    assert s2.unconverged == 0
    assert s2.last_run.total_iters > s.last_run.total_iters                      # (it is the slower iteration)
    assert rel_l2(s2.x, x) <= 1e-10                                             # north star tolerance on T


def test_config4_strong_8_slabs_reproduce_the_one_rank_state():
    """The exact 512^3 problem slab-decomposed over 8 (virtual) ranks: every rank exchanges halos; the assembled state
    must equal the 1-rank state.  Subprocess with a hard timeout: a failed rank would leave the others in a barrier."""
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "virtual_ranks_fullsize.py"), "8", "512", "3", "strong", "state"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["slabs_sum_to_one_rank"], out
    assert sum(out["virtual"]["n_own"]) == N512_ROWS
    assert all(g > 0 for g in out["virtual"]["n_ghost"])
    # the halo-exchanging slabs run the loop the 1-GPU benchmark runs: the compact system on every rank (pg_reduce.hip)
    v = out["virtual"]
    assert all(e > 0 for e in v["rows_alone_on_their_diagonal"]), v
    assert sum(v["rows_alone_on_their_diagonal"]) == out["one_rank"]["rows_alone_on_their_diagonal"][0], (v, out["one_rank"])
    assert sum(v["loop_rows"]) == out["one_rank"]["loop_rows"][0]
    assert all(0 < g for g in v["loop_ghosts"]), v
    assert out["virtual"]["iters"][0] == out["one_rank"]["iters"][0], out       # identical iteration counts
    assert out["state_max"] > 0.5                                               # a real field, not zeros
    assert out["state_rel_l2"] <= 1e-10, out

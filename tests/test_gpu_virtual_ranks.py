"""Slab decomposition verified on ONE GPU: N virtual ranks (host threads, in-process halo exchange
and all-reduce standing in for RCCL) run exactly the per-rank code and must reproduce the single-rank result.
Runs in a subprocess with a hard timeout: a rank that failed would leave the others in a barrier."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_virtual_ranks_reproduce_single_rank():
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "virtual_ranks_check.py")], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(out) >= {"sphere_2", "sphere_3", "sphere_4", "two_spheres_2", "two_spheres_4", "gmres_sphere_2", "ramp_sphere_2",
                        "ramp_sphere_3"}
    assert sum(1 for k in out if k.startswith("random")) == 8
    for case, v in out.items():
        assert v["rel_l2"] <= 1e-12, (case, v)                       # same solution as one rank
        assert sum(v["n_own"]) == v["n_total_1"], (case, v)          # every active unknown owned exactly once
        assert sum(v["nnz"]) == v["nnz_total_1"], (case, v)          # every matrix entry assembled exactly once
        assert len(set(v["iters"])) == 1, (case, v)                  # every rank sees the same scalars
        if v.get("random"):
            # the all-reduced dots are summed in another order than on one rank: the history may shift by an iteration
            assert abs(v["iters"][0] - v["iters_1"]) <= 2, (case, v)
            continue
        assert v["iters"][0] == v["iters_1"], (case, v)              # identical Krylov history
        assert all(g > 0 for g in v["n_ghost"]), (case, v)           # every rank exchanges a halo
        if "loop" in v:
            # the halo-exchanging slabs iterate on the COMPACT system, as one rank does: rows alone on their diagonal are
            # solved in the right-hand-side pass on every rank, the loop's vectors keep the remaining ghosts only
            for (full, rows, ghosts), ng in zip(v["loop"], v["n_ghost"]):
                assert 0 < rows < full, (case, v["loop"])
                assert 0 <= ghosts <= ng, (case, v["loop"], v["n_ghost"])
                if case.startswith(("sphere", "ramp")):
                    assert ghosts > 0, (case, v["loop"])             # one body across all slabs: every face exchanges
    for case in ("ramp_sphere_2", "ramp_sphere_3"):
        assert out[case]["ref_max"] > 1.2                            # the ramp reached the field
    # one sphere per slab (the weak-scaling body): the partition by active rows is balanced
    v = out["two_spheres_2"]
    assert max(v["n_own"]) <= 1.15 * min(v["n_own"])


def test_strong_scaling_shape_splits_the_spmv_around_the_halo():
    """One 96^3 sphere problem cut into 3 slabs (every rank exchanges halos): the slices that wait for the halo run in a
    second launch that accumulates into the first one's partial sums (spmv_with_halo); sizes add up to the 1-rank system
    and the Krylov history matches it, with the overlap split and without."""
    import os
    for overlap in ("1", "0"):
        r = subprocess.run([sys.executable, str(ROOT / "scripts" / "virtual_ranks_fullsize.py"), "3", "96", "2", "strong"], cwd=ROOT,
                           capture_output=True, text=True, timeout=300, env={**os.environ, "PG_HALO_OVERLAP": overlap, "PG_DEBUG": "1"})
        assert r.returncode == 0, (overlap, r.stdout[-500:], r.stderr[-1500:])
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["slabs_sum_to_one_rank"], out
        waits = [int(l.split("(")[1].split()[0]) for l in r.stderr.splitlines() if l.startswith("[pg_spmv] slices")]
        assert any(w > 0 for w in waits), r.stderr[-1500:]          # some slices do wait for the halo

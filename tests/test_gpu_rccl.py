"""The RCCL branch of pg_comm.hip inside the GPU test run.

* 1 rank behind a 1-rank RCCL communicator: `ncclCommInitRank`, `ncclCommSplit`, every scalar phase of the Krylov loop
  through `ncclAllReduce` + `k_derive` (the several-ranks code path) -- runs on the 1-GPU test box.
* 2 ranks on 2 GPUs (skipped where fewer are visible): the halo communicator on the communication stream, overlapped with
  the interior rows and not, against the 1-rank state.  Nothing else in the suite can see a reordering of the two streams.

Each run is a fresh child process (the library is initialised once per process; torch.distributed.run starts its ranks
before anything touches a GPU).
"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SCRIPT = str(ROOT / "scripts" / "dist_check.py")


def _run(cmd, env_extra, timeout=600):
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", **env_extra}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


def _gpu_count() -> int:
    """In a child process (it only counts devices; torch is not needed in the test process itself).  Round 2 found that
    importing torch INTO a process that had already loaded libpenguin_hip.so mapped a second copy of the HIP runtime and
    aborted at exit; the binding now shares one runtime with torch in either order (penguin/jl_amd/_lib.py,
    tests/test_abi_cpu.py::test_one_hip_runtime_whichever_of_torch_and_the_library_comes_first)."""
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                       timeout=300)
    return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0


def test_one_rank_rccl_communicator_matches_plain_init(tmp_path):
    ref, one = str(tmp_path / "ref"), str(tmp_path / "rccl1")
    _run([sys.executable, SCRIPT, ref, "48"], {})
    _run([sys.executable, SCRIPT, one, "48"], {"PG_TEST_RCCL": "1"})
    a, b = np.load(ref + ".rank0.npz"), np.load(one + ".rank0.npz")
    assert int(a["unconverged"]) == 0 and int(b["unconverged"]) == 0
    assert int(b["degree"]) == int(a["degree"]) >= 2            # the preconditioned loop, through the all-reduce path
    assert int(a["iters"]) == int(b["iters"])
    assert np.linalg.norm(a["x"] - b["x"]) <= 1e-13 * np.linalg.norm(a["x"])


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_two_ranks_two_gpus_match_one_rank(tmp_path, overlap):
    if _gpu_count() < 2:
        pytest.skip("needs 2 GPUs: the two-stream RCCL halo exchange cannot run on a 1-GPU box")
    ref, two = str(tmp_path / "ref"), str(tmp_path / "two")
    _run([sys.executable, SCRIPT, ref, "64"], {})
    _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
          "--master-port", "29617", SCRIPT, two, "64"], {"PG_HALO_OVERLAP": overlap})
    a = np.load(ref + ".rank0.npz")
    parts = [np.load(f"{two}.rank{r}.npz") for r in range(2)]
    assert all(int(p["unconverged"]) == 0 for p in parts)
    assert all(int(p["n_ghost"]) > 0 for p in parts)            # both ranks exchange a halo
    assert sum(int(p["n_own"]) for p in parts) == int(a["n_own"])
    assert len({int(p["iters"]) for p in parts}) == 1           # identical scalars on every rank
    assert abs(int(parts[0]["iters"]) - int(a["iters"])) <= 2   # (sums in another order: the history may shift by one)
    x = parts[0]["x"] + parts[1]["x"]                           # owned planes are disjoint
    assert np.linalg.norm(x - a["x"]) <= 1e-11 * np.linalg.norm(a["x"])

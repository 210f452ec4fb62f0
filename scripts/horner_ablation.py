"""Ablation of the loop's dominant launch (Horner step, mode 8) on the compact loop matrix of the n^3 heat problem.

    python scripts/horner_ablation.py [n=512] [reps=100]

One process per kernel setting (PG_SPMV_XCD is read once per process): the parent starts a child for every setting and
prints one line each.  Chained launches (every launch reads what the one before wrote, the chain's input vector is a
third stream), timed between two HIP events, gaps included -- what a chain of the x-space loop looks like.
PG_SPMV_XCD >> 8 = diagnostic bits (wrong products): 1 skip the packed irregular rows (G), 2 skip U/P slices, 16 skip the
marching units, 64 no lateral lines in the units, 32 no stores in the units.
"""
import ctypes as C
import os
import subprocess
import sys

SETTINGS = [("all", 1), ("no G chunks", 257), ("no U/P slices", 513), ("units only", 769), ("no units", 4097),
            ("units only, no lateral lines", 769 + (64 << 8)), ("units only, no stores", 769 + (32 << 8))]


def child(n, reps):
    sys.path.insert(0, ".")
    import penguin.jl_amd as pj
    from penguin.jl_amd import _lib as L

    pj.init(0)
    mesh = pj.Mesh((n, n, n), (4.0, 4.0, 4.0))
    cap = pj.Capacity(pj.Sphere((2.01, 2.01, 2.01), 1.0), mesh)
    bcb = pj.BorderConditions({k: pj.Dirichlet(1.0) for k in ("left", "right", "top", "bottom")})
    ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
    dt = 0.75 * (4.0 / n) ** 2
    s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
    lib = L.lib()
    ms = C.c_double()
    out = []
    for name, bits in (("warm/chained", 1024), ("cold ring", 256)):
        L.check(lib.pg_solver_time_spmv(s._h, 0 | 2048 | 4096 | bits, reps, C.byref(ms)))
        out.append(f"{name} {ms.value * 1e3:6.1f} us")
    info = s.system_info(4 + 2)
    print(f"rows {info.rows_matrix} units {info.spmv_units} slices {info.spmv_slices} G rows {info.rows_irregular} "
          f"bytes {(info.spmv_bytes + 8 * info.rows_matrix) / 1e6:.1f} MB | " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    for name, xcd in SETTINGS:
        env = dict(os.environ, PG_SPMV_XCD=str(xcd))
        r = subprocess.run([sys.executable, __file__, "--child", str(n), str(reps)], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED: " + r.stderr.strip()[-300:])
        print(f"[{name:30s} PG_SPMV_XCD={xcd:6d}] {line}", flush=True)

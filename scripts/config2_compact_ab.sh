for f in 0.02 0.0005; do for sch in BE CN; do
PG_DIAG_ELIM_FRAC=$f python - $sch <<'PY'
import sys, time, ctypes as C, numpy as np, os
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import _lib as L
pj.init(0); lib = L.lib()
sch = sys.argv[1]
n=2048; N=2
mesh = pj.Mesh((n,)*N, (4.0,)*N, (0.0,)*N)
cap = pj.Capacity(pj.Sphere((2.01,)*N, 1.0), mesh)
bcb = pj.BorderConditions({k: pj.Dirichlet(0.0) for k in ("left","right","top","bottom")})
ph = pj.Phase(cap, pj.DiffusionOps(cap), 0.0, 1.0)
dt = 0.25*(4.0/n)**2
s = pj.DiffusionUnsteadyMono(ph, bcb, pj.Dirichlet(1.0), dt, None, "BE")
opts = L.pg_krylov_opts(0, 1e-12, 0.0, 0, 4, 1); si = L.pg_step_info()
L.check(lib.pg_solver_initial_solve(s._h, C.byref(opts), C.byref(si)))
run = L.pg_run_info(); code = 1 if sch=="CN" else 0
L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(code), C.byref(opts), 0, C.c_int64(5), 0, C.byref(run)))
L.check(lib.pg_device_synchronize()); t0=time.perf_counter()
L.check(lib.pg_solver_run(s._h, C.c_double(1e9), C.c_int32(code), C.byref(opts), 0, C.c_int64(200), 0, C.byref(run)))
L.check(lib.pg_device_synchronize()); el=time.perf_counter()-t0
print("frac", os.environ["PG_DIAG_ELIM_FRAC"], sch, "steps/s %.1f"%(200/el), "products/step %.2f"%(run.products/run.steps), "states/step %.2f"%(run.guess_states_read/run.steps), "compact", s.system_info(1).loop_is_compact, "state max", float(np.max(np.abs(s._fetch_state()))))
PY
done; done

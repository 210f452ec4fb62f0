import sys
sys.path.insert(0, ".")
import penguin.jl_amd as pj
from penguin.jl_amd import convergence as cv
pj.init(0)
r = cv.run_mesh_convergence_moving([8, 16, 32, 64, 128], verbose=True, output_dir="gpurun_out/moving_convergence")
print(r["orders"], r["pair_order_all"])
r = cv.run_mesh_convergence_moving([8, 16, 32, 64, 128], verbose=True, literal=False)
print("centres + reached time:", r["orders"], r["pair_order_all"])

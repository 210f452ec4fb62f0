"""The user script of /root/reference/examples/2D/SolidMoving/MovingHeat.jl:1-66 (heat conduction around a growing disc),
line for line against the Python mirror of the API -- what a Penguin.jl user changes: the body becomes a tagged moving body.

    python examples/moving_heat_2d.py            (needs a GPU and the built library)
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from penguin.jl_amd import (BorderConditions, Capacity, DiffusionOps, Dirichlet, Mesh, MovingDiffusionUnsteadyMono, MovingSphere,
                            Phase, SpaceTimeMesh, solve_MovingDiffusionUnsteadyMono_b)

# Define the mesh
nx, ny = 128, 128
lx, ly = 16.0, 16.0
x0, y0 = -8.0, -8.0
mesh = Mesh((nx, ny), (lx, ly), (x0, y0))

# Define the body: body = (x,y,t) -> -(sqrt((x-c1)^2 + (y-c2)^2) - (radius + c*sqrt(t)))     (fluid OUTSIDE the disc)
radius, center, c = 1.0, (0.0, 0.0), 1.56
body = MovingSphere(lambda t: center, lambda t: radius + c * np.sqrt(t), complement=True)

# Define the Space-Time mesh
dt = 1.0 * (lx / nx) ** 2
Tstart, Tend = 0.0, 1.0
STmesh = SpaceTimeMesh(mesh, [0.0, dt])

# Define the capacity and the operators
capacity = Capacity(body, STmesh)
operator = DiffusionOps(capacity)

# Boundary conditions, source term, diffusion coefficient, phase
bc = Dirichlet(0.0)
bc1 = Dirichlet(1.0)
bc_b = BorderConditions({"left": bc, "right": bc, "top": bc, "bottom": bc})
f = lambda x, y, z, t: 0.0
K = lambda x, y, z: 1.0
Fluide = Phase(capacity, operator, f, K)

# Initial condition
u0 = np.concatenate([np.zeros((nx + 1) * (ny + 1)), np.ones((nx + 1) * (ny + 1))])

# Define the solver, solve
solver = MovingDiffusionUnsteadyMono(Fluide, bc_b, bc1, dt, u0, mesh, "BE")
solve_MovingDiffusionUnsteadyMono_b(solver, Fluide, body, dt, Tstart, Tend, bc_b, bc1, mesh, "BE", method="bicgstab")

T = solver.states[-1][: (nx + 1) * (ny + 1)].reshape(ny + 1, nx + 1)
print(f"{len(solver.states)} states; bulk temperature on the line y = 0 at the end:")
j = ny // 2
print(np.array2string(T[j, nx // 2:nx // 2 + 40:4], precision=4))

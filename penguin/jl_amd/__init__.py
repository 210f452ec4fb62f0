"""penguin.jl_amd -- MI355X (gfx950) implementation of Penguin.jl's cut-cell operator assembly and
unsteady-diffusion time-step hot path behind the reference's Mesh/Capacity/Phase/DiffusionOps/Solver API.

All arithmetic runs in libpenguin_hip.so (hand-written HIP kernels, C ABI in include/penguin_hip.h).
There is no CPU fallback: importing works without a GPU (so the ABI can be inspected), any compute
call without the shared library or a HIP device raises PenguinHipError.
"""
from ._lib import PenguinHipError, config_string, device_name, finalize, get_unique_id, init, init_distributed, lib  # noqa: F401
from .api import (  # noqa: F401
    AdvectionDiffusionSteadyDiph, AdvectionDiffusionSteadyMono, AdvectionDiffusionUnsteadyDiph,
    AdvectionDiffusionUnsteadyMono, ConvectionOps, solve_AdvectionDiffusionUnsteadyDiph_b,
    solve_AdvectionDiffusionSteadyDiph_b, solve_AdvectionDiffusionSteadyMono_b, solve_AdvectionDiffusionUnsteadyMono_b,
    BorderConditions, Capacity, Circle, Ellipse, Ellipsoid, DarcyFlow, DarcyFlowUnsteady, DiffusionOps, DiffusionSteadyDiph, DiffusionSteadyMono, DiffusionUnsteadyDiph,
    DiffusionUnsteadyMono, Dirichlet,
    FluxJump, HalfSpace, InterfaceConditions, Mesh, MultiSphere, Neumann, Periodic, Phase, Robin, ScalarJump, Solver, Sphere,
    check_convergence, check_convergence_diph, div, grad, lp_norm, nC, solve_DarcyFlow_b, solve_DarcyFlowUnsteady_b, solve_darcy_velocity, solve_DiffusionSteadyDiph_b, solve_DiffusionSteadyMono_b,
    solve_DiffusionUnsteadyDiph_b, solve_DiffusionUnsteadyMono_b,
)
from .utils import (  # noqa: F401,E402
    cfl_restriction, initialize_poiseuille_velocity_field, initialize_radial_velocity_field,
    initialize_rotating_velocity_field, initialize_temperature_circle_b, initialize_temperature_function_b,
    initialize_temperature_square_b, initialize_temperature_uniform_b,
)
from .vtk import write_vtk  # noqa: F401,E402
from .moving import (  # noqa: F401,E402
    MovingCircle, MovingDiffusionUnsteadyDiph, MovingDiffusionUnsteadyMono, MovingHalfSpace, MovingSphere, SpaceTimeCapacity,
    SpaceTimeMesh, solve_MovingDiffusionUnsteadyDiph_b, solve_MovingDiffusionUnsteadyMono_b,
)

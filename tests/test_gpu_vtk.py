"""write_vtk (src/vtk.jl:11-159) on states computed by the HIP path: the files are written from `pj` solvers, read back,
and every point array is compared with the ORACLE's state of the same step reshaped the way the reference reshapes it
(`reshape(state[1:end÷2], n_1+1, n_2+1[, n_3+1])`, dim 1 fastest).  SURVEY.md 8(f).4."""
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from oracle import penguin_oracle as po
from penguin.jl_amd.vtk import read_vti, write_vtk
from tests.common import oracle_capacity_from_product, rel_l2
from tests.test_gpu_parity import HEAT_BORDERS, TOL_T, _mono_pair

pytestmark = pytest.mark.gpu


def _collection(path):
    sets = ET.parse(path).getroot().find("Collection").findall("DataSet")
    return [(float(s.attrib["timestep"]), s.attrib["file"]) for s in sets]


def test_config1_states_through_write_vtk(pj, tmp_path):
    """examples/2D/Diffusion/Heat.jl (config 1: 80^2, Dirichlet(sin(pi x) sin(pi y)) on the circle) ends with
    `write_vtk("heat", mesh, solver)` (:54): one .vti per state + heat.pvd, Temperature_b / Temperature_g."""
    n = 80
    M = (n + 1) ** 2
    g = lambda x, y, z, t: np.sin(np.pi * x) * np.sin(np.pi * y)
    u0 = np.concatenate([np.zeros(M), np.ones(M)])
    dt = 0.25 * (4.0 / n) ** 2
    (s, ph, bcb, bci), (so, oph, obcb, obci) = _mono_pair(
        pj, 2, n, 4.0, (2.01, 2.01), 1.0, pj.Dirichlet(g), po.Dirichlet(g),
        {k: pj.Dirichlet(0.0) for k in HEAT_BORDERS}, {k: po.Dirichlet(0.0) for k in HEAT_BORDERS}, dt, u0, "BE")
    pj.solve_DiffusionUnsteadyMono_b(s, ph, dt, 0.01, bcb, bci, "BE", reltol=1e-13)
    po.solve_DiffusionUnsteadyMono(so, oph, dt, 0.01, obcb, obci, "BE", method="\\")
    pvd = write_vtk(str(tmp_path / "heat"), ph.capacity.mesh, s)
    files = _collection(pvd)
    assert [t for t, _ in files] == [float(i) for i in range(1, len(so.states) + 1)]          # pvd[i] = vtk, i from 1
    assert len(files) == len(so.states) == 17
    for (_, name), ref in zip(files, so.states):
        ext, f = read_vti(str(tmp_path / name))
        assert ext == [0, n, 0, n, 0, 0]                                                        # vtk_grid(.., 0:1:n_1, 0:1:n_2)
        assert list(f) == ["Temperature_b", "Temperature_g"]
        Tb = ref[:M].reshape((n + 1, n + 1), order="F")                                         # reshape(state[1:end÷2], n_1+1, n_2+1)
        Tg = ref[M:].reshape((n + 1, n + 1), order="F")
        assert rel_l2(f["Temperature_b"][:, :, 0], Tb) <= TOL_T
        assert rel_l2(f["Temperature_g"][:, :, 0], Tg) <= TOL_T
    assert float(np.max(np.abs(so.states[-1][:M]))) > 1e-3                                      # a real field


def test_diphasic_3d_states_through_write_vtk(pj, tmp_path):
    """Unsteady diphasic diffusion in 3-D (16^3; sphere and its complement, ScalarJump / FluxJump): four point arrays per
    state, each reshaped to (n+1)^3 (src/vtk.jl, 3-D diphasic branch)."""
    n, Lx, c, r = 16, 4.0, (2.03, 1.98, 2.01), 1.1
    M = (n + 1) ** 3
    mesh, omesh = pj.Mesh((n,) * 3, (Lx,) * 3), po.Mesh((n,) * 3, (Lx,) * 3)
    cap1, cap2 = pj.Capacity(pj.Sphere(c, r), mesh), pj.Capacity(pj.Sphere(c, r, complement=True), mesh)
    oc1, oc2 = oracle_capacity_from_product(cap1, omesh), oracle_capacity_from_product(cap2, omesh)
    f = lambda x, y, z, t: 0.0
    D1 = lambda x, y, z: 1.0
    D2 = lambda x, y, z: 2.0
    p1, p2 = pj.Phase(cap1, pj.DiffusionOps(cap1), f, D1), pj.Phase(cap2, pj.DiffusionOps(cap2), f, D2)
    q1, q2 = po.Phase(oc1, po.make_diffusion_ops(oc1), f, D1), po.Phase(oc2, po.make_diffusion_ops(oc2), f, D2)
    ic = pj.InterfaceConditions(pj.ScalarJump(1.0, 0.5, 0.0), pj.FluxJump(1.0, 1.0, 0.0))
    oic = po.InterfaceConditions(po.ScalarJump(1.0, 0.5, 0.0), po.FluxJump(1.0, 1.0, 0.0))
    bcb, obcb = pj.BorderConditions({}), po.BorderConditions({})
    u0 = np.concatenate([np.ones(M), np.ones(M), np.zeros(M), np.zeros(M)])
    dt = 0.5 * (Lx / n) ** 2
    s = pj.DiffusionUnsteadyDiph(p1, p2, bcb, ic, dt, u0, "BE")
    so = po.DiffusionUnsteadyDiph(q1, q2, obcb, oic, dt, u0, "BE")
    # (backward Euler: under Crank-Nicolson this 3-D problem amplifies a 1e-11 difference between two previous states a
    #  thousandfold in one step -- tests/test_gpu_parity.py::test_diphasic_cn_3d_step_by_step -- and free-running
    #  trajectories cannot be compared to 1e-9)
    pj.solve_DiffusionUnsteadyDiph_b(s, p1, p2, dt, 3 * dt, bcb, ic, "BE", reltol=1e-13)
    po.solve_DiffusionUnsteadyDiph(so, q1, q2, dt, 3 * dt, obcb, oic, "BE", method="\\")
    pvd = write_vtk(str(tmp_path / "two_phase"), mesh, s)
    files = _collection(pvd)
    assert len(files) == len(so.states) >= 3
    names = ["Temperature_1_b", "Temperature_1_g", "Temperature_2_b", "Temperature_2_g"]
    for (_, name), ref in zip(files, so.states):
        ext, fld = read_vti(str(tmp_path / name))
        assert ext == [0, n, 0, n, 0, n]
        assert list(fld) == names
        for q, nm in enumerate(names):
            want = ref[q * M:(q + 1) * M].reshape((n + 1,) * 3, order="F")
            assert rel_l2(fld[nm], want) <= 1e-9, nm                                            # (the diphasic bar of the parity suite)
    last = so.states[-1]
    assert float(np.max(last[2 * M:3 * M])) > 1e-3                                              # heat has crossed into phase 2


def test_steady_poisson_through_write_vtk(pj, tmp_path):
    """Steady monophasic case: a single .vti (no collection), as `write_vtk` does for Steady solvers."""
    n = 24
    M = (n + 1) ** 2
    mesh, omesh = pj.Mesh((n, n), (4.0, 4.0)), po.Mesh((n, n), (4.0, 4.0))
    cap = pj.Capacity(pj.Sphere((2.0, 2.0), 1.0), mesh)
    ocap = oracle_capacity_from_product(cap, omesh)
    f = lambda x, y, z, t=0.0: 4.0
    D = lambda x, y, z: 1.0
    ph, oph = pj.Phase(cap, pj.DiffusionOps(cap), f, D), po.Phase(ocap, po.make_diffusion_ops(ocap), f, D)
    s = pj.DiffusionSteadyMono(ph, pj.BorderConditions({}), pj.Dirichlet(0.5))      # (a non-zero Tγ field to compare)
    so = po.DiffusionSteadyMono(oph, po.BorderConditions({}), po.Dirichlet(0.5))
    pj.solve_DiffusionSteadyMono_b(s, reltol=1e-13)
    po.solve_DiffusionSteadyMono(so, method="\\")
    path = write_vtk(str(tmp_path / "poisson"), mesh, s)
    assert path.endswith("poisson.vti")
    ext, fld = read_vti(path)
    assert ext == [0, n, 0, n, 0, 0]
    assert rel_l2(fld["Temperature_b"][:, :, 0], so.x[:M].reshape((n + 1, n + 1), order="F")) <= TOL_T
    assert rel_l2(fld["Temperature_g"][:, :, 0], so.x[M:].reshape((n + 1, n + 1), order="F")) <= TOL_T

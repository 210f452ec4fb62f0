"""write_vtk layout (src/vtk.jl:11-159): file set, array names, extents, point order.  Host-side code: no GPU needed."""
import types
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from penguin.jl_amd.vtk import read_vti, write_vtk


def _mesh(n):
    return types.SimpleNamespace(centers=[np.arange(k, dtype=float) for k in n])


def _solver(time_type, phase_type, x, states=()):
    return types.SimpleNamespace(time_type=time_type, phase_type=phase_type, equation_type="Diffusion", x=x, states=list(states))


def test_steady_monophasic_2d_layout(tmp_path):
    n = (4, 3)
    M = 5 * 4
    x = np.arange(2 * M, dtype=float)
    path = write_vtk(str(tmp_path / "heat"), _mesh(n), _solver("Steady", "Monophasic", x))
    assert path.endswith("heat.vti")
    ext, f = read_vti(path)
    assert ext == [0, 4, 0, 3, 0, 0]                                 # vtk_grid(filename, 0:1:n_1, 0:1:n_2)
    assert set(f) == {"Temperature_b", "Temperature_g"}
    Tb = x[:M].reshape((5, 4), order="F")                            # reshape(x[1:end÷2], n_1+1, n_2+1)
    assert np.array_equal(f["Temperature_b"][:, :, 0], Tb)
    assert f["Temperature_g"][2, 1, 0] == x[M + 2 + 1 * 5]          # idx = i + (j-1)(n_1+1)


def test_unsteady_diphasic_3d_collection(tmp_path):
    n = (2, 2, 3)
    M = 3 * 3 * 4
    states = [np.full(4 * M, 1.0), np.arange(4 * M, dtype=float)]
    path = write_vtk(str(tmp_path / "run"), _mesh(n), _solver("Unsteady", "Diphasic", states[-1], states))
    assert path.endswith("run.pvd")
    sets = ET.parse(path).getroot().find("Collection").findall("DataSet")
    assert [(float(s.attrib["timestep"]), s.attrib["file"]) for s in sets] == [(1.0, "run_1.vti"), (2.0, "run_2.vti")]
    ext, f = read_vti(str(tmp_path / "run_2.vti"))
    assert ext == [0, 2, 0, 2, 0, 3]
    assert list(f) == ["Temperature_1_b", "Temperature_1_g", "Temperature_2_b", "Temperature_2_g"]
    assert np.array_equal(f["Temperature_2_b"].ravel(order="F"), states[1][2 * M:3 * M])


def test_1d_and_errors(tmp_path):
    x = np.arange(2 * 6, dtype=float)
    ext, f = read_vti(write_vtk(str(tmp_path / "rod"), _mesh((5,)), _solver("Steady", "Monophasic", x)))
    assert ext == [0, 5, 0, 0, 0, 0] and np.array_equal(f["Temperature_g"].ravel(), x[6:])
    with pytest.raises(ValueError):
        write_vtk(str(tmp_path / "bad"), _mesh((5,)), _solver("Steady", "Triphasic", x))
    with pytest.raises(ValueError):
        write_vtk(str(tmp_path / "bad"), _mesh((2, 2, 2, 2)), _solver("Steady", "Monophasic", x))

"""Host helpers around the path (penguin/jl_amd/utils.py) against literal restatements of the reference's loops
(src/utils.jl:5-131, src/solver.jl:582-587): same index ranges, same quirks.  CPU only."""
import math

import numpy as np
import pytest

from penguin.jl_amd import utils as U


def _loop_circle(nx, ny, xc, yc, center, radius, value):
    To, Tg = np.zeros((nx + 1) * (ny + 1)), np.zeros((nx + 1) * (ny + 1))
    for j in range(1, ny + 1):                      # src/utils.jl:30-41 (1-based)
        for i in range(1, nx + 1):
            idx = i + (j - 1) * (nx + 1)
            if math.sqrt((xc[i - 1] - center[0]) ** 2 + (yc[j - 1] - center[1]) ** 2) <= radius:
                To[idx - 1] = value
                Tg[idx - 1] = value
    return To, Tg


def _loop_square(nx, ny, xc, yc, center, hw, value):
    To, Tg = np.zeros((nx + 1) * (ny + 1)), np.zeros((nx + 1) * (ny + 1))
    ci = next(k + 1 for k, v in enumerate(xc) if v >= center[0])
    cj = next(k + 1 for k, v in enumerate(yc) if v >= center[1])
    for j in range(max(cj - hw, 1), min(cj + hw, ny + 1) + 1):
        for i in range(max(ci - hw, 1), min(ci + hw, nx + 1) + 1):
            To[i + (j - 1) * (nx + 1) - 1] = value
            Tg[i + (j - 1) * (nx + 1) - 1] = value
    return To, Tg


@pytest.mark.parametrize("nx,ny", [(8, 8), (13, 7), (5, 20)])
def test_temperature_initialisers(nx, ny):
    xc = np.linspace(0.0, 2.0, nx + 1)
    yc = np.linspace(-1.0, 1.0, ny + 1)
    n = (nx + 1) * (ny + 1)
    To, Tg = np.full(n, -1.0), np.full(n, -1.0)
    U.initialize_temperature_uniform_b(To, Tg, 3.5)
    assert (To == 3.5).all() and (Tg == 3.5).all()
    for center, radius in (((1.0, 0.0), 0.6), ((0.1, -0.9), 0.35), ((5.0, 5.0), 0.2)):
        To, Tg = np.zeros(n), np.zeros(n)
        U.initialize_temperature_circle_b(To, Tg, xc, yc, center, radius, 2.0, nx, ny)
        ro, rg = _loop_circle(nx, ny, xc, yc, center, radius, 2.0)
        assert np.array_equal(To, ro) and np.array_equal(Tg, rg)
    for center, hw in (((1.0, 0.0), 2), ((0.05, -0.95), 3), ((1.9, 0.9), 1)):
        To, Tg = np.zeros(n), np.zeros(n)
        U.initialize_temperature_square_b(To, Tg, xc, yc, center, hw, 4.0, nx, ny)
        ro, rg = _loop_square(nx, ny, xc, yc, center, hw, 4.0)
        assert np.array_equal(To, ro) and np.array_equal(Tg, rg)
    with pytest.raises(ValueError):
        U.initialize_temperature_square_b(np.zeros(n), np.zeros(n), xc, yc, (9.0, 0.0), 1, 1.0, nx, ny)
    To, Tg = np.zeros(n), np.zeros(n)
    f = lambda x, y: math.sin(x) + 2.0 * y
    U.initialize_temperature_function_b(To, Tg, xc, yc, f, nx, ny)
    for j in range(ny + 1):
        for i in range(nx + 1):
            want = f(xc[i], yc[j]) if (i < nx and j < ny) else 0.0      # the padding layer is left alone
            assert To[i + j * (nx + 1)] == want and Tg[i + j * (nx + 1)] == want


def test_velocity_fields_and_cfl():
    nx, ny, lx, ly, x0, y0 = 6, 4, 2.0, 1.0, 0.5, -0.25
    ux, uy = U.initialize_rotating_velocity_field(nx, ny, lx, ly, x0, y0, 3.0)
    px, py = U.initialize_poiseuille_velocity_field(nx, ny, lx, ly, x0, y0)
    rx, ry = U.initialize_radial_velocity_field(nx, ny, lx, ly, x0, y0, (0.7, 0.1), 2.0)
    assert ux.shape == uy.shape == px.shape == rx.shape == ((nx + 1) * (ny + 1),)
    for j in range(ny + 1):
        for i in range(nx + 1):
            idx = i + j * (nx + 1)
            x, y = x0 + i * (lx / nx), y0 + j * (ly / ny)
            assert ux[idx] == -(y - ly / 2) * 3.0 and uy[idx] == (x - lx / 2) * 3.0       # centre (lx/2, ly/2): no x0, y0
            assert px[idx] == x * (1 - x) and py[idx] == 0.0
            r = math.sqrt((x - 0.7) ** 2 + (y - 0.1) ** 2)
            assert rx[idx] == (x - 0.7) / r * 2.0 and ry[idx] == (y - 0.1) / r * 2.0

    class M:      # the two fields cfl_restriction reads
        nodes = (np.array([0.125, 0.375, 0.625, 0.875, 1.125]),)
        centers = (np.array([0.0, 0.25, 0.5, 0.75]),)
    assert U.cfl_restriction(M, 0.5, 2.0) == 0.5 * 0.25 / 2.0

"""Host-side helpers on either side of the hot path -- the arrays a Penguin.jl script builds before it calls the
constructors (initial temperature, prescribed velocity fields for ConvectionOps, time-step restriction).  Mirrors of
src/utils.jl:5-131 and src/solver.jl:582-587 with their index ranges and quirks kept; plain numpy, no device work
(nothing here is on the compute path)."""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import numpy as np


def initialize_temperature_uniform_b(T0ₒ: np.ndarray, T0ᵧ: np.ndarray, value: float) -> None:
    """initialize_temperature_uniform!(T0ₒ, T0ᵧ, value) -- src/utils.jl:5-8."""
    T0ₒ[:] = value
    T0ᵧ[:] = value


def initialize_temperature_square_b(T0ₒ, T0ᵧ, x_coords: Sequence[float], y_coords: Sequence[float], center: Tuple[float, float],
                                    half_width: int, value: float, nx: int, ny: int) -> None:
    """initialize_temperature_square!(...) -- src/utils.jl:11-26: the square is centred on the first coordinate >= center
    and clipped to [1, n+1] (1-based), padding layer included."""
    x_coords, y_coords = np.asarray(x_coords), np.asarray(y_coords)
    ci = int(np.argmax(x_coords >= center[0])) + 1      # findfirst (1-based); the reference errors if there is none
    cj = int(np.argmax(y_coords >= center[1])) + 1
    if not (x_coords >= center[0]).any() or not (y_coords >= center[1]).any():
        raise ValueError("center lies beyond the last coordinate (findfirst returned nothing)")
    i_min, i_max = max(ci - half_width, 1), min(ci + half_width, nx + 1)
    j_min, j_max = max(cj - half_width, 1), min(cj + half_width, ny + 1)
    for j in range(j_min, j_max + 1):
        lo = (i_min - 1) + (j - 1) * (nx + 1)
        T0ₒ[lo:lo + (i_max - i_min + 1)] = value
        T0ᵧ[lo:lo + (i_max - i_min + 1)] = value


def initialize_temperature_circle_b(T0ₒ, T0ᵧ, x_coords, y_coords, center: Tuple[float, float], radius: float, value: float,
                                    nx: int, ny: int) -> None:
    """initialize_temperature_circle!(...) -- src/utils.jl:29-42: real cells only (i <= nx, j <= ny), distance <= radius."""
    x, y = np.asarray(x_coords)[:nx], np.asarray(y_coords)[:ny]
    inside = np.sqrt((x[None, :] - center[0]) ** 2 + (y[:, None] - center[1]) ** 2) <= radius
    for T in (T0ₒ, T0ᵧ):
        view = T[:(nx + 1) * (ny + 1)].reshape(ny + 1, nx + 1)
        view[:ny, :nx][inside] = value


def initialize_temperature_function_b(T0ₒ, T0ᵧ, x_coords, y_coords, func: Callable[[float, float], float], nx: int, ny: int) -> None:
    """initialize_temperature_function!(...) -- src/utils.jl:45-57: func(x, y) at the real cells."""
    for j in range(ny):
        for i in range(nx):
            v = func(x_coords[i], y_coords[j])
            T0ₒ[i + j * (nx + 1)] = v
            T0ᵧ[i + j * (nx + 1)] = v


def _node_grid(nx, ny, lx, ly, x0, y0):
    i, j = np.arange(nx + 1), np.arange(ny + 1)
    x = x0 + i * (lx / nx)
    y = y0 + j * (ly / ny)
    return np.broadcast_to(x[None, :], (ny + 1, nx + 1)), np.broadcast_to(y[:, None], (ny + 1, nx + 1))


def initialize_rotating_velocity_field(nx, ny, lx, ly, x0, y0, magnitude):
    """src/utils.jl:62-85: solid rotation about (lx/2, ly/2) -- the centre ignores x0, y0, as in the reference."""
    X, Y = _node_grid(nx, ny, lx, ly, x0, y0)
    return (-(Y - ly / 2) * magnitude).ravel().copy(), ((X - lx / 2) * magnitude).ravel().copy()


def initialize_poiseuille_velocity_field(nx, ny, lx, ly, x0, y0):
    """src/utils.jl:88-107: uₒx = x (1 - x) (sic: the profile is in x), uₒy = 0."""
    X, _ = _node_grid(nx, ny, lx, ly, x0, y0)
    return (X * (1 - X)).ravel().copy(), np.zeros((nx + 1) * (ny + 1))


def initialize_radial_velocity_field(nx, ny, lx, ly, x0, y0, center, magnitude):
    """src/utils.jl:110-131: unit radial field times magnitude (NaN at a node that coincides with the centre, as 0/0 is
    in the reference)."""
    X, Y = _node_grid(nx, ny, lx, ly, x0, y0)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2)
        return ((X - center[0]) / r * magnitude).ravel().copy(), ((Y - center[1]) / r * magnitude).ravel().copy()


def cfl_restriction(mesh, cfl: float, w: float) -> float:
    """cfl_restriction(mesh, cfl, w) -- src/solver.jl:582-587: Δt = cfl dx / w, dx from the node span of dimension 1."""
    dx = (mesh.nodes[0][-1] - mesh.nodes[0][0]) / len(mesh.centers[0])
    return cfl * dx / w

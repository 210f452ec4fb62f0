"""write_vtk(filename, mesh, solver) -- the reference's output format (src/vtk.jl:11-159), host side.

WriteVTK's `vtk_grid(filename, 0:1:n_1, 0:1:n_2, ...)` is a VTK ImageData file (`.vti`) with origin 0, unit spacing
and `n_d + 1` points per dimension; the state is split into `Temperature_b` / `Temperature_g` (monophasic) or
`Temperature_1_b`, `Temperature_1_g`, `Temperature_2_b`, `Temperature_2_g` (diphasic) point arrays, each reshaped to
`(n_1+1, n_2+1[, n_3+1])` in the same dim-1-fastest order the solver uses, which is VTK's own point order.  Unsteady
solvers write one file per entry of `solver.states` (`<filename>_<i>.vti`, i from 1) and a ParaView collection
`<filename>.pvd` whose timestep of file i is i (`pvd[i] = vtk`).

The files are plain VTK XML (inline base64 binary, little endian, UInt64 headers, uncompressed): every VTK reader
opens them; WriteVTK's default zlib-compressed appended layout is a different encoding of the same data set.
"""
from __future__ import annotations

import base64
from pathlib import Path
from typing import Dict, Sequence

import numpy as np

__all__ = ["write_vtk", "read_vti"]


def _fields(state: np.ndarray, phase_type: str) -> Dict[str, np.ndarray]:
    state = np.asarray(state, dtype=np.float64)
    if phase_type == "Monophasic":
        half = state.shape[0] // 2
        return {"Temperature_b": state[:half], "Temperature_g": state[half:]}
    if phase_type == "Diphasic":
        part = state.shape[0] // 4
        return {"Temperature_1_b": state[:part], "Temperature_1_g": state[part:2 * part],
                "Temperature_2_b": state[2 * part:3 * part], "Temperature_2_g": state[3 * part:]}
    raise ValueError("Combination of TimeType, PhaseType, and EquationType not supported.")


def _b64(a: np.ndarray) -> str:
    raw = np.ascontiguousarray(a, dtype="<f8").tobytes()
    return base64.b64encode(np.array([len(raw)], dtype="<u8").tobytes() + raw).decode("ascii")


def _write_vti(path: Path, dims: Sequence[int], fields: Dict[str, np.ndarray]) -> None:
    ext = [int(n) for n in dims] + [0] * (3 - len(dims))            # n_d cells => n_d + 1 points, extent 0..n_d
    npts = int(np.prod([n + 1 for n in dims]))
    whole = " ".join(f"0 {e}" for e in ext)
    out = ['<?xml version="1.0"?>',
           '<VTKFile type="ImageData" version="1.0" byte_order="LittleEndian" header_type="UInt64">',
           f'  <ImageData WholeExtent="{whole}" Origin="0 0 0" Spacing="1 1 1">',
           f'    <Piece Extent="{whole}">', '      <PointData>']
    for name, arr in fields.items():
        if arr.shape[0] != npts:
            raise ValueError(f"{name}: {arr.shape[0]} values for {npts} grid points")
        out.append(f'        <DataArray type="Float64" Name="{name}" format="binary">{_b64(arr)}</DataArray>')
    out += ['      </PointData>', '      <CellData/>', '    </Piece>', '  </ImageData>', '</VTKFile>', '']
    path.write_text("\n".join(out))


def write_vtk(filename: str, mesh, solver) -> str:
    """write_vtk(filename::String, mesh::AbstractMesh, solver::Solver).  Returns the path of the file a reader opens
    (`.vti` for steady solvers, `.pvd` for unsteady ones)."""
    dims = [len(c) for c in mesh.centers]
    if not 1 <= len(dims) <= 3:
        raise ValueError("Invalid number of dimensions for mesh.centers.")
    base = Path(filename)
    if solver.time_type == "Steady":
        path = base.with_name(base.name + ".vti")
        _write_vti(path, dims, _fields(solver.x, solver.phase_type))
        print(f"VTK file written : {path}")
        return str(path)
    if solver.time_type == "Unsteady":
        entries = []
        for i, state in enumerate(solver.states, start=1):
            piece = base.with_name(f"{base.name}_{i}.vti")
            _write_vti(piece, dims, _fields(state, solver.phase_type))
            entries.append((i, piece.name))
        pvd = base.with_name(base.name + ".pvd")
        lines = ['<?xml version="1.0"?>', '<VTKFile type="Collection" version="1.0" byte_order="LittleEndian">', '  <Collection>']
        lines += [f'    <DataSet timestep="{float(i)}" part="0" file="{name}"/>' for i, name in entries]
        lines += ['  </Collection>', '</VTKFile>', '']
        pvd.write_text("\n".join(lines))
        print(f"VTK file written : {pvd}")
        return str(pvd)
    raise ValueError("Combination of TimeType, PhaseType, and EquationType not supported.")


def read_vti(path: str):
    """Minimal reader of the files written above (tests, quick looks): (whole_extent, {name: array shaped (n+1,...)})."""
    import xml.etree.ElementTree as ET

    root = ET.parse(path).getroot()
    img = root.find("ImageData")
    ext = [int(v) for v in img.attrib["WholeExtent"].split()]
    shape = [ext[2 * d + 1] - ext[2 * d] + 1 for d in range(3)]
    fields = {}
    for da in img.find("Piece").find("PointData"):
        raw = base64.b64decode(da.text.strip())
        n = int(np.frombuffer(raw[:8], dtype="<u8")[0])
        a = np.frombuffer(raw[8:8 + n], dtype="<f8")
        fields[da.attrib["Name"]] = a.reshape(shape, order="F")      # dim 1 fastest
    return ext, fields

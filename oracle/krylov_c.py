"""ORACLE (test / benchmark infrastructure): ctypes wrapper of oracle/krylov_ref.c."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "libkrylov_ref.so"
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB.exists():
            subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
        _lib = C.CDLL(str(_LIB))
        for f in (_lib.krylov_ref_bicgstab, _lib.krylov_ref_bicgstab_poly, _lib.krylov_ref_cg):
            f.restype = C.c_int
    return _lib


def _args(A):
    rp = np.ascontiguousarray(A.indptr, dtype=np.int64)
    ci = np.ascontiguousarray(A.indices, dtype=np.int32)
    v = np.ascontiguousarray(A.data, dtype=np.float64)
    return rp, ci, v


def solve(A, b, method="bicgstab", reltol=1e-12, abstol=0.0, maxiter=10000, nthreads=1):
    """A: scipy CSR (square).  Returns (x, iterations, resnorm)."""
    n = A.shape[0]
    rp, ci, v = _args(A)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    res = C.c_double()
    fn = {"bicgstab": lib().krylov_ref_bicgstab, "cg": lib().krylov_ref_cg}[method]
    P = C.POINTER(C.c_double)
    it = fn(C.c_int64(n), rp.ctypes.data_as(C.POINTER(C.c_int64)), ci.ctypes.data_as(C.POINTER(C.c_int32)),
            v.ctypes.data_as(P), b.ctypes.data_as(P), x.ctypes.data_as(P), C.c_double(reltol), C.c_double(abstol),
            C.c_int(maxiter), C.c_int(nthreads), C.byref(res))
    return x, int(it), res.value


def solve_poly(A, b, m, g, x0=None, weights=None, reltol=1e-12, abstol=0.0, maxiter=10000, nthreads=1):
    """BiCGStab right-preconditioned with the degree-m Chebyshev polynomial of [1 - g, 1 + g] in product form, weighted
    convergence test after both halves of an iteration: the iteration of pg_krylov.hip (m < 2: plain).  Returns
    (x, iterations, weighted resnorm, products with A)."""
    n = A.shape[0]
    rp, ci, v = _args(A)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    P = C.POINTER(C.c_double)
    x0p = np.ascontiguousarray(x0, dtype=np.float64).ctypes.data_as(P) if x0 is not None else None
    wts = np.ascontiguousarray(weights, dtype=np.float64) if weights is not None else None
    res, nmv = C.c_double(), C.c_int64()
    it = lib().krylov_ref_bicgstab_poly(C.c_int64(n), rp.ctypes.data_as(C.POINTER(C.c_int64)), ci.ctypes.data_as(C.POINTER(C.c_int32)),
                                        v.ctypes.data_as(P), b.ctypes.data_as(P), x.ctypes.data_as(P), x0p,
                                        wts.ctypes.data_as(P) if wts is not None else None, C.c_int(m), C.c_double(g),
                                        C.c_double(reltol), C.c_double(abstol), C.c_int(maxiter), C.c_int(nthreads),
                                        C.byref(res), C.byref(nmv))
    return x, int(it), res.value, int(nmv.value)


def spmv(A, x, nthreads=1):
    n = A.shape[0]
    rp, ci, v = _args(A)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(n)
    P = C.POINTER(C.c_double)
    lib().krylov_ref_spmv(C.c_int64(n), rp.ctypes.data_as(C.POINTER(C.c_int64)), ci.ctypes.data_as(C.POINTER(C.c_int32)),
                          v.ctypes.data_as(P), x.ctypes.data_as(P), y.ctypes.data_as(P), C.c_int(nthreads))
    return y

"""ctypes binding of oracle/cpu_bdf.cpp - TEST INFRASTRUCTURE ONLY (see that file's header): the compiled CPU
baseline of the implicit solve (BDF + KLU-style sparse LU) and the generator of tight-tolerance truths."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkin_cpu_bdf.so")
_lib = None


class CpuParams(ctypes.Structure):
    _fields_ = [("tspan0", c_double), ("tspan1", c_double), ("abstol", c_double), ("reltol", c_double),
                ("adaptive_tols", c_int32), ("solve_chunks", c_int32), ("ban_negatives", c_int32), ("reserved", c_int32),
                ("solve_chunkstep", c_double), ("maxiters", c_int64), ("save_interval", c_double), ("dtmin", c_double),
                ("lu_reuse", c_double), ("step_thresh", c_double), ("lu_cache", c_int64)]


class CpuStats(ctypes.Structure):
    _fields_ = [(n, c_int64) for n in ("n_steps", "n_rejected", "n_rhs", "n_jac", "n_factor", "n_linsolve", "n_newton_fail",
                                        "n_chunks", "n_restarts", "n_retries", "n_resets")] + \
               [("final_abstol", c_double), ("final_reltol", c_double), ("wall_seconds", c_double)] + \
               [(n, c_int64) for n in ("lu_nnz", "lu_full", "lu_refactor")] + \
               [(n, c_double) for n in ("t_rhs", "t_jac", "t_factor", "t_solve")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s", "libkin_cpu_bdf.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = ctypes.CDLL(_LIB)
        P64, PD, P32 = POINTER(c_int64), POINTER(c_double), POINTER(c_int32)
        L.cpub_create.restype = c_void_p
        L.cpub_create.argtypes = [c_int64, c_int64, P64, P64, P64, P64, P64, P64]
        L.cpub_destroy.argtypes = [c_void_p]
        L.cpub_solve.argtypes = [c_void_p, POINTER(CpuParams), PD, PD, PD, PD, c_int64, P64, POINTER(CpuStats)]
        L.cpub_solution_copy.argtypes = [c_void_p, PD, PD]
        L.cpub_rhs.argtypes = [c_void_p, PD, PD, PD]
        L.cpub_jac_nnz.restype = c_int64
        L.cpub_jac_nnz.argtypes = [c_void_p]
        L.cpub_jac.argtypes = [c_void_p, PD, PD, P32, P32, PD]
        L.cpub_newton_solve.restype = c_int64
        L.cpub_newton_solve.argtypes = [c_void_p, c_double, PD, PD, PD, PD]
        _lib = L
    return _lib


def _pd(a):
    return None if a is None else a.ctypes.data_as(POINTER(c_double))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CpuSolver:
    """One network; `solve` mirrors oracle.bdf.solve_network_oracle for static rates (k0) or discrete updates
    (tstops + k_table[S][R])."""

    def __init__(self, net):
        arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in
                (net.reac_ptr, net.reac_idx, net.reac_sto, net.prod_ptr, net.prod_idx, net.prod_sto)]
        self.n, self.nr = int(net.n_species), int(net.n_reactions)
        self._h = lib().cpub_create(self.n, self.nr, *[a.ctypes.data_as(POINTER(c_int64)) for a in arrs])

    def close(self):
        if getattr(self, "_h", None):
            lib().cpub_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rhs(self, k, u):
        k, u = _f64(k), _f64(u)
        du = np.empty(self.n)
        lib().cpub_rhs(self._h, _pd(k), _pd(u), _pd(du))
        return du

    def jac(self, k, u):
        import scipy.sparse as sp
        k, u = _f64(k), _f64(u)
        nnz = lib().cpub_jac_nnz(self._h)
        cp, ri, v = np.empty(self.n + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz)
        lib().cpub_jac(self._h, _pd(k), _pd(u), cp.ctypes.data_as(POINTER(c_int32)), ri.ctypes.data_as(POINTER(c_int32)), _pd(v))
        return sp.csc_matrix((v, ri, cp), shape=(self.n, self.n))

    def newton_solve(self, c, k, u, b):
        k, u, b = _f64(k), _f64(u), _f64(b)
        x = np.empty(self.n)
        nnz = lib().cpub_newton_solve(self._h, float(c), _pd(k), _pd(u), _pd(b), _pd(x))
        return x, nnz

    def solve(self, params: dict, u0, k0=None, tstops=None, k_table=None):
        """params: the dict solve_network_oracle takes (lu_band / lu_slots: the LU cache, same defaults as the device and
        oracle/bdf.py; lu_band=0, lu_slots=0 switches it off). Returns (t, u, retcode, stats)."""
        tspan0, tspan1 = params["tspan"]
        si = params.get("save_interval", None)
        p = CpuParams(tspan0=tspan0, tspan1=tspan1, abstol=params.get("abstol", 1e-10), reltol=params.get("reltol", 1e-8),
                      adaptive_tols=int(params.get("adaptive_tols", True)), solve_chunks=int(params.get("solve_chunks", True)),
                      ban_negatives=int(params.get("ban_negatives", False)), reserved=0,
                      solve_chunkstep=params.get("solve_chunkstep", 1e-3), maxiters=int(params.get("maxiters", 100000)),
                      save_interval=-1.0 if si is None else si, dtmin=params.get("dtmin", 0.0),
                      lu_reuse=params.get("lu_band", 0.35), step_thresh=params.get("step_thresh", 0.0),
                      lu_cache=int(params.get("lu_slots", 128)))
        u0 = _f64(u0)
        n_stops = 0
        if tstops is not None and len(tstops):
            tstops, k_table = _f64(tstops), _f64(k_table)
            n_stops = len(tstops)
            assert k_table.shape == (n_stops, self.nr)
            k0 = k_table[0]
        else:
            tstops = k_table = None
        k0 = _f64(k0)
        assert len(u0) == self.n and len(k0) == self.nr
        n_saved, st = c_int64(0), CpuStats()
        rc = lib().cpub_solve(self._h, ctypes.byref(p), _pd(u0), _pd(k0), _pd(tstops), _pd(k_table), n_stops,
                              ctypes.byref(n_saved), ctypes.byref(st))
        t = np.empty(n_saved.value)
        u = np.empty((n_saved.value, self.n))
        lib().cpub_solution_copy(self._h, _pd(t), _pd(u))
        return t, u, rc, st.as_dict()

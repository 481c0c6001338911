"""ctypes binding of tests/native/libkin_resident_host.so - TEST INFRASTRUCTURE: the CPU replay of the resident integrator
(the controller of kinetica_jl_amd/csrc/resident_core.hpp over a sequential backend)."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np

from kinetica_jl_amd import capi

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")
_LIB = os.path.join(_HERE, "libkin_resident_host.so")
_lib = None


class ResResult(ctypes.Structure):
    _fields_ = [("retcode", c_int32), ("pad", c_int32), ("n_saved", c_int64), ("final_abstol", c_double), ("final_reltol", c_double)] + \
               [(n, c_int64) for n in ("n_steps", "n_rejected", "n_rhs", "n_jac", "n_factor", "n_linsolve", "n_newton_fail", "n_chunks",
                                        "n_restarts", "n_retries", "n_lu_reused", "n_bad_pivot", "n_lu_dropped")] + [("prof", c_int64 * 20)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "prof"}


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        capi.lib()                      # libkinetica_hip.so first (the replay links its host-side C++)
        if not os.path.exists(_LIB):
            build()
        L = ctypes.CDLL(_LIB)
        P64, PD = POINTER(c_int64), POINTER(c_double)
        L.res_host_create.restype = c_void_p
        L.res_host_create.argtypes = [c_int64, c_int64, P64, P64, P64, P64, P64, P64, c_int, POINTER(c_int), P64]
        L.res_host_destroy.argtypes = [c_void_p]
        L.res_host_set_arrhenius.argtypes = [c_void_p, PD, PD, c_double, c_double]
        L.res_host_newton_solve.argtypes = [c_void_p, c_double, PD, PD, PD, PD]
        L.res_host_rows.restype = c_int64
        L.res_host_rows.argtypes = [POINTER(capi.KinParams)]
        L.res_host_solve.argtypes = [c_void_p, POINTER(capi.KinParams), PD, PD, PD, PD, PD, c_int64, c_int, PD, PD, P64, POINTER(ResResult)]
        _lib = L
    return _lib


def _pd(a):
    return None if a is None else a.ctypes.data_as(POINTER(c_double))


class HostResident:
    def __init__(self, net, lu_opt=None):
        arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (net.reac_ptr, net.reac_idx, net.reac_sto, net.prod_ptr, net.prod_idx, net.prod_sto)]
        self.n, self.nr = int(net.n_species), int(net.n_reactions)
        info = np.zeros(8, np.int64)
        opt = (c_int * 5)(*(lu_opt or [0, 0, 0, 0, 0]))
        self._h = lib().res_host_create(self.n, self.nr, *[a.ctypes.data_as(POINTER(c_int64)) for a in arrs], 0, opt, info.ctypes.data_as(POINTER(c_int64)))
        assert self._h, "res_host_create failed"
        self.info = dict(ns=int(info[0]), m=int(info[1]), rounds=int(info[2]), solve_mode=int(info[3]), w_size=int(info[4]))

    def close(self):
        if getattr(self, "_h", None):
            lib().res_host_destroy(self._h)
            self._h = None

    __del__ = close

    def set_arrhenius(self, Ea, A, k_max=None, t_mult=1.0):
        Ea, A = np.ascontiguousarray(Ea, np.float64), np.ascontiguousarray(A, np.float64)
        lib().res_host_set_arrhenius(self._h, _pd(Ea), _pd(A), float("nan") if k_max is None else k_max, t_mult)

    def newton_solve(self, c, k, u, b):
        k, u, b = (np.ascontiguousarray(a, np.float64) for a in (k, u, b))
        x = np.empty(self.n)
        bad = lib().res_host_newton_solve(self._h, c, _pd(k), _pd(u), _pd(b), _pd(x))
        return x, bad

    def solve(self, pars, u0, k0=None, tstops=None, T_stops=None, k_table=None, n_slots=0):
        u0 = np.ascontiguousarray(u0, np.float64)
        rows = lib().res_host_rows(ctypes.byref(pars))
        t = np.empty(rows); u = np.empty((rows, self.n))
        ns = c_int64(0)
        res = ResResult()
        n_stops = 0 if tstops is None else len(tstops)
        f = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        k0, tstops, T_stops, k_table = f(k0), f(tstops), f(T_stops), f(k_table)
        rc = lib().res_host_solve(self._h, ctypes.byref(pars), _pd(u0), _pd(k0), _pd(tstops), _pd(T_stops), _pd(k_table), n_stops, n_slots,
                                  _pd(t), _pd(u), ctypes.byref(ns), ctypes.byref(res))
        return t[:ns.value], u[:ns.value], rc, res.as_dict()

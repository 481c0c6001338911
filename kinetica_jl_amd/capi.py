"""ctypes binding of libkinetica_hip.so (C ABI: include/kinetica_hip.h).

This is the Python twin of the Julia `ccall` shim shown in INTEGRATION.md. There is no CPU
fallback: a missing library or a missing HIP device raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KIN_LIB_PATH", os.path.join(_HERE, "libkinetica_hip.so"))   # (override: A/B builds in tools/)

KIN_OK, KIN_ERR_INVALID_ARG, KIN_ERR_UNSUPPORTED, KIN_ERR_DEVICE, KIN_ERR_SOLVE_FAILED, KIN_ERR_CAPACITY, KIN_ERR_STATE = range(7)
RETCODE_NAMES = {0: "Success", 1: "MaxIters", 2: "DtLessThanMin", 3: "Unstable"}

# every symbol include/kinetica_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "kin_network_create", "kin_network_destroy", "kin_network_sizes", "kin_last_error",
    "kin_set_rates", "kin_get_rates", "kin_set_arrhenius", "kin_rates_at", "kin_arrhenius_eval",
    "kin_rate_table", "kin_rhs", "kin_rhs_batched", "kin_rhs_batched_dev",
    "kin_jac_nnz", "kin_jac_pattern", "kin_jac_values", "kin_solve", "kin_solve_explicit", "kin_solve_continuous", "kin_solution_size",
    "kin_solution_copy", "kin_solution_max", "kin_integrator_init", "kin_integrator_init_continuous", "kin_integrator_step",
    "kin_integrator_state",
    "kin_newton_solve", "kin_device_count", "kin_set_device", "kin_version", "kin_solution_dot", "kin_rate_table_rows",
    "kin_solution_max_dev", "kin_rate_table_dev", "kin_rhs_block_dev",
    "kin_lib_layout", "kin_lib_layout_host", "kin_states_to_lib_dev", "kin_states_from_lib_dev", "kin_rates_to_lib_dev", "kin_rate_table_lib_dev",
    "kin_rhs_tiled_dev", "kin_rhs_batched_T_dev", "kin_rhs_batched_klib_dev", "kin_abi_version", "kin_struct_size",
    "kin_solve_ensemble", "kin_lu_analyze_host",
]
ABI_VERSION = 5   # include/kinetica_hip.h: KIN_ABI_VERSION this binding was written against


class KinParams(ctypes.Structure):
    """kin_params: mirror of ODESimulationParams (src/solving/params.jl:3-27)."""
    _fields_ = [("tspan0", c_double), ("tspan1", c_double), ("abstol", c_double), ("reltol", c_double),
                ("adaptive_tols", c_int32), ("update_tols", c_int32), ("solve_chunks", c_int32),
                ("ban_negatives", c_int32), ("solve_chunkstep", c_double), ("maxiters", c_int64),
                ("save_interval", c_double), ("dtmin", c_double)]


class KinStats(ctypes.Structure):
    _fields_ = [(n, c_int64) for n in ("n_steps", "n_rejected", "n_rhs", "n_jac", "n_factor", "n_linsolve",
                                        "n_newton_fail", "n_chunks", "n_restarts", "n_retries")] + \
               [("final_abstol", c_double), ("final_reltol", c_double), ("wall_seconds", c_double)] + \
               [(n, c_int64) for n in ("lu_dense_dim", "lu_sparse_rows", "lu_rounds", "lu_nnz", "n_lu_reused", "lu_slots", "n_bad_pivot", "n_lu_dropped")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class KineticaHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[kin status {code}] {msg}")
        self.code = code


_lib = None


def lib():
    """Load libkinetica_hip.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP extension is mandatory, there is no CPU fallback)")
        # PyTorch (device buffers, streams, torch.distributed in bench / tests) ships a HIP runtime of its own; a process that
        # loads this library first and torch second ends up with torch seeing no GPU. Loading torch first works both ways.
        import sys
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = ctypes.CDLL(LIB_PATH)
        P64, PD = POINTER(c_int64), POINTER(c_double)
        L.kin_struct_size.restype = c_int64
        L.kin_struct_size.argtypes = [c_int]
        if L.kin_abi_version() != ABI_VERSION or L.kin_struct_size(0) != ctypes.sizeof(KinParams) or \
                L.kin_struct_size(1) != ctypes.sizeof(KinStats):
            raise ImportError(f"{LIB_PATH}: ABI version {L.kin_abi_version()} / struct sizes do not match this binding "
                              f"(version {ABI_VERSION}); rebuild the library")
        L.kin_version.restype = c_char_p
        L.kin_last_error.restype = c_char_p
        L.kin_last_error.argtypes = [c_void_p]
        L.kin_network_create.argtypes = [c_int64, c_int64, P64, P64, P64, P64, P64, P64, c_int, POINTER(c_void_p)]
        L.kin_network_destroy.argtypes = [c_void_p]
        L.kin_network_sizes.argtypes = [c_void_p, P64, P64]
        L.kin_set_rates.argtypes = [c_void_p, PD]
        L.kin_get_rates.argtypes = [c_void_p, PD]
        L.kin_set_arrhenius.argtypes = [c_void_p, PD, PD, c_double, c_double]
        L.kin_rates_at.argtypes = [c_void_p, c_double, PD]
        L.kin_arrhenius_eval.argtypes = [PD, PD, c_int64, c_double, c_double, c_double, PD]
        L.kin_rate_table.argtypes = [c_void_p, PD, c_int64, PD]
        L.kin_rhs.argtypes = [c_void_p, PD, PD]
        L.kin_rhs_batched.argtypes = [c_void_p, c_int64, PD, PD, PD]
        L.kin_rhs_batched_dev.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        L.kin_jac_nnz.argtypes = [c_void_p, P64]
        L.kin_jac_pattern.argtypes = [c_void_p, P64, P64, c_int]
        L.kin_jac_values.argtypes = [c_void_p, PD, PD]
        L.kin_solve.argtypes = [c_void_p, POINTER(KinParams), PD, PD, PD, PD, c_int64, P64, POINTER(c_int32),
                                POINTER(KinStats)]
        L.kin_solve_explicit.argtypes = [c_void_p, POINTER(KinParams), PD, PD, PD, PD, c_int64, P64, POINTER(c_int32),
                                POINTER(KinStats)]
        L.kin_solve_continuous.argtypes = [c_void_p, POINTER(KinParams), PD, PD, PD, c_int64, P64, POINTER(c_int32),
                                           POINTER(KinStats)]
        L.kin_solve_ensemble.argtypes = [c_void_p, POINTER(KinParams), c_int64, PD, PD, PD, PD, PD, PD, c_int64, P64, PD, PD, P64,
                                         POINTER(c_int32), POINTER(KinStats)]
        L.kin_integrator_init.argtypes = [c_void_p, POINTER(KinParams), PD, PD, PD, PD, c_int64]
        L.kin_integrator_init_continuous.argtypes = [c_void_p, POINTER(KinParams), PD, PD, PD, c_int64]
        L.kin_integrator_step.argtypes = [c_void_p, c_int64, P64]
        L.kin_integrator_state.argtypes = [c_void_p, PD, PD, POINTER(c_int32), POINTER(KinStats)]
        L.kin_solution_size.argtypes = [c_void_p, P64, P64]
        L.kin_solution_copy.argtypes = [c_void_p, PD, PD]
        L.kin_solution_max.argtypes = [c_void_p, PD]
        L.kin_newton_solve.argtypes = [c_void_p, c_double, PD, PD, PD]
        L.kin_lib_layout.argtypes = [c_void_p, c_int, P64, P64, P64, POINTER(c_int32), P64]
        L.kin_states_to_lib_dev.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
        L.kin_states_from_lib_dev.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
        L.kin_rates_to_lib_dev.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
        L.kin_rate_table_lib_dev.argtypes = [c_void_p, PD, c_int64, c_void_p]
        L.kin_rhs_tiled_dev.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.kin_rhs_batched_T_dev.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        L.kin_rhs_batched_klib_dev.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        L.kin_device_count.argtypes = [POINTER(c_int)]
        L.kin_set_device.argtypes = [c_int]
        _lib = L
    return _lib


def _pd(a):
    return None if a is None else a.ctypes.data_as(POINTER(c_double))


def _p64(a):
    return a.ctypes.data_as(POINTER(c_int64))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = c_int(0)
    lib().kin_device_count(ctypes.byref(n))
    return n.value


def lib_layout_host(net, hubs=0):
    """The tiled sweep's library order of a FlatNetwork, computed on the host (no device): dict of the layout tables."""
    L = lib()
    arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (net.reac_ptr, net.reac_idx, net.reac_sto, net.prod_ptr,
                                                              net.prod_idx, net.prod_sto)]
    P32, PU64 = POINTER(c_int32), POINTER(ctypes.c_uint64)
    L.kin_lib_layout_host.argtypes = [c_int64, c_int64] + [POINTER(c_int64)] * 6 + [c_int, c_int, POINTER(c_int64), POINTER(c_int64),
                                                                                  POINTER(c_int64), PU64, P32, P32, P32, P32, P32, P32]
    info = np.zeros(10, np.int64)
    head = [int(net.n_species), int(net.n_reactions)] + [_p64(a) for a in arrs] + [0, int(hubs), _p64(info)]
    st = L.kin_lib_layout_host(*head, None, None, None, None, None, None, None, None, None)
    if st != KIN_OK:
        raise KineticaHipError(st, "network has no tiled layout")
    h, T, P, E, n_copy, BS, wbase, Q, k_len, has_singles = [int(x) for x in info]
    sp = np.empty(net.n_species, np.int64); slot = np.empty(net.n_reactions, np.int64)
    rec = np.empty(max(P, 1), np.uint64); rowtab = np.empty(max(2 * Q, 1), np.int32); seg_q = np.empty(T + 1, np.int32)
    woff = np.empty(T, np.int32); wcnt = np.empty(T, np.int32); copy_src = np.empty(max(n_copy, 1), np.int32)
    seg_k = np.empty(2 * T, np.int32)
    i32 = lambda a: a.ctypes.data_as(P32)
    st = L.kin_lib_layout_host(*head, _p64(sp), _p64(slot), rec.ctypes.data_as(PU64), i32(rowtab), i32(seg_q), i32(woff), i32(wcnt),
                               i32(copy_src), i32(seg_k))
    assert st == KIN_OK
    return dict(k_len=k_len, has_singles=bool(has_singles), seg_k=seg_k.reshape(T, 2), h=h, T=T, P=P, E=E, n_copy=n_copy, BS=BS, wbase=wbase, Q=Q, species_of_lib=sp, slot_of_reaction=slot,
                rec=rec[:P], rowtab=rowtab[:2 * Q].reshape(Q, 2), seg_q=seg_q, win_off=woff, win_cnt=wcnt, copy_src=copy_src[:n_copy])


def lu_analyze_host(net, hub_degree=0, max_rounds=0, max_tail_degree=0, max_degree=0, min_round=0):
    """Sizes of the symbolic Newton-matrix factorisation of a FlatNetwork, computed on the host (no device). Arguments left at 0
    take the library's defaults (lu.hpp: LUOptions)."""
    L = lib()
    arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (net.reac_ptr, net.reac_idx, net.reac_sto, net.prod_ptr,
                                                              net.prod_idx, net.prod_sto)]
    L.kin_lu_analyze_host.argtypes = [c_int64, c_int64] + [POINTER(c_int64)] * 6 + [c_int] * 6 + [POINTER(c_int64)]
    info = np.zeros(12, np.int64)
    st = L.kin_lu_analyze_host(int(net.n_species), int(net.n_reactions), *[_p64(a) for a in arrs], 0, int(hub_degree), int(max_rounds),
                               int(max_tail_degree), int(max_degree), int(min_round), _p64(info))
    if st != KIN_OK:
        raise KineticaHipError(st, "symbolic LU analysis failed")
    keys = ("ns", "m", "rounds", "nnzU", "nnzZ", "nnzV", "nnzLZ", "nnzNVU", "w_size", "fused_products", "plan_entries", "plan_tasks")
    return {k: int(v) for k, v in zip(keys, info)}


def arrhenius_eval(Ea, A, T, k_max=None, t_mult=1.0):
    """calculator(; T) on the device without a network handle."""
    Ea, A = _f64(Ea), _f64(A)
    out = np.empty(len(Ea))
    st = lib().kin_arrhenius_eval(_pd(Ea), _pd(A), len(Ea), float("nan") if k_max is None else k_max, t_mult, T, _pd(out))
    if st != KIN_OK:
        raise KineticaHipError(st, lib().kin_last_error(None).decode())
    return out


class HipNetwork:
    """Owning wrapper of a kin_network handle."""

    def __init__(self, n_species, reac_ptr, reac_idx, reac_sto, prod_ptr, prod_idx, prod_sto, index_base=0):
        arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (reac_ptr, reac_idx, reac_sto, prod_ptr, prod_idx, prod_sto)]
        self._h = c_void_p()
        st = lib().kin_network_create(int(n_species), len(arrs[0]) - 1, *[_p64(a) for a in arrs], index_base,
                                      ctypes.byref(self._h))
        if st != KIN_OK:
            raise KineticaHipError(st, lib().kin_last_error(None).decode())
        self.n = int(n_species)
        self.nr = len(arrs[0]) - 1

    @classmethod
    def from_flat(cls, net):
        return cls(net.n_species, net.reac_ptr, net.reac_idx, net.reac_sto, net.prod_ptr, net.prod_idx, net.prod_sto)

    def close(self):
        if getattr(self, "_h", None):
            lib().kin_network_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass

    def _chk(self, st):
        if st != KIN_OK:
            raise KineticaHipError(st, lib().kin_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    # --- rates -------------------------------------------------------------------------
    def set_rates(self, k):
        k = _f64(k)
        assert len(k) == self.nr
        self._chk(lib().kin_set_rates(self._h, _pd(k)))

    def get_rates(self):
        out = np.empty(self.nr)
        self._chk(lib().kin_get_rates(self._h, _pd(out)))
        return out

    def set_arrhenius(self, Ea, A, k_max=None, t_mult=1.0):
        Ea, A = _f64(Ea), _f64(A)
        assert len(Ea) == self.nr and len(A) == self.nr
        self._chk(lib().kin_set_arrhenius(self._h, _pd(Ea), _pd(A), float("nan") if k_max is None else k_max, t_mult))

    def rates_at(self, T):
        out = np.empty(self.nr)
        self._chk(lib().kin_rates_at(self._h, float(T), _pd(out)))
        return out

    def rate_table(self, T_stops, fetch=True):
        T_stops = _f64(T_stops)
        out = np.empty((len(T_stops), self.nr)) if fetch else None
        self._chk(lib().kin_rate_table(self._h, _pd(T_stops), len(T_stops), _pd(out)))
        return out

    # --- RHS / Jacobian ------------------------------------------------------------------
    def rhs(self, u):
        u = _f64(u)
        assert len(u) == self.n
        du = np.empty(self.n)
        self._chk(lib().kin_rhs(self._h, _pd(u), _pd(du)))
        return du

    def rhs_batched(self, u, k=None):
        u = _f64(u)
        B = u.shape[0]
        assert u.shape == (B, self.n)
        if k is not None:
            k = _f64(k)
            assert k.shape == (B, self.nr)
        du = np.empty((B, self.n))
        self._chk(lib().kin_rhs_batched(self._h, B, _pd(u), _pd(k), _pd(du)))
        return du

    def rhs_batched_dev(self, B, d_u, d_k, d_du, stream=0):
        """Device pointers (ints), state-major u[b][N], k[b][R] or 0, du[b][N]; only enqueues."""
        self._chk(lib().kin_rhs_batched_dev(self._h, int(B), c_void_p(d_u), c_void_p(d_k) if d_k else None,
                                            c_void_p(d_du), c_void_p(stream) if stream else None))

    # --- library order (tiled sweep) --------------------------------------------------------
    def lib_layout(self):
        """dict(k_len, species_of_lib[N], slot_of_reaction[R], identity, hubs, windows, records, entries, copies, block)."""
        k_len, ident = c_int64(0), c_int32(0)
        sp = np.empty(self.n, np.int64)
        slot = np.empty(self.nr, np.int64)
        info = np.zeros(6, np.int64)
        self._chk(lib().kin_lib_layout(self._h, 0, ctypes.byref(k_len), _p64(sp), _p64(slot), ctypes.byref(ident), _p64(info)))
        return dict(k_len=k_len.value, species_of_lib=sp, slot_of_reaction=slot, identity=bool(ident.value), hubs=int(info[0]),
                    windows=int(info[1]), records=int(info[2]), entries=int(info[3]), copies=int(info[4]), block=int(info[5]))

    def states_to_lib_dev(self, B, d_in, d_out, stream=0):
        self._chk(lib().kin_states_to_lib_dev(self._h, int(B), c_void_p(d_in), c_void_p(d_out), c_void_p(stream) if stream else None))

    def states_from_lib_dev(self, B, d_in, d_out, stream=0):
        self._chk(lib().kin_states_from_lib_dev(self._h, int(B), c_void_p(d_in), c_void_p(d_out), c_void_p(stream) if stream else None))

    def rates_to_lib_dev(self, B, d_k, d_k_lib, stream=0):
        self._chk(lib().kin_rates_to_lib_dev(self._h, int(B), c_void_p(d_k), c_void_p(d_k_lib), c_void_p(stream) if stream else None))

    def rate_table_lib_dev(self, T_stops, d_out):
        """Rate table for T_stops in library order straight into a device buffer [len(T_stops)][k_len]."""
        T_stops = _f64(T_stops)
        self._chk(lib().kin_rate_table_lib_dev(self._h, _pd(T_stops), len(T_stops), c_void_p(d_out)))

    def rhs_tiled_dev(self, B, d_u_lib, d_du_lib, d_k_lib=0, d_T=0, stream=0):
        """Batched RHS in library order; exactly one of d_k_lib / d_T (device pointers as ints); only enqueues."""
        self._chk(lib().kin_rhs_tiled_dev(self._h, int(B), c_void_p(d_u_lib), c_void_p(d_k_lib) if d_k_lib else None,
                                          c_void_p(d_T) if d_T else None, c_void_p(d_du_lib), c_void_p(stream) if stream else None))

    def rhs_batched_T_dev(self, B, d_u, d_T, d_du, stream=0):
        """Batched RHS on caller-order states with rate constants formed in the sweep from T[b]; only enqueues."""
        self._chk(lib().kin_rhs_batched_T_dev(self._h, int(B), c_void_p(d_u), c_void_p(d_T), c_void_p(d_du),
                                              c_void_p(stream) if stream else None))

    def rhs_batched_klib_dev(self, B, d_u, d_k_lib, d_du, stream=0):
        """Batched RHS on caller-order states u[b][N] -> du[b][N] with rate constants in the library's slot order k_lib[b][k_len]
        (from rate_table_lib_dev / rates_to_lib_dev); device pointers as ints; only enqueues."""
        self._chk(lib().kin_rhs_batched_klib_dev(self._h, int(B), c_void_p(d_u), c_void_p(d_k_lib), c_void_p(d_du),
                                                 c_void_p(stream) if stream else None))

    def jac_pattern(self, index_base=0):
        nnz = c_int64(0)
        self._chk(lib().kin_jac_nnz(self._h, ctypes.byref(nnz)))
        rowptr = np.empty(self.n + 1, np.int64)
        col = np.empty(nnz.value, np.int64)
        self._chk(lib().kin_jac_pattern(self._h, _p64(rowptr), _p64(col), index_base))
        return rowptr, col

    def jac_values(self, u):
        u = _f64(u)
        nnz = c_int64(0)
        self._chk(lib().kin_jac_nnz(self._h, ctypes.byref(nnz)))
        vals = np.empty(nnz.value)
        self._chk(lib().kin_jac_values(self._h, _pd(u), _pd(vals)))
        return vals

    # --- solve -----------------------------------------------------------------------------
    def solve(self, params: KinParams, u0, tstops=None, T_stops=None, k_table=None, explicit=False):
        """kin_solve (explicit=True: kin_solve_explicit, Dormand-Prince 5(4)) + kin_solution_copy.
        Returns (t[M], u[M][N], retcode, stats dict, status)."""
        u0 = _f64(u0)
        assert len(u0) == self.n
        n_stops = 0
        if tstops is not None:
            tstops = _f64(tstops)
            n_stops = len(tstops)
            if T_stops is not None:
                T_stops = _f64(T_stops)
                assert len(T_stops) == n_stops
            if k_table is not None:
                k_table = _f64(k_table)
                assert k_table.shape == (n_stops, self.nr)
        n_saved, rc, stats = c_int64(0), c_int32(0), KinStats()
        fn = lib().kin_solve_explicit if explicit else lib().kin_solve
        st = fn(self._h, ctypes.byref(params), _pd(u0), _pd(tstops), _pd(T_stops), _pd(k_table), n_stops,
                ctypes.byref(n_saved), ctypes.byref(rc), ctypes.byref(stats))
        if st not in (KIN_OK, KIN_ERR_SOLVE_FAILED):
            self._chk(st)
        t = np.empty(n_saved.value)
        u = np.empty((n_saved.value, self.n))
        if n_saved.value:
            self._chk(lib().kin_solution_copy(self._h, _pd(t), _pd(u)))
        return t, u, rc.value, stats.as_dict(), st

    def solve_ensemble(self, params: KinParams, u0, k=None, T=None, tstops=None, T_stops=None, k_table=None):
        """kin_solve_ensemble: K trajectories of this network in ONE launch (one workgroup per trajectory, resident on the GPU).
        u0[K][N]; k[K][R] or T[K] (or neither: the handle's current rates); tstops / T_stops / k_table are shared by the members.
        Returns (t[M], u[K][M][N], n_saved[K], retcodes[K], [stats dict] * K)."""
        u0 = np.ascontiguousarray(np.atleast_2d(_f64(u0)))
        K = u0.shape[0]
        assert u0.shape == (K, self.n)
        k = None if k is None else np.ascontiguousarray(_f64(k).reshape(K, self.nr))
        T = None if T is None else np.ascontiguousarray(_f64(T).reshape(K))
        n_stops = 0
        if tstops is not None:
            tstops = _f64(tstops); n_stops = len(tstops)
            T_stops = None if T_stops is None else _f64(T_stops)
            k_table = None if k_table is None else np.ascontiguousarray(_f64(k_table).reshape(n_stops, self.nr))
        rows = c_int64(0)
        fn = lib().kin_solve_ensemble
        self._chk(fn(self._h, ctypes.byref(params), K, _pd(u0), _pd(k), _pd(T), _pd(tstops), _pd(T_stops), _pd(k_table), n_stops,
                     ctypes.byref(rows), None, None, None, None, None))
        M = rows.value
        t = np.empty(M); u = np.empty((K, M, self.n)); ns = np.zeros(K, np.int64); rcs = np.zeros(K, np.int32)
        stats = (KinStats * K)()
        self._chk(fn(self._h, ctypes.byref(params), K, _pd(u0), _pd(k), _pd(T), _pd(tstops), _pd(T_stops), _pd(k_table), n_stops,
                     ctypes.byref(rows), _pd(t), _pd(u), ns.ctypes.data_as(POINTER(c_int64)), rcs.ctypes.data_as(POINTER(c_int32)), stats))
        return t, u, ns, rcs, [s_.as_dict() for s_ in stats]

    def solve_continuous(self, params: KinParams, u0, t_nodes, T_nodes):
        """kin_solve_continuous + kin_solution_copy: k(t) = Arrhenius(T(t)), T piecewise linear."""
        u0, t_nodes, T_nodes = _f64(u0), _f64(t_nodes), _f64(T_nodes)
        assert len(u0) == self.n and len(t_nodes) == len(T_nodes)
        n_saved, rc, stats = c_int64(0), c_int32(0), KinStats()
        st = lib().kin_solve_continuous(self._h, ctypes.byref(params), _pd(u0), _pd(t_nodes), _pd(T_nodes), len(t_nodes),
                                        ctypes.byref(n_saved), ctypes.byref(rc), ctypes.byref(stats))
        if st not in (KIN_OK, KIN_ERR_SOLVE_FAILED):
            self._chk(st)
        t = np.empty(n_saved.value)
        u = np.empty((n_saved.value, self.n))
        if n_saved.value:
            self._chk(lib().kin_solution_copy(self._h, _pd(t), _pd(u)))
        return t, u, rc.value, stats.as_dict(), st

    def integrator_init(self, params: KinParams, u0, tstops=None, T_stops=None, k_table=None):
        """kin_integrator_init: init(oprob, solver; kwargs...) without solve! (return_integrator=true)."""
        u0 = _f64(u0)
        assert len(u0) == self.n
        n_stops = 0
        if tstops is not None:
            tstops = _f64(tstops)
            n_stops = len(tstops)
            T_stops = _f64(T_stops) if T_stops is not None else None
            k_table = _f64(k_table) if k_table is not None else None
        self._chk(lib().kin_integrator_init(self._h, ctypes.byref(params), _pd(u0), _pd(tstops), _pd(T_stops),
                                            _pd(k_table), n_stops))

    def integrator_init_continuous(self, params: KinParams, u0, t_nodes, T_nodes):
        """kin_integrator_init_continuous: the integrator of a continuous-rate solve (k(t) = Arrhenius(T(t)))."""
        u0, t_nodes, T_nodes = _f64(u0), _f64(t_nodes), _f64(T_nodes)
        assert len(u0) == self.n and len(t_nodes) == len(T_nodes)
        self._chk(lib().kin_integrator_init_continuous(self._h, ctypes.byref(params), _pd(u0), _pd(t_nodes), _pd(T_nodes),
                                                       len(t_nodes)))

    def integrator_step(self, max_steps=1):
        """step!(integ) x max_steps (<= 0: solve!(integ)); returns the number of accepted steps taken."""
        n = c_int64(0)
        self._chk(lib().kin_integrator_step(self._h, int(max_steps), ctypes.byref(n)))
        return n.value

    def integrator_state(self, with_u=True):
        """(integ.t, integ.u or None, retcode, stats dict)."""
        t, rc, stats = c_double(0.0), c_int32(0), KinStats()
        u = np.empty(self.n) if with_u else None
        self._chk(lib().kin_integrator_state(self._h, ctypes.byref(t), _pd(u), ctypes.byref(rc), ctypes.byref(stats)))
        return t.value, u, rc.value, stats.as_dict()

    def newton_solve(self, c, u, b):
        """(I - c J(u)) x = b through the solver's on-device LU (diagnostic)."""
        u, b = _f64(u), _f64(b)
        x = np.empty(self.n)
        self._chk(lib().kin_newton_solve(self._h, float(c), _pd(u), _pd(b), _pd(x)))
        return x

    def solution_max_dev(self, d_out):
        """kin_solution_max into a device buffer (pointer as int) of N doubles."""
        self._chk(lib().kin_solution_max_dev(self._h, c_void_p(d_out)))

    def rate_table_dev(self, T_stops, d_out):
        """Rows of the rate table for T_stops straight into a device buffer [len(T_stops)][R]."""
        T_stops = _f64(T_stops)
        self._chk(lib().kin_rate_table_dev(self._h, _pd(T_stops), len(T_stops), c_void_p(d_out)))

    def rhs_block_dev(self, r_lo, r_hi, d_u, d_du, stream=0):
        """Partial RHS of reactions [r_lo, r_hi) on device buffers; only enqueues."""
        self._chk(lib().kin_rhs_block_dev(self._h, int(r_lo), int(r_hi), c_void_p(d_u), c_void_p(d_du),
                                          c_void_p(stream) if stream else None))

    def solution_dot(self, w):
        """sum_i w[i] u_i(t) at every saved time, reduced on the device (conserved quantities)."""
        w = _f64(w)
        assert len(w) == self.n
        n_saved = c_int64(0)
        self._chk(lib().kin_solution_size(self._h, ctypes.byref(n_saved), None))
        out = np.empty(n_saved.value)
        self._chk(lib().kin_solution_dot(self._h, _pd(w), _pd(out)))
        return out

    def rate_table_rows(self, rows):
        """Selected rows of the device-resident rate table."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((len(rows), self.nr))
        self._chk(lib().kin_rate_table_rows(self._h, _p64(rows), len(rows), _pd(out)))
        return out

    def solution_max(self):
        out = np.empty(self.n)
        self._chk(lib().kin_solution_max(self._h, _pd(out)))
        return out

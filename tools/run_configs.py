"""Runs the BASELINE.json configurations C2..C5 on one MI355X and prints one JSON
record per configuration: kernel timings (HIP events via torch), achieved GB/s against the
algorithmic bytes of SURVEY 8(d), solver statistics. Usage: python tools/run_configs.py [c2 c3 c4 c5 c5sweep table]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import narrow_k_variant, synthetic_crn


def kp(t1, chunk, save=None, chunks=True, dtmin=0.0):
    return capi.KinParams(tspan0=0.0, tspan1=t1, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0,
                          solve_chunks=int(chunks), ban_negatives=0, solve_chunkstep=chunk, maxiters=100000,
                          save_interval=-1.0 if save is None else save, dtmin=dtmin)


# The synthetic CRN's transient at t = 0 (u0 = 1 on the top hub species, barrierless reactions at the 1e12 cap, abstol
# 1e-10 on every empty species) needs first steps of ~1e-19 s at 500 K - below eps(1e-2 s) = 1.7e-18, the dtmin the
# reference hard-codes for 10 ms chunks (methods.jl:770): with it the solve ends in DtLessThanMin and adaptive_solve!'s
# retries cannot help (tests/test_gpu_configs.py shows exactly that). The ramp configurations therefore set dtmin.
RAMP_DTMIN = 1e-30


def timed(fn, reps=5):
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def sweep_record(name, N, R, B):
    net, Ea, A = synthetic_crn(N, R)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    h.rates_at(1000.0)
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(1)
    u = torch.pow(10.0, torch.rand((B, N), dtype=torch.float64, device=dev, generator=g) * 12 - 12)
    k = torch.rand((B, R), dtype=torch.float64, device=dev, generator=g) + 0.5
    du = torch.empty_like(u)
    torch.cuda.synchronize()      # the inputs were made on torch's default stream
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    dt = timed(lambda: h.rhs_batched_dev(B, u.data_ptr(), k.data_ptr(), du.data_ptr(), st.cuda_stream))
    alg = 20 * R + B * (8 * R + 16 * N)
    rec = {"config": name, "kernel": "batched RHS sweep", "N": N, "R": R, "B": B, "ms": dt * 1e3, "evals_per_s": B / dt,
           "algorithmic_GB": alg / 1e9, "GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12}
    h.close()
    return rec


def main():
    which = sys.argv[1:] or ["c2", "c3", "c5", "table", "c4"]
    class _Out(list):
        def append(self, r):
            print(json.dumps(r), flush=True)
    out = _Out()
    if "c2" in which:     # C2: 1k / 5k, narrow-k variant, RHS kernel + short solve
        out.append(sweep_record("C2", 1000, 5000, 4096))
        net, Ea, A = synthetic_crn(1000, 5000)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(narrow_k_variant(Ea), A, k_max=1e12)
        h.rates_at(1000.0)
        u0 = np.zeros(1000); u0[0] = 1.0
        h.solve(kp(2e-3, 1e-3), u0)
        t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(kp(0.02, 1e-3), u0); dt = time.perf_counter() - t0
        out.append({"config": "C2", "kernel": "kin_solve 20 chunks", "wall_s": dt, "retcode": rc, "stats": st})
        # the configuration's own wording: RHS kernel only, explicit solver (Dormand-Prince 5(4), kin_solve_explicit)
        # (k_max = 1e3 instead of 1e12: with the cap at 1e12 every rate constant sits at the cap and the fastest
        # time scale is 1e-12 s - no explicit method integrates that to 20 ms)
        h.set_arrhenius(narrow_k_variant(Ea), A, k_max=1e3)
        h.rates_at(1000.0)
        h.solve(kp(2e-3, 1e-3), u0, explicit=True)
        t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(kp(0.02, 1e-3), u0, explicit=True); dt = time.perf_counter() - t0
        out.append({"config": "C2", "kernel": "kin_solve_explicit 20 chunks (k_max 1e3)", "wall_s": dt, "retcode": rc, "stats": st,
                    "rhs_evals_per_s": st["n_rhs"] / dt})
        h.close()
    if "c3" in which:
        out.append(sweep_record("C3", 10000, 50000, 4096))
    if "c5sweep" in which:   # the C5 sweep alone (profiling passes)
        out.append(sweep_record("C5", 50000, 250000, 1024))
    if "c5" in which:     # C5: 50k / 250k: tiled sweep path + single-state kernels + a short solve
        out.append(sweep_record("C5", 50000, 250000, 1024))
        net, Ea, A = synthetic_crn(50000, 250000)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        h.rates_at(1000.0)
        u0 = np.zeros(50000); u0[0] = 1.0
        t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(kp(2e-3, 1e-3, dtmin=RAMP_DTMIN), u0); dt = time.perf_counter() - t0
        out.append({"config": "C5", "kernel": "kin_solve 2 chunks (includes symbolic analysis)", "wall_s": dt, "retcode": rc, "stats": st})
        t0 = time.perf_counter(); t, u, rc, st, _ = h.solve(kp(5e-3, 1e-3, dtmin=RAMP_DTMIN), u0); dt = time.perf_counter() - t0
        out.append({"config": "C5", "kernel": "kin_solve 5 chunks", "wall_s": dt, "retcode": rc, "stats": st})
        h.close()
    if "table" in which:  # M3: rate table S x R generated on the device (C4 size: 14001 x 50000 = 5.6 GB)
        N, R, S = 10000, 50000, 14001
        net, Ea, A = synthetic_crn(N, R)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        T = 500.0 + 50.0 * np.arange(S) * 1e-3
        h.rate_table(T, fetch=False)
        t0 = time.perf_counter()
        for _ in range(12):
            h.rate_table(T, fetch=False)
        dt = (time.perf_counter() - t0) / 12
        alg = 16 * R + 8 * S + 8 * S * R
        out.append({"config": "C4", "kernel": "rate_table_kernel (M3)", "S": S, "R": R, "ms": dt * 1e3, "algorithmic_GB": alg / 1e9,
                    "GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12, "launches": 12, "note": "average of 12 back-to-back calls, each including the host call, the T upload and a stream sync"})
        h.close()
    if "c4" in which:     # C4: 10k / 50k, ramp 500 -> 1200 K at 50 K/s, ts_update 1 ms, chunk 10 ms, save 5 ms (bounded prefix)
        N, R = 10000, 50000
        net, Ea, A = synthetic_crn(N, R)
        h = capi.HipNetwork.from_flat(net)
        h.set_arrhenius(Ea, A, k_max=1e12)
        t_end = float(os.environ.get("C4_TEND", "0.2"))
        T0 = float(os.environ.get("C4_T0", "500"))      # start of the ramp (500 K in the configuration; higher = a later part)
        tst = np.arange(int(round(t_end / 1e-3)) + 1) * 1e-3
        u0 = np.zeros(N); u0[0] = 1.0
        h.solve(kp(0.02, 1e-2, 5e-3, dtmin=RAMP_DTMIN), u0, tstops=tst[:21], T_stops=T0 + 50.0 * tst[:21])
        t0 = time.perf_counter()
        t, u, rc, st, _ = h.solve(kp(t_end, 1e-2, 5e-3, dtmin=RAMP_DTMIN), u0, tstops=tst, T_stops=T0 + 50.0 * tst)
        dt = time.perf_counter() - t0
        out.append({"config": "C4", "kernel": f"kin_solve ramp prefix (0, {t_end}) s of the 14 s run", "wall_s": dt, "retcode": rc,
                    "n_saved": len(t), "s_per_simulated_s": dt / t_end, "stats": st})
        h.close()


if __name__ == "__main__":
    main()

"""Times the tiled batched sweep (library order) on one MI355X: k-stream form, temperature form, the layout
conversions, and the caller-order sweep of the same network next to them. One JSON record per line.
Usage: python tools/tiled_bench.py [c3] [c5] [c2]   (under rocprofv3 --pmc: add `pmc` to run few launches only)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn


def ev_time(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def run(name, N, R, B, reps, with_old=True):
    net, Ea, A = synthetic_crn(N, R)
    h = capi.HipNetwork.from_flat(net)
    h.set_arrhenius(Ea, A, k_max=1e12)
    t0 = time.perf_counter(); lay = h.lib_layout(); t_layout = time.perf_counter() - t0
    dev = torch.device("cuda")
    g = torch.Generator(device=dev); g.manual_seed(1)
    u = torch.pow(10.0, torch.rand((B, N), dtype=torch.float64, device=dev, generator=g) * 12 - 12)
    T = torch.linspace(500.0, 1200.0, B, dtype=torch.float64, device=dev)
    kl = torch.empty((B, lay["k_len"]), dtype=torch.float64, device=dev)
    h.rate_table_lib_dev(T.cpu().numpy(), kl.data_ptr())
    du = torch.empty_like(u)
    torch.cuda.synchronize()      # the inputs were made on torch's default stream
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    s = st.cuda_stream
    alg = 20 * R + B * (8 * R + 16 * N)
    algT = 36 * R + B * (16 * N + 8)
    out = {"config": name, "N": N, "R": R, "B": B, "layout": {k: v for k, v in lay.items() if not hasattr(v, "shape")},
           "layout_build_s": t_layout}
    dt = ev_time(lambda: h.rhs_tiled_dev(B, u.data_ptr(), du.data_ptr(), d_k_lib=kl.data_ptr(), stream=s), reps)
    out["k_stream"] = {"ms": dt * 1e3, "evals_per_s": B / dt, "algorithmic_GB": alg / 1e9, "GBps": alg / dt / 1e9,
                       "frac_of_8TBps": alg / dt / 8e12}
    dt = ev_time(lambda: h.rhs_tiled_dev(B, u.data_ptr(), du.data_ptr(), d_T=T.data_ptr(), stream=s), reps)
    out["temperature_form"] = {"ms": dt * 1e3, "evals_per_s": B / dt, "algorithmic_GB_M1prime": algT / 1e9,
                               "GBps": algT / dt / 1e9, "frac_of_8TBps": algT / dt / 8e12}
    if "pmc" not in sys.argv:
        u2 = torch.empty_like(u)
        dt = ev_time(lambda: h.states_to_lib_dev(B, u.data_ptr(), u2.data_ptr(), stream=s), reps)
        out["states_to_lib"] = {"ms": dt * 1e3, "GBps": 16 * N * B / dt / 1e9}
        dt = ev_time(lambda: h.states_from_lib_dev(B, u.data_ptr(), u2.data_ptr(), stream=s), reps)
        out["states_from_lib"] = {"ms": dt * 1e3, "GBps": 16 * N * B / dt / 1e9}
        dt = ev_time(lambda: h.rhs_batched_T_dev(B, u.data_ptr(), T.data_ptr(), du.data_ptr(), stream=s), reps)
        out["temperature_form_caller_order"] = {"ms": dt * 1e3, "evals_per_s": B / dt}
        if with_old:
            k = torch.rand((B, R), dtype=torch.float64, device=dev, generator=g) + 0.5
            dt = ev_time(lambda: h.rates_to_lib_dev(B, k.data_ptr(), kl.data_ptr(), stream=s), reps)
            out["rates_to_lib"] = {"ms": dt * 1e3, "GBps": 16 * R * B / dt / 1e9}
            dt = ev_time(lambda: h.rhs_batched_dev(B, u.data_ptr(), k.data_ptr(), du.data_ptr(), s), reps)
            out["caller_order_sweep"] = {"ms": dt * 1e3, "GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12}
    print(json.dumps(out), flush=True)
    h.close()


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a != "pmc"] or ["c3", "c5"]
    reps = 3 if "pmc" in sys.argv else 10
    if "c2" in which:
        run("C2", 1000, 5000, 4096, reps)
    if "c3" in which:
        run("C3", 10000, 50000, 4096, reps)
    if "c5" in which:
        run("C5", 50000, 250000, 1024, reps)

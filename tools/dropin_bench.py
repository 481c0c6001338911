"""The drop-in batched sweep (kin_rhs_batched_klib_dev: states in the caller's species order, rate constants in slot order) at C5
(50k / 250k, B = 1024) or on the post-cutoff C3 network, timed with HIP events; under rocprofv3 (`pmc`) only a few calls are made.
Usage: python tools/dropin_bench.py [c5|cut] [pmc]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

cfg = "cut" if "cut" in sys.argv else "c5"
pmc = "pmc" in sys.argv
if cfg == "c5":
    N, R, B = 50000, 250000, 1024
    net, Ea, A = synthetic_crn(N, R)
else:
    N, R0, B = 10000, 50000, 4096
    net0, Ea0, A0 = synthetic_crn(N, R0)
    keep = np.sort(np.random.default_rng(0).choice(R0, int(0.7 * R0), replace=False))
    net, Ea, A = net0.subset(keep), Ea0[keep], A0[keep]
    R = net.n_reactions
h = capi.HipNetwork.from_flat(net)
h.set_arrhenius(Ea, A, k_max=1e12)
lay = h.lib_layout()
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
u = torch.pow(10.0, torch.rand((B, N), dtype=torch.float64, device=dev, generator=g) * 12 - 12)
T = torch.linspace(500.0, 1200.0, B, dtype=torch.float64, device=dev)
kl = torch.empty((B, lay["k_len"]), dtype=torch.float64, device=dev)
h.rate_table_lib_dev(T.cpu().numpy(), kl.data_ptr())
du = torch.empty_like(u)
torch.cuda.synchronize()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
s = st.cuda_stream
call = lambda: h.rhs_batched_klib_dev(B, u.data_ptr(), kl.data_ptr(), du.data_ptr(), s)
reps = 3 if pmc else 20
call(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    call()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
alg = 20 * R + B * (8 * R + 16 * N)
print(json.dumps({"config": cfg, "N": N, "R": R, "B": B, "identity": lay["identity"], "windows": lay["windows"], "k_len": lay["k_len"], "calls": reps + 1,
                  "ms_per_call": ms, "algorithmic_bytes": alg, "frac_of_8TBps": alg / (ms * 1e-3) / 8e12,
                  "expected_hbm_over_algorithmic": 1.0 if lay["identity"] else (alg + B * 32 * N) / alg}), flush=True)
h.close()

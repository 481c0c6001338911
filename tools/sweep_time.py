"""Times the caller-order batched sweep (kin_rhs_batched_dev) on the synthetic CRNs: python tools/sweep_time.py [c2] [c3] [c5]
(KIN_LIB_PATH selects another build of the library for A/B runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
import run_configs as rc

CASES = {"c2": (1000, 5000, 16384), "c3": (10000, 50000, 4096), "c5": (50000, 250000, 1024), "mid": (4000, 20000, 8192)}
for name in [a for a in sys.argv[1:] if a in CASES] or ["c3"]:
    N, R, B = CASES[name]
    net, Ea, A = synthetic_crn(N, R)
    h = capi.HipNetwork.from_flat(net)
    dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(1)
    u = torch.pow(10.0, torch.rand((B, N), dtype=torch.float64, device=dev, generator=g) * 12 - 12)
    k = torch.rand((B, R), dtype=torch.float64, device=dev, generator=g) + 0.5
    du = torch.empty_like(u)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    ts = [rc.timed(lambda: h.rhs_batched_dev(B, u.data_ptr(), k.data_ptr(), du.data_ptr(), st.cuda_stream), reps=20) for _ in range(3)]
    alg = 20 * R + B * (8 * R + 16 * N)
    print("%s N=%d R=%d B=%d: %s ms  best %.1f %% of 8 TB/s" % (name, N, R, B, " ".join("%.4f" % (t * 1e3) for t in ts), alg / min(ts) / 8e10), flush=True)
    h.close()

"""Sweep timing on a network whose forward/reverse pairing is broken (what the low-k cutoff leaves behind):
the C3 CRN with a random 30 % of the reactions removed. Usage: python tools/unpaired_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
import run_configs as rc
N, R, B = 10000, 50000, 4096
net, Ea, A = synthetic_crn(N, R)
keep = np.sort(np.random.default_rng(0).choice(R, int(0.7 * R), replace=False))
net2 = net.subset(keep)
R2 = net2.n_reactions
h = capi.HipNetwork.from_flat(net2)
h.set_rates(np.ones(R2))
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(1)
u = torch.pow(10.0, torch.rand((B, N), dtype=torch.float64, device=dev, generator=g) * 12 - 12)
k = torch.rand((B, R2), dtype=torch.float64, device=dev, generator=g) + 0.5
du = torch.empty_like(u)
torch.cuda.synchronize()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
dt = rc.timed(lambda: h.rhs_batched_dev(B, u.data_ptr(), k.data_ptr(), du.data_ptr(), st.cuda_stream))
alg = 20 * R2 + B * (8 * R2 + 16 * N)
print("unpaired (70 %% of C3's reactions): %.3f ms  %.0f GB/s  %.1f %% of 8 TB/s" % (dt * 1e3, alg / dt / 1e9, alg / dt / 8e10))
# the same network in library order (k as the library-order rate table writes it): tiled sweep, k-stream and temperature form
h.set_arrhenius(Ea[keep], A[keep], k_max=1e12)
lay = h.lib_layout()
T = torch.linspace(500.0, 1200.0, B, dtype=torch.float64, device=dev)
kl = torch.empty((B, lay["k_len"]), dtype=torch.float64, device=dev)
h.rate_table_lib_dev(T.cpu().numpy(), kl.data_ptr())
dt = rc.timed(lambda: h.rhs_tiled_dev(B, u.data_ptr(), du.data_ptr(), d_k_lib=kl.data_ptr(), stream=st.cuda_stream))
print("  library order, k stream (%d records for %d reactions): %.3f ms  %.0f GB/s  %.1f %% of 8 TB/s" % (lay["records"], R2, dt * 1e3, alg / dt / 1e9, alg / dt / 8e10))
dt = rc.timed(lambda: h.rhs_tiled_dev(B, u.data_ptr(), du.data_ptr(), d_T=T.data_ptr(), stream=st.cuda_stream))
print("  library order, temperature form: %.3f ms  %.2f M evals/s" % (dt * 1e3, B / dt / 1e6))

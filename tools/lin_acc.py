"""Accuracy of the device's Newton-matrix solve (kin_newton_solve: unpivoted sparse rounds, explicit triangular inverses, dense
Schur inverse) against SuperLU on the same matrices: residuals and distance of the solutions. Usage: python tools/lin_acc.py [N] [seed]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
from oracle import oracle as orc, bdf as obdf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
R = 5 * N
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
net, Ea, A = synthetic_crn(N, R, seed=SEED)
on = orc.OracleNetwork.from_flat(net)
k = orc.arrhenius(Ea, A, 1000.0, k_max=1e12)
h = capi.HipNetwork.from_flat(net); h.set_rates(k)
# a relaxed state: integrate 1 chunk on GPU
u0 = np.zeros(N); u0[0] = 1.0
p = capi.KinParams(tspan0=0.0, tspan1=1e-3, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0, solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0)
t, u, rc, st, status = h.solve(p, u0)
print("solve", rc, st["n_steps"], st["n_factor"], st["n_newton_fail"], st["n_rejected"])
us = u[-1]
rng = np.random.default_rng(0)
for state_name, uu in (("relaxed", us), ("u0", u0), ("random", 10.0 ** rng.uniform(-12, 0, N))):
    J = on.jac(k, uu)
    for c in (1e-9, 1e-6, 1e-4, 1e-3):
        M = (sp.identity(N) - c * J).tocsc()
        lu = spla.splu(M, permc_spec="MMD_AT_PLUS_A")
        for bname, b in (("rand", rng.standard_normal(N)), ("f", c * on.rhs(k, uu))):
            xs = lu.solve(b)
            xg = h.newton_solve(c, uu, b)
            rs = np.linalg.norm(M @ xs - b) / np.linalg.norm(b + 1e-300)
            rg = np.linalg.norm(M @ xg - b) / np.linalg.norm(b + 1e-300)
            print(f"{state_name:8s} c={c:7.0e} b={bname:4s} resid SuperLU {rs:.2e} GPU {rg:.2e}  relerr(gpu vs slu) {np.linalg.norm(xg-xs)/np.linalg.norm(xs+1e-300):.2e}  |x| {np.linalg.norm(xs):.2e}")

"""The three forms of kin_solve_ensemble against each other around their cut-overs (KIN_ENSEMBLE_ROUTE forces one):
python tools/ensemble_route_crossover.py [small|large]   - one child process per route. small: 300-1000 species, the bench's
ensemble workload (20 chunks, members at 900-1300 K); large: 1000-10000 species, 2 chunks, members at 950-1150 K."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "small"
if "ROUTE" not in os.environ:
    for route in (("resident", "threads", "lockstep") if which == "small" else ("threads", "lockstep")):
        subprocess.run([sys.executable, os.path.abspath(__file__), which], env={**os.environ, "KIN_ENSEMBLE_ROUTE": route, "ROUTE": route}, check=False)
    sys.exit(0)
import numpy as np
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn
cases = ((300, (16, 64)), (500, (16, 64, 128)), (700, (16, 64, 128)), (1000, (16, 64, 128))) if which == "small" else \
        ((1000, (16, 30)), (3000, (16, 30)), (10000, (16, 24, 30)))
for N, Ks in cases:
    net, Ea, A = synthetic_crn(N, 5 * N)
    u0 = np.zeros(N); u0[0] = 1.0
    nch = 20 if which == "small" else 2
    p = capi.KinParams(tspan0=0.0, tspan1=1e-3 * nch, abstol=1e-10, reltol=1e-8, adaptive_tols=1, update_tols=0, solve_chunks=1, ban_negatives=0,
                       solve_chunkstep=1e-3, maxiters=100000, save_interval=-1.0, dtmin=0.0)
    h = capi.HipNetwork.from_flat(net); h.set_arrhenius(Ea, A, k_max=1e12)
    for K in Ks:
        U0 = np.tile(u0, (K, 1)); T = np.linspace(900.0, 1300.0, K) if which == "small" else np.linspace(950.0, 1150.0, K)
        try:
            h.solve_ensemble(p, U0, T=T)
            walls = []
            for _ in range(2):
                t0 = time.perf_counter(); h.solve_ensemble(p, U0, T=T); walls.append(time.perf_counter() - t0)
            print(json.dumps({"N": N, "route": os.environ["ROUTE"], "K": K, "chunks": nch, "wall": round(min(walls), 4), "solves_per_s": round(K / min(walls), 1)}), flush=True)
        except Exception as e:   # noqa: BLE001 - a route that does not take the size is a row of the table too
            print(json.dumps({"N": N, "route": os.environ["ROUTE"], "K": K, "error": str(e)[:120]}), flush=True)
    h.close()

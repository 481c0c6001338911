"""Dense Schur block of the Newton-matrix factorisation against the elimination options (lu.hpp: LUOptions), on the host
(kin_lu_analyze_host - no device): what VERDICT r03 item 5(a) asked to be measured. Writes profiles/r04_dense_block_options.json.
Usage: python tools/dense_block_table.py [N=10000]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kinetica_jl_amd import capi
from kinetica_jl_amd.synth import synthetic_crn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
net, _, _ = synthetic_crn(N, 5 * N)
rows = []
SOLVER = dict(hub_degree=32, max_rounds=16, max_tail_degree=32, max_degree=400)     # what solver.cpp sets from 4 000 species up
cases = [dict(SOLVER)]
for key, vals in (("hub_degree", (16, 24, 48, 64, 128, 1024)), ("max_rounds", (8, 24, 32)), ("max_tail_degree", (8, 16, 64, 128, 256)),
                  ("max_degree", (100, 200, 800, 1600, 8000)), ("min_round", (1, 8, 32))):
    for v in vals:
        cases.append({**SOLVER, key: v})
cases += [dict(hub_degree=256, max_rounds=32, max_tail_degree=128, max_degree=4000),
          dict(hub_degree=1024, max_rounds=32, max_tail_degree=256, max_degree=8000),
          dict(hub_degree=1024, max_rounds=32, max_tail_degree=256, max_degree=8000, min_round=1)]
for c in cases:
    t0 = time.perf_counter()
    try:
        info = capi.lu_analyze_host(net, **c)
        rows.append({"options": c, "dense_block": info["m"], "sparse_pivots": info["ns"], "rounds": info["rounds"],
                     "nnzLZ": info["nnzLZ"], "nnzNVU": info["nnzNVU"], "doubles_per_factorisation": info["w_size"],
                     "fused_products": info["fused_products"], "analysis_s": round(time.perf_counter() - t0, 3)})
    except Exception as e:            # an option set whose fused products overflow the cap is refused by the analysis
        rows.append({"options": c, "error": str(e)})
    print(json.dumps(rows[-1]), flush=True)
out = {"network": f"synthetic CRN {N} species / {5 * N} reactions (seed 12345)",
       "command": "python tools/dense_block_table.py " + str(N), "rows": rows}
if N == 10000:
    json.dump(out, open(os.path.join(ROOT, "profiles", "r04_dense_block_options.json"), "w"), indent=1)

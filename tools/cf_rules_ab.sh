#!/bin/bash
# A/B of the convergence-failure rules of the host-driven integrator: C3 as one integration over (0, 1) s,
# C3 100 chunks, and the 140- / 208-solve robustness sweeps, per setting of KIN_ETACF / KIN_CF_RESET / KIN_CF_GROWTH_CAP.
# Usage: tools/cf_rules_ab.sh "<env settings>" [sweeps]      e.g. tools/cf_rules_ab.sh "KIN_ETACF=0.5 KIN_CF_RESET=1 KIN_CF_GROWTH_CAP=0" sweeps
set -u
export KIN_RESIDENT=0
echo "=== $1"
env $1 python tools/c3_complete_trace.py 1.0 2>/dev/null | tail -1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); s = r['stats']; print('C3 complete (0,1): wall', round(r['wall_s'], 3), 'rc', r['rc'], {k: s[k] for k in ('n_steps', 'n_rejected', 'n_factor', 'n_newton_fail')})"
env $1 python tools/solve_stats.py 10000 50000 100 2>/dev/null | tail -1 | cut -c1-260
if [ "${2:-}" = "sweeps" ]; then
  env $1 python tools/robustness_sweep.py wide 2>/dev/null | python -c "
import sys, json
n = bad = ret = 0; steps = 0
for l in sys.stdin:
    if not l.startswith('{'): continue
    r = json.loads(l); n += 1; bad += r['rc'] != 0; ret += r['retries']; steps += r['steps']
print('sweep wide:', n, 'solves, failures', bad, 'retries', ret, 'steps', steps)"
  env $1 python tools/robustness_sweep.py wide2 2>/dev/null | python -c "
import sys, json
n = bad = ret = 0; steps = 0
for l in sys.stdin:
    if not l.startswith('{'): continue
    r = json.loads(l); n += 1; bad += r['rc'] != 0; ret += r['retries']; steps += r['steps']
print('sweep wide2:', n, 'solves, failures', bad, 'retries', ret, 'steps', steps)"
fi

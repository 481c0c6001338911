#!/bin/bash
# Round 5's validation battery as it was run on the GPU box (one gpurun call per block; each block writes under gpurun_out/):
#   full GPU suite + smoke + bench line;  parity statistics under three reuse bands;  robustness sweeps;  the whole 14 s ramp;
#   resident / lockstep sweeps;  tight-tolerance comparison;  drop-in sweep counters;  the switch matrix.
# Usage on the GPU box: bash tools/r05_validation.sh [suite|stats|sweeps|ramp|resident|dropin|switches]   (default: suite)
set -e
B=${1:-suite}
mkdir -p gpurun_out/r05
case $B in
  suite)    python -m pytest tests -m gpu -q > gpurun_out/r05/pytest_all.txt 2>&1; tail -n 3 gpurun_out/r05/pytest_all.txt
            python __graft_entry__.py smoke 2>&1 | tail -n 1
            python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; tail -c 900 gpurun_out/r05/bench_default.json ;;
  stats)    LU_BAND=0.32,0.35,0.38 python tools/config_stats.py > gpurun_out/r05/config_stats.jsonl ;;
  sweeps)   for s in "" big wide wide2; do python tools/robustness_sweep.py $s > gpurun_out/r05/robust_${s:-default}.txt 2>&1; tail -n 1 gpurun_out/r05/robust_${s:-default}.txt; done
            python tools/robustness_continuous.py > gpurun_out/r05/robust_continuous.txt 2>&1; tail -n 1 gpurun_out/r05/robust_continuous.txt ;;
  ramp)     C4_TEND=14 KIN_PROGRESS=30 python tools/run_configs.py c4 > gpurun_out/r05/c4_full_run.json ;;
  resident) python tools/robustness_lockstep.py > gpurun_out/r05/robustness_lockstep.jsonl; python tools/robustness_resident.py > gpurun_out/r05/robustness_resident.jsonl
            python tools/tight_tol_truth.py > gpurun_out/r05/tight_tol_truth.jsonl ;;
  dropin)   bash tools/pmc_dropin.sh c5; bash tools/pmc_dropin.sh cut ;;
  switches) bash tools/switch_matrix.sh > gpurun_out/r05/switch_matrix.txt ;;
esac

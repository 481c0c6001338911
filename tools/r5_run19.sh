mkdir -p gpurun_out/r5r
KIN_RESIDENT_PROFILE=1 python tools/solve_stats.py 300 1500 20 > gpurun_out/r5r/resident_300.json 2> gpurun_out/r5r/resident_phase_300.txt
python tools/resident_crossover.py > gpurun_out/r5r/resident_vs_host.jsonl 2> gpurun_out/r5r/resident_vs_host.err
python tools/tight_tol_truth.py > gpurun_out/r5r/tight_tol_truth.jsonl 2> gpurun_out/r5r/tight.err; tail -n 1 gpurun_out/r5r/tight_tol_truth.jsonl | cut -c1-600
python tools/ensemble_resident_scaling.py > gpurun_out/r5r/ensemble_resident_scaling.jsonl 2> gpurun_out/r5r/ens.err
python tools/robustness_lockstep.py > gpurun_out/r5r/robustness_lockstep.jsonl 2> gpurun_out/r5r/robustness_lockstep.err; tail -n 1 gpurun_out/r5r/robustness_lockstep.jsonl | cut -c1-300
python tools/robustness_resident.py > gpurun_out/r5r/robustness_resident.jsonl 2> gpurun_out/r5r/robustness_resident.err; tail -n 1 gpurun_out/r5r/robustness_resident.jsonl | cut -c1-400
python -m pytest tests -m gpu -q > gpurun_out/r5r/pytest_all.txt 2>&1; tail -n 3 gpurun_out/r5r/pytest_all.txt

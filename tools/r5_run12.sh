# round 5: the numbers that changed with the (re)initialisation rules - robustness sweeps, the whole C4 ramp, resident profiles
mkdir -p gpurun_out/r5k
KIN_RESIDENT_PROFILE=1 python tools/solve_stats.py 300 1500 20 > gpurun_out/r5k/resident_300.json 2> gpurun_out/r5k/resident_phase_300.txt
python tools/resident_crossover.py > gpurun_out/r5k/resident_vs_host.jsonl 2> gpurun_out/r5k/resident_vs_host.err
python tools/robustness_sweep.py > gpurun_out/r5k/robust_default.txt 2>&1; tail -n 2 gpurun_out/r5k/robust_default.txt
python tools/robustness_sweep.py big > gpurun_out/r5k/robust_big.txt 2>&1; tail -n 2 gpurun_out/r5k/robust_big.txt
python tools/robustness_sweep.py wide > gpurun_out/r5k/robust_wide.txt 2>&1; tail -n 2 gpurun_out/r5k/robust_wide.txt
python tools/robustness_sweep.py wide2 > gpurun_out/r5k/robust_wide2.txt 2>&1; tail -n 2 gpurun_out/r5k/robust_wide2.txt
python tools/robustness_continuous.py > gpurun_out/r5k/robust_continuous.txt 2>&1; tail -n 2 gpurun_out/r5k/robust_continuous.txt
C4_TEND=14 KIN_PROGRESS=30 python tools/run_configs.py c4 > gpurun_out/r5k/c4_full_run.json 2> gpurun_out/r5k/c4_full_run.err; cat gpurun_out/r5k/c4_full_run.json | cut -c1-400

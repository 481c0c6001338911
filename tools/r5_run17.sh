# validation of the drift-guard threshold 1.0: full GPU suite, parity statistics under three reuse bands, robustness sweeps, the whole ramp
mkdir -p gpurun_out/r5p
python -m pytest tests -m gpu -q > gpurun_out/r5p/pytest_all.txt 2>&1; tail -n 6 gpurun_out/r5p/pytest_all.txt
LU_BAND=0.32,0.35,0.38 python tools/config_stats.py > gpurun_out/r5p/config_stats.jsonl 2> gpurun_out/r5p/config_stats.err; grep -c '"rc": 0' gpurun_out/r5p/config_stats.jsonl
python tools/robustness_sweep.py > gpurun_out/r5p/robust_default.txt 2>&1; tail -n 1 gpurun_out/r5p/robust_default.txt
python tools/robustness_sweep.py big > gpurun_out/r5p/robust_big.txt 2>&1; tail -n 1 gpurun_out/r5p/robust_big.txt
python tools/robustness_sweep.py wide > gpurun_out/r5p/robust_wide.txt 2>&1; tail -n 1 gpurun_out/r5p/robust_wide.txt
python tools/robustness_sweep.py wide2 > gpurun_out/r5p/robust_wide2.txt 2>&1; tail -n 1 gpurun_out/r5p/robust_wide2.txt
python tools/robustness_continuous.py > gpurun_out/r5p/robust_continuous.txt 2>&1; tail -n 1 gpurun_out/r5p/robust_continuous.txt
C4_TEND=14 KIN_PROGRESS=30 python tools/run_configs.py c4 > gpurun_out/r5p/c4_full_run.json 2> gpurun_out/r5p/c4_full_run.err; cut -c1-420 gpurun_out/r5p/c4_full_run.json

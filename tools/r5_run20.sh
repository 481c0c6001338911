mkdir -p gpurun_out/r5s
LU_BAND=0.32,0.35,0.38 python tools/config_stats.py c5 > gpurun_out/r5s/config_stats_c5.jsonl 2> gpurun_out/r5s/err.txt; cat gpurun_out/r5s/config_stats_c5.jsonl
python -m pytest tests/test_gpu_configs.py -m gpu -q -k c5 2>&1 | tail -n 3

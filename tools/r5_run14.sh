mkdir -p gpurun_out/r5m
KIN_LU_DEBUG=1 KIN_TIMING=1 python tools/setup_cost.py 10000 50000 > gpurun_out/r5m/setup_cost_c3.txt 2>&1; cat gpurun_out/r5m/setup_cost_c3.txt | tail -n 40

mkdir -p gpurun_out/r5a
python tools/solve_stats.py 10000 50000 100 > gpurun_out/r5a/solve_100.json 2> gpurun_out/r5a/solve_100.err
python tools/solve_stats.py 10000 50000 20 > gpurun_out/r5a/solve_20.json 2>> gpurun_out/r5a/solve_100.err
python tools/solve_stats.py 300 1500 20 > gpurun_out/r5a/solve_300.json 2>> gpurun_out/r5a/solve_100.err
python tools/c3_mid_units.py mid > gpurun_out/r5a/units_mid.txt 2>&1
python tools/c3_mid_units.py long > gpurun_out/r5a/units_long.txt 2>&1
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_configs.py > gpurun_out/r5a/pytest_rest.txt 2>&1; echo "rest rc=$?" >> gpurun_out/r5a/pytest_rest.txt
python -m pytest tests/test_gpu_configs.py -m gpu -q > gpurun_out/r5a/pytest_configs.txt 2>&1; echo "configs rc=$?" >> gpurun_out/r5a/pytest_configs.txt
cat gpurun_out/r5a/solve_*.json; tail -3 gpurun_out/r5a/pytest_rest.txt; tail -3 gpurun_out/r5a/pytest_configs.txt

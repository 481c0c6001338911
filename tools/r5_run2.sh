mkdir -p gpurun_out/r5b
LU_BAND=0.32,0.35,0.38 python tools/config_stats.py > gpurun_out/r5b/config_stats.jsonl 2> gpurun_out/r5b/config_stats.err
python -m pytest tests -m gpu -q > gpurun_out/r5b/pytest_all.txt 2>&1; echo "all rc=$?" >> gpurun_out/r5b/pytest_all.txt
python tools/tiled_bench.py c5 > gpurun_out/r5b/tiled_c5.json 2> gpurun_out/r5b/tiled_c5.err
python bench.py > gpurun_out/r5b/bench_default.json 2> gpurun_out/r5b/bench_default.err
tail -5 gpurun_out/r5b/pytest_all.txt; tail -c 1500 gpurun_out/r5b/bench_default.json

mkdir -p gpurun_out/r5q
bash tools/collect_profiles.sh r05 > gpurun_out/r5q/collect.log 2>&1; tail -n 3 gpurun_out/r5q/collect.log
python bench.py > gpurun_out/r5q/bench_default.json 2> gpurun_out/r5q/bench_default.err; tail -c 700 gpurun_out/r5q/bench_default.json

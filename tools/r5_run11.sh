mkdir -p gpurun_out/r5j
bash tools/collect_profiles.sh r05 > gpurun_out/r5j/collect.log 2>&1; tail -n 5 gpurun_out/r5j/collect.log; ls gpurun_out/prof_r05/summary
python bench.py > gpurun_out/r5j/bench_default.json 2> gpurun_out/r5j/bench_default.err; tail -c 600 gpurun_out/r5j/bench_default.json

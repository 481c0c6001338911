mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_tiled.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r5d/pytest_tiled.txt 2>&1; echo "rc=$?" >> gpurun_out/r5d/pytest_tiled.txt
tail -3 gpurun_out/r5d/pytest_tiled.txt
python tools/dropin_bench.py c5 > gpurun_out/r5d/dropin_c5.json 2> gpurun_out/r5d/dropin_c5.err; cat gpurun_out/r5d/dropin_c5.json
python tools/dropin_bench.py cut > gpurun_out/r5d/dropin_cut.json 2> gpurun_out/r5d/dropin_cut.err; cat gpurun_out/r5d/dropin_cut.json
python tools/tiled_bench.py c5 > gpurun_out/r5d/tiled_c5.json 2> gpurun_out/r5d/tiled_c5.err
bash tools/band_ab.sh > gpurun_out/r5d/band_ab.txt 2>&1
cat gpurun_out/r5d/band_ab.txt

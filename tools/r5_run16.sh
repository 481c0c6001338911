mkdir -p gpurun_out/r5o
python -m pytest tests/test_gpu_resident.py tests/test_gpu_ensemble.py -m gpu -q > gpurun_out/r5o/pytest_res.txt 2>&1; tail -n 4 gpurun_out/r5o/pytest_res.txt
python tools/ensemble_resident_scaling.py > gpurun_out/r5o/ensemble_resident_scaling.jsonl 2> gpurun_out/r5o/ens.err; cat gpurun_out/r5o/ensemble_resident_scaling.jsonl

mkdir -p gpurun_out/r5l
python -m pytest tests -m gpu -q > gpurun_out/r5l/pytest_all.txt 2>&1; tail -n 8 gpurun_out/r5l/pytest_all.txt

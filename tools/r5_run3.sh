mkdir -p gpurun_out/r5c
python -m pytest tests -m gpu -q > gpurun_out/r5c/pytest_all.txt 2>&1; echo "all rc=$?" >> gpurun_out/r5c/pytest_all.txt
python tools/tight_tol_truth.py > gpurun_out/r5c/tight_tol_truth.jsonl 2> gpurun_out/r5c/tight_tol_truth.err
python tools/solve_stats.py 10000 50000 100 > gpurun_out/r5c/solve_100.json 2>&1
tail -5 gpurun_out/r5c/pytest_all.txt; cat gpurun_out/r5c/tight_tol_truth.jsonl gpurun_out/r5c/solve_100.json

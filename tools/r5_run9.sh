mkdir -p gpurun_out/r5h
python tools/tight_tol_truth.py > gpurun_out/r5h/tight_tol_truth.jsonl 2> gpurun_out/r5h/tight_tol_truth.err; tail -n 1 gpurun_out/r5h/tight_tol_truth.jsonl
bash tools/switch_matrix.sh > gpurun_out/r5h/switch_matrix.txt 2>&1; cat gpurun_out/r5h/switch_matrix.txt

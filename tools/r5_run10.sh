mkdir -p gpurun_out/r5i
T="tests/test_gpu_solve.py tests/test_gpu_resident.py tests/test_gpu_boundary_r2.py tests/test_gpu_ensemble.py"
for kv in "KIN_RESIDENT=0" "KIN_RESIDENT_MAX_N=100" "KIN_ENSEMBLE_BATCHED=1" "KIN_ENSEMBLE_THREADS=2" "KIN_LU_FUSED=0" "KIN_LU_CACHE_SLOTS=1" "KIN_LU_CACHE_SLOTS=8" "KIN_LU_BAND=0.2" "KIN_LU_CACHE_MB=64"; do
  echo "== $kv"; env "$kv" python -m pytest $T -m gpu -q --tb=line 2>&1 | grep -E "^/root|^E |Error|assert" | cut -c1-400
done > gpurun_out/r5i/switch_reasons.txt 2>&1
cat gpurun_out/r5i/switch_reasons.txt

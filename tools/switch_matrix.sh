#!/bin/bash
# The solve subset of the GPU suite once under every surviving non-default switch value that changes how a solve runs
# (INTEGRATION.md has the list). One line per value: pytest's summary, then the ids of the tests that failed - tests that assert
# a property of the DEFAULT route (a slot count, a bit-identity between two routes) are expected there and are named.
# Usage on the GPU box: bash tools/switch_matrix.sh > gpurun_out/switch_matrix.txt
TESTS="tests/test_gpu_solve.py tests/test_gpu_resident.py tests/test_gpu_boundary_r2.py tests/test_gpu_ensemble.py"
for kv in "KIN_RESIDENT=0" "KIN_RESIDENT_MAX_N=100" "KIN_RESIDENT_SHARED_CU=0" "KIN_RESIDENT_SHARED_CU=1" "KIN_ENSEMBLE_BATCHED=1" "KIN_ENSEMBLE_ROUTE=threads" \
          "KIN_ENSEMBLE_GJ_BATCHED=0" "KIN_ENSEMBLE_STREAMS=2" "KIN_ENSEMBLE_THREADS=2" "KIN_ENSEMBLE_MAX_MEMBERS=8" "KIN_REPLICA_PLACEHOLDER_STREAMS=0" \
          "KIN_LU_FUSED=0" "KIN_LU_EXPLICIT=0" "KIN_FUSE_NEWTON=0" "KIN_FUSE_NEWTON=1" "KIN_SPECULATE=0" "KIN_NO_FAST_SYNC=1" \
          "KIN_LU_CACHE_SLOTS=1" "KIN_LU_CACHE_SLOTS=8" "KIN_LU_BAND=0.2" "KIN_LU_CACHE_MB=64"; do
  out=$(env "$kv" python -m pytest $TESTS -m gpu -q 2>&1)
  echo "== $kv: $(echo "$out" | tail -n 1)"
  echo "$out" | grep -E "^FAILED" | sed 's/ - .*//' | sed 's/^/     /'
done

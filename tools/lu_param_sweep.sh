#!/bin/bash
# dense-block dimension / factorisation cost against the elimination limits (env knobs of Solver): C3 20 chunks
for hub in 32 48 64 96; do for rounds in 16 24; do for tail in 32 64; do
  echo "hub=$hub rounds=$rounds tail=$tail $(KIN_LU_HUB_DEGREE=$hub KIN_LU_MAX_ROUNDS=$rounds KIN_LU_MAX_TAIL_DEGREE=$tail KIN_LU_MAX_DEGREE=600 python tools/solve_stats.py 10000 50000 20 2>&1 | grep -v amdgpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['wall_s'], d['lu_dense_dim'], d['n_steps'], d['n_factor'])")"
done; done; done

for mode in "" "KIN_LU_FUSED=0" "KIN_LU_EXPLICIT=0"; do
  for n in 300 1000; do
    echo "== $mode N=$n"
    env $mode KIN_RESIDENT_MAX_N=2000 SOLVE_REPEATS=2 timeout -k 5 120 python3 tools/solve_stats.py $n $((5*n)) 20 2>&1 | tail -1 | cut -c1-200
  done
done

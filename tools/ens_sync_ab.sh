# lockstep ensemble at C3 size: factorisations inside the rounds (members stay in step) against parked members (ensemble.cpp)
run() { echo "== $*"; env "$@" ENS_REPEATS=3 timeout -k 5 250 python3 tools/ensemble_batched_check.py 10000 16 32 2>&1 | grep -E "solves_per_s|ensemble\]" | cut -c1-260; }
run KIN_ENSEMBLE_FACTOR_SYNC=1 KIN_TIMING=1
run KIN_ENSEMBLE_FACTOR_SYNC=0
run KIN_ENSEMBLE_FACTOR_SYNC=1

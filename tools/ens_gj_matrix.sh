# lockstep ensemble at C3 size: batched dense inverses with different hold-off settings (ensemble.cpp: gj_server)
run() { echo "== $*"; env "$@" ENS_REPEATS=4 ENS_NO_SOLO=1 timeout -k 5 200 python3 tools/ensemble_batched_check.py 10000 16 32 2>&1 | grep -E "solves_per_s" | cut -c1-150; }
run KIN_ENSEMBLE_GJ_BATCHED=0
run KIN_ENSEMBLE_GJ_MIN=1
run KIN_ENSEMBLE_GJ_MIN=4 KIN_ENSEMBLE_GJ_WAIT_US=200
run KIN_ENSEMBLE_GJ_MIN=8 KIN_ENSEMBLE_GJ_WAIT_US=500
run KIN_ENSEMBLE_GJ_BATCHED=0
run KIN_ENSEMBLE_GJ_MIN=4 KIN_ENSEMBLE_GJ_WAIT_US=200

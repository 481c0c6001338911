#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command and
# separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950), then writes the
# summaries that get committed under profiles/. Usage: tools/collect_profiles.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --solve-chunks 20 --cpu-solve-chunks 0 --no-pmc --no-tiled --no-configs --no-crossover --replicas "" --sustain-seconds 0.5 > "$OUT/bench_stats.log" 2>&1
# the same command without the spin-up and the sustained leg: the profiler's average then covers the warm-up + timed launches only (cold clocks)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_timed" -- python3 bench.py --solve-chunks 0 --no-cpu --no-pmc --no-tiled --no-configs --no-crossover --sustain-seconds 0 --spinup-seconds 0 > "$OUT/bench_stats_timed.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 5 --solve-chunks 0 --no-cpu --no-pmc --no-tiled --no-configs --no-crossover --sustain-seconds 0 --spinup-seconds 0 > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 5 --solve-chunks 0 --no-cpu --no-pmc --no-tiled --no-configs --no-crossover --sustain-seconds 0 --spinup-seconds 0 > "$OUT/bench_write.log" 2>&1
# the other two bandwidth kernels (C5 large-N sweep, C4-size rate table): same three passes
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg_stats" -- python3 tools/run_configs.py c5sweep table > "$OUT/cfg_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cfg_fetch" -- python3 tools/run_configs.py c5sweep table > "$OUT/cfg_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cfg_write" -- python3 tools/run_configs.py c5sweep table > "$OUT/cfg_write.log" 2>&1
# the implicit solve alone (C3, 20 chunks): per-kernel time of the BDF step chain
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/solve_stats" -- python3 tools/solve_stats.py 10000 50000 20 > "$OUT/solve_stats.log" 2>&1
# round 4: the resident integrator (one workgroup owns the trajectory; 300 species, 20 chunks), and a lockstep ensemble of the
# 10k-species network (16 members, first 2 chunks)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/resident_stats" -- python3 tools/solve_stats.py 300 1500 20 > "$OUT/resident_stats.log" 2>&1
# (the profiler itself has crashed once in this leg - a SIGSEGV inside its dispatch hook under the launches of ~20 host threads,
# gpurun_out/prof_r04/ensemble_stats.log of that run; the same command passes without the profiler and passed under it twice:
# the leg must not take the others' summaries with it)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ensemble_stats" -- python3 tools/ensemble_batched_check.py 10000 16 > "$OUT/ensemble_stats.log" 2>&1 || echo "ensemble leg failed under the profiler"
python3 tools/summarize_profiles.py "$OUT" "$TAG"
# keep the merge-back small
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete

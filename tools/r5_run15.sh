mkdir -p gpurun_out/r5n
python tools/robustness_lockstep.py > gpurun_out/r5n/robustness_lockstep.jsonl 2> gpurun_out/r5n/robustness_lockstep.err; tail -n 2 gpurun_out/r5n/robustness_lockstep.jsonl | cut -c1-300
python tools/robustness_resident.py > gpurun_out/r5n/robustness_resident.jsonl 2> gpurun_out/r5n/robustness_resident.err; tail -n 2 gpurun_out/r5n/robustness_resident.jsonl | cut -c1-400
python __graft_entry__.py smoke 2>&1 | tail -n 2

set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_c5; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/run_configs.py c5sweep > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/run_configs.py c5sweep > $OUT/w.log 2>&1
python3 - <<'P'
import csv, glob
def avg(path, name):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "sweep_big" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)
fs, n = avg("gpurun_out/pmc_c5/fetch", "FETCH_SIZE"); ws, _ = avg("gpurun_out/pmc_c5/write", "WRITE_SIZE")
tot = (2 * fs + ws) * 1024
print("launches", n, "FETCH KiB", fs, "WRITE KiB", ws, "bytes", tot, "x algorithmic", tot / 2.8722e9)
P

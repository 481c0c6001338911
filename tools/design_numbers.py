"""Writes the "Numbers" block of DESIGN.md (between the numbers:begin / numbers:end markers) from the files under profiles/.
Every row names the file (and key) it comes from; tests/test_docs_numbers.py runs this script in check mode and fails when the
block in DESIGN.md differs from what the files give.
A file is taken from the current round (`r05_<name>`) when it exists, else from the last round that measured it (`r04_<name>`:
kernels and probes this round did not touch); the source column names the file that was read.
Usage: python tools/design_numbers.py [--check] [--round r05]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "r05"
PREV = "r04"
if "--round" in sys.argv:
    TAG = sys.argv[sys.argv.index("--round") + 1]
P = os.path.join(ROOT, "profiles")


def pick(name):
    """file name under profiles/ for `name`: this round's if it exists, else the previous round's"""
    for tag in (TAG, PREV):
        if os.path.exists(os.path.join(P, f"{tag}_{name}")):
            return f"{tag}_{name}"
    raise FileNotFoundError(name)


def load(name):
    return json.load(open(os.path.join(P, name)))


def jsonl(name):
    return [json.loads(l) for l in open(os.path.join(P, name)) if l.startswith("{")]


def kernel_row(csv_name, needle):
    rows = list(csv.DictReader(open(os.path.join(P, csv_name))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    r = [r for r in rows if needle in r["Name"]][0]
    return int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / tot


def build():
    rows = []   # (group, what, value, source)

    def add(group, what, value, source):
        rows.append((group, what, value, source))

    b = load(pick("bench_default.json"))
    src_b = pick("bench_default.json")
    rf = b["roofline"]
    add("headline sweep", "`sweep_reg_kernel`, B = 4096, 10k / 50k: launch (bench HIP events, after 1 s spin-up)",
        f"{rf['avg_launch_ms']:.4f} ms = {rf['achieved']:.0f} GB/s = **{rf['frac']:.3f} of 8 TB/s**", f"{src_b} `roofline`")
    add("headline sweep", "RHS evaluations per second (`value`)", f"{b['value'] / 1e6:.2f} M/s", f"{src_b} `value`")
    add("headline sweep", "HBM traffic (PMC, same run) / algorithmic bytes", f"{rf['traffic_over_algorithmic']:.3f}", f"{src_b} `roofline.traffic_over_algorithmic`")
    pm = load(pick("sweep_pmc.json"))
    add("headline sweep", f"rocprofv3 average over {pm['calls']} launches (spin-up + timed + sustained)",
        f"{pm['avg_ns_rocprof'] / 1e3:.1f} µs (bench events in that run {pm['avg_launch_ms_bench_events'] * 1e3:.1f} µs)", pick("sweep_pmc.json"))
    tl = pm["timed_launches_only"]
    add("headline sweep", f"cold clocks: rocprofv3 average over the {tl['calls']} warm-up + timed launches, no spin-up",
        f"{tl['avg_ns_rocprof'] / 1e3:.1f} µs = {pm['algorithmic_bytes_per_launch'] / tl['avg_ns_rocprof'] / 8000:.3f}", pick("sweep_pmc.json") + " `timed_launches_only`")
    add("headline sweep", "PMC: (2 × FETCH_SIZE + WRITE_SIZE) KiB / algorithmic", f"{pm['traffic_over_algorithmic']:.3f}", pick("sweep_pmc.json"))
    add("headline sweep", "CPU port, oracle RHS, 1 core", f"{b['cpu_baseline']['value']:.0f} evals/s", f"{src_b} `cpu_baseline`")

    ts = b["tiled_sweep"]
    for cfg in ("C3", "C5"):
        k, t = ts[cfg]["k_stream"], ts[cfg]["temperature_form"]
        add("tiled sweep", f"{cfg} ({ts[cfg]['species']} / {ts[cfg]['reactions']}, B = {ts[cfg]['states']}), k-stream form",
            f"{k['ms']:.3f} ms = {k['frac_of_8TBps']:.3f} of 8 TB/s", f"{src_b} `tiled_sweep.{cfg}.k_stream`")
        add("tiled sweep", f"{cfg}, temperature form (no k)", f"{t['ms']:.3f} ms = {t['evals_per_s'] / 1e6:.2f} M evals/s", f"{src_b} `tiled_sweep.{cfg}.temperature_form`")
    pc = b.get("post_cutoff_sweep")
    if pc and "library_order" in pc:
        add("post-cutoff sweep", f"{pc['network']} ({pc['reactions']} reactions, {pc['records']} records, k row {pc['k_len']} doubles): caller's layouts / library order",
            f"{pc['callers_layouts']['ms']:.3f} ms = {pc['callers_layouts']['frac_of_8TBps']:.3f} / {pc['library_order']['ms']:.3f} ms = {pc['library_order']['frac_of_8TBps']:.3f} of 8 TB/s",
            f"{src_b} `post_cutoff_sweep`")
    for cfg in ("c3", "c5"):
        dv = load(pick(f"{cfg}_tiled_pmc.json"))["derived"]
        bc, ac = dv["lds_bank_conflict_cycles"], dv["lds_active_cycles"]
        add("tiled sweep", f"{cfg.upper()} k-stream under rocprofv3: PMC traffic / algorithmic; LDS bank-conflict cycles per launch, plain order → bank-aware inside chunks of 64 records (default) → from a reservoir of 256; LDS-active cycles",
            f"{dv['k_stream_traffic_over_algorithmic']:.3f}; {bc[0] / 1e6:.0f} M → {bc[1] / 1e6:.0f} M → {bc[2] / 1e6:.0f} M; "
            f"{ac[0] / 1e6:.0f} M → {ac[1] / 1e6:.0f} M → {ac[2] / 1e6:.0f} M", pick(f"{cfg}_tiled_pmc.json") + " `derived`")

    ct = load(pick("c5_table_pmc.json"))
    for kr in ct["kernels"]:
        name = kr["kernel"].split("::")[-1]
        add("rate table" if "rate_table" in name else "caller's-layout sweep at C5", f"`{name}`: rocprofv3 average over {kr['calls']} launches",
            f"{kr['avg_ns_rocprof'] / 1e6:.3f} ms = {kr['achieved_GBps']:.0f} GB/s = {kr['frac_of_8TBps']:.3f}, traffic {kr['traffic_over_algorithmic']:.2f}× algorithmic",
            pick("c5_table_pmc.json"))

    sn = b["solve_network"]
    add("C3 solve", "100 chunks (0, 0.1) s: warm / cold (handle creation + analysis + first solve)",
        f"{sn['gpu_wall_s']:.3f} s / {sn['cold_wall_s']:.3f} s ({sn['cold_wall_s'] / sn['gpu_wall_s']:.2f}×)", f"{src_b} `solve_network`")
    st = sn["stats"]
    add("C3 solve", "... steps / factorisations / cache hits / dense block", f"{st['n_steps']} / {st['n_factor']} / {st['n_lu_reused']} / {st['lu_dense_dim']}", f"{src_b} `solve_network.stats`")
    add("C3 solve", f"first {sn['same_chunks']} chunks: GPU / CPU port on 1 core", f"{sn['gpu_wall_same_chunks_s']:.3f} s / {sn['cpu_wall_same_chunks_s']:.2f} s = {sn['speedup_same_chunks_1core']:.1f}×",
        f"{src_b} `solve_network.speedup_same_chunks_1core`")
    ca = sn["cpu_all_cores"]
    add("C3 solve", f"one GPU against {ca['cores']} host cores (one replica each): single solve / best K threads", f"{ca['gpu_over_all_cores']:.2f}× / {ca['gpu_over_all_cores_best_K']:.1f}×", f"{src_b} `solve_network.cpu_all_cores`")
    add("C3 solve", "device vs CPU port on those chunks, tolerance units max / rms", f"{sn['max_dev_vs_cpu_in_tol_units']:.1f} / {sn['rms_dev_vs_cpu_in_tol_units']:.2f}", src_b)

    cf = sn["configs"]
    c3 = cf["C3_whole_span"]
    cw, cc = c3["chunkwise_1000_chunks"], c3["complete_timespan"]
    add("configurations", "C3 over (0, 1) s chunkwise (1 000 chunks): wall, steps, factorisations", f"{cw['wall_s']:.2f} s, {cw['stats']['n_steps']}, {cw['stats']['n_factor']}", f"{src_b} `configs.C3_whole_span`")
    add("configurations", "C3 over (0, 1) s as ONE integration: wall, steps, corrector failures, factorisations",
        f"{cc['wall_s']:.3f} s, {cc['stats']['n_steps']}, {cc['stats']['n_newton_fail']}, {cc['stats']['n_factor']}", f"{src_b} `configs.C3_whole_span`")
    cwm = c3.get("chunkwise_1000_chunks_warm_extension")
    if cwm:
        add("configurations", "C3 over (0, 1) s, 1 000 chunks with warm continuation (extension `solve_chunks = 2`): wall, steps, factorisations; final state vs the single integration",
            f"{cwm['wall_s']:.2f} s, {cwm['stats']['n_steps']}, {cwm['stats']['n_factor']}; {cwm['final_state_vs_complete_timespan_units']['max']:.0f} / {cwm['final_state_vs_complete_timespan_units']['rms']:.1f} units",
            f"{src_b} `configs.C3_whole_span`")
    fa = c3["final_states_apart_in_tolerance_units"]
    add("configurations", "... final states of the two apart, tolerance units max / rms", f"{fa['max']:.0f} / {fa['rms']:.1f}", src_b)
    c30 = cf.get("C3_30_chunks_vs_truth")
    if c30 and "chunkwise" in c30:
        for name in ("chunkwise", "complete_timespan", "chunkwise_warm_extension"):
            if name not in c30:
                continue
            v30 = c30[name]["vs_truth_in_tolerance_units"]
            add("configurations", f"C3 over (0, 0.03) s {name.replace('_', ' ')} against `truth_c3_mid.npz`: wall, steps; max / rms units",
                f"{c30[name]['wall_s']:.3f} s, {c30[name]['stats']['n_steps']}; {v30['max']:.0f} / {v30['rms']:.1f}", f"{src_b} `configs.C3_30_chunks_vs_truth`")
    c100 = cf.get("C3_100_chunks_vs_truth")
    if c100 and "chunkwise" in c100:
        for name in ("chunkwise", "chunkwise_warm_extension"):
            v100 = c100[name]["vs_truth_in_tolerance_units"]
            add("configurations", f"C3 over (0, 0.1) s, 100 chunks, {name.replace('_', ' ')} against `truth_c3_long.npz`: wall, steps; max / rms units",
                f"{c100[name]['wall_s']:.3f} s, {c100[name]['stats']['n_steps']}; {v100['max']:.0f} / {v100['rms']:.1f}", f"{src_b} `configs.C3_100_chunks_vs_truth`")
    c4 = cf["C4_prefix"]
    v = c4["vs_truth_in_tolerance_units"]
    add("C4 prefix", "first 20 chunks (200 restarts): wall, steps, factorisations", f"{c4['wall_s']:.2f} s, {c4['stats']['n_steps']}, {c4['stats']['n_factor']}", f"{src_b} `configs.C4_prefix`")
    add("C4 prefix", "... against `truth_c4_long.npz`: max / rms / p99.9 units (truth self-check)", f"{v['max']:.0f} / {v['rms']:.1f} / {v['p99.9']:.1f} ({v['truth_self_check']:.1f})", src_b)
    pth = os.path.join(P, pick("c4_full_run.json"))
    if True:
        c4f = json.load(open(pth))
        add("C4 prefix", "the whole 14 s ramp (tools/run_configs.py c4, C4_TEND=14): wall, steps, factorisations, restarts, retries",
            f"{c4f['wall_s']:.1f} s, {c4f['stats']['n_steps']}, {c4f['stats']['n_factor']}, {c4f['stats']['n_restarts']}, {c4f['stats']['n_retries']} (retcode {c4f['retcode']})", pick("c4_full_run.json"))
    c5 = cf["C5_static"]
    add("configurations", "C5 static, 5 chunks: warm / cold, dense block", f"{c5['wall_s']:.2f} s / {c5['cold_wall_s_incl_create_and_analysis']:.2f} s, {c5['dense_block']}", f"{src_b} `configs.C5_static`")

    for r in sn["crossover"]["rows"]:
        cpu = r.get("cpu_port_1core_s")
        extra = f", host-driven {r['gpu_warm_host_driven_s']:.3f} s" if "gpu_warm_host_driven_s" in r else ""
        cpus = f"{cpu:.3f} s" + (f" for {r['cpu_chunks']} chunks" if r.get("cpu_chunks", 20) != 20 else "") if cpu is not None else "-"
        add("crossover", f"{r['species']} species, 20 chunks: GPU warm / cold{', host-driven' if extra else ''}; CPU port 1 core",
            f"{r['gpu_warm_s']:.3f} s / {r['gpu_cold_s']:.3f} s{extra}; {cpus} (default integrator: {r['integrator'].split(' (')[0]})", f"{src_b} `crossover.rows`")

    en = sn["ensemble_one_launch"]
    for key, e in en.items():
        if not isinstance(e, dict) or "solves_per_s" not in e:
            continue
        group = "ensemble, lockstep" if "lockstep" in key else ("ensemble, threads" if "threads" in key else "ensemble, one launch")
        add(group, key.replace("_", " "), f"{e['solves_per_s']:.0f} solves/s ({e['wall_s']:.3f} s, {e['members_ok']} of {e['members']} ok" + (f", steps mean {e['steps_per_member']:.0f} / slowest {e['steps_of_the_slowest_member']}" if 'steps_of_the_slowest_member' in e else "") + ")", f"{src_b} `ensemble_one_launch.{key}`")
    for k_, e in sorted(sn["concurrent_replicas"].items()):
        if isinstance(e, dict) and "solves_per_s" in e:
            add("ensemble, lockstep", f"for comparison, round 3's K handles on K host threads, K = {k_} (2 chunks of C3)", f"{e['solves_per_s']:.1f} solves/s", f"{src_b} `concurrent_replicas`")

    path = os.path.join(P, pick("ensemble_resident_scaling.jsonl"))
    if os.path.exists(path):
        for r in jsonl(pick("ensemble_resident_scaling.jsonl")):
            if r["species"] == 1000 or r["K"] == 1024:
                add("ensemble, one launch", f"scaling probe: {r['species']} species, K = {r['K']}, members at {r['T']}",
                    f"{r['solves_per_s']:.0f} solves/s (steps mean {r['steps_mean']:.0f} / slowest {r['steps_max']}, {r['slots']} cache slots per member)",
                    pick("ensemble_resident_scaling.jsonl"))
    path = os.path.join(P, pick("resident_vs_host.jsonl"))
    if os.path.exists(path):
        for r in jsonl(pick("resident_vs_host.jsonl")):
            add("resident vs host-driven", f"{r['species']} species (dense block {r['dense_block']}), 20 chunks: resident / host-driven",
                f"{r['resident_s']:.3f} s / {r['host_driven_s']:.3f} s = {r['host_over_resident']:.2f}×", pick("resident_vs_host.jsonl"))

    calls, avg, pct = kernel_row(pick("solve_kernel_stats.csv"), "gj_update_kernel")
    add("solve kernel time", "C3 20 chunks: `gj_update_kernel` share of kernel time, average launch", f"{pct:.1f} %, {avg:.2f} µs × {calls}", pick("solve_kernel_stats.csv"))
    for needle, label in (("bdf_newton_kernel", "corrector update + decision"), ("segsum_kernel<4, 1024>", "solve stage A gather"), ("segsum_kernel<2, 1024>", "Newton residual gather"),
                          ("segsum_kernel<5, 256>", "solve stage C gather"), ("gemv_kernel", "dense GEMV"), ("rates_skip_kernel", "rates")):
        try:
            calls, avg, pct = kernel_row(pick("solve_kernel_stats.csv"), needle)
            add("solve kernel time", f"... {label} (`{needle}`)", f"{pct:.1f} %, {avg:.2f} µs", pick("solve_kernel_stats.csv"))
        except IndexError:
            pass
    gj = open(os.path.join(P, pick("gj_probe_batched.txt"))).read()
    m = re.search(r"m=(\d+)\s+GJ mean ([\d.]+) us\s+best ([\d.]+) us\s+\((\d+) launches\)", gj)
    add("Gauss-Jordan", f"dense inverse alone, m = {m.group(1)} ({m.group(4)} launches), HIP events", f"{m.group(2)} µs mean, {m.group(3)} µs best", pick("gj_probe_batched.txt"))
    for mm in re.finditer(r"batched n=\s*(\d+): mean ([\d.]+) us\s+best ([\d.]+) us\s+= ([\d.]+) us per matrix, results (.*)", gj):
        if mm.group(1) in ("4", "8", "16"):
            add("Gauss-Jordan", f"batched chain, {mm.group(1)} matrices per launch", f"{mm.group(4)} µs per matrix ({mm.group(5).strip()})", pick("gj_probe_batched.txt"))
    db = load(pick("dense_block_options.json"))["rows"]
    ms = [r["dense_block"] for r in db if "dense_block" in r]
    add("Gauss-Jordan", f"dense block at C3 over {len(ms)} elimination option sets (configured: {db[0]['dense_block']})", f"{min(ms)} ... {max(ms)}", pick("dense_block_options.json"))

    ph = open(os.path.join(P, pick("resident_phase_300.txt"))).read()
    head = re.search(r"N=(\d+) m=(\d+) slots=\d+ steps=(\d+) factor=(\d+) linsolve=(\d+) wall ([\d.]+) s", ph)

    def phase(label):
        return float(re.search(re.escape(label) + r"\s+([\d.]+) ms", ph).group(1))
    add("resident phases", f"{head.group(1)} species, 20 chunks: {head.group(3)} steps, {head.group(4)} factorisations, {head.group(5)} solves",
        f"kernel {phase('kernel'):.1f} ms = {1e3 * phase('kernel') / int(head.group(3)):.0f} µs per step", pick("resident_phase_300.txt"))
    add("resident phases", "... factorisations (of which dense inverse) / corrector attempts (of which solves)",
        f"{phase('factor'):.1f} ({phase('(of which dense inverse)'):.1f}) / {phase('corrector attempts'):.1f} ({phase('(solve)'):.1f}) ms", pick("resident_phase_300.txt"))
    wl = open(os.path.join(P, pick("wg_latency_probe.txt"))).read().split("grid = 256")[0]
    lat = {}
    for l in wl.splitlines():
        mm = re.match(r"\s+(.*?)\s{2,}([\d.]+) ns", l)
        if mm:
            lat[mm.group(1)] = float(mm.group(2))
    add("resident phases", "in-workgroup latencies: barrier / dependent global load / LDS read / 5 wave sums DPP (bpermute) / hand-over LDS (global)",
        f"{lat['barrier']:.0f} / {lat['dependent global load (idx chase)']:.0f} / {lat['dependent LDS read']:.0f} / {lat['5 wave sums (DPP)']:.0f} ({lat['5 wave sums (ds_bpermute)']:.0f}) / "
        f"{lat['stage hand-over through LDS']:.0f} ({lat['stage hand-over through global memory (store, barrier, load, barrier)']:.0f}) ns", pick("wg_latency_probe.txt"))

    # ---- round 5
    dc5, dcut = load(pick("c5_dropin_pmc.json")), load(pick("cut_dropin_pmc.json"))
    for d, label, src in ((dc5, "C5 (50000 / 250000, B = 1024)", pick("c5_dropin_pmc.json")), (dcut, "post-cutoff C3 network (35000 reactions, B = 4096)", pick("cut_dropin_pmc.json"))):
        t = d["timing_hip_events"]
        add("drop-in sweep", f"`kin_rhs_batched_klib_dev` (caller's species order, k in slot order), {label}: call (HIP events); HBM traffic (PMC) / algorithmic",
            f"{t['ms_per_call']:.3f} ms = {t['frac_of_8TBps']:.3f} of 8 TB/s; {d['traffic_over_algorithmic']:.2f}", src)
    ks = dc5["kernel_stats"]
    perm = [v["avg_ns"] / 1e3 for k_, v in ks.items() if "permute_staged_kernel" in k_]
    tile = [v["avg_ns"] / 1e3 for k_, v in ks.items() if "tiled_sweep_kernel" in k_]
    add("drop-in sweep", "... its kernels at C5 under rocprofv3: species permutation in / tiled sweep / permutation out", f"{max(perm):.0f} / {tile[0]:.0f} / {min(perm):.0f} µs", pick("c5_dropin_pmc.json") + " `kernel_stats`")
    cs = jsonl(pick("config_stats.jsonl"))
    for case, label in (("c3_long_chunkwise", "C3 100 chunks as the reference runs them"), ("c3_long_complete", "... as ONE integration"), ("c3_long_warm", "... warm chunk starts (extension)"),
                        ("c3_long_x0.1", "... 10× tighter tolerances (default units)"), ("c4_long_default", "C4 ramp, 20 chunks"), ("c5_default", "C5 ramp, 2 chunks")):
        rr = [r for r in cs if r["case"] == case]
        f3 = lambda key: " / ".join(f"{r[key]:g}" for r in rr)
        add("parity statistics", f"{label}: rms; p99.9; 50 major species; max over all (reuse band 0.32 / 0.35 / 0.38)", f"{f3('rms')}; {f3('p999')}; {f3('major')}; {f3('max')}", pick("config_stats.jsonl"))
    tt = jsonl(pick("tight_tol_truth.jsonl"))[-1]
    add("tight tolerances", "200 species at rtol 1e-10 against a Radau truth, tight units rms / max (steps, corrector failures): CPU port; resident; host-driven",
        "; ".join(f"{tt[k_]['rms_units']:.0f} / {tt[k_]['max_units']:.0f} ({tt[k_]['steps']}, {tt[k_]['fail']})" for k_ in ("cpu_port_tight", "resident_tight", "host_driven_tight")), pick("tight_tol_truth.jsonl"))
    g2 = open(os.path.join(P, pick("gj_two_step_probe.txt"))).read()
    m1 = re.search(r"one-step chain: mean ([\d.]+) us.*\n.*two-step chain: mean ([\d.]+) us", g2)
    add("Gauss-Jordan", "two block steps per launch (round 5, not adopted; bit-identical inverse), m = 1024: one-step / two-step chain", f"{m1.group(1)} / {m1.group(2)} µs", pick("gj_two_step_probe.txt"))
    sm = open(os.path.join(P, pick("switch_matrix.txt"))).read()
    runs = re.findall(r"^== (\S+): (.*)$", sm, flags=re.M)
    clean = sum(1 for _, r_ in runs if "failed" not in r_)
    add("switches", f"solve tests (48) under each of {len(runs)} non-default switch values: values with every test green / with tests that assert default-route properties failing", f"{clean} / {len(runs) - clean}", pick("switch_matrix.txt"))

    rk = b.get("ranks", {})
    add("multi-GPU", "ranks that took part in the committed bench line", f"world size {rk.get('world_size')}, backend {rk.get('backend')}; no scaling curve measured", f"{src_b} `ranks`")

    out = ["| | what | value | source (`profiles/`) |", "|---|---|---|---|"]
    last = None
    for g, w, v_, s in rows:
        out.append(f"| {g if g != last else ''} | {w} | {v_} | {s} |")
        last = g
    return "\n".join(out)


def main():
    block = build()
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    a, b = text.index("<!-- numbers:begin -->") + len("<!-- numbers:begin -->"), text.index("<!-- numbers:end -->")
    new = text[:a] + "\n" + block + "\n" + text[b:]
    if "--check" in sys.argv:
        if new != text:
            print("DESIGN.md's numbers block is out of date: run python tools/design_numbers.py")
            sys.exit(1)
        print("DESIGN.md numbers block matches profiles/")
        return
    open(path, "w").write(new)
    print(block)


if __name__ == "__main__":
    main()

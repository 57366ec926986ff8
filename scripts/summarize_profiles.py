"""Copy the rocprofv3 summaries of scripts/profile_bench.sh from gpurun_out/prof into profiles/ (tracked).

    python scripts/summarize_profiles.py <tag>          # e.g. r2a
"""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "prof")
P = os.path.join(ROOT, "profiles")


def one(pattern):
    f = glob.glob(os.path.join(O, pattern), recursive=True)
    return f[0] if f else None


for src, dst in (("stats/**/*_kernel_stats.csv", "bench_kernel_stats.csv"), ("stats_chol/**/*_kernel_stats.csv", "d20_cholesky_kernel_stats.csv"),
                 ("stats_ot/**/*_kernel_stats.csv", "ot_kernel_stats.csv"), ("stats_p1/**/*_kernel_stats.csv", "phase1_kernel_stats.csv")):
    f = one(src)
    if f:
        shutil.copy(f, os.path.join(P, tag + "_" + dst))
for name in ("bench_line.json", "bench_detail.json", "launch_profile_d500_cgd15.txt", "launch_profile_d100_cgd15.txt", "launch_profile_d20_cholesky.txt",
             "launch_profile_d500_cgd20_w32.txt", "probe.txt", "ot_probe.txt", "phase1_probe.txt", "phase1_baseline.jsonl",
             "big_factorisations.txt", "startup_timeline.json", "startup_timeline.txt", "valu_issue.txt", "split_trace.txt", "hip_init.txt", "hip_exit.txt"):
    if os.path.exists(os.path.join(O, name)):
        shutil.copy(os.path.join(O, name), os.path.join(P, tag + "_" + name))


def counters(dirs):
    """per kernel: launches and the average per launch of every counter in the given --pmc passes"""
    out = {}
    for d in dirs:
        f = one(d + "/**/*_counter_collection.csv")
        if not f:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                out.setdefault(k, {})[c] = sum(v) / len(v)
                out[k]["launches"] = len(v)
    return out


hbm = counters(["pmc_fetch", "pmc_write"])
for k, v in hbm.items():
    # MI355X_MICROARCH.md (HBM): gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads -> x2; WRITE_SIZE exact; both in KiB
    v["hbm_bytes_per_launch_corrected"] = (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --child",
           "workload": "d=500 CGD-15 w=64 p=56, one solve", "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "kernels": hbm}, open(os.path.join(P, tag + "_bench_pmc_hbm.json"), "w"), indent=1)


def derived(c):
    d = {}
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_LDS_IDX_ACTIVE") is not None:
        d["lds_busy_frac_of_sq_busy"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CYCLES"] if c["SQ_BUSY_CYCLES"] else None
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS"):
            if k in c:
                d[k + "_per_wave_cycle"] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_INSTS_LDS"):
        d["bank_conflict_cycles_per_lds_inst"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_INSTS_LDS"]
    return d


for dirs, name, what in ((["pmc_sq1", "pmc_sq2"], "mac_sq_counters.json", "python3 scripts/gpu_probe.py big (d=500, one CGD iteration)"),
                         (["pmc_quad1", "pmc_quad2"], "quad_sq_counters.json", "python3 scripts/gpu_probe.py chol (d=20 Cholesky x3: 4-wave latency kernels)")):
    c = counters(dirs)
    keep = {k: dict(v, **derived(v)) for k, v in c.items() if "gc_" in k}
    if not keep:
        continue                      # this pass was not part of the run (scripts/profile_bench.sh)
    json.dump({"command": "rocprofv3 --pmc <SQ counters, two passes> -- " + what, "note": "averages per launch; SQ_* summed over the chip",
               "kernels": keep}, open(os.path.join(P, tag + "_" + name), "w"), indent=1)
for k in hbm:
    if "mac_kernel" in k or "mack_kernel" in k:
        print(k, json.dumps(hbm[k]))

"""Copy the rocprofv3 summaries of a bench run from gpurun_out/ into profiles/ (tracked).

    python scripts/summarize_profiles.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <bench_json>
"""
import collections, csv, glob, json, os, shutil, sys
tag, stats_dir, fdir, wdir, bjson = sys.argv[1:6]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv"))[0], os.path.join(P, tag + "_bench_kernel_stats.csv"))
shutil.copy(bjson, os.path.join(P, tag + "_bench_line.json"))
out = {}
for d, cn in ((fdir, "FETCH_SIZE"), (wdir, "WRITE_SIZE")):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[cn + "_KB_avg_per_launch"] = sum(v) / len(v)
        out[k]["launches"] = len(v)
for k, v in out.items():
    f = v.get("FETCH_SIZE_KB_avg_per_launch", 0.0); w = v.get("WRITE_SIZE_KB_avg_per_launch", 0.0)
    # MI355X_MICROARCH.md (HBM): gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads -> x2; WRITE_SIZE exact
    v["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
           "workload": "d=500 CGD-15 w=64 p=56",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts wide coalesced reads at half)",
           "kernels": out}, open(os.path.join(P, tag + "_bench_pmc_hbm.json"), "w"), indent=1)
for k in out:
    if "mac_kernel" in k:
        print(k, json.dumps(out[k]))

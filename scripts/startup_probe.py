"""Start-up breakdown of the end-to-end runs: runs bin/linreg configurations with LINREG_TRACE=1 and
prints / stores the timeline of every party.  `--root DIR` points at another build (DIR/host/bin/linreg + DIR/csrc/*.so),
so two builds can be compared on one box.  Usage: python scripts/startup_probe.py [--root DIR] [--out FILE] [--reps N]"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--configs", default="c2,c3-ti,c3-ot")
    a = ap.parse_args()
    if a.root:
        bench.LINREG_EXE = os.path.join(os.path.abspath(a.root), "host", "bin", "linreg")
    IR = [] if os.environ.get("PROBE_NO_INPUT_RING") else ["--input_ring"]
    cfgs = {"c2": (1000, 20, [0, 10], "cholesky", 0, ["--table_ring"] + IR),
            "c3-ti": (10000, 100, [0, 50], "cgd", 15, ["--ti_ring", "--table_ring"] + IR),
            "c3-ot": (10000, 100, [0, 50], "cgd", 15, ["--ot_ring", "--table_ring"] + IR),
            "c1": (10, 5, [0, 1, 2], "cgd", 10, ["--ti_ring", "--table_ring"] + IR, dict(source=os.path.join(ROOT, "tests", "golden", "readme_example.in"))),
            "c4": (50000, 500, [0, 100, 200, 300, 400], "cgd", 20, ["--width_phase2=32", "--prec_phase2=30", "--ti_ring", "--table_ring"] + IR,
                   dict(prec2=30, w2=32))}
    bench.phase12_wall(np, "warm-up", 200, 4, [0, 2], "cholesky", 0, ["--table_ring"], 0)
    out = []
    for name in a.configs.split(","):
        n, d, starts, alg, iters, extra = cfgs[name][:6]
        kw = cfgs[name][6] if len(cfgs[name]) > 6 else {}
        for rep in range(a.reps if name != "c4" else min(a.reps, 2)):
            r = bench.phase12_wall(np, name, n, d, starts, alg, iters, extra, 0, **kw)
            ck = r.pop("_check", None)
            if ck:
                import shutil
                shutil.rmtree(ck["tmp"], ignore_errors=True)
            out.append(r)
            st = r.get("timeline", {}).get("steps", {})
            print("%-6s rep %d  wall %.3f s  %s" % (name, rep, r.get("phase12_wall_s", -1), r.get("error", "")))
            for k, v in st.items():
                print("    %-22s %8.3f s  (%s)" % (k, v["t_s"], v["party"]))
            pe = r.get("timeline", {}).get("process_end_s", {})
            ex = {tag: [t for t, w_ in lst if w_ == "exit"] for tag, lst in r.get("timeline", {}).get("marks", {}).items()}
            print("    teardown (exit mark -> process gone): " + ", ".join("%s %.0f ms" % (tag, 1e3 * (pe[tag] - ex[tag][0])) for tag in sorted(pe) if pe.get(tag) and ex.get(tag)))
            sys.stdout.flush()
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

"""timeline of a kernel trace (rocprofv3 --kernel-trace --output-format csv): who runs beside the MAC launches.
   python scripts/dbg/timeline.py <dir with *_kernel_trace.csv> [<dir of a second process> ...] [rows]
Several directories = several PROCESSES traced in one run (CSP and Evaluator of a two-process solve, scripts/exp/two_proc_trace.sh):
their events are merged on the GPU's clock, the queue column then reads <process>.<queue>."""
import csv, glob, os, sys
dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
rest = [a for a in sys.argv[1:] if not os.path.isdir(a)]
rows_wanted = int(rest[0]) if rest else 160
ev = []
files = []
for pi, d in enumerate(dirs):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    files.append(f)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        short = n.split("(")[0].replace("void gc::", "").replace("gc::", "")
        q = r.get("Queue_Id", "?")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, q if len(dirs) == 1 else "%d.%s" % (pi + 1, q)))
ev.sort()
f = " + ".join(files)
if len(dirs) == 1:
    # last solve: events after the last gc_input_kernel
    starts = [i for i, e in enumerate(ev) if e[2].startswith("gc_input_kernel")]
    ev = ev[starts[-1]:]
else:
    # two processes: from the first MAC kernel's neighbourhood on (start-up kernels of both processes lie far apart)
    firstm = [i for i, e in enumerate(ev) if "gc_mack_kernel" in e[2] or "gc_mac_kernel" in e[2]]
    ev = ev[max(0, (firstm[0] if firstm else 0) - 40):]
t0 = ev[0][0]
def is_mack(e): return "gc_mack_kernel" in e[2]
def role(e): return "G" if ("<true" in e[2] or "tabfill" in e[2]) else "E"
macks = [e for e in ev if is_mack(e)]
print("file", f, "events", len(ev), "solve span %.3f ms" % ((max(e[1] for e in ev) - t0) / 1e6))
tot = {"G": 0, "E": 0}
for e in macks: tot[role(e)] += e[1] - e[0]
print("MAC kernels: garbler %.2f ms evaluator %.2f ms (sum of durations)" % (tot["G"] / 1e6, tot["E"] / 1e6))
# union and pairwise overlap of MAC kernels
pts = sorted([(e[0], 1, role(e)) for e in macks] + [(e[1], -1, role(e)) for e in macks])
cnt = {"G": 0, "E": 0}; last = None; both = 0; anym = 0
for t, dlt, r in pts:
    if last is not None:
        if cnt["G"] and cnt["E"]: both += t - last
        if cnt["G"] or cnt["E"]: anym += t - last
    cnt[r] += dlt; last = t
print("some MAC kernel running %.2f ms, both roles' MAC kernels at once %.2f ms" % (anym / 1e6, both / 1e6))
# small kernels: time spent while a MAC kernel of the OTHER role runs, and stretch
small = [e for e in ev if not is_mack(e)]
def overl(a, b): return max(0, min(a[1], b[1]) - max(a[0], b[0]))
for r in ("G", "E"):
    mine = [e for e in small if role(e) == r]
    other = [m for m in macks if role(m) != r]
    dur = sum(e[1] - e[0] for e in mine)
    beside = sum(overl(e, m) for e in mine for m in other)
    span_idle = 0
    print("small kernels of %s: %d, %.2f ms of durations, %.2f ms of them beside a MAC kernel of the other role" % (r, len(mine), dur / 1e6, beside / 1e6))
print("%10s %10s %5s %s" % ("start_ms", "dur_ms", "q", "kernel"))
for e in ev[:rows_wanted]:
    print("%10.3f %10.3f %5s %s %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], role(e), e[2][:60]))

"""teardown cost of a HIP process (scripts/exp/hip_exit_probe.hip): time from the end of main to the process being gone"""
import subprocess, time, sys, os
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "hip_exit_probe")
def run(args, par=1):
    ps = [subprocess.Popen([exe] + [str(a) for a in args], stdout=subprocess.PIPE) for _ in range(par)]
    res = []
    for p in ps:
        out = p.stdout.readline()
        p.wait()
        res.append(time.monotonic() - float(out))
    return max(res) * 1e3
run([0, 0, 1, 0])
for par in (1, 4):
    for args, name in (([0, 0, 1, 0], "init + one kernel, return"), ([0, 0, 1, 1], "the same, _exit"), ([0, 0, 1, 2], "the same, hipDeviceReset + return"),
                       ([2, 0, 1, 0], "+ 2 streams, return"), ([8, 0, 1, 0], "+ 8 streams, return"), ([8, 0, 1, 1], "+ 8 streams, _exit"),
                       ([0, 4096, 1, 0], "+ 4 GiB allocated and freed, return"), ([0, 4096, 0, 0], "+ 4 GiB allocated, NOT freed, return"),
                       ([0, 4096, 0, 1], "+ 4 GiB allocated, NOT freed, _exit")):
        v = [run(args, par) for _ in range(3)]
        print("%d at once  %-44s teardown %6.0f %6.0f %6.0f ms" % (par, name, v[0], v[1], v[2]))
        sys.stdout.flush()

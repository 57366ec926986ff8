"""N > 1 on the GPU box: the driver only has one GPU per gpurun box, so two ranks share it (gloo
process group, LGC_BENCH_BACKEND=gloo); the data path is the one an 8-GPU node runs -- same seed on
all ranks, prefix garbled on rank 0, broadcast, imported, disjoint gate ids, all_gather.  The
hipIpc table ring across two devices runs only where two GPUs are visible."""
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import oracle_solve

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    from helpers import free_ports
    return free_ports(1)[0]


def _last_line(stdout):
    """what the driver does: the LAST line of stdout must be one complete JSON object, and short enough to survive its
    8 000-character tail (a 30 KB line went unparsed)"""
    lines = stdout.decode().rstrip("\n").splitlines()
    assert lines, "no output"
    assert len(lines[-1]) <= 4096, len(lines[-1])
    return json.loads(lines[-1])


def test_bench_two_ranks_share_one_gpu(tmp_path, oracle):
    env = dict(os.environ, LGC_BENCH_BACKEND="gloo", LGC_BENCH_DUMP=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0",
               LGC_BENCH_DETAIL_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # plain `python bench.py --gpus 2`, no launcher: bench.py starts its own ranks (what a driver without torchrun gets)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--dimension", "24", "--iters", "2", "--sweep-d", "6", "--sweep-iters", "3", "--sweep-lambdas", "5",
           "--no-cpu-baseline", "--no-traffic", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = _last_line(r.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["n_gpus"] == out["rccl_ranks"] or out["barrier_backend"] == "gloo"
    assert out["rccl_ranks"] is None and len(out["devices"]) == 1           # two ranks, one GPU, no RCCL group: said so
    sw = out["sweep64"]
    for k in ("create_s", "prefix_garble_s", "broadcast_s", "block_s", "gather_s"):
        assert sw[k] >= 0.0
    assert sw["block_s"] > 0.0 and sw["prefix_garble_s"] > 0.0
    assert sw["n_gpus"] == 2 and sw["lambdas"] == 5 and sw["prefix_bytes_broadcast"] > 0
    detail = json.load(open(os.path.join(str(tmp_path), out["detail"])))      # the long form: bench_detail_n2.json
    assert "broadcast" in detail["sweep64"]["collectives"] and detail["value"] == pytest.approx(out["value"], rel=1e-4)
    dumps = [json.load(open(os.path.join(str(tmp_path), "sweep_rank%d.json" % k))) for k in (0, 1)]
    assert dumps[0]["beta"] == dumps[1]["beta"]                      # every rank holds the gathered results
    d0 = dumps[0]
    d, w, p, iters = d0["d"], d0["width"], d0["precision"], d0["iters"]
    sh = np.array(d0["shares"], dtype=np.uint64)
    tot = sh.sum(axis=0, dtype=np.uint64)
    T = d * (d + 1) // 2
    for k, lam in enumerate(d0["lambdas"]):                          # rank 0 ran 0..2, rank 1 ran 3..4 from the broadcast prefix
        exp, _, _ = oracle_solve(oracle, tot[:T], tot[T:], d, w, p, "cgd", iters, lam, 1)
        assert [int(v) for v in exp] == d0["beta"][k], (k, lam)


def test_bench_eight_ranks_rehearse_the_full_sweep_on_one_gpu(tmp_path, oracle):
    """BASELINE config 5 at its real size -- d = 100, CGD-15, 64 lambdas -- as EIGHT ranks (gloo), 8 lambdas each, all on the
    one GPU of the box: the N = 8 code path of bench.py / python/sweep.py (one seed, prefix garbled on rank 0, broadcast,
    imported on seven ranks, blocks at gate-step offsets 8k x stride, all_gather) with every one of the 64 results checked
    against the oracle.  Each rank's table ring gets 512 MiB of run-ahead room instead of 8 GiB so that eight of them fit
    288 GB together (LGC_RING_SLACK_MB); on an 8-GPU node the default applies.  The headline part of the bench is kept tiny."""
    env = dict(os.environ, LGC_BENCH_BACKEND="gloo", LGC_BENCH_DUMP=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0", LGC_RING_SLACK_MB="512",
               LGC_BENCH_DETAIL_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
           "--dimension", "24", "--iters", "2", "--no-cpu-baseline", "--no-traffic", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = _last_line(r.stdout)            # the line the driver's SCALE run will have to parse: whole, short, every key there
    assert out["n_gpus"] == 8 and out["barrier_backend"] == "gloo" and out["rccl_ranks"] is None and len(out["devices"]) == 1
    for k in ("metric", "value", "unit", "ms_per_step", "scaling", "rccl_ranks", "devices", "config", "roofline", "sweep64"):
        assert k in out, k
    sw = out["sweep64"]
    assert (sw["n_gpus"], sw["lambdas"], sw["d"], sw["iterations"]) == (8, 64, 100, 15)
    for k in ("seconds", "create_s", "prefix_garble_s", "broadcast_s", "block_s", "gather_s"):
        assert sw[k] >= 0.0, k
    assert sw["prefix_bytes_broadcast"] > 10e6 and sw["block_s"] > 0 and sw["block_lambdas"] == 8
    # the N = 1 model's prediction for eight GPUs rides in the same line (profiles/sweep_model.json here: no N = 1 run
    # left a detail file in this directory), so the first real 8-GPU run shows measured / predicted
    assert sw["predicted_seconds"] > 0 and sw["measured_over_predicted"] == pytest.approx(sw["seconds"] / sw["predicted_seconds"], rel=1e-3)
    assert "8 lambdas per rank" in json.load(open(os.path.join(str(tmp_path), out["detail"])))["sweep64"]["sharding"]
    dumps = [json.load(open(os.path.join(str(tmp_path), "sweep_rank%d.json" % k))) for k in range(8)]
    assert all(dk["beta"] == dumps[0]["beta"] for dk in dumps)               # every rank holds the gathered results
    d0 = dumps[0]
    d, w, p, iters = d0["d"], d0["width"], d0["precision"], d0["iters"]
    tot = np.array(d0["shares"], dtype=np.uint64).sum(axis=0, dtype=np.uint64)
    T = d * (d + 1) // 2
    assert len(d0["lambdas"]) == 64
    for k, lam in enumerate(d0["lambdas"]):
        exp, _, _ = oracle_solve(oracle, tot[:T], tot[T:], d, w, p, "cgd", iters, lam, 1)
        assert [int(v) for v in exp] == d0["beta"][k], (k, lam)


def test_bench_launcher_and_self_launch_agree(tmp_path):
    """the torch.distributed.run form the driver documents still works (WORLD_SIZE set: no second level of children)"""
    env = dict(os.environ, LGC_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", LGC_BENCH_DETAIL_DIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--dimension", "12", "--iters", "1", "--no-sweep", "--no-cpu-baseline", "--no-traffic", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = _last_line(r.stdout)
    assert out["n_gpus"] == 2 and out["barrier_backend"] == "gloo" and out["rccl_ranks"] is None


def test_bench_self_launch_reports_a_failing_rank():
    """--gpus 2 over RCCL on a one-GPU box cannot work (two ranks, one device): the launcher must say so with a non-zero
    exit code instead of printing a one-rank line"""
    import linreg_gc
    if linreg_gc.device_count() >= 2:
        pytest.skip("this box has two GPUs: the run would succeed")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LGC_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--dimension", "12",
           "--iters", "1", "--no-sweep", "--no-cpu-baseline", "--no-traffic", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode != 0
    assert not [l for l in r.stdout.decode().splitlines() if l.startswith("{") and '"n_gpus": 1' in l]


def test_bench_collectives_through_rccl_one_rank(tmp_path, oracle):
    """The box has one GPU, so the RCCL group has one rank -- but every collective of the N > 1 path (barrier, broadcast
    of the seed and of the exported prefix in device memory, all_gather, all_reduce of the timings) goes through RCCL on
    device tensors here, which the two-rank gloo run above cannot show."""
    env = dict(os.environ, LGC_BENCH_FORCE_DIST="1", LGC_BENCH_DUMP=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0", LGC_BENCH_DETAIL_DIR=str(tmp_path))
    env.pop("LGC_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--dimension", "24", "--iters", "2", "--sweep-d", "6", "--sweep-iters", "3", "--sweep-lambdas", "5",
           "--no-cpu-baseline", "--no-traffic", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = _last_line(r.stdout)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["barrier_backend"] == "nccl"
    assert out["rccl_ranks"] == 1 and len(out["devices"]) == 1
    d0 = json.load(open(os.path.join(str(tmp_path), "sweep_rank0.json")))
    d, w, p, iters = d0["d"], d0["width"], d0["precision"], d0["iters"]
    tot = np.array(d0["shares"], dtype=np.uint64).sum(axis=0, dtype=np.uint64)
    T = d * (d + 1) // 2
    for k, lam in enumerate(d0["lambdas"]):
        exp, _, _ = oracle_solve(oracle, tot[:T], tot[T:], d, w, p, "cgd", iters, lam, 1)
        assert [int(v) for v in exp] == d0["beta"][k], (k, lam)


def test_table_ring_across_two_gpus(tmp_path, oracle, lgc):
    """CSP on GPU 0, Evaluator on GPU 1, garbled tables through the hipIpc ring (xGMI peer access)"""
    if lgc.device_count() < 2:
        pytest.skip("needs two visible GPUs (the driver's multi-GPU node; one-GPU boxes skip)")
    from test_host import HOST, _free_ports
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(23)
    d, iters, w, p = 6, 4, 64, 56
    X = rng.standard_normal((60, d)); X /= np.abs(X).max(axis=0)
    A = X.T @ X / (60 * d) + np.eye(d) * 1e-2
    b = A @ rng.random(d)
    path = str(tmp_path / "ls.in")
    with open(path, "w") as f:
        f.write("%d %d\n" % (d, d))
        np.savetxt(f, A, fmt="%.17g")
        f.write("%d\n" % d)
        np.savetxt(f, b[None, :], fmt="%.17g")
        f.write("%d\n" % d)
        np.savetxt(f, np.zeros((1, d)), fmt="%g")
    port = _free_ports(1)[0]
    exe = os.path.join(HOST, "bin", "test_linear_system")
    procs = [subprocess.Popen([exe, str(port), str(k), path, "cgd", str(iters), str(p), "--host=127.0.0.1", "--table_ring=4"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, LINREG_DEVICE=str(k - 1)))
             for k in (1, 2)]
    outs = [q.communicate(timeout=300) for q in procs]
    for q, (o, e) in zip(procs, outs):
        assert q.returncode == 0, e.decode()[-1000:]
    got = re.findall("-?[0-9]+\\.[0-9]+", outs[1][0].decode().strip().splitlines()[-1])
    aq = np.array([oracle.lib.orc_double_to_fixed(float(A[i, j]), p, w) for i in range(d) for j in range(i + 1)], dtype=np.int64)
    bq = np.array([oracle.lib.orc_double_to_fixed(float(v), p, w) for v in b], dtype=np.int64)
    assert got == ["%.15f" % (int(v) / 2.0 ** p) for v in oracle.cgd(aq, bq, d, p, w, iters)]

"""BASELINE.json config 5: 64 regularisation values x config 3 (d=100, CGD-15, 64-bit), one circuit
per lambda, dealt to the ranks in contiguous blocks (python/sweep.py).  Single GPU:
    python tests/tools/gpu_c5_sweep.py [--concurrency K]
N GPUs of one node:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tests/tools/gpu_c5_sweep.py
All circuits of a rank's block are garbled and evaluated as one merged program
(lgc_solver_create_sweep): the latency-bound divider / reveal launches of different circuits share
launches."""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import linreg_gc as lgc
import sweep

ap = argparse.ArgumentParser()
ap.add_argument("--group", type=int, default=0, help="circuits per merged program (0: the whole block)")
ap.add_argument("--lambdas", type=int, default=64)
ap.add_argument("--d", type=int, default=100)
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--check", type=int, default=2, help="compare this many circuits with the CPU oracle")
ap.add_argument("--gate_hash", default="aes128", help="aes128 (default, the reference's) or chaskey12 (lgc_set_gate_hash)")
args = ap.parse_args()
lgc.set_gate_hash(args.gate_hash)

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
dist = None
if world > 1:
    import torch, torch.distributed as dist
    torch.cuda.set_device(local % torch.cuda.device_count())
    dist.init_process_group("nccl")
device = local % max(1, lgc.device_count())

d, n, P, w, p, iters = args.d, args.n, 2, 64, 56, 15
rng = np.random.default_rng(0)
# phase-1 outputs of a synthetic regression (generate_tests.py distribution), split into P shares
import orc
from helpers import oracle_solve, split_shares, synth_system
oracle = orc.load()
A, b = synth_system(oracle, rng, n, d, w, p)
sh = split_shares(rng, A, b, P, w)
lams = sweep.c5_lambdas(args.lambdas)
lo, hi = sweep.partition(len(lams), world, rank)

gates = {}
def run_all():
    """this rank's block as ONE merged program (lgc_solver_create_sweep), in groups of --group circuits"""
    out = {}
    ks = list(range(lo, hi))
    G = args.group if args.group > 0 else max(1, len(ks))
    for g0 in range(0, len(ks), G):
        blk = ks[g0:g0 + G]
        sysm = lgc.make_system(d, w, p, "cgd", iters, 0.0, P, 1, 0, 0)
        sv = lgc.Solver(sysm, seed=os.urandom(16), device=device, lambdas=[lams[k] for k in blk])
        sv.set_shares(sh); sv.run()
        res = sv.beta(); st = sv.stats(); sv.close()
        for t, k in enumerate(blk):
            out[k] = res[t]; gates[k] = st["and_gates"] / len(blk)
    return out

if hi > lo:                                             # warm-up (module load, constants)
    sv = lgc.Solver(lgc.make_system(d, w, p, "cgd", iters, 0.0, P, 1, 0, 0), device=device, lambdas=[lams[lo]])
    sv.set_shares(sh); sv.run(); sv.close()
if dist is not None: dist.barrier()
t0 = time.time()
out = run_all()
if dist is not None:
    import torch
    torch.cuda.synchronize(); dist.barrier()
dt = time.time() - t0
if dist is not None:
    tt = torch.tensor([dt], device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
ok = None
if rank == 0 and args.check:
    ok = True
    for k in list(range(lo, hi))[:args.check]:
        exp, _, _ = oracle_solve(oracle, A, b, d, w, p, "cgd", iters, lams[k], 1)
        ok = ok and [int(v) for v in exp] == [int(v) for v in out[k]]
if rank == 0:
    total_gates = sum(gates.values()) / max(1, len(gates)) * len(lams)
    print(json.dumps(dict(config="c5", lambdas=len(lams), d=d, iters=iters, n_gpus=world, group=args.group,
                          seconds=round(dt, 3), circuits_per_s=round(len(lams) / dt, 2), and_per_s=total_gates / dt,
                          exact_vs_oracle=ok)), flush=True)

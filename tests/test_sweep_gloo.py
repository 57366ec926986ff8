"""N > 1 path on CPU: world_size-2 gloo run of the per-lambda sharding (the per-circuit solve is
replaced by the oracle here; on GPUs it is the HIP solver, see test_gpu_sweep_single_rank)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import orc, sweep
    from helpers import oracle_solve, split_shares, synth_system
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = orc.load()
    rng = np.random.default_rng(0)
    w, p, d, n = 64, 56, 4, 30
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    lams = sweep.c5_lambdas(7)                       # ragged: 7 circuits over 2 ranks -> 4 + 3
    calls = []

    def solve(sh, block, first):
        calls.extend(range(first, first + len(block)))
        tot = sh.sum(axis=0, dtype=np.uint64)
        return np.array([oracle_solve(oracle, tot[:d * (d + 1) // 2], tot[d * (d + 1) // 2:], d, w, p, "cgd", 3, lam, 1)[0]
                         for lam in block])
    res = sweep.lambda_sweep(shares, lams, d, solve, dist=dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, calls, res.tolist()))


class _FakeSolver:
    """stands in for linreg_gc.Solver on CPU: its "garbled prefix" is the share sums, so a non-root rank can
    only produce right answers if the broadcast delivered them (it is never given the shares)"""

    def __init__(self, oracle, d, w, p, iters, block, first, seed, log):
        self.o, self.d, self.w, self.p, self.iters, self.block = oracle, d, w, p, iters, block
        self.sums = None
        log.append(("create", first, len(block), seed))
        self.log = log

    def prefix_bytes(self):
        return 8 * (self.d * (self.d + 1) // 2 + self.d)

    def set_shares(self, sh):
        self.shares = np.asarray(sh, dtype=np.uint64)
        self.log.append(("set_shares",))

    def prefix_garble(self):
        self.sums = self.shares.sum(axis=0, dtype=np.uint64)
        self.log.append(("garble",))

    def prefix_export(self, ptr):
        import ctypes
        ctypes.memmove(ptr, self.sums.ctypes.data, self.sums.nbytes)

    def prefix_import(self, ptr):
        import ctypes
        self.sums = np.zeros(self.prefix_bytes() // 8, dtype=np.uint64)
        ctypes.memmove(self.sums.ctypes.data, ptr, self.sums.nbytes)
        self.log.append(("import",))

    def run(self):
        from helpers import oracle_solve
        T = self.d * (self.d + 1) // 2
        self.res = np.array([oracle_solve(self.o, self.sums[:T], self.sums[T:], self.d, self.w, self.p, "cgd", self.iters, lam, 1)[0]
                             for lam in self.block])

    def beta(self):
        return self.res

    def stats(self):
        return {}

    def close(self):
        pass


def _worker_shared(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import orc, sweep
    from helpers import split_shares, synth_system
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = orc.load()
    rng = np.random.default_rng(0)
    w, p, d, n = 64, 56, 4, 30
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w) if rank == 0 else None       # only rank 0 holds the inputs
    lams = sweep.c5_lambdas(7)
    log = []
    res = sweep.shared_prefix_sweep(shares, lams, d, lambda blk, first, seed: _FakeSolver(oracle, d, w, p, 3, blk, first, seed, log), dist=dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, log, res.tolist()))


def test_shared_prefix_sweep_two_ranks_gloo(oracle):
    """host logic of the multi-GPU sweep: one seed for all ranks, prefix garbled on rank 0 only, broadcast,
    imported by the others, blocks offset by their first circuit, results gathered on every rank"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_shared, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    outs = sorted(q.get(timeout=120) for _ in range(2))
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    log0, log1 = outs[0][1], outs[1][1]
    assert log0[0][:3] == ("create", 0, 4) and log1[0][:3] == ("create", 4, 3)
    assert log0[0][3] == log1[0][3] and len(log0[0][3]) == 16              # the same garbler seed
    assert [e[0] for e in log0] == ["create", "set_shares", "garble"]     # rank 0 garbles, never imports
    assert [e[0] for e in log1] == ["create", "import"]                   # rank 1 never sees the shares
    assert outs[0][2] == outs[1][2]
    sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
    import sweep
    from helpers import oracle_solve, synth_system
    rng = np.random.default_rng(0)
    A, b = synth_system(oracle, rng, 30, 4, 64, 56)
    exp = [oracle_solve(oracle, A, b, 4, 64, 56, "cgd", 3, lam, 1)[0].tolist() for lam in sweep.c5_lambdas(7)]
    assert outs[0][2] == exp


def test_partition_is_contiguous_and_complete():
    sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
    import sweep
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            blocks = [sweep.partition(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
    assert sweep.partition(64, 8, 3) == (24, 32)
    lams = sweep.c5_lambdas()
    assert len(lams) == 64 and abs(lams[0] - 1e-6) < 1e-18 and abs(lams[-1] - 1.0) < 1e-12


def test_lambda_sweep_two_ranks_gloo(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    outs = sorted(q.get(timeout=120) for _ in range(2))
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    assert outs[0][1] == [0, 1, 2, 3] and outs[1][1] == [4, 5, 6]      # each rank solved only its block
    assert outs[0][2] == outs[1][2]                                    # every rank holds all results
    # and they are what a single process computes
    sys.path.insert(0, os.path.join(ROOT, "linreg-mpc_amd", "python"))
    import sweep
    from helpers import oracle_solve, split_shares, synth_system
    rng = np.random.default_rng(0)
    w, p, d, n = 64, 56, 4, 30
    A, b = synth_system(oracle, rng, n, d, w, p)
    exp = [oracle_solve(oracle, A, b, d, w, p, "cgd", 3, lam, 1)[0].tolist() for lam in sweep.c5_lambdas(7)]
    assert outs[0][2] == exp
    assert len(set(map(tuple, exp))) > 1                               # lambda actually matters


@pytest.mark.gpu
def test_gpu_sweep_single_rank(lgc, oracle):
    import sweep
    from helpers import oracle_solve, split_shares, synth_system
    rng = np.random.default_rng(1)
    w, p, d, n = 64, 56, 5, 40
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 2, w)
    lams = sweep.c5_lambdas(4)
    solve = sweep.gpu_solve_factory(d, w, p, "cgd", 4, 2, 0)
    res = sweep.lambda_sweep(shares, lams, d, solve)
    exp = [oracle_solve(oracle, A, b, d, w, p, "cgd", 4, lam, 1)[0].tolist() for lam in lams]
    assert res.tolist() == exp
    # shared-prefix path, single rank: prefix garbled separately, then the merged program
    import torch
    st = {}
    res = sweep.shared_prefix_sweep(shares, lams, d, sweep.gpu_block_solver_factory(d, w, p, "cgd", 4, 2, 0), tensor_device="cuda", stats=st)
    assert res.tolist() == exp and st["prefix_bytes"] > 0
    # a block that does not start at circuit 0, its prefix exported by one solver and imported by another
    blk = lgc.Solver(lgc.make_system(d, w, p, "cgd", 4, 0.0, 2, 1, 0, 0), seed=bytes(range(16)), lambdas=lams[0:1], first=0)
    blk.set_shares(shares); blk.prefix_garble()
    buf = torch.empty(blk.prefix_bytes(), dtype=torch.uint8, device="cuda")
    blk.prefix_export(buf.data_ptr())
    oth = lgc.Solver(lgc.make_system(d, w, p, "cgd", 4, 0.0, 2, 1, 0, 0), seed=bytes(range(16)), lambdas=lams[1:], first=1)
    oth.prefix_import(buf.data_ptr())                 # never given the shares
    oth.run(); blk.run()
    assert blk.beta().tolist() == exp[0:1] and oth.beta().tolist() == exp[1:]
    blk.close(); oth.close()
    # 32-bit direct solver, merged program
    w, p = 32, 28
    A, b = synth_system(oracle, rng, n, d, w, p)
    shares = split_shares(rng, A, b, 3, w)
    lams = [0.0, 0.01, 0.5]
    s = lgc.Solver(lgc.make_system(d, w, p, "cholesky", 0, 9.0, 3, 1, 0, 0), seed=bytes(range(16)), lambdas=lams)
    s.set_shares(shares); s.run()
    exp = [oracle_solve(oracle, A, b, d, w, p, "cholesky", 0, lam, 1)[0].tolist() for lam in lams]
    assert s.beta().tolist() == exp
    with pytest.raises(RuntimeError):
        lgc.Solver(lgc.make_system(d, w, p, "cgd", 2, 0.0, 2, 0, 0, 0), lambdas=lams)

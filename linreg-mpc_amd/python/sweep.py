"""Sharding of the per-lambda sweep over the GPUs of one node.

A single solve is one dependent chain and stays on one GPU.  The circuits of a regularisation sweep
differ only in lambda, a public constant added to the diagonal AFTER the data providers' shares are
summed (reference src/linear.oc:52-57), so they share a prefix: the input wire labels, the
share-summation launches and the division by the public normalizer.  shared_prefix_sweep garbles and evaluates
that prefix ONCE, on rank 0, broadcasts what it leaves (RCCL over xGMI: both roles' words of the shared region --
no tables), and every rank then garbles and
evaluates its contiguous block of lambdas as one merged program; an all_gather collects the d-word
results.  One process per GPU (torch.distributed; backend "nccl" = RCCL on GPUs, "gloo" in the CPU
tests and single-GPU dry runs).  lambda_sweep is the collective-free variant (independent circuits,
e.g. bootstrap resamples): only the final gather communicates.
"""
import numpy as np


def partition(n_items, world, rank):
    """contiguous block [lo, hi) of rank `rank`; sizes differ by at most one"""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def c5_lambdas(count=64):
    """the 64-lambda sweep of BASELINE.json config 5: lambda_k = 10^(-6 + 6k/63)"""
    return [10.0 ** (-6.0 + 6.0 * k / (count - 1)) for k in range(count)]


def gpu_solve_factory(d, width, precision, algorithm, num_iterations, nshares, device, seed=None):
    """solve of this rank's block of lambdas on its GPU through the C ABI (data-provider input
    path): all circuits of the block are garbled and evaluated as ONE program
    (lgc_solver_create_sweep), so their latency-bound stages share launches"""
    import os
    import linreg_gc as lgc

    def solve(shares, lams, first_index):
        sysm = lgc.make_system(d, width, precision, algorithm, num_iterations, 0.0, nshares, 1, 0, 0)
        s = lgc.Solver(sysm, seed=seed or os.urandom(16), device=device, lambdas=lams)
        s.set_shares(shares)
        s.run()
        beta = s.beta()
        s.close()
        return beta
    return solve


def lambda_sweep(shares, lambdas, d, solve, dist=None, tensor_device="cpu"):
    """Solve one circuit per lambda; rank r takes the contiguous block partition(len, world, r) and
    hands it to solve(shares, block_of_lambdas, first_index) -> (len(block), d) int64.
    Returns an (len(lambdas), d) int64 array with every rank's results (gathered on all ranks)."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = partition(len(lambdas), world, rank)
    per = (len(lambdas) + world - 1) // world
    mine = np.zeros((per, d), dtype=np.int64)
    if hi > lo:
        mine[:hi - lo] = solve(shares, list(lambdas[lo:hi]), lo)
    if dist is None:
        return mine[:hi - lo]
    buf = torch.from_numpy(mine).to(tensor_device)
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    res = np.zeros((len(lambdas), d), dtype=np.int64)
    for r in range(world):
        rlo, rhi = partition(len(lambdas), world, r)
        res[rlo:rhi] = outs[r].cpu().numpy()[:rhi - rlo]
    return res


def _bcast_bytes(buf, dist, src=0):
    """broadcast a uint8 tensor in place; device tensors go over RCCL, or through the host when the
    process group is gloo (CPU tests, N ranks sharing one GPU)"""
    if dist is None:
        return
    if buf.is_cuda and dist.get_backend() != "nccl":
        host = buf.cpu()
        dist.broadcast(host, src=src)
        buf.copy_(host)
    else:
        dist.broadcast(buf, src=src)


def gpu_block_solver_factory(d, width, precision, algorithm, num_iterations, nshares, device):
    """make_solver for shared_prefix_sweep: this rank's block on its GPU through the C ABI
    (lgc_solver_create_sweep_at: same seed on all ranks, gate ids offset by the block's first circuit)"""
    import linreg_gc as lgc

    def make(block, first, seed):
        sysm = lgc.make_system(d, width, precision, algorithm, num_iterations, 0.0, nshares, 1, 0, 0)
        return lgc.Solver(sysm, seed=seed, device=device, lambdas=block, first=first)
    return make


def shared_prefix_sweep(shares, lambdas, d, make_solver, dist=None, tensor_device="cpu", seed=None, stats=None):
    """One circuit per lambda with the lambda-independent prefix garbled once.

    shares: (nshares, T + d) uint64, needed on rank 0 only (the other ranks receive the words
    the garbled and evaluated prefix left for both roles, not the inputs and not its tables).  make_solver(block_of_lambdas, first_index, seed16) returns an object with
    set_shares / prefix_bytes / prefix_garble / prefix_export(ptr) / prefix_import(ptr) / run / beta /
    close (linreg_gc.Solver).  Returns the (len(lambdas), d) int64 results on every rank.
    stats (dict, optional) receives this rank's wall-clock per phase: create_s (program + device memory, all ranks at
    once), prefix_garble_s (rank 0: garble + export; the other ranks have their solver and receive buffer ready and
    wait at the broadcast), broadcast_s (the collective as this rank sees it), block_s, gather_s."""
    import os
    import time
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = partition(len(lambdas), world, rank)
    clock = time.perf_counter
    t = {}
    # one garbler seed for the whole sweep: the ranks share the prefix, hence R and the input labels
    sd = torch.tensor(list(seed if seed is not None else os.urandom(16)), dtype=torch.uint8)
    if dist is not None:
        if dist.get_backend() == "nccl":
            sd = sd.to(tensor_device)
        dist.broadcast(sd, src=0)
    seed16 = bytes(sd.cpu().tolist())
    t0 = clock()
    solver = make_solver(list(lambdas[lo:hi]), lo, seed16) if hi > lo else None
    # the size of the prefix is a function of the program alone: every rank with a block knows it without a message;
    # a rank without one (more ranks than lambdas) builds no solver and takes no part in the broadcast's payload
    use_bcast = world > 1 or (dist is not None and os.environ.get("LGC_BENCH_FORCE_DIST") == "1")   # one-rank hardware check
    nb = torch.tensor([solver.prefix_bytes() if solver is not None else 0], dtype=torch.int64)
    if dist is not None and world > len(lambdas):
        if dist.get_backend() == "nccl":
            nb = nb.to(tensor_device)
        dist.all_reduce(nb, op=dist.ReduceOp.MAX)
    nbytes = int(nb.item())
    buf = torch.empty(nbytes, dtype=torch.uint8, device=tensor_device) if use_bcast else None
    t["create_s"] = clock() - t0
    t0 = clock()
    if rank == 0:
        solver.set_shares(shares)
        solver.prefix_garble()
        if use_bcast:
            solver.prefix_export(buf.data_ptr())
    t["prefix_garble_s"] = clock() - t0
    t0 = clock()
    if use_bcast:
        _bcast_bytes(buf, dist)
        if buf.is_cuda:
            torch.cuda.synchronize()
        if solver is not None and rank != 0:
            solver.prefix_import(buf.data_ptr())
        del buf
    t["broadcast_s"] = clock() - t0
    per = (len(lambdas) + world - 1) // world
    mine = np.zeros((per, d), dtype=np.int64)
    t0 = clock()
    if solver is not None:
        solver.run()
        mine[:hi - lo] = solver.beta()
        if stats is not None:
            stats.update(solver.stats())
        solver.close()
    t["block_s"] = clock() - t0
    if stats is not None:
        stats["prefix_bytes"] = nbytes
    if dist is None:
        t["gather_s"] = 0.0
        if stats is not None:
            stats.update(t)
        return mine[:hi - lo]
    t0 = clock()
    out = torch.from_numpy(mine)
    if dist.get_backend() == "nccl":
        out = out.to(tensor_device)
    outs = [torch.empty_like(out) for _ in range(world)]
    dist.all_gather(outs, out)
    res = np.zeros((len(lambdas), d), dtype=np.int64)
    for r in range(world):
        rlo, rhi = partition(len(lambdas), world, r)
        res[rlo:rhi] = outs[r].cpu().numpy()[:rhi - rlo]
    t["gather_s"] = clock() - t0
    if stats is not None:
        stats.update(t)
    return res

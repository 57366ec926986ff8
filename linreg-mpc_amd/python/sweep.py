"""Sharding of independent circuits over the GPUs of one node.

A single solve is one dependent chain and stays on one GPU.  Independent circuits -- one per
regularisation value lambda (lambda is a public constant added to the diagonal, reference
src/linear.oc:52-57) or per bootstrap resample -- are dealt to ranks in contiguous blocks; there is
no data-path collective: the only communication is the final gather of the revealed d-word results.
One process per GPU (torch.distributed; backend "nccl" = RCCL on GPUs, "gloo" in the CPU tests).
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

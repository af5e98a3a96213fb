"""Multi-GPU host logic for the NanoGICP path.  One process per GPU; torch.distributed
("nccl" == RCCL over xGMI on the GPU node, "gloo" in CPU tests) is plumbing only.

Two ways the path shards:
 1. independent alignments (keyframes / scans): `partition_items` deals them to ranks, no data-path
    collective; `gather_results` collects the 4x4 results for reporting.
 2. one large alignment, point-sharded: the SOURCE points are split into contiguous blocks
    (`shard_bounds`), the target (+ covariances) is replicated, every rank runs the fused pass on its
    block and the 32-double partial-sum vector {H upper-tri 21, b 6, y0, yi, candidates, valid, 0} is
    summed with one all-reduce per pass (256 B: latency-bound; xGMI link bandwidth is irrelevant);
    every rank then advances the identical LM state machine, so poses stay bit-identical without a
    broadcast (`sharded_align`).
"""
from __future__ import annotations

import numpy as np

SUMS_LEN = 32  # include/ngicp.h: ngicp_sharded_pass


def partition_items(n_items: int, world_size: int, rank: int) -> list[int]:
    """Round-robin deal of independent alignments to ranks."""
    return list(range(rank, n_items, world_size))


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one point."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def tri21(r: int, c: int) -> int:
    """Index of H(r,c), r <= c, in the packed upper triangle (csrc/ngicp_pass.h tri21)."""
    return r * 6 - (r * (r - 1)) // 2 + (c - r)


def pack_sums(H: np.ndarray, b: np.ndarray, y0: float, yi: float = 0.0, cand: float = 0.0, valid: float = 0.0) -> np.ndarray:
    v = np.zeros(SUMS_LEN)
    for r in range(6):
        for c in range(r, 6):
            v[tri21(r, c)] = H[r, c]
    v[21:27] = b
    v[27], v[28], v[29], v[30] = y0, yi, cand, valid
    return v


def unpack_sums(v: np.ndarray):
    H = np.zeros((6, 6))
    for r in range(6):
        for c in range(r, 6):
            H[r, c] = H[c, r] = v[tri21(r, c)]
    return H, np.array(v[21:27]), float(v[27]), float(v[28])


def gather_results(T_local: list[np.ndarray], dist, device=None):
    """All ranks' 4x4 results -> list on every rank (reporting only; not on the data path)."""
    import torch
    t = torch.tensor(np.stack(T_local).astype(np.float32) if T_local else np.zeros((0, 4, 4), np.float32), device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, torch.tensor([t.shape[0]], dtype=torch.int64, device=device))
    m = max(int(c.item()) for c in counts)
    pad = torch.zeros((m, 4, 4), dtype=torch.float32, device=device)
    pad[: t.shape[0]] = t
    outs = [torch.zeros_like(pad) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, pad)
    return [o[: int(c.item())].cpu().numpy() for o, c in zip(outs, counts)]


class _DevicePointer:
    """A raw device allocation as something torch can alias (CUDA array interface)."""
    def __init__(self, ptr: int, n_doubles: int):
        self.__cuda_array_interface__ = {"shape": (int(n_doubles),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def device_doubles(ptr: int, n_doubles: int, torch_device):
    """torch tensor aliasing `n_doubles` FP64 values at device pointer `ptr` (no copy)."""
    import torch
    return torch.as_tensor(_DevicePointer(ptr, n_doubles), device=torch.device(torch_device))


def sharded_covariances(engine, which: int, dist, torch_device):
    """K1 sharded over ranks (the loop being split is impl/nano_gicp_impl.hpp:309-354).  `engine` holds the WHOLE
    cloud as its source (which = 0) or target (which = 1) on every rank - the k-NN of a point looks at all of it, and the index
    build is deterministic, so the packed [n][6] FP64 covariance array has the same layout everywhere.  Rank r computes the rows of
    the sorted positions shard_bounds(n, world, r) in place, the blocks are exchanged (one all_gather_into_tensor when they are of
    equal size, else one broadcast per rank), and the set is committed as the cloud's covariances.  The engine offers
    covsShardBegin / covsShardTensor / covsShardCompute / covsShardCommit (the HIP engine's NanoGICP, or a stand-in)."""
    import contextlib
    import torch
    dev = torch.device(torch_device)
    world, rank = dist.get_world_size(), dist.get_rank()
    ptr, n = engine.covsShardBegin(which)
    covs = engine.covsShardTensor(which, dev) if hasattr(engine, "covsShardTensor") else device_doubles(ptr, n * 6, dev)
    lo, hi = shard_bounds(n, world, rank)
    if dev.type == "cuda":  # engine kernel and collective ordered on one (non-null) stream, as in sharded_align
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        ctx, stream = torch.cuda.stream(side), side.cuda_stream
    else:
        side, ctx, stream = None, contextlib.nullcontext(), 0
    with ctx:
        engine.covsShardCompute(which, lo, hi, stream)
        if world > 1:
            if n % world == 0:
                dist.all_gather_into_tensor(covs, covs[lo * 6:hi * 6].clone())
            else:
                for r in range(world):
                    a, b = shard_bounds(n, world, r)
                    dist.broadcast(covs[a * 6:b * 6], src=r)
        if side is not None:
            side.synchronize()
    engine.covsShardCommit(which)
    return n


MAX_SHARDED_PASSES = 64 * 10 + 8  # max_iterations x lm_max_iterations of the reference's defaults, plus the reporting lag


def sharded_align(engine, guess, dist, torch_device, max_passes: int = MAX_SHARDED_PASSES):
    """Point-sharded alignment.  `engine` already holds this rank's source block, the full target and both
    covariance sets; it offers sharded_begin / sharded_pass / sharded_step / sharded_finish (the HIP engine's
    NanoGICP, or any stand-in with the same protocol).  Per pass: the engine leaves its 32 partial sums in `sums`
    (device memory on the GPU node), ONE all-reduce adds the ranks' sums (what is reduced is the reference's own
    per-thread partial sum, impl/nano_gicp_impl.hpp:260-267), every rank advances the identical LM state machine.
    The engine reports `done` with a constant lag, identically on every rank, so all ranks leave the loop in the
    same step.  Returns the final 4x4 (identical on every rank)."""
    import contextlib
    import torch
    dev = torch.device(torch_device)
    # The engine's kernels and the collective must be ordered on ONE stream.  The C ABI reads a null stream handle as "the engine's
    # own stream" (hipStreamNonBlocking: not ordered against the legacy null stream), and torch's default stream IS the null handle,
    # so the loop runs on a stream of its own, made current for the collective and handed to the engine by its (non-zero) handle.
    if dev.type == "cuda":
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        ctx, stream = torch.cuda.stream(side), side.cuda_stream
        assert stream != 0
    else:
        side, ctx, stream = None, contextlib.nullcontext(), 0
    with ctx:
        sums = torch.zeros(SUMS_LEN, dtype=torch.float64, device=dev)
        engine.sharded_begin(guess)
        for _ in range(max_passes):
            engine.sharded_pass(sums.data_ptr(), stream)
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            if engine.sharded_step(sums.data_ptr(), stream):
                break
        else:
            raise RuntimeError("sharded_align: the alignment did not report completion")  # (bounded: a hang would take the node)
        T = engine.sharded_finish()  # synchronises the stepping stream
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)
    return T

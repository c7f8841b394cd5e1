"""
Multi-GPU sharding of the batch axis: independent tracks / width sweeps / IQP re-linearisations are
independent QPs, so the batch is block-partitioned over one-process-per-GPU ranks with NO data-path collective inside
the solve; exactly one all-gather (RCCL over xGMI when the backend is "nccl") collects the alpha vectors afterwards.
"""
import numpy as np


def shard_bounds(batch: int, world: int, rank: int) -> tuple:
    """Contiguous block partition [lo, hi) of `batch` items; the first `batch % world` ranks get one extra item."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def solve_sharded(problems: list, engine, dist=None, device=None, **opt_kw):
    """Every rank solves its contiguous shard of `problems` on its own GPU; alpha (padded to the longest track),
    curvature errors and status words are all-gathered so that every rank returns the full batch.

    dist: torch.distributed (initialised) or None for single-process.  Returns (alphas list, curv [B], status [B]).
    """
    import torch

    bsz = len(problems)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = shard_bounds(bsz, world, rank)
    nmax = max(int(np.asarray(p["reftrack"]).shape[0]) for p in problems)
    per = max(shard_bounds(bsz, world, r)[1] - shard_bounds(bsz, world, r)[0] for r in range(world))
    local = torch.zeros((per, nmax + 2), dtype=torch.float64)
    if hi > lo:
        alphas, curv, status, _ = engine.solve_batch(problems[lo:hi], **opt_kw)
        for k, a in enumerate(alphas):
            local[k, :a.shape[0]] = torch.from_numpy(a)
            local[k, nmax] = float(curv[k])
            local[k, nmax + 1] = float(status[k])
    if dist is None or world == 1:
        full = local[None]
    else:
        if device is not None:
            local = local.to(device)
        gathered = torch.zeros((world * per, nmax + 2), dtype=torch.float64, device=local.device)
        dist.all_gather_into_tensor(gathered, local)       # the single collective of the job
        full = gathered.cpu().view(world, per, nmax + 2)
    out_a, out_c, out_s = [], np.zeros(bsz), np.zeros(bsz, dtype=np.int32)
    for r in range(world):
        rlo, rhi = shard_bounds(bsz, world, r)
        for k in range(rhi - rlo):
            n = int(np.asarray(problems[rlo + k]["reftrack"]).shape[0])
            row = full[r, k].numpy()
            out_a.append(row[:n].copy())
            out_c[rlo + k] = row[nmax]
            out_s[rlo + k] = int(row[nmax + 1])
    return out_a, out_c, out_s

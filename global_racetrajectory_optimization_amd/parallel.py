"""
Multi-GPU sharding of the batch axis: independent tracks / width sweeps / IQP re-linearisations are
independent QPs, so the batch is block-partitioned over one-process-per-GPU ranks with NO data-path collective inside
the solve; exactly one all-gather (RCCL over xGMI when the backend is "nccl") collects the alpha vectors afterwards.

Device-resident since round 3: a rank's shard is packed once into padded tensors ON ITS DEVICE, solved there through the
device entry of the C ABI (mcq_solve_device_ragged_params), and the result tensor the engine wrote is what the all-gather
reads -- no numpy round trip between the solve and the collective.

Order of initialisation on a GPU: create the process group / touch torch.cuda BEFORE the Engine (the engine's library links its
own copy of the HIP runtime; torch's must be up first -- bench.py does the same).
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

    problems: dicts {reftrack [n,4], normvec [n,2], scaling [n] or None, kappa_bound, w_veh} (normvec given for all).
    dist: torch.distributed (initialised) or None for single-process.  device: torch device of this rank's tensors (the GPU the
    engine runs on; None / "cpu" with the gloo backend and the SIMT-interpreted library of the tests).
    Returns (alphas list, curv [B], status [B]).
    """
    import torch

    bsz = len(problems)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    dev = torch.device(device) if device is not None else torch.device("cpu")
    lo, hi = shard_bounds(bsz, world, rank)
    nmax = max(int(np.asarray(p["reftrack"]).shape[0]) for p in problems)
    per = max(shard_bounds(bsz, world, r)[1] - shard_bounds(bsz, world, r)[0] for r in range(world))
    # ---- this rank's shard, padded to [per][nmax] (slots beyond the shard: n = 0, skipped by the kernels) -------------------------
    ref = np.zeros((per, nmax, 4))
    nv = np.zeros((per, nmax, 2))
    sc = np.ones((per, nmax))
    ns = np.zeros(per, dtype=np.int32)
    kb = np.ones(per)
    wv = np.zeros(per)
    for k, p in enumerate(problems[lo:hi]):
        r = np.asarray(p["reftrack"], dtype=np.float64)
        n = r.shape[0]
        if p.get("normvec") is None:
            raise ValueError("solve_sharded: normvec is required")
        ref[k, :n] = r
        nv[k, :n] = p["normvec"]
        if p.get("scaling") is not None:
            sc[k, :n] = p["scaling"]
        ns[k], kb[k], wv[k] = n, float(p["kappa_bound"]), float(p["w_veh"])
    d_ref, d_nv, d_sc = (torch.from_numpy(a).to(dev) for a in (ref, nv, sc))
    d_n, d_kb, d_wv = (torch.from_numpy(a).to(dev) for a in (ns, kb, wv))
    # one result tensor per rank: [per][nmax + 2] = alpha | curv_error | status -- the engine writes alpha straight into its rows
    # (row stride nmax + 2 is not what the entry expects, so alpha gets its own [per][nmax] tensor and is packed on the device)
    d_alpha = torch.zeros((per, nmax), dtype=torch.float64, device=dev)
    d_curv = torch.zeros((per,), dtype=torch.float64, device=dev)
    d_status = torch.zeros((per,), dtype=torch.int32, device=dev)
    if hi > lo:
        engine.solve_device_ragged_params(per, nmax, d_n.data_ptr(), d_ref.data_ptr(), d_nv.data_ptr(), d_sc.data_ptr(), 0.0, 0.0,
                                          d_kb.data_ptr(), d_wv.data_ptr(), d_alpha.data_ptr(), d_curv.data_ptr(), d_status.data_ptr(),
                                          **opt_kw)
        engine.sync()                                       # the engine's stream is not torch's: done before the collective reads
    local = torch.cat((d_alpha, d_curv[:, None], d_status.to(torch.float64)[:, None]), dim=1).contiguous()
    if dist is None or world == 1:
        full = local[None]
    else:
        gathered = torch.zeros((world * per, nmax + 2), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gathered, local)       # the single collective of the job
        assert dist.get_world_size() == world
        full = gathered.view(world, per, nmax + 2)
    full = full.cpu().numpy()
    out_a, out_c, out_s = [], np.zeros(bsz), np.zeros(bsz, dtype=np.int32)
    for r in range(world):
        rlo, rhi = shard_bounds(bsz, world, r)
        for k in range(rhi - rlo):
            n = int(np.asarray(problems[rlo + k]["reftrack"]).shape[0])
            row = full[r, k]
            out_a.append(row[:n].copy())
            out_c[rlo + k] = row[nmax]
            out_s[rlo + k] = int(row[nmax + 1])
    return out_a, out_c, out_s

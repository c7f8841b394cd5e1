"""
Multi-GPU sharding of the batch axis: independent tracks / width sweeps / IQP re-linearisations are
independent QPs, so the batch is block-partitioned over one-process-per-GPU ranks with NO data-path collective inside
the solve; exactly one all-gather collects the alpha vectors afterwards.

One process per GPU, one Engine per process.  Round 4: the gather is the ENGINE'S OWN -- RCCL's ncclAllGather behind the C ABI
(mcq_comm_init / mcq_comm_allgather, include/mcq.h), on a comm stream of the engine ordered behind the solve that filled the send
buffer; the receive buffer is read after mcq_comm_wait.  Nothing of torch touches the GPU: the shard lives in memory of the engine's HIP runtime (mcq_device_alloc), so there is
no second runtime in the process, no stream of another runtime to order against and no initialisation order to get
right.  The launcher's process group (torch.distributed, any backend -- gloo is enough) is only the rendezvous that carries rank
0's 128-byte RCCL id to the other ranks (`init_engine_comm`).

The CPU tests (SIMT-interpreted library, world 2 and 4) run this very path: $MCQ_RCCL_LIB points the engine at tests/stub/librccl_stub_sync.so,
a shared-memory stand-in for the five RCCL entry points it binds (round 5; until then a gloo branch in this file served them).
"""
import numpy as np


def shard_bounds(batch: int, world: int, rank: int) -> tuple:
    """Contiguous block partition [lo, hi) of `batch` items; the first `batch % world` ranks get one extra item."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_indices(batch: int, world: int, rank: int, partition: str = "block"):
    """The batch indices rank `rank` solves.  "block": the contiguous range of shard_bounds (the default: a rank's alphas are one slab of the
    gathered tensor).  "interleaved": b mod world == rank (ragged batches sorted by size, sweeps whose cost grows along the
    batch axis -- neighbouring problems cost about the same, so dealing them out evens the ranks' loads; the gathered tensor is then rank-major
    and solve_sharded puts the rows back in batch order).  Every rank holds ceil(batch / world) slots either way."""
    if partition == "block":
        lo, hi = shard_bounds(batch, world, rank)
        return np.arange(lo, hi)
    if partition == "interleaved":
        return np.arange(rank, batch, world)
    raise ValueError("partition must be 'block' or 'interleaved'")


def init_engine_comm(engine, dist):
    """Collective over the ranks of `dist` (an initialised torch.distributed, any backend): gives `engine` its RCCL communicator.
    Rank 0 creates the id, one object broadcast ships it."""
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    engine.comm_init(rank, world, box[0])
    return rank, world


def _has_comm(engine):
    try:
        engine.comm_world()
        return True
    except Exception:
        return False


def solve_sharded(problems: list, engine, dist=None, partition: str = "block", **opt_kw):
    """Every rank solves its shard of `problems` on its own GPU -- a contiguous block of the batch axis, or with partition="interleaved" every
    world-th problem (shard_indices) --; alpha (padded to the longest track), curvature errors and status words are all-gathered so that every
    rank returns the full batch, in batch order.

    problems: dicts {reftrack [n,4], normvec [n,2] or None (then for all: normals and scalings are derived on the device),
    scaling [n] or None, kappa_bound, w_veh}.
    dist: torch.distributed (initialised) or None for single-process.  The gather runs through the engine's RCCL communicator
    (`init_engine_comm`; required for more than one rank).
    Returns (alphas list, curv [B], status [B]).
    """
    bsz = len(problems)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    use_rccl = _has_comm(engine)
    if world > 1 and not use_rccl:
        raise ValueError("solve_sharded: %d ranks but the engine has no communicator -- call parallel.init_engine_comm(engine, dist) first" % world)
    if use_rccl and engine.comm_world() != (rank, world):
        raise ValueError("solve_sharded: the engine's communicator is rank %d of %d, the process group says %d of %d"
                         % (engine.comm_world() + (rank, world)))
    mine = shard_indices(bsz, world, rank, partition)
    nmax = max(int(np.asarray(p["reftrack"]).shape[0]) for p in problems)
    per = max(len(shard_indices(bsz, world, r, partition)) for r in range(world))
    with_nv = [p.get("normvec") is not None for p in problems]
    if any(with_nv) and not all(with_nv):
        raise ValueError("solve_sharded: normvec must be given for all problems or for none")
    # ---- this rank's shard, padded to [per][nmax] (slots beyond the shard: n = 0, skipped by the kernels) -------------------------
    ref = np.zeros((per, nmax, 4))
    nv = np.zeros((per, nmax, 2))
    sc = np.ones((per, nmax))
    ns = np.zeros(per, dtype=np.int32)
    kb = np.ones(per)
    wv = np.zeros(per)
    for k, p in enumerate(problems[b] for b in mine):
        r = np.asarray(p["reftrack"], dtype=np.float64)
        n = r.shape[0]
        ref[k, :n] = r
        if with_nv[0]:
            nv[k, :n] = p["normvec"]
        if p.get("scaling") is not None:
            sc[k, :n] = p["scaling"]
        ns[k], kb[k], wv[k] = n, float(p["kappa_bound"]), float(p["w_veh"])
    # one send buffer per rank, [alpha per x nmax | curv per | status per (int32 in the first half of `per` doubles)]: the engine writes
    # its outputs straight into it, the collective reads it
    count = per * nmax + 2 * per
    bufs = []

    def dev(arr=None, nbytes=0):
        p = engine.alloc(arr.nbytes if arr is not None else nbytes)
        bufs.append(p)
        if arr is not None:
            engine.upload(p, arr)
        return p

    try:
        d_ref, d_sc, d_n, d_kb, d_wv = dev(ref), dev(sc), dev(ns), dev(kb), dev(wv)
        d_nv = dev(nv) if with_nv[0] else None
        d_send = dev(nbytes=8 * count)                      # (mcq_device_alloc zero-fills: empty shards gather zeros)
        d_alpha, d_curv, d_status = d_send, d_send + 8 * per * nmax, d_send + 8 * (per * nmax + per)
        if len(mine) > 0:
            engine.solve_device_ragged_params(per, nmax, d_n, d_ref, d_nv, d_sc, 0.0, 0.0, d_kb, d_wv, d_alpha, d_curv, d_status,
                                              **opt_kw)
        if world == 1 and not use_rccl:
            full = engine.download(d_send, (1, count), np.float64)
        else:
            d_recv = dev(nbytes=8 * count * world)
            engine.comm_allgather(d_send, d_recv, count, engine.DT_F64)       # the single collective of the job: on the handle's COMM stream,
            engine.comm_wait(0)                                               # behind the solve; the receive buffer is complete after this
            full = engine.download(d_recv, (world, count), np.float64)
    finally:
        for p in bufs:
            engine.free(p)
    out_a, out_c, out_s = [None] * bsz, np.zeros(bsz), np.zeros(bsz, dtype=np.int32)
    for r in range(world):
        al = full[r, :per * nmax].reshape(per, nmax)
        cu = full[r, per * nmax:per * nmax + per]
        st = np.ascontiguousarray(full[r, per * nmax + per:]).view(np.int32)[:per]
        for k, b in enumerate(shard_indices(bsz, world, r, partition)):
            n = int(np.asarray(problems[b]["reftrack"]).shape[0])
            out_a[b] = al[k, :n].copy()
            out_c[b] = cu[k]
            out_s[b] = int(st[k])
    return out_a, out_c, out_s

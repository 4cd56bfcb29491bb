"""Pair-level data parallelism: the per-pair loops of the reference
(1_FGR...py:134-147, 2_MGICP...py:187-214, ALL_FUNCTIONS.py:349-392) have no cross-pair state, so pair i of the
circuit goes to exactly one rank; the only exchange is ONE all-gather of a fixed-size record per pair
(RCCL over xGMI on GPUs, gloo in the CPU tests).  No other collective exists on this path.
"""
from __future__ import annotations

import numpy as np

RECORD_DOUBLES = 22      # 16 pose + fitness + rmse + n_corr + iterations + converged + pair index


def block_bounds(costs, world_size: int) -> list:
    """Cut points b[0] = 0 <= b[1] <= ... <= b[W] = P of the contiguous partition of P pairs into W blocks whose LARGEST block cost
    is minimal (the time of a sharded run is its slowest rank's).  Bisection on the block capacity with a greedy feasibility test
    (exact for integer costs); among the optimal cuts the greedy one is evened out from the back so that no rank is left empty
    while another holds several pairs.  Deterministic: every rank computes the same cuts from the same costs."""
    c = [int(v) for v in costs]
    P, W = len(c), int(world_size)
    if W <= 1 or P == 0:
        return [0] + [P] * max(W, 1)

    def cuts(cap):
        b, acc = [0], 0
        for i, v in enumerate(c):
            if acc + v > cap and acc > 0:
                b.append(i); acc = 0
            acc += v
        return b + [P]
    lo, hi = max(c), sum(c)
    while lo < hi:
        mid = (lo + hi) // 2
        if len(cuts(mid)) - 1 <= W:
            hi = mid
        else:
            lo = mid + 1
    b = cuts(lo)
    b = b[:-1] + [P] * (W + 1 - (len(b) - 1))                   # fewer than W blocks: the rest are empty for now
    # An empty block (the greedy cut leaves them at the back) takes a pair from the nearest block before it that holds several: every cut
    # between the two moves one pair to the left, the one-pair blocks in between pass their pair on (one pair costs at most max(c) <= lo:
    # the optimum stays).  [1, 1, 1, 3] on 3 ranks: greedy [0, 3, 4, 4] -> [0, 2, 3, 4].
    while True:
        empty = next((r for r in range(W - 1, 0, -1) if b[r] == b[r + 1]), None)
        if empty is None:
            break
        donor = next((j for j in range(empty - 1, -1, -1) if b[j + 1] - b[j] >= 2), None)
        if donor is None:
            break                                               # fewer pairs than ranks: some blocks stay empty
        for k in range(donor + 1, empty + 1):
            b[k] -= 1
    return b


def partition(n_pairs: int, world_size: int, rank: int, costs=None) -> range:
    """Contiguous block of pair indices of `rank`: neighbouring pairs share a cloud, so a rank uploads (block + 1) clouds; the
    loop-closure pair (last index) falls to the last rank.  Without `costs` the blocks are [r*P/W, (r+1)*P/W); with `costs`
    (one number per pair, e.g. the points of its two clouds: NCLT scans hold 6.9k-31k points and an index split leaves the slowest
    of 8 ranks 25 % over the mean) the blocks are balanced by cost (`block_bounds`)."""
    if costs is None:
        return range((rank * n_pairs) // world_size, ((rank + 1) * n_pairs) // world_size)
    if len(costs) != n_pairs:
        raise ValueError("partition: one cost per pair")
    b = block_bounds(costs, world_size)
    return range(b[rank], b[rank + 1])


def circuit_costs(point_counts) -> list:
    """Cost of pair i of a closed circuit = points of its two clouds (`circuit_pair`)."""
    n = len(point_counts)
    return [int(point_counts[circuit_pair(i, n)[0]]) + int(point_counts[circuit_pair(i, n)[1]]) for i in range(n)]


def circuit_pair(i: int, n_clouds: int):
    """(source cloud, target cloud) of pair i: cloud i+1 onto cloud i, the last one closing the loop
    (cloud 0 onto cloud n-1)  -- 2_MGICP...py:187-214."""
    return ((i + 1) if i < n_clouds - 1 else 0, i)


def pack_record(pair_index: int, result) -> np.ndarray:
    rec = np.zeros(RECORD_DOUBLES, dtype=np.float64)
    rec[:16] = np.asarray(result.transformation, dtype=np.float64).reshape(16)
    rec[16] = result.fitness; rec[17] = result.inlier_rmse
    rec[18] = float(len(result.correspondence_set)) if getattr(result, "_corr", None) is None else float(result._corr.shape[0])
    rec[19] = float(getattr(result, "iterations", 0)); rec[20] = float(getattr(result, "converged", False))
    rec[21] = float(pair_index)
    return rec


def unpack_record(rec):
    rec = np.asarray(rec, dtype=np.float64)
    return dict(pair=int(rec[21]), transformation=rec[:16].reshape(4, 4).copy(), fitness=float(rec[16]),
                inlier_rmse=float(rec[17]), n_corr=int(rec[18]), iterations=int(rec[19]), converged=bool(rec[20]))


def gather_records(local_records: np.ndarray, n_pairs: int, device=None):
    """All-gather the per-pair records of every rank; returns an (n_pairs, RECORD_DOUBLES) array ordered by pair
    index on EVERY rank.  With no initialised process group this is the identity (single process)."""
    import torch
    import torch.distributed as dist
    local_records = np.asarray(local_records, dtype=np.float64).reshape(-1, RECORD_DOUBLES)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = local_records
    else:
        world = dist.get_world_size()
        per = (n_pairs + world - 1) // world + 1                    # fixed-size slab per rank (ragged blocks padded)
        slab = torch.full((per, RECORD_DOUBLES), -1.0, dtype=torch.float64)
        slab[: local_records.shape[0]] = torch.from_numpy(local_records)
        if device is not None:
            slab = slab.to(device)
        out_t = torch.empty((world * per, RECORD_DOUBLES), dtype=torch.float64, device=slab.device)
        dist.all_gather_into_tensor(out_t, slab)
        out = out_t.cpu().numpy()
        out = out[out[:, 21] >= 0]
    order = np.argsort(out[:, 21], kind="stable")
    out = out[order]
    if out.shape[0] != n_pairs or not np.array_equal(out[:, 21].astype(np.int64), np.arange(n_pairs)):
        raise RuntimeError(f"gathered {out.shape[0]} pair records, expected {n_pairs} distinct pairs")
    return out

"""Multi-GPU layer: candidate views shard embarrassingly across the ranks of one node (one process per GPU,
`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" on CPU for the tests).

The reference has no distributed code; this is new work specified by section 8(e):
  * Gaussian parameters and H_inv are REPLICATED on every rank (28 MB at 500k Gaussians);
  * candidate views are partitioned contiguously; every rank scores its slice with FisherScorer;
  * ONE all-gather of the per-view scalar scores (V/world floats per rank: latency-bound, a single direct exchange);
  * H_train = keyframes sharded across ranks + ONE all-reduce(SUM) of the [P, C] fp32 accumulator.
No collective touches the per-pixel / per-splat data path.
"""
from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous, balanced partition of range(n): the first n % world ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _host_collectives(t: torch.Tensor, group=None) -> bool:
    """gloo (the CPU rehearsal backend) is driven with host tensors; nccl (RCCL) takes the device tensors as they are."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def gather_scores(local_scores: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather of per-view scores from contiguous shards of possibly unequal length -> [n_total] on every rank."""
    rank, world = _world(group)
    if world == 1:
        return local_scores
    if _host_collectives(local_scores, group):
        return gather_scores(local_scores.cpu(), n_total, group).to(local_scores.device)
    per = (n_total + world - 1) // world
    buf = torch.zeros((per,), dtype=local_scores.dtype, device=local_scores.device)
    buf[: local_scores.numel()] = local_scores
    out = torch.empty((world * per,), dtype=local_scores.dtype, device=local_scores.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts)


def sharded_scores(score_fn: Callable[[torch.Tensor], torch.Tensor], w2c_all: torch.Tensor, group=None) -> torch.Tensor:
    """score_fn(w2c[v0:v1]) -> scores of that slice (device tensor).  Returns all V scores on every rank."""
    rank, world = _world(group)
    V = int(w2c_all.shape[0])
    lo, hi = shard_bounds(V, rank, world)
    if hi > lo:
        local = score_fn(w2c_all[lo:hi])
    else:
        local = torch.zeros((0,), dtype=torch.float32, device=w2c_all.device)
    return gather_scores(local, V, group)


def sharded_h_train(accumulate_fn: Callable[[torch.Tensor, torch.Tensor], None], kf_w2c: torch.Tensor,
                    H_train: torch.Tensor, group=None) -> torch.Tensor:
    """accumulate_fn(w2c_slice, H_train) adds the slice's cur_H into H_train ([P,C], zero-filled by the caller).
    Keyframes are sharded, then one all-reduce(SUM)."""
    rank, world = _world(group)
    K = int(kf_w2c.shape[0])
    lo, hi = shard_bounds(K, rank, world)
    if hi > lo:
        accumulate_fn(kf_w2c[lo:hi], H_train)
    if world > 1:
        _all_reduce(H_train, dist.ReduceOp.SUM, group)
    return H_train


def _all_reduce(t: torch.Tensor, op, group=None):
    if _host_collectives(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def sharded_point_score_max(scorer, w2c_all: torch.Tensor, H_inv: torch.Tensor, group=None, chunk: int = 16) -> torch.Tensor:
    """max over the candidate views of every Gaussian's score sum_c cur_H[v, i, c] * H_inv[i, c] -- the `max_points_score`
    the reference keeps while it scans the candidates (gaussian.py:1284-1303).  Views sharded, ONE all-reduce(MAX) on [P]."""
    rank, world = _world(group)
    V = int(w2c_all.shape[0])
    lo, hi = shard_bounds(V, rank, world)
    best = torch.zeros((scorer.P,), dtype=torch.float32, device=w2c_all.device)       # the reference starts from zeros
    for v0 in range(lo, hi, chunk):
        w = w2c_all[v0:min(hi, v0 + chunk)]
        cur = torch.zeros((int(w.shape[0]), scorer.P, scorer.columns), dtype=torch.float32, device=w2c_all.device)
        scorer.run(w, out_H=cur, out_H_per_view=True)
        best = torch.maximum(best, (cur * H_inv.unsqueeze(0)).sum(dim=2).max(dim=0).values)
    if world > 1:
        _all_reduce(best, dist.ReduceOp.MAX, group)
    return best


def pose_eval_sharded(scorer, kf_w2c: torch.Tensor, w2c_all: torch.Tensor, reg: float = 0.1, group=None):
    """GaussianSLAM.pose_eval (gaussian.py:1354-1375) over the ranks of a node.  `scorer` is a FisherScorer holding
    this rank's replica of the map."""
    H_train = torch.zeros((scorer.P, scorer.columns), dtype=torch.float32, device=scorer.dev)
    sharded_h_train(lambda w, H: scorer.run(w, out_H=H), kf_w2c, H_train, group)
    H_inv = torch.reciprocal(H_train + reg)
    scores = sharded_scores(lambda w: scorer.run(w, H_inv=H_inv)["scores"], w2c_all, group)
    return scores, H_train

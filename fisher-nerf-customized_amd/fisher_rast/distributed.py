"""Multi-GPU layer: candidate views shard embarrassingly across the ranks of one node (one process per GPU,
`torch.distributed`, backend "nccl" == RCCL over xGMI; "gloo" on CPU for the tests).

The reference has no distributed code; this is new work specified by section 8(e):
  * Gaussian parameters and H_inv are REPLICATED on every rank (28 MB + 8 MB at 500k Gaussians): `replicate_map` broadcasts
    them from the one process that holds `slam.params` (tester_gaussians_navigation.py:1618-1649), once per planning round;
  * candidate views are partitioned contiguously; every rank scores its slice with FisherScorer;
  * ONE all-gather of the per-view scalar scores (V/world floats per rank: latency-bound, a single direct exchange);
  * H_train = keyframes sharded across ranks + ONE all-reduce(SUM) of the [P, C] fp32 accumulator.
No collective touches the per-pixel / per-splat data path.
"""
import os
from typing import Callable, Optional

import torch
import torch.distributed as dist

# FR_FORCE_COLLECTIVES=1: issue the collectives in a process group of ONE rank too (they are no-ops for the data, but the
# transport -- RCCL on a GPU box -- is loaded, initialised and run on the device tensors): the one-GPU rehearsal of the
# multi-GPU path (tests/test_gpu_bench_multirank.py)
FORCE_COLLECTIVES = os.environ.get("FR_FORCE_COLLECTIVES") == "1"


def _collective(world: int) -> bool:
    return world > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


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


def _broadcast(t: torch.Tensor, src: int, group=None):
    if _host_collectives(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def replicate_map(params: Optional[dict], H_inv: Optional[torch.Tensor] = None, src: int = 0, group=None, device=None):
    """SURVEY 8(e) "replicate ... (broadcast once per planning round)": rank `src` holds the map (a dict name -> tensor, e.g.
    `slam.params` or the activated render variables) and optionally H_inv; every other rank may pass None (or stale tensors of
    any content).  One small object broadcast carries names / shapes / dtypes, then ONE `dist.broadcast` per tensor
    (28 MB + 8 MB at 500k Gaussians).  Returns `(params, H_inv)` on every rank: on `src` the caller's own tensors, elsewhere
    tensors on `device` (reused when the caller passed tensors of the right shape / dtype / device)."""
    rank, world = _world(group)
    if not _collective(world):
        return params, H_inv
    if rank == src:
        items = [(k, v) for k, v in params.items() if torch.is_tensor(v)]
        meta = [[(k, tuple(v.shape), v.dtype) for k, v in items], None if H_inv is None else (tuple(H_inv.shape), H_inv.dtype)]
    else:
        meta = [None, None]
    dist.broadcast_object_list(meta, src=src, group=group)
    if device is None:
        device = next(iter(params.values())).device if params else torch.device("cpu")

    def _slot(old, shape, dtype):
        if rank == src:
            return old.detach().contiguous()
        if torch.is_tensor(old) and tuple(old.shape) == shape and old.dtype == dtype and old.device == torch.device(device) and old.is_contiguous():
            return old.detach()
        return torch.empty(shape, dtype=dtype, device=device)

    out = {} if rank != src else dict(params)
    for name, shape, dtype in meta[0]:
        t = _slot(params.get(name) if params else None, shape, dtype)
        _broadcast(t, src, group)
        if rank != src:
            out[name] = t
    if meta[1] is not None:
        h = _slot(H_inv, *meta[1])
        _broadcast(h, src, group)
        H_inv = H_inv if rank == src else h
    return out, H_inv


def map_fingerprint(tensors) -> torch.Tensor:
    """One fp64 number per tensor (its sum and its sum of squares folded): equal on two ranks iff the replicas agree (up to the
    astronomically unlikely collision).  NaNs are mapped to a fixed value so that two equal maps with NaNs still compare equal."""
    vals = []
    for t in tensors:
        d = torch.nan_to_num(t.detach().double().reshape(-1), nan=12345.0, posinf=1e300, neginf=-1e300)
        w = torch.arange(1, d.numel() + 1, dtype=torch.float64, device=d.device) * 1e-6
        vals += [d.sum(), (d * w).sum()]
    return torch.stack(vals) if vals else torch.zeros((0,), dtype=torch.float64)


def assert_replicated(tensors, group=None, what: str = "Gaussian map"):
    """Raise on EVERY rank when the ranks' replicas differ (two all-reduces of a few doubles: MIN and MAX of the fingerprint)."""
    rank, world = _world(group)
    if not _collective(world):
        return
    fp = map_fingerprint(tensors)
    lo, hi = fp.clone(), fp.clone()
    _all_reduce(lo, dist.ReduceOp.MIN, group)
    _all_reduce(hi, dist.ReduceOp.MAX, group)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"{what}: the ranks hold different replicas (fingerprints differ) -- call "
                           f"fisher_rast.distributed.replicate_map() before sharding the views")


class ScoreGather:
    """The per-step all-gather of the per-view scores with every buffer allocated ONCE (a planning loop calls it every step):
    contiguous shards of possibly unequal length -> [n_total] on every rank.  V % world == 0: the local scores go straight into
    `all_gather_into_tensor` and its output IS the result (no padding, no copy); otherwise the shards are padded to the longest
    one and the result is one `index_select` of the padded gather.  The returned tensor is overwritten by the next call."""

    def __init__(self, n_total: int, device, dtype=torch.float32, group=None):
        self.group, self.n_total = group, int(n_total)
        self.rank, self.world = _world(group)
        self.active = _collective(self.world)
        self.host = self.active and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo"
        dev = torch.device("cpu") if self.host else torch.device(device)
        self.device = torch.device(device)
        self.lo, self.hi = shard_bounds(self.n_total, self.rank, self.world)
        self.even = self.n_total % max(self.world, 1) == 0
        per = (self.n_total + self.world - 1) // max(self.world, 1)
        self.out = torch.empty((self.world * per,), dtype=dtype, device=dev)
        if not self.even:
            self.pad = torch.zeros((per,), dtype=dtype, device=dev)
            idx = []
            for r in range(self.world):
                lo, hi = shard_bounds(self.n_total, r, self.world)
                idx += list(range(r * per, r * per + (hi - lo)))
            self.index = torch.tensor(idx, dtype=torch.int64, device=dev)
            self.result = torch.empty((self.n_total,), dtype=dtype, device=dev)
        self.dev_result = torch.empty((self.n_total,), dtype=dtype, device=self.device) if self.host else None

    def __call__(self, local_scores: torch.Tensor) -> torch.Tensor:
        if not self.active:
            return local_scores
        assert local_scores.numel() == self.hi - self.lo
        src = local_scores
        if self.host:                         # gloo rehearsal: collectives on host tensors
            src = local_scores.cpu()
        if self.even:
            dist.all_gather_into_tensor(self.out, src.contiguous(), group=self.group)
            res = self.out
        else:
            self.pad[: src.numel()].copy_(src)
            dist.all_gather_into_tensor(self.out, self.pad, group=self.group)
            torch.index_select(self.out, 0, self.index, out=self.result)
            res = self.result
        if self.host:
            self.dev_result.copy_(res)
            return self.dev_result
        return res


def gather_scores(local_scores: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """One-off form of ScoreGather (allocates its buffers per call; a loop should hold a ScoreGather)."""
    rank, world = _world(group)
    if not _collective(world):
        return local_scores
    return ScoreGather(n_total, local_scores.device, local_scores.dtype, group)(local_scores).clone()


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
    if _collective(world):
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
    if _collective(world):
        _all_reduce(best, dist.ReduceOp.MAX, group)
    return best


def pose_eval_sharded(scorer, kf_w2c: torch.Tensor, w2c_all: torch.Tensor, reg: float = 0.1, group=None, check_replicas: bool = True):
    """GaussianSLAM.pose_eval (gaussian.py:1354-1375) over the ranks of a node.  `scorer` is a FisherScorer holding
    this rank's replica of the map (built from what `replicate_map` returned).  `check_replicas`: compare a fingerprint of the
    replicas and of the poses across the ranks first (two tiny all-reduces) and raise instead of returning scores of
    different maps."""
    if check_replicas:
        assert_replicated([scorer.means3D, scorer.colors, scorer.rotations, scorer.opacities, scorer.scales, kf_w2c, w2c_all], group)
    H_train = torch.zeros((scorer.P, scorer.columns), dtype=torch.float32, device=scorer.dev)
    sharded_h_train(lambda w, H: scorer.run(w, out_H=H), kf_w2c, H_train, group)
    H_inv = torch.reciprocal(H_train + reg)
    scores = sharded_scores(lambda w: scorer.run(w, H_inv=H_inv)["scores"], w2c_all, group)
    return scores, H_train

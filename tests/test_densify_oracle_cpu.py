"""CPU-only: oracle/densify_stats.py against the reference's own torch-op chains evaluated with CPU torch
(slam_external.py:196-200, 419-433, 354, 394-396; gaussian.py:289-291), on values away from the thresholds (torch's exp /
sigmoid round differently in the last bit; the restatement uses the oracle's fixed-sequence exp)."""
import numpy as np
import torch

from oracle import densify_stats as ods


def test_restatement_follows_the_torch_chains():
    rng = np.random.default_rng(0)
    P = 5000
    radius = np.where(rng.uniform(size=P) < 0.5, rng.integers(1, 50, P), 0).astype(np.int32)
    grad = rng.normal(0, 1e-3, (P, 3)).astype(np.float32)
    mr = rng.uniform(0, 30, P).astype(np.float32)
    acc = rng.uniform(0, 1e-2, P).astype(np.float32)
    den = rng.integers(0, 5, P).astype(np.float32)
    ls = rng.normal(np.log(0.05), 0.5, (P, 3)).astype(np.float32)
    lo = rng.normal(0, 3, (P, 1)).astype(np.float32)
    # reference chains, torch on CPU
    variables = dict(max_2D_radius=torch.from_numpy(mr.copy()), means2D_gradient_accum=torch.from_numpy(acc.copy()), denom=torch.from_numpy(den.copy()))
    r = torch.from_numpy(radius)
    seen = r > 0
    variables['max_2D_radius'][seen] = torch.max(r[seen].float(), variables['max_2D_radius'][seen])
    g = torch.from_numpy(grad)
    variables['means2D_gradient_accum'][seen] += torch.norm(g[seen, :2], dim=-1)
    variables['denom'][seen] += 1
    seen_o, mr_o = ods.seen_and_radius(radius, mr)
    acc_o, den_o = ods.accumulate_mean2d_gradient(grad, seen_o, acc, den)
    assert np.array_equal(seen_o, seen.numpy()) and np.array_equal(mr_o, variables['max_2D_radius'].numpy())
    assert np.allclose(acc_o, variables['means2D_gradient_accum'].numpy(), rtol=2e-7, atol=0) and np.array_equal(den_o, variables['denom'].numpy())
    grads = variables['means2D_gradient_accum'] / variables['denom']
    grads[grads.isnan()] = 0.0
    scale_max = torch.max(torch.exp(torch.from_numpy(ls)), dim=1).values
    away = (np.abs(scale_max.numpy() - 0.05) > 1e-6) & (np.abs(scale_max.numpy() - 0.1) > 1e-6) & (np.abs(grads.numpy() - 0.002) > 1e-8)
    to_clone = torch.logical_and(grads >= 0.002, scale_max <= 0.05).numpy()
    to_split = (scale_max > 0.05).numpy()
    c_o, s_o = ods.densify_masks(acc_o, den_o, ls, 0.002)
    assert np.array_equal(c_o[away], to_clone[away]) and np.array_equal(s_o[away], to_split[away]) and 0 < c_o.sum() < P
    op = torch.sigmoid(torch.from_numpy(lo)).squeeze().numpy()
    away_o = np.abs(op - 0.005) > 1e-7
    rm = (torch.sigmoid(torch.from_numpy(lo)) < 0.005).squeeze().numpy() | (scale_max > 0.1).numpy()
    rm_o = ods.prune_mask(lo, ls, 0.005, 0.1)
    assert np.array_equal(rm_o[away & away_o], rm[away & away_o]) and 0 < rm_o.sum() < P

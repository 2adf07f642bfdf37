"""Checkpoint format of the reference (models/SLAM/utils/common_utils.py:28-59; read back by
tester_gaussians_navigation.py:2745-2760): `params{t}.npz` holds every entry of the `params` dict as a CPU numpy array
(plus whatever extra keyword arrays the caller added, e.g. "Uncertainty", "occ_map"), `params.npz` the final map.
Saved reference maps load straight into `GaussianSLAM` / the benchmark, and maps saved here load in the reference."""
import os

import numpy as np
import torch

NON_PARAM_KEYS = ("Uncertainty", "occ_map")     # tester 2750: extras stored beside the parameters


def _as_numpy(value):
    """Tensors go to host memory as contiguous arrays; anything else is stored as it is (np.savez wraps it)."""
    return value.detach().cpu().contiguous().numpy() if isinstance(value, torch.Tensor) else value


def _write_checkpoint(params, output_dir, stem, extras=None):
    """One `.npz` with an array per parameter name, plus the caller's extra arrays under their keyword names."""
    os.makedirs(output_dir, exist_ok=True)
    arrays = {name: _as_numpy(v) for name, v in params.items()}
    arrays.update({name: _as_numpy(v) for name, v in (extras or {}).items()})
    path = os.path.join(output_dir, stem + ".npz")
    np.savez(path, **arrays)
    return path


def params2cpu(params):
    return {name: _as_numpy(v) for name, v in params.items()}


def save_params(output_params, output_dir):
    """The final map: `<output_dir>/params.npz` (common_utils.py:35-43)."""
    return _write_checkpoint(output_params, output_dir, "params")


def save_params_ckpt(output_params, output_dir, time_idx, **extra_args):
    """A checkpoint of frame `time_idx`: `<output_dir>/params<time_idx>.npz` (common_utils.py:45-59)."""
    return _write_checkpoint(output_params, output_dir, f"params{time_idx}", extra_args)


def checkpoint_time_idx(weight_file):
    """`.../params123.npz` -> 123 (tester 2746); `.../params.npz` -> None."""
    stem = os.path.basename(weight_file).split('.')[0][6:]
    return int(stem) if stem else None


def load_params_ckpt(weight_file, device="cuda", requires_grad=False):
    """Returns (params, extras): params as float32 tensors on `device` (tester 2749-2750), the non-parameter arrays
    ("Uncertainty", "occ_map") as numpy."""
    params_np = dict(np.load(weight_file, allow_pickle=True))
    params = {k: torch.tensor(params_np[k]).to(device).float().requires_grad_(requires_grad)
              for k in params_np.keys() if k not in NON_PARAM_KEYS}
    extras = {k: params_np[k] for k in params_np.keys() if k in NON_PARAM_KEYS}
    return params, extras

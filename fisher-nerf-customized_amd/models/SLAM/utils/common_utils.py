"""Checkpoint format of the reference (models/SLAM/utils/common_utils.py:28-59; read back by
tester_gaussians_navigation.py:2745-2760): `params{t}.npz` holds every entry of the `params` dict as a CPU numpy array
(plus whatever extra keyword arrays the caller added, e.g. "Uncertainty", "occ_map"), `params.npz` the final map.
Saved reference maps load straight into `GaussianSLAM` / the benchmark, and maps saved here load in the reference."""
import os

import numpy as np
import torch

NON_PARAM_KEYS = ("Uncertainty", "occ_map")     # tester 2750: extras stored beside the parameters


def params2cpu(params):
    res = {}
    for k, v in params.items():
        if isinstance(v, torch.Tensor):
            res[k] = v.detach().cpu().contiguous().numpy()
        else:
            res[k] = v
    return res


def save_params(output_params, output_dir):
    to_save = params2cpu(output_params)
    os.makedirs(output_dir, exist_ok=True)
    save_path = os.path.join(output_dir, "params.npz")
    np.savez(save_path, **to_save)
    return save_path


def save_params_ckpt(output_params, output_dir, time_idx, **extra_args):
    to_save = params2cpu(output_params)
    os.makedirs(output_dir, exist_ok=True)
    save_path = os.path.join(output_dir, "params" + str(time_idx) + ".npz")
    for k, v in extra_args.items():
        to_save[k] = v.cpu().numpy() if isinstance(v, torch.Tensor) else v
    np.savez(save_path, **to_save)
    return save_path


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

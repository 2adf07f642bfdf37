"""ctypes binding of libfisher_rast.so (C ABI declared in include/fisher_rast.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  There is no
CPU fallback: if the shared object is missing, loading raises and every op of the package fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FISHER_RAST_SO lets an A/B benchmark load another build of the same ABI; the default is the in-tree library
SO_PATH = os.environ.get("FISHER_RAST_SO", os.path.join(_HERE, "libfisher_rast.so"))

FR_OK, FR_EINVAL, FR_ELAUNCH, FR_ENOSPACE = 0, 1, 2, 3

_f32p = ctypes.c_void_p  # device pointers travel as raw addresses


class RasterCfg(ctypes.Structure):
    _fields_ = [
        ("P", ctypes.c_int32),
        ("image_height", ctypes.c_int32),
        ("image_width", ctypes.c_int32),
        ("tanfovx", ctypes.c_float),
        ("tanfovy", ctypes.c_float),
        ("scale_modifier", ctypes.c_float),
        ("sh_degree", ctypes.c_int32),
        ("sh_coeffs", ctypes.c_int32),
        ("prefiltered", ctypes.c_int32),
        ("bg", _f32p),
        ("viewmatrix", _f32p),
        ("projmatrix", _f32p),
        ("campos", _f32p),
    ]


class Gaussians(ctypes.Structure):
    _fields_ = [
        ("means3D", _f32p),
        ("colors_precomp", _f32p),
        ("shs", _f32p),
        ("opacities", _f32p),
        ("scales", _f32p),
        ("rotations", _f32p),
        ("cov3D_precomp", _f32p),
    ]


class FisherCfg(ctypes.Structure):
    _fields_ = [
        ("n_views", ctypes.c_int32),
        ("columns", ctypes.c_int32),
        ("dL_dpix", ctypes.c_float),
        ("w2c", _f32p),
        ("H_inv", _f32p),
        ("H_inv_view_stride", ctypes.c_int64),
        ("out_scores", _f32p),
        ("out_H", _f32p),
        ("out_H_view_stride", ctypes.c_int64),
        ("out_vis_count", ctypes.c_void_p),
        ("out_num_rendered", ctypes.c_void_p),
        ("dL_dpix_image", _f32p),
        ("dL_image_view_stride", ctypes.c_int64),
        ("tile_capacity", ctypes.c_int32),
        ("poses_are_c2w", ctypes.c_int32),
        ("reuse_static", ctypes.c_int32),
        ("order", ctypes.c_void_p),
    ]


class OccCfg(ctypes.Structure):
    """fr_occ_cfg (include/fisher_occ.h)"""
    _fields_ = [
        ("grid_w", ctypes.c_int32),
        ("grid_h", ctypes.c_int32),
        ("cell_size", ctypes.c_float),
        ("center_x", ctypes.c_float),
        ("center_z", ctypes.c_float),
        ("height_lower", ctypes.c_float),
        ("height_upper", ctypes.c_float),
        ("far_distance", ctypes.c_float),
    ]


# every symbol include/fisher_rast.h and include/fisher_occ.h declare
EXPORTS = (
    "fr_version", "fr_last_error", "fr_build_id", "fr_init", "fr_fisher_workspace_layout", "fr_workspace_bytes", "fr_workspace_layout", "fr_mark_visible",
    "fr_forward", "fr_backward", "fr_backward_scratch_bytes", "fr_backward_ws", "fr_forward_pair", "fr_forward_features", "fr_backward_pair", "fr_backward_pair_scratch_bytes", "fr_backward_pair_ws", "fr_fisher_workspace_bytes", "fr_fisher_views",
    "fr_densify_stats", "fr_densify_masks", "fr_prune_mask", "fr_knn_workspace_bytes", "fr_knn_dist2", "fr_spatial_order_workspace_bytes", "fr_spatial_order", "fr_profile_enable", "fr_profile_fetch",
    "fr_occ_workspace_bytes", "fr_occ_update", "fr_occ_freespace", "fr_occ_frontiers", "fr_occ_erode", "fr_occ_cells_of",
    "fr_occ_ring_candidates", "fr_occ_free_candidates",
)

_lib = None

_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
_INCLUDE = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include")
SOURCES = [os.path.join(_CSRC, n) for n in ("fisher_rast.hip", "fisher_occ.hip", "fr_math.h", "fr_internal.h")] + \
          [os.path.join(_INCLUDE, n) for n in ("fisher_rast.h", "fisher_occ.h")]


def source_hash():
    """sha256 (first 16 hex digits) over the kernel and header sources, or None when they are not beside the package."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES:
        if not os.path.exists(f):
            return None
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class FisherRastError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FisherRastError(
            f"{SO_PATH} is missing: the HIP extension has not been built "
            "(run `python -c \"import __graft_entry__ as g; g.build()\"` at the repo root). "
            "There is no CPU fallback for this path.")
    # PyTorch-ROCm ships its own HIP runtime; it has to be the one already in the process when this library's
    # libamdhip64 dependency is resolved, or the two would each hold their own (device-less) state.
    import torch  # noqa: F401
    lib = ctypes.CDLL(SO_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise FisherRastError(f"{SO_PATH} does not export {name}")
    lib.fr_version.restype = ctypes.c_int
    lib.fr_last_error.restype = ctypes.c_char_p
    lib.fr_build_id.restype = ctypes.c_char_p
    lib.fr_init.restype = ctypes.c_int
    lib.fr_fisher_workspace_layout.restype = ctypes.c_int
    lib.fr_fisher_workspace_layout.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t)]
    # a stale or foreign binary must not pass for the sources beside it (FISHER_RAST_SO builds for A/B runs are exempt)
    want = source_hash()
    got = lib.fr_build_id().decode().split(":")[-1]
    if "FISHER_RAST_SO" not in os.environ and want is not None and got != want:
        raise FisherRastError(f"{SO_PATH} was built from other sources (build id {got}, sources {want}): "
                              "run `python -c \"import __graft_entry__ as g; g.build()\"`")
    lib.fr_workspace_bytes.restype = ctypes.c_int
    lib.fr_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                       ctypes.POINTER(ctypes.c_size_t)]
    lib.fr_workspace_layout.restype = ctypes.c_int
    lib.fr_workspace_layout.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                        ctypes.POINTER(ctypes.c_size_t)]
    lib.fr_mark_visible.restype = ctypes.c_int
    lib.fr_mark_visible.argtypes = [ctypes.c_int32, _f32p, _f32p, _f32p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_forward.restype = ctypes.c_int
    lib.fr_forward.argtypes = [ctypes.POINTER(RasterCfg), ctypes.POINTER(Gaussians),
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                               _f32p, _f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_backward.restype = ctypes.c_int
    lib.fr_backward.argtypes = [ctypes.POINTER(RasterCfg), ctypes.POINTER(Gaussians), ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                _f32p, ctypes.c_int32] + [_f32p] * 9 + [ctypes.c_void_p]
    lib.fr_backward_scratch_bytes.restype = ctypes.c_size_t
    lib.fr_backward_scratch_bytes.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_int64]
    lib.fr_backward_ws.restype = ctypes.c_int
    lib.fr_backward_ws.argtypes = [ctypes.POINTER(RasterCfg), ctypes.POINTER(Gaussians), ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   _f32p, ctypes.c_int32] + [_f32p] * 9 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_forward_pair.restype = ctypes.c_int
    lib.fr_forward_pair.argtypes = [ctypes.POINTER(RasterCfg), ctypes.POINTER(Gaussians), _f32p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                    _f32p, _f32p, _f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_forward_features.restype = ctypes.c_int
    lib.fr_forward_features.argtypes = [ctypes.POINTER(RasterCfg), _f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        _f32p, ctypes.c_void_p]
    lib.fr_backward_pair.restype = ctypes.c_int
    lib.fr_backward_pair.argtypes = [ctypes.POINTER(RasterCfg), ctypes.POINTER(Gaussians), ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     _f32p, _f32p, _f32p] + [_f32p] * 10 + [ctypes.c_void_p]
    lib.fr_backward_pair_scratch_bytes.restype = ctypes.c_size_t
    lib.fr_backward_pair_scratch_bytes.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_int64]
    lib.fr_backward_pair_ws.restype = ctypes.c_int
    lib.fr_backward_pair_ws.argtypes = [ctypes.POINTER(RasterCfg), ctypes.POINTER(Gaussians), ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        _f32p, _f32p, _f32p] + [_f32p] * 10 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_fisher_workspace_bytes.restype = ctypes.c_size_t
    lib.fr_fisher_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int64, ctypes.c_int32]
    lib.fr_fisher_views.restype = ctypes.c_int
    lib.fr_fisher_views.argtypes = [ctypes.POINTER(RasterCfg), ctypes.POINTER(Gaussians),
                                    ctypes.POINTER(FisherCfg), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64,
                                    ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_densify_stats.restype = ctypes.c_int
    lib.fr_densify_stats.argtypes = [ctypes.c_int32, ctypes.c_void_p, _f32p, _f32p, _f32p, _f32p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_densify_masks.restype = ctypes.c_int
    lib.fr_densify_masks.argtypes = [ctypes.c_int32, _f32p, _f32p, _f32p, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_prune_mask.restype = ctypes.c_int
    lib.fr_prune_mask.argtypes = [ctypes.c_int32, _f32p, _f32p, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_knn_workspace_bytes.restype = ctypes.c_size_t
    lib.fr_knn_workspace_bytes.argtypes = [ctypes.c_int32]
    lib.fr_knn_dist2.restype = ctypes.c_int
    lib.fr_knn_dist2.argtypes = [ctypes.c_int32, _f32p, _f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_spatial_order_workspace_bytes.restype = ctypes.c_size_t
    lib.fr_spatial_order_workspace_bytes.argtypes = [ctypes.c_int32]
    lib.fr_spatial_order.restype = ctypes.c_int
    lib.fr_spatial_order.argtypes = [ctypes.c_int32, _f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_profile_enable.restype = ctypes.c_int
    lib.fr_profile_enable.argtypes = [ctypes.c_int]
    lib.fr_profile_fetch.restype = ctypes.c_int
    lib.fr_profile_fetch.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    occp = ctypes.POINTER(OccCfg)
    f4, f16 = ctypes.c_float * 4, ctypes.c_float * 16
    lib.fr_occ_workspace_bytes.restype = ctypes.c_size_t
    lib.fr_occ_workspace_bytes.argtypes = [occp]
    lib.fr_occ_update.restype = ctypes.c_int
    lib.fr_occ_update.argtypes = [occp, _f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(f4), ctypes.POINTER(f16),
                                  ctypes.POINTER(ctypes.c_float), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _f32p,
                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_occ_freespace.restype = ctypes.c_int
    lib.fr_occ_freespace.argtypes = [occp, _f32p, _f32p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_occ_frontiers.restype = ctypes.c_int
    lib.fr_occ_frontiers.argtypes = [occp, _f32p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_occ_erode.restype = ctypes.c_int
    lib.fr_occ_erode.argtypes = [occp, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    lib.fr_occ_ring_candidates.restype = ctypes.c_int
    lib.fr_occ_ring_candidates.argtypes = [occp, _f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                           ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int32, _f32p, ctypes.c_void_p, ctypes.c_void_p]
    lib.fr_occ_free_candidates.restype = ctypes.c_int
    lib.fr_occ_free_candidates.argtypes = [occp, ctypes.c_void_p, ctypes.c_float, ctypes.c_uint32, _f32p, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.fr_occ_cells_of.restype = ctypes.c_int
    lib.fr_occ_cells_of.argtypes = [occp, _f32p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    _lib = lib
    return lib


def check(rc, what):
    if rc != FR_OK:
        msg = load().fr_last_error().decode("utf-8", "replace")
        raise FisherRastError(f"{what} failed (code {rc}): {msg}")

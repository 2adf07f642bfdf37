"""oracle/ref.py -- TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.

ctypes front-end of oracle/fisher_oracle.c (the CPU restatement of the reference
rasteriser) plus NumPy restatements of the Python layers of the reference that
sit on the hot path:

  * setup_camera                   models/SLAM/utils/recon_helpers.py:4-32
  * _RasterizeGaussians fwd/bwd    thirdparty/diff-gaussian-rasterization-modified/
                                   diff_gaussian_rasterization/__init__.py:42-138
                                   + rasterize_points.cu:35-196
  * compute_Hessian (scene)        models/SLAM/gaussian.py:1503-1570
  * compute_Hessian (object, C=11) models/SLAM/gaussian_object.py:1940-2045
  * compute_H_train / pose_eval    models/SLAM/gaussian.py:1338-1375

The ARBITER (`arbiter=True` / `decisions=`): the same C statements compiled a second time with binary64 arithmetic
(oracle/_build/liboracle64.so, -DORC_DOUBLE) on the same binary32 inputs, every decision (culling, radii, tile lists and
their order, each pixel's contributor set) taken from the binary32 run.  |oracle - arbiter| is the rounding error of the
reference's binary32 chain itself; tests use it where that error, not the HIP path's, is what a tolerance has to cover.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package never does.
"""
import ctypes
import os
import subprocess
from typing import NamedTuple, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_SO64 = os.path.join(_HERE, "_build", "liboracle64.so")
_LIB = None
_LIB64 = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i32 = ctypes.POINTER(ctypes.c_int32)
c_u32 = ctypes.POINTER(ctypes.c_uint32)
c_u64 = ctypes.POINTER(ctypes.c_uint64)
c_u8 = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> str:
    """Compile oracle/fisher_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "fisher_oracle.c")
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in (_SO, _SO64)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        _LIB = ctypes.CDLL(_SO)
        _LIB.orc_expf.restype = ctypes.c_float
        _LIB.orc_expf.argtypes = [ctypes.c_float]
        _LIB.orc_bin.restype = ctypes.c_int64
        _LIB.orc_get_higher_msb.restype = ctypes.c_uint32
        _LIB.orc_get_higher_msb.argtypes = [ctypes.c_uint32]
    return _LIB


def lib64():
    """the arbiter build (binary64 arithmetic, decisions from the binary32 run)"""
    global _LIB64
    if _LIB64 is None:
        if not os.path.exists(_SO64):
            build()
        _LIB64 = ctypes.CDLL(_SO64)
    return _LIB64


c_d = ctypes.POINTER(ctypes.c_double)


def _f64(a, shape=None):
    """binary32 values, exactly, as a float64 array"""
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).astype(np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a, ty):
    if a is None:
        return ctypes.cast(None, ty)
    return a.ctypes.data_as(ty)


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def expf(x):
    L = lib()
    x = np.asarray(x, dtype=np.float32)
    return np.array([L.orc_expf(float(v)) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


class Camera(NamedTuple):
    """Mirror of GaussianRasterizationSettings (__init__.py:140-151) with NumPy members.
    viewmatrix / projmatrix are the flat float[16] the kernels read, i.e. the reference's
    transposed tensors flattened (column-major mathematical matrices)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    scale_modifier: float
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    sh_degree: int
    campos: np.ndarray
    prefiltered: bool


def setup_camera(w, h, k, w2c, near=0.01, far=100) -> Camera:
    """recon_helpers.py:4-32, float32 throughout like the torch original."""
    fx, fy, cx, cy = k[0][0], k[1][1], k[0][2], k[1][2]
    w2c = np.asarray(w2c, dtype=np.float32)
    cam_center = np.linalg.inv(w2c.astype(np.float64)).astype(np.float32)[:3, 3]
    w2c_t = w2c.T.copy()  # transposed => memory is column-major w2c
    opengl_proj = np.array([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                            [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                            [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                            [0.0, 0.0, 1.0, 0.0]], dtype=np.float32).T.copy()
    full_proj = (w2c_t @ opengl_proj).astype(np.float32)
    return Camera(
        image_height=int(h), image_width=int(w),
        tanfovx=w / (2 * fx), tanfovy=h / (2 * fy),
        bg=np.zeros(3, dtype=np.float32), scale_modifier=1.0,
        viewmatrix=w2c_t.reshape(16).copy(), projmatrix=full_proj.reshape(16).copy(),
        sh_degree=0, campos=cam_center.astype(np.float32), prefiltered=False)


def mark_visible(cam: Camera, means3D):
    means3D = _f32(means3D, (-1, 3))
    P = means3D.shape[0]
    present = np.zeros(P, dtype=np.uint8)
    lib().orc_mark_visible(ctypes.c_int(P), _p(means3D, c_f), _p(_f32(cam.viewmatrix), c_f),
                           _p(_f32(cam.projmatrix), c_f), _p(present, c_u8))
    return present.astype(bool)


def rasterize_forward(cam: Camera, means3D, opacities, colors_precomp=None, shs=None, scales=None,
                      rotations=None, cov3D_precomp=None, decisions: Optional[dict] = None):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-115) -> Rasterizer::forward
    (rasterizer_impl.cu:198-339).  Returns a dict holding the outputs and every
    intermediate buffer (geometry / binning / image state).
    decisions = the dict this function returned for the SAME inputs: run the arbiter build (binary64) on that run's
    radii, tile lists and contributor sets; real-valued members of the result are then float64.  `means3D` may be a
    float64 array in that case (the camera-frame means the arbiter computed itself)."""
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    arb = decisions is not None
    L = lib64() if arb else lib()
    rt, c_r, c_real = (np.float64, c_d, ctypes.c_double) if arb else (np.float32, c_f, ctypes.c_float)

    def _r(a, shape=None):
        if a is None:
            return None
        if arb and np.asarray(a).dtype == np.float64:
            a = np.ascontiguousarray(a)
            return a.reshape(shape) if shape is not None else a
        return _f64(a, shape) if arb else _f32(a, shape)
    means3D = _r(means3D, (-1, 3))
    P = means3D.shape[0]
    W, H = cam.image_width, cam.image_height
    opacities = _r(opacities, (-1,))
    colors_precomp = _r(colors_precomp, (-1, 3)) if colors_precomp is not None else None
    scales = _r(scales, (-1, 3)) if scales is not None else None
    rotations = _r(rotations, (-1, 4)) if rotations is not None else None
    cov3D_precomp = _r(cov3D_precomp, (-1, 6)) if cov3D_precomp is not None else None
    M = 0
    if shs is not None:
        shs = _r(shs)
        shs = shs.reshape(P, -1, 3)
        M = shs.shape[1]
    view = _r(cam.viewmatrix, (16,))
    proj = _r(cam.projmatrix, (16,))
    campos = _r(cam.campos, (3,))
    bg = _r(cam.bg, (3,))

    radii = np.zeros(P, dtype=np.int32)
    means2D = np.zeros((P, 2), dtype=rt)
    depths = np.zeros(P, dtype=rt)
    cov3Ds = np.zeros((P, 6), dtype=rt)
    rgb = np.zeros((P, 3), dtype=rt)
    conic_opacity = np.zeros((P, 4), dtype=rt)
    tiles_touched = np.zeros(P, dtype=np.uint32)
    clamped = np.zeros((P, 3), dtype=np.uint8)
    gx, gy = (W + 15) // 16, (H + 15) // 16

    out_color = np.zeros((3, H, W), dtype=rt)
    out_depth = np.zeros((1, H, W), dtype=rt)
    final_T = np.zeros((H, W), dtype=rt)
    n_contrib = np.zeros((H, W), dtype=np.uint32)
    ranges = np.zeros((gx * gy, 2), dtype=np.uint32)
    R = 0
    keys = np.zeros(0, dtype=np.uint64)
    point_list = np.zeros(0, dtype=np.uint32)
    if P != 0:
        L.orc_preprocess(ctypes.c_int(P), ctypes.c_int(cam.sh_degree), ctypes.c_int(M),
                         _p(means3D, c_r), _p(scales, c_r), c_real(cam.scale_modifier), _p(rotations, c_r),
                         _p(opacities, c_r), _p(shs, c_r), _p(cov3D_precomp, c_r), _p(colors_precomp, c_r),
                         _p(view, c_r), _p(proj, c_r), _p(campos, c_r),
                         ctypes.c_int(W), ctypes.c_int(H), c_real(cam.tanfovx), c_real(cam.tanfovy),
                         _p(radii, c_i32), _p(means2D, c_r), _p(depths, c_r), _p(cov3Ds, c_r), _p(rgb, c_r),
                         _p(conic_opacity, c_r), _p(tiles_touched, c_u32), _p(clamped, c_u8),
                         _p(np.ascontiguousarray(decisions["radii"], np.int32) if arb else None, c_i32))
        if arb:     # binning, sort order and tile ranges are the binary32 run's
            R, keys, point_list = decisions["num_rendered"], decisions["keys"], decisions["point_list"]
            ranges, tiles_touched = decisions["ranges"], decisions["tiles_touched"]
            assert np.array_equal(radii, decisions["radii"])
        else:
            R = int(L.orc_bin(ctypes.c_int(P), _p(means2D, c_f), _p(depths, c_f), _p(radii, c_i32),
                              _p(tiles_touched, c_u32), ctypes.c_int(W), ctypes.c_int(H),
                              ctypes.cast(None, c_u64), ctypes.cast(None, c_u32), ctypes.cast(None, c_u32)))
            keys = np.zeros(max(R, 1), dtype=np.uint64)
            point_list = np.zeros(max(R, 1), dtype=np.uint32)
            L.orc_bin(ctypes.c_int(P), _p(means2D, c_f), _p(depths, c_f), _p(radii, c_i32),
                      _p(tiles_touched, c_u32), ctypes.c_int(W), ctypes.c_int(H),
                      _p(keys, c_u64), _p(point_list, c_u32), _p(ranges, c_u32))
            keys, point_list = keys[:R], point_list[:R]
        feat = colors_precomp if colors_precomp is not None else rgb
        L.orc_render_forward(ctypes.c_int(W), ctypes.c_int(H), _p(np.ascontiguousarray(ranges), c_u32),
                             _p(np.ascontiguousarray(point_list) if R else np.zeros(1, np.uint32), c_u32),
                             _p(means2D, c_r), _p(feat, c_r), _p(conic_opacity, c_r), _p(depths, c_r), _p(bg, c_r),
                             _p(final_T, c_r), _p(n_contrib, c_u32), _p(out_color, c_r), _p(out_depth, c_r),
                             _p(decisions["means2D"] if arb else None, c_f), _p(decisions["conic_opacity"] if arb else None, c_f),
                             _p(decisions["n_contrib"] if arb else None, c_u32))
        if arb:
            assert np.array_equal(n_contrib, decisions["n_contrib"])
    return dict(color=out_color, depth=out_depth, radii=radii, num_rendered=R,
                means2D=means2D, depths=depths, cov3D=cov3Ds, rgb=rgb, conic_opacity=conic_opacity,
                tiles_touched=tiles_touched, clamped=clamped, keys=keys, point_list=point_list, ranges=ranges,
                final_T=final_T, n_contrib=n_contrib, decisions=decisions,
                inputs=dict(means3D=means3D, opacities=opacities, colors_precomp=colors_precomp, shs=shs,
                            scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp, M=M))


def rasterize_backward(cam: Camera, fwd: dict, dL_dout_color, power: int = 1):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196) -> Rasterizer::backward
    (rasterizer_impl.cu:343-434) -> renderCUDAFused (backward.cu:850-1140).
    Returns the 8 tensors of `_C.rasterize_gaussians_backward` plus dL_dconic and pair_count.
    An arbiter forward (`decisions=`) continues in the arbiter build."""
    dec = fwd.get("decisions")
    arb = dec is not None
    L = lib64() if arb else lib()
    rt, c_r, c_real = (np.float64, c_d, ctypes.c_double) if arb else (np.float32, c_f, ctypes.c_float)
    _r = _f64 if arb else _f32
    inp = fwd["inputs"]
    means3D = inp["means3D"]
    P = means3D.shape[0]
    W, H = cam.image_width, cam.image_height
    M = inp["M"]
    dL = _r(dL_dout_color, (3, H, W))
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), rt), dL_dcolors=np.zeros((P, 3), rt),
        dL_dopacity=np.zeros((P, 1), rt), dL_dmeans3D=np.zeros((P, 3), rt),
        dL_dcov3D=np.zeros((P, 6), rt), dL_dsh=np.zeros((P, M, 3), rt),
        dL_dscales=np.zeros((P, 3), rt), dL_drotations=np.zeros((P, 4), rt),
        dL_dconic=np.zeros((P, 2, 2), rt))
    pairs = ctypes.c_int64(0)
    if P != 0:
        colors = inp["colors_precomp"] if inp["colors_precomp"] is not None else fwd["rgb"]
        cov3D = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else fwd["cov3D"]
        pl = np.ascontiguousarray(fwd["point_list"]) if fwd["num_rendered"] else np.zeros(1, np.uint32)
        dsh = g["dL_dsh"] if M > 0 else np.zeros(1, rt)
        L.orc_render_backward_fused(
            ctypes.c_int(P), ctypes.c_int(cam.sh_degree), ctypes.c_int(M), ctypes.c_int(W), ctypes.c_int(H),
            _p(np.ascontiguousarray(fwd["ranges"]), c_u32), _p(pl, c_u32),
            _p(_r(cam.bg), c_r), _p(fwd["means2D"], c_r), _p(fwd["conic_opacity"], c_r), _p(colors, c_r),
            _p(fwd["final_T"], c_r), _p(fwd["n_contrib"], c_u32), _p(dL, c_r),
            _p(means3D, c_r), _p(fwd["radii"], c_i32), _p(inp["shs"], c_r), _p(fwd["clamped"], c_u8),
            _p(inp["scales"], c_r), _p(inp["rotations"], c_r), c_real(cam.scale_modifier), _p(cov3D, c_r),
            _p(_r(cam.viewmatrix), c_r), _p(_r(cam.projmatrix), c_r),
            c_real(cam.tanfovx), c_real(cam.tanfovy), _p(_r(cam.campos), c_r),
            ctypes.c_int(int(power)),
            _p(g["dL_dmeans2D"], c_r), _p(g["dL_dconic"], c_r), _p(g["dL_dopacity"], c_r), _p(g["dL_dcolors"], c_r),
            _p(g["dL_dmeans3D"], c_r), _p(g["dL_dcov3D"], c_r), _p(dsh, c_r), _p(g["dL_dscales"], c_r),
            _p(g["dL_drotations"], c_r), ctypes.byref(pairs),
            _p(dec["means2D"] if arb else None, c_f), _p(dec["conic_opacity"] if arb else None, c_f))
    g["pair_count"] = int(pairs.value)
    return g


def transform_points(w2c, pts, arbiter: bool = False):
    """World -> candidate camera frame, gaussian.py:1523-1527 (`(rel_w2c @ pts4.T).T[:, :3]`).
    The reference uses a torch fp32 matmul whose summation order is unspecified; the order fixed
    here, ((w0*x + w1*y) + w2*z) + w3 without FMA, is the one the HIP path reproduces.
    arbiter: the same sum in binary64 on the binary32 inputs."""
    rt = np.float64 if arbiter else np.float32
    w = np.asarray(w2c, dtype=np.float32).astype(rt)
    p = _f32(pts, (-1, 3)).astype(rt)
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((w[r, 0] * p[:, 0] + w[r, 1] * p[:, 1]) + w[r, 2] * p[:, 2]) + w[r, 3]
    return out


def compute_hessian(cam: Camera, w2c, means3D, rgb_colors, rotations, opacities, scales,
                    columns: int = 4, dL_scale: float = 1e-3, return_all: bool = False, arbiter: bool = False):
    """One view's Fisher-diagonal proxy.
    columns=4 : gaussian.py:1503-1570   -> [mean_cam xyz | opacity]
    columns=11: gaussian_object.py:1940-2045 -> [mean_cam xyz | opacity | scale xyz | rot rxyz]
    Inputs are the ACTIVATED render variables (normalised rotations, sigmoid opacities,
    exp scales, already tiled to 3 columns) exactly as the reference builds them at 1529-1533.
    arbiter: returns (cur_H of the binary32 oracle, cur_H of the binary64 arbiter on the same contributor sets, vis_count)."""
    pts = transform_points(w2c, means3D)
    fwd = rasterize_forward(cam, pts, opacities, colors_precomp=rgb_colors, scales=scales, rotations=rotations)
    H, W = cam.image_height, cam.image_width
    dL = np.ones((3, H, W), dtype=np.float32) * np.float32(dL_scale)
    g = rasterize_backward(cam, fwd, dL, power=2)

    def cat(g):
        parts = [g["dL_dmeans3D"], g["dL_dopacity"]]
        if columns == 11:
            parts += [g["dL_dscales"], g["dL_drotations"]]
        elif columns != 4:
            raise ValueError("columns must be 4 or 11")
        return np.concatenate(parts, axis=1)
    cur_H = cat(g)
    vis_count = int((fwd["radii"] > 0).sum())
    if arbiter:
        fwd64 = rasterize_forward(cam, transform_points(w2c, means3D, arbiter=True), opacities, colors_precomp=rgb_colors,
                                  scales=scales, rotations=rotations, decisions=fwd)
        g64 = rasterize_backward(cam, fwd64, dL, power=2)
        assert g64["pair_count"] == g["pair_count"]
        return cur_H, cat(g64), vis_count
    if return_all:
        return cur_H, vis_count, fwd, g
    return cur_H, vis_count


def compute_h_train(cam, keyframe_w2cs, means3D, rgb_colors, rotations, opacities, scales, columns=4):
    """gaussian.py:1338-1348: H_train = sum over keyframes of cur_H (fp32 adds in keyframe order)."""
    H_train = None
    for w2c in keyframe_w2cs:
        cur_H, _ = compute_hessian(cam, w2c, means3D, rgb_colors, rotations, opacities, scales, columns)
        if H_train is None:
            H_train = np.zeros_like(cur_H)
        H_train += cur_H
    return H_train


def pose_eval(cam, w2cs, H_train, means3D, rgb_colors, rotations, opacities, scales, columns=4, reg=0.1):
    """gaussian.py:1354-1375: score_v = sum(cur_H_v * 1/(H_train + 0.1)).
    Takes w2c = inv(c2w) directly (the reference inverts with torch.linalg.inv)."""
    H_inv = (np.float32(1.0) / (H_train + np.float32(reg))).astype(np.float32)
    scores = []
    vis = []
    for w2c in w2cs:
        cur_H, vc = compute_hessian(cam, w2c, means3D, rgb_colors, rotations, opacities, scales, columns)
        scores.append(float(np.sum(cur_H.astype(np.float64) * H_inv.astype(np.float64))))
        vis.append(vc)
    return np.asarray(scores, dtype=np.float64), np.asarray(vis, dtype=np.int64)


def knn_dist2(points):
    pts = _f32(points, (-1, 3))
    P = pts.shape[0]
    out = np.zeros(P, dtype=np.float32)
    lib().orc_knn_dist2(ctypes.c_int(P), _p(pts, c_f), _p(out, c_f))
    return out

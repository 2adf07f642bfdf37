"""Torch-tensor front end of the C ABI: device memory, streams and nothing else.

Three groups of entry points:
  * `rasterize_forward` / `rasterize_backward` / `mark_visible` -- the argument lists of the reference's
    pybind module `_C` (thirdparty/diff-gaussian-rasterization-modified/ext.cpp:14-18,
    rasterize_points.h:18-66); `diff_gaussian_rasterization/_C.py` re-exports them under the reference names.
  * `FisherScorer` -- the batched multi-view scorer behind GaussianSLAM.compute_Hessian / compute_H_train /
    pose_eval (models/SLAM/gaussian.py:1338-1375, 1503-1570).
  * `knn_dist2` -- simple_knn._C.distCUDA2.
"""
import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import FisherRastError, RasterCfg, Gaussians, FisherCfg


def _need_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise FisherRastError(f"{name} must live on a HIP device; fisher_rast has no CPU path")


def _prep(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    """contiguous fp32 on `device`; the reference's empty tensors (any device) become None -> null pointer."""
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _raster_cfg(P, H, W, tanfovx, tanfovy, scale_modifier, degree, M, prefiltered, bg, view, proj, campos):
    c = RasterCfg()
    c.P, c.image_height, c.image_width = int(P), int(H), int(W)
    c.tanfovx, c.tanfovy, c.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
    c.sh_degree, c.sh_coeffs, c.prefiltered = int(degree), int(M), int(bool(prefiltered))
    c.bg, c.viewmatrix, c.projmatrix, c.campos = _ptr(bg), _ptr(view), _ptr(proj), _ptr(campos)
    return c


def _gaussians(means3D, colors, sh, opacity, scales, rotations, cov3D):
    g = Gaussians()
    g.means3D, g.colors_precomp, g.shs = _ptr(means3D), _ptr(colors), _ptr(sh)
    g.opacities, g.scales, g.rotations, g.cov3D_precomp = _ptr(opacity), _ptr(scales), _ptr(rotations), _ptr(cov3D)
    return g


def workspace_bytes(P, W, H, max_rendered):
    out = (ctypes.c_size_t * 3)()
    _lib.check(_lib.load().fr_workspace_bytes(P, W, H, max_rendered, out), "fr_workspace_bytes")
    return int(out[0]), int(out[1]), int(out[2])


def workspace_layout(P, W, H, max_rendered):
    names = ("splat", "cov3D", "rgb", "clamped",
             "tile_count", "tile_offset", "tile_fill", "final_T", "n_contrib", "status", "keys")
    out = (ctypes.c_size_t * 11)()
    _lib.check(_lib.load().fr_workspace_layout(P, W, H, max_rendered, out), "fr_workspace_layout")
    return {n: int(out[i]) for i, n in enumerate(names)}


# high-water mark of tile instances per device, so that the binning buffer is normally large enough first time
_capacity_hint = {}


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (rasterize_points.cu:198-217)."""
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    P = means3D.shape[0]
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m, v, pr = _prep(means3D, dev), _prep(viewmatrix, dev), _prep(projmatrix, dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().fr_mark_visible(P, _ptr(m), _ptr(v), _ptr(pr), ctypes.c_void_p(present.data_ptr()),
                                                   _stream(dev)), "fr_mark_visible")
    return present


def rasterize_forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                      viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                      prefiltered, features=None):
    """RasterizeGaussiansCUDA (rasterize_points.cu:35-115): returns
    (num_rendered, color[3,H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer, depth[1,H,W])."""
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    lib = _lib.load()
    P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
    means3D = _prep(means3D, dev)
    colors, sh_t = _prep(colors, dev), _prep(sh, dev)
    opacity, scales, rotations, cov3D_precomp = (_prep(opacity, dev), _prep(scales, dev), _prep(rotations, dev),
                                                 _prep(cov3D_precomp, dev))
    bg, view, proj, cpos = _prep(background, dev), _prep(viewmatrix, dev), _prep(projmatrix, dev), _prep(campos, dev)
    M = int(sh.shape[1]) if (sh is not None and sh.numel() != 0) else 0

    out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    # `features` (not in the reference's argument list): a second [P,3] array composited in the same pass (fr_forward_pair);
    # its image is appended to the returned tuple
    feats = _prep(features, dev) if features is not None else None
    out_feat = torch.empty((3, H, W), dtype=torch.float32, device=dev) if features is not None else None
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    status = torch.zeros((4,), dtype=torch.int32, device=dev)
    cfg = _raster_cfg(P, H, W, tan_fovx, tan_fovy, scale_modifier, degree, M, prefiltered, bg, view, proj, cpos)
    g = _gaussians(means3D, colors, sh_t, opacity, scales, rotations, cov3D_precomp)

    key = (dev.index, P, H, W)
    capacity = max(_capacity_hint.get(key, 0), 2 * P, 1 << 16)
    with torch.cuda.device(dev):
        while True:
            gb, bb, ib = workspace_bytes(P, W, H, capacity)
            geom = torch.empty((gb,), dtype=torch.uint8, device=dev)
            binning = torch.empty((bb,), dtype=torch.uint8, device=dev)
            img = torch.empty((ib,), dtype=torch.uint8, device=dev)
            if features is not None:
                _lib.check(lib.fr_forward_pair(ctypes.byref(cfg), ctypes.byref(g), _ptr(feats), geom.data_ptr(), binning.data_ptr(),
                                               capacity, img.data_ptr(), _ptr(out_color), _ptr(out_feat), _ptr(out_depth),
                                               radii.data_ptr(), status.data_ptr(), _stream(dev)), "fr_forward_pair")
            else:
                _lib.check(lib.fr_forward(ctypes.byref(cfg), ctypes.byref(g), geom.data_ptr(), binning.data_ptr(), capacity,
                                          img.data_ptr(), _ptr(out_color), _ptr(out_depth), radii.data_ptr(),
                                          status.data_ptr(), _stream(dev)), "fr_forward")
            # same host synchronisation as the reference (rasterizer_impl.cu:282: cudaMemcpy of num_rendered)
            st = status.cpu()
            num_rendered = int(st[0])
            if int(st[3]):
                # auxiliary.h:156-160 prints this and traps the device; here the call fails and the device stays usable
                raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
            if int(st[1]) == 0:
                break
            capacity = int(num_rendered * 1.25) + 1024
    _capacity_hint[key] = max(_capacity_hint.get(key, 0), int(num_rendered * 1.25) + 1024)
    if features is not None:
        return num_rendered, out_color, radii, geom, binning, img, out_depth, out_feat
    return num_rendered, out_color, radii, geom, binning, img, out_depth


_bwd_scratch = {}


def _backward_scratch(dev, nbytes):
    """The scratch of the chunked power-2 backward, kept per (device, stream) and only ever grown: 56 MB for a 256 x 256 view of the
    benchmark room, 0.5 GB for 2M Gaussians at 512 x 512 -- a fresh `torch.empty` of that size per call sent the caching allocator
    through free / malloc cycles when image sizes alternate (1.8 ms per call instead of 0.45).  `release_backward_scratch()` drops it."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(dev).cuda_stream))
    t = _bwd_scratch.get(key)
    if t is None or t.numel() < nbytes:
        _bwd_scratch[key] = None
        t = torch.empty((nbytes + (nbytes >> 2),), dtype=torch.uint8, device=dev)
        _bwd_scratch[key] = t
    return t


def release_backward_scratch():
    _bwd_scratch.clear()


def rasterize_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                       viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer,
                       R, binningBuffer, imageBuffer, power, opacities=None, segmented=True):
    """RasterizeGaussiansBackwardCUDA (rasterize_points.cu:117-196): returns
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations).
    `segmented=False` withholds the scratch buffer of the power-2 backward (tests: the single-pass walk)."""
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    lib = _lib.load()
    P = int(means3D.shape[0])
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    M = int(sh.shape[1]) if (sh is not None and sh.numel() != 0) else 0
    means3D = _prep(means3D, dev)
    colors, sh_t = _prep(colors, dev), _prep(sh, dev)
    scales, rotations, cov3D_precomp = _prep(scales, dev), _prep(rotations, dev), _prep(cov3D_precomp, dev)
    bg, view, proj, cpos = _prep(background, dev), _prep(viewmatrix, dev), _prep(projmatrix, dev), _prep(campos, dev)
    dL = _prep(dL_dout_color, dev)

    def new(*shape):
        return torch.empty(shape, dtype=torch.float32, device=dev)
    dL_dmeans3D, dL_dmeans2D, dL_dcolors = new(P, 3), new(P, 3), new(P, 3)
    dL_dconic, dL_dopacity, dL_dcov3D = new(P, 2, 2), new(P, 1), new(P, 6)
    dL_dsh, dL_dscales, dL_drotations = torch.zeros((P, M, 3), dtype=torch.float32, device=dev), new(P, 3), new(P, 4)
    if P != 0:
        # the kernels read opacity from the splat records of the geometry buffer
        opac = _prep(opacities, dev) if opacities is not None else None
        cfg = _raster_cfg(P, H, W, tan_fovx, tan_fovy, scale_modifier, degree, M, False, bg, view, proj, cpos)
        g = _gaussians(means3D, colors, sh_t, opac, scales, rotations, cov3D_precomp)
        # scratch for the chunked backward (fisher_rast.h, fr_backward_ws; power 1 and 2)
        nscr = int(lib.fr_backward_scratch_bytes(P, W, H, int(power), int(R))) if segmented else 0
        scratch = _backward_scratch(dev, nscr) if nscr else None
        with torch.cuda.device(dev):
            _lib.check(lib.fr_backward_ws(ctypes.byref(cfg), ctypes.byref(g), radii.data_ptr(), geomBuffer.data_ptr(),
                                          binningBuffer.data_ptr(), imageBuffer.data_ptr(), _ptr(dL), int(power),
                                          _ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D),
                                          _ptr(dL_dcov3D), _ptr(dL_dsh) if M > 0 else None, _ptr(dL_dscales),
                                          _ptr(dL_drotations), _ptr(dL_dconic), int(R), scratch.data_ptr() if nscr else None,
                                          scratch.numel() if nscr else 0,
                                          _stream(dev)), "fr_backward")
    else:
        for t in (dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dconic, dL_dopacity, dL_dcov3D, dL_dscales, dL_drotations):
            t.zero_()
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def rasterize_forward_features(features, raster_cfg_args, geomBuffer, binningBuffer, imageBuffer):
    """A second [P,3] feature array composited over the geometry the last `rasterize_forward` binned
    (fr_forward_features): returns the [3,H,W] image.  raster_cfg_args = (P, H, W, tanfovx, tanfovy, scale_modifier, bg, view, proj, campos)."""
    P, H, W, tfx, tfy, mod, bg, view, proj, cpos = raster_cfg_args
    dev = geomBuffer.device if P else bg.device
    out = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    feats = _prep(features, dev)
    bg, view, proj, cpos = _prep(bg, dev), _prep(view, dev), _prep(proj, dev), _prep(cpos, dev)
    cfg = _raster_cfg(P, H, W, tfx, tfy, mod, 0, 0, False, bg, view, proj, cpos)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fr_forward_features(ctypes.byref(cfg), _ptr(feats), geomBuffer.data_ptr() if P else None,
                                                   binningBuffer.data_ptr() if P else None, imageBuffer.data_ptr() if P else None,
                                                   _ptr(out), _stream(dev)), "fr_forward_features")
    return out


def rasterize_backward_pair(background, means3D, radii, colors, features, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_features, campos,
                            geomBuffer, binningBuffer, imageBuffer, num_rendered=0, segmented=True):
    """Backward of (colour image, feature image) on shared geometry, grad_power 1 (fr_backward_pair): returns
    (dL_dmeans2D [colour image only], dL_dmeans2D_features, dL_dcolors, dL_dfeatures, dL_dopacity, dL_dmeans3D, dL_dcov3D,
     dL_dscales, dL_drotations)."""
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.shape[0])
    H, W = int(dL_dout_color.shape[1]), int(dL_dout_color.shape[2])
    means3D, colors, feats = _prep(means3D, dev), _prep(colors, dev), _prep(features, dev)
    scales, rotations, cov3D_precomp = _prep(scales, dev), _prep(rotations, dev), _prep(cov3D_precomp, dev)
    bg, view, proj, cpos = _prep(background, dev), _prep(viewmatrix, dev), _prep(projmatrix, dev), _prep(campos, dev)
    dL, dLf = _prep(dL_dout_color, dev), _prep(dL_dout_features, dev)

    def new(*shape):
        return torch.zeros(shape, dtype=torch.float32, device=dev)
    m2, m2f, dc, df = new(P, 3), new(P, 3), new(P, 3), new(P, 3)
    dop, dm3, dcov, dsc, drot, dcon = new(P, 1), new(P, 3), new(P, 6), new(P, 3), new(P, 4), new(P, 2, 2)
    if P != 0:
        cfg = _raster_cfg(P, H, W, tan_fovx, tan_fovy, scale_modifier, 0, 0, False, bg, view, proj, cpos)
        g = _gaussians(means3D, colors, None, None, scales, rotations, cov3D_precomp)
        lib = _lib.load()
        # with the forward's num_rendered: scratch for the chunked form (fisher_rast.h, fr_backward_pair_ws)
        nscr = int(lib.fr_backward_pair_scratch_bytes(P, W, H, int(num_rendered))) if (segmented and num_rendered) else 0
        scratch = _backward_scratch(dev, nscr) if nscr else None
        with torch.cuda.device(dev):
            _lib.check(lib.fr_backward_pair_ws(ctypes.byref(cfg), ctypes.byref(g), radii.data_ptr(), geomBuffer.data_ptr(),
                                               binningBuffer.data_ptr(), imageBuffer.data_ptr(), _ptr(dL), _ptr(feats), _ptr(dLf),
                                               _ptr(m2), _ptr(m2f), _ptr(dc), _ptr(df), _ptr(dop), _ptr(dm3), _ptr(dcov),
                                               _ptr(dsc), _ptr(drot), _ptr(dcon), int(num_rendered), scratch.data_ptr() if nscr else None,
                                               scratch.numel() if nscr else 0, _stream(dev)), "fr_backward_pair")
    return m2, m2f, dc, df, dop, dm3, dcov, dsc, drot


class FisherScorer:
    """Batched Fisher-information scorer for one Gaussian map and one camera.

    Holds the activated render variables (what gaussian.py:1529-1543 builds per call) and a reusable
    workspace.  `run()` scores / accumulates any number of views with no host synchronisation; the caller
    synchronises when it reads the results.  Overflow of the tile-instance buffer is detected from the
    device status word when results are fetched (`fetch`), and the batch is re-run with a larger buffer.
    """

    WORKSPACE_BUDGET = 32 << 30  # bytes of workspace (of the MI355X's 288 GB) before views are processed in chunks: ~580 views at 500k Gaussians
    MAX_KEY_BYTES_PER_VIEW = 512 << 20  # fixed key segments beyond this per view: packed lists instead (tile_capacity = 0)

    def __init__(self, raster_settings, means3D, rgb_colors, rotations, opacities, scales, columns: int = 4,
                 dL_dpix: float = 1e-3, tile_capacity: int = -1, spatial_order: bool = False):
        _need_gpu(means3D, "means3D")
        if columns not in (4, 11):
            raise ValueError("columns must be 4 or 11")
        self.lib = _lib.load()
        self.dev = means3D.device
        self.rs = raster_settings
        self.columns = columns
        self.dL = float(dL_dpix)
        d = self.dev
        self.means3D, self.colors = _prep(means3D.detach(), d), _prep(rgb_colors.detach(), d)
        self.rotations, self.opacities = _prep(rotations.detach(), d), _prep(opacities.detach().reshape(-1), d)
        scales = scales.detach()
        if scales.dim() == 2 and scales.shape[-1] == 1:  # isotropic (gaussian.py:1532-1533)
            scales = torch.tile(scales, (1, 3))
        self.scales = _prep(scales, d)
        self.P = int(self.means3D.shape[0])
        self.H, self.W = int(raster_settings.image_height), int(raster_settings.image_width)
        self.bg = _prep(raster_settings.bg, d)
        self.view = _prep(raster_settings.viewmatrix, d)
        self.proj = _prep(raster_settings.projmatrix, d)
        self.campos = _prep(raster_settings.campos, d)
        self._ws = {}
        self._static_key, self._static_hinv = None, None
        self.per_view_capacity = max(int(0.75 * self.P), 1 << 16)
        # Fixed key segments (fr_fisher_cfg.tile_capacity): every (view, tile) owns `tile_capacity` key slots, the projection
        # kernel places the keys itself and the scan / scatter kernels drop out of the launch sequence.  16384 keys (the largest
        # list the in-LDS sort tiers take) x 8 B = 128 KiB per tile -- 32 MiB per 256 x 256 view of the 288 GB; a longer list
        # raises the overflow flag and `run` grows the segments, or goes back to packed lists where they would not fit.
        self.tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        # (default: 32768 keys -- the library then partitions every list of up to 16320 keys behind itself, k_sort_part, and the
        # 1024-thread bitonic tier and its side stream only see longer lists -- or 16384 where that would pass the cap per view)
        if tile_capacity < 0:
            tile_capacity = 32768 if self.tiles * 32768 * 8 <= self.MAX_KEY_BYTES_PER_VIEW else 16384
        self.tile_capacity = int(tile_capacity)            # 0: packed key lists
        if self.tiles * self.tile_capacity * 8 > self.MAX_KEY_BYTES_PER_VIEW:
            self.tile_capacity = 0
        self.cfg = _raster_cfg(self.P, self.H, self.W, raster_settings.tanfovx, raster_settings.tanfovy,
                               raster_settings.scale_modifier, raster_settings.sh_degree, 0,
                               raster_settings.prefiltered, self.bg, self.view, self.proj, self.campos)
        self.g = _gaussians(self.means3D, self.colors, None, self.opacities, self.scales, self.rotations, None)
        # spatial_order=True: the Gaussians along a Z-curve (fr_fisher_cfg.order), computed once per map; the library lays them out and
        # processes them in that order -- a projection workgroup's 256 Gaussians are then neighbours in space (whole groups fall outside a
        # view and are skipped, a workgroup's keys land in a handful of tiles, a tile's records sit side by side).  Inputs and outputs
        # keep the caller's indexing.  OFF by default: measured on MI355X (500k Gaussians x 64 views, profiles/r04_d_spatial_order.txt) the
        # tile kernel gains 3.7 % and the projection kernel 2.8 %, the gathering k_pack_static loses as much (51 against 22 us) -- and two
        # DISTINCT splats of bit-equal depth in one tile (about one pair per view) then composite in Z-curve order, not in the reference's
        # index order (scores move by ~1e-5; exact duplicates keep their order).
        self.order = spatial_order_of(self.means3D) if (spatial_order and self.P > 0) else None

    # -- helpers -------------------------------------------------------------------------------------
    def max_views_per_launch(self):
        """Views per fr_fisher_views call that keep the workspace within WORKSPACE_BUDGET (the per-view share is what the
        library itself reports: records, visible lists, keys, per-tile arrays)."""
        one = int(self.lib.fr_fisher_workspace_bytes(self.P, self.W, self.H, 1, self._keys_per_view(), self.columns))
        eight = int(self.lib.fr_fisher_workspace_bytes(self.P, self.W, self.H, 8, 8 * self._keys_per_view(), self.columns))
        per_view = max(1, (eight - one) // 7)
        n = max(1, int(self.WORKSPACE_BUDGET // per_view))
        if self.tile_capacity > 0:                      # fixed segments are addressed with 32-bit key offsets
            n = max(1, min(n, ((1 << 32) - 1) // (self.tiles * self.tile_capacity)))
        return n

    def _keys_per_view(self):
        return max(self.per_view_capacity, self.tiles * self.tile_capacity)

    def _workspace(self, V, max_rendered, slot=0):
        nbytes = int(self.lib.fr_fisher_workspace_bytes(self.P, self.W, self.H, V, max_rendered, self.columns))
        if nbytes == 0:
            raise FisherRastError("fr_fisher_workspace_bytes: bad argument")
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < nbytes:
            self._ws[slot] = None
            ws = self._ws[slot] = torch.empty((nbytes,), dtype=torch.uint8, device=self.dev)
        return ws

    def launch(self, w2c, H_inv=None, H_inv_per_view=False, out_H=None, out_H_per_view=False, dL_image=None, poses_are_c2w=False):
        """Enqueue one batch (no sync).  w2c: [V,4,4] world->camera on the device (camera->world with `poses_are_c2w`: the
        library inverts them).
        Returns a dict of device tensors: scores [V] (if H_inv), vis_count [V], num_rendered [V], status [4]."""
        d = self.dev
        w2c = _prep(w2c.reshape(-1, 4, 4), d)
        V = int(w2c.shape[0])
        C = self.columns
        PC = self.P * C
        scores = None
        if H_inv is not None:
            H_inv = _prep(H_inv, d)
            want = (V * PC) if H_inv_per_view else PC
            if H_inv.numel() != want:
                raise ValueError(f"H_inv has {H_inv.numel()} elements, expected {want}")
            scores = torch.empty((V,), dtype=torch.float32, device=d)     # every element is written (or the status word says overflow)
        if out_H is not None:
            want = (V * PC) if out_H_per_view else PC
            if out_H.numel() != want or out_H.dtype != torch.float32 or not out_H.is_contiguous() or out_H.device != d:
                raise ValueError("out_H must be a contiguous fp32 device tensor of [V,]P*columns elements")
        HW3 = 3 * int(self.rs.image_height) * int(self.rs.image_width)
        if dL_image is not None:
            # per view an upstream-gradient image [V,3,H,W] (or one [3,H,W] shared): the random probes of the POp-GS estimators
            if out_H is None or H_inv is not None:
                raise ValueError("dL_image goes with out_H (no H_inv)")
            dL_image = _prep(dL_image, d)
            if dL_image.numel() not in (HW3, V * HW3):
                raise ValueError(f"dL_image has {dL_image.numel()} elements, expected {HW3} or {V * HW3}")
        # zero-filled / fully written by the library itself (k_zero_many, k_scan_tiles, k_reduce_scores): no fill kernels here
        vis = torch.empty((V,), dtype=torch.int32, device=d)
        nr = torch.empty((V,), dtype=torch.int32, device=d)
        status = torch.empty((4,), dtype=torch.int32, device=d)
        # ONE fr_fisher_views call per launch: a call accumulates into out_H all or nothing (overflow: nothing), which is what lets
        # `run` simply redo a batch.  (Round 3 could cut a batch into view groups on separate streams; it measured slower -- 2.47 ms
        # against 2.08 ms per step -- and a group that had not overflowed would have been added to out_H twice by the redo.)
        max_rendered = V * self._keys_per_view()
        ws = self._workspace(V, max_rendered)
        fc = FisherCfg()
        fc.n_views, fc.columns, fc.dL_dpix = V, C, self.dL
        fc.poses_are_c2w = 1 if poses_are_c2w else 0
        fc.tile_capacity = self.tile_capacity if V * self.tiles * self.tile_capacity < (1 << 32) else 0
        fc.w2c = ctypes.c_void_p(w2c.data_ptr())
        if H_inv is not None:
            fc.H_inv = ctypes.c_void_p(H_inv.data_ptr())
            fc.H_inv_view_stride = PC if H_inv_per_view else 0
            fc.out_scores = ctypes.c_void_p(scores.data_ptr())
        if out_H is not None:
            fc.out_H = ctypes.c_void_p(out_H.data_ptr())
            fc.out_H_view_stride = PC if out_H_per_view else 0
        if dL_image is not None:
            fc.dL_dpix_image = ctypes.c_void_p(dL_image.data_ptr())
            fc.dL_image_view_stride = HW3 if dL_image.numel() != HW3 else 0
        fc.out_vis_count = vis.data_ptr()
        fc.out_num_rendered = nr.data_ptr()
        fc.order = self.order.data_ptr() if self.order is not None else None
        # the per-Gaussian static records (means, cov3D, colours, shared H_inv rows) are packed into the workspace by every call; a call
        # that finds there what it would write -- same workspace and layout, same shared H_inv tensor in the same version (this scorer's
        # Gaussians never change) -- skips that kernel (fr_fisher_cfg.reuse_static)
        # (the H_inv TENSOR is held on to: a fresh tensor of a later call can then not land on its address and pass for it)
        shared = None if (H_inv is None or H_inv_per_view) else H_inv
        skey = (ws.data_ptr(), ws.numel(), V, max_rendered, fc.tile_capacity, None if shared is None else shared._version)
        if H_inv is not None and out_H is not None:
            # scores AND diagonals in one call run the two-pass fall-back kernel, which packs its own static records and leaves the
            # front end's {mean, trace} array unwritten: nothing of this call may be reused by the next one
            skey = None
        fc.reuse_static = 1 if (skey is not None and self._static_key == skey and self._static_hinv is shared) else 0
        self._static_key, self._static_hinv = None, None            # (set again once the call has been enqueued without an error)
        with torch.cuda.device(d):
            _lib.check(self.lib.fr_fisher_views(ctypes.byref(self.cfg), ctypes.byref(self.g), ctypes.byref(fc),
                                                ws.data_ptr(), ws.numel(), max_rendered,
                                                status.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream)),
                       "fr_fisher_views")
        self._static_key, self._static_hinv = skey, shared
        return dict(scores=scores, vis_count=vis, num_rendered=nr, status=status, n_views=V, _keep=(w2c, H_inv, dL_image))

    def run(self, w2c, H_inv=None, H_inv_per_view=False, out_H=None, out_H_per_view=False, dL_image=None, poses_are_c2w=False):
        """launch() + overflow handling.  Synchronises once (to read the 16-byte status word)."""
        w2c = w2c.reshape(-1, 4, 4)
        V = int(w2c.shape[0])
        chunk = self.max_views_per_launch()
        outs = []
        v0 = 0
        while v0 < V:
            v1 = min(V, v0 + chunk)
            hi = H_inv
            if H_inv is not None and H_inv_per_view:
                hi = H_inv.reshape(V, -1)[v0:v1]
            oh = out_H
            if out_H is not None and out_H_per_view:
                oh = out_H.view(V, -1)[v0:v1]
            while True:
                dl = dL_image
                if dL_image is not None and dL_image.dim() == 4 and dL_image.shape[0] == V:
                    dl = dL_image[v0:v1]
                r = self.launch(w2c[v0:v1], hi, H_inv_per_view, oh, out_H_per_view, dl, poses_are_c2w)
                st = r["status"].cpu()
                if int(st[1]) == 0:
                    break
                # tile-instance buffer too small: NOTHING was scored or accumulated (every kernel behind the scan returns on the
                # overflow flag, include/fisher_rast.h), so out_H is as it was: grow and redo this chunk
                if int(st[3]):
                    # a tile list longer than its fixed segment (st[2] = the longest): longer segments, or packed lists
                    want = (int(int(st[2]) * 1.25) + 1023) // 1024 * 1024
                    self.tile_capacity = want if self.tiles * want * 8 <= self.MAX_KEY_BYTES_PER_VIEW else 0
                self.per_view_capacity = max(self.per_view_capacity, int(int(st[0]) * 1.25 / (v1 - v0)) + 4096)
                chunk = min(chunk, self.max_views_per_launch())
                if v1 - v0 > chunk:
                    v1 = v0 + chunk
                    if H_inv is not None and H_inv_per_view:
                        hi = H_inv.reshape(V, -1)[v0:v1]
                    if out_H is not None and out_H_per_view:
                        oh = out_H.view(V, -1)[v0:v1]
            outs.append(r)
            v0 = v1
        res = dict(vis_count=torch.cat([o["vis_count"] for o in outs]),
                   num_rendered=torch.cat([o["num_rendered"] for o in outs]))
        res["scores"] = torch.cat([o["scores"] for o in outs]) if H_inv is not None else None
        return res


def spatial_order_of(means3D: torch.Tensor) -> torch.Tensor:
    """fr_spatial_order: int32 [P], entry k = index of the k-th Gaussian along the Z-curve of the means (stable for equal codes)."""
    _need_gpu(means3D, "means3D")
    dev = means3D.device
    lib = _lib.load()
    pts = _prep(means3D.detach(), dev)
    P = int(pts.shape[0])
    order = torch.empty((P,), dtype=torch.int32, device=dev)
    ws = torch.empty((max(int(lib.fr_spatial_order_workspace_bytes(P)), 1),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.fr_spatial_order(P, _ptr(pts), order.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)), "fr_spatial_order")
    return order


def knn_dist2(points: torch.Tensor) -> torch.Tensor:
    """simple_knn._C.distCUDA2: mean squared distance to the 3 nearest other points, [P,3] -> [P]."""
    _need_gpu(points, "points")
    dev = points.device
    lib = _lib.load()
    pts = _prep(points, dev)
    P = int(pts.shape[0]) if pts is not None else 0
    out = torch.zeros((P,), dtype=torch.float32, device=dev)
    if P == 0:
        return out
    nbytes = int(lib.fr_knn_workspace_bytes(P))
    ws = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.fr_knn_dist2(P, _ptr(pts), _ptr(out), ws.data_ptr(), ws.numel(), _stream(dev)), "fr_knn_dist2")
    return out

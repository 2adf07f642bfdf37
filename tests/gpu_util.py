"""Helpers for the -m gpu parity tests: run the HIP path through the C ABI on the same NumPy inputs as the oracle
and unpack the opaque work buffers with fr_workspace_layout."""
import numpy as np
import torch


def to_dev(a, dev, dtype=torch.float32):
    if a is None:
        return torch.Tensor([])
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)


def hip_forward(dev, cam, means3D, opacities, colors_precomp=None, shs=None, scales=None, rotations=None, cov3D_precomp=None):
    from fisher_rast import ops
    P = int(np.asarray(means3D).reshape(-1, 3).shape[0])
    W, H = cam.image_width, cam.image_height
    t = dict(bg=to_dev(cam.bg, dev), means3D=to_dev(np.asarray(means3D, np.float32).reshape(-1, 3), dev),
             colors=to_dev(colors_precomp, dev), opacity=to_dev(np.asarray(opacities, np.float32).reshape(-1, 1), dev),
             scales=to_dev(scales, dev), rotations=to_dev(rotations, dev), cov3D=to_dev(cov3D_precomp, dev),
             view=to_dev(cam.viewmatrix, dev), proj=to_dev(cam.projmatrix, dev), sh=to_dev(shs, dev), campos=to_dev(cam.campos, dev))
    if shs is not None:
        t["sh"] = t["sh"].reshape(P, -1, 3)
    R, color, radii, geom, binning, img, depth = ops.rasterize_forward(
        t["bg"], t["means3D"], t["colors"], t["opacity"], t["scales"], t["rotations"], cam.scale_modifier, t["cov3D"],
        t["view"], t["proj"], cam.tanfovx, cam.tanfovy, H, W, t["sh"], cam.sh_degree, t["campos"], cam.prefiltered)
    torch.cuda.synchronize()
    out = dict(num_rendered=R, color=color.cpu().numpy(), depth=depth.cpu().numpy(), radii=radii.cpu().numpy(),
               tensors=t, buffers=(geom, binning, img), radii_t=radii)
    if P == 0:
        return out
    L = ops.workspace_layout(P, W, H, 1)
    g = geom.cpu().numpy()
    im = img.cpu().numpy()
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def sec(buf, off, count, dt):
        return np.frombuffer(buf[off:off + count * np.dtype(dt).itemsize].tobytes(), dtype=dt)
    splat = sec(g, L["splat"], 8 * P, np.float32).reshape(P, 8)
    out["depths"] = splat[:, 6].copy()
    out["means2D"] = splat[:, 0:2].copy()
    out["conic_opacity"] = splat[:, 2:6].copy()
    out["cov3D"] = sec(g, L["cov3D"], 6 * P, np.float32).reshape(P, 6)
    out["rgb"] = sec(g, L["rgb"], 3 * P, np.float32).reshape(P, 3)
    out["clamped"] = sec(g, L["clamped"], 3 * P, np.uint8).reshape(P, 3)
    cnt = sec(im, L["tile_count"], T, np.uint32)
    off = sec(im, L["tile_offset"], T, np.uint32)
    out["final_T"] = sec(im, L["final_T"], W * H, np.float32).reshape(H, W)
    out["n_contrib"] = sec(im, L["n_contrib"], W * H, np.uint32).reshape(H, W)
    keys = sec(binning.cpu().numpy(), 0, R, np.uint64) if R > 0 else np.zeros(0, np.uint64)
    out["keys"] = keys
    out["point_list"] = (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    rng = np.stack([off, off + cnt], 1).astype(np.uint32)
    rng[cnt == 0] = 0     # the reference leaves untouched tiles at (0,0) (rasterizer_impl.cu:311)
    out["ranges"] = rng
    out["tile_count"] = cnt
    return out


def hip_backward(dev, cam, fwd, dL, power):
    from fisher_rast import ops
    t = fwd["tensors"]
    geom, binning, img = fwd["buffers"]
    outs = ops.rasterize_backward(t["bg"], t["means3D"], fwd["radii_t"], t["colors"], t["scales"], t["rotations"],
                                  cam.scale_modifier, t["cov3D"], t["view"], t["proj"], cam.tanfovx, cam.tanfovy,
                                  to_dev(dL, dev), t["sh"], cam.sh_degree, t["campos"], geom, fwd["num_rendered"], binning, img, power)
    torch.cuda.synchronize()
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    return {n: o.cpu().numpy() for n, o in zip(names, outs)}


def assert_close(got, want, rtol, name, atol_frac=1e-6):
    """|got - want| <= rtol*|want| + atol_frac*max|want| element-wise."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if want.size == 0:
        return
    tol = rtol * np.abs(want) + atol_frac * np.abs(want).max()
    bad = np.abs(got - want) > tol
    assert not bad.any(), (name, int(bad.sum()), float(np.abs(got - want).max()), float(np.abs(want).max()))

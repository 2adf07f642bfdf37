"""-m gpu: the two key layouts of fr_fisher_views -- packed lists (count, scan, scatter kernel; fr_fisher_cfg.tile_capacity = 0)
and fixed per-(view, tile) segments filled by the projection kernel itself (tile_capacity > 0, the scorer's default).  The sorted
lists hold the same contributing splats in the same order, so scores are bit-identical; a list longer than its segment raises the overflow flag, nothing is
scored or accumulated, and FisherScorer.run grows the segments (or goes back to packed lists) and repeats the launch."""
import ctypes

import numpy as np
import pytest
import torch

from scenes import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(gpu):
    from fisher_rast import synthetic
    from models.SLAM.utils.recon_helpers import setup_camera
    P, V, W, H = 60_000, 16, 256, 256
    act = synthetic.activate(synthetic.room_shell(P, seed=7))
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=7)).to(gpu)
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    H_inv = (torch.rand((P, 11), generator=torch.Generator().manual_seed(5)) + 0.05).to(gpu)
    return dict(P=P, V=V, W=W, H=H, act=act, w2c=w2c, cam=cam, H_inv=H_inv)


def _scorer(s, gpu, columns, tile_capacity):
    from fisher_rast.ops import FisherScorer
    sc = FisherScorer(s["cam"], *(s["act"][k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")), columns=columns)
    sc.tile_capacity = tile_capacity
    return sc


@pytest.mark.parametrize("columns", [4, 11])
def test_fixed_segments_equal_packed_lists(scene, gpu, columns):
    s = scene
    hinv = s["H_inv"][:, :columns].contiguous()
    packed = _scorer(s, gpu, columns, 0)
    fixed = _scorer(s, gpu, columns, 16384)
    a = packed.run(s["w2c"], H_inv=hinv)
    b = fixed.run(s["w2c"], H_inv=hinv)
    assert packed.tile_capacity == 0 and fixed.tile_capacity == 16384          # no overflow, no change of layout
    assert torch.equal(a["scores"], b["scores"]) and float(a["scores"].min()) > 0
    assert torch.equal(a["vis_count"], b["vis_count"]) and torch.equal(a["num_rendered"], b["num_rendered"])
    # the accumulating modes (float atomics: equal up to the order of the adds)
    Ha = torch.zeros((s["P"], columns), device=gpu)
    Hb = torch.zeros((s["P"], columns), device=gpu)
    packed.run(s["w2c"], out_H=Ha)
    fixed.run(s["w2c"], out_H=Hb)
    assert rel_err(Hb.cpu().numpy(), Ha.cpu().numpy()) < 1e-5 and float(Ha.max()) > 0


def test_a_list_longer_than_its_segment_grows_the_segments(scene, gpu):
    s = scene
    hinv = s["H_inv"][:, :4].contiguous()
    packed = _scorer(s, gpu, 4, 0)
    want = packed.run(s["w2c"], H_inv=hinv)
    listed = int(packed.launch(s["w2c"], H_inv=hinv)["status"].cpu()[0])     # tile instances listed (after the alpha-footprint cut)
    sc = _scorer(s, gpu, 4, 64)                                               # far too short: the first launch overflows
    r = sc.launch(s["w2c"], H_inv=hinv)
    st = r["status"].cpu().numpy()
    # (fixed segments list a tile only where the footprint test of the tile kernel's waves admits one of its strips: a few keys fewer)
    assert st[1] == 1 and st[3] == 1 and st[2] > 64 and 0.9 * listed < st[0] <= listed
    Hacc = torch.zeros((s["P"], 4), device=gpu)
    sc.launch(s["w2c"], out_H=Hacc)
    assert float(Hacc.abs().max()) == 0.0                                       # overflow: nothing accumulated
    got = sc.run(s["w2c"], H_inv=hinv)                                          # grows to 1.25 x the longest list, repeats
    assert sc.tile_capacity >= int(st[2]) and sc.tile_capacity % 1024 == 0
    assert torch.equal(got["scores"], want["scores"])
    # segments that could never fit: back to packed lists
    sc2 = _scorer(s, gpu, 4, 64)
    sc2.MAX_KEY_BYTES_PER_VIEW = 1 << 16
    got2 = sc2.run(s["w2c"], H_inv=hinv)
    assert sc2.tile_capacity == 0 and torch.equal(got2["scores"], want["scores"])


def test_segments_beyond_the_key_buffer_are_refused(scene, gpu):
    from fisher_rast import _lib
    from fisher_rast.ops import FisherCfg
    s = scene
    sc = _scorer(s, gpu, 4, 0)
    V, P = s["V"], s["P"]
    hinv = s["H_inv"][:, :4].contiguous()
    max_rendered = V * sc.per_view_capacity
    ws = torch.empty((int(sc.lib.fr_fisher_workspace_bytes(P, s["W"], s["H"], V, max_rendered, 4)),), dtype=torch.uint8, device=gpu)
    scores = torch.empty((V,), device=gpu)
    status = torch.zeros((4,), dtype=torch.int32, device=gpu)
    fc = FisherCfg()
    fc.n_views, fc.columns, fc.dL_dpix = V, 4, 1e-3
    fc.w2c = ctypes.c_void_p(s["w2c"].data_ptr())
    fc.H_inv = ctypes.c_void_p(hinv.data_ptr())
    fc.out_scores = ctypes.c_void_p(scores.data_ptr())
    fc.tile_capacity = max_rendered // (V * 256) + 1
    rc = sc.lib.fr_fisher_views(ctypes.byref(sc.cfg), ctypes.byref(sc.g), ctypes.byref(fc), ws.data_ptr(), ws.numel(), max_rendered,
                                status.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(gpu).cuda_stream))
    assert rc != 0 and "tile_capacity" in sc.lib.fr_last_error().decode()
    with pytest.raises(_lib.FisherRastError):
        _lib.check(rc, "fr_fisher_views")


def test_images_taller_than_the_strip_row_byte_fall_back_to_packed_lists(gpu):
    """The 8-byte list entries of the fixed-segment front end hold strip rows in a byte: beyond 63 tile rows (1008 pixels) the
    library keeps packed lists whatever tile_capacity says -- same scores, no overflow."""
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    P, V, W, H = 20_000, 8, 64, 1040
    act = synthetic.activate(synthetic.room_shell(P, seed=9))
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=9)).to(gpu)
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    hinv = (torch.rand((P, 4), generator=torch.Generator().manual_seed(6)) + 0.05).to(gpu)
    args = [act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")]
    a, b = FisherScorer(cam, *args), FisherScorer(cam, *args)
    a.tile_capacity = 0
    assert b.tile_capacity > 0
    ra, rb = a.run(w2c, H_inv=hinv), b.run(w2c, H_inv=hinv)
    assert torch.equal(ra["scores"], rb["scores"]) and float(ra["scores"].max()) > 0
    assert torch.equal(ra["num_rendered"], rb["num_rendered"])
    st = b.launch(w2c, H_inv=hinv)["status"].cpu()
    assert int(st[1]) == 0 and int(st[3]) == 0


def test_views_that_do_not_fill_the_xcds_and_tiny_maps(gpu):
    """V % 8 != 0 (no deal of the views over the XCDs), a single view, fewer Gaussians than one projection workgroup holds."""
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    W = H = 128
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    for P, V in ((100, 1), (3000, 5), (3000, 13)):
        act = synthetic.activate(synthetic.room_shell(P, seed=11))
        w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=11)).to(gpu)
        hinv = (torch.rand((P, 4), generator=torch.Generator().manual_seed(7)) + 0.05).to(gpu)
        args = [act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")]
        a, b = FisherScorer(cam, *args), FisherScorer(cam, *args)
        a.tile_capacity = 0
        ra, rb = a.run(w2c, H_inv=hinv), b.run(w2c, H_inv=hinv)
        assert torch.equal(ra["scores"], rb["scores"]) and torch.equal(ra["vis_count"], rb["vis_count"])
        # one view at a time == the batch (a view's score does not depend on its batch)
        one = torch.cat([b.run(w2c[v:v + 1], H_inv=hinv)["scores"] for v in range(V)])
        assert torch.equal(one, rb["scores"])


def test_an_image_of_4032_tiles_keeps_fixed_segments_within_the_lds(gpu):
    """64 x 63 tiles: the projection kernel's per-(view, tile) histogram AND its cursors must fit the LDS a launch may ask for --
    the library takes fewer views per workgroup there; scores equal those of the packed-list path, no overflow."""
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    P, V, W, H = 30_000, 8, 1024, 1008
    act = synthetic.activate(synthetic.room_shell(P, seed=13))
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=13)).to(gpu)
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    hinv = (torch.rand((P, 4), generator=torch.Generator().manual_seed(8)) + 0.05).to(gpu)
    args = [act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")]
    a, b = FisherScorer(cam, *args), FisherScorer(cam, *args)
    a.tile_capacity = 0
    b.tile_capacity = 2048                                   # 4032 tiles x 2048 keys x 8 B = 63 MiB per view
    ra, rb = a.run(w2c, H_inv=hinv), b.run(w2c, H_inv=hinv)
    assert b.tile_capacity == 2048
    assert torch.equal(ra["scores"], rb["scores"]) and float(ra["scores"].min()) > 0
    assert torch.equal(ra["vis_count"], rb["vis_count"])


def test_camera_to_world_poses_are_inverted_by_the_library(scene, gpu):
    """fr_fisher_cfg.poses_are_c2w: the same scores as with torch.linalg.inv on the host side (the inverse is taken in double
    and rounded once; torch's LU runs in float), for rigid poses and for a sheared, scaled one."""
    s = scene
    hinv = s["H_inv"][:, :4].contiguous()
    sc = _scorer(s, gpu, 4, 16384)
    w2c = s["w2c"].clone()
    w2c[3, :3, :3] = w2c[3, :3, :3] @ torch.tensor([[1.02, 0.03, 0.0], [0.0, 0.97, 0.01], [0.0, 0.0, 1.0]], device=gpu)
    c2w = torch.linalg.inv(w2c.double()).float()
    want = sc.run(torch.linalg.inv(c2w.double()).float(), H_inv=hinv)["scores"]
    got = sc.launch(c2w, H_inv=hinv, poses_are_c2w=True)
    assert int(got["status"].cpu()[1]) == 0
    assert rel_err(got["scores"].cpu().numpy(), want.cpu().numpy()) < 2e-5


def test_static_records_are_reused_only_when_nothing_changed(scene, gpu):
    """fr_fisher_cfg.reuse_static: a second call with the same shared H_inv tensor (same version) skips the packing kernel and gives the
    same scores bit for bit; an in-place change of H_inv, another H_inv tensor, another number of views or a grown workspace packs again
    -- checked through the scores (a stale record would give the old H_inv's scores)."""
    s = scene
    sc = _scorer(s, gpu, 4, 16384)
    h1 = s["H_inv"][:, :4].contiguous()
    w = s["w2c"]
    a = sc.run(w, H_inv=h1)["scores"].clone()
    assert sc._static_key is not None
    b = sc.run(w, H_inv=h1)["scores"].clone()                     # reuse
    assert torch.equal(a, b)
    h1.mul_(2.0)                                                  # in place: the version changes
    c = sc.run(w, H_inv=h1)["scores"].clone()
    assert torch.allclose(c, 2.0 * a, rtol=1e-6)
    h2 = (h1 * 0.5)                                               # another tensor (possibly at a recycled address)
    d = sc.run(w, H_inv=h2)["scores"].clone()
    assert torch.allclose(d, a, rtol=1e-6)
    del h2
    h3 = torch.full_like(h1, 0.25)
    e = sc.run(w[:8], H_inv=h3)["scores"].clone()                 # fewer views: another layout
    f = sc.run(w[:8], H_inv=torch.full_like(h1, 0.5))["scores"].clone()
    assert torch.allclose(f, 2.0 * e, rtol=1e-6)
    # out_H mode after a score call and back: H_inv rows are not part of the out_H records' static part, but the layout key changes
    Ht = torch.zeros((s["P"], 4), device=gpu)
    sc.run(w[:8], out_H=Ht)
    Ht2 = torch.zeros((s["P"], 4), device=gpu)
    sc.run(w[:8], out_H=Ht2)                                      # reuse in the out_H mode
    assert rel_err(Ht2.cpu().numpy(), Ht.cpu().numpy()) < 1e-5 and float(Ht.max()) > 0
    g = sc.run(w[:8], H_inv=h3)["scores"].clone()
    assert torch.allclose(g, e, rtol=1e-6)
    # scores AND diagonals in one call (the two-pass fall-back: it leaves the front end's static arrays unwritten) must not be taken
    # for a call whose static records can be reused -- on a FRESH scorer, so that nothing valid is left over from earlier calls
    sc2 = _scorer(s, gpu, 4, 16384)
    hv = torch.full((8, s["P"], 4), 0.25, device=gpu)
    cur = torch.zeros((8, s["P"], 4), device=gpu)
    both = sc2.run(w[:8], H_inv=hv, H_inv_per_view=True, out_H=cur, out_H_per_view=True)["scores"].clone()
    assert torch.allclose(both, e, rtol=1e-5)
    cur2 = torch.zeros((8, s["P"], 4), device=gpu)
    sc2.run(w[:8], out_H=cur2, out_H_per_view=True)                # the records path right after it: must pack
    assert rel_err(cur2.cpu().numpy(), cur.cpu().numpy()) < 1e-5 and float(cur2.sum(dim=(1, 2)).min()) > 0


@pytest.mark.parametrize("P,capacity", [(2300, 16384), (2600, 16384), (5200, 16384), (7000, 16384), (8192, 16384), (8300, 16384), (12000, 16384),
                                        (5200, 32768), (8300, 32768), (12000, 32768), (16300, 32768), (17000, 32768)])
def test_long_lists_in_fixed_segments_are_partitioned_and_sorted(gpu, P, capacity):
    """k_sort_part: lists of 2049 .. capacity / 2 - 64 keys in fixed segments are split at sampled pivots into wave-sized parts, sorted behind the
    list, and the tile's offset moves there.  One tile of a 48 x 48 view takes nearly every splat; depths cluster (three 'walls') and
    a quarter of the splats are exact duplicates (runs of equal depth, split by the slot).  The segment must come out strictly
    ascending with the list's own keys, and the scores must equal the packed-list path's (bitonic tiers) bit for bit."""
    import ctypes
    from fisher_rast import _lib
    from fisher_rast.ops import FisherScorer
    from fisher_rast.synthetic import intrinsics
    from models.SLAM.utils.recon_helpers import setup_camera
    rng = np.random.default_rng(P)
    W = H = 48
    z = np.concatenate([rng.normal(2.0, 0.004, P // 3), rng.normal(3.5, 0.004, P // 3), rng.uniform(1.0, 6.0, P - 2 * (P // 3))]).astype(np.float32)
    u = rng.uniform(17.0, 30.0, P); v = rng.uniform(17.0, 30.0, P)
    means = np.stack([(u - 23.5) / 24.0 * z, (v - 23.5) / 24.0 * z, z], 1).astype(np.float32)
    means[P // 2:P // 2 + P // 4] = means[:P // 4]                       # duplicates
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    args = (t(means), t(rng.uniform(0, 1, (P, 3)).astype(np.float32)), t(np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1))),
            t(rng.uniform(0.005, 0.02, P).astype(np.float32)), t(np.full((P, 3), 0.004, np.float32) * z[:, None]))
    cam = setup_camera(W, H, intrinsics(W, H), np.eye(4), device=gpu)
    w2c = torch.eye(4, device=gpu)[None].repeat(2, 1, 1)
    w2c[1, 0, 3] = 0.01
    Hi = (torch.rand((P, 4), generator=torch.Generator().manual_seed(1)) + 0.05).to(gpu)
    fixed = FisherScorer(cam, *args, tile_capacity=capacity)
    packed = FisherScorer(cam, *args, tile_capacity=0)
    a, b = fixed.run(w2c, H_inv=Hi), packed.run(w2c, H_inv=Hi)
    assert fixed.tile_capacity == capacity
    assert torch.equal(a["num_rendered"], b["num_rendered"]) and torch.equal(a["scores"], b["scores"]) and float(a["scores"].min()) > 0
    # the sorted segments themselves
    V, T = 2, 9
    cap = V * fixed._keys_per_view()
    off = (ctypes.c_size_t * 8)()
    _lib.check(_lib.load().fr_fisher_workspace_layout(P, W, H, V, cap, 4, off), "layout")
    ws = fixed._ws[0]
    cnt = ws[off[0]:off[0] + V * T * 4].view(torch.int32).cpu().numpy().astype(np.int64)
    toff = ws[off[1]:off[1] + V * T * 4].view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    keys = ws[off[2]:off[2] + cap * 8].view(torch.int64).cpu().numpy().view(np.uint64)
    assert cnt.max() > 2048 or P < 2400
    for i in range(V * T):
        seg = keys[toff[i]:toff[i] + cnt[i]]
        assert np.all(seg[1:] > seg[:-1]), (i, cnt[i])
        moved = toff[i] != i * capacity
        assert moved == (2048 < cnt[i] <= capacity // 2 - 64), (i, cnt[i], toff[i])      # (a list and its parts must both fit the segment)
        if moved:                                                   # the unsorted list still stands at the segment's start
            assert np.array_equal(np.sort(keys[i * capacity:i * capacity + cnt[i]]), seg)

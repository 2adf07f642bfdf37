"""oracle/occupancy_frontier.py -- TEST / BASELINE INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.

CPU restatement of the reference planner's occupancy-map update and frontier extraction, the step that sits either
side of view scoring in a planning round.  It is only TIMED (bench.py's `cpu_occupancy_frontier` entry)
as the reported host-core baseline the north star asks for; nothing on the product path uses it.

Restated from (paths relative to /root/reference):
  planning/astar.py:202-301   AstarPlanner.update_occ_map   (11 depth samples per pixel, unique-count binning into a
                              3 x grid x grid map at `cell_size`, Bresenham free line from every occupied cell to the camera)
  planning/astar.py:401-447   build_connected_freespace     (argmax label, Gaussian blocking with count > 25, 3x3 opening,
                              largest connected component)
  planning/astar.py:540-683   build_frontiers               (dilate - free AND unknown, dilate, components, min area 10,
                              "combined" score count / (mean distance + 20))
  datasets/util/map_utils.py:106-125  discretize_coords
cv2 (absent in this image) is replaced by scipy.ndimage with 8-connectivity (cv2's default), cv2's border rule for the
morphology (cells outside the image never constrain), and OpenCV's 8-connected LineIterator as cv::line drives it
(opencv/modules/imgproc/src/drawing.cpp, 4.x: left-to-right, count = major + 1, err = major - 2 minor), restated from the
published algorithm; torch.unique(dim=0, return_counts=True) by np.unique.  float32 throughout, `c2w @ pts` accumulated left
to right.  No golden vectors exist for this code in the reference and cv2 cannot be run here: parity unpinned.
This module is also the checker of the GPU kernels in fisher-nerf-customized_amd/csrc/fisher_occ.hip (tests only).
  planning/astar.py:1406-1430, 1432-1469  generate_candidate / generate_candidate_object   (ring_candidates below)
  planning/astar.py:1387-1401             the free-space filter of the candidate loop       (ring_candidates' `keep`)
  planning/astar.py:782-837               sample_random_candidate                           (free_candidates below)
  models/SLAM/utils/slam_external.py:25-42 build_rotation                                   (_yaw_rotation below)
The reference draws from torch.rand / numpy's default_rng; the product uses a counter-based generator (occ_uniform, defined
in csrc/fisher_occ.hip and restated here), so the restatement is the reference's arithmetic on those draws.
"""
import numpy as np
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=bool)


def discretize_coords(x, z, grid_dim, cell_size, map_center):
    x, z = np.asarray(x, dtype=np.float32), np.asarray(z, dtype=np.float32)
    cx, cz, cell = np.float32(map_center[0]), np.float32(map_center[1]), np.float32(cell_size)
    xb = np.floor((x - cx) / cell) + np.float32((grid_dim[0] - 1) / 2.0)
    zb = np.floor((z - cz) / cell) + np.float32((grid_dim[1] - 1) / 2.0)
    xb = np.clip(xb.astype(np.int32), 0, grid_dim[0] - 1)
    zb = np.clip(zb.astype(np.int32), 0, grid_dim[1] - 1)
    return np.stack([xb, zb], axis=1).astype(np.int64)


def _line(canvas, x0, y0, x1, y1):
    """cv2.line(canvas, (x0, y0), (x1, y1), 1, 1): OpenCV's 8-connected LineIterator, left to right."""
    if x1 < x0:
        x0, y0, x1, y1 = x1, y1, x0, y0
    dx, dy = x1 - x0, y1 - y0
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    err = major - 2 * minor
    h, w = canvas.shape
    x, y = x0, y0
    for _ in range(major + 1):
        if 0 <= y < h and 0 <= x < w:
            canvas[y, x] = 1
        both = err < 0
        err += -2 * minor + (2 * major if both else 0)
        if steep:
            y += sy
            x += 1 if both else 0
        else:
            x += 1
            y += sy if both else 0


def sample_fractions():
    """astar.py:236-238: torch.linspace(1e-3, 0.95, 11) clamped at 0, last entry set to 1 (the depth point itself)."""
    import torch
    f = torch.linspace(1e-3, 0.95, 11).clamp_(min=0.).numpy().astype(np.float32)
    f[-1] = 1.0
    return f


def _to_world(c2w, x, y, z):
    """c2w @ (x, y, z, 1) in float32, accumulated left to right."""
    c = c2w.astype(np.float32)
    return tuple(((c[r, 0] * x + c[r, 1] * y) + c[r, 2] * z) + c[r, 3] for r in range(3))


def _open3(a):
    er = ndimage.binary_erosion(a, structure=_EIGHT, border_value=1)
    return ndimage.binary_dilation(er, structure=_EIGHT, border_value=0)


class OccupancyMap:
    def __init__(self, intrinsics, grid_dim=(768, 768), cell_size=0.05, map_center=(0.0, 0.0), height_range=(-0.6, 0.6),
                 pcd_far_distance=10.0):
        self.K = np.asarray(intrinsics, dtype=np.float32)
        self.grid_dim = tuple(grid_dim)
        self.cell_size = float(cell_size)
        self.map_center = np.asarray(map_center, dtype=np.float32)
        self.height_lower, self.height_upper = height_range
        self.pcd_far_distance = pcd_far_distance
        self.occ_map = np.zeros((3, grid_dim[1], grid_dim[0]), dtype=np.float32)   # 0 unknown, 1 occupied, 2 free
        self.occ_map[0] = 1.0                     # planning/astar.py:94-96: every cell starts "unknown"
        self.cam_pos = np.array([grid_dim[1] // 2, grid_dim[0] // 2])

    # planning/astar.py:202-301
    def update_occ_map(self, depth, c2w, downsample=1):
        c2w = np.asarray(c2w, dtype=np.float32)
        cam_x, cam_z = c2w[0, 3], c2w[2, 3]
        cam_pos_x = int((cam_x - self.map_center[0]) / self.cell_size + self.grid_dim[0] // 2)
        cam_pos_z = int((cam_z - self.map_center[1]) / self.cell_size + self.grid_dim[1] // 2)
        self.cam_pos = np.array([cam_pos_z, cam_pos_x])
        self.occ_map[2, cam_pos_z - 1:cam_pos_z + 2, cam_pos_x - 1:cam_pos_x + 2] = 1e3
        depth = np.asarray(depth, dtype=np.float32)            # (1, H, W)
        height, width = depth.shape[1], depth.shape[2]
        CX, CY, FX, FY = self.K[0, 2], self.K[1, 2], self.K[0, 0], self.K[1, 1]
        xg, yg = np.meshgrid(np.arange(0, width, downsample, dtype=np.float32), np.arange(0, height, downsample, dtype=np.float32))
        xx, yy = ((xg - CX) / FX)[None], ((yg - CY) / FY)[None]
        sampled_z = sample_fractions().reshape(-1, 1, 1)               # (K, 1, 1): the last sample is the depth point, every pixel
        depth_z = sampled_z * depth[:, ::downsample, ::downsample]
        mask = (depth_z > 0) & (depth_z < self.pcd_far_distance)
        px, py = xx * depth_z, yy * depth_z
        grid = np.zeros((3, self.grid_dim[1], self.grid_dim[0]), dtype=np.float32)
        occ_map = np.zeros_like(self.occ_map)

        m = mask[:-1].reshape(-1)
        wx, wy, wz = _to_world(c2w, px[:-1].reshape(-1)[m], py[:-1].reshape(-1)[m], depth_z[:-1].reshape(-1)[m])
        mc = discretize_coords(wx, wz, self.grid_dim, self.cell_size, self.map_center)
        valid = (wy >= np.float32(self.height_lower)) & (wy <= np.float32(self.height_upper))
        uv, counts = np.unique(mc[valid], axis=0, return_counts=True)
        grid[2, uv[:, 1], uv[:, 0]] = counts.astype(np.float32) + np.float32(1e-5)
        occ_map += np.float32(0.01) * grid

        grid[:] = 0.0
        m = mask[-1].reshape(-1)
        wx, wy, wz = _to_world(c2w, px[-1].reshape(-1)[m], py[-1].reshape(-1)[m], depth_z[-1].reshape(-1)[m])
        valid = (wy >= np.float32(self.height_lower)) & (wy <= np.float32(self.height_upper))
        mc = discretize_coords(wx, wz, self.grid_dim, self.cell_size, self.map_center)
        uv, counts = np.unique(mc[valid], axis=0, return_counts=True)
        grid[1, uv[:, 1], uv[:, 0]] = counts.astype(np.float32) + np.float32(1e-5)
        grid[1] *= np.float32(100)
        occ_map += grid

        line_canvas = np.zeros((self.grid_dim[1], self.grid_dim[0]), dtype=np.uint8)
        for x, z in uv:
            _line(line_canvas, int(x), int(z), cam_pos_x, cam_pos_z)
        fz, fx = np.where(line_canvas > 0)
        occ_map[2, fz, fx] = 1.0
        self.occ_map += occ_map / (((occ_map[0] + occ_map[1]) + occ_map[2])[None] + np.float32(1e-5))

    # planning/astar.py:401-447
    def build_connected_freespace(self, gaussian_points=None):
        index = self.occ_map.argmax(axis=0)
        free_space = (index == 2)
        if free_space.sum() > 18 and gaussian_points is not None:
            g = np.asarray(gaussian_points)
            sel = g[(g[:, 1] >= self.height_lower) & (g[:, 1] <= self.height_upper)]
            mc = discretize_coords(sel[:, 0], sel[:, 2], self.grid_dim, self.cell_size, self.map_center)
            uv, counts = np.unique(mc, axis=0, return_counts=True)
            uv = uv[counts > 25]
            free_space[uv[:, 1], uv[:, 0]] = 0
        free_space = _open3(free_space)
        labels, n = ndimage.label(free_space, structure=_EIGHT)
        if n == 0:
            return np.zeros_like(free_space, dtype=np.uint8)
        sizes = np.bincount(labels.ravel())
        sizes[0] = 0
        return (labels == sizes.argmax()).astype(np.uint8)

    # planning/astar.py:540-683
    def build_frontiers(self, gaussian_points=None, method="combined", min_area=10, details=None):
        free_space = self.build_connected_freespace(gaussian_points)
        unknown = (self.occ_map.argmax(axis=0) == 0)
        dil = ndimage.binary_dilation(free_space.astype(bool), structure=_EIGHT)
        boundary = dil.astype(np.uint8) - free_space
        frontier = (boundary.astype(bool) & unknown)
        if details is not None:
            details["frontier"] = frontier.astype(np.uint8)
        if frontier.sum() == 0:
            return None, free_space
        frontier = ndimage.binary_dilation(frontier, structure=_EIGHT)
        labels, n = ndimage.label(frontier, structure=_EIGHT)
        counts = np.bincount(labels.ravel())[1:]
        lab = np.arange(1, n + 1)[counts > min_area]
        counts = counts[counts > min_area]
        if details is not None:
            details["components"] = int(len(lab))
        if len(lab) == 0:
            return None, free_space
        best = -1
        if method == "largest":
            best = lab[np.argsort(counts, kind="stable")[::-1][0]]
        elif method == "combined":
            best_score = 0.0
            for l, c in zip(lab, counts):
                pos = np.stack(np.where(labels == l), axis=1)
                if len(pos) < 4:
                    continue
                score = c / (np.linalg.norm(pos - self.cam_pos, axis=1).mean() + 20)
                if score > best_score:
                    best, best_score = l, score
        elif method == "closest":
            best_d = 1e4
            for l in lab:
                pos = np.stack(np.where(labels == l), axis=1)
                if len(pos) < 4:
                    continue
                d = np.linalg.norm(pos - self.cam_pos, axis=1).mean()
                if d < best_d:
                    best, best_d = l, d
        else:
            raise ValueError(method)
        if best == -1:
            return None, free_space
        if details is not None:
            details["target"] = (labels == best).astype(np.uint8)
        px = np.stack(np.where(labels == best), axis=1)[:, [1, 0]]
        pts = (px - np.array([[self.grid_dim[0] // 2, self.grid_dim[1] // 2]])) * self.cell_size + self.map_center[None]
        return pts, free_space


def room_depth(c2w, W, H, K, half=(5.0, 1.25, 5.0)):
    """Analytic depth image (z along the optical axis) of the axis-aligned synthetic room seen from c2w."""
    c2w = np.asarray(c2w, dtype=np.float64)
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    d_cam = np.stack([(xs - K[0][2]) / K[0][0], (ys - K[1][2]) / K[1][1], np.ones_like(xs, dtype=np.float64)], axis=-1)
    d_w = d_cam @ c2w[:3, :3].T
    o = c2w[:3, 3]
    t = np.full(d_w.shape[:2], np.inf)
    for a in range(3):
        for s in (-1.0, 1.0):
            with np.errstate(divide="ignore", invalid="ignore"):
                ta = (s * half[a] - o[a]) / d_w[..., a]
            with np.errstate(invalid="ignore"):
                hit = o[None, None, :] + ta[..., None] * d_w
            ok = (ta > 1e-6)
            for b in range(3):
                if b != a:
                    ok &= np.abs(hit[..., b]) <= half[b] + 1e-9
            t = np.where(ok & (ta < t), ta, t)
    return t.astype(np.float32)[None]      # (1, H, W): ray parameter along d_cam with z = 1  ==  depth z


def time_baseline(n_frames=6, W=256, H=256, seed=2, n_gaussians=200_000):
    """ms per occupancy update and per frontier build on synthetic frames of the benchmark room (1 host core)."""
    import time
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fisher-nerf-customized_amd"))
    from fisher_rast import synthetic
    K = synthetic.intrinsics(W, H)
    poses = synthetic.candidate_poses(n_frames, seed + 200).numpy()
    pts = synthetic.room_shell(n_gaussians, seed)["means3D"].numpy()
    m = OccupancyMap(K)
    depths = [room_depth(p, W, H, K) for p in poses]
    t0 = time.perf_counter()
    for p, d in zip(poses, depths):
        m.update_occ_map(d, p)
    t_up = (time.perf_counter() - t0) / n_frames
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        fr, free = m.build_frontiers(pts)
    t_fr = (time.perf_counter() - t0) / reps
    return dict(ms_per_update=1e3 * t_up, ms_per_frontier_build=1e3 * t_fr, frames=n_frames, cores=1, kind="port",
                free_cells=int(free.sum()), frontier_cells=0 if fr is None else int(len(fr)),
                sample=f"{n_frames} synthetic {W}x{H} depth frames of the benchmark room, 768x768x3 grid at 0.05 m; "
                       f"restated from planning/astar.py:202-301,401-447,540-683 with scipy.ndimage in place of cv2")


if __name__ == "__main__":
    print(time_baseline())


# ---------------------------------------------------------------------------------------------------------------------
# candidate samplers
# ---------------------------------------------------------------------------------------------------------------------
def occ_uniform(seed, k, j):
    """The product's counter-based generator (csrc/fisher_occ.hip: occ_uniform): u in [0, 1) as float32."""
    k = np.asarray(k, dtype=np.uint64)
    x = (np.uint64(seed) + np.uint64(0x9E3779B9) * (np.uint64(4) * k + np.uint64(j) + np.uint64(1))) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return ((x >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def _yaw_rotation(qr, qy):
    """slam_external.py:25-42 for the quaternion (qr, 0, qy, 0): normalise, then the matrix entries, float32."""
    qr, qy = np.asarray(qr, np.float32), np.asarray(qy, np.float32)
    norm = np.sqrt(qr * qr + qy * qy, dtype=np.float32)
    r, y = qr / norm, qy / norm
    R = np.zeros((qr.shape[0], 3, 3), np.float32)
    R[:, 0, 0] = np.float32(1) - np.float32(2) * (y * y)
    R[:, 0, 2] = np.float32(2) * (r * y)
    R[:, 1, 1] = 1
    R[:, 2, 0] = -(np.float32(2) * (r * y))
    R[:, 2, 2] = np.float32(1) - np.float32(2) * (y * y)
    return R


def ring_candidates(centers, K, min_range, radius, cam_height, seed, eroded=None, min_free=40, grid_dim=(768, 768),
                    cell_size=0.05, map_center=(0.0, 0.0)):
    """astar.py:1406-1430 on the draws of occ_uniform, float32: returns (c2w [K,4,4], keep [K] bool).
    keep follows astar.py:1387-1401: eroded[row, col] at the truncated cell of the pose, applied only when eroded.sum() > min_free."""
    f = np.float32
    centers = np.asarray(centers, np.float32)
    k = np.arange(K)
    pi = f(np.pi)
    theta = (occ_uniform(seed, k, 0) * f(2)) * pi
    rr = f(min_range) + occ_uniform(seed, k, 1) * (f(radius) - f(min_range))
    ci = np.minimum((occ_uniform(seed, k, 2) * f(centers.shape[0])).astype(np.int64), centers.shape[0] - 1)
    pos = np.zeros((K, 3), np.float32)
    pos[:, 0] = centers[ci, 0] + rr * np.sin(theta, dtype=np.float32)
    pos[:, 1] = f(cam_height)
    pos[:, 2] = centers[ci, 1] + rr * np.cos(theta, dtype=np.float32)
    phi = theta + pi
    R = _yaw_rotation(np.cos(phi / f(2), dtype=np.float32), np.sin(phi / f(2), dtype=np.float32))
    R[:, :, 0] *= -1
    R[:, :, 1] *= -1
    c2w = np.zeros((K, 4, 4), np.float32)
    c2w[:, :3, :3] = R
    c2w[:, :3, 3] = pos
    c2w[:, 3, 3] = 1
    keep = np.ones(K, bool)
    if eroded is not None and int(np.asarray(eroded).sum()) > min_free:
        col = ((pos[:, 0] - f(map_center[0])) / f(cell_size) + f(grid_dim[0] // 2)).astype(np.int64)
        row = ((pos[:, 2] - f(map_center[1])) / f(cell_size) + f(grid_dim[1] // 2)).astype(np.int64)
        inside = (col >= 0) & (col < grid_dim[0]) & (row >= 0) & (row < grid_dim[1])
        keep = np.zeros(K, bool)
        keep[inside] = np.asarray(eroded)[row[inside], col[inside]] != 0
    return c2w, keep


def free_candidates(eroded, agent_y, seed, grid_dim=(768, 768), cell_size=0.05, map_center=(0.0, 0.0)):
    """astar.py:805-835 on the draws of occ_uniform: the eroded free cells in np.where order, len // 4 draws with replacement,
    cell-centre positions (float64 like numpy, narrowed once), uniform yaw, columns 1 and 2 negated."""
    rows, cols = np.where(np.asarray(eroded) == 1)
    n = len(rows)
    m = n // 4
    i = np.arange(m)
    ci = np.minimum((occ_uniform(seed, i, 0) * np.float32(n)).astype(np.int64), max(n - 1, 0))
    wz = (rows[ci] + 0.5 - grid_dim[1] // 2) * float(np.float32(cell_size)) + float(np.float32(map_center[1]))
    wx = (cols[ci] + 0.5 - grid_dim[0] // 2) * float(np.float32(cell_size)) + float(np.float32(map_center[0]))
    ang = occ_uniform(seed, i, 1).astype(np.float64) * 6.283185307179586
    R = _yaw_rotation(np.cos(ang / 2).astype(np.float32), np.sin(ang / 2).astype(np.float32))
    pose = np.zeros((m, 4, 4), np.float32)
    pose[:, :3, :3] = R
    pose[:, 0, 3] = wx.astype(np.float32); pose[:, 1, 3] = np.float32(agent_y); pose[:, 2, 3] = wz.astype(np.float32)
    pose[:, 3, 3] = 1
    pose[:, :, 1] *= -1
    pose[:, :, 2] *= -1
    return pose


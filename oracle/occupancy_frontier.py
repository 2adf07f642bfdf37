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
cv2 (absent in this image) is replaced by scipy.ndimage with 8-connectivity (cv2's default) and a NumPy Bresenham;
torch.unique(dim=0, return_counts=True) by np.unique.  No golden vectors exist for this code in the reference; parity unpinned.
"""
import numpy as np
from scipy import ndimage

_EIGHT = np.ones((3, 3), dtype=bool)


def discretize_coords(x, z, grid_dim, cell_size, map_center):
    xb = np.floor((x - map_center[0]) / cell_size) + (grid_dim[0] - 1) / 2.0
    zb = np.floor((z - map_center[1]) / cell_size) + (grid_dim[1] - 1) / 2.0
    xb = np.clip(xb.astype(np.int32), 0, grid_dim[0] - 1)
    zb = np.clip(zb.astype(np.int32), 0, grid_dim[1] - 1)
    return np.stack([xb, zb], axis=1).astype(np.int64)


def _line(canvas, x0, y0, x1, y1):
    """Bresenham, 1-pixel wide (cv2.line(..., color=1, thickness=1))."""
    dx, dy = abs(x1 - x0), -abs(y1 - y0)
    sx, sy = (1 if x0 < x1 else -1), (1 if y0 < y1 else -1)
    err = dx + dy
    h, w = canvas.shape
    while True:
        if 0 <= y0 < h and 0 <= x0 < w:
            canvas[y0, x0] = 1
        if x0 == x1 and y0 == y1:
            break
        e2 = 2 * err
        if e2 >= dy:
            err += dy
            x0 += sx
        if e2 <= dx:
            err += dx
            y0 += sy


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
        sampled_z = np.linspace(1e-3, 0.95, 11, dtype=np.float32).reshape(-1, 1, 1) * np.ones((1, xx.shape[1], xx.shape[2]), np.float32)
        sampled_z = np.clip(sampled_z, 0.0, None)
        sampled_z[-1, 0, 0] = 1.0
        depth_z = sampled_z * depth[:, ::downsample, ::downsample]
        mask = (depth_z > 0) & (depth_z < self.pcd_far_distance)
        pts = np.stack((xx * depth_z, yy * depth_z, depth_z, np.ones_like(depth_z)), axis=0)      # 4 x K x H x W
        free_particles = pts[:, :-1].reshape(4, -1)[:, mask[:-1].reshape(-1)]
        depth_pts = pts[:, -1].reshape(4, -1)[:, mask[-1].reshape(-1)]
        grid = np.zeros((3, self.grid_dim[1], self.grid_dim[0]), dtype=np.float32)
        occ_map = np.zeros_like(self.occ_map)

        free_particles = c2w @ free_particles
        mc = discretize_coords(free_particles[0], free_particles[2], self.grid_dim, self.cell_size, self.map_center)
        valid = (free_particles[1] >= self.height_lower) & (free_particles[1] <= self.height_upper)
        uv, counts = np.unique(mc[valid], axis=0, return_counts=True)
        grid[2, uv[:, 1], uv[:, 0]] = counts + 1e-5
        occ_map += 0.01 * grid

        grid[:] = 0.0
        depth_pts = c2w @ depth_pts
        valid = (depth_pts[1] >= self.height_lower) & (depth_pts[1] <= self.height_upper)
        mc = discretize_coords(depth_pts[0], depth_pts[2], self.grid_dim, self.cell_size, self.map_center)
        uv, counts = np.unique(mc[valid], axis=0, return_counts=True)
        grid[1, uv[:, 1], uv[:, 0]] = counts + 1e-5
        grid[1] *= 100
        occ_map += grid

        line_canvas = np.zeros((self.grid_dim[1], self.grid_dim[0]), dtype=np.uint8)
        for x, z in uv:
            _line(line_canvas, int(x), int(z), cam_pos_x, cam_pos_z)
        fz, fx = np.where(line_canvas > 0)
        occ_map[2, fz, fx] = 1.0
        self.occ_map += occ_map / (occ_map.sum(axis=0, keepdims=True) + 1e-5)

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
        free_space = ndimage.binary_opening(free_space, structure=_EIGHT)
        labels, n = ndimage.label(free_space, structure=_EIGHT)
        if n == 0:
            return np.zeros_like(free_space, dtype=np.uint8)
        sizes = np.bincount(labels.ravel())
        sizes[0] = 0
        return (labels == sizes.argmax()).astype(np.uint8)

    # planning/astar.py:540-683 ("combined" selection)
    def build_frontiers(self, gaussian_points=None):
        free_space = self.build_connected_freespace(gaussian_points)
        unknown = (self.occ_map.argmax(axis=0) == 0)
        dil = ndimage.binary_dilation(free_space.astype(bool), structure=_EIGHT)
        boundary = dil.astype(np.uint8) - free_space
        frontier = (boundary.astype(bool) & unknown)
        if frontier.sum() == 0:
            return None, free_space
        frontier = ndimage.binary_dilation(frontier, structure=_EIGHT)
        labels, n = ndimage.label(frontier, structure=_EIGHT)
        counts = np.bincount(labels.ravel())[1:]
        lab = np.arange(1, n + 1)[counts > 10]
        counts = counts[counts > 10]
        if len(lab) == 0:
            return None, free_space
        best, best_score = -1, 0.0
        for l, c in zip(lab, counts):
            pos = np.stack(np.where(labels == l), axis=1)
            if len(pos) < 4:
                continue
            score = c / (np.linalg.norm(pos - self.cam_pos, axis=1).mean() + 20)
            if score > best_score:
                best, best_score = l, score
        if best == -1:
            return None, free_space
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

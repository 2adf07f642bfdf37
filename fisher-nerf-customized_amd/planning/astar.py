"""`AstarPlanner`'s occupancy-map and frontier operators on MI355X (reference: planning/astar.py).

In scope (same names, arguments and return conventions as the reference):
    init (66-103)                      update_occ_map (202-301)        build_connected_freespace (401-447)
    build_frontiers (540-683)          generate_candidate (1406-1430)  the free-space filter of the candidate loop (1383-1401)
The A* search, path shortcutting, visualisation and the VLM frontier selection are NOT rebuilt: they are reference
Python that stays as it is.  `OccupancyOps.install(cls)` grafts the accelerated methods onto the reference class;
`AstarPlanner` below is the same operator surface as a standalone object for tests and benchmarks.

What changes underneath: the reference bins the depth samples with torch ops and then walks every occupied cell in a
Python loop on the host (one `cv2.line` each, astar.py:291-297), and runs `cv2` morphology / connected components on the
CPU after copying the map down.  Here the whole step is a handful of HIP kernels on the resident map
(fisher_occ.h: fr_occ_update / fr_occ_freespace / fr_occ_frontiers); the only host traffic is the selected frontier's cells.
"""
import ctypes
import math

import numpy as np
import torch

from fisher_rast import _lib

_METHODS = {"largest": 0, "combined": 1, "closest": 2}


def build_rotation(q):
    """models/SLAM/utils/slam_external.py:25-42 (wxyz quaternion -> rotation matrix), device-agnostic."""
    norm = torch.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    q = q / norm[:, None]
    rot = torch.zeros((q.size(0), 3, 3), device=q.device)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rot[:, 0, 0] = 1 - 2 * (y * y + z * z)
    rot[:, 0, 1] = 2 * (x * y - r * z)
    rot[:, 0, 2] = 2 * (x * z + r * y)
    rot[:, 1, 0] = 2 * (x * y + r * z)
    rot[:, 1, 1] = 1 - 2 * (x * x + z * z)
    rot[:, 1, 2] = 2 * (y * z - r * x)
    rot[:, 2, 0] = 2 * (x * z - r * y)
    rot[:, 2, 1] = 2 * (y * z + r * x)
    rot[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return rot


class OccupancyOps:
    """Mixin with the accelerated occupancy / frontier methods.  Needs the attributes AstarPlanner.__init__ sets:
    device, cell_size, height_lower, height_upper, pcd_far_distance, frontier_select_method, K, radius, min_range."""

    # -- internals -----------------------------------------------------------------------------------------------
    def _occ_cfg(self):
        mc = self.map_center.detach().cpu().numpy() if isinstance(self.map_center, torch.Tensor) else np.asarray(self.map_center)
        return _lib.OccCfg(int(self.grid_dim[0]), int(self.grid_dim[1]), float(self.cell_size), float(np.float32(mc[0])),
                           float(np.float32(mc[1])), float(self.height_lower), float(self.height_upper), float(self.pcd_far_distance))

    def _occ_workspace(self, cfg):
        lib = _lib.load()
        need = lib.fr_occ_workspace_bytes(ctypes.byref(cfg))
        ws = getattr(self, "_occ_ws", None)
        if ws is None or ws.numel() < need or ws.device != self.occ_map.device:
            ws = torch.empty((need,), dtype=torch.uint8, device=self.occ_map.device)
            self._occ_ws = ws
        return ws, need

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # -- astar.py:66-103 ---------------------------------------------------------------------------------------------
    def init(self, pose, intrinsic, scene_bounds=None):
        pose = pose.detach().cpu().numpy() if isinstance(pose, torch.Tensor) else np.asarray(pose)
        self.grid_dim = np.array([768, 768])
        self.intrinsics = intrinsic
        self.cam_height = float(pose[1, 3])
        if scene_bounds is not None:
            self.scene_bounds = scene_bounds
            scene_lower, scene_upper = scene_bounds
            map_center_np = (scene_upper[[0, 2]] + scene_lower[[0, 2]]) / 2
            grid_x = (scene_upper[0] - scene_lower[0]) / self.cell_size
            grid_z = (scene_upper[2] - scene_lower[2]) / self.cell_size
            self.grid_dim = np.array([int(grid_x + 1), int(grid_z + 1)])
        else:
            map_center_np = pose[[0, 2], 3]
        self.occ_map = torch.zeros((3, int(self.grid_dim[1]), int(self.grid_dim[0])), device=self.device)
        self.occ_map[0] = 1.
        cam_pos_x = int((pose[0, 3] - map_center_np[0]) / self.cell_size + self.grid_dim[0] // 2)
        cam_pos_z = int((pose[2, 3] - map_center_np[1]) / self.cell_size + self.grid_dim[1] // 2)
        self.cam_pos = np.array([cam_pos_z, cam_pos_x])
        self.occ_map[2, cam_pos_z - 1:cam_pos_z + 2, cam_pos_x - 1:cam_pos_x + 2] = 2.
        self.map_center = torch.from_numpy(np.asarray(map_center_np)).to(self.device)
        self.frame_idx = 0

    # -- astar.py:202-301 --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def update_occ_map(self, depth, c2w, t, downsample=1):
        """ Update Occulision map based on depth observation """
        lib = _lib.load()
        self.frame_idx = t
        c2w_h = c2w.detach().float().cpu() if isinstance(c2w, torch.Tensor) else torch.as_tensor(np.asarray(c2w)).float()
        mc = self.map_center.detach().cpu()
        cam_x, cam_z = c2w_h[0, 3], c2w_h[2, 3]
        cam_pos_x = int((cam_x - mc[0]) / self.cell_size + self.grid_dim[0] // 2)
        cam_pos_z = int((cam_z - mc[1]) / self.cell_size + self.grid_dim[1] // 2)
        self.cam_pos = np.array([cam_pos_z, cam_pos_x])
        if isinstance(depth, np.ndarray):
            depth = torch.from_numpy(depth)
        depth = depth.to(self.device).float().contiguous()
        width, height = depth.shape[2], depth.shape[1]
        K = self.intrinsics
        intr = (ctypes.c_float * 4)(float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]))
        c2w_a = (ctypes.c_float * 16)(*[float(v) for v in c2w_h.reshape(-1)])
        fr = torch.linspace(1e-3, 0.95, 11).clamp_(min=0.)
        fr[-1] = 1.
        fracs = (ctypes.c_float * 11)(*[float(v) for v in fr])
        cfg = self._occ_cfg()
        ws, need = self._occ_workspace(cfg)
        _lib.check(lib.fr_occ_update(ctypes.byref(cfg), depth.data_ptr(), width, height, int(downsample), ctypes.byref(intr),
                                     ctypes.byref(c2w_a), fracs, 11, cam_pos_x, cam_pos_z, self.occ_map.data_ptr(),
                                     ws.data_ptr(), need, self._stream()), "fr_occ_update")

    # -- astar.py:401-447 --------------------------------------------------------------------------------------------
    def build_connected_freespace(self, gaussian_points=None, as_tensor=False):
        """ find the connected free space to the robot: (H, W) uint8, 1 - free, 0 - occupied
        (np.ndarray like the reference, or the resident tensor with as_tensor=True) """
        lib = _lib.load()
        cfg = self._occ_cfg()
        ws, need = self._occ_workspace(cfg)
        free = torch.empty((int(self.grid_dim[1]), int(self.grid_dim[0])), dtype=torch.uint8, device=self.occ_map.device)
        pts, n = None, 0
        if gaussian_points is not None:
            pts_t = gaussian_points.detach().to(self.occ_map.device).float().contiguous()
            pts, n = pts_t.data_ptr(), int(pts_t.shape[0])
        _lib.check(lib.fr_occ_freespace(ctypes.byref(cfg), self.occ_map.data_ptr(), pts, n, free.data_ptr(), ws.data_ptr(), need,
                                        self._stream()), "fr_occ_freespace")
        self._free_space_dev = free
        return free if as_tensor else free.cpu().numpy()

    # -- astar.py:540-683 --------------------------------------------------------------------------------------------
    def build_frontiers(self, gaussian_points=None):
        """ Return frontiers in pixel space  """
        lib = _lib.load()
        free_dev = self.build_connected_freespace(gaussian_points, as_tensor=True)
        cfg = self._occ_cfg()
        ws, need = self._occ_workspace(cfg)
        gh, gw = int(self.grid_dim[1]), int(self.grid_dim[0])
        dev = self.occ_map.device
        frontier = torch.empty((gh, gw), dtype=torch.uint8, device=dev)
        target = torch.empty((gh, gw), dtype=torch.uint8, device=dev)
        max_cells = gh * gw
        cells = getattr(self, "_occ_cells", None)
        if cells is None or cells.shape[0] < max_cells or cells.device != dev:
            cells = torch.empty((max_cells, 2), dtype=torch.int32, device=dev)
            self._occ_cells = cells
        counts = torch.zeros((4,), dtype=torch.int32, device=dev)
        method = self.frontier_select_method
        if method not in _METHODS:
            raise ValueError(f"frontier_select_method {method!r} is not one of {sorted(_METHODS)} (the VLM selection stays reference Python)")
        _lib.check(lib.fr_occ_frontiers(ctypes.byref(cfg), self.occ_map.data_ptr(), free_dev.data_ptr(), int(self.cam_pos[0]),
                                        int(self.cam_pos[1]), _METHODS[method], 10, frontier.data_ptr(), target.data_ptr(),
                                        cells.data_ptr(), max_cells, counts.data_ptr(), ws.data_ptr(), need, self._stream()),
                   "fr_occ_frontiers")
        n_frontier, n_comp, n_target, _ = [int(v) for v in counts.cpu()]          # the one host sync of the step
        free_space = free_dev.cpu().numpy()
        self.frontier = frontier.cpu().numpy()
        if n_frontier == 0:
            self.target_frontier = None
            return None, free_space
        if n_comp == 0 or n_target == 0:
            return None, free_space
        if method == "largest":
            self.selection = 0
        self.target_frontier = target.cpu().numpy()
        map_center = self.map_center.cpu().numpy()
        select_pixels = cells[:n_target].cpu().numpy().astype(np.int64)              # (col, row) in np.where order
        select_pixels = (select_pixels - np.array([[self.grid_dim[0] // 2, self.grid_dim[1] // 2]])) * self.cell_size + map_center[None, :]
        if gaussian_points is None:                                                  # FBE logic, astar.py:655-679
            agent_pos = self.cam_pos
            min_thresh = 0.5
            distances = np.linalg.norm(select_pixels - agent_pos[None, :], axis=1)
            valid_idx = np.where(distances >= min_thresh)[0]
            if len(valid_idx) > 0:
                best_idx = valid_idx[np.argmin(distances[valid_idx])]
                frontier_point = select_pixels[best_idx:best_idx + 1]
            else:
                angle = math.pi * 5 / 4
                x, y = math.cos(angle), math.sin(angle)
                frontier_point = agent_pos[None, :] + np.array([[-x, -y]]) * 0.5
        else:
            frontier_point = select_pixels
        return frontier_point, free_space

    # -- astar.py:1406-1430 ------------------------------------------------------------------------------------------
    def generate_candidate(self, center_point: torch.Tensor, expansion=1):
        """ sample camera poses from the center point (K, 3) """
        dev = self.occ_map.device
        K, radius = self.K, self.radius * expansion
        theta = torch.rand((K,), device=dev) * 2 * torch.pi
        random_radius = self.min_range + torch.rand((K,), device=dev) * (radius - self.min_range)
        center_point = center_point.to(dev)
        center_point_height = torch.ones((center_point.shape[0],), device=dev) * float(self.cam_height)
        center_point = torch.stack([center_point[:, 0], center_point_height, center_point[:, 1]], dim=1)
        center_point = center_point[torch.randint(0, center_point.shape[0], (K,), device=dev)]
        cam_pos = torch.zeros((K, 3), device=dev)
        cam_pos[:, 0] = center_point[:, 0] + random_radius * torch.sin(theta)
        cam_pos[:, 1] = float(self.cam_height)
        cam_pos[:, 2] = center_point[:, 2] + random_radius * torch.cos(theta)
        cam_rot = torch.zeros((K, 4), device=dev)
        theta = theta + torch.pi
        cam_rot[:, 0] = torch.cos(theta / 2)
        cam_rot[:, 2] = torch.sin(theta / 2)
        cam_R = build_rotation(cam_rot)
        cam_R[:, :, 0] *= -1
        cam_R[:, :, 1] *= -1
        c2ws = torch.zeros((K, 4, 4), device=dev)
        c2ws[:, :3, 3] = cam_pos
        c2ws[:, :3, :3] = cam_R
        c2ws[:, 3, 3] = 1.
        return c2ws

    # -- astar.py:1387-1401: keep the candidates whose cell lies in the free space eroded by a 10 x 10 box ------------
    def filter_candidates_in_freespace(self, candidate_pose, free_space=None, ksize=10, min_free=40):
        lib = _lib.load()
        dev = self.occ_map.device
        if free_space is None:
            free_dev = self._free_space_dev
        else:
            free_dev = (free_space if isinstance(free_space, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(free_space))).to(dev).to(torch.uint8).contiguous()
        cfg = self._occ_cfg()
        eroded = torch.empty_like(free_dev)
        _lib.check(lib.fr_occ_erode(ctypes.byref(cfg), free_dev.data_ptr(), eroded.data_ptr(), int(ksize), self._stream()), "fr_occ_erode")
        if int(eroded.sum()) <= min_free:
            return candidate_pose
        mc = self.map_center.to(dev)
        candidate_xy = candidate_pose[:, [0, 2], 3].clone()
        candidate_xy[:, 0] = (candidate_xy[:, 0] - mc[0]) / self.cell_size + self.grid_dim[0] // 2
        candidate_xy[:, 1] = (candidate_xy[:, 1] - mc[1]) / self.cell_size + self.grid_dim[1] // 2
        candidate_xy = candidate_xy.long()
        free_pose = eroded[candidate_xy[:, 1], candidate_xy[:, 0]].bool()
        return candidate_pose[free_pose]

    def cells_of(self, xyz):
        """discretize_coords (datasets/util/map_utils.py:106-125) of the x / z columns of an [n, 3] tensor -> int32 [n, 2] (col, row)."""
        lib = _lib.load()
        xyz = xyz.detach().to(self.occ_map.device).float().contiguous()
        out = torch.empty((xyz.shape[0], 2), dtype=torch.int32, device=xyz.device)
        cfg = self._occ_cfg()
        _lib.check(lib.fr_occ_cells_of(ctypes.byref(cfg), xyz.data_ptr(), int(xyz.shape[0]), out.data_ptr(), self._stream()), "fr_occ_cells_of")
        return out

    @classmethod
    def install(cls, planner_cls):
        """Graft the accelerated methods onto the reference's AstarPlanner class."""
        for name in ("update_occ_map", "build_connected_freespace", "build_frontiers", "generate_candidate",
                     "filter_candidates_in_freespace", "cells_of", "_occ_cfg", "_occ_workspace", "_stream"):
            setattr(planner_cls, name, cls.__dict__[name])
        return planner_cls


class AstarPlanner(OccupancyOps):
    """Standalone object with the reference constructor's configuration keys (astar.py:23-60)."""

    def __init__(self, slam_config=None, eval_dir=None, device=torch.device("cuda:0"), **kw):
        cfg = slam_config or {}
        ex, pol = cfg.get("explore", {}), cfg.get("policy", {})
        self.device = torch.device(device)
        self.cell_size = kw.get("cell_size", ex.get("cell_size", 0.05))
        self.height_upper = kw.get("height_upper", pol.get("height_upper", 0.6))
        self.height_lower = kw.get("height_lower", pol.get("height_lower", -0.6))
        self.K = kw.get("sample_view_num", ex.get("sample_view_num", 64))
        self.radius = kw.get("sample_range", ex.get("sample_range", 1.0))
        self.min_range = kw.get("min_range", ex.get("min_range", 0.2))
        self.frontier_select_method = kw.get("frontier_select_method", ex.get("frontier_select_method", "combined"))
        self.pcd_far_distance = kw.get("pcd_far_distance", pol.get("pcd_far_distance", 10.0))
        self.eval_dir = eval_dir
        self.cam_pos = None
        self.frame_idx = 0

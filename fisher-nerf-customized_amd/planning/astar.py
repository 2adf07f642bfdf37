"""`AstarPlanner`'s occupancy-map and frontier operators on MI355X (reference: planning/astar.py).

In scope (same names, arguments and return conventions as the reference):
    init (66-103)                      update_occ_map (202-301)        build_connected_freespace (401-447)
    build_frontiers (540-683)          generate_candidate (1406-1430)  generate_candidate_object (1432-1469)
    the free-space filter of the candidate loop (1383-1401)            sample_random_candidate (782-837)
The A* search, path shortcutting, visualisation and the VLM frontier selection are NOT rebuilt: they are reference
Python that stays as it is.  `OccupancyOps.install(cls)` grafts the accelerated methods onto the reference class;
`AstarPlanner` below is the same operator surface as a standalone object for tests and benchmarks.

What changes underneath: the reference bins the depth samples with torch ops and then walks every occupied cell in a
Python loop on the host (one `cv2.line` each, astar.py:291-297), and runs `cv2` morphology / connected components on the
CPU after copying the map down.  Here the whole step is a handful of HIP kernels on the resident map
(fisher_occ.h: fr_occ_update / fr_occ_freespace / fr_occ_frontiers); the only host traffic is the selected frontier's cells.
Candidate poses come from one kernel each (fr_occ_ring_candidates: centre pick, ring sample, pose, eroded-free-space
test; fr_occ_free_candidates: uniform poses in the free space) driven by a counter-based generator, so a call is
reproducible from its seed and has a NumPy restatement in oracle/occupancy_frontier.py to test against.
"""
import ctypes
import math

import numpy as np
import torch

from fisher_rast import _lib

_METHODS = {"largest": 0, "combined": 1, "closest": 2}


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
    def _cell_of(self, x, z):
        """(col, row) of a world position: int((v - centre) / cell + dim // 2), the planner's own rule (astar.py:93-94, 211-212)."""
        mc = self._map_center_np
        return (int((x - mc[0]) / self.cell_size + self.grid_dim[0] // 2), int((z - mc[1]) / self.cell_size + self.grid_dim[1] // 2))

    def init(self, pose, intrinsic, scene_bounds=None):
        """Fresh map: 768 x 768 cells centred on the first pose, or the scene's x-z extent at the fixed cell size; every cell
        unknown (layer 0 = 1) except the 3 x 3 block under the camera, which is free (layer 2 = 2)."""
        T = pose.detach().cpu().numpy() if isinstance(pose, torch.Tensor) else np.asarray(pose)
        self.intrinsics = intrinsic
        self.cam_height = float(T[1, 3])
        self.frame_idx = 0
        if scene_bounds is None:
            self.grid_dim = np.array([768, 768])
            centre = T[[0, 2], 3]
        else:
            self.scene_bounds = scene_bounds
            lo, hi = (np.asarray(b) for b in scene_bounds)
            extent = (hi - lo)[[0, 2]] / self.cell_size
            self.grid_dim = np.array([int(extent[0] + 1), int(extent[1] + 1)])
            centre = (hi[[0, 2]] + lo[[0, 2]]) / 2
        self._map_center_np = np.asarray(centre)
        self.map_center = torch.from_numpy(self._map_center_np).to(self.device)
        gw, gh = int(self.grid_dim[0]), int(self.grid_dim[1])
        occ = torch.zeros((3, gh, gw), device=self.device)
        occ[0].fill_(1.)
        col, row = self._cell_of(T[0, 3], T[2, 3])
        self.cam_pos = np.array([row, col])
        occ[2, row - 1:row + 2, col - 1:col + 2] = 2.
        self.occ_map = occ

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
        if gaussian_points is not None:
            return select_pixels, free_space
        return self._fbe_point(select_pixels), free_space

    def _fbe_point(self, select_xy, min_thresh=0.5):
        """Frontier-based exploration target (astar.py:655-679): the selected cell nearest to `self.cam_pos` among those at
        least min_thresh away, else a point half a unit from it along the fixed 5 pi / 4 direction.  The reference measures
        the distance between the cells' WORLD coordinates and cam_pos, which holds GRID indices (row, col); kept as it is."""
        agent = np.asarray(self.cam_pos, dtype=np.float64)
        dist = np.sqrt(((select_xy - agent[None, :]) ** 2).sum(axis=1))
        far_enough = dist >= min_thresh
        if far_enough.any():
            pick = int(np.argmin(np.where(far_enough, dist, np.inf)))
            return select_xy[pick:pick + 1]
        ang = math.pi * 5 / 4
        return agent[None, :] - 0.5 * np.array([[math.cos(ang), math.sin(ang)]])

    # -- astar.py:1406-1430, 1432-1469 -------------------------------------------------------------------------------
    def _next_seed(self):
        """One 32-bit seed per call: from `self.candidate_seed` (an int, incremented) when set, else from torch's global CPU
        generator -- so `torch.manual_seed` makes a planning round reproducible, as it does for the reference's torch.rand."""
        fixed = getattr(self, "candidate_seed", None)
        if fixed is not None:
            self.candidate_seed = (int(fixed) + 1) & 0xFFFFFFFF
            return int(fixed) & 0xFFFFFFFF
        return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())

    def _ring_candidates(self, center_point, K, min_range, radius, eroded=None, min_free=40, seed=None):
        """K poses around the (x, z) rows of center_point (fr_occ_ring_candidates): ([K,4,4] c2w, [K] bool keep)."""
        lib = _lib.load()
        dev = self.occ_map.device
        centers = center_point.detach().to(dev).float()[:, :2].contiguous()
        c2w = torch.empty((int(K), 4, 4), dtype=torch.float32, device=dev)
        keep = torch.empty((int(K),), dtype=torch.uint8, device=dev)
        cfg = self._occ_cfg()
        seed = self._next_seed() if seed is None else int(seed) & 0xFFFFFFFF
        self.last_candidate_seed = seed
        _lib.check(lib.fr_occ_ring_candidates(ctypes.byref(cfg), centers.data_ptr(), int(centers.shape[0]), int(K), float(min_range),
                                              float(radius), float(self.cam_height), seed,
                                              None if eroded is None else eroded.data_ptr(), int(min_free), c2w.data_ptr(),
                                              keep.data_ptr(), self._stream()), "fr_occ_ring_candidates")
        return c2w, keep.bool()

    def generate_candidate(self, center_point: torch.Tensor, expansion=1, seed=None):
        """ sample camera poses from the center point (K, 4, 4) """
        return self._ring_candidates(center_point, self.K, self.min_range, self.radius * expansion, seed=seed)[0]

    def generate_candidate_object(self, center_point: torch.Tensor, expansion=1, seed=None):
        """ the object-centric ring: K_object poses between min_range_object and radius_object * expansion """
        return self._ring_candidates(center_point, self.K_object, self.min_range_object, self.radius_object * expansion, seed=seed)[0]

    def generate_candidate_in_freespace(self, center_point, free_space=None, expansion=1, ksize=10, min_free=40, seed=None):
        """generate_candidate + the free-space filter of the candidate loop (astar.py:1383-1401) in ONE launch: the poses
        whose cell lies in the free space eroded by a ksize x ksize box (all of them when that has no more than min_free cells)."""
        eroded = self._eroded_free(free_space, ksize)
        c2w, keep = self._ring_candidates(center_point, self.K, self.min_range, self.radius * expansion, eroded, min_free, seed)
        return c2w[keep]

    # -- astar.py:782-837 --------------------------------------------------------------------------------------------
    def sample_random_candidate(self, agent_pos, free_space, sample_range=1., sample_size: int = 100, seed=None):
        """ Randomly placed poses in the free space eroded by 11 x 11 (a quarter as many as it has cells), at the agent's height.
        sample_range / sample_size are accepted and unused, as in the reference. """
        lib = _lib.load()
        dev = self.occ_map.device
        eroded = self._eroded_free(free_space, 11)
        cfg = self._occ_cfg()
        ws, need = self._occ_workspace(cfg)
        max_out = (int(self.grid_dim[0]) * int(self.grid_dim[1])) // 4
        out = torch.empty((max_out, 4, 4), dtype=torch.float32, device=dev)
        counts = torch.zeros((2,), dtype=torch.int32, device=dev)
        seed = self._next_seed() if seed is None else int(seed) & 0xFFFFFFFF
        self.last_candidate_seed = seed
        y = float(agent_pos[1])
        _lib.check(lib.fr_occ_free_candidates(ctypes.byref(cfg), eroded.data_ptr(), y, seed, out.data_ptr(), max_out,
                                              counts.data_ptr(), ws.data_ptr(), need, self._stream()), "fr_occ_free_candidates")
        return out[:int(counts[1].item())]

    def _eroded_free(self, free_space, ksize):
        lib = _lib.load()
        dev = self.occ_map.device
        if free_space is None:
            free_dev = self._free_space_dev
        else:
            free_dev = (free_space if isinstance(free_space, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(free_space))).to(dev).to(torch.uint8).contiguous()
        eroded = torch.empty_like(free_dev)
        cfg = self._occ_cfg()
        _lib.check(lib.fr_occ_erode(ctypes.byref(cfg), free_dev.data_ptr(), eroded.data_ptr(), int(ksize), self._stream()), "fr_occ_erode")
        return eroded

    # -- astar.py:1387-1401: keep the candidates whose cell lies in the free space eroded by a 10 x 10 box ------------
    def filter_candidates_in_freespace(self, candidate_pose, free_space=None, ksize=10, min_free=40):
        """The filter alone, for poses that did not come from generate_candidate_in_freespace."""
        dev = self.occ_map.device
        eroded = self._eroded_free(free_space, ksize)
        if int(eroded.sum()) <= min_free:
            return candidate_pose
        mc = self.map_center.to(dev)
        col = ((candidate_pose[:, 0, 3] - mc[0]) / self.cell_size + self.grid_dim[0] // 2).long()
        row = ((candidate_pose[:, 2, 3] - mc[1]) / self.cell_size + self.grid_dim[1] // 2).long()
        return candidate_pose[eroded[row, col].bool()]

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
        for name in ("update_occ_map", "build_connected_freespace", "build_frontiers", "_fbe_point", "generate_candidate",
                     "generate_candidate_object", "generate_candidate_in_freespace", "sample_random_candidate", "_ring_candidates",
                     "_next_seed", "_eroded_free", "filter_candidates_in_freespace", "cells_of", "_occ_cfg", "_occ_workspace", "_stream"):
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
        self.K_object = kw.get("sample_view_num_object", ex.get("sample_view_num_object", self.K))
        self.radius_object = kw.get("sample_range_object", ex.get("sample_range_object", self.radius))
        self.min_range_object = kw.get("min_range_object", ex.get("min_range_object", self.min_range))
        self.frontier_select_method = kw.get("frontier_select_method", ex.get("frontier_select_method", "combined"))
        self.pcd_far_distance = kw.get("pcd_far_distance", pol.get("pcd_far_distance", 10.0))
        self.eval_dir = eval_dir
        self.cam_pos = None
        self.frame_idx = 0

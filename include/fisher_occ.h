/*
 * fisher_occ.h -- C ABI of the planner-side kernels in libfisher_rast.so (MI355X / gfx950): the occupancy-map update
 * and the frontier extraction that sit either side of view scoring in a planning round.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   fr_occ_update        <- AstarPlanner.update_occ_map           planning/astar.py:202-301
 *                           (11 depth samples per pixel, torch.unique count binning into a 3 x H x W grid,
 *                            one cv2.line per occupied cell -- a Python loop on the host -- and the normalised add)
 *   fr_occ_freespace     <- AstarPlanner.build_connected_freespace planning/astar.py:401-447
 *                           (arg-max label, Gaussian blocking with count > 25, 3x3 opening, largest 8-connected component)
 *   fr_occ_frontiers     <- AstarPlanner.build_frontiers           planning/astar.py:540-683
 *                           (dilate - free AND unknown, dilate, components, min area 10, largest / combined / closest)
 *   fr_occ_erode         <- cv2.erode(free_space, np.ones((k, k)))  planning/astar.py:805, 1388
 *   fr_occ_cells_of      <- datasets/util/map_utils.py:106-125 discretize_coords
 *   fr_occ_ring_candidates <- AstarPlanner.generate_candidate / generate_candidate_object planning/astar.py:1383-1430, 1432-1469
 *                           fused with the free-space filter of the candidate loop planning/astar.py:1387-1401
 *   fr_occ_free_candidates <- AstarPlanner.sample_random_candidate   planning/astar.py:782-837
 *
 * Conventions are fisher_rast.h's: device pointers unless noted, explicit stream, no allocation, no device
 * synchronisation, 0 or an FR_E* code, message through fr_last_error().
 * occ_map is the planner's float32 [3][grid_h][grid_w] tensor (0 unknown, 1 occupied, 2 free), updated in place.
 */
#ifndef FISHER_OCC_H_INCLUDED
#define FISHER_OCC_H_INCLUDED

#include "fisher_rast.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fr_occ_cfg {
	int32_t grid_w, grid_h;            /* grid_dim[0], grid_dim[1]                      (astar.py:74, 86) */
	float cell_size;                   /* metres per cell                                                  */
	float center_x, center_z;          /* map_center                                     (astar.py:80, 90) */
	float height_lower, height_upper;  /* floor / ceiling filter on world y              (astar.py:265, 284) */
	float far_distance;                /* pcd_far_distance                               (astar.py:247)    */
} fr_occ_cfg;

/* frontier selection (AstarPlanner.frontier_select_method) */
enum { FR_OCC_LARGEST = 0, FR_OCC_COMBINED = 1, FR_OCC_CLOSEST = 2 };

/* bytes of scratch for any of the calls below on this grid */
size_t fr_occ_workspace_bytes(const fr_occ_cfg* cfg);

/* update_occ_map.  depth: [H][W] float32 (device).  intr = {fx, fy, cx, cy} and c2w (row-major 4x4) are HOST arrays;
 * sample_fracs (HOST, n_samples <= 32): the fractions of the depth sampled along each ray, the last one being the depth
 * point itself (the reference: linspace(1e-3, 0.95, 11) with the last set to 1).  cam_col / cam_row: the camera's cell
 * (astar.py:211-213, computed by the caller exactly as the reference does on the host). */
int fr_occ_update(const fr_occ_cfg* cfg, const float* depth, int32_t W, int32_t H, int32_t downsample,
                  const float intr[4], const float c2w[16], const float* sample_fracs, int32_t n_samples,
                  int32_t cam_col, int32_t cam_row, float* occ_map,
                  void* workspace, size_t workspace_bytes, fr_stream_t stream);

/* build_connected_freespace.  points: [n_points][3] world-frame Gaussian means or null.  free_space: uint8 [grid_h][grid_w] out. */
int fr_occ_freespace(const fr_occ_cfg* cfg, const float* occ_map, const float* points, int32_t n_points,
                     uint8_t* free_space, void* workspace, size_t workspace_bytes, fr_stream_t stream);

/* build_frontiers after build_connected_freespace.  Outputs (device):
 *   frontier   uint8 [grid_h][grid_w]   boundary AND unknown, before the dilation             (self.frontier)
 *   target     uint8 [grid_h][grid_w]   the selected component                                 (self.target_frontier)
 *   cells      int32 [max_cells][2]     its (col, row) cells in raster order                   (np.where order)
 *   counts     int32 [4]                {frontier cells before dilation, components over min_area, cells of target, root cell of target or -1}
 * cam_row / cam_col: self.cam_pos. */
int fr_occ_frontiers(const fr_occ_cfg* cfg, const float* occ_map, const uint8_t* free_space,
                     int32_t cam_row, int32_t cam_col, int32_t method, int32_t min_area,
                     uint8_t* frontier, uint8_t* target, int32_t* cells, int32_t max_cells, int32_t* counts,
                     void* workspace, size_t workspace_bytes, fr_stream_t stream);

/* cv2.erode(src, ones(k, k)) with cv2's defaults: anchor (k/2, k/2), cells outside the map do not constrain. */
int fr_occ_erode(const fr_occ_cfg* cfg, const uint8_t* src, uint8_t* dst, int32_t ksize, fr_stream_t stream);

/* discretize_coords for n (x, z) pairs taken from xyz[n][3] (columns 0 and 2): cells[n][2] = (col, row), int32 */
int fr_occ_cells_of(const fr_occ_cfg* cfg, const float* xyz, int32_t n, int32_t* cells, fr_stream_t stream);

/* Candidate poses on a ring around centres drawn (with replacement) from `centers` [n_centers][2] = (x, z):
 * position = centre + r (sin t, 0, cos t) at height cam_height, t = 2 pi u0, r = min_range + u1 (radius - min_range),
 * centre index = floor(u2 n_centers); orientation = yaw (t + pi) with columns 0 and 1 negated (astar.py:1406-1423).
 * The uniforms come from a counter-based generator keyed by (seed, k): see occ_uniform in fisher_occ.hip, restated in
 * oracle/occupancy_frontier.py.  c2w: [K][16] row-major, out.  keep: [K] uint8 out or null -- 1 where the pose's cell lies in
 * `eroded_free` (uint8 [grid_h][grid_w], or null = keep all); as in the reference the filter only applies when more
 * than `min_free` cells of eroded_free are set (astar.py:1389: 40). */
int fr_occ_ring_candidates(const fr_occ_cfg* cfg, const float* centers, int32_t n_centers, int32_t K,
                           float min_range, float radius, float cam_height, uint32_t seed,
                           const uint8_t* eroded_free, int32_t min_free, float* c2w, uint8_t* keep, fr_stream_t stream);

/* Uniformly placed poses in the (already eroded) free space: the free cells in raster order, a quarter as many draws with
 * replacement, position at the cell centre and height agent_y, uniform yaw with columns 1 and 2 negated (astar.py:805-835).
 * c2w: [max_out][16] out; counts: device int32[2] = {free cells, poses written}. */
int fr_occ_free_candidates(const fr_occ_cfg* cfg, const uint8_t* eroded_free, float agent_y, uint32_t seed,
                           float* c2w, int32_t max_out, int32_t* counts, void* workspace, size_t workspace_bytes,
                           fr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

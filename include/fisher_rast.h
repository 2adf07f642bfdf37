/*
 * fisher_rast.h -- C ABI of libfisher_rast.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of davidea97/Fisher-Nerf-customized: the differentiable
 * 3D-Gaussian-splat rasteriser with the FisherRF `grad_power` backward, the per-candidate-view
 * Fisher-information scorer built on it, and simple-knn's distCUDA2.
 *
 * Every entry point
 *   - takes plain device pointers and sizes (no torch types), plus an explicit HIP stream;
 *   - never allocates device memory, never synchronises the device, never throws.  The forward / scorer pipelines
 *     fork onto two internal side streams (sort tier, scorer records) and join back onto the caller's stream; those
 *     two streams and four timing-disabled events are created once per host thread and device, by fr_init() or
 *     lazily by the first call that needs them (fr_profile_enable(1) additionally records two events per launch);
 *   - returns 0 on success or an FR_E* code; fr_last_error() gives the message (thread-local).
 *
 * Reference interfaces replaced (paths relative to the reference tree,
 * RAST = thirdparty/diff-gaussian-rasterization-modified):
 *   fr_mark_visible      <- markVisible                    RAST/rasterize_points.cu:198-217,
 *                                                          RAST/cuda_rasterizer/rasterizer_impl.cu:141-153
 *   fr_forward           <- RasterizeGaussiansCUDA         RAST/rasterize_points.cu:35-115
 *                           -> Rasterizer::forward         RAST/cuda_rasterizer/rasterizer_impl.cu:198-339
 *   fr_backward          <- RasterizeGaussiansBackwardCUDA RAST/rasterize_points.cu:117-196
 *                           -> Rasterizer::backward        RAST/cuda_rasterizer/rasterizer_impl.cu:343-434
 *   fr_forward_features, fr_backward_pair <- the second Renderer call of get_loss, models/SLAM/gaussian.py:203-211
 *   fr_fisher_views      <- the Python loop GaussianSLAM.pose_eval / compute_H_train / compute_Hessian
 *                           models/SLAM/gaussian.py:1338-1375, 1503-1570 and
 *                           models/SLAM/gaussian_object.py:1541-1551, 1591-1617, 1940-2045
 *                           (V x [forward + backward(power=2) + cat + sum]) as one batched call
 *   fr_densify_stats / fr_densify_masks / fr_prune_mask <- the statistics of get_loss / densify / prune_gaussians
 *                           models/SLAM/gaussian.py:289-291, models/SLAM/utils/slam_external.py:196-200, 345-465
 *   fr_knn_dist2         <- simple_knn._C.distCUDA2 (thirdparty/simple-knn, un-vendored submodule)
 *
 * The pybind module `_C` of the reference (RAST/ext.cpp:14-18) is re-created in Python on top of
 * this ABI by fisher-nerf-customized_amd/diff_gaussian_rasterization/_C.py; see INTEGRATION.md.
 */
#ifndef FISHER_RAST_H_INCLUDED
#define FISHER_RAST_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FR_VERSION 100

enum {
	FR_OK = 0,
	FR_EINVAL = 1,      /* bad argument (null pointer, negative size, unsupported combination) */
	FR_ELAUNCH = 2,     /* HIP runtime error at launch; message has hipGetErrorString */
	FR_ENOSPACE = 3     /* workspace too small (host-visible sizes only) */
};

/* device-side status word written by the kernels (int32[4] = {num_rendered_total, overflow, max_tile_count, reserved}) */
#define FR_STATUS_WORDS 4

typedef void* fr_stream_t; /* hipStream_t */

/* Camera / raster settings: GaussianRasterizationSettings, RAST/diff_gaussian_rasterization/__init__.py:140-151.
 * bg / viewmatrix / projmatrix / campos stay DEVICE pointers, as in the reference (they are CUDA tensors there);
 * viewmatrix and projmatrix are the 16 floats of the transposed tensors, i.e. column-major matrices. */
typedef struct fr_raster_cfg {
	int32_t P;
	int32_t image_height;
	int32_t image_width;
	float tanfovx;
	float tanfovy;
	float scale_modifier;
	int32_t sh_degree;   /* D */
	int32_t sh_coeffs;   /* M = shs.size(1), 0 when colours are precomputed */
	int32_t prefiltered; /* auxiliary.h:156-160: with it set, a point culled by the near plane is an error -- the reference traps the
	                        device; fr_forward raises status[3] instead and the Python layer throws the reference's message */
	const float* bg;
	const float* viewmatrix;
	const float* projmatrix;
	const float* campos;
} fr_raster_cfg;

/* Gaussian inputs; a null pointer stands for the reference's empty tensor. */
typedef struct fr_gaussians {
	const float* means3D;        /* [P,3] */
	const float* colors_precomp; /* [P,3] or null */
	const float* shs;            /* [P,M,3] or null */
	const float* opacities;      /* [P] (or [P,1]) */
	const float* scales;         /* [P,3] or null */
	const float* rotations;      /* [P,4] or null */
	const float* cov3D_precomp;  /* [P,6] or null */
} fr_gaussians;

int fr_version(void);
const char* fr_last_error(void);
/* Hash of the kernel / header sources this library was compiled from (stamped by __graft_entry__.build();
 * "unstamped" for a hand build).  The Python loader refuses a library whose id does not match the sources beside it. */
const char* fr_build_id(void);
/* Optional: creates the calling thread's side streams and events for the current device now instead of on first use. */
int fr_init(void);

/* ---- single-view rasteriser (the reference's _C.rasterize_gaussians / _backward / mark_visible) ---- */

/* Bytes of the three opaque work buffers (geomBuffer, binningBuffer, imgBuffer of the reference).
 * out[0] geometry (P), out[1] binning (max_rendered tile instances), out[2] image (W,H). */
int fr_workspace_bytes(int32_t P, int32_t W, int32_t H, int64_t max_rendered, size_t out[3]);

/* Byte offsets of the named sections inside the three buffers (for tests / debuggers).
 * geom:    [0] splat f32[P,8] = {mean2D.x, mean2D.y, conic.x, conic.y, conic.z, opacity, depth, pad} (valid where radii > 0),
 *          [1] cov3D f32[P,6], [2] rgb f32[P,3], [3] clamped u8[P,3]
 * image:   [4] tile_count u32[T], [5] tile_offset u32[T], [6] tile_fill u32[T], [7] final_T f32[HW],
 *          [8] n_contrib u32[HW], [9] status i32[4]
 * binning: [10] keys u64[R]  (sorted per tile: (depth_bits << 32) | gaussian_index) */
int fr_workspace_layout(int32_t P, int32_t W, int32_t H, int64_t max_rendered, size_t offsets[11]);

int fr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                    uint8_t* present, fr_stream_t stream);

/* Forward.  binning_capacity = number of tile instances binning_ws can hold.  status (device int32[4]) receives
 * {num_rendered, overflow, longest tile list, prefiltered violated}; when num_rendered > binning_capacity nothing is rendered, overflow = 1 and the caller
 * re-runs with a larger buffer (the reference instead synchronises on num_rendered before sizing the buffer,
 * rasterizer_impl.cu:282-286).  out_color [3,H,W], out_depth [1,H,W], radii [P]. */
int fr_forward(const fr_raster_cfg* cfg, const fr_gaussians* g,
               void* geom_ws, void* binning_ws, int64_t binning_capacity, void* image_ws,
               float* out_color, float* out_depth, int32_t* radii, int32_t* status, fr_stream_t stream);

/* Backward.  All nine gradient buffers are overwritten (zero-filled first, rasterize_points.cu:151-159).
 * dL_dmeans2D [P,3], dL_dcolors [P,3], dL_dopacity [P,1], dL_dmeans3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3]
 * (may be null when M == 0), dL_dscales [P,3], dL_drotations [P,4], dL_dconic [P,2,2].
 * power is the reference's `backward_power` (1 = gradients, 2 = squared per-pixel gradients). */
int fr_backward(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                const void* geom_ws, const void* binning_ws, const void* image_ws,
                const float* dL_dout_color, int32_t power,
                float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D,
                float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dconic,
                fr_stream_t stream);

/* fr_backward with a scratch buffer the power-2 backward may use: fr_backward_scratch_bytes(P, W, H, power, num_rendered) bytes,
 * num_rendered = what fr_forward reported for this image (status[0]); 0 = no use for one (other powers; more than 16384 tiles or 4 GiB).
 * (Power 1, the training backward, takes the same scratch and the same route for its nine per-candidate sums.)
 * fr_backward's power-2 pass is one workgroup per tile -- one 256 x 256 view is 256 workgroups on 256 CUs, the longest tile list sets
 * the time of the literal `backward_power=2` loop (models/SLAM/gaussian.py:1548-1549) -- and its per-candidate LDS accumulators
 * serialise under splats that cover a strip; given the scratch, the backward is cut into chunks of at most 64
 * candidates of one 16 x 4 pixel strip that run as independent pieces of work (a cheap pass leaves each chunk's effect on a pixel's
 * back-to-front state, a prefix pass the state in front of every chunk).  The same pairs contribute; a pixel's state at the start of a
 * chunk differs from the single pass's by rounding only (a few 1e-7 relative).  scratch == NULL or too small: fr_backward's single pass.
 * num_rendered below what the forward reported (a scratch laid out for too few tile instances): the kernels notice and leave every
 * gradient zero. */
size_t fr_backward_scratch_bytes(int32_t P, int32_t W, int32_t H, int32_t power, int64_t num_rendered);
int fr_backward_ws(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                   const void* geom_ws, const void* binning_ws, const void* image_ws,
                   const float* dL_dout_color, int32_t power,
                   float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D,
                   float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dconic,
                   int64_t num_rendered, void* scratch, size_t scratch_bytes, fr_stream_t stream);

/* ---- second feature image on the same geometry (training step) ------------------------
 * The reference's get_loss renders twice with identical means / scales / rotations / opacities / camera and different
 * colours: RGB, then (depth, 1, depth^2) for depth + silhouette (models/SLAM/gaussian.py:199-211,
 * slam_helpers.py:268-279) -- two full rasteriser forwards and two full backwards.
 * fr_forward_features composites another [P,3] feature array over what fr_forward just projected, binned and sorted
 * (same cfg and workspaces; no preprocess, no sort).  fr_backward_pair is the backward of both images in one call
 * (grad_power 1): one tile pass per image, the per-Gaussian Jacobian chain once.  dL_dmeans2D receives the colour image's
 * screen-space gradient only -- the statistic the densifier reads (gaussian.py:207) -- dL_dmeans2D_features the other
 * image's; every other output is the sum over both images, except dL_dcolors / dL_dfeatures. */
/* fr_forward with a second feature array composited in the same pass: out_features [3,H,W] */
int fr_forward_pair(const fr_raster_cfg* cfg, const fr_gaussians* g, const float* features,
                    void* geom_ws, void* binning_ws, int64_t binning_capacity, void* image_ws,
                    float* out_color, float* out_features, float* out_depth, int32_t* radii, int32_t* status,
                    fr_stream_t stream);
int fr_forward_features(const fr_raster_cfg* cfg, const float* features,
                        const void* geom_ws, const void* binning_ws, void* image_ws,
                        float* out_features, fr_stream_t stream);
int fr_backward_pair(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                     const void* geom_ws, const void* binning_ws, const void* image_ws,
                     const float* dL_dout_color, const float* features, const float* dL_dout_features,
                     float* dL_dmeans2D, float* dL_dmeans2D_features, float* dL_dcolors, float* dL_dfeatures,
                     float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales,
                     float* dL_drotations, float* dL_dconic, fr_stream_t stream);

/* fr_backward_pair with the scratch buffer of the chunked form (fr_backward_ws, above; six colour channels in the per-chunk state):
 * fr_backward_pair_scratch_bytes(P, W, H, num_rendered) bytes, num_rendered = what the forward of this image reported. */
size_t fr_backward_pair_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t num_rendered);
int fr_backward_pair_ws(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                        const void* geom_ws, const void* binning_ws, const void* image_ws,
                        const float* dL_dout_color, const float* features, const float* dL_dout_features,
                        float* dL_dmeans2D, float* dL_dmeans2D_features, float* dL_dcolors, float* dL_dfeatures,
                        float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales,
                        float* dL_drotations, float* dL_dconic,
                        int64_t num_rendered, void* scratch, size_t scratch_bytes, fr_stream_t stream);

/* ---- fused multi-view Fisher scorer -------------------------------------------------------------- */

typedef struct fr_fisher_cfg {
	int32_t n_views;
	int32_t columns;            /* 4 = [mean_cam xyz | opacity] (gaussian.py:1555-1556);
	                               11 = + [scale xyz | rot rxyz] (gaussian_object.py:2022-2027) */
	float dL_dpix;              /* constant upstream gradient of every pixel/channel (reference: 1e-3) */
	const float* w2c;           /* device [n_views,16], row-major 4x4 world->camera (rel_w2c) */
	const float* H_inv;         /* device [P,columns] weights, or null */
	int64_t H_inv_view_stride;  /* elements between consecutive views' H_inv blocks (0 = one block shared) */
	float* out_scores;          /* device [n_views]: sum(cur_H * H_inv) per view, or null */
	float* out_H;               /* device [.., P, columns], ACCUMULATED into (caller zero-fills), or null */
	int64_t out_H_view_stride;  /* elements between views' blocks in out_H (0 = all views sum into one block) */
	int32_t* out_vis_count;     /* device [n_views]: #Gaussians with radius > 0, or null */
	int32_t* out_num_rendered;  /* device [n_views]: tile instances per view, or null */
	const float* dL_dpix_image; /* device [n_views,3,H,W] or null: per view an upstream-gradient image instead of the constant
	                               dL_dpix -- the `im.backward(gradient=z)` probes of estimate_diag_JtJ_simple /
	                               estimate_block_JtJ (gaussian_object.py:2088-2098, 2158-2170).  out_H mode only. */
	int64_t dL_image_view_stride; /* elements between views' images (0 = one image shared by all views) */
	int32_t tile_capacity;      /* 0: the keys of all (view, tile) lists are packed into max_rendered slots (count, scan, scatter).
	                               > 0: every (view, tile) owns a fixed segment of tile_capacity keys, which the projection kernel
	                               fills itself -- no scan dependency and no scatter kernel; needs n_views * tiles * tile_capacity
	                               <= max_rendered (and < 2^32).  A tile with more instances than that: overflow, status[3] = 1. */
	int32_t poses_are_c2w;      /* 1: `w2c` holds camera-to-world poses; the library inverts them (one small kernel) */
	int32_t reuse_static;       /* 1: THIS workspace still holds the per-Gaussian static records the call would build -- the previous
	                               fr_fisher_views call on it (ordered before this one on the stream) had the same Gaussians, the same
	                               shared H_inv rows (or none / per-view ones), the same columns, order, n_views and max_rendered -- so
	                               the packing kernel is skipped.  A planner scores hundreds of pose batches against one map and one
	                               H_inv (tester_gaussians_navigation.py:1684-1705); the caller vouches for the sameness. */
	const uint32_t* order;      /* device [P] or null: a permutation of 0..P-1 -- the order in which the Gaussians are laid out and
	                               processed inside the call (fr_spatial_order: Morton order of the means).  Purely a layout
	                               hint: every input and output keeps the caller's indexing (H_inv rows, out_H rows), the contributor
	                               sets and their depth order are unchanged.  With a spatially coherent order a projection workgroup's
	                               256 Gaussians are neighbours in space: whole groups fall outside a view and are skipped by one bounding
	                               test, a workgroup's keys land in a handful of tiles, a tile's records sit side by side in memory.
	                               (Splats of EQUAL depth in one tile are ordered by their place in `order`; fr_spatial_order is stable
	                               -- equal means keep the caller's order -- so duplicated Gaussians composite as in the reference.)
	                               Ignored by the fall-back kernels (H_inv and out_H in one launch, images beyond 4096 tiles). */
} fr_fisher_cfg;

/* Morton (Z-curve, 10 bits per axis over the bounding box of the means) order of P points: order_out[k] = index of the k-th point
 * along the curve; points with equal codes keep their index order (a stable sort).  The reference has no counterpart: it processes
 * the Gaussians in the order of the parameter tensors (models/SLAM/gaussian.py:1529-1543).  Once per map, not per call. */
size_t fr_spatial_order_workspace_bytes(int32_t P);
int fr_spatial_order(int32_t P, const float* means3D, uint32_t* order_out, void* workspace, size_t workspace_bytes, fr_stream_t stream);

size_t fr_fisher_workspace_bytes(int32_t P, int32_t W, int32_t H, int32_t n_views, int64_t max_rendered, int32_t columns);
/* Byte offsets of the named sections inside the scorer's workspace (for tests / debuggers):
 * [0] tile_count u32[V,T], [1] tile_offset u32[V,T], [2] keys u64[R] (sorted per (view, tile) from tile_offset on: (depth_bits << 32) |
 *     record slot with packed lists, (depth_bits << 32) | slot << 4 | strips of the tile reached with fixed segments -- tile_capacity),
 * [3] dense {recA, recB} records [V,P] x 32 B -- only the single-view front end (images beyond 4096 tiles) fills them; the default
 *     path leaves this section unused -- [4] the scorer's records: COMPACT [V][PV] records, PV = projection workgroups x their
 *     Gaussians (>= P), one per visible (view, Gaussian) at slot = workgroup * its Gaussians + rank among the workgroup's visible
 *     splats of the view: 80 B (score form with fixed key segments: {x, y, k3, log2 o} {-cx/2, -cy, -cz/2, r+g+b} + 12 polynomial
 *     coefficients), 96 B (score form with packed lists, A-form of the 4-column out_H kernel), 112 / 208 B (general out_H form, 4 / 11
 *     columns); dense [V,P] x 64 B with the single-view front end,
 * [5] tile_scores f32[V,T], [6] status i32[4], [7] visible-list lengths u32[V, blocks] */
int fr_fisher_workspace_layout(int32_t P, int32_t W, int32_t H, int32_t n_views, int64_t max_rendered, int32_t columns,
                               size_t offsets[8]);

/* Scores n_views candidate poses in one batched launch sequence.  g->means3D are WORLD positions; each view's
 * camera-frame means are computed in-kernel from cfg_f->w2c and then rendered through cfg->viewmatrix/projmatrix
 * exactly as the reference does (gaussian.py:1523-1548; its camera has viewmatrix = I).
 * status (device int32[4]) receives {total tile instances, overflow, largest tile list, tile_capacity exceeded}; on overflow
 * (total > max_rendered, or a tile list longer than cfg_f->tile_capacity) no score is written and nothing is accumulated. */
int fr_fisher_views(const fr_raster_cfg* cfg, const fr_gaussians* g, const fr_fisher_cfg* cfg_f,
                    void* workspace, size_t workspace_bytes, int64_t max_rendered,
                    int32_t* status, fr_stream_t stream);

/* ---- densification / pruning statistics of the training step ---------------------------
 * One pass over the Gaussians each, in place of the torch-op chains of models/SLAM/gaussian.py:289-291 and
 * models/SLAM/utils/slam_external.py:196-200, 345-465.  All arrays are device float32 [P] unless noted. */

/* After a render (+ backward): seen[i] = radii[i] > 0 (uint8, may be null); where seen: max_2D_radius = max(radius, max_2D_radius)
 * (may be null) and -- when grad_means2D ([P,3], the colour render's means2D.grad) is given -- means2D_gradient_accum += |grad.xy|,
 * denom += 1 (accumulate_mean2d_gradient). */
int fr_densify_stats(int32_t P, const int32_t* radii, const float* grad_means2D, float* max_2D_radius,
                     float* means2D_gradient_accum, float* denom, uint8_t* seen, fr_stream_t stream);

/* densify(): grads = accum / denom with NaN -> 0; to_clone = grads >= grad_thresh AND max_k exp(log_scales) <= clone_max_scale;
 * to_split = max_k exp(log_scales) > split_min_scale (the reference applies no gradient test to the split).
 * log_scales: [P, scale_cols], scale_cols 1 (isotropic) or 3.  Masks: uint8 [P]. */
int fr_densify_masks(int32_t P, const float* means2D_gradient_accum, const float* denom, const float* log_scales,
                     int32_t scale_cols, float grad_thresh, float clone_max_scale, float split_min_scale,
                     uint8_t* to_clone, uint8_t* to_split, fr_stream_t stream);

/* prune_gaussians() / the removal pass of densify(): to_remove = sigmoid(logit_opacities) < opacity_thresh
 * OR (big_thresh >= 0 AND max_k exp(log_scales) > big_thresh).  uint8 [P]. */
int fr_prune_mask(int32_t P, const float* logit_opacities, const float* log_scales, int32_t scale_cols,
                  float opacity_thresh, float big_thresh, uint8_t* to_remove, fr_stream_t stream);

/* ---- simple-knn ---------------------------------------------------------------------------------- */

size_t fr_knn_workspace_bytes(int32_t P);
/* out[i] = mean of the squared distances from points[i] to its 3 nearest other points. */
int fr_knn_dist2(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes,
                 fr_stream_t stream);

/* ---- measurement hooks (not part of the reference surface) ------------------------------------------ */

/* When enabled, every fr_fisher_views call records a pair of HIP events around its dominant kernel
 * (k_fisher_tile_v4 / _v3 in the score-only mode, k_fisher_tile_v3h / _v3g / _v2 otherwise) on the stream the kernel is launched on.  Enabling or disabling clears the record. */
int fr_profile_enable(int on);
/* Waits for the recorded events and writes up to max_n per-launch durations in milliseconds; returns the count
 * (or -1 on a HIP error).  This is the only entry point that blocks. */
int fr_profile_fetch(float* ms, int max_n);

#ifdef __cplusplus
}
#endif
#endif /* FISHER_RAST_H_INCLUDED */

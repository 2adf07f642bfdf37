// fisher_rast.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) and the C ABI of
// libfisher_rast.so.  See include/fisher_rast.h for the boundary and DESIGN.md for the layout.
//
// Pipeline (single view: V = 1; Fisher scorer: V candidate views in one launch):
//   k_cov3d              once per call      scale/quaternion -> cov3D[P][6]            (forward.cu:118-152)
//   k_pack_static        once per call      records modes: one 64 / 128-byte record per Gaussian (mean, cov3D, rgb, ..., H_inv), {mean, trace},
//                                           bounds of every 256 Gaussians; skipped under fr_fisher_cfg.reuse_static
//   k_preprocess         (P/(256 G), V)     project, cull, conic, radius, tile rect, per-tile COUNT through an LDS histogram
//   k_preprocess_views_c (P/(256 G), V/VC)  the scorer's front end for many views of one camera: near-plane + early frustum test, compaction,
//                                           dense projection, compact visible lists, the scorer's per-(view, Gaussian) records written once,
//                                           and (fixed key segments) the keys themselves   (forward.cu:155-256)
//   k_tile_lists / k_scan_tiles             segment offsets (fixed / exclusive scan), device-side num_rendered / overflow flag (replaces
//                                           cub::InclusiveSum + the blocking cudaMemcpy of rasterizer_impl.cu:277-282), list of long tiles
//   k_scatter_keys/_vis  (P/(256 G), V)     packed lists: emit (depth_bits<<32 | gaussian) into the tile's segment
//                                           (duplicateWithKeys, rasterizer_impl.cu:70-111)
//   k_sort_tiles, k_sort_part, k_sort_mid/_big   per-tile bitonic network on the 64-bit keys, in registers (DPP / permlane exchanges
//                                           between lanes, LDS only where waves' runs join); long lists of fixed segments are first
//                                           partitioned at sampled pivots into wave-sized parts: replaces the global
//                                           cub::DeviceRadixSort (rasterizer_impl.cu:304-309).  Keys are unique, so the
//                                           result equals the reference's stable (tile, depth) order with ties by index.
//   k_render_forward_walk<3|6>  (T, V)      alpha compositing, median depth, wave-private strips, per-lane walk (forward.cu:261-393)
//   k_backward_lin_walk / k_backward_lin_tile<pair> + k_backward_finish   grad_power 1 (training): per-splat sums of the screen-space
//                                           gradients, Jacobian chain once per Gaussian    (backward.cu:850-1140, 276-583)
//   k_backward_sq_rows + k_backward_sq_walk grad_power 2 through the rasteriser API
//   k_backward_tile      (T, 1)             generic fused backward, any grad_power, SH colours
//   k_fisher_tile_v4     T*V                the scorer: sum(cur_H * H_inv) per view (gaussian.py:1548-1556, 1367) in ONE front-to-back
//                                           pass over the records (rolling halves), no gradient tensor materialised
//                                           (k_fisher_tile_v3: the chunk-synchronous form, for packed lists / dense records)
//   k_fisher_tile_v3h / _v3g  T*V           cur_H itself (4 / 11 columns, gradient images): two front-to-back passes over the records;
//                                           few views: <.., 1> pass 1 per tile + work list, <.., 2> pass 2 per list segment
//   k_fisher_tile_v2<4|11>  T*V             transmittance pass + backward(power=2) fused, wave-private: H_inv AND out_H in one launch,
//                                           11-column out_H beyond 4096 tiles
//   k_fisher_tile        T*V                first-generation scan kernel: fallback for tiles beyond the LDS index of the above
//   k_knn_*, fr_spatial_order               simple-knn distCUDA2; the Z-curve order of the Gaussians
//
// Wave64 mapping of a 16x16 tile: 256 threads = 4 waves, wave w owns the 16x4 pixel strip of rows 4w..4w+3, so a
// small splat is seen by 1-2 waves and the others skip it with one ballot.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fr_math.h"
#include "../../include/fisher_rast.h"
#include "fr_internal.h"

#define FR_THREADS 256
#define FR_G_MAX 32                  // upper bound of Gaussians per thread in the per-Gaussian kernels (FrParams::G)
#define FR_MAX_LDS_TILES 4096        // tile histogram kept in LDS up to 1024x1024 images
#define FR_PART_MIN_DEFAULT 2048     // fixed key segments: lists beyond this many keys are partitioned (k_sort_part) instead of sorted whole
#define FR_SORT_SMALL_KEYS 2048      // per-tile segments up to this size: 16 KiB of LDS, many workgroups per CU
#define FR_SORT_MID_KEYS 4096        // listed segments up to this size: 32 KiB of LDS, 256 threads
#define FR_SORT_BIG_KEYS 16384       // up to this size: 128 KiB of LDS, one workgroup per CU; beyond: global memory
#define FR_VC_MAX 16                 // most views one k_preprocess_views workgroup takes (its per-view LDS counters; fr_pick_VC clamps to it)
#define FR_BATCH 256                 // splats staged per round in the forward pass
#define FR_BWD_BATCH 128             // splats staged per round in the backward passes

#include <vector>
#include <utility>
static thread_local char g_err[512] = "";
// The PRODUCT build (__graft_entry__.build()) reads no environment variable and holds only kernels a default call can reach.
// -DFR_AB (tools/build_variant.sh -> tools/_build/, loaded through FISHER_RAST_SO by tools/ab.sh, loopstats.py, fe_ablate.py) is the
// experiment rig: FR_DEBUG_MODE / FR_GV / FR_VC / FR_GROUPS / FR_TILE_PRIO and the kernel generations only those switches select.
#if defined(FR_ABLATE) || defined(FR_LOOPSTATS)
#ifndef FR_AB
#define FR_AB
#endif
#endif
#ifdef FR_AB
#define FR_AB_ONLY(x) x
static int fr_debug_mode();
static int fr_env_int(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#else
#define FR_AB_ONLY(x)
static constexpr int fr_debug_mode() { return 0; }
#endif
// measurement only: HIP events recorded around the dominant kernel on the stream it is launched on
static bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
int fr_fail(int code, const char* msg)
{
	snprintf(g_err, sizeof(g_err), "%s", msg);
	return code;
}
int fr_check_launch(const char* what)
{
	hipError_t e = hipGetLastError();
	if (e != hipSuccess)
	{
		snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
		return FR_ELAUNCH;
	}
	return FR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Everything preprocess derives for one (view, Gaussian): one 32-byte record = one sector per gather.
// `ext` = two fp16 (rounded up): half extents (hx, hy) of the conservative pixel box in which alpha can reach 1/255
// (negative hx: nowhere; +inf: unknown, treat as everywhere).
struct FrSplat { float x, y, cx, cy, cz, o, depth; uint32_t ext; };

// One visible (view, Gaussian) as k_preprocess_views hands it to k_scatter_vis: everything the key scatter needs.
struct FrVisEntry { uint32_t idx, depth_bits, xy0, xy1; };   // tile rect: xy0 = x0 | y0 << 16, xy1 = x1 | y1 << 16

struct FrParams {
	int P, V, W, H;
	uint32_t gx, gy;
	int T;                       // tiles per view
	int G;                       // Gaussians per thread in k_preprocess / k_scatter_keys
	float tanfovx, tanfovy, focal_x, focal_y, mod;
	int D, M;
	const float* bg; const float* view; const float* proj; const float* campos;
	const float* w2c;            // [V][16] row-major, or null
	const float* means3D; const float* colors; const float* shs; const float* opac;
	const float* scales; const float* rots;
	const float* cov3D;          // [P][6] (precomputed input or output of k_cov3d)
	float* cov3D_out;
	int* radii;                  // [V][P]
	FrSplat* splat;              // [V][P] 32-byte records, written only where radii > 0
	float* rgb;                  // [V][P][3]  (SH path only)
	uint8_t* clamped;            // [V][P][3]
	uint32_t* tile_cnt; uint32_t* tile_off; uint32_t* tile_fill; // [V][T]
	uint64_t* keys; long long key_capacity;
	uint32_t* blk_base;          // [V][gridDim.x][T] or null: start of each preprocess workgroup's range inside a tile segment
	uint32_t* big_list;          // [0] = number of tiles with more than FR_SORT_SMALL_KEYS splats, [1] = k_tile_lists' finished-block count, [16..] the tiles (view*T + tile)
	uint32_t* view_work;         // [V] tile instances listed per view, or null  } the balanced deal of the views over the XCDs
	uint32_t* view_perm;         // [V] (round * 8 + XCD) -> view, or null        } (fr_deal_views / fr_tile_of_block)
	int* status;                 // [4]
	int* vis_count;              // [V] or null
	int* num_rendered;           // [V] or null
	// multi-view front end (k_preprocess_views / k_scatter_vis), null on the single-view API
	struct FrVisEntry* vis_list; // [V][blocks][FR_THREADS * G] compacted visible splats of each preprocess workgroup
	uint32_t* vis_n;             // [V][blocks] their number
	int VC;                      // views per preprocess workgroup
	uint32_t tile_cap;           // 0: tile segments packed by the scan; > 0: every (view, tile) owns keys[(v T + t) tile_cap ...), filled by
	                             // the projection kernel itself (k_preprocess_views_c<.., true>): no scan dependency, no scatter kernel
	uint32_t small_max;          // lists of up to this many keys are sorted by k_sort_tiles, longer ones are listed for the other sort kernels
	const uint32_t* order;       // [P] or null (fr_fisher_cfg.order): position -> the caller's Gaussian index; records modes of the multi-view front end only
	int legacy_sort;             // FR_DEBUG_MODE=6: the LDS-resident sort network of round 1 (A/B runs)
	int prefiltered;             // GaussianRasterizationSettings.prefiltered (single-view API): a near-plane-culled point raises status[3]
	int ablate;                  // -DFR_ABLATE builds only (tools/fe_ablate.py): FR_DEBUG_MODE 30..34 drop parts of the direct key scatter
};
#ifdef FR_ABLATE
#define FR_ABL(x) x
#else
#define FR_ABL(x)
#endif

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
	return v;
}

// wave votes on the scalar unit (the __any / __all builtins go through a VGPR round trip)
__device__ __forceinline__ bool fr_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }
__device__ __forceinline__ bool fr_all(bool x) { return __builtin_amdgcn_ballot_w64(!x) == 0ull; }
__device__ __forceinline__ float fr_readlane_f(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ float fr_bperm_f(float x, int l) { return __int_as_float(__builtin_amdgcn_ds_bpermute(l << 2, __float_as_int(x))); }

// Hardware exponential (v_exp_f32 after one multiply, ~1 ulp of 2^x) for the SCORER only: its bar is 1e-4 on the scores, and a
// pair whose alpha sits within an ulp of 1/255 (or a pixel whose T sits within an ulp of 1e-4) carries a weight far below
// that.  The single-view rasteriser keeps fr_expf, whose forward outputs are bit-identical to the oracle's.
__device__ __forceinline__ float fr_exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// conservative lower bound on `power` below which alpha = opacity*exp(power) is certainly < 1/255
__device__ __forceinline__ float fr_power_threshold(float opacity)
{
	return (opacity > 0.f) ? (-__logf(255.0f * opacity) - 0.01f) : INFINITY;
}

// Half extents (hx, hy), as two fp16 rounded up, of the region where `power >= thr` can hold, i.e. where alpha can reach
// 1/255: the ellipse 1/2 d^T Q d <= -thr has half extents sqrt(-2 thr Sigma_xx), sqrt(-2 thr Sigma_yy), Sigma = Q^-1.
// 1 % + 0.01 px of slack dwarfs the rounding of `power`.  hx < 0 encodes "no pixel", +inf "unknown".
__device__ __forceinline__ uint32_t fr_alpha_extent(float cx, float cy, float cz, float opacity)
{
	const float thr = fr_power_threshold(opacity);
	float hx = INFINITY, hy = INFINITY;
	if (!(thr <= 0.f)) { hx = -1.f; hy = -1.f; }        // opacity <= 1/255 (or NaN): never reaches the alpha threshold
	else
	{
		const float det = cx * cz - cy * cy;
		if (det > 0.f && cx > 0.f && cz > 0.f)
		{
			const float tau2 = -2.0f * thr;
			const float ex = sqrtf(tau2 * cz / det) * 1.01f + 0.01f;
			const float ey = sqrtf(tau2 * cx / det) * 1.01f + 0.01f;
			if (ex == ex && ey == ey) { hx = ex; hy = ey; }
		}
	}
	const __half2 h = __halves2half2(__float2half_ru(hx), __float2half_ru(hy));
	return *(const uint32_t*)&h;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FR_THREADS) void k_cov3d(int P, const float* __restrict__ scales, float mod,
                                                      const float* __restrict__ rots, float* __restrict__ cov3D)
{
	int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= P) return;
	fr_f3 s = { scales[3 * i], scales[3 * i + 1], scales[3 * i + 2] };
	fr_f4 q = { rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3] };
	float c[6];
	fr_cov3d(s, mod, q, c);
#pragma unroll
	for (int k = 0; k < 6; k++) cov3D[6 * (size_t)i + k] = c[k];
}

__global__ __launch_bounds__(FR_THREADS) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                             const float* __restrict__ view, uint8_t* __restrict__ present)
{
	int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= P) return;
	float vm[16];
#pragma unroll
	for (int k = 0; k < 16; k++) vm[k] = view[k];
	fr_f3 p = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
	fr_f3 pv = fr_xform4x3(p, vm);
	present[i] = (pv.z <= 0.001f) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FR_THREADS) void k_preprocess(FrParams p)
{
	extern __shared__ uint32_t fr_dyn_lds[];     // T counters when T <= FR_MAX_LDS_TILES (sized at launch)
	uint32_t* hist = fr_dyn_lds;
	const int tid = threadIdx.x;
	const int v = blockIdx.y;
	const bool lds_hist = p.T <= FR_MAX_LDS_TILES;
	if (lds_hist)
	{
		for (int t = tid; t < p.T; t += FR_THREADS) hist[t] = 0;
		__syncthreads();
	}
	float vm[16], pm[16], wm[12];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	const bool has_w2c = p.w2c != nullptr;
	if (has_w2c)
	{
#pragma unroll
		for (int k = 0; k < 12; k++) wm[k] = p.w2c[16 * (size_t)v + k];
	}
	uint32_t* cnt = p.tile_cnt + (size_t)v * p.T;
	const size_t vP = (size_t)v * p.P;
	int nvis = 0;
	for (int g = 0; g < p.G; g++)
	{
		const int i = (blockIdx.x * p.G + g) * FR_THREADS + tid;
		if (i >= p.P) break;
		fr_f3 pw = { p.means3D[3 * (size_t)i], p.means3D[3 * (size_t)i + 1], p.means3D[3 * (size_t)i + 2] };
		fr_f3 po = has_w2c ? fr_world_to_cam(pw, wm) : pw;
		// cheap reject first (identical to the first test of preprocess_one) so culled splats never load cov3D
		fr_f3 p_view = fr_xform4x3(po, vm);
		fr_splat s;
		s.radius = 0;
		// auxiliary.h:156-160: with `prefiltered` the reference prints "Point is filtered although prefiltered is set" and traps the
		// device; here the status word carries the fact to the host, which raises (diff_gaussian_rasterization / fisher_rast.ops)
		if (p.prefiltered && p_view.z <= 0.001f) p.status[3] = 1;
		if (!(p_view.z <= 0.001f))
		{
			float c3[6];
#pragma unroll
			for (int k = 0; k < 6; k++) c3[k] = p.cov3D[6 * (size_t)i + k];
			s = fr_preprocess_one(po, c3, vm, pm, p.W, p.H, p.tanfovx, p.tanfovy, p.focal_x, p.focal_y, p.gx, p.gy);
		}
		p.radii[vP + i] = s.radius;
		if (s.radius > 0)
		{
			nvis++;
			float4* dst = (float4*)(p.splat + vP + i);
			dst[0] = make_float4(s.px, s.py, s.conx, s.cony);
			dst[1] = make_float4(s.conz, p.opac[i], s.depth, __uint_as_float(fr_alpha_extent(s.conx, s.cony, s.conz, p.opac[i])));
			if (p.colors == nullptr)
			{
				fr_f3 cp = { p.campos[0], p.campos[1], p.campos[2] };
				uint8_t cl[3];
				fr_f3 c = fr_sh_to_rgb(p.D, po, cp, p.shs + 3 * (size_t)i * p.M, cl);
				p.rgb[3 * (vP + i)] = c.x; p.rgb[3 * (vP + i) + 1] = c.y; p.rgb[3 * (vP + i) + 2] = c.z;
				p.clamped[3 * (vP + i)] = cl[0]; p.clamped[3 * (vP + i) + 1] = cl[1]; p.clamped[3 * (vP + i) + 2] = cl[2];
			}
			for (uint32_t y = s.rect.y0; y < s.rect.y1; y++)
				for (uint32_t x = s.rect.x0; x < s.rect.x1; x++)
				{
					if (lds_hist) atomicAdd(&hist[y * p.gx + x], 1u);
					else atomicAdd(&cnt[y * p.gx + x], 1u);
				}
		}
	}
	if (lds_hist)
	{
		__syncthreads();
		for (int t = tid; t < p.T; t += FR_THREADS)
		{
			uint32_t c = hist[t];
			if (c)
			{
				// the returned running count is this workgroup's private range inside the tile's segment: k_scatter_keys,
				// launched with the same decomposition, hands out slots inside it without counting again
				const uint32_t before = atomicAdd(&cnt[t], c);
				if (p.blk_base) p.blk_base[((size_t)v * gridDim.x + blockIdx.x) * p.T + t] = before;
			}
		}
	}
	if (p.vis_count)
	{
		int s = nvis;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
		if ((tid & 63) == 0 && s) atomicAdd(&p.vis_count[v], s);
	}
}

// The XCD-aware tile map (fr_tile_of_block) gives every XCD whole views, eight rounds of one view per XCD for 64 views.  Views differ
// several-fold in their tile instances (61k .. 513k on the bench workload), so dealt in index order (view v -> XCD v mod 8) the XCDs'
// shares differ and the walk ends with most XCDs idle.  Here the views are ranked by their listed instances and dealt in a snake
// (round r: ranks 8r .. 8r+7 to XCDs 0..7, the next round 7..0), heaviest round first.  perm[r * 8 + x] = the view XCD x walks in
// round r.  One workgroup; V <= 1024 (beyond that, and for V % 8 != 0, the map stays the plain one).
__device__ __forceinline__ void fr_deal_views(const uint32_t* __restrict__ work, uint32_t* __restrict__ perm, int V, int tid, int nthreads)
{
	__shared__ uint32_t s_work[1024];
	for (int v = tid; v < V; v += nthreads) s_work[v] = __hip_atomic_load(work + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (other workgroups' atomic adds: read at the L2)
	__syncthreads();
	for (int v = tid; v < V; v += nthreads)
	{
		const uint32_t w = s_work[v];
		int rank = 0;
		for (int u = 0; u < V; u++)
		{
			const uint32_t wu = s_work[u];
			rank += (wu > w || (wu == w && u < v)) ? 1 : 0;
		}
		const int r = rank >> 3, j = rank & 7;
		perm[r * 8 + ((r & 1) ? 7 - j : j)] = (uint32_t)v;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Exclusive scan of N = V*T tile counts (one block).  status = {total, overflow, max tile count, 0}.
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off,
                                                     uint32_t* __restrict__ fill, int N, int T, int V,
                                                     long long capacity, int* __restrict__ status,
                                                     int* __restrict__ num_rendered, uint32_t* __restrict__ big_list,
                                                     uint32_t* __restrict__ view_work, uint32_t* __restrict__ view_perm)
{
	__shared__ uint32_t wsum[16];
	__shared__ uint32_t chunk_total;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t carry = 0, maxc = 0;
	// A thread takes 16 consecutive tiles (serial prefix in registers), the block scans the 1024 partial sums once: two
	// barriers per 16384 tiles (a round of 1024 tiles at a time cost 3 barriers and a load latency each: 23 us for 64 views).
	constexpr int PER = 16;
	for (int sbase = 0; sbase < N; sbase += 1024 * PER)
	{
		const int i0 = sbase + tid * PER;
		uint32_t c[PER];
		uint32_t s = 0;
		const bool whole = i0 + PER <= N;                 // (the arrays are 256-byte aligned and i0 is a multiple of 16)
		if (whole)
		{
#pragma unroll
			for (int q = 0; q < PER / 4; q++)
			{
				const uint4 t4 = ((const uint4*)(cnt + i0))[q];
				c[4 * q] = t4.x; c[4 * q + 1] = t4.y; c[4 * q + 2] = t4.z; c[4 * q + 3] = t4.w;
			}
		}
		else
		{
#pragma unroll
			for (int q = 0; q < PER; q++) c[q] = (i0 + q < N) ? cnt[i0 + q] : 0u;
		}
#pragma unroll
		for (int q = 0; q < PER; q++)
		{
			s += c[q];
			maxc = c[q] > maxc ? c[q] : maxc;
			if (c[q] > (uint32_t)FR_SORT_SMALL_KEYS) big_list[16 + atomicAdd(&big_list[0], 1u)] = (uint32_t)(i0 + q);
		}
		uint32_t x = s;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1)
		{
			uint32_t y = __shfl_up(x, d, 64);
			if (lane >= d) x += y;
		}
		if (lane == 63) wsum[wave] = x;
		__syncthreads();
		if (wave == 0)
		{
			uint32_t ws = lane < 16 ? wsum[lane] : 0u;
			uint32_t incl = ws;
#pragma unroll
			for (int d = 1; d < 16; d <<= 1)
			{
				uint32_t y = __shfl_up(incl, d, 64);
				if (lane >= d) incl += y;
			}
			if (lane < 16) wsum[lane] = incl - ws;
			if (lane == 15) chunk_total = incl;
		}
		__syncthreads();
		uint32_t run = carry + wsum[wave] + (x - s);
		if (whole)
		{
#pragma unroll
			for (int q = 0; q < PER / 4; q++)
			{
				uint4 o4;
				o4.x = run; run += c[4 * q]; o4.y = run; run += c[4 * q + 1]; o4.z = run; run += c[4 * q + 2]; o4.w = run; run += c[4 * q + 3];
				((uint4*)(off + i0))[q] = o4;
				((uint4*)(fill + i0))[q] = make_uint4(0u, 0u, 0u, 0u);
			}
		}
		else
		{
#pragma unroll
			for (int q = 0; q < PER; q++)
			{
				if (i0 + q < N) { off[i0 + q] = run; fill[i0 + q] = 0u; }
				run += c[q];
			}
		}
		carry += chunk_total;
		__syncthreads();
	}
	// block max of maxc
	{
		int m = wave_max_i((int)maxc);
		if (lane == 0) wsum[wave] = (uint32_t)m;
		__syncthreads();
		if (tid == 0)
		{
			uint32_t mm = 0;
			for (int w = 0; w < 16; w++) mm = wsum[w] > mm ? wsum[w] : mm;
			status[0] = (int)carry;
			status[1] = ((long long)carry > capacity) ? 1 : 0;
			status[2] = (int)mm;                   // (status[3]: zero-filled before the front end; k_preprocess raises it for `prefiltered`)
		}
	}
	if (num_rendered)
	{
		__syncthreads();
		for (int v = tid; v < V; v += 1024)
		{
			uint32_t a = off[(size_t)v * T];
			uint32_t b = (v + 1 < V) ? off[(size_t)(v + 1) * T] : carry;
			num_rendered[v] = (int)(b - a);
		}
	}
	if (view_perm)
	{
		__syncthreads();
		for (int v = tid; v < V; v += 1024)
		{
			uint32_t a = off[(size_t)v * T];
			uint32_t b = (v + 1 < V) ? off[(size_t)(v + 1) * T] : carry;
			view_work[v] = b - a;
		}
		__syncthreads();
		fr_deal_views((const uint32_t*)view_work, view_perm, V, tid, 1024);
	}
}

// Fixed key segments (FrParams::tile_cap): nothing to scan.  off[i] = i tile_cap, the list of long tiles, and the status word
// {total, a tile over its capacity, longest list, the same flag} by atomics on the zero-filled words; four tiles per thread.
__global__ __launch_bounds__(FR_THREADS) void k_tile_lists(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ off, int N, int T, uint32_t tile_cap,
                                                           int* __restrict__ status, uint32_t* __restrict__ big_list,
                                                           uint32_t* __restrict__ view_work, uint32_t* __restrict__ view_perm, int ablate, uint32_t list_min)
{
	const int i0 = (blockIdx.x * FR_THREADS + threadIdx.x) * 4;
	// per-view sums: in LDS first (a block's 1024 tiles span 1024 / T + 1 views at most), then one global add per view and block
	__shared__ uint32_t s_vw[4 * FR_THREADS + 2];
	const int vfirst = (blockIdx.x * 4 * FR_THREADS) / T;
	const int ilast = min(N - 1, (int)(blockIdx.x * 4 * FR_THREADS) + 4 * FR_THREADS - 1);
	const int nvb = view_work ? ilast / T - vfirst + 1 : 0;
	for (int k = threadIdx.x; k < nvb; k += FR_THREADS) s_vw[k] = 0u;
	__syncthreads();
	uint32_t sum = 0, mx = 0;
#pragma unroll
	for (int q = 0; q < 4; q++)
	{
		const int i = i0 + q;
		if (i >= N) break;
		const uint32_t c = cnt[i];
		off[i] = (uint32_t)i * tile_cap;
		sum += c;
		mx = c > mx ? c : mx;
		if (c > list_min) big_list[16 + atomicAdd(&big_list[0], 1u)] = (uint32_t)i;
		if (view_work && c) atomicAdd(&s_vw[i / T - vfirst], c);
	}
	__syncthreads();
	for (int k = threadIdx.x; k < nvb; k += FR_THREADS) if (s_vw[k]) atomicAdd(&view_work[vfirst + k], s_vw[k]);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
	{
		sum += (uint32_t)__shfl_xor((int)sum, o, 64);
		const uint32_t t = (uint32_t)__shfl_xor((int)mx, o, 64);
		mx = t > mx ? t : mx;
	}
	if ((threadIdx.x & 63) == 0)
	{
		if (sum) atomicAdd(&status[0], (int)sum);
		if (mx) atomicMax(&status[2], (int)mx);
		if (mx > tile_cap) { atomicOr(&status[1], 1); atomicOr(&status[3], 1); }
		FR_ABL(if (ablate >= 30) atomicOr(&status[1], 1);)
	}
	if (view_perm)
	{
		// the block that finishes last deals the views (its reads of view_work follow every other block's adds: fence + counter)
		__shared__ bool s_last;
		__threadfence();
		__syncthreads();
		if (threadIdx.x == 0) s_last = atomicAdd(&big_list[1], 1u) == gridDim.x - 1;
		__syncthreads();
		if (s_last)
		{
			__threadfence();
			fr_deal_views((const uint32_t*)view_work, view_perm, N / T, threadIdx.x, FR_THREADS);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FR_THREADS) void k_scatter_keys(FrParams p)
{
	// Block-aggregated slot claims: count this block's instances per tile in LDS, claim one contiguous range per
	// non-empty tile with a single global atomic, then hand out slots inside the ranges with LDS atomics.  (The order
	// of a tile's segment is irrelevant: k_sort_tiles sorts it and the keys are unique.)
	extern __shared__ uint32_t fr_dyn_lds[];     // 2 T words when T <= FR_MAX_LDS_TILES (sized at launch)
	uint32_t* s_cnt = fr_dyn_lds;
	uint32_t* s_base = fr_dyn_lds + p.T;
	if (p.status[1]) return;
	const int tid = threadIdx.x;
	const int v = blockIdx.y;
	const size_t vP = (size_t)v * p.P;
	const uint32_t* off = p.tile_off + (size_t)v * p.T;
	uint32_t* fill = p.tile_fill + (size_t)v * p.T;
	const bool lds = p.T <= FR_MAX_LDS_TILES;
	if (lds && p.blk_base)
	{
		const uint32_t* bb = p.blk_base + ((size_t)v * gridDim.x + blockIdx.x) * p.T;
		for (int t = tid; t < p.T; t += FR_THREADS) { s_cnt[t] = 0; s_base[t] = off[t] + bb[t]; }   // bb[t] is only read where this workgroup counted > 0
		__syncthreads();
	}
	else if (lds)
	{
		for (int t = tid; t < p.T; t += FR_THREADS) s_cnt[t] = 0;
		__syncthreads();
		for (int g = 0; g < p.G; g++)
		{
			const int i = (blockIdx.x * p.G + g) * FR_THREADS + tid;
			if (i >= p.P) break;
			const int rad = p.radii[vP + i];
			if (rad > 0)
			{
				const float2 xy = *(const float2*)(p.splat + vP + i);
				const fr_rect rc = fr_get_rect(xy.x, xy.y, rad, p.gx, p.gy);
				for (uint32_t y = rc.y0; y < rc.y1; y++)
					for (uint32_t x = rc.x0; x < rc.x1; x++)
						atomicAdd(&s_cnt[y * p.gx + x], 1u);
			}
		}
		__syncthreads();
		for (int t = tid; t < p.T; t += FR_THREADS)
		{
			const uint32_t c = s_cnt[t];
			if (c) s_base[t] = off[t] + atomicAdd(&fill[t], c);
			s_cnt[t] = 0;
		}
		__syncthreads();
	}
	// Batches of 8 Gaussians per thread with every load of a batch in flight before the first use: the kernel is a chain
	// of dependent loads (radii -> splat -> key store) and spends ~90 % of its wave cycles waiting otherwise.
	constexpr int NB = 8;
	for (int g0 = 0; g0 < p.G; g0 += NB)
	{
		int rad[NB];
		float4 q0[NB];
		float dep[NB];
#pragma unroll
		for (int b = 0; b < NB; b++)
		{
			const int i = (blockIdx.x * p.G + g0 + b) * FR_THREADS + tid;
			rad[b] = (g0 + b < p.G && i < p.P) ? p.radii[vP + i] : 0;
		}
#pragma unroll
		for (int b = 0; b < NB; b++)
		{
			const int i = (blockIdx.x * p.G + g0 + b) * FR_THREADS + tid;
			if (rad[b] > 0)
			{
				const float4* sp = (const float4*)(p.splat + vP + i);
				q0[b] = sp[0];
				dep[b] = sp[1].z;
			}
		}
#pragma unroll
		for (int b = 0; b < NB; b++)
		{
			if (rad[b] > 0)
			{
				const int i = (blockIdx.x * p.G + g0 + b) * FR_THREADS + tid;
				const uint64_t hi = ((uint64_t)fr_as_u32(dep[b])) << 32;
				const fr_rect rc = fr_get_rect(q0[b].x, q0[b].y, rad[b], p.gx, p.gy);
				for (uint32_t y = rc.y0; y < rc.y1; y++)
					for (uint32_t x = rc.x0; x < rc.x1; x++)
					{
						const uint32_t t = y * p.gx + x;
						const uint32_t slot = lds ? (s_base[t] + atomicAdd(&s_cnt[t], 1u)) : (off[t] + atomicAdd(&fill[t], 1u));
						p.keys[slot] = hi | (uint32_t)i;
					}
			}
		}
	}
}

// (defined with the scorer kernels below) one scorer record per visible (view, Gaussian)
struct FrRecordArgs {
	const float* H_inv; long long hinv_stride; const float* packed; float4* recq;
	const float4* mt;            // [P] {mean xyz, trace of cov3D} in processing order (k_pack_static): what phase A of the front end reads
	const float4* grp;           // [ceil(P / 256)][2] {lo xyz, largest trace} {hi xyz, 0}: bounds of every 256 consecutive Gaussians, or null
	int early;                   // 1: the early frustum test (and the group test) may be used -- cov3D is the one k_pack_static built
	// compact records (multi-view front end in a records mode), or null: one 96-byte record {recA, recB, recQ[4]} per SLOT,
	// slot = projection workgroup * (256 G) + rank of the Gaussian among the workgroup's visible ones of that view -- monotone in
	// the Gaussian index, so keys that carry the slot sort exactly like keys that carry the index, and a workgroup's records of a
	// view form one dense run instead of being strewn over [P]
	float4* comp;                // [V][PV][stride], PV = workgroups * 256 G
	uint32_t* slot_idx;          // [V][PV]: slot -> Gaussian index (k_fisher_tile_v3h / _v3g flush by index)
	int stride;                  // float4 per compact record: 6 ({recA, recB} + 4), or 7 / 13 in the general out_H form (fr_fisher_record_general)
};
// floats per Gaussian of the packed static record (k_pack_static): {mean 3, cov3D 6, rgb 3, (scale 3, rot 4), H_inv C}
template <int C> struct FrPackSize { static constexpr int value = (C >= 11) ? 32 : 16; };
template <int C, bool REWRITE, bool FORM_A = false, bool FIVE = false>
__device__ __forceinline__ void fr_fisher_record_one(const FrParams& p, const float* __restrict__ H_inv, long long hinv_stride,
                                                     const float* __restrict__ packed, float4* __restrict__ recq, int v, uint32_t id,
                                                     const float* vm, const float* pm, const float* wm, bool has_w2c,
                                                     float4* out6 = nullptr, const float4* ab_src = nullptr);
// float4 per compact record of a front-end mode: AF 0 = score form, 1 = A-form of k_fisher_tile_v3h, 2 = general out_H form
// (score form with fixed key segments: 5 -- the 80 bytes the walk parks, k3 in the place of the footprint extents, which the tile
// kernel then no longer needs: its keys say which strips a splat reaches and the footprint rows come from the conic itself)
template <int C, int AF, bool DK = false> struct FrRecStride { static constexpr int value = AF == 2 ? (C >= 11 ? 13 : 7) : ((AF == 0 && DK) ? 5 : 6); };
template <int C>
__device__ __forceinline__ void fr_fisher_record_general(const FrParams& p, const float* __restrict__ packed, int v, uint32_t id,
                                                         const float* vm, const float* pm, const float* wm, bool has_w2c,
                                                         float4* out, const float4* ab_src);

// ---------------------------------------------------------------------------------------------------------
// Multi-view front end of the Fisher path (same camera, one rigid transform per candidate view).
// k_preprocess is VALU-bound there: only ~1/4 of the (view, Gaussian) pairs are visible, so the ~400-instruction
// projection ran with a quarter of its lanes.  Here a workgroup takes FR_THREADS*G Gaussians x VC views:
//   phase A  every thread holds one Gaussian and runs the near-plane test (the first test of preprocessCUDA,
//            forward.cu:181-199) for the VC views; survivors are compacted into an LDS pair list (wave ballot);
//   phase B  the pair list is processed densely: projection, conic, radius, tile rect (fr_preprocess_one), the
//            32-byte splat record, the per-(view, tile) LDS histogram, and one 16-byte entry per VISIBLE pair in
//            this workgroup's compact list, which is all k_scatter_vis reads (no radii array, no idle lanes).
// The arithmetic per pair is fr_preprocess_one's, so radii / rects / depths are bit-identical to k_preprocess.
//   phase C  (RC != 0: score-only mode) the workgroup walks the compact lists it has just written and turns every visible
//            (view, Gaussian) into the scorer's record (fr_fisher_record_one) while the splat records are still in L2.
template <int RC, bool PHASE_C>
__global__ __launch_bounds__(FR_THREADS) void k_preprocess_views(FrParams p, FrRecordArgs ra)
{
	extern __shared__ uint32_t fr_dyn_lds[];     // hist[VC][T] | pairs[FR_THREADS * VC] | wm[VC][12] | visible bitmap[VC][8 G] | its prefix popcounts[VC][8 G]
	const int VC = p.VC;
	uint32_t* hist = fr_dyn_lds;
	uint32_t* pairs = hist + (size_t)VC * p.T;
	float* s_wm = (float*)(pairs + FR_THREADS * VC);
	const int W32 = 8 * p.G;                     // 32-bit words of a view's visibility bitmap over the workgroup's 256 G Gaussians
	uint32_t* s_bm = (uint32_t*)(s_wm + 12 * VC);
	uint32_t* s_pf = s_bm + VC * W32;
	const bool compact = RC != 0 && ra.comp != nullptr;
	__shared__ uint32_t s_np;
	__shared__ uint32_t s_n[FR_VC_MAX];
	__shared__ uint32_t s_ref[FR_VC_MAX];        // tile instances by the reference's rule (radius rectangle), per view
	const int tid = threadIdx.x, lane = tid & 63;
	const int v0 = blockIdx.y * VC;
	const int nv = min(VC, p.V - v0);
	const uint32_t nblk = gridDim.x;
	const uint32_t cap = (uint32_t)(FR_THREADS * p.G);
	for (int t = tid; t < nv * p.T; t += FR_THREADS) hist[t] = 0;
	const bool has_w2c = p.w2c != nullptr;
	if (has_w2c) for (int t = tid; t < nv * 12; t += FR_THREADS) s_wm[t] = p.w2c[16 * (size_t)(v0 + t / 12) + (t % 12)];
	if (tid < FR_VC_MAX) { s_n[tid] = 0; s_ref[tid] = 0; }
	if (tid == 0) s_np = 0;
	if (compact) for (int t = tid; t < nv * W32; t += FR_THREADS) s_bm[t] = 0u;
	float vm[16], pm[16];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	__syncthreads();
	for (int g = 0; g < p.G; g++)
	{
		const int i0 = (blockIdx.x * p.G + g) * FR_THREADS;
		if (i0 >= p.P) break;
		// ---- phase A
		{
			const int i = i0 + tid;
			const bool live = i < p.P;
			fr_f3 pw = { 0.f, 0.f, 0.f };
			if (live) pw = fr_f3{ p.means3D[3 * (size_t)i], p.means3D[3 * (size_t)i + 1], p.means3D[3 * (size_t)i + 2] };
			// Early frustum test (records modes): about half of the splats in front of the camera end with an empty tile rectangle
			// (forward.cu:233-236) -- after the whole cov2D chain.  radius = ceil(3 sqrt(lambda1)) and
			//   lambda1 <= lambda_max(cov2D) + 0.3 + sqrt(0.1),   lambda_max(cov2D) <= |J|_F^2 lambda_max(cov3D),
			//   |J|_F^2 = (fx^2 (1 + jx^2) + fy^2 (1 + jy^2)) / z^2,  jx = min(|x / z|, 1.3 tan_fovx) (the clamp of forward.cu:80-84),
			// times |W|_2^2 of the view matrix's 3 x 3, and lambda_max(cov3D) <= trace(cov3D) (k_pack_static), so a centre further than
			// that bound (+ 2 % and 2 px for rounding) outside the tile grid cannot touch it.  NaNs keep the pair.
			const bool early = RC != 0 && ra.early != 0 && ra.mt != nullptr;
			float tr = 0.f;
			if (early && live) tr = ra.mt[i].w;
			const float lx = 1.3f * p.tanfovx, ly = 1.3f * p.tanfovy;
			// |W|_2^2 <= |W|_1 |W|_inf for the 3 x 3 of the view matrix (1 for the identity the scorer's camera has)
			float wn;
			{
				const float c0 = fabsf(vm[0]) + fabsf(vm[1]) + fabsf(vm[2]), c1 = fabsf(vm[4]) + fabsf(vm[5]) + fabsf(vm[6]), c2 = fabsf(vm[8]) + fabsf(vm[9]) + fabsf(vm[10]);
				const float r0 = fabsf(vm[0]) + fabsf(vm[4]) + fabsf(vm[8]), r1 = fabsf(vm[1]) + fabsf(vm[5]) + fabsf(vm[9]), r2 = fabsf(vm[2]) + fabsf(vm[6]) + fabsf(vm[10]);
				wn = fmaxf(c0, fmaxf(c1, c2)) * fmaxf(r0, fmaxf(r1, r2));
			}
			const float fx2 = 1.02f * wn * p.focal_x * p.focal_x, fy2 = 1.02f * wn * p.focal_y * p.focal_y;
			const float xmax = (float)(p.gx * FR_BLOCK_X + FR_BLOCK_X), ymax = (float)(p.gy * FR_BLOCK_Y + FR_BLOCK_Y);
			for (int vv = 0; vv < nv; vv++)
			{
				const fr_f3 po = has_w2c ? fr_world_to_cam(pw, s_wm + 12 * vv) : pw;
				const fr_f3 p_view = fr_xform4x3(po, vm);
				bool keep = live && !(p_view.z <= 0.001f);
				if (early)
				{
					const fr_f4 ph = fr_xform4x4(po, pm);
					const float p_w = 1.0f / (ph.w + 0.0000001f);
					const float px = ((ph.x * p_w + 1.0f) * (float)p.W - 1.0f) * 0.5f, py = ((ph.y * p_w + 1.0f) * (float)p.H - 1.0f) * 0.5f;   // ndc2pix in float: the slack covers it
					const float iz = 1.0f / p_view.z;
					const float jx = fminf(fabsf(p_view.x * iz) * 1.001f, lx), jy = fminf(fabsf(p_view.y * iz) * 1.001f, ly);
					const float kc = fx2 * (1.0f + jx * jx) + fy2 * (1.0f + jy * jy);
					const float rb = 3.0f * sqrtf(kc * tr * (iz * iz) + 0.7f) + 2.0f;
					const bool outside = (px + rb < 0.f) || (px - rb > xmax) || (py + rb < 0.f) || (py - rb > ymax);
					keep = keep && !outside;
				}
				const unsigned long long m = __ballot(keep);
				if (m)
				{
					uint32_t base = 0;
					if (lane == 0) base = atomicAdd(&s_np, (uint32_t)__popcll(m));
					base = __builtin_amdgcn_readfirstlane(base);
					if (keep) pairs[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)tid | ((uint32_t)vv << 8);
				}
			}
		}
		__syncthreads();
		// ---- phase B
		const uint32_t np = s_np;
		for (uint32_t e = tid; e < np; e += FR_THREADS)
		{
			const uint32_t pr = pairs[e];
			const int i = i0 + (int)(pr & 255u);
			const int vv = (int)(pr >> 8);
			const int v = v0 + vv;
			// mean, cov3D and colour of the Gaussian: three 16-byte loads of its packed static record when there is one (records
			// modes), thirteen scattered dwords otherwise -- the same values either way
			fr_f3 pw;
			float c3[6];
			float cg = 0.f;
			if constexpr (RC != 0)
			{
				constexpr int PSB = FrPackSize<(RC < 0 ? -RC : RC)>::value;
				const float4* pk = (const float4*)(ra.packed + (size_t)i * PSB);
				const float4 t0 = pk[0], t1 = pk[1], t2 = pk[2];
				pw = fr_f3{ t0.x, t0.y, t0.z };
				c3[0] = t0.w; c3[1] = t1.x; c3[2] = t1.y; c3[3] = t1.z; c3[4] = t1.w; c3[5] = t2.x;
				cg = t2.y + t2.z + t2.w;
			}
			else
			{
				pw = fr_f3{ p.means3D[3 * (size_t)i], p.means3D[3 * (size_t)i + 1], p.means3D[3 * (size_t)i + 2] };
#pragma unroll
				for (int k = 0; k < 6; k++) c3[k] = p.cov3D[6 * (size_t)i + k];
			}
			float wm[12];
			if (has_w2c)
			{
#pragma unroll
				for (int k = 0; k < 12; k++) wm[k] = s_wm[12 * vv + k];
			}
			const fr_f3 po = has_w2c ? fr_world_to_cam(pw, wm) : pw;
			const fr_splat s = fr_preprocess_one(po, c3, vm, pm, p.W, p.H, p.tanfovx, p.tanfovy, p.focal_x, p.focal_y, p.gx, p.gy);
			if (s.radius > 0)
			{
				const float o = p.opac[i];
				const uint32_t ext = fr_alpha_extent(s.conx, s.cony, s.conz, o);
				const uint32_t slot = atomicAdd(&s_n[vv], 1u);          // position in this workgroup's list of the view
				float4* dst = (float4*)(p.splat + (size_t)v * p.P + i);
				if (compact)
				{
					// {recA, recB} wait beside the list entry (the [V][P] splat region serves as [V][workgroup][256 G] here) until
					// phase C knows the Gaussian's rank among the workgroup's visible ones
					dst = (float4*)(p.splat + ((size_t)v * nblk + blockIdx.x) * cap + slot);
					const uint32_t local = (uint32_t)(i - (int)(blockIdx.x * cap));
					atomicOr(&s_bm[vv * W32 + (int)(local >> 5)], 1u << (local & 31u));
				}
				if constexpr (RC != 0)
				{
					// score-only mode: the scorer's {recA, recB} form straight away (see k_fisher_tile_v3)
					dst[0] = make_float4(s.px, s.py, __uint_as_float(ext), __builtin_amdgcn_logf(o));
					dst[1] = make_float4(-0.5f * s.conx, -s.cony, -0.5f * s.conz, cg);
				}
				else
				{
					dst[0] = make_float4(s.px, s.py, s.conx, s.cony);
					dst[1] = make_float4(s.conz, o, s.depth, __uint_as_float(ext));
				}
				// The reference lists the splat in every tile of its radius rectangle (rasterizer_impl.cu:70-111).  The scorer
				// only ever uses a list entry where alpha can reach 1/255, so the rectangle is cut down to the tiles that the
				// conservative alpha footprint (ext: half extents, rounded up) touches -- 16 % fewer keys to scatter, sort and
				// stream on the benchmark scene; the reference's count is still what out_num_rendered reports.
				uint32_t rx0 = s.rect.x0, rx1 = s.rect.x1, ry0 = s.rect.y0, ry1 = s.rect.y1;
				atomicAdd(&s_ref[vv], (rx1 - rx0) * (ry1 - ry0));
				{
					const float hx = __half2float(__ushort_as_half((unsigned short)(ext & 0xffffu)));
					const float hy = __half2float(__ushort_as_half((unsigned short)(ext >> 16)));
					if (hx < 0.f) { rx1 = rx0; ry1 = ry0; }                  // opacity <= 1/255: contributes nowhere
					else if (hx < 1e30f)
					{
						// pixel centres sit on integer coordinates: pixel x is inside when |x - px| <= hx
						const float inv = 1.0f / (float)FR_BLOCK_X;
						const int tx0 = (int)floorf((s.px - hx) * inv), tx1 = (int)floorf((s.px + hx) * inv) + 1;
						const int ty0 = (int)floorf((s.py - hy) * inv), ty1 = (int)floorf((s.py + hy) * inv) + 1;
						rx0 = (uint32_t)max((int)rx0, tx0); rx1 = (uint32_t)max((int)rx0, min((int)rx1, tx1));
						ry0 = (uint32_t)max((int)ry0, ty0); ry1 = (uint32_t)max((int)ry0, min((int)ry1, ty1));
					}
				}
				uint32_t* h = hist + (size_t)vv * p.T;
				for (uint32_t y = ry0; y < ry1; y++)
					for (uint32_t x = rx0; x < rx1; x++)
						atomicAdd(&h[y * p.gx + x], 1u);
				FrVisEntry en;
				en.idx = (uint32_t)i; en.depth_bits = fr_as_u32(s.depth);
				en.xy0 = rx0 | (ry0 << 16); en.xy1 = rx1 | (ry1 << 16);
				*(uint4*)(p.vis_list + ((size_t)v * nblk + blockIdx.x) * cap + slot) = *(const uint4*)&en;
			}
		}
		__syncthreads();
		if (tid == 0) s_np = 0;
		// (the next phase A only appends after its own ballots; the barrier at its end orders the reset)
		__syncthreads();
	}
	for (int vv = 0; vv < nv; vv++)
	{
		const int v = v0 + vv;
		uint32_t* cnt = p.tile_cnt + (size_t)v * p.T;
		const uint32_t* h = hist + (size_t)vv * p.T;
		for (int t = tid; t < p.T; t += FR_THREADS)
		{
			const uint32_t c = h[t];
			if (c) p.blk_base[((size_t)v * nblk + blockIdx.x) * p.T + t] = atomicAdd(&cnt[t], c);
		}
	}
	if (tid < nv)
	{
		const int v = v0 + tid;
		p.vis_n[(size_t)v * nblk + blockIdx.x] = s_n[tid];
		if (p.vis_count && s_n[tid]) atomicAdd(&p.vis_count[v], (int)s_n[tid]);
		if (p.num_rendered && s_ref[tid]) atomicAdd(&p.num_rendered[v], (int)s_ref[tid]);
	}
	if constexpr (RC != 0 && PHASE_C)
	{
		// ---- phase C: the entries and splat records were written by this workgroup (same CU, same L1): visible after a barrier
		__syncthreads();
		if (compact)
		{
			// exclusive prefix popcounts of the bitmaps: rank of a Gaussian = prefix of its word + popcount of the lower bits
			for (int t = tid; t < nv * W32; t += FR_THREADS)
			{
				const int vv = t / W32, w = t - vv * W32;
				uint32_t acc = 0;
				for (int u = 0; u < w; u++) acc += (uint32_t)__popc(s_bm[vv * W32 + u]);
				s_pf[t] = acc;
			}
			__syncthreads();
		}
		const size_t PV = (size_t)nblk * cap;
		for (int vv = 0; vv < nv; vv++)
		{
			const int v = v0 + vv;
			const uint32_t n = s_n[vv];
			FrVisEntry* list = p.vis_list + ((size_t)v * nblk + blockIdx.x) * cap;
			float wm[12];
#pragma unroll
			for (int k = 0; k < 12; k++) wm[k] = has_w2c ? s_wm[12 * vv + k] : 0.f;
			for (uint32_t e = tid; e < n; e += FR_THREADS)
			{
				const uint32_t idx = list[e].idx;
				if (compact)
				{
					const uint32_t local = idx - blockIdx.x * cap;
					const uint32_t rank = s_pf[vv * W32 + (int)(local >> 5)] + (uint32_t)__popc(s_bm[vv * W32 + (int)(local >> 5)] & ((1u << (local & 31u)) - 1u));
					const uint32_t slot = blockIdx.x * cap + rank;
					const float4* tmp = (const float4*)(p.splat + ((size_t)v * nblk + blockIdx.x) * cap + e);
					fr_fisher_record_one<(RC < 0 ? -RC : RC), false, (RC < 0)>(p, ra.H_inv, ra.hinv_stride, ra.packed, ra.recq, v, idx, vm, pm, wm, has_w2c,
					                                                           ra.comp + ((size_t)v * PV + slot) * 6, tmp);
					list[e].idx = slot;                                  // the keys carry the slot from here on
					if constexpr (RC < 0) ra.slot_idx[(size_t)v * PV + slot] = idx;   // (only k_fisher_tile_v3h goes back to the index)
				}
				else fr_fisher_record_one<(RC < 0 ? -RC : RC), false, (RC < 0)>(p, ra.H_inv, ra.hinv_stride, ra.packed, ra.recq, v, idx, vm, pm, wm, has_w2c);
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// The same front end with the compact scorer records written ONCE, in their final place (the product path of the score-only and
// out_H record modes).  k_preprocess_views<RC, true> parks every visible splat's {recA, recB} in global memory until its phase C
// knows the splat's slot, reads them back, and rewrites the list entry: 1.7x the bytes the records need
// (profiles/r02_n_pmc_all.txt: WRITE_SIZE 1.46 GB against 0.86 GB of records per 64-view step).  Here a batch of 256 survivors is
//   B   projected (the same arithmetic: fr_preprocess_one), and the visible ones are RANKED IN ORDER with wave ballots: the
//       survivor list of a workgroup is grouped by view and ascending in the Gaussian index inside a view, so a visible pair's
//       rank among its view's visible splats -- its slot, monotone in the index -- is the view's running count + the visible
//       pairs of that view in the lower waves + those in the lower lanes;
//   park their {recA, recB, list entry, rank} go to LDS (13 dwords x 256, structure of arrays), densely;
//   C   thread r takes the r-th visible pair of the batch at full lane occupancy: Jacobian rows / polynomial
//       (fr_fisher_record_one), and writes the 96-byte record to its slot and the 16-byte list entry beside it -- consecutive
//       threads, consecutive records.
// No bitmaps, no prefix popcounts, no second pass over the lists.  Records, lists and counts are identical to the other form's.
// DK (direct keys, FrParams::tile_cap > 0): every (view, tile) owns a fixed segment of the key buffer, so a workgroup needs no
// scan of the tile counts to place its keys: once its histogram is complete it claims its ranges (the same one atomic per
// non-empty (view, tile) that the other form uses for blk_base) and scatters the keys of its own visible lists right away,
// underneath the projection arithmetic of the other workgroups on the CU -- k_scatter_vis (latency-bound, 0.15 ms of the
// 64-view step on its own) and the blk_base array disappear, k_scan_tiles only builds the lists of long tiles.  The order of the
// keys inside a segment is arbitrary either way; the sort makes it the reference's.
template <int C, int AF, bool DK>
__global__ __launch_bounds__(FR_THREADS) void k_preprocess_views_c(FrParams p, FrRecordArgs ra)
{
	static_assert((C == 4 || C == 11) && AF >= 0 && AF <= 2 && !(AF == 1 && C != 4), "records modes: score form, A-form (4 columns), general out_H form");
	constexpr int RS = FrRecStride<C, AF, DK>::value; // float4 per compact record
	extern __shared__ uint32_t fr_dyn_lds[];     // hist[VC][T] | pairs[FR_THREADS * (VC + 1)] | wm[VC][12] | park[13][FR_THREADS] | DK: cursor[VC][T]
	const int VC = p.VC;
	uint32_t* hist = fr_dyn_lds;
	uint32_t* pairs = hist + (size_t)VC * p.T;
	float* s_wm = (float*)(pairs + FR_THREADS * (VC + 1));
	uint32_t* park = (uint32_t*)(s_wm + 12 * VC);
	__shared__ uint32_t s_n[FR_VC_MAX];          // visible pairs per view so far = the next slot of the view
	__shared__ uint32_t s_ref[FR_VC_MAX];
	__shared__ uint32_t s_ca[FR_VC_MAX * 4];     // phase A: survivors per (view, wave)
	// Two barriers per batch, not four: the ballot counts are double-buffered (a fast wave may already count batch k + 1 while a slow
	// one still ranks batch k), the views' running counts s_n move on behind the parking barrier (their readers are all in front of it,
	// their next readers behind the next batch's first barrier), and park[] is rewritten only behind that next first barrier -- which no
	// wave passes before every wave has finished phase C of batch k.
	__shared__ uint32_t s_wk2[2][FR_VC_MAX * 4]; // visible pairs per (view, wave) of a batch
	__shared__ uint32_t s_wtot2[2][4];           // ... per wave
	uint32_t batch_no = 0;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int v0 = blockIdx.y * VC;
	const int nv = min(VC, p.V - v0);
	const uint32_t nblk = gridDim.x;
	const uint32_t cap = (uint32_t)(FR_THREADS * p.G);
	for (int t = tid; t < nv * p.T; t += FR_THREADS) hist[t] = 0;
	const bool has_w2c = p.w2c != nullptr;
	if (has_w2c) for (int t = tid; t < nv * 12; t += FR_THREADS) s_wm[t] = p.w2c[16 * (size_t)(v0 + t / 12) + (t % 12)];
	if (tid < FR_VC_MAX) { s_n[tid] = 0; s_ref[tid] = 0; }
	float vm[16], pm[16];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	const size_t PV = (size_t)nblk * cap;
	const unsigned long long lt = (1ull << lane) - 1ull;
	__syncthreads();
	// Survivors wait in `pairs` until 256 of them make a full batch for phases B and C (one 256-Gaussian round leaves ~290 on the
	// bench workload: a full batch and a nearly empty one, had every round been flushed by itself); what is left over after a round
	// (< 256) moves to the front of the list, the last round flushes.  A pending entry = index inside the workgroup's share | view << 16.
	const int iw = blockIdx.x * p.G * FR_THREADS;       // the workgroup's first Gaussian
	const bool early = ra.early != 0;
	const float lx = 1.3f * p.tanfovx, ly = 1.3f * p.tanfovy;
	float wn;                                           // |W|_2^2 <= |W|_1 |W|_inf for the 3 x 3 of the view matrix (1 for the identity the scorer's camera has)
	{
		const float c0 = fabsf(vm[0]) + fabsf(vm[1]) + fabsf(vm[2]), c1 = fabsf(vm[4]) + fabsf(vm[5]) + fabsf(vm[6]), c2 = fabsf(vm[8]) + fabsf(vm[9]) + fabsf(vm[10]);
		const float r0 = fabsf(vm[0]) + fabsf(vm[4]) + fabsf(vm[8]), r1 = fabsf(vm[1]) + fabsf(vm[5]) + fabsf(vm[9]), r2 = fabsf(vm[2]) + fabsf(vm[6]) + fabsf(vm[10]);
		wn = fmaxf(c0, fmaxf(c1, c2)) * fmaxf(r0, fmaxf(r1, r2));
	}
	const float fx2 = 1.02f * wn * p.focal_x * p.focal_x, fy2 = 1.02f * wn * p.focal_y * p.focal_y;
	const float xmax = (float)(p.gx * FR_BLOCK_X + FR_BLOCK_X), ymax = (float)(p.gy * FR_BLOCK_Y + FR_BLOCK_Y);
	// ---- group test: bit (g nv + vv) of `skipm` = no Gaussian of the workgroup's g-th round of 256 can survive phase A in view vv.
	// With the Gaussians laid out along a Z-curve (fr_fisher_cfg.order) the 256 of a round are neighbours in space: two thirds of the
	// (round, view) pairs of the benchmark hold no survivor at all.  The round's bounds (k_pack_static: box of the means, largest
	// trace) give a sphere (c, R) in world space, hence in the view's camera frame (w2c is rigid).  A Gaussian survives only in front
	// of the near plane with its centre's pixel within rb of the tile grid, rb = 3 sqrt(kc tr / z^2 + 0.7) + 2 <= a / z + c0 with
	// a = 3 sqrt(kc_max tr_max), c0 = 3 sqrt(0.7) + 2 (sqrt(u + v) <= sqrt(u) + sqrt(v)).  Where the clip-space w IS the view depth z
	// (every pinhole projection: row 3 of proj = row 2 of view), `px + rb < 0` for z > 0 follows from the LINEAR inequality
	// W hx + (W - 1 + 2 c0) z + 2 a < 0 in the camera-frame point, so the whole sphere is out when it holds at the centre with |A| R to
	// spare; likewise for the other three sides, and for z + R <= 0.001.  Slack: c0 = 6 instead of 4.51, 1 % on a and R.
	unsigned long long skipm = 0ull;
	if (early && ra.grp != nullptr && p.G * nv <= 64)
	{
		const float wrow = fabsf(pm[3] - vm[2]) + fabsf(pm[7] - vm[6]) + fabsf(pm[11] - vm[10]) + fabsf(pm[15] - vm[14]);
		const float wmag = fabsf(vm[2]) + fabsf(vm[6]) + fabsf(vm[10]) + fabsf(vm[14]);
		const bool pinhole = wrow <= 1e-6f * wmag;                     // clip w == view z
		bool skip = false;
		const int g = lane / nv, vv = lane - g * nv;
		const int blk = blockIdx.x * p.G + g;
		if (pinhole && g < p.G && blk * FR_THREADS < p.P)
		{
#pragma clang fp contract(fast)
			const float4 b0 = ra.grp[2 * (size_t)blk], b1 = ra.grp[2 * (size_t)blk + 1];
			const fr_f3 cw = { 0.5f * (b0.x + b1.x), 0.5f * (b0.y + b1.y), 0.5f * (b0.z + b1.z) };
			const float ex = b1.x - cw.x, ey = b1.y - cw.y, ez = b1.z - cw.z;
			float R = 1.01f * __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) + 1e-6f;
			const fr_f3 c = has_w2c ? fr_world_to_cam(cw, s_wm + 12 * vv) : cw;
			if (has_w2c)
			{
				// the sphere's radius in the camera frame: times |M|_2 <= sqrt(|M^T M|_inf) of the pose's 3 x 3 (1 for a rigid pose)
				const float* m = s_wm + 12 * vv;
				const float g00 = m[0] * m[0] + m[4] * m[4] + m[8] * m[8], g11 = m[1] * m[1] + m[5] * m[5] + m[9] * m[9], g22 = m[2] * m[2] + m[6] * m[6] + m[10] * m[10];
				const float g01 = fabsf(m[0] * m[1] + m[4] * m[5] + m[8] * m[9]), g02 = fabsf(m[0] * m[2] + m[4] * m[6] + m[8] * m[10]), g12 = fabsf(m[1] * m[2] + m[5] * m[6] + m[9] * m[10]);
				R *= __builtin_amdgcn_sqrtf(fmaxf(g00 + g01 + g02, fmaxf(g01 + g11 + g12, g02 + g12 + g22))) * 1.00001f;
			}
			const float kcm = fx2 * (1.0f + lx * lx) + fy2 * (1.0f + ly * ly);
			const float a2 = 2.0f * 1.01f * 3.0f * __builtin_amdgcn_sqrtf(kcm * b0.w);
			const float c0 = 6.0f;
			const float Wf = (float)p.W, Hf = (float)p.H;
			// z and the four sides as linear forms A . c + B of the camera-frame point (view-space z = vm row 2)
			const float zc = vm[2] * c.x + vm[6] * c.y + vm[10] * c.z + vm[14];
			const float zn = __builtin_amdgcn_sqrtf(vm[2] * vm[2] + vm[6] * vm[6] + vm[10] * vm[10]);
			auto side = [&](float sx, float sy, float sz, float sw, float kz, float add) -> bool {
				// L(q) = s . q + sw + kz z(q) + add ;  returns L(c) + |grad L| R < 0  (the whole sphere on the negative side)
				const float Ax = sx + kz * vm[2], Ay = sy + kz * vm[6], Az = sz + kz * vm[10];
				const float Lc = Ax * c.x + Ay * c.y + Az * c.z + (sw + kz * vm[14]) + add;
				return Lc + __builtin_amdgcn_sqrtf(Ax * Ax + Ay * Ay + Az * Az) * R < 0.f;
			};
			// px + rb < 0            <=  W hx + (W - 1 + 2 c0) z + 2 a < 0
			// px - rb > xmax         <= -W hx - (W - 1 - 2 xmax - 2 c0) z + 2 a < 0
			const bool left = side(Wf * pm[0], Wf * pm[4], Wf * pm[8], Wf * pm[12], Wf - 1.0f + 2.0f * c0, a2);
			const bool right = side(-Wf * pm[0], -Wf * pm[4], -Wf * pm[8], -Wf * pm[12], -(Wf - 1.0f - 2.0f * xmax - 2.0f * c0), a2);
			const bool top = side(Hf * pm[1], Hf * pm[5], Hf * pm[9], Hf * pm[13], Hf - 1.0f + 2.0f * c0, a2);
			const bool bottom = side(-Hf * pm[1], -Hf * pm[5], -Hf * pm[9], -Hf * pm[13], -(Hf - 1.0f - 2.0f * ymax - 2.0f * c0), a2);
			const bool behind = zc + zn * R <= 0.001f;
			skip = left || right || top || bottom || behind;            // (a NaN bound compares false everywhere: the round is kept)
		}
		skipm = __builtin_amdgcn_ballot_w64(skip);
	}
	uint32_t pend = 0;                                  // survivors waiting, uniform
	for (int g = 0; g < p.G; g++)
	{
		const int i0 = iw + g * FR_THREADS;
		if (i0 >= p.P) break;
		const bool last_round = g == p.G - 1 || i0 + FR_THREADS >= p.P;
		uint32_t np = 0;
		const uint32_t skipv = (uint32_t)(skipm >> (g * nv)) & ((1u << nv) - 1u);         // views that cannot see this round
		// ---- phase A (the tests of k_preprocess_views): near-plane + early frustum test, survivors appended to `pairs`
		if (skipv != (1u << nv) - 1u)
		{
			const int i = i0 + tid;
			const bool live = i < p.P;
			fr_f3 pw = { 0.f, 0.f, 0.f };
			float tr = 0.f;
			if (live) { const float4 m4 = ra.mt[i]; pw = fr_f3{ m4.x, m4.y, m4.z }; tr = m4.w; }
			// The survivor list has to come out in (view, Gaussian index) order: a visible pair's slot is its rank among the
			// visible pairs of its view, and the keys rely on that rank being monotone in the index (ties of equal depth).  So the
			// waves do not append with an atomic cursor: every wave counts its survivors per view, the counts are scanned in
			// (view, wave) order, and each wave then writes at its fixed offsets.
			uint32_t keepbits = 0;
			for (int vv = 0; vv < nv; vv++)
			{
				if ((skipv >> vv) & 1u) { if (lane == 0) s_ca[vv * 4 + wave] = 0u; continue; }       // (uniform: the whole round is out of this view)
				const fr_f3 po = has_w2c ? fr_world_to_cam(pw, s_wm + 12 * vv) : pw;
				const fr_f3 p_view = fr_xform4x3(po, vm);        // (the near-plane decision: the reference's arithmetic, unfused)
				bool keep = live && !(p_view.z <= 0.001f);
				if (early)
				{
					// (a bound, not the projection: fused multiply-adds, v_rcp_f32 / v_sqrt_f32 -- a few 1e-7 off; the slack below is
					// 1e-3 and two pixels)
#pragma clang fp contract(fast)
					const float hx = __builtin_fmaf(pm[8], po.z, __builtin_fmaf(pm[4], po.y, __builtin_fmaf(pm[0], po.x, pm[12])));
					const float hy = __builtin_fmaf(pm[9], po.z, __builtin_fmaf(pm[5], po.y, __builtin_fmaf(pm[1], po.x, pm[13])));
					const float hw = __builtin_fmaf(pm[11], po.z, __builtin_fmaf(pm[7], po.y, __builtin_fmaf(pm[3], po.x, pm[15])));
					const float p_w = __builtin_amdgcn_rcpf(hw + 0.0000001f);
					const float px = ((hx * p_w + 1.0f) * (float)p.W - 1.0f) * 0.5f, py = ((hy * p_w + 1.0f) * (float)p.H - 1.0f) * 0.5f;
					const float iz = __builtin_amdgcn_rcpf(p_view.z);
					const float jx = fminf(fabsf(p_view.x * iz) * 1.001f, lx), jy = fminf(fabsf(p_view.y * iz) * 1.001f, ly);
					const float kc = fx2 * (1.0f + jx * jx) + fy2 * (1.0f + jy * jy);
					const float rb = 3.0f * __builtin_amdgcn_sqrtf(kc * tr * (iz * iz) + 0.7f) * 1.000001f + 2.0f;
					const bool outside = (px + rb < 0.f) || (px - rb > xmax) || (py + rb < 0.f) || (py - rb > ymax);
					keep = keep && !outside;
				}
				keepbits |= (keep ? 1u : 0u) << vv;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
				if (lane == 0) s_ca[vv * 4 + wave] = (uint32_t)__popcll(m);
			}
			__syncthreads();
			// exclusive scan of the 4 nv counts (every wave for itself: lane l holds entry l)
			uint32_t cnt_l = lane < 4 * nv ? s_ca[lane] : 0u;
			uint32_t inc = cnt_l;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1)
			{
				const uint32_t y = (uint32_t)__shfl_up((int)inc, d, 64);
				if (lane >= d) inc += y;
			}
			const uint32_t exc = inc - cnt_l;
			np = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
			for (int vv = 0; vv < nv; vv++)
			{
				const bool keep = (keepbits >> vv) & 1u;
				const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
				const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)exc, vv * 4 + wave);
				if (keep) pairs[pend + base + (uint32_t)__popcll(m & lt)] = (uint32_t)(g * FR_THREADS + tid) | ((uint32_t)vv << 16);
			}
		}
		__syncthreads();
		FR_ABL(if (p.ablate == 35) np = 0;)                      // 35: phase A only
		pend += np;
		const uint32_t ndo = last_round ? pend : (pend & ~(uint32_t)(FR_THREADS - 1));     // whole batches, all of it in the last round
		for (uint32_t e0 = 0; e0 < ndo; e0 += FR_THREADS)
		{
			const uint32_t e = e0 + (uint32_t)tid;
			const bool active = e < ndo;
			// ---- phase B: projection of one survivor
			int i = 0, vv = 0;
			fr_splat sp;
			sp.radius = 0;
			float o = 0.f, cg = 0.f;
			uint32_t ext = 0;
			if (active)
			{
					const uint32_t pr = pairs[e];
				i = iw + (int)(pr & 0xffffu);
				vv = (int)(pr >> 16);
				constexpr int PSB = FrPackSize<C>::value;
				const float4* pk = (const float4*)(ra.packed + (size_t)i * PSB);
				const float4 t0 = pk[0], t1 = pk[1], t2 = pk[2];
				// (the opacity with them: loaded behind the projection, for the visible ones only, it was a load nothing overlapped)
				const float o_pre = p.opac[p.order ? p.order[i] : (uint32_t)i];
				const fr_f3 pw = fr_f3{ t0.x, t0.y, t0.z };
				float c3[6];
				c3[0] = t0.w; c3[1] = t1.x; c3[2] = t1.y; c3[3] = t1.z; c3[4] = t1.w; c3[5] = t2.x;
				cg = t2.y + t2.z + t2.w;
				float wm[12];
				if (has_w2c)
				{
#pragma unroll
					for (int k = 0; k < 12; k++) wm[k] = s_wm[12 * vv + k];
				}
				const fr_f3 po = has_w2c ? fr_world_to_cam(pw, wm) : pw;
				sp = fr_preprocess_one(po, c3, vm, pm, p.W, p.H, p.tanfovx, p.tanfovy, p.focal_x, p.focal_y, p.gx, p.gy);
				if (sp.radius > 0) { o = o_pre; ext = fr_alpha_extent(sp.conx, sp.cony, sp.conz, o); }
			}
			const bool vis = sp.radius > 0;
			// ---- ordered ranks.  The pending list is a sequence of rounds, each sorted by (view, index), so the pairs of one view stand
			// in index order in it, and in every batch: a visible pair's rank among its view's visible pairs is the view's count of the
			// earlier batches (s_n) + the same-view visible pairs of the lower waves + those of the lower lanes.  `pos` (its place among
			// ALL visible pairs of the batch) only packs the batch for phase C.
			uint32_t* s_wk = s_wk2[batch_no & 1u];
			uint32_t* s_wtot = s_wtot2[batch_no & 1u];
			batch_no++;
			const unsigned long long mv = __builtin_amdgcn_ballot_w64(vis);
			if (lane == 0) s_wtot[wave] = (uint32_t)__popcll(mv);
			uint32_t below = 0;                                     // same-view visible pairs in the lower lanes of this wave
			for (int k = 0; k < nv; k++)
			{
				const unsigned long long mk = __builtin_amdgcn_ballot_w64(vis && vv == k);
				if (lane == 0) s_wk[k * 4 + wave] = (uint32_t)__popcll(mk);
				if (vv == k) below = (uint32_t)__popcll(mk & lt);
			}
			__syncthreads();
			uint32_t rank = 0, pos = 0, nvis = 0;
			{
#pragma unroll
				for (int w = 0; w < 4; w++) { const uint32_t c = s_wtot[w]; nvis += c; pos += (w < wave) ? c : 0u; }
				pos += (uint32_t)__popcll(mv & lt);
			}
			if (vis)
			{
				rank = s_n[vv] + below;
				for (int w = 0; w < wave; w++) rank += s_wk[vv * 4 + w];
			}
			if (vis)
			{
				// tile rectangle: the reference's radius rectangle (rasterizer_impl.cu:70-111) cut down to the tiles the conservative
				// alpha footprint reaches, exactly as in k_preprocess_views
				uint32_t rx0 = sp.rect.x0, rx1 = sp.rect.x1, ry0 = sp.rect.y0, ry1 = sp.rect.y1;
				atomicAdd(&s_ref[vv], (rx1 - rx0) * (ry1 - ry0));
				{
					const float hx = __half2float(__ushort_as_half((unsigned short)(ext & 0xffffu)));
					const float hy = __half2float(__ushort_as_half((unsigned short)(ext >> 16)));
					if (hx < 0.f) { rx1 = rx0; ry1 = ry0; }
					else if (hx < 1e30f)
					{
						const float inv = 1.0f / (float)FR_BLOCK_X;
						const int tx0 = (int)floorf((sp.px - hx) * inv), tx1 = (int)floorf((sp.px + hx) * inv) + 1;
						const int ty0 = (int)floorf((sp.py - hy) * inv), ty1 = (int)floorf((sp.py + hy) * inv) + 1;
						rx0 = (uint32_t)max((int)rx0, tx0); rx1 = (uint32_t)max((int)rx0, min((int)rx1, tx1));
						ry0 = (uint32_t)max((int)ry0, ty0); ry1 = (uint32_t)max((int)ry0, min((int)ry1, ty1));
					}
				}
				uint32_t* h = hist + (size_t)vv * p.T;
				if constexpr (DK)
				{
					// Fixed segments: the keys also say which of the tile's four 16 x 4 strips the footprint [px -+ hx] x [py -+ hy]
					// reaches (the test a wave of the tile kernels would otherwise make per key, from a gathered record), so
					// the rectangle is kept in strip rows from here on, cut to the strips / tile columns that test admits.
					const float hx = __half2float(__ushort_as_half((unsigned short)(ext & 0xffffu)));
					const float hy = __half2float(__ushort_as_half((unsigned short)(ext >> 16)));
					ry0 *= 4u; ry1 *= 4u;
					if (hx >= 0.f && hx < 1e30f && rx1 > rx0 && ry1 > ry0)
					{
						const int tx0 = (int)ceilf((sp.px - hx - 15.0f) * 0.0625f), sy0 = (int)ceilf((sp.py - hy - 3.0f) * 0.25f);
						const int sy1 = (int)floorf((sp.py + hy) * 0.25f) + 1;
						rx0 = (uint32_t)max((int)rx0, tx0); rx1 = max(rx0, rx1);
						ry0 = (uint32_t)max((int)ry0, sy0); ry1 = (uint32_t)max((int)ry0, min((int)ry1, sy1));
					}
					if (rx1 > rx0 && ry1 > ry0)
						for (uint32_t y = ry0 >> 2; y <= (ry1 - 1u) >> 2; y++)
							for (uint32_t x = rx0; x < rx1; x++)
								atomicAdd(&h[y * p.gx + x], 1u);
				}
				else
				for (uint32_t y = ry0; y < ry1; y++)
					for (uint32_t x = rx0; x < rx1; x++)
						atomicAdd(&h[y * p.gx + x], 1u);
				park[0 * FR_THREADS + pos] = __float_as_uint(sp.px);
				park[1 * FR_THREADS + pos] = __float_as_uint(sp.py);
				park[2 * FR_THREADS + pos] = ext;
				park[3 * FR_THREADS + pos] = __float_as_uint(__builtin_amdgcn_logf(o));
				park[4 * FR_THREADS + pos] = __float_as_uint(-0.5f * sp.conx);
				park[5 * FR_THREADS + pos] = __float_as_uint(-sp.cony);
				park[6 * FR_THREADS + pos] = __float_as_uint(-0.5f * sp.conz);
				park[7 * FR_THREADS + pos] = __float_as_uint(cg);
				park[8 * FR_THREADS + pos] = (uint32_t)i;
				park[9 * FR_THREADS + pos] = fr_as_u32(sp.depth);
				park[10 * FR_THREADS + pos] = rx0 | (ry0 << 16);
				park[11 * FR_THREADS + pos] = rx1 | (ry1 << 16);
				park[12 * FR_THREADS + pos] = rank | ((uint32_t)vv << 16);
			}
			__syncthreads();
			if (tid < nv) s_n[tid] += (s_wk[tid * 4] + s_wk[tid * 4 + 1]) + (s_wk[tid * 4 + 2] + s_wk[tid * 4 + 3]);     // (every rank of this batch is taken)
			// ---- phase C: the r-th visible pair of the batch
			FR_ABL(if (p.ablate == 36) nvis = 0;)                   // 36: no phase C (no records, no list entries)
			if ((uint32_t)tid < nvis)
			{
					const int r = tid;
				float4 ab[2];
				ab[0] = make_float4(__uint_as_float(park[0 * FR_THREADS + r]), __uint_as_float(park[1 * FR_THREADS + r]),
				                    __uint_as_float(park[2 * FR_THREADS + r]), __uint_as_float(park[3 * FR_THREADS + r]));
				ab[1] = make_float4(__uint_as_float(park[4 * FR_THREADS + r]), __uint_as_float(park[5 * FR_THREADS + r]),
				                    __uint_as_float(park[6 * FR_THREADS + r]), __uint_as_float(park[7 * FR_THREADS + r]));
				const uint32_t idx = park[8 * FR_THREADS + r];
				const uint32_t rk = park[12 * FR_THREADS + r];
				const int cvv = (int)(rk >> 16);
				const uint32_t slot = blockIdx.x * cap + (rk & 0xffffu);
				const int v = v0 + cvv;
				float wm[12];
#pragma unroll
				for (int k = 0; k < 12; k++) wm[k] = has_w2c ? s_wm[12 * cvv + k] : 0.f;
				float4* rec_out = ra.comp + ((size_t)v * PV + slot) * RS;
				FR_ABL(if (p.ablate == 37) rec_out = ra.comp + (size_t)tid * RS;)     // 37: the records' arithmetic without their HBM traffic
				if constexpr (AF == 2) fr_fisher_record_general<C>(p, ra.packed, v, idx, vm, pm, wm, has_w2c, rec_out, ab);
				else fr_fisher_record_one<C, false, (AF == 1), (AF == 0 && DK)>(p, ra.H_inv, ra.hinv_stride, ra.packed, ra.recq, v, idx, vm, pm, wm, has_w2c, rec_out, ab);
				if constexpr (DK)
				{
					// 8-byte list entry {depth, x0 | y0 << 8 | width << 16 | height << 24}, y0 / height in strip rows (tile grids up to
					// 255 x 63: fr_fisher_views); the slot is the entry's place in the list
					const uint32_t xy0 = park[10 * FR_THREADS + r], xy1 = park[11 * FR_THREADS + r];
					const uint32_t rect = (xy0 & 255u) | ((xy0 >> 16) << 8) | (((xy1 & 0xffffu) - (xy0 & 0xffffu)) << 16) | (((xy1 >> 16) - (xy0 >> 16)) << 24);
					((uint2*)p.vis_list)[((size_t)v * nblk + blockIdx.x) * cap + (rk & 0xffffu)] = make_uint2(park[9 * FR_THREADS + r], rect);
				}
				else
				{
					FrVisEntry en;
					en.idx = slot; en.depth_bits = park[9 * FR_THREADS + r];          // the keys carry the slot
					en.xy0 = park[10 * FR_THREADS + r]; en.xy1 = park[11 * FR_THREADS + r];
					*(uint4*)(p.vis_list + ((size_t)v * nblk + blockIdx.x) * cap + (rk & 0xffffu)) = *(const uint4*)&en;
				}
				if constexpr (AF != 0) ra.slot_idx[(size_t)v * PV + slot] = p.order ? p.order[idx] : idx;    // (the out_H kernels go back to the caller's index)
			}
		}
		// what is left over moves to the front of the list
		const uint32_t rem = pend - ndo;
		if (rem != 0u && ndo != 0u)
		{
			uint32_t t = 0;
			if ((uint32_t)tid < rem) t = pairs[ndo + tid];
			__syncthreads();
			if ((uint32_t)tid < rem) pairs[tid] = t;
			__syncthreads();
		}
		pend = rem;
	}
	__syncthreads();                                 // every histogram add, every list entry and the last update of s_n are done and visible
	if constexpr (DK)
	{
		uint32_t* cursor = park + 13 * FR_THREADS;
		// (the barrier that ended the last batch: every histogram add and every list entry of this workgroup is done and visible)
		// The workgroup's visible lists, all its views laid end to end, are swept NB x 256 entries at a time; the first round's
		// loads are issued before the range claims, every later round's before the round in hand is scattered.
		uint32_t vend[FR_VC_MAX];
		uint32_t ntot = 0;
#pragma unroll
		for (int k = 0; k < FR_VC_MAX; k++) { if (k < nv) ntot += s_n[k]; vend[k] = ntot; }
		FR_ABL(if (p.ablate == 31) ntot = 0;)
		constexpr int NB = 4;
		uint2 en[NB], nx[NB];
		uint32_t ev[NB], nxv[NB];
		auto load_round = [&](uint32_t e0, uint2 (&de)[NB], uint32_t (&dv)[NB])
		{
#pragma unroll
			for (int b = 0; b < NB; b++)
			{
				const uint32_t e = e0 + b * FR_THREADS + tid;
				uint32_t vv = 0, lo = 0;
#pragma unroll
				for (int k = 0; k < FR_VC_MAX - 1; k++) if (e >= vend[k]) { vv = k + 1; lo = vend[k]; }
				dv[b] = vv;
				de[b] = make_uint2(0u, 0u);                                  // empty rect
				FR_ABL(if (p.ablate == 33) { if (e < ntot) de[b] = make_uint2(e, (e & 7u) | ((e >> 3 & 15u) << 10) | (2u << 16) | (2u << 24)); } else)
				if (e < ntot) de[b] = ((const uint2*)p.vis_list)[((size_t)(v0 + vv) * nblk + blockIdx.x) * cap + (e - lo)];
			}
		};
		load_round(0u, en, ev);
		// claims: four bins per thread and trip, their atomics in flight together
		const int nb = nv * p.T;
		for (int b0 = 0; b0 < nb; b0 += 4 * FR_THREADS)
		{
			uint32_t c[4], at[4], gt[4];
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const int b = b0 + k * FR_THREADS + tid;
				c[k] = b < nb ? hist[b] : 0u;
				gt[k] = (uint32_t)(v0 + b / p.T) * (uint32_t)p.T + (uint32_t)(b % p.T);
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				at[k] = 0u;
				FR_ABL(if (p.ablate != 34))
				if (c[k]) at[k] = atomicAdd(&p.tile_cnt[gt[k]], c[k]);
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
			{
				const int b = b0 + k * FR_THREADS + tid;
				if (b < nb)
				{
					// over capacity: no key of this range is written -- k_tile_lists raises the flag from the count
					hist[b] = (c[k] && at[k] + c[k] <= p.tile_cap) ? gt[k] * p.tile_cap + at[k] : 0xffffffffu;
					cursor[b] = 0u;
				}
			}
		}
		__syncthreads();
		for (uint32_t e0 = 0; e0 < ntot; e0 += NB * FR_THREADS)
		{
			const bool more = e0 + NB * FR_THREADS < ntot;
			if (more) load_round(e0 + NB * FR_THREADS, nx, nxv);
#pragma unroll
			for (int b = 0; b < NB; b++)
			{
				// key = depth | slot << 4 | strips: the slot is the entry's place in its list (monotone in the Gaussian index, so ties of
				// equal depth sort like the reference's), the low four bits say which strips of THIS tile the footprint reaches
				const uint32_t e = e0 + b * FR_THREADS + tid;
				const uint32_t lo = ev[b] ? vend[ev[b] - 1] : 0u;
				const uint64_t key = ((uint64_t)en[b].x << 32) | ((blockIdx.x * cap + (e - lo)) << 4);
				const uint32_t r = en[b].y;
				const uint32_t x0 = r & 255u, s0 = (r >> 8) & 255u, x1 = x0 + ((r >> 16) & 255u), s1 = s0 + (r >> 24);     // strip rows [s0, s1)
				const uint32_t* hb = hist + (size_t)ev[b] * p.T;
				uint32_t* cu = cursor + (size_t)ev[b] * p.T;
				if (x1 > x0 && s1 > s0)
				for (uint32_t y = s0 >> 2; y <= (s1 - 1u) >> 2; y++)
				{
					const uint32_t a = max(s0, 4u * y) - 4u * y, z = min(s1, 4u * y + 4u) - 4u * y;      // strips [a, z) of tile row y
					const uint64_t ky = key | (uint64_t)(((1u << z) - 1u) & ~((1u << a) - 1u));
					for (uint32_t x = x0; x < x1; x++)
					{
						const uint32_t t = y * p.gx + x;
						const uint32_t base = hb[t];
						const uint32_t k = atomicAdd(&cu[t], 1u);
						FR_ABL(if (p.ablate != 32 && p.ablate != 33))
						if (base != 0xffffffffu) p.keys[base + k] = ky;
					}
				}
			}
			if (more)
			{
#pragma unroll
				for (int b = 0; b < NB; b++) { en[b] = nx[b]; ev[b] = nxv[b]; }
			}
		}
	}
	else
	{
		for (int vv = 0; vv < nv; vv++)
		{
			const int v = v0 + vv;
			uint32_t* cnt = p.tile_cnt + (size_t)v * p.T;
			const uint32_t* h = hist + (size_t)vv * p.T;
			for (int t = tid; t < p.T; t += FR_THREADS)
			{
				const uint32_t c = h[t];
				if (c) p.blk_base[((size_t)v * nblk + blockIdx.x) * p.T + t] = atomicAdd(&cnt[t], c);
			}
		}
	}
	if (tid < nv)
	{
		const int v = v0 + tid;
		p.vis_n[(size_t)v * nblk + blockIdx.x] = s_n[tid];
		if (p.vis_count && s_n[tid]) atomicAdd(&p.vis_count[v], (int)s_n[tid]);
		if (p.num_rendered && s_ref[tid]) atomicAdd(&p.num_rendered[v], (int)s_ref[tid]);
	}
}
// Key scatter of the multi-view front end: one workgroup per (preprocess workgroup, view), all lanes busy.
__global__ __launch_bounds__(FR_THREADS) void k_scatter_vis(FrParams p)
{
	extern __shared__ uint32_t fr_dyn_lds[];     // s_cnt[T] | s_base[T]
	uint32_t* s_cnt = fr_dyn_lds;
	uint32_t* s_base = fr_dyn_lds + p.T;
	if (p.status[1]) return;
	const int tid = threadIdx.x;
	const int v = blockIdx.y;
	const uint32_t nblk = gridDim.x;
	const uint32_t n = p.vis_n[(size_t)v * nblk + blockIdx.x];
	if (n == 0) return;
	const uint32_t* off = p.tile_off + (size_t)v * p.T;
	const uint32_t* bb = p.blk_base + ((size_t)v * nblk + blockIdx.x) * p.T;
	for (int t = tid; t < p.T; t += FR_THREADS) { s_cnt[t] = 0; s_base[t] = off[t] + bb[t]; }   // bb[t] is only used where this workgroup counted > 0
	__syncthreads();
	const uint4* list = (const uint4*)(p.vis_list + ((size_t)v * nblk + blockIdx.x) * (size_t)(FR_THREADS * p.G));
	constexpr int NB = 4;
	for (uint32_t e0 = 0; e0 < n; e0 += NB * FR_THREADS)
	{
		uint4 en[NB];
#pragma unroll
		for (int b = 0; b < NB; b++)
		{
			const uint32_t e = e0 + b * FR_THREADS + tid;
			en[b] = e < n ? list[e] : make_uint4(0u, 0u, 0u, 0u);       // empty rect
		}
#pragma unroll
		for (int b = 0; b < NB; b++)
		{
			const uint64_t key = ((uint64_t)en[b].y << 32) | en[b].x;
			const uint32_t x0 = en[b].z & 0xffffu, y0 = en[b].z >> 16, x1 = en[b].w & 0xffffu, y1 = en[b].w >> 16;
			for (uint32_t y = y0; y < y1; y++)
				for (uint32_t x = x0; x < x1; x++)
				{
					const uint32_t t = y * p.gx + x;
					p.keys[s_base[t] + atomicAdd(&s_cnt[t], 1u)] = key;
				}
		}
	}
}

// XCD-aware (tile, view) of a 1-D grid of T*V workgroups: workgroups b and b+8 share an XCD (and its L2), so every
// XCD is given whole views -- the tiles of one view then gather that view's per-splat records through one L2.
// Placement is a speed matter only; the map is a bijection whenever V % 8 == 0 and the plain one otherwise.
__device__ __forceinline__ void fr_tile_of_block(const FrParams& p, uint32_t& tile, int& v)
{
	const uint32_t L = blockIdx.x;
	if ((p.V & 7) == 0)
	{
		const uint32_t xcd = L & 7u, q = L >> 3;
		v = (int)((q / (uint32_t)p.T) * 8u + xcd);
		if (p.view_perm) v = (int)p.view_perm[v];           // views dealt by weight (fr_deal_views)
		tile = q % (uint32_t)p.T;
	}
	else
	{
		v = (int)(L / (uint32_t)p.T);
		tile = L % (uint32_t)p.T;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Ascending-only bitonic network ("flip" form): works for any n without padding, because a comparator whose
// upper element lies beyond n would compare against +inf and never swap.
// Two network stages per pass over the keys: a thread takes the four keys that two consecutive stages connect, runs
// both compare-exchange layers in registers and writes them back -- half the LDS traffic and half the barriers of the
// one-stage-per-pass form.  Positions at or beyond n hold a virtual +inf (never stored).
__device__ __forceinline__ void fr_cx(uint64_t& lo, uint64_t& hi)
{
	const uint64_t a = lo, b = hi;
	const bool sw = a > b;
	lo = sw ? b : a; hi = sw ? a : b;
}
template <typename KeyPtr>
__device__ __forceinline__ uint64_t fr_ldk(KeyPtr keys, uint32_t i, uint32_t n) { return i < n ? keys[i] : ~0ull; }
template <typename KeyPtr>
__device__ __forceinline__ void fr_stk(KeyPtr keys, uint32_t i, uint32_t n, uint64_t v) { if (i < n) keys[i] = v; }

template <typename KeyPtr>
__device__ __forceinline__ void fr_bitonic(KeyPtr keys, uint32_t n, int tid, const uint32_t FR_NT = FR_THREADS)
{
	uint32_t n_pad = 1;
	while (n_pad < n) n_pad <<= 1;
	if (n_pad < 2) return;
	// k = 2: one stage
	for (uint32_t t = tid; t < (n_pad >> 1); t += FR_NT)
	{
		uint64_t a = fr_ldk(keys, 2 * t, n), b = fr_ldk(keys, 2 * t + 1, n);
		fr_cx(a, b);
		fr_stk(keys, 2 * t, n, a); fr_stk(keys, 2 * t + 1, n, b);
	}
	__syncthreads();
	const uint32_t groups = n_pad >> 2;
	for (uint32_t k = 4, lk = 2; k <= n_pad; k <<= 1, lk++)
	{
		// flip stage (i <-> i ^ (k-1)) fused with the j = k/4 stage: a < b = a + k/4 < c = d - k/4 < d = a ^ (k-1)
		{
			const uint32_t q = k >> 2, qm = q - 1u;
			for (uint32_t g = tid; g < groups; g += FR_NT)
			{
				const uint32_t base = (g >> (lk - 2)) << lk, r = g & qm;
				const uint32_t ia = base + r, ib = ia + q, id = base + k - 1u - r, ic = id - q;
				uint64_t a = fr_ldk(keys, ia, n), b = fr_ldk(keys, ib, n), c = fr_ldk(keys, ic, n), d = fr_ldk(keys, id, n);
				fr_cx(a, d); fr_cx(b, c);
				fr_cx(a, b); fr_cx(c, d);
				fr_stk(keys, ia, n, a); fr_stk(keys, ib, n, b); fr_stk(keys, ic, n, c); fr_stk(keys, id, n, d);
			}
			__syncthreads();
		}
		// remaining stages j = k/8 ... 1, two at a time: (j, j/2) connect i, i + j/2, i + j, i + 3j/2
		uint32_t j = k >> 3, lj = lk - 3;
		for (; j >= 2; j >>= 2, lj -= 2)
		{
			const uint32_t h = j >> 1, lh = lj - 1, hm = h - 1u;
			for (uint32_t g = tid; g < groups; g += FR_NT)
			{
				const uint32_t i0 = ((g >> lh) << (lh + 2)) + (g & hm);
				const uint32_t i1 = i0 + h, i2 = i0 + j, i3 = i2 + h;
				uint64_t a = fr_ldk(keys, i0, n), b = fr_ldk(keys, i1, n), c = fr_ldk(keys, i2, n), d = fr_ldk(keys, i3, n);
				fr_cx(a, c); fr_cx(b, d);
				fr_cx(a, b); fr_cx(c, d);
				fr_stk(keys, i0, n, a); fr_stk(keys, i1, n, b); fr_stk(keys, i2, n, c); fr_stk(keys, i3, n, d);
			}
			__syncthreads();
		}
		if (j == 1)
		{
			for (uint32_t t = tid; t < (n_pad >> 1); t += FR_NT)
			{
				uint64_t a = fr_ldk(keys, 2 * t, n), b = fr_ldk(keys, 2 * t + 1, n);
				fr_cx(a, b);
				fr_stk(keys, 2 * t, n, a); fr_stk(keys, 2 * t + 1, n, b);
			}
			__syncthreads();
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// Register-resident form of the same network.  A wave holds a run of 64 K keys: key[r] of lane l is element l * K + r, so
// every stride below K -- the last log2 K passes of EVERY stage, and the first log2 K stages whole -- is a compare-exchange
// between registers of one lane (2.5 instructions per key), and the strides from K up to the run are exchanges between lanes
// on the VALU (5 instructions per key: DPP inside a row of 16 lanes, v_permlane16/32_swap across rows; tools/valu_ceiling
// --sort prices them).  Only the stages that join the runs of different waves go through LDS memory (stored there as
// [register][lane], conflict-free): 3 round trips for 4 waves (10 for 16) instead of one
// per two network stages (36 for 2048 keys) -- the LDS-resident form was LDS-bound with half its cycles lost to bank
// conflicts (profiles/r02_final_pmc_3.txt).  Elements at or beyond n are +inf in registers and never stored.
__device__ __forceinline__ uint64_t fr_shfl64(uint64_t v, int src_lane)
{
	const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)v);
	const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(uint32_t)(v >> 32));
	return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}
__device__ __forceinline__ void fr_take(uint64_t& mine, uint64_t other, bool keepmin)
{
	const bool take = (other < mine) == keepmin;
	mine = take ? other : mine;
}
// Exchanges between lanes on the VALU (the LDS crossbar of ds_bpermute is the bottleneck otherwise: ~800 of them per wave
// for 2048 keys): DPP quad permutes / row rotations / mirrors inside a row of 16, gfx950's row and half swaps beyond.
template <int CTRL>
__device__ __forceinline__ uint64_t fr_dpp64(uint64_t v)
{
	// (mov_dpp: no `old` operand to initialise -- every lane has a source under these controls)
	const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)v, CTRL, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
	return ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo;
}
// key <-> the key of lane ^ 16 (ROWS = 16) or lane ^ 32 (ROWS = 32): after the swap both lanes of a pair hold the same two
// values (a, b), so the result is min or max of the two whatever side the lane is on.
template <int ROWS>
__device__ __forceinline__ void fr_take_rows(uint64_t& key, bool keepmin)
{
	const uint32_t l = (uint32_t)key, h = (uint32_t)(key >> 32);
	uint64_t a, b;
	if constexpr (ROWS == 16)
	{
		const auto tl = __builtin_amdgcn_permlane16_swap(l, l, false, false);
		const auto th = __builtin_amdgcn_permlane16_swap(h, h, false, false);
		a = ((uint64_t)th[0] << 32) | tl[0]; b = ((uint64_t)th[1] << 32) | tl[1];
	}
	else
	{
		const auto tl = __builtin_amdgcn_permlane32_swap(l, l, false, false);
		const auto th = __builtin_amdgcn_permlane32_swap(h, h, false, false);
		a = ((uint64_t)th[0] << 32) | tl[0]; b = ((uint64_t)th[1] << 32) | tl[1];
	}
	key = ((b < a) == keepmin) ? b : a;
}

template <int K> struct FrLog2 { static constexpr int value = K == 1 ? 0 : K == 2 ? 1 : K == 4 ? 2 : K == 8 ? 3 : 4; };
// key <-> the key of lane ^ 2^M
template <int M>
__device__ __forceinline__ void fr_take_lane_xor(uint64_t& key, bool keepmin)
{
	if constexpr (M == 0) fr_take(key, fr_dpp64<0xB1>(key), keepmin);                       // quad_perm [1,0,3,2]
	else if constexpr (M == 1) fr_take(key, fr_dpp64<0x4E>(key), keepmin);                  // quad_perm [2,3,0,1]
	else if constexpr (M == 2) fr_take(key, fr_dpp64<0x141>(fr_dpp64<0x1B>(key)), keepmin); // (i ^ 3) ^ 7 = i ^ 4
	else if constexpr (M == 3) fr_take(key, fr_dpp64<0x128>(key), keepmin);                 // row_ror:8
	else if constexpr (M == 4) fr_take_rows<16>(key, keepmin);
	else fr_take_rows<32>(key, keepmin);
}
// the key of lane ^ (2^M - 1): mirrors inside a row by DPP (quad_perm [1,0,3,2], quad_perm [3,2,1,0], row_half_mirror,
// row_mirror), across rows by ds_bpermute
template <int M>
__device__ __forceinline__ uint64_t fr_lane_mirror(uint64_t key, int lane)
{
	if constexpr (M == 1) return fr_dpp64<0xB1>(key);
	else if constexpr (M == 2) return fr_dpp64<0x1B>(key);
	else if constexpr (M == 3) return fr_dpp64<0x141>(key);
	else if constexpr (M == 4) return fr_dpp64<0x140>(key);
	else return fr_shfl64(key, lane ^ ((1 << M) - 1));
}

// flip stage of block size k = 2^S inside the wave's run (S <= 6 + log2 K): element e <-> e ^ (k - 1).
// (Stage and stride numbers are template parameters so that every register index is a constant.)
template <int K, int S>
__device__ __forceinline__ void fr_wave_flip(uint64_t (&key)[K], int lane)
{
	constexpr int LK = FrLog2<K>::value;
	if constexpr (S <= LK)
	{
		constexpr int kr = 1 << S;                     // registers per block: mirror pairs inside the lane
#pragma unroll
		for (int r = 0; r < K; r++)
			if ((r & (kr >> 1)) == 0) fr_cx(key[r], key[r ^ (kr - 1)]);
	}
	else
	{
		// blocks of 2^M lanes: element (l, r) <-> (l ^ (2^M - 1), K - 1 - r); the half an element is in is a lane bit
		constexpr int M = S - LK;
		const bool keepmin = (lane & (1 << (M - 1))) == 0;
		if constexpr (K == 1) fr_take(key[0], fr_lane_mirror<M>(key[0], lane), keepmin);
		else
		{
#pragma unroll
			for (int r = 0; r < K / 2; r++)
			{
				const uint64_t lo = key[r], hi = key[K - 1 - r];
				fr_take(key[r], fr_lane_mirror<M>(hi, lane), keepmin);
				fr_take(key[K - 1 - r], fr_lane_mirror<M>(lo, lane), keepmin);
			}
		}
	}
}
// half-cleaners of strides 2^T, 2^(T-1), ..., 1 inside the wave's run: element e <-> e ^ j
template <int K, int T>
__device__ __forceinline__ void fr_wave_xor_down(uint64_t (&key)[K], int lane)
{
	constexpr int LK = FrLog2<K>::value;
	if constexpr (T >= 0)
	{
		if constexpr (T >= LK)
		{
			const bool keepmin = (lane & (1 << (T - LK))) == 0;
#pragma unroll
			for (int r = 0; r < K; r++) fr_take_lane_xor<T - LK>(key[r], keepmin);
		}
		else
		{
			constexpr int jr = 1 << T;
#pragma unroll
			for (int r = 0; r < K; r++)
				if ((r & jr) == 0) fr_cx(key[r], key[r | jr]);
		}
		fr_wave_xor_down<K, T - 1>(key, lane);
	}
}
// stages 2 .. min(n_pad, 2^S) of the network on the wave's own run
template <int K, int S>
__device__ __forceinline__ void fr_wave_stages(uint64_t (&key)[K], uint32_t n_pad, int lane)
{
	if constexpr (S >= 1)
	{
		fr_wave_stages<K, S - 1>(key, n_pad, lane);
		if ((1u << S) <= n_pad)
		{
			fr_wave_flip<K, S>(key, lane);
			fr_wave_xor_down<K, S - 2>(key, lane);
		}
	}
}

// One wave, n <= 64 K keys of a segment in global memory.
template <int K>
__device__ __forceinline__ void fr_sort_wave_segment(uint64_t* __restrict__ gk, uint32_t n, int lane)
{
	uint64_t key[K];
#pragma unroll
	for (int r = 0; r < K; r++) { const uint32_t i = (uint32_t)(lane * K + r); key[r] = i < n ? gk[i] : ~0ull; }
	uint32_t n_pad = 1;
	while (n_pad < n) n_pad <<= 1;
	fr_wave_stages<K, 6 + FrLog2<K>::value>(key, n_pad, lane);
#pragma unroll
	for (int r = 0; r < K; r++) { const uint32_t i = (uint32_t)(lane * K + r); if (i < n) gk[i] = key[r]; }
}

// The stages that join the runs of the NW waves of a workgroup (block size 2^SW runs), through LDS memory.
template <int K, int NW, int TW>
__device__ __forceinline__ void fr_wg_xor_down(uint64_t (&key)[K], uint64_t* sk, int w, int lane)
{
	if constexpr (TW >= 0)
	{
		constexpr int RUN = 64 * K, jw = 1 << TW;
		const uint32_t base = (uint32_t)(w * RUN);
#pragma unroll
		for (int r = 0; r < K; r++) sk[base + (uint32_t)(r * 64 + lane)] = key[r];
		__syncthreads();
		const uint32_t pbase = (uint32_t)((w ^ jw) * RUN);
		const bool keepmin = (w & jw) == 0;
#pragma unroll
		for (int r = 0; r < K; r++) fr_take(key[r], sk[pbase + (uint32_t)(r * 64 + lane)], keepmin);
		__syncthreads();
		fr_wg_xor_down<K, NW, TW - 1>(key, sk, w, lane);
	}
}
template <int K, int NW, int SW>
__device__ __forceinline__ void fr_wg_stages(uint64_t (&key)[K], uint64_t* sk, uint32_t n_pad, int w, int lane)
{
	if constexpr (SW >= 1)
	{
		fr_wg_stages<K, NW, SW - 1>(key, sk, n_pad, w, lane);
		constexpr int RUN = 64 * K, kw = 1 << SW;
		if (((uint32_t)RUN << SW) <= n_pad)            // uniform over the workgroup
		{
			// flip across the runs: element (w, r, l) <-> (w ^ (kw - 1), K - 1 - r, 63 - l)
			const uint32_t base = (uint32_t)(w * RUN);
#pragma unroll
			for (int r = 0; r < K; r++) sk[base + (uint32_t)(r * 64 + lane)] = key[r];
			__syncthreads();
			const uint32_t pbase = (uint32_t)((w ^ (kw - 1)) * RUN);
			const bool keepmin = (w & (kw >> 1)) == 0;
#pragma unroll
			for (int r = 0; r < K; r++) fr_take(key[r], sk[pbase + (uint32_t)((K - 1 - r) * 64 + (63 - lane))], keepmin);
			__syncthreads();
			fr_wg_xor_down<K, NW, SW - 2>(key, sk, w, lane);               // strides of whole runs
			fr_wave_xor_down<K, 5 + FrLog2<K>::value>(key, lane);          // ... and the rest of the stage inside the runs
		}
	}
}

// NW waves (a whole workgroup), n <= NW * 64 * K keys; sk = NW * 64 * K keys of LDS.
template <int K, int NW>
__device__ __forceinline__ void fr_sort_wg_segment(uint64_t* sk, uint64_t* __restrict__ gk, uint32_t n, int tid)
{
	static_assert(NW == 4 || NW == 16, "4 or 16 waves");
	constexpr int RUN = 64 * K;
	const int lane = tid & 63;
	const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t base = (uint32_t)(w * RUN);
	uint64_t key[K];
#pragma unroll
	for (int r = 0; r < K; r++) { const uint32_t i = base + (uint32_t)(lane * K + r); key[r] = i < n ? gk[i] : ~0ull; }
	uint32_t n_pad = 1;
	while (n_pad < n) n_pad <<= 1;
	fr_wave_stages<K, 6 + FrLog2<K>::value>(key, n_pad, lane);
	fr_wg_stages<K, NW, (NW == 4 ? 2 : 4)>(key, sk, n_pad, w, lane);
#pragma unroll
	for (int r = 0; r < K; r++) { const uint32_t i = base + (uint32_t)(lane * K + r); if (i < n) gk[i] = key[r]; }
}

// Segments of up to FR_SORT_SMALL_KEYS keys: one workgroup per (tile, view).  Up to 512 keys the first wave sorts alone
// (no barrier at all; the other three leave at once), beyond that the four waves hold 256 K keys, K = 4 or 8.
__global__ __launch_bounds__(FR_THREADS) void k_sort_tiles(FrParams p)
{
	if (p.status[1]) return;
	__shared__ uint64_t skeys[FR_SORT_SMALL_KEYS];
	const int tid = threadIdx.x;
	uint32_t tile; int v;
	fr_tile_of_block(p, tile, v);
	const size_t vt = (size_t)v * p.T + tile;
	const uint32_t n = p.tile_cnt[vt];
	if (n < 2 || n > p.small_max) return;
	uint64_t* gk = p.keys + p.tile_off[vt];
#ifdef FR_AB
	if (p.legacy_sort)
	{
		for (uint32_t i = tid; i < n; i += FR_THREADS) skeys[i] = gk[i];
		__syncthreads();
		fr_bitonic(skeys, n, tid, FR_THREADS);
		for (uint32_t i = tid; i < n; i += FR_THREADS) gk[i] = skeys[i];
		return;
	}
#endif
	if (n <= 512u)
	{
		if (tid >= 64) return;
		if (n <= 64u) fr_sort_wave_segment<1>(gk, n, tid);
		else if (n <= 128u) fr_sort_wave_segment<2>(gk, n, tid);
		else if (n <= 256u) fr_sort_wave_segment<4>(gk, n, tid);
		else fr_sort_wave_segment<8>(gk, n, tid);
		return;
	}
	// (one wave with 16 keys per lane for 513 .. 1024 keys: measured slower, 0.107 against 0.097 ms)
	if (n <= 1024u) fr_sort_wg_segment<4, 4>(skeys, gk, n, tid);
	else fr_sort_wg_segment<8, 4>(skeys, gk, n, tid);
}

// ---------------------------------------------------------------------------------------------------------
// k_sort_part: the long lists of fixed key segments (FR_SORT_SMALL_KEYS < n <= part_max) are PARTITIONED before they are sorted.
// A bitonic network over n keys runs log2(n_pad) (log2(n_pad) + 1) / 2 stages on every key -- 78 for 4096, 91 for 8192 -- and the
// two long-list tiers cost 3x / 10x the time per key of the short-list tier (16 keys per lane, 1024-thread workgroups).  Here a
// workgroup
//   1  sorts a SAMPLE of 512 of the list's keys (every n/512-th; one wave, in registers) and takes NP - 1 of them as pivots, NP = the
//      power of two that makes a part ~256-320 keys.  Pivots are full 64-bit keys (depth | slot), so runs of equal depth split too, and
//      a sample follows the depth clusters of the walls a tile looks at (equal-width buckets do not: profiles/r04_e_split_sort.txt);
//   2  classifies every key (binary search over the pivots: the part index is monotone in the key), counts the parts, scans;
//   3  scatters the keys part by part into the UPPER half of the tile's segment (a fixed segment holds tile_capacity keys, the list
//      uses n of them; the order inside a part is arbitrary);
//   4  lets its four waves take the parts one after the other, each sorted by ONE wave in registers (up to 512 keys: the cheapest
//      form of the network, 45 stages, no LDS, no barrier); a part that sampling left larger goes to the whole workgroup;
//   5  moves the tile's offset to where the sorted list now stands.
// Parts are ranges of the key order, so their concatenation is the sorted list: the same result as the bitonic tiers, which keep the
// lists this kernel does not take (packed lists: no room; lists beyond part_max).
#define FR_PART_SAMPLE 512
#define FR_PART_MAXP 64
__global__ __launch_bounds__(FR_THREADS) void k_sort_part(FrParams p, uint32_t part_max)
{
	__shared__ uint64_t skeys[FR_SORT_SMALL_KEYS];          // the workgroup-level sort of an oversized part
	__shared__ uint64_t s_piv[FR_PART_MAXP];
	__shared__ uint32_t s_cnt[FR_PART_MAXP], s_off[FR_PART_MAXP], s_cur[FR_PART_MAXP];
	__shared__ uint32_t s_next;
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t count = p.big_list[0];
	for (uint32_t b = blockIdx.x; b < count; b += gridDim.x)
	{
		const size_t vt = p.big_list[16 + b];
		const uint32_t n = p.tile_cnt[vt];
		if (n > part_max) continue;                            // (uniform) a bitonic tier's list
		uint64_t* gk = p.keys + p.tile_off[vt];
		const uint32_t n_al = (n + 63u) & ~63u;
		uint64_t* tmp = gk + n_al;
		// parts of ~256-320 keys: NP = 4 (n <= 1280), 8 (<= 2560), 16 (<= 5120), 32 (<= 10240), 64
		const uint32_t NP = n <= 1280u ? 4u : n <= 2560u ? 8u : n <= 5120u ? 16u : n <= 10240u ? 32u : 64u;
		__syncthreads();                                       // the previous list's LDS is free
		if (tid < FR_PART_MAXP) { s_cnt[tid] = 0u; s_cur[tid] = 0u; }
		if (tid == 0) s_next = 0u;
		// ---- 1: the sample, sorted by wave 0 (8 keys per lane)
		if (tid < 64)
		{
			uint64_t key[8];
#pragma unroll
			for (int r = 0; r < 8; r++) key[r] = gk[(uint32_t)(((uint64_t)(uint32_t)(lane * 8 + r) * n) >> 9)];
			fr_wave_stages<8, 9>(key, (uint32_t)FR_PART_SAMPLE, lane);
			// pivot q (1 .. NP - 1) = sample element q * 512 / NP = register 0 of lane q * 64 / NP
			const uint32_t step = 64u / NP;                      // 8, 4, 2 or 1 lanes
			if (((uint32_t)lane % step) == 0u && lane > 0) s_piv[(uint32_t)lane / step] = key[0];
		}
		__syncthreads();
		auto part_of = [&](uint64_t k) -> uint32_t {
			// number of pivots <= k (pivots ascending): binary search over NP - 1 entries, s_piv[1 .. NP - 1]
			uint32_t lo = 0u;                                    // invariant: pivots 1 .. lo are <= k
			for (uint32_t stepw = NP >> 1; stepw > 0u; stepw >>= 1)
			{
				const uint32_t mid = lo + stepw;
				if (s_piv[mid] <= k) lo = mid;
			}
			return lo;
		};
		// ---- 2: count (eight keys per thread and trip: their loads and their searches in flight together)
		constexpr int KB = 8;
		for (uint32_t i0 = 0; i0 < n; i0 += KB * FR_THREADS)
		{
			uint64_t k[KB];
#pragma unroll
			for (int r = 0; r < KB; r++) { const uint32_t i = i0 + r * FR_THREADS + tid; k[r] = i < n ? gk[i] : ~0ull; }
#pragma unroll
			for (int r = 0; r < KB; r++) { const uint32_t i = i0 + r * FR_THREADS + tid; if (i < n) atomicAdd(&s_cnt[part_of(k[r])], 1u); }
		}
		__syncthreads();
		if (tid == 0)
		{
			uint32_t run = 0;
			for (uint32_t q = 0; q < NP; q++) { s_off[q] = run; run += s_cnt[q]; }
		}
		__syncthreads();
		// ---- 3: scatter
		for (uint32_t i0 = 0; i0 < n; i0 += KB * FR_THREADS)
		{
			uint64_t k[KB];
#pragma unroll
			for (int r = 0; r < KB; r++) { const uint32_t i = i0 + r * FR_THREADS + tid; k[r] = i < n ? gk[i] : ~0ull; }
#pragma unroll
			for (int r = 0; r < KB; r++)
			{
				const uint32_t i = i0 + r * FR_THREADS + tid;
				if (i < n) { const uint32_t q = part_of(k[r]); tmp[s_off[q] + atomicAdd(&s_cur[q], 1u)] = k[r]; }
			}
		}
		__syncthreads();                                       // (the workgroup's own global stores, read back by its waves below)
		// ---- 4: the parts, one wave each
		for (;;)
		{
			uint32_t q = 0;
			if (lane == 0) q = atomicAdd(&s_next, 1u);
			q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
			if (q >= NP) break;
			const uint32_t c = s_cnt[q];
			uint64_t* pk = tmp + s_off[q];
			if (c < 2u || c > 512u) continue;
			if (c <= 64u) fr_sort_wave_segment<1>(pk, c, lane);
			else if (c <= 128u) fr_sort_wave_segment<2>(pk, c, lane);
			else if (c <= 256u) fr_sort_wave_segment<4>(pk, c, lane);
			else fr_sort_wave_segment<8>(pk, c, lane);
		}
		__syncthreads();
		// ... and what sampling left larger than a wave's 512 keys, by the whole workgroup (uniform: every thread reads the same counts)
		for (uint32_t q = 0; q < NP; q++)
		{
			const uint32_t c = s_cnt[q];
			if (c <= 512u) continue;
			uint64_t* pk = tmp + s_off[q];
			if (c <= 1024u) fr_sort_wg_segment<4, 4>(skeys, pk, c, tid);
			else if (c <= (uint32_t)FR_SORT_SMALL_KEYS) fr_sort_wg_segment<8, 4>(skeys, pk, c, tid);
			else fr_bitonic(pk, c, tid, FR_THREADS);
			__syncthreads();
		}
		if (tid == 0) p.tile_off[vt] += n_al;                  // ---- 5
	}
}

// Larger segments are listed by k_scan_tiles and sorted by two grid-stride kernels over that list:
//   k_sort_mid_tiles : FR_SORT_SMALL_KEYS < n <= FR_SORT_MID_KEYS, 256 threads x 16 keys, 32 KiB of LDS
//   k_sort_big_tiles : n > FR_SORT_MID_KEYS, 1024 threads x 8 or 16 keys, 128 KiB of LDS; beyond FR_SORT_BIG_KEYS the
//                      LDS-resident form of the network runs on global memory (__syncthreads orders the workgroup's own accesses).
__global__ __launch_bounds__(FR_THREADS) void k_sort_mid_tiles(FrParams p, uint32_t part_max)
{
	if (p.status[1]) return;
	__shared__ uint64_t skeys[FR_SORT_MID_KEYS];
	const int tid = threadIdx.x;
	const uint32_t count = p.big_list[0];
	for (uint32_t b = blockIdx.x; b < count; b += gridDim.x)
	{
		const size_t vt = p.big_list[16 + b];
		const uint32_t n = p.tile_cnt[vt];
		if (n > (uint32_t)FR_SORT_MID_KEYS || n <= part_max) continue;        // (n <= part_max: k_sort_part's list)
		uint64_t* gk = p.keys + p.tile_off[vt];
		__syncthreads();
#ifdef FR_AB
		if (p.legacy_sort)
		{
			for (uint32_t i = tid; i < n; i += FR_THREADS) skeys[i] = gk[i];
			__syncthreads();
			fr_bitonic(skeys, n, tid, FR_THREADS);
			for (uint32_t i = tid; i < n; i += FR_THREADS) gk[i] = skeys[i];
		}
		else
#endif
		fr_sort_wg_segment<16, 4>(skeys, gk, n, tid);
	}
}

__global__ __launch_bounds__(1024) void k_sort_big_tiles(FrParams p, uint32_t part_max)
{
	if (p.status[1]) return;
	__shared__ uint64_t skeys[FR_SORT_BIG_KEYS];
	const int tid = threadIdx.x;
	const uint32_t count = p.big_list[0];
	for (uint32_t b = blockIdx.x; b < count; b += gridDim.x)
	{
		const size_t vt = p.big_list[16 + b];
		const uint32_t n = p.tile_cnt[vt];
		if (n <= (uint32_t)FR_SORT_MID_KEYS || n <= part_max) continue;
		uint64_t* gk = p.keys + p.tile_off[vt];
		__syncthreads();
		if (n > (uint32_t)FR_SORT_BIG_KEYS) fr_bitonic(gk, n, tid, 1024);
#ifdef FR_AB
		else if (p.legacy_sort)
		{
			for (uint32_t i = tid; i < n; i += 1024) skeys[i] = gk[i];
			__syncthreads();
			fr_bitonic(skeys, n, tid, 1024);
			for (uint32_t i = tid; i < n; i += 1024) gk[i] = skeys[i];
		}
#endif
		else if (n <= 8192u) fr_sort_wg_segment<8, 16>(skeys, gk, n, tid);
		else fr_sort_wg_segment<16, 16>(skeys, gk, n, tid);
	}
}

// ---------------------------------------------------------------------------------------------------------
// alpha of one (pixel, splat) pair, rounded exactly like forward.cu:338-351 / the oracle.
// Returns false when the pair is skipped.
__device__ __forceinline__ bool fr_pair_alpha(float xyx, float xyy, float pfx, float pfy, float cx, float cy, float cz,
                                              float o, float thr, float& dx, float& dy, float& G, float& alpha)
{
	dx = xyx - pfx; dy = xyy - pfy;
	const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
	if (power > 0.0f) return false;
	if (power < thr) return false;          // conservative: alpha < 1/255 for certain
	G = fr_expf(power);
	alpha = fminf(0.99f, o * G);
	if (alpha < 1.0f / 255.0f) return false;
	return true;
}

// Forward compositing (forward.cu:261-393), wave-private: a wave owns the 16x4 strip of rows 4w..4w+3, streams the tile's sorted
// keys 64 at a time, keeps the splats whose conservative alpha footprint meets its strip (one ballot) and walks the set bits in
// order with v_readlane broadcasts -- no LDS staging, no barrier, and a wave only evaluates the ~1/3 of the tile's splats that
// can reach its pixels.  A skipped splat is one whose alpha is below 1/255 on every pixel of the strip, i.e. one the reference
// skips pixel by pixel, so colour, depth, final_T and n_contrib are unchanged (bit-identical to the oracle).
// NCH = 6: a second [P,3] feature array composited in the same pass (fr_forward_pair).
template <int NCH>
__global__ __launch_bounds__(FR_THREADS) void k_render_forward(FrParams p, const float* __restrict__ feat, int feat_view_stride,
                                                               float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                               float* __restrict__ out_color, float* __restrict__ out_depth,
                                                               const float* __restrict__ feat2, float* __restrict__ out_color2)
{
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int v = blockIdx.y;
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const size_t vP = (size_t)v * p.P;
	const uint32_t n = p.tile_cnt[vt];
	const uint64_t* gk = p.keys + p.tile_off[vt];
	const float4* splat = (const float4*)(p.splat + vP);
	const float* fv = feat + (size_t)v * feat_view_stride;

	unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);
	float T = 1.0f;
	uint32_t last_contributor = 0;
	float C[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) C[c] = 0.f;
	float D = 15.0f;
	const float strip_lo = (float)(ty * FR_BLOCK_Y + 4u * (uint32_t)wave), strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)(tx * FR_BLOCK_X), tile_x1 = tile_x0 + 15.0f;

	uint64_t kn = ((uint32_t)lane < n) ? gk[lane] : 0ull;
	for (uint32_t base = 0; base < n; base += 64)
	{
		const uint64_t key = kn;
		if (base + 64 + lane < n) kn = gk[base + 64 + lane];
		const bool valid = base + lane < n;
		const uint32_t id = (uint32_t)key;
		float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
		if (valid) { q0 = splat[2 * (size_t)id]; q1 = splat[2 * (size_t)id + 1]; }
		const uint32_t eb = __float_as_uint(q1.w);
		const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
		const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
		const bool ov = valid && hx >= 0.f && (q0.y + hy >= strip_lo) && (q0.y - hy <= strip_hi) && (q0.x + hx >= tile_x0) && (q0.x - hx <= tile_x1);
		unsigned long long todo = __ballot(ov);
		if (todo == 0ull) continue;
		const float thr_l = fr_power_threshold(q1.y);
		const float dep_l = fr_as_f32((uint32_t)(key >> 32));
		float fl[NCH];
#pragma unroll
		for (int c = 0; c < NCH; c++) fl[c] = 0.f;
		if (ov)
		{
			fl[0] = fv[3 * (size_t)id]; fl[1] = fv[3 * (size_t)id + 1]; fl[2] = fv[3 * (size_t)id + 2];
			if constexpr (NCH == 6) { fl[3] = feat2[3 * (size_t)id]; fl[4] = feat2[3 * (size_t)id + 1]; fl[5] = feat2[3 * (size_t)id + 2]; }
		}
		while (todo)
		{
			const int j = __builtin_ctzll(todo);
			todo &= todo - 1ull;
			const float x = fr_readlane_f(q0.x, j), y = fr_readlane_f(q0.y, j);
			const float cx = fr_readlane_f(q0.z, j), cy = fr_readlane_f(q0.w, j), cz = fr_readlane_f(q1.x, j);
			const float o = fr_readlane_f(q1.y, j), thr = fr_readlane_f(thr_l, j);
			const float dx = x - pfx, dy = y - pfy;
			const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
			// forward.cu:338-357; NaN falls through both tests as it does there
			const unsigned long long skip_m = __builtin_amdgcn_fcmpf(power, 0.0f, 2 /* ogt */) | __builtin_amdgcn_fcmpf(power, thr, 4 /* olt */);
			const unsigned long long pass_m = ~(skip_m | done_m);
			if (pass_m)
			{
				const float G = fr_expf(power);
				const float alpha = fminf(0.99f, o * G);
				const unsigned long long ok_m = pass_m & ~__builtin_amdgcn_fcmpf(alpha, 1.0f / 255.0f, 4 /* olt */);
				const float test_T = T * (1 - alpha);
				const unsigned long long kill_m = ok_m & __builtin_amdgcn_fcmpf(test_T, 0.0001f, 4 /* olt */);
				const unsigned long long contrib_m = ok_m & ~kill_m;
				done_m |= kill_m;
				if (contrib_m)
				{
					const bool contrib = __builtin_amdgcn_inverse_ballot_w64(contrib_m);
#pragma unroll
					for (int c = 0; c < NCH; c++)
					{
						const float fc = fr_readlane_f(fl[c], j);
						const float add = fc * alpha * T;
						C[c] = contrib ? C[c] + add : C[c];
					}
					const float dj = fr_readlane_f(dep_l, j);
					D = (contrib && T > 0.5f && test_T < 0.5f) ? dj : D;
					T = contrib ? test_T : T;
					last_contributor = contrib ? (base + (uint32_t)j + 1u) : last_contributor;
				}
			}
		}
		if (done_m == ~0ull) break;
	}
	if (inside)
	{
		const size_t HW = (size_t)p.H * p.W;
		const size_t pix = (size_t)p.W * pxy + pxx;
		final_T[v * HW + pix] = T;
		n_contrib[v * HW + pix] = last_contributor;
		if (out_color)
		{
			out_color[(v * 3 + 0) * HW + pix] = C[0] + T * p.bg[0];
			out_color[(v * 3 + 1) * HW + pix] = C[1] + T * p.bg[1];
			out_color[(v * 3 + 2) * HW + pix] = C[2] + T * p.bg[2];
		}
		if constexpr (NCH == 6)
		{
			out_color2[(v * 3 + 0) * HW + pix] = C[3] + T * p.bg[0];
			out_color2[(v * 3 + 1) * HW + pix] = C[4] + T * p.bg[1];
			out_color2[(v * 3 + 2) * HW + pix] = C[5] + T * p.bg[2];
		}
		if (out_depth) out_depth[v * HW + pix] = D;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Per-(pixel, splat) backward core shared by the Fisher scorer and the generic backward.
// Updates the back-to-front recurrences and returns u = (m2x, m2y, cx, cy, cw), the colour weights and
// the opacity gradient (backward.cu:978-1038).
struct FrPixState {
	float T, T_final;
	float accum0, accum1, accum2;
	float lastc0, lastc1, lastc2;
	float last_alpha;
};

template <bool FAST_RCP>
__device__ __forceinline__ void fr_pair_backward_t(FrPixState& s, float alpha, float G, float dx, float dy,
                                                 float cx, float cy, float cz, float o,
                                                 float c0, float c1, float c2, float g0, float g1, float g2,
                                                 float bg_dot, float ddelx_dx, float ddely_dy,
                                                 float& m2x, float& m2y, float& qx, float& qy, float& qw,
                                                 float& wcol, float& gop)
{
#pragma clang fp contract(fast)
	// 1/(1-alpha): IEEE division in the generic backward; one v_rcp_f32 (1 ulp) in the scorer, whose tolerance is 1e-4
	const float inv = FAST_RCP ? __builtin_amdgcn_rcpf(1.f - alpha) : 1.f / (1.f - alpha);
	s.T = FAST_RCP ? s.T * inv : s.T / (1.f - alpha);
	wcol = alpha * s.T;
	float dL_dalpha;
	s.accum0 = s.last_alpha * s.lastc0 + (1.f - s.last_alpha) * s.accum0; s.lastc0 = c0;
	s.accum1 = s.last_alpha * s.lastc1 + (1.f - s.last_alpha) * s.accum1; s.lastc1 = c1;
	s.accum2 = s.last_alpha * s.lastc2 + (1.f - s.last_alpha) * s.accum2; s.lastc2 = c2;
	dL_dalpha = (c0 - s.accum0) * g0 + (c1 - s.accum1) * g1 + (c2 - s.accum2) * g2;
	dL_dalpha *= s.T;
	s.last_alpha = alpha;
	if (bg_dot != 0.f) dL_dalpha += FAST_RCP ? (-s.T_final * inv) * bg_dot : (-s.T_final / (1.f - alpha)) * bg_dot;
	const float dL_dG = o * dL_dalpha;
	const float gdx = G * dx, gdy = G * dy;
	const float dG_ddelx = -gdx * cx - gdy * cy;
	const float dG_ddely = -gdy * cz - gdx * cy;
	m2x = dL_dG * dG_ddelx * ddelx_dx;
	m2y = dL_dG * dG_ddely * ddely_dy;
	qx = -0.5f * gdx * dx * dL_dG;
	qy = -0.5f * gdx * dy * dL_dG;
	qw = -0.5f * gdy * dy * dL_dG;
	gop = G * dL_dalpha;
}

__device__ __forceinline__ void fr_pair_backward(FrPixState& s, float alpha, float G, float dx, float dy,
                                                 float cx, float cy, float cz, float o,
                                                 float c0, float c1, float c2, float g0, float g1, float g2,
                                                 float bg_dot, float ddelx_dx, float ddely_dy,
                                                 float& m2x, float& m2y, float& qx, float& qy, float& qw,
                                                 float& wcol, float& gop)
{
	fr_pair_backward_t<false>(s, alpha, G, dx, dy, cx, cy, cz, o, c0, c1, c2, g0, g1, g2, bg_dot, ddelx_dx, ddely_dy,
	                          m2x, m2y, qx, qy, qw, wcol, gop);
}

// 64x64 bit-matrix transpose across the 64 lanes of a wave: lane i holds row i in, row i of the transpose out
// (out[i] bit j == in[j] bit i).  Six block-swap rounds, each one 64-bit exchange with the partner lane.
template <int SFT>
__device__ __forceinline__ unsigned long long fr_transpose_round(unsigned long long x, int lane)
{
	// m: bits whose index has (index & SFT) == 0
	constexpr unsigned long long m = SFT == 32 ? 0x00000000FFFFFFFFull : SFT == 16 ? 0x0000FFFF0000FFFFull :
	                                 SFT == 8 ? 0x00FF00FF00FF00FFull : SFT == 4 ? 0x0F0F0F0F0F0F0F0Full :
	                                 SFT == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
	// the partner lane's row on the VALU (DPP inside a row of 16 lanes, gfx950's row / half swaps beyond) -- six dependent
	// ds_bpermute round trips otherwise
	unsigned long long other;
	if constexpr (SFT == 1) other = fr_dpp64<0xB1>(x);                          // quad_perm [1,0,3,2]
	else if constexpr (SFT == 2) other = fr_dpp64<0x4E>(x);                     // quad_perm [2,3,0,1]
	else if constexpr (SFT == 4) other = fr_dpp64<0x141>(fr_dpp64<0x1B>(x));    // (i ^ 3) ^ 7 = i ^ 4
	else if constexpr (SFT == 8) other = fr_dpp64<0x128>(x);                    // row_ror:8
	else
	{
		const uint32_t l = (uint32_t)x, h = (uint32_t)(x >> 32);
		unsigned long long a, b;                                                // a: the lower member's row on both lanes of a pair, b: the upper's
		if constexpr (SFT == 16)
		{
			const auto tl = __builtin_amdgcn_permlane16_swap(l, l, false, false);
			const auto th = __builtin_amdgcn_permlane16_swap(h, h, false, false);
			a = ((unsigned long long)th[0] << 32) | tl[0]; b = ((unsigned long long)th[1] << 32) | tl[1];
		}
		else
		{
			const auto tl = __builtin_amdgcn_permlane32_swap(l, l, false, false);
			const auto th = __builtin_amdgcn_permlane32_swap(h, h, false, false);
			a = ((unsigned long long)th[0] << 32) | tl[0]; b = ((unsigned long long)th[1] << 32) | tl[1];
		}
		other = (lane & SFT) ? a : b;
	}
	return (lane & SFT) ? ((x & ~m) | ((other & ~m) >> SFT)) : ((x & m) | ((other & m) << SFT));
}
__device__ __forceinline__ unsigned long long fr_wave_transpose64(unsigned long long x, int lane)
{
	x = fr_transpose_round<32>(x, lane);
	x = fr_transpose_round<16>(x, lane);
	x = fr_transpose_round<8>(x, lane);
	x = fr_transpose_round<4>(x, lane);
	x = fr_transpose_round<2>(x, lane);
	x = fr_transpose_round<1>(x, lane);
	return x;
}

// ---------------------------------------------------------------------------------------------------------
// Fused Fisher scorer: one workgroup per (tile, view).
struct FrFisherArgs {
	float dL;                    // constant upstream gradient
	const float* dL_img;         // or per view an upstream-gradient image [V][3][H][W] (out_H modes only), stride in floats
	long long dL_stride;
	float* full_out[8];          // k_fisher_tile_v2<25>: dL_dmeans3D, dL_dopacity, dL_dscales, dL_drotations, dL_dcolors, dL_dmeans2D, dL_dcov3D, dL_dconic
	const float* H_inv; long long hinv_stride;
	float* out_H; long long outH_stride;
	float* tile_scores;          // [V][T] partial sums, reduced in fixed order by k_reduce_scores
	const uint8_t* only_flagged; // [V][T] or null: when set, k_fisher_tile handles only the flagged tiles
	int debug_mode;              // FR_DEBUG_MODE env (timing ablations only): 1 = k_fisher_tile_v2 stops after pass 1
	int key_shift;               // 0: keys = depth | index; 4: keys = depth | slot << 4 | strips (fixed key segments: k_preprocess_views_c<.., true>)
	// the scorer's records as k_fisher_tile_v3 / _v3h address them: record r of view v has {recA, recB} at recA[v * ab_view + r * ab_stride]
	// (+ 1) and its four recQ float4 at recQ[v * q_view + r * q_stride + k] -- two dense [V][P] arrays (strides 2 and 4), or the
	// compact 96-byte records (both strides 6, r = slot); slot_idx [V][slot_view] maps a slot back to the Gaussian index, or null
	const float4* recA; long long ab_view; int ab_stride;
	const float4* recQ; long long q_view; int q_stride;
	const uint32_t* slot_idx; long long slot_view;
};

template <int C, bool HAS_HINV, bool HAS_OUTH>
__global__ __launch_bounds__(FR_THREADS) void k_fisher_tile(FrParams p, FrFisherArgs f)
{
	constexpr int NA = (C == 11) ? 36 : 15;   // Jacobian coefficients per splat
	__shared__ fr_f2 s_xy[FR_BATCH];
	__shared__ fr_f4 s_co[FR_BATCH];
	__shared__ float s_thr[FR_BATCH];
	__shared__ float s_rgb[3][FR_BWD_BATCH];
	__shared__ float s_A[NA][FR_BWD_BATCH];
	__shared__ float s_hinv[HAS_HINV ? C : 1][FR_BWD_BATCH];
	__shared__ uint32_t s_id[FR_BWD_BATCH];
	__shared__ float s_red[4];
	__shared__ int s_redi[4];

	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint32_t tile; int v;
	fr_tile_of_block(p, tile, v);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	if (f.only_flagged && !f.only_flagged[vt]) return;
	const size_t vP = (size_t)v * p.P;
	const uint32_t n = p.tile_cnt[vt];
	const uint64_t* gk = p.keys + p.tile_off[vt];

	// ---- pass 1: transmittance only (forward.cu:331-380 without colour / depth) ----
	bool done = !inside;
	float T = 1.0f;
	uint32_t contributor = 0, last_contributor = 0;
	for (uint32_t base = 0; base < n; base += FR_BATCH)
	{
		if (__syncthreads_count(done) == FR_THREADS) break;
		const uint32_t k = base + tid;
		if (k < n)
		{
			const uint32_t id = (uint32_t)gk[k];
			const float4* sp = (const float4*)(p.splat + vP + id);
			const float4 q0 = sp[0], q1 = sp[1];
			fr_f2 xy_ = { q0.x, q0.y };
			s_xy[tid] = xy_;
			const fr_f4 co = { q0.z, q0.w, q1.x, q1.y };
			s_co[tid] = co;
			s_thr[tid] = fr_power_threshold(co.w);
		}
		__syncthreads();
		const int m = (int)min((uint32_t)FR_BATCH, n - base);
		for (int j = 0; !done && j < m; j++)
		{
			contributor++;
			const fr_f2 xy = s_xy[j];
			const fr_f4 co = s_co[j];
			float dx, dy, G, alpha;
			if (!fr_pair_alpha(xy.x, xy.y, pfx, pfy, co.x, co.y, co.z, co.w, s_thr[j], dx, dy, G, alpha))
				continue;
			const float test_T = T * (1 - alpha);
			if (test_T < 0.0001f) { done = true; continue; }
			T = test_T;
			last_contributor = contributor;
		}
	}
	const int last = inside ? (int)last_contributor : 0;
	// tile-wide maximum of last_contributor: nothing behind it is touched by the backward
	{
		int m = wave_max_i(last);
		__syncthreads();
		if (lane == 0) s_redi[wave] = m;
		__syncthreads();
	}
	int remaining = max(max(s_redi[0], s_redi[1]), max(s_redi[2], s_redi[3]));

	// ---- pass 2: back to front (backward.cu:939-1138), per-pixel leaf gradients squared on the fly ----
	float vm[16], pm[16], wm[12];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	const bool has_w2c = p.w2c != nullptr;
	if (has_w2c)
	{
#pragma unroll
		for (int k = 0; k < 12; k++) wm[k] = p.w2c[16 * (size_t)v + k];
	}
	FrPixState st;
	st.T_final = inside ? T : 0.f;
	st.T = st.T_final;
	st.accum0 = st.accum1 = st.accum2 = 0.f;
	st.lastc0 = st.lastc1 = st.lastc2 = 0.f;
	st.last_alpha = 0.f;
	float g0 = f.dL, g1 = f.dL, g2 = f.dL;
	if constexpr (HAS_OUTH)
	{
		if (f.dL_img)
		{
			const size_t HW = (size_t)p.H * p.W, pix = (size_t)p.W * pxy + pxx;
			const float* gi = f.dL_img + (size_t)v * f.dL_stride;
			g0 = inside ? gi[pix] : 0.f; g1 = inside ? gi[HW + pix] : 0.f; g2 = inside ? gi[2 * HW + pix] : 0.f;
		}
	}
	const float bg_dot = p.bg[0] * g0 + p.bg[1] * g1 + p.bg[2] * g2;
	const float ddelx_dx = (float)(0.5 * p.W), ddely_dy = (float)(0.5 * p.H);
	float score = 0.f;

	while (remaining > 0)
	{
		const int m = min(FR_BWD_BATCH, remaining);
		__syncthreads();
		if (tid < m)
		{
			const uint32_t id = (uint32_t)gk[remaining - 1 - tid];
			s_id[tid] = id;
			const float4* sp = (const float4*)(p.splat + vP + id);
			const float4 q0 = sp[0], q1 = sp[1];
			fr_f2 xy_ = { q0.x, q0.y };
			s_xy[tid] = xy_;
			const fr_f4 co = { q0.z, q0.w, q1.x, q1.y };
			s_co[tid] = co;
			s_thr[tid] = fr_power_threshold(co.w);
			s_rgb[0][tid] = p.colors[3 * (size_t)id]; s_rgb[1][tid] = p.colors[3 * (size_t)id + 1]; s_rgb[2][tid] = p.colors[3 * (size_t)id + 2];
			fr_f3 pw = { p.means3D[3 * (size_t)id], p.means3D[3 * (size_t)id + 1], p.means3D[3 * (size_t)id + 2] };
			fr_f3 po = has_w2c ? fr_world_to_cam(pw, wm) : pw;
			float c3[6];
#pragma unroll
			for (int k = 0; k < 6; k++) c3[k] = p.cov3D[6 * (size_t)id + k];
			float A[3][5];
			float B[6][3];
			fr_mean_jacobian(po, c3, vm, pm, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, A, (C == 11) ? B : nullptr);
#pragma unroll
			for (int r = 0; r < 3; r++)
#pragma unroll
				for (int c = 0; c < 5; c++) s_A[r * 5 + c][tid] = A[r][c];
			if constexpr (C == 11)
			{
				fr_f3 sc = { p.scales[3 * (size_t)id], p.scales[3 * (size_t)id + 1], p.scales[3 * (size_t)id + 2] };
				fr_f4 q = { p.rots[4 * (size_t)id], p.rots[4 * (size_t)id + 1], p.rots[4 * (size_t)id + 2], p.rots[4 * (size_t)id + 3] };
				float Cm[7][3];
				fr_scale_rot_jacobian(sc, p.mod, q, B, Cm);
#pragma unroll
				for (int r = 0; r < 7; r++)
#pragma unroll
					for (int c = 0; c < 3; c++) s_A[15 + r * 3 + c][tid] = Cm[r][c];
			}
			if constexpr (HAS_HINV)
			{
				const float* hp = f.H_inv + (size_t)v * f.hinv_stride + (size_t)id * C;
#pragma unroll
				for (int c = 0; c < C; c++) s_hinv[c][tid] = hp[c];
			}
		}
		__syncthreads();
		for (int j = 0; j < m; j++)
		{
			const int k = remaining - 1 - j;     // 0-based position in the tile's list == `contributor`
			bool act = inside && (k < last);
			if (!__any(act)) continue;
			const fr_f2 xy = s_xy[j];
			const fr_f4 co = s_co[j];
			float dx, dy, G = 0.f, alpha = 0.f;
			act = act && fr_pair_alpha(xy.x, xy.y, pfx, pfy, co.x, co.y, co.z, co.w, s_thr[j], dx, dy, G, alpha);
			if (!__any(act)) continue;
			float leaf2[C];
#pragma unroll
			for (int c = 0; c < C; c++) leaf2[c] = 0.f;
			if (act)
			{
				float m2x, m2y, qx, qy, qw, wcol, gop;
				fr_pair_backward(st, alpha, G, dx, dy, co.x, co.y, co.z, co.w,
				                 s_rgb[0][j], s_rgb[1][j], s_rgb[2][j], g0, g1, g2, bg_dot, ddelx_dx, ddely_dy,
				                 m2x, m2y, qx, qy, qw, wcol, gop);
#pragma unroll
				for (int r = 0; r < 3; r++)
				{
					const float l = s_A[r * 5 + 0][j] * m2x + s_A[r * 5 + 1][j] * m2y + s_A[r * 5 + 2][j] * qx
					              + s_A[r * 5 + 3][j] * qy + s_A[r * 5 + 4][j] * qw;
					leaf2[r] = l * l;
				}
				leaf2[3] = gop * gop;
				if constexpr (C == 11)
				{
#pragma unroll
					for (int r = 0; r < 7; r++)
					{
						const float l = s_A[15 + r * 3 + 0][j] * qx + s_A[15 + r * 3 + 1][j] * qy + s_A[15 + r * 3 + 2][j] * qw;
						leaf2[4 + r] = l * l;
					}
				}
				if constexpr (HAS_HINV)
				{
#pragma unroll
					for (int c = 0; c < C; c++) score += leaf2[c] * s_hinv[c][j];
				}
			}
			if constexpr (HAS_OUTH)
			{
				// sum over the wave's pixels, then one atomic per column (the reference: one per pixel per column)
				float mine = 0.f;
#pragma unroll
				for (int c = 0; c < C; c++)
				{
					const float s = wave_sum(leaf2[c]);
					if (lane == c) mine = s;
				}
				if (lane < C && mine != 0.f)
					atomicAdd(f.out_H + (size_t)v * f.outH_stride + (size_t)s_id[j] * C + lane, mine);
			}
		}
		remaining -= m;
	}
	if constexpr (HAS_HINV)
	{
		const float ws = wave_sum(score);
		__syncthreads();
		if (lane == 0) s_red[wave] = ws;
		__syncthreads();
		if (tid == 0) f.tile_scores[vt] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
	}
}


// ---------------------------------------------------------------------------------------------------------
// Scoring path, second generation.
//
// k_build_records   one pass over the sorted tile instances: gathers everything a (pixel, splat) pair needs and the
//                   per-instance Jacobian rows into two streams laid out in list order:
//                     recA[s] (32 B) = { xy, conic, opacity, power threshold, id }            -- read by both passes
//                     recB[s]        = { rgb[3], A[3][5], (Cm[7][3]), H_inv[C], pad }         -- read by pass 2
//                   so the tile kernel streams its splats instead of gathering them, and the register-hungry
//                   Jacobian chain lives in a kernel of its own.
// k_fisher_tile_v2  pass 1 (transmittance) additionally records, per wave, WHICH splats touched its 16x4 strip.
//                   pass 2 is wave-private: a wave walks only its own list, 64 entries at a time held one per lane in
//                   registers; every lane first builds a 64-bit mask of the entries that contribute to ITS pixel
//                   (v_readlane broadcast + the same pair test), then walks its own set bits back to front, fetching
//                   the record it needs from the owning lane with ds_bpermute.  All 64 lanes do useful work on every
//                   step instead of the whole wave executing one splat for the few pixels it covers.
// Tiles whose lists do not fit the LDS index (u16 positions, FR_WCAP per wave) are flagged and redone by k_fisher_tile.
// ---------------------------------------------------------------------------------------------------------
#define FR_WCAP 3840                 // per-wave list capacity (u16 positions): 30 KiB; LDS stays under the 40 KiB that 4 workgroups/CU allow


// View-independent inputs of one Gaussian packed into one record: { mean[3], cov3D[6], rgb[3], (scale[3], rot[4]),
// H_inv[C] } = 16 floats (64 B, one cache line) at C = 4, 32 floats at C = 11.  k_build_records then needs two gathers
// per tile instance (this record and the 32-byte FrSplat) instead of six.  With per-view H_inv the weights are
// gathered separately.

template <int C>
__global__ __launch_bounds__(FR_THREADS) void k_pack_static(FrParams p, const float* __restrict__ H_inv, float* __restrict__ packed,
                                                            float4* __restrict__ mt, float4* __restrict__ grp)
{
	constexpr int PS = FrPackSize<C>::value;
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	const bool live = i < p.P;
	// position i of the processing order holds the caller's Gaussian `src` (fr_fisher_cfg.order; the identity without it)
	const size_t src = live ? (p.order ? (size_t)p.order[i] : (size_t)i) : 0;
	float b[PS];
#pragma unroll
	for (int k = 0; k < PS; k++) b[k] = 0.f;
	if (live)
	{
#pragma unroll
		for (int k = 0; k < 3; k++) b[k] = p.means3D[3 * src + k];
		if (p.cov3D)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) b[3 + k] = p.cov3D[6 * src + k];
		}
		else
		{
			// (fr_bin_pipeline leaves cov3D null when this kernel is its only reader: k_cov3d's arithmetic, forward.cu:118-152)
			const fr_f3 sc = { p.scales[3 * src], p.scales[3 * src + 1], p.scales[3 * src + 2] };
			const fr_f4 q = { p.rots[4 * src], p.rots[4 * src + 1], p.rots[4 * src + 2], p.rots[4 * src + 3] };
			fr_cov3d(sc, p.mod, q, &b[3]);
		}
#pragma unroll
		for (int k = 0; k < 3; k++) b[9 + k] = p.colors[3 * src + k];
		int o = 12;
		if constexpr (C >= 11)
		{
#pragma unroll
			for (int k = 0; k < 3; k++) b[12 + k] = p.scales[3 * src + k];
#pragma unroll
			for (int k = 0; k < 4; k++) b[15 + k] = p.rots[4 * src + k];
			o = 19;
		}
		if constexpr (C <= 11)
		{
			if (H_inv)
			{
#pragma unroll
				for (int c = 0; c < C; c++) b[o + c] = H_inv[src * C + c];
			}
		}
		float4* dst = (float4*)(packed + (size_t)i * PS);
#pragma unroll
		for (int k = 0; k < PS / 4; k++) dst[k] = make_float4(b[4 * k], b[4 * k + 1], b[4 * k + 2], b[4 * k + 3]);
	}
	// xx + yy + zz of the symmetric 3 x 3: an upper bound of its largest eigenvalue (the tighter (mod * largest scale)^2 only
	// holds for unit quaternions, which forward.cu:120-151 does not require)
	const float tr = b[3] + b[6] + b[8];
	if (mt && live) mt[i] = make_float4(b[0], b[1], b[2], tr);
	if (grp)
	{
		// bounds of this workgroup's 256 Gaussians = one 256-Gaussian round of the projection kernel: the box of the means and the largest
		// trace (a NaN anywhere makes the box NaN, which no test of the projection kernel rejects)
		__shared__ float s_b[4][7];
		float lo[3], hi[3], tm = live ? tr : 0.f;
#pragma unroll
		for (int k = 0; k < 3; k++) { lo[k] = live ? b[k] : 3.0e38f; hi[k] = live ? b[k] : -3.0e38f; }
		const bool bad = live && !(b[0] == b[0] && b[1] == b[1] && b[2] == b[2] && tr == tr);
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1)
		{
#pragma unroll
			for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, 64)); }
			tm = fmaxf(tm, __shfl_xor(tm, d, 64));
		}
		const bool any_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
		const int w = threadIdx.x >> 6;
		if ((threadIdx.x & 63) == 0)
		{
			const float nan = __uint_as_float(0x7fc00000u);
			for (int k = 0; k < 3; k++) { s_b[w][k] = any_bad ? nan : lo[k]; s_b[w][3 + k] = hi[k]; }
			s_b[w][6] = tm;
		}
		__syncthreads();
		if (threadIdx.x == 0)
		{
			float L[3], Hh[3], t = s_b[0][6];
			bool nn = false;
			for (int k = 0; k < 3; k++) { L[k] = s_b[0][k]; Hh[k] = s_b[0][3 + k]; }
			for (int ww = 0; ww < 4; ww++)
			{
				for (int k = 0; k < 3; k++) { nn = nn || !(s_b[ww][k] == s_b[ww][k]); L[k] = fminf(L[k], s_b[ww][k]); Hh[k] = fmaxf(Hh[k], s_b[ww][3 + k]); }
				t = fmaxf(t, s_b[ww][6]);
			}
			const float nan = __uint_as_float(0x7fc00000u);
			grp[2 * (size_t)blockIdx.x] = make_float4(nn ? nan : L[0], L[1], L[2], t);
			grp[2 * (size_t)blockIdx.x + 1] = make_float4(Hh[0], Hh[1], Hh[2], 0.f);
		}
	}
}

#define FR_E255 (-7.994353436858858f)     // log2(1/255)
// exponent of the scorer's alpha (see k_fisher_tile_v2): e = power*log2(e) + log2(opacity)
__device__ __forceinline__ float fr_scorer_exponent(float hcx, float ncy, float hcz, float dx, float dy, float lo, float& power)
{
	const float t = __builtin_fmaf(hcx, dx, ncy * dy);
	const float v = hcz * dy;
	power = __builtin_fmaf(dx, t, v * dy);
	return __builtin_fmaf(power, 1.44269504088896341f, lo);
}

// HAS_HINV: weighted score per view (pose_eval).  HAS_OUTH: cur_H materialised / accumulated (compute_Hessian,
// compute_H_train): the pixel-lanes of a wave add their squared leaves into a 64-entry LDS accumulator of the chunk
// (ds_add_f32), and the lane that owns an entry then issues ONE global atomic per column for the whole wave -- the
// reference issues one per pixel per column.
template <int C, bool HAS_HINV, bool HAS_OUTH>
__global__ __launch_bounds__(FR_THREADS) __attribute__((amdgpu_waves_per_eu(C == 4 ? (HAS_OUTH ? 4 : 5) : (HAS_OUTH ? 2 : 4)))) void k_fisher_tile_v2(FrParams p, FrFisherArgs f, const float* __restrict__ packed,
                                                               uint8_t* __restrict__ fallback)
{
	__shared__ double s_acc[HAS_OUTH ? 4 : 1][HAS_OUTH ? C : 1][64];   // double: ds_add_f64 is ~20x faster than ds_add_f32 on MI355X (tools/lds_atomic_rate.hip)
	constexpr int PS = FrPackSize<C>::value;
	// per-entry registers of pass 2: rgb[3], A'[15], (Cm'[21]), k3, [H_inv columns]
	// QFORM (score only): the weighted sum of squared leaves is a quadratic form in u',
	//   sum_c H_inv[c] (M_c . u')^2 = u'^T Q u',  Q = sum_c H_inv[c] M_c^T M_c  (5 x 5 symmetric, 15 numbers),
	// so the walk fetches rgb[3], Q[15], k3 instead of 51 values per entry, whatever the number of columns.
	// C = 25 (FULL): every leaf of the rasteriser's power-2 backward (backward.cu:1095-1137) for one view -- camera-frame mean 3,
	// opacity 1, scale 3, rotation 4, colour 3, mean2D 2, cov3D 6, conic 3 -- summed per Gaussian into the gradient tensors
	// of fr_backward (f.full_out); the per-entry record then also carries B' = -1/2 d(dL_dcov3D)/d(dL_dconic) (18 numbers).
	constexpr bool SR = C >= 11;
	constexpr bool FULL = C == 25;
	static_assert(!FULL || (HAS_OUTH && !HAS_HINV), "the all-leaves mode has no weights");
	constexpr bool QFORM = HAS_HINV && !HAS_OUTH;
	constexpr int KO = QFORM ? 18 : (FULL ? 57 : (SR ? 39 : 18));   // offset of k3 = 1/opacity^2 (times H_inv[3] when only the score is wanted)
	constexpr int HO = KO + 1;                    // offset of the H_inv columns
	constexpr bool FOLD3 = HAS_HINV && !HAS_OUTH; // H_inv[3] folded into k3
	constexpr int NB = QFORM ? 19 : HO + (HAS_HINV ? C : 0);
	// per-wave contributor list; the all-leaves mode trades 256 entries of it for a second resident workgroup (its 25 double
	// accumulators per candidate take 50 KiB: 80 KiB in all)
	constexpr int WCAP = FULL ? FR_WCAP - 256 : FR_WCAP;
	__shared__ uint16_t s_wl[4][WCAP];
	__shared__ float s_red[4];
	__shared__ int s_ovf;

	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction; tell the compiler
	uint32_t tile; int v;
	fr_tile_of_block(p, tile, v);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const size_t vP = (size_t)v * p.P;
	const uint32_t n = p.tile_cnt[vt];
	const uint64_t* gk = p.keys + p.tile_off[vt];
	const float4* splat = (const float4*)(p.splat + vP);
	if (n > 65535u)
	{
		if (tid == 0) fallback[vt] = 1;
		return;
	}
	if (tid == 0) s_ovf = 0;
	__syncthreads();

	// ---- pass 1: wave-private ----
	// A wave streams the tile's sorted keys 64 at a time (one per lane), gathers each splat's 32-byte record (next chunk
	// already in flight), keeps the ones whose conservative alpha footprint meets its 16x4 strip (one ballot), and walks
	// the set bits in order, broadcasting a record from its owning lane with v_readlane.  No LDS staging, no workgroup
	// barrier: a wave whose 64 pixels are all finished simply leaves.
	// The scorer's alpha: a = 2^e, e = power*log2(e) + log2(opacity), power = dx*(hcx*dx + ncy*dy) + (hcz*dy)*dy with
	// (hcx, ncy, hcz) = (-conic.x/2, -conic.y, -conic.z/2): the same value as forward.cu:338-346 up to rounding, evaluated by
	// the identical instruction sequence in both passes (fr_scorer_exponent), so the two passes agree on every pair.
	// The wave votes are kept as 64-bit scalar masks (v_cmp writes them, s_and/s_or combine them).
	unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);
	float T = 1.0f;
	int last = 0;
	int wcnt = 0;                                  // wave-uniform
#ifdef FR_LOOPSTATS
	int dbg_p1 = 0, dbg_chunks = 0, dbg_steps = 0, dbg_hits = 0, dbg_p1any = 0, dbg_p1con = 0;   // -DFR_LOOPSTATS + FR_DEBUG_MODE >= 2: loop-trip counters
#define FR_STAT(x) x
#else
#define FR_STAT(x)
#endif
	const float strip_lo = (float)(ty * FR_BLOCK_Y + 4u * (uint32_t)wave), strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)(tx * FR_BLOCK_X), tile_x1 = tile_x0 + 15.0f;
	float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;
	uint32_t idn = 0;
	if ((uint32_t)lane < n)
	{
		idn = (uint32_t)gk[lane];
		n0 = splat[2 * (size_t)idn]; n1 = splat[2 * (size_t)idn + 1];
	}
	uint32_t idnn = (64u + lane < n) ? (uint32_t)gk[64 + lane] : 0u;
	for (uint32_t base = 0; base < n; base += 64)
	{
		const float4 q0 = n0, q1 = n1;
		if (base + 64 + lane < n) { n0 = splat[2 * (size_t)idnn]; n1 = splat[2 * (size_t)idnn + 1]; }
		if (base + 128 + lane < n) idnn = (uint32_t)gk[base + 128 + lane];
		const uint32_t eb = __float_as_uint(q1.w);
		const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
		const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
		const bool ov = (base + lane < n) && hx >= 0.f && (q0.y + hy >= strip_lo) && (q0.y - hy <= strip_hi)
		                && (q0.x + hx >= tile_x0) && (q0.x - hx <= tile_x1);
		const float hcx_l = -0.5f * q0.z, ncy_l = -q0.w, hcz_l = -0.5f * q1.x;
		const float lo_l = __builtin_amdgcn_logf(q1.y);          // v_log_f32: log2(opacity); opacity <= 1/255 never gets here (hx < 0)
		unsigned long long todo = __ballot(ov);
		while (todo)
		{
			const int j = __builtin_ctzll(todo);
			todo &= todo - 1ull;
			FR_STAT(dbg_p1++;)
			const float x = fr_readlane_f(q0.x, j), y = fr_readlane_f(q0.y, j);
			const float hcx = fr_readlane_f(hcx_l, j), ncy = fr_readlane_f(ncy_l, j), hcz = fr_readlane_f(hcz_l, j);
			const float lo = fr_readlane_f(lo_l, j);
			const float dx = x - pfx, dy = y - pfy;
			float power;
			const float e = fr_scorer_exponent(hcx, ncy, hcz, dx, dy, lo, power);
			// forward.cu:347-357 (power > 0 -> skip, alpha < 1/255 -> skip); NaN falls through as it does there
			const unsigned long long skip_m = __builtin_amdgcn_fcmpf(power, 0.0f, 2 /* ogt */) | __builtin_amdgcn_fcmpf(e, FR_E255, 4 /* olt */);
			const unsigned long long pass_m = ~(skip_m | done_m);
			if (pass_m)
			{
				FR_STAT(dbg_p1any++;)
				const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(e));
				const float test_T = T * (1 - alpha);
				const unsigned long long kill_m = pass_m & __builtin_amdgcn_fcmpf(test_T, 0.0001f, 4 /* olt */);
				const unsigned long long contrib_m = pass_m & ~kill_m;
				done_m |= kill_m;
				if (contrib_m)
				{
					FR_STAT(dbg_p1con++;)
					const bool contrib = __builtin_amdgcn_inverse_ballot_w64(contrib_m);
					T = contrib ? test_T : T;
					last = contrib ? (wcnt + 1) : last;      // 1 + index in THIS wave's list of the pixel's last contributor
					// every lane stores the same value to the same address: no exec-mask juggling for a one-lane write
					if (wcnt < WCAP) s_wl[wave][wcnt] = (uint16_t)(base + j);
					wcnt = __builtin_amdgcn_readfirstlane(wcnt + 1);
				}
			}
		}
		if (done_m == ~0ull) break;
	}
	if (lane == 0 && wcnt > WCAP) s_ovf = 1;
	__syncthreads();
	if (s_ovf)
	{
		if (tid == 0) fallback[vt] = 1;
		return;
	}
	if (tid == 0) fallback[vt] = 0;
	FR_AB_ONLY(if (f.debug_mode == 1) { if (tid == 0) f.tile_scores[vt] = (float)wcnt; return; })

	// ---- pass 2: wave-private, back to front ----
	// 64 list entries at a time, one per lane: the lane gathers its splat and the packed static record and pushes unit
	// vectors through the Jacobian chain (backward.cu:335-475) -- once per entry, all 64 lanes busy.  Every lane then
	// builds a 64-bit mask of the entries that can contribute to ITS pixel and walks its own set bits back to front,
	// fetching the record it needs from the owning lane with ds_bpermute.
	float vm[16], pm[16], wm[12];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	const bool has_w2c = p.w2c != nullptr;
	if (has_w2c)
	{
#pragma unroll
		for (int k = 0; k < 12; k++) wm[k] = p.w2c[16 * (size_t)v + k];
	}
	const bool per_view_hinv = f.hinv_stride != 0;
	FrPixState st;
	st.T_final = inside ? T : 0.f;
	st.T = st.T_final;
	st.accum0 = st.accum1 = st.accum2 = 0.f;
	st.lastc0 = st.lastc1 = st.lastc2 = 0.f;
	st.last_alpha = 0.f;
	float g0 = f.dL, g1 = f.dL, g2 = f.dL;
	if constexpr (HAS_OUTH)
	{
		if (f.dL_img)
		{
			const size_t HW = (size_t)p.H * p.W, pix = (size_t)p.W * pxy + pxx;
			const float* gi = f.dL_img + (size_t)v * f.dL_stride;
			g0 = inside ? gi[pix] : 0.f; g1 = inside ? gi[HW + pix] : 0.f; g2 = inside ? gi[2 * HW + pix] : 0.f;
		}
	}
	const float bg_dot = p.bg[0] * g0 + p.bg[1] * g1 + p.bg[2] * g2;
	const float ddelx_dx = (float)(0.5 * p.W), ddely_dy = (float)(0.5 * p.H);
	float score = 0.f;
	const uint16_t* wl = s_wl[wave];

	for (int hi = wcnt; hi > 0; hi -= 64)
	{
		const int m = min(64, hi);
		FR_STAT(dbg_chunks++;)
		// lane l owns list entry hi-1-l (descending position => bit order == back-to-front order)
		int kk = -1;                  // unused lanes sort behind every real position (positions are descending in the lane index)
		float ax = 0.f, ay = 0.f, acx = 0.f, acy = 0.f, acz = 0.f, alo = 0.f, athr = INFINITY, ahx = -1.f, ahy = -1.f;
		float b[NB];
#pragma unroll
		for (int q = 0; q < NB; q++) b[q] = 0.f;
		uint32_t my_id = 0;
		if constexpr (HAS_OUTH)
		{
#pragma unroll
			for (int c = 0; c < C; c++) s_acc[wave][c][lane] = 0.0;
		}
		if (lane < m)
		{
			kk = (int)wl[hi - 1 - lane];
			const uint32_t id = (uint32_t)gk[kk];
			my_id = id;
			const float4 a0 = splat[2 * (size_t)id], a1 = splat[2 * (size_t)id + 1];
			ax = a0.x; ay = a0.y; acx = a0.z; acy = a0.w; acz = a1.x; athr = fr_power_threshold(a1.y);
			alo = __builtin_amdgcn_logf(a1.y);
			const float inv_o = __builtin_amdgcn_rcpf(a1.y);
			{
				const uint32_t eb = __float_as_uint(a1.w);
				ahx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
				ahy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
			}
			float gsv[PS];
			const float4* pk = (const float4*)(packed + (size_t)id * PS);
#pragma unroll
			for (int q = 0; q < PS / 4; q++) { const float4 t4 = pk[q]; gsv[4 * q] = t4.x; gsv[4 * q + 1] = t4.y; gsv[4 * q + 2] = t4.z; gsv[4 * q + 3] = t4.w; }
			b[0] = gsv[9]; b[1] = gsv[10]; b[2] = gsv[11];
			fr_f3 pw = { gsv[0], gsv[1], gsv[2] };
			fr_f3 po = has_w2c ? fr_world_to_cam(pw, wm) : pw;
			float A[3][5];
			float B[6][3];
			// IEEE division in the modes that materialise cur_H (1e-4 per element, near-plane splats included); reciprocals when only a sum is wanted
			fr_mean_jacobian<!HAS_OUTH>(po, &gsv[3], vm, pm, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, A, SR ? B : nullptr);
			// columns pre-scaled so that the walk can feed them u' = (-(cx dx + cy dy), -(cz dy + cy dx), dx^2, dx dy, dy^2):
			// (dL_dmean2D.xy, dL_dconic.xyw) = w * (ddelx_dx u'0, ddely_dy u'1, -u'2/2, -u'3/2, -u'4/2), w = opacity G dL_dalpha
			float Ap[3][5];
#pragma unroll
			for (int r = 0; r < 3; r++)
			{
				Ap[r][0] = A[r][0] * ddelx_dx; Ap[r][1] = A[r][1] * ddely_dy;
#pragma unroll
				for (int c = 2; c < 5; c++) Ap[r][c] = -0.5f * A[r][c];
			}
			float Cp[SR ? 7 : 1][3];
			int go = 12;
			if constexpr (SR)
			{
				fr_f3 sc = { gsv[12], gsv[13], gsv[14] };
				fr_f4 qr = { gsv[15], gsv[16], gsv[17], gsv[18] };
				float Cm[7][3];
				fr_scale_rot_jacobian(sc, p.mod, qr, B, Cm);
#pragma unroll
				for (int r = 0; r < 7; r++)
#pragma unroll
					for (int c = 0; c < 3; c++) Cp[r][c] = -0.5f * Cm[r][c];
				go = 19;
			}
			float hv[HAS_HINV ? C : 1];
			if constexpr (HAS_HINV)
			{
				if (per_view_hinv)
				{
					const float* hp = f.H_inv + (size_t)v * f.hinv_stride + (size_t)id * C;
#pragma unroll
					for (int c = 0; c < C; c++) hv[c] = hp[c];
				}
				else
				{
#pragma unroll
					for (int c = 0; c < C; c++) hv[c] = gsv[go + c];
				}
			}
			(void)go;
			if constexpr (QFORM)
			{
				// upper triangle of Q, row-major, off-diagonal entries doubled: u'^T Q u' = sum_{i <= j} Q'[ij] u_i u_j
				int q = 3;
#pragma unroll
				for (int i = 0; i < 5; i++)
#pragma unroll
					for (int j = i; j < 5; j++)
					{
						float acc = hv[0] * Ap[0][i] * Ap[0][j] + hv[1] * Ap[1][i] * Ap[1][j] + hv[2] * Ap[2][i] * Ap[2][j];
						if constexpr (SR)
						{
							if (i >= 2)
							{
#pragma unroll
								for (int r = 0; r < 7; r++) acc += hv[4 + r] * Cp[r][i - 2] * Cp[r][j - 2];
							}
						}
						b[q++] = (i == j) ? acc : 2.0f * acc;
					}
				b[KO] = inv_o * inv_o * hv[3];
			}
			else
			{
#pragma unroll
				for (int r = 0; r < 3; r++)
#pragma unroll
					for (int c = 0; c < 5; c++) b[3 + r * 5 + c] = Ap[r][c];
				if constexpr (SR)
				{
#pragma unroll
					for (int r = 0; r < 7; r++)
#pragma unroll
						for (int c = 0; c < 3; c++) b[18 + r * 3 + c] = Cp[r][c];
				}
				if constexpr (FULL)
				{
#pragma unroll
					for (int r = 0; r < 6; r++)
#pragma unroll
						for (int c = 0; c < 3; c++) b[39 + r * 3 + c] = -0.5f * B[r][c];
				}
				if constexpr (HAS_HINV)
				{
#pragma unroll
					for (int c = 0; c < C; c++) b[HO + c] = hv[c];
				}
				b[KO] = inv_o * inv_o;                       // dL_dopacity = G dL_dalpha = w / opacity
				if constexpr (FOLD3) b[KO] *= b[HO + 3];
			}
		}
		const float ahcx = -0.5f * acx, ancy = -acy, ahcz = -0.5f * acz;
		// Candidate mask.  Entry-major first: the lane that owns entry e marks the pixels of this wave's 16x4 strip that
		// lie inside the entry's conservative alpha footprint (bit = 16*row + column = the pixel's lane).  A 64x64 bit
		// transpose then hands every pixel-lane the set of entries that may touch it -- ~120 instructions per chunk
		// instead of 64 broadcast pair tests.  The exact tests (position < last, power window, alpha >= 1/255) are done
		// in the walk on the one record the lane holds there.
		unsigned long long emask = 0ull;
		if (lane < m && ahx >= 0.f)
		{
			// Row by row: power(dx, dy) >= thr  <=>  cx dx^2 + 2 cy dy dx + (cz dy^2 + 2 thr) <= 0, an interval in dx
			// (d = mean - pixel).  Widened by 1 % + 0.01 px, so it stays a superset of the exact test done in the walk.
			const bool quad_ok = acx > 0.f && athr <= 0.f && ahx < 1e30f;
			const float racx = __builtin_amdgcn_rcpf(acx);
#pragma unroll
			for (unsigned r = 0; r < 4; r++)
			{
				const float dy = ay - (strip_lo + (float)r);
				float lo = ax - ahx, hi2 = ax + ahx;                  // box fallback (unknown / degenerate conic)
				bool any_px = fabsf(dy) <= ahy;
				if (quad_ok)
				{
					const float hb = acy * dy;                         // b / 2
					const float cq = acz * dy * dy + 2.0f * athr;
					const float disc = hb * hb - acx * cq;             // (b^2 - 4ac) / 4
					any_px = any_px && (disc >= 0.f);
					// approximate sqrt / reciprocal (1 ulp): the interval is widened by 1 % + 0.01 px anyway
					const float sq = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) * 1.01f + 0.01f * acx;
					const float dlo = (-hb - sq) * racx, dhi = (-hb + sq) * racx;   // dx in [dlo, dhi]
					lo = ax - dhi - 0.01f; hi2 = ax - dlo + 0.01f;               // pixel x = mean.x - dx
				}
				const float c0f = fmaxf(ceilf(lo) - tile_x0, 0.f), c1f = fminf(floorf(hi2) - tile_x0, 15.f);
				if (any_px && c0f <= c1f)
				{
					const unsigned c0 = (unsigned)c0f, c1 = (unsigned)c1f;
					const unsigned long long cols = (unsigned long long)(((2u << c1) - 1u) & ~((1u << c0) - 1u));
					emask |= cols << (16 * r);
				}
			}
		}
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		// Lane l owns list index hi-1-l; the pixel's contributors are the list indices below `last`, so the entries at or
		// behind its last contributor are the FIRST hi-last bits.
		{
			const int t = hi - last;
			mask = (t >= 64) ? 0ull : (t > 0) ? (mask & ~((1ull << t) - 1ull)) : mask;
			if (!inside) mask = 0ull;
		}
		// every lane walks its own set bits; the loop is wave-uniform so that all lanes take part in the bpermutes
		while (fr_any(mask != 0ull))
		{
			bool has = mask != 0ull;
			FR_STAT(dbg_steps++;)
			const int j = has ? (__ffsll((long long)mask) - 1) : 0;
			mask &= mask - 1ull;
			const float x = fr_bperm_f(ax, j), y = fr_bperm_f(ay, j);
			const float hcx = fr_bperm_f(ahcx, j), ncy = fr_bperm_f(ancy, j), hcz = fr_bperm_f(ahcz, j), lo = fr_bperm_f(alo, j);
			float r[NB];
#pragma unroll
			for (int q = 0; q < NB; q++)
				if (!(FOLD3 && q == HO + 3)) r[q] = fr_bperm_f(b[q], j);
			const float dx = x - pfx, dy = y - pfy;
			float power;
			const float e = fr_scorer_exponent(hcx, ncy, hcz, dx, dy, lo, power);     // bit-identical to pass 1's
			has = has && !(power > 0.0f) && !(e < FR_E255);
			FR_STAT(dbg_hits += (int)__popcll(__ballot(has));)
			const float a_un = __builtin_amdgcn_exp2f(e);                              // opacity * G
			// A lane without a contributor at this step runs the recurrences with alpha = 0, which leaves them unchanged
			// (T / 1, and the colour recurrence folds a zero-alpha layer away exactly): one select instead of nine.
			const float alpha = has ? fminf(0.99f, a_un) : 0.f;
			{
#pragma clang fp contract(fast)
				// backward.cu:978-1038
				const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
				st.T = st.T * inv;
				const float ola = 1.f - st.last_alpha;
				st.accum0 = st.last_alpha * st.lastc0 + ola * st.accum0; st.lastc0 = r[0];
				st.accum1 = st.last_alpha * st.lastc1 + ola * st.accum1; st.lastc1 = r[1];
				st.accum2 = st.last_alpha * st.lastc2 + ola * st.accum2; st.lastc2 = r[2];
				float dL_dalpha = ((r[0] - st.accum0) * g0 + (r[1] - st.accum1) * g1 + (r[2] - st.accum2) * g2) * st.T;
				st.last_alpha = alpha;
				if (bg_dot != 0.f) dL_dalpha += (-st.T_final * inv) * bg_dot;
				const float w = a_un * dL_dalpha;                                     // opacity * G * dL_dalpha
				const float w2 = w * w;
				float u[5];
				u[0] = hcx * dx + (hcx * dx + ncy * dy);                              // -(cx dx + cy dy)
				u[1] = 2.0f * (hcz * dy) + ncy * dx;                                  // -(cz dy + cy dx)
				u[2] = dx * dx; u[3] = dx * dy; u[4] = dy * dy;
				if constexpr (QFORM)
				{
					float add = r[KO];
					int q = 3;
#pragma unroll
					for (int i = 0; i < 5; i++)
					{
						float ti = 0.f;
#pragma unroll
						for (int j = i; j < 5; j++) ti += r[q++] * u[j];
						add += u[i] * ti;
					}
					score += has ? w2 * add : 0.f;
				}
				float leaf2[C];
				if constexpr (!QFORM)
				{
#pragma unroll
					for (int q = 0; q < 3; q++)
					{
						const float l = r[3 + q * 5 + 0] * u[0] + r[3 + q * 5 + 1] * u[1] + r[3 + q * 5 + 2] * u[2]
						              + r[3 + q * 5 + 3] * u[3] + r[3 + q * 5 + 4] * u[4];
						leaf2[q] = l * l;
					}
					leaf2[3] = r[KO];
					if constexpr (SR)
					{
#pragma unroll
						for (int q = 0; q < 7; q++)
						{
							const float l = r[18 + q * 3 + 0] * u[2] + r[18 + q * 3 + 1] * u[3] + r[18 + q * 3 + 2] * u[4];
							leaf2[4 + q] = l * l;
						}
					}
					if constexpr (FULL)
					{
						// colour: dL_dcolor_c = alpha T dL_dpix_c (not a multiple of w: pre-divided by w2, which multiplies every column below)
						const float wc = alpha * st.T, iw2 = has ? __builtin_amdgcn_rcpf(w2) : 0.f;
						leaf2[11] = (wc * g0) * (wc * g0) * iw2; leaf2[12] = (wc * g1) * (wc * g1) * iw2; leaf2[13] = (wc * g2) * (wc * g2) * iw2;
						leaf2[14] = (u[0] * ddelx_dx) * (u[0] * ddelx_dx); leaf2[15] = (u[1] * ddely_dy) * (u[1] * ddely_dy);
#pragma unroll
						for (int q = 0; q < 6; q++)
						{
							const float l = r[39 + q * 3 + 0] * u[2] + r[39 + q * 3 + 1] * u[3] + r[39 + q * 3 + 2] * u[4];
							leaf2[16 + q] = l * l;
						}
						leaf2[22] = 0.25f * u[2] * u[2]; leaf2[23] = 0.25f * u[3] * u[3]; leaf2[24] = 0.25f * u[4] * u[4];
					}
					if constexpr (HAS_HINV)
					{
						float add = FOLD3 ? leaf2[3] : leaf2[3] * r[HO + 3];
#pragma unroll
						for (int c = 0; c < C; c++)
							if (c != 3) add += leaf2[c] * r[HO + c];
						score += has ? w2 * add : 0.f;
					}
				}
				(void)leaf2;
				if constexpr (HAS_OUTH)
				{
					if (has)
					{
#pragma unroll
						for (int c = 0; c < C; c++) atomicAdd(&s_acc[wave][c][j], (double)(w2 * leaf2[c]));
					}
				}
			}
		}
		if constexpr (HAS_OUTH)
		{
			// wave-private LDS: the accumulators are complete once this wave's own ds_add instructions have retired
			__builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
			// Flush with consecutive lanes on consecutive columns of one entry: an atomic instruction then touches 64 / C rows
			// of out_H (C contiguous floats each) instead of 64 -- the L2 sees a quarter (C = 4) of the requests.
			float* dst = FULL ? nullptr : f.out_H + (size_t)v * f.outH_stride;
#pragma unroll
			for (int i = 0; i < C; i++)
			{
				const int flat = i * 64 + lane;
				const int e = flat / C, c = flat - e * C;
				const uint32_t id_e = (uint32_t)__builtin_amdgcn_ds_bpermute(e << 2, (int)my_id);
				const float a = (e < m) ? (float)s_acc[wave][c][e] : 0.f;
				if constexpr (FULL)
				{
					// column -> (gradient tensor, row stride, offset): means3D 0-2, opacity 3, scales 4-6, rotations 7-10, colours 11-13,
					// means2D 14-15 (z unused), cov3D 16-21, conic 22-24 (entries 0, 1, 3 of the 2x2)
					const int arr = c < 3 ? 0 : c < 4 ? 1 : c < 7 ? 2 : c < 11 ? 3 : c < 14 ? 4 : c < 16 ? 5 : c < 22 ? 6 : 7;
					const int first = arr == 0 ? 0 : arr == 1 ? 3 : arr == 2 ? 4 : arr == 3 ? 7 : arr == 4 ? 11 : arr == 5 ? 14 : arr == 6 ? 16 : 22;
					const int stride = arr == 1 ? 1 : (arr == 3 || arr == 7) ? 4 : arr == 6 ? 6 : 3;
					const int off = (c == 24) ? 3 : c - first;
					if (a != 0.f) atomicAdd(f.full_out[arr] + (size_t)id_e * stride + off, a);
				}
				else
				{
					if (a != 0.f) atomicAdd(dst + (size_t)id_e * C + c, a);
				}
			}
		}
	}
	if constexpr (!HAS_HINV) { (void)score; return; }
	float ws = wave_sum(score);
#ifdef FR_LOOPSTATS
	if (f.debug_mode >= 2)
		ws = f.debug_mode == 2 ? (float)dbg_p1 : f.debug_mode == 3 ? (float)dbg_chunks : f.debug_mode == 4 ? (float)dbg_steps : f.debug_mode == 5 ? (float)dbg_hits : f.debug_mode == 7 ? (float)dbg_p1any : f.debug_mode == 8 ? (float)dbg_p1con : (float)wcnt;
#endif
	if (lane == 0) s_red[wave] = ws;
	__syncthreads();
	if (tid == 0) f.tile_scores[vt] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---------------------------------------------------------------------------------------------------------
// Third-generation scorer (score-only mode: H_inv given, no out_H, constant upstream gradient): ONE front-to-back pass.
//
// The reference's backward (backward.cu:960-1038) walks a pixel's contributors back to front because dL_dalpha_i needs
// the colour accumulated BEHIND splat i.  With the same upstream gradient g on every channel, write cg_i = sum_ch c_i[ch],
// b_i = 1/(1 - alpha_i), Cg_<=i = sum_{j<=i} cg_j alpha_j T_j (front-to-back prefix of the composited colour) and
// X = Cg_final + T_final * sum_ch bg[ch]  (known only when the pixel is finished).  Then
//     dL_dalpha_i / g = T_i cg_i - (X - Cg_<=i) b_i = p_i - X b_i ,      p_i = T_i cg_i + Cg_<=i b_i ,
// and the pixel's share of the score, sum_i S_i (p_i - X b_i)^2 with S_i = (opacity G_i)^2 (u'^T Q_i u' + k3_i) the
// geometry factor of the pair (see k_fisher_records), expands to  A - 2 X B + X^2 D  with three running sums
//     A = sum S_i p_i^2 ,   B = sum S_i p_i b_i ,   D = sum S_i b_i^2
// that a FORWARD walk can keep.  The transmittance pass, the per-strip contributor lists and the second gather of every
// record disappear.  To keep the expansion from cancelling on low-contrast pixels the sums are taken about a per-pixel
// centre Xt (the colour of the pixel's first contributor): p'_i = p_i - Xt b_i, result A' - 2 (X - Xt) B' + (X - Xt)^2 D.
// Contributor rules are the forward's (forward.cu:347-366): power > 0 or alpha < 1/255 skips, test_T < 1e-4 ends the pixel.
//
// Everything that depends on (view, Gaussian) only -- the Jacobian chain of backward.cu:335-475 pushed through unit vectors
// and folded with H_inv into the 5x5 form Q -- is computed ONCE per visible (view, Gaussian) by k_fisher_records instead of
// once per (strip, list entry) inside the tile kernel, and stored as 96 bytes:
//     recA {x, y, ext, log2(opacity)}   recB {-conic.x/2, -conic.y, -conic.z/2, cg}    (in place of the FrSplat record)
//     recQ {c02 c03 c04 c11 | c12 c13 c20 c21 | c22 c30 c31 c40 | k3}                   (64 B, [V][P]): the pair's geometry factor
//          u'^T Q u' + k3 written out as a polynomial in (dx, dy) -- terms of degree 2..4 only, 12 coefficients -- which the walk
//          evaluates by Horner's rule in 14 operations (26 for the quadratic form)
// ---------------------------------------------------------------------------------------------------------
// REWRITE: the (view, Gaussian) record still holds the rasteriser's FrSplat and is turned into {recA, recB} here (stand-alone
// k_fisher_records); otherwise the front end has already written {recA, recB} and only recQ is produced.
template <int C, bool REWRITE, bool FORM_A, bool FIVE>
__device__ __forceinline__ void fr_fisher_record_one(const FrParams& p, const float* __restrict__ H_inv, long long hinv_stride,
                                                     const float* __restrict__ packed, float4* __restrict__ recq, int v, uint32_t id,
                                                     const float* vm, const float* pm, const float* wm, bool has_w2c,
                                                     float4* out6, const float4* ab_src)
{
	constexpr int PS = FrPackSize<C>::value;
	constexpr bool SR = C >= 11;
	float4* sp = (float4*)(p.splat + (size_t)v * p.P + id);
	float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;          // {x, y, cx, cy} {cz, o, depth, ext}
	float opacity;
	// (p.order: `id` is a position of the processing order; opacity and per-view weights are read at the caller's index)
	const uint32_t oid = p.order ? p.order[id] : id;
	if constexpr (REWRITE) { a0 = sp[0]; a1 = sp[1]; opacity = a1.y; }
	else opacity = p.opac[oid];
	float gsv[PS];
	const float4* pk = (const float4*)(packed + (size_t)id * PS);
#pragma unroll
	for (int q = 0; q < PS / 4; q++) { const float4 t4 = pk[q]; gsv[4 * q] = t4.x; gsv[4 * q + 1] = t4.y; gsv[4 * q + 2] = t4.z; gsv[4 * q + 3] = t4.w; }
	const fr_f3 pw = { gsv[0], gsv[1], gsv[2] };
	const fr_f3 po = has_w2c ? fr_world_to_cam(pw, wm) : pw;
	// The per-Gaussian rows of the leaves over gamma(u) = (ux, uy, ux^2, ux uy, uy^2), u = -conic d (fr_math.h: the basis in which
	// the chain of backward.cu:335-475 has no structural cancellation; written in (dx, dy) a needle-shaped splat loses the
	// square of its conic's condition number).
	float Rg[3][5];
	float Bg[6][3];
	fr_mean_rows_g<true>(po, &gsv[3], vm, pm, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, p.W, p.H, Rg, SR ? Bg : nullptr, nullptr, nullptr);
	float Cg[SR ? 7 : 1][3];
	int go = 12;
	if constexpr (SR)
	{
		const fr_f3 sc = { gsv[12], gsv[13], gsv[14] };
		const fr_f4 qr = { gsv[15], gsv[16], gsv[17], gsv[18] };
		fr_scale_rot_jacobian(sc, p.mod, qr, Bg, Cg);
		go = 19;
	}
	if constexpr (FORM_A)
	{
		// out_H mode (k_fisher_tile_v3h): the walk needs the three mean rows themselves, not their H_inv-weighted form
		static_assert(!FORM_A || C == 4, "the A-form record holds the 3 x 5 mean rows only");
		const float inv_oa = __builtin_amdgcn_rcpf(opacity);
		if constexpr (REWRITE)
		{
			sp[0] = make_float4(a0.x, a0.y, a1.w, __builtin_amdgcn_logf(a1.y));
			sp[1] = make_float4(-0.5f * a0.z, -a0.w, -0.5f * a1.x, gsv[9] + gsv[10] + gsv[11]);
		}
		float4* da = out6 ? out6 + 2 : recq + ((size_t)v * p.P + id) * 4;
		da[0] = make_float4(Rg[0][0], Rg[0][1], Rg[0][2], Rg[0][3]);
		da[1] = make_float4(Rg[0][4], Rg[1][0], Rg[1][1], Rg[1][2]);
		da[2] = make_float4(Rg[1][3], Rg[1][4], Rg[2][0], Rg[2][1]);
		da[3] = make_float4(Rg[2][2], Rg[2][3], Rg[2][4], inv_oa * inv_oa);
		if (out6) { out6[0] = ab_src[0]; out6[1] = ab_src[1]; }          // compact record: {recA, recB} travel with the coefficients
		return;
	}
	float hv[C];
	if (hinv_stride != 0)
	{
		const float* hp = H_inv + (size_t)v * hinv_stride + (size_t)oid * C;
#pragma unroll
		for (int c = 0; c < C; c++) hv[c] = hp[c];
	}
	else
	{
#pragma unroll
		for (int c = 0; c < C; c++) hv[c] = gsv[go + c];
	}
	// F(u) = sum_c H_inv[c] (R_c . gamma(u))^2 is a bivariate polynomial in (ux, uy) with terms of degree 2, 3 and 4 only: 12
	// coefficients, which the walk evaluates by Horner's rule in 14 operations (26 for the quadratic form).
	float qp[12];
	fr_scorer_poly_g<C>(Rg, SR ? Cg : nullptr, hv, qp);
	const float inv_o = __builtin_amdgcn_rcpf(opacity);
	const float k3 = inv_o * inv_o * hv[3];              // dL_dopacity = G dL_dalpha = w / opacity, weighted by H_inv[3]
	if constexpr (REWRITE)
	{
		sp[0] = make_float4(a0.x, a0.y, a1.w, __builtin_amdgcn_logf(a1.y));
		sp[1] = make_float4(-0.5f * a0.z, -a0.w, -0.5f * a1.x, gsv[9] + gsv[10] + gsv[11]);
	}
	if constexpr (FIVE)
	{
		// the 80-byte record of the fixed-segment path: {x, y, k3, log2 o} {recB} {12 coefficients} -- what the walk parks
		out6[0] = make_float4(ab_src[0].x, ab_src[0].y, k3, ab_src[0].w);
		out6[1] = ab_src[1];
		out6[2] = make_float4(qp[0], qp[1], qp[2], qp[3]);
		out6[3] = make_float4(qp[4], qp[5], qp[6], qp[7]);
		out6[4] = make_float4(qp[8], qp[9], qp[10], qp[11]);
		return;
	}
	float4* dq = out6 ? out6 + 2 : recq + ((size_t)v * p.P + id) * 4;
	dq[0] = make_float4(qp[0], qp[1], qp[2], qp[3]);
	dq[1] = make_float4(qp[4], qp[5], qp[6], qp[7]);
	dq[2] = make_float4(qp[8], qp[9], qp[10], qp[11]);
	dq[3] = make_float4(k3, 0.f, 0.f, 0.f);
	if (out6) { out6[0] = ab_src[0]; out6[1] = ab_src[1]; }              // compact record: {recA, recB} travel with the coefficients (loaded late: fewer live registers)
}

// The general out_H record (k_fisher_tile_v3g: 11 columns and / or a per-pixel upstream-gradient image): {recA, recB} and
//   q[0..14] the three mean rows over gamma(u) (fr_mean_rows_g)   q[15] 1 / opacity^2   q[16..18] r, g, b   q[19] 0
//   C = 11:  q[20..40] the seven scale / rotation rows over (ux^2, ux uy, uy^2)   q[41..43] 0
// = 7 float4 at C = 4, 13 at C = 11 (odd strides: sixteen consecutive parked records cover all LDS banks).
template <int C>
__device__ __forceinline__ void fr_fisher_record_general(const FrParams& p, const float* __restrict__ packed, int v, uint32_t id,
                                                         const float* vm, const float* pm, const float* wm, bool has_w2c,
                                                         float4* out, const float4* ab_src)
{
	constexpr int PS = FrPackSize<C>::value;
	constexpr bool SR = C >= 11;
	(void)v;
	float gsv[PS];
	const float4* pk = (const float4*)(packed + (size_t)id * PS);
#pragma unroll
	for (int q = 0; q < PS / 4; q++) { const float4 t4 = pk[q]; gsv[4 * q] = t4.x; gsv[4 * q + 1] = t4.y; gsv[4 * q + 2] = t4.z; gsv[4 * q + 3] = t4.w; }
	const fr_f3 pw = { gsv[0], gsv[1], gsv[2] };
	const fr_f3 po = has_w2c ? fr_world_to_cam(pw, wm) : pw;
	float Rg[3][5];
	float Bg[6][3];
	// IEEE division (not v_rcp_f32): these records feed per-ENTRY outputs (1e-4 per element, near-plane splats included)
	fr_mean_rows_g<false>(po, &gsv[3], vm, pm, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, p.W, p.H, Rg, SR ? Bg : nullptr, nullptr, nullptr);
	const float opacity = p.opac[p.order ? p.order[id] : id];
	const float inv_o = 1.0f / opacity;
	out[2] = make_float4(Rg[0][0], Rg[0][1], Rg[0][2], Rg[0][3]);
	out[3] = make_float4(Rg[0][4], Rg[1][0], Rg[1][1], Rg[1][2]);
	out[4] = make_float4(Rg[1][3], Rg[1][4], Rg[2][0], Rg[2][1]);
	out[5] = make_float4(Rg[2][2], Rg[2][3], Rg[2][4], inv_o * inv_o);
	out[6] = make_float4(gsv[9], gsv[10], gsv[11], 0.f);
	if constexpr (SR)
	{
		const fr_f3 sc = { gsv[12], gsv[13], gsv[14] };
		const fr_f4 qr = { gsv[15], gsv[16], gsv[17], gsv[18] };
		float Cg[7][3];
		fr_scale_rot_jacobian(sc, p.mod, qr, Bg, Cg);
		out[7] = make_float4(Cg[0][0], Cg[0][1], Cg[0][2], Cg[1][0]);
		out[8] = make_float4(Cg[1][1], Cg[1][2], Cg[2][0], Cg[2][1]);
		out[9] = make_float4(Cg[2][2], Cg[3][0], Cg[3][1], Cg[3][2]);
		out[10] = make_float4(Cg[4][0], Cg[4][1], Cg[4][2], Cg[5][0]);
		out[11] = make_float4(Cg[5][1], Cg[5][2], Cg[6][0], Cg[6][1]);
		out[12] = make_float4(Cg[6][2], 0.f, 0.f, 0.f);
	}
	out[0] = ab_src[0]; out[1] = ab_src[1];
}

// Stand-alone form of phase C of k_preprocess_views, for the single-view front end (images beyond FR_MAX_LDS_TILES tiles,
// visibility from radii) -- or over the compact lists (LIST).  Needs the projection only, not the keys.
template <int C, bool LIST, bool FORM_A>
__global__ __launch_bounds__(FR_THREADS) void k_fisher_records(FrParams p, FrRecordArgs ra)
{
	if (p.status[1]) return;
	const int tid = threadIdx.x;
	const int v = blockIdx.y;
	const uint32_t nblk = gridDim.x;
	float vm[16], pm[16], wm[12];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	const bool has_w2c = p.w2c != nullptr;
#pragma unroll
	for (int k = 0; k < 12; k++) wm[k] = has_w2c ? p.w2c[16 * (size_t)v + k] : 0.f;
	if constexpr (LIST)
	{
		const uint32_t n = p.vis_n[(size_t)v * nblk + blockIdx.x];
		const FrVisEntry* list = p.vis_list + ((size_t)v * nblk + blockIdx.x) * (size_t)(FR_THREADS * p.G);
		for (uint32_t e = tid; e < n; e += FR_THREADS)
			fr_fisher_record_one<C, false, FORM_A>(p, ra.H_inv, ra.hinv_stride, ra.packed, ra.recq, v, list[e].idx, vm, pm, wm, has_w2c);   // {recA, recB} came from k_preprocess_views
	}
	else
	{
		for (int g = 0; g < p.G; g++)
		{
			const int i = (blockIdx.x * p.G + g) * FR_THREADS + tid;
			if (i < p.P && p.radii[(size_t)v * p.P + i] > 0)
				fr_fisher_record_one<C, true, FORM_A>(p, ra.H_inv, ra.hinv_stride, ra.packed, ra.recq, v, (uint32_t)i, vm, pm, wm, has_w2c);
		}
	}
}

#define FR_ENT_F4 7                   // float4 per parked candidate record: 6 used + 1 pad -- a 28-dword stride spreads sixteen consecutive
                                     // candidates over all 64 LDS banks for ds_read_b128 (a 24-dword stride repeats after eight)
typedef float fr_v4f __attribute__((ext_vector_type(4)));
// One candidate's parked record as k_fisher_tile_v3h reads it back (six ds_read_b128).
struct FrWalkRec { fr_v4f a, b4, q0, q1, q2, q3; };     // {x, y, ext, log2 o} {-cx/2, -cy, -cz/2, cg} {A'[15], 1/o^2}
// ... and as k_fisher_tile_v3 does (five): {x, y, k3, log2 o} {-cx/2, -cy, -cz/2, cg} {c02 c03 c04 c11} {c12 c13 c20 c21} {c22 c30 c31 c40}
#define FR_ENT3_F4 5                 // a 20-dword stride is conflict-free for ds_read_b128 like the 28-dword one (5 and 16 coprime)
struct FrWalkRec3 { fr_v4f a, b4, q0, q1, q2; };
// Everything about one (pixel, candidate) pair that does not depend on the pixel's running state.
struct FrWalkGeom { bool ok; float a_un, alpha, om1, bi, cg, add; };

__device__ __forceinline__ FrWalkGeom fr_walk_geom(const FrWalkRec3& r, float pfx, float pfy)
{
#pragma clang fp contract(fast)
	FrWalkGeom g;
	const float dx = r.a.x - pfx, dy = r.a.y - pfy;
	float power;
	const float e = fr_scorer_exponent(r.b4.x, r.b4.y, r.b4.z, dx, dy, r.a.w, power);
	// forward.cu:347-357 (power > 0 -> skip, alpha < 1/255 -> skip); NaN falls through as it does there
	g.ok = !(power > 0.0f) && !(e < FR_E255);
	g.a_un = __builtin_amdgcn_exp2f(e);                                    // opacity * G
	g.alpha = fminf(0.99f, g.a_un);
	g.om1 = 1.f - g.alpha;
	g.bi = __builtin_amdgcn_rcpf(g.om1);
	g.cg = r.b4.w;
	// F(u) + k3 as the bivariate polynomial of fr_fisher_record_one (fr_scorer_poly_g; terms of degree 2..4), u = -conic d,
	// Horner in ux over Horner in uy (fr_scorer_poly_eval's order)
	const float ux = __builtin_fmaf(r.b4.x, dx, __builtin_fmaf(r.b4.x, dx, r.b4.y * dy));      // 2 hcx dx + ncy dy
	const float uy = __builtin_fmaf(2.0f * r.b4.z, dy, r.b4.y * dx);                           // ncy dx + 2 hcz dy
	const float A0 = r.q0.x + uy * (r.q0.y + uy * r.q0.z);
	const float A1 = r.q0.w + uy * (r.q1.x + uy * r.q1.y);
	const float A2 = r.q1.z + uy * (r.q1.w + uy * r.q2.x);
	const float A3 = r.q2.y + uy * r.q2.z;
	const float uy2 = uy * uy;
	const float in3 = A3 + ux * r.q2.w;
	const float in2 = A2 + ux * in3;
	const float in1 = uy * A1 + ux * in2;
	const float add = (r.a.z + uy2 * A0) + ux * in1;
	g.add = (g.a_un * g.a_un) * add;                                       // S_i = (opacity G)^2 (u'^T Q u' + k3)
	return g;
}

// The pixel's recurrences for one candidate (`live`: it passed the pair tests and the pixel is not finished).
// Returns true when the candidate ENDS the pixel (forward.cu:358-363: test_T < 1e-4, nothing is added).
__device__ __forceinline__ bool fr_walk_update(const FrWalkGeom& g, bool live, float& T, float& Cg, float& Xt, float& sA, float& sB, float& sD)
{
#pragma clang fp contract(fast)
	const float test_T = T * g.om1;
	const bool con = live && !(test_T < 0.0001f);
	const bool kill = live != con;                                           // one compare; the complement is a scalar xor of the two lane masks
	const float S = con ? g.add : 0.f;
	Xt = (con && T == 1.0f) ? g.cg : Xt;                                   // centre: the first contributor's colour
	Cg = con ? Cg + g.cg * (g.alpha * T) : Cg;
	const float pc = (Cg - Xt) * g.bi + T * g.cg;                          // p_i - Xt b_i
	T = con ? test_T : T;
	const float Sp = S * pc, Sb = S * g.bi;
	sA += Sp * pc; sB += Sp * g.bi; sD += Sb * g.bi;
	return kill;
}

// FR_LDK: the tile kernel's key loads (-DFR_NT_KEYS: non-temporal, to keep the keys out of the L2 the record gathers live in:
// measured 1-4 % slower, tools/ab.sh)
#ifdef FR_NT_KEYS
#define FR_LDK(p) __builtin_nontemporal_load(p)
#else
#define FR_LDK(p) (*(p))
#endif
#define FR_QCAP 128                  // per-wave candidate queue (ring of Gaussian indices): at most 63 left over + 64 new
// One workgroup per (tile, view); the four waves own the four 16x4 strips and never synchronise until the final sum.
//  stream   a wave reads the tile's sorted keys 64 at a time (one per lane), gathers recA and keeps the splats whose
//           conservative alpha footprint meets its strip (one ballot); survivors are appended, in order, to a ring in LDS;
//  chunk    64 queued candidates, one per lane: the lane gathers recB + recQ and parks an 80-byte record in LDS (k3 in the
//           place of the footprint extents, which only this step needs); it
//           rasterises its footprint ellipse row by row into a 64-bit mask over the strip's pixels, and a 64x64 bit
//           transpose across the wave hands every pixel-lane the mask of candidates that may touch it;
//  walk     every pixel-lane walks its own set bits front to back: five ds_read_b128 of the candidate's record, the pair
//           test, the transmittance / colour prefix recurrences and the three sums.  A finished pixel clears its masks;
//           a wave whose 64 pixels are finished leaves.
// BW x BH = the 64 pixels of a wave inside the 16 x 16 tile: 16 x 4 strips (used), or 8 x 8 blocks (5 % slower on MI355X).
// MK: the keys carry, in their low four bits, the strips of the tile their splat's footprint reaches (fixed key segments; the
// front end has made the stream step's test once per (splat, tile)): a wave keeps the keys with ITS bit -- no recA gather per key.
// Waves per SIMD: 5 with the gathering stream step (6 and 7 measured no faster in round 2); the MK form needs 69 registers and
// runs 7 (0.745-0.755 ms against 0.785 at 5: the walk is a chain of dependent LDS reads and arithmetic, more waves hide more of it).
template <int BW, int BH, bool MK = false>
#ifndef FR_V3_WAVES
#define FR_V3_WAVES 5
#endif
__global__ __launch_bounds__(FR_THREADS) __attribute__((amdgpu_waves_per_eu(4, MK ? 7 : FR_V3_WAVES)))
void k_fisher_tile_v3(FrParams p, FrFisherArgs f, const float4* __restrict__ recq)
{
	static_assert(!MK || (BW == 16 && BH == 4), "the keys' strip bits are those of 16 x 4 strips");
	static_assert(BW * BH == 64 && 16 % BW == 0, "a wave owns 64 pixels of the tile");
	__shared__ uint32_t s_q[4][FR_QCAP];
	__shared__ float4 s_ent[4][64][FR_ENT3_F4];
	__shared__ float s_red[4];
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	uint32_t tile; int v;
	fr_tile_of_block(p, tile, v);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	constexpr int WPR = 16 / BW;                   // waves per tile row
	const uint32_t bx0 = tx * FR_BLOCK_X + (uint32_t)(wave % WPR) * BW, by0 = ty * FR_BLOCK_Y + (uint32_t)(wave / WPR) * BH;
	const uint32_t pxx = bx0 + (uint32_t)(lane % BW), pxy = by0 + (uint32_t)(lane / BW);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const size_t vP = (size_t)v * p.P;
	const uint32_t n = p.tile_cnt[vt];
	const uint64_t* gk = p.keys + p.tile_off[vt];
	(void)vP; (void)recq;
	const float4* rec = f.recA + (size_t)v * f.ab_view;
	const float4* rq = f.recQ + (size_t)v * f.q_view;
	const size_t rsA = (size_t)f.ab_stride, rsQ = (size_t)f.q_stride;
	uint32_t* wq = s_q[wave];
	float4 (*ent)[FR_ENT3_F4] = s_ent[wave];
	const uint32_t ent_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&s_ent[wave][0][0];   // LDS byte address
	uint32_t ent_lds_v;                            // ... kept in a vector register: the walk's address is then one v_mad per step
	asm volatile("v_mov_b32 %0, %1" : "=v"(ent_lds_v) : "s"(ent_lds));

	const float strip_lo = (float)by0, strip_hi = strip_lo + (float)(BH - 1);      // the wave's rows
	const float tile_x0 = (float)bx0, tile_x1 = tile_x0 + (float)(BW - 1);         // ... and columns
	float T = 1.0f, Cg = 0.f, Xt = 0.f, sA = 0.f, sB = 0.f, sD = 0.f;
	bool done = !inside;
	uint32_t qh = 0, qn = 0;                       // ring head / fill, wave-uniform
#ifdef FR_LOOPSTATS
	int dbg_cand = 0, dbg_chunks = 0, dbg_steps = 0, dbg_hits = 0, dbg_wsteps = 0, dbg_cs = 0, dbg_empty = 0;
	long long dbg_t0 = 0, dbg_ts = 0, dbg_tc = 0, dbg_tw = 0;     // s_memtime sums: stream / chunk set-up / walk
#endif

	// software pipeline of the key stream: id1 / r1 = indices and recA of the chunk at `base`, id2 = indices of the next one
	uint32_t id1 = 0, id2 = 0;
	float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
	if ((uint32_t)lane < n) { id1 = (uint32_t)FR_LDK(gk + lane); if constexpr (!MK) r1 = rec[rsA * id1]; }
	if (64u + lane < n) id2 = (uint32_t)FR_LDK(gk + 64 + lane);
	uint32_t base = 0;
	// ---- stream: fill the queue up to one chunk
	auto stream_fill = [&]() {
#ifdef FR_LOOPSTATS
		dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
		while (qn < 64u && base < n)
		{
			const uint32_t idc = id1; const float4 rc = r1;
			id1 = id2;
			if constexpr (!MK) { if (base + 64 + lane < n) r1 = rec[rsA * id2]; }
			if (base + 128 + lane < n) id2 = (uint32_t)FR_LDK(gk + base + 128 + lane);
			bool ov = base + lane < n;
			if constexpr (MK) ov = ov && ((idc >> wave) & 1u);
			else
			{
				const uint32_t eb = __float_as_uint(rc.z);
				const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
				const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
				ov = ov && hx >= 0.f && (rc.y + hy >= strip_lo) && (rc.y - hy <= strip_hi) && (rc.x + hx >= tile_x0) && (rc.x - hx <= tile_x1);
			}
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			if (ov) wq[(qh + qn + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) & (FR_QCAP - 1)] = MK ? (idc >> 4) : idc;
			qn += (uint32_t)__popcll(om);
			base += 64;
		}
#ifdef FR_LOOPSTATS
		{ const long long t = (long long)__builtin_amdgcn_s_memtime(); dbg_ts += t - dbg_t0; dbg_t0 = t; }
#endif
	};
	// ---- the next chunk's records, gathered into registers one chunk ahead (the gathers then fly during the walk of the
	// current chunk instead of in front of it); up to 64 candidates, one per lane
	float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pq0 = pa, pq1 = pa, pq2 = pa;
	float pk3 = 0.f;
	uint32_t pm = 0;                                 // candidates in the gathered chunk, wave-uniform
	auto gather_next = [&]() {
		pm = qn < 64u ? qn : 64u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if ((uint32_t)lane < pm)
		{
			const uint32_t id = wq[(qh + lane) & (FR_QCAP - 1)];
			pa = rec[rsA * id]; pb = rec[rsA * id + 1];
			pq0 = rq[rsQ * id]; pq1 = rq[rsQ * id + 1]; pq2 = rq[rsQ * id + 2];
			if constexpr (MK) pk3 = pa.z;              // (the 80-byte record: k3 already sits where the walk wants it)
			else pk3 = ((const float*)(rq + rsQ * id + 3))[0];
		}
		qh = (qh + pm) & (FR_QCAP - 1); qn -= pm;
	};
	stream_fill();
	gather_next();
	bool all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
	while (!all_done && pm != 0u)
	{
		// ---- chunk: park the gathered records in LDS, rasterise the footprints
		const uint32_t m = pm;
#ifdef FR_LOOPSTATS
		dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
		dbg_chunks++; dbg_cand += (int)m;
#endif
		unsigned long long emask = 0ull;
		if ((uint32_t)lane < m)
		{
			const float4 a = pa, b4 = pb;
			ent[lane][0] = make_float4(a.x, a.y, pk3, a.w);       // the footprint extents are only needed here, k3 takes their place
			ent[lane][1] = b4; ent[lane][2] = pq0; ent[lane][3] = pq1; ent[lane][4] = pq2;
			const float ax = a.x, ay = a.y;
			const uint32_t eb = __float_as_uint(a.z);
			// (MK: no extents in the record -- a listed splat's footprint rows come from the conic alone; a conic the quadratic
			// cannot take covers the strip)
			const float ahx = MK ? 1e30f : __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
			const float ahy = MK ? 1e30f : __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
			const float acx = -2.0f * b4.x, acy = -b4.y, acz = -2.0f * b4.z;
			// conservative lower bound on `power` below which alpha < 1/255: -ln(255 opacity) - 0.01 (fr_power_threshold)
			const float athr = -(5.541263545158426f + 0.6931471805599453f * a.w) - 0.01f;
			// Row by row: power(dx, dy) >= thr  <=>  cx dx^2 + 2 cy dy dx + (cz dy^2 + 2 thr) <= 0, an interval in dx
			// (d = mean - pixel).  Widened by 1 % + 0.01 px, so it stays a superset of the exact test done in the walk.
			const bool quad_ok = acx > 0.f && athr <= 0.f && (MK || ahx < 1e30f);
			const float racx = __builtin_amdgcn_rcpf(acx);
#pragma unroll
			for (unsigned r = 0; r < (unsigned)BH; r++)
			{
				const float dy = ay - (strip_lo + (float)r);
				float lo = ax - ahx, hi2 = ax + ahx;                  // box fallback (unknown / degenerate conic)
				bool any_px = fabsf(dy) <= ahy;
				if (quad_ok)
				{
					const float hb = acy * dy;                         // b / 2
					const float cq = acz * dy * dy + 2.0f * athr;
					const float disc = hb * hb - acx * cq;             // (b^2 - 4ac) / 4
					any_px = any_px && (disc >= 0.f);
					const float sq = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) * 1.01f + 0.01f * acx;
					const float dlo = (-hb - sq) * racx, dhi = (-hb + sq) * racx;   // dx in [dlo, dhi]
					lo = ax - dhi - 0.01f; hi2 = ax - dlo + 0.01f;               // pixel x = mean.x - dx
				}
				const float c0f = fmaxf(ceilf(lo) - tile_x0, 0.f), c1f = fminf(floorf(hi2) - tile_x0, (float)(BW - 1));
				if (any_px && c0f <= c1f)
				{
					const unsigned c0 = (unsigned)c0f, c1 = (unsigned)c1f;
					const unsigned long long cols = (unsigned long long)(((2u << c1) - 1u) & ~((1u << c0) - 1u));
					emask |= cols << (BW * r);
				}
			}
		}
#ifdef FR_LOOPSTATS
		{ const long long t = (long long)__builtin_amdgcn_s_memtime(); dbg_tc += t - dbg_t0; }
#endif
#ifdef FR_LOOPSTATS
		dbg_empty += (int)__popcll(__builtin_amdgcn_ballot_w64((uint32_t)lane < m && emask == 0ull));      // parked candidates whose footprint misses every pixel of the strip
#endif
		// ---- refill the queue and start the next chunk's gathers before the transpose and the walk of this one
		stream_fill();
		gather_next();
#ifdef FR_LOOPSTATS
		dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		if (done) mask = 0ull;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
#ifdef FR_LOOPSTATS
		{ const long long t = (long long)__builtin_amdgcn_s_memtime(); dbg_tc += t - dbg_t0; dbg_t0 = t; }
#endif
		// ---- walk: every pixel-lane walks its own candidates front to back (the set bits of `mask`).  The candidate's 80-byte
		// record comes back as five ds_read_b128 (hipcc splits plain float4 LDS loads into dword pairs here).
		// The number of steps (set by the busiest lane of the chunk) times a per-step cost is what this loop costs; DESIGN.md
		// section 4 lists what was tried on it.
		while (mask != 0ull)
		{
			const int j = __ffsll((long long)mask) - 1;
			mask &= mask - 1ull;
			FrWalkRec3 r;
			{
				const uint32_t addr = ent_lds_v + (uint32_t)j * (FR_ENT3_F4 * 16);
				asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\t"
				             "ds_read_b128 %3, %5 offset:48\n\tds_read_b128 %4, %5 offset:64\n\t"
				             "s_waitcnt lgkmcnt(0)"
				             : "=&v"(r.a), "=&v"(r.b4), "=&v"(r.q0), "=&v"(r.q1), "=&v"(r.q2) : "v"(addr) : "memory");
			}
#ifdef FR_LOOPSTATS
			dbg_steps++; dbg_cs++;
#endif
			const FrWalkGeom g = fr_walk_geom(r, pfx, pfy);
			const bool kill = fr_walk_update(g, g.ok, T, Cg, Xt, sA, sB, sD);
#ifdef FR_LOOPSTATS
			dbg_hits += (g.ok && !kill) ? 1 : 0;
#endif
			if (kill) { mask = 0ull; done = true; }
		}
		all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
#ifdef FR_LOOPSTATS
		{ const long long t = (long long)__builtin_amdgcn_s_memtime(); dbg_tw += t - dbg_t0; dbg_t0 = t; }
		dbg_wsteps += wave_max_i(dbg_cs); dbg_cs = 0;
#endif
	}
	// X = Cg_final + T_final * sum(bg); pixel = A' - 2 (X - Xt) B' + (X - Xt)^2 D, times dL^2
	const float X = Cg + T * (p.bg[0] + p.bg[1] + p.bg[2]);
	const float dlt = X - Xt;
	float score = inside ? (sA + dlt * (dlt * sD - 2.0f * sB)) : 0.f;
	float ws = wave_sum(score);
#ifdef FR_LOOPSTATS
	if (f.debug_mode >= 2)
	{
		// 2: candidates, 3: chunks, 4: wave-level walk iterations, 5: contributing pairs, 6: lane-level walk steps, 7: steps of the busiest lane
		// 10 / 11 / 12: the wave's s_memtime ticks (/ 64) in the key stream / the chunk set-up / the walk
		if (f.debug_mode == 15) ws = (float)dbg_empty;                        // candidates parked with an empty footprint mask
		else if (f.debug_mode == 13) ws = (float)(base < n ? base : n);            // keys this wave streamed before its 64 pixels were finished
		else if (f.debug_mode == 14) ws = (float)n;                           // ... of the tile's n (both summed over the four waves)
		else if (f.debug_mode >= 10) ws = (float)((f.debug_mode == 10 ? dbg_ts : f.debug_mode == 11 ? dbg_tc : dbg_tw) >> 6);
		else
		ws = f.debug_mode == 2 ? (float)dbg_cand : f.debug_mode == 3 ? (float)dbg_chunks : f.debug_mode == 4 ? (float)dbg_wsteps
		   : f.debug_mode == 5 ? wave_sum((float)dbg_hits) : f.debug_mode == 7 ? (float)wave_max_i(dbg_steps) : wave_sum((float)dbg_steps);
	}
	else
#endif
	ws *= f.dL * f.dL;
	if (lane == 0) s_red[wave] = ws;
	__syncthreads();
	if (tid == 0) f.tile_scores[vt] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---------------------------------------------------------------------------------------------------------
// k_fisher_tile_v4: k_fisher_tile_v3<16, 4, MK> with ROLLING HALVES.  In k_fisher_tile_v3 the wave re-converges after every chunk
// of 64 candidates, so a chunk costs as many walk iterations as its busiest pixel-lane has candidates: 44 % of the lane slots do
// work (profiles/r03_m_loopstats.txt).  Here the 64 record slots / 64 mask bits are two halves of 32: the low word of a pixel's
// mask is the OLDER half, the high word the younger one; a lane that has walked its bits of the older half goes straight on to the
// younger half's (ffs over the 64 bits gives exactly that order), and the wave only re-converges when NO lane has a bit of the older
// half left -- then the halves swap roles (mask >>= 32, the slot base flips: slot = bit ^ base) and the freed 32 slots take the next
// 32 candidates.  A lane can thus run up to a whole half ahead of the slowest one at no cost per iteration but one v_xor (round 3's
// window kernel, k_fisher_tile_v3w, paid 8 vector instructions and two scalar branches per iteration for the same freedom).
// The set-up of a half uses all 64 lanes: lane = 32 h + c parks its part of candidate c's record (h = 0: three float4, h = 1: two),
// rasterises rows 2h, 2h+1 of the candidate's footprint into a 32-bit word, and a 32 x 32 bit transpose inside each half-wave (five
// DPP / permlane16 rounds on 32-bit words; the 64 x 64 one takes six on 64-bit words) hands every pixel-lane its 32 candidate bits.
// Pairs, their order per pixel and their arithmetic are k_fisher_tile_v3's: the scores are bit-identical.
// Fixed key segments only (the keys carry the strip bits; 80-byte records at stride 5).
template <int SFT>
__device__ __forceinline__ uint32_t fr_transpose_round32(uint32_t x, int lane)
{
	constexpr uint32_t m = SFT == 16 ? 0x0000FFFFu : SFT == 8 ? 0x00FF00FFu : SFT == 4 ? 0x0F0F0F0Fu : SFT == 2 ? 0x33333333u : 0x55555555u;
	uint32_t other;
	if constexpr (SFT == 1) other = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, false);          // quad_perm [1,0,3,2]
	else if constexpr (SFT == 2) other = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
	else if constexpr (SFT == 4) other = (uint32_t)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)x, 0x1B, 0xf, 0xf, false), 0x141, 0xf, 0xf, false);   // (i ^ 3) ^ 7 = i ^ 4
	else if constexpr (SFT == 8) other = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xf, 0xf, false);    // row_ror:8
	else
	{
		const auto t = __builtin_amdgcn_permlane16_swap(x, x, false, false);     // [0]: the lower row's word on both lanes of a pair, [1]: the upper's
		other = (lane & 16) ? t[0] : t[1];
	}
	return (lane & SFT) ? ((x & ~m) | ((other & ~m) >> SFT)) : ((x & m) | ((other & m) << SFT));
}
// 32 x 32 bit-matrix transpose inside each half of the wave: lane 32 h + i holds row i in, row i of the transpose out
__device__ __forceinline__ uint32_t fr_half_transpose32(uint32_t x, int lane)
{
	x = fr_transpose_round32<16>(x, lane);
	x = fr_transpose_round32<8>(x, lane);
	x = fr_transpose_round32<4>(x, lane);
	x = fr_transpose_round32<2>(x, lane);
	x = fr_transpose_round32<1>(x, lane);
	return x;
}
// Two rows (y0, y0 + 1) of a candidate's footprint over the strip's 16 columns: bit 16 r + column.  The rule of k_fisher_tile_v3's
// chunk step for records without extents (the conic's quadratic, widened by 1 % + 0.01 px: a superset of the exact pair test).
__device__ __forceinline__ uint32_t fr_footprint_rows2(const float4& a, const float4& b4, float y0, float tile_x0)
{
	const float ax = a.x, ay = a.y;
	const float acx = -2.0f * b4.x, acy = -b4.y, acz = -2.0f * b4.z;
	const float athr = -(5.541263545158426f + 0.6931471805599453f * a.w) - 0.01f;      // fr_power_threshold: alpha < 1/255 below it
	const bool quad_ok = acx > 0.f && athr <= 0.f;
	const float racx = __builtin_amdgcn_rcpf(acx);
	uint32_t em = 0u;
#pragma unroll
	for (unsigned r = 0; r < 2u; r++)
	{
		const float dy = ay - (y0 + (float)r);
		float lo = -1e30f, hi2 = 1e30f;
		bool any_px = fabsf(dy) <= 1e30f;
		if (quad_ok)
		{
			const float hb = acy * dy;
			const float cq = acz * dy * dy + 2.0f * athr;
			const float disc = hb * hb - acx * cq;
			any_px = any_px && (disc >= 0.f);
			const float sq = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) * 1.01f + 0.01f * acx;
			const float dlo = (-hb - sq) * racx, dhi = (-hb + sq) * racx;
			lo = ax - dhi - 0.01f; hi2 = ax - dlo + 0.01f;
		}
		const float c0f = fmaxf(ceilf(lo) - tile_x0, 0.f), c1f = fminf(floorf(hi2) - tile_x0, 15.0f);
		if (any_px && c0f <= c1f)
		{
			const unsigned c0 = (unsigned)c0f, c1 = (unsigned)c1f;
			em |= (((2u << c1) - 1u) & ~((1u << c0) - 1u)) << (16u * r);
		}
	}
	return em;
}

__global__ __launch_bounds__(FR_THREADS) __attribute__((amdgpu_waves_per_eu(4, 7)))
void k_fisher_tile_v4(FrParams p, FrFisherArgs f)
{
	__shared__ uint32_t s_q[4][FR_QCAP];
	__shared__ float4 s_ent[4][64][FR_ENT3_F4];
	__shared__ float s_red[4];
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	uint32_t tile; int v;
	fr_tile_of_block(p, tile, v);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t bx0 = tx * FR_BLOCK_X, by0 = ty * FR_BLOCK_Y + (uint32_t)wave * 4u;
	const uint32_t pxx = bx0 + (uint32_t)(lane & 15), pxy = by0 + (uint32_t)(lane >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const uint32_t n = p.tile_cnt[vt];
	const uint64_t* gk = p.keys + p.tile_off[vt];
	const float4* rec = f.recA + (size_t)v * f.ab_view;          // 80-byte records: {x, y, k3, log2 o} {-cx/2, -cy, -cz/2, cg} + 12 coefficients
	uint32_t* wq = s_q[wave];
	float4 (*ent)[FR_ENT3_F4] = s_ent[wave];
	const uint32_t ent_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&s_ent[wave][0][0];
	uint32_t ent_lds_v;
	asm volatile("v_mov_b32 %0, %1" : "=v"(ent_lds_v) : "s"(ent_lds));
	const int hc = lane & 31;                                      // set-up role: candidate of the half ...
	const int hh = lane >> 5;                                      // ... and which part of it (rows 2 hh, 2 hh + 1; float4 0-2 or 3-4)
	const float rows_y0 = (float)by0 + 2.0f * (float)hh;
	const float tile_x0 = (float)bx0;
	float T = 1.0f, Cg = 0.f, Xt = 0.f, sA = 0.f, sB = 0.f, sD = 0.f;
	bool done = !inside;
	uint32_t qh = 0, qn = 0;
#ifdef FR_LOOPSTATS
	int dbg_cand = 0, dbg_chunks = 0, dbg_steps = 0, dbg_hits = 0, dbg_wsteps = 0, dbg_cs = 0;
	long long dbg_t0 = 0, dbg_ts = 0, dbg_tc = 0, dbg_tw = 0;
#define FR_V4_STAMP(acc) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); acc += t_ - dbg_t0; dbg_t0 = t_; }
#else
#define FR_V4_STAMP(acc)
#endif
	// ---- key stream (k_fisher_tile_v3's, MK form): keep the keys that carry this wave's strip bit, append their slots to the ring
	uint32_t id1 = 0, id2 = 0;
	if ((uint32_t)lane < n) id1 = (uint32_t)FR_LDK(gk + lane);
	if (64u + lane < n) id2 = (uint32_t)FR_LDK(gk + 64 + lane);
	uint32_t base = 0;
	auto stream_fill = [&]() {
		while (qn < 32u && base < n)
		{
			const uint32_t idc = id1;
			id1 = id2;
			if (base + 128 + lane < n) id2 = (uint32_t)FR_LDK(gk + base + 128 + lane);
			const bool ov = (base + lane < n) && ((idc >> wave) & 1u);
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			if (ov) wq[(qh + qn + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) & (FR_QCAP - 1)] = idc >> 4;
			qn += (uint32_t)__popcll(om);
			base += 64;
		}
	};
	// ---- the next half's records, gathered into registers one half ahead of their parking (two lanes per candidate)
	float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga, g2 = ga, g3 = ga;
	uint32_t pm = 0;
	auto gather_next = [&]() {
		pm = qn < 32u ? qn : 32u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if ((uint32_t)hc < pm)
		{
			const float4* r = rec + (size_t)FR_ENT3_F4 * wq[(qh + hc) & (FR_QCAP - 1)];
			ga = r[0]; gb = r[1];
			if (hh == 0) g2 = r[2];
			else { g2 = r[3]; g3 = r[4]; }
		}
		qh = (qh + pm) & (FR_QCAP - 1); qn -= pm;
	};
#ifdef FR_LOOPSTATS
	dbg_t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
	stream_fill();
	gather_next();
	FR_V4_STAMP(dbg_ts)
	unsigned long long mask = 0ull;                  // low word: the older half's candidates of this pixel, high word: the younger half's
	uint32_t base0 = 0;                              // slot of bit j = j ^ base0 (0 or 32, wave-uniform)
	bool all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
	while (!all_done)
	{
		// ---- set-up: park the gathered half in the younger half's slots, rasterise two footprint rows per lane
		const uint32_t m = pm;
		uint32_t em = 0u;
		if ((uint32_t)hc < m)
		{
			float4* e = ent[(32u ^ base0) + (uint32_t)hc];
			if (hh == 0) { e[0] = ga; e[1] = gb; e[2] = g2; }
			else { e[3] = g2; e[4] = g3; }
			em = fr_footprint_rows2(ga, gb, rows_y0, tile_x0);
		}
#ifdef FR_LOOPSTATS
		dbg_chunks++; dbg_cand += (int)m;
#endif
		FR_V4_STAMP(dbg_tc)
		stream_fill();
		gather_next();
		FR_V4_STAMP(dbg_ts)
		uint32_t nm = fr_half_transpose32(em, lane);
		if (done) nm = 0u;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		mask |= (unsigned long long)nm << 32;
		if (m == 0u && __builtin_amdgcn_ballot_w64(mask != 0ull) == 0ull) break;     // nothing parked, nothing left to walk
		FR_V4_STAMP(dbg_tc)
		// ---- walk until no lane holds a bit of the older half
		// (a divergent loop: a lane leaves when it has no bit left in either half -- it cannot get one before the next set-up -- and the
		// lanes still inside leave together once none of them holds a bit of the older half: the vote only needs the active lanes)
		while (mask != 0ull)
		{
			if (__builtin_amdgcn_ballot_w64((uint32_t)mask != 0u) == 0ull) break;
			const uint32_t j = (uint32_t)(__ffsll((long long)mask) - 1);
			mask &= mask - 1ull;
			FrWalkRec3 r;
			{
				uint32_t addr;                                     // (hipcc picks the quarter-rate v_mad_u64_u32 for a 32-bit multiply-add of an xor)
				asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr) : "v"(j ^ base0), "s"((uint32_t)(FR_ENT3_F4 * 16)), "v"(ent_lds_v));
				asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\t"
				             "ds_read_b128 %3, %5 offset:48\n\tds_read_b128 %4, %5 offset:64\n\t"
				             "s_waitcnt lgkmcnt(0)"
				             : "=&v"(r.a), "=&v"(r.b4), "=&v"(r.q0), "=&v"(r.q1), "=&v"(r.q2) : "v"(addr) : "memory");
			}
#ifdef FR_LOOPSTATS
			dbg_steps++; dbg_cs++;
#endif
			const FrWalkGeom g = fr_walk_geom(r, pfx, pfy);
			const bool kill = fr_walk_update(g, g.ok, T, Cg, Xt, sA, sB, sD);
#ifdef FR_LOOPSTATS
			dbg_hits += (g.ok && !kill) ? 1 : 0;
#endif
			if (kill) { mask = 0ull; done = true; }
		}
#ifdef FR_LOOPSTATS
		dbg_wsteps += wave_max_i(dbg_cs); dbg_cs = 0;
#endif
		FR_V4_STAMP(dbg_tw)
		// ---- the older half is finished by every lane: its slots are free, the younger half becomes the older one
		mask >>= 32;
		base0 ^= 32u;
		all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
	}
	const float X = Cg + T * (p.bg[0] + p.bg[1] + p.bg[2]);
	const float dlt = X - Xt;
	float score = inside ? (sA + dlt * (dlt * sD - 2.0f * sB)) : 0.f;
	float ws = wave_sum(score);
#ifdef FR_LOOPSTATS
	if (f.debug_mode >= 2)
	{
		if (f.debug_mode == 13) ws = (float)(base < n ? base : n);
		else if (f.debug_mode == 14) ws = (float)n;
		else if (f.debug_mode >= 10) ws = (float)((f.debug_mode == 10 ? dbg_ts : f.debug_mode == 11 ? dbg_tc : dbg_tw) >> 6);
		else ws = f.debug_mode == 2 ? (float)dbg_cand : f.debug_mode == 3 ? (float)dbg_chunks : f.debug_mode == 4 ? (float)dbg_wsteps
		        : f.debug_mode == 5 ? wave_sum((float)dbg_hits) : f.debug_mode == 7 ? (float)wave_max_i(dbg_steps) : wave_sum((float)dbg_steps);
	}
	else
#endif
	ws *= f.dL * f.dL;
	if (lane == 0) s_red[wave] = ws;
	__syncthreads();
	if (tid == 0) f.tile_scores[vt] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
#undef FR_V4_STAMP

// ---------------------------------------------------------------------------------------------------------
// out_H mode of the third generation (no H_inv, constant upstream gradient, 4 columns: compute_Hessian / H_train of the
// scene map): the per-Gaussian diagonal itself is wanted, so the pixel's X = Cg_final + T_final sum(bg) has to be known
// before a pair can be squared.  Two front-to-back passes of the same wave over the same list:
//   pass 1   transmittance / colour prefix only (recA + recB, two ds_read_b128 per pair)            -> X per pixel
//   pass 2   replays the recurrences (same arithmetic, hence the same contributor set), forms
//            w = opacity G (T_i cg_i - (X - Cg_<=i) b_i) and the three mean rows A'_r . u' of the record (A-form of
//            fr_fisher_record_one), squares and adds them into wave-private LDS accumulators [column][candidate] -- in
//            DOUBLE, because ds_add_f64 costs 8.6 cycles per wave instruction on distinct addresses where ds_add_f32 costs
//            ~190 whatever the addresses (tools/lds_atomic_rate.hip; with float accumulators 3/4 of this pass was LDS time);
//   flush    per chunk, consecutive lanes on consecutive columns of one Gaussian (global float atomics into out_H).
// Against k_fisher_tile_v2 this drops the per-chunk Jacobian chains (they are in the records), the contributor lists and
// the back-to-front order.
// Neighbouring pixel-lanes walk the same splat in the same step most of the time, and ds_add_f64 serialises lanes that hit one
// address (8.6 cycles per wave instruction on 64 distinct addresses, 44 with four lanes per address: tools/lds_atomic_rate.hip) --
// a third of k_fisher_tile_v3h's time (tools/fe_ablate.py --outh).  So before the LDS adds every lane collects, from the other three
// lanes of its aligned QUAD (four consecutive pixels of a row; quad_perm DPP, no LDS), the values of those that hold the SAME
// candidate, and only the first lane of such a group issues the adds -- with the group's sum (float adds of at most four terms, then
// the double accumulator).  key < 0: the lane contributes nothing.  Returns whether this lane issues.
template <int N>
__device__ __forceinline__ bool fr_quad_combine(int key, float (&h)[N], int lane)
{
#ifdef FR_NO_QUAD_COMBINE          // (A/B builds: tools/build_variant.sh noqc -DFR_NO_QUAD_COMBINE)
	return key >= 0;
#endif
	const int q = lane & 3;
	float add[N];
#pragma unroll
	for (int k = 0; k < N; k++) add[k] = 0.f;
	bool first = key >= 0;
	// rotation r: lane q reads lane (q + r) & 3 of its quad.  (A source lane that is not executing leaves `old`: key -1, value 0.)
#define FR_QC_ROUND(CTRL, LOWER)                                                                                                   \
	{                                                                                                                              \
		const int ko = __builtin_amdgcn_update_dpp(-1, key, CTRL, 0xf, 0xf, false);                                                \
		const bool same = ko == key && key >= 0;                                                                                   \
		const float mk = same ? 1.0f : 0.0f;                                                                                       \
		_Pragma("unroll") for (int k = 0; k < N; k++)                                                                              \
		{                                                                                                                          \
			const float ho = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, h[k]), CTRL, 0xf, 0xf, false)); \
			add[k] = __builtin_fmaf(mk, ho, add[k]);                                                                               \
		}                                                                                                                          \
		first = first && !(same && (LOWER));                                                                                       \
	}
	FR_QC_ROUND(0x39, q == 3)          // quad_perm [1,2,3,0]: the source is a lower lane only for q = 3
	FR_QC_ROUND(0x4E, q >= 2)          // quad_perm [2,3,0,1]
	FR_QC_ROUND(0x93, q >= 1)          // quad_perm [3,0,1,2]
#undef FR_QC_ROUND
#pragma unroll
	for (int k = 0; k < N; k++) h[k] += add[k];
	return first;
}

struct FrPairAlpha { bool ok; float dx, dy, a_un, alpha, om1; };
__device__ __forceinline__ FrPairAlpha fr_pair_alpha(const fr_v4f& a, const fr_v4f& b4, float pfx, float pfy)
{
	FrPairAlpha g;
	g.dx = a.x - pfx; g.dy = a.y - pfy;
	float power;
	const float e = fr_scorer_exponent(b4.x, b4.y, b4.z, g.dx, g.dy, a.w, power);
	g.ok = !(power > 0.0f) && !(e < FR_E255);                 // forward.cu:347-357
	g.a_un = __builtin_amdgcn_exp2f(e);
	g.alpha = fminf(0.99f, g.a_un);
	g.om1 = 1.f - g.alpha;
	return g;
}
// forward.cu:358-366 for one pair.  Returns `kill`; con = ok && !kill.  The colour prefix is summed in double: pass 2 needs the
// SUFFIX X - Cg_<=i of deep contributors (a small difference of two numbers of order one) as exactly as the reference's
// back-to-front accumulation has it, and both passes must add the same fp32 terms.
__device__ __forceinline__ bool fr_prefix_update(const FrPairAlpha& g, float cg, float& T, double& Cg, bool& con)
{
	const float test_T = T * g.om1;
	con = g.ok && !(test_T < 0.0001f);
	const bool kill = g.ok != con;
	const float term = cg * (g.alpha * T);
	Cg = con ? Cg + (double)term : Cg;
	T = con ? test_T : T;
	return kill;
}

// The footprint of one candidate over a wave's BW x BH pixels as a bit mask (bit = row * BW + column): row by row,
// power(dx, dy) >= thr  <=>  cx dx^2 + 2 cy dy dx + (cz dy^2 + 2 thr) <= 0, an interval in dx (d = mean - pixel), widened by
// 1 % + 0.01 px so that it stays a superset of the exact pair test.
template <int BW, int BH>
__device__ __forceinline__ unsigned long long fr_footprint_mask(const float4& a, const float4& b4, float strip_lo, float tile_x0)
{
	unsigned long long emask = 0ull;
	const float ax = a.x, ay = a.y;
	const uint32_t eb = __float_as_uint(a.z);
	const float ahx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
	const float ahy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
	const float acx = -2.0f * b4.x, acy = -b4.y, acz = -2.0f * b4.z;
	const float athr = -(5.541263545158426f + 0.6931471805599453f * a.w) - 0.01f;   // -ln(255 opacity) - 0.01
	const bool quad_ok = acx > 0.f && athr <= 0.f && ahx < 1e30f;
	const float racx = __builtin_amdgcn_rcpf(acx);
#pragma unroll
	for (unsigned r = 0; r < (unsigned)BH; r++)
	{
		const float dy = ay - (strip_lo + (float)r);
		float lo = ax - ahx, hi2 = ax + ahx;                  // box fallback (unknown / degenerate conic)
		bool any_px = fabsf(dy) <= ahy;
		if (quad_ok)
		{
			const float hb = acy * dy;
			const float cq = acz * dy * dy + 2.0f * athr;
			const float disc = hb * hb - acx * cq;
			any_px = any_px && (disc >= 0.f);
			const float sq = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) * 1.01f + 0.01f * acx;
			const float dlo = (-hb - sq) * racx, dhi = (-hb + sq) * racx;
			lo = ax - dhi - 0.01f; hi2 = ax - dlo + 0.01f;
		}
		const float c0f = fmaxf(ceilf(lo) - tile_x0, 0.f), c1f = fminf(floorf(hi2) - tile_x0, (float)(BW - 1));
		if (any_px && c0f <= c1f)
		{
			const unsigned c0 = (unsigned)c0f, c1 = (unsigned)c1f;
			const unsigned long long cols = (unsigned long long)(((2u << c1) - 1u) & ~((1u << c0) - 1u));
			emask |= cols << (BW * r);
		}
	}
	return emask;
}

#ifdef FR_AB
// ---------------------------------------------------------------------------------------------------------
// k_fisher_tile_v3 with a ROLLING WINDOW of two chunks (experiment, FR_DEBUG_MODE=24; NOT the default).  In k_fisher_tile_v3 the
// wave re-converges after every chunk of 64 candidates, so every chunk costs as many walk iterations as its busiest pixel-lane has
// candidates: 7.48 M wave-level iterations per 64-view step where an unsynchronised walk would need 4.5 M (45 % of the lane slots
// do work).  Here two chunks of 48 candidates are resident in LDS (96 record slots = 32 KiB per workgroup: still five workgroups
// per CU -- a 128-slot window costs 17 % for the occupancy it loses, measured by padding k_fisher_tile_v3's LDS); a lane that has
// walked its bits of the older chunk goes straight on to the younger one, and the wave only waits until EVERY lane has left the
// older chunk, whose slots then take the next 48 candidates.  Measured (profiles/r03_f_window_walk.txt): 5.69 M iterations
// (-24 %, as a simulation on the oracle's contributor lists predicted), scores bit-identical -- and 1.07 ms against 0.99 ms: the
// move to the younger chunk costs 8 vector instructions on most iterations (some lane leaves the older chunk nearly every
// iteration), the loop has three scalar branches where the chunk-synchronous one has a single exec-masked back edge, and a
// third more chunks are set up (48 instead of 64 candidates each).  Kept for A/B runs; what it shows is that the walk's
// idle lanes cannot be bought back at this price per iteration.
#ifndef FR_WCS
#define FR_WCS 48                     // candidates per chunk of the windowed walk
#endif
__global__ __launch_bounds__(FR_THREADS) __attribute__((amdgpu_waves_per_eu(4, FR_V3_WAVES)))
void k_fisher_tile_v3w(FrParams p, FrFisherArgs f)
{
	__shared__ uint32_t s_q[4][FR_QCAP];
	__shared__ float4 s_ent[4][2 * FR_WCS][FR_ENT3_F4];
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	uint32_t tile; int v;
	fr_tile_of_block(p, tile, v);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t bx0 = tx * FR_BLOCK_X, by0 = ty * FR_BLOCK_Y + (uint32_t)wave * 4u;
	const uint32_t pxx = bx0 + (uint32_t)(lane & 15), pxy = by0 + (uint32_t)(lane >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const uint32_t n = p.tile_cnt[vt];
	const uint64_t* gk = p.keys + p.tile_off[vt];
	const float4* rec = f.recA + (size_t)v * f.ab_view;
	const float4* rq = f.recQ + (size_t)v * f.q_view;
	const size_t rsA = (size_t)f.ab_stride, rsQ = (size_t)f.q_stride;
	uint32_t* wq = s_q[wave];
	float4 (*ent)[FR_ENT3_F4] = s_ent[wave];
	constexpr uint32_t HALF = FR_WCS * FR_ENT3_F4 * 16;          // bytes of one chunk's records
	const uint32_t ent_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&s_ent[wave][0][0];   // LDS byte address
	uint32_t ent_lds_v;
	asm volatile("v_mov_b32 %0, %1" : "=v"(ent_lds_v) : "s"(ent_lds));

	const float strip_lo = (float)by0, strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)bx0, tile_x1 = tile_x0 + 15.0f;
	float T = 1.0f, Cg = 0.f, Xt = 0.f, sA = 0.f, sB = 0.f, sD = 0.f;
	bool done = !inside;
	uint32_t qh = 0, qn = 0;
#ifdef FR_LOOPSTATS
	int dbg_cand = 0, dbg_chunks = 0, dbg_steps = 0, dbg_hits = 0, dbg_wsteps = 0;
#endif

	// key stream, as in k_fisher_tile_v3
	uint32_t id1 = 0, id2 = 0;
	float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
	if ((uint32_t)lane < n) { id1 = (uint32_t)gk[lane]; r1 = rec[rsA * id1]; }
	if (64u + lane < n) id2 = (uint32_t)gk[64 + lane];
	uint32_t base = 0;
	auto stream_fill = [&]() {
		while (qn < (uint32_t)FR_WCS && base < n)
		{
			const uint32_t idc = id1; const float4 rc = r1;
			id1 = id2;
			if (base + 64 + lane < n) r1 = rec[rsA * id2];
			if (base + 128 + lane < n) id2 = (uint32_t)gk[base + 128 + lane];
			const uint32_t eb = __float_as_uint(rc.z);
			const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
			const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
			const bool ov = (base + lane < n) && hx >= 0.f && (rc.y + hy >= strip_lo) && (rc.y - hy <= strip_hi)
			                && (rc.x + hx >= tile_x0) && (rc.x - hx <= tile_x1);
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			if (ov) wq[(qh + qn + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) & (FR_QCAP - 1)] = idc;
			qn += (uint32_t)__popcll(om);
			base += 64;
		}
	};
	// the next chunk's records, gathered into registers one chunk ahead of their parking
	float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pq0 = pa, pq1 = pa, pq2 = pa;
	float pk3 = 0.f;
	uint32_t pm = 0;
	auto gather_next = [&]() {
		pm = qn < (uint32_t)FR_WCS ? qn : (uint32_t)FR_WCS;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if ((uint32_t)lane < pm)
		{
			const uint32_t id = wq[(qh + lane) & (FR_QCAP - 1)];
			pa = rec[rsA * id]; pb = rec[rsA * id + 1];
			pq0 = rq[rsQ * id]; pq1 = rq[rsQ * id + 1]; pq2 = rq[rsQ * id + 2];
			pk3 = ((const float*)(rq + rsQ * id + 3))[0];
		}
		qh = (qh + pm) & (FR_QCAP - 1); qn -= pm;
	};
	stream_fill();
	gather_next();

	unsigned long long cur = 0ull, nxt = 0ull;     // the lane's candidates in the chunk it is walking / in the chunk after it
	uint32_t cbase = ent_lds_v;                      // LDS byte address of the chunk `cur` refers to
	uint32_t oldest = 0, resident = 0;               // wave-uniform: half holding the older resident chunk, resident chunks (0..2)
	bool all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
	while (!all_done)
	{
		// ---- park the gathered chunk(s) into the free half / halves
		while (resident < 2u && pm != 0u)
		{
			const uint32_t h = (oldest + resident) & 1u;
			const uint32_t m = pm;
#ifdef FR_LOOPSTATS
			dbg_chunks++; dbg_cand += (int)m;
#endif
			unsigned long long emask = 0ull;
			if ((uint32_t)lane < m)
			{
				const float4 a = pa, b4 = pb;
				float4* e = ent[h * FR_WCS + (uint32_t)lane];
				e[0] = make_float4(a.x, a.y, pk3, a.w);               // k3 in the place of the footprint extents (only needed here)
				e[1] = b4; e[2] = pq0; e[3] = pq1; e[4] = pq2;
				emask = fr_footprint_mask<16, 4>(a, b4, strip_lo, tile_x0);
			}
			stream_fill();
			gather_next();
			unsigned long long mask = fr_wave_transpose64(emask, lane);
			if (done) mask = 0ull;
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			if (resident == 0u) { cur = mask; cbase = ent_lds_v + h * HALF; }     // nothing was resident: every lane starts on this chunk
			else nxt = mask;
			resident++;
		}
		if (resident == 0u) break;
		const uint32_t ybase = ent_lds + (oldest ^ 1u) * HALF;                   // the younger chunk's
		// ---- walk until every lane has left the older chunk.  The votes are kept as 64-bit scalar masks (v_cmp writes them, s_and /
		// s_andn2 combine them): `older` = the lanes still on the older chunk (all of them at the start of a round).
		unsigned long long older = ~0ull;
		while (true)
		{
			const unsigned long long c0 = __builtin_amdgcn_uicmpl(cur, 0ull, 32 /* eq */);
			const unsigned long long nn = __builtin_amdgcn_uicmpl(nxt, 0ull, 33 /* ne */);
			// a lane that has finished the older chunk moves on to the younger one (nxt != 0 only while the lane is on the older chunk)
			const unsigned long long swm = c0 & nn;
			if (swm != 0ull)
			{
				if (__builtin_amdgcn_inverse_ballot_w64(swm)) { cur = nxt; nxt = 0ull; cbase = ybase; }
				older &= ~swm;
			}
			const unsigned long long livem = ~c0 | swm;
			if ((livem & older) == 0ull) break;
#ifdef FR_LOOPSTATS
			dbg_wsteps++;
#endif
			if (__builtin_amdgcn_inverse_ballot_w64(livem))
			{
				const int j = __builtin_ctzll(cur);                  // (cur != 0 on every live lane)
				cur &= cur - 1ull;
				FrWalkRec3 r;
				{
					const uint32_t addr = cbase + (uint32_t)j * (FR_ENT3_F4 * 16);
					asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\t"
					             "ds_read_b128 %3, %5 offset:48\n\tds_read_b128 %4, %5 offset:64\n\t"
					             "s_waitcnt lgkmcnt(0)"
					             : "=&v"(r.a), "=&v"(r.b4), "=&v"(r.q0), "=&v"(r.q1), "=&v"(r.q2) : "v"(addr) : "memory");
				}
#ifdef FR_LOOPSTATS
				dbg_steps++;
#endif
				const FrWalkGeom g = fr_walk_geom(r, pfx, pfy);
				const bool kill = fr_walk_update(g, g.ok, T, Cg, Xt, sA, sB, sD);
#ifdef FR_LOOPSTATS
				dbg_hits += (g.ok && !kill) ? 1 : 0;
#endif
				if (kill) { cur = 0ull; nxt = 0ull; done = true; }
			}
		}
		// ---- the older chunk is finished by every lane: its half is free, the younger chunk becomes the older one
		cbase = ybase;                                   // (lanes that had not moved on hold cur == 0 and nxt == 0 here)
		oldest ^= 1u; resident--;
		all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
	}
	const float X = Cg + T * (p.bg[0] + p.bg[1] + p.bg[2]);
	const float dlt = X - Xt;
	float score = inside ? (sA + dlt * (dlt * sD - 2.0f * sB)) : 0.f;
	float ws = wave_sum(score);
#ifdef FR_LOOPSTATS
	if (f.debug_mode >= 2)
		ws = f.debug_mode == 2 ? (float)dbg_cand : f.debug_mode == 3 ? (float)dbg_chunks : f.debug_mode == 4 ? (float)dbg_wsteps
		   : f.debug_mode == 5 ? wave_sum((float)dbg_hits) : f.debug_mode == 7 ? (float)wave_max_i(dbg_steps) : wave_sum((float)dbg_steps);
	else
#endif
	ws *= f.dL * f.dL;
	// the four partial sums through the (now idle) queue memory: s_q[w][0] belongs to wave w alone until the barrier
	__builtin_amdgcn_wave_barrier();
	if (lane == 0) wq[0] = __float_as_uint(ws);
	__syncthreads();
	if (tid == 0) f.tile_scores[vt] = (__uint_as_float(s_q[0][0]) + __uint_as_float(s_q[1][0])) + (__uint_as_float(s_q[2][0]) + __uint_as_float(s_q[3][0]));
}
#endif   // FR_AB

// ---------------------------------------------------------------------------------------------------------
// Forward compositing with the scorer's walk (k_fisher_tile_v3): the wave streams the tile's keys, keeps the splats whose
// alpha footprint box meets its strip in an LDS ring, and then takes 64 candidates at a time -- one per lane: the lane parks
// the candidate's record in LDS and rasterises its footprint ellipse into a 64-bit mask over the strip's pixels; a 64 x 64 bit
// transpose hands every pixel-lane the mask of candidates that may touch it, and the lane walks ITS set bits front to back.
// k_render_forward evaluates every surviving candidate on all 64 lanes (wave-uniform loop, v_readlane broadcasts): about three
// times the steps.  The arithmetic of a (pixel, splat) pair and the order of the pairs of a pixel are those of
// k_render_forward, so colour, depth, final_T and n_contrib stay bit-identical to the oracle's.
template <int NCH>
__global__ __launch_bounds__(FR_THREADS) void k_render_forward_walk(FrParams p, const float* __restrict__ feat, int feat_view_stride,
                                                                    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                                    float* __restrict__ out_color, float* __restrict__ out_depth,
                                                                    const float* __restrict__ feat2, float* __restrict__ out_color2)
{
	constexpr int EF4 = NCH == 6 ? 5 : 3;        // float4 per parked record (4 would repeat the LDS banks after four records)
	__shared__ uint2 s_q[4][FR_QCAP];            // ring of {index, position in the tile's list}
	__shared__ float4 s_ent[4][64][EF4];
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int v = blockIdx.y;
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const size_t vP = (size_t)v * p.P;
	const uint32_t n = p.tile_cnt[vt];
	const uint64_t* gk = p.keys + p.tile_off[vt];
	const float4* splat = (const float4*)(p.splat + vP);
	const float* fv = feat + (size_t)v * feat_view_stride;
	uint2* wq = s_q[wave];
	float4 (*ent)[EF4] = s_ent[wave];

	float T = 1.0f;
	uint32_t last_contributor = 0;
	float C[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) C[c] = 0.f;
	float D = 15.0f;
	bool done = !inside;
	const float strip_lo = (float)(ty * FR_BLOCK_Y + 4u * (uint32_t)wave), strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)(tx * FR_BLOCK_X), tile_x1 = tile_x0 + 15.0f;

	uint32_t qh = 0, qn = 0;
	uint32_t id1 = 0, id2 = 0;
	float4 e1 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = e1;      // {x, y, ., .} and {., ., ., ext} of the keys at `base`
	if ((uint32_t)lane < n) { id1 = (uint32_t)gk[lane]; x1 = splat[2 * (size_t)id1]; e1 = splat[2 * (size_t)id1 + 1]; }
	if (64u + lane < n) id2 = (uint32_t)gk[64 + lane];
	uint32_t base = 0;
	bool all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
#ifdef FR_FWD_STATS
	long long st_t0 = (long long)__builtin_amdgcn_s_memtime(), st_stream = 0, st_setup = 0, st_walk = 0;
	uint32_t st_chunks = 0, st_steps = 0, st_windows = 0;
#define FR_FWD_STAMP(acc) { const long long t = (long long)__builtin_amdgcn_s_memtime(); acc += t - st_t0; st_t0 = t; }
#else
#define FR_FWD_STAMP(acc)
#endif
	while (!all_done)
	{
		// ---- stream: fill the ring up to one chunk
		while (qn < 64u && base < n)
		{
			const uint32_t idc = id1; const float4 xc = x1, ec = e1;
			id1 = id2;
			if (base + 64 + lane < n) { x1 = splat[2 * (size_t)id2]; e1 = splat[2 * (size_t)id2 + 1]; }
			if (base + 128 + lane < n) id2 = (uint32_t)gk[base + 128 + lane];
			const uint32_t eb = __float_as_uint(ec.w);
			const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
			const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
			const bool ov = (base + lane < n) && hx >= 0.f && (xc.y + hy >= strip_lo) && (xc.y - hy <= strip_hi)
			                && (xc.x + hx >= tile_x0) && (xc.x - hx <= tile_x1);
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			if (ov) wq[(qh + qn + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) & (FR_QCAP - 1)] = make_uint2(idc, base + (uint32_t)lane);
			qn += (uint32_t)__popcll(om);
			base += 64;
#ifdef FR_FWD_STATS
			st_windows++;
#endif
		}
		FR_FWD_STAMP(st_stream)
		if (qn == 0) break;
		// ---- chunk: up to 64 candidates, one per lane
		const uint32_t m = qn < 64u ? qn : 64u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		unsigned long long emask = 0ull;
		if ((uint32_t)lane < m)
		{
			const uint2 qe = wq[(qh + lane) & (FR_QCAP - 1)];
			const uint32_t id = qe.x;
			const float4 q0 = splat[2 * (size_t)id], q1 = splat[2 * (size_t)id + 1];   // {x, y, conx, cony} {conz, opacity, depth, ext}
			ent[lane][0] = q0;
			ent[lane][1] = make_float4(q1.x, q1.y, fr_power_threshold(q1.y), q1.z);
			ent[lane][2] = make_float4(fv[3 * (size_t)id], fv[3 * (size_t)id + 1], fv[3 * (size_t)id + 2], __uint_as_float(qe.y));
			if constexpr (NCH == 6) ent[lane][3] = make_float4(feat2[3 * (size_t)id], feat2[3 * (size_t)id + 1], feat2[3 * (size_t)id + 2], 0.f);
			// the scorer's footprint rasteriser wants {x, y, ext, log2 opacity} {-conx / 2, -cony, -conz / 2, .}
			const float4 a = make_float4(q0.x, q0.y, q1.w, __builtin_amdgcn_logf(q1.y));
			const float4 b4 = make_float4(-0.5f * q0.z, -q0.w, -0.5f * q1.x, 0.f);
			emask = fr_footprint_mask<16, 4>(a, b4, strip_lo, tile_x0);
		}
		qh = (qh + m) & (FR_QCAP - 1); qn -= m;
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		if (done) mask = 0ull;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		FR_FWD_STAMP(st_setup)
#ifdef FR_FWD_STATS
		st_chunks++;
		{ uint32_t pc = (uint32_t)__popcll(mask); for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)pc, o, 64); pc = t > pc ? t : pc; } st_steps += pc; }
#endif
		// ---- walk
		while (mask != 0ull)
		{
			const int j = __ffsll((long long)mask) - 1;
			mask &= mask - 1ull;
			const float4 r0 = ent[j][0], r1 = ent[j][1], r2 = ent[j][2];
			const float dx = r0.x - pfx, dy = r0.y - pfy;
			const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
			// forward.cu:338-357; NaN falls through both tests as it does there
			if (power > 0.0f || power < r1.z) continue;
			const float G = fr_expf(power);
			const float alpha = fminf(0.99f, r1.y * G);
			if (alpha < 1.0f / 255.0f) continue;
			const float test_T = T * (1 - alpha);
			if (test_T < 0.0001f) { done = true; mask = 0ull; continue; }
			C[0] = C[0] + r2.x * alpha * T; C[1] = C[1] + r2.y * alpha * T; C[2] = C[2] + r2.z * alpha * T;
			if constexpr (NCH == 6)
			{
				const float4 r3 = ent[j][3];
				C[3] = C[3] + r3.x * alpha * T; C[4] = C[4] + r3.y * alpha * T; C[5] = C[5] + r3.z * alpha * T;
			}
			D = (T > 0.5f && test_T < 0.5f) ? r1.w : D;
			T = test_T;
			last_contributor = __float_as_uint(r2.w) + 1u;
		}
		all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
		FR_FWD_STAMP(st_walk)
	}
#ifdef FR_FWD_STATS
	if (out_depth && inside && lane < 8)
	{
		// (rig: the wave's counters in place of the depth of its first eight pixels -- tools/fwd_walk_stats.py)
		const float vals[8] = { (float)(st_stream >> 6), (float)(st_setup >> 6), (float)(st_walk >> 6), (float)st_chunks, (float)st_steps, (float)st_windows, (float)n, 0.f };
		D = vals[lane];
	}
#endif
	if (inside)
	{
		const size_t HW = (size_t)p.H * p.W;
		const size_t pix = (size_t)p.W * pxy + pxx;
		final_T[v * HW + pix] = T;
		n_contrib[v * HW + pix] = last_contributor;
		if (out_color)
		{
			out_color[(v * 3 + 0) * HW + pix] = C[0] + T * p.bg[0];
			out_color[(v * 3 + 1) * HW + pix] = C[1] + T * p.bg[1];
			out_color[(v * 3 + 2) * HW + pix] = C[2] + T * p.bg[2];
		}
		if constexpr (NCH == 6)
		{
			out_color2[(v * 3 + 0) * HW + pix] = C[3] + T * p.bg[0];
			out_color2[(v * 3 + 1) * HW + pix] = C[4] + T * p.bg[1];
			out_color2[(v * 3 + 2) * HW + pix] = C[5] + T * p.bg[2];
		}
		if (out_depth) out_depth[v * HW + pix] = D;
	}
}

// The stream / chunk skeleton of k_fisher_tile_v3 for one pass of one wave; NQ = float4 of recq parked per candidate.
// body(m, id, emask) runs once per chunk of m <= 64 candidates: lane l < m holds candidate l (its index `id`, its footprint
// `emask` over the wave's pixels) and has parked its record at ent[l]; the body sets `done` for finished pixels.
template <int BW, int BH, int NQ, class Body, int EF4 = FR_ENT_F4>
__device__ __forceinline__ void fr_strip_pass(const uint64_t* __restrict__ gk, uint32_t n, const float4* __restrict__ rec, size_t sA,
                                              const float4* __restrict__ rq, size_t sQ, uint32_t* wq, float4 (*ent)[EF4],
                                              int lane, float strip_lo, float tile_x0, int kshift, int wave, bool& done, Body body)
{
	// kshift = 4: keys = depth | slot << 4 | strips (fixed key segments) -- the wave keeps the keys with its bit, no gather per key
	const float strip_hi = strip_lo + (float)(BH - 1), tile_x1 = tile_x0 + (float)(BW - 1);
	uint32_t qh = 0, qn = 0;
	uint32_t id1 = 0, id2 = 0;
	float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
	if ((uint32_t)lane < n) { id1 = (uint32_t)gk[lane]; if (!kshift) r1 = rec[sA * id1]; }
	if (64u + lane < n) id2 = (uint32_t)gk[64 + lane];
	uint32_t base = 0;
	bool all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
	while (!all_done)
	{
		while (qn < 64u && base < n)
		{
			const uint32_t idc = id1; const float4 rc = r1;
			id1 = id2;
			if (!kshift && base + 64 + lane < n) r1 = rec[sA * id2];
			if (base + 128 + lane < n) id2 = (uint32_t)gk[base + 128 + lane];
			bool ov = base + lane < n;
			if (kshift) ov = ov && ((idc >> wave) & 1u);
			else
			{
				const uint32_t eb = __float_as_uint(rc.z);
				const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
				const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
				ov = ov && hx >= 0.f && (rc.y + hy >= strip_lo) && (rc.y - hy <= strip_hi) && (rc.x + hx >= tile_x0) && (rc.x - hx <= tile_x1);
			}
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			if (ov) wq[(qh + qn + (uint32_t)__popcll(om & ((1ull << lane) - 1ull))) & (FR_QCAP - 1)] = idc >> kshift;
			qn += (uint32_t)__popcll(om);
			base += 64;
		}
		if (qn == 0) break;
		const uint32_t m = qn < 64u ? qn : 64u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		unsigned long long emask = 0ull;
		uint32_t my_id = 0;
		if ((uint32_t)lane < m)
		{
			my_id = wq[(qh + lane) & (FR_QCAP - 1)];
			const float4 a = rec[sA * my_id], b4 = rec[sA * my_id + 1];
			ent[lane][0] = a; ent[lane][1] = b4;
#pragma unroll
			for (int k = 0; k < NQ; k++) ent[lane][2 + k] = rq[sQ * my_id + k];
			emask = fr_footprint_mask<BW, BH>(a, b4, strip_lo, tile_x0);
		}
		qh = (qh + m) & (FR_QCAP - 1); qn -= m;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		body(m, my_id, emask);
		all_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
	}
}

// MODE 0: one workgroup per (tile, view) walks the whole list twice (many views: the grid fills the chip).
// MODE 1 / 2 -- FEW views (compute_Hessian of one pose, the tester's per-path-step call: 256 tiles on 256 CUs, the time set by the
// longest list): the lists are cut into segments of L keys, L = a third of the launch's mean list length (status[0] / tiles; at least
// 256), so that there are at most 4 x tiles segments in all -- the 1024 workgroups the chip holds of this kernel when one view is scored.  MODE 1 is pass 1 alone, segment after segment; it enters the tile's segments in
// a work list and leaves every pixel's state {T, colour prefix, finished} at each segment's start, plus the pixel's X.  MODE 2 is
// pass 2 of ONE segment per workgroup (grid = 4 x tiles; a workgroup beyond the list leaves -- an empty workgroup still costs its
// dispatch, ~50 ns, which is why the grid is not tiles x the longest list's segments), started from the saved state.  Pass 1 stays sequential per tile: the transmittance chain T <- T (1 - alpha) is rounded
// at every step and its 1e-4 cut decides who contributes, so no segment can know its start state before its predecessors have run --
// but pass 1 is the light pass (two ds_read_b128 and ~30 instructions per pair, no accumulators), pass 2 the heavy one.  A pixel's
// pairs go through exactly the arithmetic of MODE 0, in the same order.
#define FR_SEG_PER_TILE 4            // work-list entries per tile at most (L = a third of the mean list length)
#define FR_SEG_TILES 1024             // (view, tile) pairs up to which an out_H launch cuts its lists into segments (4 views at 256 x 256)
struct FrSegArgs {
	float* snapT;                // [4 V T][256]: T at the start of work-list entry e (negative: the pixel is finished)
	double* snapC;               // [4 V T][256]: colour prefix there
	double* X;                   // [V T][256]
	uint32_t* list;              // [0] = entries (zeroed by the launcher), [1 + e] = (view * T + tile) << 12 | segment of the tile
};
__device__ __forceinline__ uint32_t fr_seg_length(const FrParams& p)
{
	const uint32_t tiles = (uint32_t)(p.V * p.T);
	const uint32_t L = (((uint32_t)p.status[0] + 3u * tiles - 1u) / (3u * tiles) + 63u) & ~63u;     // sum over tiles of ceil(n / L) <= FR_SEG_PER_TILE tiles
	return L < 256u ? 256u : L;
}
template <int BW, int BH, int MODE = 0>
__global__ __launch_bounds__(FR_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_fisher_tile_v3h(FrParams p, FrFisherArgs f, const float4* __restrict__ recq, FrSegArgs sg)
{
	static_assert(BW * BH == 64 && 16 % BW == 0, "a wave owns 64 pixels of the tile");
	__shared__ uint32_t s_q[4][FR_QCAP];
	__shared__ float4 s_ent[4][64][FR_ENT_F4];
	// accumulators [column][candidate] in DOUBLE: ds_add_f64 takes 8.6 cycles per wave instruction on distinct addresses (3 per
	// lane on equal ones), ds_add_f32 ~190 whatever the addresses (tools/lds_atomic_rate.hip)
	__shared__ double s_acc[4][4][64];
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	uint32_t tile; int v;
	uint32_t seg = 0;                              // MODE 2: this workgroup's segment
	if constexpr (MODE == 2)
	{
		if (blockIdx.x >= sg.list[0]) return;      // (uniform) beyond the work list
		const uint32_t en = sg.list[1 + blockIdx.x];
		const uint32_t vtb = en >> 12;
		seg = en & 4095u;
		v = (int)(vtb / (uint32_t)p.T); tile = vtb % (uint32_t)p.T;
	}
	else if constexpr (MODE == 1) { v = (int)(blockIdx.x / (uint32_t)p.T); tile = blockIdx.x % (uint32_t)p.T; }
	else fr_tile_of_block(p, tile, v);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	constexpr int WPR = 16 / BW;
	const uint32_t bx0 = tx * FR_BLOCK_X + (uint32_t)(wave % WPR) * BW, by0 = ty * FR_BLOCK_Y + (uint32_t)(wave / WPR) * BH;
	const uint32_t pxx = bx0 + (uint32_t)(lane % BW), pxy = by0 + (uint32_t)(lane / BW);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const size_t vP = (size_t)v * p.P;
	const uint32_t n_all = p.tile_cnt[vt];
	const uint32_t segL = MODE == 0 ? 0u : fr_seg_length(p);
	// the keys this workgroup walks: the whole list, or one segment of it
	const uint32_t n = MODE == 2 ? min(segL, n_all - seg * segL) : n_all;
	const uint64_t* gk = p.keys + p.tile_off[vt] + (MODE == 2 ? (size_t)seg * segL : (size_t)0);
	(void)vP; (void)recq;
	const float4* rec = f.recA + (size_t)v * f.ab_view;
	const float4* rq = f.recQ + (size_t)v * f.q_view;
	const size_t sA = (size_t)f.ab_stride, sQ = (size_t)f.q_stride;
	const uint32_t* slot_idx = f.slot_idx ? f.slot_idx + (size_t)v * f.slot_view : nullptr;
	uint32_t* wq = s_q[wave];
	float4 (*ent)[FR_ENT_F4] = s_ent[wave];
	double (*acc)[64] = s_acc[wave];
	const uint32_t ent_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&s_ent[wave][0][0];
	const float strip_lo = (float)by0, tile_x0 = (float)bx0;

	// ---- pass 1: X.  Every pixel-lane walks its own candidates (the set bits of the transposed footprint masks).
	float T = 1.0f;
	double Cg = 0.0;
	bool done = !inside;
	auto pass1 = [&](const uint64_t* keys, uint32_t nk) {
	fr_strip_pass<BW, BH, 0>(keys, nk, rec, sA, rq, sQ, wq, ent, lane, strip_lo, tile_x0, f.key_shift, wave, done,
		[&](uint32_t, uint32_t, unsigned long long emask) {
			unsigned long long mask = fr_wave_transpose64(emask, lane);
			if (done) mask = 0ull;
			while (mask != 0ull)
			{
				const int j = __ffsll((long long)mask) - 1;
				mask &= mask - 1ull;
				const uint32_t addr = ent_lds + (uint32_t)j * (FR_ENT_F4 * 16);
				fr_v4f a, b4;
				asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b4) : "v"(addr) : "memory");
				const FrPairAlpha g = fr_pair_alpha(a, b4, pfx, pfy);
				bool con;
				if (fr_prefix_update(g, b4.w, T, Cg, con)) { mask = 0ull; done = true; }
			}
		});
	};
	double X;
	const size_t px_slot = vt * (size_t)FR_THREADS + (size_t)tid;               // this pixel in [V T][256] arrays
	if constexpr (MODE == 0) { pass1(gk, n); X = Cg + (double)(T * (p.bg[0] + p.bg[1] + p.bg[2])); }
	else if constexpr (MODE == 1)
	{
		// segment after segment; the state at the start of a segment is what pass 2 of that segment starts from
		__shared__ uint32_t s_base;
		const uint32_t nseg = (n + segL - 1u) / segL;                   // (<= tiles <= FR_SEG_TILES: L is at least the mean list length)
		if (tid == 0 && nseg) s_base = atomicAdd(&sg.list[0], nseg);
		__syncthreads();
		const uint32_t e0 = nseg ? s_base : 0u;
		for (uint32_t s0 = 0, sI = 0; s0 < n; s0 += segL, sI++)
		{
			if (tid == 0) sg.list[1 + e0 + sI] = ((uint32_t)vt << 12) | sI;
			const size_t o = (size_t)(e0 + sI) * (size_t)FR_THREADS + (size_t)tid;
			sg.snapT[o] = done ? -1.0f : T;
			sg.snapC[o] = Cg;
			pass1(gk + s0, min(segL, n - s0));
		}
		sg.X[px_slot] = Cg + (double)(T * (p.bg[0] + p.bg[1] + p.bg[2]));
		return;
	}
	else
	{
		X = sg.X[px_slot];
		{
			const size_t o = (size_t)blockIdx.x * (size_t)FR_THREADS + (size_t)tid;
			const float t0 = sg.snapT[o];
			T = t0 < 0.f ? 1.0f : t0;                // (a finished pixel walks nothing: its T is never read)
			done = t0 < 0.f || !inside;
			Cg = sg.snapC[o];
		}
	}

	// ---- pass 2: the squares
	if constexpr (MODE == 0) { T = 1.0f; Cg = 0.0; done = !inside; }
	const float dL2 = f.dL * f.dL;
	float* dst = f.out_H + (size_t)v * f.outH_stride;
	// one (pixel, candidate) pair: replays the recurrences and returns the pair's four squared columns (zero when it does not contribute)
	auto pair = [&](uint32_t addr, float& h0, float& h1, float& h2, float& h3, bool& con) -> bool {
		FrWalkRec r;
		asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
		             "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\t"
		             "s_waitcnt lgkmcnt(0)"
		             : "=&v"(r.a), "=&v"(r.b4), "=&v"(r.q0), "=&v"(r.q1), "=&v"(r.q2), "=&v"(r.q3) : "v"(addr) : "memory");
		const FrPairAlpha g = fr_pair_alpha(r.a, r.b4, pfx, pfy);
		const float T_i = T;
		const bool kill = fr_prefix_update(g, r.b4.w, T, Cg, con);
		{
#pragma clang fp contract(fast)
			const float bi = __builtin_amdgcn_rcpf(g.om1);
			const float dLda = T_i * r.b4.w - (float)(X - Cg) * bi;               // backward.cu:1000-1016 with the suffix written as X - prefix
			const float w = g.a_un * dLda;
			const float w2 = con ? w * w : 0.f;
			const float dx = g.dx, dy = g.dy;
			const float u0 = r.b4.x * dx + (r.b4.x * dx + r.b4.y * dy);
			const float u1 = 2.0f * (r.b4.z * dy) + r.b4.y * dx;
			const float u2 = u0 * u0, u3 = u0 * u1, u4 = u1 * u1;               // gamma(u), u = -conic d (fr_math.h: fr_mean_rows_g)
			// record: {R[0][0..3]} {R[0][4], R[1][0..2]} {R[1][3..4], R[2][0..1]} {R[2][2..4], 1/opacity^2}
			const float l0 = r.q0.x * u0 + r.q0.y * u1 + r.q0.z * u2 + r.q0.w * u3 + r.q1.x * u4;
			const float l1 = r.q1.y * u0 + r.q1.z * u1 + r.q1.w * u2 + r.q2.x * u3 + r.q2.y * u4;
			const float l2 = r.q2.z * u0 + r.q2.w * u1 + r.q3.x * u2 + r.q3.y * u3 + r.q3.z * u4;
			h0 = w2 * (l0 * l0); h1 = w2 * (l1 * l1); h2 = w2 * (l2 * l2); h3 = w2 * r.q3.w;
		}
		return kill;
	};
	fr_strip_pass<BW, BH, 4>(gk, n, rec, sA, rq, sQ, wq, ent, lane, strip_lo, tile_x0, f.key_shift, wave, done,
		[&](uint32_t m, uint32_t my_id, unsigned long long emask) {
#pragma unroll
			for (int c = 0; c < 4; c++) acc[c][lane] = 0.0;
			// every pixel-lane walks its own candidates (the set bits of the transposed footprint masks)
			unsigned long long mask = fr_wave_transpose64(emask, lane);
			if (done) mask = 0ull;
			while (mask != 0ull)
			{
				const int j = __ffsll((long long)mask) - 1;
				mask &= mask - 1ull;
				float h[4]; bool con;
				const bool kill = pair(ent_lds + (uint32_t)j * (FR_ENT_F4 * 16), h[0], h[1], h[2], h[3], con);
				const bool issue = fr_quad_combine<4>(con ? j : -1, h, lane);     // same-candidate neighbours of the quad add up first
				FR_ABL(if (f.debug_mode != 29))                 // 29: ... and without the LDS atomics
				if (issue)
				{
					atomicAdd(&acc[0][j], (double)h[0]); atomicAdd(&acc[1][j], (double)h[1]);
					atomicAdd(&acc[2][j], (double)h[2]); atomicAdd(&acc[3][j], (double)h[3]);
				}
				if (kill) { mask = 0ull; done = true; }
			}
			__builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own ds_add instructions have retired
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			// flush: consecutive lanes on consecutive columns of one Gaussian (compact records: the candidate's slot back to its index)
			const uint32_t real_id = (slot_idx && (uint32_t)lane < m) ? slot_idx[my_id] : my_id;
#pragma unroll
			for (int i = 0; i < 4; i++)
			{
				const int flat = i * 64 + lane;
				const int e = flat >> 2, c = flat & 3;
				const uint32_t id_e = (uint32_t)__builtin_amdgcn_ds_bpermute(e << 2, (int)real_id);
				const float a = ((uint32_t)e < m) ? (float)acc[c][e] * dL2 : 0.f;
				FR_ABL(if (f.debug_mode != 28))                 // 28 (tools/fe_ablate.py --outh): the walk without its global atomics
				if (a != 0.f) atomicAdd(dst + (size_t)id_e * 4 + c, a);
			}
			__builtin_amdgcn_wave_barrier();
		});
}

// ---------------------------------------------------------------------------------------------------------
// k_fisher_tile_v3h generalised: NC = 4 or 11 columns [mean xyz | opacity | scale xyz | rot rxyz] and, with IMG, a per-pixel
// upstream gradient z = dL_dpix[3] (the `im.backward(gradient=z)` probes of the POp-GS estimators, gaussian_object.py:2088-2098)
// instead of one constant: GaussianObjectSLAM.compute_Hessian / compute_H_train and the probes on the record machinery.
// With z the colour sum cg of a splat becomes cgz = z . rgb per pair, X = z . (C_final + T_final bg), and nothing else
// changes: dL_dalpha_i = T_i cgz_i - (X - Cgz_<=i) b_i.  Same two front-to-back passes, per-candidate double accumulators in LDS.
// Replaces k_fisher_tile_v2<11, false, true> (199 VGPRs, a Jacobian chain per tile instance) on these paths.
// MODE: as in k_fisher_tile_v3h (0 whole lists; few views: 1 = pass 1 segment after segment + the work list, 2 = pass 2 of one segment).
template <int NC, bool IMG, int MODE = 0>
__global__ __launch_bounds__(FR_THREADS) void k_fisher_tile_v3g(FrParams p, FrFisherArgs f, FrSegArgs sg)
{
	constexpr int NQ = NC >= 11 ? 11 : 5;          // float4 of the record behind {recA, recB}
	constexpr int EF4 = 2 + NQ;
	__shared__ uint32_t s_q[4][FR_QCAP];
	__shared__ float4 s_ent[4][64][EF4];
	__shared__ double s_acc[4][NC][64];
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	uint32_t tile; int v;
	uint32_t seg = 0;
	if constexpr (MODE == 2)
	{
		if (blockIdx.x >= sg.list[0]) return;      // (uniform) beyond the work list
		const uint32_t en = sg.list[1 + blockIdx.x];
		const uint32_t vtb = en >> 12;
		seg = en & 4095u;
		v = (int)(vtb / (uint32_t)p.T); tile = vtb % (uint32_t)p.T;
	}
	else if constexpr (MODE == 1) { v = (int)(blockIdx.x / (uint32_t)p.T); tile = blockIdx.x % (uint32_t)p.T; }
	else fr_tile_of_block(p, tile, v);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t bx0 = tx * FR_BLOCK_X, by0 = ty * FR_BLOCK_Y + (uint32_t)wave * 4u;
	const uint32_t pxx = bx0 + (uint32_t)(lane & 15), pxy = by0 + (uint32_t)(lane >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const size_t vt = (size_t)v * p.T + tile;
	const uint32_t n_all = p.tile_cnt[vt];
	const uint32_t segL = MODE == 0 ? 0u : fr_seg_length(p);
	const uint32_t n = MODE == 2 ? min(segL, n_all - seg * segL) : n_all;
	const uint64_t* gk = p.keys + p.tile_off[vt] + (MODE == 2 ? (size_t)seg * segL : (size_t)0);
	const float4* rec = f.recA + (size_t)v * f.ab_view;
	const float4* rq = f.recQ + (size_t)v * f.q_view;
	const size_t sA = (size_t)f.ab_stride, sQ = (size_t)f.q_stride;
	const uint32_t* slot_idx = f.slot_idx ? f.slot_idx + (size_t)v * f.slot_view : nullptr;
	uint32_t* wq = s_q[wave];
	float4 (*ent)[EF4] = s_ent[wave];
	double (*acc)[64] = s_acc[wave];
	const float strip_lo = (float)by0, tile_x0 = (float)bx0;
	// the pixel's upstream gradient
	float z0 = 1.f, z1 = 1.f, z2 = 1.f;
	if constexpr (IMG)
	{
		const size_t HW = (size_t)p.H * p.W, pix = (size_t)p.W * pxy + pxx;
		const float* gi = f.dL_img + (size_t)v * f.dL_stride;
		z0 = inside ? gi[pix] : 0.f; z1 = inside ? gi[HW + pix] : 0.f; z2 = inside ? gi[2 * HW + pix] : 0.f;
	}

	// ---- pass 1: X = z . (C_final + T_final bg)
	float T = 1.0f;
	double Cg = 0.0;
	bool done = !inside;
	auto pass1 = [&](uint32_t, uint32_t, unsigned long long emask) {
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		if (done) mask = 0ull;
		while (mask != 0ull)
		{
			const int j = __ffsll((long long)mask) - 1;
			mask &= mask - 1ull;
			const float4 a4 = ent[j][0], b44 = ent[j][1];
			const fr_v4f a = { a4.x, a4.y, a4.z, a4.w }, b4 = { b44.x, b44.y, b44.z, b44.w };
			const FrPairAlpha g = fr_pair_alpha(a, b4, pfx, pfy);
			float cgz = b44.w;
			if constexpr (IMG) { const float4 c = ent[j][2 + 4]; cgz = z0 * c.x + z1 * c.y + z2 * c.z; }
			bool con;
			if (fr_prefix_update(g, cgz, T, Cg, con)) { mask = 0ull; done = true; }
		}
	};
	const float bgz = IMG ? (z0 * p.bg[0] + z1 * p.bg[1] + z2 * p.bg[2]) : (p.bg[0] + p.bg[1] + p.bg[2]);
	double X;
	const size_t px_slot = vt * (size_t)FR_THREADS + (size_t)tid;
	if constexpr (MODE == 0)
	{
		fr_strip_pass<16, 4, NQ, decltype(pass1), EF4>(gk, n, rec, sA, rq, sQ, wq, ent, lane, strip_lo, tile_x0, f.key_shift, wave, done, pass1);
		X = Cg + (double)(T * bgz);
	}
	else if constexpr (MODE == 1)
	{
		__shared__ uint32_t s_base;
		const uint32_t nseg = (n + segL - 1u) / segL;
		if (tid == 0 && nseg) s_base = atomicAdd(&sg.list[0], nseg);
		__syncthreads();
		const uint32_t e0 = nseg ? s_base : 0u;
		for (uint32_t s0 = 0, sI = 0; s0 < n; s0 += segL, sI++)
		{
			if (tid == 0) sg.list[1 + e0 + sI] = ((uint32_t)vt << 12) | sI;
			const size_t o = (size_t)(e0 + sI) * (size_t)FR_THREADS + (size_t)tid;
			sg.snapT[o] = done ? -1.0f : T;
			sg.snapC[o] = Cg;
			fr_strip_pass<16, 4, NQ, decltype(pass1), EF4>(gk + s0, min(segL, n - s0), rec, sA, rq, sQ, wq, ent, lane, strip_lo, tile_x0, f.key_shift, wave, done, pass1);
		}
		sg.X[px_slot] = Cg + (double)(T * bgz);
		return;
	}
	else
	{
		X = sg.X[px_slot];
		const size_t o = (size_t)blockIdx.x * (size_t)FR_THREADS + (size_t)tid;
		const float t0 = sg.snapT[o];
		T = t0 < 0.f ? 1.0f : t0;
		done = t0 < 0.f || !inside;
		Cg = sg.snapC[o];
	}

	// ---- pass 2: the squares
	if constexpr (MODE == 0) { T = 1.0f; Cg = 0.0; done = !inside; }
	const float dL2 = IMG ? 1.0f : f.dL * f.dL;
	float* dst = f.out_H + (size_t)v * f.outH_stride;
	auto pass2 = [&](uint32_t m, uint32_t my_id, unsigned long long emask) {
#pragma unroll
		for (int c = 0; c < NC; c++) acc[c][lane] = 0.0;
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		if (done) mask = 0ull;
		while (mask != 0ull)
		{
			const int j = __ffsll((long long)mask) - 1;
			mask &= mask - 1ull;
			const float4 a4 = ent[j][0], b44 = ent[j][1];
			const fr_v4f a = { a4.x, a4.y, a4.z, a4.w }, b4 = { b44.x, b44.y, b44.z, b44.w };
			const FrPairAlpha g = fr_pair_alpha(a, b4, pfx, pfy);
			float cgz = b44.w;
			if constexpr (IMG) { const float4 c = ent[j][2 + 4]; cgz = z0 * c.x + z1 * c.y + z2 * c.z; }
			const float T_i = T;
			bool con;
			const bool kill = fr_prefix_update(g, cgz, T, Cg, con);
			if (con)
			{
				const float bi = 1.0f / g.om1;
				const float dLda = T_i * cgz - (float)(X - Cg) * bi;             // backward.cu:1000-1016 with the suffix written as X - prefix
				const float w = g.a_un * dLda;
				const float w2 = w * w;
				const float dx = g.dx, dy = g.dy;
				const float u0 = b4.x * dx + (b4.x * dx + b4.y * dy);            // u = -conic d
				const float u1 = 2.0f * (b4.z * dy) + b4.y * dx;
				const float u2 = u0 * u0, u3 = u0 * u1, u4 = u1 * u1;
				const float4 q0 = ent[j][2], q1 = ent[j][3], q2 = ent[j][4], q3 = ent[j][5];
				const float l0 = q0.x * u0 + q0.y * u1 + q0.z * u2 + q0.w * u3 + q1.x * u4;
				const float l1 = q1.y * u0 + q1.z * u1 + q1.w * u2 + q2.x * u3 + q2.y * u4;
				const float l2 = q2.z * u0 + q2.w * u1 + q3.x * u2 + q3.y * u3 + q3.z * u4;
				float h[NC];
				h[0] = w2 * (l0 * l0); h[1] = w2 * (l1 * l1); h[2] = w2 * (l2 * l2); h[3] = w2 * q3.w;
				if constexpr (NC >= 11)
				{
					const float* cf = (const float*)&ent[j][2 + 5];              // seven rows x 3 over (ux^2, ux uy, uy^2)
#pragma unroll
					for (int r = 0; r < 7; r++)
					{
						const float l = cf[3 * r] * u2 + cf[3 * r + 1] * u3 + cf[3 * r + 2] * u4;
						h[4 + r] = w2 * (l * l);
					}
				}
				// (only contributing lanes are here: the others neither offer a value nor take one -- fr_quad_combine)
				if (fr_quad_combine<NC>(j, h, lane))
				{
#pragma unroll
					for (int c = 0; c < NC; c++) atomicAdd(&acc[c][j], (double)h[c]);
				}
			}
			if (kill) { mask = 0ull; done = true; }
		}
		__builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own ds_add instructions have retired
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// flush: consecutive lanes on consecutive columns of one Gaussian (compact records: the candidate's slot back to its index)
		const uint32_t real_id = (slot_idx && (uint32_t)lane < m) ? slot_idx[my_id] : my_id;
#pragma unroll 1
		for (int i = 0; i < NC; i++)
		{
			const int flat = i * 64 + lane;
			const int e = flat / NC, c = flat - e * NC;
			const uint32_t id_e = (uint32_t)__builtin_amdgcn_ds_bpermute(e << 2, (int)real_id);
			const float a = ((uint32_t)e < m) ? (float)acc[c][e] * dL2 : 0.f;
			if (a != 0.f) atomicAdd(dst + (size_t)id_e * NC + c, a);
		}
		__builtin_amdgcn_wave_barrier();
	};
	fr_strip_pass<16, 4, NQ, decltype(pass2), EF4>(gk, n, rec, sA, rq, sQ, wq, ent, lane, strip_lo, tile_x0, f.key_shift, wave, done, pass2);
}

// fr_fisher_cfg.poses_are_c2w: the caller hands camera-to-world poses (what pose_eval receives, gaussian.py:1354-1362) and the
// library inverts them -- one thread per pose, cofactor expansion in double, rounded once to float -- instead of a dozen
// rocSOLVER / elementwise launches of torch.linalg.inv on 64 4x4 matrices.
__global__ __launch_bounds__(64) void k_invert_poses(int n, const float* __restrict__ m_in, float* __restrict__ m_out)
{
	const int i = blockIdx.x * 64 + threadIdx.x;
	if (i >= n) return;
	double m[16], inv[16];
#pragma unroll
	for (int k = 0; k < 16; k++) m[k] = (double)m_in[16 * (size_t)i + k];
	inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
	inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
	inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
	inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
	inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
	inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
	inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
	inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
	inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
	inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
	inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
	inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
	inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
	inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
	inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
	inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
	const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
	const double r = 1.0 / det;                       // (a singular pose gives inf / nan, as torch.linalg.inv raises: the scores then show it)
#pragma unroll
	for (int k = 0; k < 16; k++) m_out[16 * (size_t)i + k] = (float)(inv[k] * r);
}

__global__ __launch_bounds__(FR_THREADS) void k_reduce_scores(const float* __restrict__ tile_scores, int T,
                                                              const int* __restrict__ status, int n_groups, float* __restrict__ out_scores,
                                                              int* __restrict__ status_out)
{
	// status: one {total, overflow, max tile count, tile over its fixed capacity} per view group of the call; the caller's copy is
	// {sum, any, max, any}, and an
	// overflow of ANY group leaves every score unwritten (the caller grows the buffer and repeats the call)
	int tot = 0, ovf = 0, mx = 0, tov = 0;
	for (int g = 0; g < n_groups; g++) { tot += status[4 * g]; ovf |= status[4 * g + 1]; mx = status[4 * g + 2] > mx ? status[4 * g + 2] : mx; tov |= status[4 * g + 3]; }
	if (blockIdx.x == 0 && threadIdx.x == 0) { status_out[0] = tot; status_out[1] = ovf; status_out[2] = mx; status_out[3] = tov; }
	if (ovf || !out_scores) return;
	__shared__ double s_part[FR_THREADS];
	const int v = blockIdx.x, tid = threadIdx.x;
	double s = 0.0;
	for (int t = tid; t < T; t += FR_THREADS) s += (double)tile_scores[(size_t)v * T + t];
	s_part[tid] = s;
	__syncthreads();
	for (int o = FR_THREADS / 2; o > 0; o >>= 1)
	{
		if (tid < o) s_part[tid] += s_part[tid + o];
		__syncthreads();
	}
	if (tid == 0) out_scores[v] = (float)s_part[0];
}

// ---------------------------------------------------------------------------------------------------------
// Generic fused backward of ONE view with grad_power (renderCUDAFused, backward.cu:850-1140).
struct FrBwdArgs {
	const float* dL_dpix;        // [3][H][W]
	const float* final_T; const uint32_t* n_contrib;
	const float* colors;         // colors_precomp or rgb from SH
	int power;
	const uint8_t* only_flagged; // [T] or null: k_backward_tile handles only the flagged tiles
	int u_only;                  // 1: accumulate only (mean2D, conic, colour, opacity) -- the other leaves come from k_backward_finish
	float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolors; float* dL_dmean3D;
	float* dL_dcov3D; float* dL_dscale; float* dL_drot;
	const float* dL_dmean2D_b;   // or null: screen-space gradient of a second feature image (fr_backward_pair), added in k_backward_finish
	// second image of k_backward_lin_tile<true>
	const float* dL_dpix2; const float* colors2; float* dL_dcolors2; float* dL_dmean2D_2;
};

__device__ __forceinline__ float fr_powi(float x, int power)
{
	if (power == 1) return x;
	if (power == 2) return x * x;
	return powf(x, (float)power);
}

// HAS_SH: colours came from spherical harmonics; the per-pixel colour gradient then also moves the mean through the
// view direction (backward.cu:127-138): mean += Dm * dL_dcolor with a per-splat 3x3 Dm staged next to A.  dL_dsh itself
// is finished per Gaussian by k_finish_sh.
template <bool HAS_SR, bool HAS_SH>
__global__ __launch_bounds__(FR_THREADS) void k_backward_tile(FrParams p, FrBwdArgs b)
{
	constexpr int DMO = 15 + 18 + (HAS_SR ? 21 : 0);   // offset of Dm inside the staged coefficients
	constexpr int NA = DMO + (HAS_SH ? 9 : 0);
	constexpr int NL = 18 + (HAS_SR ? 7 : 0);   // leaves: m2(2) conic(3) col(3) op(1) mean(3) cov(6) [scale(3) rot(4)]
	__shared__ fr_f2 s_xy[FR_BWD_BATCH];
	__shared__ fr_f4 s_co[FR_BWD_BATCH];
	__shared__ float s_thr[FR_BWD_BATCH];
	__shared__ float s_rgb[3][FR_BWD_BATCH];
	__shared__ float s_A[NA][FR_BWD_BATCH];
	__shared__ uint32_t s_id[FR_BWD_BATCH];

	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	if (b.only_flagged && !b.only_flagged[tile]) return;
	const uint32_t n = p.tile_cnt[tile];
	const uint64_t* gk = p.keys + p.tile_off[tile];
	const size_t HW = (size_t)p.H * p.W;
	const size_t pix = (size_t)p.W * pxy + pxx;

	float vm[16], pm[16];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }

	FrPixState st;
	st.T_final = inside ? b.final_T[pix] : 0.f;
	st.T = st.T_final;
	st.accum0 = st.accum1 = st.accum2 = 0.f;
	st.lastc0 = st.lastc1 = st.lastc2 = 0.f;
	st.last_alpha = 0.f;
	const int last = inside ? (int)b.n_contrib[pix] : 0;
	float g0 = 0.f, g1 = 0.f, g2 = 0.f;
	if (inside) { g0 = b.dL_dpix[pix]; g1 = b.dL_dpix[HW + pix]; g2 = b.dL_dpix[2 * HW + pix]; }
	const float bg_dot = p.bg[0] * g0 + p.bg[1] * g1 + p.bg[2] * g2;
	const float ddelx_dx = (float)(0.5 * p.W), ddely_dy = (float)(0.5 * p.H);
	const int power = b.power;

	int remaining = (int)n;
	while (remaining > 0)
	{
		const int m = min(FR_BWD_BATCH, remaining);
		__syncthreads();
		if (tid < m)
		{
			const uint32_t id = (uint32_t)gk[remaining - 1 - tid];
			s_id[tid] = id;
			const float4* sp = (const float4*)(p.splat + id);
			const float4 q0 = sp[0], q1 = sp[1];
			fr_f2 xy_ = { q0.x, q0.y };
			s_xy[tid] = xy_;
			const fr_f4 co = { q0.z, q0.w, q1.x, q1.y };
			s_co[tid] = co;
			s_thr[tid] = fr_power_threshold(co.w);
			s_rgb[0][tid] = b.colors[3 * (size_t)id]; s_rgb[1][tid] = b.colors[3 * (size_t)id + 1]; s_rgb[2][tid] = b.colors[3 * (size_t)id + 2];
			fr_f3 po = { p.means3D[3 * (size_t)id], p.means3D[3 * (size_t)id + 1], p.means3D[3 * (size_t)id + 2] };
			float c3[6];
#pragma unroll
			for (int k = 0; k < 6; k++) c3[k] = p.cov3D[6 * (size_t)id + k];
			float A[3][5];
			float B[6][3];
			fr_mean_jacobian(po, c3, vm, pm, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, A, B);
#pragma unroll
			for (int r = 0; r < 3; r++)
#pragma unroll
				for (int c = 0; c < 5; c++) s_A[r * 5 + c][tid] = A[r][c];
#pragma unroll
			for (int r = 0; r < 6; r++)
#pragma unroll
				for (int c = 0; c < 3; c++) s_A[15 + r * 3 + c][tid] = B[r][c];
			if constexpr (HAS_SR)
			{
				fr_f3 sc = { p.scales[3 * (size_t)id], p.scales[3 * (size_t)id + 1], p.scales[3 * (size_t)id + 2] };
				fr_f4 q = { p.rots[4 * (size_t)id], p.rots[4 * (size_t)id + 1], p.rots[4 * (size_t)id + 2], p.rots[4 * (size_t)id + 3] };
				float Cm[7][3];
				fr_scale_rot_jacobian(sc, p.mod, q, B, Cm);
#pragma unroll
				for (int r = 0; r < 7; r++)
#pragma unroll
					for (int c = 0; c < 3; c++) s_A[33 + r * 3 + c][tid] = Cm[r][c];
			}
			if constexpr (HAS_SH)
			{
				// backward.cu:1067: the fused kernel offsets the SH pointer by M*id FLOATS (not vec3s); reproduced as is
				fr_f3 cp = { p.campos[0], p.campos[1], p.campos[2] };
				float coef[16];
				float Dm[3][3];
				fr_sh_backward_jacobian(p.D, po, cp, p.shs + (size_t)p.M * id, p.clamped + 3 * (size_t)id, coef, Dm);
#pragma unroll
				for (int r = 0; r < 3; r++)
#pragma unroll
					for (int c = 0; c < 3; c++) s_A[DMO + r * 3 + c][tid] = Dm[r][c];
			}
		}
		__syncthreads();
		for (int j = 0; j < m; j++)
		{
			const int k = remaining - 1 - j;
			bool act = inside && (k < last);
			if (!__any(act)) continue;
			const fr_f2 xy = s_xy[j];
			const fr_f4 co = s_co[j];
			float dx, dy, G = 0.f, alpha = 0.f;
			act = act && fr_pair_alpha(xy.x, xy.y, pfx, pfy, co.x, co.y, co.z, co.w, s_thr[j], dx, dy, G, alpha);
			if (!__any(act)) continue;
			float leaf[NL];
#pragma unroll
			for (int c = 0; c < NL; c++) leaf[c] = 0.f;
			if (act)
			{
				float m2x, m2y, qx, qy, qw, wcol, gop;
				fr_pair_backward(st, alpha, G, dx, dy, co.x, co.y, co.z, co.w,
				                 s_rgb[0][j], s_rgb[1][j], s_rgb[2][j], g0, g1, g2, bg_dot, ddelx_dx, ddely_dy,
				                 m2x, m2y, qx, qy, qw, wcol, gop);
				leaf[0] = fr_powi(m2x, power); leaf[1] = fr_powi(m2y, power);
				leaf[2] = fr_powi(qx, power); leaf[3] = fr_powi(qy, power); leaf[4] = fr_powi(qw, power);
				leaf[5] = fr_powi(wcol * g0, power); leaf[6] = fr_powi(wcol * g1, power); leaf[7] = fr_powi(wcol * g2, power);
				leaf[8] = fr_powi(gop, power);
#pragma unroll
				for (int r = 0; r < 3; r++)
				{
					float lm = s_A[r * 5 + 0][j] * m2x + s_A[r * 5 + 1][j] * m2y + s_A[r * 5 + 2][j] * qx
					         + s_A[r * 5 + 3][j] * qy + s_A[r * 5 + 4][j] * qw;
					if constexpr (HAS_SH)
						lm += s_A[DMO + r * 3 + 0][j] * (wcol * g0) + s_A[DMO + r * 3 + 1][j] * (wcol * g1) + s_A[DMO + r * 3 + 2][j] * (wcol * g2);
					leaf[9 + r] = fr_powi(lm, power);
				}
#pragma unroll
				for (int r = 0; r < 6; r++)
					leaf[12 + r] = fr_powi(s_A[15 + r * 3 + 0][j] * qx + s_A[15 + r * 3 + 1][j] * qy + s_A[15 + r * 3 + 2][j] * qw, power);
				if constexpr (HAS_SR)
				{
#pragma unroll
					for (int r = 0; r < 7; r++)
						leaf[18 + r] = fr_powi(s_A[33 + r * 3 + 0][j] * qx + s_A[33 + r * 3 + 1][j] * qy + s_A[33 + r * 3 + 2][j] * qw, power);
				}
			}
			float mine = 0.f;
#pragma unroll
			for (int c = 0; c < NL; c++)
			{
				const float s = wave_sum(leaf[c]);
				if (lane == c) mine = s;
			}
			if (lane < (b.u_only ? 9 : NL) && mine != 0.f)
			{
				const size_t id = s_id[j];
				float* dst;
				if (lane < 2) dst = b.dL_dmean2D + 3 * id + lane;
				else if (lane < 5) dst = b.dL_dconic + 4 * id + (lane == 4 ? 3 : lane - 2);
				else if (lane < 8) dst = b.dL_dcolors + 3 * id + (lane - 5);
				else if (lane < 9) dst = b.dL_dopacity + id;
				else if (lane < 12) dst = b.dL_dmean3D + 3 * id + (lane - 9);
				else if (lane < 18) dst = b.dL_dcov3D + 6 * id + (lane - 12);
				else if (lane < 21) dst = b.dL_dscale + 3 * id + (lane - 18);
				else dst = b.dL_drot + 4 * id + (lane - 21);
				atomicAdd(dst, mine);
			}
		}
		remaining -= m;
	}
}


// ---------------------------------------------------------------------------------------------------------
// Gradient (grad_power == 1) backward of ONE view, second generation.
//
// For power 1 every leaf is J * (sum over pixels of u), so the tile kernel only has to sum the per-pixel vector
// u = (dL_dmean2D.xy, dL_dconic.xyw, dL_dcolor.rgb, dL_dopacity) per splat, and the Jacobian chain runs once per Gaussian in
// k_backward_finish (the classic split of the upstream 3DGS backward; renderCUDAFused re-derives it per pixel).
// k_backward_lin_tile has k_fisher_tile_v2's structure: wave-private transmittance pass that records which splats touched the
// strip, then a back-to-front walk in which every lane follows its own pixel's contributors; the 9 sums of a 64-entry
// chunk live in LDS (ds_add_f32) and leave as one global atomic per entry and component.
// ---------------------------------------------------------------------------------------------------------
// PAIR: two images on the same geometry in one pass (fr_backward_pair): channels 0-2 = b.colors / b.dL_dpix, channels 3-5 =
// b.colors2 / b.dL_dpix2.  Every leaf sums both images, except the screen-space gradient (kept apart: the densifier reads the
// colour image's only, gaussian.py:207) and the colour gradients.
template <bool PAIR>
__global__ __launch_bounds__(FR_THREADS) void k_backward_lin_tile(FrParams p, FrBwdArgs b, uint8_t* __restrict__ fallback)
{
	constexpr int NCH = PAIR ? 6 : 3;
	constexpr int NACC = PAIR ? 14 : 9;          // m2x, m2y, qx, qy, qw, dcolor[NCH], dopacity, (m2x, m2y of the second image)
	__shared__ uint16_t s_wl[4][FR_WCAP];
	__shared__ double s_acc[4][NACC][64];        // double: ds_add_f64 is ~20x faster than ds_add_f32 on MI355X (tools/lds_atomic_rate.hip)
	__shared__ int s_ovf;

	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const uint32_t n = p.tile_cnt[tile];
	const uint64_t* gk = p.keys + p.tile_off[tile];
	const float4* splat = (const float4*)p.splat;
	if (n > 65535u)
	{
		if (tid == 0) fallback[tile] = 1;
		return;
	}
	if (tid == 0) s_ovf = 0;
	__syncthreads();

	// ---- pass 1 (same as k_fisher_tile_v2): transmittance, last contributor, per-wave list ----
	// wave votes as 64-bit scalar masks (v_cmp -> SGPR pair), like k_fisher_tile_v2; the arithmetic is the forward pass's
	unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside);
	float T = 1.0f;
	int last = 0;                                  // 1 + index in this wave's list of the pixel's last contributor
	int wcnt = 0;
	const float strip_lo = (float)(ty * FR_BLOCK_Y + 4u * (uint32_t)wave), strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)(tx * FR_BLOCK_X), tile_x1 = tile_x0 + 15.0f;
	float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;
	uint32_t idn = 0;
	if ((uint32_t)lane < n)
	{
		idn = (uint32_t)gk[lane];
		n0 = splat[2 * (size_t)idn]; n1 = splat[2 * (size_t)idn + 1];
	}
	uint32_t idnn = (64u + lane < n) ? (uint32_t)gk[64 + lane] : 0u;
	for (uint32_t base = 0; base < n; base += 64)
	{
		const float4 q0 = n0, q1 = n1;
		if (base + 64 + lane < n) { n0 = splat[2 * (size_t)idnn]; n1 = splat[2 * (size_t)idnn + 1]; }
		if (base + 128 + lane < n) idnn = (uint32_t)gk[base + 128 + lane];
		const uint32_t eb = __float_as_uint(q1.w);
		const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
		const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
		const bool ov = (base + lane < n) && hx >= 0.f && (q0.y + hy >= strip_lo) && (q0.y - hy <= strip_hi)
		                && (q0.x + hx >= tile_x0) && (q0.x - hx <= tile_x1);
		const float thr_l = fr_power_threshold(q1.y);
		unsigned long long todo = __ballot(ov);
		while (todo)
		{
			const int j = __builtin_ctzll(todo);
			todo &= todo - 1ull;
			const float x = fr_readlane_f(q0.x, j), y = fr_readlane_f(q0.y, j);
			const float cx = fr_readlane_f(q0.z, j), cy = fr_readlane_f(q0.w, j), cz = fr_readlane_f(q1.x, j);
			const float o = fr_readlane_f(q1.y, j), thr = fr_readlane_f(thr_l, j);
			const float dx = x - pfx, dy = y - pfy;
			const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
			const unsigned long long skip_m = __builtin_amdgcn_fcmpf(power, 0.0f, 2 /* ogt */) | __builtin_amdgcn_fcmpf(power, thr, 4 /* olt */);
			const unsigned long long pass_m = ~(skip_m | done_m);
			if (pass_m)
			{
				const float G = fr_expf_inrange(power);
				const float alpha = fminf(0.99f, o * G);
				const unsigned long long ok_m = pass_m & ~__builtin_amdgcn_fcmpf(alpha, 1.0f / 255.0f, 4 /* olt */);
				const float test_T = T * (1 - alpha);
				const unsigned long long kill_m = ok_m & __builtin_amdgcn_fcmpf(test_T, 0.0001f, 4 /* olt */);
				const unsigned long long contrib_m = ok_m & ~kill_m;
				done_m |= kill_m;
				if (contrib_m)
				{
					const bool contrib = __builtin_amdgcn_inverse_ballot_w64(contrib_m);
					T = contrib ? test_T : T;
					last = contrib ? (wcnt + 1) : last;
					if (wcnt < FR_WCAP) s_wl[wave][wcnt] = (uint16_t)(base + j);
					wcnt = __builtin_amdgcn_readfirstlane(wcnt + 1);
				}
			}
		}
		if (done_m == ~0ull) break;
	}
	if (lane == 0 && wcnt > FR_WCAP) s_ovf = 1;
	__syncthreads();
	if (s_ovf)
	{
		if (tid == 0) fallback[tile] = 1;
		return;
	}
	if (tid == 0) fallback[tile] = 0;

	// ---- pass 2: back to front, sums of u per list entry ----
	const size_t HW = (size_t)p.H * p.W;
	const size_t pix = (size_t)p.W * pxy + pxx;
	const float T_final = inside ? T : 0.f;
	float Tc = T_final, last_alpha = 0.f;
	float accum[NCH], lastc[NCH], g[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) { accum[c] = 0.f; lastc[c] = 0.f; g[c] = 0.f; }
	if (inside)
	{
		g[0] = b.dL_dpix[pix]; g[1] = b.dL_dpix[HW + pix]; g[2] = b.dL_dpix[2 * HW + pix];
		if constexpr (PAIR) { g[3] = b.dL_dpix2[pix]; g[4] = b.dL_dpix2[HW + pix]; g[5] = b.dL_dpix2[2 * HW + pix]; }
	}
	const float bg_dot_a = p.bg[0] * g[0] + p.bg[1] * g[1] + p.bg[2] * g[2];
	const float bg_dot_b = PAIR ? (p.bg[0] * g[3] + p.bg[1] * g[4] + p.bg[2] * g[5]) : 0.f;
	const float ddelx_dx = (float)(0.5 * p.W), ddely_dy = (float)(0.5 * p.H);
	const uint16_t* wl = s_wl[wave];

	for (int hi = wcnt; hi > 0; hi -= 64)
	{
		const int m = min(64, hi);
		int kk = -1;
		uint32_t my_id = 0;
		float ax = 0.f, ay = 0.f, acx = 0.f, acy = 0.f, acz = 0.f, ao = 0.f, athr = INFINITY, ahx = -1.f, ahy = -1.f;
		float col[NCH];
#pragma unroll
		for (int c = 0; c < NCH; c++) col[c] = 0.f;
#pragma unroll
		for (int c = 0; c < NACC; c++) s_acc[wave][c][lane] = 0.0;
		if (lane < m)
		{
			kk = (int)wl[hi - 1 - lane];
			my_id = (uint32_t)gk[kk];
			const float4 a0 = splat[2 * (size_t)my_id], a1 = splat[2 * (size_t)my_id + 1];
			ax = a0.x; ay = a0.y; acx = a0.z; acy = a0.w; acz = a1.x; ao = a1.y; athr = fr_power_threshold(a1.y);
			const uint32_t eb = __float_as_uint(a1.w);
			ahx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
			ahy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
			col[0] = b.colors[3 * (size_t)my_id]; col[1] = b.colors[3 * (size_t)my_id + 1]; col[2] = b.colors[3 * (size_t)my_id + 2];
			if constexpr (PAIR) { col[3] = b.colors2[3 * (size_t)my_id]; col[4] = b.colors2[3 * (size_t)my_id + 1]; col[5] = b.colors2[3 * (size_t)my_id + 2]; }
		}
		unsigned long long emask = 0ull;
		if (lane < m && ahx >= 0.f)
		{
			const bool quad_ok = acx > 0.f && athr <= 0.f && ahx < 1e30f;
#pragma unroll
			for (unsigned r = 0; r < 4; r++)
			{
				const float dy = ay - (strip_lo + (float)r);
				float lo = ax - ahx, hi2 = ax + ahx;
				bool any_px = fabsf(dy) <= ahy;
				if (quad_ok)
				{
					const float hb = acy * dy;
					const float cq = acz * dy * dy + 2.0f * athr;
					const float disc = hb * hb - acx * cq;
					any_px = any_px && (disc >= 0.f);
					const float sq = sqrtf(fmaxf(disc, 0.f)) * 1.01f + 0.01f * acx;
					const float dlo = (-hb - sq) / acx, dhi = (-hb + sq) / acx;
					lo = ax - dhi - 0.01f; hi2 = ax - dlo + 0.01f;
				}
				const float c0f = fmaxf(ceilf(lo) - tile_x0, 0.f), c1f = fminf(floorf(hi2) - tile_x0, 15.f);
				if (any_px && c0f <= c1f)
				{
					const unsigned cc0 = (unsigned)c0f, cc1 = (unsigned)c1f;
					const unsigned long long cols = (unsigned long long)(((2u << cc1) - 1u) & ~((1u << cc0) - 1u));
					emask |= cols << (16 * r);
				}
			}
		}
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		{
			// lane l owns list index hi-1-l; the pixel's contributors are the indices below `last`
			const int t = hi - last;
			mask = (t >= 64) ? 0ull : (t > 0) ? (mask & ~((1ull << t) - 1ull)) : mask;
			if (!inside) mask = 0ull;
		}
		while (fr_any(mask != 0ull))
		{
			bool has = mask != 0ull;
			const int j = has ? (__ffsll((long long)mask) - 1) : 0;
			mask &= mask - 1ull;
			const float x = fr_bperm_f(ax, j), y = fr_bperm_f(ay, j);
			const float cx = fr_bperm_f(acx, j), cy = fr_bperm_f(acy, j), cz = fr_bperm_f(acz, j);
			const float o = fr_bperm_f(ao, j), thr = fr_bperm_f(athr, j);
			float rc[NCH];
#pragma unroll
			for (int c = 0; c < NCH; c++) rc[c] = fr_bperm_f(col[c], j);
			const float dx = x - pfx, dy = y - pfy;
			const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
			const float G = fr_expf_inrange(power);
			const float alpha = fminf(0.99f, o * G);
			has = has && !(power > 0.0f) && !(power < thr) && !(alpha < 1.0f / 255.0f);
			if (has)
			{
				// backward.cu:978-1038, per image: dL_dalpha = sum_ch (c_ch - accum_ch) dL_dpix_ch T  (- T_final / (1 - alpha) bg . dL_dpix)
				Tc = Tc / (1.f - alpha);
				const float wcol = alpha * Tc;
				float da = 0.f, db = 0.f;
#pragma unroll
				for (int c = 0; c < NCH; c++)
				{
					accum[c] = last_alpha * lastc[c] + (1.f - last_alpha) * accum[c];
					lastc[c] = rc[c];
					const float t = (rc[c] - accum[c]) * g[c];
					if (c < 3) da += t; else db += t;
				}
				da *= Tc; db *= Tc;
				last_alpha = alpha;
				const float bgf = -T_final / (1.f - alpha);
				if (bg_dot_a != 0.f) da += bgf * bg_dot_a;
				if (PAIR && bg_dot_b != 0.f) db += bgf * bg_dot_b;
				const float dL_dalpha = da + db;
				const float dL_dG = o * dL_dalpha;
				const float gdx = G * dx, gdy = G * dy;
				const float dG_ddelx = -gdx * cx - gdy * cy, dG_ddely = -gdy * cz - gdx * cy;
				// screen-space gradient of the first image (of the only image when !PAIR)
				const float oa = PAIR ? o * da : dL_dG;
				atomicAdd(&s_acc[wave][0][j], (double)(oa * dG_ddelx * ddelx_dx)); atomicAdd(&s_acc[wave][1][j], (double)(oa * dG_ddely * ddely_dy));
				atomicAdd(&s_acc[wave][2][j], (double)(-0.5f * gdx * dx * dL_dG)); atomicAdd(&s_acc[wave][3][j], (double)(-0.5f * gdx * dy * dL_dG));
				atomicAdd(&s_acc[wave][4][j], (double)(-0.5f * gdy * dy * dL_dG));
#pragma unroll
				for (int c = 0; c < NCH; c++) atomicAdd(&s_acc[wave][5 + c][j], (double)(wcol * g[c]));
				atomicAdd(&s_acc[wave][5 + NCH][j], (double)(G * dL_dalpha));
				if constexpr (PAIR)
				{
					const float ob = o * db;
					atomicAdd(&s_acc[wave][12][j], (double)(ob * dG_ddelx * ddelx_dx)); atomicAdd(&s_acc[wave][13][j], (double)(ob * dG_ddely * ddely_dy));
				}
			}
		}
		__builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own ds_add instructions have retired
		if (lane < m)
		{
			const size_t id = my_id;
			float a;
			if ((a = (float)s_acc[wave][0][lane]) != 0.f) atomicAdd(b.dL_dmean2D + 3 * id, a);
			if ((a = (float)s_acc[wave][1][lane]) != 0.f) atomicAdd(b.dL_dmean2D + 3 * id + 1, a);
			if ((a = (float)s_acc[wave][2][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id, a);
			if ((a = (float)s_acc[wave][3][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id + 1, a);
			if ((a = (float)s_acc[wave][4][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id + 3, a);
			if ((a = (float)s_acc[wave][5][lane]) != 0.f) atomicAdd(b.dL_dcolors + 3 * id, a);
			if ((a = (float)s_acc[wave][6][lane]) != 0.f) atomicAdd(b.dL_dcolors + 3 * id + 1, a);
			if ((a = (float)s_acc[wave][7][lane]) != 0.f) atomicAdd(b.dL_dcolors + 3 * id + 2, a);
			if constexpr (PAIR)
			{
				if ((a = (float)s_acc[wave][8][lane]) != 0.f) atomicAdd(b.dL_dcolors2 + 3 * id, a);
				if ((a = (float)s_acc[wave][9][lane]) != 0.f) atomicAdd(b.dL_dcolors2 + 3 * id + 1, a);
				if ((a = (float)s_acc[wave][10][lane]) != 0.f) atomicAdd(b.dL_dcolors2 + 3 * id + 2, a);
				if ((a = (float)s_acc[wave][12][lane]) != 0.f) atomicAdd(b.dL_dmean2D_2 + 3 * id, a);
				if ((a = (float)s_acc[wave][13][lane]) != 0.f) atomicAdd(b.dL_dmean2D_2 + 3 * id + 1, a);
			}
			if ((a = (float)s_acc[wave][5 + NCH][lane]) != 0.f) atomicAdd(b.dL_dopacity + id, a);
		}
	}
}

// k_backward_lin_tile with the scorer's machinery, and without its first pass: final_T and the position of every pixel's last
// contributor come from the forward pass (image workspace: final_T, n_contrib -- what the reference's backward reads too), so
// the wave goes straight to the back-to-front pass.  It streams the tile's keys from the END, keeps the splats whose alpha
// footprint box meets its strip in an LDS ring (descending list position), and takes 64 of them at a time: records parked in
// LDS, footprint masks, 64 x 64 bit transpose, and every pixel-lane walks ITS set bits (ds_read_b128 of the record; the
// old kernel ran this walk wave-uniformly with a dozen ds_bpermute per step, after a wave-uniform first pass that rebuilt
// final_T and a per-wave contributor list in 30 KiB of LDS).  Per pair the arithmetic is k_backward_lin_tile's.
// No list capacity, hence no fallback tiles.
template <bool PAIR>
__global__ __launch_bounds__(FR_THREADS) void k_backward_lin_walk(FrParams p, FrBwdArgs b)
{
	constexpr int NCH = PAIR ? 6 : 3;
	constexpr int NACC = PAIR ? 14 : 9;          // m2x, m2y, qx, qy, qw, dcolor[NCH], dopacity, (m2x, m2y of the second image)
	constexpr int EF4 = PAIR ? 5 : 3;            // float4 per parked record (4 would repeat the LDS banks after four records)
	__shared__ uint2 s_q[4][FR_QCAP];            // ring of {index, position in the tile's list}
	__shared__ float4 s_ent[4][64][EF4];
	__shared__ double s_acc[4][NACC][64];        // double: ds_add_f64 is ~20x faster than ds_add_f32 on MI355X (tools/lds_atomic_rate.hip)
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const uint32_t n = p.tile_cnt[tile];
	const uint64_t* gk = p.keys + p.tile_off[tile];
	const float4* splat = (const float4*)p.splat;
	uint2* wq = s_q[wave];
	float4 (*ent)[EF4] = s_ent[wave];

	const size_t HW = (size_t)p.H * p.W;
	const size_t pix = (size_t)p.W * pxy + pxx;
	const float T_final = inside ? b.final_T[pix] : 0.f;
	const uint32_t ncontrib = inside ? b.n_contrib[pix] : 0u;     // 1 + list position of the pixel's last contributor
	float Tc = T_final, last_alpha = 0.f;
	float accum[NCH], lastc[NCH], g[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) { accum[c] = 0.f; lastc[c] = 0.f; g[c] = 0.f; }
	if (inside)
	{
		g[0] = b.dL_dpix[pix]; g[1] = b.dL_dpix[HW + pix]; g[2] = b.dL_dpix[2 * HW + pix];
		if constexpr (PAIR) { g[3] = b.dL_dpix2[pix]; g[4] = b.dL_dpix2[HW + pix]; g[5] = b.dL_dpix2[2 * HW + pix]; }
	}
	const float bg_dot_a = p.bg[0] * g[0] + p.bg[1] * g[1] + p.bg[2] * g[2];
	const float bg_dot_b = PAIR ? (p.bg[0] * g[3] + p.bg[1] * g[4] + p.bg[2] * g[5]) : 0.f;
	const float ddelx_dx = (float)(0.5 * p.W), ddely_dy = (float)(0.5 * p.H);
	const float strip_lo = (float)(ty * FR_BLOCK_Y + 4u * (uint32_t)wave), strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)(tx * FR_BLOCK_X), tile_x1 = tile_x0 + 15.0f;
	// nothing in front of the wave's deepest last contributor can be skipped, everything behind it can
	uint32_t nmax = ncontrib;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o, 64); nmax = t > nmax ? t : nmax; }
	nmax = nmax < n ? nmax : n;

	uint32_t qh = 0, qn = 0;
	int base = nmax > 0u ? (int)((nmax - 1u) & ~63u) : -64;      // blocks of 64 list positions, last one first
	while (true)
	{
		// ---- stream (backwards): fill the ring up to one chunk, deepest first
		while (qn < 64u && base >= 0)
		{
			const uint32_t pos = (uint32_t)base + (uint32_t)lane;
			bool ov = false;
			uint32_t id = 0;
			if (pos < nmax)
			{
				id = (uint32_t)gk[pos];
				const float4 q0 = splat[2 * (size_t)id], q1 = splat[2 * (size_t)id + 1];
				const uint32_t eb = __float_as_uint(q1.w);
				const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
				const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
				ov = hx >= 0.f && (q0.y + hy >= strip_lo) && (q0.y - hy <= strip_hi) && (q0.x + hx >= tile_x0) && (q0.x - hx <= tile_x1);
			}
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			// descending position: the survivors of the higher lanes go first
			if (ov) wq[(qh + qn + (uint32_t)__popcll(lane < 63 ? (om >> (lane + 1)) : 0ull)) & (FR_QCAP - 1)] = make_uint2(id, pos);
			qn += (uint32_t)__popcll(om);
			base -= 64;
		}
		if (qn == 0) break;
		// ---- chunk: up to 64 candidates, one per lane (lane 0 the deepest)
		const uint32_t m = qn < 64u ? qn : 64u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int c = 0; c < NACC; c++) s_acc[wave][c][lane] = 0.0;
		unsigned long long emask = 0ull;
		uint32_t my_id = 0;
		if ((uint32_t)lane < m)
		{
			const uint2 qe = wq[(qh + lane) & (FR_QCAP - 1)];
			my_id = qe.x;
			const float4 q0 = splat[2 * (size_t)my_id], q1 = splat[2 * (size_t)my_id + 1];   // {x, y, conx, cony} {conz, opacity, depth, ext}
			ent[lane][0] = q0;
			ent[lane][1] = make_float4(q1.x, q1.y, fr_power_threshold(q1.y), __uint_as_float(qe.y));
			ent[lane][2] = make_float4(b.colors[3 * (size_t)my_id], b.colors[3 * (size_t)my_id + 1], b.colors[3 * (size_t)my_id + 2], 0.f);
			if constexpr (PAIR) ent[lane][3] = make_float4(b.colors2[3 * (size_t)my_id], b.colors2[3 * (size_t)my_id + 1], b.colors2[3 * (size_t)my_id + 2], 0.f);
			const float4 a = make_float4(q0.x, q0.y, q1.w, __builtin_amdgcn_logf(q1.y));
			const float4 b4 = make_float4(-0.5f * q0.z, -q0.w, -0.5f * q1.x, 0.f);
			emask = fr_footprint_mask<16, 4>(a, b4, strip_lo, tile_x0);
		}
		qh = (qh + m) & (FR_QCAP - 1); qn -= m;
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		if (!inside) mask = 0ull;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// ---- walk, back to front (lane order of the chunk = descending list position)
		while (mask != 0ull)
		{
			const int j = __ffsll((long long)mask) - 1;
			mask &= mask - 1ull;
			const float4 r0 = ent[j][0], r1 = ent[j][1], r2 = ent[j][2];
			if (__float_as_uint(r1.w) >= ncontrib) continue;          // behind the pixel's last contributor (backward.cu:951-957)
			float rc[NCH];
			rc[0] = r2.x; rc[1] = r2.y; rc[2] = r2.z;
			if constexpr (PAIR) { const float4 r3 = ent[j][3]; rc[3] = r3.x; rc[4] = r3.y; rc[5] = r3.z; }
			const float cx = r0.z, cy = r0.w, cz = r1.x, o = r1.y;
			const float dx = r0.x - pfx, dy = r0.y - pfy;
			const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
			if (power > 0.0f || power < r1.z) continue;
			const float G = fr_expf_inrange(power);
			const float alpha = fminf(0.99f, o * G);
			if (alpha < 1.0f / 255.0f) continue;
			// backward.cu:978-1038, per image: dL_dalpha = sum_ch (c_ch - accum_ch) dL_dpix_ch T  (- T_final / (1 - alpha) bg . dL_dpix)
			Tc = Tc / (1.f - alpha);
			const float wcol = alpha * Tc;
			float da = 0.f, db = 0.f;
#pragma unroll
			for (int c = 0; c < NCH; c++)
			{
				accum[c] = last_alpha * lastc[c] + (1.f - last_alpha) * accum[c];
				lastc[c] = rc[c];
				const float t = (rc[c] - accum[c]) * g[c];
				if (c < 3) da += t; else db += t;
			}
			da *= Tc; db *= Tc;
			last_alpha = alpha;
			const float bgf = -T_final / (1.f - alpha);
			if (bg_dot_a != 0.f) da += bgf * bg_dot_a;
			if (PAIR && bg_dot_b != 0.f) db += bgf * bg_dot_b;
			const float dL_dalpha = da + db;
			const float dL_dG = o * dL_dalpha;
			const float gdx = G * dx, gdy = G * dy;
			const float dG_ddelx = -gdx * cx - gdy * cy, dG_ddely = -gdy * cz - gdx * cy;
			// screen-space gradient of the first image (of the only image when !PAIR)
			const float oa = PAIR ? o * da : dL_dG;
			atomicAdd(&s_acc[wave][0][j], (double)(oa * dG_ddelx * ddelx_dx)); atomicAdd(&s_acc[wave][1][j], (double)(oa * dG_ddely * ddely_dy));
			atomicAdd(&s_acc[wave][2][j], (double)(-0.5f * gdx * dx * dL_dG)); atomicAdd(&s_acc[wave][3][j], (double)(-0.5f * gdx * dy * dL_dG));
			atomicAdd(&s_acc[wave][4][j], (double)(-0.5f * gdy * dy * dL_dG));
#pragma unroll
			for (int c = 0; c < NCH; c++) atomicAdd(&s_acc[wave][5 + c][j], (double)(wcol * g[c]));
			atomicAdd(&s_acc[wave][5 + NCH][j], (double)(G * dL_dalpha));
			if constexpr (PAIR)
			{
				const float ob = o * db;
				atomicAdd(&s_acc[wave][12][j], (double)(ob * dG_ddelx * ddelx_dx)); atomicAdd(&s_acc[wave][13][j], (double)(ob * dG_ddely * ddely_dy));
			}
		}
		__builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own ds_add instructions have retired
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if ((uint32_t)lane < m)
		{
			const size_t id = my_id;
			float a;
			if ((a = (float)s_acc[wave][0][lane]) != 0.f) atomicAdd(b.dL_dmean2D + 3 * id, a);
			if ((a = (float)s_acc[wave][1][lane]) != 0.f) atomicAdd(b.dL_dmean2D + 3 * id + 1, a);
			if ((a = (float)s_acc[wave][2][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id, a);
			if ((a = (float)s_acc[wave][3][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id + 1, a);
			if ((a = (float)s_acc[wave][4][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id + 3, a);
			if ((a = (float)s_acc[wave][5][lane]) != 0.f) atomicAdd(b.dL_dcolors + 3 * id, a);
			if ((a = (float)s_acc[wave][6][lane]) != 0.f) atomicAdd(b.dL_dcolors + 3 * id + 1, a);
			if ((a = (float)s_acc[wave][7][lane]) != 0.f) atomicAdd(b.dL_dcolors + 3 * id + 2, a);
			if constexpr (PAIR)
			{
				if ((a = (float)s_acc[wave][8][lane]) != 0.f) atomicAdd(b.dL_dcolors2 + 3 * id, a);
				if ((a = (float)s_acc[wave][9][lane]) != 0.f) atomicAdd(b.dL_dcolors2 + 3 * id + 1, a);
				if ((a = (float)s_acc[wave][10][lane]) != 0.f) atomicAdd(b.dL_dcolors2 + 3 * id + 2, a);
				if ((a = (float)s_acc[wave][12][lane]) != 0.f) atomicAdd(b.dL_dmean2D_2 + 3 * id, a);
				if ((a = (float)s_acc[wave][13][lane]) != 0.f) atomicAdd(b.dL_dmean2D_2 + 3 * id + 1, a);
			}
			if ((a = (float)s_acc[wave][5 + NCH][lane]) != 0.f) atomicAdd(b.dL_dopacity + id, a);
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// ---------------------------------------------------------------------------------------------------------
// grad_power == 2 through the rasteriser API (the reference's own Fisher loop, gaussian.py:1536-1556: one view, autograd,
// GaussianRasterizer(backward_power=2)): every leaf the reference accumulates is w = opacity G dL_dalpha times a per-Gaussian row
// applied to gamma(u) = (ux, uy, ux^2, ux uy, uy^2), u = -conic d (fr_math.h).  k_backward_sq_rows builds the rows ONCE per
// visible Gaussian (k_fisher_tile_v2<25> rebuilt them per (strip, list chunk): 256 VGPRs, 544 bytes of scratch); the tile
// kernel is k_backward_lin_walk's walk -- back to front, final_T / n_contrib from the forward pass, the reference's recurrences
// and the forward's exact arithmetic (fr_expf, IEEE division), no first pass, no list capacity -- with the rows parked beside the
// candidate: per pair 54 multiply-adds give the 25 leaves, their squares go into per-candidate LDS accumulators (double).
// (A cheaper form -- sum the twelve second moments of w gamma per splat and take the quadratic forms once per Gaussian -- was
// built and dropped: a row that is nearly orthogonal to a needle's u-distribution cancels in the quadratic form, 4 of 60 000
// entries of the `general` test family missed 1e-4 by a factor of 12.)
#define FR_SQ_ROWS 56                // floats per Gaussian: mean rows 3 x 5, cov3D rows 6 x 3, scale / rotation rows 7 x 3, 1 / opacity^2, pad
#define FR_SQ_ROW_STRIDE 64          // ... at 256-byte steps: a candidate's 224 bytes are two 128-byte lines, never three
#define FR_SQ_NL 25                  // leaves: mean2D 2, conic 3, colour 3, opacity 1, mean3D 3, cov3D 6, scale 3, rotation 4
__global__ __launch_bounds__(FR_THREADS) void k_backward_sq_rows(FrParams p, float* __restrict__ rows)
{
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= p.P || p.radii[i] <= 0) return;
	float vm[16], pm[16];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	const fr_f3 po = { p.means3D[3 * (size_t)i], p.means3D[3 * (size_t)i + 1], p.means3D[3 * (size_t)i + 2] };
	float c3[6];
#pragma unroll
	for (int k = 0; k < 6; k++) c3[k] = p.cov3D[6 * (size_t)i + k];
	float Rg[3][5], Bg[6][3], Cg[7][3];
	fr_mean_rows_g(po, c3, vm, pm, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, p.W, p.H, Rg, Bg, nullptr, nullptr);
	const fr_f3 sc = { p.scales[3 * (size_t)i], p.scales[3 * (size_t)i + 1], p.scales[3 * (size_t)i + 2] };
	const fr_f4 q = { p.rots[4 * (size_t)i], p.rots[4 * (size_t)i + 1], p.rots[4 * (size_t)i + 2], p.rots[4 * (size_t)i + 3] };
	fr_scale_rot_jacobian(sc, p.mod, q, Bg, Cg);
	float r[FR_SQ_ROWS];
#pragma unroll
	for (int a = 0; a < 3; a++)
#pragma unroll
		for (int c = 0; c < 5; c++) r[a * 5 + c] = Rg[a][c];
#pragma unroll
	for (int a = 0; a < 6; a++)
#pragma unroll
		for (int c = 0; c < 3; c++) r[15 + a * 3 + c] = Bg[a][c];
#pragma unroll
	for (int a = 0; a < 7; a++)
#pragma unroll
		for (int c = 0; c < 3; c++) r[33 + a * 3 + c] = Cg[a][c];
	const float o = ((const float4*)p.splat)[2 * (size_t)i + 1].y;      // the forward's record {conz, opacity, depth, ext}: the backward ABI carries no opacities
	r[54] = 1.0f / (o * o); r[55] = 0.f;
	float4* dst = (float4*)(rows + (size_t)i * FR_SQ_ROW_STRIDE);
#pragma unroll
	for (int k = 0; k < FR_SQ_ROWS / 4; k++) dst[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
}

// One pixel's back-to-front state (backward.cu:938-1003) and the two halves of a pair's step: the test (is this candidate a
// contributor of the pixel?) and the recurrences.  k_backward_sq_walk and k_backward_sq_state run the SAME code, so a pixel's
// state at a segment boundary is bit for bit what the unsegmented walk would hold there.
struct FrSqPix { float Tc, last_alpha, accum[3], lastc[3]; };
__device__ __forceinline__ bool fr_sq_test(const float4& r0, const float4& r1, float pfx, float pfy, uint32_t ncontrib,
                                           float& dx, float& dy, float& G, float& alpha)
{
	if (__float_as_uint(r1.w) >= ncontrib) return false;          // behind the pixel's last contributor (backward.cu:951-957)
	const float cx = r0.z, cy = r0.w, cz = r1.x, o = r1.y;
	dx = r0.x - pfx; dy = r0.y - pfy;
	const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
	if (power > 0.0f || power < r1.z) return false;
	G = fr_expf_inrange(power);
	alpha = fminf(0.99f, o * G);
	return !(alpha < 1.0f / 255.0f);
}
// backward.cu:978-1003: returns dL_dalpha (before the background term)
__device__ __forceinline__ float fr_sq_update(FrSqPix& st, float alpha, const float4& r2, const float (&g)[3])
{
	st.Tc = st.Tc / (1.f - alpha);
	const float rc[3] = { r2.x, r2.y, r2.z };
	float da = 0.f;
#pragma unroll
	for (int c = 0; c < 3; c++)
	{
		st.accum[c] = st.last_alpha * st.lastc[c] + (1.f - st.last_alpha) * st.accum[c];
		st.lastc[c] = rc[c];
		da += (rc[c] - st.accum[c]) * g[c];
	}
	da *= st.Tc;
	st.last_alpha = alpha;
	return da;
}

// The same for NCH colour channels (3: one image; 6: the image pair of fr_backward_pair -- channels 3-5 belong to the second image,
// whose dL_dalpha comes back in db); the statements of fr_sq_update, channel by channel.
template <int NCH> struct FrBwdPixT { float Tc, last_alpha, accum[NCH], lastc[NCH]; };
template <int NCH>
__device__ __forceinline__ void fr_bwd_update(FrBwdPixT<NCH>& st, float alpha, const float (&rc)[NCH], const float (&g)[NCH], float& da, float& db)
{
	st.Tc = st.Tc / (1.f - alpha);
	da = 0.f; db = 0.f;
#pragma unroll
	for (int c = 0; c < NCH; c++)
	{
		st.accum[c] = st.last_alpha * st.lastc[c] + (1.f - st.last_alpha) * st.accum[c];
		st.lastc[c] = rc[c];
		const float t = (rc[c] - st.accum[c]) * g[c];
		if (c < 3) da += t; else db += t;
	}
	da *= st.Tc; db *= st.Tc;
	st.last_alpha = alpha;
}

// k_backward_sq_walk is one workgroup per tile (one view of 256 x 256: 256 workgroups on 256 CUs, the longest list sets the time), and
// a splat that covers a strip puts all 64 pixel-lanes on ONE of its LDS accumulators (ds_add_f64 on one address: 192 cycles; 25 of
// them per step) -- at every image size.  With a scratch buffer (fr_backward_ws) the backward is cut into CHUNKS of at most 64
// candidates of one strip, each an independent piece of work:
//   k_backward_sq_slots    per tile: which segments of L keys (L = a third of the mean list length, >= 256, a multiple of 64) hold a
//                          contributor of some pixel; segment k of tile t has slot floor(tile_off[t] / L) + t + k (unique, no prefix pass)
//   k_backward_sq_chunks   per (tile, segment), a wave per strip: the candidates of the segment, back to front, 64 at a time; every
//                          pixel-lane walks its candidates with the recurrences only and leaves the chunk's MAP of the pixel's state
//                          {Tc, last alpha, last colour, accumulated colour} -> state: a factor for Tc, an affine map for the colour
//                          behind, and the chunk's own last contributor (9 floats); plus the chunk's candidate list, every pixel's
//                          candidate mask, and the chunk's entry on the work list (full chunks in front of the partial ones)
//   k_backward_sq_prefix   per (tile, strip): composes the maps from the back of the list forwards and leaves, per chunk, every pixel's
//                          state in front of it
//   k_backward_sq_leaves   wave w of W takes entries w, w + W, ... of the work list (dealing the raw chunk slots left the slowest
//                          wave 3.3 times the mean; a cursor cost 12 000 atomics on one address, a fifth of the kernel's time); the
//                          candidates of a chunk go ONE AT A TIME through the whole wave: every lane tests its pixel (the arithmetic
//                          of k_backward_sq_walk; lanes without the pair carry zeros), the 25 squares are summed over the wave in
//                          registers (fr_reduce_scatter32) and lane 2 l adds leaf l's total to the Gaussian's gradient: one global
//                          atomic instruction per (strip, candidate), no LDS accumulators.
// Within a chunk the recurrences are the reference's, statement by statement; across chunks Tc is multiplied by the chunk's product
// of 1 / (1 - alpha) instead of being divided step by step, and the colour behind goes through the composed affine maps (coefficients
// in [0, 1]): a relative difference of a few 1e-7 per chunk in the state a chunk starts from.  Which pairs contribute does not depend
// on the state (backward.cu:951-975), so the contributor sets are the single pass's.
struct FrSqSegArgs {
	uint32_t* slot_map;   // [n_slots] tile + 1, or 0
	uint32_t* ctl;        // [0] full chunks listed, [1] last (partial) chunks listed, [2] set when the caller's num_rendered is below status[0]
	uint2* work;          // [2][n_chunks] the work list {chunk, tile << 2 | strip}: full chunks (64 candidates), then the last chunk of every (segment, strip)
	uint2* pixmask;       // [n_chunks][64] per pixel-lane: which of the chunk's candidates may touch it (as 2 x u32)
	uint2* ctodo;         // [n_chunks] which candidates touch any pixel of the strip
	uint32_t* cnt;        // [n_slots][4] chunks of (slot, strip)
	uint2* list;          // [n_chunks][64] {Gaussian, position in the tile's list}
	float* summ;          // [n_chunks][9][64] the chunk's map; k_backward_sq_prefix overwrites rows 0-7 with the state in front of the chunk
	uint32_t n_slots, n_chunks;
	uint32_t num_rendered;     // what the caller said fr_forward reported: the scratch is laid out for it
};
__host__ __device__ __forceinline__ uint32_t fr_sq_seg_length_of(uint32_t R, uint32_t T)
{
	const uint32_t d = 3u * T;
	const uint32_t l = (((R + d - 1u) / d) + 63u) & ~63u;
	return l < 256u ? 256u : l;
}
__device__ __forceinline__ uint32_t fr_sq_seg_length(const FrParams& p) { return fr_sq_seg_length_of((uint32_t)p.status[0], (uint32_t)p.T); }

__global__ __launch_bounds__(FR_THREADS) void k_backward_sq_slots(FrParams p, FrBwdArgs b, FrSqSegArgs sg)
{
	__shared__ uint32_t s_nmax;
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	if (tid == 0) s_nmax = 0u;
	// a scratch laid out for fewer tile instances than there are: every kernel of the chain leaves (all gradients stay zero --
	// plainly wrong rather than partly right)
	if ((uint32_t)p.status[0] > sg.num_rendered) { if (tid == 0 && blockIdx.x == 0) sg.ctl[2] = 1u; return; }
	__syncthreads();
	uint32_t nmax = inside ? b.n_contrib[(size_t)p.W * pxy + pxx] : 0u;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o, 64); nmax = t > nmax ? t : nmax; }
	if (lane == 0) atomicMax(&s_nmax, nmax);
	__syncthreads();
	const uint32_t n = p.tile_cnt[tile];
	nmax = s_nmax < n ? s_nmax : n;
	const uint32_t L = fr_sq_seg_length(p);
	const uint32_t slot0 = p.tile_off[tile] / L + tile;
	const uint32_t nseg = (nmax + L - 1u) / L;
	for (uint32_t k = tid; k < nseg; k += FR_THREADS)
		if (slot0 + k < sg.n_slots) sg.slot_map[slot0 + k] = tile + 1u;
}

// The map of one chunk on one pixel's state, built contributor by contributor (back to front); 3 + 2 NCH floats
template <int NCH> struct FrBwdMapT { float Pq, A, la, B[NCH], lc[NCH]; };      // la < 0: no contributor, the identity
template <int NCH>
__device__ __forceinline__ void fr_bwd_map_step(FrBwdMapT<NCH>& mp, float alpha, const float (&rc)[NCH])
{
	if (mp.la >= 0.f)
	{
		const float k = 1.f - mp.la;
#pragma unroll
		for (int c = 0; c < NCH; c++) mp.B[c] = mp.la * mp.lc[c] + k * mp.B[c];
		mp.A = k * mp.A;
	}
	mp.Pq = mp.Pq / (1.f - alpha);
	mp.la = alpha;
#pragma unroll
	for (int c = 0; c < NCH; c++) mp.lc[c] = rc[c];
}
template <int NCH>
__device__ __forceinline__ void fr_bwd_map_apply(FrBwdPixT<NCH>& st, const FrBwdMapT<NCH>& mp)
{
	if (!(mp.la >= 0.f)) return;
#pragma unroll
	for (int c = 0; c < NCH; c++)
	{
		const float x = st.last_alpha * st.lastc[c] + (1.f - st.last_alpha) * st.accum[c];     // what the chunk's first contributor sees behind it
		st.accum[c] = mp.B[c] + mp.A * x;                                                      // ... and its last one
		st.lastc[c] = mp.lc[c];
	}
	st.Tc = st.Tc * mp.Pq;
	st.last_alpha = mp.la;
}
// a map in the scratch buffer: rows of 64 lanes {Pq, A, B[NCH], la, lc[NCH]}; the state that replaces it {Tc, last alpha, accum[NCH], lastc[NCH]}
template <int NCH> __device__ __forceinline__ void fr_bwd_map_store(float* sm, const FrBwdMapT<NCH>& mp)
{
	sm[0 * 64] = mp.Pq; sm[1 * 64] = mp.A; sm[(2 + NCH) * 64] = mp.la;
#pragma unroll
	for (int c = 0; c < NCH; c++) { sm[(2 + c) * 64] = mp.B[c]; sm[(3 + NCH + c) * 64] = mp.lc[c]; }
}
template <int NCH> __device__ __forceinline__ void fr_bwd_map_load(const float* sm, FrBwdMapT<NCH>& mp)
{
	mp.Pq = sm[0 * 64]; mp.A = sm[1 * 64]; mp.la = sm[(2 + NCH) * 64];
#pragma unroll
	for (int c = 0; c < NCH; c++) { mp.B[c] = sm[(2 + c) * 64]; mp.lc[c] = sm[(3 + NCH + c) * 64]; }
}
template <int NCH> __device__ __forceinline__ void fr_bwd_map_identity(FrBwdMapT<NCH>& mp)
{
	mp.Pq = 1.f; mp.A = 1.f; mp.la = -1.f;
#pragma unroll
	for (int c = 0; c < NCH; c++) { mp.B[c] = 0.f; mp.lc[c] = 0.f; }
}
template <int NCH> __device__ __forceinline__ void fr_bwd_state_store(float* sm, const FrBwdPixT<NCH>& st)
{
	sm[0 * 64] = st.Tc; sm[1 * 64] = st.last_alpha;
#pragma unroll
	for (int c = 0; c < NCH; c++) { sm[(2 + c) * 64] = st.accum[c]; sm[(2 + NCH + c) * 64] = st.lastc[c]; }
}
template <int NCH> __device__ __forceinline__ void fr_bwd_state_load(const float* sm, FrBwdPixT<NCH>& st)
{
	st.Tc = sm[0 * 64]; st.last_alpha = sm[1 * 64];
#pragma unroll
	for (int c = 0; c < NCH; c++) { st.accum[c] = sm[(2 + c) * 64]; st.lastc[c] = sm[(2 + NCH + c) * 64]; }
}
template <int NCH> __device__ __forceinline__ void fr_bwd_state_init(FrBwdPixT<NCH>& st, float T_final)
{
	st.Tc = T_final; st.last_alpha = 0.f;
#pragma unroll
	for (int c = 0; c < NCH; c++) { st.accum[c] = 0.f; st.lastc[c] = 0.f; }
}
#define FR_BWD_MAP_FLOATS(NCH) (3 + 2 * (NCH))

template <int NCH>
__global__ __launch_bounds__(FR_THREADS) void k_backward_sq_chunks(FrParams p, FrBwdArgs b, FrSqSegArgs sg)
{
	__shared__ uint2 s_q[4][FR_QCAP];
	constexpr int EF4 = NCH == 6 ? 4 : 3;
	__shared__ float4 s_ent[4][64][EF4];
	if (p.status[1] || (uint32_t)p.status[0] > sg.num_rendered) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t slot = blockIdx.x;
	const uint32_t t1 = sg.slot_map[slot];
	if (t1 == 0u) return;                          // no (tile, segment) has this slot
	const uint32_t tile = t1 - 1u;
	const uint32_t L = fr_sq_seg_length(p), cmax = L >> 6;
	const uint32_t seg_k = slot - (p.tile_off[tile] / L + tile);
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const uint32_t n = p.tile_cnt[tile];
	const uint64_t* gk = p.keys + p.tile_off[tile];
	const float4* splat = (const float4*)p.splat;
	uint2* wq = s_q[wave];
	float4 (*ent)[EF4] = s_ent[wave];
	const uint32_t ncontrib = inside ? b.n_contrib[(size_t)p.W * pxy + pxx] : 0u;
	const uint32_t seg_lo = seg_k * L;
	const uint32_t seg_hi = seg_lo + L < n ? seg_lo + L : n;
	const float strip_lo = (float)(ty * FR_BLOCK_Y + 4u * (uint32_t)wave), strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)(tx * FR_BLOCK_X), tile_x1 = tile_x0 + 15.0f;
	uint32_t nmax = ncontrib;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o, 64); nmax = t > nmax ? t : nmax; }
	nmax = nmax < seg_hi ? nmax : seg_hi;

	uint32_t qh = 0, qn = 0, nchunk = 0;
	const int lowest = (int)seg_lo;
	int base = nmax > seg_lo ? (int)((nmax - 1u) & ~63u) : lowest - 64;
	while (true)
	{
		while (qn < 64u && base >= lowest)
		{
			const uint32_t pos = (uint32_t)base + (uint32_t)lane;
			bool ov = false;
			uint32_t id = 0;
			if (pos < nmax)
			{
				id = (uint32_t)gk[pos];
				const float4 q0 = splat[2 * (size_t)id], q1 = splat[2 * (size_t)id + 1];
				const uint32_t eb = __float_as_uint(q1.w);
				const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
				const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
				ov = hx >= 0.f && (q0.y + hy >= strip_lo) && (q0.y - hy <= strip_hi) && (q0.x + hx >= tile_x0) && (q0.x - hx <= tile_x1);
			}
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			if (ov) wq[(qh + qn + (uint32_t)__popcll(lane < 63 ? (om >> (lane + 1)) : 0ull)) & (FR_QCAP - 1)] = make_uint2(id, pos);
			qn += (uint32_t)__popcll(om);
			base -= 64;
		}
		if (qn == 0) break;
		const uint32_t m = qn < 64u ? qn : 64u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const size_t cid = ((size_t)slot * 4 + (size_t)wave) * cmax + nchunk;
		const bool room = nchunk < cmax && cid < sg.n_chunks;          // (always: a segment of L keys has at most L / 64 chunks per strip)
		unsigned long long emask = 0ull;
		uint2 qe = make_uint2(0u, 0u);
		if ((uint32_t)lane < m)
		{
			qe = wq[(qh + lane) & (FR_QCAP - 1)];
			const float4 q0 = splat[2 * (size_t)qe.x], q1 = splat[2 * (size_t)qe.x + 1];
			ent[lane][0] = q0;
			ent[lane][1] = make_float4(q1.x, q1.y, fr_power_threshold(q1.y), __uint_as_float(qe.y));
			ent[lane][2] = make_float4(b.colors[3 * (size_t)qe.x], b.colors[3 * (size_t)qe.x + 1], b.colors[3 * (size_t)qe.x + 2], 0.f);
			if constexpr (NCH == 6) ent[lane][3] = make_float4(b.colors2[3 * (size_t)qe.x], b.colors2[3 * (size_t)qe.x + 1], b.colors2[3 * (size_t)qe.x + 2], 0.f);
			const float4 a = make_float4(q0.x, q0.y, q1.w, __builtin_amdgcn_logf(q1.y));
			const float4 b4 = make_float4(-0.5f * q0.z, -q0.w, -0.5f * q1.x, 0.f);
			emask = fr_footprint_mask<16, 4>(a, b4, strip_lo, tile_x0);
		}
		qh = (qh + m) & (FR_QCAP - 1); qn -= m;
		const unsigned long long todo = __builtin_amdgcn_ballot_w64(emask != 0ull);
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		if (!inside) mask = 0ull;
		if (room)
		{
			sg.list[cid * 64 + lane] = qe;
			sg.pixmask[cid * 64 + lane] = make_uint2((uint32_t)mask, (uint32_t)(mask >> 32));
			if (lane == 0) sg.ctodo[cid] = make_uint2((uint32_t)todo, (uint32_t)(todo >> 32));
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		FrBwdMapT<NCH> mp;
		fr_bwd_map_identity(mp);
		while (mask != 0ull)
		{
			const int j = __ffsll((long long)mask) - 1;
			mask &= mask - 1ull;
			const float4 r0 = ent[j][0], r1 = ent[j][1], r2 = ent[j][2];
			float dx, dy, G, alpha;
			if (!fr_sq_test(r0, r1, pfx, pfy, ncontrib, dx, dy, G, alpha)) continue;
			float rc[NCH];
			rc[0] = r2.x; rc[1] = r2.y; rc[2] = r2.z;
			if constexpr (NCH == 6) { const float4 r3 = ent[j][3]; rc[3] = r3.x; rc[4] = r3.y; rc[5] = r3.z; }
			fr_bwd_map_step(mp, alpha, rc);
		}
		if (room)
		{
			fr_bwd_map_store(sg.summ + cid * (FR_BWD_MAP_FLOATS(NCH) * 64) + lane, mp);
			nchunk++;
		}
		__builtin_amdgcn_wave_barrier();
	}
	if (lane == 0)
	{
		sg.cnt[(size_t)slot * 4 + wave] = nchunk;
		// the work list of k_backward_sq_leaves: all but the last chunk of a (segment, strip) hold 64 candidates, and those go first
		if (nchunk > 0u)
		{
			const uint32_t cid0 = (slot * 4u + (uint32_t)wave) * cmax, where = tile << 2 | (uint32_t)wave;
			// one 64-bit add moves both counters: ctl[0] (low word) += the full chunks, ctl[1] (high word) += 1
			const unsigned long long old = atomicAdd((unsigned long long*)sg.ctl, (1ull << 32) | (unsigned long long)(nchunk - 1u));
			const uint32_t wh = (uint32_t)old, wl = (uint32_t)(old >> 32);
			if (wl < sg.n_chunks) sg.work[(size_t)sg.n_chunks + wl] = make_uint2(cid0 + nchunk - 1u, where);
			for (uint32_t c = 0; c + 1u < nchunk; c++) if (wh + c < sg.n_chunks) sg.work[wh + c] = make_uint2(cid0 + c, where);
		}
	}
}

template <int NCH>
__global__ __launch_bounds__(FR_THREADS) void k_backward_sq_prefix(FrParams p, FrBwdArgs b, FrSqSegArgs sg)
{
	if (p.status[1] || (uint32_t)p.status[0] > sg.num_rendered) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const uint32_t n = p.tile_cnt[tile];
	const uint32_t L = fr_sq_seg_length(p), cmax = L >> 6;
	const uint32_t slot0 = p.tile_off[tile] / L + tile;
	FrBwdPixT<NCH> st;
	fr_bwd_state_init(st, inside ? b.final_T[(size_t)p.W * pxy + pxx] : 0.f);
	// segments from the back of the list, 64 at a time: lane i looks up segment kb - i, the wave then goes through them in order
	for (int kb = (int)((n + L - 1u) / L) - 1; kb >= 0; kb -= 64)
	{
		const int my_k = kb - lane;
		uint32_t my_nc = 0u;
		if (my_k >= 0)
		{
			const uint32_t slot = slot0 + (uint32_t)my_k;
			if (slot < sg.n_slots && sg.slot_map[slot] != 0u) my_nc = sg.cnt[(size_t)slot * 4 + wave];     // 0: behind every pixel's last contributor
		}
		const int nk = kb + 1 < 64 ? kb + 1 : 64;
		for (int i = 0; i < nk; i++)
		{
			const uint32_t nc = (uint32_t)__builtin_amdgcn_readlane((int)my_nc, i);
			if (nc == 0u) continue;
			const size_t cid0 = ((size_t)(slot0 + (uint32_t)(kb - i)) * 4 + (size_t)wave) * cmax;
			for (uint32_t c0 = 0; c0 < nc; c0 += 4u)
			{
				// the maps of up to four chunks first (independent loads), then the chain
				FrBwdMapT<NCH> mp[4];
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					fr_bwd_map_identity(mp[u]);
					if (c0 + (uint32_t)u < nc) fr_bwd_map_load(sg.summ + (cid0 + c0 + u) * (FR_BWD_MAP_FLOATS(NCH) * 64) + lane, mp[u]);
				}
#pragma unroll
				for (int u = 0; u < 4; u++)
					if (c0 + (uint32_t)u < nc)
					{
						fr_bwd_state_store(sg.summ + (cid0 + c0 + u) * (FR_BWD_MAP_FLOATS(NCH) * 64) + lane, st);
						fr_bwd_map_apply(st, mp[u]);
					}
			}
		}
	}
}

__global__ __launch_bounds__(FR_THREADS) void k_backward_sq_walk(FrParams p, FrBwdArgs b, const float* __restrict__ rows)
{
	constexpr int EF4 = 3 + FR_SQ_ROWS / 4;      // 17 float4 per parked candidate (odd: sixteen consecutive records cover all LDS banks)
	__shared__ uint2 s_q[4][FR_QCAP];
	__shared__ float4 s_ent[4][64][EF4];
	__shared__ double s_acc[4][FR_SQ_NL][64];
	if (p.status[1]) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t tile = blockIdx.x;
	const uint32_t tx = tile % p.gx, ty = tile / p.gx;
	const uint32_t pxx = tx * FR_BLOCK_X + (tid & 15), pxy = ty * FR_BLOCK_Y + (tid >> 4);
	const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
	const float pfx = (float)pxx, pfy = (float)pxy;
	const uint32_t n = p.tile_cnt[tile];
	const uint64_t* gk = p.keys + p.tile_off[tile];
	const float4* splat = (const float4*)p.splat;
	uint2* wq = s_q[wave];
	float4 (*ent)[EF4] = s_ent[wave];

	const size_t HW = (size_t)p.H * p.W;
	const size_t pix = (size_t)p.W * pxy + pxx;
	const float T_final = inside ? b.final_T[pix] : 0.f;
	const uint32_t ncontrib = inside ? b.n_contrib[pix] : 0u;
	FrSqPix st;
	st.Tc = T_final; st.last_alpha = 0.f;
#pragma unroll
	for (int c = 0; c < 3; c++) { st.accum[c] = 0.f; st.lastc[c] = 0.f; }
	float g[3] = { 0.f, 0.f, 0.f };
	if (inside) { g[0] = b.dL_dpix[pix]; g[1] = b.dL_dpix[HW + pix]; g[2] = b.dL_dpix[2 * HW + pix]; }
	const float bg_dot = p.bg[0] * g[0] + p.bg[1] * g[1] + p.bg[2] * g[2];
	const float hw = (float)(0.5 * p.W), hh = (float)(0.5 * p.H);
	const float strip_lo = (float)(ty * FR_BLOCK_Y + 4u * (uint32_t)wave), strip_hi = strip_lo + 3.0f;
	const float tile_x0 = (float)(tx * FR_BLOCK_X), tile_x1 = tile_x0 + 15.0f;
	uint32_t nmax = ncontrib;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)nmax, o, 64); nmax = t > nmax ? t : nmax; }
	nmax = nmax < n ? nmax : n;

	uint32_t qh = 0, qn = 0;
	int base = nmax > 0u ? (int)((nmax - 1u) & ~63u) : -64;
	while (true)
	{
		while (qn < 64u && base >= 0)
		{
			const uint32_t pos = (uint32_t)base + (uint32_t)lane;
			bool ov = false;
			uint32_t id = 0;
			if (pos < nmax)
			{
				id = (uint32_t)gk[pos];
				const float4 q0 = splat[2 * (size_t)id], q1 = splat[2 * (size_t)id + 1];
				const uint32_t eb = __float_as_uint(q1.w);
				const float hx = __half2float(__ushort_as_half((unsigned short)(eb & 0xffffu)));
				const float hy = __half2float(__ushort_as_half((unsigned short)(eb >> 16)));
				ov = hx >= 0.f && (q0.y + hy >= strip_lo) && (q0.y - hy <= strip_hi) && (q0.x + hx >= tile_x0) && (q0.x - hx <= tile_x1);
			}
			const unsigned long long om = __builtin_amdgcn_ballot_w64(ov);
			if (ov) wq[(qh + qn + (uint32_t)__popcll(lane < 63 ? (om >> (lane + 1)) : 0ull)) & (FR_QCAP - 1)] = make_uint2(id, pos);
			qn += (uint32_t)__popcll(om);
			base -= 64;
		}
		if (qn == 0) break;
		const uint32_t m = qn < 64u ? qn : 64u;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
#pragma unroll 1
		for (int c = 0; c < FR_SQ_NL; c++) s_acc[wave][c][lane] = 0.0;
		unsigned long long emask = 0ull;
		uint32_t my_id = 0;
		if ((uint32_t)lane < m)
		{
			const uint2 qe = wq[(qh + lane) & (FR_QCAP - 1)];
			my_id = qe.x;
			const float4 q0 = splat[2 * (size_t)my_id], q1 = splat[2 * (size_t)my_id + 1];   // {x, y, conx, cony} {conz, opacity, depth, ext}
			ent[lane][0] = q0;
			ent[lane][1] = make_float4(q1.x, q1.y, fr_power_threshold(q1.y), __uint_as_float(qe.y));
			ent[lane][2] = make_float4(b.colors[3 * (size_t)my_id], b.colors[3 * (size_t)my_id + 1], b.colors[3 * (size_t)my_id + 2], 0.f);
			const float4* rsrc = (const float4*)(rows + (size_t)my_id * FR_SQ_ROW_STRIDE);
#pragma unroll
			for (int k = 0; k < FR_SQ_ROWS / 4; k++) ent[lane][3 + k] = rsrc[k];
			const float4 a = make_float4(q0.x, q0.y, q1.w, __builtin_amdgcn_logf(q1.y));
			const float4 b4 = make_float4(-0.5f * q0.z, -q0.w, -0.5f * q1.x, 0.f);
			emask = fr_footprint_mask<16, 4>(a, b4, strip_lo, tile_x0);
		}
		qh = (qh + m) & (FR_QCAP - 1); qn -= m;
		unsigned long long mask = fr_wave_transpose64(emask, lane);
		if (!inside) mask = 0ull;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		while (mask != 0ull)
		{
			const int j = __ffsll((long long)mask) - 1;
			mask &= mask - 1ull;
			const float4 r0 = ent[j][0], r1 = ent[j][1], r2 = ent[j][2];
			float dx, dy, G, alpha;
			if (!fr_sq_test(r0, r1, pfx, pfy, ncontrib, dx, dy, G, alpha)) continue;
			const float cx = r0.z, cy = r0.w, cz = r1.x, o = r1.y;
			// backward.cu:978-1038
			float da = fr_sq_update(st, alpha, r2, g);
			const float wcol = alpha * st.Tc;
			if (bg_dot != 0.f) da += (-T_final / (1.f - alpha)) * bg_dot;
			const float w = (o * da) * G;                                    // dL_dG * G
			const float ux = -(cx * dx + cy * dy), uy = -(cy * dx + cz * dy);   // u = -conic d
			const float uxx = ux * ux, uxy = ux * uy, uyy = uy * uy;
			double* acc = &s_acc[wave][0][j];
			// (scheduling fences: without them the compiler forms all the addends before the first ds_add_f64)
			{
				const float l0 = w * ux * hw, l1 = w * uy * hh;                                   // dL_dmean2D (backward.cu:1021-1024)
				const float hq = -0.5f * w;
				const float l2 = hq * dx * dx, l3 = hq * dx * dy, l4 = hq * dy * dy;               // dL_dconic x, y, w (1026-1029)
				atomicAdd(acc + 0 * 64, (double)(l0 * l0)); atomicAdd(acc + 1 * 64, (double)(l1 * l1));
				atomicAdd(acc + 2 * 64, (double)(l2 * l2)); atomicAdd(acc + 3 * 64, (double)(l3 * l3)); atomicAdd(acc + 4 * 64, (double)(l4 * l4));
			}
			__builtin_amdgcn_sched_barrier(0);
			{
#pragma unroll
				for (int c = 0; c < 3; c++) { const float l = wcol * g[c]; atomicAdd(acc + (5 + c) * 64, (double)(l * l)); }   // dL_dcolors
				const float4 rz = ent[j][3 + 13];                                                 // {Cg[6][0..2] tail ..., 1/o^2}: rows[52..55]
				const float lo = w * w * rz.z;                                                      // dL_dopacity = G dL_dalpha = w / opacity
				atomicAdd(acc + 8 * 64, (double)lo);
			}
			__builtin_amdgcn_sched_barrier(0);
			// the 16 rows: mean 3 x 5, then cov3D 6 x 3 and scale / rotation 7 x 3 over (ux^2, ux uy, uy^2)
			const float* rf = (const float*)&ent[j][3];
#pragma unroll
			for (int r = 0; r < 3; r++)
			{
				const float l = w * (rf[r * 5] * ux + rf[r * 5 + 1] * uy + rf[r * 5 + 2] * uxx + rf[r * 5 + 3] * uxy + rf[r * 5 + 4] * uyy);
				atomicAdd(acc + (9 + r) * 64, (double)(l * l));
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int r = 0; r < 13; r++)
			{
				const float l = w * (rf[15 + r * 3] * uxx + rf[15 + r * 3 + 1] * uxy + rf[15 + r * 3 + 2] * uyy);
				atomicAdd(acc + (12 + r) * 64, (double)(l * l));
				if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
			}
		}
		__builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's own ds_add instructions have retired
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if ((uint32_t)lane < m)
		{
			const size_t id = my_id;
			float a;
			if ((a = (float)s_acc[wave][0][lane]) != 0.f) atomicAdd(b.dL_dmean2D + 3 * id, a);
			if ((a = (float)s_acc[wave][1][lane]) != 0.f) atomicAdd(b.dL_dmean2D + 3 * id + 1, a);
			if ((a = (float)s_acc[wave][2][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id, a);
			if ((a = (float)s_acc[wave][3][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id + 1, a);
			if ((a = (float)s_acc[wave][4][lane]) != 0.f) atomicAdd(b.dL_dconic + 4 * id + 3, a);
#pragma unroll
			for (int c = 0; c < 3; c++) if ((a = (float)s_acc[wave][5 + c][lane]) != 0.f) atomicAdd(b.dL_dcolors + 3 * id + c, a);
			if ((a = (float)s_acc[wave][8][lane]) != 0.f) atomicAdd(b.dL_dopacity + id, a);
#pragma unroll
			for (int c = 0; c < 3; c++) if ((a = (float)s_acc[wave][9 + c][lane]) != 0.f) atomicAdd(b.dL_dmean3D + 3 * id + c, a);
#pragma unroll
			for (int c = 0; c < 6; c++) if ((a = (float)s_acc[wave][12 + c][lane]) != 0.f) atomicAdd(b.dL_dcov3D + 6 * id + c, a);
#pragma unroll
			for (int c = 0; c < 3; c++) if ((a = (float)s_acc[wave][18 + c][lane]) != 0.f) atomicAdd(b.dL_dscale + 3 * id + c, a);
#pragma unroll
			for (int c = 0; c < 4; c++) if ((a = (float)s_acc[wave][21 + c][lane]) != 0.f) atomicAdd(b.dL_drot + 4 * id + c, a);
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// Sum of 32 per-lane values over the 64 lanes, all 32 at once: a butterfly in which a lane hands half of what it still holds to its
// partner and adds the partner's other half to its own (16 v_permlane32_swap, 8 v_permlane16_swap, then 4 + 2 + 1 values over DPP):
// 72 instructions instead of 32 x 6.  Lane l returns the total of v[l >> 1].  All 64 lanes must be executing.
__device__ __forceinline__ float fr_reduce_scatter32(float (&v)[32], int lane)
{
	float s[16], t[8], r[4], q[2];
#pragma unroll
	for (int k = 0; k < 16; k++)
	{
		const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 16]), false, false);
		s[k] = __uint_as_float(x[0]) + __uint_as_float(x[1]);       // lanes < 32: v[k] of both halves; lanes >= 32: v[k + 16]
	}
#pragma unroll
	for (int k = 0; k < 8; k++)
	{
		const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[k]), __float_as_uint(s[k + 8]), false, false);
		t[k] = __uint_as_float(x[0]) + __uint_as_float(x[1]);       // even rows: s[k]; odd rows: s[k + 8]
	}
	const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		const float keep = b3 ? t[k + 4] : t[k], send = b3 ? t[k] : t[k + 4];
		r[k] = keep + __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(send), 0x128, 0xf, 0xf, false));     // row_ror:8
	}
#pragma unroll
	for (int k = 0; k < 2; k++)
	{
		const float keep = b2 ? r[k + 2] : r[k], send = b2 ? r[k] : r[k + 2];
		const int o = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)__float_as_uint(send), 0x1B, 0xf, 0xf, false), 0x141, 0xf, 0xf, false);   // lane ^ 4
		q[k] = keep + __uint_as_float((uint32_t)o);
	}
	const float keep = b1 ? q[1] : q[0], send = b1 ? q[0] : q[1];
	const float h = keep + __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(send), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
	return h + __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(h), 0xB1, 0xf, 0xf, false));                 // quad_perm [1,0,3,2]
}

// The same for 16 values: lane l returns the total of v[l >> 2].
__device__ __forceinline__ float fr_reduce_scatter16(float (&v)[16], int lane)
{
	float s[8], t[4], r[2];
#pragma unroll
	for (int k = 0; k < 8; k++)
	{
		const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 8]), false, false);
		s[k] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
	}
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(s[k]), __float_as_uint(s[k + 4]), false, false);
		t[k] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
	}
	const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
	for (int k = 0; k < 2; k++)
	{
		const float keep = b3 ? t[k + 2] : t[k], send = b3 ? t[k] : t[k + 2];
		r[k] = keep + __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(send), 0x128, 0xf, 0xf, false));     // row_ror:8
	}
	const float keep = b2 ? r[1] : r[0], send = b2 ? r[0] : r[1];
	const int o = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)__float_as_uint(send), 0x1B, 0xf, 0xf, false), 0x141, 0xf, 0xf, false);   // lane ^ 4
	float h = keep + __uint_as_float((uint32_t)o);
	h = h + __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(h), 0x4E, 0xf, 0xf, false));                   // quad_perm [2,3,0,1]
	return h + __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(h), 0xB1, 0xf, 0xf, false));                // quad_perm [1,0,3,2]
}

// POW = 2: the 25 squared leaves (rows from k_backward_sq_rows).  POW = 1 (training): the nine sums k_backward_lin_walk keeps per
// candidate -- screen-space mean 2, conic 3, colour 3, opacity 1 -- which k_backward_finish then pushes through the Jacobian chain;
// no rows, 6 KiB of LDS.
template <int POW, int NCH>
__global__ __launch_bounds__(FR_THREADS, 4) void k_backward_sq_leaves(FrParams p, FrBwdArgs b, const float* __restrict__ rows, FrSqSegArgs sg)
{
	static_assert(POW == 1 || NCH == 3, "the power-2 leaves are those of one image");
	constexpr int NR4 = FR_SQ_ROWS / 4, EF4 = NCH == 6 ? 4 : 3;
	__shared__ float4 s_ent[4][32][EF4];            // half a chunk at a time: 34 KiB per workgroup, four workgroups per CU
	__shared__ float4 s_rows[4][POW == 2 ? NR4 : 1][32];      // the leaf rows, [float4 of the row block][candidate]: written by LDS-direct loads
	if (p.status[1] || (uint32_t)p.status[0] > sg.num_rendered) return;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const float4* splat = (const float4*)p.splat;
	float4 (*ent)[EF4] = s_ent[wave];
	float4 (*rws)[32] = s_rows[wave];
	const size_t HW = (size_t)p.H * p.W;
	const float hw = (float)(0.5 * p.W), hh = (float)(0.5 * p.H);

	// where this lane's total goes: lane 2 l holds leaf l (fr_reduce_scatter32; POW = 1: lane 4 l, fr_reduce_scatter16); leaves in FR_SQ_NL order
	float* dst = nullptr;
	uint32_t dstride = 0;
	{
		const int l = POW == 2 ? lane >> 1 : lane >> 2;
		constexpr int NL = POW == 2 ? FR_SQ_NL : (NCH == 6 ? 14 : 9);
		if ((lane & (POW == 2 ? 1 : 3)) == 0 && l < NL)
		{
			if (l < 2) { dst = b.dL_dmean2D + l; dstride = 3; }
			else if (l < 5) { dst = b.dL_dconic + (l == 4 ? 3 : l - 2); dstride = 4; }
			else if (l < 8) { dst = b.dL_dcolors + (l - 5); dstride = 3; }
			else if constexpr (NCH == 6)
			{
				// the image pair: colours of the second image, opacity, the second image's screen-space mean
				if (l < 11) { dst = b.dL_dcolors2 + (l - 8); dstride = 3; }
				else if (l < 12) { dst = b.dL_dopacity; dstride = 1; }
				else { dst = b.dL_dmean2D_2 + (l - 12); dstride = 3; }
			}
			else if (l < 9) { dst = b.dL_dopacity; dstride = 1; }
			else if (l < 12) { dst = b.dL_dmean3D + (l - 9); dstride = 3; }
			else if (l < 18) { dst = b.dL_dcov3D + (l - 12); dstride = 6; }
			else if (l < 21) { dst = b.dL_dscale + (l - 18); dstride = 3; }
			else { dst = b.dL_drot + (l - 21); dstride = 4; }
		}
	}

	const uint32_t n_full = sg.ctl[0] < sg.n_chunks ? sg.ctl[0] : sg.n_chunks, n_last = sg.ctl[1] < sg.n_chunks ? sg.ctl[1] : sg.n_chunks;
	const uint32_t n_work = n_full + n_last;
	// wave w of W takes items w, w + W, ... of the list (full chunks first): every wave gets its share of the full chunks, and no
	// cursor (12 000 atomics on one address were a fifth of the kernel's time)
	const uint32_t W_all = gridDim.x * 4u;
	for (uint32_t wi = blockIdx.x * 4u + (uint32_t)wave; wi < n_work; wi += W_all)
	{
		{
			const uint2 we = wi < n_full ? sg.work[wi] : sg.work[(size_t)sg.n_chunks + (wi - n_full)];
			const size_t cid = we.x;
			const uint32_t tile = we.y >> 2, strip = we.y & 3u;
			const uint32_t tx = tile % p.gx, ty = tile / p.gx;
			const uint32_t pxx = tx * FR_BLOCK_X + ((uint32_t)lane & 15u), pxy = ty * FR_BLOCK_Y + 4u * strip + ((uint32_t)lane >> 4);
			const bool inside = pxx < (uint32_t)p.W && pxy < (uint32_t)p.H;
			const float pfx = (float)pxx, pfy = (float)pxy;
			const size_t pix = (size_t)p.W * pxy + pxx;
			const float T_final = inside ? b.final_T[pix] : 0.f;
			const uint32_t ncontrib = inside ? b.n_contrib[pix] : 0u;
			float g[NCH];
#pragma unroll
			for (int c = 0; c < NCH; c++) g[c] = 0.f;
			if (inside)
			{
				g[0] = b.dL_dpix[pix]; g[1] = b.dL_dpix[HW + pix]; g[2] = b.dL_dpix[2 * HW + pix];
				if constexpr (NCH == 6) { g[3] = b.dL_dpix2[pix]; g[4] = b.dL_dpix2[HW + pix]; g[5] = b.dL_dpix2[2 * HW + pix]; }
			}
			const float bg_dot = p.bg[0] * g[0] + p.bg[1] * g[1] + p.bg[2] * g[2];
			float bg_dot_b = 0.f;
			if constexpr (NCH == 6) bg_dot_b = p.bg[0] * g[3] + p.bg[1] * g[4] + p.bg[2] * g[5];
			FrBwdPixT<NCH> st;
			fr_bwd_state_load(sg.summ + cid * (FR_BWD_MAP_FLOATS(NCH) * 64) + lane, st);          // the state in front of the chunk (k_backward_sq_prefix)
			uint32_t my_id, my_pos;
			{ const uint2 qe = sg.list[cid * 64 + lane]; my_id = qe.x; my_pos = qe.y; }          // (lanes beyond the chunk's candidates: {0, 0}, parked but never read)
			const float4 q0 = splat[2 * (size_t)my_id], q1 = splat[2 * (size_t)my_id + 1];      // {x, y, conx, cony} {conz, opacity, depth, ext}
			unsigned long long todo_all, mask;
			{ const uint2 t = sg.ctodo[cid]; todo_all = (unsigned long long)t.y << 32 | t.x; }
			{ const uint2 t = sg.pixmask[cid * 64 + lane]; mask = (unsigned long long)t.y << 32 | t.x; }
			// back to front: the chunk holds its candidates in descending list position, lane 0 first; 32 of them are parked at a time
			for (int h = 0; h < 2; h++)
			{
				unsigned long long todo = h ? (todo_all >> 32) << 32 : todo_all & 0xffffffffull;
				if (todo == 0ull) continue;
				__builtin_amdgcn_wave_barrier();                                // the reads of what ent held before are done
				if ((lane >> 5) == h)
				{
					const int e = lane & 31;
					ent[e][0] = q0;
					ent[e][1] = make_float4(q1.x, q1.y, fr_power_threshold(q1.y), __uint_as_float(my_pos));
					ent[e][2] = make_float4(b.colors[3 * (size_t)my_id], b.colors[3 * (size_t)my_id + 1], b.colors[3 * (size_t)my_id + 2], 0.f);
					if constexpr (NCH == 6) ent[e][3] = make_float4(b.colors2[3 * (size_t)my_id], b.colors2[3 * (size_t)my_id + 1], b.colors2[3 * (size_t)my_id + 2], 0.f);
				}
				if constexpr (POW == 2)
				{
					// the 224 bytes of leaf rows of the 32 candidates, global memory -> LDS with no register in between (through registers the
					// compiler waits for each of the fourteen loads in turn): one instruction moves float4 2 kk of the 32 candidates on lanes
					// 0-31 and float4 2 kk + 1 on lanes 32-63 -- the lane-linear image the LDS-direct load writes is exactly rws[2 kk .. 2 kk + 1][.]
					const uint32_t cand_id = (uint32_t)__shfl((int)my_id, (lane & 31) + 32 * h, 64);     // (a candidate beyond m: Gaussian 0's rows, never read)
					const float4* rsrc = (const float4*)(rows + (size_t)cand_id * FR_SQ_ROW_STRIDE) + (lane >> 5);
#pragma unroll
					for (int kk = 0; kk < NR4 / 2; kk++) __builtin_amdgcn_global_load_lds(rsrc + 2 * kk, &rws[2 * kk][0], 16, 0, 0);
					__builtin_amdgcn_s_waitcnt(0x0f70);                            // vmcnt(0)
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
				while (todo != 0ull)
				{
					const int jj = __builtin_amdgcn_readfirstlane(__builtin_ctzll(todo));
					todo &= todo - 1ull;
					const int j = jj & 31;
					const float4 r0 = ent[j][0], r1 = ent[j][1], r2 = ent[j][2];
					float dx = 0.f, dy = 0.f, G = 0.f, alpha = 0.f;
					const bool ok = ((mask >> jj) & 1ull) != 0ull && fr_sq_test(r0, r1, pfx, pfy, ncontrib, dx, dy, G, alpha);
					if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
					const float cx = r0.z, cy = r0.w, cz = r1.x, o = r1.y;
					float rc[NCH];
					rc[0] = r2.x; rc[1] = r2.y; rc[2] = r2.z;
					if constexpr (NCH == 6) { const float4 r3 = ent[j][3]; rc[3] = r3.x; rc[4] = r3.y; rc[5] = r3.z; }
					float tot;
					if constexpr (POW == 2)
					{
						float w = 0.f, wcol = 0.f;
						if (ok)
						{
							// backward.cu:978-1038
							float da, db;
							fr_bwd_update(st, alpha, rc, g, da, db);
							wcol = alpha * st.Tc;
							if (bg_dot != 0.f) da += (-T_final / (1.f - alpha)) * bg_dot;
							w = (o * da) * G;                                              // dL_dG * G
						}
						// from here on every lane executes: a lane without the pair carries w = wcol = 0, so all its squares are 0
						const float ux = -(cx * dx + cy * dy), uy = -(cy * dx + cz * dy);   // u = -conic d
						const float uxx = ux * ux, uxy = ux * uy, uyy = uy * uy;
						float v[32];
						{
							const float l0 = w * ux * hw, l1 = w * uy * hh;                                   // dL_dmean2D (backward.cu:1021-1024)
							const float hq = -0.5f * w;
							const float l2 = hq * dx * dx, l3 = hq * dx * dy, l4 = hq * dy * dy;               // dL_dconic x, y, w (1026-1029)
							v[0] = l0 * l0; v[1] = l1 * l1; v[2] = l2 * l2; v[3] = l3 * l3; v[4] = l4 * l4;
						}
#pragma unroll
						for (int c = 0; c < 3; c++) { const float l = wcol * g[c]; v[5 + c] = l * l; }         // dL_dcolors
						{
							const float4 rz = rws[13][j];                                                     // rows[52..55]: {.., .., 1 / opacity^2, pad}
							v[8] = w * w * rz.z;                                                               // dL_dopacity = G dL_dalpha = w / opacity
						}
						// the 16 rows: mean 3 x 5, then cov3D 6 x 3 and scale / rotation 7 x 3 over (ux^2, ux uy, uy^2), applied to w gamma
						float rf[FR_SQ_ROWS];
#pragma unroll
						for (int k = 0; k < NR4; k++)
						{
							const float4 t = rws[k][j];
							rf[4 * k] = t.x; rf[4 * k + 1] = t.y; rf[4 * k + 2] = t.z; rf[4 * k + 3] = t.w;
						}
						const float g0 = w * ux, g1 = w * uy, g2 = w * uxx, g3 = w * uxy, g4 = w * uyy;
#pragma unroll
						for (int r = 0; r < 3; r++)
						{
							const float l = rf[r * 5] * g0 + rf[r * 5 + 1] * g1 + rf[r * 5 + 2] * g2 + rf[r * 5 + 3] * g3 + rf[r * 5 + 4] * g4;
							v[9 + r] = l * l;
						}
#pragma unroll
						for (int r = 0; r < 13; r++)
						{
							const float l = rf[15 + r * 3] * g2 + rf[15 + r * 3 + 1] * g3 + rf[15 + r * 3 + 2] * g4;
							v[12 + r] = l * l;
						}
#pragma unroll
						for (int k = FR_SQ_NL; k < 32; k++) v[k] = 0.f;
						tot = fr_reduce_scatter32(v, lane);
					}
					else
					{
						// backward.cu:978-1038: the terms of k_backward_lin_walk, zero on a lane without the pair
						float da = 0.f, db = 0.f, wcol = 0.f;
						if (ok)
						{
							fr_bwd_update(st, alpha, rc, g, da, db);
							wcol = alpha * st.Tc;
							const float bgf = -T_final / (1.f - alpha);
							if (bg_dot != 0.f) da += bgf * bg_dot;
							if (NCH == 6 && bg_dot_b != 0.f) db += bgf * bg_dot_b;
						}
						const float dL_dalpha = da + db;
						const float dL_dG = o * dL_dalpha;
						const float gdx = G * dx, gdy = G * dy;
						const float dG_ddelx = -gdx * cx - gdy * cy, dG_ddely = -gdy * cz - gdx * cy;
						const float oa = NCH == 6 ? o * da : dL_dG;                   // screen-space gradient of the first image (of the only one)
						float v[16];
						v[0] = oa * dG_ddelx * hw; v[1] = oa * dG_ddely * hh;
						v[2] = -0.5f * gdx * dx * dL_dG; v[3] = -0.5f * gdx * dy * dL_dG; v[4] = -0.5f * gdy * dy * dL_dG;
#pragma unroll
						for (int c = 0; c < NCH; c++) v[5 + c] = wcol * g[c];
						v[5 + NCH] = G * dL_dalpha;
						if constexpr (NCH == 6) { const float ob = o * db; v[12] = ob * dG_ddelx * hw; v[13] = ob * dG_ddely * hh; }
#pragma unroll
						for (int k = (NCH == 6 ? 14 : 9); k < 16; k++) v[k] = 0.f;
						tot = fr_reduce_scatter16(v, lane);
					}
					const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)my_id, jj);
					if (dst != nullptr && tot != 0.f) atomicAdd(dst + (size_t)id * dstride, tot);
				}
			}
		}
	}
}

// Per Gaussian: the Jacobian chain of backward.cu:276-475,532-583 applied ONCE to the summed u (power 1 only).
template <bool HAS_SR, bool HAS_SH>
__global__ __launch_bounds__(FR_THREADS) void k_backward_finish(FrParams p, FrBwdArgs b, float* __restrict__ dL_dsh)
{
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= p.P || p.radii[i] <= 0) return;
	float vm[16], pm[16];
#pragma unroll
	for (int k = 0; k < 16; k++) { vm[k] = p.view[k]; pm[k] = p.proj[k]; }
	fr_f3 po = { p.means3D[3 * (size_t)i], p.means3D[3 * (size_t)i + 1], p.means3D[3 * (size_t)i + 2] };
	float c3[6];
#pragma unroll
	for (int k = 0; k < 6; k++) c3[k] = p.cov3D[6 * (size_t)i + k];
	float u[5] = { b.dL_dmean2D[3 * (size_t)i], b.dL_dmean2D[3 * (size_t)i + 1], b.dL_dconic[4 * (size_t)i],
	               b.dL_dconic[4 * (size_t)i + 1], b.dL_dconic[4 * (size_t)i + 3] };
	if (b.dL_dmean2D_b) { u[0] += b.dL_dmean2D_b[3 * (size_t)i]; u[1] += b.dL_dmean2D_b[3 * (size_t)i + 1]; }
	float A[3][5];
	float B[6][3];
	fr_mean_jacobian(po, c3, vm, pm, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, A, B);
	float dm[3];
#pragma unroll
	for (int r = 0; r < 3; r++) dm[r] = A[r][0] * u[0] + A[r][1] * u[1] + A[r][2] * u[2] + A[r][3] * u[3] + A[r][4] * u[4];
	if constexpr (HAS_SH)
	{
		fr_f3 cp = { p.campos[0], p.campos[1], p.campos[2] };
		float coef[16];
		float Dm[3][3];
		fr_sh_backward_jacobian(p.D, po, cp, p.shs + (size_t)p.M * i, p.clamped + 3 * (size_t)i, coef, Dm);
		const float gc[3] = { b.dL_dcolors[3 * (size_t)i], b.dL_dcolors[3 * (size_t)i + 1], b.dL_dcolors[3 * (size_t)i + 2] };
#pragma unroll
		for (int r = 0; r < 3; r++) dm[r] += Dm[r][0] * gc[0] + Dm[r][1] * gc[1] + Dm[r][2] * gc[2];
		if (p.D > 0)
		{
			const int nsh = (p.D + 1) * (p.D + 1);
			for (int k = 0; k < nsh; k++)
				for (int c = 0; c < 3; c++)
					dL_dsh[((size_t)i * p.M + k) * 3 + c] = coef[k] * (p.clamped[3 * (size_t)i + c] ? 0.f : 1.f) * gc[c];
		}
	}
#pragma unroll
	for (int r = 0; r < 3; r++) b.dL_dmean3D[3 * (size_t)i + r] = dm[r];
	float dcov[6];
#pragma unroll
	for (int r = 0; r < 6; r++)
	{
		dcov[r] = B[r][0] * u[2] + B[r][1] * u[3] + B[r][2] * u[4];
		b.dL_dcov3D[6 * (size_t)i + r] = dcov[r];
	}
	if constexpr (HAS_SR)
	{
		fr_f3 sc = { p.scales[3 * (size_t)i], p.scales[3 * (size_t)i + 1], p.scales[3 * (size_t)i + 2] };
		fr_f4 q = { p.rots[4 * (size_t)i], p.rots[4 * (size_t)i + 1], p.rots[4 * (size_t)i + 2], p.rots[4 * (size_t)i + 3] };
		fr_f3 ds; fr_f4 dr;
		fr_cov3d_backward(sc, p.mod, q, dcov, ds, dr);
		b.dL_dscale[3 * (size_t)i] = ds.x; b.dL_dscale[3 * (size_t)i + 1] = ds.y; b.dL_dscale[3 * (size_t)i + 2] = ds.z;
		b.dL_drot[4 * (size_t)i] = dr.x; b.dL_drot[4 * (size_t)i + 1] = dr.y; b.dL_drot[4 * (size_t)i + 2] = dr.z; b.dL_drot[4 * (size_t)i + 3] = dr.w;
	}
}

// dL_dsh, finished per Gaussian: every per-pixel dL_dsh[k][c] is coef_k(dir) * clamp_mask_c * dL_dcolor_c, so its sum of
// grad_power-th powers is coef_k^power * mask_c * (accumulated dL_dcolors[c]).  Reference quirk kept: nothing is
// accumulated when the SH degree is 0 (backward.cu:1117).
__global__ __launch_bounds__(FR_THREADS) void k_finish_sh(FrParams p, const float* __restrict__ dL_dcolors, int power,
                                                          float* __restrict__ dL_dsh)
{
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= p.P || p.D <= 0 || p.radii[i] <= 0) return;
	fr_f3 pos = { p.means3D[3 * (size_t)i], p.means3D[3 * (size_t)i + 1], p.means3D[3 * (size_t)i + 2] };
	fr_f3 cp = { p.campos[0], p.campos[1], p.campos[2] };
	float coef[16];
	float Dm[3][3];
	fr_sh_backward_jacobian(p.D, pos, cp, p.shs + (size_t)p.M * i, p.clamped + 3 * (size_t)i, coef, Dm);
	const int nsh = (p.D + 1) * (p.D + 1);
	for (int k = 0; k < nsh; k++)
	{
		const float ck = fr_powi(coef[k], power);
		for (int c = 0; c < 3; c++)
		{
			const float mask = p.clamped[3 * (size_t)i + c] ? 0.f : 1.f;
			dL_dsh[((size_t)i * p.M + k) * 3 + c] = ck * mask * dL_dcolors[3 * (size_t)i + c];
		}
	}
}

// =========================================================================================================
// host side
// =========================================================================================================
static inline size_t fr_align(size_t x) { return (x + 255) & ~(size_t)255; }

// Gaussians per thread of k_preprocess / k_scatter_keys: enough per workgroup for the LDS tile histograms to aggregate,
// enough workgroups to fill 256 CUs.
static inline int fr_pick_G(long long P, long long V)
{
	long long gwant = (P * V) / ((long long)FR_THREADS * 2048);
	return (int)(gwant < 1 ? 1 : (gwant > FR_G_MAX ? FR_G_MAX : gwant));
}
// multi-view front end: 256*G Gaussians x VC views per workgroup
static inline int fr_pick_G_views(long long P)
{
#ifdef FR_AB
	static const int forced = fr_env_int("FR_GV");            // FR_GV=<n>: A/B runs
	if (forced > 0) return forced > 8 ? 8 : forced;
#endif
	// ~400 workgroups along P.  (Measured on MI355X, 500k Gaussians x 64 views, ms per step, round 2: G = 1: 2.24, 2: 2.20, 3: 2.17-2.20,
	// 4: 2.19-2.22, 7: 2.26, 10: 2.37; round 3, with the survivors of the 256-Gaussian rounds batched across rounds and the key
	// scatter at the end of the workgroup: G = 2: 1.69, 3: 1.69-1.72, 5: 1.67-1.68, 6: 1.68-1.70, 8: 1.68-1.70.)
	long long gwant = (P + (long long)FR_THREADS * 200) / ((long long)FR_THREADS * 400);
	return (int)(gwant < 1 ? 1 : (gwant > 8 ? 8 : gwant));
}
static inline int fr_pick_VC(long long T)
{
	long long vc = 8192 / (T < 1 ? 1 : T);
	if (vc > FR_VC_MAX) vc = FR_VC_MAX;                       // the kernel's per-view counters (s_n / s_ref) hold FR_VC_MAX views
#ifdef FR_AB
	static const int forced = fr_env_int("FR_VC");            // FR_VC=<n>: A/B runs
	if (forced > 0) return (int)(forced > vc ? (vc < 1 ? 1 : vc) : forced);
#endif
	if (vc > 4) vc = 4;                                       // 4 views per workgroup: 2.16 ms per step against 2.20 for 8 and 2.19 for 2
	return (int)(vc < 1 ? 1 : (vc > 8 ? 8 : vc));
}
// View groups of a records-mode fr_fisher_views call (see there).  FR_GROUPS=<n> (A/B runs): n groups; default ONE launch for all
// views.  Measured on MI355X (500k Gaussians x 64 views, profiles/r03_b_groups_timeline.txt): with 2 groups the second group's
// projection kernel keeps its speed under the first group's tile kernel (334 vs 341 us), but the tile kernel takes 973 us
// instead of 540 and the scan / scatter / sort tiers beside it 2-8x their time -- 2.32 ms per step against 2.10 (4 groups: 2.56;
// the tile stream at the lowest or the highest priority: the same).  Both halves want the same wave slots and the same L2.
static inline int fr_pick_groups(int V)
{
#ifdef FR_AB
	static const int forced = fr_env_int("FR_GROUPS");
	int n = forced > 0 ? forced : 1;
	if (n > 4) n = 4;                                         // FR_MAX_GROUPS
	while (n > 1 && V < 16 * n) n--;                          // at least 16 views per group
	return n;
#else
	(void)V;
	return 1;
#endif
}
// LDS bytes of k_preprocess_views_c for vc views per workgroup (fixed key segments keep a second [vc][T] array, the cursors)
static inline size_t fr_lds_views_c(int vc, int T, bool fixed)
{
	return ((size_t)vc * T * (fixed ? 2 : 1) + (size_t)FR_THREADS * (vc + 1) + 12 * (size_t)vc + 13 * (size_t)FR_THREADS) * 4;
}
// Views per workgroup of k_preprocess_views_c, and whether fixed key segments stay: fewer views where the LDS would pass 64 KiB,
// packed lists (tile_cap = 0) where even one view does not fit.  ONE place decides this -- fr_fisher_views calls it before it fixes
// the record stride (the 80-byte score records exist with fixed segments only), fr_bin_pipeline calls it again and gets the same answer.
static inline int fr_plan_views_c(int T, int vc, uint32_t& tile_cap)
{
	while (tile_cap && vc > 1 && fr_lds_views_c(vc, T, true) > 65536) vc >>= 1;
	if (tile_cap && fr_lds_views_c(vc, T, true) > 65536) tile_cap = 0;
	return vc;
}
static inline long long fr_preprocess_blocks(long long P, long long V)
{
	// the larger of the two decompositions fr_bin_pipeline can pick (blk_base is sized with it)
	const long long g = fr_pick_G(P, V) < fr_pick_G_views(P) ? fr_pick_G(P, V) : fr_pick_G_views(P);
	const long long per_block = (long long)FR_THREADS * g;
	return (P + per_block - 1) / per_block;
}

struct FrLayout {
	// geometry
	size_t splat, cov3D, rgb, clamped, packed, geom_bytes;
	// image
	size_t tile_cnt, tile_off, tile_fill, final_T, n_contrib, status, big_list, blk_base, img_bytes;
	// binning
	size_t keys, bin_bytes;
};

static FrLayout fr_layout(int64_t P, int64_t W, int64_t H, int64_t V, int64_t max_rendered)
{
	FrLayout L;
	const int64_t T = ((W + 15) / 16) * ((H + 15) / 16);
	const size_t VP = (size_t)(V * P);
	size_t o = 0;
	L.splat = o; o = fr_align(o + VP * sizeof(FrSplat));
	L.cov3D = o; o = fr_align(o + (size_t)P * 24);
	L.rgb = o; o = fr_align(o + VP * 12);
	L.clamped = o; o = fr_align(o + VP * 3);
	L.packed = o; o = fr_align(o + (size_t)P * 256);     // per-Gaussian leaf rows of the power-2 backward (k_backward_sq_rows: 224 B; k_fisher_tile_v2<25>: 128 B), written by fr_backward
	L.geom_bytes = o > 0 ? o : 256;
	o = 0;
	L.tile_cnt = o; o = fr_align(o + (size_t)(V * T) * 4);
	L.tile_off = o; o = fr_align(o + (size_t)(V * T) * 4);
	L.tile_fill = o; o = fr_align(o + (size_t)(V * T) * 4);
	L.final_T = o; o = fr_align(o + (size_t)(V * W * H) * 4);
	L.n_contrib = o; o = fr_align(o + (size_t)(V * W * H) * 4);
	L.status = o; o = fr_align(o + 64);
	L.big_list = o; o = fr_align(o + (size_t)(V * T) * 4 + 64);
	// start of every projection workgroup's range inside a tile segment (k_preprocess claims it, k_scatter_keys then neither counts
	// nor claims again); only while the tile histogram fits LDS
	L.blk_base = o; o = fr_align(o + (T <= FR_MAX_LDS_TILES ? (size_t)V * (size_t)fr_preprocess_blocks(P, V) * (size_t)T * 4 : 0));
	L.img_bytes = o;
	L.keys = 0;
	L.bin_bytes = fr_align((size_t)(max_rendered > 0 ? max_rendered : 1) * 8);
	return L;
}

extern "C" int fr_version(void) { return FR_VERSION; }
#ifndef FR_SRC_HASH
#define FR_SRC_HASH "unstamped"
#endif
extern "C" const char* fr_build_id(void) { return FR_SRC_HASH; }
extern "C" const char* fr_last_error(void) { return g_err; }

extern "C" int fr_workspace_bytes(int32_t P, int32_t W, int32_t H, int64_t max_rendered, size_t out[3])
{
	if (P < 0 || W <= 0 || H <= 0 || max_rendered < 0 || !out) return fr_fail(FR_EINVAL, "fr_workspace_bytes: bad argument");
	FrLayout L = fr_layout(P, W, H, 1, max_rendered);
	out[0] = L.geom_bytes; out[1] = L.bin_bytes; out[2] = L.img_bytes;
	return FR_OK;
}

extern "C" int fr_workspace_layout(int32_t P, int32_t W, int32_t H, int64_t max_rendered, size_t o[11])
{
	if (P < 0 || W <= 0 || H <= 0 || max_rendered < 0 || !o) return fr_fail(FR_EINVAL, "fr_workspace_layout: bad argument");
	FrLayout L = fr_layout(P, W, H, 1, max_rendered);
	o[0] = L.splat; o[1] = L.cov3D; o[2] = L.rgb; o[3] = L.clamped;
	o[4] = L.tile_cnt; o[5] = L.tile_off; o[6] = L.tile_fill; o[7] = L.final_T; o[8] = L.n_contrib; o[9] = L.status;
	o[10] = L.keys;
	return FR_OK;
}

extern "C" int fr_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                               uint8_t* present, fr_stream_t stream)
{
	(void)projmatrix;
	if (P < 0) return fr_fail(FR_EINVAL, "fr_mark_visible: P < 0");
	if (P == 0) return FR_OK;
	if (!means3D || !viewmatrix || !present) return fr_fail(FR_EINVAL, "fr_mark_visible: null pointer");
	hipLaunchKernelGGL(k_mark_visible, dim3((P + FR_THREADS - 1) / FR_THREADS), dim3(FR_THREADS), 0, (hipStream_t)stream,
	                   P, means3D, viewmatrix, present);
	return fr_check_launch("k_mark_visible");
}

static int fr_validate(const fr_raster_cfg* cfg, const fr_gaussians* g, const char* who, bool need_opacity = true)
{
	static thread_local char buf[256];
	if (!cfg || !g) { snprintf(buf, sizeof(buf), "%s: null cfg/gaussians", who); return fr_fail(FR_EINVAL, buf); }
	if (cfg->P < 0 || cfg->image_width <= 0 || cfg->image_height <= 0) { snprintf(buf, sizeof(buf), "%s: bad P/W/H", who); return fr_fail(FR_EINVAL, buf); }
	if (cfg->P == 0) return FR_OK;
	if (!g->means3D || (need_opacity && !g->opacities) || !cfg->bg || !cfg->viewmatrix || !cfg->projmatrix)
	{ snprintf(buf, sizeof(buf), "%s: null means3D/opacities/bg/viewmatrix/projmatrix", who); return fr_fail(FR_EINVAL, buf); }
	if ((g->colors_precomp == nullptr) == (g->shs == nullptr))
	{ snprintf(buf, sizeof(buf), "%s: provide exactly one of SHs or precomputed colors", who); return fr_fail(FR_EINVAL, buf); }
	if (g->shs && (!cfg->campos || cfg->sh_coeffs <= 0 || (cfg->sh_degree + 1) * (cfg->sh_degree + 1) > cfg->sh_coeffs || cfg->sh_degree > 3 || cfg->sh_degree < 0))
	{ snprintf(buf, sizeof(buf), "%s: bad SH degree / coefficient count / campos", who); return fr_fail(FR_EINVAL, buf); }
	const bool sr = g->scales && g->rotations;
	if ((sr ? 1 : 0) + (g->cov3D_precomp ? 1 : 0) != 1 || ((g->scales != nullptr) != (g->rotations != nullptr)))
	{ snprintf(buf, sizeof(buf), "%s: provide exactly one of scale/rotation pair or precomputed 3D covariance", who); return fr_fail(FR_EINVAL, buf); }
	return FR_OK;
}

static void fr_fill_params(FrParams& p, const fr_raster_cfg* cfg, const fr_gaussians* g, int V)
{
	memset(&p, 0, sizeof(p));
	p.P = cfg->P; p.V = V; p.W = cfg->image_width; p.H = cfg->image_height;
	p.gx = (uint32_t)((p.W + 15) / 16); p.gy = (uint32_t)((p.H + 15) / 16); p.T = (int)(p.gx * p.gy);
	p.tanfovx = cfg->tanfovx; p.tanfovy = cfg->tanfovy;
	p.focal_y = p.H / (2.0f * cfg->tanfovy);   // rasterizer_impl.cu:222-223
	p.focal_x = p.W / (2.0f * cfg->tanfovx);
	p.mod = cfg->scale_modifier; p.D = cfg->sh_degree; p.M = cfg->sh_coeffs;
	p.prefiltered = cfg->prefiltered;
	p.bg = cfg->bg; p.view = cfg->viewmatrix; p.proj = cfg->projmatrix; p.campos = cfg->campos;
	p.means3D = g->means3D; p.colors = g->colors_precomp; p.shs = g->shs; p.opac = g->opacities;
	p.scales = g->scales; p.rots = g->rotations;
	p.VC = 1;
	p.small_max = FR_SORT_SMALL_KEYS;
	p.order = nullptr;
	p.legacy_sort = fr_debug_mode() == 6;
	FR_ABL(p.ablate = fr_debug_mode();)
}

// Launches cov3d, preprocess, scan, scatter, sort for V views.  p must carry the carved buffers.
// One launch instead of a dozen memsets: zero-fills up to FR_ZERO_MAX device buffers (all sizes multiples of 4 bytes).
#define FR_ZERO_MAX 12
// Up to FR_ZERO_MAX buffers cleared by one launch, in units of 16 bytes (a buffer's last unit may be shorter): end[i] = prefix sum
// of the units, dwords[i] = the buffer's length.  A buffer whose address is not 16-byte aligned is cleared dword by dword.
struct FrZeroList { uint32_t* ptr[FR_ZERO_MAX]; unsigned long long end[FR_ZERO_MAX]; unsigned long long dwords[FR_ZERO_MAX]; int n; };
__global__ __launch_bounds__(FR_THREADS) void k_zero_many(FrZeroList z)
{
	const unsigned long long total = z.end[z.n - 1];
	for (unsigned long long i = (unsigned long long)blockIdx.x * FR_THREADS + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * FR_THREADS)
	{
		int b = 0;
		while (i >= z.end[b]) b++;
		const unsigned long long u = i - (b ? z.end[b - 1] : 0ull);
		uint32_t* q = z.ptr[b] + 4 * u;
		if (4 * u + 4 <= z.dwords[b] && ((uintptr_t)q & 15u) == 0) *(uint4*)q = make_uint4(0u, 0u, 0u, 0u);
		else
			for (unsigned long long k = 4 * u; k < 4 * u + 4 && k < z.dwords[b]; k++) z.ptr[b][k] = 0u;
	}
}
struct FrZeroer {
	FrZeroList z; unsigned long long run = 0;
	FrZeroer() { z.n = 0; }
	void add(void* p, size_t bytes) { if (!p || bytes == 0) return; z.ptr[z.n] = (uint32_t*)p; z.dwords[z.n] = bytes / 4; run += (bytes / 4 + 3) / 4; z.end[z.n] = run; z.n++; }
	void launch(hipStream_t s)
	{
		if (z.n == 0) return;
		const unsigned long long blocks = (run + FR_THREADS * 4 - 1) / (FR_THREADS * 4);
		hipLaunchKernelGGL(k_zero_many, dim3((unsigned)(blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks))), dim3(FR_THREADS), 0, s, z);
	}
};

// Three side streams per host thread and device (created on first use, or by fr_init; they live with the thread):
// [0] the 1024-thread sort tier and [1] the scorer's per-(view, Gaussian) records of the single-view front end (fork / joins inside
// fr_bin_pipeline); [2] the tile kernels of fr_fisher_views when a call is cut into view groups (below).
struct FrSideStream {
	hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false; int device = -1;
};
static FrSideStream& fr_side_stream(int which = 0)
{
	static thread_local FrSideStream sides[3];
	FrSideStream& ss = sides[which];
	int dev = -1;
	(void)hipGetDevice(&dev);
	if (ss.device != dev)
	{
		if (ss.ok) { (void)hipEventDestroy(ss.fork); (void)hipEventDestroy(ss.join); (void)hipStreamDestroy(ss.stream); }
		// FR_TILE_PRIO=<-1|0|1> (A/B runs): priority of the tile stream [2] relative to the default: 1 = lowest, -1 = highest
#ifdef FR_AB
		static const int tile_prio = fr_env_int("FR_TILE_PRIO");
#else
		constexpr int tile_prio = 0;
#endif
		int lo = 0, hi = 0;
		(void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = least priority (largest number), hi = greatest
		const int prio = (which == 2 && tile_prio > 0) ? lo : (which == 2 && tile_prio < 0) ? hi : 0;
		ss.ok = hipStreamCreateWithPriority(&ss.stream, hipStreamNonBlocking, prio) == hipSuccess
		     && hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess
		     && hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
		ss.device = dev;
		(void)hipGetLastError();
	}
	return ss;
}

// Every fork onto a side stream is joined back into the caller's stream on EVERY way out of fr_bin_pipeline, the error returns
// included: the caller may free or reuse the workspace as soon as its own stream has drained, and a side-stream kernel that is
// still writing records or keys into it must therefore be ordered in front of whatever the caller enqueues next.
struct FrJoinGuard {
	hipStream_t s; FrSideStream* side[2] = { nullptr, nullptr }; int n = 0;
	explicit FrJoinGuard(hipStream_t caller) : s(caller) {}
	void forked(FrSideStream* ss) { side[n++] = ss; }
	int join()                                   // normal path: returns FR_OK or the error
	{
		int rc = FR_OK;
		for (int i = 0; i < n; i++)
		{
			if (hipEventRecord(side[i]->join, side[i]->stream) != hipSuccess || hipStreamWaitEvent(s, side[i]->join, 0) != hipSuccess)
				rc = fr_fail(FR_ELAUNCH, "side stream join failed");
		}
		n = 0;
		return rc;
	}
	~FrJoinGuard() { if (n) { char keep[sizeof(g_err)]; memcpy(keep, g_err, sizeof(keep)); (void)join(); memcpy(g_err, keep, sizeof(keep)); } }   // error path: keep the first message
};

// Score-only mode: the front end also produces the scorer's per-(view, Gaussian) records (k_pack_static, then phase C of
// k_preprocess_views; with the single-view front end, k_fisher_records after k_scatter_keys, beside the sorts, on the second side stream).
struct FrScorerPlan { int columns; bool form_a; FrRecordArgs ra; bool skip_pack = false; bool general = false; };   // general: out_H records of k_fisher_tile_v3g   // skip_pack: a view group after the first (the packed static records are per call)     // form_a: out_H mode, the records carry the mean Jacobian (k_fisher_tile_v3h)
template <int C> __global__ void k_pack_static(FrParams p, const float* __restrict__ H_inv, float* __restrict__ packed, float4* __restrict__ mt, float4* __restrict__ grp);
template <int C, bool LIST, bool FORM_A> __global__ void k_fisher_records(FrParams p, FrRecordArgs ra);

static int fr_bin_pipeline(FrParams& p, const fr_gaussians* g, hipStream_t s, const FrScorerPlan* plan = nullptr)
{
	int rc;
	const int P = p.P;
	{
		FrZeroer z;
		z.add(p.tile_cnt, (size_t)p.V * p.T * 4); z.add(p.status, 16); z.add(p.big_list, 64);
		if (p.view_work) z.add(p.view_work, (size_t)p.V * 4);
		if (p.vis_count) z.add(p.vis_count, (size_t)p.V * 4);
		if (p.num_rendered && p.vis_list != nullptr && p.T <= FR_MAX_LDS_TILES) z.add(p.num_rendered, (size_t)p.V * 4);   // counted by k_preprocess_views
		z.launch(s);
	}
	const bool multi = p.vis_list != nullptr && p.T <= FR_MAX_LDS_TILES;
	if (g->cov3D_precomp) p.cov3D = g->cov3D_precomp;
	else if (plan && multi) p.cov3D = nullptr;       // records mode of the multi-view front end: k_pack_static builds cov3D itself, nothing else reads it
	else
	{
		hipLaunchKernelGGL(k_cov3d, dim3((P + FR_THREADS - 1) / FR_THREADS), dim3(FR_THREADS), 0, s, P, g->scales, p.mod, g->rotations, p.cov3D_out);
		if ((rc = fr_check_launch("k_cov3d"))) return rc;
		p.cov3D = p.cov3D_out;
	}
	p.G = multi ? fr_pick_G_views(P) : fr_pick_G(P, p.V);
	p.VC = fr_pick_VC(p.T);
	const int per_block = FR_THREADS * p.G;
	dim3 gridP((P + per_block - 1) / per_block, p.V);
	const size_t hist_lds = p.T <= FR_MAX_LDS_TILES ? (size_t)p.T * 4 : 16;
	if (plan && !plan->skip_pack)
	{
		const float* shared_hinv = plan->ra.hinv_stride ? nullptr : plan->ra.H_inv;
		dim3 gp((P + FR_THREADS - 1) / FR_THREADS);
		if (plan->columns == 4) hipLaunchKernelGGL((k_pack_static<4>), gp, dim3(FR_THREADS), 0, s, p, shared_hinv, (float*)plan->ra.packed, (float4*)plan->ra.mt, (float4*)plan->ra.grp);
		else hipLaunchKernelGGL((k_pack_static<11>), gp, dim3(FR_THREADS), 0, s, p, shared_hinv, (float*)plan->ra.packed, (float4*)plan->ra.mt, (float4*)plan->ra.grp);
		if ((rc = fr_check_launch("k_pack_static"))) return rc;
	}
	FrJoinGuard joins(s);
	if (multi)
	{
		// compact records: written once, in place, by k_preprocess_views_c (FR_DEBUG_MODE=20 keeps the parking form for A/B runs)
		const bool once = plan && plan->ra.comp != nullptr && fr_debug_mode() != 20;
		if (!once) p.tile_cap = 0;                       // fixed key segments are filled by k_preprocess_views_c only
		const bool wanted_fixed = p.tile_cap != 0;
		if (once) p.VC = fr_plan_views_c(p.T, p.VC, p.tile_cap);
		// (fr_fisher_views has taken the same decision before it fixed the record stride: the 80-byte score records go with fixed segments)
		if (once && wanted_fixed && !p.tile_cap) return fr_fail(FR_EINVAL, "fr_bin_pipeline: fixed key segments planned, but they do not fit the LDS");
		auto lds_c_of = [&](int vc) { return fr_lds_views_c(vc, p.T, p.tile_cap != 0); };
		dim3 gridV(gridP.x, (p.V + p.VC - 1) / p.VC);
		const size_t lds = ((size_t)p.VC * p.T + (size_t)FR_THREADS * p.VC + 12 * (size_t)p.VC + 2 * (size_t)p.VC * 8 * (size_t)p.G) * 4;
		FrRecordArgs ra = plan ? plan->ra : FrRecordArgs{ nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 6 };
		// The records are phase C of the projection kernel.  (Measured on MI355X, 500k Gaussians x 64 views: as a kernel of their
		// own beside scan / scatter / sort -- on a second stream, also at the lowest stream priority -- the step takes 2.62 ms
		// against 2.53 ms: the records' waves slow the latency-bound scatter and the one-workgroup-per-CU sort tier down.)
		const bool dk = p.tile_cap != 0;
		const size_t lds_c = lds_c_of(p.VC);
		if (once && plan->general && plan->columns == 4 && dk) hipLaunchKernelGGL((k_preprocess_views_c<4, 2, true>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && plan->general && plan->columns == 4) hipLaunchKernelGGL((k_preprocess_views_c<4, 2, false>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && plan->general && dk) hipLaunchKernelGGL((k_preprocess_views_c<11, 2, true>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && plan->general) hipLaunchKernelGGL((k_preprocess_views_c<11, 2, false>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && plan->form_a && dk) hipLaunchKernelGGL((k_preprocess_views_c<4, 1, true>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && plan->form_a) hipLaunchKernelGGL((k_preprocess_views_c<4, 1, false>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && plan->columns == 4 && dk) hipLaunchKernelGGL((k_preprocess_views_c<4, 0, true>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && plan->columns == 4) hipLaunchKernelGGL((k_preprocess_views_c<4, 0, false>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once && dk) hipLaunchKernelGGL((k_preprocess_views_c<11, 0, true>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (once) hipLaunchKernelGGL((k_preprocess_views_c<11, 0, false>), gridV, dim3(FR_THREADS), lds_c, s, p, ra);
		else if (!plan) hipLaunchKernelGGL((k_preprocess_views<0, false>), gridV, dim3(FR_THREADS), lds, s, p, ra);
#ifdef FR_AB
		else if (plan->form_a) hipLaunchKernelGGL((k_preprocess_views<-4, true>), gridV, dim3(FR_THREADS), lds, s, p, ra);
		else if (plan->columns == 4) hipLaunchKernelGGL((k_preprocess_views<4, true>), gridV, dim3(FR_THREADS), lds, s, p, ra);
		else hipLaunchKernelGGL((k_preprocess_views<11, true>), gridV, dim3(FR_THREADS), lds, s, p, ra);
#else
		else return fr_fail(FR_EINVAL, "fr_bin_pipeline: a records plan of the multi-view front end needs compact records");
#endif
		if ((rc = fr_check_launch("k_preprocess_views"))) return rc;
	}
	else
	{
		p.tile_cap = 0;
		hipLaunchKernelGGL(k_preprocess, gridP, dim3(FR_THREADS), hist_lds, s, p);
		if ((rc = fr_check_launch("k_preprocess"))) return rc;
	}
	// fixed segments with room for k_sort_part: lists beyond `small_max` keys go to it (FR_PART_MIN, rig: A/B of the hand-over)
	if (multi && p.tile_cap >= 2u * (uint32_t)FR_SORT_SMALL_KEYS + 128u && fr_debug_mode() != 32)
	{
		p.small_max = (uint32_t)FR_PART_MIN_DEFAULT;
		FR_AB_ONLY({ const int e = fr_env_int("FR_PART_MIN"); if (e >= 512 && e <= FR_SORT_SMALL_KEYS) p.small_max = (uint32_t)e; })
	}
	if (multi && p.tile_cap)
		hipLaunchKernelGGL(k_tile_lists, dim3((p.V * p.T + 4 * FR_THREADS - 1) / (4 * FR_THREADS)), dim3(FR_THREADS), 0, s, p.tile_cnt, p.tile_off, p.V * p.T, p.T,
		                   p.tile_cap, p.status, p.big_list, p.view_work, p.view_perm, p.ablate, p.small_max);
	else
	hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, p.tile_cnt, p.tile_off, p.tile_fill, p.V * p.T, p.T, p.V,
	                   p.key_capacity, p.status, multi ? (int*)nullptr : p.num_rendered, p.big_list, p.view_work, p.view_perm);
	if ((rc = fr_check_launch("k_scan_tiles"))) return rc;
	if (multi && p.tile_cap) { /* the projection kernel has placed the keys */ }
	else if (multi) hipLaunchKernelGGL(k_scatter_vis, gridP, dim3(FR_THREADS), 2 * hist_lds, s, p);
	else hipLaunchKernelGGL(k_scatter_keys, gridP, dim3(FR_THREADS), 2 * hist_lds, s, p);
	if ((rc = fr_check_launch("k_scatter_keys"))) return rc;
	// Single-view front end in a records mode: k_fisher_records turns the FrSplat records into the scorer's form IN PLACE, so it
	// may only start once k_scatter_keys has read their depths; it then runs beside the sorts, which touch the keys only.
	if (plan && !multi)
	{
		FrSideStream& s2 = fr_side_stream(1);
		const bool forked2 = s2.ok && hipEventRecord(s2.fork, s) == hipSuccess && hipStreamWaitEvent(s2.stream, s2.fork, 0) == hipSuccess;
		hipStream_t rs = forked2 ? s2.stream : s;
		if (plan->form_a) hipLaunchKernelGGL((k_fisher_records<4, false, true>), gridP, dim3(FR_THREADS), 0, rs, p, plan->ra);
		else if (plan->columns == 4) hipLaunchKernelGGL((k_fisher_records<4, false, false>), gridP, dim3(FR_THREADS), 0, rs, p, plan->ra);
		else hipLaunchKernelGGL((k_fisher_records<11, false, false>), gridP, dim3(FR_THREADS), 0, rs, p, plan->ra);
		if (forked2) joins.forked(&s2);
		if ((rc = fr_check_launch("k_fisher_records"))) return rc;
	}
	// The three sort tiers touch disjoint tile segments.  The 1024-thread tier has few, long-running workgroups (one per
	// CU at most), so it goes to a side stream and runs underneath the two 256-thread tiers instead of after them.
	FrSideStream& side = fr_side_stream();
	// (a handful of views: the fork / join events cost more than the overlap gains -- one view 0.478 against 0.500 ms)
	// (FR_DEBUG_MODE=7: every sort tier on the caller's stream -- timing ablation; fixed segments of 32768 keys and more: k_sort_part
	// takes every list a segment can hold twice, the 1024-thread tier is left with lists beyond 16320 keys and runs behind it)
	const bool part_all = p.tile_cap >= 32768u && fr_debug_mode() != 32;
	const bool want_fork = fr_debug_mode() != 7 && (!multi || p.V >= 8) && !part_all;
	const bool forked = want_fork && side.ok && hipEventRecord(side.fork, s) == hipSuccess && hipStreamWaitEvent(side.stream, side.fork, 0) == hipSuccess;
	const int big_blocks = p.T * p.V < 256 ? p.T * p.V : 256;
	if (forked) joins.forked(&side);
	// Fixed key segments have room behind a list (tile_capacity keys per tile): lists of up to part_max keys are partitioned into
	// wave-sized parts and sorted there (k_sort_part: after the short-list tier on the caller's stream); the two bitonic tiers keep what
	// is longer, and everything long when the lists are packed.  FR_DEBUG_MODE=32 (rig): no partitioning.
	uint32_t part_max = 0;
	if (p.tile_cap >= 2u * (uint32_t)FR_SORT_SMALL_KEYS + 128u && fr_debug_mode() != 32)
	{
		part_max = p.tile_cap / 2u - 64u;
		if (part_max > 16320u) part_max = 16320u;
	}
	hipLaunchKernelGGL(k_sort_big_tiles, dim3(big_blocks), dim3(1024), 0, forked ? side.stream : s, p, part_max);
	if ((rc = fr_check_launch("k_sort_big_tiles"))) return rc;
	const int mid_blocks = p.T * p.V < 2048 ? p.T * p.V : 2048;
	// (the middle tier on a side stream of its own as well, all three tiers at once: 2.04 ms per 64-view step against 1.95 --
	// the 256-thread tiers take each other's LDS and wave slots; k_sort_part beside the short-list tier on a stream of its own:
	// 1.504-1.520 against 1.508-1.519 behind it, profiles/r04_q_ab_sort_part.txt)
	hipLaunchKernelGGL(k_sort_tiles, dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p);
	if ((rc = fr_check_launch("k_sort_tiles"))) return rc;
	if (part_max)
	{
		hipLaunchKernelGGL(k_sort_part, dim3(mid_blocks), dim3(FR_THREADS), 0, s, p, part_max);
		if ((rc = fr_check_launch("k_sort_part"))) return rc;
	}
	if (part_max < (uint32_t)FR_SORT_MID_KEYS)
	{
		hipLaunchKernelGGL(k_sort_mid_tiles, dim3(mid_blocks), dim3(FR_THREADS), 0, s, p, part_max);
		if ((rc = fr_check_launch("k_sort_mid_tiles"))) return rc;
	}
	return joins.join();
}

static void fr_carve_single(FrParams& p, const FrLayout& L, char* geom, char* bin, char* img)
{
	p.splat = (FrSplat*)(geom + L.splat);
	p.cov3D_out = (float*)(geom + L.cov3D);
	p.rgb = (float*)(geom + L.rgb);
	p.clamped = (uint8_t*)(geom + L.clamped);
	p.tile_cnt = (uint32_t*)(img + L.tile_cnt);
	p.tile_off = (uint32_t*)(img + L.tile_off);
	p.tile_fill = (uint32_t*)(img + L.tile_fill);
	p.status = (int*)(img + L.status);
	p.big_list = (uint32_t*)(img + L.big_list);
	p.blk_base = p.T <= FR_MAX_LDS_TILES ? (uint32_t*)(img + L.blk_base) : nullptr;
	p.keys = (uint64_t*)(bin + L.keys);
}

static int fr_forward_impl(const fr_raster_cfg* cfg, const fr_gaussians* g, const float* features2,
                           void* geom_ws, void* binning_ws, int64_t binning_capacity, void* image_ws,
                           float* out_color, float* out_features2, float* out_depth, int32_t* radii, int32_t* status, fr_stream_t stream)
{
	int rc = fr_validate(cfg, g, "fr_forward");
	if (rc) return rc;
	hipStream_t s = (hipStream_t)stream;
	const int P = cfg->P, W = cfg->image_width, H = cfg->image_height;
	if (!out_color || !out_depth || !status) return fr_fail(FR_EINVAL, "fr_forward: null output");
	if (P == 0)
	{
		// rasterize_points.cu:67-81: outputs stay zero, nothing is launched
		(void)hipMemsetAsync(out_color, 0, (size_t)3 * W * H * 4, s);
		if (out_features2) (void)hipMemsetAsync(out_features2, 0, (size_t)3 * W * H * 4, s);
		(void)hipMemsetAsync(out_depth, 0, (size_t)W * H * 4, s);
		(void)hipMemsetAsync(status, 0, 16, s);
		return FR_OK;
	}
	if (!geom_ws || !binning_ws || !image_ws || !radii || binning_capacity < 0) return fr_fail(FR_EINVAL, "fr_forward: null workspace / radii");
	FrLayout L = fr_layout(P, W, H, 1, binning_capacity);
	FrParams p;
	fr_fill_params(p, cfg, g, 1);
	fr_carve_single(p, L, (char*)geom_ws, (char*)binning_ws, (char*)image_ws);
	p.radii = radii;
	p.key_capacity = binning_capacity;
	if ((rc = fr_bin_pipeline(p, g, s))) return rc;
	const float* feat = g->colors_precomp ? g->colors_precomp : p.rgb;
	FR_AB_ONLY(const bool bcast = fr_debug_mode() == 16;)        // FR_DEBUG_MODE=16: the wave-uniform compositing kernel of round 1 (A/B runs)
	float* fT = (float*)((char*)image_ws + L.final_T);
	uint32_t* nC = (uint32_t*)((char*)image_ws + L.n_contrib);
#ifdef FR_AB
	if (bcast && features2) hipLaunchKernelGGL((k_render_forward<6>), dim3(p.T, 1), dim3(FR_THREADS), 0, s, p, feat, 0, fT, nC, out_color, out_depth, features2, out_features2);
	else if (bcast) hipLaunchKernelGGL((k_render_forward<3>), dim3(p.T, 1), dim3(FR_THREADS), 0, s, p, feat, 0, fT, nC, out_color, out_depth, (const float*)nullptr, (float*)nullptr);
	else
#endif
	if (features2) hipLaunchKernelGGL((k_render_forward_walk<6>), dim3(p.T, 1), dim3(FR_THREADS), 0, s, p, feat, 0, fT, nC, out_color, out_depth, features2, out_features2);
	else hipLaunchKernelGGL((k_render_forward_walk<3>), dim3(p.T, 1), dim3(FR_THREADS), 0, s, p, feat, 0, fT, nC, out_color, out_depth, (const float*)nullptr, (float*)nullptr);
	if ((rc = fr_check_launch("k_render_forward"))) return rc;
	(void)hipMemcpyAsync(status, p.status, 16, hipMemcpyDeviceToDevice, s);
	return FR_OK;
}

extern "C" int fr_forward(const fr_raster_cfg* cfg, const fr_gaussians* g,
                          void* geom_ws, void* binning_ws, int64_t binning_capacity, void* image_ws,
                          float* out_color, float* out_depth, int32_t* radii, int32_t* status, fr_stream_t stream)
{
	return fr_forward_impl(cfg, g, nullptr, geom_ws, binning_ws, binning_capacity, image_ws, out_color, nullptr, out_depth, radii, status, stream);
}

extern "C" int fr_forward_pair(const fr_raster_cfg* cfg, const fr_gaussians* g, const float* features,
                               void* geom_ws, void* binning_ws, int64_t binning_capacity, void* image_ws,
                               float* out_color, float* out_features, float* out_depth, int32_t* radii, int32_t* status,
                               fr_stream_t stream)
{
	if (!features || !out_features) return fr_fail(FR_EINVAL, "fr_forward_pair: null features / out_features");
	return fr_forward_impl(cfg, g, features, geom_ws, binning_ws, binning_capacity, image_ws, out_color, out_features, out_depth, radii, status, stream);
}

// Scratch of the chunked power-2 backward (k_backward_sq_slots / _chunks / _prefix / _leaves).  With R tile instances and T tiles:
// L = max(256, a third of the mean list length rounded up to 64) keys per segment, at most R / L + T + 1 segment slots, L / 64 chunk
// slots per (segment, strip); per chunk slot its candidate list (512 B) and nine floats per pixel-lane (2304 B).
#define FR_SQ_SEG_MAX_TILES 16384    // (2048 x 2048; the chunked form wins at every size measured: 2.0-3.9x from 128 x 128 to 1200 x 680, tools/backward_p2_bench.py)
#define FR_SQ_MAX_SCRATCH (4ull << 30)
struct FrSqScratch { size_t slot_map, ctl, cnt, work, ctodo, list, pixmask, summ, bytes; uint32_t n_slots, n_chunks, num_rendered; };
static FrSqScratch fr_sq_scratch(int64_t T, int64_t R, int nch = 3)
{
	FrSqScratch q;
	const uint32_t L = fr_sq_seg_length_of((uint32_t)(R > 0 ? R : 1), (uint32_t)T);
	q.num_rendered = (uint32_t)(R > 0 ? R : 1);
	q.n_slots = (uint32_t)((R > 0 ? R : 1) / L + T + 2);
	q.n_chunks = q.n_slots * 4u * (L / 64u);
	size_t o = 0;
	q.slot_map = o; o = fr_align(o + (size_t)q.n_slots * 4);
	q.ctl = o; o = fr_align(o + 64);                      // (zeroed together with the slot map: fr_backward_ws)
	q.cnt = o; o = fr_align(o + (size_t)q.n_slots * 16);
	q.work = o; o = fr_align(o + (size_t)q.n_chunks * 2 * sizeof(uint2));
	q.ctodo = o; o = fr_align(o + (size_t)q.n_chunks * sizeof(uint2));
	q.pixmask = o; o = fr_align(o + (size_t)q.n_chunks * 64 * sizeof(uint2));
	q.list = o; o = fr_align(o + (size_t)q.n_chunks * 64 * sizeof(uint2));
	q.summ = o; o = fr_align(o + (size_t)q.n_chunks * FR_BWD_MAP_FLOATS(nch) * 64 * sizeof(float));
	q.bytes = o;
	return q;
}
extern "C" size_t fr_backward_scratch_bytes(int32_t P, int32_t W, int32_t H, int32_t power, int64_t num_rendered)
{
	if (P <= 0 || W <= 0 || H <= 0 || (power != 1 && power != 2) || num_rendered <= 0 || num_rendered > 0x7fffffffll) return 0;
	const int64_t T = ((int64_t)(W + 15) / 16) * ((int64_t)(H + 15) / 16);
	if (T > FR_SQ_SEG_MAX_TILES) return 0;
	const size_t bytes = fr_sq_scratch(T, num_rendered).bytes;
	return bytes <= FR_SQ_MAX_SCRATCH ? bytes : 0;
}
extern "C" size_t fr_backward_pair_scratch_bytes(int32_t P, int32_t W, int32_t H, int64_t num_rendered)
{
	if (P <= 0 || W <= 0 || H <= 0 || num_rendered <= 0 || num_rendered > 0x7fffffffll) return 0;
	const int64_t T = ((int64_t)(W + 15) / 16) * ((int64_t)(H + 15) / 16);
	if (T > FR_SQ_SEG_MAX_TILES) return 0;
	const size_t bytes = fr_sq_scratch(T, num_rendered, 6).bytes;
	return bytes <= FR_SQ_MAX_SCRATCH ? bytes : 0;
}

// The chunked backward (k_backward_sq_slots ... _leaves) on one view; rows: the leaf rows of power 2, null for power 1.  The kernels
// size their segments from status[0]; num_rendered must be that number: the scratch layout (sq) depends on it.
template <int NCH>
static int fr_launch_chunked(const FrParams& p, const FrBwdArgs& b, const FrSqScratch& sq, void* scratch, const float* rows, hipStream_t s)
{
	FrSqSegArgs sg;
	char* sc = (char*)scratch;
	sg.slot_map = (uint32_t*)(sc + sq.slot_map); sg.ctl = (uint32_t*)(sc + sq.ctl); sg.cnt = (uint32_t*)(sc + sq.cnt);
	sg.work = (uint2*)(sc + sq.work); sg.ctodo = (uint2*)(sc + sq.ctodo); sg.pixmask = (uint2*)(sc + sq.pixmask);
	sg.list = (uint2*)(sc + sq.list); sg.summ = (float*)(sc + sq.summ);
	sg.n_slots = sq.n_slots; sg.n_chunks = sq.n_chunks; sg.num_rendered = sq.num_rendered;
	const dim3 block(FR_THREADS);
	int rc;
	hipLaunchKernelGGL(k_backward_sq_slots, dim3(p.T), block, 0, s, p, b, sg);
	if ((rc = fr_check_launch("k_backward_sq_slots"))) return rc;
	hipLaunchKernelGGL((k_backward_sq_chunks<NCH>), dim3(sg.n_slots), block, 0, s, p, b, sg);
	if ((rc = fr_check_launch("k_backward_sq_chunks"))) return rc;
	hipLaunchKernelGGL((k_backward_sq_prefix<NCH>), dim3(p.T), block, 0, s, p, b, sg);
	if ((rc = fr_check_launch("k_backward_sq_prefix"))) return rc;
	// power 2: four workgroups per CU (34 KiB of LDS each; 512 / 768 workgroups: 230 / 205 us against 190 on the benchmark room)
	const unsigned cap = rows ? 1024u : 2048u;
	const unsigned grid = sg.n_chunks / 4u + 1u < cap ? sg.n_chunks / 4u + 1u : cap;
	if constexpr (NCH == 3)
	{
		if (rows) hipLaunchKernelGGL((k_backward_sq_leaves<2, 3>), dim3(grid), block, 0, s, p, b, rows, sg);
		else hipLaunchKernelGGL((k_backward_sq_leaves<1, 3>), dim3(grid), block, 0, s, p, b, rows, sg);
	}
	else hipLaunchKernelGGL((k_backward_sq_leaves<1, NCH>), dim3(grid), block, 0, s, p, b, rows, sg);
	return fr_check_launch("k_backward_sq_leaves");
}

extern "C" int fr_backward(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                           const void* geom_ws, const void* binning_ws, const void* image_ws,
                           const float* dL_dout_color, int32_t power,
                           float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D,
                           float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dconic,
                           fr_stream_t stream)
{
	return fr_backward_ws(cfg, g, radii, geom_ws, binning_ws, image_ws, dL_dout_color, power, dL_dmeans2D, dL_dcolors, dL_dopacity,
	                      dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dconic, 0, nullptr, 0, stream);
}

extern "C" int fr_backward_ws(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                              const void* geom_ws, const void* binning_ws, const void* image_ws,
                              const float* dL_dout_color, int32_t power,
                              float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D,
                              float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dconic,
                              int64_t num_rendered, void* scratch, size_t scratch_bytes, fr_stream_t stream)
{
	int rc = fr_validate(cfg, g, "fr_backward", false);
	if (rc) return rc;
	hipStream_t s = (hipStream_t)stream;
	const int P = cfg->P, W = cfg->image_width, H = cfg->image_height;
	if (P == 0) return FR_OK;
	if (!geom_ws || !binning_ws || !image_ws || !radii || !dL_dout_color || !dL_dmeans2D || !dL_dcolors || !dL_dopacity ||
	    !dL_dmeans3D || !dL_dcov3D || !dL_dscales || !dL_drotations || !dL_dconic)
		return fr_fail(FR_EINVAL, "fr_backward: null pointer");
	if (g->shs && !dL_dsh) return fr_fail(FR_EINVAL, "fr_backward: dL_dsh is null although SHs were given");
	const bool sq_rows_path = power == 2 && g->scales && !g->shs && g->colors_precomp && !g->cov3D_precomp && fr_debug_mode() != 21;
	const int64_t T_img = ((int64_t)(W + 15) / 16) * ((int64_t)(H + 15) / 16);
	// few tiles and a scratch buffer: the chunked form (FR_DEBUG_MODE=33 in the rig: one workgroup per tile whatever the scratch)
	const bool chunk_path = sq_rows_path || (power == 1 FR_AB_ONLY(&& fr_debug_mode() != 17));
	const size_t scratch_need = chunk_path ? fr_backward_scratch_bytes(P, W, H, power, num_rendered) : 0;
	const bool segmented = chunk_path && scratch && scratch_need > 0 && scratch_bytes >= scratch_need FR_AB_ONLY(&& fr_debug_mode() != 33);
	const FrSqScratch sq = fr_sq_scratch(T_img, segmented ? num_rendered : 1);
	{
		FrZeroer z;
		z.add(dL_dmeans2D, (size_t)P * 3 * 4); z.add(dL_dcolors, (size_t)P * 3 * 4); z.add(dL_dopacity, (size_t)P * 4);
		z.add(dL_dmeans3D, (size_t)P * 3 * 4); z.add(dL_dcov3D, (size_t)P * 6 * 4); z.add(dL_dscales, (size_t)P * 3 * 4);
		z.add(dL_drotations, (size_t)P * 4 * 4); z.add(dL_dconic, (size_t)P * 4 * 4);
		if (dL_dsh && cfg->sh_coeffs > 0) z.add(dL_dsh, (size_t)P * cfg->sh_coeffs * 3 * 4);
		if (segmented) z.add((char*)scratch + sq.slot_map, sq.cnt - sq.slot_map);      // the slot map and the work list's counters
		z.launch(s);
	}

	FrLayout L = fr_layout(P, W, H, 1, 1);
	FrParams p;
	fr_fill_params(p, cfg, g, 1);
	fr_carve_single(p, L, (char*)geom_ws, (char*)binning_ws, (char*)image_ws);
	p.radii = (int*)radii;
	p.cov3D = g->cov3D_precomp ? g->cov3D_precomp : p.cov3D_out;
	FrBwdArgs b;
	b.dL_dpix = dL_dout_color;
	b.final_T = (const float*)((const char*)image_ws + L.final_T);
	b.n_contrib = (const uint32_t*)((const char*)image_ws + L.n_contrib);
	b.colors = g->colors_precomp ? g->colors_precomp : p.rgb;
	b.power = power;
	b.dL_dmean2D = dL_dmeans2D; b.dL_dconic = dL_dconic; b.dL_dopacity = dL_dopacity; b.dL_dcolors = dL_dcolors;
	b.dL_dmean3D = dL_dmeans3D; b.dL_dcov3D = dL_dcov3D; b.dL_dscale = dL_dscales; b.dL_drot = dL_drotations;
	const bool sr = g->scales != nullptr, sh = g->shs != nullptr;
	dim3 grid(p.T, 1), block(FR_THREADS);
	b.only_flagged = nullptr;
	b.u_only = 0;
	b.dL_dmean2D_b = nullptr;
	if (power == 1)
	{
		// gradients: sum u per splat in the tile kernel, Jacobian chain once per Gaussian; tiles that do not fit the LDS
		// index are redone (u only) by the scan kernel.  The flag array borrows tile_fill, which is dead after binning.
		// FR_DEBUG_MODE=17: the two-pass tile kernel of round 1 with its fallback pass (A/B runs); k_backward_lin_walk has no list
		// capacity and therefore no fallback tiles
#ifdef FR_AB
		if (fr_debug_mode() == 17)
		{
			uint8_t* fallback = (uint8_t*)p.tile_fill;
			hipLaunchKernelGGL((k_backward_lin_tile<false>), dim3(p.T), block, 0, s, p, b, fallback);
			if ((rc = fr_check_launch("k_backward_lin_tile"))) return rc;
			b.only_flagged = fallback;
			b.u_only = 1;
			if (sr && sh) hipLaunchKernelGGL((k_backward_tile<true, true>), grid, block, 0, s, p, b);
			else if (sr) hipLaunchKernelGGL((k_backward_tile<true, false>), grid, block, 0, s, p, b);
			else if (sh) hipLaunchKernelGGL((k_backward_tile<false, true>), grid, block, 0, s, p, b);
			else hipLaunchKernelGGL((k_backward_tile<false, false>), grid, block, 0, s, p, b);
			if ((rc = fr_check_launch("k_backward_tile(flagged)"))) return rc;
		}
		else
#endif
		if (segmented)
		{
			if ((rc = fr_launch_chunked<3>(p, b, sq, scratch, nullptr, s))) return rc;
			b.u_only = 1;
		}
		else
		{
			hipLaunchKernelGGL((k_backward_lin_walk<false>), dim3(p.T), block, 0, s, p, b);
			if ((rc = fr_check_launch("k_backward_lin_walk"))) return rc;
			b.u_only = 1;
		}
		dim3 gp((P + FR_THREADS - 1) / FR_THREADS);
		if (sr && sh) hipLaunchKernelGGL((k_backward_finish<true, true>), gp, block, 0, s, p, b, dL_dsh);
		else if (sr) hipLaunchKernelGGL((k_backward_finish<true, false>), gp, block, 0, s, p, b, dL_dsh);
		else if (sh) hipLaunchKernelGGL((k_backward_finish<false, true>), gp, block, 0, s, p, b, dL_dsh);
		else hipLaunchKernelGGL((k_backward_finish<false, false>), gp, block, 0, s, p, b, dL_dsh);
		return fr_check_launch("k_backward_finish");
	}
	if (sq_rows_path)
	{
		// The diagonal Fisher proxy as the reference's own loop asks for it (gaussian.py:1536-1556: one view, autograd, power 2):
		// the leaf rows once per visible Gaussian (k_backward_sq_rows), then the walking tile kernel (k_backward_sq_walk).
		// FR_DEBUG_MODE=21 keeps round 2's all-leaves tile kernel (k_fisher_tile_v2<25>) for A/B runs.
		float* rows = (float*)((char*)geom_ws + L.packed);           // [P] x 56 floats at 256-byte steps (FR_SQ_ROW_STRIDE) in the geometry buffer's per-Gaussian region
		hipLaunchKernelGGL(k_backward_sq_rows, dim3((P + FR_THREADS - 1) / FR_THREADS), block, 0, s, p, rows);
		if ((rc = fr_check_launch("k_backward_sq_rows"))) return rc;
		if (segmented) return fr_launch_chunked<3>(p, b, sq, scratch, (const float*)rows, s);
		hipLaunchKernelGGL(k_backward_sq_walk, dim3(p.T), block, 0, s, p, b, (const float*)rows);
		return fr_check_launch("k_backward_sq_walk");
	}
#ifdef FR_AB
	if (power == 2 && sr && !sh && g->colors_precomp && !g->cov3D_precomp)
	{
		// The diagonal Fisher proxy as the reference's own loop asks for it (gaussian.py:1536-1556: one view, autograd, power 2):
		// the wave-private two-pass kernel of the batched scorer with all 25 leaves, the upstream gradient as a per-pixel image;
		// tiles that do not fit its LDS index are redone by the generic kernel below.
		float* packed = (float*)((char*)geom_ws + L.packed);
		uint8_t* fallback = (uint8_t*)p.tile_fill;
		FrParams pp = p;
		pp.colors = g->colors_precomp;
		hipLaunchKernelGGL((k_pack_static<25>), dim3((P + FR_THREADS - 1) / FR_THREADS), block, 0, s, pp, (const float*)nullptr, packed, (float4*)nullptr, (float4*)nullptr);
		FrFisherArgs f;
		memset(&f, 0, sizeof(f));
		f.dL_img = dL_dout_color; f.dL_stride = 0;
		f.full_out[0] = dL_dmeans3D; f.full_out[1] = dL_dopacity; f.full_out[2] = dL_dscales; f.full_out[3] = dL_drotations;
		f.full_out[4] = dL_dcolors; f.full_out[5] = dL_dmeans2D; f.full_out[6] = dL_dcov3D; f.full_out[7] = dL_dconic;
		hipLaunchKernelGGL((k_fisher_tile_v2<25, false, true>), dim3(p.T), block, 0, s, pp, f, (const float*)packed, fallback);
		if ((rc = fr_check_launch("k_fisher_tile_v2<25>"))) return rc;
		b.only_flagged = fallback;
		hipLaunchKernelGGL((k_backward_tile<true, false>), grid, block, 0, s, p, b);
		return fr_check_launch("k_backward_tile(flagged)");
	}
#endif
	if (sr && sh) hipLaunchKernelGGL((k_backward_tile<true, true>), grid, block, 0, s, p, b);
	else if (sr) hipLaunchKernelGGL((k_backward_tile<true, false>), grid, block, 0, s, p, b);
	else if (sh) hipLaunchKernelGGL((k_backward_tile<false, true>), grid, block, 0, s, p, b);
	else hipLaunchKernelGGL((k_backward_tile<false, false>), grid, block, 0, s, p, b);
	if ((rc = fr_check_launch("k_backward_tile"))) return rc;
	if (sh)
	{
		hipLaunchKernelGGL(k_finish_sh, dim3((P + FR_THREADS - 1) / FR_THREADS), block, 0, s, p, (const float*)dL_dcolors, (int)power, dL_dsh);
		if ((rc = fr_check_launch("k_finish_sh"))) return rc;
	}
	return FR_OK;
}

// ---- fused Fisher scorer ---------------------------------------------------------------------------------
// ---- a second feature image on the geometry fr_forward binned (models/SLAM/gaussian.py:203-211) -------------------
extern "C" int fr_forward_features(const fr_raster_cfg* cfg, const float* features,
                                   const void* geom_ws, const void* binning_ws, void* image_ws,
                                   float* out_features, fr_stream_t stream)
{
	if (!cfg || cfg->P < 0 || cfg->image_width <= 0 || cfg->image_height <= 0) return fr_fail(FR_EINVAL, "fr_forward_features: bad cfg");
	hipStream_t s = (hipStream_t)stream;
	const int P = cfg->P, W = cfg->image_width, H = cfg->image_height;
	if (!out_features) return fr_fail(FR_EINVAL, "fr_forward_features: null output");
	if (P == 0)
	{
		(void)hipMemsetAsync(out_features, 0, (size_t)3 * W * H * 4, s);
		return FR_OK;
	}
	if (!features || !geom_ws || !binning_ws || !image_ws || !cfg->bg) return fr_fail(FR_EINVAL, "fr_forward_features: null pointer");
	FrLayout L = fr_layout(P, W, H, 1, 1);
	fr_gaussians g0;
	memset(&g0, 0, sizeof(g0));
	FrParams p;
	fr_fill_params(p, cfg, &g0, 1);
	fr_carve_single(p, L, (char*)geom_ws, (char*)binning_ws, (char*)image_ws);
#ifdef FR_AB
	if (fr_debug_mode() == 16)
		hipLaunchKernelGGL((k_render_forward<3>), dim3(p.T, 1), dim3(FR_THREADS), 0, s, p, features, 0,
		                   (float*)((char*)image_ws + L.final_T), (uint32_t*)((char*)image_ws + L.n_contrib), out_features, (float*)nullptr,
		                   (const float*)nullptr, (float*)nullptr);
	else
#endif
		hipLaunchKernelGGL((k_render_forward_walk<3>), dim3(p.T, 1), dim3(FR_THREADS), 0, s, p, features, 0,
		                   (float*)((char*)image_ws + L.final_T), (uint32_t*)((char*)image_ws + L.n_contrib), out_features, (float*)nullptr,
		                   (const float*)nullptr, (float*)nullptr);
	return fr_check_launch("k_render_forward(features)");
}

extern "C" int fr_backward_pair(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                                const void* geom_ws, const void* binning_ws, const void* image_ws,
                                const float* dL_dout_color, const float* features, const float* dL_dout_features,
                                float* dL_dmeans2D, float* dL_dmeans2D_features, float* dL_dcolors, float* dL_dfeatures,
                                float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales,
                                float* dL_drotations, float* dL_dconic, fr_stream_t stream)
{
	return fr_backward_pair_ws(cfg, g, radii, geom_ws, binning_ws, image_ws, dL_dout_color, features, dL_dout_features, dL_dmeans2D,
	                           dL_dmeans2D_features, dL_dcolors, dL_dfeatures, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales, dL_drotations,
	                           dL_dconic, 0, nullptr, 0, stream);
}

extern "C" int fr_backward_pair_ws(const fr_raster_cfg* cfg, const fr_gaussians* g, const int32_t* radii,
                                   const void* geom_ws, const void* binning_ws, const void* image_ws,
                                   const float* dL_dout_color, const float* features, const float* dL_dout_features,
                                   float* dL_dmeans2D, float* dL_dmeans2D_features, float* dL_dcolors, float* dL_dfeatures,
                                   float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dscales,
                                   float* dL_drotations, float* dL_dconic,
                                   int64_t num_rendered, void* scratch, size_t scratch_bytes, fr_stream_t stream)
{
	int rc = fr_validate(cfg, g, "fr_backward_pair", false);
	if (rc) return rc;
	hipStream_t s = (hipStream_t)stream;
	const int P = cfg->P, W = cfg->image_width, H = cfg->image_height;
	if (P == 0) return FR_OK;
	if (!g->colors_precomp || g->shs) return fr_fail(FR_EINVAL, "fr_backward_pair: needs colors_precomp (no SH)");
	if (!geom_ws || !binning_ws || !image_ws || !radii || !dL_dout_color || !features || !dL_dout_features || !dL_dmeans2D ||
	    !dL_dmeans2D_features || !dL_dcolors || !dL_dfeatures || !dL_dopacity || !dL_dmeans3D || !dL_dcov3D || !dL_dscales ||
	    !dL_drotations || !dL_dconic)
		return fr_fail(FR_EINVAL, "fr_backward_pair: null pointer");
	// with a scratch buffer: the chunked form on six colour channels (FR_DEBUG_MODE=33 in the rig: never)
	const int64_t T_img = ((int64_t)(W + 15) / 16) * ((int64_t)(H + 15) / 16);
	const size_t scratch_need = fr_backward_pair_scratch_bytes(P, W, H, num_rendered);
	const bool chunked = scratch && scratch_need > 0 && scratch_bytes >= scratch_need FR_AB_ONLY(&& fr_debug_mode() != 33 && fr_debug_mode() != 18);
	const FrSqScratch sq = fr_sq_scratch(T_img, chunked ? num_rendered : 1, 6);
	{
		FrZeroer z;
		z.add(dL_dmeans2D, (size_t)P * 3 * 4); z.add(dL_dmeans2D_features, (size_t)P * 3 * 4); z.add(dL_dcolors, (size_t)P * 3 * 4);
		z.add(dL_dfeatures, (size_t)P * 3 * 4); z.add(dL_dopacity, (size_t)P * 4); z.add(dL_dmeans3D, (size_t)P * 3 * 4);
		z.add(dL_dcov3D, (size_t)P * 6 * 4); z.add(dL_dscales, (size_t)P * 3 * 4); z.add(dL_drotations, (size_t)P * 4 * 4);
		z.add(dL_dconic, (size_t)P * 4 * 4);
		if (chunked) z.add((char*)scratch + sq.slot_map, sq.cnt - sq.slot_map);       // the slot map and the work list's counters
		z.launch(s);
	}
	FrLayout L = fr_layout(P, W, H, 1, 1);
	FrParams p;
	fr_fill_params(p, cfg, g, 1);
	fr_carve_single(p, L, (char*)geom_ws, (char*)binning_ws, (char*)image_ws);
	p.radii = (int*)radii;
	p.cov3D = g->cov3D_precomp ? g->cov3D_precomp : p.cov3D_out;
	const bool sr = g->scales != nullptr;
	dim3 grid(p.T, 1), block(FR_THREADS);
	uint8_t* fallback = (uint8_t*)p.tile_fill;
	FrBwdArgs b;
	b.final_T = (const float*)((const char*)image_ws + L.final_T);
	b.n_contrib = (const uint32_t*)((const char*)image_ws + L.n_contrib);
	b.power = 1;
	b.dL_dconic = dL_dconic; b.dL_dopacity = dL_dopacity;
	b.dL_dmean3D = dL_dmeans3D; b.dL_dcov3D = dL_dcov3D; b.dL_dscale = dL_dscales; b.dL_drot = dL_drotations;
	b.dL_dmean2D_b = nullptr;
	// one tile pass for both images (shared transmittance pass, lists and conic / opacity sums)
	b.dL_dpix = dL_dout_color; b.colors = g->colors_precomp; b.dL_dmean2D = dL_dmeans2D; b.dL_dcolors = dL_dcolors;
	b.dL_dpix2 = dL_dout_features; b.colors2 = features; b.dL_dmean2D_2 = dL_dmeans2D_features; b.dL_dcolors2 = dL_dfeatures;
	b.only_flagged = nullptr; b.u_only = 0;
	// (the pair keeps the two-pass tile kernel: with fourteen accumulators per candidate the walk form measured 0.83 against
	// 0.79 ms at 2M Gaussians / 512 x 512 -- the single image gains, 0.49 against 0.59; FR_DEBUG_MODE=18 forces the walk form)
	const bool pair_walk = fr_debug_mode() == 18 || chunked;
	if (chunked)
	{
		if ((rc = fr_launch_chunked<6>(p, b, sq, scratch, nullptr, s))) return rc;
	}
	else
#ifdef FR_AB
	if (pair_walk)
	{
		hipLaunchKernelGGL((k_backward_lin_walk<true>), dim3(p.T), block, 0, s, p, b);
		if ((rc = fr_check_launch("k_backward_lin_walk<pair>"))) return rc;
	}
	else
#endif
	hipLaunchKernelGGL((k_backward_lin_tile<true>), dim3(p.T), block, 0, s, p, b, fallback);
	if ((rc = fr_check_launch("k_backward_lin_tile<pair>"))) return rc;
	// (two-pass kernel only) tiles whose lists do not fit the LDS index: the scan kernel, once per image
	for (int pass = 0; pass < 2 && !pair_walk; pass++)
	{
		b.dL_dpix = pass ? dL_dout_features : dL_dout_color;
		b.colors = pass ? features : g->colors_precomp;
		b.dL_dmean2D = pass ? dL_dmeans2D_features : dL_dmeans2D;
		b.dL_dcolors = pass ? dL_dfeatures : dL_dcolors;
		b.only_flagged = fallback; b.u_only = 1;
		if (sr) hipLaunchKernelGGL((k_backward_tile<true, false>), grid, block, 0, s, p, b);
		else hipLaunchKernelGGL((k_backward_tile<false, false>), grid, block, 0, s, p, b);
		if ((rc = fr_check_launch("k_backward_tile(flagged)"))) return rc;
	}
	// Jacobian chain once per Gaussian on the summed screen-space gradients
	b.dL_dmean2D = dL_dmeans2D; b.dL_dmean2D_b = dL_dmeans2D_features;
	b.dL_dcolors = dL_dcolors; b.colors = g->colors_precomp;
	dim3 gp((P + FR_THREADS - 1) / FR_THREADS);
	if (sr) hipLaunchKernelGGL((k_backward_finish<true, false>), gp, block, 0, s, p, b, (float*)nullptr);
	else hipLaunchKernelGGL((k_backward_finish<false, false>), gp, block, 0, s, p, b, (float*)nullptr);
	return fr_check_launch("k_backward_finish");
}

// FR_DEBUG_MODE (timing ablations / loop statistics only), read once per process:
//   1  k_fisher_tile_v2 stops after its first pass            6  the LDS-resident sort network of round 1
//   7  every sort tier on the caller's stream (no fork)       8  8 x 8 pixel blocks per wave in k_fisher_tile_v3
//   9  the second-generation two-pass kernels instead of k_fisher_tile_v3 / _v3h
//   2-7, 10-12 in a -DFR_LOOPSTATS build: loop-trip counters and s_memtime shares (tools/loopstats.py)
// FR_GV / FR_VC (read once as well): Gaussians per thread / views per workgroup of k_preprocess_views.
#ifdef FR_AB
static int fr_debug_mode()
{
	static const int mode = fr_env_int("FR_DEBUG_MODE");
	return mode;
}
#endif

#define FR_MAX_GROUPS 4              // view groups of one fr_fisher_views call (fr_pick_groups)
struct FrFisherLayout {
	size_t radii, vis_n, splat, recq, slot_idx, packed, mt, grp, big_list, view_work, view_perm, blk_base, cov3D, tile_cnt, tile_off, tile_fill, tile_scores, w2c_inv, status, keys, fallback, seg_T, seg_C, seg_X, seg_list, total;
	size_t PV;                   // slots per view of the compact records: projection workgroups * 256 G (>= P)
};
static FrFisherLayout fr_fisher_layout(int64_t P, int64_t W, int64_t H, int64_t V, int64_t max_rendered, int columns)
{
	FrFisherLayout L;
	const int64_t T = ((W + 15) / 16) * ((H + 15) / 16);
	size_t o = 0;
	// radii [V][P] int32 (single-pass front end, images beyond FR_MAX_LDS_TILES tiles) and the compact visible lists
	// [V][blocks][256 G] x 16 B of the multi-view front end share one region: a launch runs one or the other
	const size_t nblk_v = (size_t)((P + FR_THREADS * fr_pick_G_views(P) - 1) / (FR_THREADS * fr_pick_G_views(P)));
	L.radii = o; o = fr_align(o + (size_t)V * nblk_v * (size_t)(FR_THREADS * fr_pick_G_views(P)) * sizeof(FrVisEntry));
	L.vis_n = o; o = fr_align(o + (size_t)V * nblk_v * 4);
	// [V][P] 32-byte splat records; with compact records the same region is [V][workgroup][256 G] (phase B parks {recA, recB} there)
	L.PV = nblk_v * (size_t)(FR_THREADS * fr_pick_G_views(P));
	const size_t VPV = (size_t)V * (L.PV > (size_t)P ? L.PV : (size_t)P);
	L.splat = o; o = fr_align(o + VPV * sizeof(FrSplat));
	// dense: [V][P] x 64 B {12 polynomial coefficients + k3, or A'[15], 1/o^2}; compact: [V][PV] x 96 B {recA, recB, the same}
	L.recq = o; o = fr_align(o + VPV * (columns == 11 ? 208 : 112));     // compact: 96 B (score form, A-form), 112 / 208 B (general out_H form, 4 / 11 columns)
	L.slot_idx = o; o = fr_align(o + VPV * 4);
	L.packed = o; o = fr_align(o + (size_t)P * 4 * (size_t)(columns == 11 ? 32 : 16));
	L.mt = o; o = fr_align(o + (size_t)P * 16);
	L.grp = o; o = fr_align(o + (size_t)((P + FR_THREADS - 1) / FR_THREADS) * 32);
	L.big_list = o; o = fr_align(o + (size_t)(V * T) * 4 + 64 * FR_MAX_GROUPS);      // one {count, pad[15], list} per view group
	L.view_work = o; o = fr_align(o + (size_t)V * 4);
	L.view_perm = o; o = fr_align(o + (size_t)V * 4);
	L.blk_base = o; o = fr_align(o + (size_t)V * (size_t)fr_preprocess_blocks(P, V) * (size_t)T * 4);
	L.cov3D = o; o = fr_align(o + (size_t)P * 24);
	L.tile_cnt = o; o = fr_align(o + (size_t)(V * T) * 4);
	L.tile_off = o; o = fr_align(o + (size_t)(V * T) * 4);
	L.tile_fill = o; o = fr_align(o + (size_t)(V * T) * 4);
	L.tile_scores = o; o = fr_align(o + (size_t)(V * T) * 4);
	L.w2c_inv = o; o = fr_align(o + (size_t)V * 64);
	L.status = o; o = fr_align(o + 64);
	const size_t R = (size_t)(max_rendered > 0 ? max_rendered : 1);
	L.keys = o; o = fr_align(o + R * 8);
	L.fallback = o; o = fr_align(o + (size_t)(V * T));
	// out_H launches of few views (k_fisher_tile_v3h<.., 1 / 2>): the pixels' state at the segment boundaries of their tile lists
	const size_t seg_tiles = (size_t)(V * T <= FR_SEG_TILES ? V * T : 0);
	L.seg_T = o; o = fr_align(o + FR_SEG_PER_TILE * seg_tiles * FR_THREADS * 4);
	L.seg_C = o; o = fr_align(o + FR_SEG_PER_TILE * seg_tiles * FR_THREADS * 8);
	L.seg_X = o; o = fr_align(o + seg_tiles * FR_THREADS * 8);
	L.seg_list = o; o = fr_align(o + (FR_SEG_PER_TILE * seg_tiles + 1) * 4);
	L.total = o;
	return L;
}

extern "C" int fr_fisher_workspace_layout(int32_t P, int32_t W, int32_t H, int32_t n_views, int64_t max_rendered, int32_t columns, size_t o[8])
{
	if (P < 0 || W <= 0 || H <= 0 || n_views <= 0 || max_rendered < 0 || (columns != 4 && columns != 11) || !o)
		return fr_fail(FR_EINVAL, "fr_fisher_workspace_layout: bad argument");
	const FrFisherLayout L = fr_fisher_layout(P, W, H, n_views, max_rendered, columns);
	o[0] = L.tile_cnt; o[1] = L.tile_off; o[2] = L.keys; o[3] = L.splat; o[4] = L.recq; o[5] = L.tile_scores; o[6] = L.status; o[7] = L.vis_n;
	return FR_OK;
}

extern "C" int fr_init(void)
{
	if (!fr_side_stream(0).ok || !fr_side_stream(1).ok || !fr_side_stream(2).ok) return fr_fail(FR_ELAUNCH, "fr_init: could not create the side streams");
	return FR_OK;
}

extern "C" size_t fr_fisher_workspace_bytes(int32_t P, int32_t W, int32_t H, int32_t n_views, int64_t max_rendered, int32_t columns)
{
	if (P < 0 || W <= 0 || H <= 0 || n_views <= 0 || max_rendered < 0 || (columns != 4 && columns != 11)) return 0;
	return fr_fisher_layout(P, W, H, n_views, max_rendered, columns).total;
}

// score-only mode: the single front-to-back pass over the records (no capacity limit, no fallback kernel)
static void fr_launch_fisher_v3(FrParams& p, FrFisherArgs f, float4* recq, hipStream_t s)
{
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if (g_prof_on)
	{
		(void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
		(void)hipEventRecord(ev0, s);
	}
	// FR_DEBUG_MODE=8: 8 x 8 pixel blocks per wave instead of 16 x 4 strips (A/B runs; measured 5 % slower on MI355X);
	// FR_DEBUG_MODE=24: the rolling two-chunk window (k_fisher_tile_v3w: 24 % fewer walk iterations, 8 % slower -- see there)
#ifdef FR_AB
	if (f.debug_mode == 8) hipLaunchKernelGGL((k_fisher_tile_v3<8, 8>), dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p, f, (const float4*)recq);
	else if (f.debug_mode == 24) hipLaunchKernelGGL(k_fisher_tile_v3w, dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p, f);
	else if (f.key_shift && f.debug_mode == 31) hipLaunchKernelGGL((k_fisher_tile_v3<16, 4, true>), dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p, f, (const float4*)recq);   // FR_DEBUG_MODE=31: the chunk-synchronous walk of round 3
	else
#endif
	if (f.key_shift) hipLaunchKernelGGL(k_fisher_tile_v4, dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p, f);
	else hipLaunchKernelGGL((k_fisher_tile_v3<16, 4>), dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p, f, (const float4*)recq);
	if (g_prof_on)
	{
		(void)hipEventRecord(ev1, s);
		g_prof_events.push_back(std::make_pair(ev0, ev1));
	}
}

// out_H mode with 4 columns and a constant upstream gradient: two front-to-back passes over the records
static void fr_launch_fisher_v3h(FrParams& p, FrFisherArgs f, float4* recq, FrSegArgs sg, hipStream_t s)
{
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if (g_prof_on)
	{
		(void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
		(void)hipEventRecord(ev0, s);
	}
	if (sg.snapT && p.T * p.V <= FR_SEG_TILES)
	{
		// few views: pass 1 per tile with the pixels' state saved at every segment boundary, pass 2 per (tile, segment)
		(void)hipMemsetAsync(sg.list, 0, 4, s);
		hipLaunchKernelGGL((k_fisher_tile_v3h<16, 4, 1>), dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p, f, (const float4*)recq, sg);
		hipLaunchKernelGGL((k_fisher_tile_v3h<16, 4, 2>), dim3(FR_SEG_PER_TILE * p.T * p.V), dim3(FR_THREADS), 0, s, p, f, (const float4*)recq, sg);
	}
	else
	hipLaunchKernelGGL((k_fisher_tile_v3h<16, 4, 0>), dim3(p.T * p.V), dim3(FR_THREADS), 0, s, p, f, (const float4*)recq, sg);
	if (g_prof_on)
	{
		(void)hipEventRecord(ev1, s);
		g_prof_events.push_back(std::make_pair(ev0, ev1));
	}
}

// the other out_H modes on records: 11 columns and / or an upstream-gradient image
static void fr_launch_fisher_v3g(FrParams& p, FrFisherArgs f, int columns, bool img, FrSegArgs sg, hipStream_t s)
{
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if (g_prof_on)
	{
		(void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
		(void)hipEventRecord(ev0, s);
	}
	dim3 grid(p.T * p.V), block(FR_THREADS);
	if (sg.snapT && p.T * p.V <= FR_SEG_TILES)
	{
		// few views (GaussianObjectSLAM.compute_Hessian of one pose, a handful of probes): the segmented passes, as for k_fisher_tile_v3h
		(void)hipMemsetAsync(sg.list, 0, 4, s);
		dim3 g2(FR_SEG_PER_TILE * p.T * p.V);
		if (columns == 11 && img) { hipLaunchKernelGGL((k_fisher_tile_v3g<11, true, 1>), grid, block, 0, s, p, f, sg); hipLaunchKernelGGL((k_fisher_tile_v3g<11, true, 2>), g2, block, 0, s, p, f, sg); }
		else if (columns == 11) { hipLaunchKernelGGL((k_fisher_tile_v3g<11, false, 1>), grid, block, 0, s, p, f, sg); hipLaunchKernelGGL((k_fisher_tile_v3g<11, false, 2>), g2, block, 0, s, p, f, sg); }
		else if (img) { hipLaunchKernelGGL((k_fisher_tile_v3g<4, true, 1>), grid, block, 0, s, p, f, sg); hipLaunchKernelGGL((k_fisher_tile_v3g<4, true, 2>), g2, block, 0, s, p, f, sg); }
		else { hipLaunchKernelGGL((k_fisher_tile_v3g<4, false, 1>), grid, block, 0, s, p, f, sg); hipLaunchKernelGGL((k_fisher_tile_v3g<4, false, 2>), g2, block, 0, s, p, f, sg); }
	}
	else if (columns == 11 && img) hipLaunchKernelGGL((k_fisher_tile_v3g<11, true, 0>), grid, block, 0, s, p, f, sg);
	else if (columns == 11) hipLaunchKernelGGL((k_fisher_tile_v3g<11, false, 0>), grid, block, 0, s, p, f, sg);
	else if (img) hipLaunchKernelGGL((k_fisher_tile_v3g<4, true, 0>), grid, block, 0, s, p, f, sg);
	else hipLaunchKernelGGL((k_fisher_tile_v3g<4, false, 0>), grid, block, 0, s, p, f, sg);
	if (g_prof_on)
	{
		(void)hipEventRecord(ev1, s);
		g_prof_events.push_back(std::make_pair(ev0, ev1));
	}
}

template <int C>
static void fr_launch_fisher(FrParams& p, FrFisherArgs f, float* packed, uint8_t* fallback, hipStream_t s)
{
	dim3 grid(p.T * p.V), block(FR_THREADS);
	const bool hi = f.H_inv != nullptr, ho = f.out_H != nullptr;
	const bool per_view = hi && f.hinv_stride != 0;
	// wave-private passes over the sorted keys; tiles whose lists do not fit the LDS index are flagged ...
	f.only_flagged = nullptr;
	hipLaunchKernelGGL((k_pack_static<C>), dim3((p.P + FR_THREADS - 1) / FR_THREADS), block, 0, s, p, (hi && !per_view) ? f.H_inv : nullptr, packed, (float4*)nullptr, (float4*)nullptr);
	// measurement hook: events around the dominant kernel only, on the stream it runs on
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if (g_prof_on)
	{
		(void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
		(void)hipEventRecord(ev0, s);
	}
	if (hi && ho) hipLaunchKernelGGL((k_fisher_tile_v2<C, true, true>), grid, block, 0, s, p, f, (const float*)packed, fallback);
	else if (hi) hipLaunchKernelGGL((k_fisher_tile_v2<C, true, false>), grid, block, 0, s, p, f, (const float*)packed, fallback);
	else hipLaunchKernelGGL((k_fisher_tile_v2<C, false, true>), grid, block, 0, s, p, f, (const float*)packed, fallback);
	if (g_prof_on)
	{
		(void)hipEventRecord(ev1, s);
		g_prof_events.push_back(std::make_pair(ev0, ev1));
	}
	// ... and redone by the scan kernel
	f.only_flagged = fallback;
	if (hi && ho) hipLaunchKernelGGL((k_fisher_tile<C, true, true>), grid, block, 0, s, p, f);
	else if (hi) hipLaunchKernelGGL((k_fisher_tile<C, true, false>), grid, block, 0, s, p, f);
	else hipLaunchKernelGGL((k_fisher_tile<C, false, true>), grid, block, 0, s, p, f);
}

extern "C" int fr_fisher_views(const fr_raster_cfg* cfg, const fr_gaussians* g, const fr_fisher_cfg* fc,
                               void* workspace, size_t workspace_bytes, int64_t max_rendered,
                               int32_t* status, fr_stream_t stream)
{
	int rc = fr_validate(cfg, g, "fr_fisher_views");
	if (rc) return rc;
	if (!fc || fc->n_views <= 0 || !fc->w2c || !status) return fr_fail(FR_EINVAL, "fr_fisher_views: bad fisher cfg");
	if (fc->columns != 4 && fc->columns != 11) return fr_fail(FR_EINVAL, "fr_fisher_views: columns must be 4 or 11");
	if (!g->colors_precomp) return fr_fail(FR_EINVAL, "fr_fisher_views: needs colors_precomp (the reference always passes rgb_colors)");
	if (fc->columns == 11 && !(g->scales && g->rotations)) return fr_fail(FR_EINVAL, "fr_fisher_views: columns=11 needs scales and rotations");
	if (fc->out_scores && !fc->H_inv) return fr_fail(FR_EINVAL, "fr_fisher_views: out_scores needs H_inv");
	if (!fc->H_inv && !fc->out_H) return fr_fail(FR_EINVAL, "fr_fisher_views: nothing to compute (no H_inv and no out_H)");
	if (fc->H_inv && !fc->out_scores) return fr_fail(FR_EINVAL, "fr_fisher_views: H_inv without out_scores");
	if (fc->dL_dpix_image && (fc->H_inv || !fc->out_H)) return fr_fail(FR_EINVAL, "fr_fisher_views: dL_dpix_image is for the out_H mode (no H_inv)");
	const int P = cfg->P, W = cfg->image_width, H = cfg->image_height, V = fc->n_views;
	hipStream_t s = (hipStream_t)stream;
	if (P == 0)
	{
		if (fc->out_scores) (void)hipMemsetAsync(fc->out_scores, 0, (size_t)V * 4, s);
		if (fc->out_vis_count) (void)hipMemsetAsync(fc->out_vis_count, 0, (size_t)V * 4, s);
		if (fc->out_num_rendered) (void)hipMemsetAsync(fc->out_num_rendered, 0, (size_t)V * 4, s);
		(void)hipMemsetAsync(status, 0, 16, s);
		return FR_OK;
	}
	FrFisherLayout L = fr_fisher_layout(P, W, H, V, max_rendered, fc->columns);
	if (!workspace || workspace_bytes < L.total) return fr_fail(FR_ENOSPACE, "fr_fisher_views: workspace smaller than fr_fisher_workspace_bytes()");
	char* ws = (char*)workspace;
	FrParams p;
	fr_fill_params(p, cfg, g, V);
	p.prefiltered = 0;            // (candidate views cull by design; status[3] means something else here)
	p.w2c = fc->w2c;
	if (fc->poses_are_c2w)
	{
		float* inv = (float*)(ws + L.w2c_inv);
		hipLaunchKernelGGL(k_invert_poses, dim3((V + 63) / 64), dim3(64), 0, s, V, fc->w2c, inv);
		if ((rc = fr_check_launch("k_invert_poses"))) return rc;
		p.w2c = inv;
	}
	p.radii = (int*)(ws + L.radii);
	p.vis_list = (FrVisEntry*)(ws + L.radii);
	p.vis_n = (uint32_t*)(ws + L.vis_n);
	p.splat = (FrSplat*)(ws + L.splat);
	p.cov3D_out = (float*)(ws + L.cov3D);
	p.tile_cnt = (uint32_t*)(ws + L.tile_cnt);
	p.tile_off = (uint32_t*)(ws + L.tile_off);
	p.tile_fill = (uint32_t*)(ws + L.tile_fill);
	p.status = (int*)(ws + L.status);
	p.big_list = (uint32_t*)(ws + L.big_list);
	// the views dealt over the XCDs by weight (FR_DEBUG_MODE=27: in index order, for A/B runs)
	const bool deal = (V & 7) == 0 && V <= 1024 && fr_debug_mode() != 27;
	p.view_work = deal ? (uint32_t*)(ws + L.view_work) : nullptr;
	p.view_perm = deal ? (uint32_t*)(ws + L.view_perm) : nullptr;
	p.blk_base = (uint32_t*)(ws + L.blk_base);
	p.keys = (uint64_t*)(ws + L.keys);
	p.key_capacity = max_rendered;
	if (fc->tile_capacity < 0) return fr_fail(FR_EINVAL, "fr_fisher_views: negative tile_capacity");
	if (fc->tile_capacity > 0)
	{
		// fixed key segments: V T tile_capacity keys must fit the key buffer (and 32-bit offsets)
		const long long need = (long long)V * p.T * (long long)fc->tile_capacity;
		if (need > max_rendered || need >= (1ll << 32)) return fr_fail(FR_EINVAL, "fr_fisher_views: n_views * tiles * tile_capacity exceeds max_rendered (or 2^32)");
		// (else packed lists: the 8-byte list entries hold tile columns and strip rows in bytes, the keys 28-bit record slots;
		// FR_DEBUG_MODE 8 / 24: tile kernels that do not read the keys' strip bits)
		const long long nblk_c = (P + FR_THREADS * fr_pick_G_views(P) - 1) / (FR_THREADS * fr_pick_G_views(P));
		const bool slots_fit = nblk_c * FR_THREADS * fr_pick_G_views(P) < (1ll << 28);
		// (FR_DEBUG_MODE 20: the parking form of the projection kernel, which fills packed lists only)
		if (p.gx <= 255u && p.gy <= 63u && slots_fit && fr_debug_mode() != 8 && fr_debug_mode() != 24 && fr_debug_mode() != 20) p.tile_cap = (uint32_t)fc->tile_capacity;
	}
	p.vis_count = fc->out_vis_count;
	p.num_rendered = fc->out_num_rendered;
	FrFisherArgs f;
	f.dL = fc->dL_dpix;
	f.dL_img = fc->dL_dpix_image; f.dL_stride = fc->dL_image_view_stride;
	f.H_inv = fc->H_inv; f.hinv_stride = fc->H_inv_view_stride;
	f.out_H = fc->out_H; f.outH_stride = fc->out_H_view_stride;
	f.tile_scores = (float*)(ws + L.tile_scores);
	f.only_flagged = nullptr;
	f.debug_mode = fr_debug_mode();
	f.key_shift = 0;
	// score-only (H_inv, no out_H, constant upstream gradient): records + one front-to-back pass; FR_DEBUG_MODE=9 keeps
	// the second-generation two-pass kernel for A/B runs
	const bool v3 = fc->H_inv && !fc->out_H && !fc->dL_dpix_image && f.debug_mode != 1 && f.debug_mode != 9;
	// the diagonal itself (out_H, no H_inv, 4 columns, constant upstream gradient): records + two front-to-back passes
	const bool v3h = !fc->H_inv && fc->out_H && !fc->dL_dpix_image && fc->columns == 4 && f.debug_mode != 1 && f.debug_mode != 9;
	// the other out_H modes (11 columns and / or a per-view upstream-gradient image: GaussianObjectSLAM, the POp-GS probes) on
	// records too, through the multi-view front end; FR_DEBUG_MODE=22 keeps round 2's two-pass kernel (k_fisher_tile_v2) for A/B runs
	const bool multi_fe = p.vis_list != nullptr && p.T <= FR_MAX_LDS_TILES;
	const bool v3g = !fc->H_inv && fc->out_H && !v3h && multi_fe && !g->cov3D_precomp && f.debug_mode != 1 && f.debug_mode != 9 && f.debug_mode != 22 && f.debug_mode != 19;
	FrScorerPlan plan;
	plan.form_a = v3h;
	plan.general = v3g;
	plan.columns = fc->columns;
	plan.ra.H_inv = fc->H_inv; plan.ra.hinv_stride = fc->H_inv_view_stride;
	plan.ra.packed = (const float*)(ws + L.packed); plan.ra.recq = (float4*)(ws + L.recq);
	// compact records with the multi-view front end (the same condition fr_bin_pipeline uses for it); FR_DEBUG_MODE=19: dense (A/B runs)
	const bool compact = (v3 || v3h || v3g) && multi_fe && f.debug_mode != 19;
	if (!compact) p.tile_cap = 0;                   // (fixed key segments are filled by the compact-record front end only)
	else (void)fr_plan_views_c(p.T, fr_pick_VC(p.T), p.tile_cap);      // ... and only while their LDS cursors fit (decided HERE, before the stride)
	// float4 per record: the general out_H forms 13 / 7, the score form 5 with fixed key segments (80 bytes: FrRecStride), else 6
	const int rstride = v3g ? (fc->columns == 11 ? 13 : 7) : ((v3 && p.tile_cap) ? 5 : 6);
	plan.ra.comp = compact ? (float4*)(ws + L.recq) : nullptr;
	plan.ra.stride = rstride;
	plan.ra.slot_idx = (compact && (v3h || v3g)) ? (uint32_t*)(ws + L.slot_idx) : nullptr;
	if (compact)
	{
		f.recA = plan.ra.comp; f.ab_view = (long long)L.PV * rstride; f.ab_stride = rstride;
		f.recQ = plan.ra.comp + 2; f.q_view = (long long)L.PV * rstride; f.q_stride = rstride;
		f.slot_idx = plan.ra.slot_idx; f.slot_view = (long long)L.PV;
	}
	else
	{
		f.recA = (const float4*)p.splat; f.ab_view = (long long)P * 2; f.ab_stride = 2;
		f.recQ = plan.ra.recq; f.q_view = (long long)P * 4; f.q_stride = 4;
		f.slot_idx = nullptr; f.slot_view = 0;
	}
	// the early frustum test needs a positive semi-definite cov3D: the one k_cov3d builds, not a caller's precomputed one;
	// FR_DEBUG_MODE=15 switches it off (A/B runs)
	plan.ra.mt = (const float4*)(ws + L.mt);
	plan.ra.grp = (const float4*)(ws + L.grp);
	plan.ra.early = (g->cov3D_precomp || f.debug_mode == 15) ? 0 : 1;
	// the processing order is honoured where the compact-record front end runs (it is a layout hint: the other paths ignore it)
	p.order = compact ? fc->order : nullptr;

	// ---- view groups (off by default: fr_pick_groups has the measurement).  The front end (projection, records, scan, scatter,
	// sorts: ~1.1 ms at 500k Gaussians x 64 views) waits on memory for half its wave cycles, the tile kernel (~1.0 ms) is bound by
	// VALU issue; to let them overlap, a records-mode call can be cut into groups of whole multiples of 8 views (the XCD map of the
	// tile kernels): the groups' front ends run one after the other on the caller's stream, every group's tile kernel on a side
	// stream as soon as ITS front end is done -- underneath the next group's front end.  A view's tiles are computed by the same
	// code on the same data whatever the grouping, so the scores are bit-identical to the single-group launch
	// (tests/test_gpu_bench_multirank.py::test_view_groups_give_the_same_scores).
	const bool multi = p.vis_list != nullptr && p.T <= FR_MAX_LDS_TILES;
	const int n_groups = (v3 && multi) ? fr_pick_groups(V) : 1;      // (score-only: an out_H launch must accumulate all or nothing on overflow)
	FrSideStream& tside = fr_side_stream(2);
	const bool use_side = n_groups > 1 && tside.ok;
	hipStream_t ts = use_side ? tside.stream : s;                  // the tile kernels' stream
	static thread_local hipEvent_t front_done[FR_MAX_GROUPS] = { nullptr, nullptr, nullptr, nullptr };
	const size_t nblk_v = (size_t)((P + FR_THREADS * fr_pick_G_views(P) - 1) / (FR_THREADS * fr_pick_G_views(P)));
	const size_t cap_v = (size_t)(FR_THREADS * fr_pick_G_views(P));
	const size_t rec_view = compact ? L.PV : (size_t)P;            // records per view in the splat / record regions
	struct TileJoin {                                              // the caller's stream waits for the tile stream on every way out
		hipStream_t s; FrSideStream* side; bool on;
		~TileJoin() { if (on) { (void)hipEventRecord(side->join, side->stream); (void)hipStreamWaitEvent(s, side->join, 0); } }
	} tile_join{ s, &tside, false };
	for (int gi = 0; gi < n_groups; gi++)
	{
		// contiguous view slices, every one but the last a multiple of 8 views
		const int per = n_groups > 1 ? ((V / n_groups + 7) / 8) * 8 : V;
		const int v0 = gi * per, Vg = (gi == n_groups - 1) ? V - v0 : per;
		if (Vg <= 0) break;
		FrParams pg = p;
		pg.V = Vg;
		pg.w2c = p.w2c + 16 * (size_t)v0;
		pg.radii = p.radii ? (int*)((char*)p.radii + (size_t)v0 * nblk_v * cap_v * sizeof(FrVisEntry)) : nullptr;
		pg.vis_list = p.vis_list + (size_t)v0 * nblk_v * cap_v;
		pg.vis_n = p.vis_n + (size_t)v0 * nblk_v;
		pg.splat = p.splat + (size_t)v0 * rec_view;
		pg.tile_cnt = p.tile_cnt + (size_t)v0 * p.T; pg.tile_off = p.tile_off + (size_t)v0 * p.T; pg.tile_fill = p.tile_fill + (size_t)v0 * p.T;
		pg.status = p.status + 4 * gi;
		pg.big_list = p.big_list + (size_t)v0 * p.T + 16 * gi;
		if (p.view_perm && (Vg & 7) == 0) { pg.view_work = p.view_work + v0; pg.view_perm = p.view_perm + v0; }
		else { pg.view_work = nullptr; pg.view_perm = nullptr; }
		pg.blk_base = p.blk_base + (size_t)v0 * (size_t)fr_preprocess_blocks(P, V) * (size_t)p.T;
		pg.tile_cap = p.tile_cap;                                  // (fr_bin_pipeline clears it where the front end cannot fill fixed segments)
		const long long k0 = (long long)((__int128)max_rendered * v0 / V), k1 = (long long)((__int128)max_rendered * (v0 + Vg) / V);
		pg.keys = p.keys + k0; pg.key_capacity = k1 - k0;
		pg.vis_count = p.vis_count ? p.vis_count + v0 : nullptr;
		pg.num_rendered = p.num_rendered ? p.num_rendered + v0 : nullptr;
		FrScorerPlan pl = plan;
		pl.skip_pack = gi > 0 || (fc->reuse_static != 0 && compact);      // (the caller vouches that the workspace holds this call's static records)
		if (pl.ra.hinv_stride) pl.ra.H_inv = plan.ra.H_inv + (size_t)v0 * plan.ra.hinv_stride;
		pl.ra.recq = plan.ra.recq + (size_t)v0 * rec_view * 4;
		if (pl.ra.comp) pl.ra.comp = plan.ra.comp + (size_t)v0 * L.PV * rstride;
		if (pl.ra.slot_idx) pl.ra.slot_idx = plan.ra.slot_idx + (size_t)v0 * L.PV;
		FrFisherArgs fg = f;
		fg.recA = f.recA + (size_t)v0 * f.ab_view; fg.recQ = f.recQ + (size_t)v0 * f.q_view;
		if (f.slot_idx) fg.slot_idx = f.slot_idx + (size_t)v0 * f.slot_view;
		fg.tile_scores = f.tile_scores + (size_t)v0 * p.T;
		if (f.H_inv && f.hinv_stride) fg.H_inv = f.H_inv + (size_t)v0 * f.hinv_stride;
		if (f.out_H && f.outH_stride) fg.out_H = f.out_H + (size_t)v0 * f.outH_stride;
		if (f.dL_img && f.dL_stride) fg.dL_img = f.dL_img + (size_t)v0 * f.dL_stride;

		if ((rc = fr_bin_pipeline(pg, g, s, (v3 || v3h || v3g) ? &pl : nullptr))) return rc;
		fg.key_shift = pg.tile_cap ? 4 : 0;                          // (fixed segments: k_preprocess_views_c<.., true> wrote depth | slot << 4 | strips)
		if (use_side)
		{
			if (!front_done[gi] && hipEventCreateWithFlags(&front_done[gi], hipEventDisableTiming) != hipSuccess)
				return fr_fail(FR_ELAUNCH, "fr_fisher_views: hipEventCreate failed");
			if (hipEventRecord(front_done[gi], s) != hipSuccess || hipStreamWaitEvent(ts, front_done[gi], 0) != hipSuccess)
				return fr_fail(FR_ELAUNCH, "fr_fisher_views: fork to the tile stream failed");
			tile_join.on = true;
		}
		if (v3) fr_launch_fisher_v3(pg, fg, pl.ra.recq, ts);
		else if (v3h || v3g)
		{
			FrSegArgs sg;
			sg.snapT = (float*)(ws + L.seg_T); sg.snapC = (double*)(ws + L.seg_C); sg.X = (double*)(ws + L.seg_X);
			sg.list = (uint32_t*)(ws + L.seg_list);
			if (pg.V * pg.T > FR_SEG_TILES) sg.snapT = nullptr;
			if (v3h) fr_launch_fisher_v3h(pg, fg, pl.ra.recq, sg, ts);
			else fr_launch_fisher_v3g(pg, fg, fc->columns, fc->dL_dpix_image != nullptr, sg, ts);
		}
		else if (fc->columns == 4) fr_launch_fisher<4>(pg, fg, (float*)(ws + L.packed), (uint8_t*)(ws + L.fallback), ts);
		else fr_launch_fisher<11>(pg, fg, (float*)(ws + L.packed), (uint8_t*)(ws + L.fallback), ts);
		if ((rc = fr_check_launch("k_fisher_tile"))) return rc;
	}
	// scores in a fixed order + the caller's status word {sum of tile instances, any overflow, max, 0}
	hipLaunchKernelGGL(k_reduce_scores, dim3(fc->out_scores ? V : 1), dim3(FR_THREADS), 0, ts, f.tile_scores, p.T, p.status, n_groups,
	                   fc->out_scores, status);
	if ((rc = fr_check_launch("k_reduce_scores"))) return rc;
	return FR_OK;                                                    // (tile_join: the caller's stream now waits for the tile stream)
}

// =========================================================================================================
// Densification / pruning statistics of the training step.  The reference spreads these over a dozen torch
// ops per mapping iteration (gaussian.py:289-292; slam_external.py:196-200, 345-465); here each is one pass over the
// Gaussians.  exp / sigmoid use fr_expf + IEEE division, the sequence the oracle restates, so the masks are reproducible
// bit for bit.
// =========================================================================================================
// after a render + backward: seen = radius > 0; max_2D_radius = max(radius, max_2D_radius) where seen (gaussian.py:289-291);
// means2D_gradient_accum += |means2D.grad.xy|, denom += 1 where seen (slam_external.py:196-200; skipped when grad is null)
__global__ __launch_bounds__(FR_THREADS) void k_densify_stats(int P, const int* __restrict__ radii, const float* __restrict__ grad_means2D,
                                                              float* __restrict__ max_radius, float* __restrict__ grad_accum,
                                                              float* __restrict__ denom, uint8_t* __restrict__ seen)
{
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= P) return;
	const int r = radii[i];
	const bool vis = r > 0;
	if (seen) seen[i] = vis ? 1 : 0;
	if (!vis) return;
	if (max_radius) max_radius[i] = fmaxf((float)r, max_radius[i]);
	if (grad_means2D)
	{
		const float gx = grad_means2D[3 * (size_t)i], gy = grad_means2D[3 * (size_t)i + 1];
		grad_accum[i] += sqrtf(gx * gx + gy * gy);
		denom[i] += 1.0f;
	}
}

__device__ __forceinline__ float fr_max_scale(const float* __restrict__ log_scales, int scale_cols, int i)
{
	float m = fr_expf(log_scales[(size_t)i * scale_cols]);
	for (int k = 1; k < scale_cols; k++) m = fmaxf(m, fr_expf(log_scales[(size_t)i * scale_cols + k]));
	return m;
}

// slam_external.py:419-433: grads = accum / denom (NaN -> 0); to_clone = grads >= grad_thresh AND max scale <= clone_max_scale;
// to_split = max scale > split_min_scale -- the reference applies no gradient test to the split (its padded_grad is unused)
__global__ __launch_bounds__(FR_THREADS) void k_densify_masks(int P, const float* __restrict__ grad_accum, const float* __restrict__ denom,
                                                              const float* __restrict__ log_scales, int scale_cols, float grad_thresh,
                                                              float clone_max_scale, float split_min_scale,
                                                              uint8_t* __restrict__ to_clone, uint8_t* __restrict__ to_split)
{
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= P) return;
	float g = grad_accum[i] / denom[i];
	if (g != g) g = 0.0f;
	const float ms = fr_max_scale(log_scales, scale_cols, i);
	to_clone[i] = (g >= grad_thresh && ms <= clone_max_scale) ? 1 : 0;
	to_split[i] = (ms > split_min_scale) ? 1 : 0;
}

// slam_external.py:354, 394-396, 452-457: to_remove = sigmoid(logit_opacity) < opacity_thresh OR (big_thresh >= 0 AND max scale > big_thresh)
__global__ __launch_bounds__(FR_THREADS) void k_prune_mask(int P, const float* __restrict__ logit_opacities, const float* __restrict__ log_scales,
                                                           int scale_cols, float opacity_thresh, float big_thresh, uint8_t* __restrict__ to_remove)
{
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= P) return;
	const float op = 1.0f / (1.0f + fr_expf(-logit_opacities[i]));
	bool rm = op < opacity_thresh;
	if (big_thresh >= 0.0f) rm = rm || (fr_max_scale(log_scales, scale_cols, i) > big_thresh);
	to_remove[i] = rm ? 1 : 0;
}

extern "C" int fr_densify_stats(int32_t P, const int32_t* radii, const float* grad_means2D, float* max_2D_radius,
                                float* means2D_gradient_accum, float* denom, uint8_t* seen, fr_stream_t stream)
{
	if (P < 0) return fr_fail(FR_EINVAL, "fr_densify_stats: P < 0");
	if (P == 0) return FR_OK;
	if (!radii || (grad_means2D && (!means2D_gradient_accum || !denom))) return fr_fail(FR_EINVAL, "fr_densify_stats: null pointer");
	hipLaunchKernelGGL(k_densify_stats, dim3((P + FR_THREADS - 1) / FR_THREADS), dim3(FR_THREADS), 0, (hipStream_t)stream,
	                   P, radii, grad_means2D, max_2D_radius, means2D_gradient_accum, denom, seen);
	return fr_check_launch("k_densify_stats");
}

extern "C" int fr_densify_masks(int32_t P, const float* means2D_gradient_accum, const float* denom, const float* log_scales,
                                int32_t scale_cols, float grad_thresh, float clone_max_scale, float split_min_scale,
                                uint8_t* to_clone, uint8_t* to_split, fr_stream_t stream)
{
	if (P < 0 || (scale_cols != 1 && scale_cols != 3)) return fr_fail(FR_EINVAL, "fr_densify_masks: bad P / scale_cols");
	if (P == 0) return FR_OK;
	if (!means2D_gradient_accum || !denom || !log_scales || !to_clone || !to_split) return fr_fail(FR_EINVAL, "fr_densify_masks: null pointer");
	hipLaunchKernelGGL(k_densify_masks, dim3((P + FR_THREADS - 1) / FR_THREADS), dim3(FR_THREADS), 0, (hipStream_t)stream,
	                   P, means2D_gradient_accum, denom, log_scales, scale_cols, grad_thresh, clone_max_scale, split_min_scale, to_clone, to_split);
	return fr_check_launch("k_densify_masks");
}

extern "C" int fr_prune_mask(int32_t P, const float* logit_opacities, const float* log_scales, int32_t scale_cols,
                             float opacity_thresh, float big_thresh, uint8_t* to_remove, fr_stream_t stream)
{
	if (P < 0 || (scale_cols != 1 && scale_cols != 3)) return fr_fail(FR_EINVAL, "fr_prune_mask: bad P / scale_cols");
	if (P == 0) return FR_OK;
	if (!logit_opacities || !to_remove || (big_thresh >= 0.0f && !log_scales)) return fr_fail(FR_EINVAL, "fr_prune_mask: null pointer");
	hipLaunchKernelGGL(k_prune_mask, dim3((P + FR_THREADS - 1) / FR_THREADS), dim3(FR_THREADS), 0, (hipStream_t)stream,
	                   P, logit_opacities, log_scales, scale_cols, opacity_thresh, big_thresh, to_remove);
	return fr_check_launch("k_prune_mask");
}

// =========================================================================================================
// simple-knn: distCUDA2.  Upstream (gitlab.inria.fr/bkerbl/simple-knn, not vendored in the reference) orders the
// points along a Morton curve, boxes them 1024 at a time and prunes boxes by their distance to the query; the
// result is the EXACT mean squared distance to the 3 nearest other points, so any exact search reproduces it.
// Here: Morton keys -> multi-block bitonic network (same ascending-only network as the tile sort) -> box AABBs
// -> one thread per point scanning the boxes that can still improve its third-best distance.
// =========================================================================================================
#define FR_KNN_BOX 1024
#define FR_KNN_CHUNK 2048

__device__ __forceinline__ uint32_t fr_enc_f(float f)
{
	uint32_t u = fr_as_u32(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fr_dec_f(uint32_t e)
{
	return fr_as_f32((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

__global__ __launch_bounds__(FR_THREADS) void k_knn_minmax(int P, const float* __restrict__ pts, uint32_t* __restrict__ mm)
{
	__shared__ float s_lo[3][4], s_hi[3][4];
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (int i = blockIdx.x * FR_THREADS + threadIdx.x; i < P; i += gridDim.x * FR_THREADS)
#pragma unroll
		for (int a = 0; a < 3; a++) { float x = pts[3 * (size_t)i + a]; lo[a] = fminf(lo[a], x); hi[a] = fmaxf(hi[a], x); }
#pragma unroll
	for (int a = 0; a < 3; a++)
	{
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
		if ((threadIdx.x & 63) == 0) { s_lo[a][threadIdx.x >> 6] = lo[a]; s_hi[a][threadIdx.x >> 6] = hi[a]; }
	}
	__syncthreads();
	// one pair of atomics per workgroup and axis: they all land on the same six words
	if (threadIdx.x < 3)
	{
		const int a = threadIdx.x;
		const float l = fminf(fminf(s_lo[a][0], s_lo[a][1]), fminf(s_lo[a][2], s_lo[a][3]));
		const float h = fmaxf(fmaxf(s_hi[a][0], s_hi[a][1]), fmaxf(s_hi[a][2], s_hi[a][3]));
		atomicMin(&mm[a], fr_enc_f(l)); atomicMax(&mm[3 + a], fr_enc_f(h));
	}
}

__device__ __forceinline__ uint32_t fr_spread10(uint32_t x)
{
	x = (x | (x << 16)) & 0x030000FF;
	x = (x | (x << 8)) & 0x0300F00F;
	x = (x | (x << 4)) & 0x030C30C3;
	x = (x | (x << 2)) & 0x09249249;
	return x;
}

__global__ __launch_bounds__(FR_THREADS) void k_knn_morton(int P, const float* __restrict__ pts, const uint32_t* __restrict__ mm,
                                                           uint64_t* __restrict__ keys)
{
	int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i >= P) return;
	uint32_t code = 0;
#pragma unroll
	for (int a = 0; a < 3; a++)
	{
		const float lo = fr_dec_f(mm[a]), hi = fr_dec_f(mm[3 + a]);
		const float ext = hi - lo;
		float u = ext > 0.f ? (pts[3 * (size_t)i + a] - lo) / ext : 0.f;
		u = fminf(fmaxf(u, 0.f), 1.f);
		code |= fr_spread10((uint32_t)(u * 1023.0f)) << a;
	}
	keys[i] = ((uint64_t)code << 32) | (uint32_t)i;
}

// sort each FR_KNN_CHUNK-aligned chunk completely in LDS (stages k = 2 .. FR_KNN_CHUNK)
__global__ __launch_bounds__(FR_THREADS) void k_bitonic_chunk_sort(uint64_t* __restrict__ keys, uint32_t n)
{
	__shared__ uint64_t sk[FR_KNN_CHUNK];
	const uint32_t base = blockIdx.x * FR_KNN_CHUNK;
	const uint32_t m = min((uint32_t)FR_KNN_CHUNK, n - base);
	for (uint32_t i = threadIdx.x; i < m; i += FR_THREADS) sk[i] = keys[base + i];
	__syncthreads();
	fr_bitonic(sk, m, threadIdx.x);
	for (uint32_t i = threadIdx.x; i < m; i += FR_THREADS) keys[base + i] = sk[i];
}

// one global stage: flip (i <-> i ^ (k-1)) or half-cleaner (i <-> i + j)
__global__ __launch_bounds__(FR_THREADS) void k_bitonic_global(uint64_t* __restrict__ keys, uint32_t n, uint32_t half_pairs,
                                                               uint32_t k, uint32_t j)
{
	const uint32_t t = blockIdx.x * FR_THREADS + threadIdx.x;
	if (t >= half_pairs) return;
	uint32_t i, l;
	if (j == 0) { const uint32_t half = k >> 1; i = ((t / half) * k) + (t % half); l = i ^ (k - 1); }
	else { i = ((t / j) * (j << 1)) + (t % j); l = i + j; }
	if (l < n)
	{
		uint64_t a = keys[i], b = keys[l];
		if (a > b) { keys[i] = b; keys[l] = a; }
	}
}

// half-cleaner stages j = FR_KNN_CHUNK/2 .. 1 inside each chunk, in LDS
__global__ __launch_bounds__(FR_THREADS) void k_bitonic_chunk_merge(uint64_t* __restrict__ keys, uint32_t n)
{
	__shared__ uint64_t sk[FR_KNN_CHUNK];
	const uint32_t base = blockIdx.x * FR_KNN_CHUNK;
	const uint32_t m = min((uint32_t)FR_KNN_CHUNK, n - base);
	for (uint32_t i = threadIdx.x; i < m; i += FR_THREADS) sk[i] = keys[base + i];
	__syncthreads();
	for (uint32_t j = FR_KNN_CHUNK >> 1; j > 0; j >>= 1)
	{
		for (uint32_t t = threadIdx.x; t < (FR_KNN_CHUNK >> 1); t += FR_THREADS)
		{
			const uint32_t i = ((t / j) * (j << 1)) + (t % j);
			const uint32_t l = i + j;
			if (l < m)
			{
				uint64_t a = sk[i], b = sk[l];
				if (a > b) { sk[i] = b; sk[l] = a; }
			}
		}
		__syncthreads();
	}
	for (uint32_t i = threadIdx.x; i < m; i += FR_THREADS) keys[base + i] = sk[i];
}

__global__ __launch_bounds__(FR_THREADS) void k_knn_gather(int P, const float* __restrict__ pts, const uint64_t* __restrict__ keys,
                                                           float* __restrict__ sorted)
{
	int t = blockIdx.x * FR_THREADS + threadIdx.x;
	if (t >= P) return;
	const uint32_t i = (uint32_t)keys[t];
	sorted[3 * (size_t)t] = pts[3 * (size_t)i]; sorted[3 * (size_t)t + 1] = pts[3 * (size_t)i + 1]; sorted[3 * (size_t)t + 2] = pts[3 * (size_t)i + 2];
}

__global__ __launch_bounds__(FR_THREADS) void k_knn_boxes(int P, const float* __restrict__ sorted, float* __restrict__ boxes)
{
	__shared__ float red[6][4];
	const int b = blockIdx.x;
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	const int end = min(P, (b + 1) * FR_KNN_BOX);
	for (int t = b * FR_KNN_BOX + threadIdx.x; t < end; t += FR_THREADS)
#pragma unroll
		for (int a = 0; a < 3; a++) { float x = sorted[3 * (size_t)t + a]; lo[a] = fminf(lo[a], x); hi[a] = fmaxf(hi[a], x); }
#pragma unroll
	for (int a = 0; a < 3; a++)
	{
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
		if ((threadIdx.x & 63) == 0) { red[a][threadIdx.x >> 6] = lo[a]; red[3 + a][threadIdx.x >> 6] = hi[a]; }
	}
	__syncthreads();
	if (threadIdx.x < 6)
	{
		const int a = threadIdx.x;
		float r = red[a][0];
		for (int w = 1; w < 4; w++) r = a < 3 ? fminf(r, red[a][w]) : fmaxf(r, red[a][w]);
		boxes[6 * (size_t)b + a] = r;
	}
}

__device__ __forceinline__ void fr_knn_update(float px, float py, float pz, float qx, float qy, float qz, float* best)
{
	const float dx = qx - px, dy = qy - py, dz = qz - pz;
	float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
	for (int j = 0; j < 3; j++)
		if (best[j] > dist) { float t = best[j]; best[j] = dist; dist = t; }
}

// AABBs of the 64-point sub-boxes (16 per 1024-point box): one wave per sub-box
#define FR_KNN_SUB 64
__global__ __launch_bounds__(FR_THREADS) void k_knn_subboxes(int P, const float* __restrict__ sorted, float* __restrict__ sboxes)
{
	const int sb = (int)((blockIdx.x * FR_THREADS + threadIdx.x) >> 6), lane = threadIdx.x & 63;
	if (sb * FR_KNN_SUB >= P) return;
	const int t = sb * FR_KNN_SUB + lane;
	float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
	if (t < P)
	{
#pragma unroll
		for (int a = 0; a < 3; a++) { lo[a] = sorted[3 * (size_t)t + a]; hi[a] = lo[a]; }
	}
#pragma unroll
	for (int a = 0; a < 3; a++)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
	if (lane == 0)
	{
#pragma unroll
		for (int a = 0; a < 3; a++) { sboxes[6 * (size_t)sb + a] = lo[a]; sboxes[6 * (size_t)sb + 3 + a] = hi[a]; }
	}
}

__device__ __forceinline__ float fr_knn_box_dist2(float px, float py, float pz, const float* __restrict__ bx)
{
	float ddx = 0.f, ddy = 0.f, ddz = 0.f;
	if (px < bx[0] || px > bx[3]) ddx = fminf(fabsf(px - bx[0]), fabsf(px - bx[3]));
	if (py < bx[1] || py > bx[4]) ddy = fminf(fabsf(py - bx[1]), fabsf(py - bx[4]));
	if (pz < bx[2] || pz > bx[5]) ddz = fminf(fabsf(pz - bx[2]), fabsf(pz - bx[5]));
	return ddx * ddx + ddy * ddy + ddz * ddz;
}

// Wave-cooperative exact search.  The 64 query points of a wave are consecutive on the Morton curve, i.e. close in space, so
// the wave walks the box hierarchy together: a box (then a 64-point sub-box) is opened when ANY lane can still improve its
// third-best distance there, and every lane then tests the same candidate -- the loop counters, box bounds and candidate
// coordinates are wave-uniform (scalar loads), with no divergence.  A lane sees a superset of the candidates its own
// pruning rule admits, so its three smallest distances are exactly those of the brute-force search.
__global__ __launch_bounds__(FR_THREADS) void k_knn_search(int P, const float* __restrict__ sorted, const uint64_t* __restrict__ keys,
                                                           const float* __restrict__ boxes, int nb, const float* __restrict__ sboxes,
                                                           int nsb, float* __restrict__ out)
{
	const int lane = threadIdx.x & 63;
	const int wbase = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * FR_THREADS + (threadIdx.x & ~63u)));
	if (wbase >= P) return;
	const int t = wbase + lane;
	const bool live = t < P;
	const int tq = live ? t : wbase;
	const float px = sorted[3 * (size_t)tq], py = sorted[3 * (size_t)tq + 1], pz = sorted[3 * (size_t)tq + 2];
	float best[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f };
	for (int s = max(0, tq - 3); s <= min(P - 1, tq + 3); s++)
	{
		if (s == tq) continue;
		fr_knn_update(px, py, pz, sorted[3 * (size_t)s], sorted[3 * (size_t)s + 1], sorted[3 * (size_t)s + 2], best);
	}
	const float reject = best[2];
	best[0] = best[1] = best[2] = 3.402823466e+38f;
	for (int b = 0; b < nb; b++)
	{
		const float d1 = fr_knn_box_dist2(px, py, pz, boxes + 6 * (size_t)b);
		if (!fr_any(live && !(d1 > reject || d1 > best[2]))) continue;
		const int sb1 = min(nsb, (b + 1) * (FR_KNN_BOX / FR_KNN_SUB));
		for (int sb = b * (FR_KNN_BOX / FR_KNN_SUB); sb < sb1; sb++)
		{
			const float d2 = fr_knn_box_dist2(px, py, pz, sboxes + 6 * (size_t)sb);
			if (!fr_any(live && !(d2 > reject || d2 > best[2]))) continue;
			const int end = min(P, (sb + 1) * FR_KNN_SUB);
			for (int s = sb * FR_KNN_SUB; s < end; s++)
			{
				const float qx = sorted[3 * (size_t)s], qy = sorted[3 * (size_t)s + 1], qz = sorted[3 * (size_t)s + 2];
				if (s != tq) fr_knn_update(px, py, pz, qx, qy, qz, best);
			}
		}
	}
	if (live) out[(uint32_t)keys[t]] = (best[0] + best[1] + best[2]) / 3.0f;
}

struct FrKnnLayout { size_t keys, sorted, boxes, sboxes, mm, total; };
static FrKnnLayout fr_knn_layout(int64_t P)
{
	FrKnnLayout L;
	size_t o = 0;
	const int64_t nb = (P + FR_KNN_BOX - 1) / FR_KNN_BOX;
	L.keys = o; o = fr_align(o + (size_t)P * 8);
	L.sorted = o; o = fr_align(o + (size_t)P * 12);
	L.boxes = o; o = fr_align(o + (size_t)nb * 24);
	L.sboxes = o; o = fr_align(o + (size_t)((P + FR_KNN_SUB - 1) / FR_KNN_SUB) * 24);
	L.mm = o; o = fr_align(o + 32);
	L.total = o;
	return L;
}

extern "C" size_t fr_knn_workspace_bytes(int32_t P)
{
	if (P < 0) return 0;
	return fr_knn_layout(P).total;
}

// keys[i] = Morton code of point i << 32 | i, sorted ascending (unique keys: the order is that of a stable sort by code)
static int fr_morton_sort(int P, const float* points, uint64_t* keys, uint32_t* mm, hipStream_t s)
{
	int rc;
	(void)hipMemsetAsync(mm, 0xFF, 12, s);
	(void)hipMemsetAsync(mm + 3, 0x00, 12, s);
	const int nblk = (P + FR_THREADS - 1) / FR_THREADS;
	hipLaunchKernelGGL(k_knn_minmax, dim3(nblk < 256 ? nblk : 256), dim3(FR_THREADS), 0, s, P, points, mm);
	if ((rc = fr_check_launch("k_knn_minmax"))) return rc;
	hipLaunchKernelGGL(k_knn_morton, dim3(nblk), dim3(FR_THREADS), 0, s, P, points, mm, keys);
	if ((rc = fr_check_launch("k_knn_morton"))) return rc;
	// bitonic network over the whole array
	const uint32_t n = (uint32_t)P;
	uint32_t n_pad = 1;
	while (n_pad < n) n_pad <<= 1;
	const uint32_t nchunks = (n + FR_KNN_CHUNK - 1) / FR_KNN_CHUNK;
	hipLaunchKernelGGL(k_bitonic_chunk_sort, dim3(nchunks), dim3(FR_THREADS), 0, s, keys, n);
	if ((rc = fr_check_launch("k_bitonic_chunk_sort"))) return rc;
	const uint32_t half_pairs = n_pad >> 1;
	const uint32_t gblk = (half_pairs + FR_THREADS - 1) / FR_THREADS;
	for (uint32_t k = FR_KNN_CHUNK << 1; k <= n_pad && k != 0; k <<= 1)
	{
		hipLaunchKernelGGL(k_bitonic_global, dim3(gblk), dim3(FR_THREADS), 0, s, keys, n, half_pairs, k, 0u);
		for (uint32_t j = k >> 2; j >= FR_KNN_CHUNK; j >>= 1)
			hipLaunchKernelGGL(k_bitonic_global, dim3(gblk), dim3(FR_THREADS), 0, s, keys, n, half_pairs, k, j);
		hipLaunchKernelGGL(k_bitonic_chunk_merge, dim3(nchunks), dim3(FR_THREADS), 0, s, keys, n);
	}
	return fr_check_launch("k_bitonic_global");
}

// ---- fr_spatial_order: the Gaussians along a Z-curve (fr_fisher_cfg.order) ---------------------------------------------------
__global__ __launch_bounds__(FR_THREADS) void k_order_from_keys(int P, const uint64_t* __restrict__ keys, uint32_t* __restrict__ order)
{
	const int i = blockIdx.x * FR_THREADS + threadIdx.x;
	if (i < P) order[i] = (uint32_t)keys[i];
}
extern "C" size_t fr_spatial_order_workspace_bytes(int32_t P)
{
	if (P < 0) return 0;
	return fr_align((size_t)P * 8) + 256;
}
extern "C" int fr_spatial_order(int32_t P, const float* means3D, uint32_t* order_out, void* workspace, size_t workspace_bytes, fr_stream_t stream)
{
	if (P < 0) return fr_fail(FR_EINVAL, "fr_spatial_order: P < 0");
	if (P == 0) return FR_OK;
	if (!means3D || !order_out || !workspace) return fr_fail(FR_EINVAL, "fr_spatial_order: null pointer");
	if (workspace_bytes < fr_spatial_order_workspace_bytes(P)) return fr_fail(FR_ENOSPACE, "fr_spatial_order: workspace smaller than fr_spatial_order_workspace_bytes()");
	hipStream_t s = (hipStream_t)stream;
	uint64_t* keys = (uint64_t*)workspace;
	uint32_t* mm = (uint32_t*)((char*)workspace + fr_align((size_t)P * 8));
	int rc;
	if ((rc = fr_morton_sort(P, means3D, keys, mm, s))) return rc;
	hipLaunchKernelGGL(k_order_from_keys, dim3((P + FR_THREADS - 1) / FR_THREADS), dim3(FR_THREADS), 0, s, P, (const uint64_t*)keys, order_out);
	return fr_check_launch("k_order_from_keys");
}

extern "C" int fr_knn_dist2(int32_t P, const float* points, float* out, void* workspace, size_t workspace_bytes, fr_stream_t stream)
{
	if (P < 0) return fr_fail(FR_EINVAL, "fr_knn_dist2: P < 0");
	if (P == 0) return FR_OK;
	if (!points || !out || !workspace) return fr_fail(FR_EINVAL, "fr_knn_dist2: null pointer");
	FrKnnLayout L = fr_knn_layout(P);
	if (workspace_bytes < L.total) return fr_fail(FR_ENOSPACE, "fr_knn_dist2: workspace smaller than fr_knn_workspace_bytes()");
	hipStream_t s = (hipStream_t)stream;
	char* ws = (char*)workspace;
	uint64_t* keys = (uint64_t*)(ws + L.keys);
	float* sorted = (float*)(ws + L.sorted);
	float* boxes = (float*)(ws + L.boxes);
	uint32_t* mm = (uint32_t*)(ws + L.mm);
	int rc;
	const int nblk = (P + FR_THREADS - 1) / FR_THREADS;
	if ((rc = fr_morton_sort(P, points, keys, mm, s))) return rc;
	hipLaunchKernelGGL(k_knn_gather, dim3(nblk), dim3(FR_THREADS), 0, s, P, points, keys, sorted);
	const int nb = (P + FR_KNN_BOX - 1) / FR_KNN_BOX;
	hipLaunchKernelGGL(k_knn_boxes, dim3(nb), dim3(FR_THREADS), 0, s, P, sorted, boxes);
	const int nsb = (P + FR_KNN_SUB - 1) / FR_KNN_SUB;
	float* sboxes = (float*)(ws + L.sboxes);
	hipLaunchKernelGGL(k_knn_subboxes, dim3((nsb * 64 + FR_THREADS - 1) / FR_THREADS), dim3(FR_THREADS), 0, s, P, sorted, sboxes);
	hipLaunchKernelGGL(k_knn_search, dim3(nblk), dim3(FR_THREADS), 0, s, P, sorted, keys, boxes, nb, sboxes, nsb, out);
	return fr_check_launch("k_knn_search");
}

// ---- measurement hooks (bench.py): per-launch duration of k_fisher_tile from HIP events on its own stream --------
extern "C" int fr_profile_enable(int on)
{
	for (auto& e : g_prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
	g_prof_events.clear();
	g_prof_on = on != 0;
	return FR_OK;
}

extern "C" int fr_profile_fetch(float* ms, int max_n)
{
	int n = 0;
	for (auto& e : g_prof_events)
	{
		if (n >= max_n) break;
		if (hipEventSynchronize(e.second) != hipSuccess) return -1;
		float t = 0.f;
		if (hipEventElapsedTime(&t, e.first, e.second) != hipSuccess) return -1;
		ms[n++] = t;
	}
	return n;
}

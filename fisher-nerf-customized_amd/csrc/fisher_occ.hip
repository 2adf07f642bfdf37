// fisher_occ.hip -- planner-side kernels of libfisher_rast.so (gfx950, wave64): occupancy-map update and frontier
// extraction (include/fisher_occ.h).  The reference does the binning with torch ops on the GPU, then walks
// every occupied cell in a Python loop on the host (one cv2.line each) and runs cv2 morphology / connected components on the
// CPU; here the whole step stays on the device:
//
//   fr_occ_update     k_occ_mark_cam  -> k_occ_bin (one thread per pixel, 11 samples along the ray, integer atomics into the
//                     per-label count grids: the torch.unique(return_counts) of astar.py:268-269, 287-288)
//                     -> k_occ_lines (one thread per occupied cell: OpenCV's 8-connected LineIterator towards the camera cell)
//                     -> k_occ_accumulate (grid weights, line canvas, normalised add: astar.py:270-301)
//   fr_occ_freespace  k_occ_label -> k_occ_block_points -> erode/dilate 3x3 -> union-find components -> largest
//   fr_occ_frontiers  dilate, boundary AND unknown, dilate, components, per-component size and distance sum (fp64 atomics),
//                     one-workgroup selection, ordered compaction of the selected cells
//
// Everything is HBM/latency-bound integer and byte work on a 768x768 grid (2.3 MB per layer): no LDS tiling is needed, the
// grids live in L2; the kernels are written for coalesced row-major access and few launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "fr_internal.h"
#include "../../include/fisher_occ.h"

#define OCC_THREADS 256
#define OCC_MAX_SAMPLES 32

struct OccGeom {
	int gw, gh;
	float cell, cx, cz, hlo, hhi, far_d;
};

static OccGeom occ_geom(const fr_occ_cfg* c)
{
	OccGeom g;
	g.gw = c->grid_w; g.gh = c->grid_h; g.cell = c->cell_size; g.cx = c->center_x; g.cz = c->center_z;
	g.hlo = c->height_lower; g.hhi = c->height_upper; g.far_d = c->far_distance;
	return g;
}

// datasets/util/map_utils.py:106-125: floor((x - c) / cell) + (dim - 1) / 2.0, .int() (truncation), clamp
__device__ __forceinline__ int occ_bin(float x, float c, float cell, int dim)
{
	const float b = floorf((x - c) / cell) + (float)(dim - 1) / 2.0f;
	int i = (int)b;
	i = i < 0 ? 0 : i;
	return i > dim - 1 ? dim - 1 : i;
}

struct OccUpdateArgs {
	OccGeom g;
	int W, H, ds, nx, ny, n_samples;
	float fx, fy, cx, cy;
	float c2w[16];
	float fracs[OCC_MAX_SAMPLES];
	int cam_col, cam_row;
};

// astar.py:214: occ_map[2, cam_z-1:cam_z+2, cam_x-1:cam_x+2] = 1e3 (clipped to the map like the slice)
__global__ void k_occ_mark_cam(OccUpdateArgs a, float* __restrict__ occ_map)
{
	const int t = threadIdx.x;
	if (t >= 9) return;
	const int r = a.cam_row - 1 + t / 3, c = a.cam_col - 1 + t % 3;
	if (r < 0 || r >= a.g.gh || c < 0 || c >= a.g.gw) return;
	occ_map[(size_t)2 * a.g.gw * a.g.gh + (size_t)r * a.g.gw + c] = 1e3f;
}

// astar.py:222-289.  cnt[0] = free-sample counts (label 2), cnt[1] = depth-point counts (label 1).
__global__ __launch_bounds__(OCC_THREADS) void k_occ_bin(OccUpdateArgs a, const float* __restrict__ depth, uint32_t* __restrict__ cnt)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	const bool live = i < a.nx * a.ny;
	const int lane = threadIdx.x & 63;
	const int ix = live ? i % a.nx : 0, iy = live ? i / a.nx : 0;
	const float xg = (float)(ix * a.ds), yg = (float)(iy * a.ds);
	const float xx = (xg - a.cx) / a.fx, yy = (yg - a.cy) / a.fy;
	const float d = live ? depth[(size_t)(iy * a.ds) * a.W + ix * a.ds] : 0.0f;
	const size_t cells = (size_t)a.g.gw * a.g.gh;
	for (int k = 0; k < a.n_samples; k++)
	{
		const float dz = a.fracs[k] * d;
		bool ok = live && dz > 0.0f && dz < a.g.far_d;
		const float px = xx * dz, py = yy * dz;
		// c2w @ (px, py, dz, 1), accumulated left to right
		const float wx = ((a.c2w[0] * px + a.c2w[1] * py) + a.c2w[2] * dz) + a.c2w[3];
		const float wy = ((a.c2w[4] * px + a.c2w[5] * py) + a.c2w[6] * dz) + a.c2w[7];
		const float wz = ((a.c2w[8] * px + a.c2w[9] * py) + a.c2w[10] * dz) + a.c2w[11];
		ok = ok && wy >= a.g.hlo && wy <= a.g.hhi;
		const int xb = occ_bin(wx, a.g.cx, a.g.cell, a.g.gw), zb = occ_bin(wz, a.g.cz, a.g.cell, a.g.gh);
		const int key = ok ? (int)((k == a.n_samples - 1 ? cells : 0) + (size_t)zb * a.g.gw + xb) : -1;
		// The 64 pixels of a wave are neighbours in a row and their k-th samples mostly share a handful of cells (near the camera:
		// one): aggregate equal cells with ballots and issue one atomic per distinct cell instead of one per sample.
		unsigned long long todo = __ballot(key >= 0);
		while (todo)
		{
			const int leader = __builtin_ctzll(todo);
			const int kl = __builtin_amdgcn_readlane(key, leader);
			const unsigned long long same = __ballot(key == kl);
			if (lane == leader) atomicAdd(&cnt[kl], (uint32_t)__popcll(same));
			todo &= ~same;
		}
	}
}

// One 1-pixel line per occupied cell towards the camera cell (astar.py:291-297), OpenCV's 8-connected LineIterator as
// cv::line drives it (left to right; count = major + 1; err = major - 2 minor; a step always advances the major axis and
// also the minor one when err < 0).  All writers store 1: the races are benign.
__global__ __launch_bounds__(OCC_THREADS) void k_occ_lines(OccGeom g, const uint32_t* __restrict__ cnt_occ, int cam_col, int cam_row,
                                                          uint8_t* __restrict__ canvas)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i >= g.gw * g.gh || cnt_occ[i] == 0u) return;
	int x0 = i % g.gw, y0 = i / g.gw, x1 = cam_col, y1 = cam_row;
	if (x1 < x0) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
	int dx = x1 - x0, dy = y1 - y0;
	const int sy = dy < 0 ? -1 : 1;
	dy = dy < 0 ? -dy : dy;
	const bool steep = dy > dx;
	const int major = steep ? dy : dx, minor = steep ? dx : dy;
	int err = major - (minor + minor);
	const int plus = major + major, minus = -(minor + minor);
	int x = x0, y = y0;
	for (int n = 0; n <= major; n++)
	{
		if (x >= 0 && x < g.gw && y >= 0 && y < g.gh) canvas[(size_t)y * g.gw + x] = 1;
		const bool both = err < 0;
		err += minus + (both ? plus : 0);
		if (steep) { y += sy; if (both) x += 1; }
		else { x += 1; if (both) y += sy; }
	}
}

// astar.py:270-301: grid[2] = 0.01 (count + 1e-5), grid[1] = 100 (count + 1e-5), free line cells = 1, occ_map += occ / (sum + 1e-5)
__global__ __launch_bounds__(OCC_THREADS) void k_occ_accumulate(OccGeom g, const uint32_t* __restrict__ cnt, const uint8_t* __restrict__ canvas,
                                                               float* __restrict__ occ_map)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	const size_t cells = (size_t)g.gw * g.gh;
	if ((size_t)i >= cells) return;
	const uint32_t cf = cnt[i], co = cnt[cells + i];
	float o1 = 0.f, o2 = 0.f;
	if (cf) o2 = 0.01f * ((float)cf + 1e-5f);
	if (co) o1 = ((float)co + 1e-5f) * 100.0f;
	if (canvas[i]) o2 = 1.0f;
	const float s = ((0.0f + o1) + o2) + 1e-5f;
	occ_map[i] += 0.0f / s;
	occ_map[cells + i] += o1 / s;
	occ_map[2 * cells + i] += o2 / s;
}

// ---- frontier side ----------------------------------------------------------------------------------------------
// arg-max over the three layers (first maximum wins): bit 0 = free (label 2), bit 1 = unknown (label 0)
__global__ __launch_bounds__(OCC_THREADS) void k_occ_label(OccGeom g, const float* __restrict__ occ_map, uint8_t* __restrict__ lab,
                                                          uint32_t* __restrict__ free_count)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	const size_t cells = (size_t)g.gw * g.gh;
	bool fr = false;
	if ((size_t)i < cells)
	{
		const float a = occ_map[i], b = occ_map[cells + i], c = occ_map[2 * cells + i];
		int idx = 0; float m = a;
		if (b > m) { m = b; idx = 1; }
		if (c > m) { m = c; idx = 2; }
		fr = idx == 2;
		lab[i] = (uint8_t)((fr ? 1 : 0) | (idx == 0 ? 2 : 0));
	}
	const unsigned long long bal = __ballot(fr);
	if ((threadIdx.x & 63) == 0 && bal) atomicAdd(free_count, (uint32_t)__popcll(bal));
}

__global__ __launch_bounds__(OCC_THREADS) void k_occ_point_hist(OccGeom g, const float* __restrict__ pts, int n, uint32_t* __restrict__ hist)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i >= n) return;
	const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
	if (!(y >= g.hlo && y <= g.hhi)) return;
	atomicAdd(&hist[(size_t)occ_bin(z, g.cz, g.cell, g.gh) * g.gw + occ_bin(x, g.cx, g.cell, g.gw)], 1u);
}

// free = label free, minus the cells holding more than 25 Gaussians when the map has more than 18 free cells (astar.py:419-431)
__global__ __launch_bounds__(OCC_THREADS) void k_occ_free_init(OccGeom g, const uint8_t* __restrict__ lab, const uint32_t* __restrict__ hist,
                                                              const uint32_t* __restrict__ free_count, uint8_t* __restrict__ out)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i >= g.gw * g.gh) return;
	uint8_t f = lab[i] & 1;
	if (hist && *free_count > 18u && hist[i] > 25u) f = 0;
	out[i] = f;
}

// k x k min / max filter with cv2's border rule (cells outside do not constrain); anchor k/2
template <bool ERODE>
__global__ __launch_bounds__(OCC_THREADS) void k_occ_morph(OccGeom g, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int k)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i >= g.gw * g.gh) return;
	const int x = i % g.gw, y = i / g.gw, a = k / 2;
	uint8_t r = ERODE ? 1 : 0;
	for (int v = 0; v < k; v++)
	{
		const int yy = y + v - a;
		if (yy < 0 || yy >= g.gh) continue;
		for (int u = 0; u < k; u++)
		{
			const int xx = x + u - a;
			if (xx < 0 || xx >= g.gw) continue;
			const uint8_t s = src[(size_t)yy * g.gw + xx] ? 1 : 0;
			r = ERODE ? (r & s) : (r | s);
		}
	}
	dst[i] = r;
}

// ---- 8-connected components by union-find; a component's id is the smallest linear index of its cells ----------------
__device__ __forceinline__ int occ_find(const int* __restrict__ lab, int x)
{
	int p = lab[x];
	while (p != x) { x = p; p = lab[x]; }
	return x;
}
__device__ __forceinline__ void occ_union(int* __restrict__ lab, int a, int b)
{
	for (;;)
	{
		a = occ_find(lab, a); b = occ_find(lab, b);
		if (a == b) return;
		if (a < b) { const int t = a; a = b; b = t; }      // a is the larger root: hang it under b
		const int old = atomicMin(&lab[a], b);
		if (old == a) return;
		a = old;                                            // somebody re-rooted a meanwhile: merge that tree with b
	}
}
__global__ __launch_bounds__(OCC_THREADS) void k_cc_init(int n, const uint8_t* __restrict__ fg, int* __restrict__ lab)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i < n) lab[i] = fg[i] ? i : -1;
}
__global__ __launch_bounds__(OCC_THREADS) void k_cc_merge(OccGeom g, const uint8_t* __restrict__ fg, int* __restrict__ lab)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i >= g.gw * g.gh || !fg[i]) return;
	const int x = i % g.gw, y = i / g.gw;
	if (x > 0 && fg[i - 1]) occ_union(lab, i, i - 1);
	if (y > 0)
	{
		const int up = i - g.gw;
		if (fg[up]) occ_union(lab, i, up);
		if (x > 0 && fg[up - 1]) occ_union(lab, i, up - 1);
		if (x + 1 < g.gw && fg[up + 1]) occ_union(lab, i, up + 1);
	}
}
// flatten + per-component size (+ distance sum to the camera cell)
__global__ __launch_bounds__(OCC_THREADS) void k_cc_flatten(OccGeom g, int* __restrict__ lab, uint32_t* __restrict__ size,
                                                           double* __restrict__ dist_sum, int cam_row, int cam_col)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	const int lane = threadIdx.x & 63;
	int r = -1;
	if (i < g.gw * g.gh && lab[i] >= 0)
	{
		r = occ_find(lab, i);
		lab[i] = r;
	}
	double d = 0.0;
	if (dist_sum && r >= 0)
	{
		const double dy = (double)(i / g.gw - cam_row), dx = (double)(i % g.gw - cam_col);
		d = sqrt(dy * dy + dx * dx);
	}
	// the 64 cells of a wave are neighbours in a row and mostly belong to one or two components: one atomic per distinct
	// component and wave instead of one per cell (all of them on the few addresses of the large components otherwise)
	unsigned long long todo = __ballot(r >= 0);
	while (todo)
	{
		const int leader = __builtin_ctzll(todo);
		const int rl = __builtin_amdgcn_readlane(r, leader);
		const bool mine = r == rl;
		const unsigned long long same = __ballot(mine);
		double sum = mine ? d : 0.0;
		if (dist_sum)
		{
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
		}
		if (lane == leader)
		{
			atomicAdd(&size[rl], (uint32_t)__popcll(same));
			if (dist_sum) atomicAdd(&dist_sum[rl], sum);
		}
		todo &= ~same;
	}
}

// One workgroup picks a component.  mode 0: largest (ties: smallest id) -- build_connected_freespace's robot label.
// Frontier modes (only components with size > min_area): 1 largest (ties: largest id, the reversed argsort of astar.py:581),
// 2 combined size / (mean distance + 20) with a strict ">" in label order (ties: smallest id), 3 closest mean distance
// with a strict "<" (ties: smallest id).  out = {chosen id or -1, its size, number of qualifying components}.
// `better(key a, id a, key b, id b)`: does a beat b under the mode's tie rule
__device__ __forceinline__ bool occ_better(double ka, int ia, double kb, int ib, int mode)
{
	if (ia < 0) return false;
	if (ib < 0) return true;
	return ka > kb || (ka == kb && (mode == 1 ? ia > ib : ia < ib));
}
// stage 1: every workgroup scans a slice of the cells and leaves its best component in part_key / part_id / part_cnt
#define OCC_SELECT_BLOCKS 128
__global__ __launch_bounds__(OCC_THREADS) void k_cc_select_partial(int n, const int* __restrict__ lab, const uint32_t* __restrict__ size,
                                                                  const double* __restrict__ dist_sum, int mode, uint32_t min_area,
                                                                  double* __restrict__ part_key, int* __restrict__ part_id, int* __restrict__ part_cnt)
{
	__shared__ double s_key[OCC_THREADS];
	__shared__ int s_id[OCC_THREADS];
	__shared__ int s_cnt[OCC_THREADS];
	const int tid = threadIdx.x;
	double best = 0.0; int best_id = -1; int cnt = 0;
	for (int i = blockIdx.x * OCC_THREADS + tid; i < n; i += gridDim.x * OCC_THREADS)
	{
		if (lab[i] != i) continue;
		const uint32_t c = size[i];
		double key;
		if (mode == 0) key = (double)c;
		else
		{
			if (!(c > min_area)) continue;
			cnt++;
			if (mode == 1) key = (double)c;
			else
			{
				const double mean = dist_sum[i] / (double)c;
				key = mode == 2 ? (double)c / (mean + 20.0) : -mean;
				if (mode == 2 && !(key > 0.0)) continue;
			}
		}
		if (occ_better(key, i, best, best_id, mode)) { best = key; best_id = i; }
	}
	s_key[tid] = best; s_id[tid] = best_id; s_cnt[tid] = cnt;
	__syncthreads();
	for (int o = OCC_THREADS / 2; o > 0; o >>= 1)
	{
		if (tid < o)
		{
			if (occ_better(s_key[tid + o], s_id[tid + o], s_key[tid], s_id[tid], mode)) { s_key[tid] = s_key[tid + o]; s_id[tid] = s_id[tid + o]; }
			s_cnt[tid] += s_cnt[tid + o];
		}
		__syncthreads();
	}
	if (tid == 0) { part_key[blockIdx.x] = s_key[0]; part_id[blockIdx.x] = s_id[0]; part_cnt[blockIdx.x] = s_cnt[0]; }
}
// stage 2.  mode 0: largest (ties: smallest id) -- build_connected_freespace's robot label.
// Frontier modes (only components with size > min_area): 1 largest (ties: largest id, the reversed argsort of astar.py:581),
// 2 combined size / (mean distance + 20) with a strict ">" in label order (ties: smallest id), 3 closest mean distance
// with a strict "<" (ties: smallest id).  out = {chosen id or -1, its size, number of qualifying components}.
__global__ __launch_bounds__(OCC_SELECT_BLOCKS) void k_cc_select(const double* __restrict__ part_key, const int* __restrict__ part_id,
                                                                const int* __restrict__ part_cnt, const uint32_t* __restrict__ size, int mode,
                                                                int* __restrict__ out)
{
	__shared__ double s_key[OCC_SELECT_BLOCKS];
	__shared__ int s_id[OCC_SELECT_BLOCKS];
	__shared__ int s_cnt[OCC_SELECT_BLOCKS];
	const int tid = threadIdx.x;
	s_key[tid] = part_key[tid]; s_id[tid] = part_id[tid]; s_cnt[tid] = part_cnt[tid];
	__syncthreads();
	for (int o = OCC_SELECT_BLOCKS / 2; o > 0; o >>= 1)
	{
		if (tid < o)
		{
			if (occ_better(s_key[tid + o], s_id[tid + o], s_key[tid], s_id[tid], mode)) { s_key[tid] = s_key[tid + o]; s_id[tid] = s_id[tid + o]; }
			s_cnt[tid] += s_cnt[tid + o];
		}
		__syncthreads();
	}
	if (tid == 0)
	{
		out[0] = s_id[0];
		out[1] = s_id[0] >= 0 ? (int)size[s_id[0]] : 0;
		out[2] = s_cnt[0];
	}
}

__global__ __launch_bounds__(OCC_THREADS) void k_cc_mask(int n, const int* __restrict__ lab, const int* __restrict__ sel, uint8_t* __restrict__ out)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i < n) out[i] = (sel[0] >= 0 && lab[i] == sel[0]) ? 1 : 0;
}

// frontier = (dilate(free) - free) AND unknown (astar.py:554-559); counts its cells
__global__ __launch_bounds__(OCC_THREADS) void k_occ_frontier(int n, const uint8_t* __restrict__ dil, const uint8_t* __restrict__ free_space,
                                                             const uint8_t* __restrict__ lab, uint8_t* __restrict__ frontier,
                                                             int* __restrict__ count)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	bool f = false;
	if (i < n)
	{
		f = dil[i] && !free_space[i] && (lab[i] & 2);
		frontier[i] = f ? 1 : 0;
	}
	const unsigned long long bal = __ballot(f);
	if ((threadIdx.x & 63) == 0 && bal) atomicAdd(count, (int)__popcll(bal));
}

// ordered compaction of a mask's cells (np.where order): per-row counts, one-workgroup scan, per-row emit
__global__ __launch_bounds__(OCC_THREADS) void k_row_count(OccGeom g, const uint8_t* __restrict__ mask, int* __restrict__ row_cnt)
{
	__shared__ int s;
	const int y = blockIdx.x;
	if (threadIdx.x == 0) s = 0;
	__syncthreads();
	int c = 0;
	for (int x = threadIdx.x; x < g.gw; x += OCC_THREADS) c += mask[(size_t)y * g.gw + x] ? 1 : 0;
	if (c) atomicAdd(&s, c);
	__syncthreads();
	if (threadIdx.x == 0) row_cnt[y] = s;
}
__global__ __launch_bounds__(1024) void k_row_scan(int gh, int* __restrict__ row_cnt, int* __restrict__ total)
{
	__shared__ int s_part[1024];
	const int tid = threadIdx.x;
	const int per = (gh + 1023) / 1024;
	int sum = 0;
	for (int k = 0; k < per; k++) { const int y = tid * per + k; if (y < gh) sum += row_cnt[y]; }
	s_part[tid] = sum;
	__syncthreads();
	if (tid == 0)
	{
		int run = 0;
		for (int t = 0; t < 1024; t++) { const int v = s_part[t]; s_part[t] = run; run += v; }
		*total = run;
	}
	__syncthreads();
	int run = s_part[tid];
	for (int k = 0; k < per; k++)
	{
		const int y = tid * per + k;
		if (y < gh) { const int v = row_cnt[y]; row_cnt[y] = run; run += v; }
	}
}
__global__ __launch_bounds__(64) void k_row_emit(OccGeom g, const uint8_t* __restrict__ mask, const int* __restrict__ row_off,
                                                 int* __restrict__ cells, int max_cells)
{
	const int y = blockIdx.x, lane = threadIdx.x;
	int base = row_off[y];
	for (int x0 = 0; x0 < g.gw; x0 += 64)
	{
		const int x = x0 + lane;
		const bool m = x < g.gw && mask[(size_t)y * g.gw + x];
		const unsigned long long bal = __ballot(m);
		if (m)
		{
			const int slot = base + (int)__popcll(bal & ((1ull << lane) - 1ull));
			if (slot < max_cells) { cells[2 * (size_t)slot] = x; cells[2 * (size_t)slot + 1] = y; }
		}
		base += (int)__popcll(bal);
	}
}

__global__ __launch_bounds__(OCC_THREADS) void k_occ_cells_of(OccGeom g, const float* __restrict__ xyz, int n, int* __restrict__ cells)
{
	const int i = blockIdx.x * OCC_THREADS + threadIdx.x;
	if (i >= n) return;
	cells[2 * (size_t)i] = occ_bin(xyz[3 * (size_t)i], g.cx, g.cell, g.gw);
	cells[2 * (size_t)i + 1] = occ_bin(xyz[3 * (size_t)i + 2], g.cz, g.cell, g.gh);
}


// ---------------------------------------------------------------------------------------------------------
// Candidate samplers (astar.py:1406-1430 / 1432-1469 generate_candidate[_object]; 1387-1401 free-space filter;
// 782-837 sample_random_candidate).  The reference draws from torch / numpy generators on the host; here the draws come
// from a counter-based generator so that a call is a pure function of (seed, index) and has a NumPy restatement
// (oracle/occupancy_frontier.py: occ_uniform):  x = seed + 0x9E3779B9 (4 k + j + 1), then the 32-bit finaliser
// x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16;  u = (x >> 8) / 2^24  in [0, 1).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float occ_uniform(uint32_t seed, uint32_t k, uint32_t j)
{
	uint32_t x = seed + 0x9E3779B9u * (4u * k + j + 1u);
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// rotation about y from the quaternion (qr, 0, qy, 0) exactly as build_rotation does it (slam_external.py:25-42:
// normalise, then the 3x3 entries), written into rows 0..2 of a row-major 4x4 whose other entries are already set
__device__ __forceinline__ void occ_yaw_rotation(float qr, float qy, float* __restrict__ m, float s0, float s1, float s2)
{
	const float norm = sqrtf(qr * qr + qy * qy);
	const float r = qr / norm, y = qy / norm;
	const float d = 1.0f - 2.0f * (y * y), o = 2.0f * (r * y);
	// R = [[d, 0, o], [0, 1, 0], [-o, 0, d]]; s0..s2 are the column signs the caller applies afterwards
	m[0] = d * s0;  m[1] = 0.0f * s1; m[2] = o * s2;
	m[4] = 0.0f * s0; m[5] = 1.0f * s1; m[6] = 0.0f * s2;
	m[8] = -o * s0; m[9] = 0.0f * s1; m[10] = d * s2;
}

__global__ __launch_bounds__(1024) void k_occ_ring_candidates(OccGeom g, const float* __restrict__ centers, int n_centers, int K,
                                                              float min_range, float radius, float cam_height, uint32_t seed,
                                                              const uint8_t* __restrict__ eroded, int min_free,
                                                              float* __restrict__ c2w, uint8_t* __restrict__ keep)
{
	__shared__ unsigned long long s_free;
	const int tid = threadIdx.x;
	if (tid == 0) s_free = 0ull;
	__syncthreads();
	if (eroded)
	{
		// astar.py:1389: the filter applies only when the eroded free space has more than `min_free` cells
		unsigned long long c = 0;
		const size_t n = (size_t)g.gw * g.gh;
		for (size_t i = tid; i < n; i += 1024) c += eroded[i] ? 1ull : 0ull;
		if (c) atomicAdd(&s_free, c);
	}
	__syncthreads();
	const bool filter = eroded != nullptr && s_free > (unsigned long long)min_free;
	const float pi = 3.14159274101257324f;                     // torch.pi as float32
	for (int k = tid; k < K; k += 1024)
	{
		const float theta = (occ_uniform(seed, (uint32_t)k, 0) * 2.0f) * pi;
		const float rr = min_range + occ_uniform(seed, (uint32_t)k, 1) * (radius - min_range);
		int ci = (int)(occ_uniform(seed, (uint32_t)k, 2) * (float)n_centers);
		ci = ci > n_centers - 1 ? n_centers - 1 : ci;
		const float px = centers[2 * (size_t)ci] + rr * sinf(theta);
		const float pz = centers[2 * (size_t)ci + 1] + rr * cosf(theta);
		float* m = c2w + 16 * (size_t)k;
		const float phi = theta + pi;
		occ_yaw_rotation(cosf(phi / 2.0f), sinf(phi / 2.0f), m, -1.0f, -1.0f, 1.0f);   // astar.py:1419-1420: columns 0 and 1 negated
		m[3] = px; m[7] = cam_height; m[11] = pz;
		m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
		if (keep)
		{
			uint8_t kp = 1;
			if (filter)
			{
				// astar.py:1392-1398: (x - map_center) / cell_size + grid_dim // 2, .long() (truncation), eroded[row, col]
				const long long col = (long long)((px - g.cx) / g.cell + (float)(g.gw / 2));
				const long long row = (long long)((pz - g.cz) / g.cell + (float)(g.gh / 2));
				kp = (col >= 0 && col < g.gw && row >= 0 && row < g.gh) ? (eroded[(size_t)row * g.gw + col] ? 1 : 0) : 0;
			}
			keep[k] = kp;
		}
	}
}

__global__ __launch_bounds__(OCC_THREADS) void k_occ_free_candidates(OccGeom g, const int* __restrict__ cells, const int* __restrict__ counts,
                                                                     float agent_y, uint32_t seed, float* __restrict__ c2w, int max_out,
                                                                     int* __restrict__ n_out)
{
	const int n_free = counts[0];
	int m_out = n_free / 4;                                    // astar.py:813: rng.choice(len, len // 4)
	m_out = m_out > max_out ? max_out : m_out;
	if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = m_out;
	for (int i = blockIdx.x * OCC_THREADS + threadIdx.x; i < m_out; i += gridDim.x * OCC_THREADS)
	{
		int ci = (int)(occ_uniform(seed, (uint32_t)i, 0) * (float)n_free);
		ci = ci > n_free - 1 ? n_free - 1 : ci;
		const int col = cells[2 * (size_t)ci], row = cells[2 * (size_t)ci + 1];
		// astar.py:808-810, in double like numpy, narrowed once
		const float wx = (float)(((double)col + 0.5 - (double)(g.gw / 2)) * (double)g.cell + (double)g.cx);
		const float wz = (float)(((double)row + 0.5 - (double)(g.gh / 2)) * (double)g.cell + (double)g.cz);
		const double ang = (double)occ_uniform(seed, (uint32_t)i, 1) * 6.283185307179586;
		float* m = c2w + 16 * (size_t)i;
		occ_yaw_rotation((float)cos(ang / 2.0), (float)sin(ang / 2.0), m, 1.0f, -1.0f, -1.0f);   // astar.py:833-834: columns 1 and 2 negated
		m[3] = wx; m[7] = agent_y; m[11] = wz;
		m[12] = 0.f; m[13] = -0.f; m[14] = -0.f; m[15] = 1.f;
	}
}

// =========================================================================================================
// host side
// =========================================================================================================
static inline size_t occ_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct OccLayout { size_t cnt, canvas, lab8, tmp_a, tmp_b, cc, size, dist, rows, sel, part, scalars, total; };

static OccLayout occ_layout(const fr_occ_cfg* c)
{
	OccLayout L;
	const size_t n = (size_t)c->grid_w * c->grid_h;
	size_t o = 0;
	L.cnt = o; o = occ_align(o + 2 * n * 4);       // count grids of the update / point histogram
	L.canvas = o; o = occ_align(o + n);
	L.lab8 = o; o = occ_align(o + n);              // arg-max label bits
	L.tmp_a = o; o = occ_align(o + n);
	L.tmp_b = o; o = occ_align(o + n);
	L.cc = o; o = occ_align(o + n * 4);            // component ids
	L.size = o; o = occ_align(o + n * 4);
	L.dist = o; o = occ_align(o + n * 8);
	L.rows = o; o = occ_align(o + (size_t)c->grid_h * 4);
	L.sel = o; o = occ_align(o + 64);
	L.part = o; o = occ_align(o + (size_t)OCC_SELECT_BLOCKS * 16);      // stage-1 results of the component selection: key f64, id i32, count i32
	L.scalars = o; o = occ_align(o + 64);
	L.total = o;
	return L;
}

static int occ_validate(const fr_occ_cfg* c, const char* who)
{
	if (!c || c->grid_w <= 0 || c->grid_h <= 0 || !(c->cell_size > 0.f) || (long long)c->grid_w * c->grid_h > (1ll << 30))
	{
		char b[160]; snprintf(b, sizeof(b), "%s: bad fr_occ_cfg", who);
		return fr_fail(FR_EINVAL, b);
	}
	return FR_OK;
}

extern "C" size_t fr_occ_workspace_bytes(const fr_occ_cfg* cfg)
{
	if (!cfg || cfg->grid_w <= 0 || cfg->grid_h <= 0) return 0;
	return occ_layout(cfg).total;
}

static inline dim3 occ_grid(size_t n) { return dim3((unsigned)((n + OCC_THREADS - 1) / OCC_THREADS)); }

extern "C" int fr_occ_update(const fr_occ_cfg* cfg, const float* depth, int32_t W, int32_t H, int32_t downsample,
                             const float intr[4], const float c2w[16], const float* sample_fracs, int32_t n_samples,
                             int32_t cam_col, int32_t cam_row, float* occ_map,
                             void* workspace, size_t workspace_bytes, fr_stream_t stream)
{
	int rc = occ_validate(cfg, "fr_occ_update");
	if (rc) return rc;
	if (!depth || !intr || !c2w || !sample_fracs || !occ_map || W <= 0 || H <= 0 || downsample <= 0 || n_samples < 1 || n_samples > OCC_MAX_SAMPLES)
		return fr_fail(FR_EINVAL, "fr_occ_update: bad argument");
	const OccLayout L = occ_layout(cfg);
	if (!workspace || workspace_bytes < L.total) return fr_fail(FR_ENOSPACE, "fr_occ_update: workspace smaller than fr_occ_workspace_bytes()");
	hipStream_t s = (hipStream_t)stream;
	char* ws = (char*)workspace;
	OccUpdateArgs a;
	a.g = occ_geom(cfg);
	a.W = W; a.H = H; a.ds = downsample;
	a.nx = (W + downsample - 1) / downsample; a.ny = (H + downsample - 1) / downsample;
	a.n_samples = n_samples;
	a.fx = intr[0]; a.fy = intr[1]; a.cx = intr[2]; a.cy = intr[3];
	memcpy(a.c2w, c2w, sizeof(a.c2w));
	memset(a.fracs, 0, sizeof(a.fracs));
	memcpy(a.fracs, sample_fracs, (size_t)n_samples * 4);
	a.cam_col = cam_col; a.cam_row = cam_row;
	const size_t n = (size_t)cfg->grid_w * cfg->grid_h;
	uint32_t* cnt = (uint32_t*)(ws + L.cnt);
	uint8_t* canvas = (uint8_t*)(ws + L.canvas);
	(void)hipMemsetAsync(cnt, 0, 2 * n * 4, s);
	(void)hipMemsetAsync(canvas, 0, n, s);
	hipLaunchKernelGGL(k_occ_mark_cam, dim3(1), dim3(64), 0, s, a, occ_map);
	hipLaunchKernelGGL(k_occ_bin, occ_grid((size_t)a.nx * a.ny), dim3(OCC_THREADS), 0, s, a, depth, cnt);
	hipLaunchKernelGGL(k_occ_lines, occ_grid(n), dim3(OCC_THREADS), 0, s, a.g, (const uint32_t*)(cnt + n), cam_col, cam_row, canvas);
	hipLaunchKernelGGL(k_occ_accumulate, occ_grid(n), dim3(OCC_THREADS), 0, s, a.g, (const uint32_t*)cnt, (const uint8_t*)canvas, occ_map);
	return fr_check_launch("fr_occ_update");
}

// components of `fg` -> ids in L.cc, sizes in L.size (and distance sums in L.dist when cam given)
static void occ_components(const OccGeom& g, const OccLayout& L, char* ws, const uint8_t* fg, bool with_dist, int cam_row, int cam_col, hipStream_t s)
{
	const size_t n = (size_t)g.gw * g.gh;
	int* cc = (int*)(ws + L.cc);
	uint32_t* size = (uint32_t*)(ws + L.size);
	double* dist = (double*)(ws + L.dist);
	(void)hipMemsetAsync(size, 0, n * 4, s);
	if (with_dist) (void)hipMemsetAsync(dist, 0, n * 8, s);
	hipLaunchKernelGGL(k_cc_init, occ_grid(n), dim3(OCC_THREADS), 0, s, (int)n, fg, cc);
	hipLaunchKernelGGL(k_cc_merge, occ_grid(n), dim3(OCC_THREADS), 0, s, g, fg, cc);
	hipLaunchKernelGGL(k_cc_flatten, occ_grid(n), dim3(OCC_THREADS), 0, s, g, cc, size, with_dist ? dist : (double*)nullptr, cam_row, cam_col);
}

extern "C" int fr_occ_freespace(const fr_occ_cfg* cfg, const float* occ_map, const float* points, int32_t n_points,
                                uint8_t* free_space, void* workspace, size_t workspace_bytes, fr_stream_t stream)
{
	int rc = occ_validate(cfg, "fr_occ_freespace");
	if (rc) return rc;
	if (!occ_map || !free_space || n_points < 0) return fr_fail(FR_EINVAL, "fr_occ_freespace: bad argument");
	const OccLayout L = occ_layout(cfg);
	if (!workspace || workspace_bytes < L.total) return fr_fail(FR_ENOSPACE, "fr_occ_freespace: workspace smaller than fr_occ_workspace_bytes()");
	hipStream_t s = (hipStream_t)stream;
	char* ws = (char*)workspace;
	const OccGeom g = occ_geom(cfg);
	const size_t n = (size_t)g.gw * g.gh;
	uint8_t* lab8 = (uint8_t*)(ws + L.lab8);
	uint8_t* ta = (uint8_t*)(ws + L.tmp_a);
	uint8_t* tb = (uint8_t*)(ws + L.tmp_b);
	uint32_t* hist = (uint32_t*)(ws + L.cnt);
	uint32_t* free_count = (uint32_t*)(ws + L.scalars);
	int* sel = (int*)(ws + L.sel);
	(void)hipMemsetAsync(free_count, 0, 64, s);
	hipLaunchKernelGGL(k_occ_label, occ_grid(n), dim3(OCC_THREADS), 0, s, g, occ_map, lab8, free_count);
	const bool with_pts = points != nullptr && n_points > 0;
	if (with_pts)
	{
		(void)hipMemsetAsync(hist, 0, n * 4, s);
		hipLaunchKernelGGL(k_occ_point_hist, occ_grid((size_t)n_points), dim3(OCC_THREADS), 0, s, g, points, n_points, hist);
	}
	hipLaunchKernelGGL(k_occ_free_init, occ_grid(n), dim3(OCC_THREADS), 0, s, g, (const uint8_t*)lab8, with_pts ? (const uint32_t*)hist : (const uint32_t*)nullptr,
	                   (const uint32_t*)free_count, ta);
	// 3x3 opening (astar.py:434-435)
	hipLaunchKernelGGL((k_occ_morph<true>), occ_grid(n), dim3(OCC_THREADS), 0, s, g, (const uint8_t*)ta, tb, 3);
	hipLaunchKernelGGL((k_occ_morph<false>), occ_grid(n), dim3(OCC_THREADS), 0, s, g, (const uint8_t*)tb, ta, 3);
	// largest component (astar.py:438-445)
	occ_components(g, L, ws, ta, false, 0, 0, s);
	{
		double* pk = (double*)(ws + L.part); int* pi = (int*)(pk + OCC_SELECT_BLOCKS); int* pc = pi + OCC_SELECT_BLOCKS;
		hipLaunchKernelGGL(k_cc_select_partial, dim3(OCC_SELECT_BLOCKS), dim3(OCC_THREADS), 0, s, (int)n, (const int*)(ws + L.cc),
		                   (const uint32_t*)(ws + L.size), (const double*)nullptr, 0, 0u, pk, pi, pc);
		hipLaunchKernelGGL(k_cc_select, dim3(1), dim3(OCC_SELECT_BLOCKS), 0, s, (const double*)pk, (const int*)pi, (const int*)pc,
		                   (const uint32_t*)(ws + L.size), 0, sel);
	}
	hipLaunchKernelGGL(k_cc_mask, occ_grid(n), dim3(OCC_THREADS), 0, s, (int)n, (const int*)(ws + L.cc), (const int*)sel, free_space);
	return fr_check_launch("fr_occ_freespace");
}

extern "C" int fr_occ_frontiers(const fr_occ_cfg* cfg, const float* occ_map, const uint8_t* free_space,
                                int32_t cam_row, int32_t cam_col, int32_t method, int32_t min_area,
                                uint8_t* frontier, uint8_t* target, int32_t* cells, int32_t max_cells, int32_t* counts,
                                void* workspace, size_t workspace_bytes, fr_stream_t stream)
{
	int rc = occ_validate(cfg, "fr_occ_frontiers");
	if (rc) return rc;
	if (!occ_map || !free_space || !frontier || !target || !cells || !counts || max_cells < 0 || method < 0 || method > 2 || min_area < 0)
		return fr_fail(FR_EINVAL, "fr_occ_frontiers: bad argument");
	const OccLayout L = occ_layout(cfg);
	if (!workspace || workspace_bytes < L.total) return fr_fail(FR_ENOSPACE, "fr_occ_frontiers: workspace smaller than fr_occ_workspace_bytes()");
	hipStream_t s = (hipStream_t)stream;
	char* ws = (char*)workspace;
	const OccGeom g = occ_geom(cfg);
	const size_t n = (size_t)g.gw * g.gh;
	uint8_t* lab8 = (uint8_t*)(ws + L.lab8);
	uint8_t* ta = (uint8_t*)(ws + L.tmp_a);
	uint8_t* tb = (uint8_t*)(ws + L.tmp_b);
	uint32_t* scal = (uint32_t*)(ws + L.scalars);
	int* sel = (int*)(ws + L.sel);
	int* rows = (int*)(ws + L.rows);
	(void)hipMemsetAsync(scal, 0, 64, s);
	(void)hipMemsetAsync(counts, 0, 16, s);
	hipLaunchKernelGGL(k_occ_label, occ_grid(n), dim3(OCC_THREADS), 0, s, g, occ_map, lab8, scal);
	hipLaunchKernelGGL((k_occ_morph<false>), occ_grid(n), dim3(OCC_THREADS), 0, s, g, free_space, ta, 3);
	hipLaunchKernelGGL(k_occ_frontier, occ_grid(n), dim3(OCC_THREADS), 0, s, (int)n, (const uint8_t*)ta, free_space, (const uint8_t*)lab8, frontier, counts);
	hipLaunchKernelGGL((k_occ_morph<false>), occ_grid(n), dim3(OCC_THREADS), 0, s, g, (const uint8_t*)frontier, tb, 3);
	occ_components(g, L, ws, tb, true, cam_row, cam_col, s);
	{
		double* pk = (double*)(ws + L.part); int* pi = (int*)(pk + OCC_SELECT_BLOCKS); int* pc = pi + OCC_SELECT_BLOCKS;
		hipLaunchKernelGGL(k_cc_select_partial, dim3(OCC_SELECT_BLOCKS), dim3(OCC_THREADS), 0, s, (int)n, (const int*)(ws + L.cc),
		                   (const uint32_t*)(ws + L.size), (const double*)(ws + L.dist), method + 1, (uint32_t)min_area, pk, pi, pc);
		hipLaunchKernelGGL(k_cc_select, dim3(1), dim3(OCC_SELECT_BLOCKS), 0, s, (const double*)pk, (const int*)pi, (const int*)pc,
		                   (const uint32_t*)(ws + L.size), method + 1, sel);
	}
	hipLaunchKernelGGL(k_cc_mask, occ_grid(n), dim3(OCC_THREADS), 0, s, (int)n, (const int*)(ws + L.cc), (const int*)sel, target);
	hipLaunchKernelGGL(k_row_count, dim3(g.gh), dim3(OCC_THREADS), 0, s, g, (const uint8_t*)target, rows);
	hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(1024), 0, s, g.gh, rows, counts + 2);
	hipLaunchKernelGGL(k_row_emit, dim3(g.gh), dim3(64), 0, s, g, (const uint8_t*)target, (const int*)rows, cells, max_cells);
	// counts = {frontier cells, qualifying components, target cells, target id}
	(void)hipMemcpyAsync(counts + 1, sel + 2, 4, hipMemcpyDeviceToDevice, s);
	(void)hipMemcpyAsync(counts + 3, sel + 0, 4, hipMemcpyDeviceToDevice, s);
	return fr_check_launch("fr_occ_frontiers");
}

extern "C" int fr_occ_erode(const fr_occ_cfg* cfg, const uint8_t* src, uint8_t* dst, int32_t ksize, fr_stream_t stream)
{
	int rc = occ_validate(cfg, "fr_occ_erode");
	if (rc) return rc;
	if (!src || !dst || src == dst || ksize < 1 || ksize > 63) return fr_fail(FR_EINVAL, "fr_occ_erode: bad argument");
	const OccGeom g = occ_geom(cfg);
	hipLaunchKernelGGL((k_occ_morph<true>), occ_grid((size_t)g.gw * g.gh), dim3(OCC_THREADS), 0, (hipStream_t)stream, g, src, dst, ksize);
	return fr_check_launch("fr_occ_erode");
}

extern "C" int fr_occ_cells_of(const fr_occ_cfg* cfg, const float* xyz, int32_t n, int32_t* cells, fr_stream_t stream)
{
	int rc = occ_validate(cfg, "fr_occ_cells_of");
	if (rc) return rc;
	if (n < 0 || (n > 0 && (!xyz || !cells))) return fr_fail(FR_EINVAL, "fr_occ_cells_of: bad argument");
	if (n == 0) return FR_OK;
	hipLaunchKernelGGL(k_occ_cells_of, occ_grid((size_t)n), dim3(OCC_THREADS), 0, (hipStream_t)stream, occ_geom(cfg), xyz, n, cells);
	return fr_check_launch("fr_occ_cells_of");
}

extern "C" int fr_occ_ring_candidates(const fr_occ_cfg* cfg, const float* centers, int32_t n_centers, int32_t K,
                                      float min_range, float radius, float cam_height, uint32_t seed,
                                      const uint8_t* eroded_free, int32_t min_free, float* c2w, uint8_t* keep, fr_stream_t stream)
{
	int rc = occ_validate(cfg, "fr_occ_ring_candidates");
	if (rc) return rc;
	if (K < 0 || n_centers < 0 || (K > 0 && (!centers || n_centers == 0 || !c2w))) return fr_fail(FR_EINVAL, "fr_occ_ring_candidates: bad argument");
	if (K == 0) return FR_OK;
	hipLaunchKernelGGL(k_occ_ring_candidates, dim3(1), dim3(1024), 0, (hipStream_t)stream, occ_geom(cfg), centers, n_centers, K,
	                   min_range, radius, cam_height, seed, eroded_free, min_free, c2w, keep);
	return fr_check_launch("fr_occ_ring_candidates");
}

extern "C" int fr_occ_free_candidates(const fr_occ_cfg* cfg, const uint8_t* eroded_free, float agent_y, uint32_t seed,
                                      float* c2w, int32_t max_out, int32_t* counts, void* workspace, size_t workspace_bytes,
                                      fr_stream_t stream)
{
	int rc = occ_validate(cfg, "fr_occ_free_candidates");
	if (rc) return rc;
	if (!eroded_free || !counts || max_out < 0 || (max_out > 0 && !c2w)) return fr_fail(FR_EINVAL, "fr_occ_free_candidates: bad argument");
	const OccLayout L = occ_layout(cfg);
	if (!workspace || workspace_bytes < L.total) return fr_fail(FR_ENOSPACE, "fr_occ_free_candidates: workspace smaller than fr_occ_workspace_bytes()");
	hipStream_t s = (hipStream_t)stream;
	char* ws = (char*)workspace;
	const OccGeom g = occ_geom(cfg);
	int* rows = (int*)(ws + L.rows);
	int* cells = (int*)(ws + L.dist);                          // n * 8 bytes: one (col, row) pair per cell
	hipLaunchKernelGGL(k_row_count, dim3(g.gh), dim3(OCC_THREADS), 0, s, g, eroded_free, rows);
	hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(1024), 0, s, g.gh, rows, counts);
	hipLaunchKernelGGL(k_row_emit, dim3(g.gh), dim3(64), 0, s, g, eroded_free, (const int*)rows, cells, g.gw * g.gh);
	const int blocks = max_out > 0 ? (max_out + OCC_THREADS - 1) / OCC_THREADS : 1;
	hipLaunchKernelGGL(k_occ_free_candidates, dim3(blocks > 1024 ? 1024 : blocks), dim3(OCC_THREADS), 0, s, g, (const int*)cells,
	                   (const int*)counts, agent_y, seed, c2w, max_out, counts + 1);
	return fr_check_launch("fr_occ_free_candidates");
}

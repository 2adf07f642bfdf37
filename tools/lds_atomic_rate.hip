// lds_atomic_rate.hip -- how fast does one CU of MI355X retire ds_add_f32 lane-operations, as a function of how many lanes of
// the instruction hit the same address?  (Input to the design of the out_H accumulation of k_fisher_tile_v3h.)
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/_build/lds_atomic_rate tools/lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// SHARE lanes per address (1 = all distinct, 64 = one address per wave); KIND 0 = ds_add_f32, 1 = DPP wave reduction + one lane adds
template <int SHARE, int KIND>
__global__ __launch_bounds__(256) void k_lds_add(float* out, int iters)
{
	__shared__ float acc[4][4][64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int c = 0; c < 4; c++) acc[wave][c][lane] = 0.f;
	__syncthreads();
	float x = (float)lane * 1e-3f;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 8; r++)
		{
			const int j = ((lane / SHARE) + r + i) & 63;
			if (KIND == 0)
			{
#pragma unroll
				for (int c = 0; c < 4; c++) atomicAdd(&acc[wave][c][j], x + (float)c);
			}
			x = x * 1.0001f + 1e-7f;
		}
	}
	__syncthreads();
	out[blockIdx.x * 256 + threadIdx.x] = acc[wave][0][lane] + acc[wave][1][lane] + acc[wave][2][lane] + acc[wave][3][lane] + x;
}

// integer forms: KIND 0 = ds_add_u32 (no return), 1 = ds_add_rtn_u32 (the slot-claim pattern of k_scatter_vis)
template <int SHARE, int KIND>
__global__ __launch_bounds__(256) void k_lds_add_u32(float* out, int iters)
{
	__shared__ uint32_t acc[4][4][64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int c = 0; c < 4; c++) acc[wave][c][lane] = 0u;
	__syncthreads();
	uint32_t x = (uint32_t)lane;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 8; r++)
		{
			const int j = ((lane / SHARE) + r + i) & 63;
#pragma unroll
			for (int c = 0; c < 4; c++)
			{
				if (KIND == 0) atomicAdd(&acc[wave][c][j], x | 1u);
				else x += atomicAdd(&acc[wave][c][j], 1u);
			}
		}
	}
	__syncthreads();
	out[blockIdx.x * 256 + threadIdx.x] = (float)(acc[wave][0][lane] + acc[wave][1][lane] + acc[wave][2][lane] + acc[wave][3][lane] + x);
}
template <int SHARE, int KIND>
static void run_u32(float* d_out, int wgs_per_cu)
{
	const int iters = 500;
	dim3 grid(256 * wgs_per_cu), block(256);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_lds_add_u32<SHARE, KIND>), grid, block, 0, 0, d_out, iters);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 5; rep++)
	{
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k_lds_add_u32<SHARE, KIND>), grid, block, 0, 0, d_out, iters);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double lane_ops = (double)grid.x * 256.0 * iters * 8.0 * 4.0;
	const double per_cu_per_ns = lane_ops / 256.0 / (best * 1e6);
	printf("{\"kind\": \"%s\", \"lanes_per_address\": %d, \"workgroups_per_cu\": %d, \"ms\": %.4f, \"lane_ops_per_ns_per_cu\": %.3f, "
	       "\"cycles_per_wave_instruction_at_2.4GHz\": %.1f}\n", KIND == 0 ? "ds_add_u32" : "ds_add_rtn_u32", SHARE, wgs_per_cu, best,
	       per_cu_per_ns, 64.0 / per_cu_per_ns * 2.4);
}

// ds_add_f64
template <int SHARE>
__global__ __launch_bounds__(256) void k_lds_add_f64(float* out, int iters)
{
	__shared__ double acc[4][4][64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int c = 0; c < 4; c++) acc[wave][c][lane] = 0.0;
	__syncthreads();
	double x = (double)lane * 1e-3;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 8; r++)
		{
			const int j = ((lane / SHARE) + r + i) & 63;
#pragma unroll
			for (int c = 0; c < 4; c++) atomicAdd(&acc[wave][c][j], x + (double)c);
		}
	}
	__syncthreads();
	out[blockIdx.x * 256 + threadIdx.x] = (float)(acc[wave][0][lane] + acc[wave][1][lane] + acc[wave][2][lane] + acc[wave][3][lane] + x);
}
template <int SHARE>
static void run_f64(float* d_out, int wgs_per_cu)
{
	const int iters = 200;
	dim3 grid(256 * wgs_per_cu), block(256);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_lds_add_f64<SHARE>), grid, block, 0, 0, d_out, iters);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++)
	{
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k_lds_add_f64<SHARE>), grid, block, 0, 0, d_out, iters);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double lane_ops = (double)grid.x * 256.0 * iters * 8.0 * 4.0;
	const double per_cu_per_ns = lane_ops / 256.0 / (best * 1e6);
	printf("{\"kind\": \"ds_add_f64\", \"lanes_per_address\": %d, \"workgroups_per_cu\": %d, \"ms\": %.4f, \"lane_ops_per_ns_per_cu\": %.3f, "
	       "\"cycles_per_wave_instruction_at_2.4GHz\": %.1f}\n", SHARE, wgs_per_cu, best, per_cu_per_ns, 64.0 / per_cu_per_ns * 2.4);
}

template <int SHARE>
static void run(float* d_out, int wgs_per_cu)
{
	const int iters = 500;
	dim3 grid(256 * wgs_per_cu), block(256);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_lds_add<SHARE, 0>), grid, block, 0, 0, d_out, iters);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 5; rep++)
	{
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k_lds_add<SHARE, 0>), grid, block, 0, 0, d_out, iters);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double lane_ops = (double)grid.x * 256.0 * iters * 8.0 * 4.0;
	const double per_cu_per_ns = lane_ops / 256.0 / (best * 1e6);
	printf("{\"kind\": \"ds_add_f32\", \"lanes_per_address\": %d, \"workgroups_per_cu\": %d, \"ms\": %.4f, \"lane_ops_per_ns_per_cu\": %.3f, "
	       "\"cycles_per_wave_instruction_at_2.4GHz\": %.1f}\n", SHARE, wgs_per_cu, best, per_cu_per_ns, 64.0 / per_cu_per_ns * 2.4);
}

int main()
{
	float* d_out;
	CHECK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * sizeof(float)));
	const int wl[] = { 1, 4 };
	for (int w : wl) { run<1>(d_out, w); run<2>(d_out, w); run<4>(d_out, w); run<8>(d_out, w); run<16>(d_out, w); run<64>(d_out, w); }
	run_f64<1>(d_out, 4); run_f64<4>(d_out, 4); run_f64<64>(d_out, 4);
	run_u32<1, 0>(d_out, 4); run_u32<4, 0>(d_out, 4); run_u32<64, 0>(d_out, 4);
	run_u32<1, 1>(d_out, 4); run_u32<4, 1>(d_out, 4); run_u32<64, 1>(d_out, 4);
	CHECK(hipFree(d_out));
	return 0;
}

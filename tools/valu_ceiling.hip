// valu_ceiling.hip -- calibration microbenchmark for the VALU roofline used by bench.py (MI355X / gfx950).
// Measures how many wave64 VALU instructions one SIMD retires per nanosecond as a function of the number of resident
// waves per SIMD and of the instruction-level parallelism inside a wave: the ceiling a VALU-bound kernel is priced against.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/_build/valu_ceiling tools/valu_ceiling.hip
//   run:   tools/_build/valu_ceiling            (prints one JSON line per configuration)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// ILP independent chains of v_pk_fma_f32 (two fp32 FMAs per lane per instruction) per lane
template <int ILP>
__global__ __launch_bounds__(256) void k_stream_pk(float* out, int iters, float c1, float c2)
{
	v2f a[ILP];
#pragma unroll
	for (int k = 0; k < ILP; k++) a[k] = v2f{ (float)(threadIdx.x + k) * 1e-3f, (float)(threadIdx.x + 2 * k) * 1e-3f };
	const v2f m = { c1, c1 * 0.999f }, b = { c2, c2 * 2.f };
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 64; r++)
#pragma unroll
			for (int k = 0; k < ILP; k++) a[k] = __builtin_elementwise_fma(a[k], m, b);
	}
	float s = 0.f;
#pragma unroll
	for (int k = 0; k < ILP; k++) s += a[k].x + a[k].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ILP independent v_fma_f32 chains per lane; 64 FMAs per chain per outer iteration.
template <int ILP, int KIND>
__global__ __launch_bounds__(256) void k_stream(float* out, int iters, float c1, float c2)
{
	float a[ILP];
#pragma unroll
	for (int k = 0; k < ILP; k++) a[k] = (float)(threadIdx.x + k) * 1e-3f;
	// per-lane multiplier / addend for KIND 3 (values the compiler cannot move to the scalar file)
	const float vb = c1 + (float)(threadIdx.x & 7) * 1e-9f, vc = c2 * (float)(1 + (threadIdx.x & 3));
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 64; r++)
#pragma unroll
			for (int k = 0; k < ILP; k++)
			{
				if (KIND == 0) a[k] = __builtin_fmaf(a[k], c1, c2);                 // v_fma_f32, one VGPR source (c1, c2 live in SGPRs)
				else if (KIND == 3) a[k] = __builtin_fmaf(a[k], vb, vc);             // v_fma_f32, three VGPR sources
				else if (KIND == 1) a[k] = __builtin_amdgcn_exp2f(a[k]) * c1;      // v_exp_f32 + v_mul_f32
				else a[k] = __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x & 63) << 2, __float_as_int(a[k]))) + c2;   // ds_bpermute_b32 + v_add_f32
			}
	}
	float s = 0.f;
#pragma unroll
	for (int k = 0; k < ILP; k++) s += a[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Issue cost of the instructions the register-resident sort is made of (inline asm so that the compiler cannot rewrite them):
// OP 0: v_cmp_lt_u64 + v_cndmask_b32   1: v_cmp_lt_u32 + v_cndmask_b32   2: v_mov_b32_dpp row_ror:8   3: v_permlane32_swap_b32
template <int OP>
__global__ __launch_bounds__(256) void k_stream_asm(float* out, int iters)
{
	uint32_t a0 = threadIdx.x, a1 = threadIdx.x * 3u, a2 = threadIdx.x * 5u, a3 = threadIdx.x * 7u;
	uint32_t b0 = 11u + threadIdx.x, b1 = 13u, b2 = 17u + threadIdx.x, b3 = 19u;
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int r = 0; r < 32; r++)
		{
			if (OP == 0)
				asm volatile("v_cmp_lt_u64 vcc, %[x], %[y]\n\tv_cndmask_b32 %[a], %[a], %[c], vcc\n\t"
				             "v_cmp_lt_u64 vcc, %[y], %[x]\n\tv_cndmask_b32 %[c], %[c], %[a], vcc"
				             : [a] "+v"(a0), [c] "+v"(a2) : [x] "v"((uint64_t)a1 << 32 | b0), [y] "v"((uint64_t)a3 << 32 | b2) : "vcc");
			else if (OP == 1)
				asm volatile("v_cmp_lt_u32 vcc, %[x], %[y]\n\tv_cndmask_b32 %[a], %[a], %[c], vcc\n\t"
				             "v_cmp_lt_u32 vcc, %[y], %[x]\n\tv_cndmask_b32 %[c], %[c], %[a], vcc"
				             : [a] "+v"(a0), [c] "+v"(a2) : [x] "v"(a1), [y] "v"(a3) : "vcc");
			else if (OP == 2)
				asm volatile("v_mov_b32_dpp %[a], %[b] row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %[c], %[d] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
				             "v_mov_b32_dpp %[b], %[a] row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %[d], %[c] row_ror:8 row_mask:0xf bank_mask:0xf"
				             : [a] "+v"(a0), [b] "+v"(a1), [c] "+v"(a2), [d] "+v"(a3));
			else
				asm volatile("v_permlane32_swap_b32 %[a], %[b]\n\tv_permlane32_swap_b32 %[c], %[d]\n\t"
				             "v_permlane32_swap_b32 %[a], %[c]\n\tv_permlane32_swap_b32 %[b], %[d]"
				             : [a] "+v"(a0), [b] "+v"(a1), [c] "+v"(a2), [d] "+v"(a3));
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3);
}
template <int OP>
static void run_asm(const char* name, int waves_per_simd, float* d_out)
{
	const int iters = 500;
	dim3 grid(256 * waves_per_simd), block(256);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_stream_asm<OP>), grid, block, 0, 0, d_out, iters);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 5; rep++)
	{
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k_stream_asm<OP>), grid, block, 0, 0, d_out, iters);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double wave_insts = (double)grid.x * 4.0 * iters * 32.0 * 4.0;
	const double per_simd_per_ns = wave_insts / 1024.0 / (best * 1e6);
	printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_ns_per_simd\": %.4f, "
	       "\"cycles_per_inst_at_2.4GHz\": %.3f, \"chip_wave_insts_per_s\": %.4e}\n",
	       name, waves_per_simd, best, per_simd_per_ns, 2.4 / per_simd_per_ns, per_simd_per_ns * 1e9 * 1024.0);
}

template <int ILP, int KIND>
static void run(const char* name, int waves_per_simd, float* d_out, int insts_per_op)
{
	const int iters = 2000 / ILP > 0 ? 2000 / ILP : 1;
	const int cus = 256;
	dim3 grid(cus * waves_per_simd), block(256);       // a 256-thread workgroup = one wave on each of the 4 SIMDs of a CU
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_stream<ILP, KIND>), grid, block, 0, 0, d_out, iters, 1.0001f, 1e-7f);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 5; rep++)
	{
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k_stream<ILP, KIND>), grid, block, 0, 0, d_out, iters, 1.0001f, 1e-7f);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double wave_insts = (double)grid.x * 4.0 * iters * 64.0 * ILP * insts_per_op;
	const double per_simd_per_ns = wave_insts / 1024.0 / (best * 1e6);
	printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ilp\": %d, \"ms\": %.4f, \"wave_insts_per_ns_per_simd\": %.4f, "
	       "\"cycles_per_inst_at_2.4GHz\": %.3f, \"chip_wave_insts_per_s\": %.4e}\n",
	       name, waves_per_simd, ILP, best, per_simd_per_ns, 2.4 / per_simd_per_ns, per_simd_per_ns * 1e9 * 1024.0);
}

template <int ILP>
static void run_pk(const char* name, int waves_per_simd, float* d_out)
{
	const int iters = 2000 / ILP > 0 ? 2000 / ILP : 1;
	dim3 grid(256 * waves_per_simd), block(256);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_stream_pk<ILP>), grid, block, 0, 0, d_out, iters, 1.0001f, 1e-7f);
	CHECK(hipDeviceSynchronize());
	float best = 1e30f;
	for (int rep = 0; rep < 5; rep++)
	{
		CHECK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((k_stream_pk<ILP>), grid, block, 0, 0, d_out, iters, 1.0001f, 1e-7f);
		CHECK(hipEventRecord(e1, 0));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (ms < best) best = ms;
	}
	const double wave_insts = (double)grid.x * 4.0 * iters * 64.0 * ILP;
	const double per_simd_per_ns = wave_insts / 1024.0 / (best * 1e6);
	printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ilp\": %d, \"ms\": %.4f, \"wave_insts_per_ns_per_simd\": %.4f, "
	       "\"cycles_per_inst_at_2.4GHz\": %.3f, \"chip_wave_insts_per_s\": %.4e}\n",
	       name, waves_per_simd, ILP, best, per_simd_per_ns, 2.4 / per_simd_per_ns, per_simd_per_ns * 1e9 * 1024.0);
}

int main(int argc, char** argv)
{
	float* d_out;
	CHECK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * sizeof(float)));
	if (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 'q')
	{
		// --quick (bench.py): the ceiling (independent FMAs) and the floor (one dependent chain) at the scorer's occupancy
		run<8, 0>("v_fma_f32", 5, d_out, 1);
		run<1, 0>("v_fma_f32 dependent chain", 5, d_out, 1);
		run<8, 3>("v_fma_f32 with three VGPR sources", 5, d_out, 1);
		CHECK(hipFree(d_out));
		return 0;
	}
	if (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 's')
	{
		// --sort: the instruction mix of the register-resident key sort
		for (int w : { 1, 4 })
		{
			run_asm<0>("v_cmp_lt_u64 + v_cndmask_b32 (two instructions counted)", w, d_out);
			run_asm<1>("v_cmp_lt_u32 + v_cndmask_b32 (two instructions counted)", w, d_out);
			run_asm<2>("v_mov_b32_dpp row_ror:8", w, d_out);
			run_asm<3>("v_permlane32_swap_b32", w, d_out);
		}
		CHECK(hipFree(d_out));
		return 0;
	}
	const int wl[] = { 1, 2, 4, 5, 8 };
	for (int w : wl) run<1, 0>("v_fma_f32 dependent chain", w, d_out, 1);
	for (int w : wl) run<4, 0>("v_fma_f32", w, d_out, 1);
	for (int w : wl) run<8, 0>("v_fma_f32", w, d_out, 1);
	for (int w : wl) run<8, 3>("v_fma_f32 with three VGPR sources", w, d_out, 1);
	for (int w : wl) run<1, 3>("v_fma_f32 with three VGPR sources, dependent chain", w, d_out, 1);
	for (int w : wl) run_pk<1>("v_pk_fma_f32 dependent chain", w, d_out);
	for (int w : wl) run_pk<4>("v_pk_fma_f32", w, d_out);
	for (int w : wl) run<4, 1>("v_exp_f32+v_mul_f32", w, d_out, 2);
	for (int w : wl) run<4, 2>("ds_bpermute_b32+v_add_f32", w, d_out, 2);
	CHECK(hipFree(d_out));
	return 0;
}

// fr_math.h -- per-Gaussian arithmetic of the rasteriser, shared by every HIP kernel.
//
// Written host/device-neutral (FR_HD) so that the same functions can be compiled with g++ into a CPU
// test harness (tests/harness) and compared against the oracle before any GPU time is spent.  The
// forward half (projection, cov2D, conic, radius, tile rectangle, exp, SH->RGB) is written so that it
// rounds exactly like oracle/fisher_oracle.c: same operand order, no FMA contraction (the translation
// units that include this header are built with -ffp-contract=off), explicit fmaf only inside fr_expf.
// The backward half only has to agree to rounding error and is free to contract.
//
// Reference arithmetic restated here (paths relative to the reference tree,
// RAST = thirdparty/diff-gaussian-rasterization-modified/cuda_rasterizer):
//   RAST/auxiliary.h:41-97,139-164   RAST/forward.cu:20-152,181-255
//   RAST/backward.cu:20-139,276-408,412-475,532-583
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define FR_HD __host__ __device__ __forceinline__
#else
#define FR_HD static inline
#endif

#define FR_BLOCK_X 16
#define FR_BLOCK_Y 16

struct fr_f2 { float x, y; };
struct fr_f3 { float x, y, z; };
struct fr_f4 { float x, y, z, w; };

FR_HD uint32_t fr_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
FR_HD float fr_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// float -> int32 with PTX cvt.rzi.s32.f32 semantics (saturate, NaN -> 0)
FR_HD int fr_f2i(float f)
{
	if (!(f == f)) return 0;
	if (f >= 2147483648.0f) return 2147483647;
	if (f <= -2147483648.0f) return (-2147483647 - 1);
	return (int)f;
}

// e^x as a fixed sequence of IEEE-754 binary32 operations (Cody-Waite reduction, degree-5 minimax
// polynomial evaluated with fmaf, scaling by two exact powers of two).  <= 1 ulp.  The oracle uses the
// same sequence, which is what makes forward parity bit-exact.
// Branch-free body of fr_expf: exact same operation sequence, valid for x in [-103.97, 88.72] and NaN (NaN in, NaN out);
// anything else gives garbage without trapping, so kernels may evaluate it on lanes whose result they then discard.
FR_HD float fr_expf_inrange(float x)
{
	float kf = rintf(x * 1.44269504088896341f);
	float r = fmaf(kf, -0.693359375f, x);
	r = fmaf(kf, 2.12194440e-4f, r);
	float p = 1.9875691500e-4f;
	p = fmaf(p, r, 1.3981999507e-3f);
	p = fmaf(p, r, 8.3334519073e-3f);
	p = fmaf(p, r, 4.1665795894e-2f);
	p = fmaf(p, r, 1.6666665459e-1f);
	p = fmaf(p, r, 5.0000001201e-1f);
	float r2 = r * r;
	float y = fmaf(p, r2, r) + 1.0f;
	int k = (int)kf;
	int k1 = k >> 1;
	int k2 = k - k1;
	return (y * fr_as_f32((uint32_t)(k1 + 127) << 23)) * fr_as_f32((uint32_t)(k2 + 127) << 23);
}

FR_HD float fr_expf(float x)
{
	if (!(x == x)) return x;
	if (x > 88.72283905206835f) return INFINITY;
	if (x < -103.97208f) return 0.0f;
	return fr_expf_inrange(x);
}

// ---- auxiliary.h ------------------------------------------------------------------------------
FR_HD float fr_ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

FR_HD uint32_t fr_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
FR_HD int fr_imax(int a, int b) { return a > b ? a : b; }

struct fr_rect { uint32_t x0, y0, x1, y1; };

FR_HD fr_rect fr_get_rect(float px, float py, int max_radius, uint32_t gx, uint32_t gy)
{
	fr_rect r;
	const float rad = (float)max_radius;
	r.x0 = fr_umin(gx, (uint32_t)fr_imax(0, fr_f2i((px - rad) / (float)FR_BLOCK_X)));
	r.y0 = fr_umin(gy, (uint32_t)fr_imax(0, fr_f2i((py - rad) / (float)FR_BLOCK_Y)));
	r.x1 = fr_umin(gx, (uint32_t)fr_imax(0, fr_f2i((((px + rad) + (float)FR_BLOCK_X) - 1.0f) / (float)FR_BLOCK_X)));
	r.y1 = fr_umin(gy, (uint32_t)fr_imax(0, fr_f2i((((py + rad) + (float)FR_BLOCK_Y) - 1.0f) / (float)FR_BLOCK_Y)));
	return r;
}

// m = 16 floats, column-major
FR_HD fr_f3 fr_xform4x3(fr_f3 p, const float* m)
{
	fr_f3 t;
	t.x = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12];
	t.y = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13];
	t.z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14];
	return t;
}
FR_HD fr_f4 fr_xform4x4(fr_f3 p, const float* m)
{
	fr_f4 t;
	t.x = ((m[0] * p.x + m[4] * p.y) + m[8] * p.z) + m[12];
	t.y = ((m[1] * p.x + m[5] * p.y) + m[9] * p.z) + m[13];
	t.z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14];
	t.w = ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15];
	return t;
}

// World -> candidate camera frame with a ROW-major 4x4 (rel_w2c of gaussian.py:1523-1527).
FR_HD fr_f3 fr_world_to_cam(fr_f3 p, const float* w)
{
	fr_f3 t;
	t.x = ((w[0] * p.x + w[1] * p.y) + w[2] * p.z) + w[3];
	t.y = ((w[4] * p.x + w[5] * p.y) + w[6] * p.z) + w[7];
	t.z = ((w[8] * p.x + w[9] * p.y) + w[10] * p.z) + w[11];
	return t;
}

// ---- forward.cu:118-152  cov3D = R S^2 R^T from an (un-normalised) quaternion ------------------------
// Rc[c][r] is the GLM matrix R (column c, row r) = transpose of the usual rotation matrix.
FR_HD void fr_quat_glmR(fr_f4 q, float Rc[3][3])
{
	const float r = q.x, x = q.y, y = q.z, z = q.w;
	Rc[0][0] = 1.f - 2.f * (y * y + z * z); Rc[0][1] = 2.f * (x * y - r * z); Rc[0][2] = 2.f * (x * z + r * y);
	Rc[1][0] = 2.f * (x * y + r * z); Rc[1][1] = 1.f - 2.f * (x * x + z * z); Rc[1][2] = 2.f * (y * z - r * x);
	Rc[2][0] = 2.f * (x * z - r * y); Rc[2][1] = 2.f * (y * z + r * x); Rc[2][2] = 1.f - 2.f * (x * x + y * y);
}

FR_HD void fr_cov3d(fr_f3 scale, float mod, fr_f4 rot, float* cov3D)
{
	float Rc[3][3];
	fr_quat_glmR(rot, Rc);
	const float s[3] = { mod * scale.x, mod * scale.y, mod * scale.z };
	// M = S * R (GLM): M[c][r] = s_r * R[c][r]   (the two zero products of the general formula add +-0)
	float Mc[3][3];
	for (int c = 0; c < 3; c++)
		for (int r = 0; r < 3; r++)
			Mc[c][r] = s[r] * Rc[c][r];
	// Sigma = transpose(M) * M : Sigma[c][r] = M[r][0]*M[c][0] + M[r][1]*M[c][1] + M[r][2]*M[c][2]
#define FR_SIG(c, r) ((Mc[r][0] * Mc[c][0] + Mc[r][1] * Mc[c][1]) + Mc[r][2] * Mc[c][2])
	cov3D[0] = FR_SIG(0, 0);
	cov3D[1] = FR_SIG(0, 1);
	cov3D[2] = FR_SIG(0, 2);
	cov3D[3] = FR_SIG(1, 1);
	cov3D[4] = FR_SIG(1, 2);
	cov3D[5] = FR_SIG(2, 2);
#undef FR_SIG
}

// FR_CONTRACT (first statement of a function body or block): the expressions below it may be fused into multiply-adds.  Used on
// the scorer's records only (the FAST instantiation of fr_cov2d_setup, fr_scorer_poly_g), whose bar is 1e-4 on the scores: never by
// the forward projection, whose radii / rectangles / depths are compared bit for bit, and not by the gradient chains either --
// fused, the rotation gradient of an isotropic Gaussian stops cancelling to the exact 0 the reference's arithmetic gives.
#if defined(__clang__)
#define FR_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define FR_CONTRACT
#endif

// Division policy of the Jacobian helpers: IEEE `/` by default (the single-view rasteriser and the host harness); the
// Fisher scorer, whose bar is 1e-4 on the scores, instantiates them with FAST = true: one v_rcp_f32 (1 ulp) + multiply.
template <bool FAST> FR_HD float fr_divt(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if (FAST) return a * __builtin_amdgcn_rcpf(b);
#endif
	return a / b;
}

// ---- forward.cu:74-113 / backward.cu:300-333: shared front half of computeCov2D ----------------------
struct fr_cov2d {
	float tx, ty, tz;       // camera-space mean after the fov clamp
	float txtz, tytz;
	float T0[3], T1[3];     // GLM T[0][.], T[1][.]  ( = rows 0,1 of J_math * W_math )
	float Wc[3][3];         // GLM W[c][r] = view[c + 4 r]
	float c3[6];
	float cov00, cov01, cov11; // before the +0.3 low-pass
};

// (one body, two instantiations: FAST = false is the forward projection's -- exact, unfused --, FAST = true the scorer's)
#define FR_COV2D_SETUP_BODY \
	fr_f3 t = fr_xform4x3(mean, view);\
	const float limx = 1.3f * tan_fovx;\
	const float limy = 1.3f * tan_fovy;\
	c.txtz = fr_divt<FAST>(t.x, t.z);\
	c.tytz = fr_divt<FAST>(t.y, t.z);\
	t.x = fminf(limx, fmaxf(-limx, c.txtz)) * t.z;\
	t.y = fminf(limy, fmaxf(-limy, c.tytz)) * t.z;\
	c.tx = t.x; c.ty = t.y; c.tz = t.z;\
	const float J00 = fr_divt<FAST>(focal_x, t.z);\
	const float J02 = fr_divt<FAST>(-(focal_x * t.x), (t.z * t.z));\
	const float J11 = fr_divt<FAST>(focal_y, t.z);\
	const float J12 = fr_divt<FAST>(-(focal_y * t.y), (t.z * t.z));\
	for (int cc = 0; cc < 3; cc++)\
		for (int r = 0; r < 3; r++)\
			c.Wc[cc][r] = view[cc + 4 * r];\
	for (int r = 0; r < 3; r++)\
	{\
		c.T0[r] = c.Wc[0][r] * J00 + c.Wc[2][r] * J02;\
		c.T1[r] = c.Wc[1][r] * J11 + c.Wc[2][r] * J12;\
	}\
	for (int i = 0; i < 6; i++) c.c3[i] = cov3D[i];\
	const float c0 = cov3D[0], c1 = cov3D[1], c2 = cov3D[2], c3 = cov3D[3], c4 = cov3D[4], c5 = cov3D[5];\
	const float A00 = (c.T0[0] * c0 + c.T0[1] * c1) + c.T0[2] * c2;\
	const float A10 = (c.T0[0] * c1 + c.T0[1] * c3) + c.T0[2] * c4;\
	const float A20 = (c.T0[0] * c2 + c.T0[1] * c4) + c.T0[2] * c5;\
	const float A01 = (c.T1[0] * c0 + c.T1[1] * c1) + c.T1[2] * c2;\
	const float A11 = (c.T1[0] * c1 + c.T1[1] * c3) + c.T1[2] * c4;\
	const float A21 = (c.T1[0] * c2 + c.T1[1] * c4) + c.T1[2] * c5;\
	c.cov00 = (A00 * c.T0[0] + A10 * c.T0[1]) + A20 * c.T0[2];\
	c.cov01 = (A01 * c.T0[0] + A11 * c.T0[1]) + A21 * c.T0[2];\
	c.cov11 = (A01 * c.T1[0] + A11 * c.T1[1]) + A21 * c.T1[2];
template <bool FAST = false>
FR_HD void fr_cov2d_setup(fr_f3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                          const float* cov3D, const float* view, fr_cov2d& c)
{
	if constexpr (FAST) { FR_CONTRACT FR_COV2D_SETUP_BODY }
	else { FR_COV2D_SETUP_BODY }
}
#undef FR_COV2D_SETUP_BODY

// ---- forward.cu:181-255: everything preprocessCUDA derives for one Gaussian -------------------------
struct fr_splat {
	int radius;          // 0 => not rendered
	float depth;
	float px, py;
	float conx, cony, conz;
	fr_rect rect;
	uint32_t tiles;
};

// p_orig is the point handed to the rasteriser (already in the candidate frame for the Fisher path).
FR_HD fr_splat fr_preprocess_one(fr_f3 p_orig, const float* cov3D, const float* view, const float* proj,
                                 int W, int H, float tan_fovx, float tan_fovy, float focal_x, float focal_y,
                                 uint32_t gx, uint32_t gy)
{
	fr_splat s;
	s.radius = 0; s.tiles = 0; s.depth = 0.f; s.px = s.py = 0.f; s.conx = s.cony = s.conz = 0.f;
	s.rect.x0 = s.rect.y0 = s.rect.x1 = s.rect.y1 = 0;

	fr_f3 p_view = fr_xform4x3(p_orig, view);
	if (p_view.z <= 0.001f)
		return s;
	fr_f4 p_hom = fr_xform4x4(p_orig, proj);
	float p_w = 1.0f / (p_hom.w + 0.0000001f);
	float projx = p_hom.x * p_w, projy = p_hom.y * p_w;

	fr_cov2d c;
	fr_cov2d_setup(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, c);
	const float covx = c.cov00 + 0.3f, covy = c.cov01, covz = c.cov11 + 0.3f;

	float det = (covx * covz - covy * covy);
	if (det == 0.0f)
		return s;
	float det_inv = 1.f / det;
	float conx = covz * det_inv, cony = -covy * det_inv, conz = covx * det_inv;

	float mid = 0.5f * (covx + covz);
	float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
	float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
	float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
	float px = fr_ndc2pix(projx, W), py = fr_ndc2pix(projy, H);
	int irad = fr_f2i(my_radius);
	fr_rect rc = fr_get_rect(px, py, irad, gx, gy);
	uint32_t tiles = (rc.x1 - rc.x0) * (rc.y1 - rc.y0);
	if (tiles == 0)
		return s;
	s.radius = irad; s.depth = p_view.z; s.px = px; s.py = py;
	s.conx = conx; s.cony = cony; s.conz = conz; s.rect = rc;
	s.tiles = (rc.y1 - rc.y0) * (rc.x1 - rc.x0);
	return s;
}

// ---- spherical harmonics, forward.cu:20-71 ---------------------------------------------------------
#define FR_SH_C0 0.28209479177387814f
#define FR_SH_C1 0.4886025119029199f
#define FR_SH_C2_0 1.0925484305920792f
#define FR_SH_C2_1 -1.0925484305920792f
#define FR_SH_C2_2 0.31539156525252005f
#define FR_SH_C2_3 -1.0925484305920792f
#define FR_SH_C2_4 0.5462742152960396f
#define FR_SH_C3_0 -0.5900435899266435f
#define FR_SH_C3_1 2.890611442640554f
#define FR_SH_C3_2 -0.4570457994644658f
#define FR_SH_C3_3 0.3731763325901154f
#define FR_SH_C3_4 -0.4570457994644658f
#define FR_SH_C3_5 1.445305721320277f
#define FR_SH_C3_6 -0.5900435899266435f

// sh points at this Gaussian's coefficients, sh[3*k + channel]; returns clamped colour, writes clamp flags
FR_HD fr_f3 fr_sh_to_rgb(int deg, fr_f3 pos, fr_f3 campos, const float* sh, uint8_t* clamped3)
{
	fr_f3 dir = { pos.x - campos.x, pos.y - campos.y, pos.z - campos.z };
	const float dx2 = dir.x * dir.x, dy2 = dir.y * dir.y, dz2 = dir.z * dir.z;
	const float len = sqrtf((dx2 + dy2) + dz2);
	dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
	const float x = dir.x, y = dir.y, z = dir.z;
	float res[3];
	for (int c = 0; c < 3; c++)
	{
#define FR_S(k) sh[3 * (k) + c]
		float result = FR_SH_C0 * FR_S(0);
		if (deg > 0)
		{
			result = result - FR_SH_C1 * y * FR_S(1) + FR_SH_C1 * z * FR_S(2) - FR_SH_C1 * x * FR_S(3);
			if (deg > 1)
			{
				float xx = x * x, yy = y * y, zz = z * z;
				float xy = x * y, yz = y * z, xz = x * z;
				result = result +
					FR_SH_C2_0 * xy * FR_S(4) +
					FR_SH_C2_1 * yz * FR_S(5) +
					FR_SH_C2_2 * (2.0f * zz - xx - yy) * FR_S(6) +
					FR_SH_C2_3 * xz * FR_S(7) +
					FR_SH_C2_4 * (xx - yy) * FR_S(8);
				if (deg > 2)
				{
					result = result +
						FR_SH_C3_0 * y * (3.0f * xx - yy) * FR_S(9) +
						FR_SH_C3_1 * xy * z * FR_S(10) +
						FR_SH_C3_2 * y * (4.0f * zz - xx - yy) * FR_S(11) +
						FR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * FR_S(12) +
						FR_SH_C3_4 * x * (4.0f * zz - xx - yy) * FR_S(13) +
						FR_SH_C3_5 * z * (xx - yy) * FR_S(14) +
						FR_SH_C3_6 * x * (xx - 3.0f * yy) * FR_S(15);
				}
			}
		}
#undef FR_S
		result += 0.5f;
		res[c] = result;
	}
	clamped3[0] = (res[0] < 0); clamped3[1] = (res[1] < 0); clamped3[2] = (res[2] < 0);
	fr_f3 out = { fmaxf(res[0], 0.0f), fmaxf(res[1], 0.0f), fmaxf(res[2], 0.0f) };
	return out;
}

// =====================================================================================================
// Backward: per-Gaussian Jacobians.
//
// Every leaf gradient that renderCUDAFused accumulates is LINEAR in the per-(pixel,Gaussian) vector
//   u = (dL_dmean2D.x, dL_dmean2D.y, dL_dconic.x, dL_dconic.y, dL_dconic.w)   and   g = dL_dcolor[3]
// with coefficients that depend on the Gaussian only (backward.cu:276-408, 412-475, 532-583).  The
// kernels therefore build those coefficient matrices once per tile instance (by pushing unit vectors
// through the literal chain below) and apply them per pixel; `powf(., grad_power)` is then taken of the
// very same per-pixel leaf values as in the reference.
// =====================================================================================================

// backward.cu:347-407, second half of computeCov2DCUDARelocated: (dL_da, dL_db, dL_dc) -- the gradient w.r.t. the three
// entries of cov2D -- -> dL_dcov3D[6] and the covariance part of dL_dmean.  `nonzero`: the reference's `denom2inv != 0` branch.
#define FR_ABC_BACKWARD_BODY \
	const float limx = 1.3f * tan_fovx; \
	const float limy = 1.3f * tan_fovy; \
	const float x_grad_mul = (c.txtz < -limx || c.txtz > limx) ? 0.f : 1.f; \
	const float y_grad_mul = (c.tytz < -limy || c.tytz > limy) ? 0.f : 1.f; \
	const float* T0 = c.T0; const float* T1 = c.T1; \
	if (nonzero) \
	{ \
		dcov[0] = (T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc); \
		dcov[3] = (T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc); \
		dcov[5] = (T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc); \
		dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc; \
		dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc; \
		dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc; \
	} \
	else \
	{ \
		for (int i = 0; i < 6; i++) dcov[i] = 0; \
	} \
	const float c0 = c.c3[0], c1 = c.c3[1], c2 = c.c3[2], c3 = c.c3[3], c4 = c.c3[4], c5 = c.c3[5]; \
	const float t0v0 = T0[0] * c0 + T0[1] * c1 + T0[2] * c2; \
	const float t0v1 = T0[0] * c1 + T0[1] * c3 + T0[2] * c4; \
	const float t0v2 = T0[0] * c2 + T0[1] * c4 + T0[2] * c5; \
	const float t1v0 = T1[0] * c0 + T1[1] * c1 + T1[2] * c2; \
	const float t1v1 = T1[0] * c1 + T1[1] * c3 + T1[2] * c4; \
	const float t1v2 = T1[0] * c2 + T1[1] * c4 + T1[2] * c5; \
	const float dL_dT00 = 2 * t0v0 * dL_da + t1v0 * dL_db; \
	const float dL_dT01 = 2 * t0v1 * dL_da + t1v1 * dL_db; \
	const float dL_dT02 = 2 * t0v2 * dL_da + t1v2 * dL_db; \
	const float dL_dT10 = 2 * t1v0 * dL_dc + t0v0 * dL_db; \
	const float dL_dT11 = 2 * t1v1 * dL_dc + t0v1 * dL_db; \
	const float dL_dT12 = 2 * t1v2 * dL_dc + t0v2 * dL_db; \
	const float dL_dJ00 = c.Wc[0][0] * dL_dT00 + c.Wc[0][1] * dL_dT01 + c.Wc[0][2] * dL_dT02; \
	const float dL_dJ02 = c.Wc[2][0] * dL_dT00 + c.Wc[2][1] * dL_dT01 + c.Wc[2][2] * dL_dT02; \
	const float dL_dJ11 = c.Wc[1][0] * dL_dT10 + c.Wc[1][1] * dL_dT11 + c.Wc[1][2] * dL_dT12; \
	const float dL_dJ12 = c.Wc[2][0] * dL_dT10 + c.Wc[2][1] * dL_dT11 + c.Wc[2][2] * dL_dT12; \
	const float tz = fr_divt<FAST>(1.f, c.tz); \
	const float tz2 = tz * tz; \
	const float tz3 = tz2 * tz; \
	const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02; \
	const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12; \
	const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * c.tx) * tz3 * dL_dJ02 + (2 * h_y * c.ty) * tz3 * dL_dJ12; \
	dmean.x = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz; \
	dmean.y = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz; \
	dmean.z = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
template <bool FAST = false>
FR_HD void fr_cov2d_abc_backward(const fr_cov2d& c, float h_x, float h_y, float tan_fovx, float tan_fovy,
                                 const float* view, float dL_da, float dL_db, float dL_dc, bool nonzero, fr_f3& dmean, float* dcov)
{
	if constexpr (FAST) { FR_CONTRACT FR_ABC_BACKWARD_BODY }
	else { FR_ABC_BACKWARD_BODY }
}
#undef FR_ABC_BACKWARD_BODY

// backward.cu:335-407: (dL_dconic.x,.y,.w) -> dL_dcov3D[6] and the covariance part of dL_dmean
template <bool FAST = false>
FR_HD void fr_cov2d_backward(const fr_cov2d& c, float h_x, float h_y, float tan_fovx, float tan_fovy,
                             const float* view, float dcx, float dcy, float dcw, fr_f3& dmean, float* dcov)
{
	const float a = c.cov00 + 0.3f, b = c.cov01, cc = c.cov11 + 0.3f;
	const float denom = a * cc - b * b;
	float dL_da = 0, dL_db = 0, dL_dc = 0;
	const float denom2inv = fr_divt<FAST>(1.0f, (denom * denom) + 0.0000001f);
	if (denom2inv != 0)
	{
		dL_da = denom2inv * (-cc * cc * dcx + 2 * b * cc * dcy + (denom - a * cc) * dcw);
		dL_dc = denom2inv * (-a * a * dcw + 2 * a * b * dcy + (denom - a * cc) * dcx);
		dL_db = denom2inv * 2 * (b * cc * dcx - (denom + 2 * b * b) * dcy + a * b * dcw);
	}
	fr_cov2d_abc_backward<FAST>(c, h_x, h_y, tan_fovx, tan_fovy, view, dL_da, dL_db, dL_dc, denom2inv != 0, dmean, dcov);
}

// backward.cu:557-574: dL_dmean3D += Mp * (dL_dmean2D.x, dL_dmean2D.y); returns Mp as 3 rows of 2
#define FR_PROJ_JACOBIAN_BODY \
	fr_f4 m_hom = fr_xform4x4(m, proj); \
	float m_w = fr_divt<FAST>(1.0f, m_hom.w + 0.0000001f); \
	float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w; \
	float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w; \
	Mp[0][0] = (proj[0] * m_w - proj[3] * mul1); Mp[0][1] = (proj[1] * m_w - proj[3] * mul2); \
	Mp[1][0] = (proj[4] * m_w - proj[7] * mul1); Mp[1][1] = (proj[5] * m_w - proj[7] * mul2); \
	Mp[2][0] = (proj[8] * m_w - proj[11] * mul1); Mp[2][1] = (proj[9] * m_w - proj[11] * mul2);
template <bool FAST = false>
FR_HD void fr_proj_jacobian(fr_f3 m, const float* proj, float Mp[3][2])
{
	if constexpr (FAST) { FR_CONTRACT FR_PROJ_JACOBIAN_BODY }
	else { FR_PROJ_JACOBIAN_BODY }
}
#undef FR_PROJ_JACOBIAN_BODY

// backward.cu:412-475: dL_dcov3D[6] -> dL_dscale, dL_drot
FR_HD void fr_cov3d_backward(fr_f3 scale, float mod, fr_f4 rot, const float* dcov, fr_f3& dscale, fr_f4& drot)
{
	const float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
	float Rc[3][3];
	fr_quat_glmR(rot, Rc);
	const float s[3] = { mod * scale.x, mod * scale.y, mod * scale.z };
	float Mc[3][3];   // M[c][r] = s_r R[c][r]
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++)
			Mc[c][rr] = s[rr] * Rc[c][rr];
	// dL_dSigma (symmetric), column-major
	const float dS[3][3] = {
		{ dcov[0], 0.5f * dcov[1], 0.5f * dcov[2] },
		{ 0.5f * dcov[1], dcov[3], 0.5f * dcov[4] },
		{ 0.5f * dcov[2], 0.5f * dcov[4], dcov[5] } };
	// dL_dM = (2 M) * dL_dSigma : dM[c][r] = sum_k 2 M[k][r] * dS[c][k]
	float dM[3][3];
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++)
			dM[c][rr] = 2.0f * Mc[0][rr] * dS[c][0] + 2.0f * Mc[1][rr] * dS[c][1] + 2.0f * Mc[2][rr] * dS[c][2];
	// Rt[i][j] = R[j][i], dMt[i][j] = dM[j][i]
	float dMt[3][3];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++)
			dMt[i][j] = dM[j][i];
	dscale.x = Rc[0][0] * dMt[0][0] + Rc[1][0] * dMt[0][1] + Rc[2][0] * dMt[0][2];
	dscale.y = Rc[0][1] * dMt[1][0] + Rc[1][1] * dMt[1][1] + Rc[2][1] * dMt[1][2];
	dscale.z = Rc[0][2] * dMt[2][0] + Rc[1][2] * dMt[2][1] + Rc[2][2] * dMt[2][2];
	for (int j = 0; j < 3; j++) { dMt[0][j] *= s[0]; dMt[1][j] *= s[1]; dMt[2][j] *= s[2]; }
	drot.x = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
	drot.y = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
	drot.z = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
	drot.w = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
}

// Jacobian of the camera-frame mean gradient: dL_dmean3D = A * (m2x, m2y, cx, cy, cw)
// A is 3 rows x 5 columns.  Optionally also B = d(dL_dcov3D)/d(cx,cy,cw) (6x3).
template <bool FAST = false>
FR_HD void fr_mean_jacobian(fr_f3 mean, const float* cov3D, const float* view, const float* proj,
                            float focal_x, float focal_y, float tan_fovx, float tan_fovy,
                            float A[3][5], float (*B)[3], float* cov2d_out = nullptr)
{
	fr_cov2d c;
	fr_cov2d_setup<FAST>(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, c);
	if (cov2d_out) { cov2d_out[0] = c.cov00; cov2d_out[1] = c.cov01; cov2d_out[2] = c.cov11; }   // before the +0.3 low-pass
	float Mp[3][2];
	fr_proj_jacobian<FAST>(mean, proj, Mp);
	for (int k = 0; k < 3; k++) { A[k][0] = Mp[k][0]; A[k][1] = Mp[k][1]; }
#if defined(__HIPCC__)
#pragma unroll
#endif
	for (int j = 0; j < 3; j++)
	{
		fr_f3 dm; float dcov[6];
		fr_cov2d_backward<FAST>(c, focal_x, focal_y, tan_fovx, tan_fovy, view,
		                  j == 0 ? 1.f : 0.f, j == 1 ? 1.f : 0.f, j == 2 ? 1.f : 0.f, dm, dcov);
		A[0][2 + j] = dm.x; A[1][2 + j] = dm.y; A[2][2 + j] = dm.z;
		if (B)
			for (int i = 0; i < 6; i++) B[i][j] = dcov[i];
	}
}

// Jacobian of (dL_dscale[3], dL_drot[4]) w.r.t. (cx, cy, cw), given B = d(dL_dcov3D)/d(cx,cy,cw)
FR_HD void fr_scale_rot_jacobian(fr_f3 scale, float mod, fr_f4 rot, const float B[6][3], float Cm[7][3])
{
#if defined(__HIPCC__)
#pragma unroll
#endif
	for (int j = 0; j < 3; j++)
	{
		float dcov[6];
		for (int i = 0; i < 6; i++) dcov[i] = B[i][j];
		fr_f3 ds; fr_f4 dr;
		fr_cov3d_backward(scale, mod, rot, dcov, ds, dr);
		Cm[0][j] = ds.x; Cm[1][j] = ds.y; Cm[2][j] = ds.z;
		Cm[3][j] = dr.x; Cm[4][j] = dr.y; Cm[5][j] = dr.z; Cm[6][j] = dr.w;
	}
}

// =====================================================================================================
// The scorer's per-(view, Gaussian) records, in the basis g = conic * d  (d = mean2D - pixel).
//
// With w = opacity G dL_dalpha the reference's per-pair screen-space gradients are (backward.cu:1016-1031)
//   dL_dmean2D = -w (W/2 gx, H/2 gy),        dL_dconic(x, y, w) = -w/2 (dx^2, dx dy, dy^2),
// and its conic -> cov2D step (backward.cu:347-353) maps the latter to
//   (dL_da, dL_db, dL_dc) = w f (gx^2 / 2, gx gy, gy^2 / 2),   f = denom^2 / (denom^2 + 1e-7),
// exactly: -c^2 dx^2 + 2 b c dx dy - b^2 dy^2 = -(c dx - b dy)^2 = -(denom gx)^2, and likewise for the other two.  Every leaf
// is therefore w times a row R_c applied to gamma = (gx, gy, gx^2, gx gy, gy^2).  The kernels evaluate that form: written
// in (dx, dy) the same rows cancel like (cx dx + cy dy) does for a needle-shaped splat (condition number kappa =
// |conic| |d| / |g|, up to the squared aspect ratio), and the H_inv-weighted sum of their SQUARES expanded into a polynomial
// in (dx, dy) cancels like kappa^2 -- 2.4e-4 on the scores of the `border` test family, against 1e-6 for the binary32
// reference chain itself (oracle/ arbiter build).  In g the rows have no structural cancellation.
// The walk uses u = -g = (2 hcx dx + ncy dy, ncy dx + 2 hcz dy) with (hcx, ncy, hcz) = (-cx/2, -cy, -cz/2) of the record.
// =====================================================================================================

// Rg[r] = row of camera-frame mean component r over gamma(u) = (ux, uy, ux^2, ux uy, uy^2), u = -g;
// Bg (optional) = d(dL_dcov3D[6]) / d(dL_da, dL_db, dL_dc) for fr_scale_rot_jacobian; cov2d_out: cov2D before the +0.3.
#define FR_MEAN_ROWS_G_BODY \
	fr_cov2d c; \
	fr_cov2d_setup<FAST>(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, view, c); \
	if (cov2d_out) { cov2d_out[0] = c.cov00; cov2d_out[1] = c.cov01; cov2d_out[2] = c.cov11; } \
	const float a = c.cov00 + 0.3f, b = c.cov01, cc = c.cov11 + 0.3f; \
	const float denom = a * cc - b * b; \
	const float d2 = denom * denom; \
	const float f = d2 < 3.0e38f ? fr_divt<FAST>(d2, d2 + 0.0000001f) : 0.f; \
	if (f_out) *f_out = f; \
	float Mp[3][2]; \
	fr_proj_jacobian<FAST>(mean, proj, Mp); \
	const float hw = (float)(0.5 * W), hh = (float)(0.5 * H); \
	for (int k = 0; k < 3; k++) { Rg[k][0] = Mp[k][0] * hw; Rg[k][1] = Mp[k][1] * hh; } \
	const float wj[3] = { 0.5f * f, f, 0.5f * f }; \
	for (int j = 0; j < 3; j++) \
	{ \
		fr_f3 dm; float dcov[6]; \
		fr_cov2d_abc_backward<FAST>(c, focal_x, focal_y, tan_fovx, tan_fovy, view, \
		                            j == 0 ? 1.f : 0.f, j == 1 ? 1.f : 0.f, j == 2 ? 1.f : 0.f, true, dm, dcov); \
		Rg[0][2 + j] = dm.x * wj[j]; Rg[1][2 + j] = dm.y * wj[j]; Rg[2][2 + j] = dm.z * wj[j]; \
		if (Bg) \
			for (int i = 0; i < 6; i++) Bg[i][j] = dcov[i] * wj[j]; \
	}
template <bool FAST = false>
FR_HD void fr_mean_rows_g(fr_f3 mean, const float* cov3D, const float* view, const float* proj,
                          float focal_x, float focal_y, float tan_fovx, float tan_fovy, int W, int H,
                          float Rg[3][5], float (*Bg)[3], float* cov2d_out, float* f_out)
{
	if constexpr (FAST) { FR_CONTRACT FR_MEAN_ROWS_G_BODY }
	else { FR_MEAN_ROWS_G_BODY }
}
#undef FR_MEAN_ROWS_G_BODY

// The 12 coefficients of F(u) = sum_c hv[c] (R_c . gamma(u))^2 as a bivariate polynomial in (ux, uy) (terms of degree 2, 3
// and 4 only): Rg = the three mean rows, Cg = the seven scale / rotation rows over (ux^2, ux uy, uy^2) (C >= 11), hv = the
// weights [mean 3 | opacity | scale 3 | rot 4].  Layout (what the walk's Horner scheme reads):
//   q[0..11] = c02 c03 c04 c11 | c12 c13 c20 c21 | c22 c30 c31 c40      (cij multiplies ux^i uy^j)
template <int C>
FR_HD void fr_scorer_poly_g(const float Rg[3][5], const float (*Cg)[3], const float* hv, float q[12])
{
	FR_CONTRACT
	// upper triangle of Q = sum_c hv[c] R_c^T R_c, off-diagonal entries doubled: (i, j) at 0 1 2 3 4 / 5 6 7 8 / 9 10 11 / 12 13 / 14
	float qf[15];
	int n = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
	for (int i = 0; i < 5; i++)
#if defined(__HIPCC__)
#pragma unroll
#endif
		for (int j = i; j < 5; j++)
		{
			float acc = hv[0] * Rg[0][i] * Rg[0][j] + hv[1] * Rg[1][i] * Rg[1][j] + hv[2] * Rg[2][i] * Rg[2][j];
			if (C >= 11 && i >= 2)
			{
				for (int r = 0; r < 7; r++) acc += hv[4 + r] * Cg[r][i - 2] * Cg[r][j - 2];
			}
			qf[n++] = (i == j) ? acc : 2.0f * acc;
		}
	q[0] = qf[5]; q[1] = qf[8]; q[2] = qf[14]; q[3] = qf[1];
	q[4] = qf[4] + qf[7]; q[5] = qf[13]; q[6] = qf[0]; q[7] = qf[3] + qf[6];
	q[8] = qf[11] + qf[12]; q[9] = qf[2]; q[10] = qf[10]; q[11] = qf[9];
}

// k3 + F(u): Horner in ux over Horner in uy, 14 operations (the order of evaluation the walk of k_fisher_tile_v3 uses)
FR_HD float fr_scorer_poly_eval(const float q[12], float k3, float ux, float uy)
{
	const float A0 = q[0] + uy * (q[1] + uy * q[2]);
	const float A1 = q[3] + uy * (q[4] + uy * q[5]);
	const float A2 = q[6] + uy * (q[7] + uy * q[8]);
	const float A3 = q[9] + uy * q[10];
	const float in3 = A3 + ux * q[11];
	const float in2 = A2 + ux * in3;
	const float in1 = uy * A1 + ux * in2;
	return (k3 + (uy * uy) * A0) + ux * in1;
}

// ---- SH backward, backward.cu:20-139 -----------------------------------------------------------------
// dL_dsh[k] = coef[k] * (dL_dRGB masked by the clamp flags); the mean receives Dm * dL_dRGB (3x3).
// `sh` is the pointer the fused kernel passes (shs + M*global_id floats, backward.cu:1067).
FR_HD void fr_sh_backward_jacobian(int deg, fr_f3 pos, fr_f3 campos, const float* sh, const uint8_t* clamped3,
                                   float coef[16], float Dm[3][3])
{
	fr_f3 d0 = { pos.x - campos.x, pos.y - campos.y, pos.z - campos.z };
	const float len = sqrtf(d0.x * d0.x + d0.y * d0.y + d0.z * d0.z);
	const float x = d0.x / len, y = d0.y / len, z = d0.z / len;
	float dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
	for (int k = 0; k < 16; k++) coef[k] = 0.f;
	coef[0] = FR_SH_C0;
#define FR_S(k, c) sh[3 * (k) + (c)]
	if (deg > 0)
	{
		coef[1] = -FR_SH_C1 * y; coef[2] = FR_SH_C1 * z; coef[3] = -FR_SH_C1 * x;
		for (int c = 0; c < 3; c++)
		{
			dRGBdx[c] = -FR_SH_C1 * FR_S(3, c);
			dRGBdy[c] = -FR_SH_C1 * FR_S(1, c);
			dRGBdz[c] = FR_SH_C1 * FR_S(2, c);
		}
		if (deg > 1)
		{
			float xx = x * x, yy = y * y, zz = z * z;
			float xy = x * y, yz = y * z, xz = x * z;
			coef[4] = FR_SH_C2_0 * xy; coef[5] = FR_SH_C2_1 * yz; coef[6] = FR_SH_C2_2 * (2.f * zz - xx - yy);
			coef[7] = FR_SH_C2_3 * xz; coef[8] = FR_SH_C2_4 * (xx - yy);
			for (int c = 0; c < 3; c++)
			{
				dRGBdx[c] += FR_SH_C2_0 * y * FR_S(4, c) + FR_SH_C2_2 * 2.f * -x * FR_S(6, c) + FR_SH_C2_3 * z * FR_S(7, c) + FR_SH_C2_4 * 2.f * x * FR_S(8, c);
				dRGBdy[c] += FR_SH_C2_0 * x * FR_S(4, c) + FR_SH_C2_1 * z * FR_S(5, c) + FR_SH_C2_2 * 2.f * -y * FR_S(6, c) + FR_SH_C2_4 * 2.f * -y * FR_S(8, c);
				dRGBdz[c] += FR_SH_C2_1 * y * FR_S(5, c) + FR_SH_C2_2 * 2.f * 2.f * z * FR_S(6, c) + FR_SH_C2_3 * x * FR_S(7, c);
			}
			if (deg > 2)
			{
				coef[9] = FR_SH_C3_0 * y * (3.f * xx - yy);
				coef[10] = FR_SH_C3_1 * xy * z;
				coef[11] = FR_SH_C3_2 * y * (4.f * zz - xx - yy);
				coef[12] = FR_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
				coef[13] = FR_SH_C3_4 * x * (4.f * zz - xx - yy);
				coef[14] = FR_SH_C3_5 * z * (xx - yy);
				coef[15] = FR_SH_C3_6 * x * (xx - 3.f * yy);
				for (int c = 0; c < 3; c++)
				{
					dRGBdx[c] += (
						FR_SH_C3_0 * FR_S(9, c) * 3.f * 2.f * xy +
						FR_SH_C3_1 * FR_S(10, c) * yz +
						FR_SH_C3_2 * FR_S(11, c) * -2.f * xy +
						FR_SH_C3_3 * FR_S(12, c) * -3.f * 2.f * xz +
						FR_SH_C3_4 * FR_S(13, c) * (-3.f * xx + 4.f * zz - yy) +
						FR_SH_C3_5 * FR_S(14, c) * 2.f * xz +
						FR_SH_C3_6 * FR_S(15, c) * 3.f * (xx - yy));
					dRGBdy[c] += (
						FR_SH_C3_0 * FR_S(9, c) * 3.f * (xx - yy) +
						FR_SH_C3_1 * FR_S(10, c) * xz +
						FR_SH_C3_2 * FR_S(11, c) * (-3.f * yy + 4.f * zz - xx) +
						FR_SH_C3_3 * FR_S(12, c) * -3.f * 2.f * yz +
						FR_SH_C3_4 * FR_S(13, c) * -2.f * xy +
						FR_SH_C3_5 * FR_S(14, c) * -2.f * yz +
						FR_SH_C3_6 * FR_S(15, c) * -3.f * 2.f * xy);
					dRGBdz[c] += (
						FR_SH_C3_1 * FR_S(10, c) * xy +
						FR_SH_C3_2 * FR_S(11, c) * 4.f * 2.f * yz +
						FR_SH_C3_3 * FR_S(12, c) * 3.f * (2.f * zz - xx - yy) +
						FR_SH_C3_4 * FR_S(13, c) * 4.f * 2.f * xz +
						FR_SH_C3_5 * FR_S(14, c) * (xx - yy));
				}
			}
		}
	}
#undef FR_S
	// dL_ddir = (dRGBdx . g, dRGBdy . g, dRGBdz . g) with g masked by clamp; dL_dmean = dnormvdv(d0, dL_ddir)
	const float sum2 = d0.x * d0.x + d0.y * d0.y + d0.z * d0.z;
	const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
	for (int c = 0; c < 3; c++)
	{
		const float mask = clamped3[c] ? 0.f : 1.f;
		const float ddx = dRGBdx[c] * mask, ddy = dRGBdy[c] * mask, ddz = dRGBdz[c] * mask;
		Dm[0][c] = ((+sum2 - d0.x * d0.x) * ddx - d0.y * d0.x * ddy - d0.z * d0.x * ddz) * invsum32;
		Dm[1][c] = (-d0.x * d0.y * ddx + (sum2 - d0.y * d0.y) * ddy - d0.z * d0.y * ddz) * invsum32;
		Dm[2][c] = (-d0.x * d0.z * ddx - d0.y * d0.z * ddy + (sum2 - d0.z * d0.z) * ddz) * invsum32;
	}
}

/*
 * oracle/fisher_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, single thread) of the reference's
 * differentiable 3D-Gaussian-splat rasteriser with the FisherRF `grad_power`
 * modification.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path
 * (fisher-nerf-customized_amd/) never imports, links or calls it.
 *
 * Reference files restated (paths relative to /root/reference,
 * RAST = thirdparty/diff-gaussian-rasterization-modified):
 *   RAST/cuda_rasterizer/auxiliary.h:41-77,89-97,107-164   (ndc2Pix, getRect, transforms, in_frustum)
 *   RAST/cuda_rasterizer/forward.cu:20-71,74-113,118-152,155-256,261-393
 *   RAST/cuda_rasterizer/rasterizer_impl.cu:35-50,54-66,70-111,116-138,198-339,343-434
 *   RAST/cuda_rasterizer/backward.cu:20-139,276-408,412-475,532-583,850-1140
 *   RAST/rasterize_points.cu:35-217 (shapes / zero initialisation)
 *
 * Parity status: the reference ships no golden vectors or tests for this path
 *  and its CUDA sources cannot be built here (no nvcc, GLM
 * submodule empty).  The restatement is pinned by (a) torch-autograd of an
 * independent dense renderer for power=1, (b) the exact power-2 identity
 * Fisher == sum_pixels grad(power=1, one-hot pixel)^2, (c) hand-computed
 * single-Gaussian answers -- see tests/test_oracle_*.py.  It is NOT pinned
 * against outputs of the CUDA reference itself ("parity unpinned" in that
 * sense; recorded in DESIGN.md).
 *
 * Arithmetic conventions (so that a second implementation can be bit-exact):
 *   - every expression is evaluated in the operand order of the reference
 *     source, in IEEE binary32, WITHOUT fused-multiply-add contraction
 *     (build with -ffp-contract=off); nvcc would contract some of these, in a
 *     compiler-dependent pattern that cannot be reproduced, so "bit-exact
 *     against CUDA" is only meaningful for the integer outputs;
 *   - GLM (absent from the reference tree) is restated from its published
 *     semantics: column-major mat3, M[i][j] = column i / row j, operator* as in
 *     glm/detail/type_mat3x3.inl (Result[c][r] = m1[0][r]*m2[c][0] +
 *     m1[1][r]*m2[c][1] + m1[2][r]*m2[c][2]);
 *   - exp() is orc_expf below: a fixed sequence of IEEE operations (Cody-Waite
 *     reduction + degree-5 polynomial in fmaf), <= 1 ulp from the true value,
 *     standing in for CUDA's expf (<= 2 ulp, implementation not public);
 *   - float->int conversions saturate and map NaN to 0 (PTX cvt.rzi.s32.f32);
 *   - powf(x, grad_power) is x for 1 and x*x for 2 (identical for finite x up
 *     to CUDA powf's own <= 2 ulp), powf() otherwise;
 *   - gradient sums are accumulated in double in a fixed order and rounded to
 *     float once (the reference uses float atomicAdd in arbitrary order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define ORC_BLOCK_X 16
#define ORC_BLOCK_Y 16
#define ORC_NUM_CHANNELS 3

/* ---------------------------------------------------------------------- */
/* `real`: the arithmetic type.  float = the oracle proper (liboracle.so).
 * -DORC_DOUBLE builds the ARBITER (liboracle64.so): the same statements
 * evaluated in binary64 on the same binary32 inputs and constants (a literal
 * such as 0.3f stays the float constant, promoted), with every DECISION --
 * culling, radii, tile rectangles, sort order, n_contrib, and the per-pair
 * `power > 0` / `alpha < 1/255` tests -- taken from the binary32 run, so that
 * both builds sum over identical contributor sets and their difference is
 * the rounding error of the binary32 chain alone.  f32 = always binary32. */
/* ---------------------------------------------------------------------- */
typedef float f32;
#ifdef ORC_DOUBLE
typedef double real;
#define R_SQRT sqrt
#define R_MIN fmin
#define R_MAX fmax
#define R_CEIL ceil
#else
typedef float real;
#define R_SQRT sqrtf
#define R_MIN fminf
#define R_MAX fmaxf
#define R_CEIL ceilf
#endif

/* ---------------------------------------------------------------------- */
/* deterministic helpers                                                   */
/* ---------------------------------------------------------------------- */

static inline int orc_f2i(real f)
{
	if (f != f) return 0;
	if (f >= 2147483648.0f) return INT_MAX;
	if (f <= -2147483648.0f) return INT_MIN;
	return (int)f;
}

static inline f32 orc_bits2f(uint32_t u) { f32 f; memcpy(&f, &u, 4); return f; }
static inline uint32_t orc_f2bits(f32 f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* 2^k for k in [-126, 127] */
static inline f32 orc_pow2i(int k) { return orc_bits2f((uint32_t)(k + 127) << 23); }

f32 orc_expf(f32 x)
{
	if (x != x) return x;
	if (x > 88.72283905206835f) return INFINITY;
	if (x < -103.97208f) return 0.0f;
	f32 kf = rintf(x * 1.44269504088896341f);
	f32 r = fmaf(kf, -0.693359375f, x);
	r = fmaf(kf, 2.12194440e-4f, r);
	f32 p = 1.9875691500e-4f;
	p = fmaf(p, r, 1.3981999507e-3f);
	p = fmaf(p, r, 8.3334519073e-3f);
	p = fmaf(p, r, 4.1665795894e-2f);
	p = fmaf(p, r, 1.6666665459e-1f);
	p = fmaf(p, r, 5.0000001201e-1f);
	f32 r2 = r * r;
	f32 y = fmaf(p, r2, r) + 1.0f;
	int k = (int)kf;
	int k1 = k >> 1;          /* floor(k/2) */
	int k2 = k - k1;
	return (y * orc_pow2i(k1)) * orc_pow2i(k2);
}

#ifdef ORC_DOUBLE
static inline real orc_exp(real x) { return exp(x); }
#else
static inline real orc_exp(real x) { return orc_expf(x); }
#endif

/* The per-pair contributor tests of forward.cu:338-351 / backward.cu:985-996, ALWAYS in binary32 on the binary32
 * projection outputs (xy2 = means2D, co4 = conic_opacity of the float build): nonzero = the pair is skipped. */
static inline int orc_pair_skipped(const f32* xy2, const f32* co4, f32 pixfx, f32 pixfy)
{
	f32 dx = xy2[0] - pixfx, dy = xy2[1] - pixfy;
	f32 power = -0.5f * (co4[0] * dx * dx + co4[2] * dy * dy) - co4[1] * dx * dy;
	if (power > 0.0f)
		return 1;
	f32 alpha = fminf(0.99f, co4[3] * orc_expf(power));
	return alpha < 1.0f / 255.0f;
}

/* --- GLM restatement --------------------------------------------------- */
typedef struct { real x, y, z; } v3;
typedef struct { real x, y, z, w; } v4;
typedef struct { real m[3][3]; } m3; /* m[col][row] */

static inline m3 m3_cols(real a, real b, real c, real d, real e, real f, real g, real h, real i)
{
	m3 r;
	r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
	r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
	r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
	return r;
}
static inline m3 m3_mul(m3 a, m3 b)
{
	m3 r;
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++)
			r.m[c][rr] = a.m[0][rr] * b.m[c][0] + a.m[1][rr] * b.m[c][1] + a.m[2][rr] * b.m[c][2];
	return r;
}
static inline m3 m3_transpose(m3 a)
{
	m3 r;
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++)
			r.m[c][rr] = a.m[rr][c];
	return r;
}
static inline m3 m3_scale(real s, m3 a)
{
	m3 r;
	for (int c = 0; c < 3; c++)
		for (int rr = 0; rr < 3; rr++)
			r.m[c][rr] = a.m[c][rr] * s;
	return r;
}
static inline real v3_dot(v3 a, v3 b)
{
	real tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
	return tx + ty + tz;
}
static inline v3 m3_col(m3 a, int c) { v3 r = { a.m[c][0], a.m[c][1], a.m[c][2] }; return r; }

/* --- auxiliary.h ------------------------------------------------------- */
static const real SH_C0 = 0.28209479177387814f;
static const real SH_C1 = 0.4886025119029199f;
static const real SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
	-1.0925484305920792f, 0.5462742152960396f };
static const real SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
	0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };

/* auxiliary.h:41-44 : double-precision literals -> evaluated in double */
static inline real orc_ndc2pix(real v, int S)
{
	return (real)(((v + 1.0) * S - 1.0) * 0.5);
}

static inline uint32_t orc_umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline int orc_imax(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:46-56 */
static inline void orc_get_rect(real px, real py, int max_radius, uint32_t gx, uint32_t gy,
	uint32_t* minx, uint32_t* miny, uint32_t* maxx, uint32_t* maxy)
{
	*minx = orc_umin(gx, (uint32_t)orc_imax(0, orc_f2i((px - max_radius) / ORC_BLOCK_X)));
	*miny = orc_umin(gy, (uint32_t)orc_imax(0, orc_f2i((py - max_radius) / ORC_BLOCK_Y)));
	*maxx = orc_umin(gx, (uint32_t)orc_imax(0, orc_f2i((px + max_radius + ORC_BLOCK_X - 1) / ORC_BLOCK_X)));
	*maxy = orc_umin(gy, (uint32_t)orc_imax(0, orc_f2i((py + max_radius + ORC_BLOCK_Y - 1) / ORC_BLOCK_Y)));
}

/* auxiliary.h:58-77,89-97 */
static inline v3 orc_tp4x3(v3 p, const real* m)
{
	v3 t = {
		m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
		m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
		m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] };
	return t;
}
static inline v4 orc_tp4x4(v3 p, const real* m)
{
	v4 t = {
		m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
		m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
		m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
		m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] };
	return t;
}
static inline v3 orc_tv4x3T(v3 p, const real* m)
{
	v3 t = {
		m[0] * p.x + m[1] * p.y + m[2] * p.z,
		m[4] * p.x + m[5] * p.y + m[6] * p.z,
		m[8] * p.x + m[9] * p.y + m[10] * p.z };
	return t;
}
/* auxiliary.h:107-117 */
static inline v3 orc_dnormvdv(v3 v, v3 dv)
{
	real sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
	real invsum32 = 1.0f / R_SQRT(sum2 * sum2 * sum2);
	v3 r;
	r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
	r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
	r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
	return r;
}
/* auxiliary.h:139-164 (prefiltered trap not modelled: returns -1 instead) */
static inline int orc_in_frustum(v3 p_orig, const real* viewmatrix, v3* p_view)
{
	*p_view = orc_tp4x3(p_orig, viewmatrix);
	if (p_view->z <= 0.001f)
		return 0;
	return 1;
}

/* rasterizer_impl.cu:35-50 */
uint32_t orc_get_higher_msb(uint32_t n)
{
	uint32_t msb = sizeof(n) * 4;
	uint32_t step = msb;
	while (step > 1)
	{
		step /= 2;
		if (n >> msb)
			msb += step;
		else
			msb -= step;
	}
	if (n >> msb)
		msb++;
	return msb;
}

/* ---------------------------------------------------------------------- */
/* forward.cu                                                              */
/* ---------------------------------------------------------------------- */

/* forward.cu:20-71 ; means/shs are the full arrays, idx selects */
static v3 orc_color_from_sh(int idx, int deg, int max_coeffs, const real* means, v3 campos,
	const real* shs, uint8_t* clamped)
{
	v3 pos = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
	v3 dir = { pos.x - campos.x, pos.y - campos.y, pos.z - campos.z };
	real len = R_SQRT(v3_dot(dir, dir));
	dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;

	const real* sh = shs + 3 * (size_t)idx * max_coeffs; /* sh[k] = (sh[3k], sh[3k+1], sh[3k+2]) */
	real res[3];
	real x = dir.x, y = dir.y, z = dir.z;
	for (int c = 0; c < 3; c++)
	{
#define SH(k) sh[3 * (k) + c]
		real result = SH_C0 * SH(0);
		if (deg > 0)
		{
			result = result - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);
			if (deg > 1)
			{
				real xx = x * x, yy = y * y, zz = z * z;
				real xy = x * y, yz = y * z, xz = x * z;
				result = result +
					SH_C2[0] * xy * SH(4) +
					SH_C2[1] * yz * SH(5) +
					SH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
					SH_C2[3] * xz * SH(7) +
					SH_C2[4] * (xx - yy) * SH(8);
				if (deg > 2)
				{
					result = result +
						SH_C3[0] * y * (3.0f * xx - yy) * SH(9) +
						SH_C3[1] * xy * z * SH(10) +
						SH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
						SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
						SH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) +
						SH_C3[5] * z * (xx - yy) * SH(14) +
						SH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
				}
			}
		}
#undef SH
		result += 0.5f;
		res[c] = result;
	}
	clamped[3 * idx + 0] = (res[0] < 0);
	clamped[3 * idx + 1] = (res[1] < 0);
	clamped[3 * idx + 2] = (res[2] < 0);
	v3 out = { R_MAX(res[0], 0.0f), R_MAX(res[1], 0.0f), R_MAX(res[2], 0.0f) };
	return out;
}

/* shared by forward.cu:74-113 and backward.cu:300-333 */
typedef struct {
	v3 t;                 /* clamped camera-space mean */
	real txtz, tytz;
	m3 J, W, T, Vrk, cov; /* GLM-convention matrices, cov BEFORE the +0.3 */
} cov2d_ctx;

static void orc_cov2d_common(v3 mean, real focal_x, real focal_y, real tan_fovx, real tan_fovy,
	const real* cov3D, const real* viewmatrix, cov2d_ctx* c)
{
	v3 t = orc_tp4x3(mean, viewmatrix);
	const real limx = 1.3f * tan_fovx;
	const real limy = 1.3f * tan_fovy;
	c->txtz = t.x / t.z;
	c->tytz = t.y / t.z;
	t.x = R_MIN(limx, R_MAX(-limx, c->txtz)) * t.z;
	t.y = R_MIN(limy, R_MAX(-limy, c->tytz)) * t.z;
	c->t = t;
	c->J = m3_cols(
		focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z),
		0.0f, focal_y / t.z, -(focal_y * t.y) / (t.z * t.z),
		0, 0, 0);
	c->W = m3_cols(
		viewmatrix[0], viewmatrix[4], viewmatrix[8],
		viewmatrix[1], viewmatrix[5], viewmatrix[9],
		viewmatrix[2], viewmatrix[6], viewmatrix[10]);
	c->T = m3_mul(c->W, c->J);
	c->Vrk = m3_cols(
		cov3D[0], cov3D[1], cov3D[2],
		cov3D[1], cov3D[3], cov3D[4],
		cov3D[2], cov3D[4], cov3D[5]);
	c->cov = m3_mul(m3_mul(m3_transpose(c->T), m3_transpose(c->Vrk)), c->T);
}

/* shared by forward.cu:118-152 and backward.cu:415-434 */
static m3 orc_quat_R(v4 rot)
{
	real r = rot.x, x = rot.y, y = rot.z, z = rot.w; /* NOT normalised (forward.cu:127) */
	return m3_cols(
		1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
		2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
		2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
}
static m3 orc_identity(void) { return m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1); }

void orc_cov3d(const real* scale, real mod, const real* rot4, real* cov3D)
{
	m3 S = orc_identity();
	S.m[0][0] = mod * scale[0];
	S.m[1][1] = mod * scale[1];
	S.m[2][2] = mod * scale[2];
	v4 q = { rot4[0], rot4[1], rot4[2], rot4[3] };
	m3 R = orc_quat_R(q);
	m3 M = m3_mul(S, R);
	m3 Sigma = m3_mul(m3_transpose(M), M);
	cov3D[0] = Sigma.m[0][0];
	cov3D[1] = Sigma.m[0][1];
	cov3D[2] = Sigma.m[0][2];
	cov3D[3] = Sigma.m[1][1];
	cov3D[4] = Sigma.m[1][2];
	cov3D[5] = Sigma.m[2][2];
}

/* forward.cu:155-256.  Outputs for culled Gaussians are left untouched except
 * radii/tiles_touched = 0 (the reference leaves them uninitialised). */
void orc_preprocess(int P, int D, int M,
	const real* means3D, const real* scales, real scale_modifier, const real* rotations,
	const real* opacities, const real* shs, const real* cov3D_precomp, const real* colors_precomp,
	const real* viewmatrix, const real* projmatrix, const real* cam_pos,
	int W, int H, real tan_fovx, real tan_fovy,
	int32_t* radii, real* means2D, real* depths, real* cov3Ds, real* rgb,
	real* conic_opacity, uint32_t* tiles_touched, uint8_t* clamped, const int32_t* radii_fixed)
{
	/* radii_fixed (arbiter build only; NULL otherwise): the radii of the binary32 run.  A Gaussian that run culled is
	 * culled here, one it kept is kept with that radius; tiles_touched is not produced (binning is the float run's). */
	const real focal_y = H / (2.0f * tan_fovy);   /* rasterizer_impl.cu:222-223 */
	const real focal_x = W / (2.0f * tan_fovx);
	const uint32_t gx = (W + ORC_BLOCK_X - 1) / ORC_BLOCK_X, gy = (H + ORC_BLOCK_Y - 1) / ORC_BLOCK_Y;

	for (int idx = 0; idx < P; idx++)
	{
		radii[idx] = 0;
		tiles_touched[idx] = 0;
		if (radii_fixed && radii_fixed[idx] <= 0)
			continue;

		v3 p_orig = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
		v3 p_view;
		if (!orc_in_frustum(p_orig, viewmatrix, &p_view) && !radii_fixed)
			continue;

		v4 p_hom = orc_tp4x4(p_orig, projmatrix);
		real p_w = 1.0f / (p_hom.w + 0.0000001f);
		v3 p_proj = { p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w };

		const real* cov3D;
		if (cov3D_precomp != NULL)
			cov3D = cov3D_precomp + (size_t)idx * 6;
		else
		{
			orc_cov3d(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, cov3Ds + (size_t)idx * 6);
			cov3D = cov3Ds + (size_t)idx * 6;
		}

		cov2d_ctx cc;
		orc_cov2d_common(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &cc);
		cc.cov.m[0][0] += 0.3f;
		cc.cov.m[1][1] += 0.3f;
		real covx = cc.cov.m[0][0], covy = cc.cov.m[0][1], covz = cc.cov.m[1][1];

		real det = (covx * covz - covy * covy);
		if (det == 0.0f)
			continue;
		real det_inv = 1.f / det;
		real conx = covz * det_inv, cony = -covy * det_inv, conz = covx * det_inv;

		real mid = 0.5f * (covx + covz);
		real lambda1 = mid + R_SQRT(R_MAX(0.1f, mid * mid - det));
		real lambda2 = mid - R_SQRT(R_MAX(0.1f, mid * mid - det));
		real my_radius = R_CEIL(3.f * R_SQRT(R_MAX(lambda1, lambda2)));
		real pix_x = orc_ndc2pix(p_proj.x, W), pix_y = orc_ndc2pix(p_proj.y, H);
		uint32_t minx, miny, maxx, maxy;
		orc_get_rect(pix_x, pix_y, orc_f2i(my_radius), gx, gy, &minx, &miny, &maxx, &maxy);
		if ((maxx - minx) * (maxy - miny) == 0 && !radii_fixed)
			continue;

		if (colors_precomp == NULL)
		{
			v3 campos = { cam_pos[0], cam_pos[1], cam_pos[2] };
			v3 c = orc_color_from_sh(idx, D, M, means3D, campos, shs, clamped);
			rgb[idx * ORC_NUM_CHANNELS + 0] = c.x;
			rgb[idx * ORC_NUM_CHANNELS + 1] = c.y;
			rgb[idx * ORC_NUM_CHANNELS + 2] = c.z;
		}

		depths[idx] = p_view.z;
		radii[idx] = radii_fixed ? radii_fixed[idx] : orc_f2i(my_radius);
		means2D[2 * idx] = pix_x;
		means2D[2 * idx + 1] = pix_y;
		conic_opacity[4 * idx + 0] = conx;
		conic_opacity[4 * idx + 1] = cony;
		conic_opacity[4 * idx + 2] = conz;
		conic_opacity[4 * idx + 3] = opacities[idx];
		tiles_touched[idx] = (maxy - miny) * (maxx - minx);
	}
}

#ifndef ORC_DOUBLE   /* visibility, binning and sort are decisions: the float build's */
/* rasterizer_impl.cu:54-66 */
void orc_mark_visible(int P, const real* means3D, const real* viewmatrix, const real* projmatrix, uint8_t* present)
{
	(void)projmatrix;
	for (int idx = 0; idx < P; idx++)
	{
		v3 p = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
		v3 pv;
		present[idx] = (uint8_t)orc_in_frustum(p, viewmatrix, &pv);
	}
}

/* ---------------------------------------------------------------------- */
/* binning: rasterizer_impl.cu:277-319                                     */
/* ---------------------------------------------------------------------- */

/* stable LSD radix sort of (key,value) pairs over bits [0,end_bit) */
static void orc_radix_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp, size_t n, int end_bit)
{
	uint64_t* ka = keys; uint64_t* kb = keys_tmp;
	uint32_t* va = vals; uint32_t* vb = vals_tmp;
	for (int shift = 0; shift < end_bit; shift += 8)
	{
		int bits = end_bit - shift < 8 ? end_bit - shift : 8;
		uint32_t mask = (1u << bits) - 1;
		size_t count[257];
		memset(count, 0, sizeof(count));
		for (size_t i = 0; i < n; i++)
			count[((ka[i] >> shift) & mask) + 1]++;
		for (int b = 0; b < 256; b++)
			count[b + 1] += count[b];
		for (size_t i = 0; i < n; i++)
		{
			size_t d = count[(ka[i] >> shift) & mask]++;
			kb[d] = ka[i];
			vb[d] = va[i];
		}
		uint64_t* tk = ka; ka = kb; kb = tk;
		uint32_t* tv = va; va = vb; vb = tv;
	}
	if (ka != keys)
	{
		memcpy(keys, ka, n * sizeof(uint64_t));
		memcpy(vals, va, n * sizeof(uint32_t));
	}
}

/* Returns num_rendered.  If point_list == NULL only counts.
 * keys_sorted[R], point_list[R], ranges[2*tiles] (x,y interleaved). */
int64_t orc_bin(int P, const real* means2D, const real* depths, const int32_t* radii,
	const uint32_t* tiles_touched, int W, int H,
	uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges)
{
	const uint32_t gx = (W + ORC_BLOCK_X - 1) / ORC_BLOCK_X, gy = (H + ORC_BLOCK_Y - 1) / ORC_BLOCK_Y;
	/* InclusiveSum (rasterizer_impl.cu:277) in uint32 like the reference */
	uint32_t* offsets = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(P > 0 ? P : 1));
	uint32_t acc = 0;
	for (int i = 0; i < P; i++) { acc += tiles_touched[i]; offsets[i] = acc; }
	int64_t R = P > 0 ? (int64_t)(int32_t)offsets[P - 1] : 0; /* int num_rendered */
	if (point_list == NULL) { free(offsets); return R; }

	uint64_t* keys_tmp = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
	uint32_t* vals_tmp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));

	/* duplicateWithKeys (rasterizer_impl.cu:70-111) */
	for (int idx = 0; idx < P; idx++)
	{
		if (radii[idx] > 0)
		{
			uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
			uint32_t minx, miny, maxx, maxy;
			orc_get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, &minx, &miny, &maxx, &maxy);
			for (int y = (int)miny; y < (int)maxy; y++)
				for (int x = (int)minx; x < (int)maxx; x++)
				{
					uint64_t key = (uint64_t)(y * gx + x);
					key <<= 32;
					key |= orc_f2bits(depths[idx]);
					keys_sorted[off] = key;
					point_list[off] = (uint32_t)idx;
					off++;
				}
		}
	}

	int bit = (int)orc_get_higher_msb(gx * gy);
	orc_radix_sort_pairs(keys_sorted, point_list, keys_tmp, vals_tmp, (size_t)R, 32 + bit);

	/* cudaMemset + identifyTileRanges (rasterizer_impl.cu:311-319, 116-138) */
	memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
	for (int64_t idx = 0; idx < R; idx++)
	{
		uint32_t currtile = (uint32_t)(keys_sorted[idx] >> 32);
		if (idx == 0)
			ranges[2 * currtile] = 0;
		else
		{
			uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
			if (currtile != prevtile)
			{
				ranges[2 * prevtile + 1] = (uint32_t)idx;
				ranges[2 * currtile] = (uint32_t)idx;
			}
		}
		if (idx == R - 1)
			ranges[2 * currtile + 1] = (uint32_t)R;
	}
	free(offsets); free(keys_tmp); free(vals_tmp);
	return R;
}

#endif /* !ORC_DOUBLE */

/* ---------------------------------------------------------------------- */
/* forward render: forward.cu:261-393                                      */
/* ---------------------------------------------------------------------- */
void orc_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
	const real* means2D, const real* features, const real* conic_opacity, const real* depths,
	const real* bg_color, real* final_T, uint32_t* n_contrib, real* out_color, real* out_depth,
	const f32* dec_means2D, const f32* dec_conic_opacity, const uint32_t* dec_n_contrib)
{
	/* dec_* (arbiter build only; NULL otherwise): means2D / conic_opacity / n_contrib of the binary32 run.  With them a
	 * pixel composites exactly the float run's contributors: its per-pair tests and its last contributor (the median
	 * depth, which no Fisher quantity reads, is still chosen by this build's own transmittance). */
	const uint32_t gx = (W + ORC_BLOCK_X - 1) / ORC_BLOCK_X, gy = (H + ORC_BLOCK_Y - 1) / ORC_BLOCK_Y;
	for (uint32_t ty = 0; ty < gy; ty++)
	for (uint32_t tx = 0; tx < gx; tx++)
	{
		uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
		for (uint32_t ly = 0; ly < ORC_BLOCK_Y; ly++)
		for (uint32_t lx = 0; lx < ORC_BLOCK_X; lx++)
		{
			uint32_t pxx = tx * ORC_BLOCK_X + lx, pxy = ty * ORC_BLOCK_Y + ly;
			if (!(pxx < (uint32_t)W && pxy < (uint32_t)H))
				continue;
			uint32_t pix_id = W * pxy + pxx;
			real pixfx = (real)pxx, pixfy = (real)pxy;

			real T = 1.0f;
			uint32_t contributor = 0, last_contributor = 0;
			real C[ORC_NUM_CHANNELS] = { 0 };
			real D = 15.0f;  /* median depth default (forward.cu:308) */

			for (uint32_t k = r0; k < r1; k++)
			{
				contributor++;
				uint32_t id = point_list[k];
				real dx = means2D[2 * id] - pixfx, dy = means2D[2 * id + 1] - pixfy;
				const real* con_o = conic_opacity + 4 * (size_t)id;
				real power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
				if (dec_means2D)
				{
					if (contributor > dec_n_contrib[pix_id])
						break;
					if (orc_pair_skipped(dec_means2D + 2 * (size_t)id, dec_conic_opacity + 4 * (size_t)id, (f32)pxx, (f32)pxy))
						continue;
				}
				else if (power > 0.0f)
					continue;
				real alpha = R_MIN(0.99f, con_o[3] * orc_exp(power));
				if (!dec_means2D && alpha < 1.0f / 255.0f)
					continue;
				real test_T = T * (1 - alpha);
				if (!dec_means2D && test_T < 0.0001f)
					break; /* done = true */
				for (int ch = 0; ch < ORC_NUM_CHANNELS; ch++)
					C[ch] += features[id * ORC_NUM_CHANNELS + ch] * alpha * T;
				if (T > 0.5f && test_T < 0.5)
					D = depths[id];
				T = test_T;
				last_contributor = contributor;
			}
			final_T[pix_id] = T;
			n_contrib[pix_id] = last_contributor;
			for (int ch = 0; ch < ORC_NUM_CHANNELS; ch++)
				out_color[ch * H * W + pix_id] = C[ch] + T * bg_color[ch];
			out_depth[pix_id] = D;
		}
	}
}

/* ---------------------------------------------------------------------- */
/* backward                                                                */
/* ---------------------------------------------------------------------- */

static inline double orc_pow(real x, int power)
{
	if (power == 1) return (double)x;
	if (power == 2) return (double)(x * x);
	return (double)powf((f32)x, (f32)power);
}

/* backward.cu:276-408 with idx == 0 on relocated pointers */
static void orc_cov2d_backward(v3 mean, int radius, const real* cov3D, real h_x, real h_y,
	real tan_fovx, real tan_fovy, const real* view_matrix, const real* dL_dconics /*float4*/,
	v3* dL_dmeans, real* dL_dcov /*6*/)
{
	if (!(radius > 0))
		return;
	v3 dL_dconic = { dL_dconics[0], dL_dconics[1], dL_dconics[3] };
	cov2d_ctx cc;
	orc_cov2d_common(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view_matrix, &cc);
	const real limx = 1.3f * tan_fovx;
	const real limy = 1.3f * tan_fovy;
	const real x_grad_mul = cc.txtz < -limx || cc.txtz > limx ? 0 : 1;
	const real y_grad_mul = cc.tytz < -limy || cc.tytz > limy ? 0 : 1;
	v3 t = cc.t;
#define T_(i,j) cc.T.m[i][j]
#define W_(i,j) cc.W.m[i][j]
#define V_(i,j) cc.Vrk.m[i][j]
	real a = cc.cov.m[0][0] += 0.3f;
	real b = cc.cov.m[0][1];
	real c = cc.cov.m[1][1] += 0.3f;

	real denom = a * c - b * b;
	real dL_da = 0, dL_db = 0, dL_dc = 0;
	real denom2inv = 1.0f / ((denom * denom) + 0.0000001f);

	if (denom2inv != 0)
	{
		dL_da = denom2inv * (-c * c * dL_dconic.x + 2 * b * c * dL_dconic.y + (denom - a * c) * dL_dconic.z);
		dL_dc = denom2inv * (-a * a * dL_dconic.z + 2 * a * b * dL_dconic.y + (denom - a * c) * dL_dconic.x);
		dL_db = denom2inv * 2 * (b * c * dL_dconic.x - (denom + 2 * b * b) * dL_dconic.y + a * b * dL_dconic.z);

		dL_dcov[0] = (T_(0,0) * T_(0,0) * dL_da + T_(0,0) * T_(1,0) * dL_db + T_(1,0) * T_(1,0) * dL_dc);
		dL_dcov[3] = (T_(0,1) * T_(0,1) * dL_da + T_(0,1) * T_(1,1) * dL_db + T_(1,1) * T_(1,1) * dL_dc);
		dL_dcov[5] = (T_(0,2) * T_(0,2) * dL_da + T_(0,2) * T_(1,2) * dL_db + T_(1,2) * T_(1,2) * dL_dc);

		dL_dcov[1] = 2 * T_(0,0) * T_(0,1) * dL_da + (T_(0,0) * T_(1,1) + T_(0,1) * T_(1,0)) * dL_db + 2 * T_(1,0) * T_(1,1) * dL_dc;
		dL_dcov[2] = 2 * T_(0,0) * T_(0,2) * dL_da + (T_(0,0) * T_(1,2) + T_(0,2) * T_(1,0)) * dL_db + 2 * T_(1,0) * T_(1,2) * dL_dc;
		dL_dcov[4] = 2 * T_(0,2) * T_(0,1) * dL_da + (T_(0,1) * T_(1,2) + T_(0,2) * T_(1,1)) * dL_db + 2 * T_(1,1) * T_(1,2) * dL_dc;
	}
	else
	{
		for (int i = 0; i < 6; i++)
			dL_dcov[i] = 0;
	}

	real dL_dT00 = 2 * (T_(0,0) * V_(0,0) + T_(0,1) * V_(0,1) + T_(0,2) * V_(0,2)) * dL_da +
		(T_(1,0) * V_(0,0) + T_(1,1) * V_(0,1) + T_(1,2) * V_(0,2)) * dL_db;
	real dL_dT01 = 2 * (T_(0,0) * V_(1,0) + T_(0,1) * V_(1,1) + T_(0,2) * V_(1,2)) * dL_da +
		(T_(1,0) * V_(1,0) + T_(1,1) * V_(1,1) + T_(1,2) * V_(1,2)) * dL_db;
	real dL_dT02 = 2 * (T_(0,0) * V_(2,0) + T_(0,1) * V_(2,1) + T_(0,2) * V_(2,2)) * dL_da +
		(T_(1,0) * V_(2,0) + T_(1,1) * V_(2,1) + T_(1,2) * V_(2,2)) * dL_db;
	real dL_dT10 = 2 * (T_(1,0) * V_(0,0) + T_(1,1) * V_(0,1) + T_(1,2) * V_(0,2)) * dL_dc +
		(T_(0,0) * V_(0,0) + T_(0,1) * V_(0,1) + T_(0,2) * V_(0,2)) * dL_db;
	real dL_dT11 = 2 * (T_(1,0) * V_(1,0) + T_(1,1) * V_(1,1) + T_(1,2) * V_(1,2)) * dL_dc +
		(T_(0,0) * V_(1,0) + T_(0,1) * V_(1,1) + T_(0,2) * V_(1,2)) * dL_db;
	real dL_dT12 = 2 * (T_(1,0) * V_(2,0) + T_(1,1) * V_(2,1) + T_(1,2) * V_(2,2)) * dL_dc +
		(T_(0,0) * V_(2,0) + T_(0,1) * V_(2,1) + T_(0,2) * V_(2,2)) * dL_db;

	real dL_dJ00 = W_(0,0) * dL_dT00 + W_(0,1) * dL_dT01 + W_(0,2) * dL_dT02;
	real dL_dJ02 = W_(2,0) * dL_dT00 + W_(2,1) * dL_dT01 + W_(2,2) * dL_dT02;
	real dL_dJ11 = W_(1,0) * dL_dT10 + W_(1,1) * dL_dT11 + W_(1,2) * dL_dT12;
	real dL_dJ12 = W_(2,0) * dL_dT10 + W_(2,1) * dL_dT11 + W_(2,2) * dL_dT12;
#undef T_
#undef W_
#undef V_
	real tz = 1.f / t.z;
	real tz2 = tz * tz;
	real tz3 = tz2 * tz;

	real dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
	real dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
	real dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;

	v3 dt = { dL_dtx, dL_dty, dL_dtz };
	*dL_dmeans = orc_tv4x3T(dt, view_matrix); /* ASSIGN (backward.cu:407) */
}

/* backward.cu:412-475 */
static void orc_cov3d_backward(const real* scale, real mod, const real* rot4, const real* dL_dcov3D,
	v3* dL_dscale, v4* dL_drot)
{
	v4 q = { rot4[0], rot4[1], rot4[2], rot4[3] };
	real r = q.x, x = q.y, y = q.z, z = q.w;
	m3 R = orc_quat_R(q);
	m3 S = orc_identity();
	v3 s = { mod * scale[0], mod * scale[1], mod * scale[2] };
	S.m[0][0] = s.x;
	S.m[1][1] = s.y;
	S.m[2][2] = s.z;
	m3 M = m3_mul(S, R);

	m3 dL_dSigma = m3_cols(
		dL_dcov3D[0], 0.5f * dL_dcov3D[1], 0.5f * dL_dcov3D[2],
		0.5f * dL_dcov3D[1], dL_dcov3D[3], 0.5f * dL_dcov3D[4],
		0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[4], dL_dcov3D[5]);

	m3 dL_dM = m3_mul(m3_scale(2.0f, M), dL_dSigma);
	m3 Rt = m3_transpose(R);
	m3 dL_dMt = m3_transpose(dL_dM);

	dL_dscale->x = v3_dot(m3_col(Rt, 0), m3_col(dL_dMt, 0));
	dL_dscale->y = v3_dot(m3_col(Rt, 1), m3_col(dL_dMt, 1));
	dL_dscale->z = v3_dot(m3_col(Rt, 2), m3_col(dL_dMt, 2));

	for (int j = 0; j < 3; j++) { dL_dMt.m[0][j] *= s.x; dL_dMt.m[1][j] *= s.y; dL_dMt.m[2][j] *= s.z; }
#define D_(i,j) dL_dMt.m[i][j]
	v4 dL_dq;
	dL_dq.x = 2 * z * (D_(0,1) - D_(1,0)) + 2 * y * (D_(2,0) - D_(0,2)) + 2 * x * (D_(1,2) - D_(2,1));
	dL_dq.y = 2 * y * (D_(1,0) + D_(0,1)) + 2 * z * (D_(2,0) + D_(0,2)) + 2 * r * (D_(1,2) - D_(2,1)) - 4 * x * (D_(2,2) + D_(1,1));
	dL_dq.z = 2 * x * (D_(1,0) + D_(0,1)) + 2 * r * (D_(2,0) - D_(0,2)) + 2 * z * (D_(1,2) + D_(2,1)) - 4 * y * (D_(2,2) + D_(0,0));
	dL_dq.w = 2 * r * (D_(0,1) - D_(1,0)) + 2 * x * (D_(2,0) + D_(0,2)) + 2 * y * (D_(1,2) + D_(2,1)) - 4 * z * (D_(1,1) + D_(0,0));
#undef D_
	*dL_drot = dL_dq; /* no normalisation backward (backward.cu:474) */
}

/* backward.cu:20-139 with idx == 0 on relocated pointers.
 * sh : real pointer as passed by the fused kernel (shs + M*global_id, backward.cu:1067 -- a real
 *      offset, not a vec3 offset; reproduced as is), read as vec3 sh[k] = sh[3k..3k+2].
 * dL_dsh : local scratch of 16 vec3. */
static void orc_sh_backward(int deg, v3 pos, v3 campos, const real* sh, const uint8_t* clamped,
	const real* dL_dcolor3, v3* dL_dmeans, real* dL_dsh /*16*3*/)
{
	v3 dir_orig = { pos.x - campos.x, pos.y - campos.y, pos.z - campos.z };
	real len = R_SQRT(v3_dot(dir_orig, dir_orig));
	v3 dir = { dir_orig.x / len, dir_orig.y / len, dir_orig.z / len };

	real dL_dRGB[3] = { dL_dcolor3[0], dL_dcolor3[1], dL_dcolor3[2] };
	dL_dRGB[0] *= clamped[0] ? 0 : 1;
	dL_dRGB[1] *= clamped[1] ? 0 : 1;
	dL_dRGB[2] *= clamped[2] ? 0 : 1;

	real dRGBdx[3] = { 0, 0, 0 }, dRGBdy[3] = { 0, 0, 0 }, dRGBdz[3] = { 0, 0, 0 };
	real x = dir.x, y = dir.y, z = dir.z;
#define SHV(k, c) sh[3 * (k) + (c)]
#define SET(k, coef) for (int c_ = 0; c_ < 3; c_++) dL_dsh[3 * (k) + c_] = (coef) * dL_dRGB[c_]
	real dRGBdsh0 = SH_C0;
	SET(0, dRGBdsh0);
	if (deg > 0)
	{
		real dRGBdsh1 = -SH_C1 * y;
		real dRGBdsh2 = SH_C1 * z;
		real dRGBdsh3 = -SH_C1 * x;
		SET(1, dRGBdsh1);
		SET(2, dRGBdsh2);
		SET(3, dRGBdsh3);
		for (int c = 0; c < 3; c++)
		{
			dRGBdx[c] = -SH_C1 * SHV(3, c);
			dRGBdy[c] = -SH_C1 * SHV(1, c);
			dRGBdz[c] = SH_C1 * SHV(2, c);
		}
		if (deg > 1)
		{
			real xx = x * x, yy = y * y, zz = z * z;
			real xy = x * y, yz = y * z, xz = x * z;
			real dRGBdsh4 = SH_C2[0] * xy;
			real dRGBdsh5 = SH_C2[1] * yz;
			real dRGBdsh6 = SH_C2[2] * (2.f * zz - xx - yy);
			real dRGBdsh7 = SH_C2[3] * xz;
			real dRGBdsh8 = SH_C2[4] * (xx - yy);
			SET(4, dRGBdsh4);
			SET(5, dRGBdsh5);
			SET(6, dRGBdsh6);
			SET(7, dRGBdsh7);
			SET(8, dRGBdsh8);
			for (int c = 0; c < 3; c++)
			{
				dRGBdx[c] += SH_C2[0] * y * SHV(4, c) + SH_C2[2] * 2.f * -x * SHV(6, c) + SH_C2[3] * z * SHV(7, c) + SH_C2[4] * 2.f * x * SHV(8, c);
				dRGBdy[c] += SH_C2[0] * x * SHV(4, c) + SH_C2[1] * z * SHV(5, c) + SH_C2[2] * 2.f * -y * SHV(6, c) + SH_C2[4] * 2.f * -y * SHV(8, c);
				dRGBdz[c] += SH_C2[1] * y * SHV(5, c) + SH_C2[2] * 2.f * 2.f * z * SHV(6, c) + SH_C2[3] * x * SHV(7, c);
			}
			if (deg > 2)
			{
				real dRGBdsh9 = SH_C3[0] * y * (3.f * xx - yy);
				real dRGBdsh10 = SH_C3[1] * xy * z;
				real dRGBdsh11 = SH_C3[2] * y * (4.f * zz - xx - yy);
				real dRGBdsh12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
				real dRGBdsh13 = SH_C3[4] * x * (4.f * zz - xx - yy);
				real dRGBdsh14 = SH_C3[5] * z * (xx - yy);
				real dRGBdsh15 = SH_C3[6] * x * (xx - 3.f * yy);
				SET(9, dRGBdsh9);
				SET(10, dRGBdsh10);
				SET(11, dRGBdsh11);
				SET(12, dRGBdsh12);
				SET(13, dRGBdsh13);
				SET(14, dRGBdsh14);
				SET(15, dRGBdsh15);
				for (int c = 0; c < 3; c++)
				{
					dRGBdx[c] += (
						SH_C3[0] * SHV(9, c) * 3.f * 2.f * xy +
						SH_C3[1] * SHV(10, c) * yz +
						SH_C3[2] * SHV(11, c) * -2.f * xy +
						SH_C3[3] * SHV(12, c) * -3.f * 2.f * xz +
						SH_C3[4] * SHV(13, c) * (-3.f * xx + 4.f * zz - yy) +
						SH_C3[5] * SHV(14, c) * 2.f * xz +
						SH_C3[6] * SHV(15, c) * 3.f * (xx - yy));
					dRGBdy[c] += (
						SH_C3[0] * SHV(9, c) * 3.f * (xx - yy) +
						SH_C3[1] * SHV(10, c) * xz +
						SH_C3[2] * SHV(11, c) * (-3.f * yy + 4.f * zz - xx) +
						SH_C3[3] * SHV(12, c) * -3.f * 2.f * yz +
						SH_C3[4] * SHV(13, c) * -2.f * xy +
						SH_C3[5] * SHV(14, c) * -2.f * yz +
						SH_C3[6] * SHV(15, c) * -3.f * 2.f * xy);
					dRGBdz[c] += (
						SH_C3[1] * SHV(10, c) * xy +
						SH_C3[2] * SHV(11, c) * 4.f * 2.f * yz +
						SH_C3[3] * SHV(12, c) * 3.f * (2.f * zz - xx - yy) +
						SH_C3[4] * SHV(13, c) * 4.f * 2.f * xz +
						SH_C3[5] * SHV(14, c) * (xx - yy));
				}
			}
		}
	}
#undef SHV
#undef SET
	v3 vx = { dRGBdx[0], dRGBdx[1], dRGBdx[2] }, vy = { dRGBdy[0], dRGBdy[1], dRGBdy[2] }, vz = { dRGBdz[0], dRGBdz[1], dRGBdz[2] };
	v3 g = { dL_dRGB[0], dL_dRGB[1], dL_dRGB[2] };
	v3 dL_ddir = { v3_dot(vx, g), v3_dot(vy, g), v3_dot(vz, g) };
	v3 dL_dmean = orc_dnormvdv(dir_orig, dL_ddir);
	dL_dmeans->x += dL_dmean.x;
	dL_dmeans->y += dL_dmean.y;
	dL_dmeans->z += dL_dmean.z;
}

/* The leaf gradients of ONE (pixel, Gaussian) pair as renderCUDAFused forms them (backward.cu:1016-1090), for a given
 * w = opacity * G * dL_dalpha and d = mean2D - pixel: out[11] = [mean xyz | opacity | scale xyz | rot rxyz].  Tests of the
 * kernels' per-Gaussian record algebra use it (binary32: the reference's own rounding; arbiter build: the exact value). */
void orc_pair_leaves(const real* mean3, const real* cov3D, const real* scale, real mod, const real* rot4,
	const real* viewmatrix, const real* projmatrix, int W, int H, real tan_fovx, real tan_fovy,
	const real* con_o, real dx, real dy, real w, real* out)
{
	const real focal_y = H / (2.0f * tan_fovy);
	const real focal_x = W / (2.0f * tan_fovx);
	const real ddelx_dx = (real)(0.5 * W);
	const real ddely_dy = (real)(0.5 * H);
	/* gdx * dL_dG = G dx * opacity dL_dalpha = w dx */
	const real wdx = w * dx, wdy = w * dy;
	real cur_dL_dmean2D[2], cur_dL_dconic2D[4] = { 0, 0, 0, 0 };
	cur_dL_dmean2D[0] = (-wdx * con_o[0] - wdy * con_o[1]) * ddelx_dx;
	cur_dL_dmean2D[1] = (-wdy * con_o[2] - wdx * con_o[1]) * ddely_dy;
	cur_dL_dconic2D[0] = -0.5f * wdx * dx;
	cur_dL_dconic2D[1] = -0.5f * wdx * dy;
	cur_dL_dconic2D[3] = -0.5f * wdy * dy;
	v3 mean = { mean3[0], mean3[1], mean3[2] };
	v3 cur_dL_dmeans = { 0.0f, 0.0f, 0.0f };
	real cur_dL_dcov3D[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
	orc_cov2d_backward(mean, 1, cov3D, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, cur_dL_dconic2D, &cur_dL_dmeans, cur_dL_dcov3D);
	const real* proj = projmatrix;
	v4 m_hom = orc_tp4x4(mean, proj);
	real m_w = 1.0f / (m_hom.w + 0.0000001f);
	real mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
	real mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
	cur_dL_dmeans.x += (proj[0] * m_w - proj[3] * mul1) * cur_dL_dmean2D[0] + (proj[1] * m_w - proj[3] * mul2) * cur_dL_dmean2D[1];
	cur_dL_dmeans.y += (proj[4] * m_w - proj[7] * mul1) * cur_dL_dmean2D[0] + (proj[5] * m_w - proj[7] * mul2) * cur_dL_dmean2D[1];
	cur_dL_dmeans.z += (proj[8] * m_w - proj[11] * mul1) * cur_dL_dmean2D[0] + (proj[9] * m_w - proj[11] * mul2) * cur_dL_dmean2D[1];
	v3 cur_dL_dscale = { 0.0f, 0.0f, 0.0f };
	v4 cur_dL_drot = { 0.0f, 0.0f, 0.0f, 0.0f };
	orc_cov3d_backward(scale, mod, rot4, cur_dL_dcov3D, &cur_dL_dscale, &cur_dL_drot);
	out[0] = cur_dL_dmeans.x; out[1] = cur_dL_dmeans.y; out[2] = cur_dL_dmeans.z;
	out[3] = w / con_o[3];
	out[4] = cur_dL_dscale.x; out[5] = cur_dL_dscale.y; out[6] = cur_dL_dscale.z;
	out[7] = cur_dL_drot.x; out[8] = cur_dL_drot.y; out[9] = cur_dL_drot.z; out[10] = cur_dL_drot.w;
}

/* renderCUDAFused, backward.cu:850-1140.
 * Outputs (real, caller allocated, overwritten): dL_dmean2D[P*3] (z stays 0), dL_dconic[P*4],
 * dL_dopacity[P], dL_dcolors[P*3], dL_dmean3D[P*3], dL_dcov3D[P*6], dL_dsh[P*M*3], dL_dscale[P*3],
 * dL_drot[P*4].  pair_count (optional) receives the number of contributing pairs. */
void orc_render_backward_fused(int P, int D, int M, int W, int H,
	const uint32_t* ranges, const uint32_t* point_list,
	const real* bg_color, const real* means2D, const real* conic_opacity, const real* colors,
	const real* final_Ts, const uint32_t* n_contrib, const real* dL_dpixels,
	const real* means3D, const int32_t* radii, const real* shs, const uint8_t* clamped,
	const real* scales, const real* rotations, real scale_modifier, const real* cov3Ds,
	const real* viewmatrix, const real* projmatrix, real tan_fovx, real tan_fovy, const real* campos,
	int grad_power,
	real* dL_dmean2D, real* dL_dconic, real* dL_dopacity, real* dL_dcolors, real* dL_dmean3D,
	real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot, int64_t* pair_count,
	const f32* dec_means2D, const f32* dec_conic_opacity)
{
	/* dec_* (arbiter build only; NULL otherwise): the binary32 run's means2D / conic_opacity, for the per-pair tests */
	const real focal_y = H / (2.0f * tan_fovy);  /* rasterizer_impl.cu:383-384 */
	const real focal_x = W / (2.0f * tan_fovx);
	const uint32_t gx = (W + ORC_BLOCK_X - 1) / ORC_BLOCK_X, gy = (H + ORC_BLOCK_Y - 1) / ORC_BLOCK_Y;
	const int C = ORC_NUM_CHANNELS;
	size_t Ps = (size_t)(P > 0 ? P : 1);
	double* a_m2 = (double*)calloc(Ps * 2, sizeof(double));
	double* a_con = (double*)calloc(Ps * 3, sizeof(double));
	double* a_op = (double*)calloc(Ps, sizeof(double));
	double* a_col = (double*)calloc(Ps * 3, sizeof(double));
	double* a_m3 = (double*)calloc(Ps * 3, sizeof(double));
	double* a_cov = (double*)calloc(Ps * 6, sizeof(double));
	double* a_sh = (double*)calloc(Ps * (size_t)(M > 0 ? M : 1) * 3, sizeof(double));
	double* a_sc = (double*)calloc(Ps * 3, sizeof(double));
	double* a_rot = (double*)calloc(Ps * 4, sizeof(double));
	int64_t pairs = 0;

	const real ddelx_dx = (real)(0.5 * W);
	const real ddely_dy = (real)(0.5 * H);
	const v3 campos_v = { campos ? campos[0] : 0.f, campos ? campos[1] : 0.f, campos ? campos[2] : 0.f };

	for (uint32_t ty = 0; ty < gy; ty++)
	for (uint32_t tx = 0; tx < gx; tx++)
	{
		uint32_t r0 = ranges[2 * (ty * gx + tx)], r1 = ranges[2 * (ty * gx + tx) + 1];
		for (uint32_t ly = 0; ly < ORC_BLOCK_Y; ly++)
		for (uint32_t lx = 0; lx < ORC_BLOCK_X; lx++)
		{
			uint32_t pxx = tx * ORC_BLOCK_X + lx, pxy = ty * ORC_BLOCK_Y + ly;
			if (!(pxx < (uint32_t)W && pxy < (uint32_t)H))
				continue;
			uint32_t pix_id = W * pxy + pxx;
			real pixfx = (real)pxx, pixfy = (real)pxy;

			const real T_final = final_Ts[pix_id];
			real T = T_final;
			uint32_t contributor = r1 - r0;
			const int last_contributor = (int)n_contrib[pix_id];
			real accum_rec[3] = { 0 };
			real dL_dpixel[3];
			for (int i = 0; i < C; i++)
				dL_dpixel[i] = dL_dpixels[i * H * W + pix_id];
			real last_alpha = 0;
			real last_color[3] = { 0 };

			for (uint32_t kk = r1; kk > r0; kk--)
			{
				contributor--;
				if (contributor >= (uint32_t)last_contributor) /* unsigned compare as in the reference */
					continue;
				const int global_id = (int)point_list[kk - 1];
				const real dx = means2D[2 * global_id] - pixfx, dy = means2D[2 * global_id + 1] - pixfy;
				const real* con_o = conic_opacity + 4 * (size_t)global_id;
				const real power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
				if (dec_means2D)
				{
					if (orc_pair_skipped(dec_means2D + 2 * (size_t)global_id, dec_conic_opacity + 4 * (size_t)global_id, (f32)pxx, (f32)pxy))
						continue;
				}
				else if (power > 0.0f)
					continue;
				const real G = orc_exp(power);
				const real alpha = R_MIN(0.99f, con_o[3] * G);
				if (!dec_means2D && alpha < 1.0f / 255.0f)
					continue;

				T = T / (1.f - alpha);
				const real dchannel_dcolor = alpha * T;

				real cur_dL_dcolors[3] = { 0, 0, 0 };
				real cur_dL_dmean2D[3] = { 0, 0, 0 };
				real cur_dL_dconic2D[4] = { 0, 0, 0, 0 };
				real cur_dL_dopacity = 0.0f;

				real dL_dalpha = 0.0f;
				for (int ch = 0; ch < C; ch++)
				{
					const real c = colors[global_id * C + ch];
					accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
					last_color[ch] = c;
					const real dL_dchannel = dL_dpixel[ch];
					dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
					cur_dL_dcolors[ch] = dchannel_dcolor * dL_dchannel;
				}
				dL_dalpha *= T;
				last_alpha = alpha;

				real bg_dot_dpixel = 0;
				for (int i = 0; i < C; i++)
					bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
				dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

				const real dL_dG = con_o[3] * dL_dalpha;
				const real gdx = G * dx;
				const real gdy = G * dy;
				const real dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
				const real dG_ddely = -gdy * con_o[2] - gdx * con_o[1];

				cur_dL_dmean2D[0] = dL_dG * dG_ddelx * ddelx_dx;
				cur_dL_dmean2D[1] = dL_dG * dG_ddely * ddely_dy;

				cur_dL_dconic2D[0] = -0.5f * gdx * dx * dL_dG;
				cur_dL_dconic2D[1] = -0.5f * gdx * dy * dL_dG;
				cur_dL_dconic2D[3] = -0.5f * gdy * dy * dL_dG;
				cur_dL_dopacity = G * dL_dalpha;

				if (global_id >= P)
					continue;

				v3 cur_dL_dmeans = { 0.0f, 0.0f, 0.0f };
				real cur_dL_dcov3D[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
				v3 mean = { means3D[3 * global_id], means3D[3 * global_id + 1], means3D[3 * global_id + 2] };
				orc_cov2d_backward(mean, radii[global_id], cov3Ds + 6 * (size_t)global_id, focal_x, focal_y,
					tan_fovx, tan_fovy, viewmatrix, cur_dL_dconic2D, &cur_dL_dmeans, cur_dL_dcov3D);

				const int num_shs = (1 + D) * (1 + D);
				real cur_dL_dshs[16 * 3];
				memset(cur_dL_dshs, 0, sizeof(cur_dL_dshs));
				v3 cur_dL_dscale = { 0.0f, 0.0f, 0.0f };
				v4 cur_dL_drot = { 0.0f, 0.0f, 0.0f, 0.0f };

				/* preprocessCUDARelocated (backward.cu:532-583), idx == 0 */
				if (radii[global_id] > 0)
				{
					v3 m = mean;
					const real* proj = projmatrix;
					v4 m_hom = orc_tp4x4(m, proj);
					real m_w = 1.0f / (m_hom.w + 0.0000001f);
					real mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
					real mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
					v3 dL_dmean;
					dL_dmean.x = (proj[0] * m_w - proj[3] * mul1) * cur_dL_dmean2D[0] + (proj[1] * m_w - proj[3] * mul2) * cur_dL_dmean2D[1];
					dL_dmean.y = (proj[4] * m_w - proj[7] * mul1) * cur_dL_dmean2D[0] + (proj[5] * m_w - proj[7] * mul2) * cur_dL_dmean2D[1];
					dL_dmean.z = (proj[8] * m_w - proj[11] * mul1) * cur_dL_dmean2D[0] + (proj[9] * m_w - proj[11] * mul2) * cur_dL_dmean2D[1];
					cur_dL_dmeans.x += dL_dmean.x;
					cur_dL_dmeans.y += dL_dmean.y;
					cur_dL_dmeans.z += dL_dmean.z;

					if (shs)
						orc_sh_backward(D, mean, campos_v, shs + (size_t)M * global_id, clamped + 3 * (size_t)global_id,
							cur_dL_dcolors, &cur_dL_dmeans, cur_dL_dshs);
					if (scales)
						orc_cov3d_backward(scales + 3 * (size_t)global_id, scale_modifier, rotations + 4 * (size_t)global_id,
							cur_dL_dcov3D, &cur_dL_dscale, &cur_dL_drot);
				}

				pairs++;
				for (int ch = 0; ch < C; ch++)
					a_col[global_id * 3 + ch] += orc_pow(cur_dL_dcolors[ch], grad_power);
				a_m2[global_id * 2 + 0] += orc_pow(cur_dL_dmean2D[0], grad_power);
				a_m2[global_id * 2 + 1] += orc_pow(cur_dL_dmean2D[1], grad_power);
				a_con[global_id * 3 + 0] += orc_pow(cur_dL_dconic2D[0], grad_power);
				a_con[global_id * 3 + 1] += orc_pow(cur_dL_dconic2D[1], grad_power);
				a_con[global_id * 3 + 2] += orc_pow(cur_dL_dconic2D[3], grad_power);
				a_m3[global_id * 3 + 0] += orc_pow(cur_dL_dmeans.x, grad_power);
				a_m3[global_id * 3 + 1] += orc_pow(cur_dL_dmeans.y, grad_power);
				a_m3[global_id * 3 + 2] += orc_pow(cur_dL_dmeans.z, grad_power);
				for (int ch = 0; ch < 6; ch++)
					a_cov[global_id * 6 + ch] += orc_pow(cur_dL_dcov3D[ch], grad_power);
				if (D > 0)
				{
					for (int ch = 0; ch < num_shs; ch++)
						for (int k = 0; k < 3; k++)
							a_sh[((size_t)global_id * M + ch) * 3 + k] += orc_pow(cur_dL_dshs[3 * ch + k], grad_power);
				}
				a_sc[global_id * 3 + 0] += orc_pow(cur_dL_dscale.x, grad_power);
				a_sc[global_id * 3 + 1] += orc_pow(cur_dL_dscale.y, grad_power);
				a_sc[global_id * 3 + 2] += orc_pow(cur_dL_dscale.z, grad_power);
				a_rot[global_id * 4 + 0] += orc_pow(cur_dL_drot.x, grad_power);
				a_rot[global_id * 4 + 1] += orc_pow(cur_dL_drot.y, grad_power);
				a_rot[global_id * 4 + 2] += orc_pow(cur_dL_drot.z, grad_power);
				a_rot[global_id * 4 + 3] += orc_pow(cur_dL_drot.w, grad_power);
				a_op[global_id] += orc_pow(cur_dL_dopacity, grad_power);
			}
		}
	}

	for (int i = 0; i < P; i++)
	{
		dL_dmean2D[3 * i + 0] = (real)a_m2[2 * i]; dL_dmean2D[3 * i + 1] = (real)a_m2[2 * i + 1]; dL_dmean2D[3 * i + 2] = 0.f;
		dL_dconic[4 * i + 0] = (real)a_con[3 * i]; dL_dconic[4 * i + 1] = (real)a_con[3 * i + 1];
		dL_dconic[4 * i + 2] = 0.f; dL_dconic[4 * i + 3] = (real)a_con[3 * i + 2];
		dL_dopacity[i] = (real)a_op[i];
		for (int k = 0; k < 3; k++) dL_dcolors[3 * i + k] = (real)a_col[3 * i + k];
		for (int k = 0; k < 3; k++) dL_dmean3D[3 * i + k] = (real)a_m3[3 * i + k];
		for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = (real)a_cov[6 * i + k];
		for (int k = 0; k < 3 * M; k++) dL_dsh[(size_t)i * M * 3 + k] = (real)a_sh[(size_t)i * M * 3 + k];
		for (int k = 0; k < 3; k++) dL_dscale[3 * i + k] = (real)a_sc[3 * i + k];
		for (int k = 0; k < 4; k++) dL_drot[4 * i + k] = (real)a_rot[4 * i + k];
	}
	if (pair_count) *pair_count = pairs;
	free(a_m2); free(a_con); free(a_op); free(a_col); free(a_m3); free(a_cov); free(a_sh); free(a_sc); free(a_rot);
}

/* ---------------------------------------------------------------------- */
/* simple-knn distCUDA2 (upstream gitlab.inria.fr/bkerbl/simple-knn, not in the reference tree):
 * mean of the squared distances to the 3 nearest OTHER points (self excluded by index).
 * Brute force O(P^2); parity unpinned (no reference source, call site or fixture). */
/* ---------------------------------------------------------------------- */
#ifndef ORC_DOUBLE
void orc_knn_dist2(int P, const real* pts, real* out)
{
	for (int i = 0; i < P; i++)
	{
		real best[3] = { 3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f };
		real rx = pts[3 * i], ry = pts[3 * i + 1], rz = pts[3 * i + 2];
		for (int j = 0; j < P; j++)
		{
			if (j == i) continue;
			real dx = pts[3 * j] - rx, dy = pts[3 * j + 1] - ry, dz = pts[3 * j + 2] - rz;
			real dist = dx * dx + dy * dy + dz * dz;
			for (int k = 0; k < 3; k++)
				if (best[k] > dist) { real t = best[k]; best[k] = dist; dist = t; }
		}
		out[i] = (best[0] + best[1] + best[2]) / 3.0f;
	}
}
#endif

// tests/harness/fr_math_harness.cpp -- TEST ONLY.
// Compiles the host/device-neutral arithmetic of fisher-nerf-customized_amd/csrc/fr_math.h with g++ so that the
// per-Gaussian maths of the HIP kernels can be compared with the oracle on the CPU, before any GPU time is spent.
// Never loaded by the product.
#include "../../fisher-nerf-customized_amd/csrc/fr_math.h"

extern "C" {

void h_expf(int n, const float* x, float* y)
{
	for (int i = 0; i < n; i++) y[i] = fr_expf(x[i]);
}

void h_cov3d(int P, const float* scales, float mod, const float* rots, float* cov3D)
{
	for (int i = 0; i < P; i++)
	{
		fr_f3 s = { scales[3 * i], scales[3 * i + 1], scales[3 * i + 2] };
		fr_f4 q = { rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3] };
		fr_cov3d(s, mod, q, cov3D + 6 * i);
	}
}

void h_world_to_cam(int P, const float* w2c_rowmajor, const float* pts, float* out)
{
	for (int i = 0; i < P; i++)
	{
		fr_f3 p = { pts[3 * i], pts[3 * i + 1], pts[3 * i + 2] };
		fr_f3 t = fr_world_to_cam(p, w2c_rowmajor);
		out[3 * i] = t.x; out[3 * i + 1] = t.y; out[3 * i + 2] = t.z;
	}
}

void h_preprocess(int P, const float* means, const float* cov3D, const float* view, const float* proj,
                  int W, int H, float tanfovx, float tanfovy,
                  int* radii, float* depths, float* means2D, float* conic, unsigned* tiles, unsigned* rect)
{
	const float focal_y = H / (2.0f * tanfovy);
	const float focal_x = W / (2.0f * tanfovx);
	const uint32_t gx = (W + 15) / 16, gy = (H + 15) / 16;
	for (int i = 0; i < P; i++)
	{
		fr_f3 p = { means[3 * i], means[3 * i + 1], means[3 * i + 2] };
		fr_splat s = fr_preprocess_one(p, cov3D + 6 * i, view, proj, W, H, tanfovx, tanfovy, focal_x, focal_y, gx, gy);
		radii[i] = s.radius; tiles[i] = s.tiles;
		if (s.radius > 0)
		{
			depths[i] = s.depth; means2D[2 * i] = s.px; means2D[2 * i + 1] = s.py;
			conic[3 * i] = s.conx; conic[3 * i + 1] = s.cony; conic[3 * i + 2] = s.conz;
			rect[4 * i] = s.rect.x0; rect[4 * i + 1] = s.rect.y0; rect[4 * i + 2] = s.rect.x1; rect[4 * i + 3] = s.rect.y1;
		}
	}
}

void h_sh_to_rgb(int P, int deg, int M, const float* means, const float* campos, const float* shs, float* rgb, unsigned char* clamped)
{
	fr_f3 cp = { campos[0], campos[1], campos[2] };
	for (int i = 0; i < P; i++)
	{
		fr_f3 p = { means[3 * i], means[3 * i + 1], means[3 * i + 2] };
		fr_f3 c = fr_sh_to_rgb(deg, p, cp, shs + 3 * (size_t)i * M, clamped + 3 * i);
		rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
	}
}

// Leaf gradients of ONE (pixel, Gaussian) pair through the Jacobian matrices the kernels stage in LDS.
// u = (m2x, m2y, cx, cy, cw).  out: mean3D[3], cov3D[6], scale[3], rot[4]
void h_leaf_from_u(const float* mean, const float* cov3D, const float* scale, float mod, const float* rot,
                   const float* view, const float* proj, int W, int H, float tanfovx, float tanfovy,
                   const float* u, float* out)
{
	const float focal_y = H / (2.0f * tanfovy);
	const float focal_x = W / (2.0f * tanfovx);
	fr_f3 m = { mean[0], mean[1], mean[2] };
	float A[3][5]; float B[6][3]; float Cm[7][3];
	fr_mean_jacobian(m, cov3D, view, proj, focal_x, focal_y, tanfovx, tanfovy, A, B);
	fr_f3 s = { scale[0], scale[1], scale[2] };
	fr_f4 q = { rot[0], rot[1], rot[2], rot[3] };
	fr_scale_rot_jacobian(s, mod, q, B, Cm);
	for (int r = 0; r < 3; r++)
		out[r] = A[r][0] * u[0] + A[r][1] * u[1] + A[r][2] * u[2] + A[r][3] * u[3] + A[r][4] * u[4];
	for (int r = 0; r < 6; r++)
		out[3 + r] = B[r][0] * u[2] + B[r][1] * u[3] + B[r][2] * u[4];
	for (int r = 0; r < 7; r++)
		out[9 + r] = Cm[r][0] * u[2] + Cm[r][1] * u[3] + Cm[r][2] * u[4];
}


// The pair factor F = sum_c hv[c] leaf_c^2 / w^2 (w = opacity G dL_dalpha) of one Gaussian at n offsets d = mean2D - pixel, as
// the scorer's records give it: out_g = the record of k_fisher_tile_v3 (polynomial in u = -conic d, fr_scorer_poly_g +
// fr_scorer_poly_eval: the kernel's arithmetic), out_d = the SAME quadratic form expanded into a polynomial in (dx, dy) --
// round 2's record, kept here only to show why it was replaced (condition number squared on needle-shaped splats).
// C = 4 or 11; hv = [mean 3 | opacity | scale 3 | rot 4]; conic3 receives the conic the record was built with.
void h_scorer_pair_factor(int C, const float* mean, const float* cov3D, const float* scale, float mod, const float* rot,
                          const float* view, const float* proj, int W, int H, float tanfovx, float tanfovy,
                          const float* hv, float opacity, int n, const float* dx, const float* dy,
                          float* out_g, float* out_d, float* conic3)
{
	const float focal_y = H / (2.0f * tanfovy);
	const float focal_x = W / (2.0f * tanfovx);
	fr_f3 m = { mean[0], mean[1], mean[2] };
	fr_f3 s = { scale[0], scale[1], scale[2] };
	fr_f4 q4 = { rot[0], rot[1], rot[2], rot[3] };
	float Rg[3][5], Bg[6][3], Cg[7][3], cov2d[3], f;
	fr_mean_rows_g(m, cov3D, view, proj, focal_x, focal_y, tanfovx, tanfovy, W, H, Rg, Bg, cov2d, &f);
	fr_scale_rot_jacobian(s, mod, q4, Bg, Cg);
	float qg[12];
	if (C >= 11) fr_scorer_poly_g<11>(Rg, Cg, hv, qg); else fr_scorer_poly_g<4>(Rg, Cg, hv, qg);
	const float k3 = hv[3] / (opacity * opacity);
	const float ca = cov2d[0] + 0.3f, cb = cov2d[1], cc = cov2d[2] + 0.3f;
	const float det = ca * cc - cb * cb;
	const float hcx = -0.5f * cc / det, ncy = cb / det, hcz = -0.5f * ca / det;
	conic3[0] = -2.0f * hcx; conic3[1] = -ncy; conic3[2] = -2.0f * hcz;
	// round 2's form: rows over u' = (-(cx dx + cy dy), -(cz dy + cy dx), dx^2, dx dy, dy^2), then expanded in (dx, dy)
	float A[3][5], B[6][3], Cm[7][3];
	fr_mean_jacobian(m, cov3D, view, proj, focal_x, focal_y, tanfovx, tanfovy, A, B);
	fr_scale_rot_jacobian(s, mod, q4, B, Cm);
	const float hw = (float)(0.5 * W), hh = (float)(0.5 * H);
	float Ap[3][5], Cp[7][3];
	for (int r = 0; r < 3; r++) { Ap[r][0] = A[r][0] * hw; Ap[r][1] = A[r][1] * hh; for (int c = 2; c < 5; c++) Ap[r][c] = -0.5f * A[r][c]; }
	for (int r = 0; r < 7; r++) for (int c = 0; c < 3; c++) Cp[r][c] = -0.5f * Cm[r][c];
	float qf[15]; int nq = 0;
	for (int i = 0; i < 5; i++)
		for (int j = i; j < 5; j++)
		{
			float acc = hv[0] * Ap[0][i] * Ap[0][j] + hv[1] * Ap[1][i] * Ap[1][j] + hv[2] * Ap[2][i] * Ap[2][j];
			if (C >= 11 && i >= 2) for (int r = 0; r < 7; r++) acc += hv[4 + r] * Cp[r][i - 2] * Cp[r][j - 2];
			qf[nq++] = (i == j) ? acc : 2.0f * acc;
		}
	const float pa0 = 2.0f * hcx, pb0 = ncy, pa1 = ncy, pb1 = 2.0f * hcz;
	float qd[12];
	qd[6] = qf[0] * pa0 * pa0 + qf[1] * pa0 * pa1 + qf[5] * pa1 * pa1;                                        // c20
	qd[3] = 2.0f * qf[0] * pa0 * pb0 + qf[1] * (pa0 * pb1 + pa1 * pb0) + 2.0f * qf[5] * pa1 * pb1;            // c11
	qd[0] = qf[0] * pb0 * pb0 + qf[1] * pb0 * pb1 + qf[5] * pb1 * pb1;                                        // c02
	qd[9] = qf[2] * pa0 + qf[6] * pa1;                                                                        // c30
	qd[7] = qf[2] * pb0 + qf[3] * pa0 + qf[6] * pb1 + qf[7] * pa1;                                            // c21
	qd[4] = qf[3] * pb0 + qf[4] * pa0 + qf[7] * pb1 + qf[8] * pa1;                                            // c12
	qd[1] = qf[4] * pb0 + qf[8] * pb1;                                                                        // c03
	qd[11] = qf[9]; qd[10] = qf[10]; qd[8] = qf[11] + qf[12]; qd[5] = qf[13]; qd[2] = qf[14];                 // c40 c31 c22 c13 c04
	for (int k = 0; k < n; k++)
	{
		const float t = hcx * dx[k] + ncy * dy[k];
		const float v = hcz * dy[k];
		const float ux = hcx * dx[k] + t;               // 2 hcx dx + ncy dy = -gx
		const float uy = 2.0f * v + ncy * dx[k];        // ncy dx + 2 hcz dy = -gy
		out_g[k] = fr_scorer_poly_eval(qg, k3, ux, uy);
		out_d[k] = fr_scorer_poly_eval(qd, k3, dx[k], dy[k]);
	}
}


// Leaves of ONE (pixel, Gaussian) pair for w = 1 through the rows over gamma(u), u = -conic d, that the record kernels store
// (fr_mean_rows_g / fr_scale_rot_jacobian): out[10] = mean3D[3], scale[3], rot[4].  conic3 = the binary32 conic of the forward pass.
void h_leaves_from_rows_g(const float* mean, const float* cov3D, const float* scale, float mod, const float* rot,
                          const float* view, const float* proj, int W, int H, float tanfovx, float tanfovy,
                          const float* conic3, float dx, float dy, float* out)
{
	const float focal_y = H / (2.0f * tanfovy);
	const float focal_x = W / (2.0f * tanfovx);
	fr_f3 m = { mean[0], mean[1], mean[2] };
	fr_f3 s = { scale[0], scale[1], scale[2] };
	fr_f4 q4 = { rot[0], rot[1], rot[2], rot[3] };
	float Rg[3][5], Bg[6][3], Cg[7][3];
	fr_mean_rows_g(m, cov3D, view, proj, focal_x, focal_y, tanfovx, tanfovy, W, H, Rg, Bg, nullptr, nullptr);
	fr_scale_rot_jacobian(s, mod, q4, Bg, Cg);
	const float ux = -(conic3[0] * dx + conic3[1] * dy), uy = -(conic3[1] * dx + conic3[2] * dy);
	const float gm[5] = { ux, uy, ux * ux, ux * uy, uy * uy };
	for (int r = 0; r < 3; r++) out[r] = Rg[r][0] * gm[0] + Rg[r][1] * gm[1] + Rg[r][2] * gm[2] + Rg[r][3] * gm[3] + Rg[r][4] * gm[4];
	for (int r = 0; r < 7; r++) out[3 + r] = Cg[r][0] * gm[2] + Cg[r][1] * gm[3] + Cg[r][2] * gm[4];
}

}

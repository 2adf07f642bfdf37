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

}

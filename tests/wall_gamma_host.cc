// Host build of gpusph_amd/csrc/sa_wall_gamma.h for tests/test_sa_wall_gamma.py (g++, no HIP): the product's formulation of
// gamma / |grad gamma| evaluated on the CPU, so that it can be held against the reference's numbers without a GPU.
#include "../gpusph_amd/csrc/sa_wall_gamma.h"

extern "C" float wg_grad_gamma(const float *ns, const float *vp /* 3 x 2 */, float h, const float *q)
{
	WallTri t;
	const float2 c0 = { vp[0], vp[1] }, c1 = { vp[2], vp[3] }, c2 = { vp[4], vp[5] };
	wall_tri_setup(t, v3(ns[0], ns[1], ns[2]), c0, c1, c2, h);
	return wall_grad_gamma(t, v3(q[0], q[1], q[2]))/h;
}

extern "C" float wg_gamma(int vertex, const float *ns, const float *vp, float h, const float *q, const float *oldGGam, float epsilon)
{
	WallTri t;
	const float2 c0 = { vp[0], vp[1] }, c1 = { vp[2], vp[3] }, c2 = { vp[4], vp[5] };
	wall_tri_setup(t, v3(ns[0], ns[1], ns[2]), c0, c1, c2, h);
	const V3 g = v3(oldGGam[0], oldGGam[1], oldGGam[2]), qq = v3(q[0], q[1], q[2]);
	return vertex ? wall_gamma<true>(t, qq, g, h, epsilon) : wall_gamma<false>(t, qq, g, h, epsilon);
}

extern "C" void wg_corners(const float *ns, const float *vp, float h, float *out9)
{
	WallTri t;
	const float2 c0 = { vp[0], vp[1] }, c1 = { vp[2], vp[3] }, c2 = { vp[4], vp[5] };
	wall_tri_setup(t, v3(ns[0], ns[1], ns[2]), c0, c1, c2, h);
	for (int k = 0; k < 3; ++k) { out9[3*k] = t.corner[k].x; out9[3*k + 1] = t.corner[k].y; out9[3*k + 2] = t.corner[k].z; }
}

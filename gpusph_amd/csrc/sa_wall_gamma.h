// sa_wall_gamma.h -- gamma and |grad gamma| of a triangular boundary element for the Wendland kernel (SA_BOUNDARY).
// Device code of sa_bounds.hip; plain C++ to a host compiler too (tests/wall_gamma_host.cc builds it with g++ and checks it
// against the reference's own numbers, tests/golden/ref_gamma.npz, on the CPU).
#ifndef SPHX_SA_WALL_GAMMA_H
#define SPHX_SA_WALL_GAMMA_H
#ifdef __HIPCC__
#define SPHX_WG_FN __device__ __forceinline__
#define SPHX_WG_FN_NOINLINE __device__
#else
#include <cmath>
struct float2 { float x, y; };
#define SPHX_WG_FN static inline
#define SPHX_WG_FN_NOINLINE static inline
#endif

// ---- the wall renormalisation factor gamma and its gradient for the Wendland kernel ------------------------------------
// What the reference computes (src/cuda/gamma.cuh:90-513; initGammaDevice src/cuda/boundary_conditions_kernel.cu:1891-1970):
//   |grad gamma_as|  the integral of the kernel over the triangular boundary element s as seen from the point a, in closed
//                    form: per edge an antiderivative taken at the edge's two corners (clipped to the kernel's support),
//                    plus the part of the element's plane inside the support that no edge accounts for (angle bookkeeping);
//   gamma_as         the volume integral of the kernel behind the element, by a 7-point quadrature over the triangle of
//                    the kernel integrated along the normal; for a vertex particle sitting on a corner, the solid angle
//                    the walls subtend there.
// This is that mathematics written for this engine: an element's geometry (corners, edge frames) is set up ONCE per
// (particle, element) and reused by every evaluation (the density summation needs |grad gamma_as| at two positions per
// element); the antiderivative is one function of (distance along the edge, distance to that point) with its polynomial
// collected by powers of the distance; elements wholly outside the support are dismissed before any transcendental.
// Lengths are in units of the smoothing length.  Same integrals as the reference, another order of operations: the results
// agree with the oracle's transcription of gamma.cuh (pinned bit for bit to the reference, tests/golden/ref_gamma.npz) to
// the rounding of a few dozen float operations (bounds in tests/test_gpu_sa.py).
#ifndef SPHX_WG_ATAN2
#define SPHX_WG_ATAN2 atan2f      // sa_wall.hip brings its own
#endif
struct V3 { float x, y, z; };
SPHX_WG_FN V3 v3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
SPHX_WG_FN V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
SPHX_WG_FN V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
SPHX_WG_FN V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
SPHX_WG_FN V3 operator*(V3 a, float s) { return v3(a.x*s, a.y*s, a.z*s); }
SPHX_WG_FN V3 operator/(V3 a, float s) { const float inv = 1.0f/s; return a*inv; }
SPHX_WG_FN float dot(V3 a, V3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
SPHX_WG_FN V3 cross(V3 a, V3 b) { return v3(a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); }
SPHX_WG_FN float length(V3 a) { return sqrtf(dot(a, a)); }
SPHX_WG_FN V3 normalize(V3 a) { return a*(1.0f/sqrtf(dot(a, a))); }

// A boundary element as seen from a particle: unit normal, the three corners relative to the element's centre, and for
// every edge e (from corner e+1 to corner e) the unit vector along it and the in-plane unit normal of the edge.
struct WallTri {
	V3 n;
	V3 corner[3];
	V3 along[3], across[3];
	float reach;           // largest distance of a corner from the centre
};

// BUFFER_VERTPOS holds the corners as 2-D coordinates in a basis of the element's plane that the neighbour-list build fixes
// (sa_boundary_niC_vars, src/cuda/buildneibs_kernel.cu:147-190): first axis = n x e_j with e_j the coordinate axis along
// which |n| is smallest (ties to the earlier axis), second axis = n x first.  The corners are MINUS those offsets.
SPHX_WG_FN void wall_plane_basis(V3 n, V3 &u, V3 &v)
{
	const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
	int j = (ax > ay) ? 1 : 0;
	if (((j == 0) ? ax : ay) > az) j = 2;
	const V3 ej = v3(j == 0 ? 1.0f : 0.0f, j == 1 ? 1.0f : 0.0f, j == 2 ? 1.0f : 0.0f);
	u = normalize(cross(n, ej));
	v = cross(n, u);
}
SPHX_WG_FN void wall_tri_setup(WallTri &w, V3 n, float2 c0, float2 c1, float2 c2, float h)
{
	V3 u, v;
	wall_plane_basis(n, u, v);
	const float ih = 1.0f/h;
	w.n = n;
	w.corner[0] = (u*(-c0.x) + v*(-c0.y))*ih;
	w.corner[1] = (u*(-c1.x) + v*(-c1.y))*ih;
	w.corner[2] = (u*(-c2.x) + v*(-c2.y))*ih;
	float reach2 = 0.0f;
#pragma unroll
	for (int e = 0; e < 3; ++e) {
		w.along[e] = normalize(w.corner[e] - w.corner[(e + 1) % 3]);
		w.across[e] = normalize(cross(n, w.along[e]));
		reach2 = fmaxf(reach2, dot(w.corner[e], w.corner[e]));
	}
	w.reach = sqrtf(reach2);
}

// the antiderivative of the edge integral at one end of an edge: s = coordinate of that end along the edge (clipped to the
// support), d = its distance from the particle (at most 2).  a = distance to the element's plane, b = signed in-plane
// distance to the edge's line, c = distance to the edge's line; powers handed over by the caller.
struct EdgeEnd { float angle, poly, log; };
struct EdgePowers { float a, a2, a4, b, b2, b4, c; };
SPHX_WG_FN EdgeEnd wall_edge_end(const EdgePowers &k, float s, float d)
{
	EdgeEnd r;
	const float s2 = s*s;
	// polynomial part, collected as P0 + d P1
	const float p1 = 87.0f*k.a4 + 33.0f*k.b4 + k.a2*(1512.0f + 38.0f*s2) + s2*(8.0f*s2 + 336.0f) + k.b2*(840.0f + 96.0f*k.a2 + 26.0f*s2);
	const float p0 = 1344.0f - 1260.0f*k.a4 - 420.0f*k.b4 - k.a2*(3360.0f + 420.0f*s2) - s2*(84.0f*s2 + 560.0f)
		- k.b2*(1680.0f + 1260.0f*k.a2 + 280.0f*s2);
	r.poly = s*(p0 + d*p1);
	r.angle = SPHX_WG_ATAN2(k.a*s, k.b*d) - SPHX_WG_ATAN2(s, k.b);
	r.log = copysignf(acoshf(fmaxf(d/fmaxf(k.c, 1e-7f), 1.0f)), s);
	return r;
}

// |grad gamma_as| * h at the point q (relative to the element's centre)
// FLAT: the three edges unrolled and the whole inlined into the caller, so that the element lives in registers (the kernels
// that give a wave to every wall particle); otherwise a real function with a loop over the edges, the element in memory (the
// one-thread-per-particle kernels, where it is called from many places).  Same operations in the same order either way.
template<bool FLAT>
SPHX_WG_FN float wall_grad_gamma_body(const WallTri &w, V3 q)
{
	const float pn = dot(w.n, q);
	EdgePowers k;
	k.a = fabsf(pn);
	if (k.a >= 2.0f) return 0.0f;
	if (dot(q, q) >= (2.0f + w.reach)*(2.0f + w.reach)) return 0.0f;     // the whole element lies outside the support
	k.a2 = k.a*k.a; k.a4 = k.a2*k.a2;
	const float a5 = k.a4*k.a;
	float edges = 0.0f, angleAll = 0.0f, angleIn = 0.0f;
	constexpr int unrollEdges = FLAT ? 3 : 1;
#pragma unroll unrollEdges
	for (int e = 0; e < 3; ++e) {
		const V3 d0 = q - w.corner[e], d1 = q - w.corner[(e + 1) % 3];
		k.b = dot(w.across[e], d0);
		k.c = sqrtf(pn*pn + k.b*k.b);
		float s0 = -dot(d0, w.along[e]), s1 = -dot(d1, w.along[e]);
		const float bAbs = fabsf(k.b);
		angleAll += copysignf(SPHX_WG_ATAN2(s1, bAbs) - SPHX_WG_ATAN2(s0, bAbs), k.b);
		if (k.c < 2.0f) {
			const float half = sqrtf(4.0f - k.c*k.c);         // half length of the edge line's chord inside the support
			s0 = copysignf(fminf(fabsf(s0), half), s0);
			s1 = copysignf(fminf(fabsf(s1), half), s1);
			k.b2 = k.b*k.b; k.b4 = k.b2*k.b2;
			const EdgeEnd e0 = wall_edge_end(k, s0, fminf(sqrtf(k.c*k.c + s0*s0), 2.0f));
			const EdgeEnd e1 = wall_edge_end(k, s1, fminf(sqrtf(k.c*k.c + s1*s1), 2.0f));
			const float logw = ((5.0f*k.b2 + 21.0f*(8.0f + k.a2))*k.b2 + 35.0f*k.a2*(16.0f + k.a2))*k.b2 + 35.0f*k.a4*(24.0f + k.a2);
			edges += 0.00015542474911f*(48.0f*a5*(28.0f + k.a2)*(e1.angle - e0.angle)
				+ k.b*((e1.poly - e0.poly) + 3.0f*logw*(e1.log - e0.log)));
			angleIn += copysignf(SPHX_WG_ATAN2(s1, bAbs) - SPHX_WG_ATAN2(s0, bAbs), k.b);
		}
	}
	const float t = 1.0f - 0.5f*k.a;
	const float t2 = t*t;
	return edges + (angleIn - angleAll)*0.05968310365947f*(t2*t2*t)*(2.0f + 5.0f*k.a + 4.0f*k.a2);
}
SPHX_WG_FN_NOINLINE float wall_grad_gamma(const WallTri &w, V3 q) { return wall_grad_gamma_body<false>(w, q); }
SPHX_WG_FN float wall_grad_gamma_flat(const WallTri &w, V3 q) { return wall_grad_gamma_body<true>(w, q); }

// the Wendland kernel integrated along a ray from distance d to the edge of the support (times h^2)
SPHX_WG_FN float wall_kernel_along_ray(float d)
{
	if (!(d < 2.0f)) return 0.0f;
	const float t = 1.0f - 0.5f*d, t2 = t*t, u = 1.0f/d;
	return 0.009947183943243458485555235210782147627153727858778528046729f*(t2*t2*t)*(((8.0f*u + 20.0f)*u + 30.0f)*u + 21.0f);
}

// gamma_as: seven-point rule over the triangle (degree 5: centroid, three points towards the corners, three towards the edge
// midpoints).  The centroid carries TWICE its weight: the reference's loop tests its exit after accumulating
// (gamma.cuh:150-158), and what the reference computes is what its problems were validated with.
template<bool VERTEX>
SPHX_WG_FN float wall_gamma_body(const WallTri &w, V3 q, V3 oldGradGamma, float h, float epsilon)
{
	const float pn = dot(w.n, q);
	const float dist = fminf(fabsf(pn), 2.0f);
	float solid = 0.0f;
	if (VERTEX && dist < epsilon) {
		// is the particle ON one of the corners?  barycentric coordinates of its projection
		const V3 e1 = w.corner[1] - w.corner[0], e2 = w.corner[2] - w.corner[0], d = q - w.corner[0];
		const float g11 = dot(e1, e1), g12 = dot(e1, e2), g22 = dot(e2, e2), r1 = dot(e1, d), r2 = dot(e2, d);
		const float idet = 1.0f/(g12*g12 - g11*g22);
		const float l1 = (g12*r2 - g22*r1)*idet, l2 = (g12*r1 - g11*r2)*idet;
		const bool z1 = fabsf(l1) < epsilon, z2 = fabsf(l2) < epsilon;
		const int at = (z1 && z2) ? 0 : (fabsf(l1 - 1.0f) < epsilon && z2) ? 1 : (fabsf(l2 - 1.0f) < epsilon && z1) ? 2 : -1;
		if (at >= 0) {
			// solid angle of the wedge between the two edges leaving that corner and the direction into the fluid
			// (Van Oosterom & Strackee), as a fraction of the full sphere
			const V3 in = (-oldGradGamma)/fmaxf(length(oldGradGamma), h*1e-3f);
			const V3 c = w.corner[at];
			const V3 f1 = w.corner[(at + 1) % 3] - c, f2 = w.corner[(at + 2) % 3] - c;
			const float i1 = 1.0f/length(f1), i2 = 1.0f/length(f2);
			const float den = 1.0f + dot(f1, in)*i1 + dot(f2, in)*i2 + dot(f1, f2)*i1*i2;
			const float num = dot(in, cross(f1, f2))*i1*i2;
			solid = fabsf(2.0f*SPHX_WG_ATAN2(num, den))*0.079577471545947667884441881686257181017229822870228224373833f;
		}
	}
	float vol = 0.0f;
	if (dist < 2.0f && dist > epsilon) {
		const float wc = 0.225f, wa = 0.132394152788506f, wb = 0.125939180544827f;
		const float a0 = 0.059715871789770f, a1 = 0.470142064105115f, b0 = 0.797426985353087f, b1 = 0.101286507323456f;
		const V3 c0 = w.corner[0], c1 = w.corner[1], c2 = w.corner[2];
		// the sample point of the reference is centre - (m0 c0 + m1 c1 + m2 c2), the rule's point mirrored through the centroid
		// (gaussQuadratureO5 is handed the NEGATED corners, gamma.cuh:427,503): kept, for the reason given above
		auto sample = [&](float m0, float m1, float m2) { return wall_kernel_along_ray(length((c0*m0 + c1*m1 + c2*m2) + q)); };
		const float third = 0.333333333333333f;
		float sum = 2.0f*wc*sample(third, third, third);
		sum += wa*(sample(a0, a1, a1) + sample(a1, a0, a1) + sample(a1, a1, a0));
		sum += wb*(sample(b0, b1, b1) + sample(b1, b0, b1) + sample(b1, b1, b0));
		const float area = 0.5f*length(cross(c1 - c0, c2 - c0));
		vol = sum*area*(pn*dot(w.n, w.n));          // n.(n (n.q)), the normal as stored (unit to rounding)
	}
	return solid + vol;
}
template<bool VERTEX>
SPHX_WG_FN_NOINLINE float wall_gamma(const WallTri &w, V3 q, V3 oldGradGamma, float h, float epsilon)
{ return wall_gamma_body<VERTEX>(w, q, oldGradGamma, h, epsilon); }
template<bool VERTEX>
SPHX_WG_FN float wall_gamma_flat(const WallTri &w, V3 q, V3 oldGradGamma, float h, float epsilon)
{ return wall_gamma_body<VERTEX>(w, q, oldGradGamma, h, epsilon); }

#endif

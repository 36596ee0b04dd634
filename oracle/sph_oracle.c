/*
 * sph_oracle.c -- CPU oracle for the WCSPH timestep hot path (TEST INFRASTRUCTURE ONLY).
 * See sph_oracle.h for scope, pinning status and floating-point conventions.
 * Every function cites the reference source (relative to /root/reference) it restates.
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp)
 */
#include "sph_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- data-model constants --------------------------------------------------------- */
/* src/multi_gpu_defines.h:56-83, src/hashkey.h:44-47, src/common_types.h:57-72 */
#define CELLTYPE_BITMASK   (~(3u << 30))
#define CELL_HASH_MAX      UINT_MAX
#define EMPTY_SEGMENT      UINT_MAX
#define CELL_EMPTY         UINT_MAX
#define CELLNUM_SHIFT      11
#define CELLNUM_ENCODED    (1u << CELLNUM_SHIFT)
#define NEIBINDEX_MASK     (CELLNUM_ENCODED - 1)
#define ENCODE_CELL(cell)  (((cell) + 1) << CELLNUM_SHIFT)
#define DECODE_CELL(data)  (((data) >> CELLNUM_SHIFT) - 1)
#define NEIBS_END          USHRT_MAX
/* src/particleinfo.h:144-300 */
enum { PT_FLUID = 0, PT_BOUNDARY, PT_VERTEX, PT_TESTPOINT, PT_NONE };
#define PART_FLAG_START      (1 << 3)
#define FG_COMPUTE_FORCE     (PART_FLAG_START << 0)
#define FG_MOVING_BOUNDARY   (PART_FLAG_START << 1)
#define FG_SURFACE           (PART_FLAG_START << 6)
#define FG_INTERFACE         (PART_FLAG_START << 7)
#define PART_TYPE(f)         ((f).x & 7)
#define FLUID(f)             (PART_TYPE(f) == PT_FLUID)
#define BOUNDARY(f)          (PART_TYPE(f) == PT_BOUNDARY)
#define VERTEX(f)            (PART_TYPE(f) == PT_VERTEX)
#define IO_BOUNDARY_X(f)     ((f).x & ((PART_FLAG_START << 2) | (PART_FLAG_START << 3)))   /* IO_BOUNDARY, defined with its section below */
#define TESTPOINT(f)         (PART_TYPE(f) == PT_TESTPOINT)
#define MOVING(f)            ((f).x & FG_MOVING_BOUNDARY)
#define FLOATING(f)          ((f).x & (FG_MOVING_BOUNDARY | FG_COMPUTE_FORCE))
#define COMPUTE_FORCE(f)     ((f).x & FG_COMPUTE_FORCE)
#define SURFACE(f)           ((f).x & FG_SURFACE)
#define ACTIVE(p)            (isfinite((p).w))
#define INACTIVE(p)          (!ACTIVE(p))
#define FLUID_NUM(f)         ((f).y >> 12)
#define OBJECT_NUM(f)        ((f).y & 0xfff)

#define BLOCK_SIZE_FORCES 128   /* src/cuda/forces.cu (BLOCK_SIZE_FORCES) */
#define BLOCK_SIZE_FMAX   256

/* src/particleinfo.h:437-441 */
uint32_t orc_info_id(orc_info i) { return (uint32_t)i.z | ((uint32_t)i.w << 16); }
int orc_info_type(orc_info i) { return PART_TYPE(i); }

/* timing runs set this to the CPUs the process may really use (cgroup quota), tests leave the OpenMP default */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
#else
	(void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* ---- kernel functions: src/cuda/sph_core.cu:66-191 -------------------------------- */

/* coefficients: src/cuda/forces.cu:274-309 (computed in double, stored as float) */
float orc_wcoeff(int kerneltype, float slength, float kernelradius)
{
	const float h = slength;
	const float h2 = h*h;
	const float h3 = h2*h;
	switch (kerneltype) {
	case ORC_CUBICSPLINE: return (float)(1.0f/(M_PI*h3));
	case ORC_QUADRATIC:   return (float)(15.0f/(16.0f*M_PI*h3));
	case ORC_WENDLAND:    return (float)(21.0f/(16.0f*M_PI*h3));
	case ORC_GAUSSIAN: {
		const float R = kernelradius;
		const float R2 = R*R;
		const float exp_R2 = exp(-R2);
		float kc = -2*exp_R2/3 * h3 * M_PI * R*(3+2*R2)
			+ h3 * 5.5683279968317078452848179821188357020136243902832439 * erf(R);
		kc = 1/kc;
		return kc;
	}
	}
	return NAN;
}

float orc_fcoeff(int kerneltype, float slength, float kernelradius)
{
	const float h = slength;
	const float h2 = h*h;
	const float h4 = h2*h2;
	const float h5 = h4*h;
	switch (kerneltype) {
	case ORC_CUBICSPLINE: return (float)(3.0f/(4.0f*M_PI*h4));
	case ORC_QUADRATIC:   return (float)(15.0f/(32.0f*M_PI*h4));
	case ORC_WENDLAND:    return (float)(105.0f/(128.0f*M_PI*h5));
	case ORC_GAUSSIAN: {
		float kc = orc_wcoeff(ORC_GAUSSIAN, slength, kernelradius);
		kc *= 2/h2;
		return kc;
	}
	}
	return NAN;
}

/* kernel value with an explicit coefficient (coefficient lives in __constant__ in the reference) */
static float W_c(int kerneltype, float r, float slength, float coeff, float wsub_gaussian)
{
	const float R = r/slength;
	float val;
	switch (kerneltype) {
	case ORC_CUBICSPLINE: /* sph_core.cu:72-84 */
		if (R < 1)
			val = 1.0f - 1.5f*R*R + 0.75f*R*R*R;
		else
			val = 0.25f*(2.0f - R)*(2.0f - R)*(2.0f - R);
		return val*coeff;
	case ORC_QUADRATIC: /* sph_core.cu:90-99 */
		val = 0.25f*R*R - R + 1.0f;
		return val*coeff;
	case ORC_WENDLAND: /* sph_core.cu:105-118 */
		val = 1.0f - 0.5f*R;
		val *= val;
		val *= val;
		val *= 1.0f + 2.0f*R;
		return val*coeff;
	case ORC_GAUSSIAN: /* sph_core.cu:128-137 */
		val = expf(-R*R);
		val -= wsub_gaussian;
		return val*coeff;
	}
	return NAN;
}

static float F_c(int kerneltype, float r, float slength, float coeff)
{
	const float R = r/slength;
	float val;
	switch (kerneltype) {
	case ORC_CUBICSPLINE: /* sph_core.cu:146-158 */
		if (R < 1.0f)
			val = (-4.0f + 3.0f*R)/slength;
		else
			val = -(-2.0f + R)*(-2.0f + R)/r;
		return val*coeff;
	case ORC_QUADRATIC: /* sph_core.cu:162-170 */
		val = (-2.0f + R)/r;
		return val*coeff;
	case ORC_WENDLAND: { /* sph_core.cu:174-181 */
		const float qm2 = r/slength - 2.0f;
		return qm2*qm2*qm2*coeff;
	}
	case ORC_GAUSSIAN: /* sph_core.cu:185-191 */
		return -expf(-R*R)*coeff;
	}
	return NAN;
}

float orc_W(int kerneltype, float r, float slength)
{
	const float kr = (kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	return W_c(kerneltype, r, slength, orc_wcoeff(kerneltype, slength, kr), expf(-kr*kr));
}
float orc_F(int kerneltype, float r, float slength)
{
	const float kr = (kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	return F_c(kerneltype, r, slength, orc_fcoeff(kerneltype, slength, kr));
}

/* ---- equation of state: src/cuda/phys_core.cu:99-142 (__powf -> powf) --------------- */
float orc_P(const orc_params *p, float rho_tilde, int i)
{
	const float rho_ratio = rho_tilde + 1.0f;
	return p->bcoeff[i]*(powf(rho_ratio, p->gammacoeff[i]) - 1.0f);
}
float orc_soundSpeed(const orc_params *p, float rho_tilde, int i)
{
	const float rho_ratio = rho_tilde + 1.0f;
	return p->sscoeff[i]*powf(rho_ratio, p->sspowercoeff[i]);
}
static inline float physical_density(const orc_params *p, float rho_tilde, int i)
{
	return (rho_tilde + 1.0f)*p->rho0[i];
}
static inline float numerical_density(const orc_params *p, float rho, int i) { return rho/p->rho0[i] - 1.0f; }
/* exported for the pin against src/vector_math.h:1093-1097 (tests/test_oracle_pinned.py) */
void orc_f4_div(const float v[4], float s, float out[4])
{
	const float inv = 1.0f/s;     /* float4 / float, as every use in this file spells it out */
	out[0] = v[0]*inv; out[1] = v[1]*inv; out[2] = v[2]*inv; out[3] = v[3]*inv;
}

/* ---- cell grid: src/cuda/cellgrid.cuh:98-127 ------------------------------------- */
uint32_t orc_calc_grid_hash(const orc_params *p, int gx, int gy, int gz)
{
	const int g[3] = { gx, gy, gz };
	const int c1 = p->coord[0], c2 = p->coord[1], c3 = p->coord[2];
	return (uint32_t)((g[c3]*(int)p->gridSize[c2])*(int)p->gridSize[c1]
		+ g[c2]*(int)p->gridSize[c1] + g[c1]);
}

void orc_grid_pos_from_hash(const orc_params *p, uint32_t cellHash, int g[3])
{
	const int c1 = p->coord[0], c2 = p->coord[1], c3 = p->coord[2];
	int temp = (int)(p->gridSize[c2]*p->gridSize[c1]);
	g[c3] = cellHash / temp;
	temp = cellHash - g[c3]*temp;
	g[c2] = temp / (int)p->gridSize[c1];
	g[c1] = temp - g[c2]*(int)p->gridSize[c1];
}

/* calcGridHashPeriodic, cellgrid.cuh:177-187 */
static uint32_t calc_grid_hash_periodic(const orc_params *p, int gx, int gy, int gz)
{
	if (gx < 0) gx = p->gridSize[0] - 1;
	if (gx >= (int)p->gridSize[0]) gx = 0;
	if (gy < 0) gy = p->gridSize[1] - 1;
	if (gy >= (int)p->gridSize[1]) gy = 0;
	if (gz < 0) gz = p->gridSize[2] - 1;
	if (gz >= (int)p->gridSize[2]) gz = 0;
	return orc_calc_grid_hash(p, gx, gy, gz);
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* clampGridPos<periodicbound>: src/cuda/buildneibs_kernel.cu:225-298 */
static void clamp_grid_pos(const orc_params *p, const int gridPos[3], int gridOffset[3],
	int newGridPos[3], int *toofar)
{
	for (int a = 0; a < 3; ++a) {
		const int gs = (int)p->gridSize[a];
		newGridPos[a] = gridPos[a] + gridOffset[a];
		if (p->periodic & (1u << a)) {
			if (newGridPos[a] < 0) newGridPos[a] += gs;
			if (newGridPos[a] >= gs) newGridPos[a] -= gs;
		} else {
			newGridPos[a] = imin(imax(0, newGridPos[a]), gs - 1);
			if (abs(gridOffset[a]) > 1 && newGridPos[a] == gridPos[a])
				*toofar = 1;
			gridOffset[a] = newGridPos[a] - gridPos[a];
		}
	}
}

/* ---- calcHashDevice: src/cuda/buildneibs_kernel.cu:659-776 -------------------------- */
void orc_calc_hash(const orc_params *p, orc_f4 *posArray, uint32_t *particleHash, uint32_t *particleIndex,
	const orc_info *particleInfo, const uint32_t *compactDeviceMap, uint32_t numParticles)
{
#pragma omp parallel for schedule(static)
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = particleInfo[index];
		uint32_t gridHash = particleHash[index] & CELLTYPE_BITMASK;

		if (FLUID(info) || MOVING(info) || (SURFACE(info) && !FLUID(info))) {
			orc_f4 pos = posArray[index];
			int gridPos[3];
			orc_grid_pos_from_hash(p, gridHash, gridPos);

			const float pc[3] = { pos.x, pos.y, pos.z };
			int gridOffset[3];
			for (int a = 0; a < 3; ++a) {
				const float half_check = pc[a] < 0 ? 0.5f : 0.49999997f;
				gridOffset[a] = (int)floorf(pc[a]/p->cellSize[a] + half_check);
			}

			int toofar = 0;
			int newGridPos[3];
			clamp_grid_pos(p, gridPos, gridOffset, newGridPos, &toofar);
			gridHash = orc_calc_grid_hash(p, newGridPos[0], newGridPos[1], newGridPos[2]);

			/* as_float3(pos) -= gridOffset*d_cellSize : contracted a - b*c */
			pos.x = fmaf(-(float)gridOffset[0], p->cellSize[0], pos.x);
			pos.y = fmaf(-(float)gridOffset[1], p->cellSize[1], pos.y);
			pos.z = fmaf(-(float)gridOffset[2], p->cellSize[2], pos.z);

			if (toofar)
				pos.w = NAN; /* disable_particle */

			if (INACTIVE(pos))
				gridHash = CELL_HASH_MAX;

			posArray[index] = pos;
		}

		if (compactDeviceMap && gridHash != CELL_HASH_MAX)
			gridHash |= compactDeviceMap[gridHash];

		particleHash[index] = gridHash;
		particleIndex[index] = index;
	}
}

/* fixHashDevice: src/cuda/buildneibs_kernel.cu:786-814 */
void orc_fix_hash(const orc_params *p, uint32_t *particleHash, uint32_t *particleIndex,
	const orc_info *info, const uint32_t *compactDeviceMap, uint32_t numParticles)
{
	(void)p; (void)info;
	for (uint32_t index = 0; index < numParticles; ++index) {
		if (particleHash) {
			const uint32_t gridHash = particleHash[index] & CELLTYPE_BITMASK;
			if (compactDeviceMap)
				particleHash[index] = particleHash[index] | compactDeviceMap[gridHash];
		}
		particleIndex[index] = index;
	}
}

/* ---- sort: ptype_hash_compare, src/cuda/buildneibs.cu:358-412 ---------------------- */
typedef struct { uint32_t hash; orc_info info; uint32_t idx; } sort_rec;

static int sort_cmp(const void *pa, const void *pb)
{
	const sort_rec *a = (const sort_rec*)pa, *b = (const sort_rec*)pb;
	const uint32_t ha = a->hash, hb = b->hash; /* preserveHighbits = true */
	if (ha == hb) {
		const int pta = PART_TYPE(a->info), ptb = PART_TYPE(b->info);
		if (pta == ptb) {
			const uint32_t ia = orc_info_id(a->info), ib = orc_info_id(b->info);
			return ia < ib ? -1 : (ia > ib ? 1 : 0);
		}
		return pta < ptb ? -1 : 1;
	}
	return ha < hb ? -1 : 1;
}

void orc_sort(uint32_t *hash, orc_info *info, uint32_t *partIndex, uint32_t numParticles)
{
	if (!numParticles) return;
	sort_rec *r = (sort_rec*)malloc(sizeof(sort_rec)*(size_t)numParticles);
	for (uint32_t i = 0; i < numParticles; ++i) {
		r[i].hash = hash[i]; r[i].info = info[i]; r[i].idx = partIndex[i];
	}
	qsort(r, numParticles, sizeof(sort_rec), sort_cmp);
	for (uint32_t i = 0; i < numParticles; ++i) {
		hash[i] = r[i].hash; info[i] = r[i].info; partIndex[i] = r[i].idx;
	}
	free(r);
}

/* ---- reorderDataAndFindCellStartDevice: src/cuda/buildneibs_kernel.cu:836-992 ------- */
void orc_reorder(const orc_params *p, uint32_t *cellStart, uint32_t *cellEnd, uint32_t *segmentStart,
	orc_f4 *sortedPos, orc_f4 *sortedVel,
	const orc_f4 *unsortedPos, const orc_f4 *unsortedVel,
	const orc_info *sortedInfo, const uint32_t *particleHash, const uint32_t *particleIndex,
	uint32_t numParticles, uint32_t *newNumParticles)
{
	(void)p; (void)sortedInfo;
	if (segmentStart)
		for (int i = 0; i < 4; ++i) segmentStart[i] = EMPTY_SEGMENT;

	for (uint32_t index = 0; index < numParticles; ++index) {
		const uint32_t cellHash = particleHash[index];
		const uint32_t prevHash = index > 0 ? particleHash[index - 1] : 0;

		if (index == 0 || cellHash != prevHash) {
			if (cellHash != CELL_HASH_MAX)
				cellStart[cellHash & CELLTYPE_BITMASK] = index;
			else
				*newNumParticles = index;
			if (index > 0)
				cellEnd[prevHash & CELLTYPE_BITMASK] = index;
		}

		if (cellHash == CELL_HASH_MAX)
			continue;

		if (index == numParticles - 1) {
			cellEnd[cellHash & CELLTYPE_BITMASK] = index + 1;
			*newNumParticles = numParticles;
		}

		if (segmentStart) {
			const uint8_t curr_type = cellHash >> 30;
			const uint8_t prev_type = prevHash >> 30;
			if (index == 0 || curr_type != prev_type)
				segmentStart[curr_type] = index;
		}

		const uint32_t sortedIndex = particleIndex[index];
		sortedPos[index] = unsortedPos[sortedIndex];
		sortedVel[index] = unsortedVel[sortedIndex];
	}
}

/* cell ranges of an already sorted sub-range (imported halo cells).  The reference derives them on
 * the host from the neighbour device's cell starts (src/GPUWorker.cc:754-776,1391-1430); the result
 * is what the adjacent-hash scan of reorderDataAndFindCellStartDevice gives on that range. */
void orc_find_cell_start(uint32_t *cellStart, uint32_t *cellEnd, const uint32_t *particleHash,
	uint32_t from, uint32_t to)
{
	for (uint32_t index = from; index < to; ++index) {
		const uint32_t cellHash = particleHash[index];
		if (cellHash == CELL_HASH_MAX) continue;
		if (index == from || cellHash != particleHash[index - 1]) {
			cellStart[cellHash & CELLTYPE_BITMASK] = index;
			if (index > from && particleHash[index - 1] != CELL_HASH_MAX)
				cellEnd[particleHash[index - 1] & CELLTYPE_BITMASK] = index;
		}
		if (index == to - 1)
			cellEnd[cellHash & CELLTYPE_BITMASK] = index + 1;
	}
}

/* ---- neighbour list: src/cuda/buildneibs_kernel.cu:300-644,1019-1185 ---------------- */

/* calcNeibCell<periodicbound>, :311-383 */
static int calc_neib_cell(const orc_params *p, int g[3], const int off[3])
{
	for (int a = 0; a < 3; ++a) {
		const int gs = (int)p->gridSize[a];
		g[a] += off[a];
		if (g[a] < 0) {
			if (p->periodic & (1u << a)) g[a] = gs - 1; else return 0;
		} else if (g[a] >= gs) {
			if (p->periodic & (1u << a)) g[a] = 0; else return 0;
		}
	}
	return 1;
}

/* neibListOffset, :464-470 */
static inline uint32_t neib_list_offset(const orc_params *p, uint32_t neib_num, int neib_type)
{
	return (neib_type == PT_FLUID) ? neib_num :
		(neib_type == PT_BOUNDARY) ? p->neibboundpos - neib_num :
		neib_num + p->neibboundpos + 1;
}

/* too_many_neibs, :489-509 */
static inline int too_many_neibs(const orc_params *p, const uint32_t *neibs_num, int neib_type)
{
	switch (neib_type) {
	case PT_FLUID:    return !(neibs_num[PT_FLUID] < p->neibboundpos);
	case PT_BOUNDARY: return !(neibs_num[PT_FLUID] + neibs_num[PT_BOUNDARY] < p->neibboundpos);
	case PT_VERTEX:   return !(neibs_num[PT_VERTEX] < p->neiblistsize - p->neibboundpos - 1);
	default: return 1;
	}
}

static inline float sqlength3(float x, float y, float z)
{
	return fmaf(z, z, fmaf(y, y, x*x));
}

/* what neibsInCell needs of a segment with SA boundaries: sa_boundary_niC_vars, :147-190 */
typedef struct {
	float boundNlSqInflRad;
	const uint32_t *vertices;        /* vertexinfo = uint4, src/particleinfo.h:97 */
	float coord1[3], coord2[3];
	float *vertPos[3];               /* float2 arrays */
} sa_nic;

static void sa_nic_coords(sa_nic *sa, const orc_f4 *boundElement)
{
	const float bx = boundElement->x, by = boundElement->y, bz = boundElement->z;
	const int j = (fabsf(bz) < fabsf(by) && fabsf(bz) < fabsf(bx)) ? 2 : (fabsf(by) < fabsf(bx) ? 1 : 0);
	/* 0 -> (0, z, -y); 1 -> (-z, 0, x); 2 -> (y, -x, 0) */
	const float c[3] = {
		-((j == 1)*bz) + (j == 2)*by,
		(j == 0)*bz - ((j == 2)*bx),
		-((j == 0)*by) + (j == 1)*bx };
	/* normalize(float4): v*rsqrtf(sqlength(v)), src/vector_math.h:1204-1208 (w = 0) */
	const float inv = 1.0f/sqrtf(fmaf(0.0f, 0.0f, fmaf(c[2], c[2], fmaf(c[1], c[1], c[0]*c[0]))));
	for (int a = 0; a < 3; ++a) sa->coord1[a] = c[a]*inv;
	/* cross3(boundElement, coord1), src/vector_math.h:1177-1180: a.y*b.z - a.z*b.y contracts to fma(a.y, b.z, -(a.z*b.y)) */
	sa->coord2[0] = fmaf(by, sa->coord1[2], -(bz*sa->coord1[1]));
	sa->coord2[1] = fmaf(bz, sa->coord1[0], -(bx*sa->coord1[2]));
	sa->coord2[2] = fmaf(bx, sa->coord1[1], -(by*sa->coord1[0]));
}

/* neibsInCell, :536-644 */
static void neibs_in_cell(const orc_params *p, uint16_t *neibsList,
	const orc_f4 *posArray, const orc_info *infoArray,
	const uint32_t *cellStart, const uint32_t *cellEnd,
	const int gridPos_in[3], const int gridOffset[3], unsigned cell,
	uint32_t index, const float pos_in[3], uint32_t *neibs_num, int boundary, float sqinfluenceradius,
	const sa_nic *sa)
{
	int gridPos[3] = { gridPos_in[0], gridPos_in[1], gridPos_in[2] };
	if (!calc_neib_cell(p, gridPos, gridOffset))
		return;

	const uint32_t gridHash = orc_calc_grid_hash(p, gridPos[0], gridPos[1], gridPos[2]);
	const uint32_t bucketStart = cellStart[gridHash];
	const uint32_t bucketEnd = cellEnd[gridHash];
	if (bucketStart == CELL_EMPTY)
		return;

	/* pos -= gridOffset*d_cellSize (:564) : contracted a - b*c */
	float pos[3];
	for (int a = 0; a < 3; ++a)
		pos[a] = fmaf(-(float)gridOffset[a], p->cellSize[a], pos_in[a]);

	int encode_cell = 1;
	int neib_type = PT_FLUID;
	for (uint32_t neib_index = bucketStart; neib_index < bucketEnd; ++neib_index) {
		if (neib_index == index)
			continue;
		const orc_info neib_info = infoArray[neib_index];
		if (TESTPOINT(neib_info))
			continue;
		if (!encode_cell && neib_type != PART_TYPE(neib_info))
			encode_cell = 1;
		neib_type = PART_TYPE(neib_info);

		/* ViscSpec::rheologytype != GRANULAR always here */
		if (p->boundarytype == ORC_LJ_BOUNDARY && boundary && BOUNDARY(neib_info))
			continue;   /* src/cuda/buildneibs_kernel.cu:588-593: LJ only -- the boundary particles of MK_BOUNDARY bodies keep theirs */
		if (p->boundarytype == ORC_DYN_BOUNDARY && p->sph_formulation != ORC_SPH_GRENIER) {   /* :598 */
			if (boundary && BOUNDARY(neib_info))
				continue;
		}

		const orc_f4 neib_pos = posArray[neib_index];
		if (INACTIVE(neib_pos))
			continue;

		const float rx = pos[0] - neib_pos.x, ry = pos[1] - neib_pos.y, rz = pos[2] - neib_pos.z;
		const float rp2 = sqlength3(rx, ry, rz);
		/* isCloseEnough, :389-408: with SA boundaries, boundary neighbours a little farther out are kept */
		const int close_enough = (rp2 < sqinfluenceradius) ||
			(p->boundarytype == ORC_SA_BOUNDARY && sa && rp2 < sa->boundNlSqInflRad && BOUNDARY(neib_info));

		if (close_enough) {
			const uint32_t offset = neib_list_offset(p, neibs_num[neib_type], neib_type);
			neibs_num[neib_type]++;
			if (!too_many_neibs(p, neibs_num, neib_type)) {
				const int neib_bucket_offset = neib_index - bucketStart;
				const int encode_offset = encode_cell ? ENCODE_CELL(cell) : 0;
				neibsList[(size_t)offset*p->neiblist_stride + index] =
					(uint16_t)(neib_bucket_offset + encode_offset);
				encode_cell = 0;
			}
		}
		/* process_niC_segment<SA_BOUNDARY>, :433-463: projected position of the segment's own vertices */
		if (boundary && sa && sa->vertices) {
			const uint32_t nid = orc_info_id(neib_info);
			const uint32_t *v = sa->vertices + 4*(size_t)index;
			const int i = (nid == v[0]) ? 0 : (nid == v[1]) ? 1 : (nid == v[2]) ? 2 : -1;
			if (i > -1) {
				sa->vertPos[i][2*(size_t)index]     = fmaf(rz, sa->coord1[2], fmaf(ry, sa->coord1[1], rx*sa->coord1[0]));
				sa->vertPos[i][2*(size_t)index + 1] = fmaf(rz, sa->coord2[2], fmaf(ry, sa->coord2[1], rx*sa->coord2[0]));
			}
		}
	}
}

void orc_build_neibs(const orc_params *p, uint16_t *neibsList,
	const orc_f4 *posArray, const orc_info *infoArray, const uint32_t *particleHash,
	const uint32_t *cellStart, const uint32_t *cellEnd,
	uint32_t numParticles, uint32_t particleRangeEnd, float sqinfluenceradius,
	orc_neibs_info *out)
{
	orc_build_neibs_sa(p, neibsList, NULL, NULL, NULL, posArray, infoArray, NULL, NULL, particleHash, cellStart, cellEnd,
		numParticles, particleRangeEnd, sqinfluenceradius, sqinfluenceradius, out);
}

/* buildNeibsListDevice with the SA_BOUNDARY members of buildneibs_params (src/cuda/buildneibs_params.h:66-115):
 * vertices / boundElements of the segments (read), vertPos0..2 (written for segments), boundNlSqInflRad */
void orc_build_neibs_sa(const orc_params *p, uint16_t *neibsList,
	float *vertPos0, float *vertPos1, float *vertPos2,
	const orc_f4 *posArray, const orc_info *infoArray,
	const uint32_t *vertices, const orc_f4 *boundElements, const uint32_t *particleHash,
	const uint32_t *cellStart, const uint32_t *cellEnd,
	uint32_t numParticles, uint32_t particleRangeEnd, float sqinfluenceradius, float boundNlSqInflRad,
	orc_neibs_info *out)
{
	(void)numParticles;
	long long numInteractions = 0;
	int maxNeibs = 0, maxVertexNeibs = 0;
	int hasTooMany = -1;
	int hasMax[3] = {0, 0, 0};

#pragma omp parallel for schedule(dynamic, 1024) reduction(+:numInteractions) reduction(max:maxNeibs) reduction(max:maxVertexNeibs)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		uint32_t neibs_num[PT_TESTPOINT] = {0, 0, 0};
		do {
			const orc_info info = infoArray[index];
			sa_nic sa;
			const int with_sa = p->boundarytype == ORC_SA_BOUNDARY && vertices && boundElements;
			if (with_sa) {
				sa.boundNlSqInflRad = boundNlSqInflRad;
				sa.vertices = vertices;
				sa.vertPos[0] = vertPos0; sa.vertPos[1] = vertPos1; sa.vertPos[2] = vertPos2;
				if (BOUNDARY(info)) sa_nic_coords(&sa, &boundElements[index]);   /* only a segment uses them */
			}
			int build_nl = FLUID(info) || TESTPOINT(info) || FLOATING(info) || COMPUTE_FORCE(info);
			if (p->boundarytype == ORC_SA_BOUNDARY)
				build_nl = build_nl || VERTEX(info) || BOUNDARY(info);
			if (p->boundarytype == ORC_DYN_BOUNDARY)
				build_nl = 1;
			if (!build_nl)
				break;
			const orc_f4 pos = posArray[index];
			if (INACTIVE(pos))
				break;
			const float pos3[3] = { pos.x, pos.y, pos.z };
			int gridPos[3];
			orc_grid_pos_from_hash(p, particleHash[index] & CELLTYPE_BITMASK, gridPos);
			for (int z = -1; z <= 1; z++)
				for (int y = -1; y <= 1; y++)
					for (int x = -1; x <= 1; x++) {
						const int off[3] = { x, y, z };
						neibs_in_cell(p, neibsList, posArray, infoArray, cellStart, cellEnd,
							gridPos, off, (x + 1) + (y + 1)*3 + (z + 1)*9,
							index, pos3, neibs_num, BOUNDARY(info), sqinfluenceradius, with_sa ? &sa : NULL);
					}
		} while (0);

		/* list terminators, :1105-1138 */
		int overflow = too_many_neibs(p, neibs_num, PT_FLUID);
		int marker_pos = overflow ? p->neibboundpos : neibs_num[PT_FLUID];
		neibsList[(size_t)marker_pos*p->neiblist_stride + index] = NEIBS_END;
		overflow |= too_many_neibs(p, neibs_num, PT_BOUNDARY);
		if (!overflow)
			neibsList[(size_t)neib_list_offset(p, neibs_num[PT_BOUNDARY], PT_BOUNDARY)*p->neiblist_stride + index] = NEIBS_END;
		if (p->boundarytype == ORC_SA_BOUNDARY) {
			overflow |= too_many_neibs(p, neibs_num, PT_VERTEX);
			marker_pos = overflow ? p->neiblistsize - 1 : p->neibboundpos + 1 + neibs_num[PT_VERTEX];
			neibsList[(size_t)marker_pos*p->neiblist_stride + index] = NEIBS_END;
		}
		if (overflow) {
#pragma omp critical
			{
				/* the reference records whichever overflowing particle wins the atomicCAS;
				 * the oracle records the lowest index one */
				const int pid = (int)orc_info_id(infoArray[index]);
				if (hasTooMany == -1 || pid < hasTooMany) {
					hasTooMany = pid;
					hasMax[0] = neibs_num[0]; hasMax[1] = neibs_num[1]; hasMax[2] = neibs_num[2];
				}
			}
		}
		/* neibcount reduction, :1141-1182 */
		const int nm = neibs_num[PT_FLUID] + neibs_num[PT_BOUNDARY];
		if (nm > maxNeibs) maxNeibs = nm;
		if (p->boundarytype == ORC_SA_BOUNDARY && (int)neibs_num[PT_VERTEX] > maxVertexNeibs) maxVertexNeibs = neibs_num[PT_VERTEX];
		numInteractions += nm + neibs_num[PT_VERTEX];
	}
	if (out) {
		out->numInteractions = (int32_t)numInteractions;
		out->maxFluidBoundaryNeibs = maxNeibs;
		out->maxVertexNeibs = maxVertexNeibs;
		out->hasTooManyNeibs = hasTooMany;
		out->hasMaxNeibs[0] = hasMax[0]; out->hasMaxNeibs[1] = hasMax[1]; out->hasMaxNeibs[2] = hasMax[2];
	}
}

/* ---- neighbour list traversal: src/cuda/neibs_iteration.cuh:83-190, cellgrid.cuh:200-228 ---- */
typedef struct {
	const orc_params *p;
	const uint32_t *cellStart;
	const uint16_t *neibsList;
	float pos[3];
	int gridPos[3];
	uint32_t index;
	float pos_corr[3];
	int64_t i;               /* idx_t list offset */
	uint32_t neib_cell_base_index;
	int neib_cellnum;
	int64_t step;
} neib_iter;

static void neib_iter_init(neib_iter *it, const orc_params *p, int ptype, uint32_t index,
	const orc_f4 *pos, const int gridPos[3], const uint32_t *cellStart, const uint16_t *neibsList)
{
	it->p = p; it->cellStart = cellStart; it->neibsList = neibsList;
	it->pos[0] = pos->x; it->pos[1] = pos->y; it->pos[2] = pos->z;
	it->gridPos[0] = gridPos[0]; it->gridPos[1] = gridPos[1]; it->gridPos[2] = gridPos[2];
	it->index = index;
	it->pos_corr[0] = it->pos_corr[1] = it->pos_corr[2] = 0.0f;
	it->neib_cell_base_index = 0;
	it->neib_cellnum = 0;
	const int64_t stride = (int64_t)p->neiblist_stride;
	const int64_t first = (ptype == PT_FLUID) ? 0 : (ptype == PT_BOUNDARY) ? p->neibboundpos : p->neibboundpos + 1;
	it->step = (ptype == PT_BOUNDARY) ? -stride : stride;
	it->i = first*stride - it->step;
}

/* returns neighbour index, or UINT_MAX at end of list */
static uint32_t neib_iter_next(neib_iter *it)
{
	it->i += it->step;
	uint16_t neib_data = it->neibsList[it->i + it->index];
	if (neib_data == NEIBS_END) return UINT_MAX;
	if (neib_data >= CELLNUM_ENCODED) {
		const orc_params *p = it->p;
		it->neib_cellnum = DECODE_CELL(neib_data);
		neib_data &= NEIBINDEX_MASK;
		/* d_cell_to_offset[c] = (c%3-1, (c/3)%3-1, c/9-1), src/cuda/forces.cu:376-386 */
		const int c = it->neib_cellnum;
		const int off[3] = { c % 3 - 1, (c / 3) % 3 - 1, c / 9 - 1 };
		/* pos_corr = pos - d_cell_to_offset*d_cellSize : contracted a - b*c */
		for (int a = 0; a < 3; ++a)
			it->pos_corr[a] = fmaf(-(float)off[a], p->cellSize[a], it->pos[a]);
		it->neib_cell_base_index = it->cellStart[calc_grid_hash_periodic(p,
			it->gridPos[0] + off[0], it->gridPos[1] + off[1], it->gridPos[2] + off[2])];
	}
	return it->neib_cell_base_index + neib_data;
}

/* ---- forces: src/cuda/forces_kernel.def ------------------------------------------- */

/* artvisc: src/cuda/visc_kernel.cu:74-85 */
static inline float artvisc(const orc_params *p, float vel_dot_pos, float rho, float neib_rho,
	float sspeed, float neib_sspeed, float r, float slength)
{
	return vel_dot_pos*slength*p->artvisccoeff*(sspeed + neib_sspeed)/
		((r*r + p->epsartvisc)*(rho + neib_rho));
}

static inline float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
	return fmaf(az, bz, fmaf(ay, by, ax*bx));
}

/* one forcesDevice<cptype,nptype> launch over [from,to): forces_kernel.def:3914-4029 */
/* visc_avg (src/cuda/visc_avg.cu:40-190): the per-pair factor of the laminar (Morris) viscous term, neighbour mass
 * included.  visc/neib_visc = get_visc_coeff = d_visccoeff of each particle's fluid (Newtonian, no k-epsilon:
 * forces_kernel.def:247-270). */
static inline float visc_avg_rho(int avgop, float rho, float neib_rho, float neib_mass)
{
	switch (avgop) {
	case ORC_ARITHMETIC: return neib_mass*(rho + neib_rho)/(rho*neib_rho);
	case ORC_HARMONIC:   return 4*neib_mass/(rho + neib_rho);
	default:             return 2*neib_mass*(1.0f/sqrtf(rho*neib_rho));   /* rsqrt */
	}
}
static inline float visc_avg_dyn(int avgop, int is_const, float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{
	if (is_const) return 2*neib_mass*visc/(rho*neib_rho);
	switch (avgop) {
	case ORC_ARITHMETIC: return neib_mass*(visc + neib_visc)/(rho*neib_rho);
	case ORC_HARMONIC:   return 4*neib_mass*(visc*neib_visc)/(visc + neib_visc)/(rho*neib_rho);
	default:             return 2*neib_mass*sqrtf(visc*neib_visc)/(rho*neib_rho);
	}
}
static inline float visc_avg(const orc_params *p, float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{
	if (p->compvisc == ORC_DYNAMIC)
		return visc_avg_dyn(p->avgop, p->is_const_visc, visc, neib_visc, rho, neib_rho, neib_mass);
	if (p->is_const_visc)
		return visc*visc_avg_rho(p->avgop, rho, neib_rho, neib_mass);
	/* non-constant kinematic: "just call the dynvisc variant" of ViscSpec::with_computational_visc<DYNAMIC> (:180-190).
	 * That alias does not carry is_const_visc over: the dynamic spec gets the DEFAULT constness of its framework flags,
	 * IS_SINGLEFLUID && NEWTONIAN (src/visc_spec.h:268-272,298-300) -- so a single-fluid NEWTONIAN spec forced to non-constant
	 * viscosity ends in the constant dynamic formula 2 m mu_i/(rho_i rho_j); pinned by tests/golden/ref_viscavg.npz */
	return visc_avg_dyn(p->avgop, !(p->simflags & ORC_ENABLE_MULTIFLUID) && p->rheologytype == ORC_NEWTONIAN &&
		p->turbmodel != ORC_KEPSILON,
		visc*rho, neib_visc*neib_rho, rho, neib_rho, neib_mass);
}

/* exported for the pinning test against the reference's own visc_avg (oracle/ref_shim.cc, tests/golden/ref_viscavg.npz) */
float orc_visc_avg(const orc_params *p, float visc, float neib_visc, float rho, float neib_rho, float neib_mass)
{ return visc_avg(p, visc, neib_visc, rho, neib_rho, neib_mass); }

/* SA_BOUNDARY members of forces_params / finalize_forces_params (src/cuda/forces_params.h): gamma and its gradient,
 * the boundary elements and the in-plane vertex offsets of the segments */
/* KEPSILON members (keps_forces_params, src/cuda/forces_params.h:283-320): k, epsilon, the eddy viscosity and the Eulerian
 * velocity are read; BUFFER_DKDE (diffusion terms of k and epsilon, Yap's C_e2) and BUFFER_TAU (the strain rate sums) are written
 * by EVERY forcesDevice launch from a freshly initialised keps_particle_output (forces_particle_output :1003-1013 default-constructs
 * it, write_keps :3331-3339 stores it), so what the finalize kernel reads is what the LAST launch over a particle left: for a fluid
 * particle the fluid <- boundary sums.  Restated launch by launch, so that this follows from the structure. */
typedef struct {
	const float *tke, *eps, *turbvisc;
	const orc_f4 *eulerVel;
	float *dkde;          /* 3 per particle */
	float *strain;        /* 6 per particle: xx, xy, xz, yy, yz, zz (off-diagonal sums premultiplied by two, :921-935) */
	float *cflKeps;       /* one per block, or NULL */
	float epsilon;
} sa_keps_ctx;
typedef struct {
	const orc_f4 *gGam, *boundelem;
	const float *vertPos[3];
	float deltap;
	float *gammaCfl;      /* per particle max of |grad gamma_as| |n.v| (dynamic gamma + ENABLE_DTADAPT), or NULL */
	const sa_keps_ctx *ke;
	/* ENABLE_INLET_OUTLET without k-epsilon (GROUNDWORK, see "Open boundaries" at the end of the file): BUFFER_EULERVEL, or NULL.
	 * Every use below is behind `io`, so that the passes without open boundaries are the instructions they were */
	const orc_f4 *ioEulerVel;
} sa_forces_ctx;
static inline void keps_add_strain(float *t, float vx, float vy, float vz, float mx, float my, float mz)
{	/* keps_particle_output::add_strain_rate :924-937 */
	t[0] += vx*mx; t[1] += vx*my + vy*mx; t[2] += vx*mz + vz*mx;
	t[3] += vy*my; t[4] += vy*mz + vz*my; t[5] += vz*mz;
}
float orc_grad_gamma_vp(float slength, float qx, float qy, float qz, const orc_f4 *belem,
	const float *vp0, const float *vp1, const float *vp2);

/* ENABLE_INTERNAL_ENERGY: BUFFER_INTERNAL_ENERGY_UPD of the forces passes (add_internal_energy, forces_kernel.def:3308-3320:
 * DEDt -= (DvDt_pair . relVel)/2 for every pair whose momentum term is computed; internal_energy_particle_output :972-980
 * starts from the value the previous pass left).  Test state of the oracle, set by orc_set_dedt like the texture of the DEM. */
static float *g_dedt;
void orc_set_dedt(float *dedt) { g_dedt = dedt; }

static void forces_pass(const orc_params *p, int cptype, int nptype, orc_f4 *forces,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *tauArray,
	uint32_t fromParticle, uint32_t toParticle, const sa_forces_ctx *sa, const float *effvisc)
{
	/* effvisc: BUFFER_EFFVISC of the generalized Newtonian rheologies (get_laminar_visc_coeff :250-259), else NULL */
	const int newtonian = p->rheologytype != ORC_INVISCID;      /* every viscous rheology has the laminar term */
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f; /* kernelradius */
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);

#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = fromParticle; index < toParticle; ++index) {
		const orc_info info = infoArray[index];
		if (PART_TYPE(info) != cptype) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;

		/* forces_particle_data, :759-812 */
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const orc_f4 vel = velArray[index];
		const int p_fluid = FLUID_NUM(info);
		const float p_sspeed = orc_soundSpeed(p, vel.w, p_fluid);
		const float p_rho = physical_density(p, vel.w, p_fluid);
		/* precalc_pressure SPH_F1: P/rho^2, :419-429 */
		/* ... SPH_F2: P (:431-441) */
		const int f2 = p->sph_formulation == ORC_SPH_F2;
		/* SPH_HA (Hu & Adams): precalc = P (:458-468), volumes V = m/rho in the pressure term, own mass in the continuity equation */
		const int ha = p->sph_formulation == ORC_SPH_HA;
		const sa_keps_ctx *ke = sa ? sa->ke : NULL;
		const int io = sa && sa->ioEulerVel != NULL && !ke;
		/* pressure_for_precalc with KEPSILON :389-401: P + 2/3 k/rho */
		const float p_precalc = ke ? (orc_P(p, vel.w, p_fluid) + 2.0f*ke->tke[index]/p_rho/3.0f)/(p_rho*p_rho) :
			(f2 || ha) ? orc_P(p, vel.w, p_fluid) : orc_P(p, vel.w, p_fluid)/(p_rho*p_rho);
		const float *p_tau = tauArray ? tauArray + 6*(size_t)index : NULL;
		/* keps_particle_data :633-655, keps_precalc_particle_data :708-722, eulerVel_particle_data :553-561; keps_particle_output() :941-948 */
		const float p_k = ke ? ke->tke[index] : 0.0f, p_e = ke ? ke->eps[index] : 0.0f, p_turb = ke ? ke->turbvisc[index] : 0.0f;
		const float p_tvv = (ke && FLUID(info)) ? p_turb : 0.0f;
		const orc_f4 p_euler = ke ? ke->eulerVel[index] : (orc_f4){ 0.0f, 0.0f, 0.0f, 0.0f };
		const float dkdt_precalc = p_rho*(p->visccoeff[p_fluid] + p_turb), dedt_precalc = p_rho*(p->visccoeff[p_fluid] + p_turb/1.3f);
		float diff_k = 0.0f, diff_e = 0.0f, ce2yap = 1.92f, strain[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };

		orc_f4 force = forces[index]; /* common_particle_output, :886-895 */
		const int energy = g_dedt && (p->simflags & ORC_ENABLE_INTERNAL_ENERGY);
		float dedt = energy ? g_dedt[index] : 0.0f;

		neib_iter it;
		neib_iter_init(&it, p, nptype, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			/* relPos = pos_corr - neibPos, .w = neib mass (vector_math.h:1064-1067) */
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const float nmass = npos.w;
			if (!isfinite(nmass)) continue;
			const float r = sqrtf(sqlength3(rx, ry, rz));
			const orc_info neib_info = infoArray[neib_index];
			/* forcesDevice :3999-4006: boundary elements of SA_BOUNDARY interact a little beyond the kernel radius */
			if (sa && nptype == PT_BOUNDARY) {
				if (r >= p->influenceradius + sa->deltap) continue;
			} else if (r >= p->influenceradius) continue;

			/* common_neib_data, :1099-1130 */
			const orc_f4 nvel = velArray[neib_index];
			const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
			const float n_rhot = nvel.w;
			const float vel_dot_pos = dot3(vx, vy, vz, rx, ry, rz);
			const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
			const int n_fluid = FLUID_NUM(neib_info);
			const float n_sspeed = orc_soundSpeed(p, n_rhot, n_fluid);
			const float n_rho = physical_density(p, n_rhot, n_fluid);
			const float n_precalc = ke ? (orc_P(p, n_rhot, n_fluid) + 2.0f*ke->tke[neib_index]/n_rho/3.0f)/(n_rho*n_rho) :
				(f2 || ha) ? orc_P(p, n_rhot, n_fluid) : orc_P(p, n_rhot, n_fluid)/(n_rho*n_rho);
			/* get_viscous_relVel :2494-2507: relVel + relEulerVel where an Eulerian velocity exists (eulerVel_neib_data :1152-1162) */
			float wx = vx, wy = vy, wz = vz;
			if (ke) {
				const orc_f4 ne = ke->eulerVel[neib_index];
				wx = vx + (p_euler.x - ne.x); wy = vy + (p_euler.y - ne.y); wz = vz + (p_euler.z - ne.z);
			} else if (io && cptype != nptype) {      /* eulerVel_neib_data exists for cptype != nptype only (:1183-1185) */
				const orc_f4 pe = sa->ioEulerVel[index], ne = sa->ioEulerVel[neib_index];
				wx = vx + (pe.x - ne.x); wy = vy + (pe.y - ne.y); wz = vz + (pe.z - ne.z);
			}

			float DvDt[3] = {0.0f, 0.0f, 0.0f};
			float DrDt = 0.0f;

			const int all_pp = (cptype == PT_FLUID && nptype == PT_FLUID) ||
				(cptype == PT_FLUID && nptype == PT_BOUNDARY && p->boundarytype == ORC_DYN_BOUNDARY) ||
				(cptype == PT_FLUID && nptype == PT_VERTEX && sa);     /* vertex particles weigh in like fluid ones, :3783-3795 */

			/* fluid <- boundary element with SA_BOUNDARY (compute_all_pp_interaction with the boundary specialisations):
			 * everything goes through |grad gamma_as| of the element */
			if (sa && cptype == PT_FLUID && nptype == PT_BOUNDARY) {
				const orc_f4 belem = sa->boundelem[neib_index];
				/* compute_gamma_gradient :1410-1428 */
				const float inv_h = 1.0f/p->slength;        /* float3/float multiplies by the reciprocal */
				const float ggamAS = orc_grad_gamma_vp(p->slength, rx*inv_h, ry*inv_h, rz*inv_h, &belem,
					sa->vertPos[0] + 2*(size_t)neib_index, sa->vertPos[1] + 2*(size_t)neib_index, sa->vertPos[2] + 2*(size_t)neib_index);
				/* mass_continuity_div_vel_term :2079-2090 (nout.DrDt starts from 0) */
				const float vn = dot3(vx, vy, vz, belem.x, belem.y, belem.z);
				if (sa->gammaCfl) {     /* compute_gamma_cfl_solid_wall :1458-1474: n.(v_a - v_s), n.v_a, n.v_s */
					const float va = dot3(vel.x, vel.y, vel.z, belem.x, belem.y, belem.z);
					const float vs = dot3(vel.x - vx, vel.y - vy, vel.z - vz, belem.x, belem.y, belem.z);
					sa->gammaCfl[index] = fmaxf(sa->gammaCfl[index], ggamAS*fmaxf(fabsf(vn), fmaxf(fabsf(va), fabsf(vs))));
					if (io) {           /* compute_gamma_cfl_open_boundary :1485-1497: n.(v_a + relEulerVel), n.(v_s - relEulerVel) */
						const float ex = wx - vx, ey = wy - vy, ez = wz - vz;      /* relEulerVel */
						const float a1 = dot3(vel.x + ex, vel.y + ey, vel.z + ez, belem.x, belem.y, belem.z);
						const float a2 = dot3(-vx + vel.x - ex, -vy + vel.y - ey, -vz + vel.z - ez, belem.x, belem.y, belem.z);
						sa->gammaCfl[index] = fmaxf(sa->gammaCfl[index], ggamAS*fmaxf(fabsf(a1), fabsf(a2)));
					}
				}
				if (!(p->simflags & ORC_ENABLE_DENSITY_SUM)) {
					DrDt -= p_rho*vn*ggamAS;
					if (f2) DrDt *= p_rho/n_rho;
					force.w += DrDt;
				}
				/* compute_pressure_contrib :2414-2427: + (P_a/rho_a^2 + P_s/rho_s^2) rho_s |grad gamma_as| n_s */
				const float pGradTerm = p_precalc + n_precalc;
				const float ps = pGradTerm*n_rho*ggamAS;
				DvDt[0] += ps*belem.x; DvDt[1] += ps*belem.y; DvDt[2] += ps*belem.z;
				if (ke) {
					const float r_as = fmaxf(fabsf(dot3(rx, ry, rz, belem.x, belem.y, belem.z)), sa->deltap);
					const float our_visc = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[p_fluid] : p->visccoeff[p_fluid]/p_rho;
					const float neib_visc = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[n_fluid] : p->visccoeff[n_fluid]/n_rho;
					/* compute_turb_visc_contrib, boundary term :2824-2878: the wall shear stress from the law of the wall (the laminar
					 * wall term is left out with k-epsilon and solid walls, :2695-2702) */
					if (!(p_k < ke->epsilon)) {
						const float ux = vx + p_euler.x, uy = vy + p_euler.y, uz = vz + p_euler.z;     /* relVel + pdata.eulerVel */
						const float un = dot3(ux, uy, uz, belem.x, belem.y, belem.z);
						const float ut[3] = { ux - un*belem.x, uy - un*belem.y, uz - un*belem.z };
						const float abs_u_t = sqrtf(dot3(ut[0], ut[1], ut[2], ut[0], ut[1], ut[2]));
						float u_star = 0.0f;
						const float uk = 0.547722558f*sqrtf(p_k);
						float y_plus = r_as/our_visc*uk;
						if (y_plus < 2.43902439f)
							u_star = abs_u_t/y_plus;
						else {
							float utau = 0.118599857f*neib_visc/r_as;
							for (int i = 0; i < 10; i++) {
								y_plus = fmaxf(r_as*utau/neib_visc, 2.43902439f);
								utau = (0.41f*abs_u_t + utau)/(logf(y_plus) + 3.132f);
							}
							u_star = abs_u_t/(logf(y_plus)/0.41f + 5.2f);
						}
						const float sc = 2.0f*ggamAS*u_star*u_star, inv = 1.0f/fmaxf(abs_u_t, 1e-6f);
						DvDt[0] -= (sc*ut[0])*inv; DvDt[1] -= (sc*ut[1])*inv; DvDt[2] -= (sc*ut[2])*inv;
					}
					/* compute_keps_term, boundary term :2949-2980 */
					const float lyap = 0.400772603f*powf(p_k, 1.5f)/(p_e*r_as);
					if (lyap > 1.0f)
						ce2yap = fminf(ce2yap, fmaxf(1.92f - 0.83f*(lyap - 1.0f)*lyap*lyap, 0.0f));
					diff_e += 0.276923077f*p_k*p_k/r_as*ggamAS;
					keps_add_strain(strain, wx, wy, wz, (ggamAS*belem.x)*n_rho, (ggamAS*belem.y)*n_rho, (ggamAS*belem.z)*n_rho);
				} else
				/* compute_laminar_visc_contrib, boundary term :2680-2718 (MORRIS, no k-epsilon, no open boundaries) */
				if (p->rheologytype == ORC_NEWTONIAN) {
					const float r_as = fmaxf(fabsf(dot3(rx, ry, rz, belem.x, belem.y, belem.z)), sa->deltap);   /* sa_boundary_neib_data :1132-1150 */
					float vt[3] = { vx - vn*belem.x, vy - vn*belem.y, vz - vn*belem.z };
					if (io) {           /* :2703-2708: relVel + relEulerVel; against an open-boundary segment the whole of it, not the tangential part */
						const float wn = IO_BOUNDARY_X(neib_info) ? 0.0f : dot3(wx, wy, wz, belem.x, belem.y, belem.z);
						vt[0] = wx - wn*belem.x; vt[1] = wy - wn*belem.y; vt[2] = wz - wn*belem.z;
					}
					/* get_laminar_dyn_visc :322-340 */
					const float our_mu = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[p_fluid]*p_rho : p->visccoeff[p_fluid];
					const float neib_mu = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[n_fluid]*n_rho : p->visccoeff[n_fluid];
					float avg;
					switch (p->avgop) {           /* average<>, src/average.h:78-100 */
					case ORC_ARITHMETIC: avg = (our_mu + neib_mu)*0.5f; break;
					case ORC_HARMONIC:   avg = 2*our_mu*neib_mu/(our_mu + neib_mu); break;
					default:             avg = sqrtf(our_mu*neib_mu);
					}
					const float c = ggamAS*2*avg/r_as;
					const float inv_rho = 1.0f/p_rho;
					DvDt[0] -= (c*vt[0])*inv_rho; DvDt[1] -= (c*vt[1])*inv_rho; DvDt[2] -= (c*vt[2])*inv_rho;
				}
				force.x += DvDt[0]; force.y += DvDt[1]; force.z += DvDt[2];
				continue;
			}
			const int dyn_bf = (cptype == PT_BOUNDARY && nptype == PT_FLUID && p->boundarytype == ORC_DYN_BOUNDARY);

			/* repulsive boundary models: fluid <- boundary always, boundary <- fluid only for particles of bodies
			 * with force feedback (compute_pp_interaction :3620-3645,3688-3705; compute_repulsive_force :3001-3016;
			 * LJForce src/cuda/forces_kernel.cu:94-103, __powf -> powf) */
			if ((p->boundarytype == ORC_LJ_BOUNDARY || p->boundarytype == ORC_MK_BOUNDARY) &&
				((cptype == PT_FLUID && nptype == PT_BOUNDARY) ||
				 (cptype == PT_BOUNDARY && nptype == PT_FLUID && COMPUTE_FORCE(info)))) {
				float ljf = 0.0f;
				if (p->boundarytype == ORC_LJ_BOUNDARY) {
					if (r <= p->r0)
						ljf = p->dcoeff*(powf(p->r0/r, p->p1coeff) - powf(p->r0/r, p->p2coeff))/(r*r);
				} else if (r <= 2*p->slength) {
					/* MKForce(r, slength, mass_f, mass_b) with both masses = the central particle's
					 * (src/cuda/forces_kernel.cu:105-133, compute_repulsive_force :3011-3013) */
					const float qq = r/p->slength;
					const float w = 1.8f*powf(1.0f - 0.5f*qq, 4.0f)*(2.0f*qq + 1.0f);
					const float dist = fmaxf(p->epsartvisc, r - p->MK_d);
					ljf = p->MK_K*w*2*pos.w/(p->MK_beta*dist*r*(pos.w + pos.w));
				}
				force.x += ljf*rx; force.y += ljf*ry; force.z += ljf*rz;
				if (energy) dedt -= dot3(ljf*rx, ljf*ry, ljf*rz, vx, vy, vz)/2;
				continue;
			}

			if ((all_pp || dyn_bf) && !(p->simflags & ORC_ENABLE_DENSITY_SUM)) {
				/* compute_density_derivative, :2176-2190 */
				DrDt = nmass*vel_dot_pos*f; /* mass_continuity_div_vel_term :2140-2151 */
				if (ha) DrDt = pos.w*vel_dot_pos*f;      /* SPH_HA: the particle's own mass, :2030-2046 */
				/* compute_density_diffusion (Colagrossi, nptype==FLUID only), :1916-1952 */
				if (ha && p->densitydiffusiontype == ORC_COLAGROSSI && nptype == PT_FLUID) {
					/* Molteni & Colagrossi for SPH_HA (:1954-1996): volume ratio instead of density ratio, own mass */
					if (p_fluid == n_fluid) {
						const float gdotr = dot3(p->gravity[0], p->gravity[1], p->gravity[2], rx, ry, rz);
						if (!(fabsf(orc_P(p, vel.w, p_fluid) - orc_P(p, n_rhot, p_fluid)) < fabsf(gdotr*p_rho))) {
							const float p_volume = pos.w/p_rho, n_volume = nmass/n_rho;
							DrDt -= p->densityDiffCoeff*p->sscoeff[p_fluid]*(p_volume/n_volume - 1)*f*pos.w;
						}
					}
				} else
				if (p->densitydiffusiontype == ORC_COLAGROSSI && nptype == PT_FLUID) {
					const int fType = p_fluid;
					if (fType == n_fluid) {
						const float gdotr = dot3(p->gravity[0], p->gravity[1], p->gravity[2], rx, ry, rz);
						if (!(fabsf(orc_P(p, vel.w, fType) - orc_P(p, n_rhot, fType)) <
								fabsf(gdotr*p_rho))) {
							const float diff_term = p->densityDiffCoeff*p->sscoeff[fType]*
								(n_rho/p_rho - 1)*f*nmass;
							DrDt -= diff_term;
						}
					}
				}
				/* compute_density_diffusion (Ferrari, fluid neighbours only outside SA), :1607-1635; d_sqC0 = sscoeff^2
				 * in float (src/cuda/forces.cu:319-325) */
				if (ha && p->densitydiffusiontype == ORC_FERRARI && nptype == PT_FLUID) {
					/* Ferrari for SPH_HA (:1639-1677): no inter-phase diffusion; (rho - rho') becomes m (1/V - theta'/(theta V')), theta = 1 outside SA */
					if (p_fluid == n_fluid) {
						const float sqC0 = p->sscoeff[p_fluid]*p->sscoeff[p_fluid];
						const float grav_corr = -dot3(p->gravity[0], p->gravity[1], p->gravity[2], rx, ry, rz)*p->rho0[p_fluid]/sqC0;
						const float p_volume = pos.w/p_rho, n_volume = nmass/n_rho;
						float fc[3] = {0.0f, 0.0f, 0.0f};
						if (r > 1e-4f*p->slength) {
							/* the reference writes 1./p_volume: the bracket, and with it the whole scalar factor, is evaluated in double */
							const float sc = (float)((double)fmaxf(p_sspeed, n_sspeed)*
								((double)pos.w*(1./(double)p_volume - (double)(1.0f/(1.0f*n_volume))) + (double)grav_corr)/(double)p_rho/(double)r);
							fc[0] = sc*rx; fc[1] = sc*ry; fc[2] = sc*rz;
						}
						DrDt += p->densityDiffCoeff*nmass*dot3(fc[0], fc[1], fc[2], rx, ry, rz)*f;
					}
				} else
				if (p->densitydiffusiontype == ORC_FERRARI && nptype == PT_FLUID) {
					const int fType = p_fluid;
					const float sqC0 = p->sscoeff[fType]*p->sscoeff[fType];
					const float grav_corr = -dot3(p->gravity[0], p->gravity[1], p->gravity[2], rx, ry, rz)*p->rho0[fType]/sqC0;
					float fc[3] = {0.0f, 0.0f, 0.0f};
					if (r > 1e-4f*p->slength) {
						const float sc = fmaxf(p_sspeed, n_sspeed)*(p_rho - n_rho + grav_corr)/p_rho/r;
						fc[0] = sc*rx; fc[1] = sc*ry; fc[2] = sc*rz;
					}
					DrDt += p->densityDiffCoeff*nmass*dot3(fc[0], fc[1], fc[2], rx, ry, rz)*f;
				}
				if (f2) DrDt *= p_rho/n_rho;   /* mass_continuity_density_ratio, after the diffusion term (:2154-2188) */
				force.w += DrDt;
			}

			if (all_pp || (dyn_bf && (COMPUTE_FORCE(info) || (p->simflags & ORC_ENABLE_INTERNAL_ENERGY)))) {      /* :3661 */
				/* compute_pressure_contrib general, :2451-2466 */
				/* pressure_gradient_term: SPH_F1 P_i/rho_i^2 + P_j/rho_j^2 (:2358-2371), SPH_F2 (P_i + P_j)/(rho_i rho_j) (:2253-2266) */
				float pGradTerm = f2 ? (p_precalc + n_precalc)/(p_rho*n_rho) : p_precalc + n_precalc;
				float s = pGradTerm*nmass*f;
				if (ha) {      /* pressure_gradient_term :2269-2286 (P_a V_a^2 + P_b V_b^2), compute_pressure_contrib :2436-2448 (/ m_a) */
					const float p_volume = pos.w/p_rho, n_volume = nmass/n_rho;
					pGradTerm = p_precalc*p_volume*p_volume + n_precalc*n_volume*n_volume;
					s = pGradTerm/pos.w*f;
				}
				DvDt[0] -= s*rx; DvDt[1] -= s*ry; DvDt[2] -= s*rz;

				/* compute_viscous_contrib: turbulent first, then laminar, :2881-2886 */
				if (p->turbmodel == ORC_ARTIFICIAL) { /* :2748-2764 */
					if (vel_dot_pos < 0.0f) {
						const float visc = artvisc(p, vel_dot_pos, p_rho, n_rho, p_sspeed, n_sspeed, r, p->slength);
						DvDt[0] += visc*rx*nmass*f;
						DvDt[1] += visc*ry*nmass*f;
						DvDt[2] += visc*rz*nmass*f;
					}
				} else if (p->turbmodel == ORC_SPS && p_tau) { /* :2777-2798 */
					const float *n_tau = tauArray + 6*(size_t)neib_index;
					/* symtensor3 order: xx, xy, xz, yy, yz, zz (src/cuda/tensor.h) */
					DvDt[0] += nmass*f*(
						(p_tau[0] + n_tau[0])*rx + (p_tau[1] + n_tau[1])*ry + (p_tau[2] + n_tau[2])*rz);
					DvDt[1] += nmass*f*(
						(p_tau[1] + n_tau[1])*rx + (p_tau[3] + n_tau[3])*ry + (p_tau[4] + n_tau[4])*rz);
					DvDt[2] += nmass*f*(
						(p_tau[2] + n_tau[2])*rx + (p_tau[4] + n_tau[4])*ry + (p_tau[5] + n_tau[5])*rz);
				}
				/* compute_laminar_visc_contrib, Newtonian + MORRIS (:2606-2625): fluid neighbours, and boundary neighbours
				 * of DYN_BOUNDARY (wants_volumic_visc_term :150-158; LJ pairs never get here) */
				if (newtonian && p->viscmodel == ORC_ESPANOL_REVENGA) {
					/* compute_laminar_visc_contrib, Espanol & Revenga (:2650-2677): shear and bulk viscosity, contributions along the
					 * relative velocity and along the relative position */
					const float pvisc = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[p_fluid]*p_rho : p->visccoeff[p_fluid];
					const float nvisc = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[n_fluid]*n_rho : p->visccoeff[n_fluid];
					const float pbulk = p->visc2coeff[p_fluid], nbulk = p->visc2coeff[n_fluid];
					float avs, avb;
					switch (p->avgop) {           /* average<>, src/average.h:78-100 */
					case ORC_ARITHMETIC: avs = (pvisc + nvisc)*0.5f; avb = (pbulk + nbulk)*0.5f; break;
					case ORC_HARMONIC:   avs = 2*pvisc*nvisc/(pvisc + nvisc); avb = 2*pbulk*nbulk/(pbulk + nbulk); break;
					default:             avs = sqrtf(pvisc*nvisc); avb = sqrtf(pbulk*nbulk);
					}
					const float visc_thirds = avs/3;
					const float coeff = nmass/(p_rho*n_rho)*f;      /* viscous_volume_coefficient :2575-2580 */
					const float pos_den = dot3(rx, ry, rz, rx, ry, rz) + p->epsartvisc;
					const float cv = 5*visc_thirds - avb, cr = 5*(visc_thirds + avb)*vel_dot_pos/pos_den;
					DvDt[0] += coeff*(cv*vx + cr*rx); DvDt[1] += coeff*(cv*vy + cr*ry); DvDt[2] += coeff*(cv*vz + cr*rz);
				} else if (newtonian && ke) {
					/* get_visc_coeff with KEPSILON :262-270: the laminar coefficient + the eddy viscosity (fluid particles only,
					 * turbViscForViscTerm :645-655); MORRIS along relVel + relEulerVel */
					const float n_tvv = FLUID(neib_info) ? ke->turbvisc[neib_index] : 0.0f;
					const float vf = visc_avg(p, p->visccoeff[p_fluid] + p_tvv, p->visccoeff[n_fluid] + n_tvv, p_rho, n_rho, nmass)*f;
					DvDt[0] += vf*wx; DvDt[1] += vf*wy; DvDt[2] += vf*wz;
					/* compute_keps_term, volumic term :2915-2946 */
					const float n_kin = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[n_fluid] : p->visccoeff[n_fluid]/n_rho;
					const float n_turb = ke->turbvisc[neib_index];
					diff_k += nmass*(dkdt_precalc + n_rho*(n_kin + n_turb))*(p_k - ke->tke[neib_index])*f/n_rho;
					diff_e += nmass*(dedt_precalc + n_rho*(n_kin + n_turb/1.3f))*(p_e - ke->eps[neib_index])*f/n_rho;
					keps_add_strain(strain, wx, wy, wz, (-nmass*rx)*f, (-nmass*ry)*f, (-nmass*rz)*f);
				} else if (newtonian) {
					const float visc = effvisc ? visc_avg(p, effvisc[index], effvisc[neib_index], p_rho, n_rho, nmass) :
						visc_avg(p, p->visccoeff[p_fluid], p->visccoeff[n_fluid], p_rho, n_rho, nmass);
					const float vf = visc*f;
					if (p->viscmodel == ORC_MONAGHAN) {
						/* viscous_vector_component<MONAGHAN> (:2531-2562): along the relative position, approaching pairs only */
						const float den = dot3(rx, ry, rz, rx, ry, rz) + p->epsartvisc;
						const float c = vel_dot_pos < 0 ? p->monaghan_visc_coeff*vel_dot_pos/den : 0.0f;
						DvDt[0] += vf*(c*rx); DvDt[1] += vf*(c*ry); DvDt[2] += vf*(c*rz);
					} else if (io) {      /* get_viscous_relVel :2494-2507 */
						DvDt[0] += vf*wx; DvDt[1] += vf*wy; DvDt[2] += vf*wz;
					} else {
						DvDt[0] += vf*vx; DvDt[1] += vf*vy; DvDt[2] += vf*vz;
					}
				}
				if (all_pp || COMPUTE_FORCE(info)) {
					force.x += DvDt[0]; force.y += DvDt[1]; force.z += DvDt[2];
				}
				if (energy) dedt -= dot3(DvDt[0], DvDt[1], DvDt[2], vx, vy, vz)/2;
			}
		}
		forces[index] = force;
		if (energy) g_dedt[index] = dedt;
		if (ke) {      /* write_keps :3331-3339 */
			float *d = ke->dkde + 3*(size_t)index;
			d[0] = diff_k; d[1] = diff_e; d[2] = ce2yap;
			memcpy(ke->strain + 6*(size_t)index, strain, sizeof strain);
		}
	}
}

/* getFmaxElements / reducefmax / round_particles: src/cuda/forces.cu:105-140,539-552,960-964 */
static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1)/b; }
static inline uint32_t round_up(uint32_t a, uint32_t b) { return div_up(a, b)*b; }
uint32_t orc_fmax_elements(uint32_t n) { return round_up(div_up(n, BLOCK_SIZE_FORCES), 4u); }
uint32_t orc_fmax_temp_elements(uint32_t nels)
{
	const uint32_t numquarts = nels/4;
	uint32_t numBlocks = div_up(numquarts, BLOCK_SIZE_FMAX);
	if (numBlocks > 1) {
		numBlocks = round_up(numBlocks, 4u);
		if (numBlocks > BLOCK_SIZE_FMAX*4) numBlocks = BLOCK_SIZE_FMAX*4;
	}
	return numBlocks;
}
uint32_t orc_round_particles(uint32_t n) { return (n/BLOCK_SIZE_FORCES)*BLOCK_SIZE_FORCES; }

/* viscous part of PlaneForce (src/cuda/forces_kernel.cu:153-185): v_t = vel - dot(vel, relPos)/r*relPos/r,
 * force += -dynvisc*partsurf/(mass*r) * v_t ; float3/float multiplies by the reciprocal (src/vector_math.h:526-530) */
static inline void plane_friction(const orc_params *p, orc_f4 *force, const orc_f4 *vel, const float rp[3], float r,
	float mass, float dynvisc)
{
	const float partsurf = (p->partsurf == 0.0f) ? p->r0*p->r0 : p->partsurf;
	const float d = (vel->x*rp[0] + vel->y*rp[1] + vel->z*rp[2])/r;
	const float inv = 1.0f/r;
	const float vt[3] = { vel->x - (d*rp[0])*inv, vel->y - (d*rp[1])*inv, vel->z - (d*rp[2])*inv };
	const float coeff = -dynvisc*partsurf/(mass*r);
	force->x += coeff*vt[0]; force->y += coeff*vt[1]; force->z += coeff*vt[2];
}

/* finalizeforcesDevice: forces_kernel.def:4032-4150 */
/* ---- ENABLE_DEM: the terrain as a height map (src/cuda/geom_core.cu:103-182, DemLJForce src/cuda/forces_kernel.cu:205-226).
 * The reference reads it through a 2D texture: unnormalised coordinates, clamped addressing, linear filtering, i.e. sample
 * centres at i + 0.5 and fractional weights kept in 1.8 fixed point by the texture unit (CUDA programming guide, "Linear
 * Filtering"); restated here on a plain array.  The map is test state of the oracle (orc_set_dem), like the texture binding. */
static const float *g_dem; static int g_dem_w, g_dem_h;
void orc_set_dem(const float *dem, int width, int height) { g_dem = dem; g_dem_w = width; g_dem_h = height; }

float orc_dem_interpol(float x, float y)
{
	const float xb = x - 0.5f, yb = y - 0.5f;
	const float fx = floorf(xb), fy = floorf(yb);
	const float al = rintf((xb - fx)*256.0f)*(1.0f/256.0f), be = rintf((yb - fy)*256.0f)*(1.0f/256.0f);
#define ORC_CLAMPI(v, hi) ((v) < 0 ? 0 : ((v) > (hi) ? (hi) : (v)))
	const int i0 = ORC_CLAMPI((int)fx, g_dem_w - 1), i1 = ORC_CLAMPI((int)fx + 1, g_dem_w - 1);
	const int j0 = ORC_CLAMPI((int)fy, g_dem_h - 1), j1 = ORC_CLAMPI((int)fy + 1, g_dem_h - 1);
	const float t00 = g_dem[(size_t)j0*g_dem_w + i0], t10 = g_dem[(size_t)j0*g_dem_w + i1];
	const float t01 = g_dem[(size_t)j1*g_dem_w + i0], t11 = g_dem[(size_t)j1*g_dem_w + i1];
	return (1.0f - al)*(1.0f - be)*t00 + al*(1.0f - be)*t10 + (1.0f - al)*be*t01 + al*be*t11;
}

/* the tangent plane of the terrain under a particle that is less than demzmin above it; 0 when it is higher */
static int dem_plane(const orc_params *p, const int gp[3], const orc_f4 *pos, float nrm[3], int pgp[3], float ppos[3])
{
	const float dx = (gp[0] + 0.5f)*(p->cellSize[0]/p->ewres) + pos->x/p->ewres + 0.5f;     /* DemPos */
	const float dy = (gp[1] + 0.5f)*(p->cellSize[1]/p->nsres) + pos->y/p->nsres + 0.5f;
	const float globalZ = p->worldOrigin[2] + (gp[2] + 0.5f)*p->cellSize[2] + pos->z;
	const float z0 = orc_dem_interpol(dx, dy);
	if (!(globalZ - z0 < p->demzmin)) return 0;
	const float z1 = orc_dem_interpol(dx + 1*p->demdx/p->ewres, dy), z2 = orc_dem_interpol(dx, dy + 1*p->demdy/p->nsres);
	const float a = p->demdy*(z0 - z1), b = p->demdx*(z0 - z2), c = p->demdx*p->demdy;
	const float inv = 1.0f/sqrtf(a*a + b*b + c*c);
	nrm[0] = a*inv; nrm[1] = b*inv; nrm[2] = c*inv;
	pgp[0] = gp[0]; pgp[1] = gp[1]; pgp[2] = (int)floorf((z0 - p->worldOrigin[2])/p->cellSize[2]);
	ppos[0] = pos->x; ppos[1] = pos->y; ppos[2] = z0 - p->worldOrigin[2] - (pgp[2] + 0.5f)*p->cellSize[2];
	return 1;
}

/* PlaneForce, src/cuda/forces_kernel.cu:140-185 */
static void plane_force(const orc_params *p, orc_f4 *force, const orc_f4 *pos, const orc_f4 *vel, const int gp[3],
	const float nrm[3], const int pgp[3], const float ppos[3], float dynvisc)
{
	const float dx = (gp[0] - pgp[0])*p->cellSize[0] + (pos->x - ppos[0]);
	const float dy = (gp[1] - pgp[1])*p->cellSize[1] + (pos->y - ppos[1]);
	const float dz = (gp[2] - pgp[2])*p->cellSize[2] + (pos->z - ppos[2]);
	const float r = fabsf(dx*nrm[0] + dy*nrm[1] + dz*nrm[2]);
	if (r < p->r0) {
		float DvDt = 0.0f;   /* LJForce(r) */
		if (r <= p->r0)
			DvDt = p->dcoeff*(powf(p->r0/r, p->p1coeff) - powf(p->r0/r, p->p2coeff))/(r*r);
		const float rp[3] = { nrm[0]*r, nrm[1]*r, nrm[2]*r };
		force->x += DvDt*rp[0]; force->y += DvDt*rp[1]; force->z += DvDt*rp[2];
		if (dynvisc != 0.0f)
			plane_friction(p, force, vel, rp, r, pos->w, dynvisc);
	}
}

static void finalize_forces(const orc_params *p, orc_f4 *forces, float *cfl,
	orc_f4 *rbforces, orc_f4 *rbtorques,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	uint32_t fromParticle, uint32_t toParticle, uint32_t numBlocks, uint32_t cflOffset, const sa_forces_ctx *sa,
	const float *sigma, const float *effvisc)
{
	const int dtadapt = !!(p->simflags & ORC_ENABLE_DTADAPT);
#pragma omp parallel for schedule(static)
	for (uint32_t block = 0; block < numBlocks; ++block) {
		float block_max = 0.0f; /* shared.init() */
		float block_max_keps = 0.0f;
		const sa_keps_ctx *ke = sa ? sa->ke : NULL;
		for (uint32_t t = 0; t < BLOCK_SIZE_FORCES; ++t) {
			const uint32_t index = block*BLOCK_SIZE_FORCES + t + fromParticle;
			if (index >= toParticle) break;
			const orc_info info = infoArray[index];
			const orc_f4 pos = posArray[index];
			if (INACTIVE(pos)) continue;
			const orc_f4 vel = velArray[index];
			orc_f4 force = forces[index];
			const int fl = FLUID_NUM(info);

			/* forces_fixup: non-SA, non-Grenier :3212-3218; SA_BOUNDARY :3192-3210 (fluid particles only: the sums are
			 * renormalised by gamma) */
			if (sigma) {     /* SPH_GRENIER :3181-3190: DvDt was summed without 1/rho, DJ/Dt without 1/sigma; every particle */
				const float rho = physical_density(p, vel.w, fl);
				force.x /= rho; force.y /= rho; force.z /= rho;
				force.w /= sigma[index];
			} else if (!sa)
				force.w /= p->rho0[fl];
			else if (FLUID(info)) {
				const float gam = sa->gGam[index].w;
				force.x /= gam; force.y /= gam; force.z /= gam; force.w /= gam;
				force.w /= p->rho0[fl];
				if (p->sph_formulation == ORC_SPH_F2 && !(p->simflags & ORC_ENABLE_DENSITY_SUM))
					force.w *= physical_density(p, vel.w, fl);
			}

			if (FLUID(info) && ke) {
				/* viscous_fixup with KEPSILON + SA_BOUNDARY :3123-3170: diffusion terms divided by rho gamma, production of k from the
				 * norm of the strain rate (limited to 0.3 k S), source term of epsilon */
				float *d = ke->dkde + 3*(size_t)index;
				const float *tau = ke->strain + 6*(size_t)index;
				const float rhoGam = physical_density(p, vel.w, fl)*sa->gGam[index].w;
				d[0] /= rhoGam; d[1] /= rhoGam;
				float SijSij_bytwo = 2.0f*(tau[0]*tau[0] + tau[3]*tau[3] + tau[5]*tau[5]) + tau[1]*tau[1] + tau[2]*tau[2] + tau[4]*tau[4];
				const float S = sqrtf(SijSij_bytwo)/rhoGam;
				SijSij_bytwo /= rhoGam*rhoGam;
				const float k = ke->tke[index];
				const float Pturb = fminf(ke->turbvisc[index]*SijSij_bytwo, 0.3f*k*S);
				d[0] += Pturb;
				d[1] += ke->eps[index]*1.44f*Pturb/k;
				if (dtadapt) block_max_keps = fmaxf(block_max_keps, ke->turbvisc[index]);      /* dyndt_keps_shared_data :3481-3501 */
			}
			if (FLUID(info)) {
				force.x += p->gravity[0]; force.y += p->gravity[1]; force.z += p->gravity[2];
				/* GeometryForce/PlaneForce (src/cuda/forces_kernel.cu:140-203), PlaneDistance + globalDistance
				 * (src/cuda/geom_core.cu:63-85, cellgrid.cuh:153-161).  Inviscid: the wall-friction coefficient is
				 * -0 (viscous_plane_coefficient :3103-3107), only the Lennard-Jones repulsion along the normal acts;
				 * Newtonian: get_laminar_dyn_visc (:322-340) = nu rho or mu. */
				const float lamvisc = effvisc ? effvisc[index] : p->visccoeff[fl];
				const float dynvisc = (p->rheologytype != ORC_INVISCID) ?
					(p->compvisc == ORC_KINEMATIC ? lamvisc*physical_density(p, vel.w, fl) : lamvisc) : 0.0f;
				if ((p->simflags & (ORC_ENABLE_PLANES | ORC_ENABLE_DEM))) {
					int gp[3];
					orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gp);
					/* DEM first, LJ_BOUNDARY only (finalizeforcesDevice :4093-4102), then the planes (:4105-4109) */
					if ((p->simflags & ORC_ENABLE_DEM) && g_dem && p->boundarytype == ORC_LJ_BOUNDARY) {
						float nrm[3], ppos[3]; int pgp[3];
						if (dem_plane(p, gp, &pos, nrm, pgp, ppos))
							plane_force(p, &force, &pos, &vel, gp, nrm, pgp, ppos, dynvisc);
					}
					if ((p->simflags & ORC_ENABLE_PLANES))
						for (uint32_t k = 0; k < p->numplanes; ++k)
							plane_force(p, &force, &pos, &vel, gp, p->plane_normal[k], p->plane_gridpos[k], p->plane_pos[k], dynvisc);
				}
				if (dtadapt) { /* dyndt_forces_shared_data::store, :3436-3457 */
					const float sspeed = orc_soundSpeed(p, vel.w, fl);
					const float a = sqrtf(sqlength3(force.x, force.y, force.z));
					block_max = fmaxf(block_max, fmaxf(a, sspeed*sspeed/p->slength));
				}
			}

			if (COMPUTE_FORCE(info) && !VERTEX(info) && rbforces) { /* :4121-4142 */
				force.x *= pos.w; force.y *= pos.w; force.z *= pos.w;
				const int obj = OBJECT_NUM(info);
				const uint32_t rbindex = (uint32_t)((int)orc_info_id(info) + p->rbstartindex[obj]);
				rbforces[rbindex] = force;
				int gp[3];
				orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gp);
				/* globalDistance, cellgrid.cuh:147-154 */
				const float ax = (gp[0] - p->rbcgGridPos[obj][0])*p->cellSize[0] + (pos.x - p->rbcgPos[obj][0]);
				const float ay = (gp[1] - p->rbcgGridPos[obj][1])*p->cellSize[1] + (pos.y - p->rbcgPos[obj][1]);
				const float az = (gp[2] - p->rbcgGridPos[obj][2])*p->cellSize[2] + (pos.z - p->rbcgPos[obj][2]);
				orc_f4 tq;
				tq.x = ay*force.z - az*force.y;
				tq.y = az*force.x - ax*force.z;
				tq.z = ax*force.y - ay*force.x;
				tq.w = 0.0f;
				rbtorques[rbindex] = tq;
			}
			forces[index] = force;
		}
		if (dtadapt && cfl)
			cfl[cflOffset + block] = block_max; /* maxBlockReduce, src/cuda/device_core.cu:40-59 */
		if (dtadapt && ke && ke->cflKeps)
			ke->cflKeps[cflOffset + block] = block_max_keps;
	}
}

/* run_forces: src/cuda/forces.cu:717-806 */
uint32_t orc_forces_effvisc(const orc_params *p, orc_f4 *forces, float *cfl,
	orc_f4 *rbforces, orc_f4 *rbtorques,
	const orc_f4 *pos, const orc_f4 *vel, const orc_info *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *tau,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	uint32_t cflOffset, int compute_object_forces, const float *effvisc)
{
	(void)numParticles;
	const uint32_t numBlocks = round_up(div_up(toParticle - fromParticle, BLOCK_SIZE_FORCES), 4u);
	forces_pass(p, PT_FLUID, PT_FLUID, forces, pos, vel, info, hash, cellStart, neibsList, tau, fromParticle, toParticle, NULL, effvisc);
	forces_pass(p, PT_FLUID, PT_BOUNDARY, forces, pos, vel, info, hash, cellStart, neibsList, tau, fromParticle, toParticle, NULL, effvisc);
	if (compute_object_forces || p->boundarytype == ORC_DYN_BOUNDARY)
		forces_pass(p, PT_BOUNDARY, PT_FLUID, forces, pos, vel, info, hash, cellStart, neibsList, tau, fromParticle, toParticle, NULL, effvisc);
	finalize_forces(p, forces, cfl, rbforces, rbtorques, pos, vel, info, hash,
		fromParticle, toParticle, numBlocks, cflOffset, NULL, NULL, effvisc);
	return numBlocks;
}

uint32_t orc_forces(const orc_params *p, orc_f4 *forces, float *cfl,
	orc_f4 *rbforces, orc_f4 *rbtorques,
	const orc_f4 *pos, const orc_f4 *vel, const orc_info *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *tau,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle,
	uint32_t cflOffset, int compute_object_forces)
{
	return orc_forces_effvisc(p, forces, cfl, rbforces, rbtorques, pos, vel, info, hash, cellStart, neibsList, tau,
		numParticles, fromParticle, toParticle, cflOffset, compute_object_forces, NULL);
}

/* run_forces with SA_BOUNDARY (solid walls, no k-epsilon, no moving bodies): fluid <- fluid, fluid <- vertex, then
 * fluid <- boundary element (the launch order of src/cuda/forces.cu:751-790; vertex particles skip their own neighbour
 * walk, skip_neiblist :1346-1358), then the finalize with the division by gamma */
static uint32_t forces_sa_impl(const orc_params *p, orc_f4 *forces, float *cfl, float *cflGamma,
	const orc_f4 *pos, const orc_f4 *vel, const orc_info *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	const orc_f4 *gGam, const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset, float deltap, const orc_f4 *ioEulerVel)
{
	/* cflGamma: BUFFER_CFL_GAMMA in the reference's layout -- one value per particle, then, from round_up(numParticles, 4)
	 * on, one value per block (src/cuda/forces.cu:576-581); used with dynamic gamma and ENABLE_DTADAPT only */
	const int gcfl = cflGamma && !(p->simflags & ORC_ENABLE_GAMMA_QUADRATURE) && (p->simflags & ORC_ENABLE_DTADAPT);
	if (gcfl) memset(cflGamma + fromParticle, 0, sizeof(float)*(toParticle - fromParticle));
	const sa_forces_ctx sa = { gGam, boundelem, { vertPos0, vertPos1, vertPos2 }, deltap, gcfl ? cflGamma : NULL, NULL, ioEulerVel };
	const uint32_t numBlocks = round_up(div_up(toParticle - fromParticle, BLOCK_SIZE_FORCES), 4u);
	forces_pass(p, PT_FLUID, PT_FLUID, forces, pos, vel, info, hash, cellStart, neibsList, NULL, fromParticle, toParticle, &sa, NULL);
	forces_pass(p, PT_FLUID, PT_VERTEX, forces, pos, vel, info, hash, cellStart, neibsList, NULL, fromParticle, toParticle, &sa, NULL);
	forces_pass(p, PT_FLUID, PT_BOUNDARY, forces, pos, vel, info, hash, cellStart, neibsList, NULL, fromParticle, toParticle, &sa, NULL);
	finalize_forces(p, forces, cfl, NULL, NULL, pos, vel, info, hash, fromParticle, toParticle, numBlocks, cflOffset, &sa, NULL, NULL);
	if (gcfl) {      /* per-block maxima behind the per-particle values */
		float *blocks = cflGamma + round_up(numParticles, 4u) + cflOffset;
		for (uint32_t b = 0; b < numBlocks; ++b) {
			float m = 0.0f;
			for (uint32_t t = 0; t < BLOCK_SIZE_FORCES; ++t) {
				const uint32_t i = b*BLOCK_SIZE_FORCES + t + fromParticle;
				if (i < toParticle) m = fmaxf(m, cflGamma[i]);
			}
			blocks[b] = m;
		}
	}
	return numBlocks;
}
uint32_t orc_forces_sa(const orc_params *p, orc_f4 *forces, float *cfl, float *cflGamma,
	const orc_f4 *pos, const orc_f4 *vel, const orc_info *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	const orc_f4 *gGam, const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset, float deltap)
{
	return forces_sa_impl(p, forces, cfl, cflGamma, pos, vel, info, hash, cellStart, neibsList, gGam, boundelem, vertPos0, vertPos1,
		vertPos2, numParticles, fromParticle, toParticle, cflOffset, deltap, NULL);
}
/* the same three launches with ENABLE_INLET_OUTLET (laminar): the viscous terms see the Eulerian velocity of the open boundaries'
 * vertices and segments (get_viscous_relVel :2494-2507; against an open segment the whole relative velocity, :2703-2708), the
 * gamma CFL condition their normal velocity (:1485-1497).  GROUNDWORK ("Open boundaries" at the end of the file).  The forces
 * pass of the pressure-driven open vertices (skip_neiblist :1375-1389) leaves the water depth only: orc_sa_io_water_depth */
uint32_t orc_forces_sa_io(const orc_params *p, orc_f4 *forces, float *cfl, float *cflGamma,
	const orc_f4 *pos, const orc_f4 *vel, const orc_f4 *eulerVel, const orc_info *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	const orc_f4 *gGam, const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset, float deltap)
{
	return forces_sa_impl(p, forces, cfl, cflGamma, pos, vel, info, hash, cellStart, neibsList, gGam, boundelem, vertPos0, vertPos1,
		vertPos2, numParticles, fromParticle, toParticle, cflOffset, deltap, eulerVel);
}

/* run_forces with SA_BOUNDARY and the k-epsilon model (solid walls): the three fluid launches of orc_forces_sa with the keps members,
 * then forcesDevice<PT_VERTEX, PT_FLUID> (vertex_forces, src/cuda/forces.cu:676-686), whose viscous term is computed into nout and
 * never added to the vertex's force (compute_pp_interaction :3750-3783) -- what it leaves is a cleared DKDE / TAU row for every
 * vertex particle -- then the finalize with viscous_fixup.  dkde: 3 floats per particle, strain: 6 (BUFFER_TAU) */
uint32_t orc_forces_sa_keps(const orc_params *p, orc_f4 *forces, float *cfl, float *cflGamma, float *cflKeps,
	float *dkde, float *strain,
	const orc_f4 *pos, const orc_f4 *vel, const orc_info *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList,
	const orc_f4 *gGam, const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	const float *tke, const float *eps, const float *turbvisc, const orc_f4 *eulerVel,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset, float deltap, float epsilon)
{
	const int gcfl = cflGamma && !(p->simflags & ORC_ENABLE_GAMMA_QUADRATURE) && (p->simflags & ORC_ENABLE_DTADAPT);
	if (gcfl) memset(cflGamma + fromParticle, 0, sizeof(float)*(toParticle - fromParticle));
	const sa_keps_ctx ke = { tke, eps, turbvisc, eulerVel, dkde, strain, cflKeps, epsilon };
	const sa_forces_ctx sa = { gGam, boundelem, { vertPos0, vertPos1, vertPos2 }, deltap, gcfl ? cflGamma : NULL, &ke, NULL };
	const uint32_t numBlocks = round_up(div_up(toParticle - fromParticle, BLOCK_SIZE_FORCES), 4u);
	forces_pass(p, PT_FLUID, PT_FLUID, forces, pos, vel, info, hash, cellStart, neibsList, NULL, fromParticle, toParticle, &sa, NULL);
	forces_pass(p, PT_FLUID, PT_VERTEX, forces, pos, vel, info, hash, cellStart, neibsList, NULL, fromParticle, toParticle, &sa, NULL);
	forces_pass(p, PT_FLUID, PT_BOUNDARY, forces, pos, vel, info, hash, cellStart, neibsList, NULL, fromParticle, toParticle, &sa, NULL);
	for (uint32_t i = fromParticle; i < toParticle; ++i)      /* forcesDevice<PT_VERTEX, PT_FLUID>: write_keps of a fresh output */
		if (VERTEX(info[i]) && !INACTIVE(pos[i])) {
			dkde[3*(size_t)i] = 0.0f; dkde[3*(size_t)i + 1] = 0.0f; dkde[3*(size_t)i + 2] = 1.92f;
			memset(strain + 6*(size_t)i, 0, 6*sizeof(float));
		}
	finalize_forces(p, forces, cfl, NULL, NULL, pos, vel, info, hash, fromParticle, toParticle, numBlocks, cflOffset, &sa, NULL, NULL);
	if (gcfl) {
		float *blocks = cflGamma + round_up(numParticles, 4u) + cflOffset;
		for (uint32_t b = 0; b < numBlocks; ++b) {
			float m = 0.0f;
			for (uint32_t t = 0; t < BLOCK_SIZE_FORCES; ++t) {
				const uint32_t i = b*BLOCK_SIZE_FORCES + t + fromParticle;
				if (i < toParticle) m = fmaxf(m, cflGamma[i]);
			}
			blocks[b] = m;
		}
	}
	return numBlocks;
}

/* ==== SPH_GRENIER (multi-fluid volume formulation; the options of the reference's Bubble, LockExchange, RTInstability and
 * OilJet problems: formulation<SPH_GRENIER>, viscosity<DYNAMICVISC>, boundary<DYN_BOUNDARY>).
 *   - densityGrenierDevice (src/cuda/forces_kernel.cu:284-398), run before each forces pass (COMPUTE_DENSITY,
 *     src/integrators/PredictorCorrectorIntegrator.cc:443-458): sigma_a = sum_b W_ab over every neighbour, the density
 *     rho_a = (sum m_b W_ab / sum W_ab)/omega_a over the neighbours of the same type and fluid, written into vel.w in place;
 *   - forces: the continuity equation gives D(log J)/Dt = -1/sigma_a sum_b (v_ab . r_ab) F_ab (mass_continuity_div_vel_term
 *     :2018-2028), the momentum equation -1/rho_a sum_b (P_a/sigma_a + P_b/sigma_b [+ eps (|.|+|.|) across an interface]) F_ab r_ab
 *     (precalc_pressure :445-455, apply_pseudo_surface_tension :2226-2238, compute_pressure_contrib :2383-2392) and the
 *     Morris viscous term with avg(mu_a, mu_b) (1/sigma_a + 1/sigma_b) F_ab v_ab (:2628-2646); the divisions by rho_a and
 *     sigma_a are the fixup of the finalize kernel (:3181-3190);
 *   - Euler integrates vol.y = log(omega/omega_0) and writes omega = exp(vol.y) omega_0 (euler_body above);
 *   - ProblemCore::init_volume (src/ProblemCore.cc:1586-1606). */
void orc_init_volume(const orc_params *p, orc_f4 *vol, const orc_f4 *pos, const orc_f4 *vel, const orc_info *info, uint32_t numParticles)
{
	for (uint32_t i = 0; i < numParticles; ++i) {
		orc_f4 v;
		v.x = v.w = pos[i].w/physical_density(p, vel[i].w, FLUID_NUM(info[i]));
		v.y = 0; v.z = 0;
		vol[i] = v;
	}
}

void orc_density_grenier(const orc_params *p, float *sigmaArray, orc_f4 *velArray,
	const orc_f4 *posArray, const orc_info *infoArray, const uint32_t *hashArray, const orc_f4 *volArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, int maxFluidBoundaryNeibs)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const int dyn = p->boundarytype == ORC_DYN_BOUNDARY;
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		if (!dyn && !FLUID(info)) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		const int fnum = FLUID_NUM(info);
		const float vol = volArray[index].w;
		orc_f4 vel = velArray[index];
		float corr = W_c(p->kerneltype, 0.0f, p->slength, wcoeff, wsub);   /* self contribution */
		float sigma = corr;
		float mass_corr = pos.w*corr;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		int has_fluid_neibs = 0;
		const int last = dyn ? PT_BOUNDARY : PT_FLUID;
		for (int ptype = PT_FLUID; ptype <= last; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				const orc_info neib_info = infoArray[neib_index];
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);
				if (!isfinite(npos.w) || r >= p->influenceradius) continue;
				const float w = W_c(p->kerneltype, r, p->slength, wcoeff, wsub);
				sigma += w;
				if (FLUID(neib_info)) has_fluid_neibs = 1;
				if ((!dyn || PART_TYPE(neib_info) == PART_TYPE(info)) && FLUID_NUM(neib_info) == fnum) {
					mass_corr += npos.w*w;
					corr += w;
				}
			}
		}
		if (dyn && !FLUID(info) && !has_fluid_neibs) {
			/* 'typical' specific volume: the largest fluid + boundary neighbour count of the last list build over the
			 * volume of the influence sphere, 3*int/(4*M_PIf*R*R*R) */
			const float R = p->influenceradius;
			sigma = (float)(3*maxFluidBoundaryNeibs)/(4*3.14159265358979323846f*R*R*R);
		}
		vel.w = mass_corr/(corr*vol);
		vel.w = numerical_density(p, vel.w, fnum);
		velArray[index] = vel;
		sigmaArray[index] = sigma;
	}
}

static void grenier_pass(const orc_params *p, int cptype, int nptype, orc_f4 *forces,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *sigmaArray,
	uint32_t fromParticle, uint32_t toParticle)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);
	const int dyn = p->boundarytype == ORC_DYN_BOUNDARY;
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = fromParticle; index < toParticle; ++index) {
		const orc_info info = infoArray[index];
		if (PART_TYPE(info) != cptype) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const orc_f4 vel = velArray[index];
		const int p_fluid = FLUID_NUM(info);
		const float p_rho = physical_density(p, vel.w, p_fluid);
		const float p_sigma = sigmaArray[index];
		const float p_precalc = orc_P(p, vel.w, p_fluid)/p_sigma;
		orc_f4 force = forces[index];
		neib_iter it;
		neib_iter_init(&it, p, nptype, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			if (!isfinite(npos.w)) continue;
			const float r = sqrtf(sqlength3(rx, ry, rz));
			if (r >= p->influenceradius) continue;
			const orc_info neib_info = infoArray[neib_index];
			const orc_f4 nvel = velArray[neib_index];
			const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
			const float vel_dot_pos = dot3(vx, vy, vz, rx, ry, rz);
			const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
			const int n_fluid = FLUID_NUM(neib_info);
			const float n_rho = physical_density(p, nvel.w, n_fluid);
			const float n_sigma = sigmaArray[neib_index];
			const float n_precalc = orc_P(p, nvel.w, n_fluid)/n_sigma;
			const int all_pp = (cptype == PT_FLUID && nptype == PT_FLUID) || (cptype == PT_FLUID && nptype == PT_BOUNDARY && dyn);
			const int dyn_bf = (cptype == PT_BOUNDARY && nptype == PT_FLUID && dyn);
			if (!all_pp && !dyn_bf) continue;
			float DrDt = 0.0f;
			DrDt -= vel_dot_pos*f;
			force.w += DrDt;
			if (all_pp || COMPUTE_FORCE(info)) {
				float pGradTerm = p_precalc + n_precalc;
				if (cptype == PT_FLUID && nptype == PT_FLUID && p_fluid != n_fluid)
					pGradTerm += p->epsinterface*(fabsf(p_precalc) + fabsf(n_precalc));
				const float s = pGradTerm*f;
				float DvDt[3] = { 0.0f, 0.0f, 0.0f };
				DvDt[0] -= s*rx; DvDt[1] -= s*ry; DvDt[2] -= s*rz;
				if (p->rheologytype == ORC_NEWTONIAN) {
					const float our_mu = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[p_fluid]*p_rho : p->visccoeff[p_fluid];
					const float neib_mu = (p->compvisc == ORC_KINEMATIC) ? p->visccoeff[n_fluid]*n_rho : p->visccoeff[n_fluid];
					float avg_mu;
					switch (p->avgop) {           /* average<>, src/average.h:78-100 */
					case ORC_ARITHMETIC: avg_mu = (our_mu + neib_mu)*0.5f; break;
					case ORC_HARMONIC:   avg_mu = 2*our_mu*neib_mu/(our_mu + neib_mu); break;
					default:             avg_mu = sqrtf(our_mu*neib_mu);
					}
					const float avg_sigma = 1/p_sigma + 1/n_sigma;
					const float c = avg_mu*avg_sigma*f;
					DvDt[0] += c*vx; DvDt[1] += c*vy; DvDt[2] += c*vz;
				}
				force.x += DvDt[0]; force.y += DvDt[1]; force.z += DvDt[2];
			}
		}
		forces[index] = force;
	}
}

/* run_forces with SPH_GRENIER + DYN_BOUNDARY: fluid <- fluid, fluid <- boundary, boundary <- fluid, finalize */
uint32_t orc_forces_grenier(const orc_params *p, orc_f4 *forces, float *cfl,
	const orc_f4 *pos, const orc_f4 *vel, const orc_info *info, const uint32_t *hash,
	const uint32_t *cellStart, const uint16_t *neibsList, const float *sigma,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset)
{
	(void)numParticles;
	const uint32_t numBlocks = round_up(div_up(toParticle - fromParticle, BLOCK_SIZE_FORCES), 4u);
	grenier_pass(p, PT_FLUID, PT_FLUID, forces, pos, vel, info, hash, cellStart, neibsList, sigma, fromParticle, toParticle);
	grenier_pass(p, PT_FLUID, PT_BOUNDARY, forces, pos, vel, info, hash, cellStart, neibsList, sigma, fromParticle, toParticle);
	if (p->boundarytype == ORC_DYN_BOUNDARY)
		grenier_pass(p, PT_BOUNDARY, PT_FLUID, forces, pos, vel, info, hash, cellStart, neibsList, sigma, fromParticle, toParticle);
	finalize_forces(p, forces, cfl, NULL, NULL, pos, vel, info, hash, fromParticle, toParticle, numBlocks, cflOffset, NULL, sigma, NULL);
	return numBlocks;
}

/* ==== repacking: run_repack src/cuda/forces.cu:828-896, repackDevice forces_kernel.def:4155-4262,
 * compute_repacking_contrib (non-SA: fluid and boundary neighbours alike) :3024-3055, finalizeRepackDevice
 * :4263-4349, repack_fixup :3238-3244.  Only fluid particles receive the mixing force
 *   F_a = -a c0^2 sum_b (m_b/rho_b) F(r_ab) r_ab ,
 * finalize adds the term alpha c0/deltap v_a with the reference's sign, the plane repulsion, and the CFL term. */
static void repack_pass(const orc_params *p, int nptype, orc_f4 *forces,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t fromParticle, uint32_t toParticle)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = fromParticle; index < toParticle; ++index) {
		const orc_info info = infoArray[index];
		if (PART_TYPE(info) != PT_FLUID) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const int fl = FLUID_NUM(info);
		orc_f4 force = forces[index];
		neib_iter it;
		neib_iter_init(&it, p, nptype, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			if (!isfinite(npos.w)) continue;
			const float r = sqrtf(sqlength3(rx, ry, rz));
			if (r >= p->influenceradius) continue;
			const float n_rho = physical_density(p, velArray[neib_index].w, FLUID_NUM(infoArray[neib_index]));
			const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
			/* fluid and (non-SA) boundary neighbours: a c0^2 V_b F r (:3024-3055); vertex neighbours of SA_BOUNDARY: the reference
			 * has a single c0 there (:3057-3072), reproduced */
			const float s = (nptype == PT_VERTEX) ? p->repack_a*p->sscoeff[fl]*npos.w/n_rho*f :
				p->repack_a*p->sscoeff[fl]*p->sscoeff[fl]*npos.w/n_rho*f;
			force.x -= s*rx; force.y -= s*ry; force.z -= s*rz;
		}
		forces[index] = force;
	}
}

static void repack_finalize(const orc_params *p, orc_f4 *forces, float *cfl, orc_f4 *rbforces, orc_f4 *rbtorques,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	uint32_t fromParticle, uint32_t toParticle, uint32_t numBlocks, uint32_t cflOffset, const orc_f4 *gGam);

uint32_t orc_repack_forces(const orc_params *p, orc_f4 *forces, float *cfl,
	orc_f4 *rbforces, orc_f4 *rbtorques,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset)
{
	(void)numParticles;
	const uint32_t numBlocks = round_up(div_up(toParticle - fromParticle, BLOCK_SIZE_FORCES), 4u);
	repack_pass(p, PT_FLUID, forces, posArray, velArray, infoArray, hashArray, cellStart, neibsList, fromParticle, toParticle);
	repack_pass(p, PT_BOUNDARY, forces, posArray, velArray, infoArray, hashArray, cellStart, neibsList, fromParticle, toParticle);
	repack_finalize(p, forces, cfl, rbforces, rbtorques, posArray, velArray, infoArray, hashArray, fromParticle, toParticle, numBlocks, cflOffset, NULL);
	return numBlocks;
}

/* finalizeRepackDevice (:4263-4349); gGam: SA_BOUNDARY, the sums of the fluid particles are divided by gamma (repack_fixup :3220-3236) */
static void repack_finalize(const orc_params *p, orc_f4 *forces, float *cfl, orc_f4 *rbforces, orc_f4 *rbtorques,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	uint32_t fromParticle, uint32_t toParticle, uint32_t numBlocks, uint32_t cflOffset, const orc_f4 *gGam)
{
	const int dtadapt = !!(p->simflags & ORC_ENABLE_DTADAPT);
#pragma omp parallel for schedule(static)
	for (uint32_t block = 0; block < numBlocks; ++block) {
		float block_max = 0.0f;
		for (uint32_t t = 0; t < BLOCK_SIZE_FORCES; ++t) {
			const uint32_t index = block*BLOCK_SIZE_FORCES + t + fromParticle;
			if (index >= toParticle) break;
			const orc_info info = infoArray[index];
			const orc_f4 pos = posArray[index];
			if (INACTIVE(pos)) continue;
			const orc_f4 vel = velArray[index];
			orc_f4 force = forces[index];
			const int fl = FLUID_NUM(info);
			if (gGam) {
				if (FLUID(info)) {
					const float gam = gGam[index].w;
					force.x /= gam; force.y /= gam; force.z /= gam; force.w /= gam;
					force.w /= p->rho0[fl];
				}
			} else
				force.w /= p->rho0[fl];
			if (FLUID(info)) {
				const float damp = p->repack_alpha*p->sscoeff[fl]/p->deltap;
				force.x += damp*vel.x; force.y += damp*vel.y; force.z += damp*vel.z;
				/* planes: dynvisc = d_visccoeff*rho whatever the computational viscosity (:4314); the reference leaves
				 * d_visccoeff NaN for inviscid problems (GPUSPH.cc:1488-1493) -- taken as 0 (free slip) here */
				const float dynvisc = (p->rheologytype == ORC_NEWTONIAN) ? p->visccoeff[fl]*physical_density(p, vel.w, fl) : 0.0f;
				if ((p->simflags & ORC_ENABLE_PLANES) && p->numplanes) {
					int gp[3];
					orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gp);
					for (uint32_t k = 0; k < p->numplanes; ++k) {
						const float dx = (gp[0] - p->plane_gridpos[k][0])*p->cellSize[0] + (pos.x - p->plane_pos[k][0]);
						const float dy = (gp[1] - p->plane_gridpos[k][1])*p->cellSize[1] + (pos.y - p->plane_pos[k][1]);
						const float dz = (gp[2] - p->plane_gridpos[k][2])*p->cellSize[2] + (pos.z - p->plane_pos[k][2]);
						const float *nrm = p->plane_normal[k];
						const float r = fabsf(dx*nrm[0] + dy*nrm[1] + dz*nrm[2]);
						if (r < p->r0) {
							const float DvDt = p->dcoeff*(powf(p->r0/r, p->p1coeff) - powf(p->r0/r, p->p2coeff))/(r*r);
							const float rp[3] = { nrm[0]*r, nrm[1]*r, nrm[2]*r };
							force.x += DvDt*rp[0]; force.y += DvDt*rp[1]; force.z += DvDt*rp[2];
							if (dynvisc != 0.0f)
								plane_friction(p, &force, &vel, rp, r, pos.w, dynvisc);
						}
					}
				}
				if (dtadapt) {
					const float sspeed = orc_soundSpeed(p, vel.w, fl);
					const float a = sqrtf(sqlength3(force.x, force.y, force.z));
					block_max = fmaxf(block_max, fmaxf(a, sspeed*sspeed/p->slength));
				}
			}
			if (COMPUTE_FORCE(info) && !VERTEX(info) && rbforces) {
				const uint32_t rbindex = (uint32_t)((int)orc_info_id(info) + p->rbstartindex[OBJECT_NUM(info)]);
				const orc_f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
				rbforces[rbindex] = zero; rbtorques[rbindex] = zero;
			}
			forces[index] = force;
		}
		if (dtadapt && cfl)
			cfl[cflOffset + block] = block_max;
	}
}

/* run_repack with SA_BOUNDARY (src/cuda/forces.cu:828-896): fluid <- fluid, fluid <- vertex, fluid <- boundary element
 * (+ a c0^2 |grad gamma_as| n_s, :3074-3086), finalize with the division by gamma */
uint32_t orc_repack_forces_sa(const orc_params *p, orc_f4 *forces, float *cfl,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList,
	const orc_f4 *gGam, const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	uint32_t numParticles, uint32_t fromParticle, uint32_t toParticle, uint32_t cflOffset, float deltap)
{
	(void)numParticles;
	const uint32_t numBlocks = round_up(div_up(toParticle - fromParticle, BLOCK_SIZE_FORCES), 4u);
	repack_pass(p, PT_FLUID, forces, posArray, velArray, infoArray, hashArray, cellStart, neibsList, fromParticle, toParticle);
	repack_pass(p, PT_VERTEX, forces, posArray, velArray, infoArray, hashArray, cellStart, neibsList, fromParticle, toParticle);
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = fromParticle; index < toParticle; ++index) {
		const orc_info info = infoArray[index];
		if (PART_TYPE(info) != PT_FLUID) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const int fl = FLUID_NUM(info);
		orc_f4 force = forces[index];
		neib_iter it;
		neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
		uint32_t j;
		while ((j = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[j];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			if (!isfinite(npos.w)) continue;
			const float r = sqrtf(sqlength3(rx, ry, rz));
			if (r >= p->influenceradius + deltap) continue;
			const orc_f4 belem = boundelem[j];
			const float inv_h = 1.0f/p->slength;
			const float ggamAS = orc_grad_gamma_vp(p->slength, rx*inv_h, ry*inv_h, rz*inv_h, &belem,
				vertPos0 + 2*(size_t)j, vertPos1 + 2*(size_t)j, vertPos2 + 2*(size_t)j);
			const float c = p->repack_a*p->sscoeff[fl]*p->sscoeff[fl]*ggamAS;
			force.x += c*belem.x; force.y += c*belem.y; force.z += c*belem.z;
		}
		forces[index] = force;
	}
	repack_finalize(p, forces, cfl, NULL, NULL, posArray, velArray, infoArray, hashArray, fromParticle, toParticle, numBlocks, cflOffset, gGam);
	return numBlocks;
}

/* cflmax + dtreduce: src/cuda/forces.cu:150-176,556-606 ; fmaxDevice forces_kernel.cu:734-793 */
float orc_dtreduce(const orc_params *p, const float *cfl, uint32_t numBlocks,
	float sspeed_cfl, float max_kinematic)
{
	float maxcfl = 0.0f;
	for (uint32_t i = 0; i < numBlocks; ++i) maxcfl = fmaxf(maxcfl, cfl[i]);
	float dt = p->dtadaptfactor*fminf(sqrtf(p->slength/maxcfl), p->slength/sspeed_cfl);
	if (p->rheologytype != ORC_INVISCID || p->turbmodel > ORC_ARTIFICIAL) {
		float visccoeff = max_kinematic;
		float dt_visc = p->slength*p->slength/visccoeff;
		dt_visc *= 0.125;
		if (dt_visc < dt) dt = dt_visc;
	}
	return dt;
}

/* ---- SPS: SPSstressMatrixDevice src/cuda/visc_kernel.cu:759-811, shearRate<MIXED_TENSOR> :307-367,
 *      shear_rate_contrib (non-SA) :207-215, shearRateNorm2<MIXED_TENSOR> :385-407 ---------------- */
void orc_sps(const orc_params *p, float *tau, float *turbvisc,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, uint32_t particleRangeEnd)
{
	(void)numParticles;
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		const orc_f4 vel = velArray[index];
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);

		float dvx[3] = {0,0,0}, dvy[3] = {0,0,0}, dvz[3] = {0,0,0};
		/* for_every_neib: PT_FLUID then PT_BOUNDARY (non-SA), neibs_iteration.cuh:325-343 */
		for (int ptype = PT_FLUID; ptype <= PT_BOUNDARY; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				const float r = sqrtf(sqlength3(rx, ry, rz));
				if (!isfinite(npos.w) || r >= p->influenceradius) continue;
				const orc_f4 nvel = velArray[neib_index];
				const orc_info ninfo = infoArray[neib_index];
				const float n_rho = physical_density(p, nvel.w, FLUID_NUM(ninfo));
				const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
				const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
				const float weight = f*npos.w/n_rho;
				const float mx = rx*weight, my = ry*weight, mz = rz*weight;
				dvx[0] -= vx*mx; dvx[1] -= vx*my; dvx[2] -= vx*mz;
				dvy[0] -= vy*mx; dvy[1] -= vy*my; dvy[2] -= vy*mz;
				dvz[0] -= vz*mx; dvz[1] -= vz*my; dvz[2] -= vz*mz;
			}
		}
		/* mixed tensor: non-doubled diagonal, doubled off-diagonal */
		float txx = dvx[0], txy = dvx[1] + dvy[0], txz = dvx[2] + dvz[0];
		float tyy = dvy[1], tyz = dvy[2] + dvz[1], tzz = dvz[2];
		float diag_terms = txx*txx + tyy*tyy + tzz*tzz;
		diag_terms *= 2.0f;
		const float off_terms = txy*txy + txz*txz + tyz*tyz;
		const float SijSij_bytwo = diag_terms + off_terms;
		const float S = sqrtf(SijSij_bytwo);
		const float nu_SPS = p->smagfactor*S;
		const float divu_SPS = 0.6666666666f*nu_SPS*(txx + tyy + tzz);
		const float Blinetal_SPS = p->kspsfactor*SijSij_bytwo;
		if (turbvisc) turbvisc[index] = nu_SPS;
		const float rho = physical_density(p, vel.w, FLUID_NUM(info));
		txx = nu_SPS*(txx + txx) - divu_SPS - Blinetal_SPS; txx /= rho;
		txy *= nu_SPS/rho;
		txz *= nu_SPS/rho;
		tyy = nu_SPS*(tyy + tyy) - divu_SPS - Blinetal_SPS; tyy /= rho;
		tyz *= nu_SPS/rho;
		tzz = nu_SPS*(tzz + tzz) - divu_SPS - Blinetal_SPS; tzz /= rho;
		float *t = tau + 6*(size_t)index;
		t[0] = txx; t[1] = txy; t[2] = txz; t[3] = tyy; t[4] = tyz; t[5] = tzz;
	}
}

/* ---- eulerDevice<step>: src/cuda/euler_kernel.def:396-538 --------------------------- */
/* ==== generalized Newtonian rheologies (BINGHAM .. ZHU): effectiveViscDevice (src/cuda/visc_kernel.cu:655-713) =========
 * Per particle (every type, non-SA): the shear rate norm S = sqrt(D:D/2) from the velocity gradient over all neighbours
 * (shearRate<MIXED_TENSOR> :307-367, shearRateNorm2 :383-407), then
 *   mu_eff = shear term (k, k S^(n-1) or k exp(-t1 S); viscShearTerm :501-531) + yield term (tau_0/S, or regularised
 *            tau_0 (1 - exp(-m S))/S with an 8th-order Horner form below m S = 1; viscYieldTerm :454-497),
 *   clamped to limiting_kinvisc rho0 (clamp_visc :561-569); stored as it is (compvisc DYNAMIC) or divided by the density
 *   (KINEMATIC) (store_effective_visc :605-620).  Returns the largest kinematic viscosity (the per-block reduction into the
 *   CFL array + cflmax of src/cuda/visc.cu:86-170), which GPUWorker hands to dtreduce (src/GPUWorker.cc:2633-2645,2013-2030). */
static float horner_one_minus_exp_minus_over8(float x)
{
	/* horner_one_minus_exp_minus_over<8> :420-451: (1 - x/2 (1 - x/3 (... (1 - x/9)))) */
	float inner = fmaf(x, -1.0f/(8 + 1.0f), 1.0f);
	for (int order = 7; order >= 2; --order)
		inner = fmaf(x*inner, -1.0f/(order + 1.0f), 1.0f);
	return fmaf(x*inner, -0.5f, 1.0f);
}

float orc_effective_visc_value(const orc_params *p, float S, int fluid)
{
	const int rh = p->rheologytype;
	float effvisc = 0.0f;
	if (p->visccoeff[fluid] != 0.0f) {
		if (rh >= ORC_DEKEE_TURCOTTE) effvisc += p->visccoeff[fluid]*expf(-p->visc_nonlinear_param[fluid]*S);
		else if (rh >= ORC_POWER_LAW) effvisc += p->visccoeff[fluid]*powf(S, p->visc_nonlinear_param[fluid] - 1);
		else effvisc += p->visccoeff[fluid];
	}
	if (p->yield_strength[fluid] != 0.0f) {
		const int reg = rh == ORC_PAPANASTASIOU || rh == ORC_ALEXANDROU || rh == ORC_ZHU;
		const int yielding = rh > ORC_NEWTONIAN && rh != ORC_POWER_LAW && rh != ORC_GRANULAR;
		if (reg) {
			const float m = p->visc_regularization_param[fluid];
			const float mx = m*S;
			float r;
			if (mx < 1) r = m*horner_one_minus_exp_minus_over8(mx);
			else r = (1 - expf(-mx))/S;
			effvisc += p->yield_strength[fluid]*r;
		} else if (yielding)
			effvisc += p->yield_strength[fluid]/S;
	}
	return fminf(effvisc, p->limiting_kinvisc*p->rho0[fluid]);
}

float orc_effective_visc(const orc_params *p, float *effviscArray, float *cfl,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t particleRangeEnd)
{
	(void)numParticles;
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);
	const uint32_t numBlocks = round_up(div_up(particleRangeEnd, BLOCK_SIZE_FORCES), 4u);     /* BLOCK_SIZE_SPS = 128 as well */
	float *kin = (float*)calloc(particleRangeEnd ? particleRangeEnd : 1, sizeof(float));
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		const orc_f4 vel = velArray[index];
		const int fluid = FLUID_NUM(info);
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		float dvx[3] = {0,0,0}, dvy[3] = {0,0,0}, dvz[3] = {0,0,0};
		for (int ptype = PT_FLUID; ptype <= PT_BOUNDARY; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				const float r = sqrtf(sqlength3(rx, ry, rz));
				if (!isfinite(npos.w) || r >= p->influenceradius) continue;
				const orc_f4 nvel = velArray[neib_index];
				const float n_rho = physical_density(p, nvel.w, FLUID_NUM(infoArray[neib_index]));
				const float vx = vel.x - nvel.x, vy = vel.y - nvel.y, vz = vel.z - nvel.z;
				const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
				const float weight = f*npos.w/n_rho;
				const float mx = rx*weight, my = ry*weight, mz = rz*weight;
				dvx[0] -= vx*mx; dvx[1] -= vx*my; dvx[2] -= vx*mz;
				dvy[0] -= vy*mx; dvy[1] -= vy*my; dvy[2] -= vy*mz;
				dvz[0] -= vz*mx; dvz[1] -= vz*my; dvz[2] -= vz*mz;
			}
		}
		const float txx = dvx[0], txy = dvx[1] + dvy[0], txz = dvx[2] + dvz[0];
		const float tyy = dvy[1], tyz = dvy[2] + dvz[1], tzz = dvz[2];
		float diag_terms = txx*txx + tyy*tyy + tzz*tzz;
		diag_terms *= 2.0f;
		const float off_terms = txy*txy + txz*txz + tyz*tyz;
		const float S = sqrtf(diag_terms + off_terms);
		const float effvisc = orc_effective_visc_value(p, S, fluid);
		const float kinvisc = effvisc/physical_density(p, vel.w, fluid);
		effviscArray[index] = (p->compvisc == ORC_KINEMATIC) ? kinvisc : effvisc;
		kin[index] = kinvisc;
	}
	float mx = 0.0f;
	for (uint32_t b = 0; b < numBlocks; ++b) {
		float m = 0.0f;
		for (uint32_t t = 0; t < BLOCK_SIZE_FORCES; ++t) {
			const uint32_t i = b*BLOCK_SIZE_FORCES + t;
			if (i < particleRangeEnd) m = fmaxf(m, kin[i]);
		}
		if (cfl) cfl[b] = m;
		mx = fmaxf(mx, m);
	}
	free(kin);
	return (p->simflags & ORC_ENABLE_DTADAPT) ? mx : NAN;
}

static void euler_body(const orc_params *p, orc_f4 *newPos, orc_f4 *newVel,
	const orc_f4 *oldPos, const orc_f4 *oldVel, const orc_info *infoArray, const uint32_t *hashArray,
	const orc_f4 *forces, const orc_f4 *xsph,
	uint32_t numParticles, float dt, int step, int repacking, orc_f4 *newVol, const orc_f4 *oldVol)
{
	/* SPH_GRENIER: the continuity equation integrates the log of the volume ratio (continuity_integration :210-216,
	 * grenier_particle_data :84-92, write_volume :281-289) */
	const int grenier = p->sph_formulation == ORC_SPH_GRENIER && !repacking && newVol && oldVol;
	/* euler_repack_params (src/cuda/euler_params.h:203): no boundary integration, no continuity, no XSPH, no body motion */
	const int integrateBoundary = !repacking &&
		(p->boundarytype == ORC_DYN_BOUNDARY || p->boundarytype == ORC_SA_BOUNDARY);
#pragma omp parallel for schedule(static)
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		const int ptype = PART_TYPE(info);
		const orc_f4 force = forces[index];
		orc_f4 pos = oldPos[index];
		orc_f4 vel = oldVel[index];
		orc_f4 vol = { 0, 0, 0, 0 };
		if (grenier) vol = oldVol[index];
		do {
			if (!ACTIVE(pos) || (ptype == PT_BOUNDARY && !integrateBoundary && !MOVING(info)))
				break;
			/* standard_corrected_velocity, :147-169: velc = vel (+ force*(dt/2) on step 2) */
			float velc[3] = { vel.x, vel.y, vel.z };
			if (step == 2) {
				const float hdt = dt/2;
				velc[0] = fmaf(force.x, hdt, velc[0]);
				velc[1] = fmaf(force.y, hdt, velc[1]);
				velc[2] = fmaf(force.z, hdt, velc[2]);
			}
			if ((p->simflags & ORC_ENABLE_XSPH) && xsph && !repacking) {
				velc[0] = fmaf(p->epsxsph, xsph[index].x, velc[0]);
				velc[1] = fmaf(p->epsxsph, xsph[index].y, velc[1]);
				velc[2] = fmaf(p->epsxsph, xsph[index].z, velc[2]);
			}
			const int obj = OBJECT_NUM(info);
			switch (ptype) {
			case PT_FLUID:
				pos.x = fmaf(velc[0], dt, pos.x);
				pos.y = fmaf(velc[1], dt, pos.y);
				pos.z = fmaf(velc[2], dt, pos.z);
				if (grenier)
					vol.y = fmaf(dt, force.w, vol.y);
				else if (!repacking)
					vel.w = fmaf(dt, force.w, vel.w); /* continuity_integration :203-209 */
				vel.x = fmaf(dt, force.x, vel.x);
				vel.y = fmaf(dt, force.y, vel.y);
				vel.z = fmaf(dt, force.z, vel.z);
				break;
			case PT_VERTEX:
			case PT_BOUNDARY:
				if (!repacking && MOVING(info)) {
					int gp[3];
					orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gp);
					const float rx = (gp[0] - p->rbcgGridPosE[obj][0])*p->cellSize[0] + (pos.x - p->rbcgPosE[obj][0]);
					const float ry = (gp[1] - p->rbcgGridPosE[obj][1])*p->cellSize[1] + (pos.y - p->rbcgPosE[obj][1]);
					const float rz = (gp[2] - p->rbcgGridPosE[obj][2])*p->cellSize[2] + (pos.z - p->rbcgPosE[obj][2]);
					const float *rot = p->rbsteprot[obj];
					/* applyrot, src/cuda/euler_kernel.cu:67-74 */
					pos.x += (rot[0] - 1.0f)*rx + rot[1]*ry + rot[2]*rz;
					pos.y += rot[3]*rx + (rot[4] - 1.0f)*ry + rot[5]*rz;
					pos.z += rot[6]*rx + rot[7]*ry + (rot[8] - 1.0f)*rz;
					pos.x += p->rbtrans[obj][0];
					pos.y += p->rbtrans[obj][1];
					pos.z += p->rbtrans[obj][2];
					const float *w = p->rbangularvel[obj];
					vel.x = p->rblinearvel[obj][0] + (w[1]*rz - w[2]*ry);
					vel.y = p->rblinearvel[obj][1] + (w[2]*rx - w[0]*rz);
					vel.z = p->rblinearvel[obj][2] + (w[0]*ry - w[1]*rx);
				}
				if (p->boundarytype == ORC_DYN_BOUNDARY) {
					if (grenier) vol.y = fmaf(dt, force.w, vol.y);
					else vel.w = fmaf(dt, force.w, vel.w);
				}
				break;
			default:
				break;
			}
		} while (0);
		newPos[index] = pos;
		newVel[index] = vel;
		if (grenier) {
			vol.w = expf(vol.y)*vol.x;
			newVol[index] = vol;
		}
	}
}

void orc_euler(const orc_params *p, orc_f4 *newPos, orc_f4 *newVel,
	const orc_f4 *oldPos, const orc_f4 *oldVel, const orc_info *infoArray, const uint32_t *hashArray,
	const orc_f4 *forces, const orc_f4 *xsph,
	uint32_t numParticles, float dt, int step)
{
	euler_body(p, newPos, newVel, oldPos, oldVel, infoArray, hashArray, forces, xsph, numParticles, dt, step, 0, NULL, NULL);
}

/* integrate_energy / write_energy of eulerDevice (euler_kernel.def:184-199,296-309): energy += dt DEDt for the particles the
 * kernel integrates (active fluid particles; boundary particles with DYN_BOUNDARY), copied for the others */
void orc_euler_energy(const orc_params *p, float *newEnergy, const float *oldEnergy, const float *DEDt,
	const orc_f4 *oldPos, const orc_info *infoArray, uint32_t numParticles, float dt)
{
	for (uint32_t i = 0; i < numParticles; ++i) {
		float e = oldEnergy[i];
		const int ptype = PART_TYPE(infoArray[i]);
		if (ACTIVE(oldPos[i]) && (ptype == PT_FLUID || ((ptype == PT_BOUNDARY || ptype == PT_VERTEX) && p->boundarytype == ORC_DYN_BOUNDARY)))
			e = fmaf(dt, DEDt[i], e);
		newEnergy[i] = e;
	}
}

/* eulerDevice with SPH_GRENIER: BUFFER_VOLUME is read (old) and written (new) next to pos and vel */
void orc_euler_grenier(const orc_params *p, orc_f4 *newPos, orc_f4 *newVel, orc_f4 *newVol,
	const orc_f4 *oldPos, const orc_f4 *oldVel, const orc_f4 *oldVol, const orc_info *infoArray, const uint32_t *hashArray,
	const orc_f4 *forces, const orc_f4 *xsph, uint32_t numParticles, float dt, int step)
{
	euler_body(p, newPos, newVel, oldPos, oldVel, infoArray, hashArray, forces, xsph, numParticles, dt, step, 0, newVol, oldVol);
}

/* run_mode == REPACK: eulerDevice with euler_repack_params (src/cuda/euler.cu:346-353) */
void orc_euler_repack(const orc_params *p, orc_f4 *newPos, orc_f4 *newVel,
	const orc_f4 *oldPos, const orc_f4 *oldVel, const orc_info *infoArray, const uint32_t *hashArray,
	const orc_f4 *forces, uint32_t numParticles, float dt, int step)
{
	euler_body(p, newPos, newVel, oldPos, oldVel, infoArray, hashArray, forces, NULL, numParticles, dt, step, 1, NULL, NULL);
}

/* disableFreeSurfPartsDevice (src/cuda/euler_kernel.cu:158-180): at the end of repacking the non-fluid particles
 * flagged FG_SURFACE (the lid that kept the free surface in place) are disabled */
void orc_disable_free_surf_parts(orc_f4 *pos, const orc_info *infoArray, uint32_t numParticles)
{
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		if (SURFACE(info) && !FLUID(info) && ACTIVE(pos[index]))
			pos[index].w = NAN; /* disable_particle */
	}
}

/* ==== density filters: shepardDevice src/cuda/forces_kernel.cu:418-505, MlsDevice :508-721 ==========
 * vector helpers follow src/vector_math.h: float4/float = float4*(1.0f/s) (:1093-1097), dot(float4,float4) (:1129-1132),
 * hypot(float4) (:1231-1240); tensor helpers src/cuda/tensor.cu:65-100 (det), :240-282 (dot, ddot, adjugate_row1). */

void orc_shepard(const orc_params *p, orc_f4 *newVel,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		orc_f4 vel = velArray[index];
		if (PART_TYPE(info) != PT_FLUID) { newVel[index] = vel; continue; }
		float temp1 = pos.w*W_c(p->kerneltype, 0.0f, p->slength, wcoeff, wsub);
		float temp2 = temp1/physical_density(p, vel.w, FLUID_NUM(info));
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		/* for_each_neib2(PT_FLUID, DYN ? PT_BOUNDARY : PT_NONE) */
		const int last = (p->boundarytype == ORC_DYN_BOUNDARY) ? PT_BOUNDARY : PT_FLUID;
		for (int ptype = PT_FLUID; ptype <= last; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				if (!isfinite(npos.w)) continue;
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);   /* length(as_float3(relPos)) */
				const float neib_rho = physical_density(p, velArray[neib_index].w, FLUID_NUM(infoArray[neib_index]));
				if (r < p->influenceradius) {
					const float w = W_c(p->kerneltype, r, p->slength, wcoeff, wsub)*npos.w;
					temp1 += w;
					temp2 += w/neib_rho;
				}
			}
		}
		vel.w = numerical_density(p, temp1/temp2, FLUID_NUM(info));
		newVel[index] = vel;
	}
}

typedef struct { float xx, xy, xz, xw, yy, yz, yw, zz, zw, ww; } symtensor4;
typedef struct { float x, y, z, w; } f4v;

static float st4_det(const symtensor4 *T)
{
	float ret = 0, M = 0;
	M += T->xx*(T->yy*T->zz - T->yz*T->yz);
	M -= T->xy*(T->xy*T->zz - T->xz*T->yz);
	M += T->xz*(T->xy*T->yz - T->xz*T->yy);
	ret += M*T->ww;
	M = 0;
	M += T->xx*(T->yy*T->zw - T->yz*T->yw);
	M -= T->xy*(T->xy*T->zw - T->xz*T->yw);
	M += T->xw*(T->xy*T->yz - T->xz*T->yy);
	ret -= M*T->zw;
	M = 0;
	M += T->xx*(T->yz*T->zw - T->zz*T->yw);
	M -= T->xz*(T->xy*T->zw - T->xz*T->yw);
	M += T->xw*(T->xy*T->zz - T->xz*T->yz);
	ret += M*T->yw;
	M = 0;
	M += T->xy*(T->yz*T->zw - T->zz*T->yw);
	M -= T->xz*(T->yy*T->zw - T->yz*T->yw);
	M += T->xw*(T->yy*T->zz - T->yz*T->yz);
	ret -= M*T->xw;
	return ret;
}
static f4v st4_adjugate_row1(const symtensor4 *T)
{
	f4v r;
	r.x = T->yy*T->zz*T->ww + T->yz*T->zw*T->yw + T->yw*T->yz*T->zw - T->yy*T->zw*T->zw - T->yz*T->yz*T->ww - T->yw*T->zz*T->yw;
	r.y = T->xy*T->zw*T->zw + T->yz*T->xz*T->ww + T->yw*T->zz*T->xw - T->xy*T->zz*T->ww - T->yz*T->zw*T->xw - T->yw*T->xz*T->zw;
	r.z = T->xy*T->yz*T->ww + T->yy*T->zw*T->xw + T->yw*T->xz*T->yw - T->xy*T->zw*T->yw - T->yy*T->xz*T->ww - T->yw*T->yz*T->xw;
	r.w = T->xy*T->zz*T->yw + T->yy*T->xz*T->zw + T->yz*T->yz*T->xw - T->xy*T->yz*T->zw - T->yy*T->zz*T->xw - T->yz*T->xz*T->yw;
	return r;
}
static f4v st4_dot(const symtensor4 *T, f4v v)
{
	f4v r;
	r.x = T->xx*v.x + T->xy*v.y + T->xz*v.z + T->xw*v.w;
	r.y = T->xy*v.x + T->yy*v.y + T->yz*v.z + T->yw*v.w;
	r.z = T->xz*v.x + T->yz*v.y + T->zz*v.z + T->zw*v.w;
	r.w = T->xw*v.x + T->yw*v.y + T->zw*v.z + T->ww*v.w;
	return r;
}
static float st4_ddot(const symtensor4 *T, f4v v)
{
	return T->xx*v.x*v.x + T->yy*v.y*v.y + T->zz*v.z*v.z + T->ww*v.w*v.w +
		2*((T->xy*v.y + T->xw*v.w)*v.x + (T->yz*v.z + T->yw*v.w)*v.y + (T->xz*v.x + T->zw*v.w)*v.z);
}
static float f4v_dot(f4v a, f4v b) { return a.x*b.x + a.y*b.y + a.z*b.z + a.w*b.w; }
static f4v f4v_scale(f4v a, float s) { f4v r = { a.x*s, a.y*s, a.z*s, a.w*s }; return r; }
static float f4v_hypot(f4v v)
{
	const float pm = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
	if (!pm) return 0;
	const f4v w = f4v_scale(v, 1.0f/pm);
	return pm*sqrtf(f4v_dot(w, w));
}

void orc_mls(const orc_params *p, orc_f4 *newVel,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const int dyn = p->boundarytype == ORC_DYN_BOUNDARY;
	const int last = dyn ? PT_BOUNDARY : PT_FLUID;
	const float inv_h = 1.0f/p->slength;
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		orc_f4 vel = velArray[index];
		symtensor4 mls = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		mls.xx = W_c(p->kerneltype, 0.0f, p->slength, wcoeff, wsub)*pos.w/physical_density(p, vel.w, FLUID_NUM(info));
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		for (int ptype = PT_FLUID; ptype <= last; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				if (!isfinite(npos.w)) continue;
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);
				const float neib_rho = physical_density(p, velArray[neib_index].w, FLUID_NUM(infoArray[neib_index]));
				if (r < p->influenceradius) {
					const float w = W_c(p->kerneltype, r, p->slength, wcoeff, wsub)*npos.w/neib_rho;
					const float sx = rx*inv_h, sy = ry*inv_h, sz = rz*inv_h;   /* relPos/slength */
					mls.xx += w;
					mls.xy += sx*w; mls.xz += sy*w; mls.xw += sz*w;
					mls.yy += sx*sx*w; mls.yz += sx*sy*w; mls.yw += sx*sz*w;
					mls.zz += sy*sy*w; mls.zw += sy*sz*w; mls.ww += sz*sz*w;
				}
			}
		}
		const f4v E = { 1, 0, 0, 0 };
		const float D = st4_det(&mls);
		f4v B;
		if (fabsf(D) < FLT_EPSILON) {
			symtensor4 me = mls;
			const float eps = fabsf(D) + FLT_EPSILON;
			me.xx += eps; me.yy += eps; me.zz += eps; me.ww += eps;
			const float De = st4_det(&me);
			B = f4v_scale(st4_adjugate_row1(&me), 1.0f/De);
		} else {
			B = f4v_scale(st4_adjugate_row1(&mls), 1.0f/D);
		}
		for (unsigned steps = 0; steps < 32; ++steps) {
			const float lenB = f4v_hypot(B);
			const f4v MdotB = st4_dot(&mls, B);
			const f4v residual = { E.x - MdotB.x, E.y - MdotB.y, E.z - MdotB.z, E.w - MdotB.w };
			const float num = st4_ddot(&mls, residual);
			const f4v Mp = st4_dot(&mls, residual);
			const float den = f4v_dot(Mp, Mp);
			const f4v corr = f4v_scale(residual, num/den);
			const float lencorr = f4v_hypot(corr);
			if (f4v_hypot(residual) < lenB*FLT_EPSILON) break;
			if (lencorr < 2*lenB*FLT_EPSILON) break;
			B.x += corr.x; B.y += corr.y; B.z += corr.z; B.w += corr.w;
		}
		B.y /= p->slength; B.z /= p->slength; B.w /= p->slength;
		vel.w = B.x*W_c(p->kerneltype, 0.0f, p->slength, wcoeff, wsub)*pos.w;
		for (int ptype = PT_FLUID; ptype <= last; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				if (!isfinite(npos.w)) continue;
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);
				const orc_info ninfo = infoArray[neib_index];
				if (r < p->influenceradius && (dyn || PART_TYPE(ninfo) == PT_FLUID)) {
					const float w = W_c(p->kerneltype, r, p->slength, wcoeff, wsub)*npos.w;
					vel.w += (B.x + B.y*rx + B.z*ry + B.w*rz)*w;
				}
			}
		}
		vel.w = numerical_density(p, vel.w, FLUID_NUM(info));
		newVel[index] = vel;
	}
}

/* ==== post-processing engines: src/cuda/post_process_kernel.cu:58-392 ============================== */

/* XSPH mean velocity of the forces pass (ENABLE_XSPH): compute_mean_vel forces_kernel.def:2986-2994 accumulated over the
 * fluid neighbours of fluid particles, written as 2*mean_vel by write_xsph :3366-3368; consumed by
 * compute_corrected_velocity (euler_body above).  Other particles' rows are left alone, as in the reference.
 *   mean_vel -= m_j W(r, h) (v_i - v_j) / (rho_i + rho_j)     (float3/float = multiply by the reciprocal) */
void orc_xsph(const orc_params *p, orc_f4 *xsph,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t fromParticle, uint32_t toParticle)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = fromParticle; index < toParticle; ++index) {
		const orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		if (PART_TYPE(info) != PT_FLUID || INACTIVE(pos)) continue;
		const orc_f4 vel = velArray[index];
		const float rho = physical_density(p, vel.w, FLUID_NUM(info));
		float mx = 0.0f, my = 0.0f, mz = 0.0f;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			if (!isfinite(npos.w)) continue;
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const float r = sqrtf(sqlength3(rx, ry, rz));
			if (r >= p->influenceradius) continue;
			const orc_f4 nvel = velArray[neib_index];
			const float n_rho = physical_density(p, nvel.w, FLUID_NUM(infoArray[neib_index]));
			const float t = npos.w*W_c(p->kerneltype, r, p->slength, wcoeff, wsub);
			const float inv = 1.0f/(rho + n_rho);
			mx = fmaf(-(t*(vel.x - nvel.x)), inv, mx);
			my = fmaf(-(t*(vel.y - nvel.y)), inv, my);
			mz = fmaf(-(t*(vel.z - nvel.z)), inv, mz);
		}
		xsph[index].x = 2.0f*mx; xsph[index].y = 2.0f*my; xsph[index].z = 2.0f*mz; xsph[index].w = 0.0f;
	}
}

/* calcVortDevice :58-135: vorticity of active fluid particles from their FLUID neighbours, NaN elsewhere */
void orc_vorticity(const orc_params *p, float *vorticity /* 3 per particle */,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		float *out = vorticity + 3*(size_t)index;
		if (PART_TYPE(info) != PT_FLUID || INACTIVE(pos)) { out[0] = out[1] = out[2] = NAN; continue; }
		const orc_f4 vel = velArray[index];
		float vx = 0.0f, vy = 0.0f, vz = 0.0f;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			if (!isfinite(npos.w)) continue;
			const float r = sqrtf(rx*rx + ry*ry + rz*rz);
			const orc_f4 nvel = velArray[neib_index];
			const float ux = vel.x - nvel.x, uy = vel.y - nvel.y, uz = vel.z - nvel.z;   /* relVel, .w = neib rho~ */
			if (r < p->influenceradius) {
				const float f = F_c(p->kerneltype, r, p->slength, fcoeff)*npos.w/
					physical_density(p, nvel.w, FLUID_NUM(infoArray[neib_index]));
				vx += f*(uy*rz - uz*ry);
				vy += f*(uz*rx - ux*rz);
				vz += f*(ux*ry - uy*rx);
			}
		}
		out[0] = vx; out[1] = vy; out[2] = vz;
	}
}

/* calcTestpointsVelocityDevice :138-236 (non-SA, no k-epsilon): Shepard-normalised velocity and pressure of the
 * FLUID neighbours, written over the test point's own velocity row (the kernel reads and writes the same array) */
void orc_testpoints(const orc_params *p, orc_f4 *velArray,
	const orc_f4 *posArray, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!TESTPOINT(info)) continue;
		const orc_f4 pos = posArray[index];
		orc_f4 avg = {0, 0, 0, 0};
		float alpha = 0.0f;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const float r = sqrtf(rx*rx + ry*ry + rz*rz);
			if (r < p->influenceradius) {
				const orc_f4 nvel = velArray[neib_index];
				const int nfl = FLUID_NUM(infoArray[neib_index]);
				const float w = W_c(p->kerneltype, r, p->slength, wcoeff, wsub)*npos.w/physical_density(p, nvel.w, nfl);
				avg.x += w*nvel.x; avg.y += w*nvel.y; avg.z += w*nvel.z;
				avg.w += w*orc_P(p, nvel.w, nfl);
				alpha += w;
			}
		}
		if (alpha > 1e-5f) {
			const float inv = 1.0f/alpha;     /* float4 /= float, src/vector_math.h */
			avg.x *= inv; avg.y *= inv; avg.z *= inv; avg.w *= inv;
		} else {
			avg.x = avg.y = avg.z = avg.w = 0.0f;
		}
		velArray[index] = avg;
	}
}

/* calcInterfaceparticleDevice :388-560 (non-SA): FG_SURFACE / FG_INTERFACE flags of fluid particles of a multi-fluid run.
 * Two SPH normals per particle: one over all neighbours (free surface), one over the neighbours of the same fluid and the
 * non-fluid ones (interface); a particle whose cone is empty for the first is at the free surface, one whose cone is empty
 * only for the second is at the interface.  Unlike the surface kernel the gradient sum carries the particle's own volume
 * (applied after the loop) and planes do not enter.  infoArray is updated in place. */
void orc_interface(const orc_params *p, orc_info *infoArray, orc_f4 *normals /* may be NULL */,
	const orc_f4 *posArray, const orc_f4 *velArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd,
	float cosconeanglefluid, float cosconeanglenonfluid)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);
	/* only type bits and fluid numbers of the neighbours are read, which this pass never changes: in place is race free */
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		if (PART_TYPE(info) != PT_FLUID || INACTIVE(pos)) {
			if (normals) { const orc_f4 nn = { NAN, NAN, NAN, NAN }; normals[index] = nn; }
			continue;
		}
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		info.x &= (uint16_t)~(FG_SURFACE | FG_INTERFACE);
		const float p_volume = pos.w/physical_density(p, velArray[index].w, FLUID_NUM(info));
		orc_f4 nfs = {0, 0, 0, 0}, nif = {0, 0, 0, 0};
		nfs.w = W_c(p->kerneltype, 0.0f, p->slength, wcoeff, wsub)*p_volume;
		nif.w = W_c(p->kerneltype, 0.0f, p->slength, wcoeff, wsub)*p_volume;
		for (int ptype = PT_FLUID; ptype <= PT_BOUNDARY; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				if (!isfinite(npos.w)) continue;
				const orc_info n_info = infoArray[neib_index];
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);
				const float n_volume = npos.w/physical_density(p, velArray[neib_index].w, FLUID_NUM(n_info));
				if (r < p->influenceradius) {
					const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
					nfs.x -= f*rx; nfs.y -= f*ry; nfs.z -= f*rz;
					nfs.w += W_c(p->kerneltype, r, p->slength, wcoeff, wsub)*n_volume;
				}
				if (r < p->influenceradius && (FLUID_NUM(info) == FLUID_NUM(n_info) || PART_TYPE(n_info) != PT_FLUID)) {
					const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
					nif.x -= f*rx; nif.y -= f*ry; nif.z -= f*rz;
					nif.w += W_c(p->kerneltype, r, p->slength, wcoeff, wsub)*n_volume;
				}
			}
		}
		nfs.x *= p_volume; nfs.y *= p_volume; nfs.z *= p_volume;
		nif.x *= p_volume; nif.y *= p_volume; nif.z *= p_volume;
		const float lfs = sqrtf(nfs.x*nfs.x + nfs.y*nfs.y + nfs.z*nfs.z);
		const float lif = sqrtf(nif.x*nif.x + nif.y*nif.y + nif.z*nif.z);
		int nc_fs = 0, nc_if = 0;
		for (int ptype = PT_FLUID; ptype <= PT_BOUNDARY; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				if (!isfinite(npos.w)) continue;
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);
				const orc_info n_info = infoArray[neib_index];
				const float cosconeangle = (PART_TYPE(n_info) == PT_FLUID) ? cosconeanglefluid : cosconeanglenonfluid;
				if (r < p->influenceradius) {
					const float criteria = -(nfs.x*rx + nfs.y*ry + nfs.z*rz);
					if (criteria > r*lfs*cosconeangle) nc_fs++;
				}
				if (r < p->influenceradius && (FLUID_NUM(info) == FLUID_NUM(n_info) || PART_TYPE(n_info) != PT_FLUID)) {
					const float criteria = -(nif.x*rx + nif.y*ry + nif.z*rz);
					if (criteria > r*lif*cosconeangle) nc_if++;
				}
			}
		}
		if (!nc_fs) info.x |= FG_SURFACE;
		if (!nc_if && nc_fs) info.x |= FG_INTERFACE;
		infoArray[index] = info;
		if (normals) {
			nfs.x /= lfs; nfs.y /= lfs; nfs.z /= lfs;
			nif.x /= lif; nif.y /= lif; nif.z /= lif;
			normals[index] = (!nc_if && nc_fs) ? nif : nfs;
		}
	}
}

/* calcSurfaceparticleDevice :239-392 (non-SA): FG_SURFACE flag of fluid particles from the cone test around the
 * (unnormalised) SPH normal; optional normals (xyz normalised, w = Shepard sum).  infoArray is updated in place. */
void orc_surface(const orc_params *p, orc_info *infoArray, orc_f4 *normals /* may be NULL */,
	const orc_f4 *posArray, const orc_f4 *velArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd,
	float cosconeanglefluid, float cosconeanglenonfluid)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, kr);
	/* first pass reads the type bits only, which the second pass never changes: in-place update is race free */
#pragma omp parallel for schedule(dynamic, 512)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		orc_info info = infoArray[index];
		const orc_f4 pos = posArray[index];
		if (PART_TYPE(info) != PT_FLUID || INACTIVE(pos)) {
			if (normals) { const orc_f4 nn = { NAN, NAN, NAN, NAN }; normals[index] = nn; }
			continue;
		}
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		info.x &= (uint16_t)~FG_SURFACE;
		orc_f4 normal = {0, 0, 0, 0};
		normal.w = W_c(p->kerneltype, 0.0f, p->slength, wcoeff, wsub)*pos.w/physical_density(p, velArray[index].w, FLUID_NUM(info));
		for (int ptype = PT_FLUID; ptype <= PT_BOUNDARY; ++ptype) {   /* for_every_neib, non-SA */
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				if (!isfinite(npos.w)) continue;
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);
				const float neib_vol = npos.w/physical_density(p, velArray[neib_index].w, FLUID_NUM(infoArray[neib_index]));
				if (r < p->influenceradius) {
					const float f = F_c(p->kerneltype, r, p->slength, fcoeff)*neib_vol;
					normal.x -= f*rx; normal.y -= f*ry; normal.z -= f*rz;
					normal.w += W_c(p->kerneltype, r, p->slength, wcoeff, wsub)*neib_vol;
				}
			}
		}
		if ((p->simflags & ORC_ENABLE_PLANES))
			for (uint32_t k = 0; k < p->numplanes; ++k) {
				const float dx = (gridPos[0] - p->plane_gridpos[k][0])*p->cellSize[0] + (pos.x - p->plane_pos[k][0]);
				const float dy = (gridPos[1] - p->plane_gridpos[k][1])*p->cellSize[1] + (pos.y - p->plane_pos[k][1]);
				const float dz = (gridPos[2] - p->plane_gridpos[k][2])*p->cellSize[2] + (pos.z - p->plane_pos[k][2]);
				const float *nrm = p->plane_normal[k];
				const float r = fabsf(dx*nrm[0] + dy*nrm[1] + dz*nrm[2]);
				if (r < p->influenceradius) {
					const float len = sqrtf(normal.x*normal.x + normal.y*normal.y + normal.z*normal.z);
					normal.x += nrm[0]*len; normal.y += nrm[1]*len; normal.z += nrm[2]*len;
				}
			}
		const float normal_length = sqrtf(normal.x*normal.x + normal.y*normal.y + normal.z*normal.z);
		int nc = 0;
		for (int ptype = PT_FLUID; ptype <= PT_BOUNDARY; ++ptype) {
			neib_iter it;
			neib_iter_init(&it, p, ptype, index, &pos, gridPos, cellStart, neibsList);
			uint32_t neib_index;
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 npos = posArray[neib_index];
				const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
				if (!isfinite(npos.w)) continue;
				const float r = sqrtf(rx*rx + ry*ry + rz*rz);
				if (r < p->influenceradius) {
					const float criteria = -(normal.x*rx + normal.y*ry + normal.z*rz);
					const float cosconeangle = (PART_TYPE(infoArray[neib_index]) == PT_FLUID) ? cosconeanglefluid : cosconeanglenonfluid;
					if (criteria > r*normal_length*cosconeangle) nc++;
				}
			}
		}
		if (!nc) info.x |= FG_SURFACE;
		infoArray[index] = info;
		if (normals) {
			normal.x /= normal_length; normal.y /= normal_length; normal.z /= normal_length;
			normals[index] = normal;
		}
	}
}

/* ====================================================================================================
 * Semi-analytical boundaries: solid walls, no open boundaries, no k-epsilon.
 * src/cuda/boundary_conditions_kernel.cu; the template switches of sa_segment_bc_params / sa_vertex_bc_params
 * (has_io, has_keps) are both off, has_moving follows ENABLE_MOVING_BODIES.
 * ==================================================================================================== */

/* RHO, src/cuda/phys_core.cu:106-112 (__powf -> powf): relative density from pressure */
float orc_RHO(const orc_params *p, float pres, int i)
{
	return (float)(powf(pres/p->bcoeff[i] + 1.0f, 1.0f/p->gammacoeff[i]) - 1.0);
}

static inline int has_vertex(const uint32_t *verts, uint32_t id)   /* src/particleinfo.h has_vertex */
{
	return verts[0] == id || verts[1] == id || verts[2] == id;
}

/* computeVertexNormalDevice, :1766-1831 */
void orc_sa_compute_vertex_normal(const orc_params *p, orc_f4 *boundelement, const uint32_t *vertices,
	const orc_info *infoArray, const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t particleRangeEnd)
{
#pragma omp parallel for schedule(dynamic, 1024)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!VERTEX(info)) continue;
		const orc_f4 pos = { 0.0f, 0.0f, 0.0f, 0.0f };
		const uint32_t our_id = orc_info_id(info);
		float avg[3] = { 0.0f, 0.0f, 0.0f };
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			/* :1813-1814: a vertex of an open boundary averages over that boundary's segments, any other vertex over the solid
			 * ones (no open-boundary flags anywhere: every adjacent segment, as before) */
			if (!IO_BOUNDARY_X(info) != !IO_BOUNDARY_X(infoArray[neib_index])) continue;
			if (!has_vertex(vertices + 4*(size_t)neib_index, our_id)) continue;
			const orc_f4 be = boundelement[neib_index];
			avg[0] += be.x*be.w; avg[1] += be.y*be.w; avg[2] += be.z*be.w;
		}
		const float inv = 1.0f/sqrtf(avg[0]*avg[0] + avg[1]*avg[1] + avg[2]*avg[2]);   /* normalize(): v*rsqrtf(sqlength) */
		boundelement[index].x = avg[0]*inv; boundelement[index].y = avg[1]*inv; boundelement[index].z = avg[2]*inv;
		boundelement[index].w = NAN;
	}
}

/* common_ndata, :633-657: a fluid neighbour of a segment or vertex */
typedef struct { float r, w, press; orc_f4 vel; } sa_ndata;
static inline sa_ndata sa_fluid_ndata(const orc_params *p, float wcoeff, float wsub, const orc_f4 *velArray,
	const orc_info *infoArray, uint32_t neib_index, float rx, float ry, float rz, float neib_mass)
{
	sa_ndata n;
	const int nfl = FLUID_NUM(infoArray[neib_index]);
	n.vel = velArray[neib_index];
	n.r = sqrtf(rx*rx + ry*ry + rz*rz);
	n.w = W_c(p->kerneltype, n.r, p->slength, wcoeff, wsub)*neib_mass/physical_density(p, n.vel.w, nfl);
	n.press = orc_P(p, n.vel.w, nfl);
	return n;
}

/* saSegmentBoundaryConditionsDevice :1425-1520 and saSegmentBoundaryConditionsRepackDevice :1544-1640
 * (they differ by the moving-body velocity, which the repack variant leaves out).  In place: reads fluid and vertex
 * rows, writes boundary rows of vel and gGam. */
static void sa_segment_bc_impl(const orc_params *p, orc_f4 *velArray, orc_f4 *gGamArray, const orc_f4 *posArray,
	const uint32_t *vertices, const orc_f4 *boundelement, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd, int step, int repack,
	float *tke, float *eps, orc_f4 *eulerVelArray, float deltap)
{
	/* tke/eps/eulerVelArray: the k-epsilon members of sa_segment_bc_params (src/cuda/sa_bc_params.h:152-200), NULL without KEPSILON */
	const int keps = tke != NULL && !repack;
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const int has_moving = (p->simflags & ORC_ENABLE_MOVING_BODIES) != 0;
	if (step == -1) step = 0;     /* "step -1 is the same as step 0", boundary_conditions.cu:177-180 */
#pragma omp parallel for schedule(dynamic, 1024)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!BOUNDARY(info)) continue;
		const orc_f4 pos = posArray[index];
		const orc_f4 normal = boundelement[index];
		const uint32_t *verts = vertices + 4*(size_t)index;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);

		/* common_pout, common_segment_pout :390-432 */
		float sumpWall = 0.0f, shepard_div = 0.0f;
		orc_f4 gGam = { 0.0f, 0.0f, 0.0f, gGamArray[index].w };
		orc_f4 vel = { 0.0f, 0.0f, 0.0f, 0.0f };
		const int calcGam = has_moving || !isfinite(gGam.w) || step == 0;
		if (calcGam) gGam.w = 0.0f;
		float sumtke = 0.0f, sumeps = 0.0f;                  /* common_keps_pout :509-520 */
		orc_f4 eulerVel = { 0.0f, 0.0f, 0.0f, 0.0f };        /* eulervel_pout :470-484 */

		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_VERTEX, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			if (INACTIVE(posArray[neib_index])) continue;
			if (!has_vertex(verts, orc_info_id(infoArray[neib_index]))) continue;
			if (has_moving && !repack && MOVING(info)) {        /* moving_vertex_contrib :781-793 */
				const orc_f4 nv = velArray[neib_index];
				vel.x += nv.x; vel.y += nv.y; vel.z += nv.z;
			}
			if (calcGam) {
				const orc_f4 g = gGamArray[neib_index];
				gGam.x += g.x; gGam.y += g.y; gGam.z += g.z; gGam.w += g.w;
			}
			if (keps) {      /* keps_vertex_contrib :748-758 */
				const orc_f4 e = eulerVelArray[neib_index];
				eulerVel.x += e.x; eulerVel.y += e.y; eulerVel.z += e.z; eulerVel.w += e.w;
			}
		}
		if (calcGam) {
			const float inv = 1.0f/3;      /* float4 /= float */
			gGam.x *= inv; gGam.y *= inv; gGam.z *= inv; gGam.w *= inv;
			gGamArray[index] = gGam;
			gGam.w = fmaxf(gGam.w, 1e-5f);
		}
		if (!repack) { vel.x /= 3; vel.y /= 3; vel.z /= 3; }

		const int fl = FLUID_NUM(info);
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			if (INACTIVE(npos)) continue;
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const sa_ndata n = sa_fluid_ndata(p, wcoeff, wsub, velArray, infoArray, neib_index, rx, ry, rz, npos.w);
			if (!(n.r < p->influenceradius && (normal.x*rx + normal.y*ry + normal.z*rz) < 0.0f)) continue;
			const float gdot = p->gravity[0]*rx + p->gravity[1]*ry + p->gravity[2]*rz;
			sumpWall += fmaxf(n.press + physical_density(p, n.vel.w, fl)*gdot, 0.0f)*n.w;
			if (keps) {      /* keps_fluid_contrib :816-826: dk/dn = 0, de/dn = 4 c_mu^(3/4) k^(3/2)/(kappa r) (de_dn_solid :806-813) */
				const float norm_dist = fmaxf(fabsf(normal.x*rx + normal.y*ry + normal.z*rz), deltap);      /* segment_keps_ndata :678-693 */
				const float nk = tke[neib_index], ne = eps[neib_index];
				sumtke += n.w*nk;
				sumeps += n.w*(ne + 1.603090412f*powf(nk, 1.5f)/norm_dist);
			}
			shepard_div += n.w;
		}
		/* impose_solid_bc :1295-1306 */
		shepard_div = fmaxf(shepard_div, 0.1f*gGam.w);
		vel.w = orc_RHO(p, sumpWall/shepard_div, fl);
		velArray[index] = vel;
		if (keps) {      /* impose_solid_keps_bc :1262-1277; the normal is the float4 boundary element, its .w (the area) rides along */
			tke[index] = sumtke/shepard_div;
			eps[index] = fmaxf(sumeps/shepard_div, 1e-5f);
			const float inv = 1.0f/3;
			eulerVel.x *= inv; eulerVel.y *= inv; eulerVel.z *= inv; eulerVel.w *= inv;
			const float d = eulerVel.x*normal.x + eulerVel.y*normal.y + eulerVel.z*normal.z;
			eulerVel.x -= d*normal.x; eulerVel.y -= d*normal.y; eulerVel.z -= d*normal.z; eulerVel.w -= d*normal.w;
			eulerVelArray[index] = eulerVel;
		}
	}
}

void orc_sa_segment_bc(const orc_params *p, orc_f4 *velArray, orc_f4 *gGamArray, const orc_f4 *posArray,
	const uint32_t *vertices, const orc_f4 *boundelement, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd, int step, int repack)
{
	sa_segment_bc_impl(p, velArray, gGamArray, posArray, vertices, boundelement, infoArray, hashArray, cellStart, neibsList,
		particleRangeEnd, step, repack, NULL, NULL, NULL, 0.0f);
}
void orc_sa_segment_bc_keps(const orc_params *p, orc_f4 *velArray, orc_f4 *gGamArray, float *tke, float *eps, orc_f4 *eulerVel,
	const orc_f4 *posArray, const uint32_t *vertices, const orc_f4 *boundelement, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd, int step, float deltap)
{
	sa_segment_bc_impl(p, velArray, gGamArray, posArray, vertices, boundelement, infoArray, hashArray, cellStart, neibsList,
		particleRangeEnd, step, 0, tke, eps, eulerVel, deltap);
}

/* saVertexBoundaryConditionsDevice :2195-2253 and its Repack twin: density of vertex particles from the fluid */
static void sa_vertex_bc_impl(const orc_params *p, orc_f4 *velArray, const orc_f4 *gGamArray, const orc_f4 *posArray,
	const orc_info *infoArray, const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t particleRangeEnd, float *tke, float *eps, orc_f4 *eulerVelArray, const uint32_t *vertices, const orc_f4 *boundelement)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
#pragma omp parallel for schedule(dynamic, 1024)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!VERTEX(info)) continue;
		const orc_f4 pos = posArray[index];
		const float gam = gGamArray[index].w;
		const int fl = FLUID_NUM(info);
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		float sumpWall = 0.0f, shepard_div = 0.0f;
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			if (INACTIVE(npos)) continue;
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const sa_ndata n = sa_fluid_ndata(p, wcoeff, wsub, velArray, infoArray, neib_index, rx, ry, rz, npos.w);
			if (n.r < p->influenceradius) {
				const float gdot = p->gravity[0]*rx + p->gravity[1]*ry + p->gravity[2]*rz;
				sumpWall += fmaxf(n.press + physical_density(p, n.vel.w, fl)*gdot, 0.0f)*n.w;
				shepard_div += n.w;
			}
		}
		/* vertex_boundary_loop :1002-1021 (KEPSILON): k and epsilon of a vertex are the means over its adjacent segments
		 * (keps_boundary_contrib :918-927, impose_vertex_keps_bc :1052-1072), its Eulerian velocity is made tangential to the wall */
		float sumtke = 0.0f, sumeps = 0.0f; int numseg = 0;
		if (tke) {
			neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				if (!has_vertex(vertices + 4*(size_t)neib_index, orc_info_id(info))) continue;
				sumtke += tke[neib_index]; sumeps += eps[neib_index]; numseg += 1;
			}
		}
		shepard_div = fmaxf(shepard_div, 0.1f*gam);
		velArray[index].w = orc_RHO(p, sumpWall/shepard_div, fl);
		if (tke) {
			tke[index] = fmaxf(sumtke/numseg, 1e-6f);
			eps[index] = fmaxf(sumeps/numseg, 1e-6f);
			const orc_f4 nrm = boundelement[index];
			orc_f4 e = eulerVelArray[index];
			const float d = e.x*nrm.x + e.y*nrm.y + e.z*nrm.z;
			e.x -= d*nrm.x; e.y -= d*nrm.y; e.z -= d*nrm.z;
			eulerVelArray[index] = e;
		}
	}
}

void orc_sa_vertex_bc(const orc_params *p, orc_f4 *velArray, const orc_f4 *gGamArray, const orc_f4 *posArray,
	const orc_info *infoArray, const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t particleRangeEnd)
{
	sa_vertex_bc_impl(p, velArray, gGamArray, posArray, infoArray, hashArray, cellStart, neibsList, particleRangeEnd,
		NULL, NULL, NULL, NULL, NULL);
}
void orc_sa_vertex_bc_keps(const orc_params *p, orc_f4 *velArray, const orc_f4 *gGamArray, float *tke, float *eps, orc_f4 *eulerVel,
	const orc_f4 *posArray, const uint32_t *vertices, const orc_f4 *boundelement, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd)
{
	sa_vertex_bc_impl(p, velArray, gGamArray, posArray, infoArray, hashArray, cellStart, neibsList, particleRangeEnd,
		tke, eps, eulerVel, vertices, boundelement);
}

/* Euler step of the k-epsilon model (src/cuda/euler_kernel.def:219-231,262-274,325-337): semi-implicit k and epsilon of the fluid
 * particles, Eulerian velocity of the wall particles (+= dt force), and the eddy viscosity 0.9 k^2/epsilon of every particle
 * (the reference's constant: C_mu = 0.09 appears as 0.9f there).  The rows are those the Euler kernel integrates. */
void orc_euler_keps(const orc_params *p, float *newTke, float *newEps, float *newTurbVisc, orc_f4 *newEulerVel,
	const float *oldTke, const float *oldEps, const orc_f4 *oldEulerVel, const float *dkde, const orc_f4 *forces,
	const orc_f4 *oldPos, const orc_info *infoArray, uint32_t particleRangeEnd, float dt)
{
	(void)p;
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (INACTIVE(oldPos[index])) continue;      /* the Euler kernel leaves the rows of disabled particles alone */
		float k = oldTke[index], e = oldEps[index];
		orc_f4 ev = oldEulerVel[index];
		if (FLUID(info)) {
			const float *d = dkde + 3*(size_t)index;
			const float oldK = k;
			k = (oldK + dt*d[0])/(1.0f + dt*e/oldK);
			e = (e + dt*d[1])/(1.0f + dt*e/oldK*d[2]);
		} else if (BOUNDARY(info) || VERTEX(info)) {
			const orc_f4 f = forces[index];
			ev.x += dt*f.x; ev.y += dt*f.y; ev.z += dt*f.z; ev.w += dt*f.w;
		}
		newTke[index] = k; newEps[index] = e; newTurbVisc[index] = 0.9f*k*k/e;
		newEulerVel[index] = ev;
	}
}

/* ---- gamma and its gradient: src/cuda/gamma.cuh (Wendland kernel only, as the reference) ------------------------
 * Written without FMA contraction, like oracle/_ref (g++ -ffp-contract=off) which pins it bit for bit
 * (tests/test_oracle_pinned.py::test_gamma_quadrature_matches_reference). */
typedef struct { float x, y, z; } v3;
static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x*s, a.y*s, a.z*s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }            /* vector_math.h:560-563 */
static inline v3 v3_cross(v3 a, v3 b) { return v3_make(a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); }
static inline float v3_sqlen(v3 a) { return v3_dot(a, a); }
static inline float v3_len(v3 a) { return sqrtf(v3_sqlen(a)); }
static inline v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f/sqrtf(v3_sqlen(a))); }      /* v*rsqrtf(sqlength(v)), :584-588 */
static inline v3 v3_divs(v3 a, float s) { return v3_scale(a, 1.0f/s); }                   /* float3 / float = v*(1/s), vector_math.h */

static const float GQ_O5_weights[3] = { 0.225f, 0.132394152788506f, 0.125939180544827f };                 /* gamma.cuh:43-55 */
static const float GQ_O5_points[3][3] = {
	{ 0.333333333333333f, 0.333333333333333f, 0.333333333333333f },
	{ 0.059715871789770f, 0.470142064105115f, 0.470142064105115f },
	{ 0.797426985353087f, 0.101286507323456f, 0.101286507323456f } };
static const int GQ_O5_mult[3] = { 1, 3, 3 };

/* wendlandOnSegment, gamma.cuh:90-110 */
float orc_wendland_on_segment(float q)
{
	float intKernel = 0.0f;
	if (q < 2.0f) {
		float tmp = (1.0f - q/2.0f);
		float tmp4 = tmp*tmp;
		tmp4 *= tmp4;
		const float uq = 1.0f/q;
		intKernel = 0.009947183943243458485555235210782147627153727858778528046729f*tmp4*tmp*((((8.0f*uq + 20.0f)*uq + 30.0f)*uq) + 21.0f);
	}
	return intKernel;
}

/* gaussQuadratureO5, gamma.cuh:138-163.  NB the break test follows the accumulation, so the loop over j runs once more
 * than mult[i] says: the centroid point (mult 1) is taken twice, the others three times (the loop ends at j = 2).
 * Reproduced as written. */
static float gauss_quadrature_O5(v3 vPos0, v3 vPos1, v3 vPos2, v3 relPos)
{
	float val = 0.0f;
	for (int i = 0; i < 3; i++) {
		for (int j = 0; j < 3; j++) {
			v3 pa = v3_add(v3_add(v3_scale(vPos0, GQ_O5_points[i][j]), v3_scale(vPos1, GQ_O5_points[i][(j + 1) % 3])),
				v3_scale(vPos2, GQ_O5_points[i][(j + 2) % 3]));
			pa = v3_sub(pa, relPos);
			val += GQ_O5_weights[i]*orc_wendland_on_segment(v3_len(pa));
			if (j >= GQ_O5_mult[i])
				break;
		}
	}
	const float vol = v3_len(v3_cross(v3_sub(vPos1, vPos0), v3_sub(vPos2, vPos0)))/2.0f;
	return val*vol;
}
float orc_gauss_quadrature_O5(const float *v0, const float *v1, const float *v2, const float *rel)
{
	return gauss_quadrature_O5(v3_make(v0[0], v0[1], v0[2]), v3_make(v1[0], v1[1], v1[2]), v3_make(v2[0], v2[1], v2[2]),
		v3_make(rel[0], rel[1], rel[2]));
}

/* calcVertexRelPos, gamma.cuh:196-227 */
static void calc_vertex_rel_pos(v3 q_vb[3], v3 ns, const float *vPos0, const float *vPos1, const float *vPos2, float slength)
{
	unsigned j = 0;
	if (fabsf(ns.x) > fabsf(ns.y))
		j = 1;
	if ((1 - j)*fabsf(ns.x) + j*fabsf(ns.y) > fabsf(ns.z))
		j = 2;
	const v3 coord1 = v3_normalize(v3_make(
		-((j == 1)*ns.z) + (j == 2)*ns.y,
		(j == 0)*ns.z - ((j == 2)*ns.x),
		-((j == 0)*ns.y) + (j == 1)*ns.x));
	const v3 coord2 = v3_cross(ns, coord1);
	const float *vp[3] = { vPos0, vPos1, vPos2 };
	for (int k = 0; k < 3; ++k)
		q_vb[k] = v3_divs(v3_neg(v3_add(v3_scale(coord1, vp[k][0]), v3_scale(coord2, vp[k][1]))), slength);
}
void orc_calc_vertex_rel_pos(const float *ns, const float *vp0, const float *vp1, const float *vp2, float slength, float *out9)
{
	v3 q_vb[3];
	calc_vertex_rel_pos(q_vb, v3_make(ns[0], ns[1], ns[2]), vp0, vp1, vp2, slength);
	for (int i = 0; i < 3; ++i) { out9[3*i] = q_vb[i].x; out9[3*i + 1] = q_vb[i].y; out9[3*i + 2] = q_vb[i].z; }
}

/* gradGamma<WENDLAND>, gamma.cuh:248-370 */
static float grad_gamma_wendland(float slength, v3 q, const v3 *q_vb, v3 ns)
{
	float pas = v3_dot(ns, q);
	float qas = fabsf(pas);
	if (qas >= 2.f)
		return 0.f;
	float qas2 = qas*qas;
	float qas3 = qas2*qas;
	float qas4 = qas2*qas2;
	float qas5 = qas3*qas2;
	unsigned sIdx[2];
	float gradGamma_as = 0.f;
	float totalSumAngles = 0.f;
	float sumAngles = 0.f;
	for (unsigned e = 0; e < 3; e++) {
		sIdx[0] = e % 3;
		sIdx[1] = (e + 1) % 3;
		v3 v01 = v3_normalize(v3_sub(q_vb[sIdx[0]], q_vb[sIdx[1]]));
		v3 ne = v3_normalize(v3_cross(ns, v01));
		float pae = v3_dot(ne, v3_sub(q, q_vb[sIdx[0]]));
		float qae = v3_len(v3_add(v3_scale(ns, pas), v3_scale(ne, pae)));
		float pav0 = -v3_dot(v3_sub(q, q_vb[sIdx[0]]), v01);
		float pav1 = -v3_dot(v3_sub(q, q_vb[sIdx[1]]), v01);
		totalSumAngles += copysignf(atan2f(pav1, fabsf(pae)) - atan2f(pav0, fabsf(pae)), pae);
		if (qae < 2.0f) {
			pav0 = copysignf(fminf(fabsf(pav0), sqrtf(4.0f - qae*qae)), pav0);
			float pav02 = pav0*pav0;
			pav1 = copysignf(fminf(fabsf(pav1), sqrtf(4.0f - qae*qae)), pav1);
			float pav12 = pav1*pav1;
			float qav0 = fminf(sqrtf(qae*qae + pav0*pav0), 2.0f);
			float qav1 = fminf(sqrtf(qae*qae + pav1*pav1), 2.0f);
			float pae2 = pae*pae;
			float pae4 = pae2*pae2;
			float pae6 = pae4*pae2;
			gradGamma_as += 0.00015542474911f*(
				+ 48.0f*qas5*(28.0f + qas2)*(
						  atan2f(qas*pav1, pae*qav1) - atan2f(pav1, pae)
						-(atan2f(qas*pav0, pae*qav0) - atan2f(pav0, pae)))
				+ pae*(
					 pav1*(3.0f*qas4*(-420.0f + 29.0f*qav1)
						+ pae4*(-420.0f + 33.0f*qav1)
						+ 2.0f*qas2*(-210.0f*(8.0f + pav12) + 756.0f*qav1 + 19.0f*pav12*qav1)
						+ 4.0f*(336.0f + pav12*(pav12*(-21.0f + 2.0f*qav1) + 28.0f*(-5.0f + 3.0f*qav1)))
						+ 2.0f*pae2*(420.0f*(-2.0f + qav1) + 6.0f*qas2*(-105.0f + 8.0f*qav1) + pav12*(-140.0f + 13.0f*qav1))
						)
					- pav0*(3.0f*qas4*(-420.0f + 29.0f*qav0)
						+ pae4*(-420.0f + 33.0f*qav0)
						+ 2.0f*qas2*(-210.0f*(8.0f + pav02) + 756.0f*qav0 + 19.0f*pav02*qav0)
						+ 4.0f*(336.0f + pav02*(pav02*(-21.0f + 2.0f*qav0) + 28.0f*(-5.0f + 3.0f*qav0)))
						+ 2.0f*pae2*(420.0f*(-2.0f + qav0) + 6.0f*qas2*(-105.0f + 8.0f*qav0) + pav02*(-140.0f + 13.0f*qav0))
						)
					+ 3.0f*(5.0f*pae6 + 21.0f*pae4*(8.0f + qas2) + 35.0f*pae2*qas2*(16.0f + qas2) + 35.0f*qas4*(24.0f + qas2))
					*(
						 copysignf(1.f, pav1)*acoshf(fmaxf(qav1/fmaxf(qae, 1e-7f), 1.f))
						- copysignf(1.f, pav0)*acoshf(fmaxf(qav0/fmaxf(qae, 1e-7f), 1.f))
						)
					)
				);
			sumAngles += copysignf(atan2f(pav1, fabsf(pae)) - atan2f(pav0, fabsf(pae)), pae);
		}
	}
	const float tmp1 = 1.0f - qas/2.0f;
	float tmp2 = tmp1*tmp1;
	tmp2 *= tmp2*tmp1;
	gradGamma_as += (sumAngles - totalSumAngles)*0.05968310365947f*tmp2*(2.0f + 5.0f*qas + 4.0f*qas2);
	return gradGamma_as/slength;
}
float orc_grad_gamma(float slength, const float *q, const float *qvb9, const float *ns)
{
	v3 q_vb[3];
	for (int i = 0; i < 3; ++i) q_vb[i] = v3_make(qvb9[3*i], qvb9[3*i + 1], qvb9[3*i + 2]);
	return grad_gamma_wendland(slength, v3_make(q[0], q[1], q[2]), q_vb, v3_make(ns[0], ns[1], ns[2]));
}

/* |grad gamma_as| of one boundary element as the forces pass asks for it (compute_gamma_gradient, forces_kernel.def:1410-1428) */
float orc_grad_gamma_vp(float slength, float qx, float qy, float qz, const orc_f4 *belem,
	const float *vp0, const float *vp1, const float *vp2)
{
	v3 q_vb[3];
	const v3 ns = v3_make(belem->x, belem->y, belem->z);
	calc_vertex_rel_pos(q_vb, ns, vp0, vp1, vp2, slength);
	return grad_gamma_wendland(slength, v3_make(qx, qy, qz), q_vb, ns);
}

/* Gamma<WENDLAND, PT_FLUID> :404-435 and Gamma<WENDLAND, PT_VERTEX> :437-513 (q_vb may be permuted) */
static float gamma_wendland(int vertex, float slength, v3 q, v3 *q_vb, v3 ns, v3 oldGGam, float epsilon)
{
	v3 r_aSigma = v3_scale(ns, v3_dot(ns, q));
	float q_aSigma = fminf(v3_len(r_aSigma), 2.0f);
	float gamma_as = 0.0f;
	float gamma_vs = 0.0f;
	if (vertex) {
		const v3 ba = v3_sub(q_vb[1], q_vb[0]);
		const v3 ca = v3_sub(q_vb[2], q_vb[0]);
		const v3 pa = v3_sub(q, q_vb[0]);
		const float uu = v3_sqlen(ba);
		const float uv = v3_dot(ba, ca);
		const float vv = v3_sqlen(ca);
		const float wu = v3_dot(ba, pa);
		const float wv = v3_dot(ca, pa);
		const float invdet = 1.0f/(uv*uv - uu*vv);
		const float u = (uv*wv - vv*wu)*invdet;
		const float v = (uv*wu - uu*wv)*invdet;
		if (((fabsf(u - 1.0f) < epsilon && fabsf(v) < epsilon) ||
			 (fabsf(v - 1.0f) < epsilon && fabsf(u) < epsilon) ||
			 (fabsf(u) < epsilon && fabsf(v) < epsilon)) && q_aSigma < epsilon) {
			if (fabsf(u - 1.0f) < epsilon && fabsf(v) < epsilon) {
				const v3 tmp = q_vb[1];
				q_vb[1] = q_vb[2];
				q_vb[2] = q_vb[0];
				q_vb[0] = tmp;
			} else if (fabsf(v - 1.0f) < epsilon && fabsf(u) < epsilon) {
				const v3 tmp = q_vb[2];
				q_vb[2] = q_vb[1];
				q_vb[1] = q_vb[0];
				q_vb[0] = tmp;
			}
			const v3 inward_normal = v3_divs(v3_neg(oldGGam), fmaxf(v3_len(oldGGam), slength*1e-3f));
			const v3 e1 = v3_sub(q_vb[1], q_vb[0]), e2 = v3_sub(q_vb[2], q_vb[0]);
			const float l1 = v3_len(e1);
			const float l2 = v3_len(e2);
			const float abc = v3_dot(e1, inward_normal)/l1 + v3_dot(e2, inward_normal)/l2 + v3_dot(e1, e2)/l1/l2;
			const float d = v3_dot(inward_normal, v3_cross(e1, e2))/l1/l2;
			const float SolidAngle = fabsf(2.0f*atan2f(d, 1.0f + abc));
			gamma_vs = SolidAngle*0.079577471545947667884441881686257181017229822870228224373833f;
		}
	}
	if (q_aSigma < 2.0f && q_aSigma > epsilon) {
		const float intVal = gauss_quadrature_O5(v3_neg(q_vb[0]), v3_neg(q_vb[1]), v3_neg(q_vb[2]), q);
		if (vertex) gamma_as += intVal*v3_dot(ns, r_aSigma);    /* "+=" in the vertex specialisation, "=" in the fluid one: */
		else gamma_as = intVal*v3_dot(ns, r_aSigma);            /* 0 + (-0) is +0, so the sign of a zero differs */
	}
	return vertex ? gamma_vs + gamma_as : gamma_as;
}
float orc_gamma(int vertex, float slength, const float *q, const float *qvb9, const float *ns, const float *oldGGam, float epsilon)
{
	v3 q_vb[3];
	for (int i = 0; i < 3; ++i) q_vb[i] = v3_make(qvb9[3*i], qvb9[3*i + 1], qvb9[3*i + 2]);
	return gamma_wendland(vertex, slength, v3_make(q[0], q[1], q[2]), q_vb, v3_make(ns[0], ns[1], ns[2]),
		v3_make(oldGGam[0], oldGGam[1], oldGGam[2]), epsilon);
}

/* initGammaDevice<WENDLAND, cptype> :1891-1970 for cptype = PT_FLUID then PT_VERTEX (saInitGamma, boundary_conditions.cu:463-540):
 * grad gamma = sum over boundary neighbours of gradGamma_as n_s, then gamma = 1 - sum of Gamma_as (which needs the direction
 * of grad gamma just computed) */
void orc_sa_init_gamma(const orc_params *p, orc_f4 *newGGam, const orc_f4 *posArray, const orc_f4 *boundelement,
	const float *vertPos0, const float *vertPos1, const float *vertPos2, const orc_info *infoArray,
	const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd,
	float deltap, float epsilon)
{
	const float slength = p->slength, influenceradius = p->influenceradius;
	for (int cptype = PT_FLUID; cptype <= PT_VERTEX; cptype += 2) {
#pragma omp parallel for schedule(dynamic, 256)
		for (uint32_t index = 0; index < particleRangeEnd; ++index) {
			const orc_info info = infoArray[index];
			if (PART_TYPE(info) != cptype) continue;
			const orc_f4 pos = posArray[index];
			float gam = 1.0f;
			v3 gGam = v3_make(0.0f, 0.0f, 0.0f);
			int gridPos[3];
			orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
			for (int pass = 0; pass < 2; ++pass) {
				neib_iter it;
				uint32_t neib_index;
				neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
				while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
					/* InitGammaVars :1833-1868 */
					const orc_f4 npos = posArray[neib_index];
					const v3 relPos = v3_make(it.pos_corr[0] - npos.x, it.pos_corr[1] - npos.y, it.pos_corr[2] - npos.z);
					if (v3_len(relPos) > influenceradius + deltap*0.5f) continue;
					const orc_f4 be = boundelement[neib_index];
					const v3 normal = v3_make(be.x, be.y, be.z);
					const v3 q = v3_divs(relPos, slength);
					v3 q_vb[3];
					calc_vertex_rel_pos(q_vb, normal, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
						vertPos2 + 2*(size_t)neib_index, slength);
					if (pass == 0) {
						const float ggamma_as = grad_gamma_wendland(slength, q, q_vb, normal);
						gGam = v3_add(gGam, v3_scale(normal, ggamma_as));
					} else {
						gam -= gamma_wendland(cptype == PT_VERTEX, slength, q, q_vb, normal, gGam, epsilon);
					}
				}
			}
			newGGam[index].x = gGam.x; newGGam[index].y = gGam.y; newGGam[index].z = gGam.z; newGGam[index].w = gam;
		}
	}
}

/* integrateGammaDevice without dynamic gamma (ENABLE_GAMMA_QUADRATURE), src/cuda/density_sum_kernel.cu:690-765: like the
 * initialisation, but every listed element counts (no distance cut) and the direction used by the vertex specialisation
 * is the OLD grad gamma */
void orc_sa_integrate_gamma_quadrature(const orc_params *p, orc_f4 *newGGam, const orc_f4 *oldGGam, const orc_f4 *newPos,
	const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2, const orc_info *infoArray,
	const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd,
	int cptype, float epsilon)
{
	const float slength = p->slength;
#pragma omp parallel for schedule(dynamic, 256)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		if (PART_TYPE(infoArray[index]) != cptype) continue;
		const orc_f4 pos = newPos[index];
		const orc_f4 og = oldGGam[index];
		const v3 oldg = v3_make(og.x, og.y, og.z);
		orc_f4 g = { 0.0f, 0.0f, 0.0f, 1.0f };
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = newPos[neib_index];
			const v3 relPos = v3_make(it.pos_corr[0] - npos.x, it.pos_corr[1] - npos.y, it.pos_corr[2] - npos.z);
			const orc_f4 be = boundelem[neib_index];
			const v3 normal = v3_make(be.x, be.y, be.z);
			const v3 q = v3_divs(relPos, slength);
			v3 q_vb[3];
			calc_vertex_rel_pos(q_vb, normal, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
				vertPos2 + 2*(size_t)neib_index, slength);
			const float ggamAS = grad_gamma_wendland(slength, q, q_vb, normal);
			g.x += ggamAS*be.x; g.y += ggamAS*be.y; g.z += ggamAS*be.z;
			g.w -= gamma_wendland(cptype == PT_VERTEX, slength, q, q_vb, normal, oldg, epsilon);
		}
		newGGam[index] = g;
	}
}

/* ---- SA_BOUNDARY with density summation and dynamic gamma (the form StillWaterSA and most SA problems of the reference run) ----
 * densitySumVolumicDevice + densitySumBoundaryDevice (src/cuda/density_sum_kernel.cu:523-655), solid walls, no moving bodies:
 * the density of a fluid particle follows from the change of its kernel sum between the old and the new positions, gamma from
 * the mean of the old and new grad gamma along the displacement:
 *   gamma^{n+1} = gamma^n + sum_s 1/2 (ggam_s^n + ggam_s^{n+1}) . (r^{n+1} - r^n)
 *   rho^{n+1}   = (gamma^n rho^n + sum_b m_b W(r^{n+1}) - sum_b m_b W(r^n)) / gamma^{n+1}
 * newVel.w and newGGam of fluid rows are written; forces.w is the scratch of the volumic sum (as in the reference); the gGam
 * rows of vertex and boundary particles are copied (density_sum_impl, src/cuda/euler.cu:112-160). */
void orc_sa_density_sum(const orc_params *p, orc_f4 *newVel, orc_f4 *newGGam, orc_f4 *forces,
	const orc_f4 *oldPos, const orc_f4 *newPos, const orc_f4 *oldVel, const orc_f4 *oldGGam, const orc_f4 *boundelem,
	const float *vertPos0, const float *vertPos1, const float *vertPos2, const orc_info *infoArray,
	const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd)
{
	const float kr = 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const float slength = p->slength;
#pragma omp parallel for schedule(dynamic, 256)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!FLUID(info)) {
			if (VERTEX(info) || BOUNDARY(info)) newGGam[index] = oldGGam[index];
			continue;
		}
		const orc_f4 posN = oldPos[index], posNp1 = newPos[index];
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;    /* posDelta */
		/* computeDensitySumVolumicTerms (:206-250): fluid, then vertex neighbours (for_each_neib2) */
		float sumPmwN = 0.0f, sumPmwNp1 = 0.0f;
		for (int nptype = PT_FLUID; nptype <= PT_VERTEX; nptype += 2) {
			neib_iter it;
			uint32_t neib_index;
			neib_iter_init(&it, p, nptype, index, &posN, gridPos, cellStart, neibsList);
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 nN = oldPos[neib_index];
				if (INACTIVE(nN)) continue;
				const orc_f4 nNp1 = newPos[neib_index];
				const float rx = it.pos_corr[0] - nN.x, ry = it.pos_corr[1] - nN.y, rz = it.pos_corr[2] - nN.z;
				const float qx = (it.pos_corr[0] - nNp1.x) + dx, qy = (it.pos_corr[1] - nNp1.y) + dy, qz = (it.pos_corr[2] - nNp1.z) + dz;
				const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
				sumPmwN -= nN.w*W_c(p->kerneltype, rN, slength, wcoeff, wsub);      /* no range test on the old distance, as the reference */
				const float rNp1 = sqrtf(qx*qx + qy*qy + qz*qz);
				if (rNp1 < p->influenceradius)
					sumPmwNp1 += nN.w*W_c(p->kerneltype, rNp1, slength, wcoeff, wsub);
			}
		}
		forces[index].w = sumPmwNp1 + sumPmwN + 0.0f;     /* + sumVmwDelta (open boundaries: none) */
		/* computeDensitySumBoundaryTerms (:419-478) */
		float gGamDotR = 0.0f;
		v3 gGam = v3_make(0.0f, 0.0f, 0.0f);
		{
			neib_iter it;
			uint32_t neib_index;
			neib_iter_init(&it, p, PT_BOUNDARY, index, &posN, gridPos, cellStart, neibsList);
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 nN = oldPos[neib_index];
				if (INACTIVE(nN)) continue;
				const orc_f4 nNp1 = newPos[neib_index];
				const float inv = 1.0f/slength;      /* float4 / float */
				const v3 qN = v3_make((it.pos_corr[0] - nN.x)*inv, (it.pos_corr[1] - nN.y)*inv, (it.pos_corr[2] - nN.z)*inv);
				const v3 qNp1 = v3_make(((it.pos_corr[0] - nNp1.x) + dx)*inv, ((it.pos_corr[1] - nNp1.y) + dy)*inv,
					((it.pos_corr[2] - nNp1.z) + dz)*inv);
				const orc_f4 be = boundelem[neib_index];
				const v3 ns = v3_make(be.x, be.y, be.z);
				v3 q_vb[3];
				calc_vertex_rel_pos(q_vb, ns, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
					vertPos2 + 2*(size_t)neib_index, slength);
				const v3 gN = v3_scale(ns, grad_gamma_wendland(slength, qN, q_vb, ns));
				const v3 gNp1 = v3_scale(ns, grad_gamma_wendland(slength, qNp1, q_vb, ns));
				gGamDotR += 0.5f*v3_dot(v3_add(gN, gNp1), v3_sub(qNp1, qN));
				gGam = v3_add(gGam, gNp1);
			}
			gGamDotR *= slength;
		}
		/* densitySumBoundaryDevice (:606-655) */
		const orc_f4 gGamN = oldGGam[index];
		orc_f4 g = { gGam.x, gGam.y, gGam.z, gGamN.w + gGamDotR };
		const int fl = FLUID_NUM(info);
		const float rho = (gGamN.w*physical_density(p, oldVel[index].w, fl) + forces[index].w)/g.w;
		if (g.w > 1.0f || sqrtf(g.x*g.x + g.y*g.y + g.z*g.z)*slength < 1e-10f)
			g.w = 1.0f;
		else if (g.w < 0.1f)
			g.w = 0.1f;
		newVel[index].w = rho/p->rho0[fl] - 1.0f;          /* numerical_density */
		newGGam[index] = g;
	}
}

/* ---- SA_BOUNDARY with MOVING bodies (ENABLE_MOVING_BODIES; prescribed motion) -------------------------------------------------
 * What changes against solid walls at rest (row f-2 of SURVEY.md 8):
 *   - BUFFER_BOUNDELEMENTS is a state buffer: the Euler step rotates the normals of the moving segments and vertices
 *     (update_normals, src/cuda/euler_kernel.def:237-254; applyrot euler_kernel.cu:67-74);
 *   - the boundary terms of the density summation see the elements where they were and where they are: positions AND normals of
 *     step n and of the new state (computeDensitySumBoundaryTerms, src/cuda/density_sum_kernel.cu:419-478: nsN from the old
 *     BOUNDELEMENTS, nsNp1 from the new, the vertices' offsets re-derived with the new normal);
 *   - gamma of the VERTEX particles is integrated like that of the fluid instead of copied (density_sum_impl /
 *     integrate_gamma_impl, src/cuda/euler.cu:112-160,202-290): with dynamic gamma gamma^{n+1} = gamma^n + the boundary terms
 *     (integrateGammaDeviceFunc :669-685), with ENABLE_GAMMA_QUADRATURE the quadrature of Gamma<PT_VERTEX> against the NEW
 *     elements (:687-765; orc_sa_integrate_gamma_quadrature with cptype PT_VERTEX and the new BOUNDELEMENTS);
 *   - the segment condition takes the velocity of a moving segment as the mean of its vertices' and re-derives gamma in every
 *     step (moving_vertex_contrib boundary_conditions_kernel.cu:781-800, calcGam :1467; orc_sa_segment_bc has both). */
void orc_sa_update_normals(const orc_params *p, orc_f4 *newBoundElem, const orc_f4 *oldBoundElem, const orc_info *infoArray,
	uint32_t numParticles)
{
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		orc_f4 normal = oldBoundElem[index];
		if (MOVING(info) && (BOUNDARY(info) || VERTEX(info))) {
			const float *rot = p->rbsteprot[OBJECT_NUM(info)];
			const float rx = normal.x, ry = normal.y, rz = normal.z;      /* applyrot(rot, normal, normal) */
			normal.x += (rot[0] - 1.0f)*rx + rot[1]*ry + rot[2]*rz;
			normal.y += rot[3]*rx + (rot[4] - 1.0f)*ry + rot[5]*rz;
			normal.z += rot[6]*rx + rot[7]*ry + (rot[8] - 1.0f)*rz;
		}
		newBoundElem[index] = normal;
	}
}

/* the boundary terms of one particle between the old and the new state, elements old and new (:419-478) */
static void density_sum_boundary_terms_moving(const orc_params *p, uint32_t index, const orc_f4 *oldPos, const orc_f4 *newPos,
	const orc_f4 *boundelemOld, const orc_f4 *boundelemNew, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList, v3 *gGamOut, float *gGamDotROut)
{
	const float slength = p->slength;
	const orc_f4 posN = oldPos[index], posNp1 = newPos[index];
	int gridPos[3];
	orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
	const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
	float gGamDotR = 0.0f;
	v3 gGam = v3_make(0.0f, 0.0f, 0.0f);
	neib_iter it;
	uint32_t neib_index;
	neib_iter_init(&it, p, PT_BOUNDARY, index, &posN, gridPos, cellStart, neibsList);
	while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
		const orc_f4 nN = oldPos[neib_index];
		if (INACTIVE(nN)) continue;
		const orc_f4 nNp1 = newPos[neib_index];
		const float inv = 1.0f/slength;
		const v3 qN = v3_make((it.pos_corr[0] - nN.x)*inv, (it.pos_corr[1] - nN.y)*inv, (it.pos_corr[2] - nN.z)*inv);
		const v3 qNp1 = v3_make(((it.pos_corr[0] - nNp1.x) + dx)*inv, ((it.pos_corr[1] - nNp1.y) + dy)*inv,
			((it.pos_corr[2] - nNp1.z) + dz)*inv);
		const orc_f4 beN = boundelemOld[neib_index], beNp1 = boundelemNew[neib_index];
		const v3 nsN = v3_make(beN.x, beN.y, beN.z), nsNp1 = v3_make(beNp1.x, beNp1.y, beNp1.z);
		v3 q_vb[3];
		calc_vertex_rel_pos(q_vb, nsN, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
			vertPos2 + 2*(size_t)neib_index, slength);
		const v3 gN = v3_scale(nsN, grad_gamma_wendland(slength, qN, q_vb, nsN));
		calc_vertex_rel_pos(q_vb, nsNp1, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
			vertPos2 + 2*(size_t)neib_index, slength);
		const v3 gNp1 = v3_scale(nsNp1, grad_gamma_wendland(slength, qNp1, q_vb, nsNp1));
		gGamDotR += 0.5f*v3_dot(v3_add(gN, gNp1), v3_sub(qNp1, qN));
		gGam = v3_add(gGam, gNp1);
	}
	*gGamOut = gGam;
	*gGamDotROut = gGamDotR*slength;
}

/* density_sum_impl<SA_BOUNDARY> with ENABLE_MOVING_BODIES (src/cuda/euler.cu:112-160): the fluid rows as orc_sa_density_sum with
 * the boundary terms above; the VERTEX rows' gamma by the same boundary terms (integrateGammaDevice<PT_VERTEX>, dynamic gamma);
 * the BOUNDARY rows are not written by any of the three kernels (the segment condition re-derives them): they keep what the
 * caller put into newGGam. */
void orc_sa_density_sum_moving(const orc_params *p, orc_f4 *newVel, orc_f4 *newGGam, orc_f4 *forces,
	const orc_f4 *oldPos, const orc_f4 *newPos, const orc_f4 *oldVel, const orc_f4 *oldGGam,
	const orc_f4 *boundelemOld, const orc_f4 *boundelemNew,
	const float *vertPos0, const float *vertPos1, const float *vertPos2, const orc_info *infoArray,
	const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd)
{
	const float kr = 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const float slength = p->slength;
#pragma omp parallel for schedule(dynamic, 256)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (VERTEX(info)) {
			v3 gGam; float gGamDotR;
			density_sum_boundary_terms_moving(p, index, oldPos, newPos, boundelemOld, boundelemNew, vertPos0, vertPos1, vertPos2,
				hashArray, cellStart, neibsList, &gGam, &gGamDotR);
			const orc_f4 g = { gGam.x, gGam.y, gGam.z, oldGGam[index].w + gGamDotR };
			newGGam[index] = g;
			continue;
		}
		if (!FLUID(info)) continue;
		const orc_f4 posN = oldPos[index], posNp1 = newPos[index];
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
		float sumPmwN = 0.0f, sumPmwNp1 = 0.0f;
		for (int nptype = PT_FLUID; nptype <= PT_VERTEX; nptype += 2) {
			neib_iter it;
			uint32_t neib_index;
			neib_iter_init(&it, p, nptype, index, &posN, gridPos, cellStart, neibsList);
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 nN = oldPos[neib_index];
				if (INACTIVE(nN)) continue;
				const orc_f4 nNp1 = newPos[neib_index];
				const float rx = it.pos_corr[0] - nN.x, ry = it.pos_corr[1] - nN.y, rz = it.pos_corr[2] - nN.z;
				const float qx = (it.pos_corr[0] - nNp1.x) + dx, qy = (it.pos_corr[1] - nNp1.y) + dy, qz = (it.pos_corr[2] - nNp1.z) + dz;
				const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
				sumPmwN -= nN.w*W_c(p->kerneltype, rN, slength, wcoeff, wsub);
				const float rNp1 = sqrtf(qx*qx + qy*qy + qz*qz);
				if (rNp1 < p->influenceradius)
					sumPmwNp1 += nN.w*W_c(p->kerneltype, rNp1, slength, wcoeff, wsub);
			}
		}
		forces[index].w = sumPmwNp1 + sumPmwN + 0.0f;
		v3 gGam; float gGamDotR;
		density_sum_boundary_terms_moving(p, index, oldPos, newPos, boundelemOld, boundelemNew, vertPos0, vertPos1, vertPos2,
			hashArray, cellStart, neibsList, &gGam, &gGamDotR);
		const orc_f4 gGamN = oldGGam[index];
		orc_f4 g = { gGam.x, gGam.y, gGam.z, gGamN.w + gGamDotR };
		const int fl = FLUID_NUM(info);
		const float rho = (gGamN.w*physical_density(p, oldVel[index].w, fl) + forces[index].w)/g.w;
		if (g.w > 1.0f || sqrtf(g.x*g.x + g.y*g.y + g.z*g.z)*slength < 1e-10f)
			g.w = 1.0f;
		else if (g.w < 0.1f)
			g.w = 0.1f;
		newVel[index].w = rho/p->rho0[fl] - 1.0f;
		newGGam[index] = g;
	}
}

/* computeDensityDiffusionDevice<.., BREZZI, SA_BOUNDARY, PT_FLUID> (src/cuda/forces_kernel.def:4515-4560): the Brezzi term over
 * the fluid neighbours (:1766-1783; boundary elements only contribute at pressure inlets), divided by gamma and rho0,
 * written to forces.w; updateDensityDevice (src/cuda/euler_kernel.cu:116-136) then adds forces.w dt to the density */
void orc_sa_density_diffusion(const orc_params *p, orc_f4 *forces, const orc_f4 *posArray, const orc_f4 *velArray,
	const orc_f4 *gGamArray, const orc_info *infoArray, const uint32_t *hashArray, const uint32_t *cellStart,
	const uint16_t *neibsList, uint32_t particleRangeEnd, float dt)
{
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, 2.0f);
#pragma omp parallel for schedule(dynamic, 256)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!FLUID(info)) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		const orc_f4 vel = velArray[index];
		const int fl = FLUID_NUM(info);
		const float rho = physical_density(p, vel.w, fl);
		const float pres = orc_P(p, vel.w, fl);
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		float DrDt = 0.0f;
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const float r = sqrtf(rx*rx + ry*ry + rz*rz);
			if (INACTIVE(npos)) continue;
			if (r >= p->influenceradius) continue;
			const orc_f4 nvel = velArray[neib_index];
			const int nfl = FLUID_NUM(infoArray[neib_index]);
			const float neib_rho = physical_density(p, nvel.w, nfl);
			const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
			const float gdotr = p->gravity[0]*rx + p->gravity[1]*ry + p->gravity[2]*rz;
			float nDrDt = 0.0f;
			nDrDt += p->densityDiffCoeff*((2.0f/(rho + neib_rho))*(pres - orc_P(p, nvel.w, nfl)) - gdotr)*npos.w/neib_rho*f*dt*2.0f*rho;
			DrDt += nDrDt;
		}
		DrDt /= gGamArray[index].w;
		forces[index].w = DrDt/p->rho0[fl];
	}
}

/* gamma part of dtreduce with dynamic gamma (src/cuda/forces.cu:576-585): dt_gamma = 0.001 / max(max CFL_gamma, 1e-5/dt) */
float orc_sa_gamma_dt(float dt, float max_gamma_cfl)
{
	const float maxcfl = fmaxf(max_gamma_cfl, 1e-5f/dt);
	const float dt_gam = 0.001f/maxcfl;
	return dt_gam < dt ? dt_gam : dt;
}

/* ====================================================================================================
 * Open boundaries of SA_BOUNDARY (the half that is NOT built): GROUNDWORK ONLY.
 * The leaf functions and the initialisation kernels of ENABLE_INLET_OUTLET, restated so that the engines of a later round
 * have their checker; the product (libsphx) has no counterpart of anything below and answers SPHX_ERR_UNSUPPORTED.
 * Further down, added later in the round: the IO branches of saSegment/VertexBoundaryConditions (:1427-1545, 2197-2270) with
 * particle creation, findOutgoingSegment (:1647), the IO branches of density_sum and of the forces, the water depth and the
 * Brezzi term of pressure-driven segments.  Not restated: the download / maximum over devices / upload of the water depth.  Parity unpinned (no reference fixture
 * holds these; the known answers are in tests/test_sa_io_oracle.py).
 * ==================================================================================================== */
#define FG_INLET             (PART_FLAG_START << 2)      /* src/particleinfo.h:153-156 */
#define FG_OUTLET            (PART_FLAG_START << 3)
#define FG_VELOCITY_DRIVEN   (PART_FLAG_START << 4)
#define FG_CORNER            (PART_FLAG_START << 5)
#define IO_BOUNDARY(f)       ((f).x & (FG_INLET | FG_OUTLET))      /* :222 */
#define VEL_IO(f)            ((f).x & FG_VELOCITY_DRIVEN)           /* :227 */
#define CORNER(f)            ((f).x & FG_CORNER)                    /* :241 */
#define MAXNEIBVERTS 30                                             /* boundary_conditions_kernel.cu:1977 */

/* Riemann celerity and its inverse, src/cuda/phys_core.cu:114-127 (__powf -> powf; the double constants of RHOR as written) */
float orc_R(const orc_params *p, float rho_tilde, int i)
{
	const float rho_ratio = rho_tilde + 1.0f;
	return 2.0f/(p->gammacoeff[i] - 1.0f)*p->sscoeff[i]*powf(rho_ratio, 0.5f*p->gammacoeff[i] - 0.5f);
}
float orc_RHOR(const orc_params *p, float r, int i)
{
	return (float)(powf((float)((p->gammacoeff[i] - 1.)*r/(2.*p->sscoeff[i])), (float)(2./(p->gammacoeff[i] - 1.))) - 1.0);
}

/* calculateIOboundaryCondition, src/cuda/boundary_conditions_kernel.cu:111-200: velocity imposed -> density from the Riemann
 * invariant; pressure (density) imposed -> normal velocity from it.  eulerVel: in {imposed u, imposed rho~}, out the condition */
void orc_io_boundary_condition(const orc_params *p, float eulerVel[4], int velocity_driven, int fluid, float rhoInt, float rhoExt,
	const float uInt[3], float unInt, float unExt, const float normal[3])
{
	const int a = fluid;
	const float rInt = orc_R(p, rhoInt, a);
	if (velocity_driven) {
		float riemannR = 0.0f;
		if (unExt <= unInt)      /* expansion wave */
			riemannR = rInt + (unExt - unInt);
		else {                   /* shock wave */
			const float riemannRho = orc_RHO(p, orc_P(p, rhoInt, a) + physical_density(p, rhoInt, a)*unInt*(unInt - unExt), a);
			riemannR = orc_R(p, riemannRho, a);
			const float riemannC = orc_soundSpeed(p, riemannRho, a);
			const float lambda = unExt + riemannC;
			const float cInt = orc_soundSpeed(p, rhoInt, a);
			const float lambdaInt = unInt + cInt;
			if (lambda <= lambdaInt)   /* a contact discontinuity */
				riemannR = rInt;
		}
		eulerVel[3] = orc_RHOR(p, riemannR, a);
	} else {
		float flux = 0.0f;
		const float cExt = orc_soundSpeed(p, rhoExt, a);
		const float cInt = orc_soundSpeed(p, rhoInt, a);
		const float lambdaInt = unInt + cInt;
		const float rExt = orc_R(p, rhoExt, a);
		if (rhoExt <= rhoInt) {      /* expansion wave */
			flux = unInt + (rExt - rInt);
			float lambda = flux + cExt;
			if (lambda > lambdaInt) {   /* shock wave */
				flux = (orc_P(p, rhoInt, a) - orc_P(p, rhoExt, a))/(physical_density(p, rhoInt, a)*fmaxf(unInt, 1e-5f*p->sscoeff[a])) + unInt;
				if (fabsf(flux) > p->sscoeff[a]*0.1f)      /* unInt was too small */
					flux = unInt;
				lambda = flux + cExt;
				if (lambda <= lambdaInt)   /* contact discontinuity */
					flux = unInt;
			}
		} else {                     /* shock wave */
			flux = (orc_P(p, rhoInt, a) - orc_P(p, rhoExt, a))/(physical_density(p, rhoInt, a)*fmaxf(unInt, 1e-5f*p->sscoeff[a])) + unInt;
			if (fabsf(flux) > p->sscoeff[a]*0.1f)
				flux = unInt;
			float lambda = flux + cExt;
			if (lambda <= lambdaInt) {   /* expansion wave */
				flux = unInt + (rExt - rInt);
				lambda = flux + cExt;
				if (lambda > lambdaInt)    /* contact discontinuity */
					flux = unInt;
			}
		}
		eulerVel[0] = eulerVel[1] = eulerVel[2] = 0.0f;
		if (rhoExt < 0.0f)           /* a negative imposed pressure only lets fluid out */
			flux = fminf(flux, 0.0f);
		if (flux < 0.0f) {           /* outflow: dv/dn = 0, the normal component removed */
			const float un = uInt[0]*normal[0] + uInt[1]*normal[1] + uInt[2]*normal[2];
			for (int k = 0; k < 3; ++k) eulerVel[k] = uInt[k] - un*normal[k];
		}
		for (int k = 0; k < 3; ++k) eulerVel[k] += normal[k]*flux;
		eulerVel[3] = rhoExt;
	}
}

/* getMassRepartitionFactor, :213-283: the share of each of a segment's three vertices in a mass that sits at the origin of
 * vertexRelPos (the segment's centre, or a particle that crosses the segment): sub-triangle areas, clipped to the segment */
void orc_mass_repartition(const float vrp[9], const float n[3], float beta[3])
{
	const v3 nn = { n[0], n[1], n[2] };
	const v3 q0 = { vrp[0], vrp[1], vrp[2] }, q1 = { vrp[3], vrp[4], vrp[5] }, q2 = { vrp[6], vrp[7], vrp[8] };
	const v3 v01 = v3_sub(q0, q1), v02 = v3_sub(q0, q2);
	v3 p0 = v3_sub(q0, v3_scale(nn, v3_dot(q0, nn)));
	v3 p1 = v3_sub(q1, v3_scale(nn, v3_dot(q1, nn)));
	v3 p2 = v3_sub(q2, v3_scale(nn, v3_dot(q2, nn)));
	const float refSurface = (float)(0.5*v3_dot(v3_cross(v01, v02), nn));
	const v3 v21 = v3_sub(q2, q1);
	float surface0 = (float)(0.5*v3_dot(v3_cross(p2, v21), nn));
	float surface1 = (float)(0.5*v3_dot(v3_cross(p0, v02), nn));
	float surface2 = (float)(-0.5*v3_dot(v3_cross(p1, v01), nn));
	if (surface0 < 0. && surface2 < 0.) { surface0 = 0.; surface1 = refSurface; surface2 = 0.; }          /* clipped to v1 */
	else if (surface0 < 0. && surface1 < 0.) { surface0 = 0.; surface1 = 0.; surface2 = refSurface; }      /* clipped to v2 */
	else if (surface1 < 0. && surface2 < 0.) { surface0 = refSurface; surface1 = 0.; surface2 = 0.; }      /* clipped to v0 */
	else if (surface0 < 0.) {
		const float coef = (float)(surface0/(0.5*v3_dot(v3_cross(p0, v21), nn)));
		p1 = v3_sub(p1, v3_scale(p0, coef));
		p0 = v3_scale(p0, (float)(1. - coef));
		surface0 = 0.;
		surface1 = (float)(0.5*v3_dot(v3_cross(p0, v02), nn));
		surface2 = (float)(-0.5*v3_dot(v3_cross(p1, v01), nn));
	} else if (surface1 < 0.) {
		const float coef = (float)(surface1/(0.5*v3_dot(v3_cross(p1, v02), nn)));
		p2 = v3_sub(p2, v3_scale(p1, coef));
		p1 = v3_scale(p1, (float)(1. - coef));
		surface0 = (float)(0.5*v3_dot(v3_cross(p2, v21), nn));
		surface1 = 0.;
		surface2 = (float)(-0.5*v3_dot(v3_cross(p1, v01), nn));
	} else if (surface2 < 0.) {
		const float coef = (float)(-surface2/(0.5*v3_dot(v3_cross(p2, v01), nn)));
		p0 = v3_sub(p0, v3_scale(p2, coef));
		p2 = v3_scale(p2, (float)(1. - coef));
		surface0 = (float)(0.5*v3_dot(v3_cross(p2, v21), nn));
		surface1 = (float)(0.5*v3_dot(v3_cross(p0, v02), nn));
		surface2 = 0.;
	}
	beta[0] = surface0/refSurface; beta[1] = surface1/refSurface; beta[2] = surface2/refSurface;
}

/* saIdentifyCornerVerticesDevice, :2319-2362: a vertex of an open boundary that also belongs to a segment which is not of that
 * open boundary gets FG_CORNER */
void orc_sa_identify_corner_vertices(const orc_params *p, const orc_f4 *posArray, orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *vertices, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles)
{
	for (uint32_t index = 0; index < numParticles; ++index) {
		orc_info info = infoArray[index];
		if (!(VERTEX(info) && IO_BOUNDARY(info))) continue;
		const uint32_t obj = OBJECT_NUM(info);
		const orc_f4 pos = posArray[index];
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_info neib_info = infoArray[neib_index];
			if (!(obj == OBJECT_NUM(neib_info) && IO_BOUNDARY(neib_info)) &&
			    has_vertex(vertices + 4*(size_t)neib_index, orc_info_id(info))) {
				info.x |= FG_CORNER;
				infoArray[index] = info;
				break;
			}
		}
	}
}

/* the ids of the other vertices of the open-boundary segments a vertex belongs to (shared by the two kernels below) */
static uint32_t io_adjacent_vertex_ids(const orc_params *p, uint32_t index, const orc_f4 *pos, const orc_info *infoArray,
	const uint32_t *hashArray, const uint32_t *vertices, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t *ids)
{
	const uint32_t my_id = orc_info_id(infoArray[index]);
	uint32_t count = 0;
	int gridPos[3];
	orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
	neib_iter it;
	neib_iter_init(&it, p, PT_BOUNDARY, index, pos, gridPos, cellStart, neibsList);
	uint32_t neib_index;
	while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
		if (!IO_BOUNDARY(infoArray[neib_index])) continue;
		const uint32_t *nv = vertices + 4*(size_t)neib_index;
		if (!has_vertex(nv, my_id)) continue;
		for (int k = 0; k < 3; ++k)
			if (my_id != nv[k] && count < MAXNEIBVERTS) ids[count++] = nv[k];
	}
	return count;
}

/* initIOmass_vertexCountDevice, :1999-2064: per non-corner open-boundary vertex, how many non-corner vertices share a segment
 * with it (one count per shared segment), into forces.w */
void orc_sa_init_io_mass_vertex_count(const orc_params *p, const uint32_t *vertices, const uint32_t *hashArray,
	const orc_info *infoArray, const uint32_t *cellStart, const uint16_t *neibsList, orc_f4 *forces, uint32_t numParticles)
{
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		if (!(VERTEX(info) && IO_BOUNDARY(info) && !CORNER(info))) continue;
		const orc_f4 pos = { 0.0f, 0.0f, 0.0f, 0.0f };
		uint32_t ids[MAXNEIBVERTS];
		const uint32_t nids = io_adjacent_vertex_ids(p, index, &pos, infoArray, hashArray, vertices, cellStart, neibsList, ids);
		uint32_t vertexCount = 0;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		neib_iter_init(&it, p, PT_VERTEX, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_info neib_info = infoArray[neib_index];
			for (uint32_t j = 0; j < nids; ++j)
				if (orc_info_id(neib_info) == ids[j] && !CORNER(neib_info)) vertexCount += 1;
		}
		forces[index].w = (float)vertexCount;
	}
}

/* initIOmassDevice, :2078-2172: the open-boundary vertices start from half a fluid particle's mass -- vertices with an odd id
 * take the difference from their even-numbered partners along the segments they share, equal shares */
void orc_sa_init_io_mass(const orc_params *p, const orc_f4 *oldPos, const orc_f4 *forces, const uint32_t *vertices,
	const uint32_t *hashArray, const orc_info *infoArray, const uint32_t *cellStart, const uint16_t *neibsList,
	orc_f4 *newPos, uint32_t numParticles, float deltap)
{
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		const orc_f4 pos = oldPos[index];
		newPos[index] = pos;
		if (!(VERTEX(info) && IO_BOUNDARY(info) && !CORNER(info))) continue;
		const int getMass = (int)(orc_info_id(info) % 2u);
		float massChange = 0.0f;
		const float refMass = 0.5f*deltap*deltap*deltap*p->rho0[FLUID_NUM(info)];
		const float massDiff = refMass - pos.w;
		const float vertexCount = forces[index].w;
		uint32_t ids[MAXNEIBVERTS];
		const uint32_t nids = io_adjacent_vertex_ids(p, index, &pos, infoArray, hashArray, vertices, cellStart, neibsList, ids);
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		neib_iter_init(&it, p, PT_VERTEX, index, &pos, gridPos, cellStart, neibsList);
		uint32_t neib_index;
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_info neib_info = infoArray[neib_index];
			for (uint32_t j = 0; j < nids; ++j) {
				if (orc_info_id(neib_info) != ids[j]) continue;
				const int neib_getMass = (int)(orc_info_id(neib_info) % 2u);
				if (getMass != neib_getMass && !CORNER(neib_info)) {
					if (getMass) {
						if (massDiff > 0.0f) massChange += massDiff/vertexCount;
					} else {
						const float neib_massDiff = refMass - oldPos[neib_index].w;
						if (neib_massDiff > 0.0f) massChange -= neib_massDiff/forces[neib_index].w;
					}
				}
			}
		}
		newPos[index].w += massChange;
	}
}

/* disableOutgoingPartsDevice, :2374-2398: a fluid particle marked by findOutgoingSegment (vertexinfo .x | .y != 0) is disabled and
 * its mark cleared */
void orc_disable_outgoing_parts(orc_f4 *posArray, uint32_t *vertices, const orc_info *infoArray, uint32_t numParticles)
{
	for (uint32_t index = 0; index < numParticles; ++index) {
		if (PART_TYPE(infoArray[index]) != PT_FLUID || !ACTIVE(posArray[index])) continue;
		uint32_t *v = vertices + 4*(size_t)index;
		if ((v[0] | v[1]) != 0u) {
			posArray[index].w = NAN;      /* disable_particle */
			v[0] = v[1] = v[2] = v[3] = 0u;
		}
	}
}

/* saSegmentBoundaryConditionsDevice with has_io (:1427-1520; laminar, not repacking): the solid-wall condition of sa_segment_bc_impl
 * for the segments of solid walls (whose Eulerian velocity is cleared, impose_solid_eulerVel :1287-1293) and, for the segments of an
 * open boundary, the Shepard means of the fluid's velocity (+ its Eulerian velocity) and pressure (io_fluid_contrib :852-866)
 * completed by the Riemann-invariant condition (impose_io_bc :1362-1413).  eulerVelArray: in, what IMPOSE_OPEN_BOUNDARY_CONDITION
 * set on the open-boundary rows; out, the completed condition; the density of an open-boundary segment is its Eulerian density.
 * GROUNDWORK, as the section above: nothing in the product calls for it yet. */
void orc_sa_segment_bc_io(const orc_params *p, orc_f4 *velArray, orc_f4 *gGamArray, orc_f4 *eulerVelArray, const orc_f4 *posArray,
	const uint32_t *vertices, const orc_f4 *boundelement, const orc_info *infoArray, const uint32_t *hashArray,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd, int step)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const int has_moving = (p->simflags & ORC_ENABLE_MOVING_BODIES) != 0;
	if (step == -1) step = 0;
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!BOUNDARY(info)) continue;
		const orc_f4 pos = posArray[index];
		const orc_f4 normal = boundelement[index];
		const uint32_t *verts = vertices + 4*(size_t)index;
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		float sumpWall = 0.0f, shepard_div = 0.0f, sump = 0.0f;
		float sumvel[3] = { 0.0f, 0.0f, 0.0f };
		orc_f4 gGam = { 0.0f, 0.0f, 0.0f, gGamArray[index].w };
		orc_f4 vel = { 0.0f, 0.0f, 0.0f, 0.0f };
		const int calcGam = has_moving || !isfinite(gGam.w) || step == 0;
		if (calcGam) gGam.w = 0.0f;
		orc_f4 eulerVel = { 0.0f, 0.0f, 0.0f, 0.0f };      /* eulervel_pout, IO constructor :488-505 */
		if (IO_BOUNDARY(info)) {
			eulerVel = eulerVelArray[index];
			if (VEL_IO(info)) eulerVel.w = 0.0f;
		}
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_VERTEX, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			if (INACTIVE(posArray[neib_index])) continue;
			if (!has_vertex(verts, orc_info_id(infoArray[neib_index]))) continue;
			if (has_moving && MOVING(info)) {
				const orc_f4 nv = velArray[neib_index];
				vel.x += nv.x; vel.y += nv.y; vel.z += nv.z;
			}
			if (calcGam) {
				const orc_f4 g = gGamArray[neib_index];
				gGam.x += g.x; gGam.y += g.y; gGam.z += g.z; gGam.w += g.w;
			}
		}
		if (calcGam) {
			const float inv = 1.0f/3;
			gGam.x *= inv; gGam.y *= inv; gGam.z *= inv; gGam.w *= inv;
			gGamArray[index] = gGam;
			gGam.w = fmaxf(gGam.w, 1e-5f);
		}
		vel.x /= 3; vel.y /= 3; vel.z /= 3;
		const int fl = FLUID_NUM(info);
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			if (INACTIVE(npos)) continue;
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const sa_ndata n = sa_fluid_ndata(p, wcoeff, wsub, velArray, infoArray, neib_index, rx, ry, rz, npos.w);
			if (!(n.r < p->influenceradius && (normal.x*rx + normal.y*ry + normal.z*rz) < 0.0f)) continue;
			const float gdot = p->gravity[0]*rx + p->gravity[1]*ry + p->gravity[2]*rz;
			sumpWall += fmaxf(n.press + physical_density(p, n.vel.w, fl)*gdot, 0.0f)*n.w;
			if (IO_BOUNDARY(info)) {      /* io_fluid_contrib, segments */
				const orc_f4 ne = eulerVelArray[neib_index];
				sumvel[0] += n.w*(n.vel.x + ne.x); sumvel[1] += n.w*(n.vel.y + ne.y); sumvel[2] += n.w*(n.vel.z + ne.z);
				sump += n.w*fmaxf(0.0f, n.press);
			}
			shepard_div += n.w;
		}
		if (IO_BOUNDARY(info)) {      /* impose_io_bc */
			if (shepard_div > 0.1f*gGam.w) {
				sumvel[0] /= shepard_div; sumvel[1] /= shepard_div; sumvel[2] /= shepard_div;
				sump /= shepard_div;
				vel.w = orc_RHO(p, sump, fl);
				if (!VEL_IO(info)) { const orc_f4 z = { 0.0f, 0.0f, 0.0f, 0.0f }; eulerVelArray[index] = z; }
			} else {
				sump = 0.0f;
				if (VEL_IO(info)) {
					sumvel[0] = eulerVel.x; sumvel[1] = eulerVel.y; sumvel[2] = eulerVel.z;
					vel.w = 0.0f;
				} else {
					sumvel[0] = sumvel[1] = sumvel[2] = 0.0f;
					vel.w = eulerVelArray[index].w;
				}
			}
			const float nrm[3] = { normal.x, normal.y, normal.z };
			const float unInt = sumvel[0]*nrm[0] + sumvel[1]*nrm[1] + sumvel[2]*nrm[2];
			const float unExt = eulerVel.x*nrm[0] + eulerVel.y*nrm[1] + eulerVel.z*nrm[2];
			float ev[4] = { eulerVel.x, eulerVel.y, eulerVel.z, eulerVel.w };
			orc_io_boundary_condition(p, ev, VEL_IO(info) != 0, fl, vel.w, eulerVel.w, sumvel, unInt, unExt, nrm);
			const orc_f4 out = { ev[0], ev[1], ev[2], ev[3] };
			eulerVelArray[index] = out;
			vel.w = out.w;
		} else {                       /* impose_solid_bc<true> */
			shepard_div = fmaxf(shepard_div, 0.1f*gGam.w);
			vel.w = orc_RHO(p, sumpWall/shepard_div, fl);
			const orc_f4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
			eulerVelArray[index] = z;
		}
		velArray[index] = vel;
	}
}

/* findOutgoingSegmentDevice, :1647-1750: a fluid particle that is behind an open-boundary segment and moving out relative to it
 * is marked with that segment's vertices (the closest such segment within the influence radius), its share of each vertex
 * (getMassRepartitionFactor at the particle's position) and its mass go where grad gamma used to be: the vertices take the mass
 * over in the vertex conditions of the last step, disableOutgoingParts removes the particle.  GROUNDWORK (see above). */
void orc_find_outgoing_segment(const orc_params *p, const orc_f4 *posArray, const orc_f4 *velArray, uint32_t *vertices,
	orc_f4 *gGam, const float *vertPos0, const float *vertPos1, const float *vertPos2, const orc_f4 *boundelement,
	const orc_info *infoArray, const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t numParticles, float influenceradius)
{
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		if (PART_TYPE(info) != PT_FLUID) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		if (vertices[4*(size_t)index] | vertices[4*(size_t)index + 1]) continue;      /* already marked ("this shouldn't happen") */
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const orc_f4 vel = velArray[index];
		float r2_min = influenceradius*influenceradius;
		uint32_t index_min = UINT_MAX;
		v3 normal_min = v3_make(0.0f, 0.0f, 0.0f), relPos_min = normal_min;
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			if (!IO_BOUNDARY(infoArray[neib_index])) continue;
			const orc_f4 npos = posArray[neib_index];
			const v3 relPos = v3_make(it.pos_corr[0] - npos.x, it.pos_corr[1] - npos.y, it.pos_corr[2] - npos.z);
			const orc_f4 nrm = boundelement[neib_index];
			const v3 normal = v3_make(nrm.x, nrm.y, nrm.z);
			const orc_f4 nvel = velArray[neib_index];
			const v3 relVel = v3_make(vel.x - nvel.x, vel.y - nvel.y, vel.z - nvel.z);
			const float r2 = v3_sqlen(relPos);
			if (r2 < r2_min && v3_dot(normal, relPos) <= 0.0f && v3_dot(normal, relVel) < 0.0f) {
				r2_min = r2; index_min = neib_index; normal_min = normal; relPos_min = relPos;
			}
		}
		if (index_min == UINT_MAX) continue;
		v3 vx[3];
		calc_vertex_rel_pos(vx, normal_min, vertPos0 + 2*(size_t)index_min, vertPos1 + 2*(size_t)index_min, vertPos2 + 2*(size_t)index_min, 1.0f);
		for (int k = 0; k < 3; ++k) vx[k] = v3_sub(relPos_min, vx[k]);      /* relative to the particle, not to the barycentre */
		const float vrp[9] = { vx[0].x, vx[0].y, vx[0].z, vx[1].x, vx[1].y, vx[1].z, vx[2].x, vx[2].y, vx[2].z };
		const float nn[3] = { normal_min.x, normal_min.y, normal_min.z };
		float beta[3];
		orc_mass_repartition(vrp, nn, beta);
		for (int k = 0; k < 4; ++k) vertices[4*(size_t)index + k] = vertices[4*(size_t)index_min + k];
		const orc_f4 w = { beta[0], beta[1], beta[2], pos.w };
		gGam[index] = w;
	}
}

/* saVertexBoundaryConditionsDevice with has_io (:2197-2252; laminar, not repacking): the solid-wall density of every vertex
 * (sa_vertex_bc_impl), and for the vertices of an open boundary that are not corners (impose_vertex_io_bc :1168-1252):
 *   - the Shepard means of the fluid's velocity (+ Eulerian velocity) and pressure, completed by the Riemann condition into the
 *     vertex's Eulerian velocity and density (io_fluid_contrib for vertices :868-909);
 *   - the mass flux through the adjacent open-boundary segments, each segment's flux rho A (eulerVel . n) shared out by
 *     getMassRepartitionFactor at the segment's centre (io_boundary_contrib :937-988), integrated into the vertex mass over dt
 *     (steps 1 and 2), clipped to +/- 2 reference masses (the second clip, by refMass * normal.w, is written against a vertex normal
 *     whose .w is NaN, computeVertexNormalDevice :1830: fminf / fmaxf return the other operand -- reproduced, a no-op);
 *   - in the last step (2): the mass of the fluid particles findOutgoingSegment marked, by their share for this vertex, and one
 *     new fluid particle of the reference mass at the vertex when it holds more than half of it, the flux is positive and the
 *     imposed normal velocity (or, at a pressure boundary, the density) is (generate_new_particles :1101-1159, createNewFluidParticle
 *     :73-104): appended at *newNumParticles, id nextIDs[vertex] which then advances by numOpenVertices.
 * newPos: the caller's copy of posArray, updated in the rows of the open-boundary vertices and extended by the clones (as the
 * other clone* arrays are: vel, gGam, eulerVel, forces, vertices, boundelement, info, hash, nextIDs have room for totParticles rows).
 * GROUNDWORK (see above). */
void orc_sa_vertex_bc_io(const orc_params *p, orc_f4 *velArray, const orc_f4 *posArray, orc_f4 *newPos, orc_f4 *gGamArray,
	orc_f4 *eulerVelArray, orc_f4 *forces, uint32_t *vertices, orc_f4 *boundelement, const float *vertPos0, const float *vertPos1,
	const float *vertPos2, orc_info *infoArray, uint32_t *hashArray, uint32_t *nextIDs, uint32_t *newNumParticles,
	const uint32_t *cellStart, const uint16_t *neibsList, uint32_t numParticles, uint32_t totParticles,
	float deltap, float dt, int step, uint32_t numOpenVertices)
{
	const float kr = (p->kerneltype == ORC_GAUSSIAN) ? 3.0f : 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	if (step == -1) step = 0;
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		if (!VERTEX(info)) continue;
		const orc_f4 pos = posArray[index];
		const float gam = gGamArray[index].w;
		const int fl = FLUID_NUM(info);
		const int io = IO_BOUNDARY(info) != 0, corner = CORNER(info) != 0;
		const orc_f4 normal = boundelement[index];
		const float refMass = deltap*deltap*deltap*p->rho0[fl];
		const uint32_t my_id = orc_info_id(info);
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		float sumpWall = 0.0f, shepard_div = 0.0f, sump = 0.0f, sumMdot = 0.0f, massFluid = 0.0f;
		float sumvel[3] = { 0.0f, 0.0f, 0.0f };
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			if (INACTIVE(npos)) continue;
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const sa_ndata n = sa_fluid_ndata(p, wcoeff, wsub, velArray, infoArray, neib_index, rx, ry, rz, npos.w);
			if (!(n.r < p->influenceradius)) continue;
			const float gdot = p->gravity[0]*rx + p->gravity[1]*ry + p->gravity[2]*rz;
			sumpWall += fmaxf(n.press + physical_density(p, n.vel.w, fl)*gdot, 0.0f)*n.w;
			shepard_div += n.w;
			if (!io) continue;             /* io_fluid_contrib, vertices */
			if (!corner) {
				const orc_f4 ne = eulerVelArray[neib_index];
				sumvel[0] += n.w*(n.vel.x + ne.x); sumvel[1] += n.w*(n.vel.y + ne.y); sumvel[2] += n.w*(n.vel.z + ne.z);
				sump += n.w*fmaxf(0.0f, n.press);
			}
			if (step == 2) {               /* a particle marked by findOutgoingSegment: its mass, by this vertex's share */
				const uint32_t *nv = vertices + 4*(size_t)neib_index;
				if ((nv[0] | nv[1]) != 0u) {
					const orc_f4 w = gGamArray[neib_index];
					const float weight = nv[0] == my_id ? w.x : nv[1] == my_id ? w.y : nv[2] == my_id ? w.z : 0.0f;
					if (weight > 0) massFluid += weight*w.w;
				}
			}
		}
		/* vertex_boundary_loop with has_io: the adjacent segments */
		neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const uint32_t *nv = vertices + 4*(size_t)neib_index;
			if (!has_vertex(nv, my_id)) continue;
			if (!io || corner) continue;   /* (a corner only sums the wall normals of its solid segments, for k-epsilon) */
			const orc_info ninfo = infoArray[neib_index];
			if (!IO_BOUNDARY(ninfo)) continue;
			const orc_f4 nn = boundelement[neib_index];
			v3 vx[3];
			calc_vertex_rel_pos(vx, v3_make(nn.x, nn.y, nn.z), vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
				vertPos2 + 2*(size_t)neib_index, -1.0f);
			const float vrp[9] = { vx[0].x, vx[0].y, vx[0].z, vx[1].x, vx[1].y, vx[1].z, vx[2].x, vx[2].y, vx[2].z };
			const float nrm[3] = { nn.x, nn.y, nn.z };
			float beta[3];
			orc_mass_repartition(vrp, nrm, beta);
			const float weight = nv[0] == my_id ? beta[0] : nv[1] == my_id ? beta[1] : nv[2] == my_id ? beta[2] : 0.0f;
			const orc_f4 ne = eulerVelArray[neib_index];
			sumMdot += physical_density(p, velArray[neib_index].w, FLUID_NUM(ninfo))*nn.w*weight*(ne.x*nn.x + ne.y*nn.y + ne.z*nn.z);
		}
		shepard_div = fmaxf(shepard_div, 0.1f*gam);
		velArray[index].w = orc_RHO(p, sumpWall/shepard_div, fl);
		if (!io || corner) continue;
		/* impose_vertex_io_bc */
		orc_f4 eulerVel = eulerVelArray[index];
		if (shepard_div > 0.1f*gam) {
			sumvel[0] /= shepard_div; sumvel[1] /= shepard_div; sumvel[2] /= shepard_div;
			sump /= shepard_div;
			const float nrm[3] = { normal.x, normal.y, normal.z };
			const float unInt = sumvel[0]*nrm[0] + sumvel[1]*nrm[1] + sumvel[2]*nrm[2];
			const float unExt = eulerVel.x*nrm[0] + eulerVel.y*nrm[1] + eulerVel.z*nrm[2];
			const float rhoInt = orc_RHO(p, sump, fl);
			float ev[4] = { eulerVel.x, eulerVel.y, eulerVel.z, eulerVel.w };
			orc_io_boundary_condition(p, ev, VEL_IO(info) != 0, fl, rhoInt, eulerVel.w, sumvel, unInt, unExt, nrm);
			eulerVel.x = ev[0]; eulerVel.y = ev[1]; eulerVel.z = ev[2]; eulerVel.w = ev[3];
		} else if (VEL_IO(info))
			eulerVel.w = 0.0f;
		else
			eulerVel.x = eulerVel.y = eulerVel.z = 0.0f;
		eulerVelArray[index] = eulerVel;
		velArray[index].w = eulerVel.w;
		orc_f4 np = pos;
		const float un = normal.x*eulerVel.x + normal.y*eulerVel.y + normal.z*eulerVel.z;
		if (step != 0) {
			np.w += dt*sumMdot;
			if (shepard_div < 0.1f*gam && sumMdot < 0.0f) np.w = 0.0f;
			np.w = fmaxf(-2.0f*refMass, fminf(2.0f*refMass, np.w));
			if (sumMdot < 0.0f || un < 1e-5f*p->sscoeff[fl]) {
				const float weightedMass = refMass*normal.w;
				np.w = fmaxf(-weightedMass, fminf(weightedMass, np.w));
			}
		}
		if (step == 2 && np.w > refMass*0.5f && sumMdot > 0 && un > 1e-5f && (VEL_IO(info) || eulerVel.w > 1e-5f)) {
			const uint32_t clone = (*newNumParticles)++;            /* createNewFluidParticle */
			if (clone < totParticles) {
				const uint32_t new_id = nextIDs[index];
				nextIDs[index] = new_id + numOpenVertices;
				orc_info ci;
				ci.x = PT_FLUID; ci.y = (uint16_t)(fl << 12); ci.z = (uint16_t)(new_id & 0xFFFFu); ci.w = (uint16_t)(new_id >> 16);
				orc_f4 cp = np;
				cp.w = refMass;
				massFluid -= cp.w;
				newPos[clone] = cp;
				infoArray[clone] = ci;
				hashArray[clone] = hashArray[index] & CELLTYPE_BITMASK;      /* calcGridHash(gridPos): the vertex's cell */
				velArray[clone] = eulerVel;
				gGamArray[clone] = gGamArray[index];
				const orc_f4 z = { 0.0f, 0.0f, 0.0f, 0.0f }, nanv = { -NAN, -NAN, -NAN, -NAN };
				eulerVelArray[clone] = z;
				forces[clone] = z;
				vertices[4*(size_t)clone] = vertices[4*(size_t)clone + 1] = vertices[4*(size_t)clone + 2] = vertices[4*(size_t)clone + 3] = 0u;
				nextIDs[clone] = UINT_MAX;
				boundelement[clone] = nanv;
			}
		}
		np.w += massFluid;
		newPos[index] = np;
	}
}

/* densitySumVolumicDevice + densitySumBoundaryDevice with ENABLE_INLET_OUTLET (src/cuda/density_sum_kernel.cu:119-140, 206-250,
 * 374-420, 606-655): orc_sa_density_sum with the open boundaries' terms.  The particles of an open boundary (its vertices, in the
 * volumic sums; its segments, in the gamma sums) count as if they had moved with their Eulerian velocity over the step: they are
 * left out of -sum m W(r^n), and -sum m W(|r^n + dt (u_E - u)|) takes their place (sumVmwDelta); the gamma the old density is
 * weighted with is advanced by the same virtual displacement through the open segments only (compute_imposed_gamma), clipped
 * to [0.1, 1].  A stream that crosses an inlet at the inlet's Eulerian velocity therefore sees neither the inlet's vertices nor
 * the change of its gamma.  GROUNDWORK (see above). */
void orc_sa_density_sum_io(const orc_params *p, orc_f4 *newVel, orc_f4 *newGGam, orc_f4 *forces,
	const orc_f4 *oldPos, const orc_f4 *newPos, const orc_f4 *oldVel, const orc_f4 *oldEulerVel, const orc_f4 *oldGGam,
	const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2, const orc_info *infoArray,
	const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t particleRangeEnd, float dt)
{
	const float kr = 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const float slength = p->slength;
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!FLUID(info)) {
			if (VERTEX(info) || BOUNDARY(info)) newGGam[index] = oldGGam[index];
			continue;
		}
		const orc_f4 posN = oldPos[index], posNp1 = newPos[index];
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
		float sumPmwN = 0.0f, sumPmwNp1 = 0.0f, sumVmwDelta = 0.0f;
		for (int nptype = PT_FLUID; nptype <= PT_VERTEX; nptype += 2) {
			neib_iter it;
			uint32_t neib_index;
			neib_iter_init(&it, p, nptype, index, &posN, gridPos, cellStart, neibsList);
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 nN = oldPos[neib_index];
				if (INACTIVE(nN)) continue;
				const orc_info ninfo = infoArray[neib_index];
				const orc_f4 nNp1 = newPos[neib_index];
				const float rx = it.pos_corr[0] - nN.x, ry = it.pos_corr[1] - nN.y, rz = it.pos_corr[2] - nN.z;
				const float qx = (it.pos_corr[0] - nNp1.x) + dx, qy = (it.pos_corr[1] - nNp1.y) + dy, qz = (it.pos_corr[2] - nNp1.z) + dz;
				if (!IO_BOUNDARY(ninfo)) {
					const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
					sumPmwN -= nN.w*W_c(p->kerneltype, rN, slength, wcoeff, wsub);
				}
				const float rNp1 = sqrtf(qx*qx + qy*qy + qz*qz);
				if (rNp1 < p->influenceradius)
					sumPmwNp1 += nN.w*W_c(p->kerneltype, rNp1, slength, wcoeff, wsub);
				if (IO_BOUNDARY(ninfo)) {      /* densitySumOpenBoundaryContribution */
					const orc_f4 e = oldEulerVel[neib_index], v = oldVel[neib_index];
					const float ex = rx + dt*(e.x - v.x), ey = ry + dt*(e.y - v.y), ez = rz + dt*(e.z - v.z);
					const float newDist = sqrtf(ex*ex + ey*ey + ez*ez);
					if (newDist < p->influenceradius)
						sumVmwDelta -= nN.w*W_c(p->kerneltype, newDist, slength, wcoeff, wsub);
				}
			}
		}
		forces[index].w = sumPmwNp1 + sumPmwN + sumVmwDelta;
		float gGamDotR = 0.0f, sumSgamDelta = 0.0f, sumSgamN = 0.0f;
		v3 gGam = v3_make(0.0f, 0.0f, 0.0f);
		{
			neib_iter it;
			uint32_t neib_index;
			neib_iter_init(&it, p, PT_BOUNDARY, index, &posN, gridPos, cellStart, neibsList);
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 nN = oldPos[neib_index];
				if (INACTIVE(nN)) continue;
				const orc_f4 nNp1 = newPos[neib_index];
				const float inv = 1.0f/slength;
				const v3 qN = v3_make((it.pos_corr[0] - nN.x)*inv, (it.pos_corr[1] - nN.y)*inv, (it.pos_corr[2] - nN.z)*inv);
				const v3 qNp1 = v3_make(((it.pos_corr[0] - nNp1.x) + dx)*inv, ((it.pos_corr[1] - nNp1.y) + dy)*inv,
					((it.pos_corr[2] - nNp1.z) + dz)*inv);
				const orc_f4 be = boundelem[neib_index];
				const v3 ns = v3_make(be.x, be.y, be.z);
				v3 q_vb[3];
				calc_vertex_rel_pos(q_vb, ns, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
					vertPos2 + 2*(size_t)neib_index, slength);
				const v3 gN = v3_scale(ns, grad_gamma_wendland(slength, qN, q_vb, ns));
				const v3 gNp1 = v3_scale(ns, grad_gamma_wendland(slength, qNp1, q_vb, ns));
				gGamDotR += 0.5f*v3_dot(v3_add(gN, gNp1), v3_sub(qNp1, qN));
				gGam = v3_add(gGam, gNp1);
				if (IO_BOUNDARY(infoArray[neib_index])) {      /* io_gamma_contrib */
					const orc_f4 e = oldEulerVel[neib_index], v = oldVel[neib_index];
					const v3 deltaR = v3_make(dt*(e.x - v.x), dt*(e.y - v.y), dt*(e.z - v.z));
					const v3 qDelta = v3_add(qN, v3_divs(deltaR, slength));
					const v3 gDelta = v3_scale(ns, grad_gamma_wendland(slength, qDelta, q_vb, ns));
					sumSgamDelta += v3_dot(deltaR, gDelta);
					sumSgamN += v3_dot(deltaR, gN);
				}
			}
			gGamDotR *= slength;
		}
		const orc_f4 gGamN = oldGGam[index];
		orc_f4 g = { gGam.x, gGam.y, gGam.z, gGamN.w + gGamDotR };
		float imposedGam = gGamN.w + (sumSgamDelta + sumSgamN)/2.0f;      /* compute_imposed_gamma */
		if (imposedGam > 1.0f) imposedGam = 1.0f;
		else if (imposedGam < 0.1f) imposedGam = 0.1f;
		const int fl = FLUID_NUM(info);
		const float rho = (imposedGam*physical_density(p, oldVel[index].w, fl) + forces[index].w)/g.w;
		if (g.w > 1.0f || sqrtf(g.x*g.x + g.y*g.y + g.z*g.z)*slength < 1e-10f)
			g.w = 1.0f;
		else if (g.w < 0.1f)
			g.w = 0.1f;
		newVel[index].w = rho/p->rho0[fl] - 1.0f;
		newGGam[index] = g;
	}
}

/* Water depth at the pressure-driven open boundaries: the one thing forcesDevice<PT_VERTEX, PT_FLUID> leaves behind when
 * ENABLE_INLET_OUTLET | ENABLE_WATER_DEPTH are set and the model is not k-epsilon (vertex_forces, src/cuda/forces.cu:676-686;
 * needs_waterdepth, forces_kernel.def:192-205; skip_neiblist :1375-1389: only the vertices of PRESSURE-driven open boundaries
 * walk their lists; compute_water_depth_at_outflow :3285-3303).  Of every fluid neighbour in range that is not above the vertex,
 * the height above the bottom of the domain is scaled to [0, UINT_MAX] and the maximum per open boundary (the object number of
 * the vertex) is kept: IOwaterdepth[numOpenBoundaries] is an atomicMax target, so it is updated and never cleared here (the
 * problem's imposeBoundaryConditionHost clears it, problems/CompleteSaExample.cu:323-325).  GROUNDWORK (see above). */
void orc_sa_io_water_depth(const orc_params *p, uint32_t *IOwaterdepth, const orc_f4 *posArray, const orc_info *infoArray,
	const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList, uint32_t fromParticle, uint32_t toParticle)
{
	/* serial: the reduction is a max, its result does not depend on the order, and the lists of the vertices are short */
	for (uint32_t index = fromParticle; index < toParticle; ++index) {
		const orc_info info = infoArray[index];
		if (!VERTEX(info)) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		if (!(IO_BOUNDARY(info) && !VEL_IO(info))) continue;        /* skip_neiblist */
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const float r = sqrtf(rx*rx + ry*ry + rz*rz);
			if (INACTIVE(npos)) continue;
			if (r >= p->influenceradius) continue;          /* forcesDevice's range test for a non-boundary neighbour */
			if (rz < 0.0f) continue;                        /* the fluid particle is higher than the vertex */
			float nZpos = pos.z - rz + gridPos[2]*p->cellSize[2] + 0.5f*p->cellSize[2];
			nZpos *= ((float)UINT_MAX)/(p->gridSize[2]*p->cellSize[2]);
			const uint32_t u = (uint32_t)nZpos;
			if (u > IOwaterdepth[OBJECT_NUM(info)]) IOwaterdepth[OBJECT_NUM(info)] = u;
		}
	}
}

/* what the problem's boundary-condition kernel makes of it (problems/CompleteSaExample.cu:266-272): an absolute z */
float orc_sa_io_water_depth_z(const orc_params *p, uint32_t waterdepth)
{
	float z = ((float)waterdepth)/((float)UINT_MAX);
	z *= p->cellSize[2]*p->gridSize[2];
	z += p->worldOrigin[2];
	return z;
}

/* computeDensityDiffusionDevice with ENABLE_INLET_OUTLET (forces_kernel.def:4536-4582): orc_sa_density_diffusion plus the
 * boundary term of the Brezzi diffusion on the segments of PRESSURE-driven open boundaries (:1836-1852): the fluid term with
 * V_b grad W replaced by |grad gamma_as| / r_as, r_as = max(|n_s . r_as|, deltap) (sa_boundary_neib_data :1133-1149), WITHOUT
 * the diffusion coefficient, evaluated in double (the literals 2.0 of :1848 promote the whole product) and subtracted.
 * Segments are in range up to influenceradius + deltap (:4476-4478).  GROUNDWORK (see above). */
void orc_sa_density_diffusion_io(const orc_params *p, orc_f4 *forces, const orc_f4 *posArray, const orc_f4 *velArray,
	const orc_f4 *gGamArray, const orc_info *infoArray, const uint32_t *hashArray, const uint32_t *cellStart,
	const uint16_t *neibsList, const orc_f4 *boundelem, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	uint32_t particleRangeEnd, float dt, float deltap)
{
	const float fcoeff = orc_fcoeff(p->kerneltype, p->slength, 2.0f);
#pragma omp parallel for schedule(dynamic, 256)
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!FLUID(info)) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		const orc_f4 vel = velArray[index];
		const int fl = FLUID_NUM(info);
		const float rho = physical_density(p, vel.w, fl);
		const float pres = orc_P(p, vel.w, fl);
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		float DrDt = 0.0f;
		neib_iter it;
		uint32_t neib_index;
		neib_iter_init(&it, p, PT_FLUID, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const float r = sqrtf(rx*rx + ry*ry + rz*rz);
			if (INACTIVE(npos)) continue;
			if (r >= p->influenceradius) continue;
			const orc_f4 nvel = velArray[neib_index];
			const int nfl = FLUID_NUM(infoArray[neib_index]);
			const float neib_rho = physical_density(p, nvel.w, nfl);
			const float f = F_c(p->kerneltype, r, p->slength, fcoeff);
			const float gdotr = p->gravity[0]*rx + p->gravity[1]*ry + p->gravity[2]*rz;
			float nDrDt = 0.0f;
			nDrDt += p->densityDiffCoeff*((2.0f/(rho + neib_rho))*(pres - orc_P(p, nvel.w, nfl)) - gdotr)*npos.w/neib_rho*f*dt*2.0f*rho;
			DrDt += nDrDt;
		}
		neib_iter_init(&it, p, PT_BOUNDARY, index, &pos, gridPos, cellStart, neibsList);
		while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
			const orc_f4 npos = posArray[neib_index];
			const float rx = it.pos_corr[0] - npos.x, ry = it.pos_corr[1] - npos.y, rz = it.pos_corr[2] - npos.z;
			const float r = sqrtf(rx*rx + ry*ry + rz*rz);
			if (INACTIVE(npos)) continue;
			if (r >= p->influenceradius + deltap) continue;
			const orc_info neib_info = infoArray[neib_index];
			/* the element, its r_as and |grad gamma_as| are made for every segment in range (the output struct's constructor) */
			const orc_f4 belem = boundelem[neib_index];
			const float r_as = fmaxf(fabsf(dot3(rx, ry, rz, belem.x, belem.y, belem.z)), deltap);
			const float inv_h = 1.0f/p->slength;
			const float ggamAS = orc_grad_gamma_vp(p->slength, rx*inv_h, ry*inv_h, rz*inv_h, &belem,
				vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index, vertPos2 + 2*(size_t)neib_index);
			float nDrDt = 0.0f;
			if (IO_BOUNDARY(neib_info) && !VEL_IO(neib_info)) {
				const float nrt = velArray[neib_index].w;
				const int nfl = FLUID_NUM(neib_info);
				const float neib_rho = physical_density(p, nrt, nfl);
				const float gdotr = p->gravity[0]*rx + p->gravity[1]*ry + p->gravity[2]*rz;
				const double t = ((2.0/(rho + neib_rho))*(pres - orc_P(p, nrt, nfl)) - gdotr)*ggamAS/r_as*dt*2.0f*rho;
				nDrDt = (float)(nDrDt - t);
			}
			DrDt += nDrDt;
		}
		DrDt /= gGamArray[index].w;
		forces[index].w = DrDt/p->rho0[fl];
	}
}

/* fluxComputationDevice (src/cuda/post_process_kernel.cu:822-840; FLUX_COMPUTATION of the post-processing engine,
 * src/cuda/post_process.cu:485-570): per open boundary (the object number of its segments) the volume flux sum A_s (u_E . n_s)
 * over its segments -- positive into the domain, the normals point at the fluid.  The reference adds onto a device array it has
 * just allocated and never cleared; here the sums start from zero.  Serial, in particle order (the device adds atomically, in any
 * order: compare to rounding).  GROUNDWORK (see above). */
void orc_flux_computation(float *IOflux, const orc_info *infoArray, const orc_f4 *eulerVel, const orc_f4 *boundelement,
	uint32_t numParticles, uint32_t numOpenBoundaries)
{
	for (uint32_t ob = 0; ob < numOpenBoundaries; ++ob) IOflux[ob] = 0.0f;
	for (uint32_t index = 0; index < numParticles; ++index) {
		const orc_info info = infoArray[index];
		if (!(IO_BOUNDARY(info) && BOUNDARY(info))) continue;
		const orc_f4 normal = boundelement[index], e = eulerVel[index];
		IOflux[OBJECT_NUM(info)] += normal.w*(e.x*normal.x + e.y*normal.y + e.z*normal.z);
	}
}

/* ---- SA_BOUNDARY: the fluid's force on the boundary elements of bodies that feel it ----------------------------------------
 * compute_boundary_pressure_force (src/cuda/forces_kernel.def:3258-3266): pout.force = -P(vel.w, fluid_num) * belem.w * belem,
 * force.w = 0, for BOUNDARY && COMPUTE_FORCE rows of finalizeforcesDevice (:4115-4120); then, as for every COMPUTE_FORCE row that
 * is not a vertex (:4121-4142; no multiplication by the mass with SA_BOUNDARY), rbforces[rbindex] = force, rbtorques[rbindex] =
 * arm x force with arm = globalDistance(particle, centre of gravity); params.forces[index] = force (:4144) */
void orc_sa_body_pressure_forces(const orc_params *p, orc_f4 *forces, orc_f4 *rbforces, orc_f4 *rbtorques,
	const orc_f4 *posArray, const orc_f4 *velArray, const orc_info *infoArray, const uint32_t *hashArray, const orc_f4 *boundElements,
	uint32_t fromParticle, uint32_t toParticle)
{
	for (uint32_t index = fromParticle; index < toParticle; ++index) {
		const orc_info info = infoArray[index];
		if (!(BOUNDARY(info) && COMPUTE_FORCE(info))) continue;
		const orc_f4 pos = posArray[index];
		if (INACTIVE(pos)) continue;
		const orc_f4 belem = boundElements[index];
		const float s = -orc_P(p, velArray[index].w, FLUID_NUM(info))*belem.w;
		orc_f4 force;
		force.x = s*belem.x; force.y = s*belem.y; force.z = s*belem.z; force.w = 0.0f;
		const int obj = OBJECT_NUM(info);
		const uint32_t rbindex = (uint32_t)((int)orc_info_id(info) + p->rbstartindex[obj]);
		rbforces[rbindex] = force;
		int gp[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gp);
		const float ax = (gp[0] - p->rbcgGridPos[obj][0])*p->cellSize[0] + (pos.x - p->rbcgPos[obj][0]);
		const float ay = (gp[1] - p->rbcgGridPos[obj][1])*p->cellSize[1] + (pos.y - p->rbcgPos[obj][1]);
		const float az = (gp[2] - p->rbcgGridPos[obj][2])*p->cellSize[2] + (pos.z - p->rbcgPos[obj][2]);
		orc_f4 tq;
		tq.x = ay*force.z - az*force.y; tq.y = az*force.x - ax*force.z; tq.z = ax*force.y - ay*force.x; tq.w = 0.0f;
		rbtorques[rbindex] = tq;
		forces[index] = force;
	}
}

/* ---- ENABLE_INLET_OUTLET | ENABLE_DENSITY_SUM | ENABLE_MOVING_BODIES (CompleteSaExample.cu:46) -----------------------------------
 * density_sum_impl (src/cuda/euler.cu:112-160) with both flags: densitySumVolumicDevice as with open boundaries alone,
 * computeDensitySumBoundaryTerms (:422-484) with the corners set up for the old and then for the new normal, io_gamma_contrib
 * called AFTER the second set-up with (qN, nsN, vertexRelPos, dt, gGamN) -- i.e. the corners of the new normal with the old
 * normal, which the reference marks "TODO check if we need the old or the new normal here" (:470-476); restated as written.
 * integrateGammaDevice<PT_VERTEX> for the vertex rows (its gamma_sum_terms carry the io sums too but only gGam and gGamDotR are
 * used, :669-685); BOUNDARY rows are left as they are. */
void orc_sa_density_sum_io_moving(const orc_params *p, orc_f4 *newVel, orc_f4 *newGGam, orc_f4 *forces,
	const orc_f4 *oldPos, const orc_f4 *newPos, const orc_f4 *oldVel, const orc_f4 *oldEulerVel, const orc_f4 *oldGGam,
	const orc_f4 *boundelemOld, const orc_f4 *boundelemNew, const float *vertPos0, const float *vertPos1, const float *vertPos2,
	const orc_info *infoArray, const uint32_t *hashArray, const uint32_t *cellStart, const uint16_t *neibsList,
	uint32_t particleRangeEnd, float dt)
{
	const float kr = 2.0f;
	const float wcoeff = orc_wcoeff(p->kerneltype, p->slength, kr), wsub = expf(-kr*kr);
	const float slength = p->slength;
	for (uint32_t index = 0; index < particleRangeEnd; ++index) {
		const orc_info info = infoArray[index];
		if (!FLUID(info) && !VERTEX(info)) continue;
		const orc_f4 posN = oldPos[index], posNp1 = newPos[index];
		int gridPos[3];
		orc_grid_pos_from_hash(p, hashArray[index] & CELLTYPE_BITMASK, gridPos);
		const float dx = posNp1.x - posN.x, dy = posNp1.y - posN.y, dz = posNp1.z - posN.z;
		float fw = 0.0f;
		if (FLUID(info)) {
			float sumPmwN = 0.0f, sumPmwNp1 = 0.0f, sumVmwDelta = 0.0f;
			for (int nptype = PT_FLUID; nptype <= PT_VERTEX; nptype += 2) {
				neib_iter it;
				uint32_t neib_index;
				neib_iter_init(&it, p, nptype, index, &posN, gridPos, cellStart, neibsList);
				while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
					const orc_f4 nN = oldPos[neib_index];
					if (INACTIVE(nN)) continue;
					const orc_info ninfo = infoArray[neib_index];
					const orc_f4 nNp1 = newPos[neib_index];
					const float rx = it.pos_corr[0] - nN.x, ry = it.pos_corr[1] - nN.y, rz = it.pos_corr[2] - nN.z;
					const float qx = (it.pos_corr[0] - nNp1.x) + dx, qy = (it.pos_corr[1] - nNp1.y) + dy, qz = (it.pos_corr[2] - nNp1.z) + dz;
					if (!IO_BOUNDARY(ninfo)) {
						const float rN = sqrtf(rx*rx + ry*ry + rz*rz);
						sumPmwN -= nN.w*W_c(p->kerneltype, rN, slength, wcoeff, wsub);
					}
					const float rNp1 = sqrtf(qx*qx + qy*qy + qz*qz);
					if (rNp1 < p->influenceradius)
						sumPmwNp1 += nN.w*W_c(p->kerneltype, rNp1, slength, wcoeff, wsub);
					if (IO_BOUNDARY(ninfo)) {
						const orc_f4 e = oldEulerVel[neib_index], v = oldVel[neib_index];
						const float ex = rx + dt*(e.x - v.x), ey = ry + dt*(e.y - v.y), ez = rz + dt*(e.z - v.z);
						const float newDist = sqrtf(ex*ex + ey*ey + ez*ez);
						if (newDist < p->influenceradius)
							sumVmwDelta -= nN.w*W_c(p->kerneltype, newDist, slength, wcoeff, wsub);
					}
				}
			}
			fw = sumPmwNp1 + sumPmwN + sumVmwDelta;
			forces[index].w = fw;
		}
		float gGamDotR = 0.0f, sumSgamDelta = 0.0f, sumSgamN = 0.0f;
		v3 gGam = v3_make(0.0f, 0.0f, 0.0f);
		{
			neib_iter it;
			uint32_t neib_index;
			neib_iter_init(&it, p, PT_BOUNDARY, index, &posN, gridPos, cellStart, neibsList);
			while ((neib_index = neib_iter_next(&it)) != UINT_MAX) {
				const orc_f4 nN = oldPos[neib_index];
				if (INACTIVE(nN)) continue;
				const orc_f4 nNp1 = newPos[neib_index];
				const float inv = 1.0f/slength;
				const v3 qN = v3_make((it.pos_corr[0] - nN.x)*inv, (it.pos_corr[1] - nN.y)*inv, (it.pos_corr[2] - nN.z)*inv);
				const v3 qNp1 = v3_make(((it.pos_corr[0] - nNp1.x) + dx)*inv, ((it.pos_corr[1] - nNp1.y) + dy)*inv,
					((it.pos_corr[2] - nNp1.z) + dz)*inv);
				const orc_f4 beN = boundelemOld[neib_index], beNp1 = boundelemNew[neib_index];
				const v3 nsN = v3_make(beN.x, beN.y, beN.z), nsNp1 = v3_make(beNp1.x, beNp1.y, beNp1.z);
				v3 q_vb[3];
				calc_vertex_rel_pos(q_vb, nsN, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
					vertPos2 + 2*(size_t)neib_index, slength);
				const v3 gN = v3_scale(nsN, grad_gamma_wendland(slength, qN, q_vb, nsN));
				calc_vertex_rel_pos(q_vb, nsNp1, vertPos0 + 2*(size_t)neib_index, vertPos1 + 2*(size_t)neib_index,
					vertPos2 + 2*(size_t)neib_index, slength);
				const v3 gNp1 = v3_scale(nsNp1, grad_gamma_wendland(slength, qNp1, q_vb, nsNp1));
				gGamDotR += 0.5f*v3_dot(v3_add(gN, gNp1), v3_sub(qNp1, qN));
				gGam = v3_add(gGam, gNp1);
				if (IO_BOUNDARY(infoArray[neib_index])) {      /* io_gamma_contrib(sumGam, .., qN, nsN, vertexRelPos [of nsNp1], dt, gGamN) */
					const orc_f4 e = oldEulerVel[neib_index], v = oldVel[neib_index];
					const v3 deltaR = v3_make(dt*(e.x - v.x), dt*(e.y - v.y), dt*(e.z - v.z));
					const v3 qDelta = v3_add(qN, v3_divs(deltaR, slength));
					const v3 gDelta = v3_scale(nsN, grad_gamma_wendland(slength, qDelta, q_vb, nsN));
					sumSgamDelta += v3_dot(deltaR, gDelta);
					sumSgamN += v3_dot(deltaR, gN);
				}
			}
			gGamDotR *= slength;
		}
		const orc_f4 gGamN = oldGGam[index];
		orc_f4 g = { gGam.x, gGam.y, gGam.z, gGamN.w + gGamDotR };
		if (VERTEX(info)) { newGGam[index] = g; continue; }
		float imposedGam = gGamN.w + (sumSgamDelta + sumSgamN)/2.0f;
		if (imposedGam > 1.0f) imposedGam = 1.0f;
		else if (imposedGam < 0.1f) imposedGam = 0.1f;
		const int fl = FLUID_NUM(info);
		const float rho = (imposedGam*physical_density(p, oldVel[index].w, fl) + fw)/g.w;
		if (g.w > 1.0f || sqrtf(g.x*g.x + g.y*g.y + g.z*g.z)*slength < 1e-10f)
			g.w = 1.0f;
		else if (g.w < 0.1f)
			g.w = 0.1f;
		newVel[index].w = rho/p->rho0[fl] - 1.0f;
		newGGam[index] = g;
	}
}

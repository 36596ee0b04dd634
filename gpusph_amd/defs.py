"""Numeric codes of GPUSPH's enums, flags and data-model constants.

References (relative to the GPUSPH tree): src/particledefine.h:67-224, src/visc_spec.h,
src/simflags.h:62-160, src/particleinfo.h:144-300, src/multi_gpu_defines.h:56-83,
src/common_types.h:57-72, src/hashkey.h:44-47.
"""
# KernelType
CUBICSPLINE, QUADRATIC, WENDLAND, GAUSSIAN = 1, 2, 3, 4
# SPHFormulation
SPH_F1, SPH_F2, SPH_GRENIER, SPH_HA = 1, 2, 3, 4
# DensityDiffusionType
DENSITY_DIFFUSION_NONE, FERRARI, COLAGROSSI, BREZZI = 0, 1, 2, 3
# BoundaryType
LJ_BOUNDARY, MK_BOUNDARY, SA_BOUNDARY, DYN_BOUNDARY = 0, 1, 2, 3
# RheologyType / TurbulenceModel / ComputationalViscosityType / ViscousModel / AverageOperator
INVISCID, NEWTONIAN, GRANULAR, BINGHAM, PAPANASTASIOU, POWER_LAW, HERSCHEL_BULKLEY, ALEXANDROU, DEKEE_TURCOTTE, ZHU = range(10)
LAMINAR_FLOW, ARTIFICIAL, SPS, KEPSILON = 0, 1, 2, 3
KINEMATIC, DYNAMIC = 0, 1
MORRIS, MONAGHAN, ESPANOL_REVENGA = 0, 1, 2
ARITHMETIC, HARMONIC, GEOMETRIC = 0, 1, 2
# Periodicity
PERIODIC_NONE, PERIODIC_X, PERIODIC_Y, PERIODIC_Z = 0, 1, 2, 4
# RunMode
REPACK, SIMULATE = 0, 1
# simflags
ENABLE_NONE = 0
ENABLE_DTADAPT = 1 << 0
ENABLE_XSPH = 1 << 1
ENABLE_PLANES = 1 << 2
ENABLE_DEM = 1 << 3
ENABLE_MOVING_BODIES = 1 << 4
ENABLE_INLET_OUTLET = 1 << 5
ENABLE_WATER_DEPTH = 1 << 6
ENABLE_DENSITY_SUM = 1 << 7
ENABLE_GAMMA_QUADRATURE = 1 << 8
ENABLE_REPACKING = 1 << 9
ENABLE_INTERNAL_ENERGY = 1 << 10
ENABLE_MULTIFLUID = 1 << 11
# ParticleType / flags
PT_FLUID, PT_BOUNDARY, PT_VERTEX, PT_TESTPOINT, PT_NONE = 0, 1, 2, 3, 4
PART_FLAG_SHIFT = 3
FG_COMPUTE_FORCE = 1 << 3
FG_MOVING_BOUNDARY = 1 << 4
# open boundaries of SA_BOUNDARY (src/particleinfo.h:153-156); their passes: gpusph_amd/csrc/sa_io.hip + the open-boundary terms of
# sa_bounds.hip, driven by MultiGpuEngine._sa_post_euler_io
FG_INLET = 1 << 5
FG_OUTLET = 1 << 6
FG_VELOCITY_DRIVEN = 1 << 7
FG_CORNER = 1 << 8
FG_SURFACE = 1 << 9
FG_INTERFACE = 1 << 10
# cell types / hash
CELLTYPE_INNER_CELL, CELLTYPE_INNER_EDGE_CELL, CELLTYPE_OUTER_EDGE_CELL, CELLTYPE_OUTER_CELL = 0, 1, 2, 3
CELLTYPE_BITMASK = 0x3FFFFFFF
CELL_HASH_MAX = 0xFFFFFFFF
EMPTY_SEGMENT = 0xFFFFFFFF
EMPTY_CELL = 0xFFFFFFFF
MAX_CELLS = 0xFFFFFFFF >> 2
# neighbour data
CELLNUM_SHIFT = 11
CELLNUM_ENCODED = 1 << 11
NEIBINDEX_MASK = CELLNUM_ENCODED - 1
NEIBS_END = 0xFFFF
MAX_FLUID_TYPES = 4
MAX_BODIES = 16
BLOCK_SIZE_FORCES = 128
# linearisations (src/linearization.h; Makefile:517-519): name -> axis index of COORD1,2,3
LINEARIZATIONS = {
    "xyz": (0, 1, 2), "xzy": (0, 2, 1), "yxz": (1, 0, 2),
    "yzx": (1, 2, 0), "zxy": (2, 0, 1), "zyx": (2, 1, 0),
}
DEFAULT_LINEARIZATION = "yzx"

# FilterType (src/particledefine.h:255-260)
SHEPARD_FILTER, MLS_FILTER = 0, 1

# PostProcessType (src/particledefine.h:290-299)
VORTICITY, TESTPOINTS, SURFACE_DETECTION, INTERFACE_DETECTION, FLUX_COMPUTATION, CALC_PRIVATE = 0, 1, 2, 3, 4, 5

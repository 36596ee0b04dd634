"""N > 1 host logic on CPU: world_size-2 and -3 gloo runs of gpusph_amd.multigpu with the test-only
oracle kernel backend; the slab-decomposed trajectory must equal the single-domain one BIT FOR BIT
(per-particle neighbour order and arithmetic do not depend on the decomposition)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, steps, case, outdir, filters=()):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    import torch.distributed as dist
    from gpusph_amd.problem import DamBreak3D
    from gpusph_amd.multigpu import MultiGpuEngine
    from oracle_kernels import OracleKernels
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    case = dict(case)
    gate = case.pop("moving_gate", False)
    which = case.pop("problem", "DamBreak3D")
    if which == "StillWater":
        from gpusph_amd.problem import StillWater
        prob = StillWater(**case)
    elif which == "WaveTank":
        from gpusph_amd.problem import WaveTank
        prob = WaveTank(**case)
    elif which == "Poiseuille":
        from gpusph_amd.problem import Poiseuille
        prob = Poiseuille(**case)
    elif which == "SABox":
        from gpusph_amd.problem import SABox
        prob = SABox(**case)
    elif which == "SAPaddleBox":
        from gpusph_amd.problem import SAPaddleBox
        prob = SAPaddleBox(**case)
    elif which == "OpenChannel":
        from gpusph_amd.problem import OpenChannel
        prob = OpenChannel(**case)
    elif which == "SAChannelIO":
        from gpusph_amd.problem import SAChannelIO
        prob = SAChannelIO(**case)
    elif which == "SAChannelIOFlap":
        from gpusph_amd.problem import SAChannelIOFlap
        prob = SAChannelIOFlap(**case)
    elif which == "PeriodicBox":
        from gpusph_amd.problem import PeriodicBox
        prob = PeriodicBox(**case)
    else:
        prob = DamBreak3D(**case)
    if gate:
        from test_oracle_physics import _gate_callback
        prob.moving_bodies_callback = _gate_callback(2.0, 60.0, 2.0)
    eng = MultiGpuEngine(prob, "cpu", rank, world, kernels=None if False else _mk(prob, rank, world))
    for ftype, freq in filters:
        eng.add_filter(ftype, freq)
    for _ in range(steps):
        eng.step()
    if getattr(eng, "io", False):
        eng.build_neibs()        # the particles released in the last step are sorted in (and the removed ones dropped) by a rebuild
    out = eng.download_internal()
    if getattr(eng, "io", False):
        out["eulervel"] = eng.eulervel[:eng.n_int].numpy()
        out["io_created"] = np.array(eng.io_created)
        out["n0"] = np.array(prob.num_particles)
    rb = eng.reduce_rb_forces() or (np.zeros((0, 3)), np.zeros((0, 3)))     # no bodies with force feedback: nothing to reduce
    np.savez(os.path.join(outdir, "r%d_of_%d.npz" % (rank, world)), n_local=eng.n_local, dt=eng.current_dt(),
             interactions=eng.neibs_info().numInteractions, rbf=rb[0], rbt=rb[1], **out)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def _mk(prob, rank, world):
    # allocation must match what MultiGpuEngine computes: build the engine's alloc the same way
    from gpusph_amd.multigpu import SlabPartition
    from oracle_kernels import OracleKernels
    arrs = prob.copy_to_array()
    part = SlabPartition(prob, world)
    n0 = int(part.local_mask(rank, arrs["hash"]).sum()) if world > 1 else len(arrs["hash"])
    return OracleKernels(prob, int(n0 * 1.25) + 4096)


def _run(world, steps, case, outdir, filters=()):
    port = _free_port()
    if world == 1:
        _worker(0, 1, port, steps, case, outdir, filters)
    else:
        mp.spawn(_worker, args=(world, port, steps, case, outdir, filters), nprocs=world, join=True)


def _gather(outdir, world):
    parts = [np.load(os.path.join(outdir, "r%d_of_%d.npz" % (r, world))) for r in range(world)]
    ids = np.concatenate([p["info"][:, 2].astype(np.uint32) | (p["info"][:, 3].astype(np.uint32) << 16) for p in parts])
    order = np.argsort(ids)
    keys = [k for k in ("pos", "vel", "info", "hash", "forces", "vol", "energy", "gradgamma", "boundelements",
                        "tke", "eps", "turbvisc", "eulervel") if k in parts[0]]
    cat = {k: np.concatenate([p[k] for p in parts])[order] for k in keys}
    return ids[order], cat, parts


@pytest.mark.parametrize("world,lin", [(2, "xzy"), (3, "yzx")])
def test_slab_runs_equal_single_domain(tmp_path, world, lin):
    case = dict(deltap=0.04, obstacle=True, linearization=lin, jitter=0.05)
    steps = 13                                   # spans two neighbour-list rebuilds (iterations 0 and 10)
    _run(1, steps, case, str(tmp_path))
    _run(world, steps, case, str(tmp_path))
    ids1, one, p1 = _gather(str(tmp_path), 1)
    idsN, many, pN = _gather(str(tmp_path), world)
    assert np.array_equal(ids1, idsN)            # every particle is owned by exactly one rank
    for k in ("pos", "vel", "forces"):
        assert np.array_equal(one[k].view(np.uint32), many[k].view(np.uint32)), k
    assert np.array_equal(one["hash"] & 0x3FFFFFFF, many["hash"] & 0x3FFFFFFF)
    assert all(float(p["dt"]) == float(p1[0]["dt"]) for p in pN)
    assert sum(int(p["interactions"]) for p in pN) == int(p1[0]["interactions"])
    assert all(int(p["n_local"]) > len(p["pos"]) for p in pN)      # every rank holds a halo
    # total force / torque on the feedback body: every rank gets the all-reduced sum, equal to the single-domain one
    scale = max(np.abs(p1[0]["rbf"]).max(), 1e-12)
    assert scale > 0
    for p in pN:
        assert np.abs(p["rbf"] - p1[0]["rbf"]).max() <= 1e-5 * scale
        assert np.abs(p["rbt"] - p1[0]["rbt"]).max() <= 1e-5 * max(np.abs(p1[0]["rbt"]).max(), 1e-12)


def test_slab_run_with_sps_and_shepard_filter_equals_single_domain(tmp_path):
    """WaveTank's option set over two slabs: viscosity<SPSVISC> (stress tensor computed for the internal particles and
    imported for the halo before each forces pass) and a Shepard filter every 4 iterations (filtered velocities imported
    for the halo): bit-equal to the single-domain run"""
    case = dict(deltap=0.04, obstacle=True, linearization="xzy", jitter=0.05, viscosity="SPSVISC", kinematic_visc=1.0e-6)
    filters = ((0, 4),)
    steps = 12
    _run(1, steps, case, str(tmp_path), filters)
    _run(2, steps, case, str(tmp_path), filters)
    ids1, one, p1 = _gather(str(tmp_path), 1)
    ids2, two, p2 = _gather(str(tmp_path), 2)
    assert np.array_equal(ids1, ids2)
    for k in ("pos", "vel", "forces"):
        assert np.array_equal(one[k].view(np.uint32), two[k].view(np.uint32)), k
    assert all(float(p["dt"]) == float(p1[0]["dt"]) for p in p2)


def test_slab_run_with_a_moving_body_equals_single_domain(tmp_path):
    """a body with prescribed motion (translation + rotation) that straddles the slab boundary: every rank runs the same
    host kinematics; bit-equal to the single-domain run, and to the reference-order oracle driver"""
    case = dict(deltap=0.04, obstacle=True, linearization="xzy", jitter=0.05, moving_gate=True)
    steps = 12
    _run(1, steps, case, str(tmp_path))
    _run(2, steps, case, str(tmp_path))
    ids1, one, p1 = _gather(str(tmp_path), 1)
    ids2, two, p2 = _gather(str(tmp_path), 2)
    assert np.array_equal(ids1, ids2)
    for k in ("pos", "vel", "forces"):
        assert np.array_equal(one[k].view(np.uint32), two[k].view(np.uint32)), k
    body = (one["info"][:, 0] & 0x10) != 0
    assert body.sum() > 50 and np.abs(one["vel"][body, :3]).max() > 0.5       # the body does move
    assert len({int(p["info"][(p["info"][:, 0] & 0x10) != 0].shape[0] > 0) for p in p2}) >= 1


def test_partition_and_device_map():
    sys.path[:0] = [ROOT]
    from gpusph_amd.problem import DamBreak3D
    from gpusph_amd.multigpu import SlabPartition
    from gpusph_amd import defs as D
    prob = DamBreak3D(0.02, linearization="xzy", obstacle=False)
    part = SlabPartition(prob, 4)
    assert part.lo[0] == 0 and part.hi[-1] == part.gs3 and all(part.hi[d] == part.lo[d + 1] for d in range(3))
    for r in range(4):
        t = part.plane_types(r)
        assert (t[part.lo[r]:part.hi[r]] <= D.CELLTYPE_INNER_EDGE_CELL).all()
        assert (t == D.CELLTYPE_OUTER_EDGE_CELL).sum() == (r > 0) + (r < 3)
        dm = part.compact_device_map(r)
        assert len(dm) == prob.grid_cells and set(np.unique(dm >> 30)) <= {0, 1, 2, 3}
    with pytest.raises(ValueError):
        SlabPartition(prob, 64)


@pytest.mark.parametrize("planes,world", [(28, 8), (21, 6), (35, 10), (24, 8), (30, 4), (9, 3), (200, 8)])
def test_partition_never_starves_a_device(planes, world):
    """the reference's rounding rule (round(planes / devices) each, the last device the rest) leaves the last device 0 planes at
    (28, 8) and one at (21, 6) although the average is >= 3: every slab must keep two distinct edge planes, on a chain and on a ring"""
    sys.path[:0] = [ROOT]
    from gpusph_amd.multigpu import SlabPartition
    from gpusph_amd import defs as D

    class P:      # what SlabPartition reads of a problem
        linearization = "xzy"

        class simparams:
            periodicbound = 0

    for ring in (False, True):
        P.m_gridsize = np.array([5, planes, 4])           # xzy: COORD3 = y
        P.simparams.periodicbound = D.PERIODIC_Y if ring else 0
        part = SlabPartition(P, world)
        assert part.ring == ring
        assert part.lo[0] == 0 and part.hi[-1] == planes and all(part.hi[d] == part.lo[d + 1] for d in range(world - 1))
        assert min(h - l for l, h in zip(part.lo, part.hi)) >= 2
        assert max(h - l for l, h in zip(part.lo, part.hi)) - min(h - l for l, h in zip(part.lo, part.hi)) <= max(2, planes // world)
        owned = np.zeros(planes, dtype=int)
        for r in range(world):
            t = part.plane_types(r)
            inner = t <= D.CELLTYPE_INNER_EDGE_CELL
            owned += inner
            assert inner.sum() == part.hi[r] - part.lo[r]
            edges = (r > 0 or ring) + (r < world - 1 or ring)
            assert (t == D.CELLTYPE_INNER_EDGE_CELL).sum() == edges and (t == D.CELLTYPE_OUTER_EDGE_CELL).sum() == edges
        assert (owned == 1).all()      # every plane belongs to exactly one device


def test_slab_run_of_the_stillwater_mirror_equals_single_domain(tmp_path):
    """StillWater's option set over two slabs: viscosity<DYNAMICVISC>, Ferrari density diffusion, DYN walls, MLS filter every
    4 iterations (filtered velocities imported for the halo), 12 steps: bit-equal to the single-domain run"""
    case = dict(problem="StillWater", ppH=10, linearization="xzy", jitter=0.05)      # 7 planes along y: 3.5 per device
    filters = ((1, 4),)      # MLS_FILTER
    steps = 12
    _run(1, steps, case, str(tmp_path), filters)
    _run(2, steps, case, str(tmp_path), filters)
    ids1, one, p1 = _gather(str(tmp_path), 1)
    ids2, two, p2 = _gather(str(tmp_path), 2)
    assert np.array_equal(ids1, ids2) and len(ids1) > 3000
    for k in ("pos", "vel", "forces"):
        assert np.array_equal(one[k].view(np.uint32), two[k].view(np.uint32)), k
    assert all(float(p["dt"]) == float(p1[0]["dt"]) for p in p2)
    assert all(int(p["n_local"]) < len(ids1) for p in p2)      # each rank holds a slab plus its halo, not the whole domain


def test_slab_run_of_the_wavetank_mirror_equals_single_domain(tmp_path):
    """WaveTank's option set (BASELINE configs[4]) over two slabs split along y: LJ box particles + six planes, SPSVISC (stress
    tensor imported for the halo), the hinged paddle straddling the slab boundary and driven by the same host kinematics on
    every rank, Shepard filter every 4 iterations, 10 steps: bit-equal to the single-domain run"""
    case = dict(problem="WaveTank", deltap=0.03, paddle_tstart=0.0, linearization="xzy")      # 0.6 m across: 7 planes along y
    filters = ((0, 4),)      # SHEPARD_FILTER
    steps = 10
    _run(1, steps, case, str(tmp_path), filters)
    _run(2, steps, case, str(tmp_path), filters)
    ids1, one, p1 = _gather(str(tmp_path), 1)
    ids2, two, p2 = _gather(str(tmp_path), 2)
    assert np.array_equal(ids1, ids2)
    for k in ("pos", "vel", "forces"):
        assert np.array_equal(one[k].view(np.uint32), two[k].view(np.uint32)), k
    assert all(float(p["dt"]) == float(p1[0]["dt"]) for p in p2)
    paddle = (one["info"][:, 0] & 0x10) != 0
    assert paddle.sum() > 50 and np.abs(one["vel"][paddle, :3]).max() > 1e-3     # the paddle moves
    assert all(int(p["n_local"]) < len(ids1) for p in p2)


def test_partition_of_a_periodic_split_axis_is_a_ring():
    from gpusph_amd import defs as D
    """a domain periodic along COORD3 makes a ring of the slabs: the first and the last one are neighbours through the face"""
    from gpusph_amd.multigpu import SlabPartition
    from gpusph_amd.problem import PeriodicBox
    assert not SlabPartition(PeriodicBox(0.05, n=(20, 20, 20)), 1).ring
    assert not SlabPartition(PeriodicBox(0.05, n=(20, 20, 20), periodic=D.PERIODIC_Y | D.PERIODIC_Z), 2).ring    # default yzx: COORD3 = x
    ring = SlabPartition(PeriodicBox(0.05, n=(12, 12, 40), linearization="xyz"), 3)     # 15 planes along z, periodic
    assert ring.ring and ring.gs3 == 15
    t0, t2 = ring.plane_types(0), ring.plane_types(2)
    assert t0[0] == D.CELLTYPE_INNER_EDGE_CELL and t0[14] == D.CELLTYPE_OUTER_EDGE_CELL      # the first slab sees the last plane as halo
    assert t2[14] == D.CELLTYPE_INNER_EDGE_CELL and t2[0] == D.CELLTYPE_OUTER_EDGE_CELL      # ... and the last slab the first plane
    two = SlabPartition(PeriodicBox(0.05, n=(12, 12, 40), linearization="xyz"), 2)      # each slab is the other's neighbour on both sides
    t = two.plane_types(0)
    assert [int(t[0]), int(t[two.hi[0] - 1]), int(t[two.hi[0]]), int(t[14])] == [
        D.CELLTYPE_INNER_EDGE_CELL, D.CELLTYPE_INNER_EDGE_CELL, D.CELLTYPE_OUTER_EDGE_CELL, D.CELLTYPE_OUTER_EDGE_CELL]


@pytest.mark.parametrize("world", [2, 3])
def test_ring_of_slabs_on_a_periodic_split_axis_equals_single_domain(tmp_path, world):
    """slabs along a periodic axis: the first and the last one exchange their outermost planes through the periodic face (with
    two slabs each is the other's neighbour on both sides), a stream along the axis carries particles across the cuts and
    across the face; bit-identical to the single domain"""
    case = dict(problem="PeriodicBox", deltap=0.05, n=(12, 12, 40), linearization="xyz", jitter=0.2, velocity=(0.1, 0.0, 1.5))
    steps = 23                                   # three neighbour-list rebuilds
    _run(1, steps, case, str(tmp_path))
    _run(world, steps, case, str(tmp_path))
    ids1, one, p1 = _gather(str(tmp_path), 1)
    idsN, many, pN = _gather(str(tmp_path), world)
    assert np.array_equal(ids1, idsN)            # every particle is owned by exactly one rank
    for k in ("pos", "vel", "forces"):
        assert np.array_equal(one[k].view(np.uint32), many[k].view(np.uint32)), k
    assert np.array_equal(one["hash"] & 0x3FFFFFFF, many["hash"] & 0x3FFFFFFF)
    assert all(float(p["dt"]) == float(p1[0]["dt"]) for p in pN)
    assert sum(int(p["interactions"]) for p in pN) == int(p1[0]["interactions"])
    assert all(int(p["n_local"]) > len(p["pos"]) for p in pN)      # every rank holds a halo on both sides
    # the run did carry particles through the periodic face and across the cuts between the slabs
    from gpusph_amd.problem import PeriodicBox
    from gpusph_amd.multigpu import SlabPartition
    prob = PeriodicBox(**{k: v for k, v in case.items() if k != "problem"})
    part = SlabPartition(prob, world)
    a0 = prob.copy_to_array()
    id0 = a0["info"].view(np.uint16).reshape(-1, 4)
    id0 = id0[:, 2].astype(np.uint32) | (id0[:, 3].astype(np.uint32) << 16)
    plane0 = ((a0["hash"].view(np.uint32) & 0x3FFFFFFF) // part.plane)[np.argsort(id0)]
    plane1 = (one["hash"] & 0x3FFFFFFF) // part.plane
    assert ((plane0 == part.gs3 - 1) & (plane1 == 0)).sum() > 0
    assert ((plane0 == part.hi[0] - 1) & (plane1 == part.lo[1])).sum() > 0


FIDELITY_CASES = {
    # SPH_GRENIER: sigma and the rewritten densities are exchanged after COMPUTE_DENSITY, the volumes travel with the halo
    "grenier": dict(deltap=0.05, obstacle=False, two_fluids=True, formulation=3, viscosity="DYNAMICVISC", density_diffusion=0, jitter=0.1),
    # generalized Newtonian: BUFFER_EFFVISC exchanged after CALC_VISC; each device limits dt by its own largest viscosity
    "papanastasiou": dict(problem="Poiseuille", ppH=10, rheology=4, linearization="xyz"),
    # internal energy: the rate travels with the forces, the energy with the halo
    "energy": dict(deltap=0.05, obstacle=False, internal_energy=True, jitter=0.1),
    # SPH_HA and the MONAGHAN model need no extra buffer: the ranges of the edge / inner stripes go through the same routing
    "ha": dict(deltap=0.05, obstacle=False, two_fluids=True, formulation=4, density_diffusion=2, jitter=0.1,
               viscosity=dict(rheologytype=1, turbmodel=0, compvisc=1, avgop=1)),
    "monaghan": dict(problem="Poiseuille", ppH=10, viscmodel=1, linearization="xyz"),
    # SA_BOUNDARY in both built forms: vertices / boundary elements / gamma travel with the halo, every boundary-condition and
    # density / gamma pass is followed by the import of what it wrote, the vertex offsets of the halo segments after the list build
    "sa-density-sum": dict(problem="SABox", deltap=0.05, options="StillWaterSA", jitter=0.1),
    "sa-quadrature": dict(problem="SABox", deltap=0.05, options="StillWaterRepackSA", jitter=0.1),
    # k-epsilon on SA walls: k, epsilon, eddy viscosity and Eulerian velocity travel with the halo, DKDE with the forces, the
    # boundary conditions export the wall rows they wrote
    "sa-keps": dict(problem="SABox", deltap=0.05, jitter=0.1, viscosity=dict(rheologytype=1, turbmodel=3)),
    # SA bodies with prescribed motion: the flap straddles the cut; BUFFER_BOUNDELEMENTS is a state buffer whose halo copies are
    # turned by the same body motion, the vertex rows' gamma travels with the density summation / gamma integration
    "sa-moving-density-sum": dict(problem="SAPaddleBox", deltap=0.05, options="StillWaterSA", jitter=0.1),
    "sa-moving-quadrature": dict(problem="SAPaddleBox", deltap=0.05, options="StillWaterRepackSA", jitter=0.1),
    # OpenChannel's options: slabs cut ACROSS the periodic stream, i.e. a ring of two devices (each is the other's left and right)
    "open-channel-ring": dict(problem="OpenChannel", deltap=0.05, linearization="yzx"),
}


@pytest.mark.parametrize("name", sorted(FIDELITY_CASES))
def test_slab_runs_of_the_fidelity_option_sets_equal_single_domain(tmp_path, name):
    case = FIDELITY_CASES[name]
    steps = 12                     # crosses a neighbour rebuild, i.e. a new halo import
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    _run(1, steps, case, str(one))
    _run(2, steps, case, str(two))
    ids1, a, p1 = _gather(str(one), 1)
    ids2, b, p2 = _gather(str(two), 2)
    assert np.array_equal(ids1, ids2)
    for k in a:
        if k == "hash":       # the two high bits are the cell type of the slab decomposition
            assert np.array_equal(a[k] & 0x3FFFFFFF, b[k] & 0x3FFFFFFF)
        else:
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (name, k)
    assert all(p["n_local"] > 0 for p in p2) and float(p2[0]["dt"]) == float(p1[0]["dt"])


@pytest.mark.parametrize("cut", ["across the stream", "along the stream", "along the stream, until particles leave",
                                 "across the stream, with a moving flap"])
def test_open_channel_over_two_slabs_equals_single_domain(tmp_path, cut):
    """SAChannelIO (open boundaries: inlet on the first slab, pressure outlet on the last, the water level the outlet's pressure
    follows measured on one device and used on both) cut across the stream: every pass of the open-boundary sequence over the
    internal particles with the UPDATE_EXTERNAL of what it wrote, the water depth as the maximum over the devices, particles
    released behind the halo rows and sorted in at the next rebuild -- bit-equal to the single-domain run."""
    if cut == "across the stream":
        case = dict(problem="SAChannelIO", deltap=0.05, U=0.6, linearization="yzx")
    else:
        # slabs along y: both open boundaries on both devices, their vertices and segments in each other's halo, particles released
        # next to the cut, the outlet's water level measured on both
        case = dict(problem="SAChannelIO", deltap=0.05, U=0.6, l=0.6, w=0.8, linearization="xzy")
    steps = 14
    if cut.endswith("flap"):
        # CompleteSaExample.cu's option set (open boundaries + density summation + moving bodies): the y = w wall turns and slides,
        # straddling the cut; BUFFER_BOUNDELEMENTS is state (re-sorted in every step, its halo copies turned by the same motion),
        # the density summation takes the elements of both states and the open faces' terms
        case = dict(problem="SAChannelIOFlap", deltap=0.05, U=0.6, linearization="yzx")
    if cut.endswith("leave"):
        # a fast stream in a short tank: the first layer crosses the outlet within the run -- the marks of FIND_OUTGOING_SEGMENT
        # travel to the halo, a vertex takes over the mass of a neighbour device's particle, both devices disable their copy
        case = dict(problem="SAChannelIO", deltap=0.05, U=2.0, l=0.5, w=0.8, h=0.3, H=0.2, linearization="xzy")
        steps = 74
    _run(1, steps, case, str(tmp_path))
    _run(2, steps, case, str(tmp_path))
    ids1, one, p1 = _gather(str(tmp_path), 1)
    ids2, two, p2 = _gather(str(tmp_path), 2)
    assert np.array_equal(ids1, ids2)
    if cut.endswith("leave"):
        assert len(ids1) < int(p1[0]["n0"]) + int(p1[0]["io_created"])       # somebody left
    assert int(p1[0]["io_created"]) > 0 and sum(int(p["io_created"]) for p in p2) == int(p1[0]["io_created"])
    for k in ("pos", "vel", "gradgamma", "eulervel") + (("boundelements",) if cut.endswith("flap") else ()):
        a, b = one[k], two[k]
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32)), k
    assert np.array_equal(one["info"], two["info"])
    assert all(float(p["dt"]) == float(p1[0]["dt"]) for p in p2)
    assert all(int(p["n_local"]) > len(p["pos"]) for p in p2)

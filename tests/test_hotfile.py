"""GPUSPH HotFile v1 round trip and the struct layouts the reference's scripts/hotdiff.py expects."""
import struct
import numpy as np
import pytest

from gpusph_amd import hotfile
from gpusph_amd.problem import DamBreak3D


def test_layouts_match_hotdiff():
    # scripts/hotdiff.py: header '@IIIII48xLdf12x', buffer '@I64sII'; sizeof(header_t) = 104, encoded_buffer_t = 76,
    # encoded_body_t = 6 x 4 + 26 x 8 + 10 x 4 = 272 (src/writers/HotFile.h:45-56, HotFile.cc:40-73)
    assert struct.calcsize(hotfile.HEADER) == 104
    assert struct.calcsize(hotfile.BUFFER) == 76
    assert struct.calcsize(hotfile.BODY) == 272


def test_round_trip(tmp_path):
    prob = DamBreak3D(deltap=0.08, obstacle=True)
    arrs = prob.copy_to_array()
    n = len(arrs["hash"])
    body = dict(index=0, id=0, type=hotfile.MB_FORCES_MOVING, numparts=prob.num_obstacle, firstindex=int(prob.rb_firstindex[0]),
                lastindex=prob.num_obstacle - 1, crot=[0.96, 0.335, 0.3], lvel=[0, 0, 0], avel=[0, 0, 0], orientation=[1, 0, 0, 0])
    path = tmp_path / "hot_00010.bin"
    hotfile.write_hotfile(path, arrs, iterations=10, t=0.0123, dt=3.5e-4, bodies=[body])
    hf = hotfile.read_hotfile(path)
    assert (hf["version"], hf["particles"], hf["iterations"], hf["buffer_count"]) == (1, n, 10, 5)
    assert hf["t"] == 0.0123 and hf["dt"] == np.float32(3.5e-4)
    for k in ("pos", "vel", "info", "hash"):
        assert np.array_equal(np.asarray(hf["arrays"][k]).view(np.uint8).ravel(), np.ascontiguousarray(arrs[k]).view(np.uint8).ravel())
    assert len(hf["bodies"]) == 1 and hf["bodies"][0]["numparts"] == prob.num_obstacle
    assert hf["bodies"][0]["crot"] == (0.96, 0.335, 0.3) and hf["bodies"][0]["initial_orientation"] == (1.0, 0.0, 0.0, 0.0)
    # the reference tool walks the file the same way: header, then (76-byte record + element_size*n) per buffer
    raw = open(path, "rb").read()
    o = 104
    names = []
    while o < len(raw) - 272:
        ln, name, elsize, cnt = struct.unpack("@I64sII", raw[o:o + 76])
        names.append(name[:ln].decode()); o += 76 + elsize * n
    assert names == ["Position", "Velocity", "Info", "Hash"] and o == len(raw) - 272


@pytest.mark.gpu
def test_resume_is_bit_identical(tmp_path):
    """save at a neighbour-rebuild iteration, resume in a fresh engine: same particles as the uninterrupted run"""
    from gpusph_amd.engine import TimestepEngine
    prob = DamBreak3D(deltap=0.04, obstacle=True, jitter=0.05, hydrostatic=False)
    a = TimestepEngine(prob)
    a.run(10)
    path = tmp_path / "hot.bin"
    a.save_hotfile(path)
    a.run(7)
    ref = a.download()
    b = TimestepEngine(prob)
    hf = b.load_hotfile(path)
    assert hf["iterations"] == 10 and b.iterations == 10
    b.run(7)
    out = b.download()
    assert b.n == a.n and b.current_dt() == a.current_dt() and abs(b.time() - a.time()) < 1e-12
    for k in ("pos", "vel", "info", "hash"):
        assert np.array_equal(np.asarray(out[k]).view(np.uint8), np.asarray(ref[k]).view(np.uint8)), k


@pytest.mark.gpu
def test_resume_with_a_moving_body_is_bit_identical(tmp_path):
    """WaveTank mirror (hinged paddle with prescribed motion, SPS: 6 host buffers): the body records carry the live
    kinematic data, a resumed run continues the paddle where it was"""
    from gpusph_amd.engine import TimestepEngine
    from gpusph_amd.problem import WaveTank
    from gpusph_amd import hotfile
    prob = WaveTank(0.06, paddle_tstart=0.0)
    a = TimestepEngine(prob)
    a.run(10)
    path = tmp_path / "hot.bin"
    a.save_hotfile(path)
    hf0 = hotfile.read_hotfile(path)
    assert hf0["buffer_count"] == 6 and len(hf0["bodies"]) == 1           # + BUFFER_SPS_TURBVISC; simparams.numbodies
    assert hf0["bodies"][0]["type"] == hotfile.MB_MOVING and np.abs(hf0["bodies"][0]["avel"]).max() > 0
    a.run(7)
    ref = a.download()
    b = TimestepEngine(WaveTank(0.06, paddle_tstart=0.0))
    b.load_hotfile(path)
    b.run(7)
    out = b.download()
    assert b.n == a.n and b.current_dt() == a.current_dt()
    for k in ("pos", "vel", "info", "hash"):
        assert np.array_equal(np.asarray(out[k]).view(np.uint8), np.asarray(ref[k]).view(np.uint8)), k


@pytest.mark.skipif(not __import__("os").path.exists("/root/reference/scripts/hotdiff.py"),
                    reason="the reference tree is only present in the build container")
def test_reference_hotdiff_reads_our_files(tmp_path):
    """the reference's own comparison tool walks two of our HotFiles end to end and finds no difference"""
    import subprocess, sys
    prob = DamBreak3D(deltap=0.08, obstacle=False)
    arrs = prob.copy_to_array()
    f1, f2 = tmp_path / "a.bin", tmp_path / "b.bin"
    for f in (f1, f2):
        hotfile.write_hotfile(f, arrs, iterations=20, t=0.5, dt=1e-4, host_buffer_count=4)
    r = subprocess.run([sys.executable, "/root/reference/scripts/hotdiff.py", str(f1), str(f2)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    for name in ("Position", "Velocity", "Info", "Hash"):
        assert name in r.stdout
    assert "First difference" not in r.stdout
    # and it does see a difference when there is one
    arrs["vel"][7, 0] += 1.0
    hotfile.write_hotfile(f2, arrs, iterations=20, t=0.5, dt=1e-4, host_buffer_count=4)
    r = subprocess.run([sys.executable, "/root/reference/scripts/hotdiff.py", str(f1), str(f2)],
                       capture_output=True, text=True, timeout=120)
    assert "First difference at particle index 7" in r.stdout


def test_option_dependent_property_buffers_round_trip(tmp_path):
    """Internal Energy / Boundary Elements / Gamma Gradient / Vertices / Volume are stored after Hash, in buffer key order
    (define_buffers.h:98-220), with the element sizes of their traits"""
    prob = DamBreak3D(deltap=0.08, obstacle=False)
    arrs = prob.copy_to_array()
    n = len(arrs["hash"])
    rng = np.random.default_rng(2)
    extra = dict(energy=rng.normal(size=n).astype(np.float32), boundelements=rng.normal(size=(n, 4)).astype(np.float32),
                 gradgamma=rng.normal(size=(n, 4)).astype(np.float32),
                 vertices=rng.integers(0, 2**31, size=(n, 4)).astype(np.uint32), vol=rng.random(size=(n, 4)).astype(np.float32))
    path = tmp_path / "x.bin"
    hotfile.write_hotfile(path, dict(arrs, **extra), iterations=3, t=0.1, dt=1e-4, host_buffer_count=11)
    hf = hotfile.read_hotfile(path)
    for k, v in extra.items():
        assert np.array_equal(hf["arrays"][k].view(np.uint8), v.view(np.uint8)), k
    raw = open(path, "rb").read()
    o, names, sizes = 104, [], []
    for _ in range(9):
        ln, nm, el, cnt = struct.unpack(hotfile.BUFFER, raw[o:o + 76])
        names.append(nm[:ln].decode()); sizes.append(el); o += 76 + el * n
    assert names == ["Position", "Velocity", "Info", "Hash", "Internal Energy", "Boundary Elements", "Gamma Gradient",
                     "Vertices", "Volume"]
    assert sizes == [16, 16, 8, 4, 4, 16, 16, 16, 16] and o == len(raw)


def _resume_case(tmp_path, make, extra_keys, buffer_count, steps=(10, 7)):
    from gpusph_amd.engine import TimestepEngine
    a = TimestepEngine(make())
    a.run(steps[0])
    path = tmp_path / "hot.bin"
    a.save_hotfile(path)
    hf0 = hotfile.read_hotfile(path)
    assert hf0["buffer_count"] == buffer_count
    for k in extra_keys:
        assert k in hf0["arrays"], k
    a.run(steps[1])
    ref = a.download()
    ref_extra = {k: t[:a.n].cpu().numpy() for k, t in a._hot_extra().items()}
    b = TimestepEngine(make())
    b.load_hotfile(path)
    b.run(steps[1])
    out = b.download()
    assert b.n == a.n and b.current_dt() == a.current_dt()
    for k in ("pos", "vel", "info", "hash"):
        assert np.array_equal(np.asarray(out[k]).view(np.uint8), np.asarray(ref[k]).view(np.uint8)), k
    for k, t in b._hot_extra().items():
        assert np.array_equal(t[:b.n].cpu().numpy().view(np.uint8), ref_extra[k].view(np.uint8)), k


@pytest.mark.gpu
def test_resume_with_grenier_volumes_is_bit_identical(tmp_path):
    """SPH_GRENIER: the volumes are the evolving state (the density is recomputed from them): 7 host buffers, Volume stored"""
    from gpusph_amd import defs as D
    _resume_case(tmp_path, lambda: DamBreak3D(0.05, obstacle=False, two_fluids=True, formulation=D.SPH_GRENIER,
                                               viscosity="DYNAMICVISC", density_diffusion=D.DENSITY_DIFFUSION_NONE, jitter=0.1),
                 ["vol"], 7)


@pytest.mark.gpu
def test_resume_with_internal_energy_is_bit_identical(tmp_path):
    _resume_case(tmp_path, lambda: DamBreak3D(0.05, obstacle=False, internal_energy=True, jitter=0.1), ["energy"], 6)


@pytest.mark.gpu
def test_resume_with_k_epsilon_is_bit_identical(tmp_path):
    """KEPSILON on SA walls: + TKE, EPSILON, TURBVISC, EULERVEL: 12 host buffers"""
    from gpusph_amd import defs as D
    from gpusph_amd.problem import SABox
    _resume_case(tmp_path, lambda: SABox(viscosity=dict(rheologytype=D.NEWTONIAN, turbmodel=D.KEPSILON)),
                 ["boundelements", "gradgamma", "vertices", "tke", "eps", "turbvisc", "eulervel"], 12)


@pytest.mark.gpu
def test_resume_with_sa_boundary_is_bit_identical(tmp_path):
    """SA_BOUNDARY: Boundary Elements, Gamma Gradient and Vertices are particle properties: 8 host buffers"""
    from gpusph_amd.problem import SABox
    _resume_case(tmp_path, lambda: SABox(), ["boundelements", "gradgamma", "vertices"], 8)

"""The OpenChannel mirror on the GPU against the same driver over the oracle's kernels (KINEMATICVISC, DYN_BOUNDARY, periodic
along the stream).  First run on an MI355X in round 5 (profiles/r05_first_gpu_call.txt)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sidewalls", [True, False])
def test_openchannel_mirror_follows_the_oracle_driver(sidewalls):
    from gpusph_amd.engine import TimestepEngine
    from gpusph_amd.multigpu import MultiGpuEngine
    from gpusph_amd.problem import OpenChannel, info_id
    from oracle_kernels import OracleKernels
    mk = lambda: OpenChannel(0.05, sidewalls=sidewalls)
    alloc = mk().num_particles + 4096
    ref = MultiGpuEngine(mk(), "cpu", 0, 1, kernels=OracleKernels(mk(), alloc), allocated=alloc)
    eng = TimestepEngine(mk(), device="cuda:0", allocated=alloc)
    for _ in range(12):
        ref.step(); eng.step()
    n = ref.n_local
    assert eng.n_local == n
    a = np.argsort(info_id(eng.info[:n].cpu().numpy().view(np.uint16)), kind="stable")
    b = np.argsort(info_id(ref.info[:n].numpy().view(np.uint16)), kind="stable")
    p = eng.problem
    gp = p.global_pos(eng.pos[:n].cpu().numpy(), eng.hash[:n].cpu().numpy().view(np.uint32))[a]
    gr = p.global_pos(ref.pos[:n].numpy(), ref.hash[:n].numpy().view(np.uint32))[b]
    d = gp - gr
    d[:, 0] -= np.round(d[:, 0] / p.l) * p.l                      # periodic along the stream
    assert np.abs(d).max() < 2e-5 * float(p.m_cellsize[0])
    v, w = eng.vel[:n].cpu().numpy()[a], ref.vel[:n].numpy()[b]
    assert np.abs(v[:, :3] - w[:, :3]).max() < 1e-4 * max(np.abs(w[:, :3]).max(), 1e-3)
    assert np.abs(v[:, 3] - w[:, 3]).max() < 2e-6
    assert float(np.float32(eng.current_dt())) == pytest.approx(float(np.float32(ref.current_dt())), rel=1e-5)

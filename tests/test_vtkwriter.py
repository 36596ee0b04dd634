"""VTK particle files: array names/types of the reference's writer, offsets, XML well-formedness, values."""
import xml.dom.minidom
import numpy as np

from gpusph_amd import vtkwriter, defs as D
from gpusph_amd.problem import DamBreak3D


def test_vtp_round_trip_and_layout(tmp_path):
    prob = DamBreak3D(deltap=0.06, obstacle=True, testpoints=[(0.2, 0.3, 0.2)])
    st = prob.copy_to_array()
    n = len(st["hash"])
    rng = np.random.default_rng(0)
    st["vel"][:, :3] = rng.normal(size=(n, 3)).astype(np.float32)
    vort = rng.normal(size=(n, 3)).astype(np.float32)
    path = tmp_path / "PART_00001.vtp"
    vtkwriter.write_vtp(path, prob, st, vorticity=vort)
    raw = open(path, "rb").read()
    head = raw[:raw.index(b"<AppendedData")].decode() + "</VTKFile>"
    xml.dom.minidom.parseString(head)                      # the XML part is well formed
    got = vtkwriter.read_vtp(path)
    names = list(got)
    assert names == ["Position", "Pressure", "Velocity", "Density", "Mass", "Part type", "Part flags", "Part object",
                     "Part id", "CellIndex", "Vorticity", "connectivity", "offsets"]     # VTKWriter.cc:630-815 order
    assert got["Position"].dtype == np.float64 and got["Part type"].dtype == np.uint8 and got["Part id"].dtype == np.uint32
    gp = prob.global_pos(st["pos"], st["hash"])
    assert np.array_equal(got["Position"], gp)
    assert np.array_equal(got["Velocity"], st["vel"][:, :3]) and np.array_equal(got["Vorticity"], vort)
    assert np.array_equal(got["Part id"], np.arange(n, dtype=np.uint32)[np.argsort(np.argsort(got["Part id"]))])
    tp = got["Part type"] == D.PT_TESTPOINT
    assert tp.sum() == 1 and np.isnan(got["Density"][tp]).all()
    fluid = got["Part type"] == D.PT_FLUID
    rho = (st["vel"][:, 3] + 1) * np.float32(1000)
    assert np.array_equal(got["Density"][fluid], rho[fluid])
    B, gam = np.float32(prob.physparams.bcoeff[0]), np.float32(prob.physparams.gammacoeff[0])
    assert np.allclose(got["Pressure"][fluid], B * ((st["vel"][fluid, 3] + 1) ** gam - 1), rtol=1e-6, atol=1e-3)
    body = (got["Part flags"] & (D.FG_COMPUTE_FORCE >> 3)) != 0
    assert body.sum() == prob.num_obstacle
    assert np.array_equal(got["offsets"], np.arange(1, n + 1, dtype=np.uint32))

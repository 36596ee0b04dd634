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


def test_array_names_order_and_types_follow_the_reference_writer_source(tmp_path):
    """Not our restatement of the layout but the reference's own writer: the sequence of `appender.append_data(..., "Name"...)`
    calls of src/writers/VTKWriter.cc gives the order of the arrays in a GPUSPH file, the return types of the accessor
    functions it passes give their element types.  Every array we write must appear there, in that order, with that type."""
    import os, re
    import pytest
    src = "/root/reference/src/writers/VTKWriter.cc"
    if not os.path.exists(src):
        pytest.skip("needs the GPUSPH tree")
    text = open(src).read()
    body = text[text.index("VTKAppender appender(fid, info, gdata, node_offset, numParts);"):]
    body = body[:body.index("</AppendedData>")] if "</AppendedData>" in body else body
    ref_names = []
    accessor = {}
    calls = [m.start() for m in re.finditer(r"appender\.append_(?:local_)?data\(", body)]
    for a, b in zip(calls, calls[1:] + [len(body)]):
        args = body[a:b].split(";")[0].split("[")[0]            # one call: up to its ';' (or to a lambda argument)
        names = re.findall(r'"([^"]+)"', args)
        ref_names += names
        fn = re.search(r",\s*(get_\w+|demote_w|id)\s*\)\s*$", args.strip())
        if fn and names:
            accessor[names[0]] = fn.group(1)
    assert ref_names[:4] == ["Position", "Neibs", "NextID", "Internal Energy"] and "CellIndex" in ref_names
    ctype = {"float": np.float32, "uchar": np.uint8, "ushort": np.uint16, "uint": np.uint32}
    ret = {fn: ctype[t] for t, fn in re.findall(r"^(float|uchar|ushort|uint)\s+(get_\w+|demote_w)\(", text, re.M)}
    # a state with every optional array this writer knows
    from gpusph_amd.problem import SABox
    prob = DamBreak3D(deltap=0.08, obstacle=True)
    prob.physparams.add_fluid(900.0); prob.physparams.set_equation_of_state(1, 7.0, 30.0)       # two fluids: "Fluid number"
    st = prob.copy_to_array()
    n = len(st["hash"])
    z4 = np.zeros((n, 4), dtype=np.float32)
    arrays = vtkwriter.particle_arrays(prob, st, vorticity=z4[:, :3], normals=z4, forces=z4, gradgamma=z4,
                                       vertices=np.zeros((n, 4), dtype=np.uint32))
    ours = [nm for nm, _ in arrays] + ["connectivity", "offsets"]
    assert {"Spatial acceleration", "Continuity derivative", "Gradient Gamma", "Gamma", "Fluid number", "Part object", "Vertices",
            "Vorticity", "Normals", "Criteria"} <= set(ours)
    pos_in_ref = [ref_names.index(nm) for nm in ours]            # every name exists in the reference ...
    assert pos_in_ref == sorted(pos_in_ref), list(zip(ours, pos_in_ref))      # ... in the same order
    for nm, a in arrays:
        if nm in accessor and accessor[nm] in ret:
            want = ret[accessor[nm]] if not (nm == "Part object") else np.uint8      # get_object_few for < 255 objects
            assert a.dtype == want, (nm, a.dtype, want)
    assert dict(arrays)["Part id"].dtype == np.uint32 and dict(arrays)["Position"].dtype == np.float64    # id(): uint; double4 positions

"""VTK PolyData (.vtp) particle files with the arrays and names GPUSPH's VTKWriter produces
(src/writers/VTKWriter.cc:610-830): appended raw data, every array prefixed by a uint32 byte count.

  Points    Position   Float64 x3 (global coordinates, the writer's double4 BUFFER_POS_GLOBAL)
  PointData Pressure (test points keep the sampled pressure), Velocity, Density (physical; NaN for test points),
            Mass, [Gradient Gamma + Gamma], Part type, Part flags (shifted down by PART_FLAG_SHIFT), [Fluid number],
            [Part object], Part id, [Vertices], CellIndex, [Vorticity], [Normals + Criteria], [Spatial acceleration + Continuity derivative]
  Verts     connectivity, offsets
Existing ParaView states / scripts written for GPUSPH output read these files unchanged.
"""
import numpy as np

from . import defs as D

_TYPES = {np.dtype(np.uint8): "UInt8", np.dtype(np.uint16): "UInt16", np.dtype(np.uint32): "UInt32",
          np.dtype(np.float32): "Float32", np.dtype(np.float64): "Float64"}


def particle_arrays(problem, state, vorticity=None, normals=None, forces=None, gradgamma=None, vertices=None):
    """the (name, array) list in the reference's order; `state` = dict pos/vel/info/hash (cell-local pos)"""
    pos, vel = state["pos"], state["vel"]
    info = np.asarray(state["info"]).reshape(-1, 4)
    n = len(pos)
    pp = problem.physparams
    gpos = problem.global_pos(pos, state["hash"])
    ptype = (info[:, 0] & 7).astype(np.uint8)
    fl = (info[:, 1] >> 12).astype(np.int64)
    rho0 = np.asarray(pp.rho0, dtype=np.float32)[fl]
    B = np.asarray(pp.bcoeff, dtype=np.float32)[fl]
    gam = np.asarray(pp.gammacoeff, dtype=np.float32)[fl]
    ratio = vel[:, 3] + np.float32(1.0)
    tp = ptype == D.PT_TESTPOINT
    pressure = np.where(tp, vel[:, 3], B * (np.power(ratio, gam) - np.float32(1.0))).astype(np.float32)
    density = np.where(tp, np.float32(np.nan), ratio * rho0).astype(np.float32)
    out = [("Position", gpos.astype(np.float64))]
    if forces is not None:
        out += [("Spatial acceleration", np.ascontiguousarray(forces[:, :3], dtype=np.float32)),
                ("Continuity derivative", np.ascontiguousarray(forces[:, 3], dtype=np.float32))]
    out += [("Pressure", pressure), ("Velocity", np.ascontiguousarray(vel[:, :3], dtype=np.float32)), ("Density", density),
            ("Mass", pos[:, 3].astype(np.float32))]
    if gradgamma is not None:      # SA_BOUNDARY (VTKWriter.cc:669-672)
        out += [("Gradient Gamma", np.ascontiguousarray(gradgamma[:, :3], dtype=np.float32)),
                ("Gamma", np.ascontiguousarray(gradgamma[:, 3], dtype=np.float32))]
    out += [("Part type", ptype), ("Part flags", ((info[:, 0] >> 3) & 0xFF).astype(np.uint8))]
    if pp.numFluids() > 1:
        out.append(("Fluid number", fl.astype(np.uint8)))
    if problem.simparams.numbodies > 0:
        out.append(("Part object", (info[:, 1] & 0xFFF).astype(np.uint8)))
    out.append(("Part id", (info[:, 2].astype(np.uint32) | (info[:, 3].astype(np.uint32) << 16))))
    if vertices is not None:       # SA_BOUNDARY: vertexinfo = uint4 (VTKWriter.cc:743-745)
        out.append(("Vertices", np.ascontiguousarray(vertices, dtype=np.uint32).reshape(-1, 4)))
    out.append(("CellIndex", (np.asarray(state["hash"]).astype(np.uint32) & np.uint32(D.CELLTYPE_BITMASK))))
    if vorticity is not None:
        out.append(("Vorticity", np.ascontiguousarray(vorticity, dtype=np.float32)))
    if normals is not None:
        out += [("Normals", np.ascontiguousarray(normals[:, :3], dtype=np.float32)),
                ("Criteria", np.ascontiguousarray(normals[:, 3], dtype=np.float32))]
    assert all(len(a) == n for _, a in out)
    return out


def write_vtp(path, problem, state, **extra):
    arrays = particle_arrays(problem, state, **extra)
    n = len(state["pos"])
    verts = [("connectivity", np.arange(n, dtype=np.uint32)), ("offsets", np.arange(1, n + 1, dtype=np.uint32))]
    offset = 0
    lines = ["<?xml version='1.0'?>", "<VTKFile type='PolyData'  version='0.1'  byte_order='LittleEndian'>", " <PolyData>",
             "  <Piece NumberOfPoints='%d' NumberOfVerts='%d'>" % (n, n)]

    def header(name, a):
        nonlocal offset
        comps = 1 if a.ndim == 1 else a.shape[1]
        s = "\t<DataArray type='%s' Name='%s'" % (_TYPES[a.dtype], name)
        if comps > 1:
            s += " NumberOfComponents='%d'" % comps
        s += " format='appended' offset='%d'/>" % offset
        offset += a.nbytes + 4
        return s

    lines.append("   <Points>"); lines.append(header(*arrays[0])); lines.append("   </Points>")
    lines.append("   <PointData Scalars='Pressure' Vectors='Velocity'>")
    lines += [header(nm, a) for nm, a in arrays[1:]]
    lines.append("   </PointData>")
    lines.append("   <Verts>"); lines += [header(nm, a) for nm, a in verts]; lines.append("   </Verts>")
    lines += ["  </Piece>", " </PolyData>", " <AppendedData encoding='raw'>"]
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n_").encode())
        for _, a in arrays + verts:
            f.write(np.uint32(a.nbytes).tobytes())
            f.write(np.ascontiguousarray(a).tobytes())
        f.write(b" </AppendedData>\n</VTKFile>\n")


def read_vtp(path):
    """minimal reader of the files above (tests, quick looks): name -> array"""
    import re
    raw = open(path, "rb").read()
    cut = raw.index(b"<AppendedData encoding='raw'>")
    head = raw[:cut].decode()
    base = raw.index(b"_", cut) + 1
    n = int(re.search(r"NumberOfPoints='(\d+)'", head).group(1))
    inv = {v: k for k, v in _TYPES.items()}
    out = {}
    for m in re.finditer(r"<DataArray type='(\w+)' Name='([^']+)'(?: NumberOfComponents='(\d+)')? format='appended' offset='(\d+)'/>", head):
        dt, name, comps, off = inv[m.group(1)], m.group(2), int(m.group(3) or 1), int(m.group(4))
        nbytes = int(np.frombuffer(raw, np.uint32, 1, base + off)[0])
        a = np.frombuffer(raw, dt, nbytes // dt.itemsize, base + off + 4)
        out[name] = a.reshape(n, comps) if comps > 1 else a
    return out

"""GPUSPH HotFile (version 1) reader / writer: the reference's checkpoint format (src/writers/HotFile.{h,cc}).

Layout (native little-endian, C struct padding), as scripts/hotdiff.py of the reference decodes it:
  header_t         '@IIIII48xLdf12x'   version, buffer_count, particle_count, body_count, numOpenBoundaries,
                                       reserved[12], iterations, t, dt, _reserved[3]              (HotFile.h:45-56)
  per stored buffer: encoded_buffer_t '@I64sII' (name_length, name, element_size, array_count)    (HotFile.cc:40-45)
                     + element_size*particle_count bytes of the first array                      (:241-272)
  per body:          encoded_body_t    (:59-73)
Stored buffers are the non-ephemeral host buffers (define_buffers.h:279-331: particle properties and support buffers)
in key order -- "Position" (float4, cell-local), "Velocity" (float4), "Info" (ushort4), "Hash" (uint), then, with the
options that allocate them (GPUSPH::allocateGlobalHostBuffers, GPUSPH.cc:868-941), "Internal Energy" (float),
"Boundary Elements", "Gamma Gradient" (float4), "Vertices" (uint4), the k-epsilon fields "Turbulent Kinetic Energy [k]",
"Turbulent Dissipation Rate [e]", "Eddy Viscosity" (float), "Eulerian velocity" (float4) and "Volume" (float4) -- while
header.buffer_count counts ALL host buffers (that includes the ephemeral "Position (double precision)", the SPS and
effective viscosities and Grenier's sigma: HotFile.cc:91-101 notes the mismatch).

This is a data format on either side of the hot path: a state saved by a GPUSPH CUDA run can be loaded into
TimestepEngine (and vice versa), and two runs can be compared with the reference's own scripts/hotdiff.py.
"""
import struct
import numpy as np

HEADER = "@IIIII48xLdf12x"
BUFFER = "@I64sII"
BODY = "@IIIIii26d10f"      # index, id, type, numparts, firstindex, lastindex, 26 doubles, reserved[10]
MB_FLOATING, MB_FORCES_MOVING, MB_MOVING = 0, 1, 2

# name -> (dtype, components); order = buffer key order (src/define_buffers.h:48-58)
STORED = [("Position", np.float32, 4), ("Velocity", np.float32, 4), ("Info", np.uint16, 4), ("Hash", np.uint32, 1),
          ("Internal Energy", np.float32, 1), ("Boundary Elements", np.float32, 4), ("Gamma Gradient", np.float32, 4),
          ("Vertices", np.uint32, 4), ("Turbulent Kinetic Energy [k]", np.float32, 1), ("Turbulent Dissipation Rate [e]", np.float32, 1),
          ("Eddy Viscosity", np.float32, 1), ("Eulerian velocity", np.float32, 4), ("Volume", np.float32, 4)]
KEYS = {"Position": "pos", "Velocity": "vel", "Info": "info", "Hash": "hash", "Internal Energy": "energy",
        "Boundary Elements": "boundelements", "Gamma Gradient": "gradgamma", "Vertices": "vertices", "Volume": "vol",
        "Turbulent Kinetic Energy [k]": "tke", "Turbulent Dissipation Rate [e]": "eps", "Eddy Viscosity": "turbvisc",
        "Eulerian velocity": "eulervel"}
REQUIRED = ("pos", "vel", "info", "hash")


def write_hotfile(path, arrays, iterations, t, dt, bodies=(), num_open_boundaries=0, host_buffer_count=5):
    """arrays: dict pos/vel/info/hash (n rows) [+ energy, boundelements, gradgamma, vertices, vol]; bodies: iterable of dicts (index, id, type, numparts, firstindex,
    lastindex, crot, lvel, avel, orientation [+ initial_*])."""
    n = len(arrays["hash"])
    with open(path, "wb") as f:
        f.write(struct.pack(HEADER, 1, host_buffer_count, n, len(bodies), num_open_boundaries, int(iterations),
                            float(t), float(dt)))
        for name, dtype, comps in STORED:
            if KEYS[name] not in arrays:
                if KEYS[name] in REQUIRED:
                    raise KeyError(KEYS[name])
                continue
            a = np.ascontiguousarray(arrays[KEYS[name]]).view(dtype).reshape(n, comps)
            f.write(struct.pack(BUFFER, len(name), name.encode(), a.dtype.itemsize * comps, 1))
            f.write(a.tobytes())
        for b in bodies:
            k = lambda key, m: list(np.asarray(b.get(key, b.get(key.replace("initial_", ""), [0.0] * m)), dtype=np.float64))
            vals = k("crot", 3) + k("lvel", 3) + k("avel", 3) + k("orientation", 4) + \
                k("initial_crot", 3) + k("initial_lvel", 3) + k("initial_avel", 3) + k("initial_orientation", 4)
            f.write(struct.pack(BODY, int(b["index"]), int(b["id"]), int(b["type"]), int(b["numparts"]),
                                int(b.get("firstindex", 0)), int(b.get("lastindex", 0)), *vals, *([0.0] * 10)))


def read_hotfile(path):
    """-> dict(version, buffer_count, particles, iterations, t, dt, num_open_boundaries, arrays{pos,vel,info,hash,...}, bodies[])"""
    with open(path, "rb") as f:
        raw = f.read()
    o = struct.calcsize(HEADER)
    version, nbuf, n, nbodies, nob, iterations, t, dt = struct.unpack(HEADER, raw[:o])
    if version != 1:
        raise ValueError("unsupported HotFile version %d" % version)
    body_sz, buf_sz = struct.calcsize(BODY), struct.calcsize(BUFFER)
    end_of_buffers = len(raw) - nbodies * body_sz
    arrays = {}
    while o < end_of_buffers:                 # header.buffer_count is not the stored count (see the module docstring)
        name_len, name, elsize, count = struct.unpack(BUFFER, raw[o:o + buf_sz])
        o += buf_sz
        name = name[:name_len].decode()
        data = raw[o:o + elsize * n]
        if len(data) != elsize * n:
            raise ValueError("truncated HotFile buffer '%s'" % name)
        o += elsize * n
        known = {nm: (dt_, c) for nm, dt_, c in STORED}
        if name in known:
            dt_, c = known[name]
            a = np.frombuffer(data, dtype=dt_).reshape(n, c).copy()
            arrays[KEYS[name]] = a[:, 0] if c == 1 else a
        else:
            arrays[name] = np.frombuffer(data, dtype=np.uint8).reshape(n, elsize).copy()
    bodies = []
    for _ in range(nbodies):
        v = struct.unpack(BODY, raw[o:o + body_sz]); o += body_sz
        d = v[6:32]
        bodies.append(dict(index=v[0], id=v[1], type=v[2], numparts=v[3], firstindex=v[4], lastindex=v[5],
                           crot=d[0:3], lvel=d[3:6], avel=d[6:9], orientation=d[9:13],
                           initial_crot=d[13:16], initial_lvel=d[16:19], initial_avel=d[19:22], initial_orientation=d[22:26]))
    return dict(version=version, buffer_count=nbuf, particles=n, iterations=iterations, t=t, dt=dt,
                num_open_boundaries=nob, arrays=arrays, bodies=bodies)

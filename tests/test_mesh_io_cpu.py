"""PLY ingestion (src/shapes/ply.cpp) by the C++ host library: every storage variant of the same mesh against an
independent numpy parse/expectation; normal regeneration (src/render/mesh.cpp:1218-1267) against a numpy restatement;
error behaviour.  No GPU involved."""
import os
import struct

import numpy as np
import pytest

TYPES = {"float": ("f", 4, np.float32), "double": ("d", 8, np.float64), "uchar": ("B", 1, np.uint8), "int": ("i", 4, np.int32),
         "uint": ("I", 4, np.uint32), "ushort": ("H", 2, np.uint16), "short": ("h", 2, np.int16)}


def write_ply(path, fmt, P, F, N=None, UV=None, coord_type="float", uv_names=("u", "v"), index_type="int", count_type="uchar",
              extra_vertex=None, extra_face=False, comment=True, crlf=False, extra_element=False):
    """minimal PLY writer used as the test-side source of truth"""
    nl = "\r\n" if crlf else "\n"
    h = ["ply", "format %s 1.0" % fmt]
    if comment:
        h += ["comment written by tests/test_mesh_io_cpu.py", "obj_info something"]
    h += ["element vertex %d" % len(P)] + ["property %s %s" % (coord_type, c) for c in "xyz"]
    if N is not None:
        h += ["property float n%s" % c for c in "xyz"]
    if extra_vertex:
        h += ["property uchar red", "property uchar green", "property uchar blue"]
    if UV is not None:
        h += ["property float %s" % uv_names[0], "property float %s" % uv_names[1]]
    if extra_element:
        h += ["element material 2", "property float shininess"]
    h += ["element face %d" % len(F), "property list %s %s vertex_indices" % (count_type, index_type)]
    if extra_face:
        h += ["property uchar flags"]
    h += ["end_header"]
    head = (nl.join(h) + nl).encode()
    body = b""
    if fmt == "ascii":
        lines = []
        for i in range(len(P)):
            row = ["%.9g" % x for x in P[i]]
            if N is not None: row += ["%.9g" % x for x in N[i]]
            if extra_vertex: row += ["10", "20", "30"]
            if UV is not None: row += ["%.9g" % x for x in UV[i]]
            lines.append(" ".join(row))
        if extra_element: lines += ["0.5", "0.25"]
        for i in range(len(F)):
            row = ["3"] + [str(int(x)) for x in F[i]]
            if extra_face: row += ["7"]
            lines.append(" ".join(row))
        body = (nl.join(lines) + nl).encode()
    else:
        e = "<" if fmt == "binary_little_endian" else ">"
        cf = TYPES[coord_type][0]
        for i in range(len(P)):
            body += struct.pack(e + cf * 3, *[float(x) for x in P[i]])
            if N is not None: body += struct.pack(e + "fff", *[float(x) for x in N[i]])
            if extra_vertex: body += struct.pack("BBB", 10, 20, 30)
            if UV is not None: body += struct.pack(e + "ff", *[float(x) for x in UV[i]])
        if extra_element: body += struct.pack(e + "ff", 0.5, 0.25)
        for i in range(len(F)):
            body += struct.pack(e + TYPES[count_type][0], 3) + struct.pack(e + TYPES[index_type][0] * 3, *[int(x) for x in F[i]])
            if extra_face: body += struct.pack("B", 7)
    with open(path, "wb") as f:
        f.write(head + body)


def grid_mesh(n=5):
    xs, ys = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    P = np.stack([xs.ravel(), ys.ravel(), 0.3 * np.sin(3 * xs.ravel()) * np.cos(2 * ys.ravel())], 1).astype(np.float32)
    F = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i; F += [[a, a + 1, a + n], [a + 1, a + n + 1, a + n]]
    N = np.tile(np.float32([0, 0, 1]), (len(P), 1)); UV = np.stack([(xs.ravel() + 1) / 2, (ys.ravel() + 1) / 2], 1).astype(np.float32)
    return P, np.asarray(F, np.uint32), N, UV


def numpy_normals(P, F):
    """Mesh::compute_normals restated with numpy in float32 (angle-weighted, Thuermer & Wuethrich)"""
    P = P.astype(np.float32); acc = np.zeros_like(P)
    def unit_angle(a, b):
        d = np.float32((a * b).sum()); am = a if d >= 0 else -a
        t = np.float32(2) * np.arcsin(np.float32(.5) * np.float32(np.linalg.norm((b - am).astype(np.float32))))
        return t if d >= 0 else np.float32(np.pi) - t
    for f in F:
        p = P[f]; n = np.cross(p[1] - p[0], p[2] - p[0]).astype(np.float32); l2 = np.float32((n * n).sum())
        if not l2 > 0: continue
        n = n / np.sqrt(l2)
        for k in range(3):
            e1 = p[(k + 1) % 3] - p[k]; e2 = p[(k + 2) % 3] - p[k]
            acc[f[k]] += n * unit_angle(e1 / np.linalg.norm(e1), e2 / np.linalg.norm(e2))
    l = np.linalg.norm(acc, axis=1, keepdims=True)
    return np.where(l > 0, acc / np.maximum(l, 1e-30), np.float32([1, 0, 0])).astype(np.float32)


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_storage_variants(mi, tmp_path, fmt):
    P, F, N, UV = grid_mesh()
    variants = [dict(), dict(N=N, UV=UV), dict(UV=UV, uv_names=("texture_u", "texture_v")), dict(UV=UV, uv_names=("s", "t"), N=N),
                dict(coord_type="double"), dict(index_type="uint", count_type="ushort"), dict(extra_vertex=True, extra_face=True, N=N, UV=UV),
                dict(crlf=True, comment=False), dict(extra_element=True, UV=UV)]
    for k, kw in enumerate(variants):
        path = os.path.join(tmp_path, "m%d.ply" % k)
        write_ply(path, fmt, P, F, **kw)
        m = mi.core.Mesh("t").from_ply(path)
        assert m.V.shape == (len(P), 8) and m.F.shape == (len(F), 4)
        assert np.array_equal(m.F[:, :3], F) and not m.F[:, 3].any()
        if fmt == "ascii":
            assert np.allclose(m.V[:, :3], P, rtol=1e-7, atol=0)                # %.9g round-trips float32
        else:
            assert np.array_equal(m.V[:, :3], P)                                 # bit-exact (double -> float conversion included)
        if "N" in kw:
            assert np.array_equal(m.V[:, 3:6], N) and m.flags & 1
        else:                                                                    # regenerated (mesh.cpp:355-356)
            assert m.flags & 1 and np.allclose(m.V[:, 3:6], numpy_normals(P, F), atol=2e-6)
        if "UV" in kw:
            assert m.flags & 2 and (np.array_equal(m.V[:, 6:8], UV) if fmt != "ascii" else np.allclose(m.V[:, 6:8], UV, rtol=1e-7))
        else:
            assert not m.flags & 2 and not m.V[:, 6:8].any()


def test_ply_options_and_scene(mi, tmp_path):
    P, F, N, UV = grid_mesh(4)
    path = os.path.join(tmp_path, "g.ply"); write_ply(path, "binary_little_endian", P, F, UV=UV)
    m = mi.core.Mesh("t").from_ply(path, face_normals=True, flip_tex_coords=True)
    assert not m.flags & 1 and not m.V[:, 3:6].any() and np.array_equal(m.V[:, 7], np.float32(1) - UV[:, 1])
    T = mi.ScalarTransform4f
    d = mi.cornell_box()
    d["blob"] = {"type": "ply", "filename": path, "to_world": T().translate([0, -0.5, 0]).scale(0.3), "bsdf": {"type": "ref", "id": "green"}}
    scene = mi.load_dict(d)
    blob = [x for x in scene.meshes if x["key"] == "blob"][0]
    assert blob["V"].shape[0] == len(P) and blob["F"].shape[0] == len(F) and blob["flags"] == 3
    assert np.allclose(blob["V"][:, :3], P * 0.3 + np.float32([0, -0.5, 0]), atol=1e-6)
    assert np.allclose(np.linalg.norm(blob["V"][:, 3:6], axis=1), 1, atol=1e-5)


def test_ply_errors(mi, tmp_path):
    P, F, N, UV = grid_mesh(3)
    def load(name, data):
        path = os.path.join(tmp_path, name)
        with open(path, "wb") as f: f.write(data)
        return mi.core.Mesh("t").from_ply(path)
    with pytest.raises(RuntimeError, match="not found"):
        mi.core.Mesh("t").from_ply(os.path.join(tmp_path, "missing.ply"))
    with pytest.raises(RuntimeError, match="invalid PLY header"):
        load("a.ply", b"plx\nformat ascii 1.0\nend_header\n")
    good = os.path.join(tmp_path, "good.ply"); write_ply(good, "binary_little_endian", P, F)
    data = open(good, "rb").read()
    with pytest.raises(RuntimeError, match="trailing content"):
        load("b.ply", data + b"\0\0")
    with pytest.raises(RuntimeError, match="end of file"):
        load("c.ply", data[:-5])
    quad = (b"ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
            b"property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    with pytest.raises(RuntimeError, match="triangle mesh"):
        load("d.ply", quad)
    with pytest.raises(RuntimeError, match="out of bounds"):
        load("e.ply", quad.replace(b"4 0 1 2 3", b"3 0 1 9"))
    with pytest.raises(RuntimeError, match="invalid vertex position"):
        load("f.ply", quad.replace(b"1 1 0", b"nan 1 0").replace(b"4 0 1 2 3", b"3 0 1 2"))


def test_ply_header_counts_are_bounded_by_the_file(tmp_path):
    """a header-controlled element count must not drive an allocation (std::bad_alloc through the C boundary aborts the process)"""
    import mitsuba3_amd as mi
    hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
           "element face 1\nproperty list uchar int vertex_indices\nend_header\n")
    for n in (2 ** 40, 2 ** 62, 10 ** 9):
        f = tmp_path / ("huge%d.ply" % n); f.write_bytes((hdr % n).encode() + b"\x00" * 64)
        with pytest.raises(Exception) as e:
            mi.load_dict({"type": "ply", "filename": str(f)})
        assert "PLY" in str(e.value)

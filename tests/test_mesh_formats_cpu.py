"""OBJ (src/shapes/obj.cpp + Mesh::from_corners, src/render/mesh_utils.cpp:210-560) and `serialized` (src/shapes/serialized.cpp,
container versions 3 / 4) ingestion by the C++ host library, against test-side writers and an independent restatement of the
corner-welding rules.  Also the order of operations shared by all mesh loaders: to_world / flip_normals are baked BEFORE missing
normals are regenerated (PackedMesh::set_transform, mesh_utils.cpp:33-44).  No GPU involved."""
import os
import struct
import zlib

import numpy as np
import pytest

from tests.test_mesh_io_cpu import grid_mesh, numpy_normals, write_ply


# ----------------------------------------------------------------------------------------------------------------- OBJ

def write_obj(path, P, faces, VT=None, VN=None, header="# test\no thing\ns off\n", crlf=False):
    """faces: list of polygons, each a list of (p, t, n) 0-based index triples (t / n may be None)"""
    nl = "\r\n" if crlf else "\n"
    out = [header.rstrip("\n")]
    out += ["v %.9g %.9g %.9g" % tuple(p) for p in P]
    if VT is not None: out += ["vt %.9g %.9g" % tuple(t) for t in VT]
    if VN is not None: out += ["vn %.9g %.9g %.9g" % tuple(n) for n in VN]
    for poly in faces:
        toks = []
        for (p, t, n) in poly:
            if t is None and n is None: toks.append("%d" % (p + 1))
            elif n is None: toks.append("%d/%d" % (p + 1, t + 1))
            elif t is None: toks.append("%d//%d" % (p + 1, n + 1))
            else: toks.append("%d/%d/%d" % (p + 1, t + 1, n + 1))
        out.append("f " + " ".join(toks))
    with open(path, "wb") as f:
        f.write((nl.join(out) + nl).encode())


def weld_reference(P, faces, VT, VN, face_normals=False, flip_tex_coords=True):
    """what from_corners must produce: (V [n x 8], F [m x 3], position_index)"""
    P = np.asarray(P, np.float32)
    VT = None if VT is None else np.asarray(VT, np.float32).copy()
    if VT is not None and flip_tex_coords: VT[:, 1] = np.float32(1) - VT[:, 1]
    VN = None if VN is None else np.asarray(VN, np.float32)
    has_uv = any(t is not None for poly in faces for (_, t, _) in poly)
    has_n = any(n is not None for poly in faces for (_, _, n) in poly) and not face_normals
    tris = []
    for poly in faces:
        for i in range(1, len(poly) - 1):
            tris.append((poly[0], poly[i], poly[i + 1]))
    def uv_of(t): return (0.0, 0.0) if t is None else tuple(float(x) + 0.0 for x in VT[t])      # + 0.0 folds -0.0
    def n_of(n): return (0.0, 0.0, 0.0) if n is None else tuple(float(x) + 0.0 for x in VN[n])
    corner_keys = []
    for tri in tris:
        flipped = 0
        if has_uv and not face_normals:
            (a, b, c) = [np.float32(uv_of(t)) for (_, t, _) in tri]
            area2 = np.float32(np.float32(b[0] - a[0]) * np.float32(c[1] - a[1])) - np.float32(np.float32(b[1] - a[1]) * np.float32(c[0] - a[0]))
            flipped = 0 if area2 > 0 else 1
        for (p, t, n) in tri:
            corner_keys.append((p, (n_of(n) if has_n else ()) + (uv_of(t) if has_uv else ()) + ((flipped,) if has_uv and not face_normals else ())))
    V, pidx, F = [], [], np.zeros((len(tris), 3), np.uint32)
    for p in range(len(P)):
        local = {}
        for c, (cp, key) in enumerate(corner_keys):
            if cp != p: continue
            if key not in local:
                local[key] = len(V)
                rec = list(P[p]) + (list(key[0:3]) if has_n else [0, 0, 0]) + (list(key[3:5] if has_n else key[0:2]) if has_uv else [0, 0])
                V.append(rec); pidx.append(p)
            F[c // 3, c % 3] = local[key]
    # position ids are dense over REFERENCED points
    used = sorted(set(pidx)); remap = {p: i for i, p in enumerate(used)}
    return np.asarray(V, np.float32).reshape(-1, 8), F, np.asarray([remap[p] for p in pidx], np.uint32), has_n, has_uv


CUBE_P = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1), (5, 5, 5)]   # last: unreferenced
CUBE_N = [(0, 0, -1), (0, 0, 1), (0, -1, 0), (1, 0, 0), (0, 1, 0), (-1, 0, 0)]
CUBE_T = [(0, 0), (1, 0), (1, 1), (0, 1)]
CUBE_Q = [(3, 2, 1, 0), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]


def cube_faces(with_t=True, with_n=True):
    return [[(q[k], k if with_t else None, s if with_n else None) for k in range(4)] for s, q in enumerate(CUBE_Q)]


def check_against_reference(m, ref, P_expected_normals=None):
    V, F, pidx, has_n, has_uv = ref
    assert m.V.shape == V.shape and m.F.shape[0] == F.shape[0]
    assert np.array_equal(m.F[:, :3], F) and not m.F[:, 3].any()
    assert np.allclose(m.V[:, :3], V[:, :3], rtol=1e-7)
    assert np.allclose(m.V[:, 6:8], V[:, 6:8], rtol=1e-7)
    if has_n:
        n = V[:, 3:6] / np.maximum(np.linalg.norm(V[:, 3:6], axis=1, keepdims=True), 1e-30)
        assert np.allclose(m.V[:, 3:6], n, atol=1e-6)
    assert bool(m.flags & 2) == has_uv
    return pidx


def test_obj_cube_quads_weld_and_triangulate(mi, tmp_path):
    path = os.path.join(tmp_path, "cube.obj")
    for crlf in (False, True):
        write_obj(path, CUBE_P, cube_faces(), CUBE_T, CUBE_N, crlf=crlf)
        m = mi.core.Mesh("t").from_obj(path)
        ref = weld_reference(CUBE_P, cube_faces(), CUBE_T, CUBE_N)
        check_against_reference(m, ref)
        assert m.V.shape[0] == 24 and m.F.shape[0] == 12 and m.flags == 3      # 8 points x 3 normals; the 9th point is dropped
        # quads split along corners 0-2; vertex ids follow the source point order
        assert np.array_equal(m.V[:3, :3], np.float32([CUBE_P[0]] * 3)) and np.array_equal(m.V[21:24, :3], np.float32([CUBE_P[7]] * 3))
        assert np.array_equal(m.V[:, 7][m.V[:, 7] != 0], np.ones((m.V[:, 7] != 0).sum(), np.float32))   # v -> 1 - v


def test_obj_corner_forms_and_missing_indices(mi, tmp_path):
    path = os.path.join(tmp_path, "forms.obj")
    P, Fi, _, UV = grid_mesh(4)
    rng = np.random.default_rng(3)
    N = rng.normal(size=(7, 3)).astype(np.float32)
    polys = []
    for k, f in enumerate(Fi):
        mode = k % 4
        polys.append([(int(p), (int(p) if mode in (1, 3) else None), (int((p + k) % 7) if mode in (2, 3) else None)) for p in f])
    write_obj(path, P, polys, UV, N)
    for fn in (False, True):
        for flip in (True, False):
            m = mi.core.Mesh("t").from_obj(path, face_normals=fn, flip_tex_coords=flip)
            ref = weld_reference(P, polys, UV, N, face_normals=fn, flip_tex_coords=flip)
            check_against_reference(m, ref)
            assert bool(m.flags & 1) == (not fn)
    # pentagon fan + triangle, positions only: normals regenerated, nothing splits
    P5 = [(np.cos(a), np.sin(a), 0.1 * k) for k, a in enumerate(np.linspace(0, 2 * np.pi, 6)[:-1])] + [(0, 0, 1)]
    polys = [[(k, None, None) for k in range(5)], [(0, None, None), (1, None, None), (5, None, None)]]
    write_obj(path, P5, polys)
    m = mi.core.Mesh("t").from_obj(path)
    assert np.array_equal(m.F[:, :3], np.uint32([[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 1, 5]])) and m.flags == 1
    assert np.allclose(m.V[:, 3:6], numpy_normals(np.float32(P5), m.F[:, :3]), atol=2e-6)


def test_obj_regenerated_normals_are_per_surface_point(mi, tmp_path):
    """a UV seam splits vertices but not the surface: the regenerated normal is shared (mesh.cpp:573-582)"""
    P, Fi, _, UV = grid_mesh(5)
    VT = np.concatenate([UV, UV + np.float32([0.5, 0.25])])          # second chart: the same points get other texcoords
    polys = [[(int(p), int(p) + (len(P) if k % 2 else 0), None) for p in f] for k, f in enumerate(Fi)]
    path = os.path.join(tmp_path, "seam.obj"); write_obj(path, P, polys, VT)
    T = mi.ScalarTransform4f
    for tw in (None, T().translate([0.5, 0, 1]).scale([1, 3, 0.5]), T().scale([-1, 1, 1])):
        for flip in (False, True):
            m = mi.core.Mesh("t").from_obj(path, to_world=tw, flip_normals=flip)
            V, F, pidx, _, _ = weld_reference(P, polys, VT, None)
            assert m.V.shape[0] == len(V) > len(P) and np.array_equal(np.sort(m.F[:, :3], axis=1), np.sort(F, axis=1))
            M = np.eye(4, dtype=np.float32) if tw is None else np.asarray(tw.matrix, np.float32).reshape(4, 4)
            Pw = (P @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
            assert np.allclose(m.V[:, :3], Pw[[sorted(set(range(len(P))))[i] for i in pidx]], atol=1e-6)
            reverse = (np.linalg.det(M[:3, :3]) < 0) != flip
            Fw = F[:, ::-1] if reverse else F
            assert np.array_equal(m.F[:, :3], Fw)
            # normals: computed on the TRANSFORMED surface points with the FINAL winding, shared across the seam
            expect = numpy_normals(Pw, pidx[Fw])
            assert np.allclose(m.V[:, 3:6], expect[pidx], atol=5e-6)


def test_mesh_loaders_bake_transform_before_normals(mi, tmp_path):
    """same rule for PLY: non-uniform scaling changes the angle weights, so the order matters"""
    P, F, N, UV = grid_mesh(5)
    path = os.path.join(tmp_path, "g.ply"); write_ply(path, "binary_little_endian", P, F)
    T = mi.ScalarTransform4f
    tw = T().rotate([0, 0, 1], 30).scale([1, 4, 0.25])
    M = np.asarray(tw.matrix, np.float32).reshape(4, 4)
    Pw = (P @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
    m = mi.core.Mesh("t").from_ply(path, to_world=tw)
    assert np.allclose(m.V[:, :3], Pw, atol=1e-6) and np.allclose(m.V[:, 3:6], numpy_normals(Pw, F), atol=5e-6)
    m2 = mi.core.Mesh("t").from_ply(path, to_world=tw, flip_normals=True)
    assert np.array_equal(m2.F[:, :3], F[:, ::-1]) and np.allclose(m2.V[:, 3:6], -m.V[:, 3:6], atol=5e-6)
    # stored normals: inverse transpose, normalised, negated by flip_normals; zero-length normals are left alone
    Nn = (N * np.float32(2.5)); Nn[0] = 0
    write_ply(path, "binary_little_endian", P, F, N=Nn)
    m3 = mi.core.Mesh("t").from_ply(path, to_world=tw, flip_normals=True)
    it = np.linalg.inv(M[:3, :3].astype(np.float64)).T
    e = Nn.astype(np.float64) @ it.T; e[1:] /= np.linalg.norm(e[1:], axis=1, keepdims=True)
    assert np.allclose(m3.V[1:, 3:6], -e[1:], atol=1e-5) and not m3.V[0, 3:6].any()


def test_obj_random_polygon_soup_welds_like_the_contract(mi, tmp_path):
    """stress of the hash-table welding: 600 points, 1500 polygons of 3..7 sides, few distinct normals / texcoords (many repeats, -0.0 vs +0.0
    texcoords, mirrored UV triangles, missing indices in some corners, a third of the points never referenced)"""
    rng = np.random.default_rng(11)
    P = rng.uniform(-1, 1, (600, 3)).astype(np.float32)
    VT = rng.integers(-2, 3, (9, 2)).astype(np.float32) * np.float32(0.5); VT[3] = (-0.0, 0.0); VT[4] = (0.0, -0.0)
    VN = rng.normal(size=(5, 3)).astype(np.float32); VN[2] = (0.0, -0.0, 1.0)
    polys = []
    for _ in range(1500):
        sides = int(rng.integers(3, 8))
        pts = rng.choice(400, sides, replace=False)              # points 400..599 stay unreferenced
        mode = int(rng.integers(0, 4))
        polys.append([(int(q), int(rng.integers(0, 9)) if mode in (1, 3) else None, int(rng.integers(0, 5)) if mode in (2, 3) else None) for q in pts])
    path = os.path.join(tmp_path, "soup.obj")
    write_obj(path, P, polys, VT, VN)
    for fn in (False, True):
        m = mi.core.Mesh("t").from_obj(path, face_normals=fn)
        ref = weld_reference(P, polys, VT, VN, face_normals=fn)
        check_against_reference(m, ref)
        assert m.F.shape[0] == sum(len(q) - 2 for q in polys)


def test_obj_errors(mi, tmp_path):
    def load(text, **kw):
        path = os.path.join(tmp_path, "e.obj")
        with open(path, "wb") as f: f.write(text.encode())
        return mi.core.Mesh("t").from_obj(path, **kw)
    tri = "v 0 0 0\nv 1 0 0\nv 0 1 0\n"
    assert load(tri + "f 1 2 3\n").F.shape[0] == 1
    assert load(tri + "f 1 2\n").F.shape[0] == 0                                # degenerate polygon: no triangle
    with pytest.raises(RuntimeError, match="invalid vertex 4"): load(tri + "f 1 2 4\n")
    with pytest.raises(RuntimeError, match="invalid vertex 0"): load(tri + "f 0 1 2\n")
    with pytest.raises(RuntimeError, match="invalid vertex -1"): load(tri + "f -1 -2 -3\n")      # relative indices: refused like the reference (strtoul wraps)
    with pytest.raises(RuntimeError, match="invalid texture coordinate 2"): load(tri + "vt 0 0\nf 1/1 2/2 3/1\n")
    with pytest.raises(RuntimeError, match="invalid normal 1"): load(tri + "f 1//1 2//1 3//1\n")
    assert load(tri + "f 1//1 2//1 3//1\n", face_normals=True).F.shape[0] == 1   # normals are ignored entirely (obj.cpp:176,262)
    with pytest.raises(RuntimeError, match="could not parse line"): load("v 0 0 abc\n")
    with pytest.raises(RuntimeError, match="could not parse line"): load(tri + "f 1/1/1/1 2 3\n")
    with pytest.raises(RuntimeError, match="invalid vertex position"): load("v 0 nan 0\n")
    with pytest.raises(RuntimeError, match="excessively long line"): load("# " + "x" * 2000 + "\n" + tri)
    with pytest.raises(RuntimeError, match="unreadable"): mi.core.Mesh("t").from_obj(os.path.join(tmp_path, "missing.obj"))


# ---------------------------------------------------------------------------------------------------------- serialized

def serialized_blob(version, P, F, N=None, UV=None, colors=None, double=False, name="mesh", face_normals_flag=False):
    flags = (0x2000 if double else 0x1000) | (1 if N is not None else 0) | (2 if UV is not None else 0) | (8 if colors is not None else 0) | (0x10 if face_normals_flag else 0)
    ft = "<f8" if double else "<f4"
    payload = struct.pack("<I", flags)
    if version == 4: payload += name.encode() + b"\0"
    payload += struct.pack("<QQ", len(P), len(F)) + np.asarray(P).astype(ft).tobytes()
    for a in (N, UV, colors):
        if a is not None: payload += np.asarray(a).astype(ft).tobytes()
    payload += np.asarray(F).astype("<u4").tobytes()
    return struct.pack("<HH", 0x041C, version) + zlib.compress(payload)


def write_serialized(path, version, blobs):
    data, offsets = b"", []
    for b in blobs:
        offsets.append(len(data)); data += b
    data += b"".join(struct.pack("<Q" if version == 4 else "<I", o) for o in offsets) + struct.pack("<I", len(blobs))
    with open(path, "wb") as f: f.write(data)


@pytest.mark.parametrize("version", [3, 4])
def test_serialized_variants(mi, tmp_path, version):
    P, F, N, UV = grid_mesh(6)
    P2, F2, N2, UV2 = grid_mesh(3)
    col = np.tile(np.float32([0.2, 0.4, 0.6]), (len(P), 1))
    path = os.path.join(tmp_path, "m.serialized")
    write_serialized(path, version, [serialized_blob(version, P, F, N, UV, col), serialized_blob(version, P2, F2, double=True),
                                     serialized_blob(version, P2 * 2, F2, N=N2 * 3, double=True)])
    m = mi.core.Mesh("t").from_serialized(path)
    assert np.array_equal(m.V[:, :3], P) and np.array_equal(m.V[:, 3:6], N) and np.array_equal(m.V[:, 6:8], UV) and np.array_equal(m.F[:, :3], F) and m.flags == 3
    m = mi.core.Mesh("t").from_serialized(path, shape_index=1)                 # doubles are narrowed; normals regenerated
    assert np.array_equal(m.V[:, :3], P2) and np.allclose(m.V[:, 3:6], numpy_normals(P2, F2), atol=2e-6) and m.flags == 1
    m = mi.core.Mesh("t").from_serialized(path, shape_index=2)                 # stored normals are normalised (mesh_utils.cpp:113-115)
    assert np.array_equal(m.V[:, :3], P2 * 2) and np.allclose(m.V[:, 3:6], N2, atol=1e-6)
    m = mi.core.Mesh("t").from_serialized(path, shape_index=0, face_normals=True)
    assert m.flags == 2 and not m.V[:, 3:6].any()
    with pytest.raises(RuntimeError, match="out of range"):
        mi.core.Mesh("t").from_serialized(path, shape_index=3)
    # scene plugin
    d = mi.cornell_box()
    d["blob"] = {"type": "serialized", "filename": path, "shape_index": 1, "to_world": mi.ScalarTransform4f().scale(0.2), "bsdf": {"type": "ref", "id": "red"}}
    scene = mi.load_dict(d)
    blob = [x for x in scene.meshes if x["key"] == "blob"][0]
    assert blob["V"].shape[0] == len(P2) and np.allclose(blob["V"][:, :3], P2 * 0.2, atol=1e-7)


def serialized_v5_blob(V, F, layout, name="mesh", face_normals_flag=False, pidx=None, pcount=0, nidx=None, ncount=0, attrs=(), single=True):
    """Mesh::write_serialized (src/render/mesh.cpp:1091-1148): magic, version 5, then ONE zlib stream: flags, length-prefixed name, four u64 counts, the packed
    vertex (8 x f32) and face (4 x u32) records verbatim, the optional index maps, the attributes"""
    flags = (0x1000 if single else 0) | layout | (0x10 if face_normals_flag else 0)
    payload = struct.pack("<I", flags) + struct.pack("<I", len(name)) + name.encode()
    payload += struct.pack("<QQQQ", len(V), len(F), pcount, ncount) + np.asarray(V, "<f4").tobytes() + np.asarray(F, "<u4").tobytes()
    if pcount: payload += np.asarray(pidx, "<u4").tobytes()
    if ncount: payload += np.asarray(nidx, "<u4").tobytes()
    payload += struct.pack("<I", len(attrs))
    for an, dim, data in attrs:
        payload += struct.pack("<I", len(an)) + an.encode() + struct.pack("<BI", 0, dim) + np.asarray(data, "<f4").tobytes()
    return struct.pack("<HH", 0x041C, 5) + zlib.compress(payload)


def test_serialized_version_5(mi, tmp_path):
    """SerializedMesh::load_v5 (src/shapes/serialized.cpp:393-450): the packed records go in as they are; to_world / flip_normals are applied afterwards
    (PackedMesh::transform_records, mesh_utils.cpp:46-99); index maps and attributes are read past; the stored FaceNormals flag applies when the property is unset"""
    P, F, N, UV = grid_mesh(5)
    V = np.zeros((len(P), 8), np.float32); V[:, :3] = P; V[:, 3:6] = N * 1.5; V[:, 6:8] = UV           # un-normalised normals: records are NOT renormalised without a transform
    F4 = np.zeros((len(F), 4), np.uint32); F4[:, :3] = F; F4[:, 3] = 7
    path = os.path.join(tmp_path, "v5.serialized")
    attrs = [("vertex_color", 3, np.ones((len(P), 3), np.float32)), ("face_id", 1, np.arange(len(F), dtype=np.float32))]
    pidx = np.arange(len(P), dtype=np.uint32)
    blobs = [serialized_v5_blob(V, F4, 1 | 4, attrs=attrs),                                           # normals + texcoords + attributes
             serialized_v5_blob(V, F4, 4, pidx=pidx // 2 * 0 + np.minimum(pidx, len(P) - 1), pcount=len(P)),   # no normals: regenerated; an (identity) position map
             serialized_v5_blob(V, F4, 1 | 4, face_normals_flag=True)]                                 # stored face_normals flag
    write_serialized(path, 4, blobs)          # the sub-mesh directory of a v5 file has the 64-bit offsets of v4 (serialized.cpp:283-290)
    m = mi.core.Mesh("t").from_serialized(path)
    assert np.array_equal(m.V, V) and np.array_equal(m.F, F4) and m.flags == 3
    m = mi.core.Mesh("t").from_serialized(path, shape_index=1)
    assert np.array_equal(m.V[:, :3], P) and np.allclose(m.V[:, 3:6], numpy_normals(P, F), atol=2e-6) and m.flags == 3 and np.array_equal(m.V[:, 6:8], UV)
    m = mi.core.Mesh("t").from_serialized(path, shape_index=2)                                      # property unset: the file's flag decides
    assert m.flags == 2 and not m.V[:, 3:6].any()
    m = mi.core.Mesh("t").from_serialized(path, shape_index=2, face_normals=False)                  # ... an explicit property wins
    assert m.flags == 3 and np.array_equal(m.V[:, 3:6], V[:, 3:6])
    m = mi.core.Mesh("t").from_serialized(path, shape_index=0, face_normals=True)
    assert m.flags == 2 and not m.V[:, 3:6].any()
    # to_world with a negative determinant + flip_normals: positions M p, normals -normalize(M^-T n), winding reversed iff (det < 0) != flip
    T = mi.ScalarTransform4f().translate([0.5, -1, 2]).scale([2, -1, 0.5])
    m = mi.core.Mesh("t").from_serialized(path, to_world=T, flip_normals=True)
    M = np.asarray(T.matrix if not callable(getattr(T, "matrix", None)) else T.matrix(), np.float64).reshape(4, 4)
    assert np.allclose(m.V[:, :3], P @ M[:3, :3].T + M[:3, 3], atol=1e-6)
    n = (N * 1.5) @ np.linalg.inv(M[:3, :3]); n /= np.linalg.norm(n, axis=1, keepdims=True)
    assert np.allclose(m.V[:, 3:6], -n, atol=1e-6) and np.array_equal(m.F[:, :3], F4[:, :3])          # det < 0 and flip: no reversal
    m = mi.core.Mesh("t").from_serialized(path, to_world=T)
    assert np.array_equal(m.F[:, :3], F4[:, [2, 1, 0]]) and np.allclose(m.V[:, 3:6], n, atol=1e-6)
    # a Tangents layout stores the frame as modified Rodrigues parameters (mesh_utils.h:59-117): the product keeps its normal
    def frame_encode(nn, ss):
        tt = np.cross(nn, ss); R = np.stack([ss, tt, nn], axis=1)
        from scipy.spatial.transform import Rotation
        q = Rotation.from_matrix(R).as_quat()            # x, y, z, w
        w = q[3]; return q[:3] / (1 + abs(w)) * (1 if w >= 0 else -1)
    Vt = V.copy(); rng = np.random.default_rng(1)
    nn = rng.normal(size=(len(P), 3)); nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    ss = np.cross(nn, rng.normal(size=(len(P), 3))); ss /= np.linalg.norm(ss, axis=1, keepdims=True)
    Vt[:, 3:6] = np.stack([frame_encode(a, b) for a, b in zip(nn, ss)])
    with open(path, "wb") as f: f.write(serialized_v5_blob(Vt, F4, 1 | 2 | 4))
    m = mi.core.Mesh("t").from_serialized(path)
    assert np.allclose(m.V[:, 3:6], nn, atol=2e-6)
    # errors of load_v5
    def load(blob, **kw):
        with open(path, "wb") as f: f.write(blob)
        return mi.core.Mesh("t").from_serialized(path, **kw)
    with pytest.raises(RuntimeError, match="single precision"): load(serialized_v5_blob(V, F4, 1, single=False))
    with pytest.raises(RuntimeError, match="invalid serialized mesh header"): load(serialized_v5_blob(V, F4, 2 | 4))                   # tangents without normals
    with pytest.raises(RuntimeError, match="invalid serialized mesh header"): load(serialized_v5_blob(V, F4, 1, pidx=pidx, pcount=len(P) + 1))
    with pytest.raises(RuntimeError, match="FaceBSDFs"): load(serialized_v5_blob(V, F4, 1 | 8))
    bad = F4.copy(); bad[0, 1] = 10 ** 6
    with pytest.raises(RuntimeError, match="out of bounds"): load(serialized_v5_blob(V, bad, 1))
    with pytest.raises(RuntimeError, match="end of stream"): load(struct.pack("<HH", 0x041C, 5) + zlib.compress(struct.pack("<II", 0x1001, 1) + b"x" + struct.pack("<QQQQ", 50, 1, 0, 0)))
    # scene plugin
    d = mi.cornell_box()
    with open(path, "wb") as f: f.write(serialized_v5_blob(V, F4, 1 | 4))
    d["blob"] = {"type": "serialized", "filename": path, "to_world": mi.ScalarTransform4f().scale(0.2), "bsdf": {"type": "ref", "id": "red"}}
    scene = mi.load_dict(d)
    blob = [x for x in scene.meshes if x["key"] == "blob"][0]
    assert blob["V"].shape[0] == len(P) and np.allclose(blob["V"][:, :3], P * 0.2, atol=1e-7)


def test_serialized_errors(mi, tmp_path):
    P, F, N, UV = grid_mesh(3)
    path = os.path.join(tmp_path, "e.serialized")
    def load(data, **kw):
        with open(path, "wb") as f: f.write(data)
        return mi.core.Mesh("t").from_serialized(path, **kw)
    good = serialized_blob(4, P, F) + struct.pack("<QI", 0, 1)
    assert load(good).F.shape[0] == len(F)
    with pytest.raises(RuntimeError, match="invalid file format"): load(b"\x1d\x04" + good[2:])
    with pytest.raises(RuntimeError, match="incompatible file version"): load(good[:2] + struct.pack("<H", 2) + good[4:])
    with pytest.raises(RuntimeError, match="incompatible file version"): load(good[:2] + struct.pack("<H", 6) + good[4:])
    with pytest.raises(RuntimeError, match="nonnegative"): load(good, shape_index=-1)
    with pytest.raises(RuntimeError, match="inflate"): load(good[:4] + b"garbage-not-zlib" * 4)
    short = struct.pack("<HH", 0x041C, 4) + zlib.compress(struct.pack("<I", 0x1000) + b"x\0" + struct.pack("<QQ", 100, 1))
    with pytest.raises(RuntimeError, match="end of stream"): load(short)
    bad = serialized_blob(4, P, np.uint32([[0, 1, 99]]))
    with pytest.raises(RuntimeError, match="out of bounds"): load(bad)


def test_obj_scene_plugin(mi, tmp_path):
    path = os.path.join(tmp_path, "cube.obj"); write_obj(path, CUBE_P, cube_faces(), CUBE_T, CUBE_N)
    d = mi.cornell_box()
    d["thing"] = {"type": "obj", "filename": path, "to_world": mi.ScalarTransform4f().translate([0, 0.2, 0]).scale(0.1), "bsdf": {"type": "ref", "id": "green"}}
    scene = mi.load_dict(d)
    thing = [x for x in scene.meshes if x["key"] == "thing"][0]
    assert thing["V"].shape[0] == 24 and thing["F"].shape[0] == 12 and thing["flags"] == 3
    assert np.allclose(np.abs(thing["V"][:, :3] - np.float32([0, 0.2, 0])).max(axis=0), 0.1, atol=1e-6)


def test_reference_normal_weighting_and_regeneration_kats(mi, O):
    """src/render/tests/test_mesh_state.py:30-48 (test01_normal_weighting_scheme) and :51-77 (test02_normal_regeneration_rules, the position-write part): generated
    normals average the unit face normals weighted by the interior angle at each vertex (Thuermer & Wuethrich); a batch that writes positions without normals
    regenerates them.  Both the product (har_mesh_compute_normals, Scene._set_vertex_positions) and the oracle's own restatement (the one its shape-gradient
    derivative differentiates) are held to the reference's numbers"""
    import ctypes as C
    a, b = 1.0, 0.5
    P = np.array([[0, 0, 0], [-a, 1, 0], [a, 1, 0], [-b, 0, 1], [b, 0, 1]], np.float32)
    F = np.array([[0, 1, 2], [0, 3, 4]], np.uint32)
    n0 = np.array([0.0, 0.0, -1.0]); n1 = np.array([0.0, 1.0, 0.0])
    n2 = n0 * (np.pi / 2.0) + n1 * np.arccos(3.0 / 5.0); n2 /= np.linalg.norm(n2)
    want = np.vstack([n2, n0, n0, n1, n1])
    m = mi.load_dict({"type": "mesh", "positions": P, "faces": F, "normals": np.tile([1, 0, 0], (5, 1)).astype(np.float32)})
    m.recompute_vertex_normals()
    assert np.allclose(m.V[:, 3:6], want, atol=5e-4)
    V = np.zeros((5, 8), np.float32); V[:, :3] = P
    F4 = np.zeros((2, 4), np.uint32); F4[:, :3] = F
    L = O.lib(); L.orc_mesh_compute_normals.restype = None
    L.orc_mesh_compute_normals.argtypes = [C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_mesh_compute_normals(5, V.ctypes.data, 2, F4.ctypes.data)
    assert np.allclose(V[:, 3:6], want, atol=5e-4) and np.abs(V[:, 3:6] - m.V[:, 3:6]).max() < 1e-6
    # test02: a unit quad; tilting it through its 'positions' regenerates the normals
    quad = {"type": "mesh", "positions": np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]), "faces": np.uint32([[0, 1, 2], [0, 2, 3]]),
            "normals": np.tile([0, 0, 1], (4, 1)).astype(np.float32)}
    d = mi.cornell_box(); d["quad"] = quad
    scene = mi.load_dict(d)
    i = scene._position_keys()["quad.positions"]
    assert np.allclose(scene.meshes[i]["V"][:, 3:6], [0, 0, 1])
    p = scene.meshes[i]["V"][:, :3].copy(); p[:, 2] = p[:, 0]
    scene._set_vertex_positions(i, p)
    assert np.allclose(scene.meshes[i]["V"][:, 3:6], [-np.sqrt(0.5), 0, np.sqrt(0.5)], atol=1e-6)

"""XML loader == dict loader on randomised scenes (row f3: "scene XML / dict loading compatible with mi.load_dict"; src/core/parser.cpp, src/core/python/parser.cpp).

The scenes of tests/test_gpu_fuzz_parity.py::random_scene -- every plugin of the variant in random combinations -- are first put into their FILE form (meshes as binary PLY,
bitmaps and environment maps as PFM: what an XML scene can name), then written as an XML document by the small writer below and loaded twice: `mi.load_dict(d)` and
`mi.load_string(xml)`.  Both scenes must expose the same parameters with the same values (mi.traverse), lower to the same records (the oracle's render of the host mirrors is
equal BIT FOR BIT, forward image and path / vertex counters) -- a property the parser dropped, a tag it mapped to another type, a default it filled differently, a transform it
composed in another order would all show up here.  No GPU: the product side is the loader and the lowering, the renderer is the oracle on the lowered scene."""
import os
import struct

import numpy as np
import pytest

from tests.test_gpu_fuzz_parity import random_scene, extend_scene


def _write_ply(path, P, F, N=None, UV=None):
    P = np.asarray(P, np.float32).reshape(-1, 3); F = np.asarray(F, np.uint32).reshape(-1, 3)
    cols = [P]; props = ["x", "y", "z"]
    if N is not None:
        cols.append(np.asarray(N, np.float32).reshape(-1, 3)); props += ["nx", "ny", "nz"]
    if UV is not None:
        cols.append(np.asarray(UV, np.float32).reshape(-1, 2)); props += ["u", "v"]
    V = np.ascontiguousarray(np.concatenate(cols, axis=1), "<f4")
    with open(path, "wb") as f:
        header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P.shape[0] + "".join("property float %s\n" % p for p in props)
        header += "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % F.shape[0]
        f.write(header.encode())
        f.write(V.tobytes())
        for tri in F:
            f.write(struct.pack("<B3i", 3, *[int(x) for x in tri]))


def matrix_transforms(mi, d):
    """every transform as what an XML <matrix> can say: its 16 numbers -- the inverse is then computed from them (a chain of translate / rotate / scale calls composes it from
    the parts' analytic inverses, which differs in the last bit: transform.h:364-400 vs Transform(matrix))"""
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out[k] = matrix_transforms(mi, v)
        elif isinstance(v, mi.ScalarTransform4f):
            m = np.asarray(v.matrix, np.float64).reshape(4, 4)
            out[k] = mi.ScalarTransform4f(np.concatenate([m.ravel(), np.linalg.inv(m).T.ravel()]).astype(np.float32))
        else:
            out[k] = v
    return out


def to_file_form(mi, d, folder, prefix=""):
    """arrays -> files, recursively: {'type': 'mesh', positions, faces, ...} -> 'ply'; {'type': 'bitmap', 'data': a} -> 'filename'; envmap 'bitmap': mi.Bitmap -> 'filename'"""
    out = {}
    for k, v in d.items():
        name = prefix + k
        if isinstance(v, dict):
            if v.get("type") == "mesh":
                path = os.path.join(folder, name + ".ply")
                _write_ply(path, v["positions"], v["faces"], v.get("normals"), v.get("texcoords"))
                w = {kk: vv for kk, vv in v.items() if kk not in ("positions", "faces", "normals", "texcoords")}
                w = to_file_form(mi, w, folder, name + "_")
                w.update({"type": "ply", "filename": path, "flip_tex_coords": False})
                if v.get("normals") is None:
                    w["face_normals"] = True            # (the `mesh` plugin without normals is flat-shaded; ply.cpp would compute vertex normals)
                out[k] = w
            elif v.get("type") == "bitmap" and "data" in v:
                path = os.path.join(folder, name + ".pfm")
                mi.write_bitmap(path, np.asarray(v["data"], np.float32))
                w = {kk: vv for kk, vv in v.items() if kk != "data"}
                w["filename"] = path
                out[k] = w
            else:
                out[k] = to_file_form(mi, v, folder, name + "_")
        elif isinstance(v, mi.Bitmap):
            path = os.path.join(folder, name + ".pfm")
            v.write(path)
            out["filename"] = path
        else:
            out[k] = v
    return out


def _num(x):
    return "%.17g" % float(x)


def to_xml(mi, d, kinds):
    """a scene dictionary in file form -> XML text (format of src/core/parser.cpp: object tags named by ObjectType, `name` = the property, `id` at scene level)"""
    lines = ['<scene version="3.0.0">']

    def obj(key, v, depth, top):
        ind = "    " * depth
        t = v["type"]
        ident = ('id="%s"' if top else 'name="%s"') % key
        if t == "rgb":
            lines.append('%s<rgb name="%s" value="%s"/>' % (ind, key, ", ".join(_num(x) for x in v["value"])))
            return
        if t == "ref":
            lines.append('%s<ref name="%s" id="%s"/>' % (ind, key, v["id"]))
            return
        lines.append('%s<%s type="%s" %s>' % (ind, kinds[t], t, ident))
        for k, w in v.items():
            if k == "type":
                continue
            prop(k, w, depth + 1)
        lines.append('%s</%s>' % (ind, kinds[t]))

    def prop(k, w, depth):
        ind = "    " * depth
        if isinstance(w, dict):
            obj(k, w, depth, False)
        elif isinstance(w, bool):
            lines.append('%s<boolean name="%s" value="%s"/>' % (ind, k, "true" if w else "false"))
        elif isinstance(w, (int, np.integer)):
            lines.append('%s<integer name="%s" value="%d"/>' % (ind, k, int(w)))
        elif isinstance(w, (float, np.floating)):
            lines.append('%s<float name="%s" value="%s"/>' % (ind, k, _num(w)))
        elif isinstance(w, str):
            lines.append('%s<string name="%s" value="%s"/>' % (ind, k, w))
        elif isinstance(w, mi.ScalarTransform4f):
            m = np.asarray(w.matrix, np.float64).reshape(4, 4)
            lines.append('%s<transform name="%s"><matrix value="%s"/></transform>' % (ind, k, " ".join(_num(x) for x in m.reshape(-1))))
        elif isinstance(w, mi.ScalarTransform3f):          # a 2-D map (`to_uv`) as the 4 x 4 an XML <transform> holds: Transform::extract reads it back (transform.h:441-456)
            m = np.asarray(w.matrix, np.float64).reshape(3, 3)
            m4 = [m[0, 0], m[0, 1], 0.0, m[0, 2], m[1, 0], m[1, 1], 0.0, m[1, 2], 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]
            lines.append('%s<transform name="%s"><matrix value="%s"/></transform>' % (ind, k, " ".join(_num(x) for x in m4)))
        elif isinstance(w, (list, tuple, np.ndarray)) and len(w) == 3:
            tag = {"position": "point", "direction": "vector"}.get(k, "rgb")          # (eta / k of the conductors: colours)
            if tag == "rgb":
                lines.append('%s<rgb name="%s" value="%s"/>' % (ind, k, ", ".join(_num(x) for x in w)))
            else:
                lines.append('%s<%s name="%s" x="%s" y="%s" z="%s"/>' % (ind, tag, k, _num(w[0]), _num(w[1]), _num(w[2])))
        else:
            raise AssertionError("no XML form for %s = %r" % (k, w))

    for k, v in d.items():
        if k == "type":
            continue
        obj(k, v, 1, True)
    lines.append("</scene>")
    return "\n".join(lines)


def _params_equal(a, b):
    assert list(a.keys()) == list(b.keys()), (sorted(set(a.keys()) ^ set(b.keys())))
    for k in a.keys():
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        assert x.shape == y.shape and np.array_equal(x, y), k


@pytest.mark.parametrize("extended", [False, True])
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("HAR_XML_SEEDS", "20")))))
def test_xml_equals_dict_on_random_scenes(mi, O, seed, extended, tmp_path):
    from mitsuba3_amd.core import _PLUGIN_KINDS
    d, cfg = random_scene(mi, int(os.environ.get("HAR_FUZZ_SEED0", "0")) + seed)
    if extended:                  # all six filters, sample_border, smooth normals in the PLY, `to_uv`, a rotated environment map, a two-child shape group
        d = extend_scene(mi, d, seed)
    d["integrator"] = {"type": "path", "max_depth": cfg["max_depth"], "rr_depth": cfg["rr_depth"], "hide_emitters": cfg["hide"]}
    d = matrix_transforms(mi, d)
    df = to_file_form(mi, d, str(tmp_path))
    xml = to_xml(mi, df, _PLUGIN_KINDS)
    s_dict = mi.load_dict(df)
    s_xml = mi.load_string(xml)
    _params_equal(mi.traverse(s_dict), mi.traverse(s_xml))
    assert s_xml.integrator().max_depth == cfg["max_depth"] and s_xml.integrator().rr_depth == cfg["rr_depth"] and bool(s_xml.integrator().hide_emitters) == cfg["hide"]
    spp = 2
    o1, sensor1 = O.scene_from_product(s_dict); o2, sensor2 = O.scene_from_product(s_xml)
    kw = dict(seed=seed, spp=spp, max_depth=cfg["max_depth"], rr_depth=cfg["rr_depth"], threads=1)      # (one thread: the film is then accumulated in lane order, bit-reproducible)
    a, st1 = o1.render_path(sensor1, **kw); b, st2 = o2.render_path(sensor2, **kw)
    assert np.array_equal(a, b) and st1.paths == st2.paths and st1.vertices == st2.vertices
    # ... and the file form itself is the array form: same picture as the scene with its arrays in memory (PLY / PFM writers and loaders are exact)
    o0, sensor0 = O.scene_from_product(mi.load_dict(d))
    c, st0 = o0.render_path(sensor0, **kw)
    if extended and "smooth" in d:        # a parsed file's normals are normalised once more (Mesh::set_vertex, mesh_utils.cpp:113-115): the last bit of a unit vector may move
        assert np.linalg.norm(a.astype(np.float64) - c.astype(np.float64)) <= 1e-5 * np.linalg.norm(c.astype(np.float64))
    else:
        assert st0.vertices == st1.vertices and np.array_equal(a, c)

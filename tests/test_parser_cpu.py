"""XML scene front end (mitsuba3_amd/parser.py) against the expectations of the reference's src/core/tests/test_parser.py
(tests 05-18: error reporting, 26-28: parameter substitution, 31-37: includes, 40-45: transforms, 59: aliases, 60: defaults, 66: escaped
dollar), and end to end: the Cornell box written as XML lowers to the same flat scene as the dict."""
import os

import numpy as np
import pytest


def T(mi):
    return mi.ScalarTransform4f


def test_transform_tags_and_composition(mi):
    P = mi.parser
    xml = '''<scene version="3.0.0">
        <shape type="sphere">
            <transform name="to_world">
                <translate x="1" y="2" z="3"/>
                <rotate angle="90" x="0" y="1" z="0"/>
                <scale value="2"/>
            </transform>
            <transform name="t2"><translate value="4 5 6"/></transform>
            <transform name="m1"><matrix value="0 0 1 0  0 1 0 0  -1 0 0 0  0 0 0 1"/></transform>
            <transform name="composite">
                <translate x="1" y="0" z="0"/><rotate angle="90" x="0" y="1" z="0"/><scale value="2"/><translate x="0" y="1" z="0"/>
            </transform>
            <transform name="cam"><lookat origin="0 0 5" target="0, 0, 0" up="0 1 0"/></transform>
            <transform name="s3"><scale x="2" z="4"/></transform>
        </shape>
    </scene>'''
    d = P.parse_string(None, xml)
    shape = d["_arg_0"]
    assert shape["type"] == "sphere"
    exp = T(mi)().scale([2, 2, 2]) @ T(mi)().rotate([0, 1, 0], 90) @ T(mi)().translate([1, 2, 3])       # test_parser.py:1115-1121
    assert np.allclose(shape["to_world"].matrix, exp.matrix, atol=1e-6)
    assert np.allclose(shape["t2"].matrix, T(mi)().translate([4, 5, 6]).matrix)
    assert np.allclose(shape["m1"].matrix, T(mi)().rotate([0, 1, 0], 90).matrix, atol=1e-6)               # :1176-1192
    exp = T(mi)().translate([0, 1, 0]) @ T(mi)().scale([2, 2, 2]) @ T(mi)().rotate([0, 1, 0], 90) @ T(mi)().translate([1, 0, 0])   # :1224-1232
    assert np.allclose(shape["composite"].matrix, exp.matrix, atol=1e-6)
    assert np.allclose(shape["cam"].matrix, T(mi)().look_at([0, 0, 5], [0, 0, 0], [0, 1, 0]).matrix, atol=1e-6)
    assert np.allclose(shape["s3"].matrix, T(mi)().scale([2, 1, 4]).matrix)
    inv = shape["to_world"].inverse()
    assert np.allclose((shape["to_world"] @ inv).matrix, np.eye(4), atol=1e-5)
    with pytest.raises(RuntimeError, match="matrix must have 9 or 16 values"):
        P.parse_string(None, '<scene version="3.0.0"><shape type="sphere"><transform name="m"><matrix value="1 2 3 4 5"/></transform></shape></scene>')
    with pytest.raises(RuntimeError, match="transform operations can only occur inside a <transform>"):
        P.parse_string(None, '<scene version="3.0.0"><shape type="sphere"><translate x="1"/></shape></scene>')


def test_properties_and_errors(mi):
    P = mi.parser
    d = P.parse_string(None, '''<scene version="3.0.0">
        <bsdf type="diffuse" id="mat"><rgb name="reflectance" value="0.1, 0.2 0.3"/></bsdf>
        <shape type="ply" id="s">
            <float name="f" value=" 1.5 "/> <integer name="i" value="-3"/> <boolean name="b" value="true"/> <string name="filename" value="a.ply"/>
            <vector name="v" x="1" z="3"/> <point name="p" value="4,5,6"/> <spectrum name="sp" value="0.5"/> <rgb name="grey" value="0.25"/>
            <ref id="mat" name="bsdf"/>
        </shape>
    </scene>''')
    s = d["s"]
    assert s["f"] == 1.5 and s["i"] == -3 and s["b"] is True and s["filename"] == "a.ply" and s["v"] == [1, 0, 3] and s["p"] == [4, 5, 6]
    assert s["sp"] == {"type": "rgb", "value": [0.5] * 3} and s["grey"]["value"] == [0.25] * 3 and s["bsdf"] == {"type": "ref", "id": "mat"}
    assert d["mat"]["reflectance"]["value"] == [0.1, 0.2, 0.3]
    cases = [
        ('<?xml version="1.0"?>', "XML parsing failed: No document element found"),
        ('<?xml version="1.0"?><invalid version="3.0.0"></invalid>', "encountered an unsupported XML element: <invalid>"),
        ('<scene version="3.0.0"><shape type="ply" id="my_id"/><shape type="ply" id="my_id"/></scene>', r'duplicate ID: "my_id" \(previous was at'),
        ('<scene version="3.0.0"><shape type="ply"><integer name="value" value="1"><shape type="ply"/></integer></shape></scene>', "<shape> element cannot occur as child of a property"),
        ('<scene version="3.0.0"><shape type="ply"><integer name="value" value="1"><float name="value" value="1"/></integer></shape></scene>', "<float> element cannot occur as child of a property"),
        ('<scene version="3.0.0"><shape type="ply"><transform name="to_world"><integer name="value" value="10"/></transform></shape></scene>', "unexpected <integer> element inside <transform>"),
        ('<scene version="3.0.0"><integer name="a" value="1"/><integer name="a" value="1"/></scene>', 'Property "a" was specified multiple times'),
        ('<scene version="3.0.0"><boolean name="10" value="a"/></scene>', 'could not parse boolean value "a" -- must be "true" or "false"'),
        ('<scene version="3.0.0"><vector name="10" value="1, 2, 3" x="4"/></scene>', 'Cannot mix "value" and "x"/"y"/"z" attributes'),
        ('<phase type="foo" version="invalid"/>', "Invalid version number"), ('<phase type="foo" version="1.2"/>', "Invalid version number"),
        ('<phase type="foo" version="1.2.4.5"/>', "Invalid version number"),
        ('<scene version="3.0.0"><float name="a" value="1" extra="2"/></scene>', 'unexpected attribute "extra" in <float>'),
        ('<scene version="3.0.0"><float name="a"/></scene>', 'missing attribute "value" in <float>'),
        ('<scene version="3.0.0"><shape id="x"/></scene>', 'missing attribute "type" in <shape>'),
    ]
    for xml, pattern in cases:
        with pytest.raises(RuntimeError, match=pattern):
            P.parse_string(None, xml)
    with pytest.raises(RuntimeError) as ex:                                   # test_parser.py:327-339: line / column of the offending element
        P.parse_string(None, '<scene version="3.0.0">\n        <shape type="sphere">\n            <float name="radius" value="invalid"/>\n        </shape>\n    </scene>')
    assert str(ex.value) == 'Error while loading string (line 3, col 14): could not parse floating point value "invalid"'
    with pytest.raises(RuntimeError, match=r"XML parsing failed: .* \(line 4, col"):
        P.parse_string(None, '<scene version="3.0.0">\n    <shape type="sphere">\n        <float name="radius" value="1.0"\n    </shape>\n</scene>')


def test_parameter_substitution_defaults_aliases(mi):
    P = mi.parser
    assert P.parse_string(None, '<phase type="$mytype" version="3.0.0"/>', mytype="isotropic")["type"] == "isotropic"
    xml = '''<scene version="3.0.0">
        <shape type="$shapetype" id="$shapeid">
            <float name="radius" value="$radius"/>
            <string name="material" value="material_$material_name"/>
        </shape>
    </scene>'''
    d = P.parse_string(None, xml, shapetype="sphere", shapeid="mysphere", radius="2.5", material_name="gold")
    assert d["mysphere"]["type"] == "sphere" and d["mysphere"]["radius"] == 2.5 and d["mysphere"]["material"] == "material_gold"
    d = P.parse_string(None, '<scene version="3.0.0"><shape type="sphere" id="$prefix_$suffix_$prefix"/></scene>', prefix="start", suffix="end")
    assert "start_end_start" in d
    with pytest.raises(RuntimeError, match=r"undefined parameter: \$undefined_param"):
        P.parse_string(None, '<phase type="$undefined_param" version="3.0.0"/>')
    with pytest.raises(RuntimeError) as ex:                                   # test_parser.py:617-632
        P.parse_string(None, '<phase type="isotropic" version="3.0.0"/>', param001="value1", param02="value2", param3="value3")
    assert str(ex.value) == "Found unused parameters:\n  - $param001=value1\n  - $param02=value2\n  - $param3=value3"
    cfg = P.ParserConfig(); cfg.unused_parameters = "debug"
    assert P.parse_string(cfg, '<phase type="isotropic" version="3.0.0"/>', unused_param="value")["type"] == "isotropic"
    xml = '''<scene version="3.0.0">
        <default name="spp" value="16"/> <default name="res" value="64"/>
        <sensor type="perspective"><sampler type="independent"><integer name="sample_count" value="$spp"/></sampler>
            <film type="hdrfilm"><integer name="width" value="$res"/><string name="note" value="costs \\$5"/></film></sensor>
    </scene>'''
    d = P.parse_string(None, xml)
    sensor = d["_arg_0"]
    assert sensor["_arg_0"]["sample_count"] == 16 and sensor["_arg_1"]["width"] == 64 and sensor["_arg_1"]["note"] == "costs $5"
    assert P.parse_string(None, xml, spp=4)["_arg_0"]["_arg_0"]["sample_count"] == 4          # keyword arguments override defaults
    d = P.parse_string(None, '''<scene version="3.0.0"><bsdf type="diffuse" id="a"/><alias id="a" as="b"/>
                                <shape type="cube"><ref id="b" name="bsdf"/></shape></scene>''')
    assert d["_arg_0"]["bsdf"] == {"type": "ref", "id": "a"}
    with pytest.raises(RuntimeError, match='referenced id "zz" not found'):
        P.parse_string(None, '<scene version="3.0.0"><alias id="zz" as="b"/></scene>')
    # pre-2.0 scenes: camelCase property names are upgraded
    d = P.parse_string(None, '<scene version="0.6.0"><integrator type="path"><integer name="maxDepth" value="5"/></integrator></scene>')
    assert d["_arg_0"]["max_depth"] == 5


def test_includes(mi, tmp_path):
    P = mi.parser
    sub = os.path.join(tmp_path, "sub"); os.makedirs(sub)
    with open(os.path.join(sub, "mat.xml"), "w") as f:
        f.write('<bsdf type="diffuse" id="$name" version="3.0.0"><rgb name="reflectance" value="$albedo"/></bsdf>')
    with open(os.path.join(tmp_path, "objects.xml"), "w") as f:
        f.write('<scene version="3.0.0"><include filename="sub/mat.xml"/><shape type="cube" id="c1"><ref id="$name" name="bsdf"/></shape></scene>')
    main = os.path.join(tmp_path, "main.xml")
    with open(main, "w") as f:
        f.write('<scene version="3.0.0"><default name="albedo" value="0.5"/><include filename="objects.xml"/><shape type="cube" id="c2"/></scene>')
    d = P.parse_file(None, main, name="red")
    assert list(k for k in d if k != "type") == ["red", "c1", "c2"]
    assert d["red"]["reflectance"]["value"] == [0.5] * 3 and d["c1"]["bsdf"] == {"type": "ref", "id": "red"}
    with open(main, "w") as f:
        f.write('<scene version="3.0.0"><include filename="nope.xml"/></scene>')
    with pytest.raises(RuntimeError, match='file "nope.xml" not found'):
        P.parse_file(None, main)
    with open(os.path.join(tmp_path, "bad.xml"), "w") as f:
        f.write('<scene version="3.0.0"><float name="x" value="oops"/></scene>')
    with open(main, "w") as f:
        f.write('<scene version="3.0.0">\n<include filename="bad.xml"/></scene>')
    with pytest.raises(RuntimeError, match=r'bad.xml" \(line 1, col \d+\): could not parse floating point value "oops"'):
        P.parse_file(None, main)
    with open(main, "w") as f:
        f.write('<scene version="3.0.0"><include filename="main.xml"/></scene>')
    with pytest.raises(RuntimeError, match="maximum include recursion depth"):
        P.parse_file(None, main)
    with pytest.raises(RuntimeError, match="does not exist"):
        P.parse_file(None, os.path.join(tmp_path, "missing.xml"))


CBOX_XML = '''<scene version="3.0.0">
    <default name="spp" value="64"/> <default name="res" value="256"/>
    <integrator type="path"><integer name="max_depth" value="8"/></integrator>
    <sensor type="perspective">
        <string name="fov_axis" value="smaller"/> <float name="near_clip" value="0.001"/> <float name="far_clip" value="100"/> <float name="fov" value="39.3077"/>
        <transform name="to_world"><lookat origin="0, 0, 3.9" target="0, 0, 0" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sample_count" value="$spp"/></sampler>
        <film type="hdrfilm"><integer name="width" value="$res"/><integer name="height" value="$res"/>
            <rfilter type="gaussian"/><string name="pixel_format" value="rgb"/><string name="component_format" value="float32"/></film>
    </sensor>
    <bsdf type="diffuse" id="white"><rgb name="reflectance" value="0.885809, 0.698859, 0.666422"/></bsdf>
    <bsdf type="diffuse" id="green"><rgb name="reflectance" value="0.105421, 0.37798, 0.076425"/></bsdf>
    <bsdf type="diffuse" id="red"><rgb name="reflectance" value="0.570068, 0.0430135, 0.0443706"/></bsdf>
    <shape type="rectangle" id="light">
        <transform name="to_world"><scale x="0.23" y="0.19" z="0.19"/><rotate x="1" angle="90"/><translate x="0" y="0.99" z="0.01"/></transform>
        <ref id="white"/> <emitter type="area"><rgb name="radiance" value="18.387, 13.9873, 6.75357"/></emitter>
    </shape>
    <shape type="rectangle" id="floor"><transform name="to_world"><rotate x="1" angle="-90"/><translate x="0" y="-1" z="0"/></transform><ref id="white"/></shape>
    <shape type="rectangle" id="ceiling"><transform name="to_world"><rotate x="1" angle="90"/><translate x="0" y="1" z="0"/></transform><ref id="white"/></shape>
    <shape type="rectangle" id="back"><transform name="to_world"><translate x="0" y="0" z="-1"/></transform><ref id="white"/></shape>
    <shape type="rectangle" id="green-wall"><transform name="to_world"><rotate y="1" angle="-90"/><translate x="1" y="0" z="0"/></transform><ref id="green"/></shape>
    <shape type="rectangle" id="red-wall"><transform name="to_world"><rotate y="1" angle="90"/><translate x="-1" y="0" z="0"/></transform><ref id="red"/></shape>
    <shape type="cube" id="small-box"><transform name="to_world"><scale value="0.3"/><rotate y="1" angle="-17"/><translate x="0.335" y="-0.7" z="0.38"/></transform><ref id="white"/></shape>
    <shape type="cube" id="large-box"><transform name="to_world"><scale x="0.3" y="0.61" z="0.3"/><rotate y="1" angle="18.25"/><translate x="-0.33" y="-0.4" z="-0.28"/></transform><ref id="white"/></shape>
</scene>'''


def test_cornell_box_xml_equals_dict(mi, tmp_path):
    """mi.load_string / mi.load_file of the Cornell box (src/python/python/util.py:569-703 written as XML) == mi.load_dict(mi.cornell_box())"""
    a = mi.load_string(CBOX_XML, res=32)
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 32; d["sensor"]["film"]["height"] = 32
    b = mi.load_dict(d)
    assert len(a.meshes) == len(b.meshes) == 8 and a.top_mesh_count == b.top_mesh_count
    for ma, mb in zip(a.meshes, b.meshes):
        assert np.allclose(ma["V"], mb["V"], atol=2e-6) and np.array_equal(ma["F"], mb["F"]) and ma["bsdf"] == mb["bsdf"] and ma["emitter"] == mb["emitter"]
    assert [tuple(x.value) for x in a.bsdf_objs] == [tuple(x.value) for x in b.bsdf_objs]
    ea, eb = a.emitters[0], b.emitters[0]
    assert np.allclose(ea["radiance"], eb["radiance"]) and np.allclose(ea["to_world"], eb["to_world"], atol=2e-6) and abs(ea["inv_area"] - eb["inv_area"]) < 1e-5
    sa, sb = a.sensors()[0].har, b.sensors()[0].har
    assert np.allclose(np.ctypeslib.as_array(sa.sample_to_camera), np.ctypeslib.as_array(sb.sample_to_camera)) and sa.film_width == 32
    assert a.sensors()[0].sampler().sample_count() == 64 and a.integrator().max_depth == 8
    path = os.path.join(tmp_path, "cbox.xml")
    with open(path, "w") as f: f.write(CBOX_XML)
    c = mi.load_file(path, spp=16)
    assert c.sensors()[0].sampler().sample_count() == 16 and len(c.meshes) == 8
    # relative mesh file names resolve next to the scene file
    from tests.test_mesh_io_cpu import grid_mesh, write_ply
    P_, F_, _, _ = grid_mesh(3); write_ply(os.path.join(tmp_path, "g.ply"), "binary_little_endian", P_, F_)
    with open(path, "w") as f:
        f.write(CBOX_XML.replace("</scene>", '<shape type="ply" id="blob"><string name="filename" value="g.ply"/><ref id="red"/></shape></scene>'))
    assert len(mi.load_file(path).meshes) == 9


def test_unreferenced_and_unsupported_properties_are_errors(mi):
    """the reference's loader rejects properties a plugin never queried; hip_ad_rgb does the same for its sensor / film / sampler / integrator
    and refuses reference properties it does not implement instead of ignoring them"""
    import pytest
    base = mi.cornell_box()
    for path, key, value, msg in [(("sensor", "film"), "widht", 64, "Unreferenced"), (("sensor", "sampler"), "samples", 4, "Unreferenced"),
                                  (("integrator",), "maxdepth", 3, "Unreferenced"), (("integrator",), "timeout", 2.0, "not implemented")]:
        d = mi.cornell_box(); node = d
        for p in path:
            node[p] = dict(node[p]); node = node[p]
        node[key] = value
        with pytest.raises(RuntimeError, match=msg):
            mi.load_dict(d)
    d = dict(base); d["sensor"] = dict(d["sensor"], shutter_open=0.0, focus_distance=5.0); d["integrator"] = dict(d["integrator"], block_size=32)
    mi.load_dict(d)                              # known, inert properties pass
    # `compensate` was removed from the reference: marked as queried, warned about, ignored (src/films/hdrfilm.cpp:218-225)
    d = mi.cornell_box(); d["sensor"] = dict(d["sensor"]); d["sensor"]["film"] = dict(d["sensor"]["film"], compensate=True)
    with pytest.warns(UserWarning, match="compensate"):
        mi.load_dict(d)


def test_reference_transform_known_answers(mi):
    """src/core/tests/test_transform.py: test02_inverse (scale / translate matrices, their inverses, point transforms, random affine matrices against
    numpy.linalg.inv), test07_transform_has_scale -- for the product's ScalarTransform4f (har_transform_*; transform.h:132-203,364-400)"""
    T = mi.ScalarTransform4f
    p = np.array([1.0, 2.0, 3.0, 1.0])
    def apply(t, q): return (t.matrix.astype(np.float64) @ q)[:3]
    t = T().scale([1.0, 10.0, 500.0])
    assert np.allclose(t.matrix, np.diag([1, 10, 500, 1])) and np.allclose(apply(t, p), [1, 20, 1500])
    assert np.allclose(t.inverse().matrix, np.diag([1, 1 / 10.0, 1 / 500.0, 1]))
    t = T().translate([1, 0, 1.5])
    want = np.eye(4); want[:3, 3] = [1, 0, 1.5]
    assert np.allclose(t.matrix, want) and np.allclose(apply(t, p), [2, 2, 4.5])
    want[:3, 3] = [-1, 0, -1.5]
    assert np.allclose(t.inverse().matrix, want)
    rng = np.random.default_rng(0)
    for _ in range(10):
        m = rng.random((4, 4)); m[3] = [0, 0, 0, 1]
        data = np.concatenate([m.ravel(), np.linalg.inv(m).T.ravel()]).astype(np.float32)   # (matrix, inverse_transpose) as Transform stores them (transform.h:47-50)
        t = T(data)
        inv = t.inverse().matrix.astype(np.float64)
        assert np.linalg.norm(np.linalg.inv(m) - inv) / np.linalg.norm(inv) < 5e-4
        q = np.append(rng.random(3), 1.0)
        assert np.linalg.norm(apply(t.inverse(), np.append(apply(t, q), 1.0)) - q[:3]) < 5e-3
        c = (t @ t.inverse()).matrix                       # composition keeps the pair consistent
        assert np.allclose(c, np.eye(4), atol=2e-4)
    assert not T().rotate([1, 0, 0], 0.5).has_scale() and not T().rotate([0, 1, 0], 50).has_scale() and not T().rotate([0, 0, 1], 1e3).has_scale()
    assert not T().translate([41, 1e3, 0]).has_scale() and not T().scale([1, 1, 1]).has_scale() and T().scale([1, 1, 1.1]).has_scale()
    assert not T().look_at(origin=[10, -1, 3], target=[1, 1, 2], up=[0, 1, 0]).has_scale() and not T().has_scale()


def test_film_pixel_formats_are_parsed_like_the_reference(mi):
    """hdrfilm.cpp:149-176: the six pixel formats, their FilmFlags::Alpha, and the reference's message for anything else"""
    for pf, colour, alpha in (("rgb", 0, False), ("rgba", 0, True), ("luminance", 1, False), ("luminance_alpha", 1, True), ("xyz", 2, False), ("XYZA", 2, True)):
        f = mi.load_dict({'type': 'hdrfilm', 'width': 4, 'height': 4, 'pixel_format': pf})
        assert (f.colour, f.alpha) == (colour, alpha)
    with pytest.raises(RuntimeError, match='"pixel_format" parameter must either be equal to'):
        mi.load_dict({'type': 'hdrfilm', 'width': 4, 'height': 4, 'pixel_format': 'yuv'})


def test_reference_film_crop_window(mi):
    """src/films/tests/test_hdrfilm.py:36-72 (test02_crops) and Film::set_crop_window (src/render/film.cpp:90-99): accessors, and a crop window that
    leaves the film is an error -- from a dict and from XML"""
    film = mi.load_dict({'type': 'hdrfilm', 'width': 32, 'height': 21, 'crop_width': 11, 'crop_height': 5, 'crop_offset_x': 2, 'crop_offset_y': 3,
                         'pixel_format': 'rgba', 'rfilter': {'type': 'box'}, 'sample_border': True})
    assert film.size() == (32, 21) and film.crop_size() == (11, 5) and film.crop_offset() == (2, 3) and film.sample_border()
    incomplete = '<film version="3.0.0" type="hdrfilm"><integer name="width" value="32"/><integer name="height" value="21"/>' \
                 '<integer name="crop_offset_x" value="30"/><integer name="crop_offset_y" value="20"/>'
    with pytest.raises(RuntimeError, match="Invalid crop window"):
        mi.load_string(incomplete + "</film>")
    film = mi.load_string(incomplete + '<integer name="crop_width" value="2"/><integer name="crop_height" value="1"/></film>')
    assert film.size() == (32, 21) and film.crop_size() == (2, 1) and film.crop_offset() == (30, 20)
    with pytest.raises(RuntimeError, match="Invalid crop window"):
        mi.load_dict({'type': 'hdrfilm', 'width': 8, 'height': 8, 'crop_width': 8, 'crop_offset_x': 1})
    # sample_border with a bordered filter (round 3): the sample grid grows by rfilter->border_size() = ceil(radius - 1/2 - 2 RayEpsilon) on every side
    f = mi.load_dict({'type': 'hdrfilm', 'width': 8, 'height': 8, 'sample_border': True})
    assert f.sample_border() and f.sample_grid() == (12, 12)          # gaussian, stddev 0.5: radius 2 -> border 2
    assert mi.load_dict({'type': 'hdrfilm', 'width': 8, 'height': 6, 'sample_border': True, 'rfilter': {'type': 'tent'}}).sample_grid() == (10, 8)
    assert mi.load_dict({'type': 'hdrfilm', 'width': 8, 'height': 6, 'sample_border': True, 'rfilter': {'type': 'box'}}).sample_grid() == (8, 6)
    assert mi.load_dict({'type': 'hdrfilm', 'width': 8, 'height': 6}).sample_grid() == (8, 6)

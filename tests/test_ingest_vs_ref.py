"""Ingest parity: this project's XML/OBJ readers (product host layer AND oracle) against the reference's
vendored tinyobjloader 2.0.0 / tinyxml2 11.0.0, compiled unchanged into oracle/_ref/libref_ingest.so
(oracle/Makefile).  The call sites replaced are R/Scene/SceneLoader.cpp:26 (LoadObj) and :76-131 (XML).
Skipped when oracle/_ref/ has not been built (it can only be built where /root/reference exists; the
built .so travels to the GPU box)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ASSETS, ROOT
from ingest_cases import polygon_soup
from metalpathtracer_amd import host
from oracle import binding as ob

REF = os.path.join(ROOT, "oracle", "_ref", "libref_ingest.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    R = C.CDLL(REF)
    R.ref_obj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64),
                               C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]
    R.ref_free.argtypes = [C.c_void_p]
    R.ref_xml_open.restype = C.c_void_p
    R.ref_xml_open.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    R.ref_xml_close.argtypes = [C.c_void_p]
    R.ref_xml_name.restype = C.c_char_p
    R.ref_xml_name.argtypes = [C.c_void_p, C.c_int64]
    R.ref_xml_attr.restype = C.c_char_p
    R.ref_xml_attr.argtypes = [C.c_void_p, C.c_int64, C.c_char_p]
    R.ref_xml_float_attr.restype = C.c_float
    R.ref_xml_float_attr.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_float]
    return R


def ref_obj(R, path):
    v, t = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, nt = C.c_uint64(), C.c_uint64()
    assert R.ref_obj_load(path.encode(), C.byref(v), C.byref(nv), C.byref(t), C.byref(nt)) == 0
    verts = np.ctypeslib.as_array(v, (max(nv.value, 1), 3))[:nv.value].copy()
    tris = np.ctypeslib.as_array(t, (max(nt.value, 1), 3))[:nt.value].copy()
    R.ref_free(v)
    R.ref_free(t)
    return verts, tris


def mesh_prims_expected(verts, tris, pos, scale):
    pos = np.asarray(pos, np.float32)
    scale = np.float32(scale)
    return np.stack([pos + scale * verts[tris[:, k]] for k in range(3)], 1)  # SceneLoader.cpp:122-130


def loaders(xml):
    """prims buffers from the product host layer and from the oracle for the same XML."""
    hs = host.Scene()
    st, _ = host.SceneLoader.LoadSceneFromXML(xml, hs)
    assert st == 0
    osn = ob.OracleScene()
    assert osn.load_xml(xml) == 0
    P = osn.prim_count
    op = np.zeros((P, 3, 4), np.float32)
    ob.lib().orc_scene_pack_prims(osn.h, op.ctypes.data_as(C.POINTER(C.c_float)))
    return hs.buffers()[1], op


def test_bunny_bits_match_tinyobj(ref):
    verts, tris = ref_obj(ref, os.path.join(ASSETS, "bunny.obj"))
    assert verts.shape == (2503, 3) and tris.shape == (4968, 3)
    want = mesh_prims_expected(verts, tris, (-25, 0, 0), 10.0)
    hp, op = loaders(os.path.join(ASSETS, "scene.xml"))
    # before buildBVH the primitives are in document order: 3 spheres then the mesh
    for got in (hp, op):
        np.testing.assert_array_equal(got[3:, :, :3].view(np.uint32), want.view(np.uint32))
        assert (got[3:, 0, 3] == 1).all()


TRICKY_OBJ = """# tricky numbers and index forms
v 1 2 3
v -0.5 +.25 1e2
v 1.5E-3 -2.25e+1 .5e1
v 0.123456789012345 3.14159265358979 -0.000001234
v 100000.125 -1e-7 12345678
v 7.0 8.000000001 9.99999999
v -.0 0.1 0.2
v 0.30000001192092896 1.17549435e-38 3.4028234e38
vn 0 0 1
vt 0.5 0.5
f 1 2 3
f 1/1/1 2/1/1 4/1/1
f 2//1 3//1 5//1
f -1 -2 -3
f 1 2 3 4
f 5/1 6/1 7/1 8/1
"""


def test_tricky_obj_matches_tinyobj(ref, tmp_path):
    obj = tmp_path / "tricky.obj"
    obj.write_text(TRICKY_OBJ)
    verts, tris = ref_obj(ref, str(obj))
    xml = tmp_path / "s.xml"
    xml.write_text('<Scene><Mesh file="tricky.obj" position="1,-2,0.5" scale="1.5" albedo="1,1,1" emission="0,0,0"/></Scene>')
    want = mesh_prims_expected(verts, tris, (1, -2, 0.5), 1.5)
    hp, op = loaders(str(xml))
    assert hp.shape[0] == want.shape[0] == op.shape[0]
    for got in (hp, op):
        np.testing.assert_array_equal(got[:, :, :3].view(np.uint32), want.view(np.uint32))


def test_obj_real_parser_matches_tinyobj_on_random_tokens(ref, tmp_path):
    rng = np.random.default_rng(5)
    toks = []
    for _ in range(3000):
        m = rng.integers(0, 10 ** int(rng.integers(1, 12)))
        frac = "".join(str(int(d)) for d in rng.integers(0, 10, int(rng.integers(0, 14))))
        s = ("-" if rng.random() < 0.5 else "") + str(int(m)) + ("." + frac if frac else "")
        if rng.random() < 0.3:
            s += "e%+d" % int(rng.integers(-30, 30))
        toks.append(s)
    obj = tmp_path / "r.obj"
    obj.write_text("".join("v %s %s %s\n" % (toks[i], toks[i + 1], toks[i + 2]) for i in range(0, 3000, 3)))
    verts, _ = ref_obj(ref, str(obj))
    got = np.array([ob.lib().orc_parse_obj_real(t.encode()) for t in toks], np.float32).reshape(-1, 3)
    np.testing.assert_array_equal(got.view(np.uint32), verts.view(np.uint32))
    # product parser through a mesh with identity transform is not bit-transparent (pos + 1*v), so compare via
    # a scene whose triangles index every vertex once
    faces = "".join("f %d %d %d\n" % (i + 1, i + 2, i + 3) for i in range(0, 999, 3))
    obj.write_text(obj.read_text() + faces)
    xml = tmp_path / "r.xml"
    xml.write_text('<Scene><Mesh file="r.obj" position="0,0,0" scale="1" albedo="1,1,1" emission="0,0,0"/></Scene>')
    hp, op = loaders(str(xml))
    verts2, tris2 = ref_obj(ref, str(obj))
    want = mesh_prims_expected(verts2, tris2, (0, 0, 0), 1.0)
    np.testing.assert_array_equal(hp[:, :, :3].view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(op[:, :, :3].view(np.uint32), want.view(np.uint32))


def test_xml_attributes_match_tinyxml2(ref, tmp_path):
    xml = tmp_path / "a.xml"
    xml.write_text("""<?xml version="1.0"?>
<!-- leading comment -->
<Scene>
  <!-- a comment with <Sphere/> inside -->
  <Sphere position="1.5,-2.25,3e1" radius="2.5" albedo="0.1,0.2,0.3" emission="1,0.5,0.25" materialType="-1" emissionPower="4.5"/>
  <Sphere position="0,1" albedo=".5,.5,.5" emission="0,0,0" />
  <Sphere position=' 7 , 8 , 9 ' radius="abc" albedo="1,1,1" emission="0,0,0" materialType="1.5e0"></Sphere>
  <Unknown foo="bar"/>
</Scene>
""")
    n = C.c_int64()
    h = ref.ref_xml_open(str(xml).encode(), C.byref(n))
    assert h and n.value == 4
    hs = host.Scene()
    st, _ = host.SceneLoader.LoadSceneFromXML(str(xml), hs)
    assert st == 0 and hs.getPrimitiveCount() == 3
    _, prims, mats, _ = hs.buffers()
    osn = ob.OracleScene()
    assert osn.load_xml(str(xml)) == 0 and osn.prim_count == 3
    for i in range(3):
        assert ref.ref_xml_name(h, i) == b"Sphere"
        radius = ref.ref_xml_float_attr(h, i, b"radius", 1.0)
        mtype = ref.ref_xml_float_attr(h, i, b"materialType", 0.0)
        power = ref.ref_xml_float_attr(h, i, b"emissionPower", 0.0)
        assert prims[i, 1, 0] == np.float32(radius)
        assert mats[i, 0, 3] == np.float32(mtype) and mats[i, 1, 3] == np.float32(power)
        pos = ref.ref_xml_attr(h, i, b"position").decode()
        import re
        # parseVec3 (SceneLoader.cpp:14-18) = sscanf "%f,%f,%f" on the attribute text tinyxml2 returns
        x = (C.c_float * 3)()
        libc = C.CDLL(None)
        libc.sscanf(pos.encode(), b"%f,%f,%f", C.byref(x, 0), C.byref(x, 4), C.byref(x, 8))
        assert prims[i, 0, :3].tolist() == [x[0], x[1], x[2]]
    ref.ref_xml_close(h)


def test_missing_files_and_roots(ref, tmp_path):
    hs = host.Scene()
    hs.addSphere((0, 0, 0), 1.0)
    st, log = host.SceneLoader.LoadSceneFromXML(str(tmp_path / "nope.xml"), hs)
    assert st == 1 and "Failed to load scene XML" in log and hs.getPrimitiveCount() == 1  # scene untouched
    n = C.c_int64()
    assert not ref.ref_xml_open(str(tmp_path / "nope.xml").encode(), C.byref(n)) and n.value == -1
    bad = tmp_path / "noroot.xml"
    bad.write_text("<Other><Sphere position='0,0,0'/></Other>")
    st, log = host.SceneLoader.LoadSceneFromXML(str(bad), hs)
    assert st == 2 and "No <Scene> root." in log and hs.getPrimitiveCount() == 0  # cleared (SceneLoader.cpp:82-88)
    assert not ref.ref_xml_open(str(bad).encode(), C.byref(n)) and n.value == -2


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_polygon_triangulation_matches_tinyobj(ref, tmp_path, seed):
    obj = tmp_path / "soup.obj"
    obj.write_bytes(polygon_soup(seed).encode())
    verts, tris = ref_obj(ref, str(obj))
    assert len(tris) > 600
    xml = tmp_path / "s.xml"
    xml.write_text('<Scene><Mesh file="soup.obj" position="0,0,0" scale="1" albedo="1,1,1" emission="0,0,0"/></Scene>')
    want = mesh_prims_expected(verts, tris, (0, 0, 0), 1.0)
    hp, op = loaders(str(xml))
    assert hp.shape[0] == want.shape[0] == op.shape[0]
    for got in (hp, op):
        np.testing.assert_array_equal(got[:, :, :3].view(np.uint32), want.view(np.uint32))


FORWARD_OBJ = """v 0 0 0
v 1 0 0
v 1 1 0
f 1 2 3
f 1 2 3 4
f 1 2 3 4 5
f 2 3 9
g later
v 0 1 0
v 0.5 1.5 0
f 1 2 3 4
f 1 2 3 4 5
g
v 9 9 9
"""


def test_forward_references_and_group_flush_match_tinyobj(ref, tmp_path):
    # faces are triangulated when their group ends, against the vertices read so far: a quad naming a vertex that
    # is only defined after the flush disappears, the same quad after it survives; `g` alone does not flush
    obj = tmp_path / "fwd.obj"
    obj.write_text(FORWARD_OBJ)
    verts, tris = ref_obj(ref, str(obj))
    xml = tmp_path / "s.xml"
    xml.write_text('<Scene><Mesh file="fwd.obj" position="0,0,0" scale="1" albedo="1,1,1" emission="0,0,0"/></Scene>')
    want = mesh_prims_expected(verts, tris, (0, 0, 0), 1.0)
    hp, op = loaders(str(xml))
    assert hp.shape[0] == want.shape[0] == op.shape[0] > 0
    for got in (hp, op):
        np.testing.assert_array_equal(got[:, :, :3].view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("face", ["f 0 1 2", "f 1 2 -9", "f 1/-5 2 3", "f 1//-2 2 3", "l 1 0", "f 1 2 x"])
def test_unreadable_obj_matches_tinyobj(ref, tmp_path, face):
    obj = tmp_path / "bad.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n%s\nf 3 2 1\n" % face)
    v, t = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, nt = C.c_uint64(), C.c_uint64()
    assert ref.ref_obj_load(str(obj).encode(), C.byref(v), C.byref(nv), C.byref(t), C.byref(nt)) == 1
    xml = tmp_path / "s.xml"
    xml.write_text('<Scene><Sphere position="0,0,0" albedo="1,1,1" emission="0,0,0"/>'
                   '<Mesh file="bad.obj" position="0,0,0" albedo="1,1,1" emission="0,0,0"/></Scene>')
    hs = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(str(xml), hs)
    assert "Failed to load OBJ" in log and hs.getPrimitiveCount() == 1   # the reference carries on with the sphere
    osn = ob.OracleScene()
    osn.load_xml(str(xml))
    assert osn.prim_count == 1


def test_zero_texcoord_index_is_accepted_like_tinyobj(ref, tmp_path):
    obj = tmp_path / "z.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/0 2/0/0 3//0\n")
    verts, tris = ref_obj(ref, str(obj))
    assert tris.tolist() == [[0, 1, 2]]
    xml = tmp_path / "s.xml"
    xml.write_text('<Scene><Mesh file="z.obj" position="0,0,0" albedo="1,1,1" emission="0,0,0"/></Scene>')
    hp, op = loaders(str(xml))
    assert hp.shape[0] == op.shape[0] == 1


def test_random_xml_formatting_matches_tinyxml2(ref, tmp_path):
    """60 <Sphere> elements with randomised formatting (quotes, whitespace, attribute order, number spellings, missing
    attributes, entities, comments between elements, open/close vs self-closing tags): every value the product and
    the oracle ingest equals what tinyxml2 + the reference's parseVec3 / FloatAttribute produce."""
    rng = np.random.default_rng(17)

    def num():
        v = rng.normal(0, 50)
        form = int(rng.integers(0, 6))
        return ["%g" % v, "%.3f" % v, "%e" % v, "%+.2f" % v, "%d" % int(v), ".5" if v > 0 else "-.25"][form]

    def vec(k):
        sep = [",", ", ", " ,", " , "][int(rng.integers(0, 4))]
        return sep.join(num() for _ in range(k))

    lines, n = ["<?xml version='1.0' encoding='UTF-8'?>", "<Scene>"], 60
    for i in range(n):
        attrs = [("position", vec(int(rng.choice([3, 3, 3, 2, 1])))), ("albedo", vec(3)), ("emission", vec(3))]
        if rng.random() < 0.8: attrs.append(("radius", num()))
        if rng.random() < 0.6: attrs.append(("materialType", num()))
        if rng.random() < 0.6: attrs.append(("emissionPower", num()))
        if rng.random() < 0.2: attrs.append(("note", "a &amp; b &lt;c&gt; &#65;"))
        rng.shuffle(attrs)
        parts = []
        for k, v in attrs:
            q = "'" if rng.random() < 0.4 else '"'
            eq = ["=", " =", "= ", " = "][int(rng.integers(0, 4))]
            parts.append("%s%s%s%s%s" % (k, eq, q, v, q))
        body = ("\n      " if rng.random() < 0.3 else " ").join(parts)
        tag = "<Sphere %s/>" % body if rng.random() < 0.6 else "<Sphere %s></Sphere>" % body
        if rng.random() < 0.3: lines.append("  <!-- comment %d <Sphere radius='9'/> -->" % i)
        lines.append("  " + tag)
    lines.append("</Scene>")
    xml = tmp_path / "fmt.xml"
    xml.write_text("\n".join(lines))
    cnt = C.c_int64()
    h = ref.ref_xml_open(str(xml).encode(), C.byref(cnt))
    assert h and cnt.value == n
    hs = host.Scene()
    st, _ = host.SceneLoader.LoadSceneFromXML(str(xml), hs)
    assert st == 0 and hs.getPrimitiveCount() == n
    _, prims, mats, _ = hs.buffers()
    osn = ob.OracleScene()
    assert osn.load_xml(str(xml)) == 0 and osn.prim_count == n
    op = np.zeros((n, 3, 4), np.float32)
    ob.lib().orc_scene_pack_prims(osn.h, op.ctypes.data_as(C.POINTER(C.c_float)))
    libc = C.CDLL(None)

    def parse_vec3(text):       # SceneLoader.cpp:14-18 on the attribute text tinyxml2 returns (NULL -> zeros)
        x = (C.c_float * 3)()
        if text is not None:
            libc.sscanf(text, b"%f,%f,%f", C.byref(x, 0), C.byref(x, 4), C.byref(x, 8))
        return [x[0], x[1], x[2]]

    for i in range(n):
        assert ref.ref_xml_name(h, i) == b"Sphere"
        radius = ref.ref_xml_float_attr(h, i, b"radius", 1.0)
        mtype = ref.ref_xml_float_attr(h, i, b"materialType", 0.0)
        power = ref.ref_xml_float_attr(h, i, b"emissionPower", 0.0)
        pos = parse_vec3(ref.ref_xml_attr(h, i, b"position"))
        alb = parse_vec3(ref.ref_xml_attr(h, i, b"albedo"))
        emi = parse_vec3(ref.ref_xml_attr(h, i, b"emission"))
        for p in (prims, op):
            assert p[i, 0, :3].tolist() == pos and p[i, 1, 0] == np.float32(radius), i
        assert mats[i, 0, :3].tolist() == alb and mats[i, 1, :3].tolist() == emi, i
        assert mats[i, 0, 3] == np.float32(mtype) and mats[i, 1, 3] == np.float32(power), i
    ref.ref_xml_close(h)


MALFORMED = {
    "unclosed child": '<Scene><Sphere position="1,2,3"></Scene>',
    "no scene close": '<Scene><Sphere position="1,2,3"/>',
    "unquoted attr": '<Scene><Sphere position=1,2,3/></Scene>',
    "truncated tag": '<Scene><Sphere position="1,2,3"/',
    "mismatched": '<Scene><Sphere position="1,2,3"></Mesh></Scene>',
    "dup attr": '<Scene><Sphere position="1,2,3" position="4,5,6"/></Scene>',
    "attr no value": '<Scene><Sphere position/></Scene>',
    "empty": '',
    "blank": '  \n\t ',
}
WELLFORMED = {
    "text content": ('<Scene>hello<Sphere position="1,2,3"/>world</Scene>', 1),
    "nested child": ('<Scene><Group><Sphere position="1,2,3"/></Group><Sphere position="4,5,6"/></Scene>', 1),
    "two roots": ('<Scene><Sphere position="1,2,3"/></Scene><Scene><Sphere position="4,5,6"/></Scene>', 1),
    "bom+decl": ('﻿<?xml version="1.0"?><Scene><Sphere position="1,2,3"/></Scene>', 1),
    "cdata": ('<Scene><![CDATA[ <Sphere position="9,9,9"/> ]]><Sphere position="1,2,3"/></Scene>', 1),
    "end tag space": ('<Scene ><Sphere position="1,2,3" ></Sphere ></Scene >', 1),
}


@pytest.mark.parametrize("name", sorted(MALFORMED))
def test_malformed_xml_is_rejected_like_tinyxml2_and_leaves_the_scene_alone(ref, tmp_path, name):
    """tinyxml2's LoadFile fails on these; the reference then returns before scene->clear() (SceneLoader.cpp:76-80)."""
    p = tmp_path / "bad.xml"
    p.write_text(MALFORMED[name], encoding="utf-8")
    n = C.c_int64()
    assert not ref.ref_xml_open(str(p).encode(), C.byref(n)) and n.value == -1
    hs = host.Scene()
    hs.addSphere((0, 0, 0), 1.0)
    st, log = host.SceneLoader.LoadSceneFromXML(str(p), hs)
    assert st in (1, 3) and "Failed to load scene XML" in log and hs.getPrimitiveCount() == 1
    osn = ob.OracleScene()
    osn.add_sphere((0, 0, 0), 1.0) if hasattr(osn, "add_sphere") else None
    assert osn.load_xml(str(p)) != 0


def test_comment_only_document_has_no_scene_root(ref, tmp_path):
    p = tmp_path / "c.xml"
    p.write_text("<!-- nothing here -->")
    n = C.c_int64()
    assert not ref.ref_xml_open(str(p).encode(), C.byref(n)) and n.value == -2      # LoadFile ok, no <Scene>
    hs = host.Scene()
    hs.addSphere((0, 0, 0), 1.0)
    st, log = host.SceneLoader.LoadSceneFromXML(str(p), hs)
    assert st == 2 and "No <Scene> root." in log and hs.getPrimitiveCount() == 0     # cleared, SceneLoader.cpp:82-88


@pytest.mark.parametrize("name", sorted(WELLFORMED))
def test_odd_but_wellformed_xml_is_accepted_like_tinyxml2(ref, tmp_path, name):
    text, spheres = WELLFORMED[name]
    p = tmp_path / "ok.xml"
    p.write_text(text, encoding="utf-8")
    n = C.c_int64()
    h = ref.ref_xml_open(str(p).encode(), C.byref(n))
    assert h
    direct = sum(1 for i in range(n.value) if ref.ref_xml_name(h, i) == b"Sphere")
    ref.ref_xml_close(h)
    assert direct == spheres
    hs = host.Scene()
    st, _ = host.SceneLoader.LoadSceneFromXML(str(p), hs)
    assert st == 0 and hs.getPrimitiveCount() == spheres
    osn = ob.OracleScene()
    assert osn.load_xml(str(p)) == 0 and osn.prim_count == spheres

"""Generates tests/golden/polygon_soup.obj and polygon_soup_expected.npz with the REAL tinyobjloader 2.0.0 the
reference vendors, compiled unchanged into oracle/_ref/libref_ingest.so (oracle/Makefile; needs /root/reference).

The .obj is a generated input (tests/ingest_cases.py, seed 11); the .npz holds what tinyobj::LoadObj returns for it
with the reference's call (R/Scene/SceneLoader.cpp:26) after the reference's filter (:40-68): vertex floats and the
triangle index list.  tests/test_golden.py::test_polygon_soup_ingest checks the product loader and the oracle
against it wherever oracle/_ref is not available.

Run:  python tests/golden/make_ingest_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ingest_cases import polygon_soup  # noqa: E402


def main():
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_ingest.so"))
    R.ref_obj_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint64),
                               C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint64)]
    R.ref_free.argtypes = [C.c_void_p]
    path = os.path.join(HERE, "polygon_soup.obj")
    with open(path, "wb") as f:
        f.write(polygon_soup(11, n_faces=250).encode())
    v, t = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)()
    nv, nt = C.c_uint64(), C.c_uint64()
    assert R.ref_obj_load(path.encode(), C.byref(v), C.byref(nv), C.byref(t), C.byref(nt)) == 0
    verts = np.ctypeslib.as_array(v, (nv.value, 3)).copy()
    tris = np.ctypeslib.as_array(t, (nt.value, 3)).copy()
    R.ref_free(v)
    R.ref_free(t)
    np.savez_compressed(os.path.join(HERE, "polygon_soup_expected.npz"), verts=verts, tris=tris)
    print("polygon_soup:", verts.shape, tris.shape)


if __name__ == "__main__":
    main()

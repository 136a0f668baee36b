// ref_ingest_shim.cpp — thin extern "C" surface over the reference's VENDORED THIRD-PARTY parsers
// (tinyxml2 11.0.0, tinyobjloader 2.0.0), which are compiled unchanged from /root/reference by
// oracle/Makefile into oracle/_ref/libref_ingest.so.  TEST INFRASTRUCTURE ONLY: it lets
// tests/test_ingest_vs_ref.py check that this project's own XML/OBJ ingest (product and oracle)
// yields bit-identical floats and the same triangle list as the libraries the reference calls
// (call sites: R/Scene/SceneLoader.cpp:26 tinyobj::LoadObj, :76-131 tinyxml2).  Only this file is
// ours; it calls the libraries' public API the way SceneLoader.cpp does and contains no reference
// first-party code.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tiny_obj_loader.h"
#include "tinyxml2.h"

extern "C" {

// tinyobj::LoadObj with the defaults SceneLoader.cpp:26 uses (triangulate = true, no mtl dir);
// keeps 3-vertex faces whose indices are in range, as SceneLoader.cpp:48-68 does.
// Returns 0 on success; caller frees with ref_free.
int ref_obj_load(const char* path, float** verts_out, uint64_t* nverts, uint32_t** tris_out, uint64_t* ntris) {
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string warn, err;
    if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, path)) return 1;
    size_t nv = attrib.vertices.size() / 3;
    float* v = (float*)malloc(sizeof(float) * 3 * (nv ? nv : 1));
    memcpy(v, attrib.vertices.data(), sizeof(float) * 3 * nv);
    std::vector<uint32_t> t;
    for (const auto& shape : shapes) {
        size_t off = 0;
        for (size_t f = 0; f < shape.mesh.num_face_vertices.size(); ++f) {
            size_t fv = shape.mesh.num_face_vertices[f];
            if (fv == 3) {
                uint32_t a = shape.mesh.indices[off + 0].vertex_index, b = shape.mesh.indices[off + 1].vertex_index,
                         c = shape.mesh.indices[off + 2].vertex_index;
                if (a < nv && b < nv && c < nv) {
                    t.push_back(a);
                    t.push_back(b);
                    t.push_back(c);
                }
            }
            off += fv;
        }
    }
    uint32_t* to = (uint32_t*)malloc(sizeof(uint32_t) * (t.size() ? t.size() : 1));
    memcpy(to, t.data(), sizeof(uint32_t) * t.size());
    *verts_out = v;
    *nverts = nv;
    *tris_out = to;
    *ntris = t.size() / 3;
    return 0;
}
void ref_free(void* p) { free(p); }

// tinyxml2: number of child elements of <Scene>, or -1 (load failure) / -2 (no <Scene> root).
struct RefXml {
    tinyxml2::XMLDocument doc;
    std::vector<const tinyxml2::XMLElement*> kids;
};
void* ref_xml_open(const char* path, int64_t* nkids) {
    RefXml* x = new RefXml();
    if (x->doc.LoadFile(path) != tinyxml2::XML_SUCCESS) {
        *nkids = -1;
        delete x;
        return nullptr;
    }
    auto* root = x->doc.FirstChildElement("Scene");
    if (!root) {
        *nkids = -2;
        delete x;
        return nullptr;
    }
    for (auto* e = root->FirstChildElement(); e; e = e->NextSiblingElement()) x->kids.push_back(e);
    *nkids = (int64_t)x->kids.size();
    return x;
}
void ref_xml_close(void* h) { delete (RefXml*)h; }
const char* ref_xml_name(void* h, int64_t i) { return ((RefXml*)h)->kids[i]->Name(); }
const char* ref_xml_attr(void* h, int64_t i, const char* key) { return ((RefXml*)h)->kids[i]->Attribute(key); }
float ref_xml_float_attr(void* h, int64_t i, const char* key, float dflt) {
    return ((RefXml*)h)->kids[i]->FloatAttribute(key, dflt);
}

}  // extern "C"

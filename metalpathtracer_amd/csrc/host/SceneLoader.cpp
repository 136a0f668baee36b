// SceneLoader.cpp — XML + OBJ ingest for the host scene layer.
//
// Follows the reference's loader semantics (R/Scene/SceneLoader.cpp:14-133):
//   <Sphere position= radius=(1) albedo= emission= materialType=(0) emissionPower=(0)/>
//   <Mesh file= position= scale=(1) albedo= emission= materialType= emissionPower=/>   one material per mesh,
//   one Triangle primitive per face, vertex = position + scale * v.
// The reference parses with tinyxml2 11.0.0 and tinyobjloader 2.0.0 (vendored there, third-party).  They are not
// vendored here; the subset the path exercises is implemented below, and the NUMBER PARSING follows those
// libraries exactly because it fixes the bits of every vertex:
//   * vec3 attributes: sscanf("%f,%f,%f")            (SceneLoader.cpp:14-18)
//   * float attributes: sscanf("%f")                 (tinyxml2 XMLUtil::ToFloat)
//   * OBJ reals: tinyobjloader's tryParseDouble (decimal digits accumulated in a double, fraction digits added
//     as d * 10^-k, optional exponent applied as ldexp(m * 5^e, e)), then narrowed to float
//     (R/tiny_obj_loader.h:897-1038).  tests/test_ingest_vs_ref.py checks bit equality against the real library.
#include "SceneLoader.h"

#include "Transform.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <utility>
#include <vector>

namespace MetalCppPathTracer {

namespace {

using mpt::float3;

// ------------------------------------------------------------------------------------------ numbers
// objDouble restates tryParseDouble of tinyobjloader 2.0.0 (the OBJ reader the reference vendors and calls,
// R/tiny_obj_loader.h:897-1028) so that every vertex gets the same bits as in the reference.  tinyobjloader is
// Copyright (c) 2012-Present Syoyo Fujita and many contributors, MIT License: "Permission is hereby granted, free of
// charge, to any person obtaining a copy of this software and associated documentation files (the "Software"), to deal
// in the Software without restriction ... The above copyright notice and this permission notice shall be included in
// all copies or substantial portions of the Software."  Full text: THIRD_PARTY_NOTICES.md.
bool objDouble(const char* s, const char* end, double* out) {
    if (s >= end) return false;
    const char* c = s;
    bool negative = false;
    if (*c == '+' || *c == '-') {
        negative = (*c == '-');
        ++c;
    }
    bool dotFirst = false;
    if (c != end && *c == '.') {
        dotFirst = true;
    } else if (c == end || !std::isdigit(static_cast<unsigned char>(*c))) {
        if (c == s) return false;           // neither sign, digit nor dot
        if (c == end) return false;         // bare sign
        if (*c != '.') return false;
    }
    double mant = 0.0;
    if (!dotFirst) {
        int digits = 0;
        while (c != end && std::isdigit(static_cast<unsigned char>(*c))) {
            mant *= 10;
            mant += static_cast<int>(*c - '0');
            ++c;
            ++digits;
        }
        if (digits == 0) return false;
    }
    int exponent = 0;
    bool haveExp = false;
    if (c != end && *c == '.') {
        ++c;
        int k = 1;
        static const double small[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
        while (c != end && std::isdigit(static_cast<unsigned char>(*c))) {
            mant += static_cast<int>(*c - '0') * (k < 8 ? small[k] : std::pow(10.0, -k));
            ++k;
            ++c;
        }
    }
    if (c != end && (*c == 'e' || *c == 'E')) {
        ++c;
        bool expNeg = false;
        if (c != end && (*c == '+' || *c == '-')) {
            expNeg = (*c == '-');
            ++c;
        } else if (c == end || !std::isdigit(static_cast<unsigned char>(*c))) {
            return false;
        }
        int digits = 0;
        while (c != end && std::isdigit(static_cast<unsigned char>(*c))) {
            if (exponent > 2147483647 / 10) return false;
            exponent = exponent * 10 + static_cast<int>(*c - '0');
            ++c;
            ++digits;
        }
        if (digits == 0) return false;
        if (expNeg) exponent = -exponent;
        haveExp = exponent != 0;
    }
    double v = haveExp ? std::ldexp(mant * std::pow(5.0, exponent), exponent) : mant;
    *out = (negative ? -1 : 1) * v;
    return true;
}

float objReal(const char*& cur) {
    cur += std::strspn(cur, " \t");
    const char* end = cur + std::strcspn(cur, " \t\r\n");
    double v = 0.0;
    objDouble(cur, end, &v);
    cur = end;
    return static_cast<float>(v);
}

float3 vec3Attr(const char* text) {
    float x = 0, y = 0, z = 0;
    if (text) std::sscanf(text, "%f,%f,%f", &x, &y, &z);
    return float3(x, y, z);
}

float floatAttr(const char* text, float fallback) {
    float v = fallback;
    if (text && std::sscanf(text, "%f", &v) == 1) return v;
    return fallback;
}

// ------------------------------------------------------------------------------------------ OBJ
struct Mesh {
    std::vector<float3> vertices;
    std::vector<mpt::uint3> triangles;
};

// Reads the vertex and face statements of a Wavefront OBJ the way the reference's call
// tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, path) does (R/Scene/SceneLoader.cpp:26,
// tinyobjloader 2.0.0 defaults: triangulate = true), then filters like SceneLoader.cpp:40-68:
//   - statements end at \n, \r\n or a lone \r; trailing blanks are dropped; `#` ends a face statement;
//   - a corner is `i`, `i/j`, `i//k` or `i/j/k` read with atoi; vertex index 0 or a relative index that reaches
//     before the first element makes the whole file unreadable (the library returns false);
//   - faces wait in their group until `g <name>`, `o <name>` or the end of the file and are triangulated against
//     the vertices read by then: triangles as they are; quads along the shorter diagonal (0-2 only if strictly
//     shorter than 1-3, a quad with an unread vertex is dropped); larger polygons by the library's ear clipping
//     in the plane of two coordinate axes chosen from the first non-degenerate corner;
//   - triangles naming a vertex beyond the final count are reported and skipped.
// Material libraries are not read (the reference never uses them; with one present tinyobj would also flush a
// group at a material change, which only matters for faces that name vertices defined later in the file).
class ObjReader {
public:
    ObjReader(Mesh& mesh, std::string& log) : mesh_(mesh), log_(log) {}

    bool read(const std::string& path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return unreadable(path);
        std::stringstream whole;
        whole << in.rdbuf();
        const std::string text = whole.str();
        size_t at = 0;
        std::string st;
        while (at < text.size()) {
            size_t stop = text.find_first_of("\r\n", at);
            if (stop == std::string::npos) stop = text.size();
            st.assign(text, at, stop - at);
            at = stop + 1;
            if (stop < text.size() && text[stop] == '\r' && at < text.size() && text[at] == '\n') ++at;
            const size_t last = st.find_last_not_of(" \t");
            st.erase(last == std::string::npos ? 0 : last + 1);
            if (!statement(st.c_str())) return unreadable(path);
        }
        flushGroup();
        const size_t nv = coords_.size() / 3;
        mesh_.vertices.reserve(nv);
        for (size_t i = 0; i < nv; ++i) mesh_.vertices.emplace_back(coords_[3 * i], coords_[3 * i + 1], coords_[3 * i + 2]);
        for (size_t i = 0; i + 2 < corners_.size(); i += 3) {
            if (corners_[i] >= nv || corners_[i + 1] >= nv || corners_[i + 2] >= nv) {
                log_ += "Invalid triangle indices\n";
                continue;
            }
            mpt::uint3 t;
            t.x = corners_[i];
            t.y = corners_[i + 1];
            t.z = corners_[i + 2];
            mesh_.triangles.push_back(t);
        }
        char msg[128];
        std::snprintf(msg, sizeof msg, "Loaded OBJ: %zu vertices, %zu triangles\n", mesh_.vertices.size(),
                      mesh_.triangles.size());
        log_ += msg;
        return true;
    }

private:
    static bool blank(char c) { return c == ' ' || c == '\t'; }

    bool unreadable(const std::string& path) {
        log_ += "Failed to load OBJ: " + path + "\n";
        return false;
    }

    // zero-based index from an OBJ index; `zeroOk` is true for the normal/texcoord fields
    static bool resolve(int raw, size_t count, bool zeroOk, long& out) {
        if (raw > 0) {
            out = raw - 1;
            return true;
        }
        if (raw == 0) {
            out = -1;
            return zeroOk;
        }
        out = static_cast<long>(count) + raw;
        return out >= 0;
    }

    bool corner(const char*& p, long& vertex) {
        long ignored = 0;
        if (!resolve(std::atoi(p), coords_.size() / 3, false, vertex)) return false;
        p += std::strcspn(p, "/ \t\r");
        if (*p != '/') return true;
        ++p;
        if (*p == '/') {
            ++p;
            if (!resolve(std::atoi(p), normals_, true, ignored)) return false;
            p += std::strcspn(p, "/ \t\r");
            return true;
        }
        if (!resolve(std::atoi(p), texcoords_, true, ignored)) return false;
        p += std::strcspn(p, "/ \t\r");
        if (*p != '/') return true;
        ++p;
        if (!resolve(std::atoi(p), normals_, true, ignored)) return false;
        p += std::strcspn(p, "/ \t\r");
        return true;
    }

    bool statement(const char* p) {
        p += std::strspn(p, " \t");
        const char k = p[0];
        if (k == 'v' && blank(p[1])) {
            p += 1;
            const float x = objReal(p), y = objReal(p), z = objReal(p);
            coords_.push_back(x);
            coords_.push_back(y);
            coords_.push_back(z);
        } else if (k == 'v' && p[1] == 'n' && blank(p[2])) {
            ++normals_;
        } else if (k == 'v' && p[1] == 't' && blank(p[2])) {
            ++texcoords_;
        } else if ((k == 'f' || k == 'l' || k == 'p') && blank(p[1])) {
            p += 2;
            p += std::strspn(p, " \t");
            std::vector<long> poly;
            while (*p != '\0' && *p != '\r' && *p != '\n' && *p != '#') {
                long v = -1;
                if (!corner(p, v)) return false;
                poly.push_back(v);
                p += std::strspn(p, " \t\r");
            }
            if (k == 'f') pending_.push_back(std::move(poly));
        } else if ((k == 'g' || k == 'o') && blank(p[1])) {
            flushGroup();
        }
        return true;
    }

    void flushGroup() {
        for (const std::vector<long>& poly : pending_) {
            if (poly.size() == 3) tri(poly[0], poly[1], poly[2]);
            else if (poly.size() == 4) quad(poly);
            else if (poly.size() > 4) clipEars(poly);
        }
        pending_.clear();
    }

    void tri(long a, long b, long c) {
        corners_.push_back(static_cast<uint32_t>(a));
        corners_.push_back(static_cast<uint32_t>(b));
        corners_.push_back(static_cast<uint32_t>(c));
    }

    bool known(long i) const { return 3 * static_cast<size_t>(i) + 2 < coords_.size(); }
    const float* at(long i) const { return &coords_[3 * static_cast<size_t>(i)]; }

    void quad(const std::vector<long>& q) {
        for (long i : q)
            if (!known(i)) return;
        auto span2 = [&](long a, long b) {
            const float dx = at(b)[0] - at(a)[0], dy = at(b)[1] - at(a)[1], dz = at(b)[2] - at(a)[2];
            return dx * dx + dy * dy + dz * dz;
        };
        if (span2(q[0], q[2]) < span2(q[1], q[3])) {
            tri(q[0], q[1], q[2]);
            tri(q[0], q[2], q[3]);
        } else {
            tri(q[0], q[1], q[3]);
            tri(q[1], q[2], q[3]);
        }
    }

    // crossing-number test of (tx, ty) against the triangle (x[], y[])
    static bool insideTriangle(const float x[3], const float y[3], float tx, float ty) {
        bool odd = false;
        for (int i = 0, j = 2; i < 3; j = i++)
            if ((y[i] > ty) != (y[j] > ty) && tx < (x[j] - x[i]) * (ty - y[i]) / (y[j] - y[i]) + x[i]) odd = !odd;
        return odd;
    }

    void clipEars(const std::vector<long>& poly) {
        const size_t n = poly.size(), nc = coords_.size();
        size_t ua = 1, ub = 2;  // the two coordinate axes the polygon is flattened onto
        for (size_t k = 0; k < n; ++k) {
            const long i0 = poly[k], i1 = poly[(k + 1) % n], i2 = poly[(k + 2) % n];
            if (!known(i0) || !known(i1) || !known(i2)) continue;
            const float ex = at(i1)[0] - at(i0)[0], ey = at(i1)[1] - at(i0)[1], ez = at(i1)[2] - at(i0)[2];
            const float fx = at(i2)[0] - at(i1)[0], fy = at(i2)[1] - at(i1)[1], fz = at(i2)[2] - at(i1)[2];
            const float nx = std::fabs(ey * fz - ez * fy);
            const float ny = std::fabs(ez * fx - ex * fz);
            const float nz = std::fabs(ex * fy - ey * fx);
            const float tiny = std::numeric_limits<float>::epsilon();
            if (nx > tiny || ny > tiny || nz > tiny) {
                if (!(nx > ny && nx > nz)) {
                    ua = 0;
                    if (nz > nx && nz > ny) ub = 1;
                }
                break;
            }
        }
        std::vector<long> ring = poly;
        size_t probe = 0, tries = n, lastSize = n;
        while (ring.size() > 3 && tries > 0) {
            const size_t m = ring.size();
            if (probe >= m) probe -= m;
            if (lastSize != m) {
                lastSize = m;
                tries = m;
            } else {
                --tries;
            }
            long id[3];
            float x[3], y[3];
            for (size_t k = 0; k < 3; ++k) {
                id[k] = ring[(probe + k) % m];
                const size_t base = 3 * static_cast<size_t>(id[k]);
                const bool have = base + ua < nc && base + ub < nc;
                x[k] = have ? coords_[base + ua] : 0.0f;
                y[k] = have ? coords_[base + ub] : 0.0f;
            }
            const float turn = (x[1] - x[0]) * (y[2] - y[1]) - (y[1] - y[0]) * (x[2] - x[1]);
            const float sign = (x[0] * y[1] - y[0] * x[1]) * 0.5f;  // the library's orientation term: two vertices only
            bool ear = !(turn * sign < 0.0f);
            for (size_t o = 3; ear && o < m; ++o) {
                const size_t base = 3 * static_cast<size_t>(ring[(probe + o) % m]);
                if (base + ua >= nc || base + ub >= nc) continue;
                if (insideTriangle(x, y, coords_[base + ua], coords_[base + ub])) ear = false;
            }
            if (!ear) {
                ++probe;
                continue;
            }
            tri(id[0], id[1], id[2]);
            ring.erase(ring.begin() + static_cast<long>((probe + 1) % m));
        }
        if (ring.size() == 3) tri(ring[0], ring[1], ring[2]);
    }

    Mesh& mesh_;
    std::string& log_;
    std::vector<float> coords_;
    size_t normals_ = 0, texcoords_ = 0;
    std::vector<std::vector<long>> pending_;
    std::vector<uint32_t> corners_;
};

bool readObj(const std::string& path, Mesh& mesh, std::string& log) { return ObjReader(mesh, log).read(path); }

// ------------------------------------------------------------------------------------------ XML subset
struct Element {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attributes;
    const char* get(const char* key) const {
        for (const auto& kv : attributes)
            if (kv.first == key) return kv.second.c_str();
        return nullptr;
    }
};

class XmlScanner {
public:
    explicit XmlScanner(const std::string& t) : text_(t) {}

    // Collects the element children of the first top-level <Scene>.  Returns Ok / NoSceneRoot / XmlMalformed.
    // Well-formedness is checked the way tinyxml2's LoadFile does for the constructs a scene file can contain — end
    // tags must match the open element, every element must be closed, attribute names are unique per element, the
    // document holds at least one node (element, comment or declaration) — because the reference leaves the scene untouched when LoadFile fails
    // (R/Scene/SceneLoader.cpp:76-80); tests/test_ingest_vs_ref.py compares accept / reject with the real library.
    SceneLoader::Status children(std::vector<Element>& out) {
        int depth = 0;
        bool inside = false, found = false;
        std::vector<std::string> open;
        size_t elements = 0;
        while (pos_ < text_.size()) {
            if (text_[pos_] != '<') {
                ++pos_;
                continue;
            }
            ++elements;  // any markup counts: only a document without a single node is "empty" for tinyxml2
            if (startsWith("<!--")) {
                if (!skipPast("-->")) return SceneLoader::XmlMalformed;
            } else if (startsWith("<?")) {
                if (!skipPast("?>")) return SceneLoader::XmlMalformed;
            } else if (startsWith("<!")) {
                if (!skipPast(">")) return SceneLoader::XmlMalformed;
            } else if (startsWith("</")) {
                size_t close = text_.find('>', pos_);
                if (close == std::string::npos) return SceneLoader::XmlMalformed;
                std::string name = text_.substr(pos_ + 2, close - pos_ - 2);
                while (!name.empty() && std::isspace(static_cast<unsigned char>(name.back()))) name.pop_back();
                if (open.empty() || open.back() != name) return SceneLoader::XmlMalformed;  // mismatched end tag
                open.pop_back();
                --depth;
                if (inside && depth == 0) inside = false;
                pos_ = close + 1;
            } else {
                Element e;
                bool selfClosing = false;
                if (!openTag(e, selfClosing)) return SceneLoader::XmlMalformed;
                if (e.name.empty()) return SceneLoader::XmlMalformed;
                for (size_t a = 0; a < e.attributes.size(); ++a)
                    for (size_t b = a + 1; b < e.attributes.size(); ++b)
                        if (e.attributes[a].first == e.attributes[b].first) return SceneLoader::XmlMalformed;
                if (!selfClosing) open.push_back(e.name);
                if (depth == 0 && !found && e.name == "Scene") {
                    found = true;
                    inside = !selfClosing;
                } else if (inside && depth == 1) {
                    out.push_back(std::move(e));
                }
                if (!selfClosing) ++depth;
            }
        }
        if (!open.empty() || elements == 0) return SceneLoader::XmlMalformed;  // unclosed element / empty document
        return found ? SceneLoader::Ok : SceneLoader::NoSceneRoot;
    }

private:
    bool startsWith(const char* lit) const { return text_.compare(pos_, std::strlen(lit), lit) == 0; }
    bool skipPast(const char* lit) {
        size_t at = text_.find(lit, pos_);
        if (at == std::string::npos) return false;
        pos_ = at + std::strlen(lit);
        return true;
    }
    void skipSpace() {
        while (pos_ < text_.size() && std::isspace(static_cast<unsigned char>(text_[pos_]))) ++pos_;
    }
    static std::string unescape(const std::string& raw) {
        static const struct { const char* from; char to; } table[] = {
            {"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
        std::string out;
        for (size_t i = 0; i < raw.size();) {
            bool hit = false;
            if (raw[i] == '&')
                for (const auto& t : table) {
                    size_t len = std::strlen(t.from);
                    if (raw.compare(i, len, t.from) == 0) {
                        out += t.to;
                        i += len;
                        hit = true;
                        break;
                    }
                }
            if (!hit) out += raw[i++];
        }
        return out;
    }
    bool openTag(Element& e, bool& selfClosing) {
        ++pos_;  // '<'
        size_t b = pos_;
        while (pos_ < text_.size() && !std::isspace(static_cast<unsigned char>(text_[pos_])) && text_[pos_] != '>' &&
               text_[pos_] != '/')
            ++pos_;
        e.name = text_.substr(b, pos_ - b);
        for (;;) {
            skipSpace();
            if (pos_ >= text_.size()) return false;
            if (text_[pos_] == '/') {
                selfClosing = true;
                ++pos_;
                continue;
            }
            if (text_[pos_] == '>') {
                ++pos_;
                return true;
            }
            size_t kb = pos_;
            while (pos_ < text_.size() && text_[pos_] != '=' && !std::isspace(static_cast<unsigned char>(text_[pos_])))
                ++pos_;
            std::string key = text_.substr(kb, pos_ - kb);
            skipSpace();
            if (pos_ >= text_.size() || text_[pos_] != '=') return false;
            ++pos_;
            skipSpace();
            if (pos_ >= text_.size() || (text_[pos_] != '"' && text_[pos_] != '\'')) return false;
            const char quote = text_[pos_++];
            size_t vb = pos_;
            size_t ve = text_.find(quote, vb);
            if (ve == std::string::npos) return false;
            e.attributes.emplace_back(std::move(key), unescape(text_.substr(vb, ve - vb)));
            pos_ = ve + 1;
        }
    }

    const std::string& text_;
    size_t pos_ = 0;
};

bool readable(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return static_cast<bool>(f);
}
std::string baseName(const std::string& p) {
    size_t s = p.find_last_of("/\\");
    return s == std::string::npos ? p : p.substr(s + 1);
}
std::string dirName(const std::string& p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

Material materialOf(const Element& e) {
    Material m;
    m.albedo = vec3Attr(e.get("albedo"));
    m.emissionColor = vec3Attr(e.get("emission"));
    m.materialType = floatAttr(e.get("materialType"), 0.0f);
    m.emissionPower = floatAttr(e.get("emissionPower"), 0.0f);
    return m;
}

}  // namespace

SceneLoader::Status SceneLoader::Load(const std::string& path, Scene* scene, const std::string& assetRoot,
                                      std::string* logOut) {
    std::string log;
    auto finish = [&](Status s) {
        if (logOut) *logOut += log;
        return s;
    };
    std::ifstream in(path, std::ios::binary);
    if (!in) {
        log += "Failed to load scene XML: " + path + "\n";
        return finish(XmlUnreadable);  // scene untouched, as the reference
    }
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();

    std::vector<Element> elems;
    XmlScanner scanner(text);
    Status st = scanner.children(elems);
    if (st == XmlMalformed) {
        log += "Failed to load scene XML: " + path + "\n";
        return finish(st);  // tinyxml2 would fail LoadFile: scene untouched
    }
    scene->clear();
    if (st == NoSceneRoot) {
        log += "No <Scene> root.\n";
        return finish(st);
    }
    Status result = Ok;
    for (const Element& e : elems) {
        if (e.name == "Sphere") {
            Primitive p;
            p.type = PrimitiveType::Sphere;
            p.data0 = vec3Attr(e.get("position"));
            p.data1 = float3(floatAttr(e.get("radius"), 1.0f), 0.0f, 0.0f);
            p.data2 = float3(0.0f);
            p.material = materialOf(e);
            scene->addPrimitive(p);
        } else if (e.name == "Mesh") {
            const char* file = e.get("file");
            std::string given = file ? file : "";
            std::string resolved = given;
            if (!readable(resolved) && !assetRoot.empty()) resolved = assetRoot + "/" + baseName(given);
            if (!readable(resolved)) resolved = dirName(path) + "/" + baseName(given);
            Mesh mesh;
            if (!readObj(resolved, mesh, log)) result = MeshUnreadable;  // the reference carries on with no triangles
            const Transform place(vec3Attr(e.get("position")), floatAttr(e.get("scale"), 1.0f));  // R/Scene/Transform.h:8-20
            const Material m = materialOf(e);
            for (const mpt::uint3& t : mesh.triangles) {
                Primitive p;
                p.type = PrimitiveType::Triangle;
                p.data0 = place.apply(mesh.vertices[t.x]);
                p.data1 = place.apply(mesh.vertices[t.y]);
                p.data2 = place.apply(mesh.vertices[t.z]);
                p.material = m;
                scene->addPrimitive(p);
            }
        }
    }
    return finish(result);
}

void SceneLoader::LoadSceneFromXML(const std::string& path, Scene* scene) {
    std::string log;
    Load(path, scene, std::string(), &log);
    if (!log.empty()) std::fputs(log.c_str(), stdout);
}

}  // namespace MetalCppPathTracer

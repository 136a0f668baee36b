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

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <utility>
#include <vector>

namespace MetalCppPathTracer {

namespace {

using mpt::float3;

// ------------------------------------------------------------------------------------------ numbers
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

bool readObj(const std::string& path, Mesh& mesh, std::string& log) {
    std::ifstream in(path, std::ios::binary);
    if (!in) {
        log += "Failed to load OBJ: " + path + "\n";
        return false;
    }
    std::string line;
    std::vector<long> corner;
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        p += std::strspn(p, " \t");
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 1;
            float x = objReal(p), y = objReal(p), z = objReal(p);
            mesh.vertices.emplace_back(x, y, z);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 1;
            corner.clear();
            const long nv = static_cast<long>(mesh.vertices.size());
            for (;;) {
                p += std::strspn(p, " \t");
                if (*p == '\0' || *p == '\r' || *p == '\n') break;
                long raw = std::atol(p);                  // "a", "a/b", "a//c", "a/b/c": the vertex index leads
                corner.push_back(raw > 0 ? raw - 1 : (raw < 0 ? nv + raw : -1));
                p += std::strcspn(p, " \t\r\n");
            }
            for (size_t k = 2; k < corner.size(); ++k) {  // triangle fan (tinyobj triangulate = true)
                const long a = corner[0], b = corner[k - 1], c = corner[k];
                if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) {
                    log += "Invalid triangle indices\n";
                    continue;
                }
                mpt::uint3 t;
                t.x = static_cast<uint32_t>(a);
                t.y = static_cast<uint32_t>(b);
                t.z = static_cast<uint32_t>(c);
                mesh.triangles.push_back(t);
            }
        }
    }
    char msg[128];
    std::snprintf(msg, sizeof msg, "Loaded OBJ: %zu vertices, %zu triangles\n", mesh.vertices.size(), mesh.triangles.size());
    log += msg;
    return true;
}

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
    SceneLoader::Status children(std::vector<Element>& out) {
        int depth = 0;
        bool inside = false, found = false;
        while (pos_ < text_.size()) {
            if (text_[pos_] != '<') {
                ++pos_;
                continue;
            }
            if (startsWith("<!--")) {
                if (!skipPast("-->")) return SceneLoader::XmlMalformed;
            } else if (startsWith("<?")) {
                if (!skipPast("?>")) return SceneLoader::XmlMalformed;
            } else if (startsWith("<!")) {
                if (!skipPast(">")) return SceneLoader::XmlMalformed;
            } else if (startsWith("</")) {
                size_t close = text_.find('>', pos_);
                if (close == std::string::npos) return SceneLoader::XmlMalformed;
                --depth;
                if (inside && depth == 0) inside = false;
                pos_ = close + 1;
            } else {
                Element e;
                bool selfClosing = false;
                if (!openTag(e, selfClosing)) return SceneLoader::XmlMalformed;
                if (depth == 0 && !found && e.name == "Scene") {
                    found = true;
                    inside = !selfClosing;
                } else if (inside && depth == 1) {
                    out.push_back(std::move(e));
                }
                if (!selfClosing) ++depth;
            }
        }
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
            const float3 pos = vec3Attr(e.get("position"));
            const float scale = floatAttr(e.get("scale"), 1.0f);
            const Material m = materialOf(e);
            for (const mpt::uint3& t : mesh.triangles) {
                Primitive p;
                p.type = PrimitiveType::Triangle;
                p.data0 = pos + scale * mesh.vertices[t.x];
                p.data1 = pos + scale * mesh.vertices[t.y];
                p.data2 = pos + scale * mesh.vertices[t.z];
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

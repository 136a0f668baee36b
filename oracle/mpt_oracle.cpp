// mpt_oracle.cpp — CPU ORACLE for the per-pixel path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (metalpathtracer_amd/, include/,
// the C-ABI library, the CLI) includes, links or calls this file.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg load it (as the checker /
// the reported CPU baseline, never as the thing measured or shipped).
//
// What it is: a scalar C++ restatement of the reference's algorithm for the path
//   SceneLoader (XML + OBJ ingest) -> Scene::buildBVH -> buffer packers ->
//   fragmentMain -> rayColor -> firstHitBVH -> intersectAABB
// each function citing the reference file:line it follows
// (R/ = "/root/reference/MetalCpp Path Tracer/").
//
// Pinning: the reference has NO tests, golden vectors or CPU path,
// and its hot path is Metal Shading Language + Apple <simd/simd.h>, which cannot be
// compiled in this image without writing stand-in headers (not done).
// The oracle is therefore pinned against the values SURVEY.md App. C records from
// the reference's own shader/BVH text (RNG known answers, BVH node/leaf counts, pixel
// values to the printed digits, per-ray work counters: misses / emissive hits exact, node
// pops / primitive tests / rays within 5e-5 — the survey's whole-image FNV-1a hashes are
// NOT reproduced, its hashing convention is not recorded) — tests/test_oracle_pins.py
// — and its OBJ/XML ingest is pinned against the reference's vendored tinyobjloader /
// tinyxml2 compiled unchanged into oracle/_ref/ (tests/test_ingest_vs_ref.py).
// Modes that do not exist in the reference (philox RNG, the Scatter.h BSDF switch,
// batch spp accumulation) are this project's own specification and are
// "parity unpinned" by construction: there the oracle is the definition.
//
// Floating point: FP32 everywhere, IEEE, no FMA contraction (build: -ffp-contract=off,
// no -march=native), single-precision literals as in MSL.

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace orc {

// ----------------------------------------------------------------------------------
// small vector type: explicit component-wise FP32, left-to-right evaluation
// ----------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 v3s(float s) { return V3{s, s, s}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// ORC_DOT / ORC_NORMALIZE select alternative operation orders for tests/experiments/pin_sweep.py only (which of them
// reproduces the work counters of SURVEY.md App. C.3 best); the defaults (0, 0) are the definition everything else uses.
#ifndef ORC_DOT
#define ORC_DOT 0
#endif
#ifndef ORC_NORMALIZE
#define ORC_NORMALIZE 0
#endif
static inline float dot(V3 a, V3 b) {
#if ORC_DOT == 0
    return a.x * b.x + a.y * b.y + a.z * b.z;
#elif ORC_DOT == 1
    return a.x * b.x + (a.y * b.y + a.z * b.z);
#else
    return (a.z * b.z + a.y * b.y) + a.x * b.x;
#endif
}
static inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float length(V3 a) { return std::sqrt(dot(a, a)); }
// MSL normalize() is implementation-defined to the ulp (MTL_FAST_MATH).  Of v/len and v*(1/len)
// the reciprocal form reproduces the per-ray work counters SURVEY.md App. C.3 records from the
// reference text most closely (node pops within 2e-6, misses exact), so it is the definition here.
static inline V3 normalize(V3 a) {
#if ORC_NORMALIZE == 0
    float inv = 1.0f / length(a);
    return a * inv;
#elif ORC_NORMALIZE == 1
    return a / length(a);
#elif ORC_NORMALIZE == 2
    return a * std::sqrt(1.0f / dot(a, a));
#else
    double l2 = (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z;  // correctly rounded rsqrt
    return a * (float)(1.0 / std::sqrt(l2));
#endif
}
static inline V3 vmin(V3 a, V3 b) { return V3{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return V3{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
static inline float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline float clamp01(float v) { return std::fmin(std::fmax(v, 0.0f), 1.0f); }

// ----------------------------------------------------------------------------------
// RNG — R/Renderer/Shaders/Random.h:6-16 (device), R/Renderer/Renderer.cpp:30-41 (host)
// ----------------------------------------------------------------------------------
// PCG-RXS-M-XS *without* the final multiply (Random.h:6-11).
static inline uint32_t pcg_hash(uint32_t s) {
    uint32_t state = s * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state);
    return (word >> 22u) ^ word;
}
// Random.h:13-16 — the seed is taken BY VALUE; can return exactly 1.0f.
static inline float pcg_float(uint32_t s) { return (float)pcg_hash(s) / (float)((uint32_t)-1); }

// Random.h:32-35 — float sin-hash of uv (literal mode pixel seed).
static inline float fractf(float v) { return v - std::floor(v); }
static inline float sin_hash(float ux, float uy, V3 rs) {
    return fractf(std::sin(ux * rs.x + uy * rs.y) * rs.z);
}

// Philox4x32-10 (Salmon et al. 2011) — this project's benchmark RNG (not in the reference).
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }  // [0,1), 24 bits

// sin/cos of 2*pi*u for u in [0,1): quadrant reduction on u (exact), degree-9/8 Taylor
// polynomials on [-pi/4, pi/4], mul/add only, no FMA.  Spec in DESIGN.md §RNG; the HIP
// kernels implement the same sequence of IEEE operations.
static inline void sincos_2pi(float u, float* s_out, float* c_out) {
    float x = u * 4.0f;
    int q = (int)(x + 0.5f);
    float r = x - (float)q;
    float th = r * 1.57079637050628662109375f;
    float t2 = th * th;
    float ps = -1.98412701138295233249664306640625e-4f + t2 * 2.755731884462875314056873321533203125e-6f;
    ps = 8.3333337679505348205566406250e-3f + t2 * ps;
    ps = -0.16666667163372039794921875f + t2 * ps;
    float s = th + (th * t2) * ps;
    float pc = -1.38888892251998186111450195312500e-3f + t2 * 2.48015876422869041562080383300781250e-5f;
    pc = 4.1666667908430099487304687500e-2f + t2 * pc;
    pc = -0.5f + t2 * pc;
    float c = 1.0f + t2 * pc;
    switch (q & 3) {
        case 0: *s_out = s;  *c_out = c;  break;
        case 1: *s_out = c;  *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

// ----------------------------------------------------------------------------------
// Scene — R/Scene/Scene.h, R/Scene/Material.h
// ----------------------------------------------------------------------------------
struct Material {  // Material.h:8-14
    V3 albedo;
    float materialType;
    V3 emissionColor;
    float emissionPower;
};
struct Primitive {  // Scene.h:17-23
    int type;       // 0 sphere, 1 triangle (Scene.h:12-15)
    V3 data0, data1, data2;
    Material material;
};
struct BVHNode {  // Scene.h:25-30
    V3 bmin, bmax;
    int leftFirst;
    int count;
};

struct Scene {
    std::vector<Primitive> prims;
    std::vector<size_t> primIdx;
    std::vector<BVHNode> nodes;
    std::string log;  // what the reference would printf
};

static inline void prim_bounds(const Primitive& p, V3* lo, V3* hi) {  // Scene.h:199-209
    if (p.type == 0) {
        float r = p.data1.x;
        *lo = p.data0 - v3s(r);
        *hi = p.data0 + v3s(r);
    } else {
        *lo = vmin(p.data0, vmin(p.data1, p.data2));
        *hi = vmax(p.data0, vmax(p.data1, p.data2));
    }
}
static inline float surface_area(V3 lo, V3 hi) {  // Scene.h:319-322
    V3 d = hi - lo;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}

// Scene.h:195-317 — recursive full-sweep SAH keyed on data0[axis], leaf <= 8.
static int build_rec(Scene& sc, size_t start, size_t end) {
    const float FMAX = std::numeric_limits<float>::max();
    V3 bMin = v3s(FMAX), bMax = v3s(-FMAX);
    for (size_t i = start; i < end; ++i) {
        V3 lo, hi;
        prim_bounds(sc.prims[sc.primIdx[i]], &lo, &hi);
        bMin = vmin(bMin, lo);
        bMax = vmax(bMax, hi);
    }
    BVHNode node;
    node.bmin = bMin;
    node.bmax = bMax;
    node.leftFirst = (int)start;
    node.count = (int)(end - start);
    int nodeIndex = (int)sc.nodes.size();
    sc.nodes.push_back(node);
    if (node.count <= 8) return nodeIndex;  // Scene.h:223

    float bestCost = FMAX;
    int bestAxis = -1;
    size_t bestSplit = start + (end - start) / 2;
    const float parentArea = surface_area(bMin, bMax);
    if (parentArea <= 0.0f) return nodeIndex;  // Scene.h:232

    const size_t n = end - start;
    std::vector<V3> leftMin(n), leftMax(n), rightMin(n), rightMax(n);
    for (int axis = 0; axis < 3; ++axis) {
        // Scene.h:235-238 — std::sort (unstable): tie order is the library's.
        std::sort(sc.primIdx.begin() + start, sc.primIdx.begin() + end, [&](size_t a, size_t b) {
            return comp(sc.prims[a].data0, axis) < comp(sc.prims[b].data0, axis);
        });
        V3 cMin = v3s(FMAX), cMax = v3s(-FMAX);
        for (size_t i = start; i < end; ++i) {  // Scene.h:245-262
            V3 lo, hi;
            prim_bounds(sc.prims[sc.primIdx[i]], &lo, &hi);
            cMin = vmin(cMin, lo);
            cMax = vmax(cMax, hi);
            leftMin[i - start] = cMin;
            leftMax[i - start] = cMax;
        }
        cMin = v3s(FMAX);
        cMax = v3s(-FMAX);
        for (size_t i = end; i-- > start;) {  // Scene.h:264-281
            V3 lo, hi;
            prim_bounds(sc.prims[sc.primIdx[i]], &lo, &hi);
            cMin = vmin(cMin, lo);
            cMax = vmax(cMax, hi);
            rightMin[i - start] = cMin;
            rightMax[i - start] = cMax;
        }
        for (size_t i = 1; i < n; ++i) {  // Scene.h:283-299
            float saLeft = surface_area(leftMin[i - 1], leftMax[i - 1]);
            float saRight = surface_area(rightMin[i], rightMax[i]);
            size_t leftCount = i, rightCount = n - i;
            float cost = 0.125f + (saLeft / parentArea) * leftCount + (saRight / parentArea) * rightCount;
            if (cost < bestCost) {
                bestCost = cost;
                bestAxis = axis;
                bestSplit = start + i;
            }
        }
    }
    if (bestAxis == -1) return nodeIndex;  // Scene.h:302-303
    std::sort(sc.primIdx.begin() + start, sc.primIdx.begin() + end, [&](size_t a, size_t b) {  // Scene.h:305-308
        return comp(sc.prims[a].data0, bestAxis) < comp(sc.prims[b].data0, bestAxis);
    });
    int leftChild = build_rec(sc, start, bestSplit);
    int rightChild = build_rec(sc, bestSplit, end);
    sc.nodes[nodeIndex].leftFirst = leftChild;  // Scene.h:313-314
    sc.nodes[nodeIndex].count = -rightChild;
    return nodeIndex;
}

static void build_bvh(Scene& sc) {  // Scene.h:71-93
    std::stable_sort(sc.prims.begin(), sc.prims.end(),
                     [](const Primitive& a, const Primitive& b) { return a.type < b.type; });
    sc.primIdx.resize(sc.prims.size());
    for (size_t i = 0; i < sc.prims.size(); ++i) sc.primIdx[i] = i;
    sc.nodes.clear();
    build_rec(sc, 0, sc.prims.size());
}

// ----------------------------------------------------------------------------------
// Ingest — R/Scene/SceneLoader.cpp:14-133 over tinyxml2 11.0.0 / tinyobjloader 2.0.0
// ----------------------------------------------------------------------------------
// tinyobjloader 2.0.0 tryParseDouble (R/tiny_obj_loader.h:897-1028), published algorithm:
// digits accumulated into a double mantissa, decimals added as digit * 10^-k (LUT for
// k < 8, pow(10,-k) beyond), exponent applied as ldexp(m * 5^e, e); then cast to float.
// (restated from tinyobjloader 2.0.0, MIT License, Copyright (c) 2012-Present Syoyo Fujita and contributors —
// THIRD_PARTY_NOTICES.md)
static bool tinyobj_parse_double(const char* s, const char* s_end, double* result) {
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char* curr = s;
    int read = 0;
    bool end_not_reached = false, leading_dot = false;
    auto isdig = [](char c) { return c >= '0' && c <= '9'; };
    if (*curr == '+' || *curr == '-') {
        sign = *curr;
        curr++;
        if (curr != s_end && *curr == '.') leading_dot = true;
    } else if (isdig(*curr)) {
    } else if (*curr == '.') {
        leading_dot = true;
    } else {
        return false;
    }
    end_not_reached = (curr != s_end);
    if (!leading_dot) {
        while (end_not_reached && isdig(*curr)) {
            mantissa *= 10;
            mantissa += (int)(*curr - '0');
            curr++;
            read++;
            end_not_reached = (curr != s_end);
        }
        if (read == 0) return false;
    }
    if (!end_not_reached) goto assemble;
    if (*curr == '.') {
        curr++;
        read = 1;
        end_not_reached = (curr != s_end);
        while (end_not_reached && isdig(*curr)) {
            static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            mantissa += (int)(*curr - '0') * (read < 8 ? lut[read] : std::pow(10.0, -read));
            read++;
            curr++;
            end_not_reached = (curr != s_end);
        }
    } else if (*curr == 'e' || *curr == 'E') {
    } else {
        goto assemble;
    }
    if (!end_not_reached) goto assemble;
    if (*curr == 'e' || *curr == 'E') {
        curr++;
        end_not_reached = (curr != s_end);
        if (end_not_reached && (*curr == '+' || *curr == '-')) {
            exp_sign = *curr;
            curr++;
        } else if (isdig(*curr)) {
        } else {
            return false;
        }
        read = 0;
        end_not_reached = (curr != s_end);
        while (end_not_reached && isdig(*curr)) {
            if (exponent > 2147483647 / 10) return false;
            exponent *= 10;
            exponent += (int)(*curr - '0');
            curr++;
            read++;
            end_not_reached = (curr != s_end);
        }
        exponent *= (exp_sign == '+' ? 1 : -1);
        if (read == 0) return false;
    }
assemble:
    *result = (sign == '+' ? 1 : -1) *
              (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}
static float tinyobj_parse_real(const char** tok) {  // tiny_obj_loader.h:1030-1038
    (*tok) += strspn(*tok, " \t");
    const char* end = (*tok) + strcspn(*tok, " \t\r");
    double val = 0.0;
    tinyobj_parse_double(*tok, end, &val);
    *tok = end;
    return (float)val;
}

// ---- OBJ ingest: restatement of what tinyobj::LoadObj (tinyobjloader 2.0.0, vendored at
// R/tiny_obj_loader.h, called with its defaults at R/Scene/SceneLoader.cpp:26) does to the vertex and
// face statements, followed by the reference's own filtering (SceneLoader.cpp:40-68).  Checked against the
// real library by tests/test_ingest_vs_ref.py (oracle/_ref).
//
//  * `v x y z`: floats via tinyobj's own parser (tinyobj_parse_real above).
//  * `f`/`l`/`p` corners `i`, `i/j`, `i//k`, `i/j/k`: atoi; a vertex index of 0, or a relative (negative)
//    index reaching before the first element, fails the WHOLE file (tiny_obj_loader.h:819-850,1188-1239),
//    which the reference reports as "Failed to load OBJ" and carries on with no triangles.
//  * faces are triangulated when their group is flushed (at `g`, `o`, end of file) against the vertices
//    read so far: 3 corners -> as is; 4 corners -> split along the shorter diagonal, 0-2 only when strictly
//    shorter (h:1510-1608); 5+ corners -> the library's ear clipping on the two axes picked from the first
//    non-degenerate corner (h:1738-1962), including its quirks (the `area` term uses two vertices only).
//  * the reference then drops triangles with an index >= the final vertex count (SceneLoader.cpp:62-66).
struct ObjFace {
    std::vector<int> v;
};

// point-in-triangle by crossing number, float arithmetic in the library's order (h:1440-1450)
static bool obj_pnpoly3(const float* px, const float* py, float tx, float ty) {
    bool inside = false;
    for (int i = 0, j = 2; i < 3; j = i++) {
        if (((py[i] > ty) != (py[j] > ty)) && (tx < (px[j] - px[i]) * (ty - py[i]) / (py[j] - py[i]) + px[i]))
            inside = !inside;
    }
    return inside;
}

static void obj_emit(std::vector<uint32_t>& out, int a, int b, int c) {
    out.push_back((uint32_t)a);
    out.push_back((uint32_t)b);
    out.push_back((uint32_t)c);
}

static void obj_triangulate(const ObjFace& face, const std::vector<float>& v, std::vector<uint32_t>& out) {
    const size_t n = face.v.size();
    const size_t vs = v.size();
    if (n < 3) return;  // "Degenerated face"
    if (n == 3) {
        obj_emit(out, face.v[0], face.v[1], face.v[2]);
        return;
    }
    auto in_range = [&](int i) { return 3 * (size_t)i + 2 < vs; };
    if (n == 4) {
        const int* q = face.v.data();
        if (!in_range(q[0]) || !in_range(q[1]) || !in_range(q[2]) || !in_range(q[3])) return;
        const float* p0 = &v[3 * (size_t)q[0]];
        const float* p1 = &v[3 * (size_t)q[1]];
        const float* p2 = &v[3 * (size_t)q[2]];
        const float* p3 = &v[3 * (size_t)q[3]];
        float ax = p2[0] - p0[0], ay = p2[1] - p0[1], az = p2[2] - p0[2];
        float bx = p3[0] - p1[0], by = p3[1] - p1[1], bz = p3[2] - p1[2];
        float d02 = ax * ax + ay * ay + az * az;
        float d13 = bx * bx + by * by + bz * bz;
        if (d02 < d13) {
            obj_emit(out, q[0], q[1], q[2]);
            obj_emit(out, q[0], q[2], q[3]);
        } else {
            obj_emit(out, q[0], q[1], q[3]);
            obj_emit(out, q[1], q[2], q[3]);
        }
        return;
    }
    // 5+ corners.  Projection axes from the first corner whose edge cross product is not ~0.
    size_t ax0 = 1, ax1 = 2;
    for (size_t k = 0; k < n; ++k) {
        int i0 = face.v[k % n], i1 = face.v[(k + 1) % n], i2 = face.v[(k + 2) % n];
        if (!in_range(i0) || !in_range(i1) || !in_range(i2)) continue;
        const float* a = &v[3 * (size_t)i0];
        const float* b = &v[3 * (size_t)i1];
        const float* c = &v[3 * (size_t)i2];
        float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2];
        float e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
        float cx = std::fabs(e0y * e1z - e0z * e1y);
        float cy = std::fabs(e0z * e1x - e0x * e1z);
        float cz = std::fabs(e0x * e1y - e0y * e1x);
        const float eps = std::numeric_limits<float>::epsilon();
        if (cx > eps || cy > eps || cz > eps) {
            if (!(cx > cy && cx > cz)) {
                ax0 = 0;
                if (cz > cx && cz > cy) ax1 = 1;
            }
            break;
        }
    }
    std::vector<int> rest = face.v;
    size_t guess = 0, budget = n, last_size = n;
    while (rest.size() > 3 && budget > 0) {
        const size_t m = rest.size();
        if (guess >= m) guess -= m;
        if (last_size != m) {
            last_size = m;
            budget = m;
        } else {
            --budget;
        }
        int ind[3];
        float px[3], py[3];
        for (size_t k = 0; k < 3; ++k) {
            ind[k] = rest[(guess + k) % m];
            size_t base = 3 * (size_t)ind[k];
            if (base + ax0 >= vs || base + ax1 >= vs) {
                px[k] = 0.0f;
                py[k] = 0.0f;
            } else {
                px[k] = v[base + ax0];
                py[k] = v[base + ax1];
            }
        }
        float e0x = px[1] - px[0], e0y = py[1] - py[0];
        float e1x = px[2] - px[1], e1y = py[2] - py[1];
        float cross = e0x * e1y - e0y * e1x;
        float area = (px[0] * py[1] - py[0] * px[1]) * 0.5f;
        if (cross * area < 0.0f) {  // reflex corner by the library's test
            ++guess;
            continue;
        }
        bool covered = false;
        for (size_t o = 3; o < m; ++o) {
            size_t base = 3 * (size_t)rest[(guess + o) % m];
            if (base + ax0 >= vs || base + ax1 >= vs) continue;
            if (obj_pnpoly3(px, py, v[base + ax0], v[base + ax1])) {
                covered = true;
                break;
            }
        }
        if (covered) {
            ++guess;
            continue;
        }
        obj_emit(out, ind[0], ind[1], ind[2]);
        rest.erase(rest.begin() + (long)((guess + 1) % m));
    }
    if (rest.size() == 3) obj_emit(out, rest[0], rest[1], rest[2]);
}

// One corner `i[/j][/k]`; counts = {vertices, normals, texcoords} read so far.  false = the file fails.
static bool obj_corner(const char** tok, const int counts[3], int* v_out) {
    auto fix = [](int raw, int n, bool zero_ok, int* out) {
        if (raw > 0) {
            *out = raw - 1;
            return true;
        }
        if (raw == 0) {
            *out = -1;
            return zero_ok;
        }
        *out = n + raw;
        return *out >= 0;
    };
    int dummy;
    if (!fix(atoi(*tok), counts[0], false, v_out)) return false;
    *tok += strcspn(*tok, "/ \t\r");
    if (**tok != '/') return true;
    ++*tok;
    if (**tok == '/') {  // i//k
        ++*tok;
        if (!fix(atoi(*tok), counts[1], true, &dummy)) return false;
        *tok += strcspn(*tok, "/ \t\r");
        return true;
    }
    if (!fix(atoi(*tok), counts[2], true, &dummy)) return false;  // i/j
    *tok += strcspn(*tok, "/ \t\r");
    if (**tok != '/') return true;
    ++*tok;  // i/j/k
    if (!fix(atoi(*tok), counts[1], true, &dummy)) return false;
    *tok += strcspn(*tok, "/ \t\r");
    return true;
}

static bool load_obj(const std::string& path, std::vector<V3>& verts, std::vector<uint32_t>& tris, std::string& log) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        log += "Failed to load OBJ: " + path + "\n";
        return false;
    }
    std::string text;
    {
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) text.append(chunk, got);
        fclose(f);
    }
    std::vector<float> v;
    std::vector<ObjFace> group;
    std::vector<uint32_t> raw;
    int counts[3] = {0, 0, 0};
    auto flush = [&]() {
        for (const ObjFace& fc : group) obj_triangulate(fc, v, raw);
        group.clear();
    };
    auto is_sp = [](char c) { return c == ' ' || c == '\t'; };
    bool ok = true;
    size_t pos = 0;
    while (ok && pos < text.size()) {
        // line ends: \n, \r\n or a lone \r (h:767-799); trailing blanks are trimmed first (h:2094-2097)
        size_t eol = text.find_first_of("\r\n", pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        if (eol < text.size() && text[eol] == '\r' && pos < text.size() && text[pos] == '\n') ++pos;
        size_t keep = line.find_last_not_of(" \t");
        line.erase(keep == std::string::npos ? 0 : keep + 1);
        const char* t = line.c_str();
        t += strspn(t, " \t");
        if (t[0] == 'v' && is_sp(t[1])) {
            t += 2;
            float x = tinyobj_parse_real(&t), y = tinyobj_parse_real(&t), z = tinyobj_parse_real(&t);
            v.push_back(x);
            v.push_back(y);
            v.push_back(z);
            counts[0]++;
        } else if (t[0] == 'v' && t[1] == 'n' && is_sp(t[2])) {
            counts[1]++;
        } else if (t[0] == 'v' && t[1] == 't' && is_sp(t[2])) {
            counts[2]++;
        } else if ((t[0] == 'f' || t[0] == 'l' || t[0] == 'p') && is_sp(t[1])) {
            const bool is_face = t[0] == 'f';
            t += 2;
            t += strspn(t, " \t");
            ObjFace fc;
            while (*t != '\0' && *t != '\r' && *t != '\n' && *t != '#') {
                int vi = -1;
                if (!obj_corner(&t, counts, &vi)) {
                    ok = false;
                    break;
                }
                fc.v.push_back(vi);
                t += strspn(t, " \t\r");
            }
            if (ok && is_face) group.push_back(std::move(fc));
        } else if ((t[0] == 'g' || t[0] == 'o') && is_sp(t[1])) {  // h:2936,2990: a bare "g" does not match
            flush();
        }
    }
    if (!ok) {
        log += "Failed to load OBJ: " + path + "\n";
        return false;
    }
    flush();
    for (size_t i = 0; i + 2 < v.size(); i += 3) verts.push_back(v3(v[i], v[i + 1], v[i + 2]));
    const size_t nv = verts.size();
    for (size_t i = 0; i + 2 < raw.size(); i += 3) {
        if (raw[i] >= nv || raw[i + 1] >= nv || raw[i + 2] >= nv) {
            log += "Invalid triangle indices\n";  // SceneLoader.cpp:62-66
            continue;
        }
        tris.push_back(raw[i]);
        tris.push_back(raw[i + 1]);
        tris.push_back(raw[i + 2]);
    }
    return true;
}

// SceneLoader.cpp:14-18 — sscanf "%f,%f,%f", missing components stay 0.
static V3 parse_vec3(const char* s) {
    float x = 0, y = 0, z = 0;
    if (s) sscanf(s, "%f,%f,%f", &x, &y, &z);
    return v3(x, y, z);
}
// tinyxml2 XMLUtil::ToFloat = sscanf("%f") (R/tinyxml2.cpp, used by FloatAttribute).
static float parse_float_attr(const char* s, float dflt) {
    float v = dflt;
    if (s && sscanf(s, "%f", &v) == 1) return v;
    return dflt;
}

struct XmlElem {
    std::string name;
    std::vector<std::pair<std::string, std::string>> attrs;
    const char* attr(const char* k) const {
        for (auto& a : attrs)
            if (a.first == k) return a.second.c_str();
        return nullptr;
    }
};
// Minimal XML reader for the reference schema (R/scene.xml): comments, a <Scene> root,
// attribute-only child elements.  Returns the children of <Scene> in document order.
// Return: 0 = ok, 1 = not well-formed (tinyxml2's LoadFile would fail: mismatched or unclosed elements, duplicate
// attribute names, unquoted values, no element at all), 2 = well-formed but no <Scene> root.
static int parse_scene_xml(const std::string& text, std::vector<XmlElem>& out, std::string& log) {
    size_t i = 0, n = text.size();
    bool in_scene = false, saw_scene = false;
    int depth = 0;
    std::vector<std::string> open_names;
    size_t n_elements = 0;
    while (i < n) {
        if (text[i] != '<') {
            ++i;
            continue;
        }
        n_elements++;  // any node: only a document without a single one is "empty" for tinyxml2
        if (text.compare(i, 4, "<!--") == 0) {
            size_t e = text.find("-->", i + 4);
            if (e == std::string::npos) return 1;
            i = e + 3;
            continue;
        }
        if (text.compare(i, 2, "<?") == 0) {
            size_t e = text.find("?>", i + 2);
            if (e == std::string::npos) return 1;
            i = e + 2;
            continue;
        }
        if (text.compare(i, 2, "<!") == 0) {  // DOCTYPE / CDATA: skipped
            size_t e = text.find('>', i);
            if (e == std::string::npos) return 1;
            i = e + 1;
            continue;
        }
        if (i + 1 < n && text[i + 1] == '/') {
            size_t e = text.find('>', i);
            if (e == std::string::npos) return 1;
            std::string nm = text.substr(i + 2, e - i - 2);
            while (!nm.empty() && isspace((unsigned char)nm.back())) nm.pop_back();
            if (open_names.empty() || open_names.back() != nm) return 1;
            open_names.pop_back();
            depth--;
            if (nm == "Scene" && depth == 0) in_scene = false;
            i = e + 1;
            continue;
        }
        size_t j = i + 1;
        while (j < n && !isspace((unsigned char)text[j]) && text[j] != '>' && text[j] != '/') ++j;
        XmlElem el;
        el.name = text.substr(i + 1, j - i - 1);
        bool selfclose = false;
        while (j < n) {
            while (j < n && isspace((unsigned char)text[j])) ++j;
            if (j >= n) return 1;
            if (text[j] == '/') {
                selfclose = true;
                ++j;
                continue;
            }
            if (text[j] == '>') {
                ++j;
                break;
            }
            size_t k = j;
            while (k < n && text[k] != '=' && text[k] != '>' && text[k] != '/' && !isspace((unsigned char)text[k])) ++k;
            std::string key = text.substr(j, k - j);
            while (k < n && isspace((unsigned char)text[k])) ++k;
            if (k >= n || text[k] != '=' || key.empty()) return 1;  // an attribute needs ="value"
            ++k;
            while (k < n && isspace((unsigned char)text[k])) ++k;
            if (k >= n || (text[k] != '"' && text[k] != '\'')) return 1;
            char qc = text[k];
            size_t e = text.find(qc, k + 1);
            if (e == std::string::npos) return 1;
            std::string val = text.substr(k + 1, e - k - 1);
            // the five predefined entities
            std::string dec;
            for (size_t p = 0; p < val.size(); ++p) {
                if (val[p] == '&') {
                    if (val.compare(p, 5, "&amp;") == 0) { dec += '&'; p += 4; continue; }
                    if (val.compare(p, 4, "&lt;") == 0) { dec += '<'; p += 3; continue; }
                    if (val.compare(p, 4, "&gt;") == 0) { dec += '>'; p += 3; continue; }
                    if (val.compare(p, 6, "&quot;") == 0) { dec += '"'; p += 5; continue; }
                    if (val.compare(p, 6, "&apos;") == 0) { dec += '\''; p += 5; continue; }
                }
                dec += val[p];
            }
            for (const auto& kv : el.attrs)
                if (kv.first == key) return 1;
            el.attrs.emplace_back(key, dec);
            j = e + 1;
        }
        if (el.name.empty()) return 1;
        if (!selfclose) open_names.push_back(el.name);
        if (depth == 0 && el.name == "Scene" && !saw_scene) {
            saw_scene = true;
            in_scene = !selfclose;
        } else if (in_scene && depth == 1) {
            out.push_back(el);
        }
        if (!selfclose) depth++;
        i = j;
    }
    if (!open_names.empty() || n_elements == 0) return 1;
    if (!saw_scene) {
        log += "No <Scene> root.\n";
        return 2;
    }
    return 0;
}

static std::string dirname_of(const std::string& p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
}
static std::string basename_of(const std::string& p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}
static bool file_exists(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

// SceneLoader.cpp:75-133.  `asset_root` is this project's remedy for the reference's absolute
// macOS mesh paths: a `file=` that does not open is retried as
// <asset_root>/<basename> and then <dir of xml>/<basename>.
static bool load_scene_xml(const std::string& path, const std::string& asset_root, Scene& sc) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        sc.log += "Failed to load scene XML: " + path + "\n";
        return false;  // SceneLoader.cpp:77-80 (scene left as-is)
    }
    std::string text;
    char tmp[4096];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) text.append(tmp, got);
    fclose(f);
    std::vector<XmlElem> elems;
    std::string xlog;
    const int parsed = parse_scene_xml(text, elems, xlog);
    if (parsed == 1) {
        sc.log += "Failed to load scene XML: " + path + "\n";
        return false;  // LoadFile fails: SceneLoader.cpp:77-80, scene left as-is
    }
    sc.prims.clear();  // SceneLoader.cpp:82
    sc.nodes.clear();
    sc.primIdx.clear();
    sc.log += xlog;
    if (parsed != 0) return false;
    for (const XmlElem& e : elems) {
        if (e.name == "Sphere") {  // SceneLoader.cpp:92-106
            Primitive p;
            p.type = 0;
            p.data0 = parse_vec3(e.attr("position"));
            float r = parse_float_attr(e.attr("radius"), 1.0f);
            p.data1 = v3(r, 0, 0);
            p.data2 = v3s(0);
            p.material.albedo = parse_vec3(e.attr("albedo"));
            p.material.emissionColor = parse_vec3(e.attr("emission"));
            p.material.materialType = parse_float_attr(e.attr("materialType"), 0);
            p.material.emissionPower = parse_float_attr(e.attr("emissionPower"), 0);
            sc.prims.push_back(p);
        } else if (e.name == "Mesh") {  // SceneLoader.cpp:107-131
            std::vector<V3> verts;
            std::vector<uint32_t> tris;
            std::string file = e.attr("file") ? e.attr("file") : "";
            std::string resolved = file;
            if (!file_exists(resolved) && !asset_root.empty()) resolved = asset_root + "/" + basename_of(file);
            if (!file_exists(resolved)) resolved = dirname_of(path) + "/" + basename_of(file);
            load_obj(resolved, verts, tris, sc.log);
            V3 pos = parse_vec3(e.attr("position"));
            float scale = parse_float_attr(e.attr("scale"), 1.0f);
            Material m;
            m.albedo = parse_vec3(e.attr("albedo"));
            m.emissionColor = parse_vec3(e.attr("emission"));
            m.materialType = parse_float_attr(e.attr("materialType"), 0);
            m.emissionPower = parse_float_attr(e.attr("emissionPower"), 0);
            for (size_t t = 0; t + 2 < tris.size(); t += 3) {
                Primitive p;
                p.type = 1;
                p.data0 = pos + scale * verts[tris[t + 0]];
                p.data1 = pos + scale * verts[tris[t + 1]];
                p.data2 = pos + scale * verts[tris[t + 2]];
                p.material = m;
                sc.prims.push_back(p);
            }
        }
    }
    return true;
}

// ----------------------------------------------------------------------------------
// The hot path — R/Renderer/Shaders/{Fragment.metal, PathTracing.h, Scatter.h}
// ----------------------------------------------------------------------------------
struct Ray {
    V3 o, d;
};
struct Hit {  // Structs.h:12-20
    float t;
    V3 point, normal;
    bool frontFace;
    int primitiveId;
    int isTriangle;
};
// Diagnostic histogram of BVH node visits per ray (bins 0..255, last bin = 255+), split primary / bounce.
// Filled only when enabled (orc_histogram_enable); used to size the GPU pipeline's traversal budgets.
static uint64_t g_hist[2][256];
static bool g_hist_on = false;
// Optional observer of every closest-hit query of a render (tests of alternative traversal orders compare their own
// answer with the reference-order one, ray by ray).  Called from the render threads; null = off.
typedef void (*RayHook)(const float o[3], const float d[3], float t, int primitiveId, void* user);
static RayHook g_ray_hook = nullptr;
static void* g_ray_hook_user = nullptr;

struct Counters {
    uint64_t rays, node_pops, aabb_pass, prim_tests, sphere_tests, tri_tests, pushes, misses, bounces, emissive_hits,
        depth_exhausted, paths;
};

// PathTracing.h:52-72
static inline bool intersect_aabb(const Ray& r, V3 bmin, V3 bmax, float tMin, float tMax) {
    for (int i = 0; i < 3; ++i) {
        float invD = 1.0f / comp(r.d, i);
        float t0 = (comp(bmin, i) - comp(r.o, i)) * invD;
        float t1 = (comp(bmax, i) - comp(r.o, i)) * invD;
        if (invD < 0.0f) {
            float tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        tMin = std::fmax(tMin, t0);
        tMax = std::fmin(tMax, t1);
        if (tMax <= tMin) return false;
    }
    return true;
}

static inline int float_bits_to_int(float f) {
    int i;
    memcpy(&i, &f, 4);
    return i;
}

// PathTracing.h:75-204
static Hit first_hit_bvh(const Ray& r, const float* bvh, const float* prims, const int* primIdx, Counters* ct) {
    Hit in;
    in.t = INFINITY;
    in.primitiveId = -1;
    in.isTriangle = 0;
    in.point = v3s(0);
    in.normal = v3s(0);
    in.frontFace = false;
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        int nodeIdx = stack[--sp];
        ct->node_pops++;
        const float* n0 = bvh + 8 * (size_t)nodeIdx;
        V3 bmin = v3(n0[0], n0[1], n0[2]);
        V3 bmax = v3(n0[4], n0[5], n0[6]);
        int leftFirst = float_bits_to_int(n0[3]);
        int second = float_bits_to_int(n0[7]);
        if (!intersect_aabb(r, bmin, bmax, 0.0001f, in.t)) continue;
        ct->aabb_pass++;
        if (second > 0) {
            for (int i = 0; i < second; ++i) {
                int pi = primIdx[leftFirst + i];
                const float* p = prims + 12 * (size_t)pi;
                int ptype = (int)p[3];
                float tHit = INFINITY;
                V3 n = v3s(0), hit = v3s(0);
                bool hitThis = false;
                ct->prim_tests++;
                if (ptype == 0) {  // PathTracing.h:120-142
                    ct->sphere_tests++;
                    V3 center = v3(p[0], p[1], p[2]);
                    float radius = p[4];
                    V3 oc = r.o - center;
                    float a = dot(r.d, r.d);
                    float b = dot(oc, r.d);
                    float c = dot(oc, oc) - radius * radius;
                    float disc = b * b - a * c;
                    if (disc > 0.0f) {
                        float sq = std::sqrt(disc);
                        float temp = (-b - sq) / a;
                        if (temp < in.t && temp > 0.0001f) {
                            tHit = temp;
                            hit = r.o + tHit * r.d;
                            n = normalize(hit - center);
                            hitThis = true;
                        }
                    }
                } else if (ptype == 1) {  // PathTracing.h:143-176
                    ct->tri_tests++;
                    V3 v0 = v3(p[0], p[1], p[2]), v1 = v3(p[4], p[5], p[6]), v2 = v3(p[8], p[9], p[10]);
                    V3 e1 = v1 - v0, e2 = v2 - v0;
                    V3 h = cross(r.d, e2);
                    float a = dot(e1, h);
                    if (std::fabs(a) > 1e-5f) {
                        float f = 1.0f / a;
                        V3 s = r.o - v0;
                        float u = f * dot(s, h);
                        if (u >= 0.0f && u <= 1.0f) {
                            V3 q = cross(s, e1);
                            float v = f * dot(r.d, q);
                            if (v >= 0.0f && u + v <= 1.0f) {
                                float tt = f * dot(e2, q);
                                if (tt > 0.0001f && tt < in.t) {
                                    tHit = tt;
                                    hit = r.o + tHit * r.d;
                                    n = normalize(cross(e1, e2));
                                    hitThis = true;
                                    in.isTriangle = 1;
                                }
                            }
                        }
                    }
                }
                if (hitThis && tHit < in.t) {  // PathTracing.h:178-185
                    in.t = tHit;
                    in.primitiveId = pi;
                    in.normal = n;
                    in.point = hit;
                    in.isTriangle = ptype;
                }
            }
        } else {  // PathTracing.h:188-193 — push left then right: right is popped first
            int rightChild = -second;
            stack[sp++] = leftFirst;
            stack[sp++] = rightChild;
            ct->pushes += 2;
        }
    }
    if (in.primitiveId != -1) {  // PathTracing.h:196-201
        in.frontFace = dot(in.normal, r.d) < 0.0f;
        if (!in.frontFace) in.normal = -in.normal;
    }
    return in;
}

enum { RNG_LITERAL = 0, RNG_PHILOX = 1 };
enum { BSDF_LAMBERT = 0, BSDF_SCATTER = 1, BSDF_SCATTER_ALL = 2 };

struct PathRng {
    int mode;
    uint32_t seed;                        // literal: the stuck PCG seed (PathTracing.h:25-31)
    uint32_t pixel, sample, key0, key1;   // philox: counter (pixel, sample, bounce, 0), key (key0, key1)
};

// PathTracing.h:25-31.  literal: z and phi share one u and the seed never advances.
// philox: u0 -> z, u1 -> phi from Philox4x32-10(counter=(pixel,sample,bounce,0)).
static inline V3 random_unit_vector(const PathRng& g, uint32_t bounce, float* u_extra) {
    if (g.mode == RNG_LITERAL) {
        float z = 2.0f * pcg_float(g.seed) - 1.0f;
        float t = 2.0f * 3.14159274101257324f * pcg_float(g.seed);
        float rr = std::sqrt(1.0f - z * z);
        *u_extra = pcg_float(g.seed);
        return v3(rr * std::cos(t), rr * std::sin(t), z);
    }
    uint32_t o[4];
    philox4x32_10(g.pixel, g.sample, bounce, 0u, g.key0, g.key1, o);
    float z = 2.0f * u01(o[0]) - 1.0f;
    float s, c;
    sincos_2pi(u01(o[1]), &s, &c);
    float rr = std::sqrt(1.0f - z * z);
    *u_extra = u01(o[2]);
    return v3(rr * c, rr * s, z);
}

// Random.h:18-30 — a point of the cube [-1,1]^3 (the seed is by value: the caller's stream does not move; literal mode
// advances a copy three times as the reference does, philox takes words 0, 1, 2 of the bounce's block: word 2 is also
// the Fresnel number, which only a dielectric bounce uses)
static inline V3 random_float3(const PathRng& g, uint32_t bounce) {
    if (g.mode == RNG_LITERAL) {
        uint32_t s = g.seed;
        const float x = pcg_float(s) * 2.0f - 1.0f;
        s = pcg_hash(s);
        const float y = pcg_float(s) * 2.0f - 1.0f;
        s = pcg_hash(s);
        const float z = pcg_float(s) * 2.0f - 1.0f;
        return v3(x, y, z);
    }
    uint32_t o[4];
    philox4x32_10(g.pixel, g.sample, bounce, 0u, g.key0, g.key1, o);
    return v3(u01(o[0]) * 2.0f - 1.0f, u01(o[1]) * 2.0f - 1.0f, u01(o[2]) * 2.0f - 1.0f);
}

static inline V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }
static inline V3 refract(V3 i, V3 n, float eta) {  // MSL refract()
    float d = dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3s(0);
    return eta * i - (eta * d + std::sqrt(k)) * n;
}
// Scatter.h:10-20
static inline bool mirror_angle(float ri, V3 normal, V3 rayDir, float u) {
    float cosT = dot(-1.0f * rayDir, normal);
    float sinT = std::sqrt(1.0f - cosT * cosT);
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    float m = 1.0f - cosT;
    float m2 = m * m;
    float refl = r0 + (1.0f - r0) * (m2 * m2 * m);  // pow(x,5) as mul chain (spec: DESIGN.md)
    return (ri * sinT > 1.0f) || (refl > u);
}

// PathTracing.h:207-259 (+ Scatter.h:22-43 when bsdf == BSDF_SCATTER)
static void ray_color(Ray r, const float* bvh, const float* prims, const float* mats, uint32_t primitiveCount,
                      const int* primIdx, const PathRng& g, int maxDepth, int bsdf, float out[4], Counters* ct) {
    float ab[4] = {1, 1, 1, 1};
    float li[4] = {0, 0, 0, 0};
    int depth = 0;
    for (; depth < maxDepth; ++depth) {
        ct->rays++;
        const uint64_t pops_before = ct->node_pops;
        Hit hit = first_hit_bvh(r, bvh, prims, primIdx, ct);
        if (g_hist_on) {
            uint64_t n = ct->node_pops - pops_before;
            __atomic_fetch_add(&g_hist[depth == 0 ? 0 : 1][n > 255 ? 255 : n], 1, __ATOMIC_RELAXED);
        }
        if (g_ray_hook) {
            const float ro[3] = {r.o.x, r.o.y, r.o.z}, rd[3] = {r.d.x, r.d.y, r.d.z};
            g_ray_hook(ro, rd, hit.t, hit.primitiveId, g_ray_hook_user);
        }
        if (hit.primitiveId == -1) {  // PathTracing.h:225-232
            V3 ud = normalize(r.d);
            float t = 0.5f * (ud.y + 1.0f);
            V3 a = v3s(1.0f), b = v3(0.6f, 0.7f, 1.0f);
            V3 sky = a + (b - a) * t;  // MSL mix(x,y,a) = x + (y-x)*a
            li[0] += ab[0] * sky.x;
            li[1] += ab[1] * sky.y;
            li[2] += ab[2] * sky.z;
            li[3] += ab[3] * 1.0f;
            ct->misses++;
            break;
        }
        int matIndex = hit.primitiveId * 2;
        if (matIndex + 1 >= (int)primitiveCount * 2) break;  // PathTracing.h:234-236
        const float* m0 = mats + 4 * (size_t)matIndex;
        const float* m1 = m0 + 4;
        V3 albedo = v3(m0[0], m0[1], m0[2]);
        float materialType = m0[3];
        V3 emission = v3(m1[0], m1[1], m1[2]);
        float emissionPower = m1[3];
        if (emissionPower > 0.0f || materialType == 2) {  // PathTracing.h:245-249
            li[0] += ab[0] * emission.x * emissionPower;
            li[1] += ab[1] * emission.y * emissionPower;
            li[2] += ab[2] * emission.z * emissionPower;
            li[3] += ab[3] * 1.0f * emissionPower;
            ct->emissive_hits++;
        }
        float u_extra;
        V3 ruv = random_unit_vector(g, (uint32_t)depth, &u_extra);
        V3 newDir;
        if (bsdf == BSDF_SCATTER_ALL && materialType == 0.0f) {
            newDir = normalize(hit.normal + normalize(random_float3(g, (uint32_t)depth)));  // Scatter.h:24-27,42
        } else if (bsdf == BSDF_LAMBERT || materialType == 0.0f) {
            newDir = normalize(hit.normal + ruv);  // PathTracing.h:252-254
        } else if (materialType < 0.0f) {          // Scatter.h:28-31
            newDir = normalize(reflect(r.d, hit.normal));
        } else {  // Scatter.h:32-40
            float ri = hit.frontFace ? 1.0f / materialType : materialType;
            newDir = mirror_angle(ri, hit.normal, r.d, u_extra) ? reflect(r.d, hit.normal)
                                                                : refract(r.d, hit.normal, ri);
            newDir = normalize(newDir);
        }
        if (bsdf != BSDF_LAMBERT && materialType > 0.0f && dot(newDir, hit.normal) < 0.0f)
            r.o = hit.point - 0.0001f * hit.normal;  // transmitted ray starts on the far side (own spec)
        else
            r.o = hit.point + 0.0001f * hit.normal;  // PathTracing.h:253
        r.d = newDir;
        ab[0] *= albedo.x;  // PathTracing.h:255
        ab[1] *= albedo.y;
        ab[2] *= albedo.z;
        ab[3] *= 1.0f;
        ct->bounces++;
    }
    if (depth == maxDepth) ct->depth_exhausted++;
    for (int k = 0; k < 4; ++k) out[k] = clamp01(li[k]);  // PathTracing.h:258
}

// Structs.h:23-41 / Renderer.cpp:12-28 — 144-byte uniforms block.
struct Uniforms {
    int32_t primitiveIndex;
    int32_t _p0[3];
    float cameraPosition[4];
    float screenSize[2];
    float _p1[2];
    float viewportU[4];
    float viewportV[4];
    float firstPixelPosition[4];
    float randomSeed[4];
    uint64_t primitiveCount;
    uint64_t triangleCount;
    uint64_t frameCount;
    uint64_t totalPrimitiveCount;
};
static_assert(sizeof(Uniforms) == 144, "UniformsData must be 144 bytes");

struct RenderParams {
    int32_t rng_mode;       // RNG_LITERAL / RNG_PHILOX
    int32_t bsdf_mode;      // BSDF_LAMBERT / BSDF_SCATTER / BSDF_SCATTER_ALL
    int32_t max_depth;      // reference: 32 (PathTracing.h:216)
    int32_t accumulate;     // 0: reference frame protocol (running mean, Fragment.metal:62-69); 1: sum of clamped samples
    uint32_t sample_begin;  // philox: first sample index; literal: ignored
    uint32_t sample_count;  // samples per pixel rendered by this call (literal frame mode: must be 1)
    uint32_t seed_lo, seed_hi;  // philox key
    int32_t row_begin, row_end; // scanline range [begin,end) (threads / sharding); -1,-1 = all
};

// Fragment.metal:8-72 for one pixel and one sample.
static inline void fragment_sample(const Uniforms& u, int px, int py, uint32_t sample, const RenderParams& rp,
                                   const float* bvh, const float* prims, const float* mats, const int* primIdx,
                                   float out[4], Counters* ct) {
    float W = u.screenSize[0], H = u.screenSize[1];
    float uvx = ((float)px + 0.5f) / W, uvy = ((float)py + 0.5f) / H;  // Vertex.metal:5-17
    PathRng g;
    g.mode = rp.rng_mode;
    float xOff, yOff;
    if (rp.rng_mode == RNG_LITERAL) {
        V3 rs = v3(u.randomSeed[0], u.randomSeed[1], u.randomSeed[2]);
        uint32_t seed = (uint32_t)(sin_hash(uvx, uvy, rs) * (float)((uint32_t)-1));  // Fragment.metal:29
        xOff = (pcg_float(seed) - 0.5f) / W;  // Fragment.metal:31-34
        seed = pcg_hash(seed);
        yOff = (pcg_float(seed) - 0.5f) / H;
        seed = pcg_hash(seed);
        g.seed = seed;
        g.pixel = g.sample = g.key0 = g.key1 = 0;
    } else {
        g.seed = 0;
        g.pixel = (uint32_t)(py * (int)W + px);
        g.sample = sample;
        g.key0 = rp.seed_lo;
        g.key1 = rp.seed_hi;
        uint32_t o[4];
        philox4x32_10(g.pixel, g.sample, 0xFFFFFFFFu, 0u, g.key0, g.key1, o);
        xOff = (u01(o[0]) - 0.5f) / W;
        yOff = (u01(o[1]) - 0.5f) / H;
    }
    V3 first = v3(u.firstPixelPosition[0], u.firstPixelPosition[1], u.firstPixelPosition[2]);
    V3 U = v3(u.viewportU[0], u.viewportU[1], u.viewportU[2]);
    V3 V = v3(u.viewportV[0], u.viewportV[1], u.viewportV[2]);
    V3 cam = v3(u.cameraPosition[0], u.cameraPosition[1], u.cameraPosition[2]);
    V3 rayDir = (first + (uvx + xOff) * U + (uvy + yOff) * V) - cam;  // Fragment.metal:36-40
    Ray r{cam, normalize(rayDir)};
    ct->paths++;
    ray_color(r, bvh, prims, mats, (uint32_t)u.primitiveCount, primIdx, g, rp.max_depth, rp.bsdf_mode, out, ct);
}

}  // namespace orc

// ====================================================================================
// C API (ctypes) — test harness surface
// ====================================================================================
using namespace orc;

extern "C" {

uint32_t orc_pcg_hash(uint32_t s) { return pcg_hash(s); }
float orc_pcg_float(uint32_t s) { return pcg_float(s); }
// Renderer.cpp:30-41 host generator; `state` is the static current_seed (initially 92407235).
uint32_t orc_bitm_random(uint32_t* state) { return (*state = pcg_hash(*state)); }
float orc_host_random_float(uint32_t* state) {
    return (float)orc_bitm_random(state) / (float)std::numeric_limits<uint32_t>::max();
}
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
void orc_sincos_2pi(float u, float* s, float* c) { sincos_2pi(u, s, c); }
float orc_u01(uint32_t x) { return u01(x); }
float orc_parse_obj_real(const char* s) {
    const char* t = s;
    return tinyobj_parse_real(&t);
}

void* orc_scene_new() { return new Scene(); }
void orc_scene_free(void* h) { delete (Scene*)h; }
void orc_scene_clear(void* h) {
    Scene* s = (Scene*)h;
    s->prims.clear();
    s->nodes.clear();
    s->primIdx.clear();
}
int orc_scene_load_xml(void* h, const char* path, const char* asset_root) {
    return load_scene_xml(path, asset_root ? asset_root : "", *(Scene*)h) ? 0 : 1;
}
// type 0: d0 = centre, d1.x = radius; type 1: three vertices.  mat = albedo rgb, type, emission rgb, power.
void orc_scene_add(void* h, int type, const float d0[3], const float d1[3], const float d2[3], const float mat[8]) {
    Primitive p;
    p.type = type;
    p.data0 = v3(d0[0], d0[1], d0[2]);
    p.data1 = v3(d1[0], d1[1], d1[2]);
    p.data2 = v3(d2[0], d2[1], d2[2]);
    p.material.albedo = v3(mat[0], mat[1], mat[2]);
    p.material.materialType = mat[3];
    p.material.emissionColor = v3(mat[4], mat[5], mat[6]);
    p.material.emissionPower = mat[7];
    ((Scene*)h)->prims.push_back(p);
}
void orc_scene_build_bvh(void* h) { build_bvh(*(Scene*)h); }
uint64_t orc_scene_prim_count(void* h) { return ((Scene*)h)->prims.size(); }
uint64_t orc_scene_triangle_count(void* h) {
    uint64_t c = 0;
    for (auto& p : ((Scene*)h)->prims) c += (p.type == 1);
    return c;
}
uint64_t orc_scene_node_count(void* h) { return ((Scene*)h)->nodes.size(); }
const char* orc_scene_log(void* h) { return ((Scene*)h)->log.c_str(); }
// Packers — Scene.h:99-167.  Caller provides the arrays.
void orc_scene_pack_prims(void* h, float* out /* 12*P */) {  // createTransformsBuffer, Scene.h:99-108
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->prims.size(); ++i) {
        const Primitive& p = s->prims[i];
        float* o = out + 12 * i;
        o[0] = p.data0.x; o[1] = p.data0.y; o[2] = p.data0.z; o[3] = (float)p.type;
        o[4] = p.data1.x; o[5] = p.data1.y; o[6] = p.data1.z; o[7] = 0;
        o[8] = p.data2.x; o[9] = p.data2.y; o[10] = p.data2.z; o[11] = 0;
    }
}
void orc_scene_pack_mats(void* h, float* out /* 8*P */) {  // createMaterialsBuffer, Scene.h:110-118
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->prims.size(); ++i) {
        const Material& m = s->prims[i].material;
        float* o = out + 8 * i;
        o[0] = m.albedo.x; o[1] = m.albedo.y; o[2] = m.albedo.z; o[3] = m.materialType;
        o[4] = m.emissionColor.x; o[5] = m.emissionColor.y; o[6] = m.emissionColor.z; o[7] = m.emissionPower;
    }
}
void orc_scene_pack_bvh(void* h, float* out /* 8*N */) {  // createBVHBuffer, Scene.h:151-159
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->nodes.size(); ++i) {
        const BVHNode& n = s->nodes[i];
        float* o = out + 8 * i;
        o[0] = n.bmin.x; o[1] = n.bmin.y; o[2] = n.bmin.z; memcpy(&o[3], &n.leftFirst, 4);
        o[4] = n.bmax.x; o[5] = n.bmax.y; o[6] = n.bmax.z; memcpy(&o[7], &n.count, 4);
    }
}
void orc_scene_pack_prim_idx(void* h, int32_t* out /* P */) {  // createPrimitiveIndexBuffer, Scene.h:161-167
    Scene* s = (Scene*)h;
    for (size_t i = 0; i < s->primIdx.size(); ++i) out[i] = (int32_t)s->primIdx[i];
}

// Renderer.cpp:153-182 (recalculateViewport) for camera pos/forward/up, vfov degrees, W x H.
void orc_viewport(const float pos[3], const float fwd[3], const float up[3], float vfov_deg, float W, float H,
                  Uniforms* u) {
    float aspect = W / H;
    float fovRad = vfov_deg * (M_PI / 180.0f);  // double product rounded to float, as Renderer.cpp:156
    float halfH = tanf(fovRad * 0.5f);
    float halfW = aspect * halfH;
    V3 P = v3(pos[0], pos[1], pos[2]), F = v3(fwd[0], fwd[1], fwd[2]), Up = v3(up[0], up[1], up[2]);
    V3 w = normalize(-F);
    V3 uu = normalize(cross(Up, w));
    V3 vv = cross(w, uu);
    V3 vU = uu * (2.0f * halfW);
    V3 vV = (-vv) * (2.0f * halfH);
    V3 first = P - w - (vU * 0.5f) - (vV * 0.5f);
    u->cameraPosition[0] = P.x; u->cameraPosition[1] = P.y; u->cameraPosition[2] = P.z;
    u->viewportU[0] = vU.x; u->viewportU[1] = vU.y; u->viewportU[2] = vU.z;
    u->viewportV[0] = vV.x; u->viewportV[1] = vV.y; u->viewportV[2] = vV.z;
    u->firstPixelPosition[0] = first.x; u->firstPixelPosition[1] = first.y; u->firstPixelPosition[2] = first.z;
    u->screenSize[0] = W; u->screenSize[1] = H;
}

// Render.  accumulate == 0: ONE reference frame (Fragment.metal:23-71): `last` is the previous
//   accumulation texture (RGBA32F, W*H*4), `cur` receives the new running mean.
// accumulate == 1: `cur` += sum over samples of the per-sample clamped colour (RGBA); `last` ignored.
// Counters (12 x u64) are added to, never cleared.
int orc_render(const Uniforms* u, const RenderParams* rp, const float* bvh, const float* prims, const float* mats,
               const int32_t* primIdx, const float* last, float* cur, uint64_t* counters12) {
    int W = (int)u->screenSize[0], H = (int)u->screenSize[1];
    int y0 = rp->row_begin < 0 ? 0 : rp->row_begin, y1 = rp->row_end < 0 ? H : rp->row_end;
    Counters ct;
    memset(&ct, 0, sizeof ct);
    for (int py = y0; py < y1; ++py) {
        for (int px = 0; px < W; ++px) {
            // coord = uint2(uv * screenSize) (Fragment.metal:62) == (px, py) for W,H < 2^23
            size_t at = 4 * ((size_t)py * W + px);
            if (rp->accumulate == 0) {
                float c[4];
                fragment_sample(*u, px, py, rp->sample_begin, *rp, bvh, prims, mats, primIdx, c, &ct);
                uint64_t fc = u->frameCount + 1;  // Fragment.metal:63
                float lastv[4] = {0, 0, 0, 0};
                if (u->frameCount != 0 && last)  // Fragment.metal:23-27 clears lastFrame when frameCount == 0
                    for (int k = 0; k < 4; ++k) lastv[k] = last[at + k];
                for (int k = 0; k < 4; ++k) {
                    float v = c[k] + lastv[k] * (float)(fc - 1);  // Fragment.metal:65
                    v = v / (float)fc;                            // :66
                    cur[at + k] = clamp01(v);                     // :67
                }
            } else {
                for (uint32_t s = 0; s < rp->sample_count; ++s) {
                    float c[4];
                    fragment_sample(*u, px, py, rp->sample_begin + s, *rp, bvh, prims, mats, primIdx, c, &ct);
                    for (int k = 0; k < 4; ++k) cur[at + k] += c[k];
                }
            }
        }
    }
    if (counters12) {
        const uint64_t* src = (const uint64_t*)&ct;
        for (int k = 0; k < 12; ++k) __atomic_fetch_add(&counters12[k], src[k], __ATOMIC_RELAXED);
    }
    return 0;
}

// Multi-threaded driver for the CPU-baseline leg: `threads` workers pull 4-row bands from an
// atomic cursor and call orc_render on them (row_begin/row_end in *rp are ignored).
int orc_render_mt(const Uniforms* u, const RenderParams* rp, const float* bvh, const float* prims, const float* mats,
                  const int32_t* primIdx, const float* last, float* cur, uint64_t* counters12, int threads) {
    int H = (int)u->screenSize[1];
    if (threads < 1) threads = 1;
    std::atomic<int> cursor(0);
    const int band = 4;
    auto worker = [&]() {
        for (;;) {
            int y0 = cursor.fetch_add(band);
            if (y0 >= H) break;
            RenderParams local = *rp;
            local.row_begin = y0;
            local.row_end = std::min(H, y0 + band);
            orc_render(u, &local, bvh, prims, mats, primIdx, last, cur, counters12);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    return 0;
}

// Single closest-hit query (unit tests of PathTracing.h:75-204 edge cases).
void orc_first_hit(const float o[3], const float d[3], const float* bvh, const float* prims, const int32_t* primIdx,
                   float* t, int32_t* prim, float normal[3], int32_t* frontFace) {
    Counters ct;
    memset(&ct, 0, sizeof ct);
    Ray r{v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2])};
    Hit h = first_hit_bvh(r, bvh, prims, primIdx, &ct);
    *t = h.t;
    *prim = h.primitiveId;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    *frontFace = h.frontFace ? 1 : 0;
}

void orc_histogram_enable(int on) {
    g_hist_on = on != 0;
    if (on) memset(g_hist, 0, sizeof g_hist);
}
void orc_histogram_read(uint64_t* out512) { memcpy(out512, g_hist, sizeof g_hist); }
void orc_set_ray_hook(RayHook hook, void* user) {
    g_ray_hook_user = user;
    g_ray_hook = hook;
}

// FNV-1a 64 over a byte range (image hashes).
uint64_t orc_fnv1a64(const void* data, uint64_t nbytes) {
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 14695981039346656037ull;
    for (uint64_t i = 0; i < nbytes; ++i) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

}  // extern "C"

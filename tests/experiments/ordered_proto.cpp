// ordered_proto.cpp — CPU experiment (test infrastructure, not product): does a closest-first traversal of an OWN
// acceleration structure built over the reference tree's LEAVES return exactly what the reference's unordered
// stack walk (PathTracing.h:75-204, restated in oracle/mpt_oracle.cpp) returns, and what does it cost?
//
// Every closest-hit query of an oracle render (orc_set_ray_hook) is repeated here with the ordered walk and compared
// bit for bit with the oracle's answer.  Printed: mismatches among rays the walk did NOT flag (must be 0), how many
// rays it flagged for an exact re-trace, and the work per ray (node visits, box tests, leaf visits, primitive tests).
//
// build: g++ -O2 -std=c++17 -ffp-contract=off tests/experiments/ordered_proto.cpp -Loracle/_build -lmpt_oracle
//        -Wl,-rpath,$PWD/oracle/_build -lpthread -o /tmp/ordered_proto
// run:   /tmp/ordered_proto assets/scene.xml 1920 1080 8 [k=4] [eps_rel=1e-3] [eps_abs=0]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "../../metalpathtracer_amd/csrc/mpt_accel.h"

extern "C" {
void* orc_scene_new();
int orc_scene_load_xml(void* h, const char* path, const char* asset_root);
void orc_scene_build_bvh(void* h);
uint64_t orc_scene_prim_count(void* h);
uint64_t orc_scene_triangle_count(void* h);
uint64_t orc_scene_node_count(void* h);
void orc_scene_pack_prims(void* h, float* out);
void orc_scene_pack_mats(void* h, float* out);
void orc_scene_pack_bvh(void* h, float* out);
void orc_scene_pack_prim_idx(void* h, int32_t* out);
struct Uniforms {
    int32_t primitiveIndex, _p0[3];
    float cameraPosition[4], screenSize[2], _p1[2], viewportU[4], viewportV[4], firstPixelPosition[4], randomSeed[4];
    uint64_t primitiveCount, triangleCount, frameCount, totalPrimitiveCount;
};
struct RenderParams {
    int32_t rng_mode, bsdf_mode, max_depth, accumulate;
    uint32_t sample_begin, sample_count, seed_lo, seed_hi;
    int32_t row_begin, row_end;
};
void orc_viewport(const float pos[3], const float fwd[3], const float up[3], float vfov_deg, float W, float H, Uniforms* u);
int orc_render_mt(const Uniforms* u, const RenderParams* rp, const float* bvh, const float* prims, const float* mats,
                  const int32_t* primIdx, const float* last, float* cur, uint64_t* counters12, int threads);
typedef void (*RayHook)(const float o[3], const float d[3], float t, int primitiveId, void* user);
void orc_set_ray_hook(RayHook hook, void* user);
}

static inline int f2i(float f) { int i; memcpy(&i, &f, 4); return i; }

using mpt_accel::Box;
using mpt_accel::empty_box;
using mpt_accel::grow;
struct Leaf { Box b; int first, count; bool sphere; };   // a reference leaf, in reference visit (DFS, right first) order

static std::vector<float> g_bvh, g_prims;
static std::vector<int32_t> g_idx;
static std::vector<Leaf> g_leaves;
static std::vector<int> g_always;        // sphere primitives: tested unconditionally, before the tree
static std::vector<Box> g_item_box;      // own box of a leaf: the reference leaf box, or (leaf with a sphere) a tight box over its triangles
static std::vector<int> g_prim_leaf;     // primitive id -> leaf

static int K = 4;
static float EPS_REL = 1e-3f, EPS_ABS = 0.0f;

static void collect_leaves() {
    std::vector<int> st;
    st.push_back(0);
    while (!st.empty()) {
        int n = st.back();
        st.pop_back();
        const float* p = &g_bvh[8 * (size_t)n];
        int lf = f2i(p[3]), cnt = f2i(p[7]);
        if (cnt > 0) {
            Leaf l;
            memcpy(l.b.lo, p, 12);
            memcpy(l.b.hi, p + 4, 12);
            l.first = lf;
            l.count = cnt;
            l.sphere = false;
            for (int i = 0; i < cnt; ++i) {
                int pi = g_idx[lf + i];
                if ((int)g_prims[12 * (size_t)pi + 3] == 0) l.sphere = true;
                g_prim_leaf[pi] = (int)g_leaves.size();
            }
            g_leaves.push_back(l);
        } else {
            st.push_back(lf);
            st.push_back(-cnt);   // popped first
        }
    }
}

static std::vector<float> g_acc;         // emitted device nodes (mpt_accel::emit): MPT_ACCEL_NODE_FLOATS each
static std::vector<uint32_t> g_first_leaf;  // "first" of a leaf ref -> leaf index (the prototype keeps primitives where they are)

struct Stats {
    std::atomic<uint64_t> rays{0}, mismatch_unflagged{0}, mismatch_raw{0}, flagged{0}, f_dir{0}, f_tie{0}, f_check{0},
        node_visits{0}, box_tests{0}, leaf_visits{0}, prim_tests{0}, restarts{0}, stack_hist[33];
    Stats() { for (auto& s : stack_hist) s = 0; }
};
static Stats S;

// the reference's exact slab quantities for one box (PathTracing.h:52-72): lo_b, hi_b with tMax = +inf
static inline void ref_slab(const float o[3], const float d[3], const Box& b, float& lo, float& hi) {
    lo = 0.0001f;
    hi = INFINITY;
    for (int i = 0; i < 3; ++i) {
        float invD = 1.0f / d[i];
        float t0 = (b.lo[i] - o[i]) * invD, t1 = (b.hi[i] - o[i]) * invD;
        if (invD < 0.0f) std::swap(t0, t1);
        lo = fmaxf(lo, t0);
        hi = fminf(hi, t1);
    }
}

struct Trav {
    const float* o;
    const float* d;
    float T;
    int W;
    bool tie;
    uint64_t prim_tests, leaf_visits;
};
static inline void test_prim(Trav& tv, int pi);
static inline void visit_leaf(Trav& tv, int li) {
    const Leaf& l = g_leaves[li];
    tv.leaf_visits++;
    for (int i = 0; i < l.count; ++i) {
        int pi = g_idx[l.first + i];
        if ((int)g_prims[12 * (size_t)pi + 3] == 0) continue;  // spheres are on the always list
        test_prim(tv, pi);
    }
}
static inline void test_prim(Trav& tv, int pi) {
    const float* o = tv.o;
    const float* d = tv.d;
    {
        const float* p = &g_prims[12 * (size_t)pi];
        tv.prim_tests++;
        float tt = INFINITY;
        bool hit = false;
        if ((int)p[3] == 0) {
            float ocx = o[0] - p[0], ocy = o[1] - p[1], ocz = o[2] - p[2];
            float a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            float b = ocx * d[0] + ocy * d[1] + ocz * d[2];
            float c = (ocx * ocx + ocy * ocy + ocz * ocz) - p[4] * p[4];
            float disc = b * b - a * c;
            if (disc > 0.0f) {
                float sq = sqrtf(disc);
                float temp = (-b - sq) / a;
                if (temp > 0.0001f) tt = temp, hit = true;
            }
        } else {
            float e1x = p[4] - p[0], e1y = p[5] - p[1], e1z = p[6] - p[2];
            float e2x = p[8] - p[0], e2y = p[9] - p[1], e2z = p[10] - p[2];
            float hx = d[1] * e2z - d[2] * e2y, hy = d[2] * e2x - d[0] * e2z, hz = d[0] * e2y - d[1] * e2x;
            float a = e1x * hx + e1y * hy + e1z * hz;
            if (fabsf(a) > 1e-5f) {
                float f = 1.0f / a;
                float sx = o[0] - p[0], sy = o[1] - p[1], sz = o[2] - p[2];
                float u = f * (sx * hx + sy * hy + sz * hz);
                if (u >= 0.0f && u <= 1.0f) {
                    float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
                    float v = f * (d[0] * qx + d[1] * qy + d[2] * qz);
                    if (v >= 0.0f && u + v <= 1.0f) {
                        float t = f * (e2x * qx + e2y * qy + e2z * qz);
                        if (t > 0.0001f) tt = t, hit = true;
                    }
                }
            }
        }
        if (!hit) return;
        if (tt < tv.T) {
            tv.T = tt;
            tv.W = pi;
        } else if (tt == tv.T && pi != tv.W) {
            tv.tie = true;
        }
    }
}

static const int STACK = 32;
static void ordered_hit(const float o[3], const float d[3], float& t_out, int& prim_out, int& flag_out) {
    Trav tv{o, d, INFINITY, -1, false, 0, 0};
    flag_out = 0;
    if (d[0] == 0.0f || d[1] == 0.0f || d[2] == 0.0f || d[0] != d[0] || d[1] != d[1] || d[2] != d[2]) {
        flag_out = 1;
        S.f_dir++;
        return;
    }
    for (int pi : g_always) test_prim(tv, pi);
    uint64_t nv = 0, bt = 0;
    int maxsp = 0;
    {   // the device walk (mpt_ordered.h: ot_walk), statement for statement
        const float idx = 1.0f / d[0], idy = 1.0f / d[1], idz = 1.0f / d[2];   // device: v_rcp_f32 (1 ulp)
        const float ox = o[0] * idx, oy = o[1] * idy, oz = o[2] * idz;
        uint32_t stack_key[STACK], stack_parent[STACK];
        int sp = 0;
        uint32_t cur = 0;
        auto lim_of = [&](float T) { return T + (T * EPS_REL + EPS_ABS); };
        auto child_ref = [&](uint32_t n, uint32_t slot) { uint32_t r; memcpy(&r, &g_acc[MPT_ACCEL_NODE_FLOATS * (size_t)n + 24 + slot], 4); return r; };
        auto pop = [&](float lim) -> uint32_t {
            while (sp > 0) {
                --sp;
                uint32_t kb = stack_key[sp] & ~3u;
                float lo;
                memcpy(&lo, &kb, 4);
                if (lo <= lim) return child_ref(stack_parent[sp], stack_key[sp] & 3u);
            }
            return 0xFFFFFFFFu;
        };
        for (;;) {
            while (cur < 0x80000000u) {
                const float* n = &g_acc[MPT_ACCEL_NODE_FLOATS * (size_t)cur];
                nv++;
                const float lim = lim_of(tv.T);
                uint32_t k[4];
                for (int c = 0; c < 4; ++c) {
                    bt++;
                    uint32_t ref = child_ref(cur, c);
                    float t0 = fmaf(n[c], idx, -ox), t1 = fmaf(n[12 + c], idx, -ox);
                    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
                    t0 = fmaf(n[4 + c], idy, -oy), t1 = fmaf(n[16 + c], idy, -oy);
                    tn = fmaxf(tn, fminf(t0, t1)), tf = fminf(tf, fmaxf(t0, t1));
                    t0 = fmaf(n[8 + c], idz, -oz), t1 = fmaf(n[20 + c], idz, -oz);
                    tn = fmaxf(fmaxf(tn, fminf(t0, t1)), 0.0f), tf = fminf(tf, fmaxf(t0, t1));
                    bool hit = ref != 0xFFFFFFFFu && tn <= tf * 1.00000048f && tn <= lim;
                    uint32_t tb;
                    memcpy(&tb, &tn, 4);
                    k[c] = hit ? ((tb & ~3u) | c) : (0x7F800000u | c);
                }
                std::sort(k, k + 4);
                const uint32_t parent = cur;
                if (k[0] < 0x7F800000u) {
                    cur = child_ref(parent, k[0] & 3u);
                    for (int j = 3; j >= 1; --j)
                        if (k[j] < 0x7F800000u) {
                            if (sp >= STACK) { fprintf(stderr, "stack overflow\n"); abort(); }
                            stack_key[sp] = k[j];
                            stack_parent[sp] = parent;
                            sp++;
                        }
                    if (sp > maxsp) maxsp = sp;
                } else {
                    cur = pop(lim);
                }
            }
            if (cur == 0xFFFFFFFFu) break;
            visit_leaf(tv, (int)g_first_leaf[cur & 0x07FFFFFFu]);
            cur = pop(lim_of(tv.T));
        }
    }
done:
    S.node_visits += nv;
    S.box_tests += bt;
    S.leaf_visits += tv.leaf_visits;
    S.prim_tests += tv.prim_tests;
    S.stack_hist[maxsp > 32 ? 32 : maxsp]++;
    t_out = tv.T;
    prim_out = tv.W;
    if (tv.tie) {
        flag_out = 2;
        S.f_tie++;
    }
    if (tv.W >= 0) {  // final check against the winner's REFERENCE leaf: reachable and consistent there
        float lo, hi;
        ref_slab(o, d, g_leaves[g_prim_leaf[tv.W]].b, lo, hi);
        if (!(hi > lo) || !(tv.T >= lo)) {
            if (!flag_out) S.f_check++;
            flag_out |= 4;
        }
    }
}

static void hook(const float o[3], const float d[3], float t, int prim, void*) {
    float t2 = INFINITY;
    int p2 = -1, flag = 0;
    ordered_hit(o, d, t2, p2, flag);
    S.rays++;
    bool same = (p2 == prim) && (prim < 0 || memcmp(&t, &t2, 4) == 0);
    if (flag) {
        S.flagged++;
        if (!same && !(flag & 1)) S.mismatch_raw++;
    } else if (!same) {
        S.mismatch_unflagged++;
        S.mismatch_raw++;
        if (S.mismatch_unflagged < 10)
            fprintf(stderr, "MISMATCH o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) ref t=%.9g prim=%d  ordered t=%.9g prim=%d\n", o[0],
                    o[1], o[2], d[0], d[1], d[2], t, prim, t2, p2);
    }
}

int main(int argc, char** argv) {
    const char* xml = argc > 1 ? argv[1] : "assets/scene.xml";
    int W = argc > 2 ? atoi(argv[2]) : 640, H = argc > 3 ? atoi(argv[3]) : 360, spp = argc > 4 ? atoi(argv[4]) : 4;
    K = argc > 5 ? atoi(argv[5]) : 4;
    EPS_REL = argc > 6 ? (float)atof(argv[6]) : 1e-3f;
    EPS_ABS = argc > 7 ? (float)atof(argv[7]) : 0.0f;
    int depth = argc > 8 ? atoi(argv[8]) : 8;
    int bsdf = argc > 9 ? atoi(argv[9]) : 0;
    void* sc = orc_scene_new();
    if (orc_scene_load_xml(sc, xml, "")) return fprintf(stderr, "cannot load %s\n", xml), 1;
    orc_scene_build_bvh(sc);
    size_t P = orc_scene_prim_count(sc), N = orc_scene_node_count(sc);
    g_bvh.resize(8 * N);
    g_prims.resize(12 * P);
    g_idx.resize(P);
    std::vector<float> mats(8 * P);
    orc_scene_pack_bvh(sc, g_bvh.data());
    orc_scene_pack_prims(sc, g_prims.data());
    orc_scene_pack_mats(sc, mats.data());
    orc_scene_pack_prim_idx(sc, g_idx.data());
    g_prim_leaf.assign(P, -1);
    collect_leaves();
    std::vector<int> items;
    float MATE_PAD = getenv("MATE_PAD") ? (float)atof(getenv("MATE_PAD")) : 0.05f;
    g_item_box.resize(g_leaves.size());
    for (int i = 0; i < (int)g_leaves.size(); ++i) {
        const Leaf& l = g_leaves[i];
        g_item_box[i] = l.b;
        int ntri = 0;
        Box tb = empty_box();
        for (int k = 0; k < l.count; ++k) {
            int pi = g_idx[l.first + k];
            const float* p = &g_prims[12 * (size_t)pi];
            if ((int)p[3] == 0) {
                g_always.push_back(pi);
                continue;
            }
            ntri++;
            for (int v = 0; v < 3; ++v) {
                Box c;
                memcpy(c.lo, p + 4 * v, 12);
                memcpy(c.hi, p + 4 * v, 12);
                grow(tb, c);
            }
        }
        if (l.sphere && ntri) {  // tight box over the sphere's leaf-mates, padded by a fraction of its size
            float ext = std::max(tb.hi[0] - tb.lo[0], std::max(tb.hi[1] - tb.lo[1], tb.hi[2] - tb.lo[2]));
            for (int a = 0; a < 3; ++a) tb.lo[a] -= MATE_PAD * ext, tb.hi[a] += MATE_PAD * ext;
            g_item_box[i] = tb;
            printf("leaf %d holds a sphere and %d triangles: own box (%g %g %g)-(%g %g %g)\n", i, ntri, tb.lo[0], tb.lo[1], tb.lo[2], tb.hi[0], tb.hi[1], tb.hi[2]);
        }
        if (ntri) items.push_back(i);
    }
    // the product's builder (mpt_accel.h) over the leaf boxes, padded as mpt_upload_scene pads them
    float tri_extent = 0.0f;
    for (size_t i = 0; i < P; ++i)
        if ((int)g_prims[12 * i + 3] == 1)
            for (int q = 0; q < 11; ++q)
                if ((q & 3) != 3) tri_extent = std::max(tri_extent, fabsf(g_prims[12 * i + q]));
    const float pad = std::max(tri_extent, 1e-6f) * 6.103515625e-05f;
    if (!getenv("EPS_ABS_OFF") && argc <= 7) EPS_ABS = tri_extent * 3.814697265625e-06f;
    std::vector<mpt_accel::Item> acc_items;
    for (int li : items) {
        mpt_accel::Item it;
        it.box = g_item_box[li];
        for (int a2 = 0; a2 < 3; ++a2) it.box.lo[a2] -= pad, it.box.hi[a2] += pad;
        it.count = (uint32_t)g_leaves[li].count;
        it.key = (uint32_t)li;
        acc_items.push_back(it);
    }
    const mpt_accel::Topology topo = mpt_accel::build_topology(acc_items);
    std::vector<uint32_t> first_of(acc_items.size());
    g_first_leaf.assign(P + 16, 0);
    for (size_t i = 0; i < acc_items.size(); ++i) {
        first_of[i] = (uint32_t)g_leaves[items[i]].first;   // unique per leaf: used as the leaf's handle here
        g_first_leaf[first_of[i]] = (uint32_t)items[i];
    }
    g_acc = mpt_accel::emit(topo, acc_items, first_of);
    {   // every item exactly once in the emitted tree
        std::vector<int> seen(acc_items.size(), 0);
        size_t refs = 0;
        for (size_t n = 0; n < topo.wn.size(); ++n)
            for (int c = 0; c < 4; ++c)
                if (topo.wn[n].child[c] < 0 && topo.wn[n].child[c] != INT32_MIN) seen[~topo.wn[n].child[c]]++, refs++;
        for (int v : seen)
            if (v != 1) { fprintf(stderr, "builder: an item appears %d times\n", v); return 1; }
        if (refs != acc_items.size() || topo.item_order.size() != acc_items.size()) return fprintf(stderr, "builder: item count\n"), 1;
    }
    printf("%s: %zu prims, %zu ref nodes, %zu ref leaves (%zu with spheres), own tree: %zu nodes of width %d\n", xml, P, N,
           g_leaves.size(), g_always.size(), g_acc.size() / MPT_ACCEL_NODE_FLOATS, K);
    Uniforms u;
    memset(&u, 0, sizeof u);
    const float pos[3] = {0, 20, 50}, fwd[3] = {0, 0, -1}, up[3] = {0, 1, 0};
    orc_viewport(pos, fwd, up, 60.0f, (float)W, (float)H, &u);
    u.primitiveCount = P;
    u.triangleCount = orc_scene_triangle_count(sc);
    RenderParams rp = {1, bsdf, depth, 1, 0, (uint32_t)spp, 1, 0, -1, -1};
    std::vector<float> img((size_t)W * H * 4, 0.0f);
    uint64_t ct[12] = {0};
    orc_set_ray_hook(hook, nullptr);
    orc_render_mt(&u, &rp, g_bvh.data(), g_prims.data(), mats.data(), g_idx.data(), nullptr, img.data(), ct, 8);
    double r = (double)S.rays.load();
    printf("rays %.0f  eps_rel %g eps_abs %g\n", r, EPS_REL, EPS_ABS);
    printf("  mismatches among unflagged rays: %llu   (ignoring flags: %llu)\n", (unsigned long long)S.mismatch_unflagged.load(),
           (unsigned long long)S.mismatch_raw.load());
    printf("  flagged for exact re-trace: %llu (%.3g of rays): degenerate dir %llu, tie %llu, final check %llu\n",
           (unsigned long long)S.flagged.load(), S.flagged.load() / r, (unsigned long long)S.f_dir.load(),
           (unsigned long long)S.f_tie.load(), (unsigned long long)S.f_check.load());
    printf("  reference walk per ray: node pops %.2f, box passes %.2f, prim tests %.2f (sphere %.2f tri %.2f)\n", ct[1] / r,
           ct[2] / r, ct[3] / r, ct[4] / r, ct[5] / r);
    printf("  ordered walk per ray:   node visits %.2f, box tests %.2f, leaf visits %.2f, (%zu always spheres) prim tests %.2f\n",
           S.node_visits.load() / r, S.box_tests.load() / r, S.leaf_visits.load() / r, g_always.size(), S.prim_tests.load() / r);
    printf("  max stack depth histogram:");
    for (int i = 0; i <= 32; ++i)
        if (S.stack_hist[i].load()) printf(" %d:%.3g", i, S.stack_hist[i].load() / r);
    printf("\n");
    return 0;
}

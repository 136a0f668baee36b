// Scene.cpp — BVH construction and buffer packing for the host scene layer.
//
// ReferenceSweep reproduces the tree of the reference builder (R/Scene/Scene.h:71-93,195-317) node for node:
// spheres first (stable), then a pre-order recursion that, per node, sorts the index range by data0[axis] for
// each axis (sphere centre / triangle vertex 0 — not the centroid), sweeps prefix and suffix boxes, takes the
// strict minimum of  0.125 + SA_L/SA_P * nL + SA_R/SA_P * nR  over all axes and positions, re-sorts by the
// winning axis and recurses.  std::sort is called on the same ranges with the same key order as the reference,
// so even its (library-specific) tie order is reproduced under libstdc++.
#include "Scene.h"

#include "mpt.h"
#include <string>
#include <stdexcept>
#include <memory>

#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <future>
#include <limits>
#include <thread>

namespace MetalCppPathTracer {

using mpt::float3;
using mpt::float4;

namespace {

struct Box {
    float3 lo{std::numeric_limits<float>::max()};
    float3 hi{-std::numeric_limits<float>::max()};
    void grow(const float3& a, const float3& b) {
        lo = mpt::min3(lo, a);
        hi = mpt::max3(hi, b);
    }
};

inline float boxArea(const float3& lo, const float3& hi) {  // R/Scene/Scene.h:319-322
    float3 d = hi - lo;
    return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x);
}

struct BuildContext {
    const std::vector<Primitive>& prims;
    std::vector<size_t>& order;
    std::vector<BVHNode>& nodes;
    std::vector<float3> lo, hi;      // per-primitive bounds (R/Scene/Scene.h:199-209)
    std::vector<float> key[3];       // data0[axis]
    std::vector<float3> cen;         // centroid of the bounds (binned builder)

    // pad_flat (the binned builder; never the reference's own): no primitive box thinner than 2^-16 of its size / position.
    // The reference's slab test never enters a box of zero thickness (PathTracing.h:68 rejects tMax <= tMin), and small
    // leaves pair up the two coplanar halves of an axis-aligned quad — a Cornell-box wall — where the reference's leaves of
    // eight mix orientations: the wall would be invisible.  (The GPU builders: the same rule, mpt_lbvh.h k_boxes.)
    BuildContext(const std::vector<Primitive>& p, std::vector<size_t>& o, std::vector<BVHNode>& n, bool pad_flat)
        : prims(p), order(o), nodes(n) {
        const size_t count = p.size();
        lo.resize(count);
        hi.resize(count);
        cen.resize(count);
        for (int a = 0; a < 3; ++a) key[a].resize(count);
        for (size_t i = 0; i < count; ++i) {
            const Primitive& q = p[i];
            if (q.type == PrimitiveType::Sphere) {
                float r = q.data1.x;
                lo[i] = q.data0 - float3(r);
                hi[i] = q.data0 + float3(r);
            } else {
                lo[i] = mpt::min3(q.data0, mpt::min3(q.data1, q.data2));
                hi[i] = mpt::max3(q.data0, mpt::max3(q.data1, q.data2));
            }
            if (pad_flat) {
                float l3[3] = {lo[i].x, lo[i].y, lo[i].z}, h3[3] = {hi[i].x, hi[i].y, hi[i].z};
                const float ext = std::max(h3[0] - l3[0], std::max(h3[1] - l3[1], h3[2] - l3[2]));
                for (int a = 0; a < 3; ++a) {
                    const float pad = std::max(std::max(std::fabs(l3[a]), std::fabs(h3[a])), ext) * 1.52587890625e-05f;
                    if (h3[a] - l3[a] < pad) {
                        l3[a] -= pad;
                        h3[a] += pad;
                    }
                }
                lo[i] = float3(l3[0], l3[1], l3[2]);
                hi[i] = float3(h3[0], h3[1], h3[2]);
            }
            cen[i] = (lo[i] + hi[i]) * 0.5f;
            key[0][i] = q.data0.x;
            key[1][i] = q.data0.y;
            key[2][i] = q.data0.z;
        }
    }

    static int emitLeaf(std::vector<BVHNode>& out, size_t start, size_t end, const Box& b) {
        BVHNode n;
        n.boundsMin = b.lo;
        n.boundsMax = b.hi;
        n.leftFirst = static_cast<int>(start);
        n.count = static_cast<int>(end - start);
        out.push_back(n);
        return static_cast<int>(out.size()) - 1;
    }

    // Subtrees are independent once their primitive range is fixed, so the two children of a big node are built
    // concurrently — the left one as a task into its own node list, the right one by the calling thread — and spliced
    // behind the parent in the sequential pre-order (parent, left subtree, right subtree), child indices shifted by
    // the splice offset.  Every decision depends only on the range's content, so the tree is identical to the
    // sequential build (tests/test_host_scene.py); the work inside one node (the chained sorts) stays sequential.
    // `forks` = levels of forking left below this node (0 = build in place).
    template <class BuildFn>
    void buildChildren(std::vector<BVHNode>& out, int self, size_t start, size_t mid, size_t end, int forks, BuildFn fn) {
        int left, right;
        if (forks > 0 && end - start >= 8192) {
            std::vector<BVHNode> l, r;
            auto task = std::async(std::launch::async, [&] { fn(l, start, mid, forks - 1); });
            fn(r, mid, end, forks - 1);
            task.get();
            auto splice = [&out](const std::vector<BVHNode>& sub) {
                const int off = static_cast<int>(out.size());
                for (BVHNode n : sub) {
                    if (n.count <= 0 && sub.size() > 1) {  // internal node: (left child, -right child)
                        n.leftFirst += off;
                        n.count -= off;
                    }
                    out.push_back(n);
                }
                return off;
            };
            left = splice(l);
            right = splice(r);
        } else {
            left = fn(out, start, mid, 0);
            right = fn(out, mid, end, 0);
        }
        out[self].leftFirst = left;
        out[self].count = -right;
    }

    Box rangeBox(size_t start, size_t end) const {
        Box b;
        for (size_t i = start; i < end; ++i) b.grow(lo[order[i]], hi[order[i]]);
        return b;
    }

    void sortRange(size_t start, size_t end, int axis) {
        const std::vector<float>& k = key[axis];
        std::sort(order.begin() + start, order.begin() + end, [&k](size_t a, size_t b) { return k[a] < k[b]; });
    }

    // ---- reference-compatible builder -------------------------------------------------------------------
    int buildSweep(std::vector<BVHNode>& out, size_t start, size_t end, int forks) {
        static thread_local std::vector<Box> prefix, suffix;  // sweep scratch, reused across the nodes of a thread
        const Box bounds = rangeBox(start, end);
        const int self = emitLeaf(out, start, end, bounds);
        const size_t n = end - start;
        if (n <= 8) return self;
        const float parentArea = boxArea(bounds.lo, bounds.hi);
        if (parentArea <= 0.0f) return self;

        float bestCost = std::numeric_limits<float>::max();
        int bestAxis = -1;
        size_t bestSplit = start + n / 2;
        if (prefix.size() < n) {
            prefix.resize(n);
            suffix.resize(n);
        }
        for (int axis = 0; axis < 3; ++axis) {
            sortRange(start, end, axis);
            Box run;
            for (size_t i = 0; i < n; ++i) {
                run.grow(lo[order[start + i]], hi[order[start + i]]);
                prefix[i] = run;
            }
            run = Box();
            for (size_t i = n; i-- > 0;) {
                run.grow(lo[order[start + i]], hi[order[start + i]]);
                suffix[i] = run;
            }
            for (size_t i = 1; i < n; ++i) {
                const float saL = boxArea(prefix[i - 1].lo, prefix[i - 1].hi);
                const float saR = boxArea(suffix[i].lo, suffix[i].hi);
                const size_t nL = i, nR = n - i;
                const float cost = 0.125f + (saL / parentArea) * nL + (saR / parentArea) * nR;
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestSplit = start + i;
                }
            }
        }
        if (bestAxis < 0) return self;
        sortRange(start, end, bestAxis);
        buildChildren(out, self, start, bestSplit, end, forks,
                      [this](std::vector<BVHNode>& o, size_t a, size_t b, int f) { return buildSweep(o, a, b, f); });
        return self;
    }

    // ---- fast builder: 16-bin SAH on centroids (image-equivalent tree, different topology) ----------------
    int buildBinned(std::vector<BVHNode>& out, size_t start, size_t end, int forks) {
        const Box bounds = rangeBox(start, end);
        const int self = emitLeaf(out, start, end, bounds);
        const size_t n = end - start;
        static const size_t leafMax = [] {
            const char* e = std::getenv("MPT_BINNED_LEAF");
            const long v = e ? std::atol(e) : 2;   // bunny x20, closest-first pipeline: 12.15 / 12.26 / 12.28 / 11.16 Grays/s with 4 / 3 / 2 / 1
            return static_cast<size_t>(v < 1 ? 1 : (v > 8 ? 8 : v));
        }();
        if (n <= leafMax) return self;
        Box cb;
        for (size_t i = start; i < end; ++i) cb.grow(cen[order[i]], cen[order[i]]);
        constexpr int BINS = 16;
        float bestCost = std::numeric_limits<float>::max();
        int bestAxis = -1, bestBin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            const float cmin = cb.lo[axis], cmax = cb.hi[axis];
            if (!(cmax > cmin)) continue;
            const float scale = BINS / (cmax - cmin);
            Box bb[BINS];
            size_t bc[BINS] = {0};
            for (size_t i = start; i < end; ++i) {
                const size_t p = order[i];
                int b = static_cast<int>((cen[p][axis] - cmin) * scale);
                b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                bb[b].grow(lo[p], hi[p]);
                bc[b]++;
            }
            float areaL[BINS], areaR[BINS];
            size_t cntL[BINS], cntR[BINS];
            Box run;
            size_t c = 0;
            for (int b = 0; b < BINS; ++b) {
                if (bc[b]) run.grow(bb[b].lo, bb[b].hi);
                c += bc[b];
                cntL[b] = c;
                areaL[b] = c ? boxArea(run.lo, run.hi) : 0.0f;
            }
            run = Box();
            c = 0;
            for (int b = BINS - 1; b >= 0; --b) {
                if (bc[b]) run.grow(bb[b].lo, bb[b].hi);
                c += bc[b];
                cntR[b] = c;
                areaR[b] = c ? boxArea(run.lo, run.hi) : 0.0f;
            }
            for (int b = 0; b + 1 < BINS; ++b) {
                if (cntL[b] == 0 || cntR[b + 1] == 0) continue;
                const float cost = areaL[b] * cntL[b] + areaR[b + 1] * cntR[b + 1];
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = axis;
                    bestBin = b;
                }
            }
        }
        size_t mid;
        if (bestAxis < 0) {
            if (n <= 2 * leafMax) return self;
            mid = start + n / 2;  // all centroids coincide: split the list
        } else {
            const float leafCost = boxArea(bounds.lo, bounds.hi) * n;
            if (n <= 2 * leafMax && bestCost >= leafCost) return self;
            const float cmin = cb.lo[bestAxis], scale = BINS / (cb.hi[bestAxis] - cmin);
            auto it = std::partition(order.begin() + start, order.begin() + end, [&](size_t p) {
                int b = static_cast<int>((cen[p][bestAxis] - cmin) * scale);
                b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                return b <= bestBin;
            });
            mid = static_cast<size_t>(it - order.begin());
            if (mid == start || mid == end) mid = start + n / 2;
        }
        buildChildren(out, self, start, mid, end, forks,
                      [this](std::vector<BVHNode>& o, size_t a, size_t b, int f) { return buildBinned(o, a, b, f); });
        return self;
    }
};

}  // namespace

void Scene::clear() {
    primitives_.clear();
    nodes_.clear();
    primitiveIndices_.clear();
}

size_t Scene::addPrimitive(const Primitive& p) {
    primitives_.push_back(p);
    return primitives_.size() - 1;
}

size_t Scene::getSphereCount() const {
    size_t c = 0;
    for (const Primitive& p : primitives_) c += (p.type == PrimitiveType::Sphere);
    return c;
}

size_t Scene::getTriangleCount() const {
    size_t c = 0;
    for (const Primitive& p : primitives_) c += (p.type == PrimitiveType::Triangle);
    return c;
}

void Scene::buildBVH() { buildBVH(BuildMode::ReferenceSweep); }

void Scene::sortPrimitives() {
    // spheres before triangles, original order kept inside each class: primitive ids are the positions
    // after this sort (R/Scene/Scene.h:72-75)
    auto by_type = [](const Primitive& a, const Primitive& b) { return static_cast<int>(a.type) < static_cast<int>(b.type); };
    if (std::is_sorted(primitives_.begin(), primitives_.end(), by_type)) return;   // (a tree built over this order stays valid)
    std::stable_sort(primitives_.begin(), primitives_.end(), by_type);
    dropBVH();   // a host tree indexes the old order
}

void Scene::dropBVH() {
    nodes_.clear();
    primitiveIndices_.clear();
}

void Scene::adoptBVH(const float* bvh, size_t nodeCount, const int32_t* primIdx) {
    nodes_.resize(nodeCount);
    for (size_t i = 0; i < nodeCount; ++i) {
        const float* q = bvh + 8 * i;
        nodes_[i].boundsMin = float3(q[0], q[1], q[2]);
        nodes_[i].boundsMax = float3(q[4], q[5], q[6]);
        std::memcpy(&nodes_[i].leftFirst, q + 3, 4);
        std::memcpy(&nodes_[i].count, q + 7, 4);
    }
    primitiveIndices_.resize(primitives_.size());
    for (size_t i = 0; i < primitiveIndices_.size(); ++i) primitiveIndices_[i] = static_cast<size_t>(primIdx[i]);
}

void Scene::buildBVH(BuildMode mode) {
    sortPrimitives();
    primitiveIndices_.resize(primitives_.size());
    for (size_t i = 0; i < primitiveIndices_.size(); ++i) primitiveIndices_[i] = i;
    nodes_.clear();
    if (primitives_.empty()) {
        // the reference emits a single empty leaf (count 0) for an empty scene; keep that shape
        BVHNode n;
        n.boundsMin = float3(std::numeric_limits<float>::max());
        n.boundsMax = float3(-std::numeric_limits<float>::max());
        nodes_.push_back(n);
        return;
    }
    if (mode == BuildMode::GpuLbvh) {
        buildOnGpu();
        return;
    }
    BuildContext ctx(primitives_, primitiveIndices_, nodes_, mode != BuildMode::ReferenceSweep);
    // levels of forking: 2^forks concurrent subtree tasks at most (MPT_BUILD_THREADS=1 builds sequentially)
    unsigned threads = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("MPT_BUILD_THREADS")) threads = static_cast<unsigned>(std::max(1, std::atoi(e)));
    int forks = 0;
    while ((1u << forks) < threads && forks < 6) ++forks;
    if (mode == BuildMode::ReferenceSweep)
        ctx.buildSweep(nodes_, 0, primitives_.size(), forks);
    else
        ctx.buildBinned(nodes_, 0, primitives_.size(), forks);
}

// The GPU builder (include/mpt.h: mpt_build_bvh) takes the packed primitive array and returns the reference's two
// buffers; they are unpacked into nodes_ / primitiveIndices_ so that every accessor and packer works as after a host build.
void Scene::buildOnGpu() {
    int device = 0;
    if (const char* e = std::getenv("MPT_BUILD_DEVICE")) device = std::atoi(e);
    mpt_ctx* ctx = nullptr;
    int rc = mpt_create(device, &ctx);
    if (rc != MPT_OK)
        throw std::runtime_error(std::string("Scene::buildBVH(GpuLbvh): mpt_create: ") + mpt_status_string(rc) +
                                 " — a MI355X GPU is required, there is no CPU fallback");
    const size_t n = primitives_.size();
    std::unique_ptr<float4[]> packed(createTransformsBuffer());
    std::vector<float> bvh((2 * n - 1) * 8);
    std::vector<int32_t> idx(n);
    uint64_t nn = 0;
    double ms = 0.0;
    rc = mpt_build_bvh(ctx, reinterpret_cast<const float*>(packed.get()), n, bvh.data(), 2 * n - 1, &nn, idx.data(), &ms);
    std::string err = rc ? mpt_last_error(ctx) : "";
    mpt_destroy(ctx);
    if (rc != MPT_OK) throw std::runtime_error("Scene::buildBVH(GpuLbvh): " + err);
    lastGpuBuildMs_ = ms;
    adoptBVH(bvh.data(), nn, idx.data());
}

int Scene::getBVHDepth() const {
    if (nodes_.empty()) return 0;
    int best = 0;
    std::vector<std::pair<int, int>> st{{0, 1}};
    while (!st.empty()) {
        auto [n, d] = st.back();
        st.pop_back();
        best = std::max(best, d);
        if (nodes_[n].count <= 0 && nodes_.size() > 1) {
            st.push_back({nodes_[n].leftFirst, d + 1});
            st.push_back({-nodes_[n].count, d + 1});
        }
    }
    return best;
}

float4* Scene::createTransformsBuffer() const {
    float4* out = new float4[primitives_.size() * 3];
    for (size_t i = 0; i < primitives_.size(); ++i) {
        const Primitive& p = primitives_[i];
        out[3 * i + 0] = float4(p.data0, static_cast<float>(static_cast<int>(p.type)));
        out[3 * i + 1] = float4(p.data1, 0.0f);
        out[3 * i + 2] = float4(p.data2, 0.0f);
    }
    return out;
}

float4* Scene::createMaterialsBuffer() const {
    float4* out = new float4[primitives_.size() * 2];
    for (size_t i = 0; i < primitives_.size(); ++i) {
        const Material& m = primitives_[i].material;
        out[2 * i + 0] = float4(m.albedo, m.materialType);
        out[2 * i + 1] = float4(m.emissionColor, m.emissionPower);
    }
    return out;
}

float4* Scene::createSphereBuffer() const {
    float4* out = new float4[getSphereCount()];
    size_t k = 0;
    for (const Primitive& p : primitives_)
        if (p.type == PrimitiveType::Sphere) out[k++] = float4(p.data0, p.data1.x);
    return out;
}

float4* Scene::createSphereMaterialsBuffer() const {
    float4* out = new float4[getSphereCount() * 2];
    size_t k = 0;
    for (const Primitive& p : primitives_) {
        if (p.type != PrimitiveType::Sphere) continue;
        out[2 * k + 0] = float4(p.material.albedo, p.material.materialType);
        out[2 * k + 1] = float4(p.material.emissionColor, p.material.emissionPower);
        ++k;
    }
    return out;
}

float4* Scene::createBVHBuffer() const {
    float4* out = new float4[nodes_.size() * 2];
    for (size_t i = 0; i < nodes_.size(); ++i) {
        const BVHNode& n = nodes_[i];
        float a, b;
        std::memcpy(&a, &n.leftFirst, 4);  // the two ints travel as float bit patterns in .w
        std::memcpy(&b, &n.count, 4);
        out[2 * i + 0] = float4(n.boundsMin, a);
        out[2 * i + 1] = float4(n.boundsMax, b);
    }
    return out;
}

int* Scene::createPrimitiveIndexBuffer() const {
    int* out = new int[primitiveIndices_.size()];
    for (size_t i = 0; i < primitiveIndices_.size(); ++i) out[i] = static_cast<int>(primitiveIndices_[i]);
    return out;
}

void Scene::createTriangleBuffers(std::vector<mpt::float3>& outVertices, std::vector<mpt::uint3>& outIndices) const {
    outVertices.clear();
    outIndices.clear();
    uint32_t base = 0;
    for (const Primitive& p : primitives_) {
        if (p.type != PrimitiveType::Triangle) continue;
        outVertices.push_back(p.data0);
        outVertices.push_back(p.data1);
        outVertices.push_back(p.data2);
        mpt::uint3 t;
        t.x = base;
        t.y = base + 1;
        t.z = base + 2;
        outIndices.push_back(t);
        base += 3;
    }
}

}  // namespace MetalCppPathTracer

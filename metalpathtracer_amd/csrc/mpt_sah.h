// mpt_sah.h — top-down binned SAH on the device, shared by the two places that build a binary tree this way: the own 4-wide
// tree of the closest-first pipeline (mpt_devbuild.h: items = the leaves) and the binary tree under the reference-format
// arrays (mpt_lbvh.h build_radix, builder "sah": items = the primitives).  It is the host builder's algorithm
// (mpt_accel.h / Scene::buildBVH(BVH_BINNED_CENTROID): 16 bins over the box centres, cost = area * primitives) run level by
// level: a WORKGROUP per node for the few long tasks at the top, a WAVE per node in the middle, and one lane per item — eight
// nodes to a wave, whole sub-tree in one go — for the nodes of at most MPT_SAH_SMALL items, which are most of the tree.
// Stands where the reference has Scene::buildBVHRecursive (R/Scene/Scene.h:195-317: full-sweep SAH over std::sort).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <utility>
#include <chrono>
#include <vector>

namespace mpt_lbvh {
// Scratch memory of a build: a bump allocator over a few large chunks (a build makes ~90 arrays; one hipMalloc each cost a
// third of the 1 M-primitive build), freed — or handed back to the pool they came from — on every exit path.  A ScratchPool
// (one per context) keeps the chunks between builds: a rebuild allocates nothing.
struct ScratchPool {
    std::vector<std::pair<void*, size_t>> chunks;
    size_t keep_bytes = (size_t)2 << 30;   // what stays allocated between builds at most
    uint32_t* pin = nullptr;               // 2 KiB of pinned host memory for the builders' read-backs (hipHostMalloc per build: ~0.1 ms each)
    hipStream_t side = nullptr;            // a second stream for the chains of a build that need nothing of one another (mpt_devbuild.h) ...
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // ... and the events that fork and join it
    ~ScratchPool() { release(); }
    void release() {
        for (auto& c : chunks) hipFree(c.first);
        chunks.clear();
        if (pin) hipHostFree(pin);
        pin = nullptr;
        if (side) hipStreamDestroy(side);
        side = nullptr;
        for (auto& e : ev) {
            if (e) hipEventDestroy(e);
            e = nullptr;
        }
    }
    hipError_t side_stream(hipStream_t* out) {
        if (!side) {
            hipError_t e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
            if (e != hipSuccess) return e;
            for (auto& v : ev)
                if ((e = hipEventCreateWithFlags(&v, hipEventDisableTiming)) != hipSuccess) return e;
        }
        *out = side;
        return hipSuccess;
    }
};
// 1 KiB of pinned host memory for one builder's read-backs: part `which` (0 / 1) of the pool's block, or the builder's own for the call
struct PinnedWords {
    uint32_t* p = nullptr;
    bool own = false;
    PinnedWords() = default;
    PinnedWords(const PinnedWords&) = delete;
    PinnedWords& operator=(const PinnedWords&) = delete;
    ~PinnedWords() { if (own && p) hipHostFree(p); }
    hipError_t get(ScratchPool* pool, int which) {
        if (pool) {
            if (!pool->pin) {
                const hipError_t e = hipHostMalloc((void**)&pool->pin, 2048, hipHostMallocDefault);
                if (e != hipSuccess) return e;
            }
            p = pool->pin + 256 * which;
            return hipSuccess;
        }
        own = true;
        return hipHostMalloc((void**)&p, 1024, hipHostMallocDefault);
    }
};
struct Scratch {
    ScratchPool* pool;
    std::vector<std::pair<void*, size_t>> mine;
    size_t off = 0;   // in the last chunk of mine
    explicit Scratch(ScratchPool* p = nullptr) : pool(p) {}
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() {
        size_t kept = 0;
        if (pool)
            for (auto& c : pool->chunks) kept += c.second;
        for (auto& c : mine) {
            if (pool && kept + c.second <= pool->keep_bytes) {
                pool->chunks.push_back(c);
                kept += c.second;
            } else {
                hipFree(c.first);
            }
        }
    }
    // a chunk of at least `bytes` becomes the current one: the pool's smallest that fits, else a new allocation
    hipError_t chunk(size_t bytes) {
        if (pool) {
            int best = -1;
            for (int i = 0; i < (int)pool->chunks.size(); ++i)
                if (pool->chunks[i].second >= bytes && (best < 0 || pool->chunks[i].second < pool->chunks[best].second)) best = i;
            if (best >= 0) {
                mine.push_back(pool->chunks[best]);
                pool->chunks.erase(pool->chunks.begin() + best);
                off = 0;
                return hipSuccess;
            }
        }
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, bytes);
        if (e != hipSuccess) return e;
        mine.emplace_back(q, bytes);
        off = 0;
        return hipSuccess;
    }
    // (call once with an estimate of everything the build will ask for: one chunk instead of several)
    hipError_t reserve(size_t bytes) { return chunk(std::max<size_t>(bytes, (size_t)1 << 20)); }
    template <class T>
    hipError_t alloc(T** p, size_t count) {
        const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
        if (mine.empty() || off + bytes > mine.back().second) {
            *p = nullptr;
            hipError_t e = chunk(std::max<size_t>(bytes, (size_t)64 << 20));
            if (e != hipSuccess) return e;
        }
        *p = (T*)((char*)mine.back().first + off);
        off += bytes;
        return hipSuccess;
    }
};

#define MPT_LB(call)                       \
    do {                                   \
        hipError_t e_ = (call);            \
        if (e_ != hipSuccess) return e_;   \
    } while (0)
}  // namespace mpt_lbvh

namespace mpt_sah {
using mpt_lbvh::Scratch;

// Item record, 32 bytes, moved along by every partition (the passes stream it, nothing is looked up through an index):
//   lo = (box min, bits(id))   hi = (box max, bits(primitives in the item))
// Node ids: an item's id (< top) is a leaf; top + k is inner node k, whose children are child[k] and whose box lo[k], hi[k]
// (lo[k].w = bits(items below it)).  Node 0 is the root when there is more than one item.
struct SahTask {
    uint32_t b, e;     // items [b, e) of the current index array
    int parent;        // inner node that waits for this sub-tree (-1: the root)
    uint32_t side;
    uint32_t node;     // the inner node this task makes.  A sub-tree of m items has m - 1 inner nodes, numbered in pre-order from
                       // here: the left child (nlft items) is node + 1, the right one node + nlft — no counter to contend for,
                       // and the same tree gets the same numbers on every run
};
#ifndef MPT_SAH_BIG
#define MPT_SAH_BIG 2048u   // tasks of at least this many items get a whole workgroup (k_sah_level_big), smaller ones a wave
#endif
#ifndef MPT_SAH_SAMPLE
#define MPT_SAH_SAMPLE 256u  // a wave's task bins an evenly spaced sample of about this many of its items
#endif
#ifndef MPT_SAH_SMALL_EXACT
#define MPT_SAH_SMALL_EXACT 1   // nodes of <= MPT_SAH_SMALL items: sweep over the sorted box centres instead of 16 bins
#endif
#ifndef MPT_SAH_SMALL
#define MPT_SAH_SMALL 8u     // tasks of at most this many items (8 or 16) are FINISHED, sub-tree and all, by as many lanes (k_sah_small)
#endif
struct SahState {
    // Task counts by level, three sets in rotation: the kernels of level L read cnt[L % 3] (what level L - 1 pushed), push into
    // cnt[(L + 1) % 3], and the level's first kernel zeroes cnt[(L + 2) % 3] for the level after (sah_level_prologue) — nobody else
    // touches that set during level L, so no kernel of its own is needed between two levels (round 4 had one: k_sah_flip, 23 launches).
    // (a set per 64-byte line: every wave of a level READS its set, and the line the pushes' atomics hammer must not be the same one — in
    //  one line the levels of 30 k .. 130 k tasks took twice as long, 114 / 193 / 82 -> 202 / 397 / 155 us)
    uint32_t cnt[3][48];  // [set][16 * kind]: kind 0 = mid tasks (a wave each), 1 = big tasks (a workgroup per chunk), 2 = small tasks (a lane per item);
                          // a line per counter as well: the three push counters of a level are bumped by different waves at the same time
    uint32_t n_nodes;     // inner nodes created
    int root;             // -1: no items
    uint32_t n_items;
};
__device__ __forceinline__ float half_area4(float4 lo, float4 hi) {
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ bool empty4(float4 lo, float4 hi) { return !(hi.x >= lo.x && hi.y >= lo.y && hi.z >= lo.z); }
__device__ __forceinline__ float axis_of(float4 v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
__device__ __forceinline__ int f2o(float f) {  // order-preserving float -> int
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }
// (bits of the box's half area) << 32 | item index: the maximum over a node's items names the one with the largest box
__device__ __forceinline__ unsigned long long sah_area_key(float4 l, float4 h, uint32_t i) {
    const float a = half_area4(l, h);
    return ((unsigned long long)(a > 0.0f && a < INFINITY ? __float_as_uint(a) : 0u) << 32) | i;
}

__global__ void k_sah_init(const uint32_t* count /* device word, or null */, uint32_t count_host, SahState* st, SahTask* tasks, SahTask* big_tasks,
                           SahTask* small_tasks) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t m = count ? *count : count_host;
    st->n_items = m;
    st->n_nodes = m != 0u ? m - 1u : 0u;
    st->root = -1;
    for (int q = 0; q < 3 * 48; ++q) (&st->cnt[0][0])[q] = 0u;
    if (m >= MPT_SAH_BIG) {
        big_tasks[0] = SahTask{0u, m, -1, 0u, 0u};
        st->cnt[0][16] = 1u;
    } else if (m >= 2u && m <= MPT_SAH_SMALL) {
        small_tasks[0] = SahTask{0u, m, -1, 0u, 0u};
        st->cnt[0][32] = 1u;
    } else if (m != 0u) {
        tasks[0] = SahTask{0u, m, -1, 0u, 0u};
        st->cnt[0][0] = 1u;
    }
}
// Start of a level, ONE thread of the level's first kernel: the set of counts for the level after is zeroed, and the level's own counts go
// to a slot of pinned host memory with a stamp behind them — the host reads them one or two levels LATER, to size the grids of the levels
// it enqueues while the device is busy with this one (run_sah).  The kernels of a level read their counts from the device: their grids
// are sized from an upper bound, the host does not wait for them.
// cover: the task counts the host sized this level's grids for (upper bounds, run_sah).  Counts beyond them would be dropped without a
// trace and leave a malformed tree: the stamp then carries MPT_SAH_STAMP_OVERFLOW and the host fails the build.
#define MPT_SAH_STAMP_OVERFLOW 0x8000u
struct SahLevel {
    uint32_t level;
    uint32_t stamp;                 // 0: this kernel is not the level's first
    unsigned long long* host_slot;  // three words of pinned host memory, each (stamp << 32 | count)
    uint32_t cover[3];
};
// (every word of the slot carries the stamp itself and is ONE 8-byte store, so nothing has to order the words: no system-scope fence —
//  a write-back of the whole L2 — in a kernel whose other workgroups are at work)
__device__ __forceinline__ void sah_level_prologue(SahState* st, const SahLevel& lv) {
    if (lv.stamp == 0u || threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t* c = st->cnt[lv.level % 3u];
    uint32_t* z = st->cnt[(lv.level + 2u) % 3u];
    const uint32_t a = c[0], b = c[16], d = c[32];
    unsigned long long stamp = lv.stamp;
    if (a > lv.cover[0] || b > lv.cover[1] || d > lv.cover[2]) stamp |= MPT_SAH_STAMP_OVERFLOW;
    z[0] = z[16] = z[32] = 0u;
    __hip_atomic_store(lv.host_slot + 0, stamp << 32 | a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(lv.host_slot + 1, stamp << 32 | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(lv.host_slot + 2, stamp << 32 | d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#ifndef MPT_SAH_WAVES
#define MPT_SAH_WAVES 16   // tasks (waves) per workgroup
#endif
// Where a node's items start in the final item order (the order the partitions leave: left sub-tree first) — known the moment the node is
// made, or an item is left alone: inner[k] for inner node k, item[id] for an item.  Either may be null.  (mpt_lbvh.h numbers the nodes of
// the reference-format tree by these; until round 5 a kernel of its own walked from every node to the root to find them, 105 us for 1 M.)
struct SahFirst {
    uint32_t *inner, *item;
};
__device__ __forceinline__ void sah_attach(SahState* st, int2* s_child, int parent, uint32_t side, int id) {
    if (parent < 0) st->root = id;
    else if (side == 0u) s_child[parent].x = id;
    else s_child[parent].y = id;
}
// a finished split: a single item is attached at once, anything larger becomes a task of the next level.  (Called by SEVERAL
// lanes of a wave at once where it matters: the counters are uniform addresses, so the compiler turns each atomicAdd into one
// atomic per wave — 150 k tasks a level each bumping the same three words one by one was 70 % of the builder's time.)
__device__ __forceinline__ void sah_push_children(uint32_t* push /* st->cnt[(level + 1) % 3] */, int2* s_child, SahTask* next, SahTask* next_big, SahTask* next_small, uint32_t b,
                                                  uint32_t e, uint32_t nlft, uint32_t k, int one_left, int one_right, const SahFirst& fs) {
    const uint32_t mid = b + nlft, nrgt = e - mid;
    const SahTask L = SahTask{b, mid, (int)k, 0u, k + 1u}, R = SahTask{mid, e, (int)k, 1u, k + nlft};
    if (nlft == 1u) {
        s_child[k].x = one_left;
        if (fs.item) fs.item[one_left] = b;
    } else if (nlft >= MPT_SAH_BIG) next_big[atomicAdd(&push[16], 1u)] = L;
    else if (nlft <= MPT_SAH_SMALL) next_small[atomicAdd(&push[32], 1u)] = L;
    else next[atomicAdd(&push[0], 1u)] = L;
    if (nrgt == 1u) {
        s_child[k].y = one_right;
        if (fs.item) fs.item[one_right] = e - 1u;
    } else if (nrgt >= MPT_SAH_BIG) next_big[atomicAdd(&push[16], 1u)] = R;
    else if (nrgt <= MPT_SAH_SMALL) next_small[atomicAdd(&push[32], 1u)] = R;
    else next[atomicAdd(&push[0], 1u)] = R;
}
// (a block role of k_sah_tasks below; `block`: the block's number among the level's mid blocks)
__device__ __forceinline__ void sah_mid_block(uint32_t block, int n, const float4* in_lo, const float4* in_hi, float4* out_lo, float4* out_hi, const SahTask* tasks,
                                              SahTask* next, SahTask* next_big, SahTask* next_small, SahState* st, int2* s_child, float4* s_lo, float4* s_hi,
                                              uint32_t level, const SahFirst& fs) {
    const uint32_t n_tasks = st->cnt[level % 3u][0];   // (the grid is sized from an upper bound: waves beyond the count hold an empty task)
    __shared__ int bins[MPT_SAH_WAVES][3][16][7];   // per wave: (lo xyz, hi xyz as ordered ints, primitive count) per axis and bin
    __shared__ uint32_t s_push[MPT_SAH_WAVES][6];   // per wave: b, e, nlft, node, first item left / right (e = 0: nothing to push)
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t t = block * MPT_SAH_WAVES + wv;
    const int TOP = 2 * n - 1;
    SahTask task = SahTask{0u, 0u, -1, 0u, 0u};
    if (t < n_tasks) task = tasks[t];
    const uint32_t b = task.b, e = task.e, m = e - b;
    if (lane == 0) s_push[wv][1] = 0u;
    if (m == 1u && lane == 0) {   // (only the root task of a one-leaf tree: children of one item are attached when they are split off)
        const int id = __float_as_int(in_lo[b].w);
        sah_attach(st, s_child, task.parent, task.side, id);
        if (fs.item) fs.item[id] = b;
    }
    if (m >= 2u) {
    // pass 1: the node's box and the bounds of the box centres
    float nl[3] = {INFINITY, INFINITY, INFINITY}, nh[3] = {-INFINITY, -INFINITY, -INFINITY}, cl[3] = {INFINITY, INFINITY, INFINITY},
          ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned long long largest = 0ull;   // (bits of the area) << 32 | item: the item with the largest box (see the sample below)
    // (four items a lane per trip, all eight loads in flight before the first is used: a task of 2000 items is 32 dependent round trips
    //  otherwise, and the levels whose few hundred tasks are all of that size took 62-68 us each with the chip nearly idle)
    for (uint32_t i0 = b + lane; i0 < e; i0 += 256u) {
        float4 l4[4], h4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            if (i < e) {
                l4[u] = in_lo[i];
                h4[u] = in_hi[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            if (i >= e) break;
            const float4 l = l4[u], h = h4[u];
            const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
            for (int a = 0; a < 3; ++a) {
                nl[a] = fminf(nl[a], lo3[a]);
                nh[a] = fmaxf(nh[a], hi3[a]);
                const float c = 0.5f * (lo3[a] + hi3[a]);
                cl[a] = fminf(cl[a], c);
                ch[a] = fmaxf(ch[a], c);
            }
            const unsigned long long key = sah_area_key(l, h, i);
            largest = key > largest ? key : largest;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(largest, off);
        largest = o > largest ? o : largest;
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; ++a) {
            nl[a] = fminf(nl[a], __shfl_xor(nl[a], off));
            nh[a] = fmaxf(nh[a], __shfl_xor(nh[a], off));
            cl[a] = fminf(cl[a], __shfl_xor(cl[a], off));
            ch[a] = fmaxf(ch[a], __shfl_xor(ch[a], off));
        }
    const uint32_t k = task.node;
    if (lane == 0) {
        s_lo[k] = make_float4(nl[0], nl[1], nl[2], __int_as_float((int)m));   // (.w: items below the node)
        if (fs.inner) fs.inner[k] = b;
        s_hi[k] = make_float4(nh[0], nh[1], nh[2], 0.0f);
        sah_attach(st, s_child, task.parent, task.side, TOP + (int)k);
    }
    // pass 2: bins
    for (uint32_t q = lane; q < 3u * 16u * 7u; q += 64u) {
        const uint32_t f = q % 7u;
        (&bins[wv][0][0][0])[q] = f < 3u ? 0x7FFFFFFF : f < 6u ? (int)0x80000000 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    float inv[3];
    for (int a = 0; a < 3; ++a) {
        const float ext = ch[a] - cl[a];
        inv[a] = ext > 0.0f && isfinite(ext) ? 16.0f / ext : 0.0f;
    }
    // (a node of more than 256 items is binned from an evenly spaced sample of ~256 of them: the 21 LDS atomics per item are
    //  what this pass costs, and 256 boxes choose among 45 planes as well as 100,000 do; box and partition stay exact)
    // ... plus THE item with the largest box when the sample missed it: a ground sphere of radius 1e4 among 5000 triangles
    // that no plane is priced against stays with half of them, and every ray walks through its box a level longer)
    const uint32_t step = m > MPT_SAH_SAMPLE ? m / MPT_SAH_SAMPLE : 1u;
    auto bin_item = [&](uint32_t i) {
        const float4 l = in_lo[i], h = in_hi[i];
        const int cnt = __float_as_int(h.w);
        const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
        for (int a = 0; a < 3; ++a) {
            if (inv[a] == 0.0f) continue;
            int q = (int)((0.5f * (lo3[a] + hi3[a]) - cl[a]) * inv[a]);
            q = q < 0 ? 0 : (q > 15 ? 15 : q);
            int* B = bins[wv][a][q];
            atomicMin(&B[0], f2o(l.x)); atomicMin(&B[1], f2o(l.y)); atomicMin(&B[2], f2o(l.z));
            atomicMax(&B[3], f2o(h.x)); atomicMax(&B[4], f2o(h.y)); atomicMax(&B[5], f2o(h.z));
            atomicAdd(&B[6], cnt);
        }
    };
    for (uint32_t i = b + lane * step; i < e; i += 64u * step) bin_item(i);
    if (lane == 0 && step > 1u && ((uint32_t)largest - b) % step != 0u) bin_item((uint32_t)largest);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // pass 3: lane = (axis, split after bin s): cost = area(L) * count(L) + area(R) * count(R)
    float cost = INFINITY;
    if (lane < 45u) {
        const int a = (int)(lane / 15u), sp = (int)(lane % 15u);
        if (inv[a] != 0.0f) {
            float L[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, R[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int cL = 0, cR = 0;
            for (int q = 0; q < 16; ++q) {
                const int* B = bins[wv][a][q];
                if (B[6] == 0) continue;
                float* D = q <= sp ? L : R;
                for (int c = 0; c < 3; ++c) {
                    D[c] = fminf(D[c], o2f(B[c]));
                    D[3 + c] = fmaxf(D[3 + c], o2f(B[3 + c]));
                }
                if (q <= sp) cL += B[6];
                else cR += B[6];
            }
            if (cL != 0 && cR != 0)
                cost = half_area4(make_float4(L[0], L[1], L[2], 0), make_float4(L[3], L[4], L[5], 0)) * (float)cL +
                       half_area4(make_float4(R[0], R[1], R[2], 0), make_float4(R[3], R[4], R[5], 0)) * (float)cR;
        }
    }
    float best = cost;
    for (int off = 32; off > 0; off >>= 1) best = fminf(best, __shfl_xor(best, off));
    const unsigned long long who = __ballot(cost == best && best < INFINITY);
    const int pick = who != 0ull ? (int)__ffsll((long long)who) - 1 : -1;   // ties: the lowest (axis, split)
    const int paxis = pick >= 0 ? pick / 15 : 0, psplit = pick >= 0 ? pick % 15 : 0;
    // partition into the other array: left from b upwards, right from e - 1 downwards
    uint32_t nlft = 0, nrgt = 0;
    int one_left = 0, one_right = 0;   // (the first item that went to either side: THE item if it stays alone)
    float4 l_next = make_float4(0, 0, 0, 0), h_next = l_next;   // (the next trip's items are on their way while this trip's are placed)
    if (b + lane < e) {
        l_next = in_lo[b + lane];
        h_next = in_hi[b + lane];
    }
    for (uint32_t base = b; base < e; base += 64u) {
        const uint32_t i = base + lane;
        const bool valid = i < e;
        float4 l = l_next, h = h_next;
        if (i + 64u < e) {
            l_next = in_lo[i + 64u];
            h_next = in_hi[i + 64u];
        }
        bool left = false;
        if (valid) {
            if (pick >= 0) {
                int q = (int)((0.5f * (axis_of(l, paxis) + axis_of(h, paxis)) - cl[paxis]) * inv[paxis]);
                q = q < 0 ? 0 : (q > 15 ? 15 : q);
                left = q <= psplit;
            } else {
                left = i - b < m / 2u;   // no plane separates the box centres: halves
            }
        }
        const unsigned long long lm = __ballot(valid && left), rm = __ballot(valid && !left);
        if (valid) {
            const uint32_t dst = left ? b + nlft + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))
                                      : e - 1u - nrgt - (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
            out_lo[dst] = l;
            out_hi[dst] = h;
        }
        if (nlft == 0u && lm != 0ull) one_left = __shfl(__float_as_int(l.w), __ffsll((long long)lm) - 1);
        if (nrgt == 0u && rm != 0ull) one_right = __shfl(__float_as_int(l.w), __ffsll((long long)rm) - 1);
        nlft += (uint32_t)__popcll(lm);
        nrgt += (uint32_t)__popcll(rm);
    }
    if (lane == 0) {
        uint32_t* P = s_push[wv];
        P[0] = b, P[1] = e, P[2] = nlft, P[3] = k, P[4] = (uint32_t)one_left, P[5] = (uint32_t)one_right;
    }
    }
    // the children of the block's tasks, pushed by the lanes of ONE wave (see sah_push_children)
    __syncthreads();
    if (wv == 0 && lane < MPT_SAH_WAVES && s_push[lane][1] != 0u) {
        const uint32_t* P = s_push[lane];
        sah_push_children(st->cnt[(level + 1u) % 3u], s_child, next, next_big, next_small, P[0], P[1], P[2], P[3], (int)P[4], (int)P[5], fs);
    }
}

// The same for the BIG tasks of a level (>= MPT_SAH_BIG items: the top ten levels of a 1 M-primitive tree, one to a few hundred
// tasks over ALL the items).  One workgroup per task would leave the chip idle exactly where the passes are longest (the root:
// one workgroup streaming 1 M items three times, 1.1 ms), so the two passes that touch every item — the bounds and the
// partition — run over CHUNKS of 2048 items, a workgroup each, whatever task a chunk belongs to:
//   k_big_prep     the tasks' first chunk numbers (prefix over ceil(items / 2048)), accumulators reset          1 workgroup
//   k_big_bounds   node box + bounds of the box centres, a chunk each, merged by atomics                          per chunk
//   k_big_pick     bins from an evenly spaced sample of ~1024 items, the 45 planes priced, the node made          per task
//   k_big_count    items that go left, per chunk                                                                  per chunk
//   k_big_scatter  the partition: left from b upwards, right from e - 1 downwards, in item order (deterministic)  per chunk
//   k_big_push     the children: tasks of the next level, or single items attached                                per task
#define MPT_SAH_BIG_THREADS 1024
#define MPT_SAH_CHUNK 2048u
#define MPT_SAH_CHUNK_THREADS 256u
#ifndef MPT_SAH_SAMPLE_BIG
#define MPT_SAH_SAMPLE_BIG 1024u
#endif
struct SahBig {
    int bounds[12];   // node box lo / hi, centre bounds lo / hi (ordered ints)
    uint32_t k;       // the node
    int pick;         // axis * 15 + split, or -1: halves
    float cl[3], inv[3];
    int one[2];       // id of the first item on either side (THE item if it stays alone)
    unsigned long long largest;   // sah_area_key of the item with the largest box
};
__device__ __forceinline__ uint32_t sah_chunks(const SahTask& t) { return (t.e - t.b + MPT_SAH_CHUNK - 1u) / MPT_SAH_CHUNK; }
__global__ __launch_bounds__(1024) void k_big_prep(const SahTask* tasks, SahState* st, SahBig* big, uint32_t* coff /* [n_big + 1] */, SahLevel lv) {
    sah_level_prologue(st, lv);
    const uint32_t n_big = st->cnt[lv.level % 3u][16];
    __shared__ uint32_t s_w[16], s_run;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) s_run = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n_big; base += 1024u) {
        const uint32_t t = base + tid;
        const uint32_t c = t < n_big ? sah_chunks(tasks[t]) : 0u;
        uint32_t x = c;   // inclusive prefix within the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if ((int)lane >= off) x += y;
        }
        if (lane == 63u) s_w[wv] = x;
        __syncthreads();
        uint32_t before = s_run;
        for (uint32_t w = 0; w < wv; ++w) before += s_w[w];
        if (t < n_big) {
            coff[t] = before + x - c;
            SahBig& B = big[t];
            for (int q = 0; q < 12; ++q) B.bounds[q] = q % 6 < 3 ? 0x7FFFFFFF : (int)0x80000000;
            B.one[0] = B.one[1] = 0;
            B.largest = 0ull;
        }
        __syncthreads();
        if (tid == 1023u) s_run = before + x;
        __syncthreads();
    }
    if (tid == 0) coff[n_big] = s_run;
}
// the task a chunk belongs to: the last t with coff[t] <= c
__device__ __forceinline__ uint32_t sah_task_of_chunk(const uint32_t* coff, uint32_t n_big, uint32_t c) {
    uint32_t lo = 0, hi = n_big;   // coff[lo] <= c < coff[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (coff[mid] <= c) lo = mid;
        else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(MPT_SAH_CHUNK_THREADS) void k_big_bounds(const float4* in_lo, const float4* in_hi, const SahTask* tasks, const SahState* st,
                                                                      const uint32_t* coff, SahBig* big, uint32_t level) {
    const uint32_t n_big = st->cnt[level % 3u][16];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    if (c >= coff[n_big]) return;
    const uint32_t t = sah_task_of_chunk(coff, n_big, c);
    const SahTask task = tasks[t];
    const uint32_t b = task.b + (c - coff[t]) * MPT_SAH_CHUNK, e = b + MPT_SAH_CHUNK < task.e ? b + MPT_SAH_CHUNK : task.e;
    float nl[3] = {INFINITY, INFINITY, INFINITY}, nh[3] = {-INFINITY, -INFINITY, -INFINITY}, cl[3] = {INFINITY, INFINITY, INFINITY},
          ch[3] = {-INFINITY, -INFINITY, -INFINITY};
    unsigned long long largest = 0ull;
    for (uint32_t i = b + tid; i < e; i += MPT_SAH_CHUNK_THREADS) {
        const float4 l = in_lo[i], h = in_hi[i];
        const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
        for (int a = 0; a < 3; ++a) {
            nl[a] = fminf(nl[a], lo3[a]);
            nh[a] = fmaxf(nh[a], hi3[a]);
            const float cc = 0.5f * (lo3[a] + hi3[a]);
            cl[a] = fminf(cl[a], cc);
            ch[a] = fmaxf(ch[a], cc);
        }
        const unsigned long long key = sah_area_key(l, h, i);
        largest = key > largest ? key : largest;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(largest, off);
        largest = o > largest ? o : largest;
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; ++a) {
            nl[a] = fminf(nl[a], __shfl_xor(nl[a], off));
            nh[a] = fmaxf(nh[a], __shfl_xor(nh[a], off));
            cl[a] = fminf(cl[a], __shfl_xor(cl[a], off));
            ch[a] = fmaxf(ch[a], __shfl_xor(ch[a], off));
        }
    __shared__ int s_b[12];
    __shared__ unsigned long long s_largest;
    if (tid < 12u) s_b[tid] = tid % 6u < 3u ? 0x7FFFFFFF : (int)0x80000000;
    if (tid == 0) s_largest = 0ull;
    __syncthreads();
    if (lane == 0) atomicMax(&s_largest, largest);
    if (lane == 0)
        for (int a = 0; a < 3; ++a) {
            atomicMin(&s_b[a], f2o(nl[a]));
            atomicMax(&s_b[3 + a], f2o(nh[a]));
            atomicMin(&s_b[6 + a], f2o(cl[a]));
            atomicMax(&s_b[9 + a], f2o(ch[a]));
        }
    __syncthreads();
    if (tid < 12u) {   // (one set of global atomics per chunk)
        int* B = big[t].bounds;
        if (tid % 6u < 3u) atomicMin(&B[tid], s_b[tid]);
        else atomicMax(&B[tid], s_b[tid]);
    }
    if (tid == 12u) atomicMax(&big[t].largest, s_largest);
}
__global__ __launch_bounds__(MPT_SAH_BIG_THREADS) void k_big_pick(int n, const float4* in_lo, const float4* in_hi, const SahTask* tasks, SahBig* big, SahState* st,
                                                                  int2* s_child, float4* s_lo, float4* s_hi, uint32_t level, SahFirst fs) {
    // (a set of bins per wave, merged afterwards — min, max and sums of integers: the same bins whatever the order; with ONE set the 16
    //  waves' atomics met on 48 hot addresses and the levels that bin every item of tasks up to 8192 took 76 / 62 / 35 us)
    constexpr uint32_t NWV = MPT_SAH_BIG_THREADS / 64u;
    __shared__ int bins_w[NWV][3][16][7];
    int (*bins)[16][7] = bins_w[0];
    __shared__ float s_cost[48];
    const uint32_t tid = threadIdx.x;
    const int TOP = 2 * n - 1;
    if (blockIdx.x >= st->cnt[level % 3u][16]) return;   // (grid sized from an upper bound)
    const SahTask task = tasks[blockIdx.x];
    SahBig& G = big[blockIdx.x];
    const uint32_t b = task.b, e = task.e, m = e - b;
    for (uint32_t q = tid; q < NWV * 3u * 16u * 7u; q += MPT_SAH_BIG_THREADS) {
        const uint32_t f = q % 7u;
        (&bins_w[0][0][0][0])[q] = f < 3u ? 0x7FFFFFFF : f < 6u ? (int)0x80000000 : 0;
    }
    float cl[3], inv[3];
    for (int a = 0; a < 3; ++a) {
        cl[a] = o2f(G.bounds[6 + a]);
        const float ext = o2f(G.bounds[9 + a]) - cl[a];
        inv[a] = ext > 0.0f && isfinite(ext) ? 16.0f / ext : 0.0f;
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t k = task.node;
        G.k = k;
        s_lo[k] = make_float4(o2f(G.bounds[0]), o2f(G.bounds[1]), o2f(G.bounds[2]), __int_as_float((int)m));
        if (fs.inner) fs.inner[k] = b;
        s_hi[k] = make_float4(o2f(G.bounds[3]), o2f(G.bounds[4]), o2f(G.bounds[5]), 0.0f);
        sah_attach(st, s_child, task.parent, task.side, TOP + (int)k);
    }
    // (binned from a sample of ~1024 items, plus the one with the largest box if the sample missed it: see k_sah_level)
    // (up to 8192 items all are binned: a scene of 5000 primitives keeps the host builder's exact choices at its root, where two
    //  unsampled spheres of radius 10 and 40 made the sample prefer another axis)
    const uint32_t step = m > 8u * MPT_SAH_SAMPLE_BIG ? m / MPT_SAH_SAMPLE_BIG : 1u;
    auto bin_item = [&](uint32_t i) {
        const float4 l = in_lo[i], h = in_hi[i];
        const int cnt = __float_as_int(h.w);
        const float lo3[3] = {l.x, l.y, l.z}, hi3[3] = {h.x, h.y, h.z};
        for (int a = 0; a < 3; ++a) {
            if (inv[a] == 0.0f) continue;
            int q = (int)((0.5f * (lo3[a] + hi3[a]) - cl[a]) * inv[a]);
            q = q < 0 ? 0 : (q > 15 ? 15 : q);
            int* B = bins_w[tid >> 6][a][q];
            atomicMin(&B[0], f2o(l.x)); atomicMin(&B[1], f2o(l.y)); atomicMin(&B[2], f2o(l.z));
            atomicMax(&B[3], f2o(h.x)); atomicMax(&B[4], f2o(h.y)); atomicMax(&B[5], f2o(h.z));
            atomicAdd(&B[6], cnt);
        }
    };
    for (uint32_t i = b + tid * step; i < e; i += MPT_SAH_BIG_THREADS * step) bin_item(i);
    if (tid == 0 && step > 1u && ((uint32_t)G.largest - b) % step != 0u) bin_item((uint32_t)G.largest);
    __syncthreads();
    if (tid < 3u * 16u * 7u) {
        const uint32_t f = tid % 7u;
        int v = (&bins_w[0][0][0][0])[tid];
        for (uint32_t w = 1; w < NWV; ++w) {
            const int o = (&bins_w[w][0][0][0])[tid];
            v = f < 3u ? (o < v ? o : v) : f < 6u ? (o > v ? o : v) : v + o;
        }
        (&bins_w[0][0][0][0])[tid] = v;
    }
    __syncthreads();
    if (tid < 48u) {
        float cost = INFINITY;
        if (tid < 45u) {
            const int a = (int)(tid / 15u), sp = (int)(tid % 15u);
            if (inv[a] != 0.0f) {
                float L[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, R[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int cL = 0, cR = 0;
                for (int q = 0; q < 16; ++q) {
                    const int* B = bins[a][q];
                    if (B[6] == 0) continue;
                    float* D = q <= sp ? L : R;
                    for (int c = 0; c < 3; ++c) {
                        D[c] = fminf(D[c], o2f(B[c]));
                        D[3 + c] = fmaxf(D[3 + c], o2f(B[3 + c]));
                    }
                    if (q <= sp) cL += B[6];
                    else cR += B[6];
                }
                if (cL != 0 && cR != 0)
                    cost = half_area4(make_float4(L[0], L[1], L[2], 0), make_float4(L[3], L[4], L[5], 0)) * (float)cL +
                           half_area4(make_float4(R[0], R[1], R[2], 0), make_float4(R[3], R[4], R[5], 0)) * (float)cR;
            }
        }
        s_cost[tid] = cost;
    }
    __syncthreads();
    if (tid == 0) {
        int pick = -1;
        float best = INFINITY;
        for (int q = 0; q < 45; ++q)
            if (s_cost[q] < best) best = s_cost[q], pick = q;   // ties: the lowest (axis, split), as in the wave kernel
        G.pick = pick;
        for (int a = 0; a < 3; ++a) G.cl[a] = cl[a], G.inv[a] = inv[a];
    }
}
// which side item i of the task goes to
struct SahSplit {
    int axis, split;   // axis < 0: no plane separates the box centres: halves
    float cl, inv;
    uint32_t b, half;
};
__device__ __forceinline__ SahSplit sah_split_of(const SahBig* G, const SahTask& task) {
    SahSplit s;
    const int pick = G->pick;
    s.axis = pick < 0 ? -1 : pick / 15;
    s.split = pick < 0 ? 0 : pick % 15;
    s.cl = s.axis == 1 ? G->cl[1] : s.axis == 2 ? G->cl[2] : G->cl[0];
    s.inv = s.axis == 1 ? G->inv[1] : s.axis == 2 ? G->inv[2] : G->inv[0];
    s.b = task.b;
    s.half = (task.e - task.b) / 2u;
    return s;
}
__device__ __forceinline__ bool sah_big_left(const SahSplit& s, uint32_t i, float4 l, float4 h) {
    if (s.axis < 0) return i - s.b < s.half;
    int q = (int)((0.5f * (axis_of(l, s.axis) + axis_of(h, s.axis)) - s.cl) * s.inv);
    q = q < 0 ? 0 : (q > 15 ? 15 : q);
    return q <= s.split;
}
__global__ __launch_bounds__(MPT_SAH_CHUNK_THREADS) void k_big_count(const float4* in_lo, const float4* in_hi, const SahTask* tasks, const SahState* st,
                                                                     const uint32_t* coff, const SahBig* big, uint32_t* chunk_left, uint32_t level) {
    const uint32_t n_big = st->cnt[level % 3u][16];
    __shared__ uint32_t s_w[MPT_SAH_CHUNK_THREADS / 64u];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (c >= coff[n_big]) return;
    const uint32_t t = sah_task_of_chunk(coff, n_big, c);
    const SahTask task = tasks[t];
    const SahSplit G = sah_split_of(big + t, task);
    const uint32_t b = task.b + (c - coff[t]) * MPT_SAH_CHUNK, e = b + MPT_SAH_CHUNK < task.e ? b + MPT_SAH_CHUNK : task.e;
    uint32_t cnt = 0;
    for (uint32_t i = b + tid; i < e; i += MPT_SAH_CHUNK_THREADS) cnt += sah_big_left(G, i, in_lo[i], in_hi[i]) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) s_w[wv] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t sum = 0;
        for (uint32_t w = 0; w < MPT_SAH_CHUNK_THREADS / 64u; ++w) sum += s_w[w];
        chunk_left[c] = sum;
    }
}
__global__ __launch_bounds__(MPT_SAH_CHUNK_THREADS) void k_big_scatter(const float4* in_lo, const float4* in_hi, float4* out_lo, float4* out_hi, const SahTask* tasks,
                                                                       const SahState* st, const uint32_t* coff, SahBig* big, const uint32_t* chunk_left, uint32_t level) {
    const uint32_t n_big = st->cnt[level % 3u][16];
    constexpr uint32_t NW = MPT_SAH_CHUNK_THREADS / 64u;
    __shared__ uint32_t s_wl[NW], s_wr[NW], s_before[NW];
    const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (c >= coff[n_big]) return;
    const uint32_t t = sah_task_of_chunk(coff, n_big, c);
    const SahTask task = tasks[t];
    const SahSplit G = sah_split_of(big + t, task);
    const uint32_t j = c - coff[t];
    const uint32_t b = task.b + j * MPT_SAH_CHUNK, e = b + MPT_SAH_CHUNK < task.e ? b + MPT_SAH_CHUNK : task.e;
    // items that went left in the task's chunks before this one
    uint32_t lb = 0;
    for (uint32_t q = tid; q < j; q += MPT_SAH_CHUNK_THREADS) lb += chunk_left[coff[t] + q];
    for (int off = 32; off > 0; off >>= 1) lb += __shfl_xor(lb, off);
    if (lane == 0) s_before[wv] = lb;
    __syncthreads();
    uint32_t nlft = 0;
    for (uint32_t w = 0; w < NW; ++w) nlft += s_before[w];
    uint32_t nrgt = j * MPT_SAH_CHUNK - nlft;
    __syncthreads();
    for (uint32_t base = b; base < e; base += MPT_SAH_CHUNK_THREADS) {
        const uint32_t i = base + tid;
        const bool valid = i < e;
        float4 l = make_float4(0, 0, 0, 0), h = l;
        bool left = false;
        if (valid) {
            l = in_lo[i];
            h = in_hi[i];
            left = sah_big_left(G, i, l, h);
        }
        const unsigned long long lm = __ballot(valid && left), rm = __ballot(valid && !left);
        if (lane == 0) {
            s_wl[wv] = (uint32_t)__popcll(lm);
            s_wr[wv] = (uint32_t)__popcll(rm);
        }
        __syncthreads();
        uint32_t pl = 0, pr = 0, tl = 0, tr = 0;
        for (uint32_t w = 0; w < NW; ++w) {
            if (w < wv) pl += s_wl[w], pr += s_wr[w];
            tl += s_wl[w];
            tr += s_wr[w];
        }
        if (valid) {
            const uint32_t dst = left ? task.b + nlft + pl + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))
                                      : task.e - 1u - nrgt - pr - (uint32_t)__popcll(rm & ((1ull << lane) - 1ull));
            out_lo[dst] = l;
            out_hi[dst] = h;
            if (dst == task.b && left) big[t].one[0] = __float_as_int(l.w);
            if (dst == task.e - 1u && !left) big[t].one[1] = __float_as_int(l.w);
        }
        nlft += tl;
        nrgt += tr;
        __syncthreads();
    }
}
// (a block role of k_sah_tasks below: it needs the level's k_big_count, nothing of its mid and small tasks.  A WAVE per task: the root's
//  489 chunk counts added up by one thread were 24 us of the first level, 14 / 10 / 7 of the next)
__device__ __forceinline__ void sah_big_push_wave(uint32_t t, const SahTask* tasks, const uint32_t* coff, const SahBig* big, const uint32_t* chunk_left, SahTask* next,
                                                  SahTask* next_big, SahTask* next_small, SahState* st, int2* s_child, uint32_t level, const SahFirst& fs) {
    if (t >= st->cnt[level % 3u][16]) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nlft = 0;
    for (uint32_t c = coff[t] + lane; c < coff[t + 1u]; c += 64u) nlft += chunk_left[c];
    for (int off = 32; off > 0; off >>= 1) nlft += __shfl_xor(nlft, off);
    if (lane == 0)
        sah_push_children(st->cnt[(level + 1u) % 3u], s_child, next, next_big, next_small, tasks[t].b, tasks[t].e, nlft, big[t].k, big[t].one[0], big[t].one[1], fs);
}

// ... and for a SMALL task (<= MPT_SAH_SMALL items: the last three or four levels, which hold most of the tree's nodes and took
// 70 % of the builder's time at a wave per node): one lane per item, 8 tasks per wave, and the group finishes the whole
// sub-tree — every round splits ALL the ranges the group holds at once (segmented reductions over the group's lanes by
// shuffles, the partition by ds_permute), so no task of the bottom levels is ever queued.  Same algorithm and same choices as
// k_sah_level (16 bins over the box centres, cost = area * primitives, ties to the lowest (axis, split)); a lane prices the
// three planes behind its own item's bins, which are all the planes that separate anything.
__device__ __forceinline__ void sah_small_thread(uint32_t thread /* among the level's small-task threads */, int n, const float4* in_lo, const float4* in_hi,
                                                 const SahTask* tasks, SahState* st, int2* s_child, float4* s_lo, float4* s_hi, uint32_t level, const SahFirst& fs) {
    const uint32_t n_tasks = st->cnt[level % 3u][32];   // (the grid is sized from an upper bound)
    constexpr uint32_t G = MPT_SAH_SMALL;
    static_assert(G == 8u || G == 16u, "MPT_SAH_SMALL: 8 or 16");
    const uint32_t t = thread / G, lane = threadIdx.x & 63u, gl = lane & (G - 1u), gbase = lane & ~(G - 1u);
    const int TOP = 2 * n - 1;
    SahTask task = SahTask{0u, 0u, -1, 0u, 0u};
    if (t < n_tasks) task = tasks[t];
    bool live = gl < task.e - task.b;
    float4 l = make_float4(0, 0, 0, 0), h = l;
    if (live) {
        l = in_lo[task.b + gl];
        h = in_hi[task.b + gl];
    }
    uint32_t rb = 0u, re = task.e - task.b;   // this lane's range, in lanes of the group
    int parent = task.parent;
    uint32_t side = task.side, nb = task.node;   // (nb: the node this lane's range makes)
    for (uint32_t round = 0; round < G; ++round) {
        if (__ballot(live) == 0ull) break;
        // the range's box, the bounds of its box centres
        const float c3[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
        float nl[3] = {INFINITY, INFINITY, INFINITY}, nh[3] = {-INFINITY, -INFINITY, -INFINITY}, cl[3] = {INFINITY, INFINITY, INFINITY},
              ch[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < G; ++i) {
            const int src = (int)(gbase + i);
            const bool same = __shfl((int)live, src) != 0 && (uint32_t)__shfl((int)rb, src) == rb;
            const float lo3[3] = {__shfl(l.x, src), __shfl(l.y, src), __shfl(l.z, src)}, hi3[3] = {__shfl(h.x, src), __shfl(h.y, src), __shfl(h.z, src)};
            if (same)
                for (int a = 0; a < 3; ++a) {
                    nl[a] = fminf(nl[a], lo3[a]);
                    nh[a] = fmaxf(nh[a], hi3[a]);
                    const float c = 0.5f * (lo3[a] + hi3[a]);
                    cl[a] = fminf(cl[a], c);
                    ch[a] = fmaxf(ch[a], c);
                }
        }
        const uint32_t m = re - rb;
        const uint32_t k = nb;
        if (live && gl == rb) {
            s_lo[k] = make_float4(nl[0], nl[1], nl[2], __int_as_float((int)m));
            if (fs.inner) fs.inner[k] = task.b + rb;
            s_hi[k] = make_float4(nh[0], nh[1], nh[2], 0.0f);
            sah_attach(st, s_child, parent, side, TOP + (int)k);
        }
        // the plane behind this lane's item on every axis, priced over the range.  MPT_SAH_SMALL_EXACT (default): "behind" is the
        // order of the box centres themselves (ties by lane) — a full sweep, every partition an axis can make; 0: the 16 bins of
        // the larger nodes (the host builder's choices exactly)
        float inv[3];
        int q3[3];
        for (int a = 0; a < 3; ++a) {
            const float ext = ch[a] - cl[a];
            inv[a] = ext > 0.0f && isfinite(ext) ? 16.0f / ext : 0.0f;
            int q = (int)((c3[a] - cl[a]) * inv[a]);
            q3[a] = q < 0 ? 0 : (q > 15 ? 15 : q);
        }
        const int qpack = q3[0] | (q3[1] << 4) | (q3[2] << 8);
        (void)qpack;
        float L[3][6], R[3][6];
        int cL[3] = {0, 0, 0}, cR[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 6; ++c) L[a][c] = R[a][c] = c < 3 ? INFINITY : -INFINITY;
        for (uint32_t i = 0; i < G; ++i) {
            const int src = (int)(gbase + i);
            const bool same = __shfl((int)live, src) != 0 && (uint32_t)__shfl((int)rb, src) == rb;
            const float b6[6] = {__shfl(l.x, src), __shfl(l.y, src), __shfl(l.z, src), __shfl(h.x, src), __shfl(h.y, src), __shfl(h.z, src)};
            const int cnt = __shfl(__float_as_int(h.w), src), qo = __shfl(qpack, src);
            if (same)
                for (int a = 0; a < 3; ++a) {
#if MPT_SAH_SMALL_EXACT
                    const float co = 0.5f * (b6[a] + b6[3 + a]);
                    const bool left = co < c3[a] || (co == c3[a] && i <= gl);
                    (void)qo;
#else
                    const bool left = ((qo >> (4 * a)) & 15) <= q3[a];
#endif
                    for (int c = 0; c < 3; ++c) {
                        if (left) L[a][c] = fminf(L[a][c], b6[c]), L[a][3 + c] = fmaxf(L[a][3 + c], b6[3 + c]);
                        else R[a][c] = fminf(R[a][c], b6[c]), R[a][3 + c] = fmaxf(R[a][3 + c], b6[3 + c]);
                    }
                    if (left) cL[a] += cnt;
                    else cR[a] += cnt;
                }
        }
        float cost = INFINITY;
        int cand = 0x7FFFFFFF;   // axis * 15 + split
        for (int a = 0; a < 3; ++a) {
            if (!live || inv[a] == 0.0f || cL[a] == 0 || cR[a] == 0) continue;
            const float c = half_area4(make_float4(L[a][0], L[a][1], L[a][2], 0), make_float4(L[a][3], L[a][4], L[a][5], 0)) * (float)cL[a] +
                            half_area4(make_float4(R[a][0], R[a][1], R[a][2], 0), make_float4(R[a][3], R[a][4], R[a][5], 0)) * (float)cR[a];
#if MPT_SAH_SMALL_EXACT
            const int id = a * 16 + (int)gl;
#else
            const int id = a * 15 + q3[a];
#endif
            if (c < cost || (c == cost && id < cand)) cost = c, cand = id;
        }
        float best = INFINITY;
        int pick = 0x7FFFFFFF;
        for (uint32_t i = 0; i < G; ++i) {
            const int src = (int)(gbase + i);
            const bool same = __shfl((int)live, src) != 0 && (uint32_t)__shfl((int)rb, src) == rb;
            const float c = __shfl(cost, src);
            const int id = __shfl(cand, src);
            if (same && (c < best || (c == best && id < pick))) best = c, pick = id;
        }
        bool left;
#if MPT_SAH_SMALL_EXACT
        {
            const int pa = pick == 0x7FFFFFFF ? 0 : pick / 16, pl = pick == 0x7FFFFFFF ? 0 : pick % 16, src = (int)gbase + pl;
            const float p0 = __shfl(c3[0], src), p1 = __shfl(c3[1], src), p2 = __shfl(c3[2], src);
            const float cp = pa == 0 ? p0 : pa == 1 ? p1 : p2, cm = pa == 0 ? c3[0] : pa == 1 ? c3[1] : c3[2];
            left = cm < cp || (cm == cp && (int)gl <= pl);
        }
        if (!(best < INFINITY)) left = gl - rb < m / 2u;   // no plane separates the box centres: halves
#else
        if (best < INFINITY) left = q3[pick / 15] <= pick % 15;
        else left = gl - rb < m / 2u;   // no plane separates the box centres: halves
#endif
        if (m == 2u) left = gl == rb;   // (two items: nothing to choose)
        // the partition, stable on both sides
        const unsigned long long seg = (re - rb >= 64u ? ~0ull : ((1ull << (re - rb)) - 1ull)) << (gbase + rb);
        const unsigned long long lm = __ballot(live && left) & seg, rm = __ballot(live && !left) & seg, below = (1ull << lane) - 1ull;
        const uint32_t nlft = (uint32_t)__popcll(lm);
        uint32_t dst = gl;
        if (live) dst = left ? rb + (uint32_t)__popcll(lm & below) : rb + nlft + (uint32_t)__popcll(rm & below);
        const int to = (int)((gbase + dst) * 4u);
        l.x = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(l.x)));
        l.y = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(l.y)));
        l.z = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(l.z)));
        l.w = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(l.w)));
        h.x = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(h.x)));
        h.y = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(h.y)));
        h.z = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(h.z)));
        h.w = __int_as_float(__builtin_amdgcn_ds_permute(to, __float_as_int(h.w)));
        // (a lane stays in its old range, so the range's nlft and k are still the ones it knows)
        if (live) {
            parent = (int)k;
            if (gl < rb + nlft) re = rb + nlft, side = 0u, nb = k + 1u;
            else rb = rb + nlft, side = 1u, nb = k + nlft;
            if (re - rb == 1u) {   // alone: attached at once
                if (side == 0u) s_child[k].x = __float_as_int(l.w);
                else s_child[k].y = __float_as_int(l.w);
                if (fs.item) fs.item[__float_as_int(l.w)] = task.b + rb;   // (= task.b + gl: the lane holds the item of its own position)
                live = false;
            }
        }
    }
}

// The three kinds of per-task work of a level that need nothing of one another, in ONE launch while the level has big tasks (until round 4:
// three — and in the levels with big tasks there are few tasks of the other kinds or none, so two of them ran nearly empty grids):
// blocks [0, mid_blocks) split the level's mid tasks, a wave each; [mid_blocks, + small_blocks) finish its small tasks; the rest push the
// children of its big tasks (whose partition the kernels before this one made).  The mid blocks come first: they run longest.
__global__ __launch_bounds__(64 * MPT_SAH_WAVES) void k_sah_tasks(int n, const float4* in_lo, const float4* in_hi, float4* out_lo, float4* out_hi, const SahTask* tasks,
                                                                  const SahTask* small_tasks, const SahTask* big_tasks, SahTask* next, SahTask* next_big,
                                                                  SahTask* next_small, SahState* st, int2* s_child, float4* s_lo, float4* s_hi, const uint32_t* coff,
                                                                  const SahBig* big, const uint32_t* chunk_left, uint32_t mid_blocks, uint32_t small_blocks, SahLevel lv, SahFirst fs) {
    sah_level_prologue(st, lv);
    if (blockIdx.x < mid_blocks)
        sah_mid_block(blockIdx.x, n, in_lo, in_hi, out_lo, out_hi, tasks, next, next_big, next_small, st, s_child, s_lo, s_hi, lv.level, fs);
    else if (blockIdx.x < mid_blocks + small_blocks)
        sah_small_thread((blockIdx.x - mid_blocks) * (64u * MPT_SAH_WAVES) + threadIdx.x, n, in_lo, in_hi, small_tasks, st, s_child, s_lo, s_hi, lv.level, fs);
    else
        sah_big_push_wave((blockIdx.x - mid_blocks - small_blocks) * MPT_SAH_WAVES + (threadIdx.x >> 6), big_tasks, coff, big, chunk_left, next, next_big, next_small, st,
                          s_child, lv.level, fs);
}

// ... and the mid and the small tasks on their own, for the levels that have no big task any more (most of the tree's nodes are made there:
// in one kernel the two kinds of blocks ran 20 % slower than one after the other, 1060 against 885 us for the last ten levels of 1 M items)
__global__ __launch_bounds__(64 * MPT_SAH_WAVES) void k_sah_level(int n, const float4* in_lo, const float4* in_hi, float4* out_lo, float4* out_hi, const SahTask* tasks,
                                                                  SahTask* next, SahTask* next_big, SahTask* next_small, SahState* st, int2* s_child, float4* s_lo,
                                                                  float4* s_hi, SahLevel lv, SahFirst fs) {
    sah_level_prologue(st, lv);
    sah_mid_block(blockIdx.x, n, in_lo, in_hi, out_lo, out_hi, tasks, next, next_big, next_small, st, s_child, s_lo, s_hi, lv.level, fs);
}
__global__ __launch_bounds__(256) void k_sah_small(int n, const float4* in_lo, const float4* in_hi, const SahTask* tasks, SahState* st, int2* s_child, float4* s_lo,
                                                   float4* s_hi, SahLevel lv, SahFirst fs) {
    sah_level_prologue(st, lv);
    sah_small_thread(blockIdx.x * blockDim.x + threadIdx.x, n, in_lo, in_hi, tasks, st, s_child, s_lo, s_hi, lv.level, fs);
}

struct SahTree {
    int2* child = nullptr;
    float4 *lo = nullptr, *hi = nullptr;
    SahState* st = nullptr;   // on the device: root, n_nodes
    SahFirst first = {nullptr, nullptr};   // (in: arrays the caller wants filled — SahFirst above)
};
// it_lo / it_hi: the items (device, consumed: the partitions ping-pong between them and two scratch arrays).  The item count
// is *d_count if d_count is not null (a device word), else count_host; max_items bounds it.  pin: >= 1 KiB of pinned host memory.
//
// No host wait inside the level loop (round 4; round 3 synchronised the stream once per level to read three counters: ~22 idle
// gaps of 30-40 us for 1 M items).  The kernels of a level read their task counts from the device (SahState::cnt, sah_level_prologue) and
// are launched on grids sized from an UPPER BOUND — a level at most doubles the number of tasks, so the bound comes from the
// exact counts of one or two levels before, which the level's first kernel left in pinned memory with a stamp.  The host only ever waits for a
// level the device has already passed (or is about to): the device always has the next level queued.  The loop ends one or two
// (empty) levels late; waves and workgroups beyond the true counts return at once.
static hipError_t run_sah(hipStream_t stream, Scratch& sc, uint32_t* pin, int top, const uint32_t* d_count, uint32_t count_host, uint32_t max_items,
                          float4* it_lo_a, float4* it_hi_a, SahTree& T) {
    float4 *it_lo_b, *it_hi_b;
    SahTask *tasks_a, *tasks_b, *big_a, *big_b, *small_a, *small_b;
    MPT_LB(sc.alloc(&T.st, 1));
    MPT_LB(sc.alloc(&it_lo_b, max_items));
    MPT_LB(sc.alloc(&it_hi_b, max_items));
    MPT_LB(sc.alloc(&tasks_a, max_items + 2));
    MPT_LB(sc.alloc(&tasks_b, max_items + 2));
    MPT_LB(sc.alloc(&big_a, max_items / MPT_SAH_BIG + 2));
    MPT_LB(sc.alloc(&big_b, max_items / MPT_SAH_BIG + 2));
    MPT_LB(sc.alloc(&small_a, max_items / 2 + 2));
    MPT_LB(sc.alloc(&small_b, max_items / 2 + 2));
    const uint32_t max_big = max_items / MPT_SAH_BIG + 2;
    SahBig* bigs;
    uint32_t *coff, *chunk_left;
    MPT_LB(sc.alloc(&bigs, max_big));
    MPT_LB(sc.alloc(&coff, max_big + 1));
    MPT_LB(sc.alloc(&chunk_left, max_items / MPT_SAH_CHUNK + max_big + 1));
    MPT_LB(sc.alloc(&T.child, max_items));
    MPT_LB(sc.alloc(&T.lo, max_items));
    MPT_LB(sc.alloc(&T.hi, max_items));
    // (k_sah_* take n with top = 2n - 1)
    const int n = (top + 1) / 2;
    // eight slots of three 8-byte words (stamp << 32 | mid tasks, big tasks, small tasks) in pinned memory, behind the words other read-backs
    // of the build use
    static uint32_t s_epoch = 0;
    const uint32_t epoch = (++s_epoch & 0x7FFFu) << 16;
    volatile unsigned long long* slots = (volatile unsigned long long*)(pin + 64);
    unsigned long long* d_slots = nullptr;
    MPT_LB(hipHostGetDevicePointer((void**)&d_slots, pin + 64, 0));
    for (int q = 0; q < 24; ++q) slots[q] = 0ull;
    hipLaunchKernelGGL(k_sah_init, dim3(1), dim3(64), 0, stream, d_count, count_host, T.st, tasks_a, big_a, small_a);
    const uint32_t cap_mid = max_items / (MPT_SAH_SMALL + 1u) + 2u, cap_small = max_items / 2u + 2u;
    auto slot_ready = [&](int level) {   // all three words of the level's slot carry its stamp
        volatile unsigned long long* sl = slots + 3 * (level & 7);
        const uint32_t want = epoch | (uint32_t)(level + 1);
        for (int q = 0; q < 3; ++q)
            if (((uint32_t)(sl[q] >> 32) & ~MPT_SAH_STAMP_OVERFLOW) != want) return false;
        return true;
    };
    auto wait_slot = [&](int level, uint32_t out[3]) -> hipError_t {   // the counts of `level`, once its first kernel has started
        volatile unsigned long long* sl = slots + 3 * (level & 7);
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned long long spin = 0; !slot_ready(level); ++spin) {
            if ((spin & 0xFFFu) == 0xFFFu) {
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipSuccess && q != hipErrorNotReady) return q;
                if (q == hipSuccess && !slot_ready(level)) return hipErrorUnknown;   // the stream is drained and the stamp never came
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return hipErrorLaunchTimeOut;   // (a level takes < 1 ms: a kernel hangs)
            }
        }
        unsigned long long w[3] = {sl[0], sl[1], sl[2]};
        for (int q = 0; q < 3; ++q) {
            if ((uint32_t)(w[q] >> 32) & MPT_SAH_STAMP_OVERFLOW) return hipErrorLaunchOutOfResources;   // a level had more tasks than its grids covered (sah_level_prologue)
            out[q] = (uint32_t)w[q];
        }
        return hipSuccess;
    };
    uint32_t b_mid = 1u, b_big = 1u, b_small = 1u;   // upper bounds of the level about to be enqueued (level 0: the root task, of one kind)
    int known = -1;                                   // the latest level whose exact counts the host has read
    bool done = false;
    int level = 0;
    for (; level < 4096 && !done; ++level) {
        // (the level's first kernel publishes its counts and zeroes the set of the level after: sah_level_prologue)
        SahLevel first = {(uint32_t)level, epoch | (uint32_t)(level + 1), d_slots + 3 * (level & 7), {b_mid, b_big, b_small}};
        const SahLevel rest = {(uint32_t)level, 0u, nullptr, {0u, 0u, 0u}};
        auto take = [&]() {   // the prologue goes to whichever kernel is launched first
            const SahLevel r = first;
            first = rest;
            return r;
        };
        const uint32_t lvl = (uint32_t)level;
        if (b_big) {
            const uint32_t chunks = max_items / MPT_SAH_CHUNK + b_big;   // >= sum of ceil(items / chunk) over the level's big tasks
            hipLaunchKernelGGL(k_big_prep, dim3(1), dim3(1024), 0, stream, (const SahTask*)big_a, T.st, bigs, coff, take());
            hipLaunchKernelGGL(k_big_bounds, dim3(chunks), dim3(MPT_SAH_CHUNK_THREADS), 0, stream, (const float4*)it_lo_a, (const float4*)it_hi_a, (const SahTask*)big_a,
                               (const SahState*)T.st, (const uint32_t*)coff, bigs, lvl);
            hipLaunchKernelGGL(k_big_pick, dim3(b_big), dim3(MPT_SAH_BIG_THREADS), 0, stream, n, (const float4*)it_lo_a, (const float4*)it_hi_a, (const SahTask*)big_a, bigs,
                               T.st, T.child, T.lo, T.hi, lvl, T.first);
            hipLaunchKernelGGL(k_big_count, dim3(chunks), dim3(MPT_SAH_CHUNK_THREADS), 0, stream, (const float4*)it_lo_a, (const float4*)it_hi_a, (const SahTask*)big_a,
                               (const SahState*)T.st, (const uint32_t*)coff, (const SahBig*)bigs, chunk_left, lvl);
            hipLaunchKernelGGL(k_big_scatter, dim3(chunks), dim3(MPT_SAH_CHUNK_THREADS), 0, stream, (const float4*)it_lo_a, (const float4*)it_hi_a, it_lo_b, it_hi_b,
                               (const SahTask*)big_a, (const SahState*)T.st, (const uint32_t*)coff, bigs, (const uint32_t*)chunk_left, lvl);
            // ... and ONE launch for the children of the big tasks and the level's mid and small tasks (k_sah_tasks)
            constexpr uint32_t TH = 64u * MPT_SAH_WAVES;
            const uint32_t mid_blocks = (b_mid + MPT_SAH_WAVES - 1) / MPT_SAH_WAVES, small_blocks = (uint32_t)(((size_t)b_small * MPT_SAH_SMALL + TH - 1) / TH),
                           push_blocks = (b_big + MPT_SAH_WAVES - 1) / MPT_SAH_WAVES;
            hipLaunchKernelGGL(k_sah_tasks, dim3(mid_blocks + small_blocks + push_blocks), dim3(TH), 0, stream, n, (const float4*)it_lo_a, (const float4*)it_hi_a, it_lo_b,
                               it_hi_b, (const SahTask*)tasks_a, (const SahTask*)small_a, (const SahTask*)big_a, tasks_b, big_b, small_b, T.st, T.child, T.lo, T.hi,
                               (const uint32_t*)coff, (const SahBig*)bigs, (const uint32_t*)chunk_left, mid_blocks, small_blocks, take(), T.first);
        } else {
            if (b_mid)
                hipLaunchKernelGGL(k_sah_level, dim3((b_mid + MPT_SAH_WAVES - 1) / MPT_SAH_WAVES), dim3(64 * MPT_SAH_WAVES), 0, stream, n, (const float4*)it_lo_a,
                                   (const float4*)it_hi_a, it_lo_b, it_hi_b, (const SahTask*)tasks_a, tasks_b, big_b, small_b, T.st, T.child, T.lo, T.hi, take(), T.first);
            if (b_small || first.stamp != 0u)   // (a level with no grid at all still publishes its counts: zeroes, which end the loop)
                hipLaunchKernelGGL(k_sah_small, dim3((uint32_t)(((size_t)std::max(b_small, 1u) * MPT_SAH_SMALL + 255) / 256)), dim3(256), 0, stream, n, (const float4*)it_lo_a,
                                   (const float4*)it_hi_a, (const SahTask*)small_a, T.st, T.child, T.lo, T.hi, take(), T.first);
        }
        MPT_LB(hipGetLastError());
        // bounds of level + 1 from the newest exact counts: those of level - 1 at the latest (the device is past them or about to
        // be: the wait is short and the device keeps `level` in its queue meanwhile), those of `level` if they are there already
        uint32_t c[3];
        int from = level - 1;
        if (slot_ready(level)) from = level;
        if (from >= 0) {
            MPT_LB(wait_slot(from, c));
            known = from;
            if ((c[0] | c[1] | c[2]) == 0u) done = true;   // nothing was left at `from`: the levels behind it are empty too
            const uint32_t steps = (uint32_t)(level + 1 - from);   // 1 or 2 doublings
            const unsigned long long g = 1ull << steps, parents = (unsigned long long)c[0] + c[1];
            b_big = (uint32_t)std::min<unsigned long long>(g * c[1], max_big);
            b_mid = (uint32_t)std::min<unsigned long long>(g * parents, cap_mid);
            b_small = (uint32_t)std::min<unsigned long long>(g * parents, cap_small);
        } else {   // level 0 enqueued, nothing read yet: its (at most one) task has at most two children
            b_big = b_mid = b_small = 2u;
        }
        std::swap(tasks_a, tasks_b);
        std::swap(big_a, big_b);
        std::swap(small_a, small_b);
        std::swap(it_lo_a, it_lo_b);
        std::swap(it_hi_a, it_hi_b);
    }
    (void)known;
    if (!done) return hipErrorUnknown;
    return hipGetLastError();
}
}  // namespace mpt_sah

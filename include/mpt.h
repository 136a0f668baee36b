/* mpt.h — C ABI of the MI355X-native path-tracing hot path (libmpt_hip.so).
 *
 * The reference (omkhairate/MetalPathtracer) has no plugin/FFI layer: its seam is the
 * Renderer <-> fragment-shader binding contract (buffers 0..6 + two accumulation textures,
 * R/Renderer/Renderer.cpp:289-301 <-> R/Renderer/Shaders/Fragment.metal:10-18; byte layouts in
 * SURVEY.md App. D).  Every entry point below replaces one piece of that contract; the cited
 * file:line is the reference interface it stands in for.  R/ = "MetalCpp Path Tracer/".
 *
 * Conventions: plain pointers and sizes only; every call returns an mpt_status (0 = ok); no
 * exceptions, printf or assert cross the boundary (the reference printf+assert(false)s,
 * Renderer.cpp:87-91); a context is owned by one host thread and is not thread-safe (the
 * reference is single-threaded).  Host arrays passed in are copied during the
 * call and may be freed immediately (as Renderer.cpp:135,146,214-215 does).
 */
#ifndef MPT_H
#define MPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpt_ctx mpt_ctx;

typedef enum mpt_status {
    MPT_OK = 0,
    MPT_ERR_INVALID_ARG = 1,   /* null pointer, zero size, bad enum                                   */
    MPT_ERR_NO_DEVICE = 2,     /* no HIP device / ordinal out of range: the product never falls back   */
    MPT_ERR_HIP = 3,           /* a HIP runtime call failed (mpt_last_error has the text)              */
    MPT_ERR_BAD_SCENE = 4,     /* BVH/primitive arrays are not a well-formed tree (cycle, range, ...)  */
    MPT_ERR_NOT_READY = 5,     /* render before scene / uniforms / size were set                       */
    MPT_ERR_OVERFLOW = 6       /* internal ray-queue capacity exceeded (a bug; never expected)         */
} mpt_status;

/* UniformsData, 144 bytes — R/Renderer/Shaders/Structs.h:23-41 == R/Renderer/Renderer.cpp:12-28
 * (simd::float3 is 16-byte aligned; offsets in SURVEY.md App. D).                                  */
typedef struct mpt_uniforms {
    int32_t primitiveIndex;        /* off   0  unused by the shader                                  */
    int32_t _pad0[3];
    float cameraPosition[4];       /* off  16                                                        */
    float screenSize[2];           /* off  32                                                        */
    float _pad1[2];
    float viewportU[4];            /* off  48                                                        */
    float viewportV[4];            /* off  64                                                        */
    float firstPixelPosition[4];   /* off  80                                                        */
    float randomSeed[4];           /* off  96  literal RNG: sin-hash parameters (Random.h:32-35)     */
    uint64_t primitiveCount;       /* off 112  material-index guard (PathTracing.h:234-236)          */
    uint64_t triangleCount;        /* off 120  unused by the shader                                  */
    uint64_t frameCount;           /* off 128  running-mean weight (Fragment.metal:23,63)            */
    uint64_t totalPrimitiveCount;  /* off 136  never set by the reference                            */
} mpt_uniforms;

enum { MPT_RNG_LITERAL = 0,  /* bit-faithful to the reference's stuck PCG stream      */
       MPT_RNG_PHILOX = 1 }; /* Philox4x32-10, counter (pixel, sample, bounce, 0): the benchmark RNG  */
enum { MPT_BSDF_LAMBERT = 0, /* what rayColor executes (PathTracing.h:251-255)                        */
       MPT_BSDF_SCATTER = 1, /* + mirror (materialType < 0) / dielectric (> 0 = IOR) per Scatter.h:28-40, which is
                                  dead code in the reference; diffuse surfaces (materialType == 0) keep rayColor's own
                                  bounce (PathTracing.h:251-255).  Own choices, DESIGN.md "RNG / math specification":
                                  pow(x, 5) is a multiply chain and a transmitted ray starts at p - 1e-4 n (the
                                  reference's + n would re-hit the surface).                                        */
       MPT_BSDF_SCATTER_ALL = 2 }; /* scatter() for every material: as MPT_BSDF_SCATTER, and diffuse surfaces take
                                  Scatter.h's own Lambert branch too (Scatter.h:24-27,42: normalize(normal +
                                  normalize(randomFloat3(seed))), a point of the cube [-1,1]^3 per Random.h:18-30 —
                                  literal RNG: three draws from a copy of the stuck seed; philox: words 0, 1, 2 of
                                  the bounce's block)                                                               */
enum { MPT_PIPE_WAVEFRONT = 0,  /* global SoA ray queues + wave64 ballot compaction, one kernel/bounce */
       MPT_PIPE_MEGAKERNEL = 1, /* one thread per path, whole bounce loop in registers                */
       MPT_PIPE_WAVELOCAL = 2,  /* persistent waves, wave-private ray rings + ballot compaction; pipelines 0-2 walk
                                   the BVH in the reference's own order (PathTracing.h:188-193)                      */
       MPT_PIPE_ORDERED = 3,    /* the same wave-local wavefront over the product's own 4-wide BVH, closest child
                                   first, with the reference-order walk for the rays whose answer could depend on the
                                   order (falls back to pipeline 2 when a scene's child boxes are not nested in their
                                   parents' or it has more than 16 spheres).  EXACTNESS: it returns the reference's closest
                                   hit for every ray EXCEPT where the reference's own answer is an artefact of its float
                                   arithmetic: a triangle accepted at a computed t that lies in FRONT of the triangle's
                                   bounding box by more than t * 2^-10 — possible only when the ray lies within about
                                   1e-5 / |e1 x e2| radians of the triangle's plane, so that |det| is just above the 1e-5
                                   of PathTracing.h:153 and t = f * dot(e2, q) has lost its digits.  The closest-first
                                   walk culls the box of such a triangle by distance and never computes that t; the
                                   reference does if it happens to visit the leaf first.  Measured: 0 such rays in 1.3e11
                                   rays of rendering (scene.xml, bunny x20, height fields, Cornell); 6e-6 of the rays
                                   AIMED along the planes of 1..30-unit slivers (tests/test_gpu_adversarial.py, which
                                   checks in exact arithmetic that every difference is of this kind).  No affordable
                                   rule closes the gap: pipelines 0-2 reproduce the artefacts too.        */
       MPT_PIPE_AUTO = 4 };     /* pipeline 3 for scenes of MPT_AUTO_ORDERED_PRIMS (8192) primitives or more — where it
                                   is 1.5-2.2x faster — and pipeline 2 below that (scene.xml: pipeline 2 leads by a few
                                   per cent since its box-test loop was rewritten for the scalar unit) and ALWAYS with
                                   MPT_RNG_LITERAL, the mode that exists to reproduce the reference's frames.
                                   mpt_accel_info out[7] tells which of the two AUTO stands for (philox) with the
                                   uploaded scene.                                                                     */
#define MPT_AUTO_ORDERED_PRIMS 8192u

typedef struct mpt_render_params {
    int32_t rng_mode;        /* MPT_RNG_*                                                             */
    int32_t bsdf_mode;       /* MPT_BSDF_*                                                            */
    int32_t max_depth;       /* reference: 32 (PathTracing.h:216)                                     */
    int32_t pipeline;        /* MPT_PIPE_*                                                            */
    uint32_t sample_begin;   /* first sample index of this call (philox counter word 1)               */
    uint32_t sample_count;   /* samples per pixel rendered by this call                               */
    uint32_t seed_lo, seed_hi; /* philox key                                                          */
    int32_t shard_rank;      /* this GPU renders 8x8 pixel tiles t with t % shard_count == shard_rank */
    int32_t shard_count;     /* 1 = whole image                                                       */
    uint32_t slots_per_iter; /* wavefront width (ray slots per iteration); 0 = default                */
    uint32_t flags;          /* MPT_FLAG_*                                                            */
} mpt_render_params;

enum { MPT_FLAG_COUNT_WORK = 1u }; /* also count node visits / primitive tests (slower; for tests)    */

typedef struct mpt_stats {   /* cumulative since mpt_reset_stats                                      */
    uint64_t paths;          /* primary rays generated                                                */
    uint64_t rays;           /* closest-hit queries (primary + bounce)                                */
    uint64_t node_visits;    /* BVH nodes box-tested       (only with MPT_FLAG_COUNT_WORK)            */
    uint64_t aabb_hits;      /* box tests passed           (only with MPT_FLAG_COUNT_WORK)            */
    uint64_t prim_tests;     /* sphere + triangle tests    (only with MPT_FLAG_COUNT_WORK)            */
    uint64_t iterations;     /* wavefront iterations launched                                         */
    double trace_kernel_ms;  /* HIP-event time of the trace/shade kernels of the last mpt_render      */
    double total_ms;         /* HIP-event time of the whole last mpt_render (all kernels, its stream) */
    uint64_t trace_launches; /* trace/shade kernel launches in the last mpt_render                    */
    /* divergence diagnostics (only with MPT_FLAG_COUNT_WORK): loop trips per WAVE; 64 x trips = issued lane
     * slots, so node_visits / (64 * wave_node_iters) is the lane utilisation of the box-test loop            */
    uint64_t wave_node_iters;   /* box-test loop trips                                                        */
    uint64_t wave_prim_iters;   /* primitive-test loop trips                                                  */
    uint64_t wave_leaf_phases;  /* leaf phases entered                                                        */
    /* MPT_PIPE_ORDERED only */
    uint64_t exact_retraces;    /* rays handed to the reference-order walk (ties, inconsistent winners, ...)          */
    uint64_t tree_parked;       /* rays parked for a full-width tree step after the top test                          */
} mpt_stats;

/* Device selection / lifetime.  Replaces MTL::CreateSystemDefaultDevice + Renderer::Renderer /
 * ~Renderer resource ownership (R/Window/ApplicationDelegate.cpp:33, R/Renderer/Renderer.cpp:43-78). */
int mpt_create(int device_ordinal, mpt_ctx** out);
int mpt_destroy(mpt_ctx* ctx);
const char* mpt_last_error(const mpt_ctx* ctx);     /* text of the last failure on this context      */
const char* mpt_status_string(int status);

/* Fragment buffers 0 (bvhNodes), 1 (primitives), 2 (materials), 6 (primitiveIndices):
 * Renderer::updateVisibleScene / buildBuffers, R/Renderer/Renderer.cpp:127-146,199-215, fed with the
 * arrays Scene::createBVHBuffer / createTransformsBuffer / createMaterialsBuffer /
 * createPrimitiveIndexBuffer return (R/Scene/Scene.h:99-167).  bvh: 2 float4 per node; prims: 3
 * float4 per primitive; mats: 2 float4 per primitive; prim_idx: one int32 per primitive.            */
int mpt_upload_scene(mpt_ctx* ctx, const float* bvh, uint64_t n_nodes, const float* prims, const float* mats,
                     const int32_t* prim_idx, uint64_t n_prims);

/* Fragment buffer 3 (uniforms): Renderer::updateUniforms / recalculateViewport,
 * R/Renderer/Renderer.cpp:153-182,251-267.                                                          */
int mpt_set_uniforms(mpt_ctx* ctx, const mpt_uniforms* u);

/* Textures 0/1 (RGBA32F accumulation targets): Renderer::buildTextures / drawableSizeWillChange,
 * R/Renderer/Renderer.cpp:228-241,312-321.  Clears both targets and the HDR sum.                     */
int mpt_resize(mpt_ctx* ctx, uint32_t width, uint32_t height);

/* One reference frame: swap targets, run the hot path for 1 sample/pixel with the current
 * uniforms (frameCount as given), write the running mean into the current target —
 * Renderer::draw, R/Renderer/Renderer.cpp:269-310 + Fragment.metal:8-72.  Returns when the frame is complete
 * (its statistics are folded into mpt_get_stats); mpt_render_async is the commit()-style call.        */
int mpt_draw(mpt_ctx* ctx, const mpt_render_params* p);

/* Batch rendering (this project's extension of the same loop): adds, for every owned pixel, the
 * sum over [sample_begin, sample_begin+sample_count) of the per-sample clamped colour
 * (PathTracing.h:258) into the HDR sum buffer.  Synchronous; fills the timing fields of mpt_stats.  */
int mpt_render(mpt_ctx* ctx, const mpt_render_params* p);

/* The same, without waiting: checks the arguments, queues the render and RETURNS (microseconds: whatever a submission has to
 * wait for — a free render lane, the trace kernel before it becoming resident — is waited for on the context's own submit thread,
 * not on the caller's; at most 64 renders are queued, a 65th call waits for room).  Up to two renders are in flight on the device:
 * their trace kernels overlap — the next render fills the compute units that the previous one's tail and resolve leave idle —
 * while the updates of the HDR sum stay in submission order, so the result is bit-identical to consecutive mpt_render calls.
 * mpt_wait collects everything queued and in flight and reports the first failure of a queued render; statistics of
 * asynchronous renders appear in mpt_get_stats after they were collected (trace_kernel_ms / total_ms / trace_launches
 * then accumulate until mpt_reset_stats or the next mpt_render).  Every other call on the context first waits until the queue
 * has been submitted (and those that touch the scene, the size or the sum buffer until the renders are done): the context is
 * still driven by ONE caller thread.  This mirrors Metal's commit() without waitUntilCompleted (Renderer.cpp:253-266,307-308). */
int mpt_render_async(mpt_ctx* ctx, const mpt_render_params* p);
int mpt_wait(mpt_ctx* ctx);
/* Diagnostics of the asynchronous path: out4 = {renders submitted by the submit thread, submissions that found the trace kernel
 * before them resident (the residency gate), submissions that did not within 200 ms and fell back to the event chain (a foreign
 * kernel holds the chip), the longest mpt_render_async call so far in microseconds of host time}.                                */
int mpt_async_info(mpt_ctx* ctx, uint64_t out4[4]);

/* HDR sum buffer (RGBA32F, W*H*4 floats, row-major, top-left origin).  The pointer is device
 * memory on the context's device, e.g. for an RCCL reduce by the caller.  mpt_set_sum_buffer lets
 * the caller supply the storage (e.g. a torch tensor); pass NULL to return to the internal one.     */
int mpt_sum_buffer(mpt_ctx* ctx, void** device_ptr, uint64_t* bytes);
int mpt_set_sum_buffer(mpt_ctx* ctx, void* device_ptr);
int mpt_clear_sum(mpt_ctx* ctx);

/* Read-back (the reference never reads back).  read_frame: the current running-mean
 * target of mpt_draw.  read_sum: the raw HDR sum.  Both RGBA32F, W*H*4 floats.                        */
int mpt_read_frame(mpt_ctx* ctx, float* rgba_host);
int mpt_read_sum(mpt_ctx* ctx, float* rgba_host);
/* Checkpoint / resume of the accumulation (the reference keeps its running mean in a GPU-private texture and never reads it back,
 * R/Renderer/Renderer.cpp:236, Fragment.metal:62-69): mpt_read_sum is the checkpoint, mpt_write_sum puts it back.
 * Continuing with sample_begin = the number of samples the sum holds gives, bit for bit, the sum of an uninterrupted render.     */
int mpt_write_sum(mpt_ctx* ctx, const float* rgba_host);

int mpt_get_stats(mpt_ctx* ctx, mpt_stats* out);
int mpt_reset_stats(mpt_ctx* ctx);
void* mpt_stream(mpt_ctx* ctx);                      /* hipStream_t of uploads, clears, mpt_draw and serial mpt_render
                                                        (mpt_render_async alternates between this and a second stream) */
int mpt_synchronize(mpt_ctx* ctx);

/* Device-side closest-hit for a batch of rays (unit tests of firstHitBVH, PathTracing.h:75-204).
 * origins/directions: 3 floats per ray (host).  Outputs (host): t, primitive id (-1 = miss),
 * normal (3 floats, flipped to face the ray), front-face flag.                                       */
int mpt_trace_rays(mpt_ctx* ctx, const float* origins, const float* directions, uint64_t n_rays, float* t_out,
                   int32_t* prim_out, float* normal_out, int32_t* front_out);

/* The same through the closest-first walk of MPT_PIPE_ORDERED (must return exactly what mpt_trace_rays returns).
 * flags_out (host, one uint32 per ray): 0 = answered by the closest-first walk; otherwise the ray was re-traced in
 * reference order because of 1 a (nearly) zero / non-finite direction component or a far origin, 2 a tie between two
 * primitives, 4 a winner in front of its own reference leaf box, 8 stack overflow.                                   */
int mpt_trace_rays_ordered(mpt_ctx* ctx, const float* origins, const float* directions, uint64_t n_rays, float* t_out,
                           int32_t* prim_out, float* normal_out, int32_t* front_out, uint32_t* flags_out);

/* Shape of the product's own acceleration structure for the uploaded scene: out[0] = 1 if MPT_PIPE_ORDERED can be used,
 * [1] own 4-wide nodes, [2] its depth, [3] nodes staged in LDS, [4] spheres on the always list, [5] reference leaves,
 * [6] primitives staged in LDS, [7] the pipeline MPT_PIPE_AUTO resolves to for this scene (MPT_PIPE_WAVELOCAL or
 * MPT_PIPE_ORDERED).                                                                                                   */
int mpt_accel_info(mpt_ctx* ctx, uint64_t out[8]);

/* BVH construction on the GPU — stands where the reference has Scene::buildBVH / buildBVHRecursive (R/Scene/Scene.h:71-93,
 * 195-317: sequential full-sweep SAH, 8.2 s for 1 M primitives): a top-down binned SAH over the primitives, built level by
 * level on the device (16 bins over the box centres, cost = area * primitives: the tree of the host's binned builder), with
 * leaves of <= mpt_gpu_leaf_max(n_prims) primitives — 6 for scenes below MPT_AUTO_ORDERED_PRIMS primitives (the ones the
 * reference-order kernel renders: tree and hot primitives sit in LDS there), 2 from there on (the closest-first kernel tests
 * every primitive of a leaf it enters); MPT_LBVH_LEAF = 1..8 overrides — written in the REFERENCE's buffer format so that
 * mpt_upload_scene (and the reference's shader, and the oracle) can consume it: bvh_out = 2 float4 per node as
 * Scene::createBVHBuffer returns them (root = node 0), prim_idx_out = Scene::createPrimitiveIndexBuffer.  The same input
 * gives the same arrays on every call.  MPT_GPU_BUILD = ploc | lbvh selects the two earlier builders instead (63-bit
 * Morton codes + radix sort, then nearest-neighbour clustering or Karras' radix tree: slower to render by 1-16 %).
 * prims: the 3-float4-per-primitive array of Scene::createTransformsBuffer (host memory, already sorted spheres first as
 * Scene::buildBVH does, Scene.h:72-75).  bvh_capacity_nodes >= 2 * n_prims - 1 is always enough.  device_ms_out
 * (optional): HIP-event time of the build kernels.  Parent boxes are exact unions of child boxes.                      */
int mpt_build_bvh(mpt_ctx* ctx, const float* prims, uint64_t n_prims, float* bvh_out, uint64_t bvh_capacity_nodes,
                  uint64_t* n_nodes_out, int32_t* prim_idx_out, double* device_ms_out);

/* Build -> render without the host: the same tree, built on the device from the caller's primitive and material arrays
 * (prims: 3 float4 each as Scene::createTransformsBuffer returns them, mats: 2 float4 each as Scene::createMaterialsBuffer)
 * and turned ON THE DEVICE into everything mpt_upload_scene derives on the host — the threaded reference-order tree, the
 * leaf-ordered primitive records, the de-duplicated materials, the product's own 4-wide tree — so that the scene is ready
 * to render when the call returns (1 M primitives: 9 ms, 4 of them the upload; mpt_build_bvh + mpt_upload_scene: 0.55 s).  Stands
 * for Scene::buildBVH + the four packers + Renderer::updateVisibleScene / buildBuffers (R/Scene/Scene.h:71-93,99-167,195-317,
 * R/Renderer/Renderer.cpp:127-146,199-215).  mpt_download_bvh returns that tree in the REFERENCE's buffer format (as
 * mpt_build_bvh does): what the reference's shader — and the oracle — would walk to produce the same image.                 */
int mpt_build_and_upload(mpt_ctx* ctx, const float* prims, const float* mats, uint64_t n_prims, double* device_ms_out);
int mpt_download_bvh(mpt_ctx* ctx, float* bvh_out, uint64_t bvh_capacity_nodes, uint64_t* n_nodes_out, int32_t* prim_idx_out);

/* The leaf limit the GPU builders (mpt_build_bvh, mpt_build_and_upload) use for a scene of n_prims primitives: 6 below
 * MPT_AUTO_ORDERED_PRIMS, 2 from there on, or what MPT_LBVH_LEAF says.  A pure function (no context).                      */
int mpt_gpu_leaf_max(uint64_t n_prims);
/* What the last scene call left on the device: out[0] = primitives of the tree mpt_download_bvh would return (0 when the scene
 * came through mpt_upload_scene: MPT_ERR_NOT_READY there), [1] nodes of that tree, [2] the leaf limit it was built with,
 * [3] MPT_AUTO_ORDERED_PRIMS (the scene size from which MPT_PIPE_AUTO means the closest-first pipeline), [4] primitives of the
 * uploaded scene, [5] threaded reference-order nodes, [6] de-duplicated materials, [7] nodes of the own 4-wide tree that the
 * closest-first walk fetches in their float form because their boxes could not be quantised (degenerate input; normally 0).  */
int mpt_build_info(mpt_ctx* ctx, uint64_t out[8]);

/* Diagnostics: a position-sensitive 64-bit digest of every device array of the uploaded scene, computed on the device —
 * out[0..8] = threaded tree, primitive records, materials, own 4-wide tree, reference leaf boxes, per-primitive leaf boxes,
 * always list, reference-format tree, reference-format primitive indices (0 where the scene has none); out[9..15] = the
 * counts (nodes, primitives, materials, own nodes, reference leaves, always-list entries, own-tree depth).  Two builds of the
 * same input must give the same 16 words (the tests' "same arrays" check covers what mpt_download_bvh does not return).
 * No reference counterpart (the reference builds once, on the host: R/Scene/Scene.h:71-93).                                */
int mpt_scene_digest(mpt_ctx* ctx, uint64_t out[16]);

/* ---- multi-GPU: tile shards + ONE RCCL reduce of the HDR sum over xGMI ---------------------------------
 * The reference is single-GPU (it presents straight to the drawable, R/Renderer/Renderer.cpp:303-307); this is the
 * product's extension.  Every GPU renders the 8x8 pixel tiles t % N == rank (mpt_render_params.shard_rank / shard_count)
 * at full spp into its own zero-initialised HDR sum; mpt_reduce_sum adds the N buffers onto the root's with
 * ncclReduce(sum, float32, 4*W*H) — each pixel has exactly one owner, so the result is bit-identical to one GPU's.
 * librccl.so is opened on first use (dlopen): nothing here needs it for N = 1, where the reduce is a no-op.
 *   one host thread, N contexts:   mpt_comm_create_all(ctxs, N, &comm)            (ncclCommInitAll)
 *   one process per GPU:           rank 0: mpt_comm_unique_id(id); every rank: mpt_comm_create_rank(ctx, r, N, id, &comm)
 * All contexts of a communicator must have the same size (mpt_resize).                                                  */
typedef struct mpt_comm mpt_comm;
#define MPT_COMM_ID_BYTES 128
int mpt_comm_unique_id(void* id_out /* MPT_COMM_ID_BYTES */);
int mpt_comm_create_all(mpt_ctx* const* ctxs, int n, mpt_comm** out);
int mpt_comm_create_rank(mpt_ctx* ctx, int rank, int nranks, const void* id /* MPT_COMM_ID_BYTES */, mpt_comm** out);
int mpt_reduce_sum(mpt_comm* comm, int root);   /* waits for the renders in flight, reduces, returns when the root holds the image */
int mpt_comm_destroy(mpt_comm* comm);
const char* mpt_comm_last_error(const mpt_comm* comm);

/* RNG known-answer hooks evaluated ON THE DEVICE (Random.h:6-16 and the philox / sincos spec).      */
int mpt_kat_pcg(mpt_ctx* ctx, const uint32_t* seeds, uint64_t n, uint32_t* hash_out, float* float_out);
int mpt_kat_philox(mpt_ctx* ctx, const uint32_t* ctr4, const uint32_t* key2, uint64_t n, uint32_t* out4);
int mpt_kat_sincos(mpt_ctx* ctx, const float* u, uint64_t n, float* sin_out, float* cos_out);
/* The reciprocal of the hot path (1.0 / r.direction[i], PathTracing.h:61; 1 / a of the triangle test, :153-165): the kernels' short
 * form (v_rcp_f32 + the compiler's own fma chain, without v_div_scale / v_div_fixup) against the correctly rounded division, over ALL
 * 2^32 operands, on the device.  out4 = {mismatches inside the range the kernels use the short form in, operands in that range,
 * mismatches outside it, operands outside it}; out4[0] must be 0.                                                                   */
int mpt_kat_rcp(mpt_ctx* ctx, uint64_t* out4);

#ifdef __cplusplus
}
#endif
#endif /* MPT_H */

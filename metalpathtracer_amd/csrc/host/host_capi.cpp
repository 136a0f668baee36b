// host_capi.cpp — extern "C" surface of the host layer (include/mpt_host.h).
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <memory>
#include <cstring>
#include <exception>
#include <limits>
#include <string>
#include <vector>

#include "Camera.h"
#include "Renderer.h"
#include "Scene.h"
#include "SceneLoader.h"
#include "mpt_host.h"

using namespace MetalCppPathTracer;

struct mpt_scene {
    Scene* sc = nullptr;
    bool owned = false;
};
struct mpt_renderer {
    Renderer* r = nullptr;
    mpt_scene scene_view;  // borrowed view of the renderer's scene, see mpt_renderer_scene
};

static void copy_text(const std::string& s, char* dst, size_t cap) {
    if (!dst || cap == 0) return;
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(dst, s.data(), n);
    dst[n] = '\0';
}

extern "C" {

// The scene entry points allocate (std::vector growth, new[]): no exception may cross the C ABI (include/mpt_host.h).
#define SCENE_GUARD(fail_value, ...)  \
    try {                             \
        __VA_ARGS__                   \
    } catch (...) {                   \
        return fail_value;            \
    }

int mpt_scene_create(mpt_scene** out) {
    if (!out) return MPT_ERR_INVALID_ARG;
    *out = nullptr;
    SCENE_GUARD(MPT_ERR_HIP, {
        mpt_scene* h = new mpt_scene();
        h->sc = nullptr;
        h->owned = true;
        try {
            h->sc = new Scene();
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
        return MPT_OK;
    })
}
int mpt_scene_destroy(mpt_scene* s) {
    if (!s || !s->owned) return MPT_ERR_INVALID_ARG;
    delete s->sc;
    delete s;
    return MPT_OK;
}
int mpt_scene_clear(mpt_scene* s) {
    if (!s) return MPT_ERR_INVALID_ARG;
    SCENE_GUARD(MPT_ERR_HIP, {
        s->sc->clear();
        return MPT_OK;
    })
}
int mpt_scene_load_xml(mpt_scene* s, const char* xml_path, const char* asset_root, char* log, size_t log_cap) {
    if (!s || !xml_path) return -1;
    SCENE_GUARD(-1, {
        std::string text;
        int st = SceneLoader::Load(xml_path, s->sc, asset_root ? asset_root : "", &text);
        copy_text(text, log, log_cap);
        return st;
    })
}
int mpt_scene_add_primitive(mpt_scene* s, int type, const float d0[3], const float d1[3], const float d2[3],
                            const float mat[8]) {
    if (!s || !d0 || !d1 || !d2 || !mat || (type != 0 && type != 1)) return MPT_ERR_INVALID_ARG;
    Primitive p;
    p.type = type == 0 ? PrimitiveType::Sphere : PrimitiveType::Triangle;
    p.data0 = mpt::float3(d0[0], d0[1], d0[2]);
    p.data1 = mpt::float3(d1[0], d1[1], d1[2]);
    p.data2 = mpt::float3(d2[0], d2[1], d2[2]);
    p.material.albedo = mpt::float3(mat[0], mat[1], mat[2]);
    p.material.materialType = mat[3];
    p.material.emissionColor = mpt::float3(mat[4], mat[5], mat[6]);
    p.material.emissionPower = mat[7];
    SCENE_GUARD(MPT_ERR_HIP, {
        s->sc->addPrimitive(p);
        return MPT_OK;
    })
}
int mpt_scene_build_bvh(mpt_scene* s, int mode) {
    if (!s || mode < 0 || mode > 2) return MPT_ERR_INVALID_ARG;
    SCENE_GUARD(MPT_ERR_HIP, {
        s->sc->buildBVH(mode == 0 ? Scene::BuildMode::ReferenceSweep : mode == 1 ? Scene::BuildMode::BinnedCentroid : Scene::BuildMode::GpuLbvh);
        return MPT_OK;
    })
}
int mpt_scene_sort_primitives(mpt_scene* s) {
    if (!s) return MPT_ERR_INVALID_ARG;
    SCENE_GUARD(MPT_ERR_HIP, {
        s->sc->sortPrimitives();
        return MPT_OK;
    })
}
int mpt_scene_counts(const mpt_scene* s, uint64_t* prims, uint64_t* triangles, uint64_t* nodes, int32_t* depth) {
    if (!s) return MPT_ERR_INVALID_ARG;
    const Scene& sc = *s->sc;
    if (prims) *prims = sc.getPrimitiveCount();
    if (triangles) *triangles = sc.getTriangleCount();
    if (nodes) *nodes = sc.getBVHNodeCount();
    if (depth) *depth = sc.getBVHDepth();
    return MPT_OK;
}
int mpt_scene_copy_buffers(const mpt_scene* s, float* bvh, float* prims, float* mats, int32_t* prim_idx) {
    if (!s) return MPT_ERR_INVALID_ARG;
    const Scene& sc = *s->sc;
    const size_t P = sc.getPrimitiveCount(), N = sc.getBVHNodeCount();
    SCENE_GUARD(MPT_ERR_HIP, {
    if (bvh) {
        mpt::float4* b = sc.createBVHBuffer();
        std::memcpy(bvh, b, N * 32);
        delete[] b;
    }
    if (prims) {
        mpt::float4* b = sc.createTransformsBuffer();
        std::memcpy(prims, b, P * 48);
        delete[] b;
    }
    if (mats) {
        mpt::float4* b = sc.createMaterialsBuffer();
        std::memcpy(mats, b, P * 32);
        delete[] b;
    }
    if (prim_idx) {
        // the index array exists only after buildBVH (R/Scene/Scene.h:157-167 sizes it by primitiveIndices); before
        // that the caller's P entries are filled with the identity order the builder would start from
        const size_t have = sc.getPrimitiveIndices().size();
        int* b = sc.createPrimitiveIndexBuffer();
        std::memcpy(prim_idx, b, std::min(have, P) * 4);
        delete[] b;
        for (size_t i = have; i < P; ++i) prim_idx[i] = static_cast<int32_t>(i);
    }
    return MPT_OK;
    })
}

int mpt_camera_reset_values(float pos[3], float fwd[3], float up[3], float* vfov_deg) {
    if (!pos || !fwd || !up || !vfov_deg) return MPT_ERR_INVALID_ARG;
    pos[0] = 0.0f; pos[1] = 20.0f; pos[2] = 50.0f;
    fwd[0] = 0.0f; fwd[1] = 0.0f; fwd[2] = -1.0f;
    up[0] = 0.0f; up[1] = 1.0f; up[2] = 0.0f;
    *vfov_deg = 60.0f;
    return MPT_OK;
}
int mpt_camera_viewport(const float pos[3], const float fwd[3], const float up[3], float vfov_deg, float width,
                        float height, mpt_uniforms* u) {
    if (!pos || !fwd || !up || !u || !(width > 0) || !(height > 0)) return MPT_ERR_INVALID_ARG;
    const float aspect = width / height;
    const float fovRad = vfov_deg * (M_PI / 180.0f);
    const float halfH = tanf(fovRad * 0.5f);
    const float halfW = aspect * halfH;
    const mpt::float3 P(pos[0], pos[1], pos[2]), F(fwd[0], fwd[1], fwd[2]), U(up[0], up[1], up[2]);
    const mpt::float3 w = mpt::normalize(-F);
    const mpt::float3 uu = mpt::normalize(mpt::cross(U, w));
    const mpt::float3 vv = mpt::cross(w, uu);
    const mpt::float3 vu = uu * (2.0f * halfW);
    const mpt::float3 vvv = (-vv) * (2.0f * halfH);
    const mpt::float3 first = P - w - (vu * 0.5f) - (vvv * 0.5f);
    u->cameraPosition[0] = P.x; u->cameraPosition[1] = P.y; u->cameraPosition[2] = P.z;
    u->viewportU[0] = vu.x; u->viewportU[1] = vu.y; u->viewportU[2] = vu.z;
    u->viewportV[0] = vvv.x; u->viewportV[1] = vvv.y; u->viewportV[2] = vvv.z;
    u->firstPixelPosition[0] = first.x; u->firstPixelPosition[1] = first.y; u->firstPixelPosition[2] = first.z;
    u->screenSize[0] = width;
    u->screenSize[1] = height;
    return MPT_OK;
}
float mpt_host_random_float(uint32_t* state) {
    const uint32_t st = *state * 747796405u + 2891336453u;
    const uint32_t word = ((st >> ((st >> 28u) + 4u)) ^ st);
    *state = (word >> 22u) ^ word;
    return static_cast<float>(*state) / static_cast<float>(std::numeric_limits<uint32_t>::max());
}

// ---- Renderer ------------------------------------------------------------------------------------------
#define GUARD(body)                          \
    try {                                    \
        body;                                \
        return MPT_OK;                       \
    } catch (const std::exception& e) {      \
        std::fprintf(stderr, "[mpt] %s\n", e.what()); \
        return MPT_ERR_HIP;                  \
    }

int mpt_renderer_create(int device, const char* xml_path, const char* asset_root, mpt_renderer** out, char* err,
                        size_t err_cap) {
    if (!out) return MPT_ERR_INVALID_ARG;
    *out = nullptr;
    try {
        std::unique_ptr<Renderer> r(new Renderer(device, xml_path ? xml_path : "", asset_root ? asset_root : ""));
        std::unique_ptr<mpt_renderer> h(new mpt_renderer());
        h->r = r.release();
        *out = h.release();
        return MPT_OK;
    } catch (const std::exception& e) {
        copy_text(e.what(), err, err_cap);
        return MPT_ERR_NO_DEVICE;
    }
}
int mpt_renderer_destroy(mpt_renderer* r) {
    if (!r) return MPT_ERR_INVALID_ARG;
    delete r->r;
    delete r;
    return MPT_OK;
}
int mpt_renderer_drawable_size_will_change(mpt_renderer* r, uint32_t width, uint32_t height) {
    if (!r || !width || !height) return MPT_ERR_INVALID_ARG;
    GUARD(r->r->drawableSizeWillChange(nullptr, DrawableSize{(double)width, (double)height}));
}
int mpt_renderer_set_params(mpt_renderer* r, const mpt_render_params* p) {
    if (!r || !p) return MPT_ERR_INVALID_ARG;
    r->r->setRenderParams(*p);
    return MPT_OK;
}
int mpt_renderer_draw(mpt_renderer* r) {
    if (!r) return MPT_ERR_INVALID_ARG;
    GUARD(r->r->draw(nullptr));
}
int mpt_renderer_input(mpt_renderer* r, const float move[3], const float rotate[2], float zoom, int reset) {
    if (!r) return MPT_ERR_INVALID_ARG;
    if (move) InputSystem::movementInput = mpt::float3(move[0], move[1], move[2]);
    if (rotate) InputSystem::rotationInput = mpt::float2{rotate[0], rotate[1]};
    InputSystem::zoomInput = zoom;
    InputSystem::resetInput = reset != 0;
    return MPT_OK;
}
int mpt_renderer_read_frame(mpt_renderer* r, float* rgba) {
    if (!r || !rgba) return MPT_ERR_INVALID_ARG;
    return mpt_read_frame(r->r->context(), rgba);
}
int mpt_renderer_render_batch(mpt_renderer* r, uint32_t sample_begin, uint32_t sample_count) {
    if (!r) return MPT_ERR_INVALID_ARG;
    GUARD(r->r->renderBatch(sample_begin, sample_count));
}
int mpt_renderer_read_sum(mpt_renderer* r, float* rgba) {
    if (!r || !rgba) return MPT_ERR_INVALID_ARG;
    return mpt_read_sum(r->r->context(), rgba);
}
int mpt_renderer_clear_sum(mpt_renderer* r) {
    if (!r) return MPT_ERR_INVALID_ARG;
    return mpt_clear_sum(r->r->context());
}
int mpt_renderer_uniforms(mpt_renderer* r, mpt_uniforms* out) {
    if (!r || !out) return MPT_ERR_INVALID_ARG;
    *out = r->r->uniforms();
    return MPT_OK;
}
int mpt_renderer_stats(mpt_renderer* r, mpt_stats* out) {
    if (!r || !out) return MPT_ERR_INVALID_ARG;
    return mpt_get_stats(r->r->context(), out);
}
mpt_ctx* mpt_renderer_context(mpt_renderer* r) { return r ? r->r->context() : nullptr; }
mpt_scene* mpt_renderer_scene(mpt_renderer* r) {
    if (!r) return nullptr;
    r->scene_view.sc = r->r->scene();
    r->scene_view.owned = false;
    return &r->scene_view;
}

// ---- image output ------------------------------------------------------------------------------------------
int mpt_write_pfm(const char* path, const float* rgba, uint32_t width, uint32_t height, float scale) {
    if (!path || !rgba || !width || !height) return MPT_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return MPT_ERR_INVALID_ARG;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);  // little-endian, rows bottom-to-top
    std::vector<float> row(static_cast<size_t>(width) * 3);
    for (uint32_t y = height; y-- > 0;) {
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) row[3 * x + c] = rgba[4 * (static_cast<size_t>(y) * width + x) + c] * scale;
        std::fwrite(row.data(), 4, row.size(), f);
    }
    std::fclose(f);
    return MPT_OK;
}
int mpt_write_ppm(const char* path, const float* rgba, uint32_t width, uint32_t height, float scale, float gamma) {
    if (!path || !rgba || !width || !height) return MPT_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return MPT_ERR_INVALID_ARG;
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<unsigned char> row(static_cast<size_t>(width) * 3);
    const float inv = gamma > 0.0f ? 1.0f / gamma : 1.0f;
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) {
                float v = rgba[4 * (static_cast<size_t>(y) * width + x) + c] * scale;
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                v = std::pow(v, inv);
                row[3 * x + c] = static_cast<unsigned char>(v * 255.0f + 0.5f);
            }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return MPT_OK;
}

}  // extern "C"

// Renderer.h — host renderer over the HIP C ABI (include/mpt.h).
//
// Public method names and call order are the reference's class Renderer (R/Renderer/Renderer.h:16-29):
//   Renderer(device) -> updateVisibleScene, buildShaders, buildBuffers, buildTextures, recalculateViewport;
//   per frame draw(view) -> updateUniforms -> (swap targets, bind, launch).
// MTL::Device* becomes a HIP device ordinal; MTK::View* becomes an OffscreenView (the reference renders into
// an MTKView drawable and never reads back).  Everything device-side goes through libmpt_hip.so;
// this class contains no tracing code and there is no CPU fallback: construction throws if no GPU is found.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "Scene.h"
#include "mpt.h"

namespace MetalCppPathTracer {

struct OffscreenView {  // stands in for MTK::View: the size of the drawable + where a frame lands
    uint32_t width = 1280, height = 720;
    std::vector<float> rgba;  // filled by Renderer::readFrame (RGBA32F, top-left origin)
};
struct DrawableSize {
    double width, height;
};

class Renderer {
public:
    // tree builder used by updateVisibleScene: the reference's own (the drop-in default), or one of the product's
    enum { BUILD_REFERENCE = 0, BUILD_BINNED = 1, BUILD_GPU = 2, BUILD_AUTO = 3 };
    explicit Renderer(int deviceOrdinal = 0, const std::string& scenePath = std::string(),
                      const std::string& assetRoot = std::string(), int buildMode = BUILD_REFERENCE);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void updateVisibleScene();   // load XML, build BVH, upload BVH + index buffers, then buildBuffers()
    void buildShaders();         // the kernels are precompiled in libmpt_hip.so: verifies the context only
    void buildBuffers();         // upload primitive + material buffers; uniforms start zero-filled
    void buildTextures();        // two RGBA32F accumulation targets at Camera::screenSize
    void recalculateViewport();  // R/Renderer/Renderer.cpp:153-182
    bool updateCamera();
    void updateUniforms();       // R/Renderer/Renderer.cpp:251-267 (frameCount / randomSeed protocol)
    void draw(OffscreenView* view);
    void drawableSizeWillChange(OffscreenView* view, DrawableSize size);

    // ---- extensions (not in the reference) ----
    void setScenePath(const std::string& xml, const std::string& assetRoot = std::string());
    void setRenderParams(const mpt_render_params& p) { params_ = p; }
    void setBuildMode(int mode) { buildMode_ = mode; }   // takes effect at the next updateVisibleScene()
    mpt_render_params& renderParams() { return params_; }
    Scene* scene() { return scene_; }
    mpt_ctx* context() { return ctx_; }
    const mpt_uniforms& uniforms() const { return uniforms_; }
    void readFrame(OffscreenView* view);                       // running-mean target of draw()
    int renderBatch(uint32_t sampleBegin, uint32_t sampleCount); // HDR sum accumulation (mpt_render)
    void readSum(std::vector<float>& rgba);
    void writeSum(const std::vector<float>& rgba);   // checkpoint / resume: the inverse of readSum (mpt_write_sum)
    void clearSum();
    mpt_stats stats();

private:
    void check(int status, const char* where);
    float hostRandomFloat();

    int device_ = 0;
    mpt_ctx* ctx_ = nullptr;
    Scene* scene_ = nullptr;
    std::string scenePath_, assetRoot_;
    mpt_uniforms uniforms_;
    mpt_render_params params_;
    uint32_t hostSeed_ = 92407235u;  // R/Renderer/Renderer.cpp:32
    bool sceneUploaded_ = false;
    int buildMode_ = BUILD_REFERENCE;
    bool deviceBuild_ = false, deviceDirty_ = false;   // BUILD_GPU: mpt_build_and_upload, no tree on the host
};

}  // namespace MetalCppPathTracer

// main.cpp — mpt_render: headless command-line front end of the host Renderer.
// (The reference's main.cpp starts an NSApplication + MTKView, R/main.cpp:15-28; that shell is out of scope.)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "Camera.h"
#include "Renderer.h"
#include "mpt_host.h"

using namespace MetalCppPathTracer;

static void usage() {
    std::puts(
        "mpt_render --scene scene.xml [--asset-root DIR] [--width 1280] [--height 720]\n"
        "           [--spp 64] [--depth 32] [--seed 1] [--rng philox|literal] [--bsdf lambert|scatter]\n"
        "           [--pipeline wavelocal|wavefront|megakernel] [--frames N] [--device 0] [--out image.pfm|image.ppm]\n"
        "           [--camera-pos x,y,z] [--camera-dir x,y,z] [--camera-up x,y,z] [--vfov degrees]\n"
        "  --frames N   run the reference's frame protocol (N draw() calls, running mean) instead of batch spp");
}

int main(int argc, char** argv) {
    std::string scene, assetRoot, out;
    int width = 1280, height = 720, spp = 64, depth = 32, device = 0, frames = 0;
    unsigned seed = 1;
    float camPos[3], camDir[3], camUp[3], vfov = 0.0f;
    bool havePos = false, haveDir = false, haveUp = false;
    mpt_render_params prm;
    std::memset(&prm, 0, sizeof prm);
    prm.rng_mode = MPT_RNG_PHILOX;
    prm.shard_count = 1;
    prm.pipeline = MPT_PIPE_ORDERED;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) {
                usage();
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--scene") scene = next();
        else if (a == "--asset-root") assetRoot = next();
        else if (a == "--width") width = std::atoi(next());
        else if (a == "--height") height = std::atoi(next());
        else if (a == "--spp") spp = std::atoi(next());
        else if (a == "--depth") depth = std::atoi(next());
        else if (a == "--seed") seed = static_cast<unsigned>(std::strtoul(next(), nullptr, 10));
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--camera-pos" || a == "--camera-dir" || a == "--camera-up") {
            float v[3] = {0, 0, 0};
            if (std::sscanf(next(), "%f,%f,%f", &v[0], &v[1], &v[2]) != 3) {
                usage();
                return 2;
            }
            float* dst = a == "--camera-pos" ? camPos : (a == "--camera-dir" ? camDir : camUp);
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
            (a == "--camera-pos" ? havePos : (a == "--camera-dir" ? haveDir : haveUp)) = true;
        }
        else if (a == "--vfov") vfov = static_cast<float>(std::atof(next()));
        else if (a == "--rng") prm.rng_mode = std::strcmp(next(), "literal") == 0 ? MPT_RNG_LITERAL : MPT_RNG_PHILOX;
        else if (a == "--bsdf") prm.bsdf_mode = std::strcmp(next(), "scatter") == 0 ? MPT_BSDF_SCATTER : MPT_BSDF_LAMBERT;
        else if (a == "--pipeline") {
            const char* v = next();
            prm.pipeline = std::strcmp(v, "megakernel") == 0 ? MPT_PIPE_MEGAKERNEL
                           : std::strcmp(v, "wavefront") == 0 ? MPT_PIPE_WAVEFRONT
                           : std::strcmp(v, "wavelocal") == 0 ? MPT_PIPE_WAVELOCAL : MPT_PIPE_ORDERED;
        }
        else {
            usage();
            return a == "--help" ? 0 : 2;
        }
    }
    if (scene.empty()) {
        usage();
        return 2;
    }
    prm.max_depth = depth;
    prm.seed_lo = seed;
    try {
        Renderer r(device, scene, assetRoot);
        r.setRenderParams(prm);
        // the reference hard-codes Camera::reset(); the flags overwrite the same globals before the viewport is built
        if (havePos) Camera::position = mpt::float3(camPos[0], camPos[1], camPos[2]);
        if (haveDir) Camera::forward = mpt::normalize(mpt::float3(camDir[0], camDir[1], camDir[2]));
        if (haveUp) Camera::up = mpt::normalize(mpt::float3(camUp[0], camUp[1], camUp[2]));
        if (vfov > 0.0f) Camera::verticalFov = vfov;
        OffscreenView view;
        r.drawableSizeWillChange(&view, DrawableSize{(double)width, (double)height});
        std::vector<float> img;
        float scale = 1.0f;
        auto t0 = std::chrono::steady_clock::now();
        if (frames > 0) {
            for (int f = 0; f < frames; ++f) r.draw(&view);
            r.readFrame(&view);
            img = view.rgba;
        } else {
            r.clearSum();
            r.renderBatch(0, static_cast<uint32_t>(spp));
            r.readSum(img);
            scale = 1.0f / static_cast<float>(spp);
        }
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mpt_stats st = r.stats();
        std::printf("{\"paths\": %llu, \"rays\": %llu, \"seconds\": %.6f, \"device_ms\": %.3f, \"mrays_per_s\": %.1f}\n",
                    (unsigned long long)st.paths, (unsigned long long)st.rays, sec, st.total_ms,
                    st.total_ms > 0 ? st.rays / st.total_ms / 1e3 : 0.0);
        if (!out.empty()) {
            bool ppm = out.size() > 4 && out.substr(out.size() - 4) == ".ppm";
            int rc = ppm ? mpt_write_ppm(out.c_str(), img.data(), width, height, scale, 2.2f)
                         : mpt_write_pfm(out.c_str(), img.data(), width, height, scale);
            if (rc) {
                std::fprintf(stderr, "cannot write %s\n", out.c_str());
                return 1;
            }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "mpt_render: %s\n", e.what());
        return 1;
    }
    return 0;
}

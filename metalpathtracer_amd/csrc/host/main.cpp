// main.cpp — mpt_render: headless command-line front end of the host Renderer.
// (The reference's main.cpp starts an NSApplication + MTKView, R/main.cpp:15-28; that shell is out of scope.)
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "Camera.h"
#include "Renderer.h"
#include "mpt_host.h"

using namespace MetalCppPathTracer;

// what a checkpoint was accumulated FROM: FNV-1a over the packed primitive and material arrays (Scene::create*Buffer, spheres first)
static unsigned long long sceneFingerprint(const Scene& sc) {
    unsigned long long h = 0xcbf29ce484222325ull;
    auto eat = [&](const void* p, size_t bytes) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 0x100000001b3ull;
    };
    const size_t n = sc.getPrimitiveCount();
    mpt::float4* prims = sc.createTransformsBuffer();
    mpt::float4* mats = sc.createMaterialsBuffer();
    eat(&n, sizeof n);
    if (prims) eat(prims, n * 3 * sizeof(mpt::float4));
    if (mats) eat(mats, n * 2 * sizeof(mpt::float4));
    delete[] prims;
    delete[] mats;
    return h;
}
static void usage() {
    std::puts(
        "mpt_render --scene scene.xml [--asset-root DIR] [--width 1280] [--height 720]\n"
        "           [--spp 64] [--depth 32] [--seed 1] [--rng philox|literal] [--bsdf lambert|scatter|scatter-all]\n"
        "           [--pipeline auto|ordered|wavelocal|wavefront|megakernel] [--frames N] [--device 0] [--out image.pfm|image.ppm]\n"
        "           [--camera-pos x,y,z] [--camera-dir x,y,z] [--camera-up x,y,z] [--vfov degrees]\n"
        "           [--gpus N | --devices a,b,...] [--camera-path FILE [--out-dir runs]] [--bvh reference|binned|gpu|auto]\n"
        "           [--checkpoint FILE] [--resume FILE]\n"
        "  --bvh             tree builder: the reference's sweep SAH (default with --rng literal, --frames and --camera-path:\n"
        "                    the drop-in behaviour) or auto (default for batch renders: the tree of every scene is built on the\n"
        "                    device, mpt_build_and_upload, with leaves of <= 6 primitives below 8192 primitives and <= 2 from there\n"
        "                    on; gpu = the same); binned = the host's 16-bin SAH builder\n"
        "  --checkpoint F    batch mode: after the render, write the HDR sum and the number of samples it holds to F\n"
        "  --resume F        batch mode: start from the checkpoint F (same scene, size, seed and RNG) and add --spp MORE samples, numbered\n"
        "                    from where it stopped: the result is bit-identical to one uninterrupted render of all the samples\n"
        "  --frames N        run the reference's frame protocol (N draw() calls, running mean) instead of batch spp\n"
        "  --gpus N          batch mode on GPUs device .. device+N-1: 8x8 pixel tiles interleaved over the GPUs, one RCCL\n"
        "                    reduce(sum) of the HDR framebuffer onto the first (mpt_comm_create_all / mpt_reduce_sum)\n"
        "  --devices a,b,..  the same on the listed device ordinals, one rank each (RCCL itself refuses an ordinal listed twice)\n"
        "  --camera-path F   headless replay of the reference's input handling (R/Window/ControllerView.mm:41-73): one\n"
        "                    line of F per frame, optionally prefixed by a repeat count, holding the keys\n"
        "                    w a s d space c (move), r (reset), `mouse dx dy`, `scroll dy`; every frame is one draw()\n"
        "                    and is written to <out-dir>/frame_NNNN.ppm (default out-dir: runs, as R/runs/)");
}

// One frame of input in the reference's vocabulary (R/Window/ControllerView.mm:41-73): held keys set the movement
// vector to +-1 per axis (keyDown), `r` requests a reset, a mouse drag gives the rotation deltas, the scroll wheel the
// zoom (negated, :70-72).  Camera::transformWithInputs() consumes and clears them inside the next draw().
static bool applyInputLine(const std::string& line, int* repeat) {
    std::vector<std::string> tok;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && std::isspace((unsigned char)line[i])) ++i;
        size_t j = i;
        while (j < line.size() && !std::isspace((unsigned char)line[j])) ++j;
        if (j > i) tok.push_back(line.substr(i, j - i));
        i = j;
    }
    *repeat = 1;
    size_t k = 0;
    if (!tok.empty() && std::isdigit((unsigned char)tok[0][0]) && tok[0].find_first_not_of("0123456789") == std::string::npos) {
        *repeat = std::atoi(tok[0].c_str());
        k = 1;
    }
    InputSystem::clearInputs();
    for (; k < tok.size(); ++k) {
        const std::string& t = tok[k];
        if (t[0] == '#') break;
        if (t == "d") InputSystem::movementInput.x = 1.0f;        // keyCode 2
        else if (t == "a") InputSystem::movementInput.x = -1.0f;  // keyCode 0
        else if (t == "space") InputSystem::movementInput.y = 1.0f;   // keyCode 49
        else if (t == "c") InputSystem::movementInput.y = -1.0f;      // keyCode 8
        else if (t == "w") InputSystem::movementInput.z = 1.0f;   // keyCode 13
        else if (t == "s") InputSystem::movementInput.z = -1.0f;  // keyCode 1
        else if (t == "r") InputSystem::resetInput = true;        // keyCode 15
        else if (t == "mouse" && k + 2 < tok.size()) {
            InputSystem::rotationInput.x = (float)std::atof(tok[k + 1].c_str());
            InputSystem::rotationInput.y = (float)std::atof(tok[k + 2].c_str());
            k += 2;
        } else if (t == "scroll" && k + 1 < tok.size()) {
            InputSystem::zoomInput = -(float)std::atof(tok[k + 1].c_str());
            k += 1;
        } else {
            std::fprintf(stderr, "camera path: unknown token '%s'\n", t.c_str());
            return false;
        }
    }
    return true;
}

// Replays a camera path: one draw() per frame with that frame's inputs, every frame written to outDir.  Returns the
// number of frames, -1 on error.  Prints one JSON line per frame (camera, frameCount) for checking against the reference's
// protocol (a camera change resets the accumulation and reseeds, R/Renderer/Renderer.cpp:255-257).
static int playCameraPath(Renderer& r, OffscreenView& view, const std::string& path, const std::string& outDir) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) {
        std::fprintf(stderr, "cannot open camera path %s\n", path.c_str());
        return -1;
    }
    std::error_code ec;
    std::filesystem::create_directories(outDir, ec);
    if (ec) std::fprintf(stderr, "cannot create %s: %s\n", outDir.c_str(), ec.message().c_str());
    char buf[512];
    int frame = 0;
    while (std::fgets(buf, sizeof buf, f)) {
        std::string line(buf);
        size_t h = line.find('#');
        if (h != std::string::npos) line.erase(h);
        if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
        int repeat = 1;
        for (int k = 0, n = 1; k < n; ++k) {
            if (!applyInputLine(line, &repeat)) {
                std::fclose(f);
                return -1;
            }
            n = repeat;
            r.draw(&view);
            r.readFrame(&view);
            const mpt_uniforms& u = r.uniforms();
            char name[64];
            std::snprintf(name, sizeof name, "/frame_%04d.ppm", frame);
            if (mpt_write_ppm((outDir + name).c_str(), view.rgba.data(), (int)view.width, (int)view.height, 1.0f, 2.2f))
                std::fprintf(stderr, "cannot write %s%s\n", outDir.c_str(), name);
            std::printf("{\"frame\": %d, \"frameCount\": %llu, \"camera\": [%.9g, %.9g, %.9g], \"forward\": [%.9g, %.9g, %.9g], \"vfov\": %.9g}\n",
                        frame, (unsigned long long)u.frameCount, u.cameraPosition[0], u.cameraPosition[1], u.cameraPosition[2],
                        Camera::forward.x, Camera::forward.y, Camera::forward.z, Camera::verticalFov);
            ++frame;
        }
    }
    std::fclose(f);
    return frame;
}

// Batch render on N GPUs driven by this one host thread (SURVEY.md 8e / include/mpt.h "multi-GPU"): every GPU gets the
// scene, renders its interleaved tile shard asynchronously, and ONE ncclReduce(sum) lands the HDR sum on the first GPU.
static int renderOnSeveralGpus(const std::string& scene, const std::string& assetRoot, const std::string& out, int width, int height,
                               int spp, const std::vector<int>& devices, int bvh, mpt_render_params prm, const float* camPos, const float* camDir,
                               const float* camUp, float vfov) {
    std::vector<std::unique_ptr<Renderer>> rs;
    mpt_comm* comm = nullptr;
    const int gpus = static_cast<int>(devices.size());
    try {
        for (int g = 0; g < gpus; ++g) rs.emplace_back(new Renderer(devices[g], scene, assetRoot, bvh));
        if (camPos) Camera::position = mpt::float3(camPos[0], camPos[1], camPos[2]);
        if (camDir) Camera::forward = mpt::normalize(mpt::float3(camDir[0], camDir[1], camDir[2]));
        if (camUp) Camera::up = mpt::normalize(mpt::float3(camUp[0], camUp[1], camUp[2]));
        if (vfov > 0.0f) Camera::verticalFov = vfov;
        std::vector<mpt_ctx*> ctxs;
        OffscreenView view;
        for (int g = 0; g < gpus; ++g) {
            rs[g]->drawableSizeWillChange(&view, DrawableSize{(double)width, (double)height});
            rs[g]->clearSum();
            ctxs.push_back(rs[g]->context());
        }
        int rc = mpt_comm_create_all(ctxs.data(), gpus, &comm);
        if (rc) throw std::runtime_error(std::string("mpt_comm_create_all: ") + mpt_last_error(ctxs[0]));
        auto t0 = std::chrono::steady_clock::now();
        for (int g = 0; g < gpus; ++g) {  // enqueue everywhere first: mpt_render_async returns at once
            mpt_render_params p = prm;
            p.sample_begin = 0;
            p.sample_count = (uint32_t)spp;
            p.shard_rank = g;
            p.shard_count = gpus;
            mpt_uniforms u = rs[g]->uniforms();
            u.primitiveCount = rs[g]->scene()->getPrimitiveCount();
            u.triangleCount = rs[g]->scene()->getTriangleCount();
            if ((rc = mpt_set_uniforms(ctxs[g], &u)) || (rc = mpt_render_async(ctxs[g], &p)))
                throw std::runtime_error(std::string("render on GPU ") + std::to_string(devices[g]) + ": " + mpt_last_error(ctxs[g]));
        }
        if ((rc = mpt_reduce_sum(comm, 0))) throw std::runtime_error(std::string("mpt_reduce_sum: ") + mpt_comm_last_error(comm));
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long rays = 0, paths = 0;
        for (int g = 0; g < gpus; ++g) {
            mpt_stats st = rs[g]->stats();
            rays += st.rays;
            paths += st.paths;
        }
        std::printf("{\"gpus\": %d, \"paths\": %llu, \"rays\": %llu, \"seconds\": %.6f, \"mrays_per_s\": %.1f}\n", gpus, paths, rays, sec,
                    sec > 0 ? rays / sec / 1e6 : 0.0);
        if (!out.empty()) {
            std::vector<float> img;
            rs[0]->readSum(img);
            bool ppm = out.size() > 4 && out.substr(out.size() - 4) == ".ppm";
            const float scale = 1.0f / (float)spp;
            if (ppm ? mpt_write_ppm(out.c_str(), img.data(), width, height, scale, 2.2f) : mpt_write_pfm(out.c_str(), img.data(), width, height, scale))
                throw std::runtime_error("cannot write " + out);
        }
        mpt_comm_destroy(comm);
        return 0;
    } catch (const std::exception& e) {
        if (comm) mpt_comm_destroy(comm);
        std::fprintf(stderr, "mpt_render: %s\n", e.what());
        return 1;
    }
}

int main(int argc, char** argv) {
    std::string scene, assetRoot, out, cameraPath, outDir = "runs", checkpoint, resume;
    int width = 1280, height = 720, spp = 64, depth = 32, device = 0, frames = 0, gpus = 1;
    int bvh = -1;   // -1 = by mode: the reference's builder for the frame protocol / the literal RNG, auto for batch renders
    unsigned seed = 1;
    float camPos[3], camDir[3], camUp[3], vfov = 0.0f;
    bool havePos = false, haveDir = false, haveUp = false;
    std::vector<int> deviceList;   // --devices a,b,...: the ordinals of a multi-GPU render, one rank each (default: --device .. --device + gpus - 1)
    mpt_render_params prm;
    std::memset(&prm, 0, sizeof prm);
    prm.rng_mode = MPT_RNG_PHILOX;
    prm.shard_count = 1;
    prm.pipeline = MPT_PIPE_AUTO;
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
        else if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--devices") {
            for (const char* q = next(); *q;) {
                char* end = nullptr;
                const long v = std::strtol(q, &end, 10);
                if (end == q || v < 0) {
                    usage();
                    return 2;
                }
                deviceList.push_back(static_cast<int>(v));
                q = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "--camera-path") cameraPath = next();
        else if (a == "--out-dir") outDir = next();
        else if (a == "--out") out = next();
        else if (a == "--checkpoint") checkpoint = next();
        else if (a == "--resume") resume = next();
        else if (a == "--bvh") {
            const char* v = next();
            bvh = std::strcmp(v, "reference") == 0 ? Renderer::BUILD_REFERENCE
                  : std::strcmp(v, "binned") == 0  ? Renderer::BUILD_BINNED
                  : std::strcmp(v, "gpu") == 0     ? Renderer::BUILD_GPU : Renderer::BUILD_AUTO;
        }
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
        else if (a == "--bsdf") {
            const char* v = next();
            prm.bsdf_mode = std::strcmp(v, "scatter") == 0 ? MPT_BSDF_SCATTER : std::strcmp(v, "scatter-all") == 0 ? MPT_BSDF_SCATTER_ALL : MPT_BSDF_LAMBERT;
        }
        else if (a == "--pipeline") {
            const char* v = next();
            prm.pipeline = std::strcmp(v, "megakernel") == 0 ? MPT_PIPE_MEGAKERNEL
                           : std::strcmp(v, "wavefront") == 0 ? MPT_PIPE_WAVEFRONT
                           : std::strcmp(v, "wavelocal") == 0 ? MPT_PIPE_WAVELOCAL
                           : std::strcmp(v, "ordered") == 0 ? MPT_PIPE_ORDERED : MPT_PIPE_AUTO;
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
    if (bvh < 0) bvh = prm.rng_mode == MPT_RNG_LITERAL || frames > 0 || !cameraPath.empty() ? Renderer::BUILD_REFERENCE : Renderer::BUILD_AUTO;
    if (!deviceList.empty()) gpus = static_cast<int>(deviceList.size());
    if (gpus > 1) {
        if (deviceList.empty())
            for (int g = 0; g < gpus; ++g) deviceList.push_back(device + g);
        return renderOnSeveralGpus(scene, assetRoot, out, width, height, spp, deviceList, bvh, prm, havePos ? camPos : nullptr, haveDir ? camDir : nullptr,
                                   haveUp ? camUp : nullptr, vfov);
    }
    if (deviceList.size() == 1) device = deviceList[0];
    try {
        Renderer r(device, scene, assetRoot, bvh);
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
        if (!cameraPath.empty()) {
            const int n = playCameraPath(r, view, cameraPath, outDir);
            if (n < 0) return 1;
            frames = n;
            r.readFrame(&view);
            img = view.rgba;
        } else if (frames > 0) {
            for (int f = 0; f < frames; ++f) r.draw(&view);
            r.readFrame(&view);
            img = view.rgba;
        } else {
            // checkpoint / resume of the accumulation (the reference's running mean lives in a GPU-private texture and is lost with the
            // process, R/Renderer/Renderer.cpp:236): header "MPTSUM2 W H samples seed rng depth bsdf scene-hash\n" + W * H * 4 raw floats.
            // The hash (FNV-1a over the packed primitive and material arrays) and the BSDF mode identify WHAT was accumulated: a resume on
            // another scene or material model is refused instead of mixing sums.  Written to a temporary file and renamed over the target,
            // so that a crash mid-write never destroys the checkpoint that --resume just read.
            uint32_t have = 0;
            r.clearSum();
            const unsigned long long sceneHash = sceneFingerprint(*r.scene());
            if (!resume.empty()) {
                FILE* f = std::fopen(resume.c_str(), "rb");
                int w = 0, h = 0, rngm = 0, dep = 0, bs = 0;
                unsigned sd = 0;
                unsigned long long sh = 0;
                if (!f || std::fscanf(f, "MPTSUM2 %d %d %u %u %d %d %d %llx", &w, &h, &have, &sd, &rngm, &dep, &bs, &sh) != 8 || std::fgetc(f) != '\n') {
                    if (f) std::fclose(f);
                    throw std::runtime_error("cannot read the checkpoint " + resume);
                }
                if (w != width || h != height || sd != seed || rngm != prm.rng_mode || dep != depth || bs != prm.bsdf_mode || sh != sceneHash) {
                    std::fclose(f);
                    throw std::runtime_error("the checkpoint " + resume + " was written with another scene, size, seed, RNG, BSDF mode or depth");
                }
                std::vector<float> sum(static_cast<size_t>(w) * h * 4);
                const size_t got = std::fread(sum.data(), sizeof(float), sum.size(), f);
                std::fclose(f);
                if (got != sum.size()) throw std::runtime_error("the checkpoint " + resume + " is truncated");
                r.writeSum(sum);
            }
            r.renderBatch(have, static_cast<uint32_t>(spp));
            r.readSum(img);
            if (!checkpoint.empty()) {
                const std::string tmp = checkpoint + ".tmp";
                FILE* f = std::fopen(tmp.c_str(), "wb");
                if (!f) throw std::runtime_error("cannot write " + tmp);
                std::fprintf(f, "MPTSUM2 %d %d %u %u %d %d %d %llx\n", width, height, have + static_cast<uint32_t>(spp), seed, prm.rng_mode, depth, prm.bsdf_mode,
                             sceneHash);
                const bool ok = std::fwrite(img.data(), sizeof(float), img.size(), f) == img.size();
                if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), checkpoint.c_str()) != 0) {
                    std::remove(tmp.c_str());
                    throw std::runtime_error("cannot write " + checkpoint);
                }
            }
            scale = 1.0f / static_cast<float>(have + static_cast<uint32_t>(spp));
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

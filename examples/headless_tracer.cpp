// headless_tracer - the reference's start-up (cudaTracer/main.cpp:66-138) without window and message loop:
//
//   headless_tracer                          the default scene (RayTracer::initScene, RayTracer.cpp:50-78) with the
//                                            start-up options of loadDefaults(), one frame, dumped as PNG
//   headless_tracer <experiment.json> <i>    phase i of an experiment file, as `cudaTracer.exe <file> <i>` runs it:
//                                            the phase's options, experimentRounds = 1 frame with dt = 0.016, PNG dump
//   options:  --frames N   --out file.png   --assets DIR (default: procedural textures, cubes for the barrels)
//             --gpus N (devices 0..N-1 of the node render every frame together; --gpus 0,0 names devices explicitly)
//
// Uses only the host classes of cudaraytracer_amd/csrc/host (RayTracer, DeviceMesh, BVH, ExperimentReader, PicDumper)
// and, through them, the C-ABI of include/rt_capi.h.  Build (from the repository root):
//
//   hipcc -x c++ -std=c++17 -O2 examples/headless_tracer.cpp -Lcudaraytracer_amd -lrt_gfx950 -lz \
//         -Wl,-rpath,'$ORIGIN/../cudaraytracer_amd' -o examples/headless_tracer
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>

#include "../cudaraytracer_amd/csrc/host/CubeModelFactory.h"
#include "../cudaraytracer_amd/csrc/host/ExperimentReader.h"
#include "../cudaraytracer_amd/csrc/host/ObjFileReader.h"
#include "../cudaraytracer_amd/csrc/host/RayTracer.h"

using namespace rt;

namespace {

TextureSource checker(unsigned n, unsigned char lo, unsigned char hi) {
    std::vector<unsigned char> px((size_t)n * n * 4);
    for (unsigned y = 0; y < n; ++y)
        for (unsigned x = 0; x < n; ++x) {
            const unsigned char c = ((x / 8 + y / 8) & 1) ? hi : lo;
            unsigned char* p = &px[((size_t)y * n + x) * 4];
            p[0] = c;
            p[1] = c;
            p[2] = (unsigned char)(c / 2 + 60);
            p[3] = 255;
        }
    return TextureSource(n, n, std::move(px));
}

// RayTracer::initScene for `opts`: a named model alone, or spheres + plane + three mesh instances; then the lights.
void initScene(RayTracer& tracer, const Options& opts, const std::string& assets) {
    Material mtl;
    mtl.materialType = Material::DIFFUSE_AND_GLOSSY;
    if (!opts.host.model.fname.empty()) {
        const std::vector<model::Model> models = ObjFileReader().readFile(opts.host.model.folder, opts.host.model.fname, false);
        if (models.empty()) throw std::runtime_error("no model in " + opts.host.model.folder + opts.host.model.fname);
        tracer.addMesh(std::make_unique<DeviceMesh>(models[0], mtl, opts.host.model.transform));
    } else {
        tracer.addSpheres();
        const float pos[3][3] = {{0.5f, -3, 9}, {-2, -3, 5}, {-4, -3, 4}};
        if (!assets.empty()) {
            const std::string t = assets + "/textures/pattern_106/";
            tracer.addPlane(t + "diffus.png", t + "normal.png", t + "specular.png");
            const std::vector<model::Model> models = ObjFileReader().readFile(assets + "/models/metalBarrel/", "metal_barrel.obj", false);
            if (models.empty()) throw std::runtime_error("metal_barrel.obj: no model read");
            const Mat44f scale = Mat44f::scale(1 / 50.0f, 1 / 50.0f, 1 / 50.0f);
            for (const auto& p : pos) tracer.addMesh(std::make_unique<DeviceMesh>(models[0], mtl, scale * Mat44f::translate(p[0], p[1], p[2])));
        } else {
            tracer.addPlane(checker(64, 70, 200), TextureSource::solid(128, 128, 255), TextureSource::solid(160, 160, 160));
            for (const auto& p : pos)
                tracer.addMesh(std::make_unique<DeviceMesh>(CubeModelFactory::createVertices(), CubeModelFactory::createIndicies(), 12,
                                                            checker(16, 90, 220), TextureSource::solid(128, 128, 255),
                                                            TextureSource::solid(255, 255, 255), Texture::CLAMP, mtl,
                                                            Mat44f::scale(0.6f, 1.0f, 0.6f) * Mat44f::translate(p[0], p[1] + 1.0f, p[2])));
        }
    }
    tracer.addLights();
    tracer.finalizeScene();
}

}  // namespace

int main(int argc, char** argv) {
    std::string experimentFname, out = "frame.png", assets;
    int phaseIdx = 0, frames = 1, positional = 0;
    bool skipUnlit = false;  // shadow rays towards lights behind the surface are not walked (same image)
    bool rebuildSah = false;  // every mesh BVH is built again on the GPU at upload, cut by the binned surface-area heuristic
    std::vector<int> devices{0};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--frames" && i + 1 < argc)
            frames = std::atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc)
            out = argv[++i];
        else if (a == "--assets" && i + 1 < argc)
            assets = argv[++i];
        else if (a == "--skip-unlit")
            skipUnlit = true;
        else if (a == "--rebuild-sah")
            rebuildSah = true;
        else if (a == "--gpus" && i + 1 < argc) {
            // a count ("4": devices 0..3) or a list ("0,0": the same GPU twice, two contexts)
            const std::string v = argv[++i];
            devices.clear();
            if (v.find(',') == std::string::npos) {
                for (int d = 0; d < std::atoi(v.c_str()); ++d) devices.push_back(d);
            } else {
                size_t at = 0;
                while (at <= v.size()) {
                    const size_t comma = v.find(',', at);
                    devices.push_back(std::atoi(v.substr(at, comma - at).c_str()));
                    if (comma == std::string::npos) break;
                    at = comma + 1;
                }
            }
            if (devices.empty()) devices.push_back(0);
        } else if (positional == 0) {
            experimentFname = a;
            positional++;
        } else if (positional == 1) {
            phaseIdx = std::atoi(a.c_str());
            positional++;
        } else {
            std::fprintf(stderr, "usage: %s [experiment.json [phase]] [--frames N] [--out file.png] [--assets DIR] [--gpus N | --gpus d0,d1,...] [--skip-unlit] [--rebuild-sah]\n", argv[0]);
            return 2;
        }
    }
    try {
        Options options = loadDefaults();
        std::string label = "default scene";
        if (!experimentFname.empty()) {
            ExperimentReader reader(experimentFname, options);
            std::cout << reader;
            for (const std::string& e : reader.errors) std::cerr << e << std::endl;
            if (phaseIdx < 0 || phaseIdx >= reader.experiment.phaseCnt) {
                std::fprintf(stderr, "phase %d does not exist (%d phase(s) read)\n", phaseIdx, reader.experiment.phaseCnt);
                return 1;
            }
            const ExperimentPhase& phase = reader.experiment.phases[phaseIdx];
            std::cout << "running experiment: " << reader.experiment.experimentName << std::endl << "\tPhase: " << phase.phaseName << std::endl;
            options = phase.opts;
            label = reader.experiment.experimentName + " / " + phase.phaseName;
        }
        RayTracer tracer(options);
        tracer.skipUnlitShadowRays = skipUnlit;
        if (rebuildSah) {
            tracer.rebuildAtUpload = true;
            tracer.deviceSplitRule = BVH::BINNED_SAH;
        }
        initScene(tracer, options, assets);

        rt_ctx* ctx = nullptr;
        rt_multi* node = nullptr;
        if (devices.size() > 1) {
            node = rt_multi_create((int)devices.size(), devices.data());
            if (!node) {
                std::fprintf(stderr, "rt_multi_create: %s\n", rt_multi_last_error(nullptr));
                return 1;
            }
        } else {
            ctx = rt_create(devices[0]);
            if (!ctx) {
                std::fprintf(stderr, "rt_create: %s\n", rt_last_error(nullptr));
                return 1;
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (node)
            tracer.upload(node);
        else
            tracer.upload(ctx);
        for (int i = 0; i < frames; ++i) tracer.update(0.016f);  // main.cpp:127
        const bool dumped = tracer.dumpFrame(out);               // main.cpp:128-129
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%s: %dx%d, %d light(s), maxDepth %d, %d frame(s) on %zu GPU context(s) + upload + dump in %.2f ms -> %s\n", label.c_str(),
                    options.device.width, options.device.height, options.host.lightCnt, options.device.maxDepth, frames, devices.size(), ms,
                    dumped ? out.c_str() : "(dump failed)");
        tracer.release();
        if (node) rt_multi_destroy(node);
        if (ctx) rt_destroy(ctx);
        return dumped ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}

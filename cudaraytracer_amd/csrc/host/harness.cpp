// harness.cpp - C entry points (rth_*) that let the Python bench / tests drive
// the C++ host API: build one of the benchmark scenes with RayTracer /
// DeviceMesh / BVH::BvhFactory, expose its flat RtSceneDesc + options + camera,
// rotate lights, dump PNGs, and run the host-API known-answer tests.
// Nothing here touches the GPU; rendering goes through include/rt_capi.h.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "CubeModelFactory.h"
#include "ExperimentReader.h"
#include "ObjFileReader.h"
#include "RayTracer.h"

using namespace rt;

namespace {

thread_local std::string g_error;

// PCG32 (XSH-RR 64/32, O'Neill 2014): the generator every procedural scene uses.
struct Pcg32 {
    uint64_t state = 0, inc = 0;
    explicit Pcg32(uint64_t seed, uint64_t seq = 54u) {
        inc = (seq << 1u) | 1u;
        next();
        state += seed;
        next();
    }
    uint32_t next() {
        const uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    float uniform() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }  // [0, 1)
    float uniform(float lo, float hi) { return lo + (hi - lo) * uniform(); }
    // three draws, x first: written out because the order in which a constructor's arguments are evaluated is
    // the compiler's choice, and the scenes must not depend on it
    Vec3f uniform3(float lo, float hi) {
        const float x = uniform(lo, hi);
        const float y = uniform(lo, hi);
        const float z = uniform(lo, hi);
        return Vec3f(x, y, z);
    }
    Vec3f uniform3(float xlo, float xhi, float ylo, float yhi, float zlo, float zhi) {
        const float x = uniform(xlo, xhi);
        const float y = uniform(ylo, yhi);
        const float z = uniform(zlo, zhi);
        return Vec3f(x, y, z);
    }
};

TextureSource checker(unsigned n, const unsigned char a[3], const unsigned char b[3]) {
    std::vector<unsigned char> px((size_t)n * n * 4);
    for (unsigned y = 0; y < n; ++y)
        for (unsigned x = 0; x < n; ++x) {
            const unsigned char* c = ((x ^ y) & 1) ? a : b;
            unsigned char* p = &px[((size_t)y * n + x) * 4];
            p[0] = c[0];
            p[1] = c[1];
            p[2] = c[2];
            p[3] = 255;
        }
    return TextureSource(n, n, std::move(px));
}
// smooth bumps for a procedural normal map: RGB = (n * 0.5 + 0.5) * 255
TextureSource bumpNormals(unsigned n) {
    std::vector<unsigned char> px((size_t)n * n * 4);
    for (unsigned y = 0; y < n; ++y)
        for (unsigned x = 0; x < n; ++x) {
            const float fx = 0.35f * std::sin(6.2831853f * x / n * 2.0f), fy = 0.35f * std::cos(6.2831853f * y / n * 2.0f);
            unsigned char* p = &px[((size_t)y * n + x) * 4];
            p[0] = (unsigned char)((fx * 0.5f + 0.5f) * 255.0f);
            p[1] = (unsigned char)((fy * 0.5f + 0.5f) * 255.0f);
            p[2] = 255;
            p[3] = 255;
        }
    return TextureSource(n, n, std::move(px));
}
TextureSource greyRamp(unsigned n) {
    std::vector<unsigned char> px((size_t)n * n * 4);
    for (unsigned y = 0; y < n; ++y)
        for (unsigned x = 0; x < n; ++x) {
            unsigned char* p = &px[((size_t)y * n + x) * 4];
            p[0] = p[1] = p[2] = (unsigned char)(40 + (x * 7 + y * 13) % 200);
            p[3] = 255;
        }
    return TextureSource(n, n, std::move(px));
}

void ringLights(RayTracer& rt, int count, uint64_t seed) {
    // deterministic stand-in for the reference's rand()-driven ring of lights
    // (cudaTracer/RayTracer.cpp:85-101): radius U(10,50), height U(0,25),
    // intensity U(0,.5)^3, power U(0,25)
    Pcg32 rng(seed);
    const float radStep = (2.0f * PI) / count;
    for (int i = 0; i < count; ++i) {
        const float power = rng.uniform(0, 25.0f);
        const float radiusX = rng.uniform(10, 50);
        const float height = rng.uniform(0, 25.0f);
        const float radiusZ = rng.uniform(10, 50);
        const Vec3f position(std::sin(i * radStep) * radiusX, height, std::cos(i * radStep) * radiusZ);
        const Vec3f intensity = rng.uniform3(0, 0.5f);
        rt.addLight(Light(position, intensity, power));
    }
}

using BVH::BoundsMode;
using BVH::BuildOptions;

Options baseOptions(int width, int height, float fov, int maxDepth) {
    Options o = loadDefaults();
    setView(o, width, height, fov);
    o.device.maxDepth = maxDepth;
    return o;
}

// cfg "cubes": n x n grid of unit cubes (CubeModelFactory tables), each its own
// mesh with its own BVH, under a scene BVH; 1x1 textures; 4 fixed lights.
std::unique_ptr<RayTracer> buildCubes(int width, int height, int n, int lightCnt, int maxDepth, BuildOptions bm) {
    if (n < 1) n = 16;
    auto rt = std::make_unique<RayTracer>(baseOptions(width, height, 90, maxDepth), bm.bounds);
    const std::vector<Vertex> verts = CubeModelFactory::createVertices();
    const std::vector<int> inds = CubeModelFactory::createIndicies();
    Material mtl;
    mtl.materialType = Material::DIFFUSE_AND_GLOSSY;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const Mat44f transform =
                Mat44f::scale(0.4f, 0.4f, 0.4f) * Mat44f::translate(1.2f * i - 0.6f * n, -2.0f, 6.0f + 1.2f * j);
            rt->addMesh(std::make_unique<DeviceMesh>(verts, inds, 12, TextureSource::solid(255, 255, 255),
                                                     TextureSource::solid(128, 128, 255), TextureSource::solid(255, 255, 255),
                                                     Texture::CLAMP, mtl, transform, bm));
        }
    static const Light table[4] = {Light(Vec3f(0, 8, 0), Vec3f(0.5f, 0.5f, 0.5f), 20), Light(Vec3f(-12, 6, 10), Vec3f(0.5f, 0.3f, 0.2f), 25),
                                   Light(Vec3f(12, 10, 20), Vec3f(0.2f, 0.3f, 0.5f), 25), Light(Vec3f(0, 3, -4), Vec3f(0.3f, 0.3f, 0.3f), 10)};
    if (lightCnt < 1 || lightCnt > 4) lightCnt = 4;
    for (int i = 0; i < lightCnt; ++i) rt->addLight(table[i]);
    rt->finalizeScene();
    return rt;
}

// cfg "soup": N random triangles in one mesh (de-indexed, like the OBJ reader
// produces): centres U[-50,50]^3, two edges U[-0.75,0.75]^3, face normals,
// UVs (0,0)(1,0)(0,1); PCG32 seed 1234; 16x16 procedural textures.
std::unique_ptr<RayTracer> buildSoup(int width, int height, int triCnt, int lightCnt, int maxDepth, BuildOptions bm) {
    if (triCnt < 1) triCnt = 1000000;
    auto rt = std::make_unique<RayTracer>(baseOptions(width, height, 60, maxDepth), bm.bounds);
    Pcg32 rng(1234);
    std::vector<Vertex> verts;
    std::vector<int> inds;
    verts.reserve((size_t)triCnt * 3);
    inds.reserve((size_t)triCnt * 3);
    for (int t = 0; t < triCnt; ++t) {
        const Vec3f c = rng.uniform3(-50, 50);
        const Vec3f e1 = rng.uniform3(-0.75f, 0.75f);
        const Vec3f e2 = rng.uniform3(-0.75f, 0.75f);
        const Vec3f n = e1.cross(e2).normalized();
        verts.push_back(Vertex(c, n, Vec2f(0.0f, 0.0f)));
        verts.push_back(Vertex(c + e1, n, Vec2f(1.0f, 0.0f)));
        verts.push_back(Vertex(c + e2, n, Vec2f(0.0f, 1.0f)));
        for (int k = 0; k < 3; ++k) inds.push_back(3 * t + k);
    }
    const unsigned char ca[3] = {230, 200, 90}, cb[3] = {60, 110, 200};
    Material mtl;
    mtl.materialType = Material::DIFFUSE_AND_GLOSSY;
    rt->addMesh(std::make_unique<DeviceMesh>(std::move(verts), std::move(inds), triCnt, checker(16, ca, cb), bumpNormals(16), greyRamp(16),
                                             Texture::WRAP, mtl, Mat44f::identity(), bm));
    rt->camera.setPosition(0, 0, -120);
    rt->camera.update();
    if (lightCnt <= 1) {
        rt->addLight(Light(Vec3f(0, 200, -120), Vec3f(0.5f, 0.5f, 0.5f), 25));
    } else {
        Pcg32 lr(77);
        for (int i = 0; i < lightCnt; ++i) {
            const Vec3f position = lr.uniform3(-150, 150, 100, 250, -200, -80);
            const Vec3f intensity = lr.uniform3(0.1f, 0.5f);
            const float power = lr.uniform(10, 40);
            rt->addLight(Light(position, intensity, power));
        }
    }
    rt->finalizeScene();
    return rt;
}

// cfg "heightfield": side x side quads of a sinusoidally displaced, jittered
// height field in ONE mesh (2 triangles per quad, de-indexed).
std::unique_ptr<RayTracer> buildHeightfield(int width, int height, int triCnt, int lightCnt, int maxDepth, BuildOptions bm) {
    int side = (int)std::floor(std::sqrt((triCnt < 2 ? 9999392 : triCnt) / 2.0));
    if (side < 1) side = 1;
    auto rt = std::make_unique<RayTracer>(baseOptions(width, height, 60, maxDepth), bm.bounds);
    Pcg32 rng(1234);
    const float extent = 200.0f, cell = extent / side;
    std::vector<float> hgt((size_t)(side + 1) * (side + 1));
    for (int z = 0; z <= side; ++z)
        for (int x = 0; x <= side; ++x) {
            const float fx = x * cell, fz = z * cell;
            hgt[(size_t)z * (side + 1) + x] =
                4.0f * std::sin(fx * 0.11f) * std::cos(fz * 0.07f) + 1.5f * std::sin(fx * 0.53f + fz * 0.31f) + rng.uniform(-0.05f, 0.05f);
        }
    auto P = [&](int x, int z) { return Vec3f(x * cell - extent * 0.5f, hgt[(size_t)z * (side + 1) + x], z * cell); };
    std::vector<Vertex> verts;
    std::vector<int> inds;
    verts.reserve((size_t)side * side * 6);
    inds.reserve((size_t)side * side * 6);
    auto emit = [&](Vec3f a, Vec3f b, Vec3f c, Vec2f ta, Vec2f tb, Vec2f tc) {
        const Vec3f n = (b - a).cross(c - a).normalized();
        const int base = (int)verts.size();
        verts.push_back(Vertex(a, n, ta));
        verts.push_back(Vertex(b, n, tb));
        verts.push_back(Vertex(c, n, tc));
        for (int k = 0; k < 3; ++k) inds.push_back(base + k);
    };
    for (int z = 0; z < side; ++z)
        for (int x = 0; x < side; ++x) {
            // counter-clockwise seen from above, so the upward face is the front face
            emit(P(x, z), P(x, z + 1), P(x + 1, z), Vec2f(0, 0), Vec2f(0, 1), Vec2f(1, 0));
            emit(P(x + 1, z), P(x, z + 1), P(x + 1, z + 1), Vec2f(1, 0), Vec2f(0, 1), Vec2f(1, 1));
        }
    const unsigned char ca[3] = {120, 170, 80}, cb[3] = {90, 120, 60};
    Material mtl;
    const int made = (int)inds.size() / 3;
    rt->addMesh(std::make_unique<DeviceMesh>(std::move(verts), std::move(inds), made, checker(16, ca, cb), bumpNormals(16), greyRamp(16),
                                             Texture::WRAP, mtl, Mat44f::identity(), bm));
    rt->camera.setPosition(0, 45, -30);
    rt->camera.setDirection(Vec3f(0, -0.45f, 1.0f));
    if (lightCnt < 1) lightCnt = 1;
    rt->addLight(Light(Vec3f(30, 120, 60), Vec3f(0.5f, 0.5f, 0.45f), 120));
    Pcg32 lr(78);
    for (int i = 1; i < lightCnt; ++i) {
        const Vec3f position = lr.uniform3(-90, 90, 60, 140, 10, 190);
        const Vec3f intensity = lr.uniform3(0.1f, 0.4f);
        const float power = lr.uniform(40, 120);
        rt->addLight(Light(position, intensity, power));
    }
    rt->finalizeScene();
    return rt;
}

// The reference's default scene content (cudaTracer/RayTracer.cpp:67-69): sphere field + mirror + glass,
// textured plane, three mesh instances.  With an asset directory the plane uses pattern_106 and the meshes
// are the metal barrel (the plastic one ships without its .obj); without one, procedural textures and
// cubes stand in.
void addDefaultContent(RayTracer& rt, const std::string& assets, BuildOptions bm) {
    rt.addSpheres();
    const unsigned char ca[3] = {200, 200, 200}, cb[3] = {70, 70, 90};
    if (!assets.empty()) {
        const std::string t = assets + "/textures/pattern_106/";
        rt.addPlane(t + "diffus.png", t + "normal.png", t + "specular.png");
    } else {
        rt.addPlane(checker(64, ca, cb), bumpNormals(64), greyRamp(64));
    }
    Material mtl;
    mtl.materialType = Material::DIFFUSE_AND_GLOSSY;
    const float pos[3][3] = {{0.5f, -3, 9}, {-2, -3, 5}, {-4, -3, 4}};
    if (!assets.empty()) {
        // cudaTracer/RayTracer.cpp:158-186
        const Mat44f scale = Mat44f::scale(1 / 50.0f, 1 / 50.0f, 1 / 50.0f);
        const std::vector<model::Model> models = ObjFileReader().readFile(assets + "/models/metalBarrel/", "metal_barrel.obj", false);
        if (models.empty()) throw std::runtime_error("metal_barrel.obj: no model read");
        for (int k = 0; k < 3; ++k)
            rt.addMesh(std::make_unique<DeviceMesh>(models[0], mtl, scale * Mat44f::translate(pos[k][0], pos[k][1], pos[k][2]), bm));
    } else {
        const std::vector<Vertex> verts = CubeModelFactory::createVertices();
        const std::vector<int> inds = CubeModelFactory::createIndicies();
        for (int k = 0; k < 3; ++k)
            rt.addMesh(std::make_unique<DeviceMesh>(verts, inds, 12, checker(8, ca, cb), bumpNormals(8), greyRamp(8), Texture::CLAMP, mtl,
                                                    Mat44f::scale(0.6f, 1.0f, 0.6f) * Mat44f::translate(pos[k][0], pos[k][1] + 1.0f, pos[k][2]), bm));
    }
}

// cfg "mixed": the reference's default scene shape with a deterministic light ring.
std::unique_ptr<RayTracer> buildMixed(int width, int height, int lightCnt, int maxDepth, const std::string& assets, BuildOptions bm) {
    auto rt = std::make_unique<RayTracer>(baseOptions(width, height, 90, maxDepth), bm.bounds);
    addDefaultContent(*rt, assets, bm);
    ringLights(*rt, lightCnt < 1 ? 16 : lightCnt, 4242);
    rt->finalizeScene();
    return rt;
}

// One experiment phase: RayTracer::initScene (cudaTracer/RayTracer.cpp:50-78) for the phase's options.  A named
// model replaces the default content and takes the phase's transform; the lights are the reference's ring.
std::unique_ptr<RayTracer> buildPhase(const Options& opts, const std::string& assets, BuildOptions bm) {
    if (opts.device.width < 1 || opts.device.height < 1) throw std::invalid_argument("phase: width and height must be positive");
    if (opts.host.lightCnt < 0) throw std::invalid_argument("phase: lightCnt must not be negative");
    auto rt = std::make_unique<RayTracer>(opts, bm.bounds);
    if (!opts.host.model.fname.empty()) {
        Material mtl;
        mtl.materialType = Material::DIFFUSE_AND_GLOSSY;
        const std::vector<model::Model> models = ObjFileReader().readFile(opts.host.model.folder, opts.host.model.fname, false);
        if (models.empty()) throw std::runtime_error(opts.host.model.folder + opts.host.model.fname + ": no model read");
        rt->addMesh(std::make_unique<DeviceMesh>(models[0], mtl, opts.host.model.transform, bm));
    } else {
        addDefaultContent(*rt, assets, bm);
    }
    rt->addLights();
    rt->finalizeScene();
    return rt;
}

// cfg "barrel": the OBJ asset with its three PNG maps + textured plane + mirror
// and glass balls + a mirror-material barrel instance.
std::unique_ptr<RayTracer> buildBarrel(int width, int height, int lightCnt, int maxDepth, const std::string& assets, BuildOptions bm) {
    if (assets.empty()) throw std::invalid_argument("scene 'barrel' needs an asset directory");
    auto rt = std::make_unique<RayTracer>(baseOptions(width, height, 90, maxDepth), bm.bounds);
    const Mat44f scale = Mat44f::scale(1 / 50.0f, 1 / 50.0f, 1 / 50.0f);
    const std::vector<model::Model> models = ObjFileReader().readFile(assets + "/models/metalBarrel/", "metal_barrel.obj", false);
    if (models.empty()) throw std::runtime_error("metal_barrel.obj: no model read");
    Material mtl;
    rt->addMesh(std::make_unique<DeviceMesh>(models[0], mtl, scale * Mat44f::translate(0.5f, -3, 5), bm));
    rt->addMesh(std::make_unique<DeviceMesh>(models[0], mtl, scale * Mat44f::translate(-2, -3, 6), bm));
    Material mirror;
    mirror.materialType = Material::REFLECTION;
    mirror.ior = 64;
    rt->addMesh(std::make_unique<DeviceMesh>(models[0], mirror, scale * Mat44f::translate(-4.5f, -3, 7), bm));
    const std::string t = assets + "/textures/pattern_106/";
    rt->addPlane(t + "diffus.png", t + "normal.png", t + "specular.png");
    Shape mirrorBall = Shape(Sphere(Vec3f(2.5f, 1.0f, 6.0f), 1.0f));
    mirrorBall.material.ior = 64;
    mirrorBall.material.materialType = Material::REFLECTION;
    rt->addShape(mirrorBall);
    Shape glassBall = Shape(Sphere(Vec3f(1.5f, -1.5f, 4.5f), 1.5f));
    glassBall.material.ior = 1.5f;
    glassBall.material.materialType = Material::REFLECTION_AND_REFRACTION;
    rt->addShape(glassBall);
    ringLights(*rt, lightCnt < 1 ? 10 : lightCnt, 99);
    rt->finalizeScene();
    return rt;
}

}  // namespace

struct rth_scene {
    std::unique_ptr<RayTracer> rt;
    std::vector<RtShape> shapeStore;
    std::vector<RtMesh> meshStore;
    RtSceneDesc desc;
    RtCamera camera;
    void refresh() {
        Scene s = rt->currentScene();
        desc = s.describe(shapeStore, meshStore);
        camera = s.cameraDesc();
    }
};

// flags of rth_scene_create / rth_experiment_scene: bit 0 tight BVH bounds, bit 1 mesh BVHs built on the device,
// bit 3 mesh BVHs cut by the binned surface-area heuristic (host build), bits 8..15 leaf threshold (0: 10, the
// reference's, for the median cut; 8 for the surface-area cut)
BuildOptions buildOptionsOf(int flags) {
    const BVH::SplitRule rule = (flags & 8) ? BVH::BINNED_SAH : BVH::MEDIAN_CUT;
    int threshold = (flags >> 8) & 0xff;
    if (threshold == 0) threshold = rule == BVH::BINNED_SAH ? 8 : 10;
    return BuildOptions((flags & 1) ? BVH::TIGHT_BOUNDS : BVH::REFERENCE_BOUNDS, (flags & 2) != 0, rule, threshold);
}

extern "C" {

const char* rth_last_error(void) { return g_error.c_str(); }

// name: "cubes" | "soup" | "heightfield" | "mixed" | "barrel".  `size` = grid side
// (cubes) or triangle count (soup, heightfield); 0 = the benchmark default.  flags bit 0: build every BVH
// with tight bounds instead of the reference's FLT_MIN-clamped ones.
rth_scene* rth_scene_create(const char* name, int width, int height, int size, int lightCnt, int maxDepth, const char* assetDir,
                            int flags) {
    try {
        const BuildOptions bm = buildOptionsOf(flags);
        const std::string n = name ? name : "";
        const std::string assets = assetDir ? assetDir : "";
        std::unique_ptr<RayTracer> rt;
        if (n == "cubes")
            rt = buildCubes(width, height, size, lightCnt, maxDepth, bm);
        else if (n == "soup")
            rt = buildSoup(width, height, size, lightCnt, maxDepth, bm);
        else if (n == "heightfield")
            rt = buildHeightfield(width, height, size, lightCnt, maxDepth, bm);
        else if (n == "mixed")
            rt = buildMixed(width, height, lightCnt, maxDepth, assets, bm);
        else if (n == "barrel")
            rt = buildBarrel(width, height, lightCnt, maxDepth, assets, bm);
        else
            throw std::invalid_argument("unknown scene '" + n + "'");
        auto* h = new rth_scene();
        h->rt = std::move(rt);
        h->refresh();
        return h;
    } catch (const std::exception& e) {
        g_error = e.what();
        return nullptr;
    }
}
void rth_scene_free(rth_scene* h) { delete h; }
const RtSceneDesc* rth_scene_desc(rth_scene* h) { return &h->desc; }
RtDeviceOpts* rth_scene_opts(rth_scene* h) { return reinterpret_cast<RtDeviceOpts*>(&h->rt->options.device); }
const RtCamera* rth_scene_camera(rth_scene* h) { return &h->camera; }
int rth_scene_triangles(rth_scene* h) {
    long long n = 0;
    for (auto& m : h->rt->meshes) n += (long long)m->triangles.size();
    return (int)n;
}
int rth_scene_bvh_depth(rth_scene* h, int* sceneDepth, int* meshDepth) {
    int md = 0;
    for (auto& m : h->rt->meshes) md = std::max(md, m->bvhDepth);
    if (sceneDepth) *sceneDepth = h->rt->sceneBvhDepth;
    if (meshDepth) *meshDepth = md;
    return 0;
}
// RayTracer::updateLights (cudaTracer/RayTracer.cpp:259-267); desc->lights stays valid.
void rth_scene_rotate_lights(rth_scene* h, float dt) {
    h->rt->updateLights(dt);
    h->refresh();
}
void rth_scene_set_camera(rth_scene* h, const float pos[3], const float dir[3]) {
    h->rt->camera.setPosition(pos[0], pos[1], pos[2]);
    h->rt->camera.setDirection(Vec3f(dir[0], dir[1], dir[2]));
    h->refresh();
}

// The headless frame loop of the reference's RayTracer (cudaTracer/RayTracer.cpp:188-208) on the C++ driver:
// upload, `frames` times { rotate the lights by dt, set them, render, wait }, then dump the last frame as PNG
// (PicDumper) and, if `rgba` is given, copy its float pixels out.  The scene's light table keeps the rotation.
int rth_scene_run_frames(rth_scene* h, int device, int frames, float dt, const char* pngPath, float* rgba) {
    rt_ctx* ctx = rt_create(device);
    if (!ctx) {
        g_error = rt_last_error(nullptr);
        return -1;
    }
    int rc = 0;
    try {
        h->rt->upload(ctx);
        for (int i = 0; i < frames; ++i) h->rt->update(dt);
        if (rgba) {
            const std::vector<float> px = h->rt->readFrame();
            std::memcpy(rgba, px.data(), px.size() * sizeof(float));
        }
        if (pngPath && !h->rt->dumpFrame(pngPath)) throw std::runtime_error("cannot write the PNG");
    } catch (const std::exception& e) {
        g_error = e.what();
        rc = -1;
    }
    h->rt->release();
    rt_destroy(ctx);
    h->refresh();
    return rc;
}

// ---- experiment files (cudaTracer/ExperimentReader.cpp, cudaTracer/main.cpp:97-135) ----
struct rth_experiment {
    std::unique_ptr<ExperimentReader> reader;
    std::string errors;
};

// Opens an experiment description: `jsonPath` names a file, or - when it is NULL - `jsonText` holds the JSON.
// Every phase starts from loadDefaults().  Never NULL for readable arguments: check rth_experiment_errors.
rth_experiment* rth_experiment_open(const char* jsonPath, const char* jsonText) {
    if (!jsonPath && !jsonText) {
        g_error = "rth_experiment_open: neither a path nor a text";
        return nullptr;
    }
    auto* e = new rth_experiment();
    if (jsonPath)
        e->reader = std::make_unique<ExperimentReader>(jsonPath, loadDefaults());
    else
        e->reader = std::make_unique<ExperimentReader>(ExperimentReader::fromString(jsonText, loadDefaults()));
    for (const std::string& m : e->reader->errors) e->errors += m + "\n\n";
    return e;
}
void rth_experiment_free(rth_experiment* e) { delete e; }
const char* rth_experiment_name(rth_experiment* e) { return e->reader->experiment.experimentName.c_str(); }
int rth_experiment_phase_count(rth_experiment* e) { return e->reader->experiment.phaseCnt; }
int rth_experiment_error_count(rth_experiment* e) { return (int)e->reader->errors.size(); }
// The collected messages, each followed by an empty line; "" when the file parsed cleanly.
const char* rth_experiment_errors(rth_experiment* e) { return e->errors.c_str(); }

// Options of phase `idx`; every output pointer may be NULL.  transform receives the model's 16 floats, row-major.
int rth_experiment_phase(rth_experiment* e, int idx, char* name, int nameLen, RtDeviceOpts* device, int* lightCnt, float* fov, char* folder,
                         int folderLen, char* fname, int fnameLen, float* transform) {
    if (idx < 0 || idx >= e->reader->experiment.phaseCnt) {
        g_error = "rth_experiment_phase: no such phase";
        return -1;
    }
    const ExperimentPhase& p = e->reader->experiment.phases[idx];
    if (name) std::snprintf(name, (size_t)nameLen, "%s", p.phaseName.c_str());
    if (device) {
        std::memcpy(device, p.opts.device.abi(), sizeof(RtDeviceOpts));
        device->pad_ = 0;  // the C++ struct's tail padding carries no value
    }
    if (lightCnt) *lightCnt = p.opts.host.lightCnt;
    if (fov) *fov = p.opts.host.fov;
    if (folder) std::snprintf(folder, (size_t)folderLen, "%s", p.opts.host.model.folder.c_str());
    if (fname) std::snprintf(fname, (size_t)fnameLen, "%s", p.opts.host.model.fname.c_str());
    if (transform) std::memcpy(transform, p.opts.host.model.transform.x, 64);
    return 0;
}

// The scene of phase `idx`, built as RayTracer::initScene builds it.  flags: bit 0 tight BVH bounds, bit 1 BVHs
// built on the device, bit 2 re-derive device.scale / imageAspectRatio from the phase's fov, width and height
// (the reference leaves them at the defaults' values).
rth_scene* rth_experiment_scene(rth_experiment* e, int idx, const char* assetDir, int flags) {
    try {
        if (idx < 0 || idx >= e->reader->experiment.phaseCnt) throw std::invalid_argument("rth_experiment_scene: no such phase");
        Options opts = e->reader->experiment.phases[idx].opts;
        if (flags & 4) setView(opts, opts.device.width, opts.device.height, opts.host.fov);
        const BuildOptions bm = buildOptionsOf(flags);
        std::unique_ptr<RayTracer> rt = buildPhase(opts, assetDir ? assetDir : "", bm);
        auto* h = new rth_scene();
        h->rt = std::move(rt);
        h->refresh();
        return h;
    } catch (const std::exception& ex) {
        g_error = ex.what();
        return nullptr;
    }
}

// PicDumper: float RGBA -> 8-bit (x256, clamped) -> PNG.
int rth_write_png(const char* path, const float* rgba, int width, int height) {
    std::vector<float> px(rgba, rgba + (size_t)width * height * 4);
    return PicDumper(width, height).toDisk(px, path) ? 0 : -1;
}
void rth_floats_to_bytes(const float* rgba, size_t count, unsigned char* out) {
    std::vector<float> px(rgba, rgba + count);
    const std::vector<unsigned char> b = PicDumper::copyFloatsToCharVector(px);
    std::memcpy(out, b.data(), count);
}
// PNG decode to RGBA8 (the texture ingest path); returns 0 and fills w/h; out may be NULL to query the size.
int rth_read_png(const char* path, unsigned char* out, size_t outBytes, int* width, int* height) {
    std::vector<unsigned char> px;
    unsigned w = 0, h = 0;
    std::string err;
    if (!png::decodeFile(path, px, w, h, &err)) {
        g_error = err;
        return -1;
    }
    *width = (int)w;
    *height = (int)h;
    if (out) {
        if (outBytes < px.size()) return -2;
        std::memcpy(out, px.data(), px.size());
    }
    return 0;
}

// OBJ ingest probe: reads folder + file with ObjFileReader and returns the first model's geometry.
// counts = {models, vertices, indices, materials}; verts (RtVertex layout) / inds may be NULL to query sizes;
// texNames receives "diffuse\nnormal\nspecular" of the first material.
int rth_obj_read(const char* folder, const char* file, int startAtZero, int* counts, RtVertex* verts, int vertCap, int* inds, int indCap,
                 char* texNames, int texNamesLen) {
    try {
        const std::vector<model::Model> models = ObjFileReader().readFile(folder, file, startAtZero != 0);
        counts[0] = (int)models.size();
        counts[1] = counts[2] = counts[3] = 0;
        if (models.empty()) return 0;
        const model::Model& m = models[0];
        counts[1] = (int)m.getVertices().size();
        counts[2] = (int)m.getIndices().size();
        counts[3] = (int)m.getMaterials().size();
        if (verts) {
            if (vertCap < counts[1]) return -2;
            std::memcpy(verts, m.getVertices().data(), sizeof(Vertex) * m.getVertices().size());
        }
        if (inds) {
            if (indCap < counts[2]) return -2;
            std::memcpy(inds, m.getIndices().data(), sizeof(int) * m.getIndices().size());
        }
        if (texNames && texNamesLen > 0) {
            std::string t;
            if (!m.getMaterials().empty()) {
                const model::Material& mt = m.getMaterials()[0];
                t = mt.texturePath + "\n" + mt.normalMapPath + "\n" + mt.specularMapPath;
            }
            std::snprintf(texNames, (size_t)texNamesLen, "%s", t.c_str());
        }
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}

// Host BVH build over a triangle list, for parity checks against the oracle's builder.
int rth_bvh_triangles(const RtVertex* verts, int vertCnt, const RtTriangle* tris, int triCnt, int threshold, int tight,
                      RtLinearNode* nodesOut, int* nodeCntOut, int* orderOut) {
    try {
        std::vector<Vertex> v(reinterpret_cast<const Vertex*>(verts), reinterpret_cast<const Vertex*>(verts) + vertCnt);
        std::vector<Triangle> t(reinterpret_cast<const Triangle*>(tris), reinterpret_cast<const Triangle*>(tris) + triCnt);
        BVH::BvhFactory<Triangle> bvh(v, t, threshold, tight ? BVH::TIGHT_BOUNDS : BVH::REFERENCE_BOUNDS);
        std::memcpy(nodesOut, bvh.nodes.data(), bvh.nodes.size() * sizeof(LinearNode));
        *nodeCntOut = (int)bvh.nodes.size();
        std::memcpy(orderOut, bvh.orderedIdx.data(), bvh.orderedIdx.size() * sizeof(int));
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        return -1;
    }
}
void rth_mat_inverse(const float in[16], float out[16]) {
    Mat44f m;
    std::memcpy(m.x, in, 64);
    const Mat44f r = m.inversed();
    std::memcpy(out, r.x, 64);
}

// Host-API known-answer tests: the assertions of the reference's VectorTest.cpp,
// Tests.cpp, BoundingBoxTest.cpp and BvhTest.cpp, applied to this library's
// classes.  Returns the number of failed checks; `log` gets one line per case.
int rth_selftest(char* log, int logLen);

}  // extern "C"

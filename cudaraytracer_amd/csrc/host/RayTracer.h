// RayTracer: headless scene assembly + per-frame driver, the caller of the
// C-ABI.  Plays the role of the reference's RayTracer service
// (cudaTracer/RayTracer.cpp:50-78 scene set-up, :188-208 per-frame update,
// :259-267 light rotation) without window, GUI or input handling.
//
// Shapes are collected in creation order, the scene BVH (leaf threshold 1) is
// built over them, and they are stored in BVH leaf order; each shape keeps its
// creation index as `tag`.  Everything stays in host memory until upload().
#ifndef RT_HOST_RAYTRACER_H
#define RT_HOST_RAYTRACER_H

#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "BVH.h"
#include "Camera.h"
#include "DeviceMesh.h"
#include "Options.h"
#include "PicDumper.h"
#include "Scene.h"

namespace rt {

class RayTracer {
   public:
    Options options;
    Camera camera;
    std::vector<Light> lights;
    std::vector<Shape> shapes;  // creation order until finalizeScene(), leaf order after
    std::vector<LinearNode> nodes;
    std::vector<std::unique_ptr<DeviceMesh>> meshes;
    int sceneBvhDepth = 0;
    BVH::BoundsMode boundsMode = BVH::REFERENCE_BOUNDS;  // applies to every BVH this driver builds
    // upload() asks the library to recompute the boxes of the trees on the device (rt_set_upload_refit): with
    // REFERENCE_BOUNDS the trees keep the reference's shape, leaf order and tie behaviour, and rays stop paying for its
    // FLT_MIN-clamped boxes.  Set to false to walk the reference's boxes exactly as built.
    bool refitAtUpload = true;
    // upload() asks the library to build every mesh BVH again on the device (rt_set_upload_rebuild): trees cut on true
    // bounds instead of the reference's FLT_MIN-clamped ones - several times fewer tests per ray than the refit alone
    // leaves, at the price of the reference's order among equal-distance hits.
    bool rebuildAtUpload = false;
    // How the device cuts the mesh BVHs it builds - meshes without nodes, and all of them under rebuildAtUpload
    // (rt_set_device_bvh_rule): at the median, as the reference does, or by the binned surface-area heuristic (needs the
    // context's device builds on tight bounds, which is their default).
    BVH::SplitRule deviceSplitRule = BVH::MEDIAN_CUT;
    // Shadow rays towards lights behind the surface are not walked (rt_set_unlit_shadow_skip): the reference adds zeros for
    // them whatever they find (tracer.cu:181-196), so the image is the same bit for bit and the shadow launch is shorter.
    bool skipUnlitShadowRays = false;

    explicit RayTracer(Options options, BVH::BoundsMode bounds = BVH::REFERENCE_BOUNDS) : options(options), boundsMode(bounds) {}
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;
    ~RayTracer() { release(); }

    // ---- scene assembly -------------------------------------------------
    void addShape(Shape s) {
        s.tag = (int)shapes.size();
        shapes.push_back(s);
    }
    DeviceMesh& addMesh(std::unique_ptr<DeviceMesh> m) {
        meshes.push_back(std::move(m));
        addShape(meshes.back()->shape);
        return *meshes.back();
    }
    void addLight(const Light& l) { lights.push_back(l); }

    // The ring of options.host.lightCnt lights of RayTracer::addLights (cudaTracer/RayTracer.cpp:80-101): light i
    // sits at angle i * 2*pi / lightCnt, radius U(10, 50), height U(0, 25), intensity U(0, 0.5) per channel, power
    // U(0, 25).  The reference draws from the C library's rand(), never seeded, in a Windows build: the generator
    // here is that runtime's documented LCG (x = x * 214013 + 2531011, output bits 16..30, RAND_MAX 32767) from seed
    // 1.  The order in which the constructor arguments consume draws is the compiler's choice in the reference
    // (C++ leaves it unspecified); here it is left to right: power, position x y z, intensity r g b.
    void addLights(unsigned seed = 1) {
        unsigned state = seed;
        auto draw = [&state]() {
            state = state * 214013u + 2531011u;
            return (float)((state >> 16) & 0x7fffu) / 32767.0f;
        };
        const int cnt = options.host.lightCnt;
        const float radStep = (2.0f * PI) / cnt;
        const float intensityFactor = 0.5f, maxDist = 50.0f;
        for (int i = 0; i < cnt; ++i) {
            const float power = draw() * 25.0f;
            const float px = std::sin(i * radStep) * (draw() * (maxDist - 10) + 10);
            const float py = draw() * 25.0f;
            const float pz = std::cos(i * radStep) * (draw() * (maxDist - 10) + 10);
            const float r = draw() * intensityFactor;
            const float g = draw() * intensityFactor;
            const float b = draw() * intensityFactor;
            addLight(Light(Vec3f(px, py, pz), Vec3f(r, g, b), power));
        }
    }

    // The reference's 10x10 sphere field with a mirror and a glass ball
    // (cudaTracer/RayTracer.cpp:103-132).
    void addSpheres() {
        const int n = 10;
        const float dist = 3.0f;
        const float scale = 1.0f / n;
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) {
                const float x = -n * dist * 0.5f + i * dist;
                const float z = 10 + 0.5f + j * dist;
                Shape s = Shape(Sphere(Vec3f(x, -2.0f, z), 1.0f));
                s.material.materialType = Material::DIFFUSE_AND_GLOSSY;
                s.material.diffuseColor = Vec3f(i * scale, 0.1f, j * scale);
                s.material.Kd = 0.8;
                s.material.Ks = scale * i * scale * j * 0.8f + 0.1f;
                s.material.specularExponent = (i * i * i * j * j * j) + 1;
                addShape(s);
            }
        }
        Shape mirrorBall = Shape(Sphere(Vec3f(2.5f, 1.0f, 3.0f), 1.0f));
        mirrorBall.material.ior = 64;
        mirrorBall.material.materialType = Material::REFLECTION;
        mirrorBall.material.diffuseColor = Vec3f(0.722f, 0.451f, 0.2f);
        addShape(mirrorBall);
        Shape glassBall = Shape(Sphere(Vec3f(1.5f, -1.5f, 6.5f), 1.5f));
        glassBall.material.ior = 1.5f;
        glassBall.material.materialType = Material::REFLECTION_AND_REFRACTION;
        glassBall.material.diffuseColor = Vec3f(0.8f, 0.7f, 0.6f);
        addShape(glassBall);
    }

    // The reference's textured ground quad: 2x2 units scaled x100, UVs tiled x25,
    // WRAP addressing, ior 4 (cudaTracer/RayTracer.cpp:134-156).
    void addPlane(TextureSource diffuse, TextureSource normal, TextureSource specular) {
        const float uvTiles = 25.0f;
        std::vector<Vertex> vertices{Vertex(Vec3f(-1.0f, 0.0f, 1.0f), Vec2f(0.0f, 0.0f)),
                                     Vertex(Vec3f(1.0f, 0.0f, 1.0f), Vec2f(uvTiles, 0.0f)),
                                     Vertex(Vec3f(1.0f, 0.0f, -1.f), Vec2f(uvTiles, uvTiles)),
                                     Vertex(Vec3f(-1.0f, 0.0f, -1.f), Vec2f(0.0f, uvTiles))};
        std::vector<int> indices{0, 1, 3, 1, 2, 3};
        Material mtl;
        mtl.materialType = Material::DIFFUSE_AND_GLOSSY;
        mtl.ior = 4;
        const Mat44f transform = Mat44f::scale(100, 1, 100) * Mat44f::translate(0, -3, 0);
        addMesh(std::make_unique<DeviceMesh>(vertices, indices, 2, diffuse, normal, specular, Texture::WRAP, mtl, transform, boundsMode));
    }

    // Scene BVH over the shapes, shapes re-ordered into leaf order
    // (cudaTracer/RayTracer.cpp:72-77).  Call once, after the last add*.
    void finalizeScene() {
        if (shapes.empty()) throw std::logic_error("RayTracer: empty scene");
        BVH::BvhFactory<Shape> bvh(shapes, boundsMode);
        shapes = std::move(bvh.orderedPrims);
        nodes = std::move(bvh.nodes);
        sceneBvhDepth = bvh.depth;
    }

    // ---- per-frame state --------------------------------------------------
    // RayTracer::updateLights: every light turns about the Y axis by 25 deg/s.
    void updateLights(float dt) {
        const Mat44f rotY = Mat44f::rotationY(dt * 25);
        for (Light& l : lights) l.position = rotY.multPoint(l.position);
    }

    // The Scene handed to the kernel this frame (cudaTracer/RayTracer.cpp:196-203).
    Scene currentScene() {
        Scene scene;
        scene.lights = lights.data();
        scene.lightCnt = (int)lights.size();
        scene.shapes = shapes.data();
        scene.shapeCnt = (int)shapes.size();
        scene.orig = camera.getPosition();
        scene.camera = camera.getCamera().inversed();
        scene.nodes = nodes.data();
        scene.nodeCnt = (int)nodes.size();
        return scene;
    }

    // ---- GPU side (all through the C-ABI) ---------------------------------
    void upload(rt_ctx* context) {
        release();
        ctx = context;
        std::vector<RtShape> shapeStore;
        std::vector<RtMesh> meshStore;
        const RtSceneDesc desc = currentScene().describe(shapeStore, meshStore);
        check(rt_set_upload_refit(ctx, refitAtUpload ? 1 : 0), "rt_set_upload_refit");
        check(rt_set_upload_rebuild(ctx, rebuildAtUpload ? 1 : 0), "rt_set_upload_rebuild");
        check(rt_set_device_bvh_rule(ctx, deviceSplitRule == BVH::BINNED_SAH ? RT_BVH_BINNED_SAH : RT_BVH_MEDIAN_CUT), "rt_set_device_bvh_rule");
        check(rt_set_unlit_shadow_skip(ctx, skipUnlitShadowRays ? 1 : 0), "rt_set_unlit_shadow_skip");
        devScene = rt_scene_upload(ctx, &desc);
        if (!devScene) throw std::runtime_error(std::string("rt_scene_upload: ") + rt_last_error(ctx));
        check(rt_surface_create(ctx, options.device.width, options.device.height, &surface, &pitch), "rt_surface_create");
    }

    // The same for a node with several GPUs (rt_multi_*): the scene goes to devices[0] and from there to the peers, every
    // device renders its share of each frame's tiles, and the frame arrives in a surface on devices[0].
    // `balanceTiles`: deal the tiles out by the measured cost of a first frame instead of tile t -> device t % n.
    void upload(rt_multi* node, bool balanceTiles = true) {
        release();
        multi = node;
        ctx = rt_multi_context(node, 0);
        std::vector<RtShape> shapeStore;
        std::vector<RtMesh> meshStore;
        const RtSceneDesc desc = currentScene().describe(shapeStore, meshStore);
        check(rt_set_upload_refit(ctx, refitAtUpload ? 1 : 0), "rt_set_upload_refit");
        check(rt_set_upload_rebuild(ctx, rebuildAtUpload ? 1 : 0), "rt_set_upload_rebuild");
        check(rt_set_device_bvh_rule(ctx, deviceSplitRule == BVH::BINNED_SAH ? RT_BVH_BINNED_SAH : RT_BVH_MEDIAN_CUT), "rt_set_device_bvh_rule");
        for (int i = 0; i < rt_multi_device_count(node); ++i)  // a render-time switch of each device's own context
            check(rt_set_unlit_shadow_skip(rt_multi_context(node, i), skipUnlitShadowRays ? 1 : 0), "rt_set_unlit_shadow_skip");
        checkMulti(rt_multi_scene_upload(multi, &desc), "rt_multi_scene_upload");
        check(rt_surface_create(ctx, options.device.width, options.device.height, &surface, &pitch), "rt_surface_create");
        if (balanceTiles) {
            const RtCamera cam = currentScene().cameraDesc();
            checkMulti(rt_multi_balance(multi, options.device.abi(), &cam), "rt_multi_balance");
        }
    }

    // One frame: advance the lights, re-upload them, render (RayTracer::update).
    void update(float dt) {
        if (multi) {
            updateLights(dt);
            checkMulti(rt_multi_scene_set_lights(multi, reinterpret_cast<const RtLight*>(lights.data()), (int)lights.size()), "rt_multi_scene_set_lights");
            const RtCamera cam = currentScene().cameraDesc();
            checkMulti(rt_multi_render(multi, options.device.abi(), &cam, surface, pitch, options.device.blockSize), "rt_multi_render");
            checkMulti(rt_multi_sync(multi), "rt_multi_sync");
            return;
        }
        if (!devScene) throw std::logic_error("RayTracer: upload() first");
        updateLights(dt);
        check(rt_scene_set_lights(devScene, reinterpret_cast<const RtLight*>(lights.data()), (int)lights.size()),
              "rt_scene_set_lights");
        const RtCamera cam = currentScene().cameraDesc();
        check(rt_render(ctx, devScene, options.device.abi(), &cam, surface, pitch, options.device.blockSize, nullptr), "rt_render");
        check(rt_sync(ctx), "rt_sync");
    }

    std::vector<float> readFrame() {
        std::vector<float> px((size_t)options.device.width * options.device.height * 4);
        check(rt_surface_download(ctx, surface, pitch, options.device.width, options.device.height, px.data()),
              "rt_surface_download");
        return px;
    }

    bool dumpFrame(const std::string& path) {
        return PicDumper(options.device.width, options.device.height).toDisk(readFrame(), path);
    }

    void release() {
        if (multi) (void)rt_multi_sync(multi);  // (the node object keeps the scene copies; its owner destroys it)
        if (devScene) rt_scene_free(devScene);
        if (surface) rt_surface_free(ctx, surface);
        devScene = nullptr;
        surface = nullptr;
        multi = nullptr;
    }

   private:
    rt_multi* multi = nullptr;
    void checkMulti(int rc, const char* what) {
        if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_multi_last_error(multi));
    }
    rt_ctx* ctx = nullptr;
    rt_scene* devScene = nullptr;
    void* surface = nullptr;
    size_t pitch = 0;

    void check(int rc, const char* what) {
        if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(ctx));
    }
};

}  // namespace rt
#endif

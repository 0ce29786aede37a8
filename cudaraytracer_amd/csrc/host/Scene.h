// Mesh / Shape / Scene of the host API: the non-owning views the reference
// passes to its kernel (cudaTracer/Mesh.cuh:17-46, cudaTracer/Shape.cuh:8-40,
// cudaTracer/Scene.cuh:12-19), here pointing at HOST arrays.  Scene::describe()
// turns the pointer graph into the flat, index-based RtSceneDesc that crosses
// the C-ABI; the library copies and re-lays it out for the GPU.
#ifndef RT_HOST_SCENE_H
#define RT_HOST_SCENE_H

#include <cstring>
#include <vector>

#include "BoundingBox.h"
#include "Options.h"

namespace rt {

class Mesh {
   public:
    int triangleCnt = -1;
    Triangle* triangles = nullptr;  // mesh-BVH leaf order
    Vertex* vertices = nullptr;
    Texture diffuse;
    Texture normals;
    Texture specular;
    LinearNode* nodes = nullptr;
    Mat44f transformInverse;
    int vertexCnt = 0;  // sizes the upload needs (the reference never bounds-checks)
    int nodeCnt = 0;

    Mesh() = default;
    // Stores the INVERSE of `transform`: rays are taken to object space
    // (cudaTracer/Mesh.cuh:45,115-117).
    Mesh(Triangle* triangles, int triangleCnt, Vertex* vertices, Texture diffuse, Texture normals, Texture specular,
         LinearNode* nodes, Mat44f transform, int vertexCnt = 0, int nodeCnt = 0)
        : triangleCnt(triangleCnt),
          triangles(triangles),
          vertices(vertices),
          diffuse(diffuse),
          normals(normals),
          specular(specular),
          nodes(nodes),
          transformInverse(transform.inversed()),
          vertexCnt(vertexCnt),
          nodeCnt(nodeCnt) {}
};

struct Shape {
    enum ShapeKind { NOT_SET = -1, SPHERE, MESH };
    ShapeKind kind = NOT_SET;
    Sphere sphere;
    Mesh mesh;
    Material material;
    BoundingBox bb;
    int tag = -1;  // creation-order id; survives the scene-BVH reorder

    Shape() = default;
    // world box of a sphere: centre -/+ radius on every axis (cudaTracer/Shape.cuh:31-37)
    Shape(Sphere s) : kind(SPHERE), sphere(s), bb(s.center - s.radius, s.center + s.radius) {}
    Shape(Mesh m, BoundingBox bb) : kind(MESH), mesh(m), bb(bb) {}
    const Material* getObject() const { return &material; }
};

struct Scene {
    Light* lights = nullptr;
    int lightCnt = 0;
    Shape* shapes = nullptr;  // scene-BVH leaf order
    int shapeCnt = 0;
    Vec3f orig;
    Mat44f camera;  // inverse view matrix
    LinearNode* nodes = nullptr;
    int nodeCnt = 0;

    // Flat description for rt_scene_upload.  `shapeStore` / `meshStore` own the
    // converted tables and must outlive the returned struct.
    RtSceneDesc describe(std::vector<RtShape>& shapeStore, std::vector<RtMesh>& meshStore) const {
        shapeStore.clear();
        meshStore.clear();
        auto tex = [](const Texture& t) { return RtTexture{t.width, t.height, t.data, (int32_t)t.mode}; };
        for (int i = 0; i < shapeCnt; ++i) {
            const Shape& s = shapes[i];
            RtShape o;
            std::memset(&o, 0, sizeof o);
            o.kind = s.kind == Shape::SPHERE ? RT_SHAPE_SPHERE : RT_SHAPE_MESH;
            o.meshIdx = -1;
            o.tag = s.tag >= 0 ? s.tag : i;
            std::memcpy(&o.material, &s.material, sizeof(RtMaterial));
            for (int k = 0; k < 3; ++k) {
                o.bbmin[k] = s.bb.bbmin[k];
                o.bbmax[k] = s.bb.bbmax[k];
                o.center[k] = s.sphere.center[k];
            }
            o.radius = s.sphere.radius;
            o.radius2 = s.sphere.radius2;
            if (s.kind == Shape::MESH) {
                const Mesh& m = s.mesh;
                RtMesh d;
                std::memset(&d, 0, sizeof d);
                d.triangleCnt = m.triangleCnt;
                d.triangles = reinterpret_cast<const RtTriangle*>(m.triangles);
                d.vertexCnt = m.vertexCnt;
                d.vertices = reinterpret_cast<const RtVertex*>(m.vertices);
                d.diffuse = tex(m.diffuse);
                d.normals = tex(m.normals);
                d.specular = tex(m.specular);
                d.nodeCnt = m.nodeCnt;
                d.nodes = reinterpret_cast<const RtLinearNode*>(m.nodes);
                std::memcpy(d.transformInverse, m.transformInverse.x, sizeof d.transformInverse);
                o.meshIdx = (int32_t)meshStore.size();
                meshStore.push_back(d);
            }
            shapeStore.push_back(o);
        }
        RtSceneDesc desc;
        desc.lights = reinterpret_cast<const RtLight*>(lights);
        desc.lightCnt = lightCnt;
        desc.shapes = shapeStore.data();
        desc.shapeCnt = shapeCnt;
        desc.meshes = meshStore.data();
        desc.meshCnt = (int32_t)meshStore.size();
        desc.nodes = reinterpret_cast<const RtLinearNode*>(nodes);
        desc.nodeCnt = nodeCnt;
        return desc;
    }
    RtCamera cameraDesc() const {
        RtCamera c;
        for (int k = 0; k < 3; ++k) c.orig[k] = orig[k];
        std::memcpy(c.camera, camera.x, sizeof c.camera);
        return c;
    }
};

}  // namespace rt
#endif

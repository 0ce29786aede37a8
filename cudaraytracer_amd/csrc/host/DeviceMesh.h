// DeviceMesh: host-side owner of one triangle mesh on its way to the GPU -
// the role of the reference's CudaMesh (cudaTracer/CudaMesh.h:14-157):
//   vertices + index list -> triangles -> mesh BVH (leaf threshold 10) ->
//   triangles re-ordered into leaf order -> per-face tangents ->
//   three RGBA8 textures -> Mesh view + world-space Shape.
// It keeps everything in host vectors; the upload to device memory happens in
// one piece when the Scene crosses the C-ABI (rt_scene_upload).
#ifndef RT_HOST_DEVICEMESH_H
#define RT_HOST_DEVICEMESH_H

#include <stdexcept>
#include <string>
#include <vector>

#include "BVH.h"
#include "Model.h"
#include "Png.h"
#include "Scene.h"

namespace rt {

// A texture either comes from a PNG on disk or is handed over as RGBA8 pixels.
struct TextureSource {
    std::string fname;
    std::vector<unsigned char> rgba;
    unsigned width = 0, height = 0;
    TextureSource() = default;
    TextureSource(const std::string& f) : fname(f) {}
    TextureSource(const char* f) : fname(f) {}
    TextureSource(unsigned w, unsigned h, std::vector<unsigned char> px) : rgba(std::move(px)), width(w), height(h) {}
    static TextureSource solid(unsigned char r, unsigned char g, unsigned char b) {
        return TextureSource(1, 1, {r, g, b, 255});
    }
};

class DeviceMesh {
   public:
    std::vector<Vertex> vertices;
    std::vector<int> indices;
    std::vector<Triangle> triangles;  // leaf order
    std::vector<int> triangleSource;  // triangles[k] was face triangleSource[k] of the index list
    std::vector<LinearNode> nodes;
    int bvhDepth = 0;

    std::vector<unsigned char> diffuseData;
    std::vector<unsigned char> normalsData;
    std::vector<unsigned char> specularData;
    Mesh mesh;
    Shape shape;

    BVH::BoundsMode boundsMode = BVH::REFERENCE_BOUNDS;
    BVH::SplitRule splitRule = BVH::MEDIAN_CUT;
    int leafThreshold = 10;
    bool bvhOnDevice = false;  // nodes stay empty; rt_scene_upload builds the BVH on the GPU
    BoundingBox rootBox;       // box of the whole mesh (object space)

    DeviceMesh(std::vector<Vertex> vertices, std::vector<int> indices, int /*triCnt*/, TextureSource diffuse,
               TextureSource normals, TextureSource specular, Texture::AddressingMode mode, Material material,
               Mat44f transformation, BVH::BuildOptions build = BVH::BuildOptions())
        : boundsMode(build.bounds), splitRule(build.split), leafThreshold(build.leafThreshold), bvhOnDevice(build.onDevice) {
        prepareGeometry(std::move(vertices), std::move(indices));
        createShape(diffuse, normals, specular, mode, material, transformation);
    }

    // First material of the model names the three maps; CLAMP addressing
    // (cudaTracer/CudaMesh.h:37-47).
    DeviceMesh(const model::Model& model, Material material, Mat44f transformation,
               BVH::BuildOptions build = BVH::BuildOptions())
        : boundsMode(build.bounds), splitRule(build.split), leafThreshold(build.leafThreshold), bvhOnDevice(build.onDevice) {
        prepareGeometry(model.getVertices(), model.getIndices());
        if (model.getMaterials().empty()) throw std::invalid_argument("DeviceMesh: model without material");
        const model::Material& m = model.getMaterials()[0];
        createShape(m.texturePath, m.normalMapPath, m.specularMapPath, Texture::CLAMP, material, transformation);
    }

    DeviceMesh(const DeviceMesh&) = delete;  // mesh/shape point into this object's vectors
    DeviceMesh& operator=(const DeviceMesh&) = delete;

    void prepareGeometry(std::vector<Vertex> verts, std::vector<int> inds) {
        vertices = std::move(verts);
        indices = std::move(inds);
        if (indices.size() % 3 != 0 || indices.empty()) throw std::invalid_argument("DeviceMesh: index count");
        for (int idx : indices)
            if (idx < 0 || idx >= (int)vertices.size()) throw std::out_of_range("DeviceMesh: vertex index");
        std::vector<Triangle> tris;
        tris.reserve(indices.size() / 3);
        for (size_t i = 0; i < indices.size(); i += 3) tris.emplace_back(indices[i], indices[i + 1], indices[i + 2]);
        if (bvhOnDevice) {
            // triangles stay in index-list order (tangents of shared vertices then follow that order, not leaf order);
            // the device orders its own copy
            rootBox = BVH::emptyBox(boundsMode);
            for (const Triangle& t : tris) rootBox = BoundingBox::unionn(rootBox, BVH::triangleBox(t, vertices, boundsMode));
            triangles = std::move(tris);
            triangleSource.resize(triangles.size());
            for (size_t k = 0; k < triangleSource.size(); ++k) triangleSource[k] = (int)k;
            generateTangents();
            nodes.clear();
            bvhDepth = BVH::medianTreeDepth((int)triangles.size(), 10);
            return;
        }
        if (splitRule != BVH::MEDIAN_CUT) {
            // the tangent frame a shared vertex ends up with depends on the order the faces are visited in: keep the
            // order of the median-cut tree for that, whatever tree the rays walk
            BVH::BvhFactory<Triangle> referenceOrder(vertices, tris, 10, boundsMode);
            triangles = std::move(referenceOrder.orderedPrims);
            generateTangents();
        }
        BVH::BvhFactory<Triangle> bvh(vertices, tris, leafThreshold, boundsMode, splitRule);
        triangles = std::move(bvh.orderedPrims);
        triangleSource = std::move(bvh.orderedIdx);
        if (splitRule == BVH::MEDIAN_CUT) generateTangents();
        nodes = std::move(bvh.nodes);
        bvhDepth = bvh.depth;
        rootBox = nodes[0].bb;
    }

    void createShape(const TextureSource& diffuseSrc, const TextureSource& normalsSrc, const TextureSource& specularSrc,
                     Texture::AddressingMode mode, Material material, Mat44f transformation) {
        Texture diffuse = loadTexture(diffuseSrc, &diffuseData, mode);
        Texture normals = loadTexture(normalsSrc, &normalsData, mode);
        Texture specular = loadTexture(specularSrc, &specularData, mode);
        mesh = Mesh(triangles.data(), (int)triangles.size(), vertices.data(), diffuse, normals, specular, nodes.empty() ? nullptr : nodes.data(),
                    transformation, (int)vertices.size(), (int)nodes.size());
        // World box = the two extreme corners of the root box, transformed
        // (cudaTracer/CudaMesh.h:77-79): exact for scale + translate, loose for rotations.
        BoundingBox shapeBb = rootBox;
        shapeBb.bbmin = transformation.multPoint(shapeBb.bbmin);
        shapeBb.bbmax = transformation.multPoint(shapeBb.bbmax);
        shape = Shape(mesh, shapeBb);
        shape.material = material;
    }

    static Texture loadTexture(const TextureSource& src, std::vector<unsigned char>* data, Texture::AddressingMode mode) {
        unsigned width = src.width, height = src.height;
        if (!src.fname.empty()) {
            std::string err;
            if (!png::decodeFile(src.fname, *data, width, height, &err))
                throw std::runtime_error("DeviceMesh: cannot decode " + src.fname + ": " + err);
        } else {
            *data = src.rgba;
        }
        if (width == 0 || height == 0 || data->size() != (size_t)width * height * 4)
            throw std::invalid_argument("DeviceMesh: texture size");
        return Texture(width, height, data->data(), mode);
    }

    // One tangent frame per face, written to all three corners; faces are
    // visited in BVH leaf order and later faces overwrite shared vertices
    // (cudaTracer/CudaMesh.h:106-130).
    void generateTangents() {
        Vec3f tangent, bitangent;
        for (const Triangle& t : triangles) {
            calcFaceTangentAndBitangent(vertices[t[0]], vertices[t[1]], vertices[t[2]], &tangent, &bitangent);
            for (int c = 0; c < 3; ++c) {
                vertices[t[c]].tangent = tangent;
                vertices[t[c]].bitangent = bitangent;
            }
        }
    }

    // cudaTracer/CudaMesh.h:132-157
    static void calcFaceTangentAndBitangent(Vertex v1, Vertex v2, Vertex v3, Vec3f* outTangent, Vec3f* outBitangent) {
        const Vec3f edge1 = v2.position - v1.position;
        const Vec3f edge2 = v3.position - v1.position;
        const Vec2f uv1 = v2.texCoord - v1.texCoord;
        const Vec2f uv2 = v3.texCoord - v1.texCoord;
        const float f = 1.0f / (uv1[0] * uv2[1] - uv2[0] * uv1[1]);
        for (int k = 0; k < 3; ++k) (*outTangent)[k] = (uv2[1] * edge1[k] - uv1[1] * edge2[k]) * f;
        *outTangent = outTangent->normalized();
        for (int k = 0; k < 3; ++k) (*outBitangent)[k] = (-uv2[0] * edge1[k] + uv1[0] * edge2[k]) * f;
        *outBitangent = outBitangent->normalized();
    }
};

}  // namespace rt
#endif

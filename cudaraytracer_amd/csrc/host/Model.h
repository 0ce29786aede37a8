// model::Model - an indexed triangle list plus the texture paths of its
// materials, as produced by ObjFileReader and CubeModelFactory
// (cudaTracer/Model.h:9-75).
#ifndef RT_HOST_MODEL_H
#define RT_HOST_MODEL_H

#include <string>
#include <vector>

#include "Types.h"

namespace rt {
namespace model {

struct Material {
    std::string mtlName;
    std::string texturePath;
    std::string normalMapPath;
    std::string specularMapPath;
};

class Model {
   public:
    std::string name;
    bool useBlendMap = false;
    std::string blendMapPath;
    std::vector<int> indices;
    std::vector<Vertex> vertices;
    std::vector<Material> materials;

    void clear() { *this = Model(); }
    void addVertex(const Vertex& v) { vertices.push_back(v); }
    void setVertices(std::vector<Vertex> v) { vertices = std::move(v); }
    void addIndex(int i) { indices.push_back(i); }
    void setIndices(std::vector<int> i) { indices = std::move(i); }
    void setBlendMapPath(const std::string& p) { blendMapPath = p; }
    void addMaterial(const Material& m) { materials.push_back(m); }
    void addMaterial(const std::string& texture, const std::string& normalMap, const std::string& specularMap) {
        materials.push_back(Material{"", texture, normalMap, specularMap});
    }
    void setUseBlendMap(bool b) { useBlendMap = b; }

    const std::vector<Vertex>& getVertices() const { return vertices; }
    const std::vector<int>& getIndices() const { return indices; }
    const std::string& getBlendMapPath() const { return blendMapPath; }
    const std::vector<Material>& getMaterials() const { return materials; }
    int getNumMaterials() const { return (int)materials.size(); }
    int getNumVertices() const { return (int)vertices.size(); }
    int getNumIndices() const { return (int)indices.size(); }
    bool getUseBlendMap() const { return useBlendMap; }
};

}  // namespace model
}  // namespace rt
#endif

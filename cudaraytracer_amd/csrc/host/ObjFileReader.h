// ObjFileReader: Wavefront OBJ / MTL ingest with the reference reader's
// behaviour (cudaTracer/ObjFileReader.cpp:28-292):
//   - words are split on single spaces, empty words dropped (so "v  x y z" works);
//   - faces must be "v/vt/vn" triples; triangles are taken as is, quads are cut
//     into (0,1,2) and (0,2,3); other polygon sizes are ignored;
//   - the output is DE-INDEXED: one Vertex per face corner, indices 0,1,2,...;
//   - `g` closes the current model and starts the next; `mtllib` loads materials,
//     `usemtl` attaches the named one; map_Kd/map_Ka -> diffuse, map_Ks/map_Ns ->
//     specular, map_bump/bump -> normal map, paths prefixed with the folder;
//   - without any `vt` line, corner i gets UV (0.3 i, 0.3 i).
// Errors (missing file, bad index) throw instead of popping up a message box.
#ifndef RT_HOST_OBJFILEREADER_H
#define RT_HOST_OBJFILEREADER_H

#include <algorithm>
#include <array>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Model.h"

namespace rt {

class ObjFileReader {
   public:
    std::vector<model::Model> readFile(std::string pFolder, std::string pFileName, bool pStartAtZero) {
        currModel.clear();
        models.clear();
        numIndices = 0;
        readNorm.clear();
        readPos.clear();
        readTexCoord.clear();
        materials.clear();
        fileName = pFileName;
        folder = pFolder;
        startAtZero = pStartAtZero;

        std::ifstream objFile(folder + fileName);
        if (!objFile) throw std::runtime_error("Could not read obj file: " + folder + fileName);
        std::string line;
        while (std::getline(objFile, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            std::vector<std::string> words = split(line, ' ');
            if (words.empty()) continue;
            words.erase(words.begin());  // drop the keyword
            const char k0 = line[0], k1 = line.size() > 1 ? line[1] : ' ';
            if (k0 == 'v') {
                if (k1 == 't')
                    readTexCoord.push_back({number(words, 0), number(words, 1)});
                else if (k1 == 'n')
                    readNorm.push_back({number(words, 0), number(words, 1), number(words, 2)});
                else
                    readPos.push_back({number(words, 0), number(words, 1), number(words, 2)});
            } else if (k0 == 'f') {
                readFaces(words);
            } else if (k0 == 'g') {
                createModel();
            } else if (k0 == 'm') {
                if (!words.empty()) readMtlFile(words[0]);
            } else if (k0 == 'u') {
                for (const model::Material& m : materials)
                    if (!words.empty() && m.mtlName == words.front()) currModel.addMaterial(m);
            }
        }
        createModel();
        return models;
    }

    static std::vector<std::string> split(const std::string& str, char delim) {
        std::vector<std::string> out;
        std::string cur;
        for (char c : str) {
            if (c != delim) {
                cur.push_back(c);
            } else {
                if (!cur.empty()) out.push_back(cur);
                cur.clear();
            }
        }
        if (!cur.empty()) out.push_back(cur);
        return out;
    }

   private:
    model::Model currModel;
    std::vector<model::Model> models;
    std::string fileName, folder;
    bool startAtZero = false;
    int numIndices = 0;
    std::vector<std::array<float, 3>> readNorm, readPos;
    std::vector<std::array<float, 2>> readTexCoord;
    std::vector<model::Material> materials;

    static float number(const std::vector<std::string>& words, size_t i) {
        if (i >= words.size()) throw std::runtime_error("obj: missing coordinate");
        return (float)std::atof(words[i].c_str());
    }

    void readMtlFile(const std::string& mtlFileName) {
        std::ifstream mtlFile(folder + mtlFileName);
        if (!mtlFile.good()) return;
        model::Material material;
        std::string line;
        while (std::getline(mtlFile, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) continue;
            std::stringstream ss(line);
            std::string prefix, value;
            ss >> prefix;
            std::transform(prefix.begin(), prefix.end(), prefix.begin(), [](unsigned char c) { return (char)std::tolower(c); });
            if (prefix == "newmtl") {
                if (!material.mtlName.empty()) {
                    materials.push_back(material);
                    material = model::Material();
                }
                ss >> material.mtlName;
            }
            if (prefix == "map_kd" || prefix == "map_ka") {
                ss >> value;
                material.texturePath = folder + value;
            } else if (prefix == "map_ks" || prefix == "map_ns") {
                ss >> value;
                material.specularMapPath = folder + value;
            } else if (prefix == "map_bump" || prefix == "bump") {
                ss >> value;
                material.normalMapPath = folder + value;
            }
        }
        materials.push_back(material);
    }

    void readFaces(const std::vector<std::string>& words) {
        if (words.size() == 3) {
            createFace({words[0], words[1], words[2]});
        } else if (words.size() == 4) {
            createFace({words[0], words[1], words[2]});
            createFace({words[0], words[2], words[3]});
        }
    }

    void createFace(const std::vector<std::string>& corners) {
        for (int i = 0; i < 3; ++i) {
            const std::vector<std::string> el = split(corners[i], '/');
            if (el.size() < 3) throw std::runtime_error("obj: face corner must be v/vt/vn");
            const int posIdx = std::atoi(el[0].c_str()), texIdx = std::atoi(el[1].c_str()), normIdx = std::atoi(el[2].c_str());
            Vertex vertex;
            vertex.tangent = Vec3f();
            vertex.bitangent = Vec3f();
            const auto& p = at(readPos, posIdx - 1);
            vertex.position = Vec3f(p[0], p[1], p[2]);
            if (!readTexCoord.empty()) {
                const auto& t = at(readTexCoord, texIdx - 1);
                vertex.texCoord = Vec2f(t[0], t[1]);
            } else {
                vertex.texCoord = Vec2f(0.3f * i, 0.3f * i);
            }
            const auto& n = at(readNorm, startAtZero ? normIdx : normIdx - 1);
            vertex.normal = Vec3f(n[0], n[1], n[2]);
            currModel.addVertex(vertex);
            currModel.addIndex(numIndices++);
        }
    }

    template <class V>
    static const typename V::value_type& at(const V& v, int i) {
        if (i < 0 || i >= (int)v.size()) throw std::runtime_error("obj: index out of range");
        return v[i];
    }

    void createModel() {
        if (currModel.getNumIndices() > 0) {
            currModel.setUseBlendMap(false);
            currModel.name = fileName;
            models.push_back(currModel);
        }
        currModel.clear();
        numIndices = 0;
    }
};

}  // namespace rt
#endif

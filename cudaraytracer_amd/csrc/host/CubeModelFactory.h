// CubeModelFactory: a 24-vertex / 12-triangle cube spanning [-1, 1]^3 with one
// flat normal and a full UV square per face.  Vertex and index tables are the
// scene data of cudaTracer/CubeModelFactory.cpp:12-67 (faces in the order
// +Z, -Z, +Y, -Y, +X, -X; two triangles {0,1,2},{2,1,3} per face).
#ifndef RT_HOST_CUBEMODELFACTORY_H
#define RT_HOST_CUBEMODELFACTORY_H

#include <string>
#include <vector>

#include "Model.h"

namespace rt {

class CubeModelFactory {
   public:
    static std::vector<Vertex> createVertices() {
        struct Face {
            float n[3];
            float corner[4][3];
        };
        // corner order per face matches UVs (0,0) (0,1) (1,0) (1,1)
        static const Face faces[6] = {
            {{0, 0, 1}, {{-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}}},
            {{0, 0, -1}, {{-1, -1, -1}, {-1, 1, -1}, {1, -1, -1}, {1, 1, -1}}},
            {{0, 1, 0}, {{-1, 1, -1}, {-1, 1, 1}, {1, 1, -1}, {1, 1, 1}}},
            {{0, -1, 0}, {{-1, -1, -1}, {1, -1, -1}, {-1, -1, 1}, {1, -1, 1}}},
            {{1, 0, 0}, {{1, -1, -1}, {1, 1, -1}, {1, -1, 1}, {1, 1, 1}}},
            {{-1, 0, 0}, {{-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}}},
        };
        std::vector<Vertex> vertices;
        for (const Face& f : faces)
            for (int c = 0; c < 4; ++c)
                vertices.push_back(Vertex(f.corner[c][0], f.corner[c][1], f.corner[c][2], f.n[0], f.n[1], f.n[2],
                                          (float)(c >> 1), (float)(c & 1)));
        return vertices;
    }

    static std::vector<int> createIndicies() {
        std::vector<int> indices;
        for (int face = 0; face < 6; ++face) {
            const int b = 4 * face;
            const int quad[6] = {b, b + 1, b + 2, b + 2, b + 1, b + 3};
            indices.insert(indices.end(), quad, quad + 6);
        }
        return indices;
    }

    static model::Model createCubeObject(const std::string& texFilePath) {
        model::Model m;
        m.setIndices(createIndicies());
        m.setVertices(createVertices());
        m.addMaterial(texFilePath, "", "");
        m.setUseBlendMap(false);
        m.name = "Cube from CubeModelfactory";
        return m;
    }
};

}  // namespace rt
#endif

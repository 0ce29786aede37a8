// Plain scene data types of the host API.  Each is layout-compatible with the
// C-ABI struct of the same role in include/rt_capi.h (checked below), so arrays
// of them cross the boundary without conversion.
//   Triangle  cudaTracer/Triangle.cuh:6-15     Vertex   cudaTracer/Vertex.cuh:6-38
//   Ray       cudaTracer/Ray.cuh:6-20          Material cudaTracer/Material.cuh:6-31
//   Light     cudaTracer/Light.cuh:6-17        Texture  cudaTracer/Texture.cuh:6-21
//   Sphere    cudaTracer/Sphere.cuh:30-40      LinearNode cudaTracer/LinearNode.cuh:4-13
#ifndef RT_HOST_TYPES_H
#define RT_HOST_TYPES_H

#include <cfloat>
#include <cstdint>
#include <vector>

#include "../../../include/rt_capi.h"
#include "Mat.h"
#include "Vec.h"

namespace rt {

struct Triangle {
    int i[3];
    Triangle() : i{-1, -1, -1} {}
    Triangle(int v0, int v1, int v2) : i{v0, v1, v2} {}
    Triangle(int first, const int v[]) : i{v[first], v[first + 1], v[first + 2]} {}
    int& operator[](int idx) { return i[idx]; }
    const int& operator[](int idx) const { return i[idx]; }
};

struct Vertex {
    Vec3f normal;
    Vec3f tangent;
    Vec3f bitangent;
    Vec2f texCoord;
    Vec3f position;
    // easily recognisable defaults, as in the reference
    Vertex() : normal(4.56f, 5.67f, 6.78f), texCoord(7.89f, 8.90f), position(1.23f, 2.34f, 3.45f) {}
    Vertex(float px, float py, float pz, float nx, float ny, float nz, float u, float v)
        : normal(nx, ny, nz), texCoord(u, v), position(px, py, pz) {}
    Vertex(Vec3f pos, Vec2f st) : normal(0.0f, 1.0f, 0.0f), texCoord(st), position(pos) {}
    Vertex(Vec3f pos, Vec3f norm, Vec2f st) : normal(norm), texCoord(st), position(pos) {}
};

struct Ray {
    Vec3f origin;
    Vec3f dir;
    float tMax;
    Ray(const Vec3f& o, const Vec3f& d, float tMax = FLT_MAX) : origin(o), dir(d), tMax(tMax) {}
    Vec3f operator()() const { return origin + dir * tMax; }
    Vec3f operator()(float t) const { return origin + dir * t; }
};

struct Material {
    enum Type { DIFFUSE_AND_GLOSSY, REFLECTION_AND_REFRACTION, REFLECTION };
    Type materialType = DIFFUSE_AND_GLOSSY;
    float ior = 1.3f;
    float Kd = 0.8f, Ks = 0.2f;
    Vec3f diffuseColor = Vec3f(0.2f, 0.2f, 0.2f);
    float specularExponent = 25.0f;
    Vec3f evalDiffuseColor(const Vec2f&) const { return diffuseColor; }
};

struct Light {
    Vec3f position;
    Vec3f intensity;
    float power = 1.0f;
    Light() = default;
    Light(const Vec3f& p, const Vec3f& i, float power) : position(p), intensity(i), power(power) {}
};

struct Texture {
    int width = -1;
    int height = -1;
    unsigned char* data = nullptr;  // RGBA8, host memory
    enum AddressingMode { WRAP, CLAMP };
    AddressingMode mode = WRAP;
    Texture() = default;
    Texture(unsigned width, unsigned height, unsigned char* data, AddressingMode mode)
        : width((int)width), height((int)height), data(data), mode(mode) {}
};

struct Sphere {
    Vec3f center;
    float radius = 0, radius2 = 0;
    Sphere() = default;
    Sphere(const Vec3f& c, const float& r) : center(c), radius(r), radius2(r * r) {}
};

static_assert(sizeof(Triangle) == sizeof(RtTriangle) && sizeof(Triangle) == 12, "Triangle layout");
static_assert(sizeof(Vertex) == sizeof(RtVertex) && sizeof(Vertex) == 56, "Vertex layout");
static_assert(sizeof(Material) == sizeof(RtMaterial) && sizeof(Material) == 32, "Material layout");
static_assert(sizeof(Light) == sizeof(RtLight) && sizeof(Light) == 28, "Light layout");
static_assert(sizeof(Ray) == 28, "Ray layout");
static_assert(sizeof(Sphere) == 20, "Sphere layout");
static_assert((int)Texture::WRAP == RT_TEX_WRAP && (int)Texture::CLAMP == RT_TEX_CLAMP, "Texture modes");
static_assert((int)Material::REFLECTION == RT_MAT_REFLECTION, "Material types");

}  // namespace rt
#endif

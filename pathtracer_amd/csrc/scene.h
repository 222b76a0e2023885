// Host scene model: the same public arrays, element layouts and material rules
// as the reference's Scene (scene.h:15-223), without the GL/glm dependencies.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "vecmath.h"

namespace pt {

class Scene {
public:
    // scene.h:18-25 -- 24 bytes; texture coordinates are always (0,0,0).
    struct VertexData {
        Vec3 normal;
        Vec3 textureCoordinates;
    };
    // scene.h:27-38 -- 80 bytes: radiance (w = 0) + column-major transform.
    struct Light {
        Light() = default;
        Light(Vec3 radiance, const float transform[16]);
        // translate(position) * eulerAngleYXZ(rotation.y, rotation.x, rotation.z) * scale(scale)
        Light(Vec3 radiance, Vec3 position, Vec3 rotation, Vec3 scale);
        float radiance[4];
        float transform[16];
    };
    // scene.h:40-51 -- 32 bytes.
    struct Material {
        Vec3 albedo;
        float roughness;
        float metallic;
        float ior;
        float anisotropy;
        float transmission;
    };

    std::vector<Vec3> m_vertices;
    std::vector<VertexData> m_vertexData;  // 3 per triangle
    std::vector<Vec4> m_indices;         // (i0, i1, i2, global triangle index) stored as floats
    std::vector<Light> m_lights;
    uint32_t m_lightCount[4] = {0, 0, 0, 0};
    std::vector<Material> m_materials;
    std::vector<uint32_t> m_materialMap;   // per triangle; 0xFFFFFFFF when the face had no usemtl

    // scene.h:64-76: one triangle and one unit-radiance light.
    Scene();
    // scene.h:79-218.  On a load failure the message goes to m_error and the
    // scene stays empty, as the reference leaves it.
    explicit Scene(const char *objFilename, const char *mtlRoot = nullptr);
    explicit Scene(const std::string &filename) : Scene(filename.c_str()) {}

    bool ok() const { return m_error.empty() || !m_vertices.empty(); }
    std::string m_error;

    // The light main.cpp:122-123 installs: radiance 4, position (0,1.95,0),
    // rotation (3.14/2, 0, 0), scale (1.25, 1.25, 1).
    void setDefaultLight();

    // Deterministic stress scene: every shape except the first (the room) of
    // `base` is copied onto an n x n x n lattice inside [-2,2]^3, each copy
    // scaled by 1/n about `center`.  Triangle order: room first, then cells in
    // x-major order.  Materials and lights are kept.
    static Scene instanceLattice(const Scene &base, uint32_t roomTriangles, uint32_t n);
};

static_assert(sizeof(Scene::VertexData) == 24, "VertexData layout");
static_assert(sizeof(Scene::Light) == 80, "Light layout");
static_assert(sizeof(Scene::Material) == 32, "Material layout");

}  // namespace pt

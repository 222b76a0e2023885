#include "scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "obj_loader.h"

namespace pt {

namespace {

// Column-major 4x4 helpers evaluating in the order glm 0.9.9.8 does, so that
// the light transform carries the same bits (and zero signs) as the
// reference's glm::translate * glm::eulerAngleYXZ * glm::scale
// (scene.h:33, glm/gtx/euler_angles.inl:220-252, glm/ext/matrix_transform.inl).
struct Mat4 {
    float c[4][4];  // c[column][row]
};

Mat4 identity()
{
    Mat4 m;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) m.c[i][j] = (i == j) ? 1.f : 0.f;
    return m;
}

// Result[j] = ((A[0]*B[j][0] + A[1]*B[j][1]) + A[2]*B[j][2]) + A[3]*B[j][3]
Mat4 mul(const Mat4 &a, const Mat4 &b)
{
    Mat4 r;
    for (int j = 0; j < 4; ++j)
        for (int row = 0; row < 4; ++row)
            r.c[j][row] = ((a.c[0][row] * b.c[j][0] + a.c[1][row] * b.c[j][1]) + a.c[2][row] * b.c[j][2]) +
                          a.c[3][row] * b.c[j][3];
    return r;
}

Mat4 translate(const Mat4 &m, Vec3 v)
{
    Mat4 r = m;
    for (int row = 0; row < 4; ++row)
        r.c[3][row] = ((m.c[0][row] * v.x + m.c[1][row] * v.y) + m.c[2][row] * v.z) + m.c[3][row];
    return r;
}

Mat4 scale(const Mat4 &m, Vec3 v)
{
    Mat4 r;
    for (int row = 0; row < 4; ++row) {
        r.c[0][row] = m.c[0][row] * v.x;
        r.c[1][row] = m.c[1][row] * v.y;
        r.c[2][row] = m.c[2][row] * v.z;
        r.c[3][row] = m.c[3][row];
    }
    return r;
}

Mat4 euler_yxz(float yaw, float pitch, float roll)
{
    float ch = std::cos(yaw), sh = std::sin(yaw);
    float cp = std::cos(pitch), sp = std::sin(pitch);
    float cb = std::cos(roll), sb = std::sin(roll);
    Mat4 r;
    r.c[0][0] = ch * cb + sh * sp * sb;
    r.c[0][1] = sb * cp;
    r.c[0][2] = -sh * cb + ch * sp * sb;
    r.c[0][3] = 0.f;
    r.c[1][0] = -ch * sb + sh * sp * cb;
    r.c[1][1] = cb * cp;
    r.c[1][2] = sb * sh + ch * sp * cb;
    r.c[1][3] = 0.f;
    r.c[2][0] = sh * cp;
    r.c[2][1] = -sp;
    r.c[2][2] = ch * cp;
    r.c[2][3] = 0.f;
    r.c[3][0] = 0.f;
    r.c[3][1] = 0.f;
    r.c[3][2] = 0.f;
    r.c[3][3] = 1.f;
    return r;
}

}  // namespace

Scene::Light::Light(Vec3 rad, const float t[16])
{
    radiance[0] = rad.x;
    radiance[1] = rad.y;
    radiance[2] = rad.z;
    radiance[3] = 0.f;
    std::memcpy(transform, t, sizeof(transform));
}

Scene::Light::Light(Vec3 rad, Vec3 position, Vec3 rotation, Vec3 scl)
{
    Mat4 m = mul(mul(translate(identity(), position), euler_yxz(rotation.y, rotation.x, rotation.z)),
                 scale(identity(), scl));
    radiance[0] = rad.x;
    radiance[1] = rad.y;
    radiance[2] = rad.z;
    radiance[3] = 0.f;
    std::memcpy(transform, m.c, sizeof(transform));
}

Scene::Scene()
{
    m_vertices = {make3(-1.f, -1.f, 0.f), make3(1.f, -1.f, 0.f), make3(0.f, 1.f, 0.f)};
    VertexData vd{make3(0.f, 0.f, -1.f), make3(0.f, 0.f, 0.f)};
    m_vertexData = {vd, vd, vd};
    m_indices = {Vec4{0.f, 1.f, 2.f, 0.f}};
    m_lights = {Light(make3(1.f, 1.f, 1.f), make3(0.f, 2.f, 0.f), make3(3.14f / 2.f, 0.f, 0.f),
                      make3(2.f, 1.f, 1.f))};
    m_lightCount[0] = 1;
}

void Scene::setDefaultLight()
{
    m_lights = {Light(make3(4.f, 4.f, 4.f), make3(0.f, 1.95f, 0.f), make3(3.14f / 2.f, 0.f, 0.f),
                      make3(1.25f, 1.25f, 1.f))};
    m_lightCount[0] = 1;
    m_lightCount[1] = m_lightCount[2] = m_lightCount[3] = 0;
}

Scene::Scene(const char *objFilename, const char *mtlRoot)
{
    ObjData obj;
    std::string err;
    if (!load_obj(objFilename, mtlRoot, &obj, &err)) {
        m_error = err.empty() ? std::string("failed to load ") + objFilename : err;
        return;
    }

    size_t vertexCount = obj.vertices.size() / 3;
    m_vertices.reserve(vertexCount);
    for (size_t i = 0; i < vertexCount; ++i)
        m_vertices.push_back(make3(obj.vertices[3 * i], obj.vertices[3 * i + 1], obj.vertices[3 * i + 2]));

    // Triangles in shape order, faces in file order (scene.h:101-136).
    for (const ObjShape &shape : obj.shapes) {
        size_t first = m_indices.size();
        size_t faces = shape.material_ids.size();
        for (size_t f = 0; f < faces; ++f) {
            Vec4 tri;
            for (int k = 0; k < 3; ++k) {
                const ObjIndex &idx = shape.indices[3 * f + k];
                tri[k] = static_cast<float>(idx.vertex_index);
                Vec3 normal = make3(0.f, 0.f, -1.f);
                if (idx.normal_index != -1)
                    normal = make3(obj.normals[3 * idx.normal_index + 0], obj.normals[3 * idx.normal_index + 1],
                                   obj.normals[3 * idx.normal_index + 2]);
                m_vertexData.push_back(VertexData{normal, make3(0.f, 0.f, 0.f)});
            }
            tri.w = static_cast<float>(first + f);
            m_materialMap.push_back(static_cast<uint32_t>(shape.material_ids[f]));
            m_indices.push_back(tri);
        }
    }

    // Material rules (scene.h:139-217).
    for (const ObjMaterial &mat : obj.materials) {
        const bool highlight = (mat.illum == 2);
        const bool reflection = (mat.illum == 3);
        Material out;
        out.albedo = make3(mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]);
        out.ior = mat.ior;
        if (!mat.isPBR) {
            // Legacy MTL: Blender's shininess -> roughness map; mirror-ness from Ka.
            if (mat.shininess < 0.f && highlight) {
                out.roughness = 0.f;
            } else {
                float ns = std::max(0.f, std::min(mat.shininess, 1000.f));
                out.roughness = 1.f - std::sqrt(ns / 1000.f);
            }
            out.metallic = 0.f;
            if (reflection) {
                out.metallic = (mat.ambient[0] + mat.ambient[1] + mat.ambient[2]) / 3.f;
                if (out.metallic < 0.f) out.metallic = 1.f;
            }
            out.anisotropy = 0.f;
            out.transmission = 0.f;
        } else {
            out.roughness = mat.roughness;
            out.metallic = mat.metallic;
            out.anisotropy = mat.anisotropy;
            out.transmission = (mat.transmittance[0] + mat.transmittance[1] + mat.transmittance[2]) / 3.f;
        }
        m_materials.push_back(out);
    }
}

Scene Scene::instanceLattice(const Scene &base, uint32_t roomTriangles, uint32_t n)
{
    Scene s;
    s.m_vertices.clear();
    s.m_vertexData.clear();
    s.m_indices.clear();
    s.m_lights = base.m_lights;
    std::memcpy(s.m_lightCount, base.m_lightCount, sizeof(s.m_lightCount));
    s.m_materials = base.m_materials;

    const size_t total = base.m_indices.size();
    const size_t room = std::min<size_t>(roomTriangles, total);
    const size_t content = total - room;

    // Each copy re-emits the whole vertex array (indices stay below 2^24, the
    // largest integer the float index format holds exactly) and the
    // triangles that reference it.
    const size_t nverts = base.m_vertices.size();
    auto emit = [&](size_t tri, size_t vertexBase) {
        Vec4 idx;
        for (int k = 0; k < 3; ++k) {
            idx[k] = static_cast<float>(vertexBase + static_cast<size_t>(base.m_indices[tri][k]));
            s.m_vertexData.push_back(base.m_vertexData[3 * tri + k]);
        }
        idx.w = static_cast<float>(s.m_indices.size());
        s.m_indices.push_back(idx);
        s.m_materialMap.push_back(base.m_materialMap[tri]);
    };

    s.m_vertices = base.m_vertices;
    for (size_t t = 0; t < room; ++t) emit(t, 0);
    if (content == 0 || n == 0) return s;

    // Bounding box of the content; copies are scaled about its centre.
    Vec3 lo = make3(INFINITY, INFINITY, INFINITY), hi = make3(-INFINITY, -INFINITY, -INFINITY);
    for (size_t t = room; t < total; ++t)
        for (int k = 0; k < 3; ++k) {
            Vec3 v = base.m_vertices[static_cast<size_t>(base.m_indices[t][k])];
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], v[a]);
                hi[a] = std::max(hi[a], v[a]);
            }
        }
    Vec3 center = make3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
    const float cell = 4.f / static_cast<float>(n);
    // 1/n, shrunk further if the content is wider than the 4-unit room so
    // that every copy stays inside its own cell.
    const float widest = std::max(hi.x - lo.x, std::max(hi.y - lo.y, hi.z - lo.z));
    const float scale = (widest > 4.f ? 4.f / widest : 1.f) / static_cast<float>(n);
    const size_t cells = static_cast<size_t>(n) * n * n;
    s.m_vertices.reserve(nverts * (cells + 1));
    s.m_indices.reserve(room + content * cells);
    for (uint32_t ix = 0; ix < n; ++ix)
        for (uint32_t iy = 0; iy < n; ++iy)
            for (uint32_t iz = 0; iz < n; ++iz) {
                Vec3 offset = make3(-2.f + (ix + 0.5f) * cell, -2.f + (iy + 0.5f) * cell, -2.f + (iz + 0.5f) * cell);
                const size_t vertexBase = s.m_vertices.size();
                for (size_t v = 0; v < nverts; ++v) {
                    Vec3 p = base.m_vertices[v];
                    s.m_vertices.push_back(make3((p.x - center.x) * scale + offset.x, (p.y - center.y) * scale + offset.y,
                                                 (p.z - center.z) * scale + offset.z));
                }
                for (size_t t = room; t < total; ++t) emit(t, vertexBase);
            }
    return s;
}

}  // namespace pt

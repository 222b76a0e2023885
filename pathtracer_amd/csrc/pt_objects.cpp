// C ABI of the host-side objects (include/pathtracer_amd.h): scenes, lights, cameras, and the tree inspection that
// needs no GPU.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "bvh.h"
#include "renderer_state.h"

using namespace pt_internal;

extern "C" {

/* ------------------------------------------------------------------ Scene */

int pt_scene_load_obj(const char *obj_path, const char *mtl_root, pt_scene **out)
{
    PT_REQUIRE(obj_path && out, "pt_scene_load_obj: null argument");
    pt_scene *s = new (std::nothrow) pt_scene{pt::Scene(obj_path, mtl_root)};
    if (!s) return fail(PT_ERR_INVALID, "out of memory");
    if (!s->scene.m_error.empty() && s->scene.m_vertices.empty()) {
        std::string msg = s->scene.m_error;
        delete s;
        return fail(PT_ERR_IO, msg);
    }
    *out = s;
    return PT_OK;
}

int pt_scene_create_default(pt_scene **out)
{
    PT_REQUIRE(out, "pt_scene_create_default: null argument");
    *out = new pt_scene{pt::Scene()};
    return PT_OK;
}

int pt_scene_from_desc(const pt_scene_desc *d, pt_scene **out)
{
    PT_REQUIRE(d && out, "pt_scene_from_desc: null argument");
    pt_scene *s = new pt_scene{pt::Scene()};
    pt::Scene &sc = s->scene;
    sc.m_vertices.assign(reinterpret_cast<const pt::Vec3 *>(d->vertices),
                         reinterpret_cast<const pt::Vec3 *>(d->vertices) + d->n_vertices);
    sc.m_indices.assign(reinterpret_cast<const pt::Vec4 *>(d->indices),
                        reinterpret_cast<const pt::Vec4 *>(d->indices) + d->n_triangles);
    sc.m_vertexData.assign(reinterpret_cast<const pt::Scene::VertexData *>(d->vertex_data),
                           reinterpret_cast<const pt::Scene::VertexData *>(d->vertex_data) + 3 * static_cast<size_t>(d->n_triangles));
    sc.m_lights.assign(reinterpret_cast<const pt::Scene::Light *>(d->lights),
                       reinterpret_cast<const pt::Scene::Light *>(d->lights) + d->n_lights);
    sc.m_lightCount[0] = d->n_lights;
    sc.m_materials.assign(reinterpret_cast<const pt::Scene::Material *>(d->materials),
                          reinterpret_cast<const pt::Scene::Material *>(d->materials) + d->n_materials);
    sc.m_materialMap.assign(d->material_map, d->material_map + d->n_triangles);
    *out = s;
    return PT_OK;
}

void pt_scene_destroy(pt_scene *scene) { delete scene; }

int pt_scene_get_desc(const pt_scene *scene, pt_scene_desc *d)
{
    PT_REQUIRE(scene && d, "pt_scene_get_desc: null argument");
    const pt::Scene &sc = scene->scene;
    d->vertices = sc.m_vertices.empty() ? nullptr : &sc.m_vertices[0].x;
    d->n_vertices = static_cast<uint32_t>(sc.m_vertices.size());
    d->indices = sc.m_indices.empty() ? nullptr : &sc.m_indices[0].x;
    d->n_triangles = static_cast<uint32_t>(sc.m_indices.size());
    d->vertex_data = sc.m_vertexData.empty() ? nullptr : &sc.m_vertexData[0].normal.x;
    d->lights = sc.m_lights.empty() ? nullptr : sc.m_lights[0].radiance;
    d->n_lights = std::min<uint32_t>(sc.m_lightCount[0], static_cast<uint32_t>(sc.m_lights.size()));
    d->materials = sc.m_materials.empty() ? nullptr : &sc.m_materials[0].albedo.x;
    d->n_materials = static_cast<uint32_t>(sc.m_materials.size());
    d->material_map = sc.m_materialMap.empty() ? nullptr : sc.m_materialMap.data();
    return PT_OK;
}

int pt_light_make(const float radiance[3], const float position[3], const float rotation[3], const float scale[3],
                  float out_light[20])
{
    PT_REQUIRE(radiance && position && rotation && scale && out_light, "pt_light_make: null argument");
    pt::Scene::Light l(pt::make3(radiance[0], radiance[1], radiance[2]), pt::make3(position[0], position[1], position[2]),
                       pt::make3(rotation[0], rotation[1], rotation[2]), pt::make3(scale[0], scale[1], scale[2]));
    std::memcpy(out_light, &l, sizeof(l));
    return PT_OK;
}

int pt_scene_set_lights(pt_scene *scene, const float *lights, uint32_t n)
{
    PT_REQUIRE(scene && (lights || n == 0), "pt_scene_set_lights: null argument");
    scene->scene.m_lights.assign(reinterpret_cast<const pt::Scene::Light *>(lights),
                                 reinterpret_cast<const pt::Scene::Light *>(lights) + n);
    scene->scene.m_lightCount[0] = n;
    return PT_OK;
}

int pt_scene_set_default_light(pt_scene *scene)
{
    PT_REQUIRE(scene, "pt_scene_set_default_light: null argument");
    scene->scene.setDefaultLight();
    return PT_OK;
}

int pt_scene_instance_lattice(const pt_scene *base, uint32_t room_triangles, uint32_t n, pt_scene **out)
{
    PT_REQUIRE(base && out && n >= 1, "pt_scene_instance_lattice: bad argument");
    const size_t nv = base->scene.m_vertices.size() * (static_cast<size_t>(n) * n * n + 1);
    PT_REQUIRE(nv < (size_t(1) << 24), "pt_scene_instance_lattice: more than 2^24 vertices");
    *out = new pt_scene{pt::Scene::instanceLattice(base->scene, room_triangles, n)};
    return PT_OK;
}

/* ----------------------------------------------------------------- Camera */

int pt_camera_create(const float eye[3], const float focus[3], uint32_t width, uint32_t height, pt_camera **out)
{
    PT_REQUIRE(eye && focus && out, "pt_camera_create: null argument");
    *out = new pt_camera(pt::Camera(pt::make3(eye[0], eye[1], eye[2]), pt::make3(focus[0], focus[1], focus[2]), width, height));
    return PT_OK;
}
void pt_camera_destroy(pt_camera *cam) { delete cam; }
int pt_camera_look_at(pt_camera *cam, const float focus[3], const float *eye)
{
    PT_REQUIRE(cam && focus, "pt_camera_look_at: null argument");
    if (eye)
        cam->cam.lookAt(pt::make3(focus[0], focus[1], focus[2]), pt::make3(eye[0], eye[1], eye[2]));
    else
        cam->cam.lookAt(pt::make3(focus[0], focus[1], focus[2]));
    return PT_OK;
}
int pt_camera_move(pt_camera *cam, const float o[3])
{
    PT_REQUIRE(cam && o, "pt_camera_move: null argument");
    cam->cam.move(pt::make3(o[0], o[1], o[2]));
    return PT_OK;
}
int pt_camera_pan(pt_camera *cam, float dx, float dy)
{
    PT_REQUIRE(cam, "pt_camera_pan: null argument");
    cam->cam.pan(dx, dy);
    return PT_OK;
}
int pt_camera_orbit(pt_camera *cam, float dx, float dy)
{
    PT_REQUIRE(cam, "pt_camera_orbit: null argument");
    cam->cam.orbit(dx, dy);
    return PT_OK;
}
int pt_camera_zoom(pt_camera *cam, float scale)
{
    PT_REQUIRE(cam, "pt_camera_zoom: null argument");
    cam->cam.zoom(scale);
    return PT_OK;
}
int pt_camera_update(pt_camera *cam)
{
    PT_REQUIRE(cam, "pt_camera_update: null argument");
    cam->cam.update();
    return PT_OK;
}
int pt_camera_get(const pt_camera *cam, float out[15], uint32_t res[2])
{
    PT_REQUIRE(cam && out, "pt_camera_get: null argument");
    const pt::Vec3 v[5] = {cam->cam.m_eye, cam->cam.m_focus, cam->cam.m_forward, cam->cam.m_up, cam->cam.m_right};
    for (int i = 0; i < 5; ++i)
        for (int k = 0; k < 3; ++k) out[3 * i + k] = v[i][k];
    if (res) {
        res[0] = cam->cam.m_resolution[0];
        res[1] = cam->cam.m_resolution[1];
    }
    return PT_OK;
}

/* ------------------------------------------------------------ BVH inspection */

int pt_bvh_inspect(const pt_scene_desc *sc, pt_bvh_info *info)
{
    PT_REQUIRE(sc && info, "pt_bvh_inspect: null argument");
    const uint32_t T = sc->n_triangles;
    PT_REQUIRE(T == 0 || (sc->vertices && sc->indices), "pt_bvh_inspect: scene arrays missing");
    std::vector<uint32_t> ids(static_cast<size_t>(T) * 3);
    for (uint32_t t = 0; t < T; ++t)
        for (int k = 0; k < 3; ++k) {
            const float f = sc->indices[4 * static_cast<size_t>(t) + k];
            PT_REQUIRE(f >= 0.f && f < static_cast<float>(sc->n_vertices), "pt_bvh_inspect: vertex index out of range");
            ids[3 * static_cast<size_t>(t) + k] = static_cast<uint32_t>(f);
        }
    const double t0 = now_ms();
    pt::Bvh bvh;
    if (T < (1u << 16))  // as pt_create does with the default (automatic) builder choice
        pt::build_bvh_best_of(sc->vertices, ids.data(), T, 64, &bvh);
    else
        pt::build_bvh(sc->vertices, ids.data(), T, 64, &bvh);
    std::memset(info, 0, sizeof(*info));
    info->build_ms = now_ms() - t0;
    info->n_triangles = T;
    info->n_nodes = static_cast<uint32_t>(bvh.nodes.size());
    info->depth = bvh.depth;
    info->stack_entries = std::max<uint32_t>(4, bvh.depth + 1u);
    info->sah_cost = bvh.sah_cost;

    // Walk the tree: boxes must enclose everything below them; leaves must
    // partition the triangle array.
    struct Item {
        int32_t code;
        float lo[3], hi[3];
        uint32_t level;
    };
    std::vector<Item> todo;
    std::vector<uint8_t> seen(T, 0);
    Item root{0, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}, 1};
    todo.push_back(root);
    uint32_t depth = 0;
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        if (it.code >= 0) {
            if (static_cast<size_t>(it.code) >= bvh.nodes.size()) return fail(PT_ERR_STATE, "BVH child index out of range");
            depth = std::max(depth, it.level);
            const pt::BvhNode &n = bvh.nodes[it.code];
            for (int c = 0; c < 2; ++c) {
                Item ch;
                ch.code = n.child[c];
                ch.level = it.level + 1;
                for (int a = 0; a < 3; ++a) {
                    ch.lo[a] = n.q[6 * c + a];
                    ch.hi[a] = n.q[6 * c + 3 + a];
                }
                const bool empty_leaf = ch.code < 0 && ((~static_cast<uint32_t>(ch.code)) & 7u) == 0;
                if (!empty_leaf)
                    for (int a = 0; a < 3; ++a)
                        if (ch.lo[a] < it.lo[a] || ch.hi[a] > it.hi[a]) return fail(PT_ERR_STATE, "BVH child box leaves its parent");
                todo.push_back(ch);
            }
        } else {
            const uint32_t code = ~static_cast<uint32_t>(it.code);
            const uint32_t first = code >> 3, cnt = code & 7u;
            if (cnt == 0) continue;
            info->n_leaves++;
            info->max_leaf_size = std::max(info->max_leaf_size, cnt);
            if (static_cast<size_t>(first) + cnt > bvh.triangles.size()) return fail(PT_ERR_STATE, "BVH leaf range out of bounds");
            for (uint32_t k = first; k < first + cnt; ++k) {
                const pt::BvhTriangle &tri = bvh.triangles[k];
                if (tri.original >= T || seen[tri.original]) return fail(PT_ERR_STATE, "triangle referenced twice or out of range");
                seen[tri.original] = 1;
                const float *vs[3] = {tri.v0, tri.v1, tri.v2};
                for (const float *v : vs)
                    for (int a = 0; a < 3; ++a)
                        if (v[a] < it.lo[a] || v[a] > it.hi[a]) return fail(PT_ERR_STATE, "triangle vertex outside its leaf box");
                for (int k2 = 0; k2 < 3; ++k2)
                    for (int a = 0; a < 3; ++a)
                        if (vs[k2][a] != sc->vertices[3 * static_cast<size_t>(ids[3 * static_cast<size_t>(tri.original) + k2]) + a])
                            return fail(PT_ERR_STATE, "triangle record does not match the scene");
            }
        }
    }
    for (uint32_t t = 0; t < T; ++t)
        if (!seen[t]) return fail(PT_ERR_STATE, "triangle " + std::to_string(t) + " is in no leaf");
    if (depth != bvh.depth) return fail(PT_ERR_STATE, "reported BVH depth does not match the tree");
    return PT_OK;
}

}  // extern "C"

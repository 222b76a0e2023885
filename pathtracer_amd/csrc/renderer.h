// C++ convenience layer over the C ABI with the reference's Renderer surface
// (renderer.h:12-39): construct from a Scene and a Camera, then draw(),
// reset(), resize(), updateCamera().  Header-only; links against
// libpathtracer_amd.so.  Failures throw std::runtime_error here (the C ABI
// itself never throws).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pathtracer_amd.h"
#include "camera.h"
#include "scene.h"

namespace pt {

class Renderer {
public:
    Scene *m_scene;
    Camera *m_camera;

    Renderer(Scene *scene, Camera *camera, uint32_t maxDepth = 10, float fovYDegrees = 19.5f, bool useBvh = true,
             int device = -1, bool buildBvhOnDevice = false)
        : m_scene(scene), m_camera(camera)
    {
        pt_scene_desc d;
        d.vertices = scene->m_vertices.empty() ? nullptr : &scene->m_vertices[0].x;
        d.n_vertices = static_cast<uint32_t>(scene->m_vertices.size());
        d.indices = scene->m_indices.empty() ? nullptr : &scene->m_indices[0].x;
        d.n_triangles = static_cast<uint32_t>(scene->m_indices.size());
        d.vertex_data = scene->m_vertexData.empty() ? nullptr : &scene->m_vertexData[0].normal.x;
        d.lights = scene->m_lights.empty() ? nullptr : scene->m_lights[0].radiance;
        d.n_lights = scene->m_lightCount[0];
        d.materials = scene->m_materials.empty() ? nullptr : &scene->m_materials[0].albedo.x;
        d.n_materials = static_cast<uint32_t>(scene->m_materials.size());
        d.material_map = scene->m_materialMap.empty() ? nullptr : scene->m_materialMap.data();
        pt_config cfg;
        pt_config_default(&cfg);
        cfg.width = camera->m_resolution[0];
        cfg.height = camera->m_resolution[1];
        cfg.max_depth = maxDepth;
        cfg.fov_y_degrees = fovYDegrees;
        cfg.use_bvh = useBvh ? 1 : 0;
        cfg.bvh_builder = buildBvhOnDevice ? 1 : 0;
        cfg.device = device;
        check(pt_create(&d, &cfg, &m_handle));
        updateCamera();
    }
    ~Renderer() { pt_destroy(m_handle); }
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    // One more sample per pixel (renderer.cpp:124-150); n > 1 batches several.
    void draw(uint32_t n = 1) { check(pt_render(m_handle, n)); }
    void reset() { check(pt_reset(m_handle)); }
    void resize(uint32_t width, uint32_t height)
    {
        m_camera->m_resolution[0] = width;
        m_camera->m_resolution[1] = height;
        check(pt_resize(m_handle, width, height));
    }
    void updateCamera()
    {
        m_camera->update();
        check(pt_set_camera(m_handle, &m_camera->m_eye.x, &m_camera->m_forward.x, &m_camera->m_up.x, &m_camera->m_right.x));
    }

    // Checkpoint / resume: the sum buffer and the iteration counter (the reference's render state,
    // renderer.h:21-29), refused on load if scene, camera, depth, size or partition differ.
    void saveState(const std::string &path) { check(pt_save_state(m_handle, path.c_str())); }
    void loadState(const std::string &path) { check(pt_load_state(m_handle, path.c_str())); }

    void synchronize() { check(pt_synchronize(m_handle)); }
    std::vector<float> radiance()
    {
        std::vector<float> out(4 * static_cast<size_t>(m_camera->m_resolution[0]) * m_camera->m_resolution[1]);
        check(pt_read_radiance(m_handle, out.data()));
        return out;
    }
    std::vector<uint8_t> tonemapped()
    {
        std::vector<uint8_t> out(3 * static_cast<size_t>(m_camera->m_resolution[0]) * m_camera->m_resolution[1]);
        check(pt_read_tonemapped(m_handle, out.data()));
        return out;
    }
    pt_stats stats()
    {
        pt_stats s;
        check(pt_get_stats(m_handle, &s));
        return s;
    }
    pt_renderer *handle() { return m_handle; }

private:
    pt_renderer *m_handle = nullptr;
    static void check(int rc)
    {
        if (rc != PT_OK) throw std::runtime_error(std::string("pathtracer_amd: ") + pt_last_error());
    }
};

// The same surface over all GPUs of one process: the scene is replicated, GPU g of G renders iterations g,
// g + G, ... of the global sequence 0, 1, 2, ... (pt_set_partition), and reduce() combines the per-GPU sums
// on GPU 0 with one RCCL reduce.  One host thread drives all GPUs: draw() only enqueues.
class MultiRenderer {
public:
    MultiRenderer(Scene *scene, Camera *camera, uint32_t gpus, uint32_t maxDepth = 10, float fovYDegrees = 19.5f,
                  bool useBvh = true, bool buildBvhOnDevice = false)
    {
        std::vector<int32_t> devices(gpus);
        for (uint32_t g = 0; g < gpus; ++g) {
            devices[g] = static_cast<int32_t>(g);
            m_renderers.emplace_back(new Renderer(scene, camera, maxDepth, fovYDegrees, useBvh, static_cast<int>(g), buildBvhOnDevice));
            check(pt_set_partition(m_renderers.back()->handle(), g, gpus));
        }
        m_comms.assign(gpus, nullptr);
        check(pt_comm_create_all(devices.data(), gpus, m_comms.data()));
    }
    ~MultiRenderer()
    {
        for (pt_comm *c : m_comms) pt_comm_destroy(c);
        for (Renderer *r : m_renderers) delete r;
    }
    MultiRenderer(const MultiRenderer &) = delete;
    MultiRenderer &operator=(const MultiRenderer &) = delete;

    // n more iterations of the global sequence, each on the GPU it belongs to.
    void draw(uint32_t n = 1)
    {
        const uint64_t G = m_renderers.size(), after = m_done + n;
        for (uint64_t g = 0; g < G; ++g) {
            const uint64_t had = m_done > g ? (m_done - g + G - 1) / G : 0, want = after > g ? (after - g + G - 1) / G : 0;
            if (want > had) m_renderers[g]->draw(static_cast<uint32_t>(want - had));
        }
        m_done = after;
    }
    // Sum of all GPUs' samples into GPU 0's renderer (asynchronous; the readbacks of root() wait for it).
    void reduce()
    {
        std::vector<pt_renderer *> rs;
        for (Renderer *r : m_renderers) rs.push_back(r->handle());
        check(pt_reduce_sum(rs.data(), m_comms.data(), static_cast<uint32_t>(rs.size()), 0));
    }
    void synchronize()
    {
        for (Renderer *r : m_renderers) r->synchronize();
    }
    Renderer &root() { return *m_renderers[0]; }
    Renderer &renderer(size_t g) { return *m_renderers[g]; }
    size_t gpus() const { return m_renderers.size(); }
    uint64_t iterations() const { return m_done; }

private:
    std::vector<Renderer *> m_renderers;
    std::vector<pt_comm *> m_comms;
    uint64_t m_done = 0;
    static void check(int rc)
    {
        if (rc != PT_OK) throw std::runtime_error(std::string("pathtracer_amd: ") + pt_last_error());
    }
};

}  // namespace pt

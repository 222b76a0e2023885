// ref_tinyobj_dump -- runs the reference's own vendored tiny_obj_loader.h
// (compiled from /root/reference/src where it lies; nothing is copied) on an
// OBJ file and prints what Scene's constructor consumes (scene.h:81-136):
// attrib vertices/normals, per-shape face indices and material ids, and the
// material_t fields the material rules read (scene.h:139-217).
//
// TEST INFRASTRUCTURE ONLY.  Output format (text, one record per line):
//   V n   followed by n lines "x y z" as hex float bits
//   N n   idem for normals
//   S nshapes ; per shape: "shape nfaces" then nfaces lines "v0 n0 v1 n1 v2 n2 mat"
//   M nmaterials ; per material one line of hex-bit fields:
//     Kd[3] Ka[3] Tf[3] Ns Ni illum isPBR Pr Pm aniso
#define TINYOBJLOADER_IMPLEMENTATION
#include <tiny_obj_loader.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static unsigned bits(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s file.obj mtl_dir/\n", argv[0]);
        return 2;
    }
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &err, argv[1], argv[2], true)) {
        std::fprintf(stderr, "LoadObj failed: %s\n", err.c_str());
        return 1;
    }
    std::printf("V %zu\n", attrib.vertices.size() / 3);
    for (size_t i = 0; i + 2 < attrib.vertices.size(); i += 3)
        std::printf("%08x %08x %08x\n", bits(attrib.vertices[i]), bits(attrib.vertices[i + 1]),
                    bits(attrib.vertices[i + 2]));
    std::printf("N %zu\n", attrib.normals.size() / 3);
    for (size_t i = 0; i + 2 < attrib.normals.size(); i += 3)
        std::printf("%08x %08x %08x\n", bits(attrib.normals[i]), bits(attrib.normals[i + 1]),
                    bits(attrib.normals[i + 2]));
    std::printf("S %zu\n", shapes.size());
    for (const auto &s : shapes) {
        size_t nf = s.mesh.num_face_vertices.size();
        std::printf("shape %zu\n", nf);
        for (size_t f = 0; f < nf; ++f) {
            const tinyobj::index_t &a = s.mesh.indices[3 * f], &b = s.mesh.indices[3 * f + 1],
                                   &c = s.mesh.indices[3 * f + 2];
            std::printf("%d %d %d %d %d %d %d\n", a.vertex_index, a.normal_index, b.vertex_index,
                        b.normal_index, c.vertex_index, c.normal_index, s.mesh.material_ids[f]);
        }
    }
    std::printf("M %zu\n", materials.size());
    for (const auto &m : materials)
        std::printf("%08x %08x %08x %08x %08x %08x %08x %08x %08x %08x %08x %d %d %08x %08x %08x\n",
                    bits(m.diffuse[0]), bits(m.diffuse[1]), bits(m.diffuse[2]), bits(m.ambient[0]),
                    bits(m.ambient[1]), bits(m.ambient[2]), bits(m.transmittance[0]),
                    bits(m.transmittance[1]), bits(m.transmittance[2]), bits(m.shininess), bits(m.ior),
                    m.illum, m.isPBR ? 1 : 0, bits(m.roughness), bits(m.metallic), bits(m.anisotropy));
    return 0;
}

// OBJ/MTL reader producing exactly what tiny_obj_loader v1.0.6's
// LoadObj(..., triangulate=true) yields for the fields Scene consumes
// (tiny_obj_loader.h:1409-1744, LoadMtl :967-1347): attribute arrays, one
// shape per o/g block with fan-triangulated faces and per-face material ids,
// and per-material Kd/Ka/Tf/Ns/Ni/illum/Pr/Pm/aniso with the isPBR flag.
#pragma once
#include <string>
#include <vector>

namespace pt {

struct ObjIndex {
    int vertex_index;
    int normal_index;
    int texcoord_index;
};

struct ObjShape {
    std::string name;
    std::vector<ObjIndex> indices;   // 3 per face
    std::vector<int> material_ids;   // 1 per face, -1 = none
};

struct ObjMaterial {
    std::string name;
    float ambient[3];
    float diffuse[3];
    float specular[3];
    float transmittance[3];
    float emission[3];
    float shininess;
    float ior;
    float dissolve;
    int illum;
    bool isPBR;
    float roughness;
    float metallic;
    float sheen;
    float clearcoat_thickness;
    float clearcoat_roughness;
    float anisotropy;
    float anisotropy_rotation;
};

struct ObjData {
    std::vector<float> vertices;   // xyz
    std::vector<float> normals;    // xyz
    std::vector<float> texcoords;  // uv
    std::vector<ObjShape> shapes;
    std::vector<ObjMaterial> materials;
};

// Returns false (and fills err) only when the OBJ file cannot be opened, like
// tinyobj::LoadObj; a missing MTL file is a warning appended to err.
bool load_obj(const char *obj_path, const char *mtl_root, ObjData *out, std::string *err);

// Decimal text -> double with tiny_obj_loader's own digit-accumulation
// procedure (tiny_obj_loader.h:475-588), so that parsed floats carry the same
// bits.  Returns false when the text is not a number in its grammar.
bool parse_decimal(const char *s, const char *s_end, double *result);

}  // namespace pt

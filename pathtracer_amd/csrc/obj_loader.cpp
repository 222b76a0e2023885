// OBJ/MTL reader with tiny_obj_loader v1.0.6 semantics (see obj_loader.h).
#include "obj_loader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace pt {

namespace {

inline bool is_blank(char c) { return c == ' ' || c == '\t'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_eol(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// Reads the whole file and splits it into lines on \n, \r\n or a lone \r
// (tiny_obj_loader.h:390-418).
bool read_lines(const std::string &path, std::vector<std::string> *lines)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) return false;
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    std::string cur;
    for (size_t i = 0; i < text.size(); ++i) {
        char c = text[i];
        if (c == '\n') {
            lines->push_back(cur);
            cur.clear();
        } else if (c == '\r') {
            if (i + 1 < text.size() && text[i + 1] == '\n') ++i;
            lines->push_back(cur);
            cur.clear();
        } else {
            cur.push_back(c);
        }
    }
    if (!cur.empty()) lines->push_back(cur);
    return true;
}

const char *skip_blank(const char *p) { return p + std::strspn(p, " \t"); }

float read_real(const char **tok, double dflt = 0.0)
{
    const char *p = skip_blank(*tok);
    const char *end = p + std::strcspn(p, " \t\r");
    double v = dflt;
    parse_decimal(p, end, &v);
    *tok = end;
    return static_cast<float>(v);
}

int read_int(const char **tok)
{
    const char *p = skip_blank(*tok);
    int v = std::atoi(p);
    *tok = p + std::strcspn(p, " \t\r");
    return v;
}

// First whitespace-delimited word, as sscanf("%s") would return it.
std::string first_word(const char *p)
{
    while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r') ++p;
    const char *e = p;
    while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\v' && *e != '\f' && *e != '\r') ++e;
    return std::string(p, e);
}

// OBJ indices are 1-based; negative ones count back from the current array end
// (tiny_obj_loader.h:426-430).
int resolve_index(int idx, int n)
{
    if (idx > 0) return idx - 1;
    if (idx == 0) return 0;
    return n + idx;
}

// "i", "i/j", "i//k", "i/j/k" (tiny_obj_loader.h:692-723)
ObjIndex read_face_corner(const char **tok, int nv, int nn, int nt)
{
    ObjIndex r{-1, -1, -1};
    const char *p = *tok;
    r.vertex_index = resolve_index(std::atoi(p), nv);
    p += std::strcspn(p, "/ \t\r");
    if (*p == '/') {
        ++p;
        if (*p == '/') {
            ++p;
            r.normal_index = resolve_index(std::atoi(p), nn);
            p += std::strcspn(p, "/ \t\r");
        } else {
            r.texcoord_index = resolve_index(std::atoi(p), nt);
            p += std::strcspn(p, "/ \t\r");
            if (*p == '/') {
                ++p;
                r.normal_index = resolve_index(std::atoi(p), nn);
                p += std::strcspn(p, "/ \t\r");
            }
        }
    }
    *tok = p;
    return r;
}

void reset_material(ObjMaterial *m)
{
    m->name.clear();
    for (int i = 0; i < 3; ++i) {
        m->ambient[i] = m->diffuse[i] = m->specular[i] = 0.f;
        m->transmittance[i] = m->emission[i] = 0.f;
    }
    m->shininess = 1.f;
    m->ior = 1.f;
    m->dissolve = 1.f;
    m->illum = 0;
    m->isPBR = false;
    m->roughness = m->metallic = m->sheen = 0.f;
    m->clearcoat_thickness = m->clearcoat_roughness = 0.f;
    m->anisotropy = m->anisotropy_rotation = 0.f;
}

bool key2(const char *t, char a, char b) { return t[0] == a && t[1] == b && is_blank(t[2]); }
bool keyword(const char *t, const char *kw)
{
    size_t n = std::strlen(kw);
    return std::strncmp(t, kw, n) == 0 && is_blank(t[n]);
}

void read_rgb(const char *t, float out[3])
{
    out[0] = read_real(&t);
    out[1] = read_real(&t);
    out[2] = read_real(&t);
}

// MTL statements in tiny_obj_loader's test order (tiny_obj_loader.h:1013-1337).
void parse_mtl(const std::vector<std::string> &lines, std::map<std::string, int> *index_of,
               std::vector<ObjMaterial> *materials)
{
    ObjMaterial cur;
    reset_material(&cur);
    bool has_d = false;
    for (std::string line : lines) {
        size_t last = line.find_last_not_of(" \t");
        line = (last == std::string::npos) ? std::string() : line.substr(0, last + 1);
        if (line.empty()) continue;
        const char *t = skip_blank(line.c_str());
        if (*t == '\0' || *t == '#') continue;

        if (keyword(t, "newmtl")) {
            if (!cur.name.empty()) {
                index_of->insert(std::make_pair(cur.name, static_cast<int>(materials->size())));
                materials->push_back(cur);
            }
            reset_material(&cur);
            has_d = false;
            cur.name = first_word(t + 7);
            continue;
        }
        if (key2(t, 'K', 'a')) { read_rgb(t + 2, cur.ambient); continue; }
        if (key2(t, 'K', 'd')) { read_rgb(t + 2, cur.diffuse); continue; }
        if (key2(t, 'K', 's')) { read_rgb(t + 2, cur.specular); continue; }
        if (key2(t, 'K', 't') || key2(t, 'T', 'f')) { read_rgb(t + 2, cur.transmittance); continue; }
        if (key2(t, 'N', 'i')) { t += 2; cur.ior = read_real(&t); continue; }
        if (key2(t, 'K', 'e')) { read_rgb(t + 2, cur.emission); continue; }
        if (key2(t, 'N', 's')) { t += 2; cur.shininess = read_real(&t); continue; }
        if (keyword(t, "illum")) { t += 6; cur.illum = read_int(&t); continue; }
        if (t[0] == 'd' && is_blank(t[1])) { t += 1; cur.dissolve = read_real(&t); has_d = true; continue; }
        if (key2(t, 'T', 'r')) {
            t += 2;
            if (!has_d) cur.dissolve = 1.0f - read_real(&t);
            continue;
        }
        if (key2(t, 'P', 'r')) { t += 2; cur.isPBR = true; cur.roughness = read_real(&t); continue; }
        if (key2(t, 'P', 'm')) { t += 2; cur.isPBR = true; cur.metallic = read_real(&t); continue; }
        if (key2(t, 'P', 's')) { t += 2; cur.isPBR = true; cur.sheen = read_real(&t); continue; }
        if (key2(t, 'P', 'c')) { t += 2; cur.isPBR = true; cur.clearcoat_thickness = read_real(&t); continue; }
        if (keyword(t, "Pcr")) { t += 4; cur.isPBR = true; cur.clearcoat_roughness = read_real(&t); continue; }
        if (keyword(t, "aniso")) { t += 6; cur.isPBR = true; cur.anisotropy = read_real(&t); continue; }
        if (keyword(t, "anisor")) { t += 7; cur.isPBR = true; cur.anisotropy_rotation = read_real(&t); continue; }
        // Texture statements carry no value the material model reads, but the
        // PBR ones still switch the material to the PBR rules.
        if (keyword(t, "map_Pr") || keyword(t, "map_Pm") || keyword(t, "map_Ps") ||
            keyword(t, "map_Ke") || keyword(t, "norm")) {
            cur.isPBR = true;
            continue;
        }
    }
    index_of->insert(std::make_pair(cur.name, static_cast<int>(materials->size())));
    materials->push_back(cur);
}

typedef std::vector<ObjIndex> Face;

// Appends the pending faces to the shape as a triangle fan per face
// (tiny_obj_loader.h:892-953).  Returns false when nothing was pending.
bool flush_faces(ObjShape *shape, const std::vector<Face> &pending, int material, const std::string &name)
{
    if (pending.empty()) return false;
    for (const Face &face : pending) {
        for (size_t k = 2; k < face.size(); ++k) {
            shape->indices.push_back(face[0]);
            shape->indices.push_back(face[k - 1]);
            shape->indices.push_back(face[k]);
            shape->material_ids.push_back(material);
        }
    }
    shape->name = name;
    return true;
}

}  // namespace

bool parse_decimal(const char *s, const char *s_end, double *result)
{
    if (s >= s_end) return false;
    static const double frac_lut[8] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
    const char *p = s;
    bool negative = false;
    if (*p == '+' || *p == '-') {
        negative = (*p == '-');
        ++p;
    } else if (!is_digit(*p)) {
        return false;
    }

    double mantissa = 0.0;
    int ndigits = 0;
    while (p != s_end && is_digit(*p)) {
        mantissa *= 10;
        mantissa += static_cast<int>(*p - '0');
        ++p;
        ++ndigits;
    }
    if (ndigits == 0) return false;

    int exponent = 0;
    bool want_exponent = false;
    if (p != s_end) {
        if (*p == '.') {
            ++p;
            int pos = 1;
            while (p != s_end && is_digit(*p)) {
                mantissa += static_cast<int>(*p - '0') * (pos < 8 ? frac_lut[pos] : std::pow(10.0, -pos));
                ++pos;
                ++p;
            }
            want_exponent = (p != s_end) && (*p == 'e' || *p == 'E');
        } else if (*p == 'e' || *p == 'E') {
            want_exponent = true;
        }
    }
    if (want_exponent) {
        ++p;
        bool exp_negative = false;
        if (p != s_end && (*p == '+' || *p == '-')) {
            exp_negative = (*p == '-');
            ++p;
        } else if (!is_digit(*p)) {
            return false;
        }
        int nexp = 0;
        while (p != s_end && is_digit(*p)) {
            exponent = exponent * 10 + static_cast<int>(*p - '0');
            ++p;
            ++nexp;
        }
        if (exp_negative) exponent = -exponent;
        if (nexp == 0) return false;
    }
    double magnitude = exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa;
    *result = negative ? -magnitude : magnitude;
    return true;
}

bool load_obj(const char *obj_path, const char *mtl_root, ObjData *out, std::string *err)
{
    *out = ObjData();
    std::vector<std::string> lines;
    if (!read_lines(obj_path, &lines)) {
        if (err) *err = std::string("Cannot open file [") + obj_path + "]\n";
        return false;
    }
    const std::string base = mtl_root ? mtl_root : "";

    std::map<std::string, int> material_index;
    int material = -1;
    std::string name;
    ObjShape shape;
    std::vector<Face> pending;

    for (const std::string &line : lines) {
        if (line.empty()) continue;
        const char *t = skip_blank(line.c_str());
        if (*t == '\0' || *t == '#') continue;

        if (t[0] == 'v' && is_blank(t[1])) {
            t += 2;
            for (int k = 0; k < 3; ++k) out->vertices.push_back(read_real(&t));
            continue;
        }
        if (t[0] == 'v' && t[1] == 'n' && is_blank(t[2])) {
            t += 3;
            for (int k = 0; k < 3; ++k) out->normals.push_back(read_real(&t));
            continue;
        }
        if (t[0] == 'v' && t[1] == 't' && is_blank(t[2])) {
            t += 3;
            for (int k = 0; k < 2; ++k) out->texcoords.push_back(read_real(&t));
            continue;
        }
        if (t[0] == 'f' && is_blank(t[1])) {
            t = skip_blank(t + 2);
            Face face;
            while (!is_eol(*t)) {
                face.push_back(read_face_corner(&t, static_cast<int>(out->vertices.size() / 3),
                                                static_cast<int>(out->normals.size() / 3),
                                                static_cast<int>(out->texcoords.size() / 2)));
                t += std::strspn(t, " \t\r");
            }
            pending.push_back(face);
            continue;
        }
        if (keyword(t, "usemtl")) {
            std::string mname = first_word(t + 7);
            int id = -1;
            std::map<std::string, int>::const_iterator it = material_index.find(mname);
            if (it != material_index.end()) id = it->second;
            if (id != material) {
                // Faces seen so far keep the previous material; the shape stays open.
                flush_faces(&shape, pending, material, name);
                pending.clear();
                material = id;
            }
            continue;
        }
        if (keyword(t, "mtllib")) {
            std::stringstream names(std::string(t + 7));
            std::string fname;
            bool found = false, any = false;
            while (!found && std::getline(names, fname, ' ')) {
                any = true;
                std::vector<std::string> mtl_lines;
                std::string path = base.empty() ? fname : base + fname;
                if (read_lines(path, &mtl_lines)) {
                    parse_mtl(mtl_lines, &material_index, &out->materials);
                    found = true;
                } else if (err) {
                    *err += "WARN: Material file [ " + path + " ] not found.\n";
                }
            }
            if (!found && err)
                *err += any ? "WARN: Failed to load material file(s). Use default material.\n"
                            : "WARN: Looks like empty filename for mtllib. Use default material. \n";
            continue;
        }
        if ((t[0] == 'g' || t[0] == 'o') && is_blank(t[1])) {
            // A new group/object closes the current shape; a shape whose last
            // statement was a material switch (nothing pending) is dropped, as
            // tiny_obj_loader does (tiny_obj_loader.h:1617-1621,1652-1656).
            if (flush_faces(&shape, pending, material, name)) out->shapes.push_back(shape);
            shape = ObjShape();
            pending.clear();
            if (t[0] == 'o') {
                name = first_word(t + 2);
            } else {
                // Words of the line including the leading "g"; the second one names the group.
                std::vector<std::string> words;
                const char *p = t;
                while (!is_eol(*p)) {
                    p = skip_blank(p);
                    size_t n = std::strcspn(p, " \t\r");
                    words.push_back(std::string(p, p + n));
                    p += n;
                    p += std::strspn(p, " \t\r");
                }
                name = words.size() > 1 ? words[1] : std::string();
            }
            continue;
        }
        // Everything else (s, t, l, ...) does not affect the arrays Scene reads.
    }
    bool flushed = flush_faces(&shape, pending, material, name);
    if (flushed || !shape.indices.empty()) out->shapes.push_back(shape);
    return true;
}

}  // namespace pt

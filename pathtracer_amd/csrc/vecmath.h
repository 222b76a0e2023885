// Minimal host-side f32 vector helpers.  The evaluation order of dot() and
// normalize() follows glm's (left-to-right sum, v * (1/sqrt(dot))), because the
// camera basis the integrator receives must carry the same bits the
// reference's glm-based Camera::update produces (camera.cpp:89-95,
// glm/detail/func_geometric.inl:48-53,82-88).
#pragma once
#include <cmath>

namespace pt {

struct Vec3 {
    float x, y, z;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
struct Vec4 {
    float x, y, z, w;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

inline Vec3 make3(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return Vec3{a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return Vec3{a.x / s, a.y / s, a.z / s}; }
inline Vec3 &operator+=(Vec3 &a, Vec3 b) { a = a + b; return a; }

// glm::dot for vec3: (x*x + y*y) + z*z
inline float dot_glm(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b)
{
    return Vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length_glm(Vec3 v) { return std::sqrt(dot_glm(v, v)); }
// glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
inline Vec3 normalize_glm(Vec3 v) { return v * (1.0f / std::sqrt(dot_glm(v, v))); }

}  // namespace pt

// Records shared by the host uploader and the gfx950 kernels.
#pragma once
#include <cstdint>

namespace pt {

// One light as the kernels read it (the reference keeps radiance + transform
// and inverts the transform per ray, pathtracer.frag.glsl:92-103; here the
// inverse is taken once on the host).
struct DeviceLight {
    float radiance[4];
    float inv[16];     // inverse transform, column-major
    float normal[4];   // transform * (0,0,1,0)  (frag:457)
};

// Per original triangle, what shading needs (frag:434-453): the three
// positions for the barycentrics, the three corner normals and the material.
//   s0 = (v0.xyz, n1.x)  s1 = (v1.xyz, n1.y)  s2 = (v2.xyz, n1.z)
//   s3 = (n2.xyz, material index bits)        s4 = (n3.xyz, 0)
struct ShadeTriangle {
    float s[20];
};
static_assert(sizeof(ShadeTriangle) == 80, "ShadeTriangle is five 16-byte lanes");

// x / d for 0 <= x < 2^31: magic = ceil(2^(31 + l) / d), shift = l - 1, l = ceil(log2 d) (Granlund-Montgomery);
// d = 1 is magic = 0 (the quotient is x itself).
struct FastDiv {
    uint32_t magic, shift;
};
inline FastDiv make_fast_div(uint32_t d)
{
    if (d <= 1u) return FastDiv{0u, 0u};
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned long long m = ((1ull << (31 + l)) + d - 1) / d;  // < 2^32 because d > 2^(l-1)
    return FastDiv{static_cast<uint32_t>(m), l - 1u};
}

struct DeviceCamera {
    float eye[3];
    float tan_half_fov;
    float forward[3];
    float aspect;
    float up[3];
    float width_f;
    float right[3];
    float height_f;
    uint32_t width, height;
    // Path slot -> pixel within one iteration (pixel_of_slot in kernels.hip).
    uint32_t map_mode;    // 0: slot = pixel; 1: strips of 64 consecutive pixels; 2: tiles of 8 x 8 pixels
    uint32_t map_units;   // strips or tiles that are permuted (the tail beyond them maps to itself)
    uint32_t map_stride;  // unit u of the queue is unit (u * map_stride) mod map_units of the image
    uint32_t tiles_x;     // tiles per image row (mode 2)
    // Division by three run-time constants without a divide: x / d = umulhi(x, magic) >> shift for x < 2^31.
    FastDiv div_pixels, div_tiles_x, div_width;
    FastDiv div_map_units;  // valid when (map_units - 1) * map_stride < 2^31; magic = 0xFFFFFFFF says it is not
    // Pixels (columns cull_x0..cull_x1, rows cull_y0..cull_y1, inclusive) outside of which no primary ray can meet
    // the scene's bounds or a light: the bounding rectangle of their projections, widened by a margin (host:
    // update_cull_window).  The whole image when the host cannot tell (geometry beside or behind the eye).
    uint32_t cull_x0, cull_x1, cull_y0, cull_y1;
};

}  // namespace pt

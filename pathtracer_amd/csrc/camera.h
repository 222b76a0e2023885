// Orbit camera with the reference's public surface (camera.h:9-38).
#pragma once
#include <cstdint>

#include "vecmath.h"

namespace pt {

class Camera {
public:
    Vec3 m_eye;
    Vec3 m_focus;

    Vec3 m_forward;
    Vec3 m_up;
    Vec3 m_right;

    uint32_t m_resolution[2];

    // Axes are not valid until update() is called, as in the reference.
    Camera(const Vec3 &eye, const Vec3 &focus, uint32_t width, uint32_t height);

    void lookAt(const Vec3 &focus, const Vec3 &eye);
    void lookAt(const Vec3 &focus);

    void move(const Vec3 &offset);
    // Moves eye and focus in the camera plane [horizontal, vertical].
    void pan(float dx, float dy);
    // Rotates the eye about the focus by radians [horizontal, vertical];
    // the polar angle is clamped to [10, 170] degrees.
    void orbit(float dx, float dy);
    // Multiplies the eye-focus distance by scale.
    void zoom(float scale);

    void update();
    // Column-major 3x3 [right | up | forward].
    void getAxes(float out[9]) const;
};

}  // namespace pt

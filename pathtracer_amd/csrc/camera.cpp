#include "camera.h"

#include <algorithm>
#include <cmath>

namespace pt {

namespace {
// camera.h:36-37: float constants evaluated in double, then narrowed.
const float kMinPolar = static_cast<float>(10.f * M_PI / 180.f);
const float kMaxPolar = static_cast<float>(170.f * M_PI / 180.f);
}  // namespace

Camera::Camera(const Vec3 &eye, const Vec3 &focus, uint32_t width, uint32_t height)
    : m_eye(eye), m_focus(focus), m_forward{0.f, 0.f, 0.f}, m_up{0.f, 0.f, 0.f}, m_right{0.f, 0.f, 0.f},
      m_resolution{width, height}
{
}

void Camera::lookAt(const Vec3 &focus, const Vec3 &eye)
{
    m_focus = focus;
    m_eye = eye;
}

void Camera::lookAt(const Vec3 &focus) { m_focus = focus; }

void Camera::move(const Vec3 &offset)
{
    m_eye += offset;
    m_focus += offset;
}

void Camera::pan(float dx, float dy) { move(m_right * dx + m_up * dy); }

// camera.cpp:38-76: spherical coordinates about the focus,
// x = cos(g) sin(t), y = cos(t), z = sin(g) sin(t).
void Camera::orbit(float dx, float dy)
{
    Vec3 arm = m_eye - m_focus;
    float radius = length_glm(arm);
    Vec3 dir = arm / radius;

    float polar = std::acos(dir.y);
    float azimuth;
    if (std::sin(polar) != 0.f)
        azimuth = std::acos(dir.x / std::sin(polar));
    else
        azimuth = std::atan(dir.z / dir.x);
    if (dir.z < 0.f) azimuth = static_cast<float>(M_PI * 2.f - azimuth);

    polar += dy;
    azimuth += dx;
    polar = std::max(kMinPolar, std::min(polar, kMaxPolar));

    dir.y = std::cos(polar);
    dir.x = std::cos(azimuth) * std::sin(polar);
    dir.z = std::sin(azimuth) * std::sin(polar);
    m_eye = m_focus + dir * radius;
}

void Camera::zoom(float scale)
{
    Vec3 arm = m_eye - m_focus;
    float radius = length_glm(arm);
    Vec3 dir = arm / radius;
    m_eye = m_focus + dir * radius * scale;
}

// camera.cpp:89-95
void Camera::update()
{
    m_forward = normalize_glm(m_focus - m_eye);
    m_right = normalize_glm(make3(-m_forward.z, 0.f, m_forward.x));  // forward x (0,1,0)
    m_up = normalize_glm(cross(m_right, m_forward));
}

void Camera::getAxes(float out[9]) const
{
    const Vec3 cols[3] = {m_right, m_up, m_forward};
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) out[3 * c + r] = cols[c][r];
}

}  // namespace pt

#include "FlyCamera.h"

#include <algorithm>

namespace vpthost {

// Mouse deltas are pixels; pitch is kept inside +-89 degrees so the basis never degenerates (behaviour of the reference's
// ProcessMouseMovement, FlyCamera.cpp:8-27).
void FlyCamera::ProcessMouseMovement(float deltaX, float deltaY, bool constrainPitch) {
    m_Pose.yawDeg += m_Controls.degreesPerPixel * deltaX;
    m_Pose.pitchDeg += m_Controls.degreesPerPixel * deltaY;
    if (constrainPitch) m_Pose.pitchDeg = std::min(89.0f, std::max(-89.0f, m_Pose.pitchDeg));
    RebuildBasis();
}

// One step along a basis vector (reference ProcessKeyboard, FlyCamera.cpp:29-55): sign and axis per direction.
void FlyCamera::ProcessKeyboard(Direction direction, float deltaTime) {
    const float step = deltaTime * m_Controls.unitsPerSecond;
    const Vec3* axis = &m_Pose.forward;
    float sign = 1.0f;
    switch (direction) {
        case Direction::FORWARD: break;
        case Direction::BACKWARD: sign = -1.0f; break;
        case Direction::RIGHT: axis = &m_Pose.right; break;
        case Direction::LEFT: axis = &m_Pose.right; sign = -1.0f; break;
        case Direction::DOWN: axis = &m_Pose.up; break;                 // Y-down world: the basis' "up" points down the screen
        case Direction::UP: axis = &m_Pose.up; sign = -1.0f; break;
    }
    m_Pose.eye += *axis * (sign * step);
}

// Spherical angles -> orthonormal basis around the world Y axis (reference UpdateCameraVectors, FlyCamera.cpp:96-108).
void FlyCamera::RebuildBasis() {
    const float yaw = radians(m_Pose.yawDeg), pitch = radians(m_Pose.pitchDeg);
    const float cp = std::cos(pitch);
    m_Pose.forward = normalize(Vec3(cp * std::cos(yaw), std::sin(pitch), cp * std::sin(yaw)));
    m_Pose.right = normalize(cross(m_Pose.forward, Vec3(0.0f, 1.0f, 0.0f)));
    m_Pose.up = normalize(cross(m_Pose.right, m_Pose.forward));
}

// Eye from the inverse view's translation, angles from the view's third row, lens from the projection's diagonal
// (reference InitializeFromMatrices, FlyCamera.cpp:110-140).
void FlyCamera::Adopt(const Mat4& view, const Mat4& projection) {
    const Mat4 toWorld = inverse(view);
    m_Pose.eye = Vec3(toWorld.at(0, 3), toWorld.at(1, 3), toWorld.at(2, 3));
    const Vec3 look = normalize(Vec3(-view.at(2, 0), -view.at(2, 1), -view.at(2, 2)));
    m_Pose.yawDeg = degrees(std::atan2(look.z, look.x));
    m_Pose.pitchDeg = degrees(std::asin(look.y));
    const float sx = projection.at(0, 0), sy = projection.at(1, 1);
    if (sy != 0.0f) {
        m_Lens.fovDeg = degrees(2.0f * std::atan(1.0f / sy));
        if (sx != 0.0f) m_Lens.aspect = sy / sx;
    }
    RebuildBasis();
}

}  // namespace vpthost

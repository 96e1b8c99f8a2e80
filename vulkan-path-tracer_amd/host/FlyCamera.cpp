#include "FlyCamera.h"

namespace vpthost {

void FlyCamera::ProcessMouseMovement(float deltaX, float deltaY, bool constrainPitch) {  // FlyCamera.cpp:8-27
    m_Yaw += deltaX * m_MouseSensitivity;
    m_Pitch += deltaY * m_MouseSensitivity;
    if (constrainPitch) { if (m_Pitch > 89.0f) m_Pitch = 89.0f; if (m_Pitch < -89.0f) m_Pitch = -89.0f; }
    UpdateCameraVectors();
}
void FlyCamera::ProcessKeyboard(Direction direction, float deltaTime) {  // FlyCamera.cpp:29-55 (UP moves against m_Up: the world is Y-down)
    const float v = m_MovementSpeed * deltaTime;
    switch (direction) {
        case Direction::FORWARD: m_Position += m_Front * v; break;
        case Direction::BACKWARD: m_Position -= m_Front * v; break;
        case Direction::LEFT: m_Position -= m_Right * v; break;
        case Direction::RIGHT: m_Position += m_Right * v; break;
        case Direction::UP: m_Position -= m_Up * v; break;
        case Direction::DOWN: m_Position += m_Up * v; break;
    }
}
void FlyCamera::UpdateCameraVectors() {  // FlyCamera.cpp:96-108
    Vec3 front(std::cos(radians(m_Yaw)) * std::cos(radians(m_Pitch)), std::sin(radians(m_Pitch)), std::sin(radians(m_Yaw)) * std::cos(radians(m_Pitch)));
    m_Front = normalize(front);
    m_Right = normalize(cross(m_Front, m_WorldUp));
    m_Up = normalize(cross(m_Right, m_Front));
}
void FlyCamera::InitializeFromMatrices(const Mat4& viewMatrix, const Mat4& projectionMatrix) {  // FlyCamera.cpp:110-140
    Mat4 invView = inverse(viewMatrix);
    m_Position = Vec3(invView.at(0, 3), invView.at(1, 3), invView.at(2, 3));
    Vec3 forward = normalize(Vec3(-viewMatrix.at(2, 0), -viewMatrix.at(2, 1), -viewMatrix.at(2, 2)));
    m_Yaw = degrees(std::atan2(forward.z, forward.x));
    m_Pitch = degrees(std::asin(forward.y));
    if (projectionMatrix.at(1, 1) != 0.0f) m_Fov = degrees(2.0f * std::atan(1.0f / projectionMatrix.at(1, 1)));
    if (projectionMatrix.at(0, 0) != 0.0f && projectionMatrix.at(1, 1) != 0.0f) m_AspectRatio = projectionMatrix.at(1, 1) / projectionMatrix.at(0, 0);
    UpdateCameraVectors();
}

}  // namespace vpthost

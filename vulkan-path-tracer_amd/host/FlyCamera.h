// FlyCamera.h — PathTracer/FlyCamera.h:5-47 without glm: yaw/pitch fly camera producing the view and projection
// matrices whose inverses the integrator consumes (Editor.cpp:1045-1050).
#pragma once
#include "Math.h"

namespace vpthost {

class FlyCamera {
public:
    FlyCamera() = default;
    FlyCamera(const Mat4& viewMatrix, const Mat4& projectionMatrix) { InitializeFromMatrices(viewMatrix, projectionMatrix); }

    void ProcessMouseMovement(float deltaX, float deltaY, bool constrainPitch = true);
    enum class Direction { FORWARD, BACKWARD, LEFT, RIGHT, UP, DOWN };
    void ProcessKeyboard(Direction direction, float deltaTime);

    void SetPosition(const Vec3& position) { m_Position = position; }
    void SetRotation(float yaw, float pitch) { m_Yaw = yaw; m_Pitch = pitch; UpdateCameraVectors(); }
    void SetFov(float fov) { m_Fov = fov; }
    void SetAspectRatio(float aspectRatio) { m_AspectRatio = aspectRatio; }
    void SetNearFar(float nearPlane, float farPlane) { m_NearPlane = nearPlane; m_FarPlane = farPlane; }
    void SetMovementSpeed(float speed) { m_MovementSpeed = speed; }
    void SetMouseSensitivity(float sensitivity) { m_MouseSensitivity = sensitivity; }

    [[nodiscard]] Mat4 GetViewMatrix() const { return lookAt(m_Position, m_Position + m_Front, m_Up); }                       // FlyCamera.cpp:84-89
    [[nodiscard]] Mat4 GetProjectionMatrix() const { return perspective(radians(m_Fov), m_AspectRatio, m_NearPlane, m_FarPlane); }  // FlyCamera.cpp:91-94
    [[nodiscard]] const Vec3& GetPosition() const { return m_Position; }
    [[nodiscard]] const Vec3& GetFront() const { return m_Front; }
    [[nodiscard]] const Vec3& GetUp() const { return m_Up; }
    [[nodiscard]] const Vec3& GetRight() const { return m_Right; }
    [[nodiscard]] float GetYaw() const { return m_Yaw; }
    [[nodiscard]] float GetPitch() const { return m_Pitch; }
    [[nodiscard]] float GetFov() const { return m_Fov; }
    [[nodiscard]] float GetAspectRatio() const { return m_AspectRatio; }

private:
    void UpdateCameraVectors();
    void InitializeFromMatrices(const Mat4& viewMatrix, const Mat4& projectionMatrix);

    Vec3 m_Position{0.0f, 0.0f, 3.0f}, m_Front{0.0f, 0.0f, -1.0f}, m_Up{0.0f, -1.0f, 0.0f}, m_Right{1.0f, 0.0f, 0.0f}, m_WorldUp{0.0f, 1.0f, 0.0f};
    float m_Yaw = -90.0f, m_Pitch = 0.0f;
    float m_MovementSpeed = 5.0f, m_MouseSensitivity = 0.2f, m_Fov = 45.0f, m_AspectRatio = 16.0f / 9.0f, m_NearPlane = 0.1f, m_FarPlane = 1000.0f;
};

}  // namespace vpthost

// FlyCamera.h — the camera Editor drives (reference class FlyCamera, PathTracer/FlyCamera.h:5-47; used by
// Editor.cpp:1045-1050 to refresh the view / projection inverses the integrator consumes).  Only the public member names
// are the reference's; the state is a pose (eye + orthonormal basis derived from yaw / pitch) and a lens description,
// and the matrices come from host/Math.h instead of glm.  The world is Y-down, so "UP" moves against the basis' up vector.
#pragma once
#include "Math.h"

namespace vpthost {

class FlyCamera {
    struct Pose {
        Vec3 eye{0.0f, 0.0f, 3.0f};
        float yawDeg = -90.0f, pitchDeg = 0.0f;
        Vec3 forward{0.0f, 0.0f, -1.0f}, up{0.0f, -1.0f, 0.0f}, right{1.0f, 0.0f, 0.0f};
    };
    struct Lens { float fovDeg = 45.0f, aspect = 16.0f / 9.0f, zNear = 0.1f, zFar = 1000.0f; };
    struct Controls { float unitsPerSecond = 5.0f, degreesPerPixel = 0.2f; };

    Pose m_Pose;
    Lens m_Lens;
    Controls m_Controls;

    void RebuildBasis();                                   // yaw / pitch -> forward, right, up
    void Adopt(const Mat4& view, const Mat4& projection);  // recover pose and lens from a view / projection pair

public:
    enum class Direction { FORWARD, BACKWARD, LEFT, RIGHT, UP, DOWN };

    FlyCamera() = default;
    FlyCamera(const Mat4& viewMatrix, const Mat4& projectionMatrix) { Adopt(viewMatrix, projectionMatrix); }

    // input
    void ProcessMouseMovement(float deltaX, float deltaY, bool constrainPitch = true);
    void ProcessKeyboard(Direction direction, float deltaTime);

    // state in
    void SetPosition(const Vec3& position) { m_Pose.eye = position; }
    void SetRotation(float yaw, float pitch) { m_Pose.yawDeg = yaw; m_Pose.pitchDeg = pitch; RebuildBasis(); }
    void SetFov(float fov) { m_Lens.fovDeg = fov; }
    void SetAspectRatio(float aspectRatio) { m_Lens.aspect = aspectRatio; }
    void SetNearFar(float nearPlane, float farPlane) { m_Lens.zNear = nearPlane; m_Lens.zFar = farPlane; }
    void SetMovementSpeed(float speed) { m_Controls.unitsPerSecond = speed; }
    void SetMouseSensitivity(float sensitivity) { m_Controls.degreesPerPixel = sensitivity; }

    // state out
    Mat4 GetViewMatrix() const { return lookAt(m_Pose.eye, m_Pose.eye + m_Pose.forward, m_Pose.up); }
    Mat4 GetProjectionMatrix() const { return perspective(radians(m_Lens.fovDeg), m_Lens.aspect, m_Lens.zNear, m_Lens.zFar); }
    const Vec3& GetPosition() const { return m_Pose.eye; }
    const Vec3& GetFront() const { return m_Pose.forward; }
    const Vec3& GetUp() const { return m_Pose.up; }
    const Vec3& GetRight() const { return m_Pose.right; }
    float GetYaw() const { return m_Pose.yawDeg; }
    float GetPitch() const { return m_Pose.pitchDeg; }
    float GetFov() const { return m_Lens.fovDeg; }
    float GetAspectRatio() const { return m_Lens.aspect; }
    float GetMovementSpeed() const { return m_Controls.unitsPerSecond; }
    float GetMouseSensitivity() const { return m_Controls.degreesPerPixel; }
};

}  // namespace vpthost

// Camera.h — the pin-hole camera state the Renderer turns into viewport uniforms.
// Interface of the reference's namespace Camera / InputSystem (R/Renderer/Camera.h:12-89,
// R/Window/InputSystem.h:11-21): global state, reset() defaults, move / rotate / zoom driven by the
// input vectors, transformWithInputs() reporting whether anything changed (which resets accumulation).
// The reference never instantiates its input view, so as shipped only reset() matters; a
// headless caller can still drive the camera by writing InputSystem::* before updateUniforms().
#pragma once
#include "VecTypes.h"

namespace MetalCppPathTracer {

namespace InputSystem {
inline mpt::float3 movementInput;   // x right, y up, z forward
inline mpt::float2 rotationInput;   // mouse delta
inline float zoomInput = 0.0f;
inline bool resetInput = false;
inline void clearInputs() {
    movementInput = mpt::float3(0.0f);
    rotationInput = mpt::float2{};
    zoomInput = 0.0f;
    resetInput = false;
}
}  // namespace InputSystem

namespace Camera {

inline mpt::float3 position;
inline mpt::float3 forward;
inline mpt::float3 up;
inline float verticalFov = 60.0f;
inline float focalLength = 1.0f;
inline mpt::float2 screenSize;

inline constexpr float movementSpeed = 0.1f;
inline constexpr float rotationSpeed = 0.002f;
inline constexpr float zoomSpeed = 0.1f;

inline void reset() {  // R/Renderer/Camera.h:24-32
    position = mpt::float3(0.0f, 20.0f, 50.0f);
    forward = mpt::float3(0.0f, 0.0f, -1.0f);
    up = mpt::float3(0.0f, 1.0f, 0.0f);
    verticalFov = 60.0f;
    focalLength = 1.0f;
}

// simd_act(simd::quatf(angle, axis), v) as Apple's <simd/quaternion.h> defines the two (that SDK header is not part of
// the reference tree): q = (sin(angle/2) * axis, cos(angle/2)) WITHOUT normalising the axis, and
// act(q, v) = v + q.real * t + cross(q.imag, t) with t = 2 cross(q.imag, v).  The reference passes
// cross(forward, worldUp) as the axis (R/Renderer/Camera.h:53-54), which is shorter than 1 once the camera pitches: the
// result is then not a pure rotation, and is reproduced as such (the callers normalise it, as the reference does).
inline mpt::float3 rotateAbout(const mpt::float3& v, float angle, const mpt::float3& axis) {
    const float half = angle * 0.5f;
    const mpt::float3 imag = axis * std::sin(half);
    const float real = std::cos(half);
    const mpt::float3 t = mpt::cross(imag, v) * 2.0f;
    return v + t * real + mpt::cross(imag, t);
}

inline bool move(const mpt::float3& dir) {  // R/Renderer/Camera.h:35-47
    if (mpt::dot(dir, dir) == 0.0f) return false;
    const mpt::float3 worldUp(0.0f, 1.0f, 0.0f);
    const mpt::float3 right = mpt::normalize(mpt::cross(forward, worldUp));
    const mpt::float3 ahead = mpt::cross(worldUp, right);
    position = position + movementSpeed * mpt::normalize(right * dir.x + worldUp * dir.y + ahead * dir.z);
    return true;
}

inline bool rotate(const mpt::float2& angles) {  // R/Renderer/Camera.h:49-63
    if (angles.x * angles.x + angles.y * angles.y == 0.0f) return false;
    const mpt::float3 worldUp(0.0f, 1.0f, 0.0f);
    mpt::float3 right = mpt::cross(forward, worldUp);
    forward = mpt::normalize(rotateAbout(forward, -angles.y * rotationSpeed, right));
    right = mpt::cross(forward, worldUp);
    up = mpt::normalize(mpt::cross(right, forward));
    forward = mpt::normalize(rotateAbout(forward, -angles.x * rotationSpeed, up));
    return true;
}

inline bool zoom(float amount) {  // R/Renderer/Camera.h:65-72
    if (amount == 0.0f) return false;
    float f = verticalFov + amount * zoomSpeed;
    verticalFov = f < 30.0f ? 30.0f : (f > 120.0f ? 120.0f : f);
    return true;
}

inline bool transformWithInputs() {  // R/Renderer/Camera.h:75-89
    const bool wasReset = InputSystem::resetInput;
    if (wasReset) reset();
    const bool moved = move(InputSystem::movementInput);
    const bool rotated = rotate(InputSystem::rotationInput);
    const bool zoomed = zoom(InputSystem::zoomInput);
    InputSystem::clearInputs();
    return wasReset || moved || rotated || zoomed;
}

}  // namespace Camera
}  // namespace MetalCppPathTracer

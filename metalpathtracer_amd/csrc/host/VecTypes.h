// VecTypes.h — the small vector vocabulary of the host layer.
// float3 is 16-byte aligned and padded (as Apple's simd::float3 is in the reference's structs, which is
// what makes UniformsData 144 bytes — SURVEY.md App. D); float4 is the element of every flat buffer.
#pragma once
#include <cmath>
#include <cstdint>

namespace mpt {

struct alignas(16) float3 {
    float x = 0, y = 0, z = 0, _pad = 0;
    float3() = default;
    float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit float3(float s) : x(s), y(s), z(s) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct alignas(8) float2 {
    float x = 0, y = 0;
};
struct alignas(16) float4 {
    float x = 0, y = 0, z = 0, w = 0;
    float4() = default;
    float4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    float4(const float3& v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
};
struct uint3 {
    uint32_t x = 0, y = 0, z = 0;
};

inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(const float3& a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(const float3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, const float3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline float3 min3(const float3& a, const float3& b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline float3 max3(const float3& a, const float3& b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }
inline float dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 cross(const float3& a, const float3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float3 normalize(const float3& a) { return a * (1.0f / std::sqrt(dot(a, a))); }

}  // namespace mpt

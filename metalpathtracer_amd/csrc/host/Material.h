// Material.h — per-primitive surface description; the field set and order of the reference's
// Material (R/Scene/Material.h:8-14), which is also the layout of the materials buffer
// (2 float4 per primitive: albedo+type, emission+power — SURVEY.md App. D buf 2).
#pragma once
#include "VecTypes.h"

namespace MetalCppPathTracer {

struct Material {
    mpt::float3 albedo;
    float materialType = 0.0f;   // 0 Lambert; <0 mirror; >0 index of refraction (Scatter.h:22-43); 2 also "emissive"
    mpt::float3 emissionColor;
    float emissionPower = 0.0f;
};

}  // namespace MetalCppPathTracer

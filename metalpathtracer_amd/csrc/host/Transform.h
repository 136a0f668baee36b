// Transform.h — placement of a mesh: vertex -> position + scale * vertex.
// The reference declares this pair (R/Scene/Transform.h:8-20) but never uses the type: its loader applies the two
// numbers inline (R/Scene/SceneLoader.cpp:122-130).  Here the loader goes through it, so that the one place that turns an
// OBJ vertex into a world-space vertex is named; the arithmetic (one multiply, one add per component, in that order) is
// what decides the vertex bits and is unchanged.
#pragma once
#include "VecTypes.h"

namespace MetalCppPathTracer {

struct Transform {
    mpt::float3 position;
    float scale;

    Transform() : position(0.0f), scale(1.0f) {}
    Transform(const mpt::float3& p, float s) : position(p), scale(s) {}

    mpt::float3 apply(const mpt::float3& v) const { return position + scale * v; }
};

}  // namespace MetalCppPathTracer

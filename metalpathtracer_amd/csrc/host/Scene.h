// Scene.h — host scene container, BVH builder and flat-buffer packers.
//
// Public surface = the reference's class Scene (R/Scene/Scene.h:34-188): same method names, argument
// meaning, return types (caller-owned new[] arrays of float4 / int) and ordering guarantees, so code written
// against the reference's Scene compiles against this one (simd::floatN -> mpt::floatN).  The implementation
// (Scene.cpp) is this project's own: per-primitive bounds are computed once, the sweep buffers are reused,
// and an optional fast binned builder exists next to the reference-compatible one.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "Material.h"
#include "VecTypes.h"

namespace MetalCppPathTracer {

enum class PrimitiveType { Sphere = 0, Triangle = 1 };

struct Primitive {
    PrimitiveType type = PrimitiveType::Sphere;
    mpt::float3 data0;  // sphere: centre            triangle: vertex 0
    mpt::float3 data1;  // sphere: (radius, 0, 0)    triangle: vertex 1
    mpt::float3 data2;  // sphere: unused            triangle: vertex 2
    Material material;
};

struct BVHNode {
    mpt::float3 boundsMin;
    mpt::float3 boundsMax;
    int leftFirst = 0;  // leaf: first slot in the primitive-index list; internal: left child (always node + 1)
    int count = 0;      // > 0: leaf primitive count; <= 0: minus the right child's index
};

class Scene {
public:
    enum class BuildMode {
        ReferenceSweep,  // full-sweep SAH keyed on data0[axis], leaf <= 8 (R/Scene/Scene.h:195-317): same tree
        BinnedCentroid,  // 16-bin SAH on centroids, O(n log n): for large scenes; image-equivalent, not tree-equal
        GpuLbvh          // built on the GPU (mpt_build_bvh: top-down binned SAH, leaves <= 6 below 8192 primitives and <= 2 from there on — mpt_gpu_leaf_max; MPT_GPU_BUILD = ploc | lbvh for the
                         // Morton-code builders); needs a device (MPT_BUILD_DEVICE, default 0) and throws without one — there
                         // is no CPU fallback
    };

    Scene() = default;

    void clear();
    size_t addPrimitive(const Primitive& p);
    size_t getPrimitiveCount() const { return primitives_.size(); }
    size_t getSphereCount() const;
    size_t getTriangleCount() const;
    const std::vector<size_t>& getPrimitiveIndices() const { return primitiveIndices_; }
    const std::vector<Primitive>& getPrimitives() const { return primitives_; }

    void sortPrimitives();           // spheres before triangles, stable: the first thing buildBVH does (R/Scene/Scene.h:72-75)
    void dropBVH();                  // forget the host tree (the device holds another one: Renderer's device build)
    void buildBVH();                 // reference-compatible tree
    void buildBVH(BuildMode mode);
    size_t getBVHNodeCount() const { return nodes_.size(); }
    const std::vector<BVHNode>& getBVHNodes() const { return nodes_; }
    int getBVHDepth() const;
    double lastGpuBuildMs() const { return lastGpuBuildMs_; }   // HIP-event time of the last GpuLbvh build
    // takes a tree in the reference's buffer format (8 floats per node, one int per primitive): what mpt_build_bvh and
    // mpt_download_bvh return
    void adoptBVH(const float* bvh, size_t nodeCount, const int32_t* primIdx);

    // Flat buffers in the layout the hot path consumes.  Caller owns the arrays (delete[]).
    mpt::float4* createTransformsBuffer() const;       // 3 float4 / primitive
    mpt::float4* createMaterialsBuffer() const;        // 2 float4 / primitive
    mpt::float4* createSphereBuffer() const;           // 1 float4 / sphere  (centre, radius)
    mpt::float4* createSphereMaterialsBuffer() const;  // 2 float4 / sphere
    mpt::float4* createBVHBuffer() const;              // 2 float4 / node
    int* createPrimitiveIndexBuffer() const;           // 1 int / primitive
    void createTriangleBuffers(std::vector<mpt::float3>& outVertices, std::vector<mpt::uint3>& outIndices) const;

private:
    void buildOnGpu();
    std::vector<Primitive> primitives_;
    std::vector<size_t> primitiveIndices_;
    std::vector<BVHNode> nodes_;
    double lastGpuBuildMs_ = 0.0;
};

}  // namespace MetalCppPathTracer

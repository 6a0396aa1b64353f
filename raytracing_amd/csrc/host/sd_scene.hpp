// sd_scene.hpp -- host-side model of the reference's flat-array scene ("sd" = SimplifiedData).
//
// Mirrors sd::DataStorage / sd::Mesh / sd::BVH::BuildBVHFromNodes / sd::Scene
// (CPURayTrace/SimplifiedData.{hpp,cpp}, Program/Scene.{hpp,cpp}) so that a headless program can
// build exactly the node and triangle arrays the reference application would hand to the tracer.
// Records are stored directly in the C ABI's rt_node / rt_tri layout (byte-compatible with sd::Node
// and sd::Triangle), so the arrays go to rt_upload_scene without conversion.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace rthost {

struct SdVertex { // sd::Vertex, SimplifiedData.hpp:37-42
    float position[3];
    float normal[3];
    float texCoord[2];
};

// One mesh as an importer delivers it: vertices + triangle indices (Include/ModelLoader.hpp:199-230).
struct MeshData {
    std::vector<SdVertex> vertices;
    std::vector<unsigned int> indices;
};

// sd::DataStorage (SimplifiedData.hpp:123-129).  Grows on demand: the reference's fixed capacities
// (2^20 triangles, 2^21-1 nodes, SimplifiedData.hpp:90-91) are not inherited.
struct SdDataStorage {
    std::vector<rt_tri> triangles;
    std::vector<rt_node> nodes;
    uint32_t rootIndex = 0xFFFFFFFFu;

    uint32_t addTriangle(const rt_tri &t);
    uint32_t addNode(const rt_node &n);
};

// sd::BVH::BuildBVHFromNodes (SimplifiedData.cpp:322-399): median split over nodeIndices[start,end),
// children emitted before their parent.  Throws std::runtime_error on an empty range as the reference does.
uint32_t BuildBVHFromNodes(SdDataStorage &storage, uint32_t *nodeIndices, size_t start, size_t end);

// sd::Mesh::Mesh (SimplifiedData.cpp:98-136): appends one leaf per triangle and the mesh BVH;
// returns the mesh root node index.
uint32_t AddMesh(SdDataStorage &storage, const MeshData &mesh);

// sd::Scene (Program/Scene.hpp:98-115)
class SdScene {
  public:
    std::unique_ptr<SdDataStorage> pDataStorage;
    std::vector<uint32_t> sceneIndices;

    SdScene();
    SdScene(const SdScene &other);
    SdScene &operator=(const SdScene &other);

    // sd::ModelLoader::LoadModelFileSync (Include/ModelLoader.hpp:260-273) followed by
    // sceneIndices.push_back(root) (Program/Scene.cpp:203)
    uint32_t addModel(const std::vector<MeshData> &meshes);
    uint32_t loadModelFile(const std::string &path);
    // scene-level BVH over the model roots, Program/Scene.cpp:205-206
    void finalize();
};

// Flat float packers of the reference's GLSL path (SimplifiedData.cpp:162-223): 9 floats per node,
// 25 floats per triangle, indices and flags bit-punned into floats.
void ConvertNodeToFlatStorage(const SdDataStorage &storage, std::vector<float> &flat);
void ConvertTriangleToFlatStorage(const SdDataStorage &storage, std::vector<float> &flat);

} // namespace rthost

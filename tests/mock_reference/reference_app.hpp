// reference_app.hpp -- test fixture: the identifiers of wcg21444/RayTracing that the HIP render modes bind to, declared
// with the reference's own names and member layout and NOTHING else of the application (no GL, no glm, no importer).
// raytracing_amd/csrc/host/reference_binding.hpp is compiled against this file by tests/test_reference_binding_cpu.py; inside
// the application the reference's headers take its place.  Where each name comes from:
//     vec3 / vec2                       glm::vec3 / glm::vec2 (Include/Utils.hpp): members x, y, z
//     sd::BoundingBox, Node, Triangle   CPURayTrace/SimplifiedData.hpp:64-88
//     sd::TriangleStorage, NodeStorage  :93-121 -- vectors at full capacity, nextIndex = entries in use
//     sd::DataStorage                   :123-129
//     sd::Scene                         Program/Scene.hpp:100-104 (pDataStorage)
//     Storage::SdScene, SdSceneMutex    Renderer/Storage.hpp:38-39
//     Camera                            Include/Camera.hpp:10-15
//     Trace::bounceLimit                CPURayTrace/Trace.hpp:16
//     Texture2D                         GLShader/Texture.hpp:18-59 (Width, Height, ID, setData)
//     ITraceMethod, ILoadMethod, IRenderPipeline, RenderPipeline   Renderer/RenderInterfaces.hpp:5-78
#pragma once

#include <cstdint>
#include <memory>
#include <shared_mutex>
#include <utility>
#include <vector>

struct vec2 {
    float x, y;
};
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};

namespace SimplifiedData {
struct BoundingBox {
    vec3 pMin, pMax;
};
struct Node {
    uint32_t left;
    uint32_t right;
    BoundingBox box;
    uint8_t flags;
};
struct Triangle {
    vec3 positions[3];
    vec3 normals[3];
    vec2 texCoords[3];
    uint16_t matFlags;
};
inline constexpr uint32_t TRIANGLESIZE = 1 << 20;
inline constexpr uint32_t NODESIZE = TRIANGLESIZE * 2 - 1;
class TriangleStorage {
  public:
    TriangleStorage() : triangles(TRIANGLESIZE) {}
    std::vector<Triangle> triangles;
    uint32_t nextIndex = 0;
};
class NodeStorage {
  public:
    NodeStorage() : nodes(NODESIZE) {}
    std::vector<Node> nodes;
    uint32_t nextIndex = 0;
    uint32_t nextIndexBack = NODESIZE - 1;
};
struct DataStorage {
    TriangleStorage triangleStorage;
    NodeStorage nodeStorage;
    uint32_t rootIndex;
};
class Scene {
  public:
    std::unique_ptr<DataStorage> pDataStorage = std::make_unique<DataStorage>();
    std::vector<uint32_t> sceneIndices;
};
} // namespace SimplifiedData
namespace sd = SimplifiedData;

namespace Storage {
inline sd::Scene SdScene;
inline std::shared_mutex SdSceneMutex;
} // namespace Storage

class Camera {
  public:
    float focalLength;
    vec3 position;
    vec3 lookAtCenter;
    float width;
    float height;
    float aspectRatio;
};

namespace Trace {
inline int bounceLimit = 4;
}

using TextureID = unsigned int;
class Texture2D {
  public:
    TextureID ID = 0;
    unsigned int Width = 0, Height = 0;
    std::vector<float> lastUpload; // (fixture only: what setData received)
    void setData(void *data) {
        if (data)
            lastUpload.assign((const float *)data, (const float *)data + (size_t)Width * Height * 4);
        else
            lastUpload.clear();
    }
};

class ITraceMethod {
  public:
    virtual void trace(const Texture2D &traceInput, Texture2D &traceOutput, int sampleCount) = 0;
    virtual ~ITraceMethod() {}
};
class ILoadMethod {
  public:
    virtual void load() = 0;
    virtual ~ILoadMethod() {}
};
class IRenderPipeline {
  public:
    virtual ~IRenderPipeline() {}
    virtual ILoadMethod *getLoadMethod() = 0;
    virtual ITraceMethod *getTraceMethod() = 0;
};
template <typename ContextType, typename LoadMethodType, typename TraceMethodType> class RenderPipeline : public IRenderPipeline {
    std::unique_ptr<ContextType> context;
    std::unique_ptr<LoadMethodType> loadMethod;
    std::unique_ptr<TraceMethodType> traceMethod;

  public:
    explicit RenderPipeline(ContextType &&c)
        : context(std::make_unique<ContextType>(std::move(c))), loadMethod(std::make_unique<LoadMethodType>(*context)),
          traceMethod(std::make_unique<TraceMethodType>(*context)) {}
    ~RenderPipeline() override {
        traceMethod.reset();
        loadMethod.reset();
    }
    ILoadMethod *getLoadMethod() override { return loadMethod.get(); }
    ITraceMethod *getTraceMethod() override { return traceMethod.get(); }
};

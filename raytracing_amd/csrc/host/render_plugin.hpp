// render_plugin.hpp -- the HIP tracer as a render mode behind the reference's plug-in interface.
//
// In the reference application a render mode is a RenderPipeline<Context, LoadMethod, TraceMethod>
// (Renderer/RenderInterfaces.hpp:55-78) built in Renderer::changeMode (Renderer/Renderer.cpp:31-62).
// This header provides the two modes that run on the MI355X:
//     RenderPipeline<SdSceneHIPContext, LoadSdSceneHIP, TraceSdSceneHIP>   (flat-array scene)
//     RenderPipeline<SceneHIPContext,   LoadSceneHIP,   TraceSceneHIP>     (pointer-based scene)
// plus headless stand-ins for what they touch in the GL application: Texture2D (only Width, Height,
// ID and setData, GLShader/Texture.hpp:18-59), Camera (fields of Include/Camera.hpp:10-15), the
// process-global scenes of Renderer/Storage.hpp:38-41 and the TracerAsync scheduler
// (Renderer/TracerImpl.cpp:9-31).  The load and trace methods themselves are in hip_render_modes.hpp, written against a
// description of the application (`App`); HeadlessApp below describes the stand-ins, reference_binding.hpp the reference's
// own types.
#pragma once

#include <atomic>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hip_render_modes.hpp"
#include "old_scene.hpp"
#include "rt_hip.h"
#include "sd_scene.hpp"

namespace rthost {

// ---- stand-ins for application types ------------------------------------------------------------
using TextureID = unsigned int;

// RGBA32F image with the bottom row first, as glTexImage2D receives it (GLShader/Texture.cpp:86-91).
class Texture2D {
  public:
    TextureID ID = 0;
    unsigned int Width = 0, Height = 0;
    std::vector<float> texels;

    void generate(unsigned int w, unsigned int h) {
        static TextureID next = 1;
        ID = next++;
        resize((int)w, (int)h);
    }
    void resize(int w, int h) {
        Width = (unsigned)w;
        Height = (unsigned)h;
        texels.assign((size_t)w * h * 4, 0.0f);
    }
    void setData(void *data) { // NULL clears, like glTexImage2D with a null pointer
        if (data)
            texels.assign((const float *)data, (const float *)data + (size_t)Width * Height * 4);
        else
            texels.assign((size_t)Width * Height * 4, 0.0f);
    }
    void swap(Texture2D &o) {
        std::swap(ID, o.ID);
        std::swap(Width, o.Width);
        std::swap(Height, o.Height);
        texels.swap(o.texels);
    }
    friend void swap(Texture2D &a, Texture2D &b) { a.swap(b); }
};

struct Camera { // data members of class Camera, Include/Camera.hpp:10-15 and its constructor :19-31
    float focalLength = 1.0f;
    float position[3] = {0.f, 0.f, 1.f};
    float lookAtCenter[3] = {0.f, 0.f, 0.f};
    float width = 2.0f;
    float height = 2.0f / (16.0f / 9.0f);
    float aspectRatio = 16.0f / 9.0f;
    Camera() {}
    Camera(float focal, const float pos[3], float w, float aspect, const float look[3])
        : focalLength(focal), width(w), height(w / aspect), aspectRatio(aspect) {
        for (int k = 0; k < 3; k++) {
            position[k] = pos[k];
            lookAtCenter[k] = look[k];
        }
    }
};

namespace Storage { // Renderer/Storage.hpp:38-41, Renderer/Storage.cpp:5-10
extern SdScene SdScene_;
extern std::shared_mutex SdSceneMutex;
extern OldScene OldScene_;
extern std::shared_mutex OldSceneMutex;
} // namespace Storage

namespace Trace {
extern int bounceLimit; // CPURayTrace/Trace.hpp:16
}
namespace BVHSettings {
extern bool toggleBVHAccel; // Program/BVHUI.hpp:14: the "BVH Acceleration" checkbox; read by the old-scene path only
}                           // (Trace.cpp:42-45, Mesh.hpp:156), as in the reference

// ---- the plug-in interface, as in Renderer/RenderInterfaces.hpp ---------------------------------
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
class ITracer {
  public:
    virtual void render(ITraceMethod &method) = 0;
    virtual TextureID getTraceOutputTextureID() = 0;
    virtual void resetSamples() = 0;
    virtual void resize(int newWidth, int newHeight) = 0;
    virtual void waitForCompletion() = 0;
    virtual ~ITracer() {}
};
class IRenderPipeline {
  public:
    virtual ~IRenderPipeline() {}
    virtual ILoadMethod *getLoadMethod() = 0;
    virtual ITraceMethod *getTraceMethod() = 0;
};
template <typename ContextType, typename LoadMethodType, typename TraceMethodType>
class RenderPipeline : public IRenderPipeline {
    std::unique_ptr<ContextType> context; // declared first: destroyed after the methods that refer to it
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

// ---- the application as the HIP render modes see it (hip_render_modes.hpp) ------------------------
struct HeadlessApp {
    using ITraceMethod = rthost::ITraceMethod;
    using ILoadMethod = rthost::ILoadMethod;
    using Texture2D = rthost::Texture2D;
    using Camera = rthost::Camera;
    static std::shared_mutex &sdSceneMutex() { return Storage::SdSceneMutex; }
    static SdSceneArrays sdScene() {
        const SdDataStorage &ds = *Storage::SdScene_.pDataStorage;
        return SdSceneArrays{ds.nodes.data(), (uint32_t)ds.nodes.size(), ds.rootIndex, ds.triangles.data(), (uint32_t)ds.triangles.size()};
    }
    static std::shared_mutex &oldSceneMutex() { return Storage::OldSceneMutex; }
    static OldSceneArrays oldScene() {
        const OldScene &s = Storage::OldScene_;
        return OldSceneArrays{s.nodes.data(),   (uint32_t)s.nodes.size(),   s.rootIndex,        s.triangles.data(), (uint32_t)s.triangles.size(),
                              s.spheres.data(), (uint32_t)s.spheres.size(), s.materials.data(), (uint32_t)s.materials.size()};
    }
    static void readCamera(const Camera &c, rt_camera &cam) {
        cam.focal_length = c.focalLength;
        for (int k = 0; k < 3; k++) {
            cam.position[k] = c.position[k];
            cam.look_at[k] = c.lookAtCenter[k];
        }
        cam.width = c.width;
        cam.height = c.height;
        cam.aspect_ratio = c.aspectRatio;
    }
    static int bounceLimit() { return Trace::bounceLimit; }
    static bool bvhAcceleration() { return BVHSettings::toggleBVHAccel; }
};
using SdSceneHIPContext = SdSceneHIPContextT<HeadlessApp>;
using SceneHIPContext = SceneHIPContextT<HeadlessApp>;
using LoadSdSceneHIP = LoadSdSceneHIPT<HeadlessApp>;
using LoadSceneHIP = LoadSceneHIPT<HeadlessApp>;
using TraceSdSceneHIP = TraceSdSceneHIPT<HeadlessApp>;
using TraceSceneHIP = TraceSceneHIPT<HeadlessApp>;

// ---- headless scheduler, Renderer/TracerImpl.cpp:4-31 ---------------------------------------------
class TracerAsync : public ITracer {
    Texture2D traceInput, traceOutput;
    int currentSampleCount = 1;

  public:
    TracerAsync(int width, int height) {
        traceInput.generate(width, height);
        traceOutput.generate(width, height);
    }
    void render(ITraceMethod &method) override {
        std::swap(traceInput, traceOutput);
        method.trace(traceInput, traceOutput, currentSampleCount++);
    }
    TextureID getTraceOutputTextureID() override { return traceOutput.ID; }
    const Texture2D &output() const { return traceOutput; }
    int sampleCount() const { return currentSampleCount; }
    void resetSamples() override {
        traceInput.setData(nullptr);
        traceOutput.setData(nullptr);
        currentSampleCount = 1;
    }
    void resize(int newWidth, int newHeight) override {
        traceInput.resize(newWidth, newHeight);
        traceOutput.resize(newWidth, newHeight);
    }
    void waitForCompletion() override {}
};

} // namespace rthost

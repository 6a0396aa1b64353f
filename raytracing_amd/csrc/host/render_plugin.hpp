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
// (Renderer/TracerImpl.cpp:9-31).  Inside the real application the stand-ins are replaced by the
// reference's own headers; the adaptor classes compile against either.
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

// ---- contexts (the counterpart of SdSceneCPUContext / SceneCPUContext, RenderContexts.hpp:31-63) --
// Owns the device context.  The scene snapshot lives in HBM, so there is no host copy to guard; the
// library serialises upload against trace itself.
struct HIPContextBase {
    rt_ctx *ctx = nullptr;
    Camera &cam;
    uint32_t seed = 1;
    float perturbStrength = 0.001f; // Renderer/TraceMethods.cpp:48
    std::atomic<uint64_t> sceneVersion{0}; // counts the loads: frames computed ahead of a load are of the old scene

    explicit HIPContextBase(Camera &c, int device = 0) : cam(c) {
        int rc = rt_create(&ctx, device);
        if (rc != RT_OK)
            throw std::runtime_error("rt_create failed (" + std::to_string(rc) + "): no usable HIP device");
    }
    // several GPUs behind the one render mode (rt_create_multi): the frame is put together on devices[0]
    HIPContextBase(Camera &c, const std::vector<int> &devices) : cam(c) {
        int rc = rt_create_multi(&ctx, devices.data(), (int)devices.size());
        if (rc != RT_OK)
            throw std::runtime_error("rt_create_multi failed (" + std::to_string(rc) + "): no usable HIP device");
    }
    HIPContextBase(HIPContextBase &&o) noexcept
        : ctx(o.ctx), cam(o.cam), seed(o.seed), perturbStrength(o.perturbStrength), sceneVersion(o.sceneVersion.load()) {
        o.ctx = nullptr;
    }
    HIPContextBase(const HIPContextBase &) = delete;
    ~HIPContextBase() {
        if (ctx)
            rt_destroy(ctx);
    }
    void check(int rc, const char *what) const {
        if (rc != RT_OK)
            throw std::runtime_error(std::string(what) + ": " + rt_last_error(ctx));
    }
};
struct SdSceneHIPContext : HIPContextBase {
    using HIPContextBase::HIPContextBase;
};
struct SceneHIPContext : HIPContextBase {
    using HIPContextBase::HIPContextBase;
};

// ---- load methods -------------------------------------------------------------------------------
// LoadSdSceneCPU::load (Renderer/LoadMethods.cpp:55-61) copies Storage::SdScene into the context;
// here the copy goes to HBM.
class LoadSdSceneHIP : public ILoadMethod {
    SdSceneHIPContext &DIContext;

  public:
    explicit LoadSdSceneHIP(SdSceneHIPContext &c) : DIContext(c) {}
    void load() override {
        std::shared_lock<std::shared_mutex> sceneReadLock(Storage::SdSceneMutex);
        const SdDataStorage &ds = *Storage::SdScene_.pDataStorage;
        DIContext.check(rt_upload_scene(DIContext.ctx, ds.nodes.data(), (uint32_t)ds.nodes.size(), ds.rootIndex,
                                        ds.triangles.data(), (uint32_t)ds.triangles.size(), nullptr, 0, nullptr, 0,
                                        RT_ORDER_RIGHT_FIRST),
                        "LoadSdSceneHIP");
        DIContext.sceneVersion++;
    }
};
// LoadSceneCPU::load (Renderer/LoadMethods.cpp:65-73) catches and logs; so does this one.
class LoadSceneHIP : public ILoadMethod {
    SceneHIPContext &DIContext;

  public:
    std::string lastError;
    explicit LoadSceneHIP(SceneHIPContext &c) : DIContext(c) {}
    void load() override {
        try {
            std::shared_lock<std::shared_mutex> sceneReadLock(Storage::OldSceneMutex);
            const OldScene &s = Storage::OldScene_;
            DIContext.check(rt_upload_scene(DIContext.ctx, s.nodes.data(), (uint32_t)s.nodes.size(), s.rootIndex,
                                            s.triangles.data(), (uint32_t)s.triangles.size(), s.spheres.data(),
                                            (uint32_t)s.spheres.size(), s.materials.data(), (uint32_t)s.materials.size(),
                                            RT_ORDER_LEFT_FIRST),
                            "LoadSceneHIP");
            DIContext.sceneVersion++;
            lastError.clear();
        } catch (std::exception &e) {
            lastError = e.what();
            fprintf(stderr, "Error loading Old Scene: %s\n", e.what());
        }
    }
};

// ---- trace methods ------------------------------------------------------------------------------
// One call = one more sample per pixel folded into the running mean, then the image handed to the
// output texture (Renderer/TraceMethods.cpp:45-80).  The running mean stays on the device between
// calls; sampleCount == 1 restarts it.
//
// Look-ahead: the reference's frame loop asks for sample n+1 right after sample n unless something changed
// (TracerAsync::render, Renderer/TracerImpl.cpp:9-12), so when a call returns sample n the samples n+1 and n+2 are
// already in flight on the device (rt_trace_begin), their images on the way to page-locked staging buffers while the
// application uploads and draws frame n.  The next call finds its frame ready if it asks for sample n+1 of the same
// frame size with the same camera, options and scene; anything else (RenderState::Dirty -> resetSamples, a resize, a
// load) drops what is in flight -- it never touched the running mean the application has seen -- and starts over.
template <int ESTIMATOR, class Context> class TraceHIPImpl : public ITraceMethod {
    Context &DIContext;
    static constexpr int kAhead = 2;          // calls kept in flight
    float *staging[kAhead + 1] = {nullptr, nullptr, nullptr}; // page-locked (pageable memory if that cannot be had)
    size_t stagingFloats = 0;
    bool stagingPinned = false;
    int nextStaging = 0;
    struct Flight {
        int sample;
        unsigned width, height;
        rt_camera cam;
        rt_options opt;
        uint64_t scene;
        float *image;
    };
    std::deque<Flight> flights;
    void releaseStaging() {
        for (float *&p : staging) {
            if (p && stagingPinned)
                rt_free_host(p);
            else
                delete[] p;
            p = nullptr;
        }
        stagingFloats = 0;
    }
    void drop() {
        if (!flights.empty())
            (void)rt_trace_discard(DIContext.ctx);
        flights.clear();
    }
    void begin(int sample, unsigned w, unsigned h, const rt_camera &cam, const rt_options &opt, uint64_t scene) {
        float *image = staging[nextStaging];
        nextStaging = (nextStaging + 1) % (kAhead + 1);
        DIContext.check(rt_trace_begin(DIContext.ctx, (int)w, (int)h, sample, 1, image), "rt_trace_begin"); // RT_ERR_NO_SCENE carries
        flights.push_back(Flight{sample, w, h, cam, opt, scene, image});                                   // "Scene is not loaded."
    }

    // rt_trace_end reports the context's counters as accumulated since they were last read out; a frame's own share is
    // the difference to what the frame before it reported (the counters restart whenever a synchronous call reads them).
    rt_stats cumulative{};
    void endFrame(const char *what, bool must) {
        rt_stats now{};
        const int rc = rt_trace_end(DIContext.ctx, &now);
        if (rc != RT_OK) {
            if (must)
                DIContext.check(rc, what);
            throw rc;
        }
        const bool restarted = now.rays < cumulative.rays;
        const rt_stats before = restarted ? rt_stats{} : cumulative;
        cumulative = now;
        lastStats = rt_stats{};
        lastStats.rays = now.rays - before.rays;
        lastStats.box_tests = now.box_tests - before.box_tests;
        lastStats.tri_tests = now.tri_tests - before.tri_tests;
        lastStats.sph_tests = now.sph_tests - before.sph_tests;
        lastStats.tri_hits = now.tri_hits - before.tri_hits;
        lastStats.replays = now.replays - before.replays;
    }

  public:
    // Counters of the last frame.  Synchronous frames (several devices, or lookAhead off and on a multi-device context):
    // exactly that call's, with its kernel time.  With frames computed ahead: what the device counted between this frame's
    // hand-over and the one before it -- over a run of frames they add up to the run's total, but a single frame's figure
    // includes whatever part of the next frames had already run; kernel_ms and launches are 0 (no timing events are taken).
    rt_stats lastStats{};
    bool lookAhead = true;
    explicit TraceHIPImpl(Context &c) : DIContext(c) {}
    ~TraceHIPImpl() override {
        drop();
        releaseStaging();
    }
    TraceHIPImpl(const TraceHIPImpl &) = delete;
    TraceHIPImpl &operator=(const TraceHIPImpl &) = delete;
    void trace(const Texture2D &traceInput, Texture2D &traceOutput, int sampleCount) override {
        const Camera &c = DIContext.cam; // read live every call, the UI may have moved it
        rt_camera cam;
        std::memset(&cam, 0, sizeof(cam));
        cam.focal_length = c.focalLength;
        for (int k = 0; k < 3; k++) {
            cam.position[k] = c.position[k];
            cam.look_at[k] = c.lookAtCenter[k];
        }
        cam.width = c.width;
        cam.height = c.height;
        cam.aspect_ratio = c.aspectRatio;
        rt_options opt;
        std::memset(&opt, 0, sizeof(opt));
        opt.bounce_limit = Trace::bounceLimit;
        opt.estimator = ESTIMATOR;
        opt.seed = DIContext.seed;
        opt.jitter = DIContext.perturbStrength;
        opt.tile_rank = 0;
        opt.tile_world = 1;
        opt.traversal = (ESTIMATOR == RT_EST_OLD_RECURSIVE && !BVHSettings::toggleBVHAccel) ? RT_TRAV_BRUTE : RT_TRAV_AUTO;
        const unsigned w = traceInput.Width, h = traceInput.Height;
        const uint64_t scene = DIContext.sceneVersion.load();
        const size_t need = (size_t)w * h * 4;
        if (need != stagingFloats) {
            drop();
            releaseStaging();
            stagingPinned = true;
            for (float *&p : staging) {
                p = stagingPinned ? (float *)rt_alloc_host(need * sizeof(float)) : nullptr;
                if (!p) {
                    stagingPinned = false; // (buffers already handed out stay page-locked only if all are)
                    break;
                }
            }
            if (!stagingPinned) {
                for (float *&p : staging)
                    if (p) {
                        rt_free_host(p);
                        p = nullptr;
                    }
                for (float *&p : staging)
                    p = new float[need];
            }
            stagingFloats = need;
        }
        const bool single = rt_device_count(DIContext.ctx) == 1;
        if (!single) { // several devices behind the context: synchronous calls
            DIContext.check(rt_set_camera(DIContext.ctx, &cam), "rt_set_camera");
            DIContext.check(rt_set_options(DIContext.ctx, &opt), "rt_set_options");
            DIContext.check(rt_trace(DIContext.ctx, (int)w, (int)h, sampleCount, 1, staging[0], nullptr, &lastStats), "rt_trace");
            traceOutput.setData(staging[0]);
            return;
        }
        auto matches = [&](const Flight &f, int sample) {
            return f.sample == sample && f.width == w && f.height == h && f.scene == scene &&
                   std::memcmp(&f.cam, &cam, sizeof(cam)) == 0 && std::memcmp(&f.opt, &opt, sizeof(opt)) == 0;
        };
        if (!flights.empty() && !matches(flights.front(), sampleCount))
            drop(); // not the frame that was computed ahead
        DIContext.check(rt_set_camera(DIContext.ctx, &cam), "rt_set_camera");
        DIContext.check(rt_set_options(DIContext.ctx, &opt), "rt_set_options");
        if (flights.empty())
            begin(sampleCount, w, h, cam, opt, scene);
        try {
            endFrame("rt_trace_end", false);
        } catch (int) {
            // a load went in between (it drops what is in flight): this frame again, from the scene as it is now
            flights.clear();
            begin(sampleCount, w, h, cam, opt, DIContext.sceneVersion.load());
            endFrame("rt_trace_end", true);
        }
        float *image = flights.front().image;
        flights.pop_front();
        while (lookAhead && (int)flights.size() < kAhead)
            begin((flights.empty() ? sampleCount : flights.back().sample) + 1, w, h, cam, opt, scene);
        traceOutput.setData(image);
    }
};
using TraceSdSceneHIP = TraceHIPImpl<RT_EST_SD_LOOP, SdSceneHIPContext>;
using TraceSceneHIP = TraceHIPImpl<RT_EST_OLD_RECURSIVE, SceneHIPContext>;

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

// hip_render_modes.hpp -- the load and trace methods of the HIP render modes, written once against a small description of
// the application they are installed in.
//
// The reference builds a render mode as RenderPipeline<Context, LoadMethod, TraceMethod>
// (Renderer/RenderInterfaces.hpp:55-78).  The classes below are such contexts and methods; everything they need from the
// application comes through ONE policy class `App`, so the same code serves the headless stand-ins of this repository
// (render_plugin.hpp: HeadlessApp) and the reference's own types (reference_binding.hpp: ReferenceApp, written against
// the member names of CPURayTrace/SimplifiedData.hpp:93-129, Renderer/Storage.hpp:38-41, Include/Camera.hpp:10-15).
//
// `App` provides
//     types    ITraceMethod, ILoadMethod (the interfaces the methods derive from), Texture2D (Width, Height, setData), Camera
//     static   std::shared_mutex &sdSceneMutex();   SdSceneArrays sdScene();     the flat-array scene, read under the mutex
//     static   std::shared_mutex &oldSceneMutex();  OldSceneArrays oldScene();   the pointer-based scene (only if that mode is built)
//     static   void readCamera(const Camera &, rt_camera &);
//     static   int bounceLimit();                                                Trace::bounceLimit, CPURayTrace/Trace.hpp:16
//     static   bool bvhAcceleration();                                           BVHSettings::toggleBVHAccel, Program/BVHUI.hpp:14
#pragma once

#include <atomic>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace rthost {

// What a load method hands to rt_upload_scene: the arrays as the application holds them, and how many entries are in use.
struct SdSceneArrays {
    const rt_node *nodes;
    uint32_t n_nodes, root;
    const rt_tri *triangles;
    uint32_t n_triangles;
};
struct OldSceneArrays {
    const rt_node *nodes;
    uint32_t n_nodes, root;
    const rt_tri *triangles;
    uint32_t n_triangles;
    const rt_sphere *spheres;
    uint32_t n_spheres;
    const rt_material *materials;
    uint32_t n_materials;
};

// ---- contexts (the counterpart of SdSceneCPUContext / SceneCPUContext, RenderContexts.hpp:31-63) --
// Owns the device context.  The scene snapshot lives in HBM, so there is no host copy to guard; the
// library serialises upload against trace itself.
template <class App> struct HIPContextBase {
    using Camera = typename App::Camera;
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
template <class App> struct SdSceneHIPContextT : HIPContextBase<App> {
    using HIPContextBase<App>::HIPContextBase;
};
template <class App> struct SceneHIPContextT : HIPContextBase<App> {
    using HIPContextBase<App>::HIPContextBase;
};

// ---- load methods -------------------------------------------------------------------------------
// LoadSdSceneCPU::load (Renderer/LoadMethods.cpp:55-61) copies Storage::SdScene into the context;
// here the copy goes to HBM.
template <class App> class LoadSdSceneHIPT : public App::ILoadMethod {
    SdSceneHIPContextT<App> &DIContext;

  public:
    explicit LoadSdSceneHIPT(SdSceneHIPContextT<App> &c) : DIContext(c) {}
    void load() override {
        std::shared_lock<std::shared_mutex> sceneReadLock(App::sdSceneMutex());
        const SdSceneArrays a = App::sdScene();
        DIContext.check(rt_upload_scene(DIContext.ctx, a.nodes, a.n_nodes, a.root, a.triangles, a.n_triangles, nullptr, 0, nullptr, 0,
                                        RT_ORDER_RIGHT_FIRST), // IntersectLoop pops the right child first, SimplifiedData.cpp:448-459
                        "LoadSdSceneHIP");
        DIContext.sceneVersion++;
    }
};
// LoadSceneCPU::load (Renderer/LoadMethods.cpp:65-73) catches and logs; so does this one.
template <class App> class LoadSceneHIPT : public App::ILoadMethod {
    SceneHIPContextT<App> &DIContext;

  public:
    std::string lastError;
    explicit LoadSceneHIPT(SceneHIPContextT<App> &c) : DIContext(c) {}
    void load() override {
        try {
            std::shared_lock<std::shared_mutex> sceneReadLock(App::oldSceneMutex());
            const OldSceneArrays s = App::oldScene();
            DIContext.check(rt_upload_scene(DIContext.ctx, s.nodes, s.n_nodes, s.root, s.triangles, s.n_triangles, s.spheres, s.n_spheres,
                                            s.materials, s.n_materials, RT_ORDER_LEFT_FIRST),
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
template <int ESTIMATOR, class Context, class App> class TraceHIPImpl : public App::ITraceMethod {
    using Texture2D = typename App::Texture2D;
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
    // Counters of the last frame.  Synchronous frames (several devices sharing a frame by samples): exactly that call's, with
    // its kernel time.  Otherwise: what the device counted between this frame's
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
        rt_camera cam;
        std::memset(&cam, 0, sizeof(cam));
        App::readCamera(DIContext.cam, cam); // read live every call, the UI may have moved it
        rt_options opt;
        std::memset(&opt, 0, sizeof(opt));
        opt.bounce_limit = App::bounceLimit();
        opt.estimator = ESTIMATOR;
        opt.seed = DIContext.seed;
        opt.jitter = DIContext.perturbStrength;
        opt.tile_rank = 0;
        opt.tile_world = 1;
        opt.traversal = (ESTIMATOR == RT_EST_OLD_RECURSIVE && !App::bvhAcceleration()) ? RT_TRAV_BRUTE : RT_TRAV_AUTO;
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
        // frames are computed ahead on one device, and on several when they share a frame by tiles (the default split)
        if (rt_device_count(DIContext.ctx) != 1 && opt.multi_split != RT_SPLIT_TILE) { // synchronous calls
            drop();
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
template <class App> using TraceSdSceneHIPT = TraceHIPImpl<RT_EST_SD_LOOP, SdSceneHIPContextT<App>, App>;
template <class App> using TraceSceneHIPT = TraceHIPImpl<RT_EST_OLD_RECURSIVE, SceneHIPContextT<App>, App>;


} // namespace rthost

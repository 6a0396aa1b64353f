// host_api.cpp -- C entry points of librt_host.so, the C++ host side of the tracer: scene builders
// and the plug-in adaptors of render_plugin.hpp driven the way Renderer::render drives them
// (Renderer/Renderer.cpp:64-102).  These entry points exist for harnesses (pytest, bench.py); the
// tracer's own ABI is include/rt_hip.h.
#include <cstdio>
#include <cstring>
#include <string>

#include "obj_reader.hpp"
#include "old_scene.hpp"
#include "render_plugin.hpp"
#include "sd_scene.hpp"

namespace rthost {
namespace Storage {
SdScene SdScene_;
std::shared_mutex SdSceneMutex;
OldScene OldScene_;
std::shared_mutex OldSceneMutex;
} // namespace Storage
namespace Trace {
int bounceLimit = 4;
}
namespace BVHSettings {
bool toggleBVHAccel = true;
}
} // namespace rthost

using namespace rthost;

namespace {
thread_local std::string g_err;
template <class F> int guarded(F f) {
    try {
        f();
        g_err.clear();
        return 0;
    } catch (std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
std::vector<MeshData> meshesFromSoup(const float *tris24, const uint32_t *mesh_sizes, uint32_t n_meshes) {
    std::vector<MeshData> meshes(n_meshes);
    size_t off = 0;
    for (uint32_t m = 0; m < n_meshes; m++) {
        for (uint32_t i = 0; i < mesh_sizes[m]; i++)
            for (int c = 0; c < 3; c++) {
                SdVertex v;
                std::memcpy(&v, tris24 + (off + i) * 24 + c * 8, sizeof(v));
                meshes[m].indices.push_back((unsigned)meshes[m].vertices.size());
                meshes[m].vertices.push_back(v);
            }
        off += mesh_sizes[m];
    }
    return meshes;
}

struct Harness { // one render mode + scheduler, as Renderer holds them (Renderer/Renderer.hpp:41-60)
    Camera cam;
    std::unique_ptr<IRenderPipeline> pipeline;
    std::unique_ptr<TracerAsync> tracer;
};
} // namespace

extern "C" {

const char *rth_last_error(void) { return g_err.c_str(); }

// ---- flat-array scene ----
void *rth_sd_create(void) { return new SdScene(); }
void rth_sd_destroy(void *s) { delete (SdScene *)s; }
int rth_sd_load_obj(void *s, const char *path) {
    return guarded([&] { ((SdScene *)s)->loadModelFile(path); });
}
int rth_sd_add_model(void *s, const float *tris24, const uint32_t *mesh_sizes, uint32_t n_meshes) {
    return guarded([&] { ((SdScene *)s)->addModel(meshesFromSoup(tris24, mesh_sizes, n_meshes)); });
}
int rth_sd_finalize(void *s) {
    return guarded([&] { ((SdScene *)s)->finalize(); });
}
void rth_sd_counts(void *s, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *root) {
    const SdDataStorage &d = *((SdScene *)s)->pDataStorage;
    *n_nodes = (uint32_t)d.nodes.size();
    *n_tris = (uint32_t)d.triangles.size();
    *root = d.rootIndex;
}
const rt_node *rth_sd_nodes(void *s) { return ((SdScene *)s)->pDataStorage->nodes.data(); }
const rt_tri *rth_sd_tris(void *s) { return ((SdScene *)s)->pDataStorage->triangles.data(); }
void rth_sd_flat(void *s, float *nodes9, float *tris25) {
    std::vector<float> a, b;
    ConvertNodeToFlatStorage(*((SdScene *)s)->pDataStorage, a);
    ConvertTriangleToFlatStorage(*((SdScene *)s)->pDataStorage, b);
    std::memcpy(nodes9, a.data(), a.size() * sizeof(float));
    std::memcpy(tris25, b.data(), b.size() * sizeof(float));
}
// publishes a scene as the process-global Storage::SdScene (main.cpp:89 initialises it there)
int rth_sd_publish(void *s) {
    return guarded([&] {
        std::unique_lock<std::shared_mutex> w(Storage::SdSceneMutex);
        Storage::SdScene_ = *(SdScene *)s;
    });
}

// ---- pointer-based scene ----
void *rth_old_create(void) { return new OldScene(); }
void rth_old_destroy(void *s) { delete (OldScene *)s; }
int rth_old_add_sphere(void *s, const float *center, float radius, uint32_t kind, const float *albedo4, float param) {
    rt_material m;
    m.kind = kind;
    std::memcpy(m.albedo, albedo4, sizeof(m.albedo));
    m.param = param;
    return guarded([&] { ((OldScene *)s)->addSphere(center, radius, m); });
}
int rth_old_add_mesh(void *s, const float *tris24, uint32_t n_tris, uint32_t kind, const float *albedo4, float param) {
    rt_material m;
    m.kind = kind;
    std::memcpy(m.albedo, albedo4, sizeof(m.albedo));
    m.param = param;
    return guarded([&] { ((OldScene *)s)->addMesh(meshesFromSoup(tris24, &n_tris, 1)[0], m); });
}
int rth_old_add_obj(void *s, const char *path, uint32_t kind, const float *albedo4, float param) {
    rt_material m;
    m.kind = kind;
    std::memcpy(m.albedo, albedo4, sizeof(m.albedo));
    m.param = param;
    return guarded([&] {
        for (const MeshData &md : ReadObjFile(path))
            ((OldScene *)s)->addMesh(md, m);
    });
}
int rth_old_build(void *s) {
    return guarded([&] { ((OldScene *)s)->build(); });
}
void rth_old_counts(void *s, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *n_spheres, uint32_t *n_mats, uint32_t *root) {
    const OldScene &o = *(OldScene *)s;
    *n_nodes = (uint32_t)o.nodes.size();
    *n_tris = (uint32_t)o.triangles.size();
    *n_spheres = (uint32_t)o.spheres.size();
    *n_mats = (uint32_t)o.materials.size();
    *root = o.rootIndex;
}
const rt_node *rth_old_nodes(void *s) { return ((OldScene *)s)->nodes.data(); }
const rt_tri *rth_old_tris(void *s) { return ((OldScene *)s)->triangles.data(); }
const rt_sphere *rth_old_spheres(void *s) { return ((OldScene *)s)->spheres.data(); }
const rt_material *rth_old_materials(void *s) { return ((OldScene *)s)->materials.data(); }
int rth_old_publish(void *s) {
    return guarded([&] {
        std::unique_lock<std::shared_mutex> w(Storage::OldSceneMutex);
        Storage::OldScene_ = *(OldScene *)s;
    });
}

// ---- plug-in harness: what Renderer::changeMode / Renderer::render do, without GL ----
// mode 0 = HIP_SdScene, 1 = HIP_Scene
static void *harness_create(int mode, const int *devices, int n_devices, int width, int height, const float *cam10,
                            int bounce_limit, uint32_t seed, float jitter) {
    Harness *h = new Harness();
    int rc = guarded([&] {
        h->cam.focalLength = cam10[0];
        std::memcpy(h->cam.position, cam10 + 1, 12);
        std::memcpy(h->cam.lookAtCenter, cam10 + 4, 12);
        h->cam.width = cam10[7];
        h->cam.height = cam10[8];
        h->cam.aspectRatio = cam10[9];
        Trace::bounceLimit = bounce_limit;
        const std::vector<int> devs(devices, devices + n_devices);
        if (mode == 0) {
            SdSceneHIPContext c(h->cam, devs);
            c.seed = seed;
            c.perturbStrength = jitter;
            h->pipeline = std::make_unique<RenderPipeline<SdSceneHIPContext, LoadSdSceneHIP, TraceSdSceneHIP>>(std::move(c));
        } else {
            SceneHIPContext c(h->cam, devs);
            c.seed = seed;
            c.perturbStrength = jitter;
            h->pipeline = std::make_unique<RenderPipeline<SceneHIPContext, LoadSceneHIP, TraceSceneHIP>>(std::move(c));
        }
        h->tracer = std::make_unique<TracerAsync>(width, height);
    });
    if (rc != 0) {
        delete h;
        return nullptr;
    }
    return h;
}
void *rth_harness_create(int mode, int device, int width, int height, const float *cam10, int bounce_limit,
                         uint32_t seed, float jitter) {
    return harness_create(mode, &device, 1, width, height, cam10, bounce_limit, seed, jitter);
}
// the same render mode over several GPUs (rt_create_multi behind the one context)
void *rth_harness_create_devices(int mode, const int *devices, int n_devices, int width, int height, const float *cam10,
                                 int bounce_limit, uint32_t seed, float jitter) {
    return harness_create(mode, devices, n_devices, width, height, cam10, bounce_limit, seed, jitter);
}
// look-ahead of the trace method on / off (on by default)
void rth_harness_set_look_ahead(void *h, int on) {
    ITraceMethod *m = ((Harness *)h)->pipeline->getTraceMethod();
    if (auto *a = dynamic_cast<TraceSdSceneHIP *>(m))
        a->lookAhead = on != 0;
    if (auto *b = dynamic_cast<TraceSceneHIP *>(m))
        b->lookAhead = on != 0;
}
// rays counted for the last frame (the trace method's lastStats)
uint64_t rth_harness_last_rays(void *h) {
    ITraceMethod *m = ((Harness *)h)->pipeline->getTraceMethod();
    if (auto *a = dynamic_cast<TraceSdSceneHIP *>(m))
        return a->lastStats.rays;
    if (auto *b = dynamic_cast<TraceSceneHIP *>(m))
        return b->lastStats.rays;
    return 0;
}
void rth_harness_destroy(void *h) { delete (Harness *)h; }
int rth_harness_load(void *h) { // uploader->upload(*pipeline->getLoadMethod()), Renderer.cpp:83
    return guarded([&] { ((Harness *)h)->pipeline->getLoadMethod()->load(); });
}
int rth_harness_render(void *h) { // tracer->render(*pipeline->getTraceMethod()), Renderer.cpp:92
    return guarded([&] { ((Harness *)h)->tracer->render(*((Harness *)h)->pipeline->getTraceMethod()); });
}
int rth_harness_reset(void *h) { // RenderState::Dirty, Renderer.cpp:86-90
    return guarded([&] { ((Harness *)h)->tracer->resetSamples(); });
}
int rth_harness_resize(void *h, int w, int hgt) { // Renderer::resize, Renderer.cpp:104-117
    return guarded([&] {
        ((Harness *)h)->tracer->resize(w, hgt);
        ((Harness *)h)->tracer->resetSamples();
    });
}
void rth_set_bvh_accel(int on) { BVHSettings::toggleBVHAccel = on != 0; }
void rth_harness_set_camera_position(void *h, const float *pos3) { std::memcpy(((Harness *)h)->cam.position, pos3, 12); }
int rth_harness_sample_count(void *h) { return ((Harness *)h)->tracer->sampleCount(); }
int rth_harness_output(void *h, float *out, uint32_t *w, uint32_t *hgt) {
    const Texture2D &t = ((Harness *)h)->tracer->output();
    *w = t.Width;
    *hgt = t.Height;
    if (out)
        std::memcpy(out, t.texels.data(), t.texels.size() * sizeof(float));
    return 0;
}

} // extern "C"

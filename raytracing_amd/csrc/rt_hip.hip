// rt_hip.hip -- MI355X (gfx950) path tracer: device kernels and the C ABI of include/rt_hip.h.  One translation unit;
// the device code lives in rt_device_common.hpp, rt_walks.hpp and rt_kernels.hpp, the host-side layout of the scene in
// scene_layout.hpp, all included below.
//
// What runs on the GPU is the per-pixel loop of the reference's CPU trace methods
// (Renderer/TraceMethods.cpp:45-119): camera ray generation (Include/Camera.hpp:33-47), closest-hit
// search over the scene BVH (CPURayTrace/SimplifiedData.cpp:430-462, CPURayTrace/BVH.hpp:124-148),
// slab / Moller-Trumbore / sphere tests (SimplifiedData.cpp:146-161, :269-307, Objects.hpp:43-72),
// the two radiance estimators of CPURayTrace/Trace.cpp and the running-mean accumulation.
//
// Numerics contract: every value is computed with the same IEEE fp32 operations in the same order as
// the reference's scalar glm expressions (dot = (x+y)+z, normalize = v*(1/sqrt(dot)), ...), with no
// fused multiply-add (build with -ffp-contract=off) and correctly rounded divide and sqrt, so hit
// primitives, hit distances and shaded pixels are reproducible bit for bit on a CPU.
//
// Traversal design.  The reference visits nodes in one fixed depth-first order that does not depend on the ray
// (children are never sorted by distance), and its interior boxes only prune: what it returns is decided by the
// leaf boxes, the primitive tests and the order in which leaves are met.  Two node arrays are therefore laid out
// in HBM, both as 32-byte records {box, skip, word} in depth-first pre-order, where "descend" is the next record and
// "prune" is a jump to the record's skip offset, so that no lane needs a traversal stack:
//   * the reference's own tree in the reference's own visiting order -- the walks over it (closest_hit,
//     closest_hit_wave, k_trace) perform exactly the reference's slab and triangle tests, in the same order, with
//     the same running closest distance;
//   * the fast tree (fast_bvh.hpp): a surface-area-heuristic hierarchy of exact unions over the reference's leaf
//     boxes, kept bit for bit.  closest_hit_fast walks it without closest-distance pruning, keeps the smallest
//     (t, position in the reference's order) and so returns the reference's hit from far fewer tests; the rare ray
//     whose winner lies in front of its own leaf box is resolved by replay_ordered.  This is the default.
// The tracer is a wavefront: one launch per bounce level (k_path, or k_refill for scenes larger than the L2s),
// paths of every depth compacted into full waves, per-sample colours folded into the running mean in sample order
// (k_fold).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "rt_hip.h"
#include "fast_bvh.hpp"
#include "fastdiv.hpp"

#define RT_HD __host__ __device__ __forceinline__
#define RT_D __device__ __forceinline__

namespace {

#include "rt_device_common.hpp"
#include "rt_walks.hpp"
#include "rt_kernels.hpp"

} // namespace

// ===================================================================== host side of the C ABI

// Built-in workspace of rt_trace_device_noalloc: one array per GPU, part of the code object, so that entry point never
// has to ask the runtime for memory.  It holds what a chunk of the wavefront tracer needs for at most kScratchMaxItems
// pixel samples: the per-level sub-queue counters, the per-sample colours and the two path queues (sized for the
// largest record, six float4).  In host code the array's name is only a handle for hipGetSymbolAddress; the address
// each device holds it at is looked up once per device (scratch_of) and that pointer is what kernels and memsets get.
constexpr size_t kScratchMaxItems = 256u << 10; // pixel samples per chunk = pixels of the largest frame, in whole 8x8 tiles
constexpr size_t round256(size_t v) { return (v + 255u) & ~(size_t)255u; }
constexpr size_t kLevelWords = (size_t)(kMaxLevels + 1) * kSubQueues * 32;
constexpr uint32_t kSlackBatches = 130;                                     // batches of 64 records a sub-queue holds beyond its share, see queue_records
constexpr size_t kQueueSlack = (size_t)kSubQueues * kSlackBatches * 64;
constexpr size_t kScratchLevelsBytes = round256(kLevelWords * sizeof(unsigned int));
constexpr size_t kScratchColorsBytes = round256(kScratchMaxItems * sizeof(float4));
constexpr size_t kScratchQueueBytes = round256((kScratchMaxItems + kQueueSlack) * 6 * sizeof(float4));
constexpr size_t kScratchDeferredBytes = round256(kScratchMaxItems * sizeof(unsigned long long));
constexpr size_t kScratchBytes = kScratchLevelsBytes + kScratchColorsBytes + kScratchDeferredBytes + 2 * kScratchQueueBytes; // 159 391 744 bytes
static_assert(kScratchBytes == 159391744, "the size stated in include/rt_hip.h");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// What one chunk of the wavefront tracer works in.
struct Workspace {
    float4 *colors = nullptr, *q[2] = {nullptr, nullptr};
    unsigned int *levels = nullptr;   // per level: 64 sub-queue slots of 32 words (count, batch cursor)
    unsigned long long *deferred = nullptr; // per pixel: which samples of the chunk the camera stage queued
    size_t colors_cap = 0, q_cap[2] = {0, 0}; // capacities in float4
    size_t deferred_cap = 0;          // pixels
};

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;
    std::mutex err_mu; // guards `err` alone: rt_last_error may be called while another thread is inside a long call
    std::string err;
    // scene in HBM
    float4 *d_fnodes = nullptr;
    uint32_t n_fnodes = 0;
    float4 *d_nodes = nullptr, *d_tri_pos = nullptr, *d_tri_nrm = nullptr, *d_spheres = nullptr, *d_mats = nullptr;
    uint32_t n_nodes = 0;
    float4 *d_cnodes = nullptr, *d_leaf_boxes = nullptr; // compact fast tree of a flat scene (SceneLayout::cn, lb)
    float4 *d_wnodes = nullptr;                          // ... and its wide form (SceneLayout::wn)
    bool has_wide = false;
    uint32_t n_wide = 0, wide_depth = 1;
    double wide_visits = 0.0; // records of the wide tree a random ray through the root box visits
    bool can_prune = false;
    float prune_c[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // c1[3], 2 c2[3], c2 W + c3 [3]
    float q_origin[3] = {0, 0, 0}, q_step[3] = {1, 1, 1};
    bool has_compact = false;
    double compact_inflation = 1.0;
    double fast_visits = 0.0; // fast-tree records a random ray through the root box meets (flat scenes; 0: not computed)
    bool has_scene = false;
    size_t scene_bytes = 0; // fast-tree records + triangle positions
    bool fast_exact = true; // the uploaded tree's interior boxes contain their children's (see build_layout)
    double expected_visits = 0.0; // sum of node box areas / root box area: nodes a random ray through the root meets
    float root_box[6] = {0, 0, 0, 0, 0, 0}; // box of the fast tree's root record: pMin.xyz, pMax.xyz
    // camera stage: the scene's cut (host copy), and the tile mask of the last (camera, frame size, jitter, scene) it was made for
    std::vector<float> cut_boxes;
    uint8_t *d_cam_mask = nullptr;
    size_t cam_mask_cap = 0;
    struct MaskKey {
        rt_camera cam;
        int W, H;
        float jitter;
        unsigned long long scene;
    } mask_key{};
    bool mask_valid = false, mask_usable = false;
    unsigned long long scene_version = 0;
    rt_camera cam;
    bool has_camera = false;
    rt_options opt;
    // running-mean image owned by the context (rt_trace)
    float4 *d_image = nullptr;
    uint32_t *d_primary = nullptr;
    int img_w = 0, img_h = 0;
    // work queue head and counters
    unsigned int *d_queue = nullptr;
    unsigned long long *d_counters = nullptr;
    bool counted = false; // something has been added to the counters since they were last read
    // one HIP event pair per launch since the last sync (the kernel time is read back in rt_sync)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t pending = 0;
    int num_cus = 0;
    int blocks_per_cu[16] = {0}; // [nested | wavefront per-lane | wavefront wave-sync | wavefront fast][estimator][stats]
    // Two pipes take the chunks of the wavefront tracer in turn, each with its own stream and workspace (per-sample
    // colour buffer, two ping-pong path queues, per-level cursors/counters), so that the latency-bound end of one chunk
    // (deep bounce levels: few rays, long walks) overlaps the throughput-bound start of the next.  The folds into the
    // running mean are chained by events and stay in sample order.
    struct Pipe {
        hipStream_t stream = nullptr;
        Workspace ws;
        hipEvent_t done = nullptr; // recorded after the pipe's last fold
        bool used = false;         // `done` has been recorded at least once
    } pipes[2];
    int next_pipe = 0;
    hipEvent_t ev_start = nullptr;  // orders a call's chunks after what precedes the call on the context's stream
    hipEvent_t last_fold = nullptr; // `done` of the pipe that ran the latest fold
    bool pipelining = false;        // rt_set_pipelining: calls may overlap; rt_join / rt_sync order their results
    bool joined = true;             // the context's stream has been made to wait for both pipes
    void *scratch_ptr = nullptr;    // this device's copy of `scratch` (rt_trace_device_noalloc)
    // rt_trace_begin / rt_trace_end: calls in flight ahead of the caller.  The running mean moves round a ring of three
    // images -- a begun call continues from the image of the call before it and writes the next one -- so that up to two
    // begun calls never touch the image of the last ENDED call (ring[ring_at] == d_image), which rt_trace_discard returns to.
    struct Begun {
        int img;          // ring index the call writes
        float *host;      // where its mean is copied
        hipEvent_t copied;
    };
    float4 *ring[3] = {nullptr, nullptr, nullptr}; // ring[ring_at] is d_image; the other two exist once rt_trace_begin has run
    int ring_at = 0;
    std::deque<Begun> begun;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_call = nullptr;
    int next_copied = 0;
    // rt_create_multi: this context only coordinates; kids[r] is an ordinary context on device r of the list, and
    // `device` / `stream` / `d_image` belong to the first device of the list, where the frame is put together.
    std::vector<rt_ctx *> kids;
    std::vector<float4 *> peer_img;    // root-device landing buffers for the frames of kids 1..n-1
    std::vector<uint32_t *> peer_prim; // ... and for their primary-id images
    std::vector<int> kid_samples;      // sample split: samples folded into each kid's own mean since the last restart
    size_t peer_cap = 0;               // pixels the landing buffers hold
    float4 *multi_frame = nullptr;     // where the frames of calls begun ahead (rt_trace_begin) are put together
    size_t multi_frame_cap = 0;
    int multi_w = 0, multi_h = 0;      // frame size of the run the kids' means belong to
};

namespace {

int fail(rt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) {
        std::lock_guard<std::mutex> g(ctx->err_mu);
        ctx->err = msg;
    }
    return code;
}
#define RT_HIP_CHECK(ctx, expr)                                                                               \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,                   \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
    } while (0)

// Makes the context's device current for the duration of a call and puts the caller's device back afterwards: a host
// application (or torch) that works on another GPU must not find its current device changed by a tracer call.
struct DeviceGuard {
    int prev = -1;
    hipError_t rc;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        rc = prev == dev ? hipSuccess : hipSetDevice(dev);
        if (prev == dev)
            prev = -1; // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};
#define RT_ON_DEVICE(ctx)                                                                                      \
    DeviceGuard device_guard_((ctx)->device);                                                                  \
    RT_HIP_CHECK(ctx, device_guard_.rc)

// Address of `scratch` on each device, looked up once per device with that device current.
std::mutex g_scratch_mu;
std::vector<void *> g_scratch_ptr;
int scratch_of(rt_ctx *ctx, int device, void **out) {
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if ((size_t)device >= g_scratch_ptr.size())
        g_scratch_ptr.resize((size_t)device + 1, nullptr);
    if (g_scratch_ptr[device] == nullptr) {
        void *p = nullptr;
        RT_HIP_CHECK(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (p == nullptr)
            return fail(ctx, RT_ERR_HIP, "hipGetSymbolAddress returned no address for the built-in workspace");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

void free_scene(rt_ctx *c) {
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_fnodes) (void)hipFree(c->d_fnodes);
    c->d_fnodes = nullptr;
    c->n_fnodes = 0;
    if (c->d_tri_pos) (void)hipFree(c->d_tri_pos);
    if (c->d_tri_nrm) (void)hipFree(c->d_tri_nrm);
    if (c->d_spheres) (void)hipFree(c->d_spheres);
    if (c->d_mats) (void)hipFree(c->d_mats);
    if (c->d_cnodes) (void)hipFree(c->d_cnodes);
    if (c->d_leaf_boxes) (void)hipFree(c->d_leaf_boxes);
    if (c->d_wnodes) (void)hipFree(c->d_wnodes);
    c->d_cnodes = c->d_leaf_boxes = c->d_wnodes = nullptr;
    c->has_compact = c->has_wide = c->can_prune = false;
    c->n_wide = 0;
    c->wide_visits = 0.0;
    c->fast_visits = 0.0;
    c->d_nodes = c->d_tri_pos = c->d_tri_nrm = c->d_spheres = c->d_mats = nullptr;
    c->n_nodes = 0;
    c->has_scene = false;
}

void free_workspace(Workspace &ws) {
    if (ws.colors) (void)hipFree(ws.colors);
    if (ws.q[0]) (void)hipFree(ws.q[0]);
    if (ws.q[1]) (void)hipFree(ws.q[1]);
    if (ws.deferred) (void)hipFree(ws.deferred);
    ws.colors = ws.q[0] = ws.q[1] = nullptr;
    ws.deferred = nullptr;
    ws.colors_cap = ws.q_cap[0] = ws.q_cap[1] = ws.deferred_cap = 0;
}

inline float u2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

template <class T> int upload(rt_ctx *ctx, float4 **dst, const std::vector<T> &src) {
    size_t bytes = src.size() * sizeof(T);
    if (bytes == 0)
        bytes = 16;
    RT_HIP_CHECK(ctx, hipMalloc((void **)dst, bytes));
    if (!src.empty())
        RT_HIP_CHECK(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

template <int KERNEL, bool OLD, bool STATS> int occupancy(int *out) {
    int n = 0;
    if (KERNEL != 0) {
        // k_path deals its batches round-robin over whatever grid it is given, so the grid only has to be large
        // enough to fill the chip: 8 blocks of 4 waves per CU is the hardware's limit of 8 waves per SIMD.  Blocks
        // that are not resident at once simply start as others finish.
        *out = 8;
        return 0;
    }
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace<OLD, STATS>, 256, 0);
    if (e != hipSuccess || n < 1)
        n = 1;
    if (n > 8)
        n = 8;
    *out = n;
    return 0;
}

// The context's stream waits (stream-level, the host does not) for everything the pipes have been given.
int join_pipes(rt_ctx *ctx) {
    if (!ctx->joined) {
        for (auto &pp : ctx->pipes)
            if (pp.used)
                RT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, pp.done, 0));
        ctx->joined = true;
    }
    return RT_OK;
}
int sync_all(rt_ctx *ctx) {
    for (auto &pp : ctx->pipes)
        if (pp.stream)
            RT_HIP_CHECK(ctx, hipStreamSynchronize(pp.stream));
    RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->joined = true;
    return RT_OK;
}

// The tracer's tuning constants; each can be overridden from the environment for measurements (RT_SLICE, RT_TAIL_RAYS,
// RT_EXTRA_ROUNDS, RT_CHUNK_SAMPLES, RT_COMPACT, RT_REFILL_TAIL), read once per process.  Results do not depend on them.
// (RT_DEBUG_LAYOUT=1 prints what an upload measured about the scene's fast tree to stderr.)
struct Tuning {
    uint32_t slice = 0, tail_rays = kTailRays, slice_tail_rays = 0; // slice, slice_tail_rays 0: chosen per scene
    int extra_rounds = kSliceExtraRounds, chunk_samples = 0; // chunk_samples 0: by the size of the share (chunk_samples())
    int compact = 1; // RT_COMPACT=0: no walk takes the compact tree; 2: traces that collect statistics take it too
    int refill_tail = 1; // RT_REFILL_TAIL=0: k_sliced finishes its last round itself
    int slice_quorum = kSliceQuorum; // RT_SLICE_QUORUM
    int wave_camera = 1; // RT_WAVE_CAMERA=0: camera rays walk lane by lane like every other ray
    int wide = 1;        // RT_WIDE=0: no walk takes the wide tree; 2: traces that collect statistics take it too
    int prune = 1;       // RT_PRUNE=0: the wide walk keeps no distance limit
    int wide_burst = 0;  // RT_WIDE_BURST=<n>: nodes per burst of the wide walk (0: by the scene, see kWideBurst)
    int tile_mask = 1;   // RT_TILE_MASK=0: the camera stage tests every tile inside the projection of the scene's box
    int lds_pad = 0; // RT_LDS_PAD: bytes of unused LDS added to every path-kernel block (fewer blocks per CU: occupancy experiments)
};
const Tuning &tuning() {
    static const Tuning t = [] {
        Tuning v;
        auto env = [](const char *name, long lo, long hi, long dflt) {
            const char *e = std::getenv(name);
            if (!e || !*e)
                return dflt;
            const long x = std::strtol(e, nullptr, 10);
            return x < lo ? lo : x > hi ? hi : x;
        };
        v.slice = (uint32_t)env("RT_SLICE", 1, 1 << 20, 0); // 0: chosen per scene (make_plan)
        v.tail_rays = (uint32_t)env("RT_TAIL_RAYS", 0, 1 << 30, kTailRays);
        v.slice_tail_rays = (uint32_t)env("RT_TAIL_RAYS", 0, 1 << 30, 0);
        v.extra_rounds = (int)env("RT_EXTRA_ROUNDS", -32, 32, kSliceExtraRounds);
        v.chunk_samples = (int)env("RT_CHUNK_SAMPLES", 1, kMaxChunkSamples, 0);
        v.compact = (int)env("RT_COMPACT", 0, 2, 1);
        v.refill_tail = (int)env("RT_REFILL_TAIL", 0, 1, 1);
        v.slice_quorum = (int)env("RT_SLICE_QUORUM", 0, 64, kSliceQuorum);
        v.lds_pad = (int)env("RT_LDS_PAD", 0, 40 << 10, 0);
        v.wave_camera = (int)env("RT_WAVE_CAMERA", 0, 1, 1);
        v.wide = (int)env("RT_WIDE", 0, 2, 1);
        v.prune = (int)env("RT_PRUNE", 0, 1, 1);
        v.wide_burst = (int)env("RT_WIDE_BURST", 0, 64, 0);
        v.tile_mask = (int)env("RT_TILE_MASK", 0, 1, 1);
        return v;
    }();
    return t;
}

// Everything about a trace call that does not depend on the chunk: kernel parameters, which kernel, how many launches.
struct Plan {
    KParams P;
    int variant = 0, walk = 0, levels = 1;
    bool nested = false, sliced = false, queued = false, refill = false;
    bool camera = false;        // the chunk starts with k_camera, the path kernels run at levels 1 .. levels
    uint32_t full_grid = 1;
    bool compact = false;       // the sliced walk steps through the compact tree
    bool wide = false;          // ... and every ray but the fresh camera rays of scenes within the L2s through the wide tree
    size_t rec4 = 4;            // float4 per queue record
    size_t npix = 0, items_per_sample = 0;
    int W = 0, H = 0, first = 1, count = 0;
};

// Part of the frame covered by the projection of the scene's root box (the bounding rectangle of its eight corners in
// the camera's uv coordinates, Camera.hpp:33-47), 1 when a corner lies behind the camera.
// Which 8x8 tiles of a W x H frame a box can be seen in, jitter included.
// In camera space a ray's direction is v + j: v the unit view direction of its pixel -- slopes (v.x / v.z, v.y / v.z), which
// are linear in the pixel's (u, v) = (x / W, y / H) (Camera::getRayDirction, Include/Camera.hpp:33-47) --, j the jitter
// turned into camera space, |j| <= js sqrt(3) =: e.  The ray can meet the box only if its slopes lie in the projection of
// the box, which -- all eight corners in front of the camera, or there is no answer -- lies in the bounding rectangle of
// the corners' slopes.  The jitter moves a slope by |j.x v.z - v.x j.z| / (v.z (v.z + j.z)) <= 2 e / (z0 (z0 - e)), z0 =
// focal / |(w/2, h/2, focal)| the smallest v.z of the frame: 7.7 pixels at 1920 x 1080 with the reference's camera and its
// 0.001.  The rectangle is widened by that (plus 5 % and half a pixel; the slab test's own rounding is eight orders of
// magnitude below a pixel) and rounded outwards to whole tiles: no camera ray of a tile outside it meets the box.
struct TileProjection {
    double pos[3], x[3], y[3], z[3], f = 0, w = 0, h = 0, mx = 0, my = 0;
    int W = 0, H = 0, tiles_x = 0, tiles_y = 0;
    bool ok = false;
    TileProjection(const rt_camera &c, int W_, int H_, float jitter) : W(W_), H(H_), tiles_x((W_ + 7) / 8), tiles_y((H_ + 7) / 8) {
        const f3 p = mk3(c.position[0], c.position[1], c.position[2]);
        const f3 zz = normalize3(mk3(c.look_at[0], c.look_at[1], c.look_at[2]) - p);
        const f3 xx = normalize3(cross3(mk3(0.f, 1.f, 0.f), zz));
        const f3 yy = cross3(zz, xx);
        const float *src[4] = {&p.x, &xx.x, &yy.x, &zz.x};
        double *dst[4] = {pos, x, y, z};
        for (int a = 0; a < 4; a++)
            for (int k = 0; k < 3; k++)
                dst[a][k] = src[a][k];
        f = c.focal_length;
        w = c.width;
        h = c.height;
        if (!(f > 0.0) || !(w > 0.0) || !(h > 0.0) || !(jitter >= 0.0f))
            return;
        for (int k = 0; k < 3; k++)
            if (!(x[k] == x[k]) || !(y[k] == y[k]) || !(z[k] == z[k]))
                return;
        const double z0 = f / std::sqrt(w * w / 4 + h * h / 4 + f * f), e = (double)jitter * 1.7320508075688772 * 1.0001;
        if (!(e < z0 / 2))
            return;
        const double slope = 2.0 * e / (z0 * (z0 - e));
        mx = 1.05 * W * f * slope / w + 0.5;
        my = 1.05 * H * f * slope / h + 0.5;
        ok = true;
    }
    // tiles [out[0], out[2]] x [out[1], out[3]] (possibly outside the frame, possibly empty); false: no answer for this box
    bool tiles_of(const float box[6], int out[4]) const {
        if (!ok)
            return false;
        double u0 = 1e30, u1 = -1e30, v0 = 1e30, v1 = -1e30;
        for (int k = 0; k < 8; k++) {
            const double r[3] = {(double)box[(k & 1) ? 3 : 0] - pos[0], (double)box[(k & 2) ? 4 : 1] - pos[1], (double)box[(k & 4) ? 5 : 2] - pos[2]};
            const double cz = r[0] * z[0] + r[1] * z[1] + r[2] * z[2];
            if (!(cz > 1e-4 * (std::fabs(r[0]) + std::fabs(r[1]) + std::fabs(r[2])) + 1e-30))
                return false; // a corner beside or behind the camera (or not a number)
            const double cx = r[0] * x[0] + r[1] * x[1] + r[2] * x[2], cy = r[0] * y[0] + r[1] * y[1] + r[2] * y[2];
            const double u = (-cx * f / cz + w / 2) / w, v = (cy * f / cz + h / 2) / h;
            if (!(std::fabs(u) < 1e6) || !(std::fabs(v) < 1e6))
                return false;
            u0 = u < u0 ? u : u0;
            u1 = u > u1 ? u : u1;
            v0 = v < v0 ? v : v0;
            v1 = v > v1 ? v : v1;
        }
        // pixel x has u = x / W (CPUImageData::uvAt, Renderer/TraceMethods.hpp:30)
        auto tile_of = [](double p, int n_tiles, int extra) {
            const double t = std::floor(p / 8.0) + extra;
            return (int)(t < -1.0 ? -1.0 : t > (double)n_tiles ? (double)n_tiles : t);
        };
        out[0] = tile_of(std::floor(u0 * W - mx), tiles_x, 0);
        out[1] = tile_of(std::floor(v0 * H - my), tiles_y, 0);
        out[2] = tile_of(std::ceil(u1 * W + mx), tiles_x, 0);
        out[3] = tile_of(std::ceil(v1 * H + my), tiles_y, 0);
        return true;
    }
};

// The tiles outside which no camera ray can meet the scene's root box (the whole frame when there is no answer).
void root_box_tiles(const rt_ctx *ctx, int W, int H, float jitter, int out[4]) {
    const TileProjection tp(ctx->cam, W, H, jitter);
    if (!tp.tiles_of(ctx->root_box, out)) {
        out[0] = out[1] = 0;
        out[2] = tp.tiles_x - 1;
        out[3] = tp.tiles_y - 1;
    }
}

// One byte per tile of the frame: 1 where some box of the scene's cut (SceneLayout::cut_boxes) can be seen, 0 where no
// camera ray can meet any of them -- hence no leaf box: such a tile's samples are sky whatever the walk would do.  False
// when there is no answer (a box beside or behind the camera, a box that is not finite, no cut): every tile is tested.
bool camera_tile_mask(const rt_ctx *ctx, int W, int H, float jitter, std::vector<uint8_t> &mask) {
    const TileProjection tp(ctx->cam, W, H, jitter);
    if (!tp.ok || ctx->cut_boxes.empty())
        return false;
    mask.assign((size_t)tp.tiles_x * tp.tiles_y, 0);
    for (size_t b = 0; b + 6 <= ctx->cut_boxes.size(); b += 6) {
        int t[4];
        if (!tp.tiles_of(&ctx->cut_boxes[b], t))
            return false;
        const int x0 = t[0] < 0 ? 0 : t[0], y0 = t[1] < 0 ? 0 : t[1], x1 = t[2] >= tp.tiles_x ? tp.tiles_x - 1 : t[2],
                  y1 = t[3] >= tp.tiles_y ? tp.tiles_y - 1 : t[3];
        for (int y = y0; y <= y1; y++)
            if (x1 >= x0)
                std::memset(&mask[(size_t)y * tp.tiles_x + x0], 1, (size_t)(x1 - x0 + 1));
    }
    return true;
}

double root_box_coverage(const rt_ctx *ctx) {
    const rt_camera &c = ctx->cam;
    const f3 pos = mk3(c.position[0], c.position[1], c.position[2]);
    const f3 z = normalize3(mk3(c.look_at[0], c.look_at[1], c.look_at[2]) - pos);
    const f3 x = normalize3(cross3(mk3(0.f, 1.f, 0.f), z));
    const f3 y = cross3(z, x);
    double u0 = 1e30, u1 = -1e30, v0 = 1e30, v1 = -1e30;
    for (int k = 0; k < 8; k++) {
        const f3 r = mk3(ctx->root_box[(k & 1) ? 3 : 0], ctx->root_box[(k & 2) ? 4 : 1], ctx->root_box[(k & 4) ? 5 : 2]) - pos;
        const double cz = dot3(r, z);
        if (!(cz > 1e-6))
            return 1.0;
        const double u = (-(double)dot3(r, x) * c.focal_length / cz + c.width / 2) / c.width;
        const double v = ((double)dot3(r, y) * c.focal_length / cz + c.height / 2) / c.height;
        if (!(u == u) || !(v == v))
            return 1.0;
        u0 = u < u0 ? u : u0;
        u1 = u > u1 ? u : u1;
        v0 = v < v0 ? v : v0;
        v1 = v > v1 ? v : v1;
    }
    auto clip = [](double a) { return a < 0.0 ? 0.0 : a > 1.0 ? 1.0 : a; };
    return (clip(u1) - clip(u0)) * (clip(v1) - clip(v0));
}

// Queue records a chunk of `items` pixel samples needs: a sub-queue of the next level receives the continuing paths
// (at most 64) of every 64th batch of level 0, or of every 64th batch of each of this level's 64 sub-queues; no level
// has more batches than level 0 has plus one partial batch per sub-queue, so (n_batches / 64 + 66) * 64 records per
// sub-queue suffice for the path kernels.  The camera stage appends a whole (tile, 64 samples) unit -- up to 64 batches --
// at a time and deals the units round-robin, full groups first and the partial last group after them: a sub-queue gets at
// most one unit of each kind beyond its share, 2 * 64 batches, hence kSlackBatches = 130.
size_t queue_records(size_t items) { return items + kQueueSlack; }

int make_plan(rt_ctx *ctx, int W, int H, int first, int count, float4 *d_img, uint32_t *d_prim, Plan &pl, const float4 *d_img_in = nullptr,
              bool may_allocate = true) {
    if (!ctx->has_scene)
        return fail(ctx, RT_ERR_NO_SCENE, "Scene is not loaded.");
    if (!ctx->has_camera)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera not set");
    if (W <= 0 || H <= 0 || first < 1 || count < 0 || (long long)W * H > 0x7FFFFFFFll)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad width/height/sample range");
    if (d_img == nullptr)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null image");
    const rt_options &o = ctx->opt;
    KParams &P = pl.P;
    P.nodes = ctx->d_nodes;
    P.fnodes = ctx->d_fnodes;
    P.n_fnodes = ctx->n_fnodes;
    P.tri_pos = ctx->d_tri_pos;
    P.tri_nrm = ctx->d_tri_nrm;
    P.spheres = ctx->d_spheres;
    P.mats = ctx->d_mats;
    P.image = d_img;
    P.image_in = d_img_in ? d_img_in : d_img;
    P.primary = d_prim;
    P.queue = ctx->d_queue;
    P.counters = ctx->d_counters;
    P.n_nodes = ctx->n_nodes;
    P.W = W;
    P.H = H;
    P.tiles_x = (W + 7) / 8;
    P.div_tiles_x = make_fastdiv((uint32_t)P.tiles_x);
    P.div_per_tile = make_fastdiv(64u);
    P.seed_hash = pcg(o.seed);
    const uint32_t tiles = (uint32_t)P.tiles_x * (uint32_t)((H + 7) / 8);
    P.tile_rank = o.tile_rank;
    P.tile_world = o.tile_world;
    P.n_my_tiles = tiles > (uint32_t)o.tile_rank ? (tiles - (uint32_t)o.tile_rank + (uint32_t)o.tile_world - 1) / (uint32_t)o.tile_world : 0u;
    P.first_sample = first;
    P.n_samples = count;
    P.bounce_limit = o.bounce_limit;
    P.seed = o.seed;
    P.sample_offset = o.sample_offset;
    P.jitter = o.jitter;
    // camera basis of Camera::getRayDirction (Include/Camera.hpp:40-44); it is the same for every pixel
    const rt_camera &c = ctx->cam;
    f3 pos = mk3(c.position[0], c.position[1], c.position[2]);
    f3 look = mk3(c.look_at[0], c.look_at[1], c.look_at[2]);
    f3 z = normalize3(look - pos);
    f3 x = normalize3(cross3(mk3(0.f, 1.f, 0.f), z));
    f3 y = cross3(z, x);
    P.cam_pos = pos;
    P.bx = x;
    P.by = y;
    P.bz = z;
    P.cam_w = c.width;
    P.cam_h = c.height;
    P.focal = c.focal_length;

    const bool old = o.estimator == RT_EST_OLD_RECURSIVE;
    const bool st = o.collect_stats != 0;
    int trav = o.traversal;
    if (trav == RT_TRAV_AUTO)
        trav = ctx->fast_exact ? RT_TRAV_FAST : RT_TRAV_REFERENCE;
    if ((trav == RT_TRAV_FAST || trav == RT_TRAV_FAST_REPLAY || trav == RT_TRAV_FAST_REFILL || trav == RT_TRAV_FAST_QUEUED || trav == RT_TRAV_FAST_INLINE || trav == RT_TRAV_FAST_SLICED) && !ctx->fast_exact)
        return fail(ctx, RT_ERR_BAD_SCENE, "RT_TRAV_FAST needs a tree whose interior boxes contain their children's boxes");
    if (trav == RT_TRAV_REFERENCE)
        trav = ctx->expected_visits >= 512.0 ? RT_TRAV_WAVE : RT_TRAV_LANE;
    P.force_replay = trav == RT_TRAV_FAST_REPLAY ? 1 : 0;
    // RT_TRAV_FAST: the loop estimator's scenes are walked in slices while a round holds more rays than the re-fill kernel
    // finishes well in one launch (P.tail_rays), then that kernel takes everything that is left: slices bound what the
    // longest walks cost while the chip is full, lane re-fill keeps the lanes busy once it is not.  Against re-fill alone
    // 8 to 25 % faster on every flat scene measured (bare meshes, fields, meshes on a ground quad, 18 k to 9.8 M triangles).
    // The recursive estimator's scenes (spheres, mesh objects): lane re-fill when the fast tree and triangle positions
    // exceed the L2s, k_path with the noted primitive tests otherwise.
    pl.refill = trav == RT_TRAV_FAST_REFILL || (trav == RT_TRAV_FAST && old && ctx->scene_bytes > kRefillSceneBytes);
    pl.sliced = trav == RT_TRAV_FAST_SLICED || (trav == RT_TRAV_FAST && !old && !pl.refill);
    pl.queued = trav == RT_TRAV_FAST_QUEUED || (trav == RT_TRAV_FAST && !pl.refill && !pl.sliced);
    if (trav == RT_TRAV_FAST_QUEUED || trav == RT_TRAV_FAST_INLINE || trav == RT_TRAV_FAST_SLICED || trav == RT_TRAV_FAST_REFILL)
        trav = RT_TRAV_FAST;
    pl.walk = trav == RT_TRAV_WAVE ? WALK_WAVE : (trav == RT_TRAV_FAST || trav == RT_TRAV_FAST_REPLAY) ? WALK_FAST
              : trav == RT_TRAV_BRUTE                                                               ? WALK_BRUTE
                                                                                                    : WALK_LANE;
    P.wave_min_rays = o.traversal == RT_TRAV_WAVE ? 0u : 65536u;
    pl.nested = trav == RT_TRAV_NESTED;
    pl.variant = (pl.nested ? 0 : 4) + (old ? 2 : 0) + (st ? 1 : 0);
    pl.full_grid = (uint32_t)ctx->num_cus * (uint32_t)ctx->blocks_per_cu[pl.variant];
    pl.npix = (size_t)W * (size_t)H;
    pl.items_per_sample = (size_t)P.n_my_tiles * 64;
    pl.levels = old ? o.bounce_limit + 1 : (o.bounce_limit > 1 ? o.bounce_limit : 1);
    if (pl.sliced) { // rounds, not bounce levels: a walk may take several; the last launch finishes whatever is left
        pl.levels += tuning().extra_rounds;
        if (pl.levels > kMaxLevels - 1)
            pl.levels = kMaxLevels - 1;
        if (pl.levels < 2)
            pl.levels = 2;
    }
    // Camera stage: worth it when most camera rays miss the scene altogether -- judged by the part of the frame the root
    // box's projection covers -- and possible when the walk is the fast tree's (the stage tests that tree's root record).
    pl.camera = false;
    if (pl.walk == WALK_FAST && !(!old && o.bounce_limit <= 0) && o.camera_stage != RT_CAMERA_FUSED)
        pl.camera = o.camera_stage == RT_CAMERA_SEPARATE || root_box_coverage(ctx) < 0.5;
    if (pl.camera && pl.levels + 1 > kMaxLevels - 1)
        pl.camera = false;
    P.camera_stage = pl.camera ? 1 : 0;
    {
        int tiles[4];
        root_box_tiles(ctx, W, H, o.jitter, tiles);
        P.cam_tx0 = tiles[0];
        P.cam_ty0 = tiles[1];
        P.cam_tx1 = tiles[2];
        P.cam_ty1 = tiles[3];
    }
    // ... and, inside that rectangle, the tiles that can see a box of the scene's cut.  The mask is made again only when the
    // camera, the frame size, the jitter or the scene has changed; launches still in flight read the old one, so the
    // context's stream first waits for them (a stream-level wait) and then takes the copy.
    P.cam_mask = nullptr;
    // (a trace that counts the exact walk's tests leaves the mask out: the tiles it spares are tiles whose rays that walk looks at)
    if (pl.camera && may_allocate && tuning().tile_mask != 0 && (!st || o.collect_stats == 2)) {
        rt_ctx::MaskKey key{};
        key.cam = ctx->cam;
        key.W = W;
        key.H = H;
        key.jitter = o.jitter;
        key.scene = ctx->scene_version;
        if (!ctx->mask_valid || std::memcmp(&key, &ctx->mask_key, sizeof(key)) != 0) {
            std::vector<uint8_t> mask;
            ctx->mask_usable = camera_tile_mask(ctx, W, H, o.jitter, mask);
            if (ctx->mask_usable) {
                const int jrc = join_pipes(ctx);
                if (jrc != RT_OK)
                    return jrc;
                if (mask.size() > ctx->cam_mask_cap) {
                    RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                    if (ctx->d_cam_mask) (void)hipFree(ctx->d_cam_mask);
                    ctx->d_cam_mask = nullptr;
                    ctx->cam_mask_cap = 0;
                    RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_cam_mask, mask.size()));
                    ctx->cam_mask_cap = mask.size();
                }
                RT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_cam_mask, mask.data(), mask.size(), hipMemcpyHostToDevice, ctx->stream));
                RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // (the copy reads a buffer that is about to go away)
            }
            std::memset(&ctx->mask_key, 0, sizeof(ctx->mask_key));
            ctx->mask_key = key;
            ctx->mask_valid = true;
        }
        if (ctx->mask_usable)
            P.cam_mask = ctx->d_cam_mask;
    }
    P.deferred = nullptr;
    // Slice length: 32 ... 192 records swept on bare meshes, fields and meshes on a ground quad (128 is best or level with
    // the best everywhere, now that a wave ends its slice when 16 or fewer of its lanes still walk).
    P.slice = tuning().slice != 0 ? tuning().slice : kSlice;
    // Rounds of the sliced walk that the re-fill kernel finishes: up to 12 Mi rays, 24 Mi in scenes beyond the L2s (on the
    // 18 520-triangle dragon that is everything after the first round; the 9.8 M-triangle field gains another 5 % at 24 Mi).
    P.tail_rays = !(pl.sliced && tuning().refill_tail != 0) ? tuning().tail_rays
                  : tuning().slice_tail_rays != 0           ? tuning().slice_tail_rays
                  : ctx->scene_bytes > kRefillSceneBytes    ? kSliceTailRaysBig
                                                            : kSliceTailRays;
    pl.rec4 = pl.sliced ? 6 : 4;
    // Compact tree: when the scene has one and every ray of the call starts close enough to it for the grid arithmetic's
    // margin (compact_walk_from): bounce rays start on the geometry, the camera within 32 grid extents per axis.
    P.cnodes = (const uint4 *)ctx->d_cnodes;
    P.wnodes = (const uint4 *)ctx->d_wnodes;
    P.wide_depth = ctx->wide_depth;
    P.wide_burst = tuning().wide_burst != 0                                                      ? tuning().wide_burst
                   : ctx->wide_visits < kWideShortVisits && ctx->scene_bytes <= kRefillSceneBytes ? kWideBurstShort
                   : ctx->wide_visits >= kWideLongVisits                                           ? kWideBurstLong
                                                                                                   : kWideBurst;
    P.prune = ctx->can_prune && tuning().prune != 0 ? 1 : 0;
    for (int k = 0; k < 3; k++) {
        P.prune_c1[k] = ctx->prune_c[k];
        P.prune_c2x2[k] = ctx->prune_c[3 + k];
        P.prune_c3[k] = ctx->prune_c[6 + k];
    }
    P.leaf_boxes = ctx->d_leaf_boxes;
    P.q_origin = mk3(ctx->q_origin[0], ctx->q_origin[1], ctx->q_origin[2]);
    P.q_step = mk3(ctx->q_step[0], ctx->q_step[1], ctx->q_step[2]);
    // A trace that collects statistics walks the exact records: its counters then describe the walk the other kernels and
    // the CPU restatement make, test for test (the compact walk makes more box tests, with the same outcome).
    pl.compact = pl.walk == WALK_FAST && (pl.sliced || pl.refill) && !old && (!st || tuning().compact == 2 || o.collect_stats == 2) && ctx->has_compact &&
                 tuning().compact != 0;
    for (int k = 0; k < 3 && pl.compact; k++) {
        const float ext = ctx->q_step[k] * 65536.0f, rel = c.position[k] - ctx->q_origin[k];
        if (!(rel > -32.0f * ext && rel < 33.0f * ext))
            pl.compact = false;
    }
    // Wide tree: the default walk of flat scenes (RT_TRAV_FAST; the variants that name a kernel keep that kernel's walk).
    // A chunk is then the camera stage (or not), one launch of k_sliced in which the batches of fresh camera rays are walked
    // wave-synchronously to their ends (scenes within the L2s), and ONE launch of the re-fill kernel that follows every
    // path to its end over the wide tree.
    pl.wide = pl.compact && ctx->has_wide && o.traversal != RT_TRAV_FAST_SLICED && o.traversal != RT_TRAV_FAST_REFILL && o.traversal != RT_TRAV_FAST_REPLAY &&
              tuning().wide != 0 && (!st || tuning().wide == 2 || o.collect_stats == 2);
    P.slice_quorum = (uint32_t)tuning().slice_quorum;
    // scenes within the L2s only: beyond them the scalar path, one request per wave at a time, waits far longer per record than
    // the 64 requests in flight of the lane-by-lane walk (907 k-triangle pair +19 %, 1.19 M-triangle field +8 %)
    // (which also keeps every triangle slot within the 25 bits of a job word: the wave walk notes all its leaves)
    P.wave_camera = tuning().wave_camera != 0 && ctx->scene_bytes <= kRefillSceneBytes ? 1 : 0;
    P.compact = !pl.compact ? 0 : ctx->scene_bytes > kRefillSceneBytes ? 2 : 1; // 2: two records per trip to memory
    if (pl.wide) {
        pl.levels = 2;
        // the wave-synchronous camera round pays only on small trees (2 426-triangle dragon, 700 wide records: 10 % faster
        // with it; 18 520 triangles, 5 472 records: 5 % slower, and its queue of path records is most of the workspace)
        if (ctx->n_wide > kWideWaveMaxNodes)
            P.wave_camera = 0;
    }
    P.npix = (uint32_t)pl.npix;
    P.cpix = (uint32_t)pl.items_per_sample;
    pl.W = W;
    pl.H = H;
    pl.first = first;
    pl.count = count;
    return RT_OK;
}

// Samples per chunk: as many as `max_items` pixel samples allow (the tail of deep bounces is paid once per chunk), and
// few enough that the colour slot of every sample (sample in chunk * slots per sample + slot, pixel_of_slot) stays a 32-bit number.
// A chunk holds up to 64 samples of a whole 1080p frame; a share of a frame (tile split over several GPUs) or a small
// frame gets as many samples more as keep the chunk that size, up to 512, so that its launches stay as large -- and as
// efficient -- as the whole frame's (measured: an eighth of the 1080p frame in chunks of 64 samples takes 10.0 ms per
// 512 spp against 44.3 for the whole frame, 5.6 ms in one chunk of 512).
int chunk_samples(const Plan &pl, size_t max_items) {
    int cap = tuning().chunk_samples;
    if (cap == 0) {
        cap = kChunkSamples;
        if (pl.items_per_sample > 0) {
            const size_t fit = kChunkItems / pl.items_per_sample / 64 * 64;
            if (fit > (size_t)cap)
                cap = fit > (size_t)kMaxChunkSamples ? kMaxChunkSamples : (int)fit;
        }
    }
    int S = pl.count < cap ? pl.count : cap;
    if (pl.items_per_sample > 0 && (size_t)S * pl.items_per_sample > max_items)
        S = (int)(max_items / pl.items_per_sample);
    if (pl.items_per_sample > 0 && (size_t)S * pl.items_per_sample > 0xFFFFFFFFull)
        S = (int)(0xFFFFFFFFull / pl.items_per_sample);
    if (S < 1)
        S = 1;
    return S;
}

int launch_nested(rt_ctx *ctx, const Plan &pl, hipStream_t stream) {
    const KParams &P = pl.P;
    RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue, 0, 64, stream));
    uint32_t blocks = pl.full_grid;
    if (blocks > (P.n_my_tiles + 3) / 4)
        blocks = (P.n_my_tiles + 3) / 4;
    if (blocks < 1)
        blocks = 1;
    dim3 grid(blocks), block(256);
    switch (pl.variant) {
    case 0: hipLaunchKernelGGL((k_trace<false, false>), grid, block, 0, stream, P); break;
    case 1: hipLaunchKernelGGL((k_trace<false, true>), grid, block, 0, stream, P); break;
    case 2: hipLaunchKernelGGL((k_trace<true, false>), grid, block, 0, stream, P); break;
    default: hipLaunchKernelGGL((k_trace<true, true>), grid, block, 0, stream, P); break;
    }
    RT_HIP_CHECK(ctx, hipGetLastError());
    return RT_OK;
}

// One chunk of the wavefront tracer on `stream`, in workspace `ws`: samples pl.first + done .. + s_now - 1 of every owned
// pixel, one launch per bounce level (or round), their colours left in ws.colors.  The caller folds them.
int launch_chunk(rt_ctx *ctx, const Plan &pl, const Workspace &ws, hipStream_t stream, int done, int s_now, KParams &P) {
    P = pl.P;
    if (done > 0)
        P.image_in = P.image; // later chunks continue from what the chunk before them folded
    const int first_level = pl.camera ? 1 : 0;
    int levels = pl.levels + first_level;
    P.n_items = (uint32_t)((size_t)s_now * pl.items_per_sample);
    // A chunk of the sliced walk that cannot fill the chip even in its first round (no round holds more rays than the
    // chunk has samples) is one launch of the re-fill kernel in tail mode: an interactive 1080p frame of one sample per
    // pixel is camera stage, that launch and the fold instead of nineteen launches, sixteen of which find nothing to do.
    if (pl.walk == WALK_FAST && pl.sliced && tuning().refill_tail != 0 && P.n_items <= P.tail_rays)
        levels = first_level + 1;
    RT_HIP_CHECK(ctx, hipMemsetAsync(ws.levels, 0, (size_t)(levels + 1) * kSubQueues * 32 * sizeof(unsigned int), stream));
    P.colors = ws.colors;
    P.deferred = ws.deferred;
    P.S = s_now;
    P.div_per_tile = make_fastdiv(64u * (uint32_t)s_now);
    P.chunk_first = pl.first + done;
    P.sub_cap = (((P.n_items + 63u) / 64u) / (uint32_t)kSubQueues + kSlackBatches) * 64u;
    const bool sliced = pl.sliced, queued = pl.queued, refill = pl.refill;
    const bool refill_tail = tuning().refill_tail != 0; // the sliced walk's last round is finished with lane re-fill
    const int walk = pl.walk;
    const unsigned lds_pad = (unsigned)tuning().lds_pad;
    dim3 block(256);
    if (pl.camera) { // level 0: camera rays, root slab test; what passes goes to level 1's queue
        P.level = 0;
        P.tail = 0;
        P.q_in = ws.q[0];
        P.q_out = ws.q[1];
        P.sub_in = ws.levels;
        P.sub_cursor = ws.levels + 1;
        P.sub_out = ws.levels + (size_t)kSubQueues * 32;
        // one (tile, group of 64 samples) per wave: they cost the same, and waves that start as others finish keep the chip full to the end
        const unsigned long long units = (unsigned long long)P.n_my_tiles * (unsigned long long)((s_now + 63) / 64);
        uint32_t blocks = (uint32_t)((units + 3ull) / 4ull < 0x7FFFFFFFull ? (units + 3ull) / 4ull : 0x7FFFFFFFull);
        if (blocks > 16u * pl.full_grid)
            blocks = 16u * pl.full_grid;
        if (blocks < 1)
            blocks = 1;
        dim3 grid(blocks);
        switch (pl.variant & 3) {
        case 0: hipLaunchKernelGGL((k_camera<false, false>), grid, block, 0, stream, P); break;
        case 1: hipLaunchKernelGGL((k_camera<false, true>), grid, block, 0, stream, P); break;
        case 2: hipLaunchKernelGGL((k_camera<true, false>), grid, block, 0, stream, P); break;
        default: hipLaunchKernelGGL((k_camera<true, true>), grid, block, 0, stream, P); break;
        }
        RT_HIP_CHECK(ctx, hipGetLastError());
    }
    if (pl.wide) {
        // first launch: k_sliced, whose batches of fresh camera rays walk the exact tree wave-synchronously to their ends (scenes
        // within the L2s, and a chunk large enough to fill the chip that way); last launch: the re-fill kernel over the wide tree,
        // every path to its end.  Without the first, the re-fill kernel takes the camera rays as well.
        const bool by_wave = P.wave_camera != 0 && P.n_items > kWideWaveMinItems;
        const int last = first_level + (by_wave ? 1 : 0);
        for (int level = first_level; level <= last; level++) {
            P.level = level;
            P.tail = level == last ? 1 : 0;
            P.refill_tail = 0;
            P.tail_rays = 0u;
            P.q_in = ws.q[level & 1];
            P.q_out = ws.q[(level + 1) & 1];
            P.sub_in = ws.levels + (size_t)level * kSubQueues * 32;
            P.sub_cursor = ws.levels + (size_t)level * kSubQueues * 32 + 1;
            P.sub_out = ws.levels + (size_t)(level + 1) * kSubQueues * 32;
            uint32_t blocks = pl.full_grid;
            if (level == 0 && blocks > (P.n_items / 64 + 3) / 4)
                blocks = (P.n_items / 64 + 3) / 4;
            if (blocks < 1)
                blocks = 1;
            dim3 grid(blocks);
            const bool stats = (pl.variant & 1) != 0;
            if (level < last) {
                P.compact = 1;
                if (stats) hipLaunchKernelGGL((k_sliced<false, true, 1>), grid, block, lds_pad, stream, P);
                else hipLaunchKernelGGL((k_sliced<false, false, 1>), grid, block, lds_pad, stream, P);
            } else {
                P.compact = 3;
                const unsigned stack_bytes = 4u * P.wide_depth * 64u * (unsigned)sizeof(uint32_t); // four waves' stacks
                if (stats) hipLaunchKernelGGL((k_refill<false, true, 3, true>), grid, block, stack_bytes + lds_pad, stream, P);
                else hipLaunchKernelGGL((k_refill<false, false, 3, true>), grid, block, stack_bytes + lds_pad, stream, P);
            }
            RT_HIP_CHECK(ctx, hipGetLastError());
        }
        return RT_OK;
    }
    for (int level = first_level; level < levels; level++) {
        P.level = level;
        P.tail = level == levels - 1 ? 1 : 0;
        P.q_in = ws.q[level & 1];
        P.q_out = ws.q[(level + 1) & 1];
        P.sub_in = ws.levels + (size_t)level * kSubQueues * 32;
        P.sub_cursor = ws.levels + (size_t)level * kSubQueues * 32 + 1;
        P.sub_out = ws.levels + (size_t)(level + 1) * kSubQueues * 32;
        uint32_t blocks = pl.full_grid;
        if (level == 0 && blocks > (P.n_items / 64 + 3) / 4)
            blocks = (P.n_items / 64 + 3) / 4;
        if (blocks < 1)
            blocks = 1;
        dim3 grid(blocks);
        // wave-synchronous walk: register allocation capped at 64 VGPRs (8 waves per SIMD hide the scalar
        // fetch latency; the spills land in the shading code, which runs once per ray); the others: uncapped
#define RT_LAUNCH_PATH(OLD_, STATS_)                                                                                  \
    do {                                                                                                              \
        if (walk == WALK_FAST && sliced && P.tail != 0 && refill_tail && !OLD_ && P.compact == 2) hipLaunchKernelGGL((k_refill<false, STATS_, 2, true>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && sliced && P.tail != 0 && refill_tail && !OLD_ && P.compact == 1) hipLaunchKernelGGL((k_refill<false, STATS_, 1, true>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && sliced && P.tail != 0 && refill_tail) hipLaunchKernelGGL((k_refill<OLD_, STATS_, 0, true>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && sliced && !OLD_ && P.compact == 2) hipLaunchKernelGGL((k_sliced<false, STATS_, 2>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && sliced && !OLD_ && P.compact == 1) hipLaunchKernelGGL((k_sliced<false, STATS_, 1>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && sliced) hipLaunchKernelGGL((k_sliced<OLD_, STATS_>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && queued) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_QUEUED>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_WAVE) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_WAVE>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && refill && !OLD_ && P.compact == 2) hipLaunchKernelGGL((k_refill<false, STATS_, 2>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && refill && !OLD_ && P.compact == 1) hipLaunchKernelGGL((k_refill<false, STATS_, 1>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST && refill) hipLaunchKernelGGL((k_refill<OLD_, STATS_>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_FAST) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_FAST>), grid, block, lds_pad, stream, P); \
        else if (walk == WALK_BRUTE) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_BRUTE>), grid, block, lds_pad, stream, P); \
        else hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_LANE>), grid, block, lds_pad, stream, P);                         \
    } while (0)
        // The sliced walk hands a round that no longer fills the chip to the re-fill kernel: launched first, it takes the
        // round only if it is that small (and then k_sliced, launched after it, leaves at once); the last round is always its.
        if (walk == WALK_FAST && sliced && refill_tail && level >= 1 && level < levels - 1) {
            P.refill_tail = 1;
            switch (pl.variant) {
            case 4:
                if (P.compact == 2) hipLaunchKernelGGL((k_refill<false, false, 2, true>), grid, block, lds_pad, stream, P);
                else if (P.compact == 1) hipLaunchKernelGGL((k_refill<false, false, 1, true>), grid, block, lds_pad, stream, P);
                else hipLaunchKernelGGL((k_refill<false, false, 0, true>), grid, block, lds_pad, stream, P);
                break;
            case 5:
                if (P.compact == 2) hipLaunchKernelGGL((k_refill<false, true, 2, true>), grid, block, lds_pad, stream, P);
                else if (P.compact == 1) hipLaunchKernelGGL((k_refill<false, true, 1, true>), grid, block, lds_pad, stream, P);
                else hipLaunchKernelGGL((k_refill<false, true, 0, true>), grid, block, lds_pad, stream, P);
                break;
            case 6: hipLaunchKernelGGL((k_refill<true, false, 0, true>), grid, block, lds_pad, stream, P); break;
            default: hipLaunchKernelGGL((k_refill<true, true, 0, true>), grid, block, lds_pad, stream, P); break;
            }
            RT_HIP_CHECK(ctx, hipGetLastError());
        } else {
            P.refill_tail = 0;
        }
        switch (pl.variant) {
        case 4: RT_LAUNCH_PATH(false, false); break;
        case 5: RT_LAUNCH_PATH(false, true); break;
        case 6: RT_LAUNCH_PATH(true, false); break;
        default: RT_LAUNCH_PATH(true, true); break;
        }
#undef RT_LAUNCH_PATH
        RT_HIP_CHECK(ctx, hipGetLastError());
    }
    return RT_OK;
}

int launch_fold(rt_ctx *ctx, const Plan &pl, const Workspace &ws, hipStream_t stream, const KParams &P) {
    const unsigned blocks = (P.n_my_tiles + 3u) / 4u; // four tiles per block, one per wave
    if (blocks == 0)
        return RT_OK;
    if (pl.variant & 2)
        hipLaunchKernelGGL((k_fold<true>), dim3(blocks), dim3(256), 0, stream, P);
    else
        hipLaunchKernelGGL((k_fold<false>), dim3(blocks), dim3(256), 0, stream, P);
    RT_HIP_CHECK(ctx, hipGetLastError());
    return RT_OK;
}

// Gives a pipe's workspace room for chunks of `need_colors` colours and queues of `need_q[0]`, `need_q[1]` float4 (allocates; the
// pipe's stream is drained first when buffers have to be replaced).
int ensure_workspace(rt_ctx *ctx, rt_ctx::Pipe &pp, size_t need_colors, const size_t need_q[2], size_t need_pixels) {
    Workspace &ws = pp.ws;
    if (need_colors <= ws.colors_cap && need_q[0] <= ws.q_cap[0] && need_q[1] <= ws.q_cap[1] && need_pixels <= ws.deferred_cap)
        return RT_OK;
    RT_HIP_CHECK(ctx, hipStreamSynchronize(pp.stream));
    if (need_colors > ws.colors_cap) {
        if (ws.colors) (void)hipFree(ws.colors);
        ws.colors = nullptr;
        ws.colors_cap = 0;
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ws.colors, need_colors * sizeof(float4)));
        ws.colors_cap = need_colors;
    }
    for (int k = 0; k < 2; k++)
        if (need_q[k] > ws.q_cap[k]) {
            if (ws.q[k]) (void)hipFree(ws.q[k]);
            ws.q[k] = nullptr;
            ws.q_cap[k] = 0;
            RT_HIP_CHECK(ctx, hipMalloc((void **)&ws.q[k], need_q[k] * sizeof(float4)));
            ws.q_cap[k] = need_q[k];
        }
    if (need_pixels > ws.deferred_cap) {
        if (ws.deferred) (void)hipFree(ws.deferred);
        ws.deferred = nullptr;
        ws.deferred_cap = 0;
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ws.deferred, need_pixels * sizeof(unsigned long long)));
        ws.deferred_cap = need_pixels;
    }
    return RT_OK;
}

// float4 the two queues of a chunk of `items` pixel samples need under plan `pl`.  A queue of path records takes rec4
// float4 per record, the queue a camera stage fills one.  The wide flow uses what its two or three launches use: the
// camera stage writes q[1]; the wave-synchronous camera round, when the chunk has one, writes the queue the re-fill
// kernel then reads (q[0] after a camera stage, q[1] without); the re-fill kernel writes none.
void queue_needs(const Plan &pl, size_t items, size_t need[2]) {
    const size_t n = queue_records(items);
    need[0] = need[1] = n * pl.rec4;
    if (!pl.wide)
        return;
    need[0] = need[1] = 0;
    const bool by_wave = pl.P.wave_camera != 0 && items > kWideWaveMinItems;
    if (pl.camera) {
        need[1] = n;
        if (by_wave)
            need[0] = n * pl.rec4;
    } else if (by_wave) {
        need[1] = n * pl.rec4;
    }
}

int launch_trace(rt_ctx *ctx, int W, int H, int first, int count, float4 *d_img, uint32_t *d_prim, const float4 *d_img_in = nullptr,
                 bool timed = true) {
    Plan pl;
    {
        const int prc = make_plan(ctx, W, H, first, count, d_img, d_prim, pl, d_img_in);
        if (prc != RT_OK)
            return prc;
    }
    if (timed && ctx->pending == ctx->events.size()) {
        hipEvent_t a, b;
        RT_HIP_CHECK(ctx, hipEventCreate(&a));
        RT_HIP_CHECK(ctx, hipEventCreate(&b));
        ctx->events.emplace_back(a, b);
    }
    ctx->counted = true;
    if (pl.nested) {
        {
            const int jrc = join_pipes(ctx); // earlier pipelined calls may still be folding into this image
            if (jrc != RT_OK)
                return jrc;
        }
        if (timed)
            RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].first, ctx->stream));
        const int nrc = launch_nested(ctx, pl, ctx->stream);
        if (nrc != RT_OK)
            return nrc;
        if (timed) {
            RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].second, ctx->stream));
            ctx->pending++;
        }
        return RT_OK;
    }
    // ---- wavefront: samples in chunks of S; per chunk one launch per bounce level (or round), then k_fold ----
    const int S = chunk_samples(pl, kMaxItems);
    const size_t need_colors = (size_t)S * pl.items_per_sample;
    size_t need_q[2];
    queue_needs(pl, (size_t)S * pl.items_per_sample, need_q);
    if (timed)
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].first, ctx->stream));
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    hipStream_t last_stream = ctx->stream;
    if (!ctx->pipelining)
        ctx->next_pipe = 0; // calls that do not overlap start on the same pipe: one chunk per call then needs one workspace, not two
    for (int done = 0; done < count; done += S) {
        const int s_now = count - done < S ? count - done : S;
        rt_ctx::Pipe &pp = ctx->pipes[ctx->next_pipe];
        ctx->next_pipe ^= 1;
        {
            const int wrc = ensure_workspace(ctx, pp, need_colors, need_q, pl.items_per_sample * (size_t)((S + 63) / 64));
            if (wrc != RT_OK)
                return wrc;
        }
        RT_HIP_CHECK(ctx, hipStreamWaitEvent(pp.stream, ctx->ev_start, 0));
        KParams P;
        {
            const int crc = launch_chunk(ctx, pl, pp.ws, pp.stream, done, s_now, P);
            if (crc != RT_OK)
                return crc;
        }
        // the running mean takes the samples in order: this fold waits for the one before it, whichever pipe ran it
        if (ctx->last_fold != nullptr && ctx->last_fold != pp.done)
            RT_HIP_CHECK(ctx, hipStreamWaitEvent(pp.stream, ctx->last_fold, 0));
        {
            const int frc = launch_fold(ctx, pl, pp.ws, pp.stream, P);
            if (frc != RT_OK)
                return frc;
        }
        RT_HIP_CHECK(ctx, hipEventRecord(pp.done, pp.stream));
        pp.used = true;
        ctx->last_fold = pp.done;
        ctx->joined = false;
        last_stream = pp.stream;
    }
    if (!ctx->pipelining) { // the call's results are ordered on the context's stream, as documented
        int jrc = join_pipes(ctx);
        if (jrc != RT_OK)
            return jrc;
        last_stream = ctx->stream;
    }
    if (timed) {
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].second, last_stream));
        ctx->pending++;
    }
    return RT_OK;
}

// The same work with no allocation of any kind: every launch goes to the context's stream, one chunk after the other, in
// the device's built-in workspace; no timing events are taken.
int launch_trace_noalloc(rt_ctx *ctx, int W, int H, int first, int count, float4 *d_img, uint32_t *d_prim) {
    Plan pl;
    {
        const int prc = make_plan(ctx, W, H, first, count, d_img, d_prim, pl, nullptr, false);
        if (prc != RT_OK)
            return prc;
    }
    if (pl.items_per_sample > kScratchMaxItems)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT,
                    "rt_trace_device_noalloc: the frame is too large for the built-in workspace (at most " +
                        std::to_string(kScratchMaxItems) + " pixels, counted in whole 8x8 tiles; this context's share of the frame has " +
                        std::to_string(pl.items_per_sample) + ")");
    if (ctx->scratch_ptr == nullptr)
        return fail(ctx, RT_ERR_HIP, "the built-in workspace has no address on this device");
    {
        const int jrc = join_pipes(ctx); // earlier pipelined calls may still be folding into this image
        if (jrc != RT_OK)
            return jrc;
    }
    ctx->counted = true;
    if (pl.nested)
        return launch_nested(ctx, pl, ctx->stream);
    unsigned char *base = (unsigned char *)ctx->scratch_ptr;
    Workspace ws;
    ws.levels = (unsigned int *)base;
    ws.colors = (float4 *)(base + kScratchLevelsBytes);
    ws.deferred = (unsigned long long *)(base + kScratchLevelsBytes + kScratchColorsBytes);
    ws.q[0] = (float4 *)(base + kScratchLevelsBytes + kScratchColorsBytes + kScratchDeferredBytes);
    ws.q[1] = (float4 *)(base + kScratchLevelsBytes + kScratchColorsBytes + kScratchDeferredBytes + kScratchQueueBytes);
    ws.deferred_cap = kScratchMaxItems;
    ws.colors_cap = kScratchMaxItems;
    ws.q_cap[0] = ws.q_cap[1] = (kScratchMaxItems + kQueueSlack) * 6;
    const int S = chunk_samples(pl, kScratchMaxItems); // S * items <= kScratchMaxItems: the colours and the `deferred` words (64 slots per owned tile and sample) fit too
    for (int done = 0; done < count; done += S) {
        const int s_now = count - done < S ? count - done : S;
        KParams P;
        int rc = launch_chunk(ctx, pl, ws, ctx->stream, done, s_now, P);
        if (rc == RT_OK)
            rc = launch_fold(ctx, pl, ws, ctx->stream, P);
        if (rc != RT_OK)
            return rc;
    }
    return RT_OK;
}

void add_stats(rt_stats &a, const rt_stats &b) {
    a.rays += b.rays;
    a.box_tests += b.box_tests;
    a.tri_tests += b.tri_tests;
    a.sph_tests += b.sph_tests;
    a.tri_hits += b.tri_hits;
    a.replays += b.replays;
    a.kernel_ms = a.kernel_ms > b.kernel_ms ? a.kernel_ms : b.kernel_ms; // devices work side by side
    a.kernel_ms_sum = a.kernel_ms_sum > b.kernel_ms_sum ? a.kernel_ms_sum : b.kernel_ms_sum;
    a.launches = a.launches > b.launches ? a.launches : b.launches;
}

int sync_and_stats(rt_ctx *ctx, rt_stats *stats) {
    {
        const int src = sync_all(ctx);
        if (src != RT_OK)
            return src;
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        if (ctx->counted) {
            unsigned long long slots[kCounterSlots * kCounterStride], h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            RT_HIP_CHECK(ctx, hipMemcpy(slots, ctx->d_counters, sizeof(slots), hipMemcpyDeviceToHost));
            for (int sl = 0; sl < kCounterSlots; sl++)
                for (int k = 0; k < 8; k++) {
                    const unsigned long long v = slots[sl * kCounterStride + k];
                    h[k] += v;
                }
            stats->rays = h[0];
            stats->box_tests = h[1];
            stats->tri_tests = h[2];
            stats->sph_tests = h[3];
            stats->tri_hits = h[4];
            stats->replays = h[5];
        }
        if (ctx->pending > 0) {
            float ms = 0.f, sum = 0.f;
            for (size_t i = 0; i < ctx->pending; i++) {
                RT_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
                sum += ms;
            }
            stats->kernel_ms = ms;
            if (ctx->pipelining) // calls overlap: the time they took together, first start to last end
                RT_HIP_CHECK(ctx, hipEventElapsedTime(&sum, ctx->events[0].first, ctx->events[ctx->pending - 1].second));
            stats->kernel_ms_sum = sum;
            stats->launches = (uint32_t)ctx->pending;
        }
    }
    if (ctx->counted) // counters run from one sync to the next
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, kCounterSlots * kCounterStride * sizeof(unsigned long long), ctx->stream));
    ctx->pending = 0;
    ctx->counted = false;
    return RT_OK;
}

// (Re)allocates the context's own running-mean image for a W x H frame; a new size starts from zero
// (CPUImageData::resize, Renderer/TraceMethods.hpp:31-36, as TracerAsync::resize + resetSamples use it).
int ensure_image(rt_ctx *ctx, int width, int height) {
    const size_t px = (size_t)width * (size_t)height;
    if (width != ctx->img_w || height != ctx->img_h || !ctx->d_image) {
        RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (float4 *&r : ctx->ring) {
            if (r) (void)hipFree(r);
            r = nullptr;
        }
        if (ctx->d_primary) (void)hipFree(ctx->d_primary);
        ctx->d_image = nullptr;
        ctx->d_primary = nullptr;
        ctx->img_w = ctx->img_h = 0;
        ctx->ring_at = 0;
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->ring[0], px * sizeof(float4)));
        ctx->d_image = ctx->ring[0];
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_primary, px * sizeof(uint32_t)));
        ctx->img_w = width;
        ctx->img_h = height;
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_image, 0, px * sizeof(float4), ctx->stream));
    }
    return RT_OK;
}

// Drops what rt_trace_begin has put in flight: waits for it, forgets it; the running mean is the last ended call's again.
int discard_begun(rt_ctx *ctx) {
    if (ctx->begun.empty())
        return RT_OK;
    {
        const int src = sync_all(ctx);
        if (src != RT_OK)
            return src;
    }
    if (ctx->copy_stream)
        RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
    ctx->begun.clear();
    return RT_OK;
}

#include "scene_layout.hpp"

int upload_layout(rt_ctx *ctx, const SceneLayout &L) {
    RT_ON_DEVICE(ctx);
    {
        const int src = sync_all(ctx); // nothing may still be walking the old scene
        if (src != RT_OK)
            return src;
    }
    free_scene(ctx);
    int rc;
    if ((rc = upload(ctx, &ctx->d_nodes, L.dn)) != RT_OK || (rc = upload(ctx, &ctx->d_fnodes, L.fn)) != RT_OK ||
        (rc = upload(ctx, &ctx->d_tri_pos, L.tp)) != RT_OK || (rc = upload(ctx, &ctx->d_tri_nrm, L.tn)) != RT_OK ||
        (rc = upload(ctx, &ctx->d_spheres, L.sp)) != RT_OK || (rc = upload(ctx, &ctx->d_mats, L.mt)) != RT_OK) {
        free_scene(ctx);
        return rc;
    }
    if (L.compact) {
        if ((rc = upload(ctx, &ctx->d_cnodes, L.cn)) != RT_OK || (rc = upload(ctx, &ctx->d_leaf_boxes, L.lb)) != RT_OK) {
            free_scene(ctx);
            return rc;
        }
        std::memcpy(ctx->q_origin, L.q_origin, sizeof(ctx->q_origin));
        std::memcpy(ctx->q_step, L.q_step, sizeof(ctx->q_step));
        if (L.wide) {
            if ((rc = upload(ctx, &ctx->d_wnodes, L.wn)) != RT_OK) {
                free_scene(ctx);
                return rc;
            }
            ctx->has_wide = true;
            ctx->n_wide = L.n_wide;
            ctx->wide_visits = L.wide_visits;
            ctx->wide_depth = L.wide_depth > 0 ? L.wide_depth : 1u;
            ctx->can_prune = L.prune;
            for (int k = 0; k < 3; k++) {
                ctx->prune_c[k] = L.prune_c1[k];
                ctx->prune_c[3 + k] = L.prune_c2x2[k];
                ctx->prune_c[6 + k] = L.prune_c3[k];
            }
        }
        ctx->has_compact = true;
        ctx->compact_inflation = L.compact_inflation;
        ctx->fast_visits = L.fast_visits;
        if (std::getenv("RT_DEBUG_LAYOUT")) {
            std::fprintf(stderr, "compact inflation %.4f expected visits %.2f fast visits %.2f scene bytes %zu; wide %d: %u nodes, depth %u, visits %.2f\n",
                         L.compact_inflation, L.expected_visits, L.fast_visits, (L.fn.size() + L.tp.size()) * sizeof(float4), (int)L.wide, L.n_wide,
                         L.wide_depth, L.wide_visits);
            std::fprintf(stderr, "distance limit %d: c1 %.3g %.3g %.3g  2 c2 %.3g %.3g %.3g  c2W+c3 %.3g %.3g %.3g, %u irregular leaves\n", (int)L.prune,
                         L.prune_c1[0], L.prune_c1[1], L.prune_c1[2], L.prune_c2x2[0], L.prune_c2x2[1], L.prune_c2x2[2], L.prune_c3[0], L.prune_c3[1],
                         L.prune_c3[2], L.n_irregular);
        }
    }
    ctx->expected_visits = L.expected_visits;
    if (L.fn.size() >= 2) {
        const float4 a = L.fn[0], b = L.fn[1];
        const float rb[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
        std::memcpy(ctx->root_box, rb, sizeof(rb));
        ctx->cut_boxes = L.cut_boxes;
    }
    ctx->fast_exact = L.nested;
    ctx->scene_bytes = (L.fn.size() + L.tp.size()) * sizeof(float4);
    ctx->n_nodes = L.n_ref;
    ctx->n_fnodes = L.n_fast;
    ctx->scene_version++;
    ctx->mask_valid = false;
    ctx->has_scene = true;
    return RT_OK;
}

bool options_ok(const rt_options *opt) {
    return !(opt->bounce_limit < 0 || opt->bounce_limit > kMaxLevels - 2 || opt->tile_world < 1 || opt->tile_rank < 0 ||
             opt->tile_rank >= opt->tile_world || (opt->estimator != RT_EST_SD_LOOP && opt->estimator != RT_EST_OLD_RECURSIVE) ||
             opt->traversal < 0 || opt->traversal > RT_TRAV_FAST_SLICED ||
             (opt->multi_split != RT_SPLIT_TILE && opt->multi_split != RT_SPLIT_SAMPLE) || opt->camera_stage < 0 ||
             opt->camera_stage > RT_CAMERA_FUSED);
}

void destroy_single(rt_ctx *ctx) {
    if (!ctx)
        return;
    DeviceGuard g(ctx->device);
    for (auto &pp : ctx->pipes)
        if (pp.stream)
            (void)hipStreamSynchronize(pp.stream);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (float4 *r : ctx->ring)
        if (r) (void)hipFree(r);
    if (ctx->d_primary) (void)hipFree(ctx->d_primary);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (hipEvent_t e : ctx->ev_copied)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev_call) (void)hipEventDestroy(ctx->ev_call);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_cam_mask) (void)hipFree(ctx->d_cam_mask);
    for (auto &pp : ctx->pipes) {
        if (pp.ws.levels) (void)hipFree(pp.ws.levels);
        pp.ws.levels = nullptr;
        free_workspace(pp.ws);
        if (pp.done) (void)hipEventDestroy(pp.done);
        if (pp.stream) (void)hipStreamDestroy(pp.stream);
    }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    for (auto &e : ctx->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (float4 *p : ctx->peer_img)
        if (p) (void)hipFree(p);
    if (ctx->multi_frame) (void)hipFree(ctx->multi_frame);
    for (uint32_t *p : ctx->peer_prim)
        if (p) (void)hipFree(p);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int create_single(rt_ctx **out, int device) {
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return RT_ERR_NO_DEVICE;
    rt_ctx *ctx = new rt_ctx();
    ctx->device = device;
    std::memset(&ctx->cam, 0, sizeof(ctx->cam));
    ctx->opt = rt_options{4, RT_EST_SD_LOOP, 1u, 0.001f, 0, 1, 0, 0u, RT_TRAV_AUTO, RT_SPLIT_TILE, RT_CAMERA_AUTO};
    DeviceGuard g(device);
    auto bail = [&](const char *what) { // whatever exists by now is released the way rt_destroy releases it
        std::fprintf(stderr, "rt_create: %s failed\n", what);
        destroy_single(ctx);
        return RT_ERR_HIP;
    };
    if (g.rc != hipSuccess) return bail("hipSetDevice");
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
    ctx->own_stream = true;
    if (hipMalloc((void **)&ctx->d_queue, 64) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void **)&ctx->d_counters, kCounterSlots * kCounterStride * sizeof(unsigned long long)) != hipSuccess) return bail("hipMalloc");
    if (hipMemset(ctx->d_counters, 0, kCounterSlots * kCounterStride * sizeof(unsigned long long)) != hipSuccess) return bail("hipMemset");
    if (hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
    for (auto &pp : ctx->pipes) {
        if (hipStreamCreateWithFlags(&pp.stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
        if (hipEventCreateWithFlags(&pp.done, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
        if (hipMalloc((void **)&pp.ws.levels, kLevelWords * sizeof(unsigned int)) != hipSuccess) return bail("hipMalloc");
    }
    if (scratch_of(ctx, device, &ctx->scratch_ptr) != RT_OK) return bail("hipGetSymbolAddress");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail("hipGetDeviceProperties");
    ctx->num_cus = prop.multiProcessorCount;
    occupancy<0, false, false>(&ctx->blocks_per_cu[0]);
    occupancy<0, false, true>(&ctx->blocks_per_cu[1]);
    occupancy<0, true, false>(&ctx->blocks_per_cu[2]);
    occupancy<0, true, true>(&ctx->blocks_per_cu[3]);
    for (int v = 4; v < 8; v++)
        occupancy<1, false, false>(&ctx->blocks_per_cu[v]);
    *out = ctx;
    return RT_OK;
}

// ------------------------------------------------------------------ several devices behind one context
// Landing buffers on the first device of a multi-device context, one per other device, for frames of `px` pixels.
int ensure_landing(rt_ctx *ctx, size_t px) {
    const int n = (int)ctx->kids.size();
    if (px <= ctx->peer_cap)
        return RT_OK;
    RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream)
        RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (int r = 1; r < n; r++) {
        if (ctx->peer_img[r]) (void)hipFree(ctx->peer_img[r]);
        if (ctx->peer_prim[r]) (void)hipFree(ctx->peer_prim[r]);
        ctx->peer_img[r] = nullptr;
        ctx->peer_prim[r] = nullptr;
    }
    ctx->peer_cap = 0;
    for (int r = 1; r < n; r++) {
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->peer_img[r], px * sizeof(float4)));
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->peer_prim[r], px * sizeof(uint32_t)));
    }
    ctx->peer_cap = px;
    return RT_OK;
}

// rt_create_multi: kid r renders its share of every call on its own device -- its tiles of the frame (RT_SPLIT_TILE:
// tile k belongs to kid k % n) or its samples of the whole frame (RT_SPLIT_SAMPLE) -- into its own image; one host
// thread per device issues the work and waits for it; the frames then travel to the first device of the list
// (hipMemcpyPeerAsync over xGMI, one landing buffer per kid) and are put together there: owned tiles copied, or the
// per-kid means added with weights samples_r / samples.  The scene is replicated on every device.
int multi_trace(rt_ctx *ctx, int W, int H, int first, int count, float4 *d_out, uint32_t *d_prim, bool zero_out) {
    const int n = (int)ctx->kids.size();
    if (!ctx->has_camera)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera not set");
    if (W <= 0 || H <= 0 || first < 1 || count < 0 || (long long)W * H > 0x7FFFFFFFll)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad width/height/sample range");
    const size_t px = (size_t)W * (size_t)H;
    const bool by_sample = ctx->opt.multi_split == RT_SPLIT_SAMPLE;
    if (by_sample && first != 1) {
        // Each device keeps the mean of ITS samples since the last restart and the frame is their weighted sum: a call can
        // only continue the run this context itself has rendered -- not a mean accumulated elsewhere, not after a resize.
        int so_far = 0;
        for (int s : ctx->kid_samples)
            so_far += s;
        if (first != so_far + 1 || W != ctx->multi_w || H != ctx->multi_h)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT,
                        "sample split over several devices: a call must restart the mean (first_sample = 1) or continue this "
                        "context's own run at the same frame size (first_sample = " + std::to_string(so_far + 1) + " expected)");
    }
    RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // the previous call's gather may still be reading the kids' frames
    if (first == 1)
        for (int &s : ctx->kid_samples)
            s = 0;
    ctx->multi_w = W;
    ctx->multi_h = H;
    {
        const int lrc = ensure_landing(ctx, px);
        if (lrc != RT_OK)
            return lrc;
    }
    // shares
    std::vector<int> share(n, 0), start(n, 0);
    for (int r = 0, at = 0; r < n; r++) {
        share[r] = by_sample ? count / n + (r < count % n ? 1 : 0) : count;
        start[r] = by_sample ? at : 0;
        at += share[r];
    }
    std::vector<int> rcs(n, RT_OK);
    auto work = [&](int r) {
        rt_ctx *k = ctx->kids[r];
        std::lock_guard<std::mutex> lock(k->mu);
        DeviceGuard g(k->device);
        if (g.rc != hipSuccess) {
            rcs[r] = fail(k, RT_ERR_HIP, "hipSetDevice failed");
            return;
        }
        rt_options o = ctx->opt;
        int kfirst = first;
        if (by_sample) { // the kid keeps a running mean of ITS samples: local numbering, the global index goes into the RNG key
            kfirst = ctx->kid_samples[r] + 1;
            o.sample_offset = ctx->opt.sample_offset + (uint32_t)(first + start[r] - kfirst);
        } else {
            o.tile_rank = ctx->opt.tile_rank + r * ctx->opt.tile_world;
            o.tile_world = ctx->opt.tile_world * n;
        }
        k->opt = o;
        k->cam = ctx->cam;
        k->has_camera = true;
        int rc = ensure_image(k, W, H);
        if (rc == RT_OK && kfirst == 1)
            rc = hipMemsetAsync(k->d_image, 0, px * sizeof(float4), k->stream) == hipSuccess ? RT_OK : fail(k, RT_ERR_HIP, "hipMemsetAsync failed");
        const bool want_prim = d_prim != nullptr && (!by_sample || r == 0);
        if (rc == RT_OK && want_prim)
            rc = hipMemsetAsync(k->d_primary, 0xFF, px * sizeof(uint32_t), k->stream) == hipSuccess ? RT_OK : fail(k, RT_ERR_HIP, "hipMemsetAsync failed");
        if (rc == RT_OK && share[r] > 0)
            rc = launch_trace(k, W, H, kfirst, share[r], k->d_image, want_prim ? k->d_primary : nullptr);
        if (rc == RT_OK)
            rc = sync_all(k);
        rcs[r] = rc;
    };
    {
        std::vector<std::thread> threads;
        for (int r = 1; r < n; r++)
            threads.emplace_back(work, r);
        work(0);
        for (auto &t : threads)
            t.join();
    }
    for (int r = 0; r < n; r++)
        if (rcs[r] != RT_OK) {
            std::string msg;
            {
                std::lock_guard<std::mutex> g(ctx->kids[r]->err_mu);
                msg = ctx->kids[r]->err;
            }
            return fail(ctx, rcs[r], "device " + std::to_string(ctx->kids[r]->device) + ": " + msg);
        }
    if (by_sample)
        for (int r = 0; r < n; r++)
            ctx->kid_samples[r] += share[r];
    // gather on the first device
    if (zero_out)
        RT_HIP_CHECK(ctx, hipMemsetAsync(d_out, 0, px * sizeof(float4), ctx->stream));
    if (d_prim != nullptr)
        RT_HIP_CHECK(ctx, hipMemsetAsync(d_prim, 0xFF, px * sizeof(uint32_t), ctx->stream));
    int total = 0;
    for (int r = 0; r < n; r++)
        total += ctx->kid_samples[r];
    const dim3 grid((unsigned)((px + 255) / 256)), block(256);
    bool first_term = true;
    for (int r = 0; r < n; r++) {
        rt_ctx *k = ctx->kids[r];
        const float4 *src = k->d_image;
        const uint32_t *psrc = k->d_primary;
        const bool want_prim = d_prim != nullptr && (!by_sample || r == 0);
        if (r > 0) {
            RT_HIP_CHECK(ctx, hipMemcpyPeerAsync(ctx->peer_img[r], ctx->device, k->d_image, k->device, px * sizeof(float4), ctx->stream));
            src = ctx->peer_img[r];
            if (want_prim) {
                RT_HIP_CHECK(ctx, hipMemcpyPeerAsync(ctx->peer_prim[r], ctx->device, k->d_primary, k->device, px * sizeof(uint32_t), ctx->stream));
                psrc = ctx->peer_prim[r];
            }
        }
        if (by_sample) {
            if (ctx->kid_samples[r] == 0)
                continue;
            const float w = (float)ctx->kid_samples[r] / (float)total;
            hipLaunchKernelGGL(k_add_weighted, grid, block, 0, ctx->stream, d_out, src, (uint32_t)px, w, first_term ? 1 : 0);
            first_term = false;
            if (want_prim)
                RT_HIP_CHECK(ctx, hipMemcpyAsync(d_prim, psrc, px * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            hipLaunchKernelGGL(k_take_tiles, grid, block, 0, ctx->stream, d_out, src, want_prim ? d_prim : nullptr, psrc, (uint32_t)px, W,
                               (W + 7) / 8, ctx->opt.tile_rank + r * ctx->opt.tile_world, ctx->opt.tile_world * n);
        }
        RT_HIP_CHECK(ctx, hipGetLastError());
    }
    return RT_OK;
}

// One call put in flight on a single-device context (rt_trace_begin; the caller holds the context's lock and has made its
// device current): the mean moves on round the ring, the call's kernels overlap the end of the call before, and `copied`
// of the new entry of `begun` is signalled when the call's image is complete -- and copied to out_rgba_mean, if given.
int begin_frame(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *out_rgba_mean) {
    const size_t px = (size_t)width * (size_t)height;
    if (width != ctx->img_w || height != ctx->img_h || !ctx->d_image) {
        int rc = discard_begun(ctx);
        if (rc == RT_OK)
            rc = join_pipes(ctx);
        if (rc == RT_OK)
            rc = ensure_image(ctx, width, height);
        if (rc != RT_OK)
            return rc;
    }
    if (!ctx->copy_stream) {
        RT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t &e : ctx->ev_copied)
            RT_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        RT_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_call, hipEventDisableTiming));
    }
    for (float4 *&r : ctx->ring)
        if (!r)
            RT_HIP_CHECK(ctx, hipMalloc((void **)&r, px * sizeof(float4)));
    const int from = ctx->begun.empty() ? ctx->ring_at : ctx->begun.back().img, to = (from + 1) % 3;
    // first_sample == 1 restarts the mean: the folds overwrite every owned pixel, and a share of the frame (tile split)
    // must find the rest zero
    if (first_sample == 1 && ctx->opt.tile_world > 1)
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->ring[to], 0, px * sizeof(float4), ctx->stream));
    // the call's chunks are ordered among themselves and behind the call before (the folds chain); nothing makes the
    // context's stream wait, so this call's kernels overlap the latency-bound end of the one before
    const bool was = ctx->pipelining;
    ctx->pipelining = true;
    const int rc = launch_trace(ctx, width, height, first_sample, n_samples, ctx->ring[to], nullptr, ctx->ring[from], false);
    ctx->pipelining = was;
    if (rc != RT_OK)
        return rc;
    if (!ctx->joined && ctx->last_fold != nullptr) {
        RT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->last_fold, 0));
    } else {
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_call, ctx->stream));
        RT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_call, 0));
    }
    if (out_rgba_mean)
        RT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgba_mean, ctx->ring[to], px * sizeof(float4), hipMemcpyDeviceToHost, ctx->copy_stream));
    hipEvent_t ev = ctx->ev_copied[ctx->next_copied];
    ctx->next_copied ^= 1;
    RT_HIP_CHECK(ctx, hipEventRecord(ev, ctx->copy_stream));
    ctx->begun.push_back(rt_ctx::Begun{to, out_rgba_mean, ev});
    return RT_OK;
}

// The oldest call in flight becomes the context's running mean (waits for it).
int end_frame(rt_ctx *ctx) {
    const rt_ctx::Begun b = ctx->begun.front();
    RT_HIP_CHECK(ctx, hipEventSynchronize(b.copied));
    ctx->begun.pop_front();
    ctx->ring_at = b.img;
    ctx->d_image = ctx->ring[b.img];
    return RT_OK;
}

// The counters as they stand (a call begun later may already have added to them); added to *stats.
int read_counters(rt_ctx *ctx, rt_stats *stats) {
    unsigned long long slots[kCounterSlots * kCounterStride], h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RT_HIP_CHECK(ctx, hipMemcpyAsync(slots, ctx->d_counters, sizeof(slots), hipMemcpyDeviceToHost, ctx->copy_stream));
    RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (int sl = 0; sl < kCounterSlots; sl++)
        for (int k = 0; k < 8; k++)
            h[k] += slots[sl * kCounterStride + k];
    stats->rays += h[0];
    stats->box_tests += h[1];
    stats->tri_tests += h[2];
    stats->sph_tests += h[3];
    stats->tri_hits += h[4];
    stats->replays += h[5];
    return RT_OK;
}

// rt_trace_begin on several devices (tile split): every device begins the call on its own share of the frame -- its own
// ring of means, exactly as a single-device context does -- and the first device's copy stream, once each share is complete,
// fetches it (hipMemcpyPeerAsync), puts the frame together and sends it to the host.  One stream carries all of that for
// every call, so one landing buffer per device and one assembled frame serve the two calls in flight.
int multi_begin(rt_ctx *ctx, int W, int H, int first, int count, float *out_rgba_mean) {
    const int n = (int)ctx->kids.size();
    if (ctx->opt.multi_split != RT_SPLIT_TILE)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_trace_begin on several devices needs the tile split (RT_SPLIT_TILE)");
    if (!ctx->has_camera)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera not set");
    const size_t px = (size_t)W * (size_t)H;
    if (!ctx->copy_stream) {
        RT_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t &e : ctx->ev_copied)
            RT_HIP_CHECK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    {
        const int lrc = ensure_landing(ctx, px);
        if (lrc != RT_OK)
            return lrc;
    }
    if (px > ctx->multi_frame_cap) {
        RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
        if (ctx->multi_frame) (void)hipFree(ctx->multi_frame);
        ctx->multi_frame = nullptr;
        ctx->multi_frame_cap = 0;
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->multi_frame, px * sizeof(float4)));
        ctx->multi_frame_cap = px;
    }
    for (int r = 0; r < n; r++) {
        rt_ctx *k = ctx->kids[r];
        std::lock_guard<std::mutex> lock(k->mu);
        DeviceGuard g(k->device);
        if (g.rc != hipSuccess)
            return fail(ctx, RT_ERR_HIP, "hipSetDevice failed");
        rt_options o = ctx->opt;
        o.tile_rank = ctx->opt.tile_rank + r * ctx->opt.tile_world;
        o.tile_world = ctx->opt.tile_world * n;
        k->opt = o;
        k->cam = ctx->cam;
        k->has_camera = true;
        const int rc = begin_frame(k, W, H, first, count, nullptr);
        if (rc != RT_OK) {
            std::string msg;
            {
                std::lock_guard<std::mutex> eg(k->err_mu);
                msg = k->err;
            }
            // a call is in flight on all devices or on none: everything begun so far is dropped (this device's lock is held here)
            if (ctx->copy_stream)
                (void)hipStreamSynchronize(ctx->copy_stream);
            ctx->begun.clear();
            (void)discard_begun(k);
            for (int q = 0; q < n; q++) {
                if (q == r)
                    continue;
                rt_ctx *kq = ctx->kids[q];
                std::lock_guard<std::mutex> ql(kq->mu);
                DeviceGuard qg(kq->device);
                (void)discard_begun(kq);
            }
            return fail(ctx, rc, "device " + std::to_string(k->device) + ": " + msg);
        }
    }
    const dim3 grid((unsigned)((px + 255) / 256)), block(256);
    for (int r = 0; r < n; r++) {
        rt_ctx *k = ctx->kids[r];
        const rt_ctx::Begun &kb = k->begun.back();
        RT_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->copy_stream, kb.copied, 0));
        const float4 *src = k->ring[kb.img];
        if (r > 0) {
            RT_HIP_CHECK(ctx, hipMemcpyPeerAsync(ctx->peer_img[r], ctx->device, src, k->device, px * sizeof(float4), ctx->copy_stream));
            src = ctx->peer_img[r];
        }
        hipLaunchKernelGGL(k_take_tiles, grid, block, 0, ctx->copy_stream, ctx->multi_frame, src, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                           (uint32_t)px, W, (W + 7) / 8, ctx->opt.tile_rank + r * ctx->opt.tile_world, ctx->opt.tile_world * n);
        RT_HIP_CHECK(ctx, hipGetLastError());
    }
    RT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgba_mean, ctx->multi_frame, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->copy_stream));
    hipEvent_t ev = ctx->ev_copied[ctx->next_copied];
    ctx->next_copied ^= 1;
    RT_HIP_CHECK(ctx, hipEventRecord(ev, ctx->copy_stream));
    ctx->begun.push_back(rt_ctx::Begun{-1, out_rgba_mean, ev});
    return RT_OK;
}

int multi_end(rt_ctx *ctx, rt_stats *stats) {
    const rt_ctx::Begun b = ctx->begun.front();
    RT_HIP_CHECK(ctx, hipEventSynchronize(b.copied));
    ctx->begun.pop_front();
    if (stats)
        std::memset(stats, 0, sizeof(*stats));
    for (rt_ctx *k : ctx->kids) {
        std::lock_guard<std::mutex> lock(k->mu);
        DeviceGuard g(k->device);
        int rc = end_frame(k); // (complete already: the frame was put together from it)
        if (rc == RT_OK && stats)
            rc = read_counters(k, stats);
        if (rc != RT_OK)
            return fail(ctx, rc, "device " + std::to_string(k->device) + ": rt_trace_end failed");
    }
    return RT_OK;
}

// Drops the calls in flight, on every device behind the context.
int discard_all(rt_ctx *ctx) {
    if (ctx->kids.empty())
        return discard_begun(ctx);
    if (ctx->begun.empty())
        return RT_OK;
    if (ctx->copy_stream)
        RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->copy_stream));
    ctx->begun.clear();
    for (rt_ctx *k : ctx->kids) {
        std::lock_guard<std::mutex> lock(k->mu);
        DeviceGuard g(k->device);
        const int rc = discard_begun(k);
        if (rc != RT_OK)
            return fail(ctx, rc, "device " + std::to_string(k->device) + ": rt_trace_discard failed");
    }
    return RT_OK;
}

int multi_sync(rt_ctx *ctx, rt_stats *stats) {
    RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (stats)
        std::memset(stats, 0, sizeof(*stats));
    for (rt_ctx *k : ctx->kids) {
        std::lock_guard<std::mutex> lock(k->mu);
        DeviceGuard g(k->device);
        rt_stats s;
        const int rc = sync_and_stats(k, &s);
        if (rc != RT_OK)
            return fail(ctx, rc, "device " + std::to_string(k->device) + ": sync failed");
        if (stats)
            add_stats(*stats, s);
    }
    return RT_OK;
}

} // namespace

extern "C" {

void *rt_alloc_host(uint64_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess)
        return nullptr;
    return p;
}
void rt_free_host(void *ptr) {
    if (ptr)
        (void)hipHostFree(ptr);
}

void rt_vec_op(int op, const float *a, const float *b, const float *c, const float *v, float *out3) {
    const f3 A = a ? mk3(a[0], a[1], a[2]) : mk3(0, 0, 0), B = b ? mk3(b[0], b[1], b[2]) : mk3(0, 0, 0),
             C = c ? mk3(c[0], c[1], c[2]) : mk3(0, 0, 0), X = v ? mk3(v[0], v[1], v[2]) : mk3(0, 0, 0);
    f3 r = mk3(0, 0, 0);
    switch (op) {
    case 0: r.x = dot3(A, B); break;
    case 1: r = cross3(A, B); break;
    case 2: r = normalize3(A); break;
    default: r = basis_mul(A, B, C, X); break;
    }
    out3[0] = r.x, out3[1] = r.y, out3[2] = r.z;
}

const char *rt_version(void) { return "rt_hip 0.2 gfx950"; }

int rt_create(rt_ctx **out, int device) {
    if (!out)
        return RT_ERR_INVALID_ARGUMENT;
    return create_single(out, device);
}

int rt_create_multi(rt_ctx **out, const int *device_ids, int n_devices) {
    if (!out)
        return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64)
        return RT_ERR_INVALID_ARGUMENT;
    if (n_devices == 1)
        return create_single(out, device_ids[0]);
    rt_ctx *root = nullptr;
    int rc = create_single(&root, device_ids[0]);
    if (rc != RT_OK)
        return rc;
    root->peer_img.assign(n_devices, nullptr);
    root->peer_prim.assign(n_devices, nullptr);
    root->kid_samples.assign(n_devices, 0);
    for (int r = 0; r < n_devices; r++) {
        rt_ctx *k = nullptr;
        rc = create_single(&k, device_ids[r]);
        if (rc != RT_OK) {
            for (rt_ctx *made : root->kids)
                destroy_single(made);
            root->kids.clear();
            destroy_single(root);
            return rc;
        }
        root->kids.push_back(k);
    }
    *out = root;
    return RT_OK;
}

int rt_device_count(rt_ctx *ctx) { return !ctx ? 0 : ctx->kids.empty() ? 1 : (int)ctx->kids.size(); }

void rt_destroy(rt_ctx *ctx) {
    if (!ctx)
        return;
    for (rt_ctx *k : ctx->kids)
        destroy_single(k);
    ctx->kids.clear();
    destroy_single(ctx);
}

const char *rt_last_error(rt_ctx *ctx) {
    static thread_local std::string copy; // stays valid for the calling thread whatever other threads do to the context
    if (!ctx)
        return "null context";
    {
        std::lock_guard<std::mutex> g(ctx->err_mu);
        copy = ctx->err;
    }
    return copy.c_str();
}

int rt_upload_scene(rt_ctx *ctx, const rt_node *nodes, uint32_t n_nodes, uint32_t root, const rt_tri *tris,
                    uint32_t n_tris, const rt_sphere *spheres, uint32_t n_spheres, const rt_material *mats,
                    uint32_t n_mats, int order) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    SceneLayout L;
    {
        std::string err;
        const int lrc = build_layout(nodes, n_nodes, root, tris, n_tris, spheres, n_spheres, mats, n_mats, order, L, err);
        if (lrc != RT_OK)
            return fail(ctx, lrc, err);
    }
    if (ctx->kids.empty()) {
        {
            DeviceGuard g(ctx->device);
            const int drc = discard_begun(ctx); // look-ahead calls saw the old scene
            if (drc != RT_OK)
                return drc;
        }
        return upload_layout(ctx, L);
    }
    {
        DeviceGuard g(ctx->device);
        const int drc = discard_all(ctx); // look-ahead calls saw the old scene
        if (drc != RT_OK)
            return drc;
    }
    for (rt_ctx *k : ctx->kids) { // replicated: every device walks its own copy
        std::lock_guard<std::mutex> klock(k->mu);
        const int rc = upload_layout(k, L);
        if (rc != RT_OK)
            return fail(ctx, rc, "device " + std::to_string(k->device) + ": " + rt_last_error(k));
    }
    ctx->has_scene = true;
    return RT_OK;
}

int rt_set_camera(rt_ctx *ctx, const rt_camera *cam) {
    if (!ctx || !cam)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->cam = *cam;
    ctx->has_camera = true;
    return RT_OK;
}

int rt_set_options(rt_ctx *ctx, const rt_options *opt) {
    if (!ctx || !opt)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!options_ok(opt))
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad options");
    ctx->opt = *opt;
    return RT_OK;
}

int rt_reserve(rt_ctx *ctx, int width, int height, int n_samples) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (width <= 0 || height <= 0 || n_samples < 1 || (long long)width * height > 0x7FFFFFFFll)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad width/height/sample count");
    auto reserve_one = [&](rt_ctx *c, int tile_world) -> int {
        RT_ON_DEVICE(c);
        Plan pl;
        pl.count = n_samples;
        pl.npix = (size_t)width * (size_t)height;
        const size_t tiles = (size_t)((width + 7) / 8) * (size_t)((height + 7) / 8);
        pl.items_per_sample = (tiles + (size_t)tile_world - 1) / (size_t)tile_world * 64;
        const int S = chunk_samples(pl, kMaxItems);
        // (for any scene and any options: two queues of the largest records)
        const size_t need_colors = (size_t)S * pl.items_per_sample, need_q[2] = {queue_records((size_t)S * pl.items_per_sample) * 6,
                                                                                 queue_records((size_t)S * pl.items_per_sample) * 6};
        const int chunks = (n_samples + S - 1) / S;
        for (int p = 0; p < (chunks > 1 || c->pipelining ? 2 : 1); p++) {
            const int rc = ensure_workspace(c, c->pipes[p], need_colors, need_q, pl.items_per_sample * (size_t)((S + 63) / 64));
            if (rc != RT_OK)
                return rc;
        }
        return ensure_image(c, width, height);
    };
    if (ctx->kids.empty())
        return reserve_one(ctx, ctx->opt.tile_world);
    const int n = (int)ctx->kids.size();
    for (rt_ctx *k : ctx->kids) {
        std::lock_guard<std::mutex> klock(k->mu);
        const int rc = reserve_one(k, ctx->opt.multi_split == RT_SPLIT_TILE ? ctx->opt.tile_world * n : ctx->opt.tile_world);
        if (rc != RT_OK)
            return fail(ctx, rc, "device " + std::to_string(k->device) + ": " + rt_last_error(k));
    }
    return RT_OK;
}

int rt_reset(rt_ctx *ctx) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_ON_DEVICE(ctx);
    {
        const int drc = discard_all(ctx);
        if (drc != RT_OK)
            return drc;
        const int jrc = join_pipes(ctx);
        if (jrc != RT_OK)
            return jrc;
    }
    for (int &s : ctx->kid_samples)
        s = 0;
    if (ctx->d_image)
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_image, 0, (size_t)ctx->img_w * ctx->img_h * sizeof(float4), ctx->stream));
    return RT_OK;
}

int rt_trace(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *out_rgba_mean,
             uint32_t *out_primary_id, rt_stats *stats) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    auto t0 = std::chrono::steady_clock::now();
    if (width <= 0 || height <= 0 || (long long)width * height > 0x7FFFFFFFll)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad width/height");
    if (!ctx->has_scene)
        return fail(ctx, RT_ERR_NO_SCENE, "Scene is not loaded.");
    RT_ON_DEVICE(ctx);
    const size_t px = (size_t)width * (size_t)height;
    {
        const int drc = discard_all(ctx); // a synchronous call takes over from the last ended call
        if (drc != RT_OK)
            return drc;
        const int jrc = join_pipes(ctx); // the image may still be the target of pipelined calls
        if (jrc != RT_OK)
            return jrc;
    }
    int rc = ensure_image(ctx, width, height);
    if (rc != RT_OK)
        return rc;
    const bool multi = !ctx->kids.empty();
    if (multi) {
        rc = multi_trace(ctx, width, height, first_sample, n_samples, ctx->d_image, out_primary_id ? ctx->d_primary : nullptr,
                         first_sample == 1 || ctx->opt.multi_split == RT_SPLIT_SAMPLE);
    } else {
        if (first_sample == 1)
            RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_image, 0, px * sizeof(float4), ctx->stream));
        if (out_primary_id)
            RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_primary, 0xFF, px * sizeof(uint32_t), ctx->stream));
        rc = launch_trace(ctx, width, height, first_sample, n_samples, ctx->d_image, out_primary_id ? ctx->d_primary : nullptr);
        if (rc == RT_OK)
            rc = join_pipes(ctx);
    }
    if (rc != RT_OK)
        return rc;
    if (out_rgba_mean)
        RT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgba_mean, ctx->d_image, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (out_primary_id)
        RT_HIP_CHECK(ctx, hipMemcpyAsync(out_primary_id, ctx->d_primary, px * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    rc = multi ? multi_sync(ctx, stats) : sync_and_stats(ctx, stats);
    if (rc != RT_OK)
        return rc;
    if (stats)
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

int rt_trace_device(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *d_rgba,
                    uint32_t *d_primary_id) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_ON_DEVICE(ctx);
    if (!ctx->kids.empty()) {
        if (!ctx->has_scene)
            return fail(ctx, RT_ERR_NO_SCENE, "Scene is not loaded.");
        if (!d_rgba)
            return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null image");
        {
            const int drc = discard_all(ctx);
            if (drc != RT_OK)
                return drc;
        }
        return multi_trace(ctx, width, height, first_sample, n_samples, (float4 *)d_rgba, d_primary_id,
                           ctx->opt.multi_split == RT_SPLIT_SAMPLE);
    }
    return launch_trace(ctx, width, height, first_sample, n_samples, (float4 *)d_rgba, d_primary_id);
}

int rt_trace_device_noalloc(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *d_rgba,
                            uint32_t *d_primary_id) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->kids.empty())
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_trace_device_noalloc works on a single-device context");
    RT_ON_DEVICE(ctx);
    return launch_trace_noalloc(ctx, width, height, first_sample, n_samples, (float4 *)d_rgba, d_primary_id);
}

uint64_t rt_noalloc_max_pixels(void) { return (uint64_t)kScratchMaxItems; }

int rt_trace_begin(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *out_rgba_mean) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (width <= 0 || height <= 0 || (long long)width * height > 0x7FFFFFFFll || n_samples < 1 || !out_rgba_mean)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad width/height/sample count, or no output buffer");
    if (ctx->begun.size() >= 2)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "two calls are already in flight: rt_trace_end one first");
    if (!ctx->has_scene)
        return fail(ctx, RT_ERR_NO_SCENE, "Scene is not loaded.");
    RT_ON_DEVICE(ctx);
    return ctx->kids.empty() ? begin_frame(ctx, width, height, first_sample, n_samples, out_rgba_mean)
                             : multi_begin(ctx, width, height, first_sample, n_samples, out_rgba_mean);
}

int rt_trace_end(rt_ctx *ctx, rt_stats *stats) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ctx->begun.empty())
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "rt_trace_end without rt_trace_begin");
    RT_ON_DEVICE(ctx);
    if (!ctx->kids.empty())
        return multi_end(ctx, stats);
    const int rc = end_frame(ctx);
    if (rc != RT_OK)
        return rc;
    if (stats) { // what has run so far: a call begun later may already have added to the counters
        std::memset(stats, 0, sizeof(*stats));
        return read_counters(ctx, stats);
    }
    return RT_OK;
}

int rt_trace_discard(rt_ctx *ctx) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_ON_DEVICE(ctx);
    return discard_all(ctx);
}

int rt_set_stream(rt_ctx *ctx, void *hip_stream) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_ON_DEVICE(ctx);
    {
        const int src = sync_all(ctx);
        if (src != RT_OK)
            return src;
    }
    if (ctx->own_stream && ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return RT_OK;
}

int rt_set_pipelining(rt_ctx *ctx, int on) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->kids.empty())
        return on ? fail(ctx, RT_ERR_INVALID_ARGUMENT, "overlapping calls is a feature of a single-device context") : RT_OK;
    RT_ON_DEVICE(ctx);
    const int jrc = join_pipes(ctx);
    if (jrc != RT_OK)
        return jrc;
    ctx->pipelining = on != 0;
    return RT_OK;
}

int rt_join(rt_ctx *ctx) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_ON_DEVICE(ctx);
    return join_pipes(ctx);
}

int rt_sync(rt_ctx *ctx, rt_stats *stats) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_ON_DEVICE(ctx);
    return ctx->kids.empty() ? sync_and_stats(ctx, stats) : multi_sync(ctx, stats);
}

int rt_layout_scene(const rt_node *nodes, uint32_t n_nodes, uint32_t root, const rt_tri *tris, uint32_t n_tris,
                    const rt_sphere *spheres, uint32_t n_spheres, const rt_material *mats, uint32_t n_mats, int order,
                    rt_layout_view *out) {
    if (!out)
        return RT_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    SceneLayout *L = new SceneLayout();
    std::string err;
    const int rc = build_layout(nodes, n_nodes, root, tris, n_tris, spheres, n_spheres, mats, n_mats, order, *L, err);
    if (rc != RT_OK) {
        delete L;
        return rc;
    }
    out->n_fast_nodes = L->n_fast;
    out->n_ref_nodes = L->n_ref;
    out->n_tri_slots = L->n_tri_slots;
    out->n_sphere_slots = L->n_sph_slots;
    out->fast_nodes = reinterpret_cast<const float *>(L->fn.data());
    out->ref_nodes = reinterpret_cast<const float *>(L->dn.data());
    out->tri_pos = reinterpret_cast<const float *>(L->tp.data());
    out->spheres = reinterpret_cast<const float *>(L->sp.data());
    if (L->compact) {
        out->n_grid_nodes = L->n_fast;
        out->grid_nodes = reinterpret_cast<const uint32_t *>(L->cn.data());
        out->leaf_records = reinterpret_cast<const float *>(L->lb.data());
        std::memcpy(out->grid_origin, L->q_origin, sizeof(out->grid_origin));
        std::memcpy(out->grid_step, L->q_step, sizeof(out->grid_step));
    }
    if (L->wide) {
        out->n_wide_nodes = L->n_wide;
        out->wide_nodes = reinterpret_cast<const uint32_t *>(L->wn.data());
        out->has_distance_limit = L->prune ? 1u : 0u;
        out->leaf_class = L->leaf_class.data();
        for (int k = 0; k < 3; k++) {
            out->limit_constants[k] = L->prune_c1[k];
            out->limit_constants[3 + k] = L->prune_c2x2[k];
            out->limit_constants[6 + k] = L->prune_c3[k];
        }
    }
    out->owner_ = L;
    return RT_OK;
}

void rt_layout_free(rt_layout_view *view) {
    if (view && view->owner_) {
        delete static_cast<SceneLayout *>(view->owner_);
        std::memset(view, 0, sizeof(*view));
    }
}

} // extern "C"

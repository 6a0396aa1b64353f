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

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
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
struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;
    std::string err;
    // scene in HBM
    float4 *d_fnodes = nullptr;
    uint32_t n_fnodes = 0;
    float4 *d_nodes = nullptr, *d_tri_pos = nullptr, *d_tri_nrm = nullptr, *d_spheres = nullptr, *d_mats = nullptr;
    uint32_t n_nodes = 0;
    bool has_scene = false;
    size_t scene_bytes = 0; // fast-tree records + triangle positions
    bool fast_exact = true; // the uploaded tree's interior boxes contain their children's (see build_layout)
    double expected_visits = 0.0; // sum of node box areas / root box area: nodes a random ray through the root meets
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
    // one HIP event pair per launch since the last sync (the kernel time is read back in rt_sync)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t pending = 0;
    int num_cus = 0;
    int blocks_per_cu[16] = {0}; // [nested | wavefront per-lane | wavefront wave-sync | wavefront fast][estimator][stats]
    // wavefront workspace: per-sample colour buffer, two ping-pong path queues, per-level cursors/counters
    // Two pipes take the chunks of the wavefront tracer in turn, each with its own stream and workspace, so that the
    // latency-bound end of one chunk (deep bounce levels: few rays, long walks) overlaps the throughput-bound start of
    // the next.  The folds into the running mean are chained by events and stay in sample order.
    struct Pipe {
        hipStream_t stream = nullptr;
        float4 *d_colors = nullptr, *d_q[2] = {nullptr, nullptr};
        unsigned int *d_levels = nullptr; // per level: 64 sub-queue slots of 32 words (count, batch cursor)
        size_t colors_cap = 0, q_cap = 0; // capacities in float4 / records
        hipEvent_t done = nullptr;        // recorded after the pipe's last fold
        bool used = false;                // `done` has been recorded at least once
    } pipes[2];
    int next_pipe = 0;
    hipEvent_t ev_start = nullptr;  // orders a call's chunks after what precedes the call on the context's stream
    hipEvent_t last_fold = nullptr; // `done` of the pipe that ran the latest fold
    bool pipelining = false;        // rt_set_pipelining: calls may overlap; rt_join / rt_sync order their results
    bool joined = true;             // the context's stream has been made to wait for both pipes
};

namespace {

int fail(rt_ctx *ctx, int code, const std::string &msg) {
    if (ctx)
        ctx->err = msg;
    return code;
}
#define RT_HIP_CHECK(ctx, expr)                                                                               \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,                   \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                   \
    } while (0)

void free_scene(rt_ctx *c) {
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_fnodes) (void)hipFree(c->d_fnodes);
    c->d_fnodes = nullptr;
    c->n_fnodes = 0;
    if (c->d_tri_pos) (void)hipFree(c->d_tri_pos);
    if (c->d_tri_nrm) (void)hipFree(c->d_tri_nrm);
    if (c->d_spheres) (void)hipFree(c->d_spheres);
    if (c->d_mats) (void)hipFree(c->d_mats);
    c->d_nodes = c->d_tri_pos = c->d_tri_nrm = c->d_spheres = c->d_mats = nullptr;
    c->n_nodes = 0;
    c->has_scene = false;
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
    hipError_t e = KERNEL == 3   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_path<OLD, STATS, WALK_FAST>, 256, 0)
                   : KERNEL == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_path<OLD, STATS, WALK_WAVE>, 256, 0)
                   : KERNEL == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_path<OLD, STATS, WALK_LANE>, 256, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_trace<OLD, STATS>, 256, 0);
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

int launch_trace(rt_ctx *ctx, int W, int H, int first, int count, float4 *d_img, uint32_t *d_prim) {
    if (!ctx->has_scene)
        return fail(ctx, RT_ERR_NO_SCENE, "Scene is not loaded.");
    if (!ctx->has_camera)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "camera not set");
    if (W <= 0 || H <= 0 || first < 1 || count < 0 || (long long)W * H > 0x7FFFFFFFll)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad width/height/sample range");
    if (d_img == nullptr)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "null image");
    const rt_options &o = ctx->opt;
    KParams P;
    P.nodes = ctx->d_nodes;
    P.fnodes = ctx->d_fnodes;
    P.n_fnodes = ctx->n_fnodes;
    P.tri_pos = ctx->d_tri_pos;
    P.tri_nrm = ctx->d_tri_nrm;
    P.spheres = ctx->d_spheres;
    P.mats = ctx->d_mats;
    P.image = d_img;
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

    RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_queue, 0, 64, ctx->stream));
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
    // RT_TRAV_FAST: lane re-fill pays when walks are long and their fetches miss the L2 -- scenes whose fast tree and
    // triangle positions exceed the L2s; smaller scenes take k_path with the noted primitive tests.
    const bool refill = trav == RT_TRAV_FAST_REFILL || (trav == RT_TRAV_FAST && ctx->scene_bytes > kRefillSceneBytes);
    // ... among those, the flat scene in slices (measured +5 %; the sphere scene loses 7 % in slices)
    const bool sliced = trav == RT_TRAV_FAST_SLICED || (trav == RT_TRAV_FAST && !refill && !old);
    const bool queued = trav == RT_TRAV_FAST_QUEUED || (trav == RT_TRAV_FAST && !refill && !sliced);
    if (trav == RT_TRAV_FAST_QUEUED || trav == RT_TRAV_FAST_INLINE || trav == RT_TRAV_FAST_SLICED)
        trav = RT_TRAV_FAST;
    if (trav == RT_TRAV_FAST_REFILL)
        trav = RT_TRAV_FAST;
    const int walk = trav == RT_TRAV_WAVE ? WALK_WAVE : (trav == RT_TRAV_FAST || trav == RT_TRAV_FAST_REPLAY) ? WALK_FAST
                     : trav == RT_TRAV_BRUTE                                                               ? WALK_BRUTE
                                                                                                           : WALK_LANE;
    P.wave_min_rays = o.traversal == RT_TRAV_WAVE ? 0u : 65536u;
    const int variant = (trav == RT_TRAV_NESTED ? 0 : 4) + (old ? 2 : 0) + (st ? 1 : 0);
    const uint32_t full_grid = (uint32_t)ctx->num_cus * (uint32_t)ctx->blocks_per_cu[variant + (variant >= 4 ? 4 * (walk == WALK_BRUTE ? WALK_LANE : walk) : 0)];
    if (ctx->pending == ctx->events.size()) {
        hipEvent_t a, b;
        RT_HIP_CHECK(ctx, hipEventCreate(&a));
        RT_HIP_CHECK(ctx, hipEventCreate(&b));
        ctx->events.emplace_back(a, b);
    }
    dim3 block(256);
    if (trav == RT_TRAV_NESTED) {
        {
            const int jrc = join_pipes(ctx); // earlier pipelined calls may still be folding into this image
            if (jrc != RT_OK)
                return jrc;
        }
        uint32_t blocks = full_grid;
        if (blocks > (P.n_my_tiles + 3) / 4)
            blocks = (P.n_my_tiles + 3) / 4;
        if (blocks < 1)
            blocks = 1;
        dim3 grid(blocks);
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].first, ctx->stream));
        switch (variant) {
        case 0: hipLaunchKernelGGL((k_trace<false, false>), grid, block, 0, ctx->stream, P); break;
        case 1: hipLaunchKernelGGL((k_trace<false, true>), grid, block, 0, ctx->stream, P); break;
        case 2: hipLaunchKernelGGL((k_trace<true, false>), grid, block, 0, ctx->stream, P); break;
        default: hipLaunchKernelGGL((k_trace<true, true>), grid, block, 0, ctx->stream, P); break;
        }
        RT_HIP_CHECK(ctx, hipGetLastError());
        RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].second, ctx->stream));
        ctx->pending++;
        return RT_OK;
    }
    // ---- wavefront: samples in chunks of S; per chunk one k_path launch per bounce level, then k_fold ----
    const size_t npix = (size_t)W * (size_t)H;
    const size_t items_per_sample = (size_t)P.n_my_tiles * 64;
    int S = count < kChunkSamples ? count : kChunkSamples;
    if (items_per_sample > 0 && (size_t)S * items_per_sample > kMaxItems)
        S = (int)(kMaxItems / items_per_sample);
    if (S < 1)
        S = 1;
    // queue capacity: a sub-queue of the next level receives the continuing paths (at most 64) of every 64th batch of
    // level 0, or of every 64th batch of each of this level's 64 sub-queues; no level has more batches than level 0
    // has plus one partial batch per sub-queue, so (n_batches / 64 + 66) * 64 records per sub-queue always suffice
    const size_t need_colors = (size_t)S * npix, need_q = (size_t)S * items_per_sample + (size_t)kSubQueues * 66 * 64;
    int levels = old ? o.bounce_limit + 1 : (o.bounce_limit > 1 ? o.bounce_limit : 1);
    if (sliced) { // rounds, not bounce levels: a walk may take several; the last launch finishes whatever is left
        levels += kSliceExtraRounds;
        if (levels > kMaxLevels - 1)
            levels = kMaxLevels - 1;
    }
    P.slice = kSlice;
    P.tail_rays = kTailRays;
    const size_t rec4 = sliced ? 6 : 4; // float4 per queue record
    P.npix = (uint32_t)npix;
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].first, ctx->stream));
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    hipStream_t last_stream = ctx->stream;
    for (int done = 0; done < count || (count == 0 && done == 0); done += S) {
        const int s_now = count - done < S ? count - done : S;
        if (s_now <= 0)
            break;
        rt_ctx::Pipe &pp = ctx->pipes[ctx->next_pipe];
        ctx->next_pipe ^= 1;
        if (need_colors > pp.colors_cap || need_q * rec4 > pp.q_cap) {
            RT_HIP_CHECK(ctx, hipStreamSynchronize(pp.stream));
            if (pp.d_colors) (void)hipFree(pp.d_colors);
            if (pp.d_q[0]) (void)hipFree(pp.d_q[0]);
            if (pp.d_q[1]) (void)hipFree(pp.d_q[1]);
            pp.d_colors = pp.d_q[0] = pp.d_q[1] = nullptr;
            pp.colors_cap = pp.q_cap = 0;
            RT_HIP_CHECK(ctx, hipMalloc((void **)&pp.d_colors, need_colors * sizeof(float4)));
            RT_HIP_CHECK(ctx, hipMalloc((void **)&pp.d_q[0], (need_q ? need_q : 1) * rec4 * sizeof(float4)));
            RT_HIP_CHECK(ctx, hipMalloc((void **)&pp.d_q[1], (need_q ? need_q : 1) * rec4 * sizeof(float4)));
            pp.colors_cap = need_colors;
            pp.q_cap = need_q * rec4;
        }
        RT_HIP_CHECK(ctx, hipStreamWaitEvent(pp.stream, ctx->ev_start, 0));
        RT_HIP_CHECK(ctx, hipMemsetAsync(pp.d_levels, 0, (size_t)(levels + 1) * kSubQueues * 32 * sizeof(unsigned int), pp.stream));
        P.colors = pp.d_colors;
        P.S = s_now;
        P.div_per_tile = make_fastdiv(64u * (uint32_t)s_now);
        P.chunk_first = first + done;
        P.n_items = (uint32_t)((size_t)s_now * items_per_sample);
        P.sub_cap = (((P.n_items + 63u) / 64u) / (uint32_t)kSubQueues + 66u) * 64u;
        for (int level = 0; level < levels; level++) {
            P.level = level;
            P.tail = level == levels - 1 ? 1 : 0;
            P.q_in = pp.d_q[level & 1];
            P.q_out = pp.d_q[(level + 1) & 1];
            P.sub_in = pp.d_levels + (size_t)level * kSubQueues * 32;
            P.sub_cursor = pp.d_levels + (size_t)level * kSubQueues * 32 + 1;
            P.sub_out = pp.d_levels + (size_t)(level + 1) * kSubQueues * 32;
            uint32_t blocks = full_grid;
            if (level == 0 && blocks > (P.n_items / 64 + 3) / 4)
                blocks = (P.n_items / 64 + 3) / 4;
            if (blocks < 1)
                blocks = 1;
            dim3 grid(blocks);
            // wave-synchronous walk: register allocation capped at 64 VGPRs (8 waves per SIMD hide the scalar
            // fetch latency; the spills land in the shading code, which runs once per ray); the others: uncapped
#define RT_LAUNCH_PATH(OLD_, STATS_)                                                                                  \
    do {                                                                                                              \
        if (walk == WALK_FAST && sliced) hipLaunchKernelGGL((k_sliced<OLD_, STATS_>), grid, block, 0, pp.stream, P); \
        else if (walk == WALK_FAST && queued) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_QUEUED>), grid, block, 0, pp.stream, P); \
        else if (walk == WALK_WAVE) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_WAVE>), grid, block, 0, pp.stream, P); \
        else if (walk == WALK_FAST && refill) hipLaunchKernelGGL((k_refill<OLD_, STATS_>), grid, block, 0, pp.stream, P); \
        else if (walk == WALK_FAST) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_FAST>), grid, block, 0, pp.stream, P); \
        else if (walk == WALK_BRUTE) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_BRUTE>), grid, block, 0, pp.stream, P); \
        else hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_LANE>), grid, block, 0, pp.stream, P);                       \
    } while (0)
            switch (variant) {
            case 4: RT_LAUNCH_PATH(false, false); break;
            case 5: RT_LAUNCH_PATH(false, true); break;
            case 6: RT_LAUNCH_PATH(true, false); break;
            default: RT_LAUNCH_PATH(true, true); break;
            }
#undef RT_LAUNCH_PATH
            RT_HIP_CHECK(ctx, hipGetLastError());
        }
        // the running mean takes the samples in order: this fold waits for the one before it, whichever pipe ran it
        if (ctx->last_fold != nullptr && ctx->last_fold != pp.done)
            RT_HIP_CHECK(ctx, hipStreamWaitEvent(pp.stream, ctx->last_fold, 0));
        hipLaunchKernelGGL(k_fold, dim3((unsigned)((npix + 255) / 256)), block, 0, pp.stream, d_img, pp.d_colors,
                           (uint32_t)npix, W, P.tiles_x, o.tile_rank, o.tile_world, P.chunk_first, s_now);
        RT_HIP_CHECK(ctx, hipGetLastError());
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
    RT_HIP_CHECK(ctx, hipEventRecord(ctx->events[ctx->pending].second, last_stream));
    ctx->pending++;
    return RT_OK;
}

int sync_and_stats(rt_ctx *ctx, rt_stats *stats) {
    {
        const int src = sync_all(ctx);
        if (src != RT_OK)
            return src;
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        if (ctx->pending > 0) {
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
    if (ctx->pending > 0) // counters run from one sync to the next
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, kCounterSlots * kCounterStride * sizeof(unsigned long long), ctx->stream));
    ctx->pending = 0;
    return RT_OK;
}

#include "scene_layout.hpp"

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

const char *rt_version(void) { return "rt_hip 0.1 gfx950"; }

int rt_create(rt_ctx **out, int device) {
    if (!out)
        return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return RT_ERR_NO_DEVICE;
    rt_ctx *ctx = new rt_ctx();
    ctx->device = device;
    std::memset(&ctx->cam, 0, sizeof(ctx->cam));
    ctx->opt = rt_options{4, RT_EST_SD_LOOP, 1u, 0.001f, 0, 1, 0, 0u, RT_TRAV_AUTO};
    auto bail = [&](const char *what) {
        std::fprintf(stderr, "rt_create: %s failed\n", what);
        delete ctx;
        return RT_ERR_HIP;
    };
    if (hipSetDevice(device) != hipSuccess) return bail("hipSetDevice");
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
    ctx->own_stream = true;
    if (hipMalloc((void **)&ctx->d_queue, 64) != hipSuccess) return bail("hipMalloc");
    if (hipMalloc((void **)&ctx->d_counters, kCounterSlots * kCounterStride * sizeof(unsigned long long)) != hipSuccess) return bail("hipMalloc");
    if (hipMemset(ctx->d_counters, 0, kCounterSlots * kCounterStride * sizeof(unsigned long long)) != hipSuccess) return bail("hipMemset");
    if (hipEventCreateWithFlags(&ctx->ev_start, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
    for (auto &pp : ctx->pipes) {
        if (hipStreamCreateWithFlags(&pp.stream, hipStreamNonBlocking) != hipSuccess) return bail("hipStreamCreate");
        if (hipEventCreateWithFlags(&pp.done, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
        if (hipMalloc((void **)&pp.d_levels, (size_t)(kMaxLevels + 1) * kSubQueues * 32 * sizeof(unsigned int)) != hipSuccess) return bail("hipMalloc");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return bail("hipGetDeviceProperties");
    ctx->num_cus = prop.multiProcessorCount;
    occupancy<0, false, false>(&ctx->blocks_per_cu[0]);
    occupancy<0, false, true>(&ctx->blocks_per_cu[1]);
    occupancy<0, true, false>(&ctx->blocks_per_cu[2]);
    occupancy<0, true, true>(&ctx->blocks_per_cu[3]);
    occupancy<1, false, false>(&ctx->blocks_per_cu[4]);
    occupancy<1, false, true>(&ctx->blocks_per_cu[5]);
    occupancy<1, true, false>(&ctx->blocks_per_cu[6]);
    occupancy<1, true, true>(&ctx->blocks_per_cu[7]);
    occupancy<2, false, false>(&ctx->blocks_per_cu[8]);
    occupancy<2, false, true>(&ctx->blocks_per_cu[9]);
    occupancy<2, true, false>(&ctx->blocks_per_cu[10]);
    occupancy<2, true, true>(&ctx->blocks_per_cu[11]);
    occupancy<3, false, false>(&ctx->blocks_per_cu[12]);
    occupancy<3, false, true>(&ctx->blocks_per_cu[13]);
    occupancy<3, true, false>(&ctx->blocks_per_cu[14]);
    occupancy<3, true, true>(&ctx->blocks_per_cu[15]);
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx *ctx) {
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    for (auto &pp : ctx->pipes)
        if (pp.stream)
            (void)hipStreamSynchronize(pp.stream);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    if (ctx->d_image) (void)hipFree(ctx->d_image);
    if (ctx->d_primary) (void)hipFree(ctx->d_primary);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    for (auto &pp : ctx->pipes) {
        if (pp.d_levels) (void)hipFree(pp.d_levels);
        if (pp.d_colors) (void)hipFree(pp.d_colors);
        if (pp.d_q[0]) (void)hipFree(pp.d_q[0]);
        if (pp.d_q[1]) (void)hipFree(pp.d_q[1]);
        if (pp.done) (void)hipEventDestroy(pp.done);
        if (pp.stream) (void)hipStreamDestroy(pp.stream);
    }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    for (auto &e : ctx->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rt_last_error(rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

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
    const std::vector<float4> &dn = L.dn, &fn = L.fn, &tp = L.tp, &tn = L.tn, &sp = L.sp, &mt = L.mt;
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    {
        const int src = sync_all(ctx); // nothing may still be walking the old scene
        if (src != RT_OK)
            return src;
    }
    free_scene(ctx);
    int rc;
    if ((rc = upload(ctx, &ctx->d_nodes, dn)) != RT_OK || (rc = upload(ctx, &ctx->d_fnodes, fn)) != RT_OK ||
        (rc = upload(ctx, &ctx->d_tri_pos, tp)) != RT_OK ||
        (rc = upload(ctx, &ctx->d_tri_nrm, tn)) != RT_OK || (rc = upload(ctx, &ctx->d_spheres, sp)) != RT_OK ||
        (rc = upload(ctx, &ctx->d_mats, mt)) != RT_OK) {
        free_scene(ctx);
        return rc;
    }
    ctx->expected_visits = L.expected_visits;
    ctx->fast_exact = L.nested;
    ctx->scene_bytes = (L.fn.size() + L.tp.size()) * sizeof(float4);
    ctx->n_nodes = L.n_ref;
    ctx->n_fnodes = L.n_fast;
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
    if (opt->bounce_limit < 0 || opt->bounce_limit > kMaxLevels - 2 || opt->tile_world < 1 || opt->tile_rank < 0 || opt->tile_rank >= opt->tile_world ||
        (opt->estimator != RT_EST_SD_LOOP && opt->estimator != RT_EST_OLD_RECURSIVE) || opt->traversal < 0 ||
        opt->traversal > RT_TRAV_FAST_SLICED)
        return fail(ctx, RT_ERR_INVALID_ARGUMENT, "bad options");
    ctx->opt = *opt;
    return RT_OK;
}

int rt_reset(rt_ctx *ctx) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    {
        const int jrc = join_pipes(ctx);
        if (jrc != RT_OK)
            return jrc;
    }
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
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)width * (size_t)height;
    {
        const int jrc = join_pipes(ctx); // the image may still be the target of pipelined calls
        if (jrc != RT_OK)
            return jrc;
    }
    if (width != ctx->img_w || height != ctx->img_h || !ctx->d_image) {
        RT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_image) (void)hipFree(ctx->d_image);
        if (ctx->d_primary) (void)hipFree(ctx->d_primary);
        ctx->d_image = nullptr;
        ctx->d_primary = nullptr;
        ctx->img_w = ctx->img_h = 0;
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_image, px * sizeof(float4)));
        RT_HIP_CHECK(ctx, hipMalloc((void **)&ctx->d_primary, px * sizeof(uint32_t)));
        ctx->img_w = width;
        ctx->img_h = height;
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_image, 0, px * sizeof(float4), ctx->stream));
    }
    if (first_sample == 1)
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_image, 0, px * sizeof(float4), ctx->stream));
    if (out_primary_id)
        RT_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_primary, 0xFF, px * sizeof(uint32_t), ctx->stream));
    int rc = launch_trace(ctx, width, height, first_sample, n_samples, ctx->d_image, out_primary_id ? ctx->d_primary : nullptr);
    if (rc != RT_OK)
        return rc;
    rc = join_pipes(ctx);
    if (rc != RT_OK)
        return rc;
    if (out_rgba_mean)
        RT_HIP_CHECK(ctx, hipMemcpyAsync(out_rgba_mean, ctx->d_image, px * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    if (out_primary_id)
        RT_HIP_CHECK(ctx, hipMemcpyAsync(out_primary_id, ctx->d_primary, px * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    rc = sync_and_stats(ctx, stats);
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
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return launch_trace(ctx, width, height, first_sample, n_samples, (float4 *)d_rgba, d_primary_id);
}

int rt_set_stream(rt_ctx *ctx, void *hip_stream) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
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
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
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
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return join_pipes(ctx);
}

int rt_sync(rt_ctx *ctx, rt_stats *stats) {
    if (!ctx)
        return RT_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(ctx->mu);
    RT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return sync_and_stats(ctx, stats);
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

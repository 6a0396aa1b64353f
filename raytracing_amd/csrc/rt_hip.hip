// rt_hip.hip -- MI355X (gfx950) path tracer: device kernels and the C ABI of include/rt_hip.h.
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

// ------------------------------------------------------------------ fp32 vector helpers (glm order)
struct f3 {
    float x, y, z;
};
RT_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RT_HD f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
RT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
RT_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD f3 cross3(f3 a, f3 b) { return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
RT_HD f3 normalize3(f3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
// mat3(c0,c1,c2) * v, row by row
RT_HD f3 basis_mul(f3 c0, f3 c1, f3 c2, f3 v) {
    return f3{(c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z,
              (c0.z * v.x + c1.z * v.y) + c2.z * v.z};
}

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kSphereBit = 0x80000000u;
constexpr uint32_t KIND_INTERNAL = 0, KIND_TRI = 1, KIND_MESH = 2, KIND_SPHERE = 4;
constexpr int kSubQueues = 64;             // a level's path queue is 64 sub-queues, each with its own append counter
constexpr int kCounterSlots = 64, kCounterStride = 16; // statistics are added into 64 slots of 16 values, summed by the host
constexpr int kMaxLevels = 64;             // bounce levels a wavefront trace can launch (bounce_limit <= 62)
constexpr uint32_t kTailRays = 256u << 10;  // a level with no more rays than this follows every path to its end in one launch
constexpr int kChunkSamples = 64;          // samples per pixel traced per wavefront chunk
constexpr size_t kMaxItems = 192u << 20;   // work items (pixel samples) per chunk: bounds the queue workspace

// ------------------------------------------------------------------ kernel parameters
struct KParams {
    const float4 *nodes; // 2 x float4 per node, reference visiting order
    const float4 *fnodes; // 2 x float4 per node of the fast tree (pre-order, exact-union boxes over the reference's leaf boxes)
    const float4 *tri_pos; // 3 x float4 per triangle slot: p0, e1, e2, prim id, material
    const float4 *tri_nrm; // 3 x float4 per triangle slot: n0, n1, n2
    const float4 *spheres; // 2 x float4 per sphere
    const float4 *mats;    // 2 x float4 per material
    float4 *image;
    uint32_t *primary;
    unsigned int *queue;
    unsigned long long *counters; // rays, box, tri, sph, tri_hits, replays
    uint32_t n_nodes;
    uint32_t n_fnodes;
    int tail;         // k_path: the last level (follow every path to its end; nothing is queued)
    int force_replay; // fast walk: resolve every ray by the ordered replay (verification aid)
    uint32_t n_my_tiles;
    int W, H, tiles_x;
    FastDiv div_tiles_x, div_per_tile; // exact division by tiles_x and by 64 * S (camera_path)
    uint32_t seed_hash;                // pcg(seed): the part of the per-pixel RNG key that does not depend on the pixel
    int tile_rank, tile_world;
    int first_sample, n_samples;
    int bounce_limit;
    uint32_t seed;
    uint32_t sample_offset;
    uint32_t wave_min_rays;
    // wavefront launches (k_path)
    const float4 *q_in;         // path records of this level (4 x float4 each)
    float4 *q_out;              // path records of the next level
    float4 *colors;             // colors[sample in chunk][pixel]
    const unsigned int *sub_in; // levels > 0: record counts of this level's sub-queues (one per 128 bytes)
    unsigned int *sub_cursor;   // levels > 0: batch cursors of this level's sub-queues (one per 128 bytes)
    unsigned int *sub_out;      // append counters of the next level's sub-queues
    uint32_t sub_cap;           // records a sub-queue can hold; sub-queue q starts at record q * sub_cap
    uint32_t n_items;           // level 0: n_my_tiles * 64 * S
    uint32_t npix;
    int level, S, chunk_first;
    float jitter;
    f3 cam_pos, bx, by, bz;
    float cam_w, cam_h, focal;
};

// ------------------------------------------------------------------ counter RNG
RT_HD uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
RT_D float rng01(uint32_t key, uint32_t dim) {
    return (float)(pcg(key + dim * 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
}
// dimensions: 0..2 camera-ray jitter; bounce b uses 3+4b (roulette), 4+4b, 5+4b (cosine sample), 6+4b (dielectric)

// cos/sin of the cosine-hemisphere sample (Include/Random.hpp:52-53) from fp32 add/mul only:
// Cody-Waite reduction by pi/2 and the classic minimax polynomials, <= 2 ulp on [0, 2pi].
RT_D void sincos_poly(float th, float &s, float &c) {
    int q = (int)(th * 0.636619772f + 0.5f);
    float qf = (float)q;
    float r = ((th - qf * 1.5703125f) - qf * 4.837512969970703125e-4f) - qf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = r + (r * z) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    float cp = (1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    int k = q & 3;
    s = (k == 0) ? sp : (k == 1) ? cp : (k == 2) ? -sp : -cp;
    c = (k == 0) ? cp : (k == 1) ? -sp : (k == 2) ? -cp : sp;
}

// Random::GenerateCosineSemiSphereVector, Include/Random.hpp:41-72
RT_D f3 cosine_dir(f3 n, float u1, float u2) {
    float r = sqrtf(u1);
    float theta = 6.28318530717958647692f * u2;
    float sn, cs;
    sincos_poly(theta, sn, cs);
    float x = r * cs;
    float y = r * sn;
    float z = sqrtf(1.0f - u1);
    f3 T;
    if (fabsf(n.z) < 0.99f)
        T = normalize3(cross3(n, mk3(0.0f, 0.0f, 1.0f)));
    else
        T = normalize3(cross3(n, mk3(1.0f, 0.0f, 0.0f)));
    f3 B = cross3(T, n);
    return basis_mul(T, B, n, mk3(x, y, z));
}

// Sky gradient, CPURayTrace/Trace.cpp:86-89 == Materials/Sky.hpp:12-14
RT_D f3 sky_color(f3 dir) {
    f3 unit = normalize3(dir);
    float a = 0.5f * (unit.y + 1.0f);
    return (1.0f - a) * mk3(1.0f, 1.0f, 1.0f) + a * mk3(0.5f, 0.7f, 1.0f);
}

struct Best {
    float t, u, v;
    uint32_t ref; // triangle slot, kSphereBit | sphere index, or kNone
};
struct Cnt {
    unsigned long long rays, box, tri, sph, trih, replay;
};

// Closest hit in the reference's own visiting order.  One iteration = one node of the reference
// traversal: slab test against (1e-6, closest.t) (SimplifiedData.cpp:443, BVH.hpp:129), then for a
// leaf the primitive test and the strict "t < closest.t" update (SimplifiedData.cpp:452, BVH.hpp:137).
// The per-axis early-outs of the reference slab test only stop early; the accept/reject decision is
// the final "tMax <= tMin" comparison because tMin never decreases and tMax never increases.
template <bool OLD, bool STATS>
RT_D void closest_hit(const KParams &P, const f3 o, const f3 d, const f3 inv, Best &best, Cnt &cnt) {
    best.t = __builtin_inff();
    best.u = 0.0f;
    best.v = 0.0f;
    best.ref = kNone;
    Best outer = best;         // OLD: the scene-level closest while inside a mesh's inner BVH
    uint32_t inner_end = kNone; // OLD: first node index after the current mesh's inner BVH
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const uint32_t n = P.n_nodes * 32u; // node records are addressed by byte offset (32 bytes each)
    uint32_t idx = 0;
    while (idx < n) {
        if (OLD) {
            if (idx == inner_end) { // Mesh::intersect returned: keep it only if strictly closer (BVH.hpp:137)
                if (!(best.t < outer.t))
                    best = outer;
                inner_end = kNone;
            }
        }
        const float4 *rec = (const float4 *)((const char *)P.nodes + idx);
        const float4 a = rec[0];
        const float4 b = rec[1];
        const uint32_t skip = __float_as_uint(b.z);
        const uint32_t kind = __float_as_uint(b.w);
        if (STATS)
            cnt.box++;
        float t0x = (a.x - o.x) * inv.x, t1x = (a.w - o.x) * inv.x;
        float t0y = (a.y - o.y) * inv.y, t1y = (b.x - o.y) * inv.y;
        float t0z = (a.z - o.z) * inv.z, t1z = (b.y - o.z) * inv.z;
        float nearx = nx ? t1x : t0x, farx = nx ? t0x : t1x;
        float neary = ny ? t1y : t0y, fary = ny ? t0y : t1y;
        float nearz = nz ? t1z : t0z, farz = nz ? t0z : t1z;
        // "t0 > tMin ? t0 : tMin" keeps tMin when t0 is NaN, as fmaxf(t0, tMin) does; likewise fminf
        float tmin = fmaxf(nearz, fmaxf(neary, fmaxf(nearx, 1e-6f)));
        float tmax = fminf(farz, fminf(fary, fminf(farx, best.t)));
        if (tmax <= tmin) {
            idx = skip;
            continue;
        }
        idx = idx + 32u;
        const uint32_t k = kind & 7u;
        if (k == KIND_INTERNAL)
            continue;
        const uint32_t slot = kind >> 3;
        if (k == KIND_TRI) {
            if (STATS)
                cnt.tri++;
            const float4 q0 = P.tri_pos[3 * slot], q1 = P.tri_pos[3 * slot + 1], q2 = P.tri_pos[3 * slot + 2];
            const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
            const float EPSILON = 1e-7f;
            f3 h = cross3(d, e2);
            float det = dot3(e1, h);
            if (!(det > -EPSILON && det < EPSILON)) {
                float f = 1.0f / det;
                f3 s = o - p0;
                float u = f * dot3(s, h);
                if (!(u < 0.0f || u > 1.0f)) {
                    f3 q = cross3(s, e1);
                    float v = f * dot3(d, q);
                    if (!(v < 0.0f || u + v > 1.0f)) {
                        float t = f * dot3(e2, q);
                        if (t > EPSILON && t < best.t) {
                            best.t = t;
                            best.u = u;
                            best.v = v;
                            best.ref = slot;
                        }
                    }
                }
            }
        } else if (OLD && k == KIND_SPHERE) { // Sphere::intersect, Objects.hpp:43-72
            if (STATS)
                cnt.sph++;
            const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
            f3 oc = mk3(s0.x, s0.y, s0.z) - o;
            float qa = dot3(d, d);
            float qb = -2.0f * dot3(d, oc);
            float qc = dot3(oc, oc) - s0.w * s0.w;
            float disc = qb * qb - 4 * qa * qc;
            if (!(disc < 0)) {
                float t = (-qb - sqrtf(disc)) / (2.0f * qa);
                bool ok = !(t < 0);
                // a ray that starts inside never hits (Objects.hpp:57-60), except for the dielectric
                // spheres this build adds, which need the exit point
                if (!ok && __float_as_uint(s1.z) != 0u) {
                    t = (-qb + sqrtf(disc)) / (2.0f * qa);
                    ok = !(t < 0);
                }
                if (ok && t < best.t) {
                    best.t = t;
                    best.ref = kSphereBit | slot;
                }
            }
        } else if (OLD && k == KIND_MESH) { // Mesh::intersect starts a fresh closest hit (Mesh.hpp:154-160)
            outer = best;
            inner_end = skip;
            best.t = __builtin_inff();
            best.ref = kNone;
        }
    }
    if (OLD) {
        if (inner_end != kNone) {
            if (!(best.t < outer.t))
                best = outer;
        }
    }
}

// "BVH acceleration" switched off (BVHSettings::toggleBVHAccel, Program/BVHUI.hpp:14): every primitive is tested, no
// box is.  Old scene: Scene::intersectClosest (Program/Scene.hpp:83-95) takes the objects in the order Scene::update
// left them in -- BVH::build sorts that very vector (BVH.hpp:116-119, :54-65), so it is the left-to-right order of the
// scene-level leaves, which is the order of the records here -- and accepts "t < closest.t"; a mesh without its inner
// BVH (Mesh::intersect, Mesh.hpp:161-172) takes its triangles in storage order, which the strict comparison turns into
// the smallest (t, triangle index).  The flat scene has no such switch in the reference; it is given the same meaning:
// all triangles in storage order.
template <bool OLD, bool STATS>
RT_D void closest_hit_brute(const KParams &P, const f3 o, const f3 d, Best &best, Cnt &cnt) {
    best = Best{__builtin_inff(), 0.0f, 0.0f, kNone};
    uint32_t bidx = kNone;      // storage index of the triangle `best` holds
    Best outer = best;          // OLD: the scene-level closest while inside a mesh
    uint32_t inner_end = kNone; // OLD: first record after the current mesh's triangles
    const uint32_t n = P.n_nodes * 32u;
    for (uint32_t idx = 0; idx < n; idx += 32u) {
        if (OLD) {
            if (idx == inner_end) {
                if (!(best.t < outer.t))
                    best = outer;
                inner_end = kNone;
                bidx = kNone;
            }
        }
        const float4 b = ((const float4 *)((const char *)P.nodes + idx))[1];
        const uint32_t kind = __float_as_uint(b.w);
        const uint32_t k = kind & 7u;
        const uint32_t slot = kind >> 3;
        if (k == KIND_TRI) {
            if (STATS)
                cnt.tri++;
            const float4 q0 = P.tri_pos[3 * slot], q1 = P.tri_pos[3 * slot + 1], q2 = P.tri_pos[3 * slot + 2];
            const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
            const float EPSILON = 1e-7f;
            f3 h = cross3(d, e2);
            float det = dot3(e1, h);
            if (!(det > -EPSILON && det < EPSILON)) {
                float f = 1.0f / det;
                f3 s = o - p0;
                float u = f * dot3(s, h);
                if (!(u < 0.0f || u > 1.0f)) {
                    f3 q = cross3(s, e1);
                    float v = f * dot3(d, q);
                    if (!(v < 0.0f || u + v > 1.0f)) {
                        float t = f * dot3(e2, q);
                        const uint32_t ti = __float_as_uint(q2.y);
                        const bool in_order_group = !OLD || inner_end != kNone; // triangles scanned in storage order
                        if (t > EPSILON && (t < best.t || (in_order_group && t == best.t && t < __builtin_inff() && ti < bidx))) {
                            best = Best{t, u, v, slot};
                            bidx = ti;
                        }
                    }
                }
            }
        } else if (OLD && k == KIND_SPHERE) { // Sphere::intersect, Objects.hpp:43-72
            if (STATS)
                cnt.sph++;
            const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
            f3 oc = mk3(s0.x, s0.y, s0.z) - o;
            float qa = dot3(d, d);
            float qb = -2.0f * dot3(d, oc);
            float qc = dot3(oc, oc) - s0.w * s0.w;
            float disc = qb * qb - 4 * qa * qc;
            if (!(disc < 0)) {
                float t = (-qb - sqrtf(disc)) / (2.0f * qa);
                bool ok = !(t < 0);
                if (!ok && __float_as_uint(s1.z) != 0u) {
                    t = (-qb + sqrtf(disc)) / (2.0f * qa);
                    ok = !(t < 0);
                }
                if (ok && t < best.t) {
                    best = Best{t, 0.0f, 0.0f, kSphereBit | slot};
                    bidx = kNone;
                }
            }
        } else if (OLD && k == KIND_MESH) { // Mesh::intersect starts a fresh closest hit (Mesh.hpp:154-160)
            outer = best;
            inner_end = __float_as_uint(b.z);
            best = Best{__builtin_inff(), 0.0f, 0.0f, kNone};
            bidx = kNone;
        }
    }
    if (OLD) {
        if (inner_end != kNone) {
            if (!(best.t < outer.t))
                best = outer;
        }
    }
}

struct RayState {
    f3 o, d, inv;
};
RT_D void set_ray(RayState &r, f3 o, f3 d) { // Ray::Ray, CPURayTrace/Ray.hpp:8
    r.o = o;
    r.d = d;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

RT_D f3 tri_normal(const KParams &P, uint32_t slot, float u, float v) { // SimplifiedData.cpp:293
    const float4 m0 = P.tri_nrm[3 * slot], m1 = P.tri_nrm[3 * slot + 1], m2 = P.tri_nrm[3 * slot + 2];
    const f3 n0 = mk3(m0.x, m0.y, m0.z), n1 = mk3(m0.w, m1.x, m1.y), n2 = mk3(m1.z, m1.w, m2.x);
    return normalize3((1 - u - v) * n0 + u * n1 + v * n2);
}

RT_D float reflectance(float cosine, float ri) {
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    float m = 1.0f - cosine;
    float m2 = m * m;
    return r0 + (1.0f - r0) * ((m2 * m2) * m);
}

// One pixel sample with the flat-scene estimator, Trace::CastRay(ray, depth, sd::DataStorage&),
// CPURayTrace/Trace.cpp:58-96.
template <bool STATS>
RT_D float4 sample_sd(const KParams &P, RayState ray, uint32_t key, Cnt &cnt, uint32_t &primary) {
    float4 color = make_float4(0.f, 0.f, 0.f, 0.f);
    f3 thr = mk3(1.f, 1.f, 1.f);
    int depth = 0;
    while (depth < P.bounce_limit) {
        const uint32_t dim = 3u + 4u * (uint32_t)depth;
        depth++;
        Best hit;
        cnt.rays++;
        closest_hit<false, STATS>(P, ray.o, ray.d, ray.inv, hit, cnt);
        if (hit.ref != kNone) {
            if (STATS)
                cnt.trih++;
            const float4 q2 = P.tri_pos[3 * hit.ref + 2];
            if (depth == 1)
                primary = __float_as_uint(q2.y);
            if (__float_as_uint(q2.z) != 0u)
                continue; // not LambertianMat: the reference loop re-casts the unchanged ray (Trace.cpp:80-82)
            f3 N = tri_normal(P, hit.ref, hit.u, hit.v);
            f3 pos = ray.o + hit.t * ray.d;
            // Lambertian::Hit, Materials/Lambertian.hpp:35-48
            f3 bias = N * 1e-5f;
            float u1 = rng01(key, dim + 1u);
            float u2 = rng01(key, dim + 2u);
            f3 nd = cosine_dir(N, u1, u2);
            set_ray(ray, pos + bias, nd);
            thr = thr * mk3(0.9f, 0.6f, 0.5f);
            continue;
        }
        if (depth == 1)
            primary = kNone;
        f3 add = thr * sky_color(ray.d);
        color = make_float4(color.x + add.x, color.y + add.y, color.z + add.z, color.w + 1.0f);
        break;
    }
    return color;
}

// One pixel sample with the old-scene estimator, Trace::CastRay(ray, depth, const Scene&),
// CPURayTrace/Trace.cpp:31-56, unrolled into a loop: the recursion multiplies albedo and roulette
// weights on the way back, the loop multiplies them into a running throughput on the way in.
template <bool STATS>
RT_D float4 sample_old(const KParams &P, RayState ray, uint32_t key, Cnt &cnt, uint32_t &primary) {
    float4 thr = make_float4(1.f, 1.f, 1.f, 1.f);
    for (int depth = 0;; depth++) {
        const uint32_t dim = 3u + 4u * (uint32_t)depth;
        float rr = 1.0f; // Trace.cpp:33, Random::RussianRoulette(0.8f), Include/Random.hpp:74-78
        if (depth > 1)
            rr = rng01(key, dim) < 0.8f ? 1.0f / 0.8f : 0.0f;
        if (depth > P.bounce_limit || rr == 0.0f)
            return make_float4(0.f, 0.f, 0.f, 0.f);
        Best hit;
        cnt.rays++;
        closest_hit<true, STATS>(P, ray.o, ray.d, ray.inv, hit, cnt);
        if (hit.ref == kNone) {
            if (depth == 0)
                primary = kNone;
            f3 s = sky_color(ray.d);
            return make_float4((thr.x * s.x) * rr, (thr.y * s.y) * rr, (thr.z * s.z) * rr, (thr.w * 1.0f) * rr);
        }
        f3 pos = ray.o + hit.t * ray.d;
        f3 N;
        uint32_t mat, prim;
        if (hit.ref & kSphereBit) {
            const uint32_t si = hit.ref & ~kSphereBit;
            const float4 s0 = P.spheres[2 * si], s1 = P.spheres[2 * si + 1];
            N = normalize3(pos - mk3(s0.x, s0.y, s0.z));
            mat = __float_as_uint(s1.x);
            prim = __float_as_uint(s1.y);
        } else {
            if (STATS)
                cnt.trih++;
            const float4 q2 = P.tri_pos[3 * hit.ref + 2];
            N = tri_normal(P, hit.ref, hit.u, hit.v);
            prim = __float_as_uint(q2.y);
            mat = __float_as_uint(q2.z);
        }
        if (depth == 0)
            primary = prim;
        const float4 m0 = P.mats[2 * mat], m1 = P.mats[2 * mat + 1];
        const uint32_t kind = __float_as_uint(m1.x);
        const float param = m1.y;
        thr = make_float4((thr.x * m0.x) * rr, (thr.y * m0.y) * rr, (thr.z * m0.z) * rr, (thr.w * m0.w) * rr);
        if (kind == RT_MAT_EMIT) // LightEmit::getIrradiance, Materials/LightEmit.hpp:11-14
            return thr;
        if (kind == RT_MAT_DIELECTRIC) { // no reference counterpart: Snell + Schlick
            f3 unit = normalize3(ray.d);
            bool front = dot3(unit, N) < 0.0f;
            f3 n = front ? N : -N;
            float ri = front ? (1.0f / param) : param;
            float dn = dot3(-unit, n);
            float cosT = (1.0f < dn) ? 1.0f : dn;
            float sinT = sqrtf(1.0f - cosT * cosT);
            bool cannot = ri * sinT > 1.0f;
            float pick = rng01(key, dim + 3u);
            f3 bias = n * 1e-4f;
            if (cannot || reflectance(cosT, ri) > pick) {
                f3 refl = unit - n * dot3(unit, n) * 2.f;
                set_ray(ray, pos + bias, refl);
            } else {
                f3 perp = ri * (unit + cosT * n);
                f3 par = n * -sqrtf(fabsf(1.0f - dot3(perp, perp)));
                set_ray(ray, pos - bias, perp + par);
            }
            continue;
        }
        // Lambertian::getIrradiance (Materials/Lambertian.hpp:16-29) / Metal::getIrradiance (Metal.hpp:15-31)
        f3 bias = N * 1e-4f;
        float u1 = rng01(key, dim + 1u);
        float u2 = rng01(key, dim + 2u);
        f3 rnd = cosine_dir(N, u1, u2);
        if (kind == RT_MAT_METAL) {
            f3 refl = ray.d - N * dot3(ray.d, N) * 2.f;
            f3 nd = normalize3(rnd * param + refl * (1 - param));
            set_ray(ray, pos + bias, nd);
        } else {
            set_ray(ray, pos + bias, rnd);
        }
    }
}

// Thousands of atomics on ONE address are serialised by the memory system (about 10 ns each: 80 us for one per
// wave of a full grid), so counters and queue appends are spread over 64 addresses in separate cache lines.
RT_D unsigned long long *counter_slot(const KParams &P, uint32_t wave_id, int k) {
    return P.counters + (size_t)(wave_id & (uint32_t)(kCounterSlots - 1)) * kCounterStride + k;
}
RT_D unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// Persistent wave64 kernel.  Each wave draws 8x8-pixel tiles from one atomic counter; a lane owns one
// pixel of the tile and runs that pixel's samples in order, which keeps the reference's running mean
// "pixel = (pixel*(n-1) + new)/n" (Renderer/TraceMethods.cpp:59) in registers.  Finished lanes drop
// out of the wave's exec mask; the wave leaves a loop when the mask is empty.
template <bool OLD, bool STATS>
__global__ void __launch_bounds__(256) k_trace(const KParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    Cnt cnt = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    for (;;) {
        unsigned int j = 0;
        if (lane == 0)
            j = atomicAdd(P.queue, 1u);
        j = __builtin_amdgcn_readfirstlane(j);
        if (j >= P.n_my_tiles)
            break;
        const uint32_t tile = (uint32_t)P.tile_rank + j * (uint32_t)P.tile_world;
        const int x = (int)(tile % (uint32_t)P.tiles_x) * 8 + (int)(lane & 7u);
        const int y = (int)(tile / (uint32_t)P.tiles_x) * 8 + (int)(lane >> 3);
        if (x < P.W && y < P.H) {
            const uint32_t pixel = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
            float4 px = make_float4(0.f, 0.f, 0.f, 0.f);
            if (P.first_sample > 1)
                px = P.image[pixel];
            // CPUImageData::uvAt, Renderer/TraceMethods.hpp:30; Camera::getRayDirction, Camera.hpp:33-47
            const float u = (float)x / (float)P.W, v = (float)y / (float)P.H;
            f3 view = normalize3(mk3(u * P.cam_w - P.cam_w / 2, v * P.cam_h - P.cam_h / 2, P.focal));
            view.x = -view.x;
            const f3 dir0 = basis_mul(P.bx, P.by, P.bz, view);
            const uint32_t pkey = pcg(pcg(P.seed) + pixel);
            for (int n = P.first_sample; n < P.first_sample + P.n_samples; n++) {
                const uint32_t key = pcg(pkey + (uint32_t)n + P.sample_offset);
                // glm::linearRand(vec3(-s), vec3(s)), Include/Random.hpp:13-16
                const float s = P.jitter;
                f3 jit = mk3(rng01(key, 0u) * (s - (-s)) + (-s), rng01(key, 1u) * (s - (-s)) + (-s),
                             rng01(key, 2u) * (s - (-s)) + (-s));
                RayState ray;
                set_ray(ray, P.cam_pos, dir0 + jit);
                uint32_t prim = kNone;
                float4 c = OLD ? sample_old<STATS>(P, ray, key, cnt, prim) : sample_sd<STATS>(P, ray, key, cnt, prim);
                if (P.primary != nullptr && n == P.first_sample)
                    P.primary[pixel] = prim;
                const float nm1 = static_cast<float>((float)n - 1.f), nf = static_cast<float>(n);
                px = make_float4((px.x * nm1 + c.x) / nf, (px.y * nm1 + c.y) / nf, (px.z * nm1 + c.z) / nf,
                                 (px.w * nm1 + c.w) / nf);
            }
            P.image[pixel] = px;
        }
    }
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned long long r = wave_sum(cnt.rays);
    if (lane == 0 && r != 0ull)
        atomicAdd(counter_slot(P, wave_id, 0), r);
    if (STATS) {
        unsigned long long b = wave_sum(cnt.box), t = wave_sum(cnt.tri), s = wave_sum(cnt.sph), h = wave_sum(cnt.trih);
        if (lane == 0 && r != 0ull) {
            atomicAdd(counter_slot(P, wave_id, 1), b);
            atomicAdd(counter_slot(P, wave_id, 2), t);
            atomicAdd(counter_slot(P, wave_id, 3), s);
            atomicAdd(counter_slot(P, wave_id, 4), h);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Wave-synchronous closest hit.  All lanes of the wave walk the node array together: the wave visits
// index `cur` when at least one lane's own traversal would visit it.  A lane that fails the slab test
// of an internal node parks until `cur` reaches that node's skip index (`resume`); because subtrees
// are nested intervals of the array, no parked lane can want a node before the skip index of a node
// every active lane has just failed, so the wave's next index is simply cur+1 or skip.  Each lane
// therefore performs exactly the tests of its own reference traversal, in the same order, with its
// own running closest distance -- but the node (and triangle / sphere) record is fetched ONCE per
// wave through the scalar data path into SGPRs instead of 64 times through the vector memory path.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef const v4f __attribute__((address_space(4))) *cf4ptr; // constant address space: scalar loads
typedef const v8f __attribute__((address_space(4))) *cf8ptr;

// OCT >= 0: every live lane's direction has the same sign pattern (the usual case for the camera rays of
// a tile).  The near/far plane of each axis is then fixed at compile time instead of chosen by six
// per-lane selects on the products; "(pNear - o) * inv" is the very product the reference keeps after
// its swap, so the values are unchanged.
template <bool OLD, bool STATS, int OCT> // OCT = sign bits (x | y<<1 | z<<2) shared by the wave, or -1 = mixed
RT_D void wave_walk(const KParams &P, const bool live, const f3 o, const f3 d, const f3 inv, Best &best, Cnt &cnt,
                    const bool nx, const bool ny, const bool nz) {
    constexpr bool UNI = OCT >= 0;
    constexpr bool ux = UNI && (OCT & 1), uy = UNI && (OCT & 2), uz = UNI && (OCT & 4);
    typedef const char __attribute__((address_space(4))) *cbytes;
    const cbytes nodes = (cbytes)(uintptr_t)P.nodes;
    const cf4ptr tri_pos = (cf4ptr)(uintptr_t)P.tri_pos;
    const cf4ptr spheres = (cf4ptr)(uintptr_t)P.spheres;
    Best outer = best;
    uint32_t inner_end = kNone;
    uint32_t resume = live ? 0u : kNone; // byte offset of the next record this lane takes part in
    uint32_t cur = 0;                     // byte offset of the wave's current record
    // The scalar pipe issues one instruction per SIMD every four cycles, so the loop is written to need as
    // few scalar instructions per node as possible: one 32-byte scalar load, byte offsets instead of indices,
    // kind word 0 for internal nodes, and no bound check (the array ends with a terminator record).
    for (;;) {
        uint32_t skip, kind;
        float tm, tmin; // kept as floats across the loop exit; flags are re-derived (a carried flag costs selects)
        // inner loop: internal nodes only (kind word 0) -- the hot part of the walk
        for (;;) {
            const v8f r = *(cf8ptr)(nodes + cur);
            const v4f a = {r[0], r[1], r[2], r[3]}, b = {r[4], r[5], r[6], r[7]};
            skip = __float_as_uint(b.z);
            kind = __float_as_uint(b.w);
            float nearx, farx, neary, fary, nearz, farz;
            if (UNI) {
                const float pnx = ux ? a.w : a.x, pfx = ux ? a.x : a.w;
                const float pny = uy ? b.x : a.y, pfy = uy ? a.y : b.x;
                const float pnz = uz ? b.y : a.z, pfz = uz ? a.z : b.y;
                nearx = (pnx - o.x) * inv.x;
                farx = (pfx - o.x) * inv.x;
                neary = (pny - o.y) * inv.y;
                fary = (pfy - o.y) * inv.y;
                nearz = (pnz - o.z) * inv.z;
                farz = (pfz - o.z) * inv.z;
            } else {
                const float t0x = (a.x - o.x) * inv.x, t1x = (a.w - o.x) * inv.x;
                const float t0y = (a.y - o.y) * inv.y, t1y = (b.x - o.y) * inv.y;
                const float t0z = (a.z - o.z) * inv.z, t1z = (b.y - o.z) * inv.z;
                nearx = nx ? t1x : t0x;
                farx = nx ? t0x : t1x;
                neary = ny ? t1y : t0y;
                fary = ny ? t0y : t1y;
                nearz = nz ? t1z : t0z;
                farz = nz ? t0z : t1z;
            }
            const bool act = resume <= cur;
            if (OLD) {
                if (act && cur == inner_end) {
                    if (!(best.t < outer.t))
                        best = outer;
                    inner_end = kNone;
                }
            }
            if (STATS)
                cnt.box += (act && kind != kNone) ? 1ull : 0ull; // the terminator record is not a node
            // reject iff min(far.., closest) <= max(near.., 1e-6); a NaN plane product drops out of min/max
            // exactly as it drops out of the reference's "t0 > tMin ? t0 : tMin" selects
            tmin = fmaxf(nearz, fmaxf(neary, fmaxf(nearx, 1e-6f)));
            tm = fminf(fminf(farz, fminf(fary, farx)), best.t);
            if (kind != 0u)
                break;
            // two ballots of plain comparisons and a scalar AND: cheaper than one ballot of the combined flag
            const bool anyi = (__builtin_amdgcn_ballot_w64(resume <= cur) & __builtin_amdgcn_ballot_w64(!(tm <= tmin))) != 0ull;
            resume = ((resume <= cur) & (tm <= tmin)) ? skip : resume;
            cur = __builtin_amdgcn_readfirstlane(anyi ? cur + 32u : skip);
        }
        if (kind == kNone) // terminator record after the last node
            break;
        const bool act = resume <= cur;
        const bool ok = !(tm <= tmin);
        const bool pass = act & ok;
        const bool any = (__builtin_amdgcn_ballot_w64(resume <= cur) & __builtin_amdgcn_ballot_w64(!(tm <= tmin))) != 0ull;
        resume = (act & !ok) ? skip : resume;
        const uint32_t k = kind & 7u;
        const uint32_t slot = kind >> 3;
        uint32_t next = cur + 32u; // == skip for every leaf kind except a mesh leaf nobody entered
        if (!any) {
            if (OLD && k == KIND_MESH)
                next = skip;
        } else if (k == KIND_TRI) {
            const v4f q0 = tri_pos[3 * slot], q1 = tri_pos[3 * slot + 1], q2 = tri_pos[3 * slot + 2];
            if (pass) {
                if (STATS)
                    cnt.tri++;
                const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
                const float EPSILON = 1e-7f;
                f3 h = cross3(d, e2);
                float det = dot3(e1, h);
                if (!(det > -EPSILON && det < EPSILON)) {
                    float f = 1.0f / det;
                    f3 sv = o - p0;
                    float u = f * dot3(sv, h);
                    if (!(u < 0.0f || u > 1.0f)) {
                        f3 q = cross3(sv, e1);
                        float v = f * dot3(d, q);
                        if (!(v < 0.0f || u + v > 1.0f)) {
                            float t = f * dot3(e2, q);
                            if (t > EPSILON && t < best.t) {
                                best.t = t;
                                best.u = u;
                                best.v = v;
                                best.ref = slot;
                            }
                        }
                    }
                }
            }
        } else if (OLD && k == KIND_SPHERE) {
            const v4f s0 = spheres[2 * slot], s1 = spheres[2 * slot + 1];
            if (pass) {
                if (STATS)
                    cnt.sph++;
                f3 oc = mk3(s0.x, s0.y, s0.z) - o;
                float qa = dot3(d, d);
                float qb = -2.0f * dot3(d, oc);
                float qc = dot3(oc, oc) - s0.w * s0.w;
                float disc = qb * qb - 4 * qa * qc;
                if (!(disc < 0)) {
                    float t = (-qb - sqrtf(disc)) / (2.0f * qa);
                    bool ok = !(t < 0);
                    if (!ok && __float_as_uint(s1.z) != 0u) {
                        t = (-qb + sqrtf(disc)) / (2.0f * qa);
                        ok = !(t < 0);
                    }
                    if (ok && t < best.t) {
                        best.t = t;
                        best.ref = kSphereBit | slot;
                    }
                }
            }
        } else if (OLD && k == KIND_MESH) {
            if (pass) {
                outer = best;
                inner_end = skip;
                best.t = __builtin_inff();
                best.ref = kNone;
            }
        }
        cur = __builtin_amdgcn_readfirstlane(next);
    }
    if (OLD) {
        if (live && inner_end != kNone) {
            if (!(best.t < outer.t))
                best = outer;
        }
    }
}

template <bool OLD, bool STATS>
RT_D void closest_hit_wave(const KParams &P, const bool live, const f3 o, const f3 d, const f3 inv, Best &best,
                           Cnt &cnt) {
    best.t = __builtin_inff();
    best.u = 0.0f;
    best.v = 0.0f;
    best.ref = kNone;
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
    const unsigned long long mx = __builtin_amdgcn_ballot_w64(live && nx), my = __builtin_amdgcn_ballot_w64(live && ny),
                             mz = __builtin_amdgcn_ballot_w64(live && nz);
    const bool uniform = (mx == 0ull || mx == lm) && (my == 0ull || my == lm) && (mz == 0ull || mz == lm);
    const int oct = uniform ? ((mx != 0ull ? 1 : 0) | (my != 0ull ? 2 : 0) | (mz != 0ull ? 4 : 0)) : -1;
    switch (oct) {
    case 0: wave_walk<OLD, STATS, 0>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    case 1: wave_walk<OLD, STATS, 1>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    case 2: wave_walk<OLD, STATS, 2>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    case 3: wave_walk<OLD, STATS, 3>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    case 4: wave_walk<OLD, STATS, 4>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    case 5: wave_walk<OLD, STATS, 5>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    case 6: wave_walk<OLD, STATS, 6>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    case 7: wave_walk<OLD, STATS, 7>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    default: wave_walk<OLD, STATS, -1>(P, live, o, d, inv, best, cnt, nx, ny, nz); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Fast closest hit: the reference's RESULT without the reference's walk.
//
// In the reference traversal (SimplifiedData.cpp:430-462, BVH.hpp:124-148) the interior boxes only prune;
// every interior box contains the leaf boxes below it, and the slab test is monotone in the box, so a leaf
// whose own box passes "(1e-6, closest)" has ancestors that passed with the (larger or equal) closest of
// their moment.  What the reference returns is therefore a function of its LEAVES alone.  Call a leaf a
// candidate when the ray meets its box at all (slab test with closest = +inf) and its primitive test
// reports a hit at distance t; let m be the slab test's entry distance max(near planes, 1e-6).  The
// reference meets the candidates in its fixed visiting order `pos` and accepts one iff closest > m and
// t < closest.  If the candidate with the smallest (t, pos) has m <= t, it is accepted when its turn comes
// (everything accepted before it has a larger t) and nothing after it can replace it: it IS the result,
// whatever happened to the other candidates.  So the walk below visits the device's own tight hierarchy
// in any order, with no closest-distance pruning at all (a pruned leaf could hide exactly such a
// candidate), keeps the minimum of (t, pos), and is done unless the winner has m > t -- rounding in a
// grazing hit, or a sphere that touches its box.  Those rays are resolved by replay_ordered, which plays
// the reference's acceptance rule over the candidates in `pos` order and needs no more than the fast tree.
struct Cand {
    float t, u, v;
    float m;       // entry distance of the candidate's own leaf box
    uint32_t ref;  // triangle slot or kSphereBit | sphere index
    uint32_t pos;  // position of the leaf in the reference's visiting order
    uint32_t mesh; // OLD: position of the enclosing mesh leaf, kNone for a scene-level primitive
    float mesh_m;  // OLD: entry distance of that mesh leaf's box
};

// Position of one lane in the fast tree, for a walk that is taken in several bursts.
struct Walk {
    uint32_t idx;       // byte offset of the next record; the walk is over when it reaches the end of the array
    uint32_t mesh;      // OLD: reference position of the enclosing mesh leaf, kNone outside a mesh
    uint32_t inner_end; // OLD: byte offset where that mesh's inline hierarchy ends
    float mesh_m;       // OLD: entry distance of that mesh leaf's box
};

// BURST = 0: the whole walk.  BURST > 0: at most that many records, resuming from and saving to `w` (the loop itself
// runs on local copies either way).
template <bool OLD, bool STATS, int BURST, class F>
RT_D void fast_walk_from(const KParams &P, const f3 o, const f3 d, const f3 inv, Walk &w, Cnt &cnt, F &&consider) {
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const uint32_t n = P.n_fnodes * 32u;
    uint32_t idx = w.idx;
    uint32_t mesh = w.mesh, inner_end = w.inner_end;
    float mesh_m = w.mesh_m;
    int budget = BURST;
    while (idx < n) {
        if (BURST > 0) {
            if (budget-- == 0)
                break;
        }
        if (OLD) {
            if (idx == inner_end) {
                mesh = kNone;
                mesh_m = 0.0f;
                inner_end = kNone;
            }
        }
        const float4 *rec = (const float4 *)((const char *)P.fnodes + idx);
        const float4 a = rec[0];
        const float4 b = rec[1];
        const uint32_t skip = __float_as_uint(b.z);
        const uint32_t kind = __float_as_uint(b.w);
        if (STATS)
            cnt.box++;
        const float t0x = (a.x - o.x) * inv.x, t1x = (a.w - o.x) * inv.x;
        const float t0y = (a.y - o.y) * inv.y, t1y = (b.x - o.y) * inv.y;
        const float t0z = (a.z - o.z) * inv.z, t1z = (b.y - o.z) * inv.z;
        const float nearx = nx ? t1x : t0x, farx = nx ? t0x : t1x;
        const float neary = ny ? t1y : t0y, fary = ny ? t0y : t1y;
        const float nearz = nz ? t1z : t0z, farz = nz ? t0z : t1z;
        const float tmin = fmaxf(nearz, fmaxf(neary, fmaxf(nearx, 1e-6f)));
        const float tmax = fminf(farz, fminf(fary, fminf(farx, __builtin_inff()))); // the slab test with closest = +inf
        if (tmax <= tmin) {
            idx = skip;
            continue;
        }
        idx = idx + 32u;
        const uint32_t k = kind & 7u;
        if (k == KIND_INTERNAL)
            continue;
        const uint32_t slot = kind >> 3;
        if (k == KIND_TRI) {
            if (STATS)
                cnt.tri++;
            const float4 q0 = P.tri_pos[3 * slot], q1 = P.tri_pos[3 * slot + 1], q2 = P.tri_pos[3 * slot + 2];
            // All nine floats of the triangle are fetched before the test starts: left alone, the compiler sinks the
            // load of p0 behind the determinant test, which puts a second memory round trip into most triangle tests.
            asm volatile("" : : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "v"(q2.x));
            const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
            const float EPSILON = 1e-7f;
            f3 h = cross3(d, e2);
            float det = dot3(e1, h);
            if (!(det > -EPSILON && det < EPSILON)) {
                float f = 1.0f / det;
                f3 s = o - p0;
                float u = f * dot3(s, h);
                if (!(u < 0.0f || u > 1.0f)) {
                    f3 q = cross3(s, e1);
                    float v = f * dot3(d, q);
                    if (!(v < 0.0f || u + v > 1.0f)) {
                        float t = f * dot3(e2, q);
                        if (t > EPSILON)
                            consider(Cand{t, u, v, tmin, slot, __float_as_uint(q2.w), mesh, mesh_m});
                    }
                }
            }
        } else if (OLD && k == KIND_SPHERE) {
            if (STATS)
                cnt.sph++;
            const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
            f3 oc = mk3(s0.x, s0.y, s0.z) - o;
            float qa = dot3(d, d);
            float qb = -2.0f * dot3(d, oc);
            float qc = dot3(oc, oc) - s0.w * s0.w;
            float disc = qb * qb - 4 * qa * qc;
            if (!(disc < 0)) {
                float t = (-qb - sqrtf(disc)) / (2.0f * qa);
                bool ok = !(t < 0);
                if (!ok && __float_as_uint(s1.z) != 0u) {
                    t = (-qb + sqrtf(disc)) / (2.0f * qa);
                    ok = !(t < 0);
                }
                if (ok)
                    consider(Cand{t, 0.0f, 0.0f, tmin, kSphereBit | slot, __float_as_uint(s1.w), kNone, 0.0f});
            }
        } else if (OLD && k == KIND_MESH) { // the mesh's inner hierarchy follows inline, up to `skip`
            mesh = slot;
            mesh_m = tmin;
            inner_end = skip;
        }
    }
    if (BURST > 0) {
        w.idx = idx;
        w.mesh = mesh;
        w.inner_end = inner_end;
        w.mesh_m = mesh_m;
    }
}

template <bool OLD, bool STATS, class F>
RT_D void fast_walk(const KParams &P, const f3 o, const f3 d, const f3 inv, Cnt &cnt, F &&consider) {
    Walk w = {0u, kNone, kNone, 0.0f};
    fast_walk_from<OLD, STATS, 0>(P, o, d, inv, w, cnt, consider);
}

// Plays the reference's acceptance rule over the candidates in the reference's visiting order.  One pass over
// the fast tree finds the next candidate the reference would ACCEPT in its current state (between two
// acceptances the state does not change, so everything in between is skipped for good); the number of passes
// is the number of acceptances plus one.  A mesh (BVH.hpp:135-137 with Mesh::intersect, Mesh.hpp:154-160)
// runs its own closest hit from +inf once its scene-level leaf box passes, and hands the result to the scene
// level when the walk leaves it.
template <bool OLD, bool STATS>
RT_D void replay_ordered(const KParams &P, const f3 o, const f3 d, const f3 inv, Best &best, Cnt &cnt) {
    const float inf = __builtin_inff();
    float c_out = inf, c_in = inf;
    Best b_out = {inf, 0.0f, 0.0f, kNone}, b_in = b_out;
    uint32_t cur_mesh = kNone;
    uint32_t lo = 0;
    for (;;) {
        const float c_after = (OLD && cur_mesh != kNone && c_in < c_out) ? c_in : c_out; // scene-level closest once the current mesh is left
        Cand f;
        f.pos = kNone;
        f.t = inf;
        f.u = f.v = f.m = f.mesh_m = 0.0f;
        f.ref = f.mesh = kNone;
        fast_walk<OLD, STATS>(P, o, d, inv, cnt, [&](const Cand &c) {
            if (c.pos < lo || c.pos >= f.pos)
                return;
            bool ok;
            if (!OLD || c.mesh == kNone)
                ok = c_after > c.m && c.t < c_after;
            else if (c.mesh == cur_mesh)
                ok = c_in > c.m && c.t < c_in;
            else
                ok = c_after > c.mesh_m && c.t < inf; // entering that mesh: its closest starts at +inf
            if (ok)
                f = c;
        });
        if (f.pos == kNone)
            break;
        if (OLD && f.mesh != cur_mesh) {
            if (cur_mesh != kNone && c_in < c_out) {
                c_out = c_in;
                b_out = b_in;
            }
            cur_mesh = f.mesh;
            c_in = inf;
            b_in.ref = kNone;
        }
        if (!OLD || f.mesh == kNone) {
            c_out = f.t;
            b_out = Best{f.t, f.u, f.v, f.ref};
        } else {
            c_in = f.t;
            b_in = Best{f.t, f.u, f.v, f.ref};
        }
        lo = f.pos + 1u;
    }
    if (OLD && cur_mesh != kNone && c_in < c_out)
        b_out = b_in;
    best = b_out;
}

template <bool OLD, bool STATS>
RT_D void closest_hit_fast(const KParams &P, const f3 o, const f3 d, const f3 inv, Best &best, Cnt &cnt) {
    best.t = __builtin_inff();
    best.u = 0.0f;
    best.v = 0.0f;
    best.ref = kNone;
    uint32_t bpos = kNone;
    float bm = 0.0f;
    fast_walk<OLD, STATS>(P, o, d, inv, cnt, [&](const Cand &c) {
        // t = +inf passes the primitive test's "t > EPSILON" but never the reference's "t < closest.t"
        if ((c.t < best.t || (c.t == best.t && c.pos < bpos)) && c.t < __builtin_inff()) {
            best = Best{c.t, c.u, c.v, c.ref};
            bpos = c.pos;
            bm = fmaxf(c.m, c.mesh_m);
        }
    });
    if (best.ref == kNone) // no candidate: the reference accepts nothing either
        return;
    if (bm > best.t || P.force_replay != 0) {
        cnt.replay++;
        replay_ordered<OLD, STATS>(P, o, d, inv, best, cnt);
    }
}

// Path state carried from one bounce level to the next.
struct Path {
    f3 o, d, inv; // current ray
    float4 thr;   // throughput (w used by the old-scene estimator only)
    float rr;     // old-scene estimator: roulette weight of the current depth
    int depth;    // traversals already started on this path
    uint32_t key; // RNG key of (pixel, sample)
};

// Shades one finished closest-hit search: the path either ENDS (its colour goes to colors[sample][pixel]; returns
// true) or CONTINUES with a new ray in `path` (returns false).  Also records the primary-hit id of the first sample.
template <bool OLD, bool STATS>
RT_D bool shade_path(const KParams &P, Path &path, const Best &hit, const uint32_t cidx, Cnt &cnt) {
    bool ended = false;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t dim = 3u + 4u * (uint32_t)path.depth;
    const bool first = path.depth == 0;
    path.depth++;
    uint32_t prim = kNone;
    if (!OLD) { // Trace::CastRay(ray, depth, sd::DataStorage&), CPURayTrace/Trace.cpp:58-96
        if (hit.ref != kNone) {
            if (STATS)
                cnt.trih++;
            const float4 q2 = P.tri_pos[3 * hit.ref + 2];
            prim = __float_as_uint(q2.y);
            if (__float_as_uint(q2.z) == 0u) { // LambertianMat; otherwise the same ray is cast again (Trace.cpp:80-82)
                f3 N = tri_normal(P, hit.ref, hit.u, hit.v);
                f3 pos = path.o + hit.t * path.d;
                f3 bias = N * 1e-5f; // Lambertian::Hit, Materials/Lambertian.hpp:35-48
                float u1 = rng01(path.key, dim + 1u);
                float u2 = rng01(path.key, dim + 2u);
                path.d = cosine_dir(N, u1, u2);
                path.o = pos + bias;
                path.thr = make_float4(path.thr.x * 0.9f, path.thr.y * 0.6f, path.thr.z * 0.5f, 1.f);
            }
            ended = !(path.depth < P.bounce_limit); // loop condition of Trace.cpp:63: colour stays (0,0,0,0)
        } else {
            f3 sk = sky_color(path.d);
            c = make_float4(0.f + path.thr.x * sk.x, 0.f + path.thr.y * sk.y, 0.f + path.thr.z * sk.z, 0.f + 1.0f);
            ended = true;
        }
    } else { // Trace::CastRay(ray, depth, const Scene&), CPURayTrace/Trace.cpp:31-56, products taken front to back
        const float rr = path.rr;
        if (hit.ref == kNone) {
            f3 sk = sky_color(path.d);
            c = make_float4((path.thr.x * sk.x) * rr, (path.thr.y * sk.y) * rr, (path.thr.z * sk.z) * rr,
                            (path.thr.w * 1.0f) * rr);
            ended = true;
        } else {
            f3 pos = path.o + hit.t * path.d;
            f3 N;
            uint32_t mat;
            if (hit.ref & kSphereBit) {
                const uint32_t si = hit.ref & ~kSphereBit;
                const float4 s0 = P.spheres[2 * si], s1 = P.spheres[2 * si + 1];
                N = normalize3(pos - mk3(s0.x, s0.y, s0.z));
                mat = __float_as_uint(s1.x);
                prim = __float_as_uint(s1.y);
            } else {
                if (STATS)
                    cnt.trih++;
                const float4 q2 = P.tri_pos[3 * hit.ref + 2];
                N = tri_normal(P, hit.ref, hit.u, hit.v);
                prim = __float_as_uint(q2.y);
                mat = __float_as_uint(q2.z);
            }
            const float4 m0 = P.mats[2 * mat], m1 = P.mats[2 * mat + 1];
            const uint32_t mkind = __float_as_uint(m1.x);
            const float param = m1.y;
            path.thr = make_float4((path.thr.x * m0.x) * rr, (path.thr.y * m0.y) * rr, (path.thr.z * m0.z) * rr,
                                   (path.thr.w * m0.w) * rr);
            if (mkind == RT_MAT_EMIT) { // LightEmit::getIrradiance, Materials/LightEmit.hpp:11-14
                c = path.thr;
                ended = true;
            } else {
                f3 no, nd;
                if (mkind == RT_MAT_DIELECTRIC) { // no reference counterpart: Snell + Schlick
                    f3 unit = normalize3(path.d);
                    bool front = dot3(unit, N) < 0.0f;
                    f3 nn = front ? N : -N;
                    float ri = front ? (1.0f / param) : param;
                    float dn = dot3(-unit, nn);
                    float cosT = (1.0f < dn) ? 1.0f : dn;
                    float sinT = sqrtf(1.0f - cosT * cosT);
                    bool cannot = ri * sinT > 1.0f;
                    float pick = rng01(path.key, dim + 3u);
                    f3 bias = nn * 1e-4f;
                    if (cannot || reflectance(cosT, ri) > pick) {
                        nd = unit - nn * dot3(unit, nn) * 2.f;
                        no = pos + bias;
                    } else {
                        f3 perp = ri * (unit + cosT * nn);
                        f3 par = nn * -sqrtf(fabsf(1.0f - dot3(perp, perp)));
                        nd = perp + par;
                        no = pos - bias;
                    }
                } else { // Lambertian (Materials/Lambertian.hpp:16-29) or Metal (Materials/Metal.hpp:15-31)
                    f3 bias = N * 1e-4f;
                    float u1 = rng01(path.key, dim + 1u);
                    float u2 = rng01(path.key, dim + 2u);
                    f3 rnd = cosine_dir(N, u1, u2);
                    if (mkind == RT_MAT_METAL) {
                        f3 refl = path.d - N * dot3(path.d, N) * 2.f;
                        nd = normalize3(rnd * param + refl * (1 - param));
                    } else {
                        nd = rnd;
                    }
                    no = pos + bias;
                }
                path.o = no;
                path.d = nd;
                // entry test of the next CastRay call (Trace.cpp:33-38): roulette, then the depth cap
                float nrr = 1.0f;
                if (path.depth > 1)
                    nrr = rng01(path.key, 3u + 4u * (uint32_t)path.depth) < 0.8f ? 1.0f / 0.8f : 0.0f;
                if (path.depth > P.bounce_limit || nrr == 0.0f)
                    ended = true; // that call returns zero
                path.rr = nrr;
            }
        }
    }
    if (first && P.primary != nullptr && P.chunk_first + (int)(cidx / P.npix) == P.first_sample)
        P.primary[cidx % P.npix] = prim;
    if (ended)
        P.colors[cidx] = c;
    return ended;
}

// Camera ray of work item `item` = (tile ordinal, sample in chunk, pixel in the 8x8 tile); returns false for a pixel
// outside the image (or when the estimator starts no traversal at all).
template <bool OLD>
RT_D bool camera_path(const KParams &P, const uint32_t item, Path &path, uint32_t &cidx) {
    const uint32_t per_tile = 64u * (uint32_t)P.S;
    const uint32_t j = fastdiv(item, P.div_per_tile);
    const uint32_t s = (item - j * per_tile) >> 6;
    const uint32_t in_tile = item & 63u;
    const uint32_t tile = (uint32_t)P.tile_rank + j * (uint32_t)P.tile_world;
    const uint32_t tile_y = fastdiv(tile, P.div_tiles_x);
    const int x = (int)(tile - tile_y * (uint32_t)P.tiles_x) * 8 + (int)(in_tile & 7u);
    const int y = (int)tile_y * 8 + (int)(in_tile >> 3);
    bool live = item < P.n_items && x < P.W && y < P.H;
    const uint32_t pixel = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
    const int n = P.chunk_first + (int)s;
    cidx = s * P.npix + pixel;
    // CPUImageData::uvAt, Renderer/TraceMethods.hpp:30; Camera::getRayDirction, Camera.hpp:33-47
    const float u = (float)x / (float)P.W, v = (float)y / (float)P.H;
    f3 view = normalize3(mk3(u * P.cam_w - P.cam_w / 2, v * P.cam_h - P.cam_h / 2, P.focal));
    view.x = -view.x;
    const f3 dir0 = basis_mul(P.bx, P.by, P.bz, view);
    path.key = pcg(pcg(P.seed_hash + pixel) + (uint32_t)n + P.sample_offset);
    // glm::linearRand(vec3(-s), vec3(s)), Include/Random.hpp:13-16
    const float js = P.jitter;
    f3 jit = mk3(rng01(path.key, 0u) * (js - (-js)) + (-js), rng01(path.key, 1u) * (js - (-js)) + (-js),
                 rng01(path.key, 2u) * (js - (-js)) + (-js));
    path.o = P.cam_pos;
    path.d = dir0 + jit;
    path.thr = make_float4(1.f, 1.f, 1.f, 1.f);
    path.rr = 1.0f;
    path.depth = 0;
    if (!OLD && P.bounce_limit <= 0) { // Trace.cpp:63: no traversal at all, the sample is (0,0,0,0)
        if (live) {
            P.colors[cidx] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (P.primary != nullptr && n == P.first_sample)
                P.primary[pixel] = kNone;
        }
        live = false;
    }
    return live;
}

// Wavefront path kernel: one launch per bounce level.
//
// Level 0 generates the camera rays of `S` consecutive samples of every pixel this context owns.  A
// wave takes 64 work items at a time; items are numbered (tile, sample, pixel in
// the 8x8 tile), so the 64 rays of a batch are the camera rays of one tile for one sample and stay
// coherent.  After the closest-hit search a path either ENDS -- its colour is stored in the per-sample
// colour buffer, colors[sample][pixel] -- or CONTINUES: its state is appended to the queue of the next
// level, which the next launch consumes 64 records per wave.  Paths of every depth therefore run in
// full waves, and no lane ever waits for a neighbour's longer path.  The per-sample colours are folded
// into the running mean afterwards, in sample order, by k_fold, which reproduces the reference's
// "pixel = (pixel*(n-1) + new)/n" (Renderer/TraceMethods.cpp:59) bit for bit.
constexpr int WALK_LANE = 0, WALK_WAVE = 1, WALK_FAST = 2, WALK_BRUTE = 3;
template <bool OLD, bool STATS, int WALK>
__global__ void __launch_bounds__(256, WALK == WALK_WAVE ? 8 : 1) k_path(const KParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    Cnt cnt = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    // Level 0: items are camera rays, batch b = items 64 b .. 64 b + 63.  Deeper levels: the queue is kSubQueues
    // sub-queues, each filled by the level before through its own append counter; lane q of every wave reads count
    // q, and a wave scan gives the level's total number of batches (for the tail decision and the early return).
    uint32_t sub_count = 0u, sub_end = 0u; // lane q: records in sub-queue q; batches in sub-queues 0..q
    uint32_t n_batches;
    if (P.level == 0) {
        n_batches = (P.n_items + 63u) >> 6;
    } else {
        sub_count = P.sub_in[lane * 32u];
        sub_end = (sub_count + 63u) >> 6;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)sub_end, off, 64);
            if ((int)lane >= off)
                sub_end += up;
        }
        n_batches = (uint32_t)__builtin_amdgcn_readlane((int)sub_end, 63);
    }
    const uint32_t n_in = P.level == 0 ? P.n_items : n_batches * 64u;
    // A level with too few rays to fill the chip is bound by the latency of one walk, which is shortest
    // when every lane walks alone; a level that saturates the chip is bound by throughput.
    const bool use_wave = WALK == WALK_WAVE && n_in >= P.wave_min_rays;
    // Tail mode: when a level holds too few rays to fill the chip, its launch lasts as long as its single longest
    // walk, and so would every launch after it; instead each lane then follows its path to the end, and the launches
    // of the deeper levels find their queues empty and return at once.
    const bool tail = P.tail != 0 || (P.level >= 1 && n_in <= kTailRays);
    // Batches of 64 items are dealt to the waves round-robin: one shared cursor would be hit by half a million
    // same-address atomics per launch at level 0, which alone takes longer than the rays.  Neighbouring
    // batches (the samples of one tile, tiles of one image row) go to different waves, so the expensive part of
    // the image is spread over the whole grid.
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Work distribution: the waves with wave_id % 64 == q form group q and draw batches from that group's own cursor
    // (64 cursors in separate cache lines: atomics on one address are serialised at about 10 ns each), so a wave
    // that met long walks simply draws fewer batches.  Level 0: group q owns the batches q, q + 64, q + 128, ... of
    // camera rays and draws them kChunk at a time.  Deeper levels: group q owns sub-queue q.
    constexpr uint32_t kChunk = 4u;
    const uint32_t my_q = wave_id & (uint32_t)(kSubQueues - 1);
    const uint32_t my_count = P.level == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)sub_count, (int)my_q);
    const uint32_t my_batches = P.level == 0 ? (n_batches > my_q ? (n_batches - my_q + (uint32_t)kSubQueues - 1u) / (uint32_t)kSubQueues : 0u)
                                             : (my_count + 63u) >> 6;
    const uint32_t chunk = P.level == 0 ? kChunk : 1u;
    uint32_t next = 0u, left = 0u; // next batch of the group this wave may take, and how many of its chunk remain
    for (;;) {
        if (left == 0u) {
            unsigned int b = 0;
            if (lane == 0)
                b = atomicAdd(P.sub_cursor + my_q * 32u, chunk);
            next = __builtin_amdgcn_readfirstlane(b);
            left = chunk;
        }
        const uint32_t b = next;
        if (b >= my_batches)
            break;
        next++;
        left--;
        uint32_t g, out_q; // g: item number (level 0) or record number in the queue; out_q: sub-queue continuing paths go to
        bool live;
        if (P.level == 0) {
            const uint32_t batch = my_q + b * (uint32_t)kSubQueues;
            g = batch * 64u + lane;
            live = g < n_in;
            out_q = my_q;
        } else {
            const uint32_t local = b * 64u + lane;
            live = local < my_count;
            g = my_q * P.sub_cap + local;
            out_q = (my_q + b) & (uint32_t)(kSubQueues - 1);
        }
        Path path;
        uint32_t cidx = 0;
        if (P.level == 0) {
            live = camera_path<OLD>(P, g, path, cidx);
        } else {
            path.o = path.d = mk3(1.f, 1.f, 1.f);
            path.thr = make_float4(1.f, 1.f, 1.f, 1.f);
            path.rr = 1.0f;
            path.depth = 0;
            path.key = 0;
            if (live) {
                const float4 r0 = P.q_in[4 * (size_t)g], r1 = P.q_in[4 * (size_t)g + 1], r2 = P.q_in[4 * (size_t)g + 2],
                             r3 = P.q_in[4 * (size_t)g + 3];
                path.o = mk3(r0.x, r0.y, r0.z);
                path.d = mk3(r0.w, r1.x, r1.y);
                path.thr = make_float4(r1.z, r1.w, r2.x, r2.y);
                path.rr = r2.z;
                path.key = __float_as_uint(r2.w);
                cidx = __float_as_uint(r3.x);
                path.depth = (int)__float_as_uint(r3.y);
            }
        }
        bool ended = false;
        // One pass = one closest-hit search and its shading.  In tail mode (the deep levels, whose few rays cannot
        // fill the chip and whose launches each last as long as their single longest walk) a lane keeps its path
        // until it ends, instead of handing it to another launch.
        for (;;) {
        if (__ballot(live) == 0ull)
            break;
        path.inv = mk3(1.0f / path.d.x, 1.0f / path.d.y, 1.0f / path.d.z); // Ray::Ray, CPURayTrace/Ray.hpp:8
        // ---- one closest-hit search per lane ----
        Best hit;
        if (live)
            cnt.rays++;
        if (WALK == WALK_WAVE && use_wave) {
            closest_hit_wave<OLD, STATS>(P, live, path.o, path.d, path.inv, hit, cnt);
        } else {
            hit.ref = kNone;
            hit.t = 0.f;
            hit.u = hit.v = 0.f;
            if (live) {
                if (WALK == WALK_FAST)
                    closest_hit_fast<OLD, STATS>(P, path.o, path.d, path.inv, hit, cnt);
                else if (WALK == WALK_BRUTE)
                    closest_hit_brute<OLD, STATS>(P, path.o, path.d, hit, cnt);
                else
                    closest_hit<OLD, STATS>(P, path.o, path.d, path.inv, hit, cnt);
            }
        }
        // ---- shade: the path ends with a colour or continues with a new ray ----
        ended = false;
        if (live)
            ended = shade_path<OLD, STATS>(P, path, hit, cidx, cnt);
        if (!tail)
            break;
        live = live && !ended;
        }
        if (tail)
            continue;
        // ---- continuing paths go to the next level's queue, compacted per wave ----
        const bool cont = live && !ended;
        const unsigned long long cmask = __ballot(cont);
        if (cmask != 0ull) {
            const uint32_t q = out_q;
            unsigned int obase = 0;
            if (lane == 0)
                obase = atomicAdd(P.sub_out + q * 32u, (unsigned int)__builtin_popcountll(cmask));
            obase = __builtin_amdgcn_readfirstlane(obase);
            if (cont) {
                const size_t slot = (size_t)q * P.sub_cap + (size_t)obase + (size_t)__builtin_popcountll(cmask & ((1ull << lane) - 1ull));
                P.q_out[4 * slot] = make_float4(path.o.x, path.o.y, path.o.z, path.d.x);
                P.q_out[4 * slot + 1] = make_float4(path.d.y, path.d.z, path.thr.x, path.thr.y);
                P.q_out[4 * slot + 2] = make_float4(path.thr.z, path.thr.w, path.rr, __uint_as_float(path.key));
                P.q_out[4 * slot + 3] = make_float4(__uint_as_float(cidx), __uint_as_float((uint32_t)path.depth), 0.f, 0.f);
            }
        }
    }
    // Counters: one atomic per wave that has something to add.  (Thousands of same-address atomics are
    // serialised by the memory system; a launch whose level is empty must not pay for them.)
    if (n_batches == 0u)
        return;
    unsigned long long r = wave_sum(cnt.rays);
    if (lane == 0 && r != 0ull)
        atomicAdd(counter_slot(P, wave_id, 0), r);
    if (STATS) {
        unsigned long long b = wave_sum(cnt.box), t = wave_sum(cnt.tri), s = wave_sum(cnt.sph), h = wave_sum(cnt.trih);
        if (lane == 0 && r != 0ull) {
            atomicAdd(counter_slot(P, wave_id, 1), b);
            atomicAdd(counter_slot(P, wave_id, 2), t);
            atomicAdd(counter_slot(P, wave_id, 3), s);
            atomicAdd(counter_slot(P, wave_id, 4), h);
        }
    }
    if (WALK == WALK_FAST) {
        unsigned long long rp = wave_sum(cnt.replay);
        if (lane == 0 && rp != 0ull)
            atomicAdd(counter_slot(P, wave_id, 5), rp);
    }
}

// Path kernel of the fast walk with lane re-fill (one launch per bounce level, like k_path).
//
// Walk lengths are heavy-tailed, so a wave that takes 64 rays and walks until the longest one is done has, at the
// bounce levels, about 11 of its 64 lanes busy per instruction.  Here a lane that finishes its walk does not wait for
// the wave: the wave walks in bursts of kBurst records; when kServiceLanes lanes have finished (or none is left
// walking) it shades the finished ones together -- a path ENDS (colour stored) or CONTINUES (appended to the next
// level's queue, or, in tail mode, walked on by the same lane) -- and hands the free lanes the next items of the
// batch the wave holds, drawing a new batch from its group's cursor when that one is used up.  Same per-ray
// arithmetic as k_path, same results.
constexpr int kBurst = 16;
constexpr uint32_t kServiceLanes = 24;
constexpr size_t kRefillSceneBytes = 32u << 20; // 8 XCDs x 4 MB of L2

template <bool OLD, bool STATS>
__global__ void __launch_bounds__(256) k_refill(const KParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    Cnt cnt = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    uint32_t sub_count = 0u, sub_end = 0u;
    uint32_t n_batches;
    if (P.level == 0) {
        n_batches = (P.n_items + 63u) >> 6;
    } else {
        sub_count = P.sub_in[lane * 32u];
        sub_end = (sub_count + 63u) >> 6;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)sub_end, off, 64);
            if ((int)lane >= off)
                sub_end += up;
        }
        n_batches = (uint32_t)__builtin_amdgcn_readlane((int)sub_end, 63);
    }
    if (n_batches == 0u)
        return;
    const bool tail = P.tail != 0 || (P.level >= 1 && n_batches * 64u <= kTailRays);
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t my_q = wave_id & (uint32_t)(kSubQueues - 1);
    const uint32_t my_count = P.level == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)sub_count, (int)my_q);
    const uint32_t my_batches = P.level == 0 ? (n_batches > my_q ? (n_batches - my_q + (uint32_t)kSubQueues - 1u) / (uint32_t)kSubQueues : 0u)
                                             : (my_count + 63u) >> 6;
    const uint32_t n = P.n_fnodes * 32u;
    // the batch the wave is handing out, and how many of its 64 items are gone (64 = none held)
    uint32_t cur_b = 0u, used = 64u;
    bool drained = my_batches == 0u; // the group's cursor has run past its last batch
    // lane state
    bool has = false;
    Path path;
    path.o = path.d = path.inv = mk3(1.f, 1.f, 1.f);
    path.thr = make_float4(1.f, 1.f, 1.f, 1.f);
    path.rr = 1.0f;
    path.depth = 0;
    path.key = 0u;
    uint32_t cidx = 0u, out_q = 0u;
    Walk w = {n, kNone, kNone, 0.0f};
    Best best = {__builtin_inff(), 0.0f, 0.0f, kNone};
    uint32_t bpos = kNone;
    float bm = 0.0f;
    auto start_walk = [&]() {
        path.inv = mk3(1.0f / path.d.x, 1.0f / path.d.y, 1.0f / path.d.z); // Ray::Ray, CPURayTrace/Ray.hpp:8
        w = Walk{0u, kNone, kNone, 0.0f};
        best = Best{__builtin_inff(), 0.0f, 0.0f, kNone};
        bpos = kNone;
        bm = 0.0f;
        cnt.rays++;
    };
    for (;;) {
        // ---- free lanes take the next items of the held batch; a used-up batch is replaced from the group's cursor ----
        for (;;) {
            const unsigned long long empty = __ballot(!has);
            if (empty == 0ull)
                break;
            if (used >= 64u) {
                if (drained)
                    break;
                unsigned int b = 0;
                if (lane == 0)
                    b = atomicAdd(P.sub_cursor + my_q * 32u, 1u);
                cur_b = __builtin_amdgcn_readfirstlane(b);
                if (cur_b >= my_batches) {
                    drained = true;
                    break;
                }
                used = 0u;
            }
            const uint32_t item = used + (uint32_t)__builtin_popcountll(empty & below); // item of the held batch for this lane
            if (!has && item < 64u) {
                if (P.level == 0) {
                    has = camera_path<OLD>(P, (my_q + cur_b * (uint32_t)kSubQueues) * 64u + item, path, cidx);
                    out_q = my_q;
                } else if (cur_b * 64u + item < my_count) {
                    const size_t g = (size_t)my_q * P.sub_cap + (size_t)cur_b * 64u + item;
                    const float4 r0 = P.q_in[4 * g], r1 = P.q_in[4 * g + 1], r2 = P.q_in[4 * g + 2], r3 = P.q_in[4 * g + 3];
                    path.o = mk3(r0.x, r0.y, r0.z);
                    path.d = mk3(r0.w, r1.x, r1.y);
                    path.thr = make_float4(r1.z, r1.w, r2.x, r2.y);
                    path.rr = r2.z;
                    path.key = __float_as_uint(r2.w);
                    cidx = __float_as_uint(r3.x);
                    path.depth = (int)__float_as_uint(r3.y);
                    out_q = (my_q + cur_b) & (uint32_t)(kSubQueues - 1);
                    has = true;
                }
                if (has)
                    start_walk();
            }
            const uint32_t taken = (uint32_t)__builtin_popcountll(empty);
            used = used + taken < 64u ? used + taken : 64u;
            if (used < 64u)
                break; // every free lane was served
        }
        if (__ballot(has) == 0ull)
            break; // nothing in flight and nothing left to hand out
        // ---- walk in bursts until enough lanes wait for service ----
        for (;;) {
            if (has && w.idx < n) {
                fast_walk_from<OLD, STATS, kBurst>(P, path.o, path.d, path.inv, w, cnt, [&](const Cand &c) {
                    if ((c.t < best.t || (c.t == best.t && c.pos < bpos)) && c.t < __builtin_inff()) {
                        best = Best{c.t, c.u, c.v, c.ref};
                        bpos = c.pos;
                        bm = fmaxf(c.m, c.mesh_m);
                    }
                });
            }
            const unsigned long long walking = __ballot(has && w.idx < n);
            if (walking == 0ull)
                break;
            const uint32_t finished = (uint32_t)__builtin_popcountll(__ballot(has && !(w.idx < n)));
            const uint32_t fillable = drained && used >= 64u ? 0u : (uint32_t)__builtin_popcountll(__ballot(!has));
            if (finished + fillable >= kServiceLanes)
                break;
        }
        // ---- shade the finished walks ----
        bool cont = false;
        if (has && !(w.idx < n)) {
            if (best.ref != kNone && (bm > best.t || P.force_replay != 0)) { // see closest_hit_fast
                cnt.replay++;
                replay_ordered<OLD, STATS>(P, path.o, path.d, path.inv, best, cnt);
            }
            const bool ended = shade_path<OLD, STATS>(P, path, best, cidx, cnt);
            if (ended) {
                has = false;
            } else if (tail) {
                start_walk();
            } else {
                cont = true;
                has = false;
            }
        }
        // continuing paths go to the next level's queue: the sub-queue their batch feeds, compacted per wave
        unsigned long long cmask = __ballot(cont);
        while (cmask != 0ull) {
            const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)out_q, __builtin_ctzll(cmask));
            const unsigned long long qmask = __ballot(cont && out_q == q);
            unsigned int obase = 0;
            if (lane == 0)
                obase = atomicAdd(P.sub_out + q * 32u, (unsigned int)__builtin_popcountll(qmask));
            obase = __builtin_amdgcn_readfirstlane(obase);
            if (cont && out_q == q) {
                const size_t slot = (size_t)q * P.sub_cap + (size_t)obase + (size_t)__builtin_popcountll(qmask & below);
                P.q_out[4 * slot] = make_float4(path.o.x, path.o.y, path.o.z, path.d.x);
                P.q_out[4 * slot + 1] = make_float4(path.d.y, path.d.z, path.thr.x, path.thr.y);
                P.q_out[4 * slot + 2] = make_float4(path.thr.z, path.thr.w, path.rr, __uint_as_float(path.key));
                P.q_out[4 * slot + 3] = make_float4(__uint_as_float(cidx), __uint_as_float((uint32_t)path.depth), 0.f, 0.f);
            }
            cmask &= ~qmask;
        }
    }
    unsigned long long r = wave_sum(cnt.rays);
    if (lane == 0 && r != 0ull)
        atomicAdd(counter_slot(P, wave_id, 0), r);
    if (STATS) {
        unsigned long long b = wave_sum(cnt.box), t = wave_sum(cnt.tri), sp = wave_sum(cnt.sph), h = wave_sum(cnt.trih);
        if (lane == 0 && r != 0ull) {
            atomicAdd(counter_slot(P, wave_id, 1), b);
            atomicAdd(counter_slot(P, wave_id, 2), t);
            atomicAdd(counter_slot(P, wave_id, 3), sp);
            atomicAdd(counter_slot(P, wave_id, 4), h);
        }
    }
    unsigned long long rp = wave_sum(cnt.replay);
    if (lane == 0 && rp != 0ull)
        atomicAdd(counter_slot(P, wave_id, 5), rp);
}

// Folds the S per-sample colours of every owned pixel into the running mean, in sample order:
// pixel = (pixel*(n-1) + colour_n)/n, Renderer/TraceMethods.cpp:59.
__global__ void __launch_bounds__(256) k_fold(float4 *image, const float4 *colors, uint32_t npix, int W, int tiles_x,
                                              int tile_rank, int tile_world, int chunk_first, int S) {
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    if (pixel >= npix)
        return;
    if (tile_world > 1) {
        const uint32_t x = pixel % (uint32_t)W, y = pixel / (uint32_t)W;
        const uint32_t tile = (y >> 3) * (uint32_t)tiles_x + (x >> 3);
        if ((int)(tile % (uint32_t)tile_world) != tile_rank)
            return;
    }
    float4 px = make_float4(0.f, 0.f, 0.f, 0.f);
    if (chunk_first > 1)
        px = image[pixel];
    for (int s = 0; s < S; s++) {
        const int n = chunk_first + s;
        const float4 c = colors[(size_t)s * npix + pixel];
        const float nm1 = static_cast<float>((float)n - 1.f), nf = static_cast<float>(n);
        px = make_float4((px.x * nm1 + c.x) / nf, (px.y * nm1 + c.y) / nf, (px.z * nm1 + c.z) / nf, (px.w * nm1 + c.w) / nf);
    }
    image[pixel] = px;
}

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
    if ((trav == RT_TRAV_FAST || trav == RT_TRAV_FAST_REPLAY || trav == RT_TRAV_FAST_REFILL) && !ctx->fast_exact)
        return fail(ctx, RT_ERR_BAD_SCENE, "RT_TRAV_FAST needs a tree whose interior boxes contain their children's boxes");
    if (trav == RT_TRAV_REFERENCE)
        trav = ctx->expected_visits >= 512.0 ? RT_TRAV_WAVE : RT_TRAV_LANE;
    P.force_replay = trav == RT_TRAV_FAST_REPLAY ? 1 : 0;
    // Lane re-fill pays when walks are long and their fetches miss the L2 (measured: +17 % on the 1.19 M-triangle field,
    // +29 % on the 9.8 M one; -12 % on the 485-sphere scene, neutral on the 18 520-triangle dragon): it is used for
    // scenes whose fast tree and triangle positions exceed the L2s, and when asked for explicitly.
    const bool refill = trav == RT_TRAV_FAST_REFILL || (trav == RT_TRAV_FAST && ctx->scene_bytes > kRefillSceneBytes);
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
    const int levels = old ? o.bounce_limit + 1 : (o.bounce_limit > 1 ? o.bounce_limit : 1);
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
        if (need_colors > pp.colors_cap || need_q > pp.q_cap) {
            RT_HIP_CHECK(ctx, hipStreamSynchronize(pp.stream));
            if (pp.d_colors) (void)hipFree(pp.d_colors);
            if (pp.d_q[0]) (void)hipFree(pp.d_q[0]);
            if (pp.d_q[1]) (void)hipFree(pp.d_q[1]);
            pp.d_colors = pp.d_q[0] = pp.d_q[1] = nullptr;
            pp.colors_cap = pp.q_cap = 0;
            RT_HIP_CHECK(ctx, hipMalloc((void **)&pp.d_colors, need_colors * sizeof(float4)));
            RT_HIP_CHECK(ctx, hipMalloc((void **)&pp.d_q[0], (need_q ? need_q : 1) * 4 * sizeof(float4)));
            RT_HIP_CHECK(ctx, hipMalloc((void **)&pp.d_q[1], (need_q ? need_q : 1) * 4 * sizeof(float4)));
            pp.colors_cap = need_colors;
            pp.q_cap = need_q;
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
        if (walk == WALK_WAVE) hipLaunchKernelGGL((k_path<OLD_, STATS_, WALK_WAVE>), grid, block, 0, pp.stream, P);     \
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

// What rt_upload_scene puts into HBM, as host arrays (built without touching a device).
struct SceneLayout {
    std::vector<float4> dn, fn, tp, tn, sp, mt; // reference-order nodes, fast-tree nodes, triangle positions / normals, spheres, materials
    uint32_t n_ref = 0, n_fast = 0, n_tri_slots = 0, n_sph_slots = 0;
    double expected_visits = 0.0;
    bool nested = true; // every interior box of the caller's tree contains its children's boxes
};

int lfail(std::string &err, int code, const std::string &msg) {
    err = msg;
    return code;
}

int build_layout(const rt_node *nodes, uint32_t n_nodes, uint32_t root, const rt_tri *tris, uint32_t n_tris,
                 const rt_sphere *spheres, uint32_t n_spheres, const rt_material *mats, uint32_t n_mats, int order,
                 SceneLayout &L, std::string &err) {
    if (!nodes || n_nodes == 0 || root >= n_nodes)
        return lfail(err, RT_ERR_BAD_SCENE, "Build Failed. end - start <= 0 (empty node array or bad root)");
    if ((n_tris && !tris) || (n_spheres && !spheres) || (n_mats && !mats))
        return lfail(err, RT_ERR_INVALID_ARGUMENT, "null primitive array");
    if (order != RT_ORDER_RIGHT_FIRST && order != RT_ORDER_LEFT_FIRST)
        return lfail(err, RT_ERR_INVALID_ARGUMENT, "bad traversal order");
    if (n_tris >= (1u << 28) || n_spheres >= (1u << 28) || n_nodes >= (1u << 27))
        return lfail(err, RT_ERR_BAD_SCENE, "too many nodes or primitives");

    // 1. Walk the tree once in the reference's visiting order and record that order.
    std::vector<uint32_t> order_of; // emission position -> node index
    std::vector<uint32_t> pos_of(n_nodes, kNone);
    order_of.reserve(n_nodes);
    std::vector<uint32_t> stack;
    stack.push_back(root);
    while (!stack.empty()) {
        uint32_t ni = stack.back();
        stack.pop_back();
        if (ni >= n_nodes)
            return lfail(err, RT_ERR_BAD_SCENE, "node index out of range");
        if (pos_of[ni] != kNone)
            return lfail(err, RT_ERR_BAD_SCENE, "node referenced twice (not a tree)");
        pos_of[ni] = (uint32_t)order_of.size();
        order_of.push_back(ni);
        const rt_node &nd = nodes[ni];
        switch (nd.flags) {
        case RT_NODE_INTERNAL:
            if (order == RT_ORDER_RIGHT_FIRST) {
                stack.push_back(nd.left);
                stack.push_back(nd.right);
            } else {
                stack.push_back(nd.right);
                stack.push_back(nd.left);
            }
            break;
        case RT_NODE_LEAF:
            if (nd.left >= n_tris)
                return lfail(err, RT_ERR_BAD_SCENE, "triangle index out of range");
            break;
        case RT_NODE_SPHERE:
            if (nd.left >= n_spheres)
                return lfail(err, RT_ERR_BAD_SCENE, "sphere index out of range");
            break;
        case RT_NODE_MESH:
            stack.push_back(nd.left);
            break;
        default:
            return lfail(err, RT_ERR_BAD_SCENE, "unknown node kind");
        }
    }
    const uint32_t m = (uint32_t)order_of.size();
    // 2. Subtree sizes, children before parents (children are emitted after their parent).
    std::vector<uint32_t> size(m, 1);
    for (uint32_t p = m; p-- > 0;) {
        const rt_node &nd = nodes[order_of[p]];
        if (nd.flags == RT_NODE_INTERNAL)
            size[p] = 1 + size[pos_of[nd.left]] + size[pos_of[nd.right]];
        else if (nd.flags == RT_NODE_MESH)
            size[p] = 1 + size[pos_of[nd.left]];
    }
    // The fast walk reproduces the reference's result only if every interior box contains the boxes of its children
    // (then a leaf that passes its own slab test has ancestors that passed theirs).  The reference's builders
    // guarantee it; a caller's hand-made tree may not, and is then walked in reference order only.
    L.nested = true;
    for (uint32_t p = 0; p < m && L.nested; p++) {
        const rt_node &nd = nodes[order_of[p]];
        if (nd.flags != RT_NODE_INTERNAL)
            continue;
        for (const uint32_t child : {nd.left, nd.right}) {
            const rt_node &c = nodes[child];
            for (int k = 0; k < 3; k++)
                if (!(nd.pmin[k] <= c.pmin[k]) || !(nd.pmax[k] >= c.pmax[k]))
                    L.nested = false;
        }
    }
    // 3. The fast tree: a surface-area-heuristic hierarchy over the reference's leaf boxes (see fast_bvh.hpp).
    //    Scene-level leaves (triangles of the flat scene, spheres, mesh objects) form the top tree; the
    //    triangles of a mesh object form that mesh's own tree, emitted inline right after the mesh's leaf
    //    record, exactly as the reference nests Mesh::intersect inside the scene traversal.
    struct MeshTree {
        std::vector<fastbvh::Item> items;
        std::vector<fastbvh::Node> nodes;
    };
    std::vector<fastbvh::Item> top_items;
    std::vector<MeshTree> mesh_trees;
    std::vector<uint32_t> mesh_of(m, kNone); // reference position of a mesh leaf -> index into mesh_trees
    {
        uint32_t mesh_end = 0;
        for (uint32_t p = 0; p < m; p++) {
            const rt_node &nd = nodes[order_of[p]];
            if (nd.flags == RT_NODE_INTERNAL)
                continue;
            const bool inside = p < mesh_end;
            fastbvh::Item it;
            for (int k = 0; k < 3; k++) {
                it.bmin[k] = nd.pmin[k];
                it.bmax[k] = nd.pmax[k];
            }
            it.id = p;
            if (nd.flags == RT_NODE_LEAF) {
                (inside ? mesh_trees.back().items : top_items).push_back(it);
            } else {
                if (inside)
                    return lfail(err, RT_ERR_BAD_SCENE, "a mesh's inner tree may hold triangles only");
                top_items.push_back(it);
                if (nd.flags == RT_NODE_MESH) {
                    mesh_of[p] = (uint32_t)mesh_trees.size();
                    mesh_trees.emplace_back();
                    mesh_end = p + size[p];
                }
            }
        }
    }
    std::vector<fastbvh::Node> top_nodes;
    fastbvh::build(top_items, top_nodes);
    for (auto &tree : mesh_trees)
        fastbvh::build(tree.items, tree.nodes);

    // 4. Emit the fast tree, and with it the primitive arrays in fast-tree leaf order.
    std::vector<float4> &fn = L.fn, &tp = L.tp, &tn = L.tn, &sp = L.sp, &mt = L.mt;
    mt.assign(2 * (size_t)n_mats, make_float4(0.f, 0.f, 0.f, 0.f));
    tp.reserve(3 * (size_t)n_tris);
    tn.reserve(3 * (size_t)n_tris);
    std::vector<uint32_t> slot_of(m, 0); // reference position of a leaf -> its triangle / sphere slot
    uint32_t tri_slots = 0, sph_slots = 0;
    auto push_node = [&](const float *bmin, const float *bmax, uint32_t skip_index, uint32_t word) {
        fn.push_back(make_float4(bmin[0], bmin[1], bmin[2], bmax[0]));
        fn.push_back(make_float4(bmax[1], bmax[2], u2f(skip_index * 32u), u2f(word)));
    };
    auto emit_tri = [&](uint32_t p) -> int { // p = reference position of the leaf
        const rt_node &nd = nodes[order_of[p]];
        const rt_tri &t = tris[nd.left];
        if (n_mats && t.mat >= n_mats)
            return RT_ERR_BAD_SCENE;
        slot_of[p] = tri_slots++;
        // edges are the same fp32 subtractions the triangle test starts with (SimplifiedData.cpp:272-273)
        const float e1x = t.pos[1][0] - t.pos[0][0], e1y = t.pos[1][1] - t.pos[0][1], e1z = t.pos[1][2] - t.pos[0][2];
        const float e2x = t.pos[2][0] - t.pos[0][0], e2y = t.pos[2][1] - t.pos[0][1], e2z = t.pos[2][2] - t.pos[0][2];
        tp.push_back(make_float4(t.pos[0][0], t.pos[0][1], t.pos[0][2], e1x));
        tp.push_back(make_float4(e1y, e1z, e2x, e2y));
        tp.push_back(make_float4(e2z, u2f(nd.left), u2f((uint32_t)t.mat), u2f(p)));
        tn.push_back(make_float4(t.nrm[0][0], t.nrm[0][1], t.nrm[0][2], t.nrm[1][0]));
        tn.push_back(make_float4(t.nrm[1][1], t.nrm[1][2], t.nrm[2][0], t.nrm[2][1]));
        tn.push_back(make_float4(t.nrm[2][2], 0.f, 0.f, 0.f));
        return RT_OK;
    };
    const uint32_t n_top = (uint32_t)top_nodes.size();
    std::vector<uint32_t> epos(n_top + 1, 0);
    for (uint32_t q = 0; q < n_top; q++) {
        const fastbvh::Node &fnode = top_nodes[q];
        epos[q] = (uint32_t)(fn.size() / 2);
        if (fnode.item == 0xFFFFFFFFu) {
            push_node(fnode.bmin, fnode.bmax, 0u, KIND_INTERNAL); // skip and the right-child offset are patched below
            continue;
        }
        const uint32_t p = top_items[fnode.item].id;
        const rt_node &nd = nodes[order_of[p]];
        if (nd.flags == RT_NODE_LEAF) {
            if (emit_tri(p) != RT_OK)
                return lfail(err, RT_ERR_BAD_SCENE, "triangle material out of range");
            push_node(fnode.bmin, fnode.bmax, 0u, (slot_of[p] << 3) | KIND_TRI);
        } else if (nd.flags == RT_NODE_SPHERE) {
            const rt_sphere &sph = spheres[nd.left];
            if (sph.mat >= n_mats)
                return lfail(err, RT_ERR_BAD_SCENE, "sphere material out of range");
            const uint32_t dielectric = mats[sph.mat].kind == RT_MAT_DIELECTRIC ? 1u : 0u;
            slot_of[p] = sph_slots++;
            sp.push_back(make_float4(sph.center[0], sph.center[1], sph.center[2], sph.radius));
            sp.push_back(make_float4(u2f(sph.mat), u2f(sph.prim_id), u2f(dielectric), u2f(p)));
            push_node(fnode.bmin, fnode.bmax, 0u, (slot_of[p] << 3) | KIND_SPHERE);
        } else { // mesh object: its leaf record carries the leaf's reference position, its own tree follows
            push_node(fnode.bmin, fnode.bmax, 0u, (p << 3) | KIND_MESH);
            const MeshTree &tree = mesh_trees[mesh_of[p]];
            const uint32_t base = (uint32_t)(fn.size() / 2);
            for (uint32_t j = 0; j < (uint32_t)tree.nodes.size(); j++) {
                const fastbvh::Node &in = tree.nodes[j];
                uint32_t word = KIND_INTERNAL;
                if (in.item == 0xFFFFFFFFu) { // interior: byte offset of the right child (low bits 0 = KIND_INTERNAL)
                    word = (base + j + 1u + tree.nodes[j + 1].size) * 32u;
                } else {
                    const uint32_t pj = tree.items[in.item].id;
                    if (emit_tri(pj) != RT_OK)
                        return lfail(err, RT_ERR_BAD_SCENE, "triangle material out of range");
                    word = (slot_of[pj] << 3) | KIND_TRI;
                }
                push_node(in.bmin, in.bmax, base + j + in.size, word);
            }
        }
    }
    const uint32_t n_fast = (uint32_t)(fn.size() / 2);
    epos[n_top] = n_fast;
    for (uint32_t q = 0; q < n_top; q++) { // a top node's subtree ends where the next top node outside it starts
        fn[2 * (size_t)epos[q] + 1].z = u2f(epos[q + top_nodes[q].size] * 32u);
        if (top_nodes[q].item == 0xFFFFFFFFu) // interior: byte offset of the right child (low bits 0 = KIND_INTERNAL)
            fn[2 * (size_t)epos[q] + 1].w = u2f(epos[q + 1u + top_nodes[q + 1].size] * 32u);
    }

    // 5. The reference's own tree in its visiting order (the exact-order walks), sharing the primitive arrays.
    std::vector<float4> &dn = L.dn;
    dn.assign(2 * (size_t)m + 2, make_float4(0.f, 0.f, 0.f, 0.f)); // + the terminator record
    for (uint32_t p = 0; p < m; p++) {
        const rt_node &nd = nodes[order_of[p]];
        const uint32_t kind = nd.flags;
        const uint32_t ref = (nd.flags == RT_NODE_LEAF || nd.flags == RT_NODE_SPHERE) ? slot_of[p] : 0u;
        dn[2 * (size_t)p] = make_float4(nd.pmin[0], nd.pmin[1], nd.pmin[2], nd.pmax[0]);
        dn[2 * (size_t)p + 1] = make_float4(nd.pmax[1], nd.pmax[2], u2f((p + size[p]) * 32u), u2f((ref << 3) | kind));
    }
    // terminator: kind = all ones, never entered
    dn[2 * (size_t)m] = make_float4(1.f, 1.f, 1.f, -1.f);
    dn[2 * (size_t)m + 1] = make_float4(-1.f, -1.f, u2f(m * 32u), u2f(kNone));
    for (uint32_t i = 0; i < n_mats; i++) {
        mt[2 * (size_t)i] = make_float4(mats[i].albedo[0], mats[i].albedo[1], mats[i].albedo[2], mats[i].albedo[3]);
        mt[2 * (size_t)i + 1] = make_float4(u2f(mats[i].kind), mats[i].param, 0.f, 0.f);
    }

    // Surface-area estimate of the nodes a ray through the root box visits.  Thousands (a tree whose boxes
    // barely prune, like the reference's stretched boxes on the 18520-triangle dragon) favour the
    // wave-synchronous walk; tens to hundreds favour the per-lane walk.
    {
        auto area = [](const rt_node &nd) {
            double e[3];
            for (int k = 0; k < 3; k++) {
                e[k] = (double)nd.pmax[k] - (double)nd.pmin[k];
                if (!(e[k] > 0.0))
                    e[k] = 0.0;
            }
            return 2.0 * (e[0] * e[1] + e[1] * e[2] + e[0] * e[2]);
        };
        double sum = 0.0;
        for (uint32_t p = 0; p < m; p++)
            sum += area(nodes[order_of[p]]);
        const double ra = area(nodes[root]);
        L.expected_visits = ra > 0.0 ? sum / ra : (double)m;
    }
    L.n_ref = m;
    L.n_fast = n_fast;
    L.n_tri_slots = tri_slots;
    L.n_sph_slots = sph_slots;
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
        opt->traversal > RT_TRAV_BRUTE)
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

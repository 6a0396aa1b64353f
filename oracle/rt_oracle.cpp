// rt_oracle.cpp -- CPU oracle: a plain restatement of the reference renderer's CPURayTrace path.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED: the reference ships no tests or
// golden vectors for this path and cannot be compiled here, so this file is pinned by hand-computed
// known-answer tests and self-consistency checks only.
//
// Arithmetic is IEEE fp32 in the operation order of the scalar glm definitions the reference relies
// on (glm itself is a third-party dependency absent from the reference tree, un-pinned in its
// CMakeLists.txt:74):
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   cross(a,b)    = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
//   normalize(v)  = v * (1.0f / sqrt(dot(v,v)))
//   mat3(c0,c1,c2)*v = per row (c0.r*v.x + c1.r*v.y) + c2.r*v.z
//   min(a,b) = (b<a)?b:a     max(a,b) = (a<b)?b:a
// Build with -ffp-contract=off so that no multiply-add is fused.
//
// The reference's random stream (a global std::default_random_engine raced by 16 threads plus
// std::rand, Include/Random.hpp:12-16) is not reproducible; it is replaced by a counter-based
// generator keyed by (seed, pixel, sample, dimension).

#include "rt_oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <vector>

namespace {

// ------------------------------------------------------------------ vector math (glm order)
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V4 {
    float x, y, z, w;
};
inline V3 v3(float a) { return V3{a, a, a}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float fmin_glm(float a, float b) { return (b < a) ? b : a; }
inline float fmax_glm(float a, float b) { return (a < b) ? b : a; }
inline V3 vmin(V3 a, V3 b) { return V3{fmin_glm(a.x, b.x), fmin_glm(a.y, b.y), fmin_glm(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return V3{fmax_glm(a.x, b.x), fmax_glm(a.y, b.y), fmax_glm(a.z, b.z)}; }
inline V3 mat3_mul(V3 c0, V3 c1, V3 c2, V3 v) {
    return V3{(c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z,
              (c0.z * v.x + c1.z * v.y) + c2.z * v.z};
}
inline V4 operator*(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator/(V4 a, float s) { return V4{a.x / s, a.y / s, a.z / s, a.w / s}; }

const float kInf = std::numeric_limits<float>::infinity();

// CPURayTrace/Ray.hpp:8 -- invDir is a component-wise IEEE divide.
struct Ray {
    V3 o, d, inv;
    Ray() : o{0, 0, 0}, d{0, 0, 0}, inv{0, 0, 0} {}
    Ray(V3 origin, V3 dir) : o(origin), d(dir), inv{1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z} {}
    V3 at(float t) const { return o + t * d; } // Ray.hpp:14-17
};

// ------------------------------------------------------------------ counter RNG (new; replaces Random.hpp:11-16)
inline uint32_t pcg(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
inline uint32_t rng_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return pcg(pcg(pcg(seed) + pixel) + sample);
}
inline uint32_t rng_bits(uint32_t key, uint32_t dim) { return pcg(key + dim * 0x9E3779B9u); }
inline float rng_u01(uint32_t key, uint32_t dim) { return (float)(rng_bits(key, dim) >> 8) * (1.0f / 16777216.0f); }
// dimension map: 0,1,2 = ray jitter x,y,z; bounce b: 3+4b = roulette, 4+4b = u1, 5+4b = u2, 6+4b = dielectric choice
enum { DIM_JITTER = 0, DIM_BOUNCE0 = 3, DIM_PER_BOUNCE = 4, DIM_RR = 0, DIM_U1 = 1, DIM_U2 = 2, DIM_CHOICE = 3 };

// cos/sin of Include/Random.hpp:52-53.  ORA_TRIG_LIBM calls the C library as the reference does;
// ORA_TRIG_POLY evaluates a fixed Cody-Waite + minimax polynomial (<= 2 ulp on [0, 2pi]) built from
// fp32 add/mul only, so that a device kernel using the same formula produces the same bits.
inline void sincos_eval(float th, int trig, float *s, float *c) {
    if (trig == ORA_TRIG_LIBM) {
        *s = std::sin(th);
        *c = std::cos(th);
        return;
    }
    int q = (int)(th * 0.636619772f + 0.5f);
    float qf = (float)q;
    float r = ((th - qf * 1.5703125f) - qf * 4.837512969970703125e-4f) - qf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = r + (r * z) * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    float cp = (1.0f - 0.5f * z) + (z * z) * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    switch (q & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}

// Include/Random.hpp:41-72
inline V3 cosine_dir(V3 n, float u1, float u2, int trig) {
    float r = std::sqrt(u1);
    float theta = 6.28318530717958647692f * u2; // 2.0f * glm::pi<float>() * u2
    float sn, cs;
    sincos_eval(theta, trig, &sn, &cs);
    float x = r * cs;
    float y = r * sn;
    float z = std::sqrt(1.0f - u1);
    V3 T, B;
    if (std::fabs(n.z) < 0.99f) {
        T = normalize(cross(n, V3{0.0f, 0.0f, 1.0f}));
        B = cross(T, n);
    } else {
        T = normalize(cross(n, V3{1.0f, 0.0f, 0.0f}));
        B = cross(T, n);
    }
    return mat3_mul(T, B, n, V3{x, y, z});
}

// Include/Camera.hpp:33-47
inline V3 camera_dir(const ora_camera &c, float u, float v) {
    V3 view = normalize(V3{u * c.width - c.width / 2, v * c.height - c.height / 2, c.focal_length});
    view.x = -view.x;
    V3 absY{0.f, 1.f, 0.f};
    V3 pos{c.position[0], c.position[1], c.position[2]};
    V3 look{c.look_at[0], c.look_at[1], c.look_at[2]};
    V3 z = normalize(look - pos); // DirectionOf(end, ori), Include/Utils.hpp:52-55
    V3 x = normalize(cross(absY, z));
    V3 y = cross(z, x);
    return mat3_mul(x, y, z, view);
}

// Sky gradient: CPURayTrace/Trace.cpp:86-89 and Materials/Sky.hpp:12-14
inline V3 sky_color(V3 dir) {
    V3 unit = normalize(dir);
    float a = 0.5f * (unit.y + 1.0f);
    return (1.0f - a) * V3{1.0f, 1.0f, 1.0f} + a * V3{0.5f, 0.7f, 1.0f};
}

// CPURayTrace/SimplifiedData.cpp:146-161 == CPURayTrace/Bounding.hpp:9-24
inline bool box_test(V3 pmin, V3 pmax, const Ray &ray, float tMin, float tMax) {
    for (int a = 0; a < 3; a++) {
        float invD = ray.inv[a];
        float t0 = (pmin[a] - ray.o[a]) * invD;
        float t1 = (pmax[a] - ray.o[a]) * invD;
        if (invD < 0.0f)
            std::swap(t0, t1);
        tMin = t0 > tMin ? t0 : tMin;
        tMax = t1 < tMax ? t1 : tMax;
        if (tMax <= tMin)
            return false;
    }
    return true;
}

struct Counters {
    uint64_t rays = 0, box = 0, tri = 0, sph = 0, tri_hits = 0;
    void add(const Counters &o) {
        rays += o.rays; box += o.box; tri += o.tri; sph += o.sph; tri_hits += o.tri_hits;
    }
};

// Moller-Trumbore of SimplifiedData.cpp:269-307 == Mesh.hpp:39-77.  tMin/tMax are ignored there too.
struct TriHit {
    bool hit = false;
    float t = kInf, u = 0, v = 0;
};
inline TriHit tri_test(V3 p0, V3 p1, V3 p2, const Ray &ray) {
    const float EPSILON = 1e-7f;
    TriHit r;
    V3 edge1 = p1 - p0;
    V3 edge2 = p2 - p0;
    V3 h = cross(ray.d, edge2);
    float a = dot(edge1, h);
    if (a > -EPSILON && a < EPSILON)
        return r;
    float f = 1.0f / a;
    V3 s = ray.o - p0;
    float u = f * dot(s, h);
    if (u < 0.0 || u > 1.0)
        return r;
    V3 q = cross(s, edge1);
    float v = f * dot(ray.d, q);
    if (v < 0.0 || u + v > 1.0)
        return r;
    float t = f * dot(edge2, q);
    if (t > EPSILON) {
        r.hit = true; r.t = t; r.u = u; r.v = v;
    }
    return r;
}
inline V3 tri_normal(V3 n0, V3 n1, V3 n2, float u, float v) {
    return normalize((1 - u - v) * n0 + u * n1 + v * n2); // SimplifiedData.cpp:293
}

static std::atomic<uint64_t> g_leaf_replays{0}; // rays ORA_TRAV_LEAVES resolved by the ordered replay
static std::atomic<uint64_t> g_leaves_skipped{0}, g_leaves_met{0}; // ORA_TRAV_LIMITED: leaves beyond the limit / leaves whose box the ray meets

// =================================================================== flat-array ("sd") scene
enum : uint32_t { NODE_INTERNAL = 0, NODE_LEAF = 1 }; // SimplifiedData.hpp:30-36
struct SdNode { // SimplifiedData.hpp:71-77
    uint32_t left, right;
    V3 pmin, pmax;
    uint32_t flags;
};
struct SdTri { // SimplifiedData.hpp:82-88
    V3 p[3], n[3];
    float uv[3][2];
    uint32_t mat;
};
struct SdHit { // SimplifiedData.hpp:52-62 plus the primitive id the reference does not keep
    bool hit = false;
    float t = kInf;
    V3 origin{0, 0, 0}, dir{0, 0, 0}, pos{0, 0, 0}, normal{0, 0, 0};
    uint32_t mat = 0, prim = 0xFFFFFFFFu;
};

struct SdScene {
    std::vector<SdNode> nodes;
    std::vector<SdTri> tris;
    std::vector<uint32_t> sceneIndices;
    uint32_t root = 0xFFFFFFFFu;

    uint32_t addNode(const SdNode &n) {
        nodes.push_back(n);
        return (uint32_t)nodes.size() - 1;
    }

    // SimplifiedData.cpp:309-320
    static void triBox(const SdTri &t, V3 &mn, V3 &mx) {
        mn = vmin(t.p[0], vmin(t.p[1], t.p[2]));
        mx = vmax(t.p[0], vmax(t.p[1], t.p[2]));
        const V3 bias = v3(1e-5f);
        mn = mn - bias;
        mx = mx + bias;
    }

    // SimplifiedData.cpp:322-399
    uint32_t build(uint32_t *idx, size_t start, size_t end) {
        if (end - start == 1)
            return idx[start];
        if (end - start == 2) {
            uint32_t l = idx[start], r = idx[start + 1];
            SdNode n;
            n.left = l; n.right = r;
            n.pmin = vmin(nodes[l].pmin, nodes[r].pmin);
            n.pmax = vmax(nodes[l].pmax, nodes[r].pmax);
            n.flags = NODE_INTERNAL;
            return addNode(n);
        }
        V3 bmin = v3(FLT_MAX), bmax = v3(FLT_MIN); // default BoundingBox, SimplifiedData.hpp:64-69
        for (uint32_t i = (uint32_t)start; i < end; i++) {
            const SdNode &s = nodes[idx[i]];
            bmin = vmin(bmin, s.pmin);
            bmax = vmax(bmax, s.pmax);
        }
        float xe = bmax.x - bmin.x, ye = bmax.y - bmin.y, ze = bmax.z - bmin.z;
        const std::vector<SdNode> &N = nodes;
        auto cx = [&N](const uint32_t &a, const uint32_t &b) { return N[a].pmin.x + N[a].pmax.x < N[b].pmin.x + N[b].pmax.x; };
        auto cy = [&N](const uint32_t &a, const uint32_t &b) { return N[a].pmin.y + N[a].pmax.y < N[b].pmin.y + N[b].pmax.y; };
        auto cz = [&N](const uint32_t &a, const uint32_t &b) { return N[a].pmin.z + N[a].pmax.z < N[b].pmin.z + N[b].pmax.z; };
        if (xe >= ye && xe >= ze)
            std::sort(idx + start, idx + end, cx);
        else if (ye >= xe && ye >= ze)
            std::sort(idx + start, idx + end, cy);
        else
            std::sort(idx + start, idx + end, cz);
        size_t mid = start + (end - start) / 2;
        uint32_t l = build(idx, start, mid);
        uint32_t r = build(idx, mid, end);
        SdNode n;
        n.left = l; n.right = r; n.pmin = bmin; n.pmax = bmax; n.flags = NODE_INTERNAL;
        return addNode(n);
    }

    // sd::Mesh::Mesh, SimplifiedData.cpp:98-136
    uint32_t addMesh(const float *t24, uint32_t n) {
        std::vector<uint32_t> leafIdx;
        for (uint32_t i = 0; i < n; i++) {
            const float *f = t24 + (size_t)i * 24;
            SdTri t;
            for (int k = 0; k < 3; k++) {
                t.p[k] = V3{f[k * 8 + 0], f[k * 8 + 1], f[k * 8 + 2]};
                t.n[k] = V3{f[k * 8 + 3], f[k * 8 + 4], f[k * 8 + 5]};
                t.uv[k][0] = f[k * 8 + 6];
                t.uv[k][1] = f[k * 8 + 7];
            }
            t.mat = 0; // LambertianMat, SimplifiedData.cpp:119
            tris.push_back(t);
            uint32_t ti = (uint32_t)tris.size() - 1;
            SdNode leaf;
            leaf.left = ti; leaf.right = ti;
            triBox(t, leaf.pmin, leaf.pmax);
            leaf.flags = NODE_LEAF;
            leafIdx.push_back(addNode(leaf));
        }
        return build(leafIdx.data(), 0, leafIdx.size());
    }

    SdHit leafHit(const SdNode &node, const Ray &ray, Counters &c) const {
        const SdTri &tri = tris[node.left];
        c.tri++;
        TriHit th = tri_test(tri.p[0], tri.p[1], tri.p[2], ray);
        SdHit h;
        if (th.hit) {
            h.hit = true; h.t = th.t; h.origin = ray.o; h.dir = ray.d;
            h.normal = tri_normal(tri.n[0], tri.n[1], tri.n[2], th.u, th.v);
            h.pos = ray.at(th.t);
            h.mat = tri.mat; h.prim = node.left;
        }
        return h;
    }

    // sd::BVH::IntersectLoop, SimplifiedData.cpp:430-462 (children pushed left then right => right popped first)
    SdHit intersectLoop(const Ray &ray, Counters &c) const {
        uint32_t stack[128];
        size_t top = 0;
        SdHit closest;
        stack[top++] = root;
        while (top > 0) {
            uint32_t index = stack[--top];
            if (index == 0xFFFFFFFFu)
                continue;
            const SdNode &node = nodes[index];
            c.box++;
            if (!box_test(node.pmin, node.pmax, ray, 1e-6f, closest.t))
                continue;
            if (node.flags == NODE_LEAF) {
                SdHit h = leafHit(node, ray, c);
                if (h.hit && h.t < closest.t)
                    closest = h;
                continue;
            }
            stack[top++] = node.left;
            stack[top++] = node.right;
        }
        return closest;
    }
    // sd::BVH::Intersect, SimplifiedData.cpp:401-428 (left subtree first)
    void intersectRec(uint32_t index, const Ray &ray, SdHit &closest, Counters &c) const {
        if (index == 0xFFFFFFFFu)
            return;
        const SdNode &node = nodes[index];
        c.box++;
        if (!box_test(node.pmin, node.pmax, ray, 1e-6f, closest.t))
            return;
        if (node.flags == NODE_LEAF) {
            SdHit h = leafHit(node, ray, c);
            if (h.hit && h.t < closest.t)
                closest = h;
            return;
        }
        intersectRec(node.left, ray, closest, c);
        intersectRec(node.right, ray, closest, c);
    }
    // ---- the reference's result from its LEAVES alone (checks the argument the product's fast walk rests on) ----
    // In the reference traversal the interior boxes only prune, and every interior box contains the leaf boxes below
    // it, so a leaf whose own box passes "(1e-6, closest)" has ancestors that passed: the result is a function of the
    // leaves.  A leaf is a candidate if the ray meets its box at all (closest = +inf) and the triangle test hits at t;
    // m = the slab test's entry distance; pos = the leaf's place in the loop's visiting order.  The reference accepts a
    // candidate iff closest > m and t < closest, in pos order.  If the candidate with the smallest (t, pos) has m <= t
    // it is the result; otherwise the acceptance rule is replayed over the candidates in pos order.  The leaves are
    // visited here in REVERSE storage order, to show that the order does not matter.
    mutable std::vector<uint32_t> leafPos; // node index -> position in IntersectLoop's visiting order (leaves only)
    void orderLeaves() const {
        leafPos.assign(nodes.size(), 0xFFFFFFFFu);
        std::vector<uint32_t> stack{root};
        uint32_t pos = 0;
        while (!stack.empty()) {
            const uint32_t index = stack.back();
            stack.pop_back();
            const SdNode &node = nodes[index];
            if (node.flags == NODE_LEAF) {
                leafPos[index] = pos++;
                continue;
            }
            pos++;
            stack.push_back(node.left);
            stack.push_back(node.right);
        }
    }
    static bool box_entry(V3 pmin, V3 pmax, const Ray &ray, float *entry) { // box_test with (1e-6, +inf), keeping tMin
        float tMin = 1e-6f, tMax = kInf;
        for (int a = 0; a < 3; a++) {
            float invD = ray.inv[a];
            float t0 = (pmin[a] - ray.o[a]) * invD;
            float t1 = (pmax[a] - ray.o[a]) * invD;
            if (invD < 0.0f)
                std::swap(t0, t1);
            tMin = t0 > tMin ? t0 : tMin;
            tMax = t1 < tMax ? t1 : tMax;
            if (tMax <= tMin)
                return false;
        }
        *entry = tMin;
        return true;
    }
    SdHit intersectLeaves(const Ray &ray, Counters &c, bool *replayed) const {
        struct Cand {
            uint32_t pos, node;
            float t, m;
        };
        std::vector<Cand> cands;
        for (uint32_t index = (uint32_t)nodes.size(); index-- > 0;) {
            const SdNode &node = nodes[index];
            if (node.flags != NODE_LEAF || leafPos[index] == 0xFFFFFFFFu)
                continue;
            c.box++;
            float m;
            if (!box_entry(node.pmin, node.pmax, ray, &m))
                continue;
            const SdTri &tri = tris[node.left];
            c.tri++;
            TriHit th = tri_test(tri.p[0], tri.p[1], tri.p[2], ray);
            if (th.hit)
                cands.push_back(Cand{leafPos[index], index, th.t, m});
        }
        const Cand *best = nullptr;
        for (const Cand &k : cands)
            if (k.t < kInf && (!best || k.t < best->t || (k.t == best->t && k.pos < best->pos)))
                best = &k;
        *replayed = false;
        if (best && best->m > best->t) { // the winner lies in front of its own leaf box: play the rule in order
            *replayed = true;
            std::sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.pos < b.pos; });
            float closest = kInf;
            best = nullptr;
            for (const Cand &k : cands)
                if (closest > k.m && k.t < closest) {
                    closest = k.t;
                    best = &k;
                }
        }
        SdHit h;
        if (best) {
            Counters scratch;
            h = leafHit(nodes[best->node], ray, scratch);
        }
        return h;
    }

    // ---- the same with a DISTANCE LIMIT (checks the argument the product's wide walk rests on) ----
    // A walk that already knows a hit at distance tau may skip a leaf whose box the ray enters at m, if no ray for which
    // the triangle test reports a hit can have m - t as large as that: then a skipped leaf cannot be a candidate with
    // t <= tau, and only such candidates can change the outcome (the candidate with the smallest (t, pos) is still found,
    // and so is every candidate below it that could force the replay).  The margin is the caller's: per size class k of the
    // leaf, skip when m >= (tau + c1[k] / D + c3[k]) * (1 + c2x2[k] * D) * 1.00001, D = |d|_inf in [0.25, 1.25]; class 3 is
    // never skipped.  The leaves are visited NEAREST BOX FIRST, the order in which the limit cuts most.
    std::vector<uint8_t> limitClass; // per triangle
    float limitC1[3] = {0, 0, 0}, limitC2x2[3] = {0, 0, 0}, limitC3[3] = {0, 0, 0};
    SdHit intersectLimited(const Ray &ray, Counters &c, bool *replayed) const {
        struct Boxed {
            float m;
            uint32_t node;
        };
        std::vector<Boxed> met;
        for (uint32_t index = 0; index < (uint32_t)nodes.size(); index++) {
            const SdNode &node = nodes[index];
            if (node.flags != NODE_LEAF || leafPos[index] == 0xFFFFFFFFu)
                continue;
            c.box++;
            float m;
            if (box_entry(node.pmin, node.pmax, ray, &m))
                met.push_back(Boxed{m, index});
        }
        std::sort(met.begin(), met.end(), [](const Boxed &a, const Boxed &b) { return a.m < b.m || (a.m == b.m && a.node < b.node); });
        const float D = std::max(std::fabs(ray.d.x), std::max(std::fabs(ray.d.y), std::fabs(ray.d.z)));
        const bool limited = !limitClass.empty() && D >= 0.25f && D <= 1.25f;
        const float invD = limited ? 1.0f / D : 1.0f;
        float tau = kInf;
        bool have = false;
        float bt = kInf, bmm = 0.0f;
        uint32_t bpos = 0xFFFFFFFFu, bnode = 0;
        uint64_t skipped = 0;
        for (const Boxed &b : met) {
            const SdNode &node = nodes[b.node];
            const uint32_t cls = limited ? limitClass[node.left] : 3u;
            if (cls < 3u) {
                const float limit = (tau + std::fma(limitC1[cls], invD, limitC3[cls])) * std::fma(limitC2x2[cls], D, 1.00001f);
                if (b.m >= limit) {
                    skipped++;
                    continue;
                }
            }
            const SdTri &tri = tris[node.left];
            c.tri++;
            TriHit th = tri_test(tri.p[0], tri.p[1], tri.p[2], ray);
            if (!th.hit || !(th.t < kInf))
                continue;
            const uint32_t pos = leafPos[b.node];
            if (!have || th.t < bt || (th.t == bt && pos < bpos)) {
                have = true;
                bt = th.t;
                bpos = pos;
                bmm = b.m;
                bnode = b.node;
            }
            tau = th.t < tau ? th.t : tau;
        }
        g_leaves_skipped += skipped;
        g_leaves_met += met.size();
        *replayed = false;
        SdHit h;
        if (!have)
            return h;
        if (bmm > bt) { // the winner lies in front of its own leaf box: nothing may be skipped, play the rule over all leaves
            Counters scratch;
            h = intersectLeaves(ray, scratch, replayed);
            *replayed = true;
            return h;
        }
        Counters scratch;
        return leafHit(nodes[bnode], ray, scratch);
    }

    // brute force over all triangles in storage order (the "BVH acceleration off" debugging aid)
    SdHit intersectBrute(const Ray &ray, Counters &c) const {
        SdHit closest;
        for (uint32_t i = 0; i < tris.size(); i++) {
            SdNode fake;
            fake.left = fake.right = i;
            SdHit h = leafHit(fake, ray, c);
            if (h.hit && h.t < closest.t)
                closest = h;
        }
        return closest;
    }
    SdHit intersect(const Ray &ray, int trav, Counters &c) const {
        c.rays++;
        SdHit h;
        if (trav == ORA_TRAV_RECURSIVE)
            intersectRec(root, ray, h, c);
        else if (trav == ORA_TRAV_BRUTE)
            h = intersectBrute(ray, c);
        else if (trav == ORA_TRAV_LEAVES || trav == ORA_TRAV_LIMITED) {
            bool replayed = false;
            h = trav == ORA_TRAV_LEAVES ? intersectLeaves(ray, c, &replayed) : intersectLimited(ray, c, &replayed);
            if (replayed)
                g_leaf_replays++;
        } else
            h = intersectLoop(ray, c);
        if (h.hit)
            c.tri_hits++;
        return h;
    }
};

// Trace::CastRay(ray, depth, sd::DataStorage&), CPURayTrace/Trace.cpp:58-96
V4 castRaySd(const SdScene &sc, Ray ray, const ora_options &opt, uint32_t key, Counters &c, uint32_t *primary) {
    V4 color{0, 0, 0, 0};
    V3 thr = v3(1.f);
    int depth = 0;
    while (depth < opt.bounce_limit) {
        uint32_t dimBase = DIM_BOUNCE0 + DIM_PER_BOUNCE * (uint32_t)depth;
        depth++;
        SdHit hit = sc.intersect(ray, opt.traversal, c);
        if (depth == 1 && primary)
            *primary = hit.hit ? hit.prim : 0xFFFFFFFFu;
        if (hit.hit) {
            if (hit.mat == 0) { // any other material leaves the ray unchanged and the loop re-casts it (Trace.cpp:80-82)
                // Lambertian::Hit, Materials/Lambertian.hpp:35-48 (bias 1e-5, fixed albedo of Trace.cpp:81)
                V3 bias = hit.normal * 1e-5f;
                float u1 = rng_u01(key, dimBase + DIM_U1);
                float u2 = rng_u01(key, dimBase + DIM_U2);
                V3 rnd = cosine_dir(hit.normal, u1, u2, opt.trig);
                ray = Ray(hit.pos + bias, rnd);
                thr = thr * V3{0.9f, 0.6f, 0.5f};
            }
            continue;
        }
        V3 sky = sky_color(ray.d);
        V3 add = thr * sky;
        color = color + V4{add.x, add.y, add.z, 1.0f};
        break;
    }
    return color;
}

// =================================================================== pointer-based ("old") scene
struct Material {
    uint32_t kind;
    V4 albedo; // Lambertian/Metal albedo, LightEmit intensity
    float param; // Metal gross / Dielectric index of refraction
};
struct OldHit { // CPURayTrace/Materials.hpp:8-17
    float t = kInf;
    V3 origin{0, 0, 0}, dir{0, 0, 0}, pos{0, 0, 0}, normal{0, 0, 0};
    const Material *mat = nullptr;
    uint32_t prim = 0xFFFFFFFFu;
};
struct OldNode;
struct Hittable {
    V3 bmin, bmax;
    Material mat;
    uint32_t id = 0;
    virtual ~Hittable() {}
    virtual bool intersect(const Ray &ray, OldHit &out, Counters &c, bool useBvh) const = 0;
};
struct OldNode { // CPURayTrace/BVH.hpp:6-12
    V3 bmin{0, 0, 0}, bmax{0, 0, 0};
    OldNode *left = nullptr, *right = nullptr;
    const Hittable *object = nullptr;
};
struct OldBVH {
    std::vector<std::unique_ptr<OldNode>> pool;
    OldNode *root = nullptr;
    OldNode *make() {
        pool.emplace_back(new OldNode());
        return pool.back().get();
    }
    // BVH::BuildBVH, CPURayTrace/BVH.hpp:20-71.  Sorts the object array in place.
    OldNode *build(std::vector<Hittable *> &objs, int start, int end) {
        if (end - start <= 0)
            return nullptr;
        if (end - start == 1) {
            OldNode *leaf = make();
            leaf->bmin = objs[start]->bmin; leaf->bmax = objs[start]->bmax;
            leaf->object = objs[start];
            return leaf;
        }
        if (end - start == 2) {
            OldNode *n = make();
            n->left = build(objs, start, start + 1);
            n->right = build(objs, start + 1, end);
            n->bmin = vmin(n->left->bmin, n->right->bmin);
            n->bmax = vmax(n->left->bmax, n->right->bmax);
            return n;
        }
        OldNode *n = make(); // value-initialised: the box starts at all zeros (BVH.hpp:42)
        for (int i = start; i < end; i++) {
            n->bmin = vmin(n->bmin, objs[i]->bmin);
            n->bmax = vmax(n->bmax, objs[i]->bmax);
        }
        float xe = n->bmax.x - n->bmin.x, ye = n->bmax.y - n->bmin.y, ze = n->bmax.z - n->bmin.z;
        if (xe >= ye && xe >= ze)
            std::sort(objs.begin() + start, objs.begin() + end, [](Hittable *a, Hittable *b) { return a->bmin.x + a->bmax.x < b->bmin.x + b->bmax.x; });
        else if (ye >= xe && ye >= ze)
            std::sort(objs.begin() + start, objs.begin() + end, [](Hittable *a, Hittable *b) { return a->bmin.y + a->bmax.y < b->bmin.y + b->bmax.y; });
        else
            std::sort(objs.begin() + start, objs.begin() + end, [](Hittable *a, Hittable *b) { return a->bmin.z + a->bmax.z < b->bmin.z + b->bmax.z; });
        int mid = start + (end - start) / 2;
        n->left = build(objs, start, mid);
        n->right = build(objs, mid, end);
        return n;
    }
    // BVH::intersect, BVH.hpp:124-148 (left subtree first)
    void traverse(const OldNode *node, const Ray &ray, OldHit &closest, Counters &c) const {
        if (!node)
            return;
        c.box++;
        if (!box_test(node->bmin, node->bmax, ray, 1e-6f, closest.t))
            return;
        if (node->object) {
            OldHit h;
            if (node->object->intersect(ray, h, c, true) && h.t < closest.t)
                closest = h;
            return;
        }
        traverse(node->left, ray, closest, c);
        traverse(node->right, ray, closest, c);
    }
};

struct Sphere : Hittable { // CPURayTrace/Objects.hpp:26-85
    V3 center;
    float radius;
    bool intersect(const Ray &ray, OldHit &out, Counters &c, bool) const override {
        c.sph++;
        V3 oc = center - ray.o;
        float a = dot(ray.d, ray.d);
        float b = -2.0f * dot(ray.d, oc);
        float cc = dot(oc, oc) - radius * radius;
        float disc = b * b - 4 * a * cc;
        if (disc < 0)
            return false;
        float t = (-b - std::sqrt(disc)) / (2.0f * a);
        if (t < 0) {
            // The reference stops here (Objects.hpp:57-60): a ray that starts inside a sphere never
            // hits it.  The dielectric material, which the reference lacks, needs the exit point,
            // so for dielectric spheres only the far root is tried as well.
            if (mat.kind != ORA_MAT_DIELECTRIC)
                return false;
            t = (-b + std::sqrt(disc)) / (2.0f * a);
            if (t < 0)
                return false;
        }
        out.t = t; out.origin = ray.o; out.dir = ray.d; out.pos = ray.at(t);
        out.normal = normalize(ray.at(t) - center);
        out.mat = &mat; out.prim = id;
        return true;
    }
};
struct OldTriangle : Hittable { // CPURayTrace/Mesh.hpp:18-90
    V3 p[3], n[3];
    bool intersect(const Ray &ray, OldHit &out, Counters &c, bool) const override {
        c.tri++;
        TriHit th = tri_test(p[0], p[1], p[2], ray);
        if (!th.hit)
            return false;
        out.t = th.t; out.origin = ray.o; out.dir = ray.d; out.pos = ray.at(th.t);
        out.normal = tri_normal(n[0], n[1], n[2], th.u, th.v);
        out.mat = &mat; out.prim = id;
        return true;
    }
};
struct OldMesh : Hittable { // CPURayTrace/Mesh.hpp:92-188
    std::vector<std::unique_ptr<OldTriangle>> tris;
    OldBVH inside;
    // Mesh::intersect returns an engaged optional even on a miss (t = inf), Mesh.hpp:159,172
    bool intersect(const Ray &ray, OldHit &out, Counters &c, bool useBvh) const override {
        OldHit closest;
        if (useBvh) {
            inside.traverse(inside.root, ray, closest, c);
        } else {
            for (auto &t : tris) {
                OldHit h;
                if (t->intersect(ray, h, c, false) && h.t < closest.t)
                    closest = h;
            }
        }
        out = closest;
        return true;
    }
};

struct OldScene { // Program/Scene.hpp:13-96
    std::vector<std::unique_ptr<Hittable>> owned;
    std::vector<Hittable *> objects;
    OldBVH bvh;
    uint32_t nSpheres = 0, nTris = 0; // primitive ids: triangle index, or 0x40000000 | sphere index

    OldHit intersectClosest(const Ray &ray, int trav, Counters &c) const {
        c.rays++;
        OldHit closest;
        if (trav == ORA_TRAV_BRUTE) { // Scene::intersectClosest, Scene.hpp:83-95
            for (auto *o : objects) {
                OldHit h;
                if (o->intersect(ray, h, c, false) && h.t < closest.t)
                    closest = h;
            }
        } else {
            bvh.traverse(bvh.root, ray, closest, c);
        }
        return closest;
    }
};

inline float reflectance(float cosine, float ri) { // Schlick, RTIOW form (no reference counterpart)
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    float m = 1.0f - cosine;
    float m2 = m * m;
    return r0 + (1.0f - r0) * ((m2 * m2) * m);
}

// One scattering event.  Returns false when the path ends here (emitter), true with the next ray.
// Lambertian: Materials/Lambertian.hpp:16-29.  Metal: Materials/Metal.hpp:15-31.
// LightEmit: Materials/LightEmit.hpp:11-14.  Dielectric: defined by this build (RTIOW Snell/Schlick).
inline bool scatter(const OldHit &hit, uint32_t key, uint32_t dimBase, int trig, Ray &next, V4 &atten) {
    const Material &m = *hit.mat;
    if (m.kind == ORA_MAT_EMIT) {
        atten = m.albedo;
        return false;
    }
    if (m.kind == ORA_MAT_DIELECTRIC) {
        V3 unit = normalize(hit.dir);
        bool front = dot(unit, hit.normal) < 0.0f;
        V3 n = front ? hit.normal : -hit.normal;
        float ri = front ? (1.0f / m.param) : m.param;
        float cosT = fmin_glm(dot(-unit, n), 1.0f);
        float sinT = std::sqrt(1.0f - cosT * cosT);
        bool cannot = ri * sinT > 1.0f;
        float pick = rng_u01(key, dimBase + DIM_CHOICE);
        V3 bias = n * 1e-4f;
        if (cannot || reflectance(cosT, ri) > pick) {
            V3 refl = unit - n * dot(unit, n) * 2.f;
            next = Ray(hit.pos + bias, refl);
        } else {
            V3 perp = ri * (unit + cosT * n);
            V3 par = n * -std::sqrt(std::fabs(1.0f - dot(perp, perp)));
            next = Ray(hit.pos - bias, perp + par);
        }
        atten = V4{1.0f, 1.0f, 1.0f, 1.0f};
        return true;
    }
    V3 bias = hit.normal * 1e-4f;
    float u1 = rng_u01(key, dimBase + DIM_U1);
    float u2 = rng_u01(key, dimBase + DIM_U2);
    V3 rnd = cosine_dir(hit.normal, u1, u2, trig);
    if (m.kind == ORA_MAT_METAL) {
        V3 refl = hit.dir - hit.normal * dot(hit.dir, hit.normal) * 2.f;
        V3 d = normalize(rnd * m.param + refl * (1 - m.param));
        next = Ray(hit.pos + bias, d);
    } else {
        next = Ray(hit.pos + bias, rnd);
    }
    atten = m.albedo;
    return true;
}

inline float roulette(int depth, uint32_t key, uint32_t dimBase) { // Trace.cpp:33, Random.hpp:74-78
    if (depth <= 1)
        return 1.0f;
    return rng_u01(key, dimBase + DIM_RR) < 0.8f ? 1.0f / 0.8f : 0.0f;
}

// Trace::CastRay(ray, depth, const Scene&), CPURayTrace/Trace.cpp:31-56, as written (recursive).
V4 castRayOldRec(const OldScene &sc, const Ray &ray, int depth, const ora_options &opt, uint32_t key, Counters &c,
                 uint32_t *primary) {
    uint32_t dimBase = DIM_BOUNCE0 + DIM_PER_BOUNCE * (uint32_t)depth;
    float rr = roulette(depth, key, dimBase);
    if (depth > opt.bounce_limit || rr == 0.0f)
        return V4{0, 0, 0, 0};
    OldHit hit = sc.intersectClosest(ray, opt.traversal, c);
    if (depth == 0 && primary)
        *primary = hit.t != kInf ? hit.prim : 0xFFFFFFFFu;
    if (hit.t != kInf) {
        Ray next;
        V4 atten;
        if (!scatter(hit, key, dimBase, opt.trig, next, atten))
            return atten * rr;
        return (atten * castRayOldRec(sc, next, depth + 1, opt, key, c, nullptr)) * rr;
    }
    V3 s = sky_color(ray.d);
    return V4{s.x, s.y, s.z, 1.0f} * rr;
}

// The same estimator with the products taken front to back (throughput form), which is the order a
// loop on a GPU uses.  Differs from the recursive form only in fp32 rounding of the product chain.
V4 castRayOldFwd(const OldScene &sc, Ray ray, const ora_options &opt, uint32_t key, Counters &c, uint32_t *primary) {
    V4 thr{1, 1, 1, 1};
    for (int depth = 0;; depth++) {
        uint32_t dimBase = DIM_BOUNCE0 + DIM_PER_BOUNCE * (uint32_t)depth;
        float rr = roulette(depth, key, dimBase);
        if (depth > opt.bounce_limit || rr == 0.0f)
            return V4{0, 0, 0, 0};
        OldHit hit = sc.intersectClosest(ray, opt.traversal, c);
        if (depth == 0 && primary)
            *primary = hit.t != kInf ? hit.prim : 0xFFFFFFFFu;
        if (hit.t != kInf) {
            Ray next;
            V4 atten;
            bool more = scatter(hit, key, dimBase, opt.trig, next, atten);
            thr = (thr * atten) * rr;
            if (!more)
                return thr;
            ray = next;
            continue;
        }
        V3 s = sky_color(ray.d);
        return (thr * V4{s.x, s.y, s.z, 1.0f}) * rr;
    }
}

// =================================================================== per-pixel driver
// TraceSdSceneCPU::trace / TraceSceneCPU::trace, Renderer/TraceMethods.cpp:45-119: one sample per
// pixel per call, numThreads row bands started with std::async, running mean per pixel.
template <class Shade>
void renderLoop(const ora_camera &cam, const ora_options &opt, int W, int H, int first, int count, float *image,
                uint32_t *primary, ora_stats *stats, Shade shadeRay) {
    auto t0 = std::chrono::steady_clock::now();
    int bands = opt.threads > 0 ? opt.threads : 1;
    std::shared_mutex sceneMutex;
    Counters total;
    std::mutex totalMutex;
    V3 camPos{cam.position[0], cam.position[1], cam.position[2]};
    for (int n = first; n < first + count; n++) {
        std::vector<std::future<void>> futures;
        size_t rowsPerThread = (size_t)H / (size_t)bands;
        for (int i = 0; i < bands; i++) {
            size_t startY = i * rowsPerThread;
            size_t endY = (i == bands - 1) ? (size_t)H : startY + rowsPerThread;
            futures.push_back(std::async(std::launch::async, [&, startY, endY, n]() {
                Counters local;
                for (size_t y = startY; y < endY; ++y) {
                    for (size_t x = 0; x < (size_t)W; ++x) {
                        if (opt.tile_world > 1) { // partition emulation only; the reference has no tiles
                            size_t tile = (y / 8) * (((size_t)W + 7) / 8) + (x / 8);
                            if ((int)(tile % (size_t)opt.tile_world) != opt.tile_rank)
                                continue;
                        }
                        uint32_t pixel = (uint32_t)(y * (size_t)W + x);
                        uint32_t key = rng_key(opt.seed, pixel, (uint32_t)n + opt.sample_offset);
                        float u = x / float(W), v = y / float(H); // CPUImageData::uvAt, TraceMethods.hpp:30
                        // glm::linearRand(vec3(-s), vec3(s)) : r * (max - min) + min per component
                        float s = opt.jitter;
                        V3 jit{rng_u01(key, DIM_JITTER + 0) * (s - (-s)) + (-s), rng_u01(key, DIM_JITTER + 1) * (s - (-s)) + (-s),
                               rng_u01(key, DIM_JITTER + 2) * (s - (-s)) + (-s)};
                        Ray ray(camPos, camera_dir(cam, u, v) + jit);
                        uint32_t *prim = (primary && n == first) ? &primary[pixel] : nullptr;
                        V4 c;
                        if (opt.lock_per_pixel) {
                            std::shared_lock<std::shared_mutex> lock(sceneMutex); // TraceMethods.cpp:57
                            c = shadeRay(ray, key, local, prim);
                        } else {
                            c = shadeRay(ray, key, local, prim);
                        }
                        float *px = image + (size_t)pixel * 4;
                        V4 old{px[0], px[1], px[2], px[3]};
                        V4 upd = (old * static_cast<float>(n - 1.f) + c) / static_cast<float>(n); // TraceMethods.cpp:59
                        px[0] = upd.x; px[1] = upd.y; px[2] = upd.z; px[3] = upd.w;
                    }
                }
                std::lock_guard<std::mutex> g(totalMutex);
                total.add(local);
            }));
        }
        for (auto &f : futures)
            f.get();
    }
    if (stats) {
        stats->rays = total.rays; stats->box_tests = total.box; stats->tri_tests = total.tri;
        stats->sph_tests = total.sph; stats->tri_hits = total.tri_hits;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
}

void packNode36(const SdNode &n, uint8_t *o) {
    uint32_t w[9];
    w[0] = n.left; w[1] = n.right;
    std::memcpy(&w[2], &n.pmin, 12);
    std::memcpy(&w[5], &n.pmax, 12);
    w[8] = n.flags;
    std::memcpy(o, w, 36);
}
void packTri100(const SdTri &t, uint8_t *o) {
    float f[25];
    for (int k = 0; k < 3; k++) {
        f[k * 3 + 0] = t.p[k].x; f[k * 3 + 1] = t.p[k].y; f[k * 3 + 2] = t.p[k].z;
        f[9 + k * 3 + 0] = t.n[k].x; f[9 + k * 3 + 1] = t.n[k].y; f[9 + k * 3 + 2] = t.n[k].z;
        f[18 + k * 2 + 0] = t.uv[k][0]; f[18 + k * 2 + 1] = t.uv[k][1];
    }
    std::memcpy(&f[24], &t.mat, 4);
    std::memcpy(o, f, 100);
}
inline float bitsToFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t floatToBits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

} // namespace

// =================================================================== C entry points
extern "C" {

void *ora_sd_create(void) { return new SdScene(); }
void ora_sd_destroy(void *s) { delete (SdScene *)s; }

int ora_sd_add_model(void *s, const float *tris24, const uint32_t *mesh_sizes, uint32_t n_meshes) {
    SdScene &sc = *(SdScene *)s;
    if (n_meshes == 0)
        return -1;
    std::vector<uint32_t> meshRoots; // PostProcess, Include/ModelLoader.hpp:238-251
    size_t off = 0;
    for (uint32_t m = 0; m < n_meshes; m++) {
        if (mesh_sizes[m] == 0)
            return -1;
        meshRoots.push_back(sc.addMesh(tris24 + off * 24, mesh_sizes[m]));
        off += mesh_sizes[m];
    }
    uint32_t root = sc.build(meshRoots.data(), 0, meshRoots.size()); // ModelLoader.hpp:271
    sc.sceneIndices.push_back(root);                                  // Program/Scene.cpp:203
    return 0;
}
int ora_sd_finalize(void *s) {
    SdScene &sc = *(SdScene *)s;
    if (sc.sceneIndices.empty())
        return -1; // "Build Failed. end - start <= 0", SimplifiedData.cpp:325-327
    std::vector<uint32_t> idx = sc.sceneIndices;
    sc.root = sc.build(idx.data(), 0, idx.size()); // Program/Scene.cpp:205-206
    sc.sceneIndices = idx;
    sc.orderLeaves();
    return 0;
}

uint64_t ora_take_leaf_replays(void) { return g_leaf_replays.exchange(0); }
void ora_take_limit_counts(uint64_t *skipped, uint64_t *met) {
    *skipped = g_leaves_skipped.exchange(0);
    *met = g_leaves_met.exchange(0);
}
int ora_sd_set_limit(void *s, const float *c9, const uint8_t *class_by_triangle, uint32_t n_triangles) {
    SdScene &sc = *(SdScene *)s;
    if (n_triangles != sc.tris.size())
        return -1;
    sc.limitClass.assign(class_by_triangle, class_by_triangle + n_triangles);
    for (int k = 0; k < 3; k++) {
        sc.limitC1[k] = c9[k];
        sc.limitC2x2[k] = c9[3 + k];
        sc.limitC3[k] = c9[6 + k];
    }
    return 0;
}
void ora_sd_counts(void *s, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *root) {
    SdScene &sc = *(SdScene *)s;
    *n_nodes = (uint32_t)sc.nodes.size();
    *n_tris = (uint32_t)sc.tris.size();
    *root = sc.root;
}
void ora_sd_get_nodes(void *s, void *out36) {
    SdScene &sc = *(SdScene *)s;
    for (size_t i = 0; i < sc.nodes.size(); i++)
        packNode36(sc.nodes[i], (uint8_t *)out36 + i * 36);
}
void ora_sd_get_tris(void *s, void *out100) {
    SdScene &sc = *(SdScene *)s;
    for (size_t i = 0; i < sc.tris.size(); i++)
        packTri100(sc.tris[i], (uint8_t *)out100 + i * 100);
}
// ConvertNodeToFlatStorage, SimplifiedData.cpp:162-185
void ora_sd_flat_nodes(void *s, float *out9) {
    SdScene &sc = *(SdScene *)s;
    for (size_t i = 0; i < sc.nodes.size(); i++) {
        const SdNode &n = sc.nodes[i];
        float *d = out9 + i * 9;
        d[0] = bitsToFloat(n.left); d[1] = bitsToFloat(n.right);
        d[2] = n.pmin.x; d[3] = n.pmin.y; d[4] = n.pmin.z;
        d[5] = n.pmax.x; d[6] = n.pmax.y; d[7] = n.pmax.z;
        d[8] = bitsToFloat(n.flags);
    }
}
// ConvertTriangleToFlatStorage, SimplifiedData.cpp:186-217
void ora_sd_flat_tris(void *s, float *out25) {
    SdScene &sc = *(SdScene *)s;
    for (size_t i = 0; i < sc.tris.size(); i++) {
        const SdTri &t = sc.tris[i];
        float *d = out25 + i * 25;
        for (int v = 0; v < 3; v++) {
            d[v * 8 + 0] = t.p[v].x; d[v * 8 + 1] = t.p[v].y; d[v * 8 + 2] = t.p[v].z;
            d[v * 8 + 3] = t.n[v].x; d[v * 8 + 4] = t.n[v].y; d[v * 8 + 5] = t.n[v].z;
            d[v * 8 + 6] = t.uv[v][0]; d[v * 8 + 7] = t.uv[v][1];
        }
        d[24] = bitsToFloat(t.mat);
    }
}
// GetNodeFromFlatStorage, SimplifiedData.cpp:224-237
void ora_flat_node_unpack(const float *flat9, uint32_t index, void *out36) {
    const float *b = flat9 + (size_t)index * 9;
    SdNode n;
    n.left = floatToBits(b[0]); n.right = floatToBits(b[1]);
    n.pmin = V3{b[2], b[3], b[4]}; n.pmax = V3{b[5], b[6], b[7]};
    n.flags = floatToBits(b[8]);
    packNode36(n, (uint8_t *)out36);
}
// GetTriangleFromFlatStorage, SimplifiedData.cpp:238-254.  Line 252 converts the bit-punned
// material word by VALUE (static_cast<uint16_t>(float)), so any small bit pattern reads back as 0.
void ora_flat_tri_unpack(const float *flat25, uint32_t index, void *out100) {
    const float *b = flat25 + (size_t)index * 25;
    SdTri t;
    for (int v = 0; v < 3; v++) {
        t.p[v] = V3{b[v * 8 + 0], b[v * 8 + 1], b[v * 8 + 2]};
        t.n[v] = V3{b[v * 8 + 3], b[v * 8 + 4], b[v * 8 + 5]};
        t.uv[v][0] = b[v * 8 + 6]; t.uv[v][1] = b[v * 8 + 7];
    }
    float m = b[24];
    t.mat = (m >= 0.0f && m < 65536.0f) ? (uint32_t)static_cast<uint16_t>(m) : 0u;
    packTri100(t, (uint8_t *)out100);
}

void ora_sd_intersect(void *s, const float *o, const float *d, int traversal, ora_hit *out) {
    SdScene &sc = *(SdScene *)s;
    Ray ray(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]});
    Counters c;
    SdHit h = sc.intersect(ray, traversal, c);
    out->hit = h.hit ? 1 : 0;
    out->t = h.t;
    out->pos[0] = h.pos.x; out->pos[1] = h.pos.y; out->pos[2] = h.pos.z;
    out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
    out->prim = h.prim; out->mat = h.mat;
    out->box_tests = (uint32_t)c.box; out->tri_tests = (uint32_t)c.tri;
}

int ora_sd_render(void *s, const ora_camera *cam, const ora_options *opt, int W, int H, int first, int count,
                  float *image, uint32_t *primary, ora_stats *stats) {
    SdScene &sc = *(SdScene *)s;
    if (sc.root == 0xFFFFFFFFu || W <= 0 || H <= 0 || first < 1 || count < 0)
        return -1; // "Scene is not loaded.", TraceMethods.cpp:54-56
    ora_options o = *opt;
    renderLoop(*cam, o, W, H, first, count, image, primary, stats,
               [&](const Ray &ray, uint32_t key, Counters &c, uint32_t *prim) { return castRaySd(sc, ray, o, key, c, prim); });
    return 0;
}

void *ora_old_create(void) { return new OldScene(); }
void ora_old_destroy(void *s) { delete (OldScene *)s; }

int ora_old_add_sphere(void *s, const float *center, float radius, uint32_t kind, const float *albedo4, float param) {
    OldScene &sc = *(OldScene *)s;
    auto sp = std::make_unique<Sphere>();
    sp->center = V3{center[0], center[1], center[2]};
    sp->radius = radius;
    sp->mat = Material{kind, V4{albedo4[0], albedo4[1], albedo4[2], albedo4[3]}, param};
    sp->bmin = sp->center - v3(radius); // Objects.hpp:40-41
    sp->bmax = sp->center + v3(radius);
    sp->id = 0x40000000u | sc.nSpheres++;
    sc.objects.push_back(sp.get());
    sc.owned.push_back(std::move(sp));
    return 0;
}
int ora_old_add_mesh(void *s, const float *tris24, uint32_t n_tris, uint32_t kind, const float *albedo4, float param) {
    OldScene &sc = *(OldScene *)s;
    if (n_tris == 0)
        return -1;
    auto mesh = std::make_unique<OldMesh>();
    mesh->mat = Material{kind, V4{albedo4[0], albedo4[1], albedo4[2], albedo4[3]}, param};
    std::vector<Hittable *> ptrs;
    for (uint32_t i = 0; i < n_tris; i++) {
        const float *f = tris24 + (size_t)i * 24;
        auto t = std::make_unique<OldTriangle>();
        for (int k = 0; k < 3; k++) {
            t->p[k] = V3{f[k * 8 + 0], f[k * 8 + 1], f[k * 8 + 2]};
            t->n[k] = V3{f[k * 8 + 3], f[k * 8 + 4], f[k * 8 + 5]};
        }
        t->mat = mesh->mat;
        t->bmin = vmin(t->p[0], vmin(t->p[1], t->p[2])) - v3(1e-6f); // Mesh.hpp:32-36
        t->bmax = vmax(t->p[0], vmax(t->p[1], t->p[2])) + v3(1e-6f);
        t->id = sc.nTris++;
        ptrs.push_back(t.get());
        mesh->tris.push_back(std::move(t));
    }
    mesh->bmin = mesh->tris[0]->bmin; // Mesh.hpp:111-119
    mesh->bmax = mesh->tris[0]->bmax;
    for (size_t i = 1; i < mesh->tris.size(); i++) {
        mesh->bmin = vmin(mesh->bmin, mesh->tris[i]->bmin);
        mesh->bmax = vmax(mesh->bmax, mesh->tris[i]->bmax);
    }
    mesh->inside.root = mesh->inside.build(ptrs, 0, (int)ptrs.size()); // Mesh.hpp:121-132
    sc.objects.push_back(mesh.get());
    sc.owned.push_back(std::move(mesh));
    return 0;
}
int ora_old_build(void *s) {
    OldScene &sc = *(OldScene *)s;
    sc.bvh.pool.clear();
    sc.bvh.root = sc.bvh.build(sc.objects, 0, (int)sc.objects.size()); // Scene::update, Scene.hpp:69-72
    return 0;
}
int ora_old_render(void *s, const ora_camera *cam, const ora_options *opt, int W, int H, int first, int count,
                   float *image, uint32_t *primary, ora_stats *stats) {
    OldScene &sc = *(OldScene *)s;
    if (W <= 0 || H <= 0 || first < 1 || count < 0)
        return -1;
    ora_options o = *opt;
    if (o.accum == ORA_ACCUM_FORWARD)
        renderLoop(*cam, o, W, H, first, count, image, primary, stats,
                   [&](const Ray &ray, uint32_t key, Counters &c, uint32_t *prim) { return castRayOldFwd(sc, ray, o, key, c, prim); });
    else
        renderLoop(*cam, o, W, H, first, count, image, primary, stats,
                   [&](const Ray &ray, uint32_t key, Counters &c, uint32_t *prim) { return castRayOldRec(sc, ray, 0, o, key, c, prim); });
    return 0;
}

void ora_old_intersect(void *s, const float *o, const float *d, int traversal, ora_hit *out) {
    OldScene &sc = *(OldScene *)s;
    Ray ray(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]});
    Counters c;
    OldHit h = sc.intersectClosest(ray, traversal, c);
    out->hit = h.t != kInf ? 1 : 0;
    out->t = h.t;
    out->pos[0] = h.pos.x; out->pos[1] = h.pos.y; out->pos[2] = h.pos.z;
    out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
    out->prim = h.prim; out->mat = h.mat ? h.mat->kind : 0;
    out->box_tests = (uint32_t)c.box; out->tri_tests = (uint32_t)(c.tri + c.sph);
}
static void flattenOld(const OldNode *n, std::vector<float> &out) {
    if (!n)
        return;
    uint32_t kind = 0, ref = 0;
    const OldMesh *mesh = nullptr;
    if (n->object) {
        if (dynamic_cast<const Sphere *>(n->object)) { kind = 4; ref = n->object->id & 0x3FFFFFFFu; }
        else if ((mesh = dynamic_cast<const OldMesh *>(n->object))) { kind = 2; }
        else { kind = 1; ref = n->object->id; }
    }
    float rec[8] = {n->bmin.x, n->bmin.y, n->bmin.z, n->bmax.x, n->bmax.y, n->bmax.z, bitsToFloat(kind), bitsToFloat(ref)};
    out.insert(out.end(), rec, rec + 8);
    if (mesh)
        flattenOld(mesh->inside.root, out);
    flattenOld(n->left, out);
    flattenOld(n->right, out);
}
uint32_t ora_old_flatten(void *s, float *out8) {
    OldScene &sc = *(OldScene *)s;
    std::vector<float> v;
    flattenOld(sc.bvh.root, v);
    if (out8)
        std::memcpy(out8, v.data(), v.size() * sizeof(float));
    return (uint32_t)(v.size() / 8);
}

int ora_box_test(const float *pmin, const float *pmax, const float *o, const float *d, float tmin, float tmax) {
    Ray ray(V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]});
    return box_test(V3{pmin[0], pmin[1], pmin[2]}, V3{pmax[0], pmax[1], pmax[2]}, ray, tmin, tmax) ? 1 : 0;
}
uint32_t ora_rng_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    return rng_bits(rng_key(seed, pixel, sample), dim);
}
void ora_sincos(float theta, int trig, float *s, float *c) { sincos_eval(theta, trig, s, c); }
void ora_camera_dir(const ora_camera *cam, float u, float v, float *out3) {
    V3 d = camera_dir(*cam, u, v);
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}
// The vector helpers themselves, for known-answer tests of their operation order: op 0 dot(a, b) -> out[0]; 1 cross(a, b);
// 2 normalize(a); 3 mat3(a, b, c) * v; 4 component-wise min(a, b); 5 component-wise max(a, b).
void ora_vec_op(int op, const float *a, const float *b, const float *c, const float *v, float *out3) {
    const V3 A{a[0], a[1], a[2]}, B = b ? V3{b[0], b[1], b[2]} : V3{0, 0, 0}, C = c ? V3{c[0], c[1], c[2]} : V3{0, 0, 0},
             X = v ? V3{v[0], v[1], v[2]} : V3{0, 0, 0};
    V3 r{0, 0, 0};
    switch (op) {
    case 0: r.x = dot(A, B); break;
    case 1: r = cross(A, B); break;
    case 2: r = normalize(A); break;
    case 3: r = mat3_mul(A, B, C, X); break;
    case 4: r = vmin(A, B); break;
    default: r = vmax(A, B); break;
    }
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void ora_cosine_dir(const float *n, float u1, float u2, int trig, float *out3) {
    V3 d = cosine_dir(V3{n[0], n[1], n[2]}, u1, u2, trig);
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

} // extern "C"

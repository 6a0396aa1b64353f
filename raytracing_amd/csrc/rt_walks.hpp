// rt_walks.hpp: closest-hit searches (reference-order lane / wave walks, brute force, the fast walk and its ordered replay) -- part of rt_hip.hip (included there, inside its unnamed namespace); see that file for the overall design.
#pragma once

struct Best {
    float t, u, v;
    uint32_t ref; // triangle slot, kSphereBit | sphere index, or kNone
};
struct Cnt {
    unsigned long long rays, box, tri, sph, trih, replay;
};

// Moller-Trumbore exactly as sd::IntersectTriangle (SimplifiedData.cpp:269-307) and Triangle::intersect (Mesh.hpp:39-77)
// evaluate it, on the stored p0, e1 = p1 - p0, e2 = p2 - p0: true when the test reports a hit (t > EPSILON; the caller
// applies its own "closer than" rule), with the hit's t, u, v.
RT_D bool tri_test(const f3 o, const f3 d, const f3 p0, const f3 e1, const f3 e2, float &t, float &u, float &v) {
    const float EPSILON = 1e-7f;
    const f3 h = cross3(d, e2);
    const float det = dot3(e1, h);
    if (det > -EPSILON && det < EPSILON)
        return false;
    const float f = 1.0f / det;
    const f3 s = o - p0;
    u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f)
        return false;
    const f3 q = cross3(s, e1);
    v = f * dot3(d, q);
    if (v < 0.0f || u + v > 1.0f)
        return false;
    t = f * dot3(e2, q);
    return t > EPSILON;
}

// Sphere::intersect (Objects.hpp:43-72): the near root, and a ray that starts inside never hits -- except for the
// dielectric spheres this build adds (`dielectric`), which need the exit point.  True with t when the test reports a hit.
RT_D bool sphere_test(const f3 o, const f3 d, const float4 s0, const bool dielectric, float &t) {
    const f3 oc = mk3(s0.x, s0.y, s0.z) - o;
    const float qa = dot3(d, d);
    const float qb = -2.0f * dot3(d, oc);
    const float qc = dot3(oc, oc) - s0.w * s0.w;
    const float disc = qb * qb - 4 * qa * qc;
    if (disc < 0)
        return false;
    t = (-qb - sqrtf(disc)) / (2.0f * qa);
    bool ok = !(t < 0);
    if (!ok && dielectric) {
        t = (-qb + sqrtf(disc)) / (2.0f * qa);
        ok = !(t < 0);
    }
    return ok;
}

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
            float t, u, v;
            if (tri_test(o, d, p0, e1, e2, t, u, v) && t < best.t) {
                best.t = t;
                best.u = u;
                best.v = v;
                best.ref = slot;
            }
        } else if (OLD && k == KIND_SPHERE) { // Sphere::intersect, Objects.hpp:43-72
            if (STATS)
                cnt.sph++;
            const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
            float t;
            if (sphere_test(o, d, s0, __float_as_uint(s1.z) != 0u, t) && t < best.t) {
                best.t = t;
                best.ref = kSphereBit | slot;
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
            float t, u, v;
            if (tri_test(o, d, p0, e1, e2, t, u, v)) {
                const uint32_t ti = __float_as_uint(q2.y);
                const bool in_order_group = !OLD || inner_end != kNone; // triangles scanned in storage order
                if (t < best.t || (in_order_group && t == best.t && t < __builtin_inff() && ti < bidx)) {
                    best = Best{t, u, v, slot};
                    bidx = ti;
                }
            }
        } else if (OLD && k == KIND_SPHERE) { // Sphere::intersect, Objects.hpp:43-72
            if (STATS)
                cnt.sph++;
            const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
            float t;
            if (sphere_test(o, d, s0, __float_as_uint(s1.z) != 0u, t) && t < best.t) {
                best = Best{t, 0.0f, 0.0f, kSphereBit | slot};
                bidx = kNone;
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
RT_D float4 mk4(const v4f v) { return make_float4(v.x, v.y, v.z, v.w); }

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
                float t, u, v;
                if (tri_test(o, d, p0, e1, e2, t, u, v) && t < best.t) {
                    best.t = t;
                    best.u = u;
                    best.v = v;
                    best.ref = slot;
                }
            }
        } else if (OLD && k == KIND_SPHERE) {
            const v4f s0 = spheres[2 * slot], s1 = spheres[2 * slot + 1];
            if (pass) {
                if (STATS)
                    cnt.sph++;
                float t;
                if (sphere_test(o, d, mk4(s0), __float_as_uint(s1.z) != 0u, t) && t < best.t) {
                    best.t = t;
                    best.ref = kSphereBit | slot;
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
    int slice;          // BURST < 0: records this call may take the walk further
};

// BURST = 0: the whole walk.  BURST > 0: at most that many records (BURST < 0: at most w.slice), resuming from and saving to `w` (the loop itself
// runs on local copies either way).
// `note(is_sphere, slot, entry distance)` may take a primitive test off the walk's hands (it returns true and the caller
// runs the test later, see WaveJobs); the default takes none.
struct NotesNothing {
    RT_D bool operator()(uint32_t, uint32_t, float) const { return false; }
};
template <bool OLD, bool STATS, int BURST, class F, class N = NotesNothing>
RT_D void fast_walk_from(const KParams &P, const f3 o, const f3 d, const f3 inv, Walk &w, Cnt &cnt, F &&consider, N &&note = N()) {
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const uint32_t n = P.n_fnodes * 32u;
    uint32_t idx = w.idx;
    uint32_t mesh = w.mesh, inner_end = w.inner_end;
    float mesh_m = w.mesh_m;
    int budget = BURST > 0 ? BURST : w.slice;
    while (idx < n) {
        if (BURST != 0) {
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
            if (note(0u, slot, OLD ? fmaxf(tmin, mesh_m) : tmin))
                continue;
            const float4 q0 = P.tri_pos[3 * slot], q1 = P.tri_pos[3 * slot + 1], q2 = P.tri_pos[3 * slot + 2];
            // All nine floats of the triangle are fetched before the test starts: left alone, the compiler sinks the
            // load of p0 behind the determinant test, which puts a second memory round trip into most triangle tests.
            asm volatile("" : : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "v"(q2.x));
            const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
            float t, u, v;
            if (tri_test(o, d, p0, e1, e2, t, u, v))
                consider(Cand{t, u, v, tmin, slot, __float_as_uint(q2.w), mesh, mesh_m});
        } else if (OLD && k == KIND_SPHERE) {
            if (STATS)
                cnt.sph++;
            if (note(1u, slot, fmaxf(tmin, 0.0f)))
                continue;
            const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
            float t;
            if (sphere_test(o, d, s0, __float_as_uint(s1.z) != 0u, t))
                consider(Cand{t, 0.0f, 0.0f, tmin, kSphereBit | slot, __float_as_uint(s1.w), kNone, 0.0f});
        } else if (OLD && k == KIND_MESH) { // the mesh's inner hierarchy follows inline, up to `skip`
            mesh = slot;
            mesh_m = tmin;
            inner_end = skip;
        }
    }
    if (BURST != 0) {
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

// closest_hit_fast for a whole wave, with the triangle tests taken OUT of the walk.
//
// In the plain walk a lane that stands on a leaf runs the triangle test (four loads, ~60 instructions) while the other
// lanes of its wave wait, and with dozens of lanes walking some lane stands on a leaf in most steps: a third of a bounce
// level's time goes into triangle tests that run at a few lanes' worth of use.  Here a lane that passes a leaf box only
// notes a job -- (its lane, the triangle slot, the box's entry distance) -- in a list the wave keeps in LDS, and walks on.
// When all lanes have finished their walks the wave runs the noted tests 64 at a time, every lane taking one job whoever
// noted it: the job's ray comes from its owner's registers (ds_bpermute), a hit goes into the owner's 64-bit LDS word
// min((t bits << 32) | position) -- t > 0, so the order of the bits is the order of the values -- and the job that holds
// the minimum afterwards leaves u, v, slot and entry distance for the owner.  The hit kept is the smallest (t, position),
// as in closest_hit_fast, so the result is the same.  A job that does not fit the list (or a slot beyond 26 bits) is
// tested on the spot.  Every lane of the wave must call this (lanes without a ray pass live = false and help testing).
constexpr uint32_t kJobCap = 512;       // jobs a wave can note per closest-hit search of its 64 rays
constexpr uint32_t kJobSlotBits = 26;     // a job's first word: owner lane (6 bits), sphere flag (bit 25), slot (25 bits)
constexpr uint32_t kJobSphere = 1u << 25;

struct WaveJobs {            // LDS of one wave
    static constexpr uint32_t kCap = kJobCap;
    uint2 jobs[kJobCap];     // (owner lane << 26) | slot, entry distance bits
    unsigned long long key[64];
    float4 win[64];          // u, v, hit reference (Best::ref), entry distance
    float win_t[64];         // the hit distance itself (the key holds +0 for -0)
    uint32_t count;
    RT_D void put(const uint32_t j, const uint32_t word) { jobs[j] = make_uint2(word, 0u); }
    RT_D uint32_t word(const uint32_t j) const { return jobs[j].x; }
};
// The same for the wide walk, whose jobs carry no entry distance (the exact leaf test finds it) and come in smaller
// numbers per look at the list: a quarter of the LDS, so that twice as many waves fit a compute unit.
constexpr uint32_t kWideJobCap = 256;
struct WideJobs {
    static constexpr uint32_t kCap = kWideJobCap;
    uint32_t jobs[kWideJobCap]; // (owner lane << 26) | slot
    unsigned long long key[64];
    float4 win[64];
    float win_t[64];
    uint32_t count;
    RT_D void put(const uint32_t j, const uint32_t word) { jobs[j] = word; }
    RT_D uint32_t word(const uint32_t j) const { return jobs[j]; }
};

// Runs the tests a wave has noted, 64 at a time, whoever noted them, and leaves each owner's best in J.key / J.win /
// J.win_t.  `o`, `d`: this lane's own ray (a job's ray is fetched from its owner's registers).  Every lane of the wave
// must call this; the list is empty afterwards.
template <bool OLD>
RT_D void resolve_jobs(const KParams &P, const uint32_t lane, const f3 o, const f3 d, WaveJobs &J) {
    const float inf = __builtin_inff();
    __builtin_amdgcn_wave_barrier();
    const uint32_t noted = J.count;
    const uint32_t n_jobs = noted < kJobCap ? noted : kJobCap;
    for (uint32_t base = 0; base < n_jobs; base += 64u) {
        const uint32_t j = base + lane;
        const bool valid = j < n_jobs;
        const uint2 job = valid ? J.jobs[j] : make_uint2(0u, 0u);
        const uint32_t owner = job.x >> kJobSlotBits, slot = job.x & (kJobSphere - 1u);
        const bool sphere = OLD && (job.x & kJobSphere) != 0u;
        const int sel = (int)(owner << 2);
        const f3 ro = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.y))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.z))));
        const f3 rd = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.y))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.z))));
        bool hit = false;
        unsigned long long key = ~0ull;
        float hu = 0.0f, hv = 0.0f, ht = 0.0f;
        uint32_t href = slot;
        if (valid && sphere) {
            const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
            float t;
            if (sphere_test(ro, rd, s0, __float_as_uint(s1.z) != 0u, t) && t < inf) { // t >= 0 here (NaN fails "t < inf"); -0 takes the key of +0, which it equals
                hit = true;
                key = ((unsigned long long)(t == 0.0f ? 0u : __float_as_uint(t)) << 32) | (unsigned long long)__float_as_uint(s1.w);
                ht = t;
                href = kSphereBit | slot;
            }
        } else if (valid) {
            const float4 q0 = P.tri_pos[3 * slot], q1 = P.tri_pos[3 * slot + 1], q2 = P.tri_pos[3 * slot + 2];
            // the whole record before the test starts, as in fast_walk_from (q2.w, the position, is needed by every hit)
            asm volatile("" : : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "v"(q2.x), "v"(q2.w));
            const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
            float t, u, v;
            // t = +inf passes "t > EPSILON" but never the reference's "t < closest.t"
            if (tri_test(ro, rd, p0, e1, e2, t, u, v) && t < inf) {
                hit = true;
                key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)__float_as_uint(q2.w);
                hu = u;
                hv = v;
                ht = t;
            }
        }
        if (hit)
            atomicMin(&J.key[owner], key);
        __builtin_amdgcn_wave_barrier();
        if (hit) {
            if (J.key[owner] == key) {
                J.win[owner] = make_float4(hu, hv, __uint_as_float(href), __uint_as_float(job.y));
                J.win_t[owner] = ht;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0)
        J.count = 0u;
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// The compact walk (flat scenes).  The fast walk's interior boxes only have to be conservative -- what it returns is
// decided by the leaves that pass their own exact test -- so the records it steps through can be half the size: six
// 16-bit grid planes that enclose the node's box with a grid unit to spare, and one word (KParams::cnodes).  A plane's
// distance along the ray is one fused multiply-add, t = q * (step * inv) + (origin - o) * inv; its rounding error stays
// below 0.4 grid units for a ray that starts within 32 grid extents of the grid's origin -- and the reference's own slab
// test, which rounds too, stays within another 0.3 -- (checked per call for the
// camera; bounce rays start on the geometry), so a record whose exact box the ray meets always passes.  A LEAF record
// that passes is only noted; its exact box test (the reference's leaf box, bit for bit: it yields the entry distance the
// acceptance rule needs) runs with its triangle test, 64 jobs at a time, from one 64-byte record that holds both
// (KParams::leaf_boxes).  One 16-byte load per step instead of two.
struct GridRay {
    f3 A, B; // t = q * A + B
};
// One axis.  A direction component of (nearly) zero makes 1 / d infinite (or so large that the products overflow), and
// infinity times a plane's number minus infinity is not a number: the axis would stop pruning anything, and a ray along a
// coordinate axis -- straight up from the ground, say -- would visit every record of the tree.  What the slab test means
// there is "is the origin between the two planes": t = (q - q_o) * (+-1e30), q_o the origin in grid units, says just that,
// and dwarfs every distance the other axes (|d| = 1: one of them has |1 / d| < 1.8) can produce.
RT_D void grid_axis(const float step, const float origin, const float o, const float inv, float &A, float &B) {
    if (__builtin_fabsf(inv) > 1e25f) {
        A = __builtin_copysignf(1e30f, inv);
        B = -((o - origin) / step) * A;
    } else {
        A = step * inv;
        B = (origin - o) * inv;
    }
}
RT_D GridRay grid_ray(const KParams &P, const f3 o, const f3 inv) {
    GridRay g;
    grid_axis(P.q_step.x, P.q_origin.x, o.x, inv.x, g.A.x, g.B.x);
    grid_axis(P.q_step.y, P.q_origin.y, o.y, inv.y, g.A.y, g.B.y);
    grid_axis(P.q_step.z, P.q_origin.z, o.z, inv.z, g.A.z, g.B.z);
    return g;
}

// The reference's slab test of a leaf box with closest = +inf (SimplifiedData.cpp:146-161 as the fast walk evaluates it):
// true when the ray meets the box, with the entry distance max(near planes, 1e-6).
RT_D bool exact_leaf_entry(const float4 q2, const float4 q3, const f3 o, const f3 inv, float &m) {
    const float4 a = make_float4(q2.y, q2.z, q3.x, q3.y), b = make_float4(q3.z, q3.w, 0.0f, 0.0f);
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const float t0x = (a.x - o.x) * inv.x, t1x = (a.w - o.x) * inv.x;
    const float t0y = (a.y - o.y) * inv.y, t1y = (b.x - o.y) * inv.y;
    const float t0z = (a.z - o.z) * inv.z, t1z = (b.y - o.z) * inv.z;
    const float nearx = nx ? t1x : t0x, farx = nx ? t0x : t1x;
    const float neary = ny ? t1y : t0y, fary = ny ? t0y : t1y;
    const float nearz = nz ? t1z : t0z, farz = nz ? t0z : t1z;
    m = fmaxf(nearz, fmaxf(neary, fmaxf(nearx, 1e-6f)));
    const float tmax = fminf(farz, fminf(fary, fminf(farx, __builtin_inff())));
    return !(tmax <= m);
}

// Takes the walk about `steps` records further from byte offset `idx` of the compact array; `note(slot)` is called for
// every leaf record that passes.  PAIR (scenes beyond the L2s, where a trip to memory is long; KParams::compact == 2): every trip
// brings the record the walk stands on AND the one behind it (the same 128-byte line seven times out of eight): after a
// record that does not prune -- an interior box that is met, or any leaf -- the walk's next record is that one, and it is
// tested at once; only a pruned subtree costs a trip for a single record (measured: 6 to 10 % faster on the 101 and 133 MB
// scenes, 2 % slower on scenes within the L2s).  The array ends with one unused record, so the second load never leaves it.
// One record of the compact walk: the slab test on its grid planes; true when the ray misses the box.
RT_D bool grid_box_missed(const uint4 r, const GridRay &g, const bool nx, const bool ny, const bool nz) {
    const float t0x = __builtin_fmaf((float)(r.x & 0xFFFFu), g.A.x, g.B.x), t1x = __builtin_fmaf((float)(r.x >> 16), g.A.x, g.B.x);
    const float t0y = __builtin_fmaf((float)(r.y & 0xFFFFu), g.A.y, g.B.y), t1y = __builtin_fmaf((float)(r.y >> 16), g.A.y, g.B.y);
    const float t0z = __builtin_fmaf((float)(r.z & 0xFFFFu), g.A.z, g.B.z), t1z = __builtin_fmaf((float)(r.z >> 16), g.A.z, g.B.z);
    const float nearx = nx ? t1x : t0x, farx = nx ? t0x : t1x;
    const float neary = ny ? t1y : t0y, fary = ny ? t0y : t1y;
    const float nearz = nz ? t1z : t0z, farz = nz ? t0z : t1z;
    const float tmin = fmaxf(nearz, fmaxf(neary, fmaxf(nearx, 1e-6f)));
    const float tmax = fminf(farz, fminf(fary, fminf(farx, __builtin_inff())));
    return tmax <= tmin;
}

// A leaf that passes is noted in two halves: `issue()` reserves a place in the wave's list (an LDS atomic whose answer
// takes a hundred cycles), `commit(place, slot)` writes the job there.  The commit waits until the walk's NEXT record has
// been asked for, so the atomic's latency runs under the memory latency every step pays anyway instead of adding to it
// (nearly every step of a wave has some lane on a leaf).
template <bool STATS, bool PAIR, class I, class C>
RT_D void compact_walk_from(const KParams &P, const f3 inv, const GridRay g, uint32_t &idx, int steps, Cnt &cnt, I &&issue, C &&commit) {
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const uint32_t n = P.n_fnodes * 16u;
    uint32_t at = idx;
    uint32_t place = 0u, pending = kNone; // the leaf slot whose place has been asked for, kNone: none
    if constexpr (!PAIR) {
        while (at < n) {
            if (steps-- == 0)
                break;
            const uint4 r = *(const uint4 *)((const char *)P.cnodes + at);
            if (pending != kNone) {
                commit(place, pending);
                pending = kNone;
            }
            if (STATS)
                cnt.box++;
            const bool miss = grid_box_missed(r, g, nx, ny, nz), leaf = (r.w & 1u) != 0u;
            if (leaf && !miss) {
                place = issue();
                pending = r.w >> 4;
            }
            at = (miss && !leaf) ? r.w : at + 16u;
        }
    } else {
        while (at < n) {
            if (steps <= 0)
                break;
            steps -= 2;
            uint4 r = *(const uint4 *)((const char *)P.cnodes + at);
            const uint4 r1 = *(const uint4 *)((const char *)P.cnodes + at + 16u);
            if (pending != kNone) {
                commit(place, pending);
                pending = kNone;
            }
#pragma unroll 1
            for (int k = 0; k < 2; k++) {
                if (STATS)
                    cnt.box++;
                const bool miss = grid_box_missed(r, g, nx, ny, nz), leaf = (r.w & 1u) != 0u;
                if (leaf && !miss) {
                    if (pending != kNone)
                        commit(place, pending);
                    place = issue();
                    pending = r.w >> 4;
                }
                if (miss && !leaf) {
                    at = r.w;
                    break;
                }
                at += 16u;
                if (!(at < n))
                    break;
                r = r1;
            }
        }
    }
    if (pending != kNone)
        commit(place, pending);
    idx = at;
}

// ------------------------------------------------------------------------------------------------
// The wide walk (flat scenes; KParams::wnodes, laid out by build_wide).  A step is ONE 64-byte record: the boxes of up
// to eight children on the compact tree's grid, 8 bits per plane relative to the node's own origin and scale.  Like the
// compact walk it only has to be conservative -- the leaves that pass are noted and run through the exact leaf test and
// the triangle test by resolve_jobs_compact -- and it visits the same leaves in some other order, which the result does
// not depend on (closest_hit_fast).  A plane's distance along the ray is evaluated in two steps,
//     t = p * (A * 2^shift) + (org * A + B),        A, B: the ray on the grid (GridRay), t = q * A + B for grid unit q,
// the second rounding being what the planes' second grid unit of spare room is for.
// The walk is depth first with a stack of one entry per level: the interior children of a node that the ray meets and
// that are not taken at once, as (index of the node's first interior child) << 9 | (order bit) << 8 | (8-bit set of
// children).  A node's children are sorted along one axis; a ray that runs up that axis takes them from the front of
// the set, one that runs down from the back, so near children tend to come first.  The stack lives in LDS, one word
// per lane and level (`stk`, the lane's column: level l at stk[64 * l]).
constexpr uint32_t kWideDone = 0xFFFFFFFFu; // value of the walk's position once the walk is over

RT_D float ubyte_f(const uint32_t w, const int i) { return (float)((w >> (8 * i)) & 0xFFu); }

// Takes the walk at most `steps` nodes further.  cur: node the lane stands on (kWideDone: the walk is over); sp: entries
// on the lane's stack.  `tau`: the smallest hit distance the ray knows of, +inf for none or for a ray that takes no
// distance limit; dmax, inv_dmax: the largest component of the ray's direction and its reciprocal.  The walk does not enter
// a child whose box the ray meets at or beyond the limit of the node's class (see LeafShape in scene_layout.hpp).
// `reserve(n)` hands out n consecutive places in the wave's list of noted tests, `commit(place, slot)` notes one.
template <bool STATS, class R, class C>
RT_D void wide_walk_from(const KParams &P, const GridRay g, uint32_t &cur, uint32_t &sp, uint32_t *stk, int steps, const float tau,
                         const float dmax, const float inv_dmax, Cnt &cnt, R &&reserve, C &&commit) {
    uint32_t at = cur, top = sp;
    // A leaf the ray meets is noted in two halves, as in compact_walk_from: the places in the wave's list are asked for (an
    // LDS atomic whose answer takes a hundred cycles) as soon as the node's tests are done, the jobs are written once the
    // NEXT node has been asked for, so that the atomic's latency runs under the memory latency every step pays anyway.
    uint32_t pend_leaves = 0u, pend_base = 0u, pend_place = 0u;
    auto flush = [&]() {
        while (pend_leaves != 0u) {
            const uint32_t b = (uint32_t)__builtin_ctz(pend_leaves);
            pend_leaves &= pend_leaves - 1u;
            commit(pend_place++, pend_base + b);
        }
    };
    while (at != kWideDone) {
        if (steps-- == 0)
            break;
        const uint4 *rec = P.wnodes + 4 * (size_t)at;
        const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        flush();
        // the node's frame on the grid
        const float ox = (float)(r3.x & 0xFFFFu), oy = (float)(r3.x >> 16), oz = (float)(r3.y & 0xFFFFu);
        const float ax = __builtin_ldexpf(g.A.x, (int)((r3.y >> 16) & 15u)), ay = __builtin_ldexpf(g.A.y, (int)((r3.y >> 20) & 15u)),
                    az = __builtin_ldexpf(g.A.z, (int)((r3.y >> 24) & 15u));
        const float bx = __builtin_fmaf(ox, g.A.x, g.B.x), by = __builtin_fmaf(oy, g.A.y, g.B.y), bz = __builtin_fmaf(oz, g.A.z, g.B.z);
        const uint32_t n_int = (r3.w >> 24) & 15u, n_all = r3.w >> 28;
        if (STATS)
            cnt.box += n_all;
        // near and far plane of each axis by the sign of the ray's slope on the grid: the byte arrays are picked once per
        // node, so a child costs six conversions, six multiply-adds and the two three-way extrema
        const bool dnx = g.A.x < 0.0f, dny = g.A.y < 0.0f, dnz = g.A.z < 0.0f;
        const uint32_t nx0 = dnx ? r0.z : r0.x, nx1 = dnx ? r0.w : r0.y, fx0 = dnx ? r0.x : r0.z, fx1 = dnx ? r0.y : r0.w;
        const uint32_t ny0 = dny ? r1.z : r1.x, ny1 = dny ? r1.w : r1.y, fy0 = dny ? r1.x : r1.z, fy1 = dny ? r1.y : r1.w;
        const uint32_t nz0 = dnz ? r2.z : r2.x, nz1 = dnz ? r2.w : r2.y, fz0 = dnz ? r2.x : r2.z, fz1 = dnz ? r2.y : r2.w;
        // the distance limit of this node's class: (tau + c1 / D + c2 W + c3) * (1 + 2 c2 D) * 1.00001; none for class 3
        const uint32_t cls = r3.y >> 30;
        const float k1 = cls == 0u ? P.prune_c1[0] : cls == 1u ? P.prune_c1[1] : P.prune_c1[2];
        const float k2 = cls == 0u ? P.prune_c2x2[0] : cls == 1u ? P.prune_c2x2[1] : P.prune_c2x2[2];
        const float k3 = cls == 0u ? P.prune_c3[0] : cls == 1u ? P.prune_c3[1] : P.prune_c3[2];
        const float here = cls == 3u ? __builtin_inff() : (tau + __builtin_fmaf(k1, inv_dmax, k3)) * __builtin_fmaf(k2, dmax, 1.00001f);
        uint32_t hits = 0u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float tnx = __builtin_fmaf(ubyte_f(i < 4 ? nx0 : nx1, i & 3), ax, bx), tfx = __builtin_fmaf(ubyte_f(i < 4 ? fx0 : fx1, i & 3), ax, bx);
            const float tny = __builtin_fmaf(ubyte_f(i < 4 ? ny0 : ny1, i & 3), ay, by), tfy = __builtin_fmaf(ubyte_f(i < 4 ? fy0 : fy1, i & 3), ay, by);
            const float tnz = __builtin_fmaf(ubyte_f(i < 4 ? nz0 : nz1, i & 3), az, bz), tfz = __builtin_fmaf(ubyte_f(i < 4 ? fz0 : fz1, i & 3), az, bz);
            const float tmin = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 1e-6f));
            const float tmax = fminf(fminf(tfx, tfy), fminf(tfz, here));
            hits |= (tmax > tmin ? 1u : 0u) << i;
        }
        hits &= (1u << n_all) - 1u;
        // leaves the ray meets: their places in the list asked for in one reservation, written a step later
        pend_leaves = hits >> n_int;
        if (pend_leaves != 0u) {
            pend_place = reserve((uint32_t)__builtin_popcount(pend_leaves));
            pend_base = r3.w & 0xFFFFFFu;
        }
        // interior children the ray meets: one now, the others on the stack
        const uint32_t inner = hits & ((1u << n_int) - 1u);
        const uint32_t axis = (r3.y >> 28) & 3u;
        const bool down = axis == 0u ? dnx : axis == 1u ? dny : dnz;
        uint32_t entry;
        if (inner != 0u) {
            entry = (r3.z << 9) | (down ? 256u : 0u) | inner;
        } else if (top != 0u) {
            top--;
            entry = stk[64u * top];
        } else {
            at = kWideDone;
            break;
        }
        const uint32_t set = entry & 255u;
        const uint32_t pick = (entry & 256u) != 0u ? 31u - (uint32_t)__builtin_clz(set) : (uint32_t)__builtin_ctz(set);
        const uint32_t rest = set & ~(1u << pick);
        if (rest != 0u) {
            stk[64u * top] = (entry & ~255u) | rest;
            top++;
        }
        at = (entry >> 9) + pick;
    }
    flush();
    cur = at;
    sp = top;
}

// resolve_jobs for the compact walk: a job is a leaf whose grid box the owner's ray met; it runs the leaf's exact box test
// and, if that passes, the triangle test.  `o`, `d`, `inv`: this lane's own ray.
template <bool STATS, class JT>
RT_D void resolve_jobs_compact(const KParams &P, const uint32_t lane, const f3 o, const f3 d, const f3 inv, JT &J, Cnt &cnt) {
    const float inf = __builtin_inff();
    __builtin_amdgcn_wave_barrier();
    const uint32_t noted = J.count;
    const uint32_t n_jobs = noted < JT::kCap ? noted : JT::kCap;
    for (uint32_t base = 0; base < n_jobs; base += 64u) {
        const uint32_t j = base + lane;
        const bool valid = j < n_jobs;
        const uint32_t jobw = valid ? J.word(j) : 0u;
        const uint32_t owner = jobw >> kJobSlotBits, slot = jobw & (kJobSphere - 1u);
        const int sel = (int)(owner << 2);
        const f3 ro = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.y))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(o.z))));
        const f3 rd = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.y))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(d.z))));
        const f3 ri = mk3(__int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.x))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.y))),
                          __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(inv.z))));
        bool hit = false;
        unsigned long long key = ~0ull;
        float hu = 0.0f, hv = 0.0f, ht = 0.0f, hm = 0.0f;
        if (valid) {
            // one 64-byte record holds the leaf's box and its triangle: a single round trip to memory per pass
            const float4 q0 = P.leaf_boxes[4 * slot], q1 = P.leaf_boxes[4 * slot + 1], q2 = P.leaf_boxes[4 * slot + 2], q3 = P.leaf_boxes[4 * slot + 3];
            if (STATS)
                cnt.box++;
            if (exact_leaf_entry(q2, q3, ro, ri, hm)) {
                if (STATS)
                    cnt.tri++;
                float t, u, v;
                // t = +inf passes "t > EPSILON" but never the reference's "t < closest.t"
                if (tri_test(ro, rd, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x), t, u, v) && t < inf) {
                    hit = true;
                    key = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)__float_as_uint(q2.w);
                    hu = u;
                    hv = v;
                    ht = t;
                }
            }
        }
        if (hit)
            atomicMin(&J.key[owner], key);
        __builtin_amdgcn_wave_barrier();
        if (hit) {
            if (J.key[owner] == key) {
                J.win[owner] = make_float4(hu, hv, __uint_as_float(slot), hm);
                J.win_t[owner] = ht;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0)
        J.count = 0u;
    __builtin_amdgcn_wave_barrier();
}

// Merges what resolve_jobs left for this lane into the lane's own best (which holds the tests it ran on the spot).
template <class JT>
RT_D void take_noted_result(const bool live, const uint32_t lane, JT &J, Best &best, uint32_t &bpos, float &bm) {
    const unsigned long long mine = J.key[lane];
    if (live && mine != ~0ull) {
        const float t = J.win_t[lane];
        const uint32_t pos = (uint32_t)mine;
        if (t < best.t || (t == best.t && pos < bpos)) {
            const float4 wv = J.win[lane];
            best = Best{t, wv.x, wv.y, __float_as_uint(wv.z)};
            bpos = pos;
            bm = wv.w;
        }
    }
}

template <bool OLD, bool STATS>
RT_D void closest_hit_fast_queued(const KParams &P, const bool live, const uint32_t lane, const f3 o, const f3 d, const f3 inv,
                                  Best &best, Cnt &cnt, WaveJobs &J) {
    const float inf = __builtin_inff();
    best = Best{inf, 0.0f, 0.0f, kNone};
    uint32_t bpos = kNone;
    float bm = 0.0f;
    J.key[lane] = ~0ull;
    if (lane == 0)
        J.count = 0u;
    // ---- the walk: box tests only; triangles are noted ----
    if (live) {
        const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
        const uint32_t n = P.n_fnodes * 32u;
        uint32_t idx = 0u;
        uint32_t inner_end = kNone;
        float mesh_m = 0.0f;
        while (idx < n) {
            if (OLD) {
                if (idx == inner_end) {
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
            const float tmax = fminf(farz, fminf(fary, fminf(farx, inf))); // the slab test with closest = +inf
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
                const float m = OLD ? fmaxf(tmin, mesh_m) : tmin;
                const uint32_t j = atomicAdd(&J.count, 1u);
                if (j < kJobCap && slot < kJobSphere) {
                    J.jobs[j] = make_uint2((lane << kJobSlotBits) | slot, __float_as_uint(m));
                } else { // no room: test it here
                    const float4 q0 = P.tri_pos[3 * slot], q1 = P.tri_pos[3 * slot + 1], q2 = P.tri_pos[3 * slot + 2];
                    const f3 p0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
                    float t, u, v;
                    const uint32_t pos = __float_as_uint(q2.w);
                    if (tri_test(o, d, p0, e1, e2, t, u, v) && (t < best.t || (t == best.t && pos < bpos)) && t < inf) {
                        best = Best{t, u, v, slot};
                        bpos = pos;
                        bm = m;
                    }
                }
            } else if (OLD && k == KIND_SPHERE) {
                if (STATS)
                    cnt.sph++;
                const uint32_t j = atomicAdd(&J.count, 1u);
                if (j < kJobCap && slot < kJobSphere) {
                    J.jobs[j] = make_uint2((lane << kJobSlotBits) | kJobSphere | slot, __float_as_uint(fmaxf(tmin, 0.0f)));
                } else { // no room: test it here
                    const float4 s0 = P.spheres[2 * slot], s1 = P.spheres[2 * slot + 1];
                    float t;
                    const uint32_t pos = __float_as_uint(s1.w);
                    if (sphere_test(o, d, s0, __float_as_uint(s1.z) != 0u, t) && (t < best.t || (t == best.t && pos < bpos)) && t < inf) {
                        best = Best{t, 0.0f, 0.0f, kSphereBit | slot};
                        bpos = pos;
                        bm = fmaxf(tmin, 0.0f);
                    }
                }
            } else if (OLD && k == KIND_MESH) { // the mesh's inner hierarchy follows inline, up to `skip`
                mesh_m = tmin;
                inner_end = skip;
            }
        }
    }
    resolve_jobs<OLD>(P, lane, o, d, J);
    take_noted_result(live, lane, J, best, bpos, bm);
    __builtin_amdgcn_wave_barrier();
    if (live && best.ref != kNone && (bm > best.t || P.force_replay != 0)) {
        cnt.replay++;
        replay_ordered<OLD, STATS>(P, o, d, inv, best, cnt);
    }
}

// ------------------------------------------------------------------------------------------------
// The fast walk for a wave of NEIGHBOURS (the camera rays of one 8x8 tile and one sample): wave-synchronous.
//
// Such rays meet nearly the same records -- on the 18 520-triangle dragon at 1080p a lane's own walk is 48 records long,
// the longest of a tile's 64 is 61, and all 64 together meet 82 different ones -- so the wave walks the UNION: it visits a
// record when any lane's own walk would, every lane whose walk includes the record tests it, a lane that fails an interior
// box sits out until the wave reaches that record's skip offset (subtrees are nested intervals of the array, so when every
// taking-part lane fails, nobody wants anything before the skip offset: the wave's next record is cur + 32 or skip).  The
// record comes ONCE per wave through the scalar data path (one s_load_dwordx8 into SGPRs) instead of 64 times, one line
// each, through the vector-memory path the per-lane walk is bound by.  Each lane performs exactly the slab tests of its own
// fast_walk_from, with the same arithmetic; leaves that pass are noted in the wave's list (places from a ballot: the lanes
// are in step, no atomic is needed) with the box's entry distance, and run by resolve_jobs, the list being emptied whenever
// the next leaf might not fit.  A mesh object's leaf (pointer scenes) is an interior record to the wave -- its inline
// hierarchy follows -- and to each lane that meets its box the start of a stretch in which entry distances are at least
// the mesh box's, as in fast_walk_from.  Every lane of the wave must call this.
template <bool OLD, bool STATS, int OCT> // OCT = sign bits (x | y<<1 | z<<2) shared by the wave's live lanes, or -1 = mixed
RT_D void fast_wave_walk_oct(const KParams &P, const bool live, const uint32_t lane, const f3 o, const f3 d, const f3 inv, Cnt &cnt,
                             WaveJobs &J, const bool nx, const bool ny, const bool nz) {
    constexpr bool UNI = OCT >= 0;
    constexpr bool ux = UNI && (OCT & 1), uy = UNI && (OCT & 2), uz = UNI && (OCT & 4);
    typedef const char __attribute__((address_space(4))) *cbytes;
    const cbytes nodes = (cbytes)(uintptr_t)P.fnodes;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t n = P.n_fnodes * 32u;
    const float inf = __builtin_inff();
    uint32_t resume = live ? 0u : kNone; // byte offset of the next record this lane takes part in
    uint32_t cur = 0u;                    // byte offset of the wave's current record
    uint32_t count = 0u;                  // jobs in the wave's list (the same in every lane)
    uint32_t inner_end = kNone;           // OLD: byte offset where the mesh this lane is inside ends
    float mesh_m = 0.0f;                  // OLD: entry distance of that mesh's box
    // (Measured and dropped: asking for the record behind this one as well -- scalar loads return in any order, so a wait for
    // one is a wait for all: 3.5 % slower --; records two at a time in one s_load_dwordx16: 15 % slower, the sixteen
    // registers spill; the record through the vector path, every lane asking for the same address: 15 % slower.)
    while (cur < n) {
        const v8f r = *(cf8ptr)(nodes + cur);
        const v4f a = {r[0], r[1], r[2], r[3]}, b = {r[4], r[5], r[6], r[7]};
        const uint32_t skip = __float_as_uint(b.z), kind = __float_as_uint(b.w);
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
        const float tmin = fmaxf(nearz, fmaxf(neary, fmaxf(nearx, 1e-6f)));
        const float tmax = fminf(farz, fminf(fary, fminf(farx, inf))); // the slab test with closest = +inf
        const bool act = resume <= cur;
        const bool ok = !(tmax <= tmin);
        if (OLD) {
            if (cur == inner_end) {
                mesh_m = 0.0f;
                inner_end = kNone;
            }
        }
        if (STATS)
            cnt.box += act ? 1ull : 0ull;
        resume = (act && !ok) ? skip : resume; // (a primitive leaf's skip offset is cur + 32)
        const unsigned long long passed = __builtin_amdgcn_ballot_w64(act && ok);
        const uint32_t k = kind & 7u;
        if (k == KIND_INTERNAL || (OLD && k == KIND_MESH)) {
            if (OLD && k == KIND_MESH && act && ok) {
                mesh_m = tmin;
                inner_end = skip;
            }
            if (passed == 0ull) { // nobody goes in: on behind the subtree
                cur = __builtin_amdgcn_readfirstlane(skip);
                continue;
            }
        } else if (passed != 0ull) {
            const uint32_t fresh = (uint32_t)__builtin_popcountll(passed);
            if (count + fresh > kJobCap) { // the list cannot take them: run what it holds (the minima per owner stay in J.key)
                if (lane == 0)
                    J.count = count;
                resolve_jobs<OLD>(P, lane, o, d, J);
                count = 0u;
            }
            if (act && ok) {
                const bool sphere = OLD && k == KIND_SPHERE;
                if (STATS) {
                    if (sphere)
                        cnt.sph++;
                    else
                        cnt.tri++;
                }
                const uint32_t slot = kind >> 3;
                const float m = sphere ? fmaxf(tmin, 0.0f) : OLD ? fmaxf(tmin, mesh_m) : tmin;
                J.jobs[count + (uint32_t)__builtin_popcountll(passed & below)] =
                    make_uint2((lane << kJobSlotBits) | (sphere ? kJobSphere : 0u) | slot, __float_as_uint(m));
            }
            count += fresh;
        }
        cur = __builtin_amdgcn_readfirstlane(cur + 32u);
    }
    if (lane == 0)
        J.count = count;
    resolve_jobs<OLD>(P, lane, o, d, J);
}

template <bool OLD, bool STATS>
RT_D void fast_wave_walk(const KParams &P, const bool live, const uint32_t lane, const f3 o, const f3 d, const f3 inv, Cnt &cnt, WaveJobs &J) {
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const unsigned long long lm = __builtin_amdgcn_ballot_w64(live);
    const unsigned long long mx = __builtin_amdgcn_ballot_w64(live && nx), my = __builtin_amdgcn_ballot_w64(live && ny),
                             mz = __builtin_amdgcn_ballot_w64(live && nz);
    const bool uniform = (mx == 0ull || mx == lm) && (my == 0ull || my == lm) && (mz == 0ull || mz == lm);
    const int oct = uniform ? ((mx != 0ull ? 1 : 0) | (my != 0ull ? 2 : 0) | (mz != 0ull ? 4 : 0)) : -1;
    switch (oct) {
    case 0: fast_wave_walk_oct<OLD, STATS, 0>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    case 1: fast_wave_walk_oct<OLD, STATS, 1>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    case 2: fast_wave_walk_oct<OLD, STATS, 2>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    case 3: fast_wave_walk_oct<OLD, STATS, 3>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    case 4: fast_wave_walk_oct<OLD, STATS, 4>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    case 5: fast_wave_walk_oct<OLD, STATS, 5>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    case 6: fast_wave_walk_oct<OLD, STATS, 6>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    case 7: fast_wave_walk_oct<OLD, STATS, 7>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    default: fast_wave_walk_oct<OLD, STATS, -1>(P, live, lane, o, d, inv, cnt, J, nx, ny, nz); break;
    }
}

// closest_hit_fast_queued for a wave of neighbours: the wave-synchronous walk, then each lane's best of the noted tests.
template <bool OLD, bool STATS>
RT_D void closest_hit_fast_wave(const KParams &P, const bool live, const uint32_t lane, const f3 o, const f3 d, const f3 inv, Best &best,
                                Cnt &cnt, WaveJobs &J) {
    best = Best{__builtin_inff(), 0.0f, 0.0f, kNone};
    uint32_t bpos = kNone;
    float bm = 0.0f;
    J.key[lane] = ~0ull;
    if (lane == 0)
        J.count = 0u;
    fast_wave_walk<OLD, STATS>(P, live, lane, o, d, inv, cnt, J);
    take_noted_result(live, lane, J, best, bpos, bm);
    __builtin_amdgcn_wave_barrier();
    if (live && best.ref != kNone && (bm > best.t || P.force_replay != 0)) {
        cnt.replay++;
        replay_ordered<OLD, STATS>(P, o, d, inv, best, cnt);
    }
}

// rt_kernels.hpp: path state, shading, camera rays, the first-generation kernel k_trace and the wavefront kernels k_camera, k_path, k_refill, k_sliced, k_fold -- part of rt_hip.hip (included there, inside its unnamed namespace); see that file for the overall design.
#pragma once

// Path state carried from one bounce level to the next.
struct Path {
    f3 o, d, inv; // current ray
    float4 thr;   // throughput (w used by the old-scene estimator only)
    float rr;     // old-scene estimator: roulette weight of the current depth
    int depth;    // traversals already started on this path
    uint32_t key; // RNG key of (pixel, sample)
};

// Shades one finished closest-hit search -- one step of either radiance estimator of CPURayTrace/Trace.cpp: the path either
// ENDS (returns true; `c` is the sample's colour) or CONTINUES with a new ray in `path` (returns false).  `prim` receives
// the primitive hit (kNone for a miss).
template <bool OLD, bool STATS>
RT_D bool shade_step(const KParams &P, Path &path, const Best &hit, Cnt &cnt, float4 &c, uint32_t &prim) {
    bool ended = false;
    c = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t dim = 3u + 4u * (uint32_t)path.depth;
    path.depth++;
    prim = kNone;
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
    return ended;
}

// shade_step for the wavefront kernels: an ended path's colour goes to colors[sample][pixel], and the primitive met by the
// camera ray of the call's first sample to the primary-id image.
// The per-sample buffers of a chunk (colours, `deferred` words) hold 64 slots per OWNED 8x8 tile, in the order this context
// takes its tiles: slot = tile ordinal * 64 + pixel within the tile.  A share of a frame therefore needs its share of the
// memory, and a wave's 64 slots are one contiguous kilobyte.  The frame pixel of a slot:
RT_D uint32_t pixel_of_slot(const KParams &P, const uint32_t slot) {
    const uint32_t tile = (uint32_t)P.tile_rank + (slot >> 6) * (uint32_t)P.tile_world;
    const uint32_t tile_y = fastdiv(tile, P.div_tiles_x);
    const uint32_t x = (tile - tile_y * (uint32_t)P.tiles_x) * 8u + (slot & 7u), y = tile_y * 8u + ((slot >> 3) & 7u);
    return y * (uint32_t)P.W + x;
}

template <bool OLD, bool STATS>
RT_D bool shade_path(const KParams &P, Path &path, const Best &hit, const uint32_t cidx, Cnt &cnt) {
    const bool first = path.depth == 0;
    float4 c;
    uint32_t prim;
    const bool ended = shade_step<OLD, STATS>(P, path, hit, cnt, c, prim);
    if (first && P.primary != nullptr && P.chunk_first + (int)(cidx / P.cpix) == P.first_sample)
        P.primary[pixel_of_slot(P, cidx % P.cpix)] = prim;
    if (ended)
        P.colors[cidx] = c;
    return ended;
}

// Direction and RNG key of sample n (1-based) of pixel (x, y): CPUImageData::uvAt, Renderer/TraceMethods.hpp:30;
// Camera::getRayDirction, Camera.hpp:33-47; the jitter of Renderer/TraceMethods.cpp:51-53.
RT_D void camera_ray(const KParams &P, const int x, const int y, const uint32_t pixel, const int n, uint32_t &key, f3 &dir) {
    const float u = (float)x / (float)P.W, v = (float)y / (float)P.H;
    f3 view = normalize3(mk3(u * P.cam_w - P.cam_w / 2, v * P.cam_h - P.cam_h / 2, P.focal));
    view.x = -view.x;
    const f3 dir0 = basis_mul(P.bx, P.by, P.bz, view);
    key = pcg(pcg(P.seed_hash + pixel) + (uint32_t)n + P.sample_offset);
    // glm::linearRand(vec3(-s), vec3(s)), Include/Random.hpp:13-16
    const float js = P.jitter;
    f3 jit = mk3(rng01(key, 0u) * (js - (-js)) + (-js), rng01(key, 1u) * (js - (-js)) + (-js),
                 rng01(key, 2u) * (js - (-js)) + (-js));
    dir = dir0 + jit;
}

// What a camera ray that meets nothing contributes: the colour shade_path stores for a depth-0 path that misses
// (throughput 1, roulette weight 1).
template <bool OLD>
RT_D float4 sky_sample(const f3 d) {
    const f3 sk = sky_color(d);
    if (!OLD)
        return make_float4(0.f + 1.f * sk.x, 0.f + 1.f * sk.y, 0.f + 1.f * sk.z, 0.f + 1.0f);
    return make_float4((1.f * sk.x) * 1.0f, (1.f * sk.y) * 1.0f, (1.f * sk.z) * 1.0f, (1.f * 1.0f) * 1.0f);
}

// The slab test of the fast walk (closest = +inf) against the root record of the fast tree: does the walk of this ray
// get past its first record at all?
RT_D bool meets_root_box(const float4 a, const float4 b, const f3 o, const f3 inv) {
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const float t0x = (a.x - o.x) * inv.x, t1x = (a.w - o.x) * inv.x;
    const float t0y = (a.y - o.y) * inv.y, t1y = (b.x - o.y) * inv.y;
    const float t0z = (a.z - o.z) * inv.z, t1z = (b.y - o.z) * inv.z;
    const float nearx = nx ? t1x : t0x, farx = nx ? t0x : t1x;
    const float neary = ny ? t1y : t0y, fary = ny ? t0y : t1y;
    const float nearz = nz ? t1z : t0z, farz = nz ? t0z : t1z;
    const float tmin = fmaxf(nearz, fmaxf(neary, fmaxf(nearx, 1e-6f)));
    const float tmax = fminf(farz, fminf(fary, fminf(farx, __builtin_inff())));
    return !(tmax <= tmin);
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
    cidx = s * P.cpix + j * 64u + in_tile;
    path.o = P.cam_pos;
    camera_ray(P, x, y, pixel, n, path.key, path.d);
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

// First-generation kernel (RT_TRAV_NESTED), kept as an independent schedule of the same arithmetic.  Persistent wave64:
// each wave draws 8x8-pixel tiles from one atomic counter; a lane owns one pixel of the tile and runs that pixel's samples
// in order -- every sample's path to its end, closest hit by the per-lane reference-order walk, shading by shade_step --
// which keeps the reference's running mean "pixel = (pixel*(n-1) + new)/n" (Renderer/TraceMethods.cpp:59) in registers.
// Finished lanes drop out of the wave's exec mask; the wave leaves a loop when the mask is empty.
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
                px = P.image_in[pixel];
            for (int n = P.first_sample; n < P.first_sample + P.n_samples; n++) {
                Path path;
                path.o = P.cam_pos;
                camera_ray(P, x, y, pixel, n, path.key, path.d);
                path.thr = make_float4(1.f, 1.f, 1.f, 1.f);
                path.rr = 1.0f;
                path.depth = 0;
                uint32_t prim = kNone, first_prim = kNone;
                float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                bool ended = !OLD && P.bounce_limit <= 0; // Trace.cpp:63: no traversal at all, the sample is (0,0,0,0)
                while (!ended) {
                    path.inv = mk3(1.0f / path.d.x, 1.0f / path.d.y, 1.0f / path.d.z); // Ray::Ray, CPURayTrace/Ray.hpp:8
                    Best hit;
                    cnt.rays++;
                    closest_hit<OLD, STATS>(P, path.o, path.d, path.inv, hit, cnt);
                    const bool first = path.depth == 0;
                    ended = shade_step<OLD, STATS>(P, path, hit, cnt, c, prim);
                    if (first)
                        first_prim = prim;
                }
                if (P.primary != nullptr && n == P.first_sample)
                    P.primary[pixel] = first_prim;
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

// What k_camera queues for a camera ray that got past the root box: 16 bytes -- direction and colour slot -- packed
// densely, one float4 each (record i of sub-queue q at float4 index q * sub_cap + i: the queue of a camera stage takes a sixth
// of the memory of a queue of path records).  The
// rest of the path state of a camera ray is known: origin, unit throughput, depth 0, and the RNG key of (pixel, sample).
RT_D void read_camera_record(const KParams &P, const float4 r, Path &path, uint32_t &cidx) {
    cidx = __float_as_uint(r.w);
    const uint32_t s = cidx / P.cpix, pixel = pixel_of_slot(P, cidx - s * P.cpix);
    path.o = P.cam_pos;
    path.d = mk3(r.x, r.y, r.z);
    path.thr = make_float4(1.f, 1.f, 1.f, 1.f);
    path.rr = 1.0f;
    path.depth = 0;
    path.key = pcg(pcg(P.seed_hash + pixel) + (uint32_t)(P.chunk_first + (int)s) + P.sample_offset);
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
constexpr int WALK_LANE = 0, WALK_WAVE = 1, WALK_FAST = 2, WALK_BRUTE = 3, WALK_QUEUED = 4;
template <bool OLD, bool STATS, int WALK>
__global__ void __launch_bounds__(256, WALK == WALK_WAVE ? 8 : WALK == WALK_QUEUED ? (OLD ? 4 : 6) : 1) k_path(const KParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    __shared__ WaveJobs s_jobs[WALK == WALK_QUEUED ? 4 : 1]; // only the queued walk touches it
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
    const bool tail = P.tail != 0 || (P.level >= 1 && n_in <= P.tail_rays);
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
            if (live && P.camera_stage != 0 && P.level == 1) {
                read_camera_record(P, P.q_in[g], path, cidx);
            } else if (live) {
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
        bool camera_batch = P.wave_camera != 0 && (P.level == 0 || (P.camera_stage != 0 && P.level == 1));
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
        } else if (WALK == WALK_QUEUED && camera_batch) { // camera rays of one tile and sample: neighbours
            closest_hit_fast_wave<OLD, STATS>(P, live, lane, path.o, path.d, path.inv, hit, cnt, s_jobs[threadIdx.x >> 6]);
        } else if (WALK == WALK_QUEUED) {
            closest_hit_fast_queued<OLD, STATS>(P, live, lane, path.o, path.d, path.inv, hit, cnt, s_jobs[threadIdx.x >> 6]);
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
        camera_batch = false; // what a lane follows from here on is its own path
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
    if (WALK == WALK_FAST || WALK == WALK_QUEUED) {
        unsigned long long rp = wave_sum(cnt.replay);
        if (lane == 0 && rp != 0ull)
            atomicAdd(counter_slot(P, wave_id, 5), rp);
    }
}

// What a lane of the compact walk does with a leaf record that passed its grid box, once it knows its place `j` in the
// wave's list (compact_walk_from): the leaf goes there (resolve_jobs_compact); with no room, its exact box test and
// triangle test run on the spot.
template <bool STATS, class JT>
RT_D void note_leaf(const KParams &P, const uint32_t lane, const uint32_t j, const uint32_t slot, const Path &path, JT &J, Best &best,
                    uint32_t &bpos, float &bm, Cnt &cnt) {
    if (j < JT::kCap && slot < kJobSphere) {
        J.put(j, (lane << kJobSlotBits) | slot);
        return;
    }
    float m, t, u, v;
    if (STATS)
        cnt.box++;
    const float4 q0 = P.leaf_boxes[4 * slot], q1 = P.leaf_boxes[4 * slot + 1], q2 = P.leaf_boxes[4 * slot + 2], q3 = P.leaf_boxes[4 * slot + 3];
    if (!exact_leaf_entry(q2, q3, path.o, path.inv, m))
        return;
    if (STATS)
        cnt.tri++;
    const uint32_t pos = __float_as_uint(q2.w);
    if (tri_test(path.o, path.d, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x), t, u, v) &&
        (t < best.t || (t == best.t && pos < bpos)) && t < __builtin_inff()) {
        best = Best{t, u, v, slot};
        bpos = pos;
        bm = m;
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
#ifndef RT_SERVICE_LANES
#define RT_SERVICE_LANES 24
#endif
constexpr uint32_t kServiceLanes = RT_SERVICE_LANES;
constexpr size_t kRefillSceneBytes = 32u << 20; // 8 XCDs x 4 MB of L2

// WALKS = true: the kernel finishes a round of k_sliced -- its queue holds that kernel's six-float4 records, rays in the
// middle of their walks among them -- and follows every path to its end (tail mode).
#ifndef RT_WIDE_WAVES
#define RT_WIDE_WAVES 5 // waves per SIMD the wide-tree instance of k_refill is compiled for (register allocation)
#endif
template <bool OLD, bool STATS, int COMPACT = 0, bool WALKS = false>
__global__ void __launch_bounds__(256, COMPACT == 3 ? RT_WIDE_WAVES : 1) k_refill(const KParams P) {
    constexpr size_t R4 = WALKS ? 6 : 4; // float4 per queue record
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    // Primitive tests are taken out of the walk as in closest_hit_fast_queued: lanes note them, the wave runs them together
    // before it shades (and whenever its list is about to fill up).
    constexpr bool WIDE = COMPACT == 3;
    using JobList = typename std::conditional<WIDE, WideJobs, WaveJobs>::type;
    __shared__ JobList s_jobs[4];
    JobList &J = s_jobs[threadIdx.x >> 6];
    // the wide walk's stacks: one word per lane and level of the scene's wide tree, a wave's 64 lanes side by side (no bank
    // conflicts), in the block's dynamic LDS (the launch sizes it: 4 waves x P.wide_depth levels x 64 lanes)
    extern __shared__ uint32_t s_dynamic[];
    uint32_t *const stk = s_dynamic + (WIDE ? (threadIdx.x >> 6) * (P.wide_depth * 64u) + lane : 0u);
    uint32_t sp = 0u;
    J.key[lane] = ~0ull;
    if (lane == 0)
        J.count = 0u;
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
    if (WALKS && P.tail == 0 && n_batches * 64u > P.tail_rays)
        return; // a round that still fills the chip: k_sliced, launched next, takes it
    const bool tail = WALKS || P.tail != 0 || (P.level >= 1 && n_batches * 64u <= P.tail_rays);
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t my_q = wave_id & (uint32_t)(kSubQueues - 1);
    const uint32_t my_count = P.level == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)sub_count, (int)my_q);
    const uint32_t my_batches = P.level == 0 ? (n_batches > my_q ? (n_batches - my_q + (uint32_t)kSubQueues - 1u) / (uint32_t)kSubQueues : 0u)
                                             : (my_count + 63u) >> 6;
    // a walk's position is a byte offset into the array it steps through (the wide walk: a node index), `n` when it is over
    const uint32_t n = WIDE ? kWideDone : P.n_fnodes * (COMPACT ? 16u : 32u);
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
    GridRay grid = {mk3(0.f, 0.f, 0.f), mk3(0.f, 0.f, 0.f)};
    // wide walk: the largest component of the ray's direction and its reciprocal, for the distance limit; `limited`: the
    // ray takes one (the scene offers a limit and the component is within the range its margin was derived for)
    float dmax = 1.0f, inv_dmax = 1.0f;
    bool limited = false;
    auto start_walk = [&]() {
        path.inv = mk3(1.0f / path.d.x, 1.0f / path.d.y, 1.0f / path.d.z); // Ray::Ray, CPURayTrace/Ray.hpp:8
        if (COMPACT)
            grid = grid_ray(P, path.o, path.inv);
        if (WIDE) {
            dmax = fmaxf(fmaxf(__builtin_fabsf(path.d.x), __builtin_fabsf(path.d.y)), __builtin_fabsf(path.d.z));
            limited = P.prune != 0 && dmax >= 0.25f && dmax <= 1.25f;
            inv_dmax = limited ? 1.0f / dmax : 1.0f;
        }
        w = Walk{0u, kNone, kNone, 0.0f};
        sp = 0u;
        best = Best{__builtin_inff(), 0.0f, 0.0f, kNone};
        bpos = kNone;
        bm = 0.0f;
        J.key[lane] = ~0ull;
        cnt.rays++;
    };
    // The closest hit this lane's walk knows of, for its distance limit: its own tests and what the noted tests that have run
    // so far left in the wave's list (the key's upper word is the hit distance, all ones -- not a number -- for none).
    auto closest_so_far = [&]() -> float {
        return limited ? fminf(best.t, __uint_as_float((uint32_t)(J.key[lane] >> 32))) : __builtin_inff();
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
                    bool walking = false;
                    if (P.camera_stage != 0 && P.level == 1) {
                        read_camera_record(P, P.q_in[g], path, cidx);
                    } else {
                        const float4 r0 = P.q_in[R4 * g], r1 = P.q_in[R4 * g + 1], r2 = P.q_in[R4 * g + 2], r3 = P.q_in[R4 * g + 3];
                        path.o = mk3(r0.x, r0.y, r0.z);
                        path.d = mk3(r0.w, r1.x, r1.y);
                        path.thr = make_float4(r1.z, r1.w, r2.x, r2.y);
                        path.rr = r2.z;
                        path.key = __float_as_uint(r2.w);
                        cidx = __float_as_uint(r3.x);
                        path.depth = (int)__float_as_uint(r3.y);
                        if (WALKS && !WIDE && __float_as_uint(r3.z) != 0u) { // a ray in the middle of its walk (a walk never returns to record 0; the wide walk is never cut)
                            const float4 r4 = P.q_in[R4 * g + 4], r5 = P.q_in[R4 * g + 5];
                            walking = true;
                            path.inv = mk3(1.0f / path.d.x, 1.0f / path.d.y, 1.0f / path.d.z); // Ray::Ray, CPURayTrace/Ray.hpp:8
                            if (COMPACT)
                                grid = grid_ray(P, path.o, path.inv);
                            w = Walk{__float_as_uint(r3.z), __float_as_uint(r5.w), __float_as_uint(r5.y), r5.z};
                            best = Best{r4.x, r4.y, r4.z, __float_as_uint(r3.w)};
                            bpos = __float_as_uint(r4.w);
                            bm = r5.x;
                            J.key[lane] = ~0ull;
                        }
                    }
                    out_q = (my_q + cur_b) & (uint32_t)(kSubQueues - 1);
                    has = true;
                    if (!walking)
                        start_walk();
                }
                if (has && P.level == 0)
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
            if constexpr (WIDE) {
                if (has && w.idx < n)
                    wide_walk_from<STATS>(P, grid, w.idx, sp, stk, P.wide_burst, closest_so_far(), dmax, inv_dmax, cnt,
                                      [&](const uint32_t k) { return atomicAdd(&J.count, k); },
                                      [&](const uint32_t j, const uint32_t slot) { note_leaf<STATS>(P, lane, j, slot, path, J, best, bpos, bm, cnt); });
            } else if constexpr (COMPACT != 0) {
                if (has && w.idx < n)
                    compact_walk_from<STATS, COMPACT == 2>(P, path.inv, grid, w.idx, kBurst, cnt,
                                         [&]() { return atomicAdd(&J.count, 1u); },
                                         [&](const uint32_t j, const uint32_t slot) { note_leaf<STATS>(P, lane, j, slot, path, J, best, bpos, bm, cnt); });
            } else {
                if (has && w.idx < n)
                fast_walk_from<OLD, STATS, kBurst>(P, path.o, path.d, path.inv, w, cnt, [&](const Cand &c) {
                    if ((c.t < best.t || (c.t == best.t && c.pos < bpos)) && c.t < __builtin_inff()) {
                        best = Best{c.t, c.u, c.v, c.ref};
                        bpos = c.pos;
                        bm = fmaxf(c.m, c.mesh_m);
                    }
                }, [&](const uint32_t sphere, const uint32_t slot, const float m) -> bool {
                    const uint32_t j = atomicAdd(&J.count, 1u);
                    if (!(j < kJobCap && slot < kJobSphere))
                        return false; // no room: the walk runs the test itself
                    J.jobs[j] = make_uint2((lane << kJobSlotBits) | (sphere != 0u ? kJobSphere : 0u) | slot, __float_as_uint(m));
                    return true;
                });
            }
            __builtin_amdgcn_wave_barrier();
            // the noted tests run when the next burst could overflow the list -- and, where the walk keeps a distance limit, as
            // soon as there is a full pass of them: what they find shortens the walks that go on
            if (J.count + JobList::kCap / 2u > JobList::kCap || (WIDE && P.prune != 0 && J.count >= 64u)) {
                if constexpr (COMPACT != 0)
                    resolve_jobs_compact<STATS>(P, lane, path.o, path.d, path.inv, J, cnt);
                else
                    resolve_jobs<OLD>(P, lane, path.o, path.d, J);
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
        if constexpr (COMPACT != 0)
            resolve_jobs_compact<STATS>(P, lane, path.o, path.d, path.inv, J, cnt);
        else
            resolve_jobs<OLD>(P, lane, path.o, path.d, J);
        take_noted_result(has && !(w.idx < n), lane, J, best, bpos, bm);
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

// Path kernel of the fast walk in SLICES (one launch per round).
//
// Walk lengths are heavy-tailed (bounce rays of the 18 520-triangle dragon: median 51 records, mean 75, 1 % above 267,
// longest 445), so a wave that walks 64 rays to the end has, at the bounce levels, a fraction of its lanes busy.  Here a
// launch takes every ray of its queue at most `slice` records further.  A ray whose walk is not over goes to the next
// round's queue WITH its walk (next record, best candidate so far); a ray whose walk is over is shaded, and a path that
// continues puts its new ray into the same queue.  The next round runs again in full waves: a lane idles for at most a
// slice, not for the difference between the longest and the mean walk.  Primitive tests are noted and run together at the
// end of the slice (WaveJobs), so that the dense walk is not interrupted for the few lanes that stand on a leaf.  A round
// whose queue cannot fill the chip finishes everything in place (tail mode, as k_path).  Records are six float4: the
// four of k_path plus the walk.  Same per-ray arithmetic as k_path, same results.
template <bool OLD, bool STATS, int COMPACT = 0>
__global__ void __launch_bounds__(256) k_sliced(const KParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    __shared__ WaveJobs s_jobs[4];
    WaveJobs &J = s_jobs[threadIdx.x >> 6];
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
    const uint32_t n_in = P.level == 0 ? P.n_items : n_batches * 64u;
    if (P.refill_tail != 0 && P.level >= 1 && n_in <= P.tail_rays)
        return; // finished by the re-fill kernel launched before this one
    const bool tail = P.tail != 0 || (P.level >= 1 && n_in <= P.tail_rays);
    const int slice = tail ? 0x7FFFFFFF : (int)P.slice;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr uint32_t kChunk = 4u;
    const uint32_t my_q = wave_id & (uint32_t)(kSubQueues - 1);
    const uint32_t my_count = P.level == 0 ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)sub_count, (int)my_q);
    const uint32_t my_batches = P.level == 0 ? (n_batches > my_q ? (n_batches - my_q + (uint32_t)kSubQueues - 1u) / (uint32_t)kSubQueues : 0u)
                                             : (my_count + 63u) >> 6;
    const uint32_t chunk = P.level == 0 ? kChunk : 1u;
    const uint32_t n = P.n_fnodes * (COMPACT ? 16u : 32u); // a walk's position is a byte offset into the array it steps through
    J.key[lane] = ~0ull;
    if (lane == 0)
        J.count = 0u;
    uint32_t next = 0u, left = 0u;
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
        uint32_t out_q;
        bool live;
        Path path;
        uint32_t cidx = 0;
        Walk w = {0u, kNone, kNone, 0.0f, slice};
        Best best = {__builtin_inff(), 0.0f, 0.0f, kNone};
        uint32_t bpos = kNone;
        float bm = 0.0f;
        if (P.level == 0) {
            const uint32_t batch = my_q + b * (uint32_t)kSubQueues;
            live = camera_path<OLD>(P, batch * 64u + lane, path, cidx);
            out_q = my_q;
        } else {
            const uint32_t local = b * 64u + lane;
            live = local < my_count;
            out_q = (my_q + b) & (uint32_t)(kSubQueues - 1);
            path.o = path.d = mk3(1.f, 1.f, 1.f);
            path.thr = make_float4(1.f, 1.f, 1.f, 1.f);
            path.rr = 1.0f;
            path.depth = 0;
            path.key = 0;
            if (live && P.camera_stage != 0 && P.level == 1) { // a camera ray: its walk has not started
                read_camera_record(P, P.q_in[(size_t)my_q * P.sub_cap + local], path, cidx);
            } else if (live) {
                const size_t g = (size_t)my_q * P.sub_cap + local;
                const float4 r0 = P.q_in[6 * g], r1 = P.q_in[6 * g + 1], r2 = P.q_in[6 * g + 2], r3 = P.q_in[6 * g + 3],
                             r4 = P.q_in[6 * g + 4], r5 = P.q_in[6 * g + 5];
                path.o = mk3(r0.x, r0.y, r0.z);
                path.d = mk3(r0.w, r1.x, r1.y);
                path.thr = make_float4(r1.z, r1.w, r2.x, r2.y);
                path.rr = r2.z;
                path.key = __float_as_uint(r2.w);
                cidx = __float_as_uint(r3.x);
                path.depth = (int)__float_as_uint(r3.y);
                w.idx = __float_as_uint(r3.z);
                best.ref = __float_as_uint(r3.w);
                best.t = r4.x;
                best.u = r4.y;
                best.v = r4.z;
                bpos = __float_as_uint(r4.w);
                bm = r5.x;
                w.inner_end = __float_as_uint(r5.y);
                w.mesh_m = r5.z;
                w.mesh = __float_as_uint(r5.w);
            }
        }
        // A batch of camera rays whose walks have not started holds neighbours -- one tile, one sample (or a few) --: the wave
        // walks the records they meet together, to the end (fast_wave_walk); everything else goes lane by lane, in slices.
        bool camera_batch = !OLD && P.wave_camera != 0 && (P.level == 0 || (P.camera_stage != 0 && P.level == 1));
        for (;;) {
            if (__ballot(live) == 0ull)
                break;
            bool ended = false;
            if (live) {
                path.inv = mk3(1.0f / path.d.x, 1.0f / path.d.y, 1.0f / path.d.z); // Ray::Ray, CPURayTrace/Ray.hpp:8
                if (w.idx == 0u)
                    cnt.rays++; // a walk starts at record 0 and never returns to it
            }
            if constexpr (!OLD) {
                if (camera_batch) {
                    fast_wave_walk<false, STATS>(P, live, lane, path.o, path.d, path.inv, cnt, J);
                    if (live)
                        w.idx = n;
                }
            }
            // The slice is walked in bursts; between two bursts the wave runs the tests noted so far if its list is more than
            // half full.  A full list makes the walk test on the spot, a few lanes at a time, which costs far more than a
            // pass of resolve_jobs (the minima per owner stay in the wave's LDS words from one pass to the next).
            GridRay grid;
            if (COMPACT)
                grid = grid_ray(P, path.o, path.inv);
            for (int left = slice; !camera_batch;) {
                if (COMPACT && live && w.idx < n) {
                    compact_walk_from<STATS, COMPACT == 2>(P, path.inv, grid, w.idx, left < kSliceBurst ? left : kSliceBurst, cnt,
                                             [&]() { return atomicAdd(&J.count, 1u); },
                                         [&](const uint32_t j, const uint32_t slot) { note_leaf<STATS>(P, lane, j, slot, path, J, best, bpos, bm, cnt); });
                } else if (live && w.idx < n) {
                    w.slice = left < kSliceBurst ? left : kSliceBurst;
                    fast_walk_from<OLD, STATS, -1>(P, path.o, path.d, path.inv, w, cnt, [&](const Cand &c) {
                        if ((c.t < best.t || (c.t == best.t && c.pos < bpos)) && c.t < __builtin_inff()) {
                            best = Best{c.t, c.u, c.v, c.ref};
                            bpos = c.pos;
                            bm = fmaxf(c.m, c.mesh_m);
                        }
                    }, [&](const uint32_t sphere, const uint32_t slot, const float m) -> bool {
                        const uint32_t j = atomicAdd(&J.count, 1u);
                        if (!(j < kJobCap && slot < kJobSphere))
                            return false; // no room: the walk runs the test itself
                        J.jobs[j] = make_uint2((lane << kJobSlotBits) | (sphere != 0u ? kJobSphere : 0u) | slot, __float_as_uint(m));
                        return true;
                    });
                }
                left = left > kSliceBurst ? left - kSliceBurst : 0;
                __builtin_amdgcn_wave_barrier();
                {
                    // the slice ends early for a wave most of whose rays are done: the few that still walk go to the next
                    // round, where they sit in full waves again, instead of keeping this one at a handful of lanes
                    const uint32_t walking = (uint32_t)__builtin_popcountll(__ballot(live && w.idx < n));
                    if (left == 0 || walking == 0u || (!tail && walking <= P.slice_quorum))
                        break;
                }
                if (J.count > kJobCap / 2u) {
                    if (COMPACT)
                        resolve_jobs_compact<STATS>(P, lane, path.o, path.d, path.inv, J, cnt);
                    else
                        resolve_jobs<OLD>(P, lane, path.o, path.d, J);
                }
            }
            // the noted tests of this slice, then every lane's best is complete up to where its walk stands
            if (COMPACT)
                resolve_jobs_compact<STATS>(P, lane, path.o, path.d, path.inv, J, cnt);
            else
                resolve_jobs<OLD>(P, lane, path.o, path.d, J);
            take_noted_result(live, lane, J, best, bpos, bm);
            J.key[lane] = ~0ull;
            camera_batch = false;
            if (live && !(w.idx < n)) {
                if (best.ref != kNone && (bm > best.t || P.force_replay != 0)) { // see closest_hit_fast
                    cnt.replay++;
                    replay_ordered<OLD, STATS>(P, path.o, path.d, path.inv, best, cnt);
                }
                ended = shade_path<OLD, STATS>(P, path, best, cidx, cnt);
                w = Walk{0u, kNone, kNone, 0.0f, slice};
                best = Best{__builtin_inff(), 0.0f, 0.0f, kNone};
                bpos = kNone;
                bm = 0.0f;
            }
            live = live && !ended;
            if (!tail)
                break;
        }
        if (tail)
            continue;
        // ---- unfinished walks and continuing paths go to the next round's queue, compacted per wave ----
        const unsigned long long cmask = __ballot(live);
        if (cmask != 0ull) {
            unsigned int obase = 0;
            if (lane == 0)
                obase = atomicAdd(P.sub_out + out_q * 32u, (unsigned int)__builtin_popcountll(cmask));
            obase = __builtin_amdgcn_readfirstlane(obase);
            if (live) {
                const size_t slot = (size_t)out_q * P.sub_cap + (size_t)obase + (size_t)__builtin_popcountll(cmask & below);
                P.q_out[6 * slot] = make_float4(path.o.x, path.o.y, path.o.z, path.d.x);
                P.q_out[6 * slot + 1] = make_float4(path.d.y, path.d.z, path.thr.x, path.thr.y);
                P.q_out[6 * slot + 2] = make_float4(path.thr.z, path.thr.w, path.rr, __uint_as_float(path.key));
                P.q_out[6 * slot + 3] = make_float4(__uint_as_float(cidx), __uint_as_float((uint32_t)path.depth), __uint_as_float(w.idx),
                                                    __uint_as_float(best.ref));
                P.q_out[6 * slot + 4] = make_float4(best.t, best.u, best.v, __uint_as_float(bpos));
                P.q_out[6 * slot + 5] = make_float4(bm, __uint_as_float(w.inner_end), w.mesh_m, __uint_as_float(w.mesh));
            }
        }
    }
    unsigned long long r = wave_sum(cnt.rays);
    if (lane == 0 && r != 0ull)
        atomicAdd(counter_slot(P, wave_id, 0), r);
    if (STATS) {
        unsigned long long bx = wave_sum(cnt.box), t = wave_sum(cnt.tri), sp = wave_sum(cnt.sph), h = wave_sum(cnt.trih);
        if (lane == 0 && (r | bx | t | sp | h) != 0ull) {
            atomicAdd(counter_slot(P, wave_id, 1), bx);
            atomicAdd(counter_slot(P, wave_id, 2), t);
            atomicAdd(counter_slot(P, wave_id, 3), sp);
            atomicAdd(counter_slot(P, wave_id, 4), h);
        }
    }
    unsigned long long rp = wave_sum(cnt.replay);
    if (lane == 0 && rp != 0ull)
        atomicAdd(counter_slot(P, wave_id, 5), rp);
}

// Camera stage of a chunk (launched instead of level 0 of the path kernels when the scene covers little of the frame).
//
// Most camera rays of such a frame fail the walk's very first slab test -- the root record of the fast tree -- and end as
// sky.  Giving each of them a queue slot, a colour slot and a lane of the path kernel costs more than the ray: 32 bytes of
// HBM traffic per sample, and the wave's walk machinery for one record.  Here a wave takes an 8x8 tile and each lane runs
// through the chunk's S samples of its pixel: camera ray, root slab test.  Only the samples that pass go to the queue (as
// 16-byte records, see read_camera_record; the path kernels take them from level 1 on); the others are noted as a clear bit in the
// pixel's `deferred` word and never touch memory again: k_fold regenerates their ray and folds the sky colour where their
// turn in the sample order comes.  One queue reservation per tile (none for a tile that is all sky); records are
// sample-major within it, so a batch of the next level holds neighbouring pixels of one sample.
template <bool OLD, bool STATS>
__global__ void __launch_bounds__(256) k_camera(const KParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    unsigned long long n_sky = 0ull;
    const float4 root_a = P.fnodes[0], root_b = P.fnodes[1]; // the root record's box, once
    // One wave per (owned tile, group of 64 samples): a share of a frame has few tiles and many samples per chunk, and would
    // leave most of the chip idle with one wave per tile.
    const uint32_t groups = ((uint32_t)P.S + 63u) >> 6;
    for (uint32_t jg = wave_id; jg < P.n_my_tiles * groups; jg += n_waves) {
        const uint32_t j = jg / groups;
        const int s0 = (int)(jg - j * groups) * 64;
        const uint32_t tile = (uint32_t)P.tile_rank + j * (uint32_t)P.tile_world;
        const uint32_t tile_y = fastdiv(tile, P.div_tiles_x);
        const int x = (int)(tile - tile_y * (uint32_t)P.tiles_x) * 8 + (int)(lane & 7u);
        const int y = (int)tile_y * 8 + (int)(lane >> 3);
        const bool live = x < P.W && y < P.H;
        const uint32_t pixel = (uint32_t)y * (uint32_t)P.W + (uint32_t)x;
        // (64 samples: one word of the pixel's `deferred` mask, one reservation in the queue)
        {
            const int sb = P.S - s0 < 64 ? P.S - s0 : 64;
            unsigned long long mask = 0ull;
            // a tile outside the widened projection of the scene's box, or of every box of the scene's cut (root_box_tiles,
            // camera_tile_mask): every sample is sky, nothing to test
            const int tile_x = (int)(tile - tile_y * (uint32_t)P.tiles_x);
            const bool may_meet = tile_x >= P.cam_tx0 && tile_x <= P.cam_tx1 && (int)tile_y >= P.cam_ty0 && (int)tile_y <= P.cam_ty1 &&
                                  (P.cam_mask == nullptr || P.cam_mask[tile] != 0);
            if (live) {
                for (int s = 0; may_meet && s < sb; s++) {
                    uint32_t key;
                    f3 d;
                    camera_ray(P, x, y, pixel, P.chunk_first + s0 + s, key, d);
                    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); // Ray::Ray, CPURayTrace/Ray.hpp:8
                    if (meets_root_box(root_a, root_b, P.cam_pos, inv))
                        mask |= 1ull << s;
                }
                P.deferred[(size_t)(s0 >> 6) * P.cpix + j * 64u + lane] = mask;
                n_sky += (unsigned long long)(sb - __builtin_popcountll(mask));
                if (P.primary != nullptr && s0 == 0 && P.chunk_first == P.first_sample && !(mask & 1ull))
                    P.primary[pixel] = kNone;
            }
            // ---- the samples that get past the root box: one reservation for the tile, records sample-major ----
            uint32_t mine = (uint32_t)__builtin_popcountll(mask), total = mine;
            for (int off = 32; off > 0; off >>= 1)
                total += (uint32_t)__shfl_xor((int)total, off, 64);
            if (total == 0u)
                continue;
            // Which sub-queue: the (tile, group) units are dealt round-robin in group-major order -- all full groups of 64
            // samples first, the chunk's last, possibly partial group after them -- so every sub-queue receives at most one
            // unit more than its share of either kind: its records stay within (tiles * S + 2 * 4096), which the queue's
            // slack covers (queue_records), however few tiles a share of a frame has and however many samples its chunks hold.
            const uint32_t q = ((uint32_t)(s0 >> 6) * P.n_my_tiles + j) & (uint32_t)(kSubQueues - 1);
            unsigned int base = 0;
            if (lane == 0)
                base = atomicAdd(P.sub_out + q * 32u, total);
            base = __builtin_amdgcn_readfirstlane(base);
            unsigned int lo32 = (unsigned int)mask, hi32 = (unsigned int)(mask >> 32);
            for (int off = 32; off > 0; off >>= 1) { // wave-wide OR of the masks: the samples somebody queued
                lo32 |= (unsigned int)__shfl_xor((int)lo32, off, 64);
                hi32 |= (unsigned int)__shfl_xor((int)hi32, off, 64);
            }
            unsigned long long any = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)hi32) << 32) |
                                     (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)lo32);
            while (any != 0ull) {
                const int s = __builtin_ctzll(any);
                any &= any - 1ull;
                const bool has = (mask >> s) & 1ull;
                const unsigned long long bm = __ballot(has);
                if (has) {
                    uint32_t key;
                    f3 d;
                    camera_ray(P, x, y, pixel, P.chunk_first + s0 + s, key, d);
                    const size_t slot = (size_t)base + (size_t)__builtin_popcountll(bm & below);
                    const uint32_t cidx = (uint32_t)(s0 + s) * P.cpix + j * 64u + lane;
                    P.q_out[(size_t)q * P.sub_cap + slot] = make_float4(d.x, d.y, d.z, __uint_as_float(cidx));
                }
                base += (unsigned int)__builtin_popcountll(bm);
            }
        }
    }
    // a sample that ends here is one closest-hit search of one slab test
    const unsigned long long r = wave_sum(n_sky);
    if (lane == 0 && r != 0ull) {
        atomicAdd(counter_slot(P, wave_id, 0), r);
        if (STATS)
            atomicAdd(counter_slot(P, wave_id, 1), r);
    }
}

// Folds the S samples of every owned pixel into the running mean, in sample order:
// pixel = (pixel*(n-1) + colour_n)/n, Renderer/TraceMethods.cpp:59.  A sample's colour comes from the per-sample colour
// buffer, or -- camera stage, sample not deferred -- is the sky seen by its camera ray, worked out here.
template <bool OLD>
__global__ void __launch_bounds__(256) k_fold(const KParams P) {
    // one wave per owned 8x8 tile (every lane has a pixel, whatever share of the frame this device renders)
    const uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (j >= P.n_my_tiles)
        return;
    const uint32_t tile = (uint32_t)P.tile_rank + j * (uint32_t)P.tile_world;
    const uint32_t tile_y = fastdiv(tile, P.div_tiles_x);
    const uint32_t x = (tile - tile_y * (uint32_t)P.tiles_x) * 8u + (lane & 7u), y = tile_y * 8u + (lane >> 3);
    if (x >= (uint32_t)P.W || y >= (uint32_t)P.H)
        return;
    const uint32_t pixel = y * (uint32_t)P.W + x;
    float4 px = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.chunk_first > 1)
        px = P.image_in[pixel];
    unsigned long long mask = ~0ull;
    for (int s = 0; s < P.S; s++) {
        const int n = P.chunk_first + s;
        if (P.camera_stage && (s & 63) == 0)
            mask = P.deferred[(size_t)(s >> 6) * P.cpix + j * 64u + lane]; // one word per 64 samples of the chunk
        float4 c;
        if ((mask >> (s & 63)) & 1ull) {
            c = P.colors[(size_t)s * P.cpix + j * 64u + lane];
        } else {
            uint32_t key;
            f3 d;
            camera_ray(P, (int)x, (int)y, pixel, n, key, d);
            c = sky_sample<OLD>(d);
        }
        const float nm1 = static_cast<float>((float)n - 1.f), nf = static_cast<float>(n);
        // (alpha: a mean of ones stays one -- (1 * (n - 1) + 1) / n is n / n while n is exact in fp32 -- without the division)
        const float pw = (px.w == 1.0f && c.w == 1.0f && n < (1 << 24)) ? 1.0f : (px.w * nm1 + c.w) / nf;
        px = make_float4((px.x * nm1 + c.x) / nf, (px.y * nm1 + c.y) / nf, (px.z * nm1 + c.z) / nf, pw);
    }
    P.image[pixel] = px;
}

// Several devices behind one context, tile split: `src` is the frame as device `rank` of `world` rendered it (its own
// 8x8 tiles k with k % world == rank, anything elsewhere); those tiles are copied into the assembled frame.
__global__ void __launch_bounds__(256) k_take_tiles(float4 *dst, const float4 *src, uint32_t *dst_prim, const uint32_t *src_prim,
                                                    uint32_t npix, int W, int tiles_x, int rank, int world) {
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    if (pixel >= npix)
        return;
    const uint32_t x = pixel % (uint32_t)W, y = pixel / (uint32_t)W;
    const uint32_t tile = (y >> 3) * (uint32_t)tiles_x + (x >> 3);
    if ((int)(tile % (uint32_t)world) != rank)
        return;
    dst[pixel] = src[pixel];
    if (dst_prim != nullptr)
        dst_prim[pixel] = src_prim[pixel];
}

// Several devices behind one context, sample split: the frame is the sum of the devices' own means weighted by their
// share of the samples, added in device order (first != 0: this is the first term).
__global__ void __launch_bounds__(256) k_add_weighted(float4 *dst, const float4 *src, uint32_t npix, float w, int first) {
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    if (pixel >= npix)
        return;
    const float4 s = src[pixel];
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!first)
        d = dst[pixel];
    dst[pixel] = make_float4(d.x + s.x * w, d.y + s.y * w, d.z + s.z * w, d.w + s.w * w);
}

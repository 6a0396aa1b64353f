// rt_device_common.hpp: fp32 vector helpers in glm order, constants, kernel parameters, counter RNG, cosine sampling, sky -- part of rt_hip.hip (included there, inside its unnamed namespace); see that file for the overall design.
#pragma once

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
constexpr int kWideStack = 16;             // entries of a lane's stack in the wide walk = interior levels of the deepest wide tree the upload accepts
constexpr uint32_t kWideWaveMinItems = 1u << 20; // wide tree: a chunk with more pixel samples than this walks its fresh camera rays wave-synchronously first
constexpr uint32_t kWideWaveMaxNodes = 2048; // wide records of the largest tree whose fresh camera rays are walked wave-synchronously first
// Nodes a lane of the wide walk visits between two looks at the wave's state, by what a random ray through the scene's box
// visits (the sum of the wide nodes' areas over the root's, measured at upload).  Short walks -- a mesh on a ground quad that
// fills the view -- finish inside a long burst and their lanes wait for its end; long walks pay every look with the ballots
// and the list check.  Swept 2 / 3 / 4 / 6 on four scenes: dragon on a ground quad (1.1 visits, within the L2s) 24.9 / 25.3 /
// 26.4 / 28.7 ms per 128 spp, the same with the 889 k-triangle dragon beside it (beyond the L2s) level from 2 to 4, bare
// dragon (10.4) 18.5 / 18.0 / 17.85 / 17.65 per 256 spp, 1.19 M-triangle field (11.8) 36.6 / 35.8 / 35.3 / 34.8 per 64 spp.
constexpr int kWideBurst = 4, kWideBurstShort = 2, kWideBurstLong = 6;
constexpr double kWideShortVisits = 2.0, kWideLongVisits = 10.0;
constexpr int kMaxLevels = 64;             // bounce levels a wavefront trace can launch (bounce_limit <= 62)
constexpr uint32_t kTailRays = 256u << 10;  // a level with no more rays than this follows every path to its end in one launch
constexpr uint32_t kSlice = 128;            // k_sliced: records a ray is taken further per round
constexpr int kSliceQuorum = 16;           // k_sliced: a wave with no more walking lanes than this ends its slice (8 / 16 / 24 / 32 swept: -1 / -3.5 / -3.8 / -2.7 % on the headline)
constexpr int kSliceBurst = 32;            // k_sliced: records between two looks at the wave's list of noted tests
constexpr int kSliceExtraRounds = 0;       // k_sliced: launches beyond one per bounce level (the last one finishes whatever is left)
constexpr uint32_t kSliceTailRays = 12u << 20;    // k_sliced: a round with no more rays than this is finished by the re-fill kernel (1 ... 32 Mi swept)
constexpr uint32_t kSliceTailRaysBig = 24u << 20; // ... in scenes beyond the L2s
constexpr int kChunkSamples = 64;          // samples per pixel traced per wavefront chunk
constexpr int kMaxChunkSamples = 512;      // ... of a share of a frame, or of a small frame
constexpr size_t kChunkItems = 144u << 20; // pixel samples a chunk is filled up to (64 samples of a 1080p frame: 133 M)
constexpr size_t kMaxItems = 192u << 20;   // work items (pixel samples) per chunk: bounds the queue workspace

// ------------------------------------------------------------------ kernel parameters
struct KParams {
    const float4 *nodes; // 2 x float4 per node, reference visiting order
    const float4 *fnodes; // 2 x float4 per node of the fast tree (pre-order, exact-union boxes over the reference's leaf boxes)
    const uint4 *cnodes;       // compact fast tree (flat scenes): 16 bytes per node, see SceneLayout::cn
    const float4 *leaf_boxes;  // ... and its leaves: triangle and exact box, 4 x float4 per triangle slot
    const uint4 *wnodes;       // wide fast tree (flat scenes): 64 bytes per node of up to eight children, see build_wide
    int wide_burst;            // ... nodes a lane visits per burst (kWideBurst and its neighbours)
    int cam_tx0, cam_ty0, cam_tx1, cam_ty1; // camera stage: the 8x8 tiles (inclusive range) outside which no camera ray can meet the scene's box
    const uint8_t *cam_mask;   // ... and, within it, one byte per tile of the frame: 0 = no camera ray of the tile can meet a box of the scene's cut (null: no mask)
    uint32_t wide_depth;       // ... and the entries a lane's stack needs for it (interior levels below the root, at least 1)
    int prune;                 // the wide walk keeps a distance limit (the scene's leaves allow one, see LeafShape in scene_layout.hpp)
    float prune_c1[3], prune_c2x2[3], prune_c3[3]; // ... its constants by leaf class: c1, 2 c2, c2 W + c3
    f3 q_origin, q_step;       // ... and the grid its 16-bit planes live on
    int compact;               // the sliced / re-fill walks go over the compact tree (2: two records per trip to memory; 3: the re-fill walk goes over the wide tree)
    uint32_t slice_quorum;     // k_sliced: a wave with no more walking lanes than this ends its slice
    int wave_camera;           // k_sliced: batches of fresh camera rays take the wave-synchronous walk (fast_wave_walk)
    int refill_tail;           // k_sliced leaves a round of at most tail_rays rays to the re-fill kernel launched before it
    const float4 *tri_pos; // 3 x float4 per triangle slot: p0, e1, e2, prim id, material
    const float4 *tri_nrm; // 3 x float4 per triangle slot: n0, n1, n2
    const float4 *spheres; // 2 x float4 per sphere
    const float4 *mats;    // 2 x float4 per material
    float4 *image;          // the running mean the folds write ...
    const float4 *image_in; // ... and the one the first fold of the call continues from (the same buffer unless look-ahead)
    uint32_t *primary;
    unsigned int *queue;
    unsigned long long *counters; // rays, box, tri, sph, tri_hits, replays
    uint32_t n_nodes;
    uint32_t n_fnodes;
    int tail;         // k_path: the last level (follow every path to its end; nothing is queued)
    uint32_t slice;   // k_sliced: records a ray is taken further per round
    uint32_t tail_rays; // k_sliced: a round with no more rays than this finishes every path in place
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
    unsigned long long *deferred; // camera stage: per pixel, bit s set = sample s of the chunk went to the queue (its colour
                                  // comes through `colors`); clear = it left to the sky at once (k_fold works it out itself)
    int camera_stage;           // the chunk starts with k_camera; the path kernels run at levels 1 ..
    const unsigned int *sub_in; // levels > 0: record counts of this level's sub-queues (one per 128 bytes)
    unsigned int *sub_cursor;   // levels > 0: batch cursors of this level's sub-queues (one per 128 bytes)
    unsigned int *sub_out;      // append counters of the next level's sub-queues
    uint32_t sub_cap;           // records a sub-queue can hold; sub-queue q starts at record q * sub_cap
    uint32_t n_items;           // level 0: n_my_tiles * 64 * S
    uint32_t npix;
    uint32_t cpix;              // colour slots per sample of a chunk: 64 per owned tile (slot = tile ordinal * 64 + pixel in the tile)
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


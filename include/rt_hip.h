/*
 * rt_hip.h -- C ABI of the MI355X (gfx950) path tracer, librt_hip.so.
 *
 * This is the seam a C++ adaptor behind the reference's tracer plug-in interface calls:
 *   ITraceMethod::trace(const Texture2D&, Texture2D&, int)   Renderer/RenderInterfaces.hpp:5-12
 *   ILoadMethod::load()                                      Renderer/RenderInterfaces.hpp:15-22
 * The reference has no FFI of its own for this path (its trace methods call Trace::CastRay
 * directly); each entry point below names the reference code whose job it takes over.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every function returns RT_OK (0) or
 * a negative rt_status and never throws; rt_last_error() gives the text for the calling context.
 * Host arrays are borrowed for the duration of the call only.  Calls on one context are serialised
 * by an internal mutex, so trace and upload may come from different threads
 * (RenderInterfaces.hpp:4-10, 14-20 require that of trace/load).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_NO_DEVICE = -2,
    RT_ERR_HIP = -3,
    RT_ERR_NO_SCENE = -4, /* "Scene is not loaded.", Renderer/TraceMethods.cpp:54-56 */
    RT_ERR_BAD_SCENE = -5,
    RT_ERR_OUT_OF_MEMORY = -6
} rt_status;

/* Node kinds carried in rt_node.flags.  0 and 1 are the reference's NODE_INTERNAL / NODE_LEAF
 * (CPURayTrace/SimplifiedData.hpp:30-36).  RT_NODE_MESH (the reference's unused NODE_MESH value) and
 * RT_NODE_SPHERE describe the pointer-based scene (CPURayTrace/BVH.hpp:6-12) once flattened. */
enum {
    RT_NODE_INTERNAL = 0, /* left/right = child node indices */
    RT_NODE_LEAF = 1,     /* left = triangle index */
    RT_NODE_MESH = 2,     /* left = root node of the mesh's inner BVH (CPURayTrace/Mesh.hpp:100) */
    RT_NODE_SPHERE = 4    /* left = sphere index */
};

/* Byte-compatible with sd::Node (SimplifiedData.hpp:71-77): 36 bytes, flags in the first byte after
 * the box; the three padding bytes are ignored. */
typedef struct rt_node {
    uint32_t left;
    uint32_t right;
    float pmin[3];
    float pmax[3];
    uint8_t flags;
    uint8_t pad_[3];
} rt_node;

/* Byte-compatible with sd::Triangle (SimplifiedData.hpp:82-88): 100 bytes. */
typedef struct rt_tri {
    float pos[3][3];
    float nrm[3][3];
    float uv[3][2];
    uint16_t mat; /* sd path: the Matierals enumerator; old path: index into the material table */
    uint16_t pad_;
} rt_tri;

/* Sphere of CPURayTrace/Objects.hpp:26-42. */
typedef struct rt_sphere {
    float center[3];
    float radius;
    uint32_t mat;     /* index into the material table */
    uint32_t prim_id; /* value reported in the primary-id image */
} rt_sphere;

enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_EMIT = 2, RT_MAT_DIELECTRIC = 4 };

/* Materials of CPURayTrace/Materials/{Lambertian,Metal,LightEmit}.hpp plus a dielectric. */
typedef struct rt_material {
    uint32_t kind;
    float albedo[4]; /* albedo, or intensity for RT_MAT_EMIT */
    float param;     /* Metal: gross; Dielectric: index of refraction */
} rt_material;

/* Field order of class Camera (Include/Camera.hpp:10-15). */
typedef struct rt_camera {
    float focal_length;
    float position[3];
    float look_at[3];
    float width;
    float height;
    float aspect_ratio;
} rt_camera;

enum {
    RT_EST_SD_LOOP = 0,      /* Trace::CastRay(ray, depth, sd::DataStorage&), CPURayTrace/Trace.cpp:58-96 */
    RT_EST_OLD_RECURSIVE = 1 /* Trace::CastRay(ray, depth, const Scene&), CPURayTrace/Trace.cpp:31-56 */
};
enum {
    RT_ORDER_RIGHT_FIRST = 0, /* sd::BVH::IntersectLoop pops the right child first, SimplifiedData.cpp:458-459 */
    RT_ORDER_LEFT_FIRST = 1   /* BVH::intersect / sd::BVH::Intersect recurse left first, BVH.hpp:143-144 */
};

/* How the closest hit of a ray is found.  Every choice returns, per ray, exactly the hit the reference
 * traversal returns (same primitive, same t, u, v bits).
 *   LANE / WAVE / NESTED perform the reference's own slab and primitive tests in the reference's visiting
 *   order over the reference's node boxes; they differ only in how node records are fetched.
 *   FAST derives the same result from the reference's LEAF boxes alone, walking a surface-area-heuristic
 *   hierarchy of exact unions the upload builds above them (see rt_upload_scene); it does far fewer tests
 *   when the reference's interior boxes prune badly.  The library picks how the walk is run: for scenes that
 *   fit the L2s FAST_SLICED (flat scene) or FAST_QUEUED (pointer scene), FAST_REFILL beyond; FAST_INLINE is the
 *   plain form.  FAST_REPLAY forces every ray through FAST's ordered replay procedure, which normally resolves
 *   only rays whose winning hit lies in front of its own leaf box's entry distance (a verification aid). */
enum {
    RT_TRAV_AUTO = 0,      /* = RT_TRAV_FAST; = RT_TRAV_REFERENCE for an uploaded tree in which some interior box does not
                              contain its children's boxes (the reference's builders never produce one; FAST is exact
                              only for nested boxes and is refused for such a tree with RT_ERR_BAD_SCENE) */
    RT_TRAV_LANE = 1,      /* reference order; every lane follows its own index; node records through the vector memory path */
    RT_TRAV_WAVE = 2,      /* reference order; the wave advances together; node records through the scalar data path */
    RT_TRAV_NESTED = 3,    /* reference order; first-generation kernel: one 8x8 tile per wave, nested sample/bounce loops */
    RT_TRAV_FAST = 4,
    RT_TRAV_FAST_REPLAY = 5,
    RT_TRAV_REFERENCE = 6, /* reference order; LANE or WAVE chosen from a surface-area estimate of the visits per ray */
    RT_TRAV_FAST_REFILL = 7, /* FAST with lane re-fill (a lane that finishes its walk takes the next ray instead of waiting
                               for its wave) and noted primitive tests; RT_TRAV_FAST itself uses it for scenes larger than
                               the L2s */
    RT_TRAV_BRUTE = 8,       /* "BVH Acceleration" unticked (BVHSettings::toggleBVHAccel, Program/BVHUI.hpp:14,25): no box is
                               tested, every primitive is -- Scene::intersectClosest (Program/Scene.hpp:83-95) and the
                               else-branch of Mesh::intersect (CPURayTrace/Mesh.hpp:161-172); for the flat scene, which the
                               reference never traces this way: all triangles in storage order.  A debugging aid: time grows
                               with rays x primitives */
    RT_TRAV_FAST_QUEUED = 9, /* FAST with the primitive tests taken out of the walk: lanes note them, the wave runs them
                               together, 64 at a time, once its walks are over; RT_TRAV_FAST itself for pointer scenes within the L2s */
    RT_TRAV_FAST_INLINE = 10, /* FAST with every primitive test run where the walk meets it */
    RT_TRAV_FAST_SLICED = 11  /* FAST in slices: a launch takes every ray a bounded number of records further and hands
                                 unfinished walks, with their state, to the next launch; primitive tests noted; RT_TRAV_FAST itself
                                 for flat scenes within the L2s */
};

/* How the devices of one context (rt_create_multi) share a trace call.  Both are the reference's own kind of split --
 * its 16 row bands share nothing either (Renderer/TraceMethods.cpp:61-72).
 *   TILE:   device r of n renders the 8x8 tiles k with k % n == r for all samples; the frame is put together by
 *           copying each device's tiles -- bit-identical to one device.
 *   SAMPLE: device r renders its share of the call's sample indices for every pixel into a running mean of its own;
 *           the frame is sum_r mean_r * (samples_r / samples) -- equal to one device's running mean to rounding.  The
 *           devices' means belong to the context, so a call either restarts them (first_sample = 1) or continues the
 *           context's own run at the same frame size (first_sample = samples so far + 1); anything else -- a mean
 *           accumulated elsewhere handed in through d_rgba, a resize without a restart -- is refused with
 *           RT_ERR_INVALID_ARGUMENT, where a single device would continue from the image it is given. */
enum { RT_SPLIT_TILE = 0, RT_SPLIT_SAMPLE = 1 };

/* Where the camera rays of the fast walks are made (results are the same bits either way).
 *   SEPARATE: a kernel of its own makes them and runs the walk's first slab test (the scene's root box); only rays that
 *             pass are queued for the path kernels, the others are sky and cost no queue or colour slot.
 *   FUSED:    the path kernel's first launch makes them and walks them in place.
 *   AUTO:     SEPARATE when the root box's projection covers less than half of the frame. */
enum { RT_CAMERA_AUTO = 0, RT_CAMERA_SEPARATE = 1, RT_CAMERA_FUSED = 2 };

typedef struct rt_options {
    int32_t bounce_limit; /* Trace::bounceLimit, CPURayTrace/Trace.hpp:16 */
    int32_t estimator;    /* RT_EST_* */
    uint32_t seed;        /* counter RNG seed (replaces Random::generator, Include/Random.hpp:12) */
    float jitter;         /* perturbStrength, Renderer/TraceMethods.cpp:48 */
    int32_t tile_rank;    /* this context renders 8x8 tiles k with k % tile_world == tile_rank */
    int32_t tile_world;   /* 1 = whole image */
    int32_t collect_stats; /* also count slab / triangle / sphere tests (slower kernel variant; it steps through the
                              exact 32-byte tree records, so the counts are those of the CPU restatement's walk, where
                              a trace without statistics may take the 16-byte grid records of a flat scene: more box
                              tests, the same hits).  2: count the walk a trace without statistics makes instead -- over
                              a flat scene's wide records every child box tested counts as a box test, as does every
                              exact leaf-box test; with the distance limit these are fewer than the exact walk's */
    uint32_t sample_offset; /* added to the sample index in the RNG key only: lets several contexts each
                               accumulate a mean over n samples of disjoint random streams (sample-range split) */
    int32_t traversal;      /* RT_TRAV_* */
    int32_t multi_split;    /* RT_SPLIT_*: how a context made by rt_create_multi shares a call among its devices */
    int32_t camera_stage;   /* RT_CAMERA_* */
} rt_options;

typedef struct rt_stats {
    uint64_t rays;      /* closest-hit traversals started, primary + bounces; every counter runs from the previous
                           rt_trace / rt_sync of the context to this one */
    uint64_t box_tests; /* filled when collect_stats != 0 */
    uint64_t tri_tests;
    uint64_t sph_tests;
    uint64_t tri_hits;  /* traversals that ended on a triangle */
    uint64_t replays;   /* RT_TRAV_FAST: rays resolved by the ordered replay (always counted) */
    float kernel_ms;     /* HIP-event time of the last trace kernel on the context's stream */
    float total_ms;      /* host wall time of the call (rt_trace only) */
    float kernel_ms_sum; /* HIP-event time summed over the calls since the previous sync; with pipelining on, the time
                            from the first call's start to the last call's end */
    uint32_t launches;   /* number of those launches */
} rt_stats;

/* Creates a context on HIP device `device`.  Stands in for the constructors of the reference's
 * trace/load method objects (Renderer/TraceMethods.cpp:44, Renderer/LoadMethods.cpp:54). */
int rt_create(rt_ctx **out, int device);
/* One context over several HIP devices (device_ids[0 .. n_devices-1], each at most once in a real installation): the
 * counterpart of the reference's numThreads = 16 row bands (Renderer/TraceMethods.hpp:44, TraceMethods.cpp:61-78) with
 * GPUs for threads.  The scene is replicated on every device; each rt_trace / rt_trace_device is shared as
 * rt_options.multi_split says, one host thread per device issuing and awaiting that device's work, and the frame is put
 * together on device_ids[0] (peer copies over xGMI), where d_rgba of rt_trace_device must live.  rt_options.tile_rank /
 * tile_world / sample_offset keep their meaning for the context as a whole, so processes and devices can be combined.
 * rt_trace_device returns when the devices have finished their shares (the assembly is enqueued on the context's
 * stream); rt_set_pipelining(1) and rt_trace_device_noalloc are refused.  rt_trace_begin / rt_trace_end work with the
 * tile split: every device keeps its own share of the two calls in flight, and the first device's copy stream puts each
 * frame together as soon as the shares are complete.  n_devices == 1 is rt_create. */
int rt_create_multi(rt_ctx **out, const int *device_ids, int n_devices);
int rt_device_count(rt_ctx *ctx); /* devices behind the context */
void rt_destroy(rt_ctx *ctx);
/* Text of the context's last error.  The pointer belongs to the calling thread and stays valid until that thread's next
 * rt_last_error call, whatever other threads do with the context. */
const char *rt_last_error(rt_ctx *ctx);

/* Takes the scene snapshot the reference's load methods take (LoadSdSceneCPU::load,
 * Renderer/LoadMethods.cpp:55-61; LoadSceneCPU::load, :65-73) and lays it out in HBM.  `nodes`/`tris`
 * may point straight at sd::NodeStorage::nodes / sd::TriangleStorage::triangles.  `order` says which
 * child the reference traversal visits first (RT_ORDER_*).  Two node arrays are built: the reference's
 * own tree re-laid in its visiting order (RT_TRAV_LANE/WAVE/NESTED), and for RT_TRAV_FAST a second
 * hierarchy over the same leaf boxes (kept bit for bit) whose interior boxes are exact unions. */
int rt_upload_scene(rt_ctx *ctx, const rt_node *nodes, uint32_t n_nodes, uint32_t root, const rt_tri *tris,
                    uint32_t n_tris, const rt_sphere *spheres, uint32_t n_spheres, const rt_material *mats,
                    uint32_t n_mats, int order);

/* The camera the reference reads live each frame (DIContext.cam, Renderer/TraceMethods.cpp:51-53). */
int rt_set_camera(rt_ctx *ctx, const rt_camera *cam);
int rt_set_options(rt_ctx *ctx, const rt_options *opt);
/* Allocates now everything later trace calls of up to n_samples samples per pixel on width x height frames need (the
 * wavefront workspace of the context's internal pipes and the context's own image), so that those calls neither
 * allocate nor wait for an allocation; without it the first call of a size does both.  The reference's counterpart is
 * CPUImageData::resize (Renderer/TraceMethods.hpp:31-36), which sizes the trace method's buffers ahead of the loop.
 * Footprint per device: 2 pipes x (S*W*H*16 B of colours + 2 queues of (S*items + 532 480) records of 96 B), where
 * items = this context's pixels in whole 8x8 tiles and S = min(n_samples, 64, 192 Mi / items) samples per chunk --
 * 4.7 GB at 1920x1080x16, 28 GB at 3840x2160 with 23 samples per chunk.  Buffers only grow; rt_destroy releases them. */
int rt_reserve(rt_ctx *ctx, int width, int height, int n_samples);

/* Zeroes the accumulated image (TracerAsync::resetSamples, Renderer/TracerImpl.cpp:18-22, together
 * with CPUImageData::resize's zero fill, Renderer/TraceMethods.hpp:31-36). */
int rt_reset(rt_ctx *ctx);

/* TraceSdSceneCPU::trace / TraceSceneCPU::trace (Renderer/TraceMethods.cpp:45-80, :84-119): renders
 * samples first_sample .. first_sample+n_samples-1 (first_sample is the reference's 1-based
 * sampleCount) for every pixel into the context's running-mean image and copies the W*H*4 float mean
 * to out_rgba_mean (host, row 0 = bottom row, may be NULL).  out_primary_id (host, W*H, may be NULL)
 * receives the primitive hit by sample first_sample's camera ray, 0xFFFFFFFF for a miss.  The image
 * is re-zeroed when width/height change. */
int rt_trace(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *out_rgba_mean,
             uint32_t *out_primary_id, rt_stats *stats);

/* rt_trace without the wait, for callers that can have the next frame computed while they use this one -- the
 * reference's frame loop asks for sample n+1 right after sample n unless the camera or the scene changed
 * (TracerAsync::render, Renderer/TracerImpl.cpp:9-12; Renderer::render resets on RenderState::Dirty, Renderer.cpp:86-90).
 * rt_trace_begin enqueues the samples and the copy of the resulting mean to out_rgba_mean (page-locked memory from
 * rt_alloc_host lets the copy overlap the next call's kernels) and returns; at most two begun calls may be in flight.
 * rt_trace_end waits for the oldest one: its image is in its out_rgba_mean, and the context's running mean is now the
 * mean after that call.  A begun call continues from the mean of the call begun before it (or of the last ended call)
 * and leaves the last ENDED call's mean untouched, so rt_trace_discard -- or any rt_trace, rt_reset, rt_upload_scene --
 * drops what is in flight as if it had never been begun.  The bits are rt_trace's.  stats (may be NULL): counters as
 * accumulated since the previous rt_sync / rt_trace, which may include part of a call begun later; no kernel times.
 * On a context over several devices: with RT_SPLIT_TILE only (RT_ERR_INVALID_ARGUMENT otherwise). */
int rt_trace_begin(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *out_rgba_mean);
int rt_trace_end(rt_ctx *ctx, rt_stats *stats);
int rt_trace_discard(rt_ctx *ctx);

/* Same work, asynchronous, into caller-owned DEVICE memory: d_rgba (W*H*4 floats, holds the running
 * mean before and after; the caller zeroes it for first_sample == 1) and d_primary_id (W*H, may be
 * NULL).  Enqueued on the context's stream; nothing is copied to the host. */
int rt_trace_device(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *d_rgba,
                    uint32_t *d_primary_id);
/* rt_trace_device for callers that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference.  Same work and same results as rt_trace_device, but the
 * call makes no allocation of any kind (no hipMalloc / hipFree, no event or stream creation): every launch goes to the
 * context's stream, one chunk of samples after the other, in a workspace that is part of the library's code object --
 * one static array of 159 MB (159 391 744 bytes) per GPU the library runs on, held until the process ends.  Limit: this context's
 * share of the frame, counted in whole 8x8 tiles, must not exceed rt_noalloc_max_pixels() = 262 144 pixels (the
 * reference's 640x360 window is 230 400); larger frames are refused with RT_ERR_INVALID_ARGUMENT.  Every context on a
 * GPU shares that GPU's workspace: calls that overlap on one GPU (two contexts, or one context whose stream was changed
 * while work was pending) race.  Kernel times are not recorded for these calls (rt_stats.launches does not count them);
 * the ray counters are. */
int rt_trace_device_noalloc(rt_ctx *ctx, int width, int height, int first_sample, int n_samples, float *d_rgba,
                            uint32_t *d_primary_id);
uint64_t rt_noalloc_max_pixels(void);
/* Lets consecutive rt_trace_device calls overlap (off by default).  Internally the tracer works in chunks on two
 * streams of its own so that the latency-bound end of one chunk overlaps the start of the next; by default every call
 * ends by making the context's stream wait for its chunks, which keeps the ordering promise above but lets nothing
 * overlap ACROSS calls.  With pipelining on, a call returns with its chunks ordered only among themselves (the folds
 * into the running mean stay in sample order); the caller orders the results with rt_join -- the context's stream
 * then waits for everything issued so far (a stream-level wait: the host does not block) -- or with rt_sync.  This is
 * for callers that issue many passes back to back (progressive rendering, bench.py); no reference counterpart. */
int rt_set_pipelining(rt_ctx *ctx, int on);
int rt_join(rt_ctx *ctx);
/* Replaces the context's stream with a caller-owned hipStream_t (NULL = the default stream). */
int rt_set_stream(rt_ctx *ctx, void *hip_stream);
/* Waits for the context's stream and reports the counters and kernel time of the last trace. */
int rt_sync(rt_ctx *ctx, rt_stats *stats);

/* Host-only view of what rt_upload_scene lays out for the device, for inspection and tests; no HIP call is made.
 * It is this build's counterpart of the reference's own flattening for its GL tracer (LoadSdSceneGPU::load,
 * Renderer/LoadMethods.cpp:16-51, via sd::ConvertNodeToFlatStorage, CPURayTrace/SimplifiedData.cpp:162-185).
 * Node records are 8 floats: pMin.xyz, pMax.xyz, skip (uint32 bits: byte offset of the first record after the
 * subtree), word (uint32 bits).  Fast tree: word = byte offset of the right child for an interior node (low three
 * bits 0), (slot << 3) | kind for a triangle / sphere leaf, (reference position << 3) | RT_NODE_MESH for a mesh
 * leaf, whose own tree follows inline.  Reference-order tree: word = (slot << 3) | kind, and one terminator record
 * follows the last node.  tri_pos: 12 floats per slot = p0, e1, e2, triangle index, material, reference position
 * of the leaf (last three as uint32 bits), in fast-tree leaf order.  spheres: 8 floats per slot = centre, radius,
 * material, primitive id, dielectric flag, reference position.
 * Flat scenes (triangles only, every one a leaf of the scene's tree) also get the grid records that traces without
 * statistics walk, one per fast-tree record and in the same order: four uint32 = x planes (low | high << 16), y planes,
 * z planes -- plane q of an axis lies at grid_origin + q * grid_step and the six planes enclose the record's box with
 * at least half a step to spare --, and a word: byte offset (16 per record) of the first record after the subtree for
 * an interior record, (slot << 4) | 1 for a leaf.  leaf_records: 16 floats per triangle slot = p0, e1, e2 (nine
 * floats), pMin.x, pMin.y, reference position of the leaf (uint32 bits), pMin.z, pMax.xyz: the leaf's own box as in
 * fast_nodes, bit for bit.  n_grid_nodes is 0 for any other scene.
 * Flat scenes whose hierarchy is at most 16 levels deep once collapsed also get the WIDE form of the fast tree, the one
 * the default walk of such scenes steps through: 64-byte nodes of up to eight children (a step of the walk is one trip to
 * memory for eight slab tests).  wide_nodes, 16 uint32 per node, node 0 the root: words 0..11 the children's boxes, 8 bits
 * per plane, as six arrays of eight bytes (lo x, hi x, lo y, hi y, lo z, hi z; child i is byte i); plane value p of axis k
 * lies at grid unit org_k + (p << shift_k) of the grid above and the planes enclose the child's box with at least a unit
 * and a half to spare; word 12 org_x | org_y << 16; word 13 org_z | shift_x << 16 | shift_y << 20 | shift_z << 24 |
 * axis << 28 | class << 30 (the children are sorted along that axis, interior ones first, then the leaves; class = the
 * largest leaf_class below the node); word 14 the index of the
 * first interior child (interior children are consecutive nodes); word 15 first leaf slot | interior children << 24 |
 * children << 28.  A node's leaves occupy consecutive triangle slots, and in a scene that has a wide tree the slots of
 * tri_pos / leaf_records are numbered in that tree's order instead of the fast tree's leaf order.
 * Distance limit: with tau the smallest hit distance a ray has found so far and D the largest component of its
 * direction (0.25 <= D <= 1.25), the walk does not enter a child of a node of class k < 3 whose box the ray meets beyond
 * (tau + c1[k] / D + (c2 W + c3)[k]) * (1 + 2 c2[k] D) * 1.00001 -- a margin that covers, with a factor of two to spare,
 * how far in front of its own leaf box the reference's fp32 triangle test (SimplifiedData.cpp:269-307) can report a hit
 * on a triangle of that size class, so that no leaf that could be the reference's answer is skipped. */
typedef struct rt_layout_view {
    uint32_t n_fast_nodes, n_ref_nodes, n_tri_slots, n_sphere_slots;
    const float *fast_nodes;
    const float *ref_nodes;
    const float *tri_pos;
    const float *spheres;
    uint32_t n_grid_nodes;
    const uint32_t *grid_nodes;
    const float *leaf_records;
    float grid_origin[3], grid_step[3];
    uint32_t n_wide_nodes;       /* 0 when the scene has no wide tree */
    const uint32_t *wide_nodes;  /* 16 uint32 per node, see below */
    uint32_t has_distance_limit; /* the wide walk skips subtrees beyond a limit derived from its closest hit so far */
    const uint8_t *leaf_class;   /* per triangle slot: 0..2 size class of a regular leaf, 3 = never skipped */
    float limit_constants[9];    /* by class: c1[3], 2 c2[3], (c2 W + c3)[3]; see below */
    void *owner_;
} rt_layout_view;
int rt_layout_scene(const rt_node *nodes, uint32_t n_nodes, uint32_t root, const rt_tri *tris, uint32_t n_tris,
                    const rt_sphere *spheres, uint32_t n_spheres, const rt_material *mats, uint32_t n_mats, int order,
                    rt_layout_view *out);
void rt_layout_free(rt_layout_view *view);

/* Page-locked host memory for the buffers handed to rt_trace: the copy of the W*H*4 float mean that stands in for
 * Texture2D::setData's upload (GLShader/Texture.cpp:86-91) then runs at PCIe rate instead of through a pageable
 * bounce buffer.  rt_alloc_host returns NULL on failure; any host pointer works with rt_trace, these are only faster. */
void *rt_alloc_host(uint64_t bytes);
void rt_free_host(void *ptr);

/* Self-test hook: the kernels' own fp32 vector helpers (one source for host and device, csrc/rt_device_common.hpp)
 * evaluated on the host, so that their operation order -- glm's scalar definitions, which every value of the reference
 * goes through (CPURayTrace/SimplifiedData.cpp:274-293, Include/Camera.hpp:35-46) -- can be pinned without a GPU.
 * op 0 dot(a, b) -> out3[0]; 1 cross(a, b); 2 normalize(a); 3 mat3(a, b, c) * v.  Unused pointers may be NULL. */
void rt_vec_op(int op, const float *a, const float *b, const float *c, const float *v, float *out3);

/* Library self-description: "rt_hip <version> gfx950". */
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif

/*
 * rt_oracle.h -- C entry points of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain CPU restatement of the
 * reference renderer's CPURayTrace path.  It exists so that tests, the smoke
 * check and the benchmark's cpu_baseline leg can compare the HIP path against
 * it.  Nothing in the shipped product (raytracing_amd/, include/) may link,
 * load or call it.
 *
 * PARITY UNPINNED: the reference holds no tests, golden images or known-answer
 * vectors for this path and cannot be built in this environment (glm, assimp,
 * glfw, glad, <format>, <Windows.h> are absent), so the oracle is pinned only
 * by its own hand-computed known-answer tests and self-consistency checks
 * (recursive vs loop traversal, BVH vs brute force, flat pack/unpack).  The one
 * literal data dump in the reference (GLSL/GPURayTrace/testScene.glsl:4-16) is
 * used as a layout check.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_camera {
    float focal_length; /* Include/Camera.hpp:10 */
    float position[3];  /* :11 */
    float look_at[3];   /* :12 */
    float width;        /* :13 */
    float height;       /* :14 (= width / aspect, Camera.hpp:26) */
} ora_camera;

enum { ORA_TRIG_LIBM = 0, ORA_TRIG_POLY = 1 };
enum { ORA_TRAV_LOOP = 0, ORA_TRAV_RECURSIVE = 1, ORA_TRAV_BRUTE = 2,
       ORA_TRAV_LEAVES = 3, /* flat scene only: IntersectLoop's result derived from the leaves alone, visited in an
                              unrelated order (the argument the product's fast walk rests on) */
       ORA_TRAV_LIMITED = 4 /* ... the same, nearest leaf box first, skipping the leaves beyond a distance limit derived from
                              the closest hit so far (ora_sd_set_limit; the argument the product's wide walk rests on) */ };
enum { ORA_ACCUM_RECURSIVE = 0, ORA_ACCUM_FORWARD = 1 };

typedef struct ora_options {
    int32_t bounce_limit;   /* Trace::bounceLimit, CPURayTrace/Trace.hpp:16 */
    uint32_t seed;          /* counter RNG seed */
    float jitter;           /* perturbStrength, Renderer/TraceMethods.cpp:48 */
    int32_t trig;           /* ORA_TRIG_* : how cos/sin of Random.hpp:52-53 are evaluated */
    int32_t traversal;      /* ORA_TRAV_* */
    int32_t threads;        /* row bands, numThreads of Renderer/TraceMethods.hpp:44 */
    int32_t lock_per_pixel; /* take the shared_lock of TraceMethods.cpp:57 per pixel */
    int32_t accum;          /* ORA_ACCUM_* (old-scene path only) */
    int32_t tile_rank;      /* render only 8x8 tiles k with k % tile_world == tile_rank ... */
    int32_t tile_world;     /* ... (0 or 1 = whole image); mirrors rt_options for the partition tests */
    uint32_t sample_offset; /* added to the sample index in the RNG key only */
} ora_options;

typedef struct ora_stats {
    uint64_t rays;      /* closest-hit traversals started (primary + bounces) */
    uint64_t box_tests; /* slab tests executed */
    uint64_t tri_tests; /* Moller-Trumbore tests executed */
    uint64_t sph_tests; /* sphere tests executed */
    uint64_t tri_hits;  /* traversals whose final closest hit is a triangle */
    double seconds;     /* wall time of the render call */
} ora_stats;

typedef struct ora_hit {
    int32_t hit;
    float t;
    float pos[3];
    float normal[3];
    uint32_t prim; /* triangle index (node.left of the leaf) or 0xFFFFFFFF */
    uint32_t mat;
    uint32_t box_tests;
    uint32_t tri_tests;
} ora_hit;

/* ---- flat-array ("sd") scene: CPURayTrace/SimplifiedData.{hpp,cpp} ---- */
void *ora_sd_create(void);
void ora_sd_destroy(void *scene);
/* One model file = n_meshes meshes; tris24 holds 3 x (pos3, nrm3, uv2) floats per
 * triangle, mesh_sizes[m] triangles for mesh m.  Follows
 * sd::ModelLoader::LoadModelFileSync (Include/ModelLoader.hpp:260-273). */
int ora_sd_add_model(void *scene, const float *tris24, const uint32_t *mesh_sizes, uint32_t n_meshes);
/* Scene-level BVH over the model roots (Program/Scene.cpp:205-206). */
int ora_sd_finalize(void *scene);
void ora_sd_counts(void *scene, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *root);
/* 36-byte records {u32 left, u32 right, f32 pMin[3], f32 pMax[3], u32 flags}. */
void ora_sd_get_nodes(void *scene, void *out36);
/* 100-byte records {f32 pos[9], f32 nrm[9], f32 uv[6], u32 mat}. */
void ora_sd_get_tris(void *scene, void *out100);
/* Flat packers of SimplifiedData.cpp:162-223 (9 floats/node, 25 floats/tri). */
void ora_sd_flat_nodes(void *scene, float *out9);
void ora_sd_flat_tris(void *scene, float *out25);
/* Unpackers of SimplifiedData.cpp:224-254; tri unpack keeps the value-cast quirk of :252. */
void ora_flat_node_unpack(const float *flat9, uint32_t index, void *out36);
void ora_flat_tri_unpack(const float *flat25, uint32_t index, void *out100);
void ora_sd_intersect(void *scene, const float *origin, const float *dir, int traversal, ora_hit *out);
/* Rays ORA_TRAV_LEAVES had to resolve by the ordered replay since the last call of this function. */
uint64_t ora_take_leaf_replays(void);
/* The distance limit of ORA_TRAV_LIMITED: c9 = c1[3], 2 c2[3], (c2 W + c3)[3] by size class, and the class of every
 * triangle (0..2, 3 = never skipped).  A leaf of class k < 3 is skipped when the ray enters its box at or beyond
 * (tau + c1[k] / D + c3[k]) * (1 + 2 c2[k] D) * 1.00001, tau the closest hit so far, D = |d|_inf in [0.25, 1.25]. */
int ora_sd_set_limit(void *scene, const float *c9, const uint8_t *class_by_triangle, uint32_t n_triangles);
/* Leaves ORA_TRAV_LIMITED skipped / leaves whose box a ray met, since the last call. */
void ora_take_limit_counts(uint64_t *skipped, uint64_t *met);
/* Renders samples first_sample .. first_sample+n_samples-1 (1-based sampleCount of
 * Renderer/TracerImpl.cpp:11) as a running mean into image (W*H*4 floats, row 0 = bottom). */
int ora_sd_render(void *scene, const ora_camera *cam, const ora_options *opt, int width, int height,
                  int first_sample, int n_samples, float *image, uint32_t *primary_id, ora_stats *stats);

/* ---- pointer-based ("old") scene: Objects.hpp, Mesh.hpp, BVH.hpp, Materials ---- */
enum { ORA_MAT_LAMBERTIAN = 0, ORA_MAT_METAL = 1, ORA_MAT_EMIT = 2, ORA_MAT_DIELECTRIC = 4 };
void *ora_old_create(void);
void ora_old_destroy(void *scene);
int ora_old_add_sphere(void *scene, const float *center, float radius, uint32_t kind, const float *albedo4,
                       float param);
int ora_old_add_mesh(void *scene, const float *tris24, uint32_t n_tris, uint32_t kind, const float *albedo4,
                     float param);
int ora_old_build(void *scene);
void ora_old_intersect(void *scene, const float *origin, const float *dir, int traversal, ora_hit *out);
/* Scene BVH in the order BVH::intersect visits it (node, left subtree, right subtree; a mesh leaf is
 * followed by its inner BVH): 8 floats per node {pMin, pMax, kind bits, ref bits}, kind 0 internal,
 * 1 triangle (ref = triangle id), 2 mesh, 4 sphere (ref = sphere index).  Returns the node count;
 * out may be NULL to query it. */
uint32_t ora_old_flatten(void *scene, float *out8);
int ora_old_render(void *scene, const ora_camera *cam, const ora_options *opt, int width, int height,
                   int first_sample, int n_samples, float *image, uint32_t *primary_id, ora_stats *stats);

/* helpers exposed for known-answer tests */
int ora_box_test(const float *pmin, const float *pmax, const float *origin, const float *dir, float tmin,
                 float tmax);
uint32_t ora_rng_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);
void ora_sincos(float theta, int trig, float *s, float *c);
void ora_camera_dir(const ora_camera *cam, float u, float v, float *out3);
/* the oracle's own vector helpers (glm's scalar definitions), for known-answer tests of their operation order:
 * op 0 dot(a, b) -> out3[0]; 1 cross(a, b); 2 normalize(a); 3 mat3(a, b, c) * v; 4 / 5 component-wise min / max(a, b) */
void ora_vec_op(int op, const float *a, const float *b, const float *c, const float *v, float *out3);
void ora_cosine_dir(const float *normal, float u1, float u2, int trig, float *out3);

#ifdef __cplusplus
}
#endif
#endif

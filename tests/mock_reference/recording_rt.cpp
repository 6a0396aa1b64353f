// recording_rt.cpp -- test fixture: the entry points of include/rt_hip.h that the render modes call, recording their
// arguments (one line each on the file named by RT_RECORD) instead of using a GPU.  rt_trace_end fills the frame with the
// sample number so the caller's hand-over of the image can be checked.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>

#include "rt_hip.h"

struct rt_ctx {
    std::deque<std::pair<int, float *>> flights; // sample, image
    int w = 0, h = 0;
};
static FILE *rec() {
    static FILE *f = fopen(getenv("RT_RECORD") ? getenv("RT_RECORD") : "/dev/null", "w");
    return f;
}
static unsigned long long fnv(const void *p, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++)
        h = (h ^ ((const unsigned char *)p)[i]) * 1099511628211ull;
    return h;
}
extern "C" {
int rt_create(rt_ctx **out, int device) {
    *out = new rt_ctx;
    fprintf(rec(), "create device=%d\n", device);
    return RT_OK;
}
int rt_create_multi(rt_ctx **out, const int *, int n) {
    *out = new rt_ctx;
    fprintf(rec(), "create_multi n=%d\n", n);
    return RT_OK;
}
int rt_device_count(rt_ctx *) { return 1; }
void rt_destroy(rt_ctx *c) {
    fprintf(rec(), "destroy\n");
    fflush(rec());
    delete c;
}
const char *rt_last_error(rt_ctx *) { return "recording stand-in"; }
int rt_upload_scene(rt_ctx *, const rt_node *nodes, uint32_t n_nodes, uint32_t root, const rt_tri *tris, uint32_t n_tris, const rt_sphere *sph,
                    uint32_t n_sph, const rt_material *mat, uint32_t n_mat, int order) {
    // the padding bytes of a record are not the caller's to define: hash the fields
    unsigned long long hn = 1469598103934665603ull, ht = hn;
    for (uint32_t i = 0; i < n_nodes; i++)
        hn = hn * 31ull + fnv(&nodes[i], 33);
    for (uint32_t i = 0; i < n_tris; i++)
        ht = ht * 31ull + fnv(&tris[i], 98);
    fprintf(rec(), "upload n_nodes=%u root=%u n_tris=%u nodes_hash=%llu tris_hash=%llu spheres=%d n_spheres=%u materials=%d n_materials=%u order=%d\n",
            n_nodes, root, n_tris, hn, ht, sph != nullptr, n_sph, mat != nullptr, n_mat, order);
    return RT_OK;
}
int rt_set_camera(rt_ctx *, const rt_camera *c) {
    fprintf(rec(), "camera focal=%.9g pos=%.9g,%.9g,%.9g look=%.9g,%.9g,%.9g width=%.9g height=%.9g aspect=%.9g\n", c->focal_length, c->position[0],
            c->position[1], c->position[2], c->look_at[0], c->look_at[1], c->look_at[2], c->width, c->height, c->aspect_ratio);
    return RT_OK;
}
int rt_set_options(rt_ctx *, const rt_options *o) {
    fprintf(rec(), "options bounce_limit=%d estimator=%d seed=%u jitter=%.9g tile_world=%d traversal=%d\n", o->bounce_limit, o->estimator, o->seed,
            o->jitter, o->tile_world, o->traversal);
    return RT_OK;
}
int rt_trace_begin(rt_ctx *c, int w, int h, int first, int n, float *image) {
    fprintf(rec(), "trace_begin w=%d h=%d first=%d n=%d\n", w, h, first, n);
    c->w = w;
    c->h = h;
    c->flights.emplace_back(first, image);
    return RT_OK;
}
int rt_trace_end(rt_ctx *c, rt_stats *st) {
    if (c->flights.empty())
        return RT_ERR_INVALID_ARGUMENT;
    auto fl = c->flights.front();
    c->flights.pop_front();
    for (size_t i = 0; i < (size_t)c->w * c->h * 4; i++)
        fl.second[i] = (float)fl.first;
    if (st)
        memset(st, 0, sizeof(*st));
    fprintf(rec(), "trace_end sample=%d\n", fl.first);
    return RT_OK;
}
int rt_trace_discard(rt_ctx *c) {
    fprintf(rec(), "trace_discard dropped=%zu\n", c->flights.size());
    c->flights.clear();
    return RT_OK;
}
int rt_trace(rt_ctx *, int w, int h, int first, int n, float *image, uint32_t *, rt_stats *) {
    fprintf(rec(), "trace w=%d h=%d first=%d n=%d\n", w, h, first, n);
    for (size_t i = 0; i < (size_t)w * h * 4; i++)
        image[i] = (float)first;
    return RT_OK;
}
void *rt_alloc_host(uint64_t bytes) { return malloc(bytes); }
void rt_free_host(void *p) { free(p); }
}

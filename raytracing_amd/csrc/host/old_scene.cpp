#include "old_scene.hpp"

#include <algorithm>
#include <cstring>

namespace rthost {

namespace {
inline float lo(float a, float b) { return (b < a) ? b : a; } // glm::min
inline float hi(float a, float b) { return (a < b) ? b : a; } // glm::max

rt_node blankNode() {
    rt_node n;
    std::memset(&n, 0, sizeof(n)); // a value-initialised BVHNode: box = 0 (BVH.hpp:42)
    return n;
}
} // namespace

void OldScene::addSphere(const float center[3], float radius, const rt_material &material) {
    rt_sphere s;
    std::memcpy(s.center, center, sizeof(s.center));
    s.radius = radius;
    s.mat = (uint32_t)materials.size();
    s.prim_id = 0x40000000u | (uint32_t)spheres.size();
    materials.push_back(material);
    Object o;
    o.kind = RT_NODE_SPHERE;
    o.ref = (uint32_t)spheres.size();
    for (int k = 0; k < 3; k++) { // Objects.hpp:40-41
        o.pmin[k] = center[k] - radius;
        o.pmax[k] = center[k] + radius;
    }
    spheres.push_back(s);
    objects.push_back(o);
}

void OldScene::addMesh(const MeshData &mesh, const rt_material &material) {
    const uint32_t matIndex = (uint32_t)materials.size();
    materials.push_back(material);
    std::vector<Object> tris;
    for (size_t i = 0; i + 2 < mesh.indices.size(); i += 3) {
        rt_tri t;
        std::memset(&t, 0, sizeof(t));
        for (int c = 0; c < 3; c++) {
            const SdVertex &v = mesh.vertices.at(mesh.indices[i + c]);
            std::memcpy(t.pos[c], v.position, sizeof(v.position));
            std::memcpy(t.nrm[c], v.normal, sizeof(v.normal));
            std::memcpy(t.uv[c], v.texCoord, sizeof(v.texCoord));
        }
        t.mat = (uint16_t)matIndex;
        Object o;
        o.kind = RT_NODE_LEAF;
        o.ref = (uint32_t)triangles.size();
        for (int k = 0; k < 3; k++) { // Triangle::Triangle, Mesh.hpp:32-36: corner min/max grown by 1e-6
            o.pmin[k] = lo(t.pos[0][k], lo(t.pos[1][k], t.pos[2][k])) - 1e-6f;
            o.pmax[k] = hi(t.pos[0][k], hi(t.pos[1][k], t.pos[2][k])) + 1e-6f;
        }
        triangles.push_back(t);
        tris.push_back(o);
    }
    if (tris.empty())
        return;
    Object m;
    m.kind = RT_NODE_MESH;
    std::memcpy(m.pmin, tris[0].pmin, sizeof(m.pmin)); // Mesh.hpp:111-119: union starting at triangle 0
    std::memcpy(m.pmax, tris[0].pmax, sizeof(m.pmax));
    for (size_t i = 1; i < tris.size(); i++)
        for (int k = 0; k < 3; k++) {
            m.pmin[k] = lo(m.pmin[k], tris[i].pmin[k]);
            m.pmax[k] = hi(m.pmax[k], tris[i].pmax[k]);
        }
    m.ref = buildRange(tris, 0, (int)tris.size()); // insideBVH.build(hittablePtrs), Mesh.hpp:132
    objects.push_back(m);
}

// BVH::BuildBVH, CPURayTrace/BVH.hpp:20-71
uint32_t OldScene::buildRange(std::vector<Object> &objs, int start, int end) {
    const int count = end - start;
    if (count == 1) {
        rt_node leaf = blankNode();
        const Object &o = objs[start];
        leaf.left = leaf.right = o.ref;
        std::memcpy(leaf.pmin, o.pmin, sizeof(leaf.pmin));
        std::memcpy(leaf.pmax, o.pmax, sizeof(leaf.pmax));
        leaf.flags = (uint8_t)o.kind;
        nodes.push_back(leaf);
        return (uint32_t)nodes.size() - 1;
    }
    rt_node node = blankNode();
    node.flags = RT_NODE_INTERNAL;
    if (count == 2) {
        node.left = buildRange(objs, start, start + 1);
        node.right = buildRange(objs, start + 1, end);
        for (int k = 0; k < 3; k++) {
            node.pmin[k] = lo(nodes[node.left].pmin[k], nodes[node.right].pmin[k]);
            node.pmax[k] = hi(nodes[node.left].pmax[k], nodes[node.right].pmax[k]);
        }
        nodes.push_back(node);
        return (uint32_t)nodes.size() - 1;
    }
    // The union of three or more boxes starts from the zeroed box, so it always contains the origin.
    for (int i = start; i < end; i++)
        for (int k = 0; k < 3; k++) {
            node.pmin[k] = lo(node.pmin[k], objs[i].pmin[k]);
            node.pmax[k] = hi(node.pmax[k], objs[i].pmax[k]);
        }
    const float ex = node.pmax[0] - node.pmin[0], ey = node.pmax[1] - node.pmin[1], ez = node.pmax[2] - node.pmin[2];
    int axis = 2;
    if (ex >= ey && ex >= ez)
        axis = 0;
    else if (ey >= ex && ey >= ez)
        axis = 1;
    std::sort(objs.begin() + start, objs.begin() + end,
              [axis](const Object &a, const Object &b) { return a.pmin[axis] + a.pmax[axis] < b.pmin[axis] + b.pmax[axis]; });
    const int mid = start + count / 2;
    node.left = buildRange(objs, start, mid);
    node.right = buildRange(objs, mid, end);
    nodes.push_back(node);
    return (uint32_t)nodes.size() - 1;
}

void OldScene::build() {
    // inner-BVH nodes of the meshes stay; scene-level nodes from an earlier build() are not reused
    rootIndex = objects.empty() ? 0xFFFFFFFFu : buildRange(objects, 0, (int)objects.size());
}

} // namespace rthost

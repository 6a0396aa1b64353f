#include "sd_scene.hpp"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <stdexcept>

#include "obj_reader.hpp"

namespace rthost {

namespace {

// glm::min / glm::max select rules: min(a,b) = (b<a)?b:a, max(a,b) = (a<b)?b:a
inline float lo(float a, float b) { return (b < a) ? b : a; }
inline float hi(float a, float b) { return (a < b) ? b : a; }

inline void growBox(float mn[3], float mx[3], const rt_node &child) {
    for (int k = 0; k < 3; k++) {
        mn[k] = lo(mn[k], child.pmin[k]);
        mx[k] = hi(mx[k], child.pmax[k]);
    }
}

struct CentroidLess {
    const rt_node *nodes;
    int axis;
    bool operator()(const uint32_t &a, const uint32_t &b) const {
        return nodes[a].pmin[axis] + nodes[a].pmax[axis] < nodes[b].pmin[axis] + nodes[b].pmax[axis];
    }
};

rt_node makeInternal(uint32_t left, uint32_t right, const float mn[3], const float mx[3]) {
    rt_node n;
    std::memset(&n, 0, sizeof(n));
    n.left = left;
    n.right = right;
    for (int k = 0; k < 3; k++) {
        n.pmin[k] = mn[k];
        n.pmax[k] = mx[k];
    }
    n.flags = RT_NODE_INTERNAL;
    return n;
}

} // namespace

uint32_t SdDataStorage::addTriangle(const rt_tri &t) {
    triangles.push_back(t);
    return (uint32_t)triangles.size() - 1;
}

uint32_t SdDataStorage::addNode(const rt_node &n) {
    nodes.push_back(n);
    return (uint32_t)nodes.size() - 1;
}

uint32_t BuildBVHFromNodes(SdDataStorage &storage, uint32_t *nodeIndices, size_t start, size_t end) {
    if (end <= start)
        throw std::runtime_error("Build Failed. end - start <= 0 ");
    const size_t count = end - start;
    if (count == 1)
        return nodeIndices[start];
    float mn[3], mx[3];
    if (count == 2) {
        // two children: plain union of the two boxes (SimplifiedData.cpp:332-349)
        const uint32_t l = nodeIndices[start], r = nodeIndices[start + 1];
        for (int k = 0; k < 3; k++) {
            mn[k] = lo(storage.nodes[l].pmin[k], storage.nodes[r].pmin[k]);
            mx[k] = hi(storage.nodes[l].pmax[k], storage.nodes[r].pmax[k]);
        }
        return storage.addNode(makeInternal(l, r, mn, mx));
    }
    // three or more: the union starts from the reference's default box {FLT_MAX, FLT_MIN}
    // (SimplifiedData.hpp:64-69).  FLT_MIN is the smallest POSITIVE float, so an all-negative extent
    // is clamped up to ~0; the quirk is kept because it decides split axes and box sizes.
    for (int k = 0; k < 3; k++) {
        mn[k] = FLT_MAX;
        mx[k] = FLT_MIN;
    }
    for (uint32_t i = (uint32_t)start; i < end; i++)
        growBox(mn, mx, storage.nodes[nodeIndices[i]]);
    const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    int axis = 2;
    if (ex >= ey && ex >= ez)
        axis = 0;
    else if (ey >= ex && ey >= ez)
        axis = 1;
    std::sort(nodeIndices + start, nodeIndices + end, CentroidLess{storage.nodes.data(), axis});
    const size_t mid = start + count / 2;
    const uint32_t l = BuildBVHFromNodes(storage, nodeIndices, start, mid);
    const uint32_t r = BuildBVHFromNodes(storage, nodeIndices, mid, end);
    return storage.addNode(makeInternal(l, r, mn, mx));
}

uint32_t AddMesh(SdDataStorage &storage, const MeshData &mesh) {
    std::vector<uint32_t> leaves;
    leaves.reserve(mesh.indices.size() / 3);
    for (size_t i = 0; i + 2 < mesh.indices.size(); i += 3) {
        rt_tri tri;
        std::memset(&tri, 0, sizeof(tri));
        for (int c = 0; c < 3; c++) {
            const SdVertex &v = mesh.vertices.at(mesh.indices[i + c]);
            std::memcpy(tri.pos[c], v.position, sizeof(v.position));
            std::memcpy(tri.nrm[c], v.normal, sizeof(v.normal));
            std::memcpy(tri.uv[c], v.texCoord, sizeof(v.texCoord));
        }
        tri.mat = 0; // LambertianMat for every triangle (SimplifiedData.cpp:119)
        const uint32_t ti = storage.addTriangle(tri);
        rt_node leaf;
        std::memset(&leaf, 0, sizeof(leaf));
        leaf.left = leaf.right = ti;
        // sd::GetBoundingBox (SimplifiedData.cpp:309-320): min/max of the corners, grown by 1e-5
        for (int k = 0; k < 3; k++) {
            leaf.pmin[k] = lo(tri.pos[0][k], lo(tri.pos[1][k], tri.pos[2][k])) - 1e-5f;
            leaf.pmax[k] = hi(tri.pos[0][k], hi(tri.pos[1][k], tri.pos[2][k])) + 1e-5f;
        }
        leaf.flags = RT_NODE_LEAF;
        leaves.push_back(storage.addNode(leaf));
    }
    return BuildBVHFromNodes(storage, leaves.data(), 0, leaves.size());
}

SdScene::SdScene() : pDataStorage(std::make_unique<SdDataStorage>()) {}

SdScene::SdScene(const SdScene &other)
    : pDataStorage(std::make_unique<SdDataStorage>(*other.pDataStorage)), sceneIndices(other.sceneIndices) {}

SdScene &SdScene::operator=(const SdScene &other) {
    if (this != &other) {
        pDataStorage = std::make_unique<SdDataStorage>(*other.pDataStorage);
        sceneIndices = other.sceneIndices;
    }
    return *this;
}

uint32_t SdScene::addModel(const std::vector<MeshData> &meshes) {
    std::vector<uint32_t> meshRoots; // PostProcess, Include/ModelLoader.hpp:238-251
    for (const MeshData &m : meshes)
        meshRoots.push_back(AddMesh(*pDataStorage, m));
    const uint32_t root = BuildBVHFromNodes(*pDataStorage, meshRoots.data(), 0, meshRoots.size());
    sceneIndices.push_back(root);
    return root;
}

uint32_t SdScene::loadModelFile(const std::string &path) { return addModel(ReadObjFile(path)); }

void SdScene::finalize() {
    std::vector<uint32_t> idx = sceneIndices;
    pDataStorage->rootIndex = BuildBVHFromNodes(*pDataStorage, idx.data(), 0, idx.size());
    sceneIndices = idx;
}

static inline float bitsAsFloat(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

void ConvertNodeToFlatStorage(const SdDataStorage &storage, std::vector<float> &flat) {
    flat.resize(storage.nodes.size() * 9);
    float *dst = flat.data();
    for (const rt_node &n : storage.nodes) {
        *dst++ = bitsAsFloat(n.left);
        *dst++ = bitsAsFloat(n.right);
        for (int k = 0; k < 3; k++) *dst++ = n.pmin[k];
        for (int k = 0; k < 3; k++) *dst++ = n.pmax[k];
        *dst++ = bitsAsFloat((uint32_t)n.flags);
    }
}

void ConvertTriangleToFlatStorage(const SdDataStorage &storage, std::vector<float> &flat) {
    flat.resize(storage.triangles.size() * 25);
    float *dst = flat.data();
    for (const rt_tri &t : storage.triangles) {
        for (int c = 0; c < 3; c++) {
            for (int k = 0; k < 3; k++) *dst++ = t.pos[c][k];
            for (int k = 0; k < 3; k++) *dst++ = t.nrm[c][k];
            *dst++ = t.uv[c][0];
            *dst++ = t.uv[c][1];
        }
        *dst++ = bitsAsFloat((uint32_t)t.mat);
    }
}

} // namespace rthost

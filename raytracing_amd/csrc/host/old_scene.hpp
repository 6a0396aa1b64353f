// old_scene.hpp -- host-side model of the reference's pointer-based scene: Hittable spheres and
// meshes (CPURayTrace/Objects.hpp, CPURayTrace/Mesh.hpp) under a median-split BVH
// (CPURayTrace/BVH.hpp:20-71), with one material per object (CPURayTrace/Materials/*.hpp).
//
// The reference keeps heap nodes and virtual objects; the tracer library wants index arrays.  The
// scene is therefore built straight into rt_node records with the same split decisions, the same
// std::sort calls and the same box arithmetic, so the flattened tree has the reference tree's shape:
//   sphere object   -> RT_NODE_SPHERE leaf
//   mesh object     -> RT_NODE_MESH leaf whose `left` is the root of the mesh's inner BVH
//   mesh triangle   -> RT_NODE_LEAF leaf inside that inner BVH
#pragma once

#include <cstdint>
#include <vector>

#include "rt_hip.h"
#include "sd_scene.hpp"

namespace rthost {

class OldScene { // Program/Scene.hpp:13-96
  public:
    std::vector<rt_node> nodes;
    std::vector<rt_tri> triangles;
    std::vector<rt_sphere> spheres;
    std::vector<rt_material> materials;
    uint32_t rootIndex = 0xFFFFFFFFu;

    // Scene::addObject(std::make_shared<Sphere>(center, radius, material))
    void addSphere(const float center[3], float radius, const rt_material &material);
    // Scene::addObject(std::make_shared<Mesh>(vertices, indices, material)); builds the inner BVH
    // like the Mesh constructor does (Mesh.hpp:102-133)
    void addMesh(const MeshData &mesh, const rt_material &material);
    // Scene::update() -> BVHTree.build(objects) (Scene.hpp:69-72); reorders the object list in place
    void build();

  private:
    struct Object {
        uint32_t kind; // RT_NODE_SPHERE / RT_NODE_LEAF / RT_NODE_MESH
        uint32_t ref;  // sphere index, triangle index, or inner-BVH root node
        float pmin[3], pmax[3];
    };
    std::vector<Object> objects;
    uint32_t buildRange(std::vector<Object> &objs, int start, int end);
};

} // namespace rthost

// reference_binding.hpp -- the HIP render modes bound to the reference application's own types.
//
// Include this file inside wcg21444/RayTracing AFTER the application's own headers
//     Renderer/RenderInterfaces.hpp   ITraceMethod, ILoadMethod, RenderPipeline            (:5-78)
//     Renderer/Storage.hpp            Storage::SdScene, Storage::SdSceneMutex              (:38-39)
//     Include/Camera.hpp              class Camera: focalLength, position, lookAtCenter,
//                                     width, height, aspectRatio (glm::vec3 / float)       (:10-15)
//     CPURayTrace/Trace.hpp           Trace::bounceLimit                                   (:16)
//     GLShader/Texture.hpp            Texture2D: Width, Height, setData                    (:18-59)
// and it defines, in the global namespace like the CPU modes of Renderer/RenderContexts.hpp:31-46,
// Renderer/LoadMethods.hpp and Renderer/TraceMethods.hpp,
//     SdSceneHIPContext, LoadSdSceneHIP, TraceSdSceneHIP
// ready for  RenderPipeline<SdSceneHIPContext, LoadSdSceneHIP, TraceSdSceneHIP>(SdSceneHIPContext(cam))  in
// Renderer::changeMode (Renderer/Renderer.cpp:39-60).
//
// Nothing here names a type of this repository's stand-ins: every identifier below that is not rt_* or from
// hip_render_modes.hpp is the reference's.  tests/mock_reference/reference_app.hpp declares exactly those identifiers with
// the reference's member names, and tests/test_reference_binding_cpu.py compiles this file against it.
#pragma once

#include <cstddef>

#include "hip_render_modes.hpp"

struct ReferenceApp {
    using ITraceMethod = ::ITraceMethod;
    using ILoadMethod = ::ILoadMethod;
    using Texture2D = ::Texture2D;
    using Camera = ::Camera;

    // sd::Node and sd::Triangle (CPURayTrace/SimplifiedData.hpp:71-77, :82-88) have the layout of rt_node / rt_tri: the
    // storage vectors go to the device as they are.
    static_assert(sizeof(sd::Node) == sizeof(rt_node) && alignof(sd::Node) == alignof(rt_node), "sd::Node is not laid out like rt_node");
    static_assert(sizeof(sd::Triangle) == sizeof(rt_tri) && alignof(sd::Triangle) == alignof(rt_tri), "sd::Triangle is not laid out like rt_tri");
    static_assert(offsetof(sd::Node, box) == offsetof(rt_node, pmin) && offsetof(sd::Node, flags) == offsetof(rt_node, flags),
                  "sd::Node is not laid out like rt_node");
    static_assert(offsetof(sd::Triangle, normals) == offsetof(rt_tri, nrm) && offsetof(sd::Triangle, matFlags) == offsetof(rt_tri, mat),
                  "sd::Triangle is not laid out like rt_tri");

    static std::shared_mutex &sdSceneMutex() { return Storage::SdSceneMutex; }
    // Both storages are vectors allocated at their full capacity once (2^21 - 1 nodes, 2^20 triangles,
    // SimplifiedData.cpp:16-20, :43-47); `nextIndex` is the number of entries in use (SimplifiedData.hpp:99, :114).  Nothing
    // in the reference adds nodes from the back (addNodeBack has no caller), so [0, nextIndex) is the whole tree.
    static rthost::SdSceneArrays sdScene() {
        sd::DataStorage &ds = *Storage::SdScene.pDataStorage;
        return rthost::SdSceneArrays{reinterpret_cast<const rt_node *>(ds.nodeStorage.nodes.data()), ds.nodeStorage.nextIndex, ds.rootIndex,
                                     reinterpret_cast<const rt_tri *>(ds.triangleStorage.triangles.data()), ds.triangleStorage.nextIndex};
    }
    static void readCamera(const Camera &c, rt_camera &cam) {
        cam.focal_length = c.focalLength;
        cam.position[0] = c.position.x;
        cam.position[1] = c.position.y;
        cam.position[2] = c.position.z;
        cam.look_at[0] = c.lookAtCenter.x;
        cam.look_at[1] = c.lookAtCenter.y;
        cam.look_at[2] = c.lookAtCenter.z;
        cam.width = c.width;
        cam.height = c.height;
        cam.aspect_ratio = c.aspectRatio;
    }
    static int bounceLimit() { return Trace::bounceLimit; }
    static bool bvhAcceleration() { return true; } // the flat-array mode does not read the checkbox (Trace.cpp:42-45 is the old scene's)
};

using SdSceneHIPContext = rthost::SdSceneHIPContextT<ReferenceApp>;
using LoadSdSceneHIP = rthost::LoadSdSceneHIPT<ReferenceApp>;
using TraceSdSceneHIP = rthost::TraceSdSceneHIPT<ReferenceApp>;

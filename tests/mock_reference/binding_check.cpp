// binding_check.cpp -- drives reference_binding.hpp the way the application would: fills Storage::SdScene (the reference's
// storage classes: vectors at full capacity, nextIndex entries in use), installs
// RenderPipeline<SdSceneHIPContext, LoadSdSceneHIP, TraceSdSceneHIP>, loads once, traces `frames` samples, writes the last
// image.  Linked against a recording stand-in of the C ABI for the CPU test, against librt_hip.so for the GPU test.
// usage: binding_check <scene.bin> <width> <height> <frames> <out.f32>
//   scene.bin: u32 n_nodes, u32 root, u32 n_tris, n_nodes * 36 bytes of nodes, n_tris * 100 bytes of triangles, 9 floats of
//   camera (focal, position, look-at, width, aspect), i32 bounce limit
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "reference_app.hpp"

#include "reference_binding.hpp"

int main(int argc, char **argv) {
    if (argc < 6)
        return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f)
        return 3;
    uint32_t head[3];
    if (fread(head, 4, 3, f) != 3)
        return 3;
    sd::DataStorage &ds = *Storage::SdScene.pDataStorage;
    if (head[0] > sd::NODESIZE || head[2] > sd::TRIANGLESIZE)
        return 4;
    if (fread(ds.nodeStorage.nodes.data(), sizeof(sd::Node), head[0], f) != head[0])
        return 3;
    if (fread(ds.triangleStorage.triangles.data(), sizeof(sd::Triangle), head[2], f) != head[2])
        return 3;
    ds.nodeStorage.nextIndex = head[0];
    ds.triangleStorage.nextIndex = head[2];
    ds.rootIndex = head[1];
    float c[9];
    int bounce;
    if (fread(c, 4, 9, f) != 9 || fread(&bounce, 4, 1, f) != 1)
        return 3;
    fclose(f);
    Camera cam;
    cam.focalLength = c[0];
    cam.position = vec3(c[1], c[2], c[3]);
    cam.lookAtCenter = vec3(c[4], c[5], c[6]);
    cam.width = c[7];
    cam.aspectRatio = c[8];
    cam.height = cam.width / cam.aspectRatio; // Camera::Camera, Include/Camera.hpp:26
    Trace::bounceLimit = bounce;
    const int W = atoi(argv[2]), H = atoi(argv[3]), frames = atoi(argv[4]);
    try {
        RenderPipeline<SdSceneHIPContext, LoadSdSceneHIP, TraceSdSceneHIP> pipeline{SdSceneHIPContext(cam)};
        pipeline.getLoadMethod()->load();
        Texture2D a, b; // TracerAsync::render swaps the two and passes the sample count, Renderer/TracerImpl.cpp:9-12
        a.Width = b.Width = (unsigned)W;
        a.Height = b.Height = (unsigned)H;
        for (int n = 1; n <= frames; n++) {
            std::swap(a, b);
            pipeline.getTraceMethod()->trace(a, b, n);
        }
        FILE *o = fopen(argv[5], "wb");
        if (!o)
            return 5;
        fwrite(b.lastUpload.data(), 4, b.lastUpload.size(), o);
        fclose(o);
    } catch (std::exception &e) {
        fprintf(stderr, "binding_check: %s\n", e.what());
        return 1;
    }
    return 0;
}

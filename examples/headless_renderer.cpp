// headless_renderer.cpp -- the reference's frame loop without a window, in C++, over the HIP render mode.
//
// What main.cpp / Renderer do per frame in the reference (main.cpp:95-108, Renderer/Renderer.cpp:64-102):
//   Storage::SdScene.initialize()                       -> load a model, build the scene BVH (Program/Scene.cpp:187-213)
//   changeMode(HIP_SdScene)                             -> RenderPipeline<Context, Load, Trace>   (Renderer.cpp:31-62)
//   uploader->upload(load) when SceneDirty              -> ILoadMethod::load()                    (Renderer.cpp:80-85)
//   tracer->render(trace) once per frame                -> ITraceMethod::trace(in, out, n++)      (TracerImpl.cpp:9-12)
// and here the accumulated image is written as a PPM with the post-process gamma (GLSL/postProcess.fs:26) instead of
// being drawn.  Build:  make -C examples      Run:  examples/headless_renderer model.obj out.ppm [frames] [W H]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>

#include "render_plugin.hpp"

using namespace rthost;

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s model.obj out.ppm [frames=32] [width=640 height=360]\n", argv[0]);
        return 2;
    }
    const std::string model = argv[1], out = argv[2];
    const int frames = argc > 3 ? std::atoi(argv[3]) : 32;
    const int W = argc > 5 ? std::atoi(argv[4]) : 640, H = argc > 5 ? std::atoi(argv[5]) : 360; // main.cpp:19-20
    try {
        {   // Storage::SdScene.initialize()
            std::unique_lock<std::shared_mutex> lock(Storage::SdSceneMutex);
            Storage::SdScene_ = SdScene();
            Storage::SdScene_.loadModelFile(model);
            Storage::SdScene_.finalize();
        }
        // a camera that frames the model's bounding box from the front (the reference's default, Renderer.hpp:61,
        // looks at the origin from (0, 0, 1))
        const SdDataStorage &ds = *Storage::SdScene_.pDataStorage;
        float c[3], ext = 0.f;
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f}; // leaf boxes: the true extent
        for (const rt_node &nd : ds.nodes)
            if (nd.flags == RT_NODE_LEAF)
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::fmin(lo[k], nd.pmin[k]);
                    hi[k] = std::fmax(hi[k], nd.pmax[k]);
                }
        for (int k = 0; k < 3; k++) {
            c[k] = 0.5f * (lo[k] + hi[k]);
            ext = std::fmax(ext, hi[k] - lo[k]);
        }
        const float pos[3] = {c[0], c[1] + 0.15f * ext, c[2] + 1.6f * ext};
        Camera cam(1.0f, pos, 2.0f, (float)W / (float)H, c);
        Trace::bounceLimit = 8;

        // changeMode(HIP_SdScene)
        std::unique_ptr<IRenderPipeline> pipeline =
            std::make_unique<RenderPipeline<SdSceneHIPContext, LoadSdSceneHIP, TraceSdSceneHIP>>(SdSceneHIPContext(cam));
        TracerAsync tracer(W, H);

        pipeline->getLoadMethod()->load(); // SceneDirty
        tracer.resetSamples();
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++)
            tracer.render(*pipeline->getTraceMethod()); // one sample per pixel per frame, running mean
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%d frames of %dx%d in %.2f ms (%.3f ms per frame incl. the copy to the host texture)\n", frames, W, H, ms,
                    ms / frames);

        const Texture2D &img = tracer.output();
        std::ofstream fh(out, std::ios::binary);
        fh << "P6\n" << W << " " << H << "\n255\n";
        for (int y = H - 1; y >= 0; y--) // row 0 is the bottom row
            for (int x = 0; x < W; x++)
                for (int k = 0; k < 3; k++) {
                    float v = std::pow(std::fmax(img.texels[((size_t)y * W + x) * 4 + k], 0.0f), 1.0f / 2.2f);
                    fh.put((char)(unsigned char)(std::fmin(v, 1.0f) * 255.0f + 0.5f));
                }
        std::printf("wrote %s\n", out.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

// demo_main.cpp -- the role of the reference's demo app (src/main.rs:32-240) for the C++ host:
// build the cover scene from components, extract, run the node once, write the frame.
//   usage: bevyray_demo [width height spp bounces seed out.bin]
// out.bin = width*height*4 floats (RGBA f32, top row first); also prints the stats.
#include <cstdio>
#include <cstdlib>

#include "raytracing.hpp"

int main(int argc, char** argv) {
    using namespace bevyray;
    const uint32_t width = argc > 1 ? std::atoi(argv[1]) : 400, height = argc > 2 ? std::atoi(argv[2]) : 225;
    const uint32_t spp = argc > 3 ? std::atoi(argv[3]) : 1, bounces = argc > 4 ? std::atoi(argv[4]) : 4;
    const float seed = argc > 5 ? (float)std::atof(argv[5]) : 0.5f;
    const char* out = argc > 6 ? argv[6] : nullptr;
    try {
        RaytracePlugin plugin({0});
        RayTracingNode node = plugin.node();
        const Buffers buffers = generate_scene(BRT_SCENE_COVER, 1);
        RaytracedCamera cam{Raytracing::Pure, spp, bounces};
        const auto view = extract_camera(cam, Transform{{13.0f, 2.0f, 3.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}},
                                         PerspectiveProjection{0.4f, (float)width / (float)height, 0.1f, 1000.0f});
        const WindowExtract window = extract_window(height, seed);
        std::vector<float> frame;
        brt_stats st{};
        if (!node.run(view, window, width, height, &buffers, nullptr, nullptr, frame, &st)) {
            std::fprintf(stderr, "pass skipped\n");
            return 2;
        }
        std::printf("%ux%u spp %u bounces %u: %llu rays, kernel %.3f ms, %.1f Mrays/s\n", width, height, spp, bounces,
                    (unsigned long long)st.rays, st.kernel_ms, st.rays / st.kernel_ms / 1e3);
        if (out) {
            FILE* f = std::fopen(out, "wb");
            if (!f) { std::perror(out); return 1; }
            std::fwrite(frame.data(), sizeof(float), frame.size(), f);
            std::fclose(f);
        }
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return e.code == BRT_ERR_NO_DEVICE ? 3 : 1;
    }
    return 0;
}

// raytracing.hpp -- C++ host side of the render node, mirroring bevyray's plugin surface
// (reference src/raytracing/mod.rs, extract.rs, pipeline.rs) on top of the C ABI
// (include/bevyray_amd.h).  The reference is Rust on Bevy; no Rust toolchain exists in this
// environment, so the host layer is C++17 with the reference's names, fields and error
// behaviour.  INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add instead.
//
//   bevyray::RaytracePlugin         mod.rs:24-84      (GPU context = RaytracingPipeline::from_world)
//   bevyray::RaytracedCamera        mod.rs:86-91
//   bevyray::Raytracing             mod.rs:94-101     #[repr(u32)] Skip..Pure = 0..3
//   bevyray::RaytracedSphere        mod.rs:103-106
//   bevyray::StandardMaterial       the fields extract.rs:200-207 reads, bevy 0.14 defaults
//   bevyray::CameraExtract / WindowExtract / RaytraceLevelExtract / RaytraceMaterial / Model / BVHNode
//                                   extract.rs:56-237 (byte-exact GPU layouts)
//   bevyray::prepare_buffers        extract.rs:280-337
//   bevyray::RayTracingNode::run    pipeline.rs:58-220
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bevyray_amd.h"

namespace bevyray {

using Vec3 = std::array<float, 3>;

enum class Raytracing : uint32_t { Skip = 0, FallbackRaster = 1, FallbackRaytraced = 2, Pure = 3 };

struct RaytracedCamera {
    Raytracing level = Raytracing::FallbackRaytraced;
    uint32_t sample_count = 4;
    uint32_t bounces = 4;
};
struct RaytracedSphere { float radius = 1.0f; };

struct StandardMaterial {            // bevy 0.14 defaults; base_color is sRGB
    Vec3 base_color{1.0f, 1.0f, 1.0f};
    float metallic = 0.0f, perceptual_roughness = 0.5f, reflectance = 0.5f, ior = 1.5f, specular_transmission = 0.0f;
};
struct Transform {                   // Transform::from_translation(t).looking_at(target, up)
    Vec3 translation{0.0f, 0.0f, 5.0f}, target{0.0f, 0.0f, 0.0f}, up{0.0f, 1.0f, 0.0f};
};
struct PerspectiveProjection { float fov = 0.7853982f, aspect_ratio = 1.0f, near = 0.1f, far = 1000.0f; };
struct OrthographicProjection {};    // unsupported by the reference (extract.rs:148)

// ---- GPU-layout structs (extract.rs:56-61, 83-104, 181-189, 213-218, 229-237) ----
struct alignas(16) WindowExtract { float random_seed; uint32_t height; float _padding[2]; };
struct alignas(16) CameraExtract {
    uint32_t sample_count, bounce_count, projection; float near, far, fov, aspect, _p0;
    float position[3], _p1, direction[3], _p2, up[3], _p3;
};
struct alignas(16) RaytraceLevelExtract { uint32_t level; uint32_t _p0[3]; float _padding[3]; uint32_t _p1; };
struct alignas(16) RaytraceMaterial { float base_color[3], metallic, roughness, reflectance, ior, specular_transmission; };
struct alignas(16) Model { float position[3], radius; uint32_t material_id, _p[3]; };
struct alignas(16) BVHNode { float bounds_min[3], _p0, bounds_max[3]; uint32_t index, model_count, _p1[3]; };
static_assert(sizeof(WindowExtract) == 16 && sizeof(CameraExtract) == 80 && sizeof(RaytraceLevelExtract) == 32, "uniform layouts");
static_assert(sizeof(RaytraceMaterial) == 32 && sizeof(Model) == 32 && sizeof(BVHNode) == 48, "storage layouts");

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int32_t rc, const brt_ctx* ctx = nullptr) {
    if (rc != BRT_OK) throw Error(rc, std::string("bevyray_amd error ") + std::to_string(rc) + ": " + brt_last_error(ctx));
}

// extract.rs:118-157; nullopt for anything but a perspective projection
inline std::optional<std::pair<RaytraceLevelExtract, CameraExtract>> extract_camera(const RaytracedCamera& cam, const Transform& t,
                                                                                   const PerspectiveProjection& p) {
    CameraExtract c{};
    check(brt_host_camera_extract(t.translation.data(), t.target.data(), t.up.data(), p.fov, p.aspect_ratio, p.near, p.far,
                                  cam.sample_count, cam.bounces, &c));
    RaytraceLevelExtract l{};
    l.level = static_cast<uint32_t>(cam.level);
    return std::make_pair(l, c);
}
inline std::optional<std::pair<RaytraceLevelExtract, CameraExtract>> extract_camera(const RaytracedCamera&, const Transform&,
                                                                                   const OrthographicProjection&) {
    return std::nullopt;
}
// extract.rs:70-80; the reference draws the seed from thread_rng every frame, here it is explicit
inline WindowExtract extract_window(uint32_t physical_height, float random_seed) {
    WindowExtract w{};
    check(brt_host_window_extract(random_seed, physical_height, &w));
    return w;
}
// extract.rs:196-208
inline RaytraceMaterial prepare_asset(const StandardMaterial& m) {
    RaytraceMaterial r{};
    check(brt_host_material(m.base_color.data(), m.metallic, m.perceptual_roughness, m.reflectance, m.ior, m.specular_transmission, &r));
    return r;
}

// ModelBuffer / MaterialBuffer / BVHBuffer (extract.rs:252-262)
struct Buffers {
    std::vector<Model> models;
    std::vector<RaytraceMaterial> materials;
    std::vector<BVHNode> bvh;
};
struct SphereEntity { Vec3 translation; RaytracedSphere sphere; StandardMaterial material; };

// extract.rs:280-337: one Model and one material entry per sphere, then the BVH
inline Buffers prepare_buffers(const std::vector<SphereEntity>& data) {
    Buffers b;
    for (size_t i = 0; i < data.size(); i++) {
        b.materials.push_back(prepare_asset(data[i].material));
        Model m{};
        std::memcpy(m.position, data[i].translation.data(), 12);
        m.radius = data[i].sphere.radius;
        m.material_id = static_cast<uint32_t>(i);
        b.models.push_back(m);
    }
    b.bvh.resize(b.models.empty() ? 0 : 2 * b.models.size() - 1);
    uint32_t n = 0;
    check(brt_build_bvh(b.models.data(), static_cast<uint32_t>(b.models.size()), b.bvh.data(), static_cast<uint32_t>(b.bvh.size()), &n));
    b.bvh.resize(n);
    return b;
}

// Seeded version of the demo's setup() (main.rs:49-240) and the other benchmark scenes
inline Buffers generate_scene(uint32_t kind, uint64_t seed) {
    Buffers b;
    b.models.resize(16384);
    b.materials.resize(16384);
    uint32_t n = 0;
    check(brt_scene_generate(kind, seed, b.models.data(), b.materials.data(), 16384, &n));
    b.models.resize(n);
    b.materials.resize(n);
    b.bvh.resize(n ? 2 * n - 1 : 0);
    uint32_t nn = 0;
    check(brt_build_bvh(b.models.data(), n, b.bvh.data(), static_cast<uint32_t>(b.bvh.size()), &nn));
    b.bvh.resize(nn);
    return b;
}

class RaytracePlugin;

// pipeline.rs:29-221
class RayTracingNode {
public:
    explicit RayTracingNode(brt_ctx* ctx) : ctx_(ctx) {}
    // pipeline.rs:136-138
    // (an empty b.bvh: the callee builds the tree itself -- binned SAH, the recommended path: include/bevyray_amd.h)
    void write_buffers(const Buffers& b) {
        check(brt_upload_scene(ctx_, b.models.data(), static_cast<uint32_t>(b.models.size()), b.materials.data(),
                               static_cast<uint32_t>(b.materials.size()), b.bvh.empty() ? nullptr : b.bvh.data(),
                               static_cast<uint32_t>(b.bvh.size())), ctx_);
    }
    // Returns false when the pass is skipped the way the reference skips it (no camera extract,
    // empty buffers: pipeline.rs:82-151); throws Error for real failures.
    bool run(const std::optional<std::pair<RaytraceLevelExtract, CameraExtract>>& view, const WindowExtract& window,
             uint32_t width, uint32_t height, const Buffers* buffers, const float* raster_rgba, const float* raster_depth,
             std::vector<float>& destination, brt_stats* stats = nullptr, uint32_t flags = 0) {
        if (!view) return false;
        if (buffers) {
            const int32_t rc = brt_upload_scene(ctx_, buffers->models.data(), static_cast<uint32_t>(buffers->models.size()),
                                                buffers->materials.data(), static_cast<uint32_t>(buffers->materials.size()),
                                                buffers->bvh.data(), static_cast<uint32_t>(buffers->bvh.size()));
            if (rc == BRT_ERR_EMPTY_SCENE) return false;
            check(rc, ctx_);
        }
        destination.resize(static_cast<size_t>(width) * height * 4);
        check(brt_render(ctx_, &view->second, &window, view->first.level, width, height, raster_rgba, raster_depth,
                         destination.data(), flags, stats), ctx_);
        return true;
    }
    // The same pass on an N-device context with the frame assembled ON THE FIRST DEVICE (brt_render_device: tiles by peer
    // copy over xGMI, one de-interleave kernel): `d_destination` is a device pointer, e.g. the mapped colour target
    // (pipeline.rs:191-203).  d_raster_* are optional device buffers on the first device.
    bool run_device(const std::optional<std::pair<RaytraceLevelExtract, CameraExtract>>& view, const WindowExtract& window,
                    uint32_t width, uint32_t height, const float* d_raster_rgba, const float* d_raster_depth, void* d_destination,
                    void* hip_stream = nullptr, brt_stats* stats = nullptr, uint32_t flags = 0) {
        if (!view) return false;
        check(brt_render_device(ctx_, &view->second, &window, view->first.level, width, height, d_raster_rgba, d_raster_depth,
                                d_destination, hip_stream, flags, stats), ctx_);
        return true;
    }
    // One process per GPU: this rank's strips into a device tile, then ONE ncclGather of the tiles to rank 0 and the
    // de-interleave kernel behind it, both inside the library (brt_gather_rccl; `comm` from RaytracePlugin::rccl_comm).
    bool run_part(const std::optional<std::pair<RaytraceLevelExtract, CameraExtract>>& view, const WindowExtract& window, uint32_t width,
                  uint32_t height, uint32_t rank, uint32_t world, float* d_tile, brt_stats* stats = nullptr, uint32_t flags = 0) {
        if (!view) return false;
        check(brt_render_part_device(ctx_, &view->second, &window, view->first.level, width, height, rank, world, nullptr, nullptr, d_tile,
                                     nullptr, flags, stats), ctx_);
        return true;
    }
    void gather(void* comm, int32_t rank, int32_t world, const float* d_tile, float* d_tiles_on_root, uint32_t width, uint32_t height,
                void* d_frame_on_root, void* hip_stream = nullptr, uint32_t flags = 0) {
        check(brt_gather_rccl(ctx_, comm, rank, world, d_tile, d_tiles_on_root, width, height, d_frame_on_root, hip_stream, flags), ctx_);
    }
private:
    brt_ctx* ctx_;
};

// mod.rs:24-84
class RaytracePlugin {
public:
    explicit RaytracePlugin(const std::vector<int32_t>& device_ids = {0}) {
        check(brt_create(device_ids.data(), static_cast<int32_t>(device_ids.size()), &ctx_));
    }
    ~RaytracePlugin() { brt_destroy(ctx_); }
    RaytracePlugin(const RaytracePlugin&) = delete;
    RaytracePlugin& operator=(const RaytracePlugin&) = delete;
    RayTracingNode node() { return RayTracingNode(ctx_); }
    // page-locked frame memory: brt_render DMAs straight into it (brt_host_alloc)
    float* alloc_frame(uint32_t width, uint32_t height) {
        void* p = nullptr;
        check(brt_host_alloc(ctx_, static_cast<uint64_t>(width) * height * 16, &p), ctx_);
        return static_cast<float*>(p);
    }
    // the reading of `||` in raytrace.wgsl:269 (BRT_POLICY_OR_SHORT_CIRCUIT or 0); scheduling knobs (never change a pixel)
    void set_policy(uint32_t flags) { check(brt_set_policy(ctx_, flags), ctx_); }
    void set_tuning(const char* name, uint32_t value) { check(brt_set_tuning(ctx_, name, value), ctx_); }
    // one process per GPU: the strips dealt out to the ranks by measured cost instead of s % world (brt_plan_strips: every rank computes
    // the same table from the same probe frame; an empty vector to set_strip_table: back to s % world)
    std::vector<uint32_t> plan_strips(const std::pair<RaytraceLevelExtract, CameraExtract>& view, const WindowExtract& window, uint32_t width,
                                      uint32_t height, uint32_t world, uint32_t probe_spp = 4) {
        std::vector<uint32_t> table((height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS);
        check(brt_plan_strips(ctx_, &view.second, &window, view.first.level, width, height, world, probe_spp, table.data()), ctx_);
        return table;
    }
    void set_strip_table(uint32_t world, const std::vector<uint32_t>& part_of_strip) {
        check(brt_set_strip_table(ctx_, world, static_cast<uint32_t>(part_of_strip.size()), part_of_strip.empty() ? nullptr : part_of_strip.data()), ctx_);
    }
    // the colour target's memory, exported by the host's graphics API as a file descriptor (pipeline.rs:191-203 renders straight
    // into post_process.destination): a device pointer that run_device / gather accept as the frame (brt_import_frame_fd)
    void* import_frame(int32_t fd, uint64_t bytes, uint32_t handle_type = BRT_EXTMEM_OPAQUE_FD) {
        void* d = nullptr;
        check(brt_import_frame_fd(ctx_, fd, bytes, handle_type, &d), ctx_);
        return d;
    }
    void release_frame(void* d_frame) { check(brt_release_frame(ctx_, d_frame), ctx_); }
    // the communicator of the one-process-per-GPU form: `id` from rccl_unique_id() on rank 0, handed to the others by the host's means
    static std::array<char, 128> rccl_unique_id() {
        std::array<char, 128> id{};
        check(brt_rccl_unique_id(id.data()));
        return id;
    }
    void* rccl_comm(const std::array<char, 128>& id, int32_t rank, int32_t world) {
        void* comm = nullptr;
        check(brt_rccl_comm_create(ctx_, id.data(), rank, world, &comm), ctx_);
        return comm;
    }
    void rccl_comm_destroy(void* comm) { check(brt_rccl_comm_destroy(ctx_, comm), ctx_); }
    brt_ctx* context() { return ctx_; }
private:
    brt_ctx* ctx_ = nullptr;
};

}  // namespace bevyray

// brt_interop.cpp -- the two places where the render node meets memory and processes that are not its own:
//
//   * the ONE collective of the path: one process per GPU, every rank's tile to rank 0 in one ncclGather
//     (/opt/rocm/include/rccl/rccl.h:745), then the de-interleave kernel -- behind the C ABI, so that a host with no RCCL
//     binding of its own (the Rust node) can run the multi-process form.  librccl is resolved with dlopen at first use:
//     a single-GPU host needs no RCCL at all, and a process that already has one loaded (PyTorch bundles its own copy
//     under the same SONAME) keeps using THAT one.
//   * frame targets that live in another API's memory: the reference's pass renders straight into
//     post_process.destination (pipeline.rs:191-203); a wgpu/Vulkan host exports that buffer's memory as a file descriptor
//     (VK_KHR_external_memory_fd) and brt_import_frame_fd maps it, so that brt_render_device writes the frame where the
//     next pass reads it -- no copy.
//
// No ray arithmetic here.
#include <dlfcn.h>
#include <unistd.h>

#include <cstdlib>
#include <mutex>

#include "brt_ctx.h"

using namespace brt;

namespace {

// ---- RCCL by dlopen --------------------------------------------------------------------------------------------------
struct NcclId { char internal[128]; };          // ncclUniqueId (rccl.h:43), passed by value to ncclCommInitRank
constexpr int kNcclFloat = 7;                   // ncclFloat32 (rccl.h:466)

struct Rccl {
    void* lib = nullptr;
    int (*get_unique_id)(NcclId*) = nullptr;                                                   // rccl.h:187
    int (*comm_init_rank)(void**, int, NcclId, int) = nullptr;                                 // rccl.h:220
    int (*comm_destroy)(void*) = nullptr;                                                      // rccl.h:260
    int (*gather)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;         // rccl.h:745
    const char* (*error_string)(int) = nullptr;                                                // rccl.h:339
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already in the process (PyTorch's, or the host's own) first: two RCCLs in one process work, but each
        // would bring up its own transports
        // BRT_RCCL_LIB=<path>: that library and no other (a host that ships its own RCCL; tests force the not-found path with it).
        // Read once, here: nothing reads the environment per frame.
        const char* forced = std::getenv("BRT_RCCL_LIB");
        const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::vector<const char*> names;
        if (forced && *forced) names.push_back(forced);
        else names.assign(defaults, defaults + 3);
        for (const char* n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        std::string err;
        for (const char* n : names) {
            if (r.lib) break;
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (!r.lib) { const char* e = dlerror(); err = e ? e : "?"; }      // (dlerror() clears what it returns: one call per failure)
        }
        if (!r.lib) { r.why = "librccl not found: " + err; return; }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p && r.why.empty()) r.why = std::string("librccl has no ") + name;
            return p;
        };
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(sym("ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(sym("ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
        r.gather = reinterpret_cast<decltype(r.gather)>(sym("ncclGather"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(sym("ncclGetErrorString"));
        if (!r.why.empty()) r.lib = nullptr;
    });
    return r;
}

int32_t nccl_fail(brt_ctx* ctx, const char* what, int rc) {
    Rccl& r = rccl();
    return ctx_fail(ctx, BRT_ERR_RCCL, std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "error") + " (" + std::to_string(rc) + ")");
}

// ---- external memory -------------------------------------------------------------------------------------------------
int32_t map_vmm(brt_ctx* ctx, int device, hipMemGenericAllocationHandle_t h, size_t bytes, void** out_ptr) {
    void* p = nullptr;
    HIP_TRY(ctx, hipMemAddressReserve(&p, bytes, 0, nullptr, 0));
    hipError_t e = hipMemMap(p, bytes, 0, h, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc ad{};
        ad.location.type = hipMemLocationTypeDevice;
        ad.location.id = device;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(p, bytes, &ad, 1);
        if (e != hipSuccess) (void)hipMemUnmap(p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipMemAddressFree(p, bytes);
        return ctx_fail(ctx, BRT_ERR_HIP, std::string("mapping the allocation: ") + hipGetErrorString(e));
    }
    *out_ptr = p;
    return BRT_OK;
}

void release_one(ExternalFrame& f) {
    if (f.type == BRT_EXTMEM_OPAQUE_FD) {
        if (f.ptr) (void)hipFree(f.ptr);
        if (f.ext) (void)hipDestroyExternalMemory(f.ext);
    } else {
        if (f.ptr) { (void)hipMemUnmap(f.ptr, f.mapped); (void)hipMemAddressFree(f.ptr, f.mapped); }
        (void)hipMemRelease(f.vmm);
    }
    f = ExternalFrame();
}

size_t vmm_granularity(int device, bool minimum = false) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, minimum ? hipMemAllocationGranularityMinimum : hipMemAllocationGranularityRecommended) != hipSuccess || g == 0)
        g = minimum ? 4096u : (2u << 20);
    return g;
}

}  // namespace

namespace brt {
void release_external_frames(brt_ctx* ctx) {
    if (ctx->external.empty()) return;
    if (hipSetDevice(ctx->devs[0].device) != hipSuccess) return;
    for (auto& f : ctx->external) release_one(f);
    ctx->external.clear();
}
}  // namespace brt

extern "C" {

int32_t brt_rccl_unique_id(void* out_id128) {
    return guard(nullptr, [&]() -> int32_t {
    if (!out_id128) return fail(BRT_ERR_INVALID_ARGUMENT, "out_id128 is null");
    Rccl& r = rccl();
    if (!r.lib) return fail(BRT_ERR_RCCL, r.why);
    NcclId id;
    const int rc = r.get_unique_id(&id);
    if (rc != 0) return nccl_fail(nullptr, "ncclGetUniqueId", rc);
    std::memcpy(out_id128, &id, sizeof id);
    return BRT_OK;
    });
}

int32_t brt_rccl_comm_create(brt_ctx* ctx, const void* id128, int32_t rank, int32_t world, void** out_comm) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx || !id128 || !out_comm) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    *out_comm = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "rank / world out of range");
    Rccl& r = rccl();
    if (!r.lib) return ctx_fail(ctx, BRT_ERR_RCCL, r.why);
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].device));       // the communicator binds to the current device: one process per GPU
    NcclId id;
    std::memcpy(&id, id128, sizeof id);
    void* comm = nullptr;
    const int rc = r.comm_init_rank(&comm, world, id, rank);
    if (rc != 0) return nccl_fail(ctx, "ncclCommInitRank", rc);
    *out_comm = comm;
    return BRT_OK;
    });
}

int32_t brt_rccl_comm_destroy(brt_ctx* ctx, void* comm) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!comm) return BRT_OK;
    Rccl& r = rccl();
    if (!r.lib) return ctx_fail(ctx, BRT_ERR_RCCL, r.why);
    if (ctx) HIP_TRY(ctx, hipSetDevice(ctx->devs[0].device));
    const int rc = r.comm_destroy(comm);
    return rc == 0 ? BRT_OK : nccl_fail(ctx, "ncclCommDestroy", rc);
    });
}

int32_t brt_gather_rccl(brt_ctx* ctx, void* nccl_comm, int32_t rank, int32_t world, const float* d_tile, float* d_tiles_on_root,
                        uint32_t width, uint32_t height, void* d_frame_on_root, void* hip_stream, uint32_t flags) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!nccl_comm || !d_tile) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "null communicator / tile");
    if (world < 1 || rank < 0 || rank >= world) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "rank / world out of range");
    if (rank == 0 && !d_tiles_on_root) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "rank 0 needs the receive buffer");
    if (width == 0 || height == 0) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "empty frame");
    Rccl& r = rccl();
    if (!r.lib) return ctx_fail(ctx, BRT_ERR_RCCL, r.why);
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    const bool own_stream = (hip_stream == nullptr) && !(flags & BRT_FLAG_CALLER_STREAM);
    hipStream_t stream = own_stream ? dc.stream : static_cast<hipStream_t>(hip_stream);
    const uint32_t tile_rows = brt_tile_rows(height, (uint32_t)world);
    const size_t count = (size_t)tile_rows * width * 4;                                           // floats per rank
    const int rc = r.gather(d_tile, d_tiles_on_root, count, kNcclFloat, 0, nccl_comm, stream);      // THE collective of the path
    if (rc != 0) return nccl_fail(ctx, "ncclGather", rc);
    if (rank == 0 && d_frame_on_root) {
        const uint32_t* part_of_strip = nullptr;                   // the context's strip table (brt_set_strip_table), if it is one for this frame and split
        FrameParams key{};
        key.height = height; key.n_parts = (uint32_t)world; key.part = 0u;
        const int32_t rt = strip_table_attach(ctx, dc, &key, &part_of_strip, stream);
        if (rt != BRT_OK) return rt;
        HIP_TRY(ctx, launch_deinterleave(d_tiles_on_root, d_frame_on_root, width, height, (uint32_t)world, tile_rows, flags & BRT_FLAG_OUT_MASK, stream, part_of_strip));
    }
    if (own_stream) HIP_TRY(ctx, hipStreamSynchronize(stream));
    return BRT_OK;
    });
}

int32_t brt_import_frame_fd(brt_ctx* ctx, int32_t fd, uint64_t bytes, uint32_t handle_type, void** out_d_frame) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx || !out_d_frame) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    *out_d_frame = nullptr;
    if (fd < 0 || bytes == 0) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "bad file descriptor / zero size");
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    ExternalFrame f;
    f.type = handle_type;
    f.bytes = (size_t)bytes;
    if (handle_type == BRT_EXTMEM_OPAQUE_FD) {
        // what a Vulkan allocation exported with VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT hands over (on success the
        // runtime owns the descriptor)
        hipExternalMemoryHandleDesc hd{};
        hd.type = hipExternalMemoryHandleTypeOpaqueFd;
        hd.handle.fd = fd;
        hd.size = bytes;
        HIP_TRY(ctx, hipImportExternalMemory(&f.ext, &hd));
        hipExternalMemoryBufferDesc bd{};
        bd.offset = 0;
        bd.size = bytes;
        const hipError_t e = hipExternalMemoryGetMappedBuffer(&f.ptr, f.ext, &bd);
        if (e != hipSuccess) {
            (void)hipDestroyExternalMemory(f.ext);
            return ctx_fail(ctx, BRT_ERR_HIP, std::string("hipExternalMemoryGetMappedBuffer: ") + hipGetErrorString(e));
        }
    } else if (handle_type == BRT_EXTMEM_DMABUF_FD) {
        // a dma-buf of a HIP virtual-memory allocation in another process or API (hipMemExportToShareableHandle)
        // (HIP reads the descriptor THROUGH the pointer -- `*(int*)osHandle` -- where CUDA's driver API takes the descriptor cast
        //  to a pointer: passing the value itself faults inside the runtime)
        int os_fd = fd;
        HIP_TRY(ctx, hipMemImportFromShareableHandle(&f.vmm, &os_fd, hipMemHandleTypePosixFileDescriptor));
        // The size of the imported allocation cannot be queried, and a mapping must not ask for more than the handle owns: `bytes`
        // rounded up to the RECOMMENDED granularity (what brt_debug_export_frame_fd and most exporters allocate), then to the
        // MINIMUM one, then `bytes` itself (an exporter that allocated exactly that).  The first that maps is the allocation.
        size_t cand[3] = {0, 0, f.bytes};
        {
            const size_t g = vmm_granularity(dc.device), gm = vmm_granularity(dc.device, true);
            cand[0] = (f.bytes + g - 1) / g * g;
            cand[1] = (f.bytes + gm - 1) / gm * gm;
        }
        int32_t rc = BRT_ERR_HIP;
        for (int i = 0; i < 3 && rc != BRT_OK; i++) {
            if (i > 0 && (cand[i] == cand[i - 1] || cand[i] == cand[0])) continue;
            f.mapped = cand[i];
            rc = map_vmm(ctx, dc.device, f.vmm, f.mapped, &f.ptr);
        }
        if (rc != BRT_OK) {
            (void)hipMemRelease(f.vmm);
            return ctx_fail(ctx, BRT_ERR_HIP, "the dma-buf could not be mapped at " + std::to_string(cand[0]) + ", " + std::to_string(cand[1]) + " or " +
                                                  std::to_string(cand[2]) + " bytes: `bytes` must be the size the exporter allocated (" + ctx->last_error + ")");
        }
    } else {
        return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "unknown external-memory handle type");
    }
    ctx->external.push_back(f);
    *out_d_frame = f.ptr;
    return BRT_OK;
    });
}

int32_t brt_release_frame(brt_ctx* ctx, void* d_frame) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    for (size_t i = 0; i < ctx->external.size(); i++)
        if (ctx->external[i].ptr == d_frame) {
            HIP_TRY(ctx, hipSetDevice(ctx->devs[0].device));
            for (auto& dc : ctx->devs) {          // nothing of this context may still be writing into it
                if (hipSetDevice(dc.device) == hipSuccess) { (void)hipEventSynchronize(dc.ev_last); (void)hipEventSynchronize(dc.ev_asm); }
            }
            HIP_TRY(ctx, hipSetDevice(ctx->devs[0].device));
            release_one(ctx->external[i]);
            ctx->external.erase(ctx->external.begin() + (long)i);
            return BRT_OK;
        }
    return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "pointer was not returned by brt_import_frame_fd / brt_debug_export_frame_fd");
    });
}

int32_t brt_debug_export_frame_fd(brt_ctx* ctx, uint64_t bytes, int32_t* out_fd, void** out_d_ptr) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx || !out_fd || !out_d_ptr || bytes == 0) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer / zero size");
    *out_fd = -1;
    *out_d_ptr = nullptr;
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dc.device;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    const size_t g = vmm_granularity(dc.device);
    ExternalFrame f;
    f.type = BRT_EXTMEM_DMABUF_FD;
    f.bytes = (size_t)bytes;
    f.mapped = (f.bytes + g - 1) / g * g;
    HIP_TRY(ctx, hipMemCreate(&f.vmm, f.mapped, &prop, 0));
    int fd = -1;
    hipError_t e = hipMemExportToShareableHandle(&fd, f.vmm, hipMemHandleTypePosixFileDescriptor, 0);
    if (e != hipSuccess) {
        (void)hipMemRelease(f.vmm);
        return ctx_fail(ctx, BRT_ERR_HIP, std::string("hipMemExportToShareableHandle: ") + hipGetErrorString(e));
    }
    const int32_t rc = map_vmm(ctx, dc.device, f.vmm, f.mapped, &f.ptr);
    if (rc != BRT_OK) { (void)close(fd); (void)hipMemRelease(f.vmm); return rc; }
    ctx->external.push_back(f);
    *out_fd = fd;
    *out_d_ptr = f.ptr;
    return BRT_OK;
    });
}

int32_t brt_debug_copy_to_host(brt_ctx* ctx, const void* d_src, void* h_dst, uint64_t bytes) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx || !d_src || !h_dst) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].device));
    HIP_TRY(ctx, hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return BRT_OK;
    });
}

}  // extern "C"

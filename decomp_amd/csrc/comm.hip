// C ABI: the handle's RCCL communicator (see include/decomp_hip.h, "multi-GPU").
//
// One process per GPU; rank 0 draws a unique id (dcp_comm_unique_id), the launcher's own channel carries
// its 128 bytes to the other ranks (decomp_amd.sharded uses torch.distributed's store for that and for
// nothing else), every rank calls dcp_comm_init.  After that the sharded solver loops
// (dcp_nmf_mu_sharded_*) and dcp_comm_allreduce_sum_* enqueue ncclAllReduce on the handle's own stream:
// no second stream, no event pair around the collective, no host round trip per step.
//
// librccl is resolved with dlopen/dlsym; <rccl/rccl.h> is used for its types only.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "comm.hpp"

using namespace dcp;

namespace {

struct RcclApi {
    void* dl = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
    bool ok = false;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy already mapped into this process first (torch's librccl has SONAME librccl.so.1): two
        // RCCL instances in one process would each claim the GPU's IPC / xGMI resources
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            api.dl = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (api.dl) break;
        }
        if (!api.dl) {
            const char* more[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
            for (const char* n : more) {
                api.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (api.dl) break;
            }
        }
        if (!api.dl) {
            const char* e = dlerror();
            api.error = std::string("librccl not found: ") + (e ? e : "dlopen failed");
            return;
        }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.dl, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.dl, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.dl, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.dl, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.dl, "ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(api.dl, "ncclGetVersion"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
            api.error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
            return;
        }
        api.ok = true;
    });
    return api;
}

int nccl_fail(dcp_handle* h, const char* what, ncclResult_t r) {
    RcclApi& api = rccl();
    const char* s = api.GetErrorString ? api.GetErrorString(r) : "?";
    return fail(h, DCP_ERR_COMM, std::string(what) + ": " + (s ? s : "?"));
}

}  // namespace

namespace dcp {

int comm_allreduce_sum(dcp_handle* h, void* buf, size_t count, int dtype) {
    if (h->comm_ext != nullptr) {   // the caller's own exchange (dcp_comm_set_external)
        const int rc = h->comm_ext(buf, (int64_t)count, dtype, reinterpret_cast<void*>(h->stream), h->comm_ext_user);
        if (rc != 0) return fail(h, DCP_ERR_COMM, "external all-reduce callback failed: " + std::to_string(rc));
        return DCP_OK;
    }
    if (!h->comm) return fail(h, DCP_ERR_COMM, "the handle has no communicator (dcp_comm_init)");
    RcclApi& api = rccl();
    const ncclDataType_t dt = dtype == COMM_F64 ? ncclFloat64 : ncclFloat32;
    ncclResult_t r = api.AllReduce(buf, buf, count, dt, ncclSum, reinterpret_cast<ncclComm_t>(h->comm), h->stream);
    if (r != ncclSuccess) return nccl_fail(h, "ncclAllReduce", r);
    return DCP_OK;
}

void comm_release(dcp_handle* h) {
    h->comm_ext = nullptr;
    h->comm_ext_user = nullptr;
    if (!h->comm) {
        h->comm_rank = 0;
        h->comm_world = 1;
        return;
    }
    RcclApi& api = rccl();
    (void)hipStreamSynchronize(h->stream);
    if (api.ok) (void)api.CommDestroy(reinterpret_cast<ncclComm_t>(h->comm));
    h->comm = nullptr;
    h->comm_rank = 0;
    h->comm_world = 1;
}

}  // namespace dcp

extern "C" {

int dcp_comm_unique_id(void* id_out, int64_t id_bytes) {
    if (!id_out || id_bytes < (int64_t)DCP_COMM_ID_BYTES) return DCP_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == DCP_COMM_ID_BYTES, "unique id size");
    RcclApi& api = rccl();
    if (!api.ok) return DCP_ERR_COMM;
    ncclUniqueId id;
    if (api.GetUniqueId(&id) != ncclSuccess) return DCP_ERR_COMM;
    memcpy(id_out, &id, sizeof(id));
    return DCP_OK;
}

int dcp_comm_init(dcp_handle* h, const void* unique_id, int rank, int world) {
    if (!h) return DCP_ERR_INVALID;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(h, DCP_ERR_INVALID, "bad rank / world / id");
    RcclApi& api = rccl();
    if (!api.ok) return fail(h, DCP_ERR_COMM, api.error);
    if (h->comm) comm_release(h);
    DCP_HIP_OK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = api.CommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return nccl_fail(h, "ncclCommInitRank", r);
    h->comm = c;
    h->comm_rank = rank;
    h->comm_world = world;
    return DCP_OK;
}

int dcp_comm_set_external(dcp_handle* h, dcp_allreduce_fn fn, void* user, int rank, int world) {
    if (!h) return DCP_ERR_INVALID;
    if (!fn || world < 1 || rank < 0 || rank >= world) return fail(h, DCP_ERR_INVALID, "bad callback / rank / world");
    comm_release(h);
    h->comm_ext = fn;
    h->comm_ext_user = user;
    h->comm_rank = rank;
    h->comm_world = world;
    return DCP_OK;
}

int dcp_memcpy(dcp_handle* h, void* dst, const void* src, int64_t bytes) {
    if (!h) return DCP_ERR_INVALID;
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(h, DCP_ERR_INVALID, "null pointer / negative size");
    if (bytes == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    DCP_HIP_OK(h, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, h->stream));
    return DCP_OK;
}

int dcp_comm_destroy(dcp_handle* h) {
    if (!h) return DCP_ERR_INVALID;
    comm_release(h);
    return DCP_OK;
}

int dcp_comm_info(dcp_handle* h, int* rank, int* world) {
    if (!h) return DCP_ERR_INVALID;
    const bool on = h->comm != nullptr || h->comm_ext != nullptr;
    if (rank) *rank = on ? h->comm_rank : 0;
    if (world) *world = on ? h->comm_world : 0;
    return DCP_OK;
}

int dcp_comm_allreduce_sum_f32(dcp_handle* h, float* buf, int64_t count) {
    if (!h) return DCP_ERR_INVALID;
    if (!buf || count < 0) return fail(h, DCP_ERR_INVALID, "null buffer / negative count");
    if (count == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    return comm_allreduce_sum(h, buf, (size_t)count, COMM_F32);
}

int dcp_comm_allreduce_sum_f64(dcp_handle* h, double* buf, int64_t count) {
    if (!h) return DCP_ERR_INVALID;
    if (!buf || count < 0) return fail(h, DCP_ERR_INVALID, "null buffer / negative count");
    if (count == 0) return DCP_OK;
    DCP_HIP_OK(h, hipSetDevice(h->device));
    return comm_allreduce_sum(h, buf, (size_t)count, COMM_F64);
}

}  // extern "C"

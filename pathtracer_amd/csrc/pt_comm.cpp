// Combining the per-GPU radiance sums: one ncclReduce(sum, float) over RCCL (xGMI inside a node) of the
// W x H x 4-float sum buffers into the root's, in place, on the renderers' own streams.
//
// The reference is single-GPU; what makes the partition exact is that a sample depends only on (pixel,
// iteration index) (pathtracer.frag.glsl:866) and that accumulation here is a float sum, whose alpha lane
// carries the sample count.  The scene is replicated, renderer g of G renders iterations g, g + G, ...
// (pt_set_partition), and nothing is exchanged until this reduce.
//
// RCCL is loaded when the first communicator is made (dlopen), so a single-GPU user of the library never
// needs it.  Which copy: the one that lies beside the HIP runtime this library is itself bound to.  A process
// may hold two HIP runtimes (a Python wheel that bundles its own libamdhip64 and librccl next to the system's
// ROCm); streams and device pointers of one are meaningless to the other, so the RCCL that reduces this
// library's buffers on this library's streams has to be the one linked against the same runtime.
#include "../../include/pathtracer_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "renderer_internal.h"

static_assert(PT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as the bytes of an ncclUniqueId");

struct pt_comm {
    ncclComm_t comm = nullptr;
    int device = 0;
    uint32_t rank = 0, world = 1;
};

namespace {

using pt_internal::set_error;

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;   // why loading failed
    std::string path;  // where the loaded library lies
};

// Directory of the shared object that holds `symbol` (with a trailing slash), empty when unknown.
std::string directory_of(const void *symbol)
{
    Dl_info info;
    if (!dladdr(symbol, &info) || !info.dli_fname) return std::string();
    const std::string file(info.dli_fname);
    const size_t slash = file.rfind('/');
    return slash == std::string::npos ? std::string() : file.substr(0, slash + 1);
}
std::string file_of(const void *symbol)
{
    Dl_info info;
    return (dladdr(symbol, &info) && info.dli_fname) ? std::string(info.dli_fname) : std::string("?");
}

Rccl &rccl_lib()
{
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        // beside the HIP runtime hipGetDevice resolves to from here, then by soname, then the usual place
        const std::string beside = directory_of(reinterpret_cast<const void *>(&hipGetDevice));
        std::vector<std::string> names;
        if (!beside.empty()) names.insert(names.end(), {beside + "librccl.so.1", beside + "librccl.so"});
        names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
        for (const std::string &name : names) {
            lib.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (lib.handle) break;
            const char *e = dlerror();
            lib.why = e ? e : "dlopen failed";
        }
        if (!lib.handle) return;
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(lib.handle, name);
            if (!p) {
                ok = false;
                lib.why = std::string("symbol missing: ") + name;
            }
            return p;
        };
        lib.GetUniqueId = reinterpret_cast<decltype(lib.GetUniqueId)>(sym("ncclGetUniqueId"));
        lib.CommInitRank = reinterpret_cast<decltype(lib.CommInitRank)>(sym("ncclCommInitRank"));
        lib.CommInitAll = reinterpret_cast<decltype(lib.CommInitAll)>(sym("ncclCommInitAll"));
        lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(sym("ncclCommDestroy"));
        lib.Reduce = reinterpret_cast<decltype(lib.Reduce)>(sym("ncclReduce"));
        lib.GroupStart = reinterpret_cast<decltype(lib.GroupStart)>(sym("ncclGroupStart"));
        lib.GroupEnd = reinterpret_cast<decltype(lib.GroupEnd)>(sym("ncclGroupEnd"));
        lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            dlclose(lib.handle);
            lib.handle = nullptr;
            return;
        }
        lib.path = file_of(reinterpret_cast<const void *>(lib.Reduce));
    });
    return lib;
}

// The loaded library, or null with the calling thread's error set.
Rccl *rccl()
{
    Rccl &lib = rccl_lib();
    if (lib.handle) return &lib;
    set_error(PT_ERR_DEVICE, "RCCL could not be loaded: " + lib.why);
    return nullptr;
}

int nccl_fail(const Rccl *lib, const char *what, ncclResult_t e)
{
    return set_error(PT_ERR_DEVICE, std::string(what) + ": " + lib->GetErrorString(e));
}

}  // namespace

extern "C" {

int pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES])
{
    if (!id) return set_error(PT_ERR_INVALID, "pt_comm_unique_id: null argument");
    Rccl *lib = rccl();
    if (!lib) return PT_ERR_DEVICE;
    ncclUniqueId u;
    const ncclResult_t e = lib->GetUniqueId(&u);
    if (e != ncclSuccess) return nccl_fail(lib, "ncclGetUniqueId", e);
    std::memcpy(id, u.internal, PT_COMM_ID_BYTES);
    return PT_OK;
}

int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], uint32_t rank, uint32_t world, int32_t device, pt_comm **out)
{
    if (!id || !out || world < 1 || rank >= world) return set_error(PT_ERR_INVALID, "pt_comm_create: need an id and rank < world");
    Rccl *lib = rccl();
    if (!lib) return PT_ERR_DEVICE;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return set_error(PT_ERR_DEVICE, "pt_comm_create: no current device");
    if (hipSetDevice(dev) != hipSuccess) return set_error(PT_ERR_DEVICE, "pt_comm_create: hipSetDevice failed");
    ncclUniqueId u;
    std::memcpy(u.internal, id, PT_COMM_ID_BYTES);
    pt_comm *c = new (std::nothrow) pt_comm();
    if (!c) return set_error(PT_ERR_INVALID, "out of memory");
    const ncclResult_t e = lib->CommInitRank(&c->comm, static_cast<int>(world), u, static_cast<int>(rank));
    if (e != ncclSuccess) {
        delete c;
        return nccl_fail(lib, "ncclCommInitRank", e);
    }
    c->device = dev;
    c->rank = rank;
    c->world = world;
    *out = c;
    return PT_OK;
}

int pt_comm_create_all(const int32_t *devices, uint32_t n, pt_comm **out)
{
    if (!devices || !out || n < 1 || n > 64) return set_error(PT_ERR_INVALID, "pt_comm_create_all: need 1..64 devices");
    Rccl *lib = rccl();
    if (!lib) return PT_ERR_DEVICE;
    std::vector<ncclComm_t> comms(n, nullptr);
    std::vector<int> devs(devices, devices + n);
    const ncclResult_t e = lib->CommInitAll(comms.data(), static_cast<int>(n), devs.data());
    if (e != ncclSuccess) return nccl_fail(lib, "ncclCommInitAll", e);
    for (uint32_t k = 0; k < n; ++k) {
        pt_comm *c = new (std::nothrow) pt_comm();
        if (!c) {
            // nothing half-made is handed out: the wrappers so far and every communicator go again
            for (uint32_t j = 0; j < k; ++j) {
                delete out[j];
                out[j] = nullptr;
            }
            for (uint32_t j = 0; j < n; ++j) {
                (void)hipSetDevice(devs[j]);
                (void)lib->CommDestroy(comms[j]);
            }
            return set_error(PT_ERR_INVALID, "out of memory");
        }
        c->comm = comms[k];
        c->device = devs[k];
        c->rank = k;
        c->world = n;
        out[k] = c;
    }
    return PT_OK;
}

int pt_comm_runtime_paths(char *out, size_t bytes)
{
    if (!out || bytes == 0) return set_error(PT_ERR_INVALID, "pt_comm_runtime_paths: null argument");
    Rccl *lib = rccl();
    if (!lib) return PT_ERR_DEVICE;
    // the HIP runtime RCCL calls into, as seen from inside RCCL: the object that holds the hipGetDevice it resolves to
    std::string rccl_hip = "?";
    if (void *sym = dlsym(lib->handle, "hipGetDevice")) rccl_hip = file_of(sym);
    const std::string text = "hip=" + file_of(reinterpret_cast<const void *>(&hipGetDevice)) + "\nrccl=" + lib->path + "\nrccl_hip=" + rccl_hip;
    std::strncpy(out, text.c_str(), bytes - 1);
    out[bytes - 1] = '\0';
    return PT_OK;
}

void pt_comm_destroy(pt_comm *c)
{
    if (!c) return;
    Rccl &lib = rccl_lib();
    if (lib.handle && c->comm) {
        (void)hipSetDevice(c->device);
        (void)lib.CommDestroy(c->comm);
    }
    delete c;
}

int pt_reduce_sum(pt_renderer *const *renderers, pt_comm *const *comms, uint32_t n, uint32_t root)
{
    if (!renderers || !comms || n < 1) return set_error(PT_ERR_INVALID, "pt_reduce_sum: null argument");
    Rccl *lib = rccl();
    if (!lib) return PT_ERR_DEVICE;
    std::vector<pt_internal::SumBuffer> bufs(n);
    for (uint32_t k = 0; k < n; ++k) {
        if (!renderers[k] || !comms[k]) return set_error(PT_ERR_INVALID, "pt_reduce_sum: null renderer or communicator");
        if (root >= comms[k]->world) return set_error(PT_ERR_INVALID, "pt_reduce_sum: root is not a rank of the communicator");
        if (!pt_internal::sum_buffer(renderers[k], &bufs[k])) return PT_ERR_STATE;
        if (bufs[k].device != comms[k]->device)
            return set_error(PT_ERR_INVALID, "pt_reduce_sum: renderer and communicator are on different devices");
        if (bufs[k].count != bufs[0].count) return set_error(PT_ERR_INVALID, "pt_reduce_sum: the renderers' image sizes differ");
    }
    if (n > 1) {
        // Several ranks driven from this one call: they must be ALL ranks of one communicator, each once.  A rank
        // left out would never issue its part of the collective and the others would wait for it on their streams.
        std::vector<bool> seen(n, false);
        for (uint32_t k = 0; k < n; ++k) {
            if (comms[k]->world != n)
                return set_error(PT_ERR_INVALID, "pt_reduce_sum: n renderers need the n communicators of one pt_comm_create_all");
            if (comms[k]->rank >= n || seen[comms[k]->rank]) return set_error(PT_ERR_INVALID, "pt_reduce_sum: ranks must be 0..n-1, each once");
            seen[comms[k]->rank] = true;
        }
    }
    // One collective: in a process that drives several GPUs the calls of all its ranks are issued as a group.
    ncclResult_t e = n > 1 ? lib->GroupStart() : ncclSuccess;
    if (e != ncclSuccess) return nccl_fail(lib, "ncclGroupStart", e);
    bool device_failed = false;
    for (uint32_t k = 0; k < n && e == ncclSuccess && !device_failed; ++k) {
        if (hipSetDevice(bufs[k].device) != hipSuccess) {
            device_failed = true;  // the group opened above is still closed below
            break;
        }
        e = lib->Reduce(bufs[k].data, bufs[k].data, bufs[k].count, ncclFloat, ncclSum, static_cast<int>(root), comms[k]->comm,
                        bufs[k].stream);
    }
    const ncclResult_t e2 = n > 1 ? lib->GroupEnd() : ncclSuccess;
    if (device_failed) return set_error(PT_ERR_DEVICE, "pt_reduce_sum: hipSetDevice failed");
    if (e != ncclSuccess) return nccl_fail(lib, "ncclReduce", e);
    if (e2 != ncclSuccess) return nccl_fail(lib, "ncclGroupEnd", e2);
    // The samples of the other ranks now live in the root's buffer: their own sums start again from zero (their
    // iteration counters go on), so that a later reduce adds only what was rendered since this one.
    for (uint32_t k = 0; k < n; ++k) {
        if (comms[k]->rank != root) {
            if (hipSetDevice(bufs[k].device) != hipSuccess ||
                hipMemsetAsync(bufs[k].data, 0, bufs[k].count * sizeof(float), bufs[k].stream) != hipSuccess)
                return set_error(PT_ERR_DEVICE, "pt_reduce_sum: clearing a reduced buffer failed");
        }
        pt_internal::main_stream_work_enqueued(renderers[k]);
    }
    return PT_OK;
}

}  // extern "C"

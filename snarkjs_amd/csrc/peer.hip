// snarkjs_amd/csrc/peer.hip — GPU-to-GPU exchange between the PROCESSES of a multi-GPU proof, at the C-ABI (include/zkmi.h: zkmi_ipc_*,
// zkmi_peer_copy). One process per GPU (zkmi_init binds one device); a chain owner exports the device buffer that holds its chain output,
// every other process maps it and pulls ITS slice device to device — over xGMI when the two GPUs differ, inside HBM when the processes share
// a device — instead of owner GPU -> pinned host pages -> the other GPUs over PCIe. The reference's analogue is ffjavascript handing chunk
// buffers to its workers and folding the results on the host (build/snarkjs.min.js:1@214651, @207729); what travels here is the same data.
//
// Handle layout (ZKMI_IPC_HANDLE_BYTES = 96): bytes 0..63 hipIpcMemHandle_t of the ALLOCATION that holds the pointer, 64..71 byte offset of
// the pointer inside it, 72..79 the exporter's pointer value, 80..83 a random per-process NONCE of the exporter (not its pid: pids repeat across
// PID namespaces and over time), 84..87 exporter device, 88..95 bytes visible from the pointer to the end of the allocation. A handle whose
// nonce is this process's own resolves to the original pointer (HIP refuses to open its own handles) — after checking that the pointer is
// inside a live device allocation of this process with that offset — so a single-process test and a world of one need no special case in
// the caller. Mappings are reference-counted per allocation AND per returned pointer: closing a pointer more often than it was opened is a
// no-op and can never unmap an allocation another open pointer still uses.
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <map>
#include "zkmi_common.hpp"

namespace zkmi {

struct IpcMapping { void* base = nullptr; int refs = 0; };
struct IpcPtr { std::string key; int refs = 0; };
static std::map<std::string, IpcMapping> g_ipc_open;           // by the 64 handle bytes: one mapping per exported allocation
static std::map<void*, IpcPtr> g_ipc_ptr;                      // pointer handed to the caller -> handle bytes, how often it was handed out
static hipStream_t g_copy_stream = nullptr;                    // zkmi_peer_copy(_async): a stream of its own (below)
static hipEvent_t g_copy_ev = nullptr;

// identifies THIS process in the handles it exports: 32 random bits, drawn once
static uint32_t ipc_nonce() {
    static const uint32_t v = [] {
        uint32_t x = 0;
        if (FILE* f = fopen("/dev/urandom", "rb")) { if (fread(&x, 1, 4, f) != 4) x = 0; fclose(f); }
        if (!x) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); x = (uint32_t)getpid() * 2654435761u ^ (uint32_t)ts.tv_nsec ^ (uint32_t)(uintptr_t)&x; }
        return x ? x : 1u;
    }();
    return v;
}

}  // namespace zkmi

using namespace zkmi;

extern "C" {

int zkmi_ipc_export(const void* d_ptr, uint8_t* handle) {
    ZK_TRY(require_ctx());
    if (!d_ptr || !handle) return fail(ZKMI_ERR_INVALID, "ipc_export: null argument");
    void* base = nullptr;
    size_t size = 0;
    ZK_HIP(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)d_ptr));
    hipIpcMemHandle_t h;
    ZK_HIP(hipIpcGetMemHandle(&h, base));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    memset(handle, 0, ZKMI_IPC_HANDLE_BYTES);
    memcpy(handle, &h, 64);
    const uint64_t off = (uint64_t)((const uint8_t*)d_ptr - (const uint8_t*)base), ptr = (uint64_t)(uintptr_t)d_ptr, avail = (uint64_t)size - off;
    const uint32_t pid = ipc_nonce(), dev = (uint32_t)ctx().device;
    memcpy(handle + 64, &off, 8); memcpy(handle + 72, &ptr, 8); memcpy(handle + 80, &pid, 4); memcpy(handle + 84, &dev, 4); memcpy(handle + 88, &avail, 8);
    return ZKMI_OK;
}

int zkmi_ipc_open(const uint8_t* handle, void** d_ptr, size_t* bytes_visible) {
    ZK_TRY(require_ctx());
    if (!handle || !d_ptr) return fail(ZKMI_ERR_INVALID, "ipc_open: null argument");
    uint64_t off, ptr, avail;
    uint32_t pid;
    memcpy(&off, handle + 64, 8); memcpy(&ptr, handle + 72, 8); memcpy(&pid, handle + 80, 4); memcpy(&avail, handle + 88, 8);
    if (bytes_visible) *bytes_visible = (size_t)avail;
    if (pid == ipc_nonce()) {                                   // our own export: the pointer itself, once it checks out as one of our allocations
        void* base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)(uintptr_t)ptr) != hipSuccess || (uint64_t)((uint8_t*)(uintptr_t)ptr - (uint8_t*)base) != off ||
            off + avail != (uint64_t)size)
            return fail(ZKMI_ERR_INVALID, "ipc_open: the handle names this process but not one of its device allocations");
        *d_ptr = (void*)(uintptr_t)ptr;
        return ZKMI_OK;
    }
    const std::string key((const char*)handle, 64);
    IpcMapping& m = g_ipc_open[key];
    if (!m.base) {
        hipIpcMemHandle_t h;
        memcpy(&h, handle, 64);
        const hipError_t e = hipIpcOpenMemHandle(&m.base, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { g_ipc_open.erase(key); return fail(ZKMI_ERR_HIP, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e)); }
        // the offset and the visible length come from the other process: hold them against the size of what was actually mapped (where the
        // runtime can tell; a mapping it cannot size is taken as exported)
        void* mb = nullptr;
        size_t msize = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t*)&mb, &msize, (hipDeviceptr_t)m.base) == hipSuccess && msize && (off > (uint64_t)msize || avail > (uint64_t)msize - off)) {
            (void)hipIpcCloseMemHandle(m.base);
            g_ipc_open.erase(key);
            return fail(ZKMI_ERR_INVALID, "ipc_open: offset / length of the handle exceed the mapped allocation");
        }
    }
    m.refs++;
    *d_ptr = (uint8_t*)m.base + off;
    IpcPtr& ip = g_ipc_ptr[*d_ptr];
    ip.key = key;
    ip.refs++;
    return ZKMI_OK;
}

int zkmi_ipc_close(void* d_ptr) {
    auto it = g_ipc_ptr.find(d_ptr);
    if (it == g_ipc_ptr.end()) return ZKMI_OK;                   // our own export, never opened, or closed as often as it was opened
    const std::string key = it->second.key;
    if (--it->second.refs <= 0) g_ipc_ptr.erase(it);
    auto mt = g_ipc_open.find(key);
    if (mt == g_ipc_open.end()) return ZKMI_OK;
    if (--mt->second.refs > 0) return ZKMI_OK;                  // other pointers into the same allocation are still open: the last close unmaps
    // nothing queued may still read the mapping: the library stream AND the copy stream of zkmi_peer_copy_async
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    if (g_copy_stream) (void)hipStreamSynchronize(g_copy_stream);
    (void)hipIpcCloseMemHandle(mt->second.base);
    g_ipc_open.erase(mt);
    for (auto p = g_ipc_ptr.begin(); p != g_ipc_ptr.end();) { if (p->second.key == key) p = g_ipc_ptr.erase(p); else ++p; }      // none can be left (their refs were part of the count)
    return ZKMI_OK;
}

// d_dst (this process's device) <- d_src (a pointer from zkmi_ipc_open, or any device pointer of this process). The copies run on a stream of
// their own: the library stream of a shard process is full of the witness-side accumulations at the moment the slices are pulled
// (zkmi_groth16_sums_w_dev was enqueued first, on purpose), and a copy queued behind them would cross xGMI only after they have finished — on the
// critical path of the H half instead of underneath the witness half.
//   zkmi_peer_copy        complete on return (waits for the copy stream only);
//   zkmi_peer_copy_async  queued; zkmi_peer_fence() then makes the LIBRARY stream wait for everything queued so far (an event, no host wait):
//                         what is enqueued on the library stream afterwards (zkmi_groth16_join_abc_dev) sees the data.
static int copy_stream_ready() {
    if (g_copy_stream) return ZKMI_OK;
    int lo = 0, hi = 0;
    ZK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    ZK_HIP(hipStreamCreateWithPriority(&g_copy_stream, hipStreamNonBlocking, hi));
    ZK_HIP(hipEventCreateWithFlags(&g_copy_ev, hipEventDisableTiming));
    return ZKMI_OK;
}
int zkmi_peer_copy_async(void* d_dst, const void* d_src, size_t bytes) {
    ZK_TRY(require_ctx());
    if (!bytes) return ZKMI_OK;
    if (!d_dst || !d_src) return fail(ZKMI_ERR_INVALID, "peer_copy: null argument");
    ZK_TRY(copy_stream_ready());
    ZK_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, g_copy_stream));
    return ZKMI_OK;
}
int zkmi_peer_copy(void* d_dst, const void* d_src, size_t bytes) {
    ZK_TRY(zkmi_peer_copy_async(d_dst, d_src, bytes));
    if (bytes) ZK_HIP(hipStreamSynchronize(g_copy_stream));
    return ZKMI_OK;
}
int zkmi_peer_fence(void) {
    ZK_TRY(require_ctx());
    if (!g_copy_stream) return ZKMI_OK;                          // nothing was ever queued
    ZK_HIP(hipEventRecord(g_copy_ev, g_copy_stream));
    ZK_HIP(hipStreamWaitEvent(ctx().stream, g_copy_ev, 0));
    return ZKMI_OK;
}

}  // extern "C"

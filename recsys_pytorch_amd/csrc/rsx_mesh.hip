// rsx_mesh.hip -- the step's exchange of the item gradients as a DIRECT full-mesh reduce-scatter + all-gather over xGMI
// ("reduce-scatter + all-gather with all 7 peers concurrently rather than a ring"), without RCCL.
//
// The reference has no multi-device code (main.py:24-27 pins one device); what must be preserved is its batch-synchronous
// step (models/MF.py:64-68): every rank applies the SAME summed item gradient before the next step reads the item table.
//
// Every rank maps the peers' item table Q, gradient buffer G and a small mailbox into its address space
// (hipIpcGetMemHandle / hipIpcOpenMemHandle; xGMI is point to point, so the N - 1 peers are read over N - 1 different
// links at once).  The rows [first, first + rows) of one exchange are cut into `world` equal slices; rank r OWNS slice r:
//   phase 1  signal READY(seq) to every peer (my G is complete) -> wait for every peer's READY(seq) ->
//            sum = G_mine[slice r] + sum over the peers p of G_p[slice r], read directly from the peers ->
//            Q_mine[slice r] -= lr * sum;  G_mine[slice r] = 0
//   phase 2  signal APPLIED(seq) -> wait for every peer's APPLIED(seq) ->
//            for every other slice q: Q_mine[slice q] = Q_q[slice q] (copied from its owner), G_mine[slice q] = 0
// Every item row is computed by exactly ONE rank, in one fixed order (own rows first, then the peers in rank order), and copied to
// the others: the replicas are identical by construction.  No host thread takes part after the launches are queued.
//
// Ordering between the ranks (all device side).  `seq` counts the exchanges of a mesh; every rank issues the same sequence.
// The mailbox of rank r holds, per peer p, the latest READY / APPLIED sequence number p has signalled TO r (p writes it with a
// system-scope release store into r's memory; r polls its own memory).  What a signal covers is what the signalling rank's
// stream had completed before the signal kernel: kernels end with their writes written back to memory (the XCDs' L2s are
// not shared, so the end-of-kernel release is a write-back on gfx950), a peer reads memory over xGMI, and every workgroup that
// reads a peer's rows first executes a system-scope ACQUIRE after it has seen the flag (stale lines of the same rows from
// the step before are dropped from its L2).
//   * G_p[slice r] is read by r in phase 1 after READY_p(seq); p clears it in ITS phase 2, after APPLIED_r(seq): r is done.
//   * Q_q[slice q] is read by the others in phase 2 after APPLIED_q(seq); q writes it next in phase 1 of a LATER exchange, which
//     waits for every peer's READY of that exchange -- signalled by a peer only after its own phase 2 of this one (stream order).
// A wait that lasts longer than the mesh's limit (default 20 s) gives up, raises the mesh's error word and lets the kernel
// finish -- wrong rows, loudly reported by rsx_mesh_check, never a hung GPU.
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <vector>

#include "rsx_common.h"

namespace {

constexpr int kMaxWorld = 16;
constexpr int kMeshBlock = 256;
constexpr int kFlagStride = 16;       // uint32 per flag slot: one 64-byte line each

struct MeshDesc {                     // what a rank tells the others (RSX_MESH_DESC_BYTES, plain old data)
    hipIpcMemHandle_t hQ, hG, hF;     // handles of the ALLOCATIONS that hold Q, G and the mailbox
    uint64_t offQ, offG;              // byte offsets of the tables inside those allocations
    uint64_t baseQ, baseG;            // the exporting process's addresses of the allocations (to recognise a shared one)
    int64_t rows;
    int32_t d, device;
    int64_t pid;
};
static_assert(sizeof(MeshDesc) <= RSX_MESH_DESC_BYTES, "RSX_MESH_DESC_BYTES too small");

struct PeerPtrs {
    const float *Q[kMaxWorld];        // peers' item tables (own entry: the local one)
    const float *G[kMaxWorld];
    uint32_t *flags[kMaxWorld];       // peers' mailboxes (own entry: the local one)
};

}  // namespace

struct rsx_mesh {
    int rank = -1, world = 0, device = 0;
    float *Q = nullptr, *G = nullptr;
    int64_t rows = 0;
    int d = 0;
    uint32_t *flags = nullptr;        // own mailbox: [kMaxWorld][2] slots of kFlagStride words + the error word at the end
    uint32_t seq = 0;
    uint64_t limit_ticks = 20ull * 100000000ull;     // wall_clock64 runs at 100 MHz
    PeerPtrs peers{};
    std::vector<void *> opened;       // what hipIpcOpenMemHandle returned (closed by rsx_mesh_destroy)
    bool connected = false;
    int export_retries = 0;           // hipIpcGetMemHandle calls that failed before the one that succeeded (0 = every export at the first try)
};

namespace {

constexpr int kReady = 0, kApplied = 1;
__host__ __device__ inline int flag_slot(int from, int kind) { return (from * 2 + kind) * kFlagStride; }
constexpr int kErrorWord = kMaxWorld * 2 * kFlagStride;
constexpr size_t kMailboxBytes = (size_t)(kErrorWord + kFlagStride) * sizeof(uint32_t);

// one thread per peer: everything this rank's stream completed before this kernel is in memory; tell the peer
__global__ void mesh_signal_kernel(PeerPtrs p, int rank, int world, int kind, uint32_t seq)
{
    const int q = threadIdx.x;
    if (q >= world || q == rank) return;
    __threadfence_system();
    __hip_atomic_store(p.flags[q] + flag_slot(rank, kind), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// every workgroup: wait until every peer has signalled `kind` for exchange `seq` (own mailbox), then acquire
__device__ __forceinline__ void mesh_wait(uint32_t *flags, int rank, int world, int kind, uint32_t seq, uint64_t limit)
{
    const int q = threadIdx.x;
    if (limit != 0 && q < world && q != rank) {      // (limit 0: the development build's one-GPU traffic model -- there are no peers to wait for)
        const uint64_t t0 = wall_clock64();
        // (seq is monotonic and compared as a signed distance: it may wrap)
        while ((int32_t)(__hip_atomic_load(flags + flag_slot(q, kind), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (wall_clock64() - t0 > limit) { atomicOr(flags + kErrorWord, 1u << kind); break; }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    __threadfence_system();                          // acquire at system scope in EVERY wavefront that goes on to read peer rows
}

// phase 1: this rank's slice [lo4, hi4) (in float4 units from the table's base): sum the peers' partial sums onto mine, apply, clear
// (development build's one-GPU traffic model only: `pace` > 0 spreads a kernel's trips evenly over `pace` ticks of the 100 MHz clock -- the
//  time the wire would take -- so that the kernel's HBM traffic and the wire time overlap as they do when the rows really come over xGMI)
__device__ __forceinline__ void mesh_pace(uint64_t t0, uint64_t pace, int64_t done, int64_t total)
{
#ifdef RSX_ABLATE
    if (pace == 0) return;
    const uint64_t due = (uint64_t)((double)pace * (double)done / (double)total);
    while (wall_clock64() - t0 < due) __builtin_amdgcn_s_sleep(8);
#endif
}

__global__ __launch_bounds__(kMeshBlock) void mesh_reduce_apply_kernel(PeerPtrs p, uint32_t *flags, int rank, int world, uint32_t seq,
                                                                       uint64_t limit, float4 *Q, float4 *G, int64_t lo4, int64_t hi4,
                                                                       float lr, uint64_t pace)
{
    mesh_wait(flags, rank, world, kReady, seq, limit);
    const int64_t stride = (int64_t)gridDim.x * kMeshBlock;
    const uint64_t t0 = pace ? wall_clock64() : 0;
    for (int64_t n = lo4 + (int64_t)blockIdx.x * kMeshBlock + threadIdx.x; n < hi4; n += stride) {
        mesh_pace(t0, pace, n - lo4, hi4 - lo4);
        float4 s = G[n];
        float4 v[kMaxWorld];
        // the peers' quads are requested together (one per link), then added in rank order: the same order on whatever rank owns the row
#pragma unroll
        for (int q = 0; q < kMaxWorld; ++q)
            if (q < world && q != rank) v[q] = reinterpret_cast<const float4 *>(p.G[q])[n];
#pragma unroll
        for (int q = 0; q < kMaxWorld; ++q)
            if (q < world && q != rank) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
        float4 w = Q[n];
        w.x = fmaf(-lr, s.x, w.x); w.y = fmaf(-lr, s.y, w.y); w.z = fmaf(-lr, s.z, w.z); w.w = fmaf(-lr, s.w, w.w);
        Q[n] = w;
        G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// phase 2: every other rank's slice: copy the updated rows from their owner, clear my partial sums of them (the owner has read them)
__global__ __launch_bounds__(kMeshBlock) void mesh_gather_kernel(PeerPtrs p, uint32_t *flags, int rank, int world, uint32_t seq,
                                                                 uint64_t limit, float4 *Q, float4 *G, int64_t first4, int64_t slice4,
                                                                 int64_t end4, uint64_t pace)
{
    mesh_wait(flags, rank, world, kApplied, seq, limit);
    const int64_t stride = (int64_t)gridDim.x * kMeshBlock;
    const uint64_t t0 = pace ? wall_clock64() : 0;
    // the slices of the other ranks, walked interleaved (thread t of a trip reads from owner (t / slice) -- consecutive workgroups hit
    // different links only through the grid stride; with N - 1 peers and hundreds of workgroups every link is busy)
    for (int64_t n = first4 + (int64_t)blockIdx.x * kMeshBlock + threadIdx.x; n < end4; n += stride) {
        mesh_pace(t0, pace, n - first4, end4 - first4);
        const int owner = (int)((n - first4) / slice4);
        if (owner == rank) continue;
        Q[n] = reinterpret_cast<const float4 *>(p.Q[owner])[n];
        G[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

int mesh_fail(const char *what, hipError_t e)
{
    rsx_set_error("%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return RSX_E_HIP;
}

// What a rank exports must be memory the library can vouch for: a device allocation of the CURRENT device, whole (the handle names
// the allocation, the table is an offset into it).  Every HIP call that can refuse has its own message with its arguments.
struct Exported {
    void *base = nullptr;
    size_t size = 0;
    hipIpcMemHandle_t handle;
    int attempts = 0;
};

constexpr int kExportAttempts = 4;

// Memory the library allocated FOR exchange (rsx_mesh_alloc): its IPC handle was taken when it was allocated -- an allocation the runtime
// refuses to export never leaves rsx_mesh_alloc -- and is kept, so a mesh over it makes no export call at all.
// rsx_mesh_free does NOT return the memory to the runtime: the block (with its handle) waits for the next rsx_mesh_alloc of its size.  An
// exporter that frees an allocation its peers had mapped and then gets the SAME address back from hipMalloc hands out a handle the peers
// resolve to the OLD, freed memory (round 6, tools/mesh_stress.py --empty-cache: wrong sums in every mesh after the first) -- or is refused
// the export outright (the driver's round-5 run; a round-6 suite run).  A block that never dies keeps one valid handle for good.
struct Owned { size_t size; hipIpcMemHandle_t handle; bool in_use; };
std::mutex g_owned_mu;
std::map<void *, Owned> g_owned;              // by allocation base
std::vector<void *> g_refused;                // allocations the runtime would not export: kept (their addresses are not handed out again)

bool owned_lookup(const void *ptr, size_t bytes_needed, Exported *x)
{
    std::lock_guard<std::mutex> lock(g_owned_mu);
    auto it = g_owned.upper_bound(const_cast<void *>(ptr));
    if (it == g_owned.begin()) return false;
    --it;
    const char *base = (const char *)it->first;
    if (!it->second.in_use || (const char *)ptr < base || (const char *)ptr + bytes_needed > base + it->second.size) return false;
    x->base = it->first; x->size = it->second.size; x->handle = it->second.handle; x->attempts = 1;
    return true;
}

int export_allocation(const char *what, const void *ptr, size_t bytes_needed, int device, Exported *x)
{
    if (owned_lookup(ptr, bytes_needed, x)) return RSX_OK;     // (allocated by rsx_mesh_alloc: proven exportable, handle cached)
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    hipError_t e = hipPointerGetAttributes(&at, ptr);
    if (e != hipSuccess) {
        rsx_set_error("rsx_mesh_local: hipPointerGetAttributes(%s = %p) failed: %s -- not a pointer this process allocated on a GPU", what,
                      ptr, hipGetErrorString(e));
        (void)hipGetLastError();
        return RSX_E_HIP;
    }
    if (at.type != hipMemoryTypeDevice || at.device != device) {
        rsx_set_error("rsx_mesh_local: %s = %p is not plain device memory of the current device %d (memory type %d, device %d): only "
                      "hipMalloc'ed memory of this GPU can be mapped by the peers", what, ptr, device, (int)at.type, at.device);
        return RSX_E_INVALID;
    }
    e = hipMemGetAddressRange((hipDeviceptr_t *)&x->base, &x->size, (hipDeviceptr_t)ptr);
    if (e != hipSuccess || x->base == nullptr) {
        rsx_set_error("rsx_mesh_local: hipMemGetAddressRange(%s = %p) failed: %s", what, ptr, hipGetErrorString(e));
        (void)hipGetLastError();
        return RSX_E_HIP;
    }
    const size_t off = (size_t)((const char *)ptr - (const char *)x->base);
    if (off + bytes_needed > x->size) {
        rsx_set_error("rsx_mesh_local: %s = %p (+%zu bytes) does not lie inside ONE allocation (base %p, %zu bytes): a table stitched from "
                      "several mappings cannot be exported by one handle", what, ptr, bytes_needed, x->base, x->size);
        return RSX_E_INVALID;
    }
    // (the runtime has been seen to refuse an export while peers were still detaching from an earlier export of the same allocation:
    //  bounded retries; how many failed first is reported by rsx_mesh_export_retries)
    for (x->attempts = 1;; ++x->attempts) {
        e = hipIpcGetMemHandle(&x->handle, x->base);
        if (e == hipSuccess) return RSX_OK;
        (void)hipGetLastError();
        if (x->attempts >= kExportAttempts) break;
        usleep(50000 * x->attempts);
    }
    rsx_set_error("rsx_mesh_local: hipIpcGetMemHandle(allocation of %s: base %p, %zu bytes; table at offset %zu, %zu bytes) failed %d "
                  "times: %s (HSA_ENABLE_IPC_MODE_LEGACY=%s).  The runtime refuses to export THIS allocation (seen for pooled allocations of "
                  "the caller's allocator in processes that had mapped and unmapped peers' memory before): keep the tables in memory from "
                  "rsx_mesh_alloc, which hands out only allocations it has exported", what, x->base, x->size, off, bytes_needed, x->attempts,
                  hipGetErrorString(e), getenv("HSA_ENABLE_IPC_MODE_LEGACY") ? getenv("HSA_ENABLE_IPC_MODE_LEGACY") : "unset");
    return RSX_E_HIP;
}

}  // namespace

// (uncached: the mailbox -- polled by this rank while the peers store into it -- in uncached device memory where the runtime has it)
static int mesh_alloc_impl(int64_t bytes, bool uncached, void **out)
{
    RSX_CHECK_ARG(out != nullptr && bytes > 0, "bad size");
    *out = nullptr;
    {   // a block of this size that an earlier rsx_mesh_free handed back: the same allocation, the same handle
        void *reuse = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_owned_mu);
            for (auto &kv : g_owned)
                if (!kv.second.in_use && kv.second.size == (size_t)bytes) { kv.second.in_use = true; reuse = kv.first; break; }
        }
        if (reuse != nullptr) {
            hipError_t e = hipMemset(reuse, 0, (size_t)bytes);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) { rsx_set_error("rsx_mesh_alloc: clearing %lld bytes failed: %s", (long long)bytes, hipGetErrorString(e)); (void)hipGetLastError(); return RSX_E_HIP; }
            *out = reuse;
            return RSX_OK;
        }
    }
    constexpr int kTries = 8;
    hipError_t last = hipSuccess;
    for (int t = 0; t < kTries; ++t) {
        void *p = nullptr;
        hipError_t e = hipErrorUnknown;
        if (uncached) {
            e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
            if (e != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        }
        if (e != hipSuccess) e = hipMalloc(&p, (size_t)bytes);
        if (e != hipSuccess) { rsx_set_error("rsx_mesh_alloc: hipMalloc(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(e)); (void)hipGetLastError(); return RSX_E_HIP; }
        Owned o;
        o.size = (size_t)bytes; o.in_use = true;
        e = hipIpcGetMemHandle(&o.handle, p);
        if (e == hipSuccess) {
            e = hipMemset(p, 0, (size_t)bytes);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) { (void)hipFree(p); rsx_set_error("rsx_mesh_alloc: clearing %lld bytes failed: %s", (long long)bytes, hipGetErrorString(e)); (void)hipGetLastError(); return RSX_E_HIP; }
            std::lock_guard<std::mutex> lock(g_owned_mu);
            g_owned[p] = o;
            *out = p;
            return RSX_OK;
        }
        // the runtime will not export this allocation: keep it (so that the next hipMalloc returns ANOTHER address) and try again
        (void)hipGetLastError();
        last = e;
        std::lock_guard<std::mutex> lock(g_owned_mu);
        g_refused.push_back(p);
    }
    rsx_set_error("rsx_mesh_alloc: hipIpcGetMemHandle refused %d fresh allocations of %lld bytes in a row: %s (HSA_ENABLE_IPC_MODE_LEGACY=%s)", kTries,
                  (long long)bytes, hipGetErrorString(last), getenv("HSA_ENABLE_IPC_MODE_LEGACY") ? getenv("HSA_ENABLE_IPC_MODE_LEGACY") : "unset");
    return RSX_E_HIP;
}

RSX_API int rsx_mesh_alloc(int64_t bytes, void **out) { return mesh_alloc_impl(bytes, false, out); }

RSX_API int rsx_mesh_free(void *p)
{
    if (p == nullptr) return RSX_OK;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lock(g_owned_mu);
    auto it = g_owned.find(p);
    if (it == g_owned.end() || !it->second.in_use) { rsx_set_error("rsx_mesh_free: invalid argument: not a live allocation of rsx_mesh_alloc"); return RSX_E_INVALID; }
    it->second.in_use = false;                       // (kept for the next rsx_mesh_alloc of this size: see Owned)
    return RSX_OK;
}

RSX_API int rsx_mesh_alloc_refused(void)      /* allocations rsx_mesh_alloc had to set aside because the runtime would not export them */
{
    std::lock_guard<std::mutex> lock(g_owned_mu);
    return (int)g_refused.size();
}

RSX_API int rsx_mesh_local(float *Q, float *G, int64_t rows, int d, void *desc_out, rsx_mesh **out)
{
    RSX_CHECK_ARG(Q && G && desc_out && out, "null pointer");
    RSX_CHECK_ARG(rows > 0 && rsx_dim_ok(d), "bad shape");
    RSX_CHECK_ARG(Q != G, "Q and G are the same buffer");
    rsx_mesh *m = new (std::nothrow) rsx_mesh();
    if (m == nullptr) { rsx_set_error("rsx_mesh_local: out of memory"); return RSX_E_INVALID; }
    m->Q = Q; m->G = G; m->rows = rows; m->d = d;
    hipError_t e = hipGetDevice(&m->device);
    if (e != hipSuccess) { const int rc = mesh_fail("rsx_mesh_local: hipGetDevice", e); rsx_mesh_destroy(m); return rc; }
    // the mailbox: polled by this rank's kernels while the peers store into it (their stores and this rank's loads are system-scope
    // atomics).  From rsx_mesh_alloc like the tables should be: an allocation that has been exported already, zero-filled.
    {
        void *mb = nullptr;
        const int rc = mesh_alloc_impl((int64_t)kMailboxBytes, true, &mb);
        if (rc != RSX_OK) { rsx_mesh_destroy(m); return rc; }
        m->flags = (uint32_t *)mb;
    }
    const size_t table_bytes = (size_t)rows * (size_t)d * sizeof(float);
    Exported xQ, xG, xF;
    int rc = export_allocation("Q", Q, table_bytes, m->device, &xQ);
    if (rc == RSX_OK) {
        // both tables inside ONE allocation (a pooled segment of the caller's allocator): exported once
        hipDeviceptr_t bG = nullptr;
        size_t szG = 0;
        if (hipMemGetAddressRange(&bG, &szG, (hipDeviceptr_t)G) == hipSuccess && (void *)bG == xQ.base &&
            (size_t)((char *)G - (char *)bG) + table_bytes <= szG) {
            xG = xQ; xG.attempts = 1;
        } else {
            (void)hipGetLastError();
            rc = export_allocation("G", G, table_bytes, m->device, &xG);
        }
    }
    if (rc == RSX_OK) rc = export_allocation("the mailbox", m->flags, kMailboxBytes, m->device, &xF);
    if (rc != RSX_OK) { rsx_mesh_destroy(m); return rc; }
    if (xF.base != (void *)m->flags) {
        rsx_set_error("rsx_mesh_local: the mailbox %p is not at the start of its allocation (base %p)", (void *)m->flags, xF.base);
        rsx_mesh_destroy(m);
        return RSX_E_INVALID;
    }
    m->export_retries = (xQ.attempts - 1) + (xG.attempts - 1) + (xF.attempts - 1);
    MeshDesc dsc;
    memset(&dsc, 0, sizeof(dsc));
    dsc.hQ = xQ.handle; dsc.hG = xG.handle; dsc.hF = xF.handle;
    dsc.offQ = (uint64_t)((char *)Q - (char *)xQ.base); dsc.offG = (uint64_t)((char *)G - (char *)xG.base);
    dsc.baseQ = (uint64_t)(uintptr_t)xQ.base; dsc.baseG = (uint64_t)(uintptr_t)xG.base;
    dsc.rows = rows; dsc.d = d; dsc.device = m->device; dsc.pid = (int64_t)getpid();
    memset(desc_out, 0, RSX_MESH_DESC_BYTES);
    memcpy(desc_out, &dsc, sizeof(dsc));
    *out = m;
    return RSX_OK;
}

RSX_API int rsx_mesh_connect(rsx_mesh *m, int rank, int world, const void *all_desc)
{
    RSX_CHECK_ARG(m != nullptr && all_desc != nullptr, "null pointer");
    RSX_CHECK_ARG(world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world, "rank / world out of range (at most 16 ranks)");
    RSX_CHECK_ARG(!m->connected, "already connected");
    m->rank = rank; m->world = world;
    for (int q = 0; q < world; ++q) {
        MeshDesc dsc;
        memcpy(&dsc, (const char *)all_desc + (size_t)q * RSX_MESH_DESC_BYTES, sizeof(dsc));
        RSX_CHECK_ARG(dsc.rows == m->rows && dsc.d == m->d, "the ranks' tables differ in shape");
        if (q == rank) {
            RSX_CHECK_ARG(dsc.pid == (int64_t)getpid(), "descriptor of this rank does not come from this process");
            m->peers.Q[q] = m->Q; m->peers.G[q] = m->G; m->peers.flags[q] = m->flags;
            continue;
        }
        RSX_CHECK_ARG(dsc.pid != (int64_t)getpid(), "two ranks in one process: hipIpcOpenMemHandle cannot open its own handle");
        void *bQ = nullptr, *bG = nullptr, *bF = nullptr;
        const char *which = "Q";
        hipError_t e = hipIpcOpenMemHandle(&bQ, dsc.hQ, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) {
            m->opened.push_back(bQ);
            if (dsc.baseG == dsc.baseQ) bG = bQ;           // both tables in ONE allocation of the peer: opened once
            else { which = "G"; e = hipIpcOpenMemHandle(&bG, dsc.hG, hipIpcMemLazyEnablePeerAccess); if (e == hipSuccess) m->opened.push_back(bG); }
        }
        if (e == hipSuccess) { which = "mailbox"; e = hipIpcOpenMemHandle(&bF, dsc.hF, hipIpcMemLazyEnablePeerAccess); if (e == hipSuccess) m->opened.push_back(bF); }
        if (e != hipSuccess) {
            rsx_set_error("rsx_mesh_connect: rank %d: hipIpcOpenMemHandle(rank %d's %s allocation; its pid %lld, device %d) failed: %s", rank, q,
                          which, (long long)dsc.pid, dsc.device, hipGetErrorString(e));
            (void)hipGetLastError();
            return RSX_E_HIP;
        }
        m->peers.Q[q] = (const float *)((char *)bQ + dsc.offQ);
        m->peers.G[q] = (const float *)((char *)bG + dsc.offG);
        m->peers.flags[q] = (uint32_t *)bF;
    }
    m->connected = true;
    return RSX_OK;
}

#ifdef RSX_ABLATE
// DEVELOPMENT BUILD ONLY: the HBM side of a world-W exchange on ONE GPU (tools/exchange_model_schedules.sh).  A mesh of one
// rank then behaves as rank 0 of W: it sums ITS 1/W of the rows from W "peers" (all of them its own G: W reads of the slice, as when the
// rank reads 7 peers and serves 7), applies that slice, and copies the other (W - 1)/W of the rows of Q onto themselves (the all-gather's
// writes) while clearing G -- each of the two phases spread over `delay_us` (the wire's time), its HBM traffic under it.  TIMING ONLY: the sums are W times too
// large (lr is divided by W to keep the tables finite).
static int g_mesh_model_world = 0, g_mesh_model_delay_us = 0;
RSX_API int rsx_debug_set_mesh_model(int world, int delay_us_per_phase)
{
    g_mesh_model_world = (world >= 2 && world <= kMaxWorld) ? world : 0;
    g_mesh_model_delay_us = delay_us_per_phase > 0 ? delay_us_per_phase : 0;
    return RSX_OK;
}
#endif

RSX_API int rsx_mesh_exchange_apply(rsx_mesh *m, int64_t first_row, int64_t rows, float lr, rsx_stream_t stream)
{
    RSX_CHECK_ARG(m != nullptr && m->connected, "mesh not connected");
    RSX_CHECK_ARG(first_row >= 0 && rows > 0 && first_row + rows <= m->rows, "rows out of range");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t seq = ++m->seq;
    int world = m->world, rank = m->rank;
    PeerPtrs peers = m->peers;
    uint64_t limit = m->limit_ticks;
#ifdef RSX_ABLATE
    const bool model = g_mesh_model_world > 1 && m->world == 1;
    if (model) {
        world = g_mesh_model_world; rank = 0; limit = 0; lr /= (float)world;
        for (int q = 0; q < world; ++q) { peers.Q[q] = m->Q; peers.G[q] = m->G; peers.flags[q] = m->flags; }
    }
    const uint64_t pace = model ? (uint64_t)g_mesh_model_delay_us * 100ull * (uint64_t)rows / (uint64_t)m->rows : 0ull;
#else
    const uint64_t pace = 0;
#endif
    const int64_t d4 = m->d / 4;
    const int64_t slice = ceil_div64(rows, world);                       // rows per owner (the last slices may be short or empty)
    const int64_t first4 = first_row * d4, end4 = (first_row + rows) * d4, slice4 = slice * d4;
    int64_t lo4 = first4 + (int64_t)rank * slice4, hi4 = lo4 + slice4;
    if (lo4 > end4) lo4 = end4;
    if (hi4 > end4) hi4 = end4;
    // few workgroups, like a collective's channels: the exchange runs beside the other ranges' step kernels
    const int cus = g_rsx_mesh_blocks > 0 ? g_rsx_mesh_blocks : rsx_num_cus();
    auto grid = [&](int64_t n4) { int64_t g = ceil_div64(n4, (int64_t)kMeshBlock * 4); if (g > cus) g = cus; return (unsigned)(g < 1 ? 1 : g); };
    if (world > 1) hipLaunchKernelGGL(mesh_signal_kernel, dim3(1), dim3(64), 0, st, peers, rank, world, kReady, seq);
    hipLaunchKernelGGL(mesh_reduce_apply_kernel, dim3(grid(hi4 - lo4)), dim3(kMeshBlock), 0, st, peers, m->flags, rank, world, seq,
                       limit, (float4 *)m->Q, (float4 *)m->G, lo4, hi4, lr, pace);
    if (world > 1) {
        hipLaunchKernelGGL(mesh_signal_kernel, dim3(1), dim3(64), 0, st, peers, rank, world, kApplied, seq);
        hipLaunchKernelGGL(mesh_gather_kernel, dim3(grid(end4 - first4)), dim3(kMeshBlock), 0, st, peers, m->flags, rank, world, seq,
                           limit, (float4 *)m->Q, (float4 *)m->G, first4, slice4, end4, pace);
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
}

RSX_API int rsx_mesh_set_wait_limit(rsx_mesh *m, double seconds)
{
    RSX_CHECK_ARG(m != nullptr && seconds > 0.0 && seconds < 3600.0, "limit must be in (0, 3600) seconds");
    m->limit_ticks = (uint64_t)(seconds * 1e8);
    if (m->limit_ticks == 0) m->limit_ticks = 1;      // (0 is the development build's "no peers" marker)
    return RSX_OK;
}

RSX_API int rsx_mesh_info(const rsx_mesh *m, int *rank, int *world, int64_t *exchanges)
{
    RSX_CHECK_ARG(m != nullptr, "null mesh");
    if (rank) *rank = m->rank;
    if (world) *world = m->world;
    if (exchanges) *exchanges = (int64_t)m->seq;
    return RSX_OK;
}

RSX_API int rsx_mesh_export_retries(const rsx_mesh *m)
{
    return m == nullptr ? -1 : m->export_retries;
}

// (internal, rsx_common.h) what the mesh was built over: the native loop refuses a mesh over other tables than its own
void rsx_mesh_tables(const rsx_mesh *m, const float **Q, const float **G, int64_t *rows, int *d)
{
    *Q = m->Q; *G = m->G; *rows = m->rows; *d = m->d;
}

RSX_API int rsx_mesh_check(rsx_mesh *m, rsx_stream_t stream)
{
    RSX_CHECK_ARG(m != nullptr, "null mesh");
    uint32_t h = 0;
    hipError_t e = hipMemcpyAsync(&h, m->flags + kErrorWord, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return mesh_fail("rsx_mesh_check", e);
    if (h != 0) {
        rsx_set_error("rsx_mesh_check: rank %d gave up waiting for a peer's %s%s signal (a peer is stuck or gone): item rows of that exchange are wrong",
                      m->rank, (h & 1u) ? "READY" : "", (h & 2u) ? " APPLIED" : "");
        return RSX_E_INVALID;
    }
    return RSX_OK;
}

RSX_API void rsx_mesh_destroy(rsx_mesh *m)
{
    if (m == nullptr) return;
    (void)hipDeviceSynchronize();
    for (void *p : m->opened) (void)hipIpcCloseMemHandle(p);
    if (m->flags) (void)rsx_mesh_free(m->flags);
    delete m;
}

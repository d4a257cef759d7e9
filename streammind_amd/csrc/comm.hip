// Gated-token exchange between the GPUs of one node by DIRECT PEER WRITES over xGMI (the reference's
// allgather_diff_shape, /root/reference/streammind/dist.py:122-146: an all-gather of per-rank row counts, then of the rows padded to
// the largest count).
//
// xGMI is point to point: every GPU has its own link to every other GPU of the node, so the natural all-gather of a few KB .. ~1 MB
// per rank is "everybody writes its rows into everybody's mailbox at once" -- one hop, all seven links of a GPU busy together, no
// ring, no intermediate copies, and nothing at all to wait for on the host.  Layout:
//
//   every rank owns ONE mailbox in its own HBM (fine-grained: remote stores land without a kernel boundary), exported once through
//   hipIpcGetMemHandle and mapped by every peer (hipIpcOpenMemHandle);   mailbox = [2 tick parities][world sources] slots,
//   slot = 128-byte header {seq, count} + max_rows * row_bytes of payload.
//
//   post(t):    one launch on the sender.  Blocks (peer p, part j) copy the sender's n rows into slot [t & 1][rank] of p's mailbox
//               with 16-byte stores, fence to system scope, and the last block of a peer publishes {count = n; seq = t + 1} with a
//               system-scope release store.  A SILENT tick (n = 0: the gate did not fire -- the common case) moves only the
//               16-byte header per peer: no collective, no payload, no host involvement.
//   collect(t): one launch on the receiver.  Block r waits (bounded: a dead peer becomes an error code, never a hung GPU) until
//               slot [t & 1][r].seq == t + 1, then copies that source's `count` rows to the caller's buffer and the count to the
//               caller's count array and to a pinned host mirror.
//
// Two parities are enough: a rank's post(t + 2) is in stream order behind its own collect(t + 1), which saw the peer's post(t + 1),
// which is in that peer's stream order behind its collect(t) -- the slot being overwritten has been drained.  That holds for the
// blocking form (post(t); collect(t)) and for the pipelined one (collect(t - 1); post(t)) the streaming loop uses.
//
// Bootstrap is the caller's: sm_comm_init creates the mailbox, sm_comm_export gives its 64-byte handle, the caller moves the
// handles of all ranks (torch.distributed all_gather of 64 bytes, a file, MPI -- exactly as an ncclUniqueId travels) and hands them
// to sm_comm_connect.  Two processes on ONE GPU are a valid world (the -m gpu test runs that on the single-GPU box).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#include "common.h"
#include "host.h"

#define SM_COMM_HDR 128
#define SM_COMM_PARTS 4                       // blocks per peer in a post launch

struct CommHdr { unsigned long long seq; int count; int pad; };

struct sm_comm {
    int rank = 0, world = 1, max_rows = 0, row_bytes = 0, device = 0;
    size_t slot_bytes = 0, box_bytes = 0;
    char* box = nullptr;                                   // this rank's mailbox (fine-grained device memory)
    std::vector<char*> peer;                               // every rank's mailbox as mapped here (peer[rank] == box)
    std::vector<bool> opened;
    unsigned* part_done = nullptr;                         // [world] arrival counters of the post launch's blocks
    int* host_counts = nullptr;                            // pinned + mapped: [2][world] counts of the last collects, [2] error words
    int* host_counts_dev = nullptr;
    unsigned long long posted = 0, collected = 0;
    bool connected = false;
    bool timed_out = false;                                // sm_comm_poll_counts reported a late peer for the last collect: sm_comm_recollect may re-issue it
    long long timeout_ticks = 0;                           // wall_clock64 ticks (100 MHz) a collect waits for a peer
    ~sm_comm() {
        for (int r = 0; r < (int)peer.size(); ++r)
            if (r != rank && opened[r] && peer[r]) (void)hipIpcCloseMemHandle(peer[r]);
        if (box) (void)hipFree(box);
        if (part_done) (void)hipFree(part_done);
        if (host_counts) (void)hipHostFree(host_counts);
    }
};

struct CommPeers { char* p[SM_COMM_MAX_RANKS]; };

// grid (world, SM_COMM_PARTS): block (p, j) writes part j of this rank's rows into peer p's slot [par][rank]
__global__ __launch_bounds__(512) void comm_post_kernel(CommPeers peers, int rank, size_t slot_bytes, int world, int par, const char* __restrict__ rows,
                                                        int n_rows, int row_bytes, unsigned long long seq, unsigned* __restrict__ part_done) {
    const int p = blockIdx.x, j = blockIdx.y;
    char* slot = peers.p[p] + ((size_t)par * world + rank) * slot_bytes;
    const size_t total16 = ((size_t)n_rows * row_bytes) >> 4;
    u32x4* dst = (u32x4*)(slot + SM_COMM_HDR);
    const u32x4* src = (const u32x4*)rows;
    for (size_t i = (size_t)j * blockDim.x + threadIdx.x; i < total16; i += (size_t)SM_COMM_PARTS * blockDim.x) dst[i] = src[i];
    __threadfence_system();                                // this thread's payload stores are visible system-wide before the flag can be
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(&part_done[p], 1u);
        if (prev == SM_COMM_PARTS - 1) {                   // the last part of this peer: publish
            part_done[p] = 0;
            __threadfence_system();
            CommHdr* h = (CommHdr*)slot;
            __hip_atomic_store(&h->count, n_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&h->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// grid (world): block r drains source r's slot of parity `par`
__global__ __launch_bounds__(512) void comm_collect_kernel(char* box, size_t slot_bytes, int world, int par, unsigned long long seq, int max_rows, int row_bytes,
                                                           int* __restrict__ counts_out, char* __restrict__ payload_out, int* __restrict__ host_counts,
                                                           long long timeout_ticks) {
    const int r = blockIdx.x;
    char* slot = box + ((size_t)par * world + r) * slot_bytes;
    CommHdr* h = (CommHdr*)slot;
    __shared__ int s_count;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int cnt = -1;
        for (;;) {
            const unsigned long long s = __hip_atomic_load(&h->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (s == seq) { cnt = __hip_atomic_load(&h->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            if (wall_clock64() - t0 > timeout_ticks) break;
            __builtin_amdgcn_s_sleep(8);
        }
        if (cnt < 0 || cnt > max_rows) { host_counts[2 * world + par] = r + 1; cnt = 0; if (counts_out) counts_out[r] = -1; }   // error word: 1 + the rank that did not arrive
        else if (counts_out) counts_out[r] = cnt;
        host_counts[par * world + r] = cnt;
        s_count = cnt;
    }
    __syncthreads();
    const int cnt = s_count;
    if (!payload_out || cnt <= 0) return;
    __threadfence_system();
    const size_t total16 = ((size_t)cnt * row_bytes) >> 4;
    const u32x4* src = (const u32x4*)(slot + SM_COMM_HDR);
    u32x4* dst = (u32x4*)(payload_out + (size_t)r * max_rows * row_bytes);
    for (size_t i = threadIdx.x; i < total16; i += blockDim.x) dst[i] = __builtin_nontemporal_load(src + i);
}

extern "C" int sm_comm_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int sm_comm_init(int rank, int world, int max_rows, int row_bytes, sm_comm** out) {
    SM_REQUIRE(out && world >= 1 && world <= SM_COMM_MAX_RANKS && rank >= 0 && rank < world, "sm_comm_init: rank %d of world %d (at most %d ranks)", rank, world,
               SM_COMM_MAX_RANKS);
    SM_REQUIRE(max_rows >= 1 && row_bytes >= 16 && (row_bytes & 15) == 0, "sm_comm_init: max_rows >= 1, row_bytes a multiple of 16 (got %d, %d)", max_rows, row_bytes);
    sm_comm* c = new sm_comm();
    c->rank = rank; c->world = world; c->max_rows = max_rows; c->row_bytes = row_bytes;
    c->slot_bytes = SM_COMM_HDR + (((size_t)max_rows * row_bytes + 127) & ~(size_t)127);
    c->box_bytes = 2 * (size_t)world * c->slot_bytes;
    c->peer.assign(world, nullptr);
    c->opened.assign(world, false);
    auto fail = [&](hipError_t e, const char* what) { snprintf(g_sm_err, sizeof(g_sm_err), "sm_comm_init: %s: %s", what, hipGetErrorString(e)); delete c; return SM_EHIP; };
    hipError_t e;
    if ((e = hipGetDevice(&c->device)) != hipSuccess) return fail(e, "hipGetDevice");
    // fine-grained: a peer's stores and this GPU's polling loads meet in memory, not in a cache that only a kernel boundary cleans
    if ((e = hipExtMallocWithFlags((void**)&c->box, c->box_bytes, hipDeviceMallocFinegrained)) != hipSuccess) return fail(e, "hipExtMallocWithFlags(mailbox)");
    if ((e = hipMemset(c->box, 0, c->box_bytes)) != hipSuccess) return fail(e, "hipMemset(mailbox)");
    if ((e = hipMalloc((void**)&c->part_done, sizeof(unsigned) * world)) != hipSuccess) return fail(e, "hipMalloc");
    if ((e = hipMemset(c->part_done, 0, sizeof(unsigned) * world)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipHostMalloc((void**)&c->host_counts, sizeof(int) * (2 * world + 2), hipHostMallocMapped)) != hipSuccess) return fail(e, "hipHostMalloc");
    memset(c->host_counts, 0, sizeof(int) * (2 * world + 2));
    if ((e = hipHostGetDevicePointer((void**)&c->host_counts_dev, c->host_counts, 0)) != hipSuccess) return fail(e, "hipHostGetDevicePointer");
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(e, "hipDeviceSynchronize");
    c->peer[rank] = c->box;
    const char* t = getenv("SM_COMM_TIMEOUT_MS");
    c->timeout_ticks = (long long)(t ? atoi(t) : 5000) * 100000LL;      // wall_clock64 runs at 100 MHz
    if (world == 1) c->connected = true;
    *out = c;
    return SM_OK;
}

extern "C" int sm_comm_export(sm_comm* c, void* handle_out) {
    SM_REQUIRE(c && handle_out, "sm_comm_export: null arg");
    hipIpcMemHandle_t h;
    SM_HIP(hipIpcGetMemHandle(&h, c->box));
    memcpy(handle_out, &h, sizeof(h));
    return SM_OK;
}

extern "C" int sm_comm_connect(sm_comm* c, const void* all_handles) {
    SM_REQUIRE(c && all_handles, "sm_comm_connect: null arg");
    SM_REQUIRE(!c->connected || c->world == 1, "sm_comm_connect: already connected");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)all_handles + (size_t)r * sizeof(h), sizeof(h));
        void* p = nullptr;
        SM_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = (char*)p;
        c->opened[r] = true;
    }
    c->connected = true;
    return SM_OK;
}

extern "C" void sm_comm_destroy(sm_comm* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    delete c;
}

extern "C" int sm_comm_post(sm_comm* c, const void* rows, int n_rows, void* stream) {
    SM_REQUIRE(c && c->connected, "sm_comm_post: not connected (sm_comm_connect first)");
    SM_REQUIRE(n_rows >= 0 && n_rows <= c->max_rows && (n_rows == 0 || rows), "sm_comm_post: n_rows=%d outside [0, %d]", n_rows, c->max_rows);
    SM_REQUIRE(c->posted == c->collected, "sm_comm_post: tick %llu has not been collected yet (post and collect alternate)", c->posted - 1);
    SM_REQUIRE(!c->timed_out, "sm_comm_post: the collect of tick %llu timed out and was not re-issued (sm_comm_recollect)", c->collected - 1);
    SM_REQUIRE(n_rows == 0 || ((size_t)rows & 15) == 0, "sm_comm_post: rows must be 16-byte aligned");
    CommPeers peers;
    for (int r = 0; r < c->world; ++r) peers.p[r] = c->peer[r];
    const unsigned long long t = c->posted;
    comm_post_kernel<<<dim3(c->world, SM_COMM_PARTS), 512, 0, (hipStream_t)stream>>>(peers, c->rank, c->slot_bytes, c->world, (int)(t & 1), (const char*)rows, n_rows,
                                                                                     c->row_bytes, t + 1, c->part_done);
    SM_LAUNCH_CHECK();
    c->posted = t + 1;
    return SM_OK;
}

extern "C" int sm_comm_collect(sm_comm* c, int32_t* counts_out, void* payload_out, void* stream) {
    SM_REQUIRE(c && c->connected, "sm_comm_collect: not connected");
    SM_REQUIRE(c->collected + 1 == c->posted, "sm_comm_collect: nothing posted for tick %llu", c->collected);
    const unsigned long long t = c->collected;
    comm_collect_kernel<<<c->world, 512, 0, (hipStream_t)stream>>>(c->box, c->slot_bytes, c->world, (int)(t & 1), t + 1, c->max_rows, c->row_bytes, counts_out,
                                                                   (char*)payload_out, c->host_counts_dev, c->timeout_ticks);
    SM_LAUNCH_CHECK();
    c->collected = t + 1;
    return SM_OK;
}

// counts of the collect before last / last (parity = tick & 1) as the pinned host mirror holds them: valid once the stream that ran
// that collect has reached it (event / synchronize); returns SM_EHIP-class error when a peer did not arrive in time
extern "C" int sm_comm_host_counts(sm_comm* c, int tick_parity, int32_t* counts_out) {
    SM_REQUIRE(c && counts_out && (tick_parity == 0 || tick_parity == 1), "sm_comm_host_counts: bad args");
    const int err = c->host_counts[2 * c->world + tick_parity];
    if (err) {
        c->host_counts[2 * c->world + tick_parity] = 0;
        SM_FAIL(SM_EHIP, "sm_comm: rank %d did not post in time (rank %d waited %lld ms)", err - 1, c->rank, c->timeout_ticks / 100000LL);
    }
    for (int r = 0; r < c->world; ++r) counts_out[r] = c->host_counts[tick_parity * c->world + r];
    return SM_OK;
}

// The same read for callers that treat a timeout as "not yet" (ranks fire on different ticks by design, and a rank that is decoding a
// long reply lags its peers by more than any sensible GPU-side spin): returns 0 with the counts, or 1 with *missing_rank = the rank that
// had not posted when the collect gave up -- no error is recorded, and the collect of that tick may be issued AGAIN with
// sm_comm_recollect (the late payload is still delivered: a peer cannot overwrite the slot before this rank has posted its next tick).
extern "C" int sm_comm_poll_counts(sm_comm* c, int tick_parity, int32_t* counts_out, int* missing_rank) {
    SM_REQUIRE(c && counts_out && (tick_parity == 0 || tick_parity == 1), "sm_comm_poll_counts: bad args");
    const int err = c->host_counts[2 * c->world + tick_parity];
    if (err) {
        c->host_counts[2 * c->world + tick_parity] = 0;
        if (missing_rank) *missing_rank = err - 1;
        c->timed_out = true;
        return 1;
    }
    for (int r = 0; r < c->world; ++r) counts_out[r] = c->host_counts[tick_parity * c->world + r];
    return SM_OK;
}

// Re-issue the collect of the LAST collected tick after sm_comm_poll_counts reported a timeout for it (same outputs, same stream rules).
extern "C" int sm_comm_recollect(sm_comm* c, int32_t* counts_out, void* payload_out, void* stream) {
    SM_REQUIRE(c && c->connected, "sm_comm_recollect: not connected");
    SM_REQUIRE(c->timed_out && c->collected >= 1 && c->collected == c->posted, "sm_comm_recollect: the last collect did not time out (poll it with sm_comm_poll_counts first)");
    c->timed_out = false;
    const unsigned long long t = c->collected - 1;
    comm_collect_kernel<<<c->world, 512, 0, (hipStream_t)stream>>>(c->box, c->slot_bytes, c->world, (int)(t & 1), t + 1, c->max_rows, c->row_bytes, counts_out,
                                                                   (char*)payload_out, c->host_counts_dev, c->timeout_ticks);
    SM_LAUNCH_CHECK();
    return SM_OK;
}

// blocking-form convenience: post + collect of one tick on `stream` (the reference's allgather_diff_shape in one call)
extern "C" int sm_allgather_gated(sm_comm* c, const void* rows, int n_rows, int32_t* counts_out, void* payload_out, void* stream) {
    int rc = sm_comm_post(c, rows, n_rows, stream);
    if (rc) return rc;
    return sm_comm_collect(c, counts_out, payload_out, stream);
}

extern "C" int sm_comm_max_rows(sm_comm* c) { return c ? c->max_rows : -1; }
extern "C" int sm_comm_tick(sm_comm* c) { return c ? (int)c->collected : -1; }

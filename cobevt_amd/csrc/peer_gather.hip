// One-shot agent-feature exchange over xGMI by direct peer writes: the V2V feature-sharing step in
// front of FuseBEVT.  The reference runs every agent in one process (opv2v/opencood/models/corpbevt.py:112-124 keeps the
// agents as a batch dimension up to `regroup`, sub_modules/fuse_utils.py:8-61), and its only collective call sites are the
// DDP set-up of opv2v/opencood/tools/multi_gpu_utils.py:32-37 / train_camera.py:105-110, so this exchange is new code.
//
// A (32, 32, 128) bf16 agent block is 256 KiB: a ring all-gather pays world-1 latency-bound steps for it.  Here every rank
// owns a WINDOW (uncached device memory, exported through hipIpc and mapped by every peer) that holds the frame's blocks
// in agent order plus a few flag words; one launch pair per exchange:
//   peer_ack_kernel   "I have finished reading the previous exchange out of my window" -> ack[rank] = epoch + 1 in every
//                     peer's window (system-scope release), then waits for every peer's ack in its own window;
//   peer_push_kernel  16-byte stores of the local blocks straight into the destination windows over xGMI (own window
//                     included), system fence, and the last workgroup publishes ready[rank] = epoch + 1 everywhere and
//                     waits until every peer's ready flag has arrived in its own window.
// When the push kernel retires, the window holds the whole frame and the kernels after it in stream order may read it.
// The epoch lives in the window (device memory), so the pair can sit inside a captured HIP graph and be replayed.  All
// waits are bounded (`spin_limit` polls): a missing peer sets status != 0 instead of hanging the GPU.
#include "common.hpp"
#include <cstring>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxLocalBlocks = 16;

struct PeerWindows {
    char* win[kMaxRanks];
};

struct PushPlan {                   // per local block: destination rank (-1 = every rank) and block slot in the window
    int dest_rank[kMaxLocalBlocks];
    int dest_block[kMaxLocalBlocks];
};

// flag words at the end of the data region (uint32 each)
//   [0 .. 8)   ack[r]    written by rank r
//   [8 .. 16)  ready[r]  written by rank r
//   16 epoch (completed exchanges), 17 workgroup counter, 18 status (0 ok, 1 ack timeout, 2 ready timeout)
constexpr int kAck = 0, kReady = 8, kEpoch = 16, kCounter = 17, kStatus = 18, kFlagWords = 32;

__device__ __forceinline__ uint32_t load_sys(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void store_sys(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ bool wait_at_least(const uint32_t* p, uint32_t want, long spin_limit) {
    for (long i = 0; i < spin_limit; ++i) {
        if ((int32_t)(load_sys(p) - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

__global__ __launch_bounds__(64) void peer_ack_kernel(PeerWindows wins, int world, int rank, long flags_off, long spin_limit) {
    uint32_t* mine = reinterpret_cast<uint32_t*>(wins.win[rank] + flags_off);
    const uint32_t next = load_sys(mine + kEpoch) + 1u;
    const int t = threadIdx.x;
    if (t < world) {
        uint32_t* theirs = reinterpret_cast<uint32_t*>(wins.win[t] + flags_off);
        store_sys(theirs + kAck + rank, next);
        if (!wait_at_least(mine + kAck + t, next, spin_limit)) store_sys(mine + kStatus, 1u);
    }
}

__global__ __launch_bounds__(256) void peer_push_kernel(const uint4* __restrict__ local, PeerWindows wins, PushPlan plan,
                                                        int world, int rank, int n_local, long block_units, long flags_off,
                                                        long spin_limit) {
    uint32_t* mine = reinterpret_cast<uint32_t*>(wins.win[rank] + flags_off);
    const uint32_t next = load_sys(mine + kEpoch) + 1u;
    const long total = (long)n_local * block_units;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int j = (int)(i / block_units);
        const long u = i - (long)j * block_units;
        const uint4 v = local[i];
        const long dst = (long)plan.dest_block[j] * block_units + u;
        const int dr = plan.dest_rank[j];
        if (dr >= 0) {
            reinterpret_cast<uint4*>(wins.win[dr])[dst] = v;
        } else {
            for (int p = 0; p < world; ++p) reinterpret_cast<uint4*>(wins.win[(rank + p) % world])[dst] = v;
        }
    }
    __threadfence_system();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) {
        const uint32_t done = __hip_atomic_fetch_add(mine + kCounter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (done == gridDim.x - 1);
    }
    __syncthreads();
    if (!last) return;
    __threadfence_system();
    const int t = threadIdx.x;
    if (t < world) {
        uint32_t* theirs = reinterpret_cast<uint32_t*>(wins.win[t] + flags_off);
        store_sys(theirs + kReady + rank, next);
        if (!wait_at_least(mine + kReady + t, next, spin_limit)) store_sys(mine + kStatus, 2u);
    }
    __syncthreads();
    if (t == 0) {
        __hip_atomic_store(mine + kCounter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        store_sys(mine + kEpoch, next);
    }
}

}  // namespace

// Window allocation: uncached device memory (remote stores and the flag polls must not sit in a non-coherent L2 line),
// zero-filled, plus its 64-byte hipIpc handle.  `bytes` = data bytes (a multiple of 16); the flag words follow the data.
extern "C" int cobevt_peer_window_alloc(long bytes, void** dptr, void* handle64) {
    if (!dptr || !handle64 || bytes < 16 || bytes % 16) return COBEVT_ERR_ARG;
    void* p = nullptr;
    const size_t total = (size_t)bytes + kFlagWords * sizeof(uint32_t);
    if (hipExtMallocWithFlags(&p, total, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p, total, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return COBEVT_ERR_LAUNCH; }
    }
    if (hipMemset(p, 0, total) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return COBEVT_ERR_LAUNCH; }
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return COBEVT_ERR_LAUNCH; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpc handle size");
    std::memcpy(handle64, &h, 64);
    *dptr = p;
    return COBEVT_OK;
}

extern "C" int cobevt_peer_window_open(const void* handle64, void** dptr) {
    if (!handle64 || !dptr) return COBEVT_ERR_ARG;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, 64);
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return COBEVT_ERR_LAUNCH; }
    *dptr = p;
    return COBEVT_OK;
}

extern "C" int cobevt_peer_window_close(void* dptr) {
    if (!dptr) return COBEVT_ERR_ARG;
    return hipIpcCloseMemHandle(dptr) == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_peer_window_free(void* dptr) {
    if (!dptr) return COBEVT_ERR_ARG;
    return hipFree(dptr) == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// status word of the local window (0 = every wait so far completed); synchronises with `stream` first.
extern "C" int cobevt_peer_window_status(const void* window, long bytes, int* status, int* epoch, hipStream_t stream) {
    if (!window || !status || !epoch) return COBEVT_ERR_ARG;
    uint32_t words[kFlagWords];
    if (hipStreamSynchronize(stream) != hipSuccess) return COBEVT_ERR_LAUNCH;
    if (hipMemcpy(words, (const char*)window + bytes, sizeof(words), hipMemcpyDeviceToHost) != hipSuccess) return COBEVT_ERR_LAUNCH;
    *status = (int)words[kStatus];
    *epoch = (int)words[kEpoch];
    return COBEVT_OK;
}

// The same read without a synchronisation: the window's flag words are copied to `host_words` (pinned host memory, kFlagWords = 32
// 32-bit words; status at [18], completed exchanges at [16]) in stream order - the caller records an event behind it and looks at the
// words once the event has completed (FrameShardedCorpBEVT checks every step this way instead of draining the pipeline).
extern "C" int cobevt_peer_window_status_async(const void* window, long bytes, unsigned int* host_words, hipStream_t stream) {
    if (!window || !host_words) return COBEVT_ERR_ARG;
    if (hipMemcpyAsync(host_words, (const char*)window + bytes, kFlagWords * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess)
        return COBEVT_ERR_LAUNCH;
    return COBEVT_OK;
}

// One exchange.  `windows`: host array of `world` device pointers (this process's mappings of every rank's window, own
// window at [rank]); local: n_local contiguous blocks of block_bytes; dest_rank[j] (-1 = all ranks) / dest_block[j]: where
// local block j goes; window_bytes: the data size passed to cobevt_peer_window_alloc.
extern "C" int cobevt_peer_exchange(const void* local, void* const* windows, int world, int rank, int n_local,
                                    long block_bytes, const int* dest_rank, const int* dest_block, long window_bytes,
                                    long spin_limit, hipStream_t stream) {
    if (!windows || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return COBEVT_ERR_ARG;
    if (n_local < 0 || n_local > kMaxLocalBlocks || block_bytes < 16 || block_bytes % 16) return COBEVT_ERR_SHAPE;
    if (n_local && (!local || !dest_rank || !dest_block)) return COBEVT_ERR_ARG;
    PeerWindows wins{};
    for (int p = 0; p < world; ++p) {
        if (!windows[p]) return COBEVT_ERR_ARG;
        wins.win[p] = (char*)windows[p];
    }
    PushPlan plan{};
    for (int j = 0; j < n_local; ++j) {
        if (dest_rank[j] >= world || dest_block[j] < 0 || (long)(dest_block[j] + 1) * block_bytes > window_bytes) return COBEVT_ERR_SHAPE;
        plan.dest_rank[j] = dest_rank[j];
        plan.dest_block[j] = dest_block[j];
    }
    if (spin_limit < 1) spin_limit = 30000000;      // ~30 s of polls: ranks may be seconds apart while one of them builds plans / captures
    const long units = block_bytes / 16;
    hipLaunchKernelGGL(peer_ack_kernel, dim3(1), dim3(64), 0, stream, wins, world, rank, window_bytes, spin_limit);
    const long total = (long)n_local * units;
    int grid = (int)((total + 255) / 256);
    if (grid > 64) grid = 64;          // 64 workgroups x 256 lanes x 16 B in flight per pass: enough to cover the xGMI latency
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(peer_push_kernel, dim3(grid), dim3(256), 0, stream, (const uint4*)local, wins, plan, world, rank,
                       n_local, units, window_bytes, spin_limit);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// fmhr_amd/csrc/peer.hip — one-shot peer-to-peer all-reduce of the view-sharded step's bucket over the direct xGMI links
// ("each GPU reads the 7 peers' shards over its 7 direct links, then all-gathers"), for gfx950.
//
// The step's one exchange is 12 V + 8 floats (2.35 MB at V = 49 k): latency-bound.  A ring all-reduce of G ranks is 2 (G - 1) dependent
// hops; on MI355X every GPU has a direct link to every other, so the reduction can be ONE launch per rank with two flag exchanges:
//   A  copy the bucket into this rank's exported staging block            -> flag "staged"  written into every peer's flag array
//   B  wait for every peer's "staged"
//   C  reduce MY shard: read the shard from every rank's staging block (7 remote reads per element, all links in parallel, fixed rank
//      order: the result is the same bits on every rank) into my exported output block -> flag "reduced" to every peer
//   D  wait for every peer's "reduced"
//   E  gather: read every peer's reduced shard into the bucket
// The exported block of a rank ([staging | reduced shard | flags]) is device memory allocated UNCACHED (hipDeviceMallocUncached: no
// L2 retention, so what a peer wrote is what a load sees inside a running kernel) and shared by hipIpcGetMemHandle / OpenMemHandle;
// the flags are written with system-scope release stores and read with system-scope acquire loads, every thread fences (acquire, system
// scope: the L1 is invalidated) behind the wait before it touches a peer's data, which then moves in plain 16-byte accesses (the first
// form read it with 8-byte system-scope atomic loads: 33 us for the 2.35 MB bucket with NO peer at all; this one: see DESIGN.md §6).  Flags carry the call's epoch (monotone:
// nothing is ever reset); every spin is BOUNDED BY WALL CLOCK — a wave that waits longer than 5 s (FMHR_PEER_WAIT_MS) sets the block's abort
// word, every other wait sees it and the whole grid drains with the bucket's loss slots set to NaN: a rank whose peer died or fell more than
// that behind leaves the kernel with an error the host reads (fmhr_peer_status; fmhr_amd.dist.ExchangeMonitor polls it) and with results
// nobody can mistake for reduced ones; it does not hang the GPU.  Host protocol, negotiation and fallback to RCCL: fmhr_amd/dist.py (PeerAllReduce).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace fmhr {

constexpr int kPeerMax = 16;
constexpr int kPeerBlocks = 128;                  // resident by construction (256 CUs): the grid barrier below needs every block running; the
                                                  // copies are as fast from 128 workgroups as from 1 024 (3 us), a barrier costs per workgroup
constexpr long long kWaitTicksDefault = 5ll * 100000000ll;   // a wait gives up after 5 s of WALL CLOCK (s_memrealtime: 100 MHz, constant) — the rank
                                                  // skew the exchange tolerates; FMHR_PEER_WAIT_MS at fmhr_peer_create overrides it.  (Until round 5
                                                  // the bound was a poll count, "a few seconds" only for an undisturbed wave: ADVICE r5)
constexpr unsigned kClockEvery = 64;              // polls between two looks at the clock

struct PeerBlock {                                // what the kernel needs of one rank's exported block
    float4* stage;                                // [n / 4] the rank's bucket as 16-byte words
    float4* out;                                  // [shard_max / 4] its reduced shard
    unsigned* flags;                              // [2 * kPeerMax] staged / reduced epochs written BY the peers + [2] local words
};
struct PeerArgs {
    const PeerBlock* blk;                         // [world] in device memory (indexed with a run-time rank: as a kernel argument the array went
                                                  // through a 16-KB LDS copy and the loads below through scratch)
    int rank, world;
    long long n2;                                 // 16-byte words in the bucket
    unsigned epoch;
    unsigned bar0;                                // the grid-barrier counter before this call's first barrier (the host keeps the count)
    int in_place;                                 // the bucket IS this rank's staging area (fmhr_peer_staging): nothing to stage
    unsigned* bar;                                // local: grid-barrier counter (monotone), abort word
    long long wait_ticks;                         // s_memrealtime ticks (10 ns) a wait may last
};

// true once a poll loop that started at `t0` has used up its wall-clock budget (looked at every kClockEvery-th poll)
__device__ __forceinline__ bool out_of_time(unsigned& spins, long long t0, long long budget)
{
    if ((++spins % kClockEvery) != 0u) return false;
    return (long long)wall_clock64() - t0 > budget;
}

// every thread, behind a wait on the peers' flags and the workgroup barrier that ends it: nothing of a peer's block may come out of
// this CU's L1 (the waiting lanes' acquire loads invalidate it as well; this makes it every wave's own)
__device__ __forceinline__ void acquire_system() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }

// FENCES.  A system-scope fence is a cache maintenance instruction per WAVE (release: L2 write-back, acquire: L1 / L2 invalidate); issued
// by all 512 waves of the grid around a barrier it is most of the barrier (tools/micro/peer_phases.hip: a grid barrier of 128 workgroups
// 3.1 us bare, 9.9 us with a system-scope release + acquire in every wave; 256 workgroups 4.9 / 19.5 us).  What the protocol needs:
//   * a wave's stores into the exported block (UNCACHED memory: they do not stay in the L2) are COMPLETE before the workgroup's arrival —
//     a workgroup-scope release in every wave (a wait for its outstanding stores) and the workgroup barrier;
//   * ONE system-scope release per workgroup, by the thread that then increments the counter / writes the peers' flags;
//   * ONE system-scope acquire per workgroup behind the wait, by the threads that polled (the L1 it invalidates is the CU's, shared by
//     the workgroup's waves), then the workgroup barrier.
// all blocks of the grid have arrived `target / gridDim.x` times; false: aborted.  The polls are RELAXED loads
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned target, long long budget)
{
    __shared__ int ok;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // this wave's stores (staging / reduced shard) have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                // system scope, once per workgroup
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        const long long t0 = (long long)wall_clock64();
        int good = 1;
        while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (out_of_time(spins, t0, budget) || __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { good = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!good) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the other workgroups' arrivals were relaxed increments behind their release fences: this acquire orders what they wrote before the
        // flag store that follows the barrier (signal_peers: a system-scope release) — ADVICE r5
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// every peer's flag word [which * kPeerMax + p] of MY block has reached `epoch`; false: gave up (abort word set)
__device__ __forceinline__ bool wait_peers(const PeerArgs& A, int which)
{
    __shared__ int ok;
    if (threadIdx.x < 64) {                                          // the workgroup's first wave polls, one lane per peer
        if (threadIdx.x < (unsigned)A.world && (int)threadIdx.x != A.rank) {
            const unsigned* f = A.blk[A.rank].flags + which * kPeerMax + threadIdx.x;
            unsigned spins = 0;
            const long long t0 = (long long)wall_clock64();
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - A.epoch) < 0) {
                if (out_of_time(spins, t0, A.wait_ticks) || __hip_atomic_load(A.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(A.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        acquire_system();                                            // once per workgroup: nothing of a peer's block out of this CU's caches
        if (threadIdx.x == 0) ok = __hip_atomic_load(A.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    }
    __syncthreads();
    return ok != 0;
}

__device__ __forceinline__ void signal_peers(const PeerArgs& A, int which)
{
    // (called by block 0 behind a grid barrier: everything this rank wrote is fenced to system scope)
    if (threadIdx.x < (unsigned)A.world && (int)threadIdx.x != A.rank)
        __hip_atomic_store(A.blk[threadIdx.x].flags + which * kPeerMax + A.rank, A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void shard_of(long long n2, int world, int r, long long& a, long long& b)
{
    const long long base = n2 / world, rem = n2 % world;
    a = r * base + (r < rem ? r : rem);
    b = a + base + (r < rem ? 1 : 0);
}

// dst[i] = src[i] for i in [0, n), grid-strided, four words per thread and round with every load issued before the first store
__device__ __forceinline__ void copy_words(float4* __restrict__ dst, const float4* __restrict__ src, long long n, long long tid, long long nth)
{
    long long i = tid;
    for (; i + 3 * nth < n; i += 4 * nth) {
        const float4 a = src[i], b = src[i + nth], c = src[i + 2 * nth], d = src[i + 3 * nth];
        dst[i] = a; dst[i + nth] = b; dst[i + 2 * nth] = c; dst[i + 3 * nth] = d;
    }
    for (; i < n; i += nth) dst[i] = src[i];
}

// A wait gave up (the abort word is set and stays set: every later call leaves at its first wait): the bucket holds this rank's share
// only.  So that NOTHING downstream can take it for the reduced one, its last 8 floats — the step's loss slots, the valid-pixel count among
// them — become NaN: fmhr_ham_combine then turns every gradient, the losses and (fused Adam) the parameters into NaN on this rank.
__device__ __forceinline__ void poison(float* bucket, long long n2)
{
    if (threadIdx.x < 8) bucket[4 * n2 - 8 + threadIdx.x] = __builtin_nanf("");
}

__global__ void __launch_bounds__(kBlock) k_peer_all_reduce(const PeerArgs A, float* __restrict__ bucket)
{
    const long long tid = (long long)blockIdx.x * kBlock + threadIdx.x, nth = (long long)gridDim.x * kBlock;
    float4* const b4 = reinterpret_cast<float4*>(bucket);
    unsigned bar_at = A.bar0;
    // A: stage (a thread's loads are all in flight before its first store: the loops below are latency-, not bandwidth-bound).
    // In place — the step's last kernels wrote the bucket straight into the exported block — there is nothing to copy and nothing of
    // THIS launch to wait for: the kernels in front of it on the stream have completed, and the flag stores are system-scope releases
    const PeerBlock mine = A.blk[A.rank];
    if (!A.in_place) {
        copy_words(mine.stage, b4, A.n2, tid, nth);
        bar_at += gridDim.x;
        if (!grid_barrier(A.bar, bar_at, A.wait_ticks)) { poison(bucket, A.n2); return; }
    }
    if (blockIdx.x == 0) signal_peers(A, 0);
    // B
    if (!wait_peers(A, 0)) { poison(bucket, A.n2); return; }
    // C: my shard, every rank's contribution in rank order (the same sum, bit for bit, whoever computes it)
    long long s0, s1;
    shard_of(A.n2, A.world, A.rank, s0, s1);
    for (long long i = s0 + tid; i < s1; i += nth) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p0 = 0; p0 < A.world; p0 += 4) {                   // four ranks' words in flight together (the links work in parallel)
            const int last = A.world - 1;
            const float4 w0 = A.blk[p0].stage[i], w1 = A.blk[min(p0 + 1, last)].stage[i], w2 = A.blk[min(p0 + 2, last)].stage[i],
                         w3 = A.blk[min(p0 + 3, last)].stage[i];
            acc.x += w0.x; acc.y += w0.y; acc.z += w0.z; acc.w += w0.w;
            if (p0 + 1 <= last) { acc.x += w1.x; acc.y += w1.y; acc.z += w1.z; acc.w += w1.w; }
            if (p0 + 2 <= last) { acc.x += w2.x; acc.y += w2.y; acc.z += w2.z; acc.w += w2.w; }
            if (p0 + 3 <= last) { acc.x += w3.x; acc.y += w3.y; acc.z += w3.z; acc.w += w3.w; }
        }
        mine.out[i - s0] = acc;
    }
    bar_at += gridDim.x;
    if (!grid_barrier(A.bar, bar_at, A.wait_ticks)) { poison(bucket, A.n2); return; }
    if (blockIdx.x == 0) signal_peers(A, 1);
    // D
    if (!wait_peers(A, 1)) { poison(bucket, A.n2); return; }
    // E: gather every rank's reduced shard
    for (int p = 0; p < A.world; p++) {
        long long a, b;
        shard_of(A.n2, A.world, p, a, b);
        copy_words(b4 + a, A.blk[p].out, b - a, tid, nth);
    }
}

struct PeerState {
    int rank, world;
    long long n2, shard_max2;
    size_t bytes, off_out, off_flags;
    char* base;                                   // this rank's exported block
    char* peer[kPeerMax];                         // mapped blocks (peer[rank] == base)
    bool opened[kPeerMax];
    unsigned* bar;                                // local counters (ordinary device memory)
    PeerBlock* table;                             // [world] device copy of the mapped blocks' pointers, refreshed when a peer is imported
    bool table_dirty;
    unsigned epoch;
    unsigned bar_count;                           // arrivals at the grid-barrier counter the launches so far account for
    long long wait_ticks;
};

}  // namespace fmhr

using namespace fmhr;

extern "C" size_t fmhr_peer_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

extern "C" int fmhr_peer_create(int rank, int world, long long n_floats, void** state_out)
{
    FMHR_CHECK_ARG(state_out != nullptr && world >= 1 && world <= kPeerMax && rank >= 0 && rank < world, "peer_create: bad rank / world (%d of %d, at most %d ranks)",
                   rank, world, kPeerMax);
    FMHR_CHECK_ARG(n_floats >= 4 && n_floats % 4 == 0, "peer_create: the bucket must hold a multiple of 4 floats (16-byte words), got %lld", n_floats);
    PeerState* S = new PeerState();
    memset(S, 0, sizeof(*S));
    S->rank = rank; S->world = world; S->n2 = n_floats / 4;
    S->shard_max2 = (S->n2 + world - 1) / world;
    S->wait_ticks = kWaitTicksDefault;
    if (const char* ms = getenv("FMHR_PEER_WAIT_MS")) { const long long v = atoll(ms); if (v > 0) S->wait_ticks = v * 100000ll; }
    S->off_out = ((size_t)S->n2 * 16 + 255) & ~(size_t)255;
    S->off_flags = (S->off_out + (size_t)S->shard_max2 * 16 + 255) & ~(size_t)255;
    S->bytes = S->off_flags + 4096;
    hipError_t e = hipExtMallocWithFlags((void**)&S->base, S->bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { delete S; return hip_fail(e, "peer_create: hipExtMallocWithFlags(uncached)"); }
    e = hipMemset(S->base, 0, S->bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&S->bar, 256);
    if (e == hipSuccess) e = hipMalloc((void**)&S->table, sizeof(PeerBlock) * kPeerMax);
    if (e == hipSuccess) e = hipMemset(S->bar, 0, 256);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(S->base); if (S->bar) (void)hipFree(S->bar); if (S->table) (void)hipFree(S->table); delete S; return hip_fail(e, "peer_create: initialisation"); }
    S->peer[rank] = S->base;
    S->table_dirty = true;
    *state_out = S;
    return FMHR_OK;
}

extern "C" int fmhr_peer_export(void* state, void* handle_out)
{
    PeerState* S = (PeerState*)state;
    FMHR_CHECK_ARG(S != nullptr && handle_out != nullptr, "peer_export: null");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, S->base);
    if (e != hipSuccess) return hip_fail(e, "peer_export: hipIpcGetMemHandle");
    memcpy(handle_out, &h, sizeof(h));
    return FMHR_OK;
}

extern "C" int fmhr_peer_import(void* state, int peer, const void* handle)
{
    PeerState* S = (PeerState*)state;
    FMHR_CHECK_ARG(S != nullptr && handle != nullptr && peer >= 0 && peer < S->world && peer != S->rank, "peer_import: bad peer %d", peer);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail(e, "peer_import: hipIpcOpenMemHandle");
    S->peer[peer] = (char*)p;
    S->opened[peer] = true;
    S->table_dirty = true;
    return FMHR_OK;
}

extern "C" int fmhr_peer_all_reduce(void* state, float* bucket, long long n_floats, void* stream)
{
    PeerState* S = (PeerState*)state;
    FMHR_CHECK_ARG(S != nullptr && bucket != nullptr && n_floats == 4 * S->n2, "peer_all_reduce: the bucket must be the %lld floats the state was created for",
                   S ? 4 * S->n2 : 0ll);
    FMHR_CHECK_ARG(((uintptr_t)bucket & 15) == 0, "peer_all_reduce: the bucket must be 16-byte aligned");
    PeerArgs A;
    memset(&A, 0, sizeof(A));
    if (S->table_dirty) {                                            // (first call after the imports: one small synchronous copy)
        PeerBlock host[kPeerMax];
        memset(host, 0, sizeof(host));
        for (int p = 0; p < S->world; p++) {
            FMHR_CHECK_ARG(S->peer[p] != nullptr, "peer_all_reduce: rank %d's block has not been imported", p);
            host[p].stage = (float4*)S->peer[p];
            host[p].out = (float4*)(S->peer[p] + S->off_out);
            host[p].flags = (unsigned*)(S->peer[p] + S->off_flags);
        }
        hipError_t e = hipMemcpy(S->table, host, sizeof(host), hipMemcpyHostToDevice);
        if (e != hipSuccess) return hip_fail(e, "peer_all_reduce: pointer table");
        S->table_dirty = false;
    }
    A.blk = S->table;
    A.rank = S->rank; A.world = S->world; A.n2 = S->n2; A.bar = S->bar;
    A.epoch = ++S->epoch;
    A.in_place = (char*)bucket == S->base ? 1 : 0;
    A.bar0 = S->bar_count;
    A.wait_ticks = S->wait_ticks;
    S->bar_count += (A.in_place ? 1u : 2u) * (unsigned)kPeerBlocks;
    k_peer_all_reduce<<<kPeerBlocks, kBlock, 0, (hipStream_t)stream>>>(A, bucket);
    FMHR_CHECK_LAUNCH("k_peer_all_reduce");
    return FMHR_OK;
}

extern "C" int fmhr_peer_staging(void* state, void** ptr_out, long long* n_floats_out)
{
    PeerState* S = (PeerState*)state;
    FMHR_CHECK_ARG(S != nullptr && ptr_out != nullptr && n_floats_out != nullptr, "peer_staging: null");
    *ptr_out = S->base;
    *n_floats_out = 4 * S->n2;
    return FMHR_OK;
}

extern "C" int fmhr_peer_status(void* state, int* gave_up)
{
    PeerState* S = (PeerState*)state;
    FMHR_CHECK_ARG(S != nullptr && gave_up != nullptr, "peer_status: null");
    unsigned w[2] = {0u, 0u};
    hipError_t e = hipMemcpy(w, S->bar, 8, hipMemcpyDeviceToHost);          // (synchronises: call it where the caller synchronises anyway)
    if (e != hipSuccess) return hip_fail(e, "peer_status");
    *gave_up = w[1] != 0u;
    return FMHR_OK;
}

extern "C" int fmhr_peer_destroy(void* state)
{
    PeerState* S = (PeerState*)state;
    if (S == nullptr) return FMHR_OK;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < S->world; p++)
        if (S->opened[p]) (void)hipIpcCloseMemHandle(S->peer[p]);
    (void)hipFree(S->base);
    (void)hipFree(S->bar);
    (void)hipFree(S->table);
    delete S;
    return FMHR_OK;
}

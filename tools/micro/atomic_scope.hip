// micro-benchmark: posted u64 atomicMin into a depth-key buffer whose slices are each written by ONE XCD only (work is handed out
// by the XCC id the workgroup actually runs on), at agent scope vs workgroup scope.  Question: does the narrower scope execute in
// the XCD's L2 (faster, fewer fabric requests) and does it still give the same result?  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

constexpr int K = 16;
// PART: 1 = slice chosen by the XCC id the block runs on (cursor per XCD), 0 = slice = item % 8 whatever the XCD (one cursor)
template <int SCOPE, int PART>
__global__ void __launch_bounds__(256) k_min(unsigned long long* keys, long long per_slice, long long window, int* cursor, int items_per_slice,
                                             int* xcd_hist)
{
    __shared__ int item_s;
    const unsigned x = xcc_id() & 7u;
    if (threadIdx.x == 0) atomicAdd(&xcd_hist[x], 1);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (;;) {
        if (threadIdx.x == 0) item_s = PART ? atomicAdd(&cursor[x * 16], 1) : atomicAdd(&cursor[0], 1);
        __syncthreads();
        const int it = item_s;
        __syncthreads();
        int slice, idx;
        if (PART) { slice = (int)x; idx = it; if (it >= items_per_slice) break; }
        else { slice = it & 7; idx = it >> 3; if (idx >= items_per_slice) break; }
        unsigned long long* sp = keys + (size_t)slice * per_slice;
        // the window slides through the slice with the item index (a view at a time)
        const long long nwin = per_slice / window, wbase = ((long long)idx * nwin / items_per_slice) * window;
        for (int k = 0; k < K; k++) {
            const unsigned h = hash32(((unsigned)(idx * 8 + slice) * 4u + wave) * 64u + k);
            const long long centre = wbase + h % (unsigned)(window - 8 * 512);
            const long long a = centre + (lane >> 3) * 512 + (lane & 7) + ((hash32(h + lane) >> 8) & 3);      // ~8-10 lines per instruction
            const unsigned long long val = ((unsigned long long)hash32(h * 64u + lane) << 32) | (unsigned)lane;
            __hip_atomic_fetch_min(&sp[a], val, __ATOMIC_RELAXED, SCOPE);
        }
    }
}

template <class F> float timeit(F f, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1000.f;
}

int main()
{
    const long long N = 48ll * 512 * 512, per_slice = N / 8;
    unsigned long long *ka, *kw, *kn; int *cursor, *hist;
    hipMalloc(&ka, N * 8); hipMalloc(&kw, N * 8); hipMalloc(&kn, N * 8); hipMalloc(&cursor, 8 * 16 * 4); hipMalloc(&hist, 8 * 4);
    const long long ops = 6000000;
    const int items_per_slice = (int)(ops / (256 * K) / 8);
    std::vector<unsigned long long> ha(N), hw(N), hn(N);
    for (long long window : {per_slice, 262144ll}) {
        auto run = [&](auto kern, unsigned long long* keys, int grid) {
            hipMemsetAsync(cursor, 0, 8 * 16 * 4, 0);
            kern<<<grid, 256, 0, 0>>>(keys, per_slice, window, cursor, items_per_slice, hist);
        };
        const int grid = 2048;
        hipMemset(ka, 0xFF, N * 8); hipMemset(kw, 0xFF, N * 8); hipMemset(kn, 0xFF, N * 8); hipMemset(hist, 0, 32);
        float t_agent = timeit([&] { run(k_min<__HIP_MEMORY_SCOPE_AGENT, 1>, ka, grid); });
        float t_wg = timeit([&] { run(k_min<__HIP_MEMORY_SCOPE_WORKGROUP, 1>, kw, grid); });
        float t_wave = timeit([&] { run(k_min<__HIP_MEMORY_SCOPE_WAVEFRONT, 1>, kn, grid); });
        float t_nopart = timeit([&] { run(k_min<__HIP_MEMORY_SCOPE_AGENT, 0>, kn, grid); });
        hipDeviceSynchronize();
        hipMemcpy(ha.data(), ka, N * 8, hipMemcpyDeviceToHost); hipMemcpy(hw.data(), kw, N * 8, hipMemcpyDeviceToHost);
        long long diff = 0, touched = 0;
        for (long long i = 0; i < N; i++) { diff += ha[i] != hw[i]; touched += ha[i] != ~0ull; }
        int hh[8]; hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost);
        printf("window %lld entries: agent+part %.1f us, workgroup+part %.1f us, wavefront+part %.1f us, agent no-part %.1f us; %.2fM ops; touched %lld, agent-vs-workgroup mismatches %lld\n",
               window, t_agent, t_wg, t_wave, t_nopart, items_per_slice * 8 * 256.0 * K / 1e6, touched, diff);
        printf("  blocks per XCC id: %d %d %d %d %d %d %d %d\n", hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6], hh[7]);
    }
    return 0;
}

// micro-benchmark: cost of a posted u64 atomicMin wave-instruction as a function of the number of distinct 64-B lines its 64 lanes
// touch (L lines x 64/L lanes each; lanes of a line hit consecutive 8-B entries, wrapping at 8 per line), and the same for plain
// 8-B stores.  Lines are chosen at random inside a moving 2 MB window of a 100 MB buffer (a view's depth keys).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>   // 0 atomicMin, 1 plain store, 2 atomic on LDS-private... (unused)
__global__ void __launch_bounds__(256) k(unsigned long long* keys, long long nlines, int L, int K, int gap)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int per = 64 / L, li = lane / per, off = (lane % per) & 7;
    float dummy = (float)lane;
    for (int k2 = 0; k2 < K; k2++) {
        const unsigned h = hash32((unsigned)(wave * 64 + k2));
        // window of 32768 lines (2 MB) sliding with the wave index; the L lines are L random lines of it
        const long long wbase = (wave * 37) % (nlines - 32768);
        const long long line = wbase + hash32(h + li * 7919u) % 32768u;
        const unsigned long long val = ((unsigned long long)hash32(h ^ lane) << 32) | lane;
        if (MODE == 0) atomicMin(&keys[line * 8 + off], val);
        else keys[line * 8 + off] = val;
        for (int g = 0; g < gap; g++) dummy = dummy * 1.0001f + 0.5f;       // ALU work between the memory instructions
    }
    if (dummy == 12345.f) keys[0] = 0;
}
template <class F> float timeit(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 3; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 3 * 1000.f;
}
int main()
{
    const long long N = 48ll * 512 * 512, nlines = N / 8;
    unsigned long long* keys; (void)hipMalloc(&keys, N * 8); (void)hipMemset(keys, 0xFF, N * 8);
    const int waves = 72816, K = 2, blocks = waves / 4;            // the triangle kernel's shape: 72.8 k waves, ~1.6 atomic instructions each
    for (int gap : {0, 600})
        for (int L : {1, 2, 4, 8, 16, 32, 64}) {
            float ta = timeit([&] { k<0><<<blocks, 256>>>(keys, nlines, L, K, gap); });
            float ts = timeit([&] { k<1><<<blocks, 256>>>(keys, nlines, L, K, gap); });
            printf("gap %3d  L %2d lines/instr: atomicMin %7.1f us (%.2f G lines/s, %.1f G ops/s)   plain store %7.1f us\n", gap, L, ta,
                   (double)waves * K * L / ta / 1e3, (double)waves * K * 64 / ta / 1e3, ts);
        }
    return 0;
}

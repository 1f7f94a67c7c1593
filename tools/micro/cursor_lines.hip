// micro-benchmark behind DESIGN.md §5.6: what a bin reservation costs.  `waves` waves each make R reservations = one returning
// atomicAdd by ONE lane on a cursor chosen among the 64 cursors of its "image region" (consecutive cursor indices: neighbouring
// bins), and — in the dependent form — wait for the value before the next one, as a wave that appends to a bin must.
//   packed  : cursors are consecutive ints            -> 16 cursors share a 64-B line
//   padded  : one cursor per 64-B line (stride 16 ints)
//   posted  : the same adds without using the result (no round trip in the wave)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/cursor_lines.hip -o tools/micro/cursor_lines.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <bool RET>
__global__ void __launch_bounds__(256) k(int* cursors, int ncur, int stride, int R, int* sink, int C, int WR)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    // regions of C neighbouring cursors (the tiles one patch of 256 faces falls into); WR consecutive waves work in the same region,
    // as the workgroups of neighbouring face ranges do
    const int region = (int)((wave / WR) % (ncur / C));
    int acc = 0;
    for (int r = 0; r < R; r++) {
        const int c = region * C + (int)(hash32((unsigned)(wave * 131 + r)) % (unsigned)C);
        if (lane == 0) {
            if (RET) acc += atomicAdd(cursors + (size_t)c * stride, 1 + (acc & 1));     // the next address does not depend on it, the wave's
            else atomicAdd(cursors + (size_t)c * stride, 1);                            // progress does (s_waitcnt before the next add)
        }
        if (RET) acc = __shfl(acc, 0, 64);
    }
    if (acc == 0x7fffffff) sink[0] = acc;
}
template <class F> float timeit(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 5; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5 * 1000.f;
}
int main()
{
    const int ncur = 16384;                        // hot cursors of a step (the covered segments / tiles of 48 views)
    int *cur, *sink; (void)hipMalloc(&cur, (size_t)ncur * 32 * 4); (void)hipMemset(cur, 0, (size_t)ncur * 32 * 4); (void)hipMalloc(&sink, 16);
    const int waves = 73920, blocks = waves / 4;   // the triangle kernel's shape
    printf("%d waves, R reservations each, %d cursors; a region = C neighbouring cursors shared by WR consecutive waves\n", waves, ncur);
    for (int C : {64, 4})
        for (int WR : {64, 512})
            for (int R : {2, 6})
                for (int stride : {1, 16}) {
                    float tr = timeit([&] { k<true><<<blocks, 256>>>(cur, ncur, stride, R, sink, C, WR); });
                    float tp = timeit([&] { k<false><<<blocks, 256>>>(cur, ncur, stride, R, sink, C, WR); });
                    printf("C %2d WR %3d R %d  stride %2d ints (%2d cursors per 64-B line): returning + waited %8.1f us   posted %8.1f us   (%d adds per cursor, %d per line)\n",
                           C, WR, R, stride, stride >= 16 ? 1 : 16, tr, tp, WR * R / C, stride >= 16 ? WR * R / C : WR * R / C * (C < 16 ? C : 16));
                }
    return 0;
}

// micro-benchmark: throughput of fire-and-forget float atomicAdd / returning atomicExch under the access patterns of the HAM backward
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// each lane: K atomics to addresses derived from (lane id): mode 0 random, mode 1 "mesh-local" (lane/2 + small jitter)
__global__ void k_add(float* out, int naddr, int K, int mode, long long N)
{
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N) return;
    unsigned base = (mode == 0) ? hash32((unsigned)gid) % (unsigned)naddr
                                : (unsigned)(((gid % (naddr / 9)) + (hash32((unsigned)gid) & 7)) % (naddr / 9)) * 9;
    for (int k = 0; k < K; k++) {
        unsigned a = (mode == 0) ? (base + k * 7919u) % (unsigned)naddr : (base + (k % 9) + (k / 9) * 9 * 3) % (unsigned)naddr;
        atomicAdd(out + a, 1.0f);
    }
}
__global__ void k_exch(int* head, int* next, int naddr, long long N)
{
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N) return;
    unsigned a = hash32((unsigned)(gid / 2)) % (unsigned)naddr;
    next[gid] = atomicExch(head + a, (int)gid);
}
__global__ void k_min64(unsigned long long* keys, int naddr, long long N)
{
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N) return;
    unsigned a = hash32((unsigned)(gid / 2)) % (unsigned)naddr;
    atomicMin(keys + a, (unsigned long long)gid);
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
    const int naddr = 438237;              // 9 floats x 48693 vertices
    float* out; hipMalloc(&out, naddr * 4 + 64); hipMemset(out, 0, naddr * 4);
    long long N = 1500000;                 // owning (face, view) lanes
    for (int mode = 0; mode < 2; mode++)
        for (int K : {9, 27}) {
            float us = timeit([&] { k_add<<<(N + 255) / 256, 256>>>(out, naddr, K, mode, N); });
            printf("float atomicAdd mode %d: %lld lanes x %d = %.1fM atomics in %.1f us = %.1f G/s\n", mode, N, K, N * K / 1e6, us, N * K / us / 1e3);
        }
    int *head, *next; long long NP = 3100000; int nh = 97088 * 48;
    hipMalloc(&head, (size_t)nh * 4); hipMalloc(&next, NP * 4);
    float us = timeit([&] { k_exch<<<(NP + 255) / 256, 256>>>(head, next, nh, NP); });
    printf("returning atomicExch: %.1fM in %.1f us = %.1f G/s\n", NP / 1e6, us, NP / us / 1e3);
    unsigned long long* keys; int nk = 48 * 512 * 512; hipMalloc(&keys, (size_t)nk * 8);
    long long NK = 6000000;
    us = timeit([&] { k_min64<<<(NK + 255) / 256, 256>>>(keys, nk, NK); });
    printf("posted atomicMin u64: %.1fM in %.1f us = %.1f G/s\n", NK / 1e6, us, NK / us / 1e3);
    return 0;
}

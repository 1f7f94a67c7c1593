// micro-benchmark for fmhr_amd/csrc/peer.hip: what the pieces of the one-shot all-reduce cost on ONE GPU — copies of the 2.35 MB bucket
// between ordinary and UNCACHED device memory (hipDeviceMallocUncached: what the exported blocks are made of), and a grid barrier of
// 128 / 256 workgroups with a system-scope release fence, an agent-scope one, or none.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kB = 256;
__global__ void __launch_bounds__(kB) k_copy(float4* __restrict__ d, const float4* __restrict__ s, long long n)
{
    const long long tid = (long long)blockIdx.x * kB + threadIdx.x, nth = (long long)gridDim.x * kB;
    long long i = tid;
    for (; i + 3 * nth < n; i += 4 * nth) {
        const float4 a = s[i], b = s[i + nth], c = s[i + 2 * nth], e = s[i + 3 * nth];
        d[i] = a; d[i + nth] = b; d[i + 2 * nth] = c; d[i + 3 * nth] = e;
    }
    for (; i < n; i += nth) d[i] = s[i];
}
template <int FENCE>
__global__ void __launch_bounds__(kB) k_barrier(unsigned* bar, unsigned target)
{
    if (FENCE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (FENCE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (FENCE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
__global__ void k_empty() {}
__global__ void __launch_bounds__(kB) k_copy1(float4* __restrict__ d, const float4* __restrict__ s, long long n)
{
    const long long i = (long long)blockIdx.x * kB + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <class F> float timeit(F f, int reps = 400)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 200; i++) f(0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 1; i <= reps; i++) f(i); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main()
{
    const long long n = (12ll * 49281 + 8) / 4;                       // the metric bucket in 16-byte words
    float4 *norm, *norm2, *uc, *uc2; unsigned* bar;
    (void)hipMalloc(&norm, n * 16); (void)hipMalloc(&norm2, n * 16); (void)hipMalloc(&bar, 256);
    if (hipExtMallocWithFlags((void**)&uc, n * 16, hipDeviceMallocUncached) != hipSuccess || hipExtMallocWithFlags((void**)&uc2, n * 16, hipDeviceMallocUncached) != hipSuccess) { printf("uncached allocation failed\n"); return 1; }
    (void)hipMemset(norm, 0, n * 16); (void)hipMemset(uc, 0, n * 16); (void)hipMemset(uc2, 0, n * 16);
    printf("empty kernel back to back                      %6.1f us\n", timeit([&](int) { k_empty<<<1, 64>>>(); }));
    printf("copy 2.35 MB, one word per thread (578 blocks)      %6.1f us\n", timeit([&](int) { k_copy1<<<(int)((n + kB - 1) / kB), kB>>>(norm2, norm, n); }));
    printf("copy 2.35 MB, one word per thread, -> UNCACHED      %6.1f us\n", timeit([&](int) { k_copy1<<<(int)((n + kB - 1) / kB), kB>>>(uc, norm, n); }));
    printf("copy 2.35 MB, one word per thread, UNCACHED ->      %6.1f us\n", timeit([&](int) { k_copy1<<<(int)((n + kB - 1) / kB), kB>>>(norm, uc, n); }));
    printf("hipMemcpyAsync device to device 2.35 MB             %6.1f us\n", timeit([&](int) { (void)hipMemcpyAsync(norm2, norm, n * 16, hipMemcpyDeviceToDevice, 0); }));
    for (int blocks : {128, 256, 1024}) {
        printf("copy 2.35 MB ordinary -> ordinary, %4d blocks   %6.1f us\n", blocks, timeit([&](int) { k_copy<<<blocks, kB>>>(norm2, norm, n); }));
        printf("copy 2.35 MB ordinary -> UNCACHED, %4d blocks   %6.1f us\n", blocks, timeit([&](int) { k_copy<<<blocks, kB>>>(uc, norm, n); }));
        printf("copy 2.35 MB UNCACHED -> UNCACHED, %4d blocks   %6.1f us\n", blocks, timeit([&](int) { k_copy<<<blocks, kB>>>(uc2, uc, n); }));
        printf("copy 2.35 MB UNCACHED -> ordinary, %4d blocks   %6.1f us\n", blocks, timeit([&](int) { k_copy<<<blocks, kB>>>(norm, uc, n); }));
    }
    for (int blocks : {128, 256}) {
        unsigned cnt = 0;
        (void)hipMemset(bar, 0, 256);
        printf("grid barrier, %3d blocks, no fence              %6.1f us\n", blocks, timeit([&](int i) { k_barrier<0><<<blocks, kB>>>(bar, (cnt += blocks)); }));
        (void)hipMemset(bar, 0, 256); (void)hipDeviceSynchronize(); cnt = 0;
        printf("grid barrier, %3d blocks, agent release fence   %6.1f us\n", blocks, timeit([&](int i) { k_barrier<1><<<blocks, kB>>>(bar, (cnt += blocks)); }));
        (void)hipMemset(bar, 0, 256); (void)hipDeviceSynchronize(); cnt = 0;
        printf("grid barrier, %3d blocks, system rel + acq      %6.1f us\n", blocks, timeit([&](int i) { k_barrier<2><<<blocks, kB>>>(bar, (cnt += blocks)); }));
    }
    return 0;
}

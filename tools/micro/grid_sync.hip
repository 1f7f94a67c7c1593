// micro-benchmark: what does a grid-wide barrier cost on MI355X (8 XCDs, non-coherent L2s) compared with a kernel boundary?
//   (a) cooperative_groups::grid_group::sync() under hipLaunchCooperativeKernel
//   (b) a hand-written barrier: agent-scope release fence, one atomic arrive, polling with agent-scope loads
//   (c) two launches (the kernel boundary it would replace)
// Each variant writes a value per thread in phase A and reads a neighbour BLOCK's value in phase B (cross-XCD visibility is
// checked).  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/grid_sync.hip -o /tmp/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(256) k_coop(float* a, float* out, int rounds)
{
    cg::grid_group g = cg::this_grid();
    const int i = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    float acc = 0.f;
    for (int r = 0; r < rounds; r++) {
        a[i] = (float)(i + r);
        g.sync();
        acc += a[(i + 256 * 5) % n];           // written by a block five further on (another XCD)
        g.sync();
    }
    out[i] = acc;
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_hand(float* a, float* out, int rounds, unsigned* ctr)
{
    const int i = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    float acc = 0.f;
    unsigned t = 0;
    for (int r = 0; r < rounds; r++) {
        a[i] = (float)(i + r);
        t += gridDim.x; grid_barrier(ctr, t);
        acc += __builtin_nontemporal_load(&a[(i + 256 * 5) % n]);
        t += gridDim.x; grid_barrier(ctr, t);
    }
    out[i] = acc;
}

__global__ void __launch_bounds__(256) k_a(float* a, int r) { const int i = blockIdx.x * 256 + threadIdx.x; a[i] = (float)(i + r); }
__global__ void __launch_bounds__(256) k_b(const float* a, float* out) { const int i = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256; out[i] += a[(i + 256 * 5) % n]; }

int main()
{
    const int blocks = 762, n = blocks * 256, rounds = 50;
    float *a, *out; unsigned* ctr;
    hipMalloc(&a, n * 4); hipMalloc(&out, n * 4); hipMalloc(&ctr, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> h(n);
    auto check = [&](const char* what) {
        hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; i++) {
            double want = 0; for (int r = 0; r < rounds; r++) want += (double)((i + 256 * 5) % n + r);
            if (fabs(h[i] - want) > 1e-3 * want + 1) bad++;
        }
        printf("  %s: %ld wrong of %d\n", what, bad, n);
    };
    float ms;
    {   // (a)
        int rr = rounds; void* args[] = {&a, &out, &rr};
        hipError_t e = hipLaunchCooperativeKernel((void*)k_coop, dim3(blocks), dim3(256), args, 0, 0);
        printf("cooperative launch: %s\n", hipGetErrorString(e));
        hipDeviceSynchronize();
        hipEventRecord(e0); e = hipLaunchCooperativeKernel((void*)k_coop, dim3(blocks), dim3(256), args, 0, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("(a) cg grid.sync(): %.2f us per sync (%d syncs, kernel %.1f us)\n", ms * 1000 / (2 * rounds), 2 * rounds, ms * 1000);
        check("cg");
    }
    {   // (b)
        hipMemset(ctr, 0, 256); hipMemset(out, 0, n * 4);
        k_hand<<<blocks, 256>>>(a, out, rounds, ctr); hipDeviceSynchronize();
        hipMemset(ctr, 0, 256);
        hipEventRecord(e0); k_hand<<<blocks, 256>>>(a, out, rounds, ctr); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("(b) hand-written barrier: %.2f us per barrier (kernel %.1f us)\n", ms * 1000 / (2 * rounds), ms * 1000);
        check("hand");
    }
    {   // (c)
        hipMemset(out, 0, n * 4);
        for (int r = 0; r < rounds; r++) { k_a<<<blocks, 256>>>(a, r); k_b<<<blocks, 256>>>(a, out); }
        hipDeviceSynchronize(); hipMemset(out, 0, n * 4);
        hipEventRecord(e0);
        for (int r = 0; r < rounds; r++) { k_a<<<blocks, 256>>>(a, r); k_b<<<blocks, 256>>>(a, out); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("(c) two launches: %.2f us per launch (%d launches back to back)\n", ms * 1000 / (2 * rounds), 2 * rounds);
        check("launches");
    }
    return 0;
}

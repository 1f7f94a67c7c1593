// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU operations the rasterizer kernels are made of.
// Every wave runs ITER x 8 independent chains of one operation; 8 waves per SIMD on every SIMD, so the SIMD's issue port is the
// only limit.  cycles/instr = elapsed * clock * SIMDs / (waves * instrs per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4096;
#define CHAIN8(init, op) \
    for (int k = 0; k < 8; k++) r[k] = init; \
    for (int i = 0; i < ITER; i++) { _Pragma("unroll") for (int k = 0; k < 8; k++) { op; } }
__global__ void __launch_bounds__(256) k_fma(float* out, float a, float b) { float r[8]; CHAIN8(threadIdx.x * a + k, r[k] = __fmaf_rn(r[k], a, b)); float s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_mullo(int* out, int a) { int r[8]; CHAIN8(threadIdx.x + k, r[k] = r[k] * a + 1); int s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_mul24(int* out, int a) { int r[8]; CHAIN8(threadIdx.x + k, r[k] = __mul24(r[k], a) + 1); int s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_mul64(long long* out, int a) { long long r[8]; CHAIN8(threadIdx.x + k, r[k] = (long long)(int)r[k] * (long long)a + 1); long long s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_shl64(unsigned long long* out, int a) { unsigned long long r[8]; CHAIN8(threadIdx.x + k, r[k] |= 1ull << ((r[k] + a) & 63)); unsigned long long s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_div(float* out, float a) { float r[8]; CHAIN8(threadIdx.x * 0.5f + k + 1.f, r[k] = a / r[k] + 1.0f); float s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_rcp(float* out, float a) { float r[8]; CHAIN8(threadIdx.x * 0.5f + k + 1.f, r[k] = __builtin_amdgcn_rcpf(r[k]) + a); float s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_add(int* out, int a) { int r[8]; CHAIN8(threadIdx.x + k, r[k] = (r[k] + a) ^ k); int s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_ffbl(int* out, int a) { unsigned long long r[8]; CHAIN8(threadIdx.x + k + 1, r[k] = r[k] * 3 + __builtin_ctzll(r[k] | 0x8000000000000000ull) + a); unsigned long long s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = (int)s; }
template <class F> float timeit(F f)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 3; i++) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 3;
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8;          // 8 blocks x 4 waves per CU = 8 waves per SIMD
    void* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
    const double clk = p.clockRate * 1e3, simds = cus * 4.0, waves = blocks * 4.0;
    auto rep = [&](const char* name, float ms, double ops_per_iter) {
        printf("%-28s %8.3f ms  -> %.2f cycles per wave-op per SIMD (nominal clock %.0f MHz, %d CUs)\n", name, ms, ms * 1e-3 * clk * simds / (waves * ITER * 8.0 * ops_per_iter), clk / 1e6, cus);
    };
    rep("v_fma_f32", timeit([&] { k_fma<<<blocks, 256>>>((float*)out, 1.0001f, 0.5f); }), 1);
    rep("add+xor (2 ops)", timeit([&] { k_add<<<blocks, 256>>>((int*)out, 3); }), 2);
    rep("mul_lo_u32 + add (mad?)", timeit([&] { k_mullo<<<blocks, 256>>>((int*)out, 3); }), 1);
    rep("mul_i24 + add", timeit([&] { k_mul24<<<blocks, 256>>>((int*)out, 3); }), 1);
    rep("64-bit product + add", timeit([&] { k_mul64<<<blocks, 256>>>((long long*)out, 3); }), 1);
    rep("1ull << n, or (64-bit)", timeit([&] { k_shl64<<<blocks, 256>>>((unsigned long long*)out, 3); }), 1);
    rep("IEEE fp32 divide + add", timeit([&] { k_div<<<blocks, 256>>>((float*)out, 3.f); }), 1);
    rep("v_rcp_f32 + add", timeit([&] { k_rcp<<<blocks, 256>>>((float*)out, 3.f); }), 1);
    rep("ctz64 + mul3 + adds", timeit([&] { k_ffbl<<<blocks, 256>>>((int*)out, 3); }), 1);
    return 0;
}

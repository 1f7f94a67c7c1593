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
// round 5: the double operations of rule R5's values (common.h: fragment_uv / r5_bwd_wide)
__global__ void __launch_bounds__(256) k_fma64(double* out, double a, double b) { double r[8]; CHAIN8(threadIdx.x * a + k, r[k] = __builtin_fma(r[k], a, b)); double s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_mul64f(double* out, double a) { double r[8]; CHAIN8(threadIdx.x * a + k + 1.0, r[k] = r[k] * a); double s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_add64f(double* out, double a) { double r[8]; CHAIN8(threadIdx.x * a + k + 1.0, r[k] = r[k] + a); double s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_cvt64(float* out, float a) { float r[8]; CHAIN8(threadIdx.x * a + k + 1.0f, r[k] = (float)((double)r[k] + 1.0)); float s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_rcp64(double* out, double a) { double r[8]; CHAIN8(threadIdx.x * 0.5 + k + 1.0, r[k] = __builtin_amdgcn_rcp(r[k]) + a); double s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
__global__ void __launch_bounds__(256) k_div64(double* out, double a) { double r[8]; CHAIN8(threadIdx.x * 0.5 + k + 1.0, r[k] = a / r[k] + 1.0); double s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
// accuracy of v_rcp_f64 and of one / two Newton steps on it, against the IEEE division: max relative error over 64 k arguments
__global__ void __launch_bounds__(256) k_rcp64_acc(double* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double x = 1.0e-12 * (1.0 + (double)i * 0.37) * (((i >> 3) & 1) ? -1.0 : 1.0), ref = 1.0 / x;
    double r0 = __builtin_amdgcn_rcp(x);
    double r1 = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
    double r2 = __builtin_fma(__builtin_fma(-x, r1, 1.0), r1, r1);
    out[i * 3 + 0] = fabs((r0 - ref) / ref); out[i * 3 + 1] = fabs((r1 - ref) / ref); out[i * 3 + 2] = fabs((r2 - ref) / ref);
}
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
    rep("v_fma_f64", timeit([&] { k_fma64<<<blocks, 256>>>((double*)out, 1.0001, 0.5); }), 1);
    rep("v_mul_f64", timeit([&] { k_mul64f<<<blocks, 256>>>((double*)out, 1.0001); }), 1);
    rep("v_add_f64", timeit([&] { k_add64f<<<blocks, 256>>>((double*)out, 1.0001); }), 1);
    rep("cvt f32->f64, add, cvt back (3)", timeit([&] { k_cvt64<<<blocks, 256>>>((float*)out, 1.0001f); }), 3);
    rep("v_rcp_f64 + add (2 ops)", timeit([&] { k_rcp64<<<blocks, 256>>>((double*)out, 3.0); }), 2);
    rep("IEEE fp64 divide + add", timeit([&] { k_div64<<<blocks, 256>>>((double*)out, 3.0); }), 1);
    {
        double* acc; (void)hipMalloc(&acc, 65536 * 3 * sizeof(double));
        k_rcp64_acc<<<256, 256>>>(acc); (void)hipDeviceSynchronize();
        double* h = (double*)malloc(65536 * 3 * sizeof(double)); (void)hipMemcpy(h, acc, 65536 * 3 * sizeof(double), hipMemcpyDeviceToHost);
        double m[3] = {0, 0, 0};
        for (int i = 0; i < 65536; i++) for (int k = 0; k < 3; k++) if (h[i * 3 + k] > m[k]) m[k] = h[i * 3 + k];
        printf("v_rcp_f64 max relative error vs IEEE 1/x: raw %.3e, one Newton step %.3e, two %.3e\n", m[0], m[1], m[2]);
    }
    return 0;
}

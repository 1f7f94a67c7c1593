// fmhr_amd/csrc/common.h — shared device helpers for the gfx950 HAM kernels.
// The whole library is compiled with -ffp-contract=off: every statement below is one IEEE fp32 operation, so the discrete
// decisions (snap, coverage, depth order, antialias edge selection) are bit-identical on every launch and can be
// re-stated exactly by a CPU checker.  Where a fused multiply-add is wanted for speed it is written as fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/fmhr_ham.h"

namespace fmhr {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
void prof_mark(int slot, int end, hipStream_t st);   // abi.cpp: records a timing event when a sink is attached

struct ProfScope {                                   // brackets a launch group with the optional timing events
    int slot; hipStream_t st;
    ProfScope(int s, hipStream_t t) : slot(s), st(t) { prof_mark(slot, 0, st); }
    ~ProfScope() { prof_mark(slot, 1, st); }
};

#define FMHR_CHECK_ARG(cond, ...)                     \
    do {                                              \
        if (!(cond)) {                                \
            ::fmhr::set_error(__VA_ARGS__);           \
            return FMHR_E_ARG;                        \
        }                                             \
    } while (0)

bool debug_sync();                                   // abi.cpp: FMHR_DEBUG_SYNC=1 -> synchronise + log after every launch

#define FMHR_CHECK_LAUNCH(what)                                    \
    do {                                                           \
        hipError_t e__ = hipGetLastError();                        \
        if (e__ != hipSuccess) return ::fmhr::hip_fail(e__, what);  \
        if (::fmhr::debug_sync()) {                                \
            fprintf(stderr, "[fmhr] %s ...", what);                \
            fflush(stderr);                                        \
            e__ = hipDeviceSynchronize();                          \
            fprintf(stderr, " %s\n", hipGetErrorString(e__));      \
            if (e__ != hipSuccess) return ::fmhr::hip_fail(e__, what); \
        }                                                          \
    } while (0)

// ---- wave-level data movement without the LDS crossbar: __shfl_up / __shfl compile to ds_bpermute_b32 (an LDS instruction: it
// queues behind the kernel's real LDS traffic); the DPP forms below are plain vector-ALU operand modifiers.
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ int dpp_int(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xF, false);   // lanes without a source (or masked rows) read 0
}
// inclusive prefix sum over the 64 lanes of a wave: four row_shr steps inside each row of 16, then the row totals via row_bcast
__device__ __forceinline__ int wave_incl_scan(int x)
{
    x += dpp_int<0x111>(x);            // row_shr:1
    x += dpp_int<0x112>(x);            // row_shr:2
    x += dpp_int<0x114>(x);            // row_shr:4
    x += dpp_int<0x118>(x);            // row_shr:8
    x += dpp_int<0x142, 0xA>(x);       // row_bcast:15 -> rows 1, 3
    x += dpp_int<0x143, 0xC>(x);       // row_bcast:31 -> rows 2, 3
    return x;
}
// value of the previous lane (lane 0 reads `first`)
__device__ __forceinline__ int wave_prev_lane(int x, int first)
{
    return __builtin_amdgcn_update_dpp(first, x, 0x138, 0xF, 0xF, false);   // wave_shr:1
}

constexpr int kBlock = 256;
constexpr int kAaReplicas = 8;   // copies of the antialias-backward vertex accumulators (antialias.hip: k_aa_bwd_items)
constexpr int kSub = 256;        // sub-pixel units per pixel (8 bits)
constexpr int kHalf = 128;
constexpr float kSnapLim = 4194304.0f;   // 2^22 sub-pixel guard band

static inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }

struct Frag { float u, v, zw; };

// DESIGN.md §3 R1+R2: whole-triangle near/far reject and snap to 1/256 px.
__device__ __forceinline__ bool snap_vertex(const float4 p, float kx, float ky, int& ox, int& oy)
{
    if (!(p.w > 0.0f) || !(p.z >= -p.w) || !(p.z <= p.w)) return false;
    float rw = 1.0f / p.w;
    float sx = p.x * rw;
    float sy = p.y * rw;
    float qx = sx * kx;
    qx = qx + kx;
    float qy = sy * ky;
    qy = qy + ky;
    if (!(qx == qx) || !(qy == qy)) return false;
    if (qx > kSnapLim) qx = kSnapLim;
    if (qx < -kSnapLim) qx = -kSnapLim;
    if (qy > kSnapLim) qy = kSnapLim;
    if (qy < -kSnapLim) qy = -kSnapLim;
    ox = (int)rintf(qx);
    oy = (int)rintf(qy);
    return true;
}

// edge test with the owner rule for pixel centres exactly on an edge (CCW-normalised triangle).
__device__ __forceinline__ bool edge_in(int ax, int ay, int bx, int by, int cx, int cy)
{
    long long dx = bx - ax, dy = by - ay;
    long long e = dx * (long long)(cy - ay) - dy * (long long)(cx - ax);
    if (e > 0) return true;
    if (e < 0) return false;
    return (dy > 0) || (dy == 0 && dx > 0);
}

// ---- rule R5.  DECISIONS — accept / reject of a fragment, its depth z/w and with it the depth order — are fp32, one
// IEEE operation per statement (fragment_zw).  VALUES — the barycentrics (u, v) a winner returns, and their gradient with respect to
// the clip positions (shade.hip: pixel_backward, raster.hip: k_raster_bwd) — are evaluated in DOUBLE from the same fp32 clip positions
// and rounded once (round 5): nvdiffrast's pixel-relative edge functions x - fx w cancel ~10 bits at the reference capture's clip-space
// scale (|x| ~ 2, a pixel 1.5e-3 of it) and a thin triangle divides what is left by its area: evaluated in fp32 — in ANY fp32 order —
// the position gradient of the thinnest triangles of the capture's own cameras was 8e-4 of the gradient's maximum away from exact
// arithmetic (profiles/parity_r04.txt; the oracle's fp32 twin showed the same), in double it is 2e-5 (tools/experiments/r05_r5_wide.py).
// gfx950 issues v_fma_f64 at the rate of the un-packed v_fma_f32 (tools/micro/valu_rate.hip): the ~45 double operations of a winner cost
// what the fp32 ones did.  The CPU checker's fp32 twin holds the same statements (r5_uv_wide / r5_bwd_wide there): `rast` stays bit-identical.
struct PixGrid { float xs, xo, ys, yo; double iW, iH; int W, H; };

// pixel-relative homogeneous coordinates p_i - f w_i in double; the pixel centre f = (2 px + 1 - W) / W as an integer times 1 / W
__device__ __forceinline__ void r5_wide_setup(float p0cx, float p0cy, float p0w, float p1cx, float p1cy, float p1w, float p2cx, float p2cy, float p2w,
                                              int px, int py, const PixGrid& pg, double (&q)[6], double& fx, double& fy)
{
    fx = (double)(2 * px + 1 - pg.W); fx = fx * pg.iW;
    fy = (double)(2 * py + 1 - pg.H); fy = fy * pg.iH;
    double t;
    t = fx * (double)p0w; q[0] = (double)p0cx - t;  t = fy * (double)p0w; q[1] = (double)p0cy - t;
    t = fx * (double)p1w; q[2] = (double)p1cx - t;  t = fy * (double)p1w; q[3] = (double)p1cy - t;
    t = fx * (double)p2w; q[4] = (double)p2cx - t;  t = fy * (double)p2w; q[5] = (double)p2cy - t;
}

// (u, v) of a fragment fragment_zw() has accepted (x, y, w of the three clip positions), clamped to [0, 1]
__device__ __forceinline__ void fragment_uv(float p0cx, float p0cy, float p0w, float p1cx, float p1cy, float p1w, float p2cx, float p2cy, float p2w,
                                            int px, int py, const PixGrid& pg, float& u, float& v)
{
    double q[6], fx, fy;
    r5_wide_setup(p0cx, p0cy, p0w, p1cx, p1cy, p1w, p2cx, p2cy, p2w, px, py, pg, q, fx, fy);
    double m0, m1;
    m0 = q[2] * q[5]; m1 = q[3] * q[4]; const double a0 = m0 - m1;
    m0 = q[4] * q[1]; m1 = q[5] * q[0]; const double a1 = m0 - m1;
    m0 = q[0] * q[3]; m1 = q[1] * q[2]; const double a2 = m0 - m1;
    double at = a0 + a1; at = at + a2;
    const double iw = 1.0 / at;                                      // IEEE division (the result is part of `rast`: bit-identical to the checker)
    const double b0 = a0 * iw;
    const double b1 = a1 * iw;
    float uf = (float)b0, vf = (float)b1;
    if (!(uf >= 0.0f)) uf = 0.0f;                                    // (also a NaN from a zero area in double: the fp32 accept test had passed)
    if (uf > 1.0f) uf = 1.0f;
    if (!(vf >= 0.0f)) vf = 0.0f;
    if (vf > 1.0f) vf = 1.0f;
    u = uf; v = vf;
}

// d (u, v) / d clip positions on the UNCLAMPED barycentrics, in double: gu, gv = d loss / d (u, v);
// qx / qy / qw [k] = d loss / d (x, y, w) of vertex k, rounded to float at the end.  Value-only (nothing discrete hangs on it): the
// reciprocal of the area is v_rcp_f64 refined by one Newton step instead of the ~15-instruction IEEE division (62 cycles per wave).
__device__ __forceinline__ void r5_bwd_wide(float p0cx, float p0cy, float p0w, float p1cx, float p1cy, float p1w, float p2cx, float p2cy, float p2w,
                                            int px, int py, const PixGrid& pg, float gu, float gv, float (&qx)[3], float (&qy)[3], float (&qw)[3])
{
    double q[6], fx, fy;
    r5_wide_setup(p0cx, p0cy, p0w, p1cx, p1cy, p1w, p2cx, p2cy, p2w, px, py, pg, q, fx, fy);
    const double p0x = q[0], p0y = q[1], p1x = q[2], p1y = q[3], p2x = q[4], p2y = q[5];
    const double e0 = p1x * p2y - p1y * p2x, e1 = p2x * p0y - p2y * p0x, e2 = p0x * p1y - p0y * p1x;
    const double at = e0 + e1 + e2;
    double iw = __builtin_amdgcn_rcp(at);                             // 24 bits (measured 4.5e-8); one Newton step: 2e-15 (tools/micro/valu_rate.hip)
    iw = __builtin_fma(__builtin_fma(-at, iw, 1.0), iw, iw);
#ifndef FMHR_X_R5_ALL_DOUBLE
    // The ill-conditioned part ends here: the barycentrics, the reciprocal area and the EDGE DIFFERENCES of the accurate pixel-relative
    // coordinates are rounded to float once each, the ~30 well-conditioned multiply-adds behind them run in fp32 (same-box A/B at the
    // metric, k_face_bwd: all-fp32 66.4 us, all of it in double 69.2, this form 68.3; the capture-camera parity case 2.1e-5 either way)
    const float b0 = (float)(e0 * iw), b1 = (float)(e1 * iw), iwf = (float)iw;
    const float gb0 = gu * iwf, gb1 = gv * iwf, gbb = gb0 * b0 + gb1 * b1;
    const float d21y = (float)(p2y - p1y), d02y = (float)(p0y - p2y), d10y = (float)(p1y - p0y);
    const float d12x = (float)(p1x - p2x), d20x = (float)(p2x - p0x), d01x = (float)(p0x - p1x);
    const float f0x = (float)p0x, f0y = (float)p0y, f1x = (float)p1x, f1y = (float)p1y, f2x = (float)p2x, f2y = (float)p2y;
    const float fxf = (float)fx, fyf = (float)fy;
    qx[0] = gbb * d21y - gb1 * f2y; qx[1] = gbb * d02y + gb0 * f2y; qx[2] = gbb * d10y - gb0 * f1y + gb1 * f0y;
    qy[0] = gbb * d12x + gb1 * f2x; qy[1] = gbb * d20x - gb0 * f2x; qy[2] = gbb * d01x + gb0 * f1x - gb1 * f0x;
#pragma unroll
    for (int k = 0; k < 3; k++) qw[k] = -fxf * qx[k] - fyf * qy[k];
#else
    const double gb0 = (double)gu * iw, gb1 = (double)gv * iw, gbb = gb0 * (e0 * iw) + gb1 * (e1 * iw);
    const double x0 = gbb * (p2y - p1y) - gb1 * p2y, x1 = gbb * (p0y - p2y) + gb0 * p2y, x2 = gbb * (p1y - p0y) - gb0 * p1y + gb1 * p0y;
    const double y0 = gbb * (p1x - p2x) + gb1 * p2x, y1 = gbb * (p2x - p0x) - gb0 * p2x, y2 = gbb * (p0x - p1x) + gb0 * p1x - gb1 * p0x;
    qx[0] = (float)x0; qx[1] = (float)x1; qx[2] = (float)x2;
    qy[0] = (float)y0; qy[1] = (float)y1; qy[2] = (float)y2;
    qw[0] = (float)(-fx * x0 - fy * y0); qw[1] = (float)(-fx * x1 - fy * y1); qw[2] = (float)(-fx * x2 - fy * y2);
#endif
}

// rule R5, the depth only: the SAME statements as fragment() for everything z/w and the accept / reject decision depend on (the
// barycentrics' reciprocal and clamps are left out): the binned rasterizer's triangle kernel evaluates this for every fragment —
// 60 % of them are hidden behind a nearer one — and the second pass evaluates fragment() in full for the ~2.4 M winners only.
__device__ __forceinline__ bool fragment_zw(const float4 p0, const float4 p1, const float4 p2, int px, int py,
                                            float xs, float xo, float ys, float yo, float& zw)
{
    float fx = xs * (float)px; fx = fx + xo;
    float fy = ys * (float)py; fy = fy + yo;
    float t;
    t = fx * p0.w; float p0x = p0.x - t;  t = fy * p0.w; float p0y = p0.y - t;
    t = fx * p1.w; float p1x = p1.x - t;  t = fy * p1.w; float p1y = p1.y - t;
    t = fx * p2.w; float p2x = p2.x - t;  t = fy * p2.w; float p2y = p2.y - t;
    float m0, m1;
    m0 = p1x * p2y; m1 = p1y * p2x; float a0 = m0 - m1;
    m0 = p2x * p0y; m1 = p2y * p0x; float a1 = m0 - m1;
    m0 = p0x * p1y; m1 = p0y * p1x; float a2 = m0 - m1;
    float at = a0 + a1; at = at + a2;
    if (!(fabsf(at) > 0.0f)) return false;
    float zn = p0.z * a0; t = p1.z * a1; zn = zn + t; t = p2.z * a2; zn = zn + t;
    float wn = p0.w * a0; t = p1.w * a1; wn = wn + t; t = p2.w * a2; wn = wn + t;
    float q = zn / wn;
    if (!(q == q)) return false;
    if (q < -1.0f) q = -1.0f;
    if (q > 1.0f) q = 1.0f;
    q = q + 0.0f;
    zw = q;
    return true;
}

// rule R5 in full: the fp32 decisions + depth of fragment_zw(), the barycentrics of fragment_uv()
__device__ __forceinline__ bool fragment(const float4 p0, const float4 p1, const float4 p2, int px, int py, const PixGrid& pg, Frag& o)
{
    float zw;
    if (!fragment_zw(p0, p1, p2, px, py, pg.xs, pg.xo, pg.ys, pg.yo, zw)) return false;
    fragment_uv(p0.x, p0.y, p0.w, p1.x, p1.y, p1.w, p2.x, p2.y, p2.w, px, py, pg, o.u, o.v);
    o.zw = zw;
    return true;
}

// NDC coordinates of a pixel centre (rule R5): fx = xs * px + xo, fy = ys * py + yo.  Computed ONCE on the host (IEEE fp32, one
// rounding per operation: bit-identical to the same statements on the device) and handed to the kernels: evaluated per wave the
// four correctly rounded divisions cost ~45 vector instructions, 6 % of the triangle kernel.
inline PixGrid make_pix_grid(int H, int W)
{
    PixGrid g;
    volatile float iw = 1.0f / (float)W, ih = 1.0f / (float)H;      // volatile: no host-side re-association / wider intermediates
    g.xs = 2.0f / (float)W; g.ys = 2.0f / (float)H;
    g.xo = iw - 1.0f; g.yo = ih - 1.0f;
    g.iW = 1.0 / (double)W; g.iH = 1.0 / (double)H; g.W = W; g.H = H;   // rule R5's values (double): the pixel centre is (2 px + 1 - W) * iW
    return g;
}

// order-preserving map float -> uint32 (smaller float => smaller key)
__device__ __forceinline__ uint32_t mono_key(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Element `idx` of a table through a 32-BIT byte offset: base (wave-uniform: SGPR pair) + zero-extended idx * sizeof(T) selects the
// `global_load ... v_off, s[base]` form — one VGPR and one shift per address instead of a sign extension, a 64-bit multiply-add and a
// VGPR pair.  For per-vertex / per-face tables (V * 32 B, F * 12 B ... < 4 GB by a wide margin); idx must be a valid index (>= 0).
template <typename T>
__device__ __forceinline__ const T& at32(const T* base, unsigned idx)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(idx * (unsigned)sizeof(T)));
}
template <typename T>
__device__ __forceinline__ T& at32(T* base, unsigned idx)
{
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)(idx * (unsigned)sizeof(T)));
}

// wave64 sum reduction, the total in every lane: the DPP scan of wave_incl_scan (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31) and a
// readlane of lane 63 — six vector-ALU operand modifiers instead of six dependent ds_bpermute round trips through the LDS crossbar
// (the sums sit at the very end of their kernels' chains).  The tree differs from the shuffle form's: partial sums of the losses move
// in the last bit.
__device__ __forceinline__ float wave_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// STREAMING accesses (the `nt` cache policy: allocate with the lowest retention priority).  The step's per-pixel arrays are written by
// one kernel and read once by a later one — 200 MB a piece at the metric, 800 MB at 48 x 1024^2 — and on their way through the 4-MB L2 of
// an XCD they push out the per-vertex / per-face tables (3.4 MB) every gather of the same kernels goes to.  With the per-pixel stores and
// the read-once loads of k_raster_bins marked: 74.8 -> 73.4 us at the metric, 387 -> 344 us at 48 x 1024^2 (C4 950 -> 1 009 it/s).
// Also kept: the per-pixel loads of k_loss_pix (shade.hip: +1 % over five alternating rounds at the metric, +0.6 % at C4).
// Tried and NOT kept (same-box A/B, metric / C4): k_loss_pix's gradient store (-0.5 %), the corner-sum stores of
// k_face_bwd (0 / -1 %), k_face_bwd's per-candidate loads (a pixel's word is read by ~2.5 candidate faces: 68 -> 94 us), the coverage
// entries of k_raster_tris (+0.2 % / +0.3 %: noise), its fragment entries (scattered 8-B stores want the L2's write combining: bins 72 -> 80 us).
typedef float v4f_stream __attribute__((ext_vector_type(4)));
typedef unsigned v2u_stream __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream(float4* p, const float4 v)
{
    v4f_stream t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f_stream*>(p));
}
__device__ __forceinline__ float4 ld_stream(const float4* p)
{
    const v4f_stream t = __builtin_nontemporal_load(reinterpret_cast<const v4f_stream*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 ld_stream(const uint2* p)
{
    const v2u_stream t = __builtin_nontemporal_load(reinterpret_cast<const v2u_stream*>(p));
    return make_uint2(t.x, t.y);
}
__device__ __forceinline__ float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }

struct alignas(4) F3U { float x, y, z; };
struct alignas(4) I3U { int x, y, z; };                            // a row of a [*,3] int32 table as ONE 12-byte load
struct alignas(4) F4U { float x, y, z, w; };     // a 16-B load from a dword-aligned address (global loads need no more)

// keep a loaded 16-B word whole and its load where it was written: without this the compiler splits a float4 load whose
// components are used at different times into several requests and sinks part of it below the first branch (an extra dependent
// round trip); the per-pixel kernels are bound by the NUMBER of vector-memory requests
__device__ __forceinline__ void pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// Exact g / d and g % d (0 <= g, 1 <= d < 2^31, quotient < 2^22) from a float estimate plus integer correction: the compiler's
// 64-bit software division costs ~200 VALU instructions per lane, and every per-pixel / per-(view, triangle) kernel splits its
// flat index this way.  With a relative error below 2^-22 in the float product and a quotient below 2^22 the truncated estimate is
// within one of the true quotient, so one correction each way makes it exact (correction LOOPS would be turned back into
// divisions by the loop optimiser).
__device__ __forceinline__ int fast_divmod(long long g, int d, int& r)
{
    int q = (int)((float)g * __builtin_amdgcn_rcpf((float)d));
    long long rem = g - (long long)q * d;
    if (rem < 0) { rem += d; q--; }
    if (rem >= d) { rem -= d; q++; }
    r = (int)rem;
    return q;
}

// Division of a wave-uniform unsigned value by a launch constant on the SCALAR unit: q = (mulhi(n, m) + n) >> l with
// l = ceil(log2 d), m = ceil(2^(32+l) / d) - 2^32 (Granlund-Montgomery; exact for n < 2^31, d >= 1).  Used where a per-lane
// integer division of a flat index would cost ~25 vector instructions per lane for a value the whole wave shares.
struct UDiv { unsigned m, l, d; };
inline UDiv make_udiv(unsigned d)
{
    UDiv u; u.d = d; u.l = 0;
    while ((1ull << u.l) < d) u.l++;
    const unsigned long long two32 = 1ull << 32;
    // ceil(2^(32+l) / d) - 2^32, in 128-bit-free arithmetic: 2^(32+l) = two32 << l fits 64 bits for l <= 31
    const unsigned long long num = two32 << u.l;
    u.m = (unsigned)((num + d - 1) / d - two32);
    return u;
}
__device__ __forceinline__ unsigned udiv(unsigned n, const UDiv& u) { return (__umulhi(n, u.m) + n) >> u.l; }

// ---- row pitch (ABI 5).  The fused step's own per-pixel buffers are [n, H, Wp] with Wp = the pitch: by default W rounded up to a
// multiple of 64, so that every 64-pixel segment (= one wave of every per-pixel kernel, = one fragment bin) lies in ONE row of ONE
// view whatever W is (W = 334: /root/reference/conf/ih_sfs.conf:14).  The pad columns are never covered (triangle boxes are clipped
// to W), never a pair of the antialias scan (a pair needs px + 1 < W) and never read from a caller tensor: img / mask / valid_mask
// stay [n, H, W].  W keeps every GEOMETRIC meaning (pixel grid, snap scale, clipping), Wp only addresses.
inline int default_row_pitch(int W) { return (W + 63) & ~63; }
// caller-tensor index of lane `lane` of segment `seg` (wave-uniform) when the pitch differs from W (then Wp % 64 == 0): the segment's
// global row b * H + py by one scalar magic division; `incol` = the lane is a pixel of the image, not of the pad
__device__ __forceinline__ long long caller_index(unsigned seg, int lane, int W, const UDiv& segx, bool& incol)
{
    const unsigned row = udiv(seg, segx), col0 = (seg - row * segx.d) << 6;
    incol = (int)col0 + lane < W;
    return (long long)row * W + (int)col0 + lane;
}

// ---- view selection (ABI 6).  A step may render a SUBSET of the capture's views — the minibatch perm[k:k+batch] of
// /root/reference/mesh_sfs_optim.py:249,255-260 — while the caller's per-view arrays (img, mask, valid_mask [n_src,H,W(,3)]; sh, w2c, proj)
// hold the whole capture: view b of the step is view idx[b] of those arrays.  Nothing is gathered: the four kernels that read the big
// arrays turn their segment into the SOURCE view's row with two scalar magic divisions and one scalar load (a segment lies in one row of
// one view: H * pitch % 64 == 0 is required), the geometry kernel reads the small per-view rows through idx once and leaves them in the
// workspace in the step's own order.  idx == nullptr: the arrays hold exactly the step's n views (the ABI-5 meaning).
typedef const int __attribute__((address_space(4))) const_as4_i32;
struct ViewSel {
    const int32_t* idx;       // device int32 [n] or null
    UDiv segv;                // division by the segments per view (H * pitch / 64)
    int H;
    __device__ __forceinline__ int src(unsigned b) const             // wave-uniform b: through the scalar cache (the caller's array is constant during a step)
    {
        return *(const_as4_i32*)(unsigned long long)(idx + b);
    }
};
// caller_index through a view selection (sel.idx != nullptr): seg wave-uniform, Wp % 64 == 0
__device__ __forceinline__ long long caller_index_sel(unsigned seg, int lane, int W, const UDiv& segx, const ViewSel& sel, bool& incol)
{
    const unsigned row = udiv(seg, segx), col0 = (seg - row * segx.d) << 6;          // row = b * H + py of the step's numbering
    const unsigned b = udiv(seg, sel.segv);
    incol = (int)col0 + lane < W;
    return ((long long)row + (long long)(sel.src(b) - (int)b) * sel.H) * W + (int)col0 + lane;
}

// one clip coordinate of a vertex under one column (c0..c3) of the combined matrix w2c @ proj: the geometry kernel and the
// face-parallel backward both go through this explicit chain, so their positions agree bit for bit whatever the compiler contracts
__device__ __forceinline__ float clip_coord(float x, float y, float z, float c0, float c1, float c2, float c3)
{
    return __fmaf_rn(x, c0, __fmaf_rn(y, c1, __fmaf_rn(z, c2, c3)));
}

// Large-box triangles (box > 64 pixels) go through a queue and dedicated kernels (forward: k_raster_large, backward:
// k_face_bwd_large, one workgroup / wave per (view, triangle)) when the mesh is coarse for the resolution — at least 8 pixels
// per face at 50 % coverage: most boxes are large and the per-(view, triangle)-lane kernels would have a few dozen workgroups
// with minutes-long lanes; on a micro-triangle mesh the odd large box is handled in place and those kernels are not launched
inline bool raster_use_queue(int H, int W, int F) { return (long long)H * W > 16ll * F; }

// Clip positions of one view: a stored [V,4] array (generic ops: the caller's `pos`) or — fused step — recomputed from the
// vertices and the view's combined matrix with the same clip_coord() chains wherever they are needed: the step never stores them
// (37 MB per 48 views written, then gathered at random by the rasterizer passes; the [V,3] vertex array stays in cache).
struct ClipView {
    const float* posv;        // [V,4] or null
    const float* verts;       // [V,3] (used when posv is null)
    const float4* cm;         // the view's 4 rows of w2c @ proj (columns x, y, z, w); loaded at the use: the callers are the rare
                              // paths (silhouette pairs, queued large triangles) and should not carry 16 registers around
    __device__ __forceinline__ float4 get(int v) const
    {
        if (posv != nullptr) return ldg4(posv + (size_t)v * 4);
        const float x = verts[(size_t)v * 3], y = verts[(size_t)v * 3 + 1], z = verts[(size_t)v * 3 + 2];
        const float4 c0 = cm[0], c1 = cm[1], c2 = cm[2], c3 = cm[3];
        return make_float4(clip_coord(x, y, z, c0.x, c1.x, c2.x, c3.x), clip_coord(x, y, z, c0.y, c1.y, c2.y, c3.y),
                           clip_coord(x, y, z, c0.z, c1.z, c2.z, c3.z), clip_coord(x, y, z, c0.w, c1.w, c2.w, c3.w));
    }
};
__device__ __forceinline__ ClipView clip_view(const float* pos, const float* verts, const float4* cmat, int b, int V)
{
    ClipView cv;
    cv.posv = (pos != nullptr) ? pos + (size_t)b * V * 4 : nullptr;
    cv.verts = verts;
    cv.cm = (pos == nullptr) ? cmat + (size_t)b * 4 : nullptr;
    return cv;
}

#ifdef FMHR_X_TS
// measurement build only: start / end timestamps (100 MHz) and the hardware slot of every wave of a kernel, one array per
// translation unit (FMHR_WTS_DEFINE), read back through fmhr_debug_wts_<unit>() by tools/analysis/wave_timeline.py
#define FMHR_WTS_DEFINE(unit)                                                                                          \
    __device__ unsigned long long g_wts_##unit[4 * 81920];                                                              \
    extern "C" int fmhr_debug_wts_##unit(void* out, size_t bytes)                                                     \
    {                                                                                                                  \
        hipError_t e = hipDeviceSynchronize();                                                                         \
        if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wts_##unit), bytes < sizeof(g_wts_##unit) ? bytes : sizeof(g_wts_##unit)); \
        return e == hipSuccess ? 0 : 1;                                                                                \
    }
#define FMHR_WTS_BEGIN(unit, wave_index)                                                                               \
    const long long wts_w_ = (wave_index);                                                                             \
    if ((threadIdx.x & 63) == 0 && wts_w_ < 81920) {                                                                   \
        unsigned hw_, xcc_;                                                                                            \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                              \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                            \
        g_wts_##unit[wts_w_ * 4 + 2] = ((unsigned long long)xcc_ << 32) | hw_;                                         \
        g_wts_##unit[wts_w_ * 4] = wall_clock64();                                                                     \
    }
#define FMHR_WTS_END(unit)                                                                                             \
    if ((threadIdx.x & 63) == 0 && wts_w_ < 81920) g_wts_##unit[wts_w_ * 4 + 1] = wall_clock64();
#else
#define FMHR_WTS_DEFINE(unit)
#define FMHR_WTS_BEGIN(unit, wave_index)
#define FMHR_WTS_END(unit)
#endif

// Scatter-add of 3 x 3 values per lane, TRANSPOSED: lane `on` adds val[3 k + m] to base[row[k] * STRIDE + (m == 2 ? C2 : m)] for k, m in
// 0..2 (a pixel's / face's three vertices x three components).  Issued lane = (item, k, m) from a per-wave LDS stage (sv [64 * 9], sx
// [64 * 3], this wave's slices), the three components of a vertex are ONE cache-line request: a posted atomic costs per distinct line per
// instruction — issued lane = item, one instruction per component, they were three.  Every lane of the wave must call.
template <int STRIDE, int C2>
__device__ __forceinline__ void wave_add_3x3(float* __restrict__ base, bool on, const float (&val)[9], const unsigned (&row)[3], float* sv,
                                             unsigned* sx)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long bal = __ballot(on);
    const int cnt = __popcll(bal);
    if (cnt == 0) return;                                            // wave-uniform
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (on) {
#pragma unroll
        for (int k = 0; k < 9; k++) sv[rank * 9 + k] = val[k];
#pragma unroll
        for (int k = 0; k < 3; k++) sx[rank * 3 + k] = row[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < cnt * 9; e += 64) {
        const float v = sv[e];
        if (v != 0.f) {
            const int kv = e / 3, m = e - kv * 3;
            atomicAdd(base + (size_t)sx[kv] * STRIDE + (m == 2 ? C2 : m), v);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the stage may be refilled by the caller's next round
    __builtin_amdgcn_wave_barrier();
}

// block (256 threads = 4 waves) sum; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sm4)
{
    v = wave_sum(v);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm4[wv] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = sm4[0] + sm4[1] + sm4[2] + sm4[3];
    __syncthreads();
    return r;
}

}  // namespace fmhr

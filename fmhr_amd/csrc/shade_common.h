// fmhr_amd/csrc/shade_common.h — per-pixel shading shared by the forward (fused into the rasterizer's resolve pass, raster.hip)
// and the backward (shade.hip): interpolate -> F.normalize -> SH radiance -> albedo product, /root/reference/mesh_sfs_optim.py:268-286.
#pragma once
#include "common.h"
#include <stdlib.h>

namespace fmhr {

// value-only kernels: hardware reciprocal / square root (1 ulp) instead of the correctly rounded expansions (~10 dependent
// instructions each) — the per-pixel kernels are bound by dependent ALU latency, and nothing discrete hangs on these values
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

__device__ __forceinline__ float sh9c(const float* c, float x, float y, float z)
{
    float r = c[0];
    r = r + c[1] * y;
    r = r + c[2] * z;
    r = r + c[3] * x;
    r = r + c[4] * x * y;
    r = r + c[5] * y * z;
    r = r + c[6] * (2.0f * z * z - x * x - y * y);
    r = r + c[7] * z * x;
    r = r + c[8] * (x * x - y * y);
    return r;
}

struct PixShade {
    float b0, b1, b2;        // barycentrics
    int i0, i1, i2;
    float nx, ny, nz, len;   // interpolated normal (raw) and its length
    float hx, hy, hz;        // normalised
    float ar, ag, ab;        // interpolated albedo
    float rad;
    float3 n0, n1, n2, a0, a1, a2;
};

__device__ __forceinline__ float3 ld3f(const float* p) { return make_float3(p[0], p[1], p[2]); }

// attr8 [V,2] float4 = (normal.xyz, albedo.x | albedo.yz, 0, 0): two 16-B gathers per vertex instead of six 4-B ones (the pixel
// kernels were bound by the number of load requests, not by bytes)
__device__ __forceinline__ void shade_eval(const float4 r, const int32_t* __restrict__ tri, const float4* __restrict__ attr8,
                                           const float* __restrict__ shc, PixShade& s)
{
    int t = (int)r.w - 1;
    s.i0 = tri[t * 3]; s.i1 = tri[t * 3 + 1]; s.i2 = tri[t * 3 + 2];
    s.b0 = r.x; s.b1 = r.y; s.b2 = 1.0f - r.x - r.y;
    {
        const float4 q0 = attr8[(size_t)s.i0 * 2], q1 = attr8[(size_t)s.i0 * 2 + 1];
        const float4 u0 = attr8[(size_t)s.i1 * 2], u1 = attr8[(size_t)s.i1 * 2 + 1];
        const float4 w0 = attr8[(size_t)s.i2 * 2], w1 = attr8[(size_t)s.i2 * 2 + 1];
        s.n0 = make_float3(q0.x, q0.y, q0.z); s.a0 = make_float3(q0.w, q1.x, q1.y);
        s.n1 = make_float3(u0.x, u0.y, u0.z); s.a1 = make_float3(u0.w, u1.x, u1.y);
        s.n2 = make_float3(w0.x, w0.y, w0.z); s.a2 = make_float3(w0.w, w1.x, w1.y);
    }
    s.nx = s.b0 * s.n0.x + s.b1 * s.n1.x + s.b2 * s.n2.x;
    s.ny = s.b0 * s.n0.y + s.b1 * s.n1.y + s.b2 * s.n2.y;
    s.nz = s.b0 * s.n0.z + s.b1 * s.n1.z + s.b2 * s.n2.z;
    s.ar = s.b0 * s.a0.x + s.b1 * s.a1.x + s.b2 * s.a2.x;
    s.ag = s.b0 * s.a0.y + s.b1 * s.a1.y + s.b2 * s.a2.y;
    s.ab = s.b0 * s.a0.z + s.b1 * s.a1.z + s.b2 * s.a2.z;
    s.len = fsqrt(s.nx * s.nx + s.ny * s.ny + s.nz * s.nz);
    float inv = frcp(fmaxf(s.len, 1e-12f));                  // F.normalize(p=2, dim=3), default eps (:273)
    s.hx = s.nx * inv; s.hy = s.ny * inv; s.hz = s.nz * inv;
    s.rad = sh9c(shc, s.hx, s.hy, s.hz);
}

// shade_eval with the six attr8 words already loaded
__device__ __forceinline__ void shade_eval_attr(float u, float v, const float4 q0, const float4 q1, const float4 u0, const float4 u1,
                                                const float4 w0, const float4 w1, const float* __restrict__ shc, PixShade& s)
{
    s.b0 = u; s.b1 = v; s.b2 = 1.0f - u - v;
    s.n0 = make_float3(q0.x, q0.y, q0.z); s.a0 = make_float3(q0.w, q1.x, q1.y);
    s.n1 = make_float3(u0.x, u0.y, u0.z); s.a1 = make_float3(u0.w, u1.x, u1.y);
    s.n2 = make_float3(w0.x, w0.y, w0.z); s.a2 = make_float3(w0.w, w1.x, w1.y);
    s.nx = s.b0 * s.n0.x + s.b1 * s.n1.x + s.b2 * s.n2.x;
    s.ny = s.b0 * s.n0.y + s.b1 * s.n1.y + s.b2 * s.n2.y;
    s.nz = s.b0 * s.n0.z + s.b1 * s.n1.z + s.b2 * s.n2.z;
    s.ar = s.b0 * s.a0.x + s.b1 * s.a1.x + s.b2 * s.a2.x;
    s.ag = s.b0 * s.a0.y + s.b1 * s.a1.y + s.b2 * s.a2.y;
    s.ab = s.b0 * s.a0.z + s.b1 * s.a1.z + s.b2 * s.a2.z;
    s.len = fsqrt(s.nx * s.nx + s.ny * s.ny + s.nz * s.nz);
    float inv = frcp(fmaxf(s.len, 1e-12f));                  // F.normalize(p=2, dim=3), default eps (:273)
    s.hx = s.nx * inv; s.hy = s.ny * inv; s.hz = s.nz * inv;
    s.rad = sh9c(shc, s.hx, s.hy, s.hz);
}

// The fused step keeps ONE 16-B word per covered pixel, (u, v, z/w, packed), read by the antialias kernels (z/w, id, silhouette
// bits) and by the face-parallel backward (u, v, owner, flags).  packed: bits 0..23 face id, 24 mask > 0, 25 large-box face,
// 26..28 silhouette bits of the face in this view (antialias rule A3).  fmhr_ham_densify rebuilds the dr.rasterize image from it.
__device__ __forceinline__ int pix_pack(int f, bool mask_pos, bool large, int silh) { return f | ((int)mask_pos << 24) | ((int)large << 25) | (silh << 26); }
__device__ __forceinline__ int pix_face(int p) { return p & 0xFFFFFF; }
__device__ __forceinline__ bool pix_mask(int p) { return (p >> 24) & 1; }
__device__ __forceinline__ bool pix_large(int p) { return (p >> 25) & 1; }
__device__ __forceinline__ int pix_silh(int p) { return (p >> 26) & 7; }

// rgba of a covered pixel from its per-pixel word: the resolve pass's forward shading, recomputed statement for statement (compile
// with -ffp-contract=off like raster.hip).  The step keeps no shaded image: the antialias forward re-shades the two pixels of the
// pairs that blend, fmhr_ham_densify re-shades everything for inspection.  shc = the view's 9 SH coefficients.
__device__ __forceinline__ float4 shade_word(const float4 w, const int32_t* __restrict__ tri, const float4* __restrict__ attr8,
                                             const float* __restrict__ shc_view)
{
    const int pk = __float_as_int(w.w), f = pix_face(pk);
    float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
    const float w2 = 1.0f - w.x - w.y;
    col.w = w.x + w.y + w2;
    if (pix_mask(pk)) {
        const int i0 = tri[f * 3], i1 = tri[f * 3 + 1], i2 = tri[f * 3 + 2];
        const float4 q0 = attr8[(size_t)i0 * 2], q1 = attr8[(size_t)i0 * 2 + 1], u0 = attr8[(size_t)i1 * 2], u1 = attr8[(size_t)i1 * 2 + 1],
                     w0 = attr8[(size_t)i2 * 2], w1 = attr8[(size_t)i2 * 2 + 1];
        float shc[9];
        {   // the view's SH row: 36 B, dword-aligned -> 2 x 16 B + 4 B requests instead of nine
            const F4U* sp = reinterpret_cast<const F4U*>(shc_view);
            const F4U s0 = sp[0], s1 = sp[1];
            shc[0] = s0.x; shc[1] = s0.y; shc[2] = s0.z; shc[3] = s0.w; shc[4] = s1.x; shc[5] = s1.y; shc[6] = s1.z; shc[7] = s1.w;
            shc[8] = shc_view[8];
        }
        PixShade s;
        shade_eval_attr(w.x, w.y, q0, q1, u0, u1, w0, w1, shc, s);
        col.x = s.rad * s.ar; col.y = s.rad * s.ag; col.z = s.rad * s.ab;
    }
    return col;
}

// entry of a fragment bin (raster.hip: binned rasterizer): 8 B = (face | large-box flag << 22 | silhouette bits << 23 | pixel-in-segment
// << 26, z/w bits); the second pass evaluates (u, v) for the winners (round 3 carried them: 16 B)
typedef uint2 BinEntry;

// what the fused stage-2 step adds to the rasterizer's passes (all null for the stand-alone dr.rasterize path)
struct RasterFused {
    int2* cov16; int* stamp; unsigned char* own8;             // coverage entries, step counter, owner marks (face-parallel backward)
    float4* pix16; const float* mask;                         // per covered pixel (u, v, z/w, packed id | flags): see pix_pack
    const unsigned char* silh;                                // silhouette bits per (view, face)
    const float4* attr8; const float* sh;                     // forward shading inputs
    unsigned long long* silw;                                 // per segment: which of its covered pixels belong to a face with a silhouette edge
    float4* aa_init; int* seg_valid;                          // forward shading output (the rgba the antialias pass starts from) +
                                                              // valid-pixel count of every segment
    int* rearm2;                                              // two list cursors the resolve pass zeroes for the next kernel
    int* rearm_aa;                                            // antialias work-item counter + overflow flag, likewise
    const float* verts; const float4* cmat;                   // non-null: clip positions / snapped vertices are recomputed from the
                                                              // [V,3] vertices and the views' combined matrices (common.h: ClipView),
                                                              // `pos` and `snap` are not read
    BinEntry* bins; int* bin_cursor; int* bin_flag; int bin_cap; // non-null: per-tile fragment bins (raster.hip: binned rasterizer)
    int2* bin_list; int* bin_list_cnt;                         // round 5: the step's compacted list of segments that hold fragments + its cursor
    UDiv segx;                                                // row pitch != W (ABI 5): division by the segments per row — the caller's
                                                              // `mask` is [n,H,W], the step's own per-pixel buffers are [n,H,pitch]
    ViewSel sel;                                              // ABI 6: the step's views are a subset of the caller's (common.h); sel.idx null = not
};

// ---- binned rasterizer (raster.hip): a bin = one 64-pixel row segment of the flattened [n, H, W] index = the wave that resolves it.
// Its entries are 8 B per fragment: (face | large-box flag << 22 | silhouette bits << 23 | pixel-in-segment << 26, z/w bits); 0xFFFFFFFF
// in the first word = a reserved slot whose fragment turned out degenerate.
constexpr int kBinCapDefault = 512;                           // 8 fragments per pixel of a 64-pixel segment on average.  The fullest segment measured: 336-448 at the
                                                              // metric over a 600-step run, 366 at the demo conf's 1280 x 1024 (bench.py: bin_depth_max); until
                                                              // round 5 the default was 1024 = 1.6 GB of reserved span at the metric for that depth
inline int bin_cap_entries(int requested)                     // entries per bin: fmhr_ham_dims.bin_cap, 0 = kBinCapDefault
{
    return requested <= 0 ? kBinCapDefault : (requested < 64 ? 64 : requested);
}
inline bool raster_binned_wanted(bool dflt)                   // FMHR_RASTER_BINNED=0 / 1 overrides the default (same-library A/B; read per call)
{
    const char* e = getenv("FMHR_RASTER_BINNED");
    return e ? atoi(e) != 0 : dflt;
}
#ifndef FMHR_BIN_CHUNK
#define FMHR_BIN_CHUNK 128
#endif
constexpr int kBinChunk = FMHR_BIN_CHUNK;                     // entries a wave stages in LDS per round (8-bit entry index in the LDS key);
                                                              // 64 / 128 / 256: k_raster_bins 83.0 / 80.2 / 83.3 us (same-box A/B)
#ifndef FMHR_CURSOR_STRIDE
#define FMHR_CURSOR_STRIDE 16
#endif
constexpr int kCursorStride = FMHR_CURSOR_STRIDE;            // ints between two bins' cursors: ONE cursor per 64-B line — atomics on one line are
                                                              // served one after the other at the memory side (~55 ns each), and with 16 neighbouring
                                                              // bins in a line every reservation of an image region queued behind the others
                                                              // (k_raster_tris 277 us with packed cursors, 124 us padded; profiles/r03_binned_ab.txt)
// W: image width (what makes a mesh a micro-triangle one), Wp: row pitch of the step's per-pixel buffers (what a bin is cut from)
inline bool raster_binned_ok(int F, int H, int W, int Wp) { return (Wp % 64) == 0 && F < (1 << 22) && !raster_use_queue(H, W, F); }
inline size_t raster_bin_count(int n, int H, int Wp) { return ((size_t)n * H * Wp) / 64; }

}  // namespace fmhr

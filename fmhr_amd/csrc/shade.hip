// fmhr_amd/csrc/shade.hip — fused per-pixel / per-vertex kernels of the stage-2 HAM step for gfx950.
// One pass replaces the reference's interpolate(C=7) -> slices -> F.normalize -> boolean gather -> get_radiance ->
// multiply -> scatter chain (/root/reference/mesh_sfs_optim.py:268-286), one pass the two loss reductions (:289,:295).
// The backward is FACE-parallel: the lanes of a (face, view group) take the candidate pixels of the face's coverage masks (left by the
// rasterizer), do the whole per-pixel backward in registers and store per-corner sums that the mesh tail adds up per vertex — no float
// atomics on that path (the scatter form ran at the 0.08 TB/s "64 lanes -> 64 rows" atomic shape: 3.6 ms per step).
#include "common.h"
#include "shade_common.h"
#include <stdlib.h>

namespace fmhr {

// ------------------------------------------------------------------------------------------------ segment sparsity
// The flattened pixel index [n*H*W] is cut into 64-pixel segments = exactly one wave of every per-pixel kernel.  A segment is
// ACTIVE when it holds a covered pixel, a 4-neighbour of one (antialias pairs and blend targets), or a non-zero valid_mask
// (silhouette MSE).  Per-pixel kernels return after one scalar load on inactive segments (~2/3 of a 19 %-coverage frame);
// every buffer is fully (re)written inside active segments each step and never read outside them.
__device__ __forceinline__ unsigned long long covw_at(const unsigned long long* __restrict__ covw, long long w, long long nSeg)
{
    return (w >= 0 && w < nSeg) ? covw[w] : 0ull;
}

// per-segment "valid_mask has a non-zero" flags: recomputed only when the caller says the mask changed (it is constant over a
// HAM run: the AA silhouette of the initial mesh, mesh_sfs_optim.py:158)
__global__ void __launch_bounds__(kBlock)
k_valid_mask_segments(const float* __restrict__ valid_mask, long long N, int W, int Wp, const UDiv segx, unsigned char* __restrict__ vmnz)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid - (gid & 63) >= N) return;
    // N counts the step's own pixels [n,H,Wp]; the caller's mask is [n,H,W] (common.h: row pitch)
    long long cidx = gid;
    bool incol = true;
    if (Wp != W) cidx = caller_index((unsigned)__builtin_amdgcn_readfirstlane((int)(gid >> 6)), (int)(threadIdx.x & 63), W, segx, incol);
    float vm = (gid < N && incol) ? valid_mask[cidx] : 0.f;
    unsigned long long bal = __ballot(vm != 0.f);
    if ((threadIdx.x & 63) == 0) vmnz[gid >> 6] = bal != 0ull ? 1 : 0;
}

// active = own coverage | 4-neighbour coverage | valid_mask flag (seven independent word loads)
__device__ __forceinline__ bool segment_is_active(const unsigned long long* __restrict__ covw, const unsigned char* __restrict__ vmnz,
                                                  long long nSeg, int W, long long seg, unsigned char* __restrict__ act, const ViewSel& sel)
{
    // the valid_mask flags are kept per SOURCE view when the step's views are a selection of the capture's (ABI 6): lane = segment here
    long long vseg = seg;
    if (sel.idx != nullptr) {
        const unsigned b = udiv((unsigned)seg, sel.segv);
        vseg = seg + (long long)(sel.idx[b] - (int)b) * sel.segv.d;
    }
    long long s0 = seg << 6, up = s0 - W, dn = s0 + W;
    long long wu = up >> 6, wd = dn >> 6;               // arithmetic shift = floor, also for negative `up`
    int su = (int)(up & 63), sd = (int)(dn & 63);
    unsigned long long c0 = covw[seg], cl = covw_at(covw, seg - 1, nSeg), cr = covw_at(covw, seg + 1, nSeg);
    unsigned long long u0 = covw_at(covw, wu, nSeg), u1 = covw_at(covw, wu + 1, nSeg);
    unsigned long long d0 = covw_at(covw, wd, nSeg), d1 = covw_at(covw, wd + 1, nSeg);
    unsigned long long any = (unsigned long long)vmnz[vseg] | c0 | (cl >> 63) | (cr & 1ull);
    any |= (u0 >> su) | (su ? (u1 << (64 - su)) : 0ull);   // covered pixels among [s0-W, s0-W+63]
    any |= (d0 >> sd) | (sd ? (d1 << (64 - sd)) : 0ull);   // and among [s0+W, s0+W+63]
    act[seg] = any != 0ull ? 1 : 0;
    return any != 0ull;
}

// single-workgroup deterministic sums of per-block partials
__device__ __forceinline__ float block1024_sum(float v, float* sm16)
{
    v = wave_sum(v);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm16[wv] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < 16; i++) r += sm16[i];
    __syncthreads();
    return r;
}

// one lane per segment; the active ones are also compacted into act_list (order arbitrary) so that the per-pixel kernels
// run as persistent loops over exactly the active segments: 2/3 of a frame's 197k waves would only be launched to exit
constexpr int kActBlock = 1024;   // few, fat workgroups: the list cursors are same-address atomics (~10 ns each, serialised)

__global__ void __launch_bounds__(kActBlock)
k_active_segments(const unsigned long long* __restrict__ covw, const unsigned char* __restrict__ vmnz, long long nSeg, int W,
                  unsigned char* __restrict__ act, int* __restrict__ act_list, int* __restrict__ act_count,
                  const int* __restrict__ seg_valid, float* __restrict__ cnt_part, long long N, float4* __restrict__ aa_init, const ViewSel sel)
{
    // act_count[0] = number of active segments (act_list): the cursor advances with ONE atomic per workgroup
    constexpr int kWaves = kActBlock / 64;
    __shared__ int wcnt[kWaves];
    __shared__ int base;
    __shared__ float sm16[16];
    long long seg = (long long)blockIdx.x * kActBlock + threadIdx.x;
    {   // valid-pixel count: the resolve pass left one number per segment (it has no workgroup barrier: clean waves leave at
        // once); summed per workgroup here, the next kernel adds the ~200 partials up
        float v = (seg < nSeg && seg_valid != nullptr) ? (float)(seg_valid[seg] & 0xff) : 0.f;
        float sv = block1024_sum(v, sm16);
        if (threadIdx.x == 0 && cnt_part != nullptr) cnt_part[blockIdx.x] = sv;
    }
    bool a = (seg < nSeg) && segment_is_active(covw, vmnz, nSeg, W, seg, act, sel);
    // bit 31 of the list entry: the antialias scan has something to find here — a pixel of this segment, of the next one (lane 63's
    // right neighbour) or of the segments below holds a face with a silhouette edge (bit 8 of the resolve pass's per-segment
    // word); the nearer pixel of a pair that blends is always such a pixel, so every other active segment is skipped by the scan.
    // (Requested HERE: behind the cursor atomic below these four loads were a round trip of their own at the end of the kernel.)
    unsigned flag = 0u;
    if (a && seg_valid != nullptr) {
        const long long dn = ((seg << 6) + W) >> 6;
        int sv = seg_valid[seg];
        if (seg + 1 < nSeg) sv |= seg_valid[seg + 1];
        if (dn < nSeg) sv |= seg_valid[dn];
        if (dn + 1 < nSeg) sv |= seg_valid[dn + 1];
        flag = (sv & 0x100) ? 0x80000000u : 0u;
    }
    bool cv = a && covw[seg] == 0ull;                                // active without coverage: needs its zeros
    unsigned long long bal = __ballot(a), cbal = __ballot(cv);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wcnt[wv] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int na = 0;
        for (int i = 0; i < kWaves; i++) na += wcnt[i];
        base = atomicAdd(act_count, na);
    }
    __syncthreads();
    if (a) {
        int off = base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int i = 0; i < wv; i++) off += wcnt[i];
        act_list[off] = (int)((unsigned)seg | flag);
    }
    // The forward shading is fused into the rasterizer's resolve pass (raster.hip), which writes the rgba for every pixel of
    // a segment that received fragments.  Active segments WITHOUT coverage (neighbours of covered ones, valid_mask-only ones) get
    // their zeros here, wave-cooperatively: one 64-pixel segment per iteration, 1 KB per store instruction
    if (aa_init != nullptr) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const long long wseg = seg - lane;                           // the wave's lanes hold consecutive segments
        while (cbal != 0ull) {
            const int src = __builtin_ctzll(cbal);
            cbal &= cbal - 1ull;
            const long long gid = ((wseg + src) << 6) + lane;
            if (gid < N) aa_init[gid] = z;
        }
    }
}

__device__ __forceinline__ bool seg_active(const unsigned char* __restrict__ act, long long gid)
{
    return __builtin_amdgcn_readfirstlane((int)act[gid >> 6]) != 0;
}

#ifndef FMHR_PERSIST_BLOCKS
#define FMHR_PERSIST_BLOCKS 2048
#endif
constexpr int kPersistBlocks = FMHR_PERSIST_BLOCKS;      // persistent per-pixel kernels: 8 workgroups per CU, waves loop over act_list

// test/inspection helper: make the sparse buffers dense (zeros wherever the step did not write)
__global__ void __launch_bounds__(kBlock)
k_densify(const unsigned long long* __restrict__ covw, const unsigned char* __restrict__ act, long long N,
          const float4* __restrict__ pix16, float4* __restrict__ rast, float4* __restrict__ aa)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= N) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    bool covered = (covw[gid >> 6] >> (gid & 63)) & 1ull;
    float4 r = z;
    if (covered) {                                   // the dr.rasterize image (u, v, z/w, id + 1) from the step's per-pixel word
        r = pix16[gid];
        r.w = (float)(pix_face(__float_as_int(r.w)) + 1);
    }
    rast[gid] = r;
    if (!act[gid >> 6]) aa[gid] = z;
}

// losses + their per-pixel gradients over the active segments:
// sfs = w * mean_valid |aa.rgb - img| (:289), mask = w * mean (aa.a - valid_mask)^2 (:295) — inactive segments contribute
// exactly 0 to both.  g_shaded [.,4] = d loss / d antialiased rgba: the antialias backward adds its colour corrections to it,
// the face-parallel backward consumes it.
__global__ void __launch_bounds__(kBlock)
k_loss_pix(const unsigned long long* __restrict__ covw, const float4* __restrict__ aa, const float* __restrict__ img,
           const float* __restrict__ mask, const float* __restrict__ valid_mask, long long N, int W, int Wp, const UDiv segx, const ViewSel sel,
           const int* __restrict__ act_list, const int* __restrict__ act_count, const float* __restrict__ n_valid_global,
           float w_sfs, float w_mask, float mse_den, float4* __restrict__ g_shaded,
           float* __restrict__ part_l1, float* __restrict__ part_l2, float* __restrict__ losses)
{
    __shared__ float sm[4];
    const int count = *act_count, lane = threadIdx.x & 63;
    float l1 = 0.f, l2 = 0.f;
    float den = 3.0f * n_valid_global[0];
    float k1 = (den > 0.f) ? w_sfs / den : 0.f;
    float k2 = w_mask / mse_den;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < count; i += gridDim.x * 4) {
        const long long seg = act_list[i] & 0x7fffffff;            // (bit 31 is the antialias scan's flag)
        const long long gid = (seg << 6) + lane;
        if (gid >= N) continue;
        // img / mask / valid_mask are the caller's [n,H,W] tensors; aa, g_shaded and the coverage words the step's own [n,H,Wp]
        long long cidx = gid;
        bool incol = true;
        if (sel.idx != nullptr) cidx = caller_index_sel((unsigned)__builtin_amdgcn_readfirstlane((int)seg), lane, W, segx, sel, incol);   // (ABI 6: a minibatch of the capture's views)
        else if (Wp != W) cidx = caller_index((unsigned)__builtin_amdgcn_readfirstlane((int)seg), lane, W, segx, incol);
        if (!incol) continue;                                       // a pad column: no pixel, no loss term
        // every request of the iteration up front: aa (16 B), the image pixel (three 4-B streaming loads: see below), the two masks
        // (streaming loads, common.h: this pass reads 160 MB exactly once — through the default policy it leaves the L2 without the
        // tables the antialias and face backward and the mesh tails gather from; 3 029 -> 3 060 it/s over five alternating rounds, the
        // tails 26.3 -> 25.0 us; the image pixel as three 4-B requests: a 12-B streaming load does not exist)
        float4 a = ld_stream(&aa[gid]);
        F3U im;
        im.x = ld_stream(img + cidx * 3); im.y = ld_stream(img + cidx * 3 + 1); im.z = ld_stream(img + cidx * 3 + 2);
        const float mk = ld_stream(mask + cidx), vm = ld_stream(valid_mask + cidx);
        const bool cov = (covw[seg] >> lane) & 1ull;          // covered pixel (the segment's coverage word)
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mk > 0.f && cov) {
            float d0 = a.x - im.x, d1 = a.y - im.y, d2 = a.z - im.z;
            l1 += fabsf(d0) + fabsf(d1) + fabsf(d2);
            g.x = (d0 > 0.f) ? k1 : ((d0 < 0.f) ? -k1 : 0.f);
            g.y = (d1 > 0.f) ? k1 : ((d1 < 0.f) ? -k1 : 0.f);
            g.z = (d2 > 0.f) ? k1 : ((d2 < 0.f) ? -k1 : 0.f);
        }
        float dm = a.w - vm;
        l2 += dm * dm;
        g.w = 2.0f * dm * k2;
        if (cov) g_shaded[gid] = g;   // read by the owner face only (the antialias backward recomputes the gradients it needs)
    }
    float s1 = block_sum(l1, sm);
    float s2 = block_sum(l2, sm);
    if (threadIdx.x == 0) { part_l1[blockIdx.x] = s1; part_l2[blockIdx.x] = s2; }
    if (blockIdx.x == 0 && threadIdx.x < 8) losses[threadIdx.x] = 0.f;      // the regulariser kernels accumulate into these later
}

constexpr int kGroup = 8;  // views summed inside one lane group of the face-parallel backward

// ------------------------------------------------------------------------------------------------ face-parallel backward
// One lane per (face, view); the kGroup lanes of a group are kGroup consecutive views of ONE face and end in one DPP sum, so any
// lane of a group may process any pixel of that face in any of the group's views.  The rasterizer left a coverage entry per
// (view, face): the candidate-pixel mask of the triangle's box (raster.hip: k_raster_tris).  A group expands the masks of its views
// into an LDS task list and its lanes take the tasks round-robin: load the pixel's (u, v, owner, flags) word and image gradient,
// skip it unless this face owns the pixel (hidden candidates), else do the whole per-pixel backward (shade -> normalise ->
// interpolate -> rasterize -> clip transform) in registers, accumulating the three corners' gradients.  Per-face data (3 x
// attr8 as corner-2-relative differences) and per-(face, view) data (3 clip positions, box) are loaded once and parked in LDS;
// the view's SH row and clip matrix come from an LDS copy.  Finally lanes 0..2 of a group store corner 0..2 as 3 x float4 into
// the corner's slot of its vertex's CSR row (corner_slot = inverse of vf_idx), where the mesh tail sums a vertex's slots with
// coalesced loads.  No per-pixel record buffer, no pixel lists for box triangles, no float atomics, deterministic.
// (Large-box triangles have no mask: the large-box queue's workers scan their boxes, face_bwd_large_items.)
// The budget of these kernels is the NUMBER of vector-memory requests (a scattered wave-load occupies the texture-address unit
// for up to 64 cycles) and the number of wave iterations of the ~250-instruction pixel backward.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum(float x)
{
    x += dpp_mov<0xB1>(x);      // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);      // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);     // row_half_mirror: lane i <-> 7 - i within each 8 lanes
    return x;
}

struct int3a { int x, y, z; };

struct FaceAttr { float3 n2, dn0, dn1, a2, da0, da1, v0, v1, v2; };    // corner-2-relative attributes + the three vertex positions
static_assert(sizeof(FaceAttr) == 27 * sizeof(float), "k_face_bwd fills the record float by float");

// backward of ONE pixel of a face for all three corners; A[k] += (g_normal.xyz, g_albedo.xyz, g_vertex.xyz) of corner k.
// c = the view's 9 SH coefficients followed by its clip matrix rows (view_c layout).
__device__ __forceinline__ void pixel_backward(float (&A)[3][9], const FaceAttr& fa_lds,
                                               const float* __restrict__ c, float u, float v, const float4 gs, int px, int py,
                                               const PixGrid& pg)
{
    const FaceAttr fa = fa_lds;                 // read at the use: the kernel's occupancy is set by its live registers
    const float b0 = u, b1 = v, b2 = 1.0f - u - v;
    // forward recompute (shade_eval, with the corner-2-relative attribute differences)
    const float nxr = fa.n2.x + b0 * fa.dn0.x + b1 * fa.dn1.x, nyr = fa.n2.y + b0 * fa.dn0.y + b1 * fa.dn1.y,
                nzr = fa.n2.z + b0 * fa.dn0.z + b1 * fa.dn1.z;
    const float ar = fa.a2.x + b0 * fa.da0.x + b1 * fa.da1.x, ag = fa.a2.y + b0 * fa.da0.y + b1 * fa.da1.y,
                ab = fa.a2.z + b0 * fa.da0.z + b1 * fa.da1.z;
    const float len = fsqrt(nxr * nxr + nyr * nyr + nzr * nzr);
    const float inv = frcp(fmaxf(len, 1e-12f));
    const float x = nxr * inv, y = nyr * inv, z = nzr * inv;
    const float rad = sh9c(c, x, y, z);
    // shade backward (mesh_sfs_optim.py:284-286, models/utils.py:208-226)
    const float g_rad = gs.x * ar + gs.y * ag + gs.z * ab;
    const float ga0 = gs.x * rad, ga1 = gs.y * rad, ga2 = gs.z * rad;
    const float dx = c[3] + c[4] * y - 2.0f * c[6] * x + c[7] * z + 2.0f * c[8] * x;
    const float dy = c[1] + c[4] * x + c[5] * z - 2.0f * c[6] * y - 2.0f * c[8] * y;
    const float dz = c[2] + c[5] * y + 4.0f * c[6] * z + c[7] * x;
    const float ghx = g_rad * dx, ghy = g_rad * dy, ghz = g_rad * dz;
    float gnx, gny, gnz;
    if (len > 1e-12f) {
        const float dt = x * ghx + y * ghy + z * ghz;
        gnx = (ghx - x * dt) * inv; gny = (ghy - y * dt) * inv; gnz = (ghz - z * dt) * inv;
    } else {
        gnx = ghx * 1e12f; gny = ghy * 1e12f; gnz = ghz * 1e12f;
    }
    // interpolate backward w.r.t. the barycentrics
    const float gu = gnx * fa.dn0.x + gny * fa.dn0.y + gnz * fa.dn0.z + ga0 * fa.da0.x + ga1 * fa.da0.y + ga2 * fa.da0.z;
    const float gv = gnx * fa.dn1.x + gny * fa.dn1.y + gnz * fa.dn1.z + ga0 * fa.da1.x + ga1 * fa.da1.y + ga2 * fa.da1.z;
    // rasterize backward (Appendix B.1) for the three corners, pushed through the view's clip transform (c[9..]: rows of
    // w2c @ proj, mesh.hip transform_body): the position gradients become view-independent
    // the three clip positions (x, y, w) of this view, recomputed from the face's vertices and the view's combined matrix
    // (c[9..24]: rows of w2c @ proj, columns x, y, w) exactly as the geometry kernel computed them: 27 FMAs instead of three
    // 16-B gathers per (face, view)
    float pp[9];
    {
        const float3 vv[3] = {fa.v0, fa.v1, fa.v2};
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int l = 0; l < 3; l++) pp[k * 3 + l] = clip_coord(vv[k].x, vv[k].y, vv[k].z, c[9 + l], c[13 + l], c[17 + l], c[21 + l]);
    }
    // rule R5's values (round 5, common.h): the gradient of (u, v) with respect to the clip positions in double — a thin triangle
    // divides the last bits of the pixel-relative edge functions by its area (capture cameras: 9.9e-4 of the gradient's maximum in fp32)
    float qx[3], qy[3], qw[3];
    r5_bwd_wide(pp[0], pp[1], pp[2], pp[3], pp[4], pp[5], pp[6], pp[7], pp[8], px, py, pg, gu, gv, qx, qy, qw);
    const float bk[3] = {b0, b1, b2};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        A[k][0] += bk[k] * gnx; A[k][1] += bk[k] * gny; A[k][2] += bk[k] * gnz;
        A[k][3] += bk[k] * ga0; A[k][4] += bk[k] * ga1; A[k][5] += bk[k] * ga2;
        A[k][6] += qx[k] * c[9] + qy[k] * c[10] + qw[k] * c[11];
        A[k][7] += qx[k] * c[13] + qy[k] * c[14] + qw[k] * c[15];
        A[k][8] += qx[k] * c[17] + qy[k] * c[18] + qw[k] * c[19];
    }
}

FMHR_WTS_DEFINE(shade)
constexpr int kTaskG = 64;                       // candidate pixels a group expands per round
#ifndef FMHR_FB_BLOCK
#define FMHR_FB_BLOCK 256
#endif
constexpr int kFbBlock = FMHR_FB_BLOCK;          // threads per workgroup of k_face_bwd
constexpr int kFacesPerBlock = kFbBlock / kGroup;
// 48 views: 6 views per lane (one 48-view group) ran 75 .. 135 us from PROCESS to process on one box — a wave then gathers from 48
// views' pixel arrays at once, 4 MB apart each — 4 views per lane (a 32-view and a 16-view group, the launch ordered group by group)
// 72 .. 88 us: 2 497 [2 397 .. 2 651] -> 2 651 [2 626 .. 2 676] it/s over 8 alternating runs each; 3 (24 + 24): 2 575; 2: 2 553; 5: 2 564
#ifndef FMHR_MAX_VPL
#define FMHR_MAX_VPL 6
#endif
constexpr int kMaxVPL = FMHR_MAX_VPL;                       // views per lane: a group of kGroup lanes covers kGroup * VPL views of one face

// VPL views per lane: the fixed part of a (face, view group) — vertex ids, attributes, output slots, the DPP sum, the output
// store, the launch — is paid once per kGroup * VPL views, and the partial planes the mesh tail has to add up shrink with it.
template <int VPL>
__device__ __forceinline__ void
face_bwd_body(const float4* __restrict__ bwd16, const int32_t* __restrict__ tri, const float* __restrict__ pos,
           const float4* __restrict__ attr8, const float* __restrict__ sh, const float4* __restrict__ g_shaded,
           const int2* __restrict__ cov16, const unsigned char* __restrict__ own8, const int* __restrict__ stamp,
           const int32_t* __restrict__ corner_slot,
           const float4* __restrict__ cmat, const float* __restrict__ verts, int n, int V, int F,
           int H, int W, float4* __restrict__ part, const PixGrid pg, const int g, const int b0, const int fb0)
{
    // (W here is the ROW PITCH of the per-pixel arrays: this body only addresses with it; the pixel grid comes in pg)
    FMHR_WTS_BEGIN(shade, ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (kFbBlock / 64) + (threadIdx.x >> 6));
    static_assert(kGroup == 8, "group_sum / the group scans cover 8 lanes");
    constexpr int GV = kGroup * VPL;                               // views per group
    __shared__ float view_c[GV][28];                               // per view of the group: 9 SH coefficients, 4 x 4 clip-matrix entries
    __shared__ int pk_box[kFbBlock * VPL];                           // box: pxa | pya << 13 | (width - 1) << 26
    __shared__ FaceAttr pk_attr[kFacesPerBlock];
    __shared__ unsigned short tasks[kFacesPerBlock][kTaskG];
    const long long P = (long long)H * W;
    const int tid = threadIdx.x, j = tid & (kGroup - 1), gl = tid / kGroup, lane = tid & 63;
    const int f = ((int)blockIdx.x - fb0) * kFacesPerBlock + gl;          // (fb0: the row's first face workgroup)
    // view of (lane j, slot v) = b0 + v * kGroup + j
    const bool facelive = f < F;                                   // uniform over a group
    // ---- the coverage entries of this lane's (face, view)s; entries of other steps carry another stamp
    const unsigned cur = (unsigned)stamp[0] & 0xFFFFFFu;           // advanced once per step by the geometry kernel
    // second word of a coverage entry (raster.hip): bits 26..31 = box width - 1; 63 = the entry of a large-box face, which has no
    // candidate mask: its pixels are the business of the large-box queue (face_bwd_large_items), not of this face's lanes
    auto entry_kind = [&](int y) -> int {
        const unsigned fld = (unsigned)y >> 26;
        return (fld == 63u) ? 0 : (int)fld + 1;
    };
    // everything whose address is known up front is requested together (the kernel's time is #waves / wave slots x the length of
    // a wave's dependent-load chain): the owner marks, the entries, the face's vertex ids, the output slot
    int3a ti = {0, 0, 0};
    unsigned es = 0u;                                              // (unsigned: as an int its sign extension sat right behind the load, with a wait)
    if (facelive) {
        ti = *reinterpret_cast<const int3a*>(tri + (size_t)f * 3);
        if (j < 3) es = (unsigned)corner_slot[f * 3 + j];
    }
    // kind: 0 nothing (also: a large-box face), 1..32 box width of a candidate mask.  Triangles that own no pixel this step (off screen, no
    // pixel centre, hidden: 2/3 of all) are dropped by the owner mark of the resolve pass; a stale mark only costs candidate checks.
    // The entries are needed once — box + candidate mask go into LDS task lists right away — and again only on one rare path
    // (a face with more than kTaskG candidates), which LOADS THEM AGAIN: kept alive through the pixel
    // loop their 36 registers were the difference between 3 and 4 waves per SIMD.
    auto load_entries = [&](int ff, int2 (&e)[VPL], int (&kind)[VPL]) {
#pragma unroll
        for (int v = 0; v < VPL; v++) {
            const int b = b0 + v * kGroup + j;
            e[v] = make_int2(0, -1);
            unsigned char mark = 0;
            const bool live = facelive && b < n;
            if (live) {
                mark = own8[(size_t)b * F + ff];
                e[v] = cov16[(size_t)b * F + ff];
            }
            kind[v] = (live && mark == (unsigned char)cur) ? entry_kind(e[v].y) : 0;
        }
    };
    // first use: every request up front (owner marks and entries of all the lane's views together with the vertex ids above)
    int2 e[VPL];
    unsigned char mark[VPL];
#pragma unroll
    for (int v = 0; v < VPL; v++) {
        const int b = b0 + v * kGroup + j;
        e[v] = make_int2(0, -1);
        mark[v] = 0;
        if (facelive && b < n) {
            mark[v] = own8[(size_t)b * F + f];
            e[v] = cov16[(size_t)b * F + f];
        }
    }
    // the per-view constants are staged while those requests are in flight (filled first, the fill's own round trip and the
    // barrier came before the first entry request of every wave)
    // (one unconditional load per element from a clamped address, all of a thread's loads issued before its first LDS store: as a
    // plain loop with the source picked by branches every one of its six iterations waited for its own load)
    {
        constexpr int IT = (GV * 28 + kFbBlock - 1) / kFbBlock;
        const float* const cmf = reinterpret_cast<const float*>(cmat);
        float vals[IT];
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = min(tid + it * kFbBlock, GV * 28 - 1);
            const int vb = i / 28, k = i - vb * 28, bb = min(b0 + vb, n - 1);
            // columns (x, y, w, z) of the view's combined matrix rows: the kernel only uses the first three of each row
            const int kk = min(max(k - 9, 0), 15), col = kk & 3;
            const int pc = (col == 2) ? 3 : ((col == 3) ? 2 : col);
            const float* src = (k < 9) ? (sh + (size_t)bb * 9 + k) : (cmf + (size_t)bb * 16 + (kk & ~3) + pc);
            const float v = *src;
            vals[it] = (b0 + vb < n && k < 25) ? v : 0.f;
        }
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int i = tid + it * kFbBlock;
            if (i < GV * 28) (&view_c[0][0])[i] = vals[it];
        }
    }
    __syncthreads();                                               // the only workgroup barrier: everything below is per wave
    int kind[VPL], cnt = 0;
    bool anyv = false;
#pragma unroll
    for (int v = 0; v < VPL; v++) {
        const bool live = facelive && (b0 + v * kGroup + j) < n;
        kind[v] = (live && mark[v] == (unsigned char)cur) ? entry_kind(e[v].y) : 0;
        if (kind[v] >= 1 && kind[v] <= 64) cnt += __popc((unsigned)e[v].x);       // (boxes hold at most 32 pixels: .y is 0)
        anyv = anyv || kind[v] != 0;
    }
    int incl = cnt;
    {   // scan over the 8 lanes of a group by DPP row shifts (a group lies inside one row of 16: the guard drops what a shift
        // pulls in from the neighbouring group), the total from the group's last lane by a quad / half-row broadcast
        int t;
        t = dpp_int<0x111>(incl); if (j >= 1) incl += t;
        t = dpp_int<0x112>(incl); if (j >= 2) incl += t;
        t = dpp_int<0x114>(incl); if (j >= 4) incl += t;
    }
    // lane 7 of the group: row_half_mirror sends lane 7 -> lane 0 of each 8; then quad_perm broadcast... simpler: the group's last
    // lane holds the total, every lane reads it through one ds_bpermute-free DPP pair (half-mirror, then lane 0 of the half row)
    const int excl = incl - cnt;
    int T;
    {
        const int mir = dpp_int<0x141>(incl);                        // row_half_mirror: lane j <- lane 7 - j, so lane 0 of the group holds the total
        const int q = dpp_int<0x00>(mir);                            // quad_perm [0,0,0,0]: lanes 0..3 <- lane 0, lanes 4..7 <- lane 4
        const int h = dpp_int<0x114>(q);                             // row_shr:4: lanes 4..7 <- lanes 0..3 (of the same group: j >= 4)
        T = (j < 4) ? q : h;
    }
    const unsigned long long anyk = __ballot(anyv);
    const bool group_any = ((anyk >> (lane & ~(kGroup - 1))) & 0xFFull) != 0ull;
    float A[3][9];                                                 // per corner: g_normal.xyz, g_albedo.xyz, g_vertex.xyz
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int c = 0; c < 9; c++) A[k][c] = 0.f;
    if (anyk != 0ull) {                                            // wave-uniform
        // ---- one memory round trip for the face's attributes (lane 0 of the group) AND the first candidate pixel of every lane:
        //      the attribute loads are issued first, the candidate masks (already here) are expanded into the task list, each lane
        //      requests its first task's pixel word and image gradient, and only then are the attributes parked
        // the face's nine 12- / 16-B pieces are requested by the group's lanes, one or two each: lane j = 2 c + h holds half h of
        // corner c's packed attributes, lanes 0..2 also a vertex position (all nine in lane 0 were 33 registers of every lane at the
        // kernel's tightest point: the spill reloads behind them made the wave wait for this round trip before it expanded its tasks)
        // (unconditional loads — lanes 6 / 7 and groups without an owner repeat a valid piece, a dead face reads vertex 0: under a
        // condition the merge with a zero default put register copies, and with them a wait, right behind the load)
        const int cj = j >> 1;
        const float4 piece = attr8[(size_t)(cj == 0 ? ti.x : (cj == 1 ? ti.y : ti.z)) * 2 + (j & 1)];
        const float3 vpos = ld3f(verts + (size_t)(j == 0 ? ti.x : (j == 1 ? ti.y : ti.z)) * 3);
#pragma unroll
        for (int v = 0; v < VPL; v++)
            if (kind[v] != 0) {
                const int at = tid * VPL + v;
                pk_box[at] = e[v].y;                                 // the entry's second word IS the packed box
            }
        // candidate pixels of the group's box triangles, kTaskG per round; task = (view slot v * kGroup + j) << 6 | bit
        auto expand = [&](int R, const int2 (&e)[VPL], const int (&kind)[VPL]) {
            if (cnt != 0 && excl < R + kTaskG && excl + cnt > R) {
                int idx = excl - R;
#pragma unroll
                for (int v = 0; v < VPL; v++) {
                    if (!(kind[v] >= 1 && kind[v] <= 64)) continue;
                    unsigned m = (unsigned)e[v].x;
                    while (m != 0u) {
                        const int bit = __builtin_ctz(m);
                        m &= m - 1u;
                        if (idx >= 0 && idx < kTaskG) tasks[gl][idx] = (unsigned short)(((v * kGroup + j) << 6) | bit);
                        idx++;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        auto locate = [&](int t, int& vs, int& px, int& py) {       // task -> view slot, pixel, flat pixel index
            const int en = tasks[gl][t], bit = en & 63;
            vs = en >> 6;
            const int src = ((tid & ~(kGroup - 1)) + (vs & (kGroup - 1))) * VPL + (vs >> 3);
            const int box = pk_box[src];
            // row / column of candidate `bit` in the box: bit < 64, 1 <= width <= 64: (bit + 0.5) / width is at least 1 / 128 away from an
            // integer, so the 1-ulp reciprocal cannot change the truncated quotient (the exact integer division costs ~25 instructions)
            const int bwd = ((box >> 26) & 63) + 1;
            const int r = (int)(((float)bit + 0.5f) * frcp((float)bwd)), c = bit - r * bwd;
            px = (box & 0x1fff) + c; py = ((box >> 13) & 0x1fff) + r;
            return (long long)(b0 + vs) * P + (long long)py * W + px;
        };
        expand(0, e, kind);
        const int nt0 = min(kTaskG, T);
        int pvs = 0, ppx = 0, ppy = 0;
        float4 pw = make_float4(0.f, 0.f, 0.f, 0.f), pgs = pw;
        if (j < nt0) {
            const long long gid = locate(j, pvs, ppx, ppy);
            pw = bwd16[gid];
            pgs = g_shaded[gid];
        }
        {
            float4 pc = piece;
            pin(pc);
            // corner-2-relative attributes: lanes 4 / 5 of the group hold corner 2's halves (same subtractions as before, one per lane)
            const int srcl = (lane & ~(kGroup - 1)) | 4 | (j & 1);
            const float4 w = make_float4(__shfl(pc.x, srcl, 64), __shfl(pc.y, srcl, 64), __shfl(pc.z, srcl, 64), __shfl(pc.w, srcl, 64));
            if (group_any) {
                float* const fo = reinterpret_cast<float*>(&pk_attr[gl]);   // n2 0, dn0 3, dn1 6, a2 9, da0 12, da1 15, v0 18, v1 21, v2 24
                if (j < 6) {
                    const float4 d = (cj == 2) ? pc : make_float4(pc.x - w.x, pc.y - w.y, pc.z - w.z, pc.w - w.w);
                    const int nb = (cj == 2) ? 0 : 3 + 3 * cj, ab = (cj == 2) ? 9 : 12 + 3 * cj;
                    if ((j & 1) == 0) { fo[nb] = d.x; fo[nb + 1] = d.y; fo[nb + 2] = d.z; fo[ab] = d.w; }
                    else { fo[ab + 1] = d.x; fo[ab + 2] = d.y; }
                }
                if (j < 3) { fo[18 + 3 * j] = vpos.x; fo[19 + 3 * j] = vpos.y; fo[20 + 3 * j] = vpos.z; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const FaceAttr& fa = pk_attr[gl];
        // owner == this face, mask > 0 (a box triangle's pixels are never list pixels), non-zero image gradient
        auto consume = [&](float4 w, float4 gs, int vs, int px, int py) {
            const int pk = __float_as_int(w.w);
            if (pix_face(pk) == f && pix_mask(pk) && !(gs.x == 0.f && gs.y == 0.f && gs.z == 0.f))
                pixel_backward(A, fa, &view_c[vs][0], w.x, w.y, gs, px, py, pg);
        };
        // a lane's tasks of one round, software-pipelined: the next task's two loads are in flight while the current one is
        // evaluated (a lane has ~5 tasks at 48 views; `have` = the first task (t) is already loaded into w / gs)
        auto run = [&](int t, int nt, bool have, float4 w, float4 gs, int vs, int px, int py) {
            if (!have && t < nt) {
                const long long gid = locate(t, vs, px, py);
                w = bwd16[gid];
                gs = g_shaded[gid];
            }
            while (t < nt) {
                const int tn = t + kGroup;
                float4 w2 = w, gs2 = gs;
                int vs2 = 0, px2 = 0, py2 = 0;
                if (tn < nt) {
                    const long long gid2 = locate(tn, vs2, px2, py2);
                    w2 = bwd16[gid2];
                    gs2 = g_shaded[gid2];
                }
                pin(w); pin(gs);
                consume(w, gs, vs, px, py);
                w = w2; gs = gs2; vs = vs2; px = px2; py = py2; t = tn;
            }
        };
        run(j, nt0, j < nt0, pw, pgs, pvs, ppx, ppy);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int fl = f;
        asm volatile("" : "+v"(fl));                                 // the rare paths below load the entries again through this copy
        if (__ballot(kTaskG < T) != 0ull) {                            // wave-uniform; further rounds are rare (> 64 candidates of a face)
            int2 e2[VPL];
            int kind2[VPL];
            load_entries(fl, e2, kind2);
            for (int R = kTaskG; __ballot(R < T) != 0ull; R += kTaskG) {
                expand(R, e2, kind2);
                run(j, min(kTaskG, T - R), false, pw, pgs, 0, 0, 0);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int c = 0; c < 9; c++) A[k][c] = group_sum(A[k][c]);
    if (facelive && j < 3) {
        float o[9];
#pragma unroll
        for (int c = 0; c < 9; c++) o[c] = (j == 0) ? A[0][c] : ((j == 1) ? A[1][c] : A[2][c]);
        float4* dst = part + ((size_t)g * 3 * F + es) * 3;
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        dst[2] = make_float4(o[8], 0.f, 0.f, 0.f);
    }
    FMHR_WTS_END(shade);
}

struct LargeArgs { const int2* snap; const int2* large_q; const int* large_cnt; const unsigned long long* covw; float* vacc; int workers; };

// Backward of the queued large-box (view, face)s: ONE WAVE per (view, face) scans the face's bounding box, 64 pixels per iteration
// (independent loads), does the per-pixel backward for the pixels the face owns and sums the 27 corner values over the wave.
// Inside k_face_bwd one lane walked a linked list of the face's pixels instead: a dependent load per pixel with its whole wave waiting —
// 187 us for a 778-vertex hand in 8 views of 512 x 334 (every triangle is a large-box one), and on the metric mesh the reason why the
// kernel slowed down during an optimisation run: sliver triangles appear (0 large boxes among the owners at step 64, 110 at step 320,
// 670 at step 960; median 36 -> 56 pixels, up to 240: tools/analysis/face_bwd_over_steps.py) and took it from 73 to 85 .. 137 us.
//   TO_VACC = false (coarse mesh, thousands of items, a launch of its own after k_face_bwd): float atomics into the face's corner
//             slots of `part`, which k_face_bwd wrote before;
//   TO_VACC = true  (micro-triangle mesh, a few hundred items, run by EXTRA WORKGROUPS of the k_face_bwd launch itself, concurrently
//             with the faces): float atomics into vacc [V] x 3 float4, per VERTEX the packing of one corner slot (normal.xyz albedo.x |
//             albedo.yz vertex.xy | vertex.z), which k_mesh_tail_a adds to the vertex's corner sums and re-zeroes — the faces' own plain
//             stores to `part` and these atomics never meet.
template <bool TO_VACC>
__device__ __forceinline__ void
face_bwd_large_items(const float4* __restrict__ bwd16, const int32_t* __restrict__ tri, const int2* __restrict__ snap,
                     const float4* __restrict__ attr8, const float* __restrict__ sh, const float4* __restrict__ g_shaded,
                     const int32_t* __restrict__ corner_slot, const float4* __restrict__ cmat, const float* __restrict__ verts,
                     const int2* __restrict__ large_q, const int* __restrict__ large_cnt, const unsigned long long* __restrict__ covw,
                     const unsigned char* __restrict__ own8, const int* __restrict__ stamp,
                     int n, int V, int F, int H, int W, int Wp, int GV, float* __restrict__ out, const PixGrid pg, int first, int stride,
                     const int* __restrict__ qstamp)
{
    constexpr int kWaves = kBlock / 64;
    __shared__ float view_c[kWaves][28];
    __shared__ FaceAttr pk[kWaves];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cnt = large_cnt[1];                                  // the second rasterizer pass kept a copy when it rearmed the queue
    if (qstamp != nullptr && (qstamp[0] & 1)) large_q += (size_t)n * F;   // the queue's half of this step's parity (raster.hip: k_raster_tris)
    const long long P = (long long)H * Wp;                         // W clips the box, Wp (row pitch) addresses
    const unsigned char cur = own8 != nullptr ? (unsigned char)stamp[0] : (unsigned char)0;
    for (int it = first; it < cnt; it += stride) {                 // wave-uniform
        const int2 bf = large_q[it];
        const int b = bf.x, f = bf.y;
        // the queue holds every large-box (view, face) of the step; only the ones that own a pixel have anything to do
        if (own8 != nullptr && own8[(size_t)b * F + f] != cur) continue;
        const int3a ti = *reinterpret_cast<const int3a*>(tri + (size_t)f * 3);
        if (lane < 28) {
            const int k = lane;                                    // the view's SH row, then the matrix rows as (x, y, w, z)
            view_c[wv][k] = (k < 9) ? sh[(size_t)b * 9 + k]
                                    : ((k < 25) ? reinterpret_cast<const float*>(cmat)[(size_t)b * 16 + ((k - 9) & ~3) + (((k - 9) & 3) == 2 ? 3 : (((k - 9) & 3) == 3 ? 2 : ((k - 9) & 3)))] : 0.f);
        }
        if (lane == 0) {
            const float4 q0 = attr8[(size_t)ti.x * 2], q1 = attr8[(size_t)ti.x * 2 + 1], u0 = attr8[(size_t)ti.y * 2], u1 = attr8[(size_t)ti.y * 2 + 1],
                         w0 = attr8[(size_t)ti.z * 2], w1 = attr8[(size_t)ti.z * 2 + 1];
            FaceAttr fa;
            fa.n2 = make_float3(w0.x, w0.y, w0.z); fa.a2 = make_float3(w0.w, w1.x, w1.y);
            fa.dn0 = make_float3(q0.x - fa.n2.x, q0.y - fa.n2.y, q0.z - fa.n2.z);
            fa.da0 = make_float3(q0.w - fa.a2.x, q1.x - fa.a2.y, q1.y - fa.a2.z);
            fa.dn1 = make_float3(u0.x - fa.n2.x, u0.y - fa.n2.y, u0.z - fa.n2.z);
            fa.da1 = make_float3(u0.w - fa.a2.x, u1.x - fa.a2.y, u1.y - fa.a2.z);
            fa.v0 = ld3f(verts + (size_t)ti.x * 3); fa.v1 = ld3f(verts + (size_t)ti.y * 3); fa.v2 = ld3f(verts + (size_t)ti.z * 3);
            pk[wv] = fa;
        }
        // the face's pixel box from its snapped vertices (rule R3, as the forward's setup)
        const int2 s0 = snap[(size_t)b * V + ti.x], s1 = snap[(size_t)b * V + ti.y], s2 = snap[(size_t)b * V + ti.z];
        const int mnx = min(s0.x, min(s1.x, s2.x)), mxx = max(s0.x, max(s1.x, s2.x));
        const int mny = min(s0.y, min(s1.y, s2.y)), mxy = max(s0.y, max(s1.y, s2.y));
        const int pxa = max((mnx - kHalf + kSub - 1) >> 8, 0), pxb = min((mxx - kHalf) >> 8, W - 1);
        const int pya = max((mny - kHalf + kSub - 1) >> 8, 0), pyb = min((mxy - kHalf) >> 8, H - 1);
        const int bw = pxb - pxa + 1, np = bw * (pyb - pya + 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float A[3][9];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 9; c++) A[k][c] = 0.f;
        for (int i = lane; i < np; i += 64) {
            int c;
            const int r = fast_divmod(i, bw, c);
            const int px = pxa + c, py = pya + r;
            const long long gid = (long long)b * P + (long long)py * Wp + px;
            // a pixel nobody covers in this step keeps the word of an earlier one (possibly this face's): the segment's coverage
            // word tells; a covered pixel's word is this step's, and the owner test settles the rest
            if (!((covw[gid >> 6] >> (gid & 63)) & 1ull)) continue;
            const float4 w = bwd16[gid];
            const int pkd = __float_as_int(w.w);
            if (pix_face(pkd) == f && pix_mask(pkd)) {
                const float4 gs = g_shaded[gid];
                if (!(gs.x == 0.f && gs.y == 0.f && gs.z == 0.f)) pixel_backward(A, pk[wv], &view_c[wv][0], w.x, w.y, gs, px, py, pg);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int c = 0; c < 9; c++)                             // wave_sum is valid in lane 0: broadcast it
                A[k][c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_sum(A[k][c]))));
        if (lane < 3) {
            float o[9];
#pragma unroll
            for (int c = 0; c < 9; c++) o[c] = (lane == 0) ? A[0][c] : ((lane == 1) ? A[1][c] : A[2][c]);
            const int vtx = (lane == 0) ? ti.x : ((lane == 1) ? ti.y : ti.z);
            float* dst = TO_VACC ? out + (size_t)vtx * 12
                                 : out + (((size_t)(b / GV) * 3 * F + corner_slot[f * 3 + lane]) * 3) * 4;
#pragma unroll
            for (int c = 0; c < 9; c++)
                if (o[c] != 0.f) atomicAdd(dst + c, o[c]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                           // the LDS slots are free for the wave's next item
    }
}

__global__ void __launch_bounds__(kBlock)
k_face_bwd_large(const float4* __restrict__ bwd16, const int32_t* __restrict__ tri, const int2* __restrict__ snap,
                 const float4* __restrict__ attr8, const float* __restrict__ sh, const float4* __restrict__ g_shaded,
                 const int32_t* __restrict__ corner_slot, const float4* __restrict__ cmat, const float* __restrict__ verts,
                 const int2* __restrict__ large_q, const int* __restrict__ large_cnt, const unsigned long long* __restrict__ covw,
                 int n, int V, int F, int H, int W, int Wp, int GV, float* __restrict__ part, const PixGrid pg, const int* __restrict__ stamp)
{
    constexpr int kWaves = kBlock / 64;
    face_bwd_large_items<false>(bwd16, tri, snap, attr8, sh, g_shaded, corner_slot, cmat, verts, large_q, large_cnt, covw, nullptr, nullptr,
                                n, V, F, H, W, Wp, GV, part, pg, blockIdx.x * kWaves + (threadIdx.x >> 6), gridDim.x * kWaves, stamp);
}

// VA views per lane in every view group but (VB != VA) the last, which holds the remainder in VB per lane: 48 views = a 32-view group
// (4 per lane) + a 16-view group that runs the 2-per-lane instantiation instead of the 4-per-lane one with half its slots empty
template <int VA, int VB>
__global__ void __launch_bounds__(kFbBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))       // <= 128 VGPRs: 4 waves per SIMD
k_face_bwd(const float4* __restrict__ bwd16, const int32_t* __restrict__ tri, const float* __restrict__ pos,
           const float4* __restrict__ attr8, const float* __restrict__ sh, const float4* __restrict__ g_shaded,
           const int2* __restrict__ cov16, const unsigned char* __restrict__ own8, const int* __restrict__ stamp,
           const int32_t* __restrict__ corner_slot,
           const float4* __restrict__ cmat, const float* __restrict__ verts, int n, int V, int F,
           int H, int W, int Wp, float4* __restrict__ part, const PixGrid pg, const LargeArgs la)
{
    const int g = blockIdx.y;
    // the FIRST la.workers workgroups of every grid row are set aside (micro-triangle mesh only); those of row 0 are the workers of the
    // large-box queue — one wave per queued (view, face), dispatched ahead of the faces and running concurrently with them
    if ((int)blockIdx.x < la.workers) {
        static_assert(kFbBlock == kBlock, "the queue workers use k_face_bwd_large's wave layout");
        if (g == 0)
            face_bwd_large_items<true>(bwd16, tri, la.snap, attr8, sh, g_shaded, corner_slot, cmat, verts, la.large_q, la.large_cnt, la.covw, own8, stamp,
                                       n, V, F, H, W, Wp, 0, la.vacc, pg, (int)blockIdx.x * (kBlock / 64) + (int)(threadIdx.x >> 6), la.workers * (kBlock / 64), stamp);
        return;
    }
    if (VB != VA && g == (int)gridDim.y - 1)                       // (block-uniform)
        face_bwd_body<VB>(bwd16, tri, pos, attr8, sh, g_shaded, cov16, own8, stamp, corner_slot, cmat, verts, n, V, F, H, Wp, part,
                          pg, g, g * kGroup * VA, la.workers);
    else
        face_bwd_body<VA>(bwd16, tri, pos, attr8, sh, g_shaded, cov16, own8, stamp, corner_slot, cmat, verts, n, V, F, H, Wp, part,
                          pg, g, g * kGroup * VA, la.workers);
}

int valid_mask_segments_impl(const float* valid_mask, int n, int H, int W, int Wp, unsigned char* vmnz, hipStream_t st)
{
    const long long N = (long long)n * H * Wp;
    if (N == 0) return FMHR_OK;
    k_valid_mask_segments<<<cdiv(N, kBlock), kBlock, 0, st>>>(valid_mask, N, W, Wp, make_udiv((unsigned)max(Wp / 64, 1)), vmnz);
    FMHR_CHECK_LAUNCH("k_valid_mask_segments");
    return FMHR_OK;
}

int active_segments_impl(const unsigned long long* covw, const unsigned char* vmnz, long long N, int W, unsigned char* act,
                         int* act_list, int* act_count, const int* seg_valid, float* cnt_part, float* aa_init, const ViewSel sel, hipStream_t st)
{
    if (N == 0) return FMHR_OK;
    long long nSeg = (N + 63) >> 6;
    k_active_segments<<<cdiv(nSeg, kActBlock), kActBlock, 0, st>>>(covw, vmnz, nSeg, W, act, act_list, act_count, seg_valid, cnt_part, N,
                                                                   (float4*)aa_init, sel);
    FMHR_CHECK_LAUNCH("k_active_segments");
    return FMHR_OK;
}

int active_segments_blocks(long long N) { return cdiv((N + 63) >> 6, kActBlock); }

int densify_impl(const unsigned long long* covw, const unsigned char* act, long long N, const float* pix16, float* rast, float* aa,
                 hipStream_t st)
{
    if (N == 0) return FMHR_OK;
    k_densify<<<cdiv(N, kBlock), kBlock, 0, st>>>(covw, act, N, (const float4*)pix16, (float4*)rast, (float4*)aa);
    FMHR_CHECK_LAUNCH("k_densify");
    return FMHR_OK;
}

int loss_pix_impl(const unsigned long long* covw, const float* aa, const float* img, const float* mask, const float* valid_mask,
                  long long N, int W, int Wp, const ViewSel sel, const int* act_list, const int* act_count, const float* n_valid_global, float w_sfs, float w_mask,
                  float mse_den, float* g_shaded, float* part_l1, float* part_l2, float* losses, hipStream_t st)
{
    if (N == 0) return FMHR_OK;
    k_loss_pix<<<kPersistBlocks, kBlock, 0, st>>>(covw, (const float4*)aa, img, mask, valid_mask, N, W, Wp, make_udiv((unsigned)max(Wp / 64, 1)), sel, act_list, act_count,
                                                  n_valid_global, w_sfs, w_mask, mse_den, (float4*)g_shaded, part_l1,
                                                  part_l2, losses);
    FMHR_CHECK_LAUNCH("k_loss_pix");
    return FMHR_OK;
}

// views per lane: one group spans the whole batch when it can (up to kGroup * kMaxVPL = 48 views): the fixed part of a (face, view
// group) is paid once and the mesh tail adds ONE plane of corner slots.  Round 3 had settled on 4 per lane (a 32-view and a 16-view
// group at 48 views) because 6 spilled at the 128 VGPRs of 4 waves per SIMD; with the pixel-list walk gone the 6-per-lane instantiation
// needs 122 and no spill: four alternating rounds, 4 + 2 per lane 2 817 [2 785 .. 2 836] it/s, k_face_bwd 72.4 us, tails 30.0 us ->
// 6 per lane 2 868 [2 860 .. 2 875], 69.6 us, 27.9 us (5 per lane over two groups, 40 + 8 views: 2 768).  Batches above 48 views keep the
// groups of 32 (their remainders run the 1 .. 4-per-lane instantiations).
static int face_bwd_vpl(int n, int H, int W)
{
    const int v = cdiv(n, kGroup);
    // (one group of 5 or 6 views per lane only up to 512 x 512 pixels per view: at 1024 x 1024 — C4, the views' pixel arrays 16 MB apart —
    // the same launch is SLOWER than 32 + 16 views, 196 vs 182 us)
    const bool one_group = v <= kMaxVPL && (long long)H * W <= 512ll * 512ll;
    return (v <= 4 || one_group) ? v : 4;
}

int face_bwd_groups(int n, int H, int W) { return cdiv(n, kGroup * face_bwd_vpl(n, H, W)); }
int face_bwd_blocks(int F) { return cdiv(F, kFacesPerBlock); }

int face_bwd_impl(const float* bwd16, const int32_t* tri, const float* pos, const float* attr8, const float* sh, const float* g_shaded,
                  const void* cov16, const unsigned char* own8, const int* stamp, const int32_t* corner_slot, const float* cmat,
                  const float* verts, int n, int V, int F, int H, int W, int Wp, float* part,
                  const int2* snap, const int2* large_q, const int* large_cnt, const unsigned long long* covw, float* vacc, hipStream_t st)
{
    if (n == 0 || F == 0) return FMHR_OK;
    // large-box (view, face)s: a coarse mesh (every triangle is one) gives them a launch of their own after this one; a micro-triangle
    // mesh (a few hundred slivers late in an optimisation run) kLargeWorkers extra workgroups of THIS launch, which add their sums to
    // `vacc` (per vertex) for the mesh tail.  Either way no lane of k_face_bwd walks a pixel list any more.
    const bool coarse = raster_use_queue(H, W, F);
#ifndef FMHR_LARGE_WORKERS
#define FMHR_LARGE_WORKERS 256
#endif
    constexpr int kLargeWorkers = FMHR_LARGE_WORKERS;              // x 4 waves: one queued (view, face) per wave and round; an idle one leaves at once
    LargeArgs la;
    la.snap = snap; la.large_q = large_q; la.large_cnt = large_cnt; la.covw = covw; la.vacc = vacc;
    const bool workers = !coarse && vacc != nullptr;
    la.workers = workers ? kLargeWorkers : 0;
    const dim3 grid(cdiv(F, kFacesPerBlock) + la.workers, face_bwd_groups(n, H, Wp));
    const PixGrid pg = make_pix_grid(H, W);
#define FMHR_FACE_BWD(VA, VB)                                                                                                       \
    k_face_bwd<VA, VB><<<grid, kFbBlock, 0, st>>>((const float4*)bwd16, tri, pos, (const float4*)attr8, sh, (const float4*)g_shaded, \
                                             (const int2*)cov16, own8, stamp, corner_slot, (const float4*)cmat, verts, n, V, F, H, W, Wp, \
                                             (float4*)part, pg, la)
    static_assert(kMaxVPL == 6, "the instantiation table below: 1 .. 4 views per lane with any remainder, 5 and 6 as single groups");
    {
        const int va = face_bwd_vpl(n, H, Wp), rem = n - (face_bwd_groups(n, H, Wp) - 1) * kGroup * va;     // views of the last group
        const int vb = min(va, cdiv(rem, kGroup));
        switch (va * 8 + vb) {
            case 1 * 8 + 1: FMHR_FACE_BWD(1, 1); break;
            case 2 * 8 + 1: FMHR_FACE_BWD(2, 1); break;
            case 2 * 8 + 2: FMHR_FACE_BWD(2, 2); break;
            case 3 * 8 + 1: FMHR_FACE_BWD(3, 1); break;
            case 3 * 8 + 2: FMHR_FACE_BWD(3, 2); break;
            case 3 * 8 + 3: FMHR_FACE_BWD(3, 3); break;
            case 4 * 8 + 1: FMHR_FACE_BWD(4, 1); break;
            case 4 * 8 + 2: FMHR_FACE_BWD(4, 2); break;
            case 4 * 8 + 3: FMHR_FACE_BWD(4, 3); break;
            case 5 * 8 + 5: FMHR_FACE_BWD(5, 5); break;
            case 6 * 8 + 6: FMHR_FACE_BWD(6, 6); break;
            default: FMHR_FACE_BWD(4, 4); break;
        }
    }
#undef FMHR_FACE_BWD
    FMHR_CHECK_LAUNCH("k_face_bwd");
    if (coarse) {
        k_face_bwd_large<<<2048, kBlock, 0, st>>>((const float4*)bwd16, tri, snap, (const float4*)attr8, sh, (const float4*)g_shaded, corner_slot,
                                                  (const float4*)cmat, verts, large_q, large_cnt, covw, n, V, F, H, W, Wp, kGroup * face_bwd_vpl(n, H, Wp), part, pg, stamp);
        FMHR_CHECK_LAUNCH("k_face_bwd_large");
    }
    return FMHR_OK;
}

}  // namespace fmhr

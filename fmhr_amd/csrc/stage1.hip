// fmhr_amd/csrc/stage1.hip — the stage-1 ("albedo") iteration of the HAM loop, fused, for gfx950.
// Replaces the loop body /root/reference/mesh_sfs_optim.py:205-236: interpolate(albedo) -> antialias -> SH radiance of the (antialiased,
// normalised) normals x albedo -> L1 over the valid pixels -> backward to albedo [V,3] and sh_coeffs [n,9] -> Adam.
//
// Stage 1 moves only the albedo and the SH coefficients; the vertices are constants (:191, :195).  Everything geometric is therefore the
// same in all of its epoch_albedo x ceil(num / batch) iterations: the rasterization, the antialiased normals, the validity mask AND
// the antialias operator itself, which for a fixed geometry is a LINEAR map of its colour input (out[tgt] += alpha (in[gid1] - in[gid])
// per blending pair, antialias.hip).  The antialiased albedo of a valid pixel is thus a fixed sparse combination of vertex albedos,
//     aa_albedo[p] = sum_k w_pk albedo[v_pk]      (3 entries: the pixel's triangle; 6 more per pair that blends INTO the pixel),
// and an iteration is one sparse product with that matrix (forward, one lane per valid pixel) and one with its transpose (backward, one
// group of lanes per vertex, gather-only: no float atomics, deterministic), with the shading, the loss, the SH gradient and both Adam
// updates fused in.  The tables ("plan") are built once per stage from the outputs of the generic ops; a step is two launches.
#include <string.h>
#include <cstring>
#include "common.h"
#include "adam.h"
#include <rocprim/device/device_scan.hpp>

namespace fmhr {

#ifndef FMHR_S1_SUB
#define FMHR_S1_SUB 16
#endif
#ifndef FMHR_S1_U
#define FMHR_S1_U 8
#endif
constexpr int kS1Sub = FMHR_S1_SUB;  // lanes per vertex in the backward
constexpr int kS1U = FMHR_S1_U;      // entries a lane has in flight (8 / 16 / 32 lanes x 2 / 4 / 8 entries: 0.090 - 0.094 ms per step, 64 lanes 0.101)
constexpr int AA_DOWN_BIT = 4;       // antialias.hip: item code bit "the pair is (pixel, pixel below)"

// ------------------------------------------------------------------------------------------------ plan: rows
__global__ void __launch_bounds__(kBlock)
k_s1_flags(const float4* __restrict__ rast, const float* __restrict__ mask, long long N, int* __restrict__ flag)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    flag[i] = (mask[i] > 0.f && rast[i].w > 0.f) ? 1 : 0;          // mesh_sfs_optim.py:221
}

// first row of every view (rows of a view start at a multiple of kBlock: a workgroup of the forward lies in ONE view)
__global__ void k_s1_view_rows(const int* __restrict__ flag, const int* __restrict__ excl, int n, long long P, int* __restrict__ view_row0,
                               int* __restrict__ view_valid)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int row = 0;
    for (int b = 0; b < n; b++) {
        const long long last = (long long)(b + 1) * P - 1;
        const int cnt = excl[last] + flag[last] - excl[(long long)b * P];
        view_row0[b] = row;
        view_valid[b] = cnt;
        row += (cnt + kBlock - 1) / kBlock * kBlock;
    }
    view_row0[n] = row;
}

__global__ void __launch_bounds__(kBlock)
k_s1_rowid(const int* __restrict__ flag, const int* __restrict__ excl, const int* __restrict__ view_row0, long long N, long long P,
           int* __restrict__ rowid, int* __restrict__ row_flags)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    int r = -1;
    if (flag[i]) {
        const long long b = i / P;
        r = view_row0[b] + excl[i] - excl[b * P];
        row_flags[r] = 0x100;                                       // a real row (padding rows stay 0)
    }
    rowid[i] = r;
}

// A blending pair (pixel gid, its right / lower neighbour gid1, signed alpha) adds alpha (in[gid1] - in[gid]) to out[tgt], tgt = gid
// if alpha > 0 else gid1 (antialias.hip: aa_fwd_pixel): the target's own value loses |alpha|, the other pixel's gains it.  A pixel is
// the target of at most four pairs, one per neighbour ("slot": 0 right, 1 below, 2 left, 3 above).
struct S1Item { long long tgt, oth; int slot; float a; };
__device__ __forceinline__ bool s1_item(const int4 it, const float alpha, int n, int H, int W, S1Item& o)
{
    const int px = it.x, py = it.y, b = it.z, down = (it.w & AA_DOWN_BIT) ? 1 : 0;
    if (!(alpha != 0.f) || (unsigned)b >= (unsigned)n || (unsigned)px >= (unsigned)W || (unsigned)py >= (unsigned)H) return false;
    if (down ? (py + 1 >= H) : (px + 1 >= W)) return false;
    const long long gid = ((long long)b * H + py) * W + px, gid1 = gid + (down ? W : 1);
    const bool own = alpha > 0.f;
    o.tgt = own ? gid : gid1; o.oth = own ? gid1 : gid;
    o.slot = (own ? 0 : 2) + down;
    o.a = fabsf(alpha);
    return true;
}

__global__ void __launch_bounds__(kBlock)
k_s1_item_flags(const int4* __restrict__ items, const float* __restrict__ alphas, int n_items, const float4* __restrict__ rast,
                const int* __restrict__ rowid, int n, int H, int W, int* __restrict__ row_flags)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_items) return;
    S1Item t;
    if (!s1_item(items[i], alphas[i], n, H, W, t)) return;
    const int r = rowid[t.tgt];
    if (r < 0) return;
    const bool oc = rast[t.oth].w > 0.f;                           // an uncovered pixel interpolates to 0: no entries for it
    atomicOr(&row_flags[r], (1 << t.slot) | (oc ? (16 << t.slot) : 0));
}

__device__ __forceinline__ int s1_slot_entries(int flags, int slot) { return ((flags >> slot) & 1) ? (3 + 3 * ((flags >> (4 + slot)) & 1)) : 0; }

__global__ void __launch_bounds__(kBlock)
k_s1_row_count(const int* __restrict__ row_flags, int rows, int* __restrict__ cnt)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r > rows) return;
    int c = 0;
    if (r < rows) {
        const int fl = row_flags[r];
        if (fl & 0x100) c = 3 + s1_slot_entries(fl, 0) + s1_slot_entries(fl, 1) + s1_slot_entries(fl, 2) + s1_slot_entries(fl, 3);
    }
    cnt[r] = c;                                                     // cnt[rows] = 0: the scan's last element is the total
}

// own entries of every valid pixel + its per-row constants
__global__ void __launch_bounds__(kBlock)
k_s1_fill_rows(const float4* __restrict__ rast, const int32_t* __restrict__ tri, const int* __restrict__ rowid, const float* __restrict__ normals,
               const float* __restrict__ img, long long N, int V, int F, const int* __restrict__ row_ptr, int* __restrict__ ent_col,
               float* __restrict__ ent_w, float* __restrict__ row_n, float* __restrict__ row_img)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const int r = rowid[i];
    if (r < 0) return;
    const float4 q = rast[i];
    const int f = (int)q.w - 1;
    const int e = row_ptr[r];
    const float w[3] = {q.x, q.y, 1.0f - q.x - q.y};                // interpolate: u a0 + v a1 + (1 - u - v) a2 (raster.hip: k_interp_fwd)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int v = ((unsigned)f < (unsigned)F) ? tri[f * 3 + k] : 0;
        if ((unsigned)v >= (unsigned)V) v = 0;
        ent_col[e + k] = v; ent_w[e + k] = ((unsigned)f < (unsigned)F) ? w[k] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { row_n[(size_t)r * 3 + c] = normals[i * 3 + c]; row_img[(size_t)r * 3 + c] = img[i * 3 + c]; }
}

// entries of the pairs that blend into a valid pixel: + |alpha| x the other pixel's interpolation (if covered), - |alpha| x its own
__global__ void __launch_bounds__(kBlock)
k_s1_fill_items(const int4* __restrict__ items, const float* __restrict__ alphas, int n_items, const float4* __restrict__ rast,
                const int32_t* __restrict__ tri, const int* __restrict__ rowid, int n, int V, int F, int H, int W,
                const int* __restrict__ row_flags, const int* __restrict__ row_ptr, int* __restrict__ ent_col, float* __restrict__ ent_w)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_items) return;
    S1Item t;
    if (!s1_item(items[i], alphas[i], n, H, W, t)) return;
    const int r = rowid[t.tgt];
    if (r < 0) return;
    const int fl = row_flags[r];
    int e = row_ptr[r] + 3;
    for (int s = 0; s < t.slot; s++) e += s1_slot_entries(fl, s);
    auto put = [&](long long pix, float scale) {
        const float4 q = rast[pix];
        const int f = (int)q.w - 1;
        const float w[3] = {q.x, q.y, 1.0f - q.x - q.y};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int v = ((unsigned)f < (unsigned)F) ? tri[f * 3 + k] : 0;
            if ((unsigned)v >= (unsigned)V) v = 0;
            ent_col[e + k] = v; ent_w[e + k] = ((unsigned)f < (unsigned)F) ? scale * w[k] : 0.f;
        }
        e += 3;
    };
    if ((fl >> (4 + t.slot)) & 1) put(t.oth, t.a);
    put(t.tgt, -t.a);
}

// ------------------------------------------------------------------------------------------------ plan: transpose, per view
__device__ __forceinline__ int s1_view_of_row(const int* __restrict__ view_row0, int n, int row)
{
    int lo = 0, hi = n;                                             // view_row0[lo] <= row < view_row0[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (view_row0[mid] <= row) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool FILL>
__global__ void __launch_bounds__(kBlock)
k_s1_transpose(const int* __restrict__ row_ptr, const int* __restrict__ ent_col, const float* __restrict__ ent_w,
               const int* __restrict__ view_row0, int n, int V, int rows, int* __restrict__ t_cnt, const int* __restrict__ t_ptr,
               int* __restrict__ t_row, float* __restrict__ t_w)
{
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= rows) return;
    const int e0 = row_ptr[r], e1 = row_ptr[r + 1];
    if (e1 == e0) return;
    const int b = s1_view_of_row(view_row0, n, r);
    for (int e = e0; e < e1; e++) {
        const long long tr = (long long)ent_col[e] * n + b;          // (vertex, view): a vertex's entries of all views are contiguous
        const int pos = atomicAdd(&t_cnt[tr], 1);
        if (FILL) { t_row[t_ptr[tr] + pos] = r; t_w[t_ptr[tr] + pos] = ent_w[e]; }
    }
}

// rows of the transpose are short (the pixels of a vertex's ~6 faces in one view): insertion sort by (row, weight bits) makes the
// summation order of the backward independent of the order in which the atomics above were served
__global__ void __launch_bounds__(kBlock)
k_s1_tsort(const int* __restrict__ t_ptr, long long nrows_t, int* __restrict__ t_row, float* __restrict__ t_w)
{
    const long long tr = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (tr >= nrows_t) return;
    const int a = t_ptr[tr], b = t_ptr[tr + 1];
    for (int i = a + 1; i < b; i++) {
        const int kr = t_row[i];
        const float kw = t_w[i];
        int j = i - 1;
        while (j >= a && (t_row[j] > kr || (t_row[j] == kr && __float_as_int(t_w[j]) > __float_as_int(kw)))) {
            t_row[j + 1] = t_row[j]; t_w[j + 1] = t_w[j];
            j--;
        }
        t_row[j + 1] = kr; t_w[j + 1] = kw;
    }
}

// ------------------------------------------------------------------------------------------------ the step
struct S1Plan {
    int n, V, rows, blocks;
    const int* view_row0; const int* view_valid; const int* row_ptr; const int* ent_col; const float* ent_w;
    const float* row_n; const float* row_img; const int* t_ptr; const int* t_row; const float* t_w;
};

// sum over the batch's views of their valid-pixel counts (F.l1_loss averages over the batch's 3 Pv elements, :231); every workgroup
// recomputes it from the n <= a few dozen counts: no kernel of its own, no host round trip
__device__ __forceinline__ float s1_batch_valid(const int* __restrict__ view_valid, const int* __restrict__ view_sel, int n, int* sm4)
{
    int c = 0;                                                      // integer sum: exact whatever the batch size
    for (int b = threadIdx.x; b < n; b += kBlock) c += (view_sel == nullptr || view_sel[b] != 0) ? view_valid[b] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm4[wv] = c;
    __syncthreads();
    const int r = sm4[0] + sm4[1] + sm4[2] + sm4[3];
    __syncthreads();
    return (float)r;
}

// forward + the pixel half of the backward, one lane per valid pixel (row): a = sum w albedo[col]; radiance = SH basis of the row's
// normal . sh[view] (models/utils.py:208-226); d = radiance a - img; leaves R = k sign(d) radiance (the gradient of the loss with respect
// to the row's antialiased albedo) for the transpose product, and the workgroup's sums of |d| and of the SH gradient.
__global__ void __launch_bounds__(kBlock)
k_s1_fwd(const S1Plan P, const float* __restrict__ albedo, const float* __restrict__ sh, const int* __restrict__ view_sel, float sfs_weight,
         float* __restrict__ R, float* __restrict__ part)
{
    __shared__ float sm[4][10];
    __shared__ int sm4[4];
    const int row = blockIdx.x * kBlock + threadIdx.x;
    const int b = s1_view_of_row(P.view_row0, P.n, blockIdx.x * kBlock);      // workgroup-uniform
    if (view_sel != nullptr && view_sel[b] == 0) {                            // not in this batch: its rows carry no gradient
        if (P.row_ptr[row + 1] > P.row_ptr[row]) { R[(size_t)row * 3] = 0.f; R[(size_t)row * 3 + 1] = 0.f; R[(size_t)row * 3 + 2] = 0.f; }
        return;
    }
    const float pv = s1_batch_valid(P.view_valid, view_sel, P.n, sm4);
    const float k = (pv > 0.f) ? sfs_weight / (3.0f * pv) : 0.f;
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; i++) acc[i] = 0.f;
    const int e0 = P.row_ptr[row], e1 = P.row_ptr[row + 1];
    if (e1 > e0) {
        // the row's constants do not depend on its entries: requested with them (one dependent chain row_ptr -> entries -> albedo)
        const F3U nrm = *reinterpret_cast<const F3U*>(P.row_n + (size_t)row * 3);
        const F3U tgt = *reinterpret_cast<const F3U*>(P.row_img + (size_t)row * 3);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        {   // every row has at least the three entries of its own triangle: all three in flight at once
            const int c0 = P.ent_col[e0], c1 = P.ent_col[e0 + 1], c2 = P.ent_col[e0 + 2];
            const float w0 = P.ent_w[e0], w1 = P.ent_w[e0 + 1], w2 = P.ent_w[e0 + 2];
            const F3U q0 = *reinterpret_cast<const F3U*>(albedo + (size_t)c0 * 3), q1 = *reinterpret_cast<const F3U*>(albedo + (size_t)c1 * 3),
                      q2 = *reinterpret_cast<const F3U*>(albedo + (size_t)c2 * 3);
            a0 = w0 * q0.x + w1 * q1.x + w2 * q2.x; a1 = w0 * q0.y + w1 * q1.y + w2 * q2.y; a2 = w0 * q0.z + w1 * q1.z + w2 * q2.z;
        }
        for (int e = e0 + 3; e < e1; e++) {                          // pairs that blend into this pixel (a few rows in a thousand)
            const float w = P.ent_w[e];
            const F3U q = *reinterpret_cast<const F3U*>(albedo + (size_t)P.ent_col[e] * 3);
            a0 += w * q.x; a1 += w * q.y; a2 += w * q.z;
        }
        const float x = nrm.x, y = nrm.y, z = nrm.z;
        const float bs[9] = {1.0f, y, z, x, x * y, y * z, 2.0f * z * z - x * x - y * y, z * x, x * x - y * y};
        float rad = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i++) rad += sh[(size_t)b * 9 + i] * bs[i];
        const float d0 = rad * a0 - tgt.x, d1 = rad * a1 - tgt.y, d2 = rad * a2 - tgt.z;
        const float s0 = (d0 > 0.f) ? 1.f : ((d0 < 0.f) ? -1.f : 0.f), s1 = (d1 > 0.f) ? 1.f : ((d1 < 0.f) ? -1.f : 0.f),
                    s2 = (d2 > 0.f) ? 1.f : ((d2 < 0.f) ? -1.f : 0.f);
        const float kr = k * rad;
        *reinterpret_cast<F3U*>(R + (size_t)row * 3) = F3U{kr * s0, kr * s1, kr * s2};
        const float t = k * (s0 * a0 + s1 * a1 + s2 * a2);          // d loss / d radiance
#pragma unroll
        for (int i = 0; i < 9; i++) acc[i] = t * bs[i];
        acc[9] = fabsf(d0) + fabsf(d1) + fabsf(d2);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const float r = wave_sum(acc[i]);
        if (lane == 0) sm[wv][i] = r;
    }
    __syncthreads();
    if (threadIdx.x < 10) part[(size_t)blockIdx.x * 10 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__device__ __forceinline__ float s1_sum_sub(float v)
{
#pragma unroll
    for (int o = kS1Sub / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// the vertex half of the backward: galbedo[v] = sum_e w_e R[row_e] over the vertex's transposed entries (kS1Sub lanes per vertex),
// + Adam on the albedo; the last n workgroups finish the views' SH gradients / loss parts and apply Adam to
// sh_coeffs (every row, every step: a view outside the batch has a zero gradient but its moments still decay, as in torch)
__global__ void __launch_bounds__(kBlock)
k_s1_bwd(const S1Plan P, const float* __restrict__ R, const float* __restrict__ part, const int* __restrict__ view_sel, float sfs_weight,
         float* __restrict__ galbedo, float* __restrict__ gsh, float* __restrict__ loss_view, const AdamTensor ad_a, const AdamTensor ad_s,
         const int vblocks)
{
    if ((int)blockIdx.x >= vblocks) {
        __shared__ float sm[4][10];
        __shared__ int sm4[4];
        const int b = (int)blockIdx.x - vblocks;
        const bool on = view_sel == nullptr || view_sel[b] != 0;
        const float pv = s1_batch_valid(P.view_valid, view_sel, P.n, sm4);
        float acc[10];
#pragma unroll
        for (int i = 0; i < 10; i++) acc[i] = 0.f;
        if (on) {
            const int b0 = P.view_row0[b] / kBlock, b1 = P.view_row0[b + 1] / kBlock;
            for (int j = b0 + (int)threadIdx.x; j < b1; j += kBlock)
#pragma unroll
                for (int i = 0; i < 10; i++) acc[i] += part[(size_t)j * 10 + i];
        }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const float r = wave_sum(acc[i]);
            if (lane == 0) sm[wv][i] = r;
        }
        __syncthreads();
        if (threadIdx.x < 10) {
            const float r = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
            if (threadIdx.x < 9) {
                gsh[(size_t)b * 9 + threadIdx.x] = r;
                if (ad_s.p != nullptr) adam_element(ad_s, (long long)b * 9 + threadIdx.x, r);
            } else {
                loss_view[b] = (pv > 0.f) ? r * (sfs_weight / (3.0f * pv)) : 0.f;
            }
        }
        return;
    }
    const int v = ((int)blockIdx.x * kBlock + (int)threadIdx.x) / kS1Sub, sub = threadIdx.x & (kS1Sub - 1);
    const bool live = v < P.V;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (live) {
        // the vertex's entries of all views (rows outside the batch hold R = 0): kS1Sub lanes stride over them, kS1U entries in
        // flight per lane — as a loop over views with a dependent (pointer -> entry -> R) chain per entry this kernel took 100 us
        const int e0 = P.t_ptr[(long long)v * P.n], e1 = P.t_ptr[(long long)(v + 1) * P.n];
        for (int e = e0 + sub; e < e1; e += kS1U * kS1Sub) {
            int row[kS1U];
            float w[kS1U];
#pragma unroll
            for (int u = 0; u < kS1U; u++) {
                const int ee = e + u * kS1Sub;
                const bool in = ee < e1;
                row[u] = in ? P.t_row[ee] : 0;
                w[u] = in ? P.t_w[ee] : 0.f;
            }
            F3U q[kS1U];
#pragma unroll
            for (int u = 0; u < kS1U; u++) q[u] = *reinterpret_cast<const F3U*>(R + (size_t)row[u] * 3);
#pragma unroll
            for (int u = 0; u < kS1U; u++)
                if (e + u * kS1Sub < e1) { g0 += w[u] * q[u].x; g1 += w[u] * q[u].y; g2 += w[u] * q[u].z; }
        }
    }
    g0 = s1_sum_sub(g0); g1 = s1_sum_sub(g1); g2 = s1_sum_sub(g2);
    if (live && sub < 3) {          // one component per lane (every lane holds the sums): a 12-B run per vertex, a third of the instructions
        const float g = (sub == 0) ? g0 : ((sub == 1) ? g1 : g2);
        galbedo[(size_t)v * 3 + sub] = g;
        if (ad_a.p != nullptr) adam_element(ad_a, (long long)v * 3 + sub, g);
    }
}

// ------------------------------------------------------------------------------------------------ SH least-squares initialisation
// mesh_sfs_optim.py:150-154,164-166: per view (and over all views) the 9 SH coefficients that fit the gray image on the valid pixels,
// np.linalg.lstsq on a [Pv, 9] float64 design matrix — on the host, one view at a time, 2.8 of the 3.0 s of a 48-view run_ham.
// The normal equations need only sum phi phi^T (45 numbers) and sum phi g (9) per view: accumulated here in float64 over the valid
// pixels (the basis products in float32 first, as get_matrix forms them from float32 normals, models/utils.py:188-205), as per-
// workgroup partials (summed on the host in a fixed order); the 9 x 9 solves stay on the host.
constexpr int kShAcc = 55;           // 45 upper-triangle entries, 9 right-hand sides, the pixel count
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(kBlock)
k_sh_normal_eq(const float* __restrict__ normals, const float4* __restrict__ rast, const float* __restrict__ mask, const float* __restrict__ gray,
               long long P, double* __restrict__ part)
{
    __shared__ double sm[4][kShAcc];
    const int b = blockIdx.y;
    double acc[kShAcc];
#pragma unroll
    for (int i = 0; i < kShAcc; i++) acc[i] = 0.0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < P; i += (long long)gridDim.x * kBlock) {
        const long long g = (long long)b * P + i;
        if (!(mask[g] > 0.f && rast[g].w > 0.f)) continue;
        const float x = normals[g * 3], y = normals[g * 3 + 1], z = normals[g * 3 + 2];
        const float t0 = 2.0f * z * z, t1 = x * x, t2 = y * y;
        const double ph[9] = {1.0, (double)y, (double)z, (double)x, (double)(x * y), (double)(y * z), (double)((t0 - t1) - t2), (double)(z * x),
                              (double)(t1 - t2)};
        const double gv = (double)gray[g];
        int k = 0;
#pragma unroll
        for (int r = 0; r < 9; r++)
#pragma unroll
            for (int c = r; c < 9; c++) acc[k++] += ph[r] * ph[c];
#pragma unroll
        for (int r = 0; r < 9; r++) acc[45 + r] += ph[r] * gv;
        acc[54] += 1.0;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kShAcc; i++) {
        const double r = wave_sum_d(acc[i]);
        if (lane == 0) sm[wv][i] = r;
    }
    __syncthreads();
    if (threadIdx.x < kShAcc)
        part[((size_t)b * gridDim.x + blockIdx.x) * kShAcc + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}

// sums of img / radiance(sh, normal) over the valid pixels (:173-174: albedo_mean), float32 per pixel as the reference's GPU tensors
__global__ void __launch_bounds__(kBlock)
k_sh_albedo_sums(const float* __restrict__ normals, const float4* __restrict__ rast, const float* __restrict__ mask, const float* __restrict__ img,
                 const float* __restrict__ sh, long long P, double* __restrict__ part)
{
    __shared__ double sm[4][4];
    const int b = blockIdx.y;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    float c[9];
#pragma unroll
    for (int i = 0; i < 9; i++) c[i] = sh[i];
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < P; i += (long long)gridDim.x * kBlock) {
        const long long g = (long long)b * P + i;
        if (!(mask[g] > 0.f && rast[g].w > 0.f)) continue;
        const float x = normals[g * 3], y = normals[g * 3 + 1], z = normals[g * 3 + 2];
        float r = c[0];                                              // models/utils.py:214-226, statement for statement
        r = r + c[1] * y; r = r + c[2] * z; r = r + c[3] * x;
        r = r + c[4] * x * y; r = r + c[5] * y * z; r = r + c[6] * (2.0f * z * z - x * x - y * y); r = r + c[7] * z * x; r = r + c[8] * (x * x - y * y);
        acc[0] += (double)(img[g * 3] / r); acc[1] += (double)(img[g * 3 + 1] / r); acc[2] += (double)(img[g * 3 + 2] / r);
        acc[3] += 1.0;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double r = wave_sum_d(acc[i]);
        if (lane == 0) sm[wv][i] = r;
    }
    __syncthreads();
    if (threadIdx.x < 4) part[((size_t)b * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}

static int scan_exclusive(const int* in, int* out, size_t count, void* tmp, size_t tmp_bytes, hipStream_t st)
{
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, in, out, 0, count, rocprim::plus<int>(), st);
    if (e != hipSuccess) return hip_fail(e, "stage1: scan size");
    FMHR_CHECK_ARG(need <= tmp_bytes, "stage1: scan scratch too small (%zu < %zu)", tmp_bytes, need);
    e = rocprim::exclusive_scan(tmp, need, in, out, 0, count, rocprim::plus<int>(), st);
    if (e != hipSuccess) return hip_fail(e, "stage1: scan");
    return FMHR_OK;
}
static size_t scan_tmp_bytes(size_t count)
{
    size_t need = 0;
    (void)rocprim::exclusive_scan(nullptr, need, (const int*)nullptr, (int*)nullptr, 0, count, rocprim::plus<int>(), (hipStream_t)0);
    return (need + 255) & ~(size_t)255;
}

}  // namespace fmhr

using namespace fmhr;

// ---- plan, step 1: number the valid pixels.  ws = [flags N ints][excl N ints][scan scratch]
extern "C" size_t fmhr_stage1_rows_workspace_bytes(int n, int H, int W)
{
    const size_t N = (size_t)n * H * W;
    return 2 * ((N * 4 + 255) & ~(size_t)255) + scan_tmp_bytes(N);
}

extern "C" int fmhr_stage1_rows(const float* rast, const float* mask, int n, int H, int W, int32_t* rowid, int32_t* view_row0,
                                int32_t* view_valid, int32_t* row_flags, long long row_flags_len, void* ws, size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(n >= 1 && H >= 1 && W >= 1 && (long long)n * H * W + (long long)n * kBlock < (1ll << 31), "stage1_rows: bad dims n=%d H=%d W=%d", n, H, W);
    FMHR_CHECK_ARG(rast && mask && rowid && view_row0 && view_valid && row_flags && ws, "stage1_rows: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_stage1_rows_workspace_bytes(n, H, W), "stage1_rows: workspace too small");
    const long long N = (long long)n * H * W, P = (long long)H * W;
    // rows <= N + n * (kBlock - 1): the caller sizes row_flags for that bound
    FMHR_CHECK_ARG(row_flags_len >= N + (long long)n * (kBlock - 1), "stage1_rows: row_flags holds %lld ints, %lld needed", row_flags_len,
                   N + (long long)n * (kBlock - 1));
    hipStream_t st = (hipStream_t)stream;
    const size_t nb = ((size_t)N * 4 + 255) & ~(size_t)255;
    int* flag = (int*)ws;
    int* excl = (int*)((char*)ws + nb);
    void* tmp = (char*)ws + 2 * nb;
    k_s1_flags<<<cdiv(N, kBlock), kBlock, 0, st>>>((const float4*)rast, mask, N, flag);
    FMHR_CHECK_LAUNCH("k_s1_flags");
    int rc = scan_exclusive(flag, excl, (size_t)N, tmp, ws_bytes - 2 * nb, st);
    if (rc) return rc;
    k_s1_view_rows<<<1, 64, 0, st>>>(flag, excl, n, P, view_row0, view_valid);
    FMHR_CHECK_LAUNCH("k_s1_view_rows");
    hipError_t e = hipMemsetAsync(row_flags, 0, (size_t)row_flags_len * 4, st);
    if (e != hipSuccess) return hip_fail(e, "stage1_rows: memset");
    k_s1_rowid<<<cdiv(N, kBlock), kBlock, 0, st>>>(flag, excl, view_row0, N, P, rowid, row_flags);
    FMHR_CHECK_LAUNCH("k_s1_rowid");
    return FMHR_OK;
}

// ---- plan, step 2: entries per row -> row_ptr [rows + 1].  ws = [cnt rows+1 ints][scan scratch]
extern "C" size_t fmhr_stage1_row_ptr_workspace_bytes(int rows)
{
    return ((((size_t)rows + 1) * 4 + 255) & ~(size_t)255) + scan_tmp_bytes((size_t)rows + 1);
}

extern "C" int fmhr_stage1_row_ptr(const float* rast, const int32_t* rowid, const int32_t* items, const float* alphas, int n_items, int n,
                                   int H, int W, int rows, int32_t* row_flags, int32_t* row_ptr, void* ws, size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(n >= 1 && H >= 1 && W >= 1 && rows >= 0 && n_items >= 0, "stage1_row_ptr: bad dims");
    FMHR_CHECK_ARG(rast && rowid && row_flags && row_ptr && ws && (n_items == 0 || (items && alphas)), "stage1_row_ptr: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_stage1_row_ptr_workspace_bytes(rows), "stage1_row_ptr: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (n_items > 0) {
        k_s1_item_flags<<<cdiv(n_items, kBlock), kBlock, 0, st>>>((const int4*)items, alphas, n_items, (const float4*)rast, rowid, n, H, W, row_flags);
        FMHR_CHECK_LAUNCH("k_s1_item_flags");
    }
    int* cnt = (int*)ws;
    const size_t nb = (((size_t)rows + 1) * 4 + 255) & ~(size_t)255;
    k_s1_row_count<<<cdiv((long long)rows + 1, kBlock), kBlock, 0, st>>>(row_flags, rows, cnt);
    FMHR_CHECK_LAUNCH("k_s1_row_count");
    return scan_exclusive(cnt, row_ptr, (size_t)rows + 1, (char*)ws + nb, ws_bytes - nb, st);
}

// ---- plan, step 3: the rows' entries and constants
extern "C" int fmhr_stage1_fill(const float* rast, const int32_t* tri, const int32_t* rowid, const int32_t* items, const float* alphas,
                                int n_items, const float* normals, const float* img, int n, int V, int F, int H, int W, int rows,
                                const int32_t* row_flags, const int32_t* row_ptr, int32_t* ent_col, float* ent_w, float* row_n,
                                float* row_img, void* stream)
{
    FMHR_CHECK_ARG(n >= 1 && V >= 1 && F >= 1 && H >= 1 && W >= 1 && rows >= 0 && n_items >= 0, "stage1_fill: bad dims");
    FMHR_CHECK_ARG(rast && tri && rowid && normals && img && row_flags && row_ptr && ent_col && ent_w && row_n && row_img &&
                       (n_items == 0 || (items && alphas)), "stage1_fill: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const long long N = (long long)n * H * W;
    hipError_t e = hipMemsetAsync(row_n, 0, (size_t)rows * 12, st);
    if (e == hipSuccess) e = hipMemsetAsync(row_img, 0, (size_t)rows * 12, st);
    if (e != hipSuccess) return hip_fail(e, "stage1_fill: memset");
    k_s1_fill_rows<<<cdiv(N, kBlock), kBlock, 0, st>>>((const float4*)rast, tri, rowid, normals, img, N, V, F, row_ptr, ent_col, ent_w, row_n, row_img);
    FMHR_CHECK_LAUNCH("k_s1_fill_rows");
    if (n_items > 0) {
        k_s1_fill_items<<<cdiv(n_items, kBlock), kBlock, 0, st>>>((const int4*)items, alphas, n_items, (const float4*)rast, tri, rowid, n, V, F, H, W,
                                                                  row_flags, row_ptr, ent_col, ent_w);
        FMHR_CHECK_LAUNCH("k_s1_fill_items");
    }
    return FMHR_OK;
}

// ---- plan, step 4: the transpose, rows keyed (vertex, view): t_ptr [V * n + 1], t_row / t_w [entries].  ws = [cnt n*V+1 ints][scan scratch]
extern "C" size_t fmhr_stage1_transpose_workspace_bytes(int n, int V)
{
    const size_t c = (size_t)n * V + 1;
    return ((c * 4 + 255) & ~(size_t)255) + scan_tmp_bytes(c);
}

extern "C" int fmhr_stage1_transpose(const int32_t* row_ptr, const int32_t* ent_col, const float* ent_w, const int32_t* view_row0, int n, int V,
                                     int rows, int32_t* t_ptr, int32_t* t_row, float* t_w, void* ws, size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(n >= 1 && V >= 1 && rows >= 0 && (long long)n * V < (1ll << 31) - 1, "stage1_transpose: bad dims");
    FMHR_CHECK_ARG(row_ptr && ent_col && ent_w && view_row0 && t_ptr && t_row && t_w && ws, "stage1_transpose: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_stage1_transpose_workspace_bytes(n, V), "stage1_transpose: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t c = (size_t)n * V + 1, nb = (c * 4 + 255) & ~(size_t)255;
    int* cnt = (int*)ws;
    hipError_t e = hipMemsetAsync(cnt, 0, c * 4, st);
    if (e != hipSuccess) return hip_fail(e, "stage1_transpose: memset");
    if (rows > 0) {
        k_s1_transpose<false><<<cdiv(rows, kBlock), kBlock, 0, st>>>(row_ptr, ent_col, ent_w, view_row0, n, V, rows, cnt, nullptr, nullptr, nullptr);
        FMHR_CHECK_LAUNCH("k_s1_transpose<count>");
    }
    int rc = scan_exclusive(cnt, t_ptr, c, (char*)ws + nb, ws_bytes - nb, st);
    if (rc) return rc;
    e = hipMemsetAsync(cnt, 0, c * 4, st);
    if (e != hipSuccess) return hip_fail(e, "stage1_transpose: memset");
    if (rows > 0) {
        k_s1_transpose<true><<<cdiv(rows, kBlock), kBlock, 0, st>>>(row_ptr, ent_col, ent_w, view_row0, n, V, rows, cnt, t_ptr, t_row, t_w);
        FMHR_CHECK_LAUNCH("k_s1_transpose<fill>");
        k_s1_tsort<<<cdiv((long long)n * V, kBlock), kBlock, 0, st>>>(t_ptr, (long long)n * V, t_row, t_w);
        FMHR_CHECK_LAUNCH("k_s1_tsort");
    }
    return FMHR_OK;
}

// ---- the step.  ws = [R rows x 3 floats][part blocks x 10 floats]
extern "C" size_t fmhr_stage1_step_workspace_bytes(const fmhr_stage1_plan* p)
{
    if (p == nullptr || p->rows < 0) return 0;
    return (((size_t)p->rows * 12 + 255) & ~(size_t)255) + (size_t)(p->rows / kBlock) * 40 + 256;
}

extern "C" int fmhr_stage1_step(const fmhr_stage1_plan* p, const float* albedo, const float* sh, const int32_t* view_sel, float sfs_weight,
                                float* galbedo, float* gsh, float* loss_view, const fmhr_adam_param* adam_albedo,
                                const fmhr_adam_param* adam_sh, void* ws, size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(p != nullptr, "stage1_step: plan is null");
    FMHR_CHECK_ARG(p->n >= 1 && p->V >= 1 && p->rows >= 0 && p->rows % kBlock == 0, "stage1_step: bad plan n=%d V=%d rows=%d", p->n, p->V, p->rows);
    FMHR_CHECK_ARG(p->view_row0 && p->view_valid && p->row_ptr && p->ent_col && p->ent_w && p->row_n && p->row_img && p->t_ptr && p->t_row &&
                       p->t_w, "stage1_step: null table in the plan");
    FMHR_CHECK_ARG(albedo && sh && galbedo && gsh && loss_view && ws, "stage1_step: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_stage1_step_workspace_bytes(p), "stage1_step: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    S1Plan P;
    P.n = p->n; P.V = p->V; P.rows = p->rows; P.blocks = p->rows / kBlock;
    P.view_row0 = p->view_row0; P.view_valid = p->view_valid; P.row_ptr = p->row_ptr; P.ent_col = p->ent_col; P.ent_w = p->ent_w;
    P.row_n = p->row_n; P.row_img = p->row_img; P.t_ptr = p->t_ptr; P.t_row = p->t_row; P.t_w = p->t_w;
    float* R = (float*)ws;
    float* part = (float*)((char*)ws + (((size_t)p->rows * 12 + 255) & ~(size_t)255));
    AdamTensor ad_a, ad_s;
    int rc = adam_tensor_from(ad_a, adam_albedo, 3ll * p->V);
    if (rc) return rc;
    rc = adam_tensor_from(ad_s, adam_sh, 9ll * p->n);
    if (rc) return rc;
    if (P.blocks > 0) {
        k_s1_fwd<<<P.blocks, kBlock, 0, st>>>(P, albedo, sh, view_sel, sfs_weight, R, part);
        FMHR_CHECK_LAUNCH("k_s1_fwd");
    }
    const int vblocks = cdiv((long long)p->V * kS1Sub, kBlock);
    k_s1_bwd<<<vblocks + p->n, kBlock, 0, st>>>(P, R, part, view_sel, sfs_weight, galbedo, gsh, loss_view, ad_a, ad_s, vblocks);
    FMHR_CHECK_LAUNCH("k_s1_bwd");
    return FMHR_OK;
}

// ---- SH least-squares initialisation: per-workgroup partial sums, part [n, nblk, 55] doubles (45 upper-triangle entries of
// sum phi phi^T row by row, 9 entries of sum phi gray, the valid-pixel count); the caller adds the nblk partials and solves 9 x 9
extern "C" int fmhr_sh_normal_equations(const float* normals, const float* rast, const float* mask, const float* gray, int n, int H, int W,
                                        int nblk, double* part, void* stream)
{
    FMHR_CHECK_ARG(n >= 1 && H >= 1 && W >= 1 && nblk >= 1 && n <= 65535, "sh_normal_equations: bad dims");
    FMHR_CHECK_ARG(normals && rast && mask && gray && part, "sh_normal_equations: null pointer");
    k_sh_normal_eq<<<dim3(nblk, n), kBlock, 0, (hipStream_t)stream>>>(normals, (const float4*)rast, mask, gray, (long long)H * W, part);
    FMHR_CHECK_LAUNCH("k_sh_normal_eq");
    return FMHR_OK;
}

// part [n, nblk, 4] doubles: sums of img / radiance(sh, normal) per channel over the valid pixels, and their count (sh: 9 floats)
extern "C" int fmhr_sh_albedo_sums(const float* normals, const float* rast, const float* mask, const float* img, const float* sh, int n,
                                   int H, int W, int nblk, double* part, void* stream)
{
    FMHR_CHECK_ARG(n >= 1 && H >= 1 && W >= 1 && nblk >= 1 && n <= 65535, "sh_albedo_sums: bad dims");
    FMHR_CHECK_ARG(normals && rast && mask && img && sh && part, "sh_albedo_sums: null pointer");
    k_sh_albedo_sums<<<dim3(nblk, n), kBlock, 0, (hipStream_t)stream>>>(normals, (const float4*)rast, mask, img, sh, (long long)H * W, part);
    FMHR_CHECK_LAUNCH("k_sh_albedo_sums");
    return FMHR_OK;
}

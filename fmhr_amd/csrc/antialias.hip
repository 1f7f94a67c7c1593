// fmhr_amd/csrc/antialias.hip — mesh topology tables and silhouette antialiasing, forward + backward, for gfx950.
// Replaces dr.antialias of nvdiffrast as called at /root/reference/mesh_sfs_optim.py:274,287 (and its per-call edge hash).
//
// Forward: one lane per pixel looks at its right and lower neighbour; a pair with different triangle ids is analysed on the
// spot and, if a silhouette edge crosses between the two centres, blended with float atomics and
// appended to the work queue by wave64 ballot + popcount prefix (one queue atomic per wave).  Only pairs that actually
// blend are queued, so the queue length is proportional to the silhouette length, not to the number of triangle seams.
#include "common.h"
#include "shade_common.h"
#include <float.h>
#include <string.h>

namespace fmhr {

// ------------------------------------------------------------------------------------------------ edge hash (topology)
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kNoCode = 0x7fffffff;

struct EdgeSlot { unsigned long long key; int c0; int c1; };

__device__ __forceinline__ unsigned hash_u64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (unsigned)k;
}

__device__ __forceinline__ unsigned long long edge_key(const int32_t* tri, int code)
{
    int f = code / 3, k = code - f * 3;
    int a = tri[f * 3 + (k + 1) % 3], b = tri[f * 3 + (k + 2) % 3];
    unsigned lo = (unsigned)min(a, b), hi = (unsigned)max(a, b);
    return ((unsigned long long)lo << 32) | hi;
}

__global__ void k_hash_clear(EdgeSlot* tab, unsigned T)
{
    unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < T) { tab[i].key = kEmptyKey; tab[i].c0 = kNoCode; tab[i].c1 = kNoCode; }
}

__global__ void k_hash_insert(const int32_t* __restrict__ tri, int F, EdgeSlot* tab, unsigned mask)
{
    int code = blockIdx.x * kBlock + threadIdx.x;
    if (code >= 3 * F) return;
    unsigned long long key = edge_key(tri, code);
    unsigned h = hash_u64(key) & mask;
    for (unsigned probe = 0; probe <= mask; probe++) {
        unsigned long long old = atomicCAS(&tab[h].key, kEmptyKey, key);
        if (old == kEmptyKey || old == key) { atomicMin(&tab[h].c0, code); return; }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ int hash_find(const EdgeSlot* tab, unsigned mask, unsigned long long key)
{
    unsigned h = hash_u64(key) & mask;
    for (unsigned probe = 0; probe <= mask; probe++) {
        unsigned long long k = tab[h].key;
        if (k == key) return (int)h;
        if (k == kEmptyKey) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

__global__ void k_hash_second(const int32_t* __restrict__ tri, int F, EdgeSlot* tab, unsigned mask)
{
    int code = blockIdx.x * kBlock + threadIdx.x;
    if (code >= 3 * F) return;
    int s = hash_find(tab, mask, edge_key(tri, code));
    if (s >= 0 && tab[s].c0 != code) atomicMin(&tab[s].c1, code);
}

__global__ void k_hash_opp(const int32_t* __restrict__ tri, int F, const EdgeSlot* tab, unsigned mask, int32_t* __restrict__ opp)
{
    int code = blockIdx.x * kBlock + threadIdx.x;
    if (code >= 3 * F) return;
    int s = hash_find(tab, mask, edge_key(tri, code));
    int r = -1;
    if (s >= 0) {
        int a = tab[s].c0, b = tab[s].c1;
        int oa = tri[a], ob = (b == kNoCode) ? -1 : tri[b];
        r = (oa == tri[code]) ? ob : oa;
    }
    opp[code] = r;
}

// ------------------------------------------------------------------------------------------------ adjacency CSR
__global__ void k_adj_degree(const EdgeSlot* tab, unsigned T, int V, int* deg)
{
    unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T || tab[i].key == kEmptyKey) return;
    int lo = (int)(tab[i].key >> 32), hi = (int)(tab[i].key & 0xffffffffu);
    if (lo == hi || lo >= V || hi >= V) return;
    atomicAdd(&deg[lo], 1);
    atomicAdd(&deg[hi], 1);
}

// single-workgroup exclusive scan (V is at most a few 1e5; runs once per mesh)
__global__ void __launch_bounds__(1024) k_scan_exclusive(const int* __restrict__ in, int n, int* __restrict__ out)
{
    __shared__ int sm[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < n) ? in[i] : 0;
        sm[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int t = (threadIdx.x >= o) ? sm[threadIdx.x - o] : 0;
            __syncthreads();
            sm[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) out[i] = carry + sm[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += sm[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

__global__ void k_adj_fill(const EdgeSlot* tab, unsigned T, int V, const int* __restrict__ row_ptr, int* cursor, int* col)
{
    unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T || tab[i].key == kEmptyKey) return;
    int lo = (int)(tab[i].key >> 32), hi = (int)(tab[i].key & 0xffffffffu);
    if (lo == hi || lo >= V || hi >= V) return;
    col[row_ptr[lo] + atomicAdd(&cursor[lo], 1)] = hi;
    col[row_ptr[hi] + atomicAdd(&cursor[hi], 1)] = lo;
}

__global__ void k_adj_sort_rows(int V, const int* __restrict__ row_ptr, int* col)
{
    int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= V) return;
    int a = row_ptr[v], b = row_ptr[v + 1];
    for (int i = a + 1; i < b; i++) {
        int x = col[i], j = i - 1;
        while (j >= a && col[j] > x) { col[j + 1] = col[j]; j--; }
        col[j + 1] = x;
    }
}

// vertex -> incident face corners (code = 3*face + corner), CSR with sorted rows: the per-vertex kernels of the fused step walk it
__global__ void k_vf_count(const int32_t* __restrict__ tri, int F, int V, int* deg)
{
    int code = blockIdx.x * kBlock + threadIdx.x;
    if (code >= 3 * F) return;
    int v = tri[code];
    if ((unsigned)v < (unsigned)V) atomicAdd(&deg[v], 1);
}

__global__ void k_vf_fill(const int32_t* __restrict__ tri, int F, int V, const int* __restrict__ row_ptr, int* cursor, int* idx)
{
    int code = blockIdx.x * kBlock + threadIdx.x;
    if (code >= 3 * F) return;
    int v = tri[code];
    if ((unsigned)v < (unsigned)V) idx[row_ptr[v] + atomicAdd(&cursor[v], 1)] = code;
}

// inverse of vf_idx: slot[3*face + corner] = position of that corner in its vertex's CSR row (the face-parallel backward of the
// fused step stores each corner's gradient at its slot; the mesh tail then sums a vertex's slots with contiguous loads)
__global__ void __launch_bounds__(kBlock) k_corner_slots(const int32_t* __restrict__ vf_idx, int n3, int32_t* __restrict__ slot)
{
    int e = blockIdx.x * kBlock + threadIdx.x;
    if (e < n3) slot[vf_idx[e]] = e;
}

// ------------------------------------------------------------------------------------------------ antialias analysis
constexpr int AA_DOWN = 4, AA_TRI1 = 8, AA_FOUND = 16, AA_CLAMPED = 32;

__device__ __forceinline__ bool same_sign(float a, float b) { return (int)(__float_as_uint(a) ^ __float_as_uint(b)) >= 0; }

__device__ __forceinline__ bool rational_gt(float n0, float n1, float d0, float d1)
{
    float dd = d0 * d1;
    float l = n0 * d1;
    float r = n1 * d0;
    if (dd < 0.0f) return l < r;
    return l > r;
}

__device__ __forceinline__ int max_idx3(float n0, float n1, float n2, float d0, float d1, float d2)
{
    bool g10 = rational_gt(n1, n0, d1, d0);
    bool g20 = rational_gt(n2, n0, d2, d0);
    bool g21 = rational_gt(n2, n1, d2, d1);
    if (g20 && g21) return 2;
    if (g10) return 1;
    return 0;
}

// rule A3: silhouette edges of a triangle in a view, decided on the snapped integer coordinates.
// bit k = edge k (opposite vertex k) is a silhouette edge; 0 when a vertex of the triangle itself is rejected.
__device__ __forceinline__ int silhouette_bits_snapped(int2 s0, int2 s1, int2 s2, int2 o0, int2 o1, int2 o2)
{
    const int REJ = (int)0x80000000;
    if (s0.x == REJ || s1.x == REJ || s2.x == REJ) return 0;
    if (o0.x == REJ) o0 = s0;
    if (o1.x == REJ) o1 = s1;
    if (o2.x == REJ) o2 = s2;
    long long bb = (long long)(s1.x - s0.x) * (s2.y - s0.y) - (long long)(s2.x - s0.x) * (s1.y - s0.y);
    long long a0 = (long long)(s1.x - o0.x) * (s2.y - o0.y) - (long long)(s2.x - o0.x) * (s1.y - o0.y);
    long long a1 = (long long)(s2.x - o1.x) * (s0.y - o1.y) - (long long)(s0.x - o1.x) * (s2.y - o1.y);
    long long a2 = (long long)(s0.x - o2.x) * (s1.y - o2.y) - (long long)(s1.x - o2.x) * (s0.y - o2.y);
    int bits = 0;
    if ((a0 < 0) == (bb < 0)) bits |= 1;
    if ((a1 < 0) == (bb < 0)) bits |= 2;
    if ((a2 < 0) == (bb < 0)) bits |= 4;
    return bits;
}

__device__ __forceinline__ int2 snap_or_reject(const ClipView& cv, int v, float kx, float ky)
{
    if (v < 0) return make_int2((int)0x80000000, 0);
    int x, y;
    bool ok = snap_vertex(cv.get(v), kx, ky, x, y);
    return ok ? make_int2(x, y) : make_int2((int)0x80000000, 0);
}

// The analysis proper: the nearer triangle's three clip positions, its silhouette bits and which of the pair's two pixels it owns
// (front0) -> work-item code + alpha.  ONE body for every caller (generic op, fused step, staged fused step): statement-for-statement
// fp32, no contraction.
__device__ __forceinline__ int aa_analyse_core(const float4 p0, const float4 p1, const float4 p2, int sb, bool front0, int vi0, int vi1, int vi2,
                                               int qx, int qy, int d, int H, int W, float& alpha_out, int* edge_va, int* edge_vb)
{
    bool s0 = sb & 1, s1 = sb & 2, s2 = sb & 4;
    float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
    float w0 = 1.0f / p0.w, w1 = 1.0f / p1.w, w2 = 1.0f / p2.w;
    float fx = (float)qx + 0.5f; fx = fx - xh;
    float fy = (float)qy + 0.5f; fy = fy - yh;
    float m;
    m = p0.x * w0; m = m * xh; float x0 = m - fx;    m = p0.y * w0; m = m * yh; float y0 = m - fy;
    m = p1.x * w1; m = m * xh; float x1 = m - fx;    m = p1.y * w1; m = m * yh; float y1 = m - fy;
    m = p2.x * w2; m = m * xh; float x2 = m - fx;    m = p2.y * w2; m = m * yh; float y2 = m - fy;
    float l, r;
    if (d) { float tt; tt = x0; x0 = y0; y0 = tt; tt = x1; x1 = y1; y1 = tt; tt = x2; x2 = y2; y2 = tt; }
    float dx0 = x2 - x1, dx1 = x0 - x2, dx2 = x1 - x0;
    float dy0 = y2 - y1, dy1 = y0 - y2, dy2 = y1 - y0;
    float ds = front0 ? 1.0f : -1.0f;
    l = x1 * dy0; r = y1 * dx0; float d0 = l - r; d0 = ds * d0;
    l = x2 * dy1; r = y2 * dx1; float d1 = l - r; d1 = ds * d1;
    l = x0 * dy2; r = y0 * dx2; float d2 = l - r; d2 = ds * d2;
    const float NEG = -FLT_MAX;
    if (same_sign(y1, y2)) { d0 = NEG; dy0 = 1.0f; }
    if (same_sign(y2, y0)) { d1 = NEG; dy1 = 1.0f; }
    if (same_sign(y0, y1)) { d2 = NEG; dy2 = 1.0f; }
    int di = max_idx3(d0, d1, d2, dy0, dy1, dy2);
    float dc = NEG;
    if (di == 0 && s0 && fabsf(dy0) >= fabsf(dx0)) dc = d0 / dy0;
    if (di == 1 && s1 && fabsf(dy1) >= fabsf(dx1)) dc = d1 / dy1;
    if (di == 2 && s2 && fabsf(dy2) >= fabsf(dx2)) dc = d2 / dy2;
    const float eps = 0.0625f;
    if (!(dc > -eps && dc < 1.0f + eps)) return 0;
    int code = di | (d ? AA_DOWN : 0) | (front0 ? 0 : AA_TRI1) | AA_FOUND;
    if (dc < 0.0f) { dc = 0.0f; code |= AA_CLAMPED; }
    if (dc > 1.0f) { dc = 1.0f; code |= AA_CLAMPED; }
    float al = 0.5f - dc;
    al = ds * al;
    alpha_out = al;
    if (edge_va != nullptr) {                    // the two vertices of the crossing edge (the backward's position gradient goes to them)
        *edge_va = (di == 0) ? vi1 : ((di == 1) ? vi2 : vi0);
        *edge_vb = (di == 0) ? vi2 : ((di == 1) ? vi0 : vi1);
    }
    return code;
}

// Analyse the pixel pair ((px,py), (px+1-d, py+d)) of view b.  z0/id0 and z1/id1 are rast.z / rast.w of the two pixels.
// `silh` (nullable) = this view's per-triangle silhouette bits precomputed by the rasterizer: most seams between
// micro-triangles are not silhouettes and leave here after one byte load.
// Returns the work-item code (0 = no blend) and alpha.  Statement-for-statement fp32 (no contraction).
__device__ __forceinline__ int aa_analyse(const ClipView& cv, const int32_t* __restrict__ tri,
                                          const int32_t* __restrict__ opp, const unsigned char* __restrict__ silh,
                                          int V, int F, int H, int W,
                                          int px, int py, int d, float z0, float id0, float z1, float id1, float& alpha_out,
                                          int* edge_va = nullptr, int* edge_vb = nullptr)
{
    int tri0 = (int)id0 - 1, tri1 = (int)id1 - 1;
    int t = (tri0 >= 0) ? tri0 : tri1;
    if (tri0 >= 0 && tri1 >= 0) t = (z0 < z1) ? tri0 : tri1;
    int qx = px, qy = py;
    if (t == tri1) { qx += 1 - d; qy += d; }
    if (t < 0 || t >= F) return 0;
    int sb;
    if (silh != nullptr) {
        sb = silh[t];
        if (!sb) return 0;
    }
    int vi0 = tri[t * 3], vi1 = tri[t * 3 + 1], vi2 = tri[t * 3 + 2];
    if ((unsigned)vi0 >= (unsigned)V || (unsigned)vi1 >= (unsigned)V || (unsigned)vi2 >= (unsigned)V) return 0;
    if (silh == nullptr) {
        const float kx = (float)W * 128.0f, ky = (float)H * 128.0f;
        int op0 = opp[t * 3], op1 = opp[t * 3 + 1], op2 = opp[t * 3 + 2];
        sb = silhouette_bits_snapped(snap_or_reject(cv, vi0, kx, ky), snap_or_reject(cv, vi1, kx, ky),
                                     snap_or_reject(cv, vi2, kx, ky), snap_or_reject(cv, op0, kx, ky),
                                     snap_or_reject(cv, op1, kx, ky), snap_or_reject(cv, op2, kx, ky));
        if (!sb) return 0;
    }
    const float4 p0 = cv.get(vi0), p1 = cv.get(vi1), p2 = cv.get(vi2);
    return aa_analyse_core(p0, p1, p2, sb, t == tri0, vi0, vi1, vi2, qx, qy, d, H, W, alpha_out, edge_va, edge_vb);
}

// append one item per flagged lane with a single queue atomic per wave
__device__ __forceinline__ void queue_push(bool has, int4 item, float alpha, int32_t* counter, int cap,
                                           int4* __restrict__ items, float* __restrict__ alphas)
{
    unsigned long long bal = __ballot(has);
    if (bal == 0) return;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)bal) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(bal));
    base = __shfl(base, leader, 64);
    if (has) {
        int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (slot < cap) { items[slot] = item; alphas[slot] = alpha; }
        else counter[1] = 1;
    }
}

// forward for one pixel: its right and lower pair.  `live` = the pixel exists (lanes of a partial last wave are not)
template <int CA, int CB>
__device__ __forceinline__ void aa_fwd_pixel(long long gid, bool live, const float* __restrict__ colorA,
         const float* __restrict__ colorB, const float4* __restrict__ rast,
         const float* __restrict__ pos, const int32_t* __restrict__ tri, const int32_t* __restrict__ opp,
         const unsigned char* __restrict__ silh, const unsigned long long* __restrict__ covw,
         int n, int V, int F, int H, int W, int Cdyn, float* __restrict__ outA, float* __restrict__ outB,
         int4* __restrict__ items, float* __restrict__ alphas, int32_t* counter, int cap)
{
    long long P = (long long)H * W;
    int b = 0, px = 0, py = 0;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int rem;
        b = fast_divmod(gid, (int)P, rem);
        py = fast_divmod(rem, W, px);
        // in sparse mode rast is only defined on covered pixels: an uncovered pixel is the all-zero fragment
        if (covw == nullptr || ((covw[gid >> 6] >> (gid & 63)) & 1ull)) r0 = rast[gid];
    }
    const ClipView cv = clip_view(pos, nullptr, nullptr, b, V);
#pragma unroll
    for (int d = 0; d < 2; d++) {
        bool has = false;
        float alpha = 0.f;
        int code = 0;
        long long gid1 = gid + (d ? W : 1);
        if (live && (d ? (py + 1 < H) : (px + 1 < W))) {
            float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (covw == nullptr || ((covw[gid1 >> 6] >> (gid1 & 63)) & 1ull)) r1 = rast[gid1];
            if (r1.w != r0.w) {
                code = aa_analyse(cv, tri, opp, silh ? silh + (size_t)b * F : nullptr, V, F, H, W, px, py, d, r0.z, r0.w, r1.z, r1.w, alpha);
                has = code != 0;
            }
        }
        if (has) {
            long long tgt = (alpha > 0.f) ? gid : gid1;
            const int C = (CA > 0) ? CA : Cdyn;
#pragma unroll
            for (int c = 0; c < C; c++) {
                float dcol = colorA[gid1 * C + c] - colorA[gid * C + c];
                if (dcol != 0.f) atomicAdd(outA + tgt * C + c, alpha * dcol);
            }
            if (CB > 0) {
#pragma unroll
                for (int c = 0; c < CB; c++) {
                    float dcol = colorB[gid1 * CB + c] - colorB[gid * CB + c];
                    if (dcol != 0.f) atomicAdd(outB + tgt * CB + c, alpha * dcol);
                }
            }
        }
        queue_push(has, make_int4(px, py, b, code), alpha, counter, cap, items, alphas);
    }
}

// generic op: one lane per pixel of the dense image
template <int CA, int CB>
__global__ void __launch_bounds__(kBlock)
k_aa_fwd(const float* __restrict__ colorA, const float* __restrict__ colorB, const float4* __restrict__ rast,
         const float* __restrict__ pos, const int32_t* __restrict__ tri, const int32_t* __restrict__ opp,
         int n, int V, int F, int H, int W, int Cdyn, float* __restrict__ outA, float* __restrict__ outB,
         int4* __restrict__ items, float* __restrict__ alphas, int32_t* counter, int cap)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    aa_fwd_pixel<CA, CB>(gid, gid < (long long)n * H * W, colorA, colorB, rast, pos, tri, opp, nullptr, nullptr, n, V, F, H, W, Cdyn,
                         outA, outB, items, alphas, counter, cap);
}

// fused step: persistent waves over the compacted list of active 64-pixel segments, rgba colour, work items into the
// workgroup's region of the list.
// Per segment: the coverage words of the own / next / lower segments are wave-uniform loads; then every lane issues its two
// rast loads (own, lower) together — the right neighbour's fragment is the next lane's register — and only pairs with different
// triangle ids go on (silhouette byte first).  Colours are read as 16-B words only by the pairs that blend.
// Workgroups of the fused antialias kernels = regions of the work-item list: ONE resident round.  Row-aligned segments (W % 64 == 0:
// the mask scan of k_aa_fwd_list): 768 workgroups at 3 waves per SIMD — the kernel needs ~137 VGPRs; at 1280 / 5 (96 VGPRs) it spilled
// 37-59 registers into its latency-bound phases: 35 vs 23.5 us; 1024 / 4: 27 us; 512 / 2: 26 us.  The per-pixel scan of the other
// case is the faster the more waves share its load latency: 1280 / 5 (36 us; 1024 / 4: 37, 768 / 3: 42).
#ifndef FMHR_AA_GRID_ROW
#define FMHR_AA_GRID_ROW 768
#endif
constexpr int kAaGridRow = FMHR_AA_GRID_ROW, kAaGridAny = 1280;
static bool aa_row_aligned(int n, int H, int W) { return (W & 63) == 0 && ((long long)n * H * W) < (1ll << 36); }   // segment ids below 2^30
static int aa_grid(bool row) { return row ? kAaGridRow : kAaGridAny; }
                                              // of k_aa_fwd_list at its 5 waves per SIMD (96 VGPRs; 2048 workgroups ran as 1.6 rounds: 40.5 -> 32.7 us)
constexpr int kCandBuf = 512;                 // candidate pairs one workgroup parks before analysing them densely
constexpr int kItemBuf = 512;                 // work items one workgroup of k_aa_fwd_list buffers in LDS

__device__ __forceinline__ unsigned long long covw_or0(const unsigned long long* __restrict__ covw, long long w, long long nSeg)
{
    return (w >= 0 && w < nSeg) ? covw[w] : 0ull;
}

// The fused step keeps no shaded image (53 MB of stores per step in the resolve pass, read back here for a few hundred thousand
// pixels): the colours of the two pixels of a pair that blends are re-shaded from their per-pixel words (shade_common.h:
// shade_word, bit-identical to the resolve pass), and the pair's colour difference is stored with its work item for the backward.
struct AaShade { const float4* words; const float4* attr8; const float* sh; float4* dcols; };

// one candidate pair of the fused forward: full analysis, blend of the rgba, work item for the backward
__device__ __forceinline__ void aa_fwd_pair(int px, int py, int b, int d, float z0, float id0, float z1, float id1,
                                            const AaShade& sd, const float* __restrict__ pos,
                                            const float* __restrict__ verts, const float4* __restrict__ cmat,
                                            const int32_t* __restrict__ tri, const int32_t* __restrict__ opp,
                                            const unsigned char* __restrict__ silh, int V, int F, int H, int W, int Wp,
                                            float* __restrict__ out, int4* __restrict__ items, float* __restrict__ alphas,
                                            int32_t* __restrict__ counter, int4* ibuf, float* abuf, int* inum, int region, size_t rbase)
{
    float alpha = 0.f;
    int va = 0, vb = 0;
    const int code = aa_analyse(clip_view(pos, verts, cmat, b, V), tri, opp, silh + (size_t)b * F, V, F, H, W, px, py, d, z0, id0, z1, id1, alpha,
                                &va, &vb);
    if (code == 0) return;
    // work item of the fused step: (px | py << 16, view | code << 16, edge vertex a, edge vertex b) — the backward starts its vertex
    // loads straight from the item instead of walking pixel word -> face -> tri (two dependent round trips of a latency-bound kernel)
    const int4 item = make_int4(px | (py << 16), b | (code << 16), va, vb);
    const long long gid = ((long long)b * H + py) * Wp + px, gid1 = gid + (d ? Wp : 1);    // Wp: row pitch of the per-pixel buffers
    const long long tgt = (alpha > 0.f) ? gid : gid1;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);           // an uncovered pixel (id 0) has the all-zero colour
    const float4 ca = (id0 != 0.f) ? shade_word(sd.words[gid], tri, sd.attr8, sd.sh + (size_t)b * 9) : zero;
    const float4 cb = (id1 != 0.f) ? shade_word(sd.words[gid1], tri, sd.attr8, sd.sh + (size_t)b * 9) : zero;
    const float dc[4] = {cb.x - ca.x, cb.y - ca.y, cb.z - ca.z, cb.w - ca.w};
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (dc[c] != 0.f) atomicAdd(out + tgt * 4 + c, alpha * dc[c]);
    const int slot = atomicAdd(inum, 1);                             // LDS
    if (slot < region) sd.dcols[rbase + slot] = make_float4(dc[0], dc[1], dc[2], dc[3]);   // straight to the item's final place
    if (slot < kItemBuf) {
        ibuf[slot] = item;
        abuf[slot] = alpha;
    } else if (slot < region) {                                      // buffer full (a silhouette-dense workgroup): straight to the region
        items[rbase + slot] = item;
        alphas[rbase + slot] = alpha;
    } else {
        counter[1] = 1; counter[3] = 1;                              // region overflow: the pair gets no gradient; [1] is rearmed every
                                                                     // step, [3] stays set until fmhr_ham_workspace_init (fused step: 4 words)
    }
}

// aa_fwd_pair in THREE round trips instead of six, for the dense candidate pass of k_aa_fwd_list (a workgroup's ~200 candidates, one
// per lane: the pass IS one lane's chain — 9.4 of the kernel's 28 us, tools/analysis/wave_timeline.py): (1) both pixels' words, both
// faces' vertex ids, the view's matrix rows and SH row, all addressed by the candidate itself — the nearer face's silhouette bits
// ride in its pixel's word (the rasterizer's own value: no byte gather); (2) the nearer face's three vertices -> aa_analyse_core;
// (3) for the pairs that blend, the twelve attribute words of both pixels' faces -> the two colours.  Same statements as aa_fwd_pair
// (aa_analyse_core, shade_eval_attr): same items, alphas and colours.
__device__ __forceinline__ void aa_fwd_pair_staged(int px, int py, int b, int d, float z0, float id0, float z1, float id1,
                                                   const AaShade& sd, const float* __restrict__ verts, const float4* __restrict__ cmat,
                                                   const int32_t* __restrict__ tri, int V, int F, int H, int W, int Wp,
                                                   float* __restrict__ out, int4* __restrict__ items, float* __restrict__ alphas,
                                                   int32_t* __restrict__ counter, int4* ibuf, float* abuf, int* inum, int region, size_t rbase)
{
    struct I3 { int x, y, z; };
    const int tri0 = (int)id0 - 1, tri1 = (int)id1 - 1;
    int t = (tri0 >= 0) ? tri0 : tri1;
    if (tri0 >= 0 && tri1 >= 0) t = (z0 < z1) ? tri0 : tri1;
    if (t < 0 || t >= F) return;
    const bool front0 = t == tri0;
    int qx = px, qy = py;
    if (!front0) { qx += 1 - d; qy += d; }
    const long long gid = ((long long)b * H + py) * Wp + px, gid1 = gid + (d ? Wp : 1);    // Wp: row pitch of the per-pixel buffers
    // ---- (1)
    float4 wa = sd.words[gid], wb = sd.words[gid1];                 // (an uncovered pixel's word is stale: masked by its id below)
    const I3 ra = *reinterpret_cast<const I3*>(tri + (size_t)max(tri0, 0) * 3), rb = *reinterpret_cast<const I3*>(tri + (size_t)max(tri1, 0) * 3);
    float4 c0 = cmat[(size_t)b * 4], c1 = cmat[(size_t)b * 4 + 1], c2 = cmat[(size_t)b * 4 + 2], c3 = cmat[(size_t)b * 4 + 3];
    const float* const shv = sd.sh + (size_t)b * 9;
    const F4U sA = reinterpret_cast<const F4U*>(shv)[0], sB = reinterpret_cast<const F4U*>(shv)[1];
    const float s8 = shv[8];
    pin(wa); pin(wb); pin(c0); pin(c1); pin(c2); pin(c3);
    const int sb = pix_silh(__float_as_int(front0 ? wa.w : wb.w));
    if (!sb) return;
    const I3 rt = front0 ? ra : rb;
    if ((unsigned)rt.x >= (unsigned)V || (unsigned)rt.y >= (unsigned)V || (unsigned)rt.z >= (unsigned)V) return;
    // ---- (2) the three clip positions exactly as ClipView::get() forms them
    const F3U x0 = *reinterpret_cast<const F3U*>(verts + (size_t)rt.x * 3), x1 = *reinterpret_cast<const F3U*>(verts + (size_t)rt.y * 3),
              x2 = *reinterpret_cast<const F3U*>(verts + (size_t)rt.z * 3);
    auto clip = [&](const F3U& v) {
        return make_float4(clip_coord(v.x, v.y, v.z, c0.x, c1.x, c2.x, c3.x), clip_coord(v.x, v.y, v.z, c0.y, c1.y, c2.y, c3.y),
                           clip_coord(v.x, v.y, v.z, c0.z, c1.z, c2.z, c3.z), clip_coord(v.x, v.y, v.z, c0.w, c1.w, c2.w, c3.w));
    };
    float alpha = 0.f;
    int va = 0, vb = 0;
    const int code = aa_analyse_core(clip(x0), clip(x1), clip(x2), sb, front0, rt.x, rt.y, rt.z, qx, qy, d, H, W, alpha, &va, &vb);
    if (code == 0) return;
    // ---- (3) both colours (shade_word with the vertex ids and the SH row already here; an uncovered pixel is the all-zero colour)
    const int4 item = make_int4(px | (py << 16), b | (code << 16), va, vb);
    const long long tgt = (alpha > 0.f) ? gid : gid1;
    float4 qa[6], qb[6];
    {
        const int ia[3] = {ra.x, ra.y, ra.z}, ib[3] = {rb.x, rb.y, rb.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            qa[2 * k] = sd.attr8[(size_t)ia[k] * 2]; qa[2 * k + 1] = sd.attr8[(size_t)ia[k] * 2 + 1];
            qb[2 * k] = sd.attr8[(size_t)ib[k] * 2]; qb[2 * k + 1] = sd.attr8[(size_t)ib[k] * 2 + 1];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) { pin(qa[k]); pin(qb[k]); }
    }
    const float shc[9] = {sA.x, sA.y, sA.z, sA.w, sB.x, sB.y, sB.z, sB.w, s8};
    auto colour = [&](const float4 w, bool covered, const float4 (&q)[6]) {
        float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!covered) return col;
        const int pk = __float_as_int(w.w);
        const float w2 = 1.0f - w.x - w.y;
        col.w = w.x + w.y + w2;
        if (pix_mask(pk)) {
            PixShade ps;
            shade_eval_attr(w.x, w.y, q[0], q[1], q[2], q[3], q[4], q[5], shc, ps);
            col.x = ps.rad * ps.ar; col.y = ps.rad * ps.ag; col.z = ps.rad * ps.ab;
        }
        return col;
    };
    const float4 ca = colour(wa, id0 != 0.f, qa), cb = colour(wb, id1 != 0.f, qb);
    const float dc[4] = {cb.x - ca.x, cb.y - ca.y, cb.z - ca.z, cb.w - ca.w};
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (dc[c] != 0.f) atomicAdd(out + tgt * 4 + c, alpha * dc[c]);
    const int slot = atomicAdd(inum, 1);                             // LDS
    if (slot < region) sd.dcols[rbase + slot] = make_float4(dc[0], dc[1], dc[2], dc[3]);   // straight to the item's final place
    if (slot < kItemBuf) {
        ibuf[slot] = item;
        abuf[slot] = alpha;
    } else if (slot < region) {                                      // buffer full (a silhouette-dense workgroup): straight to the region
        items[rbase + slot] = item;
        alphas[rbase + slot] = alpha;
    } else {
        counter[1] = 1; counter[3] = 1;                              // region overflow: the pair gets no gradient; [1] is rearmed every
                                                                     // step, [3] stays set until fmhr_ham_workspace_init (fused step: 4 words)
    }
}

// the same, out of line: called from inside the pipelined segment loop only when a workgroup's candidate buffer is full (rare);
// inlined there it would set the register budget of the whole loop
__device__ __attribute__((noinline)) void aa_fwd_pair_overflow(int px, int py, int b, int d, float z0, float id0, float z1, float id1,
                                            const AaShade& sd, const float* __restrict__ pos,
                                            const float* __restrict__ verts, const float4* __restrict__ cmat,
                                            const int32_t* __restrict__ tri, const int32_t* __restrict__ opp,
                                            const unsigned char* __restrict__ silh, int V, int F, int H, int W, int Wp,
                                            float* __restrict__ out, int4* __restrict__ items, float* __restrict__ alphas,
                                            int32_t* __restrict__ counter, int4* ibuf, float* abuf, int* inum, int region, size_t rbase)
{
    aa_fwd_pair(px, py, b, d, z0, id0, z1, id1, sd, pos, verts, cmat, tri, opp, silh, V, F, H, W, Wp, out, items, alphas, counter, ibuf, abuf,
                inum, region, rbase);
}

// ROW: W % 64 == 0 — a 64-pixel segment lies in ONE row of ONE view, so (view, row, first column) are wave-uniform and come from
// two scalar magic divisions (common.h: UDiv) instead of two ~25-instruction integer divisions in every lane of every segment
struct SegRows { UDiv per_view, per_row; unsigned segs_per_view, segs_per_row; };
FMHR_WTS_DEFINE(aaf)
template <bool ROW>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ROW ? 3 : 5, 8)))
k_aa_fwd_list(const AaShade sd, const float4* __restrict__ rast, const float* __restrict__ pos,
              const float* __restrict__ verts, const float4* __restrict__ cmat, const int32_t* __restrict__ tri, const int32_t* __restrict__ opp, const unsigned char* __restrict__ silh,
              const int* __restrict__ act_list, const int* __restrict__ act_count, const unsigned long long* __restrict__ covw,
              const unsigned long long* __restrict__ silw, int n, int V, int F, int H, int W, int Wp, float* __restrict__ out, int4* __restrict__ items,
              float* __restrict__ alphas, int32_t* __restrict__ counter, int32_t* __restrict__ region_cnt, int cap,
              const float* __restrict__ cnt_part, int nb_cnt, int* __restrict__ n_valid, float* __restrict__ n_valid_f, const SegRows sr)
{
    // work items (pairs that blend; a few % of the pixels) for the backward: every workgroup owns a fixed region of the list
    // (cap / gridDim items) and a count slot — no global atomics at all (a returning same-address atomic per wave with an item
    // serialised to ~0.25 ms per step, one per workgroup at the kernel's end still to ~13 us); the backward runs one workgroup
    // per region over dense items instead of scanning every active pixel
    FMHR_WTS_BEGIN(aaf, (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    __shared__ int4 ibuf[kItemBuf];
    __shared__ float abuf[kItemBuf];
    __shared__ int2 cpos[kCandBuf];                                // candidate pairs: px | py << 13 | d << 26, view
    __shared__ float4 cfrag[kCandBuf];                             // (z, id) of the two pixels
    __shared__ int inum, cnum;
    const int region = cap / (int)gridDim.x;
    const size_t rbase = (size_t)blockIdx.x * region;
    if (threadIdx.x == 0) { inum = 0; cnum = 0; }
    if (blockIdx.x == 0 && cnt_part != nullptr) {                  // the step's local valid-pixel count (last forward kernel:
        __shared__ float sm4[4];                                   // saves a launch): sum of k_active_segments' partials
        float acc = 0.f;
        for (int i = threadIdx.x; i < nb_cnt; i += kBlock) acc += cnt_part[i];
        float r = block_sum(acc, sm4);
        if (threadIdx.x == 0) { n_valid[0] = (int)(r + 0.5f); n_valid_f[0] = (float)(int)(r + 0.5f); }
    }
    __syncthreads();
    const int count = *act_count, lane = threadIdx.x & 63;
    // W: image width (pair exists <=> px + 1 < W; the analysis' pixel grid); Wp: row pitch of every per-pixel buffer (addressing)
    const long long P = (long long)H * Wp, N = (long long)n * P, nSeg = (N + 63) >> 6;
    // software-pipelined over the wave's segments, three stages: A(i) requests the segment id and its four coverage words, B(i)
    // turns them into coverage flags and requests the lane's per-pixel words, C(i) decodes and tests the two pairs.  While
    // C(i) runs, the loads of B(i+1) and A(i+2) are in flight: no iteration waits for a dependent chain of three loads.
    const int stride = gridDim.x * 4;
    struct Words { long long seg; unsigned long long w0, wR, wD0, wD1; };
    struct Pix { long long seg; int b, px, py; bool live, c0, cD, cR63, inR, inD; float4 r0, rD, rR; };
    auto stageA = [&](int ii, Words& wd) {
        wd.seg = -1; wd.w0 = wd.wR = wd.wD0 = wd.wD1 = 0ull;
        if (ii < count) {
            const int ent = act_list[ii];
            if (ent >= 0) return;                                    // no silhouette-face pixel in reach of this segment: nothing to find
            wd.seg = ent & 0x7fffffff;
            const long long sDn = ((wd.seg << 6) + Wp) >> 6;
            wd.w0 = covw[wd.seg]; wd.wR = covw_or0(covw, wd.seg + 1, nSeg); wd.wD0 = covw_or0(covw, sDn, nSeg);
            wd.wD1 = covw_or0(covw, sDn + 1, nSeg);
        }
    };
    auto stageB = [&](const Words& wd, Pix& q) {
        q.seg = wd.seg;
        q.r0 = make_float4(0.f, 0.f, 0.f, 0.f); q.rD = q.r0; q.rR = q.r0;
        q.b = 0; q.px = 0; q.py = 0;
        q.live = false; q.c0 = q.cD = q.cR63 = q.inR = q.inD = false;
        if (wd.seg < 0) return;
        const long long gid = (wd.seg << 6) + lane, gidD = gid + Wp, sD = ((wd.seg << 6) + Wp) >> 6;
        q.live = gid < N;
        if (ROW) {                                                   // every lane is live (N is a multiple of 64)
            const unsigned sg = (unsigned)__builtin_amdgcn_readfirstlane((int)wd.seg);
            const unsigned vb = udiv(sg, sr.per_view), r = sg - vb * sr.segs_per_view;
            const unsigned row = udiv(r, sr.per_row);
            q.b = (int)vb; q.py = (int)row; q.px = (int)((r - row * sr.segs_per_row) << 6) + lane;
        } else if (q.live) {
            int rem;
            q.b = fast_divmod(gid, (int)P, rem);
            q.py = fast_divmod(rem, Wp, q.px);
        }
        q.c0 = q.live && ((wd.w0 >> lane) & 1ull);
        q.inR = q.live && q.px + 1 < W; q.inD = q.live && q.py + 1 < H;
        q.cD = q.inD && ((((gidD >> 6) == sD ? wd.wD0 : wd.wD1) >> (gidD & 63)) & 1ull);
        q.cR63 = q.inR && (wd.wR & 1ull);                            // only lane 63 looks into the next segment
        // the per-pixel word (u, v, z/w, packed id | flags) is only defined on covered pixels: an uncovered pixel is the all-zero
        // fragment
        if (q.c0) q.r0 = rast[gid];
        if (q.cD) q.rD = rast[gidD];
        if (lane == 63 && q.cR63) q.rR = rast[gid + 1];
    };
    // the pre-test of one pixel's two pairs (right, below): which triangle is in front, does it have a silhouette edge in this view?
    // Ends almost every pair of a micro-triangle mesh; the survivors (a few per wave) are parked and analysed DENSELY by the whole
    // workgroup afterwards instead of by one or two lanes of every wave.  r0 / rR / rD: decoded words (.x silhouette bits, .z depth,
    // .w dr.rasterize id) of the pixel, its right and its lower neighbour
    auto test_pairs = [&](int b, int px, int py, bool inR, bool inD, const float4& r0, const float4& rR, const float4& rD) {
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const float4 r1 = d ? rD : rR;
            if (!((d ? inD : inR) && r1.w != r0.w)) continue;
            const int tri0 = (int)r0.w - 1, tri1 = (int)r1.w - 1;
            int t = (tri0 >= 0) ? tri0 : tri1;
            if (tri0 >= 0 && tri1 >= 0) t = (r0.z < r1.z) ? tri0 : tri1;
            // the silhouette bits of the nearer face came with its pixel's word: no gather
            if (t < 0 || t >= F || ((t == tri0) ? r0.x : r1.x) == 0.f) continue;
            const int slot = atomicAdd(&cnum, 1);                        // LDS
            if (slot < kCandBuf) {
                cpos[slot] = make_int2(px | (py << 13) | (d << 26), b);
                cfrag[slot] = make_float4(r0.z, r0.w, r1.z, r1.w);
            } else {
                aa_fwd_pair_overflow(px, py, b, d, r0.z, r0.w, r1.z, r1.w, sd, pos, verts, cmat, tri, opp, silh, V, F, H, W, Wp, out, items, alphas, counter,
                            ibuf, abuf, &inum, region, rbase);
            }
        }
    };
    // after the decode .w holds the dr.rasterize id (face + 1 as a float) and .x the face's silhouette bits
    auto decode = [](float4& r, bool cov) {
        const int p = __float_as_int(r.w);
        r.x = cov ? (float)pix_silh(p) : 0.f;
        r.w = cov ? (float)(pix_face(p) + 1) : 0.f;
    };
    if (ROW) {
        // ---- scan by MASKS.  A pair can only blend if the nearer of its two pixels belongs to a face with a silhouette edge, and the
        // resolve pass left, per segment, the word of such pixels next to the coverage word.  So the scan does not look at pixels at
        // all: per segment six 64-bit words (coverage + silhouette of the segment, of the next one in its row, of the one below) give
        // the mask M of pixels whose right or lower pair CAN blend (the pixel itself, its right or its lower neighbour is such a
        // pixel) — 4.8 of 64 on the metric mesh (tools/analysis/aa_scan_share.py).  Lane = segment for the masks, then the
        // candidate pixels are numbered over the wave, expanded into a per-wave list and tested lane = candidate: the per-pixel
        // words are loaded for them only, and the pre-test runs on dense waves.  As a per-segment pass over all 64 pixels (three
        // 16-B loads, decode, shuffle, two tests per lane) the scan was 28 of this kernel's 35 us.
        constexpr int kPixBuf = 256;                                 // candidate pixels a wave tests per round
        __shared__ int2 pixl[kBlock / 64][kPixBuf];                  // (segment << 6 | lane, flags: c0, cR, cD, inR, inD)
        const int wv = threadIdx.x >> 6;
        int2* const pl = pixl[wv];
        const unsigned spr = sr.segs_per_row;
        // lane = one of the wave's segments (a wave has ~10 of them on the metric step: usually ONE pass)
#ifdef FMHR_X_TS
        unsigned long long tsx[4] = {0ull, 0ull, 0ull, 0ull};
        const unsigned long long ts_a = wall_clock64();
#endif
        for (int base = blockIdx.x * 4 + wv; base < count; base += stride * 64) {       // wave-uniform trip count
            const int ii = base + lane * stride;
            const int ent = (ii < count) ? act_list[ii] : 0;
            unsigned long long M = 0ull, c0w = 0ull, cRw = 0ull, cDw = 0ull;
            unsigned sg = 0u;
            bool lastcol = false, lastrow = false;
            int right_end = 64;                                      // pixels of this segment that have a right neighbour INSIDE the image
            if (ent < 0) {                                           // (>= 0: no silhouette-face pixel in reach, k_active_segments)
                sg = (unsigned)ent & 0x7fffffffu;
                const unsigned vb = udiv(sg, sr.per_view), r = sg - vb * sr.segs_per_view;
                const unsigned row = udiv(r, sr.per_row);
                lastcol = r - row * spr == spr - 1; lastrow = (int)row == H - 1;
                right_end = W - 1 - (int)((r - row * spr) << 6);     // bit < right_end <=> px + 1 < W (the pitch may exceed W: pad columns)
                c0w = covw[sg];
                unsigned long long s0w = silw[sg], sRw = 0ull, sDw = 0ull;
                if (!lastcol) { cRw = covw[sg + 1]; sRw = silw[sg + 1]; }
                if (!lastrow) { cDw = covw[sg + spr]; sDw = silw[sg + spr]; }
                // the resolve pass writes a segment's silhouette word only when it resolves it: next to an empty coverage word it is stale
                s0w = c0w ? s0w : 0ull; sRw = cRw ? sRw : 0ull; sDw = cDw ? sDw : 0ull;
                M = s0w | (s0w >> 1) | (sRw << 63) | sDw;
            }
#ifdef FMHR_X_TS
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (tsx[0] == 0ull) tsx[0] = wall_clock64();
#endif
            const int cnt = __popcll(M);
            const int incl = wave_incl_scan(cnt);                    // DPP (six ds_bpermute round trips before)
            const int excl = incl - cnt, T = __builtin_amdgcn_readlane(incl, 63);
            for (int R = 0; R < T; R += kPixBuf) {                   // wave-uniform
                if (cnt != 0 && excl < R + kPixBuf && excl + cnt > R) {
                    unsigned long long m = M;
                    int idx = excl - R;
                    while (m != 0ull) {
                        const int bit = __builtin_ctzll(m);
                        m &= m - 1ull;
                        if (idx >= 0 && idx < kPixBuf) {
                            const bool c0 = (c0w >> bit) & 1ull, cR = (bit < 63) ? ((c0w >> (bit + 1)) & 1ull) : (cRw & 1ull), cD = (cDw >> bit) & 1ull;
                            const bool inR = bit < right_end, inD = !lastrow;
                            pl[idx] = make_int2((int)((sg << 6) | (unsigned)bit),
                                                (int)c0 | ((int)cR << 1) | ((int)cD << 2) | ((int)inR << 3) | ((int)inD << 4));
                        }
                        idx++;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef FMHR_X_TS
                if (tsx[1] == 0ull) tsx[1] = wall_clock64();
#endif
                const int nr = min(kPixBuf, T - R);
                for (int k0 = 0; k0 < nr; k0 += 64) {                // lane = candidate pixel
                    const int k = k0 + lane;
                    if (k < nr) {
                        const int2 e = pl[k];
                        const long long gid = (long long)(unsigned)e.x;
                        const bool c0 = e.y & 1, cR = e.y & 2, cD = e.y & 4, inR = e.y & 8, inD = e.y & 16;
                        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), rR = r0, rD = r0;
                        if (c0) r0 = rast[gid];
                        if (cR) rR = rast[gid + 1];
                        if (cD) rD = rast[gid + Wp];
                        pin(r0); pin(rR); pin(rD);
#ifdef FMHR_X_TS
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        if (tsx[2] == 0ull) tsx[2] = wall_clock64();
#endif
                        decode(r0, c0); decode(rR, cR); decode(rD, cD);
                        const unsigned sg2 = (unsigned)e.x >> 6;
                        const unsigned vb = udiv(sg2, sr.per_view), r = sg2 - vb * sr.segs_per_view;
                        const unsigned row = udiv(r, sr.per_row);
                        test_pairs((int)vb, (int)((r - row * spr) << 6) + (e.x & 63), (int)row, inR, inD, r0, rR, rD);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the list is refilled by the next round
                __builtin_amdgcn_wave_barrier();
            }
        }
#ifdef FMHR_X_TS
        if (lane == 0 && wts_w_ < 81920) {
            const unsigned long long t0_ = g_wts_aaf[wts_w_ * 4];
            auto d16 = [&](unsigned long long t) -> unsigned long long { return t == 0ull ? 0ull : ((t - t0_) & 0xffffull); };
            g_wts_aaf[wts_w_ * 4 + 2] = d16(ts_a) | (d16(tsx[0]) << 16) | (d16(tsx[1]) << 32) | (d16(tsx[2]) << 48);
        }
#endif
    } else {
    int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    Words wd;
    Pix cur, nxt;
    stageA(i, wd);
    stageB(wd, cur);
    stageA(i + stride, wd);
    for (; i < count; i += stride) {
        stageB(wd, nxt);                                             // pixel words of iteration i + 1
        stageA(i + 2 * stride, wd);                                  // segment + coverage words of iteration i + 2
        if (cur.seg < 0) { cur = nxt; continue; }                    // skipped segment (wave-uniform)
        float4 r0 = cur.r0, rD = cur.rD, rR = cur.rR;
        pin(r0); pin(rD); pin(rR);
        const bool c0 = cur.c0, cD = cur.cD, cR63 = cur.cR63;
        decode(r0, c0); decode(rD, cD); decode(rR, lane == 63 && cR63);
        {
            const float z = __shfl_down(r0.z, 1), w = __shfl_down(r0.w, 1), sb = __shfl_down(r0.x, 1);
            if (lane < 63) { rR.z = z; rR.w = w; rR.x = sb; }
        }
        test_pairs(cur.b, cur.px, cur.py, cur.inR, cur.inD, r0, rR, rD);
        cur = nxt;
    }
    }
#ifdef FMHR_X_TS
    const unsigned long long ts_scan = wall_clock64();
#endif
    __syncthreads();
#ifdef FMHR_X_TS
    const unsigned long long ts_bar = wall_clock64();
#endif
    {
        const int nc = min(cnum, kCandBuf);
        for (int c = threadIdx.x; c < nc; c += kBlock) {
            const int2 cp = cpos[c];
            const float4 cf = cfrag[c];
            if (ROW && pos == nullptr)                               // the fused step, no stored clip positions (the per-pixel-scan instantiation
                                                                     // runs at 96 registers: there the wider stages only add spills, 47.6 -> 51.4 us)
                aa_fwd_pair_staged(cp.x & 8191, (cp.x >> 13) & 8191, cp.y, (cp.x >> 26) & 1, cf.x, cf.y, cf.z, cf.w, sd, verts, cmat, tri, V, F, H, W, Wp,
                                   out, items, alphas, counter, ibuf, abuf, &inum, region, rbase);
            else
                aa_fwd_pair(cp.x & 8191, (cp.x >> 13) & 8191, cp.y, (cp.x >> 26) & 1, cf.x, cf.y, cf.z, cf.w, sd, pos, verts, cmat, tri, opp, silh, V, F, H,
                            W, Wp, out, items, alphas, counter, ibuf, abuf, &inum, region, rbase);
        }
    }
#ifdef FMHR_X_TS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long ts_dense = wall_clock64();
    if ((threadIdx.x & 63) == 0 && wts_w_ < 81920) {
        const unsigned long long t0_ = g_wts_aaf[wts_w_ * 4];
        g_wts_aaf[wts_w_ * 4 + 3] = ((ts_scan - t0_) & 0xfffff) | (((ts_bar - t0_) & 0xfffff) << 20) | (((ts_dense - t0_) & 0xfffff) << 40) | ((unsigned long long)min(cnum, 4095) << 52);
    }
#endif
    __syncthreads();
    const int nn = min(inum, min(kItemBuf, region));
    for (int i = threadIdx.x; i < nn; i += kBlock) { items[rbase + i] = ibuf[i]; alphas[rbase + i] = abuf[i]; }
    if (threadIdx.x == 0) {
        region_cnt[blockIdx.x] = min(inum, region);
        if (inum > region) { counter[1] = 1; counter[3] = 1; }
    }
    FMHR_WTS_END(aaf);
}

// backward of one work item: colour gradients (+-alpha * dy[target]) and the position gradient of the crossing point
template <int CA, int CB>
__device__ __forceinline__ void aa_bwd_item(int px, int py, int b, int code, float alpha,
         const float* __restrict__ colorA, const float* __restrict__ colorB, const float4* __restrict__ rast,
         const float* __restrict__ pos, const int32_t* __restrict__ tri, const float* __restrict__ dyA,
         const float* __restrict__ dyB, int V, int H, int W, int Cdyn,
         float* __restrict__ gcolorA, float* __restrict__ gcolorB, float* __restrict__ gpos,
         const float4* __restrict__ cmat = nullptr, float* __restrict__ gvert = nullptr, bool use_gl = false,
         float4 gl = make_float4(0.f, 0.f, 0.f, 0.f), bool packed = false, float4 dcol = make_float4(0.f, 0.f, 0.f, 0.f),
         const float* __restrict__ verts = nullptr, bool keep0 = true, bool keep1 = true, float* __restrict__ gvert2 = nullptr,
         int known_va = -1, int known_vb = -1, const float4* pre = nullptr, unsigned* st_off = nullptr, float* st_val = nullptr,
         float* split2 = nullptr, int Wp = 0)
{
    // Wp (fused step): row pitch of the per-pixel buffers the offsets below address; 0 = W (the generic op's dense images)
    // pre (fused step): the edge's two clip positions and the view's three matrix rows, loaded by the caller TOGETHER with everything
    // else the item needs: memory operations of a wave complete in order, so a load issued behind the colour atomics below would wait
    // for them (a memory-side float atomic takes microseconds) before the position atomics could even start
    const float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
    int d = (code & AA_DOWN) ? 1 : 0, di = code & 3;
    const int Wq = (Wp > 0) ? Wp : W;
    long long pix0 = ((long long)b * H + py) * Wq + px;
    long long pix1 = pix0 + (d ? Wq : 1);
    long long tgt = (alpha > 0.f) ? pix0 : pix1;
    float galpha = 0.f, galpha_a = 0.f;       // galpha_a: the alpha channel's share (fused step: the silhouette-MSE term)
    const int C = (CA > 0) ? CA : Cdyn;
#pragma unroll
    for (int c = 0; c < C; c++) {
        float g;                                                          // fused step: the target's gradient is handed in
        if (use_gl) g = (c == 0) ? gl.x : ((c == 1) ? gl.y : ((c == 2) ? gl.z : gl.w));
        else g = dyA[tgt * C + c];
        // fused step (use_gl): the colour difference of the pair is handed in as well (two 16-B loads by the caller)
        const float dcl = use_gl ? ((c == 0) ? dcol.x : ((c == 1) ? dcol.y : ((c == 2) ? dcol.z : dcol.w))) : colorA[pix1 * C + c] - colorA[pix0 * C + c];
        galpha += g * dcl;
        if (packed && c == 3) galpha_a = g * dcl;
        // fused step: an uncovered pixel's gradient is read by nobody, and neither is the alpha channel's (the interpolated
        // "ones" attribute is constant 1: no gradient flows to the barycentrics; it only moves the crossing point, via galpha)
        if (gcolorA != nullptr && g != 0.f && !(packed && c == 3)) {
            if (st_val != nullptr) {       // staged (k_aa_bwd_items): entries 0-2 = pixel 1's channels, 3-5 = pixel 0's
                if (c < 3) {
                    if (keep1) { st_off[c] = (unsigned)(pix1 * C + c); st_val[c] = alpha * g; }
                    if (keep0) { st_off[3 + c] = (unsigned)(pix0 * C + c); st_val[3 + c] = -alpha * g; }
                }
            } else {
                if (keep1) atomicAdd(gcolorA + pix1 * C + c, alpha * g);
                if (keep0) atomicAdd(gcolorA + pix0 * C + c, -alpha * g);
            }
        }
    }
    if (CB > 0) {
#pragma unroll
        for (int c = 0; c < CB; c++) {
            float g = dyB[tgt * CB + c];
            galpha += g * (colorB[pix1 * CB + c] - colorB[pix0 * CB + c]);
            if (gcolorB != nullptr && g != 0.f) {
                atomicAdd(gcolorB + pix1 * CB + c, alpha * g);
                atomicAdd(gcolorB + pix0 * CB + c, -alpha * g);
            }
        }
    }
    if ((code & AA_CLAMPED) || galpha == 0.f) return;
    int qx = px, qy = py;
    if (code & AA_TRI1) { qx += 1 - d; qy += d; }
    int va = known_va, vb = known_vb;
    if (va < 0) {                                 // generic op: pixel -> face -> the edge's vertices
        const float idw = rast[(code & AA_TRI1) ? pix1 : pix0].w;
        int t = packed ? pix_face(__float_as_int(idw)) : (int)idw - 1;
        int e1 = di + 1; if (e1 >= 3) e1 -= 3;
        int e2 = di + 2; if (e2 >= 3) e2 -= 3;
        va = tri[t * 3 + e1]; vb = tri[t * 3 + e2];
    }
    // fused step (pos == null): clip positions recomputed from the vertices and the view's combined matrix (ClipView)
    const ClipView cv = clip_view(pos, verts, cmat, b, V);
    const float4 pa = (pre != nullptr) ? pre[0] : cv.get(va), pb = (pre != nullptr) ? pre[1] : cv.get(vb);
    float fx = (float)qx + 0.5f - xh, fy = (float)qy + 0.5f - yh;
    float wa = 1.0f / pa.w, wb = 1.0f / pb.w;
    float xa = pa.x * wa * xh - fx, ya = pa.y * wa * yh - fy;
    float xb = pb.x * wb * xh - fx, yb = pb.y * wb * yh - fy;
    if (d) { float tt; tt = xa; xa = ya; ya = tt; tt = xb; xb = yb; yb = tt; }
    // the position gradient is linear in galpha: a view-sharded caller (gvert2 != null) keeps the photometric share (rgb channels,
    // normalised by the batch-global valid-pixel count AFTER the gradient all-reduce) apart from the silhouette share (alpha)
    float gxc = (gvert2 != nullptr) ? -1.0f : -galpha;
    float iden = 1.0f / (ya - yb);
    float tpar = ya * iden;
    float g_xa = gxc * (1.0f - tpar), g_xb = gxc * tpar;
    float g_t = gxc * (xb - xa);
    float g_ya = g_t * (-yb * iden * iden), g_yb = g_t * (ya * iden * iden);
    if (d) { float tt; tt = g_xa; g_xa = g_ya; g_ya = tt; tt = g_xb; g_xb = g_yb; g_yb = tt; }
    const float ax = g_xa * xh * wa, ay = g_ya * yh * wa, aw = -(g_xa * pa.x * xh + g_ya * pa.y * yh) * wa * wa;
    const float bx = g_xb * xh * wb, by = g_yb * yh * wb, bw = -(g_xb * pb.x * xh + g_yb * pb.y * yh) * wb * wb;
    if (cmat != nullptr) {
        // fused step: apply the clip-transform backward here and accumulate straight into the view-independent [V,3] buffer
        const float4 c0 = (pre != nullptr) ? pre[2] : cmat[(size_t)b * 4], c1 = (pre != nullptr) ? pre[3] : cmat[(size_t)b * 4 + 1],
                     c2 = (pre != nullptr) ? pre[4] : cmat[(size_t)b * 4 + 2];
        const float ga[3] = {ax * c0.x + ay * c0.y + aw * c0.w, ax * c1.x + ay * c1.y + aw * c1.w, ax * c2.x + ay * c2.y + aw * c2.w};
        const float gb[3] = {bx * c0.x + by * c0.y + bw * c0.w, bx * c1.x + by * c1.y + bw * c1.w, bx * c2.x + by * c2.y + bw * c2.w};
        if (st_val != nullptr) {           // staged: entries 6-8 = vertex a's components, 9-11 = vertex b's (both into gvert); the
            const float s1 = (gvert2 == nullptr) ? 1.0f : galpha - galpha_a, s2 = (gvert2 == nullptr) ? 0.f : galpha_a;     // silhouette
#pragma unroll
            for (int k = 0; k < 3; k++) {                                                                                       // share
                st_off[6 + k] = (unsigned)(va * 3 + k); st_val[6 + k] = s1 * ga[k];                                             // (gvert2)
                st_off[9 + k] = (unsigned)(vb * 3 + k); st_val[9 + k] = s1 * gb[k];                                             // is handed
                if (split2 != nullptr) { split2[k] = s2 * ga[k]; split2[3 + k] = s2 * gb[k]; }                                  // back
            }
            return;
        }
        if (gvert2 == nullptr) {
#pragma unroll
            for (int k = 0; k < 3; k++) { atomicAdd(gvert + (size_t)va * 3 + k, ga[k]); atomicAdd(gvert + (size_t)vb * 3 + k, gb[k]); }
        } else {
            const float s1 = galpha - galpha_a, s2 = galpha_a;         // (ga, gb) are per unit of galpha here
            if (s1 != 0.f)
#pragma unroll
                for (int k = 0; k < 3; k++) { atomicAdd(gvert + (size_t)va * 3 + k, s1 * ga[k]); atomicAdd(gvert + (size_t)vb * 3 + k, s1 * gb[k]); }
            if (s2 != 0.f)
#pragma unroll
                for (int k = 0; k < 3; k++) { atomicAdd(gvert2 + (size_t)va * 3 + k, s2 * ga[k]); atomicAdd(gvert2 + (size_t)vb * 3 + k, s2 * gb[k]); }
        }
        return;
    }
    float* qa = gpos + ((size_t)b * V + va) * 4;
    float* qb = gpos + ((size_t)b * V + vb) * 4;
    atomicAdd(qa + 0, ax); atomicAdd(qa + 1, ay); atomicAdd(qa + 3, aw);
    atomicAdd(qb + 0, bx); atomicAdd(qb + 1, by); atomicAdd(qb + 3, bw);
}

// backward over the compact queue (generic dr.antialias op)
template <int CA, int CB>
__global__ void __launch_bounds__(kBlock)
k_aa_bwd(const float* __restrict__ colorA, const float* __restrict__ colorB, const float4* __restrict__ rast,
         const float* __restrict__ pos, const int32_t* __restrict__ tri, const float* __restrict__ dyA,
         const float* __restrict__ dyB, const int4* __restrict__ items, const float* __restrict__ alphas,
         const int32_t* __restrict__ counter, int cap, int n, int V, int H, int W, int Cdyn,
         float* __restrict__ gcolorA, float* __restrict__ gcolorB, float* __restrict__ gpos)
{
    int cnt = min(counter[0], cap);
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < cnt; k += gridDim.x * kBlock) {
        int4 it = items[k];
        aa_bwd_item<CA, CB>(it.x, it.y, it.z, it.w, alphas[k], colorA, colorB, rast, pos, tri, dyA, dyB, V, H, W, Cdyn, gcolorA,
                            gcolorB, gpos);
    }
}

// backward over the work-item list of the fused step: one workgroup per region of the list, dense items.  d loss / d aa of an
// item's target pixel is RECOMPUTED here from the loss definition (k_loss_pix: L1 over valid pixels + silhouette MSE) — a few
// hundred thousand items — instead of being stored by k_loss_pix for all 4.4 M active pixels in a second gradient image
FMHR_WTS_DEFINE(aab)
struct LossGrad {
    const float4* aa; const float* img; const float* mask; const float* valid_mask; const unsigned long long* covw;
    const float* n_valid; float w_sfs, w_mask, mse_den;
    const int32_t* view_idx;                                       // ABI 6: img / mask / valid_mask hold the capture's views, view b of the step = view_idx[b] of them (null: the same)
};

__global__ void __launch_bounds__(kBlock)
k_aa_bwd_items(const float4* __restrict__ dcols, const float4* __restrict__ rast, const float* __restrict__ pos,
               const float* __restrict__ verts, const int32_t* __restrict__ tri, const LossGrad L, const int4* __restrict__ items,
               const float* __restrict__ alphas, const int32_t* __restrict__ region_cnt, int cap, int V, int H, int W, int Wp,
               float* __restrict__ gcolor, const float4* __restrict__ cmat, float* __restrict__ gvert, float* __restrict__ gvert2)
{
    FMHR_WTS_BEGIN(aab, (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    const int region = cap / (int)gridDim.x, count = region_cnt[blockIdx.x];      // same grid as k_aa_fwd_list
    const size_t rbase = (size_t)blockIdx.x * region;
    // A silhouette vertex collects the items of several adjacent pixel pairs (up to ~30 in the metric step) and float atomics on ONE
    // address are served one after the other at the memory side (~0.4 us each): with 2.4 k items in the whole 48-view step this
    // kernel spent half its time waiting for such chains.  The [V,3] accumulators are therefore kept in kAaReplicas copies, a
    // workgroup adds to copy (blockIdx & 7) — a vertex's items come from unrelated workgroups — and the mesh tail sums the copies.
    gvert += (size_t)(blockIdx.x & (kAaReplicas - 1)) * V * 3;
    if (gvert2 != nullptr) gvert2 += (size_t)(blockIdx.x & (kAaReplicas - 1)) * V * 3;
    const float den = 3.0f * L.n_valid[0];
    const float k1 = (den > 0.f) ? L.w_sfs / den : 0.f, k2 = L.w_mask / L.mse_den;
    // The twelve float atomics of an item go to FOUR cache lines (a pixel's three colour channels are one 16-B word, a vertex's three
    // components 12 B) and a posted atomic costs ~23 cycles of the CU's memory pipeline per distinct line it touches, whatever the
    // number of lanes on that line (tools/micro/atomic_lines.hip): issued item by item — lane = item, one instruction per component —
    // the 55 k items of a steady-state step are 660 k line requests = this kernel's whole 24 us (5 workgroups x 43 items x 12 per
    // CU).  So the contributions are staged in LDS, twelve (offset, value) entries per item, and issued TRANSPOSED: lane = entry, so
    // that the lanes of one instruction that belong to one item hit that item's four lines together.
    constexpr int kE = 12;
    __shared__ unsigned st_off[kBlock * kE];
    __shared__ float st_val[kBlock * kE];
    for (int base = 0; base < count; base += kBlock) {               // workgroup-uniform trip count (one round unless a region is dense)
        const int i = base + (int)threadIdx.x;
        const int nround = min(kBlock, count - base);
        float split2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        unsigned* const my_off = st_off + threadIdx.x * kE;
        float* const my_val = st_val + threadIdx.x * kE;
        if (i < count) {
#pragma unroll
        for (int e = 0; e < kE; e++) my_val[e] = 0.f;
        const int4 raw = items[rbase + i];                          // (px | py << 16, view | code << 16, edge vertex a, edge vertex b)
        const int4 it = make_int4(raw.x & 0xffff, (int)((unsigned)raw.x >> 16), raw.y & 0xffff, (int)((unsigned)raw.y >> 16));
        const float alpha = alphas[rbase + i];
        // the item's pixels in the step's own buffers (row pitch Wp) and the blend target in the caller's [n,H,W] tensors
        const int dn = (it.w & AA_DOWN) ? 1 : 0;
        const long long pix0 = ((long long)it.z * H + it.y) * Wp + it.x;
        const long long tgt = (alpha > 0.f) ? pix0 : pix0 + (dn ? Wp : 1);
        const int cview = (L.view_idx != nullptr) ? L.view_idx[it.z] : it.z;       // the view in the caller's arrays (a 4-B gather from a table of n entries)
        const long long ctgt = ((long long)cview * H + it.y + ((alpha > 0.f) ? 0 : dn)) * W + it.x + ((alpha > 0.f) ? 0 : 1 - dn);
        const float4 a = L.aa[tgt];
        const F3U im = *reinterpret_cast<const F3U*>(L.img + ctgt * 3);
        const float mk = L.mask[ctgt], vm = L.valid_mask[ctgt];
        const bool cov = (L.covw[tgt >> 6] >> (tgt & 63)) & 1ull;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mk > 0.f && cov) {
            const float d0 = a.x - im.x, d1 = a.y - im.y, d2 = a.z - im.z;
            g.x = (d0 > 0.f) ? k1 : ((d0 < 0.f) ? -k1 : 0.f);
            g.y = (d1 > 0.f) ? k1 : ((d1 < 0.f) ? -k1 : 0.f);
            g.z = (d2 > 0.f) ? k1 : ((d2 < 0.f) ? -k1 : 0.f);
        }
        g.w = 2.0f * (a.w - vm) * k2;
        const float4 dcol = dcols[rbase + i];                      // colour difference of the pair, left by the forward
        float4 pre[5];                                             // edge vertices' clip positions + the view's matrix rows (see aa_bwd_item)
        {
            const ClipView cv = clip_view(pos, verts, cmat, it.z, V);
            pre[0] = cv.get(raw.z); pre[1] = cv.get(raw.w);
            pre[2] = cmat[(size_t)it.z * 4]; pre[3] = cmat[(size_t)it.z * 4 + 1]; pre[4] = cmat[(size_t)it.z * 4 + 2];
            pin(pre[0]); pin(pre[1]); pin(pre[2]); pin(pre[3]); pin(pre[4]);
        }
        const long long pix1 = pix0 + (dn ? Wp : 1);
        // d loss / d shaded colour exists for covered pixels only (k_loss_pix stores it for them, the owner face reads it)
        const bool cov0 = (L.covw[pix0 >> 6] >> (pix0 & 63)) & 1ull, cov1 = (L.covw[pix1 >> 6] >> (pix1 & 63)) & 1ull;
        aa_bwd_item<4, 0>(it.x, it.y, it.z, it.w, alpha, nullptr, nullptr, rast, pos, tri, nullptr, nullptr, V, H, W, 4, gcolor, nullptr,
                          nullptr, cmat, gvert, true, g, true, dcol, verts, cov0, cov1, gvert2, raw.z, raw.w, pre, my_off, my_val,
                          (gvert2 != nullptr) ? split2 : nullptr, Wp);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nround * kE; e += kBlock) {
            const float v = st_val[e];
            if (v != 0.f) {
                const int k = e - (e / kE) * kE;
                atomicAdd(((k < 6) ? gcolor : gvert) + st_off[e], v);
            }
        }
        if (gvert2 != nullptr) {                                     // view-sharded step: the silhouette share, same vertices (entries 6-11)
            __syncthreads();
            if (i < count)
#pragma unroll
                for (int k = 0; k < 6; k++) my_val[6 + k] = split2[k];
            __syncthreads();
            for (int e = threadIdx.x; e < nround * 6; e += kBlock) {
                const int item = e / 6, k = e - item * 6;
                const float v = st_val[item * kE + 6 + k];
                if (v != 0.f) atomicAdd(gvert2 + st_off[item * kE + 6 + k], v);
            }
        }
        __syncthreads();                                             // the stage is free for the next round
    }
    FMHR_WTS_END(aab);
}

static unsigned table_size(int F)
{
    unsigned T = 64;
    while ((long long)T < 6ll * F) T <<= 1;
    return T;
}

static int build_hash(const int32_t* tri, int F, EdgeSlot* tab, unsigned T, hipStream_t st)
{
    k_hash_clear<<<cdiv(T, kBlock), kBlock, 0, st>>>(tab, T);
    FMHR_CHECK_LAUNCH("k_hash_clear");
    if (F > 0) {
        k_hash_insert<<<cdiv(3ll * F, kBlock), kBlock, 0, st>>>(tri, F, tab, T - 1);
        FMHR_CHECK_LAUNCH("k_hash_insert");
        k_hash_second<<<cdiv(3ll * F, kBlock), kBlock, 0, st>>>(tri, F, tab, T - 1);
        FMHR_CHECK_LAUNCH("k_hash_second");
    }
    return FMHR_OK;
}

// fused-step entry points (ham_step.hip): ONE 4-channel colour (rgb = shaded image, a = coverage colour), `rast` = the step's
// per-pixel word (shade_common.h), work items in per-workgroup regions of one list, persistent grids
int aa_fwd_rgba(const float* attr8, const float* sh, float* dcols, const float* rast, const float* pos, const float* verts,
                const float* cmat, const int32_t* tri, const int32_t* opp,
                const unsigned char* silh, const int* act_list, const int* act_count, const unsigned long long* covw,
                const unsigned long long* silw, int n, int V, int F, int H, int W, int Wp, float* out, int32_t* items, float* alphas, int32_t* counter, int32_t* region_cnt, int cap,
                const float* cnt_part, int nb_cnt, int* n_valid, float* n_valid_f, hipStream_t st)
{
    AaShade sd;
    sd.words = (const float4*)rast; sd.attr8 = (const float4*)attr8; sd.sh = sh; sd.dcols = (float4*)dcols;
    SegRows sr;
    memset(&sr, 0, sizeof(sr));
    const bool row = aa_row_aligned(n, H, Wp);
    if (row) {
        sr.segs_per_view = (unsigned)((long long)H * Wp >> 6); sr.segs_per_row = (unsigned)(Wp >> 6);
        sr.per_view = make_udiv(sr.segs_per_view); sr.per_row = make_udiv(sr.segs_per_row);
        k_aa_fwd_list<true><<<aa_grid(true), kBlock, 0, st>>>(sd, (const float4*)rast, pos, verts, (const float4*)cmat, tri, opp, silh, act_list, act_count, covw, silw, n, V, F, H, W, Wp,
                                           out, (int4*)items, alphas, counter, region_cnt, cap, cnt_part, nb_cnt, n_valid, n_valid_f, sr);
    } else {
        k_aa_fwd_list<false><<<aa_grid(false), kBlock, 0, st>>>(sd, (const float4*)rast, pos, verts, (const float4*)cmat, tri, opp, silh, act_list, act_count, covw, silw, n, V, F, H, W, Wp,
                                           out, (int4*)items, alphas, counter, region_cnt, cap, cnt_part, nb_cnt, n_valid, n_valid_f, sr);
    }
    FMHR_CHECK_LAUNCH("k_aa_fwd_list");
    return FMHR_OK;
}

int aa_bwd_rgba(const float* dcols, const float* rast, const float* pos, const float* verts, const int32_t* tri, const float* aa, const float* img,
                const float* mask, const float* valid_mask, const unsigned long long* covw, const float* n_valid, float w_sfs,
                float w_mask, float mse_den, const int32_t* items, const float* alphas, const int32_t* region_cnt, int cap, int n, int V,
                int H, int W, int Wp, float* gcolor, const float* cmat, float* gvert, float* gvert2, const int32_t* view_idx, hipStream_t st)
{
    LossGrad L;
    L.view_idx = view_idx;
    L.aa = (const float4*)aa; L.img = img; L.mask = mask; L.valid_mask = valid_mask; L.covw = covw; L.n_valid = n_valid;
    L.w_sfs = w_sfs; L.w_mask = w_mask; L.mse_den = mse_den;
    k_aa_bwd_items<<<aa_grid(aa_row_aligned(n, H, Wp)), kBlock, 0, st>>>(      // the forward's grid: one workgroup per region
        (const float4*)dcols, (const float4*)rast, pos, verts, tri, L, (const int4*)items, alphas, region_cnt, cap, V, H, W, Wp,
                                           gcolor, (const float4*)cmat, gvert, gvert2);
    FMHR_CHECK_LAUNCH("k_aa_bwd_items");
    return FMHR_OK;
}

}  // namespace fmhr

using namespace fmhr;

extern "C" size_t fmhr_topology_workspace_bytes(int F) { return (size_t)table_size(F) * sizeof(EdgeSlot); }

extern "C" int fmhr_topology_build(const int32_t* tri, int F, int V, int32_t* opp, void* ws, size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(F >= 0 && V >= 0, "topology_build: bad dims");
    if (F == 0) return FMHR_OK;
    FMHR_CHECK_ARG(tri && opp && ws, "topology_build: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_topology_workspace_bytes(F), "topology_build: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned T = table_size(F);
    EdgeSlot* tab = (EdgeSlot*)ws;
    int rc = build_hash(tri, F, tab, T, st);
    if (rc) return rc;
    k_hash_opp<<<cdiv(3ll * F, kBlock), kBlock, 0, st>>>(tri, F, tab, T - 1, opp);
    FMHR_CHECK_LAUNCH("k_hash_opp");
    return FMHR_OK;
}

extern "C" size_t fmhr_adjacency_workspace_bytes(int V, int F)
{
    return (size_t)table_size(F) * sizeof(EdgeSlot) + (size_t)(V + 1) * 4 * 2;
}

extern "C" int fmhr_adjacency_build(const int32_t* tri, int F, int V, int32_t* row_ptr, int32_t* col_idx, void* ws,
                                    size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(F >= 0 && V >= 0, "adjacency_build: bad dims");
    FMHR_CHECK_ARG(row_ptr && ws && (tri || F == 0) && (col_idx || F == 0), "adjacency_build: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_adjacency_workspace_bytes(V, F), "adjacency_build: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned T = table_size(F);
    EdgeSlot* tab = (EdgeSlot*)ws;
    int* deg = (int*)((char*)ws + (size_t)T * sizeof(EdgeSlot));
    int* cursor = deg + (V + 1);
    hipError_t e = hipMemsetAsync(deg, 0, (size_t)(V + 1) * 8, st);
    if (e != hipSuccess) return hip_fail(e, "memset deg");
    int rc = build_hash(tri, F, tab, T, st);
    if (rc) return rc;
    k_adj_degree<<<cdiv(T, kBlock), kBlock, 0, st>>>(tab, T, V, deg);
    FMHR_CHECK_LAUNCH("k_adj_degree");
    k_scan_exclusive<<<1, 1024, 0, st>>>(deg, V, row_ptr);
    FMHR_CHECK_LAUNCH("k_scan_exclusive");
    k_adj_fill<<<cdiv(T, kBlock), kBlock, 0, st>>>(tab, T, V, row_ptr, cursor, col_idx);
    FMHR_CHECK_LAUNCH("k_adj_fill");
    if (V > 0) {
        k_adj_sort_rows<<<cdiv(V, kBlock), kBlock, 0, st>>>(V, row_ptr, col_idx);
        FMHR_CHECK_LAUNCH("k_adj_sort_rows");
    }
    return FMHR_OK;
}

extern "C" size_t fmhr_vertex_faces_workspace_bytes(int V) { return (size_t)(V + 1) * 4 * 2; }

extern "C" int fmhr_vertex_faces_build(const int32_t* tri, int F, int V, int32_t* vf_ptr, int32_t* vf_idx, void* ws,
                                       size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(F >= 0 && V >= 0, "vertex_faces_build: bad dims");
    FMHR_CHECK_ARG(vf_ptr && ws && (tri || F == 0) && (vf_idx || F == 0), "vertex_faces_build: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_vertex_faces_workspace_bytes(V), "vertex_faces_build: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int* deg = (int*)ws;
    int* cursor = deg + (V + 1);
    hipError_t e = hipMemsetAsync(deg, 0, (size_t)(V + 1) * 8, st);
    if (e != hipSuccess) return hip_fail(e, "memset deg");
    if (F > 0) {
        k_vf_count<<<cdiv(3ll * F, kBlock), kBlock, 0, st>>>(tri, F, V, deg);
        FMHR_CHECK_LAUNCH("k_vf_count");
    }
    k_scan_exclusive<<<1, 1024, 0, st>>>(deg, V, vf_ptr);
    FMHR_CHECK_LAUNCH("k_scan_exclusive");
    if (F > 0) {
        k_vf_fill<<<cdiv(3ll * F, kBlock), kBlock, 0, st>>>(tri, F, V, vf_ptr, cursor, vf_idx);
        FMHR_CHECK_LAUNCH("k_vf_fill");
    }
    if (V > 0) {
        k_adj_sort_rows<<<cdiv(V, kBlock), kBlock, 0, st>>>(V, vf_ptr, vf_idx);
        FMHR_CHECK_LAUNCH("k_adj_sort_rows");
    }
    return FMHR_OK;
}

extern "C" int fmhr_corner_slots_build(const int32_t* vf_idx, int F, int32_t* corner_slot, void* stream)
{
    FMHR_CHECK_ARG(F >= 0, "corner_slots_build: bad F=%d", F);
    if (F == 0) return FMHR_OK;
    FMHR_CHECK_ARG(vf_idx && corner_slot, "corner_slots_build: null pointer");
    k_corner_slots<<<cdiv(3LL * F, kBlock), kBlock, 0, (hipStream_t)stream>>>(vf_idx, 3 * F, corner_slot);
    FMHR_CHECK_LAUNCH("k_corner_slots");
    return FMHR_OK;
}

extern "C" int fmhr_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                  const int32_t* opp, int n, int V, int F, int H, int W, int C, float* out, int32_t* items,
                                  float* alphas, int32_t* counter, int cap, void* stream)
{
    FMHR_CHECK_ARG(n >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0 && C > 0 && cap >= 0, "antialias_fwd: bad dims");
    long long nP = (long long)n * H * W;
    if (nP == 0) return FMHR_OK;
    FMHR_CHECK_ARG(color && rast && out && items && alphas && counter && (pos || V == 0) && ((tri && opp) || F == 0),
                   "antialias_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(out, color, (size_t)nP * C * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return hip_fail(e, "antialias_fwd copy");
    if (F == 0) return FMHR_OK;
    int grid = cdiv(nP, kBlock);
    int4* it = (int4*)items;
    const float4* r4 = (const float4*)rast;
    if (C == 1) k_aa_fwd<1, 0><<<grid, kBlock, 0, st>>>(color, nullptr, r4, pos, tri, opp, n, V, F, H, W, C, out, nullptr, it, alphas, counter, cap);
    else if (C == 3) k_aa_fwd<3, 0><<<grid, kBlock, 0, st>>>(color, nullptr, r4, pos, tri, opp, n, V, F, H, W, C, out, nullptr, it, alphas, counter, cap);
    else k_aa_fwd<0, 0><<<grid, kBlock, 0, st>>>(color, nullptr, r4, pos, tri, opp, n, V, F, H, W, C, out, nullptr, it, alphas, counter, cap);
    FMHR_CHECK_LAUNCH("k_aa_fwd");
    return FMHR_OK;
}

extern "C" int fmhr_antialias_bwd(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                  const float* dy, const int32_t* items, const float* alphas, const int32_t* counter,
                                  int cap, int n, int V, int F, int H, int W, int C, float* gcolor, float* gpos, void* stream)
{
    FMHR_CHECK_ARG(n >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0 && C > 0 && cap >= 0, "antialias_bwd: bad dims");
    long long nP = (long long)n * H * W;
    if (nP == 0) return FMHR_OK;
    FMHR_CHECK_ARG(color && rast && dy && items && alphas && counter && gcolor && (gpos || V == 0), "antialias_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(gcolor, dy, (size_t)nP * C * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return hip_fail(e, "antialias_bwd copy");
    if (F == 0 || cap == 0) return FMHR_OK;
    int grid = min(max(cdiv(cap, kBlock), 1), 2048);
    const int4* it = (const int4*)items;
    const float4* r4 = (const float4*)rast;
    if (C == 1) k_aa_bwd<1, 0><<<grid, kBlock, 0, st>>>(color, nullptr, r4, pos, tri, dy, nullptr, it, alphas, counter, cap, n, V, H, W, C, gcolor, nullptr, gpos);
    else if (C == 3) k_aa_bwd<3, 0><<<grid, kBlock, 0, st>>>(color, nullptr, r4, pos, tri, dy, nullptr, it, alphas, counter, cap, n, V, H, W, C, gcolor, nullptr, gpos);
    else k_aa_bwd<0, 0><<<grid, kBlock, 0, st>>>(color, nullptr, r4, pos, tri, dy, nullptr, it, alphas, counter, cap, n, V, H, W, C, gcolor, nullptr, gpos);
    FMHR_CHECK_LAUNCH("k_aa_bwd");
    return FMHR_OK;
}

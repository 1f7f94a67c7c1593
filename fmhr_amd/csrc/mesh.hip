// fmhr_amd/csrc/mesh.hip — view-independent mesh kernels for gfx950: vertex normals, SH radiance, uniform Laplacian,
// clip transform, edge-length hinge and delta term (forward + backward).
// Replaces get_normals / get_radiance / laplacian_smoothing("uniform") of /root/reference/models/utils.py:508-548,
// 208-226, 696-722 and the inline code of /root/reference/mesh_sfs_optim.py:262-264, 296-304.
#include "common.h"
#include "adam.h"
#include <math.h>
#include <string.h>

namespace fmhr {

__device__ __forceinline__ float3 ld3(const float* p) { return make_float3(p[0], p[1], p[2]); }
__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 cross3(float3 a, float3 b)
{
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ void atomic_add3(float* p, float3 v) { atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); }

// ------------------------------------------------------------------------------------------------ vertex normals
// one lane per (batch, face): corner k adds cross(v[k+1]-v[k], v[k+2]-v[k]) to vertex k (models/utils.py:517-543)
__global__ void __launch_bounds__(kBlock)
k_normals_faces(const float* __restrict__ verts, const int32_t* __restrict__ tri, int nb, int V, int F, float* __restrict__ nsum)
{
    __shared__ float st_val[kBlock * 9];
    __shared__ unsigned st_row[kBlock * 3];
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    bool on = gid < (long long)nb * F;
    float val[9];
    unsigned row[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 9; k++) val[k] = 0.f;
    if (on) {
        int f;
        const int b = fast_divmod(gid, F, f);
        int i[3] = {tri[f * 3], tri[f * 3 + 1], tri[f * 3 + 2]};
        on = !((unsigned)i[0] >= (unsigned)V || (unsigned)i[1] >= (unsigned)V || (unsigned)i[2] >= (unsigned)V);
        if (on) {
            const float* vb = verts + (size_t)b * V * 3;
            float3 v[3] = {ld3(vb + (size_t)i[0] * 3), ld3(vb + (size_t)i[1] * 3), ld3(vb + (size_t)i[2] * 3)};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float3 c = cross3(sub3(v[(k + 1) % 3], v[k]), sub3(v[(k + 2) % 3], v[k]));
                val[k * 3] = c.x; val[k * 3 + 1] = c.y; val[k * 3 + 2] = c.z;
                row[k] = (unsigned)((long long)b * V + i[k]);
            }
        }
    }
    // nine float atomics per face, to three 12-B vertex rows: issued transposed (common.h: wave_add_3x3)
    wave_add_3x3<3, 2>(nsum, on, val, row, st_val + (threadIdx.x & ~63) * 9, st_row + (threadIdx.x & ~63) * 3);
}

__global__ void __launch_bounds__(kBlock)
k_normals_normalize(const float* __restrict__ nsum, long long N, float* __restrict__ normals)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= N) return;
    float3 s = ld3(nsum + gid * 3);
    float len = sqrtf(s.x * s.x + s.y * s.y + s.z * s.z);
    float inv = 1.0f / fmaxf(len, 1e-6f);
    normals[gid * 3] = s.x * inv; normals[gid * 3 + 1] = s.y * inv; normals[gid * 3 + 2] = s.z * inv;
}

// g_s = d L / d nsum from d L / d normals (normalise backward with the eps clamp); gstride lets the caller pass [V,8] rows
__global__ void __launch_bounds__(kBlock)
k_normals_bwd_vertex(const float* __restrict__ nsum, const float* __restrict__ gn, int gstride, long long N, float* __restrict__ gs)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= N) return;
    float3 s = ld3(nsum + gid * 3);
    float3 g = ld3(gn + gid * gstride);
    float len = sqrtf(s.x * s.x + s.y * s.y + s.z * s.z);
    float3 o;
    if (len > 1e-6f) {
        float inv = 1.0f / len;
        float3 nh = make_float3(s.x * inv, s.y * inv, s.z * inv);
        float dt = nh.x * g.x + nh.y * g.y + nh.z * g.z;
        o = make_float3((g.x - nh.x * dt) * inv, (g.y - nh.y * dt) * inv, (g.z - nh.z * dt) * inv);
    } else {
        o = make_float3(g.x * 1e6f, g.y * 1e6f, g.z * 1e6f);
    }
    gs[gid * 3] = o.x; gs[gid * 3 + 1] = o.y; gs[gid * 3 + 2] = o.z;
}

__global__ void __launch_bounds__(kBlock)
k_normals_bwd_faces(const float* __restrict__ verts, const int32_t* __restrict__ tri, const float* __restrict__ gs,
                    int nb, int V, int F, float* __restrict__ gverts)
{
    __shared__ float st_val[kBlock * 9];
    __shared__ unsigned st_row[kBlock * 3];
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    bool on = gid < (long long)nb * F;
    float val[9];
    unsigned row[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 9; k++) val[k] = 0.f;
    if (on) {
        int f;
        const int b = fast_divmod(gid, F, f);
        int i[3] = {tri[f * 3], tri[f * 3 + 1], tri[f * 3 + 2]};
        on = !((unsigned)i[0] >= (unsigned)V || (unsigned)i[1] >= (unsigned)V || (unsigned)i[2] >= (unsigned)V);
        if (on) {
            const float* vb = verts + (size_t)b * V * 3;
            const float* gb = gs + (size_t)b * V * 3;
            float3 v[3] = {ld3(vb + (size_t)i[0] * 3), ld3(vb + (size_t)i[1] * 3), ld3(vb + (size_t)i[2] * 3)};
            float3 acc[3] = {make_float3(0, 0, 0), make_float3(0, 0, 0), make_float3(0, 0, 0)};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                float3 e1 = sub3(v[k1], v[k]), e2 = sub3(v[k2], v[k]);
                float3 gc = ld3(gb + (size_t)i[k] * 3);
                float3 ge1 = cross3(e2, gc), ge2 = cross3(gc, e1);
                acc[k1].x += ge1.x; acc[k1].y += ge1.y; acc[k1].z += ge1.z;
                acc[k2].x += ge2.x; acc[k2].y += ge2.y; acc[k2].z += ge2.z;
                acc[k].x -= ge1.x + ge2.x; acc[k].y -= ge1.y + ge2.y; acc[k].z -= ge1.z + ge2.z;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                val[k * 3] = acc[k].x; val[k * 3 + 1] = acc[k].y; val[k * 3 + 2] = acc[k].z;
                row[k] = (unsigned)((long long)b * V + i[k]);
            }
        }
    }
    wave_add_3x3<3, 2>(gverts, on, val, row, st_val + (threadIdx.x & ~63) * 9, st_row + (threadIdx.x & ~63) * 3);
}

// ------------------------------------------------------------------------------------------------ SH radiance (order 3, 9 terms)
__device__ __forceinline__ float sh9(const float* c, float x, float y, float z)
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
__device__ __forceinline__ float3 sh9_dn(const float* c, float x, float y, float z)
{
    return make_float3(c[3] + c[4] * y - 2.0f * c[6] * x + c[7] * z + 2.0f * c[8] * x,
                       c[1] + c[4] * x + c[5] * z - 2.0f * c[6] * y - 2.0f * c[8] * y,
                       c[2] + c[5] * y + 4.0f * c[6] * z + c[7] * x);
}

__global__ void __launch_bounds__(kBlock)
k_sh_fwd(const float* __restrict__ coeff, const float* __restrict__ normal, int P, int per_pixel, float* __restrict__ out)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const float* c = coeff + (per_pixel ? (size_t)i * 9 : 0);
    float cc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) cc[k] = c[k];
    out[i] = sh9(cc, normal[i * 3], normal[i * 3 + 1], normal[i * 3 + 2]);
}

__global__ void __launch_bounds__(kBlock)
k_sh_bwd(const float* __restrict__ coeff, const float* __restrict__ normal, const float* __restrict__ dy, int P, int per_pixel,
         float* __restrict__ gcoeff, float* __restrict__ gnormal)
{
    __shared__ float sm[4];
    int i = blockIdx.x * kBlock + threadIdx.x;
    bool live = i < P;
    float x = 0, y = 0, z = 0, g = 0;
    float cc[9];
    const float* c = coeff + ((per_pixel && live) ? (size_t)i * 9 : 0);
#pragma unroll
    for (int k = 0; k < 9; k++) cc[k] = c[k];
    if (live) { x = normal[i * 3]; y = normal[i * 3 + 1]; z = normal[i * 3 + 2]; g = dy[i]; }
    float basis[9] = {1.0f, y, z, x, x * y, y * z, 2.0f * z * z - x * x - y * y, z * x, x * x - y * y};
    if (live) {
        float3 dn = sh9_dn(cc, x, y, z);
        gnormal[i * 3] = g * dn.x; gnormal[i * 3 + 1] = g * dn.y; gnormal[i * 3 + 2] = g * dn.z;
    }
    if (per_pixel) {
        if (live)
#pragma unroll
            for (int k = 0; k < 9; k++) gcoeff[(size_t)i * 9 + k] = g * basis[k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            float s = block_sum(g * basis[k], sm);
            if (threadIdx.x == 0) atomicAdd(gcoeff + k, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------ uniform Laplacian on a CSR
template <int C>
__global__ void __launch_bounds__(kBlock)
k_lap_fwd(const float* __restrict__ x, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col, int V,
          float scale, float* __restrict__ lx, float* __restrict__ loss)
{
    __shared__ float sm[4];
    int i = blockIdx.x * kBlock + threadIdx.x;
    float nrm = 0.f;
    if (i < V) {
        int a = row_ptr[i], b = row_ptr[i + 1];
        float xi[C], acc[C];
#pragma unroll
        for (int c = 0; c < C; c++) { xi[c] = x[(size_t)i * C + c]; acc[c] = 0.f; }
        // (L x)_i = mean_j x_j - x_i, accumulated as sum_j (x_j - x_i): neighbours are close, so the differences are
        // (nearly) exact and the cancellation of the reference's L.mm(x) formulation (~1e-4 relative in fp32) is avoided
        for (int j = a; j < b; j++) {
            const float* xj = x + (size_t)col[j] * C;
#pragma unroll
            for (int c = 0; c < C; c++) acc[c] += xj[c] - xi[c];
        }
        float inv = (b > a) ? 1.0f / (float)(b - a) : 0.0f;
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            float v = (b > a) ? acc[c] * inv : -xi[c];
            lx[(size_t)i * C + c] = v;
            s2 += v * v;
        }
        nrm = sqrtf(s2);
    }
    float s = block_sum(nrm, sm);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(loss, s * scale);
}

// rowg_i = gscale * lx_i / |lx_i|  (0 where the row norm is 0, as torch's norm backward)
template <int C>
__global__ void __launch_bounds__(kBlock)
k_lap_bwd_rows(const float* __restrict__ lx, int V, float gscale, float* __restrict__ rowg)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= V) return;
    float v[C], s2 = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) { v[c] = lx[(size_t)i * C + c]; s2 += v[c] * v[c]; }
    float nrm = sqrtf(s2);
    float k = (nrm > 0.f) ? gscale / nrm : 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) rowg[(size_t)i * C + c] = v[c] * k;
}

// gx_j += sum_{i in N(j)} rowg_i / deg_i - rowg_j   (L^T r by gather: the adjacency is symmetric)
template <int C>
__global__ void __launch_bounds__(kBlock)
k_lap_bwd_gather(const float* __restrict__ rowg, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col, int V,
                 float* __restrict__ gx)
{
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= V) return;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) acc[c] = -rowg[(size_t)j * C + c];
    for (int e = row_ptr[j]; e < row_ptr[j + 1]; e++) {
        int i = col[e];
        float inv = 1.0f / (float)(row_ptr[i + 1] - row_ptr[i]);
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] += rowg[(size_t)i * C + c] * inv;
    }
#pragma unroll
    for (int c = 0; c < C; c++) gx[(size_t)j * C + c] += acc[c];
}

// pos[b,v,:] = [v,1] @ (w2c[b] @ proj[b])  — row-vector convention; the reference's two einsums (mesh_sfs_optim.py:258-259)
// associated the other way: equal to the last bit or two, and the oracle is evaluated at THESE positions
template <bool HAS_DELTA>
__device__ __forceinline__ void transform_body(int v, int b, const float* __restrict__ verts, const float* __restrict__ delta,
                                               const float* __restrict__ w2c, const float* __restrict__ proj, int V,
                                               float4* __restrict__ pos, int H, int W, int2* __restrict__ snap,
                                               float4* __restrict__ cmat, const int32_t* __restrict__ view_idx,
                                               const float* __restrict__ sh, float* __restrict__ sh_local, int n_src, int32_t* __restrict__ view_idx_ws,
                                               int* __restrict__ stamp)
{
    // ABI 6: view b of the step is view view_idx[b] of the caller's per-view arrays (the loop's minibatch, mesh_sfs_optim.py:255-260);
    // this kernel is the only reader of the small ones: the combined matrix lands in cmat[b], the SH row in sh_local[b]
    // The index is the caller's data: it is CLAMPED to the capture's views here, the clamped value is what every later kernel of the step
    // reads (view_idx_ws: the four pixel kernels index img / mask / valid_mask through it), and an index that needed clamping sets the
    // sticky word stamp[1] the host reads with the overflow flags — a bad index is a reported error, never an out-of-bounds read
    int sb = b;
    if (view_idx != nullptr) {
        const int raw = view_idx[b];
        sb = min(max(raw, 0), n_src - 1);
        if (blockIdx.x == 0 && threadIdx.x == 41) {
            view_idx_ws[b] = sb;
            if (raw != sb) stamp[1] = 1;
        }
    }
    const float* A = w2c + (size_t)sb * 16;
    const float* B = proj + (size_t)sb * 16;
    if (sh_local != nullptr && blockIdx.x == 0 && threadIdx.x >= 32 && threadIdx.x < 41) sh_local[(size_t)b * 9 + (threadIdx.x - 32)] = sh[(size_t)sb * 9 + (threadIdx.x - 32)];
    // the vertex (and its delta) is requested FIRST, from a clamped index, so that it travels with the matrix entries: behind the
    // barrier and a `delta != nullptr` test it was two more dependent round trips of every thread (HAS_DELTA is a template
    // parameter for that reason: a value merged under a run-time test gets its wait right behind the load)
    const int vc = min(v, V - 1);
    float x = verts[(size_t)vc * 3], y = verts[(size_t)vc * 3 + 1], z = verts[(size_t)vc * 3 + 2];
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (HAS_DELTA) { dx = delta[(size_t)vc * 3]; dy = delta[(size_t)vc * 3 + 1]; dz = delta[(size_t)vc * 3 + 2]; }
    // the combined matrix w2c @ proj, one entry per thread through LDS.  The backward recomputes a face's clip positions from
    // the published (x, y, w) columns with clip_coord() as well: the positions it differentiates at are then bit-identical to
    // the ones rasterized (an edge-on triangle amplifies a last-bit difference between the two by 1 / area)
    __shared__ float c[4][4];
    if (threadIdx.x < 16) {
        const int k = threadIdx.x >> 2, l = threadIdx.x & 3;
        c[k][l] = __fmaf_rn(A[k * 4], B[l], __fmaf_rn(A[k * 4 + 1], B[4 + l], __fmaf_rn(A[k * 4 + 2], B[8 + l], A[k * 4 + 3] * B[12 + l])));
    }
    __syncthreads();                 // block-uniform branch of k_geometry: every thread of a transform block gets here
    if (cmat != nullptr && v < 4)    // row v of w2c @ proj: rows 0..2 = d pos / d vertex_v (the clip-transform backward), row 3 the translation
        cmat[(size_t)b * 4 + v] = make_float4(c[v][0], c[v][1], c[v][2], c[v][3]);
    if (v >= V) return;
    if (HAS_DELTA) {             // fused step: vertex = vertices_tmp + delta (mesh_sfs_optim.py:255), the same single add as the normals part
        x = x + dx; y = y + dy; z = z + dz;
    }
    float o[4];
#pragma unroll
    for (int l = 0; l < 4; l++) o[l] = clip_coord(x, y, z, c[0][l], c[1][l], c[2][l], c[3][l]);
    float4 p = make_float4(o[0], o[1], o[2], o[3]);
    if (pos != nullptr) pos[(size_t)b * V + v] = p;        // the fused step keeps only the snapped vertices
    if (snap != nullptr) {          // R1 + R2 of the rasterizer, once per (view, vertex)
        int x, y;
        bool ok = snap_vertex(p, (float)W * 128.0f, (float)H * 128.0f, x, y);
        snap[(size_t)b * V + v] = ok ? make_int2(x, y) : make_int2((int)0x80000000, 0);
    }
}

// delta term: mean_v |delta_v|^2 (mesh_sfs_optim.py:304); gdelta += gscale * 2 delta
// ------------------------------------------------------------------------------------------------ Adam (A13)
// torch.optim.Adam step (amsgrad=False, weight_decay=0, maximize=False) — mesh_sfs_optim.py:244,310.  torch's fused
// multi_tensor kernel takes ~94 us per parameter group here (147k elements each); this one is bandwidth-trivial.
__global__ void __launch_bounds__(kBlock) k_adam(const AdamTensor T)
{
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= T.n) return;
    adam_element(T, i, T.g[i]);
}

// two parameters in one launch (the HAM loops step exactly two: delta + albedo, or albedo + SH): each launch is ~4.7 us of
// fixed cost against ~0.6 MB of work
__global__ void __launch_bounds__(kBlock) k_adam2(const AdamTensor A, const AdamTensor B)
{
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const bool second = i >= A.n;
    const AdamTensor& T = second ? B : A;
    if (second) i -= A.n;
    if (i >= T.n) return;
    adam_element(T, i, T.g[i]);
}

// ------------------------------------------------------------------------------------------------ CSR-gather forms (fused step)
// kSubLanes lanes per vertex walk its incident face corners (vf CSR, entry = 3*face + corner): no float atomics, deterministic
// summation order.  The face-scatter forms above took ~40 us each on atomics into a 0.6 MB array; these take a few us.
static_assert(true, "");
constexpr int kSubLanes = 8;     // lanes per vertex in the CSR-walking kernels (49k lane-per-vertex threads cannot fill 1024 SIMDs)

// sum over the kSubLanes = 8 lanes of a vertex, every lane gets the total: the xor-1 / xor-2 butterflies as DPP quad permutes, the
// last step by row_half_mirror (lane i <-> 7 - i: after two steps every lane of a quad holds the quad's sum, so the mirror pairs the
// two quads) — the same three additions per lane as the __shfl_xor form (operands swapped: bit-identical), without its three
// dependent ds_bpermute round trips through the LDS crossbar per value (k_mesh_tail_a reduces ~30 values: 8 us of LDS time per CU)
__device__ __forceinline__ float sum_sub(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));     // row_half_mirror
    return x;
}

template <bool HAS_DELTA>
__device__ __forceinline__ void normals_gather_body(int blk, const float* __restrict__ verts, const int32_t* __restrict__ tri,
                                                    const int32_t* __restrict__ vf_ptr, const int32_t* __restrict__ vf_idx, int V,
                                                    float* __restrict__ nsum, float* __restrict__ normals,
                                                    const float* __restrict__ albedo, float4* __restrict__ attr8,
                                                    const float* __restrict__ delta, float* __restrict__ vout)
{
    // fused step: vertex = verts + delta (mesh_sfs_optim.py:255), formed on the fly and stored for the later kernels by the
    // vertex's first lane (saves the separate add launch); stand-alone op: HAS_DELTA = false.
    // Written in STAGES, every load of a stage requested (unconditionally, from clamped indices) before its first use: (0) the
    // vertex, its delta, its CSR row bounds and albedo; (1) the lane's first incident corner; (2) that face's other two vertex
    // ids; (3) their positions and deltas.  As vertex(tri[...]) calls behind a `delta != nullptr` test it was TEN dependent
    // round trips per lane (vertex -> delta -> row -> corner -> id -> position -> delta -> id -> position -> delta).
    const int v = (blk * kBlock + threadIdx.x) / kSubLanes, sub = threadIdx.x & (kSubLanes - 1);
    const bool live = v < V;
    const int vv = live ? v : 0;
    auto pos_of = [&](int i, float3& q, float3& dl) {
        q = ld3(verts + (size_t)i * 3);
        dl = make_float3(0.f, 0.f, 0.f);
        if (HAS_DELTA) dl = ld3(delta + (size_t)i * 3);
    };
    auto add3 = [&](float3 q, float3 dl) { if (HAS_DELTA) { q.x = q.x + dl.x; q.y = q.y + dl.y; q.z = q.z + dl.z; } return q; };
    // ---- stage 0
    float3 p, pd;
    pos_of(vv, p, pd);
    const int c0 = vf_ptr[vv], c1 = vf_ptr[vv + 1];
    float3 al = make_float3(0.f, 0.f, 0.f);
    if (attr8 != nullptr) al = ld3(albedo + (size_t)vv * 3);        // (uniform)
    // ---- stage 1
    const int e0 = c0 + sub;
    const bool he = live && e0 < c1;
    const int code0 = vf_idx[max(min(e0, c1 - 1), 0)];
    // ---- stage 2
    const int f0 = code0 / 3, k0 = code0 - f0 * 3;
    const int k1 = (k0 + 1 == 3) ? 0 : k0 + 1, k2 = (k1 + 1 == 3) ? 0 : k1 + 1;
    const int ia = tri[f0 * 3 + k1], ib = tri[f0 * 3 + k2];
    // ---- stage 3
    float3 a, ad, b, bd;
    pos_of(ia, a, ad);
    pos_of(ib, b, bd);
    p = add3(p, pd);
    float3 acc = make_float3(0.f, 0.f, 0.f);
    if (live) {
        if (vout != nullptr && sub < 3) vout[(size_t)vv * 3 + sub] = (sub == 0) ? p.x : ((sub == 1) ? p.y : p.z);
        if (he) {
            float3 c = cross3(sub3(add3(a, ad), p), sub3(add3(b, bd), p));      // corner k: cross(v[k+1]-v[k], v[k+2]-v[k]) (models/utils.py:517-543)
            acc.x += c.x; acc.y += c.y; acc.z += c.z;
        }
        for (int e = e0 + kSubLanes; e < c1; e += kSubLanes) {              // valence > kSubLanes
            int code = vf_idx[e];
            int f = code / 3, k = code - f * 3;
            int kk1 = (k + 1 == 3) ? 0 : k + 1, kk2 = (kk1 + 1 == 3) ? 0 : kk1 + 1;
            float3 a2, a2d, b2, b2d;
            pos_of(tri[f * 3 + kk1], a2, a2d);
            pos_of(tri[f * 3 + kk2], b2, b2d);
            float3 c = cross3(sub3(add3(a2, a2d), p), sub3(add3(b2, b2d), p));
            acc.x += c.x; acc.y += c.y; acc.z += c.z;
        }
    }
    acc.x = sum_sub(acc.x); acc.y = sum_sub(acc.y); acc.z = sum_sub(acc.z);
    // results: one component per lane (lanes 0..2: the vertex, its normal sum, its normal), the two packed attribute words from lanes 3
    // and 4 — four store instructions instead of eleven from one finishing lane in eight (every lane holds the sums)
    if (!live) return;
    const float len = sqrtf(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z);
    const float inv = 1.0f / fmaxf(len, 1e-6f);
    if (sub < 3) {
        const size_t i = (size_t)v * 3 + sub;
        const float ac = (sub == 0) ? acc.x : ((sub == 1) ? acc.y : acc.z);
        nsum[i] = ac;
        normals[i] = ac * inv;
    } else if (attr8 != nullptr && sub < 5) {     // packed per-vertex attributes for the pixel kernels: (normal.xyz, albedo.x | albedo.yz, 0, 0)
        attr8[(size_t)v * 2 + (sub - 3)] = (sub == 3) ? make_float4(acc.x * inv, acc.y * inv, acc.z * inv, al.x) : make_float4(al.y, al.z, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ fused backward tail
// The stage-2 backward ends in thirteen tiny per-vertex kernels (each >= 4.5 us of launch + dependent-load latency for 49k
// vertices); they collapse into two, split where a vertex needs its NEIGHBOURS' intermediate results.
// A (own data only): sum the face-backward corner slots -> view-dependent gradients; normal-normalise backward -> gs; both uniform
//    Laplacians forward -> loss terms and scaled row gradients; delta term; edge hinge (loss + gradient).
// B (neighbour gathers): vertex-normal backward from gs, L^T r for both Laplacians -> final gdelta / galbedo.
struct MeshTailArgs {
    const float4* part; float* gvert_aa; int G;                                       // face-backward corner sums [G, 3F] x 3 float4
    float* gvert_aa2;                                                                 // split mode: the silhouette share of the antialias position gradient
    float* vacc;                                                                      // [V] x 3 float4: the large-box faces' backward sums per vertex (shade.hip), packed like a corner slot
    const float* verts; const float* delta; const float* albedo; const float* nsum;
    const int32_t* tri; const int32_t* vf_ptr; const int32_t* vf_idx; const int32_t* row_ptr; const int32_t* col;
    const int32_t* col_mult;                                                          // faces per adjacency entry (fmhr_edge_multiplicity_build)
    int V, F;
    int v0, v1;                                                                       // vertex range of the view-independent terms
    float lap_w, alb_w, edge_w, delta_w, edge_mean, rg;                               // weights (already / V or / 3F where due)
    float* gs; float* rowg_v; float* rowg_a; float* gacc_v; float* gacc_a; float* gacc_r;
    float* loss_part;                                                                 // [4, gridDim.x] weighted per-block loss sums
    // tail B outputs.  One rank: gdelta / galbedo = the whole gradient (gd_rest == null).  View-sharded (split): gdelta / galbedo =
    // the PHOTOMETRIC share, computed with the L1 divisor left out; gd_rest / ga_rest = everything else (silhouette, regularisers)
    float* gdelta; float* galbedo; float* gd_rest; float* ga_rest;
    AdamTensor ad_d, ad_a;                                                            // fused optimiser update (p == null: off)
};

// Both kernels give a vertex kSub lanes: 49k vertices are < 1 wave per SIMD, so a lane-per-vertex launch leaves every dependent
// gather chain (CSR row -> neighbour -> value) exposed; kSub lanes stride over the vertex's rows and combine with xor-shuffles.
FMHR_WTS_DEFINE(taila)
__global__ void __launch_bounds__(kBlock) k_mesh_tail_a(const MeshTailArgs A)
{
    FMHR_WTS_BEGIN(taila, (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    const int v = (blockIdx.x * kBlock + threadIdx.x) / kSubLanes, sub = threadIdx.x & (kSubLanes - 1);
    float l_lap = 0.f, l_alb = 0.f, l_delta = 0.f;
    const bool live = v < A.V;                                     // uniform over a vertex's lanes
    const int vv = live ? v : 0;
    // (1) corner sums of the face-parallel backward: a vertex's corners are 3*deg consecutive float4 per view-group plane (vf CSR
    //     order); the lanes stride over them (neighbouring vertices continue the same run: coalesced), the G plane loads of one
    //     element are independent
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    if (live) {
        const int c0 = A.vf_ptr[vv], nq = (A.vf_ptr[vv + 1] - c0) * 3;
        const size_t plane = (size_t)A.F * 9;
        for (int j = sub; j < nq; j += kSubLanes) {
            const float4* q = A.part + (size_t)c0 * 3 + j;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g = 0; g < A.G; g++) {
                float4 t = q[(size_t)g * plane];
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
            const int slot = j % 3;
            if (slot == 0) { n0 += acc.x; n1 += acc.y; n2 += acc.z; a0 += acc.w; }
            else if (slot == 1) { a1 += acc.x; a2 += acc.y; gx += acc.z; gy += acc.w; }
            else gz += acc.x;
        }
    }
    // (1b) what the workers of the large-box queue added for this vertex (a handful of vertices per step; zero everywhere else): three
    //      float4 in the packing of a corner slot, taken by lanes 0..2 and re-zeroed for the next step
    if (live && sub < 3 && A.vacc != nullptr) {
        float4* q = reinterpret_cast<float4*>(A.vacc) + (size_t)vv * 3 + sub;
        const float4 t = *q;
        if (t.x != 0.f || t.y != 0.f || t.z != 0.f || t.w != 0.f) *q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sub == 0) { n0 += t.x; n1 += t.y; n2 += t.z; a0 += t.w; }
        else if (sub == 1) { a1 += t.x; a2 += t.y; gx += t.z; gy += t.w; }
        else gz += t.x;
    }
    // (3) uniform Laplacians of the vertices and of the albedo: differences accumulated per neighbour
    float lv[3] = {0.f, 0.f, 0.f}, la[3] = {0.f, 0.f, 0.f};
    int deg = 0;
    float3 xv = make_float3(0.f, 0.f, 0.f), xa = xv;
    const bool mine = live && v >= A.v0 && v < A.v1;                // this rank evaluates the regularisers of [v0, v1) only
    // (5) the edge hinge (mesh_sfs_optim.py:296-302): mean over the 3F directed face edges of clip(|e|^2 - mean, 0, 1).  An edge {i, j}
    //     appears once per face that holds it (col_mult, 1 on the boundary, 2 inside), so the sum over faces is a sum over the
    //     adjacency with that multiplicity: this vertex owns the half-edges that leave it (half the loss of each edge) and the whole
    //     gradient with respect to its own position, sum_j m_ij [0 < d < 1] 2 w (x_i - x_j) — evaluated in the neighbour walk the
    //     Laplacians need anyway
    float hg[3] = {0.f, 0.f, 0.f}, l_edge = 0.f;
    if (mine) {
        const int ra = A.row_ptr[vv], rb = A.row_ptr[vv + 1];
        deg = rb - ra;
        xv = ld3(A.verts + (size_t)vv * 3); xa = ld3(A.albedo + (size_t)vv * 3);
        const float hk = 2.f * A.edge_w * A.rg;
        for (int j = ra + sub; j < rb; j += kSubLanes) {
            int i = A.col[j];
            const float m = (float)A.col_mult[j];
            float3 yv = ld3(A.verts + (size_t)i * 3), ya = ld3(A.albedo + (size_t)i * 3);
            const float ex = xv.x - yv.x, ey = xv.y - yv.y, ez = xv.z - yv.z;
            lv[0] -= ex; lv[1] -= ey; lv[2] -= ez;
            la[0] += ya.x - xa.x; la[1] += ya.y - xa.y; la[2] += ya.z - xa.z;
            const float d = ex * ex + ey * ey + ez * ez - A.edge_mean;
            l_edge += 0.5f * m * fminf(fmaxf(d, 0.f), 1.f);
            if (d > 0.f && d < 1.f) { const float w = hk * m; hg[0] += w * ex; hg[1] += w * ey; hg[2] += w * ez; }
        }
    }
    n0 = sum_sub(n0); n1 = sum_sub(n1); n2 = sum_sub(n2); a0 = sum_sub(a0); a1 = sum_sub(a1); a2 = sum_sub(a2);
    gx = sum_sub(gx); gy = sum_sub(gy); gz = sum_sub(gz);
#pragma unroll
    for (int c = 0; c < 3; c++) { lv[c] = sum_sub(lv[c]); la[c] = sum_sub(la[c]); hg[c] = sum_sub(hg[c]); }
    // antialias-backward atomics: kAaReplicas copies of the [V,3] accumulators (one per lane of the vertex), consumed and re-zeroed
    // for the next step
    static_assert(kAaReplicas == kSubLanes, "one replica per lane of a vertex");
    float aa1[3] = {0.f, 0.f, 0.f}, aa2[3] = {0.f, 0.f, 0.f};
    if (live) {
        float* ga = A.gvert_aa + ((size_t)sub * A.V + v) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) { aa1[c] = ga[c]; if (aa1[c] != 0.f) ga[c] = 0.f; }
        if (A.gvert_aa2 != nullptr) {
            float* g2 = A.gvert_aa2 + ((size_t)sub * A.V + v) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) { aa2[c] = g2[c]; if (aa2[c] != 0.f) g2[c] = 0.f; }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { aa1[c] = sum_sub(aa1[c]); aa2[c] = sum_sub(aa2[c]); }
    // The vertex's results, one COMPONENT per lane (lanes 0..2; every lane of the vertex holds the 8-lane sums and forms the small
    // vector quantities itself): six 4-B stores with three lanes active, a vertex's components one 12-B run — as one finishing lane per
    // vertex they were eighteen stores with one lane in eight active.  The loss terms are still taken from lane 0 alone.
    if (live && sub < 3) {
        const int c = sub;
        float r3[3] = {hg[0] + aa2[0], hg[1] + aa2[1], hg[2] + aa2[2]};   // everything that is NOT the photometric term: hinge, delta, silhouette share
        const float g3[3] = {gx + aa1[0], gy + aa1[1], gz + aa1[2]};
        const size_t i = (size_t)v * 3 + c;
        // (2) d/d nsum of the normalised vertex normal (models/utils.py:545, eps 1e-6)
        {
            float3 s = ld3(A.nsum + (size_t)v * 3);
            float len = sqrtf(s.x * s.x + s.y * s.y + s.z * s.z);
            float3 o;
            if (len > 1e-6f) {
                float inv = 1.0f / len;
                float3 nh = make_float3(s.x * inv, s.y * inv, s.z * inv);
                float dt = nh.x * n0 + nh.y * n1 + nh.z * n2;
                o = make_float3((n0 - nh.x * dt) * inv, (n1 - nh.y * dt) * inv, (n2 - nh.z * dt) * inv);
            } else {
                o = make_float3(n0 * 1e6f, n1 * 1e6f, n2 * 1e6f);
            }
            A.gs[i] = (c == 0) ? o.x : ((c == 1) ? o.y : o.z);
        }
        // (3) row norms -> loss, scaled rows -> backward (part B); an isolated vertex has the row -x (models/utils.py:684)
        if (!mine) {
            A.rowg_v[i] = 0.f; A.rowg_a[i] = 0.f;
        } else {
            const float inv = (deg > 0) ? 1.0f / (float)deg : 0.0f;
            const float xs[3] = {xv.x, xv.y, xv.z}, as[3] = {xa.x, xa.y, xa.z};
#pragma unroll
            for (int k = 0; k < 3; k++) { lv[k] = (deg > 0) ? lv[k] * inv : -xs[k]; la[k] = (deg > 0) ? la[k] * inv : -as[k]; }
            float nv = sqrtf(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2]);
            float na = sqrtf(la[0] * la[0] + la[1] * la[1] + la[2] * la[2]);
            if (sub == 0) { l_lap = nv; l_alb = na; }
            float kv = (nv > 0.f) ? A.lap_w * A.rg / nv : 0.f, ka = (na > 0.f) ? A.alb_w * A.rg / na : 0.f;
            A.rowg_v[i] = ((c == 0) ? lv[0] : ((c == 1) ? lv[1] : lv[2])) * kv;
            A.rowg_a[i] = ((c == 0) ? la[0] : ((c == 1) ? la[1] : la[2])) * ka;
        }
        // (4) delta term (mesh_sfs_optim.py:304)
        if (mine) {
            float3 d = ld3(A.delta + (size_t)v * 3);
            if (sub == 0) l_delta = d.x * d.x + d.y * d.y + d.z * d.z;
            float k = 2.f * A.delta_w * A.rg;
            r3[0] += k * d.x; r3[1] += k * d.y; r3[2] += k * d.z;
        }
        A.gacc_v[i] = (c == 0) ? g3[0] : ((c == 1) ? g3[1] : g3[2]);
        A.gacc_r[i] = (c == 0) ? r3[0] : ((c == 1) ? r3[1] : r3[2]);
        A.gacc_a[i] = (c == 0) ? a0 : ((c == 1) ? a1 : a2);
    }
    // per-block partials (the finish block of k_mesh_tail_b adds them up: 4 x 1.5k same-address atomics would serialise for ~60 us).
    // ONE reduction for the four sums: four block_sum() calls in a row were eight workgroup barriers at the end of every workgroup
    {
        __shared__ float sm4x4[4][4];
        float r[4] = {wave_sum(l_lap), wave_sum(l_alb), wave_sum(l_edge), wave_sum(l_delta)};
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 4; k++) sm4x4[wv][k] = r[k];
        __syncthreads();
        if (threadIdx.x < 4) {
            const int k = threadIdx.x;
            const float w = (k == 0) ? A.lap_w : ((k == 1) ? A.alb_w : ((k == 2) ? A.edge_w : A.delta_w));
            A.loss_part[(size_t)k * gridDim.x + blockIdx.x] = (sm4x4[0][k] + sm4x4[1][k] + sm4x4[2][k] + sm4x4[3][k]) * w;
        }
    }
    FMHR_WTS_END(taila);
}

// the step's loss vector from the per-workgroup partials of k_loss_pix and k_mesh_tail_a (run by one extra workgroup of
// k_mesh_tail_b: saves a ~7 us launch); deterministic summation order
struct FinishArgs {
    float* losses; const float* n_valid; const float* part_l1; const float* part_l2; int nb;
    const float* tail_part; int tail_nb; float w_sfs, w_mask, mse_den;
    float reg_scale;      // 0 switches the view-independent loss slots off (reg_grad_scale = 0)
    const float* nvalid_report; float nvalid_scale;   // the count this rank reports in the n_valid slot (x scale): the loss vectors of
                                                      // all ranks must SUM to the global values
    int split;            // view-sharded: the photometric slot holds w_sfs * sum |d| with the L1 divisor left out (fmhr_ham_combine)
};

// All ~10 k partials are requested before the first is used (six arrays, up to kFinU strided loads per thread and array, then ONE
// combined reduction): as serial strided loops with a block_sum() each, this single workgroup took 14 us — it, not the per-vertex
// work, was the duration of k_mesh_tail_b.
constexpr int kFinU = 8;          // loads per thread and array in flight (covers 2048 entries with 256 threads)
__device__ __forceinline__ void finish_losses_block(const FinishArgs& A, float (*sm)[6])
{
    const float* arr[6] = {A.tail_part, A.tail_part + A.tail_nb, A.tail_part + 2 * (size_t)A.tail_nb, A.tail_part + 3 * (size_t)A.tail_nb,
                           A.part_l1, A.part_l2};
    const int len[6] = {A.tail_nb, A.tail_nb, A.tail_nb, A.tail_nb, A.nb, A.nb};
    float v[6][kFinU];
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
        for (int u = 0; u < kFinU; u++) {
            const int j = threadIdx.x + u * kBlock;
            v[k][u] = (j < len[k]) ? arr[k][j] : 0.f;
        }
    float r[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < kFinU; u++) t += v[k][u];
        for (int j = threadIdx.x + kFinU * kBlock; j < len[k]; j += kBlock) t += arr[k][j];      // longer arrays (V > 65 k): the rest
        r[k] = wave_sum(t);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 6; k++) sm[wv][k] = r[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) r[k] = sm[0][k] + sm[1][k] + sm[2][k] + sm[3][k];
        const float s1 = r[4], s2 = r[5];
        float* losses = A.losses;
        float den = A.split ? 1.0f : 3.0f * A.n_valid[0];
        losses[FMHR_LOSS_SFS] = (den > 0.f) ? s1 * (A.w_sfs / den) : 0.f;
        losses[FMHR_LOSS_MASK] = s2 * (A.w_mask / A.mse_den);
        losses[FMHR_LOSS_LAP] = r[0] * A.reg_scale; losses[FMHR_LOSS_ALBEDO] = r[1] * A.reg_scale;
        losses[FMHR_LOSS_EDGE] = r[2] * A.reg_scale; losses[FMHR_LOSS_DELTA] = r[3] * A.reg_scale;
        losses[FMHR_LOSS_TOTAL] = losses[FMHR_LOSS_SFS] + losses[FMHR_LOSS_LAP] + losses[FMHR_LOSS_ALBEDO] +
                                  losses[FMHR_LOSS_MASK] + losses[FMHR_LOSS_DELTA] + losses[FMHR_LOSS_EDGE];
        losses[FMHR_LOSS_NVALID] = A.nvalid_report[0] * A.nvalid_scale;
    }
}

FMHR_WTS_DEFINE(tailb)
__global__ void __launch_bounds__(kBlock) k_mesh_tail_b(const MeshTailArgs A, const FinishArgs fin)
{
    FMHR_WTS_BEGIN(tailb, (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    if (blockIdx.x == 0) {                       // dispatched first: its one long round trip overlaps the per-vertex workgroups
        __shared__ float sm46[4][6];
        finish_losses_block(fin, sm46);
        FMHR_WTS_END(tailb);
        return;
    }
    const float* __restrict__ verts = A.verts;
    const int32_t* __restrict__ tri = A.tri;
    const float* __restrict__ gs = A.gs;
    const int v = (((int)blockIdx.x - 1) * kBlock + threadIdx.x) / kSubLanes, sub = threadIdx.x & (kSubLanes - 1);
    const bool live = v < A.V;
    const int vv = live ? v : 0;
    float3 acc = make_float3(0.f, 0.f, 0.f), acl = acc, aca = acc;   // normal backward (photometric) | L^T r vertices | L^T r albedo
    if (live) {
        // vertex-normal backward: this vertex's share of d(sum_k gs[i_k] . cross_k)/d verts over its incident faces
        for (int e = A.vf_ptr[vv] + sub; e < A.vf_ptr[vv + 1]; e += kSubLanes) {
            int code = A.vf_idx[e];
            int f = code / 3, c = code - f * 3;
            int i[3] = {tri[f * 3], tri[f * 3 + 1], tri[f * 3 + 2]};
            float3 p[3] = {ld3(verts + (size_t)i[0] * 3), ld3(verts + (size_t)i[1] * 3), ld3(verts + (size_t)i[2] * 3)};
            float3 g[3] = {ld3(gs + (size_t)i[0] * 3), ld3(gs + (size_t)i[1] * 3), ld3(gs + (size_t)i[2] * 3)};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                float3 e1 = sub3(p[k1], p[k]), e2 = sub3(p[k2], p[k]);
                float3 ge1 = cross3(e2, g[k]), ge2 = cross3(g[k], e1);
                if (c == k1) { acc.x += ge1.x; acc.y += ge1.y; acc.z += ge1.z; }
                if (c == k2) { acc.x += ge2.x; acc.y += ge2.y; acc.z += ge2.z; }
                if (c == k) { acc.x -= ge1.x + ge2.x; acc.y -= ge1.y + ge2.y; acc.z -= ge1.z + ge2.z; }
            }
        }
        // L^T r for both Laplacians (the adjacency is symmetric): sum_{i in N(v)} r_i / deg_i - r_v
        for (int e = A.row_ptr[vv] + sub; e < A.row_ptr[vv + 1]; e += kSubLanes) {
            int i = A.col[e];
            float inv = 1.0f / (float)(A.row_ptr[i + 1] - A.row_ptr[i]);
            float3 qv = ld3(A.rowg_v + (size_t)i * 3), qa = ld3(A.rowg_a + (size_t)i * 3);
            acl.x += qv.x * inv; acl.y += qv.y * inv; acl.z += qv.z * inv;
            aca.x += qa.x * inv; aca.y += qa.y * inv; aca.z += qa.z * inv;
        }
    }
    acc.x = sum_sub(acc.x); acc.y = sum_sub(acc.y); acc.z = sum_sub(acc.z);
    acl.x = sum_sub(acl.x); acl.y = sum_sub(acl.y); acl.z = sum_sub(acl.z);
    aca.x = sum_sub(aca.x); aca.y = sum_sub(aca.y); aca.z = sum_sub(aca.z);
    // Final combination, one (parameter, component) per lane: lanes 0..2 of a vertex finish delta.x/y/z, lanes 3..5 albedo.x/y/z (every
    // lane holds the 8-lane sums).  As one finishing lane per vertex this section was ~57 memory instructions of 4-B accesses with one
    // lane in eight active (five 3-component loads, up to twelve stores, 2 x 3 Adam elements of three loads + three stores each) — the
    // Adam update alone was 4.6 of the kernel's 14 us (k_mesh_tail_b 9.4 us in the split form, which leaves it to the combine kernel);
    // now it is ten, and a vertex's three components go out as one 12-B run.  Same statements per component: bit-identical.
    if (live && sub < 6) {
        const bool is_alb = sub >= 3;
        const int c = is_alb ? sub - 3 : sub;
        const size_t i = (size_t)v * 3 + c;
        const float* const src_g = is_alb ? A.gacc_a : A.gacc_v;
        const float* const src_row = is_alb ? A.rowg_a : A.rowg_v;
        const float g0 = src_g[i], row = src_row[i];
        const float r0 = is_alb ? 0.f : A.gacc_r[i];
        const float nb = (c == 0) ? acc.x : ((c == 1) ? acc.y : acc.z);
        const float lt = is_alb ? ((c == 0) ? aca.x : ((c == 1) ? aca.y : aca.z)) : ((c == 0) ? acl.x : ((c == 1) ? acl.y : acl.z));
        // photometric share (with the batch-global L1 divisor already in it on one rank, left out when view-sharded) | the rest
        float ph, rest;
        if (is_alb) { ph = g0; rest = lt - row; }
        else { ph = g0 + nb; rest = r0 + lt - row; }
        float* const out_p = is_alb ? A.galbedo : A.gdelta;
        if (A.gd_rest != nullptr) {
            float* const out_r = is_alb ? A.ga_rest : A.gd_rest;
            out_p[i] = ph; out_r[i] = rest;
        } else {
            const float g = ph + rest;
            out_p[i] = g;
            // the parameters' final gradients are in registers: apply the optimiser step here (nothing in this kernel reads them)
            if (is_alb) { if (A.ad_a.p != nullptr) adam_element(A.ad_a, (long long)i, g); }
            else if (A.ad_d.p != nullptr) adam_element(A.ad_d, (long long)i, g);
        }
    }
    FMHR_WTS_END(tailb);
}

// After the ONE all-reduce of a view-sharded step: bucket = [P_delta | P_albedo | R_delta | R_albedo | 8 loss slots], summed over
// the ranks.  g = P / (3 * n_valid_global) + R (mesh_sfs_optim.py:289: F.l1_loss averages over the batch-global 3 * Pv), written
// over P; the photometric loss slot gets the same divisor and the total is formed.  n_valid == 0: the photometric term is 0.
__global__ void __launch_bounds__(kBlock) k_ham_combine(float* __restrict__ bucket, long long n3, const AdamTensor ad_d, const AdamTensor ad_a)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    float* losses = bucket + 4 * n3;
    const float nv = losses[FMHR_LOSS_NVALID];
    const float sc = (nv > 0.f) ? 1.0f / (3.0f * nv) : 0.f;
    if (i < 2 * n3) {
        const float g = bucket[i] * sc + bucket[2 * n3 + i];
        bucket[i] = g;
        if (i < n3) { if (ad_d.p != nullptr) adam_element(ad_d, i, g); }
        else if (ad_a.p != nullptr) adam_element(ad_a, i - n3, g);
    }
    if (i == 2 * n3) {                      // one thread beyond the gradient range finishes the loss vector
        const float sfs = losses[FMHR_LOSS_SFS] * sc;
        losses[FMHR_LOSS_SFS] = sfs;
        losses[FMHR_LOSS_TOTAL] = sfs + losses[FMHR_LOSS_LAP] + losses[FMHR_LOSS_ALBEDO] + losses[FMHR_LOSS_MASK] + losses[FMHR_LOSS_DELTA] +
                                  losses[FMHR_LOSS_EDGE];
    }
}

int adam_tensor_from(AdamTensor& T, const fmhr_adam_param* a, long long n);
int ham_combine_impl(float* bucket, int V, const fmhr_adam_param* adam_delta, const fmhr_adam_param* adam_albedo, hipStream_t st)
{
    const long long n3 = 3ll * V;
    AdamTensor ad, aa;
    int rc = adam_tensor_from(ad, adam_delta, n3);
    if (rc) return rc;
    rc = adam_tensor_from(aa, adam_albedo, n3);
    if (rc) return rc;
    k_ham_combine<<<cdiv(2 * n3 + 1, kBlock), kBlock, 0, st>>>(bucket, n3, ad, aa);
    FMHR_CHECK_LAUNCH("k_ham_combine");
    return FMHR_OK;
}

// faces per adjacency entry: one lane per face corner k adds 1 to the entries (a -> b) and (b -> a) of its edge (a, b) = (tri[k+1],
// tri[k+2]); rows are sorted and short (valence ~6): linear search
__global__ void __launch_bounds__(kBlock)
k_edge_multiplicity(const int32_t* __restrict__ tri, int F, int V, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                    int32_t* __restrict__ mult)
{
    const int code = blockIdx.x * kBlock + threadIdx.x;
    if (code >= 3 * F) return;
    const int f = code / 3, k = code - f * 3;
    const int a = tri[f * 3 + (k + 1) % 3], b = tri[f * 3 + (k + 2) % 3];
    if ((unsigned)a >= (unsigned)V || (unsigned)b >= (unsigned)V || a == b) return;
    for (int e = row_ptr[a]; e < row_ptr[a + 1]; e++)
        if (col[e] == b) { atomicAdd(mult + e, 1); break; }
    for (int e = row_ptr[b]; e < row_ptr[b + 1]; e++)
        if (col[e] == a) { atomicAdd(mult + e, 1); break; }
}

int mesh_tail_blocks(int V) { return cdiv((long long)V * kSubLanes, kBlock); }

int mesh_tail_impl(const MeshTailArgs& A, const FinishArgs& fin, hipStream_t st)
{
    if (A.V == 0) return FMHR_OK;
    k_mesh_tail_a<<<cdiv((long long)A.V * kSubLanes, kBlock), kBlock, 0, st>>>(A);
    FMHR_CHECK_LAUNCH("k_mesh_tail_a");
    k_mesh_tail_b<<<cdiv((long long)A.V * kSubLanes, kBlock) + 1, kBlock, 0, st>>>(A, fin);
    FMHR_CHECK_LAUNCH("k_mesh_tail_b");
    return FMHR_OK;
}

// ---- internal launchers used by ham_step.hip

int normals_fwd_impl(const float* verts, const int32_t* tri, int nb, int V, int F, float* nsum, float* normals, hipStream_t st)
{
    long long N = (long long)nb * V;
    if (N == 0) return FMHR_OK;
    hipError_t e = hipMemsetAsync(nsum, 0, (size_t)N * 12, st);
    if (e != hipSuccess) return hip_fail(e, "memset nsum");
    if (F > 0) {
        k_normals_faces<<<cdiv((long long)nb * F, kBlock), kBlock, 0, st>>>(verts, tri, nb, V, F, nsum);
        FMHR_CHECK_LAUNCH("k_normals_faces");
    }
    k_normals_normalize<<<cdiv(N, kBlock), kBlock, 0, st>>>(nsum, N, normals);
    FMHR_CHECK_LAUNCH("k_normals_normalize");
    return FMHR_OK;
}

int normals_bwd_impl(const float* verts, const int32_t* tri, const float* nsum, const float* gn, int gstride, int nb, int V,
                     int F, float* gtmp, float* gverts, hipStream_t st)
{
    long long N = (long long)nb * V;
    if (N == 0 || F == 0) return FMHR_OK;
    k_normals_bwd_vertex<<<cdiv(N, kBlock), kBlock, 0, st>>>(nsum, gn, gstride, N, gtmp);
    FMHR_CHECK_LAUNCH("k_normals_bwd_vertex");
    k_normals_bwd_faces<<<cdiv((long long)nb * F, kBlock), kBlock, 0, st>>>(verts, tri, gtmp, nb, V, F, gverts);
    FMHR_CHECK_LAUNCH("k_normals_bwd_faces");
    return FMHR_OK;
}

int lap_fwd_impl(const float* x, const int32_t* row_ptr, const int32_t* col, int V, int C, float weight, float* lx,
                 float* loss, hipStream_t st)
{
    if (V == 0) return FMHR_OK;
    float scale = weight / (float)V;
    int grid = cdiv(V, kBlock);
    switch (C) {
        case 1: k_lap_fwd<1><<<grid, kBlock, 0, st>>>(x, row_ptr, col, V, scale, lx, loss); break;
        case 2: k_lap_fwd<2><<<grid, kBlock, 0, st>>>(x, row_ptr, col, V, scale, lx, loss); break;
        case 3: k_lap_fwd<3><<<grid, kBlock, 0, st>>>(x, row_ptr, col, V, scale, lx, loss); break;
        case 4: k_lap_fwd<4><<<grid, kBlock, 0, st>>>(x, row_ptr, col, V, scale, lx, loss); break;
        default: set_error("laplacian: C must be 1..4 (got %d)", C); return FMHR_E_ARG;
    }
    FMHR_CHECK_LAUNCH("k_lap_fwd");
    return FMHR_OK;
}

int lap_bwd_impl(const float* lx, const int32_t* row_ptr, const int32_t* col, int V, int C, float gscale, float* rowg,
                 float* gx, hipStream_t st)
{
    if (V == 0) return FMHR_OK;
    float gs = gscale / (float)V;
    int grid = cdiv(V, kBlock);
    switch (C) {
        case 1: k_lap_bwd_rows<1><<<grid, kBlock, 0, st>>>(lx, V, gs, rowg); k_lap_bwd_gather<1><<<grid, kBlock, 0, st>>>(rowg, row_ptr, col, V, gx); break;
        case 2: k_lap_bwd_rows<2><<<grid, kBlock, 0, st>>>(lx, V, gs, rowg); k_lap_bwd_gather<2><<<grid, kBlock, 0, st>>>(rowg, row_ptr, col, V, gx); break;
        case 3: k_lap_bwd_rows<3><<<grid, kBlock, 0, st>>>(lx, V, gs, rowg); k_lap_bwd_gather<3><<<grid, kBlock, 0, st>>>(rowg, row_ptr, col, V, gx); break;
        case 4: k_lap_bwd_rows<4><<<grid, kBlock, 0, st>>>(lx, V, gs, rowg); k_lap_bwd_gather<4><<<grid, kBlock, 0, st>>>(rowg, row_ptr, col, V, gx); break;
        default: set_error("laplacian: C must be 1..4 (got %d)", C); return FMHR_E_ARG;
    }
    FMHR_CHECK_LAUNCH("k_lap_bwd");
    return FMHR_OK;
}

// ONE launch for the step's geometry: the first rows of the grid form the vertex normals (8 lanes per vertex) and store vertex =
// tmp + delta, the n rows after them transform the vertices into the n views (clip positions, snapped vertices, combined matrices) —
// the two parts are independent, so the latency-bound normals walk hides under the bandwidth-bound transform
template <bool HAS_DELTA>
__global__ void __launch_bounds__(kBlock)
k_geometry(const float* __restrict__ verts_tmp, const float* __restrict__ delta, const float* __restrict__ w2c,
           const float* __restrict__ proj, int n, int V, float4* __restrict__ pos, int H, int W, int2* __restrict__ snap,
           float4* __restrict__ cmat, const int32_t* __restrict__ tri, const int32_t* __restrict__ vf_ptr,
           const int32_t* __restrict__ vf_idx, float* __restrict__ nsum, float* __restrict__ normals,
           const float* __restrict__ albedo, float4* __restrict__ attr8, float* __restrict__ vout, int* __restrict__ stamp,
           const int32_t* __restrict__ view_idx, const float* __restrict__ sh, float* __restrict__ sh_local, int n_src, int32_t* __restrict__ view_idx_ws)
{
    // the step counter that tags this step's coverage entries and owner marks (raster.hip, shade.hip): advanced by the step's first kernel
    if (stamp != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        stamp[0] = stamp[0] + 1;
        stamp[2] = 0;                                                // cursor of the step's list of segments with fragments (raster.hip: k_bins_compact)
    }
    const int rows_normals = (int)gridDim.y - n;                     // dispatched first: their long dependent chains start early
    if ((int)blockIdx.y >= rows_normals)
        transform_body<HAS_DELTA>(blockIdx.x * kBlock + threadIdx.x, (int)blockIdx.y - rows_normals, verts_tmp, delta, w2c, proj, V, pos, H, W, snap,
                       cmat, view_idx, sh, sh_local, n_src, view_idx_ws, stamp);
    else
        normals_gather_body<HAS_DELTA>((int)blockIdx.y * gridDim.x + blockIdx.x, verts_tmp, tri, vf_ptr, vf_idx, V, nsum, normals, albedo, attr8,
                            delta, vout);
}

// inspection helper (fmhr_ham_densify): the clip positions the fused step never stores, from its vertices and combined matrices
// with the same chains every consumer uses (common.h: ClipView)
__global__ void __launch_bounds__(kBlock)
k_clip_positions(const float* __restrict__ verts, const float4* __restrict__ cmat, int n, int V, float4* __restrict__ pos)
{
    const int v = blockIdx.x * kBlock + threadIdx.x, b = blockIdx.y;
    if (v >= V) return;
    pos[(size_t)b * V + v] = clip_view(nullptr, verts, cmat, b, V).get(v);
}

int clip_positions_impl(const float* verts, const float* cmat, int n, int V, float* pos, hipStream_t st)
{
    if (n == 0 || V == 0) return FMHR_OK;
    k_clip_positions<<<dim3(cdiv(V, kBlock), n), kBlock, 0, st>>>(verts, (const float4*)cmat, n, V, (float4*)pos);
    FMHR_CHECK_LAUNCH("k_clip_positions");
    return FMHR_OK;
}

int geometry_impl(const float* verts_tmp, const float* delta, const float* w2c, const float* proj, int n, int V, float* pos, int H, int W,
                  int2* snap, float* cmat, const int32_t* tri, const int32_t* vf_ptr, const int32_t* vf_idx, float* nsum,
                  float* normals, const float* albedo, float* attr8, float* vout, int* stamp, const int32_t* view_idx, const float* sh,
                  float* sh_local, int n_src, int32_t* view_idx_ws, hipStream_t st)
{
    FMHR_CHECK_ARG(view_idx == nullptr || (n_src >= 1 && view_idx_ws != nullptr && stamp != nullptr), "geometry: a view selection needs n_src >= 1 and its workspace copy");
    if (n == 0 || V == 0) return FMHR_OK;
    const int gx = cdiv(V, kBlock);
    const int rows_normals = cdiv(cdiv((long long)V * kSubLanes, kBlock), gx);
    if (delta != nullptr)
        k_geometry<true><<<dim3(gx, n + rows_normals), kBlock, 0, st>>>(verts_tmp, delta, w2c, proj, n, V, (float4*)pos, H, W, snap, (float4*)cmat, tri,
                                                                    vf_ptr, vf_idx, nsum, normals, albedo, (float4*)attr8, vout, stamp, view_idx, sh, sh_local, n_src, view_idx_ws);
    else
        k_geometry<false><<<dim3(gx, n + rows_normals), kBlock, 0, st>>>(verts_tmp, delta, w2c, proj, n, V, (float4*)pos, H, W, snap, (float4*)cmat, tri,
                                                                     vf_ptr, vf_idx, nsum, normals, albedo, (float4*)attr8, vout, stamp, view_idx, sh, sh_local, n_src, view_idx_ws);
    FMHR_CHECK_LAUNCH("k_geometry");
    return FMHR_OK;
}

}  // namespace fmhr

using namespace fmhr;

extern "C" int fmhr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                              float beta1, float beta2, float eps, int step, void* stream)
{
    FMHR_CHECK_ARG(n >= 0 && step >= 1, "adam_step: bad n/step");
    if (n == 0) return FMHR_OK;
    FMHR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamTensor T;
    T.p = param; T.g = grad; T.m = exp_avg; T.v = exp_avg_sq; T.n = n;
    T.beta1 = beta1; T.beta2 = beta2; T.eps = eps; T.step_size = (float)((double)lr / bc1); T.bc2_sqrt = (float)sqrt(bc2);
    k_adam<<<cdiv(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(T);
    FMHR_CHECK_LAUNCH("k_adam");
    return FMHR_OK;
}

static int adam_tensor(AdamTensor& T, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                       float beta1, float beta2, float eps, int step)
{
    FMHR_CHECK_ARG(n >= 0 && step >= 1, "adam_step: bad n=%lld / step=%d", n, step);
    FMHR_CHECK_ARG(n == 0 || (param && exp_avg && exp_avg_sq), "adam_step: null pointer");
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    T.p = param; T.g = grad; T.m = exp_avg; T.v = exp_avg_sq; T.n = n;
    T.beta1 = beta1; T.beta2 = beta2; T.eps = eps; T.step_size = (float)((double)lr / bc1); T.bc2_sqrt = (float)sqrt(bc2);
    return FMHR_OK;
}

namespace fmhr {
// device-side form of an optional fused optimiser update (null: switched off)
int adam_tensor_from(AdamTensor& T, const fmhr_adam_param* a, long long n)
{
    memset(&T, 0, sizeof(T));
    if (a == nullptr) return FMHR_OK;
    return adam_tensor(T, a->param, nullptr, a->exp_avg, a->exp_avg_sq, n, a->lr, a->beta1, a->beta2, a->eps, a->step);
}
}  // namespace fmhr

extern "C" int fmhr_adam_step2(float* param_a, const float* grad_a, float* exp_avg_a, float* exp_avg_sq_a, long long n_a,
                               float lr_a, float beta1_a, float beta2_a, float eps_a, int step_a, float* param_b,
                               const float* grad_b, float* exp_avg_b, float* exp_avg_sq_b, long long n_b, float lr_b,
                               float beta1_b, float beta2_b, float eps_b, int step_b, void* stream)
{
    AdamTensor A, B;
    int rc = adam_tensor(A, param_a, grad_a, exp_avg_a, exp_avg_sq_a, n_a, lr_a, beta1_a, beta2_a, eps_a, step_a);
    if (rc) return rc;
    rc = adam_tensor(B, param_b, grad_b, exp_avg_b, exp_avg_sq_b, n_b, lr_b, beta1_b, beta2_b, eps_b, step_b);
    if (rc) return rc;
    if (n_a + n_b == 0) return FMHR_OK;
    k_adam2<<<cdiv(n_a + n_b, kBlock), kBlock, 0, (hipStream_t)stream>>>(A, B);
    FMHR_CHECK_LAUNCH("k_adam2");
    return FMHR_OK;
}

extern "C" int fmhr_edge_multiplicity_build(const int32_t* tri, int F, int V, const int32_t* row_ptr, const int32_t* col_idx, int nnz,
                                            int32_t* col_mult, void* stream)
{
    FMHR_CHECK_ARG(F >= 0 && V >= 0 && nnz >= 0, "edge_multiplicity_build: bad dims");
    if (nnz == 0 || F == 0) return FMHR_OK;
    FMHR_CHECK_ARG(tri && row_ptr && col_idx && col_mult, "edge_multiplicity_build: null pointer");
    hipError_t e = hipMemsetAsync(col_mult, 0, (size_t)nnz * 4, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "memset col_mult");
    k_edge_multiplicity<<<cdiv(3ll * F, kBlock), kBlock, 0, (hipStream_t)stream>>>(tri, F, V, row_ptr, col_idx, col_mult);
    FMHR_CHECK_LAUNCH("k_edge_multiplicity");
    return FMHR_OK;
}

extern "C" int fmhr_vertex_normals_fwd(const float* verts, const int32_t* tri, int b, int V, int F, float* nsum, float* normals,
                                       void* stream)
{
    FMHR_CHECK_ARG(b >= 0 && V >= 0 && F >= 0, "vertex_normals_fwd: bad dims");
    if ((long long)b * V == 0) return FMHR_OK;
    FMHR_CHECK_ARG(verts && nsum && normals && (tri || F == 0), "vertex_normals_fwd: null pointer");
    FMHR_CHECK_ARG((long long)b * V < (1ll << 32), "vertex_normals_fwd: b * V must be below 2^32");
    return normals_fwd_impl(verts, tri, b, V, F, nsum, normals, (hipStream_t)stream);
}

extern "C" int fmhr_vertex_normals_bwd(const float* verts, const int32_t* tri, const float* nsum, const float* gnormals, int b,
                                       int V, int F, float* gtmp, float* gverts, void* stream)
{
    FMHR_CHECK_ARG(b >= 0 && V >= 0 && F >= 0, "vertex_normals_bwd: bad dims");
    if ((long long)b * V == 0) return FMHR_OK;
    FMHR_CHECK_ARG(verts && nsum && gnormals && gtmp && gverts && (tri || F == 0), "vertex_normals_bwd: null pointer");
    FMHR_CHECK_ARG((long long)b * V < (1ll << 32), "vertex_normals_bwd: b * V must be below 2^32");
    return normals_bwd_impl(verts, tri, nsum, gnormals, 3, b, V, F, gtmp, gverts, (hipStream_t)stream);
}

extern "C" int fmhr_sh_radiance_fwd(const float* coeff, const float* normal, int P, int per_pixel, float* out, void* stream)
{
    FMHR_CHECK_ARG(P >= 0, "sh_radiance_fwd: bad dims");
    if (P == 0) return FMHR_OK;
    FMHR_CHECK_ARG(coeff && normal && out, "sh_radiance_fwd: null pointer");
    k_sh_fwd<<<cdiv(P, kBlock), kBlock, 0, (hipStream_t)stream>>>(coeff, normal, P, per_pixel, out);
    FMHR_CHECK_LAUNCH("k_sh_fwd");
    return FMHR_OK;
}

extern "C" int fmhr_sh_radiance_bwd(const float* coeff, const float* normal, const float* dy, int P, int per_pixel,
                                    float* gcoeff, float* gnormal, void* stream)
{
    FMHR_CHECK_ARG(P >= 0, "sh_radiance_bwd: bad dims");
    if (P == 0) return FMHR_OK;
    FMHR_CHECK_ARG(coeff && normal && dy && gcoeff && gnormal, "sh_radiance_bwd: null pointer");
    k_sh_bwd<<<cdiv(P, kBlock), kBlock, 0, (hipStream_t)stream>>>(coeff, normal, dy, P, per_pixel, gcoeff, gnormal);
    FMHR_CHECK_LAUNCH("k_sh_bwd");
    return FMHR_OK;
}

extern "C" int fmhr_laplacian_uniform_fwd(const float* x, const int32_t* row_ptr, const int32_t* col_idx, int V, int C,
                                          float weight, float* lx, float* loss, void* stream)
{
    FMHR_CHECK_ARG(V >= 0 && C >= 1 && C <= 4, "laplacian_uniform_fwd: bad dims (C must be 1..4)");
    if (V == 0) return FMHR_OK;
    FMHR_CHECK_ARG(x && row_ptr && lx && loss, "laplacian_uniform_fwd: null pointer");
    return lap_fwd_impl(x, row_ptr, col_idx, V, C, weight, lx, loss, (hipStream_t)stream);
}

extern "C" int fmhr_laplacian_uniform_bwd(const float* lx, const int32_t* row_ptr, const int32_t* col_idx, int V, int C,
                                          float gscale, float* rowg, float* gx, void* stream)
{
    FMHR_CHECK_ARG(V >= 0 && C >= 1 && C <= 4, "laplacian_uniform_bwd: bad dims (C must be 1..4)");
    if (V == 0) return FMHR_OK;
    FMHR_CHECK_ARG(lx && row_ptr && rowg && gx, "laplacian_uniform_bwd: null pointer");
    return lap_bwd_impl(lx, row_ptr, col_idx, V, C, gscale, rowg, gx, (hipStream_t)stream);
}

// fmhr_amd/csrc/shade.hip — fused per-pixel / per-vertex kernels of the stage-2 HAM step for gfx950.
// One pass replaces the reference's interpolate(C=7) -> slices -> F.normalize -> boolean gather -> get_radiance ->
// multiply -> scatter chain (/root/reference/mesh_sfs_optim.py:268-286), one pass the two loss reductions (:289,:295).
// The backward is GATHER-style: a per-pixel pass writes one 32-B gradient record per covered pixel, then one lane per
// (view, vertex) walks the pixel lists of its incident faces (threaded by the rasterizer) and sums its own gradient — no
// float atomics (the scatter form ran at the 0.08 TB/s "64 lanes -> 64 rows" atomic shape: 3.6 ms per step).
#include "common.h"

namespace fmhr {

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

__device__ __forceinline__ void shade_eval(const float4 r, const int32_t* __restrict__ tri, const float* __restrict__ normals,
                                           const float* __restrict__ albedo, const float* __restrict__ shc, PixShade& s)
{
    int t = (int)r.w - 1;
    s.i0 = tri[t * 3]; s.i1 = tri[t * 3 + 1]; s.i2 = tri[t * 3 + 2];
    s.b0 = r.x; s.b1 = r.y; s.b2 = 1.0f - r.x - r.y;
    s.n0 = ld3f(normals + (size_t)s.i0 * 3); s.n1 = ld3f(normals + (size_t)s.i1 * 3); s.n2 = ld3f(normals + (size_t)s.i2 * 3);
    s.a0 = ld3f(albedo + (size_t)s.i0 * 3); s.a1 = ld3f(albedo + (size_t)s.i1 * 3); s.a2 = ld3f(albedo + (size_t)s.i2 * 3);
    s.nx = s.b0 * s.n0.x + s.b1 * s.n1.x + s.b2 * s.n2.x;
    s.ny = s.b0 * s.n0.y + s.b1 * s.n1.y + s.b2 * s.n2.y;
    s.nz = s.b0 * s.n0.z + s.b1 * s.n1.z + s.b2 * s.n2.z;
    s.ar = s.b0 * s.a0.x + s.b1 * s.a1.x + s.b2 * s.a2.x;
    s.ag = s.b0 * s.a0.y + s.b1 * s.a1.y + s.b2 * s.a2.y;
    s.ab = s.b0 * s.a0.z + s.b1 * s.a1.z + s.b2 * s.a2.z;
    s.len = sqrtf(s.nx * s.nx + s.ny * s.ny + s.nz * s.nz);
    float inv = 1.0f / fmaxf(s.len, 1e-12f);                 // F.normalize(p=2, dim=3), default eps (:273)
    s.hx = s.nx * inv; s.hy = s.ny * inv; s.hz = s.nz * inv;
    s.rad = sh9c(shc, s.hx, s.hy, s.hz);
}

// ------------------------------------------------------------------------------------------------ segment sparsity
// The flattened pixel index [n*H*W] is cut into 64-pixel segments = exactly one wave of every per-pixel kernel.  A segment is
// ACTIVE when it holds a covered pixel, a 4-neighbour of one (antialias pairs and blend targets), or a non-zero valid_mask
// (silhouette MSE).  Per-pixel kernels return after one scalar load on inactive segments (~2/3 of a 19 %-coverage frame);
// every buffer is fully (re)written inside active segments each step and never read outside them.
__device__ __forceinline__ bool cov_range_any(const unsigned long long* __restrict__ covw, long long start, long long nSeg)
{
    // any covered pixel with flattened index in [start, start+63]?
    if (start + 63 < 0) return false;
    long long w0 = start >> 6;              // arithmetic shift: floor
    int sh = (int)(start & 63);
    bool a = false;
    if (w0 >= 0 && w0 < nSeg) a = (covw[w0] >> sh) != 0ull;
    if (sh != 0 && w0 + 1 >= 0 && w0 + 1 < nSeg) a = a || ((covw[w0 + 1] << (64 - sh)) != 0ull);
    return a;
}

__global__ void __launch_bounds__(kBlock)
k_active_segments(const unsigned long long* __restrict__ covw, const float* __restrict__ valid_mask, long long N, int W,
                  unsigned char* __restrict__ act)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid - (gid & 63) >= N) return;
    float vm = (gid < N) ? valid_mask[gid] : 0.f;
    unsigned long long bal = __ballot(vm != 0.f);
    if ((threadIdx.x & 63) != 0) return;
    long long seg = gid >> 6, nSeg = (N + 63) >> 6, s0 = seg << 6;
    bool a = bal != 0ull || covw[seg] != 0ull;
    if (!a) a = (seg + 1 < nSeg && (covw[seg + 1] & 1ull)) || (seg > 0 && (covw[seg - 1] >> 63));
    if (!a) a = cov_range_any(covw, s0 + W, nSeg) || cov_range_any(covw, s0 - W, nSeg);
    act[seg] = a ? 1 : 0;
}

__device__ __forceinline__ bool seg_active(const unsigned char* __restrict__ act, long long gid)
{
    return __builtin_amdgcn_readfirstlane((int)act[gid >> 6]) != 0;
}

// test/inspection helper: make the sparse buffers dense (zeros wherever the step did not write)
__global__ void __launch_bounds__(kBlock)
k_densify(const unsigned long long* __restrict__ covw, const unsigned char* __restrict__ act, long long N,
          float4* __restrict__ rast, float4* __restrict__ shaded, float4* __restrict__ aa)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= N) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    bool covered = (covw[gid >> 6] >> (gid & 63)) & 1ull;
    if (!covered) rast[gid] = z;
    if (!act[gid >> 6]) { shaded[gid] = z; aa[gid] = z; }
}

// forward: rast -> rgba (rgb = shaded image, zero outside valid = mask>0 & covered; a = the interpolated ones attribute),
// written twice (the second copy is the antialias accumulation target); local valid count
__global__ void __launch_bounds__(kBlock)
k_shade_fwd(const float4* __restrict__ rast, const int32_t* __restrict__ tri, const float* __restrict__ normals,
            const float* __restrict__ albedo, const float* __restrict__ sh, const float* __restrict__ mask,
            int n, int F, long long P, const unsigned char* __restrict__ act, const unsigned long long* __restrict__ covw,
            float4* __restrict__ shaded, float4* __restrict__ aa_init, float* __restrict__ part_cnt)
{
    __shared__ float sm[4];
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    float cnt = 0.f;
    if (gid - (gid & 63) < (long long)n * P && seg_active(act, gid) && gid < (long long)n * P) {
        const bool covered = (covw[gid >> 6] >> (gid & 63)) & 1ull;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (covered) r = rast[gid];
        int t = (int)r.w - 1;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < F) {
            float w2 = 1.0f - r.x - r.y;
            o.w = r.x + r.y + w2;                                // interpolate of the ones attribute (:268-269)
            if (mask[gid] > 0.f) {
                int b = (int)(gid / P);
                float shc[9];
#pragma unroll
                for (int k = 0; k < 9; k++) shc[k] = sh[(size_t)b * 9 + k];
                PixShade s;
                shade_eval(r, tri, normals, albedo, shc, s);
                o.x = s.rad * s.ar; o.y = s.rad * s.ag; o.z = s.rad * s.ab;
                cnt = 1.f;
            }
        }
        shaded[gid] = o;
        aa_init[gid] = o;
    }
    float s = block_sum(cnt, sm);
    if (threadIdx.x == 0) part_cnt[blockIdx.x] = s;        // per-block partial: 49k same-address atomics serialise at ~10 ns each
}

// losses + their per-pixel gradients over the active segments:
// sfs = w * mean_valid |aa.rgb - img| (:289), mask = w * mean (aa.a - valid_mask)^2 (:295) — inactive segments contribute
// exactly 0 to both.  g [.,4] = d loss / d antialiased rgba, written twice (g_aa = what the antialias backward reads as dy,
// g_shaded = its accumulation target).
__global__ void __launch_bounds__(kBlock)
k_loss_pix(const float4* __restrict__ shaded, const float4* __restrict__ aa, const float* __restrict__ img,
           const float* __restrict__ mask, const float* __restrict__ valid_mask, long long N,
           const unsigned char* __restrict__ act, const float* __restrict__ n_valid_global, float w_sfs, float w_mask,
           float mse_den, float4* __restrict__ g_aa, float4* __restrict__ g_shaded, float* __restrict__ part_l1,
           float* __restrict__ part_l2)
{
    __shared__ float sm[4];
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    float l1 = 0.f, l2 = 0.f;
    float den = 3.0f * n_valid_global[0];
    float k1 = (den > 0.f) ? w_sfs / den : 0.f;
    float k2 = w_mask / mse_den;
    if (gid - (gid & 63) < N && seg_active(act, gid) && gid < N) {
        float4 a = aa[gid];
        float cov = shaded[gid].w;                            // != 0 exactly where the pixel is covered
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask[gid] > 0.f && cov != 0.f) {
            float d0 = a.x - img[gid * 3], d1 = a.y - img[gid * 3 + 1], d2 = a.z - img[gid * 3 + 2];
            l1 = fabsf(d0) + fabsf(d1) + fabsf(d2);
            g.x = (d0 > 0.f) ? k1 : ((d0 < 0.f) ? -k1 : 0.f);
            g.y = (d1 > 0.f) ? k1 : ((d1 < 0.f) ? -k1 : 0.f);
            g.z = (d2 > 0.f) ? k1 : ((d2 < 0.f) ? -k1 : 0.f);
        }
        float dm = a.w - valid_mask[gid];
        l2 = dm * dm;
        g.w = 2.0f * dm * k2;
        g_aa[gid] = g;
        g_shaded[gid] = g;
    }
    float s1 = block_sum(l1, sm);
    float s2 = block_sum(l2, sm);
    if (threadIdx.x == 0) { part_l1[blockIdx.x] = s1; part_l2[blockIdx.x] = s2; }
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

__global__ void __launch_bounds__(1024) k_reduce_count(const float* __restrict__ part, int nb, int* __restrict__ n_valid)
{
    __shared__ float sm[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 1024) acc += part[i];
    float r = block1024_sum(acc, sm);
    if (threadIdx.x == 0) n_valid[0] = (int)(r + 0.5f);
}

// per covered pixel: backward of shade + normalise down to the interpolated attributes, and the barycentric gradients.
// record = (g_normal_raw.xyz, g_albedo.x | g_albedo.yz, g_u, g_v), zeros for covered pixels outside the photometric mask.
__global__ void __launch_bounds__(kBlock)
k_pix_bwd(const float4* __restrict__ rast, const int32_t* __restrict__ tri, const float* __restrict__ normals,
          const float* __restrict__ albedo, const float* __restrict__ sh, const float* __restrict__ mask,
          const float4* __restrict__ g_shaded, int n, int F, long long P, const unsigned long long* __restrict__ covw,
          float4* __restrict__ rec)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (long long)n * P) return;
    if (!((covw[gid >> 6] >> (gid & 63)) & 1ull)) return;       // one 8-B word per wave: uncovered segments stop here
    float4 r = rast[gid];
    int t = (int)r.w - 1;
    if (t < 0 || t >= F) return;
    float4 g = g_shaded[gid];
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    if (mask[gid] > 0.f && (g.x != 0.f || g.y != 0.f || g.z != 0.f)) {
        int b = (int)(gid / P);
        float c[9];
#pragma unroll
        for (int k = 0; k < 9; k++) c[k] = sh[(size_t)b * 9 + k];
        PixShade s;
        shade_eval(r, tri, normals, albedo, c, s);
        float g_rad = g.x * s.ar + g.y * s.ag + g.z * s.ab;
        float ga0 = g.x * s.rad, ga1 = g.y * s.rad, ga2 = g.z * s.rad;
        float x = s.hx, y = s.hy, z = s.hz;
        float dx = c[3] + c[4] * y - 2.0f * c[6] * x + c[7] * z + 2.0f * c[8] * x;
        float dy = c[1] + c[4] * x + c[5] * z - 2.0f * c[6] * y - 2.0f * c[8] * y;
        float dz = c[2] + c[5] * y + 4.0f * c[6] * z + c[7] * x;
        float ghx = g_rad * dx, ghy = g_rad * dy, ghz = g_rad * dz;
        float gnx, gny, gnz;
        if (s.len > 1e-12f) {
            float inv = 1.0f / s.len;
            float dt = x * ghx + y * ghy + z * ghz;
            gnx = (ghx - x * dt) * inv; gny = (ghy - y * dt) * inv; gnz = (ghz - z * dt) * inv;
        } else {
            gnx = ghx * 1e12f; gny = ghy * 1e12f; gnz = ghz * 1e12f;
        }
        // interpolate backward w.r.t. the barycentrics; the ones channel contributes exactly 0
        float gu = gnx * (s.n0.x - s.n2.x) + gny * (s.n0.y - s.n2.y) + gnz * (s.n0.z - s.n2.z) +
                   ga0 * (s.a0.x - s.a2.x) + ga1 * (s.a0.y - s.a2.y) + ga2 * (s.a0.z - s.a2.z);
        float gv = gnx * (s.n1.x - s.n2.x) + gny * (s.n1.y - s.n2.y) + gnz * (s.n1.z - s.n2.z) +
                   ga0 * (s.a1.x - s.a2.x) + ga1 * (s.a1.y - s.a2.y) + ga2 * (s.a1.z - s.a2.z);
        r0 = make_float4(gnx, gny, gnz, ga0);
        r1 = make_float4(ga1, ga2, gu, gv);
    }
    rec[gid * 2] = r0;
    rec[gid * 2 + 1] = r1;
}

// one lane per (view, vertex): walk the incident face corners (CSR), each face's pixel list (head/next from the rasterizer),
// and sum this vertex's share: attribute gradients b_k * record (-> part[b,v,0:6]) and the rasterize-backward position
// gradient of corner k (-> gpos[b,v] +=, which already holds the antialias-backward atomics).
__global__ void __launch_bounds__(kBlock)
k_vertex_gather(const float4* __restrict__ rast, const float4* __restrict__ rec, const int* __restrict__ head,
                const int* __restrict__ next, const int32_t* __restrict__ tri, const int32_t* __restrict__ vf_ptr,
                const int32_t* __restrict__ vf_idx, const float* __restrict__ pos, int n, int V, int F, int H, int W,
                float4* __restrict__ part, float4* __restrict__ gpos)
{
    int v = blockIdx.x * kBlock + threadIdx.x;
    int b = blockIdx.y;
    if (v >= V) return;
    const long long P = (long long)H * W;
    const float xs = 2.0f / (float)W, ys = 2.0f / (float)H, xo = 1.0f / (float)W - 1.0f, yo = 1.0f / (float)H - 1.0f;
    float an0 = 0.f, an1 = 0.f, an2 = 0.f, aa0 = 0.f, aa1 = 0.f, aa2 = 0.f, gx = 0.f, gy = 0.f, gw = 0.f;
    const int* hb = head + (size_t)b * F;
    const float* vp = pos + (size_t)b * V * 4;
    for (int e = vf_ptr[v]; e < vf_ptr[v + 1]; e++) {
        int code = vf_idx[e];
        int f = code / 3, k = code - f * 3;
        int pix = hb[f];
        if (pix < 0) continue;
        int i0 = tri[f * 3], i1 = tri[f * 3 + 1], i2 = tri[f * 3 + 2];
        const float4 p0 = ldg4(vp + (size_t)i0 * 4), p1 = ldg4(vp + (size_t)i1 * 4), p2 = ldg4(vp + (size_t)i2 * 4);
        while (pix >= 0) {
            long long gid = (long long)b * P + pix;
            float4 r = rast[gid];
            float4 r0 = rec[gid * 2], r1 = rec[gid * 2 + 1];
            float bk = (k == 0) ? r.x : ((k == 1) ? r.y : 1.0f - r.x - r.y);
            an0 += bk * r0.x; an1 += bk * r0.y; an2 += bk * r0.z;
            aa0 += bk * r0.w; aa1 += bk * r1.x; aa2 += bk * r1.y;
            float gu = r1.z, gv = r1.w;
            if (gu != 0.f || gv != 0.f) {
                int py = pix / W, px = pix - py * W;
                float fx = xs * (float)px + xo, fy = ys * (float)py + yo;
                float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
                float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
                float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
                float e0 = p1x * p2y - p1y * p2x, e1 = p2x * p0y - p2y * p0x, e2 = p0x * p1y - p0y * p1x;
                float iw = 1.0f / (e0 + e1 + e2);
                float bb0 = e0 * iw, bb1 = e1 * iw;
                float gb0 = gu * iw, gb1 = gv * iw, gbb = gb0 * bb0 + gb1 * bb1;
                float qx, qy;
                if (k == 0) { qx = gbb * (p2y - p1y) - gb1 * p2y; qy = gbb * (p1x - p2x) + gb1 * p2x; }
                else if (k == 1) { qx = gbb * (p0y - p2y) + gb0 * p2y; qy = gbb * (p2x - p0x) - gb0 * p2x; }
                else { qx = gbb * (p1y - p0y) - gb0 * p1y + gb1 * p0y; qy = gbb * (p0x - p1x) + gb0 * p1x - gb1 * p0x; }
                gx += qx; gy += qy; gw += -fx * qx - fy * qy;
            }
            pix = next[gid];
        }
    }
    size_t o = (size_t)b * V + v;
    part[o * 2] = make_float4(an0, an1, an2, aa0);
    part[o * 2 + 1] = make_float4(aa1, aa2, 0.f, 0.f);
    if (gx != 0.f || gy != 0.f || gw != 0.f) {
        float4 g = gpos[o];
        g.x += gx; g.y += gy; g.w += gw;
        gpos[o] = g;
    }
}

// per vertex: sum the per-view partials (attribute grads) and apply the clip-transform backward to gpos (sum over views):
// gverts[v] += sum_b ((gpos[b,v] @ proj[b]^T) @ w2c[b]^T)[:3];  gnormal[v] = sum_b part.n;  galbedo[v] += sum_b part.a
__global__ void __launch_bounds__(kBlock)
k_reduce_views(const float4* __restrict__ part, const float4* __restrict__ gpos, const float* __restrict__ w2c,
               const float* __restrict__ proj, int n, int V, float* __restrict__ gverts, float* __restrict__ gnormal,
               float* __restrict__ galbedo)
{
    int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= V) return;
    float ax = 0.f, ay = 0.f, az = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int b = 0; b < n; b++) {
        size_t o = (size_t)b * V + v;
        float4 q0 = part[o * 2], q1 = part[o * 2 + 1];
        n0 += q0.x; n1 += q0.y; n2 += q0.z; a0 += q0.w; a1 += q1.x; a2 += q1.y;
        float4 g = gpos[o];
        if (g.x != 0.f || g.y != 0.f || g.w != 0.f || g.z != 0.f) {
            const float* A = w2c + (size_t)b * 16;
            const float* B = proj + (size_t)b * 16;
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = g.x * B[k * 4] + g.y * B[k * 4 + 1] + g.z * B[k * 4 + 2] + g.w * B[k * 4 + 3];
            ax += r[0] * A[0] + r[1] * A[1] + r[2] * A[2] + r[3] * A[3];
            ay += r[0] * A[4] + r[1] * A[5] + r[2] * A[6] + r[3] * A[7];
            az += r[0] * A[8] + r[1] * A[9] + r[2] * A[10] + r[3] * A[11];
        }
    }
    gverts[(size_t)v * 3] += ax; gverts[(size_t)v * 3 + 1] += ay; gverts[(size_t)v * 3 + 2] += az;
    gnormal[(size_t)v * 3] = n0; gnormal[(size_t)v * 3 + 1] = n1; gnormal[(size_t)v * 3 + 2] = n2;
    galbedo[(size_t)v * 3] += a0; galbedo[(size_t)v * 3 + 1] += a1; galbedo[(size_t)v * 3 + 2] += a2;
}

__global__ void __launch_bounds__(1024)
k_finish_losses(float* losses, const float* n_valid, const float* __restrict__ part_l1, const float* __restrict__ part_l2, int nb,
                float w_sfs, float w_mask, float mse_den)
{
    __shared__ float sm[16];
    float a1 = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < nb; i += 1024) { a1 += part_l1[i]; a2 += part_l2[i]; }
    float s1 = block1024_sum(a1, sm);
    float s2 = block1024_sum(a2, sm);
    if (threadIdx.x == 0) {
        float den = 3.0f * n_valid[0];
        losses[FMHR_LOSS_SFS] = (den > 0.f) ? s1 * (w_sfs / den) : 0.f;
        losses[FMHR_LOSS_MASK] = s2 * (w_mask / mse_den);
        losses[FMHR_LOSS_TOTAL] = losses[FMHR_LOSS_SFS] + losses[FMHR_LOSS_LAP] + losses[FMHR_LOSS_ALBEDO] +
                                  losses[FMHR_LOSS_MASK] + losses[FMHR_LOSS_DELTA] + losses[FMHR_LOSS_EDGE];
        losses[FMHR_LOSS_NVALID] = n_valid[0];
    }
}

int active_segments_impl(const unsigned long long* covw, const float* valid_mask, long long N, int W, unsigned char* act,
                         hipStream_t st)
{
    if (N == 0) return FMHR_OK;
    k_active_segments<<<cdiv(N, kBlock), kBlock, 0, st>>>(covw, valid_mask, N, W, act);
    FMHR_CHECK_LAUNCH("k_active_segments");
    return FMHR_OK;
}

int densify_impl(const unsigned long long* covw, const unsigned char* act, long long N, float* rast, float* shaded, float* aa,
                 hipStream_t st)
{
    if (N == 0) return FMHR_OK;
    k_densify<<<cdiv(N, kBlock), kBlock, 0, st>>>(covw, act, N, (float4*)rast, (float4*)shaded, (float4*)aa);
    FMHR_CHECK_LAUNCH("k_densify");
    return FMHR_OK;
}

int shade_fwd_impl(const float* rast, const int32_t* tri, const float* normals, const float* albedo, const float* sh,
                   const float* mask, int n, int F, int H, int W, const unsigned char* act, const unsigned long long* covw,
                   float* shaded, float* aa_init, float* part_cnt, int* n_valid, hipStream_t st)
{
    long long P = (long long)H * W;
    if (n * P == 0) return FMHR_OK;
    int nb = cdiv(n * P, kBlock);
    k_shade_fwd<<<nb, kBlock, 0, st>>>((const float4*)rast, tri, normals, albedo, sh, mask, n, F, P, act, covw,
                                       (float4*)shaded, (float4*)aa_init, part_cnt);
    FMHR_CHECK_LAUNCH("k_shade_fwd");
    k_reduce_count<<<1, 1024, 0, st>>>(part_cnt, nb, n_valid);
    FMHR_CHECK_LAUNCH("k_reduce_count");
    return FMHR_OK;
}

int loss_pix_impl(const float* shaded, const float* aa, const float* img, const float* mask, const float* valid_mask,
                  long long N, const unsigned char* act, const float* n_valid_global, float w_sfs, float w_mask, float mse_den,
                  float* g_aa, float* g_shaded, float* part_l1, float* part_l2, hipStream_t st)
{
    if (N == 0) return FMHR_OK;
    k_loss_pix<<<cdiv(N, kBlock), kBlock, 0, st>>>((const float4*)shaded, (const float4*)aa, img, mask, valid_mask, N, act,
                                                  n_valid_global, w_sfs, w_mask, mse_den, (float4*)g_aa, (float4*)g_shaded, part_l1,
                                                  part_l2);
    FMHR_CHECK_LAUNCH("k_loss_pix");
    return FMHR_OK;
}

int pix_bwd_impl(const float* rast, const int32_t* tri, const float* normals, const float* albedo, const float* sh,
                 const float* mask, const float* g_shaded, int n, int F, int H, int W, const unsigned long long* covw,
                 float* rec, hipStream_t st)
{
    long long P = (long long)H * W;
    if (n * P == 0 || F == 0) return FMHR_OK;
    k_pix_bwd<<<cdiv(n * P, kBlock), kBlock, 0, st>>>((const float4*)rast, tri, normals, albedo, sh, mask, (const float4*)g_shaded,
                                                     n, F, P, covw, (float4*)rec);
    FMHR_CHECK_LAUNCH("k_pix_bwd");
    return FMHR_OK;
}

int vertex_gather_impl(const float* rast, const float* rec, const int* head, const int* next, const int32_t* tri,
                       const int32_t* vf_ptr, const int32_t* vf_idx, const float* pos, int n, int V, int F, int H, int W,
                       float* part, float* gpos, hipStream_t st)
{
    if (n == 0 || V == 0) return FMHR_OK;
    k_vertex_gather<<<dim3(cdiv(V, kBlock), n), kBlock, 0, st>>>((const float4*)rast, (const float4*)rec, head, next, tri, vf_ptr,
                                                                vf_idx, pos, n, V, F, H, W, (float4*)part, (float4*)gpos);
    FMHR_CHECK_LAUNCH("k_vertex_gather");
    return FMHR_OK;
}

int reduce_views_impl(const float* part, const float* gpos, const float* w2c, const float* proj, int n, int V, float* gverts,
                      float* gnormal, float* galbedo, hipStream_t st)
{
    if (V == 0) return FMHR_OK;
    k_reduce_views<<<cdiv(V, kBlock), kBlock, 0, st>>>((const float4*)part, (const float4*)gpos, w2c, proj, n, V, gverts, gnormal,
                                                      galbedo);
    FMHR_CHECK_LAUNCH("k_reduce_views");
    return FMHR_OK;
}

int finish_losses_impl(float* losses, const float* n_valid, const float* part_l1, const float* part_l2, int nb, float w_sfs,
                       float w_mask, float mse_den, hipStream_t st)
{
    k_finish_losses<<<1, 1024, 0, st>>>(losses, n_valid, part_l1, part_l2, nb, w_sfs, w_mask, mse_den);
    FMHR_CHECK_LAUNCH("k_finish_losses");
    return FMHR_OK;
}

}  // namespace fmhr

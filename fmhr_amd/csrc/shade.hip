// fmhr_amd/csrc/shade.hip — fused per-pixel kernels of the stage-2 HAM step for gfx950.
// One pass replaces the reference's interpolate(C=7) -> slices -> F.normalize -> boolean gather -> get_radiance ->
// multiply -> scatter chain (/root/reference/mesh_sfs_optim.py:268-286), one pass the two loss reductions (:289,:295),
// and one pass the backward of all of it down to per-vertex attribute gradients and clip-space position gradients.
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
};

__device__ __forceinline__ void shade_eval(const float4 r, const int32_t* __restrict__ tri, const float* __restrict__ normals,
                                           const float* __restrict__ albedo, const float* __restrict__ shc, PixShade& s)
{
    int t = (int)r.w - 1;
    s.i0 = tri[t * 3]; s.i1 = tri[t * 3 + 1]; s.i2 = tri[t * 3 + 2];
    s.b0 = r.x; s.b1 = r.y; s.b2 = 1.0f - r.x - r.y;
    const float* n0 = normals + (size_t)s.i0 * 3; const float* n1 = normals + (size_t)s.i1 * 3; const float* n2 = normals + (size_t)s.i2 * 3;
    const float* a0 = albedo + (size_t)s.i0 * 3; const float* a1 = albedo + (size_t)s.i1 * 3; const float* a2 = albedo + (size_t)s.i2 * 3;
    s.nx = s.b0 * n0[0] + s.b1 * n1[0] + s.b2 * n2[0];
    s.ny = s.b0 * n0[1] + s.b1 * n1[1] + s.b2 * n2[1];
    s.nz = s.b0 * n0[2] + s.b1 * n1[2] + s.b2 * n2[2];
    s.ar = s.b0 * a0[0] + s.b1 * a1[0] + s.b2 * a2[0];
    s.ag = s.b0 * a0[1] + s.b1 * a1[1] + s.b2 * a2[1];
    s.ab = s.b0 * a0[2] + s.b1 * a1[2] + s.b2 * a2[2];
    s.len = sqrtf(s.nx * s.nx + s.ny * s.ny + s.nz * s.nz);
    float inv = 1.0f / fmaxf(s.len, 1e-12f);                 // F.normalize(p=2, dim=3), default eps (:273)
    s.hx = s.nx * inv; s.hy = s.ny * inv; s.hz = s.nz * inv;
    s.rad = sh9c(shc, s.hx, s.hy, s.hz);
}

// forward: rast -> shaded image (zero outside valid = mask>0 & covered), coverage colour, local valid count
__global__ void __launch_bounds__(kBlock)
k_shade_fwd(const float4* __restrict__ rast, const int32_t* __restrict__ tri, const float* __restrict__ normals,
            const float* __restrict__ albedo, const float* __restrict__ sh, const float* __restrict__ mask,
            int n, int F, long long P, float* __restrict__ tmp_img, float* __restrict__ mask_col, int* __restrict__ n_valid)
{
    __shared__ float sm[4];
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    float cnt = 0.f;
    if (gid < (long long)n * P) {
        float4 r = rast[gid];
        int t = (int)r.w - 1;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, mc = 0.f;
        if (t >= 0 && t < F) {
            float w2 = 1.0f - r.x - r.y;
            mc = r.x + r.y + w2;                                 // interpolate of the ones attribute (:268-269)
            if (mask[gid] > 0.f) {
                int b = (int)(gid / P);
                float shc[9];
#pragma unroll
                for (int k = 0; k < 9; k++) shc[k] = sh[(size_t)b * 9 + k];
                PixShade s;
                shade_eval(r, tri, normals, albedo, shc, s);
                o0 = s.rad * s.ar; o1 = s.rad * s.ag; o2 = s.rad * s.ab;
                cnt = 1.f;
            }
        }
        tmp_img[gid * 3] = o0; tmp_img[gid * 3 + 1] = o1; tmp_img[gid * 3 + 2] = o2;
        mask_col[gid] = mc;
    }
    float s = block_sum(cnt, sm);
    if (threadIdx.x == 0 && s != 0.f) atomicAdd(n_valid, (int)s);
}

// losses + their per-pixel gradients: sfs = w * mean_valid |aa_img - img| (:289), mask = w * mean (aa_mask - valid_mask)^2 (:295)
__global__ void __launch_bounds__(kBlock)
k_loss_pix(const float4* __restrict__ rast, const float* __restrict__ aa_img, const float* __restrict__ img,
           const float* __restrict__ mask, const float* __restrict__ aa_mask, const float* __restrict__ valid_mask,
           long long N, const float* __restrict__ n_valid_global, float w_sfs, float w_mask, float mse_den,
           float* __restrict__ g_aa_img, float* __restrict__ g_aa_mask, float* __restrict__ losses)
{
    __shared__ float sm[4];
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    float l1 = 0.f, l2 = 0.f;
    float den = 3.0f * n_valid_global[0];
    float k1 = (den > 0.f) ? w_sfs / den : 0.f;
    float k2 = w_mask / mse_den;
    if (gid < N) {
        bool valid = (mask[gid] > 0.f) && (rast[gid].w > 0.f);
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (valid) {
            float d0 = aa_img[gid * 3] - img[gid * 3], d1 = aa_img[gid * 3 + 1] - img[gid * 3 + 1], d2 = aa_img[gid * 3 + 2] - img[gid * 3 + 2];
            l1 = fabsf(d0) + fabsf(d1) + fabsf(d2);
            g0 = (d0 > 0.f) ? k1 : ((d0 < 0.f) ? -k1 : 0.f);
            g1 = (d1 > 0.f) ? k1 : ((d1 < 0.f) ? -k1 : 0.f);
            g2 = (d2 > 0.f) ? k1 : ((d2 < 0.f) ? -k1 : 0.f);
        }
        g_aa_img[gid * 3] = g0; g_aa_img[gid * 3 + 1] = g1; g_aa_img[gid * 3 + 2] = g2;
        float dm = aa_mask[gid] - valid_mask[gid];
        l2 = dm * dm;
        g_aa_mask[gid] = 2.0f * dm * k2;
    }
    float s1 = block_sum(l1, sm);
    float s2 = block_sum(l2, sm);
    if (threadIdx.x == 0) {
        if (s1 != 0.f) atomicAdd(losses + FMHR_LOSS_SFS, s1 * k1);
        if (s2 != 0.f) atomicAdd(losses + FMHR_LOSS_MASK, s2 * k2);
    }
}

// backward of shade + interpolate + rasterize for one pixel: attribute grads into gattr[V,8] (0..2 normal, 3..5 albedo),
// clip-space position grads into gpos[n,V,4]
__global__ void __launch_bounds__(kBlock)
k_shade_bwd(const float4* __restrict__ rast, const int32_t* __restrict__ tri, const float* __restrict__ pos,
            const float* __restrict__ normals, const float* __restrict__ albedo, const float* __restrict__ sh,
            const float* __restrict__ mask, const float* __restrict__ g_img, int n, int V, int F, int H, int W,
            float* __restrict__ gattr, float* __restrict__ gpos)
{
    long long P = (long long)H * W;
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (long long)n * P) return;
    float4 r = rast[gid];
    int t = (int)r.w - 1;
    if (t < 0 || t >= F || !(mask[gid] > 0.f)) return;
    float g0 = g_img[gid * 3], g1 = g_img[gid * 3 + 1], g2 = g_img[gid * 3 + 2];
    if (g0 == 0.f && g1 == 0.f && g2 == 0.f) return;
    int b = (int)(gid / P);
    int rem = (int)(gid - (long long)b * P);
    int py = rem / W, px = rem - py * W;
    float c[9];
#pragma unroll
    for (int k = 0; k < 9; k++) c[k] = sh[(size_t)b * 9 + k];
    PixShade s;
    shade_eval(r, tri, normals, albedo, c, s);
    // pred = rad * albedo
    float g_rad = g0 * s.ar + g1 * s.ag + g2 * s.ab;
    float ga0 = g0 * s.rad, ga1 = g1 * s.rad, ga2 = g2 * s.rad;
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
    const float gch[6] = {gnx, gny, gnz, ga0, ga1, ga2};
    const int idx[3] = {s.i0, s.i1, s.i2};
    const float bw[3] = {s.b0, s.b1, s.b2};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float* q = gattr + (size_t)idx[k] * 8;
#pragma unroll
        for (int ch = 0; ch < 6; ch++) atomicAdd(q + ch, bw[k] * gch[ch]);
    }
    // barycentric gradients (interpolate backward), then rasterize backward
    const float* n0 = normals + (size_t)s.i0 * 3; const float* n1 = normals + (size_t)s.i1 * 3; const float* n2 = normals + (size_t)s.i2 * 3;
    const float* a0 = albedo + (size_t)s.i0 * 3; const float* a1 = albedo + (size_t)s.i1 * 3; const float* a2 = albedo + (size_t)s.i2 * 3;
    float gu = 0.f, gv = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        gu += gch[ch] * (n0[ch] - n2[ch]) + gch[3 + ch] * (a0[ch] - a2[ch]);
        gv += gch[ch] * (n1[ch] - n2[ch]) + gch[3 + ch] * (a1[ch] - a2[ch]);
    }
    if (gu == 0.f && gv == 0.f) return;
    const float* vp = pos + (size_t)b * V * 4;
    const float4 p0 = ldg4(vp + (size_t)s.i0 * 4), p1 = ldg4(vp + (size_t)s.i1 * 4), p2 = ldg4(vp + (size_t)s.i2 * 4);
    float fx = (2.0f / (float)W) * (float)px + (1.0f / (float)W - 1.0f);
    float fy = (2.0f / (float)H) * (float)py + (1.0f / (float)H - 1.0f);
    float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    float e0 = p1x * p2y - p1y * p2x, e1 = p2x * p0y - p2y * p0x, e2 = p0x * p1y - p0y * p1x;
    float iw = 1.0f / (e0 + e1 + e2);
    float bb0 = e0 * iw, bb1 = e1 * iw;
    float gb0 = gu * iw, gb1 = gv * iw, gbb = gb0 * bb0 + gb1 * bb1;
    float q0x = gbb * (p2y - p1y) - gb1 * p2y;
    float q1x = gbb * (p0y - p2y) + gb0 * p2y;
    float q2x = gbb * (p1y - p0y) - gb0 * p1y + gb1 * p0y;
    float q0y = gbb * (p1x - p2x) + gb1 * p2x;
    float q1y = gbb * (p2x - p0x) - gb0 * p2x;
    float q2y = gbb * (p0x - p1x) + gb0 * p1x - gb1 * p0x;
    float* o0 = gpos + ((size_t)b * V + s.i0) * 4;
    float* o1 = gpos + ((size_t)b * V + s.i1) * 4;
    float* o2 = gpos + ((size_t)b * V + s.i2) * 4;
    atomicAdd(o0 + 0, q0x); atomicAdd(o0 + 1, q0y); atomicAdd(o0 + 3, -fx * q0x - fy * q0y);
    atomicAdd(o1 + 0, q1x); atomicAdd(o1 + 1, q1y); atomicAdd(o1 + 3, -fx * q1x - fy * q1y);
    atomicAdd(o2 + 0, q2x); atomicAdd(o2 + 1, q2y); atomicAdd(o2 + 3, -fx * q2x - fy * q2y);
}

// galbedo[v,:] = gattr[v,3:6] (+ existing), used after the pixel pass
__global__ void __launch_bounds__(kBlock)
k_unpack_albedo_grad(const float* __restrict__ gattr, int V, float* __restrict__ galbedo)
{
    int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= V) return;
    galbedo[(size_t)v * 3] += gattr[(size_t)v * 8 + 3];
    galbedo[(size_t)v * 3 + 1] += gattr[(size_t)v * 8 + 4];
    galbedo[(size_t)v * 3 + 2] += gattr[(size_t)v * 8 + 5];
}

__global__ void k_finish_losses(float* losses, const float* n_valid)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        losses[FMHR_LOSS_TOTAL] = losses[FMHR_LOSS_SFS] + losses[FMHR_LOSS_LAP] + losses[FMHR_LOSS_ALBEDO] +
                                  losses[FMHR_LOSS_MASK] + losses[FMHR_LOSS_DELTA] + losses[FMHR_LOSS_EDGE];
        losses[FMHR_LOSS_NVALID] = n_valid[0];
    }
}

int shade_fwd_impl(const float* rast, const int32_t* tri, const float* normals, const float* albedo, const float* sh,
                   const float* mask, int n, int F, int H, int W, float* tmp_img, float* mask_col, int* n_valid, hipStream_t st)
{
    long long P = (long long)H * W;
    if (n * P == 0) return FMHR_OK;
    k_shade_fwd<<<cdiv(n * P, kBlock), kBlock, 0, st>>>((const float4*)rast, tri, normals, albedo, sh, mask, n, F, P, tmp_img,
                                                       mask_col, n_valid);
    FMHR_CHECK_LAUNCH("k_shade_fwd");
    return FMHR_OK;
}

int loss_pix_impl(const float* rast, const float* aa_img, const float* img, const float* mask, const float* aa_mask,
                  const float* valid_mask, long long N, const float* n_valid_global, float w_sfs, float w_mask, float mse_den,
                  float* g_aa_img, float* g_aa_mask, float* losses, hipStream_t st)
{
    if (N == 0) return FMHR_OK;
    k_loss_pix<<<cdiv(N, kBlock), kBlock, 0, st>>>((const float4*)rast, aa_img, img, mask, aa_mask, valid_mask, N, n_valid_global,
                                                  w_sfs, w_mask, mse_den, g_aa_img, g_aa_mask, losses);
    FMHR_CHECK_LAUNCH("k_loss_pix");
    return FMHR_OK;
}

int shade_bwd_impl(const float* rast, const int32_t* tri, const float* pos, const float* normals, const float* albedo,
                   const float* sh, const float* mask, const float* g_img, int n, int V, int F, int H, int W, float* gattr,
                   float* gpos, hipStream_t st)
{
    long long nP = (long long)n * H * W;
    if (nP == 0 || F == 0) return FMHR_OK;
    k_shade_bwd<<<cdiv(nP, kBlock), kBlock, 0, st>>>((const float4*)rast, tri, pos, normals, albedo, sh, mask, g_img, n, V, F, H, W,
                                                    gattr, gpos);
    FMHR_CHECK_LAUNCH("k_shade_bwd");
    return FMHR_OK;
}

int unpack_albedo_grad_impl(const float* gattr, int V, float* galbedo, hipStream_t st)
{
    if (V == 0) return FMHR_OK;
    k_unpack_albedo_grad<<<cdiv(V, kBlock), kBlock, 0, st>>>(gattr, V, galbedo);
    FMHR_CHECK_LAUNCH("k_unpack_albedo_grad");
    return FMHR_OK;
}

int finish_losses_impl(float* losses, const float* n_valid, hipStream_t st)
{
    k_finish_losses<<<1, 64, 0, st>>>(losses, n_valid);
    FMHR_CHECK_LAUNCH("k_finish_losses");
    return FMHR_OK;
}

}  // namespace fmhr

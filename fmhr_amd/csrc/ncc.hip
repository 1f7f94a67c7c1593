// fmhr_amd/csrc/ncc.hip — masked normalised cross-correlation over patches, forward + backward, for gfx950.
// Replaces NCC(ref, src, ref_valid_mask, src_valid_mask) of /root/reference/models/ncc_utils.py:4-35 (row A12 of SURVEY.md §8a;
// the function is never called by the reference, but BASELINE.json's north_star names it).  One wave64 per (view, point):
// the Npx pixels of a patch are strided over the lanes and reduced with DPP shuffles — no LDS, no atomics in the forward.
#include "common.h"

namespace fmhr {

struct NccStats { float n, ma, mb, va, vb, cov, sm2a; };   // sm2a/sm2b = sum (x - mean) m^2 (non-binary masks)

__device__ __forceinline__ float wave_all_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// statistics of one (view, point) patch; every lane returns the same values
__device__ __forceinline__ void ncc_stats(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ m,
                                          int Npx, int lane, float& n, float& ma, float& mb, float& va, float& vb, float& cov,
                                          float& sa2, float& sb2)
{
    float sm = 0.f, sam = 0.f, sbm = 0.f;
    for (int j = lane; j < Npx; j += 64) { float mj = m[j]; sm += mj; sam += a[j] * mj; sbm += b[j] * mj; }
    sm = wave_all_sum(sm); sam = wave_all_sum(sam); sbm = wave_all_sum(sbm);
    n = (sm == 0.f) ? 1.f : sm;                                  // src_valid_num[src_valid_num == 0] = 1 (:16)
    ma = sam / n; mb = sbm / n;
    float qa = 0.f, qb = 0.f, qc = 0.f, ra = 0.f, rb = 0.f;
    for (int j = lane; j < Npx; j += 64) {
        float mj = m[j], da = a[j] - ma, db = b[j] - mb;
        float dam = da * mj, dbm = db * mj;
        qa += dam * dam; qb += dbm * dbm; qc += da * db * mj;
        ra += dam * mj; rb += dbm * mj;
    }
    va = wave_all_sum(qa) / n; vb = wave_all_sum(qb) / n; cov = wave_all_sum(qc) / n;
    sa2 = wave_all_sum(ra); sb2 = wave_all_sum(rb);
    if (va == 0.f) va = 1.f;                                     // tmp[ref_var == 0] = 1 (:20-22)
    if (vb == 0.f) vb = 1.f;                                     // (:27-29)
}

__global__ void __launch_bounds__(kBlock)
k_ncc_fwd(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ mask, int nv, int Np, int Npx,
          float* __restrict__ out)
{
    int wv = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wv >= nv * Np) return;
    int p = wv % Np;
    const float* a = ref + (size_t)p * Npx;
    const float* b = src + (size_t)wv * Npx;
    const float* m = mask + (size_t)wv * Npx;
    float n, ma, mb, va, vb, cov, sa2, sb2;
    ncc_stats(a, b, m, Npx, lane, n, ma, mb, va, vb, cov, sa2, sb2);
    if (lane == 0) out[wv] = cov / (sqrtf(va) * sqrtf(vb));
}

// d out / d ref (summed over views: float atomics into gref [1,Np,Npx]) and d out / d src
__global__ void __launch_bounds__(kBlock)
k_ncc_bwd(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ mask,
          const float* __restrict__ dy, int nv, int Np, int Npx, float* __restrict__ gref, float* __restrict__ gsrc)
{
    int wv = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wv >= nv * Np) return;
    int p = wv % Np;
    const float* a = ref + (size_t)p * Npx;
    const float* b = src + (size_t)wv * Npx;
    const float* m = mask + (size_t)wv * Npx;
    float n, ma, mb, va, vb, cov, sa2, sb2;
    ncc_stats(a, b, m, Npx, lane, n, ma, mb, va, vb, cov, sa2, sb2);
    // the zero-variance guard adds a constant: the raw variance still carries the gradient, as in torch
    float sa = sqrtf(va), sb = sqrtf(vb), g = dy[wv];
    float inv_ab = 1.0f / (sa * sb);
    float ka = cov / (2.0f * va * sa * sb), kb = cov / (2.0f * vb * sa * sb);   // d ncc / d va, d ncc / d vb (negated below)
    for (int j = lane; j < Npx; j += 64) {
        float mj = m[j], da = a[j] - ma, db = b[j] - mb;
        // d cov / d a_j = (db_j m_j - (m_j / n) sum_i db_i m_i) / n ;  d va / d a_j = (2/n) (da_j m_j^2 - (m_j / n) sum_i da_i m_i^2)
        float sdbm = 0.f, sdam = 0.f;   // sum_i d*_i m_i = 0 whenever n = sum m (or the mask is empty)
        float dcov_a = (db * mj - mj / n * sdbm) / n, dcov_b = (da * mj - mj / n * sdam) / n;
        float dva = 2.0f / n * (da * mj * mj - mj / n * sa2), dvb = 2.0f / n * (db * mj * mj - mj / n * sb2);
        float ga = g * (dcov_a * inv_ab - ka * dva), gb = g * (dcov_b * inv_ab - kb * dvb);
        if (gsrc != nullptr) gsrc[(size_t)wv * Npx + j] = gb;
        if (gref != nullptr && ga != 0.f) atomicAdd(gref + (size_t)p * Npx + j, ga);
    }
}

}  // namespace fmhr

using namespace fmhr;

extern "C" int fmhr_ncc_fwd(const float* ref, const float* src, const float* src_valid_mask, int nviews, int npoints, int npixels,
                            float* out, void* stream)
{
    FMHR_CHECK_ARG(nviews >= 0 && npoints >= 0 && npixels > 0, "ncc_fwd: bad dims");
    long long N = (long long)nviews * npoints;
    if (N == 0) return FMHR_OK;
    FMHR_CHECK_ARG(ref && src && src_valid_mask && out, "ncc_fwd: null pointer");
    k_ncc_fwd<<<cdiv(N * 64, kBlock), kBlock, 0, (hipStream_t)stream>>>(ref, src, src_valid_mask, nviews, npoints, npixels, out);
    FMHR_CHECK_LAUNCH("k_ncc_fwd");
    return FMHR_OK;
}

extern "C" int fmhr_ncc_bwd(const float* ref, const float* src, const float* src_valid_mask, const float* dy, int nviews,
                            int npoints, int npixels, float* gref, float* gsrc, void* stream)
{
    FMHR_CHECK_ARG(nviews >= 0 && npoints >= 0 && npixels > 0, "ncc_bwd: bad dims");
    long long N = (long long)nviews * npoints;
    if (N == 0) return FMHR_OK;
    FMHR_CHECK_ARG(ref && src && src_valid_mask && dy && (gref || gsrc), "ncc_bwd: null pointer");
    k_ncc_bwd<<<cdiv(N * 64, kBlock), kBlock, 0, (hipStream_t)stream>>>(ref, src, src_valid_mask, dy, nviews, npoints, npixels, gref,
                                                                        gsrc);
    FMHR_CHECK_LAUNCH("k_ncc_bwd");
    return FMHR_OK;
}

/*
 * oracle/raster_oracle.c — TEST INFRASTRUCTURE (see oracle/README.md).  Never linked into fmhr_amd.
 * CPU restatement of nvdiffrast's rasterize / interpolate / antialias as the reference calls them
 * (/root/reference/mesh_sfs_optim.py:267-287).  Parity against real nvdiffrast: UNPINNED.
 * Build: make -C oracle   (must keep -ffp-contract=off: the f32 twin is decision-exact with the HIP kernels).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#include <omp.h>

/* threads of the loops over views (this library links the system libgomp; torch carries a libgomp of its own, so
 * torch.set_num_threads does not reach these loops) */
void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_get_max_threads(void) { return omp_get_max_threads(); }

/* ---- topology: opp[f*3+k] = vertex opposite to edge k (= edge (v[k+1], v[k+2])) in the neighbouring face, or -1.
 * Edges are keyed by vertex index pairs (as nvdiffrast's edge hash is); an edge shared by more than two faces keeps the
 * two lowest codes 3*face+k (build-defined; nvdiffrast keeps whichever two won its hash insertion race). */
typedef struct { int64_t key; int32_t code; int32_t oppv; } edge_rec;
static int edge_cmp(const void* a, const void* b)
{
    const edge_rec* x = (const edge_rec*)a; const edge_rec* y = (const edge_rec*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->code < y->code ? -1 : (x->code > y->code);
}
int orc_topology_build(const int32_t* tri, int F, int V, int32_t* opp)
{
    edge_rec* e = (edge_rec*)malloc(sizeof(edge_rec) * (size_t)F * 3);
    if (!e) return -1;
    for (int f = 0; f < F; f++)
        for (int k = 0; k < 3; k++) {
            int64_t a = tri[f * 3 + (k + 1) % 3], b = tri[f * 3 + (k + 2) % 3];
            int64_t lo = a < b ? a : b, hi = a < b ? b : a;
            e[f * 3 + k].key = lo * (int64_t)V + hi;
            e[f * 3 + k].code = f * 3 + k;
            e[f * 3 + k].oppv = tri[f * 3 + k];
        }
    qsort(e, (size_t)F * 3, sizeof(edge_rec), edge_cmp);
    size_t i = 0, N = (size_t)F * 3;
    while (i < N) {
        size_t j = i;
        while (j < N && e[j].key == e[i].key) j++;
        int32_t oa = e[i].oppv, ob = (j - i >= 2) ? e[i + 1].oppv : -1;
        for (size_t m = i; m < j; m++) {
            int32_t vr = e[m].oppv;
            opp[e[m].code] = (oa == vr) ? ob : oa;
        }
        i = j;
    }
    free(e);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Rule R5, VALUES (round 5).  The fp32 twin keeps every DECISION of rule R5 in fp32 (accept / reject of a fragment,
 * its depth z/w and with it the depth order): those are bit-identical to the HIP kernels.  The barycentrics (u, v) it RETURNS, and
 * the gradient of (u, v) with respect to the clip positions, are evaluated in double from the same fp32 clip positions and rounded
 * once: nvdiffrast's pixel-relative edge functions x - fx w cancel ~10 bits at the reference capture's clip-space scale
 * (|x| ~ 2, a pixel 1.5e-3), and a thin triangle divides what is left by its area — in fp32 the position gradient of the thinnest
 * triangles was 8e-4 of the gradient's maximum away from exact arithmetic (profiles/parity_r04.txt).  The HIP kernels evaluate the
 * same statements in double (raster.hip: fragment_uv_wide, shade.hip: pixel_backward), one IEEE operation per statement.
 * orc_set_r5_wide(0) restores the all-fp32 evaluation (A/B: tools/experiments/r05_r5_wide.py). */
static int g_r5_wide = 1;
void orc_set_r5_wide(int on) { g_r5_wide = on ? 1 : 0; }
int orc_get_r5_wide(void) { return g_r5_wide; }

/* pixel-relative homogeneous coordinates p_i - f w_i in double; the pixel centre f = (2 px + 1 - W) / W as an integer times 1 / W */
static inline void r5_wide_setup(const float* p0, const float* p1, const float* p2, int px, int py, int H, int W, double* q /* p0x p0y p1x p1y p2x p2y */,
                                 double* fxo, double* fyo)
{
    const double iW = 1.0 / (double)W, iH = 1.0 / (double)H;
    double fx = (double)(2 * px + 1 - W); fx = fx * iW;
    double fy = (double)(2 * py + 1 - H); fy = fy * iH;
    double t;
    t = fx * (double)p0[3]; q[0] = (double)p0[0] - t;  t = fy * (double)p0[3]; q[1] = (double)p0[1] - t;
    t = fx * (double)p1[3]; q[2] = (double)p1[0] - t;  t = fy * (double)p1[3]; q[3] = (double)p1[1] - t;
    t = fx * (double)p2[3]; q[4] = (double)p2[0] - t;  t = fy * (double)p2[3]; q[5] = (double)p2[1] - t;
    *fxo = fx; *fyo = fy;
}

/* (u, v) of a fragment rule R5 has accepted, clamped to [0, 1], rounded to float once */
static inline void r5_uv_wide(const float* p0, const float* p1, const float* p2, int px, int py, int H, int W, float* u, float* v)
{
    double q[6], fx, fy;
    r5_wide_setup(p0, p1, p2, px, py, H, W, q, &fx, &fy);
    double m0, m1;
    m0 = q[2] * q[5]; m1 = q[3] * q[4]; double a0 = m0 - m1;
    m0 = q[4] * q[1]; m1 = q[5] * q[0]; double a1 = m0 - m1;
    m0 = q[0] * q[3]; m1 = q[1] * q[2]; double a2 = m0 - m1;
    double at = a0 + a1; at = at + a2;
    double iw = 1.0 / at;
    double b0 = a0 * iw;
    double b1 = a1 * iw;
    float uf = (float)b0, vf = (float)b1;
    if (!(uf >= 0.0f)) uf = 0.0f;                                /* (also a NaN from a zero area in double: the fp32 accept test passed) */
    if (uf > 1.0f) uf = 1.0f;
    if (!(vf >= 0.0f)) vf = 0.0f;
    if (vf > 1.0f) vf = 1.0f;
    *u = uf; *v = vf;
}

/* d (u, v) / d clip positions in double; g[3 k + 0 / 1 / 2] = gradient w.r.t. x, y, w of vertex k */
static inline void r5_bwd_wide(const float* p0, const float* p1, const float* p2, int px, int py, int H, int W, double du, double dv, double* g)
{
    double q[6], fx, fy;
    r5_wide_setup(p0, p1, p2, px, py, H, W, q, &fx, &fy);
    const double p0x = q[0], p0y = q[1], p1x = q[2], p1y = q[3], p2x = q[4], p2y = q[5];
    double a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
    double at = a0 + a1 + a2;
    double iw = 1.0 / at;
    double b0 = a0 * iw, b1 = a1 * iw;
    double gb0 = du * iw, gb1 = dv * iw, gbb = gb0 * b0 + gb1 * b1;
    double g0x = gbb * (p2y - p1y) - gb1 * p2y;
    double g1x = gbb * (p0y - p2y) + gb0 * p2y;
    double g2x = gbb * (p1y - p0y) - gb0 * p1y + gb1 * p0y;
    double g0y = gbb * (p1x - p2x) + gb1 * p2x;
    double g1y = gbb * (p2x - p0x) - gb0 * p2x;
    double g2y = gbb * (p0x - p1x) + gb0 * p1x - gb1 * p0x;
    g[0] = g0x; g[1] = g0y; g[2] = -fx * g0x - fy * g0y;
    g[3] = g1x; g[4] = g1y; g[5] = -fx * g1x - fy * g1y;
    g[6] = g2x; g[7] = g2y; g[8] = -fx * g2x - fy * g2y;
}

#define R5_WIDE 1
#define SFX(name) name##_f32
#define REAL float
#define RINT rintf
#define FABS fabsf
#define MAXREAL FLT_MAX
#include "raster_oracle_impl.h"
#undef SFX
#undef REAL
#undef RINT
#undef FABS
#undef MAXREAL

#undef R5_WIDE
#define R5_WIDE 0
#define SFX(name) name##_f64
#define REAL double
#define RINT rint
#define FABS fabs
#define MAXREAL DBL_MAX
#include "raster_oracle_impl.h"
#undef SFX
#undef REAL
#undef RINT
#undef FABS
#undef MAXREAL

/* ---------------------------------------------------------------------------------------------------------------------
 * "mix" variants: every DISCRETE decision (snap, coverage, depth order, antialias pair/edge selection, clamp flag) is taken
 * by the fp32 twin on float-cast inputs — i.e. exactly the decisions the HIP kernels take — and every VALUE is then
 * recomputed in fp64.  This is the reference for value tolerances: "the GPU's decisions, exact arithmetic". */
void orc_rasterize_fwd_mix(const double* pos, const int32_t* tri, int n, int V, int F, int H, int W, double* rast)
{
    size_t npos = (size_t)n * V * 4, npix = (size_t)n * H * W;
    float* posf = (float*)malloc(sizeof(float) * (npos ? npos : 1));
    float* rastf = (float*)malloc(sizeof(float) * (npix ? npix : 1) * 4);
    for (size_t i = 0; i < npos; i++) posf[i] = (float)pos[i];
    orc_rasterize_fwd_f32(posf, tri, n, V, F, H, W, rastf);
    const double xs = 2.0 / W, ys = 2.0 / H, xo = 1.0 / W - 1.0, yo = 1.0 / H - 1.0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < n; b++)
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++) {
                size_t pix = ((size_t)b * H + py) * W + px;
                double* o = rast + pix * 4;
                int t = (int)rastf[pix * 4 + 3] - 1;
                o[0] = o[1] = o[2] = o[3] = 0.0;
                if (t < 0) continue;
                const double* vp = pos + (size_t)b * V * 4;
                double u, v, zw;
                if (!fragment_f64(vp + (size_t)tri[t * 3] * 4, vp + (size_t)tri[t * 3 + 1] * 4, vp + (size_t)tri[t * 3 + 2] * 4,
                                  px, py, H, W, xs, xo, ys, yo, &u, &v, &zw)) { u = rastf[pix * 4]; v = rastf[pix * 4 + 1]; zw = rastf[pix * 4 + 2]; }
                o[0] = u; o[1] = v; o[2] = zw; o[3] = (double)(t + 1);
            }
    free(posf); free(rastf);
}

int orc_antialias_fwd_mix(const double* color, const double* rast, const double* pos, const int32_t* tri, const int32_t* opp,
                          int n, int V, int F, int H, int W, int C, double* out, int32_t* items, double* alphas, int max_items)
{
    size_t npos = (size_t)n * V * 4, npix = (size_t)n * H * W;
    float* posf = (float*)malloc(sizeof(float) * (npos ? npos : 1));
    float* rastf = (float*)malloc(sizeof(float) * (npix ? npix : 1) * 4);
    for (size_t i = 0; i < npos; i++) posf[i] = (float)pos[i];
    for (size_t i = 0; i < npix * 4; i++) rastf[i] = (float)rast[i];
    memcpy(out, color, sizeof(double) * npix * C);
    const double xh = 0.5 * W, yh = 0.5 * H;
    /* per-view item lists, concatenated in view order: see orc_antialias_fwd in raster_oracle_impl.h */
    int32_t** vit = (int32_t**)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t*));
    double** val = (double**)calloc((size_t)(n > 0 ? n : 1), sizeof(double*));
    int* vcnt = (int*)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < n; b++) {
        int cap = 1024, cnt = 0;
        int32_t* it = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)cap);
        double* al = (double*)malloc(sizeof(double) * (size_t)cap);
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++)
                for (int d = 0; d < 2; d++) {
                    if (d == 0 && px + 1 >= W) continue;
                    if (d == 1 && py + 1 >= H) continue;
                    size_t pix0 = ((size_t)b * H + py) * W + px;
                    size_t pix1 = pix0 + (d ? (size_t)W : 1);
                    if (rastf[pix0 * 4 + 3] == rastf[pix1 * 4 + 3]) continue;
                    float alpha_f;
                    int code = aa_analyse_f32(rastf, posf, tri, opp, V, F, H, W, b, px, py, d, &alpha_f);
                    if (!code) continue;
                    double alpha = (double)alpha_f;
                    if (!(code & AA_CLAMPED)) {
                        /* recompute the crossing of the selected edge in fp64 */
                        int t = (int)rastf[((code & AA_TRI1) ? pix1 : pix0) * 4 + 3] - 1;
                        int qx = px, qy = py;
                        if (code & AA_TRI1) { qx += 1 - d; qy += d; }
                        int di = AA_DI(code);
                        int va = tri[t * 3 + (di + 1) % 3], vb = tri[t * 3 + (di + 2) % 3];
                        const double* pa = pos + ((size_t)b * V + va) * 4;
                        const double* pb = pos + ((size_t)b * V + vb) * 4;
                        double fx = qx + 0.5 - xh, fy = qy + 0.5 - yh;
                        double xa = pa[0] / pa[3] * xh - fx, ya = pa[1] / pa[3] * yh - fy;
                        double xb = pb[0] / pb[3] * xh - fx, yb = pb[1] / pb[3] * yh - fy;
                        if (d) { double tt; tt = xa; xa = ya; ya = tt; tt = xb; xb = yb; yb = tt; }
                        double ds = (code & AA_TRI1) ? -1.0 : 1.0;
                        double xc = xa + ya / (ya - yb) * (xb - xa);
                        double dc = ds * xc;
                        if (dc < 0.0) dc = 0.0;
                        if (dc > 1.0) dc = 1.0;
                        alpha = ds * (0.5 - dc);
                        if ((alpha > 0.0) != (alpha_f > 0.0f)) alpha = (double)alpha_f;   /* keep the f32 target pixel */
                    }
                    const double* c0 = color + pix0 * C;
                    const double* c1 = color + pix1 * C;
                    double* o = out + ((alpha > 0.0) ? pix0 : pix1) * C;
                    for (int c = 0; c < C; c++) o[c] += alpha * (c1[c] - c0[c]);
                    if (cnt == cap) {
                        cap *= 2;
                        it = (int32_t*)realloc(it, sizeof(int32_t) * 4 * (size_t)cap);
                        al = (double*)realloc(al, sizeof(double) * (size_t)cap);
                    }
                    it[cnt * 4 + 0] = px; it[cnt * 4 + 1] = py; it[cnt * 4 + 2] = b; it[cnt * 4 + 3] = code;
                    al[cnt] = alpha;
                    cnt++;
                }
        vit[b] = it; val[b] = al; vcnt[b] = cnt;
    }
    int cnt = 0;
    for (int b = 0; b < n; b++) {
        for (int k = 0; k < vcnt[b]; k++, cnt++)
            if (cnt < max_items) {
                memcpy(items + (size_t)cnt * 4, vit[b] + (size_t)k * 4, sizeof(int32_t) * 4);
                alphas[cnt] = val[b][k];
            }
        free(vit[b]); free(val[b]);
    }
    free(vit); free(val); free(vcnt);
    free(posf); free(rastf);
    return cnt;
}

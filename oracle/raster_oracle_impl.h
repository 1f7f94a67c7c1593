/*
 * oracle/raster_oracle_impl.h — TEST INFRASTRUCTURE (see oracle/README.md), never linked into the product.
 *
 * CPU restatement of the three differentiable-rasterization primitives that the reference calls through
 * nvdiffrast (third-party, un-vendored; call sites: /root/reference/mesh_sfs_optim.py:267,269,274,287,
 * get_data.py:251-253).  PARITY UNPINNED against real nvdiffrast: the rules below are the
 * build's written spec, pinned by invariants and fp64 central differences.
 *
 * This header is included twice by raster_oracle.c: REAL=float (decision-exact twin of the HIP kernels — every
 * operation that feeds a discrete decision is written as one IEEE operation per statement and the file is compiled
 * with -ffp-contract=off) and REAL=double (finite-difference target).
 * Round 5: in the float instantiation (R5_WIDE) the VALUES of rule R5 — (u, v) of a winner and the gradient of (u, v) with respect to
 * the clip positions — are evaluated in double from the float positions (raster_oracle.c: r5_uv_wide / r5_bwd_wide), exactly as the
 * HIP kernels do; every decision (coverage, accept / reject, z/w, depth order) stays float.
 */

#define R_(x) ((REAL)(x))

/* ---------------------------------------------------------------- rule R1+R2: reject + snap to 1/256 px */
static inline int SFX(snap_vertex)(const REAL* p, int H, int W, int64_t* ox, int64_t* oy)
{
    REAL x = p[0], y = p[1], z = p[2], w = p[3];
    if (!(w > R_(0)) || !(z >= -w) || !(z <= w)) return 0;       /* whole-triangle near/far reject */
    REAL rw = R_(1) / w;
    REAL sx = x * rw;
    REAL sy = y * rw;
    REAL kx = (REAL)W * R_(128);
    REAL ky = (REAL)H * R_(128);
    REAL qx = sx * kx;
    qx = qx + kx;
    REAL qy = sy * ky;
    qy = qy + ky;
    if (!(qx == qx) || !(qy == qy)) return 0;
    const REAL LIM = R_(4194304);                               /* 2^22 sub-pixel units guard band */
    if (qx > LIM) qx = LIM;
    if (qx < -LIM) qx = -LIM;
    if (qy > LIM) qy = LIM;
    if (qy < -LIM) qy = -LIM;
    *ox = (int64_t)RINT(qx);
    *oy = (int64_t)RINT(qy);
    return 1;
}

/* floor division for possibly negative numerators */
static inline int64_t SFX(floordiv)(int64_t a, int64_t b) { int64_t q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }

/* rule R4: inside test for CCW-normalised integer triangle at sub-pixel point (cx,cy) */
static inline int SFX(edge_in)(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t cx, int64_t cy)
{
    int64_t dx = bx - ax, dy = by - ay;
    int64_t e = dx * (cy - ay) - dy * (cx - ax);
    if (e > 0) return 1;
    if (e < 0) return 0;
    return (dy > 0) || (dy == 0 && dx > 0);
}

/* rule R5: the fragment (u, v, z/w) of triangle (p0,p1,p2) at pixel (px,py); returns 0 if degenerate */
static inline int SFX(fragment)(const REAL* p0, const REAL* p1, const REAL* p2, int px, int py, int H, int W,
                                REAL xs, REAL xo, REAL ys, REAL yo, REAL* u, REAL* v, REAL* zw)
{
    (void)H; (void)W;
    REAL fx = xs * (REAL)px; fx = fx + xo;
    REAL fy = ys * (REAL)py; fy = fy + yo;
    REAL t;
    t = fx * p0[3]; REAL p0x = p0[0] - t;  t = fy * p0[3]; REAL p0y = p0[1] - t;
    t = fx * p1[3]; REAL p1x = p1[0] - t;  t = fy * p1[3]; REAL p1y = p1[1] - t;
    t = fx * p2[3]; REAL p2x = p2[0] - t;  t = fy * p2[3]; REAL p2y = p2[1] - t;
    REAL m0, m1;
    m0 = p1x * p2y; m1 = p1y * p2x; REAL a0 = m0 - m1;
    m0 = p2x * p0y; m1 = p2y * p0x; REAL a1 = m0 - m1;
    m0 = p0x * p1y; m1 = p0y * p1x; REAL a2 = m0 - m1;
    REAL at = a0 + a1; at = at + a2;
    if (!(FABS(at) > R_(0))) return 0;
    REAL iw = R_(1) / at;
    REAL b0 = a0 * iw;
    REAL b1 = a1 * iw;
    REAL zn = p0[2] * a0; t = p1[2] * a1; zn = zn + t; t = p2[2] * a2; zn = zn + t;
    REAL wn = p0[3] * a0; t = p1[3] * a1; wn = wn + t; t = p2[3] * a2; wn = wn + t;
    REAL q = zn / wn;
    if (!(q == q)) return 0;
    if (b0 < R_(0)) b0 = R_(0);
    if (b0 > R_(1)) b0 = R_(1);
    if (b1 < R_(0)) b1 = R_(0);
    if (b1 > R_(1)) b1 = R_(1);
    if (q < R_(-1)) q = R_(-1);
    if (q > R_(1)) q = R_(1);
    q = q + R_(0);                                             /* -0 -> +0 so the depth order is total */
    *u = b0; *v = b1; *zw = q;
#if R5_WIDE
    /* the fp32 twin: accept / reject and z/w above are the decisions (bit-identical to the HIP kernels); the barycentrics it returns
     * are evaluated in double from the same fp32 positions and rounded once (raster_oracle.c: r5_uv_wide) */
    if (g_r5_wide) r5_uv_wide(p0, p1, p2, px, py, H, W, u, v);
#endif
    return 1;
}

/* rasterize forward.  pos [n,V,4], tri [F,3], rast [n,H,W,4] = (u, v, z/w, tri_id+1) or zeros. */
void SFX(orc_rasterize_fwd)(const REAL* pos, const int32_t* tri, int n, int V, int F, int H, int W, REAL* rast)
{
    const REAL xs = R_(2) / (REAL)W, ys = R_(2) / (REAL)H;
    REAL xo = R_(1) / (REAL)W; xo = xo - R_(1);
    REAL yo = R_(1) / (REAL)H; yo = yo - R_(1);
    memset(rast, 0, sizeof(REAL) * (size_t)n * H * W * 4);
    /* views are independent (each writes its own image): one OpenMP thread per view at a time — same statements per view, same
     * results whatever the thread count (bench.py's cpu_baseline runs this on the box's cores; round 3 timed a single-threaded loop) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < n; b++) {
        const REAL* vp = pos + (size_t)b * V * 4;
        REAL* ro = rast + (size_t)b * H * W * 4;
        for (int f = 0; f < F; f++) {
            int i0 = tri[f * 3 + 0], i1 = tri[f * 3 + 1], i2 = tri[f * 3 + 2];
            if (i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V) continue;
            int64_t x0, y0, x1, y1, x2, y2;
            if (!SFX(snap_vertex)(vp + i0 * 4, H, W, &x0, &y0)) continue;
            if (!SFX(snap_vertex)(vp + i1 * 4, H, W, &x1, &y1)) continue;
            if (!SFX(snap_vertex)(vp + i2 * 4, H, W, &x2, &y2)) continue;
            int64_t area2 = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
            if (area2 == 0) continue;
            if (area2 < 0) { int64_t t; t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
            int64_t mnx = x0 < x1 ? x0 : x1; if (x2 < mnx) mnx = x2;
            int64_t mxx = x0 > x1 ? x0 : x1; if (x2 > mxx) mxx = x2;
            int64_t mny = y0 < y1 ? y0 : y1; if (y2 < mny) mny = y2;
            int64_t mxy = y0 > y1 ? y0 : y1; if (y2 > mxy) mxy = y2;
            int64_t pxa = -SFX(floordiv)(-(mnx - 128), 256);   /* ceil((mnx-128)/256) */
            int64_t pxb = SFX(floordiv)(mxx - 128, 256);
            int64_t pya = -SFX(floordiv)(-(mny - 128), 256);
            int64_t pyb = SFX(floordiv)(mxy - 128, 256);
            if (pxa < 0) pxa = 0;
            if (pya < 0) pya = 0;
            if (pxb > W - 1) pxb = W - 1;
            if (pyb > H - 1) pyb = H - 1;
            for (int64_t py = pya; py <= pyb; py++)
                for (int64_t px = pxa; px <= pxb; px++) {
                    int64_t cx = px * 256 + 128, cy = py * 256 + 128;
                    if (!SFX(edge_in)(x0, y0, x1, y1, cx, cy)) continue;
                    if (!SFX(edge_in)(x1, y1, x2, y2, cx, cy)) continue;
                    if (!SFX(edge_in)(x2, y2, x0, y0, cx, cy)) continue;
                    REAL u, v, zw;
                    if (!SFX(fragment)(vp + i0 * 4, vp + i1 * 4, vp + i2 * 4, (int)px, (int)py, H, W, xs, xo, ys, yo, &u, &v, &zw)) continue;
                    REAL* o = ro + ((size_t)py * W + px) * 4;
                    /* nearest z/w wins; ties keep the lower triangle index (we iterate f ascending) */
                    if (o[3] == R_(0) || zw < o[2]) { o[0] = u; o[1] = v; o[2] = zw; o[3] = (REAL)(f + 1); }
                }
        }
    }
}

/* rasterize backward.  dy [n,H,W,4] (only .x,.y used); gpos [n,V,4] accumulated. */
void SFX(orc_rasterize_bwd)(const REAL* pos, const int32_t* tri, const REAL* rast, const REAL* dy,
                            int n, int V, int F, int H, int W, REAL* gpos)
{
    const REAL xs = R_(2) / (REAL)W, ys = R_(2) / (REAL)H;
    REAL xo = R_(1) / (REAL)W; xo = xo - R_(1);
    REAL yo = R_(1) / (REAL)H; yo = yo - R_(1);
#pragma omp parallel for schedule(dynamic, 1)                  /* gpos rows of view b belong to view b: no two threads meet */
    for (int b = 0; b < n; b++)
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++) {
                size_t pix = ((size_t)b * H + py) * W + px;
                int t = (int)rast[pix * 4 + 3] - 1;
                if (t < 0 || t >= F) continue;
                REAL du = dy[pix * 4 + 0], dv = dy[pix * 4 + 1];
                if (du == R_(0) && dv == R_(0)) continue;
                int i0 = tri[t * 3], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
                const REAL* p0 = pos + ((size_t)b * V + i0) * 4;
                const REAL* p1 = pos + ((size_t)b * V + i1) * 4;
                const REAL* p2 = pos + ((size_t)b * V + i2) * 4;
                REAL* q0 = gpos + ((size_t)b * V + i0) * 4;
                REAL* q1 = gpos + ((size_t)b * V + i1) * 4;
                REAL* q2 = gpos + ((size_t)b * V + i2) * 4;
#if R5_WIDE
                if (g_r5_wide) {                                 /* the fp32 twin: this gradient in double, rounded at the accumulation */
                    double g[9];
                    r5_bwd_wide(p0, p1, p2, px, py, H, W, (double)du, (double)dv, g);
                    q0[0] += (REAL)g[0]; q0[1] += (REAL)g[1]; q0[3] += (REAL)g[2];
                    q1[0] += (REAL)g[3]; q1[1] += (REAL)g[4]; q1[3] += (REAL)g[5];
                    q2[0] += (REAL)g[6]; q2[1] += (REAL)g[7]; q2[3] += (REAL)g[8];
                    continue;
                }
#endif
                REAL fx = xs * (REAL)px + xo, fy = ys * (REAL)py + yo;
                REAL p0x = p0[0] - fx * p0[3], p0y = p0[1] - fy * p0[3];
                REAL p1x = p1[0] - fx * p1[3], p1y = p1[1] - fy * p1[3];
                REAL p2x = p2[0] - fx * p2[3], p2y = p2[1] - fy * p2[3];
                REAL a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
                REAL at = a0 + a1 + a2;
                REAL iw = R_(1) / at;
                REAL b0 = a0 * iw, b1 = a1 * iw;
                REAL gb0 = du * iw, gb1 = dv * iw, gbb = gb0 * b0 + gb1 * b1;
                REAL g0x = gbb * (p2y - p1y) - gb1 * p2y;
                REAL g1x = gbb * (p0y - p2y) + gb0 * p2y;
                REAL g2x = gbb * (p1y - p0y) - gb0 * p1y + gb1 * p0y;
                REAL g0y = gbb * (p1x - p2x) + gb1 * p2x;
                REAL g1y = gbb * (p2x - p0x) - gb0 * p2x;
                REAL g2y = gbb * (p0x - p1x) + gb0 * p1x - gb1 * p0x;
                q0[0] += g0x; q0[1] += g0y; q0[3] += -fx * g0x - fy * g0y;
                q1[0] += g1x; q1[1] += g1y; q1[3] += -fx * g1x - fy * g1y;
                q2[0] += g2x; q2[1] += g2y; q2[3] += -fx * g2x - fy * g2y;
            }
}

/* interpolate forward (Appendix A.2).  attr [an,V,C] with an in {1,n}; out [n,H,W,C]. */
void SFX(orc_interpolate_fwd)(const REAL* attr, const REAL* rast, const int32_t* tri,
                              int n, int an, int V, int F, int H, int W, int C, REAL* out)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < n; b++) {
        const REAL* A = attr + (an == 1 ? 0 : (size_t)b * V * C);
        for (size_t p = 0; p < (size_t)H * W; p++) {
            size_t pix = (size_t)b * H * W + p;
            REAL* o = out + pix * C;
            int t = (int)rast[pix * 4 + 3] - 1;
            if (t < 0 || t >= F) { for (int c = 0; c < C; c++) o[c] = R_(0); continue; }
            REAL u = rast[pix * 4], v = rast[pix * 4 + 1], w2 = R_(1) - u - v;
            const REAL* a0 = A + (size_t)tri[t * 3] * C;
            const REAL* a1 = A + (size_t)tri[t * 3 + 1] * C;
            const REAL* a2 = A + (size_t)tri[t * 3 + 2] * C;
            for (int c = 0; c < C; c++) o[c] = u * a0[c] + v * a1[c] + w2 * a2[c];
        }
    }
}

/* interpolate backward (Appendix B.2).  gattr [an,V,C] accumulated; grast [n,H,W,4] written (.z,.w zero). */
void SFX(orc_interpolate_bwd)(const REAL* attr, const REAL* rast, const int32_t* tri, const REAL* dy,
                              int n, int an, int V, int F, int H, int W, int C, REAL* gattr, REAL* grast)
{
    /* a broadcast attribute (an == 1) collects the gradient of every view in ONE array: serial then (fixed summation order) */
#pragma omp parallel for schedule(dynamic, 1) if (an != 1)
    for (int b = 0; b < n; b++) {
        const REAL* A = attr + (an == 1 ? 0 : (size_t)b * V * C);
        REAL* G = gattr + (an == 1 ? 0 : (size_t)b * V * C);
        for (size_t p = 0; p < (size_t)H * W; p++) {
            size_t pix = (size_t)b * H * W + p;
            REAL* gr = grast + pix * 4;
            gr[0] = gr[1] = gr[2] = gr[3] = R_(0);
            int t = (int)rast[pix * 4 + 3] - 1;
            if (t < 0 || t >= F) continue;
            REAL u = rast[pix * 4], v = rast[pix * 4 + 1], w2 = R_(1) - u - v;
            size_t i0 = tri[t * 3], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
            const REAL* d = dy + pix * C;
            REAL gu = R_(0), gv = R_(0);
            for (int c = 0; c < C; c++) {
                G[i0 * C + c] += u * d[c];
                G[i1 * C + c] += v * d[c];
                G[i2 * C + c] += w2 * d[c];
                gu += d[c] * (A[i0 * C + c] - A[i2 * C + c]);
                gv += d[c] * (A[i1 * C + c] - A[i2 * C + c]);
            }
            gr[0] = gu; gr[1] = gv;
        }
    }
}

/* ---------------------------------------------------------------- antialias (Appendix A.3, nvdiffrast-style) */
static inline int SFX(same_sign)(REAL a, REAL b) { return (signbit(a) != 0) == (signbit(b) != 0); }
static inline int SFX(rational_gt)(REAL n0, REAL n1, REAL d0, REAL d1)
{   /* n0/d0 > n1/d1 without dividing */
    REAL dd = d0 * d1;
    REAL l = n0 * d1;
    REAL r = n1 * d0;
    if (dd < R_(0)) return l < r;
    return l > r;
}
static inline int SFX(max_idx3)(REAL n0, REAL n1, REAL n2, REAL d0, REAL d1, REAL d2)
{
    int g10 = SFX(rational_gt)(n1, n0, d1, d0);
    int g20 = SFX(rational_gt)(n2, n0, d2, d0);
    int g21 = SFX(rational_gt)(n2, n1, d2, d1);
    if (g20 && g21) return 2;
    if (g10) return 1;
    return 0;
}

/* work item code bits: [1:0] edge index di, [2] d (1 = vertical pair), [3] tri1 chosen, [4] edge found, [5] dc clamped */
#define AA_DI(c) ((c) & 3)
#define AA_DOWN 4
#define AA_TRI1 8
#define AA_FOUND 16
#define AA_CLAMPED 32

/* rule A3: which edges of triangle (vi0,vi1,vi2) are silhouette edges in this view — decided on the SNAPPED integer
 * coordinates (exact, translation invariant, consistent with coverage): edge k (opposite vertex k) is a silhouette edge iff the
 * neighbouring triangle's opposite vertex lies on the same side of the edge as vertex k (fold) — or there is no neighbour.
 * bit k set = edge k is silhouette; returns 0 if a vertex of the triangle itself is rejected. */
static int SFX(silhouette_bits)(const REAL* P, int vi0, int vi1, int vi2, int op0, int op1, int op2, int H, int W)
{
    int64_t x0, y0, x1, y1, x2, y2, ox, oy;
    if (!SFX(snap_vertex)(P + (size_t)vi0 * 4, H, W, &x0, &y0)) return 0;
    if (!SFX(snap_vertex)(P + (size_t)vi1 * 4, H, W, &x1, &y1)) return 0;
    if (!SFX(snap_vertex)(P + (size_t)vi2 * 4, H, W, &x2, &y2)) return 0;
    int64_t bb = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    int bits = 0;
    int64_t a;
    if (op0 < 0 || !SFX(snap_vertex)(P + (size_t)op0 * 4, H, W, &ox, &oy)) { ox = x0; oy = y0; }
    a = (x1 - ox) * (y2 - oy) - (x2 - ox) * (y1 - oy);
    if ((a < 0) == (bb < 0)) bits |= 1;
    if (op1 < 0 || !SFX(snap_vertex)(P + (size_t)op1 * 4, H, W, &ox, &oy)) { ox = x1; oy = y1; }
    a = (x2 - ox) * (y0 - oy) - (x0 - ox) * (y2 - oy);
    if ((a < 0) == (bb < 0)) bits |= 2;
    if (op2 < 0 || !SFX(snap_vertex)(P + (size_t)op2 * 4, H, W, &ox, &oy)) { ox = x2; oy = y2; }
    a = (x0 - ox) * (y1 - oy) - (x1 - ox) * (y0 - oy);
    if ((a < 0) == (bb < 0)) bits |= 4;
    return bits;
}

/* Analyse one pixel pair.  Returns code (0 if the pair produces no blend) and alpha. */
static int SFX(aa_analyse)(const REAL* rast, const REAL* pos, const int32_t* tri, const int32_t* opp,
                           int V, int F, int H, int W, int b, int px, int py, int d, REAL* alpha_out)
{
    size_t pix0 = ((size_t)b * H + py) * W + px;
    size_t pix1 = pix0 + (d ? (size_t)W : 1);
    REAL z0 = rast[pix0 * 4 + 2], z1 = rast[pix1 * 4 + 2];
    int tri0 = (int)rast[pix0 * 4 + 3] - 1, tri1 = (int)rast[pix1 * 4 + 3] - 1;
    int t = (tri0 >= 0) ? tri0 : tri1;
    if (tri0 >= 0 && tri1 >= 0) t = (z0 < z1) ? tri0 : tri1;
    int qx = px, qy = py;
    if (t == tri1) { qx += 1 - d; qy += d; }
    if (t < 0 || t >= F) return 0;
    int vi0 = tri[t * 3], vi1 = tri[t * 3 + 1], vi2 = tri[t * 3 + 2];
    if (vi0 < 0 || vi0 >= V || vi1 < 0 || vi1 >= V || vi2 < 0 || vi2 >= V) return 0;
    int op0 = opp[t * 3], op1 = opp[t * 3 + 1], op2 = opp[t * 3 + 2];
    const REAL* P = pos + (size_t)b * V * 4;
    int sb = SFX(silhouette_bits)(P, vi0, vi1, vi2, op0, op1, op2, H, W);
    if (!sb) return 0;
    int s0 = sb & 1, s1 = sb & 2, s2 = sb & 4;
    const REAL* p0 = P + (size_t)vi0 * 4; const REAL* p1 = P + (size_t)vi1 * 4; const REAL* p2 = P + (size_t)vi2 * 4;
    REAL xh = R_(0.5) * (REAL)W, yh = R_(0.5) * (REAL)H;
    REAL w0 = R_(1) / p0[3], w1 = R_(1) / p1[3], w2 = R_(1) / p2[3];
    REAL fx = (REAL)qx + R_(0.5); fx = fx - xh;
    REAL fy = (REAL)qy + R_(0.5); fy = fy - yh;
    REAL m;
    m = p0[0] * w0; m = m * xh; REAL x0 = m - fx;   m = p0[1] * w0; m = m * yh; REAL y0 = m - fy;
    m = p1[0] * w1; m = m * xh; REAL x1 = m - fx;   m = p1[1] * w1; m = m * yh; REAL y1 = m - fy;
    m = p2[0] * w2; m = m * xh; REAL x2 = m - fx;   m = p2[1] * w2; m = m * yh; REAL y2 = m - fy;
    REAL l, r;
    if (d) { REAL tt; tt = x0; x0 = y0; y0 = tt; tt = x1; x1 = y1; y1 = tt; tt = x2; x2 = y2; y2 = tt; }
    REAL dx0 = x2 - x1, dx1 = x0 - x2, dx2 = x1 - x0;
    REAL dy0 = y2 - y1, dy1 = y0 - y2, dy2 = y1 - y0;
    REAL ds = (t == tri0) ? R_(1) : R_(-1);
    l = x1 * dy0; r = y1 * dx0; REAL d0 = l - r; d0 = ds * d0;
    l = x2 * dy1; r = y2 * dx1; REAL d1 = l - r; d1 = ds * d1;
    l = x0 * dy2; r = y0 * dx2; REAL d2 = l - r; d2 = ds * d2;
    const REAL NEG = -MAXREAL;
    if (SFX(same_sign)(y1, y2)) { d0 = NEG; dy0 = R_(1); }
    if (SFX(same_sign)(y2, y0)) { d1 = NEG; dy1 = R_(1); }
    if (SFX(same_sign)(y0, y1)) { d2 = NEG; dy2 = R_(1); }
    int di = SFX(max_idx3)(d0, d1, d2, dy0, dy1, dy2);
    REAL dc = NEG;
    if (di == 0 && s0 && FABS(dy0) >= FABS(dx0)) dc = d0 / dy0;
    if (di == 1 && s1 && FABS(dy1) >= FABS(dx1)) dc = d1 / dy1;
    if (di == 2 && s2 && FABS(dy2) >= FABS(dx2)) dc = d2 / dy2;
    const REAL eps = R_(0.0625);
    if (!(dc > -eps && dc < R_(1) + eps)) return 0;
    int code = di | (d ? AA_DOWN : 0) | ((t == tri0) ? 0 : AA_TRI1) | AA_FOUND;
    if (dc < R_(0)) { dc = R_(0); code |= AA_CLAMPED; }
    if (dc > R_(1)) { dc = R_(1); code |= AA_CLAMPED; }
    REAL al = R_(0.5) - dc;
    al = ds * al;
    *alpha_out = al;
    return code;
}

/* antialias forward.  Returns the number of work items (pairs with a blend); items [max,4] = (px,py,view,code). */
int SFX(orc_antialias_fwd)(const REAL* color, const REAL* rast, const REAL* pos, const int32_t* tri, const int32_t* opp,
                           int n, int V, int F, int H, int W, int C, REAL* out, int32_t* items, REAL* alphas, int max_items)
{
    memcpy(out, color, sizeof(REAL) * (size_t)n * H * W * C);
    /* one view per OpenMP thread at a time; every view collects its work items in a list of its own (scan order), and the lists are
     * concatenated in view order afterwards: the queue is the serial loop's, item for item, whatever the thread count */
    int32_t** vit = (int32_t**)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t*));
    REAL** val = (REAL**)calloc((size_t)(n > 0 ? n : 1), sizeof(REAL*));
    int* vcnt = (int*)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < n; b++) {
        int cap = 1024, cnt = 0;
        int32_t* it = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)cap);
        REAL* al = (REAL*)malloc(sizeof(REAL) * (size_t)cap);
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++)
                for (int d = 0; d < 2; d++) {
                    if (d == 0 && px + 1 >= W) continue;
                    if (d == 1 && py + 1 >= H) continue;
                    size_t pix0 = ((size_t)b * H + py) * W + px;
                    size_t pix1 = pix0 + (d ? (size_t)W : 1);
                    if (rast[pix0 * 4 + 3] == rast[pix1 * 4 + 3]) continue;
                    REAL alpha;
                    int code = SFX(aa_analyse)(rast, pos, tri, opp, V, F, H, W, b, px, py, d, &alpha);
                    if (!code) continue;
                    const REAL* c0 = color + pix0 * C;
                    const REAL* c1 = color + pix1 * C;
                    REAL* o = out + ((alpha > R_(0)) ? pix0 : pix1) * C;
                    for (int c = 0; c < C; c++) o[c] += alpha * (c1[c] - c0[c]);
                    if (cnt == cap) {
                        cap *= 2;
                        it = (int32_t*)realloc(it, sizeof(int32_t) * 4 * (size_t)cap);
                        al = (REAL*)realloc(al, sizeof(REAL) * (size_t)cap);
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
    return cnt;
}

/* antialias backward: gcolor [n,H,W,C] written (= dy + blend terms); gpos [n,V,4] accumulated. */
void SFX(orc_antialias_bwd)(const REAL* color, const REAL* rast, const REAL* pos, const int32_t* tri, const REAL* dy,
                            const int32_t* items, const REAL* alphas, int n_items,
                            int n, int V, int F, int H, int W, int C, REAL* gcolor, REAL* gpos)
{
    (void)F;
    memcpy(gcolor, dy, sizeof(REAL) * (size_t)n * H * W * C);
    REAL xh = R_(0.5) * (REAL)W, yh = R_(0.5) * (REAL)H;
    /* the forward emits the items view by view: runs of equal view index are independent (a view's pixels and gpos rows are its own),
     * one run per OpenMP thread at a time, each in list order; an unsorted list (hand-made by a test) is processed serially */
    int sorted = 1, nruns = 0;
    for (int k = 1; k < n_items; k++)
        if (items[k * 4 + 2] < items[(k - 1) * 4 + 2]) sorted = 0;
    int* run0 = (int*)malloc(sizeof(int) * (size_t)(n_items + 2));
    for (int k = 0; k < n_items; k++)
        if (k == 0 || (sorted && items[k * 4 + 2] != items[(k - 1) * 4 + 2])) run0[nruns++] = k;
    run0[nruns] = n_items;
#pragma omp parallel for schedule(dynamic, 1) if (sorted && nruns > 1)
    for (int r = 0; r < nruns; r++)
    for (int k = run0[r]; k < run0[r + 1]; k++) {
        int px = items[k * 4], py = items[k * 4 + 1], b = items[k * 4 + 2], code = items[k * 4 + 3];
        int d = (code & AA_DOWN) ? 1 : 0, di = AA_DI(code);
        REAL alpha = alphas[k];
        size_t pix0 = ((size_t)b * H + py) * W + px;
        size_t pix1 = pix0 + (d ? (size_t)W : 1);
        const REAL* g = dy + ((alpha > R_(0)) ? pix0 : pix1) * C;
        const REAL* c0 = color + pix0 * C;
        const REAL* c1 = color + pix1 * C;
        REAL galpha = R_(0);
        for (int c = 0; c < C; c++) {
            galpha += g[c] * (c1[c] - c0[c]);
            gcolor[pix1 * C + c] += alpha * g[c];
            gcolor[pix0 * C + c] -= alpha * g[c];
        }
        if (code & AA_CLAMPED) continue;                        /* d alpha / d pos = 0 on the clamp */
        int t = (int)rast[((code & AA_TRI1) ? pix1 : pix0) * 4 + 3] - 1;
        int qx = px, qy = py;
        if (code & AA_TRI1) { qx += 1 - d; qy += d; }
        int va = tri[t * 3 + (di + 1) % 3], vb = tri[t * 3 + (di + 2) % 3];
        const REAL* pa = pos + ((size_t)b * V + va) * 4;
        const REAL* pb = pos + ((size_t)b * V + vb) * 4;
        REAL fx = (REAL)qx + R_(0.5) - xh, fy = (REAL)qy + R_(0.5) - yh;
        REAL wa = R_(1) / pa[3], wb = R_(1) / pb[3];
        REAL xa = pa[0] * wa * xh - fx, ya = pa[1] * wa * yh - fy;
        REAL xb = pb[0] * wb * xh - fx, yb = pb[1] * wb * yh - fy;
        if (d) { REAL tt; tt = xa; xa = ya; ya = tt; tt = xb; xb = yb; yb = tt; }
        /* alpha = ds/2 - xc, xc = xa + tpar (xb - xa), tpar = ya / (ya - yb) */
        REAL gxc = -galpha;
        REAL den = ya - yb, iden = R_(1) / den;
        REAL tpar = ya * iden;
        REAL g_xa = gxc * (R_(1) - tpar), g_xb = gxc * tpar;
        REAL g_t = gxc * (xb - xa);
        REAL g_ya = g_t * (-yb * iden * iden), g_yb = g_t * (ya * iden * iden);
        if (d) { REAL tt; tt = g_xa; g_xa = g_ya; g_ya = tt; tt = g_xb; g_xb = g_yb; g_yb = tt; }
        /* screen x = X/W*xh - fx, screen y = Y/W*yh - fy */
        REAL* qa = gpos + ((size_t)b * V + va) * 4;
        REAL* qb = gpos + ((size_t)b * V + vb) * 4;
        qa[0] += g_xa * xh * wa; qa[1] += g_ya * yh * wa;
        qa[3] += -(g_xa * pa[0] * xh + g_ya * pa[1] * yh) * wa * wa;
        qb[0] += g_xb * xh * wb; qb[1] += g_yb * yh * wb;
        qb[3] += -(g_xb * pb[0] * xh + g_yb * pb[1] * yh) * wb * wb;
    }
    free(run0);
}

#undef R_

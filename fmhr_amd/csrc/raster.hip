// fmhr_amd/csrc/raster.hip — rasterize / interpolate, forward + backward, for gfx950.
// Replaces dr.rasterize / dr.interpolate of nvdiffrast as called at /root/reference/mesh_sfs_optim.py:267,269.
//
// Forward = triangle-parallel scan conversion into a 64-bit (depth | triangle) key buffer with atomicMin, then a
// per-pixel resolve that recomputes the winner's fragment.  The HAM meshes are micro-triangle meshes (~1.3 px per visible
// triangle at 512^2): the work is per triangle, not per pixel, so one lane owns one triangle and walks its
// (tiny) bounding box; a triangle with a large box is swept by the whole wave of its lane.
#include "common.h"
#include "shade_common.h"
#include <stdlib.h>
#include <string.h>

namespace fmhr {

// loads through this pointer type go through the scalar cache when the address is wave-uniform (data constant during the kernel)
typedef const float __attribute__((address_space(4))) const_as4_float;

constexpr bool kBinnedDefault = true;  // the fused step rasterizes through per-tile bins where it can (raster_binned_ok)
constexpr int kSmallBox = 32;   // boxes up to this many pixel centres are walked by the owning lane (32-bit coverage mask)

struct TriSetup {
    int x0, y0, x1, y1, x2, y2;     // CCW-normalised snapped coordinates
    int pxa, pxb, pya, pyb;         // inclusive pixel range
};

constexpr int kRejected = (int)0x80000000;   // snap.x of a vertex that fails R1 (or is NaN)

// one lane per (view, vertex): R1 + R2 once per vertex instead of once per incident triangle corner
__global__ void __launch_bounds__(kBlock)
k_snap_vertices(const float4* __restrict__ pos, long long N, int H, int W, int2* __restrict__ snap)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= N) return;
    int x, y;
    bool ok = snap_vertex(pos[gid], (float)W * 128.0f, (float)H * 128.0f, x, y);
    snap[gid] = ok ? make_int2(x, y) : make_int2(kRejected, 0);
}

__device__ __forceinline__ bool setup_triangle(const int2 s0, const int2 s1, const int2 s2, int H, int W, TriSetup& s)
{
    if (s0.x == kRejected || s1.x == kRejected || s2.x == kRejected) return false;
    s.x0 = s0.x; s.y0 = s0.y; s.x1 = s1.x; s.y1 = s1.y; s.x2 = s2.x; s.y2 = s2.y;
    int mnx = min(s.x0, min(s.x1, s.x2)), mxx = max(s.x0, max(s.x1, s.x2));
    int mny = min(s.y0, min(s.y1, s.y2)), mxy = max(s.y0, max(s.y1, s.y2));
    s.pxa = max((mnx - kHalf + kSub - 1) >> 8, 0);       // ceil((mn-128)/256), arithmetic shift floors
    s.pxb = min((mxx - kHalf) >> 8, W - 1);
    s.pya = max((mny - kHalf + kSub - 1) >> 8, 0);
    s.pyb = min((mxy - kHalf) >> 8, H - 1);
    if (s.pxa > s.pxb || s.pya > s.pyb) return false;    // the box holds no pixel centre (most micro-triangles end here)
    long long area2 = (long long)(s.x1 - s.x0) * (s.y2 - s.y0) - (long long)(s.y1 - s.y0) * (s.x2 - s.x0);
    if (area2 == 0) return false;
    if (area2 < 0) { int t; t = s.x1; s.x1 = s.x2; s.x2 = t; t = s.y1; s.y1 = s.y2; s.y2 = t; }
    return true;
}

__device__ __forceinline__ bool covers(const TriSetup& s, int px, int py)
{
    int cx = px * kSub + kHalf, cy = py * kSub + kHalf;
    return edge_in(s.x0, s.y0, s.x1, s.y1, cx, cy) && edge_in(s.x1, s.y1, s.x2, s.y2, cx, cy) &&
           edge_in(s.x2, s.y2, s.x0, s.y0, cx, cy);
}

// `dirty` (nullable): one byte per 64-pixel segment of the flattened [n*H*W] pixel index; set when a fragment lands there, so
// the resolve pass (and everything after it) only touches segments that can hold coverage
__device__ __forceinline__ void emit_fragment(const float4 p0, const float4 p1, const float4 p2, int px, int py, int f,
                                              float xs, float xo, float ys, float yo, unsigned long long* keys_view, int W,
                                              unsigned char* dirty, long long pixbase)
{
    float zw;                                                        // the key needs the depth only: the resolve pass evaluates the winner's (u, v)
    if (!fragment_zw(p0, p1, p2, px, py, xs, xo, ys, yo, zw)) return;
    unsigned long long key = ((unsigned long long)mono_key(zw) << 32) | ((unsigned)f << 1) | 1u;   // bit 0: large-box triangle
    const size_t pix = (size_t)py * W + px;
    atomicMin(&keys_view[pix], key);
    if (dirty != nullptr) dirty[(pixbase + (long long)pix) >> 6] = 1;
}

// antialias rule A3 on the snapped integer coordinates (same decision as antialias.hip: silhouette_bits_snapped): bit k set =
// the edge opposite vertex k is a silhouette edge in this view (its neighbour is missing, rejected or on the same side)
__device__ __forceinline__ unsigned char silh_bits(const int2 s0, const int2 s1, const int2 s2, int2 q0, int2 q1, int2 q2)
{
    const int REJ = (int)0x80000000;
    if (q0.x == REJ) q0 = s0;
    if (q1.x == REJ) q1 = s1;
    if (q2.x == REJ) q2 = s2;
    long long bb = (long long)(s1.x - s0.x) * (s2.y - s0.y) - (long long)(s2.x - s0.x) * (s1.y - s0.y);
    long long a0 = (long long)(s1.x - q0.x) * (s2.y - q0.y) - (long long)(s2.x - q0.x) * (s1.y - q0.y);
    long long a1 = (long long)(s2.x - q1.x) * (s0.y - q1.y) - (long long)(s0.x - q1.x) * (s2.y - q1.y);
    long long a2 = (long long)(s0.x - q2.x) * (s1.y - q2.y) - (long long)(s1.x - q2.x) * (s0.y - q2.y);
    return (unsigned char)((((a0 < 0) == (bb < 0)) ? 1 : 0) | (((a1 < 0) == (bb < 0)) ? 2 : 0) | (((a2 < 0) == (bb < 0)) ? 4 : 0));
}

template <class SnapFn>
__device__ __forceinline__ void write_silh(unsigned char* __restrict__ silh, const int32_t* __restrict__ opp, const SnapFn& snapv,
                                           const int2 s0, const int2 s1, const int2 s2, int b, int f, int F)
{
    if (silh == nullptr) return;
    const int REJ = (int)0x80000000;
    int o0 = opp[f * 3], o1 = opp[f * 3 + 1], o2 = opp[f * 3 + 2];
    int2 q0 = (o0 < 0) ? make_int2(REJ, 0) : snapv(o0), q1 = (o1 < 0) ? make_int2(REJ, 0) : snapv(o1),
         q2 = (o2 < 0) ? make_int2(REJ, 0) : snapv(o2);
    silh[(size_t)b * F + f] = silh_bits(s0, s1, s2, q0, q1, q2);
}

// ---- binned rasterizer: fragments go into per-segment bins instead of a global depth-key buffer -----------------------------------
// north_star: "LDS-staged per-tile triangle bins and wave64 ballot / prefix-scan fragment compaction".  A bin is one 64-pixel row
// segment (the tile a wave of the per-pixel kernels owns: everything downstream keeps its layout).  The triangle kernel appends every
// fragment (face | flags | pixel-in-segment, z/w: 8 B since round 4) to its segment's bin; k_raster_bins — one wave per segment — stages
// the bin in LDS, takes the depth minimum per pixel in LDS (ds_min_u64 on (monotone(z/w), face, entry index)), evaluates the winners'
// barycentrics and runs the resolve + forward-shading epilogue: the 8-B-per-pixel global key buffer (100 MB at the metric), its
// 2 M atomicMin line requests, its read-back + reset and the dirty-segment flags are gone; the depth is evaluated once per fragment
// (fragment_zw), the barycentrics once per covered pixel (fragment_uv).
// Slots are reserved per WORKGROUP: the fragments of a workgroup's 256 consecutive faces fall into a few dozen segments; a direct-
// mapped LDS table (segment -> count; rows mod 64 x two column segments) gives every fragment its rank inside (workgroup, segment)
// with one ds_cmpst + one ds_add_rtn, then ONE returning global atomic per table entry reserves the run — all of them in one
// instruction, in flight while the workgroup evaluates its fragments (a returning atomic per wave and round, waited for on the spot,
// cost 124 us; with the cursors packed 16 per cache line 280 us).
struct BinArgs { BinEntry* bins; int* cursor; int cap; int segsX; int segsPerView; int tiles; };
// Per-segment STATE word (packed: 16 per line, unlike the cursors), so that the second pass can leave an empty segment — 70 % of them at
// the metric — on a load that hits: the cursor of a segment no fragment ever reaches is a line of its own that only the second pass
// reads, once per step, from memory (196 k empty waves: 17 us of that pass; with the state words k_raster_bins 82.8 -> 74.4 us).
// bit 0: the segment has fragments this step — written (3) by the workgroup whose reservation found the cursor at 0 (the slow path's
// runs: by every one of them); bit 1: the segment's words (coverage, valid count) may be non-zero from the last step that had
// fragments — the second pass writes 2 when it resolves a bin, and an empty segment zeroes its words only if bit 1 is set (then
// writes 0): a segment that stays empty costs one scalar load and no store.
// The words lie right BELOW the cursors, in reverse order (segment s: cursor[-1 - s]): no pointer of their own to hold — the
// triangle kernel has no register to spare (see face() there).
constexpr int kSegHasFragments = 1, kSegWordsLive = 2;
__device__ __forceinline__ int* seg_state_word(const BinArgs& ba, long long bin) { return ba.cursor - 1 - bin; }
// ... and below them the bin-overflow word of the launch pair (k_raster_bins adds it to the host-visible sticky flag)
__device__ __forceinline__ int* bin_overflow_word(const BinArgs& ba) { return ba.cursor - 2 - ba.tiles; }
// (face | flags | pixel, z/w[, u, v]) -> an entry
__device__ __forceinline__ BinEntry make_entry(unsigned head, float zw) { return make_uint2(head, __float_as_uint(zw)); }
constexpr int kHash = 128;                       // LDS table slots per workgroup round (a round has at most 512 fragments)
constexpr unsigned kBinInvalid = 0xFFFFFFFFu;    // first word of a reserved entry whose fragment was degenerate

// Slow path (a table that ran full, large-box triangles swept by their wave): wave-uniform call; every RUN of consecutive lanes with
// the same bin reserves its slots with one returning atomic (all runs of the wave in one instruction) and stores to consecutive slots.
__device__ __forceinline__ void bin_append(bool valid, int bin, const BinEntry e, const BinArgs& ba, int lane)
{
    const int tl = valid ? bin : -1;
    const int prev = wave_prev_lane(tl, -2);
    const bool leader = valid && prev != tl;
    const unsigned long long lm = __ballot(leader);
    if (lm == 0ull) return;                                          // wave-uniform: nothing to append
    const unsigned long long brk = lm | ~__ballot(valid);            // a run ends at the next leader or the next lane without a fragment
    const unsigned long long upto = (2ull << lane) - 1ull;           // bits 0..lane (lane 63: 2 << 63 wraps to 0 -> all ones)
    int base = 0;
    if (leader) {
        const unsigned long long above = brk & ~upto;
        const int len = (above != 0ull ? __builtin_ctzll(above) : 64) - lane;
        base = atomicAdd(ba.cursor + (size_t)tl * kCursorStride, len);
        *seg_state_word(ba, tl) = kSegHasFragments | kSegWordsLive;      // marks the segment for the second pass (this rare path: every run does)
    }
    const int start = 63 - __builtin_clzll((lm & upto) | 1ull);      // the run's first lane (meaningless on lanes without a fragment)
    const int b0 = __shfl(base, start, 64);
    if (valid) {
        const int slot = b0 + (lane - start);
        if (slot < ba.cap) ba.bins[(size_t)tl * ba.cap + slot] = e;
        else *bin_overflow_word(ba) = 1;                             // reported by the host (HamStage2.loss_dict): the bin capacity was too small
    }
}

// one lane per (view, triangle): 3 x 8-B snapped-vertex gathers, integer setup, integer coverage mask of the (<= 32-pixel) box.
// The fragments are then emitted WAVE-cooperatively: the covering lanes park their triangle (clip positions, box origin) in LDS
// and expand their coverage bits into a dense per-wave task list; every lane then takes tasks round-robin.  A lane looping over
// its own bits made the whole wave iterate max(bits) times with ~a quarter of the lanes active.
// The kernel is bound by the NUMBER OF VECTOR-ALU INSTRUCTIONS it issues (profiles/r02_v62_issue.txt: 755 per wave x 4 cycles = 92 %
// of its duration), so everything here is written for instruction count: the grid is 2-D (view = blockIdx.y: no index division,
// every wave lies in one view, the view's matrix sits in SGPRs), the pixel-grid constants come from the host, the coverage
// mask is 32 bits wide and filled by ONE flat loop over the box (nested row / column loops run max(rows) x max(columns) of the
// wave's lanes), and a task's pixel is decoded with a float multiply instead of an integer division.
FMHR_WTS_DEFINE(raster)
#ifndef FMHR_TRIS_BLOCK
#define FMHR_TRIS_BLOCK 256
#endif
constexpr int kTrisBlock = FMHR_TRIS_BLOCK;     // threads (= faces of one view) per workgroup of k_raster_tris
constexpr int kTaskCap = 1024;       // tasks the workgroup expands per round (it can hold 256 x 32; more rounds then)

struct TriPark {
    float4 p0, p1, p2;
    int f; float rbw; int origin, bw;        // origin = pxa | pya << 16; rbw ~ 1 / bw (task -> pixel decode)
};
// a large-box triangle handed to its whole wave: integer setup, clip positions, face
struct LargePark {
    TriSetup s;
    float4 p0, p1, p2;
    int f;
};

// One large-box triangle rasterized by a whole WAVE, 64 box pixels per iteration (wave-uniform arguments).  Used in place by the wave
// of the lane that found it, and by the worker waves at the head of the launch for the slivers that were already there last step.
// `q`: the parked triangle — a reference into LDS for the in-place sweep (read at the use: nothing of it stays in registers over the
// loop, the triangle kernel sits on its register boundary), a local for the worker waves
template <bool BIN>
__device__ __forceinline__ void sweep_large_box(const LargePark& q, const int b, const int lane, const BinArgs& ba, const PixGrid& pg,
                                                unsigned long long* __restrict__ keys_view, const int Wp, unsigned char* __restrict__ dirty,
                                                const long long pixbase)
{
    const int lbw = q.s.pxb - q.s.pxa + 1, lbh = q.s.pyb - q.s.pya + 1, np = lbw * lbh;
    for (int i0 = 0; i0 < np; i0 += 64) {                            // wave-uniform trip count (bin_append is a wave-level operation)
        const int i = i0 + lane;
        int c;
        const int r = fast_divmod(min(i, np - 1), lbw, c);
        const int px = q.s.pxa + c, py = q.s.pya + r;
        const bool in = i < np && covers(q.s, px, py);
        if (BIN) {
            float zw = 0.f;
            const bool okf = in && fragment_zw(q.p0, q.p1, q.p2, px, py, pg.xs, pg.xo, pg.ys, pg.yo, zw);
            const BinEntry e = make_entry((unsigned)q.f | (1u << 22) | ((unsigned)(px & 63) << 26), zw);
            bin_append(okf, b * ba.segsPerView + py * ba.segsX + (px >> 6), e, ba, lane);
        } else if (in) {
            emit_fragment(q.p0, q.p1, q.p2, px, py, q.f, pg.xs, pg.xo, pg.ys, pg.yo, keys_view, Wp, dirty, pixbase);
        }
    }
}

// Forward workers for PERSISTENT slivers (round 4).  An optimisation run grows sliver triangles whose pixel box exceeds the 32-pixel
// coverage mask (0 at step 32 of the metric run, ~2 000 (view, face)s at step 640, ~3 800 at step 960: DESIGN.md §5.7); each was swept
// in place by the whole wave of its lane AFTER that wave's main work, its workgroup's LDS and wave slots held meanwhile, the stragglers
// among them the kernel's tail: k_raster_tris 85 us in the bench's window, 93-97 at step 640, 100-103 at step 960.  A sliver lives for
// hundreds of steps, so last step's queue of large-box (view, face)s — the one the backward's workers consume — is this step's work
// list: kSweepWorkers workgroups at the HEAD of the launch take one queued (view, face) per wave, redo its integer setup on THIS step's
// snapped vertices, and sweep it if its box is still large; the lane that owns the triangle skips the sweep exactly then — it sees the
// marker entry its predecessor left in cov16 (width field 63, stamped with last step) next to a box that is large again.  Both sides
// decide from the same two facts (queued last step <=> marker of last step; box large now), so every large box is swept exactly once:
// by a worker if it persisted, by its own wave if it is new (and a queued face that shrank is an ordinary small box again).  The queue
// is double-buffered by the step's parity: this step's pushes must not overwrite the list the workers are reading.
constexpr int kSweepWorkers = 256;              // workgroups (x 4 waves) behind view 0's triangle workgroups: ~3 800 queued (view, face)s at step 960 = 4 per wave (32: 30 per wave, the workers themselves became the tail)
template <bool FV, bool BIN>
__device__ __forceinline__ void
sweep_workers(LargePark* const park, const int first, const int stride, const float* __restrict__ pos, const int2* __restrict__ snap, const int32_t* __restrict__ tri,
              int V, int F, int H, int W, int Wp, unsigned long long* __restrict__ keys, unsigned char* __restrict__ dirty,
              const float* __restrict__ verts, const float4* __restrict__ cmat, const int* __restrict__ large_cnt,
              const int2* __restrict__ prev_q, const PixGrid& pg, const BinArgs& ba)
{
    const int lane = threadIdx.x & 63;
    const int cnt = large_cnt[1];                                    // last step's count (the second rasterizer pass kept it)
    for (int it = first; it < cnt; it += stride) {                   // wave-uniform
        const int2 bf = prev_q[it];
        const int b = __builtin_amdgcn_readfirstlane(bf.x), f = __builtin_amdgcn_readfirstlane(bf.y);
        if ((unsigned)f >= (unsigned)F) continue;
        const int i0 = tri[f * 3], i1 = tri[f * 3 + 1], i2 = tri[f * 3 + 2];
        if ((unsigned)i0 >= (unsigned)V || (unsigned)i1 >= (unsigned)V || (unsigned)i2 >= (unsigned)V) continue;
        const int2* sv = snap + (size_t)b * V;
        TriSetup ts;
        if (!setup_triangle(sv[i0], sv[i1], sv[i2], H, W, ts)) continue;
        if ((ts.pxb - ts.pxa + 1) * (ts.pyb - ts.pya + 1) <= kSmallBox) continue;   // shrank: its own lane handles a small box
        // parked in the wave's LDS slot like an in-place sweep's triangle: nothing of it is live in registers over the sweep loop
        if (lane == 0) {
            const ClipView cv = clip_view(FV ? nullptr : pos, verts, cmat, b, V);
            park->s = ts; park->p0 = cv.get(i0); park->p1 = cv.get(i1); park->p2 = cv.get(i2); park->f = f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const long long P = (long long)H * Wp;
        sweep_large_box<BIN>(*park, b, lane, ba, pg, keys + (size_t)b * P, Wp, dirty, (long long)b * P);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                             // every lane has read the slot before the next triangle is parked
    }
}

// FV (fused step): no stored clip positions — a lane whose triangle covers a pixel centre transforms its three vertices with the
// view's combined matrix (the same clip_coord() chains everywhere: bit-identical); the [V,3] vertex array stays in cache, the
// [n,V,4] position array (37 MB written + gathered per 48 views) does not exist.
// 64 VGPRs and < 20 KB of LDS per workgroup: 8 waves per SIMD (5 / 6 / 7 / 8 waves: 105 / 100 / 97.5 / 94 us)
// the binned instantiation needs 72 VGPRs: at the 64 of 8 waves per SIMD it spills 24 of them into its task loop (156 vs 88 us)
#ifndef FMHR_BIN_TRIS_WAVES
#define FMHR_BIN_TRIS_WAVES 7
#endif
template <bool FV, bool BIN, bool WORKERS = false>
#ifndef FMHR_WORKER_TRIS_WAVES
#define FMHR_WORKER_TRIS_WAVES 8
#endif
__global__ void __launch_bounds__(kTrisBlock) __attribute__((amdgpu_waves_per_eu((BIN && !WORKERS) ? FMHR_BIN_TRIS_WAVES : (WORKERS ? FMHR_WORKER_TRIS_WAVES : 8), 8)))    // (WORKERS: 63 VGPRs at 8 waves, scalar spills only)
k_raster_tris(const float* __restrict__ pos, const int2* __restrict__ snap, const int32_t* __restrict__ tri,
              const int32_t* __restrict__ opp, unsigned char* __restrict__ silh, int n, int V,
              int F, int H, int W, int Wp, unsigned long long* __restrict__ keys, unsigned char* __restrict__ dirty,
              int2* __restrict__ cov16, const int* __restrict__ stamp, const float* __restrict__ verts,
              const float4* __restrict__ cmat, int* __restrict__ large_cnt, int2* __restrict__ large_q, const PixGrid pg,
              unsigned char* __restrict__ own8, const BinArgs ba, const int defer_large, const int sweep_workers_n)
{
    // (the large-box queue is double-buffered by the step's parity — two halves of n * F entries: this step's pushes go to one half
    // while the forward workers read the half last step filled.  The half is picked where it is used: kept from the top of the kernel
    // the two extra scalar values pushed the kernel past its SGPR budget — 20 spilled SGPRs, and through them 52 spilled VGPRs)
    FMHR_WTS_BEGIN(raster, ((long long)blockIdx.y * gridDim.x + blockIdx.x) * (kTrisBlock / 64) + (threadIdx.x >> 6));
    // the parked triangles as FOUR arrays of 16-B words (a 64-B record per lane put the lanes of a ds_read_b128 lane group on
    // 4 of the 16 bank quads: SQ_LDS_BANK_CONFLICT 2.4e7 cycles per launch, profiles/r02_v72_issue.txt)
    __shared__ float4 pk0[kTrisBlock], pk1[kTrisBlock], pk2[kTrisBlock];
    __shared__ int4 pk3[kTrisBlock];                                 // f, rbw (float bits), origin, bw
    // the workgroup's task list: bit of its mask | lane (8 bits) << 5 — 16-bit entries, 1024 per round; the binned form keeps the
    // fragment's table slot and rank with it (32-bit entries, 512 per round: the same 2 KB)
    __shared__ unsigned tasks32[kTaskCap / 2];
    unsigned short* const tasks = reinterpret_cast<unsigned short*>(tasks32);
    __shared__ int hkey[BIN ? kHash : 1], hcnt[BIN ? kHash : 1];    // binned form: segment (inside the view) -> fragments of this workgroup | base slot
    __shared__ int wtot[kTrisBlock / 64];
    __shared__ LargePark lpark[kTrisBlock / 64];
    if (WORKERS) {                                                   // (fused step, binned form: a micro-triangle mesh)
        // the LAST sweep_workers_n workgroups of a row are set aside; those of row 0 — dispatched right behind view 0's triangle
        // workgroups, ahead of the other n - 1 rows — are the workers.  (Behind, not in front: an offset subtracted from blockIdx.x on
        // the triangle path costs the compiler the index's range — 64-bit address products, 52 spilled registers)
        const int wi = (int)blockIdx.x - ((int)gridDim.x - sweep_workers_n);
        if (wi >= 0) {
            if (blockIdx.y == 0)
                sweep_workers<FV, BIN>(lpark + (threadIdx.x >> 6), wi * (kTrisBlock / 64) + (int)(threadIdx.x >> 6), sweep_workers_n * (kTrisBlock / 64), pos, snap, tri, V, F,
                                       H, W, Wp, keys, dirty, verts, cmat, large_cnt, large_q + ((stamp[0] & 1) ? 0 : (size_t)n * F), pg, ba);
            return;
        }
    }
    const int f = blockIdx.x * kTrisBlock + threadIdx.x;
    // ... and the same index RECOMPUTED at its late uses (one v_lshl_add from the workgroup id and the thread id, both live anyway)
    // instead of carried: the allocator's choice of what to spill at the 64-register pin is this value — as a 16-byte scratch store
    // and six scratch reloads on the hot path (k_raster_tris 83 -> 111 us when one more store tipped it over)
    auto face = [&]() -> int {
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));
        return (int)(blockIdx.x * kTrisBlock) + t;
    };
    const int b = blockIdx.y;                                        // wave-uniform (SGPR)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Owner marks (resolve pass -> face-parallel backward) are the low BYTE of the step that wrote them, and the backward trusts a
    // matching mark: a (view, face) that owned a pixel in step s, covers no pixel centre afterwards while that pixel stays
    // uncovered, would meet its stale mark, stale coverage entry, stale pixel word and stale image gradient again in step s + 256.
    // So no mark lives longer than 128 steps: every 128th step this kernel — one lane per (view, face), ahead of the resolve pass
    // that writes the step's fresh marks — resets all of them to a value that cannot match before the next reset (scalar test on
    // the other 127 steps: nothing on the hot path).  tests/test_gpu_ham.py::test_stale_owner_mark_after_256_steps
#ifndef FMHR_X_NO_MARK_WIPE                                          // (measurement / regression-validation builds only)
    if (own8 != nullptr) {
        const int st = stamp[0];                                     // scalar load, wave-uniform
        if ((st & 127) == 0 && f < F) own8[(size_t)b * F + f] = (unsigned char)(st + 128);
    }
#endif
    unsigned cm = 0u;
    bool is_large = false, by_worker = false;
    int i0 = 0, i1 = 0, i2 = 0, o0 = -1, o1 = -1, o2 = -1, bw = 1;
    int2 s0 = make_int2(0, 0), s1 = s0, s2 = s0;
    TriSetup s;
    s.pxa = s.pya = 0;
    const int2* sv = snap + (size_t)b * V;
    bool ok = f < F;
    if (ok) {
        const I3U row = at32(reinterpret_cast<const I3U*>(tri), (unsigned)f);
        i0 = row.x; i1 = row.y; i2 = row.z;
        ok = (unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V;
    }
    // the view's combined matrix (FV): wave-uniform, through the scalar cache into SGPRs
    float cmx[16];
    if (FV) {
        const_as4_float* cp = (const_as4_float*)(unsigned long long)(cmat + (size_t)b * 4);
#pragma unroll
        for (int k = 0; k < 16; k++) cmx[k] = cp[k];
    }
    auto clipv = [&](int v) -> float4 {                              // clip position of vertex v in this view
        if (!FV) return ldg4(pos + ((size_t)b * V + v) * 4);
        const F3U x = at32(reinterpret_cast<const F3U*>(verts), (unsigned)v);
        return make_float4(clip_coord(x.x, x.y, x.z, cmx[0], cmx[4], cmx[8], cmx[12]), clip_coord(x.x, x.y, x.z, cmx[1], cmx[5], cmx[9], cmx[13]),
                           clip_coord(x.x, x.y, x.z, cmx[2], cmx[6], cmx[10], cmx[14]), clip_coord(x.x, x.y, x.z, cmx[3], cmx[7], cmx[11], cmx[15]));
    };
    // the snapped vertices (R1 + R2 of the rasterizer) stay a stored [n,V] array also in the fused step: snapping in every lane
    // (an IEEE division and a dozen compare / round steps per vertex, for all 4.7 M (view, triangle) pairs) doubled this kernel
    auto snapv = [&](int v) -> int2 { return at32(sv, (unsigned)v); };
    if (ok) {
        s0 = snapv(i0); s1 = snapv(i1); s2 = snapv(i2);
        ok = setup_triangle(s0, s1, s2, H, W, s);
    }
#ifdef FMHR_X_TRIS_STOP                                              // (measurement builds only, DESIGN.md 5.11: 1 = nothing after the integer setup,
    if (FMHR_X_TRIS_STOP == 1) ok = false;                           //  2 = nothing after the coverage masks, 3 = no fragment rounds)
#endif
    if (ok) {
        bw = s.pxb - s.pxa + 1;
        const int bh = s.pyb - s.pya + 1;
        const int box = bw * bh;                                     // bw <= W, bh <= H <= 8192: no overflow
        if (box > kSmallBox) {
            // a mesh of large triangles (a 778-vertex hand at 512 x 334: every triangle) goes through a queue and one workgroup
            // per triangle (k_raster_large: the triangle kernel's own grid is then a few dozen workgroups); on a micro-triangle
            // mesh the odd large box is swept by its wave below and that kernel is not launched at all
            is_large = true;
            write_silh(silh, opp, snapv, s0, s1, s2, b, f, F);
            // coverage entry of a large-box triangle: a marker (width field 63) that hides whatever small-box entry an earlier step left
            // here — the backward finds a large-box face's pixels by scanning its box (shade.hip: face_bwd_large_items).  The marker the
            // PREVIOUS step left says whether this (view, face) is on the forward workers' list: then one of them sweeps it (above)
            if (cov16 != nullptr) {
                if (WORKERS) {
                    const int old = cov16[(size_t)b * F + f].y;
                    const unsigned prev = ((unsigned)stamp[0] - 1u) & 0x3FFFFFFu;
                    // (a never-written entry is all ones — the pattern of fmhr_ham_workspace_init — and must not pass for last step's marker
                    // in the one step in 2^26 whose predecessor's stamp is all ones: then the face is swept here AND by a worker, twice the
                    // same fragments, harmless)
                    by_worker = ((unsigned)old >> 26) == 63u && ((unsigned)old & 0x3FFFFFFu) == prev && prev != 0x3FFFFFFu;
                }
                cov16[(size_t)b * F + f] = make_int2(-1, (int)((63u << 26) | ((unsigned)stamp[0] & 0x3FFFFFFu)));
            }
        } else {
            // the opposite-vertex ids (antialias rule A3) are requested now and arrive while the coverage mask is computed
            if (silh != nullptr) { const I3U orow = at32(reinterpret_cast<const I3U*>(opp), (unsigned)f); o0 = orow.x; o1 = orow.y; o2 = orow.z; }
            const int ext = max(max(s.x0, max(s.x1, s.x2)) - min(s.x0, min(s.x1, s.x2)),
                                max(s.y0, max(s.y1, s.y2)) - min(s.y0, min(s.y1, s.y2)));
            // 32-bit edge arithmetic is exact when the snapped extent is below 2^14 sub-pixel units (64 px): pixel centres lie
            // within the box +-1 px, so every product stays below 2^29 (every small-box triangle unless it pokes far off screen)
            if (ext < 16384) {
                // incremental edge functions: E_k = e_k - (tie rule admits e_k == 0 ? 0 : 1), inside <=> all three E_k >= 0 <=>
                // the OR of the three has a clear sign bit; a step to the right adds sx_k, the wrap to the next row's first
                // pixel adds wr_k = sy_k - (bw - 1) * sx_k (all values stay below 2^30)
                const int cx0 = s.pxa * kSub + kHalf, cy0 = s.pya * kSub + kHalf;
                int e0, e1, e2, sx0, sx1, sx2, wr0, wr1, wr2;
                {
                    int dx = s.x1 - s.x0, dy = s.y1 - s.y0;
                    e0 = dx * (cy0 - s.y0) - dy * (cx0 - s.x0) - (((dy > 0) || (dy == 0 && dx > 0)) ? 0 : 1);
                    sx0 = -dy * kSub; wr0 = dx * kSub - (bw - 1) * sx0;
                    dx = s.x2 - s.x1; dy = s.y2 - s.y1;
                    e1 = dx * (cy0 - s.y1) - dy * (cx0 - s.x1) - (((dy > 0) || (dy == 0 && dx > 0)) ? 0 : 1);
                    sx1 = -dy * kSub; wr1 = dx * kSub - (bw - 1) * sx1;
                    dx = s.x0 - s.x2; dy = s.y0 - s.y2;
                    e2 = dx * (cy0 - s.y2) - dy * (cx0 - s.x2) - (((dy > 0) || (dy == 0 && dx > 0)) ? 0 : 1);
                    sx2 = -dy * kSub; wr2 = dx * kSub - (bw - 1) * sx2;
                }
                unsigned om = 0u;                                    // OUTSIDE bits (the sign of the OR), complemented below
                int c = bw;                                          // columns left in the current row
                for (int bit = 0; bit < box; bit++) {
                    om |= ((unsigned)(e0 | e1 | e2) >> 31) << bit;
                    c--;
                    const bool wrap = c == 0;
                    e0 += wrap ? wr0 : sx0; e1 += wrap ? wr1 : sx1; e2 += wrap ? wr2 : sx2;
                    c = wrap ? bw : c;
                }
                cm = ~om & (0xffffffffu >> (32 - box));              // 1 <= box <= 32
            } else {
                int bit = 0;
                for (int py = s.pya; py <= s.pyb; py++)
                    for (int px = s.pxa; px <= s.pxb; px++, bit++)
                        if (covers(s, px, py)) cm |= 1u << bit;
            }
        }
    }
#ifdef FMHR_X_TRIS_STOP
    if (FMHR_X_TRIS_STOP == 1) { cm = 0u; is_large = false; }
    if (FMHR_X_TRIS_STOP == 2) { cm = 0u; }
#endif
    // ---- task numbering over the WORKGROUP: wave-wide exclusive scan of the per-lane fragment counts + the waves' totals.  The
    // four waves share one task list and take dense 64-task slices of it: a wave of a sub3-class mesh has 64 +- 15 fragments, so
    // with a list of its own every other wave ran a second, nearly empty round of the ~110-instruction fragment evaluation (1.56
    // rounds per wave; pooled: ~1.1) and issued its atomics in two instructions that touched some key lines twice
    const int cnt = __popc(cm);
    const int incl = wave_incl_scan(cnt);                            // DPP: no LDS-crossbar instructions
    if (lane == 63) wtot[wv] = incl;
    unsigned long long lbal = __ballot(is_large);                    // lanes holding a large-box triangle
    // Large-box (view, face)s go on a queue: one cursor atomic per wave (same-address atomics serialise at ~10 ns each: 1.5 k pushes
    // of a hand in one view).  A coarse mesh (defer_large) leaves their rasterization to k_raster_large, one workgroup per triangle; on a
    // micro-triangle mesh the odd large box is swept by its wave below and the queue only feeds the BACKWARD's wave-per-face workers
    // (shade.hip: face_bwd_large_items)
#ifdef FMHR_X_NO_QPUSH                                               // (measurement builds only: what the queue's one cursor costs late in a run)
    if (large_q != nullptr && lbal != 0ull && defer_large) {
#else
    if (large_q != nullptr && lbal != 0ull) {
#endif
        int base = 0;
        const int leader = __builtin_ctzll(lbal);
        if (lane == leader) base = atomicAdd(large_cnt, __popcll(lbal));
        base = __shfl(base, leader, 64);
        int2* const lq2 = large_q + ((stamp != nullptr && (stamp[0] & 1)) ? (size_t)n * F : 0);      // this step's half of the queue
        if (is_large) lq2[base + __popcll(lbal & ((1ull << lane) - 1ull))] = make_int2(b, face());
        if (defer_large) {
            is_large = false;
            lbal = 0ull;
        }
    }
    // a sliver that was queued last step is swept by a worker wave at the head of this launch, not here
    if (by_worker) is_large = false;
    lbal = __ballot(is_large);
    if (cm != 0u) {
        // one memory round trip for everything a covering triangle needs: clip positions + the snapped opposite vertices
        float4 p0 = clipv(i0), p1 = clipv(i1), p2 = clipv(i2);
        const int2 rej = make_int2(kRejected, 0);
        const int2 q0 = (o0 < 0) ? rej : snapv(o0), q1 = (o1 < 0) ? rej : snapv(o1), q2 = (o2 < 0) ? rej : snapv(o2);
        pin(p0); pin(p1); pin(p2);
        // only a triangle that covers a pixel centre can own a pixel, and only an owner's bits are ever read
        const unsigned char sbv = (silh != nullptr) ? silh_bits(s0, s1, s2, q0, q1, q2) : (unsigned char)0;
#ifndef FMHR_X_NO_SILH_STORE                                          // (measurement builds only: the store classes of this kernel, tools/r5_tris_writes.sh)
        if (silh != nullptr) at32(silh + (size_t)b * F, (unsigned)face()) = sbv;
#endif
        // coverage entry (face-parallel backward), 8 B: candidate-pixel mask of the box | origin x, y (13 bits each) and width - 1 (6 bits,
        // 63 = the marker entry of a large-box face, which carries the step stamp instead of a box).  A small-box entry needs no
        // stamp: the backward reads it only under a matching owner mark, an owner's entry was written this step, and a stale mark
        // (8 bits) meets a stale entry only to have its candidates fail the owner test
#ifndef FMHR_X_NO_COV16_STORE
        if (cov16 != nullptr)
            at32(cov16 + (size_t)b * F, (unsigned)face()) = make_int2((int)cm, s.pxa | (s.pya << 13) | ((bw - 1) << 26));
#endif
        pk0[threadIdx.x] = p0; pk1[threadIdx.x] = p1; pk2[threadIdx.x] = p2;
        pk3[threadIdx.x] = make_int4(face(), __float_as_int(__builtin_amdgcn_rcpf((float)bw)), s.pxa | (s.pya << 16), bw | ((int)sbv << 8));   // (silhouette bits ride along: binned form)
    }
    __syncthreads();                                                 // every wave of the workgroup gets here (no early return above)
    int wbase = 0, T = 0;
#pragma unroll
    for (int i = 0; i < kTrisBlock / 64; i++) {
        const int w = wtot[i];
        if (i < wv) wbase += w;
        T += w;
    }
#ifdef FMHR_X_TRIS_STOP
    if (FMHR_X_TRIS_STOP == 3) T = 0;
#endif
    const int excl = wbase + incl - cnt;                             // this lane's first task in the workgroup's numbering
    const long long P = (long long)H * Wp;                           // Wp: row pitch of the per-pixel buffers (common.h); W clips and scales
    unsigned long long* const keys_view = keys + (size_t)b * P;      // wave-uniform bases: the per-task offsets are 32-bit
    const long long segbase = ((long long)b * P) >> 6;
    const int segrem = (int)(((long long)b * P) & 63);
    if (BIN) {
        // ---- binned form: tasks -> (table slot, rank) -> one reservation per (workgroup, segment) -> entries
        constexpr int kCap = kTaskCap / 2;                           // 32-bit task entries: bit | lane << 5 | slot << 13 | rank << 21
        static_assert(kTrisBlock == 256 && kHash <= kTrisBlock && kCap <= 2048, "8-bit lane, 8-bit slot and 11-bit rank fields of a task");
        const int segsX = ba.segsX;
        auto decode = [&](unsigned e, int& o, int& px, int& py) {
            o = (int)(e >> 5) & 255;
            const int bit = (int)(e & 31u);
            const int4 m = pk3[o];
            float qf = (float)bit + 0.5f;
            qf = qf * __int_as_float(m.y);
            const int r = (int)qf, c = bit - r * (m.w & 0xff);
            px = (m.z & 0xffff) + c; py = (m.z >> 16) + r;
            return m.x | ((m.w >> 8) << 23);                         // the face | its silhouette bits in this view << 23
        };
        for (int R = 0; R < T; R += kCap) {                          // workgroup-uniform trip count (one round on a micro-triangle mesh)
            if (R > 0) __syncthreads();                              // the previous round's list and table have been consumed
            if (threadIdx.x < kHash) { hkey[threadIdx.x] = -1; hcnt[threadIdx.x] = 0; }
            if (cm != 0u && excl < R + kCap && excl + cnt > R) {
                unsigned m = cm;
                int idx = excl - R;
                while (m != 0u) {
                    const int bit = __builtin_ctz(m);
                    m &= m - 1u;
                    if (idx >= 0 && idx < kCap) tasks32[idx] = (unsigned)((threadIdx.x << 5) | bit);
                    idx++;
                }
            }
            __syncthreads();
            const int nt = min(kCap, T - R);
            // phase 1: every fragment's segment -> table slot (open addressing, ds_cmpst) and its rank inside (workgroup, segment)
            for (int t = threadIdx.x; t < nt; t += kTrisBlock) {
                const unsigned e = tasks32[t];
                int o, px, py;
                decode(e, o, px, py);
                const int sg = py * segsX + (px >> 6);               // segment inside the view: < 2^20
                // the segments of 256 consecutive faces are a compact patch: rows (mod 64) x two column segments map it onto the
                // table without collisions; a slot that is taken by another segment (a patch taller than 64 rows, faces in random
                // order) sends the fragment down the slow path (255)
                const int h = ((py & 63) << 1) | ((px >> 6) & 1);
                const int old = atomicCAS(&hkey[h], -1, sg);
                const int slot = (old == -1 || old == sg) ? h : 255;
                unsigned rank = 0u;
                if (slot != 255) rank = (unsigned)atomicAdd(&hcnt[slot], 1);
                tasks32[t] = e | ((unsigned)slot << 13) | (rank << 21);
            }
            __syncthreads();
            // phase 2: one returning atomic per table entry, all in one instruction of the first two waves; not waited for yet
            const bool own = threadIdx.x < kHash && hkey[threadIdx.x] >= 0;
            int gbase = 0;
            if (own) {
                const size_t bin = (size_t)b * ba.segsPerView + hkey[threadIdx.x];
                gbase = atomicAdd(ba.cursor + bin * kCursorStride, hcnt[threadIdx.x]);
            }
            // phase 3: the fragments (the ~110-instruction evaluation hides the reservations' round trip)
            auto evaluate = [&](int t, unsigned& e, int& bin, BinEntry& ent) {
                e = tasks32[t];
                int o, px, py;
                const int face = decode(e, o, px, py);
                bin = b * ba.segsPerView + py * segsX + (px >> 6);
                // rule R5, the depth only (common.h: fragment_zw — fragment()'s statements for z/w and for the accept / reject decision;
                // the barycentrics are evaluated by the second pass, for the winners)
                float zw = 0.f;
                const bool okf = fragment_zw(pk0[o], pk1[o], pk2[o], px, py, pg.xs, pg.xo, pg.ys, pg.yo, zw);
                ent = make_entry(okf ? ((unsigned)face | ((unsigned)(px & 63) << 26)) : kBinInvalid, zw);
            };
            auto store = [&](bool have, unsigned e, int bin, const BinEntry ent) {
                const int slot = (int)(e >> 13) & 255;
                if (have && slot != 255) {
                    const int pos = hcnt[slot] + (int)(e >> 21);
#ifndef FMHR_X_NO_ENTRY_STORE
                    if (pos < ba.cap) ba.bins[(size_t)bin * ba.cap + pos] = ent;
                    else *bin_overflow_word(ba) = 1;
#endif
                }
                const bool slow = have && slot == 255 && ent.x != kBinInvalid;
                if (__ballot(slow) != 0ull) bin_append(slow, bin, ent, ba, lane);
            };
            unsigned e0 = 0u;
            int bin0 = 0;
            BinEntry ent0 = make_entry(0u, 0.f);
            const bool have0 = (int)threadIdx.x < nt;
            if (have0) evaluate(threadIdx.x, e0, bin0, ent0);
            if (own) {
                hcnt[threadIdx.x] = gbase;                           // count -> base slot (waits for the atomic here)
                if (gbase == 0) *seg_state_word(ba, (long long)b * ba.segsPerView + hkey[threadIdx.x]) = kSegHasFragments | kSegWordsLive;
            }
            __syncthreads();
            store(have0, e0, bin0, ent0);
            for (int t0 = kTrisBlock; t0 < nt; t0 += kTrisBlock) {   // further slices of a round: more than 256 fragments per 256 faces
                const int t = t0 + threadIdx.x;
                unsigned e1 = 0u;
                int bin1 = 0;
                BinEntry ent1 = make_entry(0u, 0.f);
                if (t < nt) evaluate(t, e1, bin1, ent1);
                store(t < nt, e1, bin1, ent1);
            }
            // (the segment's mark for the second pass is the `gbase == 0` store above: the reservation that found the cursor at 0 is the
            // bin's first of this step — the second pass resets the cursor only on marked segments, so exactly one reservation per
            // segment and step sees 0, and one mark is enough; marking at the end of the round from every reservation cost the worker
            // instantiation 16 spilled VGPRs)
        }
    } else {
        for (int R = 0; R < T; R += kTaskCap) {                          // workgroup-uniform trip count (one round unless boxes are large)
            if (R > 0) __syncthreads();                                  // the previous round's list has been consumed
            if (cm != 0u && excl < R + kTaskCap && excl + cnt > R) {
                unsigned m = cm;
                int idx = excl - R;
                while (m != 0u) {
                    const int bit = __builtin_ctz(m);
                    m &= m - 1u;
                    if (idx >= 0 && idx < kTaskCap) tasks[idx] = (unsigned short)((threadIdx.x << 5) | bit);
                    idx++;
                }
            }
            __syncthreads();
            const int nt = min(kTaskCap, T - R);
            for (int t0 = wv * 64; t0 < nt; t0 += kTrisBlock) {             // wave-uniform trip count: the shuffle below needs every lane
                const int t = t0 + lane;
                int segl = -1;
                if (t < nt) {
                    const int e = tasks[t], o = e >> 5, bit = e & 31;
                    TriPark q;
                    {
                        const int4 m = pk3[o];
                        q.p0 = pk0[o]; q.p1 = pk1[o]; q.p2 = pk2[o];
                        q.f = m.x; q.rbw = __int_as_float(m.y); q.origin = m.z; q.bw = m.w & 0xff;
                    }
                    // row / column of the task's pixel in the box: bit < 32, 1 <= bw <= 32, so (bit + 0.5) / bw is at least 1 / 64 away
                    // from an integer and the 1-ulp reciprocal cannot change the truncated quotient (an integer division costs ~25 instructions)
                    float qf = (float)bit + 0.5f;
                    qf = qf * q.rbw;
                    const int r = (int)qf, c = bit - r * q.bw;
                    const int px = (q.origin & 0xffff) + c, py = (q.origin >> 16) + r;
                    float zw;
                    if (fragment_zw(q.p0, q.p1, q.p2, px, py, pg.xs, pg.xo, pg.ys, pg.yo, zw)) {
                        const unsigned long long key = ((unsigned long long)mono_key(zw) << 32) | ((unsigned)q.f << 1);
                        const int pix = py * Wp + px;                    // < 2^26
                        atomicMin(&keys_view[pix], key);
                        segl = (pix + segrem) >> 6;                      // segment index relative to the view's first segment
                    }
                }
                if (dirty != nullptr) {
                    // consecutive tasks mostly land in the same 64-pixel segment: one flag store per run
                    const int prev = wave_prev_lane(segl, -2);
                    if (segl >= 0 && (lane == 0 || prev != segl)) dirty[segbase + segl] = 1;
                }
            }
        }
    }
    // ---- large-box triangles (box > 32 pixels: none on a sub3-class mesh, all of them on a 778-vertex hand) that are NEW this step (or
    // any, without the worker workgroups): each is rasterized by its whole wave, 64 box pixels per iteration (no second launch; coarse
    // meshes took the queue above instead; the ones that were queued last step: sweep_workers)
    FMHR_WTS_END(raster);
    if (lbal == 0ull) return;                                        // wave-uniform
#ifdef FMHR_X_NO_LARGE_SWEEP                                         // (measurement builds only: what the in-wave sweep of large boxes costs)
    return;
#endif
    float4 lp0 = make_float4(0.f, 0.f, 0.f, 0.f), lp1 = lp0, lp2 = lp0;
    if (is_large) { lp0 = clipv(i0); lp1 = clipv(i1); lp2 = clipv(i2); }
    while (lbal != 0ull) {                                           // wave-uniform trip count
        const int src = __builtin_ctzll(lbal);
        lbal &= lbal - 1ull;
        if (lane == src) {
            LargePark& q = lpark[wv];
            q.s = s; q.p0 = lp0; q.p1 = lp1; q.p2 = lp2; q.f = face();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            const LargePark& q = lpark[wv];
            // (the loop of sweep_large_box, spelled out: as a call the non-binned instantiations needed 64 VGPRs + 78 spills for 60)
            const int lbw = q.s.pxb - q.s.pxa + 1, lbh = q.s.pyb - q.s.pya + 1, np = lbw * lbh;
            for (int i0 = 0; i0 < np; i0 += 64) {                    // wave-uniform trip count (bin_append is a wave-level operation)
                const int i = i0 + lane;
                int c;
                const int r = fast_divmod(min(i, np - 1), lbw, c);
                const int px = q.s.pxa + c, py = q.s.pya + r;
                const bool in = i < np && covers(q.s, px, py);
                if (BIN) {
                    float zw = 0.f;
                    const bool okf = in && fragment_zw(q.p0, q.p1, q.p2, px, py, pg.xs, pg.xo, pg.ys, pg.yo, zw);
                    const BinEntry e = make_entry((unsigned)q.f | (1u << 22) | ((unsigned)(px & 63) << 26), zw);
                    bin_append(okf, b * ba.segsPerView + py * ba.segsX + (px >> 6), e, ba, lane);
                } else if (in) {
                    emit_fragment(q.p0, q.p1, q.p2, px, py, q.f, pg.xs, pg.xo, pg.ys, pg.yo, keys_view, Wp, dirty, (long long)b * P);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                             // every lane has read the slot before the next one is parked
    }
}

// one workgroup per queued large-box triangle (grid-stride over the queue; the count lives on the device)
__global__ void __launch_bounds__(kBlock)
k_raster_large(const float* __restrict__ pos, const int2* __restrict__ snap, const int32_t* __restrict__ tri, int n, int V,
               int F, int H, int W, int Wp, unsigned long long* __restrict__ keys, const int* __restrict__ large_cnt,
               const int2* __restrict__ large_q, unsigned char* __restrict__ dirty, const float* __restrict__ verts,
               const float4* __restrict__ cmat, const PixGrid pg, const int* __restrict__ stamp)
{
    if (stamp != nullptr && (stamp[0] & 1)) large_q += (size_t)n * F;   // (the queue's half of this step's parity: k_raster_tris)
    const int cnt = *large_cnt;
    const float xs = pg.xs, ys = pg.ys, xo = pg.xo, yo = pg.yo;
    for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
        int2 bf = large_q[it];
        int b = bf.x, f = bf.y;
        int i0 = tri[f * 3], i1 = tri[f * 3 + 1], i2 = tri[f * 3 + 2];
        const int2* sv = snap + (size_t)b * V;
        TriSetup s;
        if (!setup_triangle(sv[i0], sv[i1], sv[i2], H, W, s)) continue;
        const ClipView cv = clip_view(pos, verts, cmat, b, V);       // stored positions (generic op) or recomputed (fused step)
        float4 p0 = cv.get(i0), p1 = cv.get(i1), p2 = cv.get(i2);
        pin(p0); pin(p1); pin(p2);
        int bw = s.pxb - s.pxa + 1, bh = s.pyb - s.pya + 1;
        unsigned long long* kv = keys + (size_t)b * H * Wp;
        for (int i = threadIdx.x; i < bw * bh; i += kBlock) {
            int py = s.pya + i / bw, px = s.pxa + i % bw;
            if (covers(s, px, py)) emit_fragment(p0, p1, p2, px, py, f, xs, xo, ys, yo, kv, Wp, dirty, (long long)b * H * Wp);
        }
    }
}

// per pixel: decode the winning key, recompute its fragment, write (u, v, z/w, id+1) as one 16-B store.
// Sparse mode (dirty != null, fused step): a wave owns one 64-pixel segment; a clean segment costs one scalar load; a dirty one
// is resolved, its keys are reset to "empty" (the key buffer cleans itself: no per-step 100 MB memset), and the coverage word is
// stored.  The fused step (RasterFused) also shades the covered pixels here, leaves its per-pixel word (shade_common.h) instead
// of `rast`, marks the owning (view, face) and counts the valid pixels of the segment.
// UNI: H * W is a multiple of 64, so a wave's 64-pixel segment lies in ONE view: the view index is a scalar, the base pointers
// of the view's arrays are scalar and its SH row comes through the scalar cache into SGPRs (9 VGPRs and the 64-bit address
// arithmetic of a dozen gathers less: the kernel fits 8 waves per SIMD)
// FV: the clip positions are recomputed from the vertices (fused step) instead of loaded
#ifdef FMHR_X_TS
// measurement build only (tools/analysis/resolve_timeline.py): every wave of the resolve pass leaves 100-MHz timestamps of its
// dependent steps (each taken after an s_waitcnt, which serialises what little overlap the chain has)
__device__ unsigned long long g_ts[8 * 262144];
#define FMHR_TS(k) do { if (ts_on) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tsv[k] = wall_clock64(); } } while (0)
#define FMHR_TS_FLUSH() do { if (ts_on && (threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 8; k_++) g_ts[ts_slot * 8 + k_] = tsv[k_]; } } while (0)
#else
#define FMHR_TS(k) do { } while (0)
#define FMHR_TS_FLUSH() do { } while (0)
#endif
constexpr int kResolveBlock = kBlock;
template <bool UNI, bool FV>
__global__ void __launch_bounds__(kResolveBlock) __attribute__((amdgpu_waves_per_eu((FV && !UNI) ? 5 : ((UNI && !FV) ? 8 : 7), 8)))
k_raster_resolve(const float* __restrict__ pos, const int32_t* __restrict__ tri, int n, int V, int F, int H, int W, int Wp,
                 unsigned long long* __restrict__ keys, float4* __restrict__ rast, unsigned char* __restrict__ dirty,
                 unsigned long long* __restrict__ covw, int* __restrict__ large_cnt, const RasterFused fu, const PixGrid pg)
{
    // (Workgroup i runs on XCD i % 8 and a 512-pixel row is two workgroups: with the hand in columns 256-384 of every view XCDs 1, 3,
    // 5, 7 resolve 56 % dirty waves and XCDs 0, 2, 4, 6 0.7 % (tools/analysis/resolve_timeline.py).  Rotating the blocks by a hash
    // of the group index evens that out per XCD and per CU and changes nothing — 82 vs 85 us: at four or more busy XCDs the pass
    // is bound by what lies behind the L2s; 64- / 128-thread workgroups put all dirty work on TWO XCDs: 137 / 120 us.)
    const unsigned blk = blockIdx.x;
    const long long seg_w = (long long)blk * (kResolveBlock / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long gid = (long long)blk * kResolveBlock + threadIdx.x;
    const long long P = (long long)H * Wp;
    const bool sparse = dirty != nullptr;
    const long long seg = gid >> 6;
#ifdef FMHR_X_TS
    const bool ts_on = seg < 262144;
    const long long ts_slot = seg;
    unsigned long long tsv[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    if (ts_on) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tsv[6] = ((unsigned long long)xcc << 32) | hw;
        tsv[0] = wall_clock64();
    }
#endif
    if (sparse) {
        if (gid == 0) {                                              // rearm the large-triangle queue and the caller's pairs of
            large_cnt[1] = large_cnt[0];                             // list cursors (saves memset launches); the backward's large-face
            large_cnt[0] = 0;                                        // kernel reads the copy of the count
            if (fu.rearm2 != nullptr) { fu.rearm2[0] = 0; fu.rearm2[1] = 0; }
            if (fu.rearm_aa != nullptr) { fu.rearm_aa[0] = 0; fu.rearm_aa[1] = 0; }
        }
        if (gid - (gid & 63) >= (long long)n * P) return;            // whole wave out of range
        // the segment's dirty byte through the SCALAR cache (the four waves of a workgroup and the fifteen workgroups after it
        // read the same 64-B line of flags; a wave only ever uses — and resets — its own byte, so the non-coherent scalar
        // cache cannot show it a stale value within the launch)
        typedef const unsigned __attribute__((address_space(4))) const_as4_u32;
        const unsigned dword = *(const_as4_u32*)(unsigned long long)(dirty + (seg_w & ~3ll));
        const bool is_dirty = ((dword >> (8 * (int)(seg_w & 3))) & 0xffu) != 0u;
        FMHR_TS(1);
        if (!is_dirty) {
            if ((threadIdx.x & 63) == 0) {
                covw[seg] = 0ull;
                if (fu.seg_valid != nullptr) fu.seg_valid[seg] = 0;      // (silw keeps its stale word: it is only read next to a non-zero coverage word)
            }
#ifdef FMHR_X_TS
            if (ts_on) tsv[5] = wall_clock64();
#endif
            FMHR_TS_FLUSH();
            return;                                                  // no workgroup barrier below: a clean wave leaves at once
        }
    }
    bool valid_px = false, sbits_any = false;
    const int step_tag = (fu.stamp != nullptr) ? fu.stamp[0] : 0;       // read up front: inside the branch it costs a round trip
    {
        const bool live = gid < (long long)n * P;
        unsigned long long key = live ? keys[gid] : ~0ull;
        // the caller's mask is [n,H,W]: with a row pitch the segment's row is one scalar division (a pad lane has no mask)
        long long cidx = gid;
        bool incol = true;
        if (fu.sel.idx != nullptr) cidx = caller_index_sel((unsigned)seg_w, (int)(threadIdx.x & 63), W, fu.segx, fu.sel, incol);   // (a minibatch of the capture's views: ABI 6)
        else if (Wp != W) cidx = caller_index((unsigned)seg_w, (int)(threadIdx.x & 63), W, fu.segx, incol);
        const float mk = (live && incol && fu.pix16 != nullptr) ? fu.mask[cidx] : 0.f;   // independent of the key: issued with it
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f), col = o;
        bool covered = false;
        FMHR_TS(2);
        if (key != ~0ull) {
            int rem, px;
            int b = fast_divmod(gid, (int)P, rem);
            if (UNI) b = __builtin_amdgcn_readfirstlane(b);
            const int py = fast_divmod(rem, Wp, px);
            const int f = (int)((unsigned)(key & 0xffffffffu) >> 1);
            const bool large = (key & 1ull) != 0ull;
            const int i0 = tri[f * 3], i1 = tri[f * 3 + 1], i2 = tri[f * 3 + 2];
            FMHR_TS(3);
            // clip positions: loaded (generic op), or recomputed from the three vertices and the view's combined matrix (fused step)
            constexpr bool from_verts = FV;
            float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
            F3U x0 = {0.f, 0.f, 0.f}, x1 = x0, x2 = x0;
            float cm[16];
            if (from_verts) {
                x0 = *reinterpret_cast<const F3U*>(fu.verts + (size_t)i0 * 3);
                x1 = *reinterpret_cast<const F3U*>(fu.verts + (size_t)i1 * 3);
                x2 = *reinterpret_cast<const F3U*>(fu.verts + (size_t)i2 * 3);
                if (UNI) {               // scalar view: the 16 matrix entries come through the scalar cache into SGPRs
                    const_as4_float* cp = (const_as4_float*)(unsigned long long)(fu.cmat + (size_t)b * 4);
#pragma unroll
                    for (int k = 0; k < 16; k++) cm[k] = cp[k];
                } else {
                    const float4 r0 = fu.cmat[(size_t)b * 4], r1 = fu.cmat[(size_t)b * 4 + 1], r2 = fu.cmat[(size_t)b * 4 + 2],
                                 r3 = fu.cmat[(size_t)b * 4 + 3];
                    cm[0] = r0.x; cm[1] = r0.y; cm[2] = r0.z; cm[3] = r0.w; cm[4] = r1.x; cm[5] = r1.y; cm[6] = r1.z; cm[7] = r1.w;
                    cm[8] = r2.x; cm[9] = r2.y; cm[10] = r2.z; cm[11] = r2.w; cm[12] = r3.x; cm[13] = r3.y; cm[14] = r3.z; cm[15] = r3.w;
                }
            } else {
                const float* vp = pos + (size_t)b * V * 4;
                p0 = ldg4(vp + (size_t)i0 * 4); p1 = ldg4(vp + (size_t)i1 * 4); p2 = ldg4(vp + (size_t)i2 * 4);
            }
            // fused step: the shading inputs and the face's silhouette bits ride in the same memory round trip as the vertices
            int sbits = 0;
            if (fu.pix16 != nullptr) sbits = fu.silh[(size_t)b * F + f];
            float4 q0, q1, u0, u1, w0, w1;
            float shc[9];
            const bool shade = fu.aa_init != nullptr && mk > 0.f;
            if (shade) {
                q0 = fu.attr8[(size_t)i0 * 2]; q1 = fu.attr8[(size_t)i0 * 2 + 1];
                u0 = fu.attr8[(size_t)i1 * 2]; u1 = fu.attr8[(size_t)i1 * 2 + 1];
                w0 = fu.attr8[(size_t)i2 * 2]; w1 = fu.attr8[(size_t)i2 * 2 + 1];
                if (UNI) {   // scalar loads (constant address space: the row is not written during the kernel)
                    const_as4_float* sp = (const_as4_float*)(unsigned long long)(fu.sh + (size_t)b * 9);
#pragma unroll
                    for (int k = 0; k < 9; k++) shc[k] = sp[k];
                } else {   // the view's SH row: 36 B, dword-aligned -> 2 x 16 B + 4 B requests instead of nine
                    const F4U* sp = reinterpret_cast<const F4U*>(fu.sh + (size_t)b * 9);
                    const F4U s0 = sp[0], s1 = sp[1];
                    shc[0] = s0.x; shc[1] = s0.y; shc[2] = s0.z; shc[3] = s0.w;
                    shc[4] = s1.x; shc[5] = s1.y; shc[6] = s1.z; shc[7] = s1.w;
                    shc[8] = fu.sh[(size_t)b * 9 + 8];
                }
                pin(q0); pin(q1); pin(u0); pin(u1); pin(w0); pin(w1);
            }
            pin(p0); pin(p1); pin(p2);
            FMHR_TS(4);
            if (from_verts) {            // the same clip_coord() chains as every other producer of these positions: bit-identical
                asm volatile("" : "+v"(x0.x), "+v"(x0.y), "+v"(x0.z), "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x2.x), "+v"(x2.y), "+v"(x2.z));
                p0 = make_float4(clip_coord(x0.x, x0.y, x0.z, cm[0], cm[4], cm[8], cm[12]), clip_coord(x0.x, x0.y, x0.z, cm[1], cm[5], cm[9], cm[13]),
                                 clip_coord(x0.x, x0.y, x0.z, cm[2], cm[6], cm[10], cm[14]), clip_coord(x0.x, x0.y, x0.z, cm[3], cm[7], cm[11], cm[15]));
                p1 = make_float4(clip_coord(x1.x, x1.y, x1.z, cm[0], cm[4], cm[8], cm[12]), clip_coord(x1.x, x1.y, x1.z, cm[1], cm[5], cm[9], cm[13]),
                                 clip_coord(x1.x, x1.y, x1.z, cm[2], cm[6], cm[10], cm[14]), clip_coord(x1.x, x1.y, x1.z, cm[3], cm[7], cm[11], cm[15]));
                p2 = make_float4(clip_coord(x2.x, x2.y, x2.z, cm[0], cm[4], cm[8], cm[12]), clip_coord(x2.x, x2.y, x2.z, cm[1], cm[5], cm[9], cm[13]),
                                 clip_coord(x2.x, x2.y, x2.z, cm[2], cm[6], cm[10], cm[14]), clip_coord(x2.x, x2.y, x2.z, cm[3], cm[7], cm[11], cm[15]));
            }
            Frag fr;
            if (fragment(p0, p1, p2, px, py, pg, fr)) {
                o = make_float4(fr.u, fr.v, fr.zw, (float)(f + 1));
                covered = true;
                if (fu.pix16 != nullptr) {
                    // everything the later kernels need per pixel in ONE 16-B word (shade_common.h: pix_pack); a large-box face
                    // has no candidate mask: the backward finds its pixels by scanning its box (shade.hip: face_bwd_large_items)
                    // "this (view, face) owns pixels in step <stamp>": lets the backward skip the coverage entries of hidden
                    // faces (8-bit stamp; k_raster_tris wipes all marks every 128 steps, so a mark never meets its value again)
                    fu.own8[(size_t)b * F + f] = (unsigned char)step_tag;
                    fu.pix16[gid] = make_float4(fr.u, fr.v, fr.zw, __int_as_float(pix_pack(f, mk > 0.f, large, sbits)));
                    sbits_any = sbits != 0;
                }
                if (fu.aa_init != nullptr) {
                    // forward shading of the fused step (mesh_sfs_optim.py:268-286): rgb = SH radiance x albedo where mask > 0,
                    // a = the interpolated ones attribute
                    const float w2 = 1.0f - fr.u - fr.v;
                    col.w = fr.u + fr.v + w2;
                    if (shade) {
                        PixShade s;
                        shade_eval_attr(fr.u, fr.v, q0, q1, u0, u1, w0, w1, shc, s);
                        col.x = s.rad * s.ar; col.y = s.rad * s.ag; col.z = s.rad * s.ab;
                        valid_px = true;
                    }
                }
            }
        }
        if (sparse) {
            unsigned long long bal = __ballot(covered);
            if (live) {
                if (covered && fu.pix16 == nullptr) rast[gid] = o;   // fused step: pix16 replaces it (densify rebuilds rast on demand)
                keys[gid] = ~0ull;                                   // (deletion builds: this store is worth 3 us of the pass, the next 10 us)
                if (fu.aa_init != nullptr) fu.aa_init[gid] = col;    // zeros on the uncovered lanes (blend targets start from them)
            }
            const unsigned long long vbal = __ballot(valid_px);
            // does any covered pixel of the segment belong to a face with a silhouette edge in this view?  Only such pixels can be
            // the "nearer" side of an antialias pair: the antialias scan skips segments without any in reach (k_active_segments)
            const unsigned long long sbal = __ballot(covered && sbits_any);
            if ((threadIdx.x & 63) == 0) {
                covw[seg] = bal; dirty[seg] = 0;
                // valid (mask > 0 and covered) pixels of the segment | bit 8: holds a silhouette-face pixel
                if (fu.seg_valid != nullptr) fu.seg_valid[seg] = __popcll(vbal) | (sbal != 0ull ? 0x100 : 0);
                if (fu.silw != nullptr) fu.silw[seg] = sbal;         // the antialias scan tests only pixels next to one of these
            }
        } else if (live) {
            rast[gid] = o;
        }
#ifdef FMHR_X_TS
        if (ts_on) tsv[5] = wall_clock64();
        FMHR_TS(7);                                                  // ... and once every store has been acknowledged
#endif
        FMHR_TS_FLUSH();
    }
}

// ---- binned rasterizer, second pass: one wave per 64-pixel segment = per bin (the resolve pass's shape: no workgroup barrier, an
// empty segment costs one scalar load).  The bin is staged in the wave's LDS kBinChunk entries at a time (coalesced 8-B loads, two
// in flight per lane); every entry posts ds_min_u64 on its pixel's LDS key = (monotone(z/w) << 32 | face << 8 | entry index):
// nearest fragment, ties to the lowest face index — rule R6, the order of the global atomicMin it replaces — and the pixel's lane
// reads the winner's entry back from the staged chunk.  The winner's barycentrics are evaluated here (its three vertices ride with
// the `tri` row); then the resolve pass's epilogue: shading + per-pixel word + owner mark + segment words.
typedef const int __attribute__((address_space(4))) const_as4_int;
#ifndef FMHR_BINS_BLOCK
#define FMHR_BINS_BLOCK 256
#endif
constexpr int kBinsBlock = FMHR_BINS_BLOCK;      // threads per workgroup of k_raster_bins (a wave is independent of its workgroup)
#ifndef FMHR_BINS_WAVES
#define FMHR_BINS_WAVES 8
#endif
// ---- round 5: the second pass over a COMPACTED list of the segments that hold fragments.  With one wave launched per segment 70 % of the
// waves of the metric step (85 % at 48 x 1024 x 1024) find an empty segment and leave — 137 k (670 k) launches that each take a wave slot for a
// scalar round trip; k_bins_compact (one lane per segment over the packed state words: 0.8 / 3 MB) lists the segments whose state word the
// triangle kernel marked, does the empty segments' housekeeping (words that were live last step are zeroed) and the second pass runs as
// 8 192 persistent waves over that list (LIST = true) where the launch is large (rasterize_fwd_impl); else one wave per segment as before.
constexpr int kCompactBlock = 1024;
__global__ void __launch_bounds__(kCompactBlock)
k_bins_compact(const BinArgs ba, long long nSeg, unsigned long long* __restrict__ covw, int* __restrict__ seg_valid, int2* __restrict__ list,
               int* __restrict__ list_cnt)
{
    constexpr int kWaves = kCompactBlock / 64;
    __shared__ int wcnt[kWaves];
    __shared__ int base;
    const long long seg = (long long)blockIdx.x * kCompactBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool has = false;
    int count = 0;
    if (seg < nSeg) {
        int* const stp = seg_state_word(ba, seg);
        const int state = *stp;
        has = (state & kSegHasFragments) != 0;
        if (has) {
            // the bin's fragment count, fetched and reset HERE, one lane per segment (60 k cold cursor lines in parallel instead of one
            // scalar round trip at the head of every second-pass wave): the list entry carries it
            int* const cur = ba.cursor + (size_t)seg * kCursorStride;
            count = *cur;
            *cur = 0;                                                // self-cleaning
            *stp = kSegWordsLive;
        } else if ((state & kSegWordsLive) != 0) {                  // had fragments last step, none now: its words go back to zero
            covw[seg] = 0ull;
            if (seg_valid != nullptr) seg_valid[seg] = 0;
            *stp = 0;
        }
    }
    const unsigned long long bal = __ballot(has);
    if (lane == 0) wcnt[wv] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int na = 0;
        for (int i = 0; i < kWaves; i++) na += wcnt[i];
        base = na ? atomicAdd(list_cnt, na) : 0;                    // one cursor atomic per 1 024 segments
    }
    __syncthreads();
    if (has) {
        int off = base + __popcll(bal & ((1ull << lane) - 1ull));
        for (int i = 0; i < wv; i++) off += wcnt[i];
        list[off] = make_int2((int)seg, count);
    }
}

// one segment of the second pass (wave-uniform `seg`); lkey / chunk: the wave's LDS stage.  LIST: the segment comes off the compacted list
// with its fragment count (k_bins_compact has fetched and reset the cursor and marked the words live): no state word, no cursor here
template <bool LIST>
__device__ __forceinline__ void bins_segment(const long long seg, const int count_in, const int lane, unsigned long long* const lkey, BinEntry* const chunk,
                                             const int32_t* __restrict__ tri, const int n, const int F, const int H, const int W, const int Wp,
                                             const BinArgs& ba, unsigned long long* __restrict__ covw, const RasterFused& fu, const PixGrid& pg)
{
    const long long P = (long long)H * Wp;
    const long long gid = seg * 64 + lane;
    if (seg * 64 >= (long long)n * P) return;
    int src_delta = 0, count = count_in;
    if (LIST) {
        if (fu.sel.idx != nullptr) {                                 // (ABI 6) the segment's view in the caller's arrays
            const unsigned bsel = udiv((unsigned)seg, fu.sel.segv);
            src_delta = fu.sel.src(bsel) - (int)bsel;
        }
    } else {
        // the segment's state word first (packed, hits; BinArgs): an empty segment whose words are already zero leaves here
        int* const stp = seg_state_word(ba, seg);
        int state = *(const_as4_int*)(unsigned long long)stp;
        // (ABI 6) the segment's view in the caller's arrays, requested WITH the state word: behind the waits below it was one more scalar
        // round trip in front of the mask load of every non-empty segment
        if (fu.sel.idx != nullptr) {
            const unsigned bsel = udiv((unsigned)seg, fu.sel.segv);
            src_delta = fu.sel.src(bsel) - (int)bsel;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(state), "+s"(src_delta) : : "memory");   // (the same explicit order as for the count below: this wave stores to the word)
        if ((state & kSegHasFragments) == 0) {                           // (LIST: cannot happen — the list holds marked segments only)
            if ((state & kSegWordsLive) != 0 && lane == 0) {
                covw[seg] = 0ull;
                if (fu.seg_valid != nullptr) fu.seg_valid[seg] = 0;
                *stp = 0;
            }
            return;
        }
        if (lane == 0) *stp = kSegWordsLive;
        // the bin's fragment count through the scalar cache (written by the previous kernel's atomics; only this wave resets it)
        int* const cur = ba.cursor + (size_t)seg * kCursorStride;
        count = *(const_as4_int*)(unsigned long long)cur;
        // EXPLICIT order between this scalar load and the reset store of the same word below: the load comes from the constant address
        // space (the compiler may move a store across it) and the hardware orders SMEM against VMEM no more than that.  The asm consumes
        // the loaded register behind an s_waitcnt (the load has returned) and clobbers memory (no store is hoisted above it); the order
        // no longer hangs on a branch happening to sit in between (round 3: a variant without that branch read its own zero in half of
        // the segments, a different half in every step).  tests/test_gpu_ham.py::test_bin_cursor_reset_order
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(count) : : "memory");
        if (count == 0) {
            if (lane == 0) {
                covw[seg] = 0ull;
                if (fu.seg_valid != nullptr) fu.seg_valid[seg] = 0;
            }
            return;
        }
        if (lane == 0) *cur = 0;                                     // self-cleaning (ordered behind the scalar load of the count: above)
    }
    // the caller's mask is [n,H,W]: with a row pitch the segment's row is one scalar division (a pad lane has no mask)
    long long cidx = gid;
    bool incol = true;
    if (fu.sel.idx != nullptr) {                                     // (a minibatch of the capture's views: ABI 6; common.h: caller_index_sel)
        const unsigned row = udiv((unsigned)seg, fu.segx), col0 = ((unsigned)seg - row * fu.segx.d) << 6;
        incol = (int)col0 + lane < W;
        cidx = ((long long)row + (long long)src_delta * fu.sel.H) * W + (int)col0 + lane;
    } else if (Wp != W) cidx = caller_index((unsigned)seg, lane, W, fu.segx, incol);
    const float mk = incol ? fu.mask[cidx] : 0.f;                    // independent of the bin: issued with its first loads
    // the step tag and, in the same 16-B word, the high-water mark of the fragments one segment has held since the workspace was initialised
    // (stamp[3]): what `bin_cap` has to cover — HamStage2.bin_depth_max() reads it.  ONE load; the mark is compared at the end of the segment,
    // and the atomic runs only when it moves (as a second load with its own wait up here the mark cost this pass 1.7 us)
    const int4 stamp4 = *reinterpret_cast<const int4*>(fu.stamp);
    const int step_tag = stamp4.x;
    const int cnt = min(count, ba.cap);
    const BinEntry* __restrict__ bin = ba.bins + (size_t)seg * ba.cap;
    lkey[lane] = ~0ull;
    unsigned long long best = ~0ull;
    BinEntry win = make_entry(0u, 0.f);
    for (int c0 = 0; c0 < cnt; c0 += kBinChunk) {                    // wave-uniform; one round unless > 4 fragments per pixel
        BinEntry e[kBinChunk / 64];
#pragma unroll
        for (int k = 0; k < kBinChunk / 64; k++) {
            const int i = c0 + k * 64 + lane;
            e[k] = make_entry(kBinInvalid, 0.f);
            if (i < cnt) e[k] = ld_stream(&bin[i]);                  // read once
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();                             // keys initialised / the previous chunk has been read by every lane
#pragma unroll
        for (int k = 0; k < kBinChunk / 64; k++) {
            const int idx = k * 64 + lane;
            if (e[k].x != kBinInvalid) {
                chunk[idx] = e[k];
                const unsigned long long key = ((unsigned long long)mono_key(__uint_as_float(e[k].y)) << 32) |
                                               (unsigned long long)(((e[k].x & 0x3FFFFFu) << 8) | (unsigned)idx);
                atomicMin(&lkey[e[k].x >> 26], key);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long k2 = lkey[lane];
        if (k2 != best) {                                            // the minimum moved: the winner so far is an entry of THIS chunk
            best = k2;
            win = chunk[(unsigned)k2 & (unsigned)(kBinChunk - 1)];
        }
    }
    bool covered = false, valid_px = false, sbits_any = false;
    float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
    if (best != ~0ull) {
        int rem;
        const int b = __builtin_amdgcn_readfirstlane(fast_divmod(gid, (int)P, rem));   // Wp % 64 == 0: a segment lies in one view
        const int f = (int)(win.x & 0x3FFFFFu);
        const bool large = ((win.x >> 22) & 1u) != 0u;
        const bool shade = mk > 0.f;
        // the face's silhouette bits in this view ride in the entry; a large-box face (swept by its wave, rare) has them in the byte array
        const int sbits = large ? (int)fu.silh[(size_t)b * F + f] : (int)((win.x >> 23) & 7u);
        float4 q0, q1, u0, u1, w0, w1;
        float shc[9];
        // The entry carries the depth only: the winner's (u, v) are evaluated HERE, for the ~40 % of the fragments that win their pixel
        // (round 4: 16-B entries with the triangle kernel's (u, v) -> 8-B entries: k_raster_tris 91.5 -> 86.5 us, this pass 82.8 -> 88.0 us,
        // 2 823 -> 2 846 it/s over four alternating rounds, half the bins' footprint and entry traffic).  The face's three vertices come
        // with its `tri` row, the clip positions (x, y, w: the barycentrics need no z) from the shared clip_coord() chains with the view's
        // matrix in SGPRs, and fragment_uv() is the (u, v) part of rule R5's statement sequence.  The attribute gathers are requested
        // BEHIND that arithmetic: issued with the vertex loads they kept 24 more registers live across it (94 VGPRs: 5 waves per SIMD,
        // 104 us; at 6 waves 24 spills: 143 us; requested once the clip positions stand, in front of the barycentrics: 90.6 us).
        const int i0 = tri[f * 3], i1 = tri[f * 3 + 1], i2 = tri[f * 3 + 2];
        F3U x0 = *reinterpret_cast<const F3U*>(fu.verts + (size_t)i0 * 3);
        F3U x1 = *reinterpret_cast<const F3U*>(fu.verts + (size_t)i1 * 3);
        F3U x2 = *reinterpret_cast<const F3U*>(fu.verts + (size_t)i2 * 3);
        float cm[16];
        {
            const_as4_float* cp = (const_as4_float*)(unsigned long long)(fu.cmat + (size_t)b * 4);
#pragma unroll
            for (int k = 0; k < 16; k++) cm[k] = cp[k];
        }
        asm volatile("" : "+v"(x0.x), "+v"(x0.y), "+v"(x0.z), "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x2.x), "+v"(x2.y), "+v"(x2.z));
        const float p0x = clip_coord(x0.x, x0.y, x0.z, cm[0], cm[4], cm[8], cm[12]), p0y = clip_coord(x0.x, x0.y, x0.z, cm[1], cm[5], cm[9], cm[13]),
                    p0w = clip_coord(x0.x, x0.y, x0.z, cm[3], cm[7], cm[11], cm[15]);
        const float p1x = clip_coord(x1.x, x1.y, x1.z, cm[0], cm[4], cm[8], cm[12]), p1y = clip_coord(x1.x, x1.y, x1.z, cm[1], cm[5], cm[9], cm[13]),
                    p1w = clip_coord(x1.x, x1.y, x1.z, cm[3], cm[7], cm[11], cm[15]);
        const float p2x = clip_coord(x2.x, x2.y, x2.z, cm[0], cm[4], cm[8], cm[12]), p2y = clip_coord(x2.x, x2.y, x2.z, cm[1], cm[5], cm[9], cm[13]),
                    p2w = clip_coord(x2.x, x2.y, x2.z, cm[3], cm[7], cm[11], cm[15]);
        // the pixel: the segment's row and first column are wave-uniform (one scalar division), the lane is the column inside it
        const unsigned rowg = udiv((unsigned)seg, fu.segx), col0 = ((unsigned)seg - rowg * fu.segx.d) << 6;
        const int py = (int)rowg - b * H, px = (int)col0 + lane;
        float u, v;
        fragment_uv(p0x, p0y, p0w, p1x, p1y, p1w, p2x, p2y, p2w, px, py, pg, u, v);
        const float zw = __uint_as_float(win.y);
        asm volatile("" : "+v"(u), "+v"(v));
        if (shade) {
            q0 = fu.attr8[(size_t)i0 * 2]; q1 = fu.attr8[(size_t)i0 * 2 + 1];
            u0 = fu.attr8[(size_t)i1 * 2]; u1 = fu.attr8[(size_t)i1 * 2 + 1];
            w0 = fu.attr8[(size_t)i2 * 2]; w1 = fu.attr8[(size_t)i2 * 2 + 1];
            const_as4_float* sp = (const_as4_float*)(unsigned long long)(fu.sh + (size_t)b * 9);   // the view is wave-uniform
#pragma unroll
            for (int k = 0; k < 9; k++) shc[k] = sp[k];
            pin(q0); pin(q1); pin(u0); pin(u1); pin(w0); pin(w1);
        }
        covered = true;
        fu.own8[(size_t)b * F + f] = (unsigned char)step_tag;
        st_stream(&fu.pix16[gid], make_float4(u, v, zw, __int_as_float(pix_pack(f, shade, large, sbits))));   // (read by later kernels only: common.h)
        sbits_any = sbits != 0;
        const float w2 = 1.0f - u - v;
        col.w = u + v + w2;
        if (shade) {
            PixShade s;
            shade_eval_attr(u, v, q0, q1, u0, u1, w0, w1, shc, s);
            col.x = s.rad * s.ar; col.y = s.rad * s.ag; col.z = s.rad * s.ab;
            valid_px = true;
        }
    }
    const unsigned long long bal = __ballot(covered);
    if (bal != 0ull) st_stream(&fu.aa_init[gid], col);               // zeros on the uncovered lanes (blend targets start from them)
    const unsigned long long vbal = __ballot(valid_px);
    const unsigned long long sbal = __ballot(covered && sbits_any);
    if (lane == 0) {
        covw[seg] = bal;
        if (fu.seg_valid != nullptr) fu.seg_valid[seg] = __popcll(vbal) | (sbal != 0ull ? 0x100 : 0);
        if (fu.silw != nullptr) fu.silw[seg] = sbal;
#ifndef FMHR_X_NO_DEPTH_MARK                                         // (measurement builds only: same-box A/B of the mark's cost)
        if (count > stamp4.w) atomicMax(fu.stamp + 3, count);
#endif
    }
}

#ifndef FMHR_BINS_LIST_WAVES
#define FMHR_BINS_LIST_WAVES 8
#endif
template <bool LIST>
__global__ void __launch_bounds__(kBinsBlock) __attribute__((amdgpu_waves_per_eu(LIST ? FMHR_BINS_LIST_WAVES : FMHR_BINS_WAVES, 8)))
k_raster_bins(const int32_t* __restrict__ tri, int n, int F, int H, int W, int Wp, const BinArgs ba, unsigned long long* __restrict__ covw,
              int* __restrict__ large_cnt, const RasterFused fu, const PixGrid pg)
{
    __shared__ unsigned long long lkey_all[kBinsBlock];
    __shared__ BinEntry chunk_all[(kBinsBlock / 64) * kBinChunk];
    static_assert(kBinChunk % 64 == 0 && kBinChunk <= 256, "8-bit entry index in the LDS key");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* const lkey = lkey_all + wv * 64;             // this wave's pixels
    BinEntry* const chunk = chunk_all + wv * kBinChunk;
    if (blockIdx.x == 0 && tid == 0) {                               // rearm the list cursors of the kernels after this one (as the resolve pass does)
        large_cnt[1] = large_cnt[0];
        large_cnt[0] = 0;
        if (fu.rearm2 != nullptr) { fu.rearm2[0] = 0; fu.rearm2[1] = 0; }
        if (fu.rearm_aa != nullptr) { fu.rearm_aa[0] = 0; fu.rearm_aa[1] = 0; }
        int* const ov = bin_overflow_word(ba);                       // the triangle kernel's overflow word -> the sticky flag the host reads
        if (*ov != 0) { *ov = 0; if (fu.bin_flag != nullptr) *fu.bin_flag = 1; }
    }
    if (!LIST) {                                                     // one wave per segment
        bins_segment<false>((long long)blockIdx.x * (kBinsBlock / 64) + wv, 0, lane, lkey, chunk, tri, n, F, H, W, Wp, ba, covw, fu, pg);
        return;
    }
    // persistent waves over the step's list of segments with fragments (k_bins_compact)
    const int list_n = *(const_as4_int*)(unsigned long long)fu.bin_list_cnt;
    for (int li = (int)(blockIdx.x * (kBinsBlock / 64)) + wv; li < list_n; li += (int)gridDim.x * (kBinsBlock / 64)) {
        const_as4_int* const ent = (const_as4_int*)(unsigned long long)(fu.bin_list + li);      // (segment, fragment count): one 8-B scalar load
        const int eseg = ent[0], ecnt = ent[1];
        bins_segment<true>((long long)eseg, ecnt, lane, lkey, chunk, tri, n, F, H, W, Wp, ba, covw, fu, pg);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // the wave's LDS stage is reused by its next segment
        __builtin_amdgcn_wave_barrier();
    }
}

// inspection helper (fmhr_ham_densify): the shaded rgba image before antialias, which the step itself never stores — every covered
// pixel re-shaded from its per-pixel word exactly as the resolve pass did (same translation unit: same arithmetic), zeros elsewhere
__global__ void __launch_bounds__(kBlock)
k_reshade_dense(const unsigned long long* __restrict__ covw, const float4* __restrict__ pix16, const int32_t* __restrict__ tri,
                const float4* __restrict__ attr8, const float* __restrict__ sh, long long N, long long P, float4* __restrict__ shaded)
{
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= N) return;
    float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((covw[gid >> 6] >> (gid & 63)) & 1ull) {
        int rem;
        const int b = fast_divmod(gid, (int)P, rem);
        col = shade_word(pix16[gid], tri, attr8, sh + (size_t)b * 9);
    }
    shaded[gid] = col;
}

int reshade_dense_impl(const unsigned long long* covw, const float* pix16, const int32_t* tri, const float* attr8, const float* sh,
                       int n, int H, int W, float* shaded, hipStream_t st)
{
    const long long N = (long long)n * H * W;
    if (N == 0) return FMHR_OK;
    k_reshade_dense<<<cdiv(N, kBlock), kBlock, 0, st>>>(covw, (const float4*)pix16, tri, (const float4*)attr8, sh, N, (long long)H * W,
                                                        (float4*)shaded);
    FMHR_CHECK_LAUNCH("k_reshade_dense");
    return FMHR_OK;
}

// ------------------------------------------------------------------------------------------------ rasterize backward
// The nine float atomics of a pixel (x, y, w of its triangle's three clip positions) go to THREE 16-B words: staged per wave in LDS and
// issued lane = (pixel, vertex, component), so that a vertex's three components are one cache-line request (k_interp_bwd)
__global__ void __launch_bounds__(kBlock)
k_raster_bwd(const float* __restrict__ pos, const int32_t* __restrict__ tri, const float4* __restrict__ rast,
             const float4* __restrict__ dy, int n, int V, int F, int H, int W, float* __restrict__ gpos, const PixGrid pg)
{
    __shared__ float st_val[kBlock * 9];
    __shared__ unsigned st_vtx[kBlock * 3];                          // row (view * V + vertex) of gpos: < 2^32 (checked by the launcher)
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long P = (long long)H * W;
    bool on = gid < (long long)n * P;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), g = r;
    int t = -1;
    if (on) { r = rast[gid]; t = (int)r.w - 1; on = t >= 0 && t < F; }
    if (on) { g = dy[gid]; on = !(g.x == 0.f && g.y == 0.f); }
    float val[9];
    unsigned vtx[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 9; k++) val[k] = 0.f;
    if (on) {
        int rem, px;
        const int b = fast_divmod(gid, (int)P, rem);
        const int py = fast_divmod(rem, W, px);
        int i0 = tri[t * 3], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
        const float* vp = pos + (size_t)b * V * 4;
        const float4 p0 = ldg4(vp + (size_t)i0 * 4), p1 = ldg4(vp + (size_t)i1 * 4), p2 = ldg4(vp + (size_t)i2 * 4);
        // rule R5's values: this gradient in double from the fp32 positions (common.h: r5_bwd_wide)
        float qx[3], qy[3], qw[3];
        r5_bwd_wide(p0.x, p0.y, p0.w, p1.x, p1.y, p1.w, p2.x, p2.y, p2.w, px, py, pg, g.x, g.y, qx, qy, qw);
        val[0] = qx[0]; val[1] = qy[0]; val[2] = qw[0];
        val[3] = qx[1]; val[4] = qy[1]; val[5] = qw[1];
        val[6] = qx[2]; val[7] = qy[2]; val[8] = qw[2];
        vtx[0] = (unsigned)((long long)b * V + i0); vtx[1] = (unsigned)((long long)b * V + i1); vtx[2] = (unsigned)((long long)b * V + i2);
    }
    wave_add_3x3<4, 3>(gpos, on, val, vtx, st_val + (threadIdx.x & ~63) * 9, st_vtx + (threadIdx.x & ~63) * 3);
}

// ------------------------------------------------------------------------------------------------ interpolate
__global__ void __launch_bounds__(kBlock)
k_interp_fwd(const float* __restrict__ attr, const float4* __restrict__ rast, const int32_t* __restrict__ tri,
             int n, int an, int V, int F, long long P, int C, float* __restrict__ out)
{
    long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (gid >= (long long)n * P) return;
    float4 r = rast[gid];
    int t = (int)r.w - 1;
    float* o = out + gid * C;
    if (t < 0 || t >= F) {
        for (int c = 0; c < C; c++) o[c] = 0.f;
        return;
    }
    int rem_;
    const int b = fast_divmod(gid, (int)P, rem_);
    const float* A = attr + (an == 1 ? 0 : (size_t)b * V * C);
    const float* a0 = A + (size_t)tri[t * 3] * C;
    const float* a1 = A + (size_t)tri[t * 3 + 1] * C;
    const float* a2 = A + (size_t)tri[t * 3 + 2] * C;
    float w2 = 1.0f - r.x - r.y;
    for (int c = 0; c < C; c++) o[c] = r.x * a0[c] + r.y * a1[c] + w2 * a2[c];
}

// Gradient of interpolate.  The attribute gradient is C float atomics per corner of every covered pixel — and a posted atomic costs per
// distinct cache line per INSTRUCTION, while the C channels of one vertex are contiguous.  Issued lane = pixel, one
// instruction per (channel, corner), the 48-view metric batch with C = 7 was 50 M line requests = 2.2 ms.  So the wave's covered pixels
// are compacted and the (pixel, corner, channel) entries enumerated ACROSS the lanes: the C lanes of one corner hit one or two lines
// together.  The pixel's own data stays in its lane and is fetched with lane shuffles.
__global__ void __launch_bounds__(kBlock)
k_interp_bwd(const float* __restrict__ attr, const float4* __restrict__ rast, const int32_t* __restrict__ tri,
             const float* __restrict__ dy, int n, int an, int V, int F, long long P, int C,
             float* __restrict__ gattr, float4* __restrict__ grast)
{
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const bool live = gid < (long long)n * P;
    const int lane = threadIdx.x & 63;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) r = rast[gid];
    const int t = (int)r.w - 1;
    const bool cov = live && t >= 0 && t < F;
    int i0 = 0, i1 = 0, i2 = 0;
    long long off = 0;
    if (cov) {
        int rem_;
        const int b = fast_divmod(gid, (int)P, rem_);
        off = (an == 1) ? 0 : (long long)b * V * C;
        i0 = tri[t * 3]; i1 = tri[t * 3 + 1]; i2 = tri[t * 3 + 2];
        // d loss / d (u, v): sum_c dy_c (a0_c - a2_c), sum_c dy_c (a1_c - a2_c)
        const float* A = attr + off;
        const float* d = dy + gid * C;
        float gu = 0.f, gv = 0.f;
        for (int c = 0; c < C; c++) {
            const float dc = d[c];
            const float a2 = A[(size_t)i2 * C + c];
            gu += dc * (A[(size_t)i0 * C + c] - a2);
            gv += dc * (A[(size_t)i1 * C + c] - a2);
        }
        grast[gid] = make_float4(gu, gv, 0.f, 0.f);
    } else if (live) {
        grast[gid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ---- attribute gradient: entries (covered pixel j of the wave, corner k, channel c), entry e = (j * 3 + k) * C + c
    const unsigned long long bal = __ballot(cov);
    const int ncov = __popcll(bal);
    if (ncov == 0) return;                                           // wave-uniform
    // lane j < ncov learns which lane holds the j-th covered pixel
    const int myrank = __popcll(bal & ((1ull << lane) - 1ull));
    __shared__ unsigned char src_lane[kBlock];
    unsigned char* const sl = src_lane + (threadIdx.x & ~63);
    if (cov) sl[myrank] = (unsigned char)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int per_pix = 3 * C, total = ncov * per_pix;
    const float rcp_pp = __builtin_amdgcn_rcpf((float)per_pix), rcp_c = __builtin_amdgcn_rcpf((float)C);
    const int gid_lo = (int)(gid & 0xffffffffll), gid_hi = (int)(gid >> 32);
    const int off_lo = (int)(off & 0xffffffffll), off_hi = (int)(off >> 32);
    for (int e0 = 0; e0 < total; e0 += 64) {                         // wave-uniform trip count
        const int e = e0 + lane;
        const bool on = e < total;
        int j = (int)(((float)e + 0.5f) * rcp_pp);                   // e / per_pix (e < 64 * 3 * C: exact for C < 2^14 after the fix-up)
        if (j * per_pix > e) j--;
        if ((j + 1) * per_pix <= e) j++;
        const int rest = e - j * per_pix;
        int k = (int)(((float)rest + 0.5f) * rcp_c);
        if (k * C > rest) k--;
        if ((k + 1) * C <= rest) k++;
        const int c = rest - k * C;
        const int src = on ? (int)sl[j] : 0;
        // the pixel's data, from the lane that holds it
        const float u = __shfl(r.x, src, 64), v = __shfl(r.y, src, 64);
        const int v0 = __shfl(i0, src, 64), v1 = __shfl(i1, src, 64), v2 = __shfl(i2, src, 64);
        const long long pg = ((long long)__shfl(gid_hi, src, 64) << 32) | (unsigned)__shfl(gid_lo, src, 64);
        const long long po = ((long long)__shfl(off_hi, src, 64) << 32) | (unsigned)__shfl(off_lo, src, 64);
        if (on) {
            const float dc = dy[pg * C + c];
            if (dc != 0.f) {
                const float w = (k == 0) ? u : ((k == 1) ? v : 1.0f - u - v);
                const int vi = (k == 0) ? v0 : ((k == 1) ? v1 : v2);
                atomicAdd(gattr + po + (long long)vi * C + c, w * dc);
            }
        }
    }
}

}  // namespace fmhr

using namespace fmhr;

#ifdef FMHR_X_TS
extern "C" int fmhr_debug_ts(void* out, size_t bytes)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(fmhr::g_ts), bytes < sizeof(fmhr::g_ts) ? bytes : sizeof(fmhr::g_ts));
    return e == hipSuccess ? 0 : 1;
}
#endif

namespace fmhr {
// [keys: n*H*W u64][large-triangle counter: 16 B][large-triangle queue: 2 x n*F int2, the half in use = the step's parity]
size_t raster_keys_bytes(int n, int H, int W) { return (size_t)n * H * W * 8; }
size_t raster_core_ws_bytes(int n, int F, int H, int W) { return raster_keys_bytes(n, H, W) + 16 + 2 * (size_t)n * F * 8; }   // (two queue halves)
}  // namespace fmhr

extern "C" size_t fmhr_rasterize_workspace_bytes(int n, int V, int F, int H, int W)
{
    return fmhr::raster_core_ws_bytes(n, F, H, W) + (size_t)n * V * 8;      // + snapped vertices [n,V] int2
}

namespace fmhr {
int snap_vertices_impl(const float* pos, int n, int V, int H, int W, int2* snap, hipStream_t st)
{
    long long N = (long long)n * V;
    if (N == 0) return FMHR_OK;
    k_snap_vertices<<<cdiv(N, kBlock), kBlock, 0, st>>>((const float4*)pos, N, H, W, snap);
    FMHR_CHECK_LAUNCH("k_snap_vertices");
    return FMHR_OK;
}

// One-off initialisation of the self-cleaning sparse state: keys all "empty", no dirty segment.
int raster_sparse_init(void* ws, unsigned char* dirty, int n, int H, int W, hipStream_t st)
{
    size_t nP = (size_t)n * H * W;
    const size_t kb = raster_keys_bytes(n, H, W);
    hipError_t e = hipMemsetAsync(ws, 0xFF, kb, st);
    if (e != hipSuccess) return hip_fail(e, "init keys");
    e = hipMemsetAsync(dirty, 0, (nP + 63) / 64, st);
    if (e != hipSuccess) return hip_fail(e, "init dirty");
    e = hipMemsetAsync((char*)ws + kb, 0, 16, st);                   // large-triangle queue counter
    if (e != hipSuccess) return hip_fail(e, "init large_cnt");
    return FMHR_OK;
}

// snap [n,V] int2 must already hold the snapped vertices; `fused` may be null.  dirty/covw non-null = sparse mode (the key
// buffer must have been initialised by raster_sparse_init and is left clean again).
int rasterize_fwd_impl(const float* pos, const int2* snap, const int32_t* tri, const int32_t* opp, unsigned char* silh, int n, int V,
                       int F, int H, int W, int Wp, float* rast, void* ws, unsigned char* dirty, unsigned long long* covw,
                       const RasterFused* fused, hipStream_t st)
{
    // Wp = row pitch of keys / rast / every fused per-pixel buffer (== W for the stand-alone op; common.h: default_row_pitch)
    size_t nP = (size_t)n * H * Wp;
    const size_t kb = raster_keys_bytes(n, H, Wp);
    unsigned long long* keys = (unsigned long long*)ws;
    int* large_cnt = (int*)((char*)ws + kb);
    int2* large_q = (int2*)((char*)ws + kb + 16);
    if (nP == 0) return FMHR_OK;
    FMHR_CHECK_ARG(n <= 65535, "rasterize: at most 65535 views per call (the view is a grid dimension), got %d", n);
    // large-box triangles: through the queue + k_raster_large when the mesh is coarse for the resolution (>= 8 pixels per face at
    // 50 % coverage: most boxes exceed 64 pixels), else swept by their wave inside k_raster_tris (no second launch)
    const bool use_queue = raster_use_queue(H, W, F);
    RasterFused fu;
    memset(&fu, 0, sizeof(fu));
    if (fused != nullptr) fu = *fused;
    {
        ProfScope ps(FMHR_SLOT_RASTER_CLEAR, st);
        hipError_t e;
        if (dirty == nullptr) {
            e = hipMemsetAsync(keys, 0xFF, kb, st);
            if (e != hipSuccess) return hip_fail(e, "memset keys");
            e = hipMemsetAsync(large_cnt, 0, 16, st);                // sparse mode: the resolve pass re-zeroes the counter itself
            if (e != hipSuccess) return hip_fail(e, "memset large_cnt");
        }
    }
    // the fused step always queues its large-box (view, face)s (the backward's wave-per-face workers read the queue); the generic op
    // only when it hands them to k_raster_large
    const bool fused_step = fused != nullptr && fu.cov16 != nullptr;
    int2* lq = (use_queue || fused_step) ? large_q : nullptr;
    const int defer = use_queue ? 1 : 0;
    const PixGrid pg = make_pix_grid(H, W);
    // binned form (fused step, micro-triangle mesh, W % 64 == 0): per-segment fragment bins + k_raster_bins instead of the
    // global depth keys + k_raster_resolve.  FMHR_RASTER_BINNED=0 / 1 overrides the default (same-library A/B: tools/abn.sh)
    const bool binned = fu.bins != nullptr && fu.verts != nullptr && dirty != nullptr && fu.pix16 != nullptr && fu.aa_init != nullptr &&
                        raster_binned_ok(F, H, W, Wp) && raster_binned_wanted(kBinnedDefault);
    BinArgs ba;
    memset(&ba, 0, sizeof(ba));
    // (a workspace with bins whose step is sent down the atomic-key form — FMHR_RASTER_BINNED=0, same-library A/B — leaves segment words
    // behind that the state words know nothing of: every word counts as live again)
    if (!binned && fu.bin_cursor != nullptr && raster_binned_ok(F, H, W, Wp)) {
        const size_t tiles = raster_bin_count(n, H, Wp);
        hipError_t e = hipMemsetAsync(fu.bin_cursor - tiles, 0x02, tiles * sizeof(int), st);
        if (e != hipSuccess) return hip_fail(e, "bin state");
    }
    if (binned) {
        ba.bins = fu.bins; ba.cursor = fu.bin_cursor; ba.cap = fu.bin_cap; ba.tiles = (int)raster_bin_count(n, H, Wp);
        ba.segsX = Wp / 64; ba.segsPerView = (Wp / 64) * H;
    }
    if ((long long)n * F > 0) {
        ProfScope ps(FMHR_SLOT_RASTER_TRIS, st);
        // forward workers for last step's large-box queue: the fused step on a micro-triangle mesh (FMHR_SWEEP_WORKERS=0 sweeps every
        // large box in place, as round 3 did: same-library A/B)
        const char* swe = getenv("FMHR_SWEEP_WORKERS");
        const int sw = (binned && fused_step && !use_queue && lq != nullptr && fu.stamp != nullptr && !(swe && atoi(swe) == 0)) ? kSweepWorkers : 0;
        const dim3 tgrid(cdiv(F, kTrisBlock) + sw, n);               // view = blockIdx.y
        if (binned && sw > 0)
            k_raster_tris<true, true, true><<<tgrid, kTrisBlock, 0, st>>>(pos, snap, tri, opp, silh, n, V, F, H, W, Wp, keys, dirty, fu.cov16, fu.stamp, fu.verts, fu.cmat, large_cnt, lq, pg, fu.own8, ba, defer, sw);
        else if (binned)
            k_raster_tris<true, true><<<tgrid, kTrisBlock, 0, st>>>(pos, snap, tri, opp, silh, n, V, F, H, W, Wp, keys, dirty, fu.cov16, fu.stamp, fu.verts, fu.cmat, large_cnt, lq, pg, fu.own8, ba, defer, sw);
        else if (fu.verts != nullptr)
            k_raster_tris<true, false><<<tgrid, kTrisBlock, 0, st>>>(pos, snap, tri, opp, silh, n, V, F, H, W, Wp, keys, dirty, fu.cov16, fu.stamp, fu.verts, fu.cmat, large_cnt, lq, pg, fu.own8, ba, defer, sw);
        else
            k_raster_tris<false, false><<<tgrid, kTrisBlock, 0, st>>>(pos, snap, tri, opp, silh, n, V, F, H, W, Wp, keys, dirty, fu.cov16, fu.stamp, nullptr, nullptr, large_cnt, lq, pg, fu.stamp != nullptr ? fu.own8 : nullptr, ba, defer, 0);
        FMHR_CHECK_LAUNCH("k_raster_tris");
    }
    if (binned) {
        ProfScope ps(FMHR_SLOT_RASTER_RESOLVE, st);
        // the compacted list pays where the empty segments dominate the launch: same-box A/B (k_raster_bins + k_bins_compact vs
        // k_raster_bins<false>): 48 x 1024 x 1024 (786 k segments, 85 % empty) 312 -> 244 us, C4 1 040 -> 1 120 it/s; the metric (197 k
        // segments, 70 % empty) 69 -> 78 + 6 us — 8 192 persistent waves with ~7 segments each end on their slowest wave, while one wave
        // per segment is scheduled dynamically by the hardware.  FMHR_BINS_LIST=0 / 1 forces either form (same-library A/B)
        const char* ble = getenv("FMHR_BINS_LIST");
        const bool use_list = fu.bin_list != nullptr && fu.bin_list_cnt != nullptr && (ble ? atoi(ble) != 0 : ba.tiles >= 400000);
        if (use_list) {
            const long long nSeg = (long long)ba.tiles;
            k_bins_compact<<<cdiv(nSeg, kCompactBlock), kCompactBlock, 0, st>>>(ba, nSeg, covw, fu.seg_valid, (int2*)fu.bin_list, fu.bin_list_cnt);
            FMHR_CHECK_LAUNCH("k_bins_compact");
            // (FMHR_BINS_LIST=2: one wave per LIST ENTRY, as many waves as segments — a wave past the list's end leaves after one scalar load)
            const unsigned lgrid = (ble && atoi(ble) == 2) ? (unsigned)cdiv((long long)nP, kBinsBlock) : 2048u;
            k_raster_bins<true><<<lgrid, kBinsBlock, 0, st>>>(tri, n, F, H, W, Wp, ba, covw, large_cnt, fu, pg);
        } else {
            k_raster_bins<false><<<cdiv((long long)nP, kBinsBlock), kBinsBlock, 0, st>>>(tri, n, F, H, W, Wp, ba, covw, large_cnt, fu, pg);
        }
        FMHR_CHECK_LAUNCH("k_raster_bins");
        return FMHR_OK;
    }
    {
        ProfScope ps(FMHR_SLOT_RASTER_LARGE, st);          // empty on a micro-triangle mesh
        if (use_queue && (long long)n * F > 0) {
            k_raster_large<<<1024, kBlock, 0, st>>>(pos, snap, tri, n, V, F, H, W, Wp, keys, large_cnt, large_q, dirty, fu.verts, fu.cmat, pg, fu.stamp);
            FMHR_CHECK_LAUNCH("k_raster_large");
        }
    }
    ProfScope ps(FMHR_SLOT_RASTER_RESOLVE, st);
    const dim3 rgrid(cdiv((long long)nP, kResolveBlock));
    if (fu.verts != nullptr && dirty != nullptr && (((long long)H * Wp) & 63) == 0)
        k_raster_resolve<true, true><<<rgrid, kResolveBlock, 0, st>>>(pos, tri, n, V, F, H, W, Wp, keys, (float4*)rast, dirty, covw, large_cnt, fu, pg);
    else if (fu.verts != nullptr)
        k_raster_resolve<false, true><<<rgrid, kResolveBlock, 0, st>>>(pos, tri, n, V, F, H, W, Wp, keys, (float4*)rast, dirty, covw, large_cnt, fu, pg);
    else
        k_raster_resolve<false, false><<<rgrid, kResolveBlock, 0, st>>>(pos, tri, n, V, F, H, W, Wp, keys, (float4*)rast, dirty, covw, large_cnt, fu, pg);
    FMHR_CHECK_LAUNCH("k_raster_resolve");
    return FMHR_OK;
}
}  // namespace fmhr

extern "C" int fmhr_rasterize_fwd(const float* pos, const int32_t* tri, int n, int V, int F, int H, int W, float* rast,
                                  void* ws, size_t ws_bytes, void* stream)
{
    FMHR_CHECK_ARG(n >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0, "rasterize_fwd: bad dims n=%d V=%d F=%d H=%d W=%d", n, V, F, H, W);
    FMHR_CHECK_ARG(H <= 8192 && W <= 8192, "rasterize_fwd: resolution above 8192 not supported");
    FMHR_CHECK_ARG(F < (1 << 24), "rasterize_fwd: F must be < 2^24 (triangle id is stored as fp32)");
    FMHR_CHECK_ARG(V < (1 << 27), "rasterize_fwd: V must be < 2^27 (32-bit byte offsets into the per-vertex tables: common.h at32)");
    FMHR_CHECK_ARG(n == 0 || (rast && ws && (pos || V == 0) && (tri || F == 0)), "rasterize_fwd: null pointer");
    FMHR_CHECK_ARG(ws_bytes >= fmhr_rasterize_workspace_bytes(n, V, F, H, W), "rasterize_fwd: workspace too small");
    int2* snap = (int2*)((char*)ws + raster_core_ws_bytes(n, F, H, W));
    int rc = snap_vertices_impl(pos, n, V, H, W, snap, (hipStream_t)stream);
    if (rc) return rc;
    return rasterize_fwd_impl(pos, snap, tri, nullptr, nullptr, n, V, F, H, W, W, rast, ws, nullptr, nullptr, nullptr,
                              (hipStream_t)stream);
}

extern "C" int fmhr_rasterize_bwd(const float* pos, const int32_t* tri, const float* rast, const float* dy, int n, int V,
                                  int F, int H, int W, float* gpos, void* stream)
{
    FMHR_CHECK_ARG(n >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0, "rasterize_bwd: bad dims");
    size_t nP = (size_t)n * H * W;
    if (nP == 0 || F == 0) return FMHR_OK;
    FMHR_CHECK_ARG(pos && tri && rast && dy && gpos, "rasterize_bwd: null pointer");
    FMHR_CHECK_ARG((long long)n * V < (1ll << 32), "rasterize_bwd: n * V must be below 2^32");
    k_raster_bwd<<<cdiv((long long)nP, kBlock), kBlock, 0, (hipStream_t)stream>>>(pos, tri, (const float4*)rast,
                                                                                  (const float4*)dy, n, V, F, H, W, gpos, make_pix_grid(H, W));
    FMHR_CHECK_LAUNCH("k_raster_bwd");
    return FMHR_OK;
}

extern "C" int fmhr_interpolate_fwd(const float* attr, const float* rast, const int32_t* tri, int n, int an, int V, int F,
                                    int H, int W, int C, float* out, void* stream)
{
    FMHR_CHECK_ARG(n >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0 && C > 0, "interpolate_fwd: bad dims");
    FMHR_CHECK_ARG(an == 1 || an == n, "interpolate_fwd: attr batch must be 1 or n (got %d, n=%d)", an, n);
    size_t nP = (size_t)n * H * W;
    if (nP == 0) return FMHR_OK;
    FMHR_CHECK_ARG(rast && out && (attr || V == 0) && (tri || F == 0), "interpolate_fwd: null pointer");
    k_interp_fwd<<<cdiv((long long)nP, kBlock), kBlock, 0, (hipStream_t)stream>>>(attr, (const float4*)rast, tri, n, an, V, F,
                                                                                  (long long)H * W, C, out);
    FMHR_CHECK_LAUNCH("k_interp_fwd");
    return FMHR_OK;
}

extern "C" int fmhr_interpolate_bwd(const float* attr, const float* rast, const int32_t* tri, const float* dy, int n,
                                    int an, int V, int F, int H, int W, int C, float* gattr, float* grast, void* stream)
{
    FMHR_CHECK_ARG(n >= 0 && V >= 0 && F >= 0 && H > 0 && W > 0 && C > 0, "interpolate_bwd: bad dims");
    FMHR_CHECK_ARG(an == 1 || an == n, "interpolate_bwd: attr batch must be 1 or n");
    size_t nP = (size_t)n * H * W;
    if (nP == 0) return FMHR_OK;
    FMHR_CHECK_ARG(rast && dy && grast && (attr || V == 0) && (gattr || V == 0) && (tri || F == 0), "interpolate_bwd: null pointer");
    k_interp_bwd<<<cdiv((long long)nP, kBlock), kBlock, 0, (hipStream_t)stream>>>(attr, (const float4*)rast, tri, dy, n, an, V,
                                                                                  F, (long long)H * W, C, gattr, (float4*)grast);
    FMHR_CHECK_LAUNCH("k_interp_bwd");
    return FMHR_OK;
}

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

#define SFX(name) name##_f64
#define REAL double
#define RINT rint
#define FABS fabs
#define MAXREAL DBL_MAX
#include "raster_oracle_impl.h"

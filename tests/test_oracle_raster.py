"""Pins the CPU restatement of rasterize / interpolate / antialias (oracle/raster_oracle.c).  Real nvdiffrast is absent
(parity unpinned), so the pins are: written rules, invariants, and fp64 central differences."""
import numpy as np
import pytest
import torch


def ortho_pos(xy, z=0.0, w=1.0, dtype=torch.float64):
    """clip-space positions from NDC xy (w=1)."""
    xy = torch.as_tensor(xy, dtype=dtype)
    n = xy.shape[0]
    return torch.cat([xy * w, torch.full((n, 1), z * w, dtype=dtype), torch.full((n, 1), w, dtype=dtype)], 1)[None]


def px2ndc(p, size):
    return p / size * 2 - 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_fill_rule_watertight_quad(oracle, dtype):
    """A quad whose diagonal passes exactly through pixel centres: every pixel covered exactly once (top-left rule),
    and the quad's outer edges on pixel-centre lines follow the owner rule (no double cover with a mirrored neighbour)."""
    H = W = 16
    # corners at pixel centres (2.5,2.5) .. (12.5,12.5): edges and diagonal all pass through centres
    c = [[2.5, 2.5], [12.5, 2.5], [12.5, 12.5], [2.5, 12.5], [22.5, 2.5], [22.5, 12.5]]
    pos = ortho_pos([[px2ndc(x, W), px2ndc(y, H)] for x, y in c], dtype=dtype)
    tri = torch.tensor([[0, 1, 2], [0, 2, 3], [1, 4, 5], [1, 5, 2]], dtype=torch.int32)
    rast, _ = oracle.rasterize(None, pos, tri, (H, W))
    ids = rast[0, :, :, 3].numpy().astype(int)
    # count coverage by brute force per triangle: render each alone and sum
    total = np.zeros((H, W), int)
    for f in range(4):
        r1, _ = oracle.rasterize(None, pos, tri[f:f + 1], (H, W))
        total += (r1[0, :, :, 3].numpy() > 0)
    assert total.max() == 1, "a pixel centre on a shared edge must be owned by exactly one triangle"
    # interior of [2.5,12.5]^2: pixels 3..11 fully inside -> covered; the closed region has 11x11 centres incl. borders,
    # exactly the centres owned by the rule are covered and the shared edge x=12.5 is owned once
    assert (ids[3:12, 3:12] > 0).all()
    cov = (ids > 0)
    assert cov.sum() == total.sum()
    # rows 2..12, cols 2..15 region: left quad + right quad (cols up to 15 since x=22.5 is off-screen)
    assert cov[3:12, 2:16].sum() in (9 * 13, 9 * 14)      # left edge x=2.5 owned or not, consistently for all rows
    col2 = cov[3:12, 2]
    assert col2.all() or (~col2).all()
    row2, row12 = cov[2, 3:12], cov[12, 3:12]
    assert (row2.all() and not row12.any()) or (row12.all() and not row2.any()), "top/bottom ownership must be opposite"


def test_barycentrics_and_interpolate_affine(oracle):
    H, W = 32, 48
    rng = np.random.default_rng(1)
    xy = rng.uniform(-0.9, 0.9, (3, 2))
    pos = ortho_pos(xy)
    tri = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    rast, db = oracle.rasterize(None, pos, tri, (H, W))
    assert db.shape == rast.shape
    cov = rast[0, :, :, 3] > 0
    assert cov.sum() > 20
    u, v = rast[0, :, :, 0], rast[0, :, :, 1]
    assert (u[cov] >= 0).all() and (v[cov] >= 0).all() and (u[cov] <= 1).all()
    assert (rast[0][~cov] == 0).all()
    attr = torch.cat([pos[0, :, :2], torch.ones(3, 1, dtype=torch.float64)], 1)[None]
    out, _ = oracle.interpolate(attr, rast, tri)
    ys, xs = torch.where(cov)
    # strict interior (no clamp): interpolated NDC xy reproduces the pixel centre
    inner = (u[cov] > 1e-3) & (v[cov] > 1e-3) & (u[cov] + v[cov] < 1 - 1e-3)
    fx = (xs.double() + 0.5) / W * 2 - 1
    fy = (ys.double() + 0.5) / H * 2 - 1
    assert (out[0][cov][:, 0] - fx)[inner].abs().max() < 1e-12
    assert (out[0][cov][:, 1] - fy)[inner].abs().max() < 1e-12
    ones = out[0, :, :, 2]
    assert set(np.unique(ones.numpy().round(12))) <= {0.0, 1.0}
    # area check: covered pixels ~ triangle area in px^2
    e1, e2 = xy[1] - xy[0], xy[2] - xy[0]
    a = 0.5 * abs(e1[0] * e2[1] - e1[1] * e2[0]) * (W / 2) * (H / 2)
    assert abs(int(cov.sum()) - a) < 0.15 * a + 8


def test_depth_nearest_and_tie_lowest_index(oracle):
    H = W = 8
    sq = [[-1.5, -1.5], [1.5, -1.5], [-1.5, 1.5]]
    far = ortho_pos(sq, z=0.5)[0]
    near = ortho_pos(sq, z=-0.5)[0]
    pos = torch.cat([far, near, near])[None]
    tri = torch.tensor([[0, 1, 2], [3, 4, 5], [6, 7, 8]], dtype=torch.int32)
    rast, _ = oracle.rasterize(None, pos, tri, (H, W))
    cov = rast[0, :, :, 3] > 0
    assert cov.sum() > 10
    assert (rast[0, :, :, 3][cov] == 2).all(), "nearest z/w wins, equal depth keeps the lower triangle index"
    assert torch.allclose(rast[0, :, :, 2][cov], torch.tensor(-0.5, dtype=torch.float64))
    # near/far reject: a vertex behind the near plane (z < -w) drops the whole triangle
    bad = pos.clone()
    bad[0, 3, 2] = -2.0
    r2, _ = oracle.rasterize(None, bad, tri, (H, W))
    assert (r2[0, :, :, 3][cov] == 3).all()


def test_empty_and_ragged_inputs(oracle):
    pos = ortho_pos([[0, 0], [0.5, 0], [0, 0.5]])
    tri0 = torch.zeros(0, 3, dtype=torch.int32)
    rast, _ = oracle.rasterize(None, pos, tri0, (5, 7))
    assert rast.shape == (1, 5, 7, 4) and (rast == 0).all()
    out, _ = oracle.interpolate(torch.ones(1, 3, 2, dtype=torch.float64), rast, tri0)
    assert (out == 0).all()
    # off-screen triangle and degenerate (zero-area) triangle produce nothing; W not a multiple of anything
    pos2 = ortho_pos([[3, 3], [4, 3], [3, 4], [0, 0], [0.5, 0.5], [-0.5, -0.5]])
    tri = torch.tensor([[0, 1, 2], [3, 4, 5]], dtype=torch.int32)
    rast, _ = oracle.rasterize(None, pos2, tri, (9, 11))
    assert (rast == 0).all()


def random_scene(seed, n=2, H=24, W=20, dtype=torch.float64, nv=14, nf=18):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-0.95, 0.95, (n, nv, 2))
    w = rng.uniform(0.6, 1.6, (n, nv, 1))
    z = rng.uniform(-0.5, 0.5, (n, nv, 1))
    pos = torch.tensor(np.concatenate([xy * w, z * w, w], 2), dtype=dtype)
    faces, seen = [], set()
    while len(faces) < nf:                      # unique vertex triples: coplanar duplicates would tie in depth everywhere
        t = rng.permutation(nv)[:3]
        if frozenset(t.tolist()) not in seen:
            seen.add(frozenset(t.tolist()))
            faces.append(t)
    tri = torch.tensor(np.stack(faces), dtype=torch.int32)
    return pos, tri, rng


def fd_check(fn, x, rng, n_dirs=6, eps=1e-6, tol=2e-5, same=None):
    """directional central differences of scalar fn at x against autograd."""
    x = x.clone().requires_grad_(True)
    y = fn(x)
    (g,) = torch.autograd.grad(y, x)
    checked = 0
    for _ in range(n_dirs * 3):
        d = torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=x.dtype)
        with torch.no_grad():
            if same is not None and not (same(x + eps * d) and same(x - eps * d)):
                continue            # the perturbation crossed a coverage (snap) boundary: FD undefined there
            num = (fn(x + eps * d) - fn(x - eps * d)) / (2 * eps)
        ana = (g * d).sum()
        assert abs(num - ana) <= tol * max(1.0, abs(float(ana))), (float(num), float(ana))
        checked += 1
        if checked == n_dirs:
            break
    assert checked >= 3


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fd_rasterize_interpolate(oracle, seed):
    pos, tri, rng = random_scene(seed)
    n, H, W, C = pos.shape[0], 24, 20, 5
    attr = torch.tensor(rng.standard_normal((n, pos.shape[1], C)))
    wt = torch.tensor(rng.standard_normal((n, H, W, C)))
    ids0 = oracle.rasterize(None, pos, tri, (H, W))[0][..., 3].clone()

    def loss_pos(p):
        rast, _ = oracle.rasterize(None, p, tri, (H, W))
        # exclude pixels where the barycentric clamp is active (backward is the unclamped derivative, as nvdiffrast)
        u, v = rast[..., 0], rast[..., 1]
        ok = ((u > 0) & (u < 1) & (v > 0) & (v < 1)).to(p.dtype)[..., None]
        out, _ = oracle.interpolate(attr, rast, tri)
        return (out * wt * ok.detach()).sum()

    same = lambda p: torch.equal(oracle.rasterize(None, p, tri, (H, W))[0][..., 3], ids0)
    fd_check(loss_pos, pos, rng, same=same)
    rast, _ = oracle.rasterize(None, pos, tri, (H, W))
    fd_check(lambda a: (oracle.interpolate(a, rast, tri)[0] * wt).sum(), attr, rng)
    # broadcast attr [1,V,C]
    fd_check(lambda a: (oracle.interpolate(a, rast, tri)[0] * wt).sum(), attr[:1].contiguous(), rng)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fd_antialias(oracle, seed):
    pos, tri, rng = random_scene(seed + 10, nf=10)
    n, H, W, C = pos.shape[0], 24, 20, 3
    wt = torch.tensor(rng.standard_normal((n, H, W, C)))
    rast0, _ = oracle.rasterize(None, pos, tri, (H, W))
    color = torch.tensor(rng.uniform(0, 1, (n, H, W, C)))
    out, items, alphas = oracle.antialias(color, rast0, pos, tri, return_items=True)
    assert items.shape[0] > 10
    code0 = items.clone()

    def same(p):
        r, _ = oracle.rasterize(None, p, tri, (H, W))
        if not torch.equal(r[..., 3], rast0[..., 3]):
            return False
        _, it, _ = oracle.antialias(color, r, p, tri, return_items=True)
        return torch.equal(it, code0)

    def loss_pos(p):
        r, _ = oracle.rasterize(None, p, tri, (H, W))
        return (oracle.antialias(color, r.detach(), p, tri) * wt).sum()

    fd_check(loss_pos, pos, rng, same=same, tol=5e-5)
    fd_check(lambda c: (oracle.antialias(c, rast0, pos, tri) * wt).sum(), color, rng)


def test_antialias_invariants(oracle):
    """(i) pixels in no differing pair are copied; (ii) moving a straight silhouette edge by delta px changes the AA
    coverage sum by delta * edge length; (iii) f32 and f64 oracles agree."""
    H = W = 32
    tri = torch.tensor([[0, 1, 2], [0, 2, 3]], dtype=torch.int32)

    def square(x1, dtype=torch.float64):
        c = [[6.3, 5.2], [x1, 5.2], [x1, 25.7], [6.3, 25.7]]
        return ortho_pos([[px2ndc(x, W), px2ndc(y, H)] for x, y in c], dtype=dtype)

    def aa_mask(pos):
        rast, _ = oracle.rasterize(None, pos, tri, (H, W))
        ones = torch.ones(1, 4, 1, dtype=pos.dtype)
        m, _ = oracle.interpolate(ones, rast, tri)
        return oracle.antialias(m, rast, pos, tri), m, rast

    a0, m0, rast = aa_mask(square(20.2))
    ids = rast[0, :, :, 3]
    diff = torch.zeros(H, W, dtype=torch.bool)
    diff[:, :-1] |= ids[:, :-1] != ids[:, 1:]
    diff[:, 1:] |= ids[:, :-1] != ids[:, 1:]
    diff[:-1, :] |= ids[:-1, :] != ids[1:, :]
    diff[1:, :] |= ids[:-1, :] != ids[1:, :]
    assert torch.equal(a0[0, :, :, 0][~diff], m0[0, :, :, 0][~diff])
    assert (a0 >= -1e-9).all() and (a0 <= 1 + 1e-9).all()
    # rows 8..20: the triangle at the silhouette pixel owns the moving edge (near the corners the pixel's own triangle is
    # the sliver across the diagonal, whose edges are not the silhouette -> no blend; nvdiffrast's analysis has the same limit)
    xs_ = (20.2, 20.45, 20.7, 21.1)
    sums = [float(aa_mask(square(x))[0][0, 8:21].sum()) for x in xs_]
    for k in range(3):
        assert abs((sums[k + 1] - sums[k]) - (xs_[k + 1] - xs_[k]) * 13) < 1e-6, (xs_[k], sums[k], sums[k + 1])
    # and the AA coverage of those rows equals the exact box area
    assert abs(sums[0] - (20.2 - 6.3) * 13) < 1e-6
    # analytic area: AA coverage sum equals the exact box area along x for every fully crossed row
    a32, _, _ = aa_mask(square(20.2, torch.float32))
    assert (a32.double() - a0).abs().max() < 1e-5


def test_f32_f64_oracles_agree(oracle):
    pos, tri, rng = random_scene(5, n=3, H=40, W=36, nv=40, nf=60)
    H, W = 40, 36
    r64, _ = oracle.rasterize(None, pos, tri, (H, W))
    r32, _ = oracle.rasterize(None, pos.float(), tri, (H, W))
    same = r64[..., 3] == r32[..., 3].double()
    assert same.float().mean() > 0.995
    assert (r64[..., :3] - r32[..., :3].double())[same].abs().max() < 2e-5
    color = torch.tensor(rng.uniform(0, 1, (3, H, W, 2)))
    a64 = oracle.antialias(color, r32.double(), pos, tri)
    a32 = oracle.antialias(color.float(), r32, pos.float(), tri)
    assert ((a64 - a32.double()).abs() > 1e-4).float().mean() < 0.005


def test_topology_build(oracle):
    tri = torch.tensor([[0, 1, 2], [0, 2, 3], [2, 1, 4]], dtype=torch.int32)
    opp = oracle.topology_build(tri, 5).numpy()
    # face 0 edges opposite v0:(1,2) v1:(2,0) v2:(0,1)
    assert opp[0].tolist() == [4, 3, -1]
    assert opp[1].tolist() == [-1, -1, 1]
    assert opp[2].tolist() == [-1, -1, 0]


def test_antialias_is_linear_in_colour_for_a_fixed_geometry(oracle):
    """What the fused stage-1 step (fmhr_amd/stage1.py) rests on: with rast / pos fixed, antialias is a LINEAR map of its
    colour input — out[tgt] += alpha (in[gid1] - in[gid]) per blending pair, the pairs and alphas depending on the geometry only — and
    that map is a 4-neighbour stencil: out[p] = in[p] + sum over the pairs that blend into p of |alpha| (in[other] - in[p])."""
    from util import mesh_scene, sort_items
    sc = mesh_scene((2, 48, 40, 9, 13), seed=3)
    pos, tri = sc["pos"].double(), sc["faces"]
    H, W = sc["H"], sc["W"]
    rast, _ = oracle.rasterize(None, pos, tri, (H, W))
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, H, W, 3, generator=g, dtype=torch.float64)
    y = torch.rand(2, H, W, 3, generator=g, dtype=torch.float64)
    ax, items_x, alphas_x = oracle.antialias(x, rast, pos, tri, return_items=True)
    ay, items_y, alphas_y = oracle.antialias(y, rast, pos, tri, return_items=True)
    assert items_x.shape[0] > 0, "the scene must have blending pairs"
    ix, lx = sort_items(items_x, alphas_x)
    iy, ly = sort_items(items_y, alphas_y)
    assert torch.equal(ix, iy) and torch.equal(lx, ly)              # the work items do not depend on the colour
    a, b = 0.37, -1.9
    axy = oracle.antialias(a * x + b * y, rast, pos, tri)
    assert (axy - (a * ax + b * ay)).abs().max() < 1e-12
    # the stencil form, rebuilt from the items exactly as fmhr_stage1_row_ptr / _fill do
    want = x.clone()
    for (px, py, bv, code), al in zip(ix.tolist(), lx.tolist()):
        down = (code >> 2) & 1
        p0, p1 = (bv, py, px), (bv, py + down, px + 1 - down)
        tgt, oth = (p0, p1) if al > 0 else (p1, p0)
        want[tgt] += abs(al) * (x[oth] - x[tgt])
    assert (want - ax).abs().max() < 1e-12


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_view_parallel_loops_do_not_depend_on_the_thread_count(oracle, dtype):
    """the C ops loop over views with OpenMP (bench.py's cpu_baseline runs them on the box's cores): one view per thread at a time,
    per-view work-item lists concatenated in view order — forward images, the work queue and every gradient must be bit-identical
    with 1 and with 4 threads (6 views, so threads hold several views and finish out of order)."""
    def run():
        pos, tri, rng = random_scene(5, n=6, H=40, W=36, dtype=dtype, nv=40, nf=70)
        pos = pos.clone().requires_grad_(True)
        attr = torch.tensor(rng.standard_normal((6, 40, 3)), dtype=dtype, requires_grad=True)
        rast, _ = oracle.rasterize(None, pos, tri, (40, 36))
        col, _ = oracle.interpolate(attr, rast, tri)
        out, items, alphas = oracle.antialias(col, rast, pos, tri, return_items=True)
        w = torch.tensor(rng.standard_normal(tuple(out.shape)), dtype=dtype)
        (out * w).sum().backward()
        return [rast.detach(), col.detach(), out.detach(), items, alphas, pos.grad.clone(), attr.grad.clone()]
    try:
        assert oracle.set_threads(1) == 1
        a = run()
        assert oracle.set_threads(4) == 4
        b = run()
    finally:
        oracle.set_threads()
    assert int(a[3].shape[0]) > 20 and float(a[5].abs().max()) > 0
    for x, y in zip(a, b):
        assert torch.equal(torch.as_tensor(x), torch.as_tensor(y))

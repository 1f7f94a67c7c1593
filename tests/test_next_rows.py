"""CPU tests of the "next" rows: conf parsing, camera decomposition + clip projection (N1), Loop subdivision
with the reference's index contract (N2), SH initialisation + stage 1 on the operator surface (N3, run here on the CPU checker),
OBJ writers (N4)."""
import os

import numpy as np
import pytest
import torch

from util import mesh_scene, rel_err


def test_parse_conf():
    from fmhr_amd.data import parse_conf
    c = parse_conf("""
        data_type = demo
        num = 16   # views
        w = 1280
        sfs_weight = 50
        edge_weight = 500000
        lr = 0.0001
        is_continue = False
        out_mesh_dire = "demo_sfs"
    """)
    assert c == dict(data_type="demo", num=16, w=1280, sfs_weight=50, edge_weight=500000, lr=1e-4, is_continue=False,
                     out_mesh_dire="demo_sfs")


def test_camera_decomposition_and_projection():
    from fmhr_amd.data import load_K_Rt_from_P, clip_projection
    rng = np.random.default_rng(0)
    for _ in range(5):
        K = np.array([[2300 + rng.uniform(-50, 50), 0.0, 640 + rng.uniform(-30, 30)],     # zero skew (a skew would stay in pixel
                      # units: only 8 entries of proj are rewritten, get_data.py:66-73)
                      [0, 2350 + rng.uniform(-50, 50), 512 + rng.uniform(-30, 30)], [0, 0, 1.0]])
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        R = q * np.sign(np.linalg.det(q))
        t = rng.standard_normal(3)
        P = K @ np.concatenate([R, t[:, None]], 1) * rng.uniform(0.5, 2.0) * (1 if rng.uniform() > 0.5 else -1)
        intr, w2c = load_K_Rt_from_P(P)
        assert np.allclose(intr[:3, :3], K, atol=1e-6 * 2300) and np.allclose(w2c[:3, :3], R, atol=1e-8)
        assert np.allclose(w2c[:3, 3], t, atol=1e-8)
        proj = clip_projection(intr, (1280, 1024))
        X = np.append(rng.standard_normal(3) + [0, 0, 5], 1.0)
        cam = w2c @ X
        clip = proj @ cam
        pix = K @ cam[:3]
        u, v = pix[0] / pix[2], pix[1] / pix[2]
        # NDC of the clip point must be the pixel coordinate mapped to [-1,1] (no y flip), w = camera depth, z = -0.1
        assert abs(clip[0] / clip[3] - (u / 640 - 1)) < 1e-9 and abs(clip[1] / clip[3] - (v / 512 - 1)) < 1e-9
        assert abs(clip[3] - cam[2]) < 1e-12 and abs(clip[2] + 0.1) < 1e-12
    path = "/root/reference/demo_data/1/camera/param.npz"     # present in the build container only
    if os.path.exists(path):
        cam = np.load(path)
        P = (cam["world_mat_0"] @ cam["scale_mat_0"])[:3]
        intr, w2c = load_K_Rt_from_P(P)
        Pr = intr[:3, :3] @ w2c[:3]
        assert np.allclose(Pr / Pr[2, 3], P / P[2, 3], atol=1e-6)


def test_loop_subdivision_contract():
    from fmhr_amd.subdivide import subdivide_loop, subdivide_loop_once
    from fmhr_amd.synthetic import blob_mesh
    v, f = blob_mesh(21, 37)                                   # V=778 like MANO, disc topology
    assert v.shape[0] == 778 and f.shape[0] == 1517
    E = (3 * 1517 + 37) // 2
    v1, f1 = subdivide_loop_once(v.astype(np.float64), f)
    assert v1.shape[0] == 778 + E and f1.shape[0] == 4 * 1517
    # index contract: parent corners stay at their indices in the first three children, midpoints are appended
    assert np.array_equal(f1[0::4, 0], f[:, 0]) and np.array_equal(f1[1::4, 1], f[:, 1]) and np.array_equal(f1[2::4, 2], f[:, 2])
    assert f1[:, :].min() >= 0 and (f1[3::4] >= 778).all()
    assert np.array_equal(f1[0::4, 1], f1[3::4, 0]) and np.array_equal(f1[1::4, 2], f1[3::4, 1]) and np.array_equal(f1[0::4, 2], f1[3::4, 2])
    v3, f3 = subdivide_loop(v, f, 3)
    assert f3.shape[0] == 64 * 1517
    e = np.concatenate([f3[:, [0, 1]], f3[:, [1, 2]], f3[:, [2, 0]]]); e.sort(1)
    assert v3.shape[0] - np.unique(e, axis=0).shape[0] + f3.shape[0] == 1            # Euler characteristic of a disc
    # MANO arithmetic of Appendix E: 778/1538 (E=2315) -> 49281/98432 after three levels
    V, Fc, Ec = 778, 1538, 2315
    for _ in range(3):
        V, Ec, Fc = V + Ec, 2 * Ec + 3 * Fc, 4 * Fc
    assert (V, Fc) == (49281, 98432)
    # a regular planar lattice is reproduced: interior valence-6 vertices keep their position, everything stays in the plane
    n = 6
    gx, gy = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    pv = np.stack([gx + 0.5 * gy, gy * np.sqrt(3) / 2, np.zeros_like(gx, float)], -1).reshape(-1, 3)
    idx = lambda i, j: i * n + j
    pf = np.array([[idx(i, j), idx(i + 1, j), idx(i, j + 1)] for i in range(n - 1) for j in range(n - 1)] +
                  [[idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(n - 1) for j in range(n - 1)])
    sv, sf = subdivide_loop_once(pv, pf)
    assert np.abs(sv[:, 2]).max() < 1e-12
    inner = [idx(i, j) for i in range(1, n - 1) for j in range(1, n - 1)]
    assert np.allclose(sv[inner], pv[inner], atol=1e-12)


def test_obj_writers(tmp_path):
    from fmhr_amd.data import save_obj, save_obj_mesh_with_color, load_obj
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0.5]], dtype=np.float32)
    f = np.array([[0, 1, 2]], dtype=np.int32)
    save_obj(tmp_path / "a.obj", v, f)
    v2, f2 = load_obj(tmp_path / "a.obj")
    assert np.allclose(v2, v) and np.array_equal(f2, f)
    save_obj_mesh_with_color(tmp_path / "c.obj", v, f, np.array([[0.1, 0.2, 0.3]] * 3))
    lines = open(tmp_path / "c.obj").read().splitlines()
    assert lines[0] == "v 0.0000 0.0000 0.0000 0.1000 0.2000 0.3000" and lines[-1] == "f 1 3 2"     # winding flipped as the reference


def _tiny_capture(oracle, dtype=torch.float32, seed=2):
    sc = mesh_scene((3, 72, 64, 10, 14), seed=seed)
    from checker_scenes import oracle_forward_fn
    from fmhr_amd import synthetic as S
    with torch.no_grad():
        tg = S.make_targets(sc, oracle_forward_fn(oracle, sc))
    sc.update(tg)
    sc["gray"] = 0.114 * sc["img"][..., 0] + 0.587 * sc["img"][..., 1] + 0.299 * sc["img"][..., 2]
    return sc


def test_init_and_stage1_on_the_checker(oracle):
    """the product's init_sh / stage1_loss code, executed on the CPU checker's operator surface"""
    from fmhr_amd import pipeline
    sc = _tiny_capture(oracle)
    res = (sc["H"], sc["W"])
    init = pipeline.init_sh(oracle, sc["vertices"], sc["faces"], sc["img"], sc["gray"], sc["mask"], sc["w2c"], sc["proj"], res)
    assert init["sh_coeffs"].shape == (3, 9) and init["valid_masks"].shape == (3, sc["H"], sc["W"])
    assert torch.isfinite(init["sh_coeffs"]).all() and (init["albedo_mean"] > 0).all()
    # the target images were shaded with c0 = 0.8 and albedo ~ U[0.3, 0.9]: the grey-level fit must land on c0 * mean albedo
    assert 0.3 < float(init["sh_global"][0]) < 0.7
    # stage-1 loss: fp64 directional finite differences w.r.t. albedo and the per-view SH coefficients
    d = lambda t: t.double()
    alb = (torch.zeros_like(sc["vertices"]) + init["albedo_mean"]).double()[None].clone().requires_grad_(True)
    sh = init["sh_coeffs"].double().clone().requires_grad_(True)
    f = lambda a, s: pipeline.stage1_loss(oracle, None, d(sc["vertices"]), sc["faces"], a, s, d(sc["w2c"]), d(sc["proj"]),
                                          d(sc["img"]), d(sc["mask"]), res, 50.0)
    loss = f(alb, sh)
    loss.backward()
    assert float(loss) > 0
    rng = np.random.default_rng(0)
    for _ in range(4):
        da = torch.tensor(rng.standard_normal(alb.shape)) * 1e-6
        ds = torch.tensor(rng.standard_normal(sh.shape)) * 1e-6
        with torch.no_grad():
            num = float(f(alb + da, sh + ds) - f(alb - da, sh - ds)) / 2
        ana = float((alb.grad * da).sum() + (sh.grad * ds).sum())
        assert abs(num - ana) <= 1e-3 * abs(ana) + 1e-9, (num, ana)


def test_demo_capture_cameras_against_the_captures_own_calibration():
    """Row N1 on the reference's OWN capture (demo_data/1; committed as tests/golden/demo_ingest.npz by make_demo_fixture.py): the
    cameras fmhr_amd.data.get_demo_data derives from world_mat @ scale_mat (cv2.decomposeProjectionMatrix replaced by an RQ
    decomposition) must reproduce the intrinsics / rotations the capture stores next to them (param.npz: int_i, ext_i), the clip
    projection must be the edit of get_data.py:66-73, and the capture's 3-D hand keypoints must project into the images and onto
    the hand masks."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_ingest.npz"))
    W, H = g["res"]
    X = np.concatenate([g["keypoints"], np.ones((g["keypoints"].shape[0], 1), np.float32)], 1)
    frac_in_mask = []
    for i in range(16):
        w2c, proj, K = g["w2cs"][i].T, g["projs"][i].T, g["ints"][i]          # stored transposed (row-vector convention, :96-97)
        assert abs(proj[0, 0] * W / 2 - K[0, 0]) < 1e-5 * K[0, 0] and abs(proj[1, 1] * H / 2 - K[1, 1]) < 1e-5 * K[1, 1]
        assert abs((proj[0, 2] + 1) * W / 2 - K[0, 2]) < 1e-3 and abs((proj[1, 2] + 1) * H / 2 - K[1, 2]) < 1e-3
        assert proj[2, 3] == np.float32(-0.1) and proj[3, 2] == 1 and proj[2, 2] == 0 and proj[3, 3] == 0
        assert np.abs(w2c[:3, :3] - g["exts"][i][:3, :3]).max() < 1e-6        # rotation of the stored extrinsics
        assert abs(np.linalg.det(w2c[:3, :3]) - 1) < 1e-5
        # translation: the stored extrinsic maps the normalised (scale_mat) space; P = world_mat @ scale_mat folds the scale in
        s = g["scale0"][0, 0]
        assert np.allclose(w2c[:3, 3] * s, (g["exts"][i] @ g["scale0"])[:3, 3], atol=1e-5)
        clip = (proj @ (w2c @ X.T)).T
        assert (clip[:, 3] > 0.1).all()                                       # in front of every camera, beyond the near plane
        px, py = (clip[:, 0] / clip[:, 3] + 1) * W / 2, (clip[:, 1] / clip[:, 3] + 1) * H / 2
        inside = (px >= 0) & (px < W) & (py >= 0) & (py < H)
        assert inside.mean() >= 0.9
        if g["mask_sum"][i] / (W * H) > 0.05:
            m = g["mask_thumb"][i][np.clip((py // 32).astype(int), 0, H // 32 - 1), np.clip((px // 32).astype(int), 0, W // 32 - 1)] > 0
            frac_in_mask.append((m & inside).mean())
    assert len(frac_in_mask) >= 10 and np.mean(frac_in_mask) > 0.5 and max(frac_in_mask) > 0.9


@pytest.mark.skipif(not os.path.exists("/root/reference/demo_data/1/camera/param.npz"), reason="the reference capture is only in the build container")
def test_demo_capture_ingest_runs_and_matches_the_fixture():
    """get_demo_data on /root/reference/demo_data/1 (16 views, 1280 x 1024 as conf/demo_sfs.conf:11-12), device = cpu: cameras
    identical to the committed fixture, images / grey images / masks equal to its thumbnails and checksums."""
    from fmhr_amd.data import get_demo_data
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_ingest.npz"))
    imgs, grays, masks, w2cs, projs = get_demo_data("/root/reference/demo_data", 1, 16, (1280, 1024), device="cpu")
    assert imgs.shape == (16, 1024, 1280, 3) and grays.shape == (16, 1024, 1280) and masks.shape == (16, 1024, 1280)
    assert np.array_equal(w2cs.numpy(), g["w2cs"]) and np.array_equal(projs.numpy(), g["projs"])
    thumb = lambda t: t.reshape(16, 32, 32, 40, 32, *t.shape[3:]).mean((2, 4)).numpy()
    assert np.abs(thumb(imgs) - g["img_thumb"]).max() < 1e-6 and np.abs(thumb(masks) - g["mask_thumb"]).max() < 1e-6
    assert np.allclose(imgs.double().sum((1, 2, 3)).numpy(), g["img_sum"]) and np.allclose(masks.double().sum((1, 2)).numpy(), g["mask_sum"])
    assert set(np.unique(masks.numpy())) <= {0.0, 1.0} and float(imgs.max()) <= 1.0
    assert float((imgs.sum(-1) > 0)[masks == 0].float().mean()) < 0.02       # with_mask: the background is blanked (resize halo aside)


def test_loop_subdivision_of_the_exact_mano_count_base():
    """Row N2: three Loop subdivisions of a 778-vertex / 1538-face / 2315-edge disc (MANO's counts: mano/TEMPLATE_RIGHT.obj) give
    exactly the sizes /root/reference/train_mlp.py:150-154 hard-codes, level by level, and the child-face
    layout /root/reference/repose.py:14-24 (`subdivide_weight`) relies on: of the four consecutive children tmp[0..3] of a parent,
    tmp[0][1] is the new vertex of edge (tmp[0][0], tmp[1][1]), tmp[0][2] of (tmp[0][0], tmp[2][2]), tmp[1][2] of (tmp[1][1], tmp[2][2])."""
    from fmhr_amd.subdivide import subdivide_loop_once
    from fmhr_amd.synthetic import mano_disc_mesh
    v, f = mano_disc_mesh()
    counts = []
    v, f = v.astype(np.float64), f.astype(np.int64)
    for level in range(3):
        V0 = v.shape[0]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]); e.sort(1)
        uniq = np.unique(e[:, 0] * V0 + e[:, 1])
        counts.append((V0, f.shape[0], uniq.shape[0]))
        v1, f1 = subdivide_loop_once(v, f)
        f1 = np.asarray(f1)
        # repose.py:14-24 on the new faces: every new vertex is the midpoint-vertex of exactly the edge the reference averages over
        edge_of = {V0 + k: (int(u // V0), int(u % V0)) for k, u in enumerate(uniq)}        # odd vertices in unique-edge order
        for i in range(0, f1.shape[0], 4):
            t = f1[i:i + 4]
            for new, a, b in ((t[0][1], t[0][0], t[1][1]), (t[0][2], t[0][0], t[2][2]), (t[1][2], t[1][1], t[2][2])):
                assert edge_of[int(new)] == (min(int(a), int(b)), max(int(a), int(b)))
            assert t[0][0] < V0 and t[1][1] < V0 and t[2][2] < V0 and set(t[3]) == {t[0][1], t[0][2], t[1][2]}
        assert np.array_equal(v1[:V0].shape, v.shape)                                       # even vertices keep their indices
        v, f = v1, f1
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]); e.sort(1)
    counts.append((v.shape[0], f.shape[0], np.unique(e[:, 0] * v.shape[0] + e[:, 1]).shape[0]))
    assert counts == [(778, 1538, 2315), (3093, 6152, 9244), (12337, 24608, 36944), (49281, 98432, 147712)]
    assert np.isfinite(v).all() and float(np.abs(v).max()) < 0.2                            # a hand-sized smooth surface


def test_rerender_png_writer(tmp_path):
    """mesh_sfs_optim.py:342-343: (rendered * 255).astype(np.int32) written by cv2.imwrite = truncation toward zero, saturation to
    8 bits, the loader's BGR array stored as it stands (the file shows RGB)."""
    from PIL import Image
    from fmhr_amd.data import save_rerender_png
    img = np.zeros((2, 3, 3))
    img[0, 0] = [1.0, 0.5, 0.0]           # B, G, R
    img[0, 1] = [1.2, -0.1, 0.999]        # out of range both ways
    img[1, 2] = [0.25, 0.25, 0.25]
    save_rerender_png(tmp_path / "m.png", img)
    a = np.asarray(Image.open(tmp_path / "m.png"))
    assert a.shape == (2, 3, 3) and a.dtype == np.uint8
    assert a[0, 0].tolist() == [0, 127, 255] and a[0, 1].tolist() == [254, 0, 255] and a[1, 2].tolist() == [63, 63, 63]


def test_vertex_order_is_a_relabelling_of_the_same_mesh(monkeypatch):
    """fmhr_amd.ham.VertexOrder (the first-touch renumbering run_ham and bench.py optimise in): a permutation of the vertices, the
    faces in their order with their corner order, per-vertex data there and back unchanged, an unused vertex kept (at the end), the
    gathers of 64 consecutive faces within fewer cache lines on a subdivided mesh — and nothing at all with FMHR_VERTEX_ORDER=0 /
    enabled=False or on a mesh already in that order."""
    from fmhr_amd.ham import VertexOrder
    from fmhr_amd.synthetic import mano_disc_mesh
    v, f = mano_disc_mesh(0, subdiv=2)
    V = v.shape[0] + 1                                                  # one vertex no face refers to
    v = np.concatenate([v, [[9.0, 9.0, 9.0]]]).astype(np.float32)
    faces = torch.from_numpy(f.astype(np.int32))
    order = VertexOrder(faces, V, enabled=True)
    assert not order.identity
    o = order.old_of_new.numpy()
    assert sorted(o.tolist()) == list(range(V)) and o[-1] == V - 1      # a permutation; the unused vertex goes last
    assert np.array_equal(order.new_of_old.numpy()[o], np.arange(V))
    x = torch.from_numpy(v)
    xi = order.to_internal(x)
    assert torch.equal(order.to_caller(xi), x)
    assert torch.equal(xi[order.faces.long()], x[faces.long()])        # the same triangles, face by face, corner by corner
    y = torch.arange(2 * V * 3, dtype=torch.float32).reshape(2, V, 3)  # [1,V,3]-style tensors: dim = 1
    assert torch.equal(order.to_caller(order.to_internal(y, 1), 1), y)
    with pytest.raises(RuntimeError):
        order.to_internal(torch.zeros(V + 1, 3))
    first = order.faces.reshape(-1).numpy()
    _, idx = np.unique(first, return_index=True)
    assert np.array_equal(np.argsort(idx, kind="stable"), np.arange(V - 1))     # vertex k is the k-th one the face list touches

    def lines(ft, bytes_per_vertex):                                   # distinct 128-B lines the corners of 64 consecutive faces fall in
        per = 128 // bytes_per_vertex
        return np.mean([len(np.unique(ft[s:s + 64] // per)) for s in range(0, len(ft), 64)])
    assert lines(order.faces.numpy(), 8) < 0.7 * lines(f, 8)                 # 6.6 against 10.9 here; 7.3 against 17.6 at three levels
    again = VertexOrder(order.faces, V, enabled=True)
    assert again.identity and again.faces is order.faces and again.to_internal(x) is x
    off = VertexOrder(faces, V, enabled=False)
    assert off.identity and torch.equal(off.faces, faces)
    monkeypatch.setenv("FMHR_VERTEX_ORDER", "0")                       # the environment switch run_ham and bench.py obey
    assert VertexOrder(faces, V).identity
    monkeypatch.delenv("FMHR_VERTEX_ORDER")
    assert not VertexOrder(faces, V).identity
    with pytest.raises(RuntimeError):
        VertexOrder(faces, V - 5)


def test_permutation_chunks_are_permutations_and_their_batch_masks():
    """fmhr_amd.pipeline.PermChunks (the loops' `perm = torch.randperm(num)` per epoch, mesh_sfs_optim.py:195,249, drawn 64 epochs at a
    time): every row a permutation, row j of an epoch's masks exactly the views of `perm[j*batch:(j+1)*batch]` (a ragged last batch
    included), across a refill, reproducible from the generator's seed"""
    import torch
    from fmhr_amd.pipeline import PermChunks
    for num, batch in ((7, 3), (16, 8), (5, 5), (4, 9)):
        runs = []
        for rep in range(2):
            pc = PermChunks(num, batch, torch.Generator().manual_seed(11), torch.device("cpu"))
            rows = []
            for ep in range(PermChunks.kChunk + 3):
                perm, masks = pc.next()
                assert perm.dtype == torch.int32 and sorted(perm.tolist()) == list(range(num))
                nb = (num + batch - 1) // batch
                assert tuple(masks.shape) == (nb, num) and int(masks.sum()) == num
                for j in range(nb):
                    assert sorted(torch.nonzero(masks[j]).flatten().tolist()) == sorted(perm[j * batch:(j + 1) * batch].tolist())
                rows.append(perm.clone())
            runs.append(torch.stack(rows))
        assert torch.equal(runs[0], runs[1])
        assert len({tuple(r.tolist()) for r in runs[0]}) > (1 if num > 1 else 0)


def test_permutation_sources_are_deterministic_and_well_formed():
    """The view-sharded loop relies on every rank drawing the SAME permutations from the same seed (fmhr_amd.pipeline.PermChunks) and on
    replayed ones (FixedPerms) being handed out unchanged, with batch masks that partition the views (mesh_sfs_optim.py:195,249,255-260)."""
    from fmhr_amd.pipeline import FixedPerms, PermChunks
    cpu = torch.device("cpu")
    a = PermChunks(16, 8, torch.Generator().manual_seed(5), cpu)
    b = PermChunks(16, 8, torch.Generator().manual_seed(5), cpu)
    seen = []
    for _ in range(70):                                              # across a refill of the 64-epoch chunk
        (pa, ma), (pb, mb) = a.next(), b.next()
        assert torch.equal(pa, pb) and torch.equal(ma, mb)
        assert sorted(pa.tolist()) == list(range(16)) and ma.shape == (2, 16)
        assert torch.equal(ma.sum(0), torch.ones(16, dtype=torch.int32))               # every view in exactly one batch
        assert sorted(torch.nonzero(ma[0]).flatten().tolist()) == sorted(pa[:8].tolist())
        seen.append(tuple(pa.tolist()))
    assert len(set(seen)) > 60                                       # fresh draws, not one permutation repeated
    c = PermChunks(16, 8, torch.Generator().manual_seed(6), cpu)
    assert not torch.equal(c.next()[0], torch.tensor(seen[0], dtype=torch.int32))
    given = [[2, 0, 1, 3, 4], [4, 3, 2, 1, 0]]
    f = FixedPerms(given, 2, cpu)
    p0, m0 = f.next()
    assert p0.tolist() == given[0] and m0.shape == (3, 5) and m0[2].tolist() == [0, 0, 0, 0, 1]   # ragged last batch: view 4 alone
    assert f.next()[0].tolist() == given[1]
    with pytest.raises(RuntimeError, match="2 permutations were given"):
        f.next()
    with pytest.raises(RuntimeError, match="permutation of range"):
        FixedPerms([[0, 0, 1]], 2, cpu)

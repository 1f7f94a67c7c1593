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
    from test_gpu_ham import oracle_forward_fn
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

"""The oracle's restatement of models/utils.py + ncc_utils.py against golden vectors generated from the imported reference
(tests/golden/make_goldens.py) and the hand-written known answers of SURVEY.md Appendix D."""
import numpy as np
import torch

T = torch.from_numpy


def close(a, b, tol=2e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()


def test_known_answers_appendix_d(oracle, golden):
    g = golden
    V, F = T(g["ka_V"]), T(g["ka_F"])
    close(oracle.get_normals(V[None], F), g["ka_normals"])
    # literal values written in SURVEY.md Appendix D (KA1, KA2, KA3)
    close(g["ka_normals"][0, 0], [0.2357023, -0.2357023, 0.9428091], 1e-6)
    close(oracle.get_radiance(T(g["ka_c"]), T(g["ka_n"]), 3), [1.9320002, -1.3000000, 1.2529485], 1e-6)
    close(oracle.get_matrix(T(g["ka_n"]), 3), g["ka_matrix"])
    Vg = V.clone().requires_grad_(True)
    ls = oracle.laplacian_smoothing(Vg, F, "uniform")
    ls.backward()
    close(ls.detach(), 0.8719966, 1e-6)
    close(Vg.grad, g["ka_lap_grad"])
    close(oracle.compute_laplacian(V, F).to_dense(), g["ka_L_dense"])
    e, f2e = oracle.get_edges(V, F)
    assert np.array_equal(e.numpy(), g["ka_edges"]) and np.array_equal(f2e.numpy(), g["ka_f2e"])
    ncc = oracle.NCC(T(g["ka_ncc_ref"]), T(g["ka_ncc_src"]), None, T(g["ka_ncc_mask"]))
    close(ncc, [[1, 0], [-1, 0]], 1e-6)


def test_normals_golden(oracle, golden):
    g = golden
    vt, ft = T(g["m_verts"]), T(g["m_faces"])
    vb = torch.stack([vt, vt * 1.5 + 0.1]).requires_grad_(True)
    nrm = oracle.get_normals(vb, ft)
    (nrm * T(g["m_normals_w"])).sum().backward()
    close(nrm.detach(), g["m_normals"])
    close(vb.grad, g["m_normals_grad"], 1e-5)


def test_laplacian_golden(oracle, golden):
    g = golden
    ft = T(g["m_faces"])
    for x, val, grad in ((g["m_verts"], g["m_lap"], g["m_lap_grad"]), (g["m_albedo"], g["m_alap"], g["m_alap_grad"])):
        xt = T(x).clone().requires_grad_(True)
        ls = oracle.laplacian_smoothing(xt, ft, "uniform")
        ls.backward()
        close(ls.detach(), val)
        close(xt.grad, grad, 1e-5)
    L = oracle.compute_laplacian(T(g["m_verts"]), ft)
    assert np.array_equal(L.indices().numpy(), g["m_L_idx"])
    close(L.values(), g["m_L_val"])
    e, f2e = oracle.get_edges(T(g["m_verts"]), ft)
    assert np.array_equal(e.numpy(), g["m_edges"]) and np.array_equal(f2e.numpy(), g["m_f2e"])


def test_radiance_golden(oracle, golden):
    g = golden
    n = T(g["r_n"]).clone().requires_grad_(True)
    c = T(g["r_c"]).clone().requires_grad_(True)
    rad = oracle.get_radiance(c, n, 3)
    (rad * T(g["r_w"])).sum().backward()
    close(rad.detach(), g["r_rad"])
    close(n.grad, g["r_gn"], 1e-5)
    close(c.grad, g["r_gc"], 1e-5)
    close(oracle.get_matrix(T(g["r_n"]), 3), g["r_matrix"])
    close(oracle.get_matrix(g["r_n"], 3), g["r_matrix"])
    close(oracle.get_radiance(T(g["r_c"])[0], T(g["r_n"]), 3), g["r_rad_shared"])


def test_ncc_golden(oracle, golden):
    g = golden
    out = oracle.NCC(T(g["n_ref"]), T(g["n_src"]), None, T(g["n_mask"]))
    close(out, g["n_ncc"], 1e-5)

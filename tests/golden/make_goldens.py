"""tests/golden/make_goldens.py — regenerates tests/golden/ref_helpers.npz by IMPORTING the reference (read-only, build
container only; /root/reference does not exist on the GPU box).  Run from any directory outside /root/reference:

    python tests/golden/make_goldens.py

Only inputs and the reference's outputs are stored (data, no reference source).  Functions exercised:
models/utils.py get_normals / get_matrix / get_radiance / get_edges / compute_laplacian / laplacian_smoothing("uniform")
and models/ncc_utils.py NCC  (SURVEY.md §8c, Appendix C/D).
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
for m in ['cv2', 'skimage', 'skimage.measure', 'plyfile', 'trimesh']:
    sys.modules[m] = types.ModuleType(m)
sys.modules['skimage'].measure = sys.modules['skimage.measure']
import warnings
warnings.filterwarnings("ignore")
import models.utils as mu          # noqa: E402
import models.ncc_utils as ncc     # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_helpers.npz")


def disc_mesh(r, c, rng):
    """open-bottom UV blob: r rings x c segments + a pole (disc topology like MANO)."""
    verts = [[0.0, 0.0, 1.0]]
    for i in range(1, r + 1):
        th = 0.5 * np.pi * i / r * 1.3
        for j in range(c):
            ph = 2 * np.pi * j / c
            verts.append([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
    verts = np.asarray(verts) * (1 + 0.1 * rng.standard_normal((len(verts), 1)))
    faces = []
    for j in range(c):
        faces.append([0, 1 + j, 1 + (j + 1) % c])
    for i in range(1, r):
        a0, b0 = 1 + (i - 1) * c, 1 + i * c
        for j in range(c):
            j1 = (j + 1) % c
            faces.append([a0 + j, b0 + j, b0 + j1])
            faces.append([a0 + j, b0 + j1, a0 + j1])
    return verts.astype(np.float32), np.asarray(faces, dtype=np.int64)


def main():
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    g = {}
    # ---- Appendix D known answers (hand-written inputs)
    V = torch.tensor([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0.5]])
    Fc = torch.tensor([[0, 1, 2], [0, 2, 3]])
    g["ka_V"], g["ka_F"] = V.numpy(), Fc.numpy()
    g["ka_normals"] = mu.get_normals(V[None], Fc).numpy()
    nn = torch.tensor([[.6, 0, .8], [0, 1, 0], [3 ** -0.5] * 3])
    cc = torch.arange(1, 10, dtype=torch.float32) / 10
    g["ka_n"], g["ka_c"] = nn.numpy(), cc.numpy()
    g["ka_radiance"] = mu.get_radiance(cc, nn, 3).numpy()
    g["ka_matrix"] = mu.get_matrix(nn, 3).numpy()
    Vg = V.clone().requires_grad_(True)
    ls = mu.laplacian_smoothing(Vg, Fc, "uniform")
    ls.backward()
    g["ka_lap"], g["ka_lap_grad"] = ls.detach().numpy(), Vg.grad.numpy()
    g["ka_L_dense"] = mu.compute_laplacian(V, Fc).to_dense().numpy()
    e, f2e = mu.get_edges(V, Fc)
    g["ka_edges"], g["ka_f2e"] = e.numpy(), f2e.numpy()
    ref = torch.tensor([[[1., 2, 3, 4], [1, 1, 1, 1]]])
    src = torch.tensor([[[2., 4, 6, 8], [0, 1, 0, 1]], [[4, 3, 2, 1], [1, 2, 3, 5]]])
    msk = torch.tensor([[[1., 1, 1, 1], [1, 1, 1, 1]], [[1, 1, 1, 0], [0, 0, 0, 0]]])
    g["ka_ncc_ref"], g["ka_ncc_src"], g["ka_ncc_mask"] = ref.numpy(), src.numpy(), msk.numpy()
    g["ka_ncc"] = ncc.NCC(ref, src, torch.ones_like(ref), msk).numpy()

    # ---- seeded mesh: normals fwd+bwd (batched), laplacian fwd+bwd (positions and 3-channel albedo), edges
    v, f = disc_mesh(6, 12, rng)
    g["m_verts"], g["m_faces"] = v, f
    vt = torch.from_numpy(v)
    ft = torch.from_numpy(f)
    vb = torch.stack([vt, vt * 1.5 + 0.1]).requires_grad_(True)
    nrm = mu.get_normals(vb, ft)
    wn = torch.from_numpy(rng.standard_normal(nrm.shape).astype(np.float32))
    (nrm * wn).sum().backward()
    g["m_normals"], g["m_normals_w"], g["m_normals_grad"] = nrm.detach().numpy(), wn.numpy(), vb.grad.numpy()
    vg = vt.clone().requires_grad_(True)
    ls = mu.laplacian_smoothing(vg, ft, "uniform")
    ls.backward()
    g["m_lap"], g["m_lap_grad"] = ls.detach().numpy(), vg.grad.numpy()
    alb = torch.from_numpy(rng.uniform(0.3, 0.9, v.shape).astype(np.float32)).requires_grad_(True)
    la = mu.laplacian_smoothing(alb, ft, "uniform")
    la.backward()
    g["m_albedo"], g["m_alap"], g["m_alap_grad"] = alb.detach().numpy(), la.detach().numpy(), alb.grad.numpy()
    L = mu.compute_laplacian(vt, ft).coalesce()
    g["m_L_idx"], g["m_L_val"] = L.indices().numpy(), L.values().numpy()
    e, f2e = mu.get_edges(vt, ft)
    g["m_edges"], g["m_f2e"] = e.numpy(), f2e.numpy()

    # ---- radiance: per-pixel coefficients, fwd + grads
    P = 257
    nr = torch.from_numpy(rng.standard_normal((P, 3)).astype(np.float32))
    nr = (nr / nr.norm(dim=1, keepdim=True)).requires_grad_(True)
    co = torch.from_numpy(rng.standard_normal((P, 9)).astype(np.float32)).requires_grad_(True)
    rad = mu.get_radiance(co, nr, 3)
    wr = torch.from_numpy(rng.standard_normal(P).astype(np.float32))
    (rad * wr).sum().backward()
    g["r_n"], g["r_c"], g["r_w"] = nr.detach().numpy(), co.detach().numpy(), wr.numpy()
    g["r_rad"], g["r_gn"], g["r_gc"] = rad.detach().numpy(), nr.grad.numpy(), co.grad.numpy()
    g["r_matrix"] = mu.get_matrix(nr.detach(), 3).numpy()
    g["r_rad_shared"] = mu.get_radiance(co[0].detach(), nr.detach(), 3).numpy()

    # ---- NCC seeded
    nv, Np, Npx = 3, 17, 25
    rf = torch.from_numpy(rng.uniform(0, 1, (1, Np, Npx)).astype(np.float32))
    sr = torch.from_numpy(rng.uniform(0, 1, (nv, Np, Npx)).astype(np.float32))
    mk = torch.from_numpy((rng.uniform(0, 1, (nv, Np, Npx)) > 0.3).astype(np.float32))
    mk[1, 3] = 0
    sr[2, 5] = 0.25
    g["n_ref"], g["n_src"], g["n_mask"] = rf.numpy(), sr.numpy(), mk.numpy()
    g["n_ncc"] = ncc.NCC(rf, sr, torch.ones_like(rf), mk).numpy()

    np.savez_compressed(OUT, **g)
    print("wrote", OUT, {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()

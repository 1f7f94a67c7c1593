"""tests/golden/make_boundary_goldens.py — regenerates tests/golden/boundary_loop.npz (build container only).

What the fixture pins: the reference's OWN statements of the HAM script — SH initialisation, stage 1 and stage 2 of
/root/reference/mesh_sfs_optim.py (its `main`, lines 120-317: `glctx = dr.RasterizeGLContext()` ... the end of the stage-2 loop) —
EXECUTED here, read from the reference at run time (never stored), over
  * `dr`              = the CPU checker's nvdiffrast-shaped ops (oracle/ham_oracle.py: nvdiffrast itself needs CUDA + OpenGL and is absent),
  * the helpers       = the reference's own `models/utils.py` functions (imported, as tests/golden/make_goldens.py does; in the fp64 runs
                        `laplacian_smoothing` is the checker's restatement: the reference's builds its operator in fp32 only),
  * `Adam`, `F`, `np` = the real ones (Adam and F.l1_loss wrapped only to RECORD the first batch's gradients and every batch's loss),
on a seeded synthetic capture.  Stored: the inputs (arrays) and the statements' outputs — per-view / global SH, albedo_mean, the start
silhouettes, the permutations drawn, the first stage-1 batch's loss and gradients, the parameters after stage 1, the first stage-2
batch's six losses and gradients, the parameters after stage 2.  Data only: no line of the reference is written anywhere.

The GPU test (tests/test_gpu_boundary.py) drives the PRODUCT through the build's own statements (fmhr_amd.pipeline.init_sh /
stage1_loss on fmhr_amd.dr + fmhr_amd.utils, HamStage2.step) on the same inputs and compares with these arrays.

    python tests/golden/make_boundary_goldens.py            (from any directory outside /root/reference; ~20 s)
"""
import os
import sys
import textwrap
import types
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
sys.path.insert(0, REF)
for m in ['cv2', 'skimage', 'skimage.measure', 'plyfile', 'trimesh']:
    sys.modules[m] = types.ModuleType(m)
sys.modules['skimage'].measure = sys.modules['skimage.measure']
warnings.filterwarnings("ignore")
import models.utils as mu                        # noqa: E402  (the reference's helpers)
import ham_oracle as oracle                      # noqa: E402  (the CPU checker's dr-shaped ops)

OUT = os.path.join(HERE, "boundary_loop.npz")
FIRST, LAST = 120, 317                           # mesh_sfs_optim.py: `glctx = dr.RasterizeGLContext()` ... end of the stage-2 loop body


def capture(dims=(4, 96, 80, 12, 16), seed=2):
    """the seeded synthetic capture (the scene tests/test_gpu_boundary.py used since round 3)"""
    from util import mesh_scene
    from checker_scenes import oracle_forward_fn
    from fmhr_amd import synthetic as S
    sc = mesh_scene(dims, seed=seed)
    with torch.no_grad():
        tg = S.make_targets(sc, oracle_forward_fn(oracle, sc))
    sc.update(tg)
    sc["mask"][:, : sc["H"] // 6] = 0                       # photometric mask != coverage somewhere
    sc["gray"] = 0.114 * sc["img"][..., 0] + 0.587 * sc["img"][..., 1] + 0.299 * sc["img"][..., 2]
    return sc


class Recorder:
    """what the statements leave no name for: the first batch's gradients of every optimiser, the L1 value of every batch, the
    permutation of every epoch"""

    def __init__(self):
        self.first_grads, self.l1, self.perms, self.steps = [], [], [], []

    def adam(self):
        rec = self

        class RecordingAdam(torch.optim.Adam):
            def __init__(self, groups, **kw):
                super().__init__(groups, **kw)
                rec.first_grads.append(None)
                rec.steps.append(0)
                self._slot = len(rec.first_grads) - 1

            def step(self, *a, **k):
                if rec.first_grads[self._slot] is None:
                    rec.first_grads[self._slot] = [None if p.grad is None else p.grad.detach().clone() for g in self.param_groups for p in g['params']]
                rec.steps[self._slot] += 1
                return super().step(*a, **k)
        return RecordingAdam

    def functional(self):
        import torch.nn.functional as F
        rec = self

        def l1_loss(a, b, *x, **k):
            v = F.l1_loss(a, b, *x, **k)
            rec.l1.append(float(v))
            return v
        return types.SimpleNamespace(normalize=F.normalize, l1_loss=l1_loss, mse_loss=F.mse_loss)

    def torch_module(self):
        """`torch` for the statements: the real module with randperm recorded (everything else passes through)"""
        rec = self

        class T:
            def __getattr__(self, name):
                return getattr(torch, name)

            @staticmethod
            def randperm(n, *a, **k):
                p = torch.randperm(n, *a, **k)
                rec.perms.append(p.clone())
                return p
        return T()


def run_statements(sc, dtype, conf, decisions_f32):
    """exec lines FIRST..LAST of the reference's main() on the capture `sc` in `dtype`; returns the namespace and the recorder"""
    src = open(os.path.join(REF, "mesh_sfs_optim.py")).read().splitlines()
    block = textwrap.dedent("\n".join(src[FIRST - 1:LAST]))
    rec = Recorder()
    c = lambda t: t.to(dtype)

    class Bar:                                               # tqdm(range(n)) with set_description
        def __init__(self, it):
            self.it = it

        def __iter__(self):
            return iter(self.it)

        def set_description(self, s):
            pass
    ns = dict(dr=oracle, get_normals=mu.get_normals, get_radiance=mu.get_radiance, get_matrix=mu.get_matrix,
              laplacian_smoothing=(mu.laplacian_smoothing if dtype == torch.float32 else oracle.laplacian_smoothing), Adam=rec.adam(), F=rec.functional(), torch=rec.torch_module(), np=np,
              tqdm=Bar, vertices=c(sc["vertices"]), faces=sc["faces"], imgs=c(sc["img"]), grayimgs=c(sc["gray"]), masks=c(sc["mask"]),
              w2cs=c(sc["w2c"]), projs=c(sc["proj"]), resolution=(sc["H"], sc["W"]), **conf)
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self           # the statements say .cuda(): this container has no GPU
    torch.manual_seed(conf["seed"])
    try:
        if decisions_f32:
            with oracle.f32_decisions():
                exec(compile(block, "mesh_sfs_optim.py[%d:%d]" % (FIRST, LAST), "exec"), ns)
        else:
            exec(compile(block, "mesh_sfs_optim.py[%d:%d]" % (FIRST, LAST), "exec"), ns)
    finally:
        torch.Tensor.cuda = saved
    return ns, rec


def main():
    oracle.lib()
    sc = capture()
    conf = dict(degree=3, batch=3, epoch_albedo=2, epoch_sfs=2, sfs_weight=50.0, lap_weight=2000.0, albedo_weight=1.0, mask_weight=1000.0,
                edge_weight=500000.0, delta_weight=50000.0, lr=1e-4, albedo_lr=5e-3, sh_lr=5e-3, seed=0)
    g = {"conf_keys": np.array(sorted(conf)), "conf_values": np.array([float(conf[k]) for k in sorted(conf)])}
    for k in ("vertices", "faces", "img", "gray", "mask", "w2c", "proj"):
        g["in_" + k] = sc[k].numpy()
    # ---- (1) the statements in fp32 (what the script computes with): SH initialisation outputs per degree
    for degree in (3, 2, 1):
        ns, rec = run_statements(sc, torch.float32, dict(conf, degree=degree, epoch_albedo=0, epoch_sfs=0), False)
        sfx = "" if degree == 3 else "_deg%d" % degree
        g["init_sh_coeffs" + sfx] = ns["tmp_sh_coeffs"].detach().numpy()              # the per-view fits before any optimisation (:170)
        g["init_sh_global" + sfx] = ns["sh_coeff"].detach().numpy()
        g["init_albedo_mean" + sfx] = ns["albedo_mean"].detach().numpy()
        if degree == 3:
            g["init_valid_masks"] = ns["valid_masks"].detach().numpy()
            g["init_edge_length_mean"] = np.float64(ns["edge_length_mean"])
            assert tuple(ns["albedo"].shape) == (1, sc["V"], 3) and ns["albedo"].requires_grad
    # ---- (2) the run in fp64 on the fp32 twin's rasterizer decisions (the reference for value tolerances).  batch = 3 of 4 views: a
    # ragged last batch in every epoch.  First stage 1 alone (its end state is the start state of the stage-2 comparison) ...
    ns, rec = run_statements(sc, torch.float64, dict(conf, epoch_sfs=0), True)
    g["s1_perms"] = torch.stack(rec.perms).numpy()
    g["s1_l1"] = np.array(rec.l1)                                                         # F.l1_loss of every batch, in order
    g["s1_first_galbedo"], g["s1_first_gsh"] = (t.numpy() for t in rec.first_grads[0])
    g["s1_albedo"], g["s1_sh_coeffs"] = ns["albedo"].detach().numpy(), ns["sh_coeffs"].detach().numpy()
    g["s1_steps"] = np.int64(rec.steps[0])
    n_s1 = len(rec.l1)
    # ... then the same stage 1 followed by TWO epochs of stage 2 (two batches each: 3 views + 1 view); the statements divide
    # albedo_weight by 10000 when the epoch index reaches epoch_sfs // 2 (:250-251): the first epoch runs at the full weight, the second
    # at the reduced one
    ns, rec = run_statements(sc, torch.float64, dict(conf, epoch_sfs=2), True)
    assert np.array_equal(torch.stack(rec.perms[:2]).numpy(), g["s1_perms"]) and np.allclose(rec.l1[:n_s1], g["s1_l1"], rtol=1e-6, atol=0)    # (threaded index_add: last bits)
    g["s2_perms"] = torch.stack(rec.perms[2:]).numpy()
    g["s2_l1"] = np.array(rec.l1[n_s1:])
    gd, ga, gs = rec.first_grads[1]
    assert gs is None or float(gs.abs().max()) == 0.0                                     # sh_coeffs are frozen in stage 2 (:243)
    g["s2_first_gdelta"], g["s2_first_galbedo"] = gd.numpy(), ga.numpy()
    g["s2_delta"], g["s2_albedo"] = ns["delta"].detach().numpy(), ns["albedo"].detach().numpy()
    g["s2_steps"] = np.int64(rec.steps[1])
    # the six terms and the total the statements leave behind: those of the LAST batch (the second epoch's single view, at the reduced albedo weight)
    g["s2_last_losses"] = np.array([float(ns[k]) for k in ("sfs_loss", "lap_loss", "albedo_loss", "mask_loss", "edge_loss", "delta_loss", "loss")])
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "%.0f KB" % (os.path.getsize(OUT) / 1024))
    for k in sorted(g):
        v = np.asarray(g[k])
        print("  %-26s %-18s %s" % (k, v.shape, v.dtype))


if __name__ == "__main__":
    main()

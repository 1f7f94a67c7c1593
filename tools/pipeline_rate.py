"""Wall time of fmhr_amd.pipeline.run_ham by phase on the metric-size workload (48 views x 512^2, sub3-class mesh) with the iteration
counts of conf/demo_sfs.conf (epoch_albedo 150, epoch_sfs 300), one batch per epoch:  python tools/pipeline_rate.py [views] [batch]"""
import sys, time, torch
sys.path.insert(0, '.')
from fmhr_amd import synthetic, pipeline
from fmhr_amd.ham import render_forward
views = int(sys.argv[1]) if len(sys.argv) > 1 else 48
batch = int(sys.argv[2]) if len(sys.argv) > 2 else views
sc = synthetic.scene_from_config("metric", 0, views)
dev = torch.device("cuda:0")
H, W = sc["H"], sc["W"]
faces = sc["faces"].to(dev); w2c = sc["w2c"].contiguous().to(dev); proj = sc["proj"].contiguous().to(dev)
tg = synthetic.make_targets(sc, lambda v, a, s, m: render_forward(faces, v, a, s, w2c, proj, m, H, W), device=dev)
img = tg["img"]
gray = (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).contiguous()
conf = dict(batch=batch, degree=3, epoch_albedo=150, epoch_sfs=300, sfs_weight=50.0, lap_weight=2000.0, albedo_weight=1.0,
            mask_weight=1000.0, edge_weight=500000.0, delta_weight=50000.0, lr=1e-4, albedo_lr=5e-3, sh_lr=5e-3)
vt = sc["vertices"].to(dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipeline.run_ham(vt, faces, img, gray, tg["mask"], w2c, proj, (H, W), conf, log_every=0)
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    print(f"run {rep}: run_ham total {total:.3f} s ({views} views, batch {batch}: 150 stage-1 + 300 stage-2 epochs x {-(-views // batch)} batches)")

#!/usr/bin/env python3
"""Generates tests/golden/demo_ingest.npz: what fmhr_amd.data.get_demo_data (the cv2-free restatement of
/root/reference/get_data.py:49-118) produces on the reference's own capture /root/reference/demo_data/1 — cameras exactly,
images as 1/32-scale thumbnails + checksums — together with the capture's stored intrinsics / extrinsics (param.npz: int_i, ext_i)
and its 3-D hand keypoints, which pin the camera convention independently of this build's decomposition.  Run in the build
container (the reference data does not travel):  python tests/golden/make_demo_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd.data import get_demo_data  # noqa: E402

DATA = "/root/reference/demo_data"
NUM, RES = 16, (1280, 1024)
imgs, grays, masks, w2cs, projs = get_demo_data(DATA, 1, NUM, RES, device="cpu")
cam = np.load(os.path.join(DATA, "1/camera/param.npz"))
thumb = lambda t: t.reshape(NUM, RES[1] // 32, 32, RES[0] // 32, 32, *t.shape[3:]).mean((2, 4)).numpy().astype(np.float32)
out = dict(w2cs=w2cs.numpy(), projs=projs.numpy(), img_thumb=thumb(imgs), gray_thumb=thumb(grays), mask_thumb=thumb(masks),
           img_sum=imgs.double().sum((1, 2, 3)).numpy(), gray_sum=grays.double().sum((1, 2)).numpy(), mask_sum=masks.double().sum((1, 2)).numpy(),
           ints=np.stack([cam["int_%d" % i] for i in range(NUM)]), exts=np.stack([cam["ext_%d" % i] for i in range(NUM)]),
           scale0=cam["scale_mat_0"], keypoints=np.loadtxt(os.path.join(DATA, "1/keypoints_3d_1.xyz")).astype(np.float32),
           res=np.array(RES))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "demo_ingest.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})

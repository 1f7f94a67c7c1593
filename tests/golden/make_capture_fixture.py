#!/usr/bin/env python3
"""Generates tests/golden/demo_capture.npz: the reference's own capture /root/reference/demo_data/1 as the HAM loop sees it
(/root/reference/get_data.py:49-118 through fmhr_amd.data.get_demo_data: 16 views at 1280 x 1024, BGR / 255, background zeroed by the
mask) — the masked image of every view cut to the mask's bounding box (outside it every pixel is exactly 0, so the crop IS the view), as
uint8 (the loader's floats are exactly k / 255), with the boolean mask crop and the box — plus the two MANO template meshes of
/root/reference/mano/TEMPLATE_{RIGHT,LEFT}.obj (778 vertices / 1 538 faces each: the real topology the reference subdivides,
mesh_sfs_optim.py:76-83).  Data only (arrays).  The cameras and keypoints of the same capture are in demo_ingest.npz
(make_demo_fixture.py).  The fitted MANO mesh of the capture (input_mesh_dire/1.obj + 1.pt) is among the reference's missing large
blobs: tests/test_gpu_pipeline.py places the templates on the capture's 3-D keypoints instead.

    python tests/golden/make_capture_fixture.py          (build container only: the reference data does not travel)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fmhr_amd.data import get_demo_data, load_obj  # noqa: E402

DATA = "/root/reference/demo_data"
NUM, RES = 16, (1280, 1024)
imgs, grays, masks, w2cs, projs = get_demo_data(DATA, 1, NUM, RES, device="cpu")
out = {"res": np.array(RES), "boxes": np.zeros((NUM, 4), np.int32)}
for i in range(NUM):
    m = masks[i].numpy() > 0
    ys, xs = np.where(m)
    y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
    im = imgs[i].numpy()
    im8 = np.round(im[y0:y1, x0:x1] * 255).astype(np.uint8)
    assert np.abs(im[y0:y1, x0:x1] - im8.astype(np.float32) / np.float32(255.)).max() < 1e-7        # the loader's floats are k / 255
    rest = im.copy()
    rest[y0:y1, x0:x1] = 0
    assert float(np.abs(rest).max()) == 0.0                                                        # nothing outside the box
    # the gray image is a function of the 8-bit colours (COLOR_BGR2GRAY on the uint8 image, get_data.py:83): not stored
    g8 = np.round(0.114 * im8[..., 0].astype(np.float64) + 0.587 * im8[..., 1] + 0.299 * im8[..., 2]).astype(np.uint8)
    assert np.abs(grays[i].numpy()[y0:y1, x0:x1] - g8.astype(np.float32) / np.float32(255.)).max() < 1e-7
    out["boxes"][i] = (y0, y1, x0, x1)
    out["img%02d" % i] = im8
    out["mask%02d" % i] = m[y0:y1, x0:x1]
for side in ("RIGHT", "LEFT"):
    v, f = load_obj("/root/reference/mano/TEMPLATE_%s.obj" % side)
    assert v.shape == (778, 3) and f.shape == (1538, 3)
    out["template_%s_v" % side.lower()], out["template_%s_f" % side.lower()] = v.astype(np.float32), f.astype(np.int32)
path = os.path.join(ROOT, "tests", "golden", "demo_capture.npz")
np.savez_compressed(path, **out)
print(path, "%.2f MB" % (os.path.getsize(path) / 1e6), "coverage %.4f" % float(masks.mean()))

"""fmhr_amd.data — camera / image ingest for the HAM loop without cv2 / pyhocon / trimesh.

* parse_conf        — the flat `key = value` files of /root/reference/conf/*.conf (read there with pyhocon, mesh_sfs_optim.py:31-65)
* load_K_Rt_from_P  — /root/reference/models/utils.py:29-47 (cv2.decomposeProjectionMatrix replaced by an RQ decomposition)
* clip_projection   — the proj edit of /root/reference/get_data.py:64-73
* get_demo_data     — /root/reference/get_data.py:49-118 (PNG via PIL; bilinear / nearest resize like cv2's defaults)
* save_obj_mesh_with_color / save_obj — the writers of mesh_sfs_optim.py:19-28,323-335 (row N4)
"""
import os

import numpy as np
import torch


def parse_conf(text):
    """flat HOCON subset: `key = value` lines, # or // comments; ints / floats / bools / strings typed by content."""
    out = {}
    for line in text.splitlines():
        line = line.split('#')[0].split('//')[0].strip()
        if not line or '=' not in line:
            continue
        k, v = [s.strip() for s in line.split('=', 1)]
        v = v.strip('"').strip("'")
        if v.lower() in ("true", "false"):
            out[k] = v.lower() == "true"
            continue
        try:
            out[k] = int(v)
        except ValueError:
            try:
                out[k] = float(v)
            except ValueError:
                out[k] = v
    return out


def _rq3(M):
    """M = R_upper @ Q with Q orthonormal, diag(R_upper) > 0 (what cv2.decomposeProjectionMatrix returns as K, R)."""
    Pm = np.flipud(np.eye(3))
    q, r = np.linalg.qr((Pm @ M).T)
    R_ = Pm @ r.T @ Pm
    Q_ = Pm @ q.T
    S = np.diag(np.sign(np.diag(R_)) + (np.diag(R_) == 0))
    return R_ @ S, S @ Q_


def load_K_Rt_from_P(P):
    """P [3,4] -> (intrinsics [4,4] with K / K[2,2] in the top-left block, world-to-camera pose [4,4])."""
    P = np.asarray(P, dtype=np.float64)
    K, R = _rq3(P[:, :3])
    if np.linalg.det(R) < 0:          # keep a proper rotation (P is defined up to sign)
        K, R = _rq3(-P[:, :3])
        P = -P
    C = -np.linalg.solve(P[:, :3], P[:, 3])              # camera centre: P [C;1] = 0  (cv2 returns it homogeneous)
    K = K / K[2, 2]
    intr = np.eye(4)
    intr[:3, :3] = K
    pose = np.eye(4)
    pose[:3, :3] = R.T
    pose[:3, 3] = C
    return intr, np.linalg.inv(pose)


def clip_projection(intrinsics, res):
    """the 4x4 clip projection of get_data.py:64-73 for res = (w, h): only these 8 entries are rewritten, as there."""
    proj = np.array(intrinsics, dtype=np.float64, copy=True)
    proj[0, 0] = proj[0, 0] / (res[0] / 2.)
    proj[0, 2] = proj[0, 2] / (res[0] / 2.) - 1.
    proj[1, 1] = proj[1, 1] / (res[1] / 2.)
    proj[1, 2] = proj[1, 2] / (res[1] / 2.) - 1.
    proj[2, 2] = 0.
    proj[2, 3] = -0.1
    proj[3, 2] = 1.
    proj[3, 3] = 0.
    return proj


def _load_png(path):
    from PIL import Image
    return Image.open(path)


def get_demo_data(data_path, scan_id, num, res=(1280, 1024), with_mask=True, device="cuda"):
    """-> imgs [num,h,w,3] (BGR order like cv2.imread, /255), grayimgs, masks, w2cs, projs (both transposed for the row-vector
    convention, get_data.py:96-97)."""
    from PIL import Image
    cam = np.load(os.path.join(data_path, '%d/camera/param.npz' % scan_id))
    imgs, grays, masks, w2cs, projs = [], [], [], [], []
    for i in range(num):
        P = cam['world_mat_%d' % i].astype(np.float32) @ cam['scale_mat_%d' % i].astype(np.float32)
        intr, w2c = load_K_Rt_from_P(P[:3])
        projs.append(clip_projection(intr, res).astype(np.float32))
        w2cs.append(w2c.astype(np.float32))
        img = np.asarray(_load_png(os.path.join(data_path, '%d/img/%02d.png' % (scan_id, i))).convert('RGB'))[:, :, ::-1].copy()
        m = np.asarray(_load_png(os.path.join(data_path, '%d/mask/%02d.png' % (scan_id, i))).convert('RGB'))[:, :, 0]
        m = (m > 127.5).astype(np.float32)
        if with_mask:
            img[m == 0] = 0
        gray = np.round(0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).astype(np.uint8)   # COLOR_BGR2GRAY
        img = np.asarray(Image.fromarray(img).resize(res, Image.BILINEAR))
        gray = np.asarray(Image.fromarray(gray).resize(res, Image.BILINEAR))
        m = np.asarray(Image.fromarray(m).resize(res, Image.NEAREST))
        imgs.append(torch.from_numpy(img / 255.).float())
        grays.append(torch.from_numpy(gray / 255.).float())
        masks.append(torch.from_numpy(np.array(m)).float())
    t = lambda l: torch.from_numpy(np.stack(l)).permute(0, 2, 1).contiguous().to(device)
    return (torch.stack(imgs).to(device), torch.stack(grays).to(device), torch.stack(masks).to(device), t(w2cs), t(projs))


def save_obj(mesh_path, verts, faces):
    with open(mesh_path, 'w') as f:
        for v in verts:
            f.write('v %.8f %.8f %.8f\n' % (v[0], v[1], v[2]))
        for t in faces:
            f.write('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1))


def save_obj_mesh_with_color(mesh_path, verts, faces, colors):
    """mesh_sfs_optim.py:19-28: `v x y z r g b` with %.4f, faces 1-based with the winding flipped (f0, f2, f1)."""
    with open(mesh_path, 'w') as f:
        for idx, v in enumerate(verts):
            c = colors[idx]
            f.write('v %.4f %.4f %.4f %.4f %.4f %.4f\n' % (v[0], v[1], v[2], c[0], c[1], c[2]))
        for t in faces:
            f.write('f %d %d %d\n' % (t[0] + 1, t[2] + 1, t[1] + 1))


def save_rerender_png(path, img):
    """mesh_sfs_optim.py:342-343: `cv2.imwrite(path, (rendered * 255).astype(np.int32))` for one [H,W,3] image in the loader's
    BGR channel order (cv2.imread, get_data.py:77).  cv2 saturates to 8 bits and writes the BGR array as it stands, i.e. the
    file shows the original colours: PIL gets the RGB view of the same pixels."""
    from PIL import Image
    a = np.asarray(img, dtype=np.float64) * 255.0
    a = np.clip(a.astype(np.int32), 0, 255).astype(np.uint8)             # .astype(np.int32) truncates toward zero, as there
    Image.fromarray(np.ascontiguousarray(a[:, :, ::-1])).save(path)


def load_obj(path):
    vs, fs = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == 'v':
            vs.append([float(x) for x in p[1:4]])
        elif p[0] == 'f':
            fs.append([int(x.split('/')[0]) - 1 for x in p[1:4]])
    return np.asarray(vs, dtype=np.float32), np.asarray(fs, dtype=np.int32)

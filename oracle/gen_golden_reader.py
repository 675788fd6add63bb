#!/usr/bin/env python3
"""Golden vectors that pin the remaining "small" rows against the REFERENCE itself (imported from /root/reference, build
container only; nothing of the reference travels -- only inputs and outputs are stored):

  * P10b  utils/sh_util.py::eval_sh, degrees 0-4                                    -> tests/golden/small.npz  sh_*
  * P9    model/network/embedder.py::get_embedder(8, include_input=False)           -> tests/golden/small.npz  pe_*
  * next-2 the dataset readers dataloader/dataloader.py:36-234 and dataloader/dataloaderSR.py (MultiView_ImgDataset.__getitem__,
          make_render_cond_): run on a synthetic dataset in the reference's on-disk layout (havatar_amd.synth.write_dataset)
          in the modes test / val / train (random rays and the 64x64 patch), with numpy's RNG seeded            -> reader_*

How the readers are imported without OpenCV: `cv2` is absent from this image.  The fixtures are taken at down_sample = 1.0 and
cond_render_res = 256, where the readers call exactly two cv2 functions -- imread and cvtColor(BGR2RGB) -- and both are exact
(PNG is lossless; the stand-in decodes with PIL and reverses the channel order twice).  resize / erode / getStructuringElement
RAISE in the stand-in, so no fixture can depend on this repo's restatement of cv2's interpolation arithmetic; that arithmetic
(INTER_AREA for down_sample < 1, INTER_LINEAR) stays pinned only by hand-computed values (tests/test_harness.py) and is
declared so in DESIGN.md.
"""
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from gen_golden import import_reference  # noqa: E402


def _cv2_standin():
    from PIL import Image
    m = types.ModuleType("cv2")
    m.COLOR_BGR2RGB, m.INTER_AREA, m.INTER_LINEAR, m.MORPH_RECT = 4, 3, 1, 0

    def imread(path):
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])      # BGR, like OpenCV

    def cvtColor(img, code):
        assert code == m.COLOR_BGR2RGB
        return np.ascontiguousarray(img[:, :, ::-1])

    def refuse(*a, **k):
        raise RuntimeError("this fixture must not depend on a restatement of OpenCV's resampling arithmetic")

    m.imread, m.cvtColor, m.resize, m.erode, m.getStructuringElement = imread, cvtColor, refuse, refuse, refuse
    return m


def reader_cases():
    """(tag, reader module, mode, img_res, views, patch_rgb, item indices, numpy seed)"""
    return [("t1_test", "dataloader", "test", 24, ("0", "8", "1"), False, (0, 3), 0),
            ("t1_val", "dataloader", "val", 24, ("0", "8", "1"), False, (1,), 0),
            ("t1_train_rays", "dataloader", "train", 24, ("0", "8", "1"), False, (0, 2), 123),
            ("t1_train_patch", "dataloader", "train", 72, ("0",), True, (1,), 321),
            ("sr_test", "dataloaderSR", "test", 24, ("0", "8"), False, (0, 3), 0)]


def reader_options(patch_rgb):
    from havatar_amd import synth
    from havatar_amd.utils.cfgnode import CfgNode
    cfg = synth.harness_config(render_size=24, img_res=24, rays=48)
    cfg["experiment"]["patch_rgb"] = patch_rgb
    cfg["dataset"]["cond_render_res"] = 256
    return CfgNode(cfg)


def flatten_item(prefix, idx, item, out):
    for k, v in item.items():
        a = np.asarray(v.numpy() if hasattr(v, "numpy") else v)
        if "render_cond" in k:                    # [256,256,7]: keep a strided slice + checksums (the PNG decode is what is pinned)
            out["%s_%d_%s_slice" % (prefix, idx, k)] = a[::16, ::16].copy()
            out["%s_%d_%s_cks" % (prefix, idx, k)] = np.array([a.astype(np.float64).sum(), (a.astype(np.float64) ** 2).sum()])
        else:
            out["%s_%d_%s" % (prefix, idx, k)] = a


def main():
    from havatar_amd import synth
    torch, _, _ = import_reference()
    out = {}
    # ---- P10b eval_sh ------------------------------------------------------------------------------------------------
    from utils.sh_util import eval_sh
    sh = synth.normal((7, 3, 25), 6)
    d = synth.normal((7, 3), 7)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out["sh_coeffs"], out["sh_dirs"] = sh, d
    for deg in range(5):
        K = (deg + 1) ** 2
        out["sh_deg%d" % deg] = eval_sh(deg, torch.from_numpy(sh[..., :K].copy()), torch.from_numpy(d)).numpy()
        out["sh64_deg%d" % deg] = eval_sh(deg, torch.from_numpy(sh[..., :K].astype(np.float64)), torch.from_numpy(d.astype(np.float64))).numpy()
    # ---- P9 embedder ---------------------------------------------------------------------------------------------------
    from model.network.embedder import get_embedder
    emb, dim = get_embedder(8, input_dims=3, include_input=False)
    x = synth.uniform((50, 3), 5, -1.6, 1.6)
    out["pe_x"], out["pe_out"], out["pe_dim"] = x, emb(torch.from_numpy(x)).numpy(), np.array(dim)
    # ---- next-2 dataset readers ----------------------------------------------------------------------------------------
    sys.modules["cv2"] = _cv2_standin()
    import importlib
    for tag, modname, mode, res, views, patch_rgb, idxs, seed in reader_cases():
        mod = importlib.import_module("dataloader." + modname)
        tmp = tempfile.mkdtemp()
        split = synth.write_dataset(tmp, n_frames=2, img_res=res, views=views)
        ds = mod.MultiView_ImgDataset(split, mode, reader_options(patch_rgb), down_sample=1.0, white_bg=True)
        out["reader_%s_len" % tag] = np.array(len(ds))
        for i in idxs:
            np.random.seed(seed + i)
            idx, item = ds[i]
            assert idx == i
            flatten_item("reader_" + tag, i, item, out)
    path = os.path.join(REPO, "tests", "golden", "small.npz")
    np.savez_compressed(path, **out)
    print("small.npz", os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()

"""Shared part of the two dataset readers: split-file parsing, cameras, rays, near/far, backgrounds, 3DMM condition images,
head pose.  Everything a frame needs besides its photographs."""
import copy
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import data_util, imgio

COND_VIEWS = ("front", "left", "right")


def worker_init_fn(worker_id):
    """numpy seed per DataLoader worker (dataloader.py:17-20)."""
    seed = torch.utils.data.get_worker_info().seed % 1000000000
    np.random.seed(seed + worker_id)


def make_render_cond_(normal_path, render_path, res):
    """[res,res,7] = rgb render, normal map, mask(normal != 0), all /255 (dataloader.py:224-234)."""
    normal = imgio.resize_linear(imgio.imread_rgb(normal_path), res)
    mask = (normal.astype(np.float64).sum(-1) > 0).astype(np.float32)            # ||n|| > 0  <=>  any channel > 0
    render = imgio.resize_linear(imgio.imread_rgb(render_path), res)
    return torch.from_numpy(np.concatenate([render.astype(np.float32) / 255.0, normal.astype(np.float32) / 255.0,
                                            mask[:, :, None]], axis=-1))


class SplitFileDataset(Dataset):
    """One item = one (frame, view) of a split file (`sv_v31_all.json` layout):
    {"img_res", "mutiview_intr_ls" [[fx,fy,cx/W,cy/H],..], "bg_path"?, "frames": [{"fidx", "inst_dir", "head_transformation" [4,4],
      "mutiview_info_ls": [{"view_name", "transform_matrix", "transform_matrix_ori", "cam_K"?, "file_path", "mask_path"}]}]}"""
    skip_view = None

    def __init__(self, split_file, mode, options, down_sample=1.0, white_bg=True):
        super().__init__()
        assert mode in ["train", "val", "test"]
        assert os.path.exists(split_file), split_file
        self.mode, self.options, self.down_sample, self.white_bg = mode, options, down_sample, white_bg
        with open(split_file) as f:
            meta = json.loads(f.read())
        self.img_w = self.img_h = meta["img_res"]
        self.mv_intrinsics = np.asarray(meta["mutiview_intr_ls"], dtype=np.float32)
        if down_sample < 1:
            self.mv_intrinsics[:, :2] = self.mv_intrinsics[:, :2] * down_sample
            self.img_w, self.img_h = int(self.img_w * down_sample), int(self.img_h * down_sample)
        self.view_num = self.mv_intrinsics.shape[0]
        self.load_background(meta.get("bg_path"), white_bg)
        self.frames = []
        for fd in meta["frames"]:
            for vidx, view in enumerate(fd["mutiview_info_ls"]):
                if self.skip_view is not None and view["view_name"] == self.skip_view:
                    continue
                item = copy.deepcopy(fd)
                item["vidx"] = vidx
                self.frames.append(item)
        self.frames.sort(key=lambda x: x["fidx"])
        yy, xx = torch.meshgrid(torch.arange(self.img_h), torch.arange(self.img_w), indexing="ij")
        self.coords_yx = torch.stack([yy, xx], -1).reshape(-1, 2)      # row-major pixels, (y, x)
        self.coords_yx_np = self.coords_yx.numpy()

    def load_background(self, bg_paths, white_bg):
        self.bgs = []
        if white_bg:
            self.bgs = [torch.ones((self.img_h, self.img_w, 3), dtype=torch.float32) for _ in range(self.view_num)]
            return
        for path in bg_paths:
            bg = imgio.imread_rgb(path)
            if self.down_sample < 1:
                bg = imgio.resize_area(bg, self.down_sample)
            self.bgs.append(torch.from_numpy(bg.astype(np.float32)) / 255)

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, idx):
        return idx, self.load_data(self.frames[idx])

    # ---- pieces of load_data shared by both readers -------------------------------------------------------------------
    def view_camera(self, frame_dict):
        view_idx = frame_dict["vidx"]
        view = frame_dict["mutiview_info_ls"][view_idx]
        pose = torch.from_numpy(np.asarray(view["transform_matrix"], dtype=np.float32))
        if "cam_K" in view:
            cam_K = np.asarray(view["cam_K"], dtype=np.float32)
            if self.down_sample < 1:
                cam_K[:2] = cam_K[:2] * self.down_sample
        else:
            cam_K = self.mv_intrinsics[view_idx]
        return view_idx, view, cam_K, pose

    def camera_record(self, view, cam_K, pose):
        """[18] = intr(4) | c2w[3,4](12) | near | far : everything hav_gen_rays needs to rebuild this view's full-frame ray table on
        the device (test-mode readers with `device_rays=True`; white background only)."""
        cam_t = torch.from_numpy(np.asarray(view["transform_matrix_ori"], dtype=np.float32))[:3, -1]
        dist = float(torch.norm(cam_t))
        near = dist + self.options.dataset.near * self.options.dataset.length
        far = dist + self.options.dataset.far * self.options.dataset.length
        return torch.cat([torch.as_tensor(np.asarray(cam_K, np.float32)[:4]), pose[:3, :4].reshape(-1).float(),
                          torch.tensor([near, far], dtype=torch.float32)])

    def rays_for(self, view_idx, view, cam_K, pose, select_inds, ray_m, with_mask):
        """[n, 11 (+1)] = origin3, dir3, near, far, background3 (, mask) -- dataloader.py:168-185."""
        ray_o, ray_d = data_util.get_rays(self.img_h, self.img_w, cam_K, pose[:3, :4], normalize=True)
        ys, xs = select_inds[:, 0], select_inds[:, 1]
        ray_o, ray_d = ray_o[ys, xs, :], ray_d[ys, xs, :]
        ray_bg = self.bgs[view_idx][ys, xs, :]
        cam_t = torch.from_numpy(np.asarray(view["transform_matrix_ori"], dtype=np.float32))[:3, -1]
        dist = torch.norm(cam_t.expand(ray_d.shape), dim=-1, keepdim=True)
        ones = torch.ones_like(ray_d[..., :1])
        near = (dist + self.options.dataset.near * self.options.dataset.length) * ones
        far = (dist + self.options.dataset.far * self.options.dataset.length) * ones
        cols = [ray_o, ray_d, near, far, ray_bg] + ([ray_m] if with_mask else [])
        return torch.cat(cols, dim=1)

    def add_conditions(self, data_dict, frame_dict):
        """3 orthographic 3DMM renders + the head pose (dataloader.py:205-220)."""
        res = self.options.dataset.cond_render_res
        for v in COND_VIEWS:
            data_dict[v + "_render_cond"] = make_render_cond_(
                os.path.join(frame_dict["inst_dir"], "ortho_%s_normal_256_baseGama.png" % v),
                os.path.join(frame_dict["inst_dir"], "ortho_%s_render_256_baseGama.png" % v), res)
        T = np.asarray(frame_dict["head_transformation"]).astype(np.float32)[:3]          # right-multiplied convention
        rotation, translation = T.T[:3, :3], T.T[-1:]
        data_dict["inv_head_T"] = torch.from_numpy(np.concatenate([np.linalg.inv(rotation), -translation], 0))   # [4,3]
        return data_dict

"""Stage-one dataset reader (reference: dataloader/dataloader.py): random rays / patches for training, full frames for
validation and test.  Same class names, constructor arguments and item layout."""
import numpy as np
import torch
from torch.utils.data import DataLoader

from . import data_util, dist_util, imgio
from ._base import SplitFileDataset, make_render_cond_, worker_init_fn  # noqa: F401


def random_gen_coords_mask(mask, samples, p=0.9):
    """`samples` distinct pixel indices, foreground-weighted (dataloader.py:23-33)."""
    H, W = mask.shape
    pmap = data_util.make_ray_importance_sampling_map(mask, p=p)
    return np.random.choice(H * W, size=samples, replace=False, p=pmap.reshape(-1))


class MultiView_ImgDataset(SplitFileDataset):
    skip_view = "8"                                  # dataloader.py:63

    def __init__(self, split_file, mode, options, down_sample=1.0, white_bg=True):
        super().__init__(split_file, mode, options, down_sample, white_bg)
        self.num_random_rays = options.dataset.num_random_rays
        self.patch_rgb = options.experiment.patch_rgb
        self.patch_size, self.n_patches = (64, 1) if self.patch_rgb else (11, 5)
        self.mask_thresh = 127.5

    def subsample_patches(self, patch_size, n_patches, mask=None, p=1.0, erode=True):
        """Pixel coordinates (y,x) of n_patches square patches whose centres are drawn uniformly (mask None) or from the
        (optionally eroded) mask interior (dataloader.py:93-121) -> [n_patches * patch_size^2, 2]."""
        H, W, h = self.img_h, self.img_w, patch_size // 2
        if mask is None:
            x0 = np.random.randint(h, W - h, size=(n_patches, 1, 1))
            y0 = np.random.randint(h, H - h, size=(n_patches, 1, 1))
            yx0 = np.concatenate([y0, x0], axis=-1)
        else:
            m = (mask * 255).astype(np.uint8)
            if erode:
                m = imgio.erode_rect(m, patch_size)
            inner = np.zeros_like(mask)
            inner[h:H - h, h:W - h] = m[h:H - h, h:W - h]
            pmap = data_util.make_ray_importance_sampling_map(inner, p=p)
            sel = np.random.choice(H * W, size=n_patches, replace=False, p=pmap.reshape(-1))
            yx0 = self.coords_yx_np[sel][:, np.newaxis]
        offs = np.stack(np.meshgrid(np.arange(patch_size) - h, np.arange(patch_size) - h, indexing="xy"), axis=-1).reshape(1, -1, 2)
        return (yx0 + offs).reshape(-1, 2)

    def load_data(self, frame_dict):
        view_idx, view, cam_K, pose = self.view_camera(frame_dict)
        mask = mask_tensor = None
        if self.mode != "test":
            mask = imgio.imread_rgb(view["mask_path"])
            if self.down_sample < 1:
                mask = imgio.resize_area(mask, self.down_sample)
            assert (self.img_h, self.img_w) == mask.shape[:2]
            thr = self.mask_thresh if type(self.mask_thresh) is float else self.mask_thresh[view["view_name"]]
            mask = (mask[:, :, 0] > thr).astype(np.float32)
            mask_tensor = torch.from_numpy(mask).unsqueeze(-1)
        if self.mode == "train":
            if self.patch_rgb:
                select_inds = self.subsample_patches(self.patch_size, self.n_patches, mask, p=1.0, erode=False)
            else:
                select_inds = self.coords_yx[random_gen_coords_mask(mask, samples=self.num_random_rays, p=0.95)]
        else:
            select_inds = self.coords_yx
        ray_m = mask_tensor[select_inds[:, 0], select_inds[:, 1]] if mask_tensor is not None else None
        mv_rays = self.rays_for(view_idx, view, cam_K, pose, select_inds, ray_m, with_mask=self.mode == "train")
        if self.mode == "test":
            data_dict = {"fidx": frame_dict["fidx"], "vidx": [int(view["view_name"])], "mv_rays": mv_rays}
        else:
            img = imgio.imread_rgb(view["file_path"])
            if self.down_sample < 1:
                img = imgio.resize_area(img, self.down_sample)
            assert (self.img_h, self.img_w) == img.shape[:2]
            img_tensor = torch.from_numpy((np.array(img) / 255.0).astype(np.float32))
            img_tensor = img_tensor * mask_tensor + self.bgs[view_idx] * (1.0 - mask_tensor)
            data_dict = {"mv_rays_gt_color": img_tensor[select_inds[:, 0], select_inds[:, 1], :], "mv_rays": mv_rays}
        return self.add_conditions(data_dict, frame_dict)


class Loader(DataLoader):
    """dataloader.py:236-250 (pinned memory, drop_last, sampler chosen by dist_util.data_sampler)."""

    def __init__(self, split_file, options, mode="train", batch_size=4, num_workers=0, down_sample=1.0, distributed=False,
                 white_bg=True, shuffle=None):
        self.dataset = MultiView_ImgDataset(split_file, mode, options, down_sample, white_bg=white_bg)
        self.batch_size = batch_size
        if shuffle is None:
            shuffle = mode == "train"
        self.sampler = dist_util.data_sampler(self.dataset, shuffle=shuffle, distributed=distributed)
        super().__init__(self.dataset, batch_size=batch_size, sampler=self.sampler, num_workers=num_workers,
                         worker_init_fn=worker_init_fn, pin_memory=torch.cuda.is_available(), drop_last=True)

"""Stage-two / reenactment dataset reader (reference: dataloader/dataloaderSR.py): always full frames at the (down-sampled)
render resolution; ground truth at the ORIGINAL resolution; the mask is thresholded first and area-averaged afterwards."""
import numpy as np
import torch
from torch.utils.data import DataLoader

from . import dist_util, imgio
from ._base import SplitFileDataset, make_render_cond_, worker_init_fn  # noqa: F401


class MultiView_ImgDataset(SplitFileDataset):
    skip_view = None                                 # every view is kept (dataloaderSR.py:44-49)
    device_rays = False                              # test mode: emit the 18-float camera record instead of the [N,11] ray table

    def load_data(self, frame_dict):
        view_idx, view, cam_K, pose = self.view_camera(frame_dict)
        if self.mode == "test" and self.device_rays and self.white_bg:
            data_dict = {"fidx": frame_dict["fidx"], "vidx": [int(view["view_name"])], "camera": self.camera_record(view, cam_K, pose)}
            return self.add_conditions(data_dict, frame_dict)
        select_inds = self.coords_yx
        mask = ray_m = None
        if self.mode == "train":
            mask = (imgio.imread_rgb(view["mask_path"])[:, :, 0] > 127).astype(np.float32)
            mask_ds = imgio.resize_area(mask, self.down_sample) if self.down_sample < 1 else mask
            assert (self.img_h, self.img_w) == mask_ds.shape[:2]
            ray_m = torch.from_numpy(mask_ds)[select_inds[:, 0], select_inds[:, 1]].unsqueeze(-1)
        mv_rays = self.rays_for(view_idx, view, cam_K, pose, select_inds, ray_m, with_mask=ray_m is not None)
        if self.mode == "test":
            data_dict = {"fidx": frame_dict["fidx"], "vidx": [int(view["view_name"])], "mv_rays": mv_rays}
        else:
            img = imgio.imread_rgb(view["file_path"])
            if self.white_bg:
                # dataloaderSR.py:127 indexes with the training mask; in 'val' mode the reference has no mask at this point
                if mask is None:
                    raise RuntimeError("dataloaderSR: ground-truth compositing needs the mask, which is only read in 'train' mode")
                img[mask == 0] = 255
            data_dict = {"mv_rays_gt_color": torch.from_numpy((np.array(img) / 255.0).astype(np.float32)).reshape(-1, 3),
                         "mv_rays": mv_rays}
        return self.add_conditions(data_dict, frame_dict)


class Loader(DataLoader):
    """dataloaderSR.py:171-183."""

    def __init__(self, split_file, options, mode="train", batch_size=4, num_workers=0, down_sample=1.0, distributed=False,
                 white_bg=True, shuffle=None):
        self.dataset = MultiView_ImgDataset(split_file, mode, options, down_sample, white_bg=white_bg)
        self.batch_size = batch_size
        if shuffle is None:
            shuffle = mode == "train"
        self.sampler = dist_util.data_sampler(self.dataset, shuffle=shuffle, distributed=distributed)
        super().__init__(self.dataset, batch_size=batch_size, sampler=self.sampler, num_workers=num_workers,
                         worker_init_fn=worker_init_fn, pin_memory=torch.cuda.is_available(), drop_last=True)

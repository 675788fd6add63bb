"""Trainer: owns latent codes, the tri-plane NeRF and the skinning field; renders rays
(reference model/nerf_trainer.py:11-201 -- same constructor, attributes, methods, return tuples and state_dict keys).

Difference: `predict_and_render_radiance` on HIP tensors without autograd is ONE call into libhavatar_hip.so for all the
rays it is given (ray sampling, skinning lookup, tri-plane gather, PE, MLP, compositing, resampling, second pass), and
`nerf_forward` therefore does not chunk (the reference's chunksize loop only bounds activation memory, :66-71).  CPU
tensors and autograd-tracked calls take the PyTorch statement of the same algorithm (the reference runs there too)."""
import os

import numpy as np
import torch
from einops import rearrange

from . import nerf_model
from .Skinning_Field import Deformation_Field_new
from ..render import RayMarcher
from ..utils.nerf_util import sample_pdf, volume_render_radiance_field
from ..utils.training_util import get_minibatches
from ..utils.util import UniformBoxWarp_new, get_box_warp_param


class Trainer(torch.nn.Module):
    def __init__(self, cfg, latent_codes_size=0, freeze_motion=True):
        super().__init__()
        self.cfg = cfg
        e = cfg.experiment
        self.latent_codes = torch.nn.Parameter(torch.zeros(latent_codes_size, e.latent_code_dim)) if latent_codes_size > 0 else None
        self.model_mode = e.model_mode
        latent_code_dim = e.latent_code_dim + (12 if e.cond_pose else 0) + (52 if e.cond_expr else 0)
        bounding = cfg.models.coarse.XYZ_bounding
        self.model_coarse = nerf_model.ConditionalTriplaneNeRFModel_multiRender_split_view(
            XYZ_bounding=bounding, triPlane_feat_dim=64, rgb_feat_dim=3, triplane_res=128, sh_deg=0,
            latent_code_dim=latent_code_dim, cond_c_dim=latent_code_dim - e.latent_code_dim)
        self.render_size, self.gen_size = cfg.models.StyleUnet.inp_size, cfg.models.StyleUnet.out_size
        b = {k: np.asarray(v, dtype=np.float64).copy() for k, v in zip("XYZ", bounding)}
        b["Y"][0] = 0.3 * b["Y"][1]                                     # the skinning box only covers the neck (:29-34)
        scales, trans = get_box_warp_param(b["X"], b["Y"], b["Z"])
        self.headpose_skin_net = Deformation_Field_new(gridwarper=UniformBoxWarp_new(scales=scales, trans=trans))
        if freeze_motion:
            self.headpose_skin_net.requires_grad = False               # an attribute, freezes nothing -- as in the reference (B-8)
        self._marcher = None

    # ---------------------------------------------------------------------------------------------------------------
    def _hip_marcher(self):
        if self._marcher is None:
            gw, sw = self.model_coarse.gridwarper, self.headpose_skin_net.gridwarper
            f = lambda t: t.detach().reshape(3).cpu().tolist()
            self._marcher = RayMarcher(f(gw.scale_factor), f(gw.trans_factor), f(sw.scale_factor), f(sw.trans_factor))
        return self._marcher

    def nerf_forward(self, **inputs):
        cond_c = inputs["inv_head_T"].view(inputs["inv_head_T"].shape[0], -1)
        self.model_coarse.set_conditional_embedding(
            front_render_cond=inputs["front_render_cond"], left_render_cond=inputs["left_render_cond"],
            right_render_cond=inputs["right_render_cond"], latents=inputs["latent_code"], cond_c=cond_c)
        self._planes_dirty = True
        inv_head_T, ray_batch, background_prior = inputs["inv_head_T"], inputs["ray_batch"], inputs["background_prior"]
        mode = inputs["mode"]
        opt = getattr(self.cfg.nerf, mode)
        rd = ray_batch[..., 3:6]
        restore = [rd.shape[:-1] + (-1,), rd.shape[:-1], rd.shape[:-1], rd.shape[:-1]]
        if opt.num_fine > 0:
            restore += restore[:-1]
        if ray_batch.is_cuda:
            # the reference appends the normalised view directions to the rays (:60-65) and never reads them again (the radiance MLP of this
            # model takes no direction; predict_and_render_radiance uses columns 0-7 only): on the device that is a norm, a division and an
            # 11.5 MB concatenation per 512^2 frame between the encoders and the march, for nothing -- the rays go on as they are
            rays = ray_batch
        else:
            viewdirs = rd / rd.norm(p=2, dim=-1).unsqueeze(-1)
            rays = torch.cat((ray_batch, viewdirs), dim=-1)
        if self._use_hip(rays):
            # one launch, no chunk loop.  forward(render_full_img=True) only consumes the fine maps: it declines the coarse pass's
            # composited outputs (they come back as None), which lets the kernel drop their accumulators
            self._decline_coarse = bool(inputs.get("_fine_maps_only", False)) and opt.num_fine > 0
            out = list(self.predict_and_render_radiance(mode, rays, background_prior, inv_head_T=inv_head_T))
            self._decline_coarse = False
        else:
            # the reference's chunk loop (:66-71) only bounds activation memory.  HIP tensors under autograd take all rays of the call as ONE
            # chunk: half the launches of a cfg5 step (4096 rays per frame = two 2048-ray chunks in the reference) and kernels twice
            # the size; 288 GB of HBM hold the [n, 176] field inputs of 0.5 M queries many times over.  CPU tensors chunk as the reference.
            # Bounded all the same (a grad-enabled full image -- 512^2 rays x 112 samples x 176 floats = 20 GB before gradients -- must
            # not be one piece on any device): at most HAVATAR_TRAIN_CHUNK_RAYS rays per chunk over the batch (default 32768 = eight cfg5
            # steps' worth, ~2.6 GB of field inputs), never less than the reference's chunk
            if rays.is_cuda:
                cap = max(int(os.environ.get("HAVATAR_TRAIN_CHUNK_RAYS", "32768")), opt.chunksize)
                chunk = min(rays.shape[1], max(cap // rays.shape[0], 1))
            else:
                chunk = opt.chunksize // rays.shape[0]
            rb = get_minibatches(rays, chunksize=chunk, dim=1)
            bg = get_minibatches(background_prior, chunksize=chunk, dim=1) if background_prior is not None else None
            pred = [self.predict_and_render_radiance(mode, r, None if bg is None else bg[i], inv_head_T=inv_head_T) for i, r in enumerate(rb)]
            out = [torch.cat(im, dim=1) if im[0] is not None else None for im in zip(*pred)]
        if mode == "validation":
            out = [im.view(shape) if im is not None else None for im, shape in zip(out, restore)]
            return tuple(out) if opt.num_fine > 0 else tuple(out + [None, None, None])[:7]
        return tuple(out)

    def forward(self, **data):
        ray_batch, background_prior = data["ray_batch"], data["background_prior"]
        B = ray_batch.shape[0]
        latent_code = self.latent_codes[data["fidx"]] if data["mode"] == "train" else self.latent_codes[0:1]
        latent_code_loss = self._latent_code_loss(latent_code, data["mode"])
        if data["mode"] != "train" and B > 1:
            # batched inference (B frames per call: frames.py / bench.py --workload cfg3 --batch): the one validation code serves every frame
            # of the batch.  (The reference only ever calls this with B = 1 -- its cat([latent, cond_c]) does not broadcast.)
            latent_code = latent_code.expand(B, -1)
        rgb_coarse, _, acc_coarse, weights, rgb_fine, _, acc_fine = self.nerf_forward(
            ray_batch=ray_batch, background_prior=background_prior, latent_code=latent_code, inv_head_T=data["inv_head_T"],
            front_render_cond=data["front_render_cond"], left_render_cond=data["left_render_cond"],
            right_render_cond=data["right_render_cond"], mode=data["mode"], _fine_maps_only=bool(data["render_full_img"]))
        if data["render_full_img"]:
            render = rgb_fine if rgb_fine is not None else rgb_coarse
            mask = acc_fine if acc_fine is not None else acc_coarse
            render = render.reshape(B, self.render_size, self.render_size, -1).permute(0, 3, 1, 2)
            mask = mask.reshape(B, self.render_size, self.render_size, -1).permute(0, 3, 1, 2)
            return render, mask, latent_code_loss
        return rgb_coarse, _, acc_coarse, weights, rgb_fine, _, acc_fine, latent_code_loss

    def _latent_code_loss(self, latent_code, mode):
        """mean((code - mean(codes))^2) (reference :44-46).  Outside training, on the device and without autograd it is a function of the
        parameters alone (the validation code is row 0): computed once per state of the weights instead of four launches at the head of
        every frame's chain, in front of the fork into the two generators."""
        lc = self.latent_codes
        if mode != "train" and lc.is_cuda and not torch.is_grad_enabled():
            from ..graph import weights_epoch
            key = (lc.data_ptr(), lc._version, weights_epoch())
            hit = self.__dict__.get("_lc_loss")
            if hit is None or hit[0] != key:
                hit = self.__dict__["_lc_loss"] = (key, torch.square(latent_code - lc.mean(dim=0, keepdims=True).detach()).mean())
            return hit[1]
        return torch.square(latent_code - lc.mean(dim=0, keepdims=True).detach()).mean()

    # ---------------------------------------------------------------------------------------------------------------
    def _use_hip(self, ray_batch):
        """HIP tensors that do not need gradients go to the fused (forward-only) kernel; there is no silent fallback for them.
        "Do not need gradients" = autograd is off, or nothing this call could differentiate exists: no parameter of the Trainer
        (MLP, encoders, latent codes, skinning volume) and no input requires grad.  Anything else takes the autograd statement,
        as the reference would backpropagate there."""
        if not ray_batch.is_cuda:
            return False
        if not torch.is_grad_enabled():
            return True
        planes = getattr(self.model_coarse, "triPlane_embeddings", None)
        if ray_batch.requires_grad or (planes is not None and planes.requires_grad):
            return False
        return not any(p.requires_grad for p in self.parameters())

    def predict_and_render_radiance(self, mode, ray_batch, background_prior, inv_head_T):
        opt = getattr(self.cfg.nerf, mode)
        if self._use_hip(ray_batch):
            return self._render_hip(opt, ray_batch, background_prior, inv_head_T)
        return self._render_torch(opt, ray_batch, background_prior, inv_head_T)

    def _render_hip(self, opt, ray_batch, background_prior, inv_head_T):
        m = self._hip_marcher()
        m.set_mlp(*[t.detach() for t in self.model_coarse.mlp_tensors()])
        planes = self.model_coarse.triPlane_embeddings
        key = (planes.data_ptr(), planes._version, m._blob_key)
        if getattr(self, "_planes_dirty", True) or getattr(self, "_planes_key", None) != key:
            m.set_triplane(planes.detach())
            self._planes_key, self._planes_dirty = key, False
        vol = self.headpose_skin_net.current_volume().detach()
        return m.render(ray_batch, background_prior, inv_head_T, vol, int(opt.num_coarse), int(opt.num_fine),
                        perturb=bool(opt.perturb), noise_std=float(opt.radiance_field_noise_std),
                        coarse_outputs=not getattr(self, "_decline_coarse", False))

    def _render_torch(self, opt, ray_batch, background_prior, inv_head_T):
        """PyTorch statement of predict_and_render_radiance (reference :120-201)."""
        B, R = ray_batch.shape[:2]
        ro, rd = ray_batch[..., :3], ray_batch[..., 3:6]
        near, far = ray_batch[..., 6:7], ray_batch[..., 7:8]
        t = torch.linspace(0.0, 1.0, opt.num_coarse, dtype=ro.dtype, device=ro.device)
        z = near * (1.0 - t) + far * t
        if opt.perturb:
            mids = 0.5 * (z[..., 1:] + z[..., :-1])
            upper, lower = torch.cat((mids, z[..., -1:]), dim=-1), torch.cat((z[..., :1], mids), dim=-1)
            z = lower + (upper - lower) * torch.rand(z.shape, dtype=ro.dtype, device=ro.device)
        bg = background_prior.reshape(-1, background_prior.shape[-1]) if background_prior is not None else None

        # HIP tensors under autograd: the field inputs (skinning + box warp + plane gather + encoding) and the compositing are
        # one kernel each way (hav_field_inputs_*, hav_composite_*); the MLP between them stays on rocBLAS.  No fallback: with
        # HAVATAR_HIP_TRAIN unset or 1 a missing library raises.
        fused = (ro.is_cuda and ro.dtype == torch.float32 and max(opt.num_coarse, opt.num_coarse // 2 + opt.num_coarse % 2 + opt.num_fine) <= 64
                 and self.model_coarse.sh_deg == 0 and os.environ.get("HAVATAR_HIP_TRAIN", "1") != "0")
        if fused:
            from ..native.train_ops import composite, field_inputs, field_mlp, field_mlp_eligible
            gw, sw = self.model_coarse.gridwarper, self.headpose_skin_net.gridwarper
            if getattr(self, "_boxes", None) is None:
                f = lambda t: t.detach().reshape(3).cpu().tolist()
                self._boxes = ((f(gw.scale_factor), f(gw.trans_factor)), (f(sw.scale_factor), f(sw.trans_factor)))
            vol = self.headpose_skin_net.current_volume()             # once per call: the reference re-evaluates the same
            planes = self.model_coarse.triPlane_embeddings             # VolumeDecoder (no RNG, same weights) for every pass

        def one_pass_hip(zv):
            pts = (ro[..., None, :] + rd[..., None, :] * zv[..., :, None]).reshape(B, -1, 3)
            mc = self.model_coarse
            ws = mc.mlp_tensors() if (torch.is_grad_enabled() and pts.shape[0] * pts.shape[1] >= 1024 and mc.sh_deg == 0
                                      and os.environ.get("HAVATAR_TRAIN_MLP", "bf16") == "bf16" and os.environ.get("HAVATAR_FIELD_MLP", "1") != "0") else None
            # patch training (dataloader.py:93-121: a 64 x 64 patch, row-major): neighbouring rays are neighbouring pixels.  With
            # HAVATAR_FIELD_ROWS=1 the scatter of the field-input gradients merges across 16 rays instead of along one (same sums;
            # native/train_ops.py::_field_backward).  Off by default: faster on dense in-box patches (1.82 -> 1.59 ms per call,
            # tools/bench_field_rows.py), slower inside this step on the harness' frames (0.685 -> 0.754 ms, profiles/r05_field_rows_ab.txt)
            rows = zv.shape[-1] if (os.environ.get("HAVATAR_FIELD_ROWS", "0") == "1" and bool(getattr(self.cfg.experiment, "patch_rgb", False))
                                    and R % 16 == 0) else 0
            if ws is not None and field_mlp_eligible(planes, ws):
                # field inputs + bf16-MFMA radiance MLP as one autograd node with bf16 rows in between (native/train_ops.py::FieldMlp)
                rf = field_mlp(pts, inv_head_T, vol, planes, *self._boxes, ws, ray_rows=rows).reshape(B * R, zv.shape[-1], -1)
            else:
                X = field_inputs(pts, inv_head_T, vol, planes, *self._boxes, ray_rows=rows)
                rf = mc.mlp(X).reshape(B * R, zv.shape[-1], -1).float()      # (bf16 under autocast: the compositing is fp32)
            std = float(opt.radiance_field_noise_std)
            noise = torch.randn(rf.shape[:-1], dtype=rf.dtype, device=rf.device) * std if std > 0.0 else None     # same draw as :56
            rgb, acc, w, depth = composite(rf, zv.reshape(-1, zv.shape[-1]), rd.reshape(-1, 3), noise, bg, n_sigmoid=3)
            return rgb, None, acc, w, depth

        def one_pass(zv):
            if fused:
                return one_pass_hip(zv)
            pts = ro[..., None, :] + rd[..., None, :] * zv[..., :, None]
            flat = pts.reshape(B, -1, 3)
            vd = ray_batch[..., -3:].unsqueeze(2).expand(pts.shape).reshape(B, -1, 3)
            rot, _ = self.headpose_skin_net(flat, vd, inv_head_T)
            feat = self.model_coarse.sample_pts_triplane_feat(batch_pts=rot)
            rf = self.model_coarse(rot.reshape(-1, 3), feat)
            rf = rearrange(rf, "(b r s) c -> (b r) s c", b=B, r=R)
            return volume_render_radiance_field(rf, depth_values=zv.reshape(-1, zv.shape[-1]), ray_directions=rd.reshape(-1, 3),
                                                radiance_field_noise_std=opt.radiance_field_noise_std, background_prior=bg, act_feat=False)

        rgb_c, _, acc_c, weights, depth_c = one_pass(z)
        rs = lambda x: x.reshape(B, R, -1)
        if opt.num_fine <= 0:
            return rs(rgb_c), rs(depth_c), rs(acc_c), rs(weights.max(dim=-1)[0]), None, None, None
        zf = z.reshape(-1, z.shape[-1])
        if fused and 3 <= zf.shape[-1] <= 128 and os.environ.get("HAVATAR_RESAMPLE", "hip") != "aten":
            # the three statements + sample_pdf as one launch (hav_resample_depths; ~30 ATen launches otherwise).  The stratified
            # draw is made here exactly where sample_pdf makes it (utils/nerf_util.py:95: on the CPU generator; on the device
            # generator while a training step is being captured, like utils/nerf_util.py::sample_pdf of this package)
            from ..native.train_ops import resample_depths
            zeta = None
            if opt.perturb != 0.0:
                shape = [zf.shape[0], int(opt.num_fine)]
                zeta = (torch.rand(shape, dtype=zf.dtype, device=zf.device) if torch.cuda.is_current_stream_capturing()
                        else torch.rand(shape, dtype=zf.dtype).to(zf.device))
            z2 = resample_depths(zf, weights, opt.num_fine, zeta)
        else:
            z_mid = 0.5 * (zf[..., 1:] + zf[..., :-1])
            z_s = sample_pdf(z_mid, weights[..., 1:-1], opt.num_fine, det=(opt.perturb == 0.0)).detach()
            z2, _ = torch.sort(torch.cat((zf[:, ::2], z_s), dim=-1), dim=-1)
        rgb_f, _, acc_f, weights, depth_f = one_pass(z2.reshape(B, R, -1))
        return rs(rgb_c), rs(depth_c), rs(acc_c), rs(weights.max(dim=-1)[0]), rs(rgb_f), rs(depth_f), rs(acc_f)

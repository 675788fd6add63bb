#!/usr/bin/env python3
"""Golden vectors at BASELINE config 2's FRAME SIZE: the REFERENCE's predict_and_render_radiance (imported from /root/reference, build
container only) on 2 080 rays of the 512 x 512 camera frame -- 16 random columns in every fourth image row plus the last row, and the very
last ray of the frame -- in fp32 (as it runs), in fp64 and in fp32 with the ray origins moved by one ulp (per-ray conditioning), with deterministic depths and with injected stratified
jitter.  Rays are independent in the reference (chunks are concatenated, model/nerf_trainer.py:66-77), so these rows of a full-frame call
are what the reference computes for a full frame.  The GPU test renders the WHOLE frame through the production kernels and compares these
rays with the reference itself (tests/test_render_gpu.py::test_full_frame_512_production_kernels_vs_the_reference)."""
import copy
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from gen_golden import import_reference  # noqa: E402

H = W = 512
S_C, S_F = 64, 16
SEED_T, SEED_U = 204, 205


def ray_subset():
    rng = np.random.default_rng(7)
    rows = sorted(set(range(0, H, 4)) | {H - 1})
    idx = np.concatenate([y * W + np.sort(rng.choice(W, 16, replace=False)) for y in rows])
    idx[-1] = H * W - 1
    return idx.astype(np.int64)


def main():
    from havatar_amd import synth
    torch, Trainer, cfg = import_reference()
    torch.manual_seed(0)
    tr = Trainer(cfg, 1)
    tr.requires_grad_(False)
    idx = ray_subset()
    n = idx.size
    out = {"idx": idx, "H": np.int64(H), "W": np.int64(W), "S_c": np.int64(S_C), "S_f": np.int64(S_F), "seed_t": np.int64(SEED_T), "seed_u": np.int64(SEED_U)}
    rays_all = synth.camera_rays(H, W)
    t_rand = synth.uniform((1, n, S_C), SEED_T)
    u_rand = synth.uniform((n, S_F), SEED_U)
    for recipe in ("primary", "stress"):
        sc = synth.scene(8, 8, recipe)
        rays_sub = rays_all[idx][None]
        bg = np.ones((1, n, 3), np.float32)

        def load(t, dtype):
            mc, m = t.model_coarse, sc["mlp"]
            with torch.no_grad():
                for lin, (wk, bk) in ((mc.layers_xyz[0], ("W1", "b1")), (mc.layers_xyz[1], ("W2", "b2")), (mc.fc_alpha, ("Wa", "ba")),
                                      (mc.fc_rgbFeat, ("Wf", "bf")), (mc.fc_rgb, ("Wc", "bc"))):
                    lin.weight.copy_(torch.from_numpy(m[wk])); lin.bias.copy_(torch.from_numpy(m[bk]))
            mc.triPlane_embeddings = torch.from_numpy(sc["planes"]).to(dtype)          # read at nerf_model.py:95
            sk = t.headpose_skin_net
            sk.fix_canoW = True                                                        # Skinning_Field.py:53,79
            sk.canonical_W = torch.from_numpy(sc["vol"]).to(dtype)[None]

        def run(t, dtype, perturb, rays=None):
            rays = rays_sub if rays is None else rays
            v = t.cfg.nerf.validation
            v.num_coarse, v.num_fine, v.perturb, v.radiance_field_noise_std = S_C, S_F, bool(perturb), 0.0
            r = torch.from_numpy(rays).to(dtype)
            rb = torch.cat([r, r[..., 3:6] / r[..., 3:6].norm(p=2, dim=-1, keepdim=True)], -1)          # nerf_trainer.py:52,63
            q = [torch.from_numpy(t_rand).to(dtype), torch.from_numpy(u_rand).to(dtype)] if perturb else []
            o_rand = torch.rand

            def f_rand(*a, **k):                    # nerf_trainer.py:138, then utils/nerf_util.py:95
                x = q.pop(0)
                shp = tuple(a[0]) if len(a) == 1 and not isinstance(a[0], int) else tuple(a)
                assert tuple(x.shape) == tuple(shp), (x.shape, shp)
                return x
            torch.rand = f_rand
            try:
                with torch.no_grad():
                    res = t.predict_and_render_radiance("validation", rb, torch.from_numpy(bg).to(dtype), inv_head_T=torch.from_numpy(sc["inv_T"]).to(dtype))
            finally:
                torch.rand = o_rand
            assert not q
            return [x.double().numpy().reshape(n, -1) for x in res]

        tr64 = copy.deepcopy(tr).double()
        load(tr, torch.float32); load(tr64, torch.float64)
        for tag, perturb in (("det", False), ("jit", True)):
            r32, r64 = run(tr, torch.float32, perturb), run(tr64, torch.float64, perturb)
            nudged = []
            for direction in (np.float32(np.inf), np.float32(-np.inf)):          # the fp32 reference with the ray origins one ulp up / down
                r2 = rays_sub.copy()
                r2[..., :3] = np.nextafter(rays_sub[..., :3], direction)
                nudged.append(run(tr, torch.float32, perturb, rays=r2))
            names = ["rgb_coarse", "depth_coarse", "acc_coarse", "weights_max", "rgb_fine", "depth_fine", "acc_fine"]
            key = "%s_%s_" % (recipe, tag)
            for j, (nm, a, b) in enumerate(zip(names, r32, r64)):
                if nm in ("rgb_fine", "depth_fine", "acc_fine"):
                    # per-ray conditioning: the reference's own fp32-vs-fp64 deviation, and how far its fp32 result moves when the ray
                    # origin moves by one ulp (sample_pdf divides by CDF increments at the 1e-5 floor, SURVEY B-11)
                    cond = np.abs(a - b).max(-1)
                    for q in nudged:
                        cond = np.maximum(cond, np.abs(q[j] - a).max(-1))
                    out[key + "cond_" + nm] = cond.astype(np.float32)
                if nm == "rgb_coarse":
                    a = a[:, :3]                              # (the 64 coarse feature channels are not kept: size)
                out[key + nm] = a.astype(np.float32)
            print(recipe, tag, "acc_fine in [%.3f, %.3f]" % (r32[6].min(), r32[6].max()), "max fp32-vs-fp64 deviation of rgb_fine %.2e" % np.abs(r32[4] - r64[4]).max())
    path = os.path.join(REPO, "tests", "golden", "frame512.npz")
    np.savez_compressed(path, **out)
    print("frame512.npz", os.path.getsize(path) // 1024, "KiB,", n, "rays")


if __name__ == "__main__":
    main()

"""Phase breakdown of conv3x3_split_kernel (see tools/conv_phase.sh): cycles per 16-channel chunk and wave."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from havatar_amd import _lib  # noqa: E402
from havatar_amd.native import conv  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
L.hav_conv_profile_buffer.argtypes = [C.c_void_p]
names = ["issue A loads + next chunk's x loads", "9 taps: ds_read + MFMAs (interleaved kernel: + all side work)", "s_nop + convert / split / ds_write", "barrier"]
L.hav_conv3x3_scratch_bytes.restype = C.c_int64
for (Cin, Cout, H) in ((512, 512, 64), (1024, 512, 64), (256, 256, 128)):
    x = torch.randn(1, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    pk = conv.pack(w, 1.0)
    buf = torch.zeros(1 << 18, dtype=torch.int64, device=dev)          # 4 counters per wave, more than any grid here has
    assert L.hav_conv_profile_buffer(C.c_void_p(buf.data_ptr())) == 0
    for _ in range(3):
        conv.conv3x3(x, pk, Cout, act=False, autoscale=False)
    torch.cuda.synchronize()
    r = buf.view(-1, 4).double()
    r = r[r.sum(1) > 0]
    ks = max(1, int(L.hav_conv3x3_scratch_bytes(1, Cin, Cout, H, H)) // (Cout * H * H * 4))
    nchunk = Cin // 16 // ks
    print("%d -> %d @ %d^2: %d waves reporting, %d chunks each; cycles per chunk and wave (mean over waves):" % (Cin, Cout, H, r.shape[0], nchunk))
    tot = r.sum(1).mean().item() / nchunk
    for q in range(4):
        print("   %-44s %8.0f  (%4.1f %%)" % (names[q], r[:, q].mean().item() / nchunk, 100 * r[:, q].mean().item() / nchunk / tot))
    print("   %-44s %8.0f   (MFMA issue floor per chunk: 54 x 32 = 1728 in the 64 x 128 kernels, 108 x 32 = 3456 in the interleaved 128 x 128 kernel)" % ("total", tot))

"""rbvae_conv3x3s2_halo against rbvae_gather_gemm (one-class conv descriptor) at the native / cfg 5 / LDM downsample / bench shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
from importlib import import_module
E = import_module("symbols-from-video_amd.engine")
L = sfv._lib
dev = torch.device("cuda", 0)
CASES = [("128 x 88x160 -> 44x80 (8 slices) ", 128, 88, 160, 256, 256), ("native conv2 (44x80 -> 22x40)", 128, 44, 80, 256, 256),
         ("cfg5 conv2 (64 x 32x32)", 64, 32, 32, 256, 256), ("LDM down 256 (4 x 256x256)", 4, 256, 256, 256, 256),
         ("LDM down 512 (4 x 128x128)", 4, 128, 128, 512, 512), ("LDM down 128 (4 x 512x512)", 4, 512, 512, 128, 128),
         ("bench conv2 (256 x 16x16)", 256, 16, 16, 256, 256)]
sel = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else range(len(CASES))
zero = torch.zeros(256, dtype=torch.uint8, device=dev)
desc = E.conv_classes(3)
d = (ctypes.c_int * len(desc))(*desc)


def timed(fn, it=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it


for i in sel:
    name, N, IH, IW, C, Co = CASES[i]
    OH, OW = IH // 2, IW // 2
    A = torch.randn(N * IH * IW, C, device=dev).to(torch.bfloat16)
    W = (torch.randn(Co, 9, C, device=dev) / 48).to(torch.bfloat16)
    b = torch.randn(Co, device=dev)
    o1 = torch.empty(N * OH * OW, Co, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty_like(o1)
    t_g = timed(lambda: L.call("rbvae_gather_gemm", 1, A, W, o1, b, None, None, None, zero, N, IH, IW, OH, OW, 2, OH, OW, 1, C, Co, C, Co,
                               9, 1, ctypes.addressof(d), 1, 0, 0.0, 1.0, 0, None, None))
    t_h = timed(lambda: L.call("rbvae_conv3x3s2_halo", 1, A, W, o2, b, None, None, N, IH, IW, C, Co, C, Co, 1, 0, 0.0, 1.0, 0, None, None))
    gf = 2.0 * N * OH * OW * Co * C * 9 / 1e9
    diff = float((o1.float() - o2.float()).abs().max())
    print(f"{name:40s} gather {t_g:8.1f} us ({gf / t_g * 1e3:6.0f} TF)   halo {t_h:8.1f} us ({gf / t_h * 1e3:6.0f} TF)   max|diff| {diff:.3g}", flush=True)

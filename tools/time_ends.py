"""The 3/4-channel ends standalone at a frame size (default native 128 x 4 x 88 x 160 -> 256 channels): us per launch and
TB/s of the bytes each must move.  RBVAE_LIB selects a variant library (tools/ab_variants.sh conv_first.hip ...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sfv_amd as sfv
L = sfv._lib
N, Cin, IH, IW, Nout = (int(a) for a in (sys.argv[1:6] if len(sys.argv) >= 6 else (128, 4, 88, 160, 256)))
OH, OW = (IH - 1) // 2 + 1, (IW - 1) // 2 + 1
P = N * OH * OW
x = torch.rand(N, Cin, IH, IW, device="cuda")
Wp = (torch.randn(Nout, 64, device="cuda") / 8).bfloat16()
b = torch.zeros(Nout, device="cuda")
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
out = torch.empty(P, Nout, dtype=torch.bfloat16, device="cuda")
gate = torch.ones(P, Nout, dtype=torch.bfloat16, device="cuda")
dpre = torch.randn(N, IH, IW, Cin, device="cuda") * 0.1
seed = torch.zeros(1, dtype=torch.int64, device="cuda")

def timed(f, it=30):
    for _ in range(3): f()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / it * 1e3

def fwd(): L.call("rbvae_conv_first_fused", 1, x, 0, 0, 0, 0, Cin * IH * IW, Wp, b, zero, None, out, N, Cin, IH, IW, Nout, Nout, 1, 1, 0.2, 1.25, 7, seed)
def dgr(): L.call("rbvae_deconv_last_dgrad_fused", 1, dpre, Wp, zero, None, gate, out, N, Cin, IH, IW, Nout, Nout, 1.0, None)
t = timed(fwd); by = x.numel() * 4 + out.numel() * 2
print(f"conv_first_fused  (frames -> {Nout} ch)      {t:7.1f} us  {by / t / 1e6:5.2f} TB/s of {by / 1e6:.0f} MB", flush=True)
t = timed(dgr); by = dpre.numel() * 4 + 2 * out.numel() * 2
print(f"deconv_last_dgrad (image grad -> {Nout} ch)  {t:7.1f} us  {by / t / 1e6:5.2f} TB/s of {by / 1e6:.0f} MB", flush=True)
# the last ConvTranspose2d + sigmoid + MSE: input maps OH x OW with Nout channels -> IH x IW images
rows = (torch.randn(P, Nout, device="cuda") * 0.5).bfloat16()
NY = -(-9 * Cin // 8) * 8
Vp = (torch.randn(NY, Nout, device="cuda") * 0.05).bfloat16()
b4 = torch.zeros(Cin, device="cuda")
oh2, ow2 = 2 * OH, 2 * OW
xr = torch.empty(N, Cin, oh2, ow2, device="cuda")
tgt = torch.rand(N, Cin, oh2, ow2, device="cuda")
parts = L.query("rbvae_deconv_last_fused_parts", 1, N, OH, OW, Nout, Cin)
ws = torch.zeros(5 * parts, device="cuda")
dp = torch.empty(N, oh2, ow2, Cin, device="cuda")
def last(): L.call("rbvae_deconv_last_fused", 1, rows, Vp, NY, b4, zero, N, OH, OW, Nout, Cin, xr, tgt, 0, 0, 0, 0, Cin * oh2 * ow2, ws, dp, 0.5)
t = timed(last); by = rows.numel() * 2 + 3 * xr.numel() * 4
print(f"deconv_last_fused ({Nout} ch -> image + loss)  {t:7.1f} us  {by / t / 1e6:5.2f} TB/s of {by / 1e6:.0f} MB", flush=True)
# the two weight gradients of the ends from the images themselves (rbvae_wgrad_first)
nblk = L.query("rbvae_wgrad_first_blocks", 1, Cin, IH, IW, Nout, N)
ks = max(1, min(256 // (Nout // 256), nblk)) if Nout % 256 == 0 else max(1, min(512 // (Nout // 64), nblk))
slabs = torch.empty(ks, Nout, 64, device="cuda")
dy = (torch.randn(P, Nout, device="cuda") / 8).bfloat16()
def wg0(): L.call("rbvae_wgrad_first", 1, 0, x, 0, 0, 0, 0, Cin * IH * IW, dy, slabs, zero, N, Cin, IH, IW, Nout, Nout, ks)
def wg1(): L.call("rbvae_wgrad_first", 1, 1, dpre, 0, 0, 0, 0, 0, dy, slabs, zero, N, Cin, IH, IW, Nout, Nout, ks)
by = x.numel() * 4 + dy.numel() * 2
t = timed(wg0); print(f"wgrad_first mode 0 (frames x dY, {ks} slabs)    {t:7.1f} us  {by / t / 1e6:5.2f} TB/s of {by / 1e6:.0f} MB", flush=True)
t = timed(wg1); print(f"wgrad_first mode 1 (image grad x d2)         {t:7.1f} us  {by / t / 1e6:5.2f} TB/s of {by / 1e6:.0f} MB", flush=True)

"""cfg 5 front end: frozen LDM VAE encode of 512x512 frames (frames/s, achieved TFLOP/s)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sfv_amd as sfv
torch.manual_seed(0)
m = sfv.LDMEncoder(compute_dtype="bf16", use_graph=bool(int(os.environ.get("LDM_GRAPH", "0")))).cuda()
shapes = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1:]] or [(8, 256), (4, 512)]     # e.g. 4x512
if "LDM_HALO_VARIANT" in os.environ:        # A/B of rbvae_conv3x3_halo's kernels (include/rbvae_dbg.h): 1 one tile per workgroup, 0 persistent, 2 auto
    sfv._lib.dbg_lib().rbvae_dbg_conv_halo_variant(int(os.environ["LDM_HALO_VARIANT"]))
for (N, S) in shapes:
    x = torch.rand(N, 3, S, S, device="cuda") * 2 - 1
    for _ in range(2): m.encode(x, sample=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = int(os.environ.get("LDM_ITERS", "5"))
    for _ in range(it): lat = m.encode(x, sample=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    gf = 1116.7 * (S / 512) ** 2          # GFLOP per frame (SURVEY 8d estimate at 512^2)
    print(f"{S}x{S}: {N / dt:.1f} frames/s, {dt * 1e3:.1f} ms per {N} frames, ~{gf * N / dt / 1e3:.0f} TFLOP/s, latent {tuple(lat.shape)}", flush=True)

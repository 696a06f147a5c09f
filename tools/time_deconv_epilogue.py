"""rbvae_deconv3x3s2_halo: what its epilogue pays for -- the same launch with bias / keyed dropout / ReLU gate / column sums
switched on and off (the main loop is the same in all of them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
L = sfv._lib
zero = torch.zeros(256, dtype=torch.uint8, device="cuda")
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
for N, cin, cout, TH, TW in [(128, 256, 256, 22, 40), (128, 64, 64, 64, 64)]:
    A = torch.randn(N * TH * TW, cin, device="cuda").bfloat16()
    Wp = (torch.randn(cout, 9, cin, device="cuda") / (1.5 * cin ** 0.5)).bfloat16()
    b = torch.randn(cout, device="cuda")
    rows = N * 4 * TH * TW
    o = torch.empty(rows, cout, dtype=torch.bfloat16, device="cuda")
    gate = torch.randn(rows, cout, device="cuda").bfloat16()
    ws = torch.empty(L.query("rbvae_deconv3x3s2_halo_colsum_rows", 1, N, TH, TW, cin, cout), cout, device="cuda")
    def run(bias, g, relu, drop, cs):
        L.call("rbvae_deconv3x3s2_halo", 1, A, Wp, o, bias, g, None, zero, N, TH, TW, cin, cout, cin, cout, relu, drop, 0.2, 1.25, 5, None, cs)
    res = {}
    for rnd in range(3):
        for name, args in (("plain", (None, None, 0, 0, None)), ("bias+relu", (b, None, 1, 0, None)), ("bias+relu+dropout (forward)", (b, None, 1, 1, None)),
                           ("gate", (None, gate, 0, 0, None)), ("gate+colsum (gradient)", (None, gate, 0, 0, ws))):
            res.setdefault(name, []).append(timeit(lambda: run(*args)))
    print(f"N={N} {cin}->{cout} grid {TH}x{TW}: " + "  ".join(f"{k} {min(v):.1f} us" for k, v in res.items()), flush=True)

"""Fused-step time of the bench workload at the reference's own latent sizes (best_models.txt: 25 / 50 / 100; sweeps
25-100): which LSTM kernels each size gets and what the step costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sfv_amd as sfv
dev = torch.device("cuda", 0)
for Ld in (int(a) for a in (sys.argv[1:] or ["25", "32", "50", "64", "75", "100", "128"])):
    torch.manual_seed(1)
    m = sfv.Seq2SeqBinaryVAE(4, 4, Ld, Ld, variant="percep", input_hw=(32, 32), compute_dtype="bf16").to(dev).train()
    tr = sfv.FusedTrainer(m, device_noise=True, use_graph=True, seed=1, alpha=1.0, beta_kl=1.0, bernoulli_p=0.1)
    item = torch.randn(16, 2, 8, 4, 32, 32, device=dev)
    for _ in range(10):
        tr.step(item, 0.7)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        tr.step(item, 0.7)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    eng = tr.eng
    print(f"L={Ld:4d}: {dt * 1e3:.4f} ms/step  {256 / dt:9.0f} frames/s  pair_fwd={bool(eng.lstm_pair and sfv._lib.query('rbvae_lstm_pair_fwd_ok', 8, Ld, 4))} "
          f"fc_split={eng.fc_split} lstm_cast={eng.lstm_cast} loss={tr.losses[0].item():.3f}", flush=True)

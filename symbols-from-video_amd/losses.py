"""Loss free-functions with the reference's signatures, on the HIP reductions.

  recon_loss, kl_binary_concrete, contrast_loss, triplet_loss, l1_loss
      models/percep_RBVAE/percep_RBVAE_train.py:28-107
  kl_binary_concrete_simple (no clamp, eps 1e-10)
      models/simple_RBVAE/simple_RBVAE_train.py:45-68
  contrast_term / triplet_term: the trainers' whole pairwise term in one launch
      models/percep_RBVAE/percep_RBVAE_train.py:534-543, models/triplet_RBVAE/triplet_RBVAE_train.py:461-468
"""
import torch

from . import _lib as L


def _need_cuda(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the HIP path needs CUDA/ROCm tensors (there is no CPU fallback)")


def _rows(t):
    return t.detach().reshape(-1, t.shape[-1]).float().contiguous()


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _need_cuda(a, "recon_loss")
        if a.shape != b.shape:
            raise RuntimeError(f"recon_loss: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
        af, bf = a.detach().float().contiguous(), b.detach().float().contiguous()
        out = torch.empty(1, device=a.device)
        ws = torch.empty(L.query("rbvae_mse_ws_floats", af.numel()), device=a.device)
        L.call("rbvae_mse_fwd", af, bf, af.numel(), out, ws)
        ctx.save_for_backward(af, bf)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        af, bf = ctx.saved_tensors
        da = torch.empty_like(af)
        L.call("rbvae_mse_bwd", af, bf, af.numel(), 1.0, g.reshape(1).float().contiguous(), da)
        return da, -da


def recon_loss(x_recon, x):
    return _Mse.apply(x_recon, x)


class _Kl(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p, eps, clamp):
        _need_cuda(q, "kl_binary_concrete")
        if not 0.0 < p < 1.0:
            raise ValueError(f"kl_binary_concrete: p={p} must lie in (0, 1)")
        qf = _rows(q)
        out = torch.empty(1, device=q.device)
        L.call("rbvae_kl_fwd", qf, out, qf.shape[0], qf.shape[1], float(p), float(eps), int(clamp))
        ctx.save_for_backward(qf)
        ctx.cfg = (float(p), float(eps), int(clamp), q.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (qf,) = ctx.saved_tensors
        p, eps, clamp, shape = ctx.cfg
        dq = torch.empty_like(qf)
        L.call("rbvae_kl_bwd", qf, dq, qf.shape[0], qf.shape[1], p, eps, clamp, 1.0, g.reshape(1).float().contiguous())
        return dq.reshape(shape), None, None, None


def kl_binary_concrete(q_logits, p=0.5, eps=1e-8):
    return _Kl.apply(q_logits, p, eps, True)


def kl_binary_concrete_simple(q_logits, p=0.5, eps=1e-10):
    return _Kl.apply(q_logits, p, eps, False)


class _PairDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, label, margin):
        _need_cuda(x1, "contrast_loss")
        if x1.shape != x2.shape:
            raise RuntimeError(f"contrast_loss: shapes {tuple(x1.shape)} and {tuple(x2.shape)} differ")
        a, b = _rows(x1), _rows(x2)
        out = torch.empty(1, device=x1.device)
        Ld = a.shape[1]
        L.call("rbvae_pairdist_fwd", a, b, Ld, Ld, a.shape[0], Ld, int(label), float(margin), 1e-6, out)
        ctx.save_for_backward(a, b)
        ctx.cfg = (int(label), float(margin), x1.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        label, margin, shape = ctx.cfg
        da, db = torch.empty_like(a), torch.empty_like(b)
        Ld = a.shape[1]
        L.call("rbvae_pairdist_bwd", a, b, Ld, Ld, a.shape[0], Ld, label, margin, 1e-6, 1.0,
               g.reshape(1).float().contiguous(), da, db, Ld, Ld, 0)
        return da.reshape(shape), db.reshape(shape), None, None


class _PairCos(torch.autograd.Function):
    """contrast_loss's 'cosine' branch (percep_RBVAE_train.py:94-96): F.cosine_similarity reduces over dim 1."""

    @staticmethod
    def forward(ctx, x1, x2, label, margin):
        _need_cuda(x1, "contrast_loss")
        if x1.shape != x2.shape or x1.dim() < 2:
            raise RuntimeError(f"contrast_loss: shapes {tuple(x1.shape)} and {tuple(x2.shape)}")
        a = x1.detach().float().movedim(1, -1).contiguous()
        b = x2.detach().float().movedim(1, -1).contiguous()
        Ld = a.shape[-1]
        a2, b2 = a.reshape(-1, Ld), b.reshape(-1, Ld)
        out = torch.empty(1, device=x1.device)
        L.call("rbvae_paircos_fwd", a2, b2, Ld, Ld, a2.shape[0], Ld, int(label), float(margin), 1e-8, out)
        ctx.save_for_backward(a2, b2)
        ctx.cfg = (int(label), float(margin), a.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        label, margin, shape = ctx.cfg
        da, db = torch.empty_like(a2), torch.empty_like(b2)
        Ld = a2.shape[1]
        L.call("rbvae_paircos_bwd", a2, b2, Ld, Ld, a2.shape[0], Ld, label, margin, 1e-8, 1.0,
               g.reshape(1).float().contiguous(), da, db, Ld, Ld)
        return da.reshape(shape).movedim(-1, 1), db.reshape(shape).movedim(-1, 1), None, None


def contrast_loss(x1, x2, label, margin: float = 1.0, dist="euclidean"):
    if dist not in ("euclidean", "cosine"):
        # the reference falls through with the string as the "distance" and fails inside torch.pow
        raise TypeError(f"contrast_loss: dist must be 'cosine' or 'euclidean', got {dist!r}")
    if label not in (0, 1):
        raise ValueError("label must be 0 (similar) or 1 (dissimilar)")
    if dist == "cosine":
        return _PairCos.apply(x1, x2, label, margin)
    return _PairDist.apply(x1, x2, label, margin)


class _Triplet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, p, n, margin, eps, swap):
        _need_cuda(a, "triplet_loss")
        af, pf, nf = _rows(a), _rows(p), _rows(n)
        out = torch.empty(1, device=a.device)
        Ld = af.shape[1]
        L.call("rbvae_triplet_fwd", af, pf, nf, Ld, Ld, Ld, af.shape[0], Ld, float(margin), float(eps), int(swap), out)
        ctx.save_for_backward(af, pf, nf)
        ctx.cfg = (float(margin), float(eps), int(swap), a.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        af, pf, nf = ctx.saved_tensors
        margin, eps, swap, shape = ctx.cfg
        da, dp, dn = (torch.empty_like(af) for _ in range(3))
        Ld = af.shape[1]
        L.call("rbvae_triplet_bwd", af, pf, nf, Ld, Ld, Ld, af.shape[0], Ld, margin, eps, swap, 1.0,
               g.reshape(1).float().contiguous(), da, dp, dn, Ld, Ld, Ld, 0)
        return da.reshape(shape), dp.reshape(shape), dn.reshape(shape), None, None, None


def triplet_loss(anchor, pos, neg, margin=1.0, p=2.0, eps=1e-08, swap=True, size_average=None, reduce=None,
                 reduction="mean"):
    if size_average is not None or reduce is not None:       # torch's legacy switches (F.triplet_margin_loss)
        reduction = "mean" if (size_average in (None, True) and reduce in (None, True)) else ("sum" if reduce in (None, True) else "none")
    if p != 2.0 or reduction not in ("mean", "sum"):
        raise NotImplementedError("triplet_loss covers p=2 with reduction 'mean' (what the reference trainer calls, "
                                  "triplet_RBVAE_train.py:82-96) or 'sum'")
    out = _Triplet.apply(anchor, pos, neg, margin, eps, swap)
    return out * _rows(anchor).shape[0] if reduction == "sum" else out


def l1_loss(q_logits, lamb):
    return lamb * q_logits.abs().sum()


class _ContrastTerm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h0, h1):
        _need_cuda(h0, "contrast_term")
        B, T, Ld = h0.shape
        if T < 2:
            raise ZeroDivisionError("float division by zero")   # what the reference raises for one state (:541)
        a, b = h0.detach().float().contiguous(), h1.detach().float().contiguous()
        out = torch.empty(1, device=h0.device)
        L.call("rbvae_contrast_term_fwd", a, b, B, T, Ld, out)
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, T, Ld = a.shape
        d0, d1 = torch.empty_like(a), torch.empty_like(b)
        L.call("rbvae_contrast_term_bwd", a, b, B, T, Ld, 1.0, g.reshape(1).float().contiguous(), d0, d1)
        return d0, d1


def contrast_term(h0, h1):
    return _ContrastTerm.apply(h0, h1)


class _TripletTerm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h0, h1, margin):
        _need_cuda(h0, "triplet_term")
        B, T, Ld = h0.shape
        if T < 2:
            raise ZeroDivisionError("float division by zero")
        a, b = h0.detach().float().contiguous(), h1.detach().float().contiguous()
        out = torch.empty(1, device=h0.device)
        L.call("rbvae_triplet_term_fwd", a, b, B, T, Ld, float(margin), out)
        ctx.save_for_backward(a, b)
        ctx.margin = float(margin)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        B, T, Ld = a.shape
        d0, d1 = torch.empty_like(a), torch.empty_like(b)
        L.call("rbvae_triplet_term_bwd", a, b, B, T, Ld, ctx.margin, 1.0, g.reshape(1).float().contiguous(), d0, d1)
        return d0, d1, None


def triplet_term(h0, h1, margin=1.0):
    return _TripletTerm.apply(h0, h1, margin)

"""MI355X-native RBVAE hot path (gfx950): Python host side over librbvae_hip.so.

The directory name carries a hyphen (it mirrors the reference repository's name),
so import it with importlib -- `sfv_amd.py` at the repo root does that:

    import sfv_amd as sfv
    model = sfv.Seq2SeqBinaryVAE(in_channels=4, out_channels=4, latent_dim=32, variant="percep").cuda()
"""
from . import _lib  # noqa: F401
from .engine import VARIANTS, Engine, ParamLayout  # noqa: F401
from .losses import (contrast_loss, contrast_term, kl_binary_concrete, kl_binary_concrete_simple,  # noqa: F401
                     l1_loss, recon_loss, triplet_loss, triplet_term)
from .data import (DeviceStatePairDataset, assign_label, build_pairs, consistency_from_codes,  # noqa: F401
                   split_indices, state_consistency)
from .ldm import LDMEncoder  # noqa: F401
from .model import Seq2SeqBinaryVAE, binary_concrete_logits  # noqa: F401
from .trainer import FusedTrainer, noise_key  # noqa: F401
from .compose import OnTheFlyLatentTrainer  # noqa: F401

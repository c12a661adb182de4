"""smrt_amd -- MI355X-native drop-in for SMRT's VAE train-and-generate hot path.

Public surface mirrors the reference package layout:

    smrt_amd.balance.smrt_balance            (reference: smrt/balance/smrt.py:22)
    smrt_amd.autoencode.VariationalAutoEncoder  (reference: smrt/autoencode/autoencoder.py:285)
    smrt_amd.utils                           (reference: smrt/utils.py)

All arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/smrt_hip.h`` (``smrt_amd/_lib/libsmrt_hip.so``); there is no CPU fallback.
"""
__version__ = "0.1.0"

__all__ = ["autoencode", "balance", "utils"]

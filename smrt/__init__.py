"""``smrt`` -- the reference's own import name, bound to the MI355X implementation in ``smrt_amd``.

The reference's callers write ``from smrt.balance import smrt_balance`` and ``from smrt.autoencode import
VariationalAutoEncoder`` (smrt/balance/tests/test_smrt.py:9, smrt/autoencode/tests/test_vae.py:8).  With this repository
on ``sys.path`` those lines work unchanged: every ``smrt.*`` module below IS the ``smrt_amd`` module of the same role (one
module object, two names), so there is one set of classes, one library handle, and still no CPU fallback.

    smrt.balance            -> smrt_amd.balance        (smrt_balance, smote_balance, over_/under_sample_balance)
    smrt.balance.smrt/...   -> smrt_amd.balance.smrt/...
    smrt.autoencode         -> smrt_amd.autoencode     (VariationalAutoEncoder, AutoEncoder)
    smrt.autoencode.autoencoder -> smrt_amd.autoencode (both estimator classes), .layer -> .topography
    smrt.utils              -> smrt_amd.utils

Do not put this directory on the path next to the reference's own ``smrt`` package: whichever comes first wins.
"""
import sys as _sys

import smrt_amd as _impl
from smrt_amd import autoencode, balance, utils
from smrt_amd.autoencode import ae as _ae, topography as _topography
from smrt_amd.balance import base as _bbase, over as _over, smote as _smote, smrt as _smrt, under as _under

__version__ = _impl.__version__
__all__ = ["autoencode", "balance", "utils"]

for _name, _mod in (("balance", balance), ("autoencode", autoencode), ("utils", utils),
                    ("balance.base", _bbase), ("balance.over", _over), ("balance.smote", _smote),
                    ("balance.smrt", _smrt), ("balance.under", _under),
                    ("autoencode.autoencoder", autoencode), ("autoencode.layer", _topography), ("autoencode.ae", _ae)):
    _sys.modules[__name__ + "." + _name] = _mod

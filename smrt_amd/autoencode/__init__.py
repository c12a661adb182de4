"""Auto-encoders on the MI355X HIP backend (reference package: smrt/autoencode)."""
from .topography import *      # noqa: F401,F403
from .vae import *             # noqa: F401,F403
from .data_parallel import *   # noqa: F401,F403
from .ae import *              # noqa: F401,F403

__all__ = [s for s in dir() if not s.startswith("_")]

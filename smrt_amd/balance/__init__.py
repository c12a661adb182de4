"""Class balancing on the MI355X HIP backend (reference package: smrt/balance)."""
from .base import *       # noqa: F401,F403
from .over import *       # noqa: F401,F403
from .smote import *      # noqa: F401,F403
from .smrt import *       # noqa: F401,F403
from .under import *      # noqa: F401,F403

__all__ = [s for s in dir() if not s.startswith("_")]

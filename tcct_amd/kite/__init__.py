from .losses import *       # noqa: F401,F403
from .loopback import KiteBack, setup_seed      # noqa: F401
from .loop_seg import KiteSeg                   # noqa: F401

from .tcct import *      # noqa: F401,F403  (stc_tt, tcct, FTC, CrossResNet, MPViT, mpvit_tiny ...) — reference nets/__init__.py:1
from .reg import *       # noqa: F401,F403  (RegNet)
from .tcct import stc_tt, tcct, stc_tb, gtc_tt, gtc_tb, cnnu, pnnu, vitu, FTC, CrossResNet, MPViT, mpvit_tiny
from .reg import RegNet, as_label_index, as_nhwc

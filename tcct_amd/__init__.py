"""tcct_amd — MI355X (gfx950) native implementation of the TCCT `stc_tt` training hot path.

Host side mirrors the reference's Python object protocol (nets / kite packages); compute is hand-written HIP behind
the C-ABI in include/tcct_hip.h.  Importing this package never touches the oracle and never falls back to CPU math."""
__version__ = '0.1.0'

import os as _os
import warnings as _warnings

# Environment switches that earlier rounds read and round 5 turned into module constants (tcct_amd.ops.NAME; `bench.py --set NAME=0|1` for A/B runs).
# Setting one has NO effect any more: say so once instead of silently ignoring it.
RETIRED_ENV = frozenset('''TCCT_BILINEAR_VEC8 TCCT_BN_DEFER TCCT_BN_DEFER_DW TCCT_BN_FUSE TCCT_BN_POOL TCCT_BN_RED TCCT_BN_RED_DW TCCT_BN_VEC8 TCCT_C3_BN TCCT_C3_DIRECT
TCCT_C3_ONEPASS TCCT_CONV_STREAM TCCT_DP_MARKS TCCT_DS_DICE TCCT_DW_CPT TCCT_DW_WGRAD_CPT TCCT_DW_WGRAD_SIDE TCCT_DY2 TCCT_F32_MFMA TCCT_FPL_LAZY_GRAD
TCCT_FUSED_CONV_BWD TCCT_FUSED_PW_BWD TCCT_FUSE_AUX TCCT_GRAPH_ERRMODE TCCT_GRAPH_SHARED_POOL TCCT_HEAD_COMPOSE TCCT_HEAD_UPADD TCCT_LN_POOL TCCT_LN_POOL_LN2
TCCT_MLP_GELU TCCT_MLP_TAIL TCCT_NORMADD_BAND TCCT_PACK_ALL TCCT_PW_FWD2 TCCT_TAIL_AUX TCCT_TAIL_AUX_LOW TCCT_TAIL_COMPOSE TCCT_TRAN_FUSE TCCT_TRAN_RED2
TCCT_WGRAD_FRESH_EVENT TCCT_WGRAD_GENERIC TCCT_WGRAD_LINE TCCT_WGRAD_RECORD_STREAM TCCT_WGRAD_STREAM'''.split())
_set = sorted(v for v in RETIRED_ENV if v in _os.environ)
if _set:
    _warnings.warn('tcct_amd: ' + ', '.join(_set) + ' no longer have any effect (retired environment switches; use the tcct_amd.ops module constants or '
                   '`bench.py --set NAME=0|1`)', RuntimeWarning, stacklevel=2)

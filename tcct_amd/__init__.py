"""tcct_amd — MI355X (gfx950) native implementation of the TCCT `stc_tt` training hot path.

Host side mirrors the reference's Python object protocol (nets / kite packages); compute is hand-written HIP behind
the C-ABI in include/tcct_hip.h.  Importing this package never touches the oracle and never falls back to CPU math."""
__version__ = '0.1.0'

"""clstm_amd -- MI355X-native backing of tmbdev/clstm's LSTM + CTC hot path.

Only what the path needs: csrc/ (hand-written HIP kernels + the C ABI of
include/clstm_abi.h), abi.py (ctypes plumbing) and net.py (host mirror of the reference's
INetwork / make_net / sgd_update surface).  See DESIGN.md and INTEGRATION.md.
"""
from .abi import ClstmError, load  # noqa: F401
from .net import Network, make_net, sgd_update, mktargets  # noqa: F401

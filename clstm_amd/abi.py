"""ctypes binding of libclstm_hip.so (include/clstm_abi.h).

This is the thin host-side plumbing over the C ABI; it contains no arithmetic.  The library
is the hand-written HIP build for gfx950 and MUST be present: there is no CPU fallback
(`load()` raises if clstm_amd/lib/libclstm_hip.so is missing -- run `python -c "import
__graft_entry__ as g; g.build()"` or `make -C clstm_amd/csrc`).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LAYERS = 4


class NetDesc(C.Structure):
    _fields_ = [("nlayers", C.c_int), ("unidirectional", C.c_int), ("ninput", C.c_int),
                ("nhidden", C.c_int * MAX_LAYERS), ("nclasses", C.c_int)]


class ClstmError(RuntimeError):
    pass


def default_lib_path():
    variant = os.environ.get("CLSTM_HIP_VARIANT", "")
    name = "libclstm_hip_%s.so" % variant if variant else "libclstm_hip.so"
    return os.environ.get("CLSTM_HIP_LIB", os.path.join(_HERE, "lib", name))


# (restype is always int status unless listed in _SPECIAL)
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
_SIGS = {
    "clstm_set_stream": [_P],
    "clstm_synchronize": [],
    "clstm_forward_nonlin0": [_P, _I, _I],
    "clstm_backward_nonlin0": [_P, _P, _I, _I],
    "clstm_forward_nonlin": [_P, _P, _I, _I],
    "clstm_backward_nonlin": [_P, _P, _P, _I, _I],
    "clstm_forward_lin1": [_P, _P, _P, _I, _I, _I],
    "clstm_backward_lin1": [_P, _P, _P, _P, _P, _I, _I, _I],
    "clstm_forward_full1": [_P, _P, _P, _I, _I, _I, _I],
    "clstm_backward_full1": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "clstm_forward_softmax": [_P, _P, _P, _I, _I, _I],
    "clstm_backward_softmax": [_P, _P, _P, _P, _P, _I, _I, _I],
    "clstm_forward_stack": [_P, _P, _P, _I, _I, _I],
    "clstm_backward_stack": [_P, _P, _P, _I, _I, _I],
    "clstm_forward_stack_delay": [_P, _P, _P, _I, _I, _I],
    "clstm_backward_stack_delay": [_P, _P, _P, _I, _I, _I],
    "clstm_forward_reverse": [_P, _P, _I, _I, _I],
    "clstm_backward_reverse": [_P, _P, _I, _I, _I],
    "clstm_forward_btswitch": [_P, _P, _I, _I, _I],
    "clstm_backward_btswitch": [_P, _P, _I, _I, _I],
    "clstm_forward_batchstack": [_P, _P, _I, _I, _I, _I, _I],
    "clstm_backward_batchstack": [_P, _P, _I, _I, _I, _I, _I],
    "clstm_forward_statemem": [_P, _P, _P, _P, _P, _I],
    "clstm_backward_statemem": [_P] * 9 + [_I],
    "clstm_forward_nonlingate": [_P, _P, _P, _I, _I],
    "clstm_backward_nonlingate": [_P, _P, _P, _P, _P, _I, _I],
    "clstm_clip_gradient": [_P, _I, _F],
    "clstm_sgd_update": [_P, _P, _I, _F, _F],
    "clstm_ctc_align_batch": [_P, _P, _P, _I, _P, _P, _P, _I],
    "clstm_mktargets": [_P, _P, _I],
    "clstm_trivial_decode_batch": [_P, _I, _P, _I, _P, _P, _P],
    "clstm_net_nparams_for": [_P],
    "clstm_net_create": [_P, _P, _P, _P, _P],
    "clstm_net_destroy": [_P],
    "clstm_net_nparams": [_P],
    "clstm_net_buffers": [_P, _P, _P, _P],
    "clstm_net_set_params_h": [_P, _P],
    "clstm_net_get_params_h": [_P, _P],
    "clstm_net_set_derivs_h": [_P, _P],
    "clstm_net_get_derivs_h": [_P, _P],
    "clstm_net_get_grads_h": [_P, _P],
    "clstm_net_params_changed": [_P],
    "clstm_net_set_learning_rate": [_P, _F, _F],
    "clstm_net_set_gradient_clip": [_P, _F],
    "clstm_net_set_batch": [_P, _P, _I],
    "clstm_net_set_inputs_h": [_P, _P],
    "clstm_net_set_inputs_d": [_P, _P],
    "clstm_net_set_gemm_precision": [_P, _I],
    "clstm_net_forward": [_P],
    "clstm_net_outputs": [_P, _P, _P],
    "clstm_net_get_outputs_h": [_P, _P],
    "clstm_net_set_output_deltas_h": [_P, _P],
    "clstm_net_ctc": [_P, _P, _P, _P],
    "clstm_net_backward": [_P],
    "clstm_net_enable_input_deltas": [_P, _I],
    "clstm_net_get_input_deltas_h": [_P, _P],
    "clstm_net_update": [_P],
    "clstm_net_decode": [_P, _P, _P, _P],
    "clstm_net_get_state_h": [_P, _I, _I, _I, _P],
    "clstm_net_enable_timing": [_P, _I],
    "clstm_net_kernel_time_ms": [_P, C.c_char_p, _P, _P],
    "clstm_net_reset_timing": [_P],
    "clstm_net_train_step": [_P, _P, _I, _P, _P, _P],
    "clstm_net_train_step_h": [_P, _P, _I, _P, _P, _P],
    "clstm_net_train_step_next": [_P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P],
    "clstm_host_alloc": [_P, C.c_size_t],
    "clstm_host_free": [_P],
    "clstm_net_n_states": [_P, _P],
    "clstm_net_get_states_h": [_P, _P, C.c_longlong],
    "clstm_net_set_states_h": [_P, _P, C.c_longlong],
    "clstm_comm_unique_id": [_P],
    "clstm_comm_create": [_P, _P, _I, _I],
    "clstm_comm_destroy": [_P],
    "clstm_comm_rank": [_P],
    "clstm_comm_size": [_P],
    "clstm_comm_peer_active": [_P],
    "clstm_allreduce_flat": [_P, _P, C.c_longlong],
    "clstm_net_set_comm": [_P, _P],
    "clstm_net_replica_check": [_P],
    "clstm_net_set_training": [_P, _I],
    "clstm_net_set_overlap": [_P, _I],
    "clstm_net_set_strict_f32": [_P, _I],
    "clstm_net_overlap_stats": [_P, _P, _P],
    "clstm_debug_lane_ops": [_P],
    "clstm_debug_ctc_cycles": [_P],
    "clstm_debug_gemm": [_I, _P, _P, _P, _I, _I, _I, _I],
    "clstm_debug_path_count": [_I, _P],
    "clstm_debug_set_option": [C.c_char_p, _I],
    "clstm_debug_set_device_error": [_I, _I],
}
# functions whose int return value is a result, not a status
_VALUE_RETURN = {"clstm_net_nparams_for", "clstm_net_nparams", "clstm_abi_version", "clstm_comm_rank", "clstm_comm_size", "clstm_comm_peer_active"}
EXPORTED_SYMBOLS = sorted(list(_SIGS) + ["clstm_last_error", "clstm_abi_version"])


class Lib:
    """Loaded libclstm_hip.so with checked calls: lib.call('clstm_net_forward', handle)."""

    def __init__(self, path=None):
        path = path or default_lib_path()
        if not os.path.exists(path):
            raise ClstmError(
                "HIP extension not built: %s is missing (no CPU fallback exists; build it with "
                "`make -C clstm_amd/csrc` or __graft_entry__.build())" % path)
        self.path = path
        # PyTorch-ROCm bundles its own HIP runtime; if libclstm_hip.so pulled /opt/rocm's copy into the
        # process first, torch would later find the GPU "unavailable".  Let torch load its runtime first.
        if os.path.basename(path).startswith("libclstm_hip"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.dll = C.CDLL(path)
        self.dll.clstm_last_error.restype = C.c_char_p
        self.dll.clstm_last_error.argtypes = []
        self.dll.clstm_abi_version.restype = C.c_int
        for name, args in _SIGS.items():
            if os.environ.get("CLSTM_ABI_LAX") and not hasattr(self.dll, name):
                continue                   # (A/B runs against a library built from an older commit)
            fn = getattr(self.dll, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = C.c_int
            fn.argtypes = args

    def call(self, name, *args):
        rc = getattr(self.dll, name)(*args)
        if name in _VALUE_RETURN:
            return rc
        if rc != 0:
            raise ClstmError("%s: %s" % (name, self.dll.clstm_last_error().decode()))
        return 0


_default = None


def load(path=None):
    global _default
    if path is not None:
        return Lib(path)
    if _default is None:
        _default = Lib()
    return _default


def ptr(a):
    """Raw address of a numpy array, a torch tensor, an int, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError(type(a))


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)

"""ctypes binding of the C ABI in ``include/spfm.h`` (``lib/libspfm_hip.so``).

This is the only way the package reaches the device: there is no CPU fallback.
If the shared library is missing or no HIP device is present, the calls below
raise ``RuntimeError`` -- the estimators never silently compute elsewhere.
"""
import ctypes as C
import os

import numpy as np

# Concurrent fits (sparsepoly_amd/concurrent.py) give every fit its own HIP stream.  The HIP
# runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and, once
# those are taken, doubles streams up on a queue -- two persistent passes on one queue run one
# after the other.  Four fits plus the null stream (PyTorch's default stream, any synchronous
# hipMemcpy) are five streams.  The variable is read when the HIP runtime initialises, i.e. at the
# first HIP call of the process.  It is NOT set at import (a process-wide side effect on every
# other HIP user): `ensure_hw_queues` sets it when concurrent fits are first asked for, if the
# runtime is not up yet, and warns when it is too late.


def hip_initialised():
    """Has the HIP / HSA runtime of this process initialised (then it has read its environment)?
    The runtime opens /dev/kfd when it does; looking at our own descriptors initialises nothing."""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return True
            except OSError:
                continue
    except OSError:
        pass
    return False


HIP_INITIALISED_BEFORE_IMPORT = hip_initialised()
QUEUES_SET_BY_PACKAGE = False


def ensure_hw_queues(n_streams):
    """Called before `n_streams` fits are started side by side.  Makes sure the HIP runtime maps
    them onto distinct hardware queues: exports GPU_MAX_HW_QUEUES=8 if the runtime has not
    initialised yet and the variable is unset; when it is too late for that and fewer queues than
    streams (+ the null stream) are configured, warns that fits will share queues -- two
    persistent passes on one queue run one after the other (measured: 279 / 279 / 551 / 551 ms per
    iteration instead of 4 x 285).  Returns the queue count the runtime is taken to use."""
    global QUEUES_SET_BY_PACKAGE
    import warnings

    need = int(n_streams) + 1
    val = os.environ.get("GPU_MAX_HW_QUEUES")
    have = int(val) if (val and val.isdigit()) else None
    if not hip_initialised():
        if have is None:
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
            QUEUES_SET_BY_PACKAGE = True
            have = 8
        return have
    eff = have if have is not None else 4
    if eff < need and n_streams > 1:
        warnings.warn(
            "sparsepoly_amd: %d concurrent fits, but the HIP runtime of this process was "
            "initialised with %d hardware queues (GPU_MAX_HW_QUEUES %s): fits that share a queue "
            "run one after the other.  Export GPU_MAX_HW_QUEUES=8 before the process first "
            "touches the GPU (PyTorch users: before the first torch.cuda call)."
            % (n_streams, eff, "= %d" % have if have is not None else "unset"),
            RuntimeWarning, stacklevel=3)
    return eff


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPFM_HIP_LIB") or os.path.join(_HERE, "lib", "libspfm_hip.so")

SPFM_OK, SPFM_ERR_INVALID, SPFM_ERR_RUNTIME, SPFM_ERR_UNSUPPORTED = 0, -1, -2, -3
DTYPES = {"f32": 0, "f64": 1}
LOSSES = {"squared": 0, "squared_hinge": 1, "logistic": 2}
REGULARIZERS = {"l1": 0, "l21": 1, "squaredl12": 2, "squaredl21": 3, "omegati": 4, "omegacs": 5}
SOLVERS = {"pcd": 0, "pbcd": 1, "psgd": 2}
LEARNING_RATE = {"constant": 0, "optimal": 1, "pegasos": 2, "invscaling": 3}
SCHEDULES = {"exact": 0, "colored": 1}

# every symbol include/spfm.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "spfm_create", "spfm_destroy", "spfm_last_error", "spfm_device_name", "spfm_build_tag", "spfm_set_data_csc", "spfm_set_data_csr", "spfm_share_data",
    "spfm_set_params", "spfm_get_params", "spfm_configure", "spfm_init_pred", "spfm_get_y_pred",
    "spfm_loss_sum", "spfm_predict_csr", "spfm_set_schedule", "spfm_set_schedule_raw",
    "spfm_get_schedule", "spfm_schedule_build",
    "spfm_cd_linear_epoch",
    "spfm_pcd_epoch", "spfm_pbcd_epoch", "spfm_psgd_epoch", "spfm_psgd_epoch_sharded",
    "spfm_host_epoch_begin",
    "spfm_host_pass_begin", "spfm_host_step_sums", "spfm_host_step_apply", "spfm_host_epoch_end", "spfm_comm_unique_id", "spfm_comm_init", "spfm_comm_init_shm", "spfm_peer_alloc", "spfm_peer_connect",
    "spfm_profile_enable", "spfm_profile_get", "spfm_profile_reset", "spfm_set_use_graph",
    "spfm_set_option", "spfm_get_option", "spfm_debug_prb_stamps", "spfm_debug_hop_latency", "spfm_debug_exchange_cost",
    "spfm_debug_branch_counts", "spfm_debug_stream_probe", "spfm_debug_write_probe",
]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_h = C.c_void_p

_lib = None


def load():
    """dlopen libspfm_hip.so and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "sparsepoly_amd: HIP extension %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`bash sparsepoly_amd/csrc/build.sh`; there is no CPU fallback." % LIB_PATH
        )
    L = C.CDLL(LIB_PATH)
    L.spfm_create.argtypes = [C.POINTER(_h), C.c_int, C.c_int]
    L.spfm_destroy.argtypes = [_h]
    L.spfm_destroy.restype = None
    L.spfm_last_error.argtypes = [_h]
    L.spfm_last_error.restype = C.c_char_p
    L.spfm_device_name.argtypes = [_h, C.c_char_p, C.c_int]
    L.spfm_build_tag.argtypes = []
    L.spfm_build_tag.restype = C.c_char_p
    L.spfm_set_data_csc.argtypes = [_h, C.c_int64, C.c_int32, _lp, _ip, _dp, _dp]
    L.spfm_set_data_csr.argtypes = [_h, C.c_int64, C.c_int32, _lp, _ip, _dp, _dp]
    L.spfm_share_data.argtypes = [_h, _h, _dp]
    L.spfm_set_params.argtypes = [_h, C.c_int, C.c_int, C.c_int32, _dp, _dp, _dp]
    L.spfm_get_params.argtypes = [_h, _dp, _dp]
    L.spfm_configure.argtypes = [_h, C.c_int, C.c_int, C.c_int, C.c_int]
    L.spfm_init_pred.argtypes = [_h, C.c_int, C.c_int, C.c_int]
    L.spfm_get_y_pred.argtypes = [_h, _dp]
    L.spfm_loss_sum.argtypes = [_h, _dp]
    L.spfm_predict_csr.argtypes = [_h, C.c_int64, _lp, _ip, _dp, C.c_int, C.c_int, C.c_int, _dp]
    L.spfm_set_schedule.argtypes = [_h, C.c_int, _ip, _lp, _ip, C.c_int64, _ip, _ip]
    L.spfm_set_schedule_raw.argtypes = [_h, _ip, _ip, C.c_int32, _lp, _ip, C.c_int64]
    L.spfm_get_schedule.argtypes = [_h, _ip, _ip, _ip]
    L.spfm_schedule_build.argtypes = [C.c_int, C.c_int64, C.c_int32, _lp, _ip, _ip, C.c_int, _ip,
                                      _ip, _ip]
    L.spfm_cd_linear_epoch.argtypes = [_h, C.c_double, _dp]
    L.spfm_pcd_epoch.argtypes = [_h, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _ip,
                                 C.c_int, _dp]
    L.spfm_pbcd_epoch.argtypes = [_h, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp]
    L.spfm_psgd_epoch.argtypes = [_h, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_int, C.c_double, C.c_int64, _ip, C.c_int64, C.c_int, _lp,
                                  _dp]
    L.spfm_psgd_epoch_sharded.argtypes = [_h, C.c_int, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int, C.c_double, C.c_int64, _ip,
                                          C.c_int64, C.c_int64, C.c_int, _lp, _dp]
    L.spfm_host_epoch_begin.argtypes = [_h, C.c_int, C.c_int]
    L.spfm_host_pass_begin.argtypes = [_h, C.c_int]
    L.spfm_host_step_sums.argtypes = [_h, C.c_int, _dp]
    L.spfm_host_step_apply.argtypes = [_h, C.c_int, _dp, _dp]
    L.spfm_host_epoch_end.argtypes = [_h, _dp]
    L.spfm_comm_unique_id.argtypes = [C.c_char_p]
    L.spfm_comm_init.argtypes = [_h, C.c_char_p, C.c_int, C.c_int]
    L.spfm_comm_init_shm.argtypes = [_h, C.c_char_p, C.c_int, C.c_int]
    L.spfm_profile_enable.argtypes = [_h, C.c_int]
    L.spfm_profile_get.argtypes = [_h, C.c_int, _dp, _lp, _lp]
    L.spfm_profile_reset.argtypes = [_h]
    L.spfm_set_use_graph.argtypes = [_h, C.c_int]
    L.spfm_set_option.argtypes = [_h, C.c_char_p, C.c_int]
    L.spfm_get_option.argtypes = [_h, C.c_char_p, C.POINTER(C.c_int)]
    L.spfm_debug_prb_stamps.argtypes = [_h, _lp, C.c_int]
    L.spfm_debug_hop_latency.argtypes = [_h, C.c_int, C.c_int, _dp, _ip]
    L.spfm_debug_exchange_cost.argtypes = [_h, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
    L.spfm_peer_alloc.argtypes = [_h, C.c_char_p]
    L.spfm_peer_connect.argtypes = [_h, C.c_int, C.c_int, C.c_char_p]
    L.spfm_debug_branch_counts.argtypes = [_h, C.POINTER(C.c_uint32), C.c_int]
    L.spfm_debug_stream_probe.argtypes = [_h, _lp]
    L.spfm_debug_write_probe.argtypes = [_h, C.c_int, _lp]
    for name in SYMBOLS:
        f = getattr(L, name)
        if name not in ("spfm_destroy", "spfm_last_error", "spfm_build_tag"):
            f.restype = C.c_int
    _lib = L
    return L


def f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_lp)


def build_tag():
    """Hash of the library's sources at build time (``spfm_build_tag``)."""
    return load().spfm_build_tag().decode()

"""ctypes binding of liblvt_hip.so (C ABI declared in include/lvt_hip.h).

The shared library is built in-tree by `__graft_entry__.build()` / `make -C lvt_amd/csrc`.
There is NO fallback: if the library is missing, or an op is called on a non-GPU tensor, the call
raises.  PyTorch is used only for device memory, streams and autograd plumbing.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("LVT_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "liblvt_hip.so")

EPI_BIAS, EPI_RESIDUAL, EPI_RELU, EPI_TANH, EPI_MASK, EPI_ACCUM, EPI_PLANES = 1, 2, 4, 8, 16, 32, 64
CAUSAL_KMAX, CAUSAL_KMIN, CAUSAL_TILE = 1 << 8, 1 << 9, 1 << 10      # causal attention products (include/lvt_hip.h)
ABI_VERSION = 300           # lvt_version() of the library this module binds (argument lists below)
MATH_F32 = 1 << 16          # per-call arithmetic selector of the engine entry points (include/lvt_hip.h)


class LvtError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ta", C.c_int), ("tb", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_longlong), ("a_kb", C.c_int), ("a_skb", C.c_longlong),
        ("B", C.c_void_p), ("ldb", C.c_longlong), ("b_kb", C.c_int), ("b_skb", C.c_longlong),
        ("C", C.c_void_p), ("ldc", C.c_longlong),
        ("batch_outer", C.c_int), ("batch_inner", C.c_int),
        ("sA_o", C.c_longlong), ("sA_i", C.c_longlong), ("sB_o", C.c_longlong), ("sB_i", C.c_longlong),
        ("sC_o", C.c_longlong), ("sC_i", C.c_longlong),
        ("alpha", C.c_float), ("flags", C.c_int),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ldr", C.c_longlong),
        ("mask", C.c_void_p), ("ldm", C.c_longlong), ("splits", C.c_int),
        ("a_colsum", C.c_void_p), ("c_plane", C.c_longlong),
    ]


class OptEntry(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("s0", C.c_void_p), ("s1", C.c_void_p),
                ("n", C.c_longlong), ("lr", C.c_float), ("wd", C.c_float)]


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("N", "Ti", "Hi", "Wi", "Ci", "To", "Ho", "Wo", "Co", "Kt", "Kh", "Kw",
                 "st", "sh", "sw", "pt", "ph", "pw")]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


_lib = None


def _declare(lib):
    vp, ci, cll, cf, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t
    P = C.POINTER
    sigs = {
        "lvt_last_error": (C.c_char_p, []),
        "lvt_version": (ci, []),
        "lvt_device_info": (ci, [C.c_char_p, ci, P(ci), P(ci), P(cll)]),
        "lvt_gemm_workspace_bytes": (sz, [P(GemmDesc)]),
        "lvt_gemm_f32": (ci, [P(GemmDesc), vp, sz, vp]),
        "lvt_gemm_smallm_f32": (ci, [ci, ci, ci, ci, vp, cll, vp, cll, vp, cll, ci, cll, cll, cf, ci, vp, vp, cll, vp, cll, cll, vp]),
        "lvt_gemm_smallm_splitk_workspace_bytes": (sz, [ci, ci, ci]),
        "lvt_gemm_smallm_splitk_f32": (ci, [ci, ci, ci, ci, vp, cll, vp, cll, vp, cll, cf, ci, vp, vp, cll, vp, cll, cll, vp, sz, vp]),
        "lvt_gemm_smallm_partial_f32": (ci, [ci, ci, ci, ci, vp, cll, vp, cll, vp, sz, vp]),
        "lvt_splitsum_layernorm_fwd": (ci, [vp, ci, ci, ci, vp, vp, cll, vp, cf, vp, vp, vp, vp]),
        "lvt_conv3d_pack_weight": (ci, [P(ConvGeom), vp, ci, ci, vp, vp]),
        "lvt_conv3d_pack_weight_t": (ci, [P(ConvGeom), vp, ci, ci, vp, vp]),
        "lvt_conv3d_uses_patch_kernel": (ci, [P(ConvGeom), ci]),
        "lvt_conv3d_fwd_uses_parity_kernel": (ci, [P(ConvGeom), ci]),
        "lvt_conv3d_pack_weight_parity": (ci, [P(ConvGeom), vp, ci, ci, vp, vp]),
        "lvt_conv3d_fwd_parity": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, vp]),
        "lvt_conv3d_fwd": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, vp]),
        "lvt_conv3d_bwd_data_uses_phase_kernel": (ci, [P(ConvGeom), ci]),
        "lvt_conv3d_pack_weight_phases": (ci, [P(ConvGeom), vp, ci, ci, vp, vp]),
        "lvt_conv3d_bwd_data_phases": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, vp]),
        "lvt_conv3d_bwd_data": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, vp]),
        "lvt_conv3d_bwd_weight_workspace_bytes": (sz, [P(ConvGeom)]),
        "lvt_conv3d_bwd_weight_fuses_bias": (ci, [P(ConvGeom), ci]),
        "lvt_conv3d_bwd_weight": (ci, [P(ConvGeom), vp, vp, vp, vp, ci, ci, ci, vp, sz, vp]),
        "lvt_convt4_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]),
        "lvt_colsum_workspace_bytes": (sz, [cll, ci]),
        "lvt_colsum": (ci, [vp, cll, ci, cll, vp, vp, sz, vp]),
        "lvt_vq_nearest_workspace_bytes": (sz, [cll, ci, ci]),
        "lvt_vq_nearest": (ci, [vp, cll, ci, ci, ci, ci, vp, vp, ci, ci, vp, sz, vp]),
        "lvt_vq_gather": (ci, [vp, vp, cll, ci, ci, ci, ci, vp, ci, vp]),
        "lvt_vq_ema_workspace_bytes": (sz, [cll, ci, ci, ci]),
        "lvt_vq_ema_accumulate": (ci, [vp, vp, cll, ci, ci, ci, ci, ci, vp, vp, sz, vp]),
        "lvt_vq_ema_finalize": (ci, [vp, ci, ci, ci, cf, cf, vp, vp, vp, vp]),
        "lvt_to_channels_last": (ci, [vp, ci, ci, cll, ci, ci, vp, vp, vp, vp]),
        "lvt_to_channels_first": (ci, [vp, ci, ci, cll, ci, ci, vp, vp, cf, cf, vp, vp]),
        "lvt_reduce_workspace_bytes": (sz, []),
        "lvt_mse_fwd": (ci, [vp, vp, cll, C.c_double, cf, vp, vp, sz, vp]),
        "lvt_mse_bwd": (ci, [vp, vp, cll, C.c_double, cf, vp, vp, ci, vp, vp]),
        "lvt_tanh_bwd": (ci, [vp, vp, cll, vp, vp]),
        "lvt_axpy": (ci, [vp, vp, cll, vp, cf, vp, vp]),
        "lvt_add_periodic": (ci, [vp, vp, cll, ci, ci, vp]),
        "lvt_layernorm_fwd": (ci, [vp, cll, ci, cf, vp, vp, vp, vp, vp, vp]),
        "lvt_layernorm_bwd_workspace_bytes": (sz, [ci]),
        "lvt_layernorm_bwd": (ci, [vp, vp, vp, vp, vp, cll, ci, vp, vp, vp, vp, vp, sz, vp]),
        "lvt_attn_softmax_fwd": (ci, [vp, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp]),
        "lvt_attn_softmax_bwd": (ci, [vp, vp, ci, ci, ci, cf, ci, ci, ci, vp, vp, vp, vp, vp]),
        "lvt_attn_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp]),
        "lvt_attn_planes_supported": (ci, [ci, ci, ci, ci, ci]),
        "lvt_attn_fwd_planes": (ci, [vp, cll, cll, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp]),
        "lvt_attn_bwd_planes_workspace_bytes": (sz, [ci, ci, ci, ci, ci, ci]),
        "lvt_attn_bwd_planes": (ci, [vp, cll, cll, vp, vp, vp, ci, ci, ci, ci, cf, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
        "lvt_attn_decode": (ci, [vp, cll, vp, vp, ci, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, vp, vp, cll, vp]),
        "lvt_decode_gather_codes": (ci, [vp, vp, vp, ci, ci, ci, vp, vp]),
        "lvt_decode_commit": (ci, [vp, ci, ci, vp, vp, vp]),
        "lvt_sample_categorical": (ci, [vp, cll, ci, cf, vp, vp, cll, vp, vp, cll, vp]),
        "lvt_embbag_fwd": (ci, [vp, cll, ci, cll, ci, P(ci), P(ci), vp, ci, vp, vp, vp, vp, vp]),
        "lvt_onehot_tn_workspace_bytes": (sz, [ci, ci, ci, cll]),
        "lvt_onehot_tn_gemm": (ci, [vp, ci, ci, P(ci), cll, cll, ci, cll, vp, cll, ci, vp, ci, vp, sz, vp]),
        "lvt_permute3": (ci, [vp, cll, cll, cll, ci, ci, ci, vp, vp]),
        "lvt_row_gather": (ci, [vp, vp, cll, ci, ci, vp, vp]),
        "lvt_slice_context": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, ci, cll, vp, vp, vp, vp, vp]),
        "lvt_xent_workspace_bytes": (sz, []),
        "lvt_xent_fwd": (ci, [vp, vp, cll, cll, ci, cll, ci, cll, cf, vp, vp, vp, vp, vp, sz, vp]),
        "lvt_xent_bwd": (ci, [vp, vp, cll, cll, ci, cll, ci, cll, vp, vp, vp, cf, vp, vp]),
        "lvt_adam_step": (ci, [P(OptEntry), ci, cf, cf, cf, ci, vp]),
        "lvt_rmsprop_step": (ci, [P(OptEntry), ci, cf, cf, cf, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return sigs


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    """Load (once) and return the ctypes handle.  Raises LvtError if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise LvtError("liblvt_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (or `make -C lvt_amd/csrc`); there is no CPU fallback" % _LIB_PATH)
        handle = C.CDLL(_LIB_PATH)
        handle.lvt_version.restype = C.c_int
        if handle.lvt_version() != ABI_VERSION:
            raise LvtError("liblvt_hip.so at %s has ABI version %d, this package binds version %d -- rebuild it "
                           "(`make -C lvt_amd/csrc`)" % (_LIB_PATH, handle.lvt_version(), ABI_VERSION))
        handle._lvt_sigs = _declare(handle)
        _lib = handle
    return _lib


# The library itself is stateless: every engine call carries its arithmetic in `flags` (MATH_F32 or not).  The default the
# Python wrappers pass is a host-side setting of this module (LVT_MATH environment variable, or set_math_mode()).
_math_mode = os.environ.get("LVT_MATH", "bf16x3")
if _math_mode not in ("f32", "bf16x3"):
    raise LvtError("LVT_MATH must be 'f32' or 'bf16x3' (got %r)" % _math_mode)


def set_math_mode(mode):
    """'bf16x3' (default: exact 3-way bf16 split of fp32 operands on the bf16 matrix cores, fp32 accumulation) or
    'f32' (plain fp32 MFMA): what the wrappers of lvt_amd.hip put into the `flags` of the calls they issue from now on."""
    global _math_mode
    if mode not in ("f32", "bf16x3"):
        raise LvtError("math mode must be 'f32' or 'bf16x3' (got %r)" % (mode,))
    _math_mode = mode


def get_math_mode():
    return _math_mode


def math_flag():
    return MATH_F32 if _math_mode == "f32" else 0


def declared_symbols():
    return sorted(lib()._lvt_sigs.keys())


def require(*tensors):
    """All tensors must be contiguous CUDA(HIP) tensors; the product path never runs on CPU."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise LvtError("lvt_amd kernels need tensors on a MI355X (got device %s); there is no CPU "
                           "fallback" % t.device)
        if not t.is_contiguous():
            raise LvtError("lvt_amd kernels need contiguous tensors")


def check(rc, what=""):
    if rc != 0:
        raise LvtError("%s failed (rc=%d): %s" % (what, rc, lib().lvt_last_error().decode()))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class KernelTimer:
    """Optional per-launch timing of engine calls with HIP events recorded on the launch stream
    (used by bench.py for the roofline block).  Disabled (None) by default: zero overhead."""

    def __init__(self):
        self.records = []          # (key, flops, start_event, end_event)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, key, flops, start):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((key, flops, start, ev))

    def summary(self):
        """-> {key: dict(launches, ms, flops)} after synchronising."""
        torch.cuda.synchronize()
        out = {}
        for key, flops, a, b in self.records:
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0})
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["flops"] += flops
        return out


TIMER = None     # set to a KernelTimer() to time every engine launch


_ws = {}


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag, stream), reused across calls on the same stream.  A larger request
    REPLACES the buffer, so its address must never be recorded into a hipGraph: callers that capture launches own their
    scratch (autoregressive/incremental.py), and asking for this one while a capture is running is an error."""
    if torch.cuda.is_current_stream_capturing():
        raise LvtError("binding.workspace(%r) requested during hipGraph capture: captured launches must use scratch "
                       "owned by the object that owns the graph" % (tag,))
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf

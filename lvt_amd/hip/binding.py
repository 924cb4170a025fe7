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
ABI_VERSION = 610           # lvt_version() of the library this module binds (argument lists below)
MATH_F32 = 1 << 16          # per-call arithmetic selectors of the engine entry points (include/lvt_hip.h)
MATH_F16X2 = 1 << 18
ONEHOT_DENSE = 1 << 19
WGRAD_DB_OF_X = 1 << 20
CONV_WEIGHT_IMAGE = 1 << 21   # the packed weight carries its f16x2 tile images behind it (lvt_conv3d_weight_images)


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
        ("a_amax", C.c_void_p), ("b_amax", C.c_void_p), ("a_amax2", C.c_void_p), ("b_amax2", C.c_void_p), ("c_amax", C.c_void_p),
    ]


class GemmP2Desc(C.Structure):
    """lvt_gemm_p2_desc (include/lvt_hip.h): the plane-fed NT GEMM of the f16x2 arithmetic."""
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("lda", C.c_longlong), ("a_planes", C.c_int), ("a_kb", C.c_int), ("a_skb", C.c_longlong),
        ("B", C.c_void_p), ("ldb", C.c_longlong),
        ("C", C.c_void_p), ("ldc", C.c_longlong),
        ("batch_outer", C.c_int), ("batch_inner", C.c_int),
        ("sA_o", C.c_longlong), ("sA_i", C.c_longlong), ("sB_o", C.c_longlong), ("sB_i", C.c_longlong),
        ("sC_o", C.c_longlong), ("sC_i", C.c_longlong),
        ("alpha", C.c_float), ("flags", C.c_int),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ldr", C.c_longlong),
        ("mask", C.c_void_p), ("ldm", C.c_longlong),
        ("a_amax", C.c_void_p), ("b_amax", C.c_void_p), ("c_amax", C.c_void_p),
        ("Cp", C.c_void_p), ("ldcp", C.c_longlong), ("cp_amax", C.c_void_p),
    ]


class P2PackEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld_src", C.c_longlong),
                ("ld_dst", C.c_longlong), ("transpose", C.c_int), ("amax", C.c_void_p),
                ("batch", C.c_int), ("bs_src", C.c_longlong), ("bs_dst", C.c_longlong)]


class AmaxEntry(C.Structure):
    _fields_ = [("x", C.c_void_p), ("n", C.c_longlong), ("out", C.c_void_p)]


class PackEntry(C.Structure):
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("kind", C.c_int), ("taps", C.c_int), ("Ci", C.c_int), ("Co", C.c_int),
                ("Ci_real", C.c_int), ("Co_real", C.c_int)]


class WeightImageEntry(C.Structure):
    _fields_ = [("wp", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("amax", C.c_void_p)]


class AmaxIO(C.Structure):
    """lvt_amax_io: device scalars with max |.| of the two operands (f16x2 mode) and of the result (optional)."""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p)]


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
        "lvt_vq_argmax_scores": (ci, [vp, cll, ci, cll, vp, ci, vp, vp]),
        "lvt_gemm_p2_f32": (ci, [P(GemmP2Desc), vp]),
        "lvt_p2_pack_multi": (ci, [P(P2PackEntry), ci, vp]),
        "lvt_layernorm_fwd_p2": (ci, [vp, cll, ci, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
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
        "lvt_amax": (ci, [vp, cll, vp, vp]),
        "lvt_amax_multi": (ci, [P(AmaxEntry), ci, vp]),
        "lvt_conv3d_pack_weights_multi": (ci, [P(PackEntry), ci, vp]),
        "lvt_conv3d_weight_image_bytes": (sz, [ci, ci]),
        "lvt_conv3d_weight_images": (ci, [P(WeightImageEntry), ci, vp]),
        "lvt_amax_merge": (ci, [vp, vp, vp, vp]),
        "lvt_conv3d_fwd_parity": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, P(AmaxIO), vp]),
        "lvt_conv3d_fwd": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, P(AmaxIO), vp]),
        "lvt_conv3d_bwd_data_uses_phase_kernel": (ci, [P(ConvGeom), ci]),
        "lvt_conv3d_pack_weight_phases": (ci, [P(ConvGeom), vp, ci, ci, vp, vp]),
        "lvt_conv3d_bwd_data_phases": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, P(AmaxIO), vp]),
        "lvt_conv3d_bwd_data": (ci, [P(ConvGeom), vp, vp, vp, vp, vp, vp, ci, P(AmaxIO), vp]),
        "lvt_conv3d_bwd_weight_workspace_bytes": (sz, [P(ConvGeom)]),
        "lvt_conv3d_bwd_weight_fuses_bias": (ci, [P(ConvGeom), ci]),
        "lvt_conv3d_bwd_weight": (ci, [P(ConvGeom), vp, vp, vp, vp, ci, ci, ci, P(AmaxIO), vp, sz, vp]),
        "lvt_convt4_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, ci, vp, vp]),
        "lvt_colsum_workspace_bytes": (sz, [cll, ci]),
        "lvt_colsum": (ci, [vp, cll, ci, cll, vp, vp, sz, vp]),
        "lvt_vq_nearest_workspace_bytes": (sz, [cll, ci, ci]),
        "lvt_vq_nearest": (ci, [vp, cll, ci, ci, ci, ci, vp, vp, ci, ci, vp, sz, vp]),
        "lvt_vq_gather": (ci, [vp, vp, cll, ci, ci, ci, ci, vp, ci, vp]),
        "lvt_vq_ema_workspace_bytes": (sz, [cll, ci, ci, ci]),
        "lvt_vq_ema_accumulate": (ci, [vp, vp, cll, ci, ci, ci, ci, ci, vp, vp, sz, vp]),
        "lvt_vq_ema_finalize": (ci, [vp, ci, ci, ci, cf, cf, vp, vp, vp, vp]),
        "lvt_to_channels_last": (ci, [vp, ci, ci, cll, ci, ci, vp, vp, vp, vp, vp]),
        "lvt_to_channels_first": (ci, [vp, ci, ci, cll, ci, ci, vp, vp, cf, cf, vp, vp]),
        "lvt_reduce_workspace_bytes": (sz, []),
        "lvt_mse_fwd": (ci, [vp, vp, cll, C.c_double, cf, vp, vp, sz, vp]),
        "lvt_mse_bwd": (ci, [vp, vp, cll, C.c_double, cf, vp, vp, ci, vp, vp, vp]),
        "lvt_l1_fwd": (ci, [vp, vp, cll, C.c_double, cf, vp, vp, sz, vp]),
        "lvt_l1_bwd": (ci, [vp, vp, cll, C.c_double, cf, vp, vp, ci, vp, vp, vp]),
        "lvt_tanh_bwd": (ci, [vp, vp, cll, vp, vp, vp]),
        "lvt_axpy": (ci, [vp, vp, cll, vp, cf, vp, vp]),
        "lvt_add_periodic": (ci, [vp, vp, cll, ci, ci, vp]),
        "lvt_layernorm_fwd": (ci, [vp, cll, ci, cf, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "lvt_layernorm_bwd_workspace_bytes": (sz, [ci]),
        "lvt_layernorm_bwd": (ci, [vp, vp, vp, vp, vp, cll, ci, vp, vp, vp, vp, vp, vp, sz, vp]),
        "lvt_attn_softmax_fwd": (ci, [vp, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp]),
        "lvt_attn_softmax_bwd": (ci, [vp, vp, ci, ci, ci, cf, ci, ci, ci, vp, vp, vp, vp, vp]),
        "lvt_attn_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp]),
        "lvt_attn_planes_supported": (ci, [ci, ci, ci, ci, ci]),
        "lvt_attn_fwd_planes": (ci, [vp, cll, cll, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, vp]),
        "lvt_attn_bwd_planes_workspace_bytes": (sz, [ci, ci, ci, ci, ci, ci]),
        "lvt_attn_bwd_planes": (ci, [vp, cll, cll, vp, vp, vp, ci, ci, ci, ci, cf, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
        "lvt_attn_flash_supported": (ci, [ci, ci, ci, ci, ci]),
        "lvt_attn_fwd_flash": (ci, [vp, vp, vp, cll, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, vp]),
        "lvt_attn_bwd_flash_workspace_bytes": (sz, [ci, ci, ci, ci, ci, ci]),
        "lvt_attn_bwd_flash": (ci, [vp, vp, vp, vp, cll, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, vp,
                                    vp, vp, vp, vp, sz, vp]),
        "lvt_attn_decode": (ci, [vp, cll, vp, vp, ci, ci, ci, ci, ci, cf, vp, vp, vp, ci, ci, ci, vp, vp, cll, vp]),
        "lvt_decode_gather_codes": (ci, [vp, vp, vp, ci, ci, ci, vp, vp]),
        "lvt_decode_commit": (ci, [vp, ci, ci, vp, vp, vp]),
        "lvt_sample_categorical": (ci, [vp, cll, ci, cf, vp, vp, cll, vp, vp, cll, vp]),
        "lvt_embbag_fwd": (ci, [vp, cll, ci, cll, ci, P(ci), P(ci), vp, ci, vp, vp, vp, vp, vp]),
        "lvt_onehot_tn_workspace_bytes": (sz, [ci, ci, ci, cll]),
        "lvt_onehot_tn_is_gather": (ci, [ci, ci, ci, cll, vp, ci]),
        "lvt_onehot_tn_gemm": (ci, [vp, ci, ci, P(ci), cll, cll, ci, cll, vp, cll, ci, vp, ci, vp, vp, sz, vp]),
        "lvt_permute3": (ci, [vp, cll, cll, cll, ci, ci, ci, vp, vp]),
        "lvt_row_gather": (ci, [vp, vp, cll, ci, ci, vp, vp]),
        "lvt_slice_context": (ci, [vp, ci, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, ci, cll, vp, vp, vp, vp, vp]),
        "lvt_xent_workspace_bytes": (sz, []),
        "lvt_xent_fwd": (ci, [vp, vp, cll, cll, ci, cll, ci, cll, cf, vp, vp, vp, vp, vp, sz, vp]),
        "lvt_xent_bwd": (ci, [vp, vp, cll, cll, ci, cll, ci, cll, vp, vp, vp, cf, vp, vp, vp]),
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


# The library itself is stateless: every engine call carries its arithmetic in `flags`.  The default the Python wrappers
# pass is a host-side setting of this module (LVT_MATH environment variable, or set_math_mode()).
MATH_MODES = ("f32", "bf16x3", "f16x2")
_math_mode = os.environ.get("LVT_MATH", "f16x2")
if _math_mode not in MATH_MODES:
    raise LvtError("LVT_MATH must be one of %s (got %r)" % (MATH_MODES, _math_mode))


def set_math_mode(mode):
    """What the wrappers of lvt_amd.hip put into the `flags` of the engine calls they issue from now on:
    'f16x2'  (default since round 4) 2-way fp16 split after an exact power-of-two scale from the operand's max |.|, three
             fp16 MFMAs per block (the wrappers carry the max |.| of every engine operand along: `amax_of`, `new_amax`);
    'bf16x3' exact 3-way bf16 split of the fp32 operands, six bf16 MFMAs per block (the default of rounds 1-3);
    'f32'    plain fp32 MFMA.  All three are fp32 in / fp32 out with fp32 accumulation."""
    global _math_mode
    if mode not in MATH_MODES:
        raise LvtError("math mode must be one of %s (got %r)" % (MATH_MODES, mode))
    _math_mode = mode


def get_math_mode():
    return _math_mode


def math_flag():
    return MATH_F32 if _math_mode == "f32" else (MATH_F16X2 if _math_mode == "f16x2" else 0)


def f16x2():
    return _math_mode == "f16x2"


# ---- max |.| bookkeeping of the f16x2 arithmetic -------------------------------------------------------------------
# Every operand of an f16x2 engine launch needs a device scalar >= max |operand|.  Engine launches report the max of what
# they write (c_amax), the helper kernels that produce engine operands do the same, and anything else falls back to one
# lvt_amax pass.  The scalar rides on the tensor object as `_lvt_amax = (slot, version, epoch, data_ptr)`: a torch in-place
# op bumps `_version`, `p.data = ...` changes the pointer, and whatever rewrites tensors behind torch's back bumps the epoch
# (the fused optimizers after a step; the meta-architectures at the start of every forward, which also covers parameters
# edited through `.data` between passes) -- any of them invalidates the record.
_amax_pool, _amax_pos, _epoch = None, 0, 0


def bump_epoch():
    """Called by whatever rewrites tensors behind torch's back (the fused optimizers): forget every cached max |.|."""
    global _epoch
    _epoch += 1


def amax_slot(device):
    """A fresh zeroed float32 device scalar (a 1-element view of a pool that is zero-filled 4096 slots at a time)."""
    global _amax_pool, _amax_pos
    if _amax_pool is None or _amax_pos >= _amax_pool.numel() or _amax_pool.device != device:
        _amax_pool, _amax_pos = torch.zeros(4096, dtype=torch.float32, device=device), 0
    s = _amax_pool[_amax_pos:_amax_pos + 1]
    _amax_pos += 1
    return s


def set_amax(t, slot):
    t._lvt_amax = (slot, t._version, _epoch, t.data_ptr())
    return t


def drop_amax(t):
    """EVERY wrapper of a kernel that rewrites a tensor in place through its raw pointer must call this (torch's `_version`
    does not see such writes): the record of `t` and of the tensor it is a view of are forgotten.  LVT_AMAX_CHECK=1 verifies
    every record that is used against the tensor (a debugging aid: it synchronises with the device on every engine call)."""
    if getattr(t, "_lvt_amax", None) is not None:
        t._lvt_amax = None
    base = getattr(t, "_base", None)
    if base is not None and getattr(base, "_lvt_amax", None) is not None:
        base._lvt_amax = None


AMAX_CHECK = bool(os.environ.get("LVT_AMAX_CHECK"))


def _checked(slot, t):
    if AMAX_CHECK:
        have, real = float(slot), float(t.detach().abs().max()) if t.numel() else 0.0
        if not have >= real:
            raise LvtError("stale max |.| record: %g recorded, %g in the tensor of shape %s (an in-place kernel wrapper that "
                           "did not call binding.drop_amax?)" % (have, real, tuple(t.shape)))
    return slot


def _valid_amax(t):
    rec = getattr(t, "_lvt_amax", None)
    if rec is not None and rec[1] == t._version and rec[2] == _epoch and rec[3] == t.data_ptr():
        return rec[0]
    return None


def new_amax(t):
    """Attach a fresh zeroed slot to `t` (about to be written by a launch that reports max |t| into it) and return it."""
    slot = amax_slot(t.device)
    set_amax(t, slot)
    return slot


def amax_of(t):
    """Device scalar >= max |t|: the record on `t`, or on the tensor it is a view of (the max of the whole is a bound for
    the part), else one lvt_amax pass whose result is cached on `t`."""
    slot = _valid_amax(t)
    if slot is not None:
        return _checked(slot, t)
    base = t._base
    if base is not None:
        slot = _valid_amax(base)
        if slot is not None:
            return _checked(slot, t)
    src = t
    if not t.is_contiguous():
        if base is None or not base.is_contiguous():
            raise LvtError("amax_of: non-contiguous tensor without a contiguous base")
        src = base                    # strided view: scan (and cache on) the whole it was cut from
    require(src)
    slot = amax_slot(src.device)
    check(lib().lvt_amax(ptr(src), src.numel(), ptr(slot), stream_ptr()), "lvt_amax")
    AMAX_FALLBACKS[0] += 1
    if AMAX_TRACE is not None:          # diagnostic: who needed a stand-alone pass (call sites that should carry a record)
        import traceback
        fr = [f for f in traceback.extract_stack()[:-1] if "binding.py" not in f.filename][-3:]
        key = " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr)) + " %s" % (tuple(src.shape),)
        AMAX_TRACE[key] = AMAX_TRACE.get(key, 0) + 1
    set_amax(src, slot)
    return slot


def out_amax(t):
    """Pointer argument for a helper kernel that can report max |t| of the tensor it is about to write: a fresh slot
    attached to `t` in f16x2 mode, NULL otherwise."""
    return ptr(new_amax(t)) if _math_mode == "f16x2" else None


def amax_prefetch(tensors):
    """max |.| of many tensors (the weights of a model at the start of a pass) in ONE launch per 64 instead of one
    lvt_amax pass each; tensors that still hold a valid record are skipped.  No-op outside f16x2 mode."""
    if _math_mode != "f16x2":
        return
    todo = [t for t in tensors if t is not None and t.is_cuda and _valid_amax(t) is None and t.is_contiguous() and t.numel() > 0]
    if not todo:
        return
    arr = (AmaxEntry * len(todo))()
    for e, t in zip(arr, todo):
        slot = amax_slot(t.device)
        e.x, e.n, e.out = t.data_ptr(), t.numel(), slot.data_ptr()
        set_amax(t, slot)
    check(lib().lvt_amax_multi(arr, len(todo), stream_ptr()), "lvt_amax_multi")


def prefetch_module_weights(module):
    """max |.| of every matrix-shaped parameter of `module` (the operands its engine launches will read) in one launch
    per 64 tensors: what a meta-architecture calls at the start of a pass, right after bump_epoch().  Attention modules
    contribute their packed (3, na, d, da) q/k/v buffer, which is what the launches address."""
    if _math_mode != "f16x2":
        return
    # the walk over the module tree is planned once per model (0.5 ms of generators per pass otherwise); the plan holds modules and
    # parameter NAMES, so a parameter that is replaced later is still the one that gets scanned
    plan = module.__dict__.get("_lvt_prefetch_plan")
    if plan is None:
        plan = []
        for m in module.modules():
            packed = getattr(m, "packed_qkv", None)
            skip = ("w_q", "w_k", "w_v") if packed is not None else ()
            names = [name for name, p in m.named_parameters(recurse=False)
                     if name not in skip and (p.dim() >= 2 or isinstance(m, torch.nn.LayerNorm))]
            if packed is not None or names:      # (LayerNorm weight / bias: the a-priori bound of its output, lvt_layernorm_fwd)
                plan.append((m, packed is not None, names))
        module.__dict__["_lvt_prefetch_plan"] = plan
    todo = []
    for m, has_packed, names in plan:
        if has_packed:
            todo.append(m.packed_qkv())
        params = m._parameters
        for name in names:
            todo.append(params[name])
    amax_prefetch(todo)


def amax_merged(*tensors):
    """A slot holding the max over the records of several tensors (an operand that spans them: batched launches whose
    batch strides are address differences)."""
    slots = [amax_of(t) for t in tensors]
    out = amax_slot(tensors[0].device)
    for i in range(0, len(slots), 2):
        check(lib().lvt_amax_merge(ptr(slots[i]), ptr(slots[i + 1]) if i + 1 < len(slots) else None, ptr(out), stream_ptr()),
              "lvt_amax_merge")
    return out


AMAX_TRACE = {} if os.environ.get("LVT_AMAX_TRACE") else None
AMAX_FALLBACKS = [0]        # diagnostic: stand-alone lvt_amax passes issued so far (bench.py reports them per step)


def amax_io(a=None, b=None, c=None):
    """lvt_amax_io for an engine call: operands a, b (tensors; looked up only in f16x2 mode), result c (tensor that the
    launch is about to write).  Returns None outside f16x2 mode (the entry points take a NULL pointer)."""
    if _math_mode != "f16x2":
        return None
    io = AmaxIO()
    io.a = amax_of(a).data_ptr() if a is not None else None
    io.b = amax_of(b).data_ptr() if b is not None else None
    io.c = new_amax(c).data_ptr() if c is not None else None
    return io


def io_ref(io):
    return C.byref(io) if io is not None else None


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


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object through
    several Python layers (9 us per call, 1.3 + 3.6 ms of host time per DSFVT train step at one call per launch:
    tools/profile/host_profile.py); the raw accessor answers the same question in well under a microsecond."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
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
RELU_TRACE = None   # diagnostic (tests/util_relu.py): a list that receives, in execution order, the bool mask (y > 0) of every
                    # ReLU of the transformer path (FFN hidden of each attention layer, U_k of the channel predictor)


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

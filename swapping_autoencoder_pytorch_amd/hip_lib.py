"""ctypes binding of the C-ABI hot-path library (include/sae_hip.h).

The product loads exactly one shared object: ``csrc/libsae_hip.so`` built for gfx950 by
``csrc/build.py`` (``__graft_entry__.build()``).  There is no CPU implementation behind this
module: if the library is missing, or a tensor handed to an op is not resident on a GPU, the
call raises.  (The reference decided native-vs-fallback with ``util.is_custom_kernel_supported``,
util/util.py:432-436, which raises on ROCm; that gate is not consulted here.)

``SaeLibrary`` is parametrised by path and symbol prefix only so that the test-suite can bind the
same signatures of the CPU oracle (``oracle_*``) and of the emulator build of the kernels
(tests/emu) for checking; the product never does.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "csrc", "libsae_hip.so")

SAE_CONV_FWD, SAE_CONV_DGRAD, SAE_CONV_WGRAD = 0, 1, 2

_f32p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_f32 = C.c_float
_stream = C.c_void_p


class ConvDesc(C.Structure):
    """Mirror of ``sae_conv2d_desc``."""
    _fields_ = [("n", _i64), ("c", _i64), ("h", _i64), ("w", _i64),
                ("m", _i64), ("oh", _i64), ("ow", _i64),
                ("kh", _i32), ("kw", _i32), ("stride", _i32), ("pad", _i32),
                ("w_stride_m", _i64), ("w_stride_c", _i64),
                ("prepped", C.c_void_p), ("prepped_floats", _i64), ("prepped_layout", _i64)]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_[:13])


class ConvMod(C.Structure):
    """Mirror of ``sae_conv2d_mod``: optional per-(sample, channel) activation factors and per-channel weight factors
    of a style-modulated convolution (device pointers, 0 = absent)."""
    _fields_ = [("x_scale", C.c_void_p), ("y_scale", C.c_void_p), ("wm_scale", C.c_void_p), ("wc_scale", C.c_void_p)]


ABI_VERSION = 12     # include/sae_hip.h: SAE_ABI_VERSION

_SIGNATURES = {
    "abi_version": (C.c_int, []),
    "last_error": (C.c_char_p, []),
    "set_conv_math": (C.c_int, [_i32]),
    "get_conv_math": (C.c_int, []),
    "upfirdn2d_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i32, _i32,
                                _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _stream]),
    "upfirdn2d_epilogue_workspace": (_i64, [_i64, _i64, _i64, _i64, _i32]),
    "upfirdn2d_epilogue_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                         _f32p, _f32, _f32, _f32p, _i64, _i32, _f32p, _i64, _stream]),
    "upfirdn2d_noise_bias_act_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                                               _f32p, _f32p, _f32p, _i64, _f32, _f32, _stream]),
    "bias_act_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i32, _i32, _f32, _f32, _stream]),
    "bias_act_bwd_workspace": (_i64, [_i64, _i64, _i64]),
    "bias_act_bwd_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _f32, _f32, _stream]),
    "noise_bias_act_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _f32, _f32, _stream]),
    "noise_bias_act_bwd_workspace": (_i64, [_i64, _i64, _i64]),
    "noise_bias_act_bwd_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _f32,
                                         _f32, _stream]),
    "plane_scale_dot_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _stream]),
    "plane_scale_dot_act_workspace": (_i64, [_i64, _i64]),
    "plane_scale_dot_act_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64,
                                          _f32, _f32, _stream]),
    "weight_demod_f32": (C.c_int, [_f32p, _f32p, _i64, _i64, C.c_float, C.c_float, _stream]),
    "weight_demod_bwd_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, C.c_float, _stream]),
    "random_crop_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _i64, _stream]),
    "random_crop_bwd_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _i64, _stream]),
    "reflect_pad_f32": (C.c_int, [_f32p, _f32p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _stream]),
    "reflect_pad_adj_f32": (C.c_int, [_f32p, _f32p, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _stream]),
    "conv2d_workspace": (_i64, [C.POINTER(ConvDesc), _i32]),
    "conv2d_fwd_f32": (C.c_int, [_f32p, _f32p, _f32p, C.POINTER(ConvDesc), _f32, _f32p, _i64, _stream]),
    "conv2d_fwd_bias_act_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.POINTER(ConvDesc), _f32, _f32, _f32, _f32p, _i64,
                                          _stream]),
    "conv2d_fwd_residual_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.POINTER(ConvDesc), _f32, _f32, _f32p, _i64, _stream]),
    "conv2d_dgrad_f32": (C.c_int, [_f32p, _f32p, _f32p, C.POINTER(ConvDesc), _f32, _f32p, _i64, _stream]),
    "conv2d_wgrad_f32": (C.c_int, [_f32p, _f32p, _f32p, C.POINTER(ConvDesc), _f32, _f32p, _i64, _stream]),
    "conv2d_wprep_query": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvMod), _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "conv2d_wprep_f32": (C.c_int, [_f32p, C.POINTER(ConvDesc), C.POINTER(ConvMod), _i32, _f32, _f32p, _i64, _stream]),
    "modconv2d_fwd_f32": (C.c_int, [_f32p, _f32p, _f32p, C.POINTER(ConvDesc), C.POINTER(ConvMod), _f32, _f32p, _i64, _stream]),
    "modconv2d_fwd_noise_bias_act_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.POINTER(ConvDesc), C.POINTER(ConvMod),
                                                   _f32, _f32, _f32, _f32p, _i64, _stream]),
    "modconv2d_dgrad_f32": (C.c_int, [_f32p, _f32p, _f32p, C.POINTER(ConvDesc), C.POINTER(ConvMod), _f32, _f32p, _i64, _stream]),
    "modconv2d_wgrad_f32": (C.c_int, [_f32p, _f32p, _f32p, C.POINTER(ConvDesc), C.POINTER(ConvMod), _f32, _f32p, _i64, _stream]),
    "gemm_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _stream]),
    "gemm_workspace": (_i64, [_i64, _i64, _i64]),
    "gemm_ws_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _f32p, _i64,
                              _stream]),
    "add_scale_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _f32, _stream]),
    "l2_normalize_f32": (C.c_int, [_f32p, _f32p, _i64, _i64, _i64, _f32, _stream]),
    "l2_normalize_bwd_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _f32, _stream]),
    "plane_affine_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _stream]),
    "plane_affine_bwd_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _stream]),
    "softplus_mean_f32": (C.c_int, [_f32p, _f32p, _i64, _i64, _f32, _stream]),
    "softplus_mean_bwd_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _f32, _stream]),
    "adam_multi_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _i64, C.c_double,
                                 C.c_double, C.c_double, C.c_double, C.c_double, _stream]),
    "adam_multi_dev_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _i64, C.c_double,
                                     C.c_double, C.c_double, C.c_double, C.c_double, _stream]),
    "upsample2x_bilinear_add_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _f32, _stream]),
    "upsample2x_bilinear_bwd_f32": (C.c_int, [_f32p, _f32p, _i64, _i64, _i64, _f32, _stream]),
    "wino_gemm_workspace": (_i64, [_i64, _i64, _i64, _i64, _i64]),
    "wino_gemm_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _f32p, _i64, _stream]),
    "wino_gy_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _stream]),
    "wino_wgrad_gemm_workspace": (_i64, [_i64, _i64, _i64, _i64, _i64]),
    "wino_wgrad_gemm_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _f32p, _i64, _stream]),
    "wino_wgrad_output_f32": (C.c_int, [_f32p, _f32p, _i64, _i64, _i64, _i64, _f32, _stream]),
    "wino_weights_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i32, _f32, _stream]),
    "wino_input_f32": (C.c_int, [_f32p, _f32p, _f32p, _i64, _i64, _i64, _i32, _stream]),
    "wino_output_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i32, _f32, _f32, _stream]),
    "wino_fused_weights_floats": (_i64, [_i64, _i64]),
    "wino_fused_weights_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i32, _f32, _stream]),
    "wino_fused_conv_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _i32,
                                      _i32, _f32, _f32, _stream]),
    "wino_fused_wgrad_workspace": (_i64, [_i64, _i64, _i64, _i64, _i64, _i32]),
    "wino_fused_wgrad_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _i32, _i64, _i64, _f32,
                                       _f32p, _i64, _stream]),
    "s2wino_weights_floats": (_i64, [_i64, _i64]),
    "s2wino_weights_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i32, _f32, _stream]),
    "s2wino_dgrad_f32": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _i64, _i64, _i64, _i64, _i64, _stream]),
}

EXPORTED_SYMBOLS = tuple("sae_" + name for name in _SIGNATURES)


class SaeError(RuntimeError):
    pass


class SaeLibrary:
    def __init__(self, path=DEFAULT_LIBRARY, prefix="sae_", device_only=True):
        if not os.path.exists(path):
            raise SaeError(
                "HIP hot-path library not found at %s - build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback." % path)
        self.path = path
        self.prefix = prefix
        self.device_only = device_only
        if device_only:
            # PyTorch-ROCm ships its own libamdhip64: it must be in the process BEFORE this library is loaded, so that
            # both bind to the same HIP runtime (loading ours first pulls in the system runtime, whose launches then
            # fail with "no ROCm-capable device" on pointers the other runtime allocated)
            import torch  # noqa: F401
        self._dll = C.CDLL(path)
        self._fn = {}
        # the version check runs BEFORE the other symbols are bound: a stale prebuilt library then fails with this
        # message instead of a bare AttributeError on the first entry point it lacks
        ver = getattr(self._dll, prefix + "abi_version", None)
        got = ver() if ver is not None else None
        if got != ABI_VERSION:
            raise SaeError("%s has ABI version %s, this package needs %d (include/sae_hip.h: SAE_ABI_VERSION) - "
                           "rebuild it: python -c 'import __graft_entry__ as g; g.build()'" % (path, got, ABI_VERSION))
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(self._dll, prefix + name)
            fn.restype = restype
            fn.argtypes = argtypes
            self._fn[name] = fn

    def last_error(self):
        msg = self._fn["last_error"]()
        return msg.decode("utf-8", "replace") if msg else ""

    def query(self, name, *args):
        return self._fn[name](*args)

    def call(self, name, *args):
        rc = self._fn[name](*args)
        if rc != 0:
            raise SaeError("%s%s failed (%d): %s" % (self.prefix, name, rc, self.last_error()))

    # -- tensor plumbing --------------------------------------------------------------------
    def check(self, *tensors):
        """Every tensor handed to a kernel must be fp32, contiguous and (product) on a GPU."""
        import torch
        for t in tensors:
            if t is None:
                continue
            if t.dtype != torch.float32:
                raise SaeError("hot-path kernels are fp32 only, got %s" % t.dtype)
            if not t.is_contiguous():
                raise SaeError("internal error: non-contiguous tensor reached the C-ABI")
            if self.device_only and not t.is_cuda:
                raise SaeError(
                    "tensor on %s: the MI355X hot path has no CPU implementation (move the model "
                    "and data to a GPU)" % t.device)

    def stream(self, t):
        """Raw handle of the CURRENT stream of the tensor's device (what the reference's ops launch on).  The raw query
        is one C call; `torch.cuda.current_stream(dev).cuda_stream` builds a Stream object per launch (5 us, 330 launches
        per iteration of the 32 x 32 configuration)."""
        if t.is_cuda:
            import torch
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            if raw is not None:
                return raw(t.device.index if t.device.index is not None else torch.cuda.current_device())
            return torch.cuda.current_stream(t.device).cuda_stream
        return None


_LIB = None


def get():
    """The product's library instance (loaded on first use; raises if it is not built)."""
    global _LIB
    if _LIB is None:
        _LIB = SaeLibrary()
    return _LIB


def ptr(t):
    return None if t is None else t.data_ptr()


CONV_MATH_MODES = {"f32": 0, "bf16x6": 1}


def set_conv_math(mode):
    """Arithmetic of the 3x3 conv kernels: "f32" (default; v_mfma_f32_32x32x2_f32, an exact fp32 fma chain)
    or "bf16x6" (operands split exactly into three bf16 pieces, six bf16 MFMAs per product block, fp32
    accumulate; same error class, 1.3-2x faster).  include/sae_hip.h: sae_set_conv_math."""
    if mode not in CONV_MATH_MODES:
        raise SaeError("conv math must be one of %s, got %r" % (sorted(CONV_MATH_MODES), mode))
    get().call("set_conv_math", CONV_MATH_MODES[mode])


def get_conv_math():
    code = get().query("get_conv_math")
    return next(k for k, v in CONV_MATH_MODES.items() if v == code)

"""ctypes binding of libcnn_amd.so (include/cnn_amd.h).

This is plumbing for tests and bench.py: it passes raw device pointers (torch tensors' data_ptr()) and the
current HIP stream through the C ABI.  There is NO fallback: if the shared library is missing or a call returns
non-zero, a CnnAmdError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CNN_AMD_LIB: another build of the same ABI -- tools/ load the measurement build (make -C cnn_amd/csrc measure: libcnn_amd_measure.so,
# the only build in which the result-changing timing switches exist); the product path is the in-tree libcnn_amd.so
LIB_PATH = os.environ.get("CNN_AMD_LIB") or os.path.join(_HERE, "lib", "libcnn_amd.so")


class CnnAmdError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mirror of cnn_conv2d_desc"""

    _fields_ = [(n, C.c_int) for n in ("B", "Ci", "H", "W", "Co", "k", "s", "pad", "flags")]


POOL_MASK_PACKED = 1  # CNN_CONV2D_POOL_MASK_PACKED (include/cnn_amd.h)


_lib = None

# name -> (restype, argtypes); the single source for the "exports every declared symbol" test
_P = C.c_void_p
_D = C.POINTER(ConvDesc)
SIGNATURES = {
    "cnn_amd_abi_version": (C.c_int, []),
    "cnn_amd_last_error": (C.c_char_p, []),
    "cnn_amd_device_arch": (C.c_char_p, []),
    "cnn_amd_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "cnn_amd_get_option": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "cnn_amd_measure_build": (C.c_int, []),
    "cnn_amd_kernel_timing_enable": (C.c_int, [C.c_int, C.c_char_p]),
    "cnn_amd_kernel_timing_sampling": (C.c_int, [C.c_int]),
    "cnn_amd_kernel_timing_report": (C.c_longlong, [C.c_char_p, C.c_size_t]),
    "cnn_amd_timing_span_begin": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cnn_amd_timing_span_end": (C.c_int, [C.c_void_p]),
    "cnn_conv2d_out_dim": (C.c_int, [C.c_int] * 4),
    "cnn_maxpool2d_out_dim": (C.c_int, [C.c_int] * 3),
    "cnn_conv2d_workspace_bytes": (C.c_size_t, [_D]),
    "cnn_conv2d_autotune": (C.c_int, [_D, _P]),
    "cnn_conv2d_forward": (C.c_int, [_D, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_conv2d_forward_relu": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_conv2d_relu_maxpool2_supported": (C.c_int, [_D]),
    "cnn_conv2d_pool_mask_packed_supported": (C.c_int, [_D]),
    "cnn_conv2d_pool_mask_bytes": (C.c_size_t, [_D]),
    "cnn_conv2d_pool_mask_unpack": (C.c_int, [_D, _P, _P, _P]),
    "cnn_conv2d_relu_only_supported": (C.c_int, [_D]),
    "cnn_conv2d_relu_maxpool2_forward": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_conv2d_relu_maxpool2_forward_prepared": (C.c_int, [_D, _P, _P, _P, _P, _P]),
    "cnn_conv2d_backward_pooled2_prepared": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P, C.c_int]),
    "cnn_conv2d_backward_weight_pooled2": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P]),
    "cnn_conv2d_backward_weight_pooled2_sgd": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, C.c_float, C.c_float, _P, _P, _P,
                                                          C.c_size_t, _P]),
    "cnn_conv2d_backward_data_pooled2": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_conv2d_backward_data_pooled2_prepared": (C.c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "cnn_conv2d_prepared_bytes": (C.c_size_t, [_D]),
    "cnn_conv2d_prepare_filters": (C.c_int, [C.c_int, _D, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_void_p), _P]),
    "cnn_conv2d_forward_prepared": (C.c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "cnn_conv2d_backward_data_prepared": (C.c_int, [_D, _P, _P, _P, _P]),
    "cnn_conv2d_backward_prepared": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P, C.c_int]),
    "cnn_conv2d_backward_prepared_relu": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P, C.c_int]),
    "cnn_conv2d_backward_data_relu": (C.c_int, [_D, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_conv2d_backward_data_relu_prepared": (C.c_int, [_D, _P, _P, _P, _P, _P]),
    "cnn_conv2d_backward_weight": (C.c_int, [_D, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P]),
    "cnn_conv2d_backward_data": (C.c_int, [_D, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_conv2d_backward_workspace_bytes": (C.c_size_t, [_D]),
    "cnn_conv2d_backward": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P, C.c_int]),
    "cnn_amd_side_stream_join": (C.c_int, [_P]),
    "cnn_amd_side_stream_get": (C.c_int, [C.POINTER(C.c_void_p)]),
    "cnn_amd_flush_reduces": (C.c_int, [_P]),
    "cnn_grad_cam": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "cnn_amd_publish_next_kernel": (C.c_int, [_P]),
    "cnn_linear_forward_softmax_xent_dx": (C.c_int, [_P] * 9 + [C.c_int] * 4 + [_P]),
    "cnn_amd_wait_published": (C.c_int, [_P]),
    "cnn_amd_published_is_last": (C.c_int, [_P]),
    "cnn_conv2d_im2col_workspace_bytes": (C.c_size_t, [_D]),
    "cnn_conv2d_forward_im2col": (C.c_int, [_D, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_conv2d_backward_weight_im2col": (C.c_int, [_D, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P]),
    "cnn_conv2d_backward_data_im2col": (C.c_int, [_D, _P, _P, _P, _P, C.c_size_t, _P]),
    "cnn_maxpool2d_forward": (C.c_int, [_P, _P, _P] + [C.c_int] * 6 + [_P]),
    "cnn_maxpool2d_backward": (C.c_int, [_P, _P, _P] + [C.c_int] * 6 + [_P]),
    "cnn_maxpool2d_backward_relu": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 6 + [_P]),
    "cnn_relu_forward": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cnn_relu_backward": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cnn_dropout_forward": (C.c_int, [_P, _P] + [C.c_int] * 6 + [C.c_float, _P]),
    "cnn_dropout_backward": (C.c_int, [_P] + [C.c_int] * 5 + [_P]),
    "cnn_batch_stager_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]),
    "cnn_batch_stager_create_u8": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "cnn_batch_stager_destroy": (C.c_int, [_P]),
    "cnn_batch_stager_acquire": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "cnn_batch_stager_submit": (C.c_int, [_P, C.c_int, C.POINTER(C.c_void_p)]),
    "cnn_batch_stager_wait": (C.c_int, [_P, C.c_int, _P]),
    "cnn_batch_stager_release": (C.c_int, [_P, C.c_int, _P]),
    "cnn_linear_forward": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "cnn_linear_backward": (C.c_int, [_P] * 6 + [C.c_int] * 3 + [C.c_float, _P]),
    "cnn_linear_backward_relu": (C.c_int, [_P] * 6 + [C.c_int] * 3 + [C.c_float, _P]),
    "cnn_batchnorm2d_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "cnn_batchnorm2d_forward": (C.c_int, [_P] * 8 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, _P, C.c_size_t, _P]),
    "cnn_batchnorm2d_forward_relu": (C.c_int, [_P] * 9 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, _P, C.c_size_t, _P]),
    "cnn_batchnorm2d_backward": (C.c_int, [_P] * 7 + [C.c_int] * 4 + [C.c_float, _P, C.c_size_t, _P]),
    "cnn_batchnorm2d_forward_relu_pool_supported": (C.c_int, [C.c_int] * 4),
    "cnn_batchnorm2d_forward_relu_pool": (C.c_int, [_P] * 11 + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_int, _P, C.c_size_t, _P]),
    "cnn_batchnorm2d_backward_pooled_supported": (C.c_int, [C.c_int] * 4),
    "cnn_batchnorm2d_backward_pooled": (C.c_int, [_P] * 10 + [C.c_int] * 4 + [C.c_float, _P, C.c_size_t, _P]),
    "cnn_batchnorm2d_partial_sums": (C.c_int, [_P, _P, C.c_float, _P] + [C.c_int] * 4 + [_P, C.c_size_t, _P]),
    "cnn_batchnorm2d_forward_from_sums": (C.c_int, [_P] * 10 + [C.c_float] + [C.c_int] * 4 + [C.c_float, C.c_float, _P]),
    "cnn_batchnorm2d_forward_from_sums_relu": (C.c_int, [_P] * 11 + [C.c_float] + [C.c_int] * 4 + [C.c_float, C.c_float, _P]),
    "cnn_batchnorm2d_backward_sums": (C.c_int, [_P] * 6 + [C.c_int] * 4 + [C.c_float, _P, C.c_size_t, _P]),
    "cnn_batchnorm2d_backward_from_sums": (C.c_int, [_P] * 6 + [C.c_float, _P, _P] + [C.c_int] * 4 + [C.c_float, _P]),
    "cnn_sgd_update": (C.c_int, [_P, _P, C.c_size_t, C.c_float, C.c_float, _P]),
    "cnn_sgd_update_keep": (C.c_int, [_P, _P, C.c_size_t, C.c_float, C.c_float, _P, _P]),
    "cnn_stream_wait_event_local": (C.c_int, [_P, _P]),
    "cnn_stream_create_priority": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "cnn_conv2d_backward_weight_pooled2_sgd_keep": (C.c_int, [_D, _P, _P, _P, _P, _P, _P, C.c_float, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P,
                                                               C.c_size_t, _P]),
    "cnn_comm_available": (C.c_int, []),
    "cnn_comm_version": (C.c_int, []),
    "cnn_comm_unique_id": (C.c_int, [_P]),
    "cnn_comm_init_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, _P]),
    "cnn_comm_init_all": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]),
    "cnn_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cnn_comm_destroy": (C.c_int, [_P]),
    "cnn_comm_group_start": (C.c_int, []),
    "cnn_comm_group_end": (C.c_int, []),
    "cnn_allreduce_grads": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cnn_comm_split": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cnn_comm_broadcast": (C.c_int, [_P, _P, C.c_size_t, C.c_int, _P]),
    "cnn_conv2d_autotune_workspace_bytes": (C.c_size_t, [_D]),
    "cnn_conv2d_autotune_ws": (C.c_int, [_D, _P, C.c_size_t, _P]),
    "cnn_conv2d_tune_export": (C.c_int, [_D, C.POINTER(C.c_int32)]),
    "cnn_conv2d_tune_import": (C.c_int, [_D, C.POINTER(C.c_int32)]),
    "cnn_softmax_xent": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "cnn_linear_forward_softmax_xent": (C.c_int, [_P] * 8 + [C.c_int] * 3 + [_P]),
    "cnn_loss_from_terms": (C.c_int, [_P, _P, C.c_int, _P]),
    "cnn_conv2d_backward_weight_side": (C.c_int, [_D, _P, _P, _P, _P, C.c_float, _P, C.c_size_t, _P]),
    "cnn_device_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "cnn_device_free": (C.c_int, [_P]),
    "cnn_memcpy_h2d": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cnn_memcpy_d2h": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cnn_memcpy_d2d": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "cnn_memset_zero": (C.c_int, [_P, C.c_size_t, _P]),
    "cnn_stream_synchronize": (C.c_int, [_P]),
    "cnn_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "cnn_stream_destroy": (C.c_int, [_P]),
    "cnn_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "cnn_event_destroy": (C.c_int, [_P]),
    "cnn_event_record": (C.c_int, [_P, _P]),
    "cnn_stream_wait_event": (C.c_int, [_P, _P]),
    "cnn_host_alloc_pinned": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "cnn_host_free_pinned": (C.c_int, [_P]),
}


def load():
    """dlopen the in-tree library (built by __graft_entry__.build() / make -C cnn_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CnnAmdError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        try:  # torch bundles its own libamdhip64: import it FIRST so that this process has ONE HIP runtime (loading
            import torch  # noqa: F401  ours first and torch later leaves torch with "No HIP GPUs are available")
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here == the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise CnnAmdError(f"{what} failed with code {rc}: {load().cnn_amd_last_error().decode()}")


def conv_out_dim(n, k, s, pad=0):
    return (n + 2 * pad - k) // s + 1


def pool_out_dim(n, k, step):
    return (n - k) // step + 1


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise CnnAmdError("tensors passed to the HIP path must be contiguous device tensors")


class Conv2d:
    """One Conv2D geometry: owns the scratch buffer, forwards to the C ABI.  Weight layout [Co][Ci][k][k]."""

    def __init__(self, B, Ci, H, W, Co, k=3, s=2, pad=0, device="cuda"):
        import torch

        self.desc = ConvDesc(B, Ci, H, W, Co, k, s, pad)
        self.Ho, self.Wo = conv_out_dim(H, k, s, pad), conv_out_dim(W, k, s, pad)
        self.lib = load()
        self.ws_bytes = int(self.lib.cnn_conv2d_workspace_bytes(C.byref(self.desc)))
        if self.ws_bytes == 0:
            raise CnnAmdError("cnn_conv2d_workspace_bytes: " + self.lib.cnn_amd_last_error().decode())
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=device)
        self._im2col_ws = None
        self._bwd_ws = None

    def out_shape(self):
        d = self.desc
        return (d.B, d.Co, self.Ho, self.Wo)

    def autotune(self):
        """cnn_conv2d_autotune: measure and pin the implicit-GEMM tile of this geometry (synchronises; once per process)"""
        check(self.lib.cnn_conv2d_autotune(C.byref(self.desc), _stream()), "cnn_conv2d_autotune")

    def forward(self, x, w, bias, y=None):
        import torch

        _need_gpu(x, w, bias, y)
        if y is None:
            y = torch.empty(self.out_shape(), dtype=torch.float32, device=x.device)
        check(self.lib.cnn_conv2d_forward(C.byref(self.desc), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(self.ws),
                                          self.ws_bytes, _stream()), "cnn_conv2d_forward")
        return y

    def forward_relu(self, x, w, bias, y, y_relu):
        """Conv2D::forward + the ReLU::forward behind it in one kernel: writes y and y_relu = relu(y)"""
        _need_gpu(x, w, bias, y, y_relu)
        check(self.lib.cnn_conv2d_forward_relu(C.byref(self.desc), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(y_relu),
                                               _ptr(self.ws), self.ws_bytes, _stream()), "cnn_conv2d_forward_relu")
        return y_relu

    def relu_maxpool2_supported(self):
        return bool(self.lib.cnn_conv2d_relu_maxpool2_supported(C.byref(self.desc)))

    def pool_mask_packed_supported(self):
        return bool(self.lib.cnn_conv2d_pool_mask_packed_supported(C.byref(self.desc)))

    def set_pool_mask_packed(self, on=True):
        """desc.flags: the fused block's calls of this object write / read the packed one-byte pool mask (include/cnn_amd.h)"""
        self.desc.flags = POOL_MASK_PACKED if on else 0

    def pool_mask_bytes(self):
        return int(self.lib.cnn_conv2d_pool_mask_bytes(C.byref(self.desc)))

    def pool_mask_unpack(self, packed, mask):
        _need_gpu(packed, mask)
        check(self.lib.cnn_conv2d_pool_mask_unpack(C.byref(self.desc), _ptr(packed), _ptr(mask), _stream()), "cnn_conv2d_pool_mask_unpack")
        return mask

    def relu_maxpool2_forward(self, x, w, bias, pooled, mask=None, prepared_fwd=None):
        """Conv2D -> ReLU -> MaxPool2D(2,2) in one kernel: writes pooled (and mask); from prepared filters when given"""
        _need_gpu(x, pooled)
        if prepared_fwd is not None:
            check(self.lib.cnn_conv2d_relu_maxpool2_forward_prepared(C.byref(self.desc), _ptr(x), _ptr(prepared_fwd), _ptr(pooled),
                                                                     _ptr(mask) if mask is not None else None, _stream()),
                  "cnn_conv2d_relu_maxpool2_forward_prepared")
        else:
            check(self.lib.cnn_conv2d_relu_maxpool2_forward(C.byref(self.desc), _ptr(x), _ptr(w), _ptr(bias), _ptr(pooled),
                                                            _ptr(mask) if mask is not None else None, _ptr(self.ws), self.ws_bytes,
                                                            _stream()), "cnn_conv2d_relu_maxpool2_forward")
        return pooled

    def backward_weight_pooled2(self, x, dpool, mask, pooled, divisor, gw, gb):
        """weight / bias gradient with dy = relu_backward(maxpool_backward(dpool)) rebuilt on the fly"""
        _need_gpu(x, dpool, mask, gw, gb)  # pooled=None: dpool already carries the ReLU mask
        check(self.lib.cnn_conv2d_backward_weight_pooled2(C.byref(self.desc), _ptr(x), _ptr(dpool), _ptr(mask),
                                                          _ptr(pooled) if pooled is not None else None, _ptr(gw),
                                                          _ptr(gb), float(divisor), _ptr(self.ws), self.ws_bytes, _stream()),
              "cnn_conv2d_backward_weight_pooled2")
        return gw, gb

    def backward_weight_pooled2_sgd(self, x, dpool, mask, pooled, divisor, gw, gb, w, bias, lr, grad_scale, prepared_fwd, prepared_dgrad):
        """backward_weight_pooled2 + this layer's SGD step + its re-prepared filters (one extra launch)"""
        _need_gpu(x, dpool, mask, gw, gb, w, bias)
        check(self.lib.cnn_conv2d_backward_weight_pooled2_sgd(C.byref(self.desc), _ptr(x), _ptr(dpool), _ptr(mask),
                                                              _ptr(pooled) if pooled is not None else None, _ptr(gw), _ptr(gb),
                                                              float(divisor), _ptr(w), _ptr(bias), float(lr), float(grad_scale),
                                                              _ptr(prepared_fwd) if prepared_fwd is not None else None,
                                                              _ptr(prepared_dgrad) if prepared_dgrad is not None else None,
                                                              _ptr(self.ws), self.ws_bytes, _stream()),
              "cnn_conv2d_backward_weight_pooled2_sgd")
        return gw, gb

    def backward_pooled2_prepared(self, x, dpool, mask, pooled, prepared_dgrad, divisor, gw, gb, dx, defer_join=False):
        """weight gradient (library side stream) and data gradient (current stream) of the pool-fused block in one call"""
        _need_gpu(x, dpool, mask, prepared_dgrad, gw, gb, dx)
        check(self.lib.cnn_conv2d_backward_pooled2_prepared(C.byref(self.desc), _ptr(x), _ptr(dpool), _ptr(mask),
                                                            _ptr(pooled) if pooled is not None else None, _ptr(prepared_dgrad), _ptr(gw),
                                                            _ptr(gb), _ptr(dx), float(divisor), _ptr(self.ws), self.ws_bytes, _stream(),
                                                            1 if defer_join else 0), "cnn_conv2d_backward_pooled2_prepared")
        return gw, gb, dx

    def backward_data_pooled2(self, dpool, mask, pooled, w, dx, prepared_dgrad=None):
        _need_gpu(dpool, mask, dx)
        pooled_p = _ptr(pooled) if pooled is not None else None
        if prepared_dgrad is not None:
            check(self.lib.cnn_conv2d_backward_data_pooled2_prepared(C.byref(self.desc), _ptr(dpool), _ptr(mask), pooled_p,
                                                                     _ptr(prepared_dgrad), _ptr(dx), _stream()),
                  "cnn_conv2d_backward_data_pooled2_prepared")
        else:
            check(self.lib.cnn_conv2d_backward_data_pooled2(C.byref(self.desc), _ptr(dpool), _ptr(mask), pooled_p, _ptr(w), _ptr(dx),
                                                            _ptr(self.ws), self.ws_bytes, _stream()), "cnn_conv2d_backward_data_pooled2")
        return dx

    def backward_weight(self, x, dy, divisor, gw=None, gb=None, want_bias=True):
        import torch

        _need_gpu(x, dy, gw, gb)
        d = self.desc
        if gw is None:
            gw = torch.empty((d.Co, d.Ci, d.k, d.k), dtype=torch.float32, device=x.device)
        if gb is None and want_bias:
            gb = torch.empty((d.Co,), dtype=torch.float32, device=x.device)
        check(self.lib.cnn_conv2d_backward_weight(C.byref(d), _ptr(x), _ptr(dy), _ptr(gw), _ptr(gb), float(divisor),
                                                  _ptr(self.ws), self.ws_bytes, _stream()), "cnn_conv2d_backward_weight")
        return gw, gb

    def backward_data(self, dy, w, dx=None):
        import torch

        _need_gpu(dy, w, dx)
        d = self.desc
        if dx is None:
            dx = torch.empty((d.B, d.Ci, d.H, d.W), dtype=torch.float32, device=dy.device)
        check(self.lib.cnn_conv2d_backward_data(C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), _ptr(self.ws), self.ws_bytes,
                                                _stream()), "cnn_conv2d_backward_data")
        return dx

    def backward(self, x, dy, w, divisor, gw=None, gb=None, dx=None, defer_join=False):
        """both gradients in one call (weight-gradient kernels on the library's side stream, concurrently with dgrad)"""
        import torch

        _need_gpu(x, dy, w, gw, gb, dx)
        d = self.desc
        if self._bwd_ws is None:
            n = int(self.lib.cnn_conv2d_backward_workspace_bytes(C.byref(d)))
            self._bwd_ws = torch.empty(n, dtype=torch.uint8, device=x.device)
        if gw is None:
            gw = torch.empty((d.Co, d.Ci, d.k, d.k), dtype=torch.float32, device=x.device)
        if gb is None:
            gb = torch.empty((d.Co,), dtype=torch.float32, device=x.device)
        if dx is None:
            dx = torch.empty((d.B, d.Ci, d.H, d.W), dtype=torch.float32, device=x.device)
        check(self.lib.cnn_conv2d_backward(C.byref(d), _ptr(x), _ptr(dy), _ptr(w), _ptr(gw), _ptr(gb), _ptr(dx), float(divisor),
                                           _ptr(self._bwd_ws), self._bwd_ws.numel(), _stream(), 1 if defer_join else 0), "cnn_conv2d_backward")
        return gw, gb, dx

    # ---- prepared-filter path (cnn_conv2d_prepare_filters) ----
    def prepared_buffers(self, device="cuda"):
        """(fwd, dgrad) device buffers for prepare_filters()"""
        import torch

        n = int(self.lib.cnn_conv2d_prepared_bytes(C.byref(self.desc)))
        return torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n, dtype=torch.uint8, device=device)

    def relu_only_supported(self):
        return bool(self.lib.cnn_conv2d_relu_only_supported(C.byref(self.desc)))

    def forward_prepared(self, x, prepared_fwd, bias, y, y_relu=None):
        """y may be None (with y_relu) on layers where relu_only_supported(): the pre-activation tensor is then not written"""
        _need_gpu(x, prepared_fwd, bias)
        check(self.lib.cnn_conv2d_forward_prepared(C.byref(self.desc), _ptr(x), _ptr(prepared_fwd), _ptr(bias),
                                                   _ptr(y) if y is not None else None,
                                                   _ptr(y_relu) if y_relu is not None else None, _stream()),
              "cnn_conv2d_forward_prepared")
        return y

    def backward_data_prepared(self, dy, prepared_dgrad, dx):
        _need_gpu(dy, prepared_dgrad, dx)
        check(self.lib.cnn_conv2d_backward_data_prepared(C.byref(self.desc), _ptr(dy), _ptr(prepared_dgrad), _ptr(dx), _stream()),
              "cnn_conv2d_backward_data_prepared")
        return dx

    def backward_prepared(self, x, dy, prepared_dgrad, divisor, gw, gb, dx, defer_join=False, relu_below=None):
        """relu_below: output of the ReLU layer in front of this convolution (= x when that layer feeds it directly); its
        backward pass is applied to dx inside the data-gradient kernel"""
        _need_gpu(x, dy, prepared_dgrad, gw, gb, dx)
        check(self.lib.cnn_conv2d_backward_prepared_relu(C.byref(self.desc), _ptr(x), _ptr(dy), _ptr(prepared_dgrad),
                                                         _ptr(relu_below) if relu_below is not None else None, _ptr(gw), _ptr(gb),
                                                         _ptr(dx), float(divisor), _ptr(self.ws), self.ws_bytes, _stream(),
                                                         1 if defer_join else 0), "cnn_conv2d_backward_prepared")
        return gw, gb, dx

    def backward_data_relu(self, dy, w, relu_below, dx, prepared_dgrad=None):
        _need_gpu(dy, relu_below, dx)
        if prepared_dgrad is not None:
            check(self.lib.cnn_conv2d_backward_data_relu_prepared(C.byref(self.desc), _ptr(dy), _ptr(prepared_dgrad), _ptr(relu_below),
                                                                  _ptr(dx), _stream()), "cnn_conv2d_backward_data_relu_prepared")
        else:
            check(self.lib.cnn_conv2d_backward_data_relu(C.byref(self.desc), _ptr(dy), _ptr(w), _ptr(relu_below), _ptr(dx),
                                                         _ptr(self.ws), self.ws_bytes, _stream()), "cnn_conv2d_backward_data_relu")
        return dx

    # ---- im2col functional fallback (parity cross-check only) ----
    def _iws(self, device):
        import torch

        if self._im2col_ws is None:
            n = int(self.lib.cnn_conv2d_im2col_workspace_bytes(C.byref(self.desc)))
            self._im2col_ws = torch.empty(n, dtype=torch.uint8, device=device)
        return self._im2col_ws

    def forward_im2col(self, x, w, bias):
        import torch

        _need_gpu(x, w, bias)
        y = torch.empty(self.out_shape(), dtype=torch.float32, device=x.device)
        ws = self._iws(x.device)
        check(self.lib.cnn_conv2d_forward_im2col(C.byref(self.desc), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(ws),
                                                 ws.numel(), _stream()), "cnn_conv2d_forward_im2col")
        return y

    def backward_weight_im2col(self, x, dy, divisor):
        import torch

        _need_gpu(x, dy)
        d = self.desc
        gw = torch.empty((d.Co, d.Ci, d.k, d.k), dtype=torch.float32, device=x.device)
        gb = torch.empty((d.Co,), dtype=torch.float32, device=x.device)
        ws = self._iws(x.device)
        check(self.lib.cnn_conv2d_backward_weight_im2col(C.byref(d), _ptr(x), _ptr(dy), _ptr(gw), _ptr(gb), float(divisor),
                                                         _ptr(ws), ws.numel(), _stream()), "cnn_conv2d_backward_weight_im2col")
        return gw, gb

    def backward_data_im2col(self, dy, w):
        import torch

        _need_gpu(dy, w)
        d = self.desc
        dx = torch.empty((d.B, d.Ci, d.H, d.W), dtype=torch.float32, device=dy.device)
        ws = self._iws(dy.device)
        check(self.lib.cnn_conv2d_backward_data_im2col(C.byref(d), _ptr(dy), _ptr(w), _ptr(dx), _ptr(ws), ws.numel(),
                                                       _stream()), "cnn_conv2d_backward_data_im2col")
        return dx


def maxpool_forward(x, k, step, record_mask=True):
    import torch

    _need_gpu(x)
    B, Cc, H, W = x.shape
    Ho, Wo = pool_out_dim(H, k, step), pool_out_dim(W, k, step)
    y = torch.empty((B, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    mask = torch.empty((B, Cc, Ho, Wo), dtype=torch.int32, device=x.device) if record_mask else None
    check(load().cnn_maxpool2d_forward(_ptr(x), _ptr(y), _ptr(mask), B, Cc, H, W, k, step, _stream()), "cnn_maxpool2d_forward")
    return y, mask


def maxpool_backward(dy, mask, in_shape, k, step, dx=None):
    import torch

    _need_gpu(dy, mask, dx)
    B, Cc, H, W = in_shape
    if dx is None:
        dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    check(load().cnn_maxpool2d_backward(_ptr(dy), _ptr(mask), _ptr(dx), B, Cc, H, W, k, step, _stream()), "cnn_maxpool2d_backward")
    return dx


def maxpool_backward_relu(dy, mask, pooled, in_shape, k, step, dx=None):
    """MaxPool2D::backward + the ReLU::backward of the layer in front of the pool in one kernel (pooled = pool forward output)"""
    import torch

    _need_gpu(dy, mask, pooled, dx)
    B, Cc, H, W = in_shape
    if dx is None:
        dx = torch.empty(in_shape, dtype=torch.float32, device=dy.device)
    check(load().cnn_maxpool2d_backward_relu(_ptr(dy), _ptr(mask), _ptr(pooled), _ptr(dx), B, Cc, H, W, k, step, _stream()),
          "cnn_maxpool2d_backward_relu")
    return dx


def relu_forward(x, y=None):
    import torch

    _need_gpu(x, y)
    if y is None:
        y = torch.empty_like(x)
    check(load().cnn_relu_forward(_ptr(x), _ptr(y), x.numel(), _stream()), "cnn_relu_forward")
    return y


def relu_backward(y, dy):
    """in place on dy (relu.cpp:37-39); returns dy"""
    _need_gpu(y, dy)
    check(load().cnn_relu_backward(_ptr(y), _ptr(dy), y.numel(), _stream()), "cnn_relu_backward")
    return dy


def linear_forward(x, w, bias, y=None):
    import torch

    _need_gpu(x, w, bias, y)
    B = x.shape[0]
    n_in, n_out = w.shape
    if y is None:
        y = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
    check(load().cnn_linear_forward(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, n_in, n_out, _stream()), "cnn_linear_forward")
    return y


def linear_backward(x, dy, w, divisor, gw=None, gb=None, dx=None, relu_below=False):
    """relu_below: x is a ReLU layer's output; its backward pass is applied to dx in the same kernel"""
    import torch

    _need_gpu(x, dy, w, gw, gb, dx)
    B = x.shape[0]
    n_in, n_out = w.shape
    if gw is None:
        gw = torch.empty_like(w)
    if gb is None:
        gb = torch.empty((n_out,), dtype=torch.float32, device=x.device)
    if dx is None:
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    fn = load().cnn_linear_backward_relu if relu_below else load().cnn_linear_backward
    check(fn(_ptr(x), _ptr(dy), _ptr(w), _ptr(gw), _ptr(gb), _ptr(dx), B, n_in, n_out, float(divisor), _stream()),
          "cnn_linear_backward")
    return gw, gb, dx


class BatchNorm2d:
    """BatchNorm2D on the HIP path (batchnorm2d.cpp:6-182): owns the workspace and the saved batch statistics; gamma,
    beta and the moving statistics are caller tensors of [C] (checkpoint order gamma, beta, moving_mean, moving_var,
    batchnorm2d.cpp:168-173)."""

    def __init__(self, batch, channels, H, W, eps=1e-5, momentum=0.1, device="cuda"):
        import torch

        self.B, self.C, self.H, self.W, self.eps, self.momentum = batch, channels, H, W, eps, momentum
        self.ws_bytes = load().cnn_batchnorm2d_workspace_bytes(batch, channels, H, W)
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=device)
        self.saved_mean = torch.zeros(channels, dtype=torch.float32, device=device)
        self.saved_var = torch.zeros(channels, dtype=torch.float32, device=device)

    def forward(self, x, gamma, beta, moving_mean, moving_var, y, training=True, y_relu=None):
        """y_relu: also write relu(y) (the ReLU layer behind this one) from the same pass"""
        _need_gpu(x, y, gamma, beta, moving_mean, moving_var)
        if y_relu is not None:
            _need_gpu(y_relu)
            check(load().cnn_batchnorm2d_forward_relu(_ptr(x), _ptr(y), _ptr(y_relu), _ptr(gamma), _ptr(beta), _ptr(moving_mean),
                                                      _ptr(moving_var), _ptr(self.saved_mean), _ptr(self.saved_var), self.B, self.C,
                                                      self.H, self.W, self.eps, self.momentum, 1 if training else 0, _ptr(self.ws),
                                                      self.ws_bytes, _stream()), "cnn_batchnorm2d_forward_relu")
            return y
        check(load().cnn_batchnorm2d_forward(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var),
                                             _ptr(self.saved_mean), _ptr(self.saved_var), self.B, self.C, self.H, self.W,
                                             self.eps, self.momentum, 1 if training else 0, _ptr(self.ws), self.ws_bytes,
                                             _stream()), "cnn_batchnorm2d_forward")
        return y

    # ---- sync-BN: the batch is sharded over `world` data-parallel ranks; `allreduce(t)` sums a small tensor in place ----
    def forward_sync(self, x, gamma, beta, moving_mean, moving_var, y, allreduce, global_count, y_relu=None):
        """training forward over the GLOBAL batch (batchnorm2d.cpp:46-80): two tiny all-reduces of [C] sums
        (y_relu: also write relu(y), the output of the ReLU layer behind this one)"""
        import torch

        _need_gpu(x, y, gamma, beta, moving_mean, moving_var)
        L = load()
        s1 = torch.empty(self.C, dtype=torch.float32, device=x.device)
        s2 = torch.empty_like(s1)
        dims = (self.B, self.C, self.H, self.W)
        check(L.cnn_batchnorm2d_partial_sums(_ptr(x), None, 0.0, _ptr(s1), *dims, _ptr(self.ws), self.ws_bytes, _stream()), "bn sums 1")
        allreduce(s1)
        check(L.cnn_batchnorm2d_partial_sums(_ptr(x), _ptr(s1), float(global_count), _ptr(s2), *dims, _ptr(self.ws), self.ws_bytes,
                                             _stream()), "bn sums 2")
        allreduce(s2)
        if y_relu is not None:
            _need_gpu(y_relu)
            check(L.cnn_batchnorm2d_forward_from_sums_relu(_ptr(x), _ptr(y), _ptr(y_relu), _ptr(gamma), _ptr(beta), _ptr(moving_mean),
                                                           _ptr(moving_var), _ptr(self.saved_mean), _ptr(self.saved_var), _ptr(s1), _ptr(s2),
                                                           float(global_count), *dims, self.eps, self.momentum, _stream()),
                  "cnn_batchnorm2d_forward_from_sums_relu")
            return y
        check(L.cnn_batchnorm2d_forward_from_sums(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(moving_mean), _ptr(moving_var),
                                                  _ptr(self.saved_mean), _ptr(self.saved_var), _ptr(s1), _ptr(s2), float(global_count),
                                                  *dims, self.eps, self.momentum, _stream()), "cnn_batchnorm2d_forward_from_sums")
        return y

    def backward_sync(self, x, dy, gamma, ggamma, gbeta, allreduce, global_count):
        """dy -> dx in place; ggamma / gbeta come out as the full-batch sums on every rank (one [C][4] all-reduce)"""
        import torch

        _need_gpu(x, dy, gamma, ggamma, gbeta)
        L = load()
        s4 = torch.empty(self.C * 4, dtype=torch.float32, device=x.device)
        dims = (self.B, self.C, self.H, self.W)
        check(L.cnn_batchnorm2d_backward_sums(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(self.saved_mean), _ptr(self.saved_var), _ptr(s4),
                                              *dims, self.eps, _ptr(self.ws), self.ws_bytes, _stream()), "cnn_batchnorm2d_backward_sums")
        allreduce(s4)
        check(L.cnn_batchnorm2d_backward_from_sums(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(self.saved_mean), _ptr(self.saved_var), _ptr(s4),
                                                   float(global_count), _ptr(ggamma), _ptr(gbeta), *dims, self.eps, _stream()),
              "cnn_batchnorm2d_backward_from_sums")
        return dy

    def backward(self, x, dy, gamma, ggamma, gbeta):
        """dy -> dx in place (batchnorm2d.cpp:149-155)"""
        _need_gpu(x, dy, gamma, ggamma, gbeta)
        check(load().cnn_batchnorm2d_backward(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(self.saved_mean), _ptr(self.saved_var),
                                              _ptr(ggamma), _ptr(gbeta), self.B, self.C, self.H, self.W, self.eps,
                                              _ptr(self.ws), self.ws_bytes, _stream()), "cnn_batchnorm2d_backward")
        return dy

    def forward_relu_pool(self, x, gamma, beta, moving_mean, moving_var, pooled, mask, y=None, y_relu=None, training=True):
        """BatchNorm2D -> ReLU -> MaxPool2D(2, 2) in one apply pass: pooled / mask written, y / y_relu only when given"""
        check(load().cnn_batchnorm2d_forward_relu_pool(_ptr(x), _ptr(y), _ptr(y_relu), _ptr(pooled), _ptr(mask), _ptr(gamma), _ptr(beta),
                                                       _ptr(moving_mean), _ptr(moving_var), _ptr(self.saved_mean), _ptr(self.saved_var), self.B,
                                                       self.C, self.H, self.W, self.eps, self.momentum, 1 if training else 0, _ptr(self.ws),
                                                       self.ws_bytes, _stream()), "cnn_batchnorm2d_forward_relu_pool")
        return pooled

    def backward_pooled_supported(self):
        return bool(load().cnn_batchnorm2d_backward_pooled_supported(self.B, self.C, self.H, self.W))

    def backward_pooled(self, x, dpool, mask, pooled, gamma, ggamma, gbeta, dx):
        """BatchNorm2D <- ReLU <- MaxPool2D(2, 2) backward from the pooled domain (dpool, the pool's int32 mask and output): dx written,
        ggamma / gbeta filled; bit-identical to maxpool2d_backward_relu + backward"""
        check(load().cnn_batchnorm2d_backward_pooled(_ptr(x), _ptr(dpool), _ptr(mask), _ptr(pooled), _ptr(dx), _ptr(gamma), _ptr(self.saved_mean),
                                                     _ptr(self.saved_var), _ptr(ggamma), _ptr(gbeta), self.B, self.C, self.H, self.W, self.eps,
                                                     _ptr(self.ws), self.ws_bytes, _stream()), "cnn_batchnorm2d_backward_pooled")
        return dx


def prepare_filters(convs, weights, biases, fwd_bufs, dgrad_bufs):
    """re-arrange the filters of up to 6 layers for their forward / data-gradient kernels with one or two launches"""
    n = len(convs)
    _need_gpu(*weights, *biases)
    descs = (ConvDesc * n)(*[c.desc for c in convs])
    arr = lambda ts: (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts])
    check(load().cnn_conv2d_prepare_filters(n, descs, arr(weights), arr(biases), arr(fwd_bufs) if fwd_bufs else None,
                                            arr(dgrad_bufs) if dgrad_bufs else None, _stream()), "cnn_conv2d_prepare_filters")


def sgd_update(params, grads, lr, grad_scale=1.0):
    _need_gpu(params, grads)
    check(load().cnn_sgd_update(_ptr(params), _ptr(grads), params.numel(), float(lr), float(grad_scale), _stream()), "cnn_sgd_update")
    return params


def softmax_xent(logits, labels, want_probs=True):
    import torch

    _need_gpu(logits, labels)
    B, n = logits.shape
    probs = torch.empty_like(logits) if want_probs else None
    delta = torch.empty_like(logits)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    check(load().cnn_softmax_xent(_ptr(logits), _ptr(labels), _ptr(probs), _ptr(delta), _ptr(loss), B, n, _stream()), "cnn_softmax_xent")
    return probs, delta, loss


def set_option(name, value):
    """one of the measurement switches of DESIGN.md section 10 (name with or without the CNN_AMD_ prefix); value None removes it.
    The library reads the CNN_AMD_* environment once, at first use: later changes go through here."""
    check(load().cnn_amd_set_option(name.encode(), None if value is None else str(value).encode()), "cnn_amd_set_option")


def get_option(name):
    buf = C.create_string_buffer(256)
    return buf.value.decode() if load().cnn_amd_get_option(name.encode(), buf, 256) == 0 else None


class option:
    """with capi.option("RD_SLOW", 1): ...   -- sets the switch, restores the previous state on exit"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def kernel_timing(mode, filter_key="", every=1):
    """0 = off, 1 = every kernel, 2 = only keys containing filter_key (and of those only every `every`-th launch)"""
    check(load().cnn_amd_kernel_timing_sampling(int(every)), "cnn_amd_kernel_timing_sampling")
    check(load().cnn_amd_kernel_timing_enable(int(mode), filter_key.encode()), "cnn_amd_kernel_timing_enable")


def kernel_timing_report():
    """-> {key: (launches, total_ms)} in first-launch order; clears the records"""
    lib = load()
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        need = lib.cnn_amd_kernel_timing_report(buf, cap)
        if need < 0:
            raise CnnAmdError("cnn_amd_kernel_timing_report failed")
        if need <= cap:
            break
        cap = int(need) + 64
    out = {}
    for line in buf.value.decode().splitlines():
        key, cnt, ms = line.rsplit("\t", 2)
        out[key] = (int(cnt), float(ms))
    return out


def grad_cam(feature):
    """AlexNet::grad_cam's arithmetic on a [B][C][H][W] feature map: (normalised cam [B][H][W], uint8 image [H][W] of plane 0)"""
    import torch

    _need_gpu(feature)
    B, Cc, H, W = feature.shape
    cam = torch.empty((B, H, W), dtype=torch.float32, device=feature.device)
    img = torch.empty((H, W), dtype=torch.uint8, device=feature.device)
    check(load().cnn_grad_cam(_ptr(feature), B, Cc, H, W, _ptr(cam), _ptr(img), _stream()), "cnn_grad_cam")
    return cam, img


def side_stream():
    """the library's side stream of this thread / device as a torch stream (work queued there runs behind the deferred weight gradients)"""
    import torch

    out = C.c_void_p()
    check(load().cnn_amd_side_stream_get(C.byref(out)), "cnn_amd_side_stream_get")
    return torch.cuda.ExternalStream(out.value)


def flush_reduces():
    """launch the recorded weight-gradient slab reductions on the CURRENT stream now (see cnn_amd_flush_reduces)"""
    check(load().cnn_amd_flush_reduces(_stream()), "cnn_amd_flush_reduces")


def publish_next_kernel():
    """the next library kernel on the current stream publishes its completion (see cnn_amd_publish_next_kernel)"""
    check(load().cnn_amd_publish_next_kernel(_stream()), "cnn_amd_publish_next_kernel")


def wait_published(stream=None):
    """`stream` (a torch stream; default: the current one) waits for the published kernel"""
    check(load().cnn_amd_wait_published(C.c_void_p(stream.cuda_stream) if stream is not None else _stream()), "cnn_amd_wait_published")


def side_stream_join():
    """order the library's side stream (deferred weight gradients) before the current stream"""
    check(load().cnn_amd_side_stream_join(_stream()), "cnn_amd_side_stream_join")


class BatchStager:
    """cnn_batch_stager_*: `depth` pinned host slots + device buffers, uploads on a copy stream of its own, ordered by events"""

    def __init__(self, batch_bytes=None, depth=2, u8_shape=None):
        """u8_shape=(B, H, W): the slots hold interleaved BYTES [B][H][W][3] (cv::Mat CV_8UC3), submit() returns the fp32 planar batch
        the stager's conversion kernel writes (cnn_batch_stager_create_u8)"""
        self.lib = load()
        self.h = C.c_void_p()
        self.u8 = u8_shape is not None
        if self.u8:
            B, H, W = (int(v) for v in u8_shape)
            self.bytes = B * H * W * 3
            check(self.lib.cnn_batch_stager_create_u8(C.byref(self.h), B, H, W, depth), "cnn_batch_stager_create_u8")
        else:
            self.bytes = int(batch_bytes)
            check(self.lib.cnn_batch_stager_create(C.byref(self.h), self.bytes, depth), "cnn_batch_stager_create")

    def acquire(self):
        """-> (numpy view of the pinned slot to fill -- float32, or uint8 for a u8 stager --, slot index); blocks until the slot is free"""
        import numpy as np

        host, slot = C.c_void_p(), C.c_int()
        check(self.lib.cnn_batch_stager_acquire(self.h, C.byref(host), C.byref(slot)), "cnn_batch_stager_acquire")
        if self.u8:
            return np.ctypeslib.as_array((C.c_ubyte * self.bytes).from_address(host.value)), slot.value
        buf = (C.c_float * (self.bytes // 4)).from_address(host.value)
        return np.ctypeslib.as_array(buf), slot.value

    def submit(self, slot):
        """enqueue the upload; -> device pointer (int) of the slot's device buffer"""
        dev = C.c_void_p()
        check(self.lib.cnn_batch_stager_submit(self.h, slot, C.byref(dev)), "cnn_batch_stager_submit")
        return dev.value

    def wait(self, slot):
        check(self.lib.cnn_batch_stager_wait(self.h, slot, _stream()), "cnn_batch_stager_wait")

    def release(self, slot):
        check(self.lib.cnn_batch_stager_release(self.h, slot, _stream()), "cnn_batch_stager_release")

    def close(self):
        if self.h:
            self.lib.cnn_batch_stager_destroy(self.h)
            self.h = None


def _dropout_counts(p, channels):
    """selected_num = int(p * C) and keep = 1 - p in the reference's FLOAT arithmetic (dropout.cpp:16: data_type p times int):
    Python doubles give a different count for some p (0.29 * 100 -> 28 here, 29 in the reference)"""
    import numpy as np

    return int(np.float32(p) * np.float32(channels)), float(np.float32(1) - np.float32(p))


def dropout_forward(x, p, training=True, y=None):
    """Dropout::forward (dropout.cpp:7-55): channels 0 .. int(p*C)-1 zeroed in training, x * (1-p) otherwise"""
    import torch

    _need_gpu(x, y)
    B, Cc, H, W = x.shape
    if y is None:
        y = torch.empty_like(x)
    dropped, keep = _dropout_counts(p, Cc)
    check(load().cnn_dropout_forward(_ptr(x), _ptr(y), B, Cc, H, W, dropped, 1 if training else 0, keep, _stream()), "cnn_dropout_forward")
    return y


def dropout_backward(dy, p):
    """in place on dy (dropout.cpp:57-69)"""
    _need_gpu(dy)
    B, Cc, H, W = dy.shape
    check(load().cnn_dropout_backward(_ptr(dy), B, Cc, H, W, _dropout_counts(p, Cc)[0], _stream()), "cnn_dropout_backward")
    return dy

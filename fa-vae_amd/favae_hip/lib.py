"""ctypes binding of libfavae_hip.so (the C ABI declared in include/favae_hip.h).

PyTorch-ROCm tensors are used only as device buffers: every wrapper forwards `tensor.data_ptr()` and the
current HIP stream.  There is no CPU fallback: if the library is missing the import fails loudly.
"""
import ctypes
import os

import torch

F32, BF16 = 0, 1
RAW_BF16, DACT_BF16, OUT_BF16 = 0x10, 0x20, 0x40      # storage flags of include/favae_hip.h
NOT_SPD = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfavae_hip.so")


class FavaeHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise FavaeHipError(
            "libfavae_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C fa-vae_amd/csrc`; there is no CPU fallback." % _LIB_PATH)
    return ctypes.CDLL(_LIB_PATH)


_lib = _load()
_lib.favae_last_error.restype = ctypes.c_char_p
_lib.favae_version.restype = ctypes.c_int


def raw():
    return _lib


def last_error():
    return _lib.favae_last_error().decode()


def check(rc, allow_flags=False):
    if rc < 0 or (rc > 0 and not allow_flags):
        raise FavaeHipError("libfavae_hip rc=%d: %s" % (rc, last_error()))
    return rc


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return ctypes.c_void_p(t.data_ptr())


def dtype_code(t):
    return F32 if t.dtype == torch.float32 else BF16


_i = ctypes.c_int


def conv_igemm(x, w, bias, out, ks, stride, pad, scatter, splitk=0):
    """x [Nb,Hi,Wi,Cin] (f32|bf16), w [Cout, ks*ks, Cin] same dtype, bias [Cout] f32|None, out [Nb,Ho,Wo,Cout] f32."""
    Nb, Hi, Wi, Cin = x.shape
    _, Ho, Wo, Cout = out.shape
    assert w.dtype == x.dtype and out.dtype == torch.float32 and w.shape[0] >= Cout
    check(_lib.favae_conv_igemm(ptr(x), ptr(w), ptr(bias), ptr(out), _i(Nb), _i(Hi), _i(Wi), _i(Cin), _i(Ho), _i(Wo),
                                _i(Cout), _i(ks), _i(stride), _i(pad), _i(scatter), _i(splitk), _i(dtype_code(x)),
                                stream_ptr()))
    return out


def conv_igemm_red(x, w, bias, out, ks, stride, pad, scatter, bins, mode, act=None, raw=None, mean=None, invstd=None, splitk=0):
    """conv_igemm with per-channel epilogue sums into bins [nbins, 2, Cout] f64 (zeroed by the caller); returns True if the
    launch took them (favae_conv_igemm_red; mode 1 = BN statistics, mode 2 = the two BN-backward reductions)."""
    Nb, Hi, Wi, Cin = x.shape
    _, Ho, Wo, Cout = out.shape
    assert w.dtype == x.dtype and out.dtype == torch.float32 and bins.dtype == torch.float64 and bins.shape[1:] == (2, Cout)
    fused = ctypes.c_int(0)
    check(_lib.favae_conv_igemm_red(ptr(x), ptr(w), ptr(bias), ptr(out), _i(Nb), _i(Hi), _i(Wi), _i(Cin), _i(Ho), _i(Wo),
                                    _i(Cout), _i(ks), _i(stride), _i(pad), _i(scatter), _i(splitk), _i(dtype_code(x)),
                                    ptr(bins), _i(bins.shape[0]), _i(mode), ptr(act), ptr(raw), ptr(mean), ptr(invstd),
                                    ctypes.byref(fused), stream_ptr()))
    return bool(fused.value)


def conv_wgrad(rows, rows_large, cols, cols_large, dw, Nb, Hs, Ws, Hl, Wl, ks, stride, pad, splitk=0):
    """dw [Rc, ks*ks, Cc] f32 += sum_pix rows[pix, r] * cols[pix', c]."""
    Rc, Cc = rows.shape[-1], cols.shape[-1]
    assert rows.dtype == cols.dtype and dw.dtype == torch.float32
    check(_lib.favae_conv_wgrad(ptr(rows), _i(Rc), _i(rows_large), ptr(cols), _i(Cc), _i(cols_large), ptr(dw), _i(Nb),
                                _i(Hs), _i(Ws), _i(Hl), _i(Wl), _i(ks), _i(stride), _i(pad), _i(splitk),
                                _i(dtype_code(rows)), stream_ptr()))
    return dw


def conv_wgrad_folded(rows, cols, dw, Nb, Hs, Ws, Hl, Wl, stride, splitk=0):
    """favae_conv_wgrad_folded: 3x3 / pad 1 weight gradient with <= 3 channels on one side as one GEMM over folded taps."""
    Rc, Cc = rows.shape[-1], cols.shape[-1]
    assert rows.dtype == cols.dtype and dw.dtype == torch.float32
    _lib.favae_conv_wgrad_folded_scratch.restype = ctypes.c_size_t
    nbytes = _lib.favae_conv_wgrad_folded_scratch(_i(Nb), _i(Hl), _i(Wl), _i(dtype_code(rows)))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=rows.device)
    check(_lib.favae_conv_wgrad_folded(ptr(rows), _i(Rc), ptr(cols), _i(Cc), ptr(dw), _i(Nb), _i(Hs), _i(Ws), _i(Hl), _i(Wl),
                                       _i(stride), _i(splitk), _i(dtype_code(rows)), ptr(scratch), ctypes.c_size_t(nbytes),
                                       stream_ptr()))
    torch.cuda.current_stream().synchronize()     # scratch is freed on return
    return dw


def conv_fold_taps(src, Nb, Hs, Ws, Hl, Wl, stride, shifted=0):
    """favae_conv_fold_taps: [Nb*Hs*Ws][32] matrix of the 9 taps x C (<= 3) channels of src."""
    C = src.shape[-1]
    dst = torch.empty(Nb * Hs * Ws, 32, dtype=src.dtype, device=src.device)
    check(_lib.favae_conv_fold_taps(ptr(src), ptr(dst), ctypes.c_size_t(dst.numel() * dst.element_size()), _i(Nb), _i(Hs), _i(Ws),
                                    _i(Hl), _i(Wl), _i(C), _i(stride), _i(shifted), _i(dtype_code(src)), stream_ptr()))
    return dst


def conv_folded_fwd(folded, w, bias, out, bins=None):
    """favae_conv_folded_fwd: out [P, 64] f32 = folded [P, 32] x w [64, K]^T + bias (bf16 operands); optional BatchNorm sums
    into bins [nbins, 2, 64] f64 (zeroed by the caller)."""
    P, Cout, K = folded.shape[0], w.shape[0], w.numel() // w.shape[0]
    assert folded.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.dtype == torch.float32 and out.shape == (P, Cout)
    check(_lib.favae_conv_folded_fwd(ptr(folded), ptr(w), ptr(bias), ptr(out), ctypes.c_longlong(P), _i(K), _i(Cout), ptr(bins),
                                     _i(0 if bins is None else bins.shape[0]), _i(dtype_code(folded)), stream_ptr()))
    return out


def conv_fold_taps_split(src_f32, Nb, Hs, Ws, Hl, Wl, stride):
    """favae_conv_fold_taps_split: fp32 src -> [2, Nb*Hs*Ws, 32] bf16 (plane 0 = bf16(x), plane 1 = bf16(x - bf16(x)))."""
    C = src_f32.shape[-1]
    assert src_f32.dtype == torch.float32
    dst = torch.empty(2, Nb * Hs * Ws, 32, dtype=torch.bfloat16, device=src_f32.device)
    check(_lib.favae_conv_fold_taps_split(ptr(src_f32), ptr(dst), ctypes.c_size_t(dst.numel() * 2), _i(Nb), _i(Hs), _i(Ws), _i(Hl), _i(Wl),
                                          _i(C), _i(stride), stream_ptr()))
    return dst


def conv_folded_fwd_split(folded_hl, w_f32, bias, out, bins=None):
    """favae_conv_folded_fwd_split: out [P, 64] f32 = x w^T + bias with x as hi + lo bf16 planes and w (fp32) split in the kernel."""
    P, Cout, K = folded_hl.shape[1], w_f32.shape[0], w_f32.numel() // w_f32.shape[0]
    assert folded_hl.dtype == torch.bfloat16 and w_f32.dtype == torch.float32 and out.dtype == torch.float32 and out.shape == (P, Cout)
    check(_lib.favae_conv_folded_fwd_split(ptr(folded_hl), ptr(w_f32), ptr(bias), ptr(out), ctypes.c_longlong(P), _i(K), _i(Cout), ptr(bins),
                                           _i(0 if bins is None else bins.shape[0]), stream_ptr()))
    return out


def conv_folded_dgrad(folded, w, out, bins=None, act=None, raw=None, mean=None, invstd=None, groups=1, ab=None):
    """favae_conv_folded_dgrad: out [P, 64] (f32 or bf16) = folded [P, 32] x w [K, 64] (bf16); optional BatchNorm-backward sums of the
    layer in front into bins [groups, nbins, 2, 64] f64 (zeroed by the caller) given act (bf16), raw (f32 / bf16), mean / invstd
    [groups, 64]."""
    P, K = folded.shape[0], w.numel() // 64
    assert folded.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and out.shape == (P, 64)
    assert out.dtype in (torch.float32, torch.bfloat16) and out.is_contiguous()
    flags = BF16 | (OUT_BF16 if out.dtype == torch.bfloat16 else 0) | (RAW_BF16 if raw is not None and raw.dtype == torch.bfloat16 else 0)
    check(_lib.favae_conv_folded_dgrad(ptr(folded), ptr(w), ptr(out), ctypes.c_longlong(P), _i(K), _i(64), ptr(bins),
                                       _i(0 if bins is None else bins.shape[-3]), ptr(act), ptr(raw), ptr(mean), ptr(invstd), ptr(ab), _i(groups),
                                       _i(flags), stream_ptr()))
    return out


def conv_wgrad_prefolded(rows, cols, dw, Rst, Cst, splitk=0):
    """favae_conv_wgrad_prefolded: dw [Rst][Cst] += rows^T cols over the K rows of two plain matrices."""
    assert rows.dtype == cols.dtype and dw.dtype == torch.float32 and rows.shape[0] == cols.shape[0]
    check(_lib.favae_conv_wgrad_prefolded(ptr(rows), _i(rows.shape[1]), _i(Rst), ptr(cols), _i(cols.shape[1]), _i(Cst), ptr(dw),
                                          ctypes.c_longlong(rows.shape[0]), _i(splitk), _i(dtype_code(rows)), stream_ptr()))
    return dw


def conv_igemm_bn(x, in_ab, w, bias, out, ks, stride, pad, scatter, bins=None, gamma=None, beta=None, publish=False, eps=1e-5,
                  momentum=0.1, run_mean=None, run_var=None, splitk=0):
    """favae_conv_igemm_bn.  in_ab given: x is the fp32 output of the preceding conv before its BatchNorm + LeakyReLU.
    publish: the launch turns its epilogue sums (bins) into (mean, invstd, ab) of the BatchNorm that follows this conv.
    Returns (fused, mean, invstd, ab)."""
    Nb, Hi, Wi, Cin = x.shape
    _, Ho, Wo, Cout = out.shape
    mean = invstd = ab = None
    if publish:
        mean = torch.empty(Cout, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        ab = torch.empty(2 * Cout, dtype=torch.float32, device=x.device)
    fused = ctypes.c_int(0)
    _f = ctypes.c_float
    check(_lib.favae_conv_igemm_bn(ptr(x), ptr(in_ab), ptr(w), ptr(bias), ptr(out), _i(Nb), _i(Hi), _i(Wi), _i(Cin), _i(Ho), _i(Wo),
                                   _i(Cout), _i(ks), _i(stride), _i(pad), _i(scatter), _i(splitk), _i(dtype_code(w)), ptr(bins),
                                   _i(0 if bins is None else bins.shape[0]), ptr(gamma), ptr(beta), _f(eps), _f(momentum), ptr(mean),
                                   ptr(invstd), ptr(ab), ptr(run_mean), ptr(run_var), ctypes.byref(fused), stream_ptr()))
    return bool(fused.value), mean, invstd, ab


def bn_lrelu_fwd_ab(raw, ab, act):
    R, C = raw.numel() // raw.shape[-1], raw.shape[-1]
    check(_lib.favae_bn_lrelu_fwd_ab(ptr(raw), ptr(ab), ptr(act), ctypes.c_longlong(R), _i(C), _i(dtype_code(act)), stream_ptr()))
    return act

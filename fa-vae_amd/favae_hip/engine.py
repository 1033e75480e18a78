"""Host-side handle of the HIP conv-VAE executor (csrc/vae_model.hip).

Owns the device buffers (torch tensors used as plain memory), converts between the reference's state_dict layout
(NCHW weights, cA/celeba_deepvanilla_vae.py:25-78) and the executor's arena layout ([Cout][tap][Cin], NHWC), and
forwards the step / epoch / encode / decode calls through the C ABI.  No arithmetic happens here.
"""
import ctypes

import numpy as np
import torch

from . import lib as L

_ll = ctypes.c_longlong
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

_raw = L.raw()
for _n in ("favae_vae_param_count", "favae_vae_buffer_count", "favae_vae_shadow_bytes", "favae_vae_workspace_bytes", "favae_vae_epoch_steps"):
    getattr(_raw, _n).restype = _ll

KIND_CONV_S2, KIND_LINEAR, KIND_CONVT_S2, KIND_CONV_S1 = 0, 1, 2, 3

LAYER_NAMES = ["enc.0", "enc.1", "enc.2", "enc.3", "enc.4", "fc", "decoder_input",
               "dec.0", "dec.1", "dec.2", "dec.3", "final_layer", "final_layer.3"]


def _conv_name(l):
    if l <= 4:
        return "enc.%d.0" % l, "enc.%d.1" % l
    if 7 <= l <= 10:
        return "dec.%d.0" % (l - 7), "dec.%d.1" % (l - 7)
    if l == 11:
        return "final_layer.0", "final_layer.1"
    return "final_layer.3", None


class VaeEngine:
    def __init__(self, z, channels, batch_max=64, dtype="f32", device="cuda:0"):
        self.z, self.ch, self.batch_max = int(z), int(channels), int(batch_max)
        self.dtype = L.F32 if dtype in ("f32", torch.float32, L.F32) else L.BF16
        self.tdtype = torch.float32 if self.dtype == L.F32 else torch.bfloat16
        self.dev = torch.device(device)
        self.nparam = _raw.favae_vae_param_count(_i(self.z), _i(self.ch))
        self.nbuf = _raw.favae_vae_buffer_count(_i(self.z), _i(self.ch))
        sh = _raw.favae_vae_shadow_bytes(_i(self.z), _i(self.ch), _i(self.dtype))
        ws = _raw.favae_vae_workspace_bytes(_i(self.z), _i(self.ch), _i(self.batch_max), _i(self.dtype))
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.P = torch.zeros(self.nparam, **f32)
        self.G = torch.zeros(self.nparam, **f32)
        self.M = torch.zeros(self.nparam, **f32)
        self.V = torch.zeros(self.nparam, **f32)
        self.BUF = torch.zeros(self.nbuf, **f32)
        self.shadow = torch.zeros(sh, dtype=torch.uint8, device=self.dev)
        self.workspace = torch.zeros(ws, dtype=torch.uint8, device=self.dev)
        self.step = 0
        self.layers = []
        info = (_ll * 11)()
        for l in range(_raw.favae_vae_num_layers()):
            L.check(_raw.favae_vae_layer_info(_i(self.z), _i(self.ch), _i(l), info))
            self.layers.append(dict(kind=info[0], Cin=info[1], Cout=info[2], taps=info[3], bn=bool(info[4]), w=info[5],
                                    b=info[6], g=info[7], be=info[8], rm=info[9], rv=info[10]))
        h = ctypes.c_void_p()
        L.check(_raw.favae_vae_create(ctypes.byref(h), _i(self.z), _i(self.ch), _i(self.batch_max), _i(self.dtype),
                                      L.ptr(self.P), L.ptr(self.G), L.ptr(self.M), L.ptr(self.V), L.ptr(self.BUF),
                                      L.ptr(self.shadow), _sz(sh), L.ptr(self.workspace), _sz(ws)))
        self.h = h
        self.metrics = torch.zeros(2, dtype=torch.float64, device=self.dev)
        # eval-mode passes over a whole data set (update_x, reconstruction diagnostic) are independent per sample (BN uses
        # the running statistics), so they run through a second executor over the SAME parameters with a larger batch:
        # 512-image GEMMs fill the chip, 64-image ones do not
        import os
        self.eval_batch = int(os.environ.get("FAVAE_EVAL_BATCH", "512"))
        self.h_eval = None
        self.workspace_eval = None

    def __del__(self):
        try:
            for name in ("h", "h_eval"):
                if getattr(self, name, None):
                    _raw.favae_vae_destroy(getattr(self, name))
                    setattr(self, name, None)
        except Exception:
            pass

    def _eval_handle(self):
        if self.h_eval is None:
            ws = _raw.favae_vae_workspace_bytes(_i(self.z), _i(self.ch), _i(self.eval_batch), _i(self.dtype))
            self.workspace_eval = torch.zeros(ws, dtype=torch.uint8, device=self.dev)
            h = ctypes.c_void_p()
            L.check(_raw.favae_vae_create(ctypes.byref(h), _i(self.z), _i(self.ch), _i(self.eval_batch), _i(self.dtype),
                                          L.ptr(self.P), L.ptr(self.G), L.ptr(self.M), L.ptr(self.V), L.ptr(self.BUF),
                                          L.ptr(self.shadow), _sz(self.shadow.numel()), L.ptr(self.workspace_eval), _sz(ws)))
            self.h_eval = h
        return self.h_eval

    # ---- layout conversion (pure data movement) ----------------------------------------------------------------
    def _w_to_arena(self, l, sd):
        d = self.layers[l]
        if l == 5:
            w = torch.cat([sd["fc_mu.weight"], sd["fc_var.weight"]], 0)                       # [2z, 4096], col = c*4+hw
            return w.reshape(2 * self.z, 1024, 4).permute(0, 2, 1).reshape(2 * self.z, 4096)
        if l == 6:
            w = sd["decoder_input.weight"]                                                    # [4096, z], row = c*4+hw
            return w.reshape(1024, 4, self.z).permute(1, 0, 2).reshape(4096, self.z)
        wname = _conv_name(l)[0] + ".weight"
        w = sd[wname]
        if d["kind"] == KIND_CONVT_S2:
            return w.permute(1, 2, 3, 0).reshape(d["Cout"], -1)                               # [Ci,Co,kh,kw] -> [Co][tap][Ci]
        return w.permute(0, 2, 3, 1).reshape(d["Cout"], -1)                                   # [Co,Ci,kh,kw] -> [Co][tap][Ci]

    def _w_from_arena(self, l, flat, out):
        d = self.layers[l]
        if l == 5:
            w = flat.reshape(2 * self.z, 4, 1024).permute(0, 2, 1).reshape(2 * self.z, 4096)
            out["fc_mu.weight"], out["fc_var.weight"] = w[:self.z].clone(), w[self.z:].clone()
        elif l == 6:
            out["decoder_input.weight"] = flat.reshape(4, 1024, self.z).permute(1, 0, 2).reshape(4096, self.z).clone()
        else:
            w = flat.reshape(d["Cout"], 3, 3, d["Cin"])
            wname = _conv_name(l)[0] + ".weight"
            out[wname] = (w.permute(3, 0, 1, 2) if d["kind"] == KIND_CONVT_S2 else w.permute(0, 3, 1, 2)).contiguous()

    def _b_to_arena(self, l, sd):
        if l == 5:
            return torch.cat([sd["fc_mu.bias"], sd["fc_var.bias"]], 0)
        if l == 6:
            return sd["decoder_input.bias"].reshape(1024, 4).permute(1, 0).reshape(4096)
        return sd[_conv_name(l)[0] + ".bias"]

    def _b_from_arena(self, l, flat, out):
        if l == 5:
            out["fc_mu.bias"], out["fc_var.bias"] = flat[:self.z].clone(), flat[self.z:].clone()
        elif l == 6:
            out["decoder_input.bias"] = flat.reshape(4, 1024).permute(1, 0).reshape(4096).clone()
        else:
            out[_conv_name(l)[0] + ".bias"] = flat.clone()

    def pack(self, sd, arena):
        """Write a reference-layout dict of parameter-shaped tensors (weights / grads / Adam moments) into `arena`."""
        host = torch.zeros(self.nparam, dtype=torch.float32)
        for l, d in enumerate(self.layers):
            w = self._w_to_arena(l, sd).reshape(-1).float()
            host[d["w"]:d["w"] + w.numel()] = w
            b = self._b_to_arena(l, sd).reshape(-1).float()
            host[d["b"]:d["b"] + b.numel()] = b
            if d["bn"]:
                bn = _conv_name(l)[1]
                host[d["g"]:d["g"] + d["Cout"]] = sd[bn + ".weight"].float()
                host[d["be"]:d["be"] + d["Cout"]] = sd[bn + ".bias"].float()
        arena.copy_(host)

    def unpack(self, arena):
        host = arena.detach().cpu()
        out = {}
        for l, d in enumerate(self.layers):
            self._w_from_arena(l, host[d["w"]:d["w"] + d["Cout"] * d["taps"] * d["Cin"]], out)
            self._b_from_arena(l, host[d["b"]:d["b"] + d["Cout"]], out)
            if d["bn"]:
                bn = _conv_name(l)[1]
                out[bn + ".weight"] = host[d["g"]:d["g"] + d["Cout"]].clone()
                out[bn + ".bias"] = host[d["be"]:d["be"] + d["Cout"]].clone()
        return out

    def load_buffers(self, sd):
        host = torch.zeros(self.nbuf, dtype=torch.float32)
        for l, d in enumerate(self.layers):
            if d["bn"]:
                bn = _conv_name(l)[1]
                host[d["rm"]:d["rm"] + d["Cout"]] = sd[bn + ".running_mean"].float()
                host[d["rv"]:d["rv"] + d["Cout"]] = sd[bn + ".running_var"].float()
        self.BUF.copy_(host)

    def export_buffers(self):
        host = self.BUF.detach().cpu()
        out = {}
        for l, d in enumerate(self.layers):
            if d["bn"]:
                bn = _conv_name(l)[1]
                out[bn + ".running_mean"] = host[d["rm"]:d["rm"] + d["Cout"]].clone()
                out[bn + ".running_var"] = host[d["rv"]:d["rv"] + d["Cout"]].clone()
        return out

    def load_state_dict(self, sd):
        self.pack(sd, self.P)
        self.load_buffers(sd)
        self.refresh_weights()

    def state_dict(self):
        out = self.unpack(self.P)
        out.update(self.export_buffers())
        return out

    # ---- execution -----------------------------------------------------------------------------------------------
    def refresh_weights(self):
        L.check(_raw.favae_vae_refresh_weights(self.h, L.stream_ptr()))

    def upload_images(self, img_nchw, allow_u8=None):
        """float32 NCHW numpy / CPU tensor -> resident NHWC device tensor (chunked so host memory stays flat).  8-bit images
        (every value exactly k / 255, what torchvision's ToTensor produces) stay resident as uint8 -- a quarter of the HBM,
        bit-identical batches (favae_gather_batch_u8); anything else as fp32.  FAVAE_DATA_U8=0 forces fp32."""
        import os
        t = torch.from_numpy(img_nchw) if isinstance(img_nchw, np.ndarray) else img_nchw
        N, C, H, W = t.shape
        assert (H, W) == (64, 64) and C == self.ch, "the executor is built for %dx64x64 images" % self.ch
        if allow_u8 is None:
            allow_u8 = os.environ.get("FAVAE_DATA_U8", "1") != "0"
        chunk = 8192
        if allow_u8:
            out = torch.empty(N, H, W, C, dtype=torch.uint8, device=self.dev)
            bad = torch.zeros(1, dtype=torch.int64, device=self.dev)
            for s in range(0, N, chunk):
                src = t[s:s + chunk].to(torch.float32).contiguous().to(self.dev)
                L.check(_raw.favae_nchw_to_nhwc_u8(L.ptr(src), ctypes.c_void_p(out[s:].data_ptr()), _ll(src.shape[0]), _i(C), _i(H * W),
                                                   L.ptr(bad), L.stream_ptr()))
                if s == 0 and int(bad.item()):        # the first chunk decides for data that is not 8-bit at all
                    break
            if int(bad.item()) == 0:
                return out
            del out
        out = torch.empty(N, H, W, C, dtype=torch.float32, device=self.dev)
        for s in range(0, N, chunk):
            src = t[s:s + chunk].to(torch.float32).contiguous().to(self.dev)
            L.check(_raw.favae_nchw_to_nhwc(L.ptr(src), ctypes.c_void_p(out[s:].data_ptr()), _ll(src.shape[0]), _i(C),
                                            _i(H * W), _i(C), L.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        return out

    def _fmt(self, data, h=None):
        """Tell the executor whether the data set it is about to read is uint8 or fp32."""
        L.check(_raw.favae_vae_set_data_u8(self.h if h is None else h, _i(1 if data.dtype == torch.uint8 else 0)))

    def train_fwdbwd(self, data, idx, B, eps, prior, tau, beta, var_x, metrics=None):
        m = self.metrics if metrics is None else metrics
        self._fmt(data)
        L.check(_raw.favae_vae_train_fwdbwd(self.h, L.ptr(data), L.ptr(idx), _i(B), L.ptr(eps), L.ptr(prior), _f(tau),
                                            _f(beta), _f(var_x), L.ptr(m), L.stream_ptr()))
        return m

    N_STAGES = 13

    def train_stages(self, data, idx, B, eps, prior, tau, beta, var_x, metrics, lo, hi):
        """The step of train_fwdbwd in pieces (favae_vae_train_stages): forward with lo == 0, backward stages [lo, hi)."""
        self._fmt(data)
        L.check(_raw.favae_vae_train_stages(self.h, L.ptr(data), L.ptr(idx), _i(B), L.ptr(eps), L.ptr(prior), _f(tau),
                                            _f(beta), _f(var_x), L.ptr(metrics), _i(lo), _i(hi), L.stream_ptr()))

    def fence(self, waiting):
        """Order torch stream `waiting` behind everything queued so far on the current and the weight-gradient stream."""
        L.check(_raw.favae_vae_fence(self.h, L.stream_ptr(), ctypes.c_void_p(waiting.cuda_stream)))

    def grad_buckets(self):
        """(arena_lo, arena_hi, stage_hi): the gradient slice that is complete once backward stages < stage_hi have run.
        Decoder first (its gradients are ready while the encoder is still in its backward pass), then enc.4 + heads,
        enc.3, and the small first encoder layers (the same four buckets as favae_vae_train_epoch_dp)."""
        w = [d["w"] for d in self.layers]
        return [(w[6], self.nparam, 7), (w[4], w[6], 9), (w[3], w[4], 10), (0, w[3], self.N_STAGES)]

    def adam(self, lr):
        self.step += 1
        L.check(_raw.favae_vae_adam(self.h, _f(lr), _f(0.9), _f(0.999), _f(1e-8), _ll(self.step), L.stream_ptr()))

    @staticmethod
    def epoch_steps(n, batch):
        """Optimiser steps (= rows of `metrics`) of an epoch over n images: ceil(n / batch), plus one in grouped-batch mode
        (batch > 64) when the last batch is not a whole number of BatchNorm groups (favae_vae_epoch_steps)."""
        return int(_raw.favae_vae_epoch_steps(_ll(n), _i(batch)))

    def train_epoch(self, data, perm, eps, prior, tau, beta, var_x, lr, batch, metrics):
        n = perm.numel()
        if metrics.shape[0] < self.epoch_steps(n, batch):
            raise ValueError("train_epoch: metrics has %d rows, the epoch has %d steps" % (metrics.shape[0], self.epoch_steps(n, batch)))
        self._fmt(data)
        L.check(_raw.favae_vae_train_epoch(self.h, L.ptr(data), L.ptr(perm), _ll(n), _i(batch), L.ptr(eps), L.ptr(prior),
                                           _f(tau), _f(beta), _f(var_x), _f(lr), _ll(self.step), L.ptr(metrics),
                                           L.stream_ptr()))
        self.step += self.epoch_steps(n, batch)

    def train_epoch_dp(self, comm, data, perm, eps, prior, tau, beta, var_x, lr, batch, metrics):
        """One data-parallel epoch in one call: bucketed gradient all-reduce + bucket-wise Adam on the communicator's stream."""
        n = perm.numel()
        self._fmt(data)
        L.check(_raw.favae_vae_train_epoch_dp(self.h, comm, L.ptr(data), L.ptr(perm), _ll(n), _i(batch), L.ptr(eps), L.ptr(prior),
                                              _f(tau), _f(beta), _f(var_x), _f(lr), _ll(self.step), L.ptr(metrics),
                                              L.stream_ptr()))
        self.step += self.epoch_steps(n, batch)

    def poll_error(self):
        """Raise if a dependency wait of an earlier epoch gave up (favae_vae_poll_error); call after a synchronisation."""
        L.check(_raw.favae_vae_poll_error(self.h))

    def encode(self, data, idx, B, xmean, xcov, train=False):
        self._fmt(data)
        L.check(_raw.favae_vae_encode(self.h, L.ptr(data), L.ptr(idx), _i(B), ctypes.c_void_p(xmean.data_ptr()),
                                      ctypes.c_void_p(xcov.data_ptr()), _i(int(train)), L.stream_ptr()))

    def encode_all(self, data, xmean, xcov, batch=64):
        """update_x over a resident data set; `batch` is the reference's loader batch, the eval executor uses its own."""
        self._fmt(data, self._eval_handle())
        L.check(_raw.favae_vae_encode_all(self._eval_handle(), L.ptr(data), _ll(data.shape[0]), _i(max(batch, self.eval_batch)),
                                          L.ptr(xmean), L.ptr(xcov), L.stream_ptr()))

    def decode(self, zin, B, out, train=False):
        L.check(_raw.favae_vae_decode(self.h, ctypes.c_void_p(zin.data_ptr()), _i(B), ctypes.c_void_p(out.data_ptr()),
                                      _i(int(train)), L.stream_ptr()))

    def reconstruct_stats(self, zin, data, acc3, batch=64):
        self._fmt(data, self._eval_handle())
        L.check(_raw.favae_vae_reconstruct_stats(self._eval_handle(), L.ptr(zin), L.ptr(data), _ll(zin.shape[0]),
                                                 _i(max(batch, self.eval_batch)), L.ptr(acc3), L.stream_ptr()))

    def nhwc_to_nchw_f64(self, x_nhwc, out):
        n, hh, ww, c = x_nhwc.shape
        L.check(_raw.favae_nhwc_to_nchw_f64(L.ptr(x_nhwc), L.ptr(out), _ll(n), _i(c), _i(hh * ww), _i(c), L.stream_ptr()))

    def debug(self, which, layer, shape, dtype):
        p = ctypes.c_void_p()
        L.check(_raw.favae_vae_debug_ptr(self.h, _i(which), _i(layer), ctypes.byref(p)))
        n = int(np.prod(shape))
        out = torch.empty(n, dtype=dtype, device=self.dev)
        import ctypes as ct
        hip = ct.CDLL("libamdhip64.so")
        torch.cuda.synchronize()
        hip.hipMemcpy(ct.c_void_p(out.data_ptr()), p, ct.c_size_t(n * out.element_size()), ct.c_int(3))
        return out.reshape(shape)

    # snapshot / restore of everything restore_weights() saves (cA/celeba_deepvanilla_vae.py:297-329), device side
    def snapshot(self):
        return dict(P=self.P.clone(), M=self.M.clone(), V=self.V.clone(), BUF=self.BUF.clone(), step=self.step)

    def restore(self, snap):
        self.P.copy_(snap["P"]); self.M.copy_(snap["M"]); self.V.copy_(snap["V"]); self.BUF.copy_(snap["BUF"])
        self.step = snap["step"]
        self.refresh_weights()

"""The oracle's training step (oracle/vae_torch.py) with the ROUNDING POINTS of the HIP bf16 path, on the CPU in fp32 arithmetic.

Test infrastructure.  The bf16 step differs from the fp32 oracle by design (storage roundings: ~1 % of the LeakyReLU slopes per
layer flip, gradients of the small BatchNorm tensors move by tens of per cent of their scale), which is why
tests/test_gpu_vae_step.py can hold its whole-step gradients against the fp32 oracle only loosely.  This emulation rounds where the
executor rounds, so the same step can be held tightly: a wrong tap, a dropped bias or a missing term shows.

Rounding points (csrc/vae_model.hip, bf16 executor with exact ends):
  * weights of every GEMM: the bf16 shadow of the fp32 master (encoder.0 forward and the fc heads forward: fp32 master, exact ends);
  * activations (after BatchNorm + LeakyReLU), the decoder_input output and the latent sample z: stored as bf16;
  * the gradient w.r.t. every conv / linear output (draw): stored as bf16, read by both the weight and the input gradient;
  * encoder.0: forward on the unrounded image, weight gradient on the bf16 image (the hi plane of the folded batch);
  * fc heads: forward from the unrounded encoder.4 activation with fp32 weights, gradients through the bf16 activation / shadow;
  * the gradient w.r.t. final_layer.0's activation (dact of the largest layer): stored as bf16 (its BatchNorm-backward sums are taken
    from the unrounded values in the producing launch; the emulation takes them from the rounded ones);
  * the conv outputs (raw) of decoder.3 and final_layer.0: stored as bf16, their BatchNorm statistics from the unrounded values;
  * the other conv outputs (raw) and activation gradients (dact), BatchNorm statistics, loss: fp32 (fp64 for the KL and the sums)."""
import torch
import torch.nn.functional as F

from oracle import vae_torch as vt


def _r(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundFwd(torch.autograd.Function):
    """bf16 storage of a forward quantity: rounded value, straight-through gradient."""
    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """bf16 storage of the gradient w.r.t. this quantity."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


rf, rb = _RoundFwd.apply, _RoundBwd.apply


def _value_of(exact, through):
    """Forward value of `exact`, gradient path of `through`."""
    return through + (exact - through).detach()


def _bn_act(raw, P, B, pre, store=True, raw_bf=False):
    B[pre + ".1.num_batches_tracked"] += 1
    if raw_bf:
        # the conv output is stored as bf16, its statistics come from the unrounded values: normalise the rounded values with the
        # statistics of the exact ones (affine in the input: y(r) = y(x) + (r - x) * gamma / sqrt(var + eps))
        y0 = F.batch_norm(raw, B[pre + ".1.running_mean"], B[pre + ".1.running_var"], P[pre + ".1.weight"], P[pre + ".1.bias"], True,
                          vt.BN_MOM, vt.BN_EPS)
        var = raw.detach().var(dim=(0, 2, 3), unbiased=False)
        scale = (P[pre + ".1.weight"].detach() / torch.sqrt(var + vt.BN_EPS)).view(1, -1, 1, 1)
        y = y0 + ((_r(raw) - raw) * scale).detach()
    else:
        y = F.batch_norm(raw, B[pre + ".1.running_mean"], B[pre + ".1.running_var"], P[pre + ".1.weight"], P[pre + ".1.bias"], True,
                         vt.BN_MOM, vt.BN_EPS)
    y = F.leaky_relu(y, vt.LRELU)
    return (rf(y), y) if store else (y, y)


def loss_terms_bf16(P, B, images, mu_P, tau, var_x, beta, eps):
    """vt.loss_terms with the storage roundings of the HIP bf16 step."""
    # encoder.0: exact forward, weight gradient through the bf16 image
    w0, b0 = P["enc.0.0.weight"], P["enc.0.0.bias"]
    raw = _value_of(F.conv2d(images, w0, b0, stride=2, padding=1), F.conv2d(_r(images), w0, b0, stride=2, padding=1))
    x, _ = _bn_act(rb(raw), P, B, "enc.0")
    y = None
    for i in range(1, 5):
        pre = "enc.%d" % i
        raw = F.conv2d(x, rf(P[pre + ".0.weight"]), P[pre + ".0.bias"], stride=2, padding=1)
        x, y = _bn_act(rb(raw), P, B, pre)
    xf, yf = torch.flatten(x, start_dim=1), torch.flatten(y, start_dim=1)
    wm, wv = P["fc_mu.weight"], P["fc_var.weight"]
    mu = rb(_value_of(F.linear(yf, wm, P["fc_mu.bias"]), F.linear(xf, rf(wm), P["fc_mu.bias"])))
    log_var = rb(_value_of(F.linear(yf, wv, P["fc_var.bias"]), F.linear(xf, rf(wv), P["fc_var.bias"]))) + 1e-6
    std = torch.exp(0.5 * log_var)
    z = rf(eps.mul(std).add(mu))
    d = rf(rb(F.linear(z, rf(P["decoder_input.weight"]), P["decoder_input.bias"]))).view(-1, 1024, 2, 2)
    x = d
    for i in range(4):
        pre = "dec.%d" % i
        raw = F.conv_transpose2d(x, rf(P[pre + ".0.weight"]), P[pre + ".0.bias"], stride=2, padding=1, output_padding=1)
        x, _ = _bn_act(rb(raw), P, B, pre, raw_bf=(i == 3))
    raw = F.conv_transpose2d(x, rf(P["final_layer.0.weight"]), P["final_layer.0.bias"], stride=2, padding=1, output_padding=1)
    x, _ = _bn_act(rb(raw), P, B, "final_layer", raw_bf=True)
    x = rb(x)   # the gradient w.r.t. final_layer.0's activation (dact[11]) is stored as bf16
    rec = torch.sigmoid(rb(F.conv2d(x, rf(P["final_layer.3.weight"]), P["final_layer.3.bias"], padding=1)))
    std_P = torch.sqrt(torch.ones_like(mu_P) * (1 / tau))
    rec_err = -vt.gaussian_ll(images, rec, var_x)
    kl = vt.kl_div(mu, std, mu_P, std_P)
    return rec_err + beta * kl, rec_err, kl, mu, std, z, rec

"""GPU parity of the implicit-GEMM conv kernels (C ABI: favae_conv_igemm / favae_conv_wgrad) against torch-CPU
fp32 references of the same op (the ops the reference calls: nn.Conv2d / nn.ConvTranspose2d / nn.Linear and
their autograd gradients, condition_a_VAE/celeba_deepvanilla_vae.py:25-99,249).

Tolerances: fp32 path rel 1e-5 of the output scale (exact-fp32 MFMA, only the summation order differs);
bf16 path compares against the same op evaluated on bf16-rounded operands, rel 1e-4 (fp32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from favae_hip import lib
    return lib


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def q(x, dt):
    return x.to(dt).to(torch.float32)


def assert_close(got, ref, tol):
    got, ref = got.double(), ref.double()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "max err %.3e vs scale %.3e (tol %.1e)" % (err, scale, tol)


DT = [(torch.float32, 1e-5), (torch.bfloat16, 1e-4)]


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,Cin,Cout,H,splitk", [(4, 64, 128, 16, 1), (3, 32, 64, 8, 0), (2, 256, 512, 8, 4),
                                                 (5, 3, 64, 64, 1), (160, 3, 64, 64, 0), (64, 512, 1024, 4, 0)])
def test_conv_s2_forward(dt, tol, Nb, Cin, Cout, H, splitk):
    lib = _lib()
    g = torch.Generator().manual_seed(Nb * 1000 + Cin)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1, dt)
    b = torch.randn(Cout, generator=g)
    ref = nhwc(F.conv2d(x, w, b, stride=2, padding=1))
    xd = nhwc(x).to(dt).cuda()
    wd = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().to(dt).cuda()
    out = torch.empty(Nb, H // 2, H // 2, Cout, device="cuda")
    lib.conv_igemm(xd, wd, b.cuda(), out, 3, 2, 1, 0, splitk)
    assert_close(out.cpu(), ref, tol)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,Cin,Cout,H,splitk", [(4, 128, 64, 4, 1), (3, 64, 64, 32, 0), (64, 1024, 512, 2, 0)])
def test_convT_s2_forward(dt, tol, Nb, Cin, Cout, H, splitk):
    lib = _lib()
    g = torch.Generator().manual_seed(7 + Cin)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = q(torch.randn(Cin, Cout, 3, 3, generator=g) * 0.1, dt)
    b = torch.randn(Cout, generator=g)
    ref = nhwc(F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1))
    xd = nhwc(x).to(dt).cuda()
    wd = w.permute(1, 2, 3, 0).reshape(Cout, 9, Cin).contiguous().to(dt).cuda()
    out = torch.empty(Nb, 2 * H, 2 * H, Cout, device="cuda")
    lib.conv_igemm(xd, wd, b.cuda(), out, 3, 2, 1, 1, splitk)
    assert_close(out.cpu(), ref, tol)


@pytest.mark.parametrize("dt,tol", DT)
def test_conv_s2_input_gradient_is_scatter_form(dt, tol):
    lib = _lib()
    g = torch.Generator().manual_seed(11)
    Nb, Cin, Cout, H = 3, 64, 128, 16
    x = torch.randn(Nb, Cin, H, H, generator=g, requires_grad=True)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1, dt)
    dy = q(torch.randn(Nb, Cout, H // 2, H // 2, generator=g), dt)
    F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
    dyd = nhwc(dy).to(dt).cuda()
    wt = w.permute(1, 2, 3, 0).reshape(Cin, 9, Cout).contiguous().to(dt).cuda()   # [Cin][tap][Cout]
    out = torch.empty(Nb, H, H, Cin, device="cuda")
    lib.conv_igemm(dyd, wt, None, out, 3, 2, 1, 1, 0)
    assert_close(out.cpu(), nhwc(x.grad), tol)


@pytest.mark.parametrize("dt,tol", DT)
def test_convT_s2_input_gradient_is_gather_form(dt, tol):
    lib = _lib()
    g = torch.Generator().manual_seed(12)
    Nb, Cin, Cout, H = 3, 128, 64, 8
    x = torch.randn(Nb, Cin, H, H, generator=g, requires_grad=True)
    w = q(torch.randn(Cin, Cout, 3, 3, generator=g) * 0.1, dt)
    dy = q(torch.randn(Nb, Cout, 2 * H, 2 * H, generator=g), dt)
    F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1).backward(dy)
    dyd = nhwc(dy).to(dt).cuda()
    wt = w.permute(0, 2, 3, 1).reshape(Cin, 9, Cout).contiguous().to(dt).cuda()   # [CinT][tap][CoutT]
    out = torch.empty(Nb, H, H, Cin, device="cuda")
    lib.conv_igemm(dyd, wt, None, out, 3, 2, 1, 0, 0)
    assert_close(out.cpu(), nhwc(x.grad), tol)


@pytest.mark.parametrize("dt,tol", DT)
def test_final_conv_s1_forward_and_input_gradient(dt, tol):
    lib = _lib()
    g = torch.Generator().manual_seed(13)
    Nb, Cin, Cout, H = 2, 64, 3, 64
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt).requires_grad_(True)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1, dt)
    b = torch.randn(Cout, generator=g)
    y = F.conv2d(x, w, b, stride=1, padding=1)
    dy = q(torch.randn(Nb, Cout, H, H, generator=g), dt)
    y.backward(dy)
    xd = nhwc(x.detach()).to(dt).cuda()
    wd = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().to(dt).cuda()
    out = torch.empty(Nb, H, H, Cout, device="cuda")
    lib.conv_igemm(xd, wd, b.cuda(), out, 3, 1, 1, 0, 0)
    assert_close(out.cpu(), nhwc(y.detach()), tol)
    dyd = nhwc(dy).to(dt).cuda()
    wt = w.permute(1, 2, 3, 0).reshape(Cin, 9, Cout).contiguous().to(dt).cuda()
    dx = torch.empty(Nb, H, H, Cin, device="cuda")
    lib.conv_igemm(dyd, wt, None, dx, 3, 1, 1, 1, 0)
    assert_close(dx.cpu(), nhwc(x.grad), tol)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,K,Nout", [(64, 4096, 40), (64, 20, 4096), (17, 40, 4096)])
def test_linear_as_1x1(dt, tol, Nb, K, Nout):
    lib = _lib()
    g = torch.Generator().manual_seed(K)
    x = q(torch.randn(Nb, K, generator=g), dt)
    w = q(torch.randn(Nout, K, generator=g) * 0.05, dt)
    b = torch.randn(Nout, generator=g)
    ref = F.linear(x, w, b)
    out = torch.empty(Nb, 1, 1, Nout, device="cuda")
    lib.conv_igemm(x.to(dt).cuda().view(Nb, 1, 1, K), w.to(dt).cuda().view(Nout, 1, K), b.cuda(), out, 1, 1, 0, 0, 0)
    assert_close(out.cpu().view(Nb, Nout), ref, tol)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,Cin,Cout,H", [(4, 64, 128, 16), (64, 512, 1024, 4), (3, 3, 64, 64)])
def test_conv_s2_weight_gradient(dt, tol, Nb, Cin, Cout, H):
    lib = _lib()
    g = torch.Generator().manual_seed(21 + Cin)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, requires_grad=True)
    dy = q(torch.randn(Nb, Cout, H // 2, H // 2, generator=g), dt)
    F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    dw = torch.zeros(Cout, 9, Cin, device="cuda")
    lib.conv_wgrad(nhwc(dy).to(dt).cuda(), 0, nhwc(x).to(dt).cuda(), 1, dw, Nb, H // 2, H // 2, H, H, 3, 2, 1, 0)
    assert_close(dw.cpu(), ref, tol)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,Cin,Cout,H", [(4, 128, 64, 8), (64, 1024, 512, 2), (2, 64, 64, 32)])
def test_convT_s2_weight_gradient(dt, tol, Nb, Cin, Cout, H):
    lib = _lib()
    g = torch.Generator().manual_seed(31 + Cin)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = torch.randn(Cin, Cout, 3, 3, generator=g, requires_grad=True)
    dy = q(torch.randn(Nb, Cout, 2 * H, 2 * H, generator=g), dt)
    F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1).backward(dy)
    ref = w.grad.permute(1, 2, 3, 0).reshape(Cout, 9, Cin)                         # master layout [CoutT][tap][CinT]
    dw = torch.zeros(Cout, 9, Cin, device="cuda")
    lib.conv_wgrad(nhwc(dy).to(dt).cuda(), 1, nhwc(x).to(dt).cuda(), 0, dw, Nb, H, H, 2 * H, 2 * H, 3, 2, 1, 0)
    assert_close(dw.cpu(), ref, tol)


@pytest.mark.parametrize("kind,Nb,Cin,Cout,H", [("convT", 8, 64, 64, 32), ("convT", 5, 64, 64, 32), ("convT", 8, 128, 64, 16), ("convT", 16, 256, 128, 8),
                                                ("conv", 8, 64, 128, 32), ("conv", 16, 128, 256, 16), ("conv", 3, 64, 64, 64)])
def test_weight_gradient_with_all_nine_taps_in_one_block(kind, Nb, Cin, Cout, H):
    """wgrad9_kernel (one pass over the operands for the nine taps of a stride-2 layer, csrc/igemm.hip), forced through
    favae_debug_force_tile(-1, 9), against autograd of the layer (celeba_deepvanilla_vae.py:31-35, 55-60, 69-75 with loss.backward() :249)
    on the same bf16-rounded operands, and against igemm_pm on the same call (same products, another order of the fp32 sums).
    H: input height of the layer; the small grid of the gradient is H / 2 (conv) or H (transposed) and must be 8, 16 or 32 wide."""
    import ctypes
    lib = _lib()
    dt, tol = torch.bfloat16, 1e-4
    g = torch.Generator().manual_seed(91 + Cin + Nb)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    if kind == "conv":
        w = torch.randn(Cout, Cin, 3, 3, generator=g, requires_grad=True)
        dy = q(torch.randn(Nb, Cout, H // 2, H // 2, generator=g), dt)
        F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
        ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
        args = (nhwc(dy).to(dt).cuda(), 0, nhwc(x).to(dt).cuda(), 1)
        geo = (Nb, H // 2, H // 2, H, H, 3, 2, 1)
    else:
        w = torch.randn(Cin, Cout, 3, 3, generator=g, requires_grad=True)
        dy = q(torch.randn(Nb, Cout, 2 * H, 2 * H, generator=g), dt)
        F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1).backward(dy)
        ref = w.grad.permute(1, 2, 3, 0).reshape(Cout, 9, Cin)
        args = (nhwc(dy).to(dt).cuda(), 1, nhwc(x).to(dt).cuda(), 0)
        geo = (Nb, H, H, 2 * H, 2 * H, 3, 2, 1)
    raw_ = lib.raw()
    out = {}
    try:
        for name, force in (("pm", 2), ("w9", 9)):
            lib.check(raw_.favae_debug_force_tile(ctypes.c_int(-1), ctypes.c_int(force)))
            dw = torch.full((Cout, 9, Cin), 0.5, device="cuda")       # splitk 0: the gradient is ADDED to what dw holds
            lib.conv_wgrad(*args, dw, *geo, 0)
            out[name] = dw.cpu() - 0.5
    finally:
        lib.check(raw_.favae_debug_force_tile(ctypes.c_int(-1), ctypes.c_int(-1)))
    assert_close(out["w9"], ref, tol)
    scale = float(ref.abs().max())
    assert float((out["w9"] - out["pm"]).abs().max()) <= 2e-5 * scale + 1e-6     # fp32 sums of the same products in another order (+ the 0.5 offset's rounding)


@pytest.mark.parametrize("dt,tol", DT)
def test_final_conv_and_linear_weight_gradient(dt, tol):
    lib = _lib()
    g = torch.Generator().manual_seed(41)
    Nb, Cin, Cout, H = 2, 64, 3, 64
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, requires_grad=True)
    dy = q(torch.randn(Nb, Cout, H, H, generator=g), dt)
    F.conv2d(x, w, None, stride=1, padding=1).backward(dy)
    dw = torch.zeros(Cout, 9, Cin, device="cuda")
    lib.conv_wgrad(nhwc(dy).to(dt).cuda(), 0, nhwc(x).to(dt).cuda(), 1, dw, Nb, H, H, H, H, 3, 1, 1, 0)
    assert_close(dw.cpu(), w.grad.permute(0, 2, 3, 1).reshape(Cout, 9, Cin), tol)
    # linear: dW[out][in] = dy^T x
    B, K, Nout = 64, 4096, 40
    xl = q(torch.randn(B, K, generator=g), dt)
    dyl = q(torch.randn(B, Nout, generator=g), dt)
    dwl = torch.zeros(Nout, 1, K, device="cuda")
    lib.conv_wgrad(dyl.to(dt).cuda(), 0, xl.to(dt).cuda(), 0, dwl, B, 1, 1, 1, 1, 1, 1, 0, 0)
    assert_close(dwl.cpu().view(Nout, K), dyl.t() @ xl, tol)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("kind,Nb,Cin,Cout,H", [("conv", 64, 64, 128, 32), ("conv", 7, 3, 64, 36), ("convT", 32, 128, 64, 16)])
def test_epilogue_sums_feed_batchnorm_forward(dt, tol, kind, Nb, Cin, Cout, H):
    """favae_conv_igemm_red mode 1 + favae_bn_train_fwd_bins == conv followed by nn.BatchNorm2d(train) + LeakyReLU
    (celeba_deepvanilla_vae.py:31-35, 50-59): the statistics ride in the GEMM epilogue."""
    import ctypes
    lib = _lib()
    g = torch.Generator().manual_seed(5 + Cin + H)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    b = torch.randn(Cout, generator=g)
    if kind == "conv":
        w = q(torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1, dt)
        y = F.conv2d(x, w, b, stride=2, padding=1)
        wd = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous().to(dt).cuda()
    else:
        w = q(torch.randn(Cin, Cout, 3, 3, generator=g) * 0.1, dt)
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
        wd = w.permute(1, 2, 3, 0).reshape(Cout, 9, Cin).contiguous().to(dt).cuda()
    Ho = y.shape[-1]
    R = Nb * Ho * Ho
    out = torch.empty(Nb, Ho, Ho, Cout, device="cuda")
    bins = torch.zeros(8, 2, Cout, dtype=torch.float64, device="cuda")
    fused = lib.conv_igemm_red(nhwc(x).to(dt).cuda(), wd, b.cuda(), out, 3, 2, 1, int(kind == "convT"), bins, 1)
    assert fused, "this shape runs unsplit on 64x64 tiles and must take the sums"
    assert_close(out.cpu(), nhwc(y), tol)
    o64 = out.double().view(R, Cout)
    tot = bins.sum(0)
    assert_close(tot[0].cpu(), o64.sum(0).cpu(), 1e-9)
    assert_close(tot[1].cpu(), (o64 * o64).sum(0).cpu(), 1e-9)
    gamma = torch.rand(Cout, generator=g) + 0.5
    beta = torch.randn(Cout, generator=g)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    ref = F.leaky_relu(F.batch_norm(out.cpu().permute(0, 3, 1, 2).double(), rm.double(), rv.double(), gamma.double(), beta.double(),
                                    True, 0.1, 1e-5), 0.01)
    act = torch.empty(R, Cout, device="cuda", dtype=dt)
    mean, invstd = torch.zeros(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    rmd, rvd = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    i_, ll_, f_ = ctypes.c_int, ctypes.c_longlong, ctypes.c_float
    gd, bd = gamma.cuda(), beta.cuda()
    lib.check(lib.raw().favae_bn_train_fwd_bins(lib.ptr(out), lib.ptr(bins), i_(8), lib.ptr(gd), lib.ptr(bd), lib.ptr(act),
                                                lib.ptr(mean), lib.ptr(invstd), lib.ptr(rmd), lib.ptr(rvd), ll_(R), i_(Cout), f_(1e-5),
                                                f_(0.1), i_(lib.dtype_code(act)), lib.stream_ptr()))
    assert_close(act.float().cpu(), nhwc(ref).reshape(R, Cout), 1e-5 if dt == torch.float32 else 8e-3)
    bm = o64.mean(0).cpu()
    bv = o64.var(0, unbiased=False).cpu()
    assert_close(mean.cpu(), bm, 1e-6)
    assert_close(invstd.cpu(), 1.0 / torch.sqrt(bv + 1e-5), 1e-6)
    assert_close(rmd.cpu(), 0.1 * bm, 1e-6)
    assert_close(rvd.cpu(), 0.9 + 0.1 * bv * R / (R - 1), 1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,Nb,C0,C1,C2,H,splitk", [("conv", 16, 3, 64, 128, 32, 0),      # enc.0 -> enc.1
                                                      ("conv", 64, 64, 128, 256, 16, 0),    # enc.1 -> enc.2: 256 tiles x 18 chunks, two reduction groups
                                                      ("conv", 64, 128, 256, 512, 8, 0),    # enc.2 -> enc.3: split reduction
                                                      ("convT", 16, 256, 128, 64, 8, 0),    # dec.2 -> dec.3
                                                      ("convT", 5, 128, 64, 64, 16, 0),     # dec.3 -> final_layer.0, ragged tiles
                                                      ("conv", 3, 64, 128, 64, 8, 1)])
def test_batchnorm_and_leakyrelu_applied_in_the_consumer_load(dt, kind, Nb, C0, C1, C2, H, splitk):
    """Conv -> BatchNorm(train) -> LeakyReLU -> Conv without a launch between the convolutions (favae_conv_igemm_bn): the first
    launch publishes the BatchNorm quantities from its epilogue sums (last block), the second reads the first one's fp32 output
    and forms the activation in its operand load.  Against the three-launch chain (favae_conv_igemm_red mode 1,
    favae_bn_train_fwd_bins, favae_conv_igemm): same statistics (1e-6: the bins are added in another order), same running
    statistics, output within 1e-5 (fp32) / 2e-3 (bf16: an activation may round the other way
    where a differs in its last bit) and favae_bn_lrelu_fwd_ab reproduces the materialised activation."""
    import ctypes
    lib = _lib()
    g = torch.Generator().manual_seed(11 * C1 + Nb)
    x = q(torch.randn(Nb, C0, H, H, generator=g), dt)
    w1 = q(torch.randn(C1, C0, 3, 3, generator=g) * 0.2, dt)
    b1 = torch.randn(C1, generator=g)
    gam, bet = (torch.rand(C1, generator=g) + 0.5).cuda(), torch.randn(C1, generator=g).cuda()
    H1 = H // 2
    w1d = w1.permute(0, 2, 3, 1).reshape(C1, 9, C0).contiguous().to(dt).cuda()
    raw1 = torch.empty(Nb, H1, H1, C1, device="cuda")
    bins = torch.zeros(8, 2, C1, dtype=torch.float64, device="cuda")
    rm1, rv1 = torch.zeros(C1, device="cuda"), torch.ones(C1, device="cuda")
    took, mean1, inv1, ab = lib.conv_igemm_bn(nhwc(x).to(dt).cuda(), None, w1d, b1.cuda(), raw1, 3, 2, 1, 0, bins=bins, gamma=gam, beta=bet,
                                              publish=True, run_mean=rm1, run_var=rv1, splitk=1)
    assert took
    scatter = int(kind == "convT")
    H2 = H1 * 2 if scatter else H1 // 2
    w2 = q(torch.randn(C2, C1, 3, 3, generator=g) * 0.1, dt)
    w2d = w2.permute(0, 2, 3, 1).reshape(C2, 9, C1).contiguous().to(dt).cuda()      # [Cout][tap][Cin] in both forms
    b2 = torch.randn(C2, generator=g).cuda()
    # three launches
    act = torch.empty(Nb, H1, H1, C1, dtype=dt, device="cuda")
    mean0, inv0 = torch.empty(C1, device="cuda"), torch.empty(C1, device="cuda")
    rm0, rv0 = torch.zeros(C1, device="cuda"), torch.ones(C1, device="cuda")
    i_, f_, ll_ = ctypes.c_int, ctypes.c_float, ctypes.c_longlong
    lib.check(lib.raw().favae_bn_train_fwd_bins(lib.ptr(raw1), lib.ptr(bins), i_(8), lib.ptr(gam), lib.ptr(bet), lib.ptr(act),
                                                lib.ptr(mean0), lib.ptr(inv0), lib.ptr(rm0), lib.ptr(rv0), ll_(Nb * H1 * H1), i_(C1),
                                                f_(1e-5), f_(0.1), i_(lib.dtype_code(act)), lib.stream_ptr()))
    ref = torch.empty(Nb, H2, H2, C2, device="cuda")
    lib.conv_igemm(act, w2d, b2, ref, 3, 2, 1, scatter, splitk)
    # one launch
    out = torch.empty_like(ref)
    bins2 = torch.zeros(8, 2, C2, dtype=torch.float64, device="cuda")
    took2, *_ = lib.conv_igemm_bn(raw1, ab, w2d, b2, out, 3, 2, 1, scatter, bins=bins2, splitk=splitk)
    torch.cuda.synchronize()
    assert_close(mean1.cpu(), mean0.cpu(), 1e-6)
    assert_close(inv1.cpu(), inv0.cpu(), 1e-6)
    assert_close(rm1.cpu(), rm0.cpu(), 1e-6)
    assert_close(rv1.cpu(), rv0.cpu(), 1e-6)
    assert_close(out.cpu(), ref.cpu(), 1e-5 if dt == torch.float32 else 2e-3)
    act2 = lib.bn_lrelu_fwd_ab(raw1, ab, torch.empty_like(act))
    same = (act2 == act).float().mean().item()
    assert same > 0.999, same                                     # (identical unless a / b differ in the last bit)
    assert_close(act2.float().cpu(), act.float().cpu(), 1e-5 if dt == torch.float32 else 1e-2)
    if took2:   # statistics of the second convolution from the same launch
        r = out.reshape(-1, C2).double()
        assert_close(bins2.sum(0)[0].cpu(), r.sum(0).cpu(), 1e-6)
        assert_close(bins2.sum(0)[1].cpu(), (r * r).sum(0).cpu(), 1e-6)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("kind,Nb,Cin,Cout,H", [("conv", 64, 64, 128, 32), ("convT", 32, 128, 64, 32)])
def test_epilogue_sums_feed_batchnorm_backward(dt, tol, kind, Nb, Cin, Cout, H):
    """Input gradient of a conv whose input came out of BN + LeakyReLU: favae_conv_igemm_red mode 2 takes sum(dd) and
    sum(dd * xhat) in the epilogue, favae_bn_lrelu_bwd_bins finishes the BN backward in one pass; against
    favae_bn_lrelu_bwd fed by the plain GEMM (autograd of celeba_deepvanilla_vae.py:31-35 under :249)."""
    import ctypes
    lib = _lib()
    g = torch.Generator().manual_seed(15 + Cin + H)
    # the layer in question maps Cin -> Cout; its input gradient has Cin channels = channels of the BN layer before it
    if kind == "conv":     # dgrad of Conv2d s2: scatter form, dy on the small grid
        Hs, Hl, scatter = H // 2, H, 1
    else:                  # dgrad of ConvTranspose2d s2: gather form, dy on the large grid
        Hs, Hl, scatter = H, H // 2, 0
    Hdy, Hout = Hs, Hl
    dy = q(torch.randn(Nb, Hdy, Hdy, Cout, generator=g), dt)
    wt = q(torch.randn(Cin, 9, Cout, generator=g) * 0.1, dt)
    R = Nb * Hout * Hout
    raw = (torch.randn(R, Cin, generator=g) * 2 + 0.3).cuda()
    gamma, beta = (torch.rand(Cin, generator=g) + 0.5).cuda(), torch.randn(Cin, generator=g).cuda()
    i_, ll_, f_, sz_ = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_size_t
    rawl = lib.raw()
    scratch = torch.zeros(2 * 1024 * 64 + 4096, dtype=torch.float64, device="cuda")
    act = torch.empty(R, Cin, device="cuda", dtype=dt)
    mean, invstd = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    lib.check(rawl.favae_bn_train_fwd(lib.ptr(raw), lib.ptr(gamma), lib.ptr(beta), lib.ptr(act), lib.ptr(mean), lib.ptr(invstd),
                                      lib.ptr(None), lib.ptr(None), ll_(R), i_(Cin), f_(1e-5), f_(0.1), lib.ptr(scratch),
                                      sz_(scratch.numel()), i_(lib.dtype_code(act)), lib.stream_ptr()))
    dyd, wtd = dy.to(dt).cuda(), wt.to(dt).cuda()
    dact0 = torch.empty(Nb, Hout, Hout, Cin, device="cuda")
    lib.conv_igemm(dyd, wtd, None, dact0, 3, 2, 1, scatter, 0)
    draw0 = torch.empty(R, Cin, device="cuda", dtype=dt)
    dg0, db0, dbias0 = (torch.zeros(Cin, device="cuda") for _ in range(3))
    s1s2 = torch.zeros(2 * Cin, device="cuda")
    lib.check(rawl.favae_bn_lrelu_bwd(lib.ptr(dact0), lib.ptr(act), lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma),
                                      lib.ptr(draw0), lib.ptr(dg0), lib.ptr(db0), lib.ptr(dbias0), ll_(R), i_(Cin), lib.ptr(scratch),
                                      sz_(scratch.numel()), lib.ptr(s1s2), i_(lib.dtype_code(act)), lib.stream_ptr()))
    bins = torch.zeros(16, 2, Cin, dtype=torch.float64, device="cuda")
    dact1 = torch.empty_like(dact0)
    fused = lib.conv_igemm_red(dyd, wtd, None, dact1, 3, 2, 1, scatter, bins, 2, act=act, raw=raw, mean=mean, invstd=invstd)
    assert fused
    assert torch.equal(dact1, dact0)
    draw1 = torch.empty_like(draw0)
    dg1, db1, dbias1 = (torch.zeros(Cin, device="cuda") for _ in range(3))
    lib.check(rawl.favae_bn_lrelu_bwd_bins(lib.ptr(dact1), lib.ptr(act), lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma),
                                           lib.ptr(bins), i_(16), lib.ptr(draw1), lib.ptr(dg1), lib.ptr(db1), lib.ptr(dbias1), ll_(R),
                                           i_(Cin), lib.ptr(scratch), sz_(scratch.numel()), i_(lib.dtype_code(act)), lib.stream_ptr()))
    assert_close(dg1.cpu(), dg0.cpu(), 1e-6)
    assert_close(db1.cpu(), db0.cpu(), 1e-6)
    assert_close(draw1.float().cpu(), draw0.float().cpu(), 1e-6 if dt == torch.float32 else 8e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C,dact_bf", [(64 * 64 * 8, 64, False), (64 * 64 * 8, 64, True), (3000, 128, False)])
def test_batchnorm_backward_without_the_stored_activation(dt, R, C, dact_bf):
    """favae_bn_train_fwd_bins_ab publishes (a | b) of the forward pass; favae_bn_lrelu_bwd_bins_ab takes the LeakyReLU slope from the
    sign of raw * a + b instead of reading the stored activation (final_layer.0 in the batch-64 step).  The published pair is
    invstd * gamma | beta - mean * a, the activation is lrelu(raw * a + b), and the backward pass gives bit for bit what
    favae_bn_lrelu_bwd_bins gives from the stored activation (the same expression decides the slope in both passes)."""
    import ctypes
    if dact_bf and dt != torch.bfloat16:
        pytest.skip("bf16 activation gradients belong to the bf16 executor")
    lib = _lib()
    rawl = lib.raw()
    i_, ll_, f_, sz_ = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_size_t
    g = torch.Generator().manual_seed(R + C)
    raw = (torch.randn(R, C, generator=g) * 2 + 0.3)
    raw[::7] *= 1e-4                                     # plenty of values near the kink
    raw = raw.cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.01).cuda()
    bins = torch.zeros(8, 2, C, dtype=torch.float64, device="cuda")
    bins[3, 0] = raw.double().sum(0)
    bins[5, 1] = (raw.double() ** 2).sum(0)
    act = torch.empty(R, C, device="cuda", dtype=dt)
    mean, invstd, ab = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(2 * C, device="cuda")
    code = lib.dtype_code(act)
    lib.check(rawl.favae_bn_train_fwd_bins_ab(lib.ptr(raw), lib.ptr(bins), i_(8), lib.ptr(gamma), lib.ptr(beta), lib.ptr(act), lib.ptr(mean),
                                              lib.ptr(invstd), lib.ptr(ab), lib.ptr(None), lib.ptr(None), ll_(R), i_(C), f_(1e-5), f_(0.1), i_(code),
                                              lib.stream_ptr()))
    a_ref = invstd * gamma
    assert_close(ab[:C].cpu(), a_ref.cpu(), 1e-7)
    assert_close(ab[C:].cpu(), (beta - mean * a_ref).cpu(), 1e-6)
    y = F.leaky_relu(raw * ab[:C] + ab[C:], 0.01)
    assert_close(act.float().cpu(), y.cpu(), 1e-6 if dt == torch.float32 else 8e-3)
    dact = torch.randn(R, C, generator=g).cuda()
    if dact_bf:
        dact = dact.to(torch.bfloat16)
    gb = torch.zeros(4, 2, C, dtype=torch.float64, device="cuda")
    dd = dact.double() * torch.where(act.double() > 0, 1.0, 0.01)
    xh = (raw.double() - mean.double()) * invstd.double()
    gb[1, 0], gb[2, 1] = dd.sum(0), (dd * xh).sum(0)
    scratch = torch.zeros(2 * 1024 * 64 + 4096, dtype=torch.float64, device="cuda")
    flags = code | (lib.DACT_BF16 if dact_bf else 0)
    out = []
    for with_ab in (False, True):
        draw = torch.empty(R, C, device="cuda", dtype=dt)
        dg, db, dbias = (torch.zeros(C, device="cuda") for _ in range(3))
        if with_ab:
            lib.check(rawl.favae_bn_lrelu_bwd_bins_ab(lib.ptr(dact), lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma), lib.ptr(ab),
                                                      lib.ptr(gb), i_(4), lib.ptr(draw), lib.ptr(dg), lib.ptr(db), lib.ptr(dbias), ll_(R), i_(C),
                                                      lib.ptr(scratch), sz_(scratch.numel()), i_(flags), lib.stream_ptr()))
        else:
            lib.check(rawl.favae_bn_lrelu_bwd_bins(lib.ptr(dact), lib.ptr(act), lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma),
                                                   lib.ptr(gb), i_(4), lib.ptr(draw), lib.ptr(dg), lib.ptr(db), lib.ptr(dbias), ll_(R), i_(C),
                                                   lib.ptr(scratch), sz_(scratch.numel()), i_(flags), lib.stream_ptr()))
        torch.cuda.synchronize()
        out.append((draw.cpu(), dg.cpu(), db.cpu(), dbias.cpu()))
    for u, w in zip(out[0], out[1]):
        assert torch.equal(u, w)
    ref = (gamma.double() * invstd.double()) * (dd - dd.sum(0) / R - xh * (dd * xh).sum(0) / R)
    assert_close(out[1][0].float(), ref.float().cpu(), 1e-5 if dt == torch.float32 else 8e-3)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("R,C", [(2048, 64), (16384, 128), (5000, 32), (262144, 64)])
def test_bias_gradient_finished_by_a_second_launch(dt, tol, R, C):
    """FAVAE_DBIAS_DEFER: favae_bn_lrelu_bwd_bins leaves the column sums of draw per slab of rows in the caller's buffer and
    favae_bn_dbias_finish adds them up (what the training step does beside its main chain).  With the two BatchNorm sums set to
    zero the column sum is a real number, gamma * invstd * sum(dact * slope): the two-launch form equals the one-launch form to
    the last bit (same slabs, same order) and torch to fp32 accuracy; draw is the same."""
    import ctypes
    lib = _lib()
    g = torch.Generator().manual_seed(R + C)
    raw = torch.randn(R, C, generator=g).cuda()
    dact = torch.randn(R, C, generator=g).cuda()
    act = F.leaky_relu(torch.randn(R, C, generator=g), 0.01).to(dt).cuda()
    gamma, mean, invstd = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    bins = torch.zeros(8, 2, C, dtype=torch.float64, device="cuda")
    i_, ll_, sz_ = ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t
    rawl = lib.raw()
    nslab = rawl.favae_bn_lrelu_bwd_slabs(ll_(R), i_(C))
    assert 1 <= nslab <= 256
    res = []
    for flags in (0, 0x80):      # FAVAE_DBIAS_DEFER
        scratch = torch.full((256 * C + 4096,), float("nan"), dtype=torch.float64, device="cuda")
        draw = torch.empty(R, C, device="cuda", dtype=dt)
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dbias = torch.full((C,), 7.0, device="cuda")
        lib.check(rawl.favae_bn_lrelu_bwd_bins(lib.ptr(dact), lib.ptr(act), lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma),
                                               lib.ptr(bins), i_(8), lib.ptr(draw), lib.ptr(dg), lib.ptr(db), lib.ptr(dbias), ll_(R), i_(C),
                                               lib.ptr(scratch), sz_(scratch.numel()), i_(lib.dtype_code(act) | flags), lib.stream_ptr()))
        if flags:
            assert torch.equal(dbias, torch.full((C,), 7.0, device="cuda"))     # untouched by the first launch
            lib.check(rawl.favae_bn_dbias_finish(lib.ptr(scratch), i_(nslab), i_(C), lib.ptr(dbias), lib.stream_ptr()))
        res.append((draw.float().cpu(), dbias.cpu()))
    want = (gamma * invstd).double() * (dact.double() * torch.where(act.double() > 0, 1.0, 0.01)).sum(0)
    assert torch.equal(res[1][0], res[0][0])
    assert_close(res[0][1], want.float().cpu(), 1e-5)
    assert_close(res[1][1], res[0][1], 1e-6)


@pytest.mark.parametrize("nslab,C", [(128, 64), (7, 128), (256, 256), (33, 36)])
def test_slab_sums_ride_with_the_next_weight_gradient(nslab, C):
    """favae_bn_dbias_ride: the calling thread's next favae_conv_wgrad launch adds up [nslab][C] slab sums in its first blocks
    (what favae_bn_dbias_finish does with a launch of its own); the weight gradient itself is unchanged, and the launch after
    that carries nothing."""
    import ctypes
    lib = _lib()
    g = torch.Generator().manual_seed(nslab + C)
    partial = torch.randn(nslab, C, generator=g, dtype=torch.float64).cuda()
    out = torch.full((C,), -5.0, device="cuda")
    dt = torch.bfloat16
    Nb, Hs, Hl, Rc, Cc = 8, 8, 16, 128, 64
    rows = torch.randn(Nb, Hs, Hs, Rc, generator=g).to(dt).cuda()
    cols = torch.randn(Nb, Hl, Hl, Cc, generator=g).to(dt).cuda()
    dw0 = torch.zeros(Rc, 9, Cc, device="cuda")
    lib.conv_wgrad(rows, 0, cols, 1, dw0, Nb, Hs, Hs, Hl, Hl, 3, 2, 1, -1)
    lib.check(lib.raw().favae_bn_dbias_ride(lib.ptr(partial), ctypes.c_int(nslab), ctypes.c_int(C), lib.ptr(out)))
    dw1 = torch.zeros(Rc, 9, Cc, device="cuda")
    lib.conv_wgrad(rows, 0, cols, 1, dw1, Nb, Hs, Hs, Hl, Hl, 3, 2, 1, -1)
    assert_close(dw1.cpu(), dw0.cpu(), 1e-6)      # (split reductions add with atomics: order)
    assert_close(out.cpu(), partial.sum(0).float().cpu(), 1e-6)
    out.fill_(-5.0)
    dw2 = torch.zeros(Rc, 9, Cc, device="cuda")
    lib.conv_wgrad(rows, 0, cols, 1, dw2, Nb, Hs, Hs, Hl, Hl, 3, 2, 1, -1)
    assert_close(dw2.cpu(), dw0.cpu(), 1e-6)
    assert torch.equal(out, torch.full((C,), -5.0, device="cuda"))


@pytest.mark.parametrize("dt,tol", DT)
def test_thin_input_gradient_takes_batchnorm_backward_sums(dt, tol):
    """Input gradient of final_layer.3 (3 -> 64 channels, one thread per pixel) with the BN-backward sums of the layer
    before it taken while the result is stored (favae_conv_igemm_red mode 2 on the thin-input path)."""
    lib = _lib()
    g = torch.Generator().manual_seed(77)
    Nb, H, C = 40, 64, 64          # >= 131072 pixels: the thin-input kernel
    dy = q(torch.randn(Nb, H, H, 3, generator=g), dt)
    wt = q(torch.randn(C, 9, 3, generator=g) * 0.1, dt)
    R = Nb * H * H
    raw = (torch.randn(R, C, generator=g) * 2 + 0.3).cuda()
    act = torch.nn.functional.leaky_relu(torch.randn(R, C, generator=g), 0.01).to(dt).cuda()
    mean, invstd = torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    dyd, wtd = dy.to(dt).cuda(), wt.to(dt).cuda()
    out0 = torch.empty(Nb, H, H, C, device="cuda")
    lib.conv_igemm(dyd, wtd, None, out0, 3, 1, 1, 1, 0)
    bins = torch.zeros(32, 2, C, dtype=torch.float64, device="cuda")
    out1 = torch.empty_like(out0)
    assert lib.conv_igemm_red(dyd, wtd, None, out1, 3, 1, 1, 1, bins, 2, act=act, raw=raw, mean=mean, invstd=invstd)
    assert torch.equal(out1, out0)
    dd = out0.view(R, C).double() * torch.where(act.double() > 0, 1.0, 0.01)
    xh = ((raw - mean) * invstd).double()      # fp32 arithmetic as in the kernel
    tot = bins.sum(0)
    assert_close(tot[0].cpu(), dd.sum(0).cpu(), 1e-7)
    assert_close(tot[1].cpu(), (dd * xh).sum(0).cpu(), 1e-6)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,H,Cs", [(3, 64, 3), (5, 20, 1), (64, 64, 3)])
def test_folded_weight_gradient_encoder0(dt, tol, Nb, H, Cs):
    """encoder.0 (Conv2d Cs -> 64, k3 s2 p1): favae_conv_wgrad_folded against autograd."""
    lib = _lib()
    g = torch.Generator().manual_seed(51 + H)
    Cout = 64
    x = q(torch.rand(Nb, Cs, H, H, generator=g), dt)
    w = torch.randn(Cout, Cs, 3, 3, generator=g, requires_grad=True)
    dy = q(torch.randn(Nb, Cout, H // 2, H // 2, generator=g), dt)
    F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 9, Cs)
    dw = torch.zeros(Cout, 9, Cs, device="cuda")
    lib.conv_wgrad_folded(nhwc(dy).to(dt).cuda(), nhwc(x).to(dt).cuda(), dw, Nb, H // 2, H // 2, H, H, 2)
    assert_close(dw.cpu(), ref, tol)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("split", [0, -1, -7, 5])
def test_fold_then_gemm_in_two_calls(dt, tol, split):
    """favae_conv_fold_taps + favae_conv_wgrad_prefolded (what the training step does for encoder.0: the gather of the
    batch runs at the start of the step, the GEMM at its end) against autograd; every split-K convention."""
    lib = _lib()
    g = torch.Generator().manual_seed(57)
    Nb, H, Cs, Cout = 5, 32, 3, 64
    x = q(torch.rand(Nb, Cs, H, H, generator=g), dt)
    w = torch.randn(Cout, Cs, 3, 3, generator=g, requires_grad=True)
    dy = q(torch.randn(Nb, Cout, H // 2, H // 2, generator=g), dt)
    F.conv2d(x, w, None, stride=2, padding=1).backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 27)
    xd = nhwc(x).to(dt).cuda()
    folded = lib.conv_fold_taps(xd, Nb, H // 2, H // 2, H, H, 2)
    # the folded matrix itself: column tap*Cs+c of pixel (n, h, w) is x[n, 2h-1+kh, 2w-1+kw, c] (zero outside, zero padding columns)
    want = F.unfold(x, 3, padding=1, stride=2).reshape(Nb, Cs, 9, -1).permute(0, 3, 2, 1).reshape(-1, 27)
    assert torch.equal(folded[:, :27].float().cpu(), want) and not folded[:, 27:].any()
    dw = torch.zeros(Cout, 27, device="cuda")
    lib.conv_wgrad_prefolded(nhwc(dy).to(dt).cuda().reshape(-1, Cout), folded, dw, Cout, 27, split)
    assert_close(dw.cpu(), ref, tol)


@pytest.mark.parametrize("Nb,H,Cs", [(64, 64, 3), (5, 32, 3), (3, 18, 1), (7, 10, 2)])
def test_forward_convolution_over_the_folded_batch(Nb, H, Cs):
    """favae_conv_folded_fwd: encoder.0 (Conv2d Cs -> 64, k3 s2 p1, celeba_deepvanilla_vae.py:31-33) as one K <= 32 GEMM over the
    matrix favae_conv_fold_taps builds, bf16 operands: against F.conv2d on the rounded operands, with the per-channel sums of
    the BatchNorm statistics (pixel counts that are not multiples of 32 included)."""
    lib = _lib()
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(71 + H)
    Cout = 64
    x = q(torch.rand(Nb, Cs, H, H, generator=g), dt)
    w = q(torch.randn(Cout, Cs, 3, 3, generator=g) * 0.3, dt)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    folded = lib.conv_fold_taps(nhwc(x).to(dt).cuda(), Nb, H // 2, H // 2, H, H, 2)
    wd = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cs).contiguous().to(dt).cuda()
    out = torch.full((ref.shape[0], Cout), float("nan"), device="cuda")
    bins = torch.zeros(8, 2, Cout, dtype=torch.float64, device="cuda")
    lib.conv_folded_fwd(folded, wd, b.cuda(), out, bins)
    assert_close(out.cpu().double(), ref, 1e-5)                  # fp32 accumulation of exactly representable products
    tot = bins.sum(0).cpu()
    assert_close(tot[0], out.double().sum(0).cpu(), 1e-10)
    assert_close(tot[1], (out.double() ** 2).sum(0).cpu(), 1e-10)
    out2 = torch.empty_like(out)
    lib.conv_folded_fwd(folded, wd, None, out2)                  # no bias, no sums
    assert_close(out2.cpu().double(), ref - b.double(), 1e-5)


@pytest.mark.parametrize("Nb,H,Cs", [(64, 64, 3), (5, 32, 3), (3, 18, 1), (7, 10, 2)])
def test_forward_convolution_over_the_folded_batch_in_two_bf16_terms(Nb, H, Cs):
    """favae_conv_fold_taps_split + favae_conv_folded_fwd_split: encoder.0 of the bf16 step with fp32 image and fp32 weights, each
    carried as two bf16 terms on the bf16 matrix cores (x_hi w_hi + x_lo w_hi + x_hi w_lo): against F.conv2d in float64 on the
    UNROUNDED operands at 2e-5 of scale (the plain bf16 form is 4e-3 away on these inputs); the hi plane is bit-identical to
    favae_conv_fold_taps of the bf16 batch (the weight gradient reads it)."""
    lib = _lib()
    g = torch.Generator().manual_seed(71 + H)
    Cout = 64
    x = torch.rand(Nb, Cs, H, H, generator=g)
    w = torch.randn(Cout, Cs, 3, 3, generator=g) * 0.3
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    xd = nhwc(x).cuda()
    hl = lib.conv_fold_taps_split(xd, Nb, H // 2, H // 2, H, H, 2)
    plain = lib.conv_fold_taps(xd.to(torch.bfloat16), Nb, H // 2, H // 2, H, H, 2)
    assert torch.equal(hl[0], plain)
    want = F.unfold(x, 3, padding=1, stride=2).reshape(Nb, Cs, 9, -1).permute(0, 3, 2, 1).reshape(-1, 9 * Cs)
    both = hl[0].float() + hl[1].float()
    assert (both[:, :9 * Cs].cpu() - want).abs().max() <= 2.0 ** -16 and not both[:, 9 * Cs:].any()
    wd = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cs).contiguous().cuda()
    out = torch.full((ref.shape[0], Cout), float("nan"), device="cuda")
    bins = torch.zeros(8, 2, Cout, dtype=torch.float64, device="cuda")
    lib.conv_folded_fwd_split(hl, wd, b.cuda(), out, bins)
    assert_close(out.cpu().double(), ref, 2e-5)
    tot = bins.sum(0).cpu()
    assert_close(tot[0], out.double().sum(0).cpu(), 1e-10)
    assert_close(tot[1], (out.double() ** 2).sum(0).cpu(), 1e-10)
    rounded = torch.empty_like(out)
    lib.conv_folded_fwd(hl[0], wd.to(torch.bfloat16), b.cuda(), rounded)
    assert (rounded.cpu().double() - ref).abs().max() > 20 * (out.cpu().double() - ref).abs().max()


@pytest.mark.parametrize("out_dt,raw_dt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)])
@pytest.mark.parametrize("Nb,H,Co,groups", [(2, 64, 3, 1), (3, 20, 2, 1), (4, 32, 3, 2), (2, 16, 1, 2)])
def test_input_gradient_of_the_final_conv_over_the_folded_gradient(Nb, H, Co, groups, out_dt, raw_dt):
    """favae_conv_fold_taps(shifted) + favae_conv_folded_dgrad: the input gradient of final_layer.3 (Conv2d 64 -> Co, k3 s1 p1) as a
    K = 32 GEMM on the bf16 matrix cores over the folded output gradient, against autograd on the bf16-rounded operands (1e-4 of
    scale; bf16 output: one more rounding); the BatchNorm-backward sums of the layer in front per group of images against the same
    sums formed from the fp32 result in float64."""
    lib = _lib()
    g = torch.Generator().manual_seed(97 + H + Co)
    Cin = 64
    x = torch.zeros(Nb, Cin, H, H, requires_grad=True)
    w = q(torch.randn(Co, Cin, 3, 3, generator=g) * 0.2, torch.bfloat16)
    dy = q(torch.randn(Nb, Co, H, H, generator=g), torch.bfloat16)
    F.conv2d(x, w, None, stride=1, padding=1).backward(dy)
    ref = nhwc(x.grad).reshape(-1, Cin).double()
    fold = lib.conv_fold_taps(nhwc(dy).to(torch.bfloat16).cuda(), Nb, H, H, H, H, 1, shifted=1)
    wd = w.permute(0, 2, 3, 1).reshape(Co, 9, Cin).contiguous().to(torch.bfloat16).cuda()     # the layer's weight [Cout][tap][Cin]
    P = Nb * H * H
    act = torch.randn(P, Cin, generator=g).to(torch.bfloat16)
    raw = q(torch.randn(P, Cin, generator=g), raw_dt)
    mean = torch.randn(groups, Cin, generator=g) * 0.1
    invstd = torch.rand(groups, Cin, generator=g) + 0.5
    bins = torch.zeros(groups, 8, 2, Cin, dtype=torch.float64, device="cuda")
    out = torch.full((P, Cin), float("nan"), device="cuda").to(out_dt)
    lib.conv_folded_dgrad(fold, wd, out, bins, act.cuda(), raw.to(raw_dt).cuda(), mean.cuda(), invstd.cuda(), groups)
    assert_close(out.float().cpu().double(), ref, 1e-4 if out_dt == torch.float32 else 4e-3)
    plain = torch.full((P, Cin), float("nan"), device="cuda").to(out_dt)
    lib.conv_folded_dgrad(fold, wd, plain)
    assert torch.equal(plain, out)
    dd = ref * torch.where(act.double() > 0, 1.0, 0.01)
    Pg = P // groups
    mu = mean.double().repeat_interleave(Pg, 0)
    isd = invstd.double().repeat_interleave(Pg, 0)
    xh = (raw.double() - mu) * isd
    tot = bins.sum(1).cpu()
    assert_close(tot[:, 0], dd.reshape(groups, Pg, Cin).sum(1), 1e-4)
    assert_close(tot[:, 1], (dd * xh).reshape(groups, Pg, Cin).sum(1), 1e-4)
    # the slope of the LeakyReLU from the forward pass's affine form (raw * a + b) instead of the stored activation
    ab = torch.stack([torch.rand(groups, Cin, generator=g) + 0.5, torch.randn(groups, Cin, generator=g) * 0.3], 1).contiguous()     # [groups][2][Cin]
    pre = raw.to(raw_dt).float() * ab[:, 0].repeat_interleave(Pg, 0) + ab[:, 1].repeat_interleave(Pg, 0)
    dd2 = ref * torch.where(pre.double() > 0, 1.0, 0.01)
    bins.zero_()
    lib.conv_folded_dgrad(fold, wd, out, bins, None, raw.to(raw_dt).cuda(), mean.cuda(), invstd.cuda(), groups, ab=ab.cuda())
    tot = bins.sum(1).cpu()
    assert_close(tot[:, 0], dd2.reshape(groups, Pg, Cin).sum(1), 1e-4)
    assert_close(tot[:, 1], (dd2 * xh).reshape(groups, Pg, Cin).sum(1), 1e-4)


@pytest.mark.parametrize("grad_dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("Nb,H,groups", [(2, 64, 1), (4, 32, 2), (6, 16, 3)])
def test_input_gradient_of_the_final_conv_with_the_batchnorm_backward_applied(Nb, H, groups, grad_dt):
    """favae_conv_folded_dgrad(out = NULL) + favae_conv_folded_dgrad_bn: the gradient w.r.t. final_layer.0's activation is never stored
    (celeba_deepvanilla_vae.py:72-76 under loss.backward() :249): the first call leaves the two BatchNorm-backward sums, the second forms the
    gradient again and applies the BatchNorm backward.  Against the stored form (favae_conv_folded_dgrad + favae_gbn_bwd): the same sums, the
    same draw up to the odd last bf16 bit, the same dgamma / dbeta, and per-block column sums of draw that add up to its column sums."""
    import ctypes
    lib = _lib()
    rawl = lib.raw()
    i_, ll_ = ctypes.c_int, ctypes.c_longlong
    g = torch.Generator().manual_seed(700 + Nb + H)
    Cin, Co = 64, 3
    P = Nb * H * H
    Pg = P // groups
    dy = (torch.randn(Nb, H, H, Co, generator=g) * 0.1).to(torch.bfloat16).cuda()
    fold = lib.conv_fold_taps(dy, Nb, H, H, H, H, 1, shifted=1)
    wd = (torch.randn(Co, 9, Cin, generator=g) * 0.05).to(torch.bfloat16).cuda()
    raw = torch.randn(P, Cin, generator=g).to(torch.bfloat16).cuda()
    mean = (torch.randn(groups, Cin, generator=g) * 0.1).cuda()
    invstd = (torch.rand(groups, Cin, generator=g) + 0.5).cuda()
    gamma = (torch.rand(Cin, generator=g) + 0.5).cuda()
    ab = torch.stack([torch.rand(groups, Cin, generator=g) + 0.5, torch.randn(groups, Cin, generator=g) * 0.3], 1).contiguous().cuda()
    nb = 8
    # stored form
    dact = torch.empty(P, Cin, device="cuda", dtype=grad_dt)
    bins1 = torch.zeros(groups, nb, 2, Cin, dtype=torch.float64, device="cuda")
    lib.conv_folded_dgrad(fold, wd, dact, bins1, None, raw, mean, invstd, groups, ab=ab)
    draw1 = torch.empty(P, Cin, device="cuda", dtype=torch.bfloat16)
    dg1, db1 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    flags = lib.BF16 | lib.RAW_BF16 | (lib.DACT_BF16 if grad_dt == torch.bfloat16 else 0)
    lib.check(rawl.favae_gbn_bwd(lib.ptr(dact), None, lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma), lib.ptr(ab), lib.ptr(bins1), i_(nb),
                                 lib.ptr(draw1), lib.ptr(dg1), lib.ptr(db1), None, i_(groups), ll_(Pg), i_(Cin), i_(flags), lib.stream_ptr()))
    # recomputed form
    bins2 = torch.zeros_like(bins1)
    oflag = lib.OUT_BF16 if grad_dt == torch.bfloat16 else 0
    lib.check(rawl.favae_conv_folded_dgrad(lib.ptr(fold), lib.ptr(wd), None, ll_(P), i_(27), i_(64), lib.ptr(bins2), i_(nb), None, lib.ptr(raw),
                                           lib.ptr(mean), lib.ptr(invstd), lib.ptr(ab), i_(groups), i_(lib.BF16 | lib.RAW_BF16 | oflag), lib.stream_ptr()))
    # (the blocks' sums arrive in the bins through fp64 atomics, a bin may take several blocks: equal to rounding of the fp64 sums)
    assert torch.allclose(bins2.sum(1), bins1.sum(1), rtol=1e-12, atol=1e-9)
    draw2 = torch.full((P, Cin), 9.0, device="cuda", dtype=torch.bfloat16)
    dg2, db2 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    cap = 256
    part = torch.full((cap, Cin), float("nan"), dtype=torch.float64, device="cuda")
    slabs = ctypes.c_int(0)
    lib.check(rawl.favae_conv_folded_dgrad_bn(lib.ptr(fold), lib.ptr(wd), lib.ptr(draw2), ll_(P), i_(27), i_(64), lib.ptr(bins1), i_(nb), None,
                                              lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma), lib.ptr(ab), i_(groups), lib.ptr(dg2),
                                              lib.ptr(db2), lib.ptr(part), i_(cap), ctypes.byref(slabs), i_(lib.BF16 | lib.RAW_BF16 | oflag),
                                              lib.stream_ptr()))
    torch.cuda.synchronize()

    def same_draw(a, b):
        # the same expressions in two kernels: the compiler's choice of fused multiply-adds may differ, which flips the odd last bf16 bit
        a, b = a.float(), b.float()
        diff = (a - b).abs()
        assert (diff <= 2.0 ** -7 * b.abs() + 1e-30).all(), "more than one bf16 ulp apart"
        assert (diff > 0).float().mean().item() < 0.01, "more than 1 % of the elements differ"

    same_draw(draw2, draw1)
    assert_close(dg2.cpu(), dg1.cpu(), 1e-6); assert_close(db2.cpu(), db1.cpu(), 1e-6)      # (one fp32 atomic per group: their order is not fixed)
    assert 1 <= slabs.value <= cap
    # the per-block column sums are sums of the unrounded draw; the stored draw is rounded to bf16: agreement to the rounding of P / groups terms
    col = part[:slabs.value].sum(0).cpu()
    ref = draw1.double().sum(0).cpu()
    scale = draw1.double().abs().sum(0).cpu().max().item()
    assert (col - ref).abs().max().item() <= 2.0 ** -8 * scale / np.sqrt(P) * 8 + 1e-9
    # without the partial sums: the same draw
    draw3 = torch.empty_like(draw2)
    dg3, db3 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    lib.check(rawl.favae_conv_folded_dgrad_bn(lib.ptr(fold), lib.ptr(wd), lib.ptr(draw3), ll_(P), i_(27), i_(64), lib.ptr(bins1), i_(nb), None,
                                              lib.ptr(raw), lib.ptr(mean), lib.ptr(invstd), lib.ptr(gamma), lib.ptr(ab), i_(groups), lib.ptr(dg3),
                                              lib.ptr(db3), None, i_(0), None, i_(lib.BF16 | lib.RAW_BF16 | oflag), lib.stream_ptr()))
    assert torch.equal(draw3, draw2)
    assert_close(dg3.cpu(), dg1.cpu(), 1e-6)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,H,Co", [(2, 64, 3), (3, 18, 2), (64, 64, 3)])
def test_folded_weight_gradient_final_conv(dt, tol, Nb, H, Co):
    """final_layer.3 (Conv2d 64 -> Co, k3 s1 p1): the narrow operand is the output gradient, shifted per tap."""
    lib = _lib()
    g = torch.Generator().manual_seed(61 + H)
    Cin = 64
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = torch.randn(Co, Cin, 3, 3, generator=g, requires_grad=True)
    dy = q(torch.randn(Nb, Co, H, H, generator=g), dt)
    F.conv2d(x, w, None, stride=1, padding=1).backward(dy)
    dw = torch.zeros(Co, 9, Cin, device="cuda")
    lib.conv_wgrad_folded(nhwc(dy).to(dt).cuda(), nhwc(x).to(dt).cuda(), dw, Nb, H, H, H, H, 1)
    assert_close(dw.cpu(), w.grad.permute(0, 2, 3, 1).reshape(Co, 9, Cin), tol)


@pytest.mark.parametrize("dt,tol", DT)
@pytest.mark.parametrize("Nb,H,Co", [(3, 64, 3), (5, 10, 1), (64, 64, 3)])
def test_final_conv_with_loss_in_one_launch(dt, tol, Nb, H, Co):
    """favae_conv_out_ll: Conv2d(64 -> Co, k3 s1 p1) + Sigmoid + gaussian_LL + gradient + bias gradient
    (celeba_deepvanilla_vae.py:74-77, 152-171, 238) against torch autograd of the same expression."""
    import ctypes
    lib = _lib()
    rawl = lib.raw()
    g = torch.Generator().manual_seed(71 + H)
    Cin, var_x = 64, 0.01
    a = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = q(torch.randn(Co, Cin, 3, 3, generator=g) * 0.05, dt)
    b = (torch.randn(Co, generator=g) * 0.1).requires_grad_(True)
    x = torch.rand(Nb, Co, H, H, generator=g)
    y = F.conv2d(a.double(), w.double(), b.double(), stride=1, padding=1)
    y.retain_grad()
    sg = torch.sigmoid(y)
    D = Co * H * H
    ll = (-0.5 * (D * np.log(2 * np.pi) + D * np.log(var_x) + ((x.double() - sg) ** 2 / var_x).flatten(1).sum(-1))).mean()
    (-ll).backward()
    i_, f_, sz_ = ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    wd = w.permute(0, 2, 3, 1).reshape(Co, 9, Cin).contiguous().to(dt).cuda()
    wk = torch.zeros(9, 16, 4, 4, device="cuda", dtype=dt)
    lib.check(rawl.favae_conv_out_wprep(lib.ptr(wd), lib.ptr(wk), i_(Co), i_(lib.dtype_code(wd)), lib.stream_ptr()))
    want = torch.zeros(9, 16, 4, 4)
    want[:, :, :Co, :] = w.permute(2, 3, 0, 1).reshape(9, Co, 16, 4).permute(0, 2, 1, 3)
    assert torch.equal(wk.float().cpu(), want)
    ad, xd, bd = nhwc(a).to(dt).cuda(), nhwc(x).cuda(), b.detach().cuda()
    raw = torch.zeros(Nb, H, H, Co, device="cuda")
    draw = torch.zeros(Nb, H, H, Co, device="cuda", dtype=dt)
    rec, dbias = torch.zeros(1, dtype=torch.float64, device="cuda"), torch.zeros(Co, device="cuda")
    scratch = torch.zeros(5 * ((Nb * H * H + 255) // 256) + 8, dtype=torch.float64, device="cuda")
    for _ in range(2):   # twice: the ticket must be left clean
        lib.check(rawl.favae_conv_out_ll(lib.ptr(ad), lib.ptr(wk), lib.ptr(bd), lib.ptr(xd), i_(Nb), i_(H), i_(H), i_(Co), f_(var_x),
                                         lib.ptr(raw), lib.ptr(draw), lib.ptr(rec), lib.ptr(dbias), lib.ptr(scratch), sz_(scratch.numel()),
                                         i_(lib.dtype_code(ad)), lib.stream_ptr()))
        assert_close(raw.cpu(), nhwc(y.detach()), tol)
        assert abs(rec.item() + ll.item()) <= 1e-5 * abs(ll.item())
        assert_close(draw.float().cpu(), nhwc(y.grad), 1e-4 if dt == torch.float32 else 8e-3)
        assert_close(dbias.cpu(), b.grad, 1e-4)
    if dt == torch.bfloat16 and H % 64 == 0:
        # favae_conv_out_raw (the eval decoder's final_layer.3): the same convolution without the likelihood, bit for bit
        raw2 = torch.full((Nb, H, H, Co), 5.0, device="cuda")
        lib.check(rawl.favae_conv_out_raw(lib.ptr(ad), lib.ptr(wk), lib.ptr(bd), i_(Nb), i_(H), i_(H), i_(Co), lib.ptr(raw2), i_(lib.BF16),
                                          lib.stream_ptr()))
        assert torch.equal(raw2, raw)
    # FAVAE_DBIAS_DEFER: the launch leaves five sums per block; a launch of its own (finish) or the first block of the calling
    # thread's next weight-gradient launch (ride) adds them up: the same loss term and bias gradient
    rec0, dbias0 = rec.clone(), dbias.clone()
    for how in ("finish", "ride"):
        rec.fill_(-1.0); dbias.fill_(-1.0); scratch.fill_(float("nan"))
        lib.check(rawl.favae_conv_out_ll(lib.ptr(ad), lib.ptr(wk), lib.ptr(bd), lib.ptr(xd), i_(Nb), i_(H), i_(H), i_(Co), f_(var_x),
                                         lib.ptr(raw), lib.ptr(draw), lib.ptr(rec), lib.ptr(dbias), lib.ptr(scratch), sz_(scratch.numel()),
                                         i_(lib.dtype_code(ad) | 0x80), lib.stream_ptr()))
        assert rec.item() == -1.0 and (dbias == -1.0).all()
        if how == "finish":
            lib.check(rawl.favae_conv_out_ll_finish(lib.ptr(scratch), i_(Nb), i_(H), i_(H), i_(Co), f_(var_x), lib.ptr(rec), lib.ptr(dbias),
                                                    lib.stream_ptr()))
        else:
            lib.check(rawl.favae_conv_out_ll_ride(lib.ptr(scratch), i_(Nb), i_(H), i_(H), i_(Co), f_(var_x), lib.ptr(rec), lib.ptr(dbias)))
            rows = torch.randn(4, 8, 8, 128).to(torch.bfloat16).cuda()
            cols = torch.randn(4, 16, 16, 64).to(torch.bfloat16).cuda()
            lib.conv_wgrad(rows, 0, cols, 1, torch.zeros(128, 9, 64, device="cuda"), 4, 8, 8, 16, 16, 3, 2, 1, -1)
        assert abs(rec.item() - rec0.item()) <= 1e-12 * abs(rec0.item())
        assert_close(dbias.cpu(), dbias0.cpu(), 1e-6)


@pytest.mark.parametrize("Nb,group_images", [(3, 0), (4, 2), (6, 3)])
def test_final_conv_and_its_weight_gradient_without_the_stored_activation(Nb, group_images):
    """favae_conv_out_ll_bn / favae_conv_wgrad_prefolded_bn: final_layer.3 (celeba_deepvanilla_vae.py:74-76) forward + loss and weight gradient
    reading the conv output in front of final_layer.1/.2 (BatchNorm2d(train) + LeakyReLU, :72-73) and forming the activation at the operand
    load, per BatchNorm group: bit-identical to favae_conv_out_ll / favae_conv_wgrad_prefolded on the activation tensor the BatchNorm
    kernels would have stored, bf16(lrelu(raw * a + b))."""
    import ctypes
    lib = _lib()
    rawl = lib.raw()
    i_, f_, sz_, ll_ = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_longlong
    dt = torch.bfloat16
    H, Co, C, var_x = 64, 3, 64, 0.01
    g = torch.Generator().manual_seed(500 + Nb)
    G = Nb // group_images if group_images else 1
    rawin = (torch.randn(Nb, H, H, C, generator=g) * 2).to(dt)
    ab = torch.stack([torch.rand(G, C, generator=g) + 0.5, torch.randn(G, C, generator=g) * 0.5], 1).contiguous()    # [G][2][C]
    gi = group_images if group_images else Nb
    a_full = ab[:, 0].repeat_interleave(gi, 0).view(Nb, 1, 1, C)
    b_full = ab[:, 1].repeat_interleave(gi, 0).view(Nb, 1, 1, C)
    y = rawin.float() * a_full + b_full
    act_cpu = torch.where(y > 0, y, 0.01 * y).to(dt)
    w = q(torch.randn(Co, C, 3, 3, generator=g) * 0.05, dt)
    b = torch.randn(Co, generator=g) * 0.1
    x = torch.rand(Nb, H, H, Co, generator=g)
    wd = w.permute(0, 2, 3, 1).reshape(Co, 9, C).contiguous().to(dt).cuda()
    wk = torch.zeros(9, 16, 4, 4, device="cuda", dtype=dt)
    lib.check(rawl.favae_conv_out_wprep(lib.ptr(wd), lib.ptr(wk), i_(Co), i_(lib.BF16), lib.stream_ptr()))
    scratch = torch.zeros(5 * ((Nb * H * H + 255) // 256) + 8, dtype=torch.float64, device="cuda")
    rd, abd, xd, bd = rawin.cuda(), ab.cuda(), x.cuda(), b.cuda()

    def run_bn():
        raw = torch.zeros(Nb, H, H, Co, device="cuda"); draw = torch.zeros(Nb, H, H, Co, device="cuda", dtype=dt)
        rec, dbias = torch.zeros(1, dtype=torch.float64, device="cuda"), torch.zeros(Co, device="cuda")
        lib.check(rawl.favae_conv_out_ll_bn(lib.ptr(rd), lib.ptr(abd), i_(group_images), lib.ptr(wk), lib.ptr(bd), lib.ptr(xd), i_(Nb), i_(H), i_(H),
                                            i_(Co), f_(var_x), lib.ptr(raw), lib.ptr(draw), lib.ptr(rec), lib.ptr(dbias), lib.ptr(scratch),
                                            sz_(scratch.numel()), i_(lib.BF16), lib.stream_ptr()))
        return raw, draw, rec, dbias

    def run_plain(act):
        raw = torch.zeros(Nb, H, H, Co, device="cuda"); draw = torch.zeros(Nb, H, H, Co, device="cuda", dtype=dt)
        rec, dbias = torch.zeros(1, dtype=torch.float64, device="cuda"), torch.zeros(Co, device="cuda")
        lib.check(rawl.favae_conv_out_ll(lib.ptr(act), lib.ptr(wk), lib.ptr(bd), lib.ptr(xd), i_(Nb), i_(H), i_(H), i_(Co), f_(var_x),
                                         lib.ptr(raw), lib.ptr(draw), lib.ptr(rec), lib.ptr(dbias), lib.ptr(scratch), sz_(scratch.numel()),
                                         i_(lib.BF16), lib.stream_ptr()))
        return raw, draw, rec, dbias

    # the activation as the library's BatchNorm apply stores it, group by group (fp32 copy of the bf16 conv output: the same values)
    act_dev = torch.empty(Nb, H, H, C, device="cuda", dtype=dt)
    raw32 = rd.float()
    for gidx in range(G):
        lo = gidx * gi
        lib.check(rawl.favae_bn_lrelu_fwd_ab(lib.ptr(raw32[lo:lo + gi]), lib.ptr(abd[gidx]), lib.ptr(act_dev[lo:lo + gi]), ll_(gi * H * H), i_(C),
                                             i_(lib.BF16), lib.stream_ptr()))
    assert (act_dev.cpu() != act_cpu).float().mean().item() < 0.02      # (fma in the kernels against mul + add here: the odd last bit)
    got = run_bn()
    want = run_plain(act_dev)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[3], want[3])
    assert got[2].item() == want[2].item()

    # the weight gradient over the folded output gradient: rows [K][32] (27 live columns), cols = raw / the activation [K][64]
    K = Nb * H * H
    rows = torch.zeros(K, 32, dtype=dt)
    rows[:, :27] = (torch.randn(K, 27, generator=g) * 0.1).to(dt)
    rows_d = rows.cuda()
    dw_bn = torch.zeros(27, 64, device="cuda"); dw_pl = torch.zeros(27, 64, device="cuda")
    lib.check(rawl.favae_conv_wgrad_prefolded_bn(lib.ptr(rows_d), i_(32), i_(27), lib.ptr(rd), i_(C), i_(C), lib.ptr(abd), ll_(group_images * H * H),
                                                 lib.ptr(dw_bn), ll_(K), i_(-1), i_(lib.BF16), lib.stream_ptr()))
    lib.check(rawl.favae_conv_wgrad_prefolded(lib.ptr(rows_d), i_(32), i_(27), lib.ptr(act_dev), i_(C), i_(C), lib.ptr(dw_pl), ll_(K), i_(-1),
                                              i_(lib.BF16), lib.stream_ptr()))
    ref = rows[:, :27].double().t() @ act_dev.cpu().double().view(K, C)
    assert_close(dw_bn.cpu(), ref, 1e-4)
    assert_close(dw_bn.cpu(), dw_pl.cpu(), 1e-6)      # (the slices of the reduction arrive through fp32 atomics: the order of the sums is not fixed)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,Nb,Cin,Cout,H", [("conv", 64, 512, 1024, 4), ("convT", 64, 1024, 512, 4), ("conv", 64, 256, 512, 8), ("conv", 64, 64, 128, 32)])
def test_weight_gradient_with_the_adam_step_in_its_epilogue(dt, kind, Nb, Cin, Cout, H):
    """favae_conv_wgrad_adam: where the weight-gradient GEMM runs unsplit in 64x64 tiles its epilogue takes the optimiser step and
    writes both dtype layouts of the fresh weights: bit-identical to favae_conv_wgrad + favae_adam_step + favae_weight_prep (same
    accumulators, same expressions); where it does not (the last case: 18 tiles x 9 taps split their reduction over 16 384 pixels)
    it is favae_conv_wgrad."""
    import ctypes
    lib = _lib()
    rawl = lib.raw()
    i_, ll_, f_ = ctypes.c_int, ctypes.c_longlong, ctypes.c_float
    g = torch.Generator().manual_seed(5 + Cin + H)
    Hs, Hl = (H // 2, H) if kind == "conv" else (H, H * 2)       # small / large feature map of the stride-2 layer
    if kind == "conv":      # rows = draw [Nb, Hs, Hs, Cout] (small grid), cols = layer input [Nb, Hl, Hl, Cin] (gathered)
        rows = torch.randn(Nb, Hs, Hs, Cout, generator=g).to(dt).cuda()
        cols = torch.randn(Nb, Hl, Hl, Cin, generator=g).to(dt).cuda()
        rl, cl = 0, 1
    else:                   # ConvTranspose: rows = draw on the large grid (gathered), cols = layer input on the small grid
        rows = torch.randn(Nb, Hl, Hl, Cout, generator=g).to(dt).cuda()
        cols = torch.randn(Nb, Hs, Hs, Cin, generator=g).to(dt).cuda()
        rl, cl = 1, 0
    n = Cout * 9 * Cin
    P0 = (torch.randn(n, generator=g) * 0.05).cuda()
    M0, V0 = (torch.randn(n, generator=g) * 1e-3).cuda(), (torch.rand(n, generator=g) * 1e-4).cuda()
    code = lib.dtype_code(rows)
    # reference: three launches
    dw = torch.zeros(n, device="cuda")
    lib.check(rawl.favae_conv_wgrad(lib.ptr(rows), i_(Cout), i_(rl), lib.ptr(cols), i_(Cin), i_(cl), lib.ptr(dw), i_(Nb), i_(Hs), i_(Hs), i_(Hl), i_(Hl),
                                    i_(3), i_(2), i_(1), i_(-1), i_(code), lib.stream_ptr()))
    P1, M1, V1 = P0.clone(), M0.clone(), V0.clone()
    lib.check(rawl.favae_adam_step(lib.ptr(P1), lib.ptr(dw), lib.ptr(M1), lib.ptr(V1), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8), ll_(7), lib.stream_ptr()))
    wc1, wt1 = torch.empty(n, dtype=dt, device="cuda"), torch.empty(n, dtype=dt, device="cuda")
    lib.check(rawl.favae_weight_prep(lib.ptr(P1), lib.ptr(wc1), lib.ptr(wt1), i_(Cout), i_(9), i_(Cin), i_(code), lib.stream_ptr()))
    # one launch
    P2, M2, V2 = P0.clone(), M0.clone(), V0.clone()
    dw2 = torch.zeros(n, device="cuda")
    wc2, wt2 = torch.zeros(n, dtype=dt, device="cuda"), torch.zeros(n, dtype=dt, device="cuda")
    applied = ctypes.c_int(-1)
    lib.check(rawl.favae_conv_wgrad_adam(lib.ptr(rows), i_(Cout), i_(rl), lib.ptr(cols), i_(Cin), i_(cl), lib.ptr(dw2), lib.ptr(P2), lib.ptr(M2), lib.ptr(V2),
                                         lib.ptr(wc2), lib.ptr(wt2), i_(Nb), i_(Hs), i_(Hs), i_(Hl), i_(Hl), i_(3), i_(2), i_(1), i_(-1), f_(1e-3), f_(0.9),
                                         f_(0.999), f_(1e-8), ll_(7), i_(code), ctypes.byref(applied), lib.stream_ptr()))
    torch.cuda.synchronize()
    if H == 32:
        assert applied.value == 0 and torch.equal(P2, P0)
        assert_close(dw2.cpu(), dw.cpu(), 1e-5)           # (split reduction: atomics reorder the sum)
        return
    assert applied.value == 1
    assert not dw2.any()
    assert torch.equal(M2, M1), ((M2 - M1).abs().max().item(), (M2 != M1).sum().item(), n)
    assert torch.equal(V2, V1), ((V2 - V1).abs().max().item(), (V2 != V1).sum().item())
    assert torch.equal(P2, P1), ((P2 - P1).abs().max().item(), (P2 != P1).sum().item())
    assert torch.equal(wc2, wc1) and torch.equal(wt2, wt1)


@pytest.mark.parametrize("mode", ["f32_out", "bf16_out_groups", "affine"])
def test_transposed_64_to_64_convolution_with_resident_tap_matrices(mode):
    """final_layer.0's forward shape (ConvTranspose2d 64 -> 64, 32 -> 64 pixels, cA/celeba_deepvanilla_vae.py:63-70) runs through
    convt64_fwd_kernel (nine tap matrices resident in LDS, units of two input rows).  Against the scatter form of igemm_kc (forced
    through favae_debug_force_tile) the output is bit-identical, plain, with the per-group BatchNorm sums of the grouped-batch mode
    (sums to 1e-12: atomics in another order) and with the eval epilogue (affine + LeakyReLU, bf16 out); and against
    F.conv_transpose2d on the bf16-rounded operands 1e-4.  A ragged block split (Nb * 16 units over 256 blocks) is part of every case."""
    import ctypes
    lib = _lib()
    raw_ = lib.raw()
    i_ = ctypes.c_int
    Nb = {"f32_out": 129, "bf16_out_groups": 130, "affine": 131}[mode]   # (the kernel takes 128 images or more)
    G = {"f32_out": 0, "bf16_out_groups": 65, "affine": 0}[mode]      # images per BatchNorm group (65: groups change inside blocks)
    g = torch.Generator().manual_seed(Nb)
    dt = torch.bfloat16
    x = q(torch.randn(Nb, 64, 32, 32, generator=g), dt)
    w = q(torch.randn(64, 64, 3, 3, generator=g) * 0.1, dt)
    b = torch.randn(64, generator=g)
    ab = torch.cat([torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)]).cuda()
    xd = nhwc(x).to(dt).cuda()
    wd = w.permute(1, 2, 3, 0).reshape(64, 9, 64).contiguous().to(dt).cuda()
    out_dt = torch.float32 if mode == "f32_out" else torch.bfloat16
    res = {}
    for forced in (0, 1):
        lib.check(raw_.favae_debug_force_tile(i_(0 if forced else -1), i_(-1)))
        try:
            out = torch.full((Nb, 64, 64, 64), 7.0, device="cuda").to(out_dt)
            bins = torch.zeros(max(1, (Nb + 64) // 65) * 8 * 2 * 64, dtype=torch.float64, device="cuda")
            fused = ctypes.c_int(0)
            if mode == "f32_out":
                lib.conv_igemm(xd, wd, b.cuda(), out, 3, 2, 1, 1, 1)
            elif mode == "affine":
                lib.check(raw_.favae_conv_igemm_affine(lib.ptr(xd), lib.ptr(wd), lib.ptr(b.cuda()), lib.ptr(out), i_(Nb), i_(32), i_(32), i_(64), i_(64),
                                                       i_(64), i_(64), i_(3), i_(2), i_(1), i_(1), i_(lib.BF16 | lib.OUT_BF16), lib.ptr(ab), lib.stream_ptr()))
            else:
                lib.check(raw_.favae_conv_igemm_red_groups(lib.ptr(xd), lib.ptr(wd), lib.ptr(b.cuda()), lib.ptr(out), i_(Nb), i_(32), i_(32), i_(64),
                                                           i_(64), i_(64), i_(64), i_(3), i_(2), i_(1), i_(1), i_(0), i_(lib.BF16 | lib.OUT_BF16),
                                                           lib.ptr(bins), i_(8), i_(1), None, None, None, None, None, i_(G), ctypes.byref(fused),
                                                           lib.stream_ptr()))
                assert fused.value == 1
            torch.cuda.synchronize()
            res[forced] = (out.cpu(), bins.cpu())
        finally:
            lib.check(raw_.favae_debug_force_tile(i_(-1), i_(-1)))
    assert torch.equal(res[0][0], res[1][0])
    ref = nhwc(F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1))
    if mode == "affine":
        ref = F.leaky_relu(ref * ab[:64].cpu() + ab[64:].cpu(), 0.01)
    assert_close(res[0][0].float(), ref, 1e-4 if mode == "f32_out" else 5e-3)
    if mode == "bf16_out_groups":
        s_new = res[0][1].view(-1, 8, 2, 64).sum(1)
        s_old = res[1][1].view(-1, 8, 2, 64).sum(1)
        assert_close(s_new, s_old, 1e-12)
        for gi in range(2):
            v = ref[gi * G:(gi + 1) * G].double().reshape(-1, 64)
            assert_close(s_new[gi, 0], v.sum(0), 1e-5)
            assert_close(s_new[gi, 1], (v * v).sum(0), 1e-5)

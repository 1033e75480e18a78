"""GPU parity of the HBM-bound VAE kernels (BN+LeakyReLU fwd/bwd, reparam+KL fwd/bwd, sigmoid+LL, Adam, weight
re-layout) against torch-CPU float64 autograd of the same expressions (the ops the reference calls:
condition_a_VAE/celeba_deepvanilla_vae.py:34-35, 86-92, 137-171, 131/250).  fp32 storage, tolerance 1e-5 of scale."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

i_, ll_, f_, sz_ = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_size_t


def _L():
    from favae_hip import lib
    return lib


def _err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("R,C", [(4096, 64), (65536, 64), (16384, 128), (256, 1024), (1000, 512)])
def test_bn_lrelu_forward_backward(R, C):
    L = _L()
    raw_ = L.raw()
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(R, C, generator=g, dtype=torch.float64) * 2 + 0.5).float().double().requires_grad_(True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).float().double().requires_grad_(True)
    beta = torch.randn(C, generator=g, dtype=torch.float64).float().double().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y = F.batch_norm(x.view(R, C, 1, 1), rm, rv, gamma, beta, True, 0.1, 1e-5)
    a = F.leaky_relu(y, 0.01)
    da = torch.randn(R, C, 1, 1, generator=g, dtype=torch.float64).float().double()
    a.backward(da)
    dev = "cuda"
    xd = x.detach().float().to(dev)
    mean, invstd = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rmd, rvd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    scratch = torch.zeros(2 * 1024 * 64 + 4096, dtype=torch.float64, device=dev)
    L.check(raw_.favae_bn_stats(L.ptr(xd), ll_(R), i_(C), f_(1e-5), f_(0.1), L.ptr(mean), L.ptr(invstd), L.ptr(rmd), L.ptr(rvd),
                                L.ptr(scratch), sz_(scratch.numel()), L.stream_ptr()))
    assert _err(rmd, rm) < 1e-5 and _err(rvd, rv) < 1e-5
    act = torch.zeros(R, C, device=dev)
    g32, b32 = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    L.check(raw_.favae_bn_lrelu_fwd(L.ptr(xd), L.ptr(mean), L.ptr(invstd), L.ptr(g32), L.ptr(b32), L.ptr(act), ll_(R), i_(C), i_(1),
                                    i_(0), L.stream_ptr()))
    assert _err(act, a.detach().view(R, C)) < 1e-5
    # the one-call training forward (one launch for the small layers) gives the same numbers, bit for bit
    act2, mean2, invstd2 = torch.zeros(R, C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rm2, rv2 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    L.check(raw_.favae_bn_train_fwd(L.ptr(xd), L.ptr(g32), L.ptr(b32), L.ptr(act2), L.ptr(mean2), L.ptr(invstd2), L.ptr(rm2),
                                    L.ptr(rv2), ll_(R), i_(C), f_(1e-5), f_(0.1), L.ptr(scratch), sz_(scratch.numel()), i_(0),
                                    L.stream_ptr()))
    assert _err(mean2, mean) < 1e-6 and _err(invstd2, invstd) < 1e-6 and _err(rm2, rmd) < 1e-6 and _err(rv2, rvd) < 1e-6
    assert _err(act2, act) < 1e-6
    dad = da.view(R, C).float().to(dev).contiguous()
    draw, dg, db, dbias = (torch.zeros(R, C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                           torch.zeros(C, device=dev))
    s1s2 = torch.zeros(2 * C, device=dev)
    L.check(raw_.favae_bn_lrelu_bwd(L.ptr(dad), L.ptr(act), L.ptr(xd), L.ptr(mean), L.ptr(invstd), L.ptr(g32), L.ptr(draw), L.ptr(dg),
                                    L.ptr(db), L.ptr(dbias), ll_(R), i_(C), L.ptr(scratch), sz_(scratch.numel()), L.ptr(s1s2), i_(0),
                                    L.stream_ptr()))
    assert _err(dg, gamma.grad) < 1e-5
    assert _err(db, beta.grad) < 1e-5
    assert _err(draw, x.grad) < 1e-5
    assert dbias.abs().max().item() < 1e-3 * x.grad.abs().max().item() * np.sqrt(R)


@pytest.mark.parametrize("with_prior", [True, False])
def test_reparam_kl_forward_backward(with_prior):
    L = _L()
    raw_ = L.raw()
    B, z, N = 64, 20, 300
    g = torch.Generator().manual_seed(5)
    fc = (torch.randn(B, 2 * z, generator=g) * 0.7).double().requires_grad_(True)
    eps = torch.randn(B, z, generator=g)
    prior = torch.randn(N, z, generator=g, dtype=torch.float64)
    idx = torch.randint(0, N, (B,), generator=g, dtype=torch.int32)
    tau, beta = 1.7, 0.5
    mu, lv = fc[:, :z], fc[:, z:] + 1e-6
    std = torch.exp(0.5 * lv)
    zl = eps.double() * std + mu
    mp = prior[idx.long()] if with_prior else torch.zeros(B, z, dtype=torch.float64)
    sp = torch.sqrt(torch.ones_like(mp) * float(np.float32(1.0) / np.float32(tau)))
    kl = torch.distributions.kl_divergence(torch.distributions.Normal(mu, std), torch.distributions.Normal(mp, sp)).sum(1).mean()
    dz = torch.randn(B, z, generator=g)
    (beta * kl + (zl * dz.double()).sum()).backward()
    dev = "cuda"
    fcd, epsd = fc.detach().float().to(dev), eps.to(dev)
    pd, idd = (prior.to(dev), idx.to(dev)) if with_prior else (None, None)
    zlat, stdq, klo = torch.zeros(B, z, device=dev), torch.zeros(B, z, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    L.check(raw_.favae_reparam_kl_fwd(L.ptr(fcd), L.ptr(epsd), L.ptr(pd), L.ptr(idd), f_(tau), i_(B), i_(z), L.ptr(zlat), L.ptr(stdq),
                                      L.ptr(klo), i_(0), L.stream_ptr()))
    assert abs(klo.item() - kl.item()) < 1e-6 * abs(kl.item())
    assert _err(zlat, zl.detach()) < 1e-6 and _err(stdq, std.detach()) < 1e-6
    dfc, dzd = torch.zeros(B, 2 * z, device=dev), dz.to(dev)
    L.check(raw_.favae_reparam_kl_bwd(L.ptr(fcd), L.ptr(epsd), L.ptr(pd), L.ptr(idd), L.ptr(stdq), L.ptr(dzd), f_(tau), f_(beta),
                                      i_(B), i_(z), L.ptr(dfc), i_(0), L.stream_ptr()))
    assert _err(dfc, fc.grad) < 1e-5
    # one-block form that also returns the bias gradient of the encoder head (column sums of dfc)
    dfc2, dbias = torch.zeros(B, 2 * z, device=dev), torch.zeros(2 * z, device=dev)
    L.check(raw_.favae_reparam_kl_bwd_bias(L.ptr(fcd), L.ptr(epsd), L.ptr(pd), L.ptr(idd), L.ptr(stdq), L.ptr(dzd), f_(tau), f_(beta),
                                           i_(B), i_(z), L.ptr(dfc2), L.ptr(dbias), i_(0), L.stream_ptr()))
    assert _err(dfc2, dfc) < 1e-6
    assert _err(dbias, fc.grad.sum(0)) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,z,N,with_prior", [(64, 20, 4096, True), (37, 32, 200, False), (512, 20, 4096, True),
                                               (64, 100, 4096, True), (96, 128, 2048, False), (33, 65, 200, True)])
def test_reparam_kl_with_decoder_input_in_one_launch(dt, B, z, N, with_prior):
    """favae_reparam_kl_decoder_input: zlat / stdq / KL identical to favae_reparam_kl_fwd, and
    out = decoder_input(zlat) (nn.Linear z -> N, celeba_deepvanilla_vae.py:41-45) on the stored (dtype-rounded) latent."""
    L = _L()
    raw_ = L.raw()
    dev = "cuda"
    g = torch.Generator().manual_seed(B + z)
    fc = (torch.randn(B, 2 * z, generator=g) * 0.7).to(dev)
    eps = torch.randn(B, z, generator=g).to(dev)
    prior = torch.randn(700, z, generator=g, dtype=torch.float64).to(dev) if with_prior else None
    idx = torch.randint(0, 700, (B,), generator=g, dtype=torch.int32).to(dev) if with_prior else None
    w = (torch.randn(N, z, generator=g) * 0.2).to(dt).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    code = 0 if dt == torch.float32 else 1
    z0, s0, k0 = torch.zeros(B, z, dtype=dt, device=dev), torch.zeros(B, z, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    L.check(raw_.favae_reparam_kl_fwd(L.ptr(fc), L.ptr(eps), L.ptr(prior), L.ptr(idx), f_(1.3), i_(B), i_(z), L.ptr(z0), L.ptr(s0), L.ptr(k0),
                                      i_(code), L.stream_ptr()))
    z1, s1, k1 = torch.zeros_like(z0), torch.zeros_like(s0), torch.zeros_like(k0)
    out = torch.zeros(B, N, dtype=dt, device=dev)
    L.check(raw_.favae_reparam_kl_decoder_input(L.ptr(fc), L.ptr(eps), L.ptr(prior), L.ptr(idx), f_(1.3), i_(B), i_(z), L.ptr(z1), L.ptr(s1),
                                                L.ptr(k1), L.ptr(w), L.ptr(bias), L.ptr(out), i_(N), i_(code), L.stream_ptr()))
    assert torch.equal(z0, z1) and torch.equal(s0, s1)
    assert abs(k0.item() - k1.item()) <= 1e-12 * abs(k0.item())
    want = z0.double() @ w.double().t() + bias.double()
    tol = 1e-5 if dt == torch.float32 else 5e-3          # bf16: one rounding of the result
    assert _err(out.double(), want) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,z,N,with_prior", [(64, 20, 4096, True), (37, 32, 192, False), (100, 20, 4096, True),
                                               (64, 100, 4096, True), (64, 128, 2048, False), (33, 65, 192, True)])
def test_decoder_input_backward_with_reparam_backward_in_one_launch(dt, B, z, N, with_prior):
    """favae_decoder_input_bwd_reparam == favae_colsum_cast + (dz = draw @ W) + favae_reparam_kl_bwd_bias; called twice
    in a row (the ticket word must be left at zero)."""
    L = _L()
    raw_ = L.raw()
    dev = "cuda"
    g = torch.Generator().manual_seed(B + z + N)
    fc = (torch.randn(B, 2 * z, generator=g) * 0.7).to(dev)
    eps = torch.randn(B, z, generator=g).to(dev)
    prior = torch.randn(700, z, generator=g, dtype=torch.float64).to(dev) if with_prior else None
    idx = torch.randint(0, 700, (B,), generator=g, dtype=torch.int32).to(dev) if with_prior else None
    stdq = torch.exp(0.5 * (fc[:, z:] + 1e-6)).contiguous()
    w = (torch.randn(N, z, generator=g) * 0.2).to(dt).to(dev)
    dact = (torch.randn(B, N, generator=g) * 0.05).to(dev)
    code = 0 if dt == torch.float32 else 1
    tau, beta = 1.3, 0.7
    # the separate calls
    draw0, db0 = torch.zeros(B, N, dtype=dt, device=dev), torch.zeros(N, device=dev)
    scratch = torch.zeros(2 * 1024 * 64 + 4096, dtype=torch.float64, device=dev)
    L.check(raw_.favae_colsum_cast(L.ptr(dact), L.ptr(draw0), ll_(B), i_(N), L.ptr(db0), L.ptr(scratch), sz_(scratch.numel()), i_(code),
                                   L.stream_ptr()))
    dz0 = (draw0.double() @ w.double()).float().contiguous()
    dfc0, dbf0 = torch.zeros(B, 2 * z, dtype=dt, device=dev), torch.zeros(2 * z, device=dev)
    if B * 2 * z <= 8192:
        L.check(raw_.favae_reparam_kl_bwd_bias(L.ptr(fc), L.ptr(eps), L.ptr(prior), L.ptr(idx), L.ptr(stdq), L.ptr(dz0), f_(tau), f_(beta),
                                               i_(B), i_(z), L.ptr(dfc0), L.ptr(dbf0), i_(code), L.stream_ptr()))
    else:                      # beyond the one-block form's limit: the grid form in fp32, bias gradient = its column sums
        dfc32 = torch.zeros(B, 2 * z, device=dev)
        L.check(raw_.favae_reparam_kl_bwd(L.ptr(fc), L.ptr(eps), L.ptr(prior), L.ptr(idx), L.ptr(stdq), L.ptr(dz0), f_(tau), f_(beta),
                                          i_(B), i_(z), L.ptr(dfc32), i_(0), L.stream_ptr()))
        dfc0, dbf0 = dfc32.to(dt), dfc32.sum(0)
    for _ in range(2):
        draw1, db1, dz1 = torch.zeros_like(draw0), torch.zeros_like(db0), torch.zeros(B, z, device=dev)
        dfc1, dbf1 = torch.zeros_like(dfc0), torch.zeros_like(dbf0)
        L.check(raw_.favae_decoder_input_bwd_reparam(L.ptr(dact), L.ptr(w), L.ptr(draw1), L.ptr(db1), L.ptr(dz1), L.ptr(fc), L.ptr(eps),
                                                     L.ptr(prior), L.ptr(idx), L.ptr(stdq), f_(tau), f_(beta), i_(B), i_(z), i_(N),
                                                     L.ptr(dfc1), L.ptr(dbf1), i_(code), L.stream_ptr()))
        assert torch.equal(draw0, draw1)
        assert _err(db1, db0) < 1e-6
        assert _err(dz1, dz0) < 1e-5
        assert _err(dfc1, dfc0) < (1e-5 if dt == torch.float32 else 1e-2)
        assert _err(dbf1, dbf0) < 1e-5


def test_colsum_cast():
    L = _L()
    raw_ = L.raw()
    for R, C, dt in [(64, 4096, torch.bfloat16), (64, 4096, torch.float32), (1000, 40, torch.bfloat16)]:
        x = torch.randn(R, C, generator=torch.Generator().manual_seed(R + C)).cuda()
        y = torch.zeros(R, C, dtype=dt, device="cuda")
        out = torch.zeros(C, device="cuda")
        scratch = torch.zeros(2 * 1024 * 64 + 4096, dtype=torch.float64, device="cuda")
        L.check(raw_.favae_colsum_cast(L.ptr(x), L.ptr(y), ll_(R), i_(C), L.ptr(out), L.ptr(scratch), sz_(scratch.numel()),
                                       i_(0 if dt == torch.float32 else 1), L.stream_ptr()))
        assert torch.equal(y, x.to(dt))
        assert _err(out, x.double().sum(0)) < 1e-6


def test_sigmoid_ll_and_gradient():
    L = _L()
    raw_ = L.raw()
    B, D = 16, 12288
    g = torch.Generator().manual_seed(6)
    r = torch.randn(B, D, generator=g).double().requires_grad_(True)
    x = torch.rand(B, D, generator=g)
    s = torch.sigmoid(r)
    ll = (-0.5 * (D * np.log(2 * np.pi) + D * np.log(0.01) + ((x.double() - s) ** 2 / 0.01).sum(-1))).mean()
    (-ll).backward()
    dev = "cuda"
    draw, rec = torch.zeros(B, D, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    scratch = torch.zeros(4096, dtype=torch.float64, device=dev)
    rd, xd = r.detach().float().to(dev), x.to(dev)     # keep both alive: ptr() of a temporary would be recycled
    L.check(raw_.favae_sigmoid_ll(L.ptr(rd), L.ptr(xd), i_(B), ll_(D), f_(0.01), L.ptr(draw), L.ptr(rec),
                                  L.ptr(scratch), sz_(scratch.numel()), i_(0), L.stream_ptr()))
    assert abs(rec.item() + ll.item()) < 1e-6 * abs(ll.item())
    assert _err(draw, r.grad) < 1e-5
    # one-launch form with the bias gradient of the 3-channel output conv (pixels x 3 interleaved)
    for reps in range(2):   # twice: the ticket slot must be left clean
        draw2, rec2, dbias = torch.zeros(B, D, device=dev), torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(3, device=dev)
        scratch2 = torch.zeros(5 * 1024, dtype=torch.float64, device=dev)
        L.check(raw_.favae_sigmoid_ll_bias(L.ptr(rd), L.ptr(xd), i_(B), ll_(D), i_(3), f_(0.01), L.ptr(draw2), L.ptr(rec2), L.ptr(dbias),
                                           L.ptr(scratch2), sz_(scratch2.numel()), i_(0), L.stream_ptr()))
        assert abs(rec2.item() + ll.item()) < 1e-6 * abs(ll.item())
        assert torch.equal(draw2, draw)
        assert _err(dbias, r.grad.view(B, D // 3, 3).sum((0, 1))) < 1e-5


def test_adam_matches_torch_optim():
    L = _L()
    raw_ = L.raw()
    n = 4096 * 3
    g = torch.Generator().manual_seed(8)
    p = torch.randn(n, generator=g, requires_grad=True)
    opt = torch.optim.Adam([p], 1e-3)
    dev = "cuda"
    pd, md, vd = p.detach().clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        p.grad = gr.clone()
        opt.step()
        grd = gr.to(dev)
        L.check(raw_.favae_adam_step(L.ptr(pd), L.ptr(grd), L.ptr(md), L.ptr(vd), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8),
                                     ll_(step), L.stream_ptr()))
        assert (pd.cpu() - p.detach()).abs().max().item() < 2e-7


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Co,taps,Ci", [(128, 9, 64), (3, 9, 64), (40, 1, 4096), (4096, 1, 20), (64, 9, 3), (100, 9, 36), (512, 9, 256)])
def test_weight_prep(dt, Co, taps, Ci):
    L = _L()
    raw_ = L.raw()
    w = torch.randn(Co, taps, Ci, generator=torch.Generator().manual_seed(Co))
    dev = "cuda"
    wc, wt = torch.zeros(Co, taps, Ci, dtype=dt, device=dev), torch.zeros(Ci, taps, Co, dtype=dt, device=dev)
    wd = w.to(dev)
    L.check(raw_.favae_weight_prep(L.ptr(wd), L.ptr(wc), L.ptr(wt), i_(Co), i_(taps), i_(Ci), i_(0 if dt == torch.float32 else 1),
                                   L.stream_ptr()))
    assert torch.equal(wc.cpu(), w.to(dt))
    assert torch.equal(wt.cpu(), w.to(dt).permute(2, 1, 0).contiguous())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Co,taps,Ci,extra", [(128, 9, 64, 384), (40, 1, 4096, 40), (4096, 1, 20, 4096), (512, 9, 1024, 1536), (64, 9, 64, 0)])
def test_adam_fused_with_weight_prep(dt, Co, taps, Ci, extra):
    """favae_adam_wprep == favae_adam_step(_clear) followed by favae_weight_prep, bit for bit (same arithmetic per element)."""
    L = _L()
    raw_ = L.raw()
    dev = "cuda"
    nw = Co * taps * Ci
    n = nw + extra
    g = torch.Generator().manual_seed(Co + Ci)
    p0, g0 = torch.randn(n, generator=g).to(dev), (torch.randn(n, generator=g) * 0.1).to(dev)
    m0, v0 = (torch.randn(n, generator=g) * 0.01).to(dev), (torch.rand(n, generator=g) * 1e-3).to(dev)
    code = 0 if dt == torch.float32 else 1
    pa, ga, ma, va = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    L.check(raw_.favae_adam_step_clear(L.ptr(pa), L.ptr(ga), L.ptr(ma), L.ptr(va), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8), ll_(7),
                                       L.stream_ptr()))
    wca, wta = torch.zeros(nw, dtype=dt, device=dev), torch.zeros(nw, dtype=dt, device=dev)
    L.check(raw_.favae_weight_prep(L.ptr(pa), L.ptr(wca), L.ptr(wta), i_(Co), i_(taps), i_(Ci), i_(code), L.stream_ptr()))
    pb, gb, mb, vb = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    wcb, wtb = torch.zeros(nw, dtype=dt, device=dev), torch.zeros(nw, dtype=dt, device=dev)
    L.check(raw_.favae_adam_wprep(L.ptr(pb), L.ptr(gb), L.ptr(mb), L.ptr(vb), ll_(n), i_(Co), i_(taps), i_(Ci), L.ptr(wcb), L.ptr(wtb),
                                  i_(1), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8), ll_(7), i_(code), L.stream_ptr()))
    for a, b in ((pa, pb), (ma, mb), (va, vb), (ga, gb), (wca, wcb), (wta, wtb)):
        assert torch.equal(a, b)
    assert float(gb.abs().max()) == 0.0


@pytest.mark.parametrize("ranks", [1, 2, 3, 8])
def test_adam_on_the_summed_gradient_of_the_ranks(ranks):
    """favae_adam_step_clear_scaled(g_sum, 1 / ranks) == favae_adam_step_clear(g_sum / ranks) bit for bit (the data-parallel epoch
    all-reduces the sum and leaves the division to the optimiser step); gscale = 1 is favae_adam_step_clear itself."""
    L = _L()
    raw_ = L.raw()
    n = 4096 + 64
    g = torch.Generator().manual_seed(ranks)
    p0, gs = torch.randn(n, generator=g).cuda(), (torch.randn(n, generator=g) * 0.1 * ranks).cuda()
    m0, v0 = (torch.randn(n, generator=g) * 0.01).cuda(), (torch.rand(n, generator=g) * 1e-3).cuda()
    sc = np.float32(1.0) / np.float32(ranks)
    pa, ga, ma, va = p0.clone(), gs * float(sc), m0.clone(), v0.clone()
    L.check(raw_.favae_adam_step_clear(L.ptr(pa), L.ptr(ga), L.ptr(ma), L.ptr(va), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8), ll_(3),
                                       L.stream_ptr()))
    pb, gb, mb, vb = p0.clone(), gs.clone(), m0.clone(), v0.clone()
    L.check(raw_.favae_adam_step_clear_scaled(L.ptr(pb), L.ptr(gb), L.ptr(mb), L.ptr(vb), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8),
                                              ll_(3), f_(float(sc)), L.stream_ptr()))
    for a, b in ((pa, pb), (ma, mb), (va, vb), (ga, gb)):
        assert torch.equal(a, b)


def test_adam_that_leaves_an_overwritten_gradient_prefix():
    """favae_adam_step_clear_from == favae_adam_step_clear with g[0 .. keep) left as it is (a weight gradient that the next step's
    unsplit GEMM stores whole); favae_conv_wgrad_auto_splitk names those launches: enc.4 / dec.0 / dec.1 / enc.3 at batch 64 run
    unsplit, enc.1 / enc.2 / dec.2 / dec.3 split their pixel reduction."""
    L = _L()
    raw_ = L.raw()
    n, keep = 4096 + 64, 4096
    g = torch.Generator().manual_seed(5)
    p0, g0 = torch.randn(n, generator=g).cuda(), (torch.randn(n, generator=g) * 0.1).cuda()
    m0, v0 = (torch.randn(n, generator=g) * 0.01).cuda(), (torch.rand(n, generator=g) * 1e-3).cuda()
    pa, ga, ma, va = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    L.check(raw_.favae_adam_step_clear(L.ptr(pa), L.ptr(ga), L.ptr(ma), L.ptr(va), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8), ll_(3),
                                       L.stream_ptr()))
    pb, gb, mb, vb = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    L.check(raw_.favae_adam_step_clear_from(L.ptr(pb), L.ptr(gb), L.ptr(mb), L.ptr(vb), ll_(n), ll_(keep), f_(1e-3), f_(0.9), f_(0.999),
                                            f_(1e-8), ll_(3), L.stream_ptr()))
    for a, b in ((pa, pb), (ma, mb), (va, vb)):
        assert torch.equal(a, b)
    assert torch.equal(gb[:keep], g0[:keep]) and float(gb[keep:].abs().max()) == 0.0
    pc, gc, mc, vc = p0.clone(), g0.clone(), m0.clone(), v0.clone()   # two kept ranges, gradient scaled by 1 (the data-parallel bucket form)
    L.check(raw_.favae_adam_step_clear_scaled_keep(L.ptr(pc), L.ptr(gc), L.ptr(mc), L.ptr(vc), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8),
                                                   ll_(3), f_(1.0), ll_(1024), ll_(2048), ll_(3072), L.stream_ptr()))
    for a, b in ((pa, pc), (ma, mc), (va, vc)):
        assert torch.equal(a, b)
    kept = torch.zeros(n, dtype=torch.bool, device="cuda")
    kept[:1024] = True
    kept[2048:3072] = True
    assert torch.equal(gc[kept], g0[kept]) and float(gc[~kept].abs().max()) == 0.0
    sk = lambda co, ci, k: raw_.favae_conv_wgrad_auto_splitk(i_(co), i_(ci), ll_(k), i_(3))
    assert [sk(1024, 512, 64 * 4), sk(512, 1024, 64 * 4), sk(256, 512, 64 * 16), sk(512, 256, 64 * 16)] == [1, 1, 1, 1]
    assert min(sk(128, 64, 64 * 256), sk(256, 128, 64 * 64), sk(128, 256, 64 * 64), sk(64, 128, 64 * 256)) > 1


def test_bf16_conversion_is_round_to_nearest_even():
    """The conversion every kernel uses (gfx950's v_cvt_pk_bf16_f32) against the integer round-to-nearest-even form and against
    torch's: every finite value class (all 65 536 upper halves x ties / just below / just above, subnormals, infinities, signed
    zeros) bit for bit; NaN stays NaN."""
    L = _L()
    raw_ = L.raw()
    hi = torch.arange(0, 65536, dtype=torch.int64)
    lows = torch.tensor([0x0000, 0x0001, 0x7fff, 0x8000, 0x8001, 0xffff, 0x1234, 0xfedc], dtype=torch.int64)
    bits = ((hi[:, None] << 16) | lows[None, :]).reshape(-1)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    x = bits.view(torch.float32).cuda()
    g = torch.Generator().manual_seed(0)
    x = torch.cat([x, (torch.randn(1 << 20, generator=g) * torch.exp(torch.randn(1 << 20, generator=g) * 20)).cuda()])
    hw = torch.zeros(x.numel(), dtype=torch.bfloat16, device="cuda")
    sw = torch.zeros_like(hw)
    L.check(raw_.favae_debug_bf16_convert(L.ptr(x), L.ptr(hw), L.ptr(sw), ll_(x.numel()), L.stream_ptr()))
    ref = x.to(torch.bfloat16)
    finite = torch.isfinite(x) | torch.isinf(x)
    h, s, r = hw.view(torch.int16), sw.view(torch.int16), ref.view(torch.int16)
    assert torch.equal(h[finite], s[finite]) and torch.equal(h[finite], r[finite])
    nan = torch.isnan(x)
    assert bool(torch.isnan(hw[nan].float()).all()) and bool(torch.isnan(sw[nan].float()).all())


@pytest.mark.parametrize("clear", [0, 1])
def test_adam_with_bf16_copy_of_the_weights(clear):
    """favae_adam_step_copy (encoder.0: 64 x 9 x 3 weights + bias + BN affine) == favae_adam_step(_clear) followed by
    favae_weight_prep of the weight part, bit for bit."""
    L = _L()
    raw_ = L.raw()
    dev = "cuda"
    Co, taps, Ci = 64, 9, 3
    nw, n = Co * taps * Ci, Co * taps * Ci + 3 * 64
    g = torch.Generator().manual_seed(5)
    p0, g0 = torch.randn(n, generator=g).to(dev), (torch.randn(n, generator=g) * 0.1).to(dev)
    m0, v0 = (torch.randn(n, generator=g) * 0.01).to(dev), (torch.rand(n, generator=g) * 1e-3).to(dev)
    pa, ga, ma, va = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    step = raw_.favae_adam_step_clear if clear else raw_.favae_adam_step
    L.check(step(L.ptr(pa), L.ptr(ga), L.ptr(ma), L.ptr(va), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8), ll_(3), L.stream_ptr()))
    wca = torch.zeros(nw, dtype=torch.bfloat16, device=dev)
    L.check(raw_.favae_weight_prep(L.ptr(pa), L.ptr(wca), None, i_(Co), i_(taps), i_(Ci), i_(1), L.stream_ptr()))
    pb, gb, mb, vb = p0.clone(), g0.clone(), m0.clone(), v0.clone()
    wcb = torch.zeros(nw, dtype=torch.bfloat16, device=dev)
    L.check(raw_.favae_adam_step_copy(L.ptr(pb), L.ptr(gb), L.ptr(mb), L.ptr(vb), ll_(n), f_(1e-3), f_(0.9), f_(0.999), f_(1e-8), ll_(3),
                                      i_(clear), L.ptr(wcb), ll_(nw), L.stream_ptr()))
    for a, b in ((pa, pb), (ma, mb), (va, vb), (ga, gb), (wca, wcb)):
        assert torch.equal(a, b)
    assert (float(gb.abs().max()) == 0.0) == bool(clear)

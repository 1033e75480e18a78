"""FA-VAE training-iteration benchmark on MI355X (BASELINE.json metric: samples/s per FA-VAE train iteration).

One "step" = one full SSHIBA iteration through the drop-in API (fa-vae_amd/condition_a_VAE/favae.py, variant A):
update_w, update_Z, ImgVAE.trainloop (E = 5 epochs of batch-64 steps with the FA prior, reference schedule for
iterations 1-2, condition_a_VAE/favae.py:794-795), update_x over all N, update_tau/b/alpha, the multilabel update,
the reconstruction diagnostic, update_bound and pruning.  Workload = BASELINE.json configs[1]: N = 202 599 synthetic
64x64x3 images + 40 labels, latent 20, Kc = 50, bf16 conv path + float64 FA updates.  Inputs are resident in HBM before
the timed region.  N GPUs: one process per GPU (torchrun), N sharded, weak scaling (202 599 samples per GPU).

    python bench.py --gpus 1 --steps 2 --warmup 1
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "fa-vae_amd")
for p in (ROOT, PKG, os.path.join(PKG, "condition_a_VAE")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

E_EPOCHS = 5
HBM_CLASSES = {7: "BatchNorm + LeakyReLU backward apply (colreduce<BnBwdApplyF>)", 13: "Adam step + gradient clear (adam_kernel<true>)"}
CLASS_NAMES = {14: "conv thin-input (VALU, one thread per pixel)", 2: "igemm_kc gather 64x64", 3: "igemm_kc gather 128x32", 6: "igemm_kc scatter 64x64",
               8: "igemm_kc gather elementwise-load", 9: "igemm_kc scatter elementwise-load", 10: "igemm_pm wgrad 32x128",
               11: "igemm_pm wgrad 128x32", 12: "igemm_pm wgrad 64x64", 15: "wgrad narrow-channel (VALU)"}


# kernel class -> name in the rocprofv3 traces (profiles/r1_pmc_traffic_per_launch.json, tools/pmc_traffic.py)
CLASS_KERNEL = {2: "void igemm_kc<unsigned short, 64, 64, 2, 2, false, false>(KcParams)",
                3: "void igemm_kc<unsigned short, 128, 32, 4, 1, false, false>(KcParams)",
                6: "void igemm_kc<unsigned short, 64, 64, 2, 2, true, false>(KcParams)",
                8: "void igemm_kc<unsigned short, 64, 64, 2, 2, false, true>(KcParams)",
                11: "void igemm_pm<unsigned short, 128, 32, 4, 1>(PmParams)",
                12: "void igemm_pm<unsigned short, 64, 64, 2, 2>(PmParams)"}


CLASS_KERNEL.update({7: "void colreduce<1, 4, BnBwdApplyF<unsigned short, float, float> >(int, int, int, int, int, double*, unsigned int*, BnBwdApplyF<unsigned short, float, float>)",
                     13: "void adam_kernel<true, false>(float*, float*, float*, float*, unsigned long, float, float, float, float, float, unsigned short*, unsigned long)"})
PMC_JSON = os.path.join(ROOT, "profiles", "r3_pmc_traffic_per_launch.json")


def pmc_traffic(cls, dtype, key=None):
    """HBM-side bytes per launch of a kernel class from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE runs of tools/prof_step.py, gfx950 corrections applied); None when there is no such record."""
    if dtype != "bf16" or not os.path.exists(PMC_JSON):
        return None
    d = json.load(open(PMC_JSON))
    if key is not None:
        return d.get(key, {}).get("hbm_bytes")
    name = CLASS_KERNEL.get(cls)
    rec = d.get(name) if name else None
    if rec is None and name:      # kernel signatures carry template / argument lists: match on the prefix before the arguments
        stem = name.split("(")[0].rstrip(">")      # (later template parameters have been appended since: match the leading ones)
        rec = next((v for k, v in d.items() if k.startswith(stem)), None)
    return None if rec is None else rec["hbm_bytes"]


def synth(n, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((n, 3, 64, 64), dtype=np.float32)
    rs = np.random.RandomState(seed)
    zt = rs.randn(n, 5)
    lab = (zt @ rs.randn(5, 40) > 0).astype(int)
    return img, lab


def cpu_baseline(n, z, kc):
    """The oracle (CPU restatement, `kind: port`) timed on this box's host cores on a bounded sample of the same
    workload and extrapolated to one iteration.  Reported next to the GPU number; it is not the target."""
    from oracle import fa_numpy as fo
    from oracle import vae_torch as vt
    torch.manual_seed(0)
    np.random.seed(0)
    B = 64
    vae = vt.OracleImgVAE(dimx=z, channels=3, lr=1e-3, dataset="cpu")
    img = torch.rand(B, 3, 64, 64)
    prior = torch.randn(B, z, dtype=torch.float64)
    tau = torch.Tensor(np.asarray(np.array([1.5])))
    ts = []
    for i in range(10):
        t0 = time.time()
        vae.optim.zero_grad()
        loss, *_ = vt.loss_terms(vae.P, vae.B, img, prior, tau, 0.01, 1)
        loss.backward()
        vae.optim.step()
        ts.append(time.time() - t0)
    t_step = float(np.median(ts[2:]))
    with torch.no_grad():
        t0 = time.time()
        for _ in range(8):
            vt.encoder(img, vae.P, vae.B, False)
        t_enc = (time.time() - t0) / 8
        zin = torch.randn(B, z)
        t0 = time.time()
        for _ in range(4):
            vt.decoder(zin, vae.P, vae.B, False)
        t_dec = (time.time() - t0) / 4
    rs = np.random.RandomState(3)
    zt = rs.randn(n, 5)
    x0 = zt @ rs.randn(5, z) + 0.1 * rs.randn(n, z)          # stands in for the encoder moments of the image view
    lab = (zt @ rs.randn(5, 40) > 0).astype(int)
    fa_model = fo.SSHIBAOracle(kc, 1, variant="A")
    fa_model.fit(fo.struct_data(x0, "reg"), fo.struct_data(lab, "mult"), iterate=False)
    t0 = time.time()
    fa_model.update(iter_count=1)
    fa_model.depruning(1e-15)
    fa_model.update_bound()
    fa_model.pruning(1e-3)
    t_fa = time.time() - t0
    nb = (n + B - 1) // B
    t_iter = E_EPOCHS * nb * t_step + nb * t_enc + nb * t_dec + t_fa
    return {"value": E_EPOCHS * n / t_iter, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle (torch-CPU fp32 VAE + NumPy fp64 FA): 8 train steps, 8 encoder and 4 decoder batches of 64, "
                      "one FA update at N=%d; extrapolated to E=%d epochs + update_x + reconstruction over N "
                      "(t_step %.3fs t_enc %.3fs t_dec %.3fs t_fa %.2fs)" % (n, E_EPOCHS, t_step, t_enc, t_dec, t_fa)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", dest="n", type=int, default=202599, help="samples per GPU")
    ap.add_argument("--global-samples", dest="n_global", type=int, default=0,
                    help="strong scaling: this many samples in total, sharded over the ranks (dist.shard_bounds); overrides --samples")
    ap.add_argument("--kc", type=int, default=50)
    ap.add_argument("--latent", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra (untimed) iteration that splits VAE and FA time")
    ap.add_argument("--groups", type=str, default="8,16,32", help="BatchNorm groups per step of the grouped-batch figures: the first of the list goes to "
                    "secondary.grouped, every one to secondary.grouped_sweep (0: skip)")
    a = ap.parse_args()

    if os.environ.get("FAVAE_HANG_DUMP"):                # debugging aid: dump all Python stacks after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["FAVAE_HANG_DUMP"]), exit=True)
    # stdout carries exactly ONE line, the JSON: native libraries write there too (RCCL prints a version banner when its
    # communicator comes up), so fd 1 is pointed at stderr for the run and the result goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FAVAE_DIST_BACKEND", "nccl") != "nccl":     # rehearsal of the N>1 path on a one-GPU box (gloo)
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local)
    force_dp = os.environ.get("FAVAE_FORCE_DP") == "1"     # one-rank rehearsal of the whole multi-rank code path (RCCL)
    if ws > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("FAVAE_DIST_BACKEND", "nccl") == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda:%d" % local))
        else:
            torch.distributed.init_process_group(os.environ["FAVAE_DIST_BACKEND"])
    os.environ["FAVAE_CONV_DTYPE"] = a.dtype

    import favae
    from favae_hip import lib

    scaling = "weak"
    n_total = a.n * ws
    if a.n_global > 0:                                    # strong scaling: fixed global N, np.array_split-style shards
        from favae_hip import dist as _fd
        lo, hi = _fd.shard_bounds(a.n_global, ws, rank)
        a.n, n_total, scaling = hi - lo, a.n_global, "strong"
    img, lab = synth(a.n, 1 + rank)
    np.random.seed(0)
    torch.manual_seed(0)
    model = favae.SSHIBA(a.kc, 1)
    v_img = model.struct_data(img, "img", latent_dim=a.latent, lr=1e-3, dataset="synthetic")
    v_lab = model.struct_data(lab, "mult")
    model.fit(v_img, v_lab, max_iter=0, pruning_crit=1e-3, verbose=0, store=False, pretrained=1)   # initialisation only
    model.img_vae[0]._resident(img)                      # images resident in HBM before anything is timed
    model.img_vae[0].img = None

    def one_iteration():
        model.keep_training[0] = True                    # the FA-prior training iteration (reference iterations 1-2)
        model.update(iter_count=1)
        model.depruning(1e-15)
        model.L.append(model.update_bound())
        model.pruning(1e-3)

    def fence():
        torch.cuda.synchronize()
        if ws > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    raw = lib.raw()
    for _ in range(a.warmup):
        one_iteration()
    raw.favae_prof_enable(1, 257)    # prime: coprime to the launches per step of every timed class; ~0.4 % of the launches carry events
    fence()
    t0 = time.time()
    for _ in range(a.steps):
        one_iteration()
    fence()
    dt = time.time() - t0
    raw.favae_prof_enable(0, 1)
    prof = (ctypes.c_double * 48)()
    lib.check(raw.favae_prof_collect(prof))
    if ws > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()

    # secondary figures of SURVEY 8(d), from ONE extra iteration outside the timed region with a device synchronisation
    # around the VAE trainloop: VAE-step samples/s and FA-update samples/s (everything of the iteration but the trainloop)
    secondary = None
    if not a.no_secondary:
        from favae_hip import imgvae as _iv
        acc = {"trainloop": 0.0, "update_x": 0.0, "reconstruction_r2": 0.0}
        origs = {k: getattr(_iv.ImgVAE, k) for k in acc}

        def timed(name):
            f = origs[name]

            def g(self, *args, **kw):
                torch.cuda.synchronize()
                ts = time.time()
                r = f(self, *args, **kw)
                torch.cuda.synchronize()
                acc[name] += time.time() - ts
                return r
            return g
        for k in acc:
            setattr(_iv.ImgVAE, k, timed(k))
        fence()
        ts = time.time()
        one_iteration()
        fence()
        tot = time.time() - ts
        for k, f in origs.items():
            setattr(_iv.ImgVAE, k, f)
        evalp = acc["update_x"] + acc["reconstruction_r2"]
        fa_only = max(tot - acc["trainloop"] - evalp, 1e-9)
        nsteps = E_EPOCHS * ((a.n + 63) // 64)
        secondary = {"vae_trainloop_samples_per_s": E_EPOCHS * n_total / acc["trainloop"],
                     "vae_ms_per_64_image_step": 1e3 * acc["trainloop"] / nsteps,
                     "fa_only_samples_per_s": n_total / fa_only,
                     "eval_passes_samples_per_s": 2 * n_total / max(evalp, 1e-9),
                     "seconds": {"iteration": tot, "trainloop": acc["trainloop"], "update_x": acc["update_x"],
                                 "reconstruction_r2": acc["reconstruction_r2"],
                                 "fa_updates_bound_pruning": fa_only},
                     "note": "one extra, untimed iteration with a device synchronisation around trainloop / update_x / reconstruction; "
                             "fa_only = update_w, update_Z, update_tau/b/alpha, the multilabel update, update_bound, (de)pruning"}

    # secondary.grouped: the labelled grouped-batch mode of the VAE step (G BatchNorm groups of 64 per optimiser step, the step of a
    # G-rank data-parallel run on one GPU; NOT the reference's trajectory, never `value`): two epochs over the same resident images
    # through a second executor, the second one timed
    grouped, grouped_sweep = None, []
    group_list = [int(x) for x in str(a.groups).split(",") if x.strip() and int(x) > 1]
    if not a.no_secondary and group_list and ws == 1:
        from favae_hip.engine import VaeEngine
        vae = model.img_vae[0]
        data = vae._resident(vae._source())
        peak_g = 2500.0 if a.dtype == "bf16" else 157.3
        gflop_img = 2.0 * (3 * (197_591_040 + 12_288 * a.latent) - 1_769_472) / 1e9
        for groups in group_list:
            ge = VaeEngine(a.latent, 3, batch_max=64 * groups, dtype=a.dtype)
            ge.load_state_dict({k: v.detach().cpu() for k, v in vae.state_dict().items()})
            gb = 64 * groups
            nst = ge.epoch_steps(a.n, gb)
            gm = torch.zeros(nst, 2, dtype=torch.float64, device="cuda")
            gprior = torch.zeros(a.n, a.latent, dtype=torch.float64, device="cuda")
            gt = None
            for _ in range(2):
                perm = torch.randperm(a.n).to(torch.int32).cuda()
                eps = torch.randn(a.n, a.latent).cuda()
                torch.cuda.synchronize()
                ts = time.time()
                ge.train_epoch(data, perm, eps, gprior, 1.0, 1.0, 0.01, 1e-5, gb, gm)
                torch.cuda.synchronize()
                gt = time.time() - ts
            ge.poll_error()
            rec = {"config": {"groups": groups, "batch": gb, "bn_group": 64, "steps_per_epoch": nst},
                   "samples_per_s": a.n / gt, "ms_per_step": 1e3 * gt / nst, "gflop_per_step": gflop_img * gb,
                   "tflops": gflop_img * a.n / gt / 1e3, "mfma_frac": gflop_img * a.n / gt / 1e3 / peak_g,
                   "loss_last_step": float(gm[-1].sum().item())}
            grouped_sweep.append(rec)
            if grouped is None:
                grouped = dict(rec)
                grouped["note"] = ("one timed epoch of favae_vae_train_epoch at batch 64 x groups (BatchNorm statistics per group of 64, mean loss over "
                                   "the batch, one Adam step per batch: the step of a `groups`-rank data-parallel run); same flop convention as `step`; "
                                   "compare with secondary.vae_trainloop_samples_per_s (batch 64)")
            del ge, gm, gprior
            torch.cuda.empty_cache()

    if rank == 0:
        n_global = n_total
        kern, hbm = [], []
        for c in range(16):
            cnt, sec, work = prof[3 * c], prof[3 * c + 1], prof[3 * c + 2]
            if cnt <= 0:
                continue
            if c in HBM_CLASSES:
                hbm.append({"cls": c, "kernel": HBM_CLASSES[c], "sampled_launches": int(cnt), "avg_us": 1e6 * sec / cnt,
                            "gb_per_s": work / sec / 1e9, "algorithmic_bytes_per_launch": work / cnt, "seconds": sec})
            else:
                kern.append({"cls": c, "kernel": CLASS_NAMES.get(c, str(c)), "sampled_launches": int(cnt), "avg_us": 1e6 * sec / cnt,
                             "tflops": work / sec / 1e12, "algorithmic_flops_per_launch": work / cnt, "share_of_gemm_time": sec})
        tot = sum(k["share_of_gemm_time"] for k in kern) or 1.0
        for k in kern:
            k["share_of_gemm_time"] /= tot
        dom = max(kern, key=lambda k: k["share_of_gemm_time"]) if kern else None
        domh = max(hbm, key=lambda k: k["seconds"]) if hbm else None
        peak = 2500.0 if a.dtype == "bf16" else 157.3
        # algorithmic work of one 64-image step: forward + backward of the conv VAE, dense-tap convention
        macs_fwd = 197_591_040 + 12_288 * a.latent
        step_gflop = 64 * 2.0 * (3 * macs_fwd - 1_769_472) / 1e9
        nparam = 12_587_203 + 12_290 * a.latent
        step_gb_min = (10 * 4 * nparam + 64 * 4 * 2.1e6) / 1e9      # parameters / Adam 10 x 4 B, activations + their gradients ~2 x 2.1 MB per sample
        step_gb_meas = pmc_traffic(None, a.dtype, "__per_step__")
        step_ms = secondary["vae_ms_per_64_image_step"] if secondary else None
        out = {
            "metric": "samples/sec per FA-VAE train iteration (VAE fwd-bwd + FA update)",
            "value": E_EPOCHS * n_global * a.steps / dt, "unit": "samples/s", "n_gpus": ws, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "bf16 conv MFMA (fp32 accumulate) + f64 FA" if a.dtype == "bf16" else "f32 conv MFMA + f64 FA",
            "data": "synthetic",
            "config": {"workload": "condition_a_VAE full CelebA scale: N=%d synthetic 64x64x3 images %s + 40-label view, "
                                   "latent %d, Kc=%d, E=%d epochs of batch 64 per iteration"
                                   % (n_global if scaling == "strong" else a.n, "in total, sharded over the GPUs" if scaling == "strong" else "per GPU",
                                      a.latent, a.kc, E_EPOCHS),
                       "samples_per_gpu": a.n, "global_samples": n_global, "epochs_per_iteration": E_EPOCHS, "batch": 64,
                       "parallelism": "dp%d (N-sharded, RCCL all-reduce of VAE gradients and FA statistics)" % ws,
                       "bound_last": float(model.L[-1])},
            "roofline": None if dom is None else {
                "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s",
                "frac": dom["tflops"] / peak, "traffic": pmc_traffic(dom["cls"], a.dtype),
                "note": "dominant GEMM class by time. HIP events on the launch stream around every 257th launch inside the timed "
                        "region (weight gradients run on a side stream next to the main chain, so their durations include that "
                        "overlap); achieved = algorithmic flops per launch (dense-tap convention) / mean launch duration; traffic = "
                        "mean HBM-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes "
                        "(profiles/r3_pmc_traffic_per_launch.json)"},
            "roofline_hbm": None if domh is None else {
                "bound": "hbm", "kernel": domh["kernel"], "achieved": domh["gb_per_s"], "peak": 8000.0, "unit": "GB/s",
                "frac": domh["gb_per_s"] / 8000.0, "traffic": pmc_traffic(domh["cls"], a.dtype),
                "note": "top HBM-bound kernel class by time, timed the same way; achieved = algorithmic bytes per launch / mean launch "
                        "duration (these launches run on side streams or between GEMMs of the main chain and share the chip)"},
            "step": {"gflop": step_gflop, "gbytes_algorithmic_min": step_gb_min, "gbytes_measured_pmc": None if step_gb_meas is None else step_gb_meas / 1e9,
                     "ms": step_ms,
                     "tflops": None if not step_ms else step_gflop / step_ms, "mfma_frac": None if not step_ms else step_gflop / step_ms / peak,
                     "hbm_gb_per_s": None if (not step_ms or step_gb_meas is None) else step_gb_meas / 1e6 / step_ms,
                     "note": "one 64-image training step: algorithmic flops (SURVEY 8d), minimum and PMC-measured HBM bytes, wall time per step"},
            "kernels": kern, "kernels_hbm": hbm,
        }
        if secondary is not None:
            if grouped is not None:
                secondary["grouped"] = grouped
                secondary["grouped_sweep"] = grouped_sweep
            out["secondary"] = secondary
        if ws == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.n, a.latent, a.kc)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if ws > 1 or force_dp:
        from favae_hip import dist as fdist
        fdist.shutdown()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

// The conv-VAE executor: sequences one training step / encoder pass / decoder pass of the FA-VAE network
// (reference condition_a_VAE/celeba_deepvanilla_vae.py:16-105 architecture, :227-250 step) out of the kernels in
// igemm.hip and vae_kernels.hip.  The host passes device buffers (PyTorch tensors used as memory); this file owns
// the layer table, the carve-up of the workspace and the launch order.  Nothing here allocates device memory.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <thread>
#include <vector>
#include "common.h"
#include "favae_hip.h"
#include "comm_internal.h"
#include "smallconv.h"

namespace {

enum Kind { CONV_S2 = 0, LINEAR = 1, CONVT_S2 = 2, CONV_S1 = 3 };

struct Layer {
    int kind, Cin, Cout, Hin, Hout, taps;
    bool bn;
    long long w_off, b_off, g_off, be_off;  // parameter arena (floats)
    long long rm_off, rv_off;               // BN buffer arena (floats)
    long long wc_off, wt_off;               // shadow arena (elements of dtype); -1 = not materialised
};

constexpr int NL = 13;
constexpr int L_FC = 5, L_DI = 6, L_OUT = 12;
constexpr int MAX_BINS = 32;
constexpr int DBP_SLABS = 256;   // slabs of a column reduction at most (red_geom, vae_kernels.hip)

inline long long al8(long long x) { return (x + 7) & ~7LL; }

void build_layers(int z, int ch, Layer* L, long long* nparam, long long* nbuf, long long* nshadow, bool f32) {
    const int hid[5] = {64, 128, 256, 512, 1024};
    int c = ch, h = 64;
    for (int i = 0; i < 5; ++i) {
        L[i] = Layer{CONV_S2, c, hid[i], h, h / 2, 9, true};
        c = hid[i];
        h /= 2;
    }
    L[L_FC] = Layer{LINEAR, 4096, 2 * z, 1, 1, 1, false};
    L[L_DI] = Layer{LINEAR, z, 4096, 1, 1, 1, false};
    c = 1024; h = 2;
    for (int i = 0; i < 4; ++i) {
        L[7 + i] = Layer{CONVT_S2, c, hid[3 - i], h, 2 * h, 9, true};
        c = hid[3 - i];
        h *= 2;
    }
    L[11] = Layer{CONVT_S2, 64, 64, 32, 64, 9, true};
    L[L_OUT] = Layer{CONV_S1, 64, ch, 64, 64, 9, false};
    long long po = 0, bo = 0, so = 0;
    for (int i = 0; i < NL; ++i) {
        Layer& l = L[i];
        const long long wn = (long long)l.Cout * l.taps * l.Cin;
        l.w_off = po; po = al8(po + wn);
        l.b_off = po; po = al8(po + l.Cout);
        l.g_off = l.be_off = l.rm_off = l.rv_off = -1;
        if (l.bn) {
            l.g_off = po; po = al8(po + l.Cout);
            l.be_off = po; po = al8(po + l.Cout);
            l.rm_off = bo; bo = al8(bo + l.Cout);
            l.rv_off = bo; bo = al8(bo + l.Cout);
        }
        l.wc_off = -1;
        if (!f32) { l.wc_off = so; so = al8(so + wn); }
        l.wt_off = -1;
        if (i != 0) { l.wt_off = so; so = al8(so + wn); }   // enc.0 has no input gradient
    }
    *nparam = po; *nbuf = bo; *nshadow = so;
}

// ---- enqueue worker -----------------------------------------------------------------------------------------------
// A batch-64 step is ~100 launches plus ~45 event records / cross-stream waits.  Measured on the MI355X box
// (tools/micro/host_cost.hip): a launch costs the host 2.4 us, an event record 0.9 us, but a record + wait pair that makes
// another stream depend on it 7.4 us and the first launch behind such a wait several more: the side-stream traffic
// (weight gradients, layer-wise Adam, weight re-layout, the next batch's gather) is 3/4 of the 0.8 ms of host time per step.
// During an epoch those calls are issued by a second host thread: the caller's thread keeps only the main chain (launches
// + one event record per fork, ~0.2 ms per step), hands each side-stream job over as a closure, and meets the worker once
// per step (the events the main stream waits on must have been recorded, and each event's previous use consumed).
// (A captured hipGraph of the step was measured instead and is slower on this ROCm: 1.27 vs 1.08 ms per step, and
// hipGraphLaunch returns only when the previous launch of the graph has drained.)
struct Worker {
    static constexpr unsigned CAP = 512;
    std::function<int()> q[CAP];
    std::atomic<unsigned> head{0}, tail{0};   // tail: next slot the caller fills; head: next slot the worker runs
    std::atomic<int> rc{0};
    std::atomic<bool> stop{false};
    char err[512] = "";
    std::thread th;
    double busy = 0.0, drain_wait = 0.0; long long jobs = 0;   // FAVAE_VERBOSE accounting
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void run(int dev) {
        (void)hipSetDevice(dev);
        for (;;) {
            const unsigned h = head.load(std::memory_order_relaxed);
            if (h == tail.load(std::memory_order_acquire)) {
                if (stop.load(std::memory_order_acquire)) return;
                __builtin_ia32_pause();
                continue;
            }
            std::function<int()>& f = q[h % CAP];
            if (rc.load(std::memory_order_relaxed) == 0) {   // after a failure the remaining jobs are dropped
                const double t0 = now();
                // FAVAE_DEBUG_FAIL_JOB=n (tests only): the n-th job of every worker fails without issuing its calls
                static const long long fail_at = getenv("FAVAE_DEBUG_FAIL_JOB") ? atoll(getenv("FAVAE_DEBUG_FAIL_JOB")) : -1;
                int r;
                if (fail_at >= 0 && jobs == fail_at) { favae_set_error("favae: injected failure of worker job %lld", jobs); r = FAVAE_ERR_HIP; }
                else r = f();
                busy += now() - t0; ++jobs;
                if (r != FAVAE_OK) { strncpy(err, favae_last_error(), sizeof(err) - 1); rc.store(r, std::memory_order_release); }
            }
            f = nullptr;
            head.store(h + 1, std::memory_order_release);
        }
    }
    void start(int dev) { th = std::thread([this, dev] { run(dev); }); }
    void push(std::function<int()> f) {
        const unsigned t = tail.load(std::memory_order_relaxed);
        while (t - head.load(std::memory_order_acquire) >= CAP) __builtin_ia32_pause();
        q[t % CAP] = std::move(f);
        tail.store(t + 1, std::memory_order_release);
    }
    // everything handed over so far has been issued (its HIP calls have returned)
    int drain() {
        const unsigned t = tail.load(std::memory_order_relaxed);
        const double t0 = now();
        while (head.load(std::memory_order_acquire) != t) __builtin_ia32_pause();
        drain_wait += now() - t0;
        const int r = rc.load(std::memory_order_acquire);
        if (r != FAVAE_OK) favae_set_error("%s", err);
        return r;
    }
    // queue position now; wait_until(m): everything handed over before mark() returned m has been issued
    unsigned mark() const { return tail.load(std::memory_order_relaxed); }
    int wait_until(unsigned m) {
        const double t0 = now();
        while ((int)(head.load(std::memory_order_acquire) - m) < 0) __builtin_ia32_pause();
        drain_wait += now() - t0;
        const int r = rc.load(std::memory_order_acquire);
        if (r != FAVAE_OK) favae_set_error("%s", err);
        return r;
    }
    void finish() { stop.store(true, std::memory_order_release); if (th.joinable()) th.join(); }
};

struct Ada { bool on; float lr, b1, b2, eps; long long step; bool epoch; };

struct Vae {
    int z, ch, Bmax, dtype, esz;
    Layer L[NL];
    long long nparam, nbuf, nshadow;
    float *P, *G, *M, *V, *BUF;
    char* SH;
    // workspace
    float* raw[NL]; char* act[NL]; float* dact[NL]; char* draw[NL];
    float *mean[NL], *invstd[NL];
    float *xf, *stdq, *dz, *s1s2;
    char *xt, *zlat;
    double *scratch, *metrics2;
    size_t scratch_doubles;
    // split-K targets small enough to be zeroed by one memset per training step
    bool raw_pooled[NL], dact_pooled[NL];
    // bf16 path: conv output (raw) and activation gradient (dact) of the largest BN layers are stored as bf16
    bool raw_bf[NL], dact_bf[NL];
    char* zero_lo; size_t zero_bytes;
    // per-channel sums taken in GEMM epilogues (favae_conv_igemm_red): BN statistics in the forward pass (sbins), the two
    // BN-backward reductions in the backward pass (gbins); part of the zero pool; *_fused = the last launch took them
    double *sbins[NL], *gbins[NL];
    bool stats_fused[NL], grad_fused[NL];
    // weight gradients run on a second stream next to the dgrad -> BN-backward chain of the main stream
    // and the layer-wise optimiser steps of an epoch on a third (side2), behind the weight gradient of their layer
    hipStream_t side, side2, side3; hipEvent_t ev_comm; hipEvent_t ev_fork[NL]; hipEvent_t ev_dg[NL]; hipEvent_t ev_wg[NL]; hipEvent_t ev_join; hipEvent_t ev_join2; bool overlap;   // ev_join2: end of the weight-gradient stream when encoder.0's update ran there
    // single-device epochs: Adam + weight re-layout of a layer follow its weight gradient on the side stream
    // epoch: steps of one epoch run back to back: Adam clears the gradients it has consumed (no memset per step) and the
    // main stream joins the side stream only when the next step's forward pass is about to read the weights
    Ada ada;
    int pending;   // layer whose optimiser step waits for the next fork event (-1: none); touched by side-stream jobs only
    Worker* wk;    // non-null while an epoch hands its side-stream calls to the enqueue worker
    hipStream_t probed_main;   // the caller's stream the probe of ensure_streams ran against
    unsigned beside;    // ensure_streams: pairs of streams measured to run on hardware queues of their own (bits: see there)
    int stream_score;   // ensure_streams: 21 = all three side streams run beside the main stream and each other
    bool join_pending; bool dg_all;
    bool ab_live[NL];    // this step's forward pass published (a | b) of the layer's BatchNorm: the backward pass takes the LeakyReLU slope from raw * a + b
    bool wg_keeps[NL];   // the weight gradient of the layer is stored whole by every step of this executor (unsplit GEMM): no zeroing
    // final_layer.0's activation is not materialised by the training step: final_layer.3's forward (favae_conv_out_ll_bn) and weight gradient
    // (favae_conv_wgrad_prefolded_bn) apply the BatchNorm + LeakyReLU to raw[11] at their operand loads, the backward pass takes the slope
    // from raw * a + b.  act11_skip_ok: this step may do so (set before the decoder's forward pass); act11_raw: it did.
    bool act11_skip_ok, act11_raw;
    // the gradient w.r.t. final_layer.0's activation is not stored: the input-gradient launch of final_layer.3 leaves the two BatchNorm-backward
    // sums only and bn_bwd forms the gradient again with the BatchNorm backward applied (favae_conv_folded_dgrad_bn)
    bool dact11_recompute;
    bool run_defer[NL];   // grouped-batch mode: the running statistics of the layer's BatchNorm are still to be updated (on the stream of its weight gradient)
    bool dp_epoch;   // steps of favae_vae_train_epoch_dp: gradients cleared by the bucket-wise Adam, joins through the communication stream
    hipEvent_t ev_fence[2]; bool ev_fence_ok;
    bool rk_done;   // stage 6 of this step already ran the reparameterisation backward (fused launch)
    bool join2;   // ev_join2 was recorded in the last step
    hipEvent_t ev_join_ext;   // when set: the event join_pending refers to (the communicator's), else ev_join
    unsigned fuse;   // FAVAE_FUSE_MASK (tuning / A-B knob): 1 fwd statistics, 2 bwd sums, 4 bwd sums on the thin-input path, 8 LL+bias, 16 KL+bias, 32 colsum+cast, 64 tap-folded image-side weight gradients (0x200: do not hoist encoder.0's gather to the start of the step; 0x800: encoder.0's update and its bf16 shadow as two launches; 0x1000: encoder.0's update on the optimiser stream; 0x2000: encoder.0's weight gradient + update on the weight-gradient stream instead of the caller's, memset / batch gather of the next step on the caller's; 0x4000: reparameterise + KL and decoder_input as two launches; 0x8000: their backward as three launches), 128 Adam + weight re-layout in one pass, 256 final conv + loss in one launch
    float* xf_alt; char* xt_alt;
    char* wk_out;   // weights of final_layer.3 in the MFMA operand order for favae_conv_out_ll (rebuilt whenever the layer changes)
    char* fold0_buf[2]; int fold_sel; hipEvent_t ev_batch, ev_fold; bool inline_tail;   // see train_stages
    bool fwd_folded;   // this step's encoder.0 forward reads the folded batch (the caller's stream has waited for the fold)
    bool out_folded;   // final_layer.3's output gradient is folded on the caller's stream, for its weight AND its input gradient
    char* fold0; size_t fold0_bytes; hipEvent_t ev_x; bool x_folded;   // encoder.0's folded input batch, gathered on the side stream while the forward pass runs
    // bf16 executor: the two ends of the encoder keep fp32 precision in the training step (FAVAE_BF16_EXACT_ENDS=0: plain bf16, as
    // before).  encoder.0 reads the image as two bf16 terms (hi + lo planes of the folded batch, fp32 master weights split in the
    // kernel), fc_mu | fc_var read encoder.4's activation unrounded with the fp32 master weights.  The image's rounding passes
    // through all five BatchNorm layers and was the largest single term of the error of the latent means at initialisation
    // (1.2e-2 of scale -> 0.75e-2; tools/dbg_parity_r3.py); both launches are bound by latency / their fp32 output, not by MFMA.
    bool hp; float* act4_f32; bool act4_ready;
    char* fold; size_t fold_bytes;   // [pixels][32] tap-folded operand of the two image-side weight gradients (side stream only)   // second batch buffer: the next gather must not overwrite what wgrad(encoder.0) still reads
    // The dtype copies / transposes of the weights exist twice.  In single-device epochs the optimiser step of a layer follows
    // its weight gradient on the same stream and writes the re-laid weights into the half the running step does not read
    // (the input-gradient GEMM of that layer may still be queued on the main stream): no event between the two streams for it.
    size_t sh_half; int sh_sel; bool merged; int upd_half;
    // BatchNorm + LeakyReLU of a layer applied in the operand load of the next convolution (favae_conv_igemm_bn_in): the main
    // chain has no launch between the two GEMMs; the activation the backward pass reads (weight gradient of the next layer,
    // LeakyReLU slopes) is materialised on the weight-gradient stream, which is idle during the forward pass
    bool data_u8;   // the resident data set handed to the step / encode / reconstruction calls is uint8 (favae_vae_set_data_u8)
    bool eval_fused, eval_ab_ready;   // eval-mode BatchNorm in the GEMM epilogues (large-batch eval executor); ab[] holds the running-stat form
    // column sums of draw[l] per slab of rows, left by the BN-backward apply of this step (FAVAE_DBIAS_DEFER); dbias_slabs[l] > 0:
    // the stream that takes layer l's weight gradient adds them up first (dbias_finish)
    double* dbp[NL]; int dbias_slabs[NL];
    // the five sums per block of the final conv + loss launch of this step (FAVAE_DBIAS_DEFER): added up by the first block of
    // final_layer.3's weight gradient (ll_B > 0: pending, with the loss term's destination)
    double* llp; int ll_B; float ll_var; double* ll_out;
    float* ab[NL]; bool act_deferred[NL]; bool bn_published[NL]; bool fi_on; int fi_pend[NL], fi_npend, fi_flushes; hipEvent_t ev_fi[2], ev_acts; bool acts_wait;
    // Dependency points between the streams (see sp_record / sp_wait): in threaded single-device epochs they are 32-bit words in
    // device memory written / awaited by stream memory operations with the step number as value, otherwise the events above.
    unsigned* host_words;   // pinned host memory the device can read: [0] abort (set by the host to end every pending wait), [1] a wait gave up
    unsigned* flags; unsigned step_no, join_seq; bool vals; int vals_ok;   // vals_ok: -1 not probed, 0 unsupported, 1 usable
    unsigned mark_prev, mark_cur;   // worker queue positions at the start of the previous / this step (bounded run-ahead)
    char* shp(int half) const { return SH + (size_t)half * sh_half; }
    // grouped-batch mode: the affine form of layer l's BatchNorm as the forward pass wrote it (the backward launches take the LeakyReLU
    // slope from raw * a + b and leave the stored activation unread; FAVAE_FUSE_MASK bit 0x10000000: from the activation, as before)
    const float* gab(int l) const { return (fuse & 0x10000000u) ? nullptr : ab[l]; }
    const void* Wf(int l) const { return dtype == FAVAE_F32 ? (const void*)(P + L[l].w_off) : (const void*)(shp(sh_sel) + L[l].wc_off * esz); }
    const void* Wt(int l) const { return (const void*)(shp(sh_sel) + L[l].wt_off * esz); }
    long long rows(int l, int B) const { return (long long)B * L[l].Hout * L[l].Hout; }
};

// Dependency points.  A point is recorded on one stream once per training step and awaited on others.
enum { SP_FORK = 0 /* + layer */, SP_WG = NL /* + layer */, SP_DG = 2 * NL /* + layer */, SP_BATCH = 3 * NL, SP_FOLD, SP_JOIN, SP_JOIN2, SP_X,
       SP_FI0, SP_FI1, SP_ACTS, SP_FENCE_M /* + bucket: the caller's stream */, SP_FENCE_S = SP_FENCE_M + 4 /* + bucket: weight-gradient stream */,
       SP_COUNT = SP_FENCE_S + 4 };

hipEvent_t sp_event(Vae& v, int sp) {
    if (sp < SP_WG) return v.ev_fork[sp - SP_FORK];
    if (sp < SP_DG) return v.ev_wg[sp - SP_WG];
    if (sp < SP_BATCH) return v.ev_dg[sp - SP_DG];
    if (sp >= SP_FENCE_M) return v.ev_fence[sp >= SP_FENCE_S ? 1 : 0];   // (events mode: favae_vae_fence records these itself)
    switch (sp) {
    case SP_BATCH: return v.ev_batch;
    case SP_FOLD: return v.ev_fold;
    case SP_JOIN: return v.ev_join;
    case SP_JOIN2: return v.ev_join2;
    case SP_X: return v.ev_x;
    case SP_FI0: return v.ev_fi[0];
    case SP_FI1: return v.ev_fi[1];
    default: return v.ev_acts;
    }
}

// Values mode (v.vals): hipStreamWriteValue32 / hipStreamWaitValue32 on flags[sp] with the step number `seq` of the record.
// Measured on MI355X / ROCm 7.2 (tools/micro/stream_value.hip): 2.1-2.3 us of host time per call against ~12 us for an event
// record + wait pair, half the hand-over latency on the device -- and, unlike an event, the wait may be issued before the record,
// so the caller's thread and the enqueue worker need not meet once per step.  A word only grows, so a late wait is satisfied by
// a later record of the same point, which the stream order of the recording side places after the one that was meant.
__global__ void set_flag_kernel(FlagCarry c) { carry_store(c); }
long long g_carry_flushes = 0;   // FAVAE_VERBOSE accounting: dependency words that needed a launch of their own

// Values mode (v.vals): the points are 32-bit words in device memory that hold the number of the step that last passed them.
// Measured on MI355X / ROCm 7.2 (tools/micro/stream_value.hip):
//  * an event record between two kernels of a chain costs the chain ~4.1 us (a barrier packet), hipStreamWriteValue32 ~4.6 us, a
//    one-wave kernel 2.3 us -- and a store at the top of the chain's NEXT kernel costs nothing.  So a point passed on the caller's
//    stream is not enqueued at all: its word rides with the next GEMM launch of that stream (FlagCarry, common.h), whose first
//    block stores it -- that kernel runs only when everything before it on the stream is complete;
//  * hipStreamWaitValue32 / hipStreamWriteValue32 take 2.1-2.4 us of host time, an event record + wait pair ~12 us;
//  * a wait may be issued before the write it waits for, so the caller's thread and the enqueue worker need not meet once per step.
// A word only grows, so a late wait is satisfied by a later pass of the same point, which stream order places after the one meant.
int sp_record(Vae& v, int sp, hipStream_t s, unsigned seq, bool carry = false) {
    if (!v.vals) { FAVAE_HIP(hipEventRecord(sp_event(v, sp), s)); return FAVAE_OK; }
    if (carry) {
        FlagCarry& c = favae_flag_carry();
        if (c.n < 4 && (c.n == 0 || c.val == seq)) { c.p[c.n++] = v.flags + sp; c.val = seq; return FAVAE_OK; }
    }
    FAVAE_HIP(hipStreamWriteValue32(s, v.flags + sp, seq, 0));
    return FAVAE_OK;
}
// the words handed to favae_flag_carry() that no launch has taken along: a launch of their own
int sp_flush_carry(hipStream_t s) {
    FlagCarry& c = favae_flag_carry();
    if (!c.n) return FAVAE_OK;
    const FlagCarry k = c;
    c.n = 0;
    ++g_carry_flushes;
    hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(64), 0, s, k);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
// The wait of values mode: ONE wave that polls the word until it has reached `seq`.  Every such wave reaches an exit: when the word
// arrives; when the host raises the abort word (pinned host memory: the host needs no queue, no copy engine and no stream to write
// it, so a failed enqueue can always end the waits it left behind -- a release written by a copy or a kernel may be queued on a
// hardware queue that one of these waves occupies); or when `max_ticks` have passed, in which case it reports through host_words[1]
// and the next call on the executor fails instead of the process hanging.  (hipStreamWaitValue32, used until round 3, is a polling
// wave too, with neither exit.)
__global__ void wait_word_kernel(const unsigned* word, unsigned seq, unsigned* host_words, unsigned long long max_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    unsigned it = 0;
    while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 255u) == 0u) {
            if (__hip_atomic_load(host_words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
            if (wall_clock64() - t0 > max_ticks) { __hip_atomic_store(host_words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
    }
}
constexpr unsigned long long WAIT_MAX_TICKS = 400000000ULL;   // 4 s of the 100 MHz wall clock: a step is < 1 ms

int sp_wait(Vae& v, int sp, hipStream_t s, unsigned seq) {
    static const int dbg_skip = getenv("FAVAE_DEBUG_SKIP_WAIT") ? atoi(getenv("FAVAE_DEBUG_SKIP_WAIT")) : -1;   // timing experiments only
    if (dbg_skip >= 0 && (sp == dbg_skip || (dbg_skip == 99 && (sp == SP_JOIN || sp == SP_BATCH)))) return FAVAE_OK;
    if (v.vals) {
        hipLaunchKernelGGL(wait_word_kernel, dim3(1), dim3(64), 0, s, (const unsigned*)(v.flags + sp), seq, v.host_words, WAIT_MAX_TICKS);
        FAVAE_HIP(hipGetLastError());
    } else {
        FAVAE_HIP(hipStreamWaitEvent(s, sp_event(v, sp), 0));
    }
    return FAVAE_OK;
}

struct Carver {
    char* base; size_t off, cap;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

void choose_storage(Vae& v) {
    // Batch-64 executors keep the conv outputs (raw) and activation gradients (dact) in fp32: stored as bf16 for the layers with
    // >= 4 M elements per batch (round 2, FAVAE_BF16_RAW) the step was 0.3 % shorter only -- these tensors live in the 256 MB
    // Infinity Cache between producer and consumer -- which did not pay for the extra rounding; the switch was removed at the end of
    // round 3 (the launches added since were never taught that storage at batch 64, and it was no faster: 0.852 against 0.845 ms).
    const bool off = true;
    for (int l = 0; l < NL; ++l) v.raw_bf[l] = v.dact_bf[l] = false;
    // One exception (end of round 3): the gradient w.r.t. final_layer.0's activation, 16.8 M elements per batch and half of all
    // BatchNorm-backward bytes, is stored as bf16: folded_dgrad_kernel writes and the BatchNorm backward reads 34 instead of 67 MB,
    // step 0.843 -> 0.827 ms on the same box (tools/sweep_store.sh; the other large layers and the conv outputs on top of it: no
    // further gain).  The two BatchNorm-backward sums still come from the fp32 accumulators of the producing launch, and the
    // grouped-batch mode has stored every activation gradient this way since it exists.  FAVAE_DACT11_BF16=0: fp32, as before.
    {
        const char* e = getenv("FAVAE_DACT11_BF16");
        if (v.dtype == FAVAE_BF16 && !(e && e[0] == '0')) v.dact_bf[11] = true;
        // and, with that in place, the conv outputs of decoder.3 and final_layer.0 (4.2 + 16.8 M elements, read three times each: forward
        // BatchNorm, backward sums, BatchNorm backward): 0.827 -> 0.820 ms (final_layer.0 alone 0.823); their BatchNorm statistics
        // still come from the fp32 accumulators.  FAVAE_RAW1011_BF16=0: fp32.
        const char* e2 = getenv("FAVAE_RAW1011_BF16");
        if (v.dtype == FAVAE_BF16 && !(e2 && e2[0] == '0')) v.raw_bf[10] = v.raw_bf[11] = true;
    }
    // Grouped-batch mode (bf16 executor with Bmax > 64): at G = 8 the conv outputs and activation gradients of a step are 4 GB, far
    // beyond the 256 MB Infinity Cache, and the step is bound by their HBM traffic (40 B per BatchNorm element and step; final_layer.0
    // alone is half of it).  The activation gradients (dact) of every BatchNorm layer and the conv outputs (raw) of the DECODER side
    // are stored as bf16 (28 B per element); the encoder keeps fp32 conv outputs, so the latent moments see no extra rounding, and
    // the BatchNorm sums still come from the fp32 accumulators of the producing GEMM.  FAVAE_GROUPED_BF16_STORE=0: fp32 as before.
    const char* gs = getenv("FAVAE_GROUPED_BF16_STORE");
    if (v.dtype == FAVAE_BF16 && v.Bmax > 64 && off && !(gs && gs[0] == '0')) {
        for (int l = 0; l < NL; ++l) {
            v.raw_bf[l] = v.L[l].bn && l >= 7;
            v.dact_bf[l] = v.L[l].bn && l != 4;   // (dact of encoder.4 comes from the fc heads' input-gradient GEMM, fp32)
        }
    }
}

int bins_for(long long rows);

void carve(Vae& v, char* base, size_t cap, size_t* need) {
    choose_storage(v);
    Carver c{base, 0, cap};
    const int B = v.Bmax;
    // zero pool first: the fp32 conv outputs (raw) and input gradients (dact) of the layers whose GEMMs may split the
    // reduction, i.e. the small ones; everything in [zero_lo, zero_lo + zero_bytes) is cleared once per training step
    const size_t pool_max = (size_t)1 << 20;   // elements
    v.zero_lo = (char*)c.take(0);
    // A convolution layer's tensors stay out of the pool where the launch that produces them runs unsplit at this executor's batch
    // (it stores every element; at batch 64: raw of encoder.2 / decoder.1 and the gradients flowing into them, 16.8 of the 36 MB of
    // the per-step memset).  A smaller batch (the tail of an epoch) may split after all: the launch is then told that its output
    // is not zeroed and clears it itself.  (FAVAE_FUSE_MASK bit 0x2000000: everything small in the pool, as before)
    const char* fenv = getenv("FAVAE_FUSE_MASK");   // (v.fuse is read after the workspace is laid out)
    const bool pool_all = fenv && (strtoul(fenv, nullptr, 0) & 0x2000000u) != 0;
    auto is_conv = [&](int l) { return l >= 0 && l < NL && (v.L[l].kind == CONV_S2 || v.L[l].kind == CONVT_S2); };
    for (int l = 0; l < NL; ++l) {
        const Layer& L = v.L[l];
        const size_t n = (size_t)v.rows(l, B) * L.Cout;
        v.raw_pooled[l] = v.dact_pooled[l] = n <= pool_max;
        if (n <= pool_max && !pool_all && is_conv(l)) {
            if (favae_conv_igemm_auto_splitk(B, L.Hout, L.Hout, L.Cin, L.Cout, 3, 2, L.kind == CONVT_S2) == 1) v.raw_pooled[l] = false;
            if (is_conv(l + 1)) {   // dact[l] is the input gradient of layer l + 1: scatter form for a convolution, gather form for a transposed one
                const Layer& N = v.L[l + 1];
                if (favae_conv_igemm_auto_splitk(B, N.Hin, N.Hin, N.Cout, N.Cin, 3, 2, N.kind == CONV_S2) == 1) v.dact_pooled[l] = false;
            }
        }
        if (v.raw_pooled[l]) v.raw[l] = (float*)c.take(n * 4);
        if (v.dact_pooled[l]) v.dact[l] = (float*)c.take(n * 4);
    }
    v.dz = (float*)c.take((size_t)B * v.z * 4);
    // grouped-batch mode (Bmax > 64: up to Bmax / 64 BatchNorm groups per step): every BatchNorm layer has bins, per group
    const int Gmax = std::max(1, B / 64);
    for (int l = 0; l < NL; ++l) {
        const bool use = v.L[l].bn && (v.L[l].Cout <= 256 || Gmax > 1);
        const size_t nb = Gmax > 1 ? (size_t)Gmax * bins_for(v.rows(l, 64)) : (size_t)MAX_BINS;   // (a step of <= 64 images never needs more)
        v.sbins[l] = use ? (double*)c.take(nb * 2 * v.L[l].Cout * 8) : nullptr;
        v.gbins[l] = use ? (double*)c.take(nb * 2 * v.L[l].Cout * 8) : nullptr;
        v.stats_fused[l] = v.grad_fused[l] = false;
    }
    {
        char* end = (char*)c.take(0);
        v.zero_bytes = (size_t)(end - v.zero_lo);
    }
    for (int l = 0; l < NL; ++l) {
        const size_t n = (size_t)v.rows(l, B) * v.L[l].Cout;
        if (!v.raw_pooled[l]) v.raw[l] = (float*)c.take(n * (v.raw_bf[l] ? 2 : 4));
        if (!v.dact_pooled[l]) v.dact[l] = (float*)c.take(n * (v.dact_bf[l] ? 2 : 4));
        v.act[l] = (char*)c.take(n * v.esz);
        v.draw[l] = (char*)c.take(n * v.esz);
        v.mean[l] = (float*)c.take((size_t)Gmax * v.L[l].Cout * 4);
        v.invstd[l] = (float*)c.take((size_t)Gmax * v.L[l].Cout * 4);
        v.ab[l] = (float*)c.take((size_t)Gmax * v.L[l].Cout * 8);   // a | b of the forward pass's BatchNorm, per group
        v.dbp[l] = v.L[l].bn ? (double*)c.take((size_t)DBP_SLABS * v.L[l].Cout * 8) : nullptr;
    }
    const size_t img = (size_t)B * 64 * 64 * v.ch;
    v.xf = (float*)c.take(img * 4);
    v.xt = (char*)c.take(img * v.esz);
    v.xf_alt = (float*)c.take(img * 4);
    v.xt_alt = (char*)c.take(img * v.esz);
    v.wk_out = (char*)c.take(9 * 16 * 16 * 4);
    v.llp = (double*)c.take((size_t)((B * 64 * 64 + 255) / 256) * 5 * 8);
    v.fold_bytes = favae_conv_wgrad_folded_scratch(B, 64, 64, v.dtype);
    v.fold = (char*)c.take(v.fold_bytes);
    v.fold0_bytes = (size_t)B * 32 * 32 * 32 * v.esz * (v.dtype == FAVAE_BF16 ? 2 : 1);   // bf16: hi and lo plane
    v.act4_f32 = v.dtype == FAVAE_BF16 ? (float*)c.take((size_t)B * 4096 * 4) : nullptr;
    v.fold0_buf[0] = (char*)c.take(v.fold0_bytes);
    v.fold0_buf[1] = (char*)c.take(v.fold0_bytes);
    v.fold0 = v.fold0_buf[0];
    v.zlat = (char*)c.take((size_t)B * v.z * v.esz);
    v.stdq = (float*)c.take((size_t)B * v.z * 4);
    v.s1s2 = (float*)c.take(2 * 1024 * 4);
    v.scratch_doubles = 2 * 1024 * 64 + 4096;
    v.scratch = (double*)c.take(v.scratch_doubles * 8);
    v.metrics2 = (double*)c.take(2 * 8);
    v.flags = (unsigned*)c.take((SP_COUNT + 1) * sizeof(unsigned));   // (+1: the word the support probe writes)
    *need = c.off + 256;
}

#define RC(x) do { int rc_ = (x); if (rc_ != FAVAE_OK) return rc_; } while (0)

// bins for the epilogue sums of a layer output of `rows` pixels: ~16 tiles of 64 pixels meet on one address
int bins_for(long long rows) {
    int nb = 4;
    while (nb < MAX_BINS && (long long)nb * 16 * 64 < rows) nb *= 2;
    return nb;
}

// ---- side streams on hardware queues of their own ----------------------------------------------------------------
// The runtime hands out its few hardware queues (4 per process here) to streams round-robin, so a side stream created
// after some other streams of the process (torch, RCCL, a second executor) can land on the hardware queue of the main
// stream and then runs serialised with it (measured: 1.15 -> 1.54 ms per step).  Which queue a stream got is not exposed,
// so it is measured once, at the first training step: a bounded ~150 us spin on one stream, time stamps on the others; a
// stamp taken before the spin ended ran beside it, i.e. on another queue.  Of 4 consecutively created streams the three
// that run beside the main stream and beside each other become the weight-gradient, optimiser and communication streams.
__global__ void probe_spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    int it = 0;
    while (wall_clock64() - t0 < ticks && ++it < (1 << 13)) __builtin_amdgcn_s_sleep(8);   // bounded either way
    if (threadIdx.x == 0) *out = wall_clock64();
}
__global__ void probe_stamp(unsigned long long* out) {
    if (threadIdx.x == 0) *out = wall_clock64();
}

int ensure_streams(Vae& v, hipStream_t main) {
    if (v.side) return FAVAE_OK;
    constexpr int NC = 4, NS = NC + 1;
    hipStream_t s[NS] = {main, nullptr, nullptr, nullptr, nullptr};
    // (Side streams at the lowest stream priority were tried in round 2 and removed in round 3: -0.6 % with one executor, and a
    // run that never finished with two -- DESIGN.md section 4, "the two hangs".  All streams of an executor share one priority.)
    for (int i = 1; i < NS; ++i)
        if (hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) {
            for (int j = 1; j < i; ++j) (void)hipStreamDestroy(s[j]);
            favae_set_error("favae: could not create the side streams");
            return FAVAE_ERR_HIP;
        }
    unsigned long long* buf = (unsigned long long*)v.scratch;      // NS*NS stamps; the scratch is idle before the first step
    unsigned long long host[NS * NS];
    bool beside[NS][NS];
    const char* e = getenv("FAVAE_STREAM_PROBE");
    const bool probe = !(e && e[0] == '0');
    bool measured = false;
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) beside[i][j] = i != j;
    if (probe) {
        // a failing probe is not fatal: the streams are kept and the unprobed choice (the first three) is used
        bool ok = hipDeviceSynchronize() == hipSuccess;
        for (int i = 0; i < NS && ok; ++i) {
            hipLaunchKernelGGL(probe_spin, dim3(1), dim3(64), 0, s[i], 15000ULL, buf + i * NS + i);   // 100 MHz clock: 150 us
            for (int j = 0; j < NS; ++j)
                if (j != i) hipLaunchKernelGGL(probe_stamp, dim3(1), dim3(64), 0, s[j], buf + i * NS + j);
            ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        }
        ok = ok && hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess;
        if (ok)
            for (int i = 0; i < NS; ++i)
                for (int j = 0; j < NS; ++j)
                    if (i != j) beside[i][j] = host[i * NS + j] < host[i * NS + i];
        measured = ok;
    }
    auto ok2 = [&](int a, int b) { return beside[a][b] && beside[b][a]; };
    int pick[3] = {-1, -1, -1}, best = -1;
    for (int a = 1; a < NS; ++a)
        for (int b = 1; b < NS; ++b)
            for (int c = 1; c < NS; ++c) {
                if (a == b || a == c || b == c) continue;
                // what matters most: weight gradients beside the main chain, then the optimiser / communication streams
                const int score = 8 * ok2(0, a) + 4 * ok2(0, b) + 4 * ok2(0, c) + 2 * ok2(a, b) + 2 * ok2(a, c) + ok2(b, c);
                if (score > best) { best = score; pick[0] = a; pick[1] = b; pick[2] = c; }
            }
    v.side = s[pick[0]]; v.side2 = s[pick[1]]; v.side3 = s[pick[2]];
    for (int i = 1; i < NS; ++i)
        if (i != pick[0] && i != pick[1] && i != pick[2]) (void)hipStreamDestroy(s[i]);
    v.stream_score = best;
    v.probed_main = main;
    // which pairs run side by side: bit 0 main-side, 1 main-side2, 2 main-side3, 3 side-side2, 4 side-side3, 5 side2-side3
    v.beside = (ok2(0, pick[0]) ? 1u : 0u) | (ok2(0, pick[1]) ? 2u : 0u) | (ok2(0, pick[2]) ? 4u : 0u) | (ok2(pick[0], pick[1]) ? 8u : 0u) |
               (ok2(pick[0], pick[2]) ? 16u : 0u) | (ok2(pick[1], pick[2]) ? 32u : 0u);
    if (!measured) v.beside = 0;   // (not probed: nothing is known about the queues)
    if (getenv("FAVAE_VERBOSE")) fprintf(stderr, "favae: side streams picked %d %d %d of 4 (score %d of 21)\n", pick[0], pick[1], pick[2], best);
    return FAVAE_OK;
}

template <class F> int side_do(Vae& v, F&& f);   // (defined with the enqueue worker's users below)

// fi >= 0: the input is raw[fi], the output of layer fi before its BatchNorm + LeakyReLU, applied in the operand load
// rows idx[b] of the resident data set (fp32 or uint8 NHWC) -> fp32 batch and / or dtype batch
int gather(const Vae& v, const float* data, const int* idx, int B, float* xf, void* xt, void* st) {
    const long long row = (long long)64 * 64 * v.ch;
    if (v.data_u8) return favae_gather_batch_u8((const unsigned char*)data, idx, B, row, xf, xt, v.dtype, st);
    return favae_gather_batch(data, idx, B, row, xf, xt, v.dtype, st);
}

// bins of layer l for the batch-64 launches: the layers with more than 256 channels take the channel-owner BatchNorm kernels there
// (their bins exist in executors built for the grouped-batch mode only)
inline double* sbins64(const Vae& v, int l) { return v.L[l].Cout <= 256 ? v.sbins[l] : nullptr; }
inline double* gbins64(const Vae& v, int l) { return v.L[l].Cout <= 256 ? v.gbins[l] : nullptr; }

int conv_fwd(Vae& v, int l, const void* in, int B, void* st, bool pooled = false, int fi = -1) {
    const Layer& L = v.L[l];
    const float* bias = v.P + L.b_off;
    const int sk = (pooled && v.raw_pooled[l]) ? -1 : 0;
    const int dt = v.dtype | (v.raw_bf[l] ? FAVAE_OUT_BF16 : 0);
    v.stats_fused[l] = false;
    v.bn_published[l] = false;
    if (pooled && B > 64 && L.bn) {
        // grouped-batch mode: B / 64 BatchNorm groups in one launch; the statistics of every group come from the GEMM epilogue
        // (bins per group) or, where the launch split its reduction, from a pass over the output
        const int G = B / 64, nb = bins_for(v.rows(l, 64));
        int fused = 0;
        if (l == 0 && v.fwd_folded && L.Cout == 64) {
            RC(folded_fwd_launch(v.fold0, v.hp ? (const void*)(v.P + L.w_off) : v.Wf(0), 9 * v.ch, bias, v.raw[0], v.rows(0, B), v.sbins[0], nb,
                                 v.hp ? 1 : 0, (hipStream_t)st, G));
            fused = 1;
        } else {
            const int scatter = L.kind == CONVT_S2, stride = L.kind == CONV_S1 ? 1 : 2;
            RC(favae_conv_igemm_red_groups(in, v.Wf(l), bias, v.raw[l], B, L.Hin, L.Hin, L.Cin, L.Hout, L.Hout, L.Cout, 3, stride, 1, scatter, sk,
                                           dt, v.sbins[l], nb, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 64, &fused, st));
        }
        if (!fused)
            RC(favae_gbn_sums(v.raw[l], nullptr, nullptr, nullptr, nullptr, nullptr, v.sbins[l], nb, G, v.rows(l, 64), L.Cout, 1,
                              v.dtype | (v.raw_bf[l] ? FAVAE_RAW_BF16 : 0), st));
        v.stats_fused[l] = true;
        return FAVAE_OK;
    }
    if (l == 0 && v.fwd_folded && pooled && fi < 0 && (v.fuse & 1u) && v.sbins[0] && !v.raw_bf[0]) {
        // encoder.0 of the training step as a K = 32 GEMM over the tap-folded batch (smallconv.hip, folded_fwd_kernel)
        RC(folded_fwd_launch(v.fold0, v.hp ? (const void*)(v.P + L.w_off) : v.Wf(0), 9 * v.ch, bias, v.raw[0], v.rows(0, B), v.sbins[0],
                             bins_for(v.rows(0, B)), v.hp ? 1 : 0, (hipStream_t)st));
        v.stats_fused[0] = true;
        return FAVAE_OK;
    }
    if (l == 0 && v.hp && pooled && fi < 0 && thin_in_applicable(v.ch, L.Cout, 3, 2, 0)) {
        // no folded batch (four channels, one stream): the thin-input convolution on the fp32 batch with the fp32 master weights
        const ThinRed none{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0};
        return thin_in_launch(v.xf, v.P + L.w_off, bias, v.raw[0], B, L.Hin, L.Hin, v.ch, L.Hout, L.Hout, L.Cout, 2, 0, FAVAE_F32, 0, none,
                              (hipStream_t)st);
    }
    if (l == L_FC && v.hp && pooled && v.act4_ready) {   // fc_mu | fc_var from the unrounded activation, fp32 matrix cores
        v.act4_ready = false;
        return favae_conv_igemm(v.act4_f32, v.P + L.w_off, bias, v.raw[l], B, 1, 1, L.Cin, 1, 1, L.Cout, 1, 1, 0, 0, sk, FAVAE_F32, st);
    }
    const bool publish = pooled && v.fi_on && (v.fuse & 1u) && sbins64(v, l) && L.kind != LINEAR && L.bn && !v.raw_bf[l];
    if (fi >= 0 || publish) {
        // fused chain: the launch publishes this layer's BatchNorm quantities from its epilogue sums (where the shape runs
        // unsplit) and / or reads its input through the BatchNorm + LeakyReLU of layer fi
        const bool own = (v.fuse & 1u) && sbins64(v, l) != nullptr && L.bn;
        const int scatter = L.kind == CONVT_S2, stride = L.kind == CONV_S1 ? 1 : 2;
        int fused = 0;
        RC(favae_conv_igemm_bn(fi >= 0 ? (const void*)v.raw[fi] : in, fi >= 0 ? v.ab[fi] : nullptr, v.Wf(l), bias, v.raw[l], B, L.Hin, L.Hin,
                               L.Cin, L.Hout, L.Hout, L.Cout, 3, stride, 1, scatter, sk, dt, own ? v.sbins[l] : nullptr,
                               own ? bins_for(v.rows(l, B)) : 0, own ? v.P + L.g_off : nullptr, own ? v.P + L.be_off : nullptr, 1e-5f, 0.1f,
                               v.mean[l], v.invstd[l], publish ? v.ab[l] : nullptr, own ? v.BUF + L.rm_off : nullptr,
                               own ? v.BUF + L.rv_off : nullptr, &fused, st));
        v.stats_fused[l] = fused != 0;
        v.bn_published[l] = fused != 0 && publish;
        return FAVAE_OK;
    }
    if (pooled && (v.fuse & 1u) && sbins64(v, l) && L.kind != LINEAR) {
        // training step: the zero pool (and with it sbins) was cleared at the start of the step; take the BN statistics
        // in the GEMM epilogue where this shape runs unsplit
        const int scatter = L.kind == CONVT_S2, stride = L.kind == CONV_S1 ? 1 : 2;
        int fused = 0;
        RC(favae_conv_igemm_red(in, v.Wf(l), bias, v.raw[l], B, L.Hin, L.Hin, L.Cin, L.Hout, L.Hout, L.Cout, 3, stride, 1, scatter,
                                sk, dt, v.sbins[l], bins_for(v.rows(l, B)), 1, nullptr, nullptr, nullptr, nullptr, &fused, st));
        v.stats_fused[l] = fused != 0;
        return FAVAE_OK;
    }
    switch (L.kind) {
    case CONV_S2:
        return favae_conv_igemm(in, v.Wf(l), bias, v.raw[l], B, L.Hin, L.Hin, L.Cin, L.Hout, L.Hout, L.Cout, 3, 2, 1, 0, sk, dt, st);
    case CONV_S1:
        return favae_conv_igemm(in, v.Wf(l), bias, v.raw[l], B, L.Hin, L.Hin, L.Cin, L.Hout, L.Hout, L.Cout, 3, 1, 1, 0, sk, dt, st);
    case CONVT_S2:
        return favae_conv_igemm(in, v.Wf(l), bias, v.raw[l], B, L.Hin, L.Hin, L.Cin, L.Hout, L.Hout, L.Cout, 3, 2, 1, 1, sk, dt, st);
    default:
        return favae_conv_igemm(in, v.Wf(l), bias, v.raw[l], B, 1, 1, L.Cin, 1, 1, L.Cout, 1, 1, 0, 0, sk, dt, st);
    }
}

int bn_fwd(Vae& v, int l, int B, bool train, void* st) {
    const Layer& L = v.L[l];
    const long long R = v.rows(l, B);
    const int dt = v.dtype | (v.raw_bf[l] ? FAVAE_RAW_BF16 : 0);
    if (l == 11) v.act11_raw = false;
    if (train && B > 64 && v.stats_fused[l]) {   // grouped-batch mode
        v.stats_fused[l] = false;
        const bool skip = l == 11 && v.act11_skip_ok && v.ab[l] && v.raw_bf[l];
        if (skip) v.act11_raw = true;
        // the running statistics (a 5 us launch per layer that nothing in the step waits for) leave the main chain: wgrad() queues them
        // (FAVAE_GBN_RUN_INLINE=1: behind this launch, as before)
        static const bool run_inline = getenv("FAVAE_GBN_RUN_INLINE") && atoi(getenv("FAVAE_GBN_RUN_INLINE")) == 1;
        if (v.run_defer[l])   // (a step that never reached the layer's weight gradient: the update it owes, from the statistics still in place)
            RC(favae_gbn_running(v.mean[l], v.invstd[l], B / 64, v.rows(l, 64), L.Cout, 1e-5f, 0.1f, v.BUF + L.rm_off, v.BUF + L.rv_off, st));
        const bool run_here = run_inline || l == 0;   // (encoder.0's weight gradient closes the step on the main chain, by a path of its own)
        v.run_defer[l] = !run_here;
        return favae_gbn_fwd(v.raw[l], v.sbins[l], bins_for(v.rows(l, 64)), v.P + L.g_off, v.P + L.be_off, skip ? nullptr : v.act[l], v.mean[l],
                             v.invstd[l], v.ab[l], run_here ? v.BUF + L.rm_off : nullptr, run_here ? v.BUF + L.rv_off : nullptr, B / 64,
                             v.rows(l, 64), L.Cout, 1e-5f, 0.1f, dt, st);
    }
    if (train && v.bn_published[l]) {   // the producing GEMM finished the statistics (mean, invstd, running stats, a | b)
        v.bn_published[l] = v.stats_fused[l] = false;
        return favae_bn_lrelu_fwd_ab(v.raw[l], v.ab[l], v.act[l], R, L.Cout, v.dtype, st);
    }
    if (train && B <= 64) v.ab_live[l] = false;
    if (train && v.stats_fused[l]) {
        v.stats_fused[l] = false;
        // final_layer.0 (16.8 M elements per batch): the backward pass does without the stored activation, twice 34 MB per step
        // (the input gradient of final_layer.3 with the BatchNorm-backward sums, and the BatchNorm backward itself): step 0.831 ->
        // 0.8255 ms.  The same for the other layers with epilogue sums (58 MB more, through favae_conv_igemm_red_groups with one
        // group) measured slower again: 0.824 with this layer alone, 0.825 with decoder.3 and encoder.0, 0.829 with all.
        if (l == 11 && v.dtype == FAVAE_BF16 && !(v.fuse & 0x10000000u)) {
            v.ab_live[l] = true;
            const bool skip = v.act11_skip_ok && v.raw_bf[l];
            if (skip) v.act11_raw = true;
            return favae_bn_train_fwd_bins_ab(v.raw[l], v.sbins[l], bins_for(R), v.P + L.g_off, v.P + L.be_off, skip ? nullptr : v.act[l], v.mean[l],
                                              v.invstd[l], v.ab[l], v.BUF + L.rm_off, v.BUF + L.rv_off, R, L.Cout, 1e-5f, 0.1f, dt, st);
        }
        return favae_bn_train_fwd_bins(v.raw[l], v.sbins[l], bins_for(R), v.P + L.g_off, v.P + L.be_off, v.act[l], v.mean[l],
                                       v.invstd[l], v.BUF + L.rm_off, v.BUF + L.rv_off, R, L.Cout, 1e-5f, 0.1f, dt, st);
    }
    if (train && l == 4 && v.hp && v.act4_f32 && !v.raw_bf[l] && favae_bn_train_fwd_keep32_ok(R, L.Cout)) {
        v.act4_ready = true;
        return favae_bn_train_fwd_keep32(v.raw[l], v.P + L.g_off, v.P + L.be_off, v.act[l], v.act4_f32, v.mean[l], v.invstd[l],
                                         v.BUF + L.rm_off, v.BUF + L.rv_off, R, L.Cout, 1e-5f, 0.1f, v.dtype, st);
    }
    if (train)
        return favae_bn_train_fwd(v.raw[l], v.P + L.g_off, v.P + L.be_off, v.act[l], v.mean[l], v.invstd[l], v.BUF + L.rm_off,
                                  v.BUF + L.rv_off, R, L.Cout, 1e-5f, 0.1f, v.scratch, v.scratch_doubles, dt, st);
    RC(favae_bn_eval_prep(v.BUF + L.rm_off, v.BUF + L.rv_off, L.Cout, 1e-5f, v.mean[l], v.invstd[l], st));
    return favae_bn_lrelu_fwd(v.raw[l], v.mean[l], v.invstd[l], v.P + L.g_off, v.P + L.be_off, v.act[l], R, L.Cout, 1, dt, st);
}

// may layer l's BatchNorm + LeakyReLU be left to the operand load of the next convolution?
bool fi_ok(const Vae& v, int l, bool train, bool pooled) {
    if (!(train && pooled && v.fi_on && v.bn_published[l]) || v.raw_bf[l] || (v.fuse & 0x80000u)) return false;   // (0x80000: publish only)
    const int nl = l + 1;
    if (nl >= NL || nl == L_FC || nl == L_OUT || v.L[nl].kind == LINEAR) return false;
    return v.L[l].Cout % 64 == 0 && v.L[l].Cout <= 256 && v.L[nl].Cout > 32;
}

// the consumers of the pending layers have been queued on `st` (they publish a | b): materialise the activations on the
// weight-gradient stream
int fi_flush(Vae& v, int B, void* st) {
    if (!v.fi_npend) return FAVAE_OK;
    const int k = v.fi_flushes++ & 1;
    const unsigned seq = v.step_no;
    RC(sp_record(v, SP_FI0 + k, (hipStream_t)st, seq));
    int ls[NL]; const int n = v.fi_npend;
    for (int i = 0; i < n; ++i) ls[i] = v.fi_pend[i];
    v.fi_npend = 0;
    v.acts_wait = true;
    int l0 = ls[0], l1 = n > 1 ? ls[1] : -1, l2 = n > 2 ? ls[2] : -1, l3 = n > 3 ? ls[3] : -1;
    return side_do(v, [&v, k, B, l0, l1, l2, l3, seq]() -> int {
        RC(sp_wait(v, SP_FI0 + k, v.side, seq));
        for (int l : {l0, l1, l2, l3})
            if (l >= 0) RC(favae_bn_lrelu_fwd_ab(v.raw[l], v.ab[l], v.act[l], v.rows(l, B), v.L[l].Cout, v.dtype, (void*)v.side));
        return sp_record(v, SP_ACTS, v.side, seq);
    });
}

int conv_bn_chain(Vae& v, int lo, int hi, const void* in, int B, bool train, void* st, bool pooled) {
    if (!train && v.eval_fused) {
        // eval passes over a data set (batch >= 256): BatchNorm with the running statistics is an affine map known before the
        // launch, so it rides with the LeakyReLU in the GEMM epilogue: one launch per layer, no fp32 conv output in memory
        for (int l = lo; l <= hi; ++l) {
            const Layer& L = v.L[l];
            if (L.Cin % 64 != 0) {   // encoder.0 keeps the thin-input convolution + apply
                RC(conv_fwd(v, l, in, B, st, pooled));
                RC(bn_fwd(v, l, B, false, st));
                in = v.act[l];
                continue;
            }
            if (!v.eval_ab_ready) RC(favae_bn_eval_ab(v.BUF + L.rm_off, v.BUF + L.rv_off, v.P + L.g_off, v.P + L.be_off, L.Cout, 1e-5f, v.ab[l], st));
            RC(favae_conv_igemm_affine(in, v.Wf(l), v.P + L.b_off, v.act[l], B, L.Hin, L.Hin, L.Cin, L.Hout, L.Hout, L.Cout, 3, 2, 1,
                                       L.kind == CONVT_S2, v.dtype | (v.dtype == FAVAE_BF16 ? FAVAE_OUT_BF16 : 0), v.ab[l], st));
            in = v.act[l];
        }
        return FAVAE_OK;
    }
    int fi = -1;
    for (int l = lo; l <= hi; ++l) {
        RC(conv_fwd(v, l, in, B, st, pooled, fi));
        if (fi >= 0) v.fi_pend[v.fi_npend++] = fi;       // its consumer is queued
        fi = -1;
        if (l < hi && fi_ok(v, l, train, pooled) && v.fi_npend < 3) {
            fi = l;
            v.stats_fused[l] = v.bn_published[l] = false;
            v.act_deferred[l] = true;
        } else {
            RC(fi_flush(v, B, st));
            RC(bn_fwd(v, l, B, train, st));
        }
        in = v.act[l];
    }
    return FAVAE_OK;
}

int encoder_fwd(Vae& v, int B, bool train, void* st, bool pooled = false) {
    RC(conv_bn_chain(v, 0, 4, v.xt, B, train, st, pooled));
    return conv_fwd(v, L_FC, v.act[4], B, st, pooled);
}

int decoder_fwd(Vae& v, int B, bool train, void* st, bool pooled = false, bool skip_out = false, bool skip_in = false) {
    if (!skip_in) {   // decoder_input (:41-45) has no BN / activation: the GEMM stores its result in the dtype the next layer reads
        const Layer& L = v.L[L_DI];
        RC(favae_conv_igemm(v.zlat, v.Wf(L_DI), v.P + L.b_off, (float*)v.act[L_DI], B, 1, 1, L.Cin, 1, 1, L.Cout, 1, 1, 0, 0, 0,
                            v.dtype | (v.dtype == FAVAE_BF16 ? FAVAE_OUT_BF16 : 0), st));
    }
    RC(conv_bn_chain(v, 7, 11, v.act[L_DI], B, train, st, pooled));
    if (skip_out) return FAVAE_OK;
    // eval passes: final_layer.3 through the row kernel of the training step (the generic GEMM pads 3 output channels to a 32-wide
    // tile: 225 us per 512 images against ~100)   (FAVAE_FUSE_MASK bit 0x80000000: the generic GEMM, as before)
    if (!train && !pooled && v.dtype == FAVAE_BF16 && !(v.fuse & 0x80000000u))
        return favae_conv_out_raw(v.act[11], v.wk_out, v.P + v.L[L_OUT].b_off, B, 64, 64, v.ch, v.raw[L_OUT], v.dtype, st);
    return conv_fwd(v, L_OUT, v.act[11], B, st, pooled);
}

int bn_bwd(Vae& v, int l, int B, void* st) {
    const Layer& L = v.L[l];
    if (l == 11 && v.dact11_recompute) {   // the gradient was not stored (dgrad): form it again from the folded output gradient and apply
        v.dact11_recompute = false;
        v.grad_fused[l] = false;
        const int G = B > 64 ? B / 64 : 1;
        const int dt = v.dtype | (v.raw_bf[l] ? FAVAE_RAW_BF16 : 0) | (v.dact_bf[l] ? FAVAE_OUT_BF16 : 0);
        const Layer& LO = v.L[L_OUT];
        int slabs = 0;
        // (blocks of the batch-64 launch = slabs of its bias-gradient sums; same-box sweep with room for 1024: 128 blocks 0.794 ms per step,
        // 192 0.785, 256 0.779, 320 0.780, 512 0.782, 1024 0.790)
        static const int dact11_slabs = getenv("FAVAE_DACT11_SLABS") ? std::min(DBP_SLABS, std::max(1, atoi(getenv("FAVAE_DACT11_SLABS")))) : DBP_SLABS;
        RC(favae_conv_folded_dgrad_bn(v.fold, v.Wf(L_OUT), v.draw[l], (long long)B * LO.Hin * LO.Hin, 9 * LO.Cout, LO.Cin, v.gbins[l],
                                      bins_for(v.rows(l, B > 64 ? 64 : B)), nullptr, v.raw[l], v.mean[l], v.invstd[l], v.P + L.g_off,
                                      B > 64 ? v.gab(l) : v.ab[l], G, v.G + L.g_off, v.G + L.be_off, B > 64 ? nullptr : v.dbp[l], dact11_slabs, &slabs, dt, st));
        v.dbias_slabs[l] = B > 64 ? 0 : slabs;
        return FAVAE_OK;
    }
    if (B > 64) {   // grouped-batch mode: the two sums per group from the epilogue of the GEMM that produced dact[l], else from a pass
        const int G = B / 64, nb = bins_for(v.rows(l, 64));
        const int dt = v.dtype | (v.raw_bf[l] ? FAVAE_RAW_BF16 : 0) | (v.dact_bf[l] ? FAVAE_DACT_BF16 : 0);
        if (!v.grad_fused[l])
            RC(favae_gbn_sums(v.dact[l], v.act[l], v.raw[l], v.mean[l], v.invstd[l], v.gab(l), v.gbins[l], nb, G, v.rows(l, 64), L.Cout, 2, dt, st));
        v.grad_fused[l] = false;
        v.dbias_slabs[l] = 0;
        // The bias of a convolution in front of a train-mode BatchNorm has an analytically zero gradient (the mean subtraction
        // removes it); the batch-64 step carries the column sums of draw, i.e. rounding noise, as the reference's autograd does.  Here
        // the gradient is left at its exact value, zero: 2048 blocks x C atomics onto C addresses cost up to 3.5x the apply itself
        // (enc.2 at G = 8: 60.7 us against 16.9 us for the forward apply over the same elements, tools/bench_gbn.py).
        return favae_gbn_bwd(v.dact[l], v.act[l], v.raw[l], v.mean[l], v.invstd[l], v.P + L.g_off, v.gab(l), v.gbins[l], nb, v.draw[l], v.G + L.g_off,
                             v.G + L.be_off, nullptr, G, v.rows(l, 64), L.Cout, dt, st);
    }
    if (v.grad_fused[l]) {   // the GEMM that produced dact[l] summed the two reductions in its epilogue
        v.grad_fused[l] = false;
        // the bias gradient (column sums of draw: rounding noise in front of a BatchNorm) is finished beside the main chain, by the
        // stream that takes the layer's weight gradient (FAVAE_FUSE_MASK bit 0x400000: inside this launch, as before)
        const int slabs = favae_bn_lrelu_bwd_slabs(v.rows(l, B), L.Cout);
        // (encoder.0: only when its weight gradient is the GEMM over the folded batch, which can carry the sums; a finishing
        // launch of its own would sit on the main chain when the tail of the step runs there)
        const bool defer = !(v.fuse & 0x400000u) && v.dbp[l] && slabs >= 1 && slabs <= DBP_SLABS && !(l == 0 && !v.x_folded);
        v.dbias_slabs[l] = defer ? slabs : 0;
        if (v.ab_live[l])
            return favae_bn_lrelu_bwd_bins_ab(v.dact[l], v.raw[l], v.mean[l], v.invstd[l], v.P + L.g_off, v.ab[l], v.gbins[l], bins_for(v.rows(l, B)),
                                              v.draw[l], v.G + L.g_off, v.G + L.be_off, v.G + L.b_off, v.rows(l, B), L.Cout,
                                              defer ? v.dbp[l] : v.scratch, defer ? (size_t)DBP_SLABS * L.Cout : v.scratch_doubles,
                                              v.dtype | (v.raw_bf[l] ? FAVAE_RAW_BF16 : 0) | (v.dact_bf[l] ? FAVAE_DACT_BF16 : 0) |
                                                  (defer ? FAVAE_DBIAS_DEFER : 0), st);
        return favae_bn_lrelu_bwd_bins(v.dact[l], v.act[l], v.raw[l], v.mean[l], v.invstd[l], v.P + L.g_off, v.gbins[l],
                                       bins_for(v.rows(l, B)), v.draw[l], v.G + L.g_off, v.G + L.be_off, v.G + L.b_off,
                                       v.rows(l, B), L.Cout, defer ? v.dbp[l] : v.scratch, defer ? (size_t)DBP_SLABS * L.Cout : v.scratch_doubles,
                                       v.dtype | (v.raw_bf[l] ? FAVAE_RAW_BF16 : 0) | (v.dact_bf[l] ? FAVAE_DACT_BF16 : 0) |
                                           (defer ? FAVAE_DBIAS_DEFER : 0), st);
    }
    return favae_bn_lrelu_bwd(v.dact[l], v.act[l], v.raw[l], v.mean[l], v.invstd[l], v.P + L.g_off, v.draw[l], v.G + L.g_off,
                              v.G + L.be_off, v.G + L.b_off, v.rows(l, B), L.Cout, v.scratch, v.scratch_doubles, v.s1s2,
                              v.dtype | (v.raw_bf[l] ? FAVAE_RAW_BF16 : 0) | (v.dact_bf[l] ? FAVAE_DACT_BF16 : 0), st);
}

int wgrad_launch(Vae& v, int l, const void* in, int B, void* st, bool in_raw11 = false);
int wgrad_launch_adam(Vae& v, int l, const void* in, int B, void* st, const Ada& a, int half, int* applied);
int adam_layer(Vae& v, int l, void* st, const Ada& a, int half);

// a job of side-stream calls: handed to the enqueue worker during a threaded epoch, run in place otherwise
template <class F> int side_do(Vae& v, F&& f) {
    if (v.wk) { v.wk->push(std::function<int()>(std::forward<F>(f))); return FAVAE_OK; }
    return f();
}

// weight gradient of layer l given draw[l] (just produced on the main stream) and the layer input; with overlap on it
// is queued on the side stream behind an event, so the main stream continues with the input gradient at once
// the deferred half of the BN-backward apply of layer l (see bn_bwd): on the stream that is about to take its weight gradient
int dbias_finish(Vae& v, int l, int slabs, void* st) {
    if (slabs <= 0) return FAVAE_OK;
    // layers whose weight gradient is an igemm_pm launch on this thread, next: the sums ride with that launch (a launch of its own
    // in front of every weight gradient costs the step 1.2 %, as much as the shorter BN-backward launches bring)
    if ((l != 0 || v.x_folded) && l != L_OUT && v.L[l].kind != LINEAR) {
        return favae_bn_dbias_ride(v.dbp[l], slabs, v.L[l].Cout, v.G + v.L[l].b_off);
    }
    return favae_bn_dbias_finish(v.dbp[l], slabs, v.L[l].Cout, v.G + v.L[l].b_off, st);
}
int take_dbias_slabs(Vae& v, int l) { const int s = v.dbias_slabs[l]; v.dbias_slabs[l] = 0; return s; }

// Small layers that are neighbours in the parameter arena and in the backward pass share ONE update: decoder.2 .. final_layer.3
// (0.41 M parameters), fc | decoder_input (0.25 M) and encoder.1 .. encoder.3 (1.55 M).  Their per-layer Adam and re-layout
// launches were ~4.5 us each for a few hundred KB: launch floors, beside the main chain, where a microsecond of kernel time costs
// the step as much as on it (step 0.878 -> 0.862 ms; with the 4.7 M-parameter layers in a group it is slower: 0.871 - 0.877).
// group_of: index of the layer's group or -1; a group is updated behind the weight gradient of its member that comes LAST in the
// backward pass (= its first layer).
struct UpdGroup { int l0, l1; };
struct UpdGroups { UpdGroup g[6]; int n; };
const UpdGroups& upd_groups() {
    static const UpdGroups G = [] {
        UpdGroups r{{{9, L_OUT}, {L_FC, L_DI}, {1, 3}}, 3};
        const char* e = getenv("FAVAE_UPD_GROUPS");   // e.g. "8-12,4-7": ranges of layer indices (1 .. 12, encoder.0 stays on its own)
        if (e) {
            r.n = 0;
            int a = 0, b = 0, used = 0;
            while (r.n < 6 && sscanf(e, "%d-%d%n", &a, &b, &used) == 2 && a >= 1 && b >= a && b < NL) {
                r.g[r.n++] = UpdGroup{a, b};
                e += used;
                if (*e == ',') ++e; else break;
            }
        }
        return r;
    }();
    return G;
}
#define UPD_GROUPS (upd_groups().g)
inline int group_of(const Vae& v, int l) {
    if (v.fuse & 0x1000000u) return -1;   // (bit 0x1000000: every layer on its own, as before)
    for (int g = 0; g < upd_groups().n; ++g)
        if (l >= UPD_GROUPS[g].l0 && l <= UPD_GROUPS[g].l1) return g;
    return -1;
}
int adam_group(Vae& v, int g, void* st, const Ada& a, int half) {
    const int l0 = UPD_GROUPS[g].l0, l1 = UPD_GROUPS[g].l1;
    const long long lo = v.L[l0].w_off, hi = l1 + 1 < NL ? v.L[l1 + 1].w_off : v.nparam;
    if (a.epoch) RC(favae_adam_step_clear(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, a.lr, a.b1, a.b2, a.eps, a.step, st));
    else RC(favae_adam_step(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, a.lr, a.b1, a.b2, a.eps, a.step, st));
    const float* w[16]; void* wc[16]; void* wt[16]; int Co[16], taps[16], Ci[16];
    int k = 0;
    for (int l = l0; l <= l1; ++l) {
        const Layer& L = v.L[l];
        void* cc = L.wc_off >= 0 ? v.shp(half) + L.wc_off * v.esz : nullptr;
        void* tt = L.wt_off >= 0 ? v.shp(half) + L.wt_off * v.esz : nullptr;
        if (!cc && !tt) continue;
        w[k] = v.P + L.w_off; wc[k] = cc; wt[k] = tt; Co[k] = L.Cout; taps[k] = L.taps; Ci[k] = L.Cin;
        ++k;
    }
    if (k) RC(favae_weight_prep_batch(k, w, wc, wt, Co, taps, Ci, v.dtype, st));
    if (l1 == L_OUT) {
        const Layer& L = v.L[L_OUT];
        void* cc = L.wc_off >= 0 ? v.shp(half) + L.wc_off * v.esz : nullptr;
        RC(favae_conv_out_wprep(cc ? (const void*)cc : (const void*)(v.P + L.w_off), v.wk_out, v.ch, v.dtype, st));
    }
    return FAVAE_OK;
}

int wgrad(Vae& v, int l, const void* in, int B, void* st) {
    const int slabs = take_dbias_slabs(v, l);   // (by value into the job: the caller's thread may be a step ahead of the worker)
    const int ll_B = l == L_OUT ? v.ll_B : 0;
    const float ll_var = v.ll_var;
    double* const ll_out = v.ll_out;
    if (l == L_OUT) v.ll_B = 0;
    const bool run_l = v.run_defer[l];
    v.run_defer[l] = false;
    auto running = [&v, l, B](void* s) -> int {   // grouped-batch mode: the BatchNorm's running statistics, off the main chain (bn_fwd)
        const Layer& L = v.L[l];
        return favae_gbn_running(v.mean[l], v.invstd[l], B / 64, v.rows(l, 64), L.Cout, 1e-5f, 0.1f, v.BUF + L.rm_off, v.BUF + L.rv_off, s);
    };
    // (by value into the job as well: the worker may launch final_layer.3's weight gradient while the caller's thread is in the next step's forward pass)
    const bool in_raw11 = l == L_OUT && v.act11_raw;
    if (!v.overlap) { if (run_l) RC(running(st)); RC(dbias_finish(v, l, slabs, st)); return wgrad_launch(v, l, in, B, st, in_raw11); }
    const unsigned seq = v.step_no;
    RC(sp_record(v, SP_FORK + l, (hipStream_t)st, seq, true));   // rides with the input-gradient GEMM queued next
    const Ada ada = v.ada;
    const bool merged = v.merged; const int half = v.upd_half;
    return side_do(v, [&v, l, in, B, ada, merged, half, seq, slabs, ll_B, ll_var, ll_out, run_l, running, in_raw11]() -> int {
        RC(sp_wait(v, SP_FORK + l, v.side, seq));
        if (run_l) RC(running((void*)v.side));
        RC(dbias_finish(v, l, slabs, (void*)v.side));
        if (ll_B > 0) RC(favae_conv_out_ll_ride(v.llp, ll_B, 64, 64, v.ch, ll_var, ll_out, v.G + v.L[L_OUT].b_off));
        const int grp = merged ? group_of(v, l) : -1;
        // Opt-in (FAVAE_FUSE_MASK bit 0x40000000): for the large layers that are updated on their own (encoder.4, decoder.0, decoder.1:
        // 10.6 M of the 12.8 M parameters) the optimiser step and both re-layouts of the weights in the epilogue of the weight gradient
        // (igemm_pm<..., ADAM>: bit-identical; no gradient in memory, 28 instead of 44 bytes per weight, two launches less per layer); the
        // rest of the layer's slice (bias, BatchNorm affine) follows on the same stream and the optimiser stream only passes the
        // completion on.  Measured on MI355X (tools/ab_fuse.sh, same box, three rounds): 0.869 ms per step against 0.851 ms with the
        // three launches, 3.22 against 3.20 ms at 8 groups: the weight-gradient blocks then hold their CUs through 28 bytes per weight
        // of latency-bound traffic beside the main chain, while the separate update runs on the optimiser stream in the chain's gaps.
        if (ada.on && merged && grp < 0 && !(v.fuse & 0x20000u) && (v.fuse & 0x40000000u) && l != 0 && l != L_OUT && v.L[l].kind != LINEAR) {
            int applied = 0;
            RC(wgrad_launch_adam(v, l, in, B, (void*)v.side, ada, half, &applied));
            if (applied) {
                const Layer& L = v.L[l];
                const long long lo = L.b_off, hi = l + 1 < NL ? v.L[l + 1].w_off : v.nparam;
                if (ada.epoch) RC(favae_adam_step_clear(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, ada.lr, ada.b1, ada.b2, ada.eps, ada.step, (void*)v.side));
                else RC(favae_adam_step(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, ada.lr, ada.b1, ada.b2, ada.eps, ada.step, (void*)v.side));
                RC(sp_record(v, SP_WG + l, v.side, seq));
                return sp_wait(v, SP_WG + l, v.side2, seq);   // (the step-end join is taken on the optimiser stream)
            }
        } else {
            RC(wgrad_launch(v, l, in, B, (void*)v.side, in_raw11));
        }
        if (!ada.on) return FAVAE_OK;
        if (merged && (v.fuse & 0x20000u)) return adam_layer(v, l, (void*)v.side, ada, half);   // same stream: no further event
        if (grp >= 0 && l != UPD_GROUPS[grp].l0) return FAVAE_OK;   // updated with its group (the weight-gradient stream runs in order)
        RC(sp_record(v, SP_WG + l, v.side, seq));
        if (merged) {   // optimiser stream, one wait: the update writes the shadow half no queued GEMM reads
            RC(sp_wait(v, SP_WG + l, v.side2, seq));
            return grp >= 0 ? adam_group(v, grp, (void*)v.side2, ada, half) : adam_layer(v, l, (void*)v.side2, ada, half);
        }
        // Layer-wise optimiser step on the third stream.  The layer whose weight gradient was queued before this one may be
        // updated once that gradient is complete (ev_wg) and its input-gradient GEMM, the last reader of its weight shadows,
        // has run: this fork event was recorded after that GEMM was queued on the main stream.  Its BN / bias gradients
        // were complete before its own fork.
        if (v.pending >= 0) {
            RC(sp_wait(v, SP_FORK + l, v.side2, seq));
            RC(sp_wait(v, SP_WG + v.pending, v.side2, seq));
            RC(adam_layer(v, v.pending, (void*)v.side2, ada, half));
            v.pending = -1;
        }
        if (!(v.dg_all || l == 1)) v.pending = l;
        return FAVAE_OK;
    });
}

// after the input-gradient GEMM of layer l is queued on the main stream: layers that do not wait for the next fork
// (encoder.1, whose update would otherwise sit between the last weight gradient and the end of the step) get an event
int after_dgrad(Vae& v, int l, void* st) {
    if (!(v.overlap && v.ada.on && !v.merged && (v.dg_all || l == 1))) return FAVAE_OK;
    const unsigned seq = v.step_no;
    RC(sp_record(v, SP_DG + l, (hipStream_t)st, seq));
    const Ada ada = v.ada;
    const int half = v.upd_half;
    return side_do(v, [&v, l, ada, half, seq]() -> int {
        RC(sp_wait(v, SP_DG + l, v.side2, seq));
        RC(sp_wait(v, SP_WG + l, v.side2, seq));
        return adam_layer(v, l, (void*)v.side2, ada, half);
    });
}

// Adam over the arena slice of layer l (weights, bias, BN affine) + its dtype shadows
int adam_layer(Vae& v, int l, void* st, const Ada& a, int half) {
    const Layer& L = v.L[l];
    const long long lo = L.w_off, hi = l + 1 < NL ? v.L[l + 1].w_off : v.nparam;
    void* c = L.wc_off >= 0 ? v.shp(half) + L.wc_off * v.esz : nullptr;
    void* t = L.wt_off >= 0 ? v.shp(half) + L.wt_off * v.esz : nullptr;
    if ((v.fuse & 128u) && L.Cin % 4 == 0 && L.Cout % 4 == 0 && (c || t))   // one pass: update + dtype copy + transpose (never L_OUT: Cout <= 4)
        return favae_adam_wprep(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, L.Cout, L.taps, L.Cin, c, t, a.epoch ? 1 : 0,
                                a.lr, a.b1, a.b2, a.eps, a.step, v.dtype, st);
    if (c && !t && v.dtype == FAVAE_BF16 && !(v.fuse & 0x800u) && ((long long)L.Cout * L.taps * L.Cin) % 4 == 0)
        // encoder.0 (no input gradient, so no transposed shadow): its update ends the step, the bf16 copy rides along
        return favae_adam_step_copy(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, a.lr, a.b1, a.b2, a.eps, a.step,
                                    a.epoch ? 1 : 0, c, (long long)L.Cout * L.taps * L.Cin, st);
    if (a.epoch && v.wg_keeps[l] && !(v.fuse & 0x20000000u))   // (FAVAE_FUSE_MASK bit 0x20000000: zero the whole slice, as before)
        RC(favae_adam_step_clear_from(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, L.b_off - L.w_off, a.lr, a.b1, a.b2, a.eps, a.step, st));
    else if (a.epoch) RC(favae_adam_step_clear(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, a.lr, a.b1, a.b2, a.eps, a.step, st));
    else RC(favae_adam_step(v.P + lo, v.G + lo, v.M + lo, v.V + lo, hi - lo, a.lr, a.b1, a.b2, a.eps, a.step, st));
    if (c || t) RC(favae_weight_prep(v.P + L.w_off, c, t, L.Cout, L.taps, L.Cin, v.dtype, st));
    if (l == L_OUT) RC(favae_conv_out_wprep(c ? (const void*)c : (const void*)(v.P + L.w_off), v.wk_out, v.ch, v.dtype, st));
    return FAVAE_OK;
}

// the weight gradient of a 3x3 convolution layer with the Adam step of its weights in the epilogue where the launch allows it
// (*applied), else exactly wgrad_launch
int wgrad_launch_adam(Vae& v, int l, const void* in, int B, void* st, const Ada& a, int half, int* applied) {
    const Layer& L = v.L[l];
    void* c = L.wc_off >= 0 ? v.shp(half) + L.wc_off * v.esz : nullptr;
    void* t = L.wt_off >= 0 ? v.shp(half) + L.wt_off * v.esz : nullptr;
    float *dw = v.G + L.w_off, *p = v.P + L.w_off, *m = v.M + L.w_off, *vv = v.V + L.w_off;
    if (L.kind == CONVT_S2)
        return favae_conv_wgrad_adam(v.draw[l], L.Cout, 1, in, L.Cin, 0, dw, p, m, vv, c, t, B, L.Hin, L.Hin, L.Hout, L.Hout, 3, 2, 1, -1, a.lr, a.b1,
                                     a.b2, a.eps, a.step, v.dtype, applied, st);
    return favae_conv_wgrad_adam(v.draw[l], L.Cout, 0, in, L.Cin, 1, dw, p, m, vv, c, t, B, L.Hout, L.Hout, L.Hin, L.Hin, 3, L.kind == CONV_S1 ? 1 : 2,
                                 1, -1, a.lr, a.b1, a.b2, a.eps, a.step, v.dtype, applied, st);
}

int wgrad_launch(Vae& v, int l, const void* in, int B, void* st, bool in_raw11) {
    const Layer& L = v.L[l];
    float* dw = v.G + L.w_off;
    if (l == 0 && v.x_folded) {   // the batch was folded at the start of the step (train_stages)
        // (256 slices of the 65536-pixel reduction, the automatic choice: 64 to 1024 measured the same or slower)
        return favae_conv_wgrad_prefolded(v.draw[0], L.Cout, L.Cout, v.fold0, 32, 9 * L.Cin, dw, (long long)B * L.Hout * L.Hout, -1, v.dtype, st);
    }
    if (l == L_OUT && v.out_folded && in_raw11)   // `in` (final_layer.0's activation) does not exist this step: its BatchNorm + LeakyReLU at the load
        return favae_conv_wgrad_prefolded_bn(v.fold, 32, 9 * L.Cout, v.raw[11], L.Cin, L.Cin, B > 64 ? v.gab(11) : v.ab[11],
                                             B > 64 ? (long long)64 * L.Hout * L.Hout : 0, dw, (long long)B * L.Hout * L.Hout, -1, v.dtype, st);
    if (l == L_OUT && v.out_folded)   // the output gradient was folded on the caller's stream (train_stages, stage 0): its input gradient reads it too
        return favae_conv_wgrad_prefolded(v.fold, 32, 9 * L.Cout, in, L.Cin, L.Cin, dw, (long long)B * L.Hout * L.Hout, -1, v.dtype, st);
    if ((v.fuse & 64u) && v.ch <= 3 && (l == 0 || l == L_OUT))   // image-sized channel count on one side: one GEMM over folded taps
        return favae_conv_wgrad_folded(v.draw[l], L.Cout, in, L.Cin, dw, B, L.Hout, L.Hout, L.Hin, L.Hin, l == 0 ? 2 : 1, -1, v.dtype,
                                       v.fold, v.fold_bytes, st);
    switch (L.kind) {
    case CONV_S2:
        return favae_conv_wgrad(v.draw[l], L.Cout, 0, in, L.Cin, 1, dw, B, L.Hout, L.Hout, L.Hin, L.Hin, 3, 2, 1, -1, v.dtype, st);
    case CONV_S1:
        return favae_conv_wgrad(v.draw[l], L.Cout, 0, in, L.Cin, 1, dw, B, L.Hout, L.Hout, L.Hin, L.Hin, 3, 1, 1, -1, v.dtype, st);
    case CONVT_S2:
        return favae_conv_wgrad(v.draw[l], L.Cout, 1, in, L.Cin, 0, dw, B, L.Hin, L.Hin, L.Hout, L.Hout, 3, 2, 1, -1, v.dtype, st);
    default:
        return favae_conv_wgrad(v.draw[l], L.Cout, 0, in, L.Cin, 0, dw, B, 1, 1, 1, 1, 1, 1, 0, -1, v.dtype, st);   // splitk < 0: the gradient arena is zero here
    }
}

// input gradient of layer l: draw[l] -> out (fp32, shaped like the layer input).  o >= 0: out is dact[o], the gradient
// w.r.t. the output of BN layer o; where the launch allows it the BN-backward reductions of layer o ride in its epilogue
int dgrad(Vae& v, int l, float* out, int B, void* st, bool out_zeroed, bool out_bf = false, int o = -1) {
    const Layer& L = v.L[l];
    const int sk = out_zeroed ? -1 : 0;
    const int dt = v.dtype | (out_bf ? FAVAE_OUT_BF16 : 0);
    if (l == L_OUT && v.out_folded && o == 11) {
        // final_layer.3 on the matrix cores over the folded output gradient (smallconv.hip, folded_dgrad_kernel); the sums of
        // final_layer.0's BatchNorm backward ride along, per BatchNorm group in the grouped-batch mode
        const int G = B > 64 ? B / 64 : 1;
        double* bins = B > 64 ? v.gbins[o] : ((v.fuse & 4u) ? gbins64(v, o) : nullptr);
        const float* ab = B > 64 ? v.gab(o) : (v.ab_live[o] ? v.ab[o] : nullptr);
        // (FAVAE_DACT11=1: the gradient is stored and read back by the BatchNorm backward, as before)
        static const bool dact11_keep = getenv("FAVAE_DACT11") && atoi(getenv("FAVAE_DACT11")) == 1;
        v.dact11_recompute = !dact11_keep && bins && ab && (B > 64 || (v.dbp[o] && !(v.fuse & 0x400000u)));
        RC(favae_conv_folded_dgrad(v.fold, v.Wf(L_OUT), v.dact11_recompute ? nullptr : out, (long long)B * L.Hin * L.Hin, 9 * L.Cout, L.Cin, bins,
                                   bins_for(v.rows(o, B > 64 ? 64 : B)), v.act[o], v.raw[o], v.mean[o], v.invstd[o], ab, G,
                                   dt | (v.raw_bf[o] ? FAVAE_RAW_BF16 : 0), st));
        v.grad_fused[o] = bins != nullptr;
        return FAVAE_OK;
    }
    if (o >= 0 && B > 64 && v.gbins[o] && L.kind != LINEAR) {   // grouped-batch mode: sums per BatchNorm group of layer o
        const int scatter = L.kind != CONVT_S2, stride = L.kind == CONV_S1 ? 1 : 2;
        int fused = 0;
        RC(favae_conv_igemm_red_groups(v.draw[l], v.Wt(l), nullptr, out, B, L.Hout, L.Hout, L.Cout, L.Hin, L.Hin, L.Cin, 3, stride, 1, scatter, sk,
                                       dt | (v.raw_bf[o] ? FAVAE_RAW_BF16 : 0), v.gbins[o], bins_for(v.rows(o, 64)), 2, v.act[o], v.raw[o], v.mean[o],
                                       v.invstd[o], v.gab(o), 64, &fused, st));
        v.grad_fused[o] = fused != 0;
        return FAVAE_OK;
    }
    if (o >= 0 && (v.fuse & (l == L_OUT ? 4u : 2u)) && gbins64(v, o) && L.kind != LINEAR) {
        const int scatter = L.kind != CONVT_S2, stride = L.kind == CONV_S1 ? 1 : 2;
        int fused = 0;
        RC(favae_conv_igemm_red(v.draw[l], v.Wt(l), nullptr, out, B, L.Hout, L.Hout, L.Cout, L.Hin, L.Hin, L.Cin, 3, stride, 1, scatter,
                                sk, dt | (v.raw_bf[o] ? FAVAE_RAW_BF16 : 0), v.gbins[o], bins_for(v.rows(o, B)), 2, v.act[o], v.raw[o],
                                v.mean[o], v.invstd[o], &fused, st));
        v.grad_fused[o] = fused != 0;
        return FAVAE_OK;
    }
    switch (L.kind) {
    case CONV_S2:   // scatter form with [Cin][tap][Cout]
        return favae_conv_igemm(v.draw[l], v.Wt(l), nullptr, out, B, L.Hout, L.Hout, L.Cout, L.Hin, L.Hin, L.Cin, 3, 2, 1, 1, sk, dt, st);
    case CONV_S1:
        return favae_conv_igemm(v.draw[l], v.Wt(l), nullptr, out, B, L.Hout, L.Hout, L.Cout, L.Hin, L.Hin, L.Cin, 3, 1, 1, 1, sk, dt, st);
    case CONVT_S2:  // gather form
        return favae_conv_igemm(v.draw[l], v.Wt(l), nullptr, out, B, L.Hout, L.Hout, L.Cout, L.Hin, L.Hin, L.Cin, 3, 2, 1, 0, sk, dt, st);
    default:
        return favae_conv_igemm(v.draw[l], v.Wt(l), nullptr, out, B, 1, 1, L.Cout, 1, 1, L.Cin, 1, 1, 0, 0, sk, dt, st);
    }
}

}  // namespace

struct favae_vae { Vae v; };

extern "C" long long favae_vae_param_count(int z, int channels) {
    Layer L[NL]; long long a, b, c;
    build_layers(z, channels, L, &a, &b, &c, true);
    return a;
}
extern "C" long long favae_vae_buffer_count(int z, int channels) {
    Layer L[NL]; long long a, b, c;
    build_layers(z, channels, L, &a, &b, &c, true);
    return b;
}
extern "C" long long favae_vae_shadow_bytes(int z, int channels, int dtype) {
    Layer L[NL]; long long a, b, c;
    build_layers(z, channels, L, &a, &b, &c, dtype == FAVAE_F32);
    const long long half = (c * (dtype == FAVAE_F32 ? 4 : 2) + 255) & ~255LL;
    return 2 * half + 256;   // two halves, see Vae::shp
}
extern "C" long long favae_vae_workspace_bytes(int z, int channels, int batch_max, int dtype) {
    Vae v{};
    v.z = z; v.ch = channels; v.Bmax = batch_max; v.dtype = dtype; v.esz = dtype == FAVAE_F32 ? 4 : 2;
    build_layers(z, channels, v.L, &v.nparam, &v.nbuf, &v.nshadow, dtype == FAVAE_F32);
    size_t need;
    carve(v, nullptr, 0, &need);
    return (long long)need;
}
extern "C" int favae_vae_num_layers(void) { return NL; }

/* info[0..9] = kind, Cin, Cout, taps, has_bn, w_off, b_off, gamma_off, beta_off, run_mean_off, run_var_off (11 values) */
extern "C" int favae_vae_layer_info(int z, int channels, int layer, long long* info) {
    FAVAE_CHECK_ARG(layer >= 0 && layer < NL && info, "favae_vae_layer_info: bad layer");
    Layer L[NL]; long long a, b, c;
    build_layers(z, channels, L, &a, &b, &c, true);
    const Layer& l = L[layer];
    info[0] = l.kind; info[1] = l.Cin; info[2] = l.Cout; info[3] = l.taps; info[4] = l.bn;
    info[5] = l.w_off; info[6] = l.b_off; info[7] = l.g_off; info[8] = l.be_off; info[9] = l.rm_off; info[10] = l.rv_off;
    return FAVAE_OK;
}

extern "C" int favae_vae_create(favae_vae** out, int z, int channels, int batch_max, int dtype, float* params, float* grads,
                                float* adam_m, float* adam_v, float* bn_buffers, void* shadow, size_t shadow_bytes,
                                void* workspace, size_t workspace_bytes) {
    FAVAE_CHECK_ARG(out && params && grads && adam_m && adam_v && bn_buffers && shadow && workspace, "favae_vae_create: null pointer");
    FAVAE_CHECK_ARG(z > 0 && z % 2 == 0 && channels > 0 && channels <= 4 && batch_max > 0 && batch_max <= 4096, "favae_vae_create: bad shape (z even, channels<=4)");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_vae_create: dtype");
    FAVAE_CHECK_ARG(((uintptr_t)params % 256) == 0 && ((uintptr_t)shadow % 256) == 0 && ((uintptr_t)workspace % 256) == 0 &&
                    ((uintptr_t)grads % 256) == 0, "favae_vae_create: buffers must be 256-byte aligned");
    favae_vae* h = new (std::nothrow) favae_vae();
    FAVAE_CHECK_ARG(h, "favae_vae_create: out of host memory");
    Vae& v = h->v;
    v.z = z; v.ch = channels; v.Bmax = batch_max; v.dtype = dtype; v.esz = dtype == FAVAE_F32 ? 4 : 2;
    build_layers(z, channels, v.L, &v.nparam, &v.nbuf, &v.nshadow, dtype == FAVAE_F32);
    v.P = params; v.G = grads; v.M = adam_m; v.V = adam_v; v.BUF = bn_buffers; v.SH = (char*)shadow;
    size_t need;
    carve(v, (char*)workspace, workspace_bytes, &need);
    v.sh_half = ((size_t)v.nshadow * v.esz + 255) & ~(size_t)255;
    v.sh_sel = 0; v.merged = false; v.upd_half = 0;
    if (need > workspace_bytes || 2 * v.sh_half > shadow_bytes) {
        delete h;
        favae_set_error("favae_vae_create: workspace %zu < %zu or shadow %zu < %zu", workspace_bytes, need, shadow_bytes, 2 * v.sh_half);
        return FAVAE_ERR_ARG;
    }
    v.host_words = nullptr;
    if (hipHostMalloc((void**)&v.host_words, 2 * sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && v.host_words) {
        v.host_words[0] = v.host_words[1] = 0u;
    } else {
        (void)hipGetLastError();
        v.host_words = nullptr;   // no stream words then: events
    }
    if (hipMemset(v.flags, 0, (SP_COUNT + 1) * sizeof(unsigned)) != hipSuccess) {
        if (v.host_words) (void)hipHostFree(v.host_words);
        delete h;
        favae_set_error("favae_vae_create: could not clear the dependency words");
        return FAVAE_ERR_HIP;
    }
    const char* e = getenv("FAVAE_SINGLE_STREAM");
    v.overlap = !(e && e[0] == '1');
    v.side = nullptr;
    v.side2 = nullptr;
    v.side3 = nullptr;
    v.ada = {false, 0.f, 0.f, 0.f, 0.f, 0, false};
    v.pending = -1;
    v.wk = nullptr;
    v.eval_fused = batch_max >= 256 && !(getenv("FAVAE_EVAL_FUSED") && getenv("FAVAE_EVAL_FUSED")[0] == '0');
    v.eval_ab_ready = false;
    v.fwd_folded = false;
    v.out_folded = false;
    {
        const char* hpe = getenv("FAVAE_BF16_EXACT_ENDS");
        v.hp = dtype == FAVAE_BF16 && !(hpe && hpe[0] == '0');
        v.act4_ready = false;
    }
    v.step_no = 0; v.join_seq = 0; v.vals = false; v.vals_ok = -1; v.mark_prev = v.mark_cur = 0;
    v.data_u8 = false;
    v.fi_on = false; v.fi_npend = 0; v.fi_flushes = 0; v.acts_wait = false;
    for (int l = 0; l < NL; ++l) v.act_deferred[l] = v.bn_published[l] = false;
    v.join_pending = false;
    v.dp_epoch = false;
    v.ev_join_ext = nullptr;
    v.join2 = false;
    v.rk_done = false;
    v.ev_fence_ok = false;
    // Layers whose weight-gradient GEMM runs unsplit at the largest batch of this executor (and then at every smaller one: the split
    // grows with the pixel count) overwrite their slice of the gradient arena in every step, so the optimiser step of an epoch
    // need not zero it (4 of the 32 bytes per weight it moves; encoder.4, decoder.0, decoder.1: 10.6 M of 12.8 M parameters).
    // Executors of the grouped-batch mode (Bmax > 64) keep the zeroing: their tail steps and group steps split differently.
    for (int l = 0; l < NL; ++l) v.ab_live[l] = false;
    v.act11_skip_ok = v.act11_raw = v.dact11_recompute = false;
    for (int l = 0; l < NL; ++l) v.run_defer[l] = false;
    for (int l = 0; l < NL; ++l) {
        const Layer& L = v.L[l];
        v.wg_keeps[l] = false;
        if (batch_max > 64 || l == 0 || l == L_OUT || (L.kind != CONV_S2 && L.kind != CONVT_S2)) continue;
        const int hs = L.kind == CONVT_S2 ? L.Hin : L.Hout;
        v.wg_keeps[l] = favae_conv_wgrad_auto_splitk(L.Cout, L.Cin, (long long)batch_max * hs * hs, 3) == 1;
    }
    v.x_folded = false;
    v.inline_tail = false;
    v.fold_sel = 0;
    {
        const char* d = getenv("FAVAE_DG_EVENTS");
        v.dg_all = d && d[0] == '1';
        const char* f = getenv("FAVAE_FUSE_MASK");
        v.fuse = f ? (unsigned)strtoul(f, nullptr, 0) : 0x17fu;   // 128 (fused Adam) measured slower in the step: its bandwidth bursts sit on the main chain   // (4, the sums on the thin-input path, was slower while the weight gradients beside it were issue-bound; since their addressing became lean it is 0.9 % faster)
    }
    if (v.overlap) {
        // the side streams themselves are created at the first training step (ensure_streams), next to the caller's stream
        bool ok = hipEventCreateWithFlags(&v.ev_comm, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_x, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_fi[0], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_fi[1], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_acts, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_batch, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_fold, hipEventDisableTiming) == hipSuccess;
        for (int l = 0; l < NL && ok; ++l) ok = hipEventCreateWithFlags(&v.ev_wg[l], hipEventDisableTiming) == hipSuccess;
        for (int l = 0; l < NL && ok; ++l) ok = hipEventCreateWithFlags(&v.ev_fork[l], hipEventDisableTiming) == hipSuccess;
        for (int l = 0; l < NL && ok; ++l) ok = hipEventCreateWithFlags(&v.ev_dg[l], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_join, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&v.ev_join2, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)favae_vae_destroy(h);   // null-safe: destroys the events that were created
            favae_set_error("favae_vae_create: could not create the side stream / events");
            return FAVAE_ERR_HIP;
        }
    }
    *out = h;
    return FAVAE_OK;
}

extern "C" int favae_vae_set_data_u8(favae_vae* h, int on) {
    FAVAE_CHECK_ARG(h, "favae_vae_set_data_u8: null handle");
    h->v.data_u8 = on != 0;
    return FAVAE_OK;
}

extern "C" int favae_vae_destroy(favae_vae* h) {
    if (!h) return FAVAE_OK;
    Vae& v = h->v;
    favae_slab_rider().partial = nullptr;   // (may point into this executor's workspace)
    favae_flag_carry().n = 0;
    auto kill = [](hipEvent_t& e) { if (e) { (void)hipEventDestroy(e); e = nullptr; } };   // handles are value-initialised to null
    if (v.ev_fence_ok) { kill(v.ev_fence[0]); kill(v.ev_fence[1]); }
    for (hipStream_t q : {v.side, v.side2, v.side3})
        if (q) { (void)hipStreamSynchronize(q); thin_in_release_stream(q); (void)hipStreamDestroy(q); }
    kill(v.ev_comm); kill(v.ev_x); kill(v.ev_batch); kill(v.ev_fold); kill(v.ev_fi[0]); kill(v.ev_fi[1]); kill(v.ev_acts);
    for (int l = 0; l < NL; ++l) { kill(v.ev_fork[l]); kill(v.ev_dg[l]); kill(v.ev_wg[l]); }
    kill(v.ev_join); kill(v.ev_join2);
    if (v.host_words) (void)hipHostFree(v.host_words);
    delete h;
    return FAVAE_OK;
}

/* Rebuild the dtype copies / transposes of every weight from the fp32 master arena (after Adam or load_state_dict). */
extern "C" int favae_vae_refresh_weights(favae_vae* h, void* st) {
    FAVAE_CHECK_ARG(h, "favae_vae_refresh_weights: null handle");
    Vae& v = h->v;
    const float* w[16]; void* wc[16]; void* wt[16]; int Co[16], taps[16], Ci[16];
    int n = 0;
    for (int l = 0; l < NL; ++l) {
        const Layer& L = v.L[l];
        void* c = L.wc_off >= 0 ? v.shp(v.sh_sel) + L.wc_off * v.esz : nullptr;
        void* t = L.wt_off >= 0 ? v.shp(v.sh_sel) + L.wt_off * v.esz : nullptr;
        if (!c && !t) continue;
        w[n] = v.P + L.w_off; wc[n] = c; wt[n] = t; Co[n] = L.Cout; taps[n] = L.taps; Ci[n] = L.Cin;
        ++n;
    }
    if (n) RC(favae_weight_prep_batch(n, w, wc, wt, Co, taps, Ci, v.dtype, st));
    return favae_conv_out_wprep(v.Wf(L_OUT), v.wk_out, v.ch, v.dtype, st);
}

namespace {
// Backward stages in execution order: 0 final conv, 1-5 final_layer.0 / dec.3..dec.0, 6 decoder_input, 7 fc_mu|fc_var,
// 8-12 enc.4..enc.0.  A training step = forward (with stage_lo == 0) + stages [stage_lo, stage_hi) + the join of the
// side streams (with stage_hi == N_STAGES).
constexpr int N_STAGES = 13;

// the caller's stream waits for the end of the previous step on the other streams
int wait_join(Vae& v, hipStream_t st) {
    if (v.ev_join_ext) {
        FAVAE_HIP(hipStreamWaitEvent(st, v.ev_join_ext, 0));
    } else {
        RC(sp_wait(v, SP_JOIN, st, v.join_seq));
        if (v.join2) RC(sp_wait(v, SP_JOIN2, st, v.join_seq));
    }
    return FAVAE_OK;
}

int train_stages(Vae& v, const float* data_nhwc, const int* idx, int B, const float* eps, const double* prior, float tau,
                 float beta, float var_x, double* metrics2_out, int stage_lo, int stage_hi, void* st) {
    const bool epoch = (v.overlap && v.ada.on && v.ada.epoch) || v.dp_epoch;
    if (v.overlap) RC(ensure_streams(v, (hipStream_t)st));
    favae_slab_rider().partial = nullptr;   // (a rider that an aborted call on this thread left behind must not reach another launch)
    if (stage_lo == 0) {
        favae_flag_carry().n = 0;           // (the same for dependency words: they may point into a workspace that is gone)
        const bool can_fold = v.overlap && (v.fuse & 64u) && !(v.fuse & 0x200u) && v.ch <= 3;
        // Single-device epochs: encoder.0's weight gradient and update, the last links of a step, stay on the caller's
        // stream (no event hop on that chain); what used to hide under them -- the zero-pool memset, the batch gather and
        // the tap gather of the NEXT step -- moves to the weight-gradient stream, which is idle at that point.
        v.inline_tail = epoch && can_fold && !(v.fuse & 0x2000u) && !(v.dp_epoch && (v.fuse & 0x200000u));   // (0x200000: not in data-parallel epochs)
        v.x_folded = false;
        v.fwd_folded = false;
        ++v.step_no;
        const unsigned seq = v.step_no;
        if (v.vals && v.wk) {   // bounded run-ahead: the worker has issued everything handed over before the previous step began
            RC(v.wk->wait_until(v.mark_prev));
            v.mark_prev = v.mark_cur;
            v.mark_cur = v.wk->mark();
        }
        // single-device epochs: a layer's update follows its weight gradient at once (optimiser stream, one event wait) and
        // writes the other shadow half (FAVAE_FUSE_MASK bit 0x10000: in place, behind the layer's input-gradient GEMM, i.e.
        // two more waits per layer; bit 0x20000: on the weight-gradient stream itself, no wait)
        // Opt-in (FAVAE_FUSE_MASK bit 0x40000; 0x80000: publish only): measured on MI355X the fused chain is SLOWER than the
        // launches it removes -- in a 3x3 convolution every activation is an operand of up to nine taps, so the load-side
        // BatchNorm + LeakyReLU is redone per tap and the fp32 source doubles the operand bytes: +6.6 us per consumer GEMM
        // (final_layer.0: +25 us) against 7.5-8 us per apply launch, and the last-block finalisation adds ~8 us per producer
        // (tools/bench_fi.py; step 1.06 -> 1.15 ms).  Kept as a tested option.
        v.fi_on = epoch && !v.dp_epoch && v.overlap && (v.fuse & 0x40000u) && B <= 64;
        v.fi_npend = 0; v.acts_wait = false;
        v.merged = epoch && !v.dp_epoch && v.ada.on && !(v.fuse & 0x10000u);
        v.upd_half = v.merged ? (v.sh_sel ^ 1) : v.sh_sel;
        if (v.inline_tail) {
            std::swap(v.xf, v.xf_alt); std::swap(v.xt, v.xt_alt);
            v.fold_sel ^= 1;
            v.fold0 = v.fold0_buf[v.fold_sel];   // the previous step's gradient GEMM may still read the other one
            const bool jp = v.join_pending;
            if (!jp) RC(sp_record(v, SP_X, (hipStream_t)st, seq));
            float* xf = v.xf; char* xt = v.xt; char* fold0 = v.fold0;
            RC(side_do(v, [&v, jp, xf, xt, fold0, data_nhwc, idx, B, seq]() -> int {
                // behind the previous step's last reader of the zero pool (BN backward of encoder.0), not behind its tail
                if (jp) RC(sp_wait(v, SP_FORK + 0, v.side, seq - 1)); else RC(sp_wait(v, SP_X, v.side, seq));
                FAVAE_HIP(hipMemsetAsync(v.zero_lo, 0, v.zero_bytes, v.side));
                RC(gather(v, data_nhwc, idx, B, xf, xt, (void*)v.side));
                RC(sp_record(v, SP_BATCH, v.side, seq));
                if (v.hp) RC(favae_conv_fold_taps_split(xf, fold0, v.fold0_bytes, B, 32, 32, 64, 64, v.ch, 2, (void*)v.side));
                else RC(favae_conv_fold_taps(xt, fold0, v.fold0_bytes, B, 32, 32, 64, 64, v.ch, 2, 0, v.dtype, (void*)v.side));
                return sp_record(v, SP_FOLD, v.side, seq);
            }));
            v.x_folded = true;
            // events: the worker has issued everything of the previous step and the job above, so ev_batch / ev_fold / ev_join are
            // recorded, and every event of the previous step has been consumed before this step records it again.  Values: the
            // waits below may be issued before the worker issues the writes; nothing to meet for.
            if (v.wk && !v.vals) RC(v.wk->drain());
            RC(sp_wait(v, SP_BATCH, (hipStream_t)st, seq));
            // bf16: encoder.0's forward pass reads the folded batch (FAVAE_FUSE_MASK bit 0x100000: the gathered batch, as before)
            v.fwd_folded = v.dtype == FAVAE_BF16 && v.L[0].Cout == 64 && !(v.fuse & 0x100000u) && !v.fi_on;
            if (v.fwd_folded) RC(sp_wait(v, SP_FOLD, (hipStream_t)st, seq));
        } else {
            if (!epoch) FAVAE_HIP(hipMemsetAsync(v.G, 0, (size_t)v.nparam * 4, (hipStream_t)st));
            FAVAE_HIP(hipMemsetAsync(v.zero_lo, 0, v.zero_bytes, (hipStream_t)st));
            if (epoch) { std::swap(v.xf, v.xf_alt); std::swap(v.xt, v.xt_alt); }
            RC(gather(v, data_nhwc, idx, B, v.xf, v.xt, st));
            if (can_fold) {
                // encoder.0's weight gradient ends the step: its tap gather needs the batch only, so it runs now, on the
                // weight-gradient stream (idle during the forward pass; in order behind the previous step's last reader of fold0)
                // bf16 with exact ends: encoder.0's forward pass reads the (split) folded batch, so the caller's stream waits for it
                const bool need_now = v.hp && v.L[0].Cout == 64 && !(v.fuse & 0x100000u) && !v.fi_on && (v.fuse & 1u) && v.sbins[0] && !v.raw_bf[0];
                RC(sp_record(v, SP_X, (hipStream_t)st, seq, !need_now));   // (rides with encoder.0's forward GEMM unless that waits for the fold)
                char* xt = v.xt; float* xf = v.xf; char* fold0 = v.fold0;
                RC(side_do(v, [&v, xt, xf, fold0, B, seq, need_now]() -> int {
                    RC(sp_wait(v, SP_X, v.side, seq));
                    if (v.hp) RC(favae_conv_fold_taps_split(xf, fold0, v.fold0_bytes, B, 32, 32, 64, 64, v.ch, 2, (void*)v.side));
                    else RC(favae_conv_fold_taps(xt, fold0, v.fold0_bytes, B, 32, 32, 64, 64, v.ch, 2, 0, v.dtype, (void*)v.side));
                    return need_now ? sp_record(v, SP_FOLD, v.side, seq) : FAVAE_OK;
                }));
                v.x_folded = true;
                if (need_now) {
                    if (v.wk && !v.vals) RC(v.wk->drain());
                    RC(sp_wait(v, SP_FOLD, (hipStream_t)st, seq));
                    v.fwd_folded = true;
                }
            }
            if (v.wk && !v.vals) RC(v.wk->drain());
        }
        if (v.join_pending) {   // the previous step's weight gradients / updates, still running next to the launches above
            RC(wait_join(v, (hipStream_t)st));
            v.join_pending = false;
        }
        RC(encoder_fwd(v, B, true, st, true));
        // reparameterise + KL + decoder_input in one launch (every block recomputes z: beyond 128 images the separate launches are faster,
        // 3.186 -> 3.158 ms per 512-image step)
        const bool decin = !(v.fuse & 0x4000u) && v.z <= 128 && (long long)B * v.z <= 12288 && B <= 128;
        if (decin)
            RC(favae_reparam_kl_decoder_input(v.raw[L_FC], eps, prior, idx, tau, B, v.z, v.zlat, v.stdq, metrics2_out + 1, v.Wf(L_DI),
                                              v.P + v.L[L_DI].b_off, v.act[L_DI], v.L[L_DI].Cout, v.dtype, st));
        else
            RC(favae_reparam_kl_fwd(v.raw[L_FC], eps, prior, idx, tau, B, v.z, v.zlat, v.stdq, metrics2_out + 1, v.dtype, st));
        const bool out_ll = (v.fuse & 0x100u) != 0;
        // (FAVAE_ACT11=1: final_layer.0's activation is written and read back, as before; the batch-64 BatchNorm backward without the sums
        // from the input-gradient launch, FAVAE_FUSE_MASK bit 4 off, reads the activation)
        static const bool act11_keep = getenv("FAVAE_ACT11") && atoi(getenv("FAVAE_ACT11")) == 1;
        v.act11_skip_ok = !act11_keep && out_ll && v.dtype == FAVAE_BF16 && !(v.fuse & 0x10000000u) && (B > 64 || ((v.fuse & 4u) && gbins64(v, 11))) && (v.fuse & 64u) && v.ch <= 3 && !(v.fuse & 0x8000000u) && v.L[11].Cout == 64;
        RC(decoder_fwd(v, B, true, st, true, out_ll, decin));
        v.act11_skip_ok = false;
        const long long D = (long long)64 * 64 * v.ch;
        if (out_ll) {   // final conv + sigmoid + log-likelihood + gradient + bias gradient in one launch
            // epochs: the loss term and the bias gradient are needed by nobody on the main chain; their sums over the launch's
            // blocks are left to the layer's weight gradient (folded form: one igemm_pm launch) on the stream beside it
            // (FAVAE_FUSE_MASK bit 0x4000000: inside this launch, as before)
            const bool defer = epoch && v.overlap && (v.fuse & 64u) && v.ch <= 3 && !(v.fuse & 0x4000000u);
            if (defer) { v.ll_B = B; v.ll_var = var_x; v.ll_out = metrics2_out; }
            if (v.act11_raw)
                RC(favae_conv_out_ll_bn(v.raw[11], B > 64 ? v.gab(11) : v.ab[11], B > 64 ? 64 : 0, v.wk_out, v.P + v.L[L_OUT].b_off, v.xf, B, 64, 64, v.ch,
                                        var_x, v.raw[L_OUT], v.draw[L_OUT], metrics2_out, v.G + v.L[L_OUT].b_off, defer ? v.llp : v.scratch,
                                        defer ? (size_t)((B * 64 * 64 + 255) / 256) * 5 : v.scratch_doubles, v.dtype | (defer ? FAVAE_DBIAS_DEFER : 0), st));
            else
                RC(favae_conv_out_ll(v.act[11], v.wk_out, v.P + v.L[L_OUT].b_off, v.xf, B, 64, 64, v.ch, var_x, v.raw[L_OUT], v.draw[L_OUT],
                                     metrics2_out, v.G + v.L[L_OUT].b_off, defer ? v.llp : v.scratch,
                                     defer ? (size_t)((B * 64 * 64 + 255) / 256) * 5 : v.scratch_doubles, v.dtype | (defer ? FAVAE_DBIAS_DEFER : 0), st));
        } else if (v.fuse & 8u) {
            RC(favae_sigmoid_ll_bias(v.raw[L_OUT], v.xf, B, D, v.ch, var_x, v.draw[L_OUT], metrics2_out, v.G + v.L[L_OUT].b_off, v.scratch,
                                     v.scratch_doubles, v.dtype, st));
        } else {
            RC(favae_sigmoid_ll(v.raw[L_OUT], v.xf, B, D, var_x, v.draw[L_OUT], metrics2_out, v.scratch, v.scratch_doubles, v.dtype, st));
            RC(favae_colsum(v.draw[L_OUT], v.rows(L_OUT, B), v.ch, v.G + v.L[L_OUT].b_off, v.scratch, v.scratch_doubles, v.dtype, st));
        }
    }
    for (int sg = stage_lo; sg < stage_hi; ++sg) {
        if (sg == 0 && v.acts_wait) {   // the backward pass reads activations that were materialised on the weight-gradient stream
            if (v.wk && !v.vals) RC(v.wk->drain());
            RC(sp_wait(v, SP_ACTS, (hipStream_t)st, v.step_no));
            v.acts_wait = false;
        }
        if (sg == 0) {
            // bf16, image-sized channel count: the output gradient of final_layer.3 is folded once, here, for both its weight gradient
            // (weight-gradient stream) and its input gradient (next launch)   (FAVAE_FUSE_MASK bit 0x8000000: the thin-input kernel, as before)
            v.out_folded = v.dtype == FAVAE_BF16 && (v.fuse & 64u) && v.ch <= 3 && !(v.fuse & 0x8000000u) && v.L[11].Cout == 64;
            v.dact11_recompute = false;   // (set by this step's dgrad below; a step that ended early must not leave it behind)
            if (v.out_folded)
                RC(favae_conv_fold_taps(v.draw[L_OUT], v.fold, v.fold_bytes, B, 64, 64, 64, 64, v.ch, 1, 1, v.dtype, st));
            RC(wgrad(v, L_OUT, v.act[11], B, st));
            RC(dgrad(v, L_OUT, v.dact[11], B, st, v.dact_pooled[11], v.dact_bf[11], 11));
            RC(sp_flush_carry((hipStream_t)st));
            RC(after_dgrad(v, L_OUT, st));
        } else if (sg <= 5) {
            const int l = 12 - sg;   // 11 .. 7
            RC(bn_bwd(v, l, B, st));
            RC(wgrad(v, l, v.act[l - 1], B, st));
            RC(dgrad(v, l, v.dact[l - 1], B, st, v.dact_pooled[l - 1], v.dact_bf[l - 1], l > 7 ? l - 1 : -1));
            RC(sp_flush_carry((hipStream_t)st));
            RC(after_dgrad(v, l, st));
        } else if (sg == 6) {
            v.rk_done = !(v.fuse & 0x8000u) && (v.fuse & 16u) && v.z <= 128 && (long long)B * 2 * v.z <= (v.z <= 32 ? 8192 : 16384);
            if (v.rk_done) {   // decoder_input backward (cast, bias gradient, dz) + reparameterisation / KL backward in one launch
                RC(favae_decoder_input_bwd_reparam(v.dact[L_DI], v.Wf(L_DI), v.draw[L_DI], v.G + v.L[L_DI].b_off, v.dz, v.raw[L_FC], eps, prior,
                                                   idx, v.stdq, tau, beta, B, v.z, v.L[L_DI].Cout, v.draw[L_FC], v.G + v.L[L_FC].b_off, v.dtype, st));
                RC(wgrad(v, L_DI, v.zlat, B, st));
                RC(after_dgrad(v, L_DI, st));
                continue;
            }
            if (v.fuse & 32u) {
                RC(favae_colsum_cast(v.dact[L_DI], v.draw[L_DI], B, 4096, v.G + v.L[L_DI].b_off, v.scratch, v.scratch_doubles, v.dtype, st));
            } else {
                RC(favae_cast(v.dact[L_DI], v.draw[L_DI], (long long)B * 4096, v.dtype, st));
                RC(favae_colsum(v.draw[L_DI], B, 4096, v.G + v.L[L_DI].b_off, v.scratch, v.scratch_doubles, v.dtype, st));
            }
            RC(wgrad(v, L_DI, v.zlat, B, st));
            RC(dgrad(v, L_DI, v.dz, B, st, true));
            RC(sp_flush_carry((hipStream_t)st));
            RC(after_dgrad(v, L_DI, st));
        } else if (sg == 7) {
            if (v.rk_done) {
                v.rk_done = false;
            } else if ((v.fuse & 16u) && 2 * v.z <= 256 && (long long)B * 2 * v.z <= 8192) {
                RC(favae_reparam_kl_bwd_bias(v.raw[L_FC], eps, prior, idx, v.stdq, v.dz, tau, beta, B, v.z, v.draw[L_FC],
                                             v.G + v.L[L_FC].b_off, v.dtype, st));
            } else {
                RC(favae_reparam_kl_bwd(v.raw[L_FC], eps, prior, idx, v.stdq, v.dz, tau, beta, B, v.z, v.draw[L_FC], v.dtype, st));
                RC(favae_colsum(v.draw[L_FC], B, 2 * v.z, v.G + v.L[L_FC].b_off, v.scratch, v.scratch_doubles, v.dtype, st));
            }
            RC(wgrad(v, L_FC, v.act[4], B, st));
            RC(dgrad(v, L_FC, v.dact[4], B, st, v.dact_pooled[4]));
            RC(sp_flush_carry((hipStream_t)st));
            RC(after_dgrad(v, L_FC, st));
        } else if (sg <= 11) {
            const int l = 12 - sg;   // 4 .. 1
            RC(bn_bwd(v, l, B, st));
            RC(wgrad(v, l, v.act[l - 1], B, st));
            RC(dgrad(v, l, v.dact[l - 1], B, st, v.dact_pooled[l - 1], v.dact_bf[l - 1], l - 1));
            RC(sp_flush_carry((hipStream_t)st));
            RC(after_dgrad(v, l, st));
        } else {
            RC(bn_bwd(v, 0, B, st));
            if (v.inline_tail) {
                RC(sp_record(v, SP_FORK + 0, (hipStream_t)st, v.step_no, true));   // the next step's memset / gather may start here (rides with the weight-gradient GEMM below)
                if (!v.fwd_folded) RC(sp_wait(v, SP_FOLD, (hipStream_t)st, v.step_no));
                RC(dbias_finish(v, 0, take_dbias_slabs(v, 0), st));   // (rides with the launch below)
                RC(wgrad_launch(v, 0, v.xt, B, st));
                RC(sp_flush_carry((hipStream_t)st));
            } else {
                RC(wgrad(v, 0, v.xt, B, st));
            }
        }
    }
    RC(sp_flush_carry((hipStream_t)st));   // (a point no launch of this call took along)
    if (stage_hi == N_STAGES && v.overlap && !v.dp_epoch) {   // (data-parallel epochs join through favae_vae_fence)
        if (v.ada.on && v.merged) {
            // every update has been queued behind its weight gradient on the weight-gradient stream; encoder.0's (no input
            // gradient reads its shadows) follows its gradient on the caller's stream when the tail of the step runs there
            const Ada ada = v.ada;
            const int half = v.upd_half;
            if (v.inline_tail) RC(adam_layer(v, 0, st, ada, half));
            v.join2 = false;
            // (the optimiser stream has waited for every weight gradient of the weight-gradient stream)
            const unsigned seq = v.step_no;
            v.join_seq = seq;
            RC(side_do(v, [&v, seq]() -> int { return sp_record(v, SP_JOIN, (v.fuse & 0x20000u) ? v.side : v.side2, seq); }));
            v.sh_sel = half;   // the next step reads the half this step's updates wrote
        } else if (v.ada.on) {   // encoder.0: no input gradient reads its shadows
            const Ada ada = v.ada;
            const int half = v.upd_half;
            const unsigned seq = v.step_no;
            v.join_seq = seq;
            RC(side_do(v, [&v, ada, half, seq]() -> int {
                if (v.pending >= 0 && v.pending != 0) {
                    RC(sp_wait(v, SP_WG + v.pending, v.side2, seq));
                    RC(adam_layer(v, v.pending, (void*)v.side2, ada, half));
                }
                v.pending = -1;
                return FAVAE_OK;
            }));
            // encoder.0's update follows its weight gradient on the same stream: it is the last link of the step, and an
            // event hop to the optimiser stream is a few microseconds of step time
            v.join2 = !v.inline_tail && !(v.fuse & 0x1000u);
            const bool join2 = v.join2;
            if (v.inline_tail) {
                RC(adam_layer(v, 0, st, ada, half));
            } else {
                RC(side_do(v, [&v, ada, join2, half, seq]() -> int {
                    if (!join2) RC(sp_wait(v, SP_WG + 0, v.side2, seq));
                    return adam_layer(v, 0, (void*)(join2 ? v.side : v.side2), ada, half);
                }));
            }
            // every other weight gradient of the side stream is awaited by the update of its layer on side2
            RC(side_do(v, [&v, join2, seq]() -> int {
                RC(sp_record(v, SP_JOIN, v.side2, seq));
                if (join2) RC(sp_record(v, SP_JOIN2, v.side, seq));
                return FAVAE_OK;
            }));
        } else {
            v.join2 = false;
            const unsigned seq = v.step_no;
            v.join_seq = seq;
            RC(side_do(v, [&v, seq]() -> int { return sp_record(v, SP_JOIN, v.side, seq); }));
        }
        if (epoch) v.join_pending = true;   // favae_vae_train_epoch joins before the next forward pass / at its end
        else { if (v.wk) RC(v.wk->drain()); RC(wait_join(v, (hipStream_t)st)); }
    }
    return FAVAE_OK;
}
}  // namespace

/* One forward + backward of the training step (cA/celeba_deepvanilla_vae.py:229-249): gradients land in the grads
 * arena; metrics2_out (device, 2 doubles) receives (reconstruction_error, kl_div).  No optimiser step. */
extern "C" int favae_vae_train_fwdbwd(favae_vae* h, const float* data_nhwc, const int* idx, int B, const float* eps,
                                      const double* prior, float tau, float beta, float var_x, double* metrics2_out, void* st) {
    FAVAE_CHECK_ARG(h && data_nhwc && eps && metrics2_out, "favae_vae_train_fwdbwd: null pointer");
    Vae& v = h->v;
    FAVAE_CHECK_ARG(B >= 1 && B <= v.Bmax, "favae_vae_train_fwdbwd: batch %d outside [1, %d]", B, v.Bmax);
    FAVAE_CHECK_ARG(B <= 64 || B % 64 == 0, "favae_vae_train_fwdbwd: a batch above 64 is a set of BatchNorm groups of 64 (got %d)", B);
    return train_stages(v, data_nhwc, idx, B, eps, prior, tau, beta, var_x, metrics2_out, 0, N_STAGES, st);
}

/* The same step in pieces, for data-parallel training that all-reduces gradient buckets while later layers are still in
 * their backward pass (RCCL over xGMI next to the kernels): stage_lo == 0 runs the forward pass first; backward stages
 * [stage_lo, stage_hi) follow (0 final conv, 1-5 final_layer.0 / dec.3..dec.0, 6 decoder_input, 7 fc_mu|fc_var,
 * 8-12 enc.4..enc.0; the gradients of a stage's layer are complete once favae_vae_fence has ordered a stream behind
 * it); stage_hi == 13 ends the step (the caller's stream then waits for the weight-gradient stream).  Call with the same
 * arguments and ascending, gap-free stage ranges. */
extern "C" int favae_vae_train_stages(favae_vae* h, const float* data_nhwc, const int* idx, int B, const float* eps,
                                      const double* prior, float tau, float beta, float var_x, double* metrics2_out,
                                      int stage_lo, int stage_hi, void* st) {
    FAVAE_CHECK_ARG(h && data_nhwc && eps && metrics2_out, "favae_vae_train_stages: null pointer");
    Vae& v = h->v;
    FAVAE_CHECK_ARG(B >= 1 && B <= v.Bmax, "favae_vae_train_stages: batch %d outside [1, %d]", B, v.Bmax);
    FAVAE_CHECK_ARG(B <= 64 || B % 64 == 0, "favae_vae_train_stages: a batch above 64 is a set of BatchNorm groups of 64 (got %d)", B);
    FAVAE_CHECK_ARG(stage_lo >= 0 && stage_lo <= stage_hi && stage_hi <= N_STAGES, "favae_vae_train_stages: stages [%d, %d)", stage_lo, stage_hi);
    return train_stages(v, data_nhwc, idx, B, eps, prior, tau, beta, var_x, metrics2_out, stage_lo, stage_hi, st);
}

/* Orders `waiting` behind everything queued so far on `st` (the stream the step is issued on) and on the internal
 * weight-gradient stream, without blocking either: the collective of a gradient bucket is enqueued on `waiting`. */
extern "C" int favae_vae_fence(favae_vae* h, void* st, void* waiting) {
    FAVAE_CHECK_ARG(h, "favae_vae_fence: null handle");   // (a null `waiting` is the default stream)
    Vae& v = h->v;
    if (!v.ev_fence_ok) {
        FAVAE_HIP(hipEventCreateWithFlags(&v.ev_fence[0], hipEventDisableTiming));
        FAVAE_HIP(hipEventCreateWithFlags(&v.ev_fence[1], hipEventDisableTiming));
        v.ev_fence_ok = true;
    }
    FAVAE_HIP(hipEventRecord(v.ev_fence[0], (hipStream_t)st));
    FAVAE_HIP(hipStreamWaitEvent((hipStream_t)waiting, v.ev_fence[0], 0));
    if (v.overlap && v.side) {
        FAVAE_HIP(hipEventRecord(v.ev_fence[1], v.side));
        FAVAE_HIP(hipStreamWaitEvent((hipStream_t)waiting, v.ev_fence[1], 0));
    }
    return FAVAE_OK;
}

/* Adam over the whole arena (optim.step(), :250) followed by the weight re-layout the next forward needs. */
extern "C" int favae_vae_adam(favae_vae* h, float lr, float beta1, float beta2, float eps, long long step, void* st) {
    FAVAE_CHECK_ARG(h, "favae_vae_adam: null handle");
    Vae& v = h->v;
    RC(favae_adam_step(v.P, v.G, v.M, v.V, v.nparam, lr, beta1, beta2, eps, step, st));
    return favae_vae_refresh_weights(h, st);
}

namespace {
// epochs of more than a few steps hand their side-stream calls to an enqueue worker (see struct Worker) and carry the
// dependencies between the streams as words in device memory (sp_record / sp_wait)
int epoch_worker_start(Vae& v, long long nsteps, hipStream_t st, unsigned need_beside) {
    const char* et = getenv("FAVAE_ENQUEUE_THREAD");   // "0": every call from the caller's thread (A/B and debugging)
    const bool threaded = !(et && et[0] == '0');
    v.vals = false;
    if (!(v.overlap && threaded && nsteps >= 4)) return FAVAE_OK;
    RC(ensure_streams(v, st));
    int dev = 0;
    FAVAE_HIP(hipGetDevice(&dev));
    Worker* wk = new (std::nothrow) Worker();
    if (!wk) return FAVAE_OK;
    wk->start(dev);
    v.wk = wk;
    // Under rocprofv3 the default is events: the stream memory operations are one-wave kernels that poll, and counter collection
    // runs one kernel at a time, so a wait queued ahead of its write would never see it (FAVAE_SYNC_VALUES=1 forces values).
    const char* sv = getenv("FAVAE_SYNC_VALUES");
    const char* pre = getenv("LD_PRELOAD");
    const bool profiled = getenv("ROCP_TOOL_LIBRARIES") || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") || (pre && strstr(pre, "rocprofiler"));
    if (sv ? sv[0] != '0' : !profiled) {
        if (v.vals_ok < 0) {
            v.vals_ok = hipStreamWriteValue32(st, v.flags + SP_COUNT, 1u, 0) == hipSuccess ? 1 : 0;
            if (!v.vals_ok) (void)hipGetLastError();
        }
        // A wait may be queued before the write it waits for, so the streams of this epoch must sit on hardware queues of their own:
        // a polling wait ahead of its writer in ONE queue would never end.  The probe of ensure_streams has measured exactly that,
        // pair by pair; if a pair this epoch uses shares a queue, the events stay.
        // ... and `beside` was measured against the caller's stream of the FIRST training step: a caller that comes back on
        // another stream (another hardware queue, possibly one of the side streams') gets events.
        v.vals = v.vals_ok == 1 && v.host_words && (v.beside & need_beside) == need_beside && st == v.probed_main;
        if (getenv("FAVAE_VERBOSE") && v.vals_ok == 1 && !v.vals)
            fprintf(stderr, "favae: stream words not used: streams of this epoch share a hardware queue (pairs 0x%x of 0x%x)\n", v.beside & need_beside, need_beside);
        v.mark_prev = v.mark_cur = wk->mark();
    }
    return FAVAE_OK;
}
// everything handed over has been issued; the worker ends with the epoch.  Returns the worker's status.
int epoch_worker_stop(Vae& v, long long nsteps) {
    Worker* wk = v.wk;
    if (!wk) return FAVAE_OK;
    const int wrc = wk->drain();
    wk->finish();
    if (getenv("FAVAE_VERBOSE"))
        fprintf(stderr, "favae: enqueue worker: %lld jobs, %.1f us busy per step; caller waited %.1f us per step for it; %lld dependency words launched on their own so far\n", wk->jobs,
                1e6 * wk->busy / (double)std::max(1LL, nsteps), 1e6 * wk->drain_wait / (double)std::max(1LL, nsteps), g_carry_flushes);
    v.wk = nullptr;
    delete wk;
    return wrc;
}
// a failed step may have left waits whose writes were never issued: satisfy every point, let the queues run dry, start over
// The release must not be ordered behind any stream of the executor, nor need a hardware queue at all: every stream may hold a
// wait that will never be satisfied (in data-parallel epochs side3 is the communication stream and waits on SP_FENCE_*; the
// caller's stream waits on SP_BATCH / SP_FOLD / SP_JOIN), a polling wave occupies its hardware queue, and the process has four of
// them: a memset on side3 (round 2) and a small copy on a fresh stream (first try of round 3: small copies are blit KERNELS) both
// ended behind such a wave -- tools/fail_inject.py hung in the hipDeviceSynchronize below.  So the host raises the abort word in
// pinned host memory, which every wait polls (wait_word_kernel), and only then synchronises.
void epoch_values_end(Vae& v, int rc) {
    favae_flag_carry().n = 0;   // words of an aborted step must not ride with a later launch of this thread
    if (v.vals && rc != FAVAE_OK && v.host_words) {
        __atomic_store_n(&v.host_words[0], 1u, __ATOMIC_SEQ_CST);
        (void)hipDeviceSynchronize();
        __atomic_store_n(&v.host_words[0], 0u, __ATOMIC_SEQ_CST);
        (void)hipMemset(v.flags, 0, SP_COUNT * sizeof(unsigned));
        (void)hipGetLastError();
        v.step_no = 0;
    }
    v.vals = false;
}
// a wait of an earlier call gave up (WAIT_MAX_TICKS): what ran behind it ran out of order.  Reported once; events from then on.
int check_wait_timeout(Vae& v) {
    if (v.host_words && __atomic_load_n(&v.host_words[1], __ATOMIC_SEQ_CST) != 0u) {
        (void)hipDeviceSynchronize();
        __atomic_store_n(&v.host_words[1], 0u, __ATOMIC_SEQ_CST);
        (void)hipMemset(v.flags, 0, SP_COUNT * sizeof(unsigned));
        v.step_no = 0;
        v.vals_ok = 0;
        favae_set_error("favae: a stream-word wait of a training epoch timed out (its writer never ran: streams sharing a hardware queue?); "
                        "the epoch's results are invalid; later epochs use events");
        return FAVAE_ERR_HIP;
    }
    return FAVAE_OK;
}
}  // namespace

// images of the step that starts with `rem` images left: `batch`, or what is left; in grouped-batch mode (batch > 64) the tail of
// an epoch is a step of its whole groups of 64 followed by a step of the rest
static int step_images(long long rem, int batch) {
    long long B = std::min<long long>(batch, rem);
    if (B > 64 && B % 64) B = B / 64 * 64;
    return (int)B;
}
/* Steps of an epoch over n images with `batch` images per step (rows of `metrics`): ceil(n / batch), plus one in grouped-batch mode
 * when the last batch is more than 64 images and not a whole number of groups. */
extern "C" long long favae_vae_epoch_steps(long long n, int batch) {
    if (n < 1 || batch < 1) return 0;
    long long s = n / batch;
    for (long long rem = n % batch; rem > 0; ++s) rem -= step_images(rem, batch);
    return s;
}

/* A whole epoch of steps on one device: perm [n] sample order, eps [n][z] in visiting order, metrics [favae_vae_epoch_steps][2].
 * Equivalent to the inner loop :227-255 with batch size `batch`; the last batch may be short.  batch > 64 (a multiple of 64) is the
 * grouped-batch mode: batch / 64 BatchNorm groups per optimiser step (include/favae_hip.h). */
extern "C" int favae_vae_train_epoch(favae_vae* h, const float* data_nhwc, const int* perm, long long n, int batch,
                                     const float* eps, const double* prior, float tau, float beta, float var_x, float lr,
                                     long long step0, double* metrics, void* st) {
    FAVAE_CHECK_ARG(h && perm && metrics && n >= 1 && batch >= 1, "favae_vae_train_epoch: bad argument");
    FAVAE_CHECK_ARG(batch <= 64 || batch % 64 == 0, "favae_vae_train_epoch: a batch above 64 is a whole number of BatchNorm groups of 64 (got %d)", batch);
    Vae& v = h->v;
    long long s = 0;
    RC(check_wait_timeout(v));
    if (v.overlap) FAVAE_HIP(hipMemsetAsync(v.G, 0, (size_t)v.nparam * 4, (hipStream_t)st));   // the steps keep it cleared
    RC(epoch_worker_start(v, (n + batch - 1) / batch, (hipStream_t)st, 1u | 2u | 8u));   // main, weight-gradient and optimiser stream
    int rc = FAVAE_OK;
    for (long long o = 0, Bs = 0; o < n && rc == FAVAE_OK; o += Bs, ++s) {
        const int B = step_images(n - o, batch);
        Bs = B;
        if (v.overlap) {   // optimiser step layer by layer behind the weight gradients (same arithmetic, see adam_layer)
            v.ada = {true, lr, 0.9f, 0.999f, 1e-8f, step0 + s + 1, true};
            rc = favae_vae_train_fwdbwd(h, data_nhwc, perm + o, B, eps + o * v.z, prior, tau, beta, var_x, metrics + 2 * s, st);
            v.ada.on = false;
            v.ada.epoch = false;
        } else {
            rc = favae_vae_train_fwdbwd(h, data_nhwc, perm + o, B, eps + o * v.z, prior, tau, beta, var_x, metrics + 2 * s, st);
            if (rc == FAVAE_OK) rc = favae_vae_adam(h, lr, 0.9f, 0.999f, 1e-8f, step0 + s + 1, st);
        }
    }
    {
        const int wrc = epoch_worker_stop(v, s);
        if (rc == FAVAE_OK) rc = wrc;
    }
    if (v.overlap && v.join_pending) {   // last step (or failure): the caller's stream waits for the side streams
        (void)wait_join(v, (hipStream_t)st);
        v.join_pending = false;
    }
    epoch_values_end(v, rc);
    v.pending = -1;
    v.merged = false;
    if (v.sh_sel != 0) {   // outside an epoch half 0 is the current one (a second executor may share the shadow buffer)
        v.sh_sel = 0;
        const int r2 = favae_vae_refresh_weights(h, st);
        if (rc == FAVAE_OK) rc = r2;
    }
    return rc;
}

/* Data-parallel epoch (see include/favae_hip.h). */
extern "C" int favae_vae_train_epoch_dp(favae_vae* h, favae_comm* c, const float* data_nhwc, const int* perm, long long n, int batch,
                                        const float* eps, const double* prior, float tau, float beta, float var_x, float lr,
                                        long long step0, double* metrics, void* st) {
    FAVAE_CHECK_ARG(h && c && perm && metrics && n >= 1 && batch >= 1, "favae_vae_train_epoch_dp: bad argument");
    FAVAE_CHECK_ARG(batch <= 64 || batch % 64 == 0, "favae_vae_train_epoch_dp: a batch above 64 is a whole number of BatchNorm groups of 64 (got %d)", batch);
    Vae& v = h->v;
    FAVAE_CHECK_ARG(v.overlap, "favae_vae_train_epoch_dp needs the weight-gradient stream (FAVAE_SINGLE_STREAM is set)");
    RC(check_wait_timeout(v));
    RC(ensure_streams(v, (hipStream_t)st));
    hipStream_t cs = v.side3;          // collectives + the bucket-wise optimiser step behind them
    hipEvent_t done = v.ev_comm;
    // (arena_lo, arena_hi, first layer, last layer, stage_hi): the slice is complete once backward stages < stage_hi ran
    // decoder (25.6 MB, travels under the whole encoder backward) | enc.4 + heads (19.6 MB, under enc.3 .. enc.0) | enc.3 (4.7 MB)
    // | enc.0-2 (1.6 MB: the only collective that is not hidden behind kernels of the same step)
    const long long w3 = v.L[3].w_off, w4 = v.L[4].w_off, w6 = v.L[6].w_off;
    constexpr int NBK = 4;
    const struct { long long lo, hi; int l0, l1, stage_hi; } bk[NBK] = {
        {w6, v.nparam, 6, NL - 1, 7}, {w4, w6, 4, 5, 9}, {w3, w4, 3, 3, 10}, {0, w3, 0, 2, N_STAGES}};
    FAVAE_HIP(hipMemsetAsync(v.G, 0, (size_t)v.nparam * 4, (hipStream_t)st));   // the bucket-wise Adam keeps it cleared
    // the weight gradients go through the enqueue worker here too (FAVAE_DP_THREAD=0: everything from the caller's thread)
    if (!v.ev_fence_ok) {
        FAVAE_HIP(hipEventCreateWithFlags(&v.ev_fence[0], hipEventDisableTiming));
        FAVAE_HIP(hipEventCreateWithFlags(&v.ev_fence[1], hipEventDisableTiming));
        v.ev_fence_ok = true;
    }
    const char* dpt = getenv("FAVAE_DP_THREAD");
    if (!(dpt && dpt[0] == '0')) RC(epoch_worker_start(v, (n + batch - 1) / batch, (hipStream_t)st, 1u | 4u | 16u));   // main, weight-gradient and communication stream
    int rc = FAVAE_OK;
    long long s = 0;
    v.dp_epoch = true;
    for (long long o = 0, Bs = 0; o < n && rc == FAVAE_OK; o += Bs, ++s) {
        const int B = step_images(n - o, batch);
        Bs = B;
        int lo = 0;
        for (int b = 0; b < NBK && rc == FAVAE_OK; ++b) {
            rc = train_stages(v, data_nhwc, perm + o, B, eps + o * v.z, prior, tau, beta, var_x, metrics + 2 * s, lo, bk[b].stage_hi, st);
            lo = bk[b].stage_hi;
            if (rc) break;
            // cs waits for the main and the weight-gradient stream, neither waits
            if (v.wk) {
                // the weight gradients of these stages are issued by the worker: it passes the point of the weight-gradient stream
                // behind them; the caller's stream passes its own (a one-wave launch: no GEMM follows at once to take the word along)
                const unsigned seq = v.step_no;
                rc = sp_record(v, SP_FENCE_M + b, (hipStream_t)st, seq, true);
                if (rc == FAVAE_OK) rc = sp_flush_carry((hipStream_t)st);
                if (rc == FAVAE_OK) rc = side_do(v, [&v, b, seq]() -> int { return sp_record(v, SP_FENCE_S + b, v.side, seq); });
                if (rc == FAVAE_OK && !v.vals) rc = v.wk->drain();   // events: recorded before they are awaited
                if (rc == FAVAE_OK) rc = sp_wait(v, SP_FENCE_M + b, cs, seq);
                if (rc == FAVAE_OK) rc = sp_wait(v, SP_FENCE_S + b, cs, seq);
            } else {
                rc = favae_vae_fence(h, st, (void*)cs);
            }
            if (rc) break;
            // the SUM over the ranks; the optimiser step scales the gradient it reads by 1 / ranks (for 2, 4, 8 ranks bitwise the
            // average RCCL forms by pre-multiplying; an averaging all-reduce costs the same between ranks, but RCCL runs it as a
            // pass over the bucket even in a group of one, 5.5 % of the kernel time of the one-rank rehearsal)
            rc = favae_allreduce_f32(c, v.G + bk[b].lo, bk[b].hi - bk[b].lo, 0, (void*)cs);
            if (rc) break;
            {
                // weight slices of this bucket that every step's unsplit weight-gradient GEMM overwrites are not zeroed (see wg_keeps):
                // the leading one and the largest of the others
                long long k0 = 0, k1lo = 0, k1hi = 0;
                if (!(v.fuse & 0x20000000u))
                    for (int l = bk[b].l0; l <= bk[b].l1; ++l) {
                        if (!v.wg_keeps[l]) continue;
                        const long long lo_ = v.L[l].w_off - bk[b].lo, hi_ = v.L[l].b_off - bk[b].lo;
                        if (lo_ == 0) k0 = hi_;
                        else if (hi_ - lo_ > k1hi - k1lo) { k1lo = lo_; k1hi = hi_; }
                    }
                rc = favae_adam_step_clear_scaled_keep(v.P + bk[b].lo, v.G + bk[b].lo, v.M + bk[b].lo, v.V + bk[b].lo, bk[b].hi - bk[b].lo, lr, 0.9f,
                                                       0.999f, 1e-8f, step0 + s + 1, 1.f / (float)favae_comm_size(c), k0, k1lo, k1hi, (void*)cs);
            }
            if (rc) break;
            const float* w[16]; void* wc[16]; void* wt[16]; int Co[16], taps[16], Ci[16];
            int k = 0;
            for (int l = bk[b].l0; l <= bk[b].l1; ++l) {
                const Layer& L = v.L[l];
                void* cc = L.wc_off >= 0 ? v.shp(v.sh_sel) + L.wc_off * v.esz : nullptr;
                void* tt = L.wt_off >= 0 ? v.shp(v.sh_sel) + L.wt_off * v.esz : nullptr;
                if (!cc && !tt) continue;
                w[k] = v.P + L.w_off; wc[k] = cc; wt[k] = tt; Co[k] = L.Cout; taps[k] = L.taps; Ci[k] = L.Cin;
                ++k;
            }
            if (k) rc = favae_weight_prep_batch(k, w, wc, wt, Co, taps, Ci, v.dtype, (void*)cs);
            if (rc == FAVAE_OK && bk[b].l1 == NL - 1) rc = favae_conv_out_wprep(v.Wf(L_OUT), v.wk_out, v.ch, v.dtype, (void*)cs);
        }
        if (rc) break;
        // the next forward pass (and, through its fork events, the next weight gradients) waits for the updates
        if (hipEventRecord(done, cs) != hipSuccess) { favae_set_error("favae_vae_train_epoch_dp: hipEventRecord"); rc = FAVAE_ERR_HIP; break; }
        v.ev_join_ext = done;
        v.join_pending = true;
    }
    {
        const int wrc = epoch_worker_stop(v, s);
        if (rc == FAVAE_OK) rc = wrc;
    }
    v.dp_epoch = false;
    if (v.join_pending) {
        (void)wait_join(v, (hipStream_t)st);
        v.join_pending = false;
    }
    v.ev_join_ext = nullptr;
    epoch_values_end(v, rc);
    return rc;
}

/* Non-zero (with favae_last_error set) if a dependency wait of an earlier epoch on this executor gave up; call after the epoch's
 * results have been synchronised (the host mirror calls it when it reads the epoch metrics). */
extern "C" int favae_vae_poll_error(favae_vae* h) {
    FAVAE_CHECK_ARG(h, "favae_vae_poll_error: null handle");
    return check_wait_timeout(h->v);
}

/* Encoder over B samples (update_x, :331-358 when train=0; the encoder half of forward() when train=1):
 * X.mean / X.cov rows as float64. */
extern "C" int favae_vae_encode(favae_vae* h, const float* data_nhwc, const int* idx, int B, double* xmean, double* xcov,
                                int train, void* st) {
    FAVAE_CHECK_ARG(h && data_nhwc && xmean && xcov, "favae_vae_encode: null pointer");
    Vae& v = h->v;
    FAVAE_CHECK_ARG(B > 0 && B <= v.Bmax, "favae_vae_encode: batch");
    RC(gather(v, data_nhwc, idx, B, nullptr, v.xt, st));
    RC(encoder_fwd(v, B, train != 0, st));
    return favae_encode_head(v.raw[L_FC], B, v.z, xmean, xcov, st);
}

/* update_x over a whole resident data set in batches of `batch` (:346-355). */
extern "C" int favae_vae_encode_all(favae_vae* h, const float* data_nhwc, long long n, int batch, double* xmean, double* xcov,
                                    void* st) {
    FAVAE_CHECK_ARG(h && n > 0 && batch > 0, "favae_vae_encode_all: bad argument");
    Vae& v = h->v;
    const long long row = (long long)64 * 64 * v.ch;
    v.eval_ab_ready = false;   // the first batch derives the affine forms from the current running statistics, the rest reuse them
    int rc = FAVAE_OK;
    for (long long o = 0; o < n && rc == FAVAE_OK; o += batch) {
        const int B = (int)std::min<long long>(batch, n - o);
        const float* rows = v.data_u8 ? (const float*)((const unsigned char*)data_nhwc + o * row) : data_nhwc + o * row;
        rc = favae_vae_encode(h, rows, nullptr, B, xmean + o * v.z, xcov + o * v.z, 0, st);
        v.eval_ab_ready = v.eval_fused;
    }
    v.eval_ab_ready = false;
    return rc;
}

/* Decoder (reconstruction / decoder(), :94-99, 360-385): zin [B][z] fp32 -> out [B][64][64][ch] fp32 NHWC. */
extern "C" int favae_vae_decode(favae_vae* h, const float* zin, int B, float* out_nhwc, int train, void* st) {
    FAVAE_CHECK_ARG(h && zin && out_nhwc, "favae_vae_decode: null pointer");
    Vae& v = h->v;
    FAVAE_CHECK_ARG(B > 0 && B <= v.Bmax, "favae_vae_decode: batch");
    RC(favae_cast(zin, v.zlat, (long long)B * v.z, v.dtype, st));
    RC(decoder_fwd(v, B, train != 0, st));
    return favae_sigmoid(v.raw[L_OUT], out_nhwc, (long long)B * 64 * 64 * v.ch, st);
}

/* Eval decoder over n latent rows + the sums r2_score needs against the resident data (SSHIBA.update diagnostic,
 * cA/favae.py:839-840): acc3 += (sum (y-yhat)^2, sum y, sum y^2); nothing leaves the device. */
extern "C" int favae_vae_reconstruct_stats(favae_vae* h, const float* zin, const float* data_nhwc, long long n, int batch,
                                           double* acc3, void* st) {
    FAVAE_CHECK_ARG(h && zin && data_nhwc && acc3 && n > 0 && batch > 0, "favae_vae_reconstruct_stats: bad argument");
    Vae& v = h->v;
    const long long row = (long long)64 * 64 * v.ch;
    FAVAE_HIP(hipMemsetAsync(acc3, 0, 3 * sizeof(double), (hipStream_t)st));
    v.eval_ab_ready = false;
    int rc = FAVAE_OK;
    for (long long o = 0; o < n && rc == FAVAE_OK; o += batch) {
        const int B = (int)std::min<long long>(batch, n - o);
        rc = favae_cast(zin + o * v.z, v.zlat, (long long)B * v.z, v.dtype, st);
        if (rc == FAVAE_OK) rc = decoder_fwd(v, B, false, st);
        v.eval_ab_ready = v.eval_fused;
        if (rc == FAVAE_OK)
            rc = v.data_u8 ? favae_sigmoid_sse_u8(v.raw[L_OUT], (const unsigned char*)data_nhwc + o * row, (long long)B * row, acc3, v.scratch,
                                                  v.scratch_doubles, st)
                           : favae_sigmoid_sse(v.raw[L_OUT], data_nhwc + o * row, (long long)B * row, acc3, v.scratch, v.scratch_doubles, st);
    }
    v.eval_ab_ready = false;
    return rc;
}

/* Debug/parity access to the executor's internal buffers: which = 0 raw, 1 act, 2 dact, 3 draw. */
extern "C" int favae_vae_debug_ptr(favae_vae* h, int which, int layer, void** out) {
    FAVAE_CHECK_ARG(h && out && layer >= 0 && layer < NL, "favae_vae_debug_ptr: bad argument");
    Vae& v = h->v;
    *out = which == 0 ? (void*)v.raw[layer] : which == 1 ? (void*)v.act[layer] : which == 2 ? (void*)v.dact[layer] : (void*)v.draw[layer];
    return FAVAE_OK;
}

// Error reporting and version for libfavae_hip.so.
#include <stdarg.h>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include "common.h"
#include "favae_hip.h"

static thread_local char g_err[512] = "";
SlabRider& favae_slab_rider() {
    static thread_local SlabRider r = {nullptr, nullptr, 0, 0, nullptr, 0.0, 0.0, 0};
    return r;
}

FlagCarry& favae_flag_carry() {
    static thread_local FlagCarry c = {{nullptr, nullptr, nullptr, nullptr}, 0u, 0};
    return c;
}

extern "C" void favae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* favae_last_error(void) { return g_err; }
extern "C" int favae_version(void) { return 1; }

// FAVAE_SYNC=1 in the environment makes every launch synchronous (debugging aid; never set in production).
extern "C" int favae_debug_sync_enabled(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FAVAE_SYNC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

// ---- live timing of selected launches (measurement only) -----------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double work; };
struct Prof {
    int enabled = 0, stride = 1;
    long long counter[16] = {0};
    std::vector<ProfRec> recs;
    std::mutex mu;
} g_prof;
}  // namespace

extern "C" int favae_prof_begin(int cls, double work, void* stream) {
    if (!g_prof.enabled || cls < 0 || cls >= 16) return -1;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if ((g_prof.counter[cls]++ % g_prof.stride) != 0) return -1;
    ProfRec r; r.cls = cls; r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess) return -1;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return -1; }
    (void)hipEventRecord(r.a, (hipStream_t)stream);
    g_prof.recs.push_back(r);
    return (int)g_prof.recs.size() - 1;
}
extern "C" void favae_prof_end(int idx, void* stream) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (idx >= 0 && idx < (int)g_prof.recs.size()) (void)hipEventRecord(g_prof.recs[idx].b, (hipStream_t)stream);
}

/* Enable with a sampling stride (choose one coprime to the launches per step of the classes of interest), run, then
 * collect per class (launches, seconds, work).  Classes: 2/3 gather-form GEMM tiles 64x64 / 128x32, 6 scatter-form 64x64,
 * 8/9 element-wise loaders (gather / scatter), 10-12 weight-gradient tiles 32x128 / 128x32 / 64x64, 14 thin-input conv,
 * 15 narrow-channel weight gradient (work = algorithmic flops); 7 BatchNorm-backward apply, 13 Adam (work = algorithmic bytes). */
extern "C" int favae_prof_enable(int enabled, int stride) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.enabled = enabled;
    g_prof.stride = stride > 0 ? stride : 1;
    for (auto& c : g_prof.counter) c = 0;
    return FAVAE_OK;
}
extern "C" int favae_prof_collect(double* out /* [16][3] */) {
    FAVAE_CHECK_ARG(out, "favae_prof_collect: null");
    FAVAE_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof.mu);
    for (int i = 0; i < 48; ++i) out[i] = 0.0;
    for (auto& r : g_prof.recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            out[r.cls * 3 + 0] += 1.0;
            out[r.cls * 3 + 1] += ms * 1e-3;
            out[r.cls * 3 + 2] += r.work;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.recs.clear();
    return FAVAE_OK;
}

// Error plumbing for the C-ABI: every entry point returns an int status and
// leaves a human-readable message retrievable through lbc_last_error().
#include "lbc_common.hpp"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void lbc_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int lbc_check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lbc_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LBC_ELAUNCH;
    }
    return LBC_OK;
}

extern "C" const char* lbc_last_error(void) { return g_err; }

// ---- runtime options -----------------------------------------------------------------
#include <stdlib.h>
#include <mutex>
extern "C" char** environ;      // (POSIX; <unistd.h> declares it only under _GNU_SOURCE)
#include <string.h>
namespace {
const char* const kOptNames[kOptCount] = {
    "LBC_FORCE_CFG", "LBC_HALO_BLOCKS", "LBC_HEAD_NO_MFMA", "LBC_NO_SIDE_STREAM", "LBC_NO_GEMM256", "LBC_GEMM256_MIN_TILES", "LBC_GEMM256_CFG",
    "LBC_NO_BN_BWD_FUSE", "LBC_NO_HDMA", "LBC_HDMA_CFG", "LBC_NO_GLDS_PHASED", "LBC_HDMA_PERSIST_WGS", "LBC_WGRAD_TR2_MIN_WGS", "LBC_NO_C64P_PRE",
    "LBC_NO_BN_FOLD", "LBC_C64P_BM", "LBC_HDMAP_SPLIT", "LBC_HDMA_SMALL_BELOW"};
struct OptTable {
    long long v[kOptCount];
    OptTable()
    {
        for (int i = 0; i < kOptCount; ++i) {
            const char* e = getenv(kOptNames[i]);
            v[i] = (e && *e) ? atoll(e) : -1;
        }
        // an LBC_* variable that is no option (a switch removed in an earlier round, a typo) would be ignored silently: say so once.
        // (LBC_PROF_LAUNCHES is the launch profiler's, read where it reports; LBC_PIN_STAGING the Python loader's; LBC_ARCH / LBC_BUILD_* the build scripts', LBC_TEST_* / LBC_EMU_* the test suite's and the CPU emulator's.)
        for (char** ep = environ; ep && *ep; ++ep) {
            if (strncmp(*ep, "LBC_", 4)) continue;
            const char* eq = strchr(*ep, '=');
            const size_t n = eq ? (size_t)(eq - *ep) : strlen(*ep);
            bool known = (n == 17 && !strncmp(*ep, "LBC_PROF_LAUNCHES", n)) || (n == 15 && !strncmp(*ep, "LBC_PIN_STAGING", n)) || !strncmp(*ep, "LBC_ARCH", 8) || !strncmp(*ep, "LBC_BUILD_", 10) || !strncmp(*ep, "LBC_TEST_", 9) || !strncmp(*ep, "LBC_EMU_", 8);
            for (int i = 0; i < kOptCount && !known; ++i) known = strlen(kOptNames[i]) == n && !strncmp(*ep, kOptNames[i], n);
            if (!known) fprintf(stderr, "liblbc_hip: environment variable %.*s is not an option of this library (ignored); see include/lbc_hip.h, lbc_config_set\n", (int)n, *ep);
        }
    }
};
OptTable& opts() { static OptTable t; return t; }   // built on first use = library load time for every practical purpose
}  // namespace
long long lbc_opt(LbcOpt o) { return opts().v[o]; }
extern "C" int lbc_config_set(const char* name, long long value)
{
    LBC_REQUIRE(name, "config_set: null name");
    for (int i = 0; i < kOptCount; ++i)
        if (!strcmp(name, kOptNames[i])) { opts().v[i] = value; return LBC_OK; }
    lbc_set_error("config_set: unknown option %s", name);
    return LBC_EINVAL;
}
extern "C" long long lbc_config_get(const char* name)
{
    if (name)
        for (int i = 0; i < kOptCount; ++i)
            if (!strcmp(name, kOptNames[i])) return opts().v[i];
    return -1;
}

// ---- zero page ---------------------------------------------------------------------------
// One 256-byte page of zeros per device, keyed by the calling thread's CURRENT device: the launch entry points of this library run
// with the device of their tensors current (torch's and HIP's own convention for kernel launches; the one place that launches from a
// foreign device context, the network's side stream, switches first).  The table is filled under a lock.
int lbc_zero_page(const void** p)
{
    static void* pages[64] = {nullptr};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { lbc_set_error("zero_page: bad device"); return LBC_ELAUNCH; }
    std::lock_guard<std::mutex> lock(mu);
    if (!pages[dev]) {
        // first use per device (a warm-up call, outside any stream capture): a synchronous 256-byte allocation, never freed
        void* q = nullptr;
        if (hipMalloc(&q, 256) != hipSuccess || hipMemset(q, 0, 256) != hipSuccess) { lbc_set_error("zero_page: allocation failed"); return LBC_ELAUNCH; }
        pages[dev] = q;
    }
    *p = pages[dev];
    return LBC_OK;
}

// ---- launch profiler -----------------------------------------------------------------
#include <map>
#include <string>
#include <vector>
namespace {
struct ProfRec { const char* name; double flops, bytes; hipEvent_t e0, e1; };
bool g_prof = false;
std::vector<ProfRec> g_recs;
}  // namespace
bool lbc_prof_on() { return g_prof; }
void lbc_prof_begin(const char* name, double flops, double bytes, hipStream_t s)
{
    ProfRec r;
    r.name = name; r.flops = flops; r.bytes = bytes;
    (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void lbc_prof_end(hipStream_t s) { if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, s); }
// a byte count booked under a class of its own, no events (reported with 0 launches' worth of time): what part of a launch's algorithmic bytes
// an older definition did not count (bench.py keeps both traffic ratios on its line so that rounds stay comparable)
void lbc_prof_note(const char* name, double bytes)
{
    ProfRec r;
    r.name = name; r.flops = 0.0; r.bytes = bytes; r.e0 = nullptr; r.e1 = nullptr;
    // (in FRONT of the record of the launch scope that is open around the caller: lbc_prof_end closes g_recs.back())
    if (g_recs.empty()) g_recs.push_back(r); else g_recs.insert(g_recs.end() - 1, r);
}

extern "C" int lbc_profile_enable(int on)
{
    g_prof = on != 0;
    return 0;
}
// Writes one line per kernel class: "name count total_ms total_flops total_bytes\n"; returns bytes written.
extern "C" int lbc_profile_report(char* buf, int cap)
{
    struct Agg { long long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    std::vector<std::string> order;
    // LBC_PROF_LAUNCHES=<file>: additionally one line per launch, in launch order ("name ms flops bytes")
    const char* per_launch = getenv("LBC_PROF_LAUNCHES");
    FILE* pl = (per_launch && *per_launch) ? fopen(per_launch, "a") : nullptr;
    for (ProfRec& r : g_recs) {
        float ms = 0.f;
        if (r.e0) {
            (void)hipEventSynchronize(r.e1);
            (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        }
        if (!agg.count(r.name)) order.push_back(r.name);
        Agg& a = agg[r.name];
        a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
        if (pl) fprintf(pl, "%s %.6f %.6e %.6e\n", r.name, ms, r.flops, r.bytes);
        if (r.e0) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    }
    if (pl) fclose(pl);
    g_recs.clear();
    int off = 0;
    for (const std::string& k : order) {
        const Agg& a = agg[k];
        int w = snprintf(buf + off, cap > off ? (size_t)(cap - off) : 0, "%s %lld %.6f %.6e %.6e\n", k.c_str(), a.n, a.ms, a.flops, a.bytes);
        if (w < 0 || off + w >= cap) break;
        off += w;
    }
    return off;
}

// Per-kernel HIP-event timing and error strings for libcaptra_hip.so.
// Events are recorded on the stream the kernel is launched on, so the time is the kernel's own
// device time on that stream (bench.py's roofline object is computed from these numbers).
#include "common.h"

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Span {
    hipEvent_t start, stop;
};
struct Family {
    std::string name;
    std::vector<Span> open;   // recorded, not yet folded into total
    double total_ms = 0.0;
    long long launches = 0;
};
std::atomic<int> g_enabled{0};
std::mutex g_mu;
std::vector<Family> g_fams;
std::vector<hipEvent_t> g_pool;

int family_slot(const char *name) {
    for (size_t i = 0; i < g_fams.size(); ++i)
        if (g_fams[i].name == name) return (int)i;
    g_fams.emplace_back();
    g_fams.back().name = name;
    return (int)g_fams.size() - 1;
}
hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void fold(Family &f) {
    for (auto &s : f.open) {
        hipEventSynchronize(s.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess) {
            f.total_ms += ms;
            f.launches += 1;
        }
        g_pool.push_back(s.start);
        g_pool.push_back(s.stop);
    }
    f.open.clear();
}
}  // namespace

CaptraProfScope::CaptraProfScope(const char *name, hipStream_t s) : slot(-1), stream(s) {
    if (!g_enabled.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    slot = family_slot(name);
    Span sp;
    sp.start = get_event();
    sp.stop = get_event();
    hipEventRecord(sp.start, stream);
    g_fams[slot].open.push_back(sp);
}

CaptraProfScope::~CaptraProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEventRecord(g_fams[slot].open.back().stop, stream);
    if (g_fams[slot].open.size() >= 4096) fold(g_fams[slot]);
}

extern "C" {

void captra_prof_enable(int on) { g_enabled.store(on ? 1 : 0); }

void captra_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &f : g_fams) {
        fold(f);
        f.total_ms = 0.0;
        f.launches = 0;
    }
}

int captra_prof_read(const char *name, double *total_ms, long long *launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &f : g_fams) {
        if (f.name == name) {
            fold(f);
            if (total_ms) *total_ms = f.total_ms;
            if (launches) *launches = f.launches;
            return 0;
        }
    }
    if (total_ms) *total_ms = 0.0;
    if (launches) *launches = 0;
    return 1;
}

int captra_prof_names(char *buf, int buflen) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::string s;
    for (auto &f : g_fams) {
        if (!s.empty()) s += ",";
        s += f.name;
    }
    if (buf && buflen > 0) {
        int n = (int)s.size() < buflen - 1 ? (int)s.size() : buflen - 1;
        for (int i = 0; i < n; ++i) buf[i] = s[i];
        buf[n] = 0;
    }
    return (int)s.size();
}

const char *captra_error_string(int err) {
    if (err == -1) return "captra: invalid argument";
    if (err == -2) return "captra: unsupported size";
    return hipGetErrorString((hipError_t)err);
}

const char *captra_version(void) { return "captra_hip 0.1 (gfx950)"; }

}  // extern "C"


// ---- zeroing as a kernel (common.h: captra_zero_async) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void captra_zero_kernel(uint4 *p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

int captra_zero_async(void *p, size_t nbytes, hipStream_t stream) {
    if (nbytes == 0) return 0;
    if ((nbytes & 15) != 0 || (reinterpret_cast<uintptr_t>(p) & 15) != 0) return -1;
    const size_t n16 = nbytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    CAPTRA_LAUNCH("zero", captra_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint4 *>(p), n16);
    return captra_last_error();
}

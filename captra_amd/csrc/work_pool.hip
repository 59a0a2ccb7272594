// Work tickets for the persistent SA2 kernel (sa_wave_pipe_kernel; sa_wave_lds_kernel walks its centres statically, see there).
//
// A persistent kernel sized to the chip hands its tiles out STATICALLY when every workgroup takes tile
// blockIdx.x + i * gridDim.x.  That is only optimal when the kernel has the chip to itself: the track step runs its two
// networks, and the lanes of a batch, on separate streams, so a furthest-point-sampling launch (one workgroup per cloud,
// ~290 us) or another lane's SA kernel holds some CUs while this kernel's workgroups on those CUs wait or crawl — and the
// launch then ends when its slowest workgroup has walked its fixed share.  With a ticket counter the workgroups that do
// run take the tiles (first tile = blockIdx.x, every further one = gridDim.x + atomicAdd(ticket, 1)); results do not
// depend on which workgroup computes a tile.
//
// A slot is two words {next ticket, workgroups done}; the LAST workgroup to leave resets both, so a slot is ready for
// the next launch that is stream-ordered after this one without a memset node.  Slots are handed out per LAUNCH:
//   * eager launches walk a ring of EAGER_SLOTS (two launches share a slot only if EAGER_SLOTS launches apart — far
//     beyond what a stream keeps in flight);
//   * launches recorded into a hipGraph keep their slot for the life of the graph, so they take slots that are never
//     handed out again (a bump range of GRAPH_SLOTS); when that range is used up the launcher gets nullptr and the kernel
//     falls back to the static walk — slower under contention, same result.
#include "common.h"

#include <mutex>

namespace {
constexpr int EAGER_SLOTS = 16384, GRAPH_SLOTS = 49152;
struct Pool {
    unsigned *base = nullptr;       // (EAGER_SLOTS + GRAPH_SLOTS) x 2 words, zeroed once
    unsigned eager = 0, graph = 0;
    bool failed = false;
};
std::mutex g_mu;
Pool g_pools[128];
CAPTRA_KNOB int g_dynamic = 1;
}  // namespace

// experiment knob (not part of the reference boundary): 0 = static tile walk in the persistent kernels
extern "C" void captra_sa_set_dynamic_tiles(int on) { g_dynamic = on; }

unsigned *captra_work_slot(hipStream_t stream) {
    if (!g_dynamic) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 128) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    Pool &pool = g_pools[dev];
    if (pool.base == nullptr) {
        if (pool.failed || cap != hipStreamCaptureStatusNone) return nullptr;   // no allocation while a capture is open
        const size_t bytes = (size_t)(EAGER_SLOTS + GRAPH_SLOTS) * 2 * sizeof(unsigned);
        if (hipMalloc((void **)&pool.base, bytes) != hipSuccess || hipMemset(pool.base, 0, bytes) != hipSuccess) {
            (void)hipGetLastError();
            pool.base = nullptr;
            pool.failed = true;
            return nullptr;
        }
    }
    if (cap == hipStreamCaptureStatusNone) return pool.base + 2 * (size_t)(pool.eager++ % EAGER_SLOTS);
    if (pool.graph >= (unsigned)GRAPH_SLOTS) return nullptr;
    return pool.base + 2 * (size_t)(EAGER_SLOTS + pool.graph++);
}

// HIP (gfx950 / CDNA4) execution back end: kernel launch of the per-element
// bodies, and the cooperative primitives the step needs -- LDS-staged LSD radix
// sort, exclusive scan, ordered f64 reduction.  64-lane wavefronts throughout.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <mutex>
#include <string.h>
#include <math.h>
#include <stdexcept>
#include <string>
#include <stdint.h>
#include "psim_types.h"

namespace psim {
namespace be {

inline void check(hipError_t e, const char* what)
{
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define PSIM_HIP(x) ::psim::be::check((x), #x)

typedef hipStream_t Stream;

inline void* dev_alloc(size_t bytes)
{
    void* p = nullptr;
    PSIM_HIP(hipMalloc(&p, bytes ? bytes : 16));
    return p;
}
inline void dev_free(void* p) { if (p) (void)hipFree(p); }
inline void dev_memset(Stream st, void* p, int v, size_t bytes) { if (bytes) PSIM_HIP(hipMemsetAsync(p, v, bytes, st)); }
inline void copy_h2d(Stream st, void* d, const void* h, size_t bytes)
{
    if (bytes) PSIM_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
}
inline void copy_d2h(Stream st, void* h, const void* d, size_t bytes)
{
    if (bytes) PSIM_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
}
inline void copy_d2d(Stream st, void* d, const void* s, size_t bytes)
{
    if (bytes) PSIM_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, st));
}
inline void sync(Stream st) { PSIM_HIP(hipStreamSynchronize(st)); }
inline int compute_units()
{
    int dev = 0, cus = 0;
    PSIM_HIP(hipGetDevice(&dev));
    PSIM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    return cus > 0 ? cus : 1;
}
// a second, non-blocking stream + events: memory-bound stages overlap the FP64-bound walk
inline Stream stream_create()
{
    // highest priority: the short memory-bound kernels put here get wave slots as the long
    // FP64-bound walk frees them, instead of waiting behind its whole grid
    int least = 0, greatest = 0;
    PSIM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    PSIM_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
    return s;
}
inline void stream_destroy(Stream s) { if (s) (void)hipStreamDestroy(s); }
typedef hipEvent_t Event;
inline Event event_create()
{
    hipEvent_t e = nullptr;
    PSIM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return e;
}
inline void event_destroy(Event e) { if (e) (void)hipEventDestroy(e); }
inline void event_record(Event e, Stream s) { PSIM_HIP(hipEventRecord(e, s)); }
inline void stream_wait(Stream s, Event e) { PSIM_HIP(hipStreamWaitEvent(s, e, 0)); }
// one turn of a host-side polling loop: lets the core's other thread and the HIP runtime's
// own threads (whose lock an event query takes) have the cycles
inline void cpu_relax() { __builtin_ia32_pause(); }
// has everything submitted before the event's record finished? (no waiting)
inline bool event_done(Event e)
{
    hipError_t r = hipEventQuery(e);
    if (r == hipSuccess) return true;
    if (r != hipErrorNotReady) check(r, "hipEventQuery");
    return false;
}
// page-locked host memory: small device-to-host readbacks land here without a staging copy
inline void* host_alloc_pinned(size_t bytes)
{
    void* p = nullptr;
    PSIM_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    return p;
}
inline void host_free_pinned(void* p) { if (p) (void)hipHostFree(p); }

// ------------------------------------------------------------------ small device-to-host readbacks
// The step asks the device for a few words many times (counts that size the next
// launches).  A mailbox is a block of host memory the device can write: one small kernel
// copies up to two word ranges into it and then raises a sequence number, which the host
// polls.  That is shorter than a copy engine transfer followed by a stream wait.
struct Mailbox {
    static constexpr uint32_t kWords = 1024;     // payload words; the sequence word follows them
    volatile uint32_t* host = nullptr;
    uint32_t* dev = nullptr;
    uint32_t seq = 0;
};
inline void mailbox_create(Mailbox& m)
{
    void* h = nullptr;
    PSIM_HIP(hipHostMalloc(&h, (Mailbox::kWords + 16) * sizeof(uint32_t), hipHostMallocMapped));
    memset(h, 0, (Mailbox::kWords + 16) * sizeof(uint32_t));
    void* d = nullptr;
    PSIM_HIP(hipHostGetDevicePointer(&d, h, 0));
    m.host = (volatile uint32_t*)h;
    m.dev = (uint32_t*)d;
    m.seq = 0;
}
inline void mailbox_destroy(Mailbox& m)
{
    if (m.host) (void)hipHostFree((void*)m.host);
    m.host = nullptr;
    m.dev = nullptr;
}
// (256 lanes: a workgroup the other stream can place while the walk holds most wave slots)
__global__ void __launch_bounds__(256) mailbox_kernel(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb,
                                                      uint32_t* box, uint32_t seq)
{
    for (uint32_t t = threadIdx.x; t < na + nb; t += 256) box[t] = t < na ? a[t] : b[t - na];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(box + Mailbox::kWords, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// words of a (then of b, may be null / 0) -> dst; returns when they have arrived
inline void mailbox_fetch(Stream st, Mailbox& m, void* dst, const void* a, uint32_t na, const void* b, uint32_t nb)
{
    if (na + nb > Mailbox::kWords) throw std::runtime_error("mailbox_fetch: too many words");
    uint32_t seq = ++m.seq;
    if (seq == 0) seq = ++m.seq;
    hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(256), 0, st, (const uint32_t*)a, na, (const uint32_t*)b, nb, m.dev,
                       seq);
    PSIM_HIP(hipGetLastError());
    uint64_t polls = 0;
    while (m.host[Mailbox::kWords] != seq) {
        if (++polls == (1ull << 22)) {
            // long wait: let the runtime report a failed kernel instead of spinning for ever
            PSIM_HIP(hipStreamSynchronize(st));
            if (m.host[Mailbox::kWords] != seq) throw std::runtime_error("mailbox_fetch: readback did not arrive");
        }
        cpu_relax();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    memcpy(dst, (const void*)m.host, (size_t)(na + nb) * 4);
}

// One wave that does nothing for about `microseconds` (bounded by the constant-rate clock).
// The second stream starts with it when it is released together with the walk: work of that
// stream that reaches the dispatcher before the walk's resident waves are in place was seen
// to be left without wave slots for the whole walk (a 60 us kernel lasting 3.3 ms).
__global__ void __launch_bounds__(64) delay_kernel(uint32_t microseconds)
{
    const uint64_t t0 = wall_clock64();                 // 100 MHz
    const uint64_t ticks = (uint64_t)microseconds * 100u;
    for (int i = 0; i < 4096; ++i) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(64);
    }
}
inline void delay(Stream st, uint32_t microseconds)
{
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, st, microseconds);
    PSIM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ launch
// One lane per element, one wave per workgroup.  Workgroups of a single wave fit into whatever
// wave slots are free: under the walk, which holds 26 of a CU's 32 slots, a 256-lane workgroup
// needs four free slots at once and leaves the other two idle, and the side stream's kernels
// ran 1.75 times slower for it (16 Mi: its wait at the end of the walk 0.40 -> 0.18 ms).  No
// functor launched this way uses LDS or a barrier; wave-level operations are all they share.
// With the device to itself a launch of millions of lanes is dispatched faster in workgroups
// of four waves (7 M lanes: 260 against 140 us for the short kernels of the contact solver),
// so the caller says which situation it is in: wide_groups() = no walk in flight.  Only the
// stream that works beside the walk (narrow_stream()) is concerned: what is queued on the walk's
// own stream runs when the walk has ended, whenever it is submitted.
inline bool& wide_groups()
{
    static thread_local bool wide = false;
    return wide;
}
inline Stream& narrow_stream()
{
    static thread_local Stream s = nullptr;
    return s;
}
template <class F, int LANES>
__global__ void __launch_bounds__(LANES) psim_kernel(int64_t n, F f)
{
    int64_t i = (int64_t)blockIdx.x * LANES + threadIdx.x;
    if (i < n) f(i);
}

template <class F>
inline void launch(Stream st, int64_t n, const F& f)
{
    if (n <= 0) return;
    const int lanes = (wide_groups() || st != narrow_stream()) ? 256 : 64;
    int64_t blocks = (n + lanes - 1) / lanes;
    if (blocks > 0x7fffffffLL) throw std::runtime_error("launch: grid too large");
    if (lanes == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(psim_kernel<F, 256>), dim3((unsigned)blocks), dim3(256), 0, st, n, f);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(psim_kernel<F, 64>), dim3((unsigned)blocks), dim3(64), 0, st, n, f);
    PSIM_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ block helpers
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan over the block's threads (blockDim.x = 64 * nwaves <= 1024);
// wsum must hold nwaves + 1 words.  Returns exclusive prefix; *total = block sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wsum, uint32_t* total)
{
    int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = (blockDim.x + 63) >> 6;
    uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int i = 0; i < nw; ++i) { uint32_t t = wsum[i]; wsum[i] = run; run += t; }
        wsum[nw] = run;
    }
    __syncthreads();
    uint32_t r = inc - v + wsum[w];
    *total = wsum[nw];
    __syncthreads();
    return r;
}

constexpr uint32_t kLookSpinLimit = 1u << 26;   // polls before a waiting lane gives up (seconds)

// Word set by a scan or sort kernel whose look-back gave up waiting (it cannot, as
// every tile a workgroup waits for has already started; the bound only guarantees that
// the kernels end whatever happens).  One per device; the host reads it with every
// small readback.  A tile that gives up writes no keys (sort) or zeros (scan).
inline uint32_t* lookback_fault_word()
{
    static uint32_t* words[64] = {};
    static std::mutex guard;           // (contexts may be created and used from several host threads)
    int dev = 0;
    PSIM_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) throw std::runtime_error("lookback_fault_word: device index out of range");
    std::lock_guard<std::mutex> lock(guard);
    if (!words[dev]) {
        uint32_t* w = (uint32_t*)dev_alloc(64);
        PSIM_HIP(hipMemset(w, 0, 64));
        words[dev] = w;
    }
    return words[dev];
}

// ------------------------------------------------------------------ exclusive scan (u32)
constexpr int kScanItems = 16;
// Workgroups of 1024 lanes, or of 256 for scans submitted next to a walk that holds most wave
// slots (a 1024-lane workgroup needs four free slots on every SIMD of one CU at once).
constexpr int kScanThreadsSmall = 256;
constexpr int kScanTileSmall = kScanThreadsSmall * kScanItems;

// One launch for any length: each workgroup takes the next tile of 16 items per lane
// (numbered by an atomic counter), scans it, publishes the tile's sum and finds the
// sum of everything before it by looking back over its predecessors' status words
// (64 at a time, one per lane of the first wave): a predecessor shows either its own
// sum (keep looking) or the running total up to and including itself (stop).  The
// last workgroup to finish clears the status words, so `temp` only has to be zero
// before its first use.
constexpr uint64_t kScanLocal = 1ull << 62, kScanInclusive = 2ull << 62, kScanValue = (1ull << 62) - 1;

// What is scanned and what becomes of the result.  PlainScan: the words of `in`, the exclusive
// prefix written to `out`.  Another policy can compute its item from something else and use
// the prefix as the place of a compacted record (BandSelect below): one pass over the input
// where flag, scan and write would be three, each with its own array in between.
struct PlainScan {
    static constexpr bool kPlain = true;
    __device__ __forceinline__ uint32_t item(const uint32_t* in, uint32_t i) const { return in[i]; }
    __device__ __forceinline__ void put(uint32_t* out, uint32_t i, uint32_t, uint32_t prefix) const { out[i] = prefix; }
};
// The particles whose grid cell lies in the rows [row_lo, row_hi): their cells (relative to the
// band's first) and indices, in ascending particle order.
struct BandSelect {
    static constexpr bool kPlain = false;
    int32_t gw, row_lo, row_hi;
    uint32_t cell_base;
    uint32_t* key_out;
    uint32_t* perm_out;
    __device__ __forceinline__ uint32_t item(const uint32_t* cell, uint32_t i) const
    {
        const int32_t row = (int32_t)(cell[i] / (uint32_t)gw);
        return (row >= row_lo && row < row_hi) ? 1u : 0u;
    }
    __device__ __forceinline__ void put(uint32_t*, uint32_t i, uint32_t value, uint32_t prefix, const uint32_t* cell) const
    {
        if (value) { key_out[prefix] = cell[i] - cell_base; perm_out[prefix] = i; }
    }
};

template <int kScanThreads, class Op = PlainScan>
__global__ void __launch_bounds__(kScanThreads) scan_lookback_kernel(const uint32_t* in, uint32_t* out, uint32_t n,
                                                                     uint32_t ntiles, uint32_t* temp, uint32_t* total_out,
                                                                     uint32_t* fault, Op op = Op())
{
    constexpr int kScanTile = kScanThreads * kScanItems;
    __shared__ uint32_t wsum[17];
    __shared__ uint32_t s_tile, s_failed;
    __shared__ uint64_t s_before;
    uint64_t* status = reinterpret_cast<uint64_t*>(temp + 16);
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { s_tile = atomicAdd(&temp[0], 1u); s_failed = 0; s_before = 0; }
    __syncthreads();
    const uint32_t tile = s_tile;
    uint32_t base = tile * kScanTile + tid * kScanItems;   // n < 2^32 - 16: no wrap inside a valid tile
    uint32_t v[kScanItems];
    uint32_t sum = 0;
    const bool full = base + kScanItems <= n && base + kScanItems > base;
    if (full && Op::kPlain) {
        const uint4* p = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int i = 0; i < kScanItems / 4; ++i) {
            uint4 q = p[i];
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) v[i] = (base + i < n) ? op.item(in, base + i) : 0u;
    }
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) sum += v[i];
    uint32_t total;
    uint32_t ex = block_exclusive_scan(sum, wsum, &total);
    if (tid < 64) {
        if (lane == 0)
            __hip_atomic_store(status + tile, (tile == 0 ? kScanInclusive : kScanLocal) | (uint64_t)total,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t before = 0;
        bool ok = true;
        if (tile > 0) {
            int64_t first = (int64_t)tile - 1;        // lane l looks at tile first - l
            uint32_t polls = 0;
            for (;;) {
                int64_t t = first - lane;
                uint64_t sv = (t >= 0) ? __hip_atomic_load(status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : kScanInclusive;          // before tile 0: nothing
                uint32_t flag = (uint32_t)(sv >> 62);
                uint64_t ready = __ballot(flag != 0), incl = __ballot(flag == 2);
                if (incl != 0) {
                    int f = __ffsll((long long)incl) - 1;           // nearest predecessor with a running total
                    uint64_t upto = (f == 63) ? ~0ull : ((1ull << (f + 1)) - 1);
                    if ((ready & upto) == upto) {
                        uint64_t mine = (lane <= f) ? (sv & kScanValue) : 0ull;
                        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
                        before += mine;
                        break;
                    }
                } else if (ready == ~0ull) {
                    uint64_t mine = sv & kScanValue;
                    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
                    before += mine;
                    first -= 64;
                    continue;
                }
                if ((++polls & 63u) == 0 &&
                    (polls >= kLookSpinLimit ||
                     __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (lane == 0) {
            if (ok) {
                if (tile > 0)
                    __hip_atomic_store(status + tile, kScanInclusive | ((before + total) & kScanValue), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                s_before = before;
                if (tile == ntiles - 1 && total_out) *total_out = (uint32_t)(before + total);
            } else {
                __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_failed = 1;
                if (total_out) *total_out = 0u;
            }
        }
    }
    __syncthreads();
    const bool failed = s_failed != 0;
    ex += (uint32_t)s_before;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) { uint32_t t = v[i]; v[i] = failed ? 0u : ex; ex += t; }
    if constexpr (Op::kPlain) {
        if (full) {
            uint4* p = reinterpret_cast<uint4*>(out + base);
#pragma unroll
            for (int i = 0; i < kScanItems / 4; ++i) p[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < kScanItems; ++i) if (base + i < n) out[base + i] = v[i];
        }
    } else {
        // (v[i] now holds the prefix; the item is the difference to the next prefix)
#pragma unroll
        for (int i = 0; i < kScanItems; ++i)
            if (base + i < n && !failed) op.put(out, base + i, (i + 1 < kScanItems ? v[i + 1] : ex) - v[i], v[i], in);
    }
    // the last workgroup to get here clears the scratch for the next scan: by then every
    // workgroup has finished looking back, so no status word is read any more
    __syncthreads();
    if (tid == 0) s_tile = atomicAdd(&temp[1], 1u);
    __syncthreads();
    if (s_tile == ntiles - 1) {
        for (uint32_t i = tid; i < ntiles; i += kScanThreads)
            __hip_atomic_store(status + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&temp[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&temp[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// whole scan in one workgroup (n <= kScanSmall), tile after tile with a running total
constexpr uint32_t kScanSmall = 2u * kScanTileSmall;
template <int kScanThreads>
__global__ void __launch_bounds__(kScanThreads) scan_small_kernel(const uint32_t* in, uint32_t* out, uint32_t n,
                                                                  uint32_t* total_out)
{
    constexpr int kScanTile = kScanThreads * kScanItems;
    __shared__ uint32_t wsum[17];
    uint32_t before = 0;
    for (uint32_t tile0 = 0; tile0 < n; tile0 += kScanTile) {
        uint32_t base = tile0 + threadIdx.x * kScanItems;
        uint32_t v[kScanItems];
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { v[i] = (base + i < n) ? in[base + i] : 0u; s += v[i]; }
        uint32_t total;
        uint32_t ex = block_exclusive_scan(s, wsum, &total) + before;
#pragma unroll
        for (int i = 0; i < kScanItems; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
        before += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = before;
}

inline size_t scan_temp_words(uint64_t n) { return 16 + 2 * (size_t)((n + kScanTileSmall - 1) / kScanTileSmall) + 2; }

// out may alias in.  temp: scan_temp_words(n) words, ZERO before its first use (each scan
// leaves it zero again).  total_out: device word or null.
inline void exclusive_scan_u32(Stream st, const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* temp,
                               uint32_t* total_out, bool small_groups = false)
{
    if (n == 0) { if (total_out) dev_memset(st, total_out, 0, 4); return; }
    if (n > 0xfffffff0ull) throw std::runtime_error("exclusive_scan_u32: too many items");
    if (n <= kScanSmall) {
        if (small_groups)
            hipLaunchKernelGGL(scan_small_kernel<kScanThreadsSmall>, dim3(1), dim3(kScanThreadsSmall), 0, st, in, out,
                               (uint32_t)n, total_out);
        else
            hipLaunchKernelGGL(scan_small_kernel<1024>, dim3(1), dim3(1024), 0, st, in, out, (uint32_t)n, total_out);
        PSIM_HIP(hipGetLastError());
        return;
    }
    if (small_groups) {
        uint32_t nb = (uint32_t)((n + kScanTileSmall - 1) / kScanTileSmall);
        hipLaunchKernelGGL(scan_lookback_kernel<kScanThreadsSmall>, dim3(nb), dim3(kScanThreadsSmall), 0, st, in, out,
                           (uint32_t)n, nb, temp, total_out, lookback_fault_word());
    } else {
        const uint32_t tile = 1024u * kScanItems;
        uint32_t nb = (uint32_t)((n + tile - 1) / tile);
        hipLaunchKernelGGL(scan_lookback_kernel<1024>, dim3(nb), dim3(1024), 0, st, in, out, (uint32_t)n, nb, temp,
                           total_out, lookback_fault_word());
    }
    PSIM_HIP(hipGetLastError());
}

// The particles of a band of grid rows, compacted in one pass (see BandSelect): key_out / perm_out
// receive their relative cells and indices, *total_out their number.  n > kScanSmall.
inline void band_select(Stream st, const uint32_t* cell, uint64_t n, int32_t gw, int32_t row_lo, int32_t row_hi,
                        uint32_t cell_base, uint32_t* key_out, uint32_t* perm_out, uint32_t* temp, uint32_t* total_out,
                        bool small_groups)
{
    if (n > 0xfffffff0ull) throw std::runtime_error("band_select: too many items");
    BandSelect op{gw, row_lo, row_hi, cell_base, key_out, perm_out};
    if (small_groups) {
        uint32_t nb = (uint32_t)((n + kScanTileSmall - 1) / kScanTileSmall);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_lookback_kernel<kScanThreadsSmall, BandSelect>), dim3(nb), dim3(kScanThreadsSmall),
                           0, st, cell, (uint32_t*)nullptr, (uint32_t)n, nb, temp, total_out, lookback_fault_word(), op);
    } else {
        const uint32_t tile = 1024u * kScanItems;
        uint32_t nb = (uint32_t)((n + tile - 1) / tile);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(scan_lookback_kernel<1024, BandSelect>), dim3(nb), dim3(1024), 0, st, cell,
                           (uint32_t*)nullptr, (uint32_t)n, nb, temp, total_out, lookback_fault_word(), op);
    }
    PSIM_HIP(hipGetLastError());
}
constexpr bool kHasBandSelect = true;

// ------------------------------------------------------------------ radix sort (LSD, 8-bit digits)
// One launch per digit pass ("one sweep"): a first kernel counts, for every pass at
// once, how many keys carry each digit value (the counts do not depend on the order
// the earlier passes leave).  A pass kernel's 256-lane workgroup then takes the next
// 4096-key tile (tile numbers are handed out by an atomic counter, so every tile a
// workgroup can wait for has already started), ranks it stably (wave-level digit
// matching by ballot + per-wave LDS counters), publishes its per-digit counts and
// finds its global offsets by looking back over the tiles before it: a predecessor
// shows either its own count (keep looking) or the running total up to and including
// itself (stop).  The tile is reordered in LDS so the stores are contiguous per
// digit run.  Sorts of at most one tile run every pass inside one workgroup.
// keys per lane: 16 (4096-key tiles) for long inputs -- longer digit runs per tile, half as many
// tiles to look back over; 4 (1024-key tiles) for short ones, which then spread over more CUs
constexpr int kSortItemsLong = 16, kSortItemsShort = 4;
constexpr uint64_t kSortLongFrom = 1u << 20;
constexpr int kMaxSortPasses = 8;
constexpr int kLookBatch = 8;                 // predecessors a sort tile's look-back requests at once
constexpr uint32_t kLookValue = 0x3fffffffu;
constexpr uint32_t kLookLocal = 1u << 30;       // this tile's own count
constexpr uint32_t kLookInclusive = 2u << 30;   // count of this tile and every tile before it

struct SortShared {
    uint32_t cnt[4][256];
    uint32_t dstart[256];
    uint32_t wsum[8];
};

// Stable rank of the tile's items by digit.  On return rank[k] is item k's position
// among its wave's items of the same digit, s.cnt[w][d] the number of such items in
// the waves before w, s.dstart[d] where digit d starts in the tile; the return value
// is the tile's count of digit threadIdx.x.
template <class KeyT, int kSortItems>
__device__ __forceinline__ uint32_t tile_rank(SortShared& s, const KeyT (&key)[kSortItems], uint32_t tile_n, int shift,
                                              uint32_t (&rank)[kSortItems])
{
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 4 * 256; i += 256) (&s.cnt[0][0])[i] = 0;
    __syncthreads();
    const uint64_t lane_lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        uint32_t off = (uint32_t)(w * (64 * kSortItems) + k * 64 + lane);
        bool valid = off < tile_n;
        uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            bool one = (d >> bit) & 1u;
            uint64_t b = __ballot(one);
            peers &= one ? b : ~b;
        }
        if (!valid) peers = 1ull << lane;
        uint32_t r = (uint32_t)__popcll(peers & lane_lt);
        int leader = __ffsll((long long)peers) - 1;
        uint32_t old = 0;
        if (valid && lane == leader) {
            old = s.cnt[w][d];
            s.cnt[w][d] = old + (uint32_t)__popcll(peers);
        }
        old = __shfl(old, leader, 64);
        rank[k] = old + r;
    }
    __syncthreads();
    // per digit: exclusive prefix over the 4 waves, then over digits
    uint32_t run = 0;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) { uint32_t c = s.cnt[ww][tid]; s.cnt[ww][tid] = run; run += c; }
    uint32_t total;
    uint32_t ds = block_exclusive_scan(run, s.wsum, &total);
    s.dstart[tid] = ds;
    __syncthreads();
    return run;
}

template <class KeyT>
__global__ void __launch_bounds__(256) sort_digit_counts_kernel(const KeyT* keys, uint32_t n, int bit_lo, int passes,
                                                                uint32_t items_per_block, uint32_t* digit_counts)
{
    __shared__ uint32_t h[kMaxSortPasses][256];
    for (int p = 0; p < passes; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    uint64_t lo = (uint64_t)blockIdx.x * items_per_block;
    uint64_t hi = lo + items_per_block;
    if (hi > n) hi = n;
    // four independent loads in flight per lane before their counts are taken
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024) {
        KeyT k[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint64_t idx = i + 256u * u;
            ok[u] = idx < hi;
            k[u] = ok[u] ? keys[idx] : (KeyT)0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (ok[u])
                for (int p = 0; p < passes; ++p) atomicAdd(&h[p][(uint32_t)(k[u] >> (bit_lo + 8 * p)) & 255u], 1u);
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        uint32_t c = h[p][threadIdx.x];
        if (c) atomicAdd(&digit_counts[p * 256 + threadIdx.x], c);
    }
}

template <class KeyT, int kSortItems>
__global__ void __launch_bounds__(256) sort_pass_kernel(const KeyT* kin, const uint32_t* vin, KeyT* kout, uint32_t* vout,
                                                        uint32_t n, int shift, const uint32_t* digit_counts,
                                                        uint32_t* status, uint32_t* next_tile, uint32_t* fault)
{
    constexpr int kSortTile = 256 * kSortItems;
    __shared__ SortShared s;
    __shared__ KeyT skey[kSortTile];
    __shared__ uint32_t sval[kSortTile];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t s_tile, s_failed;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (tid == 0) { s_tile = atomicAdd(next_tile, 1u); s_failed = 0; }
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t tile_base = tile * kSortTile;
    if (tile_base >= n) return;            // cannot happen: the grid has exactly ceil(n / tile) workgroups
    const uint32_t tile_n = (n - tile_base < (uint32_t)kSortTile) ? (n - tile_base) : (uint32_t)kSortTile;
    KeyT key[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        uint32_t off = (uint32_t)(w * (64 * kSortItems) + k * 64 + lane);
        bool valid = off < tile_n;
        key[k] = valid ? kin[tile_base + off] : (KeyT)0;
        val[k] = (valid && vin) ? vin[tile_base + off] : 0u;
    }
    const uint32_t mine = tile_rank<KeyT, kSortItems>(s, key, tile_n, shift, rank);
    // publish this tile's count of digit `tid`, then look back
    uint32_t* my_status = status + (size_t)tile * 256 + tid;
    __hip_atomic_store(my_status, (tile == 0 ? kLookInclusive : kLookLocal) | mine, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    uint32_t before = 0;
    if (tile > 0) {
        bool ok = true;
        int64_t t = (int64_t)tile - 1;
        uint32_t polls = 0;
        // kLookBatch predecessors are requested at once and then taken in order: the tiles of
        // a launch publish their own counts at about the same time, so a tile far into the
        // grid walks back over many of them, and one memory round trip per predecessor was
        // most of a pass (a tile that is not there counts as "everything before it: nothing")
        for (bool found = false; !found;) {
            uint32_t v[kLookBatch];
#pragma unroll
            for (int u = 0; u < kLookBatch; ++u)
                v[u] = (t - u >= 0) ? __hip_atomic_load(status + (size_t)(t - u) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : kLookInclusive;
            int taken = 0;
#pragma unroll
            for (int u = 0; u < kLookBatch; ++u) {
                const uint32_t f = v[u] >> 30;
                if (found || f == 0 || taken != u) continue;
                before += v[u] & kLookValue;
                taken = u + 1;
                if (f == 2) found = true;
            }
            t -= taken;
            if (taken == 0) {
                if ((++polls & 255u) == 0 &&
                    (polls >= kLookSpinLimit ||
                     __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (ok) {
            __hip_atomic_store(my_status, kLookInclusive | ((before + mine) & kLookValue), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_failed = 1;
        }
    }
    // digits below `tid` over the whole input
    uint32_t total;
    uint32_t below = block_exclusive_scan(digit_counts[tid], s.wsum, &total);
    gbase[tid] = below + before;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        uint32_t off = (uint32_t)(w * (64 * kSortItems) + k * 64 + lane);
        if (off < tile_n) {
            uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
            uint32_t lp = s.dstart[d] + s.cnt[w][d] + rank[k];
            skey[lp] = key[k];
            sval[lp] = val[k];
        }
    }
    __syncthreads();
    if (s_failed) return;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        uint32_t pos = (uint32_t)(k * 256 + tid);
        if (pos < tile_n) {
            KeyT kk = skey[pos];
            uint32_t d = (uint32_t)(kk >> shift) & 255u;
            uint32_t g = gbase[d] + (pos - s.dstart[d]);
            if (g < n) {
                kout[g] = kk;
                if (vout) vout[g] = sval[pos];
            }
        }
    }
}

// n <= one tile: every pass in one workgroup, keys staged through LDS between passes
template <class KeyT, int kSortItems>
__global__ void __launch_bounds__(256) sort_one_tile_kernel(const KeyT* kin, const uint32_t* vin, KeyT* kout,
                                                            uint32_t* vout, uint32_t n, int bit_lo, int passes)
{
    constexpr int kSortTile = 256 * kSortItems;
    __shared__ SortShared s;
    __shared__ KeyT skey[kSortTile];
    __shared__ uint32_t sval[kSortTile];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    KeyT key[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        uint32_t off = (uint32_t)(w * (64 * kSortItems) + k * 64 + lane);
        bool valid = off < n;
        key[k] = valid ? kin[off] : (KeyT)0;
        val[k] = (valid && vin) ? vin[off] : 0u;
    }
    for (int p = 0; p < passes; ++p) {
        const int shift = bit_lo + 8 * p;
        (void)tile_rank<KeyT, kSortItems>(s, key, n, shift, rank);
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            uint32_t off = (uint32_t)(w * (64 * kSortItems) + k * 64 + lane);
            if (off < n) {
                uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
                uint32_t lp = s.dstart[d] + s.cnt[w][d] + rank[k];
                skey[lp] = key[k];
                sval[lp] = val[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            uint32_t off = (uint32_t)(w * (64 * kSortItems) + k * 64 + lane);
            if (off < n) { key[k] = skey[off]; val[k] = sval[off]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        uint32_t off = (uint32_t)(w * (64 * kSortItems) + k * 64 + lane);
        if (off < n) {
            kout[off] = key[k];
            if (vout) vout[off] = val[k];
        }
    }
}

// A pointer handed in by the caller for kernels to work on: device memory or page-locked host
// memory.  Anything else (an ordinary host pointer) would end in a GPU memory fault.
inline void require_device_memory(const void* p, const char* what)
{
    hipPointerAttribute_t a;
    const hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess || a.type == hipMemoryTypeUnregistered)
        throw std::invalid_argument(std::string("psim: ") + what +
                                    " must be device memory or page-locked host memory (the GPU cannot reach this pointer)");
}

// (self-test) the three kinds of memory a caller might hand in, classified without a kernel
// touching any of them
inline void selftest_pointer_kinds()
{
    std::vector<char> plain(4096);
    void* dev = nullptr;
    void* pinned = nullptr;
    PSIM_HIP(hipMalloc(&dev, 4096));
    PSIM_HIP(hipHostMalloc(&pinned, 4096, hipHostMallocDefault));
    bool refused = false;
    try { require_device_memory(plain.data() + 64, "selftest"); } catch (const std::invalid_argument&) { refused = true; }
    std::string bad;
    try {
        require_device_memory(static_cast<char*>(dev) + 64, "selftest: device memory");
        require_device_memory(static_cast<char*>(pinned) + 64, "selftest: page-locked memory");
    } catch (const std::invalid_argument& e) { bad = e.what(); }
    (void)hipFree(dev);
    (void)hipHostFree(pinned);
    if (!refused) throw std::runtime_error("selftest: an ordinary host pointer passed for memory the GPU can reach");
    if (!bad.empty()) throw std::runtime_error("selftest: refused: " + bad);
}

inline size_t sort_temp_words(uint64_t n)
{
    // enough for any input of at most n keys, whichever tiling it gets
    const uint64_t ts = 256u * kSortItemsShort, tl = 256u * kSortItemsLong;
    const uint64_t n_short = n < kSortLongFrom ? n : kSortLongFrom;
    const uint64_t a = (n_short + ts - 1) / ts, b = (n + tl - 1) / tl;
    const uint64_t ntiles = a > b ? a : b;
    return (size_t)kMaxSortPasses * (256 + 256 * ntiles) + 64;
}

template <class KeyT, int ITEMS>
inline int radix_sort_pairs_tiled(Stream st, KeyT* k[2], uint32_t* v[2], uint64_t n, int bit_lo, int passes, uint32_t* temp)
{
    constexpr int kSortTile = 256 * ITEMS;
    const int result = passes & 1;
    if (n <= (uint64_t)kSortTile) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sort_one_tile_kernel<KeyT, ITEMS>), dim3(1), dim3(256), 0, st, k[0],
                           v ? v[0] : nullptr, k[result], v ? v[result] : nullptr, (uint32_t)n, bit_lo, passes);
        PSIM_HIP(hipGetLastError());
        return result;
    }
    const uint32_t ntiles = (uint32_t)((n + kSortTile - 1) / kSortTile);
    // temp layout: digit counts [passes][256] | tile counters [64] | status [passes][ntiles][256]
    uint32_t* digit_counts = temp;
    uint32_t* next_tile = temp + (size_t)kMaxSortPasses * 256;
    uint32_t* status = next_tile + 64;
    dev_memset(st, temp, 0, ((size_t)kMaxSortPasses * 256 + 64 + (size_t)passes * ntiles * 256) * sizeof(uint32_t));
    uint32_t count_blocks = ntiles < 2048u ? ntiles : 2048u;
    uint32_t per_block = (uint32_t)((n + count_blocks - 1) / count_blocks);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(sort_digit_counts_kernel<KeyT>), dim3(count_blocks), dim3(256), 0, st, k[0],
                       (uint32_t)n, bit_lo, passes, per_block, digit_counts);
    uint32_t* fault = lookback_fault_word();
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sort_pass_kernel<KeyT, ITEMS>), dim3(ntiles), dim3(256), 0, st, k[cur],
                           v ? v[cur] : nullptr, k[cur ^ 1], v ? v[cur ^ 1] : nullptr, (uint32_t)n, bit_lo + 8 * p,
                           digit_counts + p * 256, status + (size_t)p * ntiles * 256, next_tile + p, fault);
        cur ^= 1;
    }
    PSIM_HIP(hipGetLastError());
    return cur;
}

// Sorts (keys[, vals]) by key bits [bit_lo, bit_lo + 8 * ceil((bit_hi - bit_lo) / 8)),
// stable.  k[0]/v[0] hold the input; returns the index (0 or 1) of the buffers holding
// the result.  temp: sort_temp_words(n) words.
template <class KeyT>
inline int radix_sort_pairs(Stream st, KeyT* k[2], uint32_t* v[2], uint64_t n, int bit_lo, int bit_hi,
                            uint32_t* temp)
{
    if (n >= (1ull << 30)) throw std::runtime_error("radix_sort: too many items");
    if (n <= 1 || bit_hi <= bit_lo) return 0;
    const int passes = (bit_hi - bit_lo + 7) / 8;
    if (passes > kMaxSortPasses) throw std::runtime_error("radix_sort: too many digit passes");
    if (n >= kSortLongFrom || n <= 256u * kSortItemsLong)
        return radix_sort_pairs_tiled<KeyT, kSortItemsLong>(st, k, v, n, bit_lo, passes, temp);
    return radix_sort_pairs_tiled<KeyT, kSortItemsShort>(st, k, v, n, bit_lo, passes, temp);
}

// ------------------------------------------------------------------ ordered f64 reduction
// Fixed association for a given n (same result every run): 256-lane blocks each
// sum a contiguous chunk, lanes striding inside it, then one block sums the partials.
constexpr int kReduceBlocks = 1024;

__global__ void __launch_bounds__(256) reduce_partial_kernel(const double* in, uint64_t n, uint64_t chunk,
                                                             double* partial)
{
    __shared__ double s[256];
    uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk;
    if (hi > n) hi = n;
    double acc = 0.0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) acc += in[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

// three short arrays summed in one launch (one 256-lane block each, fixed association)
__global__ void __launch_bounds__(256) reduce3_small_kernel(const double* a, const double* b, const double* c,
                                                            uint64_t n, double* out3)
{
    __shared__ double s[256];
    const double* in = blockIdx.x == 0 ? a : blockIdx.x == 1 ? b : c;
    double acc = 0.0;
    for (uint64_t i = threadIdx.x; i < n; i += 256) acc += in[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out3[blockIdx.x] = s[0];
}
// the same for longer arrays: 64 blocks per array write partial sums, then
// reduce3_small_kernel folds the three lists of 64
__global__ void __launch_bounds__(256) reduce3_partial_kernel(const double* a, const double* b, const double* c,
                                                              uint64_t n, uint64_t chunk, double* partial)
{
    __shared__ double s[256];
    const double* in = blockIdx.y == 0 ? a : blockIdx.y == 1 ? b : c;
    uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk;
    if (hi > n) hi = n;
    double acc = 0.0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) acc += in[i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.y * 64 + blockIdx.x] = s[0];
}
inline void reduce3_sum_f64(Stream st, const double* a, const double* b, const double* c, uint64_t n, double* temp,
                            double* out3);

// temp: kReduceBlocks doubles.  out: device double.
inline void reduce_sum_f64(Stream st, const double* in, uint64_t n, double* temp, double* out)
{
    if (n == 0) { dev_memset(st, out, 0, 8); return; }
    uint64_t chunk = (n + kReduceBlocks - 1) / kReduceBlocks;
    chunk = (chunk + 255) / 256 * 256;
    uint32_t nb = (uint32_t)((n + chunk - 1) / chunk);
    hipLaunchKernelGGL(reduce_partial_kernel, dim3(nb), dim3(256), 0, st, in, n, chunk, temp);
    hipLaunchKernelGGL(reduce_partial_kernel, dim3(1), dim3(256), 0, st, (const double*)temp, (uint64_t)nb,
                       (uint64_t)((nb + 255) / 256 * 256), out);
    PSIM_HIP(hipGetLastError());
}

}  // namespace be
}  // namespace psim

namespace psim {
namespace be {
inline void reduce3_sum_f64(Stream st, const double* a, const double* b, const double* c, uint64_t n, double* temp,
                            double* out3)
{
    if (n == 0) { dev_memset(st, out3, 0, 24); return; }
    if (n <= 4096) {
        hipLaunchKernelGGL(reduce3_small_kernel, dim3(3), dim3(256), 0, st, a, b, c, n, out3);
        PSIM_HIP(hipGetLastError());
        return;
    }
    uint64_t chunk = (n + 63) / 64;
    hipLaunchKernelGGL(reduce3_partial_kernel, dim3(64, 3), dim3(256), 0, st, a, b, c, n, chunk, temp);
    hipLaunchKernelGGL(reduce3_small_kernel, dim3(3), dim3(256), 0, st, (const double*)temp, (const double*)(temp + 64),
                       (const double*)(temp + 128), (uint64_t)64, out3);
    PSIM_HIP(hipGetLastError());
}
}  // namespace be
}  // namespace psim

// ------------------------------------------------------------------ wave-cooperative Barnes-Hut walk
// particle_system.py:69-121 for 64 Morton-adjacent particles at once.  The wave
// walks the pre-order element array with ONE shared cursor, so every element is
// fetched once per wave through the scalar path (one 32-byte load, no per-lane
// gather, no stack).  Each lane keeps the reference's own per-particle
// decision: a lane that accepts an internal node (w^2 < theta^2 d^2, d^2 > 0)
// adds its force and parks until the cursor leaves that node's subtree
// (cursor >= skip); the cursor steps into a node's children while any
// non-parked lane still needs them.  Per particle the set of interactions is
// exactly the reference's; only the order of the additions differs.
namespace psim {
namespace be {

// m (x)^-3/2 from the hardware estimate y = v_rsq_f64(x) (about 2^-29 relative) and one
// Newton correction folded into the cube: with e = 1 - x y^2, x^-1/2 = y (1 + e/2 + ...),
// so x^-3/2 = y^3 (1 + 3e/2) up to terms of order e^2 < 2^-56.
__device__ __forceinline__ double mass_inv_r3(double m, double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double y2 = y * y;
    double e = fma(-x, y2, 1.0);
    double f0 = (m * y) * y2;
    return f0 * fma(1.5, e, 1.0);
}

typedef double double4_t __attribute__((ext_vector_type(4)));

// 32-byte scalar load whose issue point the compiler cannot move (it would otherwise sink
// the second of two independent loads below the work on the first).  The result may only
// be used after scalar_loads_wait() on it.
__device__ __forceinline__ double4_t scalar_load32_issue(const double4_t* base, uint32_t byte_offset)
{
    double4_t r;
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(r) : "s"(base), "s"(byte_offset));
    return r;
}
__device__ __forceinline__ void scalar_loads_wait(double4_t& a, double4_t& b)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b));
}
__device__ __forceinline__ void scalar_loads_wait(double4_t& a, double4_t& b, double4_t& c)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c));
}
__device__ __forceinline__ void scalar_loads_wait(double4_t& a, double4_t& b, double4_t& c, double4_t& d)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
}

// FOLD: theta^2 is a power of two, w^2 < theta^2 d^2 is decided exactly as
// (w^2 / theta^2) < d^2 and the elements' tags were built from W^2 / theta^2.
// WIDE: the element array is 4 GiB or larger, so element offsets need 64 bits.
template <bool COUNT, bool FOLD, bool WIDE>
__global__ void __launch_bounds__(256) gravity_walk_wave_kernel(
    int64_t s_begin, int64_t s_end, int sorted_out, const NodeRec* __restrict__ rec, uint32_t total_elems,
    const vec2* __restrict__ spos, const double* __restrict__ smass, const uint32_t* __restrict__ perm,
    const uint32_t* __restrict__ order, vec2* __restrict__ acc_out, uint32_t w2_lo, double theta2, double g_const,
    double softening, unsigned long long* __restrict__ counters)
{
    // targets are the entries [s_begin, s_end) of `order` (sorted positions in the order the
    // walk groups them); results go to acc_out[particle index] or, for a sharded walk, to
    // acc_out[entry number]
    const int64_t k_target = s_begin + (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = k_target < s_end;
    const uint32_t s = live ? (order ? order[k_target] : (uint32_t)k_target) : 0u;     // (no order: the sorted particles themselves)
    double px = 0.0, py = 0.0;
    if (live) { vec2 p = spos[s]; px = p.x; py = p.y; }
    uint32_t until = live ? 0u : 0xffffffffu;
    double ax = 0.0, ay = 0.0;
    uint32_t visits = 0, inter = 0, trips = 0, force_trips = 0;
    // (counting launches only) how many trips are uniform over the group's active lanes, and how
    // many of those could be told from the bounding box of the group's 64 targets alone
    uint32_t uni_accept = 0, uni_open = 0, box_accept = 0, box_open = 0, full_trips = 0;
    double bx0 = px, bx1 = px, by0 = py, by1 = py;
    if (COUNT) {
        const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
        const int first_live = live_mask ? __ffsll((long long)live_mask) - 1 : 0;
        const double fx = __shfl(px, first_live, 64), fy = __shfl(py, first_live, 64);
        if (!live) { bx0 = bx1 = fx; by0 = by1 = fy; }
        for (int o = 32; o > 0; o >>= 1) {
            bx0 = fmin(bx0, __shfl_xor(bx0, o, 64)); bx1 = fmax(bx1, __shfl_xor(bx1, o, 64));
            by0 = fmin(by0, __shfl_xor(by0, o, 64)); by1 = fmax(by1, __shfl_xor(by1, o, 64));
        }
    }
    const double4_t* __restrict__ rec4 = reinterpret_cast<const double4_t*>(rec);
    uint32_t ju = 0;
    while (ju < total_elems) {
        // one 32-byte scalar fetch per element: {mass, x, y, skip | tag << 32}
        // (byte offset in 32 bits unless the array is 4 GiB or larger)
        const double4_t rv = WIDE ? rec4[ju]
                                  : *reinterpret_cast<const double4_t*>(reinterpret_cast<const char*>(rec4) + (ju << 5));
        const uint32_t skip = (uint32_t)__double2loint(rv.w), tag = (uint32_t)__double2hiint(rv.w);
        const uint32_t ja = ju + 1u;
        const uint32_t jb = skip > ju ? skip : ja;           // strictly monotonic whatever the array holds
        const bool leaf = tag == kLeafTag;
        // internal: accept iff w^2 < theta^2 d^2 (which implies d^2 > 0 since w^2 > 0);
        // leaf: interact iff 0 < d^2.  One compare serves both: lhs < scale * d^2, with lhs
        // assembled from the element's tag (high word) and the low word all depths share.
        const double lhs = __hiloint2double((int)tag, (int)(leaf ? 0u : w2_lo));
        const bool active = until <= ju;
        const double dx = rv.y - px, dy = rv.z - py;
        const double d2 = fma(dx, dx, dy * dy);
        bool accept;
        if (FOLD) {
            accept = lhs < d2;
        } else {
            const double scale = leaf ? 1.0 : theta2;
            accept = lhs < scale * d2;
        }
        const bool take = active && accept;
        if (COUNT) {
            const unsigned long long take_mask = __builtin_amdgcn_ballot_w64(take);
            visits += active ? 1u : 0u; inter += take ? 1u : 0u; trips += 1u; force_trips += take_mask ? 1u : 0u;
            const unsigned long long act_mask = __builtin_amdgcn_ballot_w64(active);
            if (act_mask) {
                uni_accept += take_mask == act_mask ? 1u : 0u;
                uni_open += take_mask == 0ull ? 1u : 0u;
                full_trips += act_mask == ~0ull ? 1u : 0u;
                // nearest and farthest point of the box from the element's centre of mass
                const double nx = fmax(fmax(bx0 - rv.y, rv.y - bx1), 0.0), ny = fmax(fmax(by0 - rv.z, rv.z - by1), 0.0);
                const double wx = fmax(fabs(rv.y - bx0), fabs(rv.y - bx1)), wy = fmax(fabs(rv.z - by0), fabs(rv.z - by1));
                const double sc = FOLD ? 1.0 : (leaf ? 1.0 : theta2);
                box_accept += lhs < sc * fma(nx, nx, ny * ny) ? 1u : 0u;
                box_open += (!leaf && !(lhs < sc * fma(wx, wx, wy * wy))) ? 1u : 0u;
            }
        }
        if (take) {                                          // skipped as a whole when no lane accepts
            double f = mass_inv_r3(rv.x, d2 + softening);
            ax = fma(f, dx, ax);
            ay = fma(f, dy, ay);
            until = jb;
        }
        // a lane still wants this node's children iff it was active and did not accept, i.e.
        // its park index is still <= the cursor (accepting lanes just moved theirs past it)
        const bool descend = !leaf && (__builtin_amdgcn_ballot_w64(until <= ju) != 0ull);
        ju = descend ? ja : jb;
    }
    if (live) {
        uint32_t p = perm[s];
        double m = smass[s];
        double gm = g_const * m;
        double im = 1.0 / m;          // = the stored inverse mass (particle_system.py:374), bit for bit
        vec2 a;
        a.x = (gm * ax) * im;
        a.y = (gm * ay) * im;
        acc_out[sorted_out ? (uint32_t)k_target : p] = a;
    }
    if (COUNT) {
        atomicAdd(&counters[0], (unsigned long long)visits);
        atomicAdd(&counters[1], (unsigned long long)inter);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&counters[2], (unsigned long long)trips);
            atomicAdd(&counters[3], (unsigned long long)force_trips);
            atomicAdd(&counters[4], (unsigned long long)uni_accept);
            atomicAdd(&counters[5], (unsigned long long)uni_open);
            atomicAdd(&counters[6], (unsigned long long)box_accept);
            atomicAdd(&counters[7], (unsigned long long)box_open);
            atomicAdd(&counters[8], (unsigned long long)full_trips);
        }
    }
}

// The two-group loop of gravity_walk_multi_kernel, scheduled by hand.  Same arithmetic in the
// same order as the compiler's version of `trip` (results are bit-identical); what differs is
// the scalar side and the fetches.
// - Per group and trip the compiler's loop spends a third vector compare and some eight scalar
//   instructions on turning wave masks into booleans and back (is the element a leaf, does any
//   lane still want its children, has either cursor run out).  Here the lanes that still want
//   an element's children are the active ones that did not accept, `active & ~take`: one
//   s_andn2 whose SCC selects the next cursor directly; a leaf needs no case of its own there
//   (its skip is the next element, so both candidates are equal); the loop ends on an SCC branch.
// - Each fetch brings the element at the cursor and the one after it (64 bytes; group A in
//   s[40:55], group B in s[56:71]; an element is {mass, x, y, skip, tag}).  The next cursor is
//   the following element on most turns (always after a leaf, and whenever some lane opens a
//   node), and that element is then worked on straight away: no fetch, no wait.  The element
//   array ends with one element of slack, so the fetch at the last element stays inside the
//   allocation; what it brings is never used.
// - Softening and the constant 1.5 sit in vector registers, EXEC (all lanes on entry: the
//   kernel runs whole waves in uniform control flow) is put back as -1, and one scalar serves
//   as fetch offset and as skip target: the kernel has to stay within 80 scalar registers,
//   or the side stream's kernels lose the wave slots beside it.
// v_rsq_f64 is followed by an independent instruction before its result is read, as the
// compiler's schedule has it.  (Labels end in "_%=": the assembler numbers the copies of a
// block, and without the separator label `a` of copy 20 is label `a2` of copy 0.)
#define PSIM_WALK_GROUP(ACCEPT, M, X, Y, SKIP, TAG, LHS, PX, PY, AX, AY, UNTIL, CUR, L)                 \
    "v_add_f64 %[dy], " Y ", -%[" PY "]\n"                                                              \
    "v_add_f64 %[dx], " X ", -%[" PX "]\n"                                                              \
    "s_max_u32 %[jb], " SKIP ", %[ja]\n"                                                                \
    "v_mul_f64 %[d2], %[dy], %[dy]\n"                                                                   \
    "s_cmp_lg_u32 " TAG ", 0\n"                                                                         \
    "s_cselect_b32 " SKIP ", %[w2lo], 0\n"                                                              \
    "v_fmac_f64 %[d2], %[dx], %[dx]\n"                                                                  \
    "v_cmp_ge_u32 vcc, %[" CUR "], %[" UNTIL "]\n"                                                      \
    ACCEPT(LHS)                                                                                         \
    "s_and_b64 %[take], vcc, %[take]\n"                                                                 \
    "s_cbranch_scc0 .Lpsim_walk_" L "_%=\n"                                                              \
    "s_mov_b64 exec, %[take]\n"                                                                         \
    "v_add_f64 %[d2], %[soft], %[d2]\n"                                                                 \
    "v_rsq_f64 %[y], %[d2]\n"                                                                           \
    "v_mov_b32 %[" UNTIL "], %[jb]\n"                                                                   \
    "v_mul_f64 %[y2], %[y], %[y]\n"                                                                     \
    "v_mul_f64 %[y], " M ", %[y]\n"                                                                     \
    "v_fma_f64 %[d2], -%[d2], %[y2], 1.0\n"                                                             \
    "v_mul_f64 %[y], %[y], %[y2]\n"                                                                     \
    "v_fma_f64 %[d2], %[d2], %[c15], 1.0\n"                                                             \
    "v_mul_f64 %[d2], %[y], %[d2]\n"                                                                    \
    "v_fmac_f64 %[" AX "], %[d2], %[dx]\n"                                                              \
    "v_fmac_f64 %[" AY "], %[d2], %[dy]\n"                                                              \
    "s_mov_b64 exec, -1\n"                                                                              \
    ".Lpsim_walk_" L "_%=:\n"                                                                            \
    "s_andn2_b64 vcc, vcc, %[take]\n"                                                                   \
    "s_cselect_b32 %[" CUR "], %[ja], %[jb]\n"
// The acceptance test.  Folded (theta^2 a power of two, divided into the elements' tags): lhs < d^2.
// Scaled (any theta): lhs < theta^2 d^2 for a node, lhs = 0 < d^2 for a leaf; the factor is
// picked by the SCC the leaf test just left (s_cselect reads it, nothing in between writes it).
#define PSIM_WALK_ACCEPT_FOLDED(LHS) "v_cmp_lt_f64 %[take], " LHS ", %[d2]\n"
#define PSIM_WALK_ACCEPT_SCALED(LHS)                                                                    \
    "s_cselect_b64 %[sc], %[th2], %[one]\n"                                                             \
    "v_mul_f64 %[y2], %[sc], %[d2]\n"                                                                   \
    "v_cmp_lt_f64 %[take], " LHS ", %[y2]\n"
// the element fetched along, when the cursor has moved onto it and it exists
#define PSIM_WALK_SECOND(ACCEPT, M, X, Y, SKIP, TAG, LHS, PX, PY, AX, AY, UNTIL, CUR, L)                \
    "s_cmp_lg_u32 %[" CUR "], %[ja]\n"                                                                  \
    "s_cbranch_scc1 .Lpsim_walk_done_" L "_%=\n"                                                         \
    "s_cmp_ge_u32 %[" CUR "], %[tot]\n"                                                                 \
    "s_cbranch_scc1 .Lpsim_walk_done_" L "_%=\n"                                                         \
    "s_add_i32 %[ja], %[" CUR "], 1\n"                                                                  \
    PSIM_WALK_GROUP(ACCEPT, M, X, Y, SKIP, TAG, LHS, PX, PY, AX, AY, UNTIL, CUR, L)                     \
    ".Lpsim_walk_done_" L "_%=:\n"
#define PSIM_WALK_A1(ACC) PSIM_WALK_GROUP(ACC, "s[40:41]", "s[42:43]", "s[44:45]", "s46", "s47", "s[46:47]", "pxA", "pyA", "axA", "ayA", "uA", "jA", "a")
#define PSIM_WALK_A2(ACC) PSIM_WALK_SECOND(ACC, "s[48:49]", "s[50:51]", "s[52:53]", "s54", "s55", "s[54:55]", "pxA", "pyA", "axA", "ayA", "uA", "jA", "a2")
#define PSIM_WALK_B1(ACC) PSIM_WALK_GROUP(ACC, "s[56:57]", "s[58:59]", "s[60:61]", "s62", "s63", "s[62:63]", "pxB", "pyB", "axB", "ayB", "uB", "jB", "b")
#define PSIM_WALK_B2(ACC) PSIM_WALK_SECOND(ACC, "s[64:65]", "s[66:67]", "s[68:69]", "s70", "s71", "s[70:71]", "pxB", "pyB", "axB", "ayB", "uB", "jB", "b2")
// The fetch of two elements at a cursor: a 32-bit byte offset from the array's base, or, for
// arrays of 4 GiB and more, the 64-bit address formed in s[38:39].
#define PSIM_WALK_FETCH_NARROW(DST, CUR)                                                                \
        "s_lshl_b32 %[jb], %[" CUR "], 5\n"                                                             \
        "s_load_dwordx16 " DST ", %[rec], %[jb]\n"
#define PSIM_WALK_FETCH_WIDE(DST, CUR)                                                                  \
        "s_lshl_b32 s38, %[" CUR "], 5\n"                                                               \
        "s_lshr_b32 s39, %[" CUR "], 27\n"                                                              \
        "s_add_u32 s38, s38, %[reclo]\n"                                                                \
        "s_addc_u32 s39, s39, %[rechi]\n"                                                               \
        "s_load_dwordx16 " DST ", s[38:39], 0x0\n"
#define PSIM_WALK_TWO(ACC, FETCH)                                                                        \
        "s_max_u32 %[jb], %[jA], %[jB]\n"                                                               \
        "s_cmp_lt_u32 %[jb], %[tot]\n"                                                                  \
        "s_cbranch_scc0 .Lpsim_walk_end_%=\n"                                                            \
        ".Lpsim_walk_loop_%=:\n"                                                                         \
        FETCH("s[40:55]", "jA")                                                                         \
        FETCH("s[56:71]", "jB")                                                                         \
        "s_add_i32 %[ja], %[jA], 1\n"                                                                   \
        "s_waitcnt lgkmcnt(0)\n"                                                                        \
        PSIM_WALK_A1(ACC)                                                                               \
        PSIM_WALK_A2(ACC)                                                                               \
        "s_add_i32 %[ja], %[jB], 1\n"                                                                   \
        PSIM_WALK_B1(ACC)                                                                               \
        PSIM_WALK_B2(ACC)                                                                               \
        "s_max_u32 %[jb], %[jA], %[jB]\n"                                                               \
        "s_cmp_lt_u32 %[jb], %[tot]\n"                                                                  \
        "s_cbranch_scc1 .Lpsim_walk_loop_%=\n"                                                           \
        ".Lpsim_walk_end_%=:\n"
#define PSIM_WALK_ONE(ACC, FETCH)                                                                        \
        "s_cmp_lt_u32 %[jA], %[tot]\n"                                                                  \
        "s_cbranch_scc0 .Lpsim_walk_end_%=\n"                                                            \
        ".Lpsim_walk_loop_%=:\n"                                                                         \
        FETCH("s[40:55]", "jA")                                                                         \
        "s_add_i32 %[ja], %[jA], 1\n"                                                                   \
        "s_waitcnt lgkmcnt(0)\n"                                                                        \
        PSIM_WALK_A1(ACC)                                                                               \
        PSIM_WALK_A2(ACC)                                                                               \
        "s_cmp_lt_u32 %[jA], %[tot]\n"                                                                  \
        "s_cbranch_scc1 .Lpsim_walk_loop_%=\n"                                                           \
        ".Lpsim_walk_end_%=:\n"
#define PSIM_WALK_CLOBBERS_A "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"
#define PSIM_WALK_CLOBBERS_B "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71"

#define PSIM_WALK_IN_NARROW , [rec] "s"(rec4)
#define PSIM_WALK_IN_WIDE , [reclo] "s"(rec_lo), [rechi] "s"(rec_hi)
#define PSIM_WALK_CLOB_NARROW
#define PSIM_WALK_CLOB_WIDE , "s38", "s39"
#define PSIM_WALK_IN_FOLDED
#define PSIM_WALK_IN_SCALED , [th2] "s"(theta2), [one] "s"(one)
#define PSIM_WALK_OUT_FOLDED
#define PSIM_WALK_OUT_SCALED , [sc] "=&s"(scale)
#define PSIM_WALK_ASM_TWO(A, F)                                                                                          \
    asm volatile(PSIM_WALK_TWO(PSIM_WALK_ACCEPT_##A, PSIM_WALK_FETCH_##F)                                                \
        : [jA] "+s"(ja_cur), [jB] "+s"(jb_cur), [axA] "+v"(axa), [ayA] "+v"(aya), [axB] "+v"(axb), [ayB] "+v"(ayb),      \
          [uA] "+v"(untila), [uB] "+v"(untilb), [dx] "=&v"(dx), [dy] "=&v"(dy), [d2] "=&v"(d2), [y] "=&v"(y),            \
          [y2] "=&v"(y2), [take] "=&s"(take), [ja] "=&s"(ja), [jb] "=&s"(jb) PSIM_WALK_OUT_##A                           \
        : [tot] "s"(total_elems), [w2lo] "s"(w2_lo), [soft] "v"(softening), [c15] "v"(c15), [pxA] "v"(pxa),              \
          [pyA] "v"(pya), [pxB] "v"(pxb), [pyB] "v"(pyb) PSIM_WALK_IN_##A PSIM_WALK_IN_##F                               \
        : PSIM_WALK_CLOBBERS_A, PSIM_WALK_CLOBBERS_B PSIM_WALK_CLOB_##F)
#define PSIM_WALK_ASM_ONE(A, F)                                                                                          \
    asm volatile(PSIM_WALK_ONE(PSIM_WALK_ACCEPT_##A, PSIM_WALK_FETCH_##F)                                                \
        : [jA] "+s"(ja_cur), [axA] "+v"(axa), [ayA] "+v"(aya), [uA] "+v"(untila), [dx] "=&v"(dx), [dy] "=&v"(dy),        \
          [d2] "=&v"(d2), [y] "=&v"(y), [y2] "=&v"(y2), [take] "=&s"(take), [ja] "=&s"(ja),                              \
          [jb] "=&s"(jb) PSIM_WALK_OUT_##A                                                                               \
        : [tot] "s"(total_elems), [w2lo] "s"(w2_lo), [soft] "v"(softening), [c15] "v"(c15), [pxA] "v"(pxa),              \
          [pyA] "v"(pya) PSIM_WALK_IN_##A PSIM_WALK_IN_##F                                                               \
        : PSIM_WALK_CLOBBERS_A PSIM_WALK_CLOB_##F)

// FOLD, WIDE: see gravity_walk_multi_kernel; theta2 is only read when FOLD is false.
template <bool FOLD, bool WIDE>
__device__ __forceinline__ void walk_two_groups_loop(const double4_t* rec4, uint32_t& ja_cur, uint32_t& jb_cur,
                                                     uint32_t total_elems, uint32_t w2_lo, double theta2, double softening,
                                                     double pxa, double pya, double pxb, double pyb, double& axa,
                                                     double& aya, double& axb, double& ayb, uint32_t& untila,
                                                     uint32_t& untilb)
{
    double dx, dy, d2, y, y2;
    unsigned long long take;
    uint32_t ja, jb;
    const double c15 = 1.5;
    [[maybe_unused]] double scale;
    [[maybe_unused]] const double one = 1.0;
    [[maybe_unused]] const uint32_t rec_lo = (uint32_t)reinterpret_cast<uintptr_t>(rec4);
    [[maybe_unused]] const uint32_t rec_hi = (uint32_t)(reinterpret_cast<uintptr_t>(rec4) >> 32);
    if constexpr (FOLD && !WIDE) PSIM_WALK_ASM_TWO(FOLDED, NARROW);
    else if constexpr (FOLD) PSIM_WALK_ASM_TWO(FOLDED, WIDE);
    else if constexpr (!WIDE) PSIM_WALK_ASM_TWO(SCALED, NARROW);
    else PSIM_WALK_ASM_TWO(SCALED, WIDE);
}

// One group alone (what is left of a wave's pair once the other group has finished): the same
// trip with one fetch per turn.
template <bool FOLD, bool WIDE>
__device__ __forceinline__ void walk_one_group_loop(const double4_t* rec4, uint32_t& ja_cur, uint32_t total_elems,
                                                    uint32_t w2_lo, double theta2, double softening, double pxa, double pya,
                                                    double& axa, double& aya, uint32_t& untila)
{
    double dx, dy, d2, y, y2;
    unsigned long long take;
    uint32_t ja, jb;
    const double c15 = 1.5;
    [[maybe_unused]] double scale;
    [[maybe_unused]] const double one = 1.0;
    [[maybe_unused]] const uint32_t rec_lo = (uint32_t)reinterpret_cast<uintptr_t>(rec4);
    [[maybe_unused]] const uint32_t rec_hi = (uint32_t)(reinterpret_cast<uintptr_t>(rec4) >> 32);
    if constexpr (FOLD && !WIDE) PSIM_WALK_ASM_ONE(FOLDED, NARROW);
    else if constexpr (FOLD) PSIM_WALK_ASM_ONE(FOLDED, WIDE);
    else if constexpr (!WIDE) PSIM_WALK_ASM_ONE(SCALED, NARROW);
    else PSIM_WALK_ASM_ONE(SCALED, WIDE);
}
#undef PSIM_WALK_ASM_ONE
#undef PSIM_WALK_ASM_TWO
#undef PSIM_WALK_IN_NARROW
#undef PSIM_WALK_IN_WIDE
#undef PSIM_WALK_CLOB_NARROW
#undef PSIM_WALK_CLOB_WIDE
#undef PSIM_WALK_IN_FOLDED
#undef PSIM_WALK_IN_SCALED
#undef PSIM_WALK_OUT_FOLDED
#undef PSIM_WALK_OUT_SCALED
#undef PSIM_WALK_FETCH_NARROW
#undef PSIM_WALK_FETCH_WIDE
#undef PSIM_WALK_CLOBBERS_A
#undef PSIM_WALK_CLOBBERS_B
#undef PSIM_WALK_ONE
#undef PSIM_WALK_TWO
#undef PSIM_WALK_A1
#undef PSIM_WALK_A2
#undef PSIM_WALK_B1
#undef PSIM_WALK_B2
#undef PSIM_WALK_SECOND
#undef PSIM_WALK_ACCEPT_FOLDED
#undef PSIM_WALK_ACCEPT_SCALED
#undef PSIM_WALK_GROUP

// K walks per wave: each lane carries one particle of each of K groups of 64
// Morton-adjacent particles (the wave covers 64 K consecutive sorted particles), each
// group with its own cursor.  The K elements are requested together, so one wait covers
// K fetches and one group's trip hides the others' latency; when the first group
// finishes, the rest complete one after the other.
// The kernel's scalar register count matters beyond its own occupancy: it shares every SIMD
// with the side stream's kernels, and a build that took 83 instead of 77 SGPRs (two more
// pointer arguments, for a velocity kick fused into the epilogue) left those kernels 1.5x
// slower under the walk (tools/kernel_resources.py lists what every kernel allocates).
// amdgpu_num_sgpr(80) holds every variant at the 80-register allocation step.
// HAND (two groups): the two-group loop and the one-group
// tails are the hand-scheduled ones above (walk_two_groups_loop, walk_one_group_loop) instead
// of the compiler's.
template <int K, bool FOLD, bool WIDE, bool HAND = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) gravity_walk_multi_kernel(
    int64_t s_begin, int64_t s_end, int sorted_out, const NodeRec* __restrict__ rec, uint32_t total_elems,
    const vec2* __restrict__ spos, const double* __restrict__ smass, const uint32_t* __restrict__ perm,
    const uint32_t* __restrict__ order, vec2* __restrict__ acc_out, uint32_t w2_lo, double theta2, double g_const,
    double softening, uint32_t nchunks, uint32_t* __restrict__ chunk_counter)
{
    static_assert(K >= 2 && K <= 4, "two to four walks per wave");
    // One wave per workgroup (no LDS is used).  With chunk_counter the grid is a fixed number of
    // waves that take chunks of 64 K targets from a counter until none are left: the walk then
    // holds a set number of wave slots per SIMD and leaves the rest to the kernels of the other
    // stream for its whole duration.  The last wave to run dry clears the counters again.
    const int lane = threadIdx.x;
    const uint32_t total_elems_arg = total_elems;
    uint32_t chunk = blockIdx.x;
    bool first = true;
  for (;;) {
    if (chunk_counter) {
        // a wave's first chunk is its own number (no burst of same-address atomics when the
        // kernel starts); the counter hands out the chunks after those
        if (!first) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(&chunk_counter[0], 1u);
            chunk = gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        }
        first = false;
        if (chunk >= nchunks) {
            if (lane == 0 && atomicAdd(&chunk_counter[1], 1u) == gridDim.x - 1) {
                __hip_atomic_store(&chunk_counter[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&chunk_counter[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    } else if (chunk >= nchunks) {
        return;
    }
    total_elems = total_elems_arg;
    const int64_t s0 = s_begin + (int64_t)chunk * (64 * K) + lane;
    double px[K], py[K], ax[K], ay[K];
    uint32_t until[K], j[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t kt = s0 + 64 * k;          // entry of `order`: the k-th group's target of this lane
        const bool live = kt < s_end;
        px[k] = 0.0; py[k] = 0.0; ax[k] = 0.0; ay[k] = 0.0;
        if (live) { vec2 p = spos[order[kt]]; px[k] = p.x; py[k] = p.y; }
        until[k] = live ? 0u : 0xffffffffu;
        // a group that lies beyond the range altogether is not walked
        j[k] = __builtin_amdgcn_ballot_w64(live) ? 0u : total_elems;
    }
    const double4_t* __restrict__ rec4 = reinterpret_cast<const double4_t*>(rec);
    auto fetch = [&](uint32_t ju) -> double4_t {
        return WIDE ? rec4[ju] : *reinterpret_cast<const double4_t*>(reinterpret_cast<const char*>(rec4) + (ju << 5));
    };
    // one trip of one group: the body of gravity_walk_wave_kernel's loop
    auto trip = [&](const double4_t rv, uint32_t ju, double pxk, double pyk, double& axk, double& ayk,
                    uint32_t& untilk) -> uint32_t {
        const uint32_t skip = (uint32_t)__double2loint(rv.w), tag = (uint32_t)__double2hiint(rv.w);
        const uint32_t ja = ju + 1u;
        const uint32_t jb = skip > ju ? skip : ja;
        const bool leaf = tag == kLeafTag;
        const double lhs = __hiloint2double((int)tag, (int)(leaf ? 0u : w2_lo));
        const bool active = untilk <= ju;
        const double dx = rv.y - pxk, dy = rv.z - pyk;
        const double d2 = fma(dx, dx, dy * dy);
        bool accept;
        if (FOLD) {
            accept = lhs < d2;
        } else {
            const double scale = leaf ? 1.0 : theta2;
            accept = lhs < scale * d2;
        }
        if (active && accept) {
            double f = mass_inv_r3(rv.x, d2 + softening);
            axk = fma(f, dx, axk);
            ayk = fma(f, dy, ayk);
            untilk = jb;
        }
        const bool descend = !leaf && (__builtin_amdgcn_ballot_w64(untilk <= ju) != 0ull);
        return descend ? ja : jb;
    };
    if (!WIDE || HAND) {
        // the kernel arguments the loop reads are pinned into registers here, so that the
        // compiler places no scalar-load wait of its own between the requests and their use
        asm volatile("" : "+s"(w2_lo), "+s"(theta2), "+s"(softening), "+s"(total_elems));
        if constexpr (HAND) {
            static_assert(K == 2, "the hand-scheduled loop is the two-group one");
            // (three groups per wave with the same loop were measured: no faster alone, and the
            // 90 scalar registers they need leave the side stream's kernels fewer waves)
            walk_two_groups_loop<FOLD, WIDE>(rec4, j[0], j[1], total_elems, w2_lo, theta2, softening, px[0], py[0], px[1],
                                             py[1], ax[0], ay[0], ax[1], ay[1], until[0], until[1]);
            walk_one_group_loop<FOLD, WIDE>(rec4, j[0], total_elems, w2_lo, theta2, softening, px[0], py[0], ax[0], ay[0],
                                            until[0]);
            walk_one_group_loop<FOLD, WIDE>(rec4, j[1], total_elems, w2_lo, theta2, softening, px[1], py[1], ax[1], ay[1],
                                            until[1]);
        } else
        for (;;) {
            bool all = true;
#pragma unroll
            for (int k = 0; k < K; ++k) all = all && (j[k] < total_elems);
            if (!all) break;
            double4_t rv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) rv[k] = scalar_load32_issue(rec4, j[k] << 5);
            if constexpr (K == 2) scalar_loads_wait(rv[0], rv[1]);
            if constexpr (K == 3) scalar_loads_wait(rv[0], rv[1], rv[2]);
            if constexpr (K == 4) scalar_loads_wait(rv[0], rv[1], rv[2], rv[3]);
#pragma unroll
            for (int k = 0; k < K; ++k) j[k] = trip(rv[k], j[k], px[k], py[k], ax[k], ay[k], until[k]);
        }
    }
    if constexpr (!HAND) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            while (j[k] < total_elems) j[k] = trip(fetch(j[k]), j[k], px[k], py[k], ax[k], ay[k], until[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t kt = s0 + 64 * k;
        if (kt < s_end) {
            const uint32_t s = order[kt];
            uint32_t p = perm[s];
            double m = smass[s];
            double gm = g_const * m;
            double im = 1.0 / m;      // = the stored inverse mass (particle_system.py:374), bit for bit
            vec2 a;
            a.x = (gm * ax[k]) * im;
            a.y = (gm * ay[k]) * im;
            acc_out[sorted_out ? (uint32_t)kt : p] = a;
        }
    }
    if (!chunk_counter) return;
  }
}

template <int K>
inline void launch_walk_multi(Stream st, int64_t lo, int64_t hi, int sorted_out, const NodeRec* rec, uint32_t total,
                              const vec2* spos, const double* smass, const uint32_t* perm, const uint32_t* order,
                              vec2* out, bool fold, uint32_t w2_lo, double theta2, double g_const, double softening,
                              uint32_t* chunk_counter, int64_t resident_waves, bool force_wide = false,
                              bool compiler_loop = false)
{
    if (hi <= lo) return;
    // (the kernel reads `order` unconditionally: a null check in it costs two scalar registers,
    // which takes the allocation from 80 to 96)
    if (!order) throw std::runtime_error("launch_walk_multi: no target order");
    int64_t chunks = (hi - lo + 64 * K - 1) / (64 * K);
    if (chunks > 0x7fffffffLL) throw std::runtime_error("launch_walk_multi: grid too large");
    // chunk_counter (two zero words) + resident_waves > 0: a fixed number of waves share the chunks
    int64_t blocks = chunks;
    if (chunk_counter && resident_waves > 0 && resident_waves < chunks) blocks = resident_waves;
    else chunk_counter = nullptr;
    // (force_wide: the 64-bit-offset variant on a small array, for tests)
    const bool wide = force_wide || (uint64_t)total * sizeof(NodeRec) >= (1ull << 32);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64), 0, st, lo, hi, sorted_out, rec, total, spos, smass,
                           perm, order, out, w2_lo, theta2, g_const, softening, (uint32_t)chunks, chunk_counter);
    };
    // (K == 2: the hand-scheduled loops unless the compiler's are asked for)
    auto pick = [&](auto fold_c, auto wide_c) {
        constexpr bool F = decltype(fold_c)::value, W = decltype(wide_c)::value;
        if constexpr (K == 2) {
            if (!compiler_loop) go(gravity_walk_multi_kernel<2, F, W, true>);
            else go(gravity_walk_multi_kernel<2, F, W>);
        } else {
            go(gravity_walk_multi_kernel<K, F, W>);
        }
    };
    if (fold && !wide) pick(std::true_type(), std::false_type());
    else if (fold) pick(std::true_type(), std::true_type());
    else if (!wide) pick(std::false_type(), std::false_type());
    else pick(std::false_type(), std::true_type());
    PSIM_HIP(hipGetLastError());
}

template <bool COUNT>
inline void launch_walk(Stream st, int64_t lo, int64_t hi, int sorted_out, const NodeRec* rec, uint32_t total,
                        const vec2* spos, const double* smass, const uint32_t* perm, const uint32_t* order, vec2* out,
                        bool fold, uint32_t w2_lo, double theta2, double g_const, double softening,
                        unsigned long long* counters, bool force_wide = false)
{
    if (hi <= lo) return;
    int64_t blocks = (hi - lo + 255) / 256;
    if (blocks > 0x7fffffffLL) throw std::runtime_error("launch_walk: grid too large");
    const bool wide = force_wide || (uint64_t)total * sizeof(NodeRec) >= (1ull << 32);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), 0, st, lo, hi, sorted_out, rec, total, spos, smass,
                           perm, order, out, w2_lo, theta2, g_const, softening, counters);
    };
    if (fold && !wide) go(gravity_walk_wave_kernel<COUNT, true, false>);
    else if (fold) go(gravity_walk_wave_kernel<COUNT, true, true>);
    else if (!wide) go(gravity_walk_wave_kernel<COUNT, false, false>);
    else go(gravity_walk_wave_kernel<COUNT, false, true>);
    PSIM_HIP(hipGetLastError());
}

}  // namespace be
}  // namespace psim

// ------------------------------------------------------------------ stage timers (HIP events)
namespace psim {
namespace be {
struct StageTimer {
    static constexpr int kMarks = 9;
    static constexpr int kMaxSets = 64;
    // One set of events per step of the current psim_step call (the first kMaxSets steps of
    // it; later steps reuse the last set), so that a call of many steps is timed without a
    // host synchronisation per step: read() averages over the sets after the call's end.
    struct Set {
        hipEvent_t ev[kMarks];
        int src[kMarks];           // mark i was taken at event src[i]
        bool complete;
    };
    std::vector<Set> sets;
    int cur = -1, used = 0;
    void init() { if (cur < 0) next_step(); }
    void destroy()
    {
        for (Set& s : sets)
            for (int i = 0; i < kMarks; ++i) (void)hipEventDestroy(s.ev[i]);
        sets.clear();
        cur = -1; used = 0;
    }
    void begin_call() { used = 0; cur = -1; }
    void next_step()
    {
        if (used < kMaxSets) {
            if (used == (int)sets.size()) {
                Set s;
                for (int i = 0; i < kMarks; ++i) PSIM_HIP(hipEventCreate(&s.ev[i]));
                sets.push_back(s);
            }
            cur = used++;
        }
        Set& s = sets[cur];
        for (int i = 0; i < kMarks; ++i) s.src[i] = i;
        s.complete = false;
    }
    void mark(Stream st, int i)
    {
        init();
        Set& s = sets[cur];
        PSIM_HIP(hipEventRecord(s.ev[i], st));
        s.src[i] = i;
        if (i == kMarks - 1) s.complete = true;
    }
    // mark j at the place of mark i (nothing was submitted in between): no second record
    void same_as(int j, int i) { init(); sets[cur].src[j] = sets[cur].src[i]; }
    // a mark doubles as the event other streams or the host wait on (every record costs the
    // stream some 6 us, so the step does not record a second event where a mark already is)
    Event event(int i) { init(); return sets[cur].ev[i]; }
    void read(double* ms)
    {
        for (int i = 0; i < kMarks - 1; ++i) ms[i] = 0.0;
        int counted = 0;
        for (int k = 0; k < used; ++k) {
            Set& s = sets[k];
            if (!s.complete) continue;
            PSIM_HIP(hipEventSynchronize(s.ev[kMarks - 1]));
            for (int i = 0; i < kMarks - 1; ++i) {
                float t = 0.f;
                if (s.src[i] != s.src[i + 1]) PSIM_HIP(hipEventElapsedTime(&t, s.ev[s.src[i]], s.ev[s.src[i + 1]]));
                ms[i] += (double)t;
            }
            ++counted;
        }
        if (counted > 1) for (int i = 0; i < kMarks - 1; ++i) ms[i] /= counted;
    }
};
}  // namespace be
}  // namespace psim

// ------------------------------------------------------------------ brute-force cross-check
// All-pairs softened gravity, a_i = G sum_j m_j (x_j - x_i) (d^2 + s)^-1.5 over d^2 > 0
// (the force law of particle_system.py:86-93 applied to every particle instead of
// tree nodes).  256 targets per workgroup, one per lane; sources staged through LDS
// in tiles of 256 and read back as wave-wide broadcasts.  FP64 VALU-bound: used to
// validate the tree walk (theta = 0 must reproduce it) and as a roofline yardstick.
namespace psim {
namespace be {
__global__ void __launch_bounds__(256) brute_force_kernel(int64_t n, const vec2* __restrict__ pos,
                                                          const double* __restrict__ mass,
                                                          vec2* __restrict__ acc_out, double g_const, double softening)
{
    __shared__ double sx[256], sy[256], sm[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double px = 0.0, py = 0.0;
    if (i < n) { vec2 p = pos[i]; px = p.x; py = p.y; }
    double ax = 0.0, ay = 0.0;
    for (int64_t base = 0; base < n; base += 256) {
        int64_t j = base + threadIdx.x;
        if (j < n) { vec2 q = pos[j]; sx[threadIdx.x] = q.x; sy[threadIdx.x] = q.y; sm[threadIdx.x] = mass[j]; }
        else { sx[threadIdx.x] = px; sy[threadIdx.x] = py; sm[threadIdx.x] = 0.0; }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 256; ++k) {
            double dx = sx[k] - px, dy = sy[k] - py;
            double d2 = fma(dx, dx, dy * dy);
            double f = mass_inv_r3(sm[k], d2 + softening);
            f = d2 > 0.0 ? f : 0.0;
            ax = fma(f, dx, ax);
            ay = fma(f, dy, ay);
        }
        __syncthreads();
    }
    if (i < n) { vec2 a; a.x = g_const * ax; a.y = g_const * ay; acc_out[i] = a; }
}
}  // namespace be
}  // namespace psim

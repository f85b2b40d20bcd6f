// What a kernel body asks of the lanes around it (atomics, wave-wide folds), for gfx950: 64-wide
// waves, one lane per element.  Included by psim_types.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psim {

__device__ __forceinline__ uint32_t atomic_inc_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
__device__ __forceinline__ int clz64(uint64_t x) { return x ? __clzll((long long)x) : 64; }

// add 1 to *p for every calling lane; one atomic per wave (the lanes that reach
// this call together are counted by ballot and the first of them adds the sum)
__device__ __forceinline__ void aggregated_inc(uint32_t* p)
{
    unsigned long long m = __ballot(1);
    int lane = (int)(threadIdx.x & 63u);
    if (lane == __ffsll((long long)m) - 1) atomicAdd(p, (uint32_t)__popcll(m));
}

// *p = max(*p, v).  The current value is looked at first and the atomic only issued by lanes
// that would raise it: a maximum over many lanes then costs a handful of atomics, not one
// same-address atomic per wave (a stale look only costs a needless atomic).
__device__ __forceinline__ void raise_to_u32(uint32_t* p, uint32_t v)
{
    // a plain (cacheable, possibly stale) look first: most lanes stop there
    if (v > *reinterpret_cast<volatile uint32_t*>(p) &&
        v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(p, v);
}

// a[lo .. hi] = v (both ends included; nothing when hi < lo).  Every lane of the wave that runs
// the kernel body calls this, each with a run of its own.  A long run (whole empty grid rows
// around a disk of particles: tens of thousands of cells) is filled by the lanes of the wave
// together; one lane alone would keep its wave alive for as many stores.
__device__ __forceinline__ void fill_run_i32(int32_t* a, int64_t lo, int64_t hi, int32_t v)
{
    constexpr int64_t kOwn = 16;
    const bool far = hi - lo >= kOwn;
    if (!far) for (int64_t c = lo; c <= hi; ++c) a[c] = v;
    // (the lanes that run the body may be fewer than 64 in the last wave of the launch)
    const unsigned long long here = __ballot(1);
    unsigned long long todo = __ballot(far);
    const int lane = (int)(threadIdx.x & 63u);
    const int rank = __popcll(here & ((1ull << lane) - 1ull)), width = __popcll(here);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t l2 = (int64_t)__shfl((long long)lo, src, 64), h2 = (int64_t)__shfl((long long)hi, src, 64);
        const int32_t w = (int32_t)__shfl((long long)v, src, 64);
        for (int64_t c = l2 + rank; c <= h2; c += width) a[c] = w;
    }
}

// *slot = the largest v among the 64 lanes of the wave (all of them call; `leader` is true in
// lane 0 only, which stores)
__device__ __forceinline__ void wave_max_store(uint32_t v, bool leader, uint32_t* slot)
{
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
        v = t > v ? t : v;
    }
    if (leader) *slot = v;
}

}  // namespace psim

// TEST-ONLY counterpart of csrc/lane_hip.h: the kernel bodies run one element after the other in
// a host loop, in ascending element order, so "the wave" is whatever ran before.
#pragma once
#include <stdint.h>

namespace psim {

inline uint32_t atomic_inc_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
inline void aggregated_inc(uint32_t* p) { *p += 1; }
inline void raise_to_u32(uint32_t* p, uint32_t v) { if (v > *p) *p = v; }
inline void fill_run_i32(int32_t* a, int64_t lo, int64_t hi, int32_t v) { for (int64_t c = lo; c <= hi; ++c) a[c] = v; }
// (the leader is the first element of its group of 64 and therefore runs first)
inline void wave_max_store(uint32_t v, bool leader, uint32_t* slot)
{
    if (leader) *slot = 0;
    if (v > *slot) *slot = v;
}

}  // namespace psim

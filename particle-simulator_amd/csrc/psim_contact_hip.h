// Workgroup-cooperative execution of the contact list in dependency order (HIP, gfx950).
//
// A contact runs once the previous contact of each of its two slots has run.  With one lane
// per contact and one launch per dependency level, a clumped state needs 150 launches per
// fixed-point iteration, each looking at every contact.  Here a workgroup owns a tile of
// consecutive contacts and carries the chains INSIDE its tile as far as they go within one
// launch: internal rounds separated by a barrier, a contact running in the round after its
// in-tile predecessors.  A predecessor outside the tile counts only if it ran in an earlier
// launch (nothing is passed between workgroups inside a launch), so the number of launches is
// the depth of the chains across tiles, and a workgroup whose tile has nothing left returns
// after reading one word.
//
// done[h] holds the number of the launch contact h ran in (0: not yet).  The velocity half
// replays the schedule of the position half: launch l takes the contacts with done[h] == l,
// in the same in-tile order (a predecessor with the same label lies in the same tile).
#pragma once
#include "backend_hip.h"
#include "psim_kernels.h"

namespace psim {
namespace be {

constexpr int kRoundTile = 1024;        // contacts per workgroup, four per lane
constexpr uint32_t kNoLabel = 0xffffffffu;

struct RoundTiles {
    uint32_t* left;        // [tiles] contacts of the tile that have not run yet (position half)
    uint32_t* first;       // [tiles] first and last launch in which a contact of the tile ran
    uint32_t* last;
};

// POSITIONS: run what has not run (done == 0), label it.  Otherwise: run what carries this label.
// Body (ContactSimulate / ContactVelocity): State, Pre, prepare(), initial(), apply(), and the
// arrays hsi, hsj, out.  Everything a contact needs from memory is fetched before the rounds
// begin (its slots, what prepare() reads, the states left by predecessors that ran in earlier
// launches); the states passed along inside the tile travel through LDS, so a round costs the
// contact's arithmetic and a barrier, not a trip to memory.
template <bool POSITIONS, class Body>
__global__ void __launch_bounds__(256) contact_rounds_kernel(Body body, uint32_t n, const int32_t* __restrict__ prev,
                                                             uint32_t* done, uint32_t label, RoundTiles tiles)
{
    typedef typename Body::State State;
    typedef typename Body::Pre Pre;
    __shared__ uint16_t s_stamp[kRoundTile];      // internal round in which the tile's contact ran in this launch (0: it has not)
    __shared__ State s_out[2 * kRoundTile];       // [2 t + side] state after the tile's contact t
    __shared__ uint32_t s_any[3];                 // round r: slot r % 3 is set when some contact ran
    const uint32_t tile = blockIdx.x;
    if (POSITIONS) {
        if (label > 1u && tiles.left[tile] == 0u) return;
    } else {
        if (label < tiles.first[tile] || label > tiles.last[tile]) return;      // (first = kNoLabel: nothing ever ran here)
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t lo = tile * (uint32_t)kRoundTile;
    const uint32_t hi = (n - lo < (uint32_t)kRoundTile) ? n : lo + (uint32_t)kRoundTile;
    State I0[4], J0[4];            // the two slots' states when no in-tile predecessor supplies them
    Pre pre[4];
    uint32_t from_i[4], from_j[4]; // 2 * (in-tile predecessor) + side, or kNoLabel: use I0 / J0
    bool mine[4];
    uint32_t blocked = 0;          // contacts of this lane that have to wait for a later launch
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = tid + 256u * (uint32_t)k, h = lo + t;
        s_stamp[t] = 0;
        mine[k] = false; from_i[k] = kNoLabel; from_j[k] = kNoLabel;
        if (h < hi) {
            const uint32_t d = done[h];
            if (POSITIONS ? (d == 0u) : (d == label)) {
                const int32_t pi = prev[2 * (size_t)h], pj = prev[2 * (size_t)h + 1];
                const uint32_t si = body.hsi[h], sj = body.hsj[h];
                bool ok = true;
                // a predecessor: none (the slot's state before any contact); ran in an earlier
                // launch (its output is in memory); lies in this tile and runs in this launch
                // (wait for it); or none of these (then this contact is not for this launch)
                auto look = [&](int32_t p, uint32_t slot, State& st, uint32_t& from) {
                    if (p < 0) { st = body.initial(slot); return; }
                    const uint32_t dp = done[p];
                    const uint32_t side = (body.hsi[p] == slot) ? 0u : 1u;
                    if (dp != 0u && dp < label) { st = body.out[2 * (size_t)p + side]; return; }
                    const bool here = (uint32_t)p >= lo && (uint32_t)p < hi;
                    if (here && (POSITIONS ? (dp == 0u) : (dp == label))) from = 2u * ((uint32_t)p - lo) + side;
                    else ok = false;
                };
                look(pi, si, I0[k], from_i[k]);
                look(pj, sj, J0[k], from_j[k]);
                if (ok) pre[k] = body.prepare((int64_t)h, si, sj);
                mine[k] = ok;
                if (!ok) ++blocked;
            }
        }
    }
    if (tid < 3u) s_any[tid] = 0u;
    __syncthreads();
    bool ran = false;
    for (uint32_t r = 1;; ++r) {
        bool progress = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!mine[k]) continue;
            if (from_i[k] != kNoLabel) { const uint32_t s = s_stamp[from_i[k] >> 1]; if (s == 0u || s >= r) continue; }
            if (from_j[k] != kNoLabel) { const uint32_t s = s_stamp[from_j[k] >> 1]; if (s == 0u || s >= r) continue; }
            const uint32_t t = tid + 256u * (uint32_t)k, h = lo + t;
            State I = from_i[k] != kNoLabel ? s_out[from_i[k]] : I0[k];
            State J = from_j[k] != kNoLabel ? s_out[from_j[k]] : J0[k];
            body.apply((int64_t)h, pre[k], I, J);
            s_out[2 * t] = I;
            s_out[2 * t + 1] = J;
            if (POSITIONS) done[h] = label;
            s_stamp[t] = (uint16_t)r;
            mine[k] = false;
            progress = true;
        }
        ran = ran || progress;
        if (progress) s_any[r % 3u] = 1u;
        // Only LDS has to be in order across the barrier (what went to memory is read by later
        // launches), so the barrier does not wait for this round's stores.  The slot cleared is
        // the one last read after the previous barrier and next written after the next one.
        __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const uint32_t any = s_any[r % 3u];
        if (tid == 0u) s_any[(r + 2u) % 3u] = 0u;
        if (any == 0u || r >= 65534u) break;
    }
    if (POSITIONS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (mine[k]) ++blocked;       // (waiting on an in-tile contact that is itself held up)
        const int left = __syncthreads_count(blocked != 0u);     // zero or not is all that is asked of it
        const int any = __syncthreads_or(ran ? 1 : 0);
        if (tid == 0) {
            tiles.left[tile] = (uint32_t)left;
            if (label == 1u) { tiles.first[tile] = any ? 1u : kNoLabel; tiles.last[tile] = any ? 1u : 0u; }
            else if (any) { if (tiles.first[tile] == kNoLabel) tiles.first[tile] = label; tiles.last[tile] = label; }
        }
    }
}

template <bool POSITIONS, class Body>
inline void launch_contact_rounds(Stream st, const Body& body, uint32_t n, const int32_t* prev, uint32_t* done, uint32_t label,
                                  const RoundTiles& tiles)
{
    if (n == 0) return;
    const uint32_t nt = (n + kRoundTile - 1) / kRoundTile;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(contact_rounds_kernel<POSITIONS, Body>), dim3(nt), dim3(256), 0, st, body, n, prev, done,
                       label, tiles);
    PSIM_HIP(hipGetLastError());
}

}  // namespace be
}  // namespace psim

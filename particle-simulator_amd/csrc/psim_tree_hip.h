// Workgroup-cooperative emission of the pre-order element array (HIP, gfx950): what
// TreeEmitLeaves + TreeEmitNodes + most of the per-depth TreeCom passes do, in one pass that
// writes every element once.
//
// A workgroup takes a tile of consecutive sorted particles.  Their elements (each particle's
// chain of owned internal nodes followed by its leaf) form one contiguous range of the array,
// which is assembled in LDS.  Every internal node whose subtree ends inside the tile then gets
// its centre of mass right there, depth by depth from the bottom, children summed in pre-order
// = quadrant order with TreeCom's arithmetic (quadtree.py:102-114), and the finished range is
// written out with full-width stores.  Only the nodes whose subtree reaches beyond their tile
// (at most one per depth and tile) are left for tree_far_kernel, which settles them all in one
// launch: the nodes a tile leaves over are ancestors of the next tile's first particle, so each
// waits only for deeper nodes of its own tile and for nodes of later tiles.
//
// Neither kernel needs a count from the host: the number of internal nodes is read from the
// scan's total in device memory, and both return at once when that number exceeds the pool
// (more internal nodes than particles) or the key sort was left incomplete -- the host looks
// at the same words afterwards and repeats or reports.
#pragma once
#include "backend_hip.h"
#include "psim_kernels.h"

namespace psim {
namespace be {

constexpr int kEmitTile = 256;          // sorted particles per workgroup, one per lane
constexpr int kEmitCap = 704;           // elements assembled in LDS (22 KiB)
constexpr int kEmitSlots = 32;          // nodes a tile can leave over (one per depth)

// what a left-over node's record holds until tree_far_kernel has settled it; each 64-bit word
// of a record is published on its own, so each has a value of its own that says "not yet"
// (skip = 0 in the last word)
constexpr uint64_t kPendingMass = 0xbff0000000000000ull;        // -1.0
constexpr uint64_t kPendingCoord = 0x7ff8a5a55a5a0001ull;       // a NaN no arithmetic produces

struct TreeEmitArgs {
    int64_t n;
    const uint64_t* key;
    const int8_t* L;
    const uint32_t* base;
    const vec2* spos;
    const double* smass;
    const uint32_t* counts;    // [0] internal nodes (the scan's total), [4] nonzero: keys not fully sorted
    uint32_t root_tag;
    NodeRec* rec;
    uint32_t* node_first;
    uint32_t* node_last;
    uint64_t* far_list;        // [tiles * kEmitSlots]  depth << 32 | node id
    uint32_t* far_count;       // [tiles]
    uint32_t* far_offs;        // [tiles]   exclusive scan of far_count (made between the two kernels)
    uint64_t* far_nodes;       // [tiles * kEmitSlots]  the lists of all tiles, packed in tile order
    uint32_t* words;           // zero before the launch.  [0] number of left-over nodes (the scan's total),
                               // [1] set by tree_far_kernel when it gave up waiting,
                               // [2] set when the caller has to fall back to the per-element kernels: a tile
                               // ran out of room, or a node of the deepest level exists (its children are
                               // buckets of coinciding particles, which the tiles do not handle),
                               // [3] tree_far_kernel's ticket counter
};

// usable tree?  (wave-uniform; both kernels return when not)
__device__ __forceinline__ bool emit_inputs_ok(const TreeEmitArgs& a)
{
    return a.counts[0] <= (uint32_t)a.n && a.counts[4] == 0u;
}

__global__ void __launch_bounds__(kEmitTile) tree_emit_tile_kernel(TreeEmitArgs a)
{
    __shared__ NodeRec tile[kEmitCap];
    __shared__ uint64_t s_key[kEmitTile];
    __shared__ uint32_t s_base[kEmitTile + 1];
    __shared__ int8_t s_L[kEmitTile + 8];          // s_L[i] = L[s0 - 1 + i]
    __shared__ uint32_t s_far, s_dmin, s_dmax;
    if (!emit_inputs_ok(a)) return;
    const int tid = threadIdx.x;
    const uint32_t n = (uint32_t)a.n;
    const uint32_t total_elems = n + a.counts[0];
    const uint32_t s0 = blockIdx.x * (uint32_t)kEmitTile;
    const uint32_t s1 = s0 + kEmitTile < n ? s0 + kEmitTile : n;
    const uint32_t s = s0 + (uint32_t)tid;
    const bool live = s < s1;
    // the tile's keys, shared-level counts and node-id bases (and the next tile's first base)
    if (live) { s_key[tid] = a.key[s]; s_L[tid + 1] = a.L[s]; }
    if (s <= s1 && s < n) s_base[tid] = a.base[s];
    if (tid == 0) {
        s_L[0] = s0 > 0 ? a.L[s0 - 1] : (int8_t)-1;
        if (s1 < n && s1 == s0 + kEmitTile) s_base[kEmitTile] = a.base[s1];
        s_far = 0; s_dmin = 64; s_dmax = 0;
    }
    const bool more = s1 < n;
    const uint64_t key_next = more ? a.key[s1] : 0ull;          // first particle of the next tile
    // (the leaf's own data, asked for now so that it is there when the leaf is assembled)
    double leaf_m = 0.0;
    vec2 leaf_p;
    leaf_p.x = 0.0; leaf_p.y = 0.0;
    if (live) { leaf_m = a.smass[s]; leaf_p = a.spos[s]; }
    __syncthreads();
    const uint32_t e0 = s0 + s_base[0];
    const uint32_t e1 = more ? s1 + s_base[s1 - s0] : total_elems;
    // (a tile that does not fit is not assembled at all: the caller falls back)
    const bool fits = e1 - e0 <= (uint32_t)kEmitCap;
    if (!fits) {
        if (tid == 0) { raise_to_u32(&a.words[2], 1u); a.far_count[blockIdx.x] = 0; }
        return;
    }
    int lp = 0, l = -1;
    uint32_t b = 0;
    if (live) {
        lp = (int)s_L[tid];
        l = (int)s_L[tid + 1];
        b = s_base[tid];
        const int own = l > lp ? l - lp : 0;
        // levels this particle shares with the next tile's first one: its cells down to that
        // depth contain that particle too, i.e. reach beyond the tile
        const int reach = more ? shared_levels(s_key[tid], key_next) : -1;
        const uint32_t in_tile = s1 - s0;
        uint32_t lmin = 64, lmax = 0;                            // depths of this lane's local nodes
        for (int k = 0; k < own; ++k) {
            const int d = lp + 1 + k;
            const uint32_t id = b + (uint32_t)k;
            NodeRec r;
            r.mass = 0.0; r.x = 0.0; r.y = 0.0;
            r.skip = 0;
            r.tag = node_tag(a.root_tag, d);
            a.node_first[id] = s;
            if (d >= kMaxInternalDepth) raise_to_u32(&a.words[2], 1u);
            if (d <= reach || d >= kMaxInternalDepth) {
                // the range is found and the centre of mass formed by tree_far_kernel
                const uint32_t slot = atomicAdd(&s_far, 1u);
                if (slot < (uint32_t)kEmitSlots)
                    a.far_list[(size_t)blockIdx.x * kEmitSlots + slot] = ((uint64_t)(uint32_t)d << 32) | (uint64_t)id;
                r.mass = __longlong_as_double((long long)kPendingMass);
                r.x = __longlong_as_double((long long)kPendingCoord);
                r.y = r.x;
            } else {
                // last particle of the cell: the first t >= s whose successor shares fewer than d
                // levels with it (L[t] < d); it lies inside the tile
                uint32_t t = (uint32_t)tid;              // tile-relative
                int steps = 0;
                while (steps < 8 && (int)s_L[t + 1] >= d) { ++t; ++steps; }
                if (d == 0) {
                    t = in_tile - 1;                     // the root, when one tile holds everything
                } else if ((int)s_L[t + 1] >= d) {
                    // a larger cell: gallop, then bisect, on the tile's keys
                    const int sh = 64 - 2 * d;
                    const uint64_t pre = s_key[tid] >> sh;
                    uint32_t lo = t, hi, step = 1;
                    for (;;) {
                        const uint32_t u = lo + step;
                        if (u >= in_tile) { hi = in_tile; break; }
                        if ((s_key[u] >> sh) != pre) { hi = u; break; }
                        lo = u;
                        step <<= 1;
                    }
                    while (hi - lo > 1) {
                        const uint32_t mid = lo + (hi - lo) / 2;
                        if ((s_key[mid] >> sh) == pre) lo = mid; else hi = mid;
                    }
                    t = lo;
                }
                const uint32_t e = s0 + t;
                r.skip = (e + 1 < n) ? (e + 1) + s_base[t + 1] : total_elems;
                a.node_last[id] = e;
                lmin = lmin < (uint32_t)d ? lmin : (uint32_t)d;
                lmax = (uint32_t)d + 1u;
            }
            tile[s + id - e0] = r;
        }
        if (lmax) { atomicMin(&s_dmin, lmin); atomicMax(&s_dmax, lmax); }
        const uint32_t le = s + b + (uint32_t)own;
        NodeRec r;
        r.mass = leaf_m;
        r.x = leaf_p.x;
        r.y = leaf_p.y;
        r.skip = le + 1;
        r.tag = kLeafTag;
        tile[le - e0] = r;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t far = s_far;
        if (far > (uint32_t)kEmitSlots) { far = kEmitSlots; raise_to_u32(&a.words[2], 1u); }
        a.far_count[blockIdx.x] = far;
    }
    // centres of mass of the nodes whose subtree lies inside the tile, deepest level first
    // (This loop is 30 % of the kernel: 130 us without it against 185 us at 4 Mi particles.
    // Settling the tile's nodes one per lane instead of one particle's chain per lane, so that
    // no wave loops as long as its busiest lane, changed nothing; forming the centres of mass of
    // cells that hold only leaves during assembly, from memory, made the kernel 230 us.)
    const int dmin = (int)s_dmin, dmax = (int)s_dmax;
    for (int d = dmax - 1; d >= dmin; --d) {
        if (d > lp && d <= l) {                          // this particle owns a node of depth d
            const uint32_t at = s + b + (uint32_t)(d - lp - 1) - e0;
            const uint32_t end = tile[at].skip;
            if (end != 0u) {                             // (0: left to tree_far_kernel)
                uint32_t k = at + 1;
                const uint32_t stop = end - e0;
                double total = 0.0, nx = 0.0, ny = 0.0;
                for (int q = 0; q < 4 && k < stop; ++q) {
                    const NodeRec c = tile[k];
                    if (c.mass > 0) {
                        total += c.mass;
                        nx += c.x * c.mass;
                        ny += c.y * c.mass;
                    }
                    const uint32_t nxt = c.skip - e0;
                    k = nxt > k ? nxt : k + 1;
                }
                tile[at].mass = total;
                if (total > 0) { tile[at].x = nx / total; tile[at].y = ny / total; }
            }
        }
        __syncthreads();
    }
    // the finished range, 16 bytes per lane and store
    const uint32_t quads = (e1 - e0) * 2u;
    const double2* src = reinterpret_cast<const double2*>(tile);
    double2* dst = reinterpret_cast<double2*>(a.rec + e0);
    for (uint32_t i = tid; i < quads; i += kEmitTile) dst[i] = src[i];
}

// The tiles' lists packed into one, in tile order: lane = (tile, slot).  An entry keeps its
// tile: depth << 58 | tile << 32 | node id.
struct TreeFarPack {
    int64_t count;             // tiles * kEmitSlots
    TreeEmitArgs a;
    __device__ void operator()(int64_t t) const
    {
        if (!emit_inputs_ok(a) || a.words[2] != 0u) return;
        const uint32_t tile = (uint32_t)(t / kEmitSlots), slot = (uint32_t)(t % kEmitSlots);
        if (slot >= a.far_count[tile]) return;
        const uint64_t e = a.far_list[t];
        a.far_nodes[a.far_offs[tile] + slot] = ((e >> 32) << 58) | ((uint64_t)tile << 32) | (e & 0xffffffffull);
    }
};

// The nodes the tiles left over, all of them in one launch, one lane per node.  Workgroups
// take chunks of 256 nodes from the END of the packed list, in the order of a ticket counter,
// until the list is used up.  A left-over node of tile t is an ancestor of tile t+1's first
// particle that begins in tile t, so its children are: subtrees that end inside tile t
// (finished by tree_emit_tile_kernel), left-over nodes of tile t one level down, and nodes
// owned by particles of later tiles (finished by the emit kernel or left over there).  What a
// node waits for therefore lies later in the list: in its own chunk or in a chunk whose
// ticket was drawn earlier, by a workgroup that is running.  A lane that finds a child
// unsettled simply tries again on the wave's next turn, so lanes never block one another.
//
// Records travel between workgroups through memory: every 64-bit word of a left-over node's
// record is written and read with device-scope atomic accesses (which bypass the per-XCD L2
// caches), and each word has its own "not yet" value, so no ordering between the words is
// needed.  Sums are formed in child order with TreeCom's arithmetic (quadtree.py:102-114).
__global__ void __launch_bounds__(256) tree_far_kernel(TreeEmitArgs a)
{
    __shared__ uint32_t s_ticket;
    if (!emit_inputs_ok(a) || a.words[2] != 0u) return;
    const uint32_t n = (uint32_t)a.n;
    const uint32_t total_elems = n + a.counts[0];
    const uint32_t far_total = a.words[0];
    unsigned long long* words = reinterpret_cast<unsigned long long*>(a.rec);      // four per record: mass, x, y, skip | tag << 32
  for (;;) {
    __syncthreads();               // (everyone has read the previous ticket)
    if (threadIdx.x == 0) s_ticket = atomicAdd(&a.words[3], 1u);
    __syncthreads();
    const uint64_t back = (uint64_t)s_ticket * 256u + threadIdx.x;           // nodes counted from the end of the list
    if ((uint64_t)s_ticket * 256u >= far_total) return;
    bool done = true;
    uint32_t at = 0, end = 0;
    if (back < far_total) {
        {
            const uint64_t entry = a.far_nodes[far_total - 1u - (uint32_t)back];
            const int d = (int)(entry >> 58);
            const uint32_t tile = (uint32_t)(entry >> 32) & 0x3ffffffu;
            const uint32_t id = (uint32_t)entry;
            const uint32_t s = a.node_first[id];
            // Where the cell ends: the last particle whose key starts with the cell's path; it
            // lies at or after the first particle of the next tile.  Eight probes per round
            // trip to memory (the cells near the root span millions of keys).
            uint32_t e = n - 1;
            if (d > 0) {
                const int sh = 64 - 2 * d;
                const uint64_t pre = a.key[s] >> sh;
                uint32_t lo = (tile + 1u) * (uint32_t)kEmitTile;         // inside the cell
                if (lo >= n) lo = s;
                uint32_t hi = n;                                        // first position known (or assumed) outside
                while (hi - lo > 1u) {
                    const uint32_t span = hi - lo;
                    uint32_t probe[7];
                    bool inside[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        probe[j] = lo + (uint32_t)(((uint64_t)span * (uint32_t)(j + 1)) >> 3);
                        inside[j] = (a.key[probe[j]] >> sh) == pre;
                    }
                    uint32_t nlo = lo, nhi = hi;
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        if (inside[j]) { if (probe[j] > nlo) nlo = probe[j]; }
                        else if (probe[j] < nhi && probe[j] > lo) nhi = probe[j];
                    }
                    // (the keys are sorted, so the probes inside come before the probes outside)
                    lo = nlo;
                    hi = nhi;
                }
                e = lo;
            }
            end = (e + 1 < n) ? (e + 1) + a.base[e + 1] : total_elems;
            a.node_last[id] = e;
            at = s + id;
            __hip_atomic_store(words + 4 * (size_t)at + 3, (unsigned long long)end | ((unsigned long long)node_tag(a.root_tag, d) << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            done = false;
        }
    }
    // The children: element at + 1, then wherever each child's subtree ends, up to four.  A
    // child's index is known once its predecessor's link is; its record is taken once all
    // four words are there.  Every turn asks for everything still missing at once, so a turn
    // is one round trip to memory however many children are outstanding.
    uint32_t ck[4] = {at + 1u, 0u, 0u, 0u};
    double cm[4] = {0.0, 0.0, 0.0, 0.0}, cx[4] = {0.0, 0.0, 0.0, 0.0}, cy[4] = {0.0, 0.0, 0.0, 0.0};
    int known = done ? 0 : 1;           // children whose index is known
    int have = 0;                       // bit q: child q's record has been taken
    int asked = 0;                      // bit q: child q has been looked at and was not settled
    bool last_known = false;            // the last child's index is among the known ones
    for (uint32_t turn = 0;; ++turn) {
        if (!done) {
            unsigned long long w[4][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < known && !((have >> q) & 1)) {
                    const unsigned long long* c = words + 4 * (size_t)ck[q];
                    // A child seen unsettled before is watched through its mass word alone (the
                    // last one its owner writes) until that changes: tens of thousands of lanes
                    // wait at once, and four words per turn from each of them is what the
                    // memory system then spends its time on.
                    if (!((asked >> q) & 1)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[q][j] = __hip_atomic_load(c + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        w[q][0] = __hip_atomic_load(c + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w[q][1] = kPendingCoord; w[q][2] = kPendingCoord; w[q][3] = 0ull;
                        if (w[q][0] != kPendingMass) {
#pragma unroll
                            for (int j = 1; j < 4; ++j) w[q][j] = __hip_atomic_load(c + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            const int known_now = known;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < known_now && !((have >> q) & 1)) {
                    const uint32_t link = (uint32_t)w[q][3];
                    if (link != 0u) {
                        if (q + 1 == known && !last_known) {
                            // the next child's index, or the end of the list
                            const uint32_t nxt = link > ck[q] ? link : ck[q] + 1u;
                            if (q < 3 && nxt < end) { ck[(q + 1) & 3] = nxt; known = q + 2; }
                            else last_known = true;
                        }
                        if (w[q][0] != kPendingMass && w[q][1] != kPendingCoord && w[q][2] != kPendingCoord) {
                            cm[q] = __longlong_as_double((long long)w[q][0]);
                            cx[q] = __longlong_as_double((long long)w[q][1]);
                            cy[q] = __longlong_as_double((long long)w[q][2]);
                            have |= 1 << q;
                        } else {
                            asked |= 1 << q;
                        }
                    }
                }
            }
            if (last_known && have == (1 << known) - 1) {
                double total = 0.0, nx = 0.0, ny = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < known && cm[q] > 0) {
                        total += cm[q];
                        nx += cx[q] * cm[q];
                        ny += cy[q] * cm[q];
                    }
                }
                double x = 0.0, y = 0.0;
                if (total > 0) { x = nx / total; y = ny / total; }
                unsigned long long* o = words + 4 * (size_t)at;
                __hip_atomic_store(o + 1, (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 2, (unsigned long long)__double_as_longlong(y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 0, (unsigned long long)__double_as_longlong(total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
            }
        }
        if (__ballot(!done) == 0ull) break;
        if (turn > (1u << 22)) {
            // (seconds of waiting: cannot happen on a well-formed tree; reported instead of hanging)
            if (!done) raise_to_u32(&a.words[1], 1u);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (__hip_atomic_load(&a.words[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;      // (gave up: no point in going on)
  }
}

inline uint32_t emit_tiles(int64_t n) { return (uint32_t)((n + kEmitTile - 1) / kEmitTile); }

// scan_temp: scratch of exclusive_scan_u32 for `tiles` items (zero, and left zero)
inline void launch_tree_emit(Stream st, const TreeEmitArgs& a, uint32_t* scan_temp)
{
    if (a.n <= 0) return;
    const uint32_t tiles = emit_tiles(a.n);
    hipLaunchKernelGGL(tree_emit_tile_kernel, dim3(tiles), dim3(kEmitTile), 0, st, a);
    PSIM_HIP(hipGetLastError());
    // the tiles' left-over nodes packed into one list, then settled by workgroups that keep
    // drawing chunks of it (as many workgroups as can be resident, or fewer for a short list)
    exclusive_scan_u32(st, a.far_count, a.far_offs, (uint64_t)tiles, scan_temp, a.words + 0);
    launch(st, (int64_t)tiles * kEmitSlots, TreeFarPack{(int64_t)tiles * kEmitSlots, a});
    const uint64_t chunks = ((uint64_t)tiles * kEmitSlots + 255u) / 256u;
    const uint32_t groups = (uint32_t)(chunks < 2048u ? chunks : 2048u);
    hipLaunchKernelGGL(tree_far_kernel, dim3(groups), dim3(256), 0, st, a);
    PSIM_HIP(hipGetLastError());
}

}  // namespace be
}  // namespace psim

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
// (at most one per depth and tile, plus depth-31 nodes, whose children are buckets) are left
// for per-depth passes over a short list (TreeComFar).
#pragma once
#include "backend_hip.h"
#include "psim_kernels.h"

namespace psim {
namespace be {

constexpr int kEmitTile = 256;          // sorted particles per workgroup, one per lane
constexpr int kEmitCap = 704;           // elements assembled in LDS (22 KiB)
constexpr int kEmitSlots = 32;          // nodes a tile can leave to the far passes (one per depth)

struct TreeEmitArgs {
    int64_t n;
    const uint64_t* key;
    const int8_t* L;
    const uint32_t* base;
    const vec2* spos;
    const double* smass;
    uint32_t total_elems;
    uint32_t root_tag;
    NodeRec* rec;
    uint32_t* node_first;
    uint32_t* node_last;
    uint64_t* far_list;        // [tiles * kEmitSlots]  depth << 32 | node id
    uint32_t* far_count;       // [tiles]
    uint32_t* words;           // zero before the launch.  [0] 64 - smallest far depth, [1] largest far depth + 1,
                               // [2] set when the caller has to fall back to the per-element kernels: a tile
                               // ran out of room, or a node of the deepest level exists (its children are
                               // buckets of coinciding particles, which the tiles do not handle)
};

__global__ void __launch_bounds__(kEmitTile) tree_emit_tile_kernel(TreeEmitArgs a)
{
    __shared__ NodeRec tile[kEmitCap];
    __shared__ uint64_t s_key[kEmitTile];
    __shared__ uint32_t s_base[kEmitTile + 1];
    __shared__ int8_t s_L[kEmitTile + 8];          // s_L[i] = L[s0 - 1 + i]
    __shared__ uint32_t s_far, s_dmin, s_dmax, s_fmin, s_fmax;
    const int tid = threadIdx.x;
    const uint32_t n = (uint32_t)a.n;
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
        s_far = 0; s_dmin = 64; s_dmax = 0; s_fmin = 64; s_fmax = 0;
    }
    const bool more = s1 < n;
    const uint64_t key_next = more ? a.key[s1] : 0ull;          // first particle of the next tile
    __syncthreads();
    const uint32_t e0 = s0 + s_base[0];
    const uint32_t e1 = more ? s1 + s_base[s1 - s0] : a.total_elems;
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
        uint32_t lmin = 64, lmax = 0, fmin = 64, fmax = 0;       // depths of this lane's local / far nodes
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
                // the range is found and the centre of mass formed by the far passes
                const uint32_t slot = atomicAdd(&s_far, 1u);
                if (slot < (uint32_t)kEmitSlots)
                    a.far_list[(size_t)blockIdx.x * kEmitSlots + slot] = ((uint64_t)(uint32_t)d << 32) | (uint64_t)id;
                fmin = fmin < (uint32_t)d ? fmin : (uint32_t)d;
                fmax = (uint32_t)d + 1u;
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
                r.skip = (e + 1 < n) ? (e + 1) + s_base[t + 1] : a.total_elems;
                a.node_last[id] = e;
                lmin = lmin < (uint32_t)d ? lmin : (uint32_t)d;
                lmax = (uint32_t)d + 1u;
            }
            tile[s + id - e0] = r;
        }
        if (lmax) { atomicMin(&s_dmin, lmin); atomicMax(&s_dmax, lmax); }
        if (fmax) { atomicMin(&s_fmin, fmin); atomicMax(&s_fmax, fmax); }
        const uint32_t le = s + b + (uint32_t)own;
        NodeRec r;
        r.mass = a.smass[s];
        r.x = a.spos[s].x;
        r.y = a.spos[s].y;
        r.skip = le + 1;
        r.tag = kLeafTag;
        tile[le - e0] = r;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t far = s_far;
        if (far > (uint32_t)kEmitSlots) { far = kEmitSlots; raise_to_u32(&a.words[2], 1u); }
        a.far_count[blockIdx.x] = far;
        if (s_fmin < 64u) { raise_to_u32(&a.words[0], 64u - s_fmin); raise_to_u32(&a.words[1], s_fmax); }      // (few tiles get past the look)
    }
    // centres of mass of the nodes whose subtree lies inside the tile, deepest level first
    const int dmin = (int)s_dmin, dmax = (int)s_dmax;
    for (int d = dmax - 1; d >= dmin; --d) {
        if (d > lp && d <= l) {                          // this particle owns a node of depth d
            const uint32_t at = s + b + (uint32_t)(d - lp - 1) - e0;
            const uint32_t end = tile[at].skip;
            if (end != 0u) {                             // (0: left to the far passes)
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

// Where the cells the tiles left over end (they reach into later tiles): lane = (tile, slot).
struct TreeFarRanges {
    int64_t count;             // tiles * kEmitSlots
    int64_t n;
    const uint64_t* far_list;
    const uint32_t* far_count;
    const uint64_t* key;
    const uint32_t* base;
    const uint32_t* node_first;
    uint32_t total_elems;
    NodeRec* rec;
    uint32_t* node_last;
    __device__ void operator()(int64_t t) const
    {
        const uint32_t tile = (uint32_t)(t / kEmitSlots), slot = (uint32_t)(t % kEmitSlots);
        if (slot >= far_count[tile]) return;
        const uint64_t entry = far_list[t];
        const int d = (int)(entry >> 32);
        const uint32_t id = (uint32_t)entry;
        const uint32_t s = node_first[id];
        // the cell contains the first particle of the next tile
        uint32_t from = (tile + 1u) * (uint32_t)kEmitTile;
        if ((int64_t)from >= n || d >= kMaxInternalDepth) from = s;
        const uint32_t e = range_last(key, s, (uint32_t)n, d, from);
        rec[s + id].skip = (e + 1 < (uint32_t)n) ? (e + 1) + base[e + 1] : total_elems;
        node_last[id] = e;
    }
};

// The nodes the tiles left over, one depth per launch (deepest first): lane = (tile, slot).
struct TreeComFar {
    int64_t count;             // tiles * kEmitSlots
    const uint64_t* far_list;
    const uint32_t* far_count;
    uint32_t depth;
    NodeChildren ch;
    NodeRec* rec;
    __device__ void operator()(int64_t t) const
    {
        const uint32_t tile = (uint32_t)(t / kEmitSlots), slot = (uint32_t)(t % kEmitSlots);
        if (slot >= far_count[tile]) return;
        const uint64_t entry = far_list[t];
        if ((uint32_t)(entry >> 32) != depth) return;
        const uint32_t id = (uint32_t)entry;
        double cm[4], cx[4], cy[4];
        ch.load(id, (int)depth, cm, cx, cy);
        double total = 0.0, nx = 0.0, ny = 0.0;
        for (int q = 0; q < 4; ++q) {
            if (cm[q] > 0) {
                total += cm[q];
                nx += cx[q] * cm[q];
                ny += cy[q] * cm[q];
            }
        }
        NodeRec& out = rec[ch.node_first[id] + id];
        out.mass = total;
        if (total > 0) { out.x = nx / total; out.y = ny / total; }
    }
};

inline uint32_t emit_tiles(int64_t n) { return (uint32_t)((n + kEmitTile - 1) / kEmitTile); }

inline void launch_tree_emit(Stream st, const TreeEmitArgs& a)
{
    if (a.n <= 0) return;
    hipLaunchKernelGGL(tree_emit_tile_kernel, dim3(emit_tiles(a.n)), dim3(kEmitTile), 0, st, a);
    PSIM_HIP(hipGetLastError());
}

}  // namespace be
}  // namespace psim

// Per-element kernel bodies of the physics step.
//
// Every struct below is the body of one data-parallel kernel: operator()(i) is
// the work of element i.  backend_hip.h runs them as HIP kernels (one lane per
// element); the test-only host-loop build runs the same bodies in a for loop so
// the step's logic can be checked on a machine without a GPU.
//
// Reference citations are file:line into NeoLorenzo/Particle-Simulator.
// The library is compiled with -ffp-contract=off: wherever the reference stage
// is strict IEEE FP64 the expression order below IS the contract.
#pragma once
#include <math.h>
#include "psim_types.h"

namespace psim {

// --------------------------------------------------------------- helpers
// (atomic_inc_u32, clz64, aggregated_inc, raise_to_u32, fill_run_i32, wave_max_store: lane_hip.h)

// quadtree.py:13-19 (_get_quadrant) + :24-38,62-63 (child boxes, centre = x + w/2):
// descend 32 levels from the root box exactly as an insertion would, recording
// quadrant = 2*(y >= cy) + (x >= cx) per level, most significant first.
// Coordinates outside the box keep comparing against centres, so keys saturate.
PSIM_HD PSIM_FORCEINLINE uint64_t morton_key(double x, double y, double W, double H)
{
    double x0 = 0.0, y0 = 0.0, w = W, h = H;
    uint64_t key = 0;
    for (int l = 0; l < kKeyLevels; ++l) {
        double hw = w / 2, hh = h / 2;
        double cx = x0 + hw, cy = y0 + hh;
        uint64_t bx = (x < cx) ? 0u : 1u;
        uint64_t by = (y < cy) ? 0u : 1u;
        key = (key << 2) | (by << 1) | bx;
        if (bx) x0 = cx;
        if (by) y0 = cy;
        w = hw;
        h = hh;
    }
    return key;
}

// particle_system.py:744-748: (x / cell_size).astype(int32) then clip.  The
// x86 conversion numpy uses yields INT32_MIN for NaN and out-of-range values.
PSIM_HD PSIM_FORCEINLINE int32_t cell_coord(double v, double cell_size, int32_t g)
{
    double q = v / cell_size;
    int32_t c;
    if (!(q > -2147483649.0 && q < 2147483648.0)) c = INT32_MIN;
    else c = (int32_t)q;
    if (c < 0) c = 0;
    if (c > g - 1) c = g - 1;
    return c;
}

PSIM_HD PSIM_FORCEINLINE int shared_levels(uint64_t a, uint64_t b)
{
    int s = clz64(a ^ b) >> 1;          // identical keys: 32
    return s > kMaxInternalDepth ? kMaxInternalDepth : s;
}

// --------------------------------------------------------------- stage 1/2: drift, keys
// particle_system.py:528 (drift, dt = 1) fused with the Morton key (quadtree
// insertion path) and the grid cell id (particle_system.py:744-749).
struct DriftKeys {
    int64_t n;
    vec2* pos;
    const vec2* vel;
    const vec2* acc;
    uint64_t* key;
    uint32_t* perm;
    uint32_t* ckey;
    uint32_t* cperm;
    double W, H, cell_size;
    int32_t gw, gh;
    int32_t do_drift;
    const double* mass;
    const int32_t* radii;
    PackedBody* pack;           // the gathers' source (see PackedBody)
    PSIM_HD void operator()(int64_t i) const
    {
        vec2 p = pos[i];
        if (do_drift) {
            vec2 v = vel[i], a = acc[i];
            p.x = p.x + (v.x + 0.5 * a.x);
            p.y = p.y + (v.y + 0.5 * a.y);
            pos[i] = p;
        }
        PackedBody b;
        b.x = p.x; b.y = p.y; b.m = mass[i]; b.r = radii[i]; b.unused = 0u;
        pack[i] = b;
        key[i] = morton_key(p.x, p.y, W, H);
        perm[i] = (uint32_t)i;
        int32_t cx = cell_coord(p.x, cell_size, gw);
        int32_t cy = cell_coord(p.y, cell_size, gh);
        ckey[i] = (uint32_t)(cy * gw + cx);
        cperm[i] = (uint32_t)i;
    }
};

// CSR offsets of a sorted u32 key array: offsets[c] = first k with keys[k] >= c
// (particle_system.py:752-755 bincount + cumsum, without the histogram).
// Run over n + 1 elements.
struct SegmentOffsets {
    int64_t n;
    const uint32_t* keys;
    int32_t* offsets;
    uint32_t num_segments;
    PSIM_HD void operator()(int64_t k) const
    {
        int64_t lo = (k == 0) ? 0 : (int64_t)keys[k - 1] + 1;
        int64_t hi = (k == n) ? (int64_t)num_segments : (int64_t)keys[k];
        fill_run_i32(offsets, lo, hi, (int32_t)k);
    }
};

// The segments after the last key (SegmentOffsets' element k = n), one lane per segment: that
// element would sit alone in the launch's last wave and fill them one by one (whole empty grid
// rows below a disk of particles).
struct SegmentOffsetsTail {
    int64_t n;
    const uint32_t* keys;
    int32_t* offsets;
    uint32_t num_segments;
    PSIM_HD void operator()(int64_t c) const
    {
        if (c > (int64_t)keys[n - 1]) offsets[c] = (int32_t)n;      // c = 0 .. num_segments
    }
};

// largest cell population without same-address atomics: WaveMaxOccupancy (over cells
// rounded up to whole waves) stores one maximum per 64 cells, WaveMaxU32 folds such a
// list 64-fold per launch, FinalMax (one wave) folds what is left.
struct WaveMaxOccupancy {
    int64_t cells;
    const int32_t* offsets;
    uint32_t* wave_max;        // [ceil(cells / 64)]
    PSIM_HD void operator()(int64_t c) const
    {
        uint32_t occ = c < cells ? (uint32_t)(offsets[c + 1] - offsets[c]) : 0u;
        wave_max_store(occ, (c & 63) == 0, &wave_max[c >> 6]);
    }
};
struct WaveMaxU32 {        // launched over count rounded up to whole waves
    int64_t count;
    const uint32_t* in;
    uint32_t* out;             // [ceil(count / 64)]
    PSIM_HD void operator()(int64_t c) const
    {
        uint32_t v = c < count ? in[c] : 0u;
        wave_max_store(v, (c & 63) == 0, &out[c >> 6]);
    }
};
struct FinalMax {          // launched over exactly 64 elements
    int64_t count;
    const uint32_t* wave_max;
    uint32_t* out;
    PSIM_HD void operator()(int64_t lane) const
    {
        uint32_t m = 0;
        for (int64_t i = lane; i < count; i += 64) { uint32_t v = wave_max[i]; m = v > m ? v : m; }
        wave_max_store(m, lane == 0, out);
    }
};

// cell-sorted ("slot space") records used by the neighbour sweeps
struct GatherCell {
    int64_t n;
    const uint32_t* cperm;
    const uint32_t* ckey;
    const PackedBody* pack;
    CellBody* body;
    int32_t gw;
    uint32_t cell_base;        // the keys count cells from this one on (a band of rows; 0: the whole grid)
    PSIM_HD void operator()(int64_t k) const
    {
        uint32_t p = cperm[k];
        PackedBody P = pack[p];
        CellBody b;
        uint32_t c = ckey[k] + cell_base;
        b.x = P.x; b.y = P.y; b.r = P.r; b.idx = p; b.cell = c;
        b.cxy = (c % (uint32_t)gw) | ((c / (uint32_t)gw) << 16);
        body[k] = b;
    }
};

// A band of whole cell rows [row_lo, row_hi) of the grid: which particles lie in it (flag,
// scanned by the caller), then their cell ids relative to the band's first cell and their
// indices, in ascending particle order (so that the stable sort by cell leaves each cell's
// particles in ascending index, particle_system.py:759-763).
struct BandFlag {
    int64_t n;
    const uint32_t* cell;      // global cell id per particle (DriftKeys)
    int32_t gw, row_lo, row_hi;
    uint32_t* flag;
    PSIM_HD void operator()(int64_t i) const
    {
        int32_t row = (int32_t)(cell[i] / (uint32_t)gw);
        flag[i] = (row >= row_lo && row < row_hi) ? 1u : 0u;
    }
};
struct BandCompact {
    int64_t n;
    const uint32_t* cell;
    const uint32_t* flag;
    const uint32_t* at;        // exclusive scan of flag
    uint32_t cell_base;
    uint32_t* key_out;
    uint32_t* perm_out;
    PSIM_HD void operator()(int64_t i) const
    {
        if (!flag[i]) return;
        uint32_t o = at[i];
        key_out[o] = cell[i] - cell_base;
        perm_out[o] = (uint32_t)i;
    }
};

// --------------------------------------------------------------- stage 3: linear quadtree
// Position of the point (ix, iy) of a 2^levels x 2^levels grid along the Hilbert curve of
// that order.  Consecutive points along the curve are neighbours in space, whereas a Z-order
// run jumps across the domain wherever it crosses a coarse cell boundary.
PSIM_HD PSIM_FORCEINLINE uint32_t hilbert_index(uint32_t ix, uint32_t iy, int levels)
{
    uint32_t h = 0, sw = 0, fl = 0;            // the curve's orientation so far: axes swapped / flipped
    for (int l = levels - 1; l >= 0; --l) {
        uint32_t bx = ((ix >> l) & 1u) ^ fl, by = ((iy >> l) & 1u) ^ fl;
        uint32_t rx = sw ? by : bx, ry = sw ? bx : by;
        h = (h << 2) | ((3u * rx) ^ ry);
        fl ^= (ry ^ 1u) & rx;
        sw ^= ry ^ 1u;
    }
    return h;
}

struct GatherTree {
    int64_t n;
    const uint32_t* perm;
    const PackedBody* pack;
    vec2* spos;
    double* smass;
    PSIM_HD void operator()(int64_t s) const
    {
        PackedBody P = pack[perm[s]];
        vec2 q;
        q.x = P.x; q.y = P.y;
        spos[s] = q;
        smass[s] = P.m;
    }
};

// The walk's targets in Hilbert order, without a sort: the order in which the walk groups its
// targets, so that the 64 targets of a wave are neighbours in space and disagree less about
// which nodes to open (fewer loop trips, fuller force blocks; only the grouping depends on
// it, no result does).  The particles of one quadtree cell of a given depth are consecutive
// in Morton order, and where each such cell comes along the curve does not depend on the
// particles: `curve_pos` is a table made once per domain and depth (Sim::curve_table).  So
// the order is: cells along the curve, each cell's particles in their Morton order.  Three
// passes: where each cell's run starts and ends in the sorted keys; the runs' lengths filed
// under the cells' curve positions (then scanned by the caller); every particle to its place.  `skip` (when set and nonzero: the keys are not fully
// sorted, the caller is about to sort them again) makes all three do nothing.
struct HilbertCellRuns {
    int64_t n;
    const uint64_t* key;
    int shift;                  // 64 - 2 * levels
    uint32_t* run_start;        // [cells], zero before
    uint32_t* run_end;          // [cells], zero before
    const uint32_t* skip;
    PSIM_HD void operator()(int64_t s) const
    {
        if (*skip) return;
        uint32_t c = (uint32_t)(key[s] >> shift);
        if (s == 0) {
            run_start[c] = 0u;
        } else {
            uint32_t before = (uint32_t)(key[s - 1] >> shift);
            if (before != c) { run_start[c] = (uint32_t)s; run_end[before] = (uint32_t)s; }
        }
        if (s == n - 1) run_end[c] = (uint32_t)n;
    }
};

struct HilbertCellCounts {
    int64_t cells;
    const uint32_t* curve_pos;  // [cells]: position along the curve of the cell with this Morton path
    const uint32_t* run_start;
    const uint32_t* run_end;
    uint32_t* count_on_curve;   // [cells]
    const uint32_t* skip;
    PSIM_HD void operator()(int64_t c) const
    {
        if (*skip) return;
        count_on_curve[curve_pos[c]] = run_end[c] - run_start[c];
    }
};

struct HilbertOrder {
    int64_t n;
    const uint64_t* key;
    int shift;
    const uint32_t* curve_pos;
    const uint32_t* run_start;
    const uint32_t* start_on_curve;     // exclusive scan of count_on_curve
    uint32_t* order;                    // [n]: sorted positions of the particles, along the curve
    const uint32_t* skip;
    PSIM_HD void operator()(int64_t s) const
    {
        if (*skip) return;
        uint32_t c = (uint32_t)(key[s] >> shift);
        order[start_on_curve[curve_pos[c]] + ((uint32_t)s - run_start[c])] = (uint32_t)s;
    }
};

// Second half of the Morton sort.  The radix passes order the keys by their leading
// bits only (key >> shift: enough levels that few particles share them); particles
// that do share them form short runs, which the lane at each run's start finishes in
// place by a stable insertion on the full key (ties keep their order, like the stable
// radix passes).  A run longer than kMaxFixRun is left as it is and reported, and the
// caller completes the sort with the remaining radix passes.  In place: nothing is
// written where no run needs it (three quarters of the particles are alone in their
// cell of that depth); the lanes that decide whether they start a run look at leading
// bits only, which no reordering inside a run changes.
// (Doing the gather into sorted order in this pass as well -- whoever places a particle index
// fetches that particle's record -- was measured: 164 us against 46 + 94 for the two kernels,
// the lanes that finish a run fetch serially while the others of the run idle.)
constexpr uint32_t kMaxFixRun = 48;
struct FixRuns {
    int64_t n;
    uint64_t* key;
    uint32_t* perm;
    uint32_t* overflow;
    int shift;
    PSIM_HD void operator()(int64_t s) const
    {
        uint64_t hi = key[s] >> shift;
        if (s > 0 && (key[s - 1] >> shift) == hi) return;
        int64_t e = s + 1;
        while (e < n && (key[e] >> shift) == hi) ++e;
        int64_t r = e - s;
        if (r == 1) return;
        if (r > (int64_t)kMaxFixRun) {
            *overflow = 1u;
            return;
        }
        for (int64_t i = s + 1; i < e; ++i) {
            const uint64_t ki = key[i];
            const uint32_t pi = perm[i];
            int64_t j = i;
            while (j > s && key[j - 1] > ki) {
                key[j] = key[j - 1];
                perm[j] = perm[j - 1];
                --j;
            }
            if (j != i) { key[j] = ki; perm[j] = pi; }
        }
    }
};

// For sorted particle s: L[s] = number of leading levels its key shares with
// key[s+1] (capped at 31; -1 for the last particle).  The reference's internal
// nodes (quadtree.py:54-59: a leaf that receives a second particle subdivides)
// are exactly the (depth, prefix) cells holding >= 2 particles; the cell of
// depth d whose first particle is s exists iff L[s] >= d, and s is its first
// particle iff L[s-1] < d.  So s "owns" the internal nodes of depths
// L[s-1]+1 .. L[s], a chain, and numbering nodes by (s, depth) is a pre-order.
struct TreeLinks {
    int64_t n;
    const uint64_t* key;
    int8_t* L;
    uint32_t* owned;
    PSIM_HD void operator()(int64_t s) const
    {
        int l = (s + 1 < n) ? shared_levels(key[s], key[s + 1]) : -1;
        int lp = (s > 0) ? shared_levels(key[s - 1], key[s]) : -1;
        L[s] = (int8_t)l;
        owned[s] = (uint32_t)(l > lp ? l - lp : 0);
    }
};

// Last sorted particle sharing its first d levels with particle s; `from` is an index
// already known to lie inside that range (s itself if nothing better is known).
PSIM_HD PSIM_FORCEINLINE uint32_t range_last(const uint64_t* key, uint32_t s, uint32_t n, int d, uint32_t from)
{
    if (d == 0) return n - 1;
    int sh = 64 - 2 * d;
    uint64_t pre = key[s] >> sh;
    uint32_t lo = from, hi;
    uint32_t step = 1;
    for (;;) {
        uint64_t t = (uint64_t)lo + step;
        if (t >= n) { hi = n; break; }
        if ((key[t] >> sh) != pre) { hi = (uint32_t)t; break; }
        lo = (uint32_t)t;
        step <<= 1;
    }
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if ((key[mid] >> sh) == pre) lo = mid; else hi = mid;
    }
    return lo;
}

// The pre-order element array holds, for sorted particle s, its owned internal
// nodes (shallowest first) followed by its own leaf.  Internal node ids are
// base[s] + k; the element of node id owned by s sits at index s + id.
// TreeEmitLeaves (one lane per particle) writes the leaf elements and, per owned node,
// its first particle and depth; TreeEmitNodes (one lane per internal node) then finds
// where each node's particle range ends and writes its element.
struct TreeEmitLeaves {
    int64_t n;
    const int8_t* L;
    const uint32_t* owned;
    const uint32_t* base;      // exclusive scan of owned
    const vec2* spos;
    const double* smass;
    NodeRec* rec;
    uint32_t* node_first;
    uint32_t* node_depth;      // u32 so it can be a radix-sort key
    uint32_t* node_id;
    PSIM_HD void operator()(int64_t s) const
    {
        uint32_t b = base[s], own = owned[s];
        int lp = (s > 0) ? (int)L[s - 1] : -1;
        for (uint32_t k = 0; k < own; ++k) {
            uint32_t id = b + k;
            node_first[id] = (uint32_t)s;
            node_depth[id] = (uint32_t)(lp + 1 + (int)k);
            node_id[id] = id;
        }
        uint32_t le = (uint32_t)s + b + own;
        NodeRec r;
        r.mass = smass[s];
        r.x = spos[s].x;
        r.y = spos[s].y;
        r.skip = le + 1;
        r.tag = kLeafTag;
        rec[le] = r;
    }
};

// The cell of depth d that starts at particle s holds s .. e where e is the first t >= s
// with L[t] < d (L[t] = levels shared by particles t and t + 1; L[n-1] = -1).  Most cells
// hold a handful of particles, so the next kScan values of L are fetched at once; a
// larger cell is finished by a galloping search on the keys.
constexpr uint32_t kDepthDone = 32;      // depth-sort key of a cell whose centre of mass is already final
struct TreeEmitNodes {
    static constexpr int kScan = 8;
    int64_t count;             // internal nodes
    int64_t n;
    const uint64_t* key;
    const int8_t* L;
    const uint32_t* base;
    const uint32_t* node_first;
    uint32_t* node_depth;
    const vec2* spos;
    const double* smass;
    uint32_t total_elems;      // n + number of internal nodes
    uint32_t root_tag;         // NodeRec::tag of depth 0
    NodeRec* rec;
    uint32_t* node_last;
    PSIM_HD void operator()(int64_t id) const
    {
        const uint32_t s = node_first[id];
        const int d = (int)node_depth[id];
        const uint32_t last = (uint32_t)n - 1;
        int m[kScan];
        for (int i = 0; i < kScan; ++i) {
            uint32_t t = s + 1 + (uint32_t)i;       // L[s] >= d: the owner shares d levels with s + 1
            m[i] = (int)L[t < last ? t : last];
        }
        uint32_t e = 0;
        bool found = false;
        for (int i = 0; i < kScan; ++i)
            if (!found && m[i] < d) { e = s + 1 + (uint32_t)i; found = true; }
        if (!found) e = range_last(key, s, (uint32_t)n, d, s + (uint32_t)kScan);
        uint32_t skip = (e + 1 < (uint32_t)n) ? (e + 1) + base[e + 1] : total_elems;
        NodeRec r;
        r.mass = 0.0; r.x = 0.0; r.y = 0.0;
        r.skip = skip;
        r.tag = node_tag(root_tag, d);
        // A cell whose particles all part ways at the very next level (L = d between every
        // neighbouring pair) has only leaves as children: its centre of mass is formed here, in
        // the order and arithmetic of TreeCom, and the cell is filed under kDepthDone so that no
        // per-depth pass visits it again.
        if (found && d < kMaxInternalDepth && e - s <= 3u) {
            bool all_leaves = (int)L[s] == d;
            for (uint32_t i = 0; i + 1 < e - s; ++i) all_leaves = all_leaves && (m[i] == d);
            if (all_leaves) {
                double total = 0.0, nx = 0.0, ny = 0.0;
                for (uint32_t t = s; t <= e; ++t) {
                    double cm = smass[t];
                    if (cm > 0) {
                        total += cm;
                        nx += spos[t].x * cm;
                        ny += spos[t].y * cm;
                    }
                }
                r.mass = total;
                if (total > 0) { r.x = nx / total; r.y = ny / total; }
                node_depth[id] = kDepthDone;
            }
        }
        rec[s + (uint32_t)id] = r;
        node_last[id] = e;
    }
};

// Reads the (up to four) children of internal node `id` in quadrant order.
struct NodeChildren {
    int64_t n;
    const uint64_t* key;
    const int8_t* L;
    const uint32_t* base;
    const vec2* spos;
    const double* smass;
    const uint32_t* node_first;
    const uint32_t* node_last;
    const NodeRec* rec;

    // deepest level only: children are the four depth-32 cells, each a bucket of
    // particles that coincide to all 32 levels (the reference never terminates on these)
    PSIM_HD void by_search(uint32_t id, uint32_t s, int depth, double* cm, double* cx, double* cy) const
    {
        uint32_t e = node_last[id];
        int sh = 62 - 2 * depth;
        // quadrant boundaries inside [s, e]
        uint32_t bnd[5];
        bnd[0] = s;
        bnd[4] = e + 1;
        for (int q = 1; q < 4; ++q) {
            uint32_t lo = bnd[q - 1], hi = e + 1;     // first index with digit >= q
            while (lo < hi) {
                uint32_t mid = lo + (hi - lo) / 2;
                if (((key[mid] >> sh) & 3u) >= (uint64_t)q) hi = mid; else lo = mid + 1;
            }
            bnd[q] = lo;
        }
        for (int q = 0; q < 4; ++q) {
            uint32_t c0 = bnd[q], cnt = bnd[q + 1] - bnd[q];
            cm[q] = 0.0; cx[q] = 0.0; cy[q] = 0.0;
            if (cnt == 1) {
                cm[q] = smass[c0]; cx[q] = spos[c0].x; cy[q] = spos[c0].y;
            } else if (cnt >= 2) {
                if (depth + 1 <= kMaxInternalDepth) {
                    int lp = (c0 > 0) ? (int)L[c0 - 1] : -1;
                    uint32_t cid = base[c0] + (uint32_t)(depth - lp);
                    const NodeRec& r = rec[c0 + cid];
                    cm[q] = r.mass; cx[q] = r.x; cy[q] = r.y;
                } else {
                    // bucket leaf: fold its particles in order
                    double tm = 0.0, nx = 0.0, ny = 0.0;
                    for (uint32_t k = c0; k < c0 + cnt; ++k) {
                        double m = smass[k];
                        if (m > 0) { tm += m; nx += spos[k].x * m; ny += spos[k].y * m; }
                    }
                    cm[q] = tm;
                    if (tm > 0) { cx[q] = nx / tm; cy[q] = ny / tm; }
                }
            }
        }
    }

    PSIM_HD void load(uint32_t id, int depth, double* cm, double* cx, double* cy) const
    {
        uint32_t s = node_first[id];
        for (int q = 0; q < 4; ++q) { cm[q] = 0.0; cx[q] = 0.0; cy[q] = 0.0; }
        if (depth < kMaxInternalDepth) {
            // the children are the elements of this node's subtree reached by following skip
            // from its first element: pre-order = quadrant order, empty quadrants absent
            uint32_t end = rec[s + id].skip;
            uint32_t c = s + id + 1;
            for (int q = 0; q < 4 && c < end; ++q) {
                const NodeRec& r = rec[c];
                cm[q] = r.mass; cx[q] = r.x; cy[q] = r.y;
                uint32_t nxt = r.skip;
                c = nxt > c ? nxt : c + 1;
            }
        } else {
            by_search(id, s, depth, cm, cx, cy);
        }
    }
};

// quadtree.py:85-114 (_calculate_mass_distribution_jit) for the internal nodes
// of one depth, children visited in quadrant order 0..3, strict FP64:
//   total += m_c; num += com_c * m_c (mass <= 0 skipped); com = num / total.
struct TreeCom {
    int64_t count;
    const uint32_t* ids;       // node ids of this depth
    int depth;
    NodeChildren ch;
    NodeRec* rec;
    // the same pass for another depth: ids_by_depth holds all node ids sorted by depth,
    // depth d occupying [lo, lo + cnt)
    PSIM_HD TreeCom at_level(const uint32_t* ids_by_depth, int d, int64_t lo, int64_t cnt) const
    {
        TreeCom t = *this;
        t.count = cnt; t.ids = ids_by_depth + lo; t.depth = d;
        return t;
    }
    PSIM_HD void operator()(int64_t t) const
    {
        uint32_t id = ids[t];
        double cm[4], cx[4], cy[4];
        ch.load(id, depth, cm, cx, cy);
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

// particle_system.py:43-63: the sibling-pair potential energy of one internal node
// (its children taken as point masses at their centres of mass).  Only needed when
// the energies are asked for, so it is not part of the step.
struct TreePotential {
    int64_t count;             // internal nodes
    NodeChildren ch;
    double* pe_node;
    double g_const, softening;
    uint32_t root_tag;
    PSIM_HD void operator()(int64_t t) const
    {
        uint32_t id = (uint32_t)t;
        int depth = tag_depth(root_tag, ch.rec[ch.node_first[id] + id].tag);
        double cm[4], cx[4], cy[4];
        ch.load(id, depth, cm, cx, cy);
        double pe = 0.0;
        for (int i = 0; i < 4; ++i) {
            if (!(cm[i] > 0)) continue;
            for (int j = i + 1; j < 4; ++j) {
                if (!(cm[j] > 0)) continue;
                double dx = cx[j] - cx[i], dy = cy[j] - cy[i];
                double d2 = dx * dx + dy * dy;
                if (d2 > 0) pe -= (g_const * cm[i] * cm[j]) / sqrt(d2 + softening);
            }
        }
        pe_node[id] = pe;
    }
};

// --------------------------------------------------------------- stage 4: gravity
// particle_system.py:69-121 (_traverse_tree_jit) as a stackless pre-order walk:
// an internal node is accepted iff d2 > 0 and w*w < theta*theta*d2, otherwise
// the walk steps into its first child (the next element); a leaf interacts iff
// d2 > 0 (which also excludes the particle itself, :105).  Accepted or leaf:
// continue at rec.skip.  One element per sorted particle; this is the plain
// one-lane-per-particle form (the wave-cooperative kernel lives in backend_hip.h).
PSIM_HD PSIM_FORCEINLINE double inv_cube_soft(double x)
{
    double r = 1.0 / sqrt(x);
    return r * r * r;
}

struct GravityWalkSimple {
    int64_t n;                  // number of targets: entries s0 .. s0 + n - 1 of `order`
    int64_t s0;
    int32_t sorted_out;         // write acc_out[entry number] instead of acc_out[particle index]
    const uint32_t* order;      // optional: sorted positions in the order the walks take them (null: 0, 1, 2, ...)
    const NodeRec* rec;
    uint32_t total_elems;
    const vec2* spos;
    const double* smass;
    const uint32_t* perm;
    const double* inv_mass;     // index space
    vec2* acc_out;              // index space
    const double* w2tab;        // [32] squared node width per depth
    uint32_t root_tag;          // NodeRec::tag of depth 0
    double theta2, g_const, softening;
    uint32_t* counters;         // optional [2]: visits, interactions
    PSIM_HD void operator()(int64_t i) const
    {
        int64_t s = order ? (int64_t)order[s0 + i] : s0 + i;
        double px = spos[s].x, py = spos[s].y;
        double ax = 0.0, ay = 0.0;
        uint32_t j = 0, visits = 0, inter = 0;
        while (j < total_elems) {
            NodeRec r = rec[j];
            double dx = r.x - px, dy = r.y - py;
            double d2 = dx * dx + dy * dy;
            double w2 = (r.tag == kLeafTag) ? -1.0 : w2tab[tag_depth(root_tag, r.tag)];
            ++visits;
            if (d2 > 0 && w2 < theta2 * d2) {
                double f = r.mass * inv_cube_soft(d2 + softening);
                ax += f * dx;
                ay += f * dy;
                ++inter;
                j = r.skip > j ? r.skip : j + 1;
            } else {
                j = j + 1;
            }
        }
        uint32_t p = perm[s];
        double gm = g_const * smass[s];
        double im = inv_mass[p];
        vec2 a;
        a.x = (gm * ax) * im;      // particle_system.py:25  a = F * (1/m)
        a.y = (gm * ay) * im;
        acc_out[sorted_out ? (uint32_t)(s0 + i) : p] = a;
        if (counters) {
            atomic_inc_u32(&counters[0], visits);
            atomic_inc_u32(&counters[1], inter);
        }
    }
};

// particle_system.py:540  v += 0.5 * (a_old + a_new)
struct Kick {
    int64_t n;
    vec2* vel;
    const vec2* acc_old;
    const vec2* acc_new;
    PSIM_HD void operator()(int64_t i) const
    {
        vec2 v = vel[i], a = acc_old[i], b = acc_new[i];
        v.x = v.x + 0.5 * (a.x + b.x);
        v.y = v.y + 0.5 * (a.y + b.y);
        vel[i] = v;
    }
};

// --------------------------------------------------------------- stage 6: collisions
// Candidate order of one sweep (particle_system.py:148-197): home cells in
// row-major order; inside a home cell first the in-cell pairs (a < b in grid
// order), then the 8 neighbour cells in (dy, dx) order, p1 over the home cell
// (outer), p2 over the neighbour (inner), visited iff index(p1) < index(p2).
// pair_key packs that order into one integer; a pair is always keyed from its
// lower-index particle, whose cell is the home cell.
// The grid may cover a band of whole cell rows only (one rank's slab of the domain plus its
// halo rows, when the collision stage is split over several GPUs): cells keep their global
// ids and rows, `offsets` is indexed by global cell id and holds the cells of rows
// [row_lo, row_hi) only, slots are numbered from 0 inside the band.  The whole grid is the
// band [0, gh).  [own_lo, own_hi) are the rows whose results this grid's owner keeps.
struct CellOffsets {
    const int32_t* p;          // [cells of the band + 1]
    uint32_t base;             // global id of the band's first cell
    PSIM_HD PSIM_FORCEINLINE int32_t operator[](int64_t cell) const { return p[cell - (int64_t)base]; }
};
struct GridView {
    const CellBody* body;      // slot -> {position, radius, particle index (grid_indices), cell}
    CellOffsets offsets;       // cell -> first slot
    int32_t gw, gh;
    int32_t slot_bits;         // bits of the in-cell position fields of a pair key (>= bits of max occupancy)
    int32_t row_lo, row_hi;    // rows the grid holds
    int32_t own_lo, own_hi;    // rows owned
    PSIM_HD PSIM_FORCEINLINE bool has_row(int ny) const { return ny >= row_lo && ny < row_hi; }
    PSIM_HD PSIM_FORCEINLINE bool owns_row(int ny) const { return ny >= own_lo && ny < own_hi; }
};

// key = cell | phase:4 | a:slot_bits | b:slot_bits, packed tightly so that the
// radix sorts of contact lists run as few digit passes as the grid allows
PSIM_HD PSIM_FORCEINLINE uint64_t make_pair_key(const GridView& g, uint32_t cell, uint32_t phase, uint32_t a, uint32_t b)
{
    int sb = g.slot_bits;
    return ((((((uint64_t)cell << 4) | (uint64_t)phase) << sb) | (uint64_t)a) << sb) | (uint64_t)b;
}

PSIM_HD PSIM_FORCEINLINE uint32_t neighbour_phase(int dy, int dx) { return 1u + (uint32_t)((dy + 1) * 3 + (dx + 1)); }

PSIM_HD PSIM_FORCEINLINE uint64_t pair_key(const GridView& g, uint32_t sx, uint32_t sy)
{
    uint32_t lo = sx, hi = sy;
    if (g.body[sy].idx < g.body[sx].idx) { lo = sy; hi = sx; }
    uint32_t cl = g.body[lo].cell, ch = g.body[hi].cell;
    uint32_t a = lo - (uint32_t)g.offsets[cl], b = hi - (uint32_t)g.offsets[ch];
    uint32_t phase = 0;
    if (cl != ch) {
        uint32_t xl = g.body[lo].cxy, xh = g.body[hi].cxy;
        int dx = (int)(xh & 0xffffu) - (int)(xl & 0xffffu);
        int dy = (int)(xh >> 16) - (int)(xl >> 16);
        phase = neighbour_phase(dy, dx);
    }
    return make_pair_key(g, cl, phase, a, b);
}

PSIM_HD PSIM_FORCEINLINE void decode_pair_key(const GridView& g, uint64_t key, uint32_t* si, uint32_t* sj)
{
    int sb = g.slot_bits;
    uint32_t mask = (1u << sb) - 1u;
    uint32_t b = (uint32_t)key & mask, a = (uint32_t)(key >> sb) & mask;
    uint32_t phase = (uint32_t)(key >> (2 * sb)) & 15u;
    uint32_t cell = (uint32_t)(key >> (2 * sb + 4));
    uint32_t nc = cell;
    if (phase) {
        int idx = (int)phase - 1;
        nc = (uint32_t)((int)cell + (idx / 3 - 1) * g.gw + (idx % 3 - 1));
    }
    *si = (uint32_t)g.offsets[cell] + a;
    *sj = (uint32_t)g.offsets[nc] + b;
}

// contact test of particle_system.py:210-217 (strict): 0 < d2 < (ri + rj)^2
PSIM_HD PSIM_FORCEINLINE bool in_contact(vec2 pi, vec2 pj, int32_t ri, int32_t rj)
{
    double dx = pj.x - pi.x, dy = pj.y - pi.y;
    double d2 = dx * dx + dy * dy;
    double md = (double)(ri + rj);       // (ri + rj)^2 is exact in FP64 for radii below 2^25
    return d2 < md * md && d2 > 0;
}

PSIM_HD PSIM_FORCEINLINE vec2 body_pos(const CellBody& b) { vec2 v; v.x = b.x; v.y = b.y; return v; }

// squared distance is exactly zero: no contact for the collision test (:217) but
// "touching" for the conduction test (:322)
PSIM_HD PSIM_FORCEINLINE bool coincident(vec2 pi, vec2 pj)
{
    double dx = pj.x - pi.x, dy = pj.y - pi.y;
    return dx * dx + dy * dy == 0;
}

// Enumerate, for slot k, every slot y of the 3x3 cell neighbourhood (own cell
// first, then neighbours in (dy, dx) order).  F(y, phase_from_k_to_y).
template <class F>
PSIM_HD PSIM_FORCEINLINE void for_each_neighbour_slot(const GridView& g, uint32_t k, F f)
{
    uint32_t c = g.body[k].cell, cxy = g.body[k].cxy;
    int cx = (int)(cxy & 0xffffu), cy = (int)(cxy >> 16);
    for (int32_t b = g.offsets[c]; b < g.offsets[c + 1]; ++b)
        if ((uint32_t)b != k) f((uint32_t)b);
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0) continue;
            int nx = cx + dx, ny = cy + dy;
            if (nx < 0 || nx >= g.gw || !g.has_row(ny)) continue;
            uint32_t nc = (uint32_t)(ny * g.gw + nx);
            for (int32_t b = g.offsets[nc]; b < g.offsets[nc + 1]; ++b) f((uint32_t)b);
        }
}

// The same enumeration restricted to one of the nine cells: which = 0 is the own
// cell, 1..9 (skipping 5) the neighbour in (dy, dx) order.
template <class F>
PSIM_HD PSIM_FORCEINLINE void for_each_slot_in_cell(const GridView& g, uint32_t k, int which, F f)
{
    uint32_t c = g.body[k].cell;
    if (which == 0) {
        for (int32_t b = g.offsets[c]; b < g.offsets[c + 1]; ++b)
            if ((uint32_t)b != k) f((uint32_t)b);
        return;
    }
    int idx = which - 1;
    if (idx == 4) return;
    uint32_t cxy = g.body[k].cxy;
    int cx = (int)(cxy & 0xffffu), cy = (int)(cxy >> 16);
    int nx = cx + (idx % 3 - 1), ny = cy + (idx / 3 - 1);
    if (nx < 0 || nx >= g.gw || !g.has_row(ny)) return;
    uint32_t nc = (uint32_t)(ny * g.gw + nx);
    for (int32_t b = g.offsets[nc]; b < g.offsets[nc + 1]; ++b) f((uint32_t)b);
}

// The slots for_each_slot_in_cell would visit, as a range [lo, hi) (the caller skips k itself
// when which == 0).  False: nothing to visit.
PSIM_HD PSIM_FORCEINLINE bool cell_slot_range(const GridView& g, const CellBody& me, int which, int32_t* lo, int32_t* hi)
{
    uint32_t c = me.cell;
    if (which != 0) {
        int idx = which - 1;
        if (idx == 4) return false;
        int cx = (int)(me.cxy & 0xffffu), cy = (int)(me.cxy >> 16);
        int nx = cx + (idx % 3 - 1), ny = cy + (idx / 3 - 1);
        if (nx < 0 || nx >= g.gw || !g.has_row(ny)) return false;
        c = (uint32_t)(ny * g.gw + nx);
    }
    *lo = g.offsets[c];
    *hi = g.offsets[c + 1];
    return *hi > *lo;
}

// F(record of slot y, aux[y] (0 without aux), y) for y in [lo, hi) except `skip`, in ascending order, with
// the records and aux words of four slots fetched before the first of them is looked at.
template <class F>
PSIM_HD PSIM_FORCEINLINE void sweep_slots4(const CellBody* PSIM_RESTRICT body, const int32_t* PSIM_RESTRICT aux,
                                           int32_t lo, int32_t hi, uint32_t skip, F f)
{
    for (int32_t b = lo; b < hi; b += 4) {
        const int32_t last = hi - 1;
        const int32_t b1 = b + 1 < hi ? b + 1 : last, b2 = b + 2 < hi ? b + 2 : last, b3 = b + 3 < hi ? b + 3 : last;
        const CellBody o0 = body[b], o1 = body[b1], o2 = body[b2], o3 = body[b3];
        int32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (aux) { a0 = aux[b]; a1 = aux[b1]; a2 = aux[b2]; a3 = aux[b3]; }
        if ((uint32_t)b != skip) f(o0, a0, (uint32_t)b);
        if (b + 1 < hi && (uint32_t)(b + 1) != skip) f(o1, a1, (uint32_t)(b + 1));
        if (b + 2 < hi && (uint32_t)(b + 2) != skip) f(o2, a2, (uint32_t)(b + 2));
        if (b + 3 < hi && (uint32_t)(b + 3) != skip) f(o3, a3, (uint32_t)(b + 3));
    }
}

// Sweep 0: test every candidate pair against the sweep's initial positions and
// collect the keys of those in contact (unordered between lanes; sorted
// afterwards).  Two passes around an exclusive scan instead of an atomic
// cursor: pass 0 stores how many contacts each slot found and the partner slots of
// the first kDetectKept; pass 1 writes those pairs' keys from the slot's scanned
// offset on, and only the slots that found more redo their tests.
constexpr int kDetectKept = 4;
struct ContactDetect {
    int64_t n;                 // slots [k0, k0 + n) are swept
    uint32_t k0;
    GridView g;
    uint32_t* counts;          // pass 0: written; pass 1: read
    const uint32_t* offsets;   // pass 1: exclusive scan of counts
    uint32_t* first_pair;      // pass 0: the partner slots of a slot's first kDetectKept contacts, from
                               // [kDetectKept k] on (the keys' field widths are only known after
                               // pass 0); pass 1: read
    uint64_t* out_keys;        // pass 1
    int32_t write_pass;
    uint32_t* cand_counter;    // optional (pass 0): number of candidate pairs
    uint32_t* coincident_counter;   // pass 0: pairs at squared distance exactly 0
    // prune: the positions are the ones the grid was built from, so a forward cell is only
    // fetched when the slot lies within contact distance (its radius + the largest radius
    // r_max) of the border it shares with that cell; the candidate count is the reference's
    // all the same (it is a sum of cell populations)
    int32_t prune, r_max;
    double cell_size;
    // margin > 0: pairs closer than contact distance + margin are collected as well.  They do
    // not touch yet; as members of the solver's first hypothesis they cost a test when their
    // turn comes and drop out if they still do not touch, and those that a push-apart before
    // them brings into contact are there from the start instead of one iteration later.
    double margin;
    // Every unordered candidate pair is examined once: in-cell pairs from the earlier
    // slot, cross-cell pairs from the cell that has the other one among its four
    // "forward" neighbours (E, SW, S, SE).  A contact is keyed from its lower-index
    // particle, whichever side found it.  Cells are row-major, so the candidates of a
    // slot are two runs of consecutive slots: the rest of its own cell followed by the E
    // cell, and the SW, S, SE cells; the records of a run are fetched four at a time.
    template <bool WRITE>
    PSIM_HD PSIM_FORCEINLINE void run(uint32_t k) const
    {
        const CellBody* PSIM_RESTRICT body = g.body;
        const CellBody me = body[k];
        const uint32_t c = me.cell;
        const double px = me.x, py = me.y;
        const int32_t mr = me.r;
        const int cx = (int)(me.cxy & 0xffffu), cy = (int)(me.cxy >> 16);
        const bool east = cx + 1 < g.gw, west = cx > 0, south = cy + 1 < g.row_hi;
        // Every position the grid puts into column cx + 1 or beyond has x >= (cx + 1) cell_size,
        // into row cy + 1 or beyond y >= (cy + 1) cell_size, into column cx - 1 or before
        // x < cx cell_size (particle_system.py:744-749; clipping only moves positions outwards).
        // The comparisons are written so that a NaN coordinate keeps the cell.
        bool need_e = east, need_s = south, need_sw = south && west, need_se = south && east;
        if (prune) {
            const double reach = (double)(mr + r_max) + margin + 1e-6;
            const double gap_e = (double)(cx + 1) * cell_size - px, gap_s = (double)(cy + 1) * cell_size - py;
            const double gap_w = px - (double)cx * cell_size;
            need_e = need_e && !(gap_e > reach);
            need_s = need_s && !(gap_s > reach);
            need_sw = need_sw && need_s && !(gap_w > reach);
            need_se = need_se && need_s && !(gap_e > reach);
        }
        const int32_t a_hi = g.offsets[c + (need_e ? 2u : 1u)];
        int32_t b_lo = 0, b_hi = 0;
        if (need_s) {
            const uint32_t s = c + (uint32_t)g.gw;
            b_lo = g.offsets[s - (need_sw ? 1u : 0u)];
            b_hi = g.offsets[s + (need_se ? 2u : 1u)];
        }
        uint32_t found = 0;
        const uint32_t at = WRITE ? offsets[k] : 0u;
        auto visit = [&](const CellBody& o, int32_t b) {
            // particle_system.py:210-217; the squared distance does not depend on which
            // of the two is "i" (the differences only change sign)
            double dx = o.x - px, dy = o.y - py;
            double d2 = dx * dx + dy * dy;
            double md = (double)(mr + o.r) + margin;
            // (with a margin a coinciding pair is collected too: it does not touch as it is,
            // particle_system.py:217, but a push on one of the two would make it)
            if (d2 < md * md && (d2 > 0 || margin > 0)) {
                if (WRITE) out_keys[at + found] = pair_key(g, k, (uint32_t)b);
                else if (found < (uint32_t)kDetectKept) first_pair[(size_t)kDetectKept * k + found] = (uint32_t)b;
                ++found;
            }
            if (!WRITE && d2 == 0) aggregated_inc(coincident_counter);
        };
        auto sweep = [&](int32_t lo, int32_t hi) {
            for (int32_t b = lo; b < hi; b += 4) {
                const int32_t last = hi - 1;
                const CellBody o0 = body[b];
                const CellBody o1 = body[b + 1 < hi ? b + 1 : last];
                const CellBody o2 = body[b + 2 < hi ? b + 2 : last];
                const CellBody o3 = body[b + 3 < hi ? b + 3 : last];
                visit(o0, b);
                if (b + 1 < hi) visit(o1, b + 1);
                if (b + 2 < hi) visit(o2, b + 2);
                if (b + 3 < hi) visit(o3, b + 3);
            }
        };
        sweep((int32_t)k + 1, a_hi);
        sweep(b_lo, b_hi);
        if (!WRITE) {
            counts[k] = found;
            if (cand_counter && g.owns_row(cy)) {
                // particle_system.py:156-197 visits every slot after k in its cell and every slot of the forward cells
                int32_t nc = g.offsets[c + (east ? 2u : 1u)] - ((int32_t)k + 1);
                if (south) {
                    const uint32_t s = c + (uint32_t)g.gw;
                    nc += g.offsets[s + (east ? 2u : 1u)] - g.offsets[s - (west ? 1u : 0u)];
                }
                if (nc > 0) atomic_inc_u32(cand_counter, (uint32_t)nc);
            }
        }
    }
    PSIM_HD void operator()(int64_t kk) const
    {
        const uint32_t k = k0 + (uint32_t)kk;
        if (write_pass) {
            uint32_t c = counts[k];
            if (c > (uint32_t)kDetectKept) {
                run<true>(k);
            } else {
                for (uint32_t i = 0; i < c; ++i)
                    out_keys[offsets[k] + i] = pair_key(g, k, first_pair[(size_t)kDetectKept * k + i]);
            }
        } else {
            run<false>(k);
        }
    }
};

// hypothesised contact h -> its two slots, plus two adjacency entries
// (slot << 32 | h) that are then sorted to give each slot its contacts in order.
struct ContactDecode {
    int64_t n;
    GridView g;
    const uint64_t* hkeys;
    uint32_t* hsi;
    uint32_t* hsj;
    uint64_t* adj;
    uint32_t* done;
    PSIM_HD void operator()(int64_t h) const
    {
        uint32_t si, sj;
        decode_pair_key(g, hkeys[h], &si, &sj);
        hsi[h] = si;
        hsj[h] = sj;
        adj[2 * h] = ((uint64_t)si << 32) | (uint64_t)h;
        adj[2 * h + 1] = ((uint64_t)sj << 32) | (uint64_t)h;
        done[h] = 0;
    }
};

// over sorted adjacency entries: segment starts per slot and, for each contact
// and side, the previous contact of the same slot (-1: none).
struct AdjLinks {
    int64_t n;
    const uint64_t* adj;
    const uint32_t* hsi;
    int32_t* cseg;       // slot -> first adjacency entry (-1 = clean)
    int32_t* prev;       // [2h + side]
    uint32_t* zero_word; // cleared here: SegDisplacement raises it to the largest displacement bound
    uint32_t* is_first;  // [e] 1 for the first entry of a slot's segment (compacted into a list afterwards)
    PSIM_HD void operator()(int64_t e) const
    {
        if (e == 0) *zero_word = 0u;
        uint32_t slot = (uint32_t)(adj[e] >> 32), h = (uint32_t)adj[e];
        bool first = (e == 0) || ((uint32_t)(adj[e - 1] >> 32) != slot);
        is_first[e] = first ? 1u : 0u;
        if (first) cseg[slot] = (int32_t)e;
        int side = (hsi[h] == slot) ? 0 : 1;
        prev[2 * h + side] = first ? -1 : (int32_t)(uint32_t)adj[e - 1];
    }
};

// list of the first entries of the segments (= one entry per touched slot), from the scan of is_first
struct CompactFirst {
    int64_t n;
    const uint32_t* is_first;
    const uint32_t* pos;
    uint32_t* list;
    PSIM_HD void operator()(int64_t e) const { if (is_first[e]) list[pos[e]] = (uint32_t)e; }
};

// particle_system.py:200-268 (_resolve_collision_jit), strict FP64, in two halves.
// Whether two particles touch and where the push-apart leaves them depends on
// positions, masses and radii only (:207-237); the impulse, the energy book-keeping
// and the heating (:223-225, :239-266) read that half's normal and PE change and the
// velocities.  The solver therefore settles the contact list and the positions first
// (while the tree walk is still running, since none of it needs the new accelerations)
// and runs the velocity half once over the final list.
//
// Position half on copies of the two positions (i = lower particle index).  geo =
// {nx, ny, pe_change}; a pair that does not touch leaves geo = 0 (a unit normal never is).
PSIM_HD PSIM_FORCEINLINE bool resolve_pair_positions(vec2& Pi, vec2& Pj, double mi, double mj, int32_t ri, int32_t rj,
                                                     double G, double soft, double pos_correct, double* geo)
{
    int64_t min_dist = (int64_t)ri + (int64_t)rj;
    double dvx = Pj.x - Pi.x, dvy = Pj.y - Pi.y;
    double d2 = dvx * dvx + dvy * dvy;
    geo[0] = 0.0; geo[1] = 0.0; geo[2] = 0.0;
    if (!(d2 < (double)(min_dist * min_dist) && d2 > 0)) return false;
    double dist = sqrt(d2);
    double M = mi + mj;
    double pe_before = (-G * mi * mj) / sqrt(d2 + soft);
    double overlap = ((double)min_dist - dist) * pos_correct;
    double nx = dvx / dist, ny = dvy / dist;
    double ci = (mj / M) * overlap, cj = (mi / M) * overlap;
    Pi.x -= ci * nx; Pi.y -= ci * ny;
    Pj.x += cj * nx; Pj.y += cj * ny;
    double fx = Pj.x - Pi.x, fy = Pj.y - Pi.y;
    double fd2 = fx * fx + fy * fy;
    double pe_after = (-G * mi * mj) / sqrt(fd2 + soft);
    geo[0] = nx; geo[1] = ny; geo[2] = pe_after - pe_before;
    return true;
}

// Velocity half.  vals = {pe_change, ke_lost, heat}.
PSIM_HD PSIM_FORCEINLINE void resolve_pair_velocities(VelState& I, VelState& J, double mi, double mj, const double* geo,
                                                      double restitution, double* vals)
{
    double nx = geo[0], ny = geo[1], pe_change = geo[2];
    vals[0] = 0.0; vals[1] = 0.0; vals[2] = 0.0;
    if (nx == 0.0 && ny == 0.0) return;
    double M = mi + mj;
    double vib0 = I.vel.x, vib1 = I.vel.y, vjb0 = J.vel.x, vjb1 = J.vel.y;
    double ke_before = 0.5 * mi * (vib0 * vib0 + vib1 * vib1) + 0.5 * mj * (vjb0 * vjb0 + vjb1 * vjb1);
    double vrx = vjb0 - vib0, vry = vjb1 - vib1;
    double vrn = vrx * nx + vry * ny;
    if (vrn < 0) {
        double imp = (-(1 + restitution) * mi * mj * vrn) / M;
        double ii = imp / mi, ij = imp / mj;
        I.vel.x -= ii * nx; I.vel.y -= ii * ny;
        J.vel.x += ij * nx; J.vel.y += ij * ny;
    }
    double ke_after = 0.5 * mi * (I.vel.x * I.vel.x + I.vel.y * I.vel.y) +
                      0.5 * mj * (J.vel.x * J.vel.x + J.vel.y * J.vel.y);
    double ke_lost = ke_before - ke_after;
    double heat = ke_lost - pe_change;
    if (heat > 0) {
        I.temp += (heat * 0.5) / mi;
        J.temp += (heat * 0.5) / mj;
    }
    vals[0] = pe_change; vals[1] = ke_lost; vals[2] = heat;
}

// One round of executing the hypothesised contacts' position halves in dependency
// order: contact h runs once the previous contact of each of its two slots ran in an
// EARLIER round (done[] holds the round number, 0 = not yet).
struct ContactSimulate {
    int64_t n;
    const uint32_t* hsi;
    const uint32_t* hsj;
    const int32_t* prev;
    uint32_t* done;
    uint32_t round;
    vec2* out;                // [2h + side] position after contact h
    double* geo;              // [3h] normal and PE change
    const CellBody* body;     // initial positions
    const double* mass;
    double G, soft, pos_correct;
    PSIM_HD vec2 fetch(uint32_t slot, int32_t pr) const
    {
        if (pr < 0) return body_pos(body[slot]);
        int side = (hsi[pr] == slot) ? 0 : 1;
        return out[2 * (uint32_t)pr + side];
    }
    PSIM_HD void operator()(int64_t h) const
    {
        if (done[h]) return;
        int32_t pi = prev[2 * h], pj = prev[2 * h + 1];
        if (pi >= 0) { uint32_t d = done[pi]; if (d == 0 || d == round) return; }
        if (pj >= 0) { uint32_t d = done[pj]; if (d == 0 || d == round) return; }
        run(h, pi, pj);
        done[h] = round;
    }
    // contact h itself, its predecessors' outputs being in place (done[h] is the caller's to set)
    PSIM_HD void run(int64_t h, int32_t pi, int32_t pj) const
    {
        uint32_t si = hsi[h], sj = hsj[h];
        vec2 I = fetch(si, pi), J = fetch(sj, pj);
        apply(h, prepare(h, si, sj), I, J);
    }
    // the same in pieces, for a caller that holds the predecessors' outputs itself: what the
    // contact needs besides the two states, a slot's state before any contact, the contact
    typedef vec2 State;
    struct Pre { double mi, mj; int32_t ri, rj; };
    PSIM_HD Pre prepare(int64_t, uint32_t si, uint32_t sj) const
    {
        Pre p;
        p.mi = mass[body[si].idx]; p.mj = mass[body[sj].idx]; p.ri = body[si].r; p.rj = body[sj].r;
        return p;
    }
    PSIM_HD State initial(uint32_t slot) const { return body_pos(body[slot]); }
    PSIM_HD void apply(int64_t h, const Pre& p, State& I, State& J) const
    {
        double v[3];
        resolve_pair_positions(I, J, p.mi, p.mj, p.ri, p.rj, G, soft, pos_correct, v);
        out[2 * h] = I;
        out[2 * h + 1] = J;
        geo[3 * h] = v[0]; geo[3 * h + 1] = v[1]; geo[3 * h + 2] = v[2];
    }
};

// Velocity halves of the settled contact list, in the same dependency rounds: contact
// h runs in the round its position half ran in (done[h]), so the previous contact of
// each of its slots has run before it.
struct ContactVelocity {
    int64_t n;
    const uint32_t* hsi;
    const uint32_t* hsj;
    const int32_t* prev;
    const uint32_t* done;
    uint32_t round;
    VelState* out;            // [2h + side]
    const double* geo;        // [3h]
    double* vals;             // [3h] pe_change, ke_lost, heat
    const CellBody* body;
    const vec2* vel;          // initial state
    const double* temp;
    const double* mass;
    double restitution;
    const VelState* slot_state;   // optional: the slots' states before this sweep, in slot order (instead of vel / temp)
    PSIM_HD VelState fetch(uint32_t slot, int32_t pr) const
    {
        if (pr < 0) {
            if (slot_state) return slot_state[slot];
            VelState s;
            uint32_t p = body[slot].idx;
            s.vel = vel[p];
            s.temp = temp[p];
            s.unused = 0.0;
            return s;
        }
        int side = (hsi[pr] == slot) ? 0 : 1;
        return out[2 * (uint32_t)pr + side];
    }
    PSIM_HD void operator()(int64_t h) const
    {
        if (done[h] != round) return;
        run(h, prev[2 * h], prev[2 * h + 1]);
    }
    PSIM_HD void run(int64_t h, int32_t pi, int32_t pj) const
    {
        uint32_t si = hsi[h], sj = hsj[h];
        VelState I = fetch(si, pi), J = fetch(sj, pj);
        apply(h, prepare(h, si, sj), I, J);
    }
    // (in pieces, as in ContactSimulate)
    typedef VelState State;
    struct Pre { double mi, mj, g[3]; };
    PSIM_HD Pre prepare(int64_t h, uint32_t si, uint32_t sj) const
    {
        Pre p;
        p.g[0] = geo[3 * h]; p.g[1] = geo[3 * h + 1]; p.g[2] = geo[3 * h + 2];
        // (a pair that did not touch passes both states on as they are: its masses are not needed)
        const bool touched = p.g[0] != 0.0 || p.g[1] != 0.0;
        p.mi = touched ? mass[body[si].idx] : 1.0;
        p.mj = touched ? mass[body[sj].idx] : 1.0;
        return p;
    }
    PSIM_HD State initial(uint32_t slot) const { return fetch(slot, -1); }
    PSIM_HD void apply(int64_t h, const Pre& p, State& I, State& J) const
    {
        double v[3];
        resolve_pair_velocities(I, J, p.mi, p.mj, p.g, restitution, v);
        out[2 * h] = I;
        out[2 * h + 1] = J;
        vals[3 * h] = v[0]; vals[3 * h + 1] = v[1]; vals[3 * h + 2] = v[2];
    }
};

// contacts still waiting on a predecessor, and the highest round any contact ran in
struct CountPending {
    int64_t n;
    const uint32_t* done;
    uint32_t* pending;
    uint32_t* last_round;
    PSIM_HD void operator()(int64_t h) const
    {
        uint32_t d = done[h];
        if (d == 0) aggregated_inc(pending);
        else raise_to_u32(last_round, d);
    }
};

// position of a slot as seen by the candidate pair with ordering key t: the
// output of its last hypothesised contact ordered before t, else the initial one
PSIM_HD PSIM_FORCEINLINE vec2 versioned_pos(uint32_t slot, uint64_t t, const int32_t* cseg, const uint64_t* adj,
                                            int64_t n_adj, const uint64_t* hkeys, const uint32_t* hsi,
                                            const vec2* out, const CellBody* body)
{
    int32_t e0 = cseg[slot];
    int64_t last = -1;
    if (e0 >= 0) {
        for (int64_t e = e0; e < n_adj && (uint32_t)(adj[e] >> 32) == slot; ++e) {
            uint32_t h = (uint32_t)adj[e];
            if (hkeys[h] < t) last = h; else break;
        }
    }
    if (last < 0) return body_pos(body[slot]);
    int side = (hsi[last] == slot) ? 0 : 1;
    return out[2 * last + side];
}

// positive float <-> its bit pattern (positive floats order like their bit patterns, and a
// NaN pattern is above them all)
PSIM_HD PSIM_FORCEINLINE uint32_t float_bits_at_least(double v)
{
    float f = (float)v;
    if ((double)f < v) f = f * 1.0000002f;      // next float or the one after
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    return b;
}
PSIM_HD PSIM_FORCEINLINE double double_of_float_bits(uint32_t b)
{
    float f;
    __builtin_memcpy(&f, &b, 4);
    return (double)f;
}

// Is every position that the grid assigns to neighbour cell `which` (1..9, not 5) of slot
// `me` further than `reach` from me's position?  A cell holds the positions whose
// coordinate / cell_size truncates to its column and row (particle_system.py:744-749); the
// cells of the first and last column / row also hold everything clipped into them, so they
// are unbounded outwards.  Conservative by a millionth of a pixel against rounding.
PSIM_HD PSIM_FORCEINLINE bool cell_beyond_reach(const GridView& g, double cell_size, const CellBody& me, int which,
                                                double reach)
{
    int idx = which - 1;
    int cx = (int)(me.cxy & 0xffffu), cy = (int)(me.cxy >> 16);
    int nx = cx + (idx % 3 - 1), ny = cy + (idx / 3 - 1);
    if (nx < 0 || nx >= g.gw || !g.has_row(ny)) return true;       // no such cell (in this grid)
    double gx = 0.0, gy = 0.0;
    if (nx > 0) { double lo = (double)nx * cell_size - me.x; if (lo > gx) gx = lo; }
    if (nx < g.gw - 1) { double hi = me.x - (double)(nx + 1) * cell_size; if (hi > gx) gx = hi; }
    if (ny > 0) { double lo = (double)ny * cell_size - me.y; if (lo > gy) gy = lo; }
    if (ny < g.gh - 1) { double hi = me.y - (double)(ny + 1) * cell_size; if (hi > gy) gy = hi; }
    double r = reach + 1e-6;
    return gx * gx + gy * gy > r * r;     // false for NaN: nothing is skipped
}

// For each touched slot: how far any of its versioned positions lies from its
// initial one (inflated so that rounding can never make the bound too small).
// Stored at the slot's first adjacency entry.  Lets the re-test discard, exactly,
// neighbours that no version could reach.
struct SegDisplacement {
    int64_t n_adj;
    const uint64_t* adj;
    const int32_t* cseg;
    const uint32_t* hsi;
    const vec2* out;
    const CellBody* body;
    double* dseg;
    uint32_t* dmax_bits;       // raised to the largest bound of all slots (float bit pattern)
    uint32_t* zero_word;       // [0] and [5] are cleared here for the counting pass that follows
    // The same pass tells the re-test which slots it has to look at again: a slot's versioned
    // positions are a function of the sequence (ordering key, position after) of its
    // hypothesised contacts.  sig[slot] keeps a 64-bit digest of that sequence from the previous
    // iteration of the sweep (0 = no contact); changed[slot] is set where the digest differs.
    const uint64_t* hkeys;
    const uint64_t* sig;
    uint64_t* dsig;            // [first adjacency entry of the slot] this iteration's digest (EndIteration moves it to sig)
    uint8_t* changed;
    // 1: every touched slot counts as changed in every iteration, i.e. the re-test trusts no
    // digest and looks at every candidate pair around every touched slot again (the check
    // that the digests' shortcut never alters a contact list)
    int32_t all_changed;
    // The digest covers the slot's hypothesised contacts that did touch (geo != 0): one that
    // did not left the position as it was, so the slot's position at any place in the order
    // is the same with and without it.
    const double* geo;
    // far_from >= 0 (first iteration of a sweep whose hypothesis holds every pair within contact
    // distance + margin of the initial positions): a pair outside the hypothesis can only touch
    // if its two slots moved by more than the margin together, so only slots that moved by
    // more than far_from = margin / 2 (less a hair) have pairs to look at again; the pairs
    // inside the hypothesis were all tested when their turn came.
    double far_from;
    static PSIM_HD PSIM_FORCEINLINE uint64_t mix(uint64_t h)
    {
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
        return h;
    }
    // digest of a slot none of whose hypothesised contacts touched (also what sig = 0, "no
    // hypothesised contact at all", stands for: the slot is at its initial position throughout)
    static constexpr uint64_t kAtRest = 0x9e3779b97f4a7c15ull;
    PSIM_HD void operator()(int64_t e) const
    {
        if (e == 0) { zero_word[0] = 0u; zero_word[5] = 0u; }
        uint32_t x = (uint32_t)(adj[e] >> 32);
        if (cseg[x] != (int32_t)e) return;
        double x0 = body[x].x, y0 = body[x].y, m = 0.0;
        uint64_t digest = kAtRest;
        for (int64_t k = e; k < n_adj && (uint32_t)(adj[k] >> 32) == x; ++k) {
            uint32_t h = (uint32_t)adj[k];
            int side = (hsi[h] == x) ? 0 : 1;
            vec2 p = out[2 * h + side];
            double dx = p.x - x0, dy = p.y - y0;
            double d2 = dx * dx + dy * dy;
            m = d2 > m ? d2 : m;
            if (geo[3 * (size_t)h] == 0.0 && geo[3 * (size_t)h + 1] == 0.0) continue;      // did not touch
            uint64_t bx, by;
            __builtin_memcpy(&bx, &p.x, 8);
            __builtin_memcpy(&by, &p.y, 8);
            digest = mix(digest ^ hkeys[h]);
            digest = mix(digest ^ bx);
            digest = mix(digest ^ by);
        }
        if (digest == 0) digest = 1;
        double bound = sqrt(m) * (1.0 + 1e-6) + 1e-9;
        dseg[e] = bound;
        raise_to_u32(dmax_bits, float_bits_at_least(bound));
        if (far_from >= 0.0 && !all_changed) changed[x] = bound > far_from ? 1 : 0;
        else changed[x] = (all_changed || (sig[x] ? sig[x] : kAtRest) != digest) ? 1 : 0;
        dsig[e] = digest;
    }
};

// The slots the previous iteration's hypothesis touched and this one's does not: they are
// back at their initial positions, which is a change too.
struct MarkUntouched {
    int64_t cap;
    const uint32_t* prev_slots;
    const uint32_t* n_prev;
    const int32_t* cseg;
    uint64_t* sig;
    uint8_t* changed;
    PSIM_HD void operator()(int64_t i) const
    {
        if (i >= (int64_t)*n_prev) return;
        uint32_t y = prev_slots[i];
        // (a slot whose hypothesised contacts all failed to touch never left its initial position)
        if (cseg[y] < 0) { if (sig[y] != SegDisplacement::kAtRest) changed[y] = 1; sig[y] = 0; }
    }
};

// For large hypotheses the slots the re-test has to look at are compacted into a list first
// (flag, scan, write), so that its launches and scans cover the changed slots only and not
// every touched one: element i < n_adj stands for the i-th touched slot, the rest for the
// previous iteration's touched slots.  Entry = slot << 32 | first adjacency entry (kNoEntry
// for a slot without contacts).
constexpr uint32_t kNoEntry = 0xffffffffu;
struct WorkFlags {
    int64_t n_adj, cap_prev;
    const uint64_t* adj;
    const uint32_t* touched;
    const uint32_t* n_touched;
    const uint32_t* prev_slots;
    const uint32_t* n_prev;
    const int32_t* cseg;
    const uint8_t* changed;
    uint32_t* flags;
    PSIM_HD void operator()(int64_t i) const
    {
        uint32_t f = 0;
        if (i < n_adj) {
            if (i < (int64_t)*n_touched) f = changed[(uint32_t)(adj[touched[i]] >> 32)] ? 1u : 0u;
        } else if (i - n_adj < (int64_t)*n_prev) {
            f = (cseg[prev_slots[i - n_adj]] < 0 && changed[prev_slots[i - n_adj]]) ? 1u : 0u;
        }
        flags[i] = f;
    }
};
struct WorkList {
    int64_t n_adj, cap_prev;
    const uint64_t* adj;
    const uint32_t* touched;
    const uint32_t* prev_slots;
    const uint32_t* flags;
    const uint32_t* pos;
    uint64_t* list;
    PSIM_HD void operator()(int64_t i) const
    {
        if (!flags[i]) return;
        if (i < n_adj) {
            uint32_t e = touched[i];
            list[pos[i]] = (adj[e] & 0xffffffff00000000ull) | (uint64_t)e;
        } else {
            list[pos[i]] = ((uint64_t)prev_slots[i - n_adj] << 32) | (uint64_t)kNoEntry;
        }
    }
};

// this iteration's touched slots, kept for the next iteration's MarkUntouched
struct SaveTouched {
    int64_t cap;
    const uint64_t* adj;
    const uint32_t* touched;
    const uint32_t* n_touched;
    uint32_t* slots;
    uint32_t* n_out;
    PSIM_HD void operator()(int64_t i) const
    {
        uint32_t k = *n_touched;
        if (i == 0) *n_out = k;
        if (i < (int64_t)k) slots[i] = (uint32_t)(adj[touched[i]] >> 32);
    }
};

// end of an iteration that is followed by another: the touched slots' digests become the
// ones to compare with, their flags and segment starts are cleared
struct EndIteration {
    int64_t n_adj;
    const uint64_t* adj;
    const uint64_t* dsig;
    int32_t* cseg;
    uint64_t* sig;
    uint8_t* changed;
    PSIM_HD void operator()(int64_t e) const
    {
        uint32_t x = (uint32_t)(adj[e] >> 32);
        if (e > 0 && (uint32_t)(adj[e - 1] >> 32) == x) return;      // the slot's first entry does it
        sig[x] = dsig[e];
        changed[x] = 0;
        cseg[x] = -1;
    }
};

// end of the sweep: the flags (and, with clear_sig, the digests) of the given slots
struct ClearChangedAdj {
    int64_t n_adj;
    const uint64_t* adj;
    uint8_t* changed;
    uint64_t* sig;
    int32_t clear_sig;
    PSIM_HD void operator()(int64_t e) const
    {
        uint32_t x = (uint32_t)(adj[e] >> 32);
        changed[x] = 0;
        if (clear_sig) sig[x] = 0;
    }
};
struct ClearChangedList {
    int64_t cap;
    const uint32_t* slots;
    const uint32_t* count;
    uint8_t* changed;
    uint64_t* sig;
    int32_t clear_sig;
    PSIM_HD void operator()(int64_t i) const
    {
        if (i >= (int64_t)*count) return;
        uint32_t x = slots[i];
        changed[x] = 0;
        if (clear_sig) sig[x] = 0;
    }
};

// Re-test, against the versioned positions, every candidate pair that involves a slot whose
// versioned positions changed since the previous iteration (SegDisplacement, MarkUntouched;
// in the first iteration: every touched slot).  A pair of two unchanged slots keeps the
// status the previous iteration gave it, which is its membership in the current hypothesis
// (ContactCarry).  Elements [0, 10 n_adj): cell selector * n_adj + index into the list of
// touched slots; elements [10 n_adj, 10 (n_adj + cap_prev)): the same over the previous
// iteration's touched slots, of which only those no longer touched have work to do here.
// (The nine cells of one slot are re-tested by different lanes; elements past the end of
// a list have nothing to do.)  A pair of two changed slots is emitted once, by the lower slot.
// Two passes around a scan, like ContactDetect: counts[t] and the first kRetestKept hits'
// keys, then keys at offsets[t] (elements with more hits than that redo their tests).
constexpr int kRetestKept = 4;
struct ContactRetest {
    int64_t n_adj;
    GridView g;
    const uint64_t* adj;
    const int32_t* cseg;
    const uint64_t* hkeys;
    const uint32_t* hsi;
    const vec2* out;
    const double* dseg;
    // prune: the sweep's initial positions are the ones the grid was built from, so a whole
    // neighbour cell can be skipped when its region lies beyond contact distance plus the
    // displacement bounds (largest radius r_max, largest bound of any slot *dmax_bits)
    int32_t prune, r_max;
    double cell_size;
    const uint32_t* dmax_bits;
    const uint32_t* touched;   // the first adjacency entry of every touched slot
    const uint32_t* n_touched;
    const uint8_t* changed;    // slot -> versioned positions differ from the previous iteration's
    int64_t cap_prev;          // room in prev_slots
    const uint32_t* prev_slots;
    const uint32_t* n_prev;
    const uint64_t* work;      // optional: the compacted list of changed slots (WorkList); elements are then
    int64_t n_work;            // cell selector * n_work + index into it
    uint32_t* counts;
    const uint32_t* offsets;
    uint64_t* first_key;       // count pass: keys of an element's first kRetestKept hits, from [kRetestKept t] on
    uint64_t* out_keys;
    int32_t write_pass;
    PSIM_HD void operator()(int64_t t) const
    {
        if (write_pass) {
            uint32_t c = counts[t];
            if (c == 0) return;
            if (c <= (uint32_t)kRetestKept) {
                for (uint32_t i = 0; i < c; ++i) out_keys[offsets[t] + i] = first_key[(int64_t)kRetestKept * t + i];
                return;
            }
        }
        // element = cell selector * n_adj + touched-slot number: the lanes of a wave look at the same
        // one of the nine cells for 64 consecutive touched slots, so they run loops of similar length
        // (and a wave whose cell is out of reach for all of them ends at once)
        int which;
        uint32_t x = 0;
        double reach_x = 0.0;
        bool work = false;
        if (this->work) {
            const uint32_t which_u = (uint32_t)(t / n_work);
            const int64_t ti = t - (int64_t)which_u * n_work;
            which = (int)which_u;
            const uint64_t w = this->work[ti];
            x = (uint32_t)(w >> 32);
            const uint32_t e = (uint32_t)w;
            reach_x = e != kNoEntry ? dseg[e] : 0.0;
            work = true;
        } else if (t < 10 * n_adj) {
            const uint32_t which_u = (uint32_t)t / (uint32_t)n_adj;
            const int64_t ti = t - (int64_t)which_u * n_adj;
            which = (int)which_u;
            if (ti < (int64_t)*n_touched) {
                const int64_t e = touched[ti];
                x = (uint32_t)(adj[e] >> 32);
                reach_x = dseg[e];
                work = changed[x] != 0;
            }
        } else {
            const int64_t t2 = t - 10 * n_adj;
            const uint32_t which_u = (uint32_t)t2 / (uint32_t)cap_prev;
            const int64_t ti = t2 - (int64_t)which_u * cap_prev;
            which = (int)which_u;
            if (ti < (int64_t)*n_prev) {
                x = prev_slots[ti];
                work = cseg[x] < 0 && changed[x] != 0;    // no contact any more: back at its initial position throughout
            }
        }
        uint32_t found = 0;
        if (work) {
            const CellBody bx = g.body[x];
            uint32_t px = bx.idx;
            int32_t rx = bx.r;
            uint32_t at = write_pass ? offsets[t] : 0u;
            const bool skip_cell = prune && which != 0 && which != 5 &&
                cell_beyond_reach(g, cell_size, bx, which, (double)(rx + r_max) + reach_x + double_of_float_bits(*dmax_bits));
            // Two phases, so that the lanes of a wave do the expensive part together: first the
            // cell's slots are filtered by reach (cheap, every lane in step), the few that pass
            // are remembered; then each lane settles its candidates (ordering key, versioned
            // positions, contact test).  Done inside the filter loop, that part would run once
            // per loop trip in which any lane of the wave had a candidate.
            uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            int nc = 0;
            auto settle = [&](uint32_t y) {
                const CellBody by = g.body[y];
                uint32_t py = by.idx;
                int32_t ry = by.r;
                uint64_t tk = pair_key(g, x, y);
                vec2 X = versioned_pos(x, tk, cseg, adj, n_adj, hkeys, hsi, out, g.body);
                vec2 Y = versioned_pos(y, tk, cseg, adj, n_adj, hkeys, hsi, out, g.body);
                bool hit = (px < py) ? in_contact(X, Y, rx, ry) : in_contact(Y, X, ry, rx);
                if (hit) {
                    if (write_pass) out_keys[at + found] = tk;
                    else if (found < (uint32_t)kRetestKept) first_key[(int64_t)kRetestKept * t + found] = tk;
                    ++found;
                }
            };
            if (!skip_cell)
            for_each_slot_in_cell(g, x, which, [&](uint32_t y) {
                int32_t sy = cseg[y];
                if (y < x && changed[y]) return;      // two changed slots: the lower slot's pass emits the pair
                const CellBody by = g.body[y];
                int32_t ry = by.r;
                // no version of x and y can touch if their initial positions are further
                // apart than the contact distance plus both displacement bounds
                double reach = (double)(rx + ry) + reach_x + (sy >= 0 ? dseg[sy] : 0.0);
                double ddx = by.x - bx.x, ddy = by.y - bx.y;
                if (ddx * ddx + ddy * ddy > reach * reach) return;
                if (nc == 0) c0 = y; else if (nc == 1) c1 = y; else if (nc == 2) c2 = y; else if (nc == 3) c3 = y;
                else settle(y);                       // more than four near neighbours in one cell: rare
                ++nc;
            });
            if (nc > 0) settle(c0);
            if (nc > 1) settle(c1);
            if (nc > 2) settle(c2);
            if (nc > 3) settle(c3);
        }
        if (!write_pass) counts[t] = found;
    }
};

// contacts of the current hypothesis whose two slots are unchanged keep the status the
// previous iteration gave them: carry them over (same two-pass scheme; its counts follow
// the re-test's in the same array)
struct ContactCarry {
    int64_t n;
    const uint64_t* hkeys;
    const uint32_t* hsi;
    const uint32_t* hsj;
    const uint8_t* changed;
    uint32_t* counts;
    const uint32_t* offsets;
    uint64_t* out_keys;
    int32_t write_pass;
    const double* geo;         // [3h] of this iteration's execution: 0, 0 where the pair did not touch
    PSIM_HD void operator()(int64_t h) const
    {
        if (write_pass) {
            if (counts[h]) out_keys[offsets[h]] = hkeys[h];
            return;
        }
        const bool touched = geo[3 * (size_t)h] != 0.0 || geo[3 * (size_t)h + 1] != 0.0;
        counts[h] = (touched && changed[hsi[h]] == 0 && changed[hsj[h]] == 0) ? 1u : 0u;
    }
};

// One launch for the three element ranges of a re-test pass: [0, nt) re-tests, [nt, nt + n0)
// carried-over contacts (n0 = size of the hypothesis) and, in the count pass,
// [nt + n0, nt + n0 + nH) the contacts still waiting on a predecessor.
struct RetestPass {
    ContactRetest retest;
    ContactCarry carry;
    CountPending pending;
    int64_t nt, n0;
    PSIM_HD void operator()(int64_t i) const
    {
        if (i < nt) retest(i);
        else if (i < nt + n0) carry(i - nt);
        else pending(i - nt - n0);
    }
};

// every key of `a` occurs in the sorted, duplicate-free list `b` (binary search): `a` is a
// subset of `b`, and with equal lengths and duplicate-free `a` the two are equal as sets
struct AllFound {
    int64_t n;
    const uint64_t* a;
    const uint64_t* b;
    int64_t nb;
    uint32_t* missing;
    PSIM_HD void operator()(int64_t i) const
    {
        uint64_t key = a[i];
        int64_t lo = 0, hi = nb;
        while (lo < hi) {
            int64_t mid = lo + (hi - lo) / 2;
            if (b[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (!(lo < nb && b[lo] == key)) aggregated_inc(missing);
    }
};

struct CompareKeys {
    int64_t n;
    const uint64_t* a;
    const uint64_t* b;
    uint32_t* mismatch;
    PSIM_HD void operator()(int64_t i) const
    {
        if (a[i] != b[i]) aggregated_inc(mismatch);
    }
};

struct ResetSeg {
    int64_t n;
    const uint64_t* adj;
    int32_t* cseg;
    PSIM_HD void operator()(int64_t e) const { cseg[(uint32_t)(adj[e] >> 32)] = -1; }
};

// converged: write each touched slot's final position back (index space + slot copy).  The
// settled list may hold pairs that did not touch when their turn came (geo = 0, 0: they left
// both particles as they were); a slot none of whose pairs touched is not a touched slot.
struct ContactFinalizePositions {
    int64_t n_adj;
    const uint64_t* adj;
    int32_t* cseg;
    const uint32_t* hsi;
    const vec2* out;
    CellBody* body;
    vec2* pos;
    uint8_t* moved;           // slot -> touched by a contact this step
    uint32_t* newly;          // [entry] 1 where a slot is touched for the first time this step
    const double* geo;        // [3h]
    PSIM_HD void operator()(int64_t e) const
    {
        uint32_t x = (uint32_t)(adj[e] >> 32);
        bool last = (e + 1 == n_adj) || ((uint32_t)(adj[e + 1] >> 32) != x);
        newly[e] = 0u;
        if (!last) return;
        cseg[x] = -1;
        bool any = false;
        for (int64_t k = e; k >= 0 && (uint32_t)(adj[k] >> 32) == x && !any; --k) {
            uint32_t hk = (uint32_t)adj[k];
            any = geo[3 * (size_t)hk] != 0.0 || geo[3 * (size_t)hk + 1] != 0.0;
        }
        if (!any) return;
        newly[e] = moved[x] == 0 ? 1u : 0u;
        moved[x] = 1;
        uint32_t h = (uint32_t)adj[e];
        int side = (hsi[h] == x) ? 0 : 1;
        vec2 s = out[2 * h + side];
        body[x].x = s.x;
        body[x].y = s.y;
        if (pos) pos[body[x].idx] = s;      // (null: the caller exports the touched slots' states instead)
    }
};

// ... and, after the velocity pass, its final velocity and temperature
struct ContactFinalizeVelocities {
    int64_t n_adj;
    const uint64_t* adj;
    const uint32_t* hsi;
    const VelState* out;
    const CellBody* body;
    vec2* vel;
    double* temp;
    VelState* slot_out;       // optional: written (by slot) instead of vel / temp
    PSIM_HD void operator()(int64_t e) const
    {
        uint32_t x = (uint32_t)(adj[e] >> 32);
        bool last = (e + 1 == n_adj) || ((uint32_t)(adj[e + 1] >> 32) != x);
        if (!last) return;
        uint32_t h = (uint32_t)adj[e];
        int side = (hsi[h] == x) ? 0 : 1;
        VelState s = out[2 * h + side];
        if (slot_out) { slot_out[x] = s; return; }
        uint32_t p = body[x].idx;
        vel[p] = s.vel;
        temp[p] = s.temp;
    }
};

// contact log (particle indices, per-contact energies) + summands of the
// sweep totals (particle_system.py:165-168: heat only when > 0)
struct ContactLog {
    int64_t n;
    const uint32_t* hsi;
    const uint32_t* hsj;
    const CellBody* body;
    const double* vals;
    int32_t* log_ij;          // may be null
    double* log_vals;         // may be null
    int64_t log_base, log_cap;
    double* sum_pe;
    double* sum_ke;
    double* sum_heat;
    // a grid that is one rank's band of rows: only the contacts whose home cell (the cell of
    // the lower-index particle, slot hsi) lies in the owned rows are counted and logged;
    // the others are counted by the neighbouring rank
    int32_t own_lo, own_hi;
    uint32_t* owned_count;    // optional: number of contacts counted and logged
    // the settled list may hold pairs that did not touch when their turn came (geo = 0, 0):
    // they are no contacts, and are left out like the ones another rank counts
    const double* geo;        // [3h]
    PSIM_HD void operator()(int64_t h) const
    {
        double pe = vals[3 * h], ke = vals[3 * h + 1], heat = vals[3 * h + 2];
        int32_t row = (int32_t)(body[hsi[h]].cxy >> 16);
        bool own = row >= own_lo && row < own_hi && (geo[3 * h] != 0.0 || geo[3 * h + 1] != 0.0);
        sum_pe[h] = own ? pe : 0.0;
        sum_ke[h] = own ? ke : 0.0;
        sum_heat[h] = (own && heat > 0) ? heat : 0.0;
        int64_t at = log_base + h;
        if (!own) {
            if (log_ij && at < log_cap) {
                log_ij[2 * at] = -1; log_ij[2 * at + 1] = -1;
                log_vals[3 * at] = 0.0; log_vals[3 * at + 1] = 0.0; log_vals[3 * at + 2] = 0.0;
            }
            return;
        }
        if (owned_count) aggregated_inc(owned_count);
        if (log_ij && at < log_cap) {
            log_ij[2 * at] = (int32_t)body[hsi[h]].idx;
            log_ij[2 * at + 1] = (int32_t)body[hsj[h]].idx;
            log_vals[3 * at] = pe; log_vals[3 * at + 1] = ke; log_vals[3 * at + 2] = heat;
        }
    }
};

// --------------------------------------------------------------- stage 7/8: conduction + cooling
struct GatherThermal {
    int64_t n;
    const CellBody* body;
    const double* temp;
    double* ctemp;            // slot -> temperature
    PSIM_HD void operator()(int64_t k) const { ctemp[k] = temp[body[k].idx]; }
};

// particle_system.py:271-336 as a per-particle gather (Jacobi: reads the
// pre-sweep temperatures, so the per-pair updates commute):
//   touching iff d2 < (ri+rj)^2 ; E = k*(T_p1 - T_p2) ; dT_p1 -= E/m_p1 ; dT_p2 += E/m_p2
// then :498 T += dT and :555-557 T *= (1 - damping).
struct Conduction {
    int64_t n;
    GridView g;
    const double* ctemp;
    const double* mass;       // index space (own mass only)
    double heat_coeff;
    double cool_factor;       // 1 - damping
    int32_t do_conduct, do_cool;
    double* temp;             // index space, written
    double* delta_out;        // optional debug, index space
    PSIM_HD void operator()(int64_t kk) const
    {
        uint32_t k = (uint32_t)kk;
        const CellBody me = g.body[k];
        uint32_t p = me.idx;
        double T = ctemp[k];
        double delta = 0.0;
        if (do_conduct) {
            double m = mass[p];
            for_each_neighbour_slot(g, k, [&](uint32_t y) {
                const CellBody o = g.body[y];
                double dx = o.x - me.x, dy = o.y - me.y;
                double d2 = dx * dx + dy * dy;
                double cd = (double)(me.r + o.r);
                if (d2 < cd * cd) {
                    double Tq = ctemp[y];
                    if (p < o.idx) {
                        double diff = T - Tq;
                        if (diff != 0) delta -= (heat_coeff * diff) / m;
                    } else {
                        double diff = Tq - T;
                        if (diff != 0) delta += (heat_coeff * diff) / m;
                    }
                }
            });
        }
        double T1 = T + delta;
        if (do_cool) T1 = T1 * cool_factor;
        temp[p] = T1;
        if (delta_out) delta_out[p] = delta;
    }
};

// plain copy of 64-bit words as a kernel of this library.  (The runtime's own device-to-device
// copy, queued on the side stream while the walk holds most wave slots, took 5 ms for 2 MB.)
struct CopyWords {
    int64_t n;
    const uint64_t* src;
    uint64_t* dst;
    PSIM_HD void operator()(int64_t i) const { dst[i] = src[i]; }
};

// results of a sharded stage, gathered in sorted / slot order, back to index order
struct ScatterSorted {
    int64_t n;
    const uint32_t* perm;
    const uint32_t* order;      // optional: in_sorted[k] belongs to sorted position order[k] (null: k itself)
    const vec2* in_sorted;
    vec2* out;
    PSIM_HD void operator()(int64_t k) const { out[perm[order ? order[k] : (uint32_t)k]] = in_sorted[k]; }
};

// ---- conduction without a second full sweep.  After the collision stage a pair of
// slots that no contact touched is not within contact distance (it was tested against
// the very same positions by ContactDetect, and an initial contact between untouched
// slots would have survived into the final contact list and touched them), unless its
// squared distance is exactly zero.  So, when no such coincident pair was seen, the
// touching pairs of particle_system.py:322 all involve a touched slot, and conduction
// is a sweep around the touched slots only.
struct WriteMoved {
    int64_t n_adj;
    const uint64_t* adj;
    const uint32_t* newly;
    const uint32_t* offs;      // exclusive scan of newly
    uint32_t* list;            // already offset to the first free entry
    PSIM_HD void operator()(int64_t e) const
    {
        if (newly[e]) list[offs[e]] = (uint32_t)(adj[e] >> 32);
    }
};

// element = cell selector * n_moved + touched slot's list entry.  Emits, for every touching pair
// found, two entries (target slot << 32 | source slot).  A pair of two touched slots
// is emitted by the one with the lower particle index.  count / scan / write.
struct ConductionPairs {
    int64_t n_moved;
    GridView g;
    const uint32_t* list;
    const uint8_t* moved;
    uint32_t* counts;
    const uint32_t* offsets;
    uint64_t* entries;
    int32_t write_pass;
    PSIM_HD void operator()(int64_t t) const
    {
        if (write_pass && counts[t] == 0) return;
        const uint32_t which_u = (uint32_t)t / (uint32_t)n_moved;     // element = cell selector * n_moved + list entry
        int64_t m = t - (int64_t)which_u * n_moved;
        int which = (int)which_u;
        uint32_t x = list[m];
        const CellBody bx = g.body[x];
        uint32_t found = 0;
        uint32_t at = write_pass ? offsets[t] : 0u;
        int32_t lo = 0, hi = 0;
        if (cell_slot_range(g, bx, which, &lo, &hi))
        sweep_slots4(g.body, (const int32_t*)nullptr, lo, hi, x, [&](const CellBody& by, int32_t, uint32_t y) {
            double dx = by.x - bx.x, dy = by.y - bx.y;
            double d2 = dx * dx + dy * dy;
            double cd = (double)(bx.r + by.r);
            if (!(d2 < cd * cd)) return;
            if (moved[y] && by.idx < bx.idx) return;
            if (write_pass) {
                entries[at + found] = ((uint64_t)x << 32) | (uint64_t)y;
                entries[at + found + 1] = ((uint64_t)y << 32) | (uint64_t)x;
            }
            found += 2;
        });
        if (!write_pass) counts[t] = found;
    }
};

// entries sorted by target: the first entry of each target sums its sources in order
// (particle_system.py:325-336 for the pair (p1, p2) = (lower index, higher index))
struct ConductionApply {
    int64_t n_entries;
    const uint64_t* entries;
    const CellBody* body;
    const double* temp;        // index space, pre-conduction
    const double* mass;
    double heat_coeff;
    double* delta;             // index space, zero elsewhere
    // When the collision stage left its results in slot order (slot_state, valid where
    // moved[slot]) instead of writing them to temp[]: read them there, and file the sum under
    // the target's first entry (slot_delta[e]) for the caller to export.
    const VelState* slot_state;
    const uint8_t* moved;
    double* slot_delta;
    PSIM_HD PSIM_FORCEINLINE double temp_of(uint32_t slot, uint32_t p) const
    {
        return (slot_state && moved[slot]) ? slot_state[slot].temp : temp[p];
    }
    PSIM_HD void operator()(int64_t e) const
    {
        uint32_t t = (uint32_t)(entries[e] >> 32);
        if (e > 0 && (uint32_t)(entries[e - 1] >> 32) == t) return;
        uint32_t pt = body[t].idx;
        double Tt = temp_of(t, pt), m = mass[pt], d = 0.0;
        for (int64_t k = e; k < n_entries && (uint32_t)(entries[k] >> 32) == t; ++k) {
            uint32_t ys = (uint32_t)entries[k];
            uint32_t ps = body[ys].idx;
            double Ts = temp_of(ys, ps);
            if (pt < ps) {
                double diff = Tt - Ts;
                if (diff != 0) d -= (heat_coeff * diff) / m;
            } else {
                double diff = Ts - Tt;
                if (diff != 0) d += (heat_coeff * diff) / m;
            }
        }
        if (slot_delta) slot_delta[e] = d; else delta[pt] = d;
    }
};

// ---- The collision stage split over several GPUs by bands of grid rows.  A rank solves the
// band it owns plus halo rows on a grid of that band only, leaves the state arrays alone and
// exports, as 64-bit words, (a) the state of every touched slot of its OWN rows, (b) the
// conduction sum of every target in its own rows, (c) for each of its two seams, a record of
// every contact that involves a slot of the two rows along the seam.  The ranks exchange the
// blocks; the results are applied only when both ranks along every seam hold identical seam
// records (then every rank's view of its neighbours' rows was the true one, and the union of
// the owned results is the sequential sweep's result).
constexpr int kExportHeaderWords = 32;
constexpr int kExportStateWords = 6;      // particle index, x, y, vx, vy, T
constexpr int kExportDeltaWords = 2;      // particle index, sum of conduction terms
constexpr int kExportSeamWords = 11;      // index i << 32 | index j, positions after (i, j), velocity + temperature after (i, j)

PSIM_HD PSIM_FORCEINLINE uint64_t f64_bits(double v) { uint64_t b; __builtin_memcpy(&b, &v, 8); return b; }
PSIM_HD PSIM_FORCEINLINE double bits_f64(uint64_t b) { double v; __builtin_memcpy(&v, &b, 8); return v; }

struct ExportStateFlag {
    int64_t n_moved;
    const uint32_t* list;      // touched slots
    const CellBody* body;
    int32_t own_lo, own_hi;
    uint32_t* flag;
    PSIM_HD void operator()(int64_t m) const
    {
        int32_t row = (int32_t)(body[list[m]].cxy >> 16);
        flag[m] = (row >= own_lo && row < own_hi) ? 1u : 0u;
    }
};
struct ExportStateWrite {
    int64_t n_moved;
    const uint32_t* list;
    const uint32_t* flag;
    const uint32_t* at;
    const CellBody* body;
    const VelState* slot_state;
    uint64_t* out;
    PSIM_HD void operator()(int64_t m) const
    {
        if (!flag[m]) return;
        uint32_t x = list[m];
        const CellBody b = body[x];
        const VelState v = slot_state[x];
        uint64_t* o = out + (size_t)kExportStateWords * at[m];
        o[0] = (uint64_t)b.idx; o[1] = f64_bits(b.x); o[2] = f64_bits(b.y);
        o[3] = f64_bits(v.vel.x); o[4] = f64_bits(v.vel.y); o[5] = f64_bits(v.temp);
    }
};
struct ExportDeltaFlag {
    int64_t n_entries;
    const uint64_t* entries;   // sorted by target slot
    const CellBody* body;
    int32_t own_lo, own_hi;
    uint32_t* flag;
    PSIM_HD void operator()(int64_t e) const
    {
        uint32_t t = (uint32_t)(entries[e] >> 32);
        bool first = e == 0 || (uint32_t)(entries[e - 1] >> 32) != t;
        int32_t row = (int32_t)(body[t].cxy >> 16);
        flag[e] = (first && row >= own_lo && row < own_hi) ? 1u : 0u;
    }
};
struct ExportDeltaWrite {
    int64_t n_entries;
    const uint64_t* entries;
    const uint32_t* flag;
    const uint32_t* at;
    const CellBody* body;
    const double* slot_delta;  // [entry]
    uint64_t* out;
    PSIM_HD void operator()(int64_t e) const
    {
        if (!flag[e]) return;
        uint64_t* o = out + (size_t)kExportDeltaWords * at[e];
        o[0] = (uint64_t)body[(uint32_t)(entries[e] >> 32)].idx;
        o[1] = f64_bits(slot_delta[e]);
    }
};
// contacts of the settled list that involve a slot of row seam - 1 or row seam
// (pairs of the list that did not touch when their turn came, geo = 0, 0, are no contacts: which
// of them a rank's list holds depends on that rank's hypotheses, not on the state)
struct SeamFlag {
    int64_t n;
    const uint32_t* hsi;
    const uint32_t* hsj;
    const CellBody* body;
    int32_t seam;
    uint32_t* flag;
    const double* geo;        // [3h]
    PSIM_HD void operator()(int64_t h) const
    {
        int32_t ri = (int32_t)(body[hsi[h]].cxy >> 16), rj = (int32_t)(body[hsj[h]].cxy >> 16);
        bool touched = geo[3 * h] != 0.0 || geo[3 * h + 1] != 0.0;
        flag[h] = (touched && (ri == seam || ri == seam - 1 || rj == seam || rj == seam - 1)) ? 1u : 0u;
    }
};
struct SeamWrite {
    int64_t n;
    const uint32_t* hsi;
    const uint32_t* hsj;
    const uint32_t* flag;
    const uint32_t* at;
    const CellBody* body;
    const vec2* outp;
    const VelState* outv;
    uint64_t* out;             // already offset to the first free record
    PSIM_HD void operator()(int64_t h) const
    {
        if (!flag[h]) return;
        uint64_t* o = out + (size_t)kExportSeamWords * at[h];
        const vec2 a = outp[2 * h], b = outp[2 * h + 1];
        const VelState va = outv[2 * h], vb = outv[2 * h + 1];
        o[0] = ((uint64_t)body[hsi[h]].idx << 32) | (uint64_t)body[hsj[h]].idx;
        o[1] = f64_bits(a.x); o[2] = f64_bits(a.y); o[3] = f64_bits(b.x); o[4] = f64_bits(b.y);
        o[5] = f64_bits(va.vel.x); o[6] = f64_bits(va.vel.y); o[7] = f64_bits(va.temp);
        o[8] = f64_bits(vb.vel.x); o[9] = f64_bits(vb.vel.y); o[10] = f64_bits(vb.temp);
    }
};
struct ExportHeader {
    uint64_t words[kExportHeaderWords];
    uint64_t* out;
    PSIM_HD void operator()(int64_t i) const { out[i] = words[i]; }       // launched over kExportHeaderWords
};
// the blocks' headers side by side (one copy to the host instead of one per block)
struct GatherHeaders {
    int64_t world;
    int64_t block_words;
    const uint64_t* blocks;
    uint64_t* out;
    PSIM_HD void operator()(int64_t i) const
    {
        int64_t r = i / kExportHeaderWords, w = i - r * kExportHeaderWords;
        out[i] = blocks[r * block_words + w];
    }
};
struct ApplyState {
    int64_t n_rec;
    const uint64_t* rec;
    vec2* pos;
    vec2* vel;
    double* temp;
    PSIM_HD void operator()(int64_t k) const
    {
        const uint64_t* o = rec + (size_t)kExportStateWords * k;
        uint32_t p = (uint32_t)o[0];
        vec2 x, v;
        x.x = bits_f64(o[1]); x.y = bits_f64(o[2]); v.x = bits_f64(o[3]); v.y = bits_f64(o[4]);
        pos[p] = x;
        vel[p] = v;
        temp[p] = bits_f64(o[5]);
    }
};
struct ApplyDelta {
    int64_t n_rec;
    const uint64_t* rec;
    double* delta;
    PSIM_HD void operator()(int64_t k) const
    {
        const uint64_t* o = rec + (size_t)kExportDeltaWords * k;
        delta[(uint32_t)o[0]] = bits_f64(o[1]);
    }
};
// number of slots in the cells [cell_lo, cell_hi) (relative to the first cell the offsets hold)
struct OwnedSlots {
    const int32_t* offsets;
    int64_t cell_lo, cell_hi;
    uint32_t* out;
    PSIM_HD void operator()(int64_t) const { *out = (uint32_t)(offsets[cell_hi] - offsets[cell_lo]); }
};
// slot records were made from the drifted positions; bring them up to the current ones
struct RefreshBodyPositions {
    int64_t n;
    CellBody* body;
    const vec2* pos;
    PSIM_HD void operator()(int64_t k) const
    {
        vec2 p = pos[body[k].idx];
        body[k].x = p.x;
        body[k].y = p.y;
    }
};
// the grid cell ids of DriftKeys again, from the records it wrote (the cell sort works in place)
struct CellKeysFromPack {
    int64_t n;
    const PackedBody* pack;
    double cell_size;
    int32_t gw, gh;
    uint32_t* ckey;
    uint32_t* cperm;
    PSIM_HD void operator()(int64_t i) const
    {
        int32_t cx = cell_coord(pack[i].x, cell_size, gw);
        int32_t cy = cell_coord(pack[i].y, cell_size, gh);
        ckey[i] = (uint32_t)(cy * gw + cx);
        cperm[i] = (uint32_t)i;
    }
};
// slot -> velocity and temperature of its particle (the velocity half's initial states in slot order)
struct GatherSlotState {
    int64_t n;
    const CellBody* body;
    const vec2* vel;
    const double* temp;
    VelState* out;
    PSIM_HD void operator()(int64_t k) const
    {
        uint32_t p = body[k].idx;
        VelState s;
        s.vel = vel[p]; s.temp = temp[p]; s.unused = 0.0;
        out[k] = s;
    }
};

struct Cooling {
    int64_t n;
    double* temp;
    double cool_factor;
    PSIM_HD void operator()(int64_t i) const { temp[i] = temp[i] * cool_factor; }
};

// --------------------------------------------------------------- stage 9/10: walls, explosion flags
// particle_system.py:500-516: both masks from the pre-clamp position, low clamp
// then high clamp, velocity negated once if either fired.  Fused with the
// explosion test particle_system.py:658 (T >= COLOR_MAX_TEMP).
struct WallsFlag {
    int64_t n;
    vec2* pos;
    vec2* vel;
    const int32_t* radii;
    double* temp;
    double W, H, explode_temp;
    int32_t do_walls, do_flag;
    uint8_t* boom;
    uint32_t* boom_count;
    const double* delta;      // non-null: apply T = (T + delta) [* cool_factor] first (:498, :555-557)
    double cool_factor;
    int32_t do_cool;
    PSIM_HD void operator()(int64_t i) const
    {
        if (delta) {
            double T1 = temp[i] + delta[i];
            if (do_cool) T1 = T1 * cool_factor;
            temp[i] = T1;
        }
        if (do_walls) {
            vec2 p = pos[i], v = vel[i];
            double r = (double)radii[i];
            bool lo = (p.x - r) < 0, hi = (p.x + r) > W;
            if (lo) p.x = r;
            if (hi) p.x = W - r;
            if (lo || hi) v.x *= -1;
            bool lo2 = (p.y - r) < 0, hi2 = (p.y + r) > H;
            if (lo2) p.y = r;
            if (hi2) p.y = H - r;
            if (lo2 || hi2) v.y *= -1;
            if (lo || hi || lo2 || hi2) { pos[i] = p; vel[i] = v; }
        }
        if (do_flag) {
            uint8_t b = temp[i] >= explode_temp ? 1 : 0;
            boom[i] = b;
            if (b) aggregated_inc(boom_count);
        }
    }
};

// --------------------------------------------------------------- explosions
struct FlagToU32 {
    int64_t n;
    const uint8_t* flag;
    uint32_t* out;
    int32_t invert;
    PSIM_HD void operator()(int64_t i) const { out[i] = invert ? (flag[i] ? 0u : 1u) : (flag[i] ? 1u : 0u); }
};

struct ExploderList {
    int64_t n;
    const uint8_t* boom;
    const uint32_t* rank;     // exclusive scan of boom
    uint32_t* list;
    PSIM_HD void operator()(int64_t i) const { if (boom[i]) list[rank[i]] = (uint32_t)i; }
};

// particle_system.py:667-702 for one exploding particle: neighbours from the
// start-of-step grid (3x3 cells around int(x / cell_size), unclipped), minus
// itself and other exploders; weights 1/d2 (d2 == 0 -> 1e-6).  Emits one
// (neighbour << 32 | exploder rank) entry per affected neighbour.
struct ExplosionGather {
    int64_t n_ex;
    const uint32_t* list;
    GridView g;
    double cell_size;
    const vec2* pos;
    const uint8_t* boom;
    double* cell_w;            // pass 0: [9 r + cell] sum of the weights of that cell's neighbours
    uint32_t* cell_count;      // pass 0: [9 r + cell] their number; pass 1: read
    const uint32_t* cell_at;   // pass 1: [9 r + cell] where that cell's entries start inside the exploder's run
    const uint32_t* offsets;   // pass 1: exclusive scan of ex_count
    uint64_t* entries;         // pass 1
    int32_t write_pass;
    // element = 9 * exploder rank + cell of the 3x3 neighbourhood in (dy, dx) order: the nine
    // cells of one exploding particle are gathered by nine lanes
    PSIM_HD void operator()(int64_t i) const
    {
        const int64_t r = i / 9;
        const int which = (int)(i - r * 9);
        const int dy = which / 3 - 1, dx = which % 3 - 1;
        uint32_t p = list[r];
        vec2 P = pos[p];
        int64_t cx = (int64_t)(P.x / cell_size), cy = (int64_t)(P.y / cell_size);
        double total = 0.0;
        uint32_t cnt = 0;
        int64_t x = cx + dx, y = cy + dy;
        if (!(x < 0 || x >= g.gw || y < g.row_lo || y >= g.row_hi)) {
            uint32_t at = write_pass ? offsets[r] + cell_at[i] : 0u;
            int64_t c = y * g.gw + x;
            for (int32_t k = g.offsets[c]; k < g.offsets[c + 1]; ++k) {
                uint32_t q = g.body[k].idx;
                if (q == p || boom[q]) continue;
                if (write_pass) {
                    entries[at + cnt] = ((uint64_t)q << 32) | (uint64_t)r;
                } else {
                    vec2 Q = pos[q];
                    double ddx = Q.x - P.x, ddy = Q.y - P.y;
                    double d2 = ddx * ddx + ddy * ddy;
                    if (d2 == 0) d2 = 1e-6;
                    total += 1.0 / d2;
                }
                ++cnt;
            }
        }
        if (!write_pass) {
            cell_w[i] = total;
            cell_count[i] = cnt;
        }
    }
};

// per exploding particle: the nine cells' sums and counts folded in cell order
// (particle_system.py:664-700: energy, neighbour count, total weight)
struct ExplosionCombine {
    int64_t n_ex;
    const uint32_t* list;
    const double* mass;
    const double* temp;
    double efficiency;
    const double* cell_w;
    const uint32_t* cell_count;
    uint32_t* cell_at;
    double* ex_energy;
    double* ex_total_w;
    uint32_t* ex_count;
    PSIM_HD void operator()(int64_t r) const
    {
        uint32_t p = list[r];
        double total = 0.0;
        uint32_t cnt = 0;
        for (int c = 0; c < 9; ++c) {
            cell_at[9 * r + c] = cnt;
            total += cell_w[9 * r + c];
            cnt += cell_count[9 * r + c];
        }
        ex_energy[r] = (mass[p] * temp[p]) * efficiency;
        ex_total_w[r] = total;
        ex_count[r] = cnt;
    }
};

// particle_system.py:702-712: each affected neighbour receives its impulses in
// ascending order of the exploding particle's index (entries are sorted).
struct ExplosionApply {
    int64_t n_entries;
    const uint64_t* entries;
    const uint32_t* list;
    const vec2* pos;
    const double* mass;
    const double* ex_energy;
    const double* ex_total_w;
    vec2* vel;
    PSIM_HD void operator()(int64_t e) const
    {
        uint32_t q = (uint32_t)(entries[e] >> 32);
        if (e > 0 && (uint32_t)(entries[e - 1] >> 32) == q) return;
        vec2 Q = pos[q], v = vel[q];
        double mq = mass[q];
        for (int64_t k = e; k < n_entries && (uint32_t)(entries[k] >> 32) == q; ++k) {
            uint32_t r = (uint32_t)entries[k];
            vec2 P = pos[list[r]];
            double dx = Q.x - P.x, dy = Q.y - P.y;
            double d2 = dx * dx + dy * dy;
            if (d2 == 0) d2 = 1e-6;
            double w = 1.0 / d2;
            double share = ex_energy[r] * (w / ex_total_w[r]);
            double dv = sqrt(2 * share / mq);
            double dist = sqrt(d2);
            v.x += (dx / dist) * dv;
            v.y += (dy / dist) * dv;
        }
        vel[q] = v;
    }
};

// particle_system.py:715-722: boolean-mask compaction, order preserved
struct CompactState {
    int64_t n;
    const uint8_t* boom;
    const uint32_t* new_index;   // exclusive scan of survivors
    const vec2 *pos, *vel, *acc;
    const double *mass, *inv_mass, *temp;
    const int32_t* radii;
    vec2 *pos_o, *vel_o, *acc_o;
    double *mass_o, *inv_mass_o, *temp_o;
    int32_t* radii_o;
    PSIM_HD void operator()(int64_t i) const
    {
        if (boom[i]) return;
        uint32_t o = new_index[i];
        pos_o[o] = pos[i]; vel_o[o] = vel[i]; acc_o[o] = acc[i];
        mass_o[o] = mass[i]; inv_mass_o[o] = inv_mass[i]; temp_o[o] = temp[i];
        radii_o[o] = radii[i];
    }
};

// --------------------------------------------------------------- diagnostics
// particle_system.py:765-783 summands + momentum
struct EnergyTerms {
    int64_t n;
    const vec2* vel;
    const double* mass;
    const double* temp;
    double* ke;
    double* te;
    double* px;
    double* py;
    PSIM_HD void operator()(int64_t i) const
    {
        vec2 v = vel[i];
        double m = mass[i];
        ke[i] = 0.5 * m * (v.x * v.x + v.y * v.y);
        te[i] = m * temp[i];
        px[i] = m * v.x;
        py[i] = m * v.y;
    }
};

// What the reference's draw() computes per particle before it calls the rasteriser
// (particle_system.py:587-624): int-truncated coordinates and the colour of the
// temperature through the keyframe ramp (constants.py:37-45, particle_system.py:565-585:
// first segment with pos1 <= t <= pos2, channels c1 (1 - u) + c2 u truncated to int).
// 12 bytes per particle leave the device instead of the 80 of the full state.
struct RenderFeed {
    int64_t n;
    const vec2* pos;
    const double* temp;
    int32_t* xy;               // (n, 2)
    uint8_t* rgba;             // (n, 4), alpha 255
    double t_min, t_max;       // constants.py:33-34
    PSIM_HD void operator()(int64_t i) const
    {
        const double kpos[7] = {0.0, 0.1, 0.25, 0.4, 0.55, 0.7, 1.0};
        const double kcol[7][3] = {{0, 0, 0}, {75, 0, 130}, {0, 0, 255}, {0, 255, 0}, {255, 255, 0}, {255, 0, 0},
                                   {255, 255, 255}};
        double T = temp[i];
        // np.nan_to_num(nan=min, posinf=max, neginf=min), then clip((T - min) / (max - min), 0, 1)
        if (T != T) T = t_min;
        double t = (T - t_min) / (t_max - t_min);
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        int r = 255, g = 255, b = 255;
        for (int k = 0; k < 6; ++k) {
            if (kpos[k] <= t && t <= kpos[k + 1]) {
                double u = (t - kpos[k]) / (kpos[k + 1] - kpos[k]);
                r = (int)(kcol[k][0] * (1 - u) + kcol[k + 1][0] * u);
                g = (int)(kcol[k][1] * (1 - u) + kcol[k + 1][1] * u);
                b = (int)(kcol[k][2] * (1 - u) + kcol[k + 1][2] * u);
                break;
            }
        }
        vec2 p = pos[i];
        xy[2 * i] = (int32_t)p.x;          // int(): truncation toward zero
        xy[2 * i + 1] = (int32_t)p.y;
        rgba[4 * i] = (uint8_t)r; rgba[4 * i + 1] = (uint8_t)g; rgba[4 * i + 2] = (uint8_t)b; rgba[4 * i + 3] = 255;
    }
};

// one term of the exact pairwise potential: phi_i = sum_j m_j / sqrt(d_ij^2 + s) over d_ij^2 > 0
// (the all-pairs counterpart of particle_system.py:43-63, which only pairs sibling cells)
struct PairPotentialRow {
    int64_t n;
    const vec2* pos;
    const double* mass;
    double softening;
    double* out;               // m_i phi_i
    PSIM_HD void operator()(int64_t i) const
    {
        vec2 p = pos[i];
        double phi = 0.0;
        for (int64_t j = 0; j < n; ++j) {
            double dx = pos[j].x - p.x, dy = pos[j].y - p.y;
            double d2 = dx * dx + dy * dy;
            if (d2 > 0) phi += mass[j] / sqrt(d2 + softening);
        }
        out[i] = mass[i] * phi;
    }
};

struct FillI32 {
    int64_t n;
    int32_t* p;
    int32_t v;
    PSIM_HD void operator()(int64_t i) const { p[i] = v; }
};

// upload helper: int64 radii of the reference (particle_system.py:376) -> int32
struct ConvertRadii {
    int64_t n;
    const int64_t* in;
    int32_t* out;
    PSIM_HD void operator()(int64_t i) const { out[i] = (int32_t)in[i]; }
};
struct WidenRadii {
    int64_t n;
    const int32_t* in;
    int64_t* out;
    PSIM_HD void operator()(int64_t i) const { out[i] = (int64_t)in[i]; }
};
struct InverseMass {
    int64_t n;
    const double* mass;
    double* inv;
    PSIM_HD void operator()(int64_t i) const { inv[i] = 1.0 / mass[i]; }   // particle_system.py:374
};

}  // namespace psim

// Host orchestration of one physics step on one device: owns the device
// arrays and sequences the kernels in the reference's stage order
// (particle_system.py:518-563).  Included by psim_capi.cpp after a back end.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <vector>
#include "psim_kernels.h"
#if !defined(PSIM_HOSTLOOP)
#include "psim_tree_hip.h"
#include "psim_contact_hip.h"
#endif

namespace psim {

struct Config {
    double bounds_w, bounds_h;
    double gravity_constant, softening, theta;
    double heat_coeff, restitution, pos_correct, explosion_eff, damping, explode_temp;
    int32_t cell_size, grid_w, grid_h, solver_iterations;
    int32_t stage_mask;
    int32_t walk_mode;      // 0: wave-cooperative walk, two groups per wave; 1: one lane per particle; 2: one group per wave
    int32_t debug;          // keep per-step debug arrays (contact log, delta temps)
    int32_t overlap;        // 1: grid build + first contact count pass on a side stream under the walk
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { be::dev_free(p); }
    // contents are NOT preserved on growth
    void ensure(size_t n)
    {
        if (n > cap) {
            be::dev_free(p);
            p = nullptr;
            size_t want = n + n / 8 + 64;
            p = (T*)be::dev_alloc(want * sizeof(T));
            cap = want;
        }
    }
    // like ensure(), and a newly allocated buffer is cleared (on stream s)
    template <class StreamT>
    void ensure_zero(StreamT s, size_t n)
    {
        if (n > cap) {
            ensure(n);
            be::dev_memset(s, p, 0, cap * sizeof(T));
        }
    }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(cap, o.cap); }
};

struct ExplosionEvent {
    int32_t particle;       // index before compaction
    int32_t neighbours;
    double energy;
};

static inline int ceil_log2_u64(uint64_t v)
{
    int b = 0;
    while (b < 63 && (1ull << b) < v) ++b;
    return b;
}

class Sim {
public:
    Config cfg;
    be::Stream st;
    int64_t n = 0;
    bool tree_valid = false;
    int64_t tree_n = 0;            // particle count the grid was built for
    int64_t walk_n = 0;            // particle count the tree (rec, perm, spos) was built for
    uint32_t n_internal = 0;
    int sorted_key = 0, sorted_ckey = 0;
    int slot_bits = 1;            // pair-key field width, from the largest cell population
    int32_t max_radius = 0;       // largest radius uploaded (an upper bound after explosions removed particles)

    // state (index space; reference layout particle_system.py:370-376, radii narrowed)
    DevBuf<vec2> pos, vel, acc, acc2;
    DevBuf<double> mass, inv_mass, temp;
    DevBuf<int32_t> radii;
    // compaction targets
    DevBuf<vec2> pos_b, vel_b, acc_b;
    DevBuf<double> mass_b, inv_mass_b, temp_b;
    DevBuf<int32_t> radii_b;

    // Morton / tree
    DevBuf<uint64_t> key[2];
    DevBuf<uint32_t> perm[2];
    DevBuf<vec2> spos;
    DevBuf<double> smass;
    DevBuf<PackedBody> pack;       // index space: what the gathers read (written by DriftKeys)
    // order in which the walk groups its targets: sorted positions along a Hilbert curve over
    // the quadtree cells of depth hilbert_levels() (hruns: where each cell's run of particles
    // starts and ends in Morton order; hcurve: run lengths by curve position, then their scan)
    DevBuf<uint32_t> hord, hruns, hcurve, curve_pos;
    bool hilbert_ready = false;     // hord describes the current tree
    int curve_levels = -1;          // what curve_pos was made for
    double curve_w = 0.0, curve_h = 0.0;
    DevBuf<int8_t> L;
    DevBuf<uint32_t> owned, base;
    DevBuf<NodeRec> rec;
    DevBuf<uint32_t> node_first, node_last, node_depth[2], node_id[2];
    DevBuf<double> pe_node;
    DevBuf<uint64_t> far_list;           // nodes the cooperative emit leaves to per-depth passes
    DevBuf<uint32_t> far_count, far_offs;
    DevBuf<uint64_t> far_nodes;
    bool tree_cooperative = false;             // the last tree came from the cooperative kernels
    uint32_t tree_far_nodes = 0;               // ... which left this many nodes (spanning tiles) to tree_far_kernel
    static constexpr int kEmitWords = 16;      // counters[16..19]: the cooperative tree kernels' words
    DevBuf<int32_t> depth_off;
    DevBuf<double> w2tab;
    int32_t depth_off_host[40];
    // What the walk compares squared distances with at the root: W^2 / theta^2 when theta^2 is a
    // power of two (w^2 < theta^2 d^2 is then decided exactly as w^2 / theta^2 < d^2), else W^2.
    // The high word is the root's NodeRec::tag, the low word is shared by all depths.
    bool walk_fold = false;
    uint64_t w2_cmp_bits = 0;
    uint32_t root_tag() const { return (uint32_t)(w2_cmp_bits >> 32); }

    // grid
    DevBuf<uint32_t> ckey[2], cperm[2];
    DevBuf<int32_t> offsets;
    DevBuf<CellBody> cbody;
    // The grid of the step: the whole domain (one GPU), or a band of whole cell rows -- the rows
    // a rank owns plus halo rows -- when the collision stage is split over several GPUs.
    // ns = slots (particles) in it.  With `deferred` the collision stage leaves the state arrays
    // alone: results stay in slot order (cbody, slot_state, slot_delta) and are exported.
    struct Band { int32_t row_lo, row_hi, own_lo, own_hi; };
    Band band{0, 0, 0, 0};
    bool band_partial = false;
    int64_t ns = 0;
    bool full_grid_valid = false;      // cbody / offsets hold the whole grid of this step's drifted positions
    bool deferred = false;
    bool slot_state_full = false;      // slot_state holds every slot's state (sweeps after the first read it)
    DevBuf<uint32_t> bflag, bat, rkey[2], rperm[2];
    DevBuf<VelState> slot_state;
    DevBuf<double> slot_delta;
    DevBuf<uint64_t> seam_rec[2];      // records of the contacts along the band's lower / upper seam
    int64_t seam_n[2] = {0, 0};
    DevBuf<uint64_t> hdr_dev;
    int64_t own_slots = 0;
    int64_t slab_fallbacks = 0;        // steps whose seam records disagreed (solved again on the whole grid)
    DevBuf<double> ctemp;
    DevBuf<int32_t> cseg;
    bool cseg_ready = false;

    // collisions
    size_t ccap = 0;
    DevBuf<uint64_t> h0[2], hc[2], hn[2], adj[2];
    DevBuf<uint32_t> hsi, hsj, done, pcount, poffs, moved_list;
    DevBuf<uint32_t> round_tiles;             // per tile of 1024 contacts: left, first label, last label (cooperative rounds)
    DevBuf<uint32_t> tflag, tpos, tlist;     // touched slots of the current hypothesis (flags, scan, list of first entries)
    DevBuf<uint64_t> slot_sig, dsig;         // incremental re-test: per-slot digest of the previous iteration's versions; this iteration's
    DevBuf<uint8_t> slot_changed;
    bool incremental_dirty = false;
    DevBuf<uint32_t> prev_slots[2];          // touched slots of the previous / the current iteration
    DevBuf<uint32_t> wflag, wpos;            // compacted list of the changed slots (large hypotheses)
    DevBuf<uint64_t> wlist;
    DevBuf<uint64_t> first_key;
    DevBuf<uint8_t> moved;
    DevBuf<double> cdelta;
    DevBuf<uint64_t> cond_entries[2];
    uint32_t n_moved = 0, n_coincident = 0;
    bool moved_valid = false;      // moved / moved_list describe this step's collision stage
    DevBuf<int32_t> prevl;
    DevBuf<vec2> outp;             // [2h + side] position after hypothesised contact h
    DevBuf<VelState> outv;         // [2h + side] velocity and temperature after settled contact h
    DevBuf<double> cgeo;           // [3h] contact normal and PE change
    DevBuf<double> cvals, sum_pe, sum_ke, sum_heat, dseg;
    DevBuf<int32_t> log_ij;
    DevBuf<double> log_vals, delta_dbg;
    int64_t log_n = 0;

    // explosions
    DevBuf<uint8_t> boom;
    DevBuf<uint32_t> tmp32, tmp32b, ex_list, ex_count, ex_offs;
    DevBuf<double> ex_energy, ex_total, ex_cell_w;
    DevBuf<uint32_t> ex_cell_count, ex_cell_at;
    DevBuf<uint64_t> ex_entries[2];
    size_t ecap = 0;
    void* ex_stage = nullptr;        // page-locked staging for the explosion events
    size_t ex_stage_cap = 0;
    std::vector<ExplosionEvent> events;
    std::vector<int32_t> last_exploded;

    // scratch
    DevBuf<uint32_t> sort_tmp, scan_tmp, counters;   // counters: 16 words
    DevBuf<uint32_t> sort_tmp2, scan_tmp2, counters2;   // scratch of the side stream
    // What the host reads back once at the end of a step: words 0-5 the last collision sweep's
    // totals (three doubles), word 6 the number of exploding particles.
    DevBuf<uint32_t> step_words;
    bool totals_pending = false;
    DevBuf<uint32_t> walk_ctr;       // chunk counter of the persistent walk (two words, zero between launches)
    // Waves the walk keeps resident (0 = one wave per chunk of targets, the whole grid at once).
    // A whole-grid walk is dispatched ahead of anything the other stream submits: that stream's
    // kernels only get wave slots once the walk's last workgroup has been placed.  With the side
    // stream in use the walk therefore runs as a fixed set of waves that take chunks of targets
    // from a counter and leave the remaining slots of every SIMD to the other stream for the
    // whole walk.  What runs there is latency-bound and slows down by about the share of slots
    // it is denied, while the walk slows down less than its share (4 Mi, walk alone: 3.90 / 4.08 /
    // 4.38 / 4.86 ms at 8 / 7 / 6 / 5 slots).  With collisions on, the grid build and the whole
    // position half of the collision stage run underneath 6.5 of 8 slots (26 walk waves per CU).
    // 4 Mi: step 5.04 / 4.97 / 4.96 / 5.38 ms at 6.25 / 6.5 / 6.75 / 7; 16 Mi: 14.02 / 13.86 /
    // 13.86 / 14.98 -- flat up to the point where the side work no longer ends with the walk,
    // then steeply worse.  Without collisions only the grid build is there: 7 slots from 8 Mi
    // particles on, the whole grid below.  Two free slots per SIMD are also what the exchange
    // kernels of a sharded step need (workgroups of up to 512 lanes).
    bool sharded_step = false;     // the step in progress is driven through the stage_* calls
    bool positions_under_walk() const
    {
        return cfg.overlap != 0;
    }
    // (Shifting slots from the walk to the side stream when the side work is the longer of the
    // two -- dense or clumped states -- was tried with a controller fed by the measured wait:
    // the step did not change; what one stream gains the other loses.)
    // Wave slots per SIMD (of 8) the walk holds while the side stream works: measured optimum of
    // the whole step (C2 4 Mi: 4.65 / 4.54 / 4.66 / 4.72 ms at 5.5 / 5.75 / 6 / 6.25; C3 16 Mi: 13.05 /
    // 12.83 / 12.71 / 12.92).  PSIM_WALK_SLOTS overrides it at build time, for such sweeps.
    double walk_slots_beside_collisions() const
    {
#ifdef PSIM_WALK_SLOTS
        return PSIM_WALK_SLOTS;
#else
        return n >= (8ll << 20) ? 6.0 : 5.75;
#endif
    }
    int64_t walk_resident_waves() const
    {
        double per_simd = 0;
        if (cfg.overlap != 0 && n > 0) {
            if (!(cfg.stage_mask & kStageCollisions)) per_simd = n >= (8ll << 20) ? 7 : 0;
            else per_simd = sharded_step ? 6 : walk_slots_beside_collisions();
        }
        return per_simd > 0 ? (int64_t)(be::compute_units() * 4 * per_simd) : 0;
    }
    be::Stream st2;
    be::Event ev_side;
    // (the tree's end and the walk's end are the stage timer's marks 3 and 4)
    be::Event ev_release() { return timer.event(3); }
    be::Event ev_walk() { return timer.event(4); }
    bool walk_in_flight = false;       // a walk was submitted this step and has not been seen to finish
    // Scans on the side lane use small workgroups only while the walk is still holding its wave
    // slots; once it has finished the larger ones are faster (about 2x at tens of millions of items).
    bool small_groups_now(bool lane_small)
    {
        refresh_groups();
        return lane_small && walk_in_flight;
    }
    // Has the walk finished?  The one-lane-per-element kernels are launched one wave per
    // workgroup while it runs and four waves per workgroup when the device is free.
    void refresh_groups()
    {
        if (walk_in_flight && be::event_done(ev_walk())) walk_in_flight = false;
        be::narrow_stream() = st2;
        be::wide_groups() = !walk_in_flight;
    }
    // A lane = a stream with the scratch its sorts, scans and small readbacks use.  The
    // position half of the collision stage runs on the side lane underneath the walk.
    struct Lane {
        be::Stream s;
        DevBuf<uint32_t>* sort_tmp;
        DevBuf<uint32_t>* scan_tmp;
        uint32_t* ctr;             // 16 words
        be::Mailbox* box;
        bool small_groups;         // its scans use workgroups that fit next to the walk
    };
    Lane main_lane() { return Lane{st, &sort_tmp, &scan_tmp, counters.p, &mailbox, false}; }
    Lane side_lane() { return Lane{st2, &sort_tmp2, &scan_tmp2, counters2.p, &mailbox2, true}; }
    // what the position half leaves for the velocity half of the same sweep
    int rounds_hint = 6;               // dependency rounds launched before the first re-test of an iteration
    uint32_t settled_contacts = 0, settled_touching = 0, settled_rounds = 0;
    uint64_t* settled_adj = nullptr;
    bool positions_settled = false;    // sweep 0's position half has run for this step
    uint32_t cond_ne = 0;              // conduction entries (sorted) waiting for ConductionApply
    int cond_sorted = 0;
    bool cond_pairs_ready = false;
    bool predetected = false;      // sweep 0's count pass already ran on the side stream
    DevBuf<double> red_tmp, red_out, terms[4];

    // per-step outputs
    double coll_totals[3] = {0, 0, 0};
    int64_t n_candidates = 0, n_contacts = 0, n_exploded = 0;
    int64_t walk_visits = 0, walk_interactions = 0;
    int64_t solver_rounds = 0, solver_fixpoint_iters = 0;
    bool count_walk = false;
    be::StageTimer timer;
    be::Mailbox mailbox, mailbox2;

    ~Sim()
    {
        timer.destroy();
        be::mailbox_destroy(mailbox);
        be::mailbox_destroy(mailbox2);
        be::host_free_pinned(ex_stage);
        be::event_destroy(ev_side);
        be::stream_destroy(st2);
    }

    explicit Sim(const Config& c) : cfg(c), st(0)
    {
        be::mailbox_create(mailbox);
        be::mailbox_create(mailbox2);
        counters.ensure(kEmitWords + 4);
        counters2.ensure(16);
        step_words.ensure(16);
        walk_ctr.ensure_zero(st, 16);
        st2 = be::stream_create();
        ev_side = be::event_create();
        red_tmp.ensure(be::kReduceBlocks);
        red_out.ensure(8);
        depth_off.ensure(40);
        w2tab.ensure(kKeyLevels);
        set_config(c);
    }

    void set_config(const Config& c)
    {
        if (c.grid_w <= 0 || c.grid_h <= 0 || c.cell_size <= 0) throw std::invalid_argument("psim: bad grid geometry");
        if ((uint64_t)c.grid_w * (uint64_t)c.grid_h > kMaxCells) throw std::invalid_argument("psim: too many grid cells");
        if (c.grid_w > kMaxGridDim || c.grid_h > kMaxGridDim) throw std::invalid_argument("psim: grid dimension above 65535 cells");
        if (!(c.bounds_w > 0) || !(c.bounds_h > 0)) throw std::invalid_argument("psim: bounds must be positive");
        bool geometry_changed = (c.grid_w != cfg.grid_w || c.grid_h != cfg.grid_h || c.cell_size != cfg.cell_size);
        if (c.bounds_w != cfg.bounds_w || c.bounds_h != cfg.bounds_h || c.theta != cfg.theta) tree_valid = false;
        cfg = c;
        if (geometry_changed) tree_n = 0;
        {
            double theta2 = c.theta * c.theta;          // particle_system.py:88 theta * theta * dist_sq
            int e = 0;
            walk_fold = theta2 > 0.0 && frexp(theta2, &e) == 0.5 && e > -200 && e < 200;
            double w2 = c.bounds_w * c.bounds_w;         // quadtree.py:25 root width, squared as in :88
            double cmp = walk_fold ? w2 / theta2 : w2;   // exact: a power-of-two scaling
            memcpy(&w2_cmp_bits, &cmp, 8);
            // 31 levels must fit under the root's exponent (tag = root_tag - (depth << 21) stays positive)
            if (!(cmp > 1e-250) || !(cmp < 1e250)) throw std::invalid_argument("psim: bounds / theta out of range");
        }
        double w2[kKeyLevels];
        double w = c.bounds_w;
        for (int d = 0; d < kKeyLevels; ++d) { w2[d] = w * w; w = w / 2; }   // quadtree.py:25,132 width halves per level
        be::copy_h2d(st, w2tab.p, w2, sizeof(w2));
        be::sync(st);
    }

    // -------------------------------------------------------------- state I/O
    void upload(int64_t count, const double* h_pos, const double* h_vel, const double* h_acc, const double* h_mass,
                const double* h_temp, const int64_t* h_radii)
    {
        if (count < 0 || count > 0x7fff0000LL) throw std::invalid_argument("psim: particle count out of range");
        n = count;
        size_t c = (size_t)count;
        max_radius = 0;
        for (size_t i = 0; i < c; ++i) {
            int64_t r = h_radii[i];
            if (r < 0 || r > (1 << 24)) throw std::invalid_argument("psim: radius out of range");
            if (r > max_radius) max_radius = (int32_t)r;
        }
        pos.ensure(c); vel.ensure(c); acc.ensure(c); acc2.ensure(c);
        mass.ensure(c); inv_mass.ensure(c); temp.ensure(c); radii.ensure(c);
        be::copy_h2d(st, pos.p, h_pos, c * 16);
        if (h_vel) be::copy_h2d(st, vel.p, h_vel, c * 16); else be::dev_memset(st, vel.p, 0, c * 16);
        if (h_acc) be::copy_h2d(st, acc.p, h_acc, c * 16); else be::dev_memset(st, acc.p, 0, c * 16);
        be::copy_h2d(st, mass.p, h_mass, c * 8);
        be::copy_h2d(st, temp.p, h_temp, c * 8);
        DevBuf<int64_t> stage;
        stage.ensure(c);
        be::copy_h2d(st, stage.p, h_radii, c * 8);
        be::launch(st, count, ConvertRadii{count, stage.p, radii.p});
        be::launch(st, count, InverseMass{count, mass.p, inv_mass.p});
        be::sync(st);
        tree_valid = false;
        tree_n = 0;
        cseg_ready = false;
    }

    void download(double* h_pos, double* h_vel, double* h_acc, double* h_mass, double* h_inv_mass, double* h_temp,
                  int64_t* h_radii)
    {
        size_t c = (size_t)n;
        if (h_pos) be::copy_d2h(st, h_pos, pos.p, c * 16);
        if (h_vel) be::copy_d2h(st, h_vel, vel.p, c * 16);
        if (h_acc) be::copy_d2h(st, h_acc, acc.p, c * 16);
        if (h_mass) be::copy_d2h(st, h_mass, mass.p, c * 8);
        if (h_inv_mass) be::copy_d2h(st, h_inv_mass, inv_mass.p, c * 8);
        if (h_temp) be::copy_d2h(st, h_temp, temp.p, c * 8);
        if (h_radii) {
            DevBuf<int64_t> stage;
            stage.ensure(c);
            be::launch(st, n, WidenRadii{n, radii.p, stage.p});
            be::copy_d2h(st, h_radii, stage.p, c * 8);
            be::sync(st);
        }
        be::sync(st);
    }

    // small synchronous readback (whole 32-bit words) through the mailbox; with check_fault it
    // also fetches the word a scan / sort look-back sets if it gave up (asked for where the
    // values read size the next launches: tree build, step end)
    void readback(void* dst, const void* dev, size_t bytes, bool check_fault = false)
    {
        readback2(dst, dev, bytes, nullptr, 0, check_fault);
    }
    void readback2(void* dst, const void* a, size_t bytes_a, const void* b, size_t bytes_b, bool check_fault = false)
    {
        readback_on(main_lane(), dst, a, bytes_a, b, bytes_b, check_fault);
    }
    void readback_on(const Lane& ln, void* dst, const void* a, size_t bytes_a, const void* b = nullptr, size_t bytes_b = 0,
                     bool check_fault = false)
    {
        if ((bytes_a | bytes_b) & 3) throw std::runtime_error("psim: internal: readback of partial words");
        uint32_t na = (uint32_t)(bytes_a / 4), nb = (uint32_t)(bytes_b / 4);
        if (na + nb + 1 > 256) throw std::runtime_error("psim: internal: readback too large");
        uint32_t words[256];
        if (!check_fault) {
            be::mailbox_fetch(ln.s, *ln.box, words, a, na, b, nb);
        } else if (nb == 0) {
            be::mailbox_fetch(ln.s, *ln.box, words, a, na, be::lookback_fault_word(), 1);
            if (words[na] != 0) throw std::runtime_error("psim: internal: scan / sort look-back timed out");
        } else {
            throw std::runtime_error("psim: internal: readback2 with fault check");
        }
        memcpy(dst, words, bytes_a);
        if (nb) memcpy((char*)dst + bytes_a, words + na, bytes_b);
    }
    uint32_t read_u32(const uint32_t* d) { return read_u32(main_lane(), d); }
    uint32_t read_u32(const Lane& ln, const uint32_t* d)
    {
        uint32_t v = 0;
        readback_on(ln, &v, d, 4);
        return v;
    }

    GridView grid_view() const
    {
        GridView g;
        g.body = cbody.p;
        g.offsets.p = offsets.p;
        g.offsets.base = band_partial ? (uint32_t)band.row_lo * (uint32_t)cfg.grid_w : 0u;
        g.gw = cfg.grid_w;
        g.gh = cfg.grid_h;
        g.slot_bits = slot_bits;
        g.row_lo = band_partial ? band.row_lo : 0;
        g.row_hi = band_partial ? band.row_hi : cfg.grid_h;
        g.own_lo = deferred ? band.own_lo : 0;
        g.own_hi = deferred ? band.own_hi : cfg.grid_h;
        return g;
    }
    int64_t band_cells() const
    {
        return band_partial ? (int64_t)(band.row_hi - band.row_lo) * cfg.grid_w : (int64_t)cfg.grid_w * cfg.grid_h;
    }

    // -------------------------------------------------------------- stages
    void keys_and_drift(bool drift)
    {
        size_t c = (size_t)n;
        key[0].ensure(c); key[1].ensure(c); perm[0].ensure(c); perm[1].ensure(c);
        ckey[0].ensure(c); ckey[1].ensure(c); cperm[0].ensure(c); cperm[1].ensure(c);
        DriftKeys k;
        k.n = n; k.pos = pos.p; k.vel = vel.p; k.acc = acc.p;
        k.key = key[0].p; k.perm = perm[0].p; k.ckey = ckey[0].p; k.cperm = cperm[0].p;
        k.W = cfg.bounds_w; k.H = cfg.bounds_h; k.cell_size = (double)cfg.cell_size;
        k.gw = cfg.grid_w; k.gh = cfg.grid_h; k.do_drift = drift ? 1 : 0;
        pack.ensure(c);
        k.mass = mass.p; k.radii = radii.p; k.pack = pack.p;
        be::launch(st, n, k);
        cell_keys_sorted = false;
        full_grid_valid = false;
    }

    // particle_system.py:732-763
    void build_grid() { build_grid(st, sort_tmp); }
    bool cell_keys_sorted = false;     // ckey[0] / cperm[0] no longer hold what DriftKeys wrote
    void build_grid(be::Stream s, DevBuf<uint32_t>& tmp)
    {
        size_t c = (size_t)n;
        uint64_t cells = (uint64_t)cfg.grid_w * (uint64_t)cfg.grid_h;
        tmp.ensure(be::sort_temp_words(c));
        if (cell_keys_sorted && n > 0)
            be::launch(s, n, CellKeysFromPack{n, pack.p, (double)cfg.cell_size, cfg.grid_w, cfg.grid_h, ckey[0].p, cperm[0].p});
        cell_keys_sorted = true;
        uint32_t* k[2] = {ckey[0].p, ckey[1].p};
        uint32_t* v[2] = {cperm[0].p, cperm[1].p};
        int bits = ceil_log2_u64(cells);
        if (bits < 1) bits = 1;
        sorted_ckey = be::radix_sort_pairs<uint32_t>(s, k, v, (uint64_t)n, 0, bits, tmp.p);
        offsets.ensure(cells + 1);
        if (n > 0) {
            be::launch(s, n, SegmentOffsets{n, ckey[sorted_ckey].p, offsets.p, (uint32_t)cells});
            be::launch(s, (int64_t)cells + 1, SegmentOffsetsTail{n, ckey[sorted_ckey].p, offsets.p, (uint32_t)cells});
        } else {
            be::dev_memset(s, offsets.p, 0, (cells + 1) * sizeof(int32_t));
        }
        cbody.ensure(c);
        be::launch(s, n, GatherCell{n, cperm[sorted_ckey].p, ckey[sorted_ckey].p, pack.p, cbody.p, cfg.grid_w, 0u});
        tree_n = n;
        ns = n;
        band = Band{0, cfg.grid_h, 0, cfg.grid_h};
        band_partial = false;
        full_grid_valid = true;
    }
    // The whole grid, built late in a step that worked on a band of rows (explosions and the
    // full conduction sweep want every cell); the cell ids DriftKeys wrote are still in place.
    void ensure_full_grid()
    {
        if (full_grid_valid) return;
        build_grid(st, sort_tmp);
        // (the slot records are made from the drifted positions; the collision stage may have moved particles since)
        if (n > 0) be::launch(st, n, RefreshBodyPositions{n, cbody.p, pos.p});
    }

    // The grid of the rows [b.row_lo, b.row_hi) only: the particles of those rows compacted in
    // ascending index, sorted by cell (relative to the band's first cell), CSR over the
    // band's cells, slot records.  One small readback (the band's population sizes the sort).
    void build_grid_band(const Lane& ln, const Band& b)
    {
        if (b.row_lo < 0 || b.row_hi > cfg.grid_h || b.row_lo >= b.row_hi || b.own_lo < b.row_lo || b.own_hi > b.row_hi ||
            b.own_lo > b.own_hi)
            throw std::invalid_argument("psim: bad band of grid rows");
        if (b.row_lo == 0 && b.row_hi == cfg.grid_h) {
            build_grid(ln.s, *ln.sort_tmp);
            band = b;
            return;
        }
        if (cell_keys_sorted)
            be::launch(ln.s, n, CellKeysFromPack{n, pack.p, (double)cfg.cell_size, cfg.grid_w, cfg.grid_h, ckey[0].p, cperm[0].p});
        cell_keys_sorted = false;
        const size_t c = (size_t)n;
        const int64_t bcells = (int64_t)(b.row_hi - b.row_lo) * cfg.grid_w;
        const uint32_t cell_base = (uint32_t)b.row_lo * (uint32_t)cfg.grid_w;
        band = b;
        band_partial = true;
        full_grid_valid = false;
        offsets.ensure((size_t)cfg.grid_w * (size_t)cfg.grid_h + 1);
        ns = 0;
        // One pass where the input is long (the band's particles picked and compacted by a scan
        // whose items are computed from the cell ids; the output arrays then have to hold up to n
        // entries, the band's size not being known before), flag / scan / write kernels otherwise.
        const bool one_pass = be::kHasBandSelect && (uint64_t)n > (uint64_t)be::kScanSmall;
        if (n > 0) {
            ln.scan_tmp->ensure_zero(ln.s, be::scan_temp_words(c));
            if (one_pass) {
                rkey[0].ensure(c); rkey[1].ensure(c); rperm[0].ensure(c); rperm[1].ensure(c);
                be::band_select(ln.s, ckey[0].p, (uint64_t)n, cfg.grid_w, b.row_lo, b.row_hi, cell_base, rkey[0].p, rperm[0].p,
                                ln.scan_tmp->p, ln.ctr + 12, small_groups_now(ln.small_groups));
            } else {
                bflag.ensure(c); bat.ensure(c);
                be::launch(ln.s, n, BandFlag{n, ckey[0].p, cfg.grid_w, b.row_lo, b.row_hi, bflag.p});
                be::exclusive_scan_u32(ln.s, bflag.p, bat.p, (uint64_t)n, ln.scan_tmp->p, ln.ctr + 12, small_groups_now(ln.small_groups));
            }
            ns = read_u32(ln, ln.ctr + 12);
        }
        if (ns == 0) {
            be::dev_memset(ln.s, offsets.p, 0, ((size_t)bcells + 1) * sizeof(int32_t));
            return;
        }
        const size_t cs = (size_t)ns;
        if (!one_pass) {
            rkey[0].ensure(cs); rkey[1].ensure(cs); rperm[0].ensure(cs); rperm[1].ensure(cs);
            be::launch(ln.s, n, BandCompact{n, ckey[0].p, bflag.p, bat.p, cell_base, rkey[0].p, rperm[0].p});
        }
        ln.sort_tmp->ensure(be::sort_temp_words(cs));
        uint32_t* k[2] = {rkey[0].p, rkey[1].p};
        uint32_t* v[2] = {rperm[0].p, rperm[1].p};
        int bits = ceil_log2_u64((uint64_t)bcells);
        if (bits < 1) bits = 1;
        const int sorted = be::radix_sort_pairs<uint32_t>(ln.s, k, v, (uint64_t)ns, 0, bits, ln.sort_tmp->p);
        be::launch(ln.s, ns, SegmentOffsets{ns, rkey[sorted].p, offsets.p, (uint32_t)bcells});
        be::launch(ln.s, bcells + 1, SegmentOffsetsTail{ns, rkey[sorted].p, offsets.p, (uint32_t)bcells});
        cbody.ensure(c);
        be::launch(ln.s, ns, GatherCell{ns, rperm[sorted].p, rkey[sorted].p, pack.p, cbody.p, cfg.grid_w, cell_base});
    }

    // count pass of the first collision sweep (positions do not change between the drift and
    // the collision stage): per-slot contact counts, their scan, and the largest cell population
    // With a slot range [k_lo, k_hi) (a slab of whole grid rows when the caller cuts at cell
    // boundaries; any cut is valid) only that range is swept and scanned: ctr[0] is then the
    // range's contact count and offsets[] start at 0 at k_lo.  k_lo must be a multiple of 4.
    void detect_count(be::Stream s, DevBuf<uint32_t>& scan_scratch, uint32_t* ctr, bool small_groups, int64_t k_lo = 0,
                      int64_t k_hi = -1)
    {
        size_t c = (size_t)ns;
        if (k_hi < 0) k_hi = ns;
        const int64_t cells = band_cells();
        GridView g = grid_view();
        pcount.ensure(c); poffs.ensure(c); first_key.ensure((size_t)kDetectKept * c / 2 + 1); scan_scratch.ensure_zero(s, be::scan_temp_words(c));
        be::dev_memset(s, ctr, 0, 64);
        ContactDetect d;
        d.prune = detect_prune ? 1 : 0; d.r_max = max_radius; d.cell_size = (double)cfg.cell_size; d.margin = detect_margin();
        d.n = k_hi - k_lo; d.k0 = (uint32_t)k_lo; d.g = g; d.counts = pcount.p; d.offsets = poffs.p; d.first_pair = reinterpret_cast<uint32_t*>(first_key.p);
        d.out_keys = nullptr; d.write_pass = 0; d.cand_counter = cfg.debug ? ctr + 1 : nullptr;
        d.coincident_counter = ctr + 3;
        be::launch(s, k_hi - k_lo, d);
        be::exclusive_scan_u32(s, pcount.p + k_lo, poffs.p + k_lo, (uint64_t)(k_hi - k_lo), scan_scratch.p, ctr, small_groups);
        int64_t cwaves = (cells + 63) / 64;
        tmp32.ensure((size_t)cwaves + (size_t)(cwaves + 63) / 64 + 64);
        be::launch(s, cwaves * 64, WaveMaxOccupancy{cells, offsets.p, tmp32.p});
        uint32_t* cur = tmp32.p;
        uint32_t* nxt = tmp32.p + cwaves;
        int64_t left = cwaves;
        while (left > 4096) {
            int64_t folded = (left + 63) / 64;
            be::launch(s, folded * 64, WaveMaxU32{left, cur, nxt});
            uint32_t* t = cur; cur = nxt; nxt = t;       // the folded list always fits where the longer one was
            left = folded;
        }
        be::launch(s, 64, FinalMax{left, cur, ctr + 2});
    }

    // The element array by the per-element kernels: leaves, then internal nodes, then the
    // centres of mass one launch per depth.  (The HIP build normally uses the cooperative
    // pass of psim_tree_hip.h and comes here only when that pass declines: coinciding
    // particles below the deepest level, or a tile with too many far-reaching nodes.)
    void emit_per_element(const uint64_t* skey, uint32_t m)
    {
        size_t mm = m ? m : 1;
        TreeEmitLeaves e;
        e.n = n; e.L = L.p; e.owned = owned.p; e.base = base.p; e.spos = spos.p; e.smass = smass.p; e.rec = rec.p;
        e.node_first = node_first.p; e.node_depth = node_depth[0].p; e.node_id = node_id[0].p;
        be::launch(st, n, e);
        if (m > 0) {
            TreeEmitNodes en;
            en.count = m; en.n = n; en.key = skey; en.L = L.p; en.base = base.p; en.node_first = node_first.p;
            en.node_depth = node_depth[0].p; en.spos = spos.p; en.smass = smass.p; en.total_elems = (uint32_t)(n + m); en.rec = rec.p; en.root_tag = root_tag();
            en.node_last = node_last.p;
            be::launch(st, (int64_t)m, en);
        }
        if (m > 0) {
            sort_tmp.ensure(be::sort_temp_words(mm));
            uint32_t* dk[2] = {node_depth[0].p, node_depth[1].p};
            uint32_t* dv[2] = {node_id[0].p, node_id[1].p};
            int sd = be::radix_sort_pairs<uint32_t>(st, dk, dv, (uint64_t)m, 0, 8, sort_tmp.p);
            // segments 0..31: depths; segment 32 (kDepthDone): cells finished by TreeEmitNodes
            be::launch(st, (int64_t)m + 1, SegmentOffsets{(int64_t)m, node_depth[sd].p, depth_off.p, (uint32_t)(kKeyLevels + 1)});
            readback(depth_off_host, depth_off.p, sizeof(int32_t) * (kKeyLevels + 2));
            // one launch per depth, deepest first (folding the small top levels into one
            // single-workgroup launch was measured and is no faster than their launches; a single
            // bottom-up pass in which the last child to finish forms its parent was measured too:
            // records passed between workgroups inside one launch have to bypass the per-XCD L2
            // caches, and the pass took 440 us at 4 Mi particles against 300 us for the depth
            // sort plus these launches)
            TreeCom t0;
            t0.count = 0; t0.ids = node_id[sd].p; t0.depth = 0; t0.ch = node_children(); t0.rec = rec.p;
            for (int d = kMaxInternalDepth; d >= 0; --d) {
                int64_t cnt = (int64_t)depth_off_host[d + 1] - depth_off_host[d];
                if (cnt > 0) be::launch(st, cnt, t0.at_level(node_id[sd].p, d, depth_off_host[d], cnt));
            }
        }
    }

    // quadtree.py:177-196 (build) as sort + links + emit + bottom-up COM
    void build_tree()
    {
        size_t c = (size_t)n;
        tree_valid = false;
        n_internal = 0;
        walk_n = n;
        if (n == 0) { tree_valid = true; return; }
        sort_tmp.ensure(be::sort_temp_words(c));
        uint64_t* k[2] = {key[0].p, key[1].p};
        uint32_t* v[2] = {perm[0].p, perm[1].p};
        spos.ensure(c); smass.ensure(c); L.ensure(c); owned.ensure(c); base.ensure(c);
        scan_tmp.ensure_zero(st, be::scan_temp_words(c));
        uint32_t m = 0;
        bool emitted = false;
        const uint64_t* skey = nullptr;
        const uint32_t* sperm = nullptr;
        for (int attempt = 0;; ++attempt) {
            if (attempt == 0) {
                // radix passes on the high 32 bits, then FixRuns orders the (rare, short) runs
                // of particles that share them
                // enough leading bits that a cell of that depth holds a quarter of a particle on
                // average (whole 8-bit digits)
                int lead = (ceil_log2_u64((uint64_t)n) + 2 + 7) / 8 * 8;
                if (lead < 8) lead = 8;
                if (lead > 64) lead = 64;
                int half = be::radix_sort_pairs<uint64_t>(st, k, v, (uint64_t)n, 64 - lead, 64, sort_tmp.p);
                be::dev_memset(st, counters.p + 4, 0, 4);
                be::launch(st, n, FixRuns{n, key[half].p, perm[half].p, counters.p + 4, lead < 64 ? 64 - lead : 0});
                sorted_key = half;
            } else {
                // a long run of near-coincident particles was left unsorted: full LSD sort
                be::dev_memset(st, counters.p + 4, 0, 4);      // (the kernels below look at this word)
                uint64_t* k2[2] = {key[sorted_key].p, key[sorted_key ^ 1].p};
                uint32_t* v2[2] = {perm[sorted_key].p, perm[sorted_key ^ 1].p};
                int r = be::radix_sort_pairs<uint64_t>(st, k2, v2, (uint64_t)n, 0, 64, sort_tmp.p);
                sorted_key = sorted_key ^ r;
            }
            skey = key[sorted_key].p;
            sperm = perm[sorted_key].p;
            be::launch(st, n, GatherTree{n, sperm, pack.p, spos.p, smass.p});
            hilbert_ready = false;
            {
                // cells along the curve, each cell's particles in Morton order: no sort needed
                // (three streaming passes and a scan over the cells; they look at the word
                // FixRuns raises and do nothing when the keys are still to be sorted again)
                const int levels = hilbert_levels();
                const int64_t cells = 1ll << (2 * levels);
                const int shift = 64 - 2 * levels;
                curve_table(levels);
                hord.ensure(c); hruns.ensure(2 * (size_t)cells); hcurve.ensure((size_t)cells);
                be::dev_memset(st, hruns.p, 0, 2 * (size_t)cells * 4);
                be::launch(st, n, HilbertCellRuns{n, skey, shift, hruns.p, hruns.p + cells, counters.p + 4});
                be::launch(st, cells, HilbertCellCounts{cells, curve_pos.p, hruns.p, hruns.p + cells, hcurve.p, counters.p + 4});
                be::exclusive_scan_u32(st, hcurve.p, hcurve.p, (uint64_t)cells, scan_tmp.p, nullptr);
                be::launch(st, n, HilbertOrder{n, skey, shift, curve_pos.p, hruns.p, hcurve.p, hord.p, counters.p + 4});
                hilbert_ready = true;
            }
            be::launch(st, n, TreeLinks{n, skey, L.p, owned.p});
            be::exclusive_scan_u32(st, owned.p, base.p, (uint64_t)n, scan_tmp.p, counters.p);
#if !defined(PSIM_HOSTLOOP)
            {
                // The elements, by two cooperative kernels that read the node count from the
                // scan's total themselves (and do nothing when the tree is unusable: the key sort
                // incomplete, more nodes than the pool holds), so that the host reads back once,
                // after them: elements assembled tile by tile in LDS with the centre of mass of
                // every subtree that ends inside its tile, then the nodes that span tiles.
                // Buffers are sized for the pool's limit (as many internal nodes as particles).
                ensure_tree_buffers(c);
                const uint32_t tiles = be::emit_tiles(n);
                far_list.ensure((size_t)tiles * be::kEmitSlots); far_count.ensure(tiles);
                far_nodes.ensure((size_t)tiles * be::kEmitSlots); far_offs.ensure(tiles);
                uint32_t* ew = counters.p + kEmitWords;
                be::dev_memset(st, ew, 0, 16);
                be::TreeEmitArgs a;
                a.n = n; a.key = skey; a.L = L.p; a.base = base.p; a.spos = spos.p; a.smass = smass.p;
                a.counts = counters.p; a.root_tag = root_tag(); a.rec = rec.p; a.node_first = node_first.p;
                a.node_last = node_last.p; a.far_list = far_list.p; a.far_count = far_count.p; a.words = ew;
                a.far_offs = far_offs.p; a.far_nodes = far_nodes.p;
                be::launch_tree_emit(st, a, scan_tmp.p);
                uint32_t w[kEmitWords + 4];
                readback(w, counters.p, sizeof(w), true);      // [0] internal nodes, [4] FixRuns overflow, then the emit words
                m = w[0];
                if (attempt == 0 && w[4] != 0) continue;
                check_pool(m);
                if (w[kEmitWords + 1] != 0) throw std::runtime_error("psim: internal: tree nodes spanning tiles did not settle");
                emitted = w[kEmitWords + 2] == 0;
                tree_far_nodes = emitted ? w[kEmitWords + 0] : 0u;
            }
#else
            {
                uint32_t five[5];
                readback(five, counters.p, 20, true);      // [0] internal nodes, [4] FixRuns overflow
                m = five[0];
                if (attempt == 0 && five[4] != 0) continue;
                check_pool(m);
            }
#endif
            break;
        }
        n_internal = m;
        tree_cooperative = emitted;
        if (!emitted) {
            ensure_tree_buffers(m ? m : 1);
            emit_per_element(skey, m);
        }
        tree_valid = true;
    }

    // the reference's pool holds 4N+1 nodes = at most N subdivided cells (quadtree.py:165)
    void check_pool(uint32_t m) const
    {
        if ((uint64_t)m > (uint64_t)n) throw std::runtime_error("psim: quadtree node pool overflow (near-coincident particles)");
    }
    void ensure_tree_buffers(size_t mm)
    {
        rec.ensure((size_t)n + mm + 1);      // (one element of slack: the walk fetches elements two at a time)
        node_first.ensure(mm); node_last.ensure(mm); node_depth[0].ensure(mm); node_depth[1].ensure(mm);
        node_id[0].ensure(mm); node_id[1].ensure(mm); pe_node.ensure(mm);
    }

    // the walks take their targets in Hilbert order (every walk kernel accepts it; the
    // one-lane-per-particle walk gains nothing from it and is given it all the same, so that
    // the test-only host-loop build, which has no other walk, covers the ordering kernels)
    const uint32_t* walk_order() const { return hilbert_ready ? hord.p : nullptr; }
    // curve_pos[c]: where the quadtree cell with Morton path c (depth `levels`) comes when the
    // cells are put in the order of their centres along a fine Hilbert curve over the square
    // that bounds the domain.  (The quadtree's cells have the domain's aspect, 2:1 by default;
    // a curve over the cell grid itself would string such cells into long thin groups.)  Made
    // on the host whenever the depth or the domain changes: a particle count that crosses a
    // power of four, in practice once.
    void curve_table(int levels)
    {
        if (levels == curve_levels && cfg.bounds_w == curve_w && cfg.bounds_h == curve_h) return;
        const uint32_t cells = 1u << (2 * levels);
        const double side = cfg.bounds_w > cfg.bounds_h ? cfg.bounds_w : cfg.bounds_h;
        const double fx = cfg.bounds_w / side * 65536.0 / (double)(1u << levels);
        const double fy = cfg.bounds_h / side * 65536.0 / (double)(1u << levels);
        std::vector<uint64_t> by_curve(cells);
        for (uint32_t c = 0; c < cells; ++c) {
            uint32_t ix = 0, iy = 0;
            for (int l = levels - 1; l >= 0; --l) {
                uint32_t q = (c >> (2 * l)) & 3u;        // morton_key: y bit above x bit
                ix = (ix << 1) | (q & 1u);
                iy = (iy << 1) | (q >> 1);
            }
            uint32_t px = (uint32_t)(((double)ix + 0.5) * fx), py = (uint32_t)(((double)iy + 0.5) * fy);
            if (px > 65535u) px = 65535u;
            if (py > 65535u) py = 65535u;
            by_curve[c] = ((uint64_t)hilbert_index(px, py, 16) << 32) | c;
        }
        std::sort(by_curve.begin(), by_curve.end());
        std::vector<uint32_t> table(cells);
        for (uint32_t r = 0; r < cells; ++r) table[(uint32_t)by_curve[r]] = r;
        curve_pos.ensure(cells);
        be::sync(st);
        be::copy_h2d(st, curve_pos.p, table.data(), (size_t)cells * 4);
        be::sync(st);
        curve_levels = levels; curve_w = cfg.bounds_w; curve_h = cfg.bounds_h;
    }
    // depth of the cells the curve runs over: about 4 particles per cell (inside a cell the order
    // is Morton's; 16 per cell cost 2.5 % more walk trips at 4 Mi particles)
    int hilbert_levels() const
    {
        int levels = (ceil_log2_u64((uint64_t)n) - 2 + 1) / 2;
        return levels < 1 ? 1 : (levels > 11 ? 11 : levels);
    }

    NodeChildren node_children() const
    {
        NodeChildren c;
        c.n = n; c.key = key[sorted_key].p; c.L = L.p; c.base = base.p; c.spos = spos.p; c.smass = smass.p;
        c.node_first = node_first.p; c.node_last = node_last.p; c.rec = rec.p;
        return c;
    }

    void gravity(vec2* out);   // defined after the back end's walk kernel (psim_capi.cpp)
    void gravity_range(int64_t lo, int64_t hi, vec2* out, bool sorted_out);

    // particle_system.py:439-474
    void ensure_contact_capacity(const Lane& ln, size_t want)
    {
        if (want <= ccap) return;
        size_t c = want + want / 4 + 1024;
        for (int i = 0; i < 2; ++i) { h0[i].ensure(c); hc[i].ensure(c); hn[i].ensure(c); adj[i].ensure(2 * c); }
        round_tiles.ensure(3 * (c / 1024 + 1));
        hsi.ensure(c); hsj.ensure(c); done.ensure(c); prevl.ensure(2 * c); outp.ensure(2 * c); outv.ensure(2 * c);
        prev_slots[0].ensure(2 * c); prev_slots[1].ensure(2 * c);
        cgeo.ensure(3 * c); cvals.ensure(3 * c);
        tflag.ensure(2 * c); tpos.ensure(2 * c); tlist.ensure(2 * c);
        sum_pe.ensure(c); sum_ke.ensure(c); sum_heat.ensure(c);
        ln.sort_tmp->ensure(be::sort_temp_words(2 * c));
        ccap = c;
    }

    // The collision stage in two halves (see resolve_pair_positions): collisions_positions()
    // settles sweep 0's contact list and the pushed-apart positions on the given lane -- none
    // of it reads velocities, so it may run while the walk does; collisions_velocities() then
    // runs, on the main stream and after the kick, the velocity half of that sweep and both
    // halves of any further sweep (particle_system.py:453-470).
    void collisions_positions(const Lane& ln)
    {
        coll_totals[0] = coll_totals[1] = coll_totals[2] = 0.0;
        n_candidates = 0; n_contacts = 0; log_n = 0; solver_rounds = 0; solver_fixpoint_iters = 0;
        n_moved = 0; n_coincident = 0;
        settled_contacts = 0; settled_rounds = 0; settled_adj = nullptr;
        cond_pairs_ready = false;
        positions_settled = true;
        seam_n[0] = seam_n[1] = 0;
        slot_state_full = false;
        if (n == 0 || ns == 0 || cfg.solver_iterations <= 0) return;
        if (!cseg_ready || cseg.cap < (size_t)n) {
            cseg.ensure((size_t)n);
            be::launch(ln.s, (int64_t)cseg.cap, FillI32{(int64_t)cseg.cap, cseg.p, -1});
            cseg_ready = true;
        }
        if (ccap == 0) ensure_contact_capacity(ln, (size_t)n / 16 + 1024);
        ln.sort_tmp->ensure(be::sort_temp_words(2 * ccap));
        moved.ensure((size_t)n);
        moved_list.ensure((size_t)n);
        be::dev_memset(ln.s, moved.p, 0, (size_t)ns);
        sweep_positions(ln, 0);
        if (cfg.solver_iterations == 1) {
            moved_valid = true;
            if (sparse_conduction_possible()) conduction_sparse_pairs(ln);
        }
    }

    void collisions_velocities()
    {
        if (n == 0 || ns == 0 || cfg.solver_iterations <= 0) { moved_valid = cfg.solver_iterations > 0; return; }
        Lane ln = main_lane();
        if (deferred && cfg.solver_iterations > 1) {
            // sweeps after the first start from the states the sweep before left, which in this
            // mode are not in vel[] / temp[]: keep every slot's state in slot order
            slot_state.ensure((size_t)n);
            be::launch(st, ns, GatherSlotState{ns, cbody.p, vel.p, temp.p, slot_state.p});
            slot_state_full = true;
        }
        sweep_velocities(0);
        for (int it = 1; it < cfg.solver_iterations; ++it) {
            sweep_positions(ln, it);
            sweep_velocities(it);
        }
        // (the last sweep's totals stay on the device until the step's final readback)
        moved_valid = true;
        if (sparse_conduction_possible() && !cond_pairs_ready) conduction_sparse_pairs(ln);
    }

    void collisions()
    {
        collisions_positions(main_lane());
        collisions_velocities();
    }

    void set_slot_bits(uint32_t max_occupancy)
    {
        slot_bits = ceil_log2_u64((uint64_t)max_occupancy + 1);
        if (slot_bits < 1) slot_bits = 1;
    }
    int pair_key_bits() const
    {
        const int64_t cells = (int64_t)cfg.grid_w * cfg.grid_h;
        int bits = ceil_log2_u64((uint64_t)cells) + 4 + 2 * slot_bits;
        if (bits > 64) throw std::runtime_error("psim: grid cell population too large for the pair-key layout");
        return bits;
    }
    // How far beyond contact distance the first hypothesis of a sweep reaches (ContactDetect::margin).
    // In a sparse state (the benchmark's uniform clouds: 2 % of the particles in contact) the
    // hypothesis of the pairs that touch as they are is settled in three cheap iterations.  In a
    // clumped state every push-apart brings the next pair of a packed clump into contact, and
    // each of those is found one iteration late; with a margin they are in the hypothesis from
    // the start.  Switched on by the previous step's share of particles in contact.
    double margin_env = -2.0;
    bool margin_on = false;
    double sweep_margin = 0.0;         // the margin the current sweep's first hypothesis was made with
    double detect_margin()
    {
        if (margin_env < -1.5) {
            const char* e = getenv("PSIM_DETECT_MARGIN");
            margin_env = e ? atof(e) : -1.0;
        }
        if (margin_env >= 0.0) return margin_env;      // (experiments)
        return margin_on ? 1.0 : 0.0;
    }
    // ContactDetect may skip forward cells by distance to the shared border only while the
    // slot records hold the positions the grid was built from (the step's first sweep)
    bool detect_prune = true;

    void sweep_positions(const Lane& ln, int sweep)
    {
        if (sweep > 0) detect_prune = false;
        sweep_margin = detect_margin();
        GridView g = grid_view();
        const int slot_field_bits = ceil_log2_u64((uint64_t)ns + 1);
        uint32_t* const ctr = ln.ctr;
        DevBuf<uint32_t>& scan_scratch = *ln.scan_tmp;
        settled_contacts = 0; settled_rounds = 0; settled_adj = nullptr;
        // ---- sweep against the initial positions (count, scan, write)
        uint32_t n0 = 0, cand = 0;
        int key_bits = 64;
        {
            const uint32_t* dctr = ctr;
            if (sweep == 0 && predetected) {
                dctr = counters2.p;         // already computed on the side stream (this lane, or joined by the caller)
                predetected = false;
            } else {
                detect_count(ln.s, scan_scratch, ctr, ln.small_groups);
            }
            ContactDetect d;
            d.prune = detect_prune ? 1 : 0; d.r_max = max_radius; d.cell_size = (double)cfg.cell_size; d.margin = detect_margin();
            d.n = ns; d.k0 = 0; d.g = g; d.counts = pcount.p; d.offsets = poffs.p; d.first_pair = reinterpret_cast<uint32_t*>(first_key.p);
            d.coincident_counter = nullptr;
            uint32_t three[4];
            readback_on(ln, three, dctr, 16);
            if (sweep == 0 && sweep_margin == 0.0 && margin_env < 0.0 && (int64_t)three[0] * 10 > ns) {
                // a state this full of contacts on its first step (nothing to go by before): count again, with the margin
                margin_on = true;
                sweep_margin = detect_margin();
                d.margin = sweep_margin;
                detect_count(ln.s, scan_scratch, ctr, ln.small_groups);
                readback_on(ln, three, ctr, 16);
            }
            n0 = three[0]; cand = three[1];
            if (sweep == 0) n_coincident = three[3];
            set_slot_bits(three[2]);
            g.slot_bits = slot_bits;
            key_bits = pair_key_bits();
            if (n0 > 0) {
                ensure_contact_capacity(ln, n0);
                d.g = g;
                d.out_keys = h0[0].p; d.write_pass = 1; d.cand_counter = nullptr;
                be::launch(ln.s, ns, d);
            }
        }
        n_candidates += cand;
        if (n0 == 0) return;
        uint64_t* k0[2] = {h0[0].p, h0[1].p};
        int s0 = be::radix_sort_pairs<uint64_t>(ln.s, k0, nullptr, n0, 0, key_bits, ln.sort_tmp->p);
        const uint64_t* H0 = h0[s0].p;
        const uint64_t* H = H0;         // current hypothesis (sorted); the first one is the initial sweep's list
        int hn_which = 0;
        uint32_t nH = n0;
        uint64_t* adjs = nullptr;
        uint32_t round = 0, last_exec = 0, nNewSettled = 0;
        // Incremental re-test: per slot a digest of its versioned positions in the previous
        // iteration (zero at the start and at the end of every sweep), a flag where this
        // iteration's differ, and the list of the previous iteration's touched slots.
        slot_sig.ensure_zero(ln.s, (size_t)n); slot_changed.ensure_zero(ln.s, (size_t)n);
        if (incremental_dirty) {
            // a sweep that ended in an error may have left digests or flags behind
            be::dev_memset(ln.s, slot_sig.p, 0, slot_sig.cap * 8);
            be::dev_memset(ln.s, slot_changed.p, 0, slot_changed.cap);
        }
        incremental_dirty = true;
        be::dev_memset(ln.s, ctr + 9, 0, 8);
        int prev_list = 0;                 // prev_slots[prev_list] / ctr[9 + prev_list]: the previous iteration's touched slots
        int64_t cap_prev = 0;              // room that list needs (twice the previous hypothesis' size)
        // end of the sweep, however it ends: every digest and flag this sweep set is cleared
        auto wipe = [&](int64_t n_adj_now) {
            if (adjs && n_adj_now > 0)
                be::launch(ln.s, n_adj_now, ClearChangedAdj{n_adj_now, adjs, slot_changed.p, slot_sig.p, 1});
            for (int i = 0; i < 2; ++i)
                be::launch(ln.s, (int64_t)(2 * ccap), ClearChangedList{(int64_t)(2 * ccap), prev_slots[i].p, ctr + 9 + i,
                                                                      slot_changed.p, slot_sig.p, 1});
        };
        for (int iter = 0;; ++iter) {
            ++solver_fixpoint_iters;
            refresh_groups();
            if (iter > 100000) throw std::runtime_error("psim: collision solver did not converge");
            // ---- execute the hypothesis in order
            be::launch(ln.s, nH, ContactDecode{(int64_t)nH, g, H, hsi.p, hsj.p, adj[0].p, done.p});
            uint64_t* ka[2] = {adj[0].p, adj[1].p};
            // entries are generated in ascending contact order, so a stable sort on the slot
            // field alone leaves each slot's contacts in order
            int sa = be::radix_sort_pairs<uint64_t>(ln.s, ka, nullptr, 2ull * nH, 32, 32 + slot_field_bits, ln.sort_tmp->p);
            adjs = adj[sa].p;
            be::launch(ln.s, 2ll * nH, AdjLinks{2ll * nH, adjs, hsi.p, cseg.p, prevl.p, ctr + 7, tflag.p});
            scan_scratch.ensure_zero(ln.s, be::scan_temp_words(2ull * nH));
            be::exclusive_scan_u32(ln.s, tflag.p, tpos.p, 2ull * nH, scan_scratch.p, ctr + 8, small_groups_now(ln.small_groups));
            be::launch(ln.s, 2ll * nH, CompactFirst{2ll * nH, tflag.p, tpos.p, tlist.p});
            be::launch(ln.s, 2ll * nH, SaveTouched{2ll * nH, adjs, tlist.p, ctr + 8, prev_slots[prev_list ^ 1].p, ctr + 9 + (prev_list ^ 1)});
            // Dependency rounds.  A first batch is launched without asking how far it got (a
            // round with nothing ready is a no-op); whether contacts are still pending is read
            // back together with the re-test count below, and only then are more rounds run.
            round = 0;
            auto run_rounds = [&](int count) {
                for (int b = 0; b < count; ++b) {
                    ++round;
                    ContactSimulate sm;
                    sm.n = nH; sm.hsi = hsi.p; sm.hsj = hsj.p; sm.prev = prevl.p; sm.done = done.p; sm.round = round;
                    sm.out = outp.p; sm.geo = cgeo.p; sm.body = cbody.p;
                    sm.mass = mass.p; sm.G = cfg.gravity_constant; sm.soft = cfg.softening; sm.pos_correct = cfg.pos_correct;
                    // (a launch carries the chains inside each tile of contacts as far as they go)
                    be::launch_contact_rounds<true>(ln.s, sm, nH, prevl.p, done.p, round, round_tile_view());
                }
                solver_rounds += count;
            };
            // as many rounds up front as the last execution needed, plus one (a round with nothing
            // to do costs a few microseconds, a re-test that saw unfinished contacts is thrown away)
            run_rounds(rounds_hint);
            // ---- re-test what the hypothesis could have changed (count, scan, write)
            uint32_t nNew = 0;
            {
                dseg.ensure(2 * (size_t)nH); dsig.ensure(2 * (size_t)nH);
                auto displacement_and_changes = [&]() {
                    be::launch(ln.s, 2ll * nH, SegDisplacement{2ll * nH, adjs, cseg.p, hsi.p, outp.p, cbody.p, dseg.p, ctr + 7, ctr + 1,
                                                               H, slot_sig.p, dsig.p, slot_changed.p, (cfg.debug & 4) ? 1 : 0, cgeo.p,
                                                               (iter == 0 && sweep_margin > 0.0) ? 0.5 * sweep_margin - 1e-6 : -1.0});
                    if (cap_prev > 0)
                        be::launch(ln.s, cap_prev, MarkUntouched{cap_prev, prev_slots[prev_list].p, ctr + 9 + prev_list, cseg.p,
                                                                 slot_sig.p, slot_changed.p});
                };
                // A large hypothesis: make sure every contact has run before the (expensive) re-test
                // is launched -- one small readback instead of a re-test thrown away.
                const bool large = (cfg.debug & 2) || 2ll * nH + cap_prev > (1ll << 18);      // (debug bit 2: forced, for tests)
                if (large) {
                    for (;;) {
                        be::dev_memset(ln.s, ctr + 1, 0, 4);
                        be::launch(ln.s, nH, CountPending{(int64_t)nH, done.p, ctr + 1, ctr + 6});
                        if (read_u32(ln, ctr + 1) == 0) break;
                        if (round > 4102 + nH) throw std::runtime_error("psim: internal: contact schedule stalled");
                        run_rounds(6);
                    }
                }
                displacement_and_changes();
                // ... and only the changed slots get elements: their compacted list
                int64_t n_work = 0;
                if (large) {
                    const int64_t nw = 2ll * nH + cap_prev;
                    wflag.ensure((size_t)nw); wpos.ensure((size_t)nw); wlist.ensure((size_t)nw);
                    scan_scratch.ensure_zero(ln.s, be::scan_temp_words((uint64_t)nw));
                    be::launch(ln.s, nw, WorkFlags{2ll * nH, cap_prev, adjs, tlist.p, ctr + 8, prev_slots[prev_list].p, ctr + 9 + prev_list,
                                                   cseg.p, slot_changed.p, wflag.p});
                    be::exclusive_scan_u32(ln.s, wflag.p, wpos.p, (uint64_t)nw, scan_scratch.p, ctr + 11, small_groups_now(ln.small_groups));
                    be::launch(ln.s, nw, WorkList{2ll * nH, cap_prev, adjs, tlist.p, prev_slots[prev_list].p, wflag.p, wpos.p, wlist.p});
                    n_work = read_u32(ln, ctr + 11);
                }
                int64_t nt = large ? 10ll * n_work : 10ll * (2ll * nH + cap_prev), ntot = nt + nH;
                if (nt >= (1ll << 32)) throw std::runtime_error("psim: too many contacts for one sweep");
                pcount.ensure((size_t)ntot); poffs.ensure((size_t)ntot); first_key.ensure((size_t)kRetestKept * (size_t)nt + 1);
                scan_scratch.ensure_zero(ln.s, be::scan_temp_words((uint64_t)ntot));
                RetestPass rp;
                ContactRetest& r = rp.retest;
                r.n_adj = 2ll * nH; r.g = g; r.adj = adjs; r.cseg = cseg.p; r.hkeys = H; r.hsi = hsi.p; r.out = outp.p;
                r.dseg = dseg.p;
                r.prune = sweep == 0 ? 1 : 0; r.r_max = max_radius; r.cell_size = (double)cfg.cell_size; r.dmax_bits = ctr + 7;
                r.touched = tlist.p; r.n_touched = ctr + 8;
                r.changed = slot_changed.p; r.cap_prev = cap_prev > 0 ? cap_prev : 1; r.prev_slots = prev_slots[prev_list].p;
                r.n_prev = ctr + 9 + prev_list;
                r.work = large ? wlist.p : nullptr; r.n_work = n_work > 0 ? n_work : 1;
                r.counts = pcount.p; r.offsets = poffs.p; r.first_key = first_key.p; r.out_keys = nullptr;
                r.write_pass = 0;
                rp.carry = ContactCarry{(int64_t)nH, H, hsi.p, hsj.p, slot_changed.p, pcount.p + nt, poffs.p + nt, nullptr, 0, cgeo.p};
                rp.pending = CountPending{(int64_t)nH, done.p, ctr + 1, ctr + 6};
                rp.nt = nt; rp.n0 = nH;
                for (;;) {
                    be::launch(ln.s, ntot + nH, rp);
                    be::exclusive_scan_u32(ln.s, pcount.p, poffs.p, (uint64_t)ntot, scan_scratch.p, ctr, small_groups_now(ln.small_groups));
                    uint32_t two[7];
                    readback_on(ln, two, ctr, 28);
                    nNew = two[0];
                    if (two[1] == 0) {
                        last_exec = two[6];
                        // (chains tend to get a little deeper from one iteration to the next)
                        int want = (int)two[6] + 2 + (int)two[6] / 8;
                        rounds_hint = want < 2 ? 2 : (want > 4096 ? 4096 : want);
                        break;
                    }
                    // contacts were still waiting on predecessors: the re-test above saw incomplete
                    // versions; run further rounds and re-test again
                    if (round > 4102 + nH) throw std::runtime_error("psim: internal: contact schedule stalled");
                    run_rounds(6);
                    displacement_and_changes();
                }
                if (nNew > ccap) {
                    // growing reallocates every contact buffer: restart the whole sweep
                    be::launch(ln.s, 2ll * nH, ResetSeg{2ll * nH, adjs, cseg.p});
                    wipe(2ll * nH);
                    incremental_dirty = false;
                    be::sync(ln.s);
                    ensure_contact_capacity(ln, nNew);
                    n_candidates -= cand;
                    sweep_positions(ln, sweep);
                    return;
                }
                if (nNew > 0) {
                    r.out_keys = hn[0].p; r.write_pass = 1;
                    rp.carry.out_keys = hn[0].p; rp.carry.write_pass = 1;
                    be::launch(ln.s, ntot, rp);
                }
            }
            // Settled when H' is a subset of H: the pairs of H that are not in H' did not touch
            // when their turn came and changed nothing, so executing H' alone passes through the
            // very same states -- H' is its own re-test, and this execution (labels, positions,
            // normals) is already H's and its both.  The pairs that did not touch stay in the
            // list as such (normal 0, 0): the velocity half passes them by, the log, the counts,
            // the seam records and the list of touched slots leave them out.  Both lists are
            // duplicate-free, so "every new key is found in the (sorted) hypothesis" is the subset
            // test; the new list is only sorted if it is needed.
            bool same = nNew > 0 && nNew <= nH;
            if (same) {
                be::dev_memset(ln.s, ctr, 0, 4);
                be::launch(ln.s, nNew, AllFound{(int64_t)nNew, hn[0].p, H, (int64_t)nH, ctr});
                same = read_u32(ln, ctr) == 0;
            }
            if (same) { nNewSettled = nNew; break; }
            uint64_t* kn[2] = {hn[0].p, hn[1].p};
            hn_which = be::radix_sort_pairs<uint64_t>(ln.s, kn, nullptr, nNew, 0, key_bits, ln.sort_tmp->p);
            // this iteration's digests become the ones the next compares with; flags and segment starts cleared
            be::launch(ln.s, 2ll * nH, EndIteration{2ll * nH, adjs, dsig.p, cseg.p, slot_sig.p, slot_changed.p});
            if (cap_prev > 0)
                be::launch(ln.s, cap_prev, ClearChangedList{cap_prev, prev_slots[prev_list].p, ctr + 9 + prev_list,
                                                            slot_changed.p, slot_sig.p, 0});
            prev_list ^= 1;                 // (SaveTouched filled the other list with this iteration's touched slots)
            cap_prev = 2ll * nH;
            hc[0].swap(hn[hn_which]);       // the sorted new list becomes the hypothesis (same capacity)
            H = hc[0].p;
            nH = nNew;
            if (nH == 0) { adjs = nullptr; wipe(0); incremental_dirty = false; return; }
        }
        // ---- settled: positions written back, list of slots touched for the first time
        wipe(2ll * nH);
        incremental_dirty = false;
        pcount.ensure(2 * (size_t)nH); poffs.ensure(2 * (size_t)nH);
        ContactFinalizePositions f;
        f.n_adj = 2ll * nH; f.adj = adjs; f.cseg = cseg.p; f.hsi = hsi.p; f.out = outp.p; f.body = cbody.p;
        f.pos = deferred ? nullptr : pos.p; f.moved = moved.p; f.newly = pcount.p; f.geo = cgeo.p;
        be::launch(ln.s, 2ll * nH, f);
        be::exclusive_scan_u32(ln.s, pcount.p, poffs.p, 2ull * nH, scan_scratch.p, ctr + 5, small_groups_now(ln.small_groups));
        be::launch(ln.s, 2ll * nH, WriteMoved{2ll * nH, adjs, pcount.p, poffs.p, moved_list.p + n_moved});
        // (ContactFinalizePositions has already marked every touched slot clean again)
        n_moved += read_u32(ln, ctr + 5);
        // (H' is a subset of H and holds exactly the pairs of H that touched)
        settled_contacts = nH; settled_touching = nNewSettled; settled_rounds = last_exec; settled_adj = adjs;
    }

    // velocity half of the sweep whose position half ran last (main stream)
    be::RoundTiles round_tile_view() const
    {
        const size_t t = round_tiles.cap / 3;
        return be::RoundTiles{round_tiles.p, round_tiles.p + t, round_tiles.p + 2 * t};
    }

    void sweep_velocities(int sweep)
    {
        const uint32_t nH = settled_contacts;
        if (nH == 0) return;
        for (uint32_t r = 1; r <= settled_rounds; ++r) {
            ContactVelocity v;
            v.n = nH; v.hsi = hsi.p; v.hsj = hsj.p; v.prev = prevl.p; v.done = done.p; v.round = r; v.out = outv.p;
            v.geo = cgeo.p; v.vals = cvals.p; v.body = cbody.p; v.vel = vel.p; v.temp = temp.p; v.mass = mass.p;
            v.restitution = cfg.restitution;
            v.slot_state = slot_state_full ? slot_state.p : nullptr;
            be::launch_contact_rounds<false>(st, v, nH, prevl.p, done.p, r, round_tile_view());
        }
        // ---- totals, log, write-back
        size_t log_cap = 0;
        if (cfg.debug) {
            log_cap = (size_t)log_n + nH;
            if (log_ij.cap < 2 * log_cap || log_vals.cap < 3 * log_cap) {
                DevBuf<int32_t> nij; DevBuf<double> nv;
                nij.ensure(2 * log_cap); nv.ensure(3 * log_cap);
                if (log_n) {
                    be::copy_d2d(st, nij.p, log_ij.p, (size_t)log_n * 8);
                    be::copy_d2d(st, nv.p, log_vals.p, (size_t)log_n * 24);
                }
                be::sync(st);
                log_ij.swap(nij); log_vals.swap(nv);
            }
        }
        ContactLog cl;
        cl.n = nH; cl.hsi = hsi.p; cl.hsj = hsj.p; cl.body = cbody.p; cl.vals = cvals.p;
        cl.log_ij = cfg.debug ? log_ij.p : nullptr; cl.log_vals = cfg.debug ? log_vals.p : nullptr;
        cl.log_base = log_n; cl.log_cap = (int64_t)log_cap;
        cl.sum_pe = sum_pe.p; cl.sum_ke = sum_ke.p; cl.sum_heat = sum_heat.p;
        cl.own_lo = deferred ? band.own_lo : 0; cl.own_hi = deferred ? band.own_hi : cfg.grid_h;
        cl.owned_count = deferred ? step_words.p + 8 : nullptr;
        cl.geo = cgeo.p;
        if (deferred) be::dev_memset(st, step_words.p + 8, 0, 16);
        be::launch(st, nH, cl);
        const bool last_sweep = sweep + 1 >= cfg.solver_iterations;
        double* totals_dev = last_sweep ? reinterpret_cast<double*>(step_words.p) : red_out.p;
        be::reduce3_sum_f64(st, sum_pe.p, sum_ke.p, sum_heat.p, nH, red_tmp.p, totals_dev);
        ContactFinalizeVelocities f;
        f.n_adj = 2ll * nH; f.adj = settled_adj; f.hsi = hsi.p; f.out = outv.p; f.body = cbody.p; f.vel = vel.p;
        f.temp = temp.p;
        f.slot_out = nullptr;
        if (deferred) { slot_state.ensure((size_t)n); f.slot_out = slot_state.p; }
        be::launch(st, 2ll * nH, f);
        if (deferred) {
            // the contacts along the band's two seams, for the neighbours to compare with theirs
            // (appended sweep by sweep), and the number of contacts of the owned rows
            const int32_t seams[2] = {band.own_lo, band.own_hi};
            bool any = false;
            pcount.ensure(2 * (size_t)nH); poffs.ensure(2 * (size_t)nH);
            scan_tmp.ensure_zero(st, be::scan_temp_words(nH));
            for (int k = 0; k < 2; ++k) {
                if (seams[k] <= 0 || seams[k] >= cfg.grid_h) continue;       // the domain's edge: no neighbour
                any = true;
                be::launch(st, nH, SeamFlag{(int64_t)nH, hsi.p, hsj.p, cbody.p, seams[k], pcount.p + (size_t)k * nH, cgeo.p});
                be::exclusive_scan_u32(st, pcount.p + (size_t)k * nH, poffs.p + (size_t)k * nH, nH, scan_tmp.p, step_words.p + 9 + k);
            }
            uint32_t w[3];
            readback(w, step_words.p + 8, 12);
            n_contacts += w[0];
            for (int k = 0; k < 2 && any; ++k) {
                if (seams[k] <= 0 || seams[k] >= cfg.grid_h || w[1 + k] == 0) continue;
                const size_t have = (size_t)seam_n[k], add = w[1 + k];
                if (seam_rec[k].cap < (have + add) * kExportSeamWords) {
                    DevBuf<uint64_t> grown;
                    grown.ensure((have + add) * kExportSeamWords * 2);
                    if (have) be::copy_d2d(st, grown.p, seam_rec[k].p, have * kExportSeamWords * 8);
                    be::sync(st);
                    seam_rec[k].swap(grown);
                }
                be::launch(st, nH, SeamWrite{(int64_t)nH, hsi.p, hsj.p, pcount.p + (size_t)k * nH, poffs.p + (size_t)k * nH, cbody.p,
                                             outp.p, outv.p, seam_rec[k].p + have * kExportSeamWords});
                seam_n[k] += add;
            }
        } else {
            n_contacts += settled_touching;
        }
        if (last_sweep) {
            totals_pending = true;       // fetched with the explosion count (walls_and_flags) or by the export
        } else {
            double t[3];
            readback(t, red_out.p, 24);
            coll_totals[0] += t[0]; coll_totals[1] += t[1]; coll_totals[2] += t[2];
        }
        if (cfg.debug) log_n += nH;
        settled_contacts = 0;
    }

    // particle_system.py:476-498 + :555-557 as a sweep over every slot of the (whole) grid
    void conduction_and_cooling()
    {
        bool cond = (cfg.stage_mask & kStageConduction) != 0;
        bool cool = (cfg.stage_mask & kStageCooling) != 0 && cfg.damping > 0;
        if (n == 0 || (!cond && !cool)) return;
        double factor = 1.0 - cfg.damping;
        if (!cond) {
            be::launch(st, n, Cooling{n, temp.p, factor});
            return;
        }
        if (band_partial || ns != n) throw std::runtime_error("psim: internal: full conduction sweep on a partial grid");
        size_t c = (size_t)n;
        ctemp.ensure(c);
        GridView g = grid_view();
        be::launch(st, n, GatherThermal{n, cbody.p, temp.p, ctemp.p});
        if (cfg.debug) delta_dbg.ensure(c);
        Conduction k;
        k.n = n; k.g = g; k.ctemp = ctemp.p; k.mass = mass.p;
        k.heat_coeff = cfg.heat_coeff; k.cool_factor = factor; k.do_conduct = 1; k.do_cool = cool ? 1 : 0;
        k.temp = temp.p; k.delta_out = cfg.debug ? delta_dbg.p : nullptr;
        be::launch(st, n, k);
    }

    // The touching pairs all involve a slot the collision stage touched (see
    // ConductionPairs): possible when that stage ran this step and met no pair at
    // squared distance exactly zero.
    bool sparse_conduction_possible() const
    {
        return (cfg.stage_mask & kStageConduction) && (cfg.stage_mask & kStageCollisions) && moved_valid &&
               n_coincident == 0 && n > 0;
    }

    // Conduction around the touched slots.  The pair list depends on positions only and is
    // built (on the given lane) as soon as the last sweep's positions are settled;
    // conduction_sparse_apply() then fills cdelta (index space) from the post-collision
    // temperatures, and the caller applies T = (T + delta) * cooling in its next streaming
    // pass (walls_and_flags).
    void conduction_sparse_pairs(const Lane& ln)
    {
        size_t c = (size_t)n;
        GridView g = grid_view();
        cdelta.ensure(c);
        be::dev_memset(ln.s, cdelta.p, 0, c * 8);
        cond_ne = 0;
        cond_pairs_ready = true;
        if (n_moved == 0) return;
        int64_t nt = 10ll * n_moved;
        pcount.ensure((size_t)nt); poffs.ensure((size_t)nt);
        ln.scan_tmp->ensure_zero(ln.s, be::scan_temp_words((uint64_t)nt));
        ConductionPairs k;
        k.n_moved = n_moved; k.g = g; k.list = moved_list.p; k.moved = moved.p; k.counts = pcount.p; k.offsets = poffs.p;
        k.entries = nullptr; k.write_pass = 0;
        be::launch(ln.s, nt, k);
        be::exclusive_scan_u32(ln.s, pcount.p, poffs.p, (uint64_t)nt, ln.scan_tmp->p, ln.ctr, small_groups_now(ln.small_groups));
        uint32_t ne = read_u32(ln, ln.ctr);
        if (ne == 0) return;
        cond_entries[0].ensure(ne); cond_entries[1].ensure(ne);
        ln.sort_tmp->ensure(be::sort_temp_words(ne));
        k.entries = cond_entries[0].p; k.write_pass = 1;
        be::launch(ln.s, nt, k);
        uint64_t* ke[2] = {cond_entries[0].p, cond_entries[1].p};
        int bits = 32 + ceil_log2_u64((uint64_t)ns + 1);
        // targets in ascending slot order; a stable sort on the target field alone keeps the
        // sources of one target in the (fixed) order the pairs were written
        cond_sorted = be::radix_sort_pairs<uint64_t>(ln.s, ke, nullptr, ne, 32, bits, ln.sort_tmp->p);
        cond_ne = ne;
    }
    void conduction_sparse_apply()
    {
        if (cond_ne > 0) {
            ConductionApply a;
            a.n_entries = (int64_t)cond_ne; a.entries = cond_entries[cond_sorted].p; a.body = cbody.p; a.temp = temp.p;
            a.mass = mass.p; a.heat_coeff = cfg.heat_coeff; a.delta = cdelta.p;
            a.slot_state = nullptr; a.moved = nullptr; a.slot_delta = nullptr;
            if (deferred) {
                // (the touched slots' temperatures are in slot_state, the sums go to slot_delta for the export)
                slot_delta.ensure(cond_ne);
                a.slot_state = slot_state.p; a.moved = moved.p; a.slot_delta = slot_delta.p;
            }
            be::launch(st, cond_ne, a);
        }
        if (deferred) return;
        if (cfg.debug) {
            delta_dbg.ensure((size_t)n);
            be::copy_d2d(st, delta_dbg.p, cdelta.p, (size_t)n * 8);
        }
    }

    // particle_system.py:500-516 and the explosion test of :658
    uint32_t walls_and_flags(bool apply_delta = false)
    {
        bool w = (cfg.stage_mask & kStageWalls) != 0, f = (cfg.stage_mask & kStageExplosions) != 0;
        auto fetch = [&](bool want_count) -> uint32_t {
            // one synchronisation for the collision totals, the explosion count and the sorts' fault word
            if (!want_count && !totals_pending) return 0;
            struct { double t[3]; uint32_t count; } back;
            readback(&back, step_words.p, 28, true);
            if (totals_pending) {
                coll_totals[0] += back.t[0]; coll_totals[1] += back.t[1]; coll_totals[2] += back.t[2];
                totals_pending = false;
            }
            return want_count ? back.count : 0;
        };
        if (n == 0 || (!w && !f && !apply_delta)) return fetch(false);
        boom.ensure((size_t)n);
        be::dev_memset(st, step_words.p + 6, 0, 4);
        WallsFlag k;
        k.n = n; k.pos = pos.p; k.vel = vel.p; k.radii = radii.p; k.temp = temp.p; k.W = cfg.bounds_w;
        k.H = cfg.bounds_h; k.explode_temp = cfg.explode_temp; k.do_walls = w; k.do_flag = f; k.boom = boom.p;
        k.boom_count = step_words.p + 6;
        k.delta = apply_delta ? cdelta.p : nullptr;
        k.cool_factor = 1.0 - cfg.damping;
        k.do_cool = ((cfg.stage_mask & kStageCooling) != 0 && cfg.damping > 0) ? 1 : 0;
        be::launch(st, n, k);
        return fetch(f);
    }

    // particle_system.py:651-730
    void explosions(uint32_t count)
    {
        size_t c = (size_t)n;
        GridView g = grid_view();
        tmp32.ensure(c); tmp32b.ensure(c); scan_tmp.ensure_zero(st, be::scan_temp_words(c));
        ex_list.ensure(count); ex_count.ensure(count); ex_energy.ensure(count); ex_total.ensure(count);
        be::launch(st, n, FlagToU32{n, boom.p, tmp32.p, 0});
        be::exclusive_scan_u32(st, tmp32.p, tmp32b.p, (uint64_t)n, scan_tmp.p, nullptr);
        be::launch(st, n, ExploderList{n, boom.p, tmp32b.p, ex_list.p});
        uint32_t ne = 0;
        {
            ExplosionGather k;
            ex_cell_w.ensure(9 * (size_t)count); ex_cell_count.ensure(9 * (size_t)count); ex_cell_at.ensure(9 * (size_t)count);
            k.n_ex = count; k.list = ex_list.p; k.g = g; k.cell_size = (double)cfg.cell_size; k.pos = pos.p;
            k.boom = boom.p; k.cell_w = ex_cell_w.p; k.cell_count = ex_cell_count.p; k.cell_at = ex_cell_at.p;
            ex_offs.ensure(count);
            k.offsets = ex_offs.p; k.entries = nullptr; k.write_pass = 0;
            be::launch(st, 9ll * count, k);
            be::launch(st, count, ExplosionCombine{(int64_t)count, ex_list.p, mass.p, temp.p, cfg.explosion_eff, ex_cell_w.p,
                                                   ex_cell_count.p, ex_cell_at.p, ex_energy.p, ex_total.p, ex_count.p});
            be::exclusive_scan_u32(st, ex_count.p, ex_offs.p, (uint64_t)count, scan_tmp.p, counters.p);
            ne = read_u32(counters.p);
            if (ne > 0) {
                if (ne > ecap) { ecap = (size_t)ne + ne / 4 + 1024; ex_entries[0].ensure(ecap); ex_entries[1].ensure(ecap); }
                k.entries = ex_entries[0].p; k.write_pass = 1;
                be::launch(st, 9ll * count, k);
            }
        }
        if (ne > 0) {
            sort_tmp.ensure(be::sort_temp_words(ne));
            uint64_t* ke[2] = {ex_entries[0].p, ex_entries[1].p};
            // written in ascending exploder rank: a stable sort on the neighbour field suffices
            int bits = 32 + ceil_log2_u64((uint64_t)n + 1);
            int se = be::radix_sort_pairs<uint64_t>(st, ke, nullptr, ne, 32, bits, sort_tmp.p);
            be::launch(st, ne, ExplosionApply{(int64_t)ne, ex_entries[se].p, ex_list.p, pos.p, mass.p, ex_energy.p,
                                              ex_total.p, vel.p});
        }
        // events for the host-side log (particle_system.py:683-689)
        // (through page-locked memory: three copies in flight, one wait)
        const size_t stage_bytes = (size_t)count * 16 + 64;
        if (stage_bytes > ex_stage_cap) {
            be::host_free_pinned(ex_stage);
            ex_stage = nullptr; ex_stage_cap = 0;
            ex_stage = be::host_alloc_pinned(stage_bytes + stage_bytes / 2);
            ex_stage_cap = stage_bytes + stage_bytes / 2;
        }
        double* he = reinterpret_cast<double*>(ex_stage);
        uint32_t* hl = reinterpret_cast<uint32_t*>(he + count);
        uint32_t* hc2 = hl + count;
        be::copy_d2h(st, hl, ex_list.p, (size_t)count * 4);
        be::copy_d2h(st, hc2, ex_count.p, (size_t)count * 4);
        be::copy_d2h(st, he, ex_energy.p, (size_t)count * 8);
        be::sync(st);
        last_exploded.resize(count);
        for (uint32_t i = 0; i < count; ++i) {
            events.push_back(ExplosionEvent{(int32_t)hl[i], (int32_t)hc2[i], he[i]});
            last_exploded[i] = (int32_t)hl[i];
        }
        // compaction
        pos_b.ensure(c); vel_b.ensure(c); acc_b.ensure(c); mass_b.ensure(c); inv_mass_b.ensure(c); temp_b.ensure(c);
        radii_b.ensure(c);
        be::launch(st, n, FlagToU32{n, boom.p, tmp32.p, 1});
        be::exclusive_scan_u32(st, tmp32.p, tmp32b.p, (uint64_t)n, scan_tmp.p, nullptr);
        CompactState k;
        k.n = n; k.boom = boom.p; k.new_index = tmp32b.p; k.pos = pos.p; k.vel = vel.p; k.acc = acc.p; k.mass = mass.p;
        k.inv_mass = inv_mass.p; k.temp = temp.p; k.radii = radii.p; k.pos_o = pos_b.p; k.vel_o = vel_b.p;
        k.acc_o = acc_b.p; k.mass_o = mass_b.p; k.inv_mass_o = inv_mass_b.p; k.temp_o = temp_b.p; k.radii_o = radii_b.p;
        be::launch(st, n, k);
        pos.swap(pos_b); vel.swap(vel_b); acc.swap(acc_b); mass.swap(mass_b); inv_mass.swap(inv_mass_b);
        temp.swap(temp_b); radii.swap(radii_b);
        acc2.ensure(c);
        n -= count;
        n_exploded = count;
    }

    // -------------------------------------------------------------- entry points
    // particle_system.py:409-437 for a freshly built tree (as main.py:151 uses it)
    void prime_gravity()
    {
        keys_and_drift(false);
        build_tree();
        gravity(acc.p);
        be::sync(st);
    }

    // Grid build and the position half of the collision stage on the side stream.  They are
    // released when the tree is built, so that they share the device with the walk
    // (FP64-bound) and not with the tree build (memory-bound like themselves).  Joined by
    // join_side_work().
    bool side_running = false;
    // The side stream's head start for the walk: only when the walk fills the slots it is allowed
    // (a walk of fewer chunks than that leaves room anyway, and 40 us is a tenth of a
    // 2000-particle step)
    void side_delay()
    {
        const int64_t chunks = (n + 127) / 128, resident = walk_resident_waves();
        if (resident > 0 && chunks >= resident) be::delay(st2, 40);
    }
    // The side work may start where the tree is built (timer mark 3); the walk and the kick are
    // launched first, and only then is the side work submitted: its launches would
    // otherwise sit between the tree build and the walk on the host's submission path, and
    // the position half waits for small readbacks.
    void start_side_work()
    {
        be::stream_wait(st2, ev_release());
        // (lets the walk, released by the same event, take its wave slots first.  Releasing the
        // side work earlier instead -- in front of the tree's last kernel, which is all latency --
        // was measured: whatever side kernel is running when the walk arrives keeps the walk's
        // persistent waves from being placed, and the step got 4 % slower)
        side_delay();
        build_grid(st2, sort_tmp2);
        if (cfg.stage_mask & kStageCollisions) {
            detect_count(st2, scan_tmp2, counters2.p, true);
            predetected = true;
        }
        be::event_record(ev_side, st2);
        side_running = true;
    }
    // position half of the collision stage, on the side stream when there is one
    void settle_positions(bool after_join = false)
    {
        if (!(cfg.stage_mask & kStageCollisions) || positions_settled) return;
        if (side_running && !positions_under_walk() && !after_join) return;   // the caller joins first
        if (side_running) {
            collisions_positions(side_lane());
            be::event_record(ev_side, st2);
        } else {
            collisions_positions(main_lane());
        }
    }
    void join_side_work()
    {
        if (side_running) {
            // The host has nothing to submit before the walk ends.  If the side work ends first
            // (the usual case) the main stream needs no wait packet for it, which otherwise
            // sits between the kick and the velocity half for some 25 us.
            while (walk_in_flight && !be::event_done(ev_side) && !be::event_done(ev_walk())) be::cpu_relax();
            if (!be::event_done(ev_side)) be::stream_wait(st, ev_side);
        }
        side_running = false;
        refresh_groups();
    }
    int64_t updates_in_call = 0;       // particles alive after each step, summed over the steps of the last psim_step call
    void begin_step()
    {
        timer.next_step();
        // (n_contacts still holds the previous step's count)
        if (n > 0) {
            if (n_contacts * 10 > n) margin_on = true;
            else if (n_contacts * 20 < n) margin_on = false;
        }
        n_exploded = 0;
        last_exploded.clear();
        moved_valid = false;
        positions_settled = false;
        cond_pairs_ready = false;
        cond_ne = 0;
        totals_pending = false;
        walk_in_flight = false;
        detect_prune = true;
        deferred = false;
        applied_deltas = false;
        be::wide_groups() = true;
    }
    void no_collisions()
    {
        coll_totals[0] = coll_totals[1] = coll_totals[2] = 0.0;
        n_candidates = n_contacts = 0; log_n = 0;
    }

    // particle_system.py:518-563
    void step()
    {
        begin_step();
        timer.mark(st, 0);
        keys_and_drift(true);
        timer.mark(st, 1);
        // The grid build and the position half of the collision stage depend only on the
        // drifted positions: they run on a side stream underneath the FP64-bound walk and
        // are joined before the velocity half.
        const bool side = cfg.overlap != 0 && n > 0;
        if (!side) { build_grid(); timer.mark(st, 2); }
        else timer.same_as(2, 1);
        build_tree();
        timer.mark(st, 3);             // (tree built: also what releases the side stream)
        gravity(acc2.p);
        timer.mark(st, 4);             // (also the walk's end, as the host and the side lane ask for it)
        be::launch(st, n, Kick{n, vel.p, acc.p, acc2.p});      // (runs when the walk has ended)
        walk_in_flight = true;
        be::narrow_stream() = st2;
        be::wide_groups() = false;
        acc.swap(acc2);
        if (side) start_side_work();
        settle_positions();
        join_side_work();
        settle_positions(true);
        timer.mark(st, 5);
        if (cfg.stage_mask & kStageCollisions) collisions_velocities();
        else no_collisions();
        timer.mark(st, 6);
        bool sparse = sparse_conduction_possible();
        if (sparse) conduction_sparse_apply(); else conduction_and_cooling();
        timer.mark(st, 7);
        uint32_t nb = walls_and_flags(sparse);
        if (nb > 0) explosions(nb);
        timer.mark(st, 8);
        updates_in_call += n;
    }

    // ---- The same step cut at its exchange points, for one simulation split over several
    // GPUs.  Every rank holds the full state and builds the same tree; the walk runs on a
    // rank's share of the sorted particles, the collision stage (and conduction with it) on
    // the band of grid rows the rank owns plus halo rows, and the caller exchanges the
    // results between the stages (accelerations; the bands' export blocks).
    void stage_build()
    {
        begin_step();
        sharded_step = true;
        deferred = true;
        timer.mark(st, 0);
        keys_and_drift(true);
        timer.mark(st, 1);
        timer.same_as(2, 1);
        build_tree();
        timer.mark(st, 3);
    }
    void stage_gravity(int64_t lo, int64_t hi, vec2* acc_sorted)
    {
        if (lo < 0 || hi > n || lo > hi) throw std::invalid_argument("psim: bad sorted range");
        // (mark 3, recorded when stage_build finished the tree, releases the side stream)
        gravity_range(lo, hi, acc_sorted, true);
        timer.mark(st, 4);
        walk_in_flight = hi > lo;
        be::narrow_stream() = st2;
        be::wide_groups() = !walk_in_flight;
    }
    // The half of the collision stage that needs no velocities, for the band of grid rows
    // [row_lo, row_hi) of which [own_lo, own_hi) are this rank's own: grid of the band,
    // contact detection, contact list and pushed-apart positions.  While a walk submitted by
    // stage_gravity is in flight it runs on the side stream underneath it (it waits for small
    // readbacks of that stream, so a caller with an exchange of accelerations to do starts
    // that exchange first).
    void stage_slab_positions(const Band& b)
    {
        if (!sharded_step) throw std::runtime_error("psim: stage_slab_positions outside a staged step");
        deferred = true;
        positions_settled = false;
        moved_valid = false;
        predetected = false;
        detect_prune = true;
        n_moved = 0; n_coincident = 0; cond_ne = 0; cond_pairs_ready = false;
        seam_n[0] = seam_n[1] = 0;
        own_slots = 0;
        band = b;
        if (b.row_lo < 0 || b.row_hi > cfg.grid_h || b.row_lo > b.row_hi || b.own_lo < b.row_lo || b.own_hi > b.row_hi || b.own_lo > b.own_hi)
            throw std::invalid_argument("psim: bad band of grid rows");
        if (!(cfg.stage_mask & kStageCollisions) || n == 0 || b.row_lo == b.row_hi) {
            ns = 0;
            band_partial = true;
            full_grid_valid = false;
            no_collisions();
            positions_settled = true;
            return;
        }
        refresh_groups();
        const bool side = cfg.overlap != 0 && walk_in_flight;
        Lane ln = side ? side_lane() : main_lane();
        if (side) {
            be::stream_wait(st2, ev_release());
            side_delay();            // (the walk's persistent waves take their slots first, see start_side_work)
            side_running = true;
        }
        build_grid_band(ln, b);
        collisions_positions(ln);
        if (side) be::event_record(ev_side, st2);
    }
    // second half-kick from the gathered accelerations (sorted order)
    void stage_kick(const vec2* acc_sorted)
    {
        if (!sharded_step) throw std::runtime_error("psim: stage_kick outside a staged step");
        be::launch(st, n, ScatterSorted{n, perm[sorted_key].p, walk_order(), acc_sorted, acc2.p});
        be::launch(st, n, Kick{n, vel.p, acc.p, acc2.p});
        acc.swap(acc2);
    }
    // Velocity half of the band's collision stage and its conduction sums, after stage_kick.
    // Everything stays in slot order; slab_export() writes the block the other ranks need.
    void stage_slab_finish()
    {
        if (!sharded_step || !deferred) throw std::runtime_error("psim: stage_slab_finish outside a staged step");
        join_side_work();
        timer.mark(st, 5);
        if ((cfg.stage_mask & kStageCollisions) && ns > 0) collisions_velocities();
        else moved_valid = (cfg.stage_mask & kStageCollisions) != 0;
        timer.mark(st, 6);
        if (sparse_conduction_possible()) conduction_sparse_apply();
    }
    // Writes the band's block: header, then the state of the touched slots of the owned rows,
    // the conduction sums of the targets in the owned rows, the two seams' contact records.
    // Returns the number of words the block needs; when that exceeds cap_words only the
    // header is written (marked incomplete) and the call can be repeated with more room.
    int64_t slab_export(uint64_t* out, int64_t cap_words)
    {
        if (!sharded_step || !deferred) throw std::runtime_error("psim: slab_export outside a staged step");
        if (!out || cap_words < kExportHeaderWords) throw std::invalid_argument("psim: export block smaller than its header");
        const bool with_deltas = sparse_conduction_possible() && cond_ne > 0;
        uint32_t n_state = 0, n_delta = 0;
        const size_t tmp_n = (size_t)(n_moved > cond_ne ? n_moved : cond_ne) + 1;
        bflag.ensure(2 * tmp_n); bat.ensure(2 * tmp_n);
        scan_tmp.ensure_zero(st, be::scan_temp_words(tmp_n));
        be::dev_memset(st, step_words.p + 12, 0, 16);
        if (n_moved > 0) {
            be::launch(st, n_moved, ExportStateFlag{(int64_t)n_moved, moved_list.p, cbody.p, band.own_lo, band.own_hi, bflag.p});
            be::exclusive_scan_u32(st, bflag.p, bat.p, n_moved, scan_tmp.p, step_words.p + 12);
        }
        if (with_deltas) {
            be::launch(st, cond_ne, ExportDeltaFlag{(int64_t)cond_ne, cond_entries[cond_sorted].p, cbody.p, band.own_lo, band.own_hi,
                                                    bflag.p + tmp_n});
            be::exclusive_scan_u32(st, bflag.p + tmp_n, bat.p + tmp_n, cond_ne, scan_tmp.p, step_words.p + 13);
        }
        if (ns > 0) {
            // slots of the owned rows: the band's cells are row-major, so they are one run of slots
            be::launch(st, 1, OwnedSlots{offsets.p, (int64_t)(band.own_lo - band.row_lo) * cfg.grid_w,
                                         (int64_t)(band.own_hi - band.row_lo) * cfg.grid_w, step_words.p + 14});
        }
        {
            struct { double t[3]; uint32_t pad[2]; uint32_t pad2[4]; uint32_t c[4]; } back;
            static_assert(sizeof(back) == 64, "layout of the step words");
            readback(&back, step_words.p, 64, true);
            if (totals_pending) {
                coll_totals[0] += back.t[0]; coll_totals[1] += back.t[1]; coll_totals[2] += back.t[2];
                totals_pending = false;
            }
            n_state = back.c[0]; n_delta = back.c[1]; own_slots = back.c[2];
        }
        const int64_t need = kExportHeaderWords + (int64_t)kExportStateWords * n_state + (int64_t)kExportDeltaWords * n_delta +
                             (int64_t)kExportSeamWords * (seam_n[0] + seam_n[1]);
        const bool fits = need <= cap_words;
        ExportHeader h;
        for (int i = 0; i < kExportHeaderWords; ++i) h.words[i] = 0;
        h.words[0] = n_state; h.words[1] = n_delta; h.words[2] = (uint64_t)seam_n[0]; h.words[3] = (uint64_t)seam_n[1];
        h.words[4] = (uint64_t)n_contacts; h.words[5] = (uint64_t)n_candidates; h.words[6] = n_coincident;
        h.words[7] = (uint64_t)own_slots;
        h.words[8] = f64_bits(coll_totals[0]); h.words[9] = f64_bits(coll_totals[1]); h.words[10] = f64_bits(coll_totals[2]);
        h.words[11] = (uint64_t)need; h.words[12] = (uint64_t)solver_fixpoint_iters; h.words[13] = (uint64_t)solver_rounds;
        h.words[14] = (uint64_t)(uint32_t)band.own_lo | ((uint64_t)(uint32_t)band.own_hi << 32);
        h.words[15] = fits ? 1u : 0u;
        h.words[16] = (uint64_t)n;
        h.out = out;
        be::launch(st, kExportHeaderWords, h);
        if (fits) {
            uint64_t* at = out + kExportHeaderWords;
            if (n_state > 0)
                be::launch(st, n_moved, ExportStateWrite{(int64_t)n_moved, moved_list.p, bflag.p, bat.p, cbody.p, slot_state.p, at});
            at += (size_t)kExportStateWords * n_state;
            if (n_delta > 0)
                be::launch(st, cond_ne, ExportDeltaWrite{(int64_t)cond_ne, cond_entries[cond_sorted].p, bflag.p + tmp_n, bat.p + tmp_n,
                                                         cbody.p, slot_delta.p, at});
            at += (size_t)kExportDeltaWords * n_delta;
            for (int k = 0; k < 2; ++k) {
                if (seam_n[k] > 0)
                    be::launch(st, seam_n[k] * kExportSeamWords, CopyWords{seam_n[k] * kExportSeamWords, seam_rec[k].p, at});
                at += (size_t)kExportSeamWords * (size_t)seam_n[k];
            }
        }
        be::sync(st);       // (the caller's exchange reads the block from another stream)
        return need;
    }
    // The ranks' blocks, side by side (`world` blocks of block_words words, rank order, device
    // memory).  Returns 0 when every pair of neighbouring ranks recorded the same contacts along
    // their seam: all blocks' results are then written to the state arrays.  1: some seam's
    // records differ -- nothing was applied, the caller runs stage_fallback().  2: some block
    // was too small (*needed = the largest need) -- export again with room and call again.
    int stage_apply(const uint64_t* blocks, int world, int64_t block_words, int64_t* own_slots_out, int64_t* needed)
    {
        if (!sharded_step || !deferred) throw std::runtime_error("psim: stage_apply outside a staged step");
        if (!blocks || world < 1 || world > 4096 || block_words < kExportHeaderWords) throw std::invalid_argument("psim: bad export blocks");
        std::vector<uint64_t> hd((size_t)world * kExportHeaderWords);
        hdr_dev.ensure(hd.size());
        be::launch(st, (int64_t)hd.size(), GatherHeaders{world, block_words, blocks, hdr_dev.p});
        be::copy_d2h(st, hd.data(), hdr_dev.p, hd.size() * 8);
        be::sync(st);
        auto H = [&](int r, int w) { return hd[(size_t)r * kExportHeaderWords + w]; };
        int64_t most = 0;
        bool complete = true;
        for (int r = 0; r < world; ++r) {
            if (H(r, 16) != (uint64_t)n) throw std::runtime_error("psim: export block of another step or particle count");
            if ((int64_t)H(r, 11) > most) most = (int64_t)H(r, 11);
            if (H(r, 15) != 1) complete = false;
            if (own_slots_out) own_slots_out[r] = (int64_t)H(r, 7);
        }
        if (needed) *needed = most;
        if (!complete) return 2;
        auto section = [&](int r, int which) {
            const uint64_t* p = blocks + (size_t)r * block_words + kExportHeaderWords;
            if (which > 0) p += (size_t)kExportStateWords * H(r, 0);
            if (which > 1) p += (size_t)kExportDeltaWords * H(r, 1);
            if (which > 2) p += (size_t)kExportSeamWords * H(r, 2);
            return p;
        };
        // ---- seams: the upper seam of a rank against the lower seam of the next rank that owns rows
        bool agree = true;
        be::dev_memset(st, step_words.p + 12, 0, 4);
        int prev = -1;
        for (int r = 0; r < world; ++r) {
            const int32_t lo = (int32_t)(uint32_t)H(r, 14), hi = (int32_t)(uint32_t)(H(r, 14) >> 32);
            if (hi <= lo) continue;          // owns no rows
            if (prev >= 0) {
                const int32_t phi = (int32_t)(uint32_t)(H(prev, 14) >> 32);
                if (phi != lo) throw std::invalid_argument("psim: the ranks' bands do not tile the grid rows");
                const uint64_t na = H(prev, 3), nb = H(r, 2);
                if (na != nb) agree = false;
                else if (na > 0)
                    be::launch(st, (int64_t)(na * kExportSeamWords), CompareKeys{(int64_t)(na * kExportSeamWords), section(prev, 3),
                                                                                 section(r, 2), step_words.p + 12});
            }
            prev = r;
        }
        if (agree) agree = read_u32(step_words.p + 12) == 0;
        if (!agree) { ++slab_fallbacks; return 1; }
        // ---- apply
        double tot[3] = {0, 0, 0};
        int64_t contacts = 0, candidates = 0, coincident = 0;
        cdelta.ensure((size_t)n);
        if (n > 0) be::dev_memset(st, cdelta.p, 0, (size_t)n * 8);
        for (int r = 0; r < world; ++r) {
            if (H(r, 0) > 0) be::launch(st, (int64_t)H(r, 0), ApplyState{(int64_t)H(r, 0), section(r, 0), pos.p, vel.p, temp.p});
            coincident += (int64_t)H(r, 6);
        }
        for (int r = 0; r < world && coincident == 0; ++r)
            if (H(r, 1) > 0) be::launch(st, (int64_t)H(r, 1), ApplyDelta{(int64_t)H(r, 1), section(r, 1), cdelta.p});
        for (int r = 0; r < world; ++r) {
            tot[0] += bits_f64(H(r, 8)); tot[1] += bits_f64(H(r, 9)); tot[2] += bits_f64(H(r, 10));
            contacts += (int64_t)H(r, 4); candidates += (int64_t)H(r, 5);
        }
        coll_totals[0] = tot[0]; coll_totals[1] = tot[1]; coll_totals[2] = tot[2];
        n_contacts = contacts; n_candidates = candidates; n_coincident = (uint32_t)(coincident > 0xffffffffll ? 0xffffffffll : coincident);
        moved_valid = (cfg.stage_mask & kStageCollisions) != 0;
        deferred = false;
        applied_deltas = true;
        return 0;
    }
    bool applied_deltas = false;       // cdelta holds the step's conduction sums, from the ranks' blocks
    // After a seam disagreement: the whole collision stage again on the whole grid, on every
    // rank alike (state arrays written directly, as in step()).
    void stage_fallback()
    {
        if (!sharded_step) throw std::runtime_error("psim: stage_fallback outside a staged step");
        deferred = false;
        applied_deltas = false;
        positions_settled = false;
        moved_valid = false;
        predetected = false;
        detect_prune = true;
        totals_pending = false;
        build_grid(st, sort_tmp);
        if (cfg.stage_mask & kStageCollisions) collisions();
        else no_collisions();
    }
    void stage_finish()
    {
        if (!sharded_step || deferred) throw std::runtime_error("psim: stage_finish before stage_apply / stage_fallback");
        const bool sparse = sparse_conduction_possible();
        if (sparse) {
            if (!applied_deltas) conduction_sparse_apply();
        } else {
            if (n > 0 && (cfg.stage_mask & kStageConduction)) ensure_full_grid();
            conduction_and_cooling();
        }
        timer.mark(st, 7);
        uint32_t nb = walls_and_flags(sparse);
        if (nb > 0) { ensure_full_grid(); explosions(nb); }
        timer.mark(st, 8);
        sharded_step = false;
        applied_deltas = false;
    }

    // particle_system.py:765-804 (+ total momentum)
    void energies(double* out5)
    {
        out5[0] = out5[1] = out5[2] = out5[3] = out5[4] = 0.0;
        if (n > 0) {
            for (int i = 0; i < 4; ++i) terms[i].ensure((size_t)n);
            be::launch(st, n, EnergyTerms{n, vel.p, mass.p, temp.p, terms[0].p, terms[1].p, terms[2].p, terms[3].p});
            for (int i = 0; i < 4; ++i) be::reduce_sum_f64(st, terms[i].p, (uint64_t)n, red_tmp.p, red_out.p + i);
        } else {
            be::dev_memset(st, red_out.p, 0, 32);
        }
        if (tree_valid && n_internal > 0) {
            be::launch(st, (int64_t)n_internal, TreePotential{(int64_t)n_internal, node_children(), pe_node.p,
                                                              cfg.gravity_constant, cfg.softening, root_tag()});
            be::reduce_sum_f64(st, pe_node.p, n_internal, red_tmp.p, red_out.p + 4);
        }
        else be::dev_memset(st, red_out.p + 4, 0, 8);
        double t[5];
        readback(t, red_out.p, 40);
        out5[0] = t[0]; out5[1] = t[4]; out5[2] = t[1]; out5[3] = t[2]; out5[4] = t[3];
    }
};

}  // namespace psim

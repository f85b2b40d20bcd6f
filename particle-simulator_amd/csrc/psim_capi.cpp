// C ABI of libpsim_hip.so (declared in include/psim.h).  Compiled as HIP for
// gfx950; the test-only host-loop build (tests/hostloop) compiles the same file
// with -DPSIM_HOSTLOOP and its own back end.
#if defined(PSIM_HOSTLOOP)
#include "backend_hostloop.h"
#else
#include "backend_hip.h"
#endif
#include "psim_sim.h"
#include "../../include/psim.h"

#include <algorithm>
#include <mutex>
#include <random>
#include <stdio.h>
#include <stdlib.h>

static_assert(sizeof(psim_config) == sizeof(psim::Config), "psim_config and psim::Config must match");
static_assert(sizeof(psim_explosion_event) == sizeof(psim::ExplosionEvent), "event layout");

namespace psim {

void Sim::gravity_range(int64_t lo, int64_t hi, vec2* out, bool sorted_out)
{
    if (hi <= lo) return;
    uint32_t total = (uint32_t)(n + n_internal);
    double theta2 = cfg.theta * cfg.theta;          // particle_system.py:88 theta * theta * dist_sq
#if !defined(PSIM_HOSTLOOP)
    if (cfg.walk_mode == 0 || cfg.walk_mode == 3 || cfg.walk_mode == 5) {
        // default: two walks per wave (3: with 64-bit element offsets forced, as on arrays of 4 GiB and more;
        // 5: with the compiler's two-group loop instead of the hand-scheduled one)
        be::launch_walk_multi<2>(st, lo, hi, sorted_out ? 1 : 0, (const NodeRec*)rec.p, total, (const vec2*)spos.p,
                                 (const double*)smass.p, (const uint32_t*)perm[sorted_key].p,
                                 walk_order(), out, walk_fold, (uint32_t)w2_cmp_bits, theta2,
                                 cfg.gravity_constant, cfg.softening, walk_ctr.p, walk_resident_waves(),
                                 cfg.walk_mode == 3, cfg.walk_mode == 5);
        return;
    }
    if (cfg.walk_mode == 2 || cfg.walk_mode == 4) {
        // one walk per wave (the kernel the visit / interaction counts are taken with)
        be::launch_walk<false>(st, lo, hi, sorted_out ? 1 : 0, (const NodeRec*)rec.p, total, (const vec2*)spos.p,
                               (const double*)smass.p, (const uint32_t*)perm[sorted_key].p, walk_order(),
                               out, walk_fold, (uint32_t)w2_cmp_bits, theta2, cfg.gravity_constant, cfg.softening,
                               (unsigned long long*)nullptr, cfg.walk_mode == 4);
        return;
    }
#endif
    GravityWalkSimple k;
    k.n = hi - lo; k.s0 = lo; k.sorted_out = sorted_out ? 1 : 0; k.order = walk_order(); k.rec = rec.p; k.total_elems = total; k.spos = spos.p;
    k.smass = smass.p; k.perm = perm[sorted_key].p; k.inv_mass = inv_mass.p; k.acc_out = out; k.w2tab = w2tab.p;
    k.root_tag = root_tag(); k.theta2 = theta2; k.g_const = cfg.gravity_constant; k.softening = cfg.softening; k.counters = nullptr;
    be::launch(st, hi - lo, k);
}

void Sim::gravity(vec2* out) { gravity_range(0, n, out, false); }

}  // namespace psim

struct psim_context {
    int device;
    psim::Sim* sim;
};

// ---- scratch of psim_pair_potential_energy_noalloc: one row sum per particle, then the
// reduction's partial sums and its result.  Device memory that exists from the moment the
// library is loaded on a GPU until the process ends (2 MiB + 8 KiB on every GPU used).
namespace {
constexpr int64_t kNoallocMaxParticles = 1 << 18;
constexpr size_t kNoallocWords = (size_t)kNoallocMaxParticles + psim::be::kReduceBlocks + 8;
#if defined(PSIM_HOSTLOOP)
double scratch[kNoallocWords];
#else
// In host code this name is only a stand-in for the device array: the array's address on a
// given GPU comes from hipGetSymbolAddress (each GPU holds its own copy at its own address).
static __device__ __align__(256) double scratch[kNoallocWords];
#endif
std::mutex g_scratch_lock;
double* g_scratch_ptr[64] = {};      // per device ordinal

// called with `device` current (psim_create: the library's set-up call for a device)
void resolve_scratch(int device)
{
    if (device < 0 || device >= 64) throw std::invalid_argument("psim: device ordinal out of range");
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (g_scratch_ptr[device]) return;
#if defined(PSIM_HOSTLOOP)
    g_scratch_ptr[device] = scratch;
#else
    void* p = nullptr;
    PSIM_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) throw std::runtime_error("psim: hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[device] = static_cast<double*>(p);
#endif
}
double* scratch_of(int device)
{
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    double* p = (device >= 0 && device < 64) ? g_scratch_ptr[device] : nullptr;
    if (!p) throw std::runtime_error("psim: scratch array not resolved for this device");
    return p;
}
}  // namespace

static thread_local std::string g_err;

static void use_device(int device)
{
#if !defined(PSIM_HOSTLOOP)
    PSIM_HIP(hipSetDevice(device));
#else
    (void)device;
#endif
}

#define PSIM_TRY(ctx_check, ...)                                    \
    try {                                                           \
        if (!(ctx_check)) { g_err = "psim: null handle or argument"; return PSIM_ERR_INVALID; } \
        __VA_ARGS__;                                                \
        return PSIM_OK;                                             \
    } catch (const std::invalid_argument& e) {                      \
        g_err = e.what();                                           \
        return PSIM_ERR_INVALID;                                    \
    } catch (const std::exception& e) {                             \
        g_err = e.what();                                           \
        return PSIM_ERR_RUNTIME;                                    \
    }

static psim::Config to_config(const psim_config* c)
{
    psim::Config k;
    memcpy(&k, c, sizeof(k));
    if (k.solver_iterations < 0) throw std::invalid_argument("psim: solver_iterations < 0");
    return k;
}

extern "C" {

const char* psim_last_error(void) { return g_err.c_str(); }

const char* psim_version(void)
{
#if defined(PSIM_HOSTLOOP)
    return "psim 0.1 (host-loop test build, not a product)";
#else
    return "psim 0.1 (HIP gfx950)";
#endif
}

int psim_create(const psim_config* cfg, int device_ordinal, psim_context** out)
{
    PSIM_TRY(cfg && out, {
        use_device(device_ordinal);
        resolve_scratch(device_ordinal);
        psim_context* c = new psim_context;
        c->device = device_ordinal;
        c->sim = nullptr;
        try {
            c->sim = new psim::Sim(to_config(cfg));
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
    })
}

int psim_destroy(psim_context* ctx)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        delete ctx->sim;
        delete ctx;
    })
}

int psim_set_config(psim_context* ctx, const psim_config* cfg)
{
    PSIM_TRY(ctx && cfg, {
        use_device(ctx->device);
        ctx->sim->set_config(to_config(cfg));
    })
}

int psim_upload(psim_context* ctx, int64_t n, const double* positions, const double* velocities,
                const double* accelerations, const double* masses, const double* temperatures, const int64_t* radii)
{
    PSIM_TRY(ctx && (n == 0 || (positions && masses && temperatures && radii)), {
        use_device(ctx->device);
        ctx->sim->upload(n, positions, velocities, accelerations, masses, temperatures, radii);
    })
}

int psim_download(psim_context* ctx, double* positions, double* velocities, double* accelerations, double* masses,
                  double* inverse_masses, double* temperatures, int64_t* radii)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        ctx->sim->download(positions, velocities, accelerations, masses, inverse_masses, temperatures, radii);
    })
}

int64_t psim_num_particles(psim_context* ctx) { return ctx ? ctx->sim->n : -1; }

int64_t psim_particle_updates(psim_context* ctx) { return ctx ? ctx->sim->updates_in_call : -1; }

int psim_prime_gravity(psim_context* ctx)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        ctx->sim->prime_gravity();
    })
}

int psim_step(psim_context* ctx, int64_t k_steps)
{
    PSIM_TRY(ctx && k_steps >= 0, {
        use_device(ctx->device);
        ctx->sim->timer.begin_call();
        ctx->sim->updates_in_call = 0;
        for (int64_t k = 0; k < k_steps; ++k) ctx->sim->step();
        psim::be::sync(ctx->sim->st);
    })
}

int psim_stage_build(psim_context* ctx)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        ctx->sim->timer.begin_call();
        ctx->sim->stage_build();
        psim::be::sync(ctx->sim->st);
    })
}

int psim_stage_gravity(psim_context* ctx, int64_t s_lo, int64_t s_hi, double* acc_sorted)
{
    // not waited for: the caller's exchange is ordered after the walk on the same stream
    PSIM_TRY(ctx && acc_sorted, {
        use_device(ctx->device);
        psim::be::require_device_memory(acc_sorted, "psim_stage_gravity: acc_sorted");
        ctx->sim->stage_gravity(s_lo, s_hi, reinterpret_cast<psim::vec2*>(acc_sorted));
    })
}

int psim_stage_slab_positions(psim_context* ctx, int32_t row_lo, int32_t row_hi, int32_t own_lo, int32_t own_hi)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        ctx->sim->stage_slab_positions(psim::Sim::Band{row_lo, row_hi, own_lo, own_hi});
    })
}

int psim_stage_kick(psim_context* ctx, const double* acc_sorted)
{
    PSIM_TRY(ctx && acc_sorted, {
        use_device(ctx->device);
        psim::be::require_device_memory(acc_sorted, "psim_stage_kick: acc_sorted");
        ctx->sim->stage_kick(reinterpret_cast<const psim::vec2*>(acc_sorted));
    })
}

int psim_stage_slab_finish(psim_context* ctx)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        ctx->sim->stage_slab_finish();
    })
}

int psim_stage_slab_export(psim_context* ctx, uint64_t* block, int64_t cap_words, int64_t* words_needed)
{
    PSIM_TRY(ctx && block && words_needed, {
        use_device(ctx->device);
        psim::be::require_device_memory(block, "psim_stage_slab_export: block");
        *words_needed = ctx->sim->slab_export(block, cap_words);
    })
}

int psim_stage_apply(psim_context* ctx, const uint64_t* blocks, int32_t world, int64_t block_words, int32_t* status,
                     int64_t* own_slots, int64_t* words_needed)
{
    PSIM_TRY(ctx && blocks && status, {
        use_device(ctx->device);
        psim::be::require_device_memory(blocks, "psim_stage_apply: blocks");
        *status = ctx->sim->stage_apply(blocks, world, block_words, own_slots, words_needed);
    })
}

int psim_stage_fallback(psim_context* ctx)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        ctx->sim->stage_fallback();
    })
}

int psim_stage_finish(psim_context* ctx)
{
    PSIM_TRY(ctx, {
        use_device(ctx->device);
        ctx->sim->stage_finish();
        psim::be::sync(ctx->sim->st);
    })
}

int psim_stage_stats(psim_context* ctx, int64_t out[4])
{
    PSIM_TRY(ctx && out, {
        out[0] = ctx->sim->slab_fallbacks;
        out[1] = ctx->sim->own_slots;
        out[2] = ctx->sim->ns;
        out[3] = psim::kExportHeaderWords;
    })
}

int psim_energies(psim_context* ctx, double out[5])
{
    PSIM_TRY(ctx && out, {
        use_device(ctx->device);
        ctx->sim->energies(out);
    })
}

int psim_collision_stats(psim_context* ctx, double out[3])
{
    PSIM_TRY(ctx && out, {
        out[0] = ctx->sim->coll_totals[0];
        out[1] = ctx->sim->coll_totals[1];
        out[2] = ctx->sim->coll_totals[2];
    })
}

int psim_counts(psim_context* ctx, int64_t out[8])
{
    PSIM_TRY(ctx && out, {
        psim::Sim* s = ctx->sim;
        out[0] = s->n;
        out[1] = s->tree_valid ? (s->n_internal ? 1 + 4 * (int64_t)s->n_internal : (s->tree_n || s->n ? 1 : 0)) : 0;
        out[2] = s->n_candidates;
        out[3] = s->n_contacts;
        out[4] = s->n_exploded;
        out[5] = s->n_internal;
        out[6] = s->solver_fixpoint_iters;
        out[7] = s->solver_rounds;
    })
}

int64_t psim_explosion_events(psim_context* ctx, psim_explosion_event* buf, int64_t cap)
{
    if (!ctx || (cap > 0 && !buf) || cap < 0) { g_err = "psim: null handle or argument"; return PSIM_ERR_INVALID; }
    auto& ev = ctx->sim->events;
    int64_t k = std::min<int64_t>(cap, (int64_t)ev.size());
    if (k > 0) memcpy(buf, ev.data(), (size_t)k * sizeof(psim_explosion_event));
    ev.erase(ev.begin(), ev.begin() + k);
    return k;
}

static void walk_counts_impl(psim_context* ctx, int64_t* out, int n_out)
{
    use_device(ctx->device);
    psim::Sim* s = ctx->sim;
    for (int i = 0; i < n_out; ++i) out[i] = 0;
    if (!s->tree_valid || s->walk_n == 0) throw std::runtime_error("psim: no tree built yet");
    // the tree describes the particles it was built from: after a step that removed particles
    // by explosion that is more than the current count (the removed ones are still in it)
    const int64_t tn = s->walk_n;
    uint32_t total = (uint32_t)(tn + s->n_internal);
    double theta2 = s->cfg.theta * s->cfg.theta;
    psim::DevBuf<psim::vec2> scratch;
    scratch.ensure((size_t)tn);
#if !defined(PSIM_HOSTLOOP)
    psim::DevBuf<unsigned long long> cnt;
    cnt.ensure(32);
    psim::be::dev_memset(s->st, cnt.p, 0, 256);
    psim::be::launch_walk<true>(s->st, (int64_t)0, tn, 0, (const psim::NodeRec*)s->rec.p, total,
                                (const psim::vec2*)s->spos.p, (const double*)s->smass.p,
                                (const uint32_t*)s->perm[s->sorted_key].p, s->walk_order(), scratch.p,
                                s->walk_fold, (uint32_t)s->w2_cmp_bits, theta2, s->cfg.gravity_constant,
                                s->cfg.softening, cnt.p);
    unsigned long long h[32];
    psim::be::copy_d2h(s->st, h, cnt.p, 256);
    psim::be::sync(s->st);
    for (int i = 0; i < n_out && i < 32; ++i) out[i] = (int64_t)h[i];
#else
    psim::DevBuf<uint32_t> cnt;
    cnt.ensure(2);
    psim::be::dev_memset(s->st, cnt.p, 0, 8);
    psim::GravityWalkSimple k;
    k.n = tn; k.s0 = 0; k.sorted_out = 0; k.order = nullptr; k.rec = s->rec.p; k.total_elems = total; k.spos = s->spos.p; k.smass = s->smass.p;
    k.perm = s->perm[s->sorted_key].p; k.inv_mass = s->inv_mass.p; k.acc_out = scratch.p; k.w2tab = s->w2tab.p;
    k.root_tag = s->root_tag(); k.theta2 = theta2; k.g_const = s->cfg.gravity_constant; k.softening = s->cfg.softening; k.counters = cnt.p;
    psim::be::launch(s->st, tn, k);
    out[0] = cnt.p[0];
    out[1] = cnt.p[1];
#endif
}

int psim_walk_counts(psim_context* ctx, int64_t out[4])
{
    PSIM_TRY(ctx && out, walk_counts_impl(ctx, out, 4))
}

int psim_walk_profile(psim_context* ctx, int64_t out[16])
{
    PSIM_TRY(ctx && out, walk_counts_impl(ctx, out, 16))
}

static void time_walk_impl(psim_context* ctx, int32_t reps, double slots_per_simd, double* ms_avg)
{
    use_device(ctx->device);
    psim::Sim* s = ctx->sim;
    *ms_avg = 0.0;
    if (!s->tree_valid || s->walk_n == 0) throw std::runtime_error("psim: no tree built yet");
    if (reps < 1 || reps > 1000 || !(slots_per_simd >= 0.0) || slots_per_simd > 8.0)
        throw std::invalid_argument("psim: bad repetition count or slot count");
#if !defined(PSIM_HOSTLOOP)
    const int64_t tn = s->walk_n;       // (see walk_counts_impl)
    uint32_t total = (uint32_t)(tn + s->n_internal);
    double theta2 = s->cfg.theta * s->cfg.theta;
    psim::DevBuf<psim::vec2> scratch;
    scratch.ensure((size_t)tn);
    int64_t resident = (int64_t)(psim::be::compute_units() * 4 * slots_per_simd);
    auto once = [&]() {
        psim::be::launch_walk_multi<2>(s->st, (int64_t)0, tn, 0, (const psim::NodeRec*)s->rec.p, total,
                                       (const psim::vec2*)s->spos.p, (const double*)s->smass.p,
                                       (const uint32_t*)s->perm[s->sorted_key].p, s->walk_order(), scratch.p,
                                       s->walk_fold, (uint32_t)s->w2_cmp_bits, theta2, s->cfg.gravity_constant,
                                       s->cfg.softening, s->walk_ctr.p, resident);
    };
    once();
    hipEvent_t e0, e1;
    PSIM_HIP(hipEventCreate(&e0));
    PSIM_HIP(hipEventCreate(&e1));
    PSIM_HIP(hipEventRecord(e0, s->st));
    for (int32_t r = 0; r < reps; ++r) once();
    PSIM_HIP(hipEventRecord(e1, s->st));
    PSIM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PSIM_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_avg = (double)ms / reps;
#endif
}

int psim_time_walk(psim_context* ctx, int32_t reps, double slots_per_simd, double* ms_avg)
{
    PSIM_TRY(ctx && ms_avg, time_walk_impl(ctx, reps, slots_per_simd, ms_avg))
}

static void brute_force_impl(psim_context* ctx, double* out, double* kernel_ms)
{
    use_device(ctx->device);
    psim::Sim* s = ctx->sim;
    if (kernel_ms) *kernel_ms = 0.0;
    if (s->n == 0) return;
    psim::DevBuf<psim::vec2> acc;
    acc.ensure((size_t)s->n);
#if !defined(PSIM_HOSTLOOP)
    hipEvent_t e0, e1;
    PSIM_HIP(hipEventCreate(&e0));
    PSIM_HIP(hipEventCreate(&e1));
    PSIM_HIP(hipEventRecord(e0, s->st));
    hipLaunchKernelGGL(psim::be::brute_force_kernel, dim3((unsigned)((s->n + 255) / 256)), dim3(256), 0, s->st, s->n,
                       (const psim::vec2*)s->pos.p, (const double*)s->mass.p, acc.p, s->cfg.gravity_constant,
                       s->cfg.softening);
    PSIM_HIP(hipGetLastError());
    PSIM_HIP(hipEventRecord(e1, s->st));
    PSIM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    PSIM_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (kernel_ms) *kernel_ms = (double)ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
#else
    for (int64_t i = 0; i < s->n; ++i) {
        double ax = 0.0, ay = 0.0;
        for (int64_t j = 0; j < s->n; ++j) {
            double dx = s->pos.p[j].x - s->pos.p[i].x, dy = s->pos.p[j].y - s->pos.p[i].y;
            double d2 = dx * dx + dy * dy;
            if (d2 > 0) {
                double f = s->mass.p[j] * psim::inv_cube_soft(d2 + s->cfg.softening);
                ax += f * dx;
                ay += f * dy;
            }
        }
        acc.p[i].x = s->cfg.gravity_constant * ax;
        acc.p[i].y = s->cfg.gravity_constant * ay;
    }
#endif
    psim::be::copy_d2h(s->st, out, acc.p, (size_t)s->n * 16);
    psim::be::sync(s->st);
}

int psim_brute_force_gravity(psim_context* ctx, double* accelerations_out, double* kernel_ms)
{
    PSIM_TRY(ctx && accelerations_out, brute_force_impl(ctx, accelerations_out, kernel_ms))
}

static void render_feed_impl(psim_context* ctx, int32_t* xy, uint8_t* rgba)
{
    use_device(ctx->device);
    psim::Sim* s = ctx->sim;
    if (s->n == 0) return;
    psim::DevBuf<int32_t> dxy;
    psim::DevBuf<uint8_t> drgba;
    dxy.ensure(2 * (size_t)s->n);
    drgba.ensure(4 * (size_t)s->n);
    psim::be::launch(s->st, s->n, psim::RenderFeed{s->n, s->pos.p, s->temp.p, dxy.p, drgba.p, 0.0, s->cfg.explode_temp});
    psim::be::copy_d2h(s->st, xy, dxy.p, (size_t)s->n * 8);
    psim::be::copy_d2h(s->st, rgba, drgba.p, (size_t)s->n * 4);
    psim::be::sync(s->st);
}

int psim_render_feed(psim_context* ctx, int32_t* xy_out, uint8_t* rgba_out)
{
    PSIM_TRY(ctx && xy_out && rgba_out, render_feed_impl(ctx, xy_out, rgba_out))
}

static void pair_potential_impl(psim_context* ctx, double* out)
{
    use_device(ctx->device);
    psim::Sim* s = ctx->sim;
    *out = 0.0;
    if (s->n == 0) return;
    psim::DevBuf<double> rows;
    rows.ensure((size_t)s->n);
    psim::be::launch(s->st, s->n, psim::PairPotentialRow{s->n, s->pos.p, s->mass.p, s->cfg.softening, rows.p});
    psim::be::reduce_sum_f64(s->st, rows.p, (uint64_t)s->n, s->red_tmp.p, s->red_out.p + 5);
    double sum = 0.0;
    s->readback(&sum, s->red_out.p + 5, 8);
    *out = -0.5 * s->cfg.gravity_constant * sum;
}

int psim_pair_potential_energy(psim_context* ctx, double* out)
{
    PSIM_TRY(ctx && out, pair_potential_impl(ctx, out))
}

// The same computation as pair_potential_impl with every intermediate in the library's
// file-scope device array: no hipMalloc / hipFree, no allocating call of any kind.
static void pair_potential_noalloc_impl(psim_context* ctx, double* out)
{
    use_device(ctx->device);
    psim::Sim* s = ctx->sim;
    *out = 0.0;
    if (s->n > kNoallocMaxParticles)
        throw std::invalid_argument("psim_pair_potential_energy_noalloc: " + std::to_string(s->n) +
                                    " particles, the fixed scratch array holds at most " +
                                    std::to_string(kNoallocMaxParticles));
    if (s->n == 0) return;
    double* const base = scratch_of(ctx->device);     // the array's address on this GPU
    double* const rows = base;
    double* const partial = base + kNoallocMaxParticles;
    double* const result = partial + psim::be::kReduceBlocks;
    // (every word read below is written first by this call: nothing has to start at zero)
    psim::be::launch(s->st, s->n, psim::PairPotentialRow{s->n, s->pos.p, s->mass.p, s->cfg.softening, rows});
    psim::be::reduce_sum_f64(s->st, rows, (uint64_t)s->n, partial, result);
    double sum = 0.0;
    s->readback(&sum, result, 8);
    *out = -0.5 * s->cfg.gravity_constant * sum;
}

int psim_pair_potential_energy_noalloc(psim_context* ctx, double* out)
{
    PSIM_TRY(ctx && out, pair_potential_noalloc_impl(ctx, out))
}

int psim_stage_times(psim_context* ctx, double out_ms[8])
{
    PSIM_TRY(ctx && out_ms, {
        use_device(ctx->device);
        ctx->sim->timer.read(out_ms);
    })
}

// ---- debug access
struct DebugArray {
    const void* ptr;
    int64_t count;
    size_t elem;
};

static bool find_debug_array(psim::Sim* s, const std::string& name, DebugArray* a)
{
    int64_t n = s->tree_n ? s->tree_n : s->n;
    int64_t cells = (int64_t)s->cfg.grid_w * s->cfg.grid_h;
    if (name == "key") *a = {s->key[s->sorted_key].p, n, 8};
    else if (name == "perm") *a = {s->perm[s->sorted_key].p, n, 4};
    else if (name == "L") *a = {s->L.p, n, 1};
    else if (name == "owned") *a = {s->owned.p, n, 4};
    else if (name == "base") *a = {s->base.p, n, 4};
    else if (name == "rec") *a = {s->rec.p, n + s->n_internal, sizeof(psim::NodeRec)};
    else if (name == "node_first") *a = {s->node_first.p, s->n_internal, 4};
    else if (name == "node_last") *a = {s->node_last.p, s->n_internal, 4};
    else if (name == "node_depth") *a = {s->node_depth[0].p, s->n_internal, 4};
    else if (name == "depth_offsets") *a = {s->depth_off.p, psim::kKeyLevels + 2, 4};
    // (the grid arrays exist when the last step built the whole grid: a step split by bands of rows may not)
    else if (name == "grid_offsets") *a = {s->offsets.p, s->full_grid_valid ? cells + 1 : 0, 4};
    else if (name == "grid_indices") *a = {s->cperm[s->sorted_ckey].p, s->full_grid_valid ? n : 0, 4};
    else if (name == "cell_key") *a = {s->ckey[s->sorted_ckey].p, s->full_grid_valid ? n : 0, 4};
    else if (name == "contact_ij") *a = {s->log_ij.p, s->log_n, 8};
    else if (name == "contact_vals") *a = {s->log_vals.p, s->log_n, 24};
    else if (name == "delta_temps") *a = {s->delta_dbg.p, s->delta_dbg.p ? n : 0, 8};
    else if (name == "exploded") *a = {nullptr, (int64_t)s->last_exploded.size(), 4};
    else if (name == "tree_path") *a = {nullptr, 2, 4};
    else return false;
    return true;
}

int64_t psim_debug_size(psim_context* ctx, const char* name)
{
    if (!ctx || !name) { g_err = "psim: null handle or argument"; return PSIM_ERR_INVALID; }
    DebugArray a;
    if (!find_debug_array(ctx->sim, name, &a)) { g_err = std::string("psim: unknown debug array ") + name; return PSIM_ERR_INVALID; }
    return a.count;
}

int psim_debug_fetch(psim_context* ctx, const char* name, void* dst, size_t bytes)
{
    PSIM_TRY(ctx && name && (dst || bytes == 0), {
        use_device(ctx->device);
        DebugArray a;
        if (!find_debug_array(ctx->sim, name, &a)) throw std::invalid_argument(std::string("psim: unknown debug array ") + name);
        size_t have = (size_t)a.count * a.elem;
        size_t k = std::min(bytes, have);
        if (std::string(name) == "exploded") {
            if (k) memcpy(dst, ctx->sim->last_exploded.data(), k);
        } else if (std::string(name) == "tree_path") {
            // how the last tree's elements were made: 1 the cooperative kernels, 2 the per-element ones
            // (and how many nodes spanning tiles the cooperative kernels settled separately)
            const uint32_t path[2] = {ctx->sim->tree_cooperative ? 1u : 2u, ctx->sim->tree_far_nodes};
            if (k) memcpy(dst, path, k);
        } else if (k) {
            psim::be::copy_d2h(ctx->sim->st, dst, a.ptr, k);
            psim::be::sync(ctx->sim->st);
        }
    })
}

// ---- primitive self-test
int psim_selftest(int device_ordinal, int64_t n, uint64_t seed)
{
    PSIM_TRY(n >= 0 && n < (1ll << 31), {
        use_device(device_ordinal);
        using namespace psim;
        be::Stream st = 0;
        be::selftest_pointer_kinds();
        std::mt19937_64 rng(seed);
        size_t c = (size_t)n;
        // --- u64 sort with payload, all 64 bits, duplicates forced
        std::vector<uint64_t> hk(c);
        std::vector<uint32_t> hv(c);
        for (size_t i = 0; i < c; ++i) { hk[i] = rng(); if (i % 7 == 3 && i) hk[i] = hk[i - 1]; hv[i] = (uint32_t)i; }
        DevBuf<uint64_t> k0, k1;
        DevBuf<uint32_t> v0, v1, tmp, stmp;
        k0.ensure(c); k1.ensure(c); v0.ensure(c); v1.ensure(c); tmp.ensure(be::sort_temp_words(c));
        be::copy_h2d(st, k0.p, hk.data(), c * 8);
        be::copy_h2d(st, v0.p, hv.data(), c * 4);
        uint64_t* kk[2] = {k0.p, k1.p};
        uint32_t* vv[2] = {v0.p, v1.p};
        int r = be::radix_sort_pairs<uint64_t>(st, kk, vv, (uint64_t)n, 0, 64, tmp.p);
        std::vector<uint64_t> gk(c);
        std::vector<uint32_t> gv(c);
        be::copy_d2h(st, gk.data(), kk[r], c * 8);
        be::copy_d2h(st, gv.data(), vv[r], c * 4);
        be::sync(st);
        std::vector<uint32_t> order(c);
        for (size_t i = 0; i < c; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; });
        for (size_t i = 0; i < c; ++i)
            if (gk[i] != hk[order[i]] || gv[i] != order[i]) throw std::runtime_error("selftest: u64 sort mismatch at " + std::to_string(i));
        // --- u32 sort on a bit sub-range [0, 20)
        std::vector<uint32_t> h32(c);
        for (size_t i = 0; i < c; ++i) h32[i] = (uint32_t)rng() & 0xfffffu;
        DevBuf<uint32_t> a0, a1;
        a0.ensure(c); a1.ensure(c);
        be::copy_h2d(st, a0.p, h32.data(), c * 4);
        be::copy_h2d(st, v0.p, hv.data(), c * 4);
        uint32_t* ak[2] = {a0.p, a1.p};
        r = be::radix_sort_pairs<uint32_t>(st, ak, vv, (uint64_t)n, 0, 20, tmp.p);
        std::vector<uint32_t> g32(c);
        be::copy_d2h(st, g32.data(), ak[r], c * 4);
        be::copy_d2h(st, gv.data(), vv[r], c * 4);
        be::sync(st);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return a < b; });
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h32[a] < h32[b]; });
        for (size_t i = 0; i < c; ++i)
            if (g32[i] != h32[order[i]] || gv[i] != order[i]) throw std::runtime_error("selftest: u32 sort mismatch at " + std::to_string(i));
        // --- exclusive scan
        std::vector<uint32_t> hs(c), gs(c);
        for (size_t i = 0; i < c; ++i) hs[i] = (uint32_t)(rng() % 5);
        stmp.ensure_zero(st, be::scan_temp_words(c));
        DevBuf<uint32_t> tot;
        tot.ensure(1);
        be::copy_h2d(st, a0.p, hs.data(), c * 4);
        be::exclusive_scan_u32(st, a0.p, a1.p, (uint64_t)n, stmp.p, tot.p);
        uint32_t gtot = 0;
        be::copy_d2h(st, gs.data(), a1.p, c * 4);
        be::copy_d2h(st, &gtot, tot.p, 4);
        be::sync(st);
        uint32_t run = 0;
        for (size_t i = 0; i < c; ++i) { if (gs[i] != run) throw std::runtime_error("selftest: scan mismatch at " + std::to_string(i)); run += hs[i]; }
        if (gtot != run) throw std::runtime_error("selftest: scan total mismatch");
        // again, in place, on the scratch the first scan left behind, with the small workgroups
        be::exclusive_scan_u32(st, a0.p, a0.p, (uint64_t)n, stmp.p, tot.p, true);
        be::copy_d2h(st, gs.data(), a0.p, c * 4);
        be::copy_d2h(st, &gtot, tot.p, 4);
        be::sync(st);
        run = 0;
        for (size_t i = 0; i < c; ++i) { if (gs[i] != run) throw std::runtime_error("selftest: in-place scan mismatch at " + std::to_string(i)); run += hs[i]; }
        if (gtot != run) throw std::runtime_error("selftest: in-place scan total mismatch");
        // --- reduction (integers stored as doubles: exact in any order)
        std::vector<double> hd(c);
        double want = 0.0;
        for (size_t i = 0; i < c; ++i) { hd[i] = (double)(rng() % 1000); want += hd[i]; }
        DevBuf<double> d0, dt, dout;
        d0.ensure(c); dt.ensure(be::kReduceBlocks); dout.ensure(1);
        be::copy_h2d(st, d0.p, hd.data(), c * 8);
        be::reduce_sum_f64(st, d0.p, (uint64_t)n, dt.p, dout.p);
        double got = -1.0;
        be::copy_d2h(st, &got, dout.p, 8);
        be::sync(st);
        if (got != want) throw std::runtime_error("selftest: reduction mismatch");
    })
}

}  // extern "C"

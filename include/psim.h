/*
 * psim.h -- C ABI of libpsim_hip.so, the MI355X (gfx950) implementation of the
 * per-timestep physics step of NeoLorenzo/Particle-Simulator.
 *
 * The reference has no FFI: its boundary is the Python class ParticleSystem
 * (particle_system.py:339-804) as main.py drives it.  Each entry point below
 * names the reference interface it stands in for (file:line into the reference
 * checkout); psim_amd/particle_system.py binds them with ctypes and re-creates
 * that class on top.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success and a negative code on failure, with the message available from
 * psim_last_error() (the reference raises Python exceptions; the binding turns
 * codes back into exceptions).  One context per device; a context is not
 * re-entrant; all calls are synchronous (state is complete on return).
 * Host arrays use the reference's layouts: positions / velocities /
 * accelerations (n,2) float64 C-contiguous, masses / inverse_masses /
 * temperatures (n,1) float64, radii (n,1) int64 (particle_system.py:370-376).
 */
#ifndef PSIM_H
#define PSIM_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSIM_OK 0
#define PSIM_ERR_INVALID (-1)   /* bad argument / handle */
#define PSIM_ERR_RUNTIME (-2)   /* HIP failure, node-pool overflow, ... */

/* stage mask bits for psim_config.stage_mask (all set = the reference's update()) */
#define PSIM_STAGE_COLLISIONS 1
#define PSIM_STAGE_CONDUCTION 2
#define PSIM_STAGE_COOLING 4
#define PSIM_STAGE_WALLS 8
#define PSIM_STAGE_EXPLOSIONS 16
#define PSIM_STAGE_ALL 31

/* The "simulation" section of config.json (config.json:8-24) plus the values
 * ParticleSystem.__init__ derives from it (particle_system.py:358-397) and the
 * constants.py values the step reads (WIDTH/HEIGHT via bounds, COLOR_MAX_TEMP). */
typedef struct psim_config {
    double bounds_w, bounds_h;        /* main.py:140, constants.py:15-16 */
    double gravity_constant;          /* config.json:12 */
    double softening;                 /* config.json:13 softening_factor */
    double theta;                     /* config.json:14 barnes_hut_theta */
    double heat_coeff;                /* config.json:17 heat_transfer_coefficient */
    double restitution;               /* config.json:18 */
    double pos_correct;               /* config.json:20 position_correction_factor (default 1.0) */
    double explosion_eff;             /* config.json:22 explosion_efficiency (default 0.1) */
    double damping;                   /* config.json:23 thermal_damping_factor (default 0.0) */
    double explode_temp;              /* constants.py:33 COLOR_MAX_TEMP */
    int32_t cell_size, grid_w, grid_h;/* particle_system.py:381-390 */
    int32_t solver_iterations;        /* config.json:21 collision_solver_iterations (default 1) */
    int32_t stage_mask;               /* PSIM_STAGE_*; the reference has no such switch */
    int32_t walk_mode;                /* 0 = wave-cooperative tree walk, two particle groups per wave;
                                         1 = one lane per particle; 2 = wave-cooperative, one group per wave;
                                         3 / 4 = as 0 / 2 with the 64-bit element offsets the kernels switch to
                                         for trees of 4 GiB and more forced on (tests); 5 = as 0 with the
                                         compiler's instruction schedule of the walk loop instead of the
                                         hand-written one (tests: both must agree bit for bit) */
    int32_t debug;                    /* bit 0: keep the per-step contact log and delta-temperature array;
                                         bit 1: take the contact solver's large-hypothesis path at any size (tests);
                                         bit 2: the contact solver's re-test trusts no per-slot digest and looks at
                                         every candidate pair around every touched slot in every iteration (tests:
                                         the contact lists must be the same with and without) */
    int32_t overlap;                  /* 1 = run the grid build and the first contact-sweep count pass on a
                                         side stream underneath the tree walk (default in the Python binding) */
} psim_config;

typedef struct psim_context psim_context;

/* replaces: ParticleSystem.__init__ up to the state draws (particle_system.py:355-368).
 * device_ordinal selects the HIP device for every later call on the context. */
int psim_create(const psim_config* cfg, int device_ordinal, psim_context** out);
int psim_destroy(psim_context* ctx);

/* replaces: re-reading self.config each step (particle_system.py:430,462,555,673) */
int psim_set_config(psim_context* ctx, const psim_config* cfg);

/* replaces: the state arrays created in __init__ (particle_system.py:370-376).
 * vel / acc may be NULL (zeros).  inverse_masses are derived as 1.0 / masses. */
int psim_upload(psim_context* ctx, int64_t n, const double* positions, const double* velocities,
                const double* accelerations, const double* masses, const double* temperatures,
                const int64_t* radii);

/* replaces: reading the public state arrays after a step (particle_system.py:594-624, 716-725).
 * Any pointer may be NULL.  Arrays must hold psim_num_particles() rows. */
int psim_download(psim_context* ctx, double* positions, double* velocities, double* accelerations,
                  double* masses, double* inverse_masses, double* temperatures, int64_t* radii);

int64_t psim_num_particles(psim_context* ctx);   /* particle_system.py:725 num_particles */

/* replaces: ParticleSystem._calculate_gravity() on a tree of the current positions
 * (particle_system.py:409-437; main.py:151 primes a(0) with it right after __init__) */
int psim_prime_gravity(psim_context* ctx);

/* replaces: ParticleSystem.update() called k_steps times (particle_system.py:518-563) */
int psim_step(psim_context* ctx, int64_t k_steps);

/* particles alive after each step, summed over the steps of the last psim_step call (the
 * number of particle-updates it performed; explosions shrink the count, particle_system.py:725) */
int64_t psim_particle_updates(psim_context* ctx);

/* One simulation split over several GPUs (no counterpart in the reference, which is
 * single-process): psim_step cut at its exchange points.  Every rank holds the full state
 * and builds the same tree.  The walk runs on a rank's share of the Morton-sorted
 * particles.  The collision stage and conduction (particle_system.py:439-498) run on bands
 * of whole grid rows: cells are row-major (particle_system.py:749), so a band is a slab of
 * the domain; a rank solves the rows it owns plus halo rows on a grid of that band only and
 * exports the results of its own rows; along every seam both ranks also export the contacts
 * they saw in the two rows next to it, and the results are applied only when those records
 * are identical (each rank's view of its neighbour's rows was then the true one, and the
 * owned results together are exactly the sequential sweep's).  Otherwise the step's
 * collision stage is run again on the whole grid by every rank.  Calls, in order:
 *   psim_stage_build          drift, Morton keys, tree (identical on every rank)
 *   psim_stage_gravity        walk for the sorted particles [s_lo, s_hi) -> acc_sorted[s]
 *                             (device pointer, (n,2) float64, indexed by sorted position).
 *                             Returns once the walk is submitted to the context's stream (the
 *                             device's default stream): an exchange submitted to that stream,
 *                             or to one that waits for it, is ordered after the walk.
 *   -- all-gather acc_sorted (may be started asynchronously) --
 *   psim_stage_slab_positions grid of the rows [row_lo, row_hi), contact list and pushed-apart
 *                             positions of that band ([own_lo, own_hi) = the rows whose results
 *                             this rank keeps; the rest is halo).  On the library's second
 *                             stream while a walk is in flight.
 *   psim_stage_kick           second half-kick from the gathered accelerations
 *   psim_stage_slab_finish    velocity half of the band's contacts, conduction sums
 *   psim_stage_slab_export    the band's block of 64-bit words into `block` (device pointer, room
 *                             for cap_words): header, state of the touched particles of the own
 *                             rows, conduction sums of the own rows, seam records.
 *                             *words_needed > cap_words: only the header was written; call again
 *                             with room (any number of times before the next band or step).
 *   -- all-gather the blocks (equal size on every rank) --
 *   psim_stage_apply          blocks = `world` blocks of block_words words in rank order (device
 *                             pointer).  *status 0: seams agree, all results applied.  1: a seam
 *                             disagrees, nothing applied: call psim_stage_fallback.  2: a block was
 *                             incomplete (*words_needed = the largest need): export again, gather
 *                             again, call again.  own_slots (may be NULL): [world] particles in
 *                             each rank's own rows, for balancing the next step's bands.
 *   psim_stage_fallback       the whole collision stage on the whole grid (every rank alike)
 *   psim_stage_finish         conduction applied, cooling, walls, explosions
 * Results are bit-identical to psim_step for any number of ranks and any bands that tile the
 * grid rows (halo >= 2 rows on either side of the own rows, clipped to the grid), except the
 * three collision totals, which are sums of the ranks' partial sums.
 * acc_sorted, block and blocks are memory the GPU works on directly: device memory or
 * page-locked host memory; any other pointer is PSIM_ERR_INVALID (checked before a kernel sees it).
 * psim_stage_stats: [0] steps that took the fallback so far, [1] particles in the own rows of
 * the last band, [2] particles in the last band, [3] words of a block's header. */
int psim_stage_build(psim_context* ctx);
int psim_stage_gravity(psim_context* ctx, int64_t s_lo, int64_t s_hi, double* acc_sorted);
int psim_stage_slab_positions(psim_context* ctx, int32_t row_lo, int32_t row_hi, int32_t own_lo, int32_t own_hi);
int psim_stage_kick(psim_context* ctx, const double* acc_sorted);
int psim_stage_slab_finish(psim_context* ctx);
int psim_stage_slab_export(psim_context* ctx, uint64_t* block, int64_t cap_words, int64_t* words_needed);
int psim_stage_apply(psim_context* ctx, const uint64_t* blocks, int32_t world, int64_t block_words, int32_t* status,
                     int64_t* own_slots, int64_t* words_needed);
int psim_stage_fallback(psim_context* ctx);
int psim_stage_finish(psim_context* ctx);
int psim_stage_stats(psim_context* ctx, int64_t out[4]);

/* replaces: get_total_kinetic_energy / get_total_potential_energy /
 * get_total_thermal_energy (particle_system.py:765-804); out[0..2] = KE, PE, TE,
 * out[3..4] = total momentum (sum m v), which the reference does not report. */
int psim_energies(psim_context* ctx, double out[5]);

/* replaces: .pe_gain_collisions / .ke_loss_collisions / .heat_generated_collisions
 * of the last step (particle_system.py:472-474) */
int psim_collision_stats(psim_context* ctx, double out[3]);

/* counters of the last step / tree:
 * [0] particles, [1] reference-equivalent node count 1 + 4 * internal nodes
 * (quadtree.py:142 num_active_nodes), [2] candidate pairs tested (debug builds of
 * the sweep only, else 0), [3] contacts resolved, [4] particles removed by
 * explosions, [5] internal nodes, [6] fixed-point iterations of the contact
 * solver, [7] dependency rounds of the contact solver */
int psim_counts(psim_context* ctx, int64_t out[8]);

/* replaces: the per-explosion log lines (particle_system.py:683-689): drains the
 * queued events; each event = {particle index before removal, affected
 * neighbours, energy}.  Returns the number written (<= cap) or < 0. */
typedef struct psim_explosion_event {
    int32_t particle;
    int32_t neighbours;
    double energy;
} psim_explosion_event;
int64_t psim_explosion_events(psim_context* ctx, psim_explosion_event* buf, int64_t cap);

/* gravity work counters: runs one counted walk over the current tree and returns
 * [0] node visits and [1] accepted interactions summed over particles (the
 * reference's own per-particle recursion makes exactly these), [2] loop trips
 * summed over 64-lane waves, [3] trips in which some lane accepted */
int psim_walk_counts(psim_context* ctx, int64_t out[4]);

/* the same counted walk with the lane statistics of its loop (no counterpart in the reference):
 * [0..3] as psim_walk_counts; of the trips with at least one active lane, [4] those in which
 * every active lane accepted the element, [5] those in which none did, [6] / [7] those of [4]
 * / of the internal-node trips of [5] that follow from the bounding box of the group's 64
 * targets alone (nearest corner already beyond the opening distance / farthest corner still
 * inside it), [8] trips with all 64 lanes active; the rest zero */
int psim_walk_profile(psim_context* ctx, int64_t out[16]);

/* timing hook for benchmarks: average duration in milliseconds of `reps` launches of the
 * default walk kernel on the current tree with nothing else on the device (results go to
 * a scratch buffer).  slots_per_simd = 0 launches the whole grid at once; a value up to 8
 * keeps that many waves per SIMD resident and has them take chunks of targets from a
 * counter, as psim_step does when it runs other work next to the walk.  No counterpart in
 * the reference. */
int psim_time_walk(psim_context* ctx, int32_t reps, double slots_per_simd, double* ms_avg);

/* cross-check (no counterpart in the reference): all-pairs softened gravity of the
 * current state with the force law of particle_system.py:86-93, O(n^2) FP64.
 * accelerations_out: host (n,2) float64.  kernel_ms (may be NULL) receives the
 * device time of the kernel; it performs 20 n (n-1) flops by the walk's accounting. */
int psim_brute_force_gravity(psim_context* ctx, double* accelerations_out, double* kernel_ms);

/* draw() feed, replaces the per-particle Python of particle_system.py:587-624 up to the
 * rasteriser call: xy_out host (n,2) int32 = int(position), rgba_out host (n,4) uint8 =
 * colour of the temperature through the keyframe ramp (constants.py:37-45,
 * particle_system.py:565-585), alpha 255.  12 bytes per particle cross PCIe instead of 80. */
int psim_render_feed(psim_context* ctx, int32_t* xy_out, uint8_t* rgba_out);

/* diagnostic (no counterpart in the reference, whose potential energy pairs sibling cells
 * only, particle_system.py:27-67): exact softened pairwise potential energy
 * -G sum_{i<j} m_i m_j / sqrt(d_ij^2 + s) of the current state, O(n^2) FP64. */
int psim_pair_potential_energy(psim_context* ctx, double* out);

/* No counterpart in the reference.  psim_pair_potential_energy for callers that cannot
 * allocate device memory (tight serving loops, harnesses that forbid hipMalloc): same
 * kernels, same result bit for bit, but every intermediate (one row sum per particle, the
 * reduction's partial sums) lives in a fixed device array inside the library (2 MiB + 8 KiB
 * per GPU, held until the process ends) and the call makes no hipMalloc / hipFree or other
 * allocating call.  It is synchronous like every other entry point (the result is read back
 * from the context's stream, the device's default stream), so it cannot itself run under a
 * HIP graph capture.  At most 262,144 particles: more is PSIM_ERR_INVALID with a message
 * naming the limit.  All contexts on one GPU share that array: calls that overlap in time on
 * the same GPU race (calls on different GPUs do not). */
int psim_pair_potential_energy_noalloc(psim_context* ctx, double* out);

/* parity hooks: copy a named internal device array to the host.
 * psim_debug_size returns the element count (or < 0 for an unknown name);
 * psim_debug_fetch copies min(bytes, size) bytes.  Names: key perm L owned base
 * rec node_first node_last node_depth grid_offsets grid_indices cell_key
 * contact_ij contact_vals delta_temps exploded depth_offsets.
 * contact_ij / contact_vals (debug bit 0) hold one row per pair of the contact solver's settled
 * list, in the reference's visiting order; a row of the list that is no contact of this context
 * -- the pair did not touch when its turn came, or, in one rank's band of a split step, the
 * contact is another rank's to count -- has both indices -1 and zeros for values. */
int64_t psim_debug_size(psim_context* ctx, const char* name);
int psim_debug_fetch(psim_context* ctx, const char* name, void* dst, size_t bytes);

/* self-test of the device primitives (radix sort, scan, reduction) on n seeded
 * items against host results, and of the check that tells memory the GPU can reach from
 * ordinary host memory; 0 = all matched */
int psim_selftest(int device_ordinal, int64_t n, uint64_t seed);

/* timing hook for benchmarks: device-side duration in milliseconds of each
 * stage, averaged over the steps of the LAST psim_step call (its first 64 steps; HIP events
 * on the context's stream, read after the call: no synchronisation inside it):
 * [0] drift+keys, [1] grid build, [2] tree build, [3] gravity walk, [4] kick,
 * [5] collisions, [6] conduction+cooling, [7] walls+explosions */
int psim_stage_times(psim_context* ctx, double out_ms[8]);

const char* psim_last_error(void);
const char* psim_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PSIM_H */

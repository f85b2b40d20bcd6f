/*
 * psim_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A sequential, scalar, strict-IEEE FP64 restatement in C of the per-timestep
 * physics step of NeoLorenzo/Particle-Simulator, written from the behaviour of
 * the reference's Python source.  Every function cites the reference lines it
 * follows (paths are relative to the reference checkout).  It exists so that
 * tests can check the HIP library against the reference's algorithm at sizes
 * the pure-Python reference cannot reach, and so that bench.py can time the
 * reference's (single-core) CPU algorithm on the GPU box's host.
 *
 * Parity status: PINNED by the fixtures in tests/golden/, which were produced
 * by executing the reference's own source under CPython in this build's
 * container (tests/golden/make_golden.py).  The reference itself ships no
 * tests or golden vectors.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this file.  The product library never does.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 *        (contraction off: the reference's strict stages must not fuse a*b+c).
 *
 * Threads: everything is sequential, as in the reference, except that
 * orc_set_threads(k > 1) lets the per-particle gravity loop (independent
 * iterations, each writing only its own acceleration) run on k threads so the
 * tests can afford full-size clouds.  Results do not depend on k.  The
 * benchmark's CPU baseline runs with k = 1.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Timing build only (-DORC_FAST_STAGES, libpsim_oracle_fast.so): the functions that restate
 * the reference's six `@numba.jit(fastmath=True)` kernels (particle_system.py:17, 27, 69,
 * 137, 271, 312) are compiled with -ffast-math, as Numba compiles them; everything else,
 * in particular the strict pair resolver (:200) and the tree build, stays strict.  The
 * parity tests never use this build. */
#if defined(ORC_FAST_STAGES)
#define ORC_FAST __attribute__((optimize("fast-math")))
#else
#define ORC_FAST
#endif

typedef struct {
    /* configuration (reference: config.json "simulation", particle_system.py:355-397) */
    double min_mass, max_mass;
    double gravity_constant, softening, theta;
    double heat_coeff, restitution, pos_correct, explosion_eff, damping;
    double explode_temp;           /* constants.py:33 COLOR_MAX_TEMP */
    double bounds_w, bounds_h;
    int solver_iterations;
    int cell_size, grid_w, grid_h;

    /* particle state, reference layout (particle_system.py:370-376) */
    int64_t n, cap;
    double *pos, *vel, *acc;       /* (n,2) */
    double *mass, *inv_mass, *temp;/* (n,1) */
    int64_t *radii;                /* (n,1) int64 */

    /* grid CSR (particle_system.py:393-395) */
    int32_t *grid_offsets, *grid_indices;
    int64_t grid_n;                /* particle count the grid was built for */

    /* node pool (quadtree.py:161-175) */
    int64_t max_nodes, num_nodes;
    double *node_bounds;           /* (max_nodes,4) x,y,w,h */
    double *node_data;             /* (max_nodes,4) mass,comx,comy,width */
    int32_t *node_children;        /* (max_nodes,4) */
    uint8_t *node_is_leaf;
    int32_t *node_particle;
    int32_t *node_map;             /* (max_nodes,3) start,count,-1 */
    int32_t *particle_list;
    int tree_cached;               /* gravity ran since the build (particle_system.py:421) */

    /* per-step outputs */
    double pe_gain, ke_loss, heat_generated;      /* particle_system.py:472-474 */
    int64_t n_candidates, n_contacts;
    /* optional contact log */
    int64_t log_cap, log_n;
    int32_t *log_ij;               /* (log_cap,2) */
    double *log_vals;              /* (log_cap,3) pe,ke,heat */
    double *delta_temps;           /* (cap) last conduction delta */
    int32_t *exploded;             /* indices removed by the last step */
    int64_t n_exploded;
    int fast_pow;                  /* 0: libm pow as CPython; 1: rsqrt form (timing) */
    int threads;                   /* gravity loop threads (tests only), default 1 */
} orc_system;

/* ---------------------------------------------------------------- grid */
/* particle_system.py:732-763  _build_spatial_grid */
static int32_t f64_to_i32_numpy(double v)
{
    /* numpy .astype(np.int32) on x86-64 is cvttsd2si: out-of-range and NaN
       give INT32_MIN */
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

static void build_grid(orc_system *s)
{
    int64_t n = s->n;
    int64_t cells = (int64_t)s->grid_w * s->grid_h;
    int32_t *cell_of = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    int32_t *cursor = (int32_t *)calloc(cells + 1, sizeof(int32_t));
    double cs = (double)s->cell_size;
    memset(s->grid_offsets, 0, sizeof(int32_t) * (cells + 1));
    for (int64_t i = 0; i < n; i++) {
        int32_t cx = f64_to_i32_numpy(s->pos[2 * i] / cs);
        int32_t cy = f64_to_i32_numpy(s->pos[2 * i + 1] / cs);
        if (cx < 0) cx = 0;
        if (cx > s->grid_w - 1) cx = s->grid_w - 1;
        if (cy < 0) cy = 0;
        if (cy > s->grid_h - 1) cy = s->grid_h - 1;
        cell_of[i] = cy * s->grid_w + cx;
        s->grid_offsets[cell_of[i] + 1]++;
    }
    for (int64_t c = 0; c < cells; c++) s->grid_offsets[c + 1] += s->grid_offsets[c];
    memcpy(cursor, s->grid_offsets, sizeof(int32_t) * cells);
    for (int64_t i = 0; i < n; i++) s->grid_indices[cursor[cell_of[i]]++] = (int32_t)i;
    s->grid_n = n;
    free(cell_of);
    free(cursor);
}

/* ---------------------------------------------------------------- tree */
/* quadtree.py:13-19 _get_quadrant */
static inline int quadrant(const double *p, double cx, double cy)
{
    if (p[0] < cx) return (p[1] < cy) ? 0 : 2;
    return (p[1] < cy) ? 1 : 3;
}

/* quadtree.py:21-40 _subdivide_jit */
static int64_t subdivide(orc_system *s, int64_t node, int64_t next)
{
    double x = s->node_bounds[4 * node], y = s->node_bounds[4 * node + 1];
    double hw = s->node_bounds[4 * node + 2] / 2, hh = s->node_bounds[4 * node + 3] / 2;
    for (int q = 0; q < 4; q++) {
        int64_t c = next + q;
        s->node_children[4 * node + q] = (int32_t)c;
        s->node_bounds[4 * c] = (q & 1) ? x + hw : x;
        s->node_bounds[4 * c + 1] = (q & 2) ? y + hh : y;
        s->node_bounds[4 * c + 2] = hw;
        s->node_bounds[4 * c + 3] = hh;
    }
    return next + 4;
}

/* quadtree.py:42-83 _insert_jit.  Returns -1 when the pool would overflow
   (the reference corrupts memory there; coincident particles never separate). */
static int64_t insert(orc_system *s, int32_t p, int64_t node, int64_t next)
{
    for (;;) {
        double x = s->node_bounds[4 * node], y = s->node_bounds[4 * node + 1];
        double w = s->node_bounds[4 * node + 2], h = s->node_bounds[4 * node + 3];
        double cx = x + w / 2, cy = y + h / 2;
        if (s->node_is_leaf[node]) {
            if (s->node_particle[node] == -1) {
                s->node_particle[node] = p;
                return next;
            }
            int32_t old = s->node_particle[node];
            s->node_particle[node] = -1;
            s->node_is_leaf[node] = 0;
            if (next + 4 > s->max_nodes) return -1;
            next = subdivide(s, node, next);
            /* the old particle lands in a fresh empty child: direct placement */
            int qo = quadrant(&s->pos[2 * old], cx, cy);
            s->node_particle[s->node_children[4 * node + qo]] = old;
            node = s->node_children[4 * node + quadrant(&s->pos[2 * p], cx, cy)];
        } else {
            node = s->node_children[4 * node + quadrant(&s->pos[2 * p], cx, cy)];
        }
    }
}

/* quadtree.py:85-114 _calculate_mass_distribution_jit (post-order, children 0..3) */
static void mass_distribution(orc_system *s, int64_t node)
{
    if (s->node_is_leaf[node]) {
        int32_t p = s->node_particle[node];
        if (p != -1) {
            s->node_data[4 * node] = s->mass[p];
            s->node_data[4 * node + 1] = s->pos[2 * p];
            s->node_data[4 * node + 2] = s->pos[2 * p + 1];
        }
        return;
    }
    double total = 0.0, nx = 0.0, ny = 0.0;
    for (int q = 0; q < 4; q++) {
        int32_t c = s->node_children[4 * node + q];
        if (c != -1) {
            mass_distribution(s, c);
            double cm = s->node_data[4 * c];
            if (cm > 0) {
                total += cm;
                nx += s->node_data[4 * c + 1] * cm;
                ny += s->node_data[4 * c + 2] * cm;
            }
        }
    }
    s->node_data[4 * node] = total;
    if (total > 0) {
        s->node_data[4 * node + 1] = nx / total;
        s->node_data[4 * node + 2] = ny / total;
    }
}

/* quadtree.py:177-196 build + :116-142 _build_tree_and_map_jit */
static int build_tree(orc_system *s)
{
    int64_t n = s->n;
    int64_t used = s->num_nodes > 0 ? s->num_nodes : 1;
    /* the reference refills the whole 4N+1 pool (quadtree.py:182-188); the
       observable result is the same when only the previously used prefix is
       reset, which keeps the timed baseline honest about that cost: we do the
       full refill like the reference. */
    used = s->max_nodes;
    for (int64_t i = 0; i < used; i++) {
        s->node_is_leaf[i] = 1;
        s->node_particle[i] = -1;
        s->node_map[3 * i] = -1; s->node_map[3 * i + 1] = 0; s->node_map[3 * i + 2] = -1;
    }
    memset(s->node_data, 0, sizeof(double) * 4 * used);
    for (int64_t i = 0; i < 4 * used; i++) s->node_children[i] = -1;
    s->node_bounds[0] = 0.0; s->node_bounds[1] = 0.0;
    s->node_bounds[2] = s->bounds_w; s->node_bounds[3] = s->bounds_h;
    s->tree_cached = 0;
    if (n == 0) return 0;            /* num_active_nodes keeps its old value (quadtree.py:191) */
    int64_t next = 1;
    for (int64_t i = 0; i < n; i++) {
        next = insert(s, (int32_t)i, 0, next);
        if (next < 0) return -1;
    }
    mass_distribution(s, 0);
    int32_t counter = 0;
    for (int64_t i = 0; i < next; i++) {
        s->node_data[4 * i + 3] = s->node_bounds[4 * i + 2];
        if (s->node_is_leaf[i]) {
            int32_t p = s->node_particle[i];
            if (p != -1) {
                s->particle_list[counter] = p;
                s->node_map[3 * i] = counter;
                s->node_map[3 * i + 1] = 1;
                counter++;
            }
        }
    }
    s->num_nodes = next;
    return 0;
}

/* ---------------------------------------------------------------- gravity */
static inline ORC_FAST double inv_cubed(const orc_system *s, double x)
{
    if (s->fast_pow) { double r = 1.0 / sqrt(x); return r * r * r; }
    return pow(x, -1.5);            /* particle_system.py:90,117 */
}

/* particle_system.py:69-121 _traverse_tree_jit */
static ORC_FAST void traverse(const orc_system *s, int64_t p, int64_t node, double *fx, double *fy,
                     int64_t *visits, int64_t *inter)
{
    const double *nd = &s->node_data[4 * node];
    double nfx = 0.0, nfy = 0.0;
    (*visits)++;
    if (s->node_children[4 * node] != -1) {
        double dx = nd[1] - s->pos[2 * p], dy = nd[2] - s->pos[2 * p + 1];
        double d2 = dx * dx + dy * dy;
        if (d2 > 0 && (nd[3] * nd[3]) < (s->theta * s->theta * d2)) {
            double f = s->gravity_constant * s->mass[p] * nd[0] * inv_cubed(s, d2 + s->softening);
            *fx = f * dx; *fy = f * dy;
            (*inter)++;
            return;
        }
        for (int q = 0; q < 4; q++) {
            int32_t c = s->node_children[4 * node + q];
            if (c != -1 && s->node_data[4 * c] > 0) {
                double cfx, cfy;
                traverse(s, p, c, &cfx, &cfy, visits, inter);
                nfx += cfx; nfy += cfy;
            }
        }
        *fx = nfx; *fy = nfy;
        return;
    }
    int32_t start = s->node_map[3 * node], count = s->node_map[3 * node + 1];
    for (int32_t k = 0; k < count; k++) {
        int32_t o = s->particle_list[start + k];
        if (o != p) {
            double dx = s->pos[2 * o] - s->pos[2 * p], dy = s->pos[2 * o + 1] - s->pos[2 * p + 1];
            double d2 = dx * dx + dy * dy;
            if (d2 > 0) {
                double f = s->gravity_constant * s->mass[p] * s->mass[o] * inv_cubed(s, d2 + s->softening);
                nfx += f * dx; nfy += f * dy;
                (*inter)++;
            }
        }
    }
    *fx = nfx; *fy = nfy;
}

/* particle_system.py:409-437 _calculate_gravity + :17-25 */
static ORC_FAST void gravity(orc_system *s, int64_t *visits, int64_t *inter)
{
    int64_t v = 0, it = 0;
    memset(s->acc, 0, sizeof(double) * 2 * s->n);
    s->tree_cached = 1;
    if (s->num_nodes == 0) return;
    int threads = s->threads > 1 ? s->threads : 1;
    #pragma omp parallel for reduction(+:v, it) num_threads(threads) schedule(static, 4096) if (threads > 1)
    for (int64_t i = 0; i < s->n; i++) {
        double fx, fy;
        traverse(s, i, 0, &fx, &fy, &v, &it);
        s->acc[2 * i] = fx * s->inv_mass[i];
        s->acc[2 * i + 1] = fy * s->inv_mass[i];
    }
    if (visits) *visits = v;
    if (inter) *inter = it;
}

/* particle_system.py:27-67 _calculate_potential_energy_jit */
static ORC_FAST double potential(const orc_system *s, int64_t node)
{
    double total = 0.0;
    if (s->node_children[4 * node] == -1) return 0.0;
    for (int i = 0; i < 4; i++) {
        int32_t ci = s->node_children[4 * node + i];
        if (ci != -1 && s->node_data[4 * ci] > 0) {
            total += potential(s, ci);
            for (int j = i + 1; j < 4; j++) {
                int32_t cj = s->node_children[4 * node + j];
                if (cj != -1 && s->node_data[4 * cj] > 0) {
                    double mi = s->node_data[4 * ci], mj = s->node_data[4 * cj];
                    double dx = s->node_data[4 * cj + 1] - s->node_data[4 * ci + 1];
                    double dy = s->node_data[4 * cj + 2] - s->node_data[4 * ci + 2];
                    double d2 = dx * dx + dy * dy;
                    if (d2 > 0) total -= (s->gravity_constant * mi * mj) / sqrt(d2 + s->softening);
                }
            }
        }
    }
    return total;
}

/* ---------------------------------------------------------------- collisions */
/* particle_system.py:200-268 _resolve_collision_jit.  Strict FP64; the order of
   every operation follows the source expression.  Returns 1 on contact. */
static int resolve(orc_system *s, int32_t i, int32_t j, double *pe_o, double *ke_o, double *heat_o)
{
    double *pi = &s->pos[2 * i], *pj = &s->pos[2 * j];
    double *vi = &s->vel[2 * i], *vj = &s->vel[2 * j];
    int64_t min_dist = s->radii[i] + s->radii[j];
    double dvx = pj[0] - pi[0], dvy = pj[1] - pi[1];
    double d2 = dvx * dvx + dvy * dvy;
    *pe_o = 0.0; *ke_o = 0.0; *heat_o = 0.0;
    if (!(d2 < (double)(min_dist * min_dist) && d2 > 0)) return 0;
    double dist = sqrt(d2);
    double mi = s->mass[i], mj = s->mass[j], M = mi + mj;
    double vib0 = vi[0], vib1 = vi[1], vjb0 = vj[0], vjb1 = vj[1];
    double G = s->gravity_constant, soft = s->softening;
    double ke_before = 0.5 * mi * (vib0 * vib0 + vib1 * vib1) + 0.5 * mj * (vjb0 * vjb0 + vjb1 * vjb1);
    double pe_before = (-G * mi * mj) / sqrt(d2 + soft);
    double overlap = ((double)min_dist - dist) * s->pos_correct;
    double nx = dvx / dist, ny = dvy / dist;
    double ci = (mj / M) * overlap, cj = (mi / M) * overlap;
    pi[0] -= ci * nx; pi[1] -= ci * ny;
    pj[0] += cj * nx; pj[1] += cj * ny;
    double fx = pj[0] - pi[0], fy = pj[1] - pi[1];
    double fd2 = fx * fx + fy * fy;
    double pe_after = (-G * mi * mj) / sqrt(fd2 + soft);
    double pe_change = pe_after - pe_before;
    double vrx = vjb0 - vib0, vry = vjb1 - vib1;
    double vrn = vrx * nx + vry * ny;
    if (vrn < 0) {
        double imp = (-(1 + s->restitution) * mi * mj * vrn) / M;
        double ii = imp / mi, ij = imp / mj;
        vi[0] -= ii * nx; vi[1] -= ii * ny;
        vj[0] += ij * nx; vj[1] += ij * ny;
    }
    double ke_after = 0.5 * mi * (vi[0] * vi[0] + vi[1] * vi[1]) + 0.5 * mj * (vj[0] * vj[0] + vj[1] * vj[1]);
    double ke_lost = ke_before - ke_after;
    double heat = ke_lost - pe_change;
    if (heat > 0) {
        s->temp[i] += (heat * 0.5) / mi;
        s->temp[j] += (heat * 0.5) / mj;
    }
    *pe_o = pe_change; *ke_o = ke_lost; *heat_o = heat;
    return 1;
}

static void candidate(orc_system *s, int32_t a, int32_t b, double *tpe, double *tke, double *th)
{
    double pe, ke, heat;
    s->n_candidates++;
    if (resolve(s, a, b, &pe, &ke, &heat)) {
        if (s->log_ij && s->log_n < s->log_cap) {
            s->log_ij[2 * s->log_n] = a; s->log_ij[2 * s->log_n + 1] = b;
            s->log_vals[3 * s->log_n] = pe; s->log_vals[3 * s->log_n + 1] = ke;
            s->log_vals[3 * s->log_n + 2] = heat;
        }
        s->log_n++;
        s->n_contacts++;
    }
    *tpe += pe; *tke += ke;
    if (heat > 0) *th += heat;
}

/* particle_system.py:137-198 _handle_collisions_jit: one Gauss-Seidel sweep */
static ORC_FAST void collision_sweep(orc_system *s, double *tpe, double *tke, double *th)
{
    int gw = s->grid_w, gh = s->grid_h;
    const int32_t *off = s->grid_offsets, *idx = s->grid_indices;
    *tpe = 0.0; *tke = 0.0; *th = 0.0;
    for (int64_t cell = 0; cell < (int64_t)gw * gh; cell++) {
        int cx = (int)(cell % gw), cy = (int)(cell / gw);
        int32_t st = off[cell], en = off[cell + 1];
        for (int32_t a = st; a < en; a++)
            for (int32_t b = a + 1; b < en; b++)
                candidate(s, idx[a], idx[b], tpe, tke, th);
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (dx == 0 && dy == 0) continue;
                int nx = cx + dx, ny = cy + dy;
                if (nx < 0 || nx >= gw || ny < 0 || ny >= gh) continue;
                int64_t nc = (int64_t)ny * gw + nx;
                int32_t ns = off[nc], ne = off[nc + 1];
                for (int32_t a = st; a < en; a++) {
                    int32_t p1 = idx[a];
                    for (int32_t b = ns; b < ne; b++) {
                        int32_t p2 = idx[b];
                        if (p1 < p2) candidate(s, p1, p2, tpe, tke, th);
                    }
                }
            }
    }
}

/* particle_system.py:439-474 _handle_collisions */
static void collisions(orc_system *s)
{
    double tpe = 0.0, tke = 0.0, th = 0.0;
    s->n_candidates = 0; s->n_contacts = 0; s->log_n = 0;
    for (int it = 0; it < s->solver_iterations; it++) {
        double a, b, c;
        collision_sweep(s, &a, &b, &c);
        tpe += a; tke += b; th += c;
    }
    s->pe_gain = tpe; s->ke_loss = tke; s->heat_generated = th;
}

/* ---------------------------------------------------------------- conduction */
/* particle_system.py:312-336 _transfer_heat_between_pair_jit */
static inline ORC_FAST void heat_pair(orc_system *s, int32_t a, int32_t b, double *delta)
{
    double dx = s->pos[2 * b] - s->pos[2 * a], dy = s->pos[2 * b + 1] - s->pos[2 * a + 1];
    double d2 = dx * dx + dy * dy;
    int64_t cd = s->radii[a] + s->radii[b];
    if (d2 < (double)(cd * cd)) {
        double diff = s->temp[a] - s->temp[b];
        if (diff != 0) {
            double e = s->heat_coeff * diff;
            delta[a] -= e / s->mass[a];
            delta[b] += e / s->mass[b];
        }
    }
}

/* particle_system.py:476-498 + :271-310 */
static ORC_FAST void conduction(orc_system *s)
{
    int gw = s->grid_w, gh = s->grid_h;
    const int32_t *off = s->grid_offsets, *idx = s->grid_indices;
    double *delta = s->delta_temps;
    memset(delta, 0, sizeof(double) * s->n);
    for (int64_t cell = 0; cell < (int64_t)gw * gh; cell++) {
        int cx = (int)(cell % gw), cy = (int)(cell / gw);
        int32_t st = off[cell], en = off[cell + 1];
        for (int32_t a = st; a < en; a++)
            for (int32_t b = a + 1; b < en; b++) heat_pair(s, idx[a], idx[b], delta);
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (dx == 0 && dy == 0) continue;
                int nx = cx + dx, ny = cy + dy;
                if (nx < 0 || nx >= gw || ny < 0 || ny >= gh) continue;
                int64_t nc = (int64_t)ny * gw + nx;
                int32_t ns = off[nc], ne = off[nc + 1];
                for (int32_t a = st; a < en; a++) {
                    int32_t p1 = idx[a];
                    for (int32_t b = ns; b < ne; b++)
                        if (p1 < idx[b]) heat_pair(s, p1, idx[b], delta);
                }
            }
    }
    for (int64_t i = 0; i < s->n; i++) s->temp[i] += delta[i];
}

/* ---------------------------------------------------------------- walls */
/* particle_system.py:500-516 _check_boundary_collisions: masks first, then writes */
static void walls(orc_system *s)
{
    for (int64_t i = 0; i < s->n; i++) {
        double r = (double)s->radii[i];
        for (int d = 0; d < 2; d++) {
            double bound = d == 0 ? s->bounds_w : s->bounds_h;
            double x = s->pos[2 * i + d];
            int lo = (x - r) < 0, hi = (x + r) > bound;
            if (lo) s->pos[2 * i + d] = r;
            if (hi) s->pos[2 * i + d] = bound - r;
            if (lo || hi) s->vel[2 * i + d] *= -1;
        }
    }
}

/* ---------------------------------------------------------------- explosions */
/* particle_system.py:651-730 _handle_explosions + :626-649 _get_neighbors_from_grid.
   The reference takes neighbours from a Python set, whose order is not defined;
   here they are taken in grid order (cells dy,dx ascending, slots ascending). */
static void explosions(orc_system *s)
{
    int64_t n = s->n;
    s->n_exploded = 0;
    uint8_t *boom = (uint8_t *)calloc(n > 0 ? n : 1, 1);
    for (int64_t i = 0; i < n; i++)
        if (s->temp[i] >= s->explode_temp) { boom[i] = 1; s->exploded[s->n_exploded++] = (int32_t)i; }
    if (s->n_exploded == 0) { free(boom); return; }
    int32_t *nb = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    double *w = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    double cs = (double)s->cell_size;
    for (int64_t e = 0; e < s->n_exploded; e++) {
        int32_t p = s->exploded[e];
        double energy = (s->mass[p] * s->temp[p]) * s->explosion_eff;
        /* Python int() truncates toward zero */
        int64_t cx = (int64_t)(s->pos[2 * p] / cs), cy = (int64_t)(s->pos[2 * p + 1] / cs);
        int64_t cnt = 0;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                int64_t x = cx + dx, y = cy + dy;
                if (x < 0 || x >= s->grid_w || y < 0 || y >= s->grid_h) continue;
                int64_t c = y * s->grid_w + x;
                for (int32_t k = s->grid_offsets[c]; k < s->grid_offsets[c + 1]; k++) {
                    int32_t q = s->grid_indices[k];
                    if (q != p && !boom[q]) nb[cnt++] = q;
                }
            }
        if (cnt == 0) continue;
        double total = 0.0;
        for (int64_t k = 0; k < cnt; k++) {
            int32_t q = nb[k];
            double dx = s->pos[2 * q] - s->pos[2 * p], dy = s->pos[2 * q + 1] - s->pos[2 * p + 1];
            double d2 = dx * dx + dy * dy;
            if (d2 == 0) d2 = 1e-6;
            w[k] = 1.0 / d2;
            total += w[k];
        }
        for (int64_t k = 0; k < cnt; k++) {
            int32_t q = nb[k];
            double dx = s->pos[2 * q] - s->pos[2 * p], dy = s->pos[2 * q + 1] - s->pos[2 * p + 1];
            double d2 = dx * dx + dy * dy;
            if (d2 == 0) d2 = 1e-6;
            double share = energy * (w[k] / total);
            double dv = sqrt(2 * share / s->mass[q]);
            double dist = sqrt(d2);
            s->vel[2 * q] += (dx / dist) * dv;
            s->vel[2 * q + 1] += (dy / dist) * dv;
        }
    }
    /* boolean-mask compaction, order preserved (particle_system.py:715-725) */
    int64_t o = 0;
    for (int64_t i = 0; i < n; i++) {
        if (boom[i]) continue;
        if (o != i) {
            s->pos[2 * o] = s->pos[2 * i]; s->pos[2 * o + 1] = s->pos[2 * i + 1];
            s->vel[2 * o] = s->vel[2 * i]; s->vel[2 * o + 1] = s->vel[2 * i + 1];
            s->acc[2 * o] = s->acc[2 * i]; s->acc[2 * o + 1] = s->acc[2 * i + 1];
            s->mass[o] = s->mass[i]; s->inv_mass[o] = s->inv_mass[i];
            s->temp[o] = s->temp[i]; s->radii[o] = s->radii[i];
        }
        o++;
    }
    s->n = o;
    free(boom); free(nb); free(w);
}

/* ================================================================ public */
orc_system *orc_create(int64_t n, const double *cfg, int solver_iterations, int cell_size,
                       int grid_w, int grid_h)
{
    /* cfg: min_mass,max_mass,G,softening,theta,heat_coeff,restitution,pos_correct,
            explosion_eff,damping,explode_temp,bounds_w,bounds_h */
    orc_system *s = (orc_system *)calloc(1, sizeof(orc_system));
    s->min_mass = cfg[0]; s->max_mass = cfg[1]; s->gravity_constant = cfg[2];
    s->softening = cfg[3]; s->theta = cfg[4]; s->heat_coeff = cfg[5];
    s->restitution = cfg[6]; s->pos_correct = cfg[7]; s->explosion_eff = cfg[8];
    s->damping = cfg[9]; s->explode_temp = cfg[10]; s->bounds_w = cfg[11]; s->bounds_h = cfg[12];
    s->solver_iterations = solver_iterations;
    s->cell_size = cell_size; s->grid_w = grid_w; s->grid_h = grid_h;
    s->n = n; s->cap = n;
    int64_t c = n > 0 ? n : 1;
    s->pos = (double *)calloc(2 * c, sizeof(double));
    s->vel = (double *)calloc(2 * c, sizeof(double));
    s->acc = (double *)calloc(2 * c, sizeof(double));
    s->mass = (double *)calloc(c, sizeof(double));
    s->inv_mass = (double *)calloc(c, sizeof(double));
    s->temp = (double *)calloc(c, sizeof(double));
    s->radii = (int64_t *)calloc(c, sizeof(int64_t));
    s->delta_temps = (double *)calloc(c, sizeof(double));
    s->exploded = (int32_t *)calloc(c, sizeof(int32_t));
    s->grid_offsets = (int32_t *)calloc((int64_t)grid_w * grid_h + 1, sizeof(int32_t));
    s->grid_indices = (int32_t *)calloc(c, sizeof(int32_t));
    s->max_nodes = 4 * n + 1;                 /* quadtree.py:165 */
    s->node_bounds = (double *)calloc(4 * s->max_nodes, sizeof(double));
    s->node_data = (double *)calloc(4 * s->max_nodes, sizeof(double));
    s->node_children = (int32_t *)calloc(4 * s->max_nodes, sizeof(int32_t));
    s->node_is_leaf = (uint8_t *)calloc(s->max_nodes, 1);
    s->node_particle = (int32_t *)calloc(s->max_nodes, sizeof(int32_t));
    s->node_map = (int32_t *)calloc(3 * s->max_nodes, sizeof(int32_t));
    s->particle_list = (int32_t *)calloc(c, sizeof(int32_t));
    return s;
}

void orc_destroy(orc_system *s)
{
    if (!s) return;
    free(s->pos); free(s->vel); free(s->acc); free(s->mass); free(s->inv_mass); free(s->temp);
    free(s->radii); free(s->delta_temps); free(s->exploded); free(s->grid_offsets);
    free(s->grid_indices); free(s->node_bounds); free(s->node_data); free(s->node_children);
    free(s->node_is_leaf); free(s->node_particle); free(s->node_map); free(s->particle_list);
    free(s->log_ij); free(s->log_vals);
    free(s);
}

void orc_set_fast_pow(orc_system *s, int on) { s->fast_pow = on; }
void orc_set_threads(orc_system *s, int k) { s->threads = k; }

void orc_enable_contact_log(orc_system *s, int64_t cap)
{
    free(s->log_ij); free(s->log_vals);
    s->log_cap = cap;
    s->log_ij = (int32_t *)calloc(2 * (cap > 0 ? cap : 1), sizeof(int32_t));
    s->log_vals = (double *)calloc(3 * (cap > 0 ? cap : 1), sizeof(double));
}

/* field access for the ctypes wrapper: id -> pointer */
void *orc_ptr(orc_system *s, int which)
{
    switch (which) {
    case 0: return s->pos;      case 1: return s->vel;      case 2: return s->acc;
    case 3: return s->mass;     case 4: return s->inv_mass; case 5: return s->temp;
    case 6: return s->radii;    case 7: return s->grid_offsets; case 8: return s->grid_indices;
    case 9: return s->node_bounds; case 10: return s->node_data; case 11: return s->node_children;
    case 12: return s->node_is_leaf; case 13: return s->node_particle; case 14: return s->node_map;
    case 15: return s->particle_list; case 16: return s->delta_temps; case 17: return s->exploded;
    case 18: return s->log_ij;  case 19: return s->log_vals;
    default: return 0;
    }
}

int64_t orc_get_i(orc_system *s, int which)
{
    switch (which) {
    case 0: return s->n;        case 1: return s->num_nodes; case 2: return s->n_candidates;
    case 3: return s->n_contacts; case 4: return s->n_exploded; case 5: return s->log_n;
    case 6: return s->grid_n;   case 7: return s->max_nodes;
    default: return -1;
    }
}

void orc_get_collision_totals(orc_system *s, double *out)
{
    out[0] = s->pe_gain; out[1] = s->ke_loss; out[2] = s->heat_generated;
}

/* stage entry points (each mirrors one method of the reference class) */
int orc_build_tree(orc_system *s) { return build_tree(s); }
void orc_build_grid(orc_system *s) { build_grid(s); }
void orc_gravity(orc_system *s, int64_t *visits, int64_t *inter) { gravity(s, visits, inter); }
void orc_collisions(orc_system *s) { collisions(s); }
void orc_conduction(orc_system *s) { conduction(s); }
void orc_walls(orc_system *s) { walls(s); }
void orc_explosions(orc_system *s) { explosions(s); }

double orc_potential_energy(orc_system *s)
{
    /* particle_system.py:785-804 */
    if (s->num_nodes == 0 || !s->tree_cached) return 0.0;
    return potential(s, 0);
}

/* particle_system.py:518-563 update().  stage_mask bit0: collisions, bit1:
   conduction, bit2: cooling, bit3: walls, bit4: explosions (all set = the
   reference; gravity+integration always run).  Returns 0, or -1 when the
   node pool overflowed. */
int orc_update(orc_system *s, int stage_mask)
{
    int64_t n = s->n;
    double *old = (double *)malloc(sizeof(double) * 2 * (n > 0 ? n : 1));
    for (int64_t i = 0; i < 2 * n; i++) {
        s->pos[i] += s->vel[i] + 0.5 * s->acc[i];          /* :528 */
        old[i] = s->acc[i];                                /* :529 */
    }
    build_grid(s);                                         /* :533 */
    if (build_tree(s) != 0) { free(old); return -1; }      /* :534 */
    gravity(s, 0, 0);                                      /* :537 */
    for (int64_t i = 0; i < 2 * n; i++) s->vel[i] += 0.5 * (old[i] + s->acc[i]);   /* :540 */
    free(old);
    if (stage_mask & 1) collisions(s);                     /* :546 */
    else { s->pe_gain = s->ke_loss = s->heat_generated = 0.0; s->n_candidates = s->n_contacts = 0; }
    if (stage_mask & 2) conduction(s);                     /* :549 */
    if ((stage_mask & 4) && s->damping > 0) {              /* :555-557 */
        double f = 1.0 - s->damping;
        for (int64_t i = 0; i < n; i++) s->temp[i] *= f;
    }
    if (stage_mask & 8) walls(s);                          /* :560 */
    s->n_exploded = 0;
    if (stage_mask & 16) explosions(s);                    /* :563 */
    return 0;
}

/*
 * zrk_oracle.c -- CPU restatement of the reference's L1 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under zrk_modulation_amd/ may import,
 * link or call this file.  It is the checker for tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py, never the thing that is shipped.
 *
 * Parity status: the reference (Ollegorii/ZRK_modulation, pure Python + numpy)
 * holds no golden vectors for this path, so the restatement is pinned against
 * outputs of the reference itself, captured in this container by
 * tests/golden/gen_golden.py and committed under tests/golden/ (see DESIGN.md).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).  All arithmetic is IEEE binary64, compiled with
 * -ffp-contract=off so that the only fused operations are the explicit fma()
 * calls that model OpenBLAS ddot for n=3.
 *
 * Entity table layout (structure of arrays, same as the device side):
 *   a "vec3 plane set" is double[3*cap]; component c of slot i is p[c*cap + i].
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ZO_API __attribute__((visibility("default")))

/* numpy.degrees(x) = x * (180/pi), the quotient folded in binary64. */
static const double ZO_RAD2DEG = 180.0 / 3.14159265358979323846;

/* np.dot for two 3-vectors as OpenBLAS ddot rounds it: fma(z,z', fma(y,y', x*x')).
 * Used by np.linalg.norm (modules/Radar.py:56, modules/Missile.py:186,
 * modules/AirObject.py:35-36) and np.dot (modules/Missile.py:65-67). */
static inline double zo_dot3(double ax, double ay, double az, double bx, double by, double bz)
{
    return fma(az, bz, fma(ay, by, ax * bx));
}

ZO_API double zo_norm3(double x, double y, double z)
{
    return sqrt(zo_dot3(x, y, z, x, y, z));
}

/* numpy / CPython float floor-mod:  a % b  (modules/Radar.py:62-63, :103, :109, :114, :117).
 * fmod, then shift into the sign of b; an exact zero takes the sign of b. */
ZO_API double zo_floormod(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0) {               /* NaN != 0 is true, and the branch below is then a no-op */
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* AirObject.__init__ (modules/AirObject.py:35-36): unit velocity and speed_mod. */
ZO_API void zo_unit_velocity(const double v[3], double unit[3], double *speed_mod)
{
    double n = zo_norm3(v[0], v[1], v[2]);
    unit[0] = v[0] / n; unit[1] = v[1] / n; unit[2] = v[2] / n;
    *speed_mod = n;
}

/* Trajectory.get_pos (modules/AirObject.py:23-25): s + v*(t - t0), three roundings. */
static inline void zo_get_pos(const double *sp, const double *vel, const double *t0,
                              int64_t cap, int64_t i, double t, double out[3])
{
    double d = t - t0[i];
    for (int c = 0; c < 3; ++c) {
        double step = vel[c * cap + i] * d;
        out[c] = sp[c * cap + i] + step;
    }
}

/*
 * AirEnv.step() entity loop (modules/AirEnv.py:45-48), literally in slot order:
 *   Target.step -> AirObject.step (modules/utils.py:38-39, modules/AirObject.py:39-42)
 *   Missile.step 'active' branch (modules/Missile.py:162-193; the re-aim loop
 *   :163-179 never iterates).
 *
 * kind[i]      0 = target, 1 = missile
 * mrow[i]      row of slot i in the missile table (-1 for targets)
 * m_status     1 = active, 2 = detonated (0 = not yet in AirEnv; never stepped here)
 * prev/pos     prev receives the position each live slot held before this step
 *              (AirObject.py:41); prev_valid[i] = 0 where the reference sets None.
 * events       (missile slot, target slot or -1, self_detonation) in message order.
 * returns      number of events.
 */
ZO_API int64_t zo_airenv_step(int64_t n, int64_t cap, int64_t time_ms, int64_t dt_ms,
                              const double *sp, const double *vel, const double *t0,
                              const uint8_t *alive, const uint8_t *kind, const int32_t *mrow,
                              double *pos, double *prev, uint8_t *prev_valid,
                              const int32_t *m_tgt, const double *m_radius, double *m_period,
                              uint8_t *m_status,
                              int32_t *ev_missile, int32_t *ev_target, uint8_t *ev_self)
{
    double t = (double)time_ms / 1000.0;         /* to_seconds, modules/AirObject.py:5-7 */
    double dts = (double)dt_ms / 1000.0;
    int64_t nev = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!alive[i]) continue;                 /* tombstone, AirEnv.py:46-47 */
        if (kind[i] == 1) {
            int32_t r = mrow[i];
            if (m_status[r] != 1) continue;      /* 'detonated': pass (Missile.py:195-196) */
        }
        double p[3];
        for (int c = 0; c < 3; ++c) prev[c * cap + i] = pos[c * cap + i];
        prev_valid[i] = (t0[i] != t);            /* AirObject.py:41 */
        zo_get_pos(sp, vel, t0, cap, i, t, p);
        for (int c = 0; c < 3; ++c) pos[c * cap + i] = p[c];
        if (kind[i] == 1) {
            int32_t r = mrow[i];
            int32_t j = m_tgt[r];
            double dx = pos[0 * cap + j] - p[0];   /* target.pos - self.pos, Missile.py:186 */
            double dy = pos[1 * cap + j] - p[1];
            double dz = pos[2 * cap + j] - p[2];
            double dist = zo_norm3(dx, dy, dz);
            if (dist <= m_radius[r]) {             /* Missile.py:187-189 */
                ev_missile[nev] = (int32_t)i; ev_target[nev] = j; ev_self[nev] = 0; ++nev;
                m_status[r] = 2;
                continue;                          /* return before the period decrement */
            }
            m_period[r] -= dts;                    /* Missile.py:191 */
            if (m_period[r] <= 0.0) {              /* Missile.py:192-193 */
                ev_missile[nev] = (int32_t)i; ev_target[nev] = -1; ev_self[nev] = 1; ++nev;
                m_status[r] = 2;
            }
        }
    }
    return nev;
}

/*
 * The same step on `threads` host cores, for the all-core CPU baseline of bench.py (the reference's own loop is one
 * thread by construction, modules/Manager.py:130-131; this is what a host port of L1 could do at best).
 * Targets do not depend on one another within a tick: phase 1 advances every live target (kind 0) in parallel.  Missiles
 * keep the list order (phase 2, sequential: their events are ordered and a missile may chase a missile).  What the
 * literal loop gives a missile whose target stands BEHIND it in the list -- the target's position of the tick before,
 * because the target has not stepped yet -- is what phase 1 left in `prev`.  Results identical to zo_airenv_step
 * (tests/test_oracle_golden.py).
 */
ZO_API int64_t zo_airenv_step_mt(int64_t n, int64_t cap, int64_t time_ms, int64_t dt_ms,
                                 const double *sp, const double *vel, const double *t0,
                                 const uint8_t *alive, const uint8_t *kind, const int32_t *mrow,
                                 double *pos, double *prev, uint8_t *prev_valid,
                                 const int32_t *m_tgt, const double *m_radius, double *m_period,
                                 uint8_t *m_status,
                                 int32_t *ev_missile, int32_t *ev_target, uint8_t *ev_self, int threads)
{
    double t = (double)time_ms / 1000.0;
    double dts = (double)dt_ms / 1000.0;
    int64_t nev = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        if (!alive[i] || kind[i] != 0) continue;
        double p[3];
        for (int c = 0; c < 3; ++c) prev[c * cap + i] = pos[c * cap + i];
        prev_valid[i] = (t0[i] != t);
        zo_get_pos(sp, vel, t0, cap, i, t, p);
        for (int c = 0; c < 3; ++c) pos[c * cap + i] = p[c];
    }
    for (int64_t i = 0; i < n; ++i) {
        if (!alive[i] || kind[i] != 1) continue;
        int32_t r = mrow[i];
        if (m_status[r] != 1) continue;
        double p[3];
        for (int c = 0; c < 3; ++c) prev[c * cap + i] = pos[c * cap + i];
        prev_valid[i] = (t0[i] != t);
        zo_get_pos(sp, vel, t0, cap, i, t, p);
        for (int c = 0; c < 3; ++c) pos[c * cap + i] = p[c];
        int32_t j = m_tgt[r];
        /* a live target behind the missile in the list has not stepped yet in the literal loop */
        const double *tp = (kind[j] == 0 && j > i && alive[j]) ? prev : pos;
        double dx = tp[0 * cap + j] - p[0], dy = tp[1 * cap + j] - p[1], dz = tp[2 * cap + j] - p[2];
        double dist = zo_norm3(dx, dy, dz);
        if (dist <= m_radius[r]) {
            ev_missile[nev] = (int32_t)i; ev_target[nev] = j; ev_self[nev] = 0; ++nev;
            m_status[r] = 2;
            continue;
        }
        m_period[r] -= dts;
        if (m_period[r] <= 0.0) {
            ev_missile[nev] = (int32_t)i; ev_target[nev] = -1; ev_self[nev] = 1; ++nev;
            m_status[r] = 2;
        }
    }
    return nev;
}

/* Radar parameter block: 8 doubles per radar, the fields find_visible_objects reads. */
typedef struct {
    double px, py, pz;       /* self.pos                    */
    double max_distance;     /* self.max_distance           */
    double caz, az_range;    /* current_azimuth, azimuth_range   */
    double cel, el_range;    /* current_elevation, elevation_range */
} zo_radar;

/* One (radar, entity) visibility decision: SectorRadar.find_visible_objects body,
 * modules/Radar.py:55-71. */
static inline int zo_visible(const zo_radar *rd, double x, double y, double z)
{
    double dx = x - rd->px, dy = y - rd->py, dz = z - rd->pz;
    double dist = zo_norm3(dx, dy, dz);                          /* :56 */
    if (dist > rd->max_distance) return 0;                       /* :57 */
    double az = zo_floormod(atan2(dy, dx) * ZO_RAD2DEG, 360.0);  /* :62 */
    double el = zo_floormod(asin(dz / dist) * ZO_RAD2DEG, 180.0);/* :63 */
    double az_hi = rd->caz + rd->az_range;
    double el_hi = rd->cel + rd->el_range;
    return (rd->caz <= az) && (az <= az_hi) && (rd->cel <= el) && (el <= el_hi);  /* :67-70 */
}

/* SectorRadar.find_visible_objects (modules/Radar.py:44-73) over the live slots,
 * in slot order.  Writes the ordered slot list; returns its length. */
ZO_API int64_t zo_radar_sweep(int64_t n, int64_t cap, const double *pos, const uint8_t *alive,
                              const zo_radar *rd, int32_t *out_idx)
{
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!alive[i]) continue;
        if (zo_visible(rd, pos[i], pos[cap + i], pos[2 * cap + i])) out_idx[k++] = (int32_t)i;
    }
    return k;
}

/* SectorRadar.smooth_objects (modules/Radar.py:138-142): pos += noise, in list order.
 * noise is k x 3 row-major, the values np.random.normal(0, 5, (k, 3)) returned. */
ZO_API void zo_noise_apply(int64_t k, const int32_t *idx, const double *noise, int64_t cap, double *pos)
{
    for (int64_t j = 0; j < k; ++j) {
        int64_t i = idx[j];
        for (int c = 0; c < 3; ++c) pos[c * cap + i] += noise[3 * j + c];
    }
}

/* SectorRadar.move_to_next_sector_circular (modules/Radar.py:96-117), quirks kept:
 * azimuth wraps to elevation_start (:105); anything but the two mode strings is a no-op.
 * mode: 0 = "horizontal", 1 = "vertical", other = unknown string. */
ZO_API void zo_scan_next(int mode, double az_range, double az_speed, double el_speed,
                         double el_start, double *caz, double *cel)
{
    if (mode == 0) {
        if (*caz + az_range < 360.0) *caz = zo_floormod(*caz + az_speed, 360.0);
        else *caz = el_start;
        if (*caz < az_speed) {
            if (*cel + el_speed < 90.0) *cel = zo_floormod(*cel + el_speed, 90.0);
            else *cel = el_start;
        }
    } else if (mode == 1) {
        *cel = zo_floormod(*cel + el_speed, 90.0);
        if (*cel < el_speed) *caz = zo_floormod(*caz + az_speed, 360.0);
    }
}

/*
 * Missile._calculate_trajectory_params (modules/Missile.py:35-102).
 * tp/mp: target and missile positions; tvu/tsm: target.velocity (unit) and
 * target.speed_mod; v0: missile speed; period: detonate_period.
 * Returns 0 and fills V[3], *t_hit on success; otherwise the index (1..6) of the
 * ValueError raised, in source order (:73, :78, :82, :89, :94).
 * b**2 and v0**2 are restated as exact squares (numpy scalar pow(b, 2)); see DESIGN.md.
 */
ZO_API int zo_launch_solve(const double tp[3], const double mp[3], const double tvu[3], double tsm,
                           double v0, double period, double V[3], double *t_hit)
{
    double d[3], vt[3];
    for (int c = 0; c < 3; ++c) { d[c] = tp[c] - mp[c]; vt[c] = tvu[c] * tsm; }   /* :55, :58 */
    double a = zo_dot3(vt[0], vt[1], vt[2], vt[0], vt[1], vt[2]) - v0 * v0;        /* :65 */
    double b = 2.0 * zo_dot3(d[0], d[1], d[2], vt[0], vt[1], vt[2]);               /* :66 */
    double c = zo_dot3(d[0], d[1], d[2], d[0], d[1], d[2]);                        /* :67 */
    double t;
    if (fabs(a) < 1e-6) {                                                          /* :70 */
        if (fabs(b) < 1e-6) return 1;                                              /* :72-74 */
        t = -c / b;                                                                /* :75 */
        if (t <= 0.0) return 2;                                                    /* :77-78 */
    } else {
        double disc = b * b - 4.0 * a * c;                                         /* :80 */
        if (disc < 0.0) return 3;                                                  /* :81-82 */
        double sq = sqrt(disc);
        double t1 = (-b + sq) / (2.0 * a);                                         /* :84 */
        double t2 = (-b - sq) / (2.0 * a);                                         /* :85 */
        int have = 0; t = 0.0;
        if (t1 > 0.0) { t = t1; have = 1; }                                        /* :87 */
        if (t2 > 0.0) { if (!have || t2 < t) t = t2; have = 1; }                   /* :90 min() */
        if (!have) return 4;                                                       /* :88-89 */
    }
    if (t > period) return 5;                                                      /* :92-94 */
    double W[3];
    for (int k = 0; k < 3; ++k) W[k] = d[k] / t + vt[k];                           /* :97 */
    double n = zo_norm3(W[0], W[1], W[2]);
    for (int k = 0; k < 3; ++k) V[k] = W[k] / n * v0;                              /* :100 */
    *t_hit = t;
    return 0;
}

/* ------------------------------------------------------------------------- *
 * Counter-based noise of the throughput mode (no counterpart in the reference,
 * which draws from numpy's global MT19937 stream).
 * Philox4x32-10 (Salmon et al., SC'11) keyed by seed, counter (entity, tick), seeds a
 * xoshiro128++ stream per entity and tick.  The device computes the same integers;
 * its log/sin/cos are hardware approximations, so values agree to ~1e-5 absolute,
 * not bitwise (tests feed the device's own values back through noise tables).
 * ------------------------------------------------------------------------- */
static inline void zo_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                    uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Noise stream of one (entity, tick): Philox block -> xoshiro128 state; each detection advances the state once
 * and takes two scrambled words from it = four 16-bit uniforms = two Box-Muller pairs in binary32, three values
 * used.  r = 5 sqrt(-2 ln((h + 0.5) / 65536)) written as sqrt(C1 * log2(h + 0.5) + C0), as the device evaluates it. */
typedef struct { uint32_t s[4]; } zo_noise_state;

static inline zo_noise_state zo_noise_init(uint64_t seed, uint64_t tick, uint64_t entity)
{
    zo_noise_state st;
    zo_philox4x32_10((uint32_t)entity, (uint32_t)(entity >> 32), (uint32_t)tick, (uint32_t)(tick >> 32),
                     (uint32_t)seed, (uint32_t)(seed >> 32), st.s);
    return st;
}

static inline uint32_t zo_rotl(uint32_t v, int k) { return (v << k) | (v >> (32 - k)); }

static inline void zo_noise_next2(zo_noise_state *st, uint32_t *a, uint32_t *b)
{
    uint32_t *s = st->s;
    *a = zo_rotl(s[0] + s[3], 7) + s[0];
    *b = zo_rotl(s[1] + s[2], 13) + s[2];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = zo_rotl(s[3], 11);
}

static inline void zo_noise_draw3(zo_noise_state *st, double out[3])
{
    uint32_t a, b;
    zo_noise_next2(st, &a, &b);
    const float k16 = 1.52587890625e-5f;
    const float c1 = -34.657359027997266f, c0 = 554.51774444795626f;
    float l0 = log2f((float)(a >> 16) + 0.5f), u1 = (float)(a & 0xFFFFu) * k16;
    float l1 = log2f((float)(b >> 16) + 0.5f), u3 = (float)(b & 0xFFFFu) * k16;
    float r0 = sqrtf(fmaf(l0, c1, c0));
    float r1 = sqrtf(fmaf(l1, c1, c0));
    const float two_pi = 6.283185307179586f;
    out[0] = (double)(r0 * cosf(two_pi * u1));
    out[1] = (double)(r0 * sinf(two_pi * u1));
    out[2] = (double)(r1 * cosf(two_pi * u3));
}

/* The triple the `ordinal`-th detection (0-based) of `entity` draws in `tick`. */
ZO_API void zo_philox_noise(uint64_t seed, uint64_t tick, uint32_t ordinal, uint64_t entity, double out[3])
{
    zo_noise_state st = zo_noise_init(seed, tick, entity);
    out[0] = out[1] = out[2] = 0.0;
    for (uint32_t k = 0; k <= ordinal; ++k) zo_noise_draw3(&st, out);
}

/*
 * Full L1 radar phase of one tick, entity-major ("fused") form, for noise modes in
 * which entities are independent (hard part 3):
 *   mode 0  no noise
 *   mode 1  counter-based noise (zo_noise_*), sigma = 5
 *   mode 2  noise table: table[(k*n + i)*3 + c] = what the k-th detection of slot i adds
 * Per live entity, radars in order: zo_visible on the current (already perturbed)
 * position, then pos += noise (modules/Radar.py:163-164 applied radar after radar).
 * vis_mask bit r = seen by radar r.  gid0 = global index of slot 0 (multi-GPU shards).
 * With threads > 1 the entity loop is split with OpenMP (result identical).
 */
ZO_API void zo_radar_phase_fused(int64_t n, int64_t cap, double *pos, const uint8_t *alive,
                                 int R, const zo_radar *radars, int mode, const double *table,
                                 uint64_t seed, uint64_t tick, int64_t gid0, uint32_t *vis_mask,
                                 int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        if (!alive[i]) { vis_mask[i] = 0; continue; }
        double x = pos[i], y = pos[cap + i], z = pos[2 * cap + i];
        uint32_t m = 0;
        int ordinal = 0;
        zo_noise_state ns;
        if (mode == 1) ns = zo_noise_init(seed, tick, (uint64_t)(gid0 + i));
        for (int r = 0; r < R; ++r) {
            if (!zo_visible(&radars[r], x, y, z)) continue;
            m |= 1u << r;
            if (mode == 1) {
                double nz[3];
                zo_noise_draw3(&ns, nz);
                x += nz[0]; y += nz[1]; z += nz[2];
            } else if (mode == 2) {
                const double *nz = table + ((int64_t)ordinal * n + i) * 3;
                x += nz[0]; y += nz[1]; z += nz[2];
            }
            ++ordinal;
        }
        pos[i] = x; pos[cap + i] = y; pos[2 * cap + i] = z;
        vis_mask[i] = m;
    }
}

/* Entity-major advance only (modules/AirObject.py:39-42 for every live slot), the
 * data-parallel half of zo_airenv_step; used with zo_radar_phase_fused for timing. */
ZO_API void zo_advance_all(int64_t n, int64_t cap, int64_t time_ms, const double *sp, const double *vel,
                           const double *t0, const uint8_t *alive, double *pos, double *prev, int threads)
{
    double t = (double)time_ms / 1000.0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        if (!alive[i]) continue;
        double p[3];
        zo_get_pos(sp, vel, t0, cap, i, t, p);
        for (int c = 0; c < 3; ++c) {
            if (prev) prev[c * cap + i] = pos[c * cap + i];
            pos[c * cap + i] = p[c];
        }
    }
}

/* Stable compaction of one radar's bit out of vis_mask: the order FoundObjectsMessage
 * lists objects in (modules/Radar.py:53, :71, :168-174). */
ZO_API int64_t zo_compact_bit(int64_t n, const uint32_t *vis_mask, int r, int32_t base, int32_t *out_idx)
{
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i)
        if ((vis_mask[i] >> r) & 1u) out_idx[k++] = base + (int32_t)i;
    return k;
}

/* ------------------------------------------------------------------------- *
 * CombatControlPoint.link_object for every detection of a tick, in the order
 * CombatControlPoint.step makes the calls (modules/CCP.py:171-219, :414-429):
 * the strictly nearest track -- scan order = index order: target tracks, then
 * missile tracks -- whose distance lies in [max(0, v (age - slack)),
 * max(0, v (age + slack))] (calc_range, :176-186), tracks updated "now"
 * skipped (:188, :203); a match updates the track (old_target / old_rocket,
 * :338-366), so it is out of every later detection's scan.  match[d] = track
 * index or -1 (NEW_TARGET).  np.linalg.norm as elsewhere: sqrt of the fma chain.
 * ------------------------------------------------------------------------- */
ZO_API void zo_ccp_link(int64_t D, const double *det_pos, const double *det_speed, int64_t T, const double *trk_ref,
                        const double *trk_upd_in, double now_s, double slack_s, int32_t *match, double *upd_scratch)
{
    for (int64_t t = 0; t < T; ++t) upd_scratch[t] = trk_upd_in[t];
    for (int64_t d = 0; d < D; ++d) {
        double best = INFINITY;
        int32_t m = -1;
        for (int64_t t = 0; t < T; ++t) {
            if (upd_scratch[t] == now_s) continue;
            double age = now_s - upd_scratch[t];
            double lo = det_speed[d] * (age - slack_s), hi = det_speed[d] * (age + slack_s);
            if (!(lo > 0.0)) lo = 0.0;
            if (!(hi > 0.0)) hi = 0.0;
            double dist = zo_norm3(trk_ref[3 * t] - det_pos[3 * d], trk_ref[3 * t + 1] - det_pos[3 * d + 1],
                                   trk_ref[3 * t + 2] - det_pos[3 * d + 2]);
            if (dist < best && lo <= dist && dist <= hi) { best = dist; m = (int32_t)t; }
        }
        match[d] = m;
        if (m >= 0) upd_scratch[m] = now_s;
    }
}

/* ------------------------------------------------------------------------- *
 * One tick of the command post's detection loop, literally:
 * CombatControlPoint.step (modules/CCP.py:406-429) over the FoundObjectsMessage
 * sequence -- skip ids already processed in this tick (:414), link_object
 * (:171-219), then new_target / old_target / old_rocket (:322-366) with
 * try_to_launch_missile (:287-320) -- on arrays.
 *
 * Objects are rows of a table (pos = obj.pos as the command post sees it, prev =
 * obj.prev_pos, prev_none[i] != 0 where the reference holds None, speed =
 * obj.speed_mod).  Tracks live in two arrays in dict order: target tracks
 * [0, *n_tt) and missile tracks [0, *n_tm); a track is keyed by the row whose id
 * created it (tt_key / tm_key, never changes: add_target on an existing key
 * replaces the entry IN PLACE, dict order kept, :92) and holds the row it last
 * matched (tt_obj: TargetCCP.target, the live handle whose prev_pos link_object
 * reads).  key_tt[row] = target track keyed by that row or -1.
 * Launchers in dict order: position, capacity, launched (in/out).
 *
 * Per processed detection (in order) out_obj / out_verdict (0 new, 1 old target,
 * 2 old missile) / out_match (track index within its array, -1) / out_launcher
 * (launcher index the request went to, -1).  Returns their number, or -1 - d if the
 * reference would have raised at sequence element d (a target track whose
 * handle has prev_pos None: `None - array`, :197).
 * The launcher distance is (np.sum(diff ** 2)) ** 0.5 (:297): pow(.., 0.5).
 * ------------------------------------------------------------------------- */
ZO_API int64_t zo_ccp_step(int64_t D, const int32_t *seq, int64_t cap, const double *pos, const double *prev,
                           const uint8_t *prev_none, const double *speed, double now_s, double slack_s,
                           int32_t *tt_key, int32_t *tt_obj, double *tt_upd, uint8_t *tt_follow, int64_t *n_tt,
                           const int32_t *tm_key, int32_t *tm_obj, double *tm_upd, int64_t *n_tm, int32_t *key_tt,
                           int64_t L, const double *l_pos, const int32_t *l_capacity, int32_t *l_launched,
                           uint8_t *processed /* [cap] scratch, zeroed here */,
                           int32_t *out_obj, int32_t *out_verdict, int32_t *out_match, int32_t *out_launcher)
{
    int64_t n_out = 0;
    memset(processed, 0, (size_t)cap);
    for (int64_t d = 0; d < D; ++d) {
        int32_t o = seq[d];
        if (processed[o]) continue;                               /* :414 */
        processed[o] = 1;
        double px = pos[o], py = pos[cap + o], pz = pos[2 * cap + o], v = speed[o];
        double best = INFINITY;
        int verdict = 0;
        int64_t m = -1;
        for (int64_t t = 0; t < *n_tt; ++t) {                     /* :193-204 */
            if (tt_upd[t] == now_s) continue;
            int32_t h = tt_obj[t];
            if (prev_none[h]) return -1 - d;                      /* target.prev_pos is None: the reference raises */
            double dist = zo_norm3(prev[h] - px, prev[cap + h] - py, prev[2 * cap + h] - pz);
            double age = now_s - tt_upd[t];
            double lo = v * (age - slack_s), hi = v * (age + slack_s);
            if (!(lo > 0.0)) lo = 0.0;
            if (!(hi > 0.0)) hi = 0.0;
            if (dist < best && lo <= dist && dist <= hi) { best = dist; verdict = 1; m = t; }
        }
        for (int64_t t = 0; t < *n_tm; ++t) {                     /* :207-218 */
            if (tm_upd[t] == now_s) continue;
            int32_t h = tm_obj[t];
            const double *ref = prev_none[h] ? pos : prev;        /* :211-213 */
            double dist = zo_norm3(ref[h] - px, ref[cap + h] - py, ref[2 * cap + h] - pz);
            double age = now_s - tm_upd[t];
            double lo = v * (age - slack_s), hi = v * (age + slack_s);
            if (!(lo > 0.0)) lo = 0.0;
            if (!(hi > 0.0)) hi = 0.0;
            if (dist < best && lo <= dist && dist <= hi) { best = dist; verdict = 2; m = t; }
        }
        int launcher = -1;
        int wants = verdict == 0 || (verdict == 1 && !tt_follow[m]);   /* :330, :342-343 */
        if (wants) {                                              /* try_to_launch_missile, :291-299 */
            double mind = INFINITY;
            for (int64_t l = 0; l < L; ++l) {
                if (l_launched[l] < l_capacity[l]) {
                    double ax = l_pos[3 * l] - px, ay = l_pos[3 * l + 1] - py, az = l_pos[3 * l + 2] - pz;
                    double dist = pow((ax * ax + ay * ay) + az * az, 0.5);
                    if (dist < mind) { mind = dist; launcher = (int)l; }
                }
            }
            if (launcher >= 0) l_launched[launcher] += 1;
        }
        if (verdict == 0) {                                       /* new_target -> add_target, :322-332, :88-93 */
            int64_t t = key_tt[o];
            if (t < 0) { t = (*n_tt)++; tt_key[t] = o; key_tt[o] = (int32_t)t; }
            tt_obj[t] = o; tt_upd[t] = now_s; tt_follow[t] = launcher >= 0;
        } else if (verdict == 1) {                                /* old_target, :334-361 */
            if (!tt_follow[m]) tt_follow[m] = launcher >= 0;
            tt_obj[m] = o; tt_upd[m] = now_s;
        } else {                                                  /* old_rocket, :363-366 (the dict key, tm_key[m], stays) */
            (void)tm_key;
            tm_obj[m] = o; tm_upd[m] = now_s;
        }
        out_obj[n_out] = o; out_verdict[n_out] = verdict; out_match[n_out] = (int32_t)m; out_launcher[n_out] = launcher;
        ++n_out;
    }
    return n_out;
}

/* check_if_missiles_launched -> add_missile (modules/CCP.py:160-169, :102-108): a missile of our own enters the
 * missile dict (keyed by its row; an existing key is replaced in place), updated "now". */
ZO_API void zo_ccp_add_missile(int32_t row, double now_s, int32_t *tm_key, int32_t *tm_obj, double *tm_upd, int64_t *n_tm)
{
    int64_t t = -1;
    for (int64_t k = 0; k < *n_tm; ++k) if (tm_key[k] == row) t = k;
    if (t < 0) { t = (*n_tm)++; tm_key[t] = row; }
    tm_obj[t] = row; tm_upd[t] = now_s;
}

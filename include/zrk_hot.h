/*
 * zrk_hot.h -- C ABI of libzrk_hot.so: the per-tick hot path of the ZRK simulator
 * (advance every air object, sweep every sector radar over them, compact the
 * detections, step every in-flight missile) as HIP kernels for gfx950 (MI355X).
 *
 * The reference (Ollegorii/ZRK_modulation) is pure Python and has no FFI; the entry
 * points below are what a binding for this path replaces, cited as reference
 * file:line (paths relative to the reference root).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - Every pointer marked DEVICE is HIP device memory owned by the caller; the
 *     library never allocates or frees across this boundary.  `workspace` is a
 *     caller-owned DEVICE scratch buffer of at least zrk_workspace_bytes() bytes that
 *     belongs to one context and one stream at a time; the library initialises it on
 *     first use and expects to find it as it left it (a buffer released and allocated
 *     again at the same address must be zero-filled before it comes back).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work
 *     is enqueued there and nothing synchronises unless stated.
 *   - One host thread per context.  Return value 0 = ok, negative = ZRK_E_*;
 *     zrk_last_error(ctx) gives the message of the last failure.
 *   - Vec3 arrays are structure-of-arrays plane sets: double[3*capacity], component
 *     c of slot i at p[c*capacity + i].  All arithmetic is IEEE binary64 with the
 *     rounding sequence of the reference ("Numerics").
 */
#ifndef ZRK_HOT_H
#define ZRK_HOT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZRK_ABI_VERSION 11
#define ZRK_MAX_RADARS 32           /* one bit per radar in the visibility mask */
#define ZRK_BLOCK 256               /* table rows per sweep workgroup */

#define ZRK_E_INVALID (-1)          /* bad argument (null pointer, size out of range) */
#define ZRK_E_HIP (-2)              /* a HIP call failed; see zrk_last_error */
#define ZRK_E_CAPACITY (-3)         /* an output buffer is too small */
#define ZRK_E_STATE (-4)            /* the workspace was modified behind the library's back; a bounded wait ran out
                                       (helper thread, peer rank, device); a collective went out poisoned */

/* flags of zrk_tick_sweep */
#define ZRK_F_ADVANCE 1u            /* recompute pos from the trajectory before sweeping */
#define ZRK_F_PHILOX 2u             /* counter-based measurement noise inside the sweep */
#define ZRK_F_EXACT_ONLY 4u         /* diagnostics: skip the float32 pre-classification */
#define ZRK_F_UNION_BITS 8u         /* zrk_run_ticks: write `packed` in the bitmap wire format (zrk_compact_bits) */

typedef struct zrk_ctx zrk_ctx;

/* Entity table: AirEnv's object list (modules/AirEnv.py:24) as SoA columns.
 * Rows are never reused.  Unless list_index says otherwise, row order is list order (targets, then
 * missiles as appended).  Everything that names an entity across this ABI (missile `slot` / `target`,
 * kill lists, launch requests, event rows) is a ROW; detection lists hold LIST indices. */
typedef struct {
    int64_t capacity;               /* plane stride of every vec3 column */
    const double *start_pos;        /* DEVICE [3][cap]  Trajectory.start_pos   modules/AirObject.py:20 */
    const double *velocity;         /* DEVICE [3][cap]  Trajectory.velocity    modules/AirObject.py:19 */
    const double *start_time;       /* DEVICE [cap]     Trajectory.start_time  modules/AirObject.py:21 */
    uint8_t *alive;                 /* DEVICE [cap]     0 = tombstone          modules/AirEnv.py:39-40 */
    const uint8_t *kind;            /* DEVICE [cap]     0 target, 1 missile */
    double *pos[2];                 /* DEVICE [3][cap] x2  obj.pos, double-buffered: pos[cur] is this tick's,
                                       pos[cur^1] holds what prev_pos aliases (modules/AirObject.py:41) */
    uint32_t *vis_mask;             /* DEVICE [cap]     bit r = seen by radar r this tick; indexed by LIST index */
    uint32_t *vis_mask_alt;         /* DEVICE [cap] or NULL: second mask buffer for zrk_run_ticks (see there) */
    const int32_t *list_index;      /* DEVICE [cap] or NULL.  NULL: row i of the table is element i of AirEnv's
                                       list.  Otherwise the rows may be stored in any order (e.g. spatially sorted,
                                       which makes waves coherent) and list_index[row] is the element's position in
                                       the list: detection lists, the noise key and the "already stepped this tick"
                                       rule of Missile.step follow the list, never the storage order. */
} zrk_entities;

/* The fields SectorRadar.find_visible_objects reads (modules/Radar.py:44-73). */
typedef struct {
    double pos[3];
    double max_distance;
    double cur_azimuth, azimuth_range;
    double cur_elevation, elevation_range;
} zrk_radar;

/* In-flight missiles, rows in the order they entered AirEnv (= slot order). */
typedef struct {
    int64_t capacity;
    const int32_t *slot;            /* DEVICE [cap] entity slot of the missile */
    const int32_t *target;          /* DEVICE [cap] entity slot of Missile.target */
    const double *radius;           /* DEVICE [cap] detonate_radius */
    double *period;                 /* DEVICE [cap] detonate_period, decremented in place */
    uint8_t *status;                /* DEVICE [cap] 1 active, 2 detonated */
    uint8_t *ev_code;               /* DEVICE [cap] scratch: 0 none, 1 hit, 2 timeout */
    int32_t *ev_missile;            /* DEVICE [cap] out: missile slots that detonated, list order */
    int32_t *ev_target;             /* DEVICE [cap] out: target slot, or -1 for self-detonation */
    int32_t *ev_count;              /* DEVICE [1]   out: number of events */
} zrk_missiles;

/* One pending Missile._launch (modules/Missile.py:104-133). */
typedef struct {
    int32_t target_slot;
    int32_t _pad;
    double missile_pos[3];
    double speed;                   /* velocity_module */
    double period;                  /* detonate_period */
    double radius;                  /* detonate_radius (used by zrk_launch_salvo only) */
} zrk_launch_req;

typedef struct {
    int32_t rc;                     /* 0 launched; 1..5 = which ValueError of modules/Missile.py:70-94; 6 no such target row */
    int32_t _pad;
    double velocity[3];             /* V, modules/Missile.py:97-100 */
    double t_hit;
} zrk_launch_res;

int zrk_abi_version(void);
int zrk_ctx_create(int device, zrk_ctx **out);
void zrk_ctx_destroy(zrk_ctx *ctx);
/* Diagnostics: the tuning variables (ZRK_SWEEP_ORDER, ZRK_DIAG, ZRK_COMPACT_*) are read when a context is created;
 * this reads them again. */
void zrk_ctx_reload_env(zrk_ctx *ctx);
/* zrk_run_ticks keeps, in the workspace, a bounding box per block of table rows between ticks (rows move along
 * straight lines, so an old box grown by the block's top speed still holds them), and, in memory of the context, a
 * 64-byte record of every row's trajectory and list index for the missile phase's gathers.  It notices rows appended
 * to the table by itself; a caller that rewrites the trajectory, the list index or the alive flag of EXISTING rows
 * says so here. */
void zrk_ctx_invalidate_boxes(zrk_ctx *ctx);
const char *zrk_last_error(zrk_ctx *ctx);

/* Bytes of DEVICE scratch the sweep + compaction need for a table of capacity n_max (zrk_run_ticks finds its
 * own arrays in the workspace from ents->capacity: size the workspace for exactly that). */
int64_t zrk_workspace_bytes(int64_t n_max);

/*
 * AirEnv.step()'s entity loop for non-missile work plus every SectorRadar's
 * find_visible_objects in one pass over the table.
 *   replaces: `for object in self.__objects: object.step()`   modules/AirEnv.py:45-48
 *             -> AirObject.step / Trajectory.get_pos          modules/AirObject.py:39-42, :23-25
 *             SectorRadar.find_visible_objects                modules/Radar.py:44-73
 *             SectorRadar.smooth_objects (ZRK_F_PHILOX only)  modules/Radar.py:138-142
 * For each live slot i < n: (ZRK_F_ADVANCE) pos[cur][i] = start_pos + velocity*(t - start_time)
 * with t = time_ms/1000; then radars 0..R-1 in order test the current position and, with
 * ZRK_F_PHILOX, every detection perturbs it by N(0, 5^2) per axis (a stream keyed by
 * (seed, tick, gid0 + i)) before the next radar looks.  vis_mask[i] gets bit r per detecting
 * radar (0 for dead slots).
 */
int zrk_tick_sweep(zrk_ctx *ctx, const zrk_entities *ents, int64_t n, int cur, int64_t time_ms,
                   const zrk_radar *radars /* HOST */, int R, uint32_t flags,
                   uint64_t seed, uint64_t tick, int64_t gid0, void *workspace, void *stream);

/*
 * Ordered detection lists: for each radar r the slots with bit r set, ascending, i.e. the
 * order of FoundObjectsMessage.visible_objects (modules/Radar.py:48-73, :168-174).
 * Radar r's list is det_idx[r*det_stride .. r*det_stride + det_cnt[r]), each entry base_index + slot;
 * det_cnt has R+1 entries, det_cnt[R] = number of slots seen by at least one radar.
 * Must follow zrk_tick_sweep on the same stream with the same n and R.
 * A list longer than det_stride is truncated, det_cnt[r] stays exact (det_cnt[r] > det_stride tells
 * the caller); det_stride = n can never truncate.
 * `packed` (optional) receives the union list used by the multi-GPU exchange: packed[0] = number
 * of slots seen by at least one radar, then for each such slot, ascending,
 * ((gid0 + slot) << 32) | vis_mask -- per-radar lists are stable filters of it.
 * One launch for tables up to about 8e6 rows (count, wait for the lower-numbered workgroups' counts,
 * scatter); three launches (count, scan, scatter) beyond.
 */
int zrk_compact(zrk_ctx *ctx, const uint32_t *vis_mask /* DEVICE */, int64_t n, int R,
                int32_t base_index, void *workspace, int32_t *det_idx /* DEVICE [R][det_stride], may be NULL */,
                int64_t det_stride, int32_t *det_cnt /* DEVICE [R+1] */,
                int64_t *packed /* DEVICE, may be NULL */, int64_t packed_capacity, int64_t gid0,
                void *stream);

/*
 * zrk_compact with the union list in its WIRE format, for the per-tick all-gather between GPUs: word 0 = number
 * of slots seen by at least one radar, word 1 = n, words 2 .. 1 + ceil(n/64) = one bit per slot (bit i%64 of word
 * 2 + i/64), then the masks of the seen slots in ascending slot order, packed 16 bits each when R <= 16, else 32.  About a quarter
 * of the bytes of the (index, mask) pairs; indices are implicit (the receiver knows each shard's offset).
 * `union_words` is the size of `union_bits` in 64-bit words (zrk_union_bits_words(n, R, entries) for `entries`
 * masks); masks beyond it are dropped, word 0 stays exact.  zrk_run_ticks writes this format into `packed` when
 * zrk_loop.flags has ZRK_F_UNION_BITS.
 */
int zrk_compact_bits(zrk_ctx *ctx, const uint32_t *vis_mask /* DEVICE */, int64_t n, int R, int32_t base_index,
                     void *workspace, int32_t *det_idx /* DEVICE [R][det_stride], may be NULL */, int64_t det_stride,
                     int32_t *det_cnt /* DEVICE [R+1] */, int64_t *union_bits /* DEVICE */, int64_t union_words,
                     void *stream);
int64_t zrk_union_bits_words(int64_t n, int R, int64_t entries);

/* Synchronises `stream` and reports whether every compaction on `workspace` so far ran to completion:
 * 0, or ZRK_E_STATE if a workgroup found control words it did not expect (a workspace shared between
 * contexts or overwritten by the caller).  Diagnostics; the hot loop never calls it. */
int zrk_compact_status(zrk_ctx *ctx, void *workspace /* DEVICE */, void *stream);

/* SectorRadar.smooth_objects with caller-supplied draws (modules/Radar.py:138-142):
 * pos[idx[j]] += noise[j] for j < k, noise row-major k x 3. */
int zrk_noise_apply(zrk_ctx *ctx, double *pos /* DEVICE [3][cap] */, int64_t capacity,
                    const int32_t *idx /* DEVICE */, int32_t idx_base, const double *noise /* DEVICE */,
                    int64_t k, void *stream);

/*
 * Missile.step(), 'active' branch, for every in-flight missile (modules/Missile.py:162-193):
 * own position from its trajectory, distance to the target's position of this tick, proximity
 * fuse, else life-timer decrement and timeout.  Emits MissileDetonateMessage contents
 * (modules/Missile.py:138-146) as (missile slot, target slot | -1) rows in list order.
 * `cur` is the buffer this tick's zrk_tick_sweep(ZRK_F_ADVANCE) writes; the kernel reads
 * pos[cur^1] (last tick's final positions) for targets that have not been stepped yet or
 * are no longer live, and recomputes from trajectories otherwise, so it may run before,
 * after or beside the sweep of the same tick.  With apply_kills != 0 the detonated missiles and
 * their targets are tombstoned at the end of the call (positions frozen from pos[cur], so the
 * sweep of this tick must already be enqueued): AirEnv.step()'s removal (modules/AirEnv.py:33-40)
 * without a host round trip.  With apply_kills == 0 the caller removes them (zrk_kill_slots).
 */
int zrk_missile_step(zrk_ctx *ctx, const zrk_entities *ents, int cur, const zrk_missiles *mis,
                     int64_t m, int64_t time_ms, int64_t dt_ms, int apply_kills, void *stream);

/* AirEnv.step()'s tombstoning (modules/AirEnv.py:33-40) for k slots: alive = 0 and the final
 * position (in pos[src]) frozen into both buffers so later readers see it. */
int zrk_kill_slots(zrk_ctx *ctx, const zrk_entities *ents, int src, const int32_t *slots /* DEVICE */,
                   int64_t k, void *stream);

/* Same, taking the event rows straight from the missile table on the device. */
int zrk_apply_events(zrk_ctx *ctx, const zrk_entities *ents, int src, const zrk_missiles *mis,
                     void *stream);

/* Missile._calculate_trajectory_params for a batch (modules/Missile.py:35-102), reading the
 * targets' current positions from pos[cur]. */
int zrk_launch_solve(zrk_ctx *ctx, const zrk_entities *ents, int cur, const zrk_launch_req *req /* DEVICE */,
                     zrk_launch_res *res /* DEVICE */, int64_t k, void *stream);

/*
 * A salvo without the host: zrk_launch_solve for k requests, then the successful ones enter the air in request
 * order -- table rows n .. n + count - 1 (trajectory (V, missile_pos, time_ms / 1000), alive, kind 1, both position
 * buffers at missile_pos, list_index = list_base + p when the table has that column) and missile rows
 * m .. m + count - 1 (slot, target, radius, period, status 1).
 *   replaces: MissileLauncher.step -> launch_missile -> Missile.step 'ready' -> _launch   modules/MissileLauncher.py:82-138,
 *             modules/Missile.py:104-133, and AirEnv taking the missile in, modules/AirEnv.py:42-43, for a whole salvo.
 * The k - count failed requests take the rows behind them, dead and inactive, so the caller continues with n + k and
 * m + k (upper bounds) without reading anything back; *count_out (DEVICE, may be NULL) receives count.  Writes the
 * trajectory columns, kind and list_index of the new rows although zrk_entities / zrk_missiles declare them const.
 */
int zrk_launch_salvo(zrk_ctx *ctx, const zrk_entities *ents, int cur, const zrk_missiles *mis, int64_t n, int64_t m,
                     const zrk_launch_req *req /* DEVICE */, zrk_launch_res *res /* DEVICE */, int64_t k,
                     int64_t time_ms, int32_t list_base, int32_t *count_out /* DEVICE */, void *stream);

/*
 * Track association of the command post for all detections of a tick.
 *   replaces: the `link_object` calls of CombatControlPoint.step (modules/CCP.py:171-219, made in the order of
 *             modules/CCP.py:414-429) -- the part of the command post that is quadratic in the reference.
 * Detection d (in processing order: radar after radar, list order, every object once) has position det_pos[d]
 * (obj.pos as the command post sees it) and speed det_speed[d] (obj.speed_mod).  Track t (target tracks first, then
 * missile tracks, each in the order the command post's dictionaries hold them) has the reference position trk_ref[t]
 * (track.target.prev_pos; for a missile track missile.prev_pos, or missile.pos where that is None) and the time
 * trk_upd[t] of its last update [s].  A track updated at now_s is skipped; a detection takes the strictly nearest
 * track whose distance lies within [max(0, speed * (age - slack_s)), max(0, speed * (age + slack_s))], age = now_s -
 * trk_upd[t] (slack_s = POSSIBLE_TARGET_RADIUS * dt, modules/constants.py:31); a track taken by an earlier detection of
 * the tick is gone for the later ones.  match[d] = index of the track, or -1 (NEW_TARGET).  Synchronises the
 * stream (it reads a counter back after every resolution round).
 */
int zrk_ccp_link(zrk_ctx *ctx, const double *det_pos /* DEVICE [D][3] */, const double *det_speed /* DEVICE [D] */,
                 int64_t D, const double *trk_ref /* DEVICE [T][3] */, const double *trk_upd /* DEVICE [T] */, int64_t T,
                 double now_s, double slack_s, int32_t *match /* DEVICE [D] out */,
                 void *scratch /* DEVICE, zrk_ccp_scratch_bytes(D, T) */, void *stream);
int64_t zrk_ccp_scratch_bytes(int64_t D, int64_t T);

/*
 * The command post's detection loop of one tick on the device: replaces the body of CombatControlPoint.step,
 * modules/CCP.py:406-429 -- for every detection, in FoundObjectsMessage order, link_object (:171-219), then new_target /
 * old_target / old_rocket (:322-366) with try_to_launch_missile (:287-320) -- with the result of that sequential loop.
 * Nothing is read back: the detections are rows of the entity table (`seq`, each row once: the reference skips ids it has
 * processed in the tick, :414; their number in DEVICE memory), positions come from the table (obj.pos = pos[cur],
 * obj.prev_pos = pos[cur ^ 1], None where start_time == now), the dictionaries live in device arrays.
 *   Tracks: target tracks [0, counts[0]) and missile tracks [0, counts[1]) in the dictionaries' insertion order.  A track is
 * keyed by the row whose id created it (`*_key`, fixed: add_target on an existing key replaces the entry in place, :88-93)
 * and holds the row it last matched (`*_obj`: TargetCCP.target / MissileCCP.missile, the live handle whose prev_pos
 * link_object reads); key_tt[row] = the target track keyed by that row, or -1.
 *   Launchers in dictionary order (at most 64): try_to_launch_missile takes the nearest one with launched < capacity.
 *   Output per detection d (= seq[d]): verdict 0 new / 1 old target / 2 old missile, the matched track's index within its
 * array (-1), the launcher the request goes to (-1); count[0] = detections, count[1] = launch requests; status 0 ok,
 * 2 a target track's handle has no prev_pos (the reference raises there), 3 track capacity exhausted (1 is not returned any
 * more).  The order-dependent part is resolved in at most `rounds` parallel rounds that end themselves on a device flag
 * (2-3 in practice); what a long chain of detections waiting for each other leaves unresolved after them, one workgroup
 * settles in detection order, so the step always ends decided.
 */
typedef struct {
    int64_t capacity;               /* tracks that fit each array */
    int32_t *tt_key, *tt_obj;       /* DEVICE [capacity] */
    double *tt_upd;                 /* DEVICE [capacity]  TargetCCP.upd_time [s] */
    uint8_t *tt_follow;             /* DEVICE [capacity]  TargetCCP.following */
    int32_t *tm_key, *tm_obj;       /* DEVICE [capacity] */
    double *tm_upd;                 /* DEVICE [capacity] */
    int32_t *counts;                /* DEVICE [2]: target tracks, missile tracks */
    int32_t *key_tt;                /* DEVICE [ents->capacity] */
    /* DEVICE [capacity][3] or NULL: where set (first component not NaN), the position link_object compares with instead of
     * the handle's row in pos[cur ^ 1].  For tracks whose object has left the air: the reference's handle keeps the
     * prev_pos of its last step, the table keeps only the object's last position (in both buffers). */
    const double *tt_ref_fixed, *tm_ref_fixed;
    /* (ABI 11) DEVICE [ents->capacity][3] or NULL: the same per table ROW -- where set (not NaN), what a handle of that row holds as
     * prev_pos, whichever track holds it: the buffer zrk_ctx_keep_prev has the loop fill when a row leaves the air. */
    const double *row_ref_fixed;
} zrk_ccp_tracks;
typedef struct {
    int32_t L, _pad;
    const double *pos;              /* DEVICE [L][3]  missile_launcher_coords */
    const int32_t *capacity;        /* DEVICE [L]     missile_launcher_capacity */
    int32_t *launched;              /* DEVICE [L]     missile_launcher_launched, in/out */
} zrk_ccp_launchers;
typedef struct {
    int32_t *obj, *verdict, *match, *launcher;   /* DEVICE [dmax] */
    int32_t *count;                 /* DEVICE [2] */
    int32_t *status;                /* DEVICE [1] */
} zrk_ccp_out;
int zrk_ccp_step(zrk_ctx *ctx, const zrk_entities *ents, int cur, const double *speed_mod /* DEVICE [cap]: obj.speed_mod */,
                 const int32_t *seq /* DEVICE [dmax] */, const int32_t *seq_count /* DEVICE [1] */, int64_t dmax,
                 const zrk_ccp_tracks *tracks /* HOST */, const zrk_ccp_launchers *launchers /* HOST */,
                 const zrk_ccp_out *out /* HOST */, double now_s, double slack_s, int rounds,
                 void *scratch /* DEVICE, zrk_ccp_step_scratch_bytes(dmax, tracks->capacity) */, void *stream);
int64_t zrk_ccp_step_scratch_bytes(int64_t dmax, int64_t track_capacity);
/*
 * The launch decisions of a zrk_ccp_step as the requests zrk_launch_salvo takes, in request order (the order
 * try_to_launch_missile was called in, modules/CCP.py:330, :342-343) and without the host:
 *   replaces: CPPLaunchMissileRequestMessage -> MissileLauncher.step -> launch_missile (modules/CCP.py:310-318,
 *             modules/MissileLauncher.py:82-101) for all requests of a tick -- request q = {target row of the detection, the chosen
 *             launcher's position, that launcher's missile parameters missile_params[l] = {velocity_module, detonate_period,
 *             detonate_radius}}.
 * req[0 .. k_max) is written in full: the requests first, then requests for no row (target_slot -1; zrk_launch_solve fails
 * them with rc 6), so zrk_launch_salvo can be called for k_max requests with nothing read back -- k_max bounds the
 * launches of a tick (the missiles the launchers have left); *count (may be NULL) receives their number.
 */
int zrk_ccp_requests(zrk_ctx *ctx, const zrk_ccp_out *out /* HOST */, int64_t dmax, const zrk_ccp_launchers *launchers /* HOST */,
                     const double *missile_params /* DEVICE [L][3] */, zrk_launch_req *req /* DEVICE [k_max] out */, int64_t k_max,
                     int32_t *count /* DEVICE [1] out */, void *stream);
/* check_if_missiles_launched -> add_missile (modules/CCP.py:160-169, :102-108): the missile in table row `row` enters the
 * missile dictionary (or replaces the entry of its key), updated "now". */
int zrk_ccp_add_missile(zrk_ctx *ctx, const zrk_ccp_tracks *tracks /* HOST */, int32_t row, double now_s, void *stream);

/*
 * (ABI 11) The battery's CLOSED loop without the host: launchers with their magazines, the two-tick way of a missile from the
 * command post's request into the air, on the device -- what lets sweep -> lists -> zrk_ccp_step -> launches -> rows in the air
 * run tick after tick with nothing read back.
 *   replaces: MissileLauncher.step (modules/MissileLauncher.py:82-138: the launcher's inbox of the tick before -- cancelled
 *             missiles back to the END of its list, launched ones announced, every request served by the missile popped from the
 *             END of the list, :58-80), Missile.step 'ready' -> _launch (modules/Missile.py:153-160, :104-133),
 *             check_if_missiles_launched (modules/CCP.py:160-169) and AirEnv taking a missile in (modules/AirEnv.py:42-43).
 * Latencies as in the reference: a request the command post makes in tick b (zrk_battery_requests) is solved in
 * tick b + 1 against the target's position after that tick's radars (zrk_battery_launchers), announced in tick b + 2 (the
 * launcher re-uses a cancelled missile from then on; the command post takes a launched one into its missile dictionary:
 * zrk_battery_announce) and flies from tick b + 3 (zrk_battery_activate, before that tick's sweep; its fuse timer starts there).
 *   The magazine is part of the table from the start: table rows row0 .. row0 + n_missiles - 1 and missile-table rows
 * 0 .. n_missiles - 1 are RESERVED (dead, status 0); the j-th missile that gets a trajectory takes table row row0 + j and
 * missile row j -- AirEnv's list order: launcher after launcher (module order), each in the order it served its requests --
 * so the table never grows and no count ever has to come back to the host.  A salvo lives in slot (build tick % 3) of the ring.
 */
typedef struct {
    int32_t L, k_max;               /* launchers (<= 64, the command post's order = module order); requests per tick at most */
    int32_t n_missiles, row0;
    const double *mi_pos;           /* DEVICE [n_missiles][3]  where missile q stands: its launcher's position (Missile.pos) */
    const double *mi_speed, *mi_period, *mi_radius;   /* DEVICE [n_missiles]  velocity_module, detonate_period, detonate_radius */
    int32_t *stack;                 /* DEVICE [L][n_missiles]  each launcher's list of missile numbers; it pops from the end */
    int32_t *top;                   /* DEVICE [L]              entries in each list */
    int32_t *sal_row, *sal_launcher, *sal_missile, *sal_rc, *sal_air;   /* DEVICE [3][k_max]: target row, launcher, missile served
                                     * (-1: none left), zrk_launch_res rc (7: no missile), air ordinal j of a launch (else -1) */
    double *sal_V;                  /* DEVICE [3][k_max][3] */
    int32_t *sal_count;             /* DEVICE [3][2]: requests, launches */
    int32_t *air_count;             /* DEVICE [1]: missiles that have a trajectory (rows given out) */
    int32_t *air_missile;           /* DEVICE [n_missiles]: missile number of air ordinal j */
    double *speed_mod;              /* DEVICE [ents->capacity]: the command post's obj.speed_mod column; rows that get a trajectory
                                     * are written (Missile.speed_mod = velocity_module, modules/Missile.py:28) */
    /* logs for the host, read after the loop: every solve in order -- {tick, missile, target row, rc, air ordinal} and V -- and
     * every detonation -- {tick, missile's table row, target's table row or -1}; log_count = {solves, detonations} */
    int32_t *log_solve;             /* DEVICE [log_cap][5] */
    double *log_V;                  /* DEVICE [log_cap][3] */
    int32_t *log_event;             /* DEVICE [log_cap][3] */
    int32_t *log_count;             /* DEVICE [2] */
    int32_t log_cap, _pad;
} zrk_battery;

/* obj.speed_mod of the first n rows (targets): |velocity| rounded as numpy's norm rounds it (modules/AirObject.py:35-36). */
int zrk_battery_speed_column(zrk_ctx *ctx, const zrk_entities *ents, int64_t n, double *speed_mod /* DEVICE */, void *stream);
/* Before the sweep of tick `tick`: the launches solved in tick - 2 enter the air (flag up, missile status 1; the loop's box records
 * of their row blocks start afresh). */
int zrk_battery_activate(zrk_ctx *ctx, const zrk_battery *bat, const zrk_entities *ents, const zrk_missiles *mis, int64_t tick,
                         void *workspace, void *stream);
/* After the sweep of tick `tick`: every launcher's step -- the salvo of tick - 2: cancelled missiles back on the lists; the salvo
 * of tick - 1: a missile popped per request, solved against pos[cur], the launches given their rows (trajectory written, still
 * dead) in AirEnv's order. */
int zrk_battery_launchers(zrk_ctx *ctx, const zrk_battery *bat, const zrk_entities *ents, int cur, const zrk_missiles *mis,
                          int64_t tick, int64_t time_ms, void *stream);
/* The command post's check_if_missiles_launched of tick `tick`: the launches solved in tick - 1, in order, into its missile
 * dictionary. */
int zrk_battery_announce(zrk_ctx *ctx, const zrk_battery *bat, const zrk_ccp_tracks *tracks, int64_t tick, double now_s, void *stream);
/* The detections of a tick in FoundObjectsMessage order, every object once (modules/CCP.py:406-417), as table rows: from the R
 * lists of zrk_run_ticks (det_idx / det_cnt, list indices) and the tick's masks (indexed by list index): an entry belongs to the
 * first radar that saw it.  row_of_list (DEVICE, may be NULL: the table is in list order) maps list indices to rows. */
int zrk_battery_sequence(zrk_ctx *ctx, const int32_t *det_idx, int64_t det_stride, const int32_t *det_cnt, int R, int32_t base_index,
                         const uint32_t *vis_mask, const int32_t *row_of_list, int32_t *seq /* DEVICE [seq_cap] out */,
                         int32_t *seq_count /* DEVICE [1] out */, int64_t seq_cap, void *stream);
/* The launch decisions of the zrk_ccp_step of tick `tick` as that tick's salvo: per launcher (module order), each in request order --
 * the order the launchers will serve them in -- with the command post's own view of who asked for what. */
int zrk_battery_requests(zrk_ctx *ctx, const zrk_battery *bat, const zrk_ccp_out *out, int64_t dmax, int64_t tick, void *stream);
/* The tick's detonations (zrk_missiles::ev_*) appended to the log. */
int zrk_battery_log_events(zrk_ctx *ctx, const zrk_battery *bat, const zrk_missiles *mis, int64_t tick, void *stream);
/* Rows that leave the air keep the prev_pos their handle held in buf ([capacity][3], DEVICE, the caller fills it with NaN; NULL:
 * off) -- for zrk_ccp_tracks::row_ref_fixed.  Plain loop only (calls of fewer than four ticks). */
int zrk_ctx_keep_prev(zrk_ctx *ctx, double *buf);

/* Static scan parameters of one radar (modules/Radar.py:13-42): what
 * move_to_next_sector_circular reads besides the current angles. */
typedef struct {
    double azimuth_speed, elevation_speed, elevation_start;
    int32_t mode;                   /* 0 "horizontal", 1 "vertical", anything else: never moves */
    int32_t _pad;
} zrk_scan;

/* SectorRadar.move_to_next_sector_circular (modules/Radar.py:96-117) for R radars, on the host
 * (a7 in SURVEY.md section 8: scalar per-radar state, uploaded with the next sweep). */
int zrk_scan_advance(zrk_radar *radars /* HOST, in/out */, const zrk_scan *scan /* HOST */, int R);

/* Loop state of zrk_run_ticks (in/out). */
typedef struct {
    int64_t n;                      /* slots in use */
    int64_t time_ms, dt_ms;         /* Timer.get_time / get_dt (modules/Timer.py:20-30) */
    int64_t gid0;                   /* global index of slot 0 (multi-GPU shard offset) */
    uint64_t seed, tick;            /* Philox key / tick counter */
    int32_t cur;                    /* position buffer holding the LAST completed tick */
    int32_t base_index;             /* added to slots in det_idx */
    uint32_t flags;                 /* ZRK_F_PHILOX etc.; ZRK_F_ADVANCE is implied */
    int32_t vis_cur;                /* out: 0 = vis_mask, 1 = vis_mask_alt holds the last tick's masks */
} zrk_loop;

/*
 * K ticks of the L1 path, enqueued back to back on `stream` without returning to the caller:
 * per tick  flip buffer -> zrk_tick_sweep(ADVANCE) -> zrk_compact -> zrk_missile_step(apply_kills)
 * -> zrk_scan_advance -> time += dt.
 *   replaces the per-tick `module.step()` calls of Manager.run_simulation for AirEnv and every
 *   SectorRadar (modules/Manager.py:123-131, :140).
 * Detection outputs hold the last tick's lists.  If ents->vis_mask_alt is given (and compaction is on)
 * the ticks alternate between the two mask buffers: each tick stores only its detections (list-indexed
 * stores are scattered when the table is spatially sorted, so the zeros are not worth writing) into the
 * buffer the previous tick's compaction cleared; st->vis_cur says which buffer is current afterwards.
 * From the second tick on the two buffers belong to this loop, also between calls: do not write them.
 * If sweep_ms != NULL the sweep launch that holds the LAST tick of every window of prof_stride ticks (and the call's last
 * tick) is timed with a pair of HIP events riding on its dispatch, the stream is synchronised at the end and
 * sweep_ms[k / prof_stride] receives the kernel's duration in milliseconds (prof_stride < 0: the events are only
 * recorded, zrk_read_sweep_ms reads them later; zrk_read_sweep_ticks says how many ticks each timed launch swept).
 *
 * Calls of four ticks or more on a table of 5e4 rows or more, with compaction, a second mask buffer and a missile table
 * the single-workgroup finisher covers, run OVERLAPPED: the lists and the ordered event list of tick t are compacted on
 * a side stream of the context beside the sweep of tick t + 1 (launched by a thread of the context's own when the
 * compute stream says, through a word of pinned host memory, that sweep t + 1 has started), and `stream` carries one
 * launch per tick: removals decided by a tick's missile phase are marks that the rows' own threads carry out in the
 * next sweep (tombstones again when the call returns), radar records travel in the sweep's arguments.  Same results;
 * everything the caller reads is written on `stream` or taken in by it before the call returns, so the outputs are the
 * caller's as before (a wait for whatever the side stream may still be running then stands behind the call's last launch
 * on `stream`).  The call's LAST
 * compaction (no exchange) is launched on `stream` itself, in order behind the last sweep -- by the context's thread: a
 * caller must not be capturing `stream` into a graph, and must order its own streams if consecutive calls come on
 * different ones (ZRK_TAIL_COMPUTE=0: on the side stream like the others, released by a launch behind the last sweep;
 * ZRK_TAIL_FREE=0: behind an event that takes the side stream's compaction before it in -- by default a two-tick call's last
 * compaction has control words of its own and waits for nobody, and the lists of every tick but the call's last go to
 * buffers of the context).  Differences
 * a caller can see: the masks of all ticks but the LAST of such a call live in buffers of the context
 * (ents->vis_mask / vis_mask_alt hold the last tick's, as st->vis_cur says), likewise mis->ev_code; the dispatch order
 * of a sweep's workgroups is never the same twice (it does not enter any result).
 *   One scenario without an exchange goes one step further: a sweep launch covers TWO consecutive ticks (a tick's
 * positions never depend on the tick before -- Trajectory.get_pos recomputes them, modules/AirObject.py:23-25 -- so the
 * trajectory columns are read once for both; the second tick's positions go to the other position buffer, its masks to
 * a buffer of their own), and one launch on the side stream compacts both ticks' lists (the first tick's into buffers
 * of the context: a call's intermediate lists are overwritten by the next tick's wherever they are written).  What the
 * first tick's missile phase removes is swept once more by the row threads, which cannot know, and put right afterwards;
 * a missile that needs, in the second tick, the position a row held after the first (a target removed by that tick, a
 * target behind the missile in the list) replays that row's radar phase.  Same results, tested against the two-launch
 * loop and the oracle (tests/test_gpu_overlap.py).  ZRK_PAIR=0: one tick per launch; ZRK_PAIR_COMPACT=0: two compaction
 * launches per pair.
 * Environment (read at zrk_ctx_create / zrk_ctx_reload_env): ZRK_OVERLAP=0 never overlap; ZRK_OVERLAP_MIN=k from k
 * ticks per call; ZRK_OVERLAP_MIN_ROWS=n from n rows; ZRK_GATHER_RECORDS=0 the missile phase reads its targets from the
 * columns instead of the 64-byte records the loop keeps per row (64 B x capacity of device memory, the context's);
 * ZRK_SWEEP_ORDER=0 sweep in table order; ZRK_COMPACT_ORDER=block|ticket; ZRK_TIME_BY_RECORDS=1 time between two
 * recorded events; ZRK_HELPER_IDLE_MS how long the library's threads spin after their last item (default 1) and
 * ZRK_HELPER_YIELD_MS how long they then stay runnable, yielding their core between looks (default 5; bench.py asks for 250),
 * before they sleep -- calls less than that apart find them awake; ZRK_STALL_US=n reports on stderr every host-side wait of the library longer
 * than n microseconds with the line it stands in; tuning / diagnostics, defaults chosen by measurement:
 * ZRK_PAIR_THREADS=256|512|1024 and ZRK_PAIR_COMPACT_BLOCKS (workgroup size of the pair compaction, and up to how many
 * workgroups a pair's compactions are one launch), ZRK_COMPACT_GROUP=0|4|8|16|32 (two-level sums of the workgroup records),
 * ZRK_COMPACT_ITEMS, ZRK_COMPACT_FUSED_MAX_BLOCKS, ZRK_SIDE_CUS=n[,first] / ZRK_SIDE_PRIORITY=high|low (the side stream's
 * place on the device), ZRK_TAIL_COMPUTE=0 / ZRK_TAIL_EVENT=1 (the call's last compaction on the side stream, released by a launch / an event), ZRK_CCP_GRID=0|1 (zrk_ccp_step's
 * candidate pass: all pairs / spatial index), ZRK_SIDE_SETTLE=0 (the side stream's thread does not finish that stream behind a
 * call's last item), ZRK_MARKS_IN_TAIL=0 (a call's removal marks are carried out by a launch of
 * their own behind the last sweep instead of by extra workgroups of the call's last pair compaction), ZRK_TRACE=1 (host time
 * stamps of a call on stderr as it returns; =2: as the next call starts, outside what a caller times).
 * Every host-side wait is bounded by ZRK_HOST_WAIT_MS (default 30000): if the
 * side stream's thread waits that long for the compute stream to reach the next sweep (the caller had queued more work in
 * front of the loop than that, or the device is gone) it gives up, THAT call returns ZRK_E_STATE (its lists are not
 * valid; zrk_compact_status says so too), and the next call starts the side stream afresh -- the failure is not sticky.
 */
/* 1 if the last zrk_run_ticks / _x / _ensemble call of this context ran overlapped, 0 if not (diagnostics). */
int zrk_last_run_overlapped(zrk_ctx *ctx);
int zrk_run_ticks(zrk_ctx *ctx, const zrk_entities *ents, const zrk_missiles *mis, int64_t m, zrk_loop *st,
                  zrk_radar *radars /* HOST, in/out */, const zrk_scan *scan /* HOST */, int R, void *workspace,
                  int32_t *det_idx, int64_t det_stride, int32_t *det_cnt, int64_t *packed,
                  int64_t packed_capacity, int K, float *sweep_ms /* HOST, may be NULL */, int prof_stride,
                  void *stream);

/*
 * Per-tick exchange of the compacted detection list between the GPUs of one node (no counterpart in the
 * reference, which is one process: modules/Manager.py:123-131).  One process per GPU; every
 * rank holds a contiguous shard of AirEnv's list, and what FoundObjectsMessage (modules/Radar.py:168-174) would
 * carry for the whole population is the rank-ordered concatenation of the shards' lists.  The exchange object
 * owns an RCCL communicator (librccl is bound at run time from `rccl_path`, or found by name when NULL) and a
 * stream of its own; zrk_run_ticks_x issues one ncclAllGather per tick on it, behind that tick's compaction and
 * beside the next tick's sweep.  Rank 0 makes the id and hands it to the others by any means (128 bytes).
 */
typedef struct { char internal[128]; } zrk_rccl_id;          /* ncclUniqueId */
/* Lists in flight: a tick's list is rewritten ZRK_EXCHANGE_SLOTS ticks later, and its collective must be through by
 * then (two were enough for correctness; with two ticks per launch four are only two launches deep, and the calling thread
 * then waits for a collective before every other launch -- 18 us gaps between sweeps in the kernel trace; eight keep a
 * late collective from ever stalling the compute stream's host). */
#define ZRK_EXCHANGE_SLOTS 8
typedef struct zrk_exchange zrk_exchange;

int zrk_exchange_unique_id(const char *rccl_path, zrk_rccl_id *id /* HOST out */);
/* Collective over all ranks (ncclCommInitRank).  *out is set also on failure, for zrk_exchange_last_error. */
int zrk_exchange_create(const char *rccl_path, const zrk_rccl_id *id, int world, int rank, int device,
                        zrk_exchange **out);
void zrk_exchange_destroy(zrk_exchange *x);
const char *zrk_exchange_last_error(zrk_exchange *x);
/* One all-gather of `words` 64-bit words per rank: recv = [world][words].  Enqueued on the exchange's stream
 * behind everything `stream` holds so far; ZRK_EXCHANGE_SLOTS slots may be in flight.  The pattern is ncclAllGather, as RCCL
 * chooses to run it, or -- ZRK_EXCHANGE_ALGO=direct at zrk_exchange_create -- world - 1 grouped ncclSend / ncclRecv pairs per
 * rank, one hop over each peer's own xGMI link; results are the same.  How much a rank sends is the
 * caller's `words`: zrk_union_bits_words(n, R, 0) carries count, n and the bitmap of the slots seen by any radar -- the
 * compaction writes no mask where the list has no room for one --, zrk_union_bits_words(n, R, entries) the radar masks too. */
int zrk_exchange_all_gather(zrk_exchange *x, int slot, const int64_t *send /* DEVICE */, int64_t *recv /* DEVICE */,
                            int64_t words, void *stream);
/* The last all-gather posted on `slot` is over before anything launched on `stream` after this call runs (nothing if
 * none was posted).  By default the HOST waits for it (the collective is ZRK_EXCHANGE_SLOTS ticks old, and a wait in the stream costs
 * the stream ~5 us of idle device per tick); ZRK_EXCHANGE_WAIT_IN_STREAM=1 at zrk_exchange_create makes it a stream
 * wait.  Inside zrk_run_ticks_x the hand-over to the exchange's stream needs no event either: the next tick's sweep
 * raises a word of device memory as it starts, a one-lane kernel on the exchange's stream waits for it, and the
 * collective is issued by a thread of the exchange's own (ZRK_EXCHANGE_EVENTS=1: events instead;
 * ZRK_EXCHANGE_THREAD=0: issued by the calling thread).  In the overlapped loop (see zrk_run_ticks) the word is raised
 * by a one-thread launch behind the compaction on the context's side stream.
 * Helper threads: the exchange's own thread exists only where the rank has three host cores or more to itself
 * (usable cores / world >= 3, ZRK_HELPERS=1|2 forces either); otherwise the side stream's thread issues the collectives
 * as well, and all helpers sleep between calls.
 * Failure: the wait kernel does not wait for ever (~1-2 s, ZRK_WAIT_FLAG_SPINS looks).  When it gives up, the collective
 * behind it still runs (the peers are in it), but the list goes out POISONED -- count -1, which exchange.decode_* reject
 * -- as does every later list of this exchange; zrk_run_ticks_x / zrk_exchange_sync / zrk_exchange_all_gather return
 * ZRK_E_STATE from then on.  A profiler that serialises kernels across streams (rocprofv3 --pmc) forces exactly that:
 * do not collect counters on the exchange path.  A host-side wait for a collective that does not end (a dead peer) is
 * bounded by ZRK_HOST_WAIT_MS and returns ZRK_E_STATE.  Unlike a failure of the context's side stream (zrk_run_ticks: the
 * next call starts afresh), a failed exchange -- a give-up, a collective that could not be launched -- is final for that
 * object: the communicator's ranks are no longer in step; destroy it on every rank and create a new one. */
int zrk_exchange_wait(zrk_exchange *x, int slot, void *stream);
/* Block the host until the exchange's stream is idle. */
int zrk_exchange_sync(zrk_exchange *x);
/* What the first multi-rank record needs to check itself: the communicator's own idea of its size (ncclCommCount; -1: this RCCL
 * does not say), which pattern the collectives use (ZRK_EXCHANGE_ALGO=direct: grouped ncclSend / ncclRecv to every peer, one hop
 * each, instead of ncclAllGather), how many collectives were issued, and how often / how long in all the calling thread had to
 * wait for a collective posted ZRK_EXCHANGE_SLOTS ticks earlier before it could reuse its send buffer.  No reference
 * counterpart (the reference is one process). */
typedef struct zrk_exchange_stats {
    int32_t world, rank, comm_ranks, direct, helper_threads;
    int32_t grouped_pairs;          /* 1: the two collectives of a two-tick launch go out as one RCCL group (ZRK_EXCHANGE_GROUP=0: not) */
    int64_t collectives, host_waits;
    double host_wait_us;
} zrk_exchange_stats;
int zrk_exchange_info(zrk_exchange *x, zrk_exchange_stats *out /* HOST */);
/* (ABI 11) The helper threads per rank an exchange of `world` ranks on this host would start: 2 where the rank has three usable host
 * cores or more to itself (affinity mask, capped by the cgroup CPU quota), else 1 -- the side stream's thread then issues the
 * collectives as well; ZRK_HELPERS=1|2 forces either.  Touches no device: a launcher can say how many host threads its ranks
 * will keep busy (1 + this) before it starts them. */
int zrk_exchange_plan_helpers(int world);

/* What zrk_run_ticks_x sends each tick: this rank's list in the wire format of zrk_compact_bits, followed -- when
 * ev_capacity > 0 -- by the tick's detonations, so that MissileDetonateMessage (modules/Missile.py:138-146) reaches
 * every rank with the list: word `words - 1 - ev_capacity` = number of events, then one word per event,
 * (global list index of the missile << 32) | global list index of the target, 0xFFFFFFFF for a self-detonation.
 * Tick t uses slot t % ZRK_EXCHANGE_SLOTS of send / recv. */
typedef struct {
    zrk_exchange *x;
    int64_t *send[ZRK_EXCHANGE_SLOTS];   /* DEVICE [words]; tick t goes through slot t % ZRK_EXCHANGE_SLOTS */
    int64_t *recv[ZRK_EXCHANGE_SLOTS];   /* DEVICE [world][words] */
    int64_t words;                  /* zrk_union_bits_words(capacity, R, entries) + (ev_capacity ? 1 + ev_capacity : 0) */
    int32_t ev_capacity;
    /* (ABI 11) radars of interest, one bit per radar; 0: all of them.  What a consumer on another rank reads is one
     * FoundObjectsMessage per radar IT listens to (modules/CCP.py:409-417 keeps radar_id = msg.sender_id): with a mask here the list
     * that crosses the links is the union list of these radars alone -- a slot is on it when one of them saw it, its mask carries
     * their bits -- instead of everything or nothing (zrk_union_bits_words(n, R, entries) with room for masks, or (n, R, 0) for the
     * bitmap of the slots any of THEM saw).  The rank's own per-radar lists are not available beside it (det_idx must be NULL). */
    uint32_t interest;
} zrk_exchange_io;

/* zrk_run_ticks with the per-tick exchange: `packed` must be NULL when xio is given (the list goes to xio->send),
 * zrk_loop.flags must have ZRK_F_UNION_BITS.  xio == NULL: exactly zrk_run_ticks. */
int zrk_run_ticks_x(zrk_ctx *ctx, const zrk_entities *ents, const zrk_missiles *mis, int64_t m, zrk_loop *st,
                    zrk_radar *radars /* HOST, in/out */, const zrk_scan *scan /* HOST */, int R, void *workspace,
                    int32_t *det_idx, int64_t det_stride, int32_t *det_cnt, int64_t *packed,
                    int64_t packed_capacity, const zrk_exchange_io *xio /* HOST, may be NULL */, int K,
                    float *sweep_ms /* HOST, may be NULL */, int prof_stride, void *stream);

/*
 * Batched Monte-Carlo ensemble (BASELINE configs[4]; no counterpart in the reference, which runs one scenario per
 * process: main.py:151-174): S independent scenarios in ONE table, scenario s in rows [s * rows_per_scenario,
 * (s + 1) * rows_per_scenario), each with its own `radars` SectorRadars, scan state, noise key, missiles and
 * detection lists -- one sweep launch and one compaction launch per tick for all of them.  The radars live on the
 * device: workgroups riding in the compaction move every scan on (modules/Radar.py:96-117, :205) and derive the
 * next tick's records, so the host has no per-scenario work in the loop.  list_index is required and counts
 * within the scenario; rows past a scenario's population are padding (alive = 0).  Missile rows name table rows.
 */
typedef struct {
    int32_t scenarios;              /* S */
    int32_t radars;                 /* R per scenario, <= ZRK_MAX_RADARS */
    int64_t rows_per_scenario;      /* a multiple of 1024 */
    zrk_radar *radar_state;         /* DEVICE [S][R], in/out: advanced by every tick */
    const zrk_scan *scan;           /* DEVICE [S][R] */
    const double *d2_max;           /* DEVICE [S][R]: zrk_d2_threshold(max_distance) of each radar */
    const uint64_t *seeds;          /* DEVICE [S]: noise key of each scenario (zrk_loop.seed is not used) */
    void *tables;                   /* DEVICE scratch, zrk_ensemble_table_bytes(S) */
} zrk_ensemble;

int64_t zrk_ensemble_table_bytes(int scenarios);
/* The largest d2 whose correctly rounded square root is <= max_distance: `distance > max_distance`
 * (modules/Radar.py:57) as a comparison of squares. */
double zrk_d2_threshold(double max_distance);
/* zrk_run_ticks for an ensemble.  det_idx: [S][R][det_stride] (scenario-local list indices), det_cnt: [S][R + 1];
 * zrk_loop.n must be S * rows_per_scenario, gid0 0. */
int zrk_run_ticks_ensemble(zrk_ctx *ctx, const zrk_entities *ents, const zrk_missiles *mis, int64_t m, zrk_loop *st,
                           const zrk_ensemble *ens /* HOST */, void *workspace, int32_t *det_idx, int64_t det_stride,
                           int32_t *det_cnt, int K, float *sweep_ms /* HOST, may be NULL */, int prof_stride, void *stream);

/* zrk_run_ticks* with prof_stride < 0 (sweep_ms may be NULL) only records the event pairs, every -prof_stride-th tick,
 * and returns without synchronising; this reads the first n durations [ms] afterwards (it waits for them). */
int zrk_read_sweep_ms(zrk_ctx *ctx, float *sweep_ms /* HOST out */, int n);
/* The overlapped loop of one scenario without an exchange sweeps TWO consecutive ticks per launch (the trajectory columns
 * are read once for both; ZRK_PAIR=0: one tick per launch): how many ticks the launch behind each timing sample swept, and
 * what the last zrk_run_ticks* call did (1 or 2). */
int zrk_read_sweep_ticks(zrk_ctx *ctx, int32_t *ticks /* HOST out */, int n);
int zrk_last_run_ticks_per_launch(zrk_ctx *ctx);
/* The sweeps time THEMSELVES: with stamps on, the sweep launches of zrk_run_ticks* (all of a call, or every k-th so that at most
 * 64 are kept) write the wall clock (s_memrealtime, 100 MHz) when their first waves start and when each of their waves ends;
 * zrk_read_sweep_stamps reduces them on `stream` (the stream the call ran on; it synchronises it) to one duration per sampled
 * launch [us] -- first wave in to last wave out -- and says how many ticks each swept.  Unlike an event pair on the dispatch
 * this puts no signal, no barrier and no packet on the stream: the launches run as untimed ones do.  Returns the number of
 * samples written (<= cap), or an error.  No reference counterpart (measurement). */
int zrk_sweep_stamps(zrk_ctx *ctx, int on);
int zrk_read_sweep_stamps(zrk_ctx *ctx, float *sweep_us /* HOST out */, int32_t *ticks /* HOST out, may be NULL */, int cap, void *stream);
/* ... and WHEN those launches ran: first wave in / last wave out of each sampled launch of the last zrk_read_sweep_stamps, in
 * microseconds from the first one's first wave (the device's own clock): the span of a call's sweeps and the gaps between them,
 * without a profiler (bench.py: setup.sweeps_span_us, setup.sweep_gaps_us).  Returns how many. */
int zrk_last_sweep_stamp_times(zrk_ctx *ctx, double *begin_us /* HOST out */, double *end_us /* HOST out */, int cap);

/* Numerics self-test hooks used by tests/: y[i] = op(a[i], b[i]) in device binary64.
 * op: 0 sqrt(a), 1 a/b, 2 atan2(a,b), 3 asin(a), 4 fma-chain norm of (a,b,0). */
int zrk_selftest_math(zrk_ctx *ctx, int op, const double *a, const double *b, double *y /* DEVICE */,
                      int64_t n, void *stream);
/* out[i*3..] = the noise triple the `ordinal`-th detection (0-based) of entity entity0 + i draws
 * in tick `tick` under ZRK_F_PHILOX. */
int zrk_selftest_noise(zrk_ctx *ctx, uint64_t seed, uint64_t tick, uint32_t ordinal, int64_t entity0,
                       double *out /* DEVICE [n][3] */, int64_t n, void *stream);
/* Every host-side wait of the library (for a helper thread, for a collective posted ZRK_EXCHANGE_SLOTS ticks ago, for a
 * compaction on the side stream) is bounded by ZRK_HOST_WAIT_MS (default 30000) and ends in ZRK_E_STATE.  This runs the
 * wait on a condition that never comes for limit_ms and returns what such a wait returns; no device needed.
 * what: 0 the wait itself, 1 a full hand-over ring that nobody empties. */
int zrk_selftest_host_wait(int what, int limit_ms);

#ifdef __cplusplus
}
#endif
#endif /* ZRK_HOT_H */

// zrk_hot.hip -- HIP kernels (gfx950 / MI355X) and the C ABI of include/zrk_hot.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (see csrc/Makefile).
// -ffp-contract=off is part of the numerics contract: the reference rounds
// start_pos + velocity*(t - t0) three times (modules/AirObject.py:23-25) and the only fused
// operations are the explicit fma() chains that reproduce OpenBLAS ddot for n = 3
// (np.linalg.norm / np.dot).
//
// No MFMA anywhere: the path has no dense contraction.  The sweep is one pass over the
// entity columns (85 algorithmic bytes per live entity) with the radar parameter block
// passed by value in the kernel-argument segment, so every wave reads it through the scalar
// cache into SGPRs; an LDS tile of radars would only add a copy.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <sched.h>

#include "zrk_hot.h"

#define ZRK_API extern "C" __attribute__((visibility("default")))

namespace {

// Diagnostics build only (make probe -> libzrk_hot_probe.so, used by tools/sweep_phases.py): wall-clock
// stamps (s_memrealtime, 100 MHz) per sweep wave.  The product library compiles these to nothing.
#ifdef ZRK_PROBE_BUILD
__device__ long long *g_wave_probe;
#define ZRK_WAVE_PROBE(wave, slot, value)                                                                  \
    do {                                                                                                   \
        if ((threadIdx.x & 63) == 0 && g_wave_probe) g_wave_probe[(int64_t)(wave) * 8 + (slot)] = (value);  \
    } while (0)
#define ZRK_PROBE_ENTRY() const long long probe_entry = wall_clock64()      // (a wave's first instruction: slot 0)
#else
#define ZRK_WAVE_PROBE(wave, slot, value)
#define ZRK_PROBE_ENTRY()
#endif

constexpr double kRad2Deg = 180.0 / 3.14159265358979323846;   // numpy.degrees factor
constexpr int kCompBlock = 1024;          // list slots per compaction workgroup (count / scan / scatter unit)
constexpr uint32_t kSparseVis = 1u << 16;  // internal sweep flag: write only non-zero masks
constexpr uint32_t kSparseVis2 = 1u << 17; // ... the same for the second tick of a pair launch
constexpr uint32_t kNoInside = 1u << 21;   // diagnostics (ZRK_DIAG bit 0): no wave-level "certainly visible" shortcut
constexpr uint32_t kNoBoxCache = 1u << 22; // diagnostics (ZRK_DIAG bit 1): every wave takes its box from the rows it has just computed
constexpr float kGuard = 3e-5f;                               // relative half-width of the "ambiguous" band

// Device-side radar records.  Hot: what every (radar, entity) pair touches -- the float32
// pre-classification that settles every pair farther than kGuard (relative) from a sector edge or
// the range sphere; 20 dwords, fetched with one scalar load burst per radar.  Cold: the exact
// binary64 gate of modules/Radar.py:56-70, read only by the rare pairs inside a guard band.
struct RadarHot {
    double px, py, pz;
    float d2f_in, d2f_out;      // float32 d2 below / above which the range gate is already decided
    float elx, ely, ehx, ehy;   // unit vectors of the (clamped) azimuth edges
    float s_lo_up, s_hi_up;     // sin(elevation) bounds for dz >= 0
    float s_lo_dn, s_hi_dn;     // and for dz < 0 (elevation wraps to (90,180])
    float az_guard;             // kGuard * max_distance: farther than this outside an azimuth edge = certainly out
    float az_sgn;               // +1: inside = both edge tests (width <= 180); -1: either (edges stored negated)
    uint32_t pad[2];
};
static_assert(sizeof(RadarHot) == 80, "RadarHot must be 20 dwords");

// What the pre-pass of sweep_row reads: the range sphere and the azimuth wedge, widened (host: derive_pre)
// by everything this tick's noise can add to a row before the radar looks at it.
struct RadarPre {
    float px, py, pz;           // radar position, binary32
    float d2_out;               // beyond this squared distance: certainly out of range
    float elx, ely, ehx, ehy;   // azimuth edges as in RadarHot
    float az_sgn;
    float az_out;               // farther than this outside the wedge: certainly outside
    // "every row of the box is certainly visible, whatever this tick's noise did to it before":
    float d2_in;                // farthest corner of the box closer than this: certainly in range (-1: never)
    float t_in;                 // margin [m] the box must keep from every face of the sector
    float s_lo_up, s_hi_up;     // elevation bounds as in RadarHot
    float s_lo_dn, s_hi_dn;
    float z_in;                 // rows higher than this above the radar's plane stay above it (below: mirrored)
    uint32_t pad;
    double pz64;                // the radar's height in binary64, for the per-row side-of-the-plane test
};
static_assert(sizeof(RadarPre) == 80, "RadarPre must be 20 dwords");
constexpr int kPreVec = (int)(ZRK_MAX_RADARS * sizeof(RadarPre) / 16);      // 16-byte pieces of the table
static_assert(kPreVec <= ZRK_BLOCK, "one piece per thread");

struct RadarCold {
    double d2_max;              // largest d2 with sqrt(d2) <= max_distance  (== `dist > max` gate)
    double az_lo, az_hi;        // current_azimuth, current_azimuth + azimuth_range
    double el_lo, el_hi;
    // the sector's faces once more in binary64 (same construction as RadarHot's): the middle tier of the decision
    double elx, ely, ehx, ehy;  // unit vectors of the (clamped) azimuth edges, times az_sgn
    double s_lo_up, s_hi_up, s_lo_dn, s_hi_dn;
    double az_sgn;              // 0: no middle tier for this radar (degenerate sector: always the reference formula)
};

struct RadarBlock {                                 // lives in the kernel-argument segment (by value)
    uint32_t prew[ZRK_MAX_RADARS][20];              // pre-pass records (RadarPre) as dwords, see sweep_rows
    uint32_t hotw[ZRK_MAX_RADARS][20];              // RadarHot records as dwords (indexed, never addressed)
    RadarCold cold[ZRK_MAX_RADARS];
};

struct WaveBox;

// The first kSweepHeadBytes of the sweep's arguments are what a wave needs before it can address its rows; the kernel takes
// them in with ONE burst of scalar loads (SweepHead) instead of a dozen dependent ones, each waited for at its first use
// (the prologue was a chain of ten scalar round trips in front of the row loads: a microsecond of every wave's life).
struct SweepHead {
    const int32_t *order;       // row block of each workgroup, expensive ones first (NULL: identity)
    unsigned long long *stamps; // wall-clock stamps of THIS launch, see SweepParams::stamps
    uint32_t *flag;
    const char *rb_table;       // DEVICE RadarBlock per scenario (NULL: one scenario, records in `rb` below)
    const uint64_t *seeds;
    WaveBox *boxes;             // per-block box records (NULL: none kept), see WaveBox
    const double *sp, *vel, *t0;
    const uint8_t *alive;
    const int32_t *lidx;        // list index of each table row (NULL: the table is in list order)
    uint8_t *pend;              // removal marks, see below
    double *pos;
    int64_t n, cap;
    uint64_t seed;
    int32_t mb, nb, bps;        // mb leading workgroups step the missiles, nb sweep; bps: row blocks per scenario of an ensemble
    uint32_t bps_magic;         // ceil(2^32 / bps): block / bps == umulhi(block, magic) for block < 2^16
    uint32_t flags, flag_value;
    int32_t R, _pad0;
};
constexpr int kSweepHeadBytes = 160;
static_assert(sizeof(SweepHead) == kSweepHeadBytes, "the head is forty dwords");

struct SweepParams : SweepHead {
    uint32_t *vis;              // indexed by LIST index
    int32_t *order_next;        // the same for the next tick, built as this sweep goes (NULL: not built), see below
    // overlapped loop: removals decided by this launch's missile phase are MARKS (pend[row] = the removal tick's mark value,
    // written only where there is none yet, never cleared inside a call) that the row's own thread carries out in the next
    // LAUNCH -- flag down, position frozen in both buffers -- instead of a launch between two sweeps (AirEnv.py:33-40:
    // effective from the next tick either way).  A mark value is 2 + 2 * (tick % 126) + b, b = the index of the position
    // buffer that was current in the removal tick: pos_abs[b] holds the frozen position, pos_abs[b ^ 1] is to receive it.
    // A mark other than this launch's means "removed before this launch".  NULL: no marks
    uint32_t mark, mark2;       // this launch's mark values (mark2: the second tick of a pair, else = mark)
    uint8_t *alive_w;           // (the alive column again, writable)
    double *pos_prev;           // the other position buffer: last tick's positions -- and, in a PAIR launch, the second tick's output
    double *pos_abs[2];         // the two position buffers by absolute index (see the marks above)
    // PAIR launch: two consecutive ticks t, t + 1 in one pass over the table -- the trajectory columns are read once, both
    // ticks' positions, gates and noise computed from them (a tick's positions never depend on the tick before: the
    // reference recomputes them from the trajectory, AirObject.py:23-25), tick t + 1's positions go to pos_prev, its masks to
    // vis2, its radar records are rb2.  What tick t removes is not known to the row threads (its missile phase runs in
    // this very grid): they sweep tick t + 1 as if nothing was removed, and the rows tick t did remove are put right
    // afterwards -- mask bits by tick t's event builder (MissileArgs::clear_vis), flag and position by the next launch.
    double t2;
    uint32_t *vis2;
    uint32_t *order_ctr;        // kOrderRegions pairs (expensive / cheap row blocks recorded so far), kOrderCtrStride words apart
    uint32_t *order_ctr_next;   // the next tick's set, cleared here
    // the mask buffer the CALL's last tick will write (the caller's), zeroed in passing by the call's first launch so that the
    // last one writes its detections only (a dense write of a million words made a call's last launch 40-42 us instead of ~33)
    uint32_t *vis_clear;
    double t;
    uint64_t tick;
    int64_t gid0;
    // (flag_value is written to *flag by the first workgroup as it starts: see zrk_exchange, hand-over by flag)
    // batched ensemble of independent scenarios (zrk_run_ticks_ensemble): scenario s owns rows_ps consecutive rows
    // (bps row blocks), its radar records are block s of rb_table, its noise key is seeds[s]; lists restart per scenario
    int64_t rows_ps;
    // stamps (head): wall-clock stamps of THIS launch (s_memrealtime, 100 MHz), NULL: none -- words [0, kStampBegins): when each
    // wave of the first workgroups started; word kStampBegins + w: when wave w of the grid ended.  The launch ran from the
    // smallest of the former to the largest of the latter (k_reduce_stamps): a duration measured without an event, a signal or
    // a profiler on the stream (zrk_sweep_stamps)
    RadarBlock rb;
};
// a PAIR launch carries the second tick's records behind the first's (a plain launch does not pay for their 10 KB)
struct SweepParamsPair : SweepParams {
    RadarBlock rb2;
};
// (rb is SweepParams' last member: its offset without offsetof, which a struct with a base does not officially have)
constexpr size_t kSweepRbOffset = sizeof(SweepParams) - sizeof(RadarBlock);
static_assert(sizeof(SweepParams) % 8 == 0 && sizeof(SweepParamsPair) == sizeof(SweepParams) + sizeof(RadarBlock), "rb2 sits right behind SweepParams");

// The missile phase rides along in other kernels' grids (its own launches would cost more in kernel
// boundaries than in work): the per-row step as extra workgroups of the sweep, the ordered event list
// and the tombstones as an extra workgroup of the compaction.  m == 0: nothing to do.
struct MissileArgs {
    const double *sp, *vel, *t0;
    uint8_t *alive;
    const int32_t *lidx;
    const double *pos_cur;
    double *pos_prev;
    int64_t cap;
    const int32_t *m_slot, *m_tgt;
    const double *m_radius;
    double *m_period;
    uint8_t *m_status, *ev_code;
    int32_t *ev_missile, *ev_target, *ev_count;
    int64_t m;
    double t, dts;
    int32_t apply, ev_wire_cap;
    int64_t *ev_wire;           // tail of the exchange list: [count, (missile index << 32 | target index or 0xFFFFFFFF) ...]
    int64_t gid0;               // global list index of list element 0 (indices on the wire are global)
    uint8_t *pend;              // overlapped loop: removals as marks (see SweepParams::pend); NULL: tombstones by the finisher
    uint32_t mark, _pad3;       // this tick's mark value
    // per-row gather records, 64 bytes: start_pos, velocity, start_time, list index -- what a missile needs of its TARGET
    // from one cache line instead of eight (ten thousand missiles' gathers were a tenth of the sweep's traffic); kept
    // by zrk_run_ticks (ensure_gather_records), NULL: the columns
    const double *grec;
    const double *pos_abs[2];   // the two position buffers by absolute index: a removed target's frozen position is in
                                // pos_abs[its mark & 1] (SweepParams::pend)
    // PAIR launch (see SweepParams): the missile workgroups step tick t, meet at a barrier of their own (bar[0] counts
    // arrivals up to bar_target; bar[1]: somebody gave up) so that tick t's marks are everybody's, and step tick t + 1
    uint8_t *ev_code2;          // tick t + 1's per-row event codes
    uint32_t mark2, bar_target;
    uint32_t *bar;
    double t2;
    // compaction side: the event builder of a pair's FIRST tick clears, in the second tick's mask buffer, the bits of the rows
    // its events removed (they were swept once more as if nothing had happened)
    uint32_t *clear_vis;
    // ... or, where the pair's two compactions are ONE launch (k_compact_pair), a list the sweep's missile threads append the
    // removed rows' list indices to as they decide the first tick's events: rm[0] = entries, rm_cap of them fit
    int32_t *rm;
    int32_t rm_cap, _pad4;
    // where a row that leaves the air keeps the prev_pos its handle held (zrk_ctx_keep_prev: [capacity][3], NaN = still in the air):
    // the tombstone below makes both position buffers the row's last position; the reference's handle keeps pos AND prev_pos of
    // its last step, and the command post's link_object reads the latter of every track it ever made (modules/CCP.py:196)
    double *frozen_prev;
};

// Dispatch order of the next sweep.  The sweep's duration is set by the expensive waves (rows inside some
// sector: ten-odd radars walked one after the other, noise drawn) that start last; which row blocks are
// expensive changes slowly from tick to tick (the sectors turn a few degrees), so each sweep leaves next tick's
// order behind as it goes: the first wave of every workgroup takes a number from one of two counters -- blocks with
// a radar to walk count up from the front of a list, the others down from its end -- and writes its block there.
// Four thousand atomics on two words would take 50 us (they serialise at ~12 ns each), so the list is kOrderRegions
// interleaved lists (workgroup bid belongs to list bid % kOrderRegions, whose q-th entry is entry q * kOrderRegions +
// bid % kOrderRegions of the whole), each with a pair of counters on cache lines of their own: expensive blocks still
// come first, list by list.  Purely a schedule: every block is swept exactly once whatever the order says (and
// whichever order the workgroups happened to take their numbers in).  A launch of its own for this (sorting the
// per-block costs) took 2.7 us of every tick on the compute stream.
constexpr int kOrderRegions = 64;
constexpr int kOrderCtrStride = 32;                     // words between counters: 128 bytes
constexpr int kOrderCtrSet = kOrderRegions * 2 * kOrderCtrStride;

__device__ __forceinline__ double dot3(double ax, double ay, double az, double bx, double by, double bz)
{
    return __builtin_fma(az, bz, __builtin_fma(ay, by, ax * bx));
}

// a % b of numpy / CPython for |a| < b, b > 0:  fmod is the identity there.
__device__ __forceinline__ double floormod_small(double a, double b)
{
    double m = a;
    if (m != 0.0) {
        if (m < 0.0) m += b;
    } else {
        m = 0.0;                // -0.0 % b == +0.0
    }
    return m;
}

// modules/Radar.py:56-70 in binary64: the decision every guard-band pair falls back to.  Two tiers.  The reference
// compares angles that numpy's arctan2 / arcsin return, which costs a few thousand cycles per call here -- and a
// row that lies within the binary32 guard band of a face shared by several radars (sixteen radars in a line with
// parallel sectors share the line itself) pays it once per radar, with its whole wave waiting.  So first the
// same signed distance to the nearest face as the fast path, in binary64: edge vectors and cone sines rounded from
// the exact bounds (relative error ~1e-16), sqrt correctly rounded, everything else a handful of roundings --
// whatever is farther than 1e-11 of the distance from every face is on the side it appears to be on, also for the
// angles (their own rounding is ~1e-15 relative).  Only rows inside that band (nanometres) take the angle formula.
// Device-side "this must not happen" word, read by zrk_compact_status.  Bit 0: visible_exact was handed a record
// address without a high word.  THE ADDRESS OF A RADAR RECORD IS FORMED IN THE __global__ BODY ONLY (k_tick_sweep: `rbp`):
// outside a kernel __builtin_amdgcn_kernarg_segment_ptr() is the constant 0 -- the backend lowers the intrinsic to
// null in every function that is not an entry point, without a diagnostic -- and a record "at" 0 + offsetof(SweepParams,
// rb.cold) + r * sizeof(RadarCold) is a load from page 0x1000: the memory access fault of round 2.
__device__ uint32_t g_device_fault;

__device__ __forceinline__ bool visible_exact_body(uint64_t cold_record, double dx, double dy, double dz)
{
    // radar r's cold record in the kernel-argument segment, by address (a by-value copy of the record would travel
    // through scratch memory); the address is wave-uniform, so the fourteen words come in through scalar loads
    typedef const double __attribute__((address_space(4))) *ConstDoubles;
    const uint64_t addr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(cold_record >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cold_record);
    if ((uint32_t)(addr >> 32) == 0u) {            // (wave-uniform) never a device or kernarg address: do not touch it
        atomicOr(&g_device_fault, 1u);
        return false;
    }
    const ConstDoubles q = (ConstDoubles)addr;
    static_assert(sizeof(RadarCold) == 14 * sizeof(double), "RadarCold is fourteen doubles");
    RadarCold c;
    c.d2_max = q[0]; c.az_lo = q[1]; c.az_hi = q[2]; c.el_lo = q[3]; c.el_hi = q[4];
    c.elx = q[5]; c.ely = q[6]; c.ehx = q[7]; c.ehy = q[8];
    c.s_lo_up = q[9]; c.s_hi_up = q[10]; c.s_lo_dn = q[11]; c.s_hi_dn = q[12]; c.az_sgn = q[13];
    const double d2 = dot3(dx, dy, dz, dx, dy, dz);
    if (d2 > c.d2_max) return false;                           // == `distance > max_distance`
    const double dist = sqrt(d2);
    if (c.az_sgn != 0.0) {
        const double cl = c.elx * dy - c.ely * dx, ch = dx * c.ehy - dy * c.ehx;
        const double m_az = c.az_sgn * fmin(cl, ch);                  // az_sgn is +-1 here
        const bool up = dz >= 0.0;
        const double a = dz - (up ? c.s_lo_up : c.s_lo_dn) * dist, b = (up ? c.s_hi_up : c.s_hi_dn) * dist - dz;
        const double t = fmin(fmin(m_az, a), b);
        const double g = 1e-11 * dist;
        if (t > g) return true;
        if (t < -g) return false;
    }
    const double az = floormod_small(atan2(dy, dx) * kRad2Deg, 360.0);
    const double el = floormod_small(asin(dz / dist) * kRad2Deg, 180.0);
    return (c.az_lo <= az) && (az <= c.az_hi) && (c.el_lo <= el) && (el <= c.el_hi);
}

// Out of line: a rare path inside the sweep's hottest loop, kept small there.  (Inline it costs the kernel 126 vector
// registers; what lives across a call sits in the callee-saved ranges, so calls are not free either: see the replay.)
__device__ __noinline__ bool visible_exact(uint64_t cold_record, double dx, double dy, double dz)
{
    return visible_exact_body(cold_record, dx, dy, dz);
}

// ---------------------------------------------------------------------------------------------
// Per-radar derivations, host and device: the host runs them for the radars of one scenario every tick (they
// travel in the kernel-argument segment); for a batched ensemble of scenarios a workgroup riding in the compaction
// runs them on the device for the next tick (k_compact_fused, EnsembleArgs).
// ---------------------------------------------------------------------------------------------
// CPython / numpy float floor-mod (modules/Radar.py:103, :109, :114, :117).
__host__ __device__ inline double floormod_py(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

// SectorRadar.move_to_next_sector_circular (modules/Radar.py:96-117) for one radar.
__host__ __device__ inline void scan_advance_one(zrk_radar &rd, const zrk_scan &sc)
{
    if (sc.mode == 0) {            // "horizontal"
        if (rd.cur_azimuth + rd.azimuth_range < 360.0) rd.cur_azimuth = floormod_py(rd.cur_azimuth + sc.azimuth_speed, 360.0);
        else rd.cur_azimuth = sc.elevation_start;           // sic, modules/Radar.py:105
        if (rd.cur_azimuth < sc.azimuth_speed) {
            if (rd.cur_elevation + sc.elevation_speed < 90.0) rd.cur_elevation = floormod_py(rd.cur_elevation + sc.elevation_speed, 90.0);
            else rd.cur_elevation = sc.elevation_start;
        }
    } else if (sc.mode == 1) {     // "vertical"
        rd.cur_elevation = floormod_py(rd.cur_elevation + sc.elevation_speed, 90.0);
        if (rd.cur_elevation < sc.elevation_speed) rd.cur_azimuth = floormod_py(rd.cur_azimuth + sc.azimuth_speed, 360.0);
    }                               // any other mode string: the reference does nothing
}

// The host derives the records of all radars of a scenario twice per launch, and radars of one battery often share their
// angles (the benchmark's sixteen turn in phase): the last few sines are kept, by the argument's bits -- the same calls, the
// same results, a tenth of the trigonometry (6 us per sixteen radars otherwise, on the thread that launches the sweeps).
struct TrigMemo {
    int ns = 0, nsc = 0;
    double sx[8], sv[8], scx[8], scs[8], scc[8];
};

__host__ __device__ inline double memo_sin(TrigMemo *m, double x)
{
#ifndef __HIP_DEVICE_COMPILE__
    if (m) {
        for (int k = 0; k < (m->ns < 8 ? m->ns : 8); ++k)
            if (std::memcmp(&m->sx[k], &x, sizeof(x)) == 0) return m->sv[k];
        const double v = sin(x);
        const int k = m->ns++ & 7;
        m->sx[k] = x; m->sv[k] = v;
        return v;
    }
#endif
    return sin(x);
}

__host__ __device__ inline void memo_sincos(TrigMemo *m, double x, double *sv, double *cv)
{
#ifndef __HIP_DEVICE_COMPILE__
    if (m) {
        for (int k = 0; k < (m->nsc < 8 ? m->nsc : 8); ++k)
            if (std::memcmp(&m->scx[k], &x, sizeof(x)) == 0) { *sv = m->scs[k]; *cv = m->scc[k]; return; }
        sincos(x, sv, cv);
        const int k = m->nsc++ & 7;
        m->scx[k] = x; m->scs[k] = *sv; m->scc[k] = *cv;
        return;
    }
#endif
    sincos(x, sv, cv);
}

__host__ __device__ inline void derive_radar(const zrk_radar &hr, double d2_max, bool exact_only, RadarHot &h, RadarCold &c, TrigMemo *memo = nullptr)
{
    const double deg = 3.14159265358979323846 / 180.0;
    h = RadarHot{};
    h.px = hr.pos[0]; h.py = hr.pos[1]; h.pz = hr.pos[2];
    c.d2_max = d2_max;                                  // d2_threshold(max_distance), from the host
    c.az_lo = hr.cur_azimuth; c.az_hi = hr.cur_azimuth + hr.azimuth_range;
    c.el_lo = hr.cur_elevation; c.el_hi = hr.cur_elevation + hr.elevation_range;
    // float32 pre-gate: d2 computed in binary32 from rounded differences is within ~1e-6 relative.
    // "always binary64" is encoded as d2f_in = -1 (every in-range pair counts as shell),
    // "never visible" as d2f_out = -1 (nothing is in range).
    const bool plain_range = __builtin_isfinite(c.d2_max) && c.d2_max > 0.0 && c.d2_max < 1e30;
    h.d2f_in = plain_range ? (float)(c.d2_max * (1.0 - 1e-5)) : -1.f;
    h.d2f_out = plain_range ? (float)(c.d2_max * (1.0 + 1e-5)) : INFINITY;
    h.s_lo_up = h.s_lo_dn = 2.f; h.s_hi_up = h.s_hi_dn = -2.f;
    c.s_lo_up = c.s_lo_dn = 2.0; c.s_hi_up = c.s_hi_dn = -2.0;
    c.elx = c.ely = c.ehx = c.ehy = 0.0; c.az_sgn = 0.0;
    h.az_guard = plain_range ? (float)(kGuard * sqrt(c.d2_max) * 1.001) : INFINITY;
    h.az_sgn = 1.f;
    const bool finite = __builtin_isfinite(c.az_lo) && __builtin_isfinite(c.az_hi) && __builtin_isfinite(c.el_lo) && __builtin_isfinite(c.el_hi);
    if (!finite || exact_only) h.d2f_in = -1.f;   // decide every in-range pair in binary64
    if (!finite) return;
    // azimuth lives in [0, 360]; the comparison has no wrap-around (modules/Radar.py:67-68)
    const double lo = fmax(c.az_lo, 0.0), hi = fmin(c.az_hi, 360.0);
    if (!(lo <= hi) || c.d2_max < 0.0) { h.d2f_out = -1.f; return; }
    // width <= 180: inside <=> cross(e_lo,p) > 0 and cross(p,e_hi) > 0; wider: either one, which is
    // min(-c1,-c2) < 0, so the edges are stored negated and az_sgn = -1 flips the minimum back
    h.az_sgn = (hi - lo <= 180.0) ? 1.f : -1.f;
    double s_lo, c_lo, s_hi, c_hi;
    memo_sincos(memo, lo * deg, &s_lo, &c_lo);
    memo_sincos(memo, hi * deg, &s_hi, &c_hi);
    c.az_sgn = exact_only ? 0.0 : (double)h.az_sgn;        // ZRK_F_EXACT_ONLY: the reference's formula and nothing else
    c.elx = (double)h.az_sgn * c_lo; c.ely = (double)h.az_sgn * s_lo;
    c.ehx = (double)h.az_sgn * c_hi; c.ehy = (double)h.az_sgn * s_hi;
    h.elx = (float)c.elx; h.ely = (float)c.ely; h.ehx = (float)c.ehx; h.ehy = (float)c.ehy;
    // elevation: dz >= 0 -> el = theta in [0,90];  dz < 0 -> el = 180 + theta in [90,180]
    const double lo_u = fmax(c.el_lo, 0.0), hi_u = fmin(c.el_hi, 90.0);
    if (lo_u <= hi_u) {
        h.s_lo_up = (c.el_lo <= 0.0) ? -2.f : (float)memo_sin(memo, lo_u * deg);
        h.s_hi_up = (c.el_hi >= 90.0) ? 2.f : (float)memo_sin(memo, hi_u * deg);
        c.s_lo_up = (c.el_lo <= 0.0) ? -2.0 : memo_sin(memo, lo_u * deg);
        c.s_hi_up = (c.el_hi >= 90.0) ? 2.0 : memo_sin(memo, hi_u * deg);
    }
    const double lo_d = fmax(c.el_lo - 180.0, -90.0), hi_d = fmin(c.el_hi - 180.0, 0.0);
    if (lo_d <= hi_d) {
        h.s_lo_dn = (c.el_lo - 180.0 <= -90.0) ? -2.f : (float)memo_sin(memo, lo_d * deg);
        h.s_hi_dn = (c.el_hi >= 180.0) ? 2.f : (float)memo_sin(memo, hi_d * deg);
        c.s_lo_dn = (c.el_lo - 180.0 <= -90.0) ? -2.0 : memo_sin(memo, lo_d * deg);
        c.s_hi_dn = (c.el_hi >= 180.0) ? 2.0 : memo_sin(memo, hi_d * deg);
    }
}

// Bounds of the pre-pass.  Outside: a row that, seen from its position BEFORE this tick's noise, is farther than
// d2_out from the radar or more than az_out outside the azimuth wedge cannot be in the sector whatever
// happens earlier in the tick.  Inside: a box that keeps t_in metres from every face of the sector and whose
// farthest corner is closer than sqrt(d2_in) holds only rows the radar certainly sees.  Slack: every detection
// moves a row by at most kNoiseReach (Box-Muller on 16-bit uniforms: radius <= 5 * sqrt(-2 ln(2^-17)) = 24.3 m in
// the x-y plane and along z, 34.4 m in space), and radar number `index` looks after at most `index` of them;
// the distance to a wedge face and to the range sphere are 1-Lipschitz in the position, the elevation margins
// dz -+ s * dist 2-Lipschitz.  Binary32 coordinates are off by 2^-24 relative per axis: covered by
// 4e-6 * (|radar| + reach) + 1 m and the factor on the square.  Degenerate radars keep the encoding of their
// hot record (d2f_out < 0: never visible; infinite: every row is a candidate) and are never "certainly inside".
__host__ __device__ inline void derive_pre(const zrk_radar &hr, const RadarHot &h, bool philox, int index, RadarPre &p)
{
    constexpr double kNoiseReach = 34.4;
    p = RadarPre{};
    p.px = (float)hr.pos[0]; p.py = (float)hr.pos[1]; p.pz = (float)hr.pos[2];
    p.elx = h.elx; p.ely = h.ely; p.ehx = h.ehx; p.ehy = h.ehy; p.az_sgn = h.az_sgn;
    p.s_lo_up = h.s_lo_up; p.s_hi_up = h.s_hi_up; p.s_lo_dn = h.s_lo_dn; p.s_hi_dn = h.s_hi_dn;
    p.az_out = INFINITY;
    p.d2_in = -1.f; p.t_in = INFINITY; p.z_in = INFINITY; p.pz64 = hr.pos[2];
    if (h.d2f_out < 0.f) { p.d2_out = -1.f; return; }
    p.d2_out = INFINITY;
    if (__builtin_isinf(h.d2f_out) || __builtin_isnan(h.d2f_out)) return;
    const double reach = sqrt((double)h.d2f_out);
    const double centre = fabs(hr.pos[0]) + fabs(hr.pos[1]) + fabs(hr.pos[2]);
    if (!__builtin_isfinite(centre)) return;
    const double slack = (philox ? kNoiseReach * (index > 0 ? index : 0) : 0.0) + 4e-6 * (centre + reach) + 1.0;
    const double b = (reach + slack) * (reach + slack) * (1.0 + 1e-5);
    if (b < 3e38) p.d2_out = (float)b;
    const double a = ((double)h.az_guard + slack) * (1.0 + 1e-5);
    if (__builtin_isfinite(a) && a < 3e38) p.az_out = (float)a;
    // inside: only for radars whose every in-range pair may be settled in binary32 (d2f_in > 0)
    if (h.d2f_in > 0.f) {
        const double inner = sqrt((double)h.d2f_in) - slack;
        const double t = ((double)h.az_guard + 2.0 * slack) * (1.0 + 1e-5);
        if (inner > 0.0 && __builtin_isfinite(t) && t < 3e38) {
            p.d2_in = (float)(inner * inner * (1.0 - 1e-5));
            p.t_in = (float)t;
            // along z a draw moves a row by at most 24.3 m
            p.z_in = (float)(((philox ? 24.3 * (index > 0 ? index : 0) : 0.0) + 4e-6 * (centre + reach) + 1.0) * (1.0 + 1e-5));
        }
    }
}

// Philox4x32-10, counter (entity lo, entity hi, tick, radar), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (both halves of a product from one 64-bit multiply-add: the 32-bit multiplies are quarter-rate instructions, forty
        // of them a seeding)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Measurement-noise stream of the throughput mode.  One Philox4x32-10 block per (entity, tick)
// seeds a xoshiro128++ state; every detection of that entity in that tick (radar after radar) draws
// two words from it = four 16-bit uniforms = two Box-Muller pairs in binary32 on the hardware
// log2 / sqrt / sin / cos (v_sin/v_cos take revolutions), of which three values are used.
struct NoiseState {
    uint32_t s0, s1, s2, s3;
};

__device__ __forceinline__ NoiseState noise_init(uint64_t seed, uint64_t tick, uint64_t entity)
{
    uint32_t x[4];
    philox4x32_10((uint32_t)entity, (uint32_t)(entity >> 32), (uint32_t)tick, (uint32_t)(tick >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), x);
    return NoiseState{x[0], x[1], x[2], x[3]};
}

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int k) { return (v << k) | (v >> (32 - k)); }

// One step of the xoshiro128 state (Blackman & Vigna), two 32-bit outputs per step: the "++" scrambler on
// (s0, s3) and the same scrambler on the other half of the state, (s1, s2).
__device__ __forceinline__ void noise_next2(NoiseState &st, uint32_t &a, uint32_t &b)
{
    a = rotl32(st.s0 + st.s3, 7) + st.s0;
    b = rotl32(st.s1 + st.s2, 13) + st.s2;
    const uint32_t t = st.s1 << 9;
    st.s2 ^= st.s0; st.s3 ^= st.s1; st.s1 ^= st.s2; st.s0 ^= st.s3;
    st.s2 ^= t;
    st.s3 = rotl32(st.s3, 11);
}

// Three N(0, 5^2) values: two Box-Muller pairs from four 16-bit uniforms, on the hardware log2 / sqrt / sin / cos
// (v_sin / v_cos take revolutions).  r = 5 sqrt(-2 ln((h + 0.5) / 65536)) = sqrt(kC1 * log2(h + 0.5) + kC0).
__device__ __forceinline__ void noise_draw3(NoiseState &st, float out[3])
{
    constexpr float kC1 = -34.657359027997266f;                       // -2 * 25 * ln 2
    constexpr float kC0 = 554.51774444795626f;                        // 32 * 25 * ln 2
    uint32_t a, b;
    noise_next2(st, a, b);
    const float k16 = 1.52587890625e-5f;                              // 2^-16
    const float l0 = __builtin_amdgcn_logf((float)(a >> 16) + 0.5f), u1 = (float)(a & 0xFFFFu) * k16;
    const float l1 = __builtin_amdgcn_logf((float)(b >> 16) + 0.5f), u3 = (float)(b & 0xFFFFu) * k16;
    const float r0 = __builtin_amdgcn_sqrtf(__builtin_fmaf(l0, kC1, kC0));
    const float r1 = __builtin_amdgcn_sqrtf(__builtin_fmaf(l1, kC1, kC0));
    out[0] = r0 * __builtin_amdgcn_cosf(u1);
    out[1] = r0 * __builtin_amdgcn_sinf(u1);
    out[2] = r1 * __builtin_amdgcn_cosf(u3);
}

// ---------------------------------------------------------------------------------------------
// Fused advance + radar sweep.  One thread per entity slot, ZRK_BLOCK slots per workgroup.
// ---------------------------------------------------------------------------------------------
// The CU has ONE scalar unit for its four SIMDs, so the loop body keeps scalar work minimal: one
// scalar-load burst per radar, no uniform branches (degenerate radars are encoded in the thresholds
// by the host), predicates folded into two float minima, and only two divergent regions -- the
// binary64 fallback for guard-band pairs and the noise draw for detections.
__device__ __forceinline__ uint8_t missile_step_row(const double *__restrict__ sp, const double *__restrict__ vel,
                                    const double *__restrict__ t0, const uint8_t *alive,
                                    const int32_t *__restrict__ lidx, const double *pos_prev, int64_t cap,
                                    const int32_t *__restrict__ m_slot, const int32_t *__restrict__ m_tgt,
                                    const double *__restrict__ m_radius, double *__restrict__ m_period,
                                    uint8_t *__restrict__ m_status, int64_t row, double t, double dts,
                                    uint8_t *pend = nullptr, uint32_t mark = 0, const double *grec = nullptr,
                                    const double *pos_abs0 = nullptr, const double *pos_abs1 = nullptr, uint32_t mark_first = 0,
                                    const char *rb_first = nullptr, int R = 0, bool philox = false, uint64_t seed = 0,
                                    uint64_t tick_first = 0, int64_t gid0 = 0, double t_first = 0.0);

// Horizontal bounding box of the wave: two minima and two maxima over the 64 lanes, wave-uniform on return.
// min / max are idempotent, so rotations inside each row of 16 (by 1, 2, 4, 8) and the two row broadcasts
// may overlap freely; lane 63 ends up with the result of all four rows.  One v_min/v_max with a DPP source
// per step; the four reductions are interleaved, which also covers the wait states a DPP read needs after a
// VALU write of the same register.
__device__ __forceinline__ void wave_bbox(float &lx, float &ly, float &hx, float &hy)
{
#define ZRK_BBOX_STEP(ctrl)                                                                                \
    asm volatile("s_nop 0\n\t"                                                                            \
                 "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                        \
                 "v_min_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                        \
                 "v_max_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                        \
                 "v_max_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf"                             \
                 : "+v"(lx), "+v"(ly), "+v"(hx), "+v"(hy))
    asm volatile("s_nop 1" ::: );
    ZRK_BBOX_STEP("row_ror:1");
    ZRK_BBOX_STEP("row_ror:2");
    ZRK_BBOX_STEP("row_ror:4");
    ZRK_BBOX_STEP("row_ror:8");
    ZRK_BBOX_STEP("row_bcast:15");
    ZRK_BBOX_STEP("row_bcast:31");
#undef ZRK_BBOX_STEP
    lx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lx), 63));
    ly = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ly), 63));
    hx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hx), 63));
    hy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hy), 63));
}

// One minimum and one maximum over the wave, same scheme (the vertical extent of the box, taken only by waves
// some radar may see).
__device__ __forceinline__ void wave_minmax(float &lo, float &hi)
{
#define ZRK_MM_STEP(ctrl)                                                                                  \
    asm volatile("s_nop 1\n\t"                                                                            \
                 "v_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"                        \
                 "v_max_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf"                             \
                 : "+v"(lo), "+v"(hi))
    asm volatile("s_nop 1" ::: );
    ZRK_MM_STEP("row_ror:1");
    ZRK_MM_STEP("row_ror:2");
    ZRK_MM_STEP("row_ror:4");
    ZRK_MM_STEP("row_ror:8");
    ZRK_MM_STEP("row_bcast:15");
    ZRK_MM_STEP("row_bcast:31");
#undef ZRK_MM_STEP
    lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), 63));
    hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi), 63));
}

// The pre-pass table of the workgroup: 32 records of 80 bytes in LDS, staged once per workgroup from the
// kernel-argument segment (the sweep parameters are its first bytes) -- one 16-byte piece per thread, issued
// before the row loads so that nobody waits for it -- instead of every wave fetching lane r's record with
// vector loads of its own (8.5 MB per launch at C3, PMC).  Lane r then reads record r with ds_read_b128
// (80-byte stride: conflict-free within each group of 16 lanes).
struct PreTable {
    uint4 v[kPreVec];
};

// Issued as the wave's first vector load and waited for by count (stage_pre_table: all but the `younger` loads
// issued since), so that the row loads behind it stay in flight; the compiler does not see the load and would
// otherwise sink it below them and wait for everything.
__device__ __forceinline__ uint4 pre_table_fetch(const char *rbp)
{
    const char *tab = rbp + offsetof(RadarBlock, prew);
    const uint32_t off = (threadIdx.x < kPreVec ? threadIdx.x : 0u) * 16u;
    uint4 piece;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(piece) : "v"(off), "s"(tab) : "memory");
    return piece;
}

template <int YOUNGER>
__device__ __forceinline__ void stage_pre_table(PreTable &T, uint4 &piece)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(piece.x), "+v"(piece.y), "+v"(piece.z), "+v"(piece.w) : "n"(YOUNGER) : "memory");
    if (threadIdx.x < kPreVec) T.v[threadIdx.x] = piece;
    __syncthreads();
}

// Box cache.  What a workgroup needs to know before its rows arrive is where they can be, not where they are:
// every row moves along a straight line (modules/AirObject.py:23-25), so the bounding box of the row block's live
// rows taken at some earlier time, grown by the block's largest speed per axis times the time since, still holds
// them.  A workgroup whose block has such a record (48 bytes in the workspace, fetched with scalar loads) lets
// its first wave run the whole box-level classification below -- which radars can see any of the block's rows, which
// certainly see all of them -- while the row loads are in flight and hands the three masks to the others through
// LDS; a block no radar can reach (four in five at C3) has nothing left to do once its rows land but the three
// roundings of the advance and the stores.  A block without a usable record (first tick, rows appended since,
// record grown by more than kBoxMaxGrow) has each wave take the box of the positions it has just computed (DPP
// reductions) and classify late, and leaves a fresh record behind.  Conservative by construction: the
// classification only ever uses "no point of the box can be in the sector" / "every point of the box is".
struct WaveBox {
    float lo[3], hi[3];          // box of the live rows' positions at time t_ref (before any measurement noise)
    float vmax[3];               // largest |velocity component| over the block's rows [m/s]
    uint32_t state;              // 0: none, 1: valid, 2: valid but some live row is not finite (never culled), 3: no live row
    double t_ref;
};
static_assert(sizeof(WaveBox) == 48, "WaveBox must be 12 dwords");
constexpr float kBoxMaxGrow = 192.f;                   // metres a cached box may have grown per side before it is retaken
constexpr uint32_t kBoxOk = 1u, kBoxWild = 2u, kBoxEmpty = 3u;

struct Cull {
    uint32_t cand;               // radars that may see some row of the wave
    uint32_t inside;             // ... that certainly see every live row
    uint32_t plane;              // ... that certainly see every live row above their own horizontal plane, none below
    uint32_t pz_lo, pz_hi;       // lane r: radar r's height (binary64 halves), for the plane test
};

// Lane r puts the wave's box to radar r.  Out: closest approach beyond the range sphere, or the whole box
// beyond an edge of the azimuth wedge by more than noise can bridge.  In: farthest corner in range, lower bounds of
// the wedge cross products and of the elevation margins dz - s_lo * dist, s_hi * dist - dz above every guard band
// plus what this tick's noise can add before radar r looks (host: derive_pre), box wholly above or wholly
// below the radar.  The commonest sector floor is the radar's own horizontal plane (elevation from 0 degrees,
// nothing visible below): rows may sit arbitrarily close to that face, so there it is left to a per-row test -- the
// sign of dz in binary64, which is all the reference's arcsin decides -- while range, wedge and the upper cone are
// settled for the part of the box above the plane.
__device__ __forceinline__ Cull cull_box(const PreTable &T, int R, bool shortcuts, float blx, float bly, float blz,
                                         float bhx, float bhy, float bhz)
{
    Cull c;
    const int rl = (int)(threadIdx.x & (ZRK_MAX_RADARS - 1)) * (int)(sizeof(RadarPre) / 16);
    const uint4 w0 = T.v[rl], w1 = T.v[rl + 1], w2 = T.v[rl + 2], w3 = T.v[rl + 3], w4 = T.v[rl + 4];
    const float qpx = __builtin_bit_cast(float, w0.x), qpy = __builtin_bit_cast(float, w0.y);
    const float qpz = __builtin_bit_cast(float, w0.z), qd2_out = __builtin_bit_cast(float, w0.w);
    const float qelx = __builtin_bit_cast(float, w1.x), qely = __builtin_bit_cast(float, w1.y);
    const float qehx = __builtin_bit_cast(float, w1.z), qehy = __builtin_bit_cast(float, w1.w);
    const float qaz_sgn = __builtin_bit_cast(float, w2.x), qaz_out = __builtin_bit_cast(float, w2.y);
    const float qd2_in = __builtin_bit_cast(float, w2.z), qt_in = __builtin_bit_cast(float, w2.w);
    const float s_lo_up = __builtin_bit_cast(float, w3.x), s_hi_up = __builtin_bit_cast(float, w3.y);
    const float s_lo_dn = __builtin_bit_cast(float, w3.z), s_hi_dn = __builtin_bit_cast(float, w3.w);
    const float qz_in = __builtin_bit_cast(float, w4.x);
    c.pz_lo = w4.z; c.pz_hi = w4.w;
    const float ex_lo = blx - qpx, ex_hi = bhx - qpx, ey_lo = bly - qpy, ey_hi = bhy - qpy;
    const float ez_lo = blz - qpz, ez_hi = bhz - qpz;
    const float gx = fmaxf(fmaxf(ex_lo, -ex_hi), 0.f), gy = fmaxf(fmaxf(ey_lo, -ey_hi), 0.f);
    const float gz = fmaxf(fmaxf(ez_lo, -ez_hi), 0.f);
    // the cull's closest approach is horizontal only: the thresholds of derive_pre are horizontal distances
    const float d2min = __builtin_fmaf(gy, gy, gx * gx);
    // cl = elx * ey - ely * ex,  ch = ehy * ex - ehx * ey  over the box
    const float a1 = qelx * ey_lo, a2 = qelx * ey_hi, b1 = qely * ex_lo, b2 = qely * ex_hi;
    const float c1 = qehy * ex_lo, c2 = qehy * ex_hi, d1 = qehx * ey_lo, d2 = qehx * ey_hi;
    const float cl_hi = fmaxf(a1, a2) - fminf(b1, b2), cl_lo = fminf(a1, a2) - fmaxf(b1, b2);
    const float ch_hi = fmaxf(c1, c2) - fminf(d1, d2), ch_lo = fminf(c1, c2) - fmaxf(d1, d2);
    // m_az = az_sgn * min(cl, ch) is at most ub and at least lb over the box
    const bool narrow = qaz_sgn > 0.f;
    const float ub = narrow ? fminf(cl_hi, ch_hi) : -fminf(cl_lo, ch_lo);
    const float lb = narrow ? fminf(cl_lo, ch_lo) : -fminf(cl_hi, ch_hi);
    const bool out = (d2min > qd2_out) | (ub < -qaz_out);
    const uint32_t all = (R >= 32) ? 0xFFFFFFFFu : ((1u << R) - 1u);
    c.cand = (uint32_t)__ballot(!out) & all;
    c.inside = 0u; c.plane = 0u;
    if (c.cand && shortcuts) {                            // wave-uniform
        const float mx = fmaxf(fabsf(ex_lo), fabsf(ex_hi)), my = fmaxf(fabsf(ey_lo), fabsf(ey_hi));
        const float mz = fmaxf(fabsf(ez_lo), fabsf(ez_hi));
        const float d2max = __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
        const float dmin = __builtin_amdgcn_sqrtf(__builtin_fmaf(gz, gz, d2min));
        const float dmax = __builtin_amdgcn_sqrtf(d2max);
        const bool up = ez_lo > qz_in, dn = ez_hi < -qz_in;           // wholly above / below the radar
        const float s_lo = up ? s_lo_up : s_lo_dn, s_hi = up ? s_hi_up : s_hi_dn;
        // a = dz - s_lo * dist, b = s_hi * dist - dz over the box, from below
        const float a_min = ez_lo - s_lo * ((s_lo >= 0.f) ? dmax : dmin);
        const float b_min = s_hi * ((s_hi >= 0.f) ? dmin : dmax) - ez_hi;
        const float t_box = fminf(lb, fminf(a_min, b_min));
        const bool ranged = d2max < qd2_in;
        const bool sure = (up | dn) & ranged & (t_box > qt_in);
        const float b_up = s_hi_up * ((s_hi_up >= 0.f) ? dmin : dmax) - ez_hi;
        const bool floor_is_plane = (s_lo_up == -2.f) & (s_lo_dn == 2.f) & (s_hi_dn == -2.f);
        const bool sure_but_plane = floor_is_plane & ranged & (fminf(lb, b_up) > qt_in);
        c.inside = (uint32_t)__ballot(sure) & c.cand;
        c.plane = (uint32_t)__ballot(sure_but_plane) & c.cand & ~c.inside;
    }
    return c;
}

// Radars in `c.cand`, in order, over the wave's rows at (x, y, z); leaves the visibility mask in `mask` and the
// (perturbed) position in place.  Called with the whole wave converged (the early-outs are wave-level votes).
template <bool PHILOX, bool LAZY>
__device__ __forceinline__ void sweep_rows(uint64_t tick, int64_t gid0, const char *rbp, uint64_t seed, const Cull &c, int64_t li,
                                           bool live, double &x, double &y, double &z, uint32_t &mask, int64_t probe_wave)
{
    typedef const uint32_t __attribute__((address_space(4))) *ConstWords;
    // The three masks are the workgroup's (one box per row block, handed round through LDS): scalars, so that the walk's
    // control flow is the scalar unit's.  The noise stream is keyed by list index: layout-independent.  !LAZY (the
    // overlapped loop's kernels): it is seeded once per wave that has any radar to walk (one in four at C3; the ten Philox
    // rounds cost such a wave less than one radar) -- seeding at the first detection made every radar of the walk carry the
    // "seeded yet?" selects, a third of the instructions of a radar that sees the whole block (77 -> 41 vector instructions).
    // LAZY (stand-alone sweeps, the plain loop, ensembles): at the wave's first detection -- small scenarios with few
    // radars have many waves that walk a radar or two and detect nothing.
    const uint32_t cands = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.cand);
    const uint32_t easy = (uint32_t)__builtin_amdgcn_readfirstlane((int)(c.inside | c.plane));
    const uint32_t planes = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.plane);
    NoiseState ns = NoiseState{0u, 0u, 0u, 0u};
    bool seeded = !LAZY;
    const unsigned long long live_m = __ballot(live);
    mask = 0u;
    if (PHILOX && !LAZY && cands) ns = noise_init(seed, tick, (uint64_t)(gid0 + li));
#ifdef ZRK_PROBE_BUILD
    int probe_deep = 0, probe_all = 0, probe_none = 0;   // radars walked to the full classification; ... that saw every live row / none
#endif
    for (uint32_t cand = cands; cand; cand &= cand - 1) {
        const int r = __builtin_ctz(cand);
        if ((easy >> r) & 1u) {                       // every live row (above the radar's plane) is visible: no geometry
            bool vis = live;
            if ((planes >> r) & 1u) {                 // (wave-uniform)
                const double rpz = __builtin_bit_cast(double, ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)c.pz_hi, r) << 32) |
                                                                   (uint32_t)__builtin_amdgcn_readlane((int)c.pz_lo, r));
                vis = live & (z - rpz >= 0.0);
            }
            if (PHILOX && LAZY && !seeded && __ballot(vis)) {
                ns = noise_init(seed, tick, (uint64_t)(gid0 + li));
                seeded = true;
            }
            if (vis) {
                mask |= 1u << r;
                if (PHILOX) {
                    float nz[3];
                    noise_draw3(ns, nz);
                    x += (double)nz[0]; y += (double)nz[1]; z += (double)nz[2];               // modules/Radar.py:142
                }
            }
            continue;
        }
        // one scalar-load burst for the whole hot record, resident in SGPRs before any use
        uint32_t w[18];
        const ConstWords hw = (ConstWords)(uint64_t)(rbp + offsetof(RadarBlock, hotw) + (size_t)r * sizeof(RadarHot));
#pragma unroll
        for (int k = 0; k < 18; ++k) w[k] = hw[k];
        asm volatile("" ::"s"(w[0]), "s"(w[1]), "s"(w[2]), "s"(w[3]), "s"(w[4]), "s"(w[5]), "s"(w[6]), "s"(w[7]),
                     "s"(w[8]), "s"(w[9]), "s"(w[10]), "s"(w[11]), "s"(w[12]), "s"(w[13]), "s"(w[14]), "s"(w[15]),
                     "s"(w[16]), "s"(w[17]));
        const double rpx = __builtin_bit_cast(double, ((uint64_t)w[1] << 32) | w[0]);
        const double rpy = __builtin_bit_cast(double, ((uint64_t)w[3] << 32) | w[2]);
        const double rpz = __builtin_bit_cast(double, ((uint64_t)w[5] << 32) | w[4]);
        const float d2f_in = __builtin_bit_cast(float, w[6]), d2f_out = __builtin_bit_cast(float, w[7]);
        const float elx = __builtin_bit_cast(float, w[8]), ely = __builtin_bit_cast(float, w[9]);
        const float ehx = __builtin_bit_cast(float, w[10]), ehy = __builtin_bit_cast(float, w[11]);
        const float s_lo_up = __builtin_bit_cast(float, w[12]), s_hi_up = __builtin_bit_cast(float, w[13]);
        const float s_lo_dn = __builtin_bit_cast(float, w[14]), s_hi_dn = __builtin_bit_cast(float, w[15]);
        const float az_guard = __builtin_bit_cast(float, w[16]), az_sgn = __builtin_bit_cast(float, w[17]);

        const double dx = x - rpx, dy = y - rpy, dz = z - rpz;
        const float fx = (float)dx, fy = (float)dy, fz = (float)dz;
        const float d2f = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
        const bool ranged = d2f <= d2f_out;
        const bool in_range = live & ranged;
        // float32 range gate.  With rows stored in spatial order the lanes of a wave mostly agree,
        // so a wave none of whose lanes is in range (or, below, anywhere near the azimuth wedge)
        // leaves the radar here instead of paying for the rest of the classification.  (The votes as masks of the
        // comparisons themselves, joined by the scalar unit: a vote on a compound predicate costs two vector instructions
        // to make the predicate a value again.)
        const unsigned long long range_m = __ballot(ranged) & live_m;
        if (!range_m) continue;
        const float cl = __builtin_fmaf(elx, fy, -(ely * fx));               // az_sgn * cross(e_lo, p)
        const float ch = __builtin_fmaf(fx, ehy, -(fy * ehx));               // az_sgn * cross(p, e_hi)
        const float m_az = az_sgn * fminf(cl, ch);                           // > 0 inside the azimuth sector [m]
        if (!(__ballot(!(m_az < -az_guard)) & range_m)) continue;            // every lane certainly outside the wedge
#ifdef ZRK_PROBE_BUILD
        ++probe_deep;
#endif
        const float dist = __builtin_amdgcn_sqrtf(d2f);
        // elevation: el = theta for dz >= 0, 180 + theta for dz < 0 (sign taken in binary64, so a
        // tiny negative dz that rounds to -0.0f still selects the lower-hemisphere bounds)
        const float a_up = __builtin_fmaf(-s_lo_up, dist, fz), b_up = __builtin_fmaf(s_hi_up, dist, -fz);
        const float a_dn = __builtin_fmaf(-s_lo_dn, dist, fz), b_dn = __builtin_fmaf(s_hi_dn, dist, -fz);
        const bool up = dz >= 0.0;
        const float a = up ? a_up : a_dn, b = up ? b_up : b_dn;
        // t: signed distance [m] to the nearest sector face (> 0 inside).  Pairs with |t| within
        // kGuard * dist of a face, and pairs in the thin shell around the range sphere, are decided
        // in binary64; everything else is settled here.  NaN / overflow fall out as "not in range"
        // (degenerate ranges are encoded by the host as d2f_out = inf, d2f_in = -1 -> always exact).
        const float t = fminf(fminf(m_az, a), b);
        const float gd = kGuard * dist;
        bool vis = in_range & (t > gd);
        const bool amb = in_range & ((fabsf(t) <= gd) | !(d2f < d2f_in));
        if (amb) vis = visible_exact((uint64_t)(rbp + offsetof(RadarBlock, cold) + (size_t)r * sizeof(RadarCold)), dx, dy, dz);
#ifdef ZRK_PROBE_BUILD
        probe_all += (__ballot(vis) == __ballot(live)) ? 1 : 0;
        probe_none += (__ballot(vis) == 0ull) ? 1 : 0;
#endif
        if (PHILOX && LAZY && !seeded && __ballot(vis)) {
            ns = noise_init(seed, tick, (uint64_t)(gid0 + li));
            seeded = true;
        }
        if (vis) {
            mask |= 1u << r;
            if (PHILOX) {
                float nz[3];
                noise_draw3(ns, nz);
                x += (double)nz[0]; y += (double)nz[1]; z += (double)nz[2];                   // modules/Radar.py:142
            }
        }
    }
#ifdef ZRK_PROBE_BUILD
    ZRK_WAVE_PROBE(probe_wave, 5, (long long)(probe_deep | (probe_all << 8) | (probe_none << 16)));
#endif
}

// One row's radar phase of tick `tick` once more, by ONE lane: every radar's gate in order by the reference's formula
// (visible_exact -- the tiers in front of it in sweep_rows only ever settle pairs it would settle the same way, which the
// ZRK_F_EXACT_ONLY tests pin), a noise draw per detection from the row's stream.  (x, y, z): in, the row's position by
// its trajectory; out, what the sweep leaves in the position buffer.  For the missile phase of a PAIR launch's second tick,
// which needs such a position of a row whose own thread, somewhere in the same grid, may not have run yet.  Rare (a target
// removed in the pair's first tick, a target behind its missile in the list): the cost does not matter.
struct Vec3d {
    double x, y, z;
};

__device__ __noinline__ Vec3d replay_row_radar_phase(const char *rbp, int R, bool philox, uint64_t seed, uint64_t tick, uint64_t key,
                                                     double x, double y, double z)
{
    NoiseState ns = NoiseState{0u, 0u, 0u, 0u};
    bool seeded = false;
    for (int r = 0; r < R; ++r) {
        const double *hot = (const double *)(rbp + offsetof(RadarBlock, hotw) + (size_t)r * sizeof(RadarHot));   // px, py, pz
        const double dx = x - hot[0], dy = y - hot[1], dz = z - hot[2];
        // (the out-of-line function, not its body: this callee that calls keeps a frame -- its return address in a saved register, 16 bytes
        // per lane, the pair kernel's only scratch memory -- but with the body inlined here the register allocator gives THIS function
        // 196 vector registers, which become the kernel's: two waves per SIMD instead of seven)
        if (!visible_exact((uint64_t)(rbp + offsetof(RadarBlock, cold) + (size_t)r * sizeof(RadarCold)), dx, dy, dz)) continue;
        if (philox) {
            if (!seeded) { ns = noise_init(seed, tick, key); seeded = true; }
            float nz[3];
            noise_draw3(ns, nz);
            x += (double)nz[0]; y += (double)nz[1]; z += (double)nz[2];
        }
    }
    return Vec3d{x, y, z};
}

// The barrier of a PAIR launch's missile workgroups (the `mb` leading workgroups of the grid -- dispatched first, a few
// dozen, resident all at once on a device that holds nothing of this stream's but them): arrivals are counted up to
// `target` (monotonic over launches, compared modulo 2^32); behind it tick t's marks and missile rows are everybody's.
// Bounded: a workgroup that is not joined in time raises bar[1] (zrk_compact_status reports it) and goes on.
__device__ __forceinline__ void missile_pair_barrier(uint32_t *bar, uint32_t target)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's marks and missile rows have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while ((int32_t)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (++spins > (1 << 22)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// The wave-level classification of one tick: from the block's box record grown to time `t` (first wave classifies, the
// others take its result from LDS), or -- no usable record -- from the box of the positions just computed.  Returns whether the
// record was used; `fresh` receives the box of the computed positions when it was not (for the record left behind).
struct FreshBox {
    float blx, bly, blz, bhx, bhy, bhz;
    bool any_wild, any_live;
};

struct CullShared {
    uint32_t cand, inside, plane, pad;
    uint32_t pz_lo[ZRK_MAX_RADARS], pz_hi[ZRK_MAX_RADARS];
};

struct CullOrNot {
    Cull c;
    bool have;
};

__device__ __forceinline__ CullOrNot cull_from_record(const PreTable &T, CullShared &S, int R, bool shortcuts, double t, uint32_t bw0,
                                                      uint32_t bw1, uint32_t bw2, uint32_t bw3, uint32_t bw4, uint32_t bw5, uint32_t bw6,
                                                      uint32_t bw7, uint32_t bv0, uint32_t bstate, uint32_t tref_lo, uint32_t tref_hi)
{
    CullOrNot out;
    out.c.cand = 0u; out.c.inside = 0u; out.c.plane = 0u; out.c.pz_lo = 0u; out.c.pz_hi = 0u;
    out.have = false;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double t_ref = __builtin_bit_cast(double, ((uint64_t)tref_hi << 32) | tref_lo);
    const float age = fabsf((float)(t - t_ref)) * 1.000001f + 1e-6f;      // seconds, rounded up
    const float g0 = __builtin_bit_cast(float, bw6) * age, g1 = __builtin_bit_cast(float, bw7) * age;
    const float g2 = __builtin_bit_cast(float, bv0) * age;
    const float grow = fmaxf(fmaxf(g0, g1), g2);
    if (bstate == kBoxEmpty) { out.have = true; return out; }   // nobody alive at t_ref, nobody is revived: nothing to sweep
    if (!(bstate == kBoxOk && grow <= kBoxMaxGrow)) return out;
    if (wv == 0) {
        // (the 1e-3 m covers the rounding of the grown bounds; derive_pre's slack has a metre for the rest)
        const Cull c0 = cull_box(T, R, shortcuts,
                                 __builtin_bit_cast(float, bw0) - g0 - 1e-3f, __builtin_bit_cast(float, bw1) - g1 - 1e-3f,
                                 __builtin_bit_cast(float, bw2) - g2 - 1e-3f, __builtin_bit_cast(float, bw3) + g0 + 1e-3f,
                                 __builtin_bit_cast(float, bw4) + g1 + 1e-3f, __builtin_bit_cast(float, bw5) + g2 + 1e-3f);
        if (lane == 0) { S.cand = c0.cand; S.inside = c0.inside; S.plane = c0.plane; }
        if (lane < ZRK_MAX_RADARS) { S.pz_lo[lane] = c0.pz_lo; S.pz_hi[lane] = c0.pz_hi; }
    }
    __syncthreads();
    out.c.cand = S.cand; out.c.inside = S.inside; out.c.plane = S.plane;
    out.c.pz_lo = S.pz_lo[lane & (ZRK_MAX_RADARS - 1)]; out.c.pz_hi = S.pz_hi[lane & (ZRK_MAX_RADARS - 1)];
    out.have = true;
    return out;
}

struct CullAndBox {
    Cull c;
    FreshBox fb;
};

__device__ __forceinline__ CullAndBox cull_from_rows(const PreTable &T, int R, bool shortcuts, bool live, double x, double y, double z)
{
    CullAndBox out;
    Cull &c = out.c;
    FreshBox &fb = out.fb;
    c.cand = 0u; c.inside = 0u; c.plane = 0u; c.pz_lo = 0u; c.pz_hi = 0u;
    // the box of the positions just computed (wave-uniform after the reductions)
    const float inf = __builtin_inff(), kBig = 1e30f;
    const float fx0 = (float)x, fy0 = (float)y, fz0 = (float)z;
    const bool wild = live & !((fabsf(fx0) < kBig) & (fabsf(fy0) < kBig) & (fabsf(fz0) < kBig));
    fb.blx = live ? fx0 : inf; fb.bly = live ? fy0 : inf; fb.blz = live ? fz0 : inf;
    fb.bhx = live ? fx0 : -inf; fb.bhy = live ? fy0 : -inf; fb.bhz = live ? fz0 : -inf;
    wave_bbox(fb.blx, fb.bly, fb.bhx, fb.bhy);
    wave_minmax(fb.blz, fb.bhz);
    fb.any_wild = __ballot(wild) != 0ull; fb.any_live = __ballot(live) != 0ull;
    const uint32_t all = (R >= 32) ? 0xFFFFFFFFu : ((1u << R) - 1u);
    if (!fb.any_live) { c.cand = 0u; c.inside = 0u; c.plane = 0u; }
    else if (fb.any_wild) { c.cand = all; c.inside = 0u; c.plane = 0u; }   // min / max drop NaNs: never cull such a wave
    else c = cull_box(T, R, shortcuts, fb.blx, fb.bly, fb.blz, fb.bhx, fb.bhy, fb.bhz);
    return out;
}

constexpr int kStampSlots = 64;                        // launches a call's stamps are kept of (zrk_ctx::stamp_ring)
constexpr int kStampBegins = 64;                       // begin stamps: the waves of the first kStampBegins * 64 / ZRK_BLOCK workgroups

__device__ __forceinline__ void stamp_begin(unsigned long long *stamps)
{
    if (stamps && blockIdx.x < (unsigned)(kStampBegins * 64 / ZRK_BLOCK) && (threadIdx.x & 63) == 0)
        stamps[blockIdx.x * (ZRK_BLOCK / 64) + (threadIdx.x >> 6)] = wall_clock64();
}

__device__ __forceinline__ void stamp_end(unsigned long long *stamps)
{
    if (stamps && (threadIdx.x & 63) == 0)
        stamps[kStampBegins + (int64_t)blockIdx.x * (ZRK_BLOCK / 64) + (threadIdx.x >> 6)] = wall_clock64();
}

// min of the begin stamps, max of the end stamps of one launch (one workgroup): out[0], out[1]
__global__ __launch_bounds__(1024) void k_reduce_stamps(const unsigned long long *stamps, int64_t waves, unsigned long long *out)
{
    __shared__ unsigned long long s_lo[16], s_hi[16];
    unsigned long long lo = ~0ull, hi = 0ull;
    const int64_t nb = waves < kStampBegins ? waves : kStampBegins;
    for (int64_t k = threadIdx.x; k < nb; k += 1024) { const unsigned long long v = stamps[k]; lo = v < lo ? v : lo; }
    for (int64_t k = threadIdx.x; k < waves; k += 1024) { const unsigned long long v = stamps[kStampBegins + k]; hi = v > hi ? v : hi; }
    for (int d = 32; d; d >>= 1) {
        const unsigned long long a = __shfl_xor(lo, d), b = __shfl_xor(hi, d);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; }
        out[0] = lo; out[1] = hi;
    }
}

// One pass over the table: 64 consecutive rows per wave.  ADVANCE / LIDX mirror ZRK_F_ADVANCE and
// list_index != NULL as template parameters so that the row's column loads sit in one basic block and are all in
// flight before anything waits for one.  PAIR: two consecutive ticks in the one pass (SweepParams::t2).
template <bool PHILOX, bool ADVANCE, bool LIDX, bool MARKS = false, bool PAIR = false>
// (amdgpu_num_sgpr: up to 80 scalar registers a compute unit holds eight workgroups, up to 96 seven; the pair variant, which
// calls out of line from its missile phase, would take 102 -- six -- if left alone)
__global__ __launch_bounds__(ZRK_BLOCK) __attribute__((amdgpu_num_sgpr(96))) void k_tick_sweep(const std::conditional_t<PAIR, SweepParamsPair, SweepParams> P, const MissileArgs M)
{
    static_assert(!PAIR || (MARKS && ADVANCE), "a pair launch advances and carries removals as marks");
    ZRK_PROBE_ENTRY();
    // one scenario: the radar records travel in this launch's kernel-argument segment (THE ONLY PLACE where that address is
    // formed: see g_device_fault); a batched ensemble: a table in device memory, indexed below
    const char *const kernarg = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    // the head of the arguments in one burst of scalar loads, waited for once (SweepHead)
    typedef const uint32_t __attribute__((address_space(4))) *ConstWordsK;
    uint32_t hw[kSweepHeadBytes / 4];
    {
        const ConstWordsK kw = (ConstWordsK)(uint64_t)kernarg;
#pragma unroll
        for (int k = 0; k < kSweepHeadBytes / 4; ++k) hw[k] = kw[k];
        // (all forty words resident at this point: left to itself the compiler asks for them in three rounds, by first use)
        static_assert(kSweepHeadBytes / 4 == 40, "the two statements below name every word of the head");
        asm volatile("" ::"s"(hw[0]), "s"(hw[1]), "s"(hw[2]), "s"(hw[3]), "s"(hw[4]), "s"(hw[5]), "s"(hw[6]), "s"(hw[7]), "s"(hw[8]), "s"(hw[9]),
                     "s"(hw[10]), "s"(hw[11]), "s"(hw[12]), "s"(hw[13]), "s"(hw[14]), "s"(hw[15]), "s"(hw[16]), "s"(hw[17]), "s"(hw[18]), "s"(hw[19]),
                     "s"(hw[20]), "s"(hw[21]), "s"(hw[22]), "s"(hw[23]), "s"(hw[24]), "s"(hw[25]), "s"(hw[26]), "s"(hw[27]), "s"(hw[28]), "s"(hw[29]));
        asm volatile("" ::"s"(hw[30]), "s"(hw[31]), "s"(hw[32]), "s"(hw[33]), "s"(hw[34]), "s"(hw[35]), "s"(hw[36]), "s"(hw[37]), "s"(hw[38]), "s"(hw[39]),
                     "s"(hw[0]), "s"(hw[10]), "s"(hw[20]));
    }
    SweepHead H;
    __builtin_memcpy(&H, hw, sizeof(H));
    stamp_begin(H.stamps);
    // the previous tick's compaction is over and visible once this grid starts: tell the exchange stream, which waits
    // for this word instead of an event (an event record costs the compute stream a barrier packet per tick)
    if (H.flag && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(H.flag, H.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)blockIdx.x < H.mb) {              // leading workgroups: Missile.step for every in-flight row (a long
        // dependent chain -- dispatched first, it is over long before the sweep's last wave is)
        const int64_t row = (int64_t)blockIdx.x * ZRK_BLOCK + threadIdx.x;
        const char *rb_first = P.rb_table ? P.rb_table : kernarg + kSweepRbOffset;
        if (row < M.m) {
            const uint8_t code = missile_step_row(M.sp, M.vel, M.t0, M.alive, M.lidx, M.pos_prev, M.cap, M.m_slot, M.m_tgt,
                                                  M.m_radius, M.m_period, M.m_status, row, M.t, M.dts, M.pend, M.mark, M.grec,
                                                  M.pos_abs[0], M.pos_abs[1]);
            M.ev_code[row] = code;
            if (PAIR && M.rm && code) {            // the rows this event removes are swept once more below: say which (k_compact_pair)
                const int cnt = code == 1 ? 2 : 1;
                const int k = atomicAdd(M.rm, cnt);
                const int32_t ms = M.m_slot[row], ts = M.m_tgt[row];
                if (k + cnt <= M.rm_cap) {
                    M.rm[1 + k] = M.lidx ? M.lidx[ms] : ms;
                    if (code == 1) M.rm[2 + k] = M.lidx ? M.lidx[ts] : ts;
                }
            }
        }
        if (PAIR) {
            missile_pair_barrier(M.bar, M.bar_target);
            if (row < M.m)
                M.ev_code2[row] = missile_step_row(M.sp, M.vel, M.t0, M.alive, M.lidx, M.pos_cur, M.cap, M.m_slot, M.m_tgt,
                                                   M.m_radius, M.m_period, M.m_status, row, M.t2, M.dts, M.pend, M.mark2, M.grec,
                                                   M.pos_abs[0], M.pos_abs[1], /* second tick of a pair: */ M.mark, rb_first, P.R,
                                                   PHILOX, P.seed, P.tick, P.gid0, M.t);
        }
        stamp_end(P.stamps);
        return;
    }
    const int tid = threadIdx.x;
    const int bid = (int)blockIdx.x - H.mb;
    // (scalar loads by hand: the compiler cannot prove these words read-only and would fetch a wave-uniform word
    // with a vector load, whose full latency then sits in front of every row load of the wave)
    int blk = bid;
    if (H.order) {
        const int32_t *po = H.order + bid;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(blk) : "s"(po) : "memory");
    }
    const int64_t wave = (int64_t)blk * (ZRK_BLOCK / 64) + (tid >> 6);
    const int64_t cap = H.cap;
    // a batched ensemble: block `scen` of a table in device memory (written by the previous tick's compaction launch),
    // its own noise key, its own lists
    const int scen = H.bps ? (int)__umulhi((uint32_t)blk, H.bps_magic) : 0;
    const char *rbp = H.rb_table ? H.rb_table + (size_t)scen * sizeof(RadarBlock) : kernarg + kSweepRbOffset;
    const char *rbp2 = kernarg + sizeof(SweepParams);             // (PAIR: SweepParamsPair::rb2)
    uint64_t seed = H.seed;
    if (H.seeds) {
        const uint64_t *ps = H.seeds + scen;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seed) : "s"(ps) : "memory");
    }
    __shared__ PreTable s_pre;
    __shared__ PreTable s_pre2;                                   // (PAIR)
    uint4 pre_piece = pre_table_fetch(rbp);
    uint4 pre_piece2 = pre_piece;
    if (PAIR) pre_piece2 = pre_table_fetch(rbp2);
    ZRK_WAVE_PROBE(wave, 0, probe_entry);
    // where the wave runs: HW_REG_HW_ID (wave / simd / cu / sh / se) and HW_REG_XCC_ID
    // (bits 40 up: the workgroup's place in the dispatch)
    ZRK_WAVE_PROBE(wave, 6, (long long)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                ((long long)((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 32) | ((long long)bid << 40));
    const int64_t i = wave * 64 + (tid & 63);
    // every column load of the row is issued before anything waits for one: rows past the end read row 0
    const int64_t ic = (i < H.n) ? i : 0;
    const uint8_t al = H.alive[ic];
    // (removal marks: a variant of its own -- carried along unused they cost the plain loop's sweep 0.5 us)
    const uint32_t pk = MARKS ? (uint32_t)H.pend[ic] : 0u;
    const int32_t lix = LIDX ? H.lidx[ic] : 0;
    // (only the loads here: the arithmetic waits for them and comes after everything that does not)
    double t0 = 0.0, vx = 0.0, vy = 0.0, vz = 0.0, sx0, sy0, sz0;
    if (ADVANCE) {
        t0 = H.t0[ic];
        vx = H.vel[ic]; vy = H.vel[cap + ic]; vz = H.vel[2 * cap + ic];
        sx0 = H.sp[ic]; sy0 = H.sp[cap + ic]; sz0 = H.sp[2 * cap + ic];
    } else {
        sx0 = H.pos[ic]; sy0 = H.pos[cap + ic]; sz0 = H.pos[2 * cap + ic];
    }
    const bool removed = MARKS && (pk != 0u) & (pk != P.mark) & (pk != P.mark2);   // marked in an earlier launch of this call
    const bool live = (i < H.n) & (al != 0) & !removed;
    const int64_t li = (LIDX && i < H.n) ? (int64_t)lix : i;     // where this row sits in (its scenario's) AirEnv list
    // the wave's box record, if the caller keeps any (zrk_run_ticks does): twelve scalar words
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 b0 = {0u, 0u, 0u, 0u}, b1 = {0u, 0u, 0u, 0u}, bv = {0u, 0u, 0u, 0u};     // lo[3] hi[0] | hi[1..2] vmax[0..1] | vmax[2] state t_ref
    const bool cached = ADVANCE && H.boxes != nullptr;
    if (cached) {
        const WaveBox *pb = H.boxes + blk;
        asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(b0), "=&s"(b1), "=&s"(bv) : "s"(pb) : "memory");
    }
    const uint32_t bw[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    constexpr int kRowLoads = 1 + (MARKS ? 1 : 0) + (LIDX ? 1 : 0) + (ADVANCE ? 7 : 3);
    stage_pre_table<(PAIR ? 1 : 0) + kRowLoads>(s_pre, pre_piece);
    if (PAIR) stage_pre_table<kRowLoads>(s_pre2, pre_piece2);
    const bool shortcuts = !(P.flags & kNoInside);
    Cull c, c2;
    c.cand = 0u; c.inside = 0u; c.plane = 0u; c.pz_lo = 0u; c.pz_hi = 0u;
    c2 = c;
    bool have = false, have2 = !PAIR;                    // classified from the record, before the rows are here
    const uint32_t bstate = bv[1];
    __shared__ CullShared s_cull, s_cull2;
    __shared__ float s_wbox[ZRK_BLOCK / 64][12];
    const int wv = tid >> 6, lane = tid & 63;
    // (every wave of the workgroup reads the same record, so all of them take the same way through here)
    if (cached && bstate != 0u && !(P.flags & kNoBoxCache)) {
        const CullOrNot r1 = cull_from_record(s_pre, s_cull, P.R, shortcuts, P.t, bw[0], bw[1], bw[2], bw[3], bw[4], bw[5], bw[6], bw[7],
                                              bv[0], bstate, bv[2], bv[3]);
        if (r1.have) { c = r1.c; have = true; }
        if (PAIR) {
            const CullOrNot r2 = cull_from_record(s_pre2, s_cull2, P.R, shortcuts, P.t2, bw[0], bw[1], bw[2], bw[3], bw[4], bw[5], bw[6],
                                                  bw[7], bv[0], bstate, bv[2], bv[3]);
            if (r2.have) { c2 = r2.c; have2 = true; }
        }
    }
    double x = sx0, y = sy0, z = sz0, x2 = sx0, y2 = sy0, z2 = sz0;
    if (ADVANCE) {
        // Trajectory.get_pos: three separate roundings per axis
        const double d = P.t - t0;
        const double sx = vx * d, sy = vy * d, sz = vz * d;
        x = sx0 + sx; y = sy0 + sy; z = sz0 + sz;
        if (PAIR) {
            const double d2 = P.t2 - t0;
            const double ux = vx * d2, uy = vy * d2, uz = vz * d2;
            x2 = sx0 + ux; y2 = sy0 + uy; z2 = sz0 + uz;
        }
    }
#ifdef ZRK_PROBE_BUILD
    asm volatile("" ::"v"(x), "v"(y), "v"(z) : "memory");
#endif
    ZRK_WAVE_PROBE(wave, 1, wall_clock64());
    if (!have) {
        // no usable record: the box of the positions just computed
        const CullAndBox fresh = cull_from_rows(s_pre, P.R, shortcuts, live, x, y, z);
        c = fresh.c;
        const FreshBox fb = fresh.fb;
        if (cached) {
            // leave a record behind: every wave puts its box to LDS, the first wave joins them and lanes 0..11 store
            // one word each.  The speeds are taken over every row of the block, live or not (rows past the end read
            // row 0's: harmless, it only widens), once per record's life.
            const float inf = __builtin_inff(), kBig = 1e30f;
            float m0 = __builtin_bit_cast(float, bw[6]), m1 = __builtin_bit_cast(float, bw[7]), m2 = __builtin_bit_cast(float, bv[0]);
            if (bstate == 0u) {
                float n0 = fabsf((float)vx) * 1.000001f, n1 = fabsf((float)vy) * 1.000001f, n2 = fabsf((float)vz) * 1.000001f;
                float lo_unused = 0.f;
                wave_minmax(lo_unused, n0);
                lo_unused = 0.f; wave_minmax(lo_unused, n1);
                lo_unused = 0.f; wave_minmax(lo_unused, n2);
                const bool vfin = (n0 < kBig) & (n1 < kBig) & (n2 < kBig);      // NaN / inf speeds: the record never holds
                m0 = vfin ? n0 : inf; m1 = vfin ? n1 : inf; m2 = vfin ? n2 : inf;
            }
            if (lane == 0) {
                float *wb = s_wbox[wv];
                wb[0] = fb.blx; wb[1] = fb.bly; wb[2] = fb.blz; wb[3] = fb.bhx; wb[4] = fb.bhy; wb[5] = fb.bhz;
                wb[6] = m0; wb[7] = m1; wb[8] = m2;
                wb[9] = fb.any_wild ? 1.f : 0.f; wb[10] = fb.any_live ? 1.f : 0.f;
            }
            __syncthreads();
            if (wv == 0 && lane < 12) {
                const bool is_lo = lane < 3;
                float v = s_wbox[0][lane < 11 ? lane : 0];
#pragma unroll
                for (int k = 1; k < ZRK_BLOCK / 64; ++k) {
                    const float u = s_wbox[k][lane < 11 ? lane : 0];
                    v = is_lo ? fminf(v, u) : fmaxf(v, u);
                }
                float wild_any = 0.f, live_any = 0.f;
#pragma unroll
                for (int k = 0; k < ZRK_BLOCK / 64; ++k) { wild_any = fmaxf(wild_any, s_wbox[k][9]); live_any = fmaxf(live_any, s_wbox[k][10]); }
                const uint32_t st = live_any == 0.f ? kBoxEmpty : (wild_any != 0.f ? kBoxWild : kBoxOk);
                const uint64_t tb = __builtin_bit_cast(uint64_t, P.t);
                uint32_t word = __builtin_bit_cast(uint32_t, v);
                word = lane == 9 ? st : word;
                word = lane == 10 ? (uint32_t)tb : word;
                word = lane == 11 ? (uint32_t)(tb >> 32) : word;
                ((uint32_t *)(P.boxes + blk))[lane] = word;
            }
        }
    }
    if (PAIR && !have2) {                                // (the record, if any, did not reach to t2 either: classify from the rows)
        c2 = cull_from_rows(s_pre2, P.R, shortcuts, live, x2, y2, z2).c;
    }
    // wave-uniform: what this wave costs, for the next launch's order
    const int walked = __builtin_popcount(c.cand) + (PAIR ? __builtin_popcount(c2.cand) : 0);
    ZRK_WAVE_PROBE(wave, 7, (long long)(__builtin_popcount(c.cand) | (__builtin_popcount(c.inside) << 8) | (__builtin_popcount(c.plane) << 16) | ((int)have << 24)));
    uint32_t mask = 0u;
    // (PAIR: the second tick's positions wait in LDS while a wave walks the first tick's radars -- the walk calls out of line
    // for guard-band pairs, and every value that lives across a call costs the kernel registers for all its waves)
    __shared__ double s_hold[PAIR ? 3 : 1][PAIR ? ZRK_BLOCK : 1];
    if (c.cand) {
        if (PAIR) { s_hold[0][tid] = x2; s_hold[1][tid] = y2; s_hold[2][tid] = z2; }
        sweep_rows<PHILOX, !MARKS>(P.tick, P.gid0, rbp, seed, c, li, live, x, y, z, mask, wave);
        if (PAIR) { x2 = s_hold[0][tid]; y2 = s_hold[1][tid]; z2 = s_hold[2][tid]; }
    }
    // (the number is asked for here and used at the very end: its round trip hides behind the rows' stores)
    int next_slot = -1;
    if (P.order_next && tid == 0) {
        const int reg = bid % kOrderRegions, reg_n = (P.nb - reg + kOrderRegions - 1) / kOrderRegions;
        const int k = (int)atomicAdd(P.order_ctr + (reg * 2 + (walked ? 0 : 1)) * kOrderCtrStride, 1u);
        next_slot = (walked ? k : reg_n - 1 - k) * kOrderRegions + reg;
    }
    if (P.order_next && bid == 0 && tid < 2 * kOrderRegions) P.order_ctr_next[tid * kOrderCtrStride] = 0u;
    ZRK_WAVE_PROBE(wave, 2, wall_clock64());
    ZRK_WAVE_PROBE(wave, 4, (long long)__popcll(__ballot(mask != 0)));
    if (live && (PHILOX || ADVANCE)) {
        P.pos[i] = x; P.pos[cap + i] = y; P.pos[2 * cap + i] = z;
    }
    // sparse mode: the buffer is known to be all zero (the previous tick's compaction cleared it), so only
    // detections are written -- list-indexed stores are scattered when the table is spatially sorted
    if (i < P.n && (mask || !(P.flags & kSparseVis))) P.vis[(int64_t)scen * P.rows_ps + li] = mask;
    if (P.vis_clear && i < P.n) P.vis_clear[i] = 0u;         // (word i, not the row's list index: the same words in all, written side by side)
    if (PAIR) {
        // the second tick: same rows, the next radar records, the next noise key.  (A row that tick t's missile phase removes
        // is swept here all the same -- its thread cannot know; see SweepParams::t2 for who puts that right.)
        uint32_t mask2 = 0u;
        if (c2.cand) sweep_rows<PHILOX, !MARKS>(P.tick + 1, P.gid0, rbp2, seed, c2, li, live, x2, y2, z2, mask2, wave);
        ZRK_WAVE_PROBE(wave, 4, wall_clock64());              // (pair: slot 4 is the end of the second tick's radar loop)
        if (live) { P.pos_prev[i] = x2; P.pos_prev[cap + i] = y2; P.pos_prev[2 * cap + i] = z2; }
        if (i < P.n && (mask2 || !(P.flags & kSparseVis2))) P.vis2[li] = mask2;
    }
    if (MARKS && removed && al != 0 && i < P.n) {            // carry the removal out: once, by the row's own thread
        P.alive_w[i] = 0;
        const double *src = P.pos_abs[pk & 1u];
        double *dst = P.pos_abs[(pk & 1u) ^ 1u];
        dst[i] = src[i]; dst[cap + i] = src[cap + i]; dst[2 * cap + i] = src[2 * cap + i];
    }
    if (P.order_next && tid == 0 && (unsigned)next_slot < (unsigned)P.nb) P.order_next[next_slot] = blk;
    ZRK_WAVE_PROBE(wave, 3, wall_clock64());
    stamp_end(P.stamps);
}


// Compaction, phase 1: per-block detection counts per radar (row R: seen by any radar) from vis_mask,
// in list order.  Lane r of each wave collects the wave's count for radar r, then one LDS add per lane.
__global__ __launch_bounds__(kCompBlock) void k_count_blocks(const uint32_t *__restrict__ vis, int64_t n, int R, int nb,
                                                             int32_t *__restrict__ counts)
{
    __shared__ int s_cnt[ZRK_MAX_RADARS + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t i = (int64_t)blockIdx.x * kCompBlock + tid;
    if (tid <= ZRK_MAX_RADARS) s_cnt[tid] = 0;
    __syncthreads();
    const uint32_t mask = (i < n) ? vis[i] : 0u;
    int cnt_lane = 0;
    for (int r = 0; r < R; ++r) {
        const unsigned long long b = __ballot((mask >> r) & 1u);
        cnt_lane = (lane == r) ? (int)__popcll(b) : cnt_lane;
    }
    {
        const unsigned long long b = __ballot(mask != 0u);
        cnt_lane = (lane == R) ? (int)__popcll(b) : cnt_lane;
    }
    if (lane <= R && cnt_lane) atomicAdd(&s_cnt[lane], cnt_lane);
    __syncthreads();
    if (tid <= R) counts[(int64_t)tid * nb + blockIdx.x] = s_cnt[tid];
}

// ---------------------------------------------------------------------------------------------
// Compaction, phase 2: exclusive scan of the per-workgroup counts, one workgroup per radar.
// ---------------------------------------------------------------------------------------------
constexpr int kScanThreads = 1024;
constexpr int kScanItems = 4;

__device__ void missile_finish_entry(int *s_wave, const MissileArgs &M);
__device__ void missile_kills(const MissileArgs &M, int part);

__global__ __launch_bounds__(kScanThreads) void k_scan_counts(const int32_t *__restrict__ counts,
                                                              int32_t *__restrict__ offs,
                                                              int32_t *__restrict__ totals, int nb, int rows,
                                                              const MissileArgs M)
{
    __shared__ int s_wave[kScanThreads / 64];
    __shared__ int s_carry;
    if ((int)blockIdx.x >= rows) {              // the one extra workgroup: missile events + tombstones
        missile_finish_entry(s_wave, M);
        return;
    }
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *c = counts + (int64_t)r * nb;
    int32_t *o = offs + (int64_t)r * nb;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += kScanThreads * kScanItems) {
        int v[kScanItems], sum = 0;
        const int j0 = base + tid * kScanItems;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            v[k] = (j0 + k < nb) ? c[j0 + k] : 0;
            sum += v[k];
        }
        int incl = sum;                                    // inclusive scan of `sum` across the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int wave_off = 0, total = 0;
        for (int w = 0; w < kScanThreads / 64; ++w) {
            int t = s_wave[w];
            if (w < wave) wave_off += t;
            total += t;
        }
        int run = s_carry + wave_off + incl - sum;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            if (j0 + k < nb) o[j0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 0) s_carry += total;
        __syncthreads();
    }
    if (tid == 0) totals[r] = s_carry;
}

// The union list in its wire format (ZRK_F_UNION_BITS / zrk_compact_bits): word 0 = number of slots seen by any
// radar, word 1 = n, then one bit per slot (ceil(n / 64) words), then the masks of the seen slots in ascending
// order, 16 bits each when R <= 16, else 32.  A quarter of the bytes of the (index, mask) pairs at the usual 1-in-6 density --
// what crosses xGMI every tick.
struct UnionBits {
    int64_t words;               // bitmap words (0: the pairs format)
    int64_t mask_cap;            // masks that fit behind the bitmap
    int32_t mask_bytes, _pad;
};

__device__ __forceinline__ void union_bits_mask(int64_t *packed, const UnionBits &U, int64_t rank, uint32_t mask)
{
    if (rank >= U.mask_cap) return;
    if (U.mask_bytes == 2) ((uint16_t *)(packed + 2 + U.words))[rank] = (uint16_t)mask;
    else ((uint32_t *)(packed + 2 + U.words))[rank] = mask;
}

// ---------------------------------------------------------------------------------------------
// Compaction, phase 3: stable scatter.  Lane order == slot order inside a wave, so
// popcount(ballot & lanes-below) is the rank; waves and workgroups are ordered by the scans.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kCompBlock) void k_scatter(const uint32_t *__restrict__ vis, int64_t n, int R,
                                                        int nb, const int32_t *__restrict__ offs,
                                                        const int32_t *__restrict__ totals, int32_t base_index,
                                                        int32_t *__restrict__ det_idx, int64_t det_stride,
                                                        int32_t *__restrict__ det_cnt, int64_t *__restrict__ packed,
                                                        int64_t packed_capacity, int64_t gid0,
                                                        uint32_t *__restrict__ zero_next, const UnionBits U)
{
    constexpr int kWaves = kCompBlock / 64;
    __shared__ int s_wcnt[kWaves];
    __shared__ int s_base[ZRK_MAX_RADARS + 1];
    __shared__ unsigned short s_idx[kCompBlock];      // detected slots of this block, in list order ...
    __shared__ uint32_t s_msk[kCompBlock];            // ... and their masks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t blk0 = (int64_t)blockIdx.x * kCompBlock;
    const int64_t i = blk0 + tid;
    const uint32_t m = (i < n) ? vis[i] : 0u;
    if (zero_next && i < n) zero_next[i] = 0u;        // next tick's (other) mask buffer, cleared in passing
    if (tid <= R) {
        // s_base[r] = rank of this workgroup's first detection of radar r in that radar's list
        // (row R: in the packed union list)
        if (det_cnt && blockIdx.x == 0) det_cnt[tid] = totals[tid];
        s_base[tid] = offs[(int64_t)tid * nb + blockIdx.x];
    }
    // Only ~15 % of the slots carry a detection: squeeze those into LDS first (stable), then each wave
    // walks the short list for "its" radars instead of every wave walking every radar over all slots.
    const unsigned long long bu = __ballot(m != 0u);
    if (lane == 0) {
        s_wcnt[wave] = (int)__popcll(bu);
        const int64_t word = (blk0 + wave * 64) >> 6;
        if (packed && U.words && word < U.words) packed[2 + word] = (int64_t)bu;
    }
    __syncthreads();
    int woff = 0, found = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        const int c = s_wcnt[w];
        if (w < wave) woff += c;
        found += c;
    }
    if (m != 0u) {
        const int k = woff + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bu, 0u));
        s_idx[k] = (unsigned short)tid;
        s_msk[k] = m;
    }
    __syncthreads();
    if (packed) {
        // union list for the multi-GPU exchange: packed[0] = count, then (global index << 32 | mask)
        if (blockIdx.x == 0 && tid == 0) {
            packed[0] = totals[R];
            if (U.words) packed[1] = n;
        }
        const int64_t ubase = s_base[R];
        for (int k = tid; k < found; k += kCompBlock) {
            const int64_t dst = ubase + k;
            if (U.words) union_bits_mask(packed, U, dst, s_msk[k]);
            else if (dst + 1 < packed_capacity) packed[dst + 1] = ((gid0 + blk0 + s_idx[k]) << 32) | (int64_t)s_msk[k];
        }
    }
    if (det_idx) {
        for (int r = wave; r < R; r += kWaves) {           // this wave's radars
            int run = s_base[r];
            for (int c = 0; c < found; c += 64) {
                const int k = c + lane;
                const bool bit = (k < found) && ((s_msk[k] >> r) & 1u);
                const unsigned long long b = __ballot(bit);
                if (bit) {
                    const int64_t dst = (int64_t)run + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                    if (dst < det_stride) det_idx[(int64_t)r * det_stride + dst] = base_index + (int32_t)(blk0 + s_idx[k]);
                }
                run += (int)__popcll(b);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Compaction in ONE launch (the default up to kFusedMaxBlocks workgroups): every workgroup takes a ticket
// (its position in list order), squeezes its detections into LDS, publishes its R+1 counts as
// (epoch << 32 | count) words and adds up the words of all lower tickets -- which belong to workgroups that
// are already running, so the wait cannot deadlock whatever the residency -- then scatters.  No chained
// dependency: a workgroup's counts depend on nobody.  The last ticket writes the totals; the last
// workgroup to finish rearms the ticket counter.  `epoch` differs per launch, so the words need no reset.
// ---------------------------------------------------------------------------------------------
#ifndef ZRK_PROBE
#define ZRK_PROBE(slot)                          // tools/compact_phases.hip records wall-clock stamps here
#endif
constexpr int kFusedMaxItems = 8;                // list slots per thread
constexpr int kFusedMaxBlocks = 2048;             // (16.8e6 rows at eight slots per thread)
constexpr int kAggStride = 40;                   // 64-bit words per workgroup record (>= ZRK_MAX_RADARS + 1)
constexpr int kGroupOffset = kFusedMaxBlocks * 2 * kAggStride;   // the group records' place behind the widest records (words)
constexpr int kMinGroup = 4;
constexpr int kFusedCtlInts = 64;                // ticket, done, error, padding
constexpr int kSpinLimit = 1 << 22;

// One workgroup: order[] = row blocks with cost > 0 (ascending), then the rest (ascending); cost[] cleared.
// (Sorting by cost instead of two classes measured the same.)
__device__ __forceinline__ unsigned long long agg_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of counter c over the records [a0, a1) of `ra` and then [b0, b1) of `rb` (either range may be empty), as one run of
// loads: record q of the run is dealt to the threads p_step apart, kBatch loads in flight per thread, each waited for until
// it carries this launch's epoch.
template <int STRIDE>
__device__ __forceinline__ int sum_epoch_records(const unsigned long long *ra, int a0, int a1, const unsigned long long *rb, int b0, int b1,
                                                 int c, int p_first, int p_step, uint32_t epoch, bool &timed_out)
{
    constexpr int kBatch = 8;
    const int na = a1 - a0, total = na + (b1 - b0);
    int acc = 0;
    for (int q0 = p_first; q0 < total; q0 += p_step * kBatch) {
        unsigned long long v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int q = q0 + u * p_step;
            const unsigned long long *w = q < na ? ra + (int64_t)(a0 + q) * STRIDE + c : rb + (int64_t)(b0 + q - na) * STRIDE + c;
            v[u] = (q < total) ? agg_load(w) : ((unsigned long long)epoch << 32);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int q = q0 + u * p_step;
            const unsigned long long *w = q < na ? ra + (int64_t)(a0 + q) * STRIDE + c : rb + (int64_t)(b0 + q - na) * STRIDE + c;
            int spins = 0;
            while ((uint32_t)(v[u] >> 32) != epoch) {
                if (++spins > kSpinLimit) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(1);
                v[u] = agg_load(w);
            }
            acc += (int)(uint32_t)v[u];
        }
    }
    return acc;
}

// The walks over a workgroup's squeezed list (detected slots in list order: idx[], msk[]) that count and scatter per radar.
// They are most of what a compaction launch executes -- 9e4 (64-entry step, radar) pairs per launch at C3, each walked twice --
// and beside a sweep whose heavy waves are arithmetic-bound their instruction count is what the compaction costs (DESIGN.md
// section 5.2).  So: the list is zero-padded to a multiple of 256 entries (no read of it hangs on a predicate), a wave walks it
// ONCE for up to two radars (one LDS read per entry and per pass, the entries' list slots added up once), the lists' positions
// are wave-uniform scalars, the stores take a 32-bit offset against the list's base, and the check against the list's capacity
// is made once per four steps (256 entries), not per lane (a list about to run out of room takes the careful store).  27 -> 12 instructions per
// (step, radar) in the scatter, 12 -> 5 in the count.
template <int JOBS>
__device__ __forceinline__ void walk_count(const uint32_t *msk, int len_pad, const uint32_t (&sel)[JOBS], int (&cnt)[JOBS])
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < JOBS; ++q) cnt[q] = 0;
    for (int c = 0; c < len_pad; c += 256) {
        uint32_t m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) m[u] = msk[c + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < JOBS; ++q) cnt[q] += (int)__popcll(__ballot((m[u] & sel[q]) != 0u));
    }
}

template <int JOBS>
__device__ __forceinline__ void walk_scatter(const uint32_t *msk, const unsigned short *idx, int len_pad, const uint32_t (&sel)[JOBS],
                                             int (&run)[JOBS], int32_t *const (&out)[JOBS], int limit, int32_t slot0)
{
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < len_pad; c += 256) {
        uint32_t m[4];
        int32_t slot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            m[u] = msk[c + u * 64 + lane];
            slot[u] = slot0 + (int32_t)idx[c + u * 64 + lane];
        }
#pragma unroll
        for (int q = 0; q < JOBS; ++q) {
            if (run[q] + 256 <= limit) {                 // (wave-uniform: the list holds whatever these four steps add)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool bit = (m[u] & sel[q]) != 0u;
                    const unsigned long long bb = __ballot(bit);
                    const uint32_t at = (uint32_t)run[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb, 0u));
                    if (bit) out[q][at] = slot[u];
                    run[q] += (int)__popcll(bb);
                }
            } else {                                     // a list about to run out of room: every store looks
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool bit = (m[u] & sel[q]) != 0u;
                    const unsigned long long bb = __ballot(bit);
                    const uint32_t at = (uint32_t)run[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(bb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bb, 0u));
                    if (bit && (int)at < limit) out[q][at] = slot[u];
                    run[q] += (int)__popcll(bb);
                }
            }
        }
    }
}

struct CompactArgs {
    const uint32_t *vis;           // NULL: nothing to compact
    uint32_t *zero_next;
    int64_t n;
    int32_t R, nb, items, lanes;   // nb workgroups of `items` slots per thread; lanes: power of two >= R + 1
    uint32_t epoch;
    int32_t base_index;
    int32_t *ctl;
    unsigned long long *agg;
    int32_t *det_idx;
    int64_t det_stride;
    int32_t *det_cnt;
    int64_t *packed;
    int64_t packed_capacity, gid0;
    UnionBits bits;
    // batched ensemble: the lists restart every seg_blocks workgroups (= seg_slots list slots, one scenario); radar r
    // of scenario s writes to det_idx[(s * R + r) * det_stride ...], counts to det_cnt[s * (R + 1) + r]; 0: one list
    int32_t seg_blocks, zero_own;  // zero_own: zero_next is `vis` itself, all zero but for the detections
    int64_t seg_slots;
    // two-level sums (0: every workgroup adds up all lower records): the last ticket of every `group` consecutive ones also
    // publishes the group's totals (behind the records, at agg + kGroupOffset), and a workgroup adds up the
    // lower records of its own group and the totals of the groups before it: group + nb / group words instead of nb
    int32_t group;
    // radars of interest (zrk_exchange_io::interest; 0: all): the union list is that of these radars alone -- a slot counts as
    // seen when one of THEM saw it, its mask carries their bits.  Lists only for the wire (det_idx == NULL)
    uint32_t select;
};

// What a compaction workgroup needs only at its END -- where the lists go -- is read there, from the kernel-argument segment,
// through an address the compiler cannot connect with the arguments it has seen (launder_kernarg): left to itself it fetches
// both ticks' CompactArgs in sixteen-word bursts half-way through the kernel and keeps them in scalar registers across the
// squeeze, the walks and the wait for the lower tickets -- 50-54 of them spilt into vector lanes in every k_compact_pair variant,
// 30 in k_compact_side / k_compact_fused (profiles/r04: every compaction kernel at the 106-SGPR cap).  THE ADDRESS IS FORMED IN
// THE __global__ BODY ONLY (see g_device_fault: outside an entry point the intrinsic is the constant 0).
struct CompactLate {
    int32_t *det_idx;
    int64_t det_stride;
    int32_t *det_cnt;
    int64_t *packed;
    int64_t packed_capacity, gid0;
    UnionBits bits;
    int32_t base_index;
    int64_t seg_slots;
};

__device__ __forceinline__ const char *launder_kernarg(const char *p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// ... and what it needs between the squeeze and the scatter: where the records are, this launch's tag, how the sums are laid out
struct CompactMid {
    unsigned long long *agg;
    int32_t *ctl;
    uint32_t epoch;
    int32_t lanes, group;
};

__device__ __forceinline__ CompactMid compact_mid(const char *kc)
{
    typedef const uint64_t __attribute__((address_space(4))) *Q;
    typedef const uint32_t __attribute__((address_space(4))) *W;
    const char *k = launder_kernarg(kc);
    CompactMid M;
    M.agg = (unsigned long long *)*(Q)(uint64_t)(k + offsetof(CompactArgs, agg));
    M.ctl = (int32_t *)*(Q)(uint64_t)(k + offsetof(CompactArgs, ctl));
    M.epoch = *(W)(uint64_t)(k + offsetof(CompactArgs, epoch));
    M.lanes = (int32_t)*(W)(uint64_t)(k + offsetof(CompactArgs, lanes));
    M.group = (int32_t)*(W)(uint64_t)(k + offsetof(CompactArgs, group));
    return M;
}

// `kc`: where this workgroup's CompactArgs stand in the kernel-argument segment
__device__ __forceinline__ CompactLate compact_late(const char *kc)
{
    typedef const uint64_t __attribute__((address_space(4))) *Q;
    typedef const uint32_t __attribute__((address_space(4))) *W;
    const char *k = launder_kernarg(kc);
    CompactLate L;
    L.det_idx = (int32_t *)*(Q)(uint64_t)(k + offsetof(CompactArgs, det_idx));
    L.det_stride = (int64_t)*(Q)(uint64_t)(k + offsetof(CompactArgs, det_stride));
    L.det_cnt = (int32_t *)*(Q)(uint64_t)(k + offsetof(CompactArgs, det_cnt));
    L.packed = (int64_t *)*(Q)(uint64_t)(k + offsetof(CompactArgs, packed));
    L.packed_capacity = (int64_t)*(Q)(uint64_t)(k + offsetof(CompactArgs, packed_capacity));
    L.gid0 = (int64_t)*(Q)(uint64_t)(k + offsetof(CompactArgs, gid0));
    L.bits.words = (int64_t)*(Q)(uint64_t)(k + offsetof(CompactArgs, bits) + offsetof(UnionBits, words));
    L.bits.mask_cap = (int64_t)*(Q)(uint64_t)(k + offsetof(CompactArgs, bits) + offsetof(UnionBits, mask_cap));
    L.bits.mask_bytes = (int32_t)*(W)(uint64_t)(k + offsetof(CompactArgs, bits) + offsetof(UnionBits, mask_bytes));
    L.bits._pad = 0;
    L.base_index = (int32_t)*(W)(uint64_t)(k + offsetof(CompactArgs, base_index));
    L.seg_slots = (int64_t)*(Q)(uint64_t)(k + offsetof(CompactArgs, seg_slots));
    return L;
}

template <int THREADS>
struct CompactShared {
    int wcnt[kFusedMaxItems * (THREADS / 64)];
    int cnt[ZRK_MAX_RADARS + 1];
    int pre[ZRK_MAX_RADARS + 1];
    int grp[ZRK_MAX_RADARS + 1];
    int ticket, found;
    unsigned short idx[kFusedMaxItems * THREADS];  // detected slots of this workgroup, in list order ...
    uint32_t msk[kFusedMaxItems * THREADS];        // ... and their masks
};

// One workgroup of the single-launch compaction (THREADS threads, C.items slots each).  Runs as the stand-alone
// kernel below (1024 threads) and as the leading workgroups of the next tick's sweep (ZRK_BLOCK threads).
template <int THREADS>
__device__ __forceinline__ void compact_block(CompactShared<THREADS> &S, const CompactArgs &C, int by_ticket, const char *kc)
{
    // (the wave's number as a scalar: what follows from it -- the radars it walks, their lists' bases -- then lives in SGPRs)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) S.ticket = by_ticket ? atomicAdd(&C.ctl[0], 1) : (int)blockIdx.x;
    if (tid <= ZRK_MAX_RADARS) S.pre[tid] = S.grp[tid] = 0;
    __syncthreads();
    const int b = S.ticket;
    if (b < 0 || b >= C.nb) {                        // a workspace that was not ours: refuse rather than scribble
        if (tid == 0) atomicExch(&C.ctl[2], 1);
        return;
    }
    const int64_t blk0 = (int64_t)b * C.items * THREADS;
    const int seg = C.seg_blocks ? b / C.seg_blocks : 0;          // scenario of this workgroup
    const int first = seg * C.seg_blocks;                         // its first workgroup: nobody before it counts
    ZRK_PROBE(0);
    // (32-bit slot numbers against the block's own base address; a slot past the table's end reads the block's last one and
    // counts as empty: no load of the burst hangs on a branch -- see compact_block_pair)
    const int rem = (int)((C.n - blk0) < (int64_t)C.items * THREADS ? (C.n - blk0) : (int64_t)C.items * THREADS);
    const uint32_t *vb = C.vis + blk0;
    uint32_t mk[kFusedMaxItems];
#pragma unroll
    for (int it = 0; it < kFusedMaxItems; ++it) {  // every load in flight before anything looks at one
        const int slot = it * THREADS + tid;
        mk[it] = (it < C.items) ? vb[slot < rem ? slot : rem - 1] : 0u;
    }
    uint32_t *zb = C.zero_next ? C.zero_next + blk0 : nullptr;
    const bool own = C.zero_own != 0;
    int64_t *words = C.packed + 2 + (blk0 >> 6);
    const int64_t wlimit = (C.packed && C.bits.words) ? C.bits.words - (blk0 >> 6) : 0;
#pragma unroll
    for (int it = 0; it < kFusedMaxItems; ++it) {
        if (it < C.items) {
            const int slot = it * THREADS + tid;
            mk[it] = slot < rem ? mk[it] : 0u;
            // next tick's (other) mask buffer, cleared in passing; the overlapped loop clears the buffer it has just read,
            // where only the detections are not zero already
            if (zb && (own ? mk[it] != 0u : slot < rem)) zb[slot] = 0u;
            if (C.select) mk[it] &= C.select;        // (radars of interest: behind the clearing, which goes by what any radar saw)
            const unsigned long long bu = __ballot(mk[it] != 0u);
            if (lane == 0) {
                S.wcnt[it * (THREADS / 64) + wave] = (int)__popcll(bu);
                const int w = it * (THREADS / 64) + wave;                                    // 64 consecutive slots: one word of the bitmap
                if (w < wlimit) words[w] = (int64_t)bu;
            }
        }
    }
    ZRK_PROBE(1);
    __syncthreads();
    if (wave == 0) {                               // exclusive scan of the (item, wave) counts: list order
        const int m2 = C.items * (THREADS / 64);
        const int a0 = (2 * lane < m2) ? S.wcnt[2 * lane] : 0, a1 = (2 * lane + 1 < m2) ? S.wcnt[2 * lane + 1] : 0;
        int incl = a0 + a1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const int excl = incl - a0 - a1;
        if (2 * lane < m2) S.wcnt[2 * lane] = excl;
        if (2 * lane + 1 < m2) S.wcnt[2 * lane + 1] = excl + a0;
        if (lane == 63) S.found = incl;
    }
    __syncthreads();
    const int found = S.found;
#pragma unroll
    for (int it = 0; it < kFusedMaxItems; ++it) {
        if (it < C.items) {
            const unsigned long long bu = __ballot(mk[it] != 0u);
            if (mk[it] != 0u) {
                const int k = S.wcnt[it * (THREADS / 64) + wave] +
                              (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bu, 0u));
                S.idx[k] = (unsigned short)(it * THREADS + tid);
                S.msk[k] = mk[it];
            }
        }
    }
    // (the list is zero-padded to a multiple of 256 entries: the walks read it without a predicate)
    const int len_pad = (found + 255) & ~255;
    if (tid < 256 && found + tid < len_pad) S.msk[found + tid] = 0u;
    __syncthreads();
    ZRK_PROBE(2);
    // per-radar counts over the short list: wave w takes radars w and w + WAVES in one walk
    {
        constexpr int WAVES = THREADS / 64;
        for (int rb = wave; rb < C.R; rb += 2 * WAVES) {
            const uint32_t sel[2] = {1u << rb, (rb + WAVES < C.R) ? (1u << (rb + WAVES)) : 0u};
            int cnt[2];
            if (sel[1]) walk_count<2>(S.msk, len_pad, sel, cnt);
            else { const uint32_t sel1[1] = {sel[0]}; int c1[1]; walk_count<1>(S.msk, len_pad, sel1, c1); cnt[0] = c1[0]; cnt[1] = 0; }
            if (lane == 0) { S.cnt[rb] = cnt[0]; if (sel[1]) S.cnt[rb + WAVES] = cnt[1]; }
        }
    }
    if (tid == 0) S.cnt[C.R] = found;
    __syncthreads();
    ZRK_PROBE(3);
    const CompactMid Q = compact_mid(kc);
    if (tid <= C.R)
        __hip_atomic_store(&Q.agg[(int64_t)b * kAggStride + tid], ((unsigned long long)Q.epoch << 32) | (uint32_t)S.cnt[tid],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        // lower tickets: record p, counter c is word p * Q.lanes + c of a (p, c) grid dealt out to the
        // threads THREADS apart (Q.lanes: a power of two >= C.R+1)
        const int c = tid & (Q.lanes - 1);
        const int p_first = tid / Q.lanes, p_step = THREADS / Q.lanes;
        const int G = Q.group, g = G ? b / G : 0, gs = g * G;
        const unsigned long long *gagg = Q.agg + kGroupOffset;
        bool timed_out = false;
        if (G && b - gs == G - 1) {                   // the group's last ticket: its totals first, they wait for nobody before the group
            if (c <= C.R) {
                const int in = sum_epoch_records<kAggStride>(Q.agg, gs, b, gagg, 0, 0, c, p_first, p_step, Q.epoch, timed_out);
                if (in) { atomicAdd(&S.grp[c], in); atomicAdd(&S.pre[c], in); }
            }
            __syncthreads();
            if (tid <= C.R)
                __hip_atomic_store(const_cast<unsigned long long *>(gagg) + (int64_t)g * kAggStride + tid,
                                   ((unsigned long long)Q.epoch << 32) | (uint32_t)(S.grp[tid] + S.cnt[tid]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (c <= C.R) {
                const int acc = sum_epoch_records<kAggStride>(Q.agg, 0, 0, gagg, 0, g, c, p_first, p_step, Q.epoch, timed_out);
                if (acc) atomicAdd(&S.pre[c], acc);
            }
        } else if (c <= C.R) {
            const int acc = G ? sum_epoch_records<kAggStride>(Q.agg, gs, b, gagg, 0, g, c, p_first, p_step, Q.epoch, timed_out)
                              : sum_epoch_records<kAggStride>(Q.agg, first, b, gagg, 0, 0, c, p_first, p_step, Q.epoch, timed_out);
            if (acc) atomicAdd(&S.pre[c], acc);
        }
        if (timed_out) atomicExch(&Q.ctl[2], 2);
    }
    __syncthreads();
    ZRK_PROBE(4);
    const CompactLate L = compact_late(kc);          // (where the lists go: asked for here, not carried through the kernel)
    if (L.packed) {
        const int64_t ubase = S.pre[C.R];
        for (int k = tid; k < found; k += THREADS) {
            const int64_t dst = ubase + k;
            if (L.bits.words) union_bits_mask(L.packed, L.bits, dst, S.msk[k]);
            else if (dst + 1 < L.packed_capacity) L.packed[dst + 1] = ((L.gid0 + blk0 + S.idx[k]) << 32) | (int64_t)S.msk[k];
        }
    }
    if (L.det_idx) {
        constexpr int WAVES = THREADS / 64;
        const int limit = (int)(L.det_stride < 0x7FFFFFFF ? L.det_stride : 0x7FFFFFFF);
        const int32_t slot0 = L.base_index + (int32_t)(blk0 - (int64_t)seg * L.seg_slots);
        int32_t *seg_out = L.det_idx + (int64_t)seg * C.R * L.det_stride;
        for (int rb = wave; rb < C.R; rb += 2 * WAVES) {
            const bool two = rb + WAVES < C.R;
            const uint32_t sel[2] = {1u << rb, two ? (1u << (rb + WAVES)) : 0u};
            int run[2] = {__builtin_amdgcn_readfirstlane(S.pre[rb]), two ? __builtin_amdgcn_readfirstlane(S.pre[rb + WAVES]) : 0};
            int32_t *const out[2] = {seg_out + (int64_t)rb * L.det_stride, seg_out + (int64_t)(two ? rb + WAVES : rb) * L.det_stride};
            if (two) walk_scatter<2>(S.msk, S.idx, len_pad, sel, run, out, limit, slot0);
            else {
                const uint32_t sel1[1] = {sel[0]}; int run1[1] = {run[0]}; int32_t *const out1[1] = {out[0]};
                walk_scatter<1>(S.msk, S.idx, len_pad, sel1, run1, out1, limit, slot0);
            }
        }
    }
    typedef const uint32_t __attribute__((address_space(4))) *LateWords;
    const int late_seg_blocks = (int)*(LateWords)(uint64_t)(launder_kernarg(kc) + offsetof(CompactArgs, seg_blocks));
    const int late_nb = (int)*(LateWords)(uint64_t)(launder_kernarg(kc) + offsetof(CompactArgs, nb));
    const bool last_of_seg = late_seg_blocks ? (b == seg * late_seg_blocks + late_seg_blocks - 1) : (b == late_nb - 1);
    if (last_of_seg && tid <= C.R) {                   // the end of the (scenario's) list: totals
        const int tot = S.pre[tid] + S.cnt[tid];
        if (L.det_cnt) L.det_cnt[(int64_t)seg * (C.R + 1) + tid] = tot;
        if (L.packed && tid == C.R) {
            L.packed[0] = tot;
            if (L.bits.words) L.packed[1] = C.n;
        }
    }
    ZRK_PROBE(5);
    // (by_ticket: the kernels' second argument, right behind C)
    if (*(LateWords)(uint64_t)(launder_kernarg(kc) + sizeof(CompactArgs)) != 0u && tid == 0) {
        int32_t *ctl = compact_mid(kc).ctl;
        if (atomicAdd(&ctl[1], 1) == late_nb - 1) {    // everybody holds a ticket and is done with it
            atomicExch(&ctl[0], 0);
            atomicExch(&ctl[1], 0);
        }
    }
}

// Batched ensemble: the radars of every scenario live on the device.  One thread per (scenario, radar slot) moves
// the scan on (SectorRadar.move_to_next_sector_circular, modules/Radar.py:96-117, :205) and derives the records the
// NEXT tick's sweep reads -- as extra workgroups of this tick's compaction, so that the host has no per-scenario
// work in the loop at all.
struct EnsembleArgs {
    zrk_radar *state;            // [S][R] current angles, in/out
    const zrk_scan *scan;        // [S][R]
    const double *d2max;         // [S][R] d2_threshold(max_distance)
    RadarBlock *table_out;       // [S]
    int32_t S, R;
    uint32_t flags;              // ZRK_F_PHILOX / ZRK_F_EXACT_ONLY
    int32_t advance;             // 0: records for the angles as they are
};

__device__ void ensemble_derive(const EnsembleArgs &E, int part)
{
    // one thread per radar in use, densely (records of slots beyond R are never read: the classification masks them)
    const int64_t g = (int64_t)part * blockDim.x + threadIdx.x;
    if (g >= (int64_t)E.S * E.R) return;
    const int sc = (int)(g / E.R), r = (int)(g % E.R);
    RadarBlock *rb = E.table_out + sc;
    zrk_radar rd = E.state[g];
    if (E.advance) {
        scan_advance_one(rd, E.scan[g]);
        E.state[g] = rd;
    }
    RadarHot hot;
    RadarCold cold;
    RadarPre pre;
    derive_radar(rd, E.d2max[g], (E.flags & ZRK_F_EXACT_ONLY) != 0, hot, cold);
    derive_pre(rd, hot, (E.flags & ZRK_F_PHILOX) != 0, r, pre);
    rb->cold[r] = cold;
    const uint32_t *hwords = (const uint32_t *)&hot;
    const uint32_t *pwords = (const uint32_t *)&pre;
    for (int k = 0; k < 20; ++k) { rb->hotw[r][k] = hwords[k]; rb->prew[r][k] = pwords[k]; }
}

__global__ __launch_bounds__(kCompBlock) void k_ensemble_derive(const EnsembleArgs E) { ensemble_derive(E, (int)blockIdx.x); }

// One scenario: the host derived the records; this puts them where the sweep's scalar and vector loads find them in
// ordinary (cacheable) device memory instead of the kernel-argument segment.
struct PutArgs {
    uint32_t *dst;               // NULL: nothing to put
    RadarBlock rb;
};

__device__ void put_radar_block(const PutArgs &U)
{
    const uint32_t *src = (const uint32_t *)&U.rb;
    for (int k = threadIdx.x; k < (int)(sizeof(RadarBlock) / 4); k += blockDim.x) U.dst[k] = src[k];
}

__global__ __launch_bounds__(256) void k_put_radar_block(const PutArgs U) { put_radar_block(U); }

__global__ __launch_bounds__(kCompBlock) void k_compact_fused(const CompactArgs C, int by_ticket, const MissileArgs M,
                                                              const EnsembleArgs E, const PutArgs U)
{
    __shared__ int s_wave[kCompBlock / 64];
    __shared__ CompactShared<kCompBlock> S;
    if ((int)blockIdx.x >= C.nb) {                 // extra workgroups: missile events + tombstones, radars
        int extra = (int)blockIdx.x - C.nb;
        if (M.m > 0 && extra-- == 0) { missile_finish_entry(s_wave, M); return; }
        if (E.S > 0) { ensemble_derive(E, extra); return; }
        if (U.dst) put_radar_block(U);              // one scenario: the next tick's records, derived by the host
        return;
    }
    compact_block<kCompBlock>(S, C, by_ticket, (const char *)__builtin_amdgcn_kernarg_segment_ptr());      // (C: the first argument)
}

// The side stream's compactions tell the HOST which of them are over through a word in pinned memory: the first thread of
// a compaction writes the number of the item BEFORE it -- launches of one stream run one after the other, so when this one
// runs that one is over, its stores written back.  The loop's back-pressure (a mask buffer is free again when the
// compaction that read it is over) then costs the side stream no packet of its own -- an event record behind every launch
// held the next launch back by 2.8 us (tools/backtoback_probe.hip) -- and the waiting host thread no call into the runtime.
// (Workgroups that count themselves out and a last one that tells the host would need an agent-scope release each, i.e. a
// write-back of their XCD's L2 under the running sweep: measured, 75 instead of 21 us per tick.)
struct DoneWord {
    uint32_t *word;                // pinned host memory as the device addresses it; NULL: nobody asks
    uint32_t value;
};

__device__ __forceinline__ void previous_launch_is_over(const DoneWord &dw)
{
    if (dw.word && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(dw.word, dw.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same for the overlapped loop's side stream: lists and (M.apply == 0) the ordered event list only.  An entry of its
// own because of its arguments: the 9 KB of radar records that the entry above carries (unused there) made every launch
// cost the side stream's thread 6-8 us, which is what bounded the loop.
__global__ __launch_bounds__(kCompBlock) void k_compact_side(const CompactArgs C, int by_ticket, const MissileArgs M, const DoneWord dw)
{
    __shared__ int s_wave[kCompBlock / 64];
    __shared__ CompactShared<kCompBlock> S;
    previous_launch_is_over(dw);
    if ((int)blockIdx.x >= C.nb) {
        if (M.m > 0 && (int)blockIdx.x == C.nb) missile_finish_entry(s_wave, M);
        return;
    }
    compact_block<kCompBlock>(S, C, by_ticket, (const char *)__builtin_amdgcn_kernarg_segment_ptr());      // (C: the first argument)
}

// The ordered event list of a tick from the per-row codes, by ONE workgroup of any size (each thread owns a run of
// consecutive rows and walks it twice: count, then emit).  The side stream's variant: no tombstones (removals are marks).
__device__ __forceinline__ void missile_events_any(int *s_wave /* [16] */, const MissileArgs &M)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = (int)blockDim.x, waves = T >> 6;
    // each thread's run of rows: a multiple of 16, so that its codes come in with 16-byte loads that are all in flight at once
    // (a thread of a 256-thread workgroup owns 40-odd rows of ten thousand: read a byte at a time, one after the other, that
    // was 30 us of dependent loads -- the whole compaction launch waited for its event workgroups)
    constexpr int kMaxVec = 4;                       // 16-byte pieces per thread: 16 384 rows (1024 * kMissileItems) at 256 threads
    const int64_t per = (((M.m + T - 1) / T) + 15) & ~(int64_t)15;
    const int64_t row0 = (int64_t)tid * per, row1 = (row0 + per < M.m) ? row0 + per : M.m;
    const bool vec = (((uintptr_t)M.ev_code & 15) == 0) && per <= 16 * kMaxVec;
    uint4 cv[kMaxVec];
#pragma unroll
    for (int j = 0; j < kMaxVec; ++j) {
        cv[j] = uint4{0u, 0u, 0u, 0u};
        const int64_t r = row0 + 16 * j;
        if (vec && 16 * j < per && r + 16 <= M.m) cv[j] = *(const uint4 *)(M.ev_code + r);
    }
    int cnt = 0;
    if (vec) {
#pragma unroll
        for (int j = 0; j < kMaxVec; ++j) {
            const uint32_t w[4] = {cv[j].x, cv[j].y, cv[j].z, cv[j].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)                      // bytes that are not zero (codes are 0, 1 or 2)
                cnt += (int)__popc(((w[q] | (w[q] >> 1)) & 0x01010101u));
            const int64_t r = row0 + 16 * j;
            if (16 * j < per && r < M.m && r + 16 > M.m)          // the table's last, partial piece: byte by byte
                for (int64_t row = r; row < M.m; ++row) cnt += M.ev_code[row] != 0;
        }
    } else {
        for (int64_t row = row0; row < row1; ++row) cnt += M.ev_code[row] != 0;
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = incl - cnt, total = 0;
    for (int w = 0; w < waves; ++w) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
    if (tid == 0) {
        *M.ev_count = total;
        if (M.ev_wire) M.ev_wire[0] = total;
    }
    if (cnt == 0) return;
    for (int64_t row = row0; row < row1; ++row) {
        const uint8_t code = M.ev_code[row];
        if (!code) continue;
        const int32_t ms = M.m_slot[row], ts = (code == 1) ? M.m_tgt[row] : -1;
        M.ev_missile[base] = ms; M.ev_target[base] = ts;
        if (M.ev_wire && base < M.ev_wire_cap)                // MissileDetonateMessage (modules/Missile.py:138-146) for the other ranks
            M.ev_wire[1 + base] = (int64_t)(((uint64_t)(M.gid0 + (M.lidx ? M.lidx[ms] : ms)) << 32) |
                                            (ts >= 0 ? (uint64_t)(uint32_t)(M.gid0 + (M.lidx ? M.lidx[ts] : ts)) : 0xFFFFFFFFull));
        ++base;
    }
}

// ---------------------------------------------------------------------------------------------
// The two compactions of a PAIR launch's ticks in ONE launch (side stream): every workgroup squeezes the same slots of
// both ticks' mask buffers, publishes ONE record with both ticks' R + 1 counts and waits ONCE for its predecessors -- the
// single-launch compaction is a chain of dependent round trips, which two ticks then share.  The first tick's lists go to
// buffers of the context's own (a call's intermediate lists are overwritten by the next tick's wherever they are
// written: nobody can have read them), the second tick's where the caller reads.  C1 differs from C0 in vis / zero_next /
// det_idx / det_cnt / packed only.  `rm`: list indices of the rows the first tick's missile phase removed (rm[0] = how
// many), whose bits the sweep's second tick set all the same (SweepParams::t2): taken out of the second tick's masks here.
// ---------------------------------------------------------------------------------------------
// Workgroup size: THREADS threads over kPairSlots list slots (kPairSlots / THREADS per thread and tick).  Beside a sweep
// that fills the register files (seven waves of 72 registers per SIMD), a workgroup of 1024 threads -- four waves on every
// SIMD of one compute unit at once -- finds room only when four sweep workgroups of that unit have retired and none was
// put in their place; smaller workgroups take the room one retiring sweep workgroup leaves.
constexpr int kPairSlots = 4096;
constexpr int kPairItems = kFusedMaxItems / 2;                       // (per thread at 1024 threads)
constexpr int kPairAggStride = 2 * kAggStride;

struct CompactSharedPair {
    int wcnt[2][64];                                                  // (item, wave) counts: kPairSlots / 64 of them
    int cnt[2][ZRK_MAX_RADARS + 1];
    int pre[2 * (ZRK_MAX_RADARS + 1)];
    int grp[2 * (ZRK_MAX_RADARS + 1)];
    int ticket, found[2];
    unsigned short idx[2][kPairSlots];
    uint32_t msk[2][kPairSlots];
};

template <int THREADS>
__device__ __forceinline__ void compact_block_pair(CompactSharedPair &S, const CompactArgs &C0, const CompactArgs &C1, const int32_t *rm,
                                                   int rm_cap, const char *kc0)
{
    constexpr int WAVES = THREADS / 64, kItems = kPairSlots / THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) S.ticket = atomicAdd(&C0.ctl[0], 1);
    if (tid < 2 * (ZRK_MAX_RADARS + 1)) S.pre[tid] = S.grp[tid] = 0;
    __syncthreads();
    const int b = S.ticket;
    if (b < 0 || b >= C0.nb) {                       // a workspace that was not ours: refuse rather than scribble
        if (tid == 0) atomicExch(&C0.ctl[2], 1);
        return;
    }
    const int R = C0.R;
    constexpr int items = kItems;                    // (C0.items says the same: kPairSlots slots per workgroup whatever its size)
    const int64_t blk0 = (int64_t)b * kPairSlots;
    const bool last = b == C0.nb - 1;
    // Every mask load of both ticks in flight before anything looks at one: 32-bit slot numbers against the block's own base
    // addresses; a slot past the table's end reads the block's last one and counts as empty, so that no load of the burst
    // hangs on a branch (as compiled before, each load sat behind a scalar reload of `n` and the first behind the removed-row
    // list's round trip: a chain of ten dependent waits in front of the masks).
    const int rem = (int)((C0.n - blk0) < (int64_t)kPairSlots ? (C0.n - blk0) : (int64_t)kPairSlots);
    uint32_t mk[2][kItems];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t *vb = (s ? C1.vis : C0.vis) + blk0;
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int slot = it * THREADS + tid;
            mk[s][it] = vb[slot < rem ? slot : rem - 1];
        }
    }
    // the rows the first tick removed (few, mostly none): their list indices, asked for while the masks are on their way
    int rmn = rm ? __hip_atomic_load(rm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    rmn = rmn < rm_cap ? rmn : rm_cap;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const CompactArgs &C = s ? C1 : C0;
        uint32_t *zb = C.zero_next ? C.zero_next + blk0 : nullptr;
        const bool own = C.zero_own != 0;
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            const int slot = it * THREADS + tid;
            mk[s][it] = slot < rem ? mk[s][it] : 0u;
            // (the loop's own mask buffers are cleared by the compaction that reads them: only the detections are not zero)
            if (zb && (own ? mk[s][it] != 0u : slot < rem)) zb[slot] = 0u;
            if (C.select) mk[s][it] &= C.select;      // (radars of interest: behind the clearing, which goes by what any radar saw)
        }
    }
    for (int q = 0; q < rmn; ++q) {                  // (wave-uniform trip count; every thread compares its own slots)
        const int64_t off = (int64_t)rm[1 + q] - blk0;
#pragma unroll
        for (int it = 0; it < kItems; ++it)
            if (off == (int64_t)it * THREADS + tid) {
                mk[1][it] = 0u;
                const_cast<uint32_t *>(C1.vis)[blk0 + off] = 0u;     // (the call's last tick: the caller reads this buffer)
            }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const CompactArgs &C = s ? C1 : C0;
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            if (it < items) {
                const unsigned long long bu = __ballot(mk[s][it] != 0u);
                if (lane == 0) {
                    S.wcnt[s][it * WAVES + wave] = (int)__popcll(bu);
                    const int64_t word = (blk0 + (int64_t)it * THREADS + wave * 64) >> 6;      // 64 consecutive slots
                    if (C.packed && C.bits.words && word < C.bits.words) C.packed[2 + word] = (int64_t)bu;
                }
            }
        }
    }
    __syncthreads();
    if (wave < 2) {                                  // exclusive scans of the (item, wave) counts: wave s takes tick s
        const int s = wave, m2 = items * WAVES;
        const int a0 = (2 * lane < m2) ? S.wcnt[s][2 * lane] : 0, a1 = (2 * lane + 1 < m2) ? S.wcnt[s][2 * lane + 1] : 0;
        int incl = a0 + a1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const int excl = incl - a0 - a1;
        if (2 * lane < m2) S.wcnt[s][2 * lane] = excl;
        if (2 * lane + 1 < m2) S.wcnt[s][2 * lane + 1] = excl + a0;
        if (lane == 63) S.found[s] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int it = 0; it < kItems; ++it) {
            if (it < items) {
                const unsigned long long bu = __ballot(mk[s][it] != 0u);
                if (mk[s][it] != 0u) {
                    const int k = S.wcnt[s][it * WAVES + wave] +
                                  (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bu >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bu, 0u));
                    S.idx[s][k] = (unsigned short)(it * THREADS + tid);
                    S.msk[s][k] = mk[s][it];
                }
            }
        }
    }
    for (int t = tid; t < 512; t += THREADS) {       // zero padding of both lists up to the next multiple of 256 entries
        const int s = t >> 8, k = S.found[s] + (t & 255);
        if (k < ((S.found[s] + 255) & ~255)) S.msk[s][k] = 0u;
    }
    __syncthreads();
    // (the lists are zero-padded to a multiple of 256 entries: the walks read them without a predicate; the barrier in front
    // of this is the one behind the squeeze)
    const int len_pad[2] = {(S.found[0] + 255) & ~255, (S.found[1] + 255) & ~255};
    // per-radar counts over the two short lists: the first half of the waves takes the first tick, the second half the second;
    // wave wv of a half walks its list once for radars wv and wv + WPT
    constexpr int WPT = WAVES / 2;
    const int ws = wave / WPT, wv = wave % WPT;
    for (int rb = wv; rb < R; rb += 2 * WPT) {
        const bool two = rb + WPT < R;
        const uint32_t sel[2] = {1u << rb, two ? (1u << (rb + WPT)) : 0u};
        int cnt[2];
        if (two) walk_count<2>(S.msk[ws], len_pad[ws], sel, cnt);
        else { const uint32_t sel1[1] = {sel[0]}; int c1[1]; walk_count<1>(S.msk[ws], len_pad[ws], sel1, c1); cnt[0] = c1[0]; cnt[1] = 0; }
        if (lane == 0) { S.cnt[ws][rb] = cnt[0]; if (two) S.cnt[ws][rb + WPT] = cnt[1]; }
    }
    if (tid < 2) S.cnt[tid][R] = S.found[tid];
    __syncthreads();
    const CompactMid Q = compact_mid(kc0);
    const int nctr = 2 * (R + 1);                    // counter c: tick c / (R + 1), radar (or R: the union) c % (R + 1)
    if (tid < nctr)
        __hip_atomic_store(&Q.agg[(int64_t)b * kPairAggStride + tid],
                           ((unsigned long long)Q.epoch << 32) | (uint32_t)S.cnt[tid / (R + 1)][tid % (R + 1)], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    {
        // lower tickets: record p, counter c is word p * lanes + c of a (p, c) grid dealt out to the threads
        const int lanes = Q.lanes;                  // a power of two >= nctr
        const int c = tid & (lanes - 1);
        const int p_first = tid / lanes, p_step = THREADS / lanes;
        const int G = Q.group, g = G ? b / G : 0, gs = g * G;
        const unsigned long long *gagg = Q.agg + kGroupOffset;
        const uint32_t epoch = Q.epoch;
        bool timed_out = false;
        if (G && b - gs == G - 1) {                   // the group's last ticket: its totals first (see compact_block)
            if (c < nctr) {
                const int in = sum_epoch_records<kPairAggStride>(Q.agg, gs, b, gagg, 0, 0, c, p_first, p_step, epoch, timed_out);
                if (in) { atomicAdd(&S.grp[c], in); atomicAdd(&S.pre[c], in); }
            }
            __syncthreads();
            if (tid < nctr)
                __hip_atomic_store(const_cast<unsigned long long *>(gagg) + (int64_t)g * kPairAggStride + tid,
                                   ((unsigned long long)epoch << 32) | (uint32_t)(S.grp[tid] + S.cnt[tid / (R + 1)][tid % (R + 1)]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c < nctr) {
                const int acc = sum_epoch_records<kPairAggStride>(Q.agg, 0, 0, gagg, 0, g, c, p_first, p_step, epoch, timed_out);
                if (acc) atomicAdd(&S.pre[c], acc);
            }
        } else if (c < nctr) {
            const int acc = G ? sum_epoch_records<kPairAggStride>(Q.agg, gs, b, gagg, 0, g, c, p_first, p_step, epoch, timed_out)
                              : sum_epoch_records<kPairAggStride>(Q.agg, 0, b, gagg, 0, 0, c, p_first, p_step, epoch, timed_out);
            if (acc) atomicAdd(&S.pre[c], acc);
        }
        if (timed_out) atomicExch(&Q.ctl[2], 2);
    }
    __syncthreads();
    const int64_t n_rows = C0.n;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // (where this tick's lists go: asked for here, see CompactLate; C1 stands right behind C0 in the argument segment)
        const CompactLate C = compact_late(kc0 + (s ? sizeof(CompactArgs) : 0));
        const int found = S.found[s];
        const int *pre = S.pre + s * (R + 1);
        if (C.packed) {
            const int64_t ubase = pre[R];
            for (int k = tid; k < found; k += THREADS) {
                const int64_t dst = ubase + k;
                if (C.bits.words) union_bits_mask(C.packed, C.bits, dst, S.msk[s][k]);
                else if (dst + 1 < C.packed_capacity) C.packed[dst + 1] = ((C.gid0 + blk0 + S.idx[s][k]) << 32) | (int64_t)S.msk[s][k];
            }
        }
        if (last && tid <= R) {                          // the end of the list: totals
            const int tot = pre[tid] + S.cnt[s][tid];
            if (C.det_cnt) C.det_cnt[tid] = tot;
            if (C.packed && tid == R) {
                C.packed[0] = tot;
                if (C.bits.words) C.packed[1] = n_rows;
            }
        }
    }
    const CompactLate C = compact_late(kc0 + (ws ? sizeof(CompactArgs) : 0));     // (this half of the waves: its tick's lists)
    if (C.det_idx) {
        const int limit = (int)(C.det_stride < 0x7FFFFFFF ? C.det_stride : 0x7FFFFFFF);
        const int32_t slot0 = C.base_index + (int32_t)blk0;
        for (int rb = wv; rb < R; rb += 2 * WPT) {
            const bool two = rb + WPT < R;
            const uint32_t sel[2] = {1u << rb, two ? (1u << (rb + WPT)) : 0u};
            int run[2] = {__builtin_amdgcn_readfirstlane(S.pre[ws * (R + 1) + rb]),
                          two ? __builtin_amdgcn_readfirstlane(S.pre[ws * (R + 1) + rb + WPT]) : 0};
            int32_t *const out[2] = {C.det_idx + (int64_t)rb * C.det_stride, C.det_idx + (int64_t)(two ? rb + WPT : rb) * C.det_stride};
            if (two) walk_scatter<2>(S.msk[ws], S.idx[ws], len_pad[ws], sel, run, out, limit, slot0);
            else {
                const uint32_t sel1[1] = {sel[0]}; int run1[1] = {run[0]}; int32_t *const out1[1] = {out[0]};
                walk_scatter<1>(S.msk[ws], S.idx[ws], len_pad[ws], sel1, run1, out1, limit, slot0);
            }
        }
    }
    if (tid == 0) {
        int32_t *ctl = compact_mid(kc0).ctl;
        if (atomicAdd(&ctl[1], 1) == C0.nb - 1) {        // everybody holds a ticket and is done with it
            atomicExch(&ctl[0], 0);
            atomicExch(&ctl[1], 0);
            // (the list is this launch's to clear; its address once more from the argument segment -- k_compact_pair's fifth
            // argument, behind the two CompactArgs and the two MissileArgs -- instead of two scalar registers held from the first line on)
            static_assert(sizeof(CompactArgs) % 8 == 0 && sizeof(MissileArgs) % 8 == 0, "k_compact_pair's arguments stand back to back");
            typedef const uint64_t __attribute__((address_space(4))) *Q;
            int32_t *rm_late = (int32_t *)*(Q)(uint64_t)(launder_kernarg(kc0) + 2 * sizeof(CompactArgs) + 2 * sizeof(MissileArgs));
            if (rm_late) __hip_atomic_store(rm_late, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// A call's removal marks carried out and cleared (k_apply_marks) by extra workgroups of the call's LAST compaction launch: a
// launch of its own between the last sweep and that compaction was 6 us of every call (16 rows per thread: one 16-byte load
// of marks, all zero but for a handful).
__device__ __forceinline__ void kill_one(uint8_t *alive, const double *src, double *dst, int64_t cap, int32_t s);
struct MarksArgs {
    uint8_t *pend, *alive;
    double *pos0, *pos1;
    int64_t cap, n;
    int32_t blocks, _pad;
};

template <int THREADS>
__device__ __forceinline__ void apply_marks_block(const MarksArgs &A, int part)
{
    const int64_t r0 = ((int64_t)part * THREADS + threadIdx.x) * 16;
    if (r0 >= A.n) return;
    const uint4 w = *(const uint4 *)(A.pend + r0);          // (pend holds at least `cap` bytes, a multiple of 16 past n)
    if ((w.x | w.y | w.z | w.w) == 0u) return;
    for (int64_t i = r0; i < r0 + 16 && i < A.n; ++i) {
        const uint8_t pk = A.pend[i];
        if (pk == 0) continue;
        // (the mark's low bit: the buffer that was current in the removal tick, i.e. the one that holds the frozen position)
        if (A.alive[i]) { if (pk & 1u) kill_one(A.alive, A.pos1, A.pos0, A.cap, (int32_t)i); else kill_one(A.alive, A.pos0, A.pos1, A.cap, (int32_t)i); }
        A.pend[i] = 0;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_compact_pair(const CompactArgs C0, const CompactArgs C1, const MissileArgs M0,
                                                          const MissileArgs M1, const int32_t *rm, int rm_cap, const DoneWord dw,
                                                          const MarksArgs marks)
{
    __shared__ int s_wave[16];
    __shared__ CompactSharedPair S;
    previous_launch_is_over(dw);
    if ((int)blockIdx.x >= C0.nb) {                  // extra workgroups: the ticks' ordered event lists (two), a call's marks
        const int extra = (int)blockIdx.x - C0.nb, lists = M0.m > 0 ? 2 : 0;
        if (extra < lists) {
            if (extra == 0) missile_events_any(s_wave, M0);
            else missile_events_any(s_wave, M1);
        }
        else apply_marks_block<THREADS>(marks, extra - lists);
        return;
    }
    static_assert(sizeof(CompactArgs) % 8 == 0, "C1 stands right behind C0 in the argument segment");
    compact_block_pair<THREADS>(S, C0, C1, rm, rm_cap, (const char *)__builtin_amdgcn_kernarg_segment_ptr());    // (C0: the first argument)
}

// Overlapped loop of an ENSEMBLE: what the next sweep needs of a tick's second launch -- the tombstones, every
// scenario's scan step and radar records -- as a launch of its own (a dozen workgroups) on the compute stream, while
// the lists and the ordered events are compacted on a side stream beside the next sweep.  (One scenario needs no such
// launch: removal marks, records in the sweep's arguments.)
__global__ __launch_bounds__(kCompBlock) void k_tick_small(const MissileArgs M, const EnsembleArgs E)
{
    int extra = (int)blockIdx.x;
    const int kparts = (int)((M.m + kCompBlock - 1) / kCompBlock);
    if (extra < kparts) { missile_kills(M, extra); return; }
    extra -= kparts;
    if (E.S > 0 && (int64_t)extra * kCompBlock < (int64_t)E.S * E.R) ensemble_derive(E, extra);
}

// SectorRadar.smooth_objects with supplied draws: pos[idx[j]] += noise[j].
__global__ void k_noise_apply(double *__restrict__ pos, int64_t cap, const int32_t *__restrict__ idx,
                              int32_t idx_base, const double *__restrict__ noise, int64_t k)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const int64_t i = (int64_t)idx[j] - idx_base;
    pos[i] += noise[3 * j];
    pos[cap + i] += noise[3 * j + 1];
    pos[2 * cap + i] += noise[3 * j + 2];
}

// Missile.step 'active' branch for one row (modules/Missile.py:162-193).  Returns 0 none, 1 hit, 2 timeout.
// With removal marks (`pend`, the overlapped loop; SweepParams::pend) a row's state is read through them: a mark other than
// this tick's -- and, in the second tick of a PAIR launch (mark_first != 0), the first tick's mark as well -- means "removed
// before this tick" whatever the flag says, because the flag is lowered by the row's own thread somewhere in this very grid.
// The position a removed target froze at: pos_abs[its mark & 1] when an earlier launch removed it.  In a pair's second tick
// the positions "after the tick before" are the first tick's, which this grid is still computing: a target that the first
// tick removed, or that stands behind its missile in the list, has its first-tick radar phase replayed here
// (replay_row_radar_phase) instead of being read.
__device__ __forceinline__ uint8_t missile_step_row(const double *__restrict__ sp, const double *__restrict__ vel,
                                    const double *__restrict__ t0, const uint8_t *alive,
                                    const int32_t *__restrict__ lidx, const double *pos_prev, int64_t cap,
                                    const int32_t *__restrict__ m_slot, const int32_t *__restrict__ m_tgt,
                                    const double *__restrict__ m_radius, double *__restrict__ m_period,
                                    uint8_t *__restrict__ m_status, int64_t row, double t, double dts,
                                    uint8_t *pend, uint32_t mark, const double *grec, const double *pos_abs0, const double *pos_abs1,
                                    uint32_t mark_first, const char *rb_first, int R, bool philox, uint64_t seed, uint64_t tick_first,
                                    int64_t gid0, double t_first)
{
    uint8_t code = 0;
    const int32_t s = m_slot[row];
    const bool second = mark_first != 0u;
    // (a missile that is itself somebody's target and was hit LAST tick is removed from this tick on -- AirEnv.py:33-40 --
    // but only its own row thread, somewhere in this very grid, lowers the flag: the mark says it)
    bool flying = m_status[row] == 1 && alive[s] != 0;
    if (pend && flying) { const uint32_t ps = pend[s]; flying = !(ps != 0u && ps != mark); }
    if (flying) {
        const int32_t j = m_tgt[row];
        double tx, ty, tz;
        // the target's trajectory and list index: one 64-byte record, or seven columns and the index column
        double g[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int32_t lj = j;
        if (grec) {
            const double *r = grec + 8 * (int64_t)j;
#pragma unroll
            for (int q = 0; q < 8; ++q) g[q] = r[q];
            lj = (int32_t)(uint32_t)__builtin_bit_cast(uint64_t, g[7]);
        } else {
            if (lidx) lj = lidx[j];
            g[0] = sp[j]; g[1] = sp[cap + j]; g[2] = sp[2 * cap + j];
            g[3] = vel[j]; g[4] = vel[cap + j]; g[5] = vel[2 * cap + j];
            g[6] = t0[j];
        }
        const bool earlier = lidx ? (lj < lidx[s]) : (j < s);
        const bool alive_j = alive[j] != 0;
        const uint32_t pj = pend ? (uint32_t)pend[j] : 0u;
        const bool old_mark = pj != 0u && pj != mark && pj != mark_first;     // removed by an earlier launch
        const bool there = alive_j && !old_mark && !(second && pj == mark_first);
        if (there && earlier) {           // already stepped this tick (list order): fresh, noise-free
            const double dj = t - g[6];
            tx = g[0] + g[3] * dj; ty = g[1] + g[4] * dj; tz = g[2] + g[5] * dj;
        } else if (old_mark) {            // removed, perhaps not carried out yet: where it froze
            const double *fz = (pj & 1u) ? pos_abs1 : pos_abs0;
            tx = fz[j]; ty = fz[cap + j]; tz = fz[2 * cap + j];
        } else if (second && alive_j) {   // what it held after the pair's first tick, which nobody may have written yet
            const double dj = t_first - g[6];
            tx = g[0] + g[3] * dj; ty = g[1] + g[4] * dj; tz = g[2] + g[5] * dj;
            const Vec3d after = replay_row_radar_phase(rb_first, R, philox, seed, tick_first, (uint64_t)(gid0 + (int64_t)(lidx ? lj : j)), tx, ty, tz);
            tx = after.x; ty = after.y; tz = after.z;
        } else {                          // not stepped yet, or removed: what it held after last tick
            tx = pos_prev[j]; ty = pos_prev[cap + j]; tz = pos_prev[2 * cap + j];
        }
        // (the missile's own position after the target's: the replay above calls out of line, and what lives across a call
        // costs the whole kernel registers)
        const double d = t - t0[s];
        const double px = sp[s] + vel[s] * d, py = sp[cap + s] + vel[cap + s] * d,
                     pz = sp[2 * cap + s] + vel[2 * cap + s] * d;
        const double dx = tx - px, dy = ty - py, dz = tz - pz;
        const double dist = sqrt(dot3(dx, dy, dz, dx, dy, dz));
        if (dist <= m_radius[row]) {
            code = 1; m_status[row] = 2;
        } else {
            const double p = m_period[row] - dts;
            m_period[row] = p;
            if (p <= 0.0) { code = 2; m_status[row] = 2; }
        }
        if (pend && code) {               // removals as marks: first one stands (a mark is never overwritten inside a call)
            if (pend[s] == 0) pend[s] = (uint8_t)mark;
            if (code == 1 && pend[j] == 0) pend[j] = (uint8_t)mark;
        }
    }
    return code;
}

// One thread per in-flight missile (any table size); events are ordered by k_missile_events.
__global__ void k_missile_step(const double *__restrict__ sp, const double *__restrict__ vel,
                               const double *__restrict__ t0, const uint8_t *__restrict__ alive,
                               const int32_t *__restrict__ lidx, const double *__restrict__ pos_prev, int64_t cap,
                               const int32_t *__restrict__ m_slot,
                               const int32_t *__restrict__ m_tgt, const double *__restrict__ m_radius,
                               double *__restrict__ m_period, uint8_t *__restrict__ m_status,
                               uint8_t *__restrict__ ev_code, int64_t m, double t, double dts)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    ev_code[row] = missile_step_row(sp, vel, t0, alive, lidx, pos_prev, cap, m_slot, m_tgt, m_radius, m_period, m_status, row, t,
                                    dts);
}

__device__ __forceinline__ void kill_one(uint8_t *alive, const double *src, double *dst, int64_t cap, int32_t s)
{
    alive[s] = 0;
    dst[s] = src[s]; dst[cap + s] = src[cap + s]; dst[2 * cap + s] = src[2 * cap + s];
}

// Second half of the missile phase in ONE workgroup: the ordered event list out of ev_code (each
// thread owns a run of consecutive rows) and, with apply != 0, the tombstones of the detonated
// missiles and their targets.  Runs after k_missile_step has finished (kernel boundary), so no row
// of this tick sees a half-applied removal.
constexpr int kMissileItems = 16;

__device__ __forceinline__ void missile_finish_block(int *s_wave, const uint8_t *__restrict__ ev_code,
                                                     const int32_t *__restrict__ m_slot,
                                                     const int32_t *__restrict__ m_tgt, int64_t m,
                                                     int32_t *__restrict__ ev_missile, int32_t *__restrict__ ev_target,
                                                     int32_t *__restrict__ ev_count, int apply, uint8_t *alive,
                                                     const double *pos_cur, double *pos_prev, int64_t cap,
                                                     int64_t *ev_wire = nullptr, int ev_wire_cap = 0, int64_t gid0 = 0,
                                                     const int32_t *__restrict__ lidx = nullptr, uint32_t *clear_vis = nullptr,
                                                     double *frozen_prev = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (int)((m + 1023) / 1024);                 // consecutive rows per thread (<= kMissileItems)
    const int64_t row0 = (int64_t)tid * per;
    uint8_t codes[kMissileItems];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < kMissileItems; ++k) {
        const int64_t row = row0 + k;
        codes[k] = (k < per && row < m) ? ev_code[row] : 0;
        cnt += codes[k] != 0;
    }
    // (zrk_ctx_keep_prev) what the rows this tick removes held as prev_pos, BEFORE any tombstone overwrites that buffer: two
    // missiles may remove one target in the same tick, from different threads -- both then keep the same value; the barrier of
    // the scan below stands between this and the tombstones
    if (frozen_prev && apply) {
#pragma unroll
        for (int k = 0; k < kMissileItems; ++k)
            if (codes[k]) {
                const int64_t row = row0 + k;
                const int32_t ms = m_slot[row], ts = (codes[k] == 1) ? m_tgt[row] : -1;
                if (alive[ms]) for (int c = 0; c < 3; ++c) frozen_prev[3 * (int64_t)ms + c] = pos_prev[c * cap + ms];
                if (ts >= 0 && alive[ts]) for (int c = 0; c < 3; ++c) frozen_prev[3 * (int64_t)ts + c] = pos_prev[c * cap + ts];
            }
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = incl - cnt, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
    if (tid == 0) {
        *ev_count = total;
        if (ev_wire) ev_wire[0] = total;
    }
    if (cnt == 0) return;
#pragma unroll
    for (int k = 0; k < kMissileItems; ++k) {
        if (codes[k]) {
            const int64_t row = row0 + k;
            const int32_t ms = m_slot[row], ts = (codes[k] == 1) ? m_tgt[row] : -1;
            ev_missile[base] = ms; ev_target[base] = ts;
            if (ev_wire && base < ev_wire_cap)                // MissileDetonateMessage (modules/Missile.py:138-146) for the other ranks
                ev_wire[1 + base] = (int64_t)(((uint64_t)(gid0 + (lidx ? lidx[ms] : ms)) << 32) |
                                              (ts >= 0 ? (uint64_t)(uint32_t)(gid0 + (lidx ? lidx[ts] : ts)) : 0xFFFFFFFFull));
            ++base;
            if (clear_vis) {                                  // (a pair's first tick: these rows were swept once more -- not seen)
                clear_vis[lidx ? lidx[ms] : ms] = 0u;
                if (ts >= 0) clear_vis[lidx ? lidx[ts] : ts] = 0u;
            }
            if (apply) {                                      // AirEnv.py:33-40, effective from the next tick
                kill_one(alive, pos_cur, pos_prev, cap, ms);
                if (ts >= 0) kill_one(alive, pos_cur, pos_prev, cap, ts);
            }
        }
    }
}

__global__ __launch_bounds__(1024) void k_missile_finish(const uint8_t *__restrict__ ev_code,
                                                         const int32_t *__restrict__ m_slot,
                                                         const int32_t *__restrict__ m_tgt, int64_t m,
                                                         int32_t *__restrict__ ev_missile,
                                                         int32_t *__restrict__ ev_target,
                                                         int32_t *__restrict__ ev_count, int apply, uint8_t *alive,
                                                         const double *pos_cur, double *pos_prev, int64_t cap)
{
    __shared__ int s_wave[16];
    missile_finish_block(s_wave, ev_code, m_slot, m_tgt, m, ev_missile, ev_target, ev_count, apply, alive, pos_cur,
                         pos_prev, cap);
}

// Overlap mode: the tombstones of this tick's detonations, unordered (one thread per missile row, any number of
// workgroups); the ordered event list, nobody's input on the compute stream, is built on the side stream.
__device__ void missile_kills(const MissileArgs &M, int part)
{
    const int64_t row = (int64_t)part * blockDim.x + threadIdx.x;
    if (row >= M.m) return;
    const uint8_t code = M.ev_code[row];
    const int32_t ms = M.m_slot[row], ts = M.m_tgt[row];
    if (!code) return;
    kill_one(M.alive, M.pos_cur, M.pos_prev, M.cap, ms);
    if (code == 1) kill_one(M.alive, M.pos_cur, M.pos_prev, M.cap, ts);
}

__device__ void missile_finish_entry(int *s_wave, const MissileArgs &M)
{
    missile_finish_block(s_wave, M.ev_code, M.m_slot, M.m_tgt, M.m, M.ev_missile, M.ev_target, M.ev_count, M.apply, M.alive,
                         M.pos_cur, M.pos_prev, M.cap, M.ev_wire, M.ev_wire_cap, M.gid0, M.lidx, M.clear_vis, M.frozen_prev);
}

// Ordered event list from ev_code: one workgroup walks the (short) missile table in row order.
__global__ __launch_bounds__(1024) void k_missile_events(const uint8_t *__restrict__ ev_code,
                                                         const int32_t *__restrict__ m_slot,
                                                         const int32_t *__restrict__ m_tgt, int64_t m,
                                                         int32_t *__restrict__ ev_missile,
                                                         int32_t *__restrict__ ev_target, int32_t *__restrict__ ev_count)
{
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < m; base += 1024) {
        const int64_t row = base + tid;
        const uint8_t c = (row < m) ? ev_code[row] : 0;
        const unsigned long long b = __ballot(c != 0);
        if (lane == 0) s_wave[wave] = (int)__popcll(b);
        __syncthreads();
        int off = s_carry, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) off += s_wave[w]; total += s_wave[w]; }
        if (c) {
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const int k = off + (int)__popcll(b & below);
            ev_missile[k] = m_slot[row];
            ev_target[k] = (c == 1) ? m_tgt[row] : -1;
        }
        __syncthreads();
        if (tid == 0) s_carry += total;
        __syncthreads();
    }
    if (tid == 0) *ev_count = s_carry;
}

// The event rows of this tick in wire form (see MissileArgs::ev_wire), for the ticks whose event list was built by
// the stand-alone launches.
__global__ void k_events_wire(const int32_t *__restrict__ ev_missile, const int32_t *__restrict__ ev_target,
                              const int32_t *__restrict__ ev_count, const int32_t *__restrict__ lidx, int64_t gid0,
                              int64_t *__restrict__ ev_wire, int ev_wire_cap)
{
    const int n = *ev_count;
    if (threadIdx.x == 0) ev_wire[0] = n;
    for (int j = threadIdx.x; j < n && j < ev_wire_cap; j += blockDim.x) {
        const int32_t ms = ev_missile[j], ts = ev_target[j];
        ev_wire[1 + j] = (int64_t)(((uint64_t)(gid0 + (lidx ? lidx[ms] : ms)) << 32) |
                                   (ts >= 0 ? (uint64_t)(uint32_t)(gid0 + (lidx ? lidx[ts] : ts)) : 0xFFFFFFFFull));
    }
}

// Gather records (MissileArgs::grec) of rows [lo, hi) from the columns.
__global__ void k_build_gather_records(const double *__restrict__ sp, const double *__restrict__ vel, const double *__restrict__ t0,
                                       const int32_t *__restrict__ lidx, int64_t cap, int64_t lo, int64_t hi, double *__restrict__ grec)
{
    const int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    double *r = grec + 8 * i;
    r[0] = sp[i]; r[1] = sp[cap + i]; r[2] = sp[2 * cap + i];
    r[3] = vel[i]; r[4] = vel[cap + i]; r[5] = vel[2 * cap + i];
    r[6] = t0[i];
    r[7] = __builtin_bit_cast(double, (uint64_t)(uint32_t)(lidx ? lidx[i] : (int32_t)i));
}

// Overlapped loop, behind the last tick of a call: the removals that tick decided (marks nobody has carried out yet)
// as the tombstones the caller expects (kill_one), and every mark of the call cleared.
__global__ void k_apply_marks(uint8_t *__restrict__ pend, uint8_t *alive, double *pos0, double *pos1, int64_t cap, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || pend[i] == 0) return;
    // (the mark's low bit: the buffer that was current in the removal tick, i.e. the one that holds the frozen position)
    if (alive[i]) { if (pend[i] & 1u) kill_one(alive, pos1, pos0, cap, (int32_t)i); else kill_one(alive, pos0, pos1, cap, (int32_t)i); }
    pend[i] = 0;
}

__global__ void k_kill_slots(uint8_t *alive, const double *src, double *dst, int64_t cap,
                             const int32_t *__restrict__ slots, int64_t k)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < k && slots[j] >= 0) kill_one(alive, src, dst, cap, slots[j]);
}

__global__ void k_apply_events(uint8_t *alive, const double *src, double *dst, int64_t cap,
                               const int32_t *__restrict__ ev_missile, const int32_t *__restrict__ ev_target,
                               const int32_t *ev_count, int64_t mcap)
{
    const int n = *ev_count;
    for (int64_t j = threadIdx.x; j < n && j < mcap; j += blockDim.x) {
        kill_one(alive, src, dst, cap, ev_missile[j]);
        if (ev_target[j] >= 0) kill_one(alive, src, dst, cap, ev_target[j]);
    }
}

// Missile._calculate_trajectory_params for one request (modules/Missile.py:35-102).
__device__ __forceinline__ zrk_launch_res launch_solve_one(const double *__restrict__ vel, const uint8_t *__restrict__ kind,
                                                           const double *__restrict__ pos, int64_t cap, const zrk_launch_req &rq)
{
    const int32_t j = rq.target_slot;
    zrk_launch_res out;
    out.rc = 0; out._pad = 0; out.velocity[0] = out.velocity[1] = out.velocity[2] = 0.0; out.t_hit = 0.0;
    if (j < 0 || j >= cap) {                       // no such row (the padding behind a device-built request list): a failed request
        out.rc = 6;
        return out;
    }
    // target.velocity (unit) * target.speed_mod, modules/AirObject.py:35-36, modules/Missile.py:58.
    // A missile used as a target has velocity NaN forever (Missile.py:26-27).
    double vt[3];
    {
        const double vx = vel[j], vy = vel[cap + j], vz = vel[2 * cap + j];
        const double nrm = sqrt(dot3(vx, vy, vz, vx, vy, vz));
        if (kind[j] == 1) {
            const double qnan = __builtin_nan("");
            vt[0] = vt[1] = vt[2] = qnan;
        } else {
            vt[0] = (vx / nrm) * nrm; vt[1] = (vy / nrm) * nrm; vt[2] = (vz / nrm) * nrm;
        }
    }
    const double d0 = pos[j] - rq.missile_pos[0], d1 = pos[cap + j] - rq.missile_pos[1],
                 d2 = pos[2 * cap + j] - rq.missile_pos[2];
    const double v0 = rq.speed;
    const double a = dot3(vt[0], vt[1], vt[2], vt[0], vt[1], vt[2]) - v0 * v0;
    const double b = 2.0 * dot3(d0, d1, d2, vt[0], vt[1], vt[2]);
    const double c = dot3(d0, d1, d2, d0, d1, d2);
    double t = 0.0;
    if (fabs(a) < 1e-6) {
        if (fabs(b) < 1e-6) out.rc = 1;
        else {
            t = -c / b;
            if (t <= 0.0) out.rc = 2;
        }
    } else {
        const double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) out.rc = 3;
        else {
            const double sq = sqrt(disc);
            const double t1 = (-b + sq) / (2.0 * a), t2 = (-b - sq) / (2.0 * a);
            bool have = false;
            if (t1 > 0.0) { t = t1; have = true; }
            if (t2 > 0.0) { if (!have || t2 < t) t = t2; have = true; }
            if (!have) out.rc = 4;
        }
    }
    if (out.rc == 0 && t > rq.period) out.rc = 5;
    if (out.rc == 0) {
        const double w0 = d0 / t + vt[0], w1 = d1 / t + vt[1], w2 = d2 / t + vt[2];
        const double nrm = sqrt(dot3(w0, w1, w2, w0, w1, w2));
        out.velocity[0] = w0 / nrm * v0; out.velocity[1] = w1 / nrm * v0; out.velocity[2] = w2 / nrm * v0;
        out.t_hit = t;
    }
    return out;
}

// ... one thread per request
__global__ void k_launch_solve(const double *__restrict__ vel, const uint8_t *__restrict__ kind,
                               const double *__restrict__ pos, int64_t cap, const zrk_launch_req *__restrict__ req,
                               zrk_launch_res *__restrict__ res, int64_t k)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= k) return;
    res[q] = launch_solve_one(vel, kind, pos, cap, req[q]);
}

// The successful launches of a salvo enter the air (Missile._launch, modules/Missile.py:104-133; MissileLauncher
// hands the missile to AirEnv, modules/MissileLauncher.py:103-124, modules/AirEnv.py:42-43), in request order, without
// the host: an exclusive prefix sum of (rc == 0) gives success number p the next table row n + p -- trajectory
// (V, launcher position, now), alive, kind 1, both position buffers at the launcher -- and the next missile row m + p.
// The k - count requests that failed take the rows behind, dead and inactive, so that the host's upper bounds
// n + k and m + k hold without reading anything back; list indices stay dense for the missiles in the air.
__global__ __launch_bounds__(1024) void k_launch_append(const zrk_launch_req *__restrict__ req,
                                                        const zrk_launch_res *__restrict__ res, int64_t k, double *sp,
                                                        double *vel, double *t0, uint8_t *alive, uint8_t *kind, double *pos0,
                                                        double *pos1, int32_t *lidx, int64_t cap, int64_t n, int32_t list_base,
                                                        int32_t *m_slot, int32_t *m_tgt, double *m_radius, double *m_period,
                                                        uint8_t *m_status, int64_t m, double t, int32_t *count_out)
{
    __shared__ int s_wave[16];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pass 1: how many succeeded (the failures' rows start behind all of them)
    int mine = 0;
    for (int64_t q = tid; q < k; q += 1024) mine += res[q].rc == 0;
    for (int d = 32; d; d >>= 1) mine += __shfl_xor(mine, d);
    if (lane == 0) s_wave[wave] = mine;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += s_wave[w];
        s_total = tot;
        if (count_out) *count_out = tot;
    }
    __syncthreads();
    const int total = s_total;
    int carry_ok = 0, carry_bad = 0;
    for (int64_t base = 0; base < k; base += 1024) {
        const int64_t q = base + tid;
        const bool in = q < k;
        const bool ok = in && res[q].rc == 0;
        const unsigned long long b_ok = __ballot(ok), b_bad = __ballot(in && !ok);
        __syncthreads();
        if (lane == 0) s_wave[wave] = (int)__popcll(b_ok) | ((int)__popcll(b_bad) << 16);
        __syncthreads();
        int o_ok = carry_ok, o_bad = carry_bad, t_ok = 0, t_bad = 0;
        for (int w = 0; w < 16; ++w) {
            const int v = s_wave[w];
            if (w < wave) { o_ok += v & 0xFFFF; o_bad += v >> 16; }
            t_ok += v & 0xFFFF; t_bad += v >> 16;
        }
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (in) {
            const int p = ok ? o_ok + (int)__popcll(b_ok & below) : total + o_bad + (int)__popcll(b_bad & below);
            const int64_t row = n + p, mrow = m + p;
            const zrk_launch_req rq = req[q];
            for (int c = 0; c < 3; ++c) {
                sp[c * cap + row] = rq.missile_pos[c];
                vel[c * cap + row] = ok ? res[q].velocity[c] : 0.0;
                pos0[c * cap + row] = rq.missile_pos[c];
                pos1[c * cap + row] = rq.missile_pos[c];
            }
            t0[row] = t;
            alive[row] = ok ? 1 : 0;
            kind[row] = 1;
            if (lidx) lidx[row] = list_base + p;
            m_slot[mrow] = (int32_t)row; m_tgt[mrow] = rq.target_slot;
            m_radius[mrow] = rq.radius; m_period[mrow] = rq.period;
            m_status[mrow] = ok ? 1 : 0;
        }
        carry_ok += t_ok; carry_bad += t_bad;
    }
}

// ---------------------------------------------------------------------------------------------
// Track association of the command post: CombatControlPoint.link_object
// (modules/CCP.py:171-219) for ALL detections of a tick, with the result the reference's sequential loop
// (modules/CCP.py:414-429) produces.  For one detection the reference scans the target tracks, then the missile
// tracks, and keeps the strictly nearest one whose distance lies in an annulus [speed * (age - slack),
// speed * (age + slack)] clipped at 0, skipping tracks updated this tick; a match updates the track, which takes it
// out of every later detection's scan.  So the tick is a greedy matching with removal, in detection order: the
// pairwise part (every detection against every track) is independent per detection and runs LDS-tiled below; the
// order-dependent part only ever chooses among a detection's few in-gate candidates and is resolved in rounds.
// ---------------------------------------------------------------------------------------------
constexpr int kCcpK = 16;                      // in-gate candidates kept per detection (nearest first)
constexpr int kCcpTile = 1024;                 // tracks staged in LDS per step

struct CcpCand {
    double dist[kCcpK];
    int32_t idx[kCcpK];
    int32_t n;                                 // candidates kept
    int32_t total;                             // candidates in gate (> kCcpK: the list is a prefix)
};

// One thread per detection; the tracks stream through LDS in tiles.  `taken` (may be NULL) excludes tracks already
// given away (the re-scan of a detection whose kept candidates were all taken); `only` (may be NULL) restricts the
// pass to the listed detections.
__global__ __launch_bounds__(256) void k_ccp_candidates(const double *__restrict__ det_pos, const double *__restrict__ det_speed,
                                                        int64_t D, const double *__restrict__ trk_ref,
                                                        const double *__restrict__ trk_upd, int64_t T, double now_s,
                                                        double slack_s, const uint8_t *__restrict__ taken,
                                                        const uint8_t *__restrict__ only, CcpCand *__restrict__ out,
                                                        const int32_t *__restrict__ dev_sizes = nullptr /* {D, T} on the device */,
                                                        const int32_t *__restrict__ gate = nullptr /* run only if *gate != 0 */)
{
    __shared__ double s_ref[3][kCcpTile];
    __shared__ double s_upd[kCcpTile];
    if (gate && *gate == 0) return;
    if (dev_sizes) { D = D < dev_sizes[0] ? D : dev_sizes[0]; T = dev_sizes[1]; }
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool mine = d < D && (!only || only[d]);
    double px = 0.0, py = 0.0, pz = 0.0, sp = 0.0;
    if (mine) { px = det_pos[3 * d]; py = det_pos[3 * d + 1]; pz = det_pos[3 * d + 2]; sp = det_speed[d]; }
    CcpCand c;
    c.n = 0; c.total = 0;
#pragma unroll
    for (int k = 0; k < kCcpK; ++k) { c.dist[k] = __builtin_inf(); c.idx[k] = -1; }
    for (int64_t base = 0; base < T; base += kCcpTile) {
        __syncthreads();
        for (int j = threadIdx.x; j < kCcpTile; j += 256) {
            const int64_t t = base + j;
            const bool ok = t < T && !(taken && taken[t]);
            s_ref[0][j] = ok ? trk_ref[3 * t] : 0.0; s_ref[1][j] = ok ? trk_ref[3 * t + 1] : 0.0;
            s_ref[2][j] = ok ? trk_ref[3 * t + 2] : 0.0;
            s_upd[j] = ok ? trk_upd[t] : now_s;               // updated "now": skipped, modules/CCP.py:188, :203
        }
        __syncthreads();
        if (!mine) continue;
        const int lim = (int)((T - base < kCcpTile) ? (T - base) : kCcpTile);
        for (int j = 0; j < lim; ++j) {
            const double upd = s_upd[j];
            if (upd == now_s) continue;
            // calc_range, modules/CCP.py:176-186: np.linalg.norm(track_pos - detected.pos), the two clipped ranges
            const double dx = s_ref[0][j] - px, dy = s_ref[1][j] - py, dz = s_ref[2][j] - pz;
            const double dist = sqrt(dot3(dx, dy, dz, dx, dy, dz));
            const double age = now_s - upd;
            double lo = sp * (age - slack_s), hi = sp * (age + slack_s);
            lo = (lo > 0.0) ? lo : 0.0; hi = (hi > 0.0) ? hi : 0.0;
            if (!(lo <= dist && dist <= hi)) continue;
            c.total += 1;
            // keep the kCcpK nearest, earlier track first among equals (the scan's strict `<`)
            if (c.n < kCcpK || dist < c.dist[kCcpK - 1]) {
                int pos = (c.n < kCcpK) ? c.n : kCcpK - 1;
#pragma unroll
                for (int k = kCcpK - 1; k > 0; --k) {
                    if (k <= pos && c.dist[k - 1] > dist) { c.dist[k] = c.dist[k - 1]; c.idx[k] = c.idx[k - 1]; pos = k - 1; }
                }
                c.dist[pos] = dist; c.idx[pos] = (int32_t)(base + j);
                if (c.n < kCcpK) c.n += 1;
            }
        }
    }
    if (mine) out[d] = c;
}

// ---------------------------------------------------------------------------------------------
// The candidate pass with a spatial index (many tracks): the tracks are binned, per tick and on the device, by the size of
// their annulus and by where they are, and a detection looks only into the bins its annulus can reach -- with the same
// arithmetic per pair and the same result as the tiled pass above (the kept lists are ordered by (distance, track), which
// is the order the sequential scan's strict `<` produces).
//   A track's annulus for a detection of speed v ends at v * (age + slack): tracks fall into classes by a = age + slack,
// class c holding a <= base * 2^c (base = slack), the last class whatever is left (and what has no finite position): one
// bin.  Class c has a uniform grid over the tracks' x-y extent with cells at least v_max * base * 2^c wide (v_max: the
// fastest detection of the tick), so that no detection's reach spans more than 3 x 3 cells of any class -- unless the cell
// budget caps the grid, which only makes cells larger.  A counting sort by (class, cell) puts the tracks of a cell next to
// each other (their order within a cell does not matter: the kept lists carry their own order).
// ---------------------------------------------------------------------------------------------
constexpr int kGridClasses = 12;

struct CcpGrid {
    double x0, y0, base;
    double inv_h[kGridClasses];
    double reach[kGridClasses];                // base * 2^c: what a = age + slack is at most in class c (the last: unbounded)
    int32_t nx[kGridClasses], ny[kGridClasses], cell0[kGridClasses + 1];
};

struct CcpGridArgs {
    const double *det_pos, *det_speed, *trk_ref, *trk_upd;
    const int32_t *sizes;                      // {D, T} on the device
    int64_t dmax, tmax;
    double now_s, slack_s;
    double *partial;                           // [blocks][5]: min x, max x, min y, max y, max speed
    int32_t blocks;
    CcpGrid *grid;
    int32_t cells_cap;                         // cells available in all
    int32_t *count, *start, *key;              // [cells_cap + 1], [cells_cap + 1], [tmax]
    int32_t *block_sum;                        // [1024]
    int32_t *sorted_idx;                       // [tmax]
    double *sorted_ref, *sorted_upd;           // [tmax][3], [tmax]
};

__device__ __forceinline__ bool finite3(double x, double y, double z)
{
    return fabs(x) < __builtin_inf() && fabs(y) < __builtin_inf() && fabs(z) < __builtin_inf();
}

// extent of the tracks that have a finite position, fastest finite detection: per-block partial results
__global__ __launch_bounds__(256) void k_ccp_grid_partials(const CcpGridArgs G)
{
    __shared__ double s_v[5][256];
    const int D = (int)(G.sizes[0] < G.dmax ? G.sizes[0] : G.dmax), T = (int)(G.sizes[1] < G.tmax ? G.sizes[1] : G.tmax);
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v[5] = {__builtin_inf(), -__builtin_inf(), __builtin_inf(), -__builtin_inf(), 0.0};
    if (g < T) {
        const double x = G.trk_ref[3 * g], y = G.trk_ref[3 * g + 1], z = G.trk_ref[3 * g + 2];
        if (finite3(x, y, z)) { v[0] = v[1] = x; v[2] = v[3] = y; }
    }
    if (g < D) {
        const double sp = G.det_speed[g];
        if (sp > 0.0 && sp < __builtin_inf()) v[4] = sp;
    }
    for (int k = 0; k < 5; ++k) s_v[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int off = 128; off; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_v[0][threadIdx.x] = fmin(s_v[0][threadIdx.x], s_v[0][threadIdx.x + off]);
            s_v[1][threadIdx.x] = fmax(s_v[1][threadIdx.x], s_v[1][threadIdx.x + off]);
            s_v[2][threadIdx.x] = fmin(s_v[2][threadIdx.x], s_v[2][threadIdx.x + off]);
            s_v[3][threadIdx.x] = fmax(s_v[3][threadIdx.x], s_v[3][threadIdx.x + off]);
            s_v[4][threadIdx.x] = fmax(s_v[4][threadIdx.x], s_v[4][threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 5) G.partial[(int64_t)blockIdx.x * 5 + threadIdx.x] = s_v[threadIdx.x][0];
}

// one workgroup: the grids of the tick
__global__ __launch_bounds__(256) void k_ccp_grid_params(const CcpGridArgs G)
{
    __shared__ double s_v[5][256];
    double v[5] = {__builtin_inf(), -__builtin_inf(), __builtin_inf(), -__builtin_inf(), 0.0};
    for (int b = threadIdx.x; b < G.blocks; b += 256) {
        const double *q = G.partial + (int64_t)b * 5;
        v[0] = fmin(v[0], q[0]); v[1] = fmax(v[1], q[1]); v[2] = fmin(v[2], q[2]); v[3] = fmax(v[3], q[3]); v[4] = fmax(v[4], q[4]);
    }
    for (int k = 0; k < 5; ++k) s_v[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int off = 128; off; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_v[0][threadIdx.x] = fmin(s_v[0][threadIdx.x], s_v[0][threadIdx.x + off]);
            s_v[1][threadIdx.x] = fmax(s_v[1][threadIdx.x], s_v[1][threadIdx.x + off]);
            s_v[2][threadIdx.x] = fmin(s_v[2][threadIdx.x], s_v[2][threadIdx.x + off]);
            s_v[3][threadIdx.x] = fmax(s_v[3][threadIdx.x], s_v[3][threadIdx.x + off]);
            s_v[4][threadIdx.x] = fmax(s_v[4][threadIdx.x], s_v[4][threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    CcpGrid g;
    const bool any = s_v[0][0] <= s_v[1][0] && s_v[2][0] <= s_v[3][0];
    g.x0 = any ? s_v[0][0] : 0.0; g.y0 = any ? s_v[2][0] : 0.0;
    const double ex = any ? s_v[1][0] - s_v[0][0] : 0.0, ey = any ? s_v[3][0] - s_v[2][0] : 0.0;
    const double vmax = s_v[4][0];
    g.base = G.slack_s > 1e-9 ? G.slack_s : 1e-9;
    const int per_class = G.cells_cap / kGridClasses;
    int axis = 1;
    while ((axis + 1) * (axis + 1) <= per_class && axis < 2048) ++axis;
    int cell = 0;
    double reach = g.base;
    for (int c = 0; c < kGridClasses; ++c, reach *= 2.0) {
        g.cell0[c] = cell;
        g.reach[c] = reach;
        int nx = 1, ny = 1;
        double h = vmax * reach;                                  // no detection reaches further than this in class c
        if (c < kGridClasses - 1 && any) {
            // (at least the reach, at least what the cell budget allows; a zero reach -- nobody moves -- takes the finest grid)
            const double hx = fmax(h, ex / axis), hy = fmax(h, ey / axis);
            h = fmax(hx, hy);
            if (h > 0.0 && h < __builtin_inf()) {
                nx = (int)fmin((double)axis, floor(ex / h) + 1.0);
                ny = (int)fmin((double)axis, floor(ey / h) + 1.0);
            }
        }
        g.nx[c] = nx; g.ny[c] = ny;
        g.inv_h[c] = (nx > 1 || ny > 1) ? 1.0 / h : 0.0;
        cell += nx * ny;
    }
    g.cell0[kGridClasses] = cell;
    *G.grid = g;
}

__device__ __forceinline__ int grid_class(const CcpGrid &g, double a)
{
    int c = 0;
    while (c < kGridClasses - 1 && !(a <= g.reach[c])) ++c;       // (a NaN age lands in the last class)
    return c;
}

__device__ __forceinline__ int grid_axis_cell(double v, double v0, double inv_h, int n)
{
    const double q = floor((v - v0) * inv_h);
    return q < 0.0 ? 0 : (q > (double)(n - 1) ? n - 1 : (int)q);  // (NaN: 0)
}

__global__ __launch_bounds__(256) void k_ccp_grid_count(const CcpGridArgs G)
{
    const int T = (int)(G.sizes[1] < G.tmax ? G.sizes[1] : G.tmax);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const CcpGrid &g = *G.grid;
    const double x = G.trk_ref[3 * t], y = G.trk_ref[3 * t + 1], z = G.trk_ref[3 * t + 2];
    int c = kGridClasses - 1;
    if (finite3(x, y, z)) c = grid_class(g, (G.now_s - G.trk_upd[t]) + G.slack_s);
    const int ix = grid_axis_cell(x, g.x0, g.inv_h[c], g.nx[c]), iy = grid_axis_cell(y, g.y0, g.inv_h[c], g.ny[c]);
    const int key = g.cell0[c] + iy * g.nx[c] + ix;
    G.key[t] = key;
    atomicAdd(&G.count[key], 1);
}

// exclusive scan of count[0 .. cells_cap] into start[]: blocks of 4096 cells, their sums, then the offsets
__global__ __launch_bounds__(1024) void k_ccp_grid_scan(const CcpGridArgs G, int phase)
{
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = G.cells_cap + 1;
    if (phase == 1) {                                             // one workgroup: the block sums
        const int nb = (n + 4095) / 4096;
        int v = tid < nb ? G.block_sum[tid] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (tid < nb) G.block_sum[tid] = off + incl - v;
        return;
    }
    const int base = (int)blockIdx.x * 4096 + tid * 4;
    int v[4], sum = 0;
    for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? G.count[base + k] : 0; sum += v[k]; }
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int off = 0, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) off += s_w[w]; total += s_w[w]; }
    if (phase == 0) { if (tid == 0) G.block_sum[blockIdx.x] = total; return; }
    int run = G.block_sum[blockIdx.x] + off + incl - sum;
    for (int k = 0; k < 4; ++k) { if (base + k < n) G.start[base + k] = run; run += v[k]; }
}

__global__ __launch_bounds__(256) void k_ccp_grid_scatter(const CcpGridArgs G)
{
    const int T = (int)(G.sizes[1] < G.tmax ? G.sizes[1] : G.tmax);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const int key = G.key[t];
    const int at = G.start[key] + (atomicAdd(&G.count[key], -1) - 1);     // (the counts run down to zero: cleared for the next tick)
    G.sorted_idx[at] = (int32_t)t;
    G.sorted_ref[3 * (int64_t)at] = G.trk_ref[3 * t]; G.sorted_ref[3 * (int64_t)at + 1] = G.trk_ref[3 * t + 1];
    G.sorted_ref[3 * (int64_t)at + 2] = G.trk_ref[3 * t + 2];
    G.sorted_upd[at] = G.trk_upd[t];
}

// One thread per detection, through the bins its annuli can reach.  Same outputs as k_ccp_candidates.
// The kCcpK nearest (nearer first, the earlier track first among equals) of the n <= kCandBuf (distance, track) pairs a wave has
// collected, left sorted in the buffer's first min(n, kCcpK) places.  The whole wave; the buffer is the wave's own.
constexpr int kCandBuf = 256;
__device__ __forceinline__ void wave_select_nearest(double *bd, int *bi, int n)
{
    const int lane = (int)(threadIdx.x & 63);
    constexpr int kPer = kCandBuf / 64;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double ed[kPer];
    int ei[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int e = k * 64 + lane;
        ed[k] = e < n ? bd[e] : __builtin_inf();
        ei[k] = e < n ? bi[e] : 0x7FFFFFFF;
    }
    double rd = __builtin_inf();
    int ri = -1;
    const int keep = n < kCcpK ? n : kCcpK;
    for (int r = 0; r < keep; ++r) {
        double md = ed[0];
        int mi = ei[0];
#pragma unroll
        for (int k = 1; k < kPer; ++k)
            if (ed[k] < md || (ed[k] == md && ei[k] < mi)) { md = ed[k]; mi = ei[k]; }
        double wd = md;
        int wi = mi;
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const double od = __shfl_xor(wd, off);
            const int oi = __shfl_xor(wi, off);
            if (od < wd || (od == wd && oi < wi)) { wd = od; wi = oi; }
        }
        if (lane == r) { rd = wd; ri = wi; }
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (ei[k] == wi && ed[k] == wd) { ed[k] = __builtin_inf(); ei[k] = 0x7FFFFFFF; }      // (a track stands in the buffer once)
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < keep) { bd[lane] = rd; bi[lane] = ri; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The candidate pass over the spatial index: ONE WAVE per detection -- its lanes take the tracks of the cells its gate can reach
// side by side, what is in gate goes into the wave's buffer in LDS, and the nearest kCcpK are picked out of it (whenever it
// fills up, and at the end).  A thread per detection, walking those cells alone and keeping its sixteen in registers, was
// 1.5 ms of a C2-battery tick: fifteen thousand detections are 58 workgroups, one wave per SIMD on a fifth of the device, each
// a chain of a few thousand dependent loads.
__global__ __launch_bounds__(256) void k_ccp_candidates_grid(const CcpGridArgs G, const uint8_t *__restrict__ taken,
                                                             const uint8_t *__restrict__ only, CcpCand *__restrict__ out,
                                                             const int32_t *__restrict__ gate)
{
    __shared__ double s_bd[4][kCandBuf];
    __shared__ int s_bi[4][kCandBuf];
    if (gate && *gate == 0) return;
    const int D = (int)(G.sizes[0] < G.dmax ? G.sizes[0] : G.dmax);
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int64_t d = (int64_t)blockIdx.x * 4 + wave;
    if (d >= D || (only && !only[d])) return;                       // (wave-uniform)
    double *bd = s_bd[wave];
    int *bi = s_bi[wave];
    const CcpGrid &g = *G.grid;
    const double px = G.det_pos[3 * d], py = G.det_pos[3 * d + 1], pz = G.det_pos[3 * d + 2], sp = G.det_speed[d];
    const double now_s = G.now_s, slack_s = G.slack_s;
    int nbuf = 0, total = 0;
    if (sp == sp) {                                                // (a NaN speed: every gate fails)
        for (int cl = 0; cl < kGridClasses; ++cl) {
            const int nx = g.nx[cl], ny = g.ny[cl];
            int ix0 = 0, ix1 = nx - 1, iy0 = 0, iy1 = ny - 1;
            if (nx > 1 || ny > 1) {
                // how far a track of this class can be and still be in gate: v * reach, a little more for the roundings between
                // the distance's square root and the coordinates
                double r = sp * g.reach[cl];
                r = (r > 0.0 ? r * (1.0 + 1e-9) : 0.0) + 1e-9 + 1e-12 * (fabs(px) + fabs(py) + fabs(g.x0) + fabs(g.y0));
                ix0 = grid_axis_cell(px - r, g.x0, g.inv_h[cl], nx); ix1 = grid_axis_cell(px + r, g.x0, g.inv_h[cl], nx);
                iy0 = grid_axis_cell(py - r, g.y0, g.inv_h[cl], ny); iy1 = grid_axis_cell(py + r, g.y0, g.inv_h[cl], ny);
                if (!(px == px) || !(py == py)) { ix0 = 0; ix1 = -1; }       // (no distance from a NaN position is in any gate)
            }
            for (int iy = iy0; iy <= iy1; ++iy) {
                const int row = g.cell0[cl] + iy * nx;
                const int j0 = G.start[row + ix0], j1 = G.start[row + ix1 + 1];     // (the cells of a row are neighbours in the sort)
                for (int jb = j0; jb < j1; jb += 64) {
                    const int j = jb + lane;
                    bool in = false;
                    double dist = 0.0;
                    int t = -1;
                    if (j < j1) {
                        const double upd = G.sorted_upd[j];
                        t = G.sorted_idx[j];
                        if (upd != now_s && !(taken && taken[t])) {
                            const double dx = G.sorted_ref[3 * (int64_t)j] - px, dy = G.sorted_ref[3 * (int64_t)j + 1] - py,
                                         dz = G.sorted_ref[3 * (int64_t)j + 2] - pz;
                            dist = sqrt(dot3(dx, dy, dz, dx, dy, dz));
                            const double age = now_s - upd;
                            double lo = sp * (age - slack_s), hi = sp * (age + slack_s);
                            lo = (lo > 0.0) ? lo : 0.0; hi = (hi > 0.0) ? hi : 0.0;
                            in = lo <= dist && dist <= hi;
                        }
                    }
                    const unsigned long long ib = __ballot(in);
                    if (!ib) continue;
                    const int cnt = (int)__popcll(ib);
                    total += cnt;
                    if (nbuf + cnt > kCandBuf) { wave_select_nearest(bd, bi, nbuf); nbuf = nbuf < kCcpK ? nbuf : kCcpK; }
                    if (in) {
                        const int pos = nbuf + (int)__popcll(ib & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                        bd[pos] = dist; bi[pos] = t;
                    }
                    nbuf += cnt;
                }
            }
        }
    }
    wave_select_nearest(bd, bi, nbuf);
    const int kept = nbuf < kCcpK ? nbuf : kCcpK;
    CcpCand *o = out + d;
    if (lane < kCcpK) { o->dist[lane] = lane < kept ? bd[lane] : __builtin_inf(); o->idx[lane] = lane < kept ? bi[lane] : -1; }
    if (lane == 0) { o->n = kept; o->total = total; }
}

// One round of the order-dependent part.  Phase 0: every unresolved detection names the first candidate of its list
// that nobody has been given yet and registers, for every such candidate still in its list, the smallest index of an
// unresolved detection interested in it.  Phase 1: a detection whose named track has nobody earlier interested in it
// takes it for good -- no earlier detection can ever claim it, and nothing nearer in its own list is free.
// `kill` (may be NULL): the target track KEYED by the detection's own row, which a NEW_TARGET verdict replaces in place
// (add_target on an existing key, modules/CCP.py:88-93: updated "now", hence gone for every later detection of the tick) --
// an unresolved detection is "interested" in that track too, and takes it out when it resolves as new.
// `dcount` (may be NULL): the number of detections lives on the device (zrk_ccp_step); counters[3] != 0: all resolved,
// the launch has nothing to do.
__global__ void k_ccp_round(int phase, int64_t D, const CcpCand *__restrict__ cand, uint8_t *taken, int32_t *interest,
                            int32_t *match, uint8_t *state /* 0 unresolved, 1 resolved, 2 needs a re-scan */, int32_t *counters,
                            const int32_t *__restrict__ kill = nullptr, const int32_t *__restrict__ dcount = nullptr)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (dcount && (counters[3] != 0 || d >= *dcount)) return;
    if (d >= D || state[d] == 1) return;
    const int32_t kx = kill ? kill[d] : -1;
    if (phase == 0 && kx >= 0 && !taken[kx]) atomicMin(&interest[kx], (int32_t)d);
    if (state[d] == 2) {                         // waiting for its re-scan: still holds everybody behind it
        if (phase == 1) atomicMin(&counters[2], (int32_t)d);
        return;
    }
    const CcpCand &c = cand[d];
    int first = -1;
    bool safe = false;                           // some free candidate of its list that nobody earlier is interested in
    for (int k = 0; k < c.n; ++k) {
        const int t = c.idx[k];
        if (taken[t]) continue;
        if (first < 0) first = t;
        if (phase == 0) atomicMin(&interest[t], (int32_t)d);
        else if (interest[t] == (int32_t)d) safe = true;
    }
    if (phase == 0) return;
    if (phase == 1) {
        // A detection whose kept list is only a prefix of what is in its gate may yet turn to tracks it has not named -- but
        // only if every free candidate it kept can still be taken from it by somebody earlier.  With one that nobody earlier
        // wants it ends inside its list whatever happens; without, nobody behind it takes anything for good before it is
        // settled.  (The reference's gates are a hundred steps wide, modules/constants.py:31: in a dense scene EVERY list is a
        // prefix, and nearly every detection's nearest free track is its own, uncontested.)
        if (c.total > c.n && !safe) atomicMin(&counters[2], (int32_t)d);
        return;
    }
    if ((int32_t)d > counters[2]) { atomicAdd(&counters[0], 1); return; }
    if (first < 0) {
        if (c.total > c.n) { state[d] = 2; atomicAdd(&counters[1], 1); }      // the kept prefix is used up: look again
        else { match[d] = -1; state[d] = 1; if (kx >= 0) taken[kx] = 1; }     // NEW_TARGET (its key's old track is replaced)
        return;
    }
    if (interest[first] == (int32_t)d) { match[d] = first; state[d] = 1; taken[first] = 1; }
    else atomicAdd(&counters[0], 1);                                          // still waiting for somebody earlier
}

__global__ void k_ccp_rescan_prepare(int64_t D, uint8_t *only, uint8_t *state)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const bool again = only[d] == 2;
    only[d] = again ? 1 : 0;
    if (again) state[d] = 0;
}

__global__ void k_ccp_reset_interest(int64_t D, const CcpCand *__restrict__ cand, const uint8_t *__restrict__ state, int32_t *interest,
                                     const int32_t *__restrict__ kill = nullptr, const int32_t *__restrict__ dcount = nullptr,
                                     const int32_t *__restrict__ counters = nullptr)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (dcount && (counters[3] != 0 || d >= *dcount)) return;
    if (d >= D || state[d] == 1) return;
    const CcpCand &c = cand[d];
    for (int k = 0; k < c.n; ++k) interest[c.idx[k]] = 0x7FFFFFFF;
    if (kill && kill[d] >= 0) interest[kill[d]] = 0x7FFFFFFF;
}


// ---------------------------------------------------------------------------------------------
// The command post's detection loop of one tick on the device: CombatControlPoint.step,
// modules/CCP.py:406-429 -- link_object for every detection (:171-219), new_target / old_target / old_rocket (:322-366)
// with try_to_launch_missile (:287-320) -- with the result of the reference's sequential loop, without a read-back:
// the detections are rows of the entity table, the tracks live in device arrays in the dictionaries' order, the
// order-dependent part runs a bounded number of rounds that end themselves on a device flag.
// ---------------------------------------------------------------------------------------------
struct CcpStepArgs {
    const double *pos_cur, *pos_prev, *t0;        // entity table: obj.pos, obj.prev_pos, start_time (prev_pos is None where t0 == now)
    const double *speed;                          // obj.speed_mod per row
    int64_t cap;
    const int32_t *seq, *seq_count;               // rows in processing order (each row once), their number (DEVICE)
    int64_t dmax;
    zrk_ccp_tracks trk;
    zrk_ccp_launchers lch;
    zrk_ccp_out out;
    double now_s, slack_s;
    // scratch
    double *det_pos, *det_speed, *trk_ref, *trk_upd;
    int32_t *kill, *sizes /* {D, T, Tt} */, *want, *new_rank, *winner, *counters;
};

// sizes, per-detection attributes, per-track references (targets then missiles, the order link_object scans them in)
struct CcpClearArgs {
    uint32_t *zero[4];
    int64_t zero_words[4];
    uint32_t *ones;                 // 0x7F7F7F7F: "nobody interested"
    int64_t ones_words;
    int32_t *status;
};

__global__ void k_ccp_clear(const CcpClearArgs Z)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (Z.zero[k] && i < Z.zero_words[k]) Z.zero[k][i] = 0u;
    if (i < Z.ones_words) Z.ones[i] = 0x7F7F7F7Fu;
    if (i == 0) *Z.status = 0;
}

__global__ void k_ccp_gather(const CcpStepArgs A)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int D = *A.seq_count;
    D = D < (int)A.dmax ? D : (int)A.dmax;
    const int Tt = A.trk.counts[0], Tm = A.trk.counts[1];
    if (g == 0) { A.sizes[0] = D; A.sizes[1] = Tt + Tm; A.sizes[2] = Tt; A.out.count[0] = D; }
    if (g < D) {
        const int32_t row = A.seq[g];
        A.det_pos[3 * g] = A.pos_cur[row]; A.det_pos[3 * g + 1] = A.pos_cur[A.cap + row]; A.det_pos[3 * g + 2] = A.pos_cur[2 * A.cap + row];
        A.det_speed[g] = A.speed[row];
        A.kill[g] = A.trk.key_tt[row];
        A.out.obj[g] = row;
    }
    if (g < Tt + Tm) {
        const bool missile = g >= Tt;
        const int32_t h = missile ? A.trk.tm_obj[g - Tt] : A.trk.tt_obj[g];
        const double upd = missile ? A.trk.tm_upd[g - Tt] : A.trk.tt_upd[g];
        // (an object that left the air keeps the prev_pos of its last step in the reference; the table keeps only its last
        // position, in both buffers -- the caller supplies what the handle held: zrk_ccp_tracks::ref_fixed)
        const double *fixed = missile ? A.trk.tm_ref_fixed : A.trk.tt_ref_fixed;
        const int t = missile ? (int)(g - Tt) : (int)g;
        // ... or, per ROW, what the loop itself kept when the row left the air (zrk_ctx_keep_prev)
        const double *rowfix = A.trk.row_ref_fixed;
        if (fixed && fixed[3 * t] == fixed[3 * t]) {                 // (not NaN: set)
            A.trk_ref[3 * g] = fixed[3 * t]; A.trk_ref[3 * g + 1] = fixed[3 * t + 1]; A.trk_ref[3 * g + 2] = fixed[3 * t + 2];
        } else if (rowfix && rowfix[3 * (int64_t)h] == rowfix[3 * (int64_t)h]) {
            A.trk_ref[3 * g] = rowfix[3 * (int64_t)h]; A.trk_ref[3 * g + 1] = rowfix[3 * (int64_t)h + 1]; A.trk_ref[3 * g + 2] = rowfix[3 * (int64_t)h + 2];
        } else {
            const bool none = A.t0[h] == A.now_s;                    // AirObject.py:41: prev_pos is None in the object's first tick
            // a missile track falls back to pos (:211-213); for a target track the reference raises (None - array): status 2
            if (none && !missile && upd != A.now_s) atomicMax(A.out.status, 2);
            const double *ref = (none && missile) ? A.pos_cur : A.pos_prev;
            A.trk_ref[3 * g] = ref[h]; A.trk_ref[3 * g + 1] = ref[A.cap + h]; A.trk_ref[3 * g + 2] = ref[2 * A.cap + h];
        }
        A.trk_upd[g] = upd;
        A.winner[g] = -1;
    }
}

__global__ void k_ccp_round_begin(int32_t *counters)
{
    if (counters[3] != 0) return;
    counters[0] = 0; counters[1] = 0; counters[2] = 0x7FFFFFFF; counters[4] = 0;
}

// the round is over: everybody resolved -> done; some lists ran out although more was in gate -> the re-scan below runs.
// The unresolved detections' interest words cleared for the next round (k_ccp_reset_interest), and who is to be looked at again
// marked, in one launch: a detection's interest words are its own lists', its state and `only` bytes its own.
__global__ void k_ccp_round_close(int64_t dmax, const CcpCand *__restrict__ cand, uint8_t *state, int32_t *interest,
                                  const int32_t *__restrict__ kill, const int32_t *__restrict__ sizes, uint8_t *only, int32_t *counters)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (counters[3] != 0 || d >= dmax || d >= sizes[0]) return;
    if (state[d] != 1) {
        const CcpCand &c = cand[d];
        for (int k = 0; k < c.n; ++k) interest[c.idx[k]] = 0x7FFFFFFF;
        if (kill && kill[d] >= 0) interest[kill[d]] = 0x7FFFFFFF;
    }
    const bool again = counters[1] > 0 && state[d] == 2;
    only[d] = again ? 1 : 0;
    if (again) state[d] = 0;
}

__global__ void k_ccp_round_flags(int32_t *counters)
{
    if (counters[3] != 0) return;
    if (counters[0] == 0 && counters[1] == 0) counters[3] = 1;
    counters[4] = counters[1] > 0 ? 1 : 0;
}

// One workgroup: which detections ask for a missile (new targets, old targets nobody follows yet: :330, :342-343), in
// order; the rank of every new target without a track of its key among such (the slot its new track takes).
__global__ __launch_bounds__(1024) void k_ccp_scan(const CcpStepArgs A, const int32_t *__restrict__ match)
{
    __shared__ int s_w[16], s_n[16];
    __shared__ int s_cw, s_cn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = A.sizes[0], Tt = A.sizes[2];
    if (tid == 0) { s_cw = 0; s_cn = 0; }
    __syncthreads();
    for (int base = 0; base < D; base += 1024) {
        const int d = base + tid;
        bool want = false, fresh = false;
        if (d < D) {
            const int m = match[d];
            const int verdict = m < 0 ? 0 : (m < Tt ? 1 : 2);
            A.out.verdict[d] = verdict;
            A.out.match[d] = m < 0 ? -1 : (m < Tt ? m : m - Tt);
            A.out.launcher[d] = -1;
            want = verdict == 0 || (verdict == 1 && !A.trk.tt_follow[m]);
            fresh = verdict == 0 && A.kill[d] < 0;
        }
        const unsigned long long bw = __ballot(want), bn = __ballot(fresh);
        if (lane == 0) { s_w[wave] = (int)__popcll(bw); s_n[wave] = (int)__popcll(bn); }
        __syncthreads();
        int ow = s_cw, on = s_cn, tw = 0, tn = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) { ow += s_w[w]; on += s_n[w]; } tw += s_w[w]; tn += s_n[w]; }
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (want) A.want[1 + ow + (int)__popcll(bw & below)] = d;
        if (d < D) A.new_rank[d] = on + (int)__popcll(bn & below);
        __syncthreads();
        if (tid == 0) { s_cw += tw; s_cn += tn; }
        __syncthreads();
    }
    if (tid == 0) { A.want[0] = s_cw; A.sizes[3] = s_cn; }
}

// try_to_launch_missile for the asking detections, in order (modules/CCP.py:287-320): the nearest launcher that still has
// a missile, strictly nearer wins, earlier launcher among equals; its count goes up.  One wave, lane l = launcher l: the
// loop is sequential in the launchers' counts and runs at most as long as there are missiles left.
// (dist: sqrt of the sum of squares; the reference's `(...) ** 0.5` is pow(x, 0.5), which may differ from sqrt in the last
// bit -- it only ever decides between two launchers whose distances agree to 1 ulp.)
__global__ __launch_bounds__(64) void k_ccp_launch(const CcpStepArgs A)
{
    const int lane = threadIdx.x, L = A.lch.L;
    const bool mine = lane < L;
    const double lx = mine ? A.lch.pos[3 * lane] : 0.0, ly = mine ? A.lch.pos[3 * lane + 1] : 0.0, lz = mine ? A.lch.pos[3 * lane + 2] : 0.0;
    const int capacity = mine ? A.lch.capacity[lane] : 0;
    int launched = mine ? A.lch.launched[lane] : 0;
    const int n = A.want[0];
    int launches = 0;
    for (int q = 0; q < n; ++q) {
        if (!__ballot(mine && launched < capacity)) break;         // nobody has a missile left: every later request fails
        const int d = A.want[1 + q];
        const double ax = lx - A.det_pos[3 * d], ay = ly - A.det_pos[3 * d + 1], az = lz - A.det_pos[3 * d + 2];
        double dist = (mine && launched < capacity) ? sqrt((ax * ax + ay * ay) + az * az) : __builtin_inf();
        if (dist != dist) dist = __builtin_inf();                  // (NaN never compares less: such a launcher is never chosen)
        // minimum over the lanes, the lowest lane among equals
        double best = dist;
        int who = lane;
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int ow = __shfl_xor(who, off);
            if (ob < best || (ob == best && ow < who)) { best = ob; who = ow; }
        }
        if (best < __builtin_inf()) {
            if (lane == who) launched += 1;
            if (lane == 0) A.out.launcher[d] = who;
            ++launches;
        }
    }
    if (mine) A.lch.launched[lane] = launched;
    if (lane == 0) A.out.count[1] = launches;
}

// The dictionaries after the tick.  Every detection writes one track: the one it matched, the one its key names (a new
// target whose id is a key already: replaced in place, :88-93), or a new one behind the others in detection order.  Two
// detections may write the same target track (matched early in the tick, replaced later): the later one stands.
__global__ void k_ccp_apply(const CcpStepArgs A, int phase)
{
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D = A.sizes[0], Tt = A.sizes[2];
    if (d >= D) return;
    const int verdict = A.out.verdict[d];
    const int32_t row = A.out.obj[d];
    if (verdict == 2) {                                             // old_rocket, :363-366
        if (phase == 1) { const int t = A.out.match[d]; A.trk.tm_obj[t] = row; A.trk.tm_upd[t] = A.now_s; }
        return;
    }
    const int w = verdict == 1 ? A.out.match[d] : (A.kill[d] >= 0 ? A.kill[d] : Tt + A.new_rank[d]);
    if (w >= A.trk.capacity) { atomicMax(A.out.status, 3); return; }
    if (phase == 0) { if (w < Tt) atomicMax(&A.winner[w], (int32_t)d); return; }
    if (w < Tt && A.winner[w] != (int32_t)d) return;
    const bool launched = A.out.launcher[d] >= 0;
    if (verdict == 0) {                                             // new_target -> add_target, :322-332
        if (w >= Tt) { A.trk.tt_key[w] = row; A.trk.key_tt[row] = w; }
        A.trk.tt_follow[w] = launched ? 1 : 0;
    } else if (!A.trk.tt_follow[w]) {                               // old_target, :334-361
        A.trk.tt_follow[w] = launched ? 1 : 0;
    }
    A.trk.tt_obj[w] = row; A.trk.tt_upd[w] = A.now_s;
}

__global__ void k_ccp_count_new(const CcpStepArgs A)
{
    A.trk.counts[0] = A.sizes[2] + A.sizes[3];                      // the new targets' tracks stand behind the old ones
}

// The rounds are over and somebody is still unresolved (a long chain of detections each waiting for the one before it: every
// round settles only its head): ONE workgroup walks the unresolved detections in order, as the reference's loop does, with
// the scan over the tracks spread over its threads -- the strictly nearest free track in the annulus, the earlier track
// among equals.  What the rounds gave away stays given (nobody earlier could have claimed it).  Bounded and exact, so the
// step never ends undecided; the rounds are what makes the common case parallel.
__global__ __launch_bounds__(1024) void k_ccp_tail(const CcpStepArgs A, uint8_t *taken, int32_t *match, uint8_t *state, int32_t *counters,
                                                   const CcpCand *__restrict__ cand)
{
    __shared__ double s_dist[16];
    __shared__ int s_trk[16];
    // The detections the rounds have left (a tenth of them at configs[1] scale with a salvo in the air), in order.  Stepping through
    // ALL detections and skipping the settled ones by their state byte -- a dependent load each -- was 1.8 ms of a C2-battery
    // tick's 2.7, so the open ones are squeezed into a list first, kTailChunk detections at a time; and their kept lists with
    // those tracks' `taken` bytes are fetched kTailBatch detections TOGETHER (two round trips a batch instead of two a detection),
    // after which the first wave settles the batch out of LDS, one detection after the other -- the tracks that have gone (picked, or
    // killed) since the batch was fetched are one per lane, a compare and a vote to look through.
    // A detection's kept list is the nearest in-gate tracks that were free when the list was made, nearest first (the order the
    // sequential scan's strict < produces); tracks are only ever taken, so the first entry that is still free IS the scan's
    // answer, and a complete list with no free entry means "no track".  Only a truncated list whose every entry has been taken
    // since -- the middle of a salvo that flies in a cluster -- sends the detection through the scan of all tracks below (a salvo
    // of a thousand missiles two ticks off the rails made this workgroup scan 10^5 tracks a thousand times: 25 ms a tick); the
    // batch starts afresh behind it.
    constexpr int kTailChunk = 8192, kTailBatch = 64;
    __shared__ int s_list[kTailChunk];
    __shared__ int s_cnt[16];
    __shared__ int s_n;
    __shared__ int s_bidx[kTailBatch * kCcpK];
    __shared__ uint8_t s_btaken[kTailBatch * kCcpK];
    __shared__ int s_bkill[kTailBatch];
    __shared__ uint8_t s_bfull[kTailBatch];
    static_assert(kCcpK <= 64 && (kTailBatch * kCcpK) % 64 == 0, "a lane per kept candidate; the strike-out goes by 64 entries");
    if (counters[3] != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = A.sizes[0], T = A.sizes[1];
  for (int base = 0; base < D; base += kTailChunk) {
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int sub = 0; sub < kTailChunk && base + sub < D; sub += 1024) {
        const int dd = base + sub + tid;
        const bool open = dd < D && state[dd] != 1;
        const unsigned long long ob = __ballot(open);
        if (lane == 0) s_cnt[wave] = (int)__popcll(ob);
        __syncthreads();
        int off = s_n, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) off += s_cnt[w]; tot += s_cnt[w]; }
        if (open) s_list[off + (int)__popcll(ob & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = dd;
        __syncthreads();
        if (tid == 0) s_n += tot;
        __syncthreads();
    }
    const int n_open = s_n;
    for (int q0 = 0; q0 < n_open;) {
        const int nbat = (n_open - q0 < kTailBatch) ? n_open - q0 : kTailBatch;
        for (int e = tid; e < nbat * kCcpK; e += 1024) {
            const int dj = s_list[q0 + e / kCcpK], k = e % kCcpK;
            const int ix = (k < cand[dj].n) ? cand[dj].idx[k] : -1;
            s_bidx[e] = ix;
            s_btaken[e] = (ix >= 0) ? taken[ix] : (uint8_t)1;
        }
        if (tid < nbat) {
            const int dj = s_list[q0 + tid];
            s_bfull[tid] = (cand[dj].total <= cand[dj].n) ? 1 : 0;       // the list is complete
            s_bkill[tid] = A.kill[dj];
        }
        __syncthreads();
        if (wave == 0) {
            int done = 0, gone_n = 0;
            int mine = -1;                                                  // lane p: the p-th track that has gone in this batch
            for (; done < nbat; ++done) {
                const int ix = lane < kCcpK ? s_bidx[done * kCcpK + lane] : -1;
                const bool tk = lane < kCcpK ? s_btaken[done * kCcpK + lane] != 0 : true;
                unsigned long long fb = __ballot(ix >= 0 && !tk);           // free when the batch was fetched ...
                int pick = -1;
                while (fb) {                                                // ... and not gone since (at most kCcpK turns)
                    const int cand_k = (int)__builtin_ctzll(fb);
                    const int tr = __builtin_amdgcn_readlane(ix, cand_k);
                    if (!__ballot(mine == tr)) { pick = tr; break; }
                    fb &= fb - 1ull;
                }
                if (pick < 0 && !s_bfull[done]) break;                      // (wave-uniform: the scan of all tracks)
                const int gone = pick >= 0 ? pick : s_bkill[done];
                if (lane == 0) {
                    const int dj = s_list[q0 + done];
                    match[dj] = pick;
                    if (gone >= 0) taken[gone] = 1;
                    state[dj] = 1;
                }
                if (gone >= 0) { if (lane == gone_n) mine = gone; ++gone_n; }
            }
            if (lane == 0) s_n = done;                                      // (s_n: free since n_open was read)
        }
        __syncthreads();
        const int done = s_n;
        q0 += done;
        if (done == nbat) continue;
        // detection s_list[q0]: its kept list is a prefix and all of it has been taken
        const int d = s_list[q0];
        q0 += 1;
        const double px = A.det_pos[3 * d], py = A.det_pos[3 * d + 1], pz = A.det_pos[3 * d + 2], sp = A.det_speed[d];
        double best = __builtin_inf();
        int who = 0x7FFFFFFF;
        for (int t = tid; t < T; t += 1024) {
            const double upd = A.trk_upd[t];
            if (upd == A.now_s || taken[t]) continue;
            const double dx = A.trk_ref[3 * t] - px, dy = A.trk_ref[3 * t + 1] - py, dz = A.trk_ref[3 * t + 2] - pz;
            const double dist = sqrt(dot3(dx, dy, dz, dx, dy, dz));
            const double age = A.now_s - upd;
            double lo = sp * (age - A.slack_s), hi = sp * (age + A.slack_s);
            lo = (lo > 0.0) ? lo : 0.0; hi = (hi > 0.0) ? hi : 0.0;
            if (!(lo <= dist && dist <= hi)) continue;
            if (dist < best) { best = dist; who = t; }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int ow = __shfl_xor(who, off);
            if (ob < best || (ob == best && ow < who)) { best = ob; who = ow; }
        }
        if (lane == 0) { s_dist[wave] = best; s_trk[wave] = who; }
        __syncthreads();
        best = s_dist[0]; who = s_trk[0];
        for (int w = 1; w < 16; ++w)
            if (s_dist[w] < best || (s_dist[w] == best && s_trk[w] < who)) { best = s_dist[w]; who = s_trk[w]; }
        if (tid == 0) {
            const int32_t kx = A.kill[d];
            if (who != 0x7FFFFFFF) { match[d] = who; taken[who] = 1; }
            else { match[d] = -1; if (kx >= 0) taken[kx] = 1; }
            state[d] = 1;
        }
        __syncthreads();                             // (the track is gone before anybody scans for the next detection)
    }
    __syncthreads();                                 // (the list is everybody's until here)
  }
    if (tid == 0) counters[3] = 1;
}

// The step's launch decisions as the requests zrk_launch_salvo takes, in request order (= detection order, the order
// try_to_launch_missile was called in), without the host: detection d with out.launcher[d] >= 0 becomes {target row, the
// launcher's position, that launcher's missile parameters}.  One workgroup; entries [count, k_max) are filled with requests
// for no row at all (target_slot -1: they fail the solve and take the dead rows behind the successes), so that the salvo can
// be launched for k_max requests without anybody reading the count.
__global__ __launch_bounds__(1024) void k_ccp_requests(const zrk_ccp_out out, int64_t dmax, const double *__restrict__ lpos,
                                                       const double *__restrict__ params, zrk_launch_req *__restrict__ req,
                                                       int64_t k_max, int32_t *count_out)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int D = out.count[0];
    D = D < (int)dmax ? D : (int)dmax;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < D; base += 1024) {
        const int d = base + tid;
        const int l = d < D ? out.launcher[d] : -1;
        const unsigned long long b = __ballot(l >= 0);
        if (lane == 0) s_w[wave] = (int)__popcll(b);
        __syncthreads();
        int off = s_carry, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) off += s_w[w]; tot += s_w[w]; }
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int64_t q = off + (int)__popcll(b & below);
        if (l >= 0 && q < k_max) {
            zrk_launch_req r;
            r.target_slot = out.obj[d]; r._pad = 0;
            r.missile_pos[0] = lpos[3 * l]; r.missile_pos[1] = lpos[3 * l + 1]; r.missile_pos[2] = lpos[3 * l + 2];
            r.speed = params[3 * l]; r.period = params[3 * l + 1]; r.radius = params[3 * l + 2];
            req[q] = r;
        }
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    const int count = s_carry;
    if (tid == 0 && count_out) *count_out = count;
    for (int64_t q = count + tid; q < k_max; q += 1024) {
        zrk_launch_req r;
        r.target_slot = -1; r._pad = 0;
        r.missile_pos[0] = r.missile_pos[1] = r.missile_pos[2] = 0.0;
        r.speed = 0.0; r.period = 0.0; r.radius = 0.0;
        req[q] = r;
    }
}

// the rounds are over: not everybody resolved -> status 1 (cannot happen behind k_ccp_tail; kept as the check that it ran)
__global__ void k_ccp_finish(const int32_t *counters, int32_t *status)
{
    if (counters[3] == 0) atomicMax(status, 1);
}

// check_if_missiles_launched -> add_missile (modules/CCP.py:160-169, :102-108): a missile of our own enters the missile
// dictionary, keyed by its row (an existing key is replaced in place), updated "now".
__global__ void k_ccp_add_missile(const zrk_ccp_tracks trk, int32_t row, double now_s)
{
    __shared__ int s_hit;
    if (threadIdx.x == 0) s_hit = -1;
    __syncthreads();
    const int n = trk.counts[1];
    for (int k = threadIdx.x; k < n; k += blockDim.x) if (trk.tm_key[k] == row) atomicMax(&s_hit, k);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = s_hit;
        if (t < 0) {
            t = n;
            if (t >= trk.capacity) return;
            trk.tm_key[t] = row; trk.counts[1] = n + 1;
        }
        trk.tm_obj[t] = row; trk.tm_upd[t] = now_s;
    }
}

// ---------------------------------------------------------------------------------------------
// The battery's closed loop on the device (include/zrk_hot.h: zrk_battery): launchers, magazines and the two-tick way of a
// missile from the command post's request into the air, with the reference's latencies and orders.  Everything here is
// event-rate work on a few thousand requests at most: one workgroup per step, ordered by ballot scans.
// ---------------------------------------------------------------------------------------------
__global__ void k_battery_speed(const double *__restrict__ vel, int64_t cap, int64_t n, double *__restrict__ speed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double vx = vel[i], vy = vel[cap + i], vz = vel[2 * cap + i];
    speed[i] = sqrt(dot3(vx, vy, vz, vx, vy, vz));            // np.linalg.norm of a 3-vector (modules/AirObject.py:36)
}

__device__ __forceinline__ int battery_requests_of(const zrk_battery &B, int slot)
{
    const int n = B.sal_count[2 * slot];
    return n < B.k_max ? n : B.k_max;
}

// AirEnv takes the launched missiles in (modules/AirEnv.py:42-43): their rows have held their trajectories since the solve
__global__ void k_battery_activate(const zrk_battery B, int slot, uint8_t *alive, uint8_t *m_status, WaveBox *boxes)
{
    const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (q >= battery_requests_of(B, slot)) return;
    const int j = B.sal_air[(int64_t)slot * B.k_max + q];
    if (j < 0) return;
    const int64_t row = (int64_t)B.row0 + j;
    alive[row] = 1;
    m_status[j] = 1;
    // (a row block that held nobody is on record as empty for good: its record starts afresh)
    if (boxes) boxes[row / ZRK_BLOCK].state = 0u;
}

// MissileLauncher.step of every launcher (one thread each): last tick's cancelled missiles go back to the end of the list
// (MissileLauncher.py:126-129) -- before this tick's requests, whose messages were posted later (the command post steps after
// the launchers) --, then every request is served by the missile popped from the end (:58-80)
__global__ void k_battery_launchers(const zrk_battery B, int slot_back, int slot_serve)
{
    const int l = (int)threadIdx.x;
    if (l >= B.L) return;
    int32_t *stack = B.stack + (int64_t)l * B.n_missiles;
    int top = B.top[l];
    if (slot_back >= 0) {
        const int nb = battery_requests_of(B, slot_back);
        const int64_t o = (int64_t)slot_back * B.k_max;
        for (int q = 0; q < nb; ++q)
            if (B.sal_launcher[o + q] == l && B.sal_rc[o + q] != 0 && B.sal_missile[o + q] >= 0 && top < B.n_missiles) stack[top++] = B.sal_missile[o + q];
    }
    if (slot_serve >= 0) {
        const int ns = battery_requests_of(B, slot_serve);
        const int64_t o = (int64_t)slot_serve * B.k_max;
        for (int q = 0; q < ns; ++q)
            if (B.sal_launcher[o + q] == l) B.sal_missile[o + q] = top > 0 ? stack[--top] : -1;
    }
    B.top[l] = top;
}

// Missile._launch of every served request (modules/Missile.py:104-133): against the target's position as it stands now
__global__ void k_battery_solve(const zrk_battery B, int slot, const double *__restrict__ vel, const uint8_t *__restrict__ kind,
                                const double *__restrict__ pos, int64_t cap)
{
    const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (q >= battery_requests_of(B, slot)) return;
    const int64_t o = (int64_t)slot * B.k_max + q;
    const int m = B.sal_missile[o];
    zrk_launch_res res;
    res.rc = 7; res._pad = 0; res.velocity[0] = res.velocity[1] = res.velocity[2] = 0.0; res.t_hit = 0.0;    // 7: the launcher had no missile left
    if (m >= 0) {
        zrk_launch_req rq;
        rq.target_slot = B.sal_row[o]; rq._pad = 0;
        rq.missile_pos[0] = B.mi_pos[3 * m]; rq.missile_pos[1] = B.mi_pos[3 * m + 1]; rq.missile_pos[2] = B.mi_pos[3 * m + 2];
        rq.speed = B.mi_speed[m]; rq.period = B.mi_period[m]; rq.radius = B.mi_radius[m];
        res = launch_solve_one(vel, kind, pos, cap, rq);
    }
    B.sal_rc[o] = res.rc;
    B.sal_V[3 * o] = res.velocity[0]; B.sal_V[3 * o + 1] = res.velocity[1]; B.sal_V[3 * o + 2] = res.velocity[2];
}

// exclusive ranks of the set flags over [0, n), in order, by ONE workgroup of 1024 threads: rank (or -1) per thread and step
// through `visit(q, rank)`; returns the number of set flags
template <class Flag, class Visit>
__device__ __forceinline__ int ordered_ranks_1024(int n, int *s_w /* [16] */, int *s_carry, Flag flag, Visit visit)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int q = base + tid;
        const bool f = q < n && flag(q);
        const unsigned long long b = __ballot(f);
        if (lane == 0) s_w[wave] = (int)__popcll(b);
        __syncthreads();
        int off = *s_carry, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) off += s_w[w]; tot += s_w[w]; }
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        if (q < n) visit(q, f ? off + (int)__popcll(b & below) : -1);
        __syncthreads();
        if (tid == 0) *s_carry += tot;
        __syncthreads();
    }
    const int total = *s_carry;
    __syncthreads();                                  // (the next call clears the word)
    return total;
}

// The launches get their rows, in AirEnv's order: the requests stand launcher after launcher, each in the order it was served, and
// that is the order the launchers announce their missiles in a tick later (MissileLauncher.py:103-124 -> AirEnv.py:42-43)
__global__ __launch_bounds__(1024) void k_battery_assign(const zrk_battery B, int slot, int tick, double now_s, double *sp, double *vel,
                                                         double *t0, uint8_t *alive, uint8_t *kind, double *pos0, double *pos1, int64_t cap,
                                                         int32_t *m_slot, int32_t *m_tgt, double *m_radius, double *m_period, uint8_t *m_status)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int n = battery_requests_of(B, slot);
    const int64_t o = (int64_t)slot * B.k_max;
    const int base_air = *B.air_count, base_log = B.log_count[0];
    const int total = ordered_ranks_1024(n, s_w, &s_carry, [&](int q) { return B.sal_rc[o + q] == 0; }, [&](int q, int rank) {
        const int m = B.sal_missile[o + q];
        int j = rank >= 0 ? base_air + rank : -1;
        if (j >= B.n_missiles) j = -1;                       // (cannot happen: every launch is a missile of the magazine)
        B.sal_air[o + q] = j;
        if (j >= 0) {
            const int64_t row = (int64_t)B.row0 + j;
            for (int c = 0; c < 3; ++c) {
                sp[c * cap + row] = B.mi_pos[3 * m + c]; vel[c * cap + row] = B.sal_V[3 * (o + q) + c];
                pos0[c * cap + row] = B.mi_pos[3 * m + c]; pos1[c * cap + row] = B.mi_pos[3 * m + c];
            }
            t0[row] = now_s; alive[row] = 0; kind[row] = 1;
            m_slot[j] = (int32_t)row; m_tgt[j] = B.sal_row[o + q]; m_radius[j] = B.mi_radius[m]; m_period[j] = B.mi_period[m]; m_status[j] = 0;
            B.speed_mod[row] = B.mi_speed[m];               // Missile.speed_mod = velocity_module (modules/Missile.py:28)
            B.air_missile[j] = m;
        }
        const int64_t lg = (int64_t)base_log + q;
        if (lg < B.log_cap) {
            B.log_solve[5 * lg] = tick; B.log_solve[5 * lg + 1] = m; B.log_solve[5 * lg + 2] = B.sal_row[o + q];
            B.log_solve[5 * lg + 3] = B.sal_rc[o + q]; B.log_solve[5 * lg + 4] = j;
            for (int c = 0; c < 3; ++c) B.log_V[3 * lg + c] = B.sal_V[3 * (o + q) + c];
        }
    });
    if (threadIdx.x == 0) {
        *B.air_count = base_air + total < B.n_missiles ? base_air + total : B.n_missiles;
        B.sal_count[2 * slot + 1] = total;
        B.log_count[0] = base_log + n;
    }
}

// check_if_missiles_launched (modules/CCP.py:160-169): the launches of the salvo, in order, into the missile dictionary
__global__ __launch_bounds__(1024) void k_battery_announce(const zrk_battery B, int slot, const zrk_ccp_tracks trk, double now_s)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int n = battery_requests_of(B, slot);
    const int64_t o = (int64_t)slot * B.k_max;
    const int base = trk.counts[1];
    const int total = ordered_ranks_1024(n, s_w, &s_carry, [&](int q) { return B.sal_air[o + q] >= 0; }, [&](int q, int rank) {
        const int t = rank >= 0 ? base + rank : -1;
        if (t >= 0 && t < trk.capacity) {
            const int32_t row = B.row0 + B.sal_air[o + q];
            trk.tm_key[t] = row; trk.tm_obj[t] = row; trk.tm_upd[t] = now_s;
        }
    });
    if (threadIdx.x == 0) trk.counts[1] = base + total < (int)trk.capacity ? base + total : (int)trk.capacity;
}

// FoundObjectsMessage after FoundObjectsMessage, every object at its first mention (modules/CCP.py:406-417): an entry of radar r's
// list is a first mention when r is the lowest radar that saw the row
__global__ __launch_bounds__(1024) void k_battery_sequence(const int32_t *__restrict__ det_idx, int64_t det_stride, const int32_t *__restrict__ det_cnt,
                                                           int R, int32_t base_index, const uint32_t *__restrict__ vis,
                                                           const int32_t *__restrict__ row_of_list, int32_t *__restrict__ seq, int32_t *seq_count,
                                                           int64_t seq_cap)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    int run = 0;
    for (int r = 0; r < R; ++r) {
        const int cnt = det_cnt[r] < det_stride ? det_cnt[r] : (int)det_stride;
        const int32_t *list = det_idx + (int64_t)r * det_stride;
        const int run0 = run;
        // (four consecutive entries per thread, 4096 a turn: the entries' loads, then their masks', in flight together)
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        if (tid == 0) s_carry = 0;
        __syncthreads();
        for (int base = 0; base < cnt; base += 4096) {
            const int q0 = base + tid * 4;
            int32_t li[4];
            uint32_t m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) li[j] = list[q0 + j < cnt ? q0 + j : 0] - base_index;
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = vis[li[j]];
            bool f[4];
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            int mine = 0, tot_w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f[j] = q0 + j < cnt && (m[j] & (0u - m[j])) == (1u << r);          // first mentioned by radar r: its lowest set bit
                const unsigned long long bb = __ballot(f[j]);
                mine += (int)__popcll(bb & below); tot_w += (int)__popcll(bb);
            }
            if (lane == 0) s_w[wave] = tot_w;
            __syncthreads();
            int off = run0 + s_carry + mine, tot = 0;
            for (int w = 0; w < 16; ++w) { if (w < wave) off += s_w[w]; tot += s_w[w]; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (f[j]) { if (off < seq_cap) seq[off] = row_of_list ? row_of_list[li[j]] : li[j]; ++off; }
            __syncthreads();
            if (tid == 0) s_carry += tot;
            __syncthreads();
        }
        run += s_carry;
        __syncthreads();                                  // (the next radar clears the word)
    }
    if (threadIdx.x == 0) *seq_count = run < seq_cap ? run : (int32_t)seq_cap;
}

// The command post's requests of the tick as the salvo the launchers will serve: launcher after launcher (module order), each in
// request order.  The requesters are first squeezed, in order, into the slot's own (still unused) missile / air columns.
__global__ __launch_bounds__(1024) void k_battery_requests(const zrk_battery B, int slot, const zrk_ccp_out out, int64_t dmax)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    int D = out.count[0];
    D = D < (int)dmax ? D : (int)dmax;
    const int64_t o = (int64_t)slot * B.k_max;
    int32_t *tmp_obj = B.sal_missile + o, *tmp_l = B.sal_air + o;
    int asked = ordered_ranks_1024(D, s_w, &s_carry, [&](int d) { return out.launcher[d] >= 0; }, [&](int d, int rank) {
        if (rank >= 0 && rank < B.k_max) { tmp_obj[rank] = out.obj[d]; tmp_l[rank] = out.launcher[d]; }
    });
    asked = asked < B.k_max ? asked : B.k_max;
    __syncthreads();
    int run = 0;
    for (int l = 0; l < B.L; ++l) {
        const int run0 = run;
        run += ordered_ranks_1024(asked, s_w, &s_carry, [&](int q) { return tmp_l[q] == l; }, [&](int q, int rank) {
            if (rank >= 0) { B.sal_row[o + run0 + rank] = tmp_obj[q]; B.sal_launcher[o + run0 + rank] = l; }
        });
    }
    __syncthreads();
    for (int q = threadIdx.x; q < B.k_max; q += 1024) { B.sal_missile[o + q] = -1; B.sal_air[o + q] = -1; B.sal_rc[o + q] = (q < run) ? 7 : 6; }
    if (threadIdx.x == 0) { B.sal_count[2 * slot] = run; B.sal_count[2 * slot + 1] = 0; }
}

__global__ void k_battery_log_events(const zrk_battery B, const int32_t *__restrict__ ev_missile, const int32_t *__restrict__ ev_target,
                                     const int32_t *ev_count, int tick)
{
    const int n = *ev_count, base = B.log_count[1];
    for (int j = threadIdx.x; j < n; j += blockDim.x)
        if (base + j < B.log_cap) { B.log_event[3 * (int64_t)(base + j)] = tick; B.log_event[3 * (int64_t)(base + j) + 1] = ev_missile[j]; B.log_event[3 * (int64_t)(base + j) + 2] = ev_target[j]; }
    __syncthreads();
    if (threadIdx.x == 0) B.log_count[1] = base + n;
}

__global__ void k_selftest_math(int op, const double *a, const double *b, double *y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (op) {
    case 0: r = sqrt(a[i]); break;
    case 1: r = a[i] / b[i]; break;
    case 2: r = atan2(a[i], b[i]); break;
    case 3: r = asin(a[i]); break;
    default: r = sqrt(dot3(a[i], b[i], 0.0, a[i], b[i], 0.0)); break;
    }
    y[i] = r;
}

__global__ void k_selftest_noise(uint64_t seed, uint64_t tick, uint32_t ordinal, int64_t entity0, double *out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    NoiseState ns = noise_init(seed, tick, (uint64_t)(entity0 + i));
    float nz[3] = {0.f, 0.f, 0.f};
    for (uint32_t k = 0; k <= ordinal; ++k) noise_draw3(ns, nz);
    out[3 * i] = (double)nz[0]; out[3 * i + 1] = (double)nz[1]; out[3 * i + 2] = (double)nz[2];
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
double d2_threshold(double m)
{
    if (std::isnan(m)) return INFINITY;            // `dist > nan` is never true
    if (m < 0.0) return -1.0;                      // every dist >= 0 > m
    if (std::isinf(m)) return INFINITY;
    double c = m * m;
    if (std::isinf(c)) c = DBL_MAX;
    while (std::sqrt(c) > m) c = std::nextafter(c, 0.0);
    for (;;) {
        double nx = std::nextafter(c, INFINITY);
        if (std::isinf(nx) || std::sqrt(nx) > m) break;
        c = nx;
    }
    return c;
}


}  // namespace

// Diagnostics (ZRK_TRACE=1): host time stamps of one zrk_run_ticks call -- the calling thread's and the side stream's
// thread's -- printed to stderr when the call returns (=2: when the next call starts, so that the printing -- 40 to 100 us
// for a 20-tick call -- is not part of what a caller times around the call).  Off: one predictable branch per stamp.
namespace {
struct HostTrace {
    bool on = false, deferred = false;     // deferred (ZRK_TRACE=2): printed when the NEXT call starts, not inside the call it describes
    std::mutex mu;
    std::vector<std::pair<const char *, int64_t>> marks;
    static int64_t now() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void mark(const char *what) { if (on) { std::lock_guard<std::mutex> lk(mu); marks.emplace_back(what, now()); } }
    void dump()
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(mu);
        if (marks.empty()) return;
        std::stable_sort(marks.begin(), marks.end(), [](const auto &a, const auto &b) { return a.second < b.second; });
        const int64_t t0 = marks.front().second;
        for (const auto &m : marks) std::fprintf(stderr, "[zrk trace] %10.1f us  %s\n", (double)(m.second - t0) / 1e3, m.first);
        marks.clear();
    }
};
HostTrace g_trace;
}  // namespace

// Every host-side wait of this library is bounded (a dead peer rank, a helper thread that failed, a device that does
// not come back must end in ZRK_E_STATE, not in a process that spins for ever): ZRK_HOST_WAIT_MS, default 30 s.
namespace {
std::atomic<int> g_host_wait_ms{-1};              // read at first use and by zrk_ctx_reload_env
void host_wait_reload()
{
    const char *v = std::getenv("ZRK_HOST_WAIT_MS");
    g_host_wait_ms.store(v ? std::max(1, std::atoi(v)) : 30000);
}
std::chrono::milliseconds host_wait_limit()
{
    if (g_host_wait_ms.load(std::memory_order_relaxed) < 0) host_wait_reload();
    return std::chrono::milliseconds(g_host_wait_ms.load(std::memory_order_relaxed));
}

// Host threads this process may really use: the affinity mask, capped by the cgroup's CPU quota (what sizes the number of
// helper threads a rank starts: three busy threads per rank on an 8-rank node want 24 cores).
int usable_host_cores()
{
    int cores = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
            cores = std::min<long long>(cores, std::max<long long>(1, std::atoll(quota) / period));
        std::fclose(f);
    }
    return std::max(1, cores);
}

// Spins (pause) until ok() holds; false when the limit ran out first.  The clock is read every 4096 looks.
// ZRK_STALL_US=n (diagnostics): a host-side wait of the library, or a launch call of the loop, that takes longer than n
// microseconds is reported on stderr with the source line it stands in.
inline int stall_us()
{
    static const int us = [] { const char *v = std::getenv("ZRK_STALL_US"); return v ? std::atoi(v) : 0; }();
    return us;
}
inline void stall_report(std::chrono::steady_clock::time_point t0, int line, const char *what)
{
    const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    if (us > stall_us()) std::fprintf(stderr, "[zrk stall] %lld us in %s (zrk_hot.hip:%d)\n", (long long)us, what, line);
}

template <class Pred>
bool spin_until(Pred ok, std::chrono::milliseconds limit = host_wait_limit(), int line = __builtin_LINE())
{
    if (ok()) return true;
    const auto t0 = std::chrono::steady_clock::now();
    bool yielding = false;
    for (unsigned spins = 1;; ++spins) {
        // (a wait that has lasted 50 us may be for a thread that wants this very core: give it up between looks)
        if (yielding) sched_yield(); else __builtin_ia32_pause();
        if (ok()) { if (stall_us() > 0) stall_report(t0, line, "a host-side wait"); return true; }
        if ((spins & (yielding ? 0x3Fu : 0x3FFu)) == 0) {
            const auto waited = std::chrono::steady_clock::now() - t0;
            if (waited > limit) return false;
            if (waited > std::chrono::microseconds(50)) yielding = true;
        }
    }
}
}  // namespace

// Overlap mode of zrk_run_ticks: the side stream on which a tick's lists are compacted beside the next tick's sweep,
// and the thread that issues its work (waiting for a flag word, the compaction, in an exchange the collective behind it,
// an event) so that the calling thread is left with the two launches of the compute stream.
struct zrk_exchange;
struct SideItem {
    hipStream_t stream;
    uint32_t flag_value;            // flag[0] >= this: the tick's two launches on the compute stream are over (raised by
                                    // the first thread of the next sweep, or by a launch of its own behind the last tick)
    CompactArgs C;
    int by_ticket;
    MissileArgs M;                  // apply == 0: the ordered event list only (and the events in the list's tail)
    // an exchange's collective runs on the exchange's own stream and waits for *raise >= raise_value, which a launch of
    // its own behind the compaction writes
    uint32_t *raise;
    uint32_t raise_value;
    int done_slot;                  // the ring slot whose buffers this compaction reads (done[done_slot] is recorded last)
    // the item's number (set by side_enqueue): its compaction's first thread writes done_value - 1 to the side stream's pinned
    // word (DoneWord: the launch before it is over); record_event: an event behind the launch (the call's last items, which
    // no launch follows: the compute stream takes the side stream in through it)
    uint32_t done_value;
    int record_event;
    // one helper thread per rank (few host cores per rank): this thread also issues the tick's collective, right behind
    // the launch that raises the word it waits for (otherwise the exchange's own thread does)
    // the LAST tick of a call has no next sweep to raise the word: its compaction waits, on the device, for an event recorded
    // behind that sweep (an event record costs the compute stream a barrier packet -- harmless where no sweep follows)
    hipEvent_t wait_event;
    // both ticks of a pair launch in one compaction launch (k_compact_pair): C / M are the first tick's, C2 / M2 the second's
    int pair, done_slot2, pair_threads;
    // the LAST compaction of a call has no sweep to hide behind: it goes to the COMPUTE stream (`stream` is then the caller's),
    // right behind the last sweep -- in order on one queue, instead of a word raised behind that sweep, seen by this thread and
    // answered with a launch on the other queue (15 us of a 20-tick call).  The side stream's compaction before it is taken in
    // through an event (Side::tail_ev)
    int on_compute;
    CompactArgs C2;
    MissileArgs M2;
    const int32_t *rm;
    int rm_cap;
    MarksArgs marks;                // blocks > 0: the call's removal marks are carried out by this (pair) launch
    zrk_exchange *post_x;
    int post_slot;
    const int64_t *post_send;
    int64_t *post_recv;
    int64_t post_words;
    // (a pair's second tick: its collective behind the same compaction launch)
    int post2_slot;
    const int64_t *post2_send;
    int64_t *post2_recv;
};

struct Side {
    hipStream_t stream = nullptr;
    // the word the compute stream raises (SideItem::flag_value) lives in pinned HOST memory: the side stream's thread
    // polls it there and launches the compaction when it is up -- a one-lane wait kernel in front of every compaction
    // cost the side stream 5 us a tick (a lone wave is slow to find a slot on a device full of sweep waves)
    std::chrono::steady_clock::time_point last_launch{};   // of the side stream's thread (side_issue)
    volatile uint32_t *hflag = nullptr;
    uint32_t *hflag_dev = nullptr;  // the same word as the device addresses it
    // ... and the word in which a compaction says that the one before it is over (item numbers, in launch order), 64 bytes on
    volatile uint32_t *hdone = nullptr;
    uint32_t *hdone_dev = nullptr;
    uint32_t seq = 0;
    // mask buffers of its own for all ticks of a call but the last (whose masks the caller may read): each is all zero
    // except between the sweep that writes it and the compaction that reads and clears it; slot kMasks stands for the
    // caller's buffer of the last tick
    static constexpr int kMasks = 8;
    uint32_t *masks[kMasks] = {nullptr, nullptr, nullptr};
    int64_t mask_rows = 0;
    // ... and, with the same lifetimes, the per-row event codes of the missile phase (written by a tick's sweep, read by
    // its tombstones on the compute stream and by its event list on the side stream, beside the next sweep)
    uint8_t *codes[kMasks] = {nullptr, nullptr, nullptr};
    int64_t code_rows = 0;
    uint8_t *pend = nullptr;        // removal marks of the running call (SweepParams::pend), all zero between calls
    int64_t pend_rows = 0;
    bool masks_dirty = false;       // a call failed half-way: clear them before the next use
    uint64_t mask_pos = 0;
    hipEvent_t done[kMasks + 1] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done_of[kMasks + 1] = {nullptr};   // the event that says a slot's compaction is over (a pair item records ONE for its two)
    hipEvent_t last_sweep = nullptr;   // recorded behind the last sweep of a call (SideItem::wait_event)
    hipEvent_t tail_ev = nullptr;      // recorded behind the side stream's last launch when an item goes to the compute stream (SideItem::on_compute)
    bool side_busy = false;            // the side stream has launches the compute stream has not taken in (the thread's; the caller's between items)
    // on_compute == 2 (a pair's ONE compaction launch as the call's last): that launch has control words and records of its
    // own (tail_ws) and nothing of the caller's is written by the launch before it, so it waits for nobody -- the event above
    // cost 11 us between the last sweep and it.  The side stream's launch before it then stands alone: an event behind it
    // (tail_ev, asked only by the host, when its mask buffers come up for reuse), and the tail itself is known by number
    void *tail_ws = nullptr;
    uint64_t tail_item = 0;            // the last call's last item, issued on tail_stream
    hipStream_t tail_stream = nullptr;
    int last_side_slot[2] = {-1, -1};  // ring slots of the side stream's last launch (the thread's)
    bool tail_ev_live = false;         // tail_ev stands behind the side stream's last launch of the running call (the thread's)
    uint32_t *bar = nullptr;           // DEVICE: the barrier of a pair launch's missile workgroups (arrivals, gave-up word)
    uint32_t bar_epoch = 0;            // arrivals asked for so far
    // k_compact_pair: per ring slot the list of rows a pair's first tick removed (MissileArgs::rm), and where the FIRST tick's
    // lists, union list and ordered events go (the second tick's go where the caller reads)
    int32_t *rm[kMasks + 1] = {nullptr};
    int rm_cap = 0;
    int32_t *scratch_det = nullptr, *scratch_cnt = nullptr, *scratch_ev = nullptr;
    int64_t *scratch_packed = nullptr;
    int64_t scratch_det_ints = 0, scratch_packed_words = 0, scratch_ev_rows = 0;
    int cu_count = 0;                  // ZRK_SIDE_CUS: the side stream is confined to this many compute units (0: all)
    bool posted[kMasks + 1] = {false, false, false, false};
    uint64_t item_no[kMasks + 1] = {0, 0, 0, 0};
    uint64_t joined_upto = 0;       // items up to this number belong to calls whose side work joined_stream has taken in
    hipStream_t joined_stream = nullptr;
    static constexpr uint64_t kRing = 8;
    SideItem ring[kRing];
    std::atomic<uint64_t> head{0}, tail{0};
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<bool> asleep{false}, stop{false};
    std::atomic<int> rc{0};
    std::string err;                // written by the thread before rc, read after
};

struct zrk_ctx {
    int device;
    int cus = 256;                     // compute units of the device
    int fused_max_blocks = 0;          // single-launch compaction up to this many workgroups (0: never)
    uint32_t epoch = 0;                // tag of the next single-launch compaction
    const void *fused_ws = nullptr;    // workspace whose control words this context has cleared
    const void *order_ws = nullptr;    // workspace holding a sweep order built by the last tick of zrk_run_ticks ...
    int order_nb = 0;                  // ... for this many row blocks
    bool order_ready = false;
    int order_phase = 0;               // which of the workspace's two order lists the next sweep reads
    bool order_enabled = true;
    int env_items = 0;                 // ZRK_COMPACT_ITEMS (0: automatic), read once: getenv per launch costs microseconds
    int env_order = -1;                // ZRK_COMPACT_ORDER: 0 "block", 1 anything else, -1 automatic
    int env_group = -1;                // ZRK_COMPACT_GROUP: 0 flat sums, a power of two <= 32 the group size, -1 automatic
    uint32_t diag = 0;                 // ZRK_DIAG: bit 0 no "certainly visible" shortcut, bit 1 no box records
    bool time_on_dispatch = true;      // ZRK_TIME_BY_RECORDS=1: time sweeps between two recorded events instead
    const void *box_ws = nullptr;      // workspace whose box records belong to ...
    const void *box_key = nullptr;     // ... this table (its start_pos column) ...
    int64_t box_n = 0;                 // ... up to this many rows
    std::string err;
    // d2_threshold(max_distance) per radar slot: static for a radar, not worth a search every tick
    uint64_t d2_key[ZRK_MAX_RADARS] = {};
    double d2_val[ZRK_MAX_RADARS] = {};
    bool d2_set[ZRK_MAX_RADARS] = {};
    double d2_of(double max_distance, int r)
    {
        uint64_t key;
        std::memcpy(&key, &max_distance, sizeof(key));
        if (!d2_set[r] || d2_key[r] != key) { d2_key[r] = key; d2_val[r] = d2_threshold(max_distance); d2_set[r] = true; }
        return d2_val[r];
    }
    std::vector<hipEvent_t> tev;       // timing events of zrk_run_ticks, reused
    int tev_pending = 0;               // event pairs recorded by a deferred-profile call, not read yet
    const void *ring_key = nullptr;    // mask buffers zrk_run_ticks has been alternating between ...
    bool ring_clean[2] = {false, false};   // ... and whether each is all zero when its next sweep starts
    Side *side = nullptr;              // overlap mode, created on first use
    int overlap = 1;                   // ZRK_OVERLAP: 0 never, 1 (default) for calls of at least overlap_min ticks
    int overlap_min = 4;
    int64_t overlap_min_rows = 50000;  // ZRK_OVERLAP_MIN_ROWS: below, the side stream's machinery costs more than it hides
    int last_overlapped = 0;           // whether the last zrk_run_ticks* call ran overlapped
    // gather records of the table's rows for the missile phase (MissileArgs::grec), kept like the box records: for
    // this table (its start_pos column), up to this many rows; rebuilt when the key changes, when rows were rewritten
    // under the loop (zrk_ctx_invalidate_boxes) and for rows appended since
    double *grec = nullptr;
    int64_t grec_rows = 0, grec_n = 0;
    const void *grec_key = nullptr;
    bool grec_enabled = true;          // ZRK_GATHER_RECORDS=0: the columns
    struct RadarBlock *rb_cache = nullptr;   // the next launch's radar records (two ticks' worth), derived ahead (fill_radar_block)
    zrk_radar rb_cache_radars[2][ZRK_MAX_RADARS];
    int rb_cache_R[2] = {-1, -1};
    uint32_t rb_cache_flags[2] = {0, 0};
    bool pair_enabled = true;          // ZRK_PAIR=0: one tick per launch in the overlapped loop
    bool pair_compact = true;          // ZRK_PAIR_COMPACT=0: a pair's two compactions as two launches
    int pair_compact_blocks = 1024;    // ... beyond this many workgroups too (ZRK_PAIR_COMPACT_BLOCKS, <= kFusedMaxBlocks)
    int pair_threads = 0;              // ZRK_PAIR_THREADS=256|512|1024: workgroup size of k_compact_pair (0: by the number of workgroups)
    int last_ticks_per_launch = 1;     // of the last zrk_run_ticks* call
    std::vector<int> tev_alias, tev_ticks;   // per timing sample: which event pair holds it, and the ticks its launch swept
    // zrk_sweep_stamps: the sweeps of zrk_run_ticks* time themselves (SweepParams::stamps): a ring of kStampSlots launches'
    // worth of stamps, the sampled launches of the last call in its first stamp_used slots
    bool stamps_on = false;
    unsigned long long *stamp_ring = nullptr, *stamp_out = nullptr;      // DEVICE
    int64_t stamp_slot_words = 0;
    int stamp_used = 0;
    std::vector<int> stamp_ticks;
    std::vector<int64_t> stamp_waves;
    std::vector<unsigned long long> stamp_host;   // what the last zrk_read_sweep_stamps read: (first wave in, last wave out) per sampled launch
    bool tail_by_event = false;        // ZRK_TAIL_EVENT=1: the last compaction of a call is released by an event recorded behind the last sweep, not by
                                       // a launch that raises the host word (medians equal, 24.6 / 24.7 us per tick in 20-step runs; the event has the worse tail)
    bool tail_free = true;             // ... without waiting for the side stream's launch before it (Side::tail_ws; ZRK_TAIL_FREE=0: an event)
    double *frozen_prev = nullptr;     // zrk_ctx_keep_prev: where rows that leave the air keep their handle's prev_pos (plain loop)
    bool marks_in_tail = true;         // ZRK_MARKS_IN_TAIL=0: a launch of its own (k_apply_marks) between the call's last sweep and its last compaction
    bool tail_on_compute = true;       // the last compaction of a call goes to the compute stream, behind the last sweep (SideItem::on_compute);
                                       // ZRK_TAIL_COMPUTE=0: to the side stream like the others, released as above.  Calls with an exchange: always the latter
};

namespace {

int fail(zrk_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

int check_launch(zrk_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, ZRK_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return 0;
}

inline int nblocks(int64_t n, int per) { return (int)((n + per - 1) / per); }

struct Workspace {
    int32_t *ctl;                  // single-launch compaction: ticket, done, error
    unsigned long long *agg;       // ... and its per-workgroup records
    int32_t *order[2];             // per sweep row block: dispatch order, this tick's and the one being built
    uint32_t *order_ctr;           // two sets of counters (kOrderCtrSet words each), alternating
    WaveBox *boxes;                // per sweep wave: box record
    int32_t *counts, *offs, *totals;
};

constexpr int64_t kFusedBytes = kFusedCtlInts * (int64_t)sizeof(int32_t) +
                                ((int64_t)kFusedMaxBlocks * kPairAggStride + (int64_t)(kFusedMaxBlocks / kMinGroup) * kPairAggStride) *
                                    (int64_t)sizeof(unsigned long long);

inline int64_t order_ints(int64_t n) { return ((n + ZRK_BLOCK - 1) / ZRK_BLOCK + 64) & ~(int64_t)63; }
inline int64_t box_ints(int64_t n) { return ((((n + ZRK_BLOCK - 1) / ZRK_BLOCK + 4) * (int64_t)(sizeof(WaveBox) / 4)) + 63) & ~(int64_t)63; }

// Layout: [single-launch control words and records][totals][three-launch counts, offsets: sized for n_max]
// [cost][order][box records].  The first three parts start at fixed offsets and are all a stand-alone compaction
// touches (its n may be anything up to n_max); the loop's arrays behind them are found from the table's capacity,
// which zrk_run_ticks knows and which must be the n_max the workspace was sized for.
inline int64_t comp_blocks(int64_t n_max) { return (n_max + kCompBlock - 1) / kCompBlock + 1; }

Workspace carve(void *ws, int nb, int64_t n_max)
{
    Workspace w;
    w.ctl = (int32_t *)ws;
    w.agg = (unsigned long long *)(w.ctl + kFusedCtlInts);
    w.totals = (int32_t *)((char *)ws + kFusedBytes);   // [ZRK_MAX_RADARS + 1] (+ pad to 64 ints)
    w.counts = w.totals + 64;
    w.offs = w.counts + (int64_t)(ZRK_MAX_RADARS + 1) * nb;
    w.order[0] = w.counts + ((2 * (int64_t)(ZRK_MAX_RADARS + 1) * comp_blocks(n_max) + 63) & ~(int64_t)63);
    w.order[1] = w.order[0] + order_ints(n_max);
    w.order_ctr = (uint32_t *)(w.order[1] + order_ints(n_max));
    w.boxes = (WaveBox *)((char *)(w.order_ctr + 2 * kOrderCtrSet) + 2 * sizeof(RadarBlock));   // (two RadarBlocks in front)
    return w;
}

}  // namespace

#ifdef ZRK_PROBE_BUILD
ZRK_API int zrk_debug_wave_probe(long long *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wave_probe), &buf, sizeof(buf)) == hipSuccess ? 0 : ZRK_E_HIP;
}
#endif

ZRK_API int zrk_abi_version(void) { return ZRK_ABI_VERSION; }

ZRK_API void zrk_ctx_invalidate_boxes(zrk_ctx *ctx)
{
    if (ctx) { ctx->box_ws = nullptr; ctx->box_n = 0; ctx->grec_n = 0; }
}

ZRK_API void zrk_ctx_reload_env(zrk_ctx *c)
{
    if (!c) return;
    host_wait_reload();
    { const char *v = std::getenv("ZRK_TRACE"); g_trace.on = v && (v[0] == '1' || v[0] == '2'); g_trace.deferred = v && v[0] == '2'; }
    c->order_enabled = true; c->diag = 0; c->env_items = 0; c->env_order = -1;
    if (const char *v = std::getenv("ZRK_SWEEP_ORDER")) c->order_enabled = std::atoi(v) != 0;
    if (const char *v = std::getenv("ZRK_DIAG")) c->diag = (uint32_t)std::strtoul(v, nullptr, 0);
    { const char *v = std::getenv("ZRK_TIME_BY_RECORDS"); c->time_on_dispatch = !(v && v[0] == '1'); }
    { const char *v = std::getenv("ZRK_OVERLAP"); c->overlap = v ? std::atoi(v) : 1; }
    { const char *v = std::getenv("ZRK_OVERLAP_MIN"); c->overlap_min = v ? std::max(2, std::atoi(v)) : 4; }
    { const char *v = std::getenv("ZRK_OVERLAP_MIN_ROWS"); c->overlap_min_rows = v ? std::atoll(v) : 50000; }
    { const char *v = std::getenv("ZRK_GATHER_RECORDS"); c->grec_enabled = !(v && v[0] == '0'); }
    { const char *v = std::getenv("ZRK_TAIL_EVENT"); c->tail_by_event = v && v[0] == '1'; }
    { const char *v = std::getenv("ZRK_TAIL_FREE"); c->tail_free = !(v && v[0] == '0'); }
    { const char *v = std::getenv("ZRK_MARKS_IN_TAIL"); c->marks_in_tail = !(v && v[0] == '0'); }
    { const char *v = std::getenv("ZRK_TAIL_COMPUTE"); c->tail_on_compute = !(v && v[0] == '0') && !c->tail_by_event; }
    { const char *v = std::getenv("ZRK_PAIR"); c->pair_enabled = !(v && v[0] == '0'); }
    { const char *v = std::getenv("ZRK_PAIR_COMPACT"); c->pair_compact = !(v && v[0] == '0'); }
    if (const char *v = std::getenv("ZRK_PAIR_COMPACT_BLOCKS"))
        c->pair_compact_blocks = (int)std::min<long>(std::max<long>(std::strtol(v, nullptr, 10), 0), kFusedMaxBlocks);
    { const char *v = std::getenv("ZRK_PAIR_THREADS"); const int t = v ? std::atoi(v) : 0; c->pair_threads = (t == 256 || t == 512 || t == 1024) ? t : 0; }
    if (const char *v = std::getenv("ZRK_COMPACT_ITEMS")) c->env_items = std::max(1, std::atoi(v));
    if (const char *v = std::getenv("ZRK_COMPACT_ORDER")) c->env_order = std::strcmp(v, "block") != 0;
    if (const char *v = std::getenv("ZRK_COMPACT_GROUP")) {
        const long k = std::strtol(v, nullptr, 10);
        c->env_group = (k == 0 || k == kMinGroup || k == 8 || k == 16 || k == 32) ? (int)k : -1;
    }
    c->fused_max_blocks = kFusedMaxBlocks;
    if (const char *v = std::getenv("ZRK_COMPACT_FUSED_MAX_BLOCKS")) {     // 0 = always the three-launch path
        const long k = std::strtol(v, nullptr, 10);
        c->fused_max_blocks = (int)std::min<long>(std::max<long>(k, 0), kFusedMaxBlocks);
    }
}

ZRK_API int zrk_ctx_create(int device, zrk_ctx **out)
{
    if (!out) return ZRK_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return ZRK_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return ZRK_E_HIP;
    zrk_ctx *c = new zrk_ctx;
    c->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = cus;
    zrk_ctx_reload_env(c);
    *out = c;
    return 0;
}

namespace { void side_destroy(Side *sd); }

ZRK_API void zrk_ctx_destroy(zrk_ctx *ctx)
{
    if (!ctx) return;
    side_destroy(ctx->side);
    std::free(ctx->rb_cache);
    if (ctx->grec) (void)hipFree(ctx->grec);
    if (ctx->stamp_ring) (void)hipFree(ctx->stamp_ring);
    if (ctx->stamp_out) (void)hipFree(ctx->stamp_out);
    for (hipEvent_t e : ctx->tev) (void)hipEventDestroy(e);
    delete ctx;
}

ZRK_API const char *zrk_last_error(zrk_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

ZRK_API int64_t zrk_workspace_bytes(int64_t n_max)
{
    if (n_max < 0) return ZRK_E_INVALID;
    return kFusedBytes + 2 * (int64_t)sizeof(RadarBlock) +
           (2 * order_ints(n_max) + 2 * kOrderCtrSet + box_ints(n_max) + 64 + 64 + 2 * (int64_t)(ZRK_MAX_RADARS + 1) * comp_blocks(n_max)) *
               (int64_t)sizeof(int32_t);
}

namespace {

MissileArgs no_missiles()
{
    MissileArgs M;
    std::memset(&M, 0, sizeof(M));
    return M;
}

MissileArgs missile_args(const zrk_entities *e, int cur, const zrk_missiles *mis, int64_t m, int64_t time_ms,
                         int64_t dt_ms, int apply)
{
    MissileArgs M;
    M.sp = e->start_pos; M.vel = e->velocity; M.t0 = e->start_time; M.alive = e->alive; M.lidx = e->list_index;
    M.pos_cur = e->pos[cur]; M.pos_prev = e->pos[cur ^ 1]; M.cap = e->capacity;
    M.m_slot = mis->slot; M.m_tgt = mis->target; M.m_radius = mis->radius; M.m_period = mis->period;
    M.m_status = mis->status; M.ev_code = mis->ev_code;
    M.ev_missile = mis->ev_missile; M.ev_target = mis->ev_target; M.ev_count = mis->ev_count;
    M.m = m;
    M.t = (double)time_ms / 1000.0; M.dts = (double)dt_ms / 1000.0;
    M.apply = apply; M.ev_wire_cap = 0; M.ev_wire = nullptr; M.gid0 = 0;
    M.pend = nullptr; M.mark = 0; M._pad3 = 0; M.grec = nullptr;
    M.pos_abs[0] = e->pos[0]; M.pos_abs[1] = e->pos[1];
    M.ev_code2 = nullptr; M.mark2 = 0; M.bar_target = 0; M.bar = nullptr; M.t2 = M.t; M.clear_vis = nullptr;
    M.rm = nullptr; M.rm_cap = 0; M._pad4 = 0;
    M.frozen_prev = nullptr;
    return M;
}

// The records of one scenario's radars, derived on the host from their current angles.  Sixteen radars cost ~6 us of
// trigonometry; the loop derives the NEXT tick's block right behind a launch (radar_block_ahead), so that the launch of
// a tick -- the first one of a call above all, with the device idle -- finds it ready (keyed by the radars' bytes).
void fill_radar_block(zrk_ctx *ctx, const zrk_radar *radars, int R, uint32_t flags, RadarBlock &rb)
{
    const uint32_t key_flags = flags & (ZRK_F_EXACT_ONLY | ZRK_F_PHILOX);
    for (int k = 0; k < 2 && ctx->rb_cache && R > 0; ++k) {
        if (ctx->rb_cache_R[k] == R && ctx->rb_cache_flags[k] == key_flags &&
            std::memcmp(ctx->rb_cache_radars[k], radars, sizeof(zrk_radar) * (size_t)R) == 0) {
            std::memcpy(&rb, ctx->rb_cache + k, sizeof(rb));
            return;
        }
    }
    std::memset(&rb, 0, sizeof(rb));
    TrigMemo memo;
    for (int r = 0; r < ZRK_MAX_RADARS; ++r) {
        RadarPre pre;
        std::memset(&pre, 0, sizeof(pre));
        pre.d2_out = -1.f;                                  // beyond R: nobody is a candidate
        if (r < R) {
            RadarHot hot;
            derive_radar(radars[r], ctx->d2_of(radars[r].max_distance, r), (flags & ZRK_F_EXACT_ONLY) != 0, hot, rb.cold[r], &memo);
            std::memcpy(rb.hotw[r], &hot, sizeof(hot));
            derive_pre(radars[r], hot, (flags & ZRK_F_PHILOX) != 0, r, pre);
        }
        std::memcpy(rb.prew[r], &pre, sizeof(pre));
    }
}

void radar_block_ahead(zrk_ctx *ctx, const zrk_radar *radars, int R, uint32_t flags, int k = 0)
{
    if (R <= 0 || !radars) return;
    if (!ctx->rb_cache) ctx->rb_cache = (RadarBlock *)std::malloc(2 * sizeof(RadarBlock));
    if (!ctx->rb_cache) return;
    ctx->rb_cache_R[k] = -1;                              // (derive afresh: the two slots hold consecutive ticks, never the same)
    RadarBlock *dst = ctx->rb_cache + k;
    fill_radar_block(ctx, radars, R, flags, *dst);
    std::memcpy(ctx->rb_cache_radars[k], radars, sizeof(zrk_radar) * (size_t)R);
    ctx->rb_cache_R[k] = R; ctx->rb_cache_flags[k] = flags & (ZRK_F_EXACT_ONLY | ZRK_F_PHILOX);
}

// What a batched ensemble adds to the two launches of a tick.
struct EnsLaunch {
    const char *rb_table;          // this tick's records, [S] RadarBlock
    const uint64_t *seeds;
    int64_t rows_ps;
    int32_t bps, items;            // sweep row blocks / compaction workgroups... per scenario: rows_ps / (1024 * items)
    EnsembleArgs next;             // what the compaction's extra workgroups derive for the next tick
};

// A PAIR launch (SweepParams::t2): what the second tick adds.
struct PairLaunch {
    int64_t time2_ms;
    const zrk_radar *radars2;      // the radars as they stand in the second tick
    uint32_t *vis2;
    uint32_t mark2;
};

int launch_sweep(zrk_ctx *ctx, const zrk_entities *e, int64_t n, int cur, int64_t time_ms, const zrk_radar *radars,
                 int R, uint32_t flags, uint64_t seed, uint64_t tick, int64_t gid0, void *workspace, void *stream,
                 const MissileArgs &M, uint32_t *vis = nullptr, int32_t *order_next = nullptr, const int32_t *order = nullptr,
                 WaveBox *boxes = nullptr, const EnsLaunch *ens = nullptr, const RadarBlock *rb_device = nullptr,
                 uint32_t *flag = nullptr, uint32_t flag_value = 0, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                 uint32_t *order_ctr = nullptr, uint32_t *order_ctr_next = nullptr, uint8_t *pend = nullptr, uint32_t mark = 0,
                 const PairLaunch *pair = nullptr, unsigned long long *stamps = nullptr, uint32_t *vis_clear = nullptr)
{
    if (!ctx || !e || !workspace || (R > 0 && !radars && !ens && !rb_device)) return fail(ctx, ZRK_E_INVALID, "zrk_tick_sweep: null argument");
    if (R < 0 || R > ZRK_MAX_RADARS)
        return fail(ctx, ZRK_E_INVALID, "zrk_tick_sweep: radar count out of range");
    if (n < 0 || n > e->capacity || (cur != 0 && cur != 1)) return fail(ctx, ZRK_E_INVALID, "zrk_tick_sweep: n/cur out of range");
    if (n == 0 && M.m == 0) return 0;
    static SweepParamsPair PP;                              // (20 KB: not on the stack; one host thread per context -- and the
                                                            // launch copies it before it returns)
    static std::mutex pp_mu;
    std::lock_guard<std::mutex> pp_lock(pp_mu);             // (contexts on different threads share the buffer)
    SweepParams &P = PP;
    P.sp = e->start_pos; P.vel = e->velocity; P.t0 = e->start_time; P.alive = e->alive; P.lidx = e->list_index;
    P.pos = e->pos[cur]; P.vis = vis ? vis : e->vis_mask;
    P.order = order; P.order_next = order_next; P.order_ctr = order_ctr; P.order_ctr_next = order_ctr_next;
    P.vis_clear = vis_clear;
    P.n = n; P.cap = e->capacity;
    P.t = (double)time_ms / 1000.0;                 // to_seconds, modules/AirObject.py:5-7
    P.seed = seed; P.tick = tick; P.gid0 = gid0;
    P.R = R; P.nb = nblocks(n, ZRK_BLOCK); P.flags = flags;
    P.mb = nblocks(M.m, ZRK_BLOCK); P.flag = flag; P.flag_value = flag_value;
    P.boxes = boxes; P.stamps = stamps;
    P.pend = pend; P.mark = mark; P.mark2 = pair ? pair->mark2 : mark; P.alive_w = e->alive; P.pos_prev = e->pos[cur ^ 1];
    P.pos_abs[0] = e->pos[0]; P.pos_abs[1] = e->pos[1];
    P.t2 = pair ? (double)pair->time2_ms / 1000.0 : P.t; P.vis2 = pair ? pair->vis2 : nullptr;
    P.rb_table = ens ? ens->rb_table : nullptr; P.seeds = ens ? ens->seeds : nullptr;
    P.rows_ps = ens ? ens->rows_ps : 0; P.bps = ens ? ens->bps : 0;
    P.bps_magic = ens ? (uint32_t)((0x100000000ull + (uint64_t)ens->bps - 1) / (uint64_t)ens->bps) : 0u;
    if (ctx->diag & 1u) P.flags |= kNoInside;
    if (ctx->diag & 2u) P.flags |= kNoBoxCache;
    // the records: in the kernel-argument segment (a stand-alone sweep), or already in device memory -- an ensemble's
    // table, or the block the previous tick's compaction launch put there (zrk_run_ticks), which the sweep's loads find
    // in cacheable memory instead of re-fetching the argument segment in every workgroup
    if (rb_device) P.rb_table = (const char *)rb_device;
    if (ens || rb_device) std::memset(&P.rb, 0, sizeof(P.rb));
    else fill_radar_block(ctx, radars, R, flags, P.rb);
    const dim3 grid(P.nb + P.mb);                           // leading workgroups step the missiles
    if (pair) {
        if (!pend || !(flags & ZRK_F_ADVANCE) || ens || rb_device || !pair->radars2 || !pair->vis2)
            return fail(ctx, ZRK_E_INVALID, "zrk_tick_sweep: a pair launch advances, carries marks and has its records in its arguments");
        fill_radar_block(ctx, pair->radars2, R, flags, PP.rb2);   // (derived ahead by the loop: radar_block_ahead)
        using PairKernel = void (*)(const SweepParamsPair, const MissileArgs);
        static const PairKernel pairs[4] = {k_tick_sweep<false, true, false, true, true>, k_tick_sweep<true, true, false, true, true>,
                                            k_tick_sweep<false, true, true, true, true>,  k_tick_sweep<true, true, true, true, true>};
        const int w = ((flags & ZRK_F_PHILOX) ? 1 : 0) + (P.lidx ? 2 : 0);
        g_trace.mark("launch_sweep: arguments ready");
        if (ev_start && ev_stop) hipExtLaunchKernelGGL(pairs[w], grid, dim3(ZRK_BLOCK), 0, (hipStream_t)stream, ev_start, ev_stop, 0, PP, M);
        else hipLaunchKernelGGL(pairs[w], grid, dim3(ZRK_BLOCK), 0, (hipStream_t)stream, PP, M);
        return check_launch(ctx, "k_tick_sweep (pair)");
    }
    using Kernel = void (*)(const SweepParams, const MissileArgs);
    static const Kernel variants[12] = {
        k_tick_sweep<false, false, false>, k_tick_sweep<true, false, false>, k_tick_sweep<false, true, false>,
        k_tick_sweep<true, true, false>,   k_tick_sweep<false, false, true>, k_tick_sweep<true, false, true>,
        k_tick_sweep<false, true, true>,   k_tick_sweep<true, true, true>,
        // with removal marks (the overlapped loop, always advancing)
        k_tick_sweep<false, true, false, true>, k_tick_sweep<true, true, false, true>,
        k_tick_sweep<false, true, true, true>,  k_tick_sweep<true, true, true, true>};
    int which = ((flags & ZRK_F_PHILOX) ? 1 : 0) | ((flags & ZRK_F_ADVANCE) ? 2 : 0) | (P.lidx ? 4 : 0);
    if (pend) {
        if (!(flags & ZRK_F_ADVANCE)) return fail(ctx, ZRK_E_INVALID, "zrk_tick_sweep: removal marks only in the advancing sweep");
        which = 8 + ((flags & ZRK_F_PHILOX) ? 1 : 0) + (P.lidx ? 2 : 0);
    }
    // timed: the events ride on the dispatch itself (they read the kernel's own begin and end stamps, and cost the stream
    // no packet of their own -- a pair of hipEventRecord calls around the launch costs ~3 us of idle device each)
    if (ev_start && ev_stop) hipExtLaunchKernelGGL(variants[which], grid, dim3(ZRK_BLOCK), 0, (hipStream_t)stream, ev_start, ev_stop, 0, P, M);
    else hipLaunchKernelGGL(variants[which], grid, dim3(ZRK_BLOCK), 0, (hipStream_t)stream, P, M);
    return check_launch(ctx, "k_tick_sweep");
}

}  // namespace

ZRK_API int zrk_tick_sweep(zrk_ctx *ctx, const zrk_entities *e, int64_t n, int cur, int64_t time_ms,
                           const zrk_radar *radars, int R, uint32_t flags, uint64_t seed, uint64_t tick,
                           int64_t gid0, void *workspace, void *stream)
{
    return launch_sweep(ctx, e, n, cur, time_ms, radars, R, flags, seed, tick, gid0, workspace, stream, no_missiles());
}

namespace {

// Slots per thread of the single-launch compaction and its workgroup count (about one per compute unit).
int fused_items(const zrk_ctx *ctx, int64_t n)
{
    int items = (int)std::min<int64_t>(kFusedMaxItems, std::max<int64_t>(1, (n + (int64_t)kCompBlock * ctx->cus - 1) / ((int64_t)kCompBlock * ctx->cus)));
    if (ctx->env_items > 0) items = std::min(kFusedMaxItems, ctx->env_items);
    return items;
}

bool compacts_in_one_launch(const zrk_ctx *ctx, int64_t n)
{
    const int items = fused_items(ctx, n);
    return n > 0 && (n + (int64_t)kCompBlock * items - 1) / ((int64_t)kCompBlock * items) <= ctx->fused_max_blocks;
}

int launch_compact(zrk_ctx *ctx, const uint32_t *vis_mask, int64_t n, int R, int32_t base_index, void *workspace,
                   int32_t *det_idx, int64_t det_stride, int32_t *det_cnt, int64_t *packed, int64_t packed_capacity,
                   int64_t gid0, void *stream, const MissileArgs &M, uint32_t *zero_next,
                   bool union_bits = false, const EnsLaunch *ens = nullptr, const PutArgs *put = nullptr,
                   SideItem *defer = nullptr, int force_items = 0, uint32_t select = 0)
{
    if (!ctx || !vis_mask || !workspace) return fail(ctx, ZRK_E_INVALID, "zrk_compact: null argument");
    if ((det_idx && !det_cnt) || (!det_idx && !packed)) return fail(ctx, ZRK_E_INVALID, "zrk_compact: no output requested");
    if (R < 0 || R > ZRK_MAX_RADARS || n < 0 || det_stride < 0 || (packed && packed_capacity < 1))
        return fail(ctx, ZRK_E_INVALID, "zrk_compact: size out of range");
    hipStream_t s = (hipStream_t)stream;
    UnionBits U{0, 0, 0, 0};
    if (packed && union_bits) {
        U.words = (n + 63) / 64;
        U.mask_bytes = R <= 16 ? 2 : 4;
        U.mask_cap = (packed_capacity - 2 - U.words) * (8 / U.mask_bytes);
        if (U.mask_cap < 0) return fail(ctx, ZRK_E_CAPACITY, "zrk_compact: the union buffer does not hold the bitmap");
        if (U.words == 0) U.words = 1;               // n == 0 never gets here with work to do; keep the format flag
    }
    if (n == 0) {
        if (det_idx && hipMemsetAsync(det_cnt, 0, sizeof(int32_t) * (R + 1), s) != hipSuccess) return fail(ctx, ZRK_E_HIP, "memset det_cnt");
        if (packed && hipMemsetAsync(packed, 0, sizeof(int64_t) * (union_bits && packed_capacity > 1 ? 2 : 1), s) != hipSuccess)
            return fail(ctx, ZRK_E_HIP, "memset packed");
        return 0;
    }
    // single launch: about one workgroup per compute unit, each thread holding up to kFusedMaxItems slots
    const int items = force_items > 0 ? force_items : (ens ? ens->items : fused_items(ctx, n));
    const int64_t nbf = (n + (int64_t)kCompBlock * items - 1) / ((int64_t)kCompBlock * items);
    if (ens && (packed || nbf > kFusedMaxBlocks || nbf > ctx->fused_max_blocks))
        return fail(ctx, ZRK_E_INVALID, "zrk_compact: an ensemble takes the single-launch path and has no union list");
    if (ens || (force_items > 0 && nbf <= kFusedMaxBlocks) || compacts_in_one_launch(ctx, n)) {
        Workspace w = carve(workspace, 0, n);
        if (ctx->fused_ws != workspace) {          // first use by this context: no ticket, no record, no error
            if (hipMemsetAsync(workspace, 0, kFusedBytes, s) != hipSuccess) return fail(ctx, ZRK_E_HIP, "memset workspace");
            ctx->fused_ws = workspace;
        }
        if (++ctx->epoch == 0) ctx->epoch = 1;
        int lanes = 1;
        while (lanes < R + 1) lanes <<= 1;
        // Waiting in blockIdx order is safe because workgroups are DISPATCHED in blockIdx order: whoever is waited for
        // got its slot before the waiter did, whatever else shares the device (another rank's kernels, a collective).
        // HIP does not promise that order; this part does it, and the loop does not rest on it blindly: a wait that
        // runs out raises the workspace's error word (zrk_compact_status, checked by HotPathEngine.detections() and
        // after bench.py's timed region).  Grids that cannot be resident at once even alone take tickets from an atomic
        // counter instead (about 5 us of serialised atomics at 500 workgroups), which needs no assumption at all;
        // ZRK_COMPACT_ORDER=ticket forces that everywhere.
        int by_ticket = nbf > 2 * (int64_t)ctx->cus;
        if (ctx->env_order >= 0) by_ticket = ctx->env_order;
        CompactArgs C;
        C.vis = vis_mask; C.zero_next = zero_next; C.n = n; C.R = R; C.nb = (int)nbf; C.items = items; C.lanes = lanes;
        C.epoch = ctx->epoch; C.base_index = base_index; C.ctl = w.ctl; C.agg = w.agg; C.det_idx = det_idx;
        C.det_stride = det_stride; C.det_cnt = det_cnt; C.packed = packed; C.packed_capacity = packed_capacity; C.gid0 = gid0;
        C.bits = U;
        C.seg_blocks = ens ? (int32_t)(ens->rows_ps / ((int64_t)kCompBlock * items)) : 0;
        C.zero_own = (zero_next == vis_mask) ? 1 : 0;
        // (one scenario only: a batched ensemble's lists restart per scenario)
        C.group = (!ens && nbf > 64) ? 32 : 0;
        if (!ens && ctx->env_group >= 0) C.group = ctx->env_group;
        C.seg_slots = ens ? ens->rows_ps : 0;
        C.select = select;
        if (select && (det_idx || !packed || ens)) return fail(ctx, ZRK_E_INVALID, "zrk_compact: radars of interest apply to the union list alone (no per-radar lists, one scenario)");
        EnsembleArgs E;
        std::memset(&E, 0, sizeof(E));
        if (ens) E = ens->next;
        if (defer) {                                 // overlap mode: the side stream's thread launches it (lists only)
            // Beside other kernels its workgroups are placed as room appears, in dispatch order only within each XCD:
            // waiting in blockIdx order could then wait for a workgroup that has no compute unit yet while holding one
            // that another waiting kernel needs.  Tickets (a workgroup's place is taken when it starts to run) cannot.
            defer->C = C; defer->by_ticket = ctx->env_order >= 0 ? ctx->env_order : 1;
            return 0;
        }
        const int eparts = ens ? nblocks((int64_t)E.S * E.R, kCompBlock) : ((put && put->dst) ? 1 : 0);
        static PutArgs no_put;                       // (zero-initialised: dst == NULL)
        hipLaunchKernelGGL(k_compact_fused, dim3((int)nbf + (M.m > 0 ? 1 : 0) + eparts), dim3(kCompBlock), 0, s,
                           C, by_ticket, M, E, (put && !ens) ? *put : no_put);
        return check_launch(ctx, "k_compact_fused");
    }
    if (defer) return fail(ctx, ZRK_E_INVALID, "zrk_compact: overlap mode needs the single-launch compaction");
    if (select) return fail(ctx, ZRK_E_INVALID, "zrk_compact: radars of interest need the single-launch compaction");
    const int nb = nblocks(n, kCompBlock);
    Workspace w = carve(workspace, nb, n);
    hipLaunchKernelGGL(k_count_blocks, dim3(nb), dim3(kCompBlock), 0, s, vis_mask, n, R, nb, w.counts);
    hipLaunchKernelGGL(k_scan_counts, dim3(R + 1 + (M.m > 0 ? 1 : 0)), dim3(kScanThreads), 0, s, w.counts, w.offs, w.totals,
                       nb, R + 1, M);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(kCompBlock), 0, s, vis_mask, n, R, nb, w.offs, w.totals, base_index,
                       det_idx, det_stride, det_cnt, packed, packed_capacity, gid0, zero_next, U);
    return check_launch(ctx, "zrk_compact");
}

}  // namespace

ZRK_API int zrk_compact(zrk_ctx *ctx, const uint32_t *vis_mask, int64_t n, int R, int32_t base_index,
                        void *workspace, int32_t *det_idx, int64_t det_stride, int32_t *det_cnt, int64_t *packed,
                        int64_t packed_capacity, int64_t gid0, void *stream)
{
    return launch_compact(ctx, vis_mask, n, R, base_index, workspace, det_idx, det_stride, det_cnt, packed, packed_capacity,
                          gid0, stream, no_missiles(), nullptr);
}

ZRK_API int zrk_compact_bits(zrk_ctx *ctx, const uint32_t *vis_mask, int64_t n, int R, int32_t base_index,
                             void *workspace, int32_t *det_idx, int64_t det_stride, int32_t *det_cnt, int64_t *union_bits,
                             int64_t union_words, void *stream)
{
    return launch_compact(ctx, vis_mask, n, R, base_index, workspace, det_idx, det_stride, det_cnt, union_bits, union_words,
                          0, stream, no_missiles(), nullptr, true);
}

ZRK_API int64_t zrk_union_bits_words(int64_t n, int R, int64_t entries)
{
    if (n < 0 || R < 0 || R > ZRK_MAX_RADARS || entries < 0) return ZRK_E_INVALID;
    const int64_t mb = R <= 16 ? 2 : 4;
    return 2 + (n + 63) / 64 + (entries * mb + 7) / 8;
}

ZRK_API int zrk_compact_status(zrk_ctx *ctx, void *workspace, void *stream)
{
    if (!ctx || !workspace) return fail(ctx, ZRK_E_INVALID, "zrk_compact_status: null argument");
    int32_t ctl[4] = {0, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    // (the side stream's last launch of a call is not always taken in by the caller's stream: Side::tail_ws)
    if (hipStreamSynchronize(s) != hipSuccess || (ctx->side && hipStreamSynchronize(ctx->side->stream) != hipSuccess) ||
        hipMemcpy(ctl, workspace, sizeof(ctl), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(ctx, ZRK_E_HIP, "zrk_compact_status: copy failed");
    if (ctx->side && ctx->side->tail_ws) {
        int32_t tctl[4] = {0, 0, 0, 0};
        if (hipMemcpy(tctl, ctx->side->tail_ws, sizeof(tctl), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ctx, ZRK_E_HIP, "zrk_compact_status: copy failed");
        if (tctl[2] != 0 || tctl[0] != 0 || tctl[1] != 0) {
            (void)hipMemset(ctx->side->tail_ws, 0, kFusedCtlInts * sizeof(int32_t));
            return fail(ctx, ZRK_E_STATE, tctl[2] == 2 ? "zrk_compact: a workgroup of a call's last compaction gave up waiting for its predecessors"
                                                        : "zrk_compact: the control words of a call's last compaction were not as this library left them");
        }
    }
    if (ctx->side && ctx->side->rc.load() != 0)    // overlap mode: the side stream's thread gave up (its lists are not valid)
        return fail(ctx, ZRK_E_STATE, ctx->side->err);
    if (ctx->side && ctx->side->bar) {
        uint32_t bar[2] = {0, 0};
        if (hipMemcpy(bar, ctx->side->bar, sizeof(bar), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ctx, ZRK_E_HIP, "zrk_compact_status: the missile barrier's words could not be read");
        if (bar[1] != 0) {
            // reported once: the word goes down again (every workgroup counts itself in whether it gave up or not, so the arrival
            // counter still matches the epoch), and the next call starts clean
            (void)hipMemset(ctx->side->bar + 1, 0, sizeof(uint32_t));
            return fail(ctx, ZRK_E_STATE, "a pair launch's missile workgroups did not all reach their barrier: the second tick's events are not valid");
        }
    }
    uint32_t dev_fault = 0;
    if (hipMemcpyFromSymbol(&dev_fault, HIP_SYMBOL(g_device_fault), sizeof(dev_fault)) != hipSuccess)
        return fail(ctx, ZRK_E_HIP, "zrk_compact_status: the device fault word could not be read");
    if (dev_fault != 0)
        return fail(ctx, ZRK_E_STATE, "a sweep was handed a radar record without an address (device fault word " + std::to_string(dev_fault) +
                                      "): its visibility masks are not valid");
    if (ctx->fused_ws != workspace) return 0;      // never used with this context
    if (ctl[2] == 0 && ctl[0] == 0 && ctl[1] == 0) return 0;
    ctx->fused_ws = nullptr;                       // cleared again on the next use
    return fail(ctx, ZRK_E_STATE, ctl[2] == 2 ? "zrk_compact: a workgroup gave up waiting for its predecessors"
                                              : "zrk_compact: workspace control words were not as this library left them");
}

ZRK_API int zrk_noise_apply(zrk_ctx *ctx, double *pos, int64_t capacity, const int32_t *idx, int32_t idx_base,
                            const double *noise, int64_t k, void *stream)
{
    if (!ctx || !pos || (k > 0 && (!idx || !noise))) return fail(ctx, ZRK_E_INVALID, "zrk_noise_apply: null argument");
    if (k <= 0) return 0;
    hipLaunchKernelGGL(k_noise_apply, dim3(nblocks(k, 256)), dim3(256), 0, (hipStream_t)stream, pos, capacity, idx,
                       idx_base, noise, k);
    return check_launch(ctx, "k_noise_apply");
}

ZRK_API int zrk_missile_step(zrk_ctx *ctx, const zrk_entities *e, int cur, const zrk_missiles *mis, int64_t m,
                             int64_t time_ms, int64_t dt_ms, int apply_kills, void *stream)
{
    if (!ctx || !e || !mis) return fail(ctx, ZRK_E_INVALID, "zrk_missile_step: null argument");
    if (m < 0 || m > mis->capacity || (cur != 0 && cur != 1)) return fail(ctx, ZRK_E_INVALID, "zrk_missile_step: m/cur out of range");
    hipStream_t s = (hipStream_t)stream;
    if (m == 0) {
        if (hipMemsetAsync(mis->ev_count, 0, sizeof(int32_t), s) != hipSuccess) return fail(ctx, ZRK_E_HIP, "memset ev_count");
        return 0;
    }
    const double t = (double)time_ms / 1000.0, dts = (double)dt_ms / 1000.0;
    hipLaunchKernelGGL(k_missile_step, dim3(nblocks(m, 256)), dim3(256), 0, s, e->start_pos, e->velocity, e->start_time,
                       e->alive, e->list_index, e->pos[cur ^ 1], e->capacity, mis->slot, mis->target, mis->radius,
                       mis->period, mis->status, mis->ev_code, m, t, dts);
    if (m <= 1024 * (int64_t)kMissileItems) {
        hipLaunchKernelGGL(k_missile_finish, dim3(1), dim3(1024), 0, s, mis->ev_code, mis->slot, mis->target, m,
                           mis->ev_missile, mis->ev_target, mis->ev_count, apply_kills, e->alive, e->pos[cur],
                           e->pos[cur ^ 1], e->capacity);
        return check_launch(ctx, "k_missile_finish");
    }
    hipLaunchKernelGGL(k_missile_events, dim3(1), dim3(1024), 0, s, mis->ev_code, mis->slot, mis->target, m,
                       mis->ev_missile, mis->ev_target, mis->ev_count);
    if (apply_kills)
        hipLaunchKernelGGL(k_apply_events, dim3(1), dim3(256), 0, s, e->alive, e->pos[cur], e->pos[cur ^ 1], e->capacity,
                           mis->ev_missile, mis->ev_target, mis->ev_count, mis->capacity);
    return check_launch(ctx, "zrk_missile_step");
}

ZRK_API int zrk_kill_slots(zrk_ctx *ctx, const zrk_entities *e, int src, const int32_t *slots, int64_t k, void *stream)
{
    if (!ctx || !e || (k > 0 && !slots)) return fail(ctx, ZRK_E_INVALID, "zrk_kill_slots: null argument");
    if (src != 0 && src != 1) return fail(ctx, ZRK_E_INVALID, "zrk_kill_slots: src out of range");
    if (k <= 0) return 0;
    hipLaunchKernelGGL(k_kill_slots, dim3(nblocks(k, 256)), dim3(256), 0, (hipStream_t)stream, e->alive, e->pos[src],
                       e->pos[src ^ 1], e->capacity, slots, k);
    return check_launch(ctx, "k_kill_slots");
}

ZRK_API int zrk_apply_events(zrk_ctx *ctx, const zrk_entities *e, int src, const zrk_missiles *mis, void *stream)
{
    if (!ctx || !e || !mis) return fail(ctx, ZRK_E_INVALID, "zrk_apply_events: null argument");
    if (src != 0 && src != 1) return fail(ctx, ZRK_E_INVALID, "zrk_apply_events: src out of range");
    hipLaunchKernelGGL(k_apply_events, dim3(1), dim3(256), 0, (hipStream_t)stream, e->alive, e->pos[src], e->pos[src ^ 1],
                       e->capacity, mis->ev_missile, mis->ev_target, mis->ev_count, mis->capacity);
    return check_launch(ctx, "k_apply_events");
}

ZRK_API int zrk_launch_solve(zrk_ctx *ctx, const zrk_entities *e, int cur, const zrk_launch_req *req,
                             zrk_launch_res *res, int64_t k, void *stream)
{
    if (!ctx || !e || (k > 0 && (!req || !res))) return fail(ctx, ZRK_E_INVALID, "zrk_launch_solve: null argument");
    if (cur != 0 && cur != 1) return fail(ctx, ZRK_E_INVALID, "zrk_launch_solve: cur out of range");
    if (k <= 0) return 0;
    hipLaunchKernelGGL(k_launch_solve, dim3(nblocks(k, 64)), dim3(64), 0, (hipStream_t)stream, e->velocity, e->kind,
                       e->pos[cur], e->capacity, req, res, k);
    return check_launch(ctx, "k_launch_solve");
}

ZRK_API int zrk_launch_salvo(zrk_ctx *ctx, const zrk_entities *e, int cur, const zrk_missiles *mis, int64_t n, int64_t m,
                             const zrk_launch_req *req, zrk_launch_res *res, int64_t k, int64_t time_ms, int32_t list_base,
                             int32_t *count_out, void *stream)
{
    if (!ctx || !e || !mis || (k > 0 && (!req || !res))) return fail(ctx, ZRK_E_INVALID, "zrk_launch_salvo: null argument");
    if ((cur != 0 && cur != 1) || n < 0 || m < 0 || k < 0) return fail(ctx, ZRK_E_INVALID, "zrk_launch_salvo: size out of range");
    if (n + k > e->capacity || m + k > mis->capacity)
        return fail(ctx, ZRK_E_CAPACITY, "zrk_launch_salvo: the tables need room for every request (n + k rows, m + k missile rows)");
    hipStream_t s = (hipStream_t)stream;
    if (k == 0) {
        if (count_out && hipMemsetAsync(count_out, 0, sizeof(int32_t), s) != hipSuccess) return fail(ctx, ZRK_E_HIP, "memset count");
        return 0;
    }
    if (int rc = zrk_launch_solve(ctx, e, cur, req, res, k, stream)) return rc;
    hipLaunchKernelGGL(k_launch_append, dim3(1), dim3(1024), 0, s, req, res, k, (double *)e->start_pos, (double *)e->velocity,
                       (double *)e->start_time, e->alive, (uint8_t *)e->kind, e->pos[0], e->pos[1], (int32_t *)e->list_index,
                       e->capacity, n, list_base, (int32_t *)mis->slot, (int32_t *)mis->target, (double *)mis->radius, mis->period,
                       mis->status, m, (double)time_ms / 1000.0, count_out);
    return check_launch(ctx, "k_launch_append");
}

namespace {
inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
}

ZRK_API int64_t zrk_ccp_scratch_bytes(int64_t D, int64_t T)
{
    if (D < 0 || T < 0) return ZRK_E_INVALID;
    return align256(D * (int64_t)sizeof(CcpCand)) + align256(T) + align256(4 * T) + align256(D) + align256(D) + 256;
}

ZRK_API int zrk_ccp_link(zrk_ctx *ctx, const double *det_pos, const double *det_speed, int64_t D, const double *trk_ref,
                         const double *trk_upd, int64_t T, double now_s, double slack_s, int32_t *match, void *scratch,
                         void *stream)
{
    if (!ctx || D < 0 || T < 0 || (D > 0 && (!det_pos || !det_speed || !match || !scratch)) || (T > 0 && (!trk_ref || !trk_upd)))
        return fail(ctx, ZRK_E_INVALID, "zrk_ccp_link: null argument");
    if (D == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    char *p = (char *)scratch;
    CcpCand *cand = (CcpCand *)p;            p += align256(D * (int64_t)sizeof(CcpCand));
    uint8_t *taken = (uint8_t *)p;           p += align256(T);
    int32_t *interest = (int32_t *)p;        p += align256(4 * T);
    uint8_t *state = (uint8_t *)p;           p += align256(D);
    uint8_t *only = (uint8_t *)p;            p += align256(D);
    int32_t *counters = (int32_t *)p;
    if (hipMemsetAsync(taken, 0, (size_t)align256(T), s) != hipSuccess || hipMemsetAsync(state, 0, (size_t)align256(D), s) != hipSuccess ||
        hipMemsetAsync(interest, 0x7F, (size_t)align256(4 * T), s) != hipSuccess)
        return fail(ctx, ZRK_E_HIP, "zrk_ccp_link: memset");
    const int gd = nblocks(D, 256);
    hipLaunchKernelGGL(k_ccp_candidates, dim3(gd), dim3(256), 0, s, det_pos, det_speed, D, trk_ref, trk_upd, T, now_s, slack_s,
                       (const uint8_t *)nullptr, (const uint8_t *)nullptr, cand);
    for (int round = 0; round < 1000000; ++round) {
        if (hipMemsetAsync(counters, 0, 8, s) != hipSuccess || hipMemsetAsync(counters + 2, 0x7F, 4, s) != hipSuccess)
            return fail(ctx, ZRK_E_HIP, "zrk_ccp_link: memset");
        for (int phase = 0; phase < 3; ++phase)
            hipLaunchKernelGGL(k_ccp_round, dim3(gd), dim3(256), 0, s, phase, D, cand, taken, interest, match, state, counters,
                               (const int32_t *)nullptr, (const int32_t *)nullptr);
        hipLaunchKernelGGL(k_ccp_reset_interest, dim3(gd), dim3(256), 0, s, D, cand, state, interest, (const int32_t *)nullptr,
                           (const int32_t *)nullptr, (const int32_t *)nullptr);
        int32_t h[2] = {0, 0};
        if (hipMemcpyAsync(h, counters, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return fail(ctx, ZRK_E_HIP, "zrk_ccp_link: reading the round's counters");
        if (h[0] == 0 && h[1] == 0) break;
        if (h[1] > 0) {
            // every detection left has used up its kept candidates although more were in gate: scan again for those,
            // without the tracks that are gone (state 2 -> `only`, back to unresolved)
            if (hipMemcpyAsync(only, state, (size_t)D, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(ctx, ZRK_E_HIP, "zrk_ccp_link: copy");
            hipLaunchKernelGGL(k_ccp_rescan_prepare, dim3(gd), dim3(256), 0, s, D, only, state);
            hipLaunchKernelGGL(k_ccp_candidates, dim3(gd), dim3(256), 0, s, det_pos, det_speed, D, trk_ref, trk_upd, T, now_s, slack_s,
                               (const uint8_t *)taken, (const uint8_t *)only, cand);
        }
    }
    return check_launch(ctx, "zrk_ccp_link");
}

namespace {
// the spatial index of the candidate pass (k_ccp_candidates_grid): cells, partial results, the sorted copy of the tracks
int32_t ccp_grid_cells(int64_t T) { return (int32_t)std::min<int64_t>(std::max<int64_t>(4 * T, 4096), (1 << 22) - 4096); }
int32_t ccp_grid_blocks(int64_t dmax, int64_t T) { return nblocks(std::max<int64_t>(std::max(dmax, T), 1), 256); }
int64_t ccp_grid_bytes(int64_t dmax, int64_t T)
{
    const int64_t cells = ccp_grid_cells(T);
    return align256(40 * (int64_t)ccp_grid_blocks(dmax, T)) + align256((int64_t)sizeof(CcpGrid)) + 2 * align256(4 * (cells + 1)) +
           align256(4 * T) + 4096 + align256(4 * T) + align256(24 * T) + align256(8 * T);
}
}  // namespace

ZRK_API int64_t zrk_ccp_step_scratch_bytes(int64_t dmax, int64_t track_capacity)
{
    if (dmax < 0 || track_capacity < 0) return ZRK_E_INVALID;
    const int64_t T = 2 * track_capacity;
    return zrk_ccp_scratch_bytes(dmax, T) + align256(24 * dmax) + align256(8 * dmax) + align256(24 * T) + align256(8 * T) +
           align256(4 * dmax) + 256 + align256(4 * (dmax + 1)) + align256(4 * dmax) + align256(4 * T) + align256(4 * dmax) + 256 +
           ccp_grid_bytes(dmax, T);
}

ZRK_API int zrk_ccp_step(zrk_ctx *ctx, const zrk_entities *e, int cur, const double *speed_mod, const int32_t *seq,
                         const int32_t *seq_count, int64_t dmax, const zrk_ccp_tracks *trk, const zrk_ccp_launchers *lch,
                         const zrk_ccp_out *out, double now_s, double slack_s, int rounds, void *scratch, void *stream)
{
    if (!ctx || !e || !speed_mod || !seq || !seq_count || !trk || !lch || !out || !scratch || (cur != 0 && cur != 1))
        return fail(ctx, ZRK_E_INVALID, "zrk_ccp_step: null argument");
    if (dmax < 0 || trk->capacity < 0 || lch->L < 0 || lch->L > 64 || rounds < 1)
        return fail(ctx, ZRK_E_INVALID, "zrk_ccp_step: size out of range (at most 64 launchers)");
    if (dmax == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t T = 2 * trk->capacity;
    // the association's scratch first (as zrk_ccp_link lays it out), then this step's own
    char *p = (char *)scratch;
    CcpCand *cand = (CcpCand *)p;            p += align256(dmax * (int64_t)sizeof(CcpCand));
    uint8_t *taken = (uint8_t *)p;           p += align256(T);
    int32_t *interest = (int32_t *)p;        p += align256(4 * T);
    uint8_t *state = (uint8_t *)p;           p += align256(dmax);
    uint8_t *only = (uint8_t *)p;            p += align256(dmax);
    int32_t *counters = (int32_t *)p;        p += 256;
    CcpStepArgs A;
    A.pos_cur = e->pos[cur]; A.pos_prev = e->pos[cur ^ 1]; A.t0 = e->start_time; A.speed = speed_mod; A.cap = e->capacity;
    A.seq = seq; A.seq_count = seq_count; A.dmax = dmax; A.trk = *trk; A.lch = *lch; A.out = *out;
    A.now_s = now_s; A.slack_s = slack_s;
    A.det_pos = (double *)p;                 p += align256(24 * dmax);
    A.det_speed = (double *)p;               p += align256(8 * dmax);
    A.trk_ref = (double *)p;                 p += align256(24 * T);
    A.trk_upd = (double *)p;                 p += align256(8 * T);
    A.kill = (int32_t *)p;                   p += align256(4 * dmax);
    A.sizes = (int32_t *)p;                  p += 256;
    A.want = (int32_t *)p;                   p += align256(4 * (dmax + 1));
    A.new_rank = (int32_t *)p;               p += align256(4 * dmax);
    A.winner = (int32_t *)p;                 p += align256(4 * T);
    A.counters = counters;
    int32_t *match = (int32_t *)p;           p += align256(4 * dmax) + 256;
    // many tracks: the candidate pass goes through a spatial index built on the device each tick (ZRK_CCP_GRID=0 never, 1 always)
    bool use_grid = T >= 8192;
    if (const char *v = std::getenv("ZRK_CCP_GRID")) use_grid = v[0] == '1';
    CcpGridArgs G;
    std::memset(&G, 0, sizeof(G));
    G.det_pos = A.det_pos; G.det_speed = A.det_speed; G.trk_ref = A.trk_ref; G.trk_upd = A.trk_upd; G.sizes = A.sizes;
    G.dmax = dmax; G.tmax = T; G.now_s = now_s; G.slack_s = slack_s;
    G.blocks = ccp_grid_blocks(dmax, T); G.cells_cap = ccp_grid_cells(T);
    G.partial = (double *)p;                 p += align256(40 * (int64_t)G.blocks);
    G.grid = (CcpGrid *)p;                   p += align256((int64_t)sizeof(CcpGrid));
    G.count = (int32_t *)p;                  p += align256(4 * ((int64_t)G.cells_cap + 1));
    G.start = (int32_t *)p;                  p += align256(4 * ((int64_t)G.cells_cap + 1));
    G.key = (int32_t *)p;                    p += align256(4 * T);
    G.block_sum = (int32_t *)p;              p += 4096;
    G.sorted_idx = (int32_t *)p;             p += align256(4 * T);
    G.sorted_ref = (double *)p;              p += align256(24 * T);
    G.sorted_upd = (double *)p;              p += align256(8 * T);
    // (the step's working words cleared by ONE launch: six memsets were six dependent launches of a closed-loop tick's seventy)
    {
        CcpClearArgs Z;
        Z.zero[0] = (uint32_t *)taken;    Z.zero_words[0] = align256(T) / 4;
        Z.zero[1] = (uint32_t *)state;    Z.zero_words[1] = align256(dmax) / 4;
        Z.zero[2] = (uint32_t *)counters; Z.zero_words[2] = 64;
        Z.zero[3] = use_grid ? (uint32_t *)G.count : nullptr; Z.zero_words[3] = use_grid ? (int64_t)G.cells_cap + 1 : 0;
        Z.ones = (uint32_t *)interest;    Z.ones_words = align256(4 * T) / 4;
        Z.status = out->status;
        int64_t most = Z.ones_words;
        for (int k = 0; k < 4; ++k) most = std::max(most, Z.zero_words[k]);
        hipLaunchKernelGGL(k_ccp_clear, dim3(nblocks(most, 256)), dim3(256), 0, s, Z);
    }
    const int gd = nblocks(dmax, 256), gt = nblocks(std::max<int64_t>(dmax, T), 256);
    const int gw = nblocks(dmax, 4);                 // (k_ccp_candidates_grid: a wave per detection)
    hipLaunchKernelGGL(k_ccp_gather, dim3(gt), dim3(256), 0, s, A);
    if (use_grid) {
        const int sb = (G.cells_cap + 1 + 4095) / 4096;
        hipLaunchKernelGGL(k_ccp_grid_partials, dim3(G.blocks), dim3(256), 0, s, G);
        hipLaunchKernelGGL(k_ccp_grid_params, dim3(1), dim3(256), 0, s, G);
        hipLaunchKernelGGL(k_ccp_grid_count, dim3(nblocks(T, 256)), dim3(256), 0, s, G);
        hipLaunchKernelGGL(k_ccp_grid_scan, dim3(sb), dim3(1024), 0, s, G, 0);
        hipLaunchKernelGGL(k_ccp_grid_scan, dim3(1), dim3(1024), 0, s, G, 1);
        hipLaunchKernelGGL(k_ccp_grid_scan, dim3(sb), dim3(1024), 0, s, G, 2);
        hipLaunchKernelGGL(k_ccp_grid_scatter, dim3(nblocks(T, 256)), dim3(256), 0, s, G);
        hipLaunchKernelGGL(k_ccp_candidates_grid, dim3(gw), dim3(256), 0, s, G, (const uint8_t *)nullptr, (const uint8_t *)nullptr, cand,
                           (const int32_t *)nullptr);
    } else
    hipLaunchKernelGGL(k_ccp_candidates, dim3(gd), dim3(256), 0, s, A.det_pos, A.det_speed, dmax, A.trk_ref, A.trk_upd, T, now_s, slack_s,
                       (const uint8_t *)nullptr, (const uint8_t *)nullptr, cand, (const int32_t *)A.sizes, (const int32_t *)nullptr);
    for (int r = 0; r < rounds; ++r) {               // (each launch looks at the `done` word first: the rounds end themselves)
        hipLaunchKernelGGL(k_ccp_round_begin, dim3(1), dim3(1), 0, s, counters);
        for (int phase = 0; phase < 3; ++phase)
            hipLaunchKernelGGL(k_ccp_round, dim3(gd), dim3(256), 0, s, phase, dmax, cand, taken, interest, match, state, counters,
                               (const int32_t *)A.kill, (const int32_t *)A.sizes);
        hipLaunchKernelGGL(k_ccp_round_close, dim3(gd), dim3(256), 0, s, dmax, cand, state, interest, A.kill, A.sizes, only, counters);
        hipLaunchKernelGGL(k_ccp_round_flags, dim3(1), dim3(1), 0, s, counters);
        if (use_grid)
            hipLaunchKernelGGL(k_ccp_candidates_grid, dim3(gw), dim3(256), 0, s, G, (const uint8_t *)taken, (const uint8_t *)only, cand,
                               (const int32_t *)(counters + 4));
        else
        hipLaunchKernelGGL(k_ccp_candidates, dim3(gd), dim3(256), 0, s, A.det_pos, A.det_speed, dmax, A.trk_ref, A.trk_upd, T, now_s, slack_s,
                           (const uint8_t *)taken, (const uint8_t *)only, cand, (const int32_t *)A.sizes, (const int32_t *)(counters + 4));
    }
    hipLaunchKernelGGL(k_ccp_tail, dim3(1), dim3(1024), 0, s, A, taken, match, state, counters, (const CcpCand *)cand);
    hipLaunchKernelGGL(k_ccp_finish, dim3(1), dim3(1), 0, s, counters, out->status);
    hipLaunchKernelGGL(k_ccp_scan, dim3(1), dim3(1024), 0, s, A, (const int32_t *)match);
    hipLaunchKernelGGL(k_ccp_launch, dim3(1), dim3(64), 0, s, A);
    hipLaunchKernelGGL(k_ccp_apply, dim3(gd), dim3(256), 0, s, A, 0);
    hipLaunchKernelGGL(k_ccp_apply, dim3(gd), dim3(256), 0, s, A, 1);
    hipLaunchKernelGGL(k_ccp_count_new, dim3(1), dim3(1), 0, s, A);
    return check_launch(ctx, "zrk_ccp_step");
}

ZRK_API int zrk_ccp_requests(zrk_ctx *ctx, const zrk_ccp_out *out, int64_t dmax, const zrk_ccp_launchers *lch, const double *missile_params,
                             zrk_launch_req *req, int64_t k_max, int32_t *count, void *stream)
{
    if (!ctx || !out || !lch || !missile_params || (k_max > 0 && !req) || !out->obj || !out->launcher || !out->count)
        return fail(ctx, ZRK_E_INVALID, "zrk_ccp_requests: null argument");
    if (dmax < 0 || k_max < 0 || lch->L < 0 || lch->L > 64) return fail(ctx, ZRK_E_INVALID, "zrk_ccp_requests: size out of range");
    hipLaunchKernelGGL(k_ccp_requests, dim3(1), dim3(1024), 0, (hipStream_t)stream, *out, dmax, lch->pos, missile_params, req, k_max, count);
    return check_launch(ctx, "k_ccp_requests");
}

ZRK_API int zrk_ccp_add_missile(zrk_ctx *ctx, const zrk_ccp_tracks *trk, int32_t row, double now_s, void *stream)
{
    if (!ctx || !trk) return fail(ctx, ZRK_E_INVALID, "zrk_ccp_add_missile: null argument");
    hipLaunchKernelGGL(k_ccp_add_missile, dim3(1), dim3(64), 0, (hipStream_t)stream, *trk, row, now_s);
    return check_launch(ctx, "k_ccp_add_missile");
}

// What every bounded host wait of the library does when its condition never comes: `what` 0 the spin helper itself, 1 a
// ring the helper thread never empties.  No device needed (tests/test_host_logic.py).
// ---- the battery's closed loop (include/zrk_hot.h: zrk_battery) ----
namespace {
int battery_check(zrk_ctx *ctx, const zrk_battery *b, const char *who)
{
    if (!ctx || !b) return fail(ctx, ZRK_E_INVALID, std::string(who) + ": null argument");
    if (b->L < 0 || b->L > 64 || b->k_max < 1 || b->n_missiles < 0 || b->row0 < 0 || b->log_cap < 0 || !b->sal_row || !b->sal_launcher ||
        !b->sal_missile || !b->sal_rc || !b->sal_air || !b->sal_V || !b->sal_count || !b->air_count || !b->air_missile || !b->stack ||
        !b->top || !b->mi_pos || !b->mi_speed || !b->mi_period || !b->mi_radius || !b->speed_mod || !b->log_solve || !b->log_V ||
        !b->log_event || !b->log_count)
        return fail(ctx, ZRK_E_INVALID, std::string(who) + ": incomplete battery description");
    return 0;
}
inline int battery_slot(int64_t build_tick) { return build_tick < 0 ? -1 : (int)(build_tick % 3); }
}  // namespace

ZRK_API int zrk_ctx_keep_prev(zrk_ctx *ctx, double *buf)
{
    if (!ctx) return ZRK_E_INVALID;
    ctx->frozen_prev = buf;
    return 0;
}

ZRK_API int zrk_battery_speed_column(zrk_ctx *ctx, const zrk_entities *e, int64_t n, double *speed_mod, void *stream)
{
    if (!ctx || !e || !speed_mod || n < 0 || n > e->capacity) return fail(ctx, ZRK_E_INVALID, "zrk_battery_speed_column: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_battery_speed, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, e->velocity, e->capacity, n, speed_mod);
    return check_launch(ctx, "k_battery_speed");
}

ZRK_API int zrk_battery_activate(zrk_ctx *ctx, const zrk_battery *b, const zrk_entities *e, const zrk_missiles *mis, int64_t tick,
                                 void *workspace, void *stream)
{
    if (int rc = battery_check(ctx, b, "zrk_battery_activate")) return rc;
    if (!e || !mis) return fail(ctx, ZRK_E_INVALID, "zrk_battery_activate: null argument");
    if (tick < 3) return 0;                                   // (the first requests are made in tick 0 at the earliest)
    if ((int64_t)b->row0 + b->n_missiles > e->capacity) return fail(ctx, ZRK_E_INVALID, "zrk_battery_activate: the magazine's rows are not in the table");
    WaveBox *boxes = workspace ? carve(workspace, 0, e->capacity).boxes : nullptr;
    hipLaunchKernelGGL(k_battery_activate, dim3(nblocks(b->k_max, 256)), dim3(256), 0, (hipStream_t)stream, *b, battery_slot(tick - 3), e->alive,
                       mis->status, boxes);
    return check_launch(ctx, "k_battery_activate");
}

ZRK_API int zrk_battery_launchers(zrk_ctx *ctx, const zrk_battery *b, const zrk_entities *e, int cur, const zrk_missiles *mis, int64_t tick,
                                  int64_t time_ms, void *stream)
{
    if (int rc = battery_check(ctx, b, "zrk_battery_launchers")) return rc;
    if (!e || !mis || (cur != 0 && cur != 1)) return fail(ctx, ZRK_E_INVALID, "zrk_battery_launchers: bad argument");
    if (tick < 1) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int serve = battery_slot(tick - 1), back = battery_slot(tick - 2);
    hipLaunchKernelGGL(k_battery_launchers, dim3(1), dim3(64), 0, s, *b, back, serve);
    hipLaunchKernelGGL(k_battery_solve, dim3(nblocks(b->k_max, 64)), dim3(64), 0, s, *b, serve, e->velocity, e->kind, e->pos[cur], e->capacity);
    hipLaunchKernelGGL(k_battery_assign, dim3(1), dim3(1024), 0, s, *b, serve, (int)tick, (double)time_ms / 1000.0, const_cast<double *>(e->start_pos),
                       const_cast<double *>(e->velocity), const_cast<double *>(e->start_time), e->alive, const_cast<uint8_t *>(e->kind), e->pos[0],
                       e->pos[1], e->capacity, const_cast<int32_t *>(mis->slot), const_cast<int32_t *>(mis->target),
                       const_cast<double *>(mis->radius), mis->period, mis->status);
    return check_launch(ctx, "zrk_battery_launchers");
}

ZRK_API int zrk_battery_announce(zrk_ctx *ctx, const zrk_battery *b, const zrk_ccp_tracks *trk, int64_t tick, double now_s, void *stream)
{
    if (int rc = battery_check(ctx, b, "zrk_battery_announce")) return rc;
    if (!trk) return fail(ctx, ZRK_E_INVALID, "zrk_battery_announce: null argument");
    if (tick < 2) return 0;
    hipLaunchKernelGGL(k_battery_announce, dim3(1), dim3(1024), 0, (hipStream_t)stream, *b, battery_slot(tick - 2), *trk, now_s);
    return check_launch(ctx, "k_battery_announce");
}

ZRK_API int zrk_battery_sequence(zrk_ctx *ctx, const int32_t *det_idx, int64_t det_stride, const int32_t *det_cnt, int R, int32_t base_index,
                                 const uint32_t *vis_mask, const int32_t *row_of_list, int32_t *seq, int32_t *seq_count, int64_t seq_cap,
                                 void *stream)
{
    if (!ctx || !det_idx || !det_cnt || !vis_mask || !seq || !seq_count || R < 0 || R > ZRK_MAX_RADARS || det_stride < 0 || seq_cap < 0)
        return fail(ctx, ZRK_E_INVALID, "zrk_battery_sequence: bad argument");
    hipLaunchKernelGGL(k_battery_sequence, dim3(1), dim3(1024), 0, (hipStream_t)stream, det_idx, det_stride, det_cnt, R, base_index, vis_mask,
                       row_of_list, seq, seq_count, seq_cap);
    return check_launch(ctx, "k_battery_sequence");
}

ZRK_API int zrk_battery_requests(zrk_ctx *ctx, const zrk_battery *b, const zrk_ccp_out *out, int64_t dmax, int64_t tick, void *stream)
{
    if (int rc = battery_check(ctx, b, "zrk_battery_requests")) return rc;
    if (!out || tick < 0) return fail(ctx, ZRK_E_INVALID, "zrk_battery_requests: bad argument");
    hipLaunchKernelGGL(k_battery_requests, dim3(1), dim3(1024), 0, (hipStream_t)stream, *b, battery_slot(tick), *out, dmax);
    return check_launch(ctx, "k_battery_requests");
}

ZRK_API int zrk_battery_log_events(zrk_ctx *ctx, const zrk_battery *b, const zrk_missiles *mis, int64_t tick, void *stream)
{
    if (int rc = battery_check(ctx, b, "zrk_battery_log_events")) return rc;
    if (!mis) return fail(ctx, ZRK_E_INVALID, "zrk_battery_log_events: null argument");
    hipLaunchKernelGGL(k_battery_log_events, dim3(1), dim3(256), 0, (hipStream_t)stream, *b, mis->ev_missile, mis->ev_target, mis->ev_count, (int)tick);
    return check_launch(ctx, "k_battery_log_events");
}

ZRK_API int zrk_selftest_host_wait(int what, int limit_ms)
{
    const auto limit = std::chrono::milliseconds(std::max(1, limit_ms));
    if (what == 0) return spin_until([] { return false; }, limit) ? 0 : ZRK_E_STATE;
    std::atomic<uint64_t> head{8}, tail{0};
    return spin_until([&] { return head.load() - tail.load(std::memory_order_acquire) < 8; }, limit) ? 0 : ZRK_E_STATE;
}

ZRK_API int zrk_selftest_math(zrk_ctx *ctx, int op, const double *a, const double *b, double *y, int64_t n, void *stream)
{
    if (!ctx || !a || !b || !y) return fail(ctx, ZRK_E_INVALID, "zrk_selftest_math: null argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_selftest_math, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, op, a, b, y, n);
    return check_launch(ctx, "k_selftest_math");
}

ZRK_API int zrk_selftest_noise(zrk_ctx *ctx, uint64_t seed, uint64_t tick, uint32_t ordinal, int64_t entity0,
                               double *out, int64_t n, void *stream)
{
    if (!ctx || !out) return fail(ctx, ZRK_E_INVALID, "zrk_selftest_noise: null argument");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_selftest_noise, dim3(nblocks(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, tick, ordinal,
                       entity0, out, n);
    return check_launch(ctx, "k_selftest_noise");
}

// ---------------------------------------------------------------------------------------------
// Host loop: K ticks of the L1 path without returning to the caller (the headless counterpart of
// Manager.run_simulation's per-tick module calls, reference modules/Manager.py:111-140).
// ---------------------------------------------------------------------------------------------
ZRK_API int zrk_scan_advance(zrk_radar *radars, const zrk_scan *scan, int R)
{
    if (R < 0 || (R > 0 && (!radars || !scan))) return ZRK_E_INVALID;
    for (int r = 0; r < R; ++r) scan_advance_one(radars[r], scan[r]);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Per-tick exchange of the detection list between the GPUs of a node: RCCL all-gather issued from here, on a
// stream of its own, with events both ways -- no interpreter between a tick's compaction and its collective.
// The library is bound at run time (dlopen: the process usually holds PyTorch's copy of librccl already, and
// two copies in one process is asking for trouble), the communicator is this module's own.
// ---------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, zrk_rccl_id, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    // (optional: the direct pattern and the self-check)
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
};

constexpr int kNcclInt64 = 4;          // ncclDataType_t: ncclInt64

bool load_rccl(const char *path, RcclApi &api, std::string &err)
{
    const char *names[3] = {path, "librccl.so.1", "librccl.so"};
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) { err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return false; }
    api.GetUniqueId = (int (*)(void *))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void **, int, zrk_rccl_id, int))dlsym(api.lib, "ncclCommInitRank");
    api.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(api.lib, "ncclAllGather");
    api.CommDestroy = (int (*)(void *))dlsym(api.lib, "ncclCommDestroy");
    api.GetErrorString = (const char *(*)(int))dlsym(api.lib, "ncclGetErrorString");
    api.Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(api.lib, "ncclSend");
    api.Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(api.lib, "ncclRecv");
    api.GroupStart = (int (*)())dlsym(api.lib, "ncclGroupStart");
    api.GroupEnd = (int (*)())dlsym(api.lib, "ncclGroupEnd");
    api.CommCount = (int (*)(void *, int *))dlsym(api.lib, "ncclCommCount");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) { err = "RCCL symbols missing"; return false; }
    return true;
}

}  // namespace

struct zrk_exchange {
    RcclApi api;
    void *comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t cstream = nullptr;
    hipEvent_t ready[ZRK_EXCHANGE_SLOTS] = {}, done[ZRK_EXCHANGE_SLOTS] = {};
    bool posted[ZRK_EXCHANGE_SLOTS] = {};
    // hand-over by flag (zrk_run_ticks_x): a word of device memory that the NEXT tick's sweep raises to `seq` as it starts;
    // a one-lane kernel on the exchange stream waits for the value.  NULL (ZRK_EXCHANGE_EVENTS=1): an event per tick on
    // the compute stream instead
    uint32_t *flag = nullptr;
    uint32_t seq = 0;
    // ... and what that kernel does when the value does not come (k_wait_flag): it raises this word of PINNED HOST memory,
    // which the host reads without a synchronisation at the end of every zrk_run_ticks_x call and in zrk_exchange_sync, and
    // poisons the list it was waiting for (count -1) -- as it does with every later list of this exchange, so that the peers
    // learn it from the wire
    volatile uint32_t *gave_up = nullptr;
    uint32_t *gave_up_dev = nullptr;
    int wait_spins = 1 << 20;                          // looks (s_sleep 32 between them) before it gives up: ~1-2 s
    // one helper thread per rank: the side stream's thread issues the collectives (SideItem::post_x); set for the duration of
    // a call, with the number of the side ring's item that carries each slot's collective
    struct Side *via_side = nullptr;
    uint64_t side_item_no[ZRK_EXCHANGE_SLOTS] = {};
    bool one_helper = false;
    bool wait_in_stream = false;                        // ZRK_EXCHANGE_WAIT_IN_STREAM=1, see zrk_exchange_wait
    // ZRK_EXCHANGE_ALGO=direct: every rank sends its list straight to each of its world - 1 peers (grouped ncclSend / ncclRecv:
    // one hop over the peer's own xGMI link, all links at once) instead of ncclAllGather, whose ring passes every list through
    // world - 1 links one after the other.  Default: ncclAllGather, as RCCL chooses to run it
    bool direct = false;
    bool group_pairs = true;                            // a pair launch's two collectives in one RCCL group (ZRK_EXCHANGE_GROUP=0: not)
    // self-check for the first multi-rank record: what the communicator says its size is, how often and how long the calling
    // thread had to wait for a collective posted ZRK_EXCHANGE_SLOTS ticks before (zrk_exchange_info)
    std::atomic<int64_t> waits{0}, wait_ns{0}, collectives{0};
    // ... and those collectives are issued by a thread of the exchange's own: waiting for the value, the RCCL call and the
    // event record take the calling thread longer than the two launches of a tick, and the device would wait for its host
    // (slot2 >= 0: the two ticks of a pair launch, whose lists one compaction launch hands over: ONE wait, the two
    // collectives in one RCCL group -- one launch on the exchange stream instead of two)
    struct PostItem { int slot; const int64_t *send; int64_t *recv; int64_t words; uint32_t value;
                      int slot2 = -1; const int64_t *send2 = nullptr; int64_t *recv2 = nullptr; };
    static constexpr uint64_t kRing = 8;
    PostItem ring[kRing];
    std::atomic<uint64_t> head{0}, tail{0};             // items handed to the thread / items it has issued
    uint64_t item_no[ZRK_EXCHANGE_SLOTS] = {};          // per slot: `head` after its last item went in
    std::thread poster;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<bool> asleep{false}, stop{false};
    std::atomic<int> post_rc{0};
    std::string post_err;                               // written by the thread before post_rc, read after
    std::string err;
};

namespace {

int exchange_post_behind_flag(zrk_exchange *x, int slot, const int64_t *send, int64_t *recv, int64_t words, uint32_t value,
                              int slot2 = -1, const int64_t *send2 = nullptr, int64_t *recv2 = nullptr);
int exchange_collective(zrk_exchange *x, const int64_t *send, int64_t *recv, int64_t words);
__global__ void k_poison_if_gave_up(const uint32_t *gave_up, int64_t *send);

// the wait kernel of some collective gave up (k_wait_flag): that list and every later one went out poisoned
int exchange_gave_up(zrk_exchange *x)
{
    if (x->gave_up && *x->gave_up != 0u) {
        x->err = "the exchange stream gave up waiting for a list to be handed over (k_wait_flag): the collective went out with a poisoned "
                 "list (count -1), as does every later one of this exchange";
        return ZRK_E_STATE;
    }
    return 0;
}

// How long the library's helper threads (an exchange's, a context's side stream's) keep spinning after their last
// item before they sleep on their condition variable.  Inside a call the next item is never more than a tick away, so
// they spin only there; between calls they sleep (eight ranks with two spinning helpers each would otherwise hold
// sixteen cores of a node for nothing).  Waking one costs 50-100 us, but zrk_run_ticks wakes them at its entry, before
// its first launches, and a 20-tick call measures the same with ZRK_HELPER_IDLE_MS=0 (sleep at once) as with 500.
std::chrono::milliseconds helper_idle()
{
    static const int ms = [] { const char *v = std::getenv("ZRK_HELPER_IDLE_MS"); return v ? std::max(0, std::atoi(v)) : 1; }();
    return std::chrono::milliseconds(ms);
}
// ... and between the two a phase in which they stay runnable but give their core to whoever wants it (sched_yield between
// looks): ZRK_HELPER_YIELD_MS, default 5 (bench.py asks for 250: its warm-up and timed call are tens of milliseconds apart; an
// application that calls a few times a second would otherwise keep one or two cores per context at 100 % between its calls --
// eight ranks per node sixteen and more).  A thread that sleeps on its condition variable is woken at the next call's entry,
// and three times in some 150 runs of the driver's 20-tick command that took 3-12 ms instead of 50 us (the scheduler's
// slice: the calling thread, which spins while it waits for this one, had the core the wakee was put on): a 20-tick call
// of 0.45 ms then took 3-13.  Calls less than a quarter of a second apart -- bench.py's warm-up and timed call, any loop of
// calls -- now find the threads awake; spin_until yields as well once a wait has lasted 50 us.
std::chrono::milliseconds helper_yield()
{
    static const int ms = [] { const char *v = std::getenv("ZRK_HELPER_YIELD_MS"); return v ? std::max(0, std::atoi(v)) : 5; }();
    return std::chrono::milliseconds(ms);
}

void exchange_poster_main(zrk_exchange *x)
{
    if (hipSetDevice(x->device) != hipSuccess) { x->post_err = "hipSetDevice failed in the exchange thread"; x->post_rc.store(ZRK_E_HIP); }
    auto idle_since = std::chrono::steady_clock::now();
    for (;;) {
        const uint64_t t = x->tail.load(std::memory_order_relaxed);
        if (x->head.load(std::memory_order_acquire) != t) {
            const zrk_exchange::PostItem it = x->ring[t % zrk_exchange::kRing];
            if (x->post_rc.load() == 0 && exchange_post_behind_flag(x, it.slot, it.send, it.recv, it.words, it.value, it.slot2, it.send2, it.recv2) != 0) {
                x->post_err = x->err;
                x->post_rc.store(ZRK_E_HIP);
            }
            x->tail.store(t + 1, std::memory_order_release);
            idle_since = std::chrono::steady_clock::now();
            continue;
        }
        if (x->stop.load()) return;
        {
            const auto idle = std::chrono::steady_clock::now() - idle_since;
            if (idle < helper_idle()) { __builtin_ia32_pause(); continue; }
            if (idle < helper_idle() + helper_yield()) { sched_yield(); continue; }
        }
        std::unique_lock<std::mutex> lk(x->mu);         // nothing for a while: sleep (the bounded wait covers a lost wake-up)
        x->asleep.store(true);
        if (x->head.load(std::memory_order_acquire) == x->tail.load() && !x->stop.load())
            x->cv.wait_for(lk, std::chrono::milliseconds(20));
        x->asleep.store(false);
    }
}

int exchange_enqueue(zrk_exchange *x, const zrk_exchange::PostItem &it)
{
    const uint64_t h = x->head.load(std::memory_order_relaxed);
    if (!spin_until([&] { return h - x->tail.load(std::memory_order_acquire) < zrk_exchange::kRing; })) {
        x->err = "the exchange's thread did not take an item within the host wait limit (ZRK_HOST_WAIT_MS)";
        return ZRK_E_STATE;
    }
    x->ring[h % zrk_exchange::kRing] = it;
    x->head.store(h + 1, std::memory_order_release);
    x->item_no[it.slot] = h + 1;
    if (it.slot2 >= 0) x->item_no[it.slot2] = h + 1;
    if (x->asleep.load()) { std::lock_guard<std::mutex> lk(x->mu); x->cv.notify_one(); }
    return 0;
}

int side_issued_upto(Side *sd, uint64_t upto, std::string &err);

// everything handed to the thread has been issued on the exchange stream (0), or the thread's failure
int exchange_drain(zrk_exchange *x, uint64_t upto)
{
    if (!spin_until([&] { return x->tail.load(std::memory_order_acquire) >= upto; })) {
        x->err = "the exchange's thread did not issue its collectives within the host wait limit (ZRK_HOST_WAIT_MS)";
        return ZRK_E_STATE;
    }
    if (x->post_rc.load() != 0) { x->err = x->post_err; return x->post_rc.load(); }
    return 0;
}

// the collective of `slot` has been issued, whoever issues them in this call
int exchange_issued(zrk_exchange *x, int slot)
{
    if (x->via_side) return side_issued_upto(x->via_side, x->side_item_no[slot], x->err);
    return exchange_drain(x, x->item_no[slot]);
}

}  // namespace

ZRK_API int zrk_exchange_unique_id(const char *rccl_path, zrk_rccl_id *id)
{
    if (!id) return ZRK_E_INVALID;
    RcclApi api;
    std::string err;
    if (!load_rccl(rccl_path, api, err)) return ZRK_E_HIP;
    return api.GetUniqueId(id) == 0 ? 0 : ZRK_E_HIP;
}

ZRK_API int zrk_exchange_create(const char *rccl_path, const zrk_rccl_id *id, int world, int rank, int device,
                                zrk_exchange **out)
{
    if (!out) return ZRK_E_INVALID;
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return ZRK_E_INVALID;
    zrk_exchange *x = new zrk_exchange;
    x->world = world; x->rank = rank; x->device = device;
    *out = x;                                            // handed back also on failure, for zrk_exchange_last_error
    if (hipSetDevice(device) != hipSuccess) { x->err = "hipSetDevice failed"; return ZRK_E_HIP; }
    if (!load_rccl(rccl_path, x->api, x->err)) return ZRK_E_HIP;
    const int rc = x->api.CommInitRank(&x->comm, world, *id, rank);
    if (rc != 0) {
        x->err = std::string("ncclCommInitRank: ") + (x->api.GetErrorString ? x->api.GetErrorString(rc) : "error");
        x->comm = nullptr;
        return ZRK_E_HIP;
    }
    bool ok = hipStreamCreateWithFlags(&x->cstream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < ZRK_EXCHANGE_SLOTS && ok; ++k)
        ok = hipEventCreateWithFlags(&x->ready[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&x->done[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) { x->err = "stream / event creation failed"; return ZRK_E_HIP; }
    if (const char *algo = std::getenv("ZRK_EXCHANGE_ALGO")) {
        x->direct = std::strcmp(algo, "direct") == 0;
        if (x->direct && !(x->api.Send && x->api.Recv && x->api.GroupStart && x->api.GroupEnd)) { x->err = "ZRK_EXCHANGE_ALGO=direct: this RCCL has no ncclSend / ncclRecv"; return ZRK_E_HIP; }
    }
    { const char *v = std::getenv("ZRK_EXCHANGE_GROUP"); x->group_pairs = !(v && v[0] == '0') && x->api.GroupStart && x->api.GroupEnd; }
    const char *in_stream = std::getenv("ZRK_EXCHANGE_WAIT_IN_STREAM");
    x->wait_in_stream = in_stream && in_stream[0] == '1';
    const char *force_events = std::getenv("ZRK_EXCHANGE_EVENTS");
    if (!(force_events && force_events[0] == '1')) {
        if (hipMalloc((void **)&x->flag, 8) != hipSuccess || hipMemset(x->flag, 0, 8) != hipSuccess ||
            hipHostMalloc((void **)&x->gave_up, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void **)&x->gave_up_dev, (void *)x->gave_up, 0) != hipSuccess) {
            x->err = "the exchange's flag words could not be allocated"; return ZRK_E_HIP;
        }
        *x->gave_up = 0u;
    }
    if (const char *v = std::getenv("ZRK_WAIT_FLAG_SPINS")) x->wait_spins = std::max(0, std::atoi(v));   // (0: gives up at once -- tests)
    // Helper threads.  By default the collectives are issued by a thread of the exchange's own (the side stream of the
    // overlapped loop has another); where the rank has fewer than three host cores to itself the side stream's thread
    // issues them too (ZRK_HELPERS=1 / 2 forces either), and a rank alone on less than two cores ... still works, slower.
    x->one_helper = zrk_exchange_plan_helpers(world) == 1;
    const char *no_thread = std::getenv("ZRK_EXCHANGE_THREAD");
    if (x->flag && !x->one_helper && !(no_thread && no_thread[0] == '0')) x->poster = std::thread(exchange_poster_main, x);
    return 0;
}

// How many helper threads an exchange of `world` ranks on this host starts per rank (what zrk_exchange_create decides): 2 where
// the rank has three host cores or more to itself (the side stream's thread and a poster of the collectives), else 1 (the side
// stream's thread issues the collectives too); ZRK_HELPERS=1|2 forces either.  No device is touched.
ZRK_API int zrk_exchange_plan_helpers(int world)
{
    int helpers = usable_host_cores() / std::max(1, world) < 3 ? 1 : 2;
    if (const char *v = std::getenv("ZRK_HELPERS")) helpers = std::atoi(v) <= 1 ? 1 : 2;
    return helpers;
}

ZRK_API void zrk_exchange_destroy(zrk_exchange *x)
{
    if (!x) return;
    if (x->poster.joinable()) {
        x->stop.store(true);
        { std::lock_guard<std::mutex> lk(x->mu); x->cv.notify_one(); }
        x->poster.join();
    }
    if (x->cstream) (void)hipStreamSynchronize(x->cstream);
    if (x->comm) (void)x->api.CommDestroy(x->comm);
    for (int k = 0; k < ZRK_EXCHANGE_SLOTS; ++k) {
        if (x->ready[k]) (void)hipEventDestroy(x->ready[k]);
        if (x->done[k]) (void)hipEventDestroy(x->done[k]);
    }
    if (x->cstream) (void)hipStreamDestroy(x->cstream);
    if (x->flag) (void)hipFree(x->flag);
    if (x->gave_up) (void)hipHostFree((void *)x->gave_up);
    delete x;
}

ZRK_API const char *zrk_exchange_last_error(zrk_exchange *x) { return x ? x->err.c_str() : "null exchange"; }

namespace {
// One tick's collective on the exchange's stream: recv[g] = rank g's send, for every g.
int exchange_collective(zrk_exchange *x, const int64_t *send, int64_t *recv, int64_t words)
{
    x->collectives.fetch_add(1, std::memory_order_relaxed);
    auto why = [&](const char *what, int rc) { x->err = std::string(what) + ": " + (x->api.GetErrorString ? x->api.GetErrorString(rc) : "error"); return ZRK_E_HIP; };
    if (!x->direct) {
        const int rc = x->api.AllGather(send, recv, (size_t)words, kNcclInt64, x->comm, x->cstream);
        return rc != 0 ? why("ncclAllGather", rc) : 0;
    }
    // own list: a copy on the same stream; the others: one send and one receive per peer, grouped
    if (hipMemcpyAsync(recv + (int64_t)x->rank * words, send, sizeof(int64_t) * (size_t)words, hipMemcpyDeviceToDevice, x->cstream) != hipSuccess) {
        x->err = "the exchange's local copy failed"; return ZRK_E_HIP;
    }
    if (x->world == 1) return 0;
    int rc = x->api.GroupStart();
    for (int k = 1; k < x->world && rc == 0; ++k) {
        // (rank r talks to r + k and r - k in step k: every step pairs the ranks off differently, every link is used once)
        const int to = (x->rank + k) % x->world, from = (x->rank - k + x->world) % x->world;
        rc = x->api.Send(send, (size_t)words, kNcclInt64, to, x->comm, x->cstream);
        if (rc == 0) rc = x->api.Recv(recv + (int64_t)from * words, (size_t)words, kNcclInt64, from, x->comm, x->cstream);
    }
    const int rce = x->api.GroupEnd();
    if (rc != 0) return why("ncclSend / ncclRecv", rc);
    return rce != 0 ? why("ncclGroupEnd", rce) : 0;
}
}  // namespace

ZRK_API int zrk_exchange_all_gather(zrk_exchange *x, int slot, const int64_t *send, int64_t *recv, int64_t words, void *stream)
{
    if (!x || !x->comm || !send || !recv || words <= 0 || slot < 0 || slot >= ZRK_EXCHANGE_SLOTS) return ZRK_E_INVALID;
    if (int rc = exchange_gave_up(x)) return rc;                       // (no further list of this exchange passes for a tick's)
    if (int rc = exchange_drain(x, x->head.load())) return rc;         // behind whatever the exchange's thread still had to issue
    if (hipEventRecord(x->ready[slot], (hipStream_t)stream) != hipSuccess ||
        hipStreamWaitEvent(x->cstream, x->ready[slot], 0) != hipSuccess) { x->err = "event hand-over to the exchange stream failed"; return ZRK_E_HIP; }
    if (x->gave_up_dev) {                                // (a give-up the host has not seen yet: this list goes out poisoned too)
        hipLaunchKernelGGL(k_poison_if_gave_up, dim3(1), dim3(1), 0, x->cstream, x->gave_up_dev, (int64_t *)send);
        if (hipGetLastError() != hipSuccess) { x->err = "k_poison_if_gave_up launch failed"; return ZRK_E_HIP; }
    }
    if (int rc = exchange_collective(x, send, recv, words)) return rc;
    if (hipEventRecord(x->done[slot], x->cstream) != hipSuccess) { x->err = "hipEventRecord failed"; return ZRK_E_HIP; }
    x->posted[slot] = true;
    return 0;
}

namespace {

__global__ void k_raise_flag_system(uint32_t *flag, uint32_t value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_raise_flag(uint32_t *flag, uint32_t value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One lane waits for the word to reach `value` (sleeping between looks).  It does not hold the device for ever: after
// `spins` looks (~1-2 s) it gives up -- and then the collective behind it must not pass for a tick's list: the word in
// pinned host memory goes up (the host fails the call: zrk_run_ticks_x, zrk_exchange_sync) and the list's count is
// overwritten with -1, which every decoder rejects (exchange.decode_union_bits / decode_events / overflowed).  The
// late compaction may still write the real count over the poison of THIS list before the collective reads it, so every
// later list of the exchange is poisoned as well (the word stays up), behind its compaction: the peers find out from
// the wire at the latest one tick on.  (A profiler that serialises kernels across streams -- rocprofv3 --pmc -- makes the
// producer on the other stream wait for this kernel: the give-up is then certain.  Do not collect counters on the
// exchange path.)
__global__ void k_wait_flag(const uint32_t *flag, uint32_t value, uint32_t *gave_up, int64_t *send, int spins, int64_t *send2)
{
    bool up = false;
    // (once it has given up, the later lists are poisoned BEHIND their compaction: those waits take the full default)
    if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) spins = 1 << 20;
    for (int k = 0; k < spins && !up; ++k) {
        up = (int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - value) >= 0;
        if (!up) __builtin_amdgcn_s_sleep(32);
    }
    if (!up) __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!up || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
        __hip_atomic_store(send, (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (send2) __hip_atomic_store(send2, (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a pair's second list)
    }
}

// The same for a list that is handed over by an event (the last tick of a call): poisoned if the exchange has given up before.
__global__ void k_poison_if_gave_up(const uint32_t *gave_up, int64_t *send)
{
    if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
        __hip_atomic_store(send, (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The collective of a list whose producer has no event behind it: the exchange stream waits until the flag word
// reaches `value`, which a kernel launched BEHIND the producer on the compute stream writes as it starts.
int exchange_post_behind_flag(zrk_exchange *x, int slot, const int64_t *send, int64_t *recv, int64_t words, uint32_t value,
                              int slot2, const int64_t *send2, int64_t *recv2)
{
    const bool pair = slot2 >= 0 && send2 && recv2;
    // (a wait kernel of our own on a word of device memory: hipStreamWaitValue32 on signal memory does the same
    // job but cost the compute stream 3.5 us a tick in the measurement, this costs it nothing measurable)
    hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(1), 0, x->cstream, x->flag, value, x->gave_up_dev, (int64_t *)send, x->wait_spins,
                       pair ? (int64_t *)send2 : (int64_t *)nullptr);
    if (hipGetLastError() != hipSuccess) { x->err = "k_wait_flag launch failed"; return ZRK_E_HIP; }
    if (pair) {
        // the two ticks of a pair launch: both collectives in ONE group -- RCCL issues a group's operations as one launch
        // (ZRK_EXCHANGE_GROUP=0: one after the other, as two single ticks would go)
        int rcg = x->group_pairs ? x->api.GroupStart() : 0;
        int rc = rcg != 0 ? ZRK_E_HIP : exchange_collective(x, send, recv, words);
        if (rc == 0) rc = exchange_collective(x, send2, recv2, words);
        if (x->group_pairs && rcg == 0 && x->api.GroupEnd() != 0 && rc == 0) { x->err = "ncclGroupEnd failed"; rc = ZRK_E_HIP; }
        if (rcg != 0) x->err = "ncclGroupStart failed";
        if (rc != 0) return rc;
    } else if (int rc = exchange_collective(x, send, recv, words)) return rc;
    if (hipEventRecord(x->done[slot], x->cstream) != hipSuccess) { x->err = "hipEventRecord failed"; return ZRK_E_HIP; }
    x->posted[slot] = true;
    if (pair) {
        if (hipEventRecord(x->done[slot2], x->cstream) != hipSuccess) { x->err = "hipEventRecord failed"; return ZRK_E_HIP; }
        x->posted[slot2] = true;
    }
    return 0;
}

}  // namespace

ZRK_API int zrk_exchange_wait(zrk_exchange *x, int slot, void *stream)
{
    if (!x || slot < 0 || slot >= ZRK_EXCHANGE_SLOTS) return ZRK_E_INVALID;
    if (int rc = exchange_issued(x, slot)) return rc;                  // its collective has been issued (by a helper thread)
    if (!x->posted[slot]) return 0;
    // usually that collective is long over (it was posted ZRK_EXCHANGE_SLOTS ticks ago): then the host knows, and the stream is spared a
    // barrier packet, which costs it more than the wait it would do
    if (hipEventQuery(x->done[slot]) == hipSuccess) return 0;
    // not yet -- mostly because the host runs ticks ahead of the device.  Waiting HERE (the host is then at most two ticks
    // ahead, which still keeps a launch queued behind the running one) is cheaper than a wait in the stream: that is a
    // barrier packet in front of every compaction, ~5 us of idle device each
    if (x->wait_in_stream) {
        if (hipStreamWaitEvent((hipStream_t)stream, x->done[slot], 0) != hipSuccess) { x->err = "hipStreamWaitEvent failed"; return ZRK_E_HIP; }
        return 0;
    }
    hipError_t q = hipErrorNotReady;
    const auto t_wait = std::chrono::steady_clock::now();
    const bool came = spin_until([&] { q = hipEventQuery(x->done[slot]); return q != hipErrorNotReady; });
    x->waits.fetch_add(1, std::memory_order_relaxed);
    x->wait_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_wait).count(), std::memory_order_relaxed);
    if (!came) {
        x->err = "a collective posted ZRK_EXCHANGE_SLOTS ticks ago is not through within the host wait limit (ZRK_HOST_WAIT_MS): a peer rank is gone or stuck";
        return ZRK_E_STATE;
    }
    if (q != hipSuccess) { x->err = "hipEventQuery failed"; return ZRK_E_HIP; }
    return 0;
}

ZRK_API int zrk_exchange_info(zrk_exchange *x, zrk_exchange_stats *out)
{
    if (!x || !out) return ZRK_E_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->world = x->world; out->rank = x->rank; out->direct = x->direct ? 1 : 0; out->helper_threads = x->one_helper ? 1 : 2;
    out->grouped_pairs = x->group_pairs ? 1 : 0;
    out->comm_ranks = -1;
    if (x->comm && x->api.CommCount) { int c = -1; if (x->api.CommCount(x->comm, &c) == 0) out->comm_ranks = c; }
    out->collectives = x->collectives.load(); out->host_waits = x->waits.load(); out->host_wait_us = (double)x->wait_ns.load() * 1e-3;
    return 0;
}

ZRK_API int zrk_exchange_sync(zrk_exchange *x)
{
    if (!x) return ZRK_E_INVALID;
    if (int rc = exchange_drain(x, x->head.load())) return rc;
    if (hipStreamSynchronize(x->cstream) != hipSuccess) { x->err = "hipStreamSynchronize failed"; return ZRK_E_HIP; }
    return exchange_gave_up(x);
}


namespace {

// ---- overlap mode: the side stream and its thread --------------------------------------------------------------

__global__ void k_side_prime() {}

int side_issue(Side *sd, const SideItem &it)
{
    // The first launch on a stream that has been idle costs its thread 15-20 us instead of 4-5 (the runtime's own
    // bookkeeping for an idle queue).  This thread has nothing to do until the next sweep starts: it pays that now, with an
    // empty launch, instead of in front of the call's first compaction -- which everything behind it would then lag.
    {
        const auto now = std::chrono::steady_clock::now();
        if (!it.wait_event && !it.on_compute && now - sd->last_launch > std::chrono::microseconds(300)) {
            hipLaunchKernelGGL(k_side_prime, dim3(1), dim3(64), 0, it.stream);
            (void)hipGetLastError();
            g_trace.mark("side: primed");
        }
    }
    // The compaction is launched when its input is there, not before: letting its workgroups wait on the device -- resident
    // ahead of their input -- deadlocks the device as soon as anything else on it needs whole compute units in dispatch
    // order (e.g. another engine's single-launch compaction), and a one-lane wait kernel in front of it costs the side
    // stream 5 us a tick.  This thread has nothing else to do.
    if (it.on_compute == 2) {
        // in order behind the call's last sweep and independent of what the side stream still runs: an event behind that, for
        // the host (side_wait)
        sd->tail_ev_live = false;
        if (sd->side_busy && sd->last_side_slot[0] >= 0) {
            if (hipEventRecord(sd->tail_ev, sd->stream) != hipSuccess) { sd->err = "side stream: hipEventRecord failed"; return ZRK_E_HIP; }
            for (int q = 0; q < 2; ++q)
                if (sd->last_side_slot[q] >= 0) { sd->done_of[sd->last_side_slot[q]] = sd->tail_ev; sd->posted[sd->last_side_slot[q]] = true; }
            sd->tail_ev_live = true;
        }
    } else if (it.on_compute) {
        // in order behind the call's last sweep; what the side stream still runs (the compaction before this one) comes first
        if (sd->side_busy) {
            if (hipEventRecord(sd->tail_ev, sd->stream) != hipSuccess || hipStreamWaitEvent(it.stream, sd->tail_ev, 0) != hipSuccess) {
                sd->err = "side stream: the compute stream could not take it in (hipEventRecord / hipStreamWaitEvent)";
                return ZRK_E_HIP;
            }
            sd->side_busy = false;
        }
    } else if (it.wait_event) {
        if (hipStreamWaitEvent(it.stream, it.wait_event, 0) != hipSuccess) { sd->err = "side stream: hipStreamWaitEvent failed"; return ZRK_E_HIP; }
    } else if (!spin_until([&] { return (int32_t)(*sd->hflag - it.flag_value) >= 0 || sd->stop.load(); })) {
        // (the caller queued more work in front of the loop than the limit allows for, or the device is gone)
        sd->err = "side stream: the compute stream did not reach the next sweep within the host wait limit (ZRK_HOST_WAIT_MS)";
        return ZRK_E_STATE;
    }
    if (!it.wait_event && !it.on_compute && (int32_t)(*sd->hflag - it.flag_value) < 0) { sd->err = "side stream: stopped"; return ZRK_E_STATE; }
    g_trace.mark(it.on_compute ? "side: the call's last item, to the compute stream" : "side: flag seen");
    const auto t_issue = std::chrono::steady_clock::now();
    // (a launch that waited for nobody cannot say that the one before it is over)
    const DoneWord dw{it.on_compute == 2 ? nullptr : sd->hdone_dev, it.done_value - 1u};
    if (it.pair) {
        const int threads = it.pair_threads == 256 ? 256 : (it.pair_threads == 512 ? 512 : 1024);
        MarksArgs mk = it.marks;
        mk.blocks = mk.pend ? (int)((mk.n + (int64_t)threads * 16 - 1) / ((int64_t)threads * 16)) : 0;
        const dim3 grid(it.C.nb + (it.M.m > 0 ? 2 : 0) + mk.blocks);
        if (threads == 256)
            hipLaunchKernelGGL(k_compact_pair<256>, grid, dim3(256), 0, it.stream, it.C, it.C2, it.M, it.M2, it.rm, it.rm_cap, dw, mk);
        else if (threads == 512)
            hipLaunchKernelGGL(k_compact_pair<512>, grid, dim3(512), 0, it.stream, it.C, it.C2, it.M, it.M2, it.rm, it.rm_cap, dw, mk);
        else
            hipLaunchKernelGGL(k_compact_pair<1024>, grid, dim3(1024), 0, it.stream, it.C, it.C2, it.M, it.M2, it.rm, it.rm_cap, dw, mk);
    }
    else
        hipLaunchKernelGGL(k_compact_side, dim3(it.C.nb + (it.M.m > 0 ? 1 : 0)), dim3(kCompBlock), 0, it.stream, it.C, it.by_ticket, it.M, dw);
    // an exchange's collective (on the exchange's own stream) waits for this word: the list and its events are complete
    if (it.raise) hipLaunchKernelGGL(k_raise_flag, dim3(1), dim3(1), 0, it.stream, it.raise, it.raise_value);
    if (hipGetLastError() != hipSuccess) { sd->err = "side stream: compaction launch failed"; return ZRK_E_HIP; }
    if (it.record_event) {
        if (hipEventRecord(sd->done[it.done_slot], it.stream) != hipSuccess) { sd->err = "side stream: hipEventRecord failed"; return ZRK_E_HIP; }
        sd->done_of[it.done_slot] = sd->done[it.done_slot];
        sd->posted[it.done_slot] = true;
        if (it.pair) { sd->done_of[it.done_slot2] = sd->done[it.done_slot]; sd->posted[it.done_slot2] = true; }
    } else {
        sd->posted[it.done_slot] = false;
        if (it.pair) sd->posted[it.done_slot2] = false;
    }
    // The side stream's launch that a free tail did not wait for is taken in by the compute stream BEHIND the tail (it has been
    // over for a while when that packet is reached: it costs the next launch nothing), so that everything of the call -- also
    // what only read the caller's tables or used the caller's workspace -- is the caller's stream's when the call returns.
    // (From this thread, which has nothing else to do: the call takes the calling thread tens of microseconds.)
    if (it.on_compute == 2 && sd->tail_ev_live && hipStreamWaitEvent(it.stream, sd->tail_ev, 0) != hipSuccess) {
        sd->err = "side stream: hipStreamWaitEvent failed"; return ZRK_E_HIP;
    }
    if (!it.on_compute) {
        sd->last_launch = std::chrono::steady_clock::now(); sd->side_busy = true;
        sd->last_side_slot[0] = it.done_slot; sd->last_side_slot[1] = it.pair ? it.done_slot2 : -1;
    }
    if (stall_us() > 0) stall_report(t_issue, __LINE__, "the side stream's thread issuing an item");
    g_trace.mark("side: compaction issued");
    if (it.post_x && exchange_post_behind_flag(it.post_x, it.post_slot, it.post_send, it.post_recv, it.post_words, it.raise_value,
                                                it.post2_send ? it.post2_slot : -1, it.post2_send, it.post2_recv) != 0) {
        sd->err = std::string("side stream: ") + it.post_x->err;
        return ZRK_E_HIP;
    }
    return 0;
}

void side_main(Side *sd, int device)
{
    if (hipSetDevice(device) != hipSuccess) { sd->err = "hipSetDevice failed in the side stream's thread"; sd->rc.store(ZRK_E_HIP); }
    auto idle_since = std::chrono::steady_clock::now();
    for (;;) {
        const uint64_t t = sd->tail.load(std::memory_order_relaxed);
        if (sd->head.load(std::memory_order_acquire) != t) {
            const SideItem it = sd->ring[t % Side::kRing];
            g_trace.mark("side: item taken");
            if (sd->rc.load() == 0) { const int rc = side_issue(sd, it); if (rc != 0) sd->rc.store(rc); }
            sd->tail.store(t + 1, std::memory_order_release);
            // Behind a call's last item (it went to the caller's stream; nothing else is waiting) this thread finishes the side
            // stream itself -- its last launch is over before the call's last sweep is -- so that the runtime knows the stream to be
            // empty: a caller that synchronises the DEVICE behind the call otherwise pays for a marker on this stream too
            // (hipDeviceSynchronize 16 us instead of 5 behind a 20-tick call; ZRK_SIDE_SETTLE=0: as before)
            static const bool settle = [] { const char *v = std::getenv("ZRK_SIDE_SETTLE"); return !(v && v[0] == '0'); }();
            if (settle && it.on_compute && sd->head.load(std::memory_order_acquire) == t + 1) (void)hipStreamSynchronize(sd->stream);
            idle_since = std::chrono::steady_clock::now();
            continue;
        }
        if (sd->stop.load()) return;
        {
            const auto idle = std::chrono::steady_clock::now() - idle_since;
            if (idle < helper_idle()) { __builtin_ia32_pause(); continue; }
            if (idle < helper_idle() + helper_yield()) { sched_yield(); continue; }
        }
        std::unique_lock<std::mutex> lk(sd->mu);
        sd->asleep.store(true);
        bool woken = false;
        if (sd->head.load(std::memory_order_acquire) == sd->tail.load() && !sd->stop.load())
            woken = sd->cv.wait_for(lk, std::chrono::milliseconds(20)) == std::cv_status::no_timeout;
        sd->asleep.store(false);
        if (woken) g_trace.mark("side: woken");
        if (woken) idle_since = std::chrono::steady_clock::now();      // somebody is about to hand over work: stay up
    }
}

Side *side_of(zrk_ctx *ctx)
{
    if (ctx->side) return ctx->side;
    Side *sd = new Side;
    // ZRK_SIDE_CUS=n[,first]: confine the side stream to n compute units (mask bits first .. first + n - 1; on this part
    // consecutive bits fall on different XCDs), so that the compaction beside a sweep takes whole compute units from it
    // instead of wave slots on every one
    bool made = false;
    if (const char *v = std::getenv("ZRK_SIDE_CUS")) {
        int n = 0, first = 0;
        if (std::sscanf(v, "%d,%d", &n, &first) >= 1 && n > 0 && first >= 0 && first + n <= 512) {
            uint32_t mask[16] = {};
            for (int b = first; b < first + n; ++b) mask[b / 32] |= 1u << (b % 32);
            const uint32_t words = (uint32_t)((std::max(ctx->cus, first + n) + 31) / 32);
            made = hipExtStreamCreateWithCUMask(&sd->stream, words, mask) == hipSuccess;
            if (made) sd->cu_count = n; else (void)hipGetLastError();
        }
    }
    if (!made) {
        // ZRK_SIDE_PRIORITY=high|low: the side stream's place among the device's queues
        if (const char *v = std::getenv("ZRK_SIDE_PRIORITY")) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess)
                made = hipStreamCreateWithPriority(&sd->stream, hipStreamNonBlocking, v[0] == 'h' ? hi : lo) == hipSuccess;
            if (!made) (void)hipGetLastError();
        }
    }
    bool ok = (made || hipStreamCreateWithFlags(&sd->stream, hipStreamNonBlocking) == hipSuccess) &&
              hipEventCreateWithFlags(&sd->last_sweep, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&sd->tail_ev, hipEventDisableTiming) == hipSuccess &&
              hipMalloc((void **)&sd->bar, 64) == hipSuccess && hipMemset(sd->bar, 0, 64) == hipSuccess &&
              hipHostMalloc((void **)&sd->hflag, 128, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer((void **)&sd->hflag_dev, (void *)sd->hflag, 0) == hipSuccess;
    if (ok) { *sd->hflag = 0u; sd->hdone = sd->hflag + 16; sd->hdone_dev = sd->hflag_dev + 16; *sd->hdone = 0u; }
    for (int k = 0; k <= Side::kMasks && ok; ++k) ok = hipEventCreateWithFlags(&sd->done[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (int k = 0; k <= Side::kMasks; ++k) if (sd->done[k]) (void)hipEventDestroy(sd->done[k]);
        if (sd->last_sweep) (void)hipEventDestroy(sd->last_sweep);
        if (sd->tail_ev) (void)hipEventDestroy(sd->tail_ev);
        if (sd->bar) (void)hipFree(sd->bar);
        if (sd->hflag) (void)hipHostFree((void *)sd->hflag);
        if (sd->stream) (void)hipStreamDestroy(sd->stream);
        delete sd;
        return nullptr;
    }
    sd->worker = std::thread(side_main, sd, ctx->device);
    ctx->side = sd;
    return sd;
}

int side_enqueue(zrk_ctx *ctx, Side *sd, const SideItem &it)
{
    const uint64_t h = sd->head.load(std::memory_order_relaxed);
    if (!spin_until([&] { return h - sd->tail.load(std::memory_order_acquire) < Side::kRing; }))
        return fail(ctx, ZRK_E_STATE, "side stream: its thread did not take an item within the host wait limit (ZRK_HOST_WAIT_MS)");
    sd->ring[h % Side::kRing] = it;
    sd->ring[h % Side::kRing].done_value = (uint32_t)(h + 1);
    sd->head.store(h + 1, std::memory_order_release);
    sd->item_no[it.done_slot] = h + 1;
    if (it.pair) sd->item_no[it.done_slot2] = h + 1;
    if (sd->asleep.load()) { std::lock_guard<std::mutex> lk(sd->mu); sd->cv.notify_one(); }
    return 0;
}

// Everything up to item number `upto` has been issued (0), or the thread's failure.
int side_issued_upto(Side *sd, uint64_t upto, std::string &err)
{
    // (twice the limit: the thread itself may be sitting out one limit in side_issue, and reports that)
    if (!spin_until([&] { return sd->tail.load(std::memory_order_acquire) >= upto; }, 2 * host_wait_limit())) {
        err = "side stream: its thread did not issue its work within the host wait limit (ZRK_HOST_WAIT_MS)";
        return ZRK_E_STATE;
    }
    if (sd->rc.load() != 0) { err = sd->err; return sd->rc.load(); }
    return 0;
}

int side_drain(zrk_ctx *ctx, Side *sd, uint64_t upto)
{
    std::string err;
    if (int rc = side_issued_upto(sd, upto, err)) return fail(ctx, rc, err);
    return 0;
}

// The compaction that last used mask buffer `slot` is over (the host waits: see zrk_exchange_wait for why not the stream).
int side_wait(zrk_ctx *ctx, Side *sd, int slot, hipStream_t compute)
{
    // (work of earlier calls: the compute stream took the side stream in when that call returned, whatever is launched
    // on it now is behind that -- nothing to ask the runtime, whose first answer after a pause takes 10 us)
    if (sd->item_no[slot] == 0) return 0;                    // (never used, or started afresh after a failure)
    if (sd->item_no[slot] <= sd->joined_upto && compute == sd->joined_stream) return 0;
    if (sd->item_no[slot] == sd->tail_item && compute == sd->tail_stream) return 0;     // (a call's last item, launched on this very stream)
    // the word in which the NEXT compaction of the side stream says that this one is over (the next one is always in the
    // thread's hands by now -- the ring is several items deep -- and starts without this thread's help)
    const uint32_t want = (uint32_t)sd->item_no[slot];
    if ((int32_t)(*sd->hdone - want) >= 0) return 0;
    if (int rc = side_drain(ctx, sd, sd->item_no[slot])) return rc;
    if (sd->posted[slot]) {                                  // a call's last item: no launch follows it, an event does
        hipError_t q = hipErrorNotReady;
        if (!spin_until([&] { q = hipEventQuery(sd->done_of[slot]); return q != hipErrorNotReady; }))
            return fail(ctx, ZRK_E_STATE, "side stream: a compaction launched ticks ago is not through within the host wait limit (ZRK_HOST_WAIT_MS)");
        return q == hipSuccess ? 0 : fail(ctx, ZRK_E_HIP, "side stream: hipEventQuery failed");
    }
    if (!spin_until([&] { return (int32_t)(*sd->hdone - want) >= 0 || sd->rc.load() != 0; }))
        return fail(ctx, ZRK_E_STATE, "side stream: a compaction launched ticks ago is not through within the host wait limit (ZRK_HOST_WAIT_MS)");
    if (sd->rc.load() != 0) return fail(ctx, sd->rc.load(), sd->err);
    return 0;
}

void side_destroy(Side *sd)
{
    if (!sd) return;
    if (sd->worker.joinable()) {
        sd->stop.store(true);
        { std::lock_guard<std::mutex> lk(sd->mu); sd->cv.notify_one(); }
        sd->worker.join();
    }
    if (sd->stream) { (void)hipStreamSynchronize(sd->stream); (void)hipStreamDestroy(sd->stream); }
    for (int k = 0; k <= Side::kMasks; ++k) if (sd->done[k]) (void)hipEventDestroy(sd->done[k]);
    if (sd->last_sweep) (void)hipEventDestroy(sd->last_sweep);
    if (sd->tail_ev) (void)hipEventDestroy(sd->tail_ev);
    if (sd->tail_ws) (void)hipFree(sd->tail_ws);
    for (int k = 0; k < Side::kMasks; ++k) if (sd->masks[k]) (void)hipFree(sd->masks[k]);
    for (int k = 0; k < Side::kMasks; ++k) if (sd->codes[k]) (void)hipFree(sd->codes[k]);
    if (sd->pend) (void)hipFree(sd->pend);
    if (sd->bar) (void)hipFree(sd->bar);
    for (int k = 0; k <= Side::kMasks; ++k) if (sd->rm[k]) (void)hipFree(sd->rm[k]);
    if (sd->scratch_det) (void)hipFree(sd->scratch_det);
    if (sd->scratch_cnt) (void)hipFree(sd->scratch_cnt);
    if (sd->scratch_ev) (void)hipFree(sd->scratch_ev);
    if (sd->scratch_packed) (void)hipFree(sd->scratch_packed);
    if (sd->hflag) (void)hipHostFree((void *)sd->hflag);
    delete sd;
}

// Timing events of zrk_run_ticks: created once per context and reused (the multi-rank loop calls with K = 1).
bool ensure_events(zrk_ctx *ctx, int pairs)
{
    while ((int)ctx->tev.size() < 2 * pairs) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return false;
        ctx->tev.push_back(e);
    }
    if ((int)ctx->tev_alias.size() < pairs) { ctx->tev_alias.resize(pairs); ctx->tev_ticks.resize(pairs); }
    for (int k = 0; k < pairs; ++k) { ctx->tev_alias[k] = k; ctx->tev_ticks[k] = 1; }
    return true;
}

}  // namespace

namespace {

int run_ticks(zrk_ctx *ctx, const zrk_entities *e, const zrk_missiles *mis, int64_t m, zrk_loop *st,
              zrk_radar *radars, const zrk_scan *scan, int R, void *workspace, int32_t *det_idx,
              int64_t det_stride, int32_t *det_cnt, int64_t *packed, int64_t packed_capacity,
              const zrk_exchange_io *xio, const zrk_ensemble *ens, int K, float *sweep_ms, int prof_stride, void *stream)
{
    if (!ctx || !e || !st || !workspace) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks: null argument");
    if (m > 0 && !mis) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks: missiles without a table");
    if (K < 0 || (st->cur != 0 && st->cur != 1)) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks: K/cur out of range");
    if (xio) {
        if (packed) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks: `packed` and an exchange are alternatives");
        bool buffers = true;
        for (int k = 0; k < ZRK_EXCHANGE_SLOTS; ++k) buffers = buffers && xio->send[k] && xio->recv[k];
        if (!xio->x || !buffers || xio->ev_capacity < 0 ||
            xio->words < 3 + xio->ev_capacity + (xio->ev_capacity > 0 ? 1 : 0))
            return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks: incomplete exchange description");
        if (!(st->flags & ZRK_F_UNION_BITS)) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks: the exchange carries the wire format (ZRK_F_UNION_BITS)");
    }
    hipStream_t s = (hipStream_t)stream;
    if (g_trace.deferred) g_trace.dump();
    g_trace.mark("run_ticks: entry");
    // a batched ensemble: S scenarios of rows_per_scenario rows each, radars and scan state on the device
    EnsLaunch EL;
    std::memset(&EL, 0, sizeof(EL));
    RadarBlock *ens_tables[2] = {nullptr, nullptr};
    if (ens) {
        if (packed || xio) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks_ensemble: no union list for an ensemble");
        if (ens->scenarios < 1 || ens->radars < 0 || ens->radars > ZRK_MAX_RADARS || ens->rows_per_scenario < kCompBlock ||
            ens->rows_per_scenario % kCompBlock != 0 || !ens->radar_state || !ens->scan || !ens->d2_max || !ens->tables)
            return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks_ensemble: incomplete ensemble description");
        if (st->n != (int64_t)ens->scenarios * ens->rows_per_scenario || st->n > e->capacity || !e->list_index)
            return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks_ensemble: the table must hold scenarios x rows_per_scenario rows and a list_index column");
        if (ens->rows_per_scenario / ZRK_BLOCK >= 65536 || st->n / ZRK_BLOCK >= 65536 * (int64_t)16)
            return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks_ensemble: scenario too long");
        R = ens->radars;
        const int64_t granules = ens->rows_per_scenario / kCompBlock;
        int want = fused_items(ctx, st->n), items = 1;
        for (int k = 1; k <= std::min(want, kFusedMaxItems); ++k) if (granules % k == 0) items = k;
        while (st->n / ((int64_t)kCompBlock * items) > std::min(ctx->fused_max_blocks, kFusedMaxBlocks)) {
            int bigger = 0;
            for (int k = items + 1; k <= kFusedMaxItems; ++k) if (granules % k == 0) { bigger = k; break; }
            if (!bigger) return fail(ctx, ZRK_E_CAPACITY, "zrk_run_ticks_ensemble: too many rows for the single-launch compaction");
            items = bigger;
        }
        ens_tables[0] = (RadarBlock *)ens->tables;
        ens_tables[1] = ens_tables[0] + ens->scenarios;
        EL.seeds = ens->seeds; EL.rows_ps = ens->rows_per_scenario; EL.bps = (int32_t)(ens->rows_per_scenario / ZRK_BLOCK);
        EL.items = items;
        EL.next.state = ens->radar_state; EL.next.scan = ens->scan; EL.next.d2max = ens->d2_max;
        EL.next.S = ens->scenarios; EL.next.R = R; EL.next.flags = st->flags; EL.next.advance = 1;
        // the records of the first tick of this call, from the angles as they stand
        EnsembleArgs now = EL.next;
        now.advance = 0; now.table_out = ens_tables[st->tick & 1u];
        hipLaunchKernelGGL(k_ensemble_derive, dim3(std::max(1, nblocks((int64_t)now.S * now.R, kCompBlock))), dim3(kCompBlock), 0, s, now);
        if (int rc0 = check_launch(ctx, "k_ensemble_derive")) return rc0;
    }
    // prof_stride < 0: record the events only; zrk_read_sweep_ms collects the times later (keeps the synchronisation and
    // the reads out of a region the caller is timing)
    const bool deferred = prof_stride < 0;
    // one scenario with compaction every tick: the radar records go through device memory (two blocks in the workspace,
    // alternating); the first of this call by a launch of its own, the later ones by the previous tick's compaction
    PutArgs put;
    put.dst = nullptr;
    RadarBlock *rb_dev[2] = {nullptr, nullptr};
    const bool rbm_possible = !ens && R > 0 && radars && (det_idx || packed || xio) && st->n > 0 &&
                              compacts_in_one_launch(ctx, st->n) && K > 0;
    // Overlap mode (below) has no second launch on the compute stream for the records to ride in: there the sweep reads
    // them from its own argument segment (0.3 us slower than from memory, against 5.7 us for a launch of their own)
    const bool want_overlap = ctx->overlap > 0 && K >= ctx->overlap_min && st->n >= ctx->overlap_min_rows &&
                              (det_idx || packed || xio) && st->n > 0 && R > 0 && e->vis_mask_alt && (ens || rbm_possible) &&
                              (m == 0 || m <= 1024 * (int64_t)kMissileItems) && (!xio || xio->x->flag);
    const bool rb_through_memory = rbm_possible && !want_overlap;
    if (rb_through_memory) {
        Workspace w0 = carve(workspace, 0, e->capacity);
        rb_dev[0] = (RadarBlock *)((char *)w0.boxes - 2 * sizeof(RadarBlock));
        rb_dev[1] = rb_dev[0] + 1;
        fill_radar_block(ctx, radars, R, st->flags, put.rb);
        put.dst = (uint32_t *)rb_dev[st->tick & 1u];
        hipLaunchKernelGGL(k_put_radar_block, dim3(1), dim3(256), 0, s, put);
        if (int rc0 = check_launch(ctx, "k_put_radar_block")) return rc0;
    }
    const int stride = prof_stride > 0 ? prof_stride : (prof_stride < 0 ? -prof_stride : 1);
    const int n_prof = (sweep_ms || deferred) ? (K + stride - 1) / stride : 0;
    if (n_prof && !ensure_events(ctx, n_prof)) return fail(ctx, ZRK_E_HIP, "hipEventCreate");
    ctx->tev_pending = deferred ? n_prof : 0;
    hipEvent_t *ev = n_prof ? ctx->tev.data() : nullptr;
    // the tail of an exchanged list carries this tick's detonations: [count, rows ...]
    const int64_t ev_words = (xio && xio->ev_capacity > 0) ? 1 + (int64_t)xio->ev_capacity : 0;
    // radars of interest: what the other ranks' consumers read (zrk_exchange_io::interest; 0: every radar)
    const uint32_t wire_select = xio ? xio->interest : 0u;
    if (wire_select && det_idx) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks: radars of interest shape the exchanged list; per-radar lists (det_idx) are not available beside them");
    // hand-over of a tick's list to the exchange stream: by the flag the next tick's sweep raises (no packet of its own on
    // the compute stream), the last tick of the call -- which has no next sweep -- by an event
    // Overlap mode: the lists of tick t are compacted on a side stream beside the sweep of tick t + 1, and the compute
    // stream carries nothing but sweeps: removals travel as marks that the next sweep's threads carry out themselves
    // (SweepParams::pend), radar records in the sweep's arguments, the dispatch order is built by the sweep before.
    // (An ensemble keeps a small second launch, k_tick_small: its scenarios' scan steps and records, and its removals.)
    // For calls of a few ticks at least: the last tick's compaction has nothing to run beside.
    Side *sd = nullptr;
    if (want_overlap) {
        sd = side_of(ctx);
        if (!sd) return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the side stream could not be created");
        if (xio) { if (int rcx = exchange_drain(xio->x, xio->x->head.load())) return fail(ctx, rcx, zrk_exchange_last_error(xio->x)); }
        if (sd->rc.load() != 0) {
            // Its thread failed in an earlier call (reported then: that call returned the error).  The failure is not
            // sticky: the thread skips what it still holds, both streams are drained, the hand-over word, the ring slots and
            // the buffers start afresh -- and this call runs overlapped again.
            std::string ignored;
            (void)side_issued_upto(sd, sd->head.load(), ignored);
            if (hipStreamSynchronize(sd->stream) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
                return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the side stream did not drain after its thread's failure");
            *sd->hflag = 0u; sd->seq = 0;
            // (both streams are drained: the missile barrier's words start afresh with the rest)
            if (hipMemset(sd->bar, 0, 64) != hipSuccess) return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the missile barrier's words could not be cleared");
            sd->bar_epoch = 0;
            sd->masks_dirty = true; sd->pend_rows = 0; sd->joined_upto = 0; sd->joined_stream = nullptr;
            sd->err.clear();
            sd->rc.store(0);
        }
        // (its thread sleeps after a millisecond without work: wake it now, not at the first item two launches from here)
        if (sd->asleep.load()) { std::lock_guard<std::mutex> lk(sd->mu); sd->cv.notify_one(); g_trace.mark("run_ticks: side thread notified"); }
        if (xio && xio->x->asleep.load()) { std::lock_guard<std::mutex> lk(xio->x->mu); xio->x->cv.notify_one(); }
        if (sd->mask_rows < e->capacity || sd->masks_dirty) {
            if (int rc0 = side_drain(ctx, sd, sd->head.load())) return rc0;
            if (hipStreamSynchronize(sd->stream) != hipSuccess) return fail(ctx, ZRK_E_HIP, "hipStreamSynchronize");
            for (int k = 0; k < Side::kMasks; ++k) {
                if (sd->mask_rows < e->capacity) {
                    if (sd->masks[k]) (void)hipFree(sd->masks[k]);
                    sd->masks[k] = nullptr;
                    if (hipMalloc((void **)&sd->masks[k], sizeof(uint32_t) * (size_t)e->capacity) != hipSuccess)
                        return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the side stream's mask buffers could not be allocated");
                }
                if (hipMemsetAsync(sd->masks[k], 0, sizeof(uint32_t) * (size_t)e->capacity, s) != hipSuccess)
                    return fail(ctx, ZRK_E_HIP, "memset masks");
            }
            sd->mask_rows = e->capacity; sd->masks_dirty = false;
            for (int k = 0; k <= Side::kMasks; ++k) { sd->posted[k] = false; sd->item_no[k] = 0; }
        }
        if (!sd->tail_ws && ctx->tail_on_compute && ctx->tail_free && !ens && !xio) {
            // (the control words and records of a call's last compaction, Side::tail_ws: with the first overlapped call, not
            // in front of the first launch that needs them)
            if (hipMalloc(&sd->tail_ws, (size_t)kFusedBytes) != hipSuccess || hipMemsetAsync(sd->tail_ws, 0, (size_t)kFusedBytes, s) != hipSuccess) {
                (void)hipGetLastError();
                if (sd->tail_ws) (void)hipFree(sd->tail_ws);
                sd->tail_ws = nullptr;               // (then: the event)
            }
        }
        if (sd->pend_rows < e->capacity) {
            if (sd->pend) (void)hipFree(sd->pend);
            sd->pend = nullptr; sd->pend_rows = 0;
            // (a multiple of sixteen bytes past the capacity: MarksArgs reads the marks sixteen at a time)
            const size_t pend_bytes = ((size_t)e->capacity + 31) & ~(size_t)15;
            if (hipMalloc((void **)&sd->pend, pend_bytes) != hipSuccess ||
                hipMemsetAsync(sd->pend, 0, pend_bytes, s) != hipSuccess)
                return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the removal marks could not be allocated");
            sd->pend_rows = e->capacity;
        }
        if (m > sd->code_rows) {
            if (int rc0 = side_drain(ctx, sd, sd->head.load())) return rc0;
            if (hipStreamSynchronize(sd->stream) != hipSuccess) return fail(ctx, ZRK_E_HIP, "hipStreamSynchronize");
            const int64_t rows = std::max<int64_t>(2 * m, 4096);
            for (int k = 0; k < Side::kMasks; ++k) {
                if (sd->codes[k]) (void)hipFree(sd->codes[k]);
                sd->codes[k] = nullptr;
                if (hipMalloc((void **)&sd->codes[k], (size_t)rows) != hipSuccess)
                    return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the side stream's event-code buffers could not be allocated");
            }
            // (with them: the lists of rows a pair's first tick removes, one per ring slot, and where that tick's events go)
            const size_t rm_bytes = sizeof(int32_t) * (size_t)(1 + 2 * rows);
            for (int k = 0; k <= Side::kMasks; ++k) {
                if (sd->rm[k]) (void)hipFree(sd->rm[k]);
                sd->rm[k] = nullptr;
                if (hipMalloc((void **)&sd->rm[k], rm_bytes) != hipSuccess || hipMemsetAsync(sd->rm[k], 0, rm_bytes, s) != hipSuccess)
                    return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the removed-row lists could not be allocated");
            }
            sd->rm_cap = (int)(2 * rows);
            if (sd->scratch_ev) (void)hipFree(sd->scratch_ev);
            sd->scratch_ev = nullptr;
            if (hipMalloc((void **)&sd->scratch_ev, sizeof(int32_t) * (size_t)(2 * rows + 16)) != hipSuccess)
                return fail(ctx, ZRK_E_HIP, "zrk_run_ticks: the scratch event list could not be allocated");
            sd->scratch_ev_rows = rows;
            sd->code_rows = rows;
        }
        if (sd->seq > 0x7FFF0000u) {                     // far from wrapping: the comparison is on 32 bits
            if (int rc0 = side_drain(ctx, sd, sd->head.load())) return rc0;
            if (hipStreamSynchronize(sd->stream) != hipSuccess) return fail(ctx, ZRK_E_HIP, "hipStreamSynchronize");
            if (hipStreamSynchronize(s) != hipSuccess) return fail(ctx, ZRK_E_HIP, "hipStreamSynchronize");
            *sd->hflag = 0u;
            sd->seq = 0;
        }
    }
    ctx->last_overlapped = sd ? 1 : 0;
    g_trace.mark("run_ticks: side ready");
    hipStream_t side_stream = sd ? sd->stream : nullptr;
    int side_last = -1;
    bool tail_on_s = false;                              // the call's last compaction went to the compute stream (SideItem::on_compute)
    bool tail_alone = false;                             // ... and waited for nobody (on_compute == 2)
    bool marks_in_tail = false;                          // ... and carries out the call's removal marks (MarksArgs)
    zrk_exchange *fx = (xio && xio->x->flag) ? xio->x : nullptr;
    if (fx && fx->seq > 0x7FFF0000u) {                   // far from wrapping: the comparison is on 32 bits
        if (hipStreamSynchronize(fx->cstream) != hipSuccess) return fail(ctx, ZRK_E_HIP, "hipStreamSynchronize");
        hipLaunchKernelGGL(k_raise_flag, dim3(1), dim3(1), 0, s, fx->flag, 0u);
        if (hipStreamSynchronize(s) != hipSuccess) return fail(ctx, ZRK_E_HIP, "hipStreamSynchronize");
        fx->seq = 0;
    }
    // the missile phase's gather records (MissileArgs::grec): for tables whose missile phase rides in the sweep
    const double *grec = nullptr;
    if (ctx->grec_enabled && m > 0 && m <= 1024 * (int64_t)kMissileItems && st->n > 0 && R > 0 && (det_idx || packed || xio)) {
        if (ctx->grec_rows < e->capacity) {
            if (ctx->grec) (void)hipFree(ctx->grec);
            ctx->grec = nullptr; ctx->grec_rows = 0; ctx->grec_n = 0;
            if (hipMalloc((void **)&ctx->grec, sizeof(double) * 8 * (size_t)e->capacity) != hipSuccess) (void)hipGetLastError();   // (then: the columns)
            else ctx->grec_rows = e->capacity;
        }
        if (ctx->grec) {
            if (ctx->grec_key != (const void *)e->start_pos || st->n < ctx->grec_n) { ctx->grec_key = e->start_pos; ctx->grec_n = 0; }
            if (ctx->grec_n < st->n) {
                hipLaunchKernelGGL(k_build_gather_records, dim3(nblocks(st->n - ctx->grec_n, 256)), dim3(256), 0, s, e->start_pos, e->velocity,
                                   e->start_time, e->list_index, e->capacity, ctx->grec_n, st->n, ctx->grec);
                if (int rc0 = check_launch(ctx, "k_build_gather_records")) return rc0;
                ctx->grec_n = st->n;
            }
            grec = ctx->grec;
        }
    }
    struct { bool on; int slot; int64_t *list; uint32_t value; } pend = {false, 0, nullptr, 0u};
    struct { bool on; SideItem it; int slot; int64_t *list; } held;
    held.on = false;
    auto issue_held = [&]() -> int {
        held.on = false;
        zrk_exchange *x = xio->x;
        if (int rcw = zrk_exchange_wait(x, held.slot, stream)) return fail(ctx, rcw, zrk_exchange_last_error(x));
        if (x->one_helper) {                             // the side stream's thread issues the collective behind its launches
            held.it.post_x = x; held.it.post_slot = held.slot; held.it.post_send = held.list;
            held.it.post_recv = xio->recv[held.slot]; held.it.post_words = xio->words;
        }
        if (int rce = side_enqueue(ctx, sd, held.it)) return rce;
        if (x->one_helper) { x->via_side = sd; x->side_item_no[held.slot] = sd->head.load(); return 0; }
        if (x->poster.joinable()) {
            if (int rce = exchange_enqueue(x, zrk_exchange::PostItem{held.slot, held.list, xio->recv[held.slot], xio->words, held.it.raise_value}))
                return fail(ctx, rce, zrk_exchange_last_error(x));
        } else if (exchange_post_behind_flag(x, held.slot, held.list, xio->recv[held.slot], xio->words, held.it.raise_value) != 0)
            return fail(ctx, ZRK_E_HIP, zrk_exchange_last_error(x));
        return 0;
    };
    int rc = 0;
    bool marks_used = false;                             // some sweep of this call ran with removal marks
    // ---- the overlapped loop with TWO TICKS PER LAUNCH (SweepParams::t2) ---------------------------------------------------
    // One scenario, no exchange: launch L sweeps ticks t and t + 1 in one pass over the table, the side stream compacts both
    // ticks' lists (two launches, in order) beside launch L + 1.  An odd tick at the end of a call is a launch of one.
    const bool pairing = sd && !ens && ctx->pair_enabled && K >= 2 && (m == 0 || m <= 1024 * (int64_t)kMissileItems);
    ctx->last_ticks_per_launch = pairing ? 2 : 1;
    // the sweeps time themselves (zrk_sweep_stamps): every stamp_every-th launch of this call writes its waves' stamps into
    // the next slot of the ring (all of them, up to kStampSlots launches per call)
    int stamp_every = 0;
    ctx->stamp_used = 0;
    if (ctx->stamps_on && st->n > 0 && K > 0) {
        const int64_t words = kStampBegins + ((int64_t)nblocks(e->capacity, ZRK_BLOCK) + nblocks(std::max<int64_t>(m, 0), ZRK_BLOCK) + 8) * (ZRK_BLOCK / 64);
        if (ctx->stamp_slot_words < words) {
            if (ctx->stamp_ring) (void)hipFree(ctx->stamp_ring);
            ctx->stamp_ring = nullptr; ctx->stamp_slot_words = 0;
            if (hipMalloc((void **)&ctx->stamp_ring, sizeof(unsigned long long) * (size_t)words * kStampSlots) == hipSuccess) ctx->stamp_slot_words = words;
            else (void)hipGetLastError();
        }
        if (!ctx->stamp_out && hipMalloc((void **)&ctx->stamp_out, sizeof(unsigned long long) * 2 * kStampSlots) != hipSuccess) { (void)hipGetLastError(); ctx->stamp_out = nullptr; }
        if (ctx->stamp_ring && ctx->stamp_out) {
            const int launches = pairing ? (K + 1) / 2 : K;
            stamp_every = (launches + kStampSlots - 1) / kStampSlots;
            ctx->stamp_ticks.assign(kStampSlots, 0); ctx->stamp_waves.assign(kStampSlots, 0);
        }
    }
    int stamp_launch = 0;
    auto next_stamps = [&](int ticks, int64_t missile_rows) -> unsigned long long * {
        if (!stamp_every) return nullptr;
        const int l = stamp_launch++;
        if (l % stamp_every != 0 || ctx->stamp_used >= kStampSlots) return nullptr;
        const int slot = ctx->stamp_used++;
        ctx->stamp_ticks[slot] = ticks;
        ctx->stamp_waves[slot] = ((int64_t)nblocks(st->n, ZRK_BLOCK) + nblocks(missile_rows, ZRK_BLOCK)) * (ZRK_BLOCK / 64);
        return ctx->stamp_ring + (int64_t)slot * ctx->stamp_slot_words;
    };
    const int vis_cur_call = st->vis_cur;               // (which of the caller's two mask buffers is current as the call starts)
    for (int k = 0; pairing && k < K && rc == 0;) {
        const int nt = (k + 1 < K) ? 2 : 1;
        const int32_t cur_before = st->cur, vis_cur_before = st->vis_cur;
        const int cur_a = st->cur ^ 1, cur_b = st->cur;                     // tick t's buffer, tick t + 1's
        // timing: the launch that holds the last tick of a window of `stride` ticks (or the call's last tick)
        int prof_idx = -1;
        for (int j = k; j < k + nt; ++j)
            if (n_prof && ((j + 1) % stride == 0 || j + 1 == K)) {
                if (prof_idx >= 0) ctx->tev_alias[prof_idx] = j / stride;       // (both ticks are samples: one measurement)
                prof_idx = j / stride;
                ctx->tev_alias[prof_idx] = prof_idx; ctx->tev_ticks[prof_idx] = nt;
            }
        const bool fused = m > 0;
        MissileArgs M = fused ? missile_args(e, cur_a, mis, m, st->time_ms, st->dt_ms, 0) : no_missiles();
        M.pos_abs[0] = e->pos[0]; M.pos_abs[1] = e->pos[1];
        if (fused) M.grec = grec;
        // an exchange: every tick's list goes to its own send buffer (slot tick % ZRK_EXCHANGE_SLOTS), the events behind it
        const int xslot[2] = {(int)(st->tick % (uint64_t)ZRK_EXCHANGE_SLOTS), (int)((st->tick + 1) % (uint64_t)ZRK_EXCHANGE_SLOTS)};
        int64_t *list_t[2] = {xio ? xio->send[xslot[0]] : packed, xio ? xio->send[xslot[1]] : packed};
        const int64_t list_words = xio ? xio->words - ev_words : packed_capacity;
        if (fused && ev_words) { M.ev_wire_cap = xio->ev_capacity; M.gid0 = st->gid0; }
        // masks: every tick but the call's last writes into one of the loop's own buffers, the last one where the caller reads
        uint32_t *vis_t[2] = {nullptr, nullptr};
        int slot_t[2] = {Side::kMasks, Side::kMasks};
        uint32_t sparse_t[2] = {kSparseVis, kSparseVis2};
        const bool two_vis = e->vis_mask_alt != nullptr;
        if (ctx->ring_key != (const void *)e->vis_mask) { ctx->ring_key = e->vis_mask; ctx->ring_clean[0] = ctx->ring_clean[1] = false; }
        for (int j = 0; j < nt && rc == 0; ++j) {
            if (two_vis) st->vis_cur ^= 1; else st->vis_cur = 0;
            if (k + j + 1 < K) {
                slot_t[j] = (int)(sd->mask_pos++ % Side::kMasks);
                vis_t[j] = sd->masks[slot_t[j]];
            } else {
                vis_t[j] = st->vis_cur ? e->vis_mask_alt : e->vis_mask;
                if (!(two_vis && ctx->ring_clean[st->vis_cur])) sparse_t[j] = 0u;
            }
            rc = side_wait(ctx, sd, slot_t[j], s);
        }
        if (rc != 0) { st->vis_cur = vis_cur_before; break; }
        if (fused) {
            if (slot_t[0] < Side::kMasks) M.ev_code = sd->codes[slot_t[0]];
            M.ev_code2 = (nt == 2) ? ((slot_t[1] < Side::kMasks) ? sd->codes[slot_t[1]] : mis->ev_code) : nullptr;
        }
        const uint32_t mark_a = 2u + 2u * (uint32_t)(st->tick % 126u) + (uint32_t)cur_a;
        const uint32_t mark_b = (nt == 2) ? 2u + 2u * (uint32_t)((st->tick + 1) % 126u) + (uint32_t)cur_b : mark_a;
        M.pend = sd->pend; M.mark = mark_a; M.mark2 = mark_b; M.apply = 0;
        M.t2 = (double)(st->time_ms + st->dt_ms) / 1000.0;
        const int mb = nblocks(M.m, ZRK_BLOCK);
        // (the arrivals asked for so far move on only once the launch is out: a launch that fails must not leave the target ahead
        // of the counter, or every later pair launch would spin its barrier out)
        if (nt == 2 && mb > 0) { M.bar = sd->bar; M.bar_target = sd->bar_epoch + (uint32_t)mb; }
        // the two compactions of the pair as ONE launch, where the table allows it (k_compact_pair)
        // (up to 512 workgroups: beyond, beside a sweep of that size, two launches of half as many fatter workgroups are faster
        // -- 89 against 94 us per tick at 4e6 rows)
        bool pc = nt == 2 && ctx->pair_compact && st->n <= (int64_t)kPairItems * kCompBlock * ctx->pair_compact_blocks;
        if (pc && det_idx && sd->scratch_det_ints < (int64_t)R * det_stride) {
            if (sd->scratch_det) (void)hipFree(sd->scratch_det);
            sd->scratch_det = nullptr; sd->scratch_det_ints = 0;
            if (hipMalloc((void **)&sd->scratch_det, sizeof(int32_t) * (size_t)std::max<int64_t>(1, (int64_t)R * det_stride)) != hipSuccess) { (void)hipGetLastError(); pc = false; }
            else sd->scratch_det_ints = (int64_t)R * det_stride;
        }
        if (pc && !sd->scratch_cnt && hipMalloc((void **)&sd->scratch_cnt, sizeof(int32_t) * 64) != hipSuccess) { (void)hipGetLastError(); pc = false; }
        if (pc && packed && !xio && sd->scratch_packed_words < packed_capacity) {
            if (sd->scratch_packed) (void)hipFree(sd->scratch_packed);
            sd->scratch_packed = nullptr; sd->scratch_packed_words = 0;
            if (hipMalloc((void **)&sd->scratch_packed, sizeof(int64_t) * (size_t)packed_capacity) != hipSuccess) { (void)hipGetLastError(); pc = false; }
            else sd->scratch_packed_words = packed_capacity;
        }
        if (pc && fused) { M.rm = sd->rm[slot_t[1]]; M.rm_cap = sd->rm_cap; }
        const int nbs = nblocks(st->n, ZRK_BLOCK);
        const bool ordering = ctx->order_enabled && R > 0 && nbs > 8 * ctx->cus;
        Workspace w = carve(workspace, 0, e->capacity);
        if (ctx->box_ws != workspace || ctx->box_key != (const void *)e->start_pos || st->n < ctx->box_n) {
            ctx->box_ws = workspace; ctx->box_key = e->start_pos; ctx->box_n = 0;
        }
        if (st->n != ctx->box_n) {
            const int64_t w0 = ctx->box_n / ZRK_BLOCK, w1 = (st->n + ZRK_BLOCK - 1) / ZRK_BLOCK;
            if (hipMemsetAsync(w.boxes + w0, 0, sizeof(WaveBox) * (size_t)(w1 - w0), s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "memset boxes"); break; }
            ctx->box_n = st->n;
        }
        if (ordering && (ctx->order_ws != workspace || ctx->order_nb != nbs || !ctx->order_ready)) {
            if (hipMemsetAsync(w.order_ctr, 0, 2 * kOrderCtrSet * sizeof(uint32_t), s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "memset order counters"); break; }
            ctx->order_ws = workspace; ctx->order_nb = nbs; ctx->order_ready = false; ctx->order_phase = 0;
        }
        if (!ordering) ctx->order_ready = false;
        const int oph = ctx->order_phase;
        // the radars as they stand in the second tick
        zrk_radar radars_b[ZRK_MAX_RADARS];
        std::memcpy(radars_b, radars, sizeof(zrk_radar) * (size_t)R);
        zrk_scan_advance(radars_b, scan, R);
        PairLaunch pl{st->time_ms + st->dt_ms, radars_b, vis_t[1], mark_b};
        const bool on_dispatch = prof_idx >= 0;
        if (k == 0) g_trace.mark("run_ticks: first launch prepared");
        const auto t_launch = std::chrono::steady_clock::now();
        // the call's first launch zeroes the caller's buffer that its last tick will write (SweepParams::vis_clear)
        uint32_t *vis_clear = nullptr;
        const int last_buf = vis_cur_call ^ (K & 1);
        if (k == 0 && two_vis && K > nt && !ctx->ring_clean[last_buf]) vis_clear = last_buf ? e->vis_mask_alt : e->vis_mask;
        const int rc_sweep =
            launch_sweep(ctx, e, st->n, cur_a, st->time_ms, radars, R, st->flags | ZRK_F_ADVANCE | sparse_t[0] | (nt == 2 ? sparse_t[1] : 0u),
                         st->seed, st->tick, st->gid0, workspace, stream, M, vis_t[0], ordering ? w.order[oph ^ 1] : nullptr,
                         (ordering && ctx->order_ready) ? w.order[oph] : nullptr, w.boxes, nullptr, nullptr,
                         k > 0 ? sd->hflag_dev : nullptr, sd->seq,
                         on_dispatch ? ev[2 * prof_idx] : nullptr, on_dispatch ? ev[2 * prof_idx + 1] : nullptr,
                         w.order_ctr + kOrderCtrSet * (oph ^ 1), w.order_ctr + kOrderCtrSet * oph, sd->pend, mark_a, nt == 2 ? &pl : nullptr,
                         next_stamps(nt, M.m), vis_clear);
        rc = rc_sweep;
        if (rc_sweep == 0 && vis_clear) ctx->ring_clean[last_buf] = true;
        if (stall_us() > 0) stall_report(t_launch, __LINE__, "launch_sweep");
        g_trace.mark(nt == 2 ? "run_ticks: pair launched" : "run_ticks: sweep launched");
        if (rc_sweep != 0) { st->vis_cur = vis_cur_before; break; }
        if (nt == 2 && mb > 0) sd->bar_epoch += (uint32_t)mb;
        marks_used = true;
        if (ordering) { ctx->order_ready = true; ctx->order_phase = oph ^ 1; }
        // the radars move on by the ticks swept, and the next launch's records are derived while this one runs
        std::memcpy(radars, radars_b, sizeof(zrk_radar) * (size_t)R);
        if (nt == 2) zrk_scan_advance(radars, scan, R);
        // (also behind the call's last launch: the next call's first launch, from an idle device, then finds them ready)
        radar_block_ahead(ctx, radars, R, st->flags, 0);
        std::memcpy(radars_b, radars, sizeof(zrk_radar) * (size_t)R);
        zrk_scan_advance(radars_b, scan, R);
        radar_block_ahead(ctx, radars_b, R, st->flags, 1);
        const bool last_launch = k + nt == K;
        const bool tail_here = last_launch && !xio && ctx->tail_on_compute;       // (SideItem::on_compute)
        // ... and independent of the side stream's launch before it (Side::tail_ws): a pair's one compaction launch, the second
        // tick's lists of every launch but the call's last in the context's own buffers like the first tick's
        const bool spare_outputs = pc && !xio && ctx->tail_on_compute && ctx->tail_free && sd->tail_ws;
        const bool tail_free = tail_here && spare_outputs;
        if (last_launch && ctx->tail_by_event && hipEventRecord(sd->last_sweep, s) != hipSuccess) rc = fail(ctx, ZRK_E_HIP, "hipEventRecord");
        // the side stream's work, released when the NEXT launch starts: both ticks' compactions in one launch ...
        if (pc && rc == 0) {
            SideItem a, b;
            std::memset((void *)&a, 0, sizeof(a));
            std::memset((void *)&b, 0, sizeof(b));
            // (kPairSlots list slots per workgroup whatever its size: the slots per thread follow)
            const int items = kPairItems;
            // (the lists' send buffers were last sent ZRK_EXCHANGE_SLOTS ticks ago: those collectives must be through)
            for (int j = 0; j < 2 && rc == 0 && xio; ++j)
                if (int rcw = zrk_exchange_wait(xio->x, xslot[j], stream)) rc = fail(ctx, rcw, zrk_exchange_last_error(xio->x));
            if (rc == 0)
                rc = launch_compact(ctx, vis_t[0], st->n, R, st->base_index, workspace, det_idx ? sd->scratch_det : nullptr, det_stride,
                                    det_idx ? sd->scratch_cnt : nullptr, xio ? list_t[0] : (packed ? sd->scratch_packed : nullptr), list_words,
                                    st->gid0, stream, no_missiles(), vis_t[0], (st->flags & ZRK_F_UNION_BITS) != 0, nullptr, nullptr, &a, items, wire_select);
            // (the second tick's lists: the caller's -- of the call's last launch only, where its last compaction stands alone)
            const bool spare = spare_outputs && !last_launch;
            if (rc == 0)
                rc = launch_compact(ctx, vis_t[1], st->n, R, st->base_index, workspace, (det_idx && spare) ? sd->scratch_det : det_idx, det_stride,
                                    (det_idx && spare) ? sd->scratch_cnt : det_cnt, (spare && packed) ? sd->scratch_packed : list_t[1], list_words,
                                    st->gid0, stream, no_missiles(), (slot_t[1] < Side::kMasks) ? vis_t[1] : nullptr,
                                    (st->flags & ZRK_F_UNION_BITS) != 0, nullptr, nullptr, &b, items, wire_select);
            for (int j = 0; j < 2 && rc == 0 && ev_words && !fused; ++j)     // (no missiles: an empty event list behind each list)
                if (hipMemsetAsync(list_t[j] + list_words, 0, sizeof(int64_t), s) != hipSuccess) rc = fail(ctx, ZRK_E_HIP, "memset events");
            if (rc == 0) {
                // (workgroups of 1024 threads up to 512 of them; beyond, beside a sweep that keeps every compute unit full for
                // longer, half-size ones find room sooner: C3x4 83 against 119 us/tick, C3 the other way round, 22.7 against 20.9)
                a.pair = 1; a.C2 = b.C; a.pair_threads = ctx->pair_threads ? ctx->pair_threads : (a.C.nb > 512 ? 512 : 1024);
                a.C.items = a.C2.items = kPairSlots / a.pair_threads;
                int lanes = 1;
                while (lanes < 2 * (R + 1)) lanes <<= 1;
                a.C.lanes = lanes;
                a.C.group = a.C.nb > 512 ? 32 : (a.C.nb > 32 ? 16 : 0);   // (half as many records per batch of loads as the single compaction)
                if (ctx->env_group >= 0) a.C.group = ctx->env_group;
                a.stream = tail_here ? s : side_stream; a.on_compute = tail_here ? (tail_free ? 2 : 1) : 0;
                std::memset(&a.marks, 0, sizeof(a.marks));
                if (tail_here && ctx->marks_in_tail && sd->pend && sd->pend_rows >= e->capacity) {
                    // (the call's last launch, in order behind its last sweep: the marks are carried out here, MarksArgs)
                    a.marks.pend = sd->pend; a.marks.alive = e->alive; a.marks.pos0 = e->pos[0]; a.marks.pos1 = e->pos[1];
                    a.marks.cap = e->capacity; a.marks.n = st->n;
                    marks_in_tail = true;
                }
                if (tail_free) {                           // control words and records of its own
                    const Workspace tw = carve(sd->tail_ws, 0, 0);
                    a.C.ctl = a.C2.ctl = tw.ctl; a.C.agg = a.C2.agg = tw.agg;
                }
                a.flag_value = ++sd->seq; a.done_slot = slot_t[0]; a.done_slot2 = slot_t[1];
                a.M = M; a.M.apply = 0; a.M.clear_vis = nullptr;
                if (fused) {                               // the first tick's ordered events: a list of the context's own
                    a.M.ev_missile = sd->scratch_ev; a.M.ev_target = sd->scratch_ev + sd->scratch_ev_rows;
                    a.M.ev_count = sd->scratch_ev + 2 * sd->scratch_ev_rows;
                }
                a.M2 = M; a.M2.apply = 0; a.M2.ev_code = M.ev_code2; a.M2.clear_vis = nullptr;
                if (fused && spare) {                      // (its ordered events as well)
                    a.M2.ev_missile = sd->scratch_ev; a.M2.ev_target = sd->scratch_ev + sd->scratch_ev_rows;
                    a.M2.ev_count = sd->scratch_ev + 2 * sd->scratch_ev_rows;
                }
                if (fused && ev_words) { a.M.ev_wire = list_t[0] + list_words; a.M2.ev_wire = list_t[1] + list_words; }
                a.rm = fused ? sd->rm[slot_t[1]] : nullptr; a.rm_cap = sd->rm_cap;
                if (last_launch && ctx->tail_by_event) a.wait_event = sd->last_sweep;
                a.record_event = last_launch ? 1 : 0;
                uint32_t v_first = 0;
                if (xio) {
                    // both collectives hang behind the one compaction launch: a launch behind it raises the exchange's word
                    zrk_exchange *x = xio->x;
                    v_first = ++x->seq;
                    a.raise = x->flag; a.raise_value = ++x->seq;
                    if (x->one_helper) {
                        a.post_x = x; a.post_words = xio->words;
                        a.post_slot = xslot[0]; a.post_send = list_t[0]; a.post_recv = xio->recv[xslot[0]];
                        a.post2_slot = xslot[1]; a.post2_send = list_t[1]; a.post2_recv = xio->recv[xslot[1]];
                    }
                }
                rc = side_enqueue(ctx, sd, a);
                if (rc == 0 && xio) {
                    zrk_exchange *x = xio->x;
                    if (x->one_helper) { x->via_side = sd; x->side_item_no[xslot[0]] = x->side_item_no[xslot[1]] = sd->head.load(); }
                    else {
                        (void)v_first;
                        if (x->poster.joinable()) {
                            zrk_exchange::PostItem pi{xslot[0], list_t[0], xio->recv[xslot[0]], xio->words, a.raise_value};
                            pi.slot2 = xslot[1]; pi.send2 = list_t[1]; pi.recv2 = xio->recv[xslot[1]];
                            if (int rce = exchange_enqueue(x, pi)) rc = fail(ctx, rce, zrk_exchange_last_error(x));
                        } else if (exchange_post_behind_flag(x, xslot[0], list_t[0], xio->recv[xslot[0]], xio->words, a.raise_value,
                                                             xslot[1], list_t[1], xio->recv[xslot[1]]) != 0)
                            rc = fail(ctx, ZRK_E_HIP, zrk_exchange_last_error(x));
                    }
                }
                side_last = slot_t[1];
                tail_on_s = tail_here;
                tail_alone = tail_free;
                if (slot_t[1] == Side::kMasks) ctx->ring_clean[st->vis_cur] = false;
            }
        }
        // ... or one compaction per tick, in tick order
        for (int j = 0; j < nt && rc == 0 && !pc; ++j) {
            SideItem it;
            std::memset((void *)&it, 0, sizeof(it));
            if (xio) { if (int rcw = zrk_exchange_wait(xio->x, xslot[j], stream)) { rc = fail(ctx, rcw, zrk_exchange_last_error(xio->x)); break; } }
            rc = launch_compact(ctx, vis_t[j], st->n, R, st->base_index, workspace, det_idx, det_stride, det_cnt, list_t[j], list_words,
                                st->gid0, stream, no_missiles(), (slot_t[j] < Side::kMasks) ? vis_t[j] : nullptr,
                                (st->flags & ZRK_F_UNION_BITS) != 0, nullptr, nullptr, &it, 0, wire_select);
            if (rc != 0) break;
            if (ev_words && !fused && hipMemsetAsync(list_t[j] + list_words, 0, sizeof(int64_t), s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "memset events"); break; }
            it.stream = tail_here ? s : side_stream; it.on_compute = tail_here ? 1 : 0;
            it.flag_value = ++sd->seq; it.done_slot = slot_t[j];
            it.M = M; it.M.apply = 0;
            if (fused && j == 1) it.M.ev_code = M.ev_code2;
            if (fused && ev_words) it.M.ev_wire = list_t[j] + list_words;
            it.M.clear_vis = (fused && nt == 2 && j == 0) ? vis_t[1] : nullptr;
            if (last_launch && ctx->tail_by_event) it.wait_event = sd->last_sweep;
            it.record_event = last_launch ? 1 : 0;
            if (xio) {
                zrk_exchange *x = xio->x;
                it.raise = x->flag; it.raise_value = ++x->seq;
                if (x->one_helper) {
                    it.post_x = x; it.post_words = xio->words;
                    it.post_slot = xslot[j]; it.post_send = list_t[j]; it.post_recv = xio->recv[xslot[j]];
                }
            }
            rc = side_enqueue(ctx, sd, it);
            if (rc == 0 && xio) {
                zrk_exchange *x = xio->x;
                if (x->one_helper) { x->via_side = sd; x->side_item_no[xslot[j]] = sd->head.load(); }
                else if (x->poster.joinable()) {
                    if (int rce = exchange_enqueue(x, zrk_exchange::PostItem{xslot[j], list_t[j], xio->recv[xslot[j]], xio->words, it.raise_value}))
                        rc = fail(ctx, rce, zrk_exchange_last_error(x));
                } else if (exchange_post_behind_flag(x, xslot[j], list_t[j], xio->recv[xslot[j]], xio->words, it.raise_value) != 0)
                    rc = fail(ctx, ZRK_E_HIP, zrk_exchange_last_error(x));
            }
            side_last = slot_t[j];
            tail_on_s = tail_here;
            if (slot_t[j] == Side::kMasks) ctx->ring_clean[st->vis_cur] = false;
        }
        st->cur = (nt == 2) ? cur_b : cur_a;
        st->time_ms += nt * st->dt_ms;
        st->tick += (uint64_t)nt;
        k += nt;
        (void)cur_before;
    }
    for (int k = pairing ? K : 0; k < K && rc == 0; ++k) {
        // (a tick that fails before its sweep is launched leaves the loop state as the last complete tick left it)
        const int32_t cur_before = st->cur, vis_cur_before = st->vis_cur;
        st->cur ^= 1;
        // (the LAST tick of every window of `stride` ticks is the timed one, and the call's last tick: the first launch of a
        // call, from an idle stream, is not what a tick of the loop looks like, and a timed launch costs the stream ~5 us)
        const bool prof = n_prof && ((k + 1) % stride == 0 || k + 1 == K);
        const bool on_dispatch = prof && ctx->time_on_dispatch && (st->n > 0 || (m > 0 && m <= 1024 * (int64_t)kMissileItems));
        if (prof && !on_dispatch && hipEventRecord(ev[2 * (k / stride)], s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "hipEventRecord"); st->cur = cur_before; break; }
        const int slot = (int)(st->tick % (uint64_t)ZRK_EXCHANGE_SLOTS);
        int64_t *list = xio ? xio->send[slot] : packed;
        const int64_t list_words = xio ? xio->words - ev_words : packed_capacity;
        // Missiles read last tick's positions (pos[cur^1]) and trajectories only, so their per-row step rides in
        // the sweep's grid; the ordered event list and the tombstones (effective from the next tick,
        // AirEnv.py:33-40) ride in the compaction's grid, behind the sweep.  Tables too long for one
        // finishing workgroup, or ticks without compaction, take the stand-alone launches instead.
        const bool fused = m > 0 && m <= 1024 * (int64_t)kMissileItems && (det_idx || list) && st->n > 0 && R > 0;
        MissileArgs M = fused ? missile_args(e, st->cur, mis, m, st->time_ms, st->dt_ms, 1) : no_missiles();
        if (fused) M.grec = grec;
        if (fused && !sd) M.frozen_prev = ctx->frozen_prev;                   // (zrk_ctx_keep_prev: the plain loop's tombstones)
        if (fused && ev_words) { M.ev_wire = list + list_words; M.ev_wire_cap = xio->ev_capacity; M.gid0 = st->gid0; }
        // two mask buffers: tick t writes only its detections into one (cleared by the previous tick's
        // scatter) while its own scatter clears the other for tick t+1.  The first tick on a pair of buffers
        // writes densely; from then on the pair belongs to this loop, also between calls (a caller that
        // writes them itself must pass other buffers or call with vis_mask_alt = NULL).
        const bool two_vis = e->vis_mask_alt && (det_idx || list) && st->n > 0 && R > 0;
        if (ctx->ring_key != (const void *)e->vis_mask || !two_vis) {
            ctx->ring_key = e->vis_mask; ctx->ring_clean[0] = ctx->ring_clean[1] = false;
        }
        if (two_vis) st->vis_cur ^= 1; else st->vis_cur = 0;
        uint32_t *vis_now = st->vis_cur ? e->vis_mask_alt : e->vis_mask;
        uint32_t *vis_next = two_vis ? (st->vis_cur ? e->vis_mask : e->vis_mask_alt) : nullptr;
        uint32_t sparse = (two_vis && ctx->ring_clean[st->vis_cur]) ? kSparseVis : 0u;
        int side_slot = Side::kMasks;                     // overlap mode: the last tick's masks go where the caller can read them
        if (sd && k + 1 < K) {
            // ... the others into one of the side stream's own buffers: clean, once the compaction that used it three
            // ticks ago is through (the host waits, see zrk_exchange_wait for why not the stream)
            side_slot = (int)(sd->mask_pos++ % Side::kMasks);
            vis_now = sd->masks[side_slot];
            sparse = kSparseVis;
        }
        if (sd && (rc = side_wait(ctx, sd, side_slot, s)) != 0) { st->cur = cur_before; st->vis_cur = vis_cur_before; break; }
        if (sd && side_slot < Side::kMasks && M.m > 0) M.ev_code = sd->codes[side_slot];
        // removals as marks (one scenario): carried out by the next sweep's own threads, or behind the call's last tick
        const bool marks = sd && !ens && M.m > 0;
        // (the mark's low bit: which position buffer is current in this tick, i.e. holds what a removed row froze at)
        const uint32_t mark = marks ? 2u + 2u * (uint32_t)(st->tick % 126u) + (uint32_t)st->cur : 0u;
        if (marks) { M.pend = sd->pend; M.mark = mark; M.mark2 = mark; M.apply = 0; }
        // next tick's dispatch order: this tick's sweep builds it as it goes
        const int nbs = nblocks(st->n, ZRK_BLOCK);
        // (a grid that is resident all at once has no "last": eight workgroups of four waves fit a compute unit)
        const bool ordering = ctx->order_enabled && R > 0 && nbs > 8 * ctx->cus;
        if (ens) {
            EL.rb_table = (const char *)ens_tables[st->tick & 1u];
            EL.next.table_out = ens_tables[(st->tick + 1) & 1u];
        }
        Workspace w = carve(workspace, 0, e->capacity);
        // box records: none for rows the loop has not swept yet on this table (new table, new workspace, rows appended)
        if (ctx->box_ws != workspace || ctx->box_key != (const void *)e->start_pos || st->n < ctx->box_n) {
            ctx->box_ws = workspace; ctx->box_key = e->start_pos; ctx->box_n = 0;
        }
        if (st->n != ctx->box_n) {
            const int64_t w0 = ctx->box_n / ZRK_BLOCK, w1 = (st->n + ZRK_BLOCK - 1) / ZRK_BLOCK;
            if (hipMemsetAsync(w.boxes + w0, 0, sizeof(WaveBox) * (size_t)(w1 - w0), s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "memset boxes"); break; }
            ctx->box_n = st->n;
        }
        if (ordering && (ctx->order_ws != workspace || ctx->order_nb != nbs || !ctx->order_ready)) {
            if (hipMemsetAsync(w.order_ctr, 0, 2 * kOrderCtrSet * sizeof(uint32_t), s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "memset order counters"); break; }
            ctx->order_ws = workspace; ctx->order_nb = nbs; ctx->order_ready = false; ctx->order_phase = 0;
        }
        if (!ordering) ctx->order_ready = false;
        const int oph = ctx->order_phase;                                    // reads list oph, builds list oph ^ 1
        const int rc_sweep =
             launch_sweep(ctx, e, st->n, st->cur, st->time_ms, radars, R, st->flags | ZRK_F_ADVANCE | sparse, st->seed,
                          st->tick, st->gid0, workspace, stream, M, vis_now, ordering ? w.order[oph ^ 1] : nullptr,
                          (ordering && ctx->order_ready) ? w.order[oph] : nullptr, w.boxes, ens ? &EL : nullptr,
                          rb_through_memory ? rb_dev[st->tick & 1u] : nullptr,
                          pend.on ? fx->flag : ((sd && k > 0) ? sd->hflag_dev : nullptr), pend.on ? pend.value : (sd ? sd->seq : 0u),
                          on_dispatch ? ev[2 * (k / stride)] : nullptr, on_dispatch ? ev[2 * (k / stride) + 1] : nullptr,
                          w.order_ctr + kOrderCtrSet * (oph ^ 1), w.order_ctr + kOrderCtrSet * oph, marks ? sd->pend : nullptr, mark, nullptr,
                          next_stamps(1, M.m));
        rc = rc_sweep;
        g_trace.mark("run_ticks: sweep launched");
        if (rc_sweep == 0 && marks) marks_used = true;
        if (rc == 0 && ordering) { ctx->order_ready = true; ctx->order_phase = oph ^ 1; }
        if (rc == 0 && held.on) rc = issue_held();                           // (the previous tick's item, see below)
        if (pend.on) {                                                       // the previous tick's collective, behind this sweep's start
            if (rc == 0 && !(st->n > 0 || M.m > 0)) {                        // (no sweep was launched: raise the flag by itself)
                hipLaunchKernelGGL(k_raise_flag, dim3(1), dim3(1), 0, s, fx->flag, pend.value);
                rc = check_launch(ctx, "k_raise_flag");
            }
            if (rc == 0 && fx->poster.joinable()) {
                if (int rce = exchange_enqueue(fx, zrk_exchange::PostItem{pend.slot, pend.list, xio->recv[pend.slot], xio->words, pend.value}))
                    rc = fail(ctx, rce, zrk_exchange_last_error(fx));
            } else if (rc == 0 && exchange_post_behind_flag(fx, pend.slot, pend.list, xio->recv[pend.slot], xio->words, pend.value) != 0)
                rc = fail(ctx, ZRK_E_HIP, zrk_exchange_last_error(fx));
            pend.on = false;
        }
        if (!ens && rc_sweep == 0) zrk_scan_advance(radars, scan, R);         // Radar.py:205 (an ensemble's: on the device)
        if (!ens && rc_sweep == 0 && !rb_through_memory) radar_block_ahead(ctx, radars, R, st->flags);
        if (rb_through_memory) {                                             // the next tick's records ride with this compaction
            fill_radar_block(ctx, radars, R, st->flags, put.rb);
            put.dst = (uint32_t *)rb_dev[(st->tick + 1) & 1u];
        }
        if (prof && !on_dispatch && rc == 0 && hipEventRecord(ev[2 * (k / stride) + 1], s) != hipSuccess) rc = fail(ctx, ZRK_E_HIP, "hipEventRecord");
        if (sd) {
            if (rc != 0) {
                // no sweep: as after the last complete tick; swept but its side work cannot be handed over: the tick counts
                // (its removals are marks, carried out below), its lists are lost with the call
                if (rc_sweep != 0) { st->cur = cur_before; st->vis_cur = vis_cur_before; }
                else { st->time_ms += st->dt_ms; st->tick += 1; }
                break;
            }
            // (the next sweep's first thread, or a launch behind the loop, tells the side stream's thread that this tick's
            // launches on the compute stream are over)
            const int eparts = ens ? nblocks((int64_t)EL.next.S * EL.next.R, kCompBlock) : 0;
            const int small_grid = std::max(1, nblocks(M.m, kCompBlock) + eparts);
            SideItem it;
            std::memset((void *)&it, 0, sizeof(it));
            // (the single-launch workspace's first-use clearing, if any, goes to the compute stream, ahead of the flag)
            rc = launch_compact(ctx, vis_now, st->n, R, st->base_index, workspace, det_idx, det_stride, det_cnt, list, list_words,
                                st->gid0, stream, no_missiles(), (side_slot < Side::kMasks) ? vis_now : nullptr,
                                (st->flags & ZRK_F_UNION_BITS) != 0, ens ? &EL : nullptr, nullptr, &it, 0, wire_select);
            if (rc != 0) break;
            const uint32_t v = ++sd->seq;
            if (ens) {
                hipLaunchKernelGGL(k_tick_small, dim3(small_grid), dim3(kCompBlock), 0, s, M, EL.next);
                if ((rc = check_launch(ctx, "k_tick_small")) != 0) break;
            }
            if (ev_words && !fused && hipMemsetAsync(list + list_words, 0, sizeof(int64_t), s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "memset events"); break; }
            const bool tail_here = k + 1 == K && !xio && ctx->tail_on_compute;
            it.stream = tail_here ? s : side_stream; it.on_compute = tail_here ? 1 : 0;
            it.flag_value = v; it.done_slot = side_slot;
            it.M = M; it.M.apply = 0;
            it.record_event = (k + 1 == K) ? 1 : 0;
            if (k + 1 == K && ctx->tail_by_event) {
                if (hipEventRecord(sd->last_sweep, s) != hipSuccess) { rc = fail(ctx, ZRK_E_HIP, "hipEventRecord"); break; }
                it.wait_event = sd->last_sweep;
            }
            if (xio) {
                // The collective of this tick runs on the exchange's own stream, released by a launch behind the compaction.
                // The list it sends was last sent ZRK_EXCHANGE_SLOTS ticks ago, and that collective must be through before the compaction
                // may write it -- checked as late as possible, i.e. when the NEXT sweep has been launched (the item is
                // held until then: the side stream cannot start it before that sweep runs anyway), so that a late
                // collective never delays a launch on the compute stream.
                it.raise = fx->flag; it.raise_value = ++fx->seq;
                held.on = true; held.it = it; held.slot = slot; held.list = list;
            } else if ((rc = side_enqueue(ctx, sd, it)) != 0) {
                break;
            }
            side_last = side_slot;
            tail_on_s = tail_here;
            if (side_slot == Side::kMasks) ctx->ring_clean[st->vis_cur] = false;
            st->time_ms += st->dt_ms;
            st->tick += 1;
            continue;
        }
        // this slot's list was last sent ZRK_EXCHANGE_SLOTS ticks ago: that collective must have read it before it is rewritten
        if (rc == 0 && xio) { if (int rcw = zrk_exchange_wait(xio->x, slot, stream)) rc = fail(ctx, rcw, zrk_exchange_last_error(xio->x)); }
        if (rc == 0 && two_vis) { ctx->ring_clean[st->vis_cur] = false; ctx->ring_clean[st->vis_cur ^ 1] = true; }
        if (rc == 0 && (det_idx || list))
            rc = launch_compact(ctx, vis_now, st->n, R, st->base_index, workspace, det_idx, det_stride, det_cnt, list,
                                list_words, st->gid0, stream, M, vis_next,
                                (st->flags & ZRK_F_UNION_BITS) != 0, ens ? &EL : nullptr, rb_through_memory ? &put : nullptr, nullptr, 0, wire_select);
        if (rc == 0 && m > 0 && !fused) rc = zrk_missile_step(ctx, e, st->cur, mis, m, st->time_ms, st->dt_ms, 1, stream);
        if (rc == 0 && ev_words && !fused) {
            if (m > 0) {
                hipLaunchKernelGGL(k_events_wire, dim3(1), dim3(256), 0, s, mis->ev_missile, mis->ev_target, mis->ev_count,
                                   e->list_index, st->gid0, list + list_words, xio->ev_capacity);
                rc = check_launch(ctx, "k_events_wire");
            } else if (hipMemsetAsync(list + list_words, 0, sizeof(int64_t), s) != hipSuccess) rc = fail(ctx, ZRK_E_HIP, "memset events");
        }
        // the collective of this tick, behind the compaction on RCCL's own stream: it overlaps the next tick's sweep
        if (rc == 0 && xio) {
            if (fx && k + 1 < K) { pend.on = true; pend.slot = slot; pend.list = list; pend.value = ++fx->seq; }
            else if (zrk_exchange_all_gather(xio->x, slot, list, xio->recv[slot], xio->words, stream) != 0)
                rc = fail(ctx, ZRK_E_HIP, zrk_exchange_last_error(xio->x));
        }
        st->time_ms += st->dt_ms;                                            // Manager.py:140
        st->tick += 1;
    }
    g_trace.mark("run_ticks: loop done");
    if (sd) {
        if (held.on && rc == 0) rc = issue_held();
        // behind the last tick: its removals as tombstones, the word the side stream's thread waits for, then the side
        // stream's work is issued to the last item and the compute stream takes it in: the lists are the caller's
        if (rc == 0 && side_last >= 0 && !ctx->tail_by_event && !tail_on_s) {   // (first: the last compaction may start as soon as the last sweep is over)
            hipLaunchKernelGGL(k_raise_flag_system, dim3(1), dim3(1), 0, s, sd->hflag_dev, sd->seq);
            rc = check_launch(ctx, "k_raise_flag");
        }
        if (marks_used && sd->pend && sd->pend_rows >= e->capacity && !(marks_in_tail && rc == 0)) {
            // the last tick's removals are still marks: tombstones now, and the call's marks cleared -- also when the call
            // has failed (a helper that gave up, a wait that ran out): the table then stands as after the st->tick ticks
            // that were swept, only the lists of this call are not valid
            hipLaunchKernelGGL(k_apply_marks, dim3(nblocks(st->n, 256)), dim3(256), 0, s, sd->pend, e->alive, e->pos[0],
                               e->pos[1], e->capacity, st->n);
            const int rcm = check_launch(ctx, rc == 0 ? "k_apply_marks" : ctx->err.c_str());
            if (rc == 0) rc = rcm;
        }
        if (rc != 0) {
            // the call has failed: nobody will raise the word for the items the side stream's thread still holds -- do it
            // from here, behind everything the compute stream has, so that the thread is not left to sit out its limit
            (void)hipStreamSynchronize(s);
            *sd->hflag = sd->seq;
        }
        const int rc_side = side_drain(ctx, sd, sd->head.load());
        if (rc == 0) rc = rc_side;
        g_trace.mark("run_ticks: side drained");
        if (xio) xio->x->via_side = nullptr;                 // (everything it carried has been issued, or has failed with it)
        if (rc != 0) { sd->masks_dirty = true; sd->pend_rows = 0; ctx->ring_clean[0] = ctx->ring_clean[1] = false; }  // (marks: allocated and cleared anew)
        if (rc == 0 && side_last >= 0 && sd->posted[side_last]) {
            // (a last item on the compute stream is in it already; the event behind it is for a later call on ANOTHER stream)
            if (!tail_on_s && hipStreamWaitEvent(s, sd->done_of[side_last], 0) != hipSuccess) rc = fail(ctx, ZRK_E_HIP, "hipStreamWaitEvent");
            else if (tail_alone) {
                // (the side stream's last launch stood alone, and the thread has put the wait for it BEHIND the call's last
                // compaction on the compute stream: see side_issue)
                sd->tail_item = sd->head.load(); sd->tail_stream = s; sd->joined_upto = sd->head.load(); sd->joined_stream = s; sd->side_busy = false;
            }
            else { sd->joined_upto = sd->head.load(); sd->joined_stream = s; if (!tail_on_s) sd->side_busy = false; }
        }
    }
    g_trace.mark("run_ticks: exit");
    if (!g_trace.deferred) g_trace.dump();
    // a collective whose hand-over never came went out poisoned (k_wait_flag): the host knows without a synchronisation
    if (rc == 0 && xio && exchange_gave_up(xio->x) != 0) rc = fail(ctx, ZRK_E_STATE, zrk_exchange_last_error(xio->x));
    if (n_prof && !deferred) {
        if (hipStreamSynchronize(s) != hipSuccess && rc == 0) rc = fail(ctx, ZRK_E_HIP, "hipStreamSynchronize");
        for (int k = 0; k < n_prof; ++k) {
            float ms = 0.f;
            const int a = ctx->tev_alias[k];
            if (rc != 0 || hipEventElapsedTime(&ms, ev[2 * a], ev[2 * a + 1]) != hipSuccess) {
                ms = -1.f;
                if (rc == 0) rc = fail(ctx, ZRK_E_HIP, "hipEventElapsedTime");
            }
            sweep_ms[k] = ms;
        }
    }
    return rc;
}

}  // namespace

ZRK_API int zrk_run_ticks_x(zrk_ctx *ctx, const zrk_entities *e, const zrk_missiles *mis, int64_t m, zrk_loop *st,
                            zrk_radar *radars, const zrk_scan *scan, int R, void *workspace, int32_t *det_idx,
                            int64_t det_stride, int32_t *det_cnt, int64_t *packed, int64_t packed_capacity,
                            const zrk_exchange_io *xio, int K, float *sweep_ms, int prof_stride, void *stream)
{
    return run_ticks(ctx, e, mis, m, st, radars, scan, R, workspace, det_idx, det_stride, det_cnt, packed, packed_capacity, xio,
                     nullptr, K, sweep_ms, prof_stride, stream);
}

ZRK_API int zrk_run_ticks_ensemble(zrk_ctx *ctx, const zrk_entities *e, const zrk_missiles *mis, int64_t m, zrk_loop *st,
                                   const zrk_ensemble *ens, void *workspace, int32_t *det_idx, int64_t det_stride,
                                   int32_t *det_cnt, int K, float *sweep_ms, int prof_stride, void *stream)
{
    if (!ens) return fail(ctx, ZRK_E_INVALID, "zrk_run_ticks_ensemble: null ensemble");
    return run_ticks(ctx, e, mis, m, st, nullptr, nullptr, ens->radars, workspace, det_idx, det_stride, det_cnt, nullptr, 0, nullptr,
                     ens, K, sweep_ms, prof_stride, stream);
}

ZRK_API int64_t zrk_ensemble_table_bytes(int scenarios)
{
    return scenarios < 0 ? ZRK_E_INVALID : 2 * (int64_t)scenarios * (int64_t)sizeof(RadarBlock);
}

ZRK_API double zrk_d2_threshold(double max_distance) { return d2_threshold(max_distance); }

ZRK_API int zrk_last_run_overlapped(zrk_ctx *ctx) { return ctx ? ctx->last_overlapped : ZRK_E_INVALID; }

ZRK_API int zrk_read_sweep_ms(zrk_ctx *ctx, float *sweep_ms, int n)
{
    if (!ctx || !sweep_ms || n < 0) return fail(ctx, ZRK_E_INVALID, "zrk_read_sweep_ms: null argument");
    if (n > ctx->tev_pending) return fail(ctx, ZRK_E_INVALID, "zrk_read_sweep_ms: fewer events were recorded");
    for (int k = 0; k < n; ++k) {
        const int a = ctx->tev_alias[k];
        if (hipEventSynchronize(ctx->tev[2 * a + 1]) != hipSuccess ||
            hipEventElapsedTime(&sweep_ms[k], ctx->tev[2 * a], ctx->tev[2 * a + 1]) != hipSuccess)
            return fail(ctx, ZRK_E_HIP, "zrk_read_sweep_ms: event not readable");
    }
    return 0;
}

ZRK_API int zrk_read_sweep_ticks(zrk_ctx *ctx, int32_t *ticks, int n)
{
    if (!ctx || !ticks || n < 0) return fail(ctx, ZRK_E_INVALID, "zrk_read_sweep_ticks: null argument");
    if (n > (int)ctx->tev_ticks.size()) return fail(ctx, ZRK_E_INVALID, "zrk_read_sweep_ticks: fewer samples were taken");
    for (int k = 0; k < n; ++k) ticks[k] = ctx->tev_ticks[k];
    return 0;
}

ZRK_API int zrk_last_run_ticks_per_launch(zrk_ctx *ctx) { return ctx ? ctx->last_ticks_per_launch : ZRK_E_INVALID; }

ZRK_API int zrk_sweep_stamps(zrk_ctx *ctx, int on)
{
    if (!ctx) return ZRK_E_INVALID;
    ctx->stamps_on = on != 0;
    if (!on) ctx->stamp_used = 0;
    return 0;
}

ZRK_API int zrk_read_sweep_stamps(zrk_ctx *ctx, float *sweep_us, int32_t *ticks, int cap, void *stream)
{
    if (!ctx || !sweep_us || cap < 0) return fail(ctx, ZRK_E_INVALID, "zrk_read_sweep_stamps: null argument");
    const int n = std::min(cap, ctx->stamp_used);
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < n; ++k)
        hipLaunchKernelGGL(k_reduce_stamps, dim3(1), dim3(1024), 0, s, ctx->stamp_ring + (int64_t)k * ctx->stamp_slot_words, ctx->stamp_waves[k],
                           ctx->stamp_out + 2 * k);
    if (int rc = check_launch(ctx, "k_reduce_stamps")) return rc;
    std::vector<unsigned long long> host(2 * (size_t)n);
    if (hipMemcpyAsync(host.data(), ctx->stamp_out, sizeof(unsigned long long) * 2 * (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
        return fail(ctx, ZRK_E_HIP, "zrk_read_sweep_stamps: the stamps could not be read");
    for (int k = 0; k < n; ++k) {
        sweep_us[k] = (float)((double)(host[2 * k + 1] - host[2 * k]) * 0.01);       // 100 MHz: 10 ns a count
        if (ticks) ticks[k] = ctx->stamp_ticks[k];
    }
    ctx->stamp_host = host;
    return n;
}

ZRK_API int zrk_last_sweep_stamp_times(zrk_ctx *ctx, double *begin_us, double *end_us, int cap)
{
    if (!ctx || !begin_us || !end_us || cap < 0) return fail(ctx, ZRK_E_INVALID, "zrk_last_sweep_stamp_times: null argument");
    const int n = std::min(cap, (int)(ctx->stamp_host.size() / 2));
    for (int k = 0; k < n; ++k) {
        begin_us[k] = (double)(long long)(ctx->stamp_host[2 * k] - ctx->stamp_host[0]) * 0.01;
        end_us[k] = (double)(long long)(ctx->stamp_host[2 * k + 1] - ctx->stamp_host[0]) * 0.01;
    }
    return n;
}

ZRK_API int zrk_run_ticks(zrk_ctx *ctx, const zrk_entities *e, const zrk_missiles *mis, int64_t m, zrk_loop *st,
                          zrk_radar *radars, const zrk_scan *scan, int R, void *workspace, int32_t *det_idx,
                          int64_t det_stride, int32_t *det_cnt, int64_t *packed, int64_t packed_capacity, int K,
                          float *sweep_ms, int prof_stride, void *stream)
{
    return zrk_run_ticks_x(ctx, e, mis, m, st, radars, scan, R, workspace, det_idx, det_stride, det_cnt, packed, packed_capacity,
                           nullptr, K, sweep_ms, prof_stride, stream);
}

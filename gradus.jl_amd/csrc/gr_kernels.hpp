// gr_kernels.hpp -- the trace kernels and their launcher, written against namespace GR_NS and its
// `real` type.  Included by gradus_mi355x.hip (GR_NS = gr, real = double) and by
// the fp32 objects (GR_NS = gr32, real = float, compiled with single-precision constants).
#pragma once

#include "gr_device.hpp"
#include <type_traits>

namespace GR_NS {
namespace {


template <class Metric, int DISC>
struct LaneStats {
    // per-lane 32-bit counters (a lane handles far fewer than 2^32 steps per launch)
    unsigned rays = 0, acc = 0, rej = 0, flagged = 0, st[4] = { 0, 0, 0, 0 };
    GR_DEV void add(const Ray<Metric, DISC>& r)
    {
        rays += 1;
        acc += (unsigned)r.nacc;
        rej += (unsigned)r.nrej;
        flagged += (r.flags & GR_FLAG_MASK) ? 1 : 0;
        const int s = (r.flags & GR_FLAG_MASK) ? GR_STATUS_NO_STATUS : r.status;
        st[0] += (s == 0); st[1] += (s == 1); st[2] += (s == 2); st[3] += (s == 3);
    }
    GR_DEV void flush(unsigned long long* out) const
    {
        if (!out) return;
        unsigned long long v[N_STAT] = { rays, acc, rej, 2ull * rays + 6ull * ((unsigned long long)acc + rej), flagged, st[0], st[1], st[2], st[3] };
#pragma unroll
        for (int i = 0; i < N_STAT; ++i) {
            unsigned long long x = v[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(out + i, x);
        }
    }
};

// LDS staging shared by both trace kernels: [ hist (lds_bins) | plunging table (4 x lds_plunge_rows) | cold lane storage ]
#ifndef GR_HOST_HARNESS
extern __shared__ double gr_lds[];

// Cold lane storage (gr_device.hpp, LdsColdStoreT): LDS as the place for what a block of the step does not read.
//   GR_COLD_LDS = 3 (default): around the EVENT SAMPLING only (taken in ~2 % of the wave-steps), for the metrics that run at
//                 the 168-register cap (Metric::kColdRare: Kerr).  The sampling block needs a dozen registers of its own; left to
//                 the compiler they are made by spilling to scratch, whose write-back was 113 of the 147 MB the bench kernel
//                 sent to HBM per launch.  Parked in LDS instead: 50.7 MB written (1.5 x the 33.6 MB image), same time
//                 (19.22 vs 19.19 ms, profiles/r3h_*).
//   GR_COLD_LDS = 2: the same for every metric;  1: additionally across the hot region of EVERY step (the round-3 instrument
//                 that attributed the write excess: no scratch at all, +1.3 % time, 1.7 % on Johannsen);  0: off.
// The tangent scalar and the fp64-controller build do not fit the 8-byte slots: off there.
#ifndef GR_COLD_LDS
#if defined(GR_REAL_IS_TAN2) || defined(GR_CONTROLLER_F64)
#define GR_COLD_LDS 0
#else
#define GR_COLD_LDS 3
#endif
#endif
#if GR_COLD_LDS && (defined(GR_REAL_IS_TAN2) || defined(GR_CONTROLLER_F64))
#error "the cold lane storage is written for the fp64 / fp32 scalars and the fp32 controller"
#endif
template <class Metric, class = void>
struct ColdRareOf { static constexpr bool value = false; };
template <class Metric>
struct ColdRareOf<Metric, decltype((void)Metric::kColdRare)> { static constexpr bool value = Metric::kColdRare; };

// How many stage accelerations the one-ray-per-lane kernel of a metric parks in LDS (ParkA, gr_device.hpp): Metric::kParkStages,
// GR_PARK_STAGES overrides it for every metric of a build (the tangent objects: 4)
template <class Metric, class = void>
struct ParkStagesOf { static constexpr int value = 0; };
template <class Metric>
struct ParkStagesOf<Metric, decltype((void)Metric::kParkStages)> { static constexpr int value = Metric::kParkStages; };

template <class Metric, bool LANE_KERNEL = true>      // the persistent kernel runs below the register cap: nothing to park
struct ColdSel {
#if GR_COLD_LDS == 1
    typedef LdsColdStore base;
#elif GR_COLD_LDS == 2
    typedef LdsColdStoreRare base;
#elif GR_COLD_LDS == 3
    typedef typename std::conditional<LANE_KERNEL && ColdRareOf<Metric>::value, LdsColdStoreRare, NoColdStore>::type base;
#else
    typedef NoColdStore base;
#endif
#ifdef GR_PARK_STAGES
    static constexpr int kPark = LANE_KERNEL ? GR_PARK_STAGES : 0;
#else
    static constexpr int kPark = LANE_KERNEL ? ParkStagesOf<Metric>::value : 0;
#endif
    typedef typename std::conditional<(kPark > 0), ParkA<base, kPark>, base>::type type0;
    // a tabulated metric's patch cache (TabLds): kTabLdsBytesPerWave bytes per wave behind every other LDS region
    static constexpr bool kTab = ByThetaOf<Metric>::value && GR_HAS_TABULATED;
    typedef typename std::conditional<kTab, TabLds<type0>, type0>::type type;
    static constexpr size_t kColdBytes = base::kOn ? sizeof(double) * COLD_SLOTS : 0;
    static constexpr size_t kParkBytes = sizeof(real) * 4 * (size_t)kPark;
    static constexpr size_t kBytesPerThread = kColdBytes + kParkBytes;
#if GR_HAS_TABULATED
    static constexpr size_t kTabBytesPerWave = kTab ? kTabLdsBytesPerWave : 0;
#else
    static constexpr size_t kTabBytesPerWave = 0;
#endif
};
// wave w of the workgroup owns bytes [w * 64 * kBytesPerThread, (w + 1) * 64 * kBytesPerThread) of the region behind the
// histogram and the plunging table: its cold slots first (64 lanes x COLD_SLOTS doubles), then its parked accelerations
template <class Sel>
__device__ __forceinline__ typename Sel::type cold_store_of(const Params& p)
{
    typedef typename Sel::type Store;
    Store st{};
    if constexpr (Store::kOn || Store::kParkA > 0) {
        const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
        char* region = reinterpret_cast<char*>(gr_lds + p.lds_bins + 4 * p.lds_plunge_rows) + (size_t)w * (64 * Sel::kBytesPerThread);
        if constexpr (Store::kOn) st.lane = reinterpret_cast<double*>(region) + l;
        if constexpr (Store::kParkA > 0) st.park = reinterpret_cast<real*>(region + 64 * Sel::kColdBytes) + l;
    }
#if GR_HAS_TABULATED
    if constexpr (Store::kTabLds) {
        // the patch cache of wave w: p.lds_tab_off bytes into the workgroup's LDS (launch_tmpl), empty to begin with
        typedef char __attribute__((address_space(3))) lds_char;
        typedef int __attribute__((address_space(3))) lds_int;
        const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
        st.tab = (lds_char*)((lds_char*)gr_lds + p.lds_tab_off + (size_t)w * kTabLdsBytesPerWave);
        // Every active lane writes all 16 ints (the same values): the last wave of a launch may have fewer lanes than the head
        // has entries -- a ray-set launch of 5000 rays ends in a wave of 8 -- and what they left unwritten would be whatever the
        // previous workgroup on this CU had there: a "tag" that matches by accident, a round-robin counter out of range.
        (void)l;
#ifdef GR_WAVE_TIMELINE      // ints 12..14 of the head count copies / global evaluations (no tag vector reads them with 12 slots)
        static_assert(kTabSlots == 12, "the timeline build keeps its counters behind twelve tags");
#pragma unroll
        for (int i = 0; i < 16; ++i) ((lds_int*)st.tab)[i] = i < 12 ? -1 : 0;
#else
#pragma unroll
        for (int i = 0; i < 16; ++i) ((lds_int*)st.tab)[i] = i < kTabRR ? -1 : 0;
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#endif
    return st;
}

// bytes of LDS per work-item of the end-point region: its record and its address slot
constexpr size_t kPointLdsBytesPerThread = sizeof(double) * (POINT_UNITS + 1);

template <size_t COLD_BYTES_PER_THREAD>
__device__ __forceinline__ LdsView lds_prologue(const Params& p)
{
    LdsView v{ nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    const int bins = p.lds_bins, rows = p.lds_plunge_rows;
    if (p.lds_points) {
        // wave w owns (POINT_UNITS + 1) * 64 doubles: 64 records back to back (lane stride 19 doubles = 38 banks: the
        // 64-bit accesses of a half wave fall on 32 distinct bank pairs), then the 64 address slots
        const unsigned w = threadIdx.x >> 6, l = threadIdx.x & 63;
        // a one-wave workgroup reuses the cold lane storage's bytes (that is dead once the step loop has ended); with
        // several waves per workgroup another wave may still be stepping, so the records get their own region
        const unsigned cold_doubles = blockDim.x == 64 ? 0u : (unsigned)(COLD_BYTES_PER_THREAD / sizeof(double)) * blockDim.x;
        double* reg = gr_lds + bins + 4 * rows + cold_doubles + w * (64 * (POINT_UNITS + 1));
        v.point = reg + l * POINT_UNITS;
        v.point_addr = reinterpret_cast<uint64_t*>(reg + 64 * POINT_UNITS + l);
        *v.point_addr = 0;
    }
    if (bins == 0 && rows == 0) return v;
    const Cold& cd = cold_of(p);
    double* hist = gr_lds;
    double* tab = gr_lds + bins;
    for (int i = threadIdx.x; i < bins; i += blockDim.x) hist[i] = 0.0;
    for (int i = threadIdx.x; i < rows; i += blockDim.x) {
        tab[i] = cd.pf.plunge_r[i];
        tab[rows + i] = cd.pf.plunge_vt[i];
        tab[2 * rows + i] = cd.pf.plunge_vr[i];
        tab[3 * rows + i] = cd.pf.plunge_vphi[i];
    }
    __syncthreads();
    if (bins) v.hist = hist;
    if (rows) { v.pl_r = tab; v.pl_vt = tab + rows; v.pl_vr = tab + 2 * rows; v.pl_vp = tab + 3 * rows; }
    return v;
}

// The wave's 64 end-point records, written to LDS by finalize(), leave as 19 store instructions of 64 consecutive 8-byte
// units each: unit u of the region belongs to record u / 19, and wherever neighbouring lanes hold neighbouring rays (eight
// rows of a column in an 8 x 8 tile = 1216 bytes, the whole 9.7 KB in ray order) consecutive units are consecutive
// addresses -- whole 64- and 128-byte segments instead of 64 eight-byte pieces 152 bytes apart per instruction.  That is
// what makes a destination in pinned HOST memory practical (the stores cross PCIe as full-size packets) and takes the
// write excess out of the HBM path.  Every lane of the workgroup calls this (uniform control flow).
__device__ __forceinline__ void points_epilogue(const LdsView& v)
{
    if (!v.point) return;
    __syncthreads();
    const unsigned l = threadIdx.x & 63;
    const double* reg = v.point - l * POINT_UNITS;
    const uint64_t* addr = reinterpret_cast<const uint64_t*>(reg + 64 * POINT_UNITS);
#pragma unroll
    for (int t = 0; t < POINT_UNITS; ++t) {
        const unsigned u = l + 64u * t;
        const unsigned rec = (u * 3450u) >> 16;          // u / 19 for u < 1216 (checked exhaustively, tests/test_host_api.py)
        const unsigned f = u - rec * POINT_UNITS;
        const uint64_t base = addr[rec];
        if (base) reinterpret_cast<double*>(base)[f] = reg[u];
    }
}

// one global atomic per non-empty bin per workgroup
__device__ __forceinline__ void lds_epilogue(const Cold* cold, int lds_bins, const LdsView& v)
{
    if (!v.hist) return;
    __syncthreads();
    asm volatile("" : "+s"(cold));
    const Cold& cd = *cold;
    for (int i = threadIdx.x; i < lds_bins; i += blockDim.x) {
        const double h = v.hist[i];
        if (h != 0.0) atomicAdd(cd.lp_flux + i, h);
    }
}
#endif

// The launch parameters once more, read from the kernel-argument segment behind a pointer the optimiser cannot see through.
// What only the code after the step loop reads (the cold-block pointer, the statistics pointer, table pointers) then takes
// no scalar registers while the loop runs: with every Params field loaded at kernel entry the loop's hot block parked nine
// SGPRs in vector lanes and fetched them back on EVERY step (18 of the step's non-FP64 vector instructions).
#ifndef GR_LATE_PARAMS
#define GR_LATE_PARAMS 1
#endif
#if GR_LATE_PARAMS
// an offset of zero the optimiser cannot know to be zero (one SGPR), formed at kernel entry
#define GR_PARAMS_OPAQUE_ZERO(z) \
    int z = 0;                   \
    asm volatile("" : "+s"(z));
#define GR_PARAMS_AFTER_LOOP(p, pl, z)                                                                             \
    const Params __attribute__((address_space(4)))* pl##_k = (const Params __attribute__((address_space(4)))*)(      \
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + z);                  \
    Params pl = p;                                                                                                   \
    pl.cold = pl##_k->cold;                                                                                          \
    pl.stats = pl##_k->stats;
#else
#define GR_PARAMS_OPAQUE_ZERO(z)
#define GR_PARAMS_AFTER_LOOP(p, pl, z) const Params& pl = p;
#endif

// Which chunk of the rays workgroup b traces when the rays come in CALLER order (ray arrays, impact-parameter sets).
// Workgroups go to the 8 XCDs round-robin (b mod 8), so with chunk = b every ray pattern whose period is a multiple of 8
// chunks lands on the same XCDs: the α ≈ 0 column of a 1024-wide grid of impact parameters -- the rays that pass the polar
// axis and take five times the steps -- is chunk 7 / 8 of every 16, i.e. XCD 7 and XCD 0 only; those two ran for 38 ms while
// the other six had finished after 16 (measured wave by wave, scripts/wave_timeline.py -> profiles/r4_tangent_timeline.txt).
// The map below keeps b's group of 8 (so neighbouring chunks still run at about the same time) but rotates the chunk within
// the group by the base-8 digit sum of the group's index: periodic patterns of any period 8^k and clustered ones are both
// dealt to all eight XCDs.  A bijection on [0, 8 ceil(W / 8)); the launcher rounds the grid up to a multiple of 8.
__device__ __forceinline__ unsigned xcd_chunk(unsigned b)
{
    const unsigned r = b >> 3;
    unsigned sum = 0;
    for (unsigned t = r; t; t >>= 3) sum += t & 7u;
    return (r << 3) | ((b - sum) & 7u);
}

// ---- kernel 0: one ray per work-item ----
template <class Metric, int DISC>
#ifndef GR_LANE_MIN_WAVES
// (a composite geometry samples several conditions on every step: two waves per SIMD at most, no spills on its main path)
#define GR_LANE_MIN_WAVES ((DISC == GR_DISC_COMPOSITE || DISC == GR_DISC_MESH) && Metric::kLaneWavesPerSimd > 2 ? 2 : Metric::kLaneWavesPerSimd)
#endif
__global__ void __launch_bounds__(256, GR_LANE_MIN_WAVES) k_trace_lane(const Params p)
{
    GR_PARAMS_OPAQUE_ZERO(zoff)
#ifdef GR_WAVE_TIMELINE      // debug builds only (scripts/wave_timeline.py): when did this wave run, where, and how long was its longest ray
    const unsigned long long tl0 = wall_clock64();
    int tl_steps = 0;
    unsigned long long tl_extra = 0;      // a tabulated metric: patches copied << 16 | 640 ns units spent copying << 32 | ... evaluating from global memory << 48
#endif
    Metric m;
    m.load(p.cfg);
    // (the tangent build traces a ray with a PAIR of lanes, one per direction of the Jacobian: LANES_PER_RAY_LOG2 = 1)
    const unsigned chunk = p.xcd_spread ? xcd_chunk(blockIdx.x) : blockIdx.x;
    const int64_t gid = ((int64_t)chunk * blockDim.x + threadIdx.x) >> LANES_PER_RAY_LOG2;
    LaneStats<Metric, DISC> ls;
    const LdsView lds = lds_prologue<ColdSel<Metric>::kBytesPerThread>(p);
#if GR_HAS_MESH
    if constexpr (DISC == GR_DISC_MESH) {
        // A mesh: the wave stays together until its last ray has ended -- after every step the tests that are due are run by the
        // whole wave (Ray::mesh_phase), finished lanes helping.  Otherwise the tail of the launch is a few rays that wind round
        // the hole inside the bounding box, each walking ~100 candidate triangles per step alone (25.2 -> ms for 3840 triangles
        // at 1024², DESIGN_measurements.md §M15)
        Ray<Metric, DISC> ray;
        const typename ColdSel<Metric>::type cs = cold_store_of<ColdSel<Metric>>(p);
        bool live = gid < p.n;
        if (live) {
            ray.init(m, p, tile_swizzle(cold_of(p), gid));
            ray.mesh_coop = 1;
        }
        while (__any(live)) {
            bool fin = true;
            if (live) fin = ray.step(m, p, cs);
            fin = ray.mesh_phase(p, live, fin);
            if (fin) live = false;
        }
        __builtin_amdgcn_wave_barrier();
        if (gid < p.n) {
            GR_PARAMS_AFTER_LOOP(p, pl, zoff)
            ray.finalize(m, pl, lds, cs);
            ls.add(ray);
#ifdef GR_WAVE_TIMELINE
            tl_steps = ray.nacc + ray.nrej;
#endif
        }
    } else
#endif
    if (gid < p.n) {
        Ray<Metric, DISC> ray;
        const typename ColdSel<Metric>::type cs = cold_store_of<ColdSel<Metric>>(p);
        unsigned long long tab_t0 = 0;
        if constexpr (ColdSel<Metric>::kTab) tab_t0 = wall_clock64();
        ray.init(m, p, tile_swizzle(cold_of(p), gid));
        while (!ray.step(m, p, cs)) {}
        // In a one-wave workgroup finalize() lays the end-point record down in the LDS bytes other lanes of this wave use as
        // cold storage inside step() (lds_prologue): every lane must have LEFT the loop before any lane stores.  The
        // structured loop exit guarantees that today; the barrier states it, so that no later pass may sink finalize() into a
        // per-lane exit block.  It emits no instruction.
        __builtin_amdgcn_wave_barrier();
        GR_PARAMS_AFTER_LOOP(p, pl, zoff)
        ray.finalize(m, pl, lds, cs);
        if constexpr (ColdSel<Metric>::kTab && LANES_PER_RAY_LOG2 == 0) {
            // A tabulated metric: what a tile costs is how long its wave lived, not how many steps its rays took -- a wave at
            // the shadow's edge, its lanes in a dozen different patches, spends 4x the time per step of one inside a single patch
            // (scripts/wave_timeline.py, WT_TAB=1).  The longest-first order of the next launches is learned from the clock.
            const Cold& cdt = cold_of(pl);
            if (cdt.tile_cost) {
                const int64_t ti = ray.tile_cost_index(cdt);
                const unsigned long long dt = (wall_clock64() - tab_t0) >> 4;
                if (ti >= 0) cdt.tile_cost[ti] = (uint32_t)(dt < 1ull ? 1ull : dt > 0xffffffull ? 0xffffffull : dt);
            }
        }
        if (LANES_PER_RAY_LOG2 == 0 || tan_dir() == 0) ls.add(ray);      // a ray is counted once
#ifdef GR_WAVE_TIMELINE
        tl_steps = ray.nacc + ray.nrej;
        if constexpr (ColdSel<Metric>::kTab) {
            typedef int __attribute__((address_space(3))) lds_int_t;
            const unsigned nc = (unsigned)((lds_int_t*)cs.tab)[12], tc = (unsigned)((lds_int_t*)cs.tab)[13] >> 6, tf = (unsigned)((lds_int_t*)cs.tab)[14] >> 6;
            tl_extra = ((unsigned long long)(nc > 0xffffu ? 0xffffu : nc) << 16) | ((unsigned long long)(tc > 0xffffu ? 0xffffu : tc) << 32)
                       | ((unsigned long long)(tf > 0xffffu ? 0xffffu : tf) << 48);
        }
#endif
    }
#ifdef GR_WAVE_TIMELINE
    if (p.queue) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_down(tl_steps, off, 64); tl_steps = o > tl_steps ? o : tl_steps; }
        if ((threadIdx.x & 63) == 0) {
            unsigned hw = 0, xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o = p.queue + 4ull * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
            o[0] = tl0; o[1] = wall_clock64(); o[2] = ((unsigned long long)xcc << 32) | hw; o[3] = (unsigned long long)tl_steps | tl_extra;
        }
    }
#endif
    points_epilogue(lds);
    lds_epilogue(p.cold, p.lds_bins, lds);
    ls.flush(p.stats);
}

// ---- kernel 1: persistent grid with wave-ballot refill ----
#ifndef GR_PERSISTENT_MIN_WAVES
#define GR_PERSISTENT_MIN_WAVES Metric::kMinWavesPerSimd
#endif
template <class Metric, int DISC>
__global__ void __launch_bounds__(256, GR_PERSISTENT_MIN_WAVES) k_trace_persistent(const Params p)
{
    Metric m;
    m.load(p.cfg);
    Ray<Metric, DISC> ray;
    LaneStats<Metric, DISC> ls;
    bool active = false, pending = false, queue_empty = false;
    const int lane = threadIdx.x & 63;
    const int threshold = p.refill_threshold;
    const LdsView lds = lds_prologue<ColdSel<Metric, false>::kBytesPerThread>(p);
    const typename ColdSel<Metric, false>::type cs = cold_store_of<ColdSel<Metric, false>>(p);

    for (;;) {
        const unsigned long long act = __ballot(active);
        const int n_idle = 64 - __popcll(act);
        if (n_idle >= threshold || act == 0ull) {
            if (__ballot(pending)) {
                if (pending) {
                    ray.finalize(m, p, lds);
                    ls.add(ray);
                    pending = false;
                }
            }
            if (!queue_empty) {
                const unsigned long long idle = __ballot(!active);
                const int n = __popcll(idle);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(p.queue, (unsigned long long)n);
                base = __shfl(base, 0, 64);
                const int64_t mine = (int64_t)base + __popcll(idle & ((1ull << lane) - 1ull));
                if (!active && mine < p.n) {
                    ray.init(m, p, tile_swizzle(cold_of(p), mine));
                    active = true;
                }
                if ((int64_t)base + n >= p.n) queue_empty = true;
            }
            if (__ballot(active) == 0ull) break;
        }
        if (active) {
            if (ray.step(m, p, cs)) {
                active = false;
                pending = true;
            }
        }
    }
    lds_epilogue(p.cold, p.lds_bins, lds);
    ls.flush(p.stats);
}

// ---- geodesics with every accepted step saved: one ray per lane, ray j writes rows of 9 doubles
// (λ, x[4], v[4]) into path[j * cap ...] ----
template <class Metric, int DISC>
__global__ void __launch_bounds__(64) k_trace_path(const Params p, double* path, int64_t cap, unsigned long long* n_rows)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p.n) return;
    Metric m;
    m.load(p.cfg);
    Ray<Metric, DISC> ray;
    ray.init(m, p, j);
    double* const mine = path + 9 * cap * j;
    int64_t n = 0;
    auto save = [&]() {
        if (n < cap) {
            double* row = mine + 9 * n;
            row[0] = ray.t;
#pragma unroll
            for (int q = 0; q < 4; ++q) { row[1 + q] = ray.x[q]; row[5 + q] = ray.v[q]; }
        }
        ++n;
    };
    save();
    for (;;) {
        const int before = ray.nacc;
        const bool fin = ray.step(m, p);
        if (fin) break;
        if (ray.nacc != before) save();
    }
    const LdsView no_lds{ nullptr, nullptr, nullptr, nullptr, nullptr };
    ray.finalize(m, p, no_lds);     // resolves a pending event and writes the endpoint record
    save();
    n_rows[j] = (unsigned long long)n;
}

// ---- apply(pf, points) ----
template <class Metric>
__global__ void __launch_bounds__(256) k_apply_pf(const Params p, const gr_point* pts, double max_time, double* out)
{
    Metric m;
    m.load(p.cfg);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const gr_point gp = pts[i];
    const Cold& cd = *p.cold;
    bool pass = true;
    if (cd.pf.filter_id == GR_FILTER_EARLY_TERM) pass = gp.lambda_max < max_time;
    else if (cd.pf.filter_id == GR_FILTER_INTERSECTED) pass = gp.status == GR_STATUS_INTERSECTED_WITH_GEOMETRY;
    double val = cd.pf.fill;
    if (pass) {
        if (cd.pf.pf_id == GR_PF_AFFINE_TIME) val = gp.lambda_max;
        else if (cd.pf.pf_id == GR_PF_STATUS) val = (double)gp.status;
        else if (cd.pf.pf_id == GR_PF_WINDING) val = (double)((uint32_t)gp.flags >> 16);
        else if (cd.pf.pf_id == GR_PF_RADIUS) val = gp.x[1] * ::fabs(::sin(gp.x[2]));
        else {
            const LdsView no_lds{ nullptr, nullptr, nullptr, nullptr, nullptr };
            real xi[4], vi[4], xe[4], ve[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xi[q] = gp.x_init[q]; vi[q] = gp.v_init[q]; xe[q] = gp.x[q]; ve[q] = gp.v[q]; }
            val = redshift_pf(m, p, cd, no_lds, xi, vi, xe, ve);
        }
    }
    out[i] = val;
}


#ifndef GR_NO_LAUNCHER      // scripts/kernel_probe.sh instantiates single kernels without the dispatch table
// launch knobs handed over by the host side (gr_ctx is not visible here)
struct LaunchKnobs {
    int kernel;              // 0 = lane, 1 = persistent
    int block;
    int n_cu;
    int waves_per_simd;      // 0 = from the occupancy query
    unsigned long long* queue;   // zeroed work counter for the persistent kernel
};

template <class Metric, int DISC>
hipError_t launch_tmpl(const LaunchKnobs& k, Params& p, hipStream_t stream)
{
    const int block = k.block;
    if (k.kernel != 0) p.lds_points = 0;     // the persistent kernel refills lanes one by one: no wave-wide moment to send records
    // a kernel that parks its stage accelerations in LDS (10 KB per wave) leaves the plunging table in L2: twelve one-wave
    // workgroups per CU cannot each hold a copy as well (staged and L2-served look-ups measured equal, gradus_mi355x.hip)
    if (k.kernel == 0 && ColdSel<Metric, true>::kPark > 0) p.lds_plunge_rows = 0;
    // (so does a tabulated metric: its LDS is the patch cache, 18 KB per wave)
    if (ColdSel<Metric, true>::kTab) p.lds_plunge_rows = 0;
    const size_t cold_b = (k.kernel == 0 ? ColdSel<Metric, true>::kBytesPerThread : ColdSel<Metric, false>::kBytesPerThread) * (size_t)block;
    const size_t point_b = p.lds_points ? kPointLdsBytesPerThread * (size_t)block : 0;
    // one-wave workgroups: the end-point records reuse the cold lane storage (lds_prologue)
    size_t lds = sizeof(double) * ((size_t)p.lds_bins + 4 * (size_t)p.lds_plunge_rows)
                 + (block == 64 ? (cold_b > point_b ? cold_b : point_b) : cold_b + point_b);
    {   // a tabulated metric's patch caches, one per wave, behind everything else (16-byte aligned)
        const size_t tab_w = k.kernel == 0 ? ColdSel<Metric, true>::kTabBytesPerWave : ColdSel<Metric, false>::kTabBytesPerWave;
        lds = (lds + 15) & ~(size_t)15;
        p.lds_tab_off = (int32_t)lds;
        lds += tab_w * (size_t)((block + 63) / 64);
    }
#ifdef GR_LANE_ONLY
    {
#else
    if (k.kernel == 0) {
#endif
        int64_t grid = ((p.n << LANES_PER_RAY_LOG2) + block - 1) / block;
        if (p.xcd_spread) grid = (grid + 7) / 8 * 8;       // xcd_chunk permutes whole groups of 8 workgroups
        hipLaunchKernelGGL((k_trace_lane<Metric, DISC>), dim3((unsigned)grid), dim3(block), lds, stream, p);
    }
#ifndef GR_LANE_ONLY
    else {
        int per_cu = 0;
        if (lds > ((size_t)64 << 10)) {      // four patch caches of a tabulated metric: more dynamic LDS than a kernel may have unasked
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace_persistent<Metric, DISC>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (ea != hipSuccess) return ea;
        }
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_persistent<Metric, DISC>, block, lds);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        if (k.waves_per_simd > 0) {
            const int want = k.waves_per_simd * 256 / block;
            if (want >= 1 && want < per_cu) per_cu = want;
        }
        int64_t grid = (int64_t)k.n_cu * per_cu;
        const int64_t need = (p.n + block - 1) / block;
        if (grid > need) grid = need;
        if (grid < 1) grid = 1;
        p.queue = k.queue;
        e = hipMemsetAsync(p.queue, 0, sizeof(unsigned long long), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_trace_persistent<Metric, DISC>), dim3((unsigned)grid), dim3(block), lds, stream, p);
    }
#endif
    return hipGetLastError();
}

// DISC is the geometry id itself (GR_DISC_*): compile-time in the kernels
template <class Metric>
hipError_t launch_metric(const LaunchKnobs& k, Params& p, hipStream_t stream)
{
    switch (p.cfg.disc_id) {
    case GR_DISC_THIN: return launch_tmpl<Metric, GR_DISC_THIN>(k, p, stream);
    case GR_DISC_SHAKURA_SUNYAEV: return launch_tmpl<Metric, GR_DISC_SHAKURA_SUNYAEV>(k, p, stream);
    case GR_DISC_TABULATED: return launch_tmpl<Metric, GR_DISC_TABULATED>(k, p, stream);
    case GR_DISC_DATUM: return launch_tmpl<Metric, GR_DISC_DATUM>(k, p, stream);
    case GR_DISC_ELLIPTICAL: return launch_tmpl<Metric, GR_DISC_ELLIPTICAL>(k, p, stream);
    case GR_DISC_PRECESSING_THIN: return launch_tmpl<Metric, GR_DISC_PRECESSING_THIN>(k, p, stream);
    case GR_DISC_COMPOSITE: return launch_tmpl<Metric, GR_DISC_COMPOSITE>(k, p, stream);
#if GR_HAS_MESH
    case GR_DISC_MESH: return launch_tmpl<Metric, GR_DISC_MESH>(k, p, stream);
#endif
    default: return launch_tmpl<Metric, GR_DISC_NONE>(k, p, stream);
    }
}

template <class Metric>
hipError_t launch_path_metric(const Params& p, double* d_path, int64_t cap, unsigned long long* d_n, hipStream_t stream)
{
    const unsigned grid = (unsigned)((p.n + 63) / 64);
#define GR_PATH_LAUNCH(D) hipLaunchKernelGGL((k_trace_path<Metric, D>), dim3(grid), dim3(64), 0, stream, p, d_path, cap, d_n)
    switch (p.cfg.disc_id) {
    case GR_DISC_THIN: GR_PATH_LAUNCH(GR_DISC_THIN); break;
    case GR_DISC_SHAKURA_SUNYAEV: GR_PATH_LAUNCH(GR_DISC_SHAKURA_SUNYAEV); break;
    case GR_DISC_TABULATED: GR_PATH_LAUNCH(GR_DISC_TABULATED); break;
    case GR_DISC_DATUM: GR_PATH_LAUNCH(GR_DISC_DATUM); break;
    case GR_DISC_ELLIPTICAL: GR_PATH_LAUNCH(GR_DISC_ELLIPTICAL); break;
    case GR_DISC_PRECESSING_THIN: GR_PATH_LAUNCH(GR_DISC_PRECESSING_THIN); break;
    case GR_DISC_COMPOSITE: GR_PATH_LAUNCH(GR_DISC_COMPOSITE); break;
#if GR_HAS_MESH
    case GR_DISC_MESH: GR_PATH_LAUNCH(GR_DISC_MESH); break;
#endif
    default: GR_PATH_LAUNCH(GR_DISC_NONE); break;
    }
#undef GR_PATH_LAUNCH
    return hipGetLastError();
}

template <class Metric>
hipError_t launch_apply_metric(const Params& p, const gr_point* pts, double max_time, double* out, hipStream_t stream)
{
    const int block = 256;
    const int64_t grid = (p.n + block - 1) / block;
    hipLaunchKernelGGL((k_apply_pf<Metric>), dim3((unsigned)grid), dim3(block), 0, stream, p, pts, max_time, out);
    return hipGetLastError();
}
#endif  // GR_NO_LAUNCHER

}  // namespace
}  // namespace GR_NS

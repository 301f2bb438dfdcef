// ab_tile.hpp — tile machinery shared by flux_kernel (ab_kernels.hip), turb_kernel (ab_turb_kernels.hip) and the sea-ice kernels: LDS
// budget of a block, forecast of a cell's divergent predicates, counting sort of a tile's cells by bucket, tile size on small grids.
// The method is described at flux_kernel.  Internal; include after ab_physics.hpp and ab_launch.hpp.
#pragma once
#include <atomic>

namespace ab {

// Resident blocks per CU = waves per SIMD, at least.  Every flux kernel fits 128 VGPRs without scratch, so four blocks of 256
// lanes share a CU (and its 160 KB of LDS): measured 1.5-4 % faster than three blocks with larger tiles and the height constants
// parked in VGPRs (profiles/r1_notes.md).  Kernels that need fewer registers take more (Tile::kOcc).
#ifndef AB_WAVES_PER_EU
#define AB_WAVES_PER_EU 4
#endif

constexpr int kBuckets = 16;
static_assert(kBlock >= 2 * fm::kLogN, "fm::lds_tables_init() copies one table entry per thread");

// detach a wave-uniform value from the scalar-load tuple it arrived in (see flux_kernel, phase 3)
template <class T> __device__ __forceinline__ void uniform_scalar(T &x)
{
    T y;   // a real copy: a tied "+s" operand is coalesced back into the tuple
    if (sizeof(T) == 8)
        asm volatile("s_mov_b64 %0, %1" : "=s"(y) : "s"(x));
    else
        asm volatile("s_mov_b32 %0, %1" : "=s"(y) : "s"(x));
    x = y;
}
// The kernel arguments arrive as 8- and 16-register scalar loads, and the register allocator spills and reloads such a tuple
// as a whole (16 v_readlane_b32 per reload, inside the iteration loop): the loop invariants of the iteration are detached
// from it one by one.  MI355X, 4320x3600 fp64 nb_iter=5: COARE3p6 + skin 3.49 -> 3.46 ms, no skin 1.86 -> 1.81 ms.
template <class R> __device__ __forceinline__ Heights<R> detached(const Heights<R> &g)
{
    Heights<R> h = g;
#ifndef AB_ARGS_AS_LOADED
    uniform_scalar(h.zt); uniform_scalar(h.zu); uniform_scalar(h.log_zt); uniform_scalar(h.log_zu); uniform_scalar(h.log_10);
    uniform_scalar(h.log_ztu); uniform_scalar(h.log_zu10); uniform_scalar(h.fg_ca); uniform_scalar(h.inv_zu);
    uniform_scalar(h.zt_o_zu); uniform_scalar(h.zt_eq_zu); uniform_scalar(h.fg_cb);
#endif
    return h;
}     // 4 stability bins x 4 warm-layer bins
// MIXED: the AB_F32_MIXED flux kernels (R = float with fp64 anchors): one more field (the low part of theta), the fp64 math tables
// and the e_sat table next to the fp32 psi tables
// POLICY: kTileFlux = the flux kernels of ab_kernels.hip (occupancies tuned per kernel family, piecewise tables in LDS); kTileFour = four
// waves per SIMD and no LDS tables (turb_kernel, ice_kernel).  A template parameter, not a macro redefined per translation unit: the same
// specialisation must mean the same thing everywhere (round-2 advisory: ODR).
constexpr int kTileFlux = 0, kTileFour = 1;
template <class R, int ALGO, bool SKIN, bool MIXED = false, int POLICY = kTileFlux> struct Tile {
    static constexpr int kFields = (SKIN ? 8 : 6) + (MIXED ? 1 : 0);   // flux: sst theta q_zt u v slp [qsw rlw] [theta_lo] ; turb: 8 / 6 too
    // Waves per SIMD (= resident blocks per CU) a kernel is built for.  The fp64 kernels with the skin schemes need 107-127 VGPRs:
    // four.  Without them 72-95 VGPRs: five, on two-round tiles (-3...-4 % COARE, -1 % ECMWF; config 2 -3 %).  The fp32 kernels,
    // at 36-69 VGPRs once their psi functions come from the LDS tables: seven with the skin schemes (ECMWF: six), eight without,
    // on two-round tiles (-3...-8 % at five / six, another -2...-4 % at six / seven).  Same-box A/Bs in profiles/r2_notes.md.
#ifndef AB_NOSKIN_OCC
#define AB_NOSKIN_OCC 5
#endif
#ifndef AB_COARE3P0_NOSKIN_OCC   // COARE 3.0 without the skin schemes: 96 VGPRs + 36 B of scratch at five waves once its psi comes from the L1 tables; four waves, no scratch: -12 % against -8 %
#define AB_COARE3P0_NOSKIN_OCC 4
#endif
// COARE 3.6 without the skin schemes (round 4): its psi tables indexed by the bits of their argument (no logarithm: 16 % of its short
// iteration) are 2 x 6 400 B: two-round tiles fit beside them at FOUR blocks per CU, not at five
#ifndef AB_COARE3P6_NOSKIN_OCC
#ifdef AB_NOSKIN_PSI_LOGTAB
#define AB_COARE3P6_NOSKIN_OCC AB_NOSKIN_OCC
#else
#define AB_COARE3P6_NOSKIN_OCC 4
#endif
#endif
#ifndef AB_F32_OCC
#define AB_F32_OCC 7
#endif
#ifndef AB_F32_ECMWF_OCC
#define AB_F32_ECMWF_OCC 6
#endif
#ifndef AB_F32_NOSKIN_OCC
#define AB_F32_NOSKIN_OCC 8
#endif
    // The mixed kernels (fp32 work, fp64 anchors): ECMWF + skin 78 VGPRs: six; COARE + skin 80 with scratch at six: five; without the
    // skin schemes 54-59: seven (the LDS of the fp64 math tables and the e_sat table leaves room for seven two-round tiles, not eight).
#ifndef AB_MIXED_OCC
#define AB_MIXED_OCC 6
#endif
#ifndef AB_MIXED_COARE_OCC
#define AB_MIXED_COARE_OCC 5
#endif
#ifndef AB_MIXED_NOSKIN_OCC
#define AB_MIXED_NOSKIN_OCC 7
#endif
    static constexpr int kOcc = POLICY == kTileFour ? 4
                                : MIXED ? (SKIN ? (ALGO == 4 ? AB_MIXED_OCC : AB_MIXED_COARE_OCC) : AB_MIXED_NOSKIN_OCC)
                                : sizeof(R) == 8 ? (SKIN ? AB_WAVES_PER_EU : ((ALGO == 1 && AB_COARE3P0_NOSKIN_OCC) ? AB_COARE3P0_NOSKIN_OCC : (ALGO == 2 ? AB_COARE3P6_NOSKIN_OCC : AB_NOSKIN_OCC)))
                                               : (SKIN ? (ALGO == 4 ? AB_F32_ECMWF_OCC : AB_F32_OCC) : AB_F32_NOSKIN_OCC);
    static constexpr int kWaves = kOcc * 256 / kBlock;      // resident blocks per CU
    // The kWaves blocks of a CU share its 160 KB of LDS (allocated in 512-byte granules: tools/micro/lds_granule.hip).  Per block, besides
    // the tile (fields + a 2-byte index per cell): the sort's counters and the queue head (136 B; 160 reserved), the tables of the fp64
    // log / exp and the constants (fm::s_logtab 1024 B, s_exptab 512 B, s_ctab 96 B: 1 632 B), and, in the translation units that define
    // AB_PSI_LDS_TABLES, the piecewise tables of ab_physics.hpp.  fp64: the e_sat table (1 536 B, kernels with the skin schemes) + either
    // the Kansas psi_m / psi_h pair (2 x 1 792 B: ECMWF, ANDREAS; ANDREAS + 800 B: its stable psi_m) or, COARE: with the skin schemes the cool skin's g(u) (1 280 B) and the
    // blended psi_h (2 560 B; psi_m through L1, ab_gtables.hpp), without them psi_m and psi_h (2 x 2 560 B); fp32: the three psi tables
    // (1 536 B; + e_sat: mixed).  The fp64 COARE kernels with the skin schemes come out at exactly two rounds with 24 B to spare.
#ifdef AB_NOSKIN_PSI_LOGTAB
#define AB_COARE_NOSKIN_TAB_BYTES 5120    // the blended psi_m / psi_h in s = LOG(y), degree 9 x 32 intervals (round 3)
#else
#define AB_COARE_NOSKIN_TAB_BYTES 12800   // ... indexed by the bits of y, degree 9 x 8 intervals per binade (round 4)
#endif
#ifdef AB_PSI_NOBITS
#define AB_COARE_SKIN_TAB_BYTES 3840   // g(u) 1 280 B + the blended psi_h 2 560 B (round 3)
#else
#define AB_COARE_SKIN_TAB_BYTES 3584   // the cool skin's T(u) table (degree 7 x 56 intervals); psi_m and psi_h through L1, indexed by the bits of their argument (round 4)
#endif
    static constexpr int kPsiTabBytes = POLICY == kTileFour ? 0 : (sizeof(R) == 8 ? (SKIN ? 1536 : 0) + ((ALGO == 1 || ALGO == 2) ? (SKIN ? AB_COARE_SKIN_TAB_BYTES : AB_COARE_NOSKIN_TAB_BYTES) : (ALGO == 5 ? 4384 : 3584)) : (MIXED ? 3072 : 1536));
    static constexpr int kBudget = 160 * 1024 / kWaves - 160 - ((sizeof(R) == 8 || MIXED) ? 1632 : 0) - kPsiTabBytes;
    static constexpr int kRounds = kBudget / (kBlock * (kFields * (int)sizeof(R) + 2)); // f64: 2 (skin, 4 blocks) / 2 (5 blocks); f32: 2
    static constexpr int kCells = kRounds * kBlock;
    static constexpr int kGroups = kCells / 64;
};

// Forecast of (warm layer gains heat, stable) for a cell, with an uncertainty band around each threshold so that the
// doubtful cells sit together at the bucket borders.  fp32, hardware transcendentals: < 1 % of the work of one cell.
template <int ALGO, bool SKIN>
__device__ __forceinline__ int forecast_bucket(float sst, float theta, float q, float uu, float vv, float slp, float qsw,
                                               float rlw, bool wl_load, float dTprev, float Hzprev)
{
    using F = float;
    using M = Mth<F>;
    F Ts = SKIN ? sst - 0.25f : sst;
    if (SKIN && wl_load) Ts += dTprev;
    const F qs = 0.98f * q_sat<F>(Ts, slp);
    F dthv = theta * (1.f + 0.608f * q) - Ts * (1.f + 0.608f * qs);   // sign of the bulk Richardson number
    int wbin = 0;
    // WL_ECMWF runs its ten-pass erosion (a square root and a division per pass) only where a warm layer EXISTS and loses heat
    // (mod_skin_ecmwf.f90:210-225; exact shortcuts in wl_ecmwf): worth a bin from the second record of a series on
    const bool ecmwf_layer = SKIN && ALGO == 4 && wl_load && dTprev > 0.f;
    if ((SKIN && ALGO != 4) || ecmwf_layer) {
        const F w2 = uu * uu + vv * vv;
        const F wnd = M::sqrt(w2), Ub = vmax(M::sqrt(w2 + 0.25f), 0.5f);
        const F Cx = dthv > 0.f ? 0.96e-3f : 1.38e-3f;
        const F t2 = Ts * Ts;
        const F qns = 1.2f * Ub * Cx * (1005.f * (theta - Ts) + 2.45e6f * (q - qs)) + 0.98f * (rlw - 5.67e-8f * t2 * t2);
        if (ALGO == 4) {
            const F qabs = 0.635176f * qsw + qns;   // absorbed fraction of the fixed 3 m layer (mod_skin_ecmwf.f90:152)
            wbin = qabs < -40.f ? 3 : (qabs < 0.f ? 2 : (qabs < 40.f ? 1 : 0));
        } else {   // WL_COARE runs its depth solve only where the layer gains heat (mod_skin_coare.f90:171-185)
            // first record of a series: the layer depth is its initial 20 m, the absorbed fraction a constant (three exponentials less)
            const F fabs = wl_load ? wl_absorb<F>(vmax(vmin(Hzprev, 20.f), 0.1f)) : 0.76714447f;
            const F qabs = fabs * qsw + qns;
            wbin = qabs < -40.f ? 0 : (qabs < 0.f ? 1 : (qabs < 40.f ? 2 : 3));
            if (wl_load && M::abs(dTprev) >= 1.e-6f && wbin < 2) wbin = 2;
#ifndef AB_BUCKET_NOSHIFT
            if (qabs > 0.f) {      // the warming of this record shifts the stability: estimate of mod_skin_coare.f90:199-224
                const F alpha = alpha_sw<F>(sst);
                const F tac = vmax(1.44e-3f * Ub * wnd, 0.002f) * 3600.f;
                const F qac = qabs * 3600.f;
                const F hz = vmax(vmin(20.f, M::sqrt(5.422e-1f * M::rcp(alpha)) * tac * M::rsqrt_pos(qac)), 0.1f);
                F dT = M::sqrt(2.942e-2f * alpha) * 3.687e-6f * qac * M::sqrt(qac) * M::rcp(tac);
                if (hz < 1.f) dT *= M::rcp(hz);
                dthv -= dT * (1.f + 11.5f * qs);
            }
#endif
        }
    }
    const int sbin = dthv < -0.3f ? 0 : (dthv < 0.f ? 1 : (dthv < 0.3f ? 2 : 3));
    // stability is the major key (it matters in every iteration, the warm layer only in those that commit); snake order:
    // neighbouring buckets differ in one predicate only.  Measured: -4.6 % against warm layer major; band widths
    // 0.15-0.6 K / 40-80 W/m2 and 8 x 2 bins are all within 2 % (profiles/r1_notes.md)
    return sbin * 4 + ((sbin & 1) ? 3 - wbin : wbin);
}

// Counting sort of the tile's cells by bucket with LDS atomics (the LDS pipe is all but idle in these kernels, the VALU is the
// bottleneck: the previous sort by packed 16-bit histograms and wave scans cost about 150 VALU slots per cell, this one about 7):
//   tile_sort_reset   before the block's first barrier: zero the counters;
//   tile_sort_note    phase 1, per cell: rank = atomicAdd(counter[bucket]) (ds_add_rtn_u32); rank and bucket are parked in the
//                     cell's own s_inv entry (11 + 4 bits: tiles have at most 1280 cells);
//   tile_sort_place   after the barrier that ends phase 1: wave 0 turns the counts into first slots, every thread takes its
//                     cells' keys into registers, barrier, s_inv[first slot of the bucket + rank] = cell.
// The order of the cells INSIDE a bucket depends on the order the atomics were served in, so the composition of a wave may differ
// from run to run; the arithmetic of a cell does not depend on its neighbours, so every output bit is the same
// (tests/test_gpu_regroup.py).
// Lanes of one wave-instruction that hit the same counter are served one after the other (without the skin schemes only four of the
// sixteen buckets exist).  kSortSub counters per bucket, picked by the low bits of the lane, would spread them: measured, four
// sub-counters are no faster than one (coare3p6 skin 2.831 / 2.852 ms, no skin n=8 1.926 / 1.943 ms; profiles/r2_notes.md).
#ifndef AB_SORT_SUB
#define AB_SORT_SUB 1
#endif
constexpr int kSortRankBits = 11, kSortSub = AB_SORT_SUB, kSortCounters = kBuckets * kSortSub;
__device__ __forceinline__ void tile_sort_reset(unsigned *s_cnt, int tid)
{
    if (tid < kSortCounters) s_cnt[tid] = 0u;
}
__device__ __forceinline__ void tile_sort_note(unsigned *s_cnt, unsigned short *s_inv, int j, int bkt, int tid)
{
    const unsigned rank = atomicAdd(&s_cnt[bkt * kSortSub + (tid & (kSortSub - 1))], 1u);
    s_inv[j] = (unsigned short)(rank | ((unsigned)bkt << kSortRankBits));
}
struct BlockSync {
    __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
// `sync`: the barrier of the threads that share the tile (the block's; a team's in flux_kernel_cu)
template <int MAXROUNDS, class Sync = BlockSync>
__device__ __forceinline__ void tile_sort_place(const unsigned *s_cnt, unsigned *s_base, unsigned short *s_inv, int tid, int rounds, Sync sync = Sync())
{
    static_assert(kSortCounters <= 64, "one wave scans the counters");
    if (tid < 64) {   // exclusive prefix over the counters, bucket-major (wave 0)
        const unsigned c = tid < kSortCounters ? s_cnt[tid] : 0u;
        unsigned incl = c;
#pragma unroll
        for (int d = 1; d < kSortCounters; d <<= 1) {
            const unsigned t = __shfl_up(incl, d, 64);
            if (tid >= d) incl += t;
        }
        if (tid < kSortCounters) s_base[tid] = incl - c;
    }
    unsigned key[MAXROUNDS];
#pragma unroll
    for (int r = 0; r < MAXROUNDS; ++r) key[r] = (r < rounds) ? (unsigned)s_inv[r * kBlock + tid] : 0u;
    sync();
    const unsigned sub = (unsigned)tid & (kSortSub - 1);
#pragma unroll
    for (int r = 0; r < MAXROUNDS; ++r)
        if (r < rounds)
            s_inv[s_base[(key[r] >> kSortRankBits) * kSortSub + sub] + (key[r] & ((1u << kSortRankBits) - 1u))] = (unsigned short)(r * kBlock + tid);
}

// blocks the chip holds at once: AB_WAVES_PER_EU per CU (one wave of each block per SIMD).  The CU count is cached per DEVICE, in atomics:
// the shards of a sharded session call this from their worker threads, each with its own current device (round-2 advisory)
static inline long resident_block_slots(int occ = AB_WAVES_PER_EU)
{
    static std::atomic<int> cus_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::atomic<int> &slot = cus_of[dev & 63];
    int cus = slot.load(std::memory_order_relaxed);
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slot.store(cus, std::memory_order_relaxed);
    }
    return (long)cus * (occ * 256 / kBlock);
}

// full tiles when the grid fills the chip several times over; smaller ones on small grids so that every CU gets work
// (a 360x180 grid is 127 two-round tiles for 256 CUs, but 254 one-round tiles)
static inline int tile_rounds(long n, int max_rounds, int occ = AB_WAVES_PER_EU)
{
    long rounds = n / ((long)kBlock * resident_block_slots(occ));
#ifdef AB_FORCE_ROUNDS
    rounds = AB_FORCE_ROUNDS;
#endif
    return (int)(rounds < 1 ? 1 : (rounds > max_rounds ? max_rounds : rounds));
}

}  // namespace ab

// lchd_sweep_common.h -- what the sweep kernel families share (lchd_sweep.hip, lchd_sweep_team.hip, lchd_sweep_wide.hip, the record pass
// and the launch logic in lchd_kernels.hip): distance / key modes, tuning macros, the pass's small-pair rule, the status hand-over
// without a fence, the generic statistical distances and the per-pair weight-function registers.
#pragma once
#include <algorithm>
#include <type_traits>

#include "lchd_kcommon.h"
#include "lchd_team_tile.h"

namespace lchd {

// ------------------------------------------------------------------------------------------------
// K2: the sweep.  One wavefront per anchor pair, four pairs per 256-thread workgroup.
//
// S = sum_k [F(t_{k+1}) - F(t_k)] * H(state after k events), t_0 = 0, t_{M+1} = inf, where the events are
// the merged non-anchor points of both environments (the reference's two-pointer
// loop src/locohd.rs:97-223 evaluates exactly this sum; cross-list ties collapse because a zero-width
// interval contributes exactly 0).
//
// Events are processed in tiles of 384: lane l owns ceil(T/64) <= 6 consecutive merged events of the tile, found with a
// merge-path binary search in LDS.  A packed (16-bit fields) wavefront prefix scan of the per-lane
// category histograms gives every lane the exact integer category counts at its first event; it then
// walks its events sequentially with the per-category state in registers.
//
// MODE_H2U / MODE_H2W: Hellinger distance with exponent 2 (the default, src/locohd.rs:365-370), unit /
//   arbitrary category weights.  The per-lane state is just the packed integer category counts plus the
//   running Bhattacharyya numerator D = sum_c sqrt(a_c b_c); an event touches one category, so D is updated
//   in O(1) from an LDS table of sqrt(k) and H^2 = 1 - D / sqrt(N_a N_b).  Where that cancellation form would
//   lose accuracy (H^2 < kExactH2Below = 1e-6) the literal sum_c (sqrt(a_c/N_a) - sqrt(b_c/N_b))^2 / 2 is evaluated instead,
//   which also gives exactly 0 for identical environments.
// MODE_GEN: every other StatisticalDistance (statistical_distances.rs:4-78): weighted counts in registers,
//   normalised like pmf.rs:65-83, distance through one out-of-line call.
// ------------------------------------------------------------------------------------------------
enum { MODE_H2U = 0, MODE_H2W = 1, MODE_GEN = 2 };
// where F(t) comes from: the environment keys already are F values / inline CDFs only / any CDF
enum { F_KEY = 0, F_FAST = 1, F_ANY = 2 };
#ifndef LCHD_PASS1_FUSED
#define LCHD_PASS1_FUSED 1  // k_sweep: the chunk histogram is one fixed-trip loop over the lane's points
#endif
#ifndef LCHD_LDS_COUNTS
#define LCHD_LDS_COUNTS 1   // k_sweep (Hellinger-2, LDS tables, > 12 category slots): per-lane category counts live in LDS during the event loop
#endif
#ifndef LCHD_HEADS_REREAD
#define LCHD_HEADS_REREAD 1   // k_sweep: both list heads are re-read from LDS after every event
#endif
#ifndef LCHD_CAT_HEADS
#define LCHD_CAT_HEADS 1      // k_sweep / k_sweep_duo: the categories of both list heads are read together with their keys
#endif
#ifndef LCHD_BRANCHFREE_HEADS
#define LCHD_BRANCHFREE_HEADS 1
#endif
#ifndef LCHD_SWEEP_WAVES
#define LCHD_SWEEP_WAVES 4
#endif
#ifndef LCHD_BIG_SQRT_COMPUTE
#define LCHD_BIG_SQRT_COMPUTE 1
#endif
#ifndef LCHD_SWEEP_W3MAX
#define LCHD_SWEEP_W3MAX 16   // largest category-slot count that is compiled for 3 waves per SIMD (above: 2)
#endif
#ifndef LCHD_SWEEP_MINW
#define LCHD_SWEEP_MINW 2
#endif
#ifndef LCHD_GEN_W3MAX
#define LCHD_GEN_W3MAX 0   // generic-distance sweeps (MODE_GEN) with at most this many category slots are compiled for 3 waves/SIMD
#endif
#ifndef LCHD_EPL_WGEN
#define LCHD_EPL_WGEN 7  // ... of the sweeps with category weights and of the generic distances, CDF-keyed environments (measured on C2a: weights 2.86 -> 2.54 ms, KS 4.62 -> 4.29 ms; the plain 16-bit Hellinger sweep and the sweeps that evaluate the CDF themselves are faster with 6: their LDS tables + tiles of 448 leave 3 workgroups per CU)
#endif
#ifndef LCHD_EPL_C8S
#define LCHD_EPL_C8S 8   // ... of the 8-bit-count sweep with at most 16 category slots: see LCHD_EPL_C8 (C2a: 343 events per pair on average; tiles of 384: 1.77 ms, 448: 1.61 ms, 512 with whole-list staging: 1.585 ms)
#endif
#ifndef LCHD_EPL_C8
#define LCHD_EPL_C8 8    // ... of the 8-bit-count sweep: tiles of 512 -- two environments of <= 255 points never merge to more, so every pair is ONE tile (a list is staged whole: 256 entries; C5: 448-event tiles + tile-sized staging 3.08 ms, whole-list staging 2.88 ms, 512-event tiles 2.80 ms)
#endif
#ifndef LCHD_C8_WAVES
#define LCHD_C8_WAVES 3  // waves per SIMD the 8-bit-count sweep with more than 16 category slots is compiled for
#endif
#ifndef LCHD_EPL_DENSE
#define LCHD_EPL_DENSE 9   // ... of the sweeps without LDS tables (environments beyond 512 points: dense rows, thousands of events per pair)
#endif
#ifndef LCHD_EPL_BIG
#define LCHD_EPL_BIG 8   // merged events per lane per tile of the many-slot Hellinger-2 sweep (k_sweep<20..32>): tiles of 512
#endif
constexpr int kDuoTileFwd = kDuoTile;  // (lchd_team_tile.h)
// The small rule in force in this pass, or -1 (the plain sweep takes every pair).  With a hint the host launched exactly the
// kernels that have to run (forced); without one every candidate kernel is launched and all of them decide here, from the
// counts of k_pair_meta: the first-choice rule if its pairs are the majority, else the second-choice rule if ITS pairs are --
// the same function of the pair list the host evaluates for the next pass's hint.
__device__ __forceinline__ int rule_in_force(const SweepArgs& args) {
    if (args.forced) return args.small_rule;
    const unsigned long long P = (unsigned long long)args.n_pairs;
    if (2 * args.st->n_small >= P) return args.small_rule;
    if (args.second_rule && 2 * args.st->n_c8 >= P) return args.second_rule;
    return -1;
}
#ifndef LCHD_INLINE_META_PAIRS
#define LCHD_INLINE_META_PAIRS 4096
#endif
constexpr int64_t kInlineMetaPairs = LCHD_INLINE_META_PAIRS;   // calls of at most this many pairs: the sweep works out the pair records itself (one launch)
constexpr int kSqrtTab = 512;  // LDSTAB kernels: environments of at most 512 points, sqrt tables entirely in LDS
constexpr int kSweepWaves = LCHD_SWEEP_WAVES;  // anchor pairs (wavefronts) per workgroup

// A sweep kernel reports a (rare) condition: plain store of 1 into the condition's word of the host-mapped mirror (every
// writer stores the same value; no atomics on host memory, no device-to-host copy afterwards).
__device__ __forceinline__ void sweep_report(HostStatus* h, uint32_t bit) { h->sweep_flags[__builtin_ctz(bit)] = 1u; }

// "Which workgroup finishes last, and what did all of them add up to?" -- without a fence.  An agent-scope release fence on
// this part writes the XCD's whole L2 back (the L2s of the eight XCDs are not coherent with each other), and a kernel that
// has just written 16 MB of pair records pays that per workgroup: 3 900 fences turned an 18 us kernel into a 137 us one.
// Device-scope atomics are performed at the memory side and are coherent by themselves, so everything the workgroups
// hand over travels IN atomics: up to 64 accumulators / counters on separate cache lines (thousands of atomics on one
// word would cost ~11 ns each), a workgroup's counter increment carries a data dependency on the values its accumulator
// atomics RETURNED (so they have been performed), and the workgroup that completes its counter bumps the top-level one.
// Called by ONE thread per workgroup; returns true in exactly one workgroup, which then collects the accumulators with
// atomic exchanges (resetting them).  Everything is left at zero.
constexpr int kDoneStride = 32;  // u32 per slot: 128 bytes apart
static_assert(sizeof(DoneState) == (65 + 64 + 64) * kDoneStride * 4, "DoneState layout (lchd_device.h)");
__device__ __forceinline__ bool last_workgroup_done(DoneState* d, unsigned long long add_sum, uint32_t add_max) {
    const uint32_t n = gridDim.x, G = n < 64u ? n : 64u, g = blockIdx.x % G;
    const uint32_t gs = n / G + (g < n % G ? 1u : 0u);
    const unsigned long long r0 = atomicAdd(&d->acc_sum[g * (kDoneStride / 2)], add_sum);
    const uint32_t r1 = atomicMax(&d->acc_max[g * kDoneStride], add_max);
    uint32_t dep = (uint32_t)r0 | r1;
    asm volatile("v_and_b32 %0, 0, %0" : "+v"(dep));  // 0, but only known once both atomics have returned
    const uint32_t c = atomicAdd(&d->ctr[g * kDoneStride], 1u + dep);
    if (c != gs - 1u) return false;
    uint32_t dep2 = atomicExch(&d->ctr[g * kDoneStride], 0u);  // (= gs: every workgroup of the group is through)
    asm volatile("v_and_b32 %0, 0, %0" : "+v"(dep2));
    if (atomicAdd(&d->ctr[64 * kDoneStride], 1u + dep2) != G - 1u) return false;
    atomicExch(&d->ctr[64 * kDoneStride], 0u);
    return true;
}
// by the threads of the LAST workgroup (slot k handled by thread k < 64): the totals, accumulators reset
__device__ __forceinline__ void collect_done(DoneState* d, int k, unsigned long long& sum, uint32_t& mx) {
    sum = atomicExch(&d->acc_sum[k * (kDoneStride / 2)], 0ull);
    mx = atomicExch(&d->acc_max[k * kDoneStride], 0u);
}

// The end of a pass's record phase, by ONE thread of the last workgroup: what the host wants to know goes into the
// host-mapped mirror (plain stores), the device status is reset for the next pass.
__device__ __forceinline__ void publish_status(const SweepArgs& args, unsigned long long n_small, uint32_t biggest_env) {
    DeviceStatus* st = args.st;
    HostStatus* h = args.hst;
    st->n_small = n_small;  // read by the sweep kernels of this pass when the host did not pick them itself
    const uint32_t over = __hip_atomic_load(&st->max_env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // set by an overflowing environment
    h->flags = __hip_atomic_load(&st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->max_env = over > biggest_env ? over : biggest_env;
    h->n_unique[0] = st->n_unique[0];
    h->n_unique[1] = st->n_unique[1];
    h->n_small = n_small;
    h->n_overflow[0] = st->n_overflow[0];
    h->n_overflow[1] = st->n_overflow[1];
    h->max_bound = __hip_atomic_load(&st->max_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->n_dup_b = __hip_atomic_load(&st->n_dup_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->snapshot_seq = args.seq;
    st->flags = 0u;
    st->max_env = 0u;
    st->n_overflow[0] = 0u;
    st->n_overflow[1] = 0u;
    st->max_bound = 0u;
    st->n_dup_b = 0u;
}


// spread the four 4-bit fields of the low 16 bits of x into four 16-bit fields
__device__ __forceinline__ uint64_t spread4(uint64_t x) {
    // two 32-bit halves, three operations each (and, and / bfe, shift-or); the 64-bit shift-or-mask form is compiled to
    // quarter-rate 32x32 multiplies
    const uint32_t v = (uint32_t)x;
    const uint32_t lo = (v & 0xFu) | ((v & 0xF0u) << 12);
    const uint32_t hi = ((v >> 8) & 0xFu) | ((v & 0xF000u) << 4);
    return ((uint64_t)hi << 32) | lo;
}

// StatisticalDistance::run for Hellinger with a general exponent (statistical_distances.rs:4-10) and Renyi (:31-78) on the
// weighted category counts va / vb with sums sa / sb (pmf.rs:65-83 normalises by the sums).  Inlined (a call from a kernel
// with ~200 live registers costs more in saves and restores than the arithmetic), but with RUNTIME loops over the categories
// on a scratch copy of the counts: one copy of the per-category code, not one per unrolled slot.
//   Hellinger: p^(1/e) = va^(1/e) * sa^(-1/e).  With unit category weights va and sa are integers (< 65536: the count fields
//   are 16 bits), so both factors come from the configuration's tables pow_tab[k] = k^(1/e), pow_tab[65536 + k] = k^(-1/e)
//   (library pow, filled when the configuration is set): one pow per category -- |x - y|^e -- instead of three, none when
//   e is 1, 2, 3 or 4.  Weighted categories take pow_fast for all three.
//   Renyi: ratio^(alpha - 1) = exp((alpha - 1) ln ratio) through the fast log / exp.
__device__ __forceinline__ double sd_generic_fast(int kind, double p0, double p1, const double* va, const double* vb, double sa, double sb, int C,
                                               const double* __restrict__ pow_tab, int tab_half = 65536) {
    const double ia = 1.0 / sa, ib = 1.0 / sb;  // (one reciprocal per side: <= 1 ulp from pmf.rs:78-81's per-category divisions)
    if (kind == SD_HELLINGER) {
        const double e = p0, einv = 1.0 / e;
        const int ie = (e == 1.0 || e == 2.0 || e == 3.0 || e == 4.0) ? (int)e : 0;  // |d|^e by multiplication
        double na1 = 0.0, nb1 = 0.0;
        if (pow_tab) { na1 = pow_tab[tab_half + (int)sa]; nb1 = pow_tab[tab_half + (int)sb]; }
        double dist = 0.0;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            double x, y;
            if (pow_tab) { x = pow_tab[(int)va[c]] * na1; y = pow_tab[(int)vb[c]] * nb1; }
            else { x = pow_fast(va[c] * ia, einv); y = pow_fast(vb[c] * ib, einv); }
            const double d = fabs(x - y);
            dist += ie == 1 ? d : (ie == 2 ? d * d : (ie == 3 ? d * d * d : (ie == 4 ? (d * d) * (d * d) : pow_fast(d, e))));
        }
        return pow_fast(dist / 2.0, einv);
    }
    const double alpha = p0, eps = p1;
    if (alpha == (double)INFINITY) {
        double best = 0.0;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const double r = (va[c] * ia + eps) / (vb[c] * ib + eps);
            best = (c == 0 || r >= best) ? r : best;
        }
        return log_fast(best);
    }
    if (alpha == 0.0) {
        double sm = 0.0;
#pragma unroll 1
        for (int c = 0; c < C; ++c) sm += (va[c] > 0.0) ? vb[c] * ib : 0.0;
        return -log_fast(sm);
    }
    double sm = 0.0;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const double x = va[c] * ia;
        sm += x * pow_fast((x + eps) / (vb[c] * ib + eps), alpha - 1.0);
    }
    return log_fast(sm) / (alpha - 1.0);
}

// One pair's weight function.  hyper_exp with <= 4 terms and uniform keep their parameters in (scalar)
// registers; everything else goes through the out-of-line evaluator with the parameter pointer.
struct WfRegs {
    int kind, np, nterm;
    bool fast;
    double a[4], b[4];
    double inv;  // DevConfig::wf_inv
    const double* p;
};
__device__ __forceinline__ WfRegs wf_load(const WfEntry& e, const double* p, double inv) {
    WfRegs w;
    w.inv = inv;
    w.kind = e.kind;
    w.np = e.n_params;
    w.nterm = e.n_params / 2;
    w.p = p;
    w.fast = (e.kind == WF_UNIFORM) || (e.kind == WF_HYPER_EXP && w.nterm <= 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { w.a[i] = 0.0; w.b[i] = 0.0; }
    if (e.kind == WF_UNIFORM) { w.a[0] = p[0]; w.a[1] = p[1]; }
    else if (w.fast) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < w.nterm) { w.a[i] = p[i]; w.b[i] = p[w.nterm + i]; }
    }
    return w;
}
template <bool WFANY>
__device__ __forceinline__ double cdf_dev(const WfRegs& w, double x) {
    if (w.kind == WF_UNIFORM) {  // cdfs.rs:39-45
        if (x < w.a[0]) return 0.0;
        if (x > w.a[1]) return 1.0;
        return (x - w.a[0]) * w.inv;
    }
    if (w.fast) {  // cdfs.rs:5-21, same accumulation order
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < w.nterm) sum += w.a[i] * exp_nonpos(-w.b[i] * x);
        return 1.0 - sum * w.inv;
    }
    if constexpr (WFANY) return cdf_pow_based(w.kind, w.p, w.np, x);
    else return 0.0;  // unreachable: the host routes tables with other weight functions to the WFANY build
}

// StatisticalDistance::run for the non-default distances; out of line so that the sweep kernel stays small.
static __device__ __noinline__ double sd_generic(int kind, double p0, double p1, const double* p, const double* q, int C) {
    return sd_eval<0>(kind, p0, p1, [&](int c) { return p[c]; }, [&](int c) { return q[c]; }, C);
}

#define LCHD_DUO_TL 16   // lanes per pair of k_sweep_duo's <= 240-event form: four pairs per wavefront (round 1 / 2: 32 lanes, two pairs, 224 events)
#ifndef LCHD_COMPANION_GRID
#define LCHD_COMPANION_GRID 2048u   // (measured: 1024 -> 2048: C2a 19.4 -> 16.6 us, C4 47.3 -> 37.3 us per pass; 4096: no further gain) workgroups of the INDIRECT companion sweep (it walks every pair record and sweeps the few the team kernel left)
#endif

// ---- launchers of the kernel families (one translation unit each) ---------------------------------------------------
// lchd_sweep.hip: one pair per wavefront
void launch_sweep_plain(hipStream_t s, int mode, bool ldstab, int cmax, unsigned grid, int fmode, const SweepArgs& a);  // k_sweep<CMAX, MODE, FMODE, LDSTAB>
void launch_sweep_inline(hipStream_t s, int cmax, unsigned grid, const SweepArgs& a);   // small calls: the records worked out by the sweep itself (INLINE_META)
void launch_sweep_c8(hipStream_t s, int cmax, unsigned grid, const SweepArgs& a);       // the 8-bit-count form (CNT8)
void launch_sweep_indirect(hipStream_t s, int cmax, int tm, unsigned grid, const SweepArgs& a);  // the pairs a small-pair rule leaves over (INDIRECT); tm as launch_team
// lchd_sweep_team.hip: several pairs per wavefront (tile240: four pairs of <= 240 events, else two 8-bit-count pairs of <= 480);
// tm: 0 Hellinger-2 with unit weights, 1 with category weights, 2 Kolmogorov-Smirnov with unit weights
void launch_team(hipStream_t s, int cmax, int tm, bool tile240, unsigned grid, const SweepArgs& a);
// lchd_sweep_wide.hip: 33 .. 65534 categories, environments beyond 65535 points
void launch_sweep_wide(hipStream_t s, int mode, int n_cat, int64_t n_pairs, int fmode, const SweepArgs& a);
// per-device attributes (dynamic LDS above 64 KB) of the families, called by init_device_kernels
void init_prologue_kernels();
void init_env_cells_kernels();
void init_env_rows_kernels();
void init_sweep_wide_kernels();

}  // namespace lchd

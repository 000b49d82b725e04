// lchd_sweep.hip -- K2, one anchor pair per wavefront: k_sweep<CMAX, MODE, FMODE, LDSTAB, INDIRECT, INLINE_META, CNT8>
// (LoCoHD::stat_dist_integral, /root/reference/src/locohd.rs:61-226; PMFSystem, pmf.rs; statistical_distances.rs; cdfs.rs) and
// k_anchors_literal (from_anchors on lists that do not ascend).
#include "lchd_sweep_common.h"

namespace lchd {
// from_anchors on lists whose distances do NOT ascend.  The reference never checks (src/locohd.rs:70-77 only looks at dists[0]) and
// its two-pointer loop then still computes a well-defined number: the heads are compared as they come, the tail of the list that
// is left over is walked in list order, and the first tail interval starts at the LAST element of the finished list (:134-221).
// None of the sort-based kernels can reproduce that, so this one walks the loop itself: one lane, the two weighted count vectors
// (pmf.rs:47-63) in LDS, the statistical distance through the generic evaluator on the normalised vectors (pmf.rs:65-88) at
// every step, F(to) - F(from) per interval (weight_function.rs:118-120).  O((n_A + n_B) C) on one lane: an edge path, not a fast one.
__global__ __launch_bounds__(64) void k_anchors_literal(const DevConfig* __restrict__ cfgp, EnvStore ea, EnvStore eb, int nA, int nB, int wfi,
                                                        double* __restrict__ out) {
    extern __shared__ double lit_s[];  // [4][C]: weighted counts of A, of B, the two normalised vectors
    const int C = cfgp->n_categories;
    double *pa = lit_s, *pb = lit_s + C, *qa = lit_s + 2 * C, *qb = lit_s + 3 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 64) lit_s[c] = 0.0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const DevConfig cfg = *cfgp;
    const WfEntry wf = cfg.wf[wfi];
    const double* prm = cfg.wf_params + wf.offset;
    auto cat_of = [&](const EnvStore& e, int i) -> int { return e.cat16 ? (int)reinterpret_cast<const uint16_t*>(e.cat)[i] : (int)e.cat[i]; };
    auto dist_of = [&](const EnvStore& e, int i) -> double { return u2d(e.key[i]); };
    auto F = [&](double x) -> double { return x == INFINITY ? cfg.wf_finf[wfi] : cdf_eval(wf.kind, prm, wf.n_params, x); };
    auto range = [&](double from, double to) -> double { const double hi = F(to); return hi - F(from); };
    auto H = [&]() -> double {  // pmf.rs:65-88: fresh sums, normalised copies, the configured distance
        double sa = 0.0, sb = 0.0;
        for (int c = 0; c < C; ++c) { sa += pa[c]; sb += pb[c]; }
        for (int c = 0; c < C; ++c) { qa[c] = pa[c] / sa; qb[c] = pb[c] / sb; }
        return sd_generic(cfg.sd_kind, cfg.sd_p0, cfg.sd_p1, qa, qb, C);
    };
    auto add_a = [&](int i) { const int c = cat_of(ea, i); pa[c] += cfg.cat_w[c]; };
    auto add_b = [&](int j) { const int c = cat_of(eb, j); pb[c] += cfg.cat_w[c]; };
    add_a(0);
    add_b(0);
    int i = 0, j = 0;
    double acc = 0.0, prev = 0.0;
    while (i < nA - 1 && j < nB - 1) {
        const double h = H();
        const double a = dist_of(ea, i + 1), b = dist_of(eb, j + 1);
        double nd;
        if (a < b) { ++i; add_a(i); nd = a; }
        else if (a > b) { ++j; add_b(j); nd = b; }
        else { ++i; ++j; add_a(i); add_b(j); nd = a; }  // (equal: NaN distances were refused on the host)
        acc += range(prev, nd) * h;
        prev = nd;
    }
    const double last_a = dist_of(ea, nA - 1), last_b = dist_of(eb, nB - 1);
    if (j < nB - 1) {  // list A is finished
        double h = H();
        ++j;
        acc += range(last_a, dist_of(eb, j)) * h;
        add_b(j);
        while (j < nB - 1) {
            ++j;
            h = H();
            acc += range(dist_of(eb, j - 1), dist_of(eb, j)) * h;
            add_b(j);
        }
        acc += range(last_b, INFINITY) * H();
    } else if (i < nA - 1) {  // list B is finished
        double h = H();
        ++i;
        acc += range(last_b, dist_of(ea, i)) * h;
        add_a(i);
        while (i < nA - 1) {
            ++i;
            h = H();
            acc += range(dist_of(ea, i - 1), dist_of(ea, i)) * h;
            add_a(i);
        }
        acc += range(last_a, INFINITY) * H();
    } else {
        acc += range(last_a, INFINITY) * H();
    }
    *out = acc;
}
void launch_anchors_literal(hipStream_t s, const DevConfig* cfg, int n_categories, const EnvStore& ea, const EnvStore& eb, int nA, int nB, int wfi,
                            double* out) {
    k_anchors_literal<<<1, 64, sizeof(double) * 4 * (size_t)n_categories, s>>>(cfg, ea, eb, nA, nB, wfi, out);
}

// Diagnostic build only (-DLCHD_SWEEP_STAMPS, never the shipped library): per-phase s_memtime deltas summed over all
// wavefronts, read back with lchd_debug_sweep_stamps().
#ifdef LCHD_SWEEP_STAMPS
__device__ unsigned long long g_sweep_stamps[8];
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// register budget: 4 waves/SIMD (<= 128 VGPRs) up to 12 category slots, 3 (<= 168) up to 16, 2 beyond
// INLINE_META (small calls: a few thousand pairs, where launches cost more than arithmetic): the kernel works out every
// pair's record itself instead of reading what k_pair_meta wrote, and its last workgroup publishes the status snapshot --
// ONE launch does the whole sweep phase.
// CNT8 (environments of at most 255 points on both sides, i.e. most pairs at protein-like densities): the packed category
// counts are 8-bit fields, eight per word instead of four -- half the words to scan across the wavefront, to unpack at every
// tile and to keep per lane, and 3 KB less LDS per wavefront, which lets the many-slot variants run at 3 waves per SIMD
// instead of 2.  Pairs with a larger environment are left to the INDIRECT instantiation of the 16-bit kernel.
#ifndef LCHD_C8S_WAVES
#define LCHD_C8S_WAVES 4   // waves per SIMD the 8-bit-count sweep with at most 12 category slots is compiled for
#endif
// Waves per SIMD an instantiation is COMPILED for (the register budget: 512 / waves).  The default configurations (Hellinger-2, unit
// weights, CDF keys, LDS tables) are register-bound: 4 up to 12 slots, 3 up to 16, 2 beyond.  The others are bound by their LDS
// footprint -- 576-event tiles without the LDS tables (deterministic mode, environments beyond 512 points), the weighted sweeps'
// 448-event tiles + per-category multipliers, the CDF-evaluating FMODEs -- and were asking for an occupancy their LDS does not admit
// (round 5: 16 -Wpass-failed warnings, "desired occupancy was 4, final occupancy is 3"): they now ask for what they get, and the
// allocator may use the registers that occupancy leaves.
template <int CMAX, int MODE, int FMODE, bool LDSTAB, bool CNT8>
constexpr int sweep_waves_per_simd() {
    if (MODE == MODE_GEN) return CMAX <= LCHD_GEN_W3MAX ? 3 : 2;
    if (CMAX <= 12) {
        if (CNT8) return LCHD_C8S_WAVES;
        if (!LDSTAB) return ((CMAX > 8 && (FMODE == 2 || MODE == MODE_H2W)) || (MODE == MODE_H2W && FMODE == 2)) ? 2 : 3;
        return MODE == MODE_H2W ? 3 : 4;
    }
    if (CMAX <= LCHD_SWEEP_W3MAX) return (MODE == MODE_H2W && LDSTAB) ? 2 : 3;
    return CNT8 ? LCHD_C8_WAVES : 2;
}
template <int CMAX, int MODE, int FMODE, bool LDSTAB, bool INDIRECT = false, bool INLINE_META = false, bool CNT8 = false>
#ifndef LCHD_DENSE_PARTTAB
#define LCHD_DENSE_PARTTAB 1024   // entries of the partial sqrt table of the sweeps without full LDS tables (0: none)
#endif
#ifndef LCHD_EXACT_H2_LOOP
#define LCHD_EXACT_H2_LOOP 1
#endif
__global__ __launch_bounds__(64 * kSweepWaves, (sweep_waves_per_simd<CMAX, MODE, FMODE, LDSTAB, CNT8>())) void k_sweep(SweepArgs args) {
    static_assert(!(INDIRECT && INLINE_META), "the indirect instantiation reads the records of k_pair_meta");
    static_assert(!CNT8 || (MODE == MODE_H2U && FMODE == F_KEY && LDSTAB && !INDIRECT && !INLINE_META), "8-bit counts: default configuration only");
    // Merged events per lane per tile.  The per-tile prologue (staging, merge path, scan of the packed counts, state reload)
    // costs about as many instructions as the events of a 384-event tile themselves, and it grows with the category slots:
    // the variants with many slots (25 categories at 0.05 atoms/A^3: ~416 events per pair) take tiles of 64 x LCHD_EPL_BIG so
    // that such a pair is ONE tile instead of a full one plus a nearly empty one.
    constexpr bool H2_ = (MODE != MODE_GEN);
    constexpr int EPL = (CNT8 && CMAX > 16) ? LCHD_EPL_C8 : (CNT8 && CMAX <= 16) ? LCHD_EPL_C8S : ((H2_ && LDSTAB && CMAX > 16) ? LCHD_EPL_BIG : ((H2_ && !LDSTAB) ? LCHD_EPL_DENSE : ((MODE != MODE_H2U && FMODE == F_KEY) ? LCHD_EPL_WGEN : kSweepEPL))),
                  TILE = 64 * EPL, WPB = kSweepWaves;
    // entries staged per list and tile: a tile's worth -- but the pairs of the 8-bit-count sweep have at most 254 non-anchor
    // points per environment, so 256 entries hold a whole list (4 KB of keys per wave instead of 7) and a tile of 512 events
    // holds a whole pair
    constexpr int LT = CNT8 ? 256 : TILE, LU = LT / 64;
    static_assert(!CNT8 || (kCount8MaxEnv <= LT && 2 * (kCount8MaxEnv - 1) <= TILE), "a whole list per staging buffer, a whole pair per tile");
    static_assert(EPL <= 15, "4-bit chunk-local counters");
    constexpr int FB = CNT8 ? 8 : 16;     // bits per count field
    constexpr int FPW = 64 / FB;          // count fields per u64 word
    constexpr uint64_t FMASK = CNT8 ? 0xFFull : 0xFFFFull;
    constexpr int NW = (CMAX + FPW - 1) / FPW;  // u64 words of count fields per side
    constexpr int NH = (CMAX + 15) / 16;  // u64 words of 4-bit histogram fields per side
    constexpr bool H2 = (MODE != MODE_GEN);
    constexpr int NV = H2 ? 1 : CMAX;     // only the generic path keeps per-category values in registers
    constexpr int NT = LDSTAB ? (CNT8 ? 256 + 8 : kSqrtTab + 8) : 1;  // sqrt(k), 1/sqrt(k) for k <= 512 (255) in LDS; otherwise read from the global tables
    __shared__ double t_sqrt[NT], t_rsqrt[NT];
    constexpr bool PARTTAB = !LDSTAB && H2_ && (LCHD_DENSE_PARTTAB != 0);
    constexpr int kPartTab = LCHD_DENSE_PARTTAB > 0 ? LCHD_DENSE_PARTTAB : 1;
    __shared__ double t_part[PARTTAB ? kPartTab : 1];
    __shared__ double w_s[32], sw_s[32];
    // MODE_GEN, Hellinger with a general exponent, environments of at most kSqrtTab points: k^(1/e) and k^(-1/e) for k <= 512 in
    // LDS (the two look-ups per category and event went to the 1 MB tables in global memory: latency-bound at 2 waves per SIMD)
    constexpr int kGenTab = kSqrtTab + 8;
    __shared__ double t_pow[(MODE == MODE_GEN) ? 2 * kGenTab : 1];
    __shared__ uint64_t sA_[WPB][LT], sB_[WPB][LT];
    __shared__ uint8_t cA_[WPB][LT], cB_[WPB][LT];
    // per-lane category counts of the event loop: [side][word][lane] u64 of four 16-bit fields (a lane only ever touches its own)
    // (13 and more category slots only: up to 12 the register form runs at 4 waves/SIMD, which the extra 3 KB of LDS per wave
    // would cut to 3 -- measured 2-6 % slower -- while from 13 on the LDS form is 4-13 % faster at unchanged occupancy)
#ifndef LCHD_C8_REGCNT
#define LCHD_C8_REGCNT 0
#endif
#ifndef LCHD_C8_LDSCNT_ALL
#define LCHD_C8_LDSCNT_ALL 1   // the 8-bit-count sweeps keep their per-lane counts in LDS for every slot count (<= 12 slots: their 2 KB per wave do not cost a wave of occupancy, and the byte read-modify-write replaces the word select + 4-bit counter chains: C2a sweep 1.577 -> 1.523 ms)
#endif
    constexpr bool LDSCNT = H2 && LDSTAB && (NW > 3 || (CNT8 && LCHD_C8_LDSCNT_ALL)) && (LCHD_LDS_COUNTS != 0) && !(CNT8 && LCHD_C8_REGCNT);  // (16-bit fields: from 13 category slots on)
    __shared__ uint64_t lc_[LDSCNT ? WPB : 1][LDSCNT ? 2 * NW * 64 : 1];
    // When pairs with at most kDuoTile merged events are the majority of a launch, k_sweep_duo sweeps them two per wavefront
    // and the INDIRECT instantiation of this kernel picks the remaining ones out of the pair records; otherwise the plain
    // instantiation sweeps everything.  All three decide from the same word (k_pair_meta: DeviceStatus::n_small).
    const int small_rule = (INDIRECT || CNT8 || !args.forced) ? rule_in_force(args) : -1;
    if (!args.forced) {  // (forced: the host launched exactly the kernels that have to run)
        if constexpr (CNT8) { if (small_rule != 1) return; }          // (the small-pair kernels and their companion: only when
        else if constexpr (INDIRECT) { if (small_rule < 0) return; }  //  the pairs of their rule are the majority)
        else { if (args.duo_enabled && small_rule >= 0) return; }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform => everything derived from it stays scalar
    const DevConfig* __restrict__ cfgp = args.cfg;
    const int C = cfgp->n_categories;
    const double* __restrict__ g_sqrt = args.sqrt_tab;    // [65536] sqrt(k)
    const double* __restrict__ g_rsqrt = args.rsqrt_tab;  // [65536] 1/sqrt(k)
    if constexpr (LDSTAB)
        for (int k = tid; k < NT; k += 64 * WPB) {
            t_sqrt[k] = g_sqrt[k];
            t_rsqrt[k] = g_rsqrt[k];
        }
    if constexpr (PARTTAB)
        for (int k = tid; k < kPartTab; k += 64 * WPB) t_part[k] = g_sqrt[k];
    if (tid < 32) {
        const double wv_ = tid < C ? cfgp->cat_w[tid] : 0.0;
        w_s[tid] = wv_;
        sw_s[tid] = sqrt(wv_);
    }
    bool gen_lds = false;
    if constexpr (MODE == MODE_GEN) {
        gen_lds = args.gen_tab && cfgp->pow_tab && args.env_a.stride <= kSqrtTab && args.env_b.stride <= kSqrtTab;  // (wave-uniform)
        if (gen_lds)
            for (int k = tid; k < kGenTab; k += 64 * WPB) {
                t_pow[k] = cfgp->pow_tab[k];
                t_pow[kGenTab + k] = cfgp->pow_tab[65536 + k];
            }
    }
    __syncthreads();
    uint64_t* sA = sA_[wv];
    uint64_t* sB = sB_[wv];
    uint8_t* cA = cA_[wv];
    uint8_t* cB = cB_[wv];
    unsigned char* lcl = reinterpret_cast<unsigned char*>(lc_[LDSCNT ? wv : 0]) + lane * 8;  // this lane's slot of word 0, side A
    constexpr int kLcSide = NW * 512;  // bytes from a side-A field to the same field of side B

#if LCHD_BIG_SQRT_COMPUTE
    // environments beyond the LDS tables: sqrt(count) is computed (rsq seed + Goldschmidt, <= 1 ulp from the table value)
    // instead of being fetched from the 65536-entry global tables -- four dependent L2 round trips per event otherwise
    // (dense rows: counts below kPartTab -- per-category counts of a 10^4-point row with ten categories stay there until the row's
    //  last tiles -- come from a partial LDS table, larger ones are computed; a per-lane branch, both arms only near a row's end)
    auto sqrt_cnt = [&](int cnt) -> double {
        if constexpr (LDSTAB) return t_sqrt[cnt];
        else if constexpr (PARTTAB) { if (cnt < kPartTab) return t_part[cnt]; else return sqrt_unit((double)cnt); }
        else return sqrt_unit((double)cnt);
    };
    auto rsqrt_cnt = [&](int cnt) -> double {
        if constexpr (LDSTAB) return t_rsqrt[cnt];
        else {
            const double x = (double)cnt;
            double y = __builtin_amdgcn_rsq(x);
            y = y * fma(-0.5 * x, y * y, 1.5);
            y = y * fma(-0.5 * x, y * y, 1.5);
            return y;
        }
    };
#else
    auto sqrt_cnt = [&](int cnt) -> double { if constexpr (LDSTAB) return t_sqrt[cnt]; else return g_sqrt[cnt]; };
    auto rsqrt_cnt = [&](int cnt) -> double { if constexpr (LDSTAB) return t_rsqrt[cnt]; else return g_rsqrt[cnt]; };
#endif
    // sqrt of the weighted count of category c (c may be dynamic)
    auto root_of = [&](int c, int cnt) -> double {
        if constexpr (MODE == MODE_H2W) return sqrt_cnt(cnt) * sw_s[c & 31];
        else return sqrt_cnt(cnt);
    };

#ifdef LCHD_SWEEP_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
    // One 16-byte record per pair (k_pair_meta) replaces the dependent chain anchors -> slot -> len -> first category; the
    // record of the wave's NEXT pair is requested before the current pair is processed.
    // configuration words the loop needs: read once (the compiler must assume the status atomics may alias *cfgp)
    const int n_wf = cfgp->n_wf;
    const double* __restrict__ finf_tab = cfgp->wf_finf;
    const double Finf0 = finf_tab[0];
    const int64_t pstride = (int64_t)gridDim.x * WPB;
    const int64_t total = args.n_pairs;
    int64_t q = (int64_t)blockIdx.x * WPB + wv;
    int biggest_env = 0;  // INLINE_META: largest environment this wave has met
    auto record_of = [&](int64_t pp) -> int4 {  // pp wave-uniform
        if constexpr (INLINE_META) {  // the arithmetic of k_pair_meta
            int64_t ea = pp, eb = pp;
            bool ok = true;
            if (args.anchors) {
                const int64_t ia_ = args.anchors[2 * pp], ib_ = args.anchors[2 * pp + 1];
                ok = !(ia_ < 0 || ib_ < 0 || ia_ >= args.n_slot_a || ib_ >= args.n_slot_b);
                if (ok) { ea = args.slot_a[ia_]; eb = args.slot_b ? args.slot_b[ib_] : pp; }
            }
            int nA_ = 0, nB_ = 0, c0a_ = 0, c0b_ = 0;
            if (ok) {
                nA_ = args.env_a.len[ea];
                nB_ = args.env_b.len[eb];
                if (nA_ > 0 && nB_ > 0) {
                    c0a_ = args.env_a.cat[ea * args.env_a.stride];
                    c0b_ = args.env_b.cat[eb * args.env_b.stride];
                } else {
                    nA_ = nB_ = 0;
                }
            }
            return make_int4((int)ea, (int)eb, nA_ | (c0a_ << 24), nB_ | (c0b_ << 24));
        } else {
            return args.meta[pp];
        }
    };
    // the record lives in four scalar registers; the next one is moved there as soon as its (early) load has returned, so the
    // loop's back edge never waits on vector memory (in particular not on the score store of the pair just finished)
    int mx, my, mz, mw;
    {
        const int4 m0 = record_of(q < total ? q : 0);
        mx = __builtin_amdgcn_readfirstlane(m0.x); my = __builtin_amdgcn_readfirstlane(m0.y);
        mz = __builtin_amdgcn_readfirstlane(m0.z); mw = __builtin_amdgcn_readfirstlane(m0.w);
    }
    int nx = mx, ny = my, nz = mz, nw = mw;
    // INDIRECT: the wave walks blocks of 64 consecutive pairs, every lane holding one record; the pairs that are too large for
    // k_sweep_duo are picked out of a block with a ballot and swept one after the other (the plain instantiation folds all of
    // this away and keeps its one-record-ahead loop)
    int64_t blk = (int64_t)blockIdx.x * WPB + wv, p_cur = 0;
    unsigned long long todo = 0;
    int4 mm = make_int4(0, 0, 0, 0);
    bool ok = true;
    // (the leftover list of a forced pass, SweepArgs::left_listing: one listed pair per step instead of the scan below)
    const uint32_t n_left = (INDIRECT && args.left_listing) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*args.left_count) : 0u;
    auto advance = [&]() -> bool {
        if (args.left_listing) {  // (uniform)
            if (blk >= (int64_t)n_left) return false;
            p_cur = (int64_t)args.left_list[blk];
            blk += pstride;
            const int4 m1 = args.meta[p_cur];
            mx = __builtin_amdgcn_readfirstlane(m1.x); my = __builtin_amdgcn_readfirstlane(m1.y);
            mz = __builtin_amdgcn_readfirstlane(m1.z); mw = __builtin_amdgcn_readfirstlane(m1.w);
            return true;
        }
        while (todo == 0) {
            if (blk * 64 >= total) return false;
            const int64_t pp = blk * 64 + lane;
            mm = pp < total ? args.meta[pp] : make_int4(0, 0, 0, 0);
            // the pairs the small-pair kernel of this launch leaves over: more than kDuoTile merged events (k_sweep_duo), or an
            // environment of more than 255 points (the 8-bit-count k_sweep)
            const int za = mm.z & 0xFFFFFF, zb = mm.w & 0xFFFFFF;
            todo = __ballot(za > 0 && !pair_is_small(small_rule, za, zb));
            p_cur = blk * 64;
            blk += pstride;
        }
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        p_cur = (p_cur & ~(int64_t)63) + b;
        mx = __builtin_amdgcn_readlane(mm.x, b); my = __builtin_amdgcn_readlane(mm.y, b);
        mz = __builtin_amdgcn_readlane(mm.z, b); mw = __builtin_amdgcn_readlane(mm.w, b);
        return true;
    };
    if constexpr (INDIRECT) ok = advance();
    for (; INDIRECT ? ok : (q < total);
         INDIRECT ? (void)(ok = advance()) : (void)(q += pstride, mx = nx, my = ny, mz = nz, mw = nw)) {
        const int64_t p = INDIRECT ? p_cur : q;
        const int4 mn = INDIRECT ? make_int4(0, 0, 0, 0) : record_of(q + pstride < total ? q + pstride : q);
        auto take_next = [&]() {
            if constexpr (!INDIRECT) {
                nx = __builtin_amdgcn_readfirstlane(mn.x); ny = __builtin_amdgcn_readfirstlane(mn.y);
                nz = __builtin_amdgcn_readfirstlane(mn.z); nw = __builtin_amdgcn_readfirstlane(mn.w);
            }
        };
        const int nA = mz & 0xFFFFFF, nB = mw & 0xFFFFFF;
        if constexpr (INLINE_META) biggest_env = max(biggest_env, max(nA, nB));
        if (nA <= 0 || nB <= 0) {  // anchor out of range (flagged by k_mark_anchors) or overflow / empty environment (flagged by K1)
            if (lane == 0) args.out[p] = nan("");
            take_next();
            continue;
        }
        if constexpr (CNT8) {
            if (max(nA, nB) > kCount8MaxEnv) {  // a count could leave its 8-bit field: the indirect 16-bit kernel takes this pair
                take_next();
                continue;
            }
        }
        const int64_t ea = mx, eb = my;
        const int c0a = (mz >> 24) & 255, c0b = (mw >> 24) & 255;  // categories of the two anchors
        // (a dictionary's key sets, EnvStore::cdf_keys > 1: the set of this pair's weight function)
        const int kset = (FMODE == F_KEY && args.wf_index) ? args.wf_index[p] : 0;
        const int64_t kset_ok = (kset >= 0 && kset < n_wf) ? kset : 0;
        const uint64_t* __restrict__ kA = args.env_a.key + ea * args.env_a.stride + kset_ok * args.env_a.set_stride;
        const uint64_t* __restrict__ kB = args.env_b.key + eb * args.env_b.stride + kset_ok * args.env_b.set_stride;
        const uint8_t* __restrict__ tA = args.env_a.cat + ea * args.env_a.stride;
        const uint8_t* __restrict__ tB = args.env_b.cat + eb * args.env_b.stride;
        const int wfi = args.wf_index ? args.wf_index[p] : 0;
        if (args.wf_index && (wfi < 0 || wfi >= n_wf)) {
            if (lane == 0) { sweep_report(args.hst, ST_BAD_WF); args.out[p] = nan(""); }
            take_next();
            continue;
        }
        constexpr bool WFANY = (FMODE == F_ANY);
        WfRegs wf{};
        if constexpr (FMODE != F_KEY) {
            const WfEntry wfe = cfgp->wf[wfi];
            wf = wf_load(wfe, cfgp->wf_params + wfe.offset, cfgp->wf_inv[wfi]);
            if (kA[0] != 0ull || kB[0] != 0ull) {  // src/locohd.rs:74-77 (F_KEY: checked by the environment kernels)
                if (lane == 0) { sweep_report(args.hst, ST_FIRST_NOT_ZERO); args.out[p] = nan(""); }
                take_next();
                continue;
            }
        }
        auto cdf_of_key = [&](uint64_t k) -> double {
            if constexpr (FMODE == F_KEY) return u2d(k);
            else return cdf_dev<WFANY>(wf, u2d(k));
        };

        bool zero_norm = false;
        // wave-uniform packed integer category counts (16-bit fields), seeded with the two anchors (:82-84)
        uint64_t cntA[NW], cntB[NW];
        {
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                cntA[k] = ((c0a / FPW) == k) ? (1ull << ((c0a % FPW) * FB)) : 0ull;
                cntB[k] = ((c0b / FPW) == k) ? (1ull << ((c0b % FPW) * FB)) : 0ull;
            }
        }

        // ---- per-lane state -------------------------------------------------------------------------
        uint64_t exA[NW], exB[NW];   // packed category counts at the start of this lane's chunk
        uint64_t dA[NH], dB[NH];     // what the chunk has added so far, 4 bits per category
#pragma unroll
        for (int k = 0; k < NH; ++k) dA[k] = dB[k] = 0;
        int totA = 1, totB = 1;      // points seen per side (incl. anchor)
        double ra = 0.0, rb = 0.0;   // H2: 1/sqrt(total weight)
        double na = 0.0, nb = 0.0;   // H2W: total weights
        double D = 0.0;              // H2: sum_c sqrt(a_c * b_c)  (Bhattacharyya numerator)
        double va[NV], vb[NV];       // GEN: weighted category counts (pmf.rs:16-17)

        auto field = [&](const uint64_t (&ex)[NW], int c) -> int {  // static c
            return (int)((ex[c / FPW] >> ((c % FPW) * FB)) & FMASK);
        };
        auto load_state = [&]() {  // registers <- packed counts exA/exB and totals totA/totB
            if constexpr (H2) {
                D = 0.0;
                if constexpr (MODE == MODE_H2W) na = nb = 0.0;
#pragma unroll
                for (int k = 0; k < NW; ++k) {  // padded categories have count 0 on both sides: contribute 0
#pragma unroll
                    for (int f = 0; f < FPW; ++f) {
                        const int c = FPW * k + f;
                        if (c >= CMAX) continue;
                        const int ca = field(exA, c), cb = field(exB, c);
                        if constexpr (MODE == MODE_H2W) {
                            D += w_s[c] * (sqrt_cnt(ca) * sqrt_cnt(cb));
                            na += w_s[c] * (double)ca;
                            nb += w_s[c] * (double)cb;
                        } else {
                            D += sqrt_cnt(ca) * sqrt_cnt(cb);
                        }
                    }
                    // <= 16 slots run at 3-4 waves/SIMD on a tight register budget: one word's table look-ups in flight at a time;
                    // the larger variants (2 waves/SIMD, 256 registers) profit from every second word's being in flight together
                    if constexpr (CMAX <= 16 || CNT8) __builtin_amdgcn_sched_barrier(0);
                    else if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (MODE == MODE_H2W) { ra = 1.0 / sqrt(na); rb = 1.0 / sqrt(nb); }
                else { ra = rsqrt_cnt(totA); rb = rsqrt_cnt(totB); }
            } else {
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    va[c] = w_s[c] * (double)field(exA, c);
                    vb[c] = w_s[c] * (double)field(exB, c);
                }
            }
        };
        // exact squared Hellinger distance in the literal difference-of-roots form (statistical_distances.rs:4-10)
        auto exact_h2 = [&]() -> double {
            double acc2 = 0.0;
            if constexpr (LDSCNT && CMAX > 16 && (LCHD_EXACT_H2_LOOP != 0)) {
                // many slots, counts in LDS: a runtime loop over the count words (one copy of the eight-field body): the rarely
                // taken path no longer sizes the kernel's register allocation
#pragma unroll 1
                for (int k = 0; k < NW; ++k) {
                    const uint64_t wa = *reinterpret_cast<const uint64_t*>(lcl + k * 512), wb = *reinterpret_cast<const uint64_t*>(lcl + kLcSide + k * 512);
#pragma unroll
                    for (int f = 0; f < FPW; ++f) {
                        const int ca = (int)((wa >> (f * FB)) & FMASK), cb = (int)((wb >> (f * FB)) & FMASK);
                        const double d = root_of(FPW * k + f, ca) * ra - root_of(FPW * k + f, cb) * rb;  // (padded slots: 0 - 0)
                        acc2 = fma(d, d, acc2);
                    }
                }
                return 0.5 * acc2;
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                // the chunk-start words go through an empty volatile asm: they do not change during the event loop, and the
                // optimiser otherwise hoists all 2 * CMAX table addresses of this rarely taken path out of the loop, where
                // they occupy registers the common path has to spill for
                uint64_t ea = exA[k], eb = exB[k];
                if constexpr (!LDSCNT) asm volatile("" : "+v"(ea), "+v"(eb));
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
                    const int c = FPW * k + f;
                    if (c >= CMAX) continue;
                    int ca, cb;
                    if constexpr (LDSCNT) {
                        if constexpr (CNT8) {
                            ca = *reinterpret_cast<const uint8_t*>(lcl + k * 512 + f);
                            cb = *reinterpret_cast<const uint8_t*>(lcl + kLcSide + k * 512 + f);
                        } else {
                            ca = *reinterpret_cast<const uint16_t*>(lcl + k * 512 + f * 2);
                            cb = *reinterpret_cast<const uint16_t*>(lcl + kLcSide + k * 512 + f * 2);
                        }
                    } else {
                        ca = (int)((ea >> (f * FB)) & FMASK) + (int)((dA[c >> 4] >> ((c & 15) * 4)) & 15ull);
                        cb = (int)((eb >> (f * FB)) & FMASK) + (int)((dB[c >> 4] >> ((c & 15) * 4)) & 15ull);
                    }
                    const double d = root_of(c, ca) * ra - root_of(c, cb) * rb;  // equal inputs cancel exactly
                    acc2 = fma(d, d, acc2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return 0.5 * acc2;
        };
        auto distance = [&]() -> double {  // pmf.rs:85-88
            if constexpr (H2) {
                // H^2 = 1 - sum_c sqrt(p_c q_c): O(1) per event from the running D.  Its rounding error (~1e-16
                // absolute) only matters when H^2 itself is tiny, so small values are recomputed in the exact form
                // (which also returns exactly 0 for identical environments).
                double h2 = 1.0 - (ra * rb) * D;
                if (h2 < kExactH2Below) h2 = exact_h2();
                return sqrt_unit(h2);
            } else {
                double sa_ = 0.0, sb_ = 0.0;  // pmf.rs:67-68: fresh sums
#pragma unroll
                for (int c = 0; c < CMAX; ++c) sa_ += va[c];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) sb_ += vb[c];
                if (sa_ == 0.0 || sb_ == 0.0) zero_norm = true;
                const double ia_ = 1.0 / sa_, ib_ = 1.0 / sb_;  // one reciprocal per side (<= 1 ulp from pmf.rs:78-81's per-category divisions)
                const int kind = cfgp->sd_kind;
                if (kind == SD_KS) {  // statistical_distances.rs:12-21, straight from the registers
                    double best = 0.0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) best = fmax(best, fabs(va[c] * ia_ - vb[c] * ib_));  // padded slots give |0 - 0|
                    return best;
                }
                const double prm0 = cfgp->sd_p0, prm1 = cfgp->sd_p1;
                if (kind == SD_KL || (kind == SD_RENYI && prm0 == 1.0)) {  // :23-29 (Renyi with alpha = 1: :36-38)
                    const double eps = kind == SD_KL ? prm0 : prm1;
                    double dist = 0.0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if (c < C) {
                            const double x = va[c] * ia_;
                            dist += x * log_fast((x + eps) / (vb[c] * ib_ + eps));
                        }
                    }
                    return dist;
                }
                if constexpr (MODE == MODE_GEN) {
                    // Hellinger with exponent 1, 2, 3 or 4, unit weights, environments inside the LDS power tables: k^(1/e) from
                    // the tables, |x - y|^e by multiplication -- no transcendental per category, so the per-category code is a dozen
                    // instructions and can be unrolled over the slots straight from the registers (the runtime-loop form below
                    // goes through a scratch copy of the counts)
                    if (gen_lds && kind == SD_HELLINGER && (prm0 == 1.0 || prm0 == 2.0 || prm0 == 3.0 || prm0 == 4.0)) {
                        const int ie = (int)prm0;
                        const double na1 = t_pow[kGenTab + (int)sa_], nb1 = t_pow[kGenTab + (int)sb_];
                        double dist = 0.0;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) {
                            const double d = fabs(t_pow[(int)va[c]] * na1 - t_pow[(int)vb[c]] * nb1);  // (padded slots: |0 - 0|)
                            dist += ie == 1 ? d : (ie == 2 ? d * d : (ie == 3 ? d * d * d : (d * d) * (d * d)));
                        }
                        return pow_fast(dist / 2.0, 1.0 / prm0);
                    }
                }
                // Hellinger with a general exponent, Renyi: runtime loops over a scratch copy of the weighted counts (unrolled per
                // category slot these branches tripled the kernel's size; as an out-of-line call the register saves cost more
                // than the arithmetic)
                double ca_[CMAX], cb_[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) { ca_[c] = va[c]; cb_[c] = vb[c]; }
                if constexpr (MODE == MODE_GEN) {
                    if (gen_lds) return sd_generic_fast(kind, prm0, prm1, ca_, cb_, sa_, sb_, C, t_pow, kGenTab);
                }
                return sd_generic_fast(kind, prm0, prm1, ca_, cb_, sa_, sb_, C, args.gen_tab ? cfgp->pow_tab : nullptr);
            }
        };

#pragma unroll
        for (int k = 0; k < NW; ++k) { exA[k] = cntA[k]; exB[k] = cntB[k]; }
        double F_carry = cdf_of_key(kA[0]);  // F(0): both anchors sit at distance 0
        double H_carry;
        if constexpr (H2) {
            // only the two anchors: both PMFs are point masses => H = 0 if they share the category, else 1 (exactly)
            H_carry = (c0a == c0b) ? 0.0 : 1.0;
        } else {
            load_state();
            H_carry = distance();
        }
        double acc = 0.0;

        const int mA = nA - 1, mB = nB - 1, M = mA + mB;  // non-anchor events
        int ia = 0, ib = 0;
        for (int k0 = 0; k0 < M; k0 += TILE) {
            const int T = min(TILE, M - k0);
            const int nAt = min(LT, mA - ia), nBt = min(LT, mB - ib);
            STAMP(0);
            wave_sync_lds();  // previous tile fully consumed
            // stage the tile: the global loads of BOTH lists are issued before the first LDS write (one memory latency per tile)
            {
                // (wave-uniform base + 32-bit lane offset + immediate: one address register pair serves all loads of a list)
                uint64_t rkA[LU], rkB[LU];
                uint8_t rcA[LU], rcB[LU];
                const char* pkA = reinterpret_cast<const char*>(kA + (1 + ia));
                const char* pkB = reinterpret_cast<const char*>(kB + (1 + ib));
                const uint8_t* pcA = tA + (1 + ia);
                const uint8_t* pcB = tB + (1 + ib);
                const uint32_t lane8 = (uint32_t)lane * 8u, lane1 = (uint32_t)lane;
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const bool in = lane + 64 * u < nAt;
                    rkA[u] = in ? *reinterpret_cast<const uint64_t*>(pkA + lane8 + 512u * u) : 0ull;
                    rcA[u] = in ? pcA[lane1 + 64u * u] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const bool in = lane + 64 * u < nBt;
                    rkB[u] = in ? *reinterpret_cast<const uint64_t*>(pkB + lane8 + 512u * u) : 0ull;
                    rcB[u] = in ? pcB[lane1 + 64u * u] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int t = lane + 64 * u;
                    if (t < nAt) { sA[t] = rkA[u]; cA[t] = rcA[u]; }
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int t = lane + 64 * u;
                    if (t < nBt) { sB[t] = rkB[u]; cB[t] = rcB[u]; }
                }
            }
            wave_sync_lds();
            STAMP(1);
            // lane l owns merged events [d0, d1); each lane searches the END of its chunk
            const int epl = (T + 63) >> 6;  // <= EPL (<= 15: the 4-bit histogram fields)
            const int d0 = min(lane * epl, T), d1 = min(d0 + epl, T);
            const int i1 = merge_path(sA, nAt, sB, nBt, d1);
            int i0 = __shfl_up(i1, 1);
            if (lane == 0) i0 = 0;
            const int iend = __builtin_amdgcn_readlane(i1, 63);
            const int j0 = d0 - i0, j1 = d1 - i1;
            STAMP(2);

            // pass 1: 4-bit-per-category histogram of this lane's chunk (at most 8 points per side)
            uint64_t hA[NH], hB[NH];
#pragma unroll
            for (int k = 0; k < NH; ++k) hA[k] = hB[k] = 0;
#if LCHD_PASS1_FUSED
            if constexpr (NH == 1) {
                // one fixed-trip loop over the chunk's (at most EPL) points, A's run first, then B's: the two data-dependent
                // loops it replaces each ran for the longest run of any lane.  hT counts every point, hA only A's.
                const int nAl = i1 - i0, nl = d1 - d0;
                const uint8_t* pa_ = cA + i0;
                const uint8_t* pb_ = cB + (j0 - nAl);
                uint64_t hT = 0;
#pragma unroll
                for (int m = 0; m < EPL; ++m) {
                    if (m < epl) {  // wave-uniform
                        const bool isA = m < nAl;
                        const int ct = (isA ? pa_ : pb_)[m < nl ? m : 0];
                        const uint64_t inc = (m < nl) ? (1ull << ((ct & 15) * 4)) : 0ull;
                        hT += inc;
                        hA[0] += isA ? inc : 0ull;
                    }
                }
                hB[0] = hT - hA[0];
            } else
#endif
            {
            for (int i = i0; i < i1; ++i) {
                const int ct = cA[i];
#pragma unroll
                for (int k = 0; k < NH; ++k) hA[k] += ((ct >> 4) == k) ? (1ull << ((ct & 15) * 4)) : 0ull;
            }
            for (int j = j0; j < j1; ++j) {
                const int ct = cB[j];
#pragma unroll
                for (int k = 0; k < NH; ++k) hB[k] += ((ct >> 4) == k) ? (1ull << ((ct & 15) * 4)) : 0ull;
            }
            }
            STAMP(3);
            // widen to 16-bit fields and exclusive-scan across the wavefront
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint64_t va_ = CNT8 ? spread8(hA[(k * 8) / 16] >> (((k * 8) % 16) * 4)) : spread4(hA[(k * 4) / 16] >> (((k * 4) % 16) * 4));
                const uint64_t vb_ = CNT8 ? spread8(hB[(k * 8) / 16] >> (((k * 8) % 16) * 4)) : spread4(hB[(k * 4) / 16] >> (((k * 4) % 16) * 4));
                const uint64_t sa_ = wave_incl_scan_fields(va_), sb_ = wave_incl_scan_fields(vb_);
                exA[k] = cntA[k] + sa_ - va_;
                exB[k] = cntB[k] + sb_ - vb_;
                cntA[k] += readlane_u64(sa_, 63);  // carry for the next tile (scalar)
                cntB[k] += readlane_u64(sb_, 63);
            }
            totA = 1 + ia + i0;
            totB = 1 + ib + j0;
            STAMP(4);
            load_state();
            if constexpr (LDSCNT) {
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    *reinterpret_cast<uint64_t*>(lcl + k * 512) = exA[k];
                    *reinterpret_cast<uint64_t*>(lcl + kLcSide + k * 512) = exB[k];
                }
            }
            STAMP(5);

            // pass 2: sequential sweep of this lane's events.  Branch-free: both list heads stay in registers and the one
            // that was consumed is refilled with a single (address-selected) LDS read.  The packed counts exA/exB stay
            // fixed at their chunk-start values; what the chunk itself adds (<= 6 per category) is kept in 4-bit fields.
            int i = i0, j = j0;
#if LCHD_HEADS_REREAD
            uint64_t ka = sA[i], kb = sB[j];  // both heads are re-read after every event; run ends are tested on the indices
#if LCHD_CAT_HEADS
            int cta = cA[i], ctb = cB[j];     // ... and so are their categories: the event's category is a select, not an LDS round trip behind takeA
#endif
#else
            uint64_t ka = (i < i1) ? sA[i] : kPadKey, kb = (j < j1) ? sB[j] : kPadKey;
#endif
#pragma unroll
            for (int k = 0; k < NH; ++k) dA[k] = dB[k] = 0;
            double Fp = 0.0, Hp = 0.0, firstF = 0.0, local = 0.0;
            for (int e = 0; e < epl; ++e) {
                if (d0 + e < d1) {
#if LCHD_HEADS_REREAD
                    // A-first on ties; an exhausted run cannot be taken.  Two LDS reads per event instead of one, but none of
                    // the selects that steer a single refill into the right head register (the kernel is VALU-issue bound).
                    const bool takeA = (i < i1) & ((j >= j1) | (ka <= kb));
                    const uint64_t key = takeA ? ka : kb;
#if LCHD_CAT_HEADS
                    const int ct = takeA ? cta : ctb;
#else
                    const int ct = (takeA ? cA : cB)[takeA ? i : j];
#endif
                    i += takeA ? 1 : 0;
                    j += takeA ? 0 : 1;
                    ka = sA[i];  // (one past the run's end at most: inside the tile buffers, never used)
                    kb = sB[j];
#if LCHD_CAT_HEADS
                    cta = cA[i];
                    ctb = cB[j];
#endif
#else
                    const bool takeA = (ka <= kb);  // an exhausted list shows the pad key (> every real key)
                    const uint64_t key = takeA ? ka : kb;
#if LCHD_BRANCHFREE_HEADS
                    const int ct = (takeA ? cA : cB)[takeA ? i : j];
                    i += takeA ? 1 : 0;
                    j += takeA ? 0 : 1;
                    {
                        const int nidx = takeA ? i : j, nend = takeA ? i1 : j1;
                        const uint64_t nk = (takeA ? sA : sB)[min(nidx, LT - 1)];
                        const uint64_t nh = nidx < nend ? nk : kPadKey;
                        ka = takeA ? nh : ka;
                        kb = takeA ? kb : nh;
                    }
#else
                    const int ct = takeA ? cA[i] : cB[j];
                    if (takeA) { ++i; ka = (i < i1) ? sA[i] : kPadKey; } else { ++j; kb = (j < j1) ? sB[j] : kPadKey; }
#endif
#endif
                    const double F = cdf_of_key(key);
                    if (e == 0) firstF = F; else local += (F - Fp) * Hp;
                    totA += takeA ? 1 : 0;
                    totB += takeA ? 0 : 1;
                    if constexpr (H2) {
                        // pmf.rs:47-63: one more point of category ct on one side
                        int cntA_, cntB_;
                        if constexpr (LDSCNT) {
                            // counts of category ct on both sides: two 16-bit LDS reads at one address (+ an immediate for side
                            // B); the side that took the event writes its count back incremented.  LDS serves a wave's requests
                            // in order, so the next event of this lane sees the update.
                            if constexpr (CNT8) {
                                unsigned char* pf = lcl + ((ct >> 3) << 9) + (ct & 7);
                                cntA_ = *pf;
                                cntB_ = *(pf + kLcSide);
                                *(pf + (takeA ? 0 : kLcSide)) = (unsigned char)((takeA ? cntA_ : cntB_) + 1);
                            } else {
                            unsigned char* pf = lcl + ((ct >> 2) << 9) + ((ct & 3) << 1);
                            cntA_ = *reinterpret_cast<const uint16_t*>(pf);
                            cntB_ = *reinterpret_cast<const uint16_t*>(pf + kLcSide);
                            *reinterpret_cast<uint16_t*>(pf + (takeA ? 0 : kLcSide)) = (uint16_t)((takeA ? cntA_ : cntB_) + 1);
                            }
                        } else {
                        const int sh = (ct % FPW) * FB, sh4 = (ct & 15) * 4;
                        uint64_t wA = exA[0], wB = exB[0];  // (every category is inside the map: checked at the environment build)
#pragma unroll
                        for (int k = 1; k < NW; ++k) {
                            const bool hit = ((ct / FPW) == k);
                            wA = hit ? exA[k] : wA;
                            wB = hit ? exB[k] : wB;
                        }
                        uint64_t qA = dA[0], qB = dB[0];
                        if constexpr (NH == 2) { qA = (ct & 16) ? dA[1] : qA; qB = (ct & 16) ? dB[1] : qB; }
                        cntA_ = (int)((wA >> sh) & FMASK) + (int)((qA >> sh4) & 15ull);  // before the update
                        cntB_ = (int)((wB >> sh) & FMASK) + (int)((qB >> sh4) & 15ull);
                        const uint64_t inc4 = 1ull << sh4;
                        if constexpr (NH == 2) {
                            dA[0] += (takeA && !(ct & 16)) ? inc4 : 0ull;
                            dA[1] += (takeA && (ct & 16)) ? inc4 : 0ull;
                            dB[0] += (!takeA && !(ct & 16)) ? inc4 : 0ull;
                            dB[1] += (!takeA && (ct & 16)) ? inc4 : 0ull;
                        } else {
                            dA[0] += takeA ? inc4 : 0ull;
                            dB[0] += takeA ? 0ull : inc4;
                        }
                        }
                        const int mine = takeA ? cntA_ : cntB_, other = takeA ? cntB_ : cntA_;
                        double delta = (sqrt_cnt(mine + 1) - sqrt_cnt(mine)) * sqrt_cnt(other);
                        if constexpr (MODE == MODE_H2W) {
                            const double wv_ = w_s[ct & 31];
                            delta *= wv_;
                            na += takeA ? wv_ : 0.0;
                            nb += takeA ? 0.0 : wv_;
                            const double r = 1.0 / sqrt(takeA ? na : nb);
                            ra = takeA ? r : ra;
                            rb = takeA ? rb : r;
                        } else if constexpr (LDSTAB && !LDSCNT) {
                            ra = rsqrt_cnt(totA);  // two table reads instead of one read and five selects (the LDS-count variants already
                                                   // queue five LDS operations per event: there the select form is the faster one)
                            rb = rsqrt_cnt(totB);
                        } else {
                            const double r = rsqrt_cnt(takeA ? totA : totB);
                            ra = takeA ? r : ra;
                            rb = takeA ? rb : r;
                        }
                        D += delta;
                    } else {
                        const double wv_ = w_s[ct & 31];
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) {
                            const bool hit = (c == ct);
                            va[c] += (hit && takeA) ? wv_ : 0.0;
                            vb[c] += (hit && !takeA) ? wv_ : 0.0;
                        }
                    }
                    Hp = distance();
                    Fp = F;
                }
            }
            STAMP(6);
            // stitch lane chunks: (F_first - F_last_of_previous_lane) * H_before_my_first_event
            double prevF = wave_shr1_f64(Fp), prevH = wave_shr1_f64(Hp);  // DPP, no LDS round trip
            if (lane == 0) { prevF = F_carry; prevH = H_carry; }
            if (d0 < d1) local += (firstF - prevF) * prevH;
            acc += local;
            const int last = (T - 1) / epl;  // wave-uniform
            F_carry = readlane_f64(Fp, last);
            H_carry = readlane_f64(Hp, last);
            ia += iend;
            ib += T - iend;
        }
        take_next();  // its load was issued before this pair's tile loads, which have all been waited for
        // wave64 reduction + the last interval to +inf (:165-171,204-210,212-221)
        acc = wave_sum_f64(acc);
        const double Finf = args.wf_index ? finf_tab[wfi] : Finf0;
        acc += (Finf - F_carry) * H_carry;
        const unsigned long long anyzero = __ballot(zero_norm);  // (categories were checked when the environments were built)
        if (lane == 0) {
            if (anyzero) sweep_report(args.hst, ST_ZERO_NORM);
            args.out[p] = acc;
        }
        STAMP(7);
    }
#ifdef LCHD_SWEEP_STAMPS
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_sweep_stamps[k], stamp_acc[k]);
#endif
    if constexpr (INLINE_META) {
        // what k_pair_meta's last workgroup does for the other sweeps: largest environment, status snapshot for the host,
        // device status reset for the next pass (n_small is not counted here: the host keeps its previous hint)
        __shared__ int big_s[WPB];
        __shared__ bool last_s;
        if (lane == 0) big_s[wv] = biggest_env;
        __syncthreads();
        if (tid == 0) {
            int b = big_s[0];
#pragma unroll
            for (int k = 1; k < WPB; ++k) b = max(b, big_s[k]);
            last_s = last_workgroup_done(args.done, 0ull, (uint32_t)b);
        }
        __syncthreads();
        if (last_s && tid < 64) {
            unsigned long long v_;
            uint32_t mx_;
            collect_done(args.done, tid, v_, mx_);
            for (int m = 32; m > 0; m >>= 1) mx_ = max(mx_, (uint32_t)__shfl_xor((int)mx_, m));
            if (tid == 0) publish_status(args, ~0ull, mx_);  // n_small is not counted here: the host keeps its previous hint
        }
    }
}

template <int MODE, int FMODE, bool LDSTAB>
static void launch_sweep_mode(hipStream_t s, int cmax, unsigned grid, const SweepArgs& a) {
    constexpr int NTH = 64 * kSweepWaves;
    if (cmax <= 8) k_sweep<8, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 12) k_sweep<12, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 16) k_sweep<16, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 20) k_sweep<20, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 24) k_sweep<24, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 28) k_sweep<28, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else k_sweep<32, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
}
template <int MODE, bool LDSTAB>
static void launch_sweep_f(hipStream_t s, int cmax, unsigned grid, int fmode, const SweepArgs& a) {
    if (fmode == F_KEY) launch_sweep_mode<MODE, F_KEY, LDSTAB>(s, cmax, grid, a);
    else if (fmode == F_FAST && MODE == MODE_H2U) launch_sweep_mode<MODE_H2U, F_FAST, LDSTAB>(s, cmax, grid, a);
    else launch_sweep_mode<MODE, F_ANY, LDSTAB>(s, cmax, grid, a);
}

void launch_sweep_plain(hipStream_t s, int mode, bool ldstab, int cmax, unsigned grid, int fmode, const SweepArgs& a) {
    if (mode == MODE_GEN) {
        if (ldstab) launch_sweep_f<MODE_GEN, true>(s, cmax, grid, fmode, a);
        else launch_sweep_f<MODE_GEN, false>(s, cmax, grid, fmode, a);
    } else if (mode == MODE_H2U) {
        if (ldstab) launch_sweep_f<MODE_H2U, true>(s, cmax, grid, fmode, a);
        else launch_sweep_f<MODE_H2U, false>(s, cmax, grid, fmode, a);
    } else {
        if (ldstab) launch_sweep_f<MODE_H2W, true>(s, cmax, grid, fmode, a);
        else launch_sweep_f<MODE_H2W, false>(s, cmax, grid, fmode, a);
    }
}
void launch_sweep_inline(hipStream_t s, int cm, unsigned g, const SweepArgs& a) {
    constexpr int NTH = 64 * kSweepWaves;
    if (cm <= 8) k_sweep<8, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
    else if (cm <= 12) k_sweep<12, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
    else if (cm <= 16) k_sweep<16, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
    else if (cm <= 20) k_sweep<20, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
    else if (cm <= 24) k_sweep<24, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
    else if (cm <= 28) k_sweep<28, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
    else k_sweep<32, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
}
void launch_sweep_c8(hipStream_t s, int cmax, unsigned grid, const SweepArgs& a) {
    constexpr int NTH = 64 * kSweepWaves;
    if (cmax <= 8) k_sweep<8, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 12) k_sweep<12, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 16) k_sweep<16, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 20) k_sweep<20, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 24) k_sweep<24, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 28) k_sweep<28, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a);
    else k_sweep<32, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a);
}
template <int CM>
static void launch_indirect_c(hipStream_t s, int tm, unsigned grid, const SweepArgs& a) {
    constexpr int NTH = 64 * kSweepWaves;
    if constexpr (CM <= 16) {
        if (tm == 2) { k_sweep<CM, MODE_GEN, F_KEY, false, true><<<grid, NTH, 0, s>>>(a); return; }
        if (tm == 1) { k_sweep<CM, MODE_H2W, F_KEY, true, true><<<grid, NTH, 0, s>>>(a); return; }
    }
    k_sweep<CM, MODE_H2U, F_KEY, true, true><<<grid, NTH, 0, s>>>(a);
}
void launch_sweep_indirect(hipStream_t s, int cmax, int tm, unsigned grid, const SweepArgs& a) {
    if (cmax <= 8) launch_indirect_c<8>(s, tm, grid, a);
    else if (cmax <= 12) launch_indirect_c<12>(s, tm, grid, a);
    else if (cmax <= 16) launch_indirect_c<16>(s, tm, grid, a);
    else if (cmax <= 20) launch_indirect_c<20>(s, tm, grid, a);
    else if (cmax <= 24) launch_indirect_c<24>(s, tm, grid, a);
    else if (cmax <= 28) launch_indirect_c<28>(s, tm, grid, a);
    else launch_indirect_c<32>(s, tm, grid, a);
}

}  // namespace lchd

#ifdef LCHD_SWEEP_STAMPS
extern "C" int lchd_debug_sweep_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lchd::g_sweep_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lchd::g_sweep_stamps), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif

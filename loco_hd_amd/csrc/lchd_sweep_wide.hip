// lchd_sweep_wide.hip -- K2 for configurations the register-resident sweeps do not take: 33 .. 65534 categories (per-lane count
// columns in LDS or, beyond 512 categories, in global memory) and environments of more than 65535 points (64-bit count words).
// The reference's category map is an arbitrary HashMap (/root/reference/src/locohd.rs:312-316) and its environments have no capacity.
#include "lchd_sweep_common.h"

namespace lchd {
constexpr int kWideMaxCat = kWideCategories;  // (512: the per-lane count columns, 256 bytes per category, must fit the LDS)

// Many-categories variant (32 < C <= 255): the per-lane category counts live in LDS columns instead of registers, all
// category loops are runtime loops, and the sqrt tables are read from global memory.  Slower per pair than k_sweep,
// but independent of the category count in registers.  WPB = anchor pairs (wavefronts) per workgroup.
// BIG: environments of more than 65 535 points (the reference sorts and sweeps any length, utils.rs:25-39): the two counts of a
// category are the halves of a 64-bit word instead of a 32-bit one, square roots beyond the 65 536-entry tables are computed.
// HUGE (more than kWideCategories categories, up to kHugeCategories): the per-lane count columns, the carry row and the generic
// distances' normalised vectors live in a global-memory scratch block per workgroup (SweepArgs::wide_scratch) instead of LDS /
// registers, the category weights are read from the configuration.  Nothing here is fast; it exists so that the reference's
// arbitrary category map (src/locohd.rs:312-316) has no upper size short of the 16-bit ids of the store.
template <int MODE, int FMODE, int WPB, bool CAT16 = false, bool BIG = false, bool HUGE = false>  // CAT16: 16-bit category ids in the environment store (EnvStore::cat16)
__global__ __launch_bounds__(64 * WPB) void k_sweep_wide(SweepArgs args) {
    static_assert(!(BIG && CAT16), "the pair record holds a 24-bit length next to an 8-bit category");
    static_assert(!HUGE || (CAT16 && !BIG && WPB == 1), "the global-memory form: 16-bit ids, one wavefront per workgroup");
    constexpr int kLdsCat = HUGE ? 1 : kWideMaxCat;
    using CT = typename std::conditional<CAT16, uint16_t, uint8_t>::type;
    using W = typename std::conditional<BIG, uint64_t, uint32_t>::type;  // count of side A | count of side B << SH
    constexpr int SH = BIG ? 32 : 16;
    constexpr W kOneA = (W)1, kOneB = (W)1 << SH, kMaskA = kOneB - 1;
    constexpr int TILE = kSweepTile;
    constexpr bool LDSTAB = false;
    constexpr bool H2 = (MODE != MODE_GEN);
    constexpr int NT = LDSTAB ? kSqrtTab + 8 : 1;  // sqrt(k), 1/sqrt(k) for k <= 512 in LDS; otherwise read from the global tables
    // Dynamic LDS: per-lane category counts, cnt[wave][category][lane] = count_A | count_B << 16.  A lane only
    // ever touches its own column, and column-major placement makes every access conflict-free.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    __shared__ double t_sqrt[NT], t_rsqrt[NT];
    __shared__ double w_s[kLdsCat], sw_s[kLdsCat];
    __shared__ W carry_[WPB][BIG ? 256 : kLdsCat];  // per category: counts before the current tile (A | B << SH)
    __shared__ uint64_t sA_[WPB][TILE], sB_[WPB][TILE];
    __shared__ CT cA_[WPB][TILE], cB_[WPB][TILE];
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
    for (int c = tid; c < kLdsCat; c += 64 * WPB) {
        const double wv_ = (!HUGE && c < C) ? cfgp->cat_w[c] : 0.0;
        w_s[c] = wv_;
        sw_s[c] = sqrt(wv_);
    }
    __syncthreads();
    uint64_t* sA = sA_[wv];
    uint64_t* sB = sB_[wv];
    CT* cA = cA_[wv];
    CT* cB = cB_[wv];
    W* carry = carry_[wv];
    W* cnt = reinterpret_cast<W*>(smem_dyn) + (size_t)wv * C * 64 + lane;  // this lane's column: cnt[c * 64]
    // HUGE: the same objects in this workgroup's scratch block: [C][64] count columns | [C] carry row | 2 x [64][C] doubles
    unsigned char* const hbase = HUGE ? args.wide_scratch + (size_t)blockIdx.x * (size_t)args.wide_scratch_per_wave : nullptr;
    W* const cnt_g = reinterpret_cast<W*>(hbase) + lane;
    W* const carry_g = reinterpret_cast<W*>(hbase + (size_t)C * 64 * sizeof(W));
    double* const pn_g = reinterpret_cast<double*>(hbase + (((size_t)C * 65 * sizeof(W) + 15) & ~(size_t)15)) + (size_t)lane * C;
    double* const qn_g = pn_g + (size_t)64 * C;
    auto cnt_at = [&](int c) -> W& { if constexpr (HUGE) return cnt_g[(size_t)c * 64]; else return cnt[c * 64]; };
    auto carry_at = [&](int c) -> W& { if constexpr (HUGE) return carry_g[c]; else return carry[c]; };
    auto wgt = [&](int c) -> double { if constexpr (HUGE) return cfgp->cat_w[c]; else return w_s[c]; };
    auto swgt = [&](int c) -> double { if constexpr (HUGE) return sqrt(cfgp->cat_w[c]); else return sw_s[c]; };
    // (HUGE: the carry row goes from lane 63 to lane 0 through global memory)
    auto wsync = [&]() { if constexpr (HUGE) { __threadfence(); __builtin_amdgcn_wave_barrier(); } wave_sync_lds(); };

    auto sqrt_cnt = [&](int k) -> double {
        if constexpr (LDSTAB) return t_sqrt[k];
        else if constexpr (BIG) return k < 65536 ? g_sqrt[k] : sqrt((double)k);  // (k_fill_sqrt_tables: the same expressions)
        else return g_sqrt[k];
    };
    auto rsqrt_cnt = [&](int k) -> double {
        if constexpr (LDSTAB) return t_rsqrt[k];
        else if constexpr (BIG) return k < 65536 ? g_rsqrt[k] : 1.0 / sqrt((double)k);
        else return g_rsqrt[k];
    };
    auto scan_counts = [&](W x) -> W {  // inclusive wave scan of both halves at once (no half overflows: counts stay below 2^SH)
        if constexpr (BIG) return (W)wave_incl_scan_u32((uint32_t)x) | ((W)wave_incl_scan_u32((uint32_t)(x >> 32)) << 32);
        else return wave_incl_scan_u32(x);
    };

    // One 16-byte record per pair (k_pair_meta) replaces the dependent chain anchors -> slot -> len -> first category; the
    // record of the wave's NEXT pair is requested before the current pair is processed.
    // configuration words the loop needs: read once (the compiler must assume the status atomics may alias *cfgp)
    const int n_wf = cfgp->n_wf;
    const double* __restrict__ finf_tab = cfgp->wf_finf;
    const double Finf0 = finf_tab[0];
    const int64_t pstride = (int64_t)gridDim.x * WPB;
    int64_t p = (int64_t)blockIdx.x * WPB + wv;
    // the record lives in four scalar registers; the next one is moved there as soon as its (early) load has returned, so the
    // loop's back edge never waits on vector memory (in particular not on the score store of the pair just finished)
    int mx, my, mz, mw;
    {
        const int4 m0 = args.meta[p < args.n_pairs ? p : 0];
        mx = __builtin_amdgcn_readfirstlane(m0.x); my = __builtin_amdgcn_readfirstlane(m0.y);
        mz = __builtin_amdgcn_readfirstlane(m0.z); mw = __builtin_amdgcn_readfirstlane(m0.w);
    }
    int nx = mx, ny = my, nz = mz, nw = mw;
    for (; p < args.n_pairs; p += pstride, mx = nx, my = ny, mz = nz, mw = nw) {
        const int4 mn = args.meta[p + pstride < args.n_pairs ? p + pstride : p];
        auto take_next = [&]() {
            nx = __builtin_amdgcn_readfirstlane(mn.x); ny = __builtin_amdgcn_readfirstlane(mn.y);
            nz = __builtin_amdgcn_readfirstlane(mn.z); nw = __builtin_amdgcn_readfirstlane(mn.w);
        };
        const int nA = CAT16 ? (mz & 0xFFFF) : (mz & 0xFFFFFF), nB = CAT16 ? (mw & 0xFFFF) : (mw & 0xFFFFFF);
        if (nA <= 0 || nB <= 0) {  // anchor out of range (flagged by k_mark_anchors) or overflow / empty environment (flagged by K1)
            if (lane == 0) args.out[p] = nan("");
            take_next();
            continue;
        }
        const int64_t ea = mx, eb = my;
        const int c0a = CAT16 ? ((mz >> 16) & 0xFFFF) : ((mz >> 24) & 255), c0b = CAT16 ? ((mw >> 16) & 0xFFFF) : ((mw >> 24) & 255);  // categories of the two anchors
        // (a dictionary's key sets, EnvStore::cdf_keys > 1: the set of this pair's weight function)
        const int kset = (FMODE == F_KEY && args.wf_index) ? args.wf_index[p] : 0;
        const int64_t kset_ok = (kset >= 0 && kset < n_wf) ? kset : 0;
        const uint64_t* __restrict__ kA = args.env_a.key + ea * args.env_a.stride + kset_ok * args.env_a.set_stride;
        const uint64_t* __restrict__ kB = args.env_b.key + eb * args.env_b.stride + kset_ok * args.env_b.set_stride;
        const CT* __restrict__ tA = reinterpret_cast<const CT*>(args.env_a.cat) + ea * args.env_a.stride;
        const CT* __restrict__ tB = reinterpret_cast<const CT*>(args.env_b.cat) + eb * args.env_b.stride;
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

        bool bad_cat = false, zero_norm = false;
        // ---- per-lane state (category counts live in LDS) ------------------------------------------------
        int totA = 1, totB = 1;      // points seen per side (incl. anchor)
        double ra = 1.0, rb = 1.0;   // H2: 1/sqrt(total weight)
        double na = 0.0, nb = 0.0;   // H2W: total weights
        double D = 0.0;              // H2: sum_c sqrt(a_c * b_c)  (Bhattacharyya numerator)

        // exact squared Hellinger distance in the literal difference-of-roots form (statistical_distances.rs:4-10)
        auto exact_h2 = [&]() -> double {
            double acc2 = 0.0;
            for (int c = 0; c < C; ++c) {
                const W v = cnt_at(c);
                double xa = sqrt_cnt((int)(v & kMaskA)), xb = sqrt_cnt((int)(v >> SH));
                if constexpr (MODE == MODE_H2W) { xa *= swgt(c); xb *= swgt(c); }
                const double d = xa * ra - xb * rb;  // equal inputs cancel exactly
                acc2 = fma(d, d, acc2);
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
                double pn_l[kLdsCat], qn_l[kLdsCat];
                double* const pn = HUGE ? pn_g : pn_l;
                double* const qn = HUGE ? qn_g : qn_l;
                double sa_ = 0.0, sb_ = 0.0;  // pmf.rs:67-68: fresh sums
                for (int c = 0; c < C; ++c) {
                    const W v = cnt_at(c);
                    pn[c] = wgt(c) * (double)(v & kMaskA);
                    qn[c] = wgt(c) * (double)(v >> SH);
                    sa_ += pn[c];
                    sb_ += qn[c];
                }
                if (sa_ == 0.0 || sb_ == 0.0) zero_norm = true;
                const double ia_ = 1.0 / sa_, ib_ = 1.0 / sb_;
                for (int c = 0; c < C; ++c) { pn[c] *= ia_; qn[c] *= ib_; }
                return sd_generic(cfgp->sd_kind, cfgp->sd_p0, cfgp->sd_p1, pn, qn, C);
            }
        };

        // seed with the two anchors (:82-84): carry row and every lane's column
        if (c0a >= C || c0b >= C) bad_cat = true;
        wsync();
        for (int c = lane; c < C; c += 64) carry_at(c) = (c == c0a ? kOneA : (W)0) | (c == c0b ? kOneB : (W)0);
        for (int c = 0; c < C; ++c) cnt_at(c) = (c == c0a ? kOneA : (W)0) | (c == c0b ? kOneB : (W)0);
        if constexpr (H2) {
            // (a category outside the map: bad_cat, the score is NaN whatever is computed here)
            const int w0a = HUGE ? (c0a < C ? c0a : 0) : (c0a & (kWideMaxCat - 1)), w0b = HUGE ? (c0b < C ? c0b : 0) : (c0b & (kWideMaxCat - 1));
            if (c0a == c0b && !bad_cat) D = (MODE == MODE_H2W) ? wgt(w0a) : 1.0;
            if constexpr (MODE == MODE_H2W) {
                na = wgt(w0a);
                nb = wgt(w0b);
                ra = 1.0 / sqrt(na);
                rb = 1.0 / sqrt(nb);
            }
        }
        wsync();
        double F_carry = cdf_of_key(kA[0]);  // F(0): both anchors sit at distance 0
        double H_carry = bad_cat ? 0.0 : distance();
        double acc = 0.0;

        const int mA = nA - 1, mB = nB - 1, M = mA + mB;  // non-anchor events
        int ia = 0, ib = 0;
        for (int k0 = 0; k0 < M; k0 += TILE) {
            const int T = min(TILE, M - k0);
            const int nAt = min(TILE, mA - ia), nBt = min(TILE, mB - ib);
            wsync();  // previous tile fully consumed
            for (int t = lane; t < nAt; t += 64) { sA[t] = kA[1 + ia + t]; cA[t] = tA[1 + ia + t]; }
            for (int t = lane; t < nBt; t += 64) { sB[t] = kB[1 + ib + t]; cB[t] = tB[1 + ib + t]; }
            wave_sync_lds();
            // lane l owns merged events [d0, d1); each lane searches the END of its chunk
            const int epl = (T + 63) >> 6;
            const int d0 = min(lane * epl, T), d1 = min(d0 + epl, T);
            const int i1 = merge_path(sA, nAt, sB, nBt, d1);
            int i0 = __shfl_up(i1, 1);
            if (lane == 0) i0 = 0;
            const int iend = __builtin_amdgcn_readlane(i1, 63);
            const int j0 = d0 - i0, j1 = d1 - i1;

            // pass 1: histogram of this lane's chunk into its LDS column
            for (int c = 0; c < C; ++c) cnt_at(c) = (W)0;
            for (int i = i0; i < i1; ++i) {
                const int ct = cA[i];
                if (ct >= C) bad_cat = true; else cnt_at(ct) += kOneA;
            }
            for (int j = j0; j < j1; ++j) {
                const int ct = cB[j];
                if (ct >= C) bad_cat = true; else cnt_at(ct) += kOneB;
            }
            // per category: wave64 inclusive scan (the carry of earlier tiles enters through lane 0); the exclusive
            // prefix = counts at this lane's first event.  Both 16-bit halves scan at once (every count < 65536).
            totA = 1 + ia + i0;
            totB = 1 + ib + j0;
            if constexpr (H2) D = 0.0;
            if constexpr (MODE == MODE_H2W) na = nb = 0.0;
            for (int c = 0; c < C; ++c) {
                const W own = cnt_at(c);
                const W incl = scan_counts(own + (lane == 0 ? carry_at(c) : (W)0));
                const W excl = incl - own;
                cnt_at(c) = excl;
                if (lane == 63) carry_at(c) = incl;
                if constexpr (H2) {
                    const int ca = (int)(excl & kMaskA), cb = (int)(excl >> SH);
                    if constexpr (MODE == MODE_H2W) {
                        D += wgt(c) * (sqrt_cnt(ca) * sqrt_cnt(cb));
                        na += wgt(c) * (double)ca;
                        nb += wgt(c) * (double)cb;
                    } else {
                        D += sqrt_cnt(ca) * sqrt_cnt(cb);
                    }
                }
            }
            if constexpr (MODE == MODE_H2W) { ra = 1.0 / sqrt(na); rb = 1.0 / sqrt(nb); }
            else if constexpr (MODE == MODE_H2U) { ra = rsqrt_cnt(totA); rb = rsqrt_cnt(totB); }

            // pass 2: sequential sweep of this lane's events (the two list heads stay in registers)
            int i = i0, j = j0;
            uint64_t ka = (i < i1) ? sA[i] : kPadKey, kb = (j < j1) ? sB[j] : kPadKey;
            double Fp = 0.0, Hp = 0.0, firstF = 0.0, local = 0.0;
            for (int e = d0; e < d1; ++e) {
                const bool takeA = (ka <= kb);  // an exhausted list shows the pad key (> every real key)
                const uint64_t key = takeA ? ka : kb;
                const int ct = takeA ? cA[i] : cB[j];
                if (takeA) { ++i; ka = (i < i1) ? sA[i] : kPadKey; } else { ++j; kb = (j < j1) ? sB[j] : kPadKey; }
                const double F = cdf_of_key(key);
                if (e == d0) firstF = F; else local += (F - Fp) * Hp;
                // pmf.rs:47-63: one more point of category ct on one side
                const bool okc = ct < C;
                const int cs = okc ? ct : 0;
                const W old = cnt_at(cs);
                cnt_at(cs) = old + (okc ? (takeA ? kOneA : kOneB) : (W)0);
                totA += takeA ? 1 : 0;
                totB += takeA ? 0 : 1;
                if constexpr (H2) {
                    const int cntA_ = (int)(old & kMaskA), cntB_ = (int)(old >> SH);
                    const int mine = takeA ? cntA_ : cntB_, other = takeA ? cntB_ : cntA_;
                    double delta = (sqrt_cnt(mine + 1) - sqrt_cnt(mine)) * sqrt_cnt(other);
                    if constexpr (MODE == MODE_H2W) {
                        const double wv_ = wgt(cs);
                        delta *= wv_;
                        na += takeA ? wv_ : 0.0;
                        nb += takeA ? 0.0 : wv_;
                        const double r = 1.0 / sqrt(takeA ? na : nb);
                        ra = takeA ? r : ra;
                        rb = takeA ? rb : r;
                    } else {
                        const double r = rsqrt_cnt(takeA ? totA : totB);
                        ra = takeA ? r : ra;
                        rb = takeA ? rb : r;
                    }
                    D += okc ? delta : 0.0;
                }
                Hp = distance();
                Fp = F;
            }
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
        const unsigned long long anybad = __ballot(bad_cat), anyzero = __ballot(zero_norm);
        if (lane == 0) {
            if (anybad) { sweep_report(args.hst, ST_BAD_CATEGORY); acc = nan(""); }
            if (anyzero) sweep_report(args.hst, ST_ZERO_NORM);
            args.out[p] = acc;
        }
    }
}


template <int MODE>
static void launch_sweep_wide_m(hipStream_t s, int n_cat, int64_t n_pairs, int fmode, const SweepArgs& a) {
    // dynamic LDS = WPB * C * 64 * 4 bytes of per-lane count columns
    if (a.env_a.stride > 65535 || a.env_b.stride > 65535) {  // environments of more than 65 535 points: 64-bit count words (<= 255 categories: the host checks)
        const unsigned grid = (unsigned)(n_pairs < 8192 ? n_pairs : 8192);
        const size_t dyn = (size_t)n_cat * 512;
        if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1, false, true><<<grid, 64, dyn, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 1, false, true><<<grid, 64, dyn, s>>>(a);
    } else if (n_cat <= 64) {
        const int64_t blocks = (n_pairs + 3) / 4;
        const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
        const size_t dyn = (size_t)4 * n_cat * 256;
        if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 4><<<grid, 256, dyn, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 4><<<grid, 256, dyn, s>>>(a);
    } else if (n_cat > kWideCategories) {  // the global-memory form (the caller has checked that the scratch block exists)
        const unsigned grid = (unsigned)std::min<int64_t>(n_pairs, a.wide_scratch_waves);
        if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1, true, false, true><<<grid, 64, 0, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 1, true, false, true><<<grid, 64, 0, s>>>(a);
    } else {
        const unsigned grid = (unsigned)(n_pairs < 8192 ? n_pairs : 8192);
        const size_t dyn = (size_t)n_cat * 256;  // (> 64 KB from 257 categories' worth on: init_device_kernels raised the limit)
        if (a.env_a.cat16) {  // more than 255 categories: 16-bit ids in the store
            if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1, true><<<grid, 64, dyn, s>>>(a);
            else k_sweep_wide<MODE, F_ANY, 1, true><<<grid, 64, dyn, s>>>(a);
        } else if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1><<<grid, 64, dyn, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 1><<<grid, 64, dyn, s>>>(a);
    }
}

void launch_sweep_wide(hipStream_t s, int mode, int n_cat, int64_t n_pairs, int fmode, const SweepArgs& a) {
    if (mode == MODE_GEN) launch_sweep_wide_m<MODE_GEN>(s, n_cat, n_pairs, fmode, a);
    else if (mode == MODE_H2U) launch_sweep_wide_m<MODE_H2U>(s, n_cat, n_pairs, fmode, a);
    else launch_sweep_wide_m<MODE_H2W>(s, n_cat, n_pairs, fmode, a);
}
void init_sweep_wide_kernels() {
    auto raise = [](const void* fn, int bytes) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_KEY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_ANY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_KEY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_ANY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_KEY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_ANY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_KEY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_ANY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_KEY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_ANY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_KEY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_ANY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_KEY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_ANY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_KEY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_ANY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_KEY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_ANY, 1, true>), kWideCategories * 256);
    (void)hipGetLastError();
}

}  // namespace lchd

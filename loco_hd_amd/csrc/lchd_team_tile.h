// lchd_team_tile.h -- ONE tile of the team sweeps: the merged events of one anchor pair swept by a TEAM of 16 or 32 lanes
// (reference: LoCoHD::stat_dist_integral, /root/reference/src/locohd.rs:61-226; PMFSystem, src/locohd/pmf.rs:47-88;
// hellinger_distance / kolmogorov_smirnov_distance, src/locohd/pmf/statistical_distances.rs:4-21).
//
// Used by k_sweep_duo (lchd_sweep_team.hip: both environments staged from the environment store).  The caller stages the two sorted lists (keys = bits of F(distance), ascending; categories; anchors excluded) into LDS and hands
// over pointers; everything from the merge-path partition to the team's reduced integral happens here.
#pragma once
#include <type_traits>

#include "lchd_kcommon.h"

#ifndef LCHD_CAT_HEADS
#define LCHD_CAT_HEADS 1      // k_sweep / k_sweep_duo: the categories of both list heads are read together with their keys
#endif
#ifndef LCHD_TEAM_EXACT_UNROLL_MAX
#define LCHD_TEAM_EXACT_UNROLL_MAX 8    // category slots up to which the (rare) literal Hellinger form of k_sweep_duo is unrolled (above: a rolled loop -- the unrolled look-ups of 12+ slots cost registers in the event loop)
#endif
#ifndef LCHD_TEAM_LOOP_UNROLL
#define LCHD_TEAM_LOOP_UNROLL 1
#endif
#ifndef LCHD_TEAM_LDSCNT
#define LCHD_TEAM_LDSCNT 1   // k_sweep_duo with 20 .. 28 category slots: per-lane counts of the event loop in LDS bytes (0: packed registers)
#endif
#ifndef LCHD_LCNT_HIST
#define LCHD_LCNT_HIST 1     // the instantiations with per-lane counts in LDS bytes build the chunk histogram there too (LDS adds) when that takes two 4-bit words (17 and more slots; with one word the register form is as fast); 0: always in registers
#endif
#ifndef LCHD_WGT_LDSCNT
#define LCHD_WGT_LDSCNT 1   // 1: the weighted instantiations with 9 .. 16 slots keep their per-lane counts in LDS bytes too
#endif

namespace lchd {

constexpr int kDuoTile = 240;  // merged events per pair: 16 lanes x 15, the most the 4-bit chunk fields take (224 = 16 x 14 until late in round 3: at ~95 points per environment 8.7 % of C4's pairs were longer than that, 2.3 % are longer than 240 -- C4 sweep 3.27 -> 3.19 ms, C3 0.867 -> 0.837)
constexpr int kCount8MaxEnv = 255;        // the 8-bit-count sweep takes pairs whose environments both have at most this many points
constexpr int kTeam8Tile = 480;           // ... and its two-pairs-per-wavefront form (k_sweep_duo<CMAX, 32, 480>) those of at most 32 x 15 merged events
// which pairs the small-pair kernel of a launch sweeps (SweepArgs::small_rule); nA, nB: environment sizes incl. the anchor, both > 0
__device__ __forceinline__ bool pair_is_small(int rule, int nA, int nB) {
    if (rule == 0) return nA + nB - 2 <= kDuoTile;
    const bool c8 = max(nA, nB) <= kCount8MaxEnv;
    return rule == 1 ? c8 : (c8 && nA + nB - 2 <= kTeam8Tile);
}
// H^2 = 1 - D / sqrt(N_a N_b) carries an absolute rounding error of a few 1e-16 (D is rebuilt from the exact integer counts at
// every lane chunk, so nothing drifts); sqrt() turns that into an error of ~3e-16 / (2 sqrt(H^2)) in H.  Below this bound the
// literal difference-of-roots form is evaluated instead (exactly 0 for identical environments); at the bound the cancellation
// form is still good to ~2e-13.  (It used to be 1e-3: large random clouds -- dense from_coords rows -- sit at H^2 ~ 1e-4 and
// paid the O(C) literal form with 2C square roots on nearly every event.)
constexpr double kExactH2Below = 1e-6;

__device__ __forceinline__ int merge_path(const uint64_t* A, int nA, const uint64_t* B, int nB, int d) {
    int lo = max(0, d - nB), hi = min(d, nA);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// spread the eight 4-bit fields of the low 32 bits of x into eight 8-bit fields
__device__ __forceinline__ uint64_t spread8(uint64_t x) {
    const uint32_t v = (uint32_t)x;
    auto half = [](uint32_t h) -> uint32_t {  // four nibbles (16 bits) -> four bytes
        const uint32_t t = (h | (h << 8)) & 0x00FF00FFu;
        return (t | (t << 4)) & 0x0F0F0F0Fu;
    };
    return ((uint64_t)half(v >> 16) << 32) | half(v & 0xFFFFu);
}

// sqrt for x in [0, ~1]: v_rsq_f64 seed + Goldschmidt refinement (the same scheme the compiler's IEEE sqrt uses, minus
// its range scaling and special-case fix-ups, which H^2 in [0, 1] never needs).  Result within 1 ulp; sqrt(0) = 0.
__device__ __forceinline__ double sqrt_unit(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return fmax(g, 0.0);  // x == 0: rsq gives inf, the chain NaN, and fmax returns its non-NaN operand: sqrt(0) = 0 in one instruction
}

// inclusive scan / sum inside each team of TL consecutive lanes (TL = 16: one DPP row; 32: two rows joined by row_bcast:15)
template <int TL>
__device__ __forceinline__ uint32_t team_incl_scan_u32(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    if constexpr (TL == 32) v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    return (uint32_t)v;
}
template <int TL>
__device__ __forceinline__ uint64_t team_incl_scan_fields(uint64_t x) {
    const uint32_t lo = team_incl_scan_u32<TL>((uint32_t)x), hi = team_incl_scan_u32<TL>((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int TL>
__device__ __forceinline__ double team_sum_f64(double v) {  // the last lane of every team ends up with the team's sum
    v += dpp_mov_f64_or_zero<0x111, 0xf>(v);
    v += dpp_mov_f64_or_zero<0x112, 0xf>(v);
    v += dpp_mov_f64_or_zero<0x114, 0xf>(v);
    v += dpp_mov_f64_or_zero<0x118, 0xf>(v);
    if constexpr (TL == 32) v += dpp_mov_f64_or_zero<0x142, 0xa>(v);
    return v;
}

// CMAX category slots, teams of TL lanes, pairs of at most TILE_ merged events; WGT: category weights other than 1; KSM: the
// Kolmogorov-Smirnov distance on unit weights (see k_sweep_duo for the two forms).
// PRE: the packed counts at a lane's chunk start are READ from the environments' prefix-count rows (EnvStore::pre: row j = the counts
// of the first kPreStep * j + 1 sorted points of the environment, the anchor included, as 8-bit fields -- written once per environment
// by k_env_group) instead of being built per tile from a histogram of the lane's chunk and a scan over the team: the position i the
// merge path hands a lane names row i / kPreStep, and the one-hot fields of the i % kPreStep staged category bytes between that row
// and the chunk's start are added here.  An environment that is swept many times (C2a: 100 pairs per anchor) pays for its rows once.
// (Round 5 stored a row per point: a lane's 16-byte read then touched a 128-byte line of its own -- 8.0 GB per 10^6 C2a pairs.)
template <int CMAX, int TL, int TILE_, bool WGT = false, bool KSM = false, bool PRE = false>
struct TeamTile {
    static_assert(!KSM || (!WGT && CMAX <= 16), "the Kolmogorov-Smirnov form: unit weights, one or two count words per side");
    static_assert(TL == 16 || TL == 32, "a team is one or two DPP rows");
    static constexpr int TEAMS = 64 / TL, EPL = TILE_ / TL, TILE = TILE_;
    // category counts as 8-bit fields: no count of a pair of this kernel exceeds 255 (TILE 240: at most 242 points in all; TILE 480:
    // environments of at most 255 points) -- one word per side up to 8 category slots, two up to 16: half the scans and no word select
    // for the common 8-slot case
    static constexpr int FPW = 8, FB = 8;
    static constexpr int NW = (CMAX + FPW - 1) / FPW;  // u64 words of count fields per side
    static constexpr int NH = (CMAX + 15) / 16;        // words of 4-bit chunk fields per side (two from 17 category slots on)
    static_assert(CMAX <= 32, "two words of 4-bit chunk fields");
    static_assert(TILE_ == kDuoTile || TILE_ == kTeam8Tile, "the two rules of pair_is_small");
    static_assert(kDuoTile + 1 < 256 && kCount8MaxEnv < 256, "8-bit count fields");
    static_assert(EPL * TL == TILE_ && EPL <= 15, "4-bit chunk-local counters");
    static constexpr int NT = 256 + 8;  // entries of the sqrt / 1/sqrt tables (no count and no total of these pairs exceeds 256)
    // 4-bit chunk-local fields, one per category slot: 32 bits hold them up to 8 slots (half the selects and adds of a 64-bit word)
    using H4 = typename std::conditional<(CMAX <= 8), uint32_t, uint64_t>::type;
    // 20 .. 28 category slots (three or four count words per side): the per-lane counts of the event loop live in LDS BYTES -- slot
    // c of side A at byte c, of side B at byte CMAX + c of the lane's row, [word][lane][8 bytes] so that a lane writes its chunk-start
    // counts as whole words and no two lanes of a 16-lane group share a bank.  Two byte reads at computed addresses and one byte
    // write replace the word-select chains over the count words and the 4-bit chunk fields (C5, 28 slots: 25 of the event's 88
    // vector instructions were v_cndmask_b32_e64).  The rows cost (2 CMAX / 8) x 512 bytes per wavefront: three workgroups per CU
    // -- what these instantiations are compiled for -- still fit up to 28 slots; with 32 they would not (registers there).
    static constexpr bool LCNT = (((CMAX > 16) && (CMAX <= 28)) || (WGT && (LCHD_WGT_LDSCNT != 0) && CMAX > 8 && CMAX <= 16)) && (LCHD_TEAM_LDSCNT != 0);
    static constexpr int LW = LCNT ? (2 * CMAX + 7) / 8 : 1;  // u64 words of a lane's LDS count row ([LW][64] per wavefront)
    static_assert(!PRE || (!LCNT && !WGT && CMAX <= 16), "prefix-count rows: the register-resident forms of up to two count words per side");

    // sA / cA, sB / cB: the two staged lists (mA, mB points; spare entries behind each: the head re-reads may touch one past the end);
    // T = mA + mB; epl = ceil(T / TL); epl_w = the largest epl of the wavefront's teams (wave-uniform trip counts);
    // c0a / c0b: the anchors' categories; F0 = F(0); Finf0 = F(+inf); t_sqrt / t_rsqrt: [NT] tables in LDS; w_s: [32] category
    // weights in LDS (WGT); lcl: this lane's eight bytes of word 0 of the wavefront's count rows (LCNT).
    // Returns the pair's integral in the LAST lane of the team (lane tl == TL - 1).
    static __device__ __forceinline__ double run(const uint64_t* sA, const uint8_t* cA, const uint64_t* sB, const uint8_t* cB, const int mA,
                                                 const int mB, const int T, const int epl, const int epl_w, const int c0a, const int c0b,
                                                 const double F0, const double Finf0, const double* t_sqrt, const double* t_rsqrt,
                                                 const double* w_s, unsigned char* lcl, const int tl,
                                                 const uint64_t* __restrict__ preA = nullptr, const uint64_t* __restrict__ preB = nullptr) {
        auto field = [&](const uint64_t (&ex)[NW], int c) -> int { return (int)((ex[c / FPW] >> ((c % FPW) * FB)) & 0xFFull); };
        const double H0 = (c0a == c0b) ? 0.0 : 1.0;            // two point masses
        // lane tl of a team owns merged events [d0, d1) of its pair
        const int d0 = min(tl * epl, T), d1 = min(d0 + epl, T);
        const int i1 = merge_path(sA, mA, sB, mB, d1);
        int i0 = __builtin_amdgcn_update_dpp(i1, i1, 0x138, 0xf, 0xf, false);  // wave_shr:1
        if (tl == 0) i0 = 0;
        const int j0 = d0 - i0, j1 = d1 - i1;

        uint64_t exA[NW], exB[NW];
        if constexpr (PRE) {
            // the rows at or below the two chunk starts (i0 / j0 non-anchor points of A / B lie before this lane's first event), and the
            // one-hot fields of the up to kPreStep - 1 points between a row and its chunk start
            const int ra_ = i0 / kPreStep, rb_ = j0 / kPreStep;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                exA[k] = preA[(size_t)ra_ * NW + k];
                exB[k] = preB[(size_t)rb_ * NW + k];
            }
#pragma unroll
            for (int m = 0; m < kPreStep - 1; ++m) {
                const unsigned ca_ = cA[min(ra_ * kPreStep + m, max(mA - 1, 0))], cb_ = cB[min(rb_ * kPreStep + m, max(mB - 1, 0))];
                const uint64_t oa_ = m < (i0 % kPreStep) ? (1ull << ((ca_ % FPW) * FB)) : 0ull, ob_ = m < (j0 % kPreStep) ? (1ull << ((cb_ % FPW) * FB)) : 0ull;
                if constexpr (NW == 1) { exA[0] += oa_; exB[0] += ob_; }
                else {
#pragma unroll
                    for (int k = 0; k < NW; ++k) { exA[k] += (int)(ca_ / FPW) == k ? oa_ : 0ull; exB[k] += (int)(cb_ / FPW) == k ? ob_ : 0ull; }
                }
            }
        } else {
        // pass 1: 4-bit-per-category histogram of the lane's chunk
        H4 hA[NH], hB[NH];
#pragma unroll
        for (int w = 0; w < NH; ++w) hA[w] = hB[w] = 0;
        // ... LCNT: in the lane's LDS row instead -- the row the event loop keeps its running counts in is zeroed, every point of the chunk
        // is ONE non-returning 32-bit LDS add of 1 << (8 x byte) (a chunk holds at most 15 points: no byte overflows into its neighbour),
        // and the row read back IS the chunk's counts as 8-bit fields, A's CMAX bytes then B's: five instructions per point instead of the
        // 24 of the two-word 4-bit form, and no 4-bit -> 8-bit spreading afterwards
        uint64_t hw[LW];
        if constexpr (LCNT && NH > 1 && (LCHD_LCNT_HIST != 0)) {
#pragma unroll
            for (int k = 0; k < LW; ++k) *reinterpret_cast<uint64_t*>(lcl + k * 512) = 0ull;
            const int nAl = i1 - i0, nl = d1 - d0;
            const uint8_t* pa_ = cA + i0;
            const uint8_t* pb_ = cB + (j0 - nAl);
#pragma unroll
            for (int m = 0; m < EPL; ++m) {
                if (m < epl_w) {
                    const bool isA = m < nAl;
                    const unsigned ct = (isA ? pa_ : pb_)[m < nl ? m : 0];
                    const unsigned b = ct + (isA ? 0u : (unsigned)CMAX);
                    if (m < nl) atomicAdd(reinterpret_cast<unsigned*>(lcl + ((b >> 3) << 9) + (b & 4u)), 1u << ((b & 3u) * 8u));
                }
            }
#pragma unroll
            for (int k = 0; k < LW; ++k) hw[k] = *reinterpret_cast<const uint64_t*>(lcl + k * 512);
        } else
        {   // one fixed-trip loop over the chunk's points, A's run first (see k_sweep)
            const int nAl = i1 - i0, nl = d1 - d0;
            const uint8_t* pa_ = cA + i0;
            const uint8_t* pb_ = cB + (j0 - nAl);
            H4 hT[NH];
#pragma unroll
            for (int w = 0; w < NH; ++w) hT[w] = 0;
#pragma unroll
            for (int m = 0; m < EPL; ++m) {
                if (m < epl_w) {  // wave-uniform (the longest chunk of the wavefront's teams): whole rounds are skipped; a lane's own bound is m < nl
                    const bool isA = m < nAl;
                    const int ct = (isA ? pa_ : pb_)[m < nl ? m : 0];
                    const H4 inc = (m < nl) ? ((H4)1 << ((ct & 15) * 4)) : (H4)0;
                    if constexpr (NH == 1) {
                        hT[0] += inc;
                        hA[0] += isA ? inc : (H4)0;
                    } else {
                        const bool hi = (ct & 16) != 0;
                        hT[0] += hi ? (H4)0 : inc;
                        hT[1] += hi ? inc : (H4)0;
                        hA[0] += (isA && !hi) ? inc : (H4)0;
                        hA[1] += (isA && hi) ? inc : (H4)0;
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < NH; ++w) hB[w] = hT[w] - hA[w];
        }
        // packed counts at the start of the chunk: the anchors + an exclusive scan over the team's lanes
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            uint64_t va_, vb_;
            if constexpr (LCNT && NH > 1 && (LCHD_LCNT_HIST != 0)) {
                // count word k of a side = its bytes [8 k, 8 k + 8) in the row (A's CMAX bytes | B's CMAX bytes, CMAX a multiple of 4)
                va_ = hw[k];
                if (8 * k + 8 > CMAX) va_ &= 0xFFFFFFFFull;  // (A's last four categories; the upper half is B's first)
                if constexpr (CMAX % 8 == 0) vb_ = hw[CMAX / 8 + k];
                else vb_ = (hw[CMAX / 8 + k] >> 32) | ((CMAX / 8 + k + 1 < LW) ? (hw[CMAX / 8 + k + 1] << 32) : 0ull);
            } else {
                va_ = spread8((uint64_t)hA[(k * 8) / 16] >> (((k * 8) % 16) * 4));
                vb_ = spread8((uint64_t)hB[(k * 8) / 16] >> (((k * 8) % 16) * 4));
            }
            uint64_t sa_, sb_;
            if (FPW * k + FPW / 2 >= CMAX) {  // (compile-time after unrolling: the side's last count word holds four categories: its upper half stays zero)
                sa_ = team_incl_scan_u32<TL>((uint32_t)va_);
                sb_ = team_incl_scan_u32<TL>((uint32_t)vb_);
            } else {
                sa_ = team_incl_scan_fields<TL>(va_);
                sb_ = team_incl_scan_fields<TL>(vb_);
            }
            exA[k] = (((c0a / FPW) == k) ? (1ull << ((c0a % FPW) * FB)) : 0ull) + sa_ - va_;
            exB[k] = (((c0b / FPW) == k) ? (1ull << ((c0b % FPW) * FB)) : 0ull) + sb_ - vb_;
        }
        }
        double D = 0.0;  // (points seen per side incl. the anchor: 1 + i and 1 + j -- the list positions ARE the totals)
        double na = 0.0, nb = 0.0;  // WGT: weighted totals of the two sides
#pragma unroll
        for (int k = 0; k < NW; ++k) {
#pragma unroll
            for (int f = 0; f < FPW; ++f) {
                const int c = FPW * k + f;
                if (c < CMAX) {
                    if constexpr (WGT) {
                        const int fa = field(exA, c), fb = field(exB, c);
                        D = fma(w_s[c], t_sqrt[fa] * t_sqrt[fb], D);
                        na = fma(w_s[c], (double)fa, na);
                        nb = fma(w_s[c], (double)fb, nb);
                    } else {
                        D = fma(t_sqrt[field(exA, c)], t_sqrt[field(exB, c)], D);
                    }
                }
                if ((f & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        auto rsqrt_w = [](double x) -> double {  // 1 / sqrt(weighted total): v_rsq_f64 + two Newton steps
            double y = __builtin_amdgcn_rsq(x);
            y = y * fma(-0.5 * x, y * y, 1.5);
            y = y * fma(-0.5 * x, y * y, 1.5);
            return y;
        };
        double ra = WGT ? rsqrt_w(na) : t_rsqrt[1 + i0], rb = WGT ? rsqrt_w(nb) : t_rsqrt[1 + j0];
        if constexpr (LCNT) {
            // bytes [8 k, 8 k + 8) of  A's CMAX count bytes | B's CMAX count bytes  (CMAX is a multiple of 4: B starts on a word or a half word)
#pragma unroll
            for (int k = 0; k < LW; ++k) {
                uint64_t w;
                if (8 * k + 8 <= CMAX) w = exA[k];
                else if (8 * k < CMAX) w = (exA[k] & 0xFFFFFFFFull) | (exB[0] << 32);
                else if (CMAX % 8 == 0) w = exB[(8 * k - CMAX) / 8];
                else w = (exB[(8 * k - CMAX) / 8] >> 32) | ((8 * k - CMAX) / 8 + 1 < NW ? exB[(8 * k - CMAX) / 8 + 1] << 32 : 0ull);
                *reinterpret_cast<uint64_t*>(lcl + k * 512) = w;
            }
        }

        // pass 2 (same scheme as k_sweep): both list heads in registers, chunk-local additions in 4-bit fields
        int i = i0, j = j0;
#if LCHD_CAT_HEADS
        uint64_t ka = sA[i], kb = sB[j];
        int cta = cA[i], ctb = cB[j];
#else
        uint64_t ka = (i < i1) ? sA[i] : kPadKey, kb = (j < j1) ? sB[j] : kPadKey;
#endif
        H4 dA[NH], dB[NH];
#pragma unroll
        for (int w = 0; w < NH; ++w) dA[w] = dB[w] = 0;
        double Fp = 0.0, Hp = 0.0, local = 0.0;
#if LCHD_CAT_HEADS
        // F of the chunk's first event, for the stitching below: known from the heads.  Inside the loop the first event adds
        // (F - 0) * 0 = 0 like any other -- no "first event" selects per event.
        const double firstF = u2d(((i < i1) & ((j >= j1) | (ka <= kb))) ? ka : kb);
#else
        double firstF = 0.0;
#endif
#pragma unroll LCHD_TEAM_LOOP_UNROLL
        for (int e = 0; e < epl_w; ++e) {
            if (d0 + e < d1) {
#if LCHD_CAT_HEADS
                // both heads and their categories are re-read after every event (see k_sweep); run ends are tested on the indices
                const bool takeA = (i < i1) & ((j >= j1) | (ka <= kb));
                const uint64_t key = takeA ? ka : kb;
                const int ct = takeA ? cta : ctb;
                i += takeA ? 1 : 0;
                j += takeA ? 0 : 1;
                ka = sA[i];  // (one past the run's end at most: the buffers have spare entries)
                kb = sB[j];
                cta = cA[i];
                ctb = cB[j];
#else
                const bool takeA = (ka <= kb);
                const uint64_t key = takeA ? ka : kb;
                const int ct = (takeA ? cA : cB)[takeA ? i : j];
                i += takeA ? 1 : 0;
                j += takeA ? 0 : 1;
                {
                    const int nidx = takeA ? i : j, nend = takeA ? i1 : j1;
                    const uint64_t nk = (takeA ? sA : sB)[nidx];
                    const uint64_t nh = nidx < nend ? nk : kPadKey;
                    ka = takeA ? nh : ka;
                    kb = takeA ? kb : nh;
                }
#endif
                const double F = u2d(key);
#if LCHD_CAT_HEADS
                local = fma(F - Fp, Hp, local);  // (fused on purpose, like the two updates below: one rounding less and one instruction less per
                                                 //  event; the translation unit's -ffp-contract=off is there for the DISTANCES, whose roundings decide
                                                 //  ties and the strict threshold)
#else
                if (e == 0) firstF = F; else local += (F - Fp) * Hp;
#endif
                // (unsigned: a signed `% 8` is five instructions; with one count word the category is below 8 -- checked where the
                // environments were built, foreign ones stored as 0)
                const unsigned uct = (unsigned)ct;
                const int sh = (int)((NW == 1 ? uct : (uct % FPW)) * FB), sh4 = (int)((uct & 15u) * 4u);
                int cntA_, cntB_;  // counts of category ct before the update
                if constexpr (LCNT) {
                    // (LDS serves a wavefront's requests in order: the lane's next event sees the incremented byte)
                    const unsigned sb_ = uct + (unsigned)CMAX;
                    unsigned char* pa_ = lcl + ((uct >> 3) << 9) + (uct & 7u);
                    unsigned char* pb_ = lcl + ((sb_ >> 3) << 9) + (sb_ & 7u);
                    cntA_ = *pa_;
                    cntB_ = *pb_;
                    *(takeA ? pa_ : pb_) = (unsigned char)((takeA ? cntA_ : cntB_) + 1);
                } else if constexpr (KSM && NW > 1) {
                    // every category is looked at after every event: the event goes straight into the count words
                    cntA_ = cntB_ = 0;
                    const uint64_t inc8 = 1ull << sh;
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
                        const bool hit = ((uct / FPW) == (unsigned)k);
                        exA[k] += (hit && takeA) ? inc8 : 0ull;
                        exB[k] += (hit && !takeA) ? inc8 : 0ull;
                    }
                } else if constexpr (NW == 1) {
                    // one count word per side (<= 8 slots): the event is added to the word itself -- no chunk-local fields, no second
                    // shift-and-mask pair per side
                    cntA_ = (int)((exA[0] >> sh) & 0xFFull);
                    cntB_ = (int)((exB[0] >> sh) & 0xFFull);
                    const uint64_t inc8 = 1ull << sh;
                    exA[0] += takeA ? inc8 : 0ull;
                    exB[0] += takeA ? 0ull : inc8;
                } else {
                uint64_t wA = exA[0], wB = exB[0];
#pragma unroll
                for (int k = 1; k < NW; ++k) {
                    const bool hit = ((uct / FPW) == (unsigned)k);
                    wA = hit ? exA[k] : wA;
                    wB = hit ? exB[k] : wB;
                }
                H4 qA = dA[0], qB = dB[0];
                if constexpr (NH == 2) { qA = (uct & 16u) ? dA[1] : qA; qB = (uct & 16u) ? dB[1] : qB; }
                cntA_ = (int)((wA >> sh) & 0xFFull) + (int)((qA >> sh4) & (H4)15);
                cntB_ = (int)((wB >> sh) & 0xFFull) + (int)((qB >> sh4) & (H4)15);
                const H4 inc4 = (H4)1 << sh4;
                if constexpr (NH == 2) {
                    const bool hi = (ct & 16) != 0;
                    dA[0] += (takeA && !hi) ? inc4 : (H4)0;
                    dA[1] += (takeA && hi) ? inc4 : (H4)0;
                    dB[0] += (!takeA && !hi) ? inc4 : (H4)0;
                    dB[1] += (!takeA && hi) ? inc4 : (H4)0;
                } else {
                    dA[0] += takeA ? inc4 : (H4)0;
                    dB[0] += takeA ? (H4)0 : inc4;
                }
                }
                if constexpr (KSM) {
                    const uint32_t Na = (uint32_t)(1 + i), Nb = (uint32_t)(1 + j);  // (the list positions are the totals)
                    uint32_t best = 0u;
#pragma unroll
                    for (int k = 0; k < NW; ++k) {
#pragma unroll
                        for (int f = 0; f < FPW; ++f) {
                            if (FPW * k + f < CMAX) {
                                const uint32_t ca = (uint32_t)(exA[k] >> (f * FB)) & 0xFFu, cb = (uint32_t)(exB[k] >> (f * FB)) & 0xFFu;
                                const uint32_t x = ca * Nb, y = cb * Na;  // (< 2^16 each)
                                best = max(best, x > y ? x - y : y - x);
                            }
                        }
                    }
                    const double ia = t_rsqrt[Na], ib = t_rsqrt[Nb];
                    Hp = (double)best * ((ia * ia) * (ib * ib));
                    Fp = F;
                } else {
                const int mine_ = takeA ? cntA_ : cntB_, other = takeA ? cntB_ : cntA_;
                if constexpr (WGT) {
                    const double wv_ = w_s[uct & 31u];
                    D = fma(wv_ * (t_sqrt[mine_ + 1] - t_sqrt[mine_]), t_sqrt[other], D);
                    na += takeA ? wv_ : 0.0;
                    nb += takeA ? 0.0 : wv_;
                    const double rr = rsqrt_w(takeA ? na : nb);
                    ra = takeA ? rr : ra;
                    rb = takeA ? rb : rr;
                } else {
                    D = fma(t_sqrt[mine_ + 1] - t_sqrt[mine_], t_sqrt[other], D);
                    ra = t_rsqrt[1 + i];
                    rb = t_rsqrt[1 + j];
                }
                double h2 = fma(-(ra * rb), D, 1.0);
                if (h2 < kExactH2Below) {  // literal difference-of-roots form where the cancellation form loses accuracy (k_sweep::exact_h2)
                    double acc2 = 0.0;
                    if constexpr (LCNT) {
#pragma unroll 1
                        for (int c = 0; c < CMAX; ++c) {
                            const int ca = lcl[((c >> 3) << 9) + (c & 7)], cb = lcl[(((c + CMAX) >> 3) << 9) + ((c + CMAX) & 7)];
                            const double dd = t_sqrt[ca] * ra - t_sqrt[cb] * rb;
                            acc2 = WGT ? fma(w_s[c & 31] * dd, dd, acc2) : fma(dd, dd, acc2);
                        }
                    } else if constexpr (CMAX <= LCHD_TEAM_EXACT_UNROLL_MAX) {
#pragma unroll
                        for (int k = 0; k < NW; ++k) {
#pragma unroll
                            for (int f = 0; f < FPW; ++f) {
                                const int c = FPW * k + f;
                                if (c < CMAX) {
                                    const int ca = field(exA, c) + (NW == 1 ? 0 : (int)((dA[0] >> (c * 4)) & (H4)15));
                                    const int cb = field(exB, c) + (NW == 1 ? 0 : (int)((dB[0] >> (c * 4)) & (H4)15));
                                    const double dd = t_sqrt[ca] * ra - t_sqrt[cb] * rb;
                                    acc2 = WGT ? fma(w_s[c & 31] * dd, dd, acc2) : fma(dd, dd, acc2);
                                }
                                if ((f & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    } else {  // (rare path, many slots: a rolled loop -- no unrolled copy of 28 look-up pairs competing for registers)
#pragma unroll 1
                        for (int c = 0; c < CMAX; ++c) {
                            uint64_t wA = exA[0], wB = exB[0];
#pragma unroll
                            for (int k = 1; k < NW; ++k) { wA = (c / FPW == k) ? exA[k] : wA; wB = (c / FPW == k) ? exB[k] : wB; }
                            const H4 qA = (c & 16) ? dA[NH - 1] : dA[0], qB = (c & 16) ? dB[NH - 1] : dB[0];
                            const int ca = (int)((wA >> ((c % FPW) * FB)) & 0xFFull) + (int)((qA >> ((c & 15) * 4)) & (H4)15);
                            const int cb = (int)((wB >> ((c % FPW) * FB)) & 0xFFull) + (int)((qB >> ((c & 15) * 4)) & (H4)15);
                            const double dd = t_sqrt[ca] * ra - t_sqrt[cb] * rb;
                            acc2 = WGT ? fma(w_s[c & 31] * dd, dd, acc2) : fma(dd, dd, acc2);
                        }
                    }
                    h2 = 0.5 * acc2;
                }
                Hp = sqrt_unit(h2);
                Fp = F;
                }
            }
        }
        // stitch the lane chunks of a team, add the last interval to +inf, reduce over the team
        double prevF = wave_shr1_f64(Fp), prevH = wave_shr1_f64(Hp);
        if (tl == 0) { prevF = F0; prevH = H0; }
        if (d0 < d1) local += (firstF - prevF) * prevH;
        // the team lane that holds the last event (its chunk is not empty and ends at T; lane 0 if there is no event at all)
        const bool is_last = T > 0 ? (d0 < d1 && d1 == T) : (tl == 0);
        if (is_last) local += (T > 0) ? (Finf0 - Fp) * Hp : (Finf0 - F0) * H0;
        return team_sum_f64<TL>(local);
    }
};

}  // namespace lchd

// lchd_sweep_team.hip -- K2 for the pairs that fit ONE tile, several per wavefront: k_sweep_duo<CMAX, TL, TILE, WGT, KSM> stages the
// two environments of every pair of a wavefront and runs the team tile (lchd_team_tile.h) -- LoCoHD::stat_dist_integral,
// /root/reference/src/locohd.rs:61-226, for Hellinger-2 (unit or weighted categories) and Kolmogorov-Smirnov.
#include "lchd_sweep_common.h"

namespace lchd {
// ------------------------------------------------------------------------------------------------
// K2 for small environments: TWO anchor pairs per wavefront, 32 lanes each.
//
// With environments of ~70-100 points per side (coarse-grained typing, the reference's main use) a pair has ~150 merged
// events: one wavefront per pair spends most of its instructions on the per-tile prologue (staging, merge path, scan, state
// reload, reduction), all of them executed for 64 lanes of which a third idle.  Here every wave-wide instruction serves two
// pairs.  A pair qualifies if it has at most kDuoTile merged events (exactly one tile, no carries between tiles); the
// configuration must be Hellinger-2 with unit category weights, CDF-keyed environments, at most 16 category slots.  The
// host launches this kernel AND k_sweep; k_pair_meta counts the qualifying pairs (DeviceStatus::n_small): when they are
// the majority this kernel sweeps them and k_sweep only the rest, otherwise this kernel returns at once.
// ------------------------------------------------------------------------------------------------
#ifndef LCHD_TEAM_BIG_WAVES
#define LCHD_TEAM_BIG_WAVES 3   // waves per SIMD k_sweep_duo is compiled for with more than 16 category slots
#endif
// (the name is historic: round 1 swept TWO pairs per wavefront; with TL = 16 a wavefront sweeps FOUR -- the per-tile prologue, which
// is two thirds of this kernel's instructions at ~150 events per pair, is shared by twice as many pairs, the event loop costs the
// same per pair: C3 459 -> see DESIGN section 4)
// TILE_ = 240: pairs of at most 240 merged events (small_rule 0); TILE_ = 480 (TL = 32): pairs whose environments both have at most
// 255 points and that have at most 480 merged events (small_rule 2) -- the 8-bit-count k_sweep's pairs, two per wavefront (C2a: ~343
// events per pair)
// WGT: category weights other than 1 (pmf.rs:47-63 adds weight[c] per point): H^2 = 1 - sum_c w_c sqrt(a_c b_c) / sqrt(W_a W_b) with the
// weighted totals W = sum_c w_c count_c -- the same integer count fields and tables, one multiplier per category from LDS, two
// running totals and one reciprocal square root per event instead of the two table look-ups of the unit-weight form.
// KSM: the Kolmogorov-Smirnov distance max_c |a_c / N_a - b_c / N_b| (statistical_distances.rs:12-21) with unit weights instead of
// Hellinger-2: every event needs all categories, but as INTEGERS -- max_c |a_c N_b - b_c N_a| over the 8-bit count fields (two 24-bit
// multiplies, one v_sad_u32, one max per category), scaled once by 1 / (N_a N_b) from the reciprocal-root table; no square root.
#ifndef LCHD_STAGE_PAIRS
#define LCHD_STAGE_PAIRS 1   // the team sweeps stage two buffer entries per lane and round (0: one)
#endif
#ifndef LCHD_WGT_W3
#define LCHD_WGT_W3 0       // 1: ... are compiled for 3 waves per SIMD (170 registers: no spills)
#endif
template <int CMAX, int TL = LCHD_DUO_TL, int TILE_ = kDuoTile, bool WGT = false, bool KSM = false, bool PRE = false>
__global__ __launch_bounds__(64 * kSweepWaves, ((CMAX <= 16 && !(WGT && CMAX > 8 && (LCHD_WGT_LDSCNT || LCHD_WGT_W3))) ? 4 : LCHD_TEAM_BIG_WAVES)) void k_sweep_duo(SweepArgs args) {
    // (the tile itself -- merge path, chunk histogram, count scans, event loop, stitching -- is lchd_team_tile.h)
    using TT = TeamTile<CMAX, TL, TILE_, WGT, KSM, PRE>;
    constexpr int TEAMS = TT::TEAMS, EPL = TT::EPL, TILE = TT::TILE, WPB = kSweepWaves, NT = TT::NT, LW = TT::LW;
    constexpr bool LCNT = TT::LCNT;
    constexpr int RULE = TILE_ == kDuoTile ? 0 : 2;
    __shared__ double t_sqrt[NT], t_rsqrt[NT];
    // one buffer per team: list A's points, then list B's (at most TILE together; + the spare entries the head re-reads may touch)
    __shared__ uint64_t s_[WPB][TEAMS][TILE + 2];
    __shared__ uint8_t c_[WPB][TEAMS][TILE + 8];
    __shared__ uint64_t lc_[LCNT ? WPB : 1][LCNT ? LW * 64 : 1];  // (TeamTile::LCNT: per-lane count rows of the event loop)
    __shared__ double w_s[WGT ? 32 : 1];
    if (!args.forced && rule_in_force(args) != RULE) return;  // another rule's pairs are the majority, or none's: k_sweep sweeps everything
    const int tid = threadIdx.x, lane = tid & 63, tl = lane & (TL - 1), team = lane / TL;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const DevConfig* __restrict__ cfgp = args.cfg;
    const double Finf0 = cfgp->wf_finf[0];
    for (int k = tid; k < NT; k += 64 * WPB) {
        t_sqrt[k] = args.sqrt_tab[k];
        t_rsqrt[k] = args.rsqrt_tab[k];
    }
    if constexpr (WGT) {
        if (tid < 32) w_s[tid] = tid < cfgp->n_categories ? cfgp->cat_w[tid] : 0.0;
    }
    __syncthreads();
    uint64_t* sA = s_[wv][team];
    uint8_t* cA = c_[wv][team];
    unsigned char* lcl = reinterpret_cast<unsigned char*>(lc_[LCNT ? wv : 0]) + lane * 8;  // this lane's eight bytes of word 0

    const int64_t pstride = (int64_t)gridDim.x * WPB * TEAMS;
    for (int64_t pb = ((int64_t)blockIdx.x * WPB + wv) * TEAMS; pb < args.n_pairs; pb += pstride) {
        const int64_t p = pb + team;
        const bool live = p < args.n_pairs;
        const int4 m = args.meta[live ? p : pb];
        const bool usable = live && (m.z & 0xFFFFFF) > 0 && (m.w & 0xFFFFFF) > 0;
        const bool mine = !usable || pair_is_small(RULE, m.z & 0xFFFFFF, m.w & 0xFFFFFF);  // larger pairs belong to k_sweep
        const bool valid = usable && mine;
        const int mA = valid ? (m.z & 0xFFFFFF) - 1 : 0, mB = valid ? (m.w & 0xFFFFFF) - 1 : 0, T = mA + mB;  // non-anchor events
        const int c0a = (m.z >> 24) & 255, c0b = (m.w >> 24) & 255;
        // (a dictionary's key sets: the set of this pair's weight function -- k_pair_meta has checked the index of every usable pair)
        int64_t ksetA = 0, ksetB = 0;
        if (args.wf_index) {  // (wave-uniform: configurations with one weight function never multiply)
            const int64_t kset = valid ? args.wf_index[p] : 0;
            ksetA = kset * args.env_a.set_stride;
            ksetB = kset * args.env_b.set_stride;
        }
        // (slot x stride as ONE 32 x 32 -> 64-bit multiply: slots and strides are below 2^31)
        const uint64_t offA = (uint64_t)(uint32_t)m.x * (uint32_t)args.env_a.stride, offB = (uint64_t)(uint32_t)m.y * (uint32_t)args.env_b.stride;
        const uint64_t* __restrict__ kA = args.env_a.key + offA + ksetA;
        const uint64_t* __restrict__ kB = args.env_b.key + offB + ksetB;
        const uint8_t* __restrict__ tA = args.env_a.cat + offA;
        const uint8_t* __restrict__ tB = args.env_b.cat + offB;
        const double F0 = valid ? u2d(kA[0]) : 0.0;            // F(0): both anchors sit at distance 0
#if LCHD_STAGE_PAIRS
        // list B starts at an EVEN entry of the buffer (one unused entry behind an odd list A): the staging below moves two entries per
        // lane and round -- one 16-byte key load, one 16-byte LDS write -- and no pair of entries straddles the two lists
        const int mAe = (mA + 1) & ~1, Tb = mAe + mB;  // <= TILE + 1: the buffers hold TILE + 2 entries
        uint64_t* sB = sA + mAe;
        uint8_t* cB = cA + mAe;
#else
        uint64_t* sB = sA + mA;
        uint8_t* cB = cA + mA;
#endif

        // lane tl of a team owns merged events [d0, d1) of its pair
        const int epl = (T + TL - 1) / TL;  // <= EPL
        int epl_w = __builtin_amdgcn_readlane(epl, 0);  // wave-uniform trip count: the longest of the teams' chunks
#pragma unroll
        for (int k = 1; k < TEAMS; ++k) epl_w = max(epl_w, __builtin_amdgcn_readlane(epl, k * TL));

        wave_sync_lds();  // the previous pairs' tiles are fully consumed
#if LCHD_STAGE_PAIRS
        {   // stage [A's points | pad | B's points]: entries 2 q and 2 q + 1 of the buffer by lane q % TL in round q / TL; all loads before
            // the first LDS write.  A pair's second entry may lie one past its list's last point (still inside the environment's slot or,
            // for the last slot, the workspace's slack): it lands in the pad entry or behind the buffer's used part and is never read.
            constexpr int EPL2 = (EPL + 1) / 2;
            static_assert(2 * TL * EPL2 >= TILE_ + 1, "the rounds cover the buffer's used part (pad entry included)");
            typedef unsigned long long __attribute__((ext_vector_type(2), aligned(8))) key2_t;
            const int epl2 = (Tb + 2 * TL - 1) / (2 * TL);
            int epl2_w = __builtin_amdgcn_readlane(epl2, 0);
#pragma unroll
            for (int k = 1; k < TEAMS; ++k) epl2_w = max(epl2_w, __builtin_amdgcn_readlane(epl2, k * TL));
            key2_t rk[EPL2];
            uint32_t rc[EPL2];
            const uint64_t* kBs = kB - mAe;
            const uint8_t* tBs = tB - mAe;
#pragma unroll
            for (int u = 0; u < EPL2; ++u) { rk[u] = key2_t{0ull, 0ull}; rc[u] = 0u; }
            if (valid) {
#pragma unroll
                for (int u = 0; u < EPL2; ++u) {
                    if (u < epl2_w) {  // (wave-uniform: rounds no team of this wavefront needs are skipped)
                        const int t0 = 2 * (tl + TL * u);
                        const int tt = t0 < Tb ? t0 : 0;  // (beyond the used part: re-read the row's first pair, nothing is written)
                        const bool isA = tt < mAe;
                        const uint64_t* src = (isA ? kA : kBs) + 1 + tt;
                        const uint8_t* csrc = (isA ? tA : tBs) + 1 + tt;
                        rk[u] = *reinterpret_cast<const key2_t*>(src);
                        rc[u] = (uint32_t)csrc[0] | ((uint32_t)csrc[1] << 8);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EPL2; ++u) {
                if (u < epl2_w) {
                    const int t0 = 2 * (tl + TL * u);
                    if (t0 < Tb) {
                        *reinterpret_cast<ulonglong2*>(sA + t0) = ulonglong2{rk[u].x, rk[u].y};
                        *reinterpret_cast<uint16_t*>(cA + t0) = (uint16_t)rc[u];
                    }
                }
            }
        }
#else
        {   // stage [A's points | B's points]: entry t of the buffer is A[1 + t] or B[1 + t - mA]; all loads before the first LDS write.
            // One predicate for the whole team (the pair is swept here), none per entry: an entry beyond T re-reads the pair's last
            // point (index clamped: inside the row) and lands in the buffer's unused tail (t < TILE).
            uint64_t rk[EPL];
            uint8_t rc[EPL];
            const uint64_t* kBs = kB - mA;
            const uint8_t* tBs = tB - mA;
#pragma unroll
            for (int u = 0; u < EPL; ++u) { rk[u] = 0ull; rc[u] = 0; }
            if (valid) {
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    if (u < epl_w) {  // (wave-uniform: rounds no team of this wavefront needs are skipped)
                        const int t = min(tl + TL * u, T - 1);  // (T = 0: entry 0 of list A's row, the anchor)
                        const bool isA = t < mA;
                        rk[u] = (isA ? kA : kBs)[1 + t];
                        rc[u] = (isA ? tA : tBs)[1 + t];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                if (u < epl_w) {
                    const int t = tl + TL * u;
                    sA[t] = rk[u];
                    cA[t] = rc[u];
                }
            }
        }
#endif
        wave_sync_lds();

        // (PRE: the prefix-count rows of the two environments; an unusable pair's records may name slots that do not exist: row 0 of slot 0)
        const uint64_t* preA = PRE ? args.env_a.pre + (valid ? offA / kPreStep * (uint64_t)TT::NW : 0ull) : nullptr;  // (slot strides are multiples of kPreStep)
        const uint64_t* preB = PRE ? args.env_b.pre + (valid ? offB / kPreStep * (uint64_t)TT::NW : 0ull) : nullptr;
        const double acc = TT::run(sA, cA, sB, cB, mA, mB, T, epl, epl_w, c0a, c0b, F0, Finf0, t_sqrt, t_rsqrt, w_s, lcl, tl, preA, preB);
        if (tl == TL - 1 && live && mine) args.out[p] = valid ? acc : nan("");  // (categories were checked when the environments were built)
    }
}

template <int CM, int TM, bool PRE>
static void launch_team_c(hipStream_t s, bool tile240, unsigned grid, const SweepArgs& a) {
    constexpr int NTH = 64 * kSweepWaves;
    constexpr bool WGT = TM == 1, KSM = TM == 2;
    if (tile240) k_sweep_duo<CM, LCHD_DUO_TL, kDuoTile, WGT, KSM, PRE><<<grid, NTH, 0, s>>>(a);
    else k_sweep_duo<CM, 32, kTeam8Tile, WGT, KSM, PRE><<<grid, NTH, 0, s>>>(a);
}
template <int TM>
static void launch_team_t(hipStream_t s, int cmax, bool tile240, unsigned grid, const SweepArgs& a) {
    // both stores carry prefix-count rows of the width this slot count reads (k_env_group wrote them): the PRE instantiations
    if constexpr (TM != 1) {
        const int nw = team_pre_words(cmax);
        if (cmax <= 16 && a.env_a.pre && a.env_b.pre && a.env_a.pre_words == nw && a.env_b.pre_words == nw) {
            if (cmax <= 8) launch_team_c<8, TM, true>(s, tile240, grid, a);
            else if (cmax <= 12) launch_team_c<12, TM, true>(s, tile240, grid, a);
            else launch_team_c<16, TM, true>(s, tile240, grid, a);
            return;
        }
    }
    if (cmax <= 8) launch_team_c<8, TM, false>(s, tile240, grid, a);
    else if (cmax <= 12) launch_team_c<12, TM, false>(s, tile240, grid, a);
    else if (cmax <= 16 || TM != 0) launch_team_c<16, TM, false>(s, tile240, grid, a);  // (weights / Kolmogorov-Smirnov: at most 16 slots, checked by launch_sweep)
    else if constexpr (TM == 0) {
        if (cmax <= 20) launch_team_c<20, 0, false>(s, tile240, grid, a);
        else if (cmax <= 24) launch_team_c<24, 0, false>(s, tile240, grid, a);
        else if (cmax <= 28) launch_team_c<28, 0, false>(s, tile240, grid, a);
        else launch_team_c<32, 0, false>(s, tile240, grid, a);
    }
}
void launch_team(hipStream_t s, int cmax, int tm, bool tile240, unsigned grid, const SweepArgs& a) {
    if (tm == 2) launch_team_t<2>(s, cmax, tile240, grid, a);
    else if (tm == 1) launch_team_t<1>(s, cmax, tile240, grid, a);
    else launch_team_t<0>(s, cmax, tile240, grid, a);
}

}  // namespace lchd

// lchd_kernels.hip -- gfx950 (MI355X, CDNA4): the launch logic of the LoCoHD scoring path and its small kernels.
//
// Pipeline for one from_primitives call (reference: /root/reference/src/locohd.rs:479-567), one kernel family per translation unit:
//
//   K0  cell lists + anchor de-duplication   lchd_prologue.hip      (KdTree::build_by_ordered_float, :504-510)
//   K1  environment build                    lchd_env_group.hip     several environments of <= 512 points per wavefront (the default)
//                                            lchd_env_cells.hip     one environment per workgroup, capacities 1024 .. 65535 and beyond
//                                            lchd_env_rows.hip      dense rows (from_coords / from_dmxs), lchd_dense_fused.hip: sort + sweep in one
//                                            (env_from_idx :514-542, utils::sort_together utils.rs:25-39)
//   K2' pair records                         k_pair_meta (here): one 16-byte record per pair {slot A, slot B, n_A | cat, n_B | cat}
//   K2  sweep                                lchd_sweep.hip         one pair per wavefront (every distance, every size)
//                                            lchd_sweep_team.hip    the pairs that fit one tile, four or two per wavefront (lchd_team_tile.h)
//                                            lchd_sweep_wide.hip    33 .. 65534 categories, environments beyond 65535 points
//                                            lchd_sweep_inc.hip     Kullback-Leibler / Renyi in O(1) per event
//                                            (stat_dist_integral :61-226, pmf.rs, statistical_distances.rs, cdfs.rs)
//
// Here: launch_sweep (which of the sweep kernels take a pass -- from the configuration, the call's size and the previous pass's
// pair statistics), the record pass, and the kernels around the path: multi-GPU sharding of a pair list, the second pass over
// overflowed environments, weight-function key sets, tables, trajectory frames (SoA unpack, centroids of primitive atoms).
// All arithmetic is f64 like the reference; no MFMA (there is no contraction in this path).
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "lchd_sweep_common.h"

namespace lchd {

// One record per anchor pair for the sweep kernels: {environment slot A, slot B, n_A | category of anchor A << 24,
// n_B | category of anchor B << 24}; n = 0 marks a pair the sweep must answer with NaN (anchor index out of range -- already
// flagged by k_mark_anchors -- or an environment that overflowed / is empty -- flagged by K1).
__global__ void k_pair_meta(SweepArgs args) {
    int n_duo = 0, n_c8 = 0, biggest = 0;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < args.n_pairs; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t ea = p, eb = p;
        bool ok = true;
        if (args.anchors) {
            const int64_t ia_ = args.anchors[2 * p], ib_ = args.anchors[2 * p + 1];
            ok = !(ia_ < 0 || ib_ < 0 || ia_ >= args.n_slot_a || ib_ >= args.n_slot_b);
            if (ok) { ea = args.slot_a[ia_]; eb = args.slot_b ? args.slot_b[ib_] : p; }
        }
        if (args.wf_index && args.env_a.cdf_keys > 1) {
            // key sets: the pair's weight-function index picks the set its sweep reads -- an index outside the dictionary is reported
            // here and the pair marked unusable (the sweep answers NaN)
            const int wfi = args.wf_index[p];
            if (wfi < 0 || wfi >= args.env_a.cdf_keys) { ok = false; atomicOr(&args.st->flags, ST_BAD_WF); }
        }
        int nA = 0, nB = 0, c0a = 0, c0b = 0;
        if (ok) {
            nA = args.env_a.len[ea];
            nB = args.env_b.len[eb];
            if (nA > 0 && nB > 0) {
                if (args.env_a.cat16) {
                    c0a = reinterpret_cast<const uint16_t*>(args.env_a.cat)[ea * args.env_a.stride];
                    c0b = reinterpret_cast<const uint16_t*>(args.env_b.cat)[eb * args.env_b.stride];
                } else if (args.env_a.cat0 && args.env_b.cat0) {  // (the grouped environment kernel's header array: L2-resident, unlike the store)
                    c0a = args.env_a.cat0[ea];
                    c0b = args.env_b.cat0[eb];
                } else {
                    c0a = args.env_a.cat[ea * args.env_a.stride];
                    c0b = args.env_b.cat[eb * args.env_b.stride];
                }
            } else {
                nA = nB = 0;
            }
        }
        // (16-bit categories, k_sweep_wide only: the environment length -- at most 65535 -- in the low half, the category above it)
        if (args.env_a.cat16) args.meta[p] = make_int4((int)ea, (int)eb, nA | (c0a << 16), nB | (c0b << 16));
        else args.meta[p] = make_int4((int)ea, (int)eb, nA | (c0a << 24), nB | (c0b << 24));
        biggest = max(biggest, max(nA, nB));
        // pairs k_sweep_duo takes: everything that fits its tile, and the unusable ones (it writes their NaN); pairs the
        // 8-bit-count sweep takes: both environments of at most 255 points.  Both are counted whichever rule this pass uses:
        // the host picks the next pass's kernels from them.
        n_duo += (nA + nB - 2 <= kDuoTileFwd) ? 1 : 0;
        n_c8 += pair_is_small(args.c8_rule, nA, nB) ? 1 : 0;
        if (args.left_listing) {  // (uniform) the pairs the team kernel of this pass leaves to the INDIRECT companion: the test of its scan
            const bool left = nA > 0 && !pair_is_small(args.small_rule, nA, nB);
            const unsigned long long lm = __ballot(left);
            if (lm) {  // (rare: a few hundred pairs of a million)
                const int lane = threadIdx.x & 63, leader = __ffsll((long long)lm) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(args.left_count, (uint32_t)__popcll(lm));
                base = (uint32_t)__shfl((int)base, leader);
                if (left) args.left_list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u))] = (uint32_t)p;
            }
        }
    }
    if (args.left_zero && blockIdx.x == 0 && threadIdx.x == 0) *args.left_zero = 0u;  // the next pass's counter slot
    // pairs that fit one 32-lane tile: if they are the majority, k_sweep_duo sweeps them and k_sweep only the rest.  One
    // partial count per workgroup; the workgroup that finishes LAST folds them, publishes what the host wants to know into
    // the host-mapped mirror and resets the device status for the next pass -- no separate summing kernel, no memset before
    // a pass, no copy after it.
    __shared__ int big_s[4];
    __shared__ bool last_s;
    unsigned long long n_both = (unsigned long long)n_duo | ((unsigned long long)n_c8 << 32);  // (a launch has < 2^32 pairs)
    for (int m = 32; m > 0; m >>= 1) { n_both += shfl_u64(n_both, (threadIdx.x & 63) ^ m); biggest = max(biggest, __shfl_xor(biggest, m)); }
    __shared__ unsigned long long both_s[4];
    if ((threadIdx.x & 63) == 0) { both_s[threadIdx.x >> 6] = n_both; big_s[threadIdx.x >> 6] = biggest; }
    __syncthreads();
    if (threadIdx.x == 0)  // (largest environment of the pass: the host lets its capacity hint decay with it)
        last_s = last_workgroup_done(args.done, both_s[0] + both_s[1] + both_s[2] + both_s[3],
                                     (uint32_t)max(max(big_s[0], big_s[1]), max(big_s[2], big_s[3])));
    __syncthreads();
    if (!last_s || threadIdx.x >= 64) return;
    unsigned long long v;
    uint32_t mx;
    collect_done(args.done, threadIdx.x, v, mx);
    for (int m = 32; m > 0; m >>= 1) { v += shfl_u64(v, threadIdx.x ^ m); mx = max(mx, (uint32_t)__shfl_xor((int)mx, m)); }
    if (threadIdx.x == 0) {
        const unsigned long long duo = v & 0xFFFFFFFFull, c8 = v >> 32;
        args.hst->n_duo = duo;
        args.hst->n_c8 = c8;
        args.st->n_c8 = c8;
        publish_status(args, args.small_rule ? c8 : duo, mx);  // (small_rule 1 or 2 == c8_rule whenever it is not 0)
    }
}

int launch_sweep(hipStream_t s, const Tuning& t, int n_categories, bool hellinger2, bool unit_weights, bool wf_pow, int sweep_hint,
                 const SweepArgs& a_in) {
    if (a_in.n_pairs <= 0) return 0;
    SweepArgs a = a_in;
    a.duo_enabled = 0;
    a.forced = 0;
    a.gen_tab = unit_weights ? 1 : 0;  // (MODE_GEN, Hellinger with a general exponent: the configuration's power tables apply)
    if (t.force_generic) hellinger2 = false;  // test hook
    if (a.n_pairs <= kInlineMetaPairs && !t.no_inline_meta && hellinger2 && unit_weights && n_categories <= 32 && !t.force_wide &&
        a.env_a.cdf_keys && a.env_b.cdf_keys && a.env_a.stride <= kSqrtTab && a.env_b.stride <= kSqrtTab && !t.force_bigenv) {
        // small call, default configuration: one launch (records worked out by the sweep itself, one pair per wavefront)
        const int cm = std::max(n_categories, t.force_cmax);
        const unsigned g = (unsigned)((a.n_pairs + kSweepWaves - 1) / kSweepWaves);
        launch_sweep_inline(s, cm, g, a);
        return 0;
    }
    const bool wide = n_categories > 32 || t.force_wide || a.env_a.stride > 65535 || a.env_b.stride > 65535;  // (long environments: the 64-bit-count form of the wide sweep)
    const int64_t blocks = (a.n_pairs + kSweepWaves - 1) / kSweepWaves;
    // grid-stride: LDS tables are built once per block.  8192 workgroups = 8 rounds of the 1024 that are resident at a time: finer
    // than that the table loads show, coarser the last round's imbalance does (measured on C2a: 4096 +2.8 %, 16384 +0.5 %)
    const int64_t gcap = 8192;
    // ... the team sweeps (shorter iterations, a smaller table load per workgroup): 16384 (C2a 1.3648 -> 1.358 ms, C3 0.7835 -> 0.7769; 32768: no further gain)
    // ... but a launch of at most 32768 team workgroups' worth of pairs (a rank's 125 000-pair share of C2a under strong scaling) is cut into
    // 4096: every workgroup then amortises its LDS table load over ~4 rounds (measured with --emulate-world 8, per-rank step median:
    // 0.214 -> 0.200 ms; the same cap on the 10^6-pair launches costs 1-5 %: C4 sweep 2.69 -> 2.83 ms)
    const int64_t tcap_big = 16384, tcap_small = 4096, tcap_switch = 32768;
    const unsigned grid = (unsigned)(blocks < gcap ? blocks : gcap);
    const int cmax = std::max(n_categories, t.force_cmax);  // (force_cmax: test hook)
    const bool small = a.env_a.stride <= kSqrtTab && a.env_b.stride <= kSqrtTab && !t.force_bigenv;  // every count fits the LDS tables
    const int fmode = (a.env_a.cdf_keys && a.env_b.cdf_keys) ? F_KEY : (wf_pow ? F_ANY : F_FAST);
    // Two kernels for "small" pairs exist for the default configuration (Hellinger-2, unit weights, CDF-keyed environments):
    // k_sweep_duo (two pairs of <= 240 merged events per wavefront, <= 16 category slots) and the 8-bit-count k_sweep (both
    // environments <= 255 points, more than 16 slots); the INDIRECT 16-bit k_sweep takes what they leave over.
    // ... and, up to 16 slots, for category weights other than 1 (the WGT instantiations of the team kernels; the one-pair-per-wavefront
    // 8-bit-count sweep has no weighted form, so both team rules must be available)
    const bool weighted_team = !unit_weights && cmax <= 16 && !t.no_duo && !t.no_c8_team && !t.no_count8;
    // ... and for the Kolmogorov-Smirnov distance with unit weights (SweepArgs::sd_fast == 3: the KSM instantiations)
    const bool ks_team = !hellinger2 && a.sd_fast == 3 && unit_weights && cmax <= 16 && !t.no_duo && !t.no_c8_team && !t.no_count8 && !t.force_generic;
    const bool fast_cfg = !wide && ((hellinger2 && (unit_weights || weighted_team)) || ks_team) && small && fmode == F_KEY;  // (F_KEY with a weight-function dictionary: the store holds one key set per function)
    // sweep_hint (what k_pair_meta counted in the previous pass of this configuration): 0 = nothing known, else
    // 4 | (pairs of <= 240 events were the majority ? 1 : 0) | (pairs with both environments <= 255 points were ? 2 : 0).
    // Up to 16 slots k_sweep_duo is the first choice and the 8-bit-count sweep the second (C2a: environments of ~170 points,
    // pairs of ~340 events -- too long for a 32-lane tile, but their counts fit 8 bits: 2 count words per side instead of 3);
    // above 16 slots only the 8-bit-count sweep exists.
    const int hint_bits = t.no_sweep_hint ? 0 : sweep_hint;
    const bool known = (hint_bits & 4) != 0, duo_major = (hint_bits & 1) != 0, c8_major = (hint_bits & 2) != 0;
    const bool c8_small_slots = fast_cfg && cmax <= 16 && !t.no_count8 && known && !(duo_major && !t.no_duo) && c8_major;
    const bool use_duo = fast_cfg && cmax <= 16 && !t.no_duo && !c8_small_slots;
    const bool use_c8 = fast_cfg && !t.no_count8 && (cmax > 16 || c8_small_slots);
    // up to 16 slots the 8-bit-count pairs are swept two per wavefront (rule 2: and at most 480 merged events)
    const bool team_ok = !t.no_c8_team && cmax <= 32;
    const bool c8_team = use_c8 && team_ok;
    a.c8_rule = team_ok ? 2 : 1;  // (what k_pair_meta counts as n_c8 -- whichever small-pair kernel this pass uses)
    a.small_rule = use_c8 ? a.c8_rule : 0;
    // no hint and up to 16 slots: k_sweep_duo's rule first, the two-pairs-per-wavefront 8-bit-count rule second
    a.second_rule = (!known && use_duo && fast_cfg && !t.no_count8 && !t.no_c8_team) ? 2 : 0;
    const int hint = !known ? 0 : ((use_c8 ? c8_major : duo_major) ? 1 : 2);
    // ... | 8 (EVERY pair of the previous pass had at most 240 events) | 16 (... both environments <= 255 points): the companion
    // launch for the larger pairs would find nothing to do and is left out; the host checks the counts of THIS pass afterwards
    // and repeats it with the full launch set if a larger pair turned up after all (the returned bit 2 says the launch was left out)
    const bool no_others = hint == 1 && (hint_bits & (use_c8 ? 16 : 8)) != 0;
    // the leftover list: only where the rule is known at launch and a companion will read it (otherwise the device decides the rule
    // from this very record pass and the companion scans the records)
    a.left_listing = (a.left_list && a.left_count && !wide && (use_duo || use_c8) && hint == 1 && !no_others) ? 1 : 0;
    const int info = (use_c8 ? 1 : 0) | (no_others ? 2 : 0) | (a.left_zero ? 4 : 0);
    {
        const int64_t nb = (a.n_pairs + 255) / 256;
        const int mgrid = (int)(nb < kMetaPartials ? nb : kMetaPartials);
        k_pair_meta<<<mgrid, 256, 0, s>>>(a);
    }
    if (wide) {
        const int fm = (a.env_a.cdf_keys && a.env_b.cdf_keys) ? F_KEY : F_ANY;
        launch_sweep_wide(s, !hellinger2 ? MODE_GEN : (unit_weights ? MODE_H2U : MODE_H2W), n_categories, a.n_pairs, fm, a);
        return info & 4;
    }
    if (use_duo || use_c8) {
        // Without a hint the small-pair kernel, its companion and the plain sweep are all launched and the number of small
        // pairs (k_pair_meta) decides on the device which of them do the work; with the hint of the previous pass only the
        // kernels that will work are launched.
        a.forced = hint != 0;
        if (hint != 2) {
            a.duo_enabled = 1;
            unsigned bgrid = grid < LCHD_COMPANION_GRID ? grid : LCHD_COMPANION_GRID;  // the listed (larger) pairs are a minority whenever this launch does anything
            if (a.left_listing) {  // one wavefront per listed pair, sized from what the previous pass left over (a grid-stride loop: any grid is correct)
                const int64_t want = (a.left_expected + a.left_expected / 4 + kSweepWaves - 1) / kSweepWaves + 8;
                bgrid = (unsigned)std::min<int64_t>(bgrid, std::max<int64_t>(want, 16));
            }
            // (team mode: 0 Hellinger-2 with unit weights, 1 with category weights, 2 Kolmogorov-Smirnov with unit weights)
            const int tm = ks_team ? 2 : (unit_weights ? 0 : 1);
            if (use_duo) {
                constexpr int kTeamPairs = (64 / LCHD_DUO_TL) * kSweepWaves;  // pairs per workgroup and round
                const int64_t dblocks = (a.n_pairs + kTeamPairs - 1) / kTeamPairs;
                const int64_t tcap = dblocks <= tcap_switch ? tcap_small : tcap_big;
                const unsigned dgrid = (unsigned)(dblocks < tcap ? dblocks : tcap);
                const int64_t tblocks = (a.n_pairs + 2 * kSweepWaves - 1) / (2 * kSweepWaves);
                const int64_t tcap2 = tblocks <= tcap_switch ? tcap_small : tcap_big;
                const unsigned tgrid = (unsigned)(tblocks < tcap2 ? tblocks : tcap2);
                // the four-pairs team kernel, the INDIRECT companion for the pairs its rule leaves over and -- a pass without a hint --
                // the second team rule's kernel
                launch_team(s, cmax, tm, true, dgrid, a);
                if (!no_others) launch_sweep_indirect(s, cmax, tm, bgrid, a);
                if (a.second_rule) launch_team(s, cmax, tm, false, tgrid, a);
            } else if (c8_team) {
                constexpr int kTeamPairs = 2 * kSweepWaves;
                const int64_t dblocks = (a.n_pairs + kTeamPairs - 1) / kTeamPairs;
                const int64_t tcap = dblocks <= tcap_switch ? tcap_small : tcap_big;
                const unsigned dgrid = (unsigned)(dblocks < tcap ? dblocks : tcap);
                launch_team(s, cmax, tm, false, dgrid, a);
                if (!no_others) launch_sweep_indirect(s, cmax, tm, bgrid, a);
            } else {
                launch_sweep_c8(s, cmax, grid, a);
                if (!no_others) launch_sweep_indirect(s, cmax, 0, bgrid, a);
            }
            if (hint == 1) return info;
        }
    }
    if (!hellinger2) {
        if ((a.sd_fast == 1 || a.sd_fast == 2) && unit_weights && small && fmode == F_KEY && !a.wf_index && cmax <= 32) launch_sweep_inc(s, a.sd_fast, cmax, a);
        else launch_sweep_plain(s, MODE_GEN, false, cmax, grid, fmode, a);
    } else {
        launch_sweep_plain(s, unit_weights ? MODE_H2U : MODE_H2W, small, cmax, grid, fmode, a);
    }
    return info & 5;
}

// Kernels that may be launched with more than 64 KB of dynamic LDS need the limit raised per DEVICE: lchd_ctx_create calls this
// with the context's device current (a process-wide "done once" flag would leave a second device without the attribute).
void init_device_kernels() {
    init_prologue_kernels();
    init_env_cells_kernels();
    init_env_rows_kernels();
    init_sweep_wide_kernels();
}

// sum over pairs of n_A + n_B (algorithmic-bytes accounting for bench.py; not part of the scoring path)
__global__ void k_env_points(SweepArgs args, unsigned long long* out) {
    unsigned long long local = 0;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < args.n_pairs; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t ea = p, eb = p;
        if (args.anchors) {
            const int64_t ia_ = args.anchors[2 * p], ib_ = args.anchors[2 * p + 1];
            if (ia_ < 0 || ib_ < 0 || ia_ >= args.n_slot_a || ib_ >= args.n_slot_b) continue;
            ea = args.slot_a[ia_];
            eb = args.slot_b ? args.slot_b[ib_] : p;
        }
        local += (unsigned long long)max(args.env_a.len[ea], 0) + (unsigned long long)max(args.env_b.len[eb], 0);
    }
    for (int m = 32; m > 0; m >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)local, m), hi = __shfl_xor((uint32_t)(local >> 32), m);
        local += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(out, local);
}
void launch_env_points(hipStream_t s, const SweepArgs& a, unsigned long long* out) {
    (void)hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
    if (a.n_pairs > 0) k_env_points<<<1024, 256, 0, s>>>(a, out);
}

}  // namespace lchd

namespace lchd {
// ------------------------------------------------------------------------------------------------
// Multi-GPU sharding of an anchor-pair list (one process per GPU, every rank holds the whole list and both structures).
// A rank that scores a contiguous slice of RANDOM pairs builds almost every environment of both structures itself; pairs
// binned by their side-A anchor make every rank build ~1/world of side A's environments.  The rule is a pure function of
// the list, so every rank computes the same partition without talking to the others:
//   bin(p)  = floor(a_p * kShardBins / n_atoms_a)                      (a_p = side-A anchor index, clamped into range)
//   rank(b) = min(world - 1, floor(#pairs in bins < b * world / P))    (the rank in which the bin's first pair falls)
// k_shard_plan: histogram of the bins (LDS-private per workgroup), the last workgroup turns it into rank(b) and the
// per-rank pair counts (also stored into host-mapped memory) and zeroes the histogram and the selection cursor again;
// k_shard_select: this rank's pairs, compacted (order = workgroup arrival, the original positions travel with them);
// k_unshard_scores: on the gathering rank, score k of rank r goes to its pair's original position.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int shard_bin(int64_t a, int64_t n_atoms) {
    a = a < 0 ? 0 : (a >= n_atoms ? n_atoms - 1 : a);
    return (int)((a * kShardBins) / n_atoms);
}
// The rule (loco_hd_amd/dist.py: shard_rule; lchd_capi.hip: shard_rule_host -- the same arithmetic in all three places):
//   key side   0: bins of the side-A anchor; if that partition is unbalanced (a rank would hold more than 1.25 P / world + 1
//              pairs: a list with one reference anchor against thousands, python_codes/kras_scan.py:46-52) 1: bins of the side-B
//              anchor; if that one is unbalanced too (or n_atoms_b is not given) 2: contiguous slices of the pair list
//   rank(bin) = min(world - 1, floor(#pairs in lower bins * world / P));   key side 2: rank(pair p) = floor(p * world / P)
__global__ __launch_bounds__(1024) void k_shard_plan(const int64_t* __restrict__ anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                                                      int world, ShardState* st, int64_t* counts_host) {
    __shared__ uint32_t ha[kShardBins], hb[kShardBins];
    __shared__ bool last_s;
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long cnt_s[kShardMaxWorld];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ha[tid] = 0u;  // kShardBins == blockDim.x == 1024
    hb[tid] = 0u;
    __syncthreads();
    const bool with_b = n_atoms_b > 0;
    for (int64_t p = blockIdx.x * 1024ll + tid; p < n_pairs; p += (int64_t)gridDim.x * 1024) {
        const longlong2 ab = reinterpret_cast<const longlong2*>(anchors)[p];
        atomicAdd(&ha[shard_bin(ab.x, n_atoms_a)], 1u);
        if (with_b) atomicAdd(&hb[shard_bin(ab.y, n_atoms_b)], 1u);
    }
    __syncthreads();
    {   // returning form, and the value is consumed: the atomics have been PERFORMED when the wave passes this point
        uint32_t r = 0;
        if (ha[tid]) r = atomicAdd(&st->hist[tid], ha[tid]);
        if (hb[tid]) r += atomicAdd(&st->hist_b[tid], hb[tid]);
        asm volatile("" ::"v"(r));
    }
    __syncthreads();
    if (tid == 0) last_s = last_workgroup_done(&st->done, 0ull, 0u);  // (the workgroup's histogram atomics completed before the barrier)
    __syncthreads();
    if (!last_s) return;
    // the last workgroup: for a key side, the exclusive scan of its 1024 bins (one per thread), rank(b) and the per-rank counts;
    // side A first, side B if A's partition is unbalanced, contiguous slices if B's is too
    const uint32_t va = __hip_atomic_load(&st->hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t vb = __hip_atomic_load(&st->hist_b[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto plan_side = [&](uint32_t v) -> bool {  // true: balanced (no rank holds more than 1.25 P / world + 1 pairs)
        const uint32_t incl = wave_incl_scan_u32(v);
        __syncthreads();
        if (lane == 63) wsum[wave] = incl;
        if (tid < kShardMaxWorld) cnt_s[tid] = 0ull;
        __syncthreads();
        unsigned long long pre = incl - v;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        int r = (int)((pre * (unsigned long long)world) / (unsigned long long)n_pairs);
        r = r < world - 1 ? r : world - 1;
        st->rank_of_bin[tid] = (uint16_t)r;
        if (v) atomicAdd(&cnt_s[r], (unsigned long long)v);
        __syncthreads();
        unsigned long long mx = 0;
        for (int w = 0; w < world; ++w) mx = cnt_s[w] > mx ? cnt_s[w] : mx;
        return mx * 4ull * (unsigned long long)world <= 5ull * (unsigned long long)n_pairs + 4ull * (unsigned long long)world;
    };
    int mode = 0;
    if (!plan_side(va)) mode = (with_b && plan_side(vb)) ? 1 : 2;
    st->hist[tid] = 0u;
    st->hist_b[tid] = 0u;
    __syncthreads();
    if (tid < world) {
        long long c = (long long)cnt_s[tid];
        if (mode == 2) {  // pairs p with floor(p * world / P) == tid: [ceil(tid P / world), ceil((tid + 1) P / world))
            const long long lo = ((long long)tid * n_pairs + world - 1) / world, hi = ((long long)(tid + 1) * n_pairs + world - 1) / world;
            c = hi - lo;
        }
        st->counts[tid] = c;
        counts_host[tid] = c;
    }
    if (tid == 0) { st->cursor = 0ull; st->mode = mode; counts_host[kShardMaxWorld] = mode; }
}
__global__ __launch_bounds__(256) void k_shard_select(const int64_t* __restrict__ anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                                                      int rank, int world, ShardState* st, int64_t* __restrict__ sel_anchors,
                                                      int64_t* __restrict__ sel_index) {
    __shared__ uint32_t wcnt[4];
    __shared__ unsigned long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = 8;  // pairs per thread and round: one cursor atomic per 2048 pairs
    const uint16_t* __restrict__ rob = st->rank_of_bin;
    const int mode = st->mode;
    for (int64_t p0 = (int64_t)blockIdx.x * 256 * PER; p0 < n_pairs; p0 += (int64_t)gridDim.x * 256 * PER) {
        longlong2 ab[PER];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t p = p0 + (int64_t)tid * PER + u;  // a thread owns PER consecutive pairs: original order inside a round
            if (p < n_pairs) {
                ab[u] = reinterpret_cast<const longlong2*>(anchors)[p];
                int r;
                if (mode == 2) r = (int)((p * world) / n_pairs);
                else r = rob[mode == 1 ? shard_bin(ab[u].y, n_atoms_b) : shard_bin(ab[u].x, n_atoms_a)];
                if (r == rank) mine |= 1u << u;
            }
        }
        const uint32_t c = (uint32_t)__popc(mine);
        const uint32_t incl = wave_incl_scan_u32(c);
        if (lane == 63) wcnt[wave] = incl;
        __syncthreads();
        if (tid == 0) base_s = atomicAdd(&st->cursor, (unsigned long long)(wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]));
        __syncthreads();
        unsigned long long o = base_s + incl - c;
        for (int w = 0; w < wave; ++w) o += wcnt[w];
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if ((mine >> u) & 1u) {
                reinterpret_cast<longlong2*>(sel_anchors)[o] = ab[u];
                sel_index[o] = p0 + (int64_t)tid * PER + u;
                ++o;
            }
        __syncthreads();
    }
}
__global__ void k_unshard_scores(const double* __restrict__ gathered, ShardCounts counts, int world, int64_t stride, double* __restrict__ out,
                                 int64_t n_pairs, uint32_t* bad) {
    // gathered: [world][2][stride] -- scores, then the original pair positions as int64 bit patterns
    for (int r = 0; r < world; ++r) {
        const double* sc = gathered + (int64_t)r * 2 * stride;
        const int64_t* ix = reinterpret_cast<const int64_t*>(sc + stride);
        for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < counts.n[r]; k += (int64_t)gridDim.x * blockDim.x) {
            const int64_t p = ix[k];
            if (p < 0 || p >= n_pairs) { *bad = 1u; continue; }
            out[p] = sc[k];
        }
    }
}
void launch_shard_plan(hipStream_t s, const int64_t* anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int world, ShardState* st,
                       int64_t* counts_host) {
    const int64_t nb = (n_pairs + 8191) / 8192;
    k_shard_plan<<<(unsigned)(nb < 256 ? (nb > 0 ? nb : 1) : 256), 1024, 0, s>>>(anchors, n_pairs, n_atoms_a, n_atoms_b, world, st, counts_host);
}
void launch_shard_select(hipStream_t s, const int64_t* anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int rank, int world,
                         ShardState* st, int64_t* sel_anchors, int64_t* sel_index) {
    const int64_t nb = (n_pairs + 2047) / 2048;
    k_shard_select<<<(unsigned)(nb < 1024 ? (nb > 0 ? nb : 1) : 1024), 256, 0, s>>>(anchors, n_pairs, n_atoms_a, n_atoms_b, rank, world, st, sel_anchors,
                                                                                    sel_index);
}
// ---- weight-function dictionaries: one set of F keys per function ------------------------------------------------------------
// The grouped environment kernel left DISTANCE keys in set 0 of the store (EnvStore::set_stride elements per set); one wavefront per
// environment writes F_w(distance) into set 1 + w for every function w of the dictionary (src/locohd.rs:230-283: each pair names its
// function; the sweep then reads the set of the pair's function and never evaluates a CDF).  A running maximum keeps every set sorted
// whatever the last bits of the floating-point CDF do (keys_to_cdf_lds does the same; equal F values are zero-width intervals).
__global__ __launch_bounds__(256) void k_env_key_sets(const DevConfig* __restrict__ cfgp, EnvStore ea, EnvStore eb, int n_sets, const DeviceStatus* st) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nu_a = st->n_unique[0], nu_b = st->n_unique[1];
    for (int64_t e = (int64_t)blockIdx.x * 4 + wave; e < nu_a + nu_b; e += (int64_t)gridDim.x * 4) {
        const bool on_b = e >= nu_a;
        const int64_t slot = on_b ? e - nu_a : e;
        uint64_t* const base = on_b ? eb.key : ea.key;
        const int64_t stride = on_b ? eb.stride : ea.stride, set_stride = on_b ? eb.set_stride : ea.set_stride;
        const int n = (on_b ? eb.len : ea.len)[slot];
        const uint64_t* __restrict__ src = base + slot * stride;
        for (int w = 0; w < n_sets; ++w) {
            const WfEntry wf = cfgp->wf[w];
            const double* __restrict__ prm = cfgp->wf_params + wf.offset;
            const double winv = cfgp->wf_inv[w];
            uint64_t* __restrict__ dst = base + (int64_t)(1 + w) * set_stride + slot * stride;
            uint64_t carry = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                uint64_t f = i < n ? d2u(cdf_lean(wf.kind, prm, wf.n_params, winv, u2d(src[i])) + 0.0) : 0ull;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint64_t t = shfl_up_u64(f, d);
                    if (lane >= d) f = t > f ? t : f;
                }
                f = carry > f ? carry : f;
                if (i < n) dst[i] = f;
                carry = shfl_u64(f, 63);
            }
        }
    }
}
void launch_env_key_sets(hipStream_t s, const DevConfig* cfg, const EnvStore& ea, const EnvStore& eb, int n_sets, int64_t max_envs, const DeviceStatus* st) {
    if (n_sets <= 0 || max_envs <= 0) return;
    const int64_t nb = (max_envs + 3) / 4;
    k_env_key_sets<<<(unsigned)(nb < 8192 ? nb : 8192), 256, 0, s>>>(cfg, ea, eb, n_sets, st);
}

// ---- deterministic mode: one order among equal keys ---------------------------------------------------------------------------
// The reference sorts an environment with a STABLE sort (utils.rs:25-39): equal distances keep their input order, one order for one
// input.  The environment kernels here break ties by the position a point happened to get in a cell list or a bucket (global / LDS
// atomics: another order in another run), and although a zero-width interval never counts, the O(1) Bhattacharyya update sees the
// categories in that order: scores moved by a few 1e-16 from run to run on lattice inputs.  Under lchd_ctx_set_deterministic one wavefront
// per environment sorts the CATEGORIES of every run of equal keys (positions >= 1: the first point is the anchor) -- two points of one
// key and one category are interchangeable, so the stored (key, category) sequence is then a function of the input alone.  Environments
// without ties (every random cloud) cost one read of their keys.
__global__ __launch_bounds__(256) void k_env_canon(EnvStore ea, EnvStore eb, int64_t n_a, int64_t n_b, const DeviceStatus* st) {
    __shared__ uint32_t cnt_s[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nu_a = st ? (int64_t)st->n_unique[0] : n_a, nu_b = st ? (int64_t)st->n_unique[1] : n_b;
    for (int64_t e = (int64_t)blockIdx.x * 4 + wave; e < nu_a + nu_b; e += (int64_t)gridDim.x * 4) {
        const bool on_b = e >= nu_a;
        const EnvStore& es = on_b ? eb : ea;
        const int64_t slot = on_b ? e - nu_a : e;
        const int len = es.len[slot];
        if (len < 3) continue;
        const uint64_t* __restrict__ key = es.key + slot * es.stride;
        uint8_t* const c8 = es.cat + slot * es.stride * (es.cat16 ? 2 : 1);
        uint16_t* const c16 = reinterpret_cast<uint16_t*>(c8);
        const bool wide = es.cat16 != 0;
        bool any = false;
        for (int i0 = 2; i0 < len; i0 += 64) {
            const int i = i0 + lane;
            any |= __ballot(i < len && key[i] == key[i - 1]) != 0ull;
        }
        if (!any) continue;
        for (int i0 = 1; i0 < len; i0 += 64) {
            const int i = i0 + lane;
            const bool head = i + 1 < len && (i == 1 || key[i] != key[i - 1]) && key[i + 1] == key[i];
            int end = i + 1;
            if (head) {
                const uint64_t k0 = key[i];
                while (end < len && key[end] == k0) ++end;
                if (end - i <= 64 || wide) {  // a short run (every lattice): insertion sort by this lane
                    for (int p = i + 1; p < end; ++p) {
                        const uint32_t v = wide ? (uint32_t)c16[p] : (uint32_t)c8[p];
                        int q = p - 1;
                        while (q >= i && (wide ? (uint32_t)c16[q] : (uint32_t)c8[q]) > v) {
                            if (wide) c16[q + 1] = c16[q]; else c8[q + 1] = c8[q];
                            --q;
                        }
                        if (wide) c16[q + 1] = (uint16_t)v; else c8[q + 1] = (uint8_t)v;
                    }
                }
            }
            // long runs (thousands of +inf entries of a distance matrix): the wavefront counts the run's categories and writes them out in order
            unsigned long long longs = __ballot(head && end - i > 64 && !wide);
            while (longs) {
                const int src = __ffsll((long long)longs) - 1;
                longs &= longs - 1;
                const int s0 = __shfl(i, src), s1 = __shfl(end, src);
#pragma unroll
                for (int k = 0; k < 4; ++k) cnt_s[wave][4 * lane + k] = 0u;
                wave_sync_lds();
                for (int p = s0 + lane; p < s1; p += 64) atomicAdd(&cnt_s[wave][c8[p]], 1u);
                wave_sync_lds();
                uint32_t c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) c[k] = cnt_s[wave][4 * lane + k];
                const uint32_t mine = c[0] + c[1] + c[2] + c[3];
                uint32_t at = (uint32_t)s0 + wave_incl_scan_u32(mine) - mine;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    for (uint32_t r = 0; r < c[k]; ++r) c8[at++] = (uint8_t)(4 * lane + k);
                wave_sync_lds();
            }
        }
    }
}
void launch_env_canon(hipStream_t s, const EnvStore& ea, const EnvStore& eb, int64_t n_a, int64_t n_b, const DeviceStatus* st) {
    const int64_t n = n_a + n_b;
    if (n <= 0) return;
    const int64_t nb = (n + 3) / 4;
    k_env_canon<<<(unsigned)(nb < 8192 ? nb : 8192), 256, 0, s>>>(ea, eb, n_a, n_b, st);
}

// ---- the pairs of a finished pass that touch an overflowed environment (EnvSide::ovf_list) ----------------------------------
// k_mark_overflow: overflow lists -> bit sets over the sides' slots (zeroed by the host).  k_select_overflow<false>: every
// wavefront counts the marked pairs of its contiguous share of the list; k_scan_overflow: exclusive scan of the (at most
// kOverflowWaves) counts; k_select_overflow<true>: the same walk again, now writing pair index, anchor pair and weight-function
// index in list order (ordered compaction: the second pass sees the pairs in the caller's order).
__global__ void k_mark_overflow(const uint32_t* __restrict__ list_a, uint32_t na, const uint32_t* __restrict__ list_b, uint32_t nb,
                                uint32_t* bits_a, uint32_t* bits_b) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < na + nb; k += gridDim.x * blockDim.x) {
        const uint32_t e = k < na ? list_a[k] : list_b[k - na];
        atomicOr(&(k < na ? bits_a : bits_b)[e >> 5], 1u << (e & 31));
    }
}
template <bool WRITE>
__global__ __launch_bounds__(64) void k_select_overflow(OverflowSelect a) {
    const int lane = threadIdx.x;
    const int64_t share = (a.n_pairs + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * share, p1 = p0 + share < a.n_pairs ? p0 + share : a.n_pairs;
    unsigned long long at = WRITE ? a.wave_count[blockIdx.x] : 0ull;  // (after the scan: the share's first position in the selection)
    for (int64_t q = p0; q < p1; q += 64) {
        const int64_t p = q + lane;
        bool big = false;
        int64_t ia = 0, ib = 0;
        if (p < p1) {
            ia = a.anchors[2 * p];
            ib = a.anchors[2 * p + 1];
            const uint32_t sa = a.slot_a[ia], sb = a.slot_b[ib];
            big = ((a.bits_a[sa >> 5] >> (sa & 31)) & 1u) | ((a.bits_b[sb >> 5] >> (sb & 31)) & 1u);
        }
        const unsigned long long m = __ballot(big);
        if constexpr (WRITE) {
            if (big) {
                const unsigned long long k = at + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                a.sel_index[k] = p;
                a.sel_anchors[2 * k] = ia;
                a.sel_anchors[2 * k + 1] = ib;
                if (a.wf) a.sel_wf[k] = a.wf[p];
            }
        }
        at += (unsigned long long)__popcll(m);
    }
    if constexpr (!WRITE)
        if (lane == 0) a.wave_count[blockIdx.x] = at;
}
__global__ __launch_bounds__(1024) void k_scan_overflow(unsigned long long* wave_count, int n, unsigned long long* total) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    constexpr int PER = kOverflowWaves / 1024;
    unsigned long long v[PER], sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) { v[u] = tid * PER + u < n ? wave_count[tid * PER + u] : 0ull; sum += v[u]; }
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long add = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    unsigned long long pre = part[tid] - sum;
#pragma unroll
    for (int u = 0; u < PER; ++u) { if (tid * PER + u < n) wave_count[tid * PER + u] = pre; pre += v[u]; }
    if (tid == 1023) *total = part[1023];
}
void launch_mark_overflow(hipStream_t s, const uint32_t* list_a, uint32_t na, const uint32_t* list_b, uint32_t nb, uint32_t* bits_a, uint32_t* bits_b) {
    if (na + nb == 0) return;
    const uint32_t nbk = (na + nb + 255) / 256;
    k_mark_overflow<<<nbk < 1024 ? nbk : 1024, 256, 0, s>>>(list_a, na, list_b, nb, bits_a, bits_b);
}
int overflow_select_waves(int64_t n_pairs) {
    const int64_t w = (n_pairs + 255) / 256;
    return (int)(w < kOverflowWaves ? (w > 0 ? w : 1) : kOverflowWaves);
}
void launch_count_overflow(hipStream_t s, const OverflowSelect& a, unsigned long long* total) {
    const int w = overflow_select_waves(a.n_pairs);
    k_select_overflow<false><<<w, 64, 0, s>>>(a);
    k_scan_overflow<<<1, 1024, 0, s>>>(a.wave_count, w, total);
}
void launch_write_overflow(hipStream_t s, const OverflowSelect& a) {
    k_select_overflow<true><<<overflow_select_waves(a.n_pairs), 64, 0, s>>>(a);
}

// one share's scores back to their positions in the caller's order (out: any device-visible memory, e.g. a host-mapped block)
__global__ void k_scatter_scores(const double* __restrict__ scores, const int64_t* __restrict__ index, int64_t n, double* __restrict__ out) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) out[index[k]] = scores[k];
}
void launch_scatter_scores(hipStream_t s, const double* scores, const int64_t* index, int64_t n, double* out) {
    if (n <= 0) return;
    const int64_t nb = (n + 255) / 256;
    k_scatter_scores<<<(unsigned)(nb < 4096 ? nb : 4096), 256, 0, s>>>(scores, index, n, out);
}
void launch_unshard_scores(hipStream_t s, const double* gathered, const ShardCounts& counts, int world, int64_t stride, double* out,
                           int64_t n_pairs, uint32_t* bad) {
    k_unshard_scores<<<1024, 256, 0, s>>>(gathered, counts, world, stride, out, n_pairs, bad);
}
}  // namespace lchd

namespace lchd {
__global__ void k_fill_sqrt_tables(double* sqrt_tab, double* rsqrt_tab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 65536) {
        const double r = sqrt((double)k);
        sqrt_tab[k] = r;
        rsqrt_tab[k] = 1.0 / r;
    }
}
__global__ void k_fill_pow_tables(double* tab, double einv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 65536) {
        tab[k] = pow((double)k, einv);
        tab[65536 + k] = pow((double)k, -einv);
    }
}
void launch_fill_pow_tables(hipStream_t s, double* tab, double exponent) { k_fill_pow_tables<<<256, 256, 0, s>>>(tab, 1.0 / exponent); }
void launch_fill_sqrt_tables(hipStream_t s, double* sqrt_tab, double* rsqrt_tab) {
    k_fill_sqrt_tables<<<256, 256, 0, s>>>(sqrt_tab, rsqrt_tab);
}
}  // namespace lchd

namespace lchd {
__global__ void k_frames_labels(const uint8_t* tcat, const int32_t* ttag, int64_t n_tmpl, int64_t total, uint8_t* cat, int32_t* tag,
                                int32_t* sid) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / n_tmpl, k = i - f * n_tmpl;
        cat[i] = tcat[k];
        tag[i] = ttag[k];
        sid[i] = (int32_t)f;
    }
}
void launch_frames_labels(hipStream_t s, const uint8_t* tcat, const int32_t* ttag, int64_t n_tmpl, int32_t n_frames, uint8_t* cat,
                          int32_t* tag, int32_t* sid) {
    k_frames_labels<<<2048, 256, 0, s>>>(tcat, ttag, n_tmpl, n_tmpl * n_frames, cat, tag, sid);
}

// order-preserving map double -> u64 (so that atomicMin / atomicMax on integers order like the doubles)
__device__ __forceinline__ unsigned long long ordered_key(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__global__ void k_frames_unpack(const double* __restrict__ raw, int64_t n, double* __restrict__ x, double* __restrict__ y,
                                double* __restrict__ z, unsigned long long* bbox7) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double vx = raw[3 * i], vy = raw[3 * i + 1], vz = raw[3 * i + 2];
        x[i] = vx; y[i] = vy; z[i] = vz;
        bad = bad || !(fabs(vx) < INFINITY) || !(fabs(vy) < INFINITY) || !(fabs(vz) < INFINITY);
        mn[0] = fmin(mn[0], vx); mn[1] = fmin(mn[1], vy); mn[2] = fmin(mn[2], vz);
        mx[0] = fmax(mx[0], vx); mx[1] = fmax(mx[1], vy); mx[2] = fmax(mx[2], vz);
    }
    for (int m = 32; m > 0; m >>= 1)
        for (int k = 0; k < 3; ++k) { mn[k] = fmin(mn[k], shfl_xor_f64(mn[k], m)); mx[k] = fmax(mx[k], shfl_xor_f64(mx[k], m)); }
    const unsigned long long anybad = __ballot(bad);
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; ++k) { atomicMin(&bbox7[k], ordered_key(mn[k])); atomicMax(&bbox7[3 + k], ordered_key(mx[k])); }
        if (anybad) atomicOr(&bbox7[6], 1ull);
    }
}
void launch_frames_unpack(hipStream_t s, const double* raw, int64_t n_atoms, double* x, double* y, double* z, unsigned long long* bbox7) {
    static const unsigned long long init[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    (void)hipMemcpyAsync(bbox7, init, sizeof init, hipMemcpyHostToDevice, s);
    const int64_t nb = (n_atoms + 255) / 256;
    k_frames_unpack<<<(unsigned)(nb < 1024 ? nb : 1024), 256, 0, s>>>(raw, n_atoms, x, y, z, bbox7);
}

// Frames given as SOURCE atoms (float32, the precision of Bio.PDB Atom.coord / MDAnalysis Timestep.positions): primitive
// atom p of every frame is the centroid of source atoms src_idx[src_start[p] .. src_start[p+1]) -- what
// PrimitiveAssigner.assign_primitive_structure computes per frame on the host with np.mean(atom_coords, axis=0)
// (/root/reference/loco_hd/atom_converter_utils.py:106-126, python_codes/trajectory_analyzer.py:55-72).  Same arithmetic
// as that call: float32 accumulator starting from +0, members added in list order, one IEEE float32 division
// by the member count; the result is widened to f64 exactly like PrimitiveAtom.coordinates.
//
// HBM-bound by construction: the host cuts the primitive atoms into tiles whose members span at most kCentroidSpan
// consecutive source atoms (typing schemes walk residues in order, so the CSR map is local); a workgroup streams one
// (frame, tile) slice of the source coordinates into LDS with 16-byte coalesced loads, gathers the members from LDS
// (stride 3 floats: conflict-free) and writes x/y/z coalesced.  Every source atom is read once per frame, every
// primitive atom written once: 12 B x n_src + 24 B x n_prim per frame.  A tile whose span does not fit (lo == hi == -1)
// gathers straight from global memory.  The bounding box is reduced per workgroup before it touches the 7 global words.
constexpr int kCentroidSpan = 4096;  // source atoms per tile: 48 KB of LDS
constexpr int kBboxParts = 4096;     // capacity of a frames buffer's per-workgroup bounding-box partials
__global__ __launch_bounds__(256) void k_frames_centroids(const float* __restrict__ raw, int64_t n_src, const int32_t* __restrict__ src_start,
                                                          const int32_t* __restrict__ src_idx, const int4* __restrict__ tiles, int n_tiles,
                                                          int64_t n_prim, int64_t n_items, double* __restrict__ x, double* __restrict__ y,
                                                          double* __restrict__ z, unsigned long long* __restrict__ bbox_part) {
    __shared__ __attribute__((aligned(16))) float s_xyz[kCentroidSpan * 3 + 8];
    __shared__ double s_red[4][6];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int64_t w = blockIdx.x; w < n_items; w += gridDim.x) {
        const int64_t f = w / n_tiles;
        const int4 t = tiles[(int)(w - f * n_tiles)];  // {first primitive, end primitive, first source atom, end source atom}
        const float* fr = raw + 3 * f * n_src;
        const bool staged = t.z >= 0;
        int head = 0;
        if (staged) {
            // floats [g0, g1) of the buffer = source atoms [lo, hi) of this frame.  LDS origin `base` = the 16-byte aligned
            // ADDRESS at or below g0 (the caller's pointer need not be 16-byte aligned; base can be < 0 only for the very
            // first floats of the buffer, which are then copied one by one).
            const int64_t g0 = 3 * f * n_src + 3 * (int64_t)t.z, g1 = 3 * f * n_src + 3 * (int64_t)t.w;
            const int m = (int)((reinterpret_cast<uintptr_t>(raw) >> 2) & 3);
            const int64_t base = ((g0 + m) & ~(int64_t)3) - m;
            const int64_t v0 = base < 0 ? base + 4 : base;  // first aligned float4 inside the buffer
            head = (int)(g0 - base);
            const int n4 = g1 > v0 ? (int)((g1 - v0) >> 2) : 0;
            const float4* src4 = reinterpret_cast<const float4*>(raw + v0);
            float4* dst4 = reinterpret_cast<float4*>(s_xyz) + ((v0 - base) >> 2);
            for (int k = tid; k < n4; k += 256) dst4[k] = src4[k];
            for (int64_t k = g0 + tid; k < v0; k += 256) s_xyz[k - base] = raw[k];
            for (int64_t k = v0 + 4 * (int64_t)n4 + tid; k < g1; k += 256) s_xyz[k - base] = raw[k];
            __syncthreads();
        }
        for (int p = t.x + tid; p < t.y; p += 256) {
            const int b = src_start[p], e = src_start[p + 1];
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;  // NumPy's add.reduce starts from +0: a lone -0.0 member comes out as +0.0
            if (staged) {
                for (int k = b; k < e; ++k) {
                    const float* a = s_xyz + head + 3 * (src_idx[k] - t.z);
                    sx += a[0]; sy += a[1]; sz += a[2];
                }
            } else {
                for (int k = b; k < e; ++k) {
                    const float* a = fr + 3 * (int64_t)src_idx[k];
                    sx += a[0]; sy += a[1]; sz += a[2];
                }
            }
            const float cnt = (float)(e - b);
            const double vx = (double)__fdiv_rn(sx, cnt), vy = (double)__fdiv_rn(sy, cnt), vz = (double)__fdiv_rn(sz, cnt);
            const int64_t o = f * n_prim + p;
            x[o] = vx; y[o] = vy; z[o] = vz;
            bad = bad || !(fabs(vx) < INFINITY) || !(fabs(vy) < INFINITY) || !(fabs(vz) < INFINITY);
            mn[0] = fmin(mn[0], vx); mn[1] = fmin(mn[1], vy); mn[2] = fmin(mn[2], vz);
            mx[0] = fmax(mx[0], vx); mx[1] = fmax(mx[1], vy); mx[2] = fmax(mx[2], vz);
        }
        if (staged) __syncthreads();  // the tile is consumed before the next one overwrites it
    }
    for (int m = 32; m > 0; m >>= 1)
        for (int k = 0; k < 3; ++k) { mn[k] = fmin(mn[k], shfl_xor_f64(mn[k], m)); mx[k] = fmax(mx[k], shfl_xor_f64(mx[k], m)); }
    const unsigned long long anybad = __ballot(bad);
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if ((tid & 63) == 0) {
        for (int k = 0; k < 3; ++k) { s_red[tid >> 6][k] = mn[k]; s_red[tid >> 6][3 + k] = mx[k]; }
        if (anybad) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    // per-workgroup partial result, no atomics: k_bbox_finish folds the partials into the 7 words the host reads
    if (tid < 6) {
        double v = s_red[0][tid];
        for (int wv = 1; wv < 4; ++wv) v = tid < 3 ? fmin(v, s_red[wv][tid]) : fmax(v, s_red[wv][tid]);
        bbox_part[(size_t)blockIdx.x * 7 + tid] = ordered_key(v);
    }
    if (tid == 6) bbox_part[(size_t)blockIdx.x * 7 + 6] = s_bad ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void k_bbox_finish(const unsigned long long* __restrict__ part, int n_parts, unsigned long long* bbox7) {
    __shared__ unsigned long long red[4][7];
    const int tid = threadIdx.x;
    unsigned long long v[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    for (int b = tid; b < n_parts; b += 256) {
        for (int k = 0; k < 3; ++k) { v[k] = min(v[k], part[(size_t)b * 7 + k]); v[3 + k] = max(v[3 + k], part[(size_t)b * 7 + 3 + k]); }
        v[6] |= part[(size_t)b * 7 + 6];
    }
    for (int m = 32; m > 0; m >>= 1)
        for (int k = 0; k < 7; ++k) {
            const unsigned long long o = shfl_u64(v[k], (tid & 63) ^ m);
            v[k] = k < 3 ? min(v[k], o) : (k < 6 ? max(v[k], o) : (v[k] | o));
        }
    if ((tid & 63) == 0) for (int k = 0; k < 7; ++k) red[tid >> 6][k] = v[k];
    __syncthreads();
    if (tid < 7) {
        unsigned long long r = red[0][tid];
        for (int w = 1; w < 4; ++w) r = tid < 3 ? min(r, red[w][tid]) : (tid < 6 ? max(r, red[w][tid]) : (r | red[w][tid]));
        bbox7[tid] = r;
    }
}
void launch_frames_centroids(hipStream_t s, const float* raw, int64_t n_src, const int32_t* src_start, const int32_t* src_idx,
                             const int32_t* tiles, int n_tiles, int64_t n_prim, int32_t n_frames, double* x, double* y, double* z,
                             unsigned long long* bbox7, unsigned long long* bbox_part) {
    const int64_t items = (int64_t)n_tiles * n_frames;
    const int grid = (int)(items < kBboxParts ? items : kBboxParts);  // more workgroups than fit at once: the dispatcher balances them
    k_frames_centroids<<<grid, 256, 0, s>>>(raw, n_src, src_start, src_idx, reinterpret_cast<const int4*>(tiles), n_tiles, n_prim, items, x, y,
                                            z, bbox_part);
    k_bbox_finish<<<1, 256, 0, s>>>(bbox_part, grid, bbox7);
}
int centroid_tile_span() { return kCentroidSpan; }
int bbox_parts_capacity() { return kBboxParts; }
}  // namespace lchd

// lchd_env_group.hip -- K1 (thresholded), the common case: environments of at most 512 points, SEVERAL per wavefront.
//
// Replaces, for the default capacity, the one-environment-per-wavefront kernel k_env_cells (lchd_env_cells.hip), which spent
// most of its vector instructions outside the distance arithmetic (reference: env_from_idx, /root/reference/src/locohd.rs:514-542;
// utils::sort_together, utils.rs:25-39; euclidean_distance, utils.rs:1-8; KdTree::within_radius, :521):
//
//   finer grid     cells of thr / 2 instead of thr, neighbourhood 5 x 5 x 5: the candidate volume around a sphere of radius thr
//                  falls from ~22 thr^3 (17 of 27 cells of 1.1 thr) to ~9 thr^3, i.e. 2.4x fewer records loaded and tested
//   group table    the (up to) 25 contiguous cell-row runs of an anchor are cut into groups of 8 records; a table of
//                  (first record, count) per group lives in LDS, so candidate t of a search step finds its record with ONE
//                  LDS read instead of a 16-instruction compare-select chain over the row offsets
//   paired set-up  the row bounds of TWO anchors are worked out per pass (lanes 0..24 and 32..56: one row each)
//   groups         a wavefront appends the survivors of consecutive anchors to ONE flat LDS buffer (<= 512 points, <= 8
//                  environments) and then sorts / converts / writes them together: the bucket sort, the CDF keying and the
//                  write-out run on full wavefronts however small the individual environments are (C4: ~96 points each)
//
// Semantics are those of k_env_cells: keep p iff sum(diff^2) < thr^2 (uncontracted, same summation order) and (p is the anchor
// itself or the tag rule accepts), distance = sqrt(sum), ascending order (ties in any order: zero-width intervals), optional
// F(distance) keys, categories outside the map flagged and stored as 0.  An environment of more than 512 points, or an anchor
// with more than kGTab candidate groups, is reported as ST_ENV_OVERFLOW and the host repeats the pass with the larger
// instantiation or with k_env_cells.
#include "lchd_kcommon.h"

#ifndef LCHD_GROUP_U
#define LCHD_GROUP_U 4   // search steps (64 candidates each) whose record loads are issued together
#endif

namespace lchd {

#define LCHD_AS4 __attribute__((address_space(4)))  // the constant address space: kernel arguments, the configuration blob

// Two instantiations: CAP = points of one group (flat LDS buffer) = the largest environment the instantiation handles.
//   <kEnvGroupCap, 5 waves/SIMD>       8.0 KB of LDS, 20 workgroups per CU
//   <kEnvGroupCapSmall, 6 waves/SIMD>  6.1 KB, 24 workgroups per CU: ~5 % faster where every environment fits (the host picks it
//                                      when the previous pass of the context had no environment beyond kEnvGroupSmallUpTo points)
constexpr int kGMax = 8;             // environments per group
constexpr int kGBuckets = 512;       // distance buckets of a group's sort (split evenly between its environments), 16-bit counters
constexpr int kGTab = 224;           // candidate groups (8 records each) per anchor; the table is padded to whole search rounds

template <int CAP>
struct GroupLds {
    uint64_t key[CAP];                 // d^2 while a group is being collected, then sorted distances
    uint16_t val[CAP];                 // category | environment-in-group << 8
    uint32_t hist[kGBuckets / 2 + 4];  // two 16-bit bucket counters per word (+ the end marker)
    uint32_t tab[2][kGTab];            // (byte offset of the first record) | records (1..8; 0 = padding), per half-wave anchor
    uint32_t gstart[kGMax], gcount[kGMax];
};
static_assert(sizeof(GroupLds<kEnvGroupCap>) <= 8192 && sizeof(GroupLds<kEnvGroupCapSmall>) <= 6400, "20 / 24 workgroups per CU (160 KB of LDS)");
static_assert(alignof(EnvSides) == 8 && sizeof(void*) == 8, "kernel-argument layout assumed by k_env_group");
static_assert(kEnvGroupCap == 512, "k_env_group addresses environment slots with << 9");

#ifdef LCHD_SWEEP_STAMPS
__device__ unsigned long long g_envg_stamps[8];
#define GSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && (blockIdx.x & 63) == 0) atomicAdd(&g_envg_stamps[i], t_ - gstamp_last); gstamp_last = t_; } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

// inclusive prefix sum inside each 32-lane half (row_shr 1/2/4/8, then row 0 -> 1 and row 2 -> 3)
__device__ __forceinline__ uint32_t half_scan_u32(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    return (uint32_t)v;
}

// The kernel is one long loop (set-up -> search -> ... -> sort -> write-out) whose phases need different sets of wave-uniform
// values (grid geometry; anchor coordinates; weight-function parameters and store pointers).  Held live across the whole loop
// they exceed the scalar register file and the compiler spills them into VGPR lanes -- every use then costs a v_readlane,
// which was a third of the first version's vector instructions.  So every phase RE-LOADS what it needs from the constant
// address space (kernel arguments / configuration blob: scalar loads, a few per phase); the empty asm makes the pointer opaque
// so that the loads are not hoisted out of the loop again.
template <class T>
__device__ __forceinline__ const LCHD_AS4 T* opaque(const LCHD_AS4 T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
template <class T>
__device__ __forceinline__ const LCHD_AS4 T* as_const(const T* p) {  // memory that no kernel of the pass writes
    return (const LCHD_AS4 T*)(unsigned long long)p;
}

// An environment that does not fit this instantiation (one lane): flagged, its size reported, its slot index appended to the side's
// overflow list; the slot itself receives the anchor alone -- a valid one-point environment, so that the sweeps of this pass run
// cleanly over the pairs of this anchor (the host scores those pairs again with larger slots: lchd_ctx_finish).
// bound != 0: the anchor's candidate table overflowed before anything was counted; `bound` candidates are an upper bound of the environment.
#ifndef LCHD_OVF_VARIANT
#define LCHD_OVF_VARIANT 1
#endif
#if LCHD_OVF_VARIANT == 2
__device__ __attribute__((noinline))
#else
__device__ __forceinline__
#endif
void env_group_overflow(const LCHD_AS4 EnvSide* p, DeviceStatus* st, int side, int e, uint32_t count, uint32_t bound, uint32_t apos32, int n_cat) {
#if LCHD_OVF_VARIANT == 0
    atomicOr(&st->flags, ST_ENV_OVERFLOW); atomicMax(&st->max_env, count); p->env.len[e] = 0; return;
#endif
    atomicOr(&st->flags, ST_ENV_OVERFLOW);
    atomicMax(&st->max_env, count);
    if (bound) atomicMax(&st->max_bound, bound);
    const uint32_t k = atomicAdd(&st->n_overflow[side], 1u);
    if (p->ovf_list) p->ovf_list[k] = (uint32_t)e;
    const uint32_t cat = reinterpret_cast<const CellRec*>(reinterpret_cast<const char*>(p->g.rec) + apos32)->cat;
    p->env.len[e] = 1;
    p->env.key[(uint32_t)e << 9] = 0ull;  // distance 0 = F(0) bits for the weight functions of cdfs.rs (all start at 0)
    p->env.cat[(uint32_t)e << 9] = (int)cat < n_cat ? (uint8_t)cat : (uint8_t)0;
    if (p->env.cat0) p->env.cat0[e] = (int)cat < n_cat ? (uint8_t)cat : (uint8_t)0;
    if (p->env.pre) {  // (row 0 of the one-point environment)
        const uint32_t cs = (int)cat < n_cat ? cat : 0u;
        const uint64_t one = 1ull << ((cs & 7u) * 8u);
        const size_t r0 = ((size_t)(uint32_t)e << 9) / kPreStep;
        for (int k = 0; k < p->env.pre_words; ++k) p->env.pre[r0 * p->env.pre_words + k] = (cs >> 3) == (uint32_t)k ? one : 0ull;
    }
}

#ifndef LCHD_CAT0_STORE
#define LCHD_CAT0_STORE 1
#endif
#ifndef LCHD_GROUP_WPB
#define LCHD_GROUP_WPB 1   // independent wavefronts per workgroup (each with its own LDS block; no workgroup barrier anywhere)
#endif
template <bool TAGLIST, int kGCap, int WAVES>
__global__ __launch_bounds__(64 * LCHD_GROUP_WPB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_env_group(
    const DevConfig* __restrict__ cfgp, EnvSides sides, double thr, int apw, int nwa, DeviceStatus* st) {
    static_assert(kGCap % 64 == 0 && kGCap <= kEnvGroupCap, "whole wavefronts; environment slots hold kEnvGroupCap points");
    __shared__ __attribute__((aligned(16))) GroupLds<kGCap> lds_all[LCHD_GROUP_WPB];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    GroupLds<kGCap>& lds = lds_all[wave];
    const int wid = (int)blockIdx.x * LCHD_GROUP_WPB + wave;  // nwa wavefronts build side A, the rest side B
    const int side = wid >= nwa ? 1 : 0;
    const int lane = threadIdx.x & 63;
#ifdef LCHD_SWEEP_STAMPS
    unsigned long long gstamp_last = __builtin_amdgcn_s_memtime();
#endif
    const int n_uniq = (int)st->n_unique[side];
    const int e_begin = (wid - (side ? nwa : 0)) * apw;
    if (e_begin >= n_uniq) return;
    const int e_end = min(e_begin + apw, n_uniq);
    // this side's block of the kernel arguments (cfgp at offset 0, `sides` behind it)
    const LCHD_AS4 EnvSide* const ks =
        (const LCHD_AS4 EnvSide*)((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + 8ull + (unsigned long long)side * sizeof(EnvSide));
    const LCHD_AS4 DevConfig* const kc = as_const(cfgp);
    const double thr2 = thr * thr;
    const bool accept_same = kc->tag_accept_same != 0;
    DevConfig tcfg{};  // the tag-rule words only (TAGLIST)
    if constexpr (TAGLIST) {
        tcfg.tag_mode = kc->tag_mode; tcfg.tag_accept_same = kc->tag_accept_same; tcfg.tag_accepted_pairs = kc->tag_accepted_pairs;
        tcfg.tag_ordered = kc->tag_ordered; tcfg.n_tag_pairs = kc->n_tag_pairs; tcfg.tag_pairs = kc->tag_pairs;
    }
    const char* const recb = reinterpret_cast<const char*>(ks->g.rec);
    const uint32_t sub32 = (uint32_t)(lane & 7) << 5, subc = (uint32_t)(lane & 7);

    // the group being collected (all wave-uniform)
    int fill = 0, ngrp = 0, e_first = 0;
    // kept points per candidate record, running estimate of this wavefront (starts pessimistic) -- only used to close a group
    // BEFORE an environment that is unlikely to fit
    float seen_cand = 16.f, seen_pts = 8.f;
    // set-up state: tables and per-lane anchor values of anchors setup_base (lanes 0..31) and setup_base + 1 (lanes 32..63)
    int setup_base = -2;
    double v_ax = 0.0, v_ay = 0.0, v_az = 0.0;
    uint32_t v_tag = 0u, v_apos32 = 0u;
    int ng_h0 = 0, ng_h1 = 0;
    bool bad = false;

    int j = e_begin;
    while (true) {
        bool do_flush = false;
        if (j < e_end) {
            if (j >= setup_base + 2) {
                // ---------------------------------------------------------------------------------- set-up of anchors j, j + 1
                const LCHD_AS4 EnvSide* p = opaque(ks);
                const double gmin0 = p->g.min[0], gmin1 = p->g.min[1], gmin2 = p->g.min[2];
                const double ginv0 = p->g.inv[0], ginv1 = p->g.inv[1], ginv2 = p->g.inv[2];
                const double gcell0 = p->g.cell[0], gcell1 = p->g.cell[1], gcell2 = p->g.cell[2];
                const int dim0 = p->g.dim[0], dim1 = p->g.dim[1], dim2 = p->g.dim[2];
                const uint32_t* __restrict__ cell_start = p->g.cell_start;
                const AnchorRec* __restrict__ uniq = p->uniq;
                const double thr2m = thr2 * (1.0 + 1e-6);
                // (lane-derived predicates of this phase must not be hoisted out of the loop: as loop invariants they are 64-bit
                // lane masks, two scalar registers each)
                int ls = lane;
                asm volatile("" : "+v"(ls));
                const int h = ls >> 5, r = ls & 31;
                const AnchorRec a = uniq[min(j + h, e_end - 1)];
                v_ax = a.x; v_ay = a.y; v_az = a.z; v_tag = a.tag; v_apos32 = a.apos << 5;
                const int cx = cell_coord(a.x, gmin0, ginv0, dim0);
                const int cy = cell_coord(a.y, gmin1, ginv1, dim1);
                const int cz = cell_coord(a.z, gmin2, ginv2, dim2);
                const double fx = (a.x - gmin0) * ginv0 - (double)cx, fy = (a.y - gmin1) * ginv1 - (double)cy,
                             fz = (a.z - gmin2) * ginv2 - (double)cz;
                const int rr = r < 25 ? r : 24;
                const int kz = (rr * 13) >> 6, ky = rr - 5 * kz;  // rr / 5, rr % 5
                const int oy = ky - 2, oz = kz - 2;
                const int yy = cy + oy, zz = cz + oz;
                // Cells that lie wholly outside the radius are skipped: with the anchor at fractional position f in its cell, a
                // row at offset o != 0 along an axis is at least (f + |o| - 1) (o < 0) or (1 - f + o - 1) (o > 0) cell edges
                // away along that axis.  The test carries a relative margin of 1e-6 on thr^2 (the rounding of the cell
                // assignment is ~1e-16).
                const double ty = oy < 0 ? fy - (double)(oy + 1) : (1.0 - fy) + (double)(oy - 1);
                const double tz = oz < 0 ? fz - (double)(oz + 1) : (1.0 - fz) + (double)(oz - 1);
                const double gy = oy == 0 ? 0.0 : fmax(ty, 0.0) * gcell1, gz = oz == 0 ? 0.0 : fmax(tz, 0.0) * gcell2;
                const double r2 = gy * gy + gz * gz;
                const double fxl = fmax(fx, 0.0), fxh = fmax(1.0 - fx, 0.0);
                const double xm1 = fxl * gcell0, xm2 = (fxl + 1.0) * gcell0, xp1 = fxh * gcell0, xp2 = (fxh + 1.0) * gcell0;
                const int lo = (r2 + xm2 * xm2 < thr2m) ? -2 : ((r2 + xm1 * xm1 < thr2m) ? -1 : 0);
                const int hi = (r2 + xp2 * xp2 < thr2m) ? 2 : ((r2 + xp1 * xp1 < thr2m) ? 1 : 0);
                const int xl = max(cx + lo, 0), xh = min(cx + hi, dim0 - 1);
                const bool in = (r < 25) & (j + h < e_end) & ((unsigned)zz < (unsigned)dim2) & ((unsigned)yy < (unsigned)dim1) & (r2 < thr2m);
                const int row = in ? ((a.sid * dim2 + zz) * dim1 + yy) * dim0 : 0;  // (the host bounds the number of cells: 2^23)
                const int b_ = (int)cell_start[row + xl], e_ = (int)cell_start[row + xh + 1];
                const int len = in ? e_ - b_ : 0;
                const uint32_t ng = (uint32_t)(len + 7) >> 3;
                const uint32_t incl = half_scan_u32(ng);
                const uint32_t goff = incl - ng;
                ng_h0 = __builtin_amdgcn_readlane((int)incl, 31);
                ng_h1 = __builtin_amdgcn_readlane((int)incl, 63);
                uint32_t* tb = lds.tab[h];
                wave_sync_lds();  // (the searches of the previous set-up have read their tables)
                {
                    uint32_t ent = ((uint32_t)b_ << 5) | 8u;  // byte offset of the group's first record | records in the group
                    int left = len;
                    uint32_t at = goff;
                    while (__builtin_amdgcn_ballot_w64(left > 0)) {
                        if (left > 0 && at < (uint32_t)kGTab) tb[at] = left >= 8 ? ent : (ent & ~15u) | (uint32_t)left;
                        ent += 8u << 5;
                        left -= 8;
                        ++at;
                    }
                    // padding: the search reads whole rounds of 8 * U groups; entries past the anchor's last group hold 0 records
                    const uint32_t ngh = h ? (uint32_t)ng_h1 : (uint32_t)ng_h0;
                    if (ngh + (uint32_t)r < (uint32_t)kGTab) tb[ngh + (uint32_t)r] = 0u;
                }
                wave_sync_lds();
                setup_base = j;
                GSTAMP(0);
            }
            const int hh = j - setup_base;                 // 0 or 1
            const int NG = hh ? ng_h1 : ng_h0;
            const int e = j;
            if (NG > kGTab) {
                // more candidate groups than the table holds (a very dense neighbourhood): the host repeats the pass with the
                // one-environment-per-workgroup kernel, whose capacity grows
                if (lane == 0)
                    env_group_overflow(opaque(ks), st, (unsigned long long)opaque(ks) != (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + 8ull ? 1 : 0, e, (uint32_t)(kGCap + 1), (uint32_t)NG * 8u, (uint32_t)__builtin_amdgcn_readlane((int)v_apos32, 32 * hh), opaque(kc)->n_categories);
                ++j;
                do_flush = true;
            } else if (ngrp > 0 && fill + (int)((float)(NG * 8) * seen_pts * __builtin_amdgcn_rcpf(seen_cand)) > kGCap) {
                do_flush = true;  // (the anchor is searched after the flush, into an empty buffer)
            } else {
                // ---------------------------------------------------------------------------------- radius search of anchor j
                const int src = 32 * hh;
                const double ax = readlane_f64(v_ax, src), ay = readlane_f64(v_ay, src), az = readlane_f64(v_az, src);
                const int32_t atag = __builtin_amdgcn_readlane((int)v_tag, src);
                const uint32_t apos32 = (uint32_t)__builtin_amdgcn_readlane((int)v_apos32, src);
                const uint32_t* tb = lds.tab[hh] + (lane >> 3);
                const uint32_t qbits = (uint32_t)ngrp << 8;
                auto tag_ok = [&](int32_t t_other) -> bool {  // tag_pairing_rule.rs:49-75
                    if constexpr (TAGLIST) return tag_pair_accepted(tcfg, atag, t_other);
                    else return (atag == t_other) == accept_same;
                };
                int count = 0;
                constexpr int U = LCHD_GROUP_U;
                static_assert(8 * U <= 32, "table padding covers one round of 8 U groups");
                for (int g0 = 0; g0 < NG; g0 += 8 * U) {
                    uint32_t off[U], ent[U];
                    double2 R0[U], R1[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        ent[u] = tb[g0 + 8 * u];
                        off[u] = (ent[u] & ~31u) + sub32;  // (may run up to 7 records past the row: the record array is padded)
                        R0[u] = *reinterpret_cast<const double2*>(recb + off[u]);
                        R1[u] = *reinterpret_cast<const double2*>(recb + off[u] + 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // all 2U loads are issued before the first distance is computed
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (g0 + 8 * u < NG) {  // wave-uniform
                            const double dx = R0[u].x - ax, dy = R0[u].y - ay, dz = R1[u].x - az;
                            double d2 = dx * dx;   // TU is built with -ffp-contract=off: same roundings as the
                            d2 = d2 + dy * dy;     // reference's `distance += diff * diff`
                            d2 = d2 + dz * dz;
                            const uint64_t tc = d2u(R1[u].y);  // tag | cat << 32
                            const bool vld = subc < (ent[u] & 15u);
                            bool ok = false;
                            if constexpr (TAGLIST) {
                                if (vld && d2 < thr2) ok = (off[u] == apos32) || tag_ok((int32_t)(uint32_t)tc);
                            } else {
                                ok = (vld & (d2 < thr2)) & ((off[u] == apos32) | tag_ok((int32_t)(uint32_t)tc));
                            }
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
                            if (ok) {
                                const int pos = fill + count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                                if (pos < kGCap) {
                                    lds.key[pos] = d2u(d2);  // the square root is taken after compaction
                                    lds.val[pos] = (uint16_t)(((uint32_t)(tc >> 32) & 0xFFu) | qbits);
                                }
                            }
                            count += __popcll(m);
                        }
                    }
                }
                GSTAMP(1);
                seen_cand += (float)(NG * 8);
                seen_pts += (float)(count + 8);
                if (fill + count > kGCap) {
                    if (ngrp == 0) {  // this environment alone is too large: the host re-launches a larger variant
                        if (lane == 0) env_group_overflow(opaque(ks), st, (unsigned long long)opaque(ks) != (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + 8ull ? 1 : 0, e, (uint32_t)count, 0u, apos32, opaque(kc)->n_categories);
                        ++j;
                    } else {
                        do_flush = true;  // close the group; the anchor is searched again into the empty buffer
                    }
                } else if (count == 0) {
                    if (lane == 0) { atomicOr(&st->flags, ST_EMPTY_ENV); opaque(ks)->env.len[e] = 0; }
                    ++j;
                    do_flush = true;  // (a group holds CONSECUTIVE anchors)
                } else {
                    if (ngrp == 0) e_first = e;
                    if (lane == 0) { lds.gstart[ngrp] = (uint32_t)fill; lds.gcount[ngrp] = (uint32_t)count; }
                    fill += count;
                    ++ngrp;
                    ++j;
                    do_flush = ngrp == kGMax;
                }
            }
        } else {
            if (ngrp == 0) break;
            do_flush = true;
        }
        if (!do_flush || ngrp == 0) continue;

        // ------------------------------------------------------------------------------------------ sort of the group
        // O(n) bucket sort as in k_env_cells: inside a sphere the number of points grows like d^3, so bucket =
        // floor(Bq (d / thr)^3) spreads an environment's points almost evenly over its Bq buckets; the environments of the
        // group own consecutive bucket ranges, so ONE histogram / scan / scatter sorts all of them and leaves every
        // environment contiguous.  Every element then ranks itself among the members of its own bucket on the exact key
        // (any bucket size is handled; clustered inputs just take longer).  The bucket counters are 16-bit halves of 32-bit
        // words (a group has at most 512 points), so 512 buckets cost 1 KB.
        const int n = fill, G = ngrp;
        int lf = lane;
        asm volatile("" : "+v"(lf));
        const int sh = G > 4 ? 3 : (G > 2 ? 2 : (G > 1 ? 1 : 0));
        const int Bq = kGBuckets >> sh;
        {
            reinterpret_cast<uint4*>(lds.hist)[lf] = make_uint4(0u, 0u, 0u, 0u);  // 256 words = 512 counters
            if (lf == 0) lds.hist[kGBuckets / 2] = 0u;
        }
        wave_sync_lds();
        constexpr int EPT = kGCap / 64;
        const double qs = (double)Bq / (thr2 * thr);
        uint64_t rk[EPT];
        uint32_t rp[EPT];  // category | env << 8 | bucket << 11 | (slot in bucket, then position, then rank) << 20
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            rk[q] = 0; rp[q] = 0;
            if (lf + 64 * q < n) {
                const int i = lf + 64 * q;
                const double d2 = u2d(lds.key[i]);
                const uint32_t vv = lds.val[i];
                const double d = sqrt(d2);  // utils.rs:1-8
                rk[q] = d2u(d);
                const double t = d2 * d * qs;
                const int b = (t < (double)Bq ? (int)t : Bq - 1) + (int)((vv >> 8) << (9 - sh));
                const uint32_t hsh = (uint32_t)(b & 1) << 4;
                const uint32_t old = atomicAdd(&lds.hist[b >> 1], 1u << hsh);
                rp[q] = vv | ((uint32_t)b << 11) | (((old >> hsh) & 0xFFFFu) << 20);
            }
        }
        wave_sync_lds();
        {   // exclusive scan of the bucket sizes: lane l owns buckets 8l .. 8l+7 (four words of two counters)
            uint4* h4 = reinterpret_cast<uint4*>(lds.hist);
            const uint4 a = h4[lf];
            const uint32_t c0 = a.x & 0xFFFFu, c1 = a.x >> 16, c2 = a.y & 0xFFFFu, c3 = a.y >> 16, c4 = a.z & 0xFFFFu, c5 = a.z >> 16,
                           c6 = a.w & 0xFFFFu, c7 = a.w >> 16;
            const uint32_t mine = (c0 + c1) + (c2 + c3) + (c4 + c5) + (c6 + c7);
            const uint32_t incl = wave_incl_scan_u32(mine);
            const uint32_t p0 = incl - mine, p1 = p0 + c0, p2 = p1 + c1, p3 = p2 + c2, p4 = p3 + c3, p5 = p4 + c4, p6 = p5 + c5, p7 = p6 + c6;
            h4[lf] = make_uint4(p0 | (p1 << 16), p2 | (p3 << 16), p4 | (p5 << 16), p6 | (p7 << 16));
            if (lf == 63) lds.hist[kGBuckets / 2] = incl;  // = n (low half: the "first slot" of the bucket past the last)
        }
        wave_sync_lds();
        const uint16_t* h16 = reinterpret_cast<const uint16_t*>(lds.hist);
#pragma unroll
        for (int q = 0; q < EPT; ++q) {  // group by bucket (arrival order inside a bucket)
            if (lf + 64 * q < n) {
                const uint32_t pos = (uint32_t)h16[(rp[q] >> 11) & 0x1FFu] + (rp[q] >> 20);
                lds.key[pos] = rk[q];
                rp[q] = (rp[q] & 0xFFFFFu) | (pos << 20);
            }
        }
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {  // rank among the members of the own bucket on the exact f64 key
            if (lf + 64 * q < n) {
                const uint32_t b = (rp[q] >> 11) & 0x1FFu, pos = rp[q] >> 20;
                const uint32_t s0 = h16[b], s1 = h16[b + 1];
                uint32_t rank = s0;
                for (uint32_t k = s0; k < s1; ++k) {
                    const uint64_t kj = lds.key[k];
                    rank += (kj < rk[q]) | ((kj == rk[q]) & (k < pos));
                }
                rp[q] = (rp[q] & 0xFFFFFu) | (rank << 20);
            }
        }
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            if (lf + 64 * q < n) {
                lds.key[rp[q] >> 20] = rk[q];
                lds.val[rp[q] >> 20] = (uint16_t)(rp[q] & 0x7FFu);
            }
        }
        wave_sync_lds();
        GSTAMP(2);
        // ------------------------------------------------------------------------------------------ keys + write-out
        // One pass: sorted distance -> F(distance) (single-weight-function configurations) -> global memory; the monotonicity
        // of the converted keys is checked on the way (F is monotone; its floating-point evaluation may produce a last-bit
        // inversion between neighbours, which the repair pass below removes with a running maximum -- rare).
        // Categories outside the map are reported HERE (pmf.rs:38-42 raises for a point of a used environment) and stored as 0.
        {
            const LCHD_AS4 EnvSide* p = opaque(ks);
            const LCHD_AS4 DevConfig* c = opaque(kc);
            uint64_t* __restrict__ okey = p->env.key;
            uint8_t* __restrict__ ocat = p->env.cat;
            const bool cdfk = p->env.cdf_keys != 0;
            const int n_cat = c->n_categories;
            const LCHD_AS4 WfEntry* wfe = as_const(c->wf);
            const int wkind = wfe->kind, wnp = wfe->n_params;
            const double* prm = c->wf_params + wfe->offset;
            const LCHD_AS4 double* cprm = as_const(prm);
            const double winv = *as_const(c->wf_inv);
            // uniform: (x_min, x_max); hyper_exp with one or two terms: (a_1, [a_2,] b_1, [b_2]) in scalar registers
            const bool w_uni = wkind == WF_UNIFORM, w_he = wkind == WF_HYPER_EXP && wnp <= 4;
            const double w0 = cprm[0], w1 = cprm[1], w2 = wnp > 2 ? cprm[2] : 0.0, w3 = wnp > 3 ? cprm[3] : 0.0;
            auto cdf = [&](double x) -> double {
                if (w_uni) {  // cdfs.rs:39-45
                    const double v = (x - w0) * winv;
                    return x < w0 ? 0.0 : (x > w1 ? 1.0 : v);
                }
                if (w_he) {  // cdfs.rs:5-21, same accumulation order
                    double sum;
                    if (wnp == 2) sum = 0.0 + w0 * exp_nonpos(-w1 * x);
                    else { sum = 0.0 + w0 * exp_nonpos(-w2 * x); sum += w1 * exp_nonpos(-w3 * x); }
                    return 1.0 - sum * winv;
                }
                return cdf_lean(wkind, prm, wnp, winv, x);
            };
            bool inv = false;
            double carry_f = 0.0;
            uint32_t carry_q = 0xFFu;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lf;
                const bool act = i < n;
                const int ic = act ? i : n - 1;
                const uint32_t vv = lds.val[ic];
                const uint32_t q = vv >> 8, cat = vv & 0xFFu;
                const double d = u2d(lds.key[ic]);
                const uint32_t st_q = lds.gstart[q];
                const double f = cdfk ? cdf(d) + 0.0 : d;
                double pf = wave_shr1_f64(f);
                uint32_t pq = (uint32_t)__builtin_amdgcn_update_dpp((int)q, (int)q, 0x138, 0xf, 0xf, false);  // wave_shr:1
                if (lf == 0) { pf = carry_f; pq = carry_q; }
                inv |= act && pq == q && f < pf;
                carry_f = readlane_f64(f, 63);
                carry_q = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
                if (act) {
                    const uint32_t o = (uint32_t)(((e_first + (int)q) << 9) + (i - (int)st_q));  // slot stride = kEnvGroupCap = 512; < 2^31 (host)
                    bad |= (int)cat >= n_cat;
                    okey[o] = d2u(f);
                    ocat[o] = (int)cat < n_cat ? (uint8_t)cat : (uint8_t)0;
                }
            }
            if (__builtin_amdgcn_ballot_w64(inv)) {
                // repair: F values into LDS, one lane per environment applies the running maximum, everything is written again
                for (int i = lf; i < n; i += 64) lds.key[i] = d2u(cdf(u2d(lds.key[i])) + 0.0);
                wave_sync_lds();
                if (lf < G) {
                    const int s0 = (int)lds.gstart[lf], c0 = (int)lds.gcount[lf];
                    uint64_t m = 0;
                    for (int i = s0; i < s0 + c0; ++i) { const uint64_t k = lds.key[i]; m = k > m ? k : m; lds.key[i] = m; }
                }
                wave_sync_lds();
                for (int i = lf; i < n; i += 64) {
                    const uint32_t q = (uint32_t)lds.val[i] >> 8;
                    okey[(uint32_t)(((e_first + (int)q) << 9) + (i - (int)lds.gstart[q]))] = lds.key[i];
                }
            }
            // prefix-count rows (EnvStore::pre): per environment an inclusive scan of the sorted points' one-hot category fields -- 8-bit
            // fields, one to four u64 words per row, one row per kPreStep points (rows 0, 4, 8, ...: the counts of the first 1, 5, 9, ...
            // points); the team sweeps read a lane's chunk-start counts from them (lchd_team_tile.h, PRE).
            // (An environment of more than 255 points overflows its fields: no team rule sweeps such a pair.)
            if (p->env.pre) {
                // lane l takes points kPreStep l .. kPreStep l + kPreStep - 1 of a round: ONE scan per count word over the lanes' sums serves
                // 64 rows (a scan per point, as round 5 wrote its rows, cost the environment kernel four times the vector instructions)
                uint64_t* __restrict__ opre = p->env.pre;
                const int nw = p->env.pre_words;  // 1 .. 4 (8 .. 32 category slots)
                for (int q = 0; q < G; ++q) {
                    const int s0 = (int)lds.gstart[q], c0 = (int)lds.gcount[q];
                    const uint32_t row0 = (uint32_t)((e_first + q) << 9) / (uint32_t)kPreStep;
                    uint64_t car[4] = {0ull, 0ull, 0ull, 0ull};
                    for (int i0 = 0; i0 < c0; i0 += 64 * kPreStep) {
                        const int i = i0 + lf * kPreStep;  // this lane's first point: the point of its row
                        uint64_t mine[4] = {0ull, 0ull, 0ull, 0ull}, first[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
                        for (int m = 0; m < kPreStep; ++m) {
                            const bool act = i + m < c0;
                            const uint32_t cat_raw = (uint32_t)lds.val[s0 + (act ? i + m : 0)] & 0xFFu;
                            const uint32_t cat = (int)cat_raw < n_cat ? cat_raw : 0u;  // (as stored)
                            const uint64_t one = act ? (1ull << ((cat & 7u) * 8u)) : 0ull;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const uint64_t o = (cat >> 3) == (uint32_t)k ? one : 0ull;
                                mine[k] += o;
                                if (m == 0) first[k] = o;
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < nw) {  // (wave-uniform)
                                const uint64_t incl = wave_incl_scan_fields(mine[k]);
                                const uint64_t row = incl - mine[k] + car[k] + first[k];  // the counts of the first i + 1 points
                                car[k] += readlane_u64(incl, 63);
                                if (i < c0) opre[(size_t)(row0 + (uint32_t)(i / kPreStep)) * (size_t)nw + k] = row;
                            }
                    }
                }
            }
            if (lf < G) {
                p->env.len[e_first + lf] = (int32_t)lds.gcount[lf];
#if LCHD_CAT0_STORE
                const uint32_t c0 = (uint32_t)lds.val[lds.gstart[lf]] & 0xFFu;  // (the sorted first point's category, as it was stored)
                if (p->env.cat0) p->env.cat0[e_first + lf] = (int)c0 < n_cat ? (uint8_t)c0 : (uint8_t)0;
#endif
            }
        }
        wave_sync_lds();  // (gstart / gcount / key are rewritten by the next group)
        fill = 0;
        ngrp = 0;
        GSTAMP(3);
    }
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) atomicOr(&st->flags, ST_BAD_CATEGORY);
}

bool launch_env_group(hipStream_t s, const DevConfig* cfg, bool tag_list, bool small_cap, const EnvSide& a, const EnvSide& b, double thr,
                      int anchors_per_wave, DeviceStatus* st) {
    if (a.max_envs + b.max_envs <= 0) return true;
    // slots of exactly kEnvGroupCap points, 32-bit element offsets into the store and 32-bit byte offsets into the record arrays
    if (anchors_per_wave < 1 || a.env.stride != kEnvGroupCap || (b.max_envs > 0 && b.env.stride != kEnvGroupCap)) return false;
    if (a.max_envs >= (1 << 22) || b.max_envs >= (1 << 22) || a.c.n >= (1 << 27) || b.c.n >= (1 << 27)) return false;
    EnvSides sides;
    sides.s[0] = a;
    sides.s[1] = b;
    const int64_t nwa = (a.max_envs + anchors_per_wave - 1) / anchors_per_wave, nwb = (b.max_envs + anchors_per_wave - 1) / anchors_per_wave;
    constexpr int WPB = LCHD_GROUP_WPB;
    const dim3 grid((unsigned)((nwa + nwb + WPB - 1) / WPB));
    if (small_cap) {
        if (tag_list) k_env_group<true, kEnvGroupCapSmall, 6><<<grid, 64 * WPB, 0, s>>>(cfg, sides, thr, anchors_per_wave, (int)nwa, st);
        else k_env_group<false, kEnvGroupCapSmall, 6><<<grid, 64 * WPB, 0, s>>>(cfg, sides, thr, anchors_per_wave, (int)nwa, st);
    } else {
        if (tag_list) k_env_group<true, kEnvGroupCap, 5><<<grid, 64 * WPB, 0, s>>>(cfg, sides, thr, anchors_per_wave, (int)nwa, st);
        else k_env_group<false, kEnvGroupCap, 5><<<grid, 64 * WPB, 0, s>>>(cfg, sides, thr, anchors_per_wave, (int)nwa, st);
    }
    return true;
}

}  // namespace lchd

#ifdef LCHD_SWEEP_STAMPS
extern "C" int lchd_debug_envg_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lchd::g_envg_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(lchd::g_envg_stamps), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

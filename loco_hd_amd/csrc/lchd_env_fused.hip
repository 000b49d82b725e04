// lchd_env_fused.hip -- K1 + K2 in one kernel for environments that are used ONCE.
//
// The reference builds and consumes an environment inside one closure (env_from_idx + stat_dist_integral,
// /root/reference/src/locohd.rs:514-554): radius query, tag filter, distances, sort, sweep -- per anchor pair, per side.  The regular
// pipeline here builds every UNIQUE anchor's environment once (k_env_group), writes it to the environment store and lets the team sweeps
// (k_sweep_duo) read it back for every pair that uses it: the right trade when anchors are re-used (C2a: 100 pairs per anchor), a
// detour when they are not -- the frames of a trajectory, (i, i) lists over two structures, a rank's side-B partners under strong
// scaling: there the store is written once and read once, the pair record (k_pair_meta) describes one use, and the sweep stages what
// the environment kernel had in LDS a moment ago.
//
// k_env_sweep: a wavefront takes TEAMS consecutive anchor pairs.  For their side-B anchors it runs k_env_group's set-up, radius search
// and group sort (same arithmetic, same LDS layout: lchd_env_group.hip), converts the sorted distances to F(distance) IN LDS, stages
// the pairs' side-A environments (built by k_env_group before, kept in the store: side A is typically the re-used side) next to them
// and runs the team tile (lchd_team_tile.h) at once: no store write, no pair record read, no second staging of list B.  There is no
// de-duplication of side B at all -- a side-B anchor that occurs in several pairs is simply built several times, exactly like the
// reference does; the host only takes this kernel when (almost) every side-B anchor of the previous pass was unique.
//
// Pairs the team tile cannot take (more merged events than the tile, an environment beyond 255 points under the 8-bit rule) get their
// side-B environment written to slot p of side B's store and are swept by the INDIRECT k_sweep behind this kernel, which walks the
// pair records this kernel writes for EVERY pair (16 bytes each).  An environment beyond the group buffer is reported as
// ST_ENV_OVERFLOW: the host repeats the pass on the regular pipeline.
#include "lchd_kcommon.h"
#include "lchd_team_tile.h"

#ifndef LCHD_FUSED_U
#define LCHD_FUSED_U 4   // search steps (64 candidates each) whose record loads are issued together
#endif

namespace lchd {

#define LCHD_AS4 __attribute__((address_space(4)))

constexpr int kFGBuckets = 512;  // distance buckets of a group's sort, 16-bit counters
constexpr int kFGTab = 224;      // candidate groups (8 records each) per anchor

// One wavefront's LDS.  GCAP: points of one group (flat buffer) = the largest environment the instantiation handles;
// APOOL: staged side-A points of the wavefront's pairs (keys), shared by its teams.
template <int GCAP, int APOOL, int TEAMS>
struct FusedLds {
    uint64_t key[GCAP + 2];            // d^2 while a group is being collected, then sorted distances, then F(distance) (+ the head re-read's spare entries)
    uint16_t val[GCAP];                // category | environment-in-group << 8
    uint8_t cat8[GCAP + 8];            // categories of the sorted points (what the tile reads)
    union {
        struct {
            uint32_t hist[kFGBuckets / 2 + 4];  // two 16-bit bucket counters per word (+ the end marker)
            uint32_t tab[2][kFGTab];            // (byte offset of the first record) | records, per half-wave anchor
        } srch;                                  // set-up, search and sort ...
        uint64_t a_key[APOOL + 4 * TEAMS];       // ... then the staged side-A lists of the sub-round's pairs
    } u;
    uint8_t a_cat[APOOL + 8 * TEAMS + 8];
    uint32_t gstart[TEAMS], gcount[TEAMS];
    // per pair of the round (written by lane t of phase 0, read by everybody)
    double ax[TEAMS], ay[TEAMS], az[TEAMS];
    uint32_t atag[TEAMS], apos32[TEAMS];
    int32_t asid[TEAMS], slot_a[TEAMS], n_a[TEAMS], c0a[TEAMS], state[TEAMS];  // state: 0 unusable, 1 to be searched
};

// inclusive prefix sum inside each 32-lane half
__device__ __forceinline__ uint32_t fhalf_scan_u32(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    return (uint32_t)v;
}
template <class T>
__device__ __forceinline__ const LCHD_AS4 T* fopaque(const LCHD_AS4 T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
template <class T>
__device__ __forceinline__ const LCHD_AS4 T* fas_const(const T* p) {
    return (const LCHD_AS4 T*)(unsigned long long)p;
}

#ifdef LCHD_FUSED_STAMPS
__device__ unsigned long long g_fused_stamps[16];
#define FSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && (blockIdx.x & 31) == 0) atomicAdd(&g_fused_stamps[i], t_ - fstamp_last); fstamp_last = t_; } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif
constexpr int kFusedCapSmall = 384;  // group buffer of the four-pairs form (four environments of ~70-100 points: 320 closed one sub-round in five early)
constexpr int kFusedWPB = 4;  // independent wavefronts per workgroup (they share the square-root tables; one barrier, at the start)

template <bool TAGLIST, int CMAX, int TL, int TILE_, int GCAP, int APOOL, int WAVES>
__global__ __launch_bounds__(64 * kFusedWPB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_env_sweep(FusedArgs fa) {
    using TT = TeamTile<CMAX, TL, TILE_, false, false>;
    constexpr int TEAMS = TT::TEAMS, NT = TT::NT, LW = TT::LW;
    constexpr bool LCNT = TT::LCNT;
    constexpr int RULE = TILE_ == kDuoTile ? 0 : 2;
    static_assert(GCAP % 64 == 0 && GCAP <= kEnvGroupCap, "whole wavefronts; environment slots hold kEnvGroupCap points");
    static_assert(APOOL >= TILE_ + 2, "one pair's side-A list always fits the pool");
    static_assert(TEAMS == 2 || TEAMS == 4, "the set-up handles two anchors per pass");
    using Lds = FusedLds<GCAP, APOOL, TEAMS>;
    __shared__ double t_sqrt[NT], t_rsqrt[NT];
    __shared__ __attribute__((aligned(16))) Lds lds_all[kFusedWPB];
    __shared__ uint64_t lc_[LCNT ? kFusedWPB : 1][LCNT ? LW * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, tl = lane & (TL - 1), team = lane / TL;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < NT; k += 64 * kFusedWPB) {
        t_sqrt[k] = fa.sqrt_tab[k];
        t_rsqrt[k] = fa.rsqrt_tab[k];
    }
    __syncthreads();
    Lds& lds = lds_all[wv];
    unsigned char* lcl = reinterpret_cast<unsigned char*>(lc_[LCNT ? wv : 0]) + lane * 8;
    const LCHD_AS4 FusedArgs* const ka = (const LCHD_AS4 FusedArgs*)((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr());
    const LCHD_AS4 DevConfig* const kc = fas_const(fa.cfg);
    const double thr = fa.thr, thr2 = thr * thr;
    const bool accept_same = kc->tag_accept_same != 0;
    DevConfig tcfg{};  // the tag-rule words only (TAGLIST)
    if constexpr (TAGLIST) {
        tcfg.tag_mode = kc->tag_mode; tcfg.tag_accept_same = kc->tag_accept_same; tcfg.tag_accepted_pairs = kc->tag_accepted_pairs;
        tcfg.tag_ordered = kc->tag_ordered; tcfg.n_tag_pairs = kc->n_tag_pairs; tcfg.tag_pairs = kc->tag_pairs;
    }
    const char* const recb = reinterpret_cast<const char*>(fa.b.g.rec);
    const uint32_t sub32 = (uint32_t)(lane & 7) << 5, subc = (uint32_t)(lane & 7);
    const double Finf0 = kc->wf_finf[0];
    const int64_t n_pairs = fa.n_pairs;
    bool bad = false;
    unsigned long long n_taken = 0;  // pairs swept here (lane TL - 1 of each team counts its own)
    int biggest = 0;

#ifdef LCHD_FUSED_STAMPS
    unsigned long long fstamp_last = __builtin_amdgcn_s_memtime();
#endif
    const int64_t wid = (int64_t)blockIdx.x * kFusedWPB + wv, nwv = (int64_t)gridDim.x * kFusedWPB;
    for (int64_t pb = wid * TEAMS; pb < n_pairs; pb += nwv * TEAMS) {
        // ------------------------------------------------------------------------------------------ phase 0: the round's pairs
        wave_sync_lds();  // (the previous round's tile has read gstart / a_key / the per-pair words)
        if (lane < TEAMS) {
            const LCHD_AS4 FusedArgs* p = fopaque(ka);
            const int64_t pp = pb + lane;
            int st_ = 0, sl = 0, na = 0, c0 = 0, sid = 0;
            double x = 0.0, y = 0.0, z = 0.0;
            uint32_t tg = 0u, ap = 0u;
            if (pp < n_pairs) {
                const longlong2 ab = reinterpret_cast<const longlong2*>(p->anchors)[pp];
                if (ab.x < 0 || ab.y < 0 || ab.x >= p->n_atoms_a || ab.y >= p->n_atoms_b) {
                    atomicOr(&p->st->flags, ST_BAD_ANCHOR);
                } else {
                    sl = (int)p->slot_a[ab.x];
                    na = p->env_a.len[sl];
                    c0 = p->env_a.cat0 ? (int)p->env_a.cat0[sl] : (int)p->env_a.cat[(uint64_t)(uint32_t)sl * (uint32_t)p->env_a.stride];
                    x = p->b.c.x[ab.y]; y = p->b.c.y[ab.y]; z = p->b.c.z[ab.y];
                    tg = (uint32_t)p->b.c.tag[ab.y];
                    ap = p->b.g.pos_of[ab.y] << 5;
                    sid = p->b.c.sid ? p->b.c.sid[ab.y] : 0;
                    st_ = na > 0 ? 1 : 0;  // (an empty / overflowed side-A environment was flagged where it was built)
                }
            }
            lds.ax[lane] = x; lds.ay[lane] = y; lds.az[lane] = z;
            lds.atag[lane] = tg; lds.apos32[lane] = ap; lds.asid[lane] = sid;
            lds.slot_a[lane] = sl; lds.n_a[lane] = na; lds.c0a[lane] = c0; lds.state[lane] = st_;
        }
        wave_sync_lds();
        FSTAMP(0);

        // the sub-round being collected: teams [t0, j) are in the group buffer (all wave-uniform)
        int fill = 0, ngrp = 0, t0 = 0, a_fill = 0;
        int setup_base = -2;
        double v_ax = 0.0, v_ay = 0.0, v_az = 0.0;
        uint32_t v_tag = 0u, v_apos32 = 0u;
        int ng_h0 = 0, ng_h1 = 0;
        int j = 0;
        while (true) {
            bool do_flush = false;
            if (j < TEAMS) {
                const int stj = __builtin_amdgcn_readfirstlane(lds.state[j]);
                if (stj == 0) {  // unusable pair (bad anchor, no side-A environment): NaN, and an empty slot in the sub-round
                    if (lane == 0) { lds.gstart[j - t0] = (uint32_t)fill; lds.gcount[j - t0] = 0u; }
                    ++ngrp; ++j;
                    do_flush = j == TEAMS;
                    if (!do_flush) continue;
                } else {
                if (j >= setup_base + 2 || j < setup_base) {
                    // ------------------------------------------------------------------------------ set-up of anchors j, j + 1
                    const LCHD_AS4 FusedArgs* p = fopaque(ka);
                    const double gmin0 = p->b.g.min[0], gmin1 = p->b.g.min[1], gmin2 = p->b.g.min[2];
                    const double ginv0 = p->b.g.inv[0], ginv1 = p->b.g.inv[1], ginv2 = p->b.g.inv[2];
                    const double gcell0 = p->b.g.cell[0], gcell1 = p->b.g.cell[1], gcell2 = p->b.g.cell[2];
                    const int dim0 = p->b.g.dim[0], dim1 = p->b.g.dim[1], dim2 = p->b.g.dim[2];
                    const uint32_t* __restrict__ cell_start = p->b.g.cell_start;
                    const double thr2m = thr2 * (1.0 + 1e-6);
                    int ls = lane;
                    asm volatile("" : "+v"(ls));
                    const int h = ls >> 5, r = ls & 31;
                    const int jt = min(j + h, TEAMS - 1);
                    const double a_x = lds.ax[jt], a_y = lds.ay[jt], a_z = lds.az[jt];
                    const int a_sid = lds.asid[jt];
                    const bool a_on = (j + h < TEAMS) && lds.state[jt] != 0;
                    v_ax = a_x; v_ay = a_y; v_az = a_z; v_tag = lds.atag[jt]; v_apos32 = lds.apos32[jt];
                    const int cx = cell_coord(a_x, gmin0, ginv0, dim0);
                    const int cy = cell_coord(a_y, gmin1, ginv1, dim1);
                    const int cz = cell_coord(a_z, gmin2, ginv2, dim2);
                    const double fx = (a_x - gmin0) * ginv0 - (double)cx, fy = (a_y - gmin1) * ginv1 - (double)cy,
                                 fz = (a_z - gmin2) * ginv2 - (double)cz;
                    const int rr = r < 25 ? r : 24;
                    const int kz = (rr * 13) >> 6, ky = rr - 5 * kz;  // rr / 5, rr % 5
                    const int oy = ky - 2, oz = kz - 2;
                    const int yy = cy + oy, zz = cz + oz;
                    // cells that lie wholly outside the radius are skipped (k_env_group: relative margin 1e-6 on thr^2)
                    const double ty = oy < 0 ? fy - (double)(oy + 1) : (1.0 - fy) + (double)(oy - 1);
                    const double tz = oz < 0 ? fz - (double)(oz + 1) : (1.0 - fz) + (double)(oz - 1);
                    const double gy = oy == 0 ? 0.0 : fmax(ty, 0.0) * gcell1, gz = oz == 0 ? 0.0 : fmax(tz, 0.0) * gcell2;
                    const double r2 = gy * gy + gz * gz;
                    const double fxl = fmax(fx, 0.0), fxh = fmax(1.0 - fx, 0.0);
                    const double xm1 = fxl * gcell0, xm2 = (fxl + 1.0) * gcell0, xp1 = fxh * gcell0, xp2 = (fxh + 1.0) * gcell0;
                    const int lo = (r2 + xm2 * xm2 < thr2m) ? -2 : ((r2 + xm1 * xm1 < thr2m) ? -1 : 0);
                    const int hi = (r2 + xp2 * xp2 < thr2m) ? 2 : ((r2 + xp1 * xp1 < thr2m) ? 1 : 0);
                    const int xl = max(cx + lo, 0), xh = min(cx + hi, dim0 - 1);
                    const bool in = (r < 25) & a_on & ((unsigned)zz < (unsigned)dim2) & ((unsigned)yy < (unsigned)dim1) & (r2 < thr2m);
                    const int row = in ? ((a_sid * dim2 + zz) * dim1 + yy) * dim0 : 0;
                    const int b_ = (int)cell_start[row + xl], e_ = (int)cell_start[row + xh + 1];
                    const int len = in ? e_ - b_ : 0;
                    const uint32_t ng = (uint32_t)(len + 7) >> 3;
                    const uint32_t incl = fhalf_scan_u32(ng);
                    const uint32_t goff = incl - ng;
                    ng_h0 = __builtin_amdgcn_readlane((int)incl, 31);
                    ng_h1 = __builtin_amdgcn_readlane((int)incl, 63);
                    uint32_t* tb = lds.u.srch.tab[h];
                    wave_sync_lds();
                    {
                        uint32_t ent = ((uint32_t)b_ << 5) | 8u;
                        int left = len;
                        uint32_t at = goff;
                        while (__builtin_amdgcn_ballot_w64(left > 0)) {
                            if (left > 0 && at < (uint32_t)kFGTab) tb[at] = left >= 8 ? ent : (ent & ~15u) | (uint32_t)left;
                            ent += 8u << 5;
                            left -= 8;
                            ++at;
                        }
                        const uint32_t ngh = h ? (uint32_t)ng_h1 : (uint32_t)ng_h0;
                        if (ngh + (uint32_t)r < (uint32_t)kFGTab) tb[ngh + (uint32_t)r] = 0u;
                    }
                    wave_sync_lds();
                    setup_base = j;
                    FSTAMP(1);
                }
                const int hh = j - setup_base;  // 0 or 1
                const int NG = hh ? ng_h1 : ng_h0;
                const int nA_j = __builtin_amdgcn_readfirstlane(lds.n_a[j]);
                // (the pair's side-A list needs room in the pool: an even start, one spare entry behind it)
                const int a_need = ((nA_j - 1 + 1) & ~1) + 2;
                if (NG > kFGTab) {
                    // more candidate groups than the table holds: the host repeats the pass on the regular pipeline (whose kernels grow)
                    if (lane == 0) {
                        const LCHD_AS4 FusedArgs* p = fopaque(ka);
                        atomicOr(&p->st->flags, ST_ENV_OVERFLOW);
                        atomicMax(&p->st->max_env, (uint32_t)(kEnvGroupCap + 1));
                        atomicMax(&p->st->max_bound, (uint32_t)NG * 8u);
                        lds.state[j] = 0;
                        lds.gstart[j - t0] = (uint32_t)fill; lds.gcount[j - t0] = 0u;
                    }
                    ++ngrp; ++j;
                    do_flush = j == TEAMS;
                    if (!do_flush) continue;
                } else if (ngrp > 0 && a_fill + a_need > APOOL) {
                    // (no early closing on an ESTIMATE of the environment's size, unlike k_env_group: a sub-round of fewer than TEAMS
                    //  pairs runs the tile with idle teams -- an occasional wasted search costs less)
                    do_flush = true;  // (the anchor is searched after the flush, into an empty buffer)
                } else {
                    // ------------------------------------------------------------------------------ radius search of anchor j
                    const int src = 32 * hh;
                    const double ax = readlane_f64(v_ax, src), ay = readlane_f64(v_ay, src), az = readlane_f64(v_az, src);
                    const int32_t atag = __builtin_amdgcn_readlane((int)v_tag, src);
                    const uint32_t apos32 = (uint32_t)__builtin_amdgcn_readlane((int)v_apos32, src);
                    const uint32_t* tb = lds.u.srch.tab[hh] + (lane >> 3);
                    const uint32_t qbits = (uint32_t)ngrp << 8;
                    auto tag_ok = [&](int32_t t_other) -> bool {  // tag_pairing_rule.rs:49-75
                        if constexpr (TAGLIST) return tag_pair_accepted(tcfg, atag, t_other);
                        else return (atag == t_other) == accept_same;
                    };
                    int count = 0;
                    constexpr int U = LCHD_FUSED_U;
                    static_assert(8 * U <= 32, "table padding covers one round of 8 U groups");
                    for (int g0 = 0; g0 < NG; g0 += 8 * U) {
                        uint32_t off[U], ent[U];
                        double2 R0[U], R1[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            ent[u] = tb[g0 + 8 * u];
                            off[u] = (ent[u] & ~31u) + sub32;
                            R0[u] = *reinterpret_cast<const double2*>(recb + off[u]);
                            R1[u] = *reinterpret_cast<const double2*>(recb + off[u] + 16);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (g0 + 8 * u < NG) {  // wave-uniform
                                const double dx = R0[u].x - ax, dy = R0[u].y - ay, dz = R1[u].x - az;
                                double d2 = dx * dx;   // -ffp-contract=off: the reference's `distance += diff * diff` (utils.rs:1-8)
                                d2 = d2 + dy * dy;
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
                                    if (pos < GCAP) {
                                        lds.key[pos] = d2u(d2);
                                        lds.val[pos] = (uint16_t)(((uint32_t)(tc >> 32) & 0xFFu) | qbits);
                                    }
                                }
                                count += __popcll(m);
                            }
                        }
                    }
                    FSTAMP(2);
                    if (fill + count > GCAP) {
                        if (ngrp == 0) {  // this environment alone is too large for the group buffer: the regular pipeline takes the pass
                            if (lane == 0) {
                                const LCHD_AS4 FusedArgs* p = fopaque(ka);
                                atomicOr(&p->st->flags, ST_ENV_OVERFLOW);
                                atomicMax(&p->st->max_env, (uint32_t)count);
                                lds.state[j] = 0;
                                lds.gstart[0] = 0u; lds.gcount[0] = 0u;
                            }
                            ++ngrp; ++j;
                            do_flush = j == TEAMS;
                            if (!do_flush) continue;
                        } else {
                            do_flush = true;  // close the sub-round; the anchor is searched again into the empty buffer
                        }
                    } else {
                        if (count == 0 && lane == 0) { atomicOr(&fopaque(ka)->st->flags, ST_EMPTY_ENV); lds.state[j] = 0; }
                        if (lane == 0) { lds.gstart[ngrp] = (uint32_t)fill; lds.gcount[ngrp] = (uint32_t)count; }
                        fill += count;
                        a_fill += a_need;
                        ++ngrp;
                        ++j;
                        do_flush = j == TEAMS;
                    }
                }
                }
            } else {
                if (ngrp == 0) break;
                do_flush = true;
            }
            if (!do_flush || ngrp == 0) continue;

            // -------------------------------------------------------------------------------------- sort of the group (k_env_group's)
            const int n = fill, G = ngrp;
            int lf = lane;
            asm volatile("" : "+v"(lf));
            if (n > 0) {
                const int sh = G > 2 ? 2 : (G > 1 ? 1 : 0);
                const int Bq = kFGBuckets >> sh;
                {
                    reinterpret_cast<uint4*>(lds.u.srch.hist)[lf] = make_uint4(0u, 0u, 0u, 0u);
                    if (lf == 0) lds.u.srch.hist[kFGBuckets / 2] = 0u;
                }
                wave_sync_lds();
                constexpr int EPT = GCAP / 64;
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
                        const uint32_t old = atomicAdd(&lds.u.srch.hist[b >> 1], 1u << hsh);
                        rp[q] = vv | ((uint32_t)b << 11) | (((old >> hsh) & 0xFFFFu) << 20);
                    }
                }
                wave_sync_lds();
                {
                    uint4* h4 = reinterpret_cast<uint4*>(lds.u.srch.hist);
                    const uint4 a = h4[lf];
                    const uint32_t c0 = a.x & 0xFFFFu, c1 = a.x >> 16, c2 = a.y & 0xFFFFu, c3 = a.y >> 16, c4 = a.z & 0xFFFFu, c5 = a.z >> 16,
                                   c6 = a.w & 0xFFFFu, c7 = a.w >> 16;
                    const uint32_t mine = (c0 + c1) + (c2 + c3) + (c4 + c5) + (c6 + c7);
                    const uint32_t incl = wave_incl_scan_u32(mine);
                    const uint32_t p0 = incl - mine, p1 = p0 + c0, p2 = p1 + c1, p3 = p2 + c2, p4 = p3 + c3, p5 = p4 + c4, p6 = p5 + c5, p7 = p6 + c6;
                    h4[lf] = make_uint4(p0 | (p1 << 16), p2 | (p3 << 16), p4 | (p5 << 16), p6 | (p7 << 16));
                    if (lf == 63) lds.u.srch.hist[kFGBuckets / 2] = incl;
                }
                wave_sync_lds();
                const uint16_t* h16 = reinterpret_cast<const uint16_t*>(lds.u.srch.hist);
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    if (lf + 64 * q < n) {
                        const uint32_t pos = (uint32_t)h16[(rp[q] >> 11) & 0x1FFu] + (rp[q] >> 20);
                        lds.key[pos] = rk[q];
                        rp[q] = (rp[q] & 0xFFFFFu) | (pos << 20);
                    }
                }
                wave_sync_lds();
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
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
                FSTAMP(3);
                // ---------------------------------------------------------------------------------- F(distance) keys, in place
                {
                    const LCHD_AS4 DevConfig* c = fopaque(kc);
                    const int n_cat = c->n_categories;
                    const LCHD_AS4 WfEntry* wfe = fas_const(c->wf);
                    const int wkind = wfe->kind, wnp = wfe->n_params;
                    const double* prm = c->wf_params + wfe->offset;
                    const LCHD_AS4 double* cprm = fas_const(prm);
                    const double winv = *fas_const(c->wf_inv);
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
                        const double f = cdf(u2d(lds.key[ic])) + 0.0;
                        double pf = wave_shr1_f64(f);
                        uint32_t pq = (uint32_t)__builtin_amdgcn_update_dpp((int)q, (int)q, 0x138, 0xf, 0xf, false);  // wave_shr:1
                        if (lf == 0) { pf = carry_f; pq = carry_q; }
                        inv |= act && pq == q && f < pf;
                        carry_f = readlane_f64(f, 63);
                        carry_q = (uint32_t)__builtin_amdgcn_readlane((int)q, 63);
                        if (act) {
                            bad |= (int)cat >= n_cat;
                            lds.key[i] = d2u(f);
                            lds.cat8[i] = (int)cat < n_cat ? (uint8_t)cat : (uint8_t)0;
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(inv)) {  // repair: one lane per environment applies the running maximum (rare)
                        wave_sync_lds();
                        if (lf < G) {
                            const int s0 = (int)lds.gstart[lf], c0 = (int)lds.gcount[lf];
                            uint64_t m = 0;
                            for (int i = s0; i < s0 + c0; ++i) { const uint64_t k = lds.key[i]; m = k > m ? k : m; lds.key[i] = m; }
                        }
                    }
                }
                wave_sync_lds();
            }
            FSTAMP(4);
            // -------------------------------------------------------------------------------------- the sub-round's pairs
            // team q of the wavefront sweeps pair t0 + q of the round (teams beyond the sub-round idle)
            {
                const LCHD_AS4 FusedArgs* pa = fopaque(ka);
                const int q = team - t0;                     // this team's environment in the group
                const bool in_sub = q >= 0 && q < G;
                const int qq = in_sub ? q : 0;
                const int64_t p = pb + team;
                const bool live = in_sub && p < n_pairs;
                const int stt = lds.state[team];
                const int nB = live ? (int)lds.gcount[qq] : 0, gs = (int)lds.gstart[qq];
                const int nA = live ? lds.n_a[team] : 0;
                const bool usable = live && stt != 0 && nA > 0 && nB > 0;
                const bool small = usable && pair_is_small(RULE, nA, nB);
                const int slA = lds.slot_a[team];
                const int c0a = lds.c0a[team];
                const int c0b = usable ? (int)lds.cat8[gs] : 0;
                const int mA = small ? nA - 1 : 0, mB = small ? nB - 1 : 0, T = mA + mB;
                biggest = max(biggest, max(nA, nB));
                // the pair record (every pair: the INDIRECT sweep behind this kernel walks them all) and the length of slot p
                if (tl == 0 && live) {
                    pa->meta[p] = usable ? make_int4(slA, (int)p, nA | (c0a << 24), nB | (c0b << 24)) : make_int4(slA, (int)p, 0, 0);
                    pa->b.env.len[p] = usable ? nB : 0;
                }
                // pairs the tile cannot take: their side-B environment goes to slot p of side B's store (wave-uniform loop, rare)
                {
                    unsigned long long left = __builtin_amdgcn_ballot_w64(usable && !small && tl == 0);
                    while (left) {
                        const int ln = __ffsll((long long)left) - 1;
                        left &= left - 1;
                        const int tq = (ln / TL) - t0;
                        const int s0 = (int)lds.gstart[tq], c0 = (int)lds.gcount[tq];
                        const uint64_t base = (uint64_t)(pb + ln / TL) << 9;  // slot stride = kEnvGroupCap = 512
                        for (int i = lane; i < c0; i += 64) {
                            pa->b.env.key[base + i] = lds.key[s0 + i];
                            pa->b.env.cat[base + i] = lds.cat8[s0 + i];
                        }
                        if (lane == 0 && pa->b.env.cat0) pa->b.env.cat0[pb + ln / TL] = lds.cat8[s0];
                    }
                }
                // stage list A of the small pairs into the pool (the search tables are dead: the pool overlays them)
                const uint64_t offA = (uint64_t)(uint32_t)slA * (uint32_t)pa->env_a.stride;
                const uint64_t* __restrict__ kA = pa->env_a.key + offA;
                const uint8_t* __restrict__ tA = pa->env_a.cat + offA;
                const int mAe = (mA + 1) & ~1;
                // pool offsets: an exclusive scan of the teams' needs (even starts; one spare entry each)
                int a_off = 0;
                {
                    const int need = small ? mAe + 2 : 0;
#pragma unroll
                    for (int k = 0; k < TEAMS; ++k) {
                        const int nk = __builtin_amdgcn_readlane(need, k * TL);
                        a_off += (k < team) ? nk : 0;
                    }
                }
                uint64_t* sA = lds.u.a_key + a_off;
                uint8_t* cA = lds.a_cat + a_off;
                const uint64_t* sB = lds.key + gs + 1;
                const uint8_t* cB = lds.cat8 + gs + 1;
                const double F0 = small ? u2d(lds.key[gs]) : 0.0;  // F(0): both anchors sit at distance 0
                const int epl = (T + TL - 1) / TL;
                int epl_w = __builtin_amdgcn_readlane(epl, 0);
#pragma unroll
                for (int k = 1; k < TEAMS; ++k) epl_w = max(epl_w, __builtin_amdgcn_readlane(epl, k * TL));
                wave_sync_lds();  // (everybody has read the search tables / histogram for the last time)
                {
                    constexpr int EPL2 = (TT::EPL + 1) / 2;
                    typedef unsigned long long __attribute__((ext_vector_type(2), aligned(8))) key2_t;
                    const int epl2 = (mAe + 2 * TL - 1) / (2 * TL);
                    int epl2_w = __builtin_amdgcn_readlane(epl2, 0);
#pragma unroll
                    for (int k = 1; k < TEAMS; ++k) epl2_w = max(epl2_w, __builtin_amdgcn_readlane(epl2, k * TL));
                    key2_t rk2[EPL2];
                    uint32_t rc2[EPL2];
#pragma unroll
                    for (int u = 0; u < EPL2; ++u) { rk2[u] = key2_t{0ull, 0ull}; rc2[u] = 0u; }
                    if (small) {
#pragma unroll
                        for (int u = 0; u < EPL2; ++u) {
                            if (u < epl2_w) {
                                const int t0_ = 2 * (tl + TL * u);
                                const int tt = t0_ < mAe ? t0_ : 0;  // (beyond the list: re-read its first pair, nothing is written)
                                rk2[u] = *reinterpret_cast<const key2_t*>(kA + 1 + tt);
                                rc2[u] = (uint32_t)tA[1 + tt] | ((uint32_t)tA[2 + tt] << 8);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EPL2; ++u) {
                        if (u < epl2_w) {
                            const int t0_ = 2 * (tl + TL * u);
                            if (small && t0_ < mAe) {
                                *reinterpret_cast<ulonglong2*>(sA + t0_) = ulonglong2{rk2[u].x, rk2[u].y};
                                *reinterpret_cast<uint16_t*>(cA + t0_) = (uint16_t)rc2[u];
                            }
                        }
                    }
                }
                wave_sync_lds();
                FSTAMP(5);
                const double acc = TT::run(sA, cA, sB, cB, mA, mB, T, epl, epl_w, c0a, c0b, F0, Finf0, t_sqrt, t_rsqrt, nullptr, lcl, tl);
                if (tl == TL - 1 && live) {
                    if (small) { pa->out[p] = acc; ++n_taken; }
                    else if (!usable) pa->out[p] = nan("");
                }
            }
            wave_sync_lds();  // (gstart / gcount / key / the pool are rewritten by the next sub-round)
            FSTAMP(6);
#ifdef LCHD_FUSED_STAMPS
            if ((threadIdx.x & 63) == 0 && (blockIdx.x & 31) == 0) { atomicAdd(&g_fused_stamps[8], 1ull); atomicAdd(&g_fused_stamps[9], (unsigned long long)G); }
#endif
            t0 = j;
            fill = 0;
            ngrp = 0;
            a_fill = 0;
            setup_base = -2 - TEAMS;  // (the pool overlaid the search tables: the next anchor is set up again)
        }
    }
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) atomicOr(&fa.st->flags, ST_BAD_CATEGORY);
    // what the host wants to know about the pass: pairs swept here, the largest environment (sharded accumulators: DoneState)
    {
        unsigned long long tk = n_taken;
        for (int m = 32; m > 0; m >>= 1) { tk += shfl_u64(tk, lane ^ m); biggest = max(biggest, __shfl_xor(biggest, m)); }
        if (lane == 0) {
            const uint32_t g = (uint32_t)(wid & 63);
            atomicAdd(&fa.done->acc_sum[g * 16], tk);
            atomicMax(&fa.done->acc_max[g * 32], (uint32_t)biggest);
        }
    }
}

// behind the fused kernel (and its companion): the status hand-over k_pair_meta's last workgroup does in a regular pass
__global__ __launch_bounds__(64) void k_fused_publish(FusedArgs fa, HostStatus* hst, uint32_t seq) {
    const int k = threadIdx.x;
    unsigned long long v = atomicExch(&fa.done->acc_sum[k * 16], 0ull);
    uint32_t mx = atomicExch(&fa.done->acc_max[k * 32], 0u);
    for (int m = 32; m > 0; m >>= 1) { v += shfl_u64(v, k ^ m); mx = max(mx, (uint32_t)__shfl_xor((int)mx, m)); }
    if (k != 0) return;
    DeviceStatus* st = fa.st;
    st->n_small = v;
    st->n_c8 = v;
    const uint32_t over = __hip_atomic_load(&st->max_env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hst->flags = __hip_atomic_load(&st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hst->max_env = over > mx ? over : mx;
    hst->n_unique[0] = st->n_unique[0];
    hst->n_unique[1] = 0u;
    hst->n_small = v;
    hst->n_duo = v;
    hst->n_c8 = v;
    hst->n_overflow[0] = st->n_overflow[0];
    hst->n_overflow[1] = 0u;
    hst->max_bound = __hip_atomic_load(&st->max_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    hst->n_dup_b = 0u;
    hst->snapshot_seq = seq;
    st->flags = 0u;
    st->max_env = 0u;
    st->n_overflow[0] = 0u;
    st->n_overflow[1] = 0u;
    st->max_bound = 0u;
    st->n_dup_b = 0u;
}

template <bool TAGLIST, int CMAX>
static void launch_fused_c(hipStream_t s, int rule, unsigned grid, const FusedArgs& fa) {
    // rule 0: four pairs of at most 240 merged events per wavefront, group buffer of 384 points (coarse-grained typing, ~70-100
    // points per environment); rule 2: two pairs of 8-bit-count environments, at most 480 events, group buffer of 512 points
    constexpr int W = CMAX <= 16 ? 4 : 3;
    if (rule == 0) k_env_sweep<TAGLIST, CMAX, 16, kDuoTile, kFusedCapSmall, 384, W><<<grid, 64 * kFusedWPB, 0, s>>>(fa);
    else k_env_sweep<TAGLIST, CMAX, 32, kTeam8Tile, kEnvGroupCap, 512, W><<<grid, 64 * kFusedWPB, 0, s>>>(fa);
}

bool fused_applies(int n_categories) { return n_categories <= 28; }

bool launch_env_sweep(hipStream_t s, int n_categories, bool tag_list, int rule, const FusedArgs& fa, HostStatus* hst, uint32_t seq,
                      int grid_cap) {
    if (fa.n_pairs <= 0 || !fused_applies(n_categories) || (rule != 0 && rule != 2)) return false;
    if (fa.b.env.stride != kEnvGroupCap || fa.n_pairs >= ((int64_t)1 << 22) || fa.b.c.n >= (1 << 27)) return false;
    const int teams = rule == 0 ? 4 : 2;
    const int64_t rounds = (fa.n_pairs + teams - 1) / teams, blocks = (rounds + kFusedWPB - 1) / kFusedWPB;
    const unsigned grid = (unsigned)std::min<int64_t>(blocks, grid_cap > 0 ? grid_cap : 4096);
    const int cm = n_categories;
#define LCHD_FUSED_CASE(C) (tag_list ? launch_fused_c<true, C>(s, rule, grid, fa) : launch_fused_c<false, C>(s, rule, grid, fa))
    // (an opt-in path: four slot counts instead of the team sweeps' seven -- 17 .. 28 categories share the 28-slot instantiation)
    if (cm <= 8) LCHD_FUSED_CASE(8);
    else if (cm <= 12) LCHD_FUSED_CASE(12);
    else if (cm <= 16) LCHD_FUSED_CASE(16);
    else LCHD_FUSED_CASE(28);
#undef LCHD_FUSED_CASE
    (void)hst; (void)seq;
    return true;
}
void launch_fused_publish(hipStream_t s, const FusedArgs& fa, HostStatus* hst, uint32_t seq) { k_fused_publish<<<1, 64, 0, s>>>(fa, hst, seq); }

}  // namespace lchd

#ifdef LCHD_FUSED_STAMPS
extern "C" int lchd_debug_fused_stamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lchd::g_fused_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(lchd::g_fused_stamps), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

// lchd_env_cells.hip -- K1 (thresholded) for the capacities beyond the grouped kernel's (lchd_env_group.hip): one workgroup per
// unique anchor (env_from_idx, /root/reference/src/locohd.rs:514-542; utils::sort_together, utils.rs:25-39).
#include <algorithm>

#include "lchd_env_sort.h"

#ifndef LCHD_ENV_FLAT
#define LCHD_ENV_FLAT 4   // steps of 64 candidates whose record loads are issued together in the radius search
#endif

namespace lchd {
// ------------------------------------------------------------------------------------------------
// K1 (thresholded): one wavefront builds the sorted environment of one unique anchor.
//   radius search   kd-tree crate within_radius semantics: keep p iff sum(diff^2) < thr^2   (:521)
//   tag filter      p is the anchor itself, or pair_accepted(anchor.tag, p.tag)             (:524-528)
//   distance        sqrt(sum(diff^2)), same summation order as utils.rs:1-8                 (:537)
//   sort            ascending distance                                                       (:541)
// ------------------------------------------------------------------------------------------------
#ifdef LCHD_SWEEP_STAMPS
__device__ unsigned long long g_env_stamps[8];
#define ESTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && (blockIdx.x & 127) == 0) atomicAdd(&g_env_stamps[i], t_ - estamp_last); estamp_last = t_; } while (0)
#else
#define ESTAMP(i) do { } while (0)
#endif
// VT: the category type of the LDS buffer and of the store (uint16_t: more than 255 categories, EnvStore::cat16; no O(n) bucket sort)
template <int NT, bool TAGLIST, class VT = uint8_t>  // TAGLIST: the tag rule is a pair list (binary searches); otherwise one comparison, no branch
#ifdef ENV_W8
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 64 ? 8 : 1, NT == 64 ? 8 : 8))) void k_env_cells(
#else
__global__ __launch_bounds__(NT) void k_env_cells(
#endif
const DevConfig* __restrict__ cfgp, EnvSides sides, double thr, int cap, DeviceStatus* st) {
    // both structures in one launch: workgroups [0, sides.s[0].max_envs) build side A, the rest side B; the side's block of
    // kernel arguments is read with a wave-uniform index (scalar loads from the kernarg segment, no per-field selects)
    const int side = (int64_t)blockIdx.x >= sides.s[0].max_envs ? 1 : 0;
    const EnvSide& S = sides.s[side];
    const GridView g = S.g;
    const AnchorRec* __restrict__ uniq = S.uniq;
    const EnvStore env = S.env;
    // dynamic LDS: cap * (8 + sizeof(VT)) bytes (u64 keys, then the categories)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* key = reinterpret_cast<uint64_t*>(smem);
    VT* val = reinterpret_cast<VT*>(smem + (size_t)cap * 8);
    constexpr bool NARROW = sizeof(VT) == 1;
    __shared__ int count_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t e = (int64_t)blockIdx.x - (side ? sides.s[0].max_envs : 0);
#ifdef LCHD_SWEEP_STAMPS
    unsigned long long estamp_last = __builtin_amdgcn_s_memtime();
#endif
    if (e >= (int64_t)st->n_unique[side]) return;
    const DevConfig cfg = *cfgp;
    const AnchorRec arec = uniq[e];
    const double ax = arec.x, ay = arec.y, az = arec.z;
    const int32_t atag = (int32_t)arec.tag;
    const uint32_t apos = arec.apos;  // the anchor's own record in cell order
    const int asid = arec.sid;
    const double thr2 = thr * thr;
    const bool accept_same = cfg.tag_accept_same != 0;
    auto tag_ok = [&](int32_t t_other) -> bool {  // tag_pairing_rule.rs:49-75
        if constexpr (TAGLIST) return tag_pair_accepted(cfg, atag, t_other);
        else return (atag == t_other) == accept_same;
    };
    const int cx = cell_coord(ax, g.min[0], g.inv[0], g.dim[0]);
    const int cy = cell_coord(ay, g.min[1], g.inv[1], g.dim[1]);
    const int cz = cell_coord(az, g.min[2], g.inv[2], g.dim[2]);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
    if (NT > 64) {
        if (tid == 0) count_s = 0;
        __syncthreads();
    }

    int count = 0;  // NT == 64: the wave's running count; NT > 64: unused (count_s is the shared cursor)
    if constexpr (NT == 64) {
        // One wavefront.  The (up to) nine (y,z) rows of neighbour cells are contiguous runs of the cell-ordered records; their
        // bounds are fetched first (one dependent-load latency), then the runs are walked as ONE concatenated candidate list,
        // 64 candidates per step (every step is a full wavefront, however short the individual runs are); the record loads
        // of U steps are issued together.
        // The row bounds are worked out by lanes 0..8 (one row each: vector address arithmetic and two vector loads), turned
        // into offsets of the concatenated list by a wave scan and handed to every lane through v_readlane.  Done row by row
        // in scalar code the same thing took ~300 scalar instructions per environment, and the scalar unit (one per CU,
        // shared by all resident waves) was what bounded this kernel.
        const int kk = lane < 9 ? lane : 8;
        const int kz = (kk * 11) >> 5, ky = kk - 3 * kz;  // kk / 3, kk % 3 for kk < 9
        const int zz = cz - 1 + kz, yy = cy - 1 + ky;
        // Neighbour cells that lie wholly outside the radius are skipped: with the anchor at fractional position f in its
        // cell, a neighbour row / cell is at least (f or 1 - f) * edge away along every axis in which it differs.  A sphere
        // of radius thr meets on average 17 of the 27 cells (edge = 1.1 thr), so a third of the candidates never get loaded.
        // The test carries a relative margin of 1e-6 on thr^2 (rounding of the cell assignment is ~1e-16).
        const double fx = (ax - g.min[0]) * g.inv[0] - (double)cx, fy = (ay - g.min[1]) * g.inv[1] - (double)cy,
                     fz = (az - g.min[2]) * g.inv[2] - (double)cz;
        const double gy = fmax((ky == 0 ? fy : (ky == 2 ? 1.0 - fy : 0.0)) * g.cell[1], 0.0);
        const double gz = fmax((kz == 0 ? fz : (kz == 2 ? 1.0 - fz : 0.0)) * g.cell[2], 0.0);
        const double gxl = fmax(fx * g.cell[0], 0.0), gxh = fmax((1.0 - fx) * g.cell[0], 0.0);
        const double r2 = gy * gy + gz * gz, thr2m = thr2 * (1.0 + 1e-6);
        const int xl = (r2 + gxl * gxl < thr2m) ? x0 : cx, xh = (r2 + gxh * gxh < thr2m) ? x1 : cx;  // this row's x range
        const bool in = lane < 9 && zz >= 0 && zz < g.dim[2] && yy >= 0 && yy < g.dim[1] && r2 < thr2m;
        const int row = in ? (int)((((int64_t)asid * g.dim[2] + zz) * g.dim[1] + yy) * g.dim[0]) : 0;
        const int b_ = (int)g.cell_start[row + xl], e_ = (int)g.cell_start[row + xh + 1];
        const uint32_t len = in ? (uint32_t)(e_ - b_) : 0u;
        const uint32_t incl = wave_incl_scan_u32(len);
        const int roff_v = (int)(incl - len), dl_v = b_ - roff_v;
        // (each value passes through an empty asm: a select between two readlanes of one register is otherwise folded into
        // ONE readlane with a per-lane lane index, which the backend can only implement through a table in scratch memory)
#define LCHD_ROW(k)                                                                               \
    int dl##k = __builtin_amdgcn_readlane(dl_v, k), ro##k = __builtin_amdgcn_readlane(roff_v, k); \
    asm("" : "+s"(dl##k), "+s"(ro##k));
        LCHD_ROW(0) LCHD_ROW(1) LCHD_ROW(2) LCHD_ROW(3) LCHD_ROW(4) LCHD_ROW(5) LCHD_ROW(6) LCHD_ROW(7) LCHD_ROW(8)
#undef LCHD_ROW
        (void)ro0;
        const int total = __builtin_amdgcn_readlane((int)incl, 8);
        ESTAMP(0);
        const double2* __restrict__ rec2 = reinterpret_cast<const double2*>(g.rec);
        constexpr int U = LCHD_ENV_FLAT;
        for (int c0 = 0; c0 < total; c0 += 64 * U) {
            int idx[U];
            double2 R0[U], R1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = min(c0 + 64 * u + lane, total - 1);  // lanes past the end re-read the last candidate (masked below)
                int d = dl0;  // candidate t of the concatenated list -> record index t + dl[row of t]
                d = (t >= ro1) ? dl1 : d;
                d = (t >= ro2) ? dl2 : d;
                d = (t >= ro3) ? dl3 : d;
                d = (t >= ro4) ? dl4 : d;
                d = (t >= ro5) ? dl5 : d;
                d = (t >= ro6) ? dl6 : d;
                d = (t >= ro7) ? dl7 : d;
                d = (t >= ro8) ? dl8 : d;
                idx[u] = t + d;
                R0[u] = rec2[2 * (uint64_t)(uint32_t)idx[u]];  // (record indices are non-negative: zero extension is cheaper)
                R1[u] = rec2[2 * (uint64_t)(uint32_t)idx[u] + 1];
            }
            __builtin_amdgcn_sched_barrier(0);  // all 2U loads are issued before the first distance is computed
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c0 + 64 * u < total) {  // wave-uniform
                    const bool v = c0 + 64 * u + lane < total;
                    const double dx = R0[u].x - ax, dy = R0[u].y - ay, dz = R1[u].x - az;
                    double d2 = dx * dx;   // TU is built with -ffp-contract=off: same roundings as the
                    d2 = d2 + dy * dy;     // reference's `distance += diff * diff`
                    d2 = d2 + dz * dz;
                    const uint64_t tc = d2u(R1[u].y);  // tag | cat << 32
                    bool ok = false;
                    if constexpr (TAGLIST) {
                        if (v && d2 < thr2) ok = ((uint32_t)idx[u] == apos) || tag_ok((int32_t)(uint32_t)tc);
                    } else {  // four compares and scalar mask logic, no branch
                        ok = (v & (d2 < thr2)) & (((uint32_t)idx[u] == apos) | tag_ok((int32_t)(uint32_t)tc));
                    }
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
                    if (ok) {
                        const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (pos < cap) {
                            key[pos] = d2u(d2);  // the square root is taken after compaction (a sixth of the candidates survive)
                            val[pos] = (VT)(tc >> 32);
                        }
                    }
                    count += __popcll(m);
                }
            }
        }
    } else {
    for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.dim[2] - 1); ++zz) {
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.dim[1] - 1); ++yy) {
            // the (up to) three x-neighbour cells of one (y,z) row are contiguous in the cell-ordered arrays
            const int row = (int)((((int64_t)asid * g.dim[2] + zz) * g.dim[1] + yy) * g.dim[0]);  // neighbours of the anchor's own structure only
            const int beg = (int)g.cell_start[row + x0], end = (int)g.cell_start[row + x1 + 1];
            for (int base = beg + wave * 64; base < end; base += NT) {
                const int idx = base + lane;
                bool ok = false;
                double d2 = 0.0;
                uint32_t ccat = 0;
                if (idx < end) {
                    const CellRec r = g.rec[idx];
                    const double dx = r.x - ax, dy = r.y - ay, dz = r.z - az;
                    d2 = dx * dx;
                    d2 = d2 + dy * dy;
                    d2 = d2 + dz * dz;
                    ccat = r.cat;
                    if (d2 < thr2) ok = ((uint32_t)idx == apos) || tag_ok((int32_t)r.tag);
                }
                const unsigned long long m = __ballot(ok);
                int wbase = 0;
                // several waves append concurrently: reserve a slice of the list per wave-iteration
                if (lane == 0 && m) wbase = atomicAdd(&count_s, __popcll(m));
                wbase = __shfl(wbase, 0);
                if (ok) {
                    const int pos = wbase + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < cap) {
                        key[pos] = d2u(sqrt(d2));
                        val[pos] = (VT)ccat;
                    }
                }
            }
        }
    }
    }
    if (NT > 64) {
        __syncthreads();
        count = count_s;
    }
    ESTAMP(1);
    if (count > cap) {
        // The environment does not fit its slot: flagged, its size reported, its slot index appended to the side's overflow list.  The
        // slot receives the anchor alone -- a valid one-point environment, so the sweeps of this pass run cleanly over the pairs of
        // this anchor; the host scores those pairs again with larger slots (lchd_ctx_finish).
        if (tid == 0) {
            atomicOr(&st->flags, ST_ENV_OVERFLOW);
            atomicMax(&st->max_env, (uint32_t)count);
            const uint32_t k = atomicAdd(&st->n_overflow[side], 1u);
            if (S.ovf_list) S.ovf_list[k] = (uint32_t)e;
            const uint32_t acat = g.rec[arec.apos].cat;
            env.len[e] = 1;
            env.key[e * env.stride] = 0ull;
            reinterpret_cast<VT*>(env.cat)[e * env.stride] = (int)acat < cfg.n_categories ? (VT)acat : (VT)0;
        }
        return;
    }
    if (count == 0) {
        if (tid == 0) { atomicOr(&st->flags, ST_EMPTY_ENV); env.len[e] = 0; }
        return;
    }
    bool sorted = false;
    if constexpr (NT == 64 && NARROW) {
        // Typical environments (<= 512 points) are sorted in O(n) by one wavefront: inside a sphere the number of points
        // grows like d^3, so bucket = floor(256 * (d / thr)^3) spreads them almost evenly over 256 buckets (any
        // monotone map is correct; it only has to be balanced to be fast).  LDS histogram with returned slots -> wave
        // scan of the bucket sizes -> scatter from registers, grouped by bucket -> every element ranks itself among the
        // members of its own bucket on the exact f64 key (lane-parallel, a few LDS reads each) -> final placement.
        // Clustered inputs (a bucket with > 16 points) use the bitonic network.
        constexpr int B = 256, EPT = 8;
        if (count <= 64 * EPT) {
            uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 9 + ((16 - (((size_t)cap * 9) & 15)) & 15));  // [B + 1]
            for (int b = lane; b <= B; b += 64) hist[b] = 0u;
            __syncthreads();
            const double qs = (double)B / (thr2 * thr);  // B / thr^3
            uint64_t rk[EPT];
            uint32_t rp[EPT];  // category | bucket << 8 | slot inside the bucket << 16 (one register per element)
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int i = lane + 64 * q;
                rk[q] = 0; rp[q] = 0;
                if (i < count) {
                    const double d2 = u2d(key[i]);
                    const double d = sqrt(d2);  // utils.rs:1-8
                    rk[q] = d2u(d);
                    const double t = d2 * d * qs;
                    const int b = t < (double)B ? (int)t : B - 1;
                    rp[q] = (uint32_t)val[i] | ((uint32_t)b << 8) | (atomicAdd(&hist[b], 1u) << 16);
                }
            }
            __syncthreads();
            // exclusive scan of the 256 bucket sizes: lane l owns buckets 4l .. 4l+3
            uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            const uint32_t incl = wave_incl_scan_u32(mine);
            const uint32_t seg_lo = incl - mine;
            const unsigned long long too_big = __ballot(max(max(h0, h1), max(h2, h3)) > 16u);
            __syncthreads();
            hist[4 * lane] = seg_lo;
            hist[4 * lane + 1] = seg_lo + h0;
            hist[4 * lane + 2] = seg_lo + h0 + h1;
            hist[4 * lane + 3] = seg_lo + h0 + h1 + h2;
            if (lane == 63) hist[B] = incl;  // = count
            __syncthreads();
            if (!too_big) {
                // group by bucket (arrival order inside a bucket) ...
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = lane + 64 * q;
                    if (i < count) {
                        const uint32_t pos = hist[(rp[q] >> 8) & 0xFFu] + (rp[q] >> 16);
                        key[pos] = rk[q];
                        rp[q] = (rp[q] & 0xFFFFu) | (pos << 16);
                    }
                }
                __syncthreads();
                // ... then every element ranks itself among the (one to a few) members of its bucket on the exact f64 key
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = lane + 64 * q;
                    if (i < count) {
                        const uint32_t b = (rp[q] >> 8) & 0xFFu, pos = rp[q] >> 16;
                        const uint32_t s0 = hist[b], s1 = hist[b + 1];
                        uint32_t rank = s0;
                        for (uint32_t j = s0; j < s1; ++j) {
                            const uint64_t kj = key[j];
                            rank += (kj < rk[q]) | ((kj == rk[q]) & (j < pos));
                        }
                        rp[q] = (rp[q] & 0xFFFFu) | (rank << 16);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = lane + 64 * q;
                    if (i < count) {
                        key[rp[q] >> 16] = rk[q];
                        val[rp[q] >> 16] = (uint8_t)rp[q];
                    }
                }
                __syncthreads();
                sorted = true;
            }
        }
    }
    if (!sorted) {
        if constexpr (NT == 64) {  // the keys still hold d^2
            for (int i = tid; i < count; i += NT) key[i] = d2u(sqrt(u2d(key[i])));
            __syncthreads();
        }
        const int n2 = next_pow2(count);
        for (int i = count + tid; i < n2; i += NT) { key[i] = kPadKey; val[i] = (VT)0; }
        __syncthreads();
        bitonic_sort_lds<NT, VT>(key, val, n2, tid);
    }
    ESTAMP(2);
    uint64_t* ok_ = env.key + e * env.stride;
    VT* oc_ = reinterpret_cast<VT*>(env.cat) + e * env.stride;
    // categories outside the map are reported HERE (pmf.rs:38-42 raises for a point of a used environment, which is exactly
    // what gets written below) and stored as 0: the sweep kernels do not test categories again
    bool bad = false;
    bool written = false;
    if constexpr (NT == 64) {
        if (env.cdf_keys) {
            // One pass: sorted distance -> F(distance) -> global memory, the monotonicity of the converted keys checked on the
            // way (F is monotone; its floating-point evaluation may produce a last-bit inversion between neighbours, which the
            // running maximum of the separate path below repairs -- rare enough to pay a second pass then).  The separate
            // conversion pass over LDS, its barrier and the second read of the keys were a tenth of this kernel.
            const WfEntry wf = cfg.wf[0];
            const double* __restrict__ prm = cfg.wf_params + wf.offset;
            const double winv = cfg.wf_inv[0];
            bool inv = false;
            double carry = 0.0;  // F of the previous round's last key (F >= 0)
            for (int i0 = 0; i0 < count; i0 += 64) {
                const int i = i0 + lane;
                const bool act = i < count;
                const double f = act ? cdf_lean(wf.kind, prm, wf.n_params, winv, u2d(key[act ? i : 0])) + 0.0 : INFINITY;
                double prev = wave_shr1_f64(f);
                if (lane == 0) prev = carry;
                inv |= act && f < prev;
                carry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(f), 63), __builtin_amdgcn_readlane(__double2loint(f), 63));
                if (act) {
                    const VT v = val[i];
                    bad |= (int)v >= cfg.n_categories;
                    ok_[i] = d2u(f);
                    oc_[i] = (int)v < cfg.n_categories ? v : (VT)0;
                }
            }
            written = !__ballot(inv);
        }
    }
    if (!written) {
        if (env.cdf_keys) {
            if constexpr (NT == 64) keys_to_cdf_wave(key, count, lane, cfgp);
            else keys_to_cdf_lds<NT>(key, count, tid, cfgp);
        }
        for (int i = tid; i < count; i += NT) {
            const VT v = val[i];
            bad |= (int)v >= cfg.n_categories;
            ok_[i] = key[i];
            oc_[i] = (int)v < cfg.n_categories ? v : (VT)0;
        }
    }
    ESTAMP(3);
    if (__ballot(bad) && (tid & 63) == 0) atomicOr(&st->flags, ST_BAD_CATEGORY);
    if (tid == 0) env.len[e] = count;
    ESTAMP(4);
}

template <int NT>
static void launch_env_cells_nt(hipStream_t s, dim3 grid, size_t lds, bool tag_list, const DevConfig* cfg, const EnvSide& a, const EnvSide& b,
                                double thr, int cap, DeviceStatus* st) {
    EnvSides sides;
    sides.s[0] = a;
    sides.s[1] = b;
    if (a.env.cat16) {
        if (tag_list) k_env_cells<NT, true, uint16_t><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
        else k_env_cells<NT, false, uint16_t><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
        return;
    }
    if (tag_list) k_env_cells<NT, true><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
    else k_env_cells<NT, false><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
}

// Thresholded environments of more than 16384 points (a threshold that swallows most of a large structure): too many keys for
// LDS.  One 1024-thread workgroup per unique anchor walks the neighbour cells like k_env_cells and appends the survivors --
// distance and category -- UNSORTED to the environment's scratch row in global memory; k_env_rows then sorts each scratch row
// into the environment store in global memory, exactly as it does for given distance rows.
template <bool TAGLIST>
__global__ __launch_bounds__(1024) void k_env_collect(const DevConfig* __restrict__ cfgp, EnvSides sides, double thr, int cap, DeviceStatus* st) {
    const int side = (int64_t)blockIdx.x >= sides.s[0].max_envs ? 1 : 0;
    const EnvSide& S = sides.s[side];
    const GridView g = S.g;
    const int64_t e = (int64_t)blockIdx.x - (side ? sides.s[0].max_envs : 0);
    __shared__ int count_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (e >= (int64_t)st->n_unique[side]) return;
    const DevConfig cfg = *cfgp;
    const AnchorRec arec = S.uniq[e];
    const double ax = arec.x, ay = arec.y, az = arec.z, thr2 = thr * thr;
    const int32_t atag = (int32_t)arec.tag;
    const bool accept_same = cfg.tag_accept_same != 0;
    const int cx = cell_coord(ax, g.min[0], g.inv[0], g.dim[0]);
    const int cy = cell_coord(ay, g.min[1], g.inv[1], g.dim[1]);
    const int cz = cell_coord(az, g.min[2], g.inv[2], g.dim[2]);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
    double* __restrict__ rk = S.raw_key + e * (int64_t)cap;
    uint8_t* __restrict__ rc = S.raw_cat + e * (int64_t)cap;
    if (tid == 0) count_s = 0;
    __syncthreads();
    for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.dim[2] - 1); ++zz)
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.dim[1] - 1); ++yy) {
            const int row = (int)((((int64_t)arec.sid * g.dim[2] + zz) * g.dim[1] + yy) * g.dim[0]);
            const int beg = (int)g.cell_start[row + x0], end = (int)g.cell_start[row + x1 + 1];
            for (int base = beg + wave * 64; base < end; base += 1024) {
                const int idx = base + lane;
                bool ok = false;
                double d2 = 0.0;
                uint32_t ccat = 0;
                if (idx < end) {
                    const CellRec r = g.rec[idx];
                    const double dx = r.x - ax, dy = r.y - ay, dz = r.z - az;
                    d2 = dx * dx;  // utils.rs:1-8 order, uncontracted
                    d2 = d2 + dy * dy;
                    d2 = d2 + dz * dz;
                    ccat = r.cat;
                    if (d2 < thr2) {
                        if constexpr (TAGLIST) ok = ((uint32_t)idx == arec.apos) || tag_pair_accepted(cfg, atag, (int32_t)r.tag);
                        else ok = ((uint32_t)idx == arec.apos) || ((atag == (int32_t)r.tag) == accept_same);
                    }
                }
                const unsigned long long m = __ballot(ok);
                int wbase = 0;
                if (lane == 0 && m) wbase = atomicAdd(&count_s, __popcll(m));
                wbase = __shfl(wbase, 0);
                if (ok) {
                    const int pos = wbase + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < cap) { rk[pos] = sqrt(d2); rc[pos] = (uint8_t)ccat; }
                }
            }
        }
    __syncthreads();
    if (tid == 0) {
        const int count = count_s;
        if (count > cap) { atomicOr(&st->flags, ST_ENV_OVERFLOW); atomicMax(&st->max_env, (uint32_t)count); S.env.len[e] = 0; }
        else if (count == 0) { atomicOr(&st->flags, ST_EMPTY_ENV); S.env.len[e] = 0; }
        else S.env.len[e] = count;  // (k_env_rows sorts the row into the store and keeps this length)
    }
}

bool launch_env_cells(hipStream_t s, int cap, const DevConfig* cfg, bool tag_list, const EnvSide& a, const EnvSide& b, double thr,
                      DeviceStatus* st) {
    if (a.max_envs + b.max_envs <= 0) return true;
    const bool cat16 = a.env.cat16 != 0;
    if (cap > 16384 && cap <= (1 << 23) && !(cap & (cap - 1))) {  // collect unsorted, then the global-memory row sort (beyond 65536: swept by k_sweep_wide<.., BIG>)
        if (!a.raw_key || !b.raw_key || cat16) return false;
        EnvSides sides;
        sides.s[0] = a;
        sides.s[1] = b;
        const dim3 grid((unsigned)(a.max_envs + b.max_envs));
        if (tag_list) k_env_collect<true><<<grid, 1024, 0, s>>>(cfg, sides, thr, cap, st);
        else k_env_collect<false><<<grid, 1024, 0, s>>>(cfg, sides, thr, cap, st);
        for (int side = 0; side < 2; ++side) {
            const EnvSide& S = side ? b : a;
            if (S.max_envs <= 0) continue;
            RowExtras ex{S.raw_cat, S.env.len, &st->n_unique[side]};
            if (!launch_env_rows(s, cap, cfg, S.c, S.raw_key, cap, S.max_envs, cap, 0.0, S.env, st, ex)) return false;
        }
        return true;
    }
    if (cap < 64 || cap > 16384 || (cap & (cap - 1))) return false;
    if (cat16 && cap > 8192) return false;  // (10 bytes per point: 16384 points would need the CU's whole LDS)
    const dim3 grid((unsigned)(a.max_envs + b.max_envs));
    const size_t lds = (size_t)cap * (cat16 ? 10 : 9);
    if (cap <= 2048) {
        launch_env_cells_nt<64>(s, grid, lds + 16 + 257 * sizeof(uint32_t), tag_list, cfg, a, b, thr, cap, st);
    } else if (cap <= 4096) {
        launch_env_cells_nt<256>(s, grid, lds, tag_list, cfg, a, b, thr, cap, st);
    } else {
        launch_env_cells_nt<1024>(s, grid, lds, tag_list, cfg, a, b, thr, cap, st);
    }
    return true;
}

void init_env_cells_kernels() {
    auto raise = [](const void* fn, int bytes) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, false>), 16384 * 9);
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, true>), 16384 * 9);
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, false, uint16_t>), 8192 * 10);
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, true, uint16_t>), 8192 * 10);
    (void)hipGetLastError();
}

}  // namespace lchd

#ifdef LCHD_SWEEP_STAMPS
extern "C" int lchd_debug_env_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lchd::g_env_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lchd::g_env_stamps), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// lchd_env_rows.hip -- K1 (dense): one workgroup sorts one full row -- from_coords (utils::calculate_distance_matrix,
// /root/reference/src/locohd/utils.rs:10-22 + sort_together :25-39) or from_dmxs (src/locohd.rs:439-440) -- for the configurations
// and row lengths the fused dense kernel (lchd_dense_fused.hip) does not take.
#include <algorithm>

#include "lchd_env_sort.h"

namespace lchd {
#ifdef LCHD_SWEEP_STAMPS
__device__ unsigned long long g_rows_stamps[8];
#define ESTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && (blockIdx.x & 127) == 0) atomicAdd(&g_rows_stamps[i], t_ - estamp_last); estamp_last = t_; } while (0)
#else
#define ESTAMP(i) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------------
// K1 (dense): one workgroup sorts one full row -- from_coords (distances from anchor `row` to every atom,
// utils.rs:10-22 + :25-39) or from_dmxs (a caller-supplied distance-matrix row, src/locohd.rs:439-440).
// Dynamic LDS: n2 * 9 bytes.
// ------------------------------------------------------------------------------------------------
constexpr int kRowBucketsSmall = 2048;   // distance buckets of the dense-row sort (rows <= 16384 points)
constexpr int kRowBucketsMax = 8192;     // ... of k_env_rows2 when the row leaves room for them
constexpr int kRowBucketsBig = 16384;    // ... for rows of up to 65535 points (keys stay in global memory)
constexpr int kRowBucketsHuge = 32768;   // ... for longer rows
constexpr int kRowCoarse = 256;       // uniform bins of the row's empirical distance CDF
constexpr int kRowBucketLimit = 64;   // a fuller bucket sends the row to the bitonic network instead

template <int NT, bool GLOBALKV, class VT = uint8_t>  // VT uint16_t: more than 255 categories (EnvStore::cat16, CloudView::cat_hi)
__global__ __launch_bounds__(NT) void k_env_rows(const DevConfig* __restrict__ cfgp, CloudView c, const double* __restrict__ dmx,
                                                 int64_t ld, int64_t row_len, int n2, int n_buckets, double image_bound, EnvStore env,
                                                 DeviceStatus* st, RowExtras ex) {
    // ex (thresholded environments of more than 16384 points, collected unsorted by k_env_collect): the row's own categories
    // and length instead of the structure's, rows beyond the side's unique anchors are not there
    if (ex.n_unique && (uint32_t)blockIdx.x >= *ex.n_unique) return;
    // Sorting one row of n <= 16384 distances in O(n): an empirical CDF of the row on kRowCoarse uniform bins of
    // [0, max] of a monotone image of the distance (d^2 for coordinates) gives every point an interpolated rank; rank * kRowBuckets / n is its bucket, so buckets hold
    // ~n / kRowBuckets points whatever the shape of the cloud.  One LDS histogram + scan + scatter puts the points
    // into bucket order, then one thread finishes each bucket with an insertion sort on the exact f64 keys.  The map
    // distance -> bucket is monotone, which is all correctness needs; a pathological row (a bucket with more than
    // kRowBucketLimit points, e.g. thousands of identical distances) takes the bitonic network instead.
    // Distances are recomputed in every phase (3 L2-resident loads + a sqrt) instead of being kept in registers.
    // GLOBALKV (rows of 16 385 .. 65 535 points): the keys are sorted in place in the environment store (global memory,
    // L2-resident per row) and only the bucket histogram lives in LDS; otherwise keys and categories are in LDS too.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t r_ = blockIdx.x;
    uint64_t* key = GLOBALKV ? env.key + r_ * env.stride : reinterpret_cast<uint64_t*>(smem);
    VT* val = GLOBALKV ? reinterpret_cast<VT*>(env.cat) + r_ * env.stride : reinterpret_cast<VT*>(smem + (size_t)n2 * 8);
    constexpr size_t kPer = 8 + sizeof(VT);
    uint32_t* hist = GLOBALKV ? reinterpret_cast<uint32_t*>(smem)
                              : reinterpret_cast<uint32_t*>(smem + (size_t)n2 * kPer + ((16 - (((size_t)n2 * kPer) & 15)) & 15));  // [n_buckets + 1]
    const int kRowBuckets = n_buckets;
    __shared__ double red_max[16];
    __shared__ uint32_t red_cnt[16];
    __shared__ uint32_t scan_carry;
    __shared__ uint32_t coarse[kRowCoarse + 1], cum[kRowCoarse + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = blockIdx.x;
    const int n = ex.row_lens ? ex.row_lens[r] : (int)row_len;
    if (n <= 0) return;  // (an environment its collector flagged as empty or too large)
    const double* __restrict__ row = dmx ? dmx + r * ld : nullptr;
    const uint8_t* __restrict__ cats = ex.row_cat ? ex.row_cat + r * ld : c.cat;
    auto cat_at = [&](int i) -> VT {  // (two-byte ids: the structure's own planes; the collected rows of k_env_collect are one-byte only)
        if constexpr (sizeof(VT) == 2) return (VT)cat_of_atom(c, i);
        else return cats[i];
    };
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (!dmx) { ax = c.x[r]; ay = c.y[r]; az = c.z[r]; }
    // The bucketing phases work on a MONOTONE image of the distance -- the squared distance for coordinates (no square root
    // until the key is written), the distance itself for a given row -- and only the scatter takes the root of the survivors'
    // d^2; `dist_of` of the same image is what the reference computes (utils.rs:1-8).
    auto image_of = [&](int i, bool& bad) -> double {
        if (dmx) {
            double v = row[i];
            if (!(v >= 0.0)) { bad = true; v = 0.0; }  // negative or NaN
            return v + 0.0;                              // -0.0 -> +0.0
        }
        const double dx = ax - c.x[i], dy = ay - c.y[i], dz = az - c.z[i];
        double d2 = dx * dx;  // utils.rs:1-8 order, uncontracted
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        return d2;
    };
    auto dist_from_image = [&](double m) -> double { return dmx ? m : sqrt(m); };
    auto dist_of = [&](int i, bool& bad) -> double { return dist_from_image(image_of(i, bad)); };

    // 1. largest finite distance image -- or, for coordinates, the caller's bound (squared diagonal of the bounding box): any
    //    upper bound will do, the empirical CDF below adapts the buckets to wherever the points really are
    bool bad = false;
    double dmax = image_bound > 0.0 ? image_bound : 0.0;
    if (!(image_bound > 0.0))
        for (int i = tid; i < n; i += NT) {
            const double v = image_of(i, bad);
            if (v < 1.0e300 && v > dmax) dmax = v;
        }
    if (bad) atomicOr(&st->flags, ST_BAD_DISTANCE);
    for (int m = 32; m > 0; m >>= 1) dmax = fmax(dmax, shfl_xor_f64(dmax, m));
    if (lane == 0) red_max[wave] = dmax;
    for (int b = tid; b <= kRowBuckets; b += NT) hist[b] = 0u;
    for (int b = tid; b <= kRowCoarse; b += NT) coarse[b] = 0u;
    if (tid == 0) scan_carry = 0;
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) dmax = fmax(dmax, red_max[w]);
    // 2. empirical CDF on the coarse bins
    const double inv_w = dmax > 0.0 ? (double)kRowCoarse / dmax : 0.0;
    for (int i = tid; i < n; i += NT) {
        const double v = image_of(i, bad);
        if (v <= dmax) atomicAdd(&coarse[min((int)(v * inv_w), kRowCoarse - 1)], 1u);
    }
    __syncthreads();
    if (wave == 0) {  // cum[b] = points below bin b
        uint32_t carry = 0;
        for (int base = 0; base < kRowCoarse; base += 64) {
            const uint32_t v = coarse[base + lane];
            const uint32_t incl = wave_incl_scan_u32(v);
            cum[base + lane] = carry + incl - v;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) cum[kRowCoarse] = carry;
    }
    __syncthreads();
    const double rank_scale = n > 0 ? (double)kRowBuckets / (double)n : 0.0;
    auto bucket_of = [&](double v) -> int {
        if (!(v <= dmax)) return kRowBuckets - 1;  // +inf entries of a distance matrix
        const double t = v * inv_w;
        const int bin = min((int)t, kRowCoarse - 1);
        const double frac = fmin(t - (double)bin, 1.0);
        const double q = ((double)cum[bin] + frac * (double)coarse[bin]) * rank_scale;
        return q < (double)kRowBuckets ? (int)q : kRowBuckets - 1;
    };
    // 3. bucket histogram
    uint32_t biggest = 0;
    for (int i = tid; i < n; i += NT) biggest = max(biggest, atomicAdd(&hist[bucket_of(image_of(i, bad))], 1u) + 1u);
    for (int m = 32; m > 0; m >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, m));
    if (lane == 0) red_cnt[wave] = biggest;
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) biggest = max(biggest, red_cnt[w]);

    if (biggest > (uint32_t)kRowBucketLimit) {
        for (int i = tid; i < n2; i += NT) {
            key[i] = i < n ? d2u(dist_of(i, bad)) : kPadKey;
            val[i] = i < n ? cat_at(i) : (VT)0;
        }
        __syncthreads();
        bitonic_sort_lds<NT, VT>(key, val, n2, tid);
    } else {
        // 4. exclusive scan: hist[b] = first slot of bucket b
        for (int base = 0; base < kRowBuckets; base += NT) {
            const int b = base + tid;
            const uint32_t v = b < kRowBuckets ? hist[b] : 0u;
            const uint32_t incl = wave_incl_scan_u32(v);
            if (lane == 63) red_cnt[wave] = incl;
            __syncthreads();
            uint32_t wpre = 0;
            for (int w = 0; w < wave; ++w) wpre += red_cnt[w];
            const uint32_t carry = scan_carry;
            if (b < kRowBuckets) hist[b] = carry + wpre + incl - v;
            __syncthreads();
            if (tid == NT - 1) scan_carry = carry + wpre + incl;
            __syncthreads();
        }
        // 5. scatter; the bucket cursor advances in place, so afterwards hist[b] = END of bucket b
        for (int i = tid; i < n; i += NT) {
            const double m = image_of(i, bad);
            const uint32_t pos = atomicAdd(&hist[bucket_of(m)], 1u);
            key[pos] = d2u(dist_from_image(m));
            val[pos] = cat_at(i);
        }
        __syncthreads();
        // 6. finish every bucket with an insertion sort on the exact keys
        for (int b = tid; b < kRowBuckets; b += NT) {
            const int lo = b ? (int)hist[b - 1] : 0, hi = (int)hist[b];
            for (int i = lo + 1; i < hi; ++i) {
                const uint64_t k = key[i];
                const VT v = val[i];
                int j = i - 1;
                while (j >= lo && key[j] > k) { key[j + 1] = key[j]; val[j + 1] = val[j]; --j; }
                key[j + 1] = k;
                val[j + 1] = v;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        env.len[r] = n;
        if (n > 0 && key[0] != 0ull) atomicOr(&st->flags, ST_FIRST_NOT_ZERO);  // src/locohd.rs:74-77, on the distance
    }
    __syncthreads();
    if (env.cdf_keys) keys_to_cdf_lds<NT>(key, n, tid, cfgp);
    {   // categories outside the map: reported here, stored as 0 (see k_env_cells)
        const int C = cfgp->n_categories;
        bool bad_c = false;
        if constexpr (!GLOBALKV) {
            uint64_t* ok_ = env.key + r * env.stride;
            VT* oc_ = reinterpret_cast<VT*>(env.cat) + r * env.stride;
            for (int i = tid; i < n; i += NT) {
                const VT v = val[i];
                bad_c |= (int)v >= C;
                ok_[i] = key[i];
                oc_[i] = (int)v < C ? v : (VT)0;
            }
        } else {
            for (int i = tid; i < n; i += NT) {
                const VT v = val[i];
                if ((int)v >= C) { bad_c = true; val[i] = 0; }
            }
        }
        if (__ballot(bad_c) && (tid & 63) == 0) atomicOr(&st->flags, ST_BAD_CATEGORY);
    }
}

// ------------------------------------------------------------------------------------------------
// K1 (dense), rows of at most 16 384 points: the same bucket sort with every point's distance image computed ONCE and held
// in registers (EPT points per thread), one LDS atomic per histogram, a scatter without atomics (bucket start + the slot
// the histogram atomic returned), and the last step done by ALL threads: every point ranks itself among the handful of
// members of its bucket on the exact f64 key, then writes its (CDF-converted) key to its final place.  (k_env_rows
// recomputes the distances in three passes and finishes the buckets with one thread each -- 3.6 ms for the 2 x 10^4 rows
// of two 10^4-atom structures; this kernel builds both structures' rows in one launch.)
// Dynamic LDS: n2 * 9 bytes (keys, categories) + the bucket histogram.
// ------------------------------------------------------------------------------------------------
constexpr int kRowSegCap = 11776;  // rows of more than 16384 points: most points of one distance segment (keys in LDS)
constexpr int kRowLongEpt = 20;    // ... and the points per thread of such a row (<= 20480 points, 1024 threads)
constexpr size_t kRowSegLds = (size_t)kRowSegCap * 9 + (size_t)(kRowBucketsMax + 1) * 4;  // keys, categories, histogram
static_assert(((size_t)kRowSegCap * 9) % 16 == 0, "histogram alignment");
// EPT: points per thread of ONE sort -- the whole row, or one distance segment of a long row (NSEG = 2).
// Long rows (16385 .. 20480 points: more keys than the LDS holds) are sorted segment by segment: the coarse empirical CDF
// says which half of the buckets -- the nearer or the farther half of the row, ~n/2 points each -- a point falls into;
// for each segment the block compacts its points into the key array (ballot prefix inside a wave, wave totals through LDS:
// a deterministic order), every thread takes EPT of them back into registers, and from there the sort is the one of a
// 10^4-point row.  (A first version kept all 20 points of a thread in registers through both segments and tested
// "is it in this segment?" per point and phase: half the lanes idle in every instruction, 136 bytes of spill: 22.8 ms
// for the 4 x 10^4 rows of two 2 x 10^4-atom structures.)
template <int NT, int EPT, int NSEG>
__global__ __launch_bounds__(NT, (NT == 512 ? 2 : 4)) void k_env_rows2(const DevConfig* __restrict__ cfgp, RowSides sides, int n2, DeviceStatus* st) {
    static_assert(NSEG == 1 || NSEG == 2, "a point's segment is one bit");
    constexpr int n_seg = NSEG;
    constexpr int PEPT = NSEG > 1 ? kRowLongEpt : EPT;  // points per thread of the whole row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* key = reinterpret_cast<uint64_t*>(smem);
    uint8_t* val = smem + (NSEG > 1 ? (size_t)kRowSegCap : (size_t)n2) * 8;
    __shared__ double red_max[NT / 64];
    __shared__ uint32_t red_cnt[NT / 64], far_cnt[NT / 64];
    __shared__ uint32_t seg_tot[kRowBucketsMax / 64 + 1];
    __shared__ uint32_t coarse[kRowCoarse + 1], cum[kRowCoarse + 1];
    __shared__ uint32_t seg_n_s;
    __shared__ uint64_t carry_key_s;
    __shared__ double split_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int side = (int64_t)blockIdx.x >= sides.n_rows ? 1 : 0;
    const RowSide& S = sides.s[side];
    const int64_t r = (int64_t)blockIdx.x - (side ? sides.n_rows : 0);
    const CloudView c = S.c;
    const EnvStore env = S.env;
    const int n = S.row_lens ? S.row_lens[r] : (int)S.row_len;  // (ragged distance matrices: every row its own length)
    // Buckets: as many as fit (up to kRowBucketsMax, ~1 point per bucket: the ranking step reads a bucket's members once
    // per member).  The histogram lives in the part of the key array the row does not need -- the array is sized for the
    // bitonic fallback, a power of two --, or behind the categories when the row fills it.
    int NB = kRowBucketsSmall;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (size_t)n2 * 9 + ((16 - (((size_t)n2 * 9) & 15)) & 15));  // [NB + 1]
    if (n_seg > 1) {
        NB = kRowBucketsMax;
        hist = reinterpret_cast<uint32_t*>(smem + (size_t)kRowSegCap * 9);
    } else {
        const int nk = (n + 63) & ~63;
        while (NB > 64 && NB >= 4 * nk) NB >>= 1;  // short rows: no more than ~2 buckets per point
        for (int cand = kRowBucketsMax; cand > kRowBucketsSmall; cand >>= 1)
            if (cand <= 2 * nk && (size_t)(n2 - nk) * 8 >= (size_t)(cand + 1) * 4) {
                NB = cand;
                hist = reinterpret_cast<uint32_t*>(key + nk);
                break;
            }
    }
    const double* __restrict__ row = S.dmx ? S.dmx + r * S.ld : nullptr;
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (!row) { ax = c.x[r]; ay = c.y[r]; az = c.z[r]; }
#ifdef LCHD_SWEEP_STAMPS
    unsigned long long estamp_last = __builtin_amdgcn_s_memtime();
#endif

    // 1. the distance image of this thread's points (d^2 for coordinates, utils.rs:1-8 order, uncontracted; the distance
    //    itself for a given row) and their categories (four to a register).  Point i = tid + q * NT: coalesced.
    //    (The "given row or coordinates?" test stays OUTSIDE the loops over a thread's points: inside, it was a branch per
    //    point -- wave-uniform, but the loads behind it were issued one point after the other.)
    bool bad = false;
    auto image_row = [&](int tid, int q) -> double {
        const int i = tid + q * NT;
        double v = row[i < n ? i : 0];
        if (i < n && !(v >= 0.0)) { bad = true; v = 0.0; }  // negative or NaN
        return v + 0.0;                                      // -0.0 -> +0.0
    };
    auto image_xyz = [&](int tid, int q) -> double {
        const int i = tid + q * NT;
        const int ii = i < n ? i : 0;
        const double dx = ax - c.x[ii], dy = ay - c.y[ii], dz = az - c.z[ii];
        double d2 = dx * dx;
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        return d2;
    };
    constexpr int IB = PEPT <= 10 ? PEPT : (PEPT % 10 == 0 ? 10 : 8);
    double m[EPT];                   // the images of the points being sorted (the row, or the current segment)
    double mp[NSEG > 1 ? PEPT : 1];  // long rows: the images of all the thread's points while they are dealt to the segments
    uint32_t ct4[(EPT + 3) / 4];
    uint32_t cp4[NSEG > 1 ? (PEPT + 3) / 4 : 1];  // ... and their categories
#pragma unroll
    for (int q = 0; q < (EPT + 3) / 4; ++q) ct4[q] = 0u;
#pragma unroll
    for (int q = 0; q < (NSEG > 1 ? (PEPT + 3) / 4 : 1); ++q) cp4[q] = 0u;
    if constexpr (NSEG == 1) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * NT;
            ct4[q >> 2] |= (uint32_t)c.cat[i < n ? i : 0] << ((q & 3) * 8);
        }
        if (row) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) m[q] = image_row(tid, q);
        } else {
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                m[q] = image_xyz(tid, q);
                if ((q + 1) % IB == 0) __builtin_amdgcn_sched_barrier(0);  // (at most 3 * IB coordinate loads in flight)
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < PEPT; ++q) {
            const int i = tid + q * NT;
            cp4[q >> 2] |= (uint32_t)c.cat[i < n ? i : 0] << ((q & 3) * 8);
        }
        if (row) {
#pragma unroll
            for (int q = 0; q < PEPT; ++q) mp[q] = image_row(tid, q);
        } else {
#pragma unroll
            for (int q = 0; q < PEPT; ++q) {
                mp[q] = image_xyz(tid, q);
                if ((q + 1) % IB == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    auto row_img = [&](int q) -> double {  // q static
        if constexpr (NSEG > 1) return mp[q];
        else return m[q];
    };
    auto cat_of = [&](int q) -> uint8_t { return (uint8_t)(ct4[q >> 2] >> ((q & 3) * 8)); };  // q static
    if (__ballot(bad) && lane == 0) atomicOr(&st->flags, ST_BAD_DISTANCE);
    ESTAMP(0);
    // largest finite image -- or, for coordinates, the caller's bound (squared diagonal of the bounding box): any upper
    // bound will do, the empirical CDF below adapts the buckets to wherever the points really are
    double dmax = S.image_bound > 0.0 ? S.image_bound : 0.0;
    if (!(S.image_bound > 0.0)) {
#pragma unroll
        for (int q = 0; q < PEPT; ++q)
            if (tid + q * NT < n && row_img(q) < 1.0e300 && row_img(q) > dmax) dmax = row_img(q);
        for (int k = 32; k > 0; k >>= 1) dmax = fmax(dmax, shfl_xor_f64(dmax, k));
        if (lane == 0) red_max[wave] = dmax;
    }
    for (int b = tid; b <= kRowCoarse; b += NT) coarse[b] = 0u;
    if (tid == 0) carry_key_s = 0ull;
    __syncthreads();
    if (!(S.image_bound > 0.0))
        for (int w = 0; w < NT / 64; ++w) dmax = fmax(dmax, red_max[w]);
    // 2. empirical CDF of the row on kRowCoarse uniform bins of [0, dmax]
    const double inv_w = dmax > 0.0 ? (double)kRowCoarse / dmax : 0.0;
#pragma unroll
    for (int q = 0; q < PEPT; ++q)
        if (tid + q * NT < n && row_img(q) <= dmax) atomicAdd(&coarse[min((int)(row_img(q) * inv_w), kRowCoarse - 1)], 1u);
    __syncthreads();
    if (wave == 0) {  // cum[b] = points below bin b
        uint32_t carry = 0;
        for (int base = 0; base < kRowCoarse; base += 64) {
            const uint32_t v = coarse[base + lane];
            const uint32_t incl = wave_incl_scan_u32(v);
            cum[base + lane] = carry + incl - v;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) cum[kRowCoarse] = carry;
    }
    __syncthreads();
    ESTAMP(1);
    const DevConfig cfg = *cfgp;
    const WfEntry wf = cfg.wf[0];
    const double* __restrict__ prm = cfg.wf_params + wf.offset;
    const double winv = cfg.wf_inv[0];
    const int NBT = NB * n_seg;  // buckets of the whole row; distance segment sg owns buckets [sg * NB, (sg + 1) * NB)
    const double rank_scale = n > 0 ? (double)NBT / (double)n : 0.0;
    // interpolated rank of an image in the row -> one of NBT balanced buckets.  Single precision: any map that never
    // decreases with the image sorts correctly (rounding to float, the product with a positive constant, the truncation and
    // the interpolation inside a bin -- which never exceeds the next bin's start -- all are), it only has to balance the
    // buckets, and the double-precision conversions were a third of this phase's instructions.
    const float inv_wf = (float)inv_w, rank_scale_f = (float)rank_scale;
    auto bucket_of = [&](double v) -> int {
        int gb = NBT - 1;  // +inf entries of a distance matrix
        if (v <= dmax) {
            const float t = (float)v * inv_wf;
            const int bin = min((int)t, kRowCoarse - 1);
            const float frac = fminf(t - (float)bin, 1.0f);
            const float qq = ((float)cum[bin] + frac * (float)coarse[bin]) * rank_scale_f;
            gb = qq < (float)NBT ? (int)qq : NBT - 1;
        }
        return gb;
    };
    uint64_t* ok_ = env.key + r * env.stride;
    uint8_t* oc_ = env.cat + r * env.stride;
    uint32_t seg_total[2] = {0u, 0u};
    if constexpr (NSEG > 1) {
        // Long rows: deal the points to the two segments, ONCE and from the registers (every further pass over the row's
        // coordinates costs ~8 000 cycles of this CU's 64-byte-per-clock L1 path: 480 KB).  The nearer segment's points go
        // into the key array, the farther segment's into the row's own slot of the environment store (which its sorted
        // keys overwrite at the end); position = points of the lower waves + of this wave's earlier q + of the lower lanes:
        // a deterministic order.
        // Which segment?  One comparison with the image at which the empirical CDF reaches n / 2 (any threshold keeps the
        // two segments ordered; this one balances them).
        if (wave == 0) {
            const uint32_t half = (uint32_t)n / 2u;
            int bin = 0;  // the last bin that starts at or below the median rank
            for (int b = lane; b < kRowCoarse; b += 64) bin = cum[b] <= half ? b : bin;
            for (int k = 32; k > 0; k >>= 1) bin = max(bin, __shfl_xor(bin, k));
            if (lane == 0) {
                const double inside = coarse[bin] ? (double)(half - cum[bin]) / (double)coarse[bin] : 0.0;
                split_s = inv_w > 0.0 ? ((double)bin + fmin(inside, 1.0)) / inv_w : 0.0;
            }
        }
        __syncthreads();
        const double split = split_s;
        uint32_t seg_bits = 0u;  // bit q = the segment of point q
#pragma unroll
        for (int q = 0; q < PEPT; ++q)
            if (tid + q * NT < n && mp[q] >= split) seg_bits |= 1u << q;
        ESTAMP(0);  // (diagnostic builds: the segment bits are booked on the image phase, the dealing on the coarse-CDF phase)
        uint32_t wn0 = 0, wn1 = 0;
#pragma unroll
        for (int q = 0; q < PEPT; ++q) {
            const bool in = tid + q * NT < n, far = (seg_bits >> q) & 1u;
            wn0 += (uint32_t)__popcll(__ballot(in && !far));
            wn1 += (uint32_t)__popcll(__ballot(in && far));
        }
        if (lane == 0) { red_cnt[wave] = wn0; far_cnt[wave] = wn1; }
        __syncthreads();
        uint32_t at0 = 0, at1 = 0;
        for (int w = 0; w < NT / 64; ++w) {
            const uint32_t v0 = red_cnt[w], v1 = far_cnt[w];
            at0 += w < wave ? v0 : 0u;
            at1 += w < wave ? v1 : 0u;
            seg_total[0] += v0;
            seg_total[1] += v1;
        }
        const uint32_t seg_max = max(seg_total[0], seg_total[1]);
        if (seg_max > (uint32_t)kRowSegCap || seg_max > (uint32_t)(EPT * NT)) {
            // the empirical CDF balanced the segments badly (no in-LDS fallback for these rows: the host repeats the call
            // with k_env_rows)
            if (tid == 0) { atomicOr(&st->flags, ST_ROW_RETRY); env.len[r] = 0; }
            return;
        }
#pragma unroll
        for (int q = 0; q < PEPT; ++q) {
            const int i = tid + q * NT;
            const bool in = i < n, far = (seg_bits >> q) & 1u;
            const unsigned long long m0 = __ballot(in && !far), m1 = __ballot(in && far);
            const uint8_t cv = (uint8_t)(cp4[q >> 2] >> ((q & 3) * 8));
            if (in && !far) {
                const uint32_t pos = at0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                key[pos] = d2u(mp[q]);
                val[pos] = cv;
            }
            if (in && far) {
                const uint32_t pos = seg_total[0] + at1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                ok_[pos] = d2u(mp[q]);
                oc_[pos] = cv;
            }
            at0 += (uint32_t)__popcll(m0);
            at1 += (uint32_t)__popcll(m1);
        }
        __syncthreads();
        ESTAMP(1);
    }
    bool bad_c = false;
    int seg_base = 0;
#pragma unroll 1
    for (int sg = 0; sg < NSEG; ++sg) {
        // (an opaque copy of the thread index: addresses derived from it -- 60 coordinate pointers -- are otherwise hoisted out of
        //  the segment loop and kept alive through it: 280 spilled registers)
        int tl = tid;
        if constexpr (NSEG > 1) asm volatile("" : "+v"(tl));
        int n_pts = n;  // points of this sort
        if constexpr (NSEG > 1) {
            // every thread takes EPT of the segment's points back into registers: from the key array, or from the row's slot
            // of the environment store
            const uint32_t total = seg_total[sg];
            n_pts = (int)total;
#pragma unroll
            for (int q = 0; q < (EPT + 3) / 4; ++q) ct4[q] = 0u;
            if (sg == 0) {
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = tl + q * NT, ii = i < n_pts ? i : 0;
                    m[q] = u2d(key[ii]);
                    ct4[q >> 2] |= (uint32_t)val[ii] << ((q & 3) * 8);
                }
            } else {  // (clamped, unconditional loads: all in flight together; .glc -- written by other waves of this block)
                const uint64_t* src_k = ok_ + seg_total[0];
                const uint8_t* src_c = oc_ + seg_total[0];
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = tl + q * NT, ii = i < n_pts ? i : 0;
                    m[q] = u2d(__hip_atomic_load(&src_k[ii], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    ct4[q >> 2] |= (uint32_t)__hip_atomic_load(&src_c[ii], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << ((q & 3) * 8);
                }
            }
            __syncthreads();  // (the sort below reuses the key array)
        }
        for (int b = tl; b <= NB; b += NT) hist[b] = 0u;
        __syncthreads();
        // 3. the point's bucket; the histogram atomic returns its slot inside the bucket
        uint32_t bs[EPT];  // bucket | slot << 13
        uint32_t biggest = 0;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            bs[q] = ~0u;
            if (tl + q * NT < n_pts) {
                const int b = min(max(bucket_of(m[q]) - sg * NB, 0), NB - 1);  // (in range by the choice of the segment)
                const uint32_t slot = atomicAdd(&hist[b], 1u);
                bs[q] = (uint32_t)b | (slot << 13);
                biggest = max(biggest, slot + 1u);
            }
        }
        for (int k = 32; k > 0; k >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, k));
        if (lane == 0) red_cnt[wave] = biggest;
        __syncthreads();
        for (int w = 0; w < NT / 64; ++w) biggest = max(biggest, red_cnt[w]);
        __syncthreads();
        ESTAMP(2);
        if (biggest > (uint32_t)kRowBucketLimit) {
            if (n_seg > 1) {  // (rows of more than 16384 points have no in-LDS fallback: the host repeats the call with k_env_rows)
                if (tl == 0) { atomicOr(&st->flags, ST_ROW_RETRY); env.len[r] = 0; }
                return;
            }
            // a pathological row (thousands of identical distances): the bitonic network on the exact keys
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int i = tl + q * NT;
                if (i < n) { key[i] = d2u(row ? m[q] : sqrt(m[q])); val[i] = cat_of(q); }
            }
            for (int i = n + tl; i < n2; i += NT) { key[i] = kPadKey; val[i] = 0; }
            __syncthreads();
            bitonic_sort_lds<NT>(key, val, n2, tl);
            if (tl == 0 && n > 0 && key[0] != 0ull) atomicOr(&st->flags, ST_FIRST_NOT_ZERO);  // src/locohd.rs:74-77
            __syncthreads();
            if (env.cdf_keys) keys_to_cdf_lds<NT>(key, n, tl, cfgp);
            if (tl == 0) seg_n_s = (uint32_t)n;
            __syncthreads();
        } else {
            // 4. exclusive scan in groups of 64 buckets (one wavefront scan each), then of the group totals, then one
            //    coalesced pass adds the group offsets: hist[b] = first slot of bucket b, hist[NB] = points of the segment
            const int n_grp = NB >> 6;
            for (int gq = wave; gq < n_grp; gq += NT / 64) {
                const uint32_t v = hist[gq * 64 + lane];
                const uint32_t incl = wave_incl_scan_u32(v);
                hist[gq * 64 + lane] = incl - v;
                if (lane == 63) seg_tot[gq] = incl;
            }
            __syncthreads();
            if (wave == 0) {
                uint32_t carry = 0;
                for (int base = 0; base < n_grp; base += 64) {
                    const uint32_t v = base + lane < n_grp ? seg_tot[base + lane] : 0u;
                    const uint32_t incl = wave_incl_scan_u32(v);
                    if (base + lane < n_grp) seg_tot[base + lane] = carry + incl - v;
                    carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                if (lane == 0) seg_n_s = carry;
            }
            __syncthreads();
            for (int b = tl; b < NB; b += NT) hist[b] += seg_tot[b >> 6];
            if (tl == 0) hist[NB] = seg_n_s;
            __syncthreads();
            ESTAMP(3);
            // 5. scatter (no atomics): position = bucket start + slot; the exact distance replaces the image in the register
#pragma unroll
            for (int q = 0; q < EPT; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t b = bs[q] & 8191u, pos = hist[b] + (bs[q] >> 13);
                    if (!row) m[q] = sqrt(m[q]);  // utils.rs:1-8
                    key[pos] = d2u(m[q]);
                    bs[q] = b | (pos << 13);
                }
            __syncthreads();
            ESTAMP(4);
            // 6. every point ranks itself among the members of its bucket on the exact key (ties: by position); four members
            //    per step, their LDS reads in flight together (a bucket holds one or two points on average)
#pragma unroll
            for (int q = 0; q < EPT; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t b = bs[q] & 8191u, pos = bs[q] >> 13;
                    const uint32_t lo = hist[b], hi = hist[b + 1];
                    const uint64_t mine = d2u(m[q]);
                    uint32_t rank = lo;
                    for (uint32_t j = lo; j < hi; j += 4) {
                        uint64_t kj[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) kj[u] = key[min(j + u, hi - 1)];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            rank += (j + u < hi) & ((kj[u] < mine) | ((kj[u] == mine) & (j + u < pos)));
                    }
                    bs[q] = rank;
                }
            __syncthreads();
            ESTAMP(5);
            // 7. final placement, keys converted to F(distance) for single-weight-function configurations
            bool nz = false;
#pragma unroll
            for (int q = 0; q < EPT; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t rank = bs[q];
                    nz |= (sg == 0 && rank == 0u && m[q] != 0.0);  // src/locohd.rs:74-77, on the distance
                    key[rank] = env.cdf_keys ? d2u(cdf_lean(wf.kind, prm, wf.n_params, winv, m[q]) + 0.0) : d2u(m[q]);
                    val[rank] = cat_of(q);
                }
            if (__ballot(nz) && lane == 0) atomicOr(&st->flags, ST_FIRST_NOT_ZERO);
            __syncthreads();
        }
        ESTAMP(6);
        {   // 8. write-out; categories outside the map: reported here, stored as 0 (see k_env_cells).  F is monotone, but its
            //    floating-point evaluation may produce a last-bit inversion between neighbours: looked for on the way out
            //    (the keys are being read anyway) and, in the rare case, repaired by a running maximum and written again.
            const int seg_n = (int)seg_n_s, C = cfg.n_categories;
            bool inv = false;
            for (int i = tl; i < seg_n; i += NT) {
                const uint8_t v = val[i];
                const uint64_t k = key[i];
                bad_c |= (int)v >= C;
                if (env.cdf_keys) inv |= k < (i ? key[i - 1] : carry_key_s);
                ok_[seg_base + i] = k;
                oc_[seg_base + i] = (int)v < C ? v : (uint8_t)0;
            }
            if (__syncthreads_or(inv ? 1 : 0)) {
                if (tl == 0) {
                    uint64_t mx = carry_key_s;
                    for (int i = 0; i < seg_n; ++i) { mx = key[i] > mx ? key[i] : mx; key[i] = mx; }
                }
                __syncthreads();
                for (int i = tl; i < seg_n; i += NT) ok_[seg_base + i] = key[i];
                __syncthreads();
            }
            if (tl == 0 && seg_n > 0) carry_key_s = key[seg_n - 1];
            seg_base += seg_n;
            __syncthreads();
        }
        ESTAMP(7);
    }
    if (tid == 0) env.len[r] = n;
    if (__ballot(bad_c) && lane == 0) atomicOr(&st->flags, ST_BAD_CATEGORY);
}

bool launch_env_rows2(hipStream_t s, const DevConfig* cfg, const RowSide& a, const RowSide& b, int64_t n_rows, DeviceStatus* st) {
    if (n_rows <= 0) return true;
    const int64_t longest = std::max(a.row_len, b.row_len);
    if (longest > 20480 || a.row_len < 1 || b.row_len < 1) return false;
    int n2 = 64;
    while (n2 < longest && n2 < 16384) n2 <<= 1;
    // rows of 16385 .. 20480 points: sorted in distance segments of ~10^4 points each (the segment's keys in LDS)
    const int n_seg = longest > 16384 ? 2 : 1;
    if (n_seg > 1 && (std::min(a.row_len, b.row_len) <= 16384)) return false;  // (one launch, one segment count: both sides must be long)
    if (n_seg > 1 && (a.row_lens || b.row_lens)) return false;  // (ragged rows may be short: same reason)
    RowSides sides;
    sides.s[0] = a; sides.s[1] = b;
    if (a.dmx) sides.s[0].image_bound = 0.0;  // given rows: the kernel finds the largest finite entry itself
    if (b.dmx) sides.s[1].image_bound = 0.0;
    sides.n_rows = n_rows;
    const dim3 grid((unsigned)(2 * n_rows));
    const size_t lds = (size_t)n2 * 9 + 16 + (size_t)(kRowBucketsSmall + 1) * sizeof(uint32_t);
    if (n_seg > 1) k_env_rows2<1024, 12, 2><<<grid, 1024, kRowSegLds, s>>>(cfg, sides, n2, st);
    else if (n2 <= 1024) k_env_rows2<64, 16, 1><<<grid, 64, lds, s>>>(cfg, sides, n2, st);
    else if (n2 <= 4096) k_env_rows2<256, 16, 1><<<grid, 256, lds, s>>>(cfg, sides, n2, st);
    else if (n2 <= 8192) k_env_rows2<1024, 8, 1><<<grid, 1024, lds, s>>>(cfg, sides, n2, st);
#ifdef LCHD_ROWS_NT512
    else if (longest <= 10240) k_env_rows2<512, 20, 1><<<grid, 512, lds, s>>>(cfg, sides, n2, st);
#endif
    else if (longest <= 10240) k_env_rows2<1024, 10, 1><<<grid, 1024, lds, s>>>(cfg, sides, n2, st);
    else k_env_rows2<1024, 16, 1><<<grid, 1024, lds, s>>>(cfg, sides, n2, st);
    return true;
}

bool launch_env_rows(hipStream_t s, int cap, const DevConfig* cfg, const CloudView& c, const double* dmx, int64_t ld,
                     int64_t n_rows, int64_t row_len, double image_bound, EnvStore env, DeviceStatus* st, const RowExtras& ex) {
    if (dmx) image_bound = 0.0;  // given rows: the kernel finds the largest finite entry itself
    if (n_rows <= 0) return true;
    if (row_len > cap || cap > (1 << 23)) return false;
    const dim3 grid((unsigned)n_rows);
    if (cap > 65536) {  // rows of more than 65 535 points (swept by k_sweep_wide<.., BIG>): keys in the store, 32768 buckets (128 KB of LDS)
        if (env.cat16) return false;
        const size_t lds = (size_t)(kRowBucketsHuge + 1) * sizeof(uint32_t) + 16;
        k_env_rows<1024, true><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsHuge, image_bound, env, st, ex);
        return true;
    }
    if (env.cat16) {  // more than 255 categories: two bytes per point (rows of up to 8192 points in LDS, longer ones in the store)
        if (ex.row_cat) return false;
        if (cap > 8192) {
            const size_t lds = (size_t)(kRowBucketsBig + 1) * sizeof(uint32_t) + 16;
            k_env_rows<1024, true, uint16_t><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsBig, image_bound, env, st, ex);
        } else {
            const size_t lds = (size_t)cap * 10 + 16 + (size_t)(kRowBucketsSmall + 1) * sizeof(uint32_t);
            if (cap <= 1024) k_env_rows<64, false, uint16_t><<<grid, 64, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
            else k_env_rows<1024, false, uint16_t><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
        }
        return true;
    }
    if (cap > 16384) {  // keys in global memory, 64 KB histogram in LDS
        const size_t lds = (size_t)(kRowBucketsBig + 1) * sizeof(uint32_t) + 16;
        k_env_rows<1024, true><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsBig, image_bound, env, st, ex);
        return true;
    }
    const size_t lds = (size_t)cap * 9 + 16 + (size_t)(kRowBucketsSmall + 1) * sizeof(uint32_t);
    if (cap <= 1024) {
        k_env_rows<64, false><<<grid, 64, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
    } else if (cap <= 4096) {
        k_env_rows<256, false><<<grid, 256, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
    } else {
        k_env_rows<1024, false><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
    }
    return true;
}

void init_env_rows_kernels() {
    auto raise = [](const void* fn, int bytes) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, true>), (int)((kRowBucketsHuge + 1) * sizeof(uint32_t) + 16));
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, false>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, true, uint16_t>), (int)((kRowBucketsBig + 1) * sizeof(uint32_t) + 16));
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, false, uint16_t>), 8192 * 10 + 16 + (kRowBucketsSmall + 1) * 4);
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 16, 1>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 10, 1>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
#ifdef LCHD_ROWS_NT512
    raise(reinterpret_cast<const void*>(&k_env_rows2<512, 20, 1>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
#endif
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 12, 2>), (int)kRowSegLds);
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 8, 1>), 8192 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
    (void)hipGetLastError();
}

}  // namespace lchd

#ifdef LCHD_SWEEP_STAMPS
extern "C" int lchd_debug_rows_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lchd::g_rows_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lchd::g_rows_stamps), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif

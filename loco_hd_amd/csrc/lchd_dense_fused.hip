// lchd_dense_fused.hip -- dense rows (from_coords / from_dmxs): sort AND sweep in one kernel, nothing but the score leaves the CU.
//
// Replaces, for the common configuration (Hellinger-2, unit category weights, at most 16 categories, rows of 1 025 .. 20 480
// points), the pair  k_env_rows2 (sort every row of both structures, write 9 bytes per point to the environment store)  +
// k_sweep (read them back, merge, integrate)  of lchd_env_rows.hip / lchd_sweep.hip:
//
//   reference                                     here
//   calculate_distance_matrix  utils.rs:10-22     ONE pass over the row pair's distances (plus a quarter of a pass for the plan)
//   sort_together              utils.rs:25-39     bucket sort of the UNION of both rows' points, one distance segment at a time
//   stat_dist_integral         locohd.rs:61-226   the sorted segment is integrated where it lies (LDS), counts carried on
//   PMFSystem / hellinger      pmf.rs, statistical_distances.rs:4-10   O(1) update of the Bhattacharyya sum per event
//
// A launch has 512 workgroups of 512 threads (two per CU, 78 KB of LDS each); a workgroup takes row pair blockIdx (row r of A and
// row r of B), then whatever row pair a ticket counter hands out next, and owns one scratch region in global memory.
//
// Why the union: the reference's two-pointer loop integrates  S = sum_k [F(t_k+1) - F(t_k)] H(after k events)  over the merged
// order of both lists; equal distances give zero-width intervals whose H never counts, so ANY order among equal keys gives
// the same sum -- sorting the two rows together (key, side | category) is that merged order without a merge step.  The two
// "anchors" (first element of each sorted row, src/locohd.rs:82-84) need no special case either: both rows start with a
// distance of 0 (checked: :74-77), every interval before the last zero-distance event has width F(0) - F(0) = 0, and H is
// only evaluated once both sides hold a point.
//
// Why segments, and how a row pair is cut into them (round 5): 2 x 10^4 points x (8-byte key + 1-byte side | category) do not fit
// 80 KB.  The coarse CDF of the row pair (512 bins of the distance image) is ESTIMATED from every 4th point; it cuts the distance
// axis into S balanced segments at bin boundaries, each planned with a margin of six standard deviations of what such a sample
// predicts, so that the real segment fits the LDS arrays (kCap = 6144 events).  Then ONE pass over both rows computes every
// squared distance once and appends the point -- one returning LDS atomic on its segment's fill count -- to its segment: segment 0
// straight into the LDS arrays, the others into the workgroup's scratch region (1.35 GB written and read back per 10^4-atom
// call: 1 TB/s, a quarter of what the fabric delivers to this kernel's access pattern, and the price of not recomputing every
// distance per segment as round 4 did -- three passes at 10^4 atoms, six at 2 x 10^4).  A segment then comes back into the sort's
// registers (coalesced), is bucket-sorted (interpolated rank on the sample's CDF -> 8192 buckets, ranks inside a bucket on the
// exact (key, value) pair: deterministic whatever order the distance pass produced), and swept with the category counts, totals
// and the last (F, H) carried from the previous segment.  The bin of a point is a non-decreasing function of its exact key, so
// segments are exact key ranges and the concatenation of the sorted segments is the sorted union.
//
// Rows this kernel gives up on (a bucket of more than 64 points: thousands of identical distances; a segment that outgrew its plan:
// a sample that does not represent the row -- six standard deviations; a row pair that needs more segments than its scratch region
// holds) are reported as ST_ROW_RETRY and the host repeats the call with the two-kernel path.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "lchd_kcommon.h"

#define LCHD_AS4 __attribute__((address_space(4)))  // the constant address space: kernel arguments, the configuration blob

namespace lchd {

namespace {

// Every phase of the kernel RE-LOADS the wave-uniform values it needs (structure pointers, anchor coordinates, weight-function
// parameters) from the constant address space through a pointer made opaque by an empty asm: held live across the whole row
// loop they exceed the scalar register file, and a spilled scalar costs a v_readlane at every use (lchd_env_group.hip).
template <class T>
__device__ __forceinline__ const LCHD_AS4 T* df_opaque(const LCHD_AS4 T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
template <class T>
__device__ __forceinline__ const LCHD_AS4 T* df_const(const T* p) {  // memory that no kernel of the pass writes
    return (const LCHD_AS4 T*)(unsigned long long)p;
}

#ifndef LCHD_DF_NT
#define LCHD_DF_NT 512     // threads per workgroup (tuning builds: 1024 with LCHD_DF_CAP 14336 and LCHD_DF_BUCKETS 8192 = one workgroup per CU)
#endif
#ifndef LCHD_DF_CAP
#define LCHD_DF_CAP 6144
#endif
#ifndef LCHD_DF_BUCKETS
#define LCHD_DF_BUCKETS 8192
#endif
constexpr int kNT = LCHD_DF_NT, kWaves = kNT / 64;
constexpr int kCap = LCHD_DF_CAP;     // events of one distance segment (keys + values in LDS)
constexpr int kEpt = kCap / kNT;      // 14 events per thread in the sort and in the sweep (<= 15: 4-bit chunk-local counters)
#ifndef LCHD_DF_SAMPLE
#define LCHD_DF_SAMPLE 4
#endif
constexpr int kSample = LCHD_DF_SAMPLE;  // the coarse CDF of a row pair is estimated from every kSample-th point of either row
constexpr int kCoarse = 512;          // uniform bins of the distance image; bin kCoarse holds +inf entries of a distance matrix
#ifndef LCHD_DF_GRID
#define LCHD_DF_GRID 512              // workgroups of a launch (two per CU): each owns one scratch region and takes rows blockIdx, + grid, ...
#endif
// events a segment may be PLANNED to hold: the plan sees a sample, the segment must hold the real thing (six standard deviations of the
// count that a sample of one in kSample predicts, and a coarse bin's worth)
__host__ __device__ inline int df_seg_margin(int est) { return 6 * (int)sqrtf((float)(est * kSample)) + 70; }
constexpr int kBuckets = LCHD_DF_BUCKETS;  // buckets of a segment's sort, two 16-bit counters per LDS word
constexpr int kBucketLimit = 64;      // a fuller bucket sends the call to the two-kernel path
constexpr int kPart = kBuckets / 4;   // sqrt(k) for k < kPart from LDS (the histogram's bytes), larger counts are computed
#ifndef LCHD_DF_MAXSEG
#define LCHD_DF_MAXSEG 16
#endif
constexpr int kMaxSeg = LCHD_DF_MAXSEG;  // (<= 64: the plan gives every boundary a lane)
constexpr double kExactBelow = 1e-6;  // as lchd_team_tile.h (kExactH2Below): below this H^2 the literal difference-of-roots form
static_assert(kEpt <= 15 && kCap % kNT == 0, "chunk-local counters are 4-bit fields");
static_assert((kBuckets / 2) % (4 * kNT) == 0 && kBuckets <= 8192, "the bucket scan gives every thread whole 16-byte groups of histogram words; 13-bit bucket ids");
static_assert(kBuckets * 2 >= kPart * 8, "histogram and sqrt table share their bytes");

constexpr int kKeyPad = 4;           // sentinel keys behind a segment's last event (the ranking reads four members from a bucket's start)
constexpr size_t kValOff = (size_t)(kCap + kKeyPad) * 8, kHistOff = kValOff + (size_t)kCap;
static_assert(kHistOff % 16 == 0, "the bucket scan reads 16-byte groups");
constexpr size_t kDynLds = kHistOff + (size_t)kBuckets * 2 + 16;  // (+ the histogram's end word)

__device__ __forceinline__ uint64_t df_spread4(uint64_t x) {  // four 4-bit fields -> four 16-bit fields
    const uint32_t v = (uint32_t)x;
    const uint32_t lo = (v & 0xFu) | ((v & 0xF0u) << 12);
    const uint32_t hi = ((v >> 8) & 0xFu) | ((v & 0xF000u) << 4);
    return ((uint64_t)hi << 32) | lo;
}
// sqrt for moderate x (counts, H^2): v_rsq_f64 seed + Goldschmidt, within 1 ulp; sqrt(0) = 0
__device__ __forceinline__ double df_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return fmax(g, 0.0);
}
__device__ __forceinline__ double df_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x, y * y, 1.5);
    y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}

// a value every lane holds alike, moved to scalar registers (loads through non-restrict pointers land in vector registers)
__device__ __forceinline__ double df_uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int df_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// exp(x) for x <= 0 as exp_nonpos (lchd_kcommon.h), the 64-entry table of 2^(j/64) in LDS
__device__ __forceinline__ double df_exp_nonpos(double x, const double* tab) {
    const double nd = rint(x * 0x1.71547652b82fep+6);
    double r = fma(-nd, 0x1.62e42fefa39efp-7, x);
    r = fma(-nd, 0x1.abc9e3b39803fp-62, r);
    const int n = (int)nd;
    const double t = tab[n & 63];
    double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = q * r;
    const double v = ldexp(fma(t, q, t), n >> 6);
    return x < -746.0 ? 0.0 : v;
}

// the row's weight function in registers (hyper_exp with <= 4 terms and uniform inline, the pow-based ones through a call)
struct DfWf {
    int kind, np, nterm;
    bool fast;
    double a[4], b[4], inv;
    const double* p;
};
__device__ __forceinline__ DfWf df_wf_load(const LCHD_AS4 DevConfig* kc, int wfi) {  // (scalar loads: the configuration blob is constant)
    DfWf w;
    const LCHD_AS4 WfEntry* e = df_const(kc->wf) + wfi;
    const LCHD_AS4 double* p = df_const(kc->wf_params) + e->offset;
    w.inv = df_const(kc->wf_inv)[wfi]; w.kind = e->kind; w.np = e->n_params; w.nterm = w.np / 2; w.p = kc->wf_params + e->offset;
    w.fast = (w.kind == WF_UNIFORM) || (w.kind == WF_HYPER_EXP && w.nterm <= 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { w.a[i] = 0.0; w.b[i] = 0.0; }
    if (w.kind == WF_UNIFORM) { w.a[0] = p[0]; w.a[1] = p[1]; }
    else if (w.fast) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < w.nterm) { w.a[i] = p[i]; w.b[i] = p[w.nterm + i]; }
    }
    return w;
}
__device__ __forceinline__ double df_cdf(const DfWf& w, double x, const double* exp_tab) {
    if (w.kind == WF_UNIFORM) {  // cdfs.rs:39-45
        if (x < w.a[0]) return 0.0;
        if (x > w.a[1]) return 1.0;
        return (x - w.a[0]) * w.inv;
    }
    if (w.fast) {  // cdfs.rs:5-21, same accumulation order
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < w.nterm) sum += w.a[i] * df_exp_nonpos(-w.b[i] * x, exp_tab);
        return 1.0 - sum * w.inv;
    }
    return cdf_pow_based(w.kind, w.p, w.np, x);
}

}  // namespace

// Diagnostic build only (-DLCHD_DF_STAMPS, never the shipped library): s_memtime deltas per phase as wavefront 0 of every
// workgroup sees them (barrier waits included), read back with lchd_debug_dense_stamps().
#ifdef LCHD_DF_STAMPS
__device__ unsigned long long g_df_stamps[16];
// (accumulated in LDS and flushed once per workgroup: global atomics per phase sit in front of every wait for the phase's own loads)
#define DSTAMP(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dstamp_acc[i] += t_ - dstamp_last; dstamp_last = t_; } } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif

// CMAX: category slots (8, 12, 16); DMX: the rows are given distances (from_dmxs), otherwise they come from coordinates
template <int CMAX, bool DMX>
__global__ __launch_bounds__(kNT, 4) void k_dense_fused(DenseArgs a) {
    constexpr int NW = CMAX / 4;  // u64 words of four 16-bit count fields per side
    static_assert(CMAX % 4 == 0 && CMAX <= 16, "one u64 of 4-bit chunk counters per side");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* skey = reinterpret_cast<uint64_t*>(smem);                  // [kCap + kKeyPad] key bits of the current segment
    uint8_t* sval = smem + kValOff;                                      // [kCap] side << 7 | category
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + kHistOff);       // [kBuckets / 2 + 1] (sort)
    double* t_part = reinterpret_cast<double*>(smem + kHistOff);         // [kPart] sqrt(k) (sweep): the same bytes
    __shared__ uint32_t coarse[kCoarse + 1], cum[kCoarse + 2];  // the SAMPLE's counts per coarse bin / below a bin
    __shared__ int bnd[kMaxSeg + 1];
    __shared__ uint8_t segtab[kCoarse + 1];                      // coarse bin -> distance segment
    __shared__ uint32_t fill32[kMaxSeg];                         // events of the segments
    __shared__ int n_seg_s;
    __shared__ uint32_t rowflags_s, wsum[kWaves], wbig[kWaves], wtot_a[kWaves];
    __shared__ uint64_t wtot[kWaves][2 * NW], base_cnt[2][2 * NW];
    __shared__ double st_f[kWaves], st_h[kWaves], carry_f[2], carry_h[2], red_s[kWaves], red_max[kWaves], exp_tab[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // kernel arguments and configuration in the constant address space (scalar loads, re-issued per phase: df_opaque)
    const LCHD_AS4 DenseArgs* const ka = (const LCHD_AS4 DenseArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const LCHD_AS4 DevConfig* const kc = df_const(a.cfg);
    enum : uint32_t { RF_ZERO_A = 1u, RF_ZERO_B = 2u, RF_BAD_DIST = 4u, RF_BAD_CAT = 8u, RF_RETRY = 16u };
    // one side's inputs
    struct SideIn {
        const double *row, *x, *y, *z;
        const uint8_t* cat;
        double ax, ay, az;
        int n;
    };

    if (tid < 64) exp_tab[tid] = kExp2Tab[tid];  // (made visible by the row loop's first barrier)
#ifdef LCHD_DF_STAMPS
    __shared__ unsigned long long dstamp_acc[16];
    if (tid < 16) dstamp_acc[tid] = 0ull;
    __syncthreads();
    unsigned long long dstamp_last = __builtin_amdgcn_s_memtime();
#endif
    // rows: blockIdx first, then whatever row the ticket counter hands out next (the launch has a fixed number of workgroups, one per
    // scratch region; k_dense_publish zeroes the counter again)
    __shared__ long long next_row_s[2];  // (two slots: a thread may still have to read row i's successor when thread 0 fetches row i + 1's)
    int it = 0;
    for (int64_t r = blockIdx.x; r < a.n_rows; r = next_row_s[it & 1], ++it) {
        int nA, nB, wfi;
        {
            const LCHD_AS4 DenseArgs* kp = df_opaque(ka);
            nA = kp->s[0].row_lens ? df_const(kp->s[0].row_lens)[r] : kp->s[0].row_len;
            nB = kp->s[1].row_lens ? df_const(kp->s[1].row_lens)[r] : kp->s[1].row_len;
            wfi = kp->wf_index ? df_const(kp->wf_index)[r] : 0;
        }
        const int total = nA + nB;
        const bool bad_wf = wfi < 0 || wfi >= kc->n_wf;
        if (bad_wf) wfi = 0;
        uint32_t myflags = 0u;
        // (side is workgroup-uniform; everything below is a scalar load or a scalar select)
        auto side_in = [&](int side) -> SideIn {
            const LCHD_AS4 DenseSide* ks = &df_opaque(ka)->s[side];
            SideIn si;
            si.n = side ? nB : nA;
            si.cat = ks->c.cat;
            si.row = nullptr; si.x = si.y = si.z = nullptr;
            si.ax = si.ay = si.az = 0.0;
            if constexpr (DMX) {
                si.row = ks->dmx + r * ks->ld;
            } else {
                si.x = ks->c.x; si.y = ks->c.y; si.z = ks->c.z;
                si.ax = df_const(si.x)[r]; si.ay = df_const(si.y)[r]; si.az = df_const(si.z)[r];  // from_coords: anchor r of either structure
            }
            return si;
        };
        // the key of point i of a side: d^2 from the coordinates (utils.rs:1-8 order, uncontracted: this translation unit is
        // compiled with -ffp-contract=off) or the given distance
        auto key_of = [&](const SideIn& si, int i) -> double {
            if constexpr (DMX) {
                double v = si.row[i];
                if (!(v >= 0.0)) { myflags |= RF_BAD_DIST; v = 0.0; }  // negative or NaN
                return v + 0.0;                                          // -0.0 -> +0.0
            } else {
                const double dx = si.ax - si.x[i], dy = si.ay - si.y[i], dz = si.az - si.z[i];
                double d2 = dx * dx;
                d2 = d2 + dy * dy;
                d2 = d2 + dz * dz;
                return d2;
            }
        };
        // ---- 0. upper bound of the finite images -----------------------------------------------------------------------
        double dmax = df_opaque(ka)->image_bound;
        if constexpr (DMX) {
            dmax = 0.0;
            for (int side = 0; side < 2; ++side) {
                const SideIn si = side_in(side);
                for (int i = tid; i < si.n; i += kNT) {
                    const double v = si.row[i];
                    if (v < 1.0e300 && v > dmax) dmax = v;
                }
            }
            for (int k = 32; k > 0; k >>= 1) dmax = fmax(dmax, shfl_xor_f64(dmax, k));
            if (lane == 0) red_max[wave] = dmax;
        }
        for (int b = tid; b <= kCoarse; b += kNT) coarse[b] = 0u;
        if (tid < 2 * NW) base_cnt[0][tid] = 0ull;
        if (tid == 0) { rowflags_s = 0u; carry_f[0] = carry_f[1] = 0.0; carry_h[0] = carry_h[1] = 0.0; }
        __syncthreads();
        if constexpr (DMX)
            for (int w = 0; w < kWaves; ++w) dmax = fmax(dmax, red_max[w]);
        const float inv_wf = dmax > 0.0 ? (float)((double)kCoarse / dmax) : 0.0f;
        // coarse bin of a key: never decreases with the key (rounding to float, the product with a non-negative constant and
        // the truncation all are monotone); entries beyond the bound (+inf in a distance matrix) take the extra bin
        auto bin_of = [&](double key) -> int {
            const int b = min((int)((float)key * inv_wf), kCoarse - 1);
            return key <= dmax ? b : kCoarse;
        };
        // ---- 1. the row pair's CDF on the coarse bins, estimated from every kSample-th point -------------------------------
        // (a row pair that fits one segment needs no plan, only balanced buckets: every 2nd point then -- 0.319 -> 0.299 ms at 3000 atoms)
        const int smp = total <= kCap ? 2 : kSample;
        uint32_t m_tot = 0;  // points of the sample
        for (int side = 0; side < 2; ++side) {
            const SideIn si = side_in(side);
            const int off = (int)((r + side) & (smp - 1));
            m_tot += (uint32_t)((si.n - off + smp - 1) / smp);
            for (int i = smp * tid + off; i < si.n; i += smp * kNT) atomicAdd(&coarse[bin_of(key_of(si, i))], 1u);
        }
        if (tid < kMaxSeg) fill32[tid] = 0u;
        __syncthreads();
        DSTAMP(0);
        // (the next row's ticket is fetched by a wavefront that would otherwise wait for the plan)
        if (tid == kNT - 1) next_row_s[it & 1] = (long long)gridDim.x + (long long)atomicAdd(df_opaque(ka)->ticket, 1ull);
        if (wave == 0) {  // cum[b] = sampled points below bin b; then the segment plan
            uint32_t carry = 0;
            for (int base = 0; base <= kCoarse; base += 64) {
                const uint32_t v = base + lane <= kCoarse ? coarse[base + lane] : 0u;
                const uint32_t incl = wave_incl_scan_u32(v);
                if (base + lane <= kCoarse) cum[base + lane] = carry + incl - v;
                carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
            if (lane == 0) cum[kCoarse + 1] = carry;
            wave_sync_lds();
            // S segments at bin boundaries: boundary s = the first bin at which the sample's CDF reaches s / S; a segment is accepted
            // when the events the sample predicts for it, plus the margin, fit the LDS arrays (one segment: whatever fits fits)
            const float per = (float)total / (float)max(m_tot, 1u);  // events per sampled point
            const int s_max = min(df_opaque(ka)->scr_segs, kMaxSeg);
            int S = max(1, (total + kCap - 1) / kCap);
            for (;; ++S) {
                if (S > s_max) { S = 0; break; }
                int b = lane >= S ? kCoarse + 1 : 0;
                if (lane >= 1 && lane < S) {
                    const uint32_t tgt = (uint32_t)(((uint64_t)lane * (uint64_t)m_tot) / (uint64_t)S);
                    int lo = 0, hi = kCoarse + 1;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (cum[mid] >= tgt) hi = mid; else lo = mid + 1;
                    }
                    b = lo;
                }
                const int bn = __shfl_down(b, 1);
                const int est = lane < S ? (int)((float)(cum[bn] - cum[b]) * per) + 1 : 0;
                const bool fits = S == 1 ? total <= kCap : est + df_seg_margin(est) <= kCap;
                if (__ballot(lane < S && !fits) == 0ull) {
                    if (lane <= S) bnd[lane] = b;
                    break;
                }
            }
            if (lane == 0) n_seg_s = S;
        }
        __syncthreads();
        const int S = n_seg_s;
        for (int b = tid; b <= kCoarse; b += kNT) {
            int sg = 0;
            for (int k = 1; k < S; ++k) sg += b >= bnd[k] ? 1 : 0;
            segtab[b] = (uint8_t)sg;
        }
        __syncthreads();
        DSTAMP(1);
        // ---- 2. ONE pass over the distances: every point goes to its segment's scratch array ---------------------------------
        if (S > 0) {
            const size_t my_scr = (size_t)blockIdx.x * (size_t)df_opaque(ka)->scr_segs * (size_t)kCap;
            uint64_t* const gkey = df_opaque(ka)->scr_key + my_scr;
            uint8_t* const gval = df_opaque(ka)->scr_val + my_scr;
            const int n_cat = kc->n_categories;
            for (int side = 0; side < 2; ++side) {
                const SideIn si = side_in(side);
                const int ns = si.n;
                bool zero = false, badc = false;
                for (int i0 = 0; i0 < ns; i0 += 4 * kNT) {
                    double k[4];
                    uint32_t sg[4];
                    uint8_t v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * kNT + tid;
                        const int ii = i < ns ? i : 0;
                        k[u] = key_of(si, ii);
                        const uint32_t c = si.cat[ii];
                        zero |= (k[u] == 0.0);
                        badc |= (int)c >= n_cat;  // pmf.rs:38-42: every point of a row enters a PMF
                        sg[u] = i < ns ? (uint32_t)segtab[bin_of(k[u])] : 0xFFu;
                        v[u] = (uint8_t)((c & 15u) | (uint32_t)(side << 7));
                    }
                    if (S == 1) {  // (workgroup-uniform) one segment: ballot compaction straight into the LDS arrays
                        unsigned long long bm[4];
                        uint32_t cnt = 0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) { bm[u] = __ballot(sg[u] != 0xFFu); cnt += (uint32_t)__popcll(bm[u]); }
                        uint32_t at = 0;
                        if (lane == 0 && cnt) at = atomicAdd(&fill32[0], cnt);
                        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (sg[u] != 0xFFu) {
                                const uint32_t pos = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm[u], 0u));
                                if (pos < (uint32_t)kCap) { skey[pos] = d2u(k[u]); sval[pos] = v[u]; }
                            }
                            at += (uint32_t)__popcll(bm[u]);
                        }
                    } else {  // several: one returning LDS atomic per point on its segment's fill count (measured against a wave scan of
                              // one-hot count fields with one atomic per wavefront: 11.58 -> 10.83 ms at 2 x 10^4 atoms, 2.80 -> 2.76 at 10^4)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (sg[u] != 0xFFu) {
                                const uint32_t pos = atomicAdd(&fill32[sg[u]], 1u);
                                if (pos < (uint32_t)kCap) {
                                    if (sg[u] == 0u) {  // the first segment is sorted first: it waits in the LDS arrays themselves
                                        skey[pos] = d2u(k[u]);
                                        sval[pos] = v[u];
                                    } else {
                                        const size_t at = (size_t)sg[u] * (size_t)kCap + pos;
                                        gkey[at] = d2u(k[u]);
                                        gval[at] = v[u];
                                    }
                                }
                            }
                    }
                }
                if (zero) myflags |= side ? RF_ZERO_B : RF_ZERO_A;
                if (badc) myflags |= RF_BAD_CAT;
            }
        }
        for (int k = 32; k > 0; k >>= 1) myflags |= (uint32_t)__shfl_xor((int)myflags, k);
        if (lane == 0 && myflags) atomicOr(&rowflags_s, myflags);
        __syncthreads();  // (orders the scratch writes before the segment loop's reads: one workgroup, one CU)
        DSTAMP(2);
        uint32_t rf = rowflags_s;
        if (S == 0) rf |= RF_RETRY;
        for (int sg = 0; sg < S; ++sg)
            if (fill32[sg] > (uint32_t)kCap) rf |= RF_RETRY;  // the sample under-estimated a segment
        if (!(rf & RF_ZERO_A) || !(rf & RF_ZERO_B) || (rf & (RF_BAD_DIST | RF_BAD_CAT | RF_RETRY)) || bad_wf) {
            // src/locohd.rs:74-77 (the sorted rows must start with a distance of 0), pmf.rs:38-42, a NaN / negative entry,
            // a row pair this kernel cannot segment: reported, the host turns the flags into the reference's errors
            if (tid == 0) {
                uint32_t f = 0u;
                if (!(rf & RF_ZERO_A) || !(rf & RF_ZERO_B)) f |= ST_FIRST_NOT_ZERO;
                if (rf & RF_BAD_DIST) f |= ST_BAD_DISTANCE;
                if (rf & RF_BAD_CAT) f |= ST_BAD_CATEGORY;
                if (rf & RF_RETRY) f |= ST_ROW_RETRY;
                if (bad_wf) f |= ST_BAD_WF;
                atomicOr(&a.st->flags, f);
                a.out[r] = nan("");
            }
            __syncthreads();
            continue;
        }

        double acc = 0.0;           // this thread's share of the integral
        int base_na = 0, base_nb = 0;  // points of A / B in the segments already swept
        bool give_up = false;
        int swept = 0;  // non-empty segments swept so far (parity of the carry slot)
        // a segment's (key, value) pairs come back from the scratch arrays into the sort's registers (issuing the loads a phase or a
        // segment ahead was measured: the sort's 32 registers then live through the sweep, the allocator spills them, and a spill waits
        // for its load -- 3.08 -> 3.17 / 3.61 ms)
        double m[kEpt];
        uint32_t vv[(kEpt + 3) / 4];
        auto fill_of = [&](int sgi) -> int { return (int)fill32[sgi]; };
        auto load_segment = [&](int sgi) {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            const int nn = fill_of(sgi);
            const size_t seg_scr = ((size_t)blockIdx.x * (size_t)df_opaque(ka)->scr_segs + (size_t)sgi) * (size_t)kCap;
            const uint64_t* const gkey = df_opaque(ka)->scr_key + seg_scr;
            const uint8_t* const gval = df_opaque(ka)->scr_val + seg_scr;
#pragma unroll
            for (int q = 0; q < (kEpt + 3) / 4; ++q) vv[q] = 0u;
            if (sgi == 0) {  // (workgroup-uniform)
#pragma unroll
                for (int q = 0; q < kEpt; ++q) {
                    const int e = tl + q * kNT, ee = e < nn ? e : 0;
                    m[q] = u2d(skey[ee]);
                    vv[q >> 2] |= (uint32_t)sval[ee] << ((q & 3) * 8);
                }
            } else {
#pragma unroll
                for (int q = 0; q < kEpt; ++q) {
                    const int e = tl + q * kNT, ee = e < nn ? e : 0;
                    m[q] = u2d(gkey[ee]);
                    vv[q >> 2] |= (uint32_t)gval[ee] << ((q & 3) * 8);
                }
            }
        };
#pragma unroll 1
        for (int sg = 0; sg < S; ++sg) {
            const int lo_bin = bnd[sg], hi_bin = bnd[sg + 1];
            const uint32_t cum_lo = cum[lo_bin];
            const int n = fill_of(sg);  // events of this segment
            if (n == 0) continue;       // (workgroup-uniform)
            load_segment(sg);
            for (int w = tid; w < kBuckets / 2; w += kNT) hist[w] = 0u;
            // ---- 3. bucket sort ------------------------------------------------------------------------------------------
            // (an opaque copy of the thread index per phase: the slot indices tid + q * 512 are otherwise computed once at the
            //  kernel's entry, for every unrolled loop below, and spilled)
            int ts = tid;
            asm volatile("" : "+v"(ts));
            uint32_t bs[kEpt];
            __syncthreads();
            DSTAMP(3);
            const float scale = (float)kBuckets / (float)max(cum[hi_bin] - cum_lo, 1u);  // (sample counts: buckets per sampled point)
            uint32_t biggest = 0;
#pragma unroll
            for (int q = 0; q < kEpt; ++q) {
                bs[q] = ~0u;
                if (ts + q * kNT < n) {
                    // interpolated rank inside the segment -> bucket (single precision: it only has to be monotone and balanced)
                    const int j = bin_of(m[q]);
                    const float t = (float)m[q] * inv_wf;
                    const float frac = j < kCoarse ? fminf(fmaxf(t - (float)j, 0.0f), 1.0f) : 0.0f;
                    const float rk = ((float)(cum[j] - cum_lo) + frac * (float)coarse[j]) * scale;
                    const int b = rk < (float)kBuckets ? max((int)rk, 0) : kBuckets - 1;
                    const int sh = (b & 1) * 16;
                    const uint32_t old = atomicAdd(&hist[b >> 1], 1u << sh);
                    const uint32_t slot = (old >> sh) & 0xFFFFu;
                    bs[q] = (uint32_t)b | (slot << 13);
                    biggest = max(biggest, slot + 1u);
                }
            }
            for (int k = 32; k > 0; k >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, k));
            if (lane == 0) wbig[wave] = biggest;
            __syncthreads();
            for (int w = 0; w < kWaves; ++w) biggest = max(biggest, wbig[w]);  // (checked behind the scan's barrier: an over-full bucket does not hurt the scan)
            DSTAMP(3);
            {   // exclusive scan of the bucket counters: kScanW words (2 kScanW buckets) per thread
                constexpr int kScanW = kBuckets / 2 / kNT;
                uint32_t c[2 * kScanW];
#pragma unroll
                for (int g = 0; g < kScanW / 4; ++g) {
                    const uint4 w4 = *reinterpret_cast<const uint4*>(&hist[kScanW * tid + 4 * g]);
                    const uint32_t ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { c[8 * g + 2 * k] = ww[k] & 0xFFFFu; c[8 * g + 2 * k + 1] = ww[k] >> 16; }
                }
                uint32_t T = 0;
#pragma unroll
                for (int k = 0; k < 2 * kScanW; ++k) T += c[k];
                const uint32_t incl = wave_incl_scan_u32(T);
                if (lane == 63) wsum[wave] = incl;
                __syncthreads();
                if (biggest > (uint32_t)kBucketLimit) { give_up = true; break; }  // (workgroup-uniform)
                uint32_t run = incl - T;
                for (int w = 0; w < kWaves; ++w) run += w < wave ? wsum[w] : 0u;
#pragma unroll
                for (int g = 0; g < kScanW / 4; ++g) {
                    uint32_t o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t lo_ = run;
                        run += c[8 * g + 2 * k];
                        o[k] = lo_ | (run << 16);
                        run += c[8 * g + 2 * k + 1];
                    }
                    *reinterpret_cast<uint4*>(&hist[kScanW * tid + 4 * g]) = make_uint4(o[0], o[1], o[2], o[3]);
                }
                if (tid == 0) hist[kBuckets / 2] = (uint32_t)n;
            }
            __syncthreads();
            auto start_of = [&](uint32_t b) -> uint32_t {  // first position of bucket b (b == kBuckets: the end word = the segment's end)
                return (hist[b >> 1] >> ((b & 1u) * 16u)) & 0xFFFFu;
            };
            DSTAMP(4);
            // scatter: position = bucket start + the slot the histogram atomic returned
#pragma unroll
            for (int q = 0; q < kEpt; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t b = bs[q] & 8191u, pos = start_of(b) + (bs[q] >> 13);
                    skey[pos] = d2u(m[q]);
                    sval[pos] = (uint8_t)(vv[q >> 2] >> ((q & 3) * 8));
                    bs[q] = b | (pos << 13);
                }
            if (tid < kKeyPad) skey[n + tid] = ~0ull;  // (n <= kCap; larger than every key)
            __syncthreads();
            DSTAMP(5);
            // every event ranks itself among the members of its bucket on (key, value, position): the sequence of (key, value)
            // pairs that results does not depend on the order in which the distance pass happened to place the points.  The
            // first four members are read without a look at the bucket's end: what lies behind it belongs to later buckets (larger
            // keys: a bucket is a non-decreasing function of the key) or is a sentinel, and counts neither as smaller nor as equal.
            // Fuller buckets (one event in three hundred at 0.6 events per bucket) and equal keys (lattices, +inf entries) take the loops.
#pragma unroll
            for (int q = 0; q < kEpt; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t b = bs[q] & 8191u, pos = bs[q] >> 13;
                    const uint32_t w0 = hist[b >> 1], w1 = hist[(b >> 1) + 1u];
                    const uint32_t lo = (b & 1u) ? w0 >> 16 : w0 & 0xFFFFu, hi = (b & 1u) ? w1 & 0xFFFFu : w0 >> 16;
                    const uint64_t mine = d2u(m[q]);
                    uint64_t kj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) kj[u] = skey[lo + u];
                    uint32_t less = 0, same = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        less += kj[u] < mine ? 1u : 0u;
                        same += kj[u] == mine ? 1u : 0u;
                    }
                    for (uint32_t j = lo + 4u; j < hi; ++j) {
                        const uint64_t kk = skey[j];
                        less += kk < mine ? 1u : 0u;
                        same += kk == mine ? 1u : 0u;
                    }
                    if (same > 1u) {  // (itself and at least one other member): recount with the full order
                        const uint32_t myv = (vv[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
                        less = 0;
                        for (uint32_t j = lo; j < hi; ++j) {
                            const uint32_t vj = sval[j];
                            const uint64_t kk = skey[j];
                            less += (kk < mine) | ((kk == mine) & ((vj < myv) | ((vj == myv) & (j < pos)))) ? 1u : 0u;
                        }
                    }
                    bs[q] = lo + less;
                }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kEpt; ++q)
                if (bs[q] != ~0u) {
                    skey[bs[q]] = d2u(m[q]);
                    sval[bs[q]] = (uint8_t)(vv[q >> 2] >> ((q & 3) * 8));
                }
            __syncthreads();
            DSTAMP(6);
            // keys -> F(distance): the sweep needs nothing else of a point's distance, and here every lane converts (the event
            // loop is a chain of dependent steps per lane).  utils.rs:1-8: the distance is the root of the sum of squares.
            {
                // (one loop per kind of weight function -- the kind is uniform: a loop that may CALL the pow-based CDFs keeps the
                //  parameters of the inline ones in spilled scalar registers, a v_readlane per use and event)
                const DfWf wf = df_wf_load(df_opaque(kc), wfi);
                if (wf.kind == WF_HYPER_EXP && wf.fast) {  // cdfs.rs:5-21, same accumulation order
                    for (int p = tid; p < n; p += kNT) {
                        const double kv = u2d(skey[p]);
                        const double x = DMX ? kv : df_sqrt(kv);  // (within 1 ulp of utils.rs:1-8's powf(0.5): 1e-16 of F)
                        double sum = 0.0;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (i < wf.nterm) sum += wf.a[i] * df_exp_nonpos(-wf.b[i] * x, exp_tab);
                        skey[p] = d2u(1.0 - sum * wf.inv);
                    }
                } else {
                    for (int p = tid; p < n; p += kNT) {
                        const double kv = u2d(skey[p]);
                        skey[p] = d2u(df_cdf(wf, DMX ? kv : df_sqrt(kv), exp_tab));
                    }
                }
            }
            for (int k = tid; k < kPart; k += kNT) t_part[k] = df_opaque(ka)->sqrt_tab[k];  // (the histogram's bytes: no longer needed)
            __syncthreads();
            DSTAMP(7);
            // ---- 4. sweep the sorted segment -------------------------------------------------------------------------------
            const int epl = (n + kNT - 1) / kNT;  // events per lane (<= kEpt)
            const int d0 = min(tid * epl, n), d1 = min(d0 + epl, n);
            uint64_t hA = 0ull, hT = 0ull;  // 4-bit-per-category histograms of this lane's chunk: side A, both sides
            uint32_t n_al = 0;
            for (int e = 0; e < epl; ++e) {
                if (d0 + e < d1) {
                    const uint32_t v = sval[d0 + e];
                    const uint64_t inc = 1ull << ((v & 15u) * 4u);
                    hT += inc;
                    hA += v < 128u ? inc : 0ull;
                    n_al += v < 128u ? 1u : 0u;
                }
            }
            const uint64_t hB = hT - hA;
            uint64_t exA[NW], exB[NW];  // packed 16-bit category counts before this lane's first event
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint64_t va_ = df_spread4(hA >> (16 * k)), vb_ = df_spread4(hB >> (16 * k));
                const uint64_t sa_ = wave_incl_scan_fields(va_), sb_ = wave_incl_scan_fields(vb_);
                exA[k] = sa_ - va_;
                exB[k] = sb_ - vb_;
                if (lane == 63) { wtot[wave][k] = sa_; wtot[wave][NW + k] = sb_; }
            }
            const uint32_t sna = wave_incl_scan_u32(n_al);
            if (lane == 63) wtot_a[wave] = sna;
            __syncthreads();
            // exclusive prefix over the wavefronts on top of the previous segments' counts: lane k of every wavefront adds up word k
            // (no second barrier: the next segment's base goes to the other slot of base_cnt)
            const int par = swept & 1;
            uint64_t run = 0ull;
            uint32_t run_a = 0u, tot_a = 0u;
            if (lane < 2 * NW) {
                const uint64_t base = base_cnt[par][lane];
                uint64_t tot = 0ull;
                run = base;
                for (int w = 0; w < kWaves; ++w) { const uint64_t t = wtot[w][lane]; run += w < wave ? t : 0ull; tot += t; }
                if (wave == 0) base_cnt[par ^ 1][lane] = base + tot;
            } else if (lane == 2 * NW) {
                for (int w = 0; w < kWaves; ++w) { const uint32_t t = wtot_a[w]; run_a += w < wave ? t : 0u; tot_a += t; }
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) { exA[k] += readlane_u64(run, k); exB[k] += readlane_u64(run, NW + k); }
            const int a_before = (int)(sna - n_al) + __builtin_amdgcn_readlane((int)run_a, 2 * NW);
            int totA = base_na + a_before, totB = base_nb + (d0 - a_before);
            const int seg_a = __builtin_amdgcn_readlane((int)tot_a, 2 * NW);

            DSTAMP(8);
            auto sqrt_cnt = [&](int cnt) -> double { if (cnt < kPart) return t_part[cnt]; else return df_sqrt((double)cnt); };
            // Bhattacharyya sum from the exact integer counts at the start of the chunk (nothing drifts from chunk to chunk).
            // One copy of the two look-ups per count WORD (its four fields in a rolled loop): unrolled per category the
            // table-or-compute branches of sqrt_cnt were a third of the kernel's code.
            double D = 0.0;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                uint64_t wa = exA[k], wb = exB[k];
#pragma unroll 1
                for (int f = 0; f < 4; ++f) {
                    D += sqrt_cnt((int)(wa & 0xFFFFull)) * sqrt_cnt((int)(wb & 0xFFFFull));
                    wa >>= 16;
                    wb >>= 16;
                }
            }
            double ra = df_rsqrt((double)totA), rb = df_rsqrt((double)totB);
            uint64_t dA = 0ull, dB = 0ull;  // what the chunk has added so far, 4 bits per category
            auto exact_h2 = [&]() -> double {  // statistical_distances.rs:4-10, literal: equal inputs cancel exactly (rarely taken)
                double acc2 = 0.0;
                uint64_t qa = dA, qb = dB;
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    uint64_t wa = exA[k], wb = exB[k];
#pragma unroll 1
                    for (int f = 0; f < 4; ++f) {
                        const int ca = (int)(wa & 0xFFFFull) + (int)(qa & 15ull), cb = (int)(wb & 0xFFFFull) + (int)(qb & 15ull);
                        const double d = sqrt_cnt(ca) * ra - sqrt_cnt(cb) * rb;
                        acc2 = fma(d, d, acc2);
                        wa >>= 16; wb >>= 16; qa >>= 4; qb >>= 4;
                    }
                }
                return 0.5 * acc2;
            };
            double Fp = 0.0, Hp = 0.0, firstF = 0.0, local = 0.0;
            for (int e = 0; e < epl; ++e) {
                if (d0 + e < d1) {
                    const double F = u2d(skey[d0 + e]);
                    const uint32_t v = sval[d0 + e];
                    const int ct = (int)(v & 15u);
                    const bool takeA = v < 128u;
                    if (e == 0) firstF = F; else local = fma(F - Fp, Hp, local);
                    totA += takeA ? 1 : 0;
                    totB += takeA ? 0 : 1;
                    // pmf.rs:47-63: one more point of category ct on one side
                    const int sh = (ct & 3) * 16, sh4 = ct * 4;
                    uint64_t wA = exA[0], wB = exB[0];
#pragma unroll
                    for (int k = 1; k < NW; ++k) {
                        const bool hit = ((ct >> 2) == k);
                        wA = hit ? exA[k] : wA;
                        wB = hit ? exB[k] : wB;
                    }
                    const int cntA_ = (int)((wA >> sh) & 0xFFFFull) + (int)((dA >> sh4) & 15ull);
                    const int cntB_ = (int)((wB >> sh) & 0xFFFFull) + (int)((dB >> sh4) & 15ull);
                    const uint64_t inc4 = 1ull << sh4;
                    dA += takeA ? inc4 : 0ull;
                    dB += takeA ? 0ull : inc4;
                    const int mine = takeA ? cntA_ : cntB_, other = takeA ? cntB_ : cntA_;
                    double s0, s1, so;
                    if (max(mine + 1, other) < kPart) { s0 = t_part[mine]; s1 = t_part[mine + 1]; so = t_part[other]; }  // (one branch for the three look-ups)
                    else { s0 = df_sqrt((double)mine); s1 = df_sqrt((double)(mine + 1)); so = df_sqrt((double)other); }
                    D = fma(s1 - s0, so, D);
                    const double rr = df_rsqrt((double)(takeA ? totA : totB));
                    ra = takeA ? rr : ra;
                    rb = takeA ? rb : rr;
                    // H is only defined once both sides hold a point; every event before that sits at distance 0 (checked above),
                    // so the interval it would weigh has width F(0) - F(0) = 0
                    const bool both = (totA > 0) & (totB > 0);
                    double h2 = fma(-(ra * rb), D, 1.0);
                    if (both && h2 < kExactBelow) h2 = exact_h2();
                    Hp = both ? df_sqrt(h2) : 0.0;
                    Fp = F;
                }
            }
            DSTAMP(9);
            // stitch the chunks: (F_first - F_last of the previous chunk) * H before my first event
            const int last = (n - 1) / epl;  // the last thread that has events; every thread before it has a full chunk
            if (lane == 63 || tid == last) {
                if (lane == 63) { st_f[wave] = Fp; st_h[wave] = Hp; }
                if (tid == last) { carry_f[(swept + 1) & 1] = Fp; carry_h[(swept + 1) & 1] = Hp; }
            }
            __syncthreads();
            double prevF = wave_shr1_f64(Fp), prevH = wave_shr1_f64(Hp);
            if (lane == 0) {
                if (wave == 0) { prevF = carry_f[swept & 1]; prevH = carry_h[swept & 1]; }
                else { prevF = st_f[wave - 1]; prevH = st_h[wave - 1]; }
            }
            if (d0 < d1) local = fma(firstF - prevF, prevH, local);
            acc += local;
            base_na += seg_a;
            base_nb += n - seg_a;
            ++swept;
            DSTAMP(10);
        }
        if (give_up) {
            if (tid == 0) { atomicOr(&a.st->flags, ST_ROW_RETRY); a.out[r] = nan(""); }
            __syncthreads();
            continue;
        }
        // the last interval to +inf (:165-171, 204-210, 212-221) and the workgroup's sum
        acc = wave_sum_f64(acc);
        if (lane == 0) red_s[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < kWaves; ++w) sum += red_s[w];
            const int slot = swept & 1;  // the carry slot the last swept segment wrote
            sum = fma(df_const(kc->wf_finf)[wfi] - carry_f[slot], carry_h[slot], sum);
            a.out[r] = sum;
        }
        __syncthreads();
        DSTAMP(11);
    }
#ifdef LCHD_DF_STAMPS
    __syncthreads();
    if (tid < 16) atomicAdd(&g_df_stamps[tid], dstamp_acc[tid]);
#endif
}

// The end of a dense pass that did not run k_pair_meta: what the kernels reported goes into the host-mapped mirror, the
// device status is reset for the next pass (one thread; ordered behind the fused kernel by the stream).
__global__ void k_dense_publish(DeviceStatus* st, HostStatus* h, uint32_t seq, unsigned long long* ticket) {
    *ticket = 0ull;
    h->flags = st->flags;
    h->max_env = 0u;
    h->n_unique[0] = h->n_unique[1] = 0u;
    h->n_small = 0ull;
    h->n_duo = h->n_c8 = 0ull;
    h->snapshot_seq = seq;
    st->flags = 0u;
    st->max_env = 0u;
}

#ifdef LCHD_DF_STAMPS
extern "C" int lchd_debug_dense_stamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_df_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_df_stamps), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif

bool dense_fused_applies(int n_categories, int64_t len_a, int64_t len_b) {
    const int64_t longest = len_a > len_b ? len_a : len_b;
    return n_categories <= 16 && longest > kDenseFusedMinRow && longest <= kDenseFusedMaxRow && len_a >= 1 && len_b >= 1;
}

size_t dense_fused_scratch(int64_t n_rows, int64_t len_a, int64_t len_b, int32_t* grid_out, int32_t* segs_out) {
    const int64_t total = len_a + len_b;
    // the plan starts at ceil(total / kCap) segments and adds some until the sample's estimates fit with their margin: two more
    // than the balanced plan needs cover every row pair whose sample is not grossly off (otherwise: ST_ROW_RETRY)
    int S = (int)std::max<int64_t>(1, (total + kCap - 1) / kCap);
    while (S < kMaxSeg && (int)(total / S) + df_seg_margin((int)(total / S)) > kCap) ++S;
    const int segs = std::min(kMaxSeg, S + 2);
    const int grid = (int)std::min<int64_t>(n_rows, LCHD_DF_GRID);
    if (grid_out) *grid_out = grid;
    if (segs_out) *segs_out = segs;
    return (size_t)std::max(grid, 1) * (size_t)segs * (size_t)kCap;
}

void init_dense_fused_kernels() {
    auto raise = [](const void* fn) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynLds); };
    raise(reinterpret_cast<const void*>(&k_dense_fused<8, false>));
    raise(reinterpret_cast<const void*>(&k_dense_fused<12, false>));
    raise(reinterpret_cast<const void*>(&k_dense_fused<16, false>));
    raise(reinterpret_cast<const void*>(&k_dense_fused<8, true>));
    raise(reinterpret_cast<const void*>(&k_dense_fused<12, true>));
    raise(reinterpret_cast<const void*>(&k_dense_fused<16, true>));
    (void)hipGetLastError();
}

bool launch_dense_fused(hipStream_t s, int n_categories, const DenseArgs& a, HostStatus* hst, uint32_t seq) {
    if (a.n_rows <= 0) return true;
    const bool dmx = a.s[0].dmx != nullptr;
    if (dmx != (a.s[1].dmx != nullptr)) return false;
    if (!dense_fused_applies(n_categories, a.s[0].row_len, a.s[1].row_len)) return false;
    if (!a.scr_key || !a.scr_val || !a.ticket || a.scr_segs < 1 || a.scr_grid < 1) return false;
    const unsigned grid = (unsigned)std::min<int64_t>(a.n_rows, a.scr_grid);  // (a workgroup per scratch region; rows blockIdx, + grid, ...)
    if (n_categories <= 8) {
        if (dmx) k_dense_fused<8, true><<<grid, kNT, kDynLds, s>>>(a); else k_dense_fused<8, false><<<grid, kNT, kDynLds, s>>>(a);
    } else if (n_categories <= 12) {
        if (dmx) k_dense_fused<12, true><<<grid, kNT, kDynLds, s>>>(a); else k_dense_fused<12, false><<<grid, kNT, kDynLds, s>>>(a);
    } else {
        if (dmx) k_dense_fused<16, true><<<grid, kNT, kDynLds, s>>>(a); else k_dense_fused<16, false><<<grid, kNT, kDynLds, s>>>(a);
    }
    k_dense_publish<<<1, 1, 0, s>>>(a.st, hst, seq, a.ticket);
    return true;
}

}  // namespace lchd

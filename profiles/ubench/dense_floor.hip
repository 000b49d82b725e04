// dense_floor.hip -- micro-benchmark: what does the CHEAPEST in-LDS counting sort + sweep of a dense row pair cost on gfx950?
//
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 profiles/ubench/dense_floor.hip -o dense_floor && ./dense_floor
//   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d out -- ./dense_floor   (instruction counts)
//
// VERDICT r05 asked: if k_dense_fused does not reach 0.33 of the byte model's roofline (<= 2.08 ms per 10^4-atom from_coords call = 10^4 row
// pairs of 2 x 10^4 events), commit the floor of "bucket-sorting 2 x 10^4 f64 keys + one sweep per row in LDS", so that the bound is
// evidence.  This file is that floor.  It keeps every step the product cannot do without and drops everything else:
//
//   kept (the same arithmetic as lchd_dense_fused.hip)          dropped
//   one f64 key per event, 20 events per thread in registers    global loads (keys come from an integer hash: uniform on [0, 1)),
//   counting sort: one returning LDS atomic per event,            so the bucket is (int)(key * buckets): no sampled CDF, no plan,
//     scan, 4-byte surrogate to its bucket position,              no interpolation; no distance segments, no scratch memory;
//     rank among the first six surrogates of the bucket           no ties (distinct surrogates by construction of the hash: checked),
//   label (side | category) to its rank                           no error flags, no ragged rows, no weight-function dictionary,
//   F = 1 - exp(-b sqrt(key)) in the key's register               no ticket counter (rows are strided over the workgroups)
//   label-only sweep: packed chunk counts + DPP scans, the O(1)
//     Bhattacharyya update, H = sqrt(1 - D / sqrt(nA nB)),
//     one f64 weight H_k-1 - H_k per rank to LDS
//   every thread picks up the weights of its events (summation by parts: S = F(inf) H_last + sum_k F_k [H_k-1 - H_k])
//
// Two shapes, the same work per event:
//   NT = 512, 10 240 events per row pair, 66 KB of LDS: TWO workgroups per CU (what k_dense_fused runs, and what hides a phase's
//            latencies and barrier tails behind the other workgroup's arithmetic);
//   NT = 1024, 20 480 events per row pair, 135 KB: ONE workgroup per CU (profiles/r06/x_dense_one_design.hip.txt).
// Output: ms per launch, ns per event and CU, and -- with the PMC run -- vector instructions per event.  The 0.40 goal of BASELINE.json
// is 1.75 ms per 2 x 10^8 events on 256 CUs = 2.24 ns per event and CU; 0.33 is 2.66 ns.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

constexpr int kPart = 2048;   // sqrt(k), k < kPart, in LDS

__device__ __forceinline__ uint32_t scan_u32(uint32_t x) {  // inclusive prefix sum over the 64 lanes (DPP)
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return (uint32_t)v;
}
__device__ __forceinline__ uint64_t scan_fields(uint64_t x) {
    return ((uint64_t)scan_u32((uint32_t)(x >> 32)) << 32) | scan_u32((uint32_t)x);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double shr1_f64(double v) {  // value of lane - 1
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t spread4(uint64_t x) {  // four 4-bit fields -> four 16-bit fields
    const uint32_t v = (uint32_t)x;
    const uint32_t lo = (v & 0xFu) | ((v & 0xF0u) << 12);
    const uint32_t hi = ((v >> 8) & 0xFu) | ((v & 0xF000u) << 4);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double f_sqrt(double x) {  // v_rsq_f64 + Goldschmidt, 1 ulp
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    g = fma(d, h, g);
    return fmax(g, 0.0);
}
__device__ __forceinline__ double f_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x, y * y, 1.5);
    y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}
__device__ __forceinline__ double f_exp_nonpos(double x, const double* tab) {  // 64-entry table of 2^(j/64) + degree-5 polynomial
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
    return ldexp(fma(t, q, t), n >> 6);
}
__device__ __forceinline__ uint32_t mix(uint32_t x) {  // a bijection of the 32-bit integers (distinct inputs -> distinct surrogates)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// -DFLOOR_STAMPS: s_memtime deltas of wavefront 0 per phase (barrier waits included), summed over the launch
#ifdef FLOOR_STAMPS
__device__ unsigned long long g_floor_stamps[16];
#define FSTAMP(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); fst_acc[i] += t_ - fst_last; fst_last = t_; } } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

// NT threads, 2 * kQ * NT events per row pair, NB buckets; rows r = blockIdx, + gridDim, ...
// WPE: wavefronts per SIMD the register budget is cut for (4: 128 registers, 2: 256); kQ: events per thread and side
template <int NT, int NB, int WPE, int kQ>
__global__ __launch_bounds__(NT, WPE) void k_floor(int n_rows, const double* __restrict__ sqrt_tab, const double* __restrict__ exp2_tab, double* out) {
    constexpr int kWaves = NT / 64, kCap = 2 * kQ * NT, NW = 3;  // 12 category slots
    constexpr int kEpl = 13;                                    // events per lane and sweep round (odd: no LDS bank folding)
    constexpr int kR0 = kEpl * NT;                              // ranks of the first round
    static_assert((NB / 2) % (4 * NT) == 0, "whole 16-byte groups of histogram words per thread");
    constexpr size_t kHistOff = (size_t)(kCap + 8) * 4, kPartOff = (size_t)kR0 * 8;
    constexpr size_t kA = kHistOff + (size_t)NB * 2 + 16, kB = kPartOff + (size_t)kPart * 8 + 16;
    constexpr size_t kLabOff = kA > kB ? kA : kB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sur = reinterpret_cast<uint32_t*>(smem);               // sort: surrogates in bucket order
    double* Ws = reinterpret_cast<double*>(smem);                    // sweep: the round's weights (the same bytes)
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + kHistOff);   // sort: two 16-bit counters per word
    double* t_part = reinterpret_cast<double*>(smem + kPartOff);     // sweep: sqrt(k)
    uint8_t* lab = smem + kLabOff;                                   // side << 7 | category, in rank order
    __shared__ uint32_t wsum[kWaves], wtot_a[kWaves];
    __shared__ uint64_t wtot[kWaves][2 * NW], base_cnt[2][2 * NW];
    __shared__ double st_h[kWaves], carry_h[2], red_s[kWaves], exp_tab[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) exp_tab[tid] = exp2_tab[tid];
    if (tid == 64) Ws[kR0 + kPart] = 0.0;  // the weight of "not a rank of this round"
#ifdef FLOOR_STAMPS
    __shared__ unsigned long long fst_acc[16];
    if (tid < 16) fst_acc[tid] = 0ull;
    __syncthreads();
    unsigned long long fst_last = __builtin_amdgcn_s_memtime();
#endif
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        // keys: an integer hash per (row, event) -- the high 32 bits of the f64 key ARE the surrogate, distinct by construction
        double kk[2][kQ];
        uint32_t bs[2][kQ];
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t e = (uint32_t)((side * kQ + q) * NT + tid);
                const uint32_t h = mix(e * 0x9E3779B9u + (uint32_t)r * 0x85EBCA6Bu + 1u);
                kk[side][q] = (double)h * 0x1p-32 + (double)(e & 1023u) * 0x1p-44;  // in [0, 1), (uint32_t)(key * 2^32) == h
            }
        for (int w = tid; w < NB / 2 + 4; w += NT) hist[w] = 0u;
        if (tid < 2 * NW) base_cnt[0][tid] = 0ull;
        if (tid == 0) carry_h[0] = carry_h[1] = 0.0;
        __syncthreads();
        FSTAMP(0);   // keys (hash), histogram cleared
        // counting sort: bucket, slot from the returning atomic
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t b = (uint32_t)(kk[side][q] * (double)NB);
                const uint32_t sh = (b & 1u) * 16u;
                const uint32_t old = atomicAdd(&hist[b >> 1], 1u << sh);
                bs[side][q] = b | (((old >> sh) & 0xFFFFu) << 15);
            }
        __syncthreads();
        FSTAMP(1);   // bucket + returning atomic
        {
            constexpr int kScanW = NB / 2 / NT;
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
            const uint32_t incl = scan_u32(T);
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
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
            if (tid == 0) hist[NB / 2] = (uint32_t)kCap;
        }
        __syncthreads();
        FSTAMP(2);   // scan of the bucket counters
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t b = bs[side][q] & 32767u;
                const uint32_t pos = ((hist[b >> 1] >> ((b & 1u) * 16u)) & 0xFFFFu) + (bs[side][q] >> 15);
                sur[pos] = (uint32_t)(kk[side][q] * 4294967296.0);
                bs[side][q] = b;
            }
        if (tid < 8) sur[kCap + tid] = ~0u;
        __syncthreads();
        FSTAMP(3);   // surrogates to their bucket positions
        // rank among the bucket's surrogates (six read blind, the rest in a loop), label to the rank
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t b = bs[side][q];
                const uint32_t w0 = hist[b >> 1], w1 = hist[(b >> 1) + 1u];
                const uint32_t lo = (b & 1u) ? w0 >> 16 : w0 & 0xFFFFu, hi = (b & 1u) ? w1 & 0xFFFFu : w0 >> 16;
                const uint32_t mine = (uint32_t)(kk[side][q] * 4294967296.0);
                uint32_t less = 0;
#pragma unroll
                for (int m = 0; m < 6; ++m) less += sur[lo + m] < mine ? 1u : 0u;
                for (uint32_t j = lo + 6u; j < hi; ++j) less += sur[j] < mine ? 1u : 0u;
                bs[side][q] = lo + less;
            }
        __syncthreads();  // (labels below go to their own array, but the weights will overwrite the surrogates)
        FSTAMP(4);   // ranking
        uint32_t rk2[kQ];
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            lab[bs[0][q]] = (uint8_t)(mix((uint32_t)(q * NT + tid) + 77u) % 10u);
            lab[bs[1][q]] = (uint8_t)((mix((uint32_t)(q * NT + tid) + 991u) % 10u) | 128u);
            rk2[q] = bs[0][q] | (bs[1][q] << 16);
        }
        for (int k = tid; k < kPart; k += NT) t_part[k] = sqrt_tab[k];
        // F(distance) in the key's register: sqrt, one exponential (hyper_exp[1, 0.1] on a 100 A image)
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int q = 0; q < kQ; ++q) kk[side][q] = 1.0 - f_exp_nonpos(-0.1 * f_sqrt(kk[side][q] * 1.0e4), exp_tab);
        __syncthreads();
        FSTAMP(5);   // labels to their ranks, sqrt table, F in registers
        // label-only sweep, rounds of kR0 ranks
        double acc = 0.0;
        int base_na = 0, base_nb = 0, swept = 0;
#pragma unroll 1
        for (int rbase = 0; rbase < kCap; rbase += kR0) {
            const int n = min(kCap - rbase, kR0);
            const uint8_t* const sval = lab + rbase;
            const int epl = ((n + NT - 1) / NT) | 1;
            const int d0 = min(tid * epl, n), d1 = min(d0 + epl, n);
            uint64_t hA = 0ull, hT = 0ull;
            uint32_t n_al = 0;
            for (int e = 0; e < epl; ++e)
                if (d0 + e < d1) {
                    const uint32_t v = sval[d0 + e];
                    const uint64_t inc = 1ull << ((v & 15u) * 4u);
                    hT += inc;
                    hA += v < 128u ? inc : 0ull;
                    n_al += v < 128u ? 1u : 0u;
                }
            const uint64_t hB = hT - hA;
            uint64_t exA[NW], exB[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint64_t va_ = spread4(hA >> (16 * k)), vb_ = spread4(hB >> (16 * k));
                const uint64_t sa_ = scan_fields(va_), sb_ = scan_fields(vb_);
                exA[k] = sa_ - va_;
                exB[k] = sb_ - vb_;
                if (lane == 63) { wtot[wave][k] = sa_; wtot[wave][NW + k] = sb_; }
            }
            const uint32_t sna = scan_u32(n_al);
            if (lane == 63) wtot_a[wave] = sna;
            __syncthreads();
            FSTAMP(6);   // chunk counts: label histogram per lane, scans, barrier
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
            auto sqrt_cnt = [&](int cnt) -> double { if (cnt < kPart) return t_part[cnt]; else return f_sqrt((double)cnt); };
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
            double ra = f_rsqrt((double)totA), rb = f_rsqrt((double)totB);
            uint64_t dA = 0ull, dB = 0ull;
            double Hp = 0.0, firstH = 0.0;
            for (int e = 0; e < epl; ++e)
                if (d0 + e < d1) {
                    const uint32_t v = sval[d0 + e];
                    const int ct = (int)(v & 15u);
                    const bool takeA = v < 128u;
                    totA += takeA ? 1 : 0;
                    totB += takeA ? 0 : 1;
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
                    if (max(mine + 1, other) < kPart) { s0 = t_part[mine]; s1 = t_part[mine + 1]; so = t_part[other]; }
                    else { s0 = f_sqrt((double)mine); s1 = f_sqrt((double)(mine + 1)); so = f_sqrt((double)other); }
                    D = fma(s1 - s0, so, D);
                    const double rr = f_rsqrt((double)(takeA ? totA : totB));
                    ra = takeA ? rr : ra;
                    rb = takeA ? rb : rr;
                    const bool both = (totA > 0) & (totB > 0);
                    const double h2 = fmax(fma(-(ra * rb), D, 1.0), 0.0);
                    const double H = both ? f_sqrt(h2) : 0.0;
                    if (e == 0) firstH = H; else Ws[d0 + e] = Hp - H;
                    Hp = H;
                }
            FSTAMP(7);   // event loop
            const int last = (n - 1) / epl;
            if (lane == 63) st_h[wave] = Hp;
            if (tid == last) carry_h[(swept + 1) & 1] = Hp;
            const double prevH = shr1_f64(Hp);
            if (lane != 0 && d0 < d1) Ws[d0] = prevH - firstH;
            __syncthreads();
            if (lane == 0 && d0 < d1) Ws[d0] = (wave == 0 ? carry_h[swept & 1] : st_h[wave - 1]) - firstH;
            __syncthreads();
            FSTAMP(8);   // stitch: weights of the chunks' first ranks (two barriers)
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const uint32_t ia = (rk2[q] & 0xFFFFu) - (uint32_t)rbase, ib = (rk2[q] >> 16) - (uint32_t)rbase;
                acc = fma(kk[0][q], Ws[ia < (uint32_t)n ? ia : (uint32_t)(kR0 + kPart)], acc);
                acc = fma(kk[1][q], Ws[ib < (uint32_t)n ? ib : (uint32_t)(kR0 + kPart)], acc);
            }
            FSTAMP(10);  // every thread picks up the weights of its events
            base_na += seg_a;
            base_nb += n - seg_a;
            ++swept;
        }
        for (int k = 32; k > 0; k >>= 1) {
            const int lo = __shfl_xor(__double2loint(acc), k), hi = __shfl_xor(__double2hiint(acc), k);
            acc += __hiloint2double(hi, lo);
        }
        if (lane == 0) red_s[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int w = 0; w < kWaves; ++w) sum += red_s[w];
            out[r] = sum + carry_h[swept & 1];  // F(inf) = 1
        }
        __syncthreads();
        FSTAMP(9);   // sum, write
    }
#ifdef FLOOR_STAMPS
    __syncthreads();
    if (tid < 16) atomicAdd(&g_floor_stamps[tid], fst_acc[tid]);
#endif
}

template <int NT, int NB, int WPE, int kQ>
void run(const char* name, int n_rows, int grid, const double* d_sqrt, const double* d_exp, double* d_out, int reps) {
    constexpr int kCap = 2 * kQ * NT, kR0 = 13 * NT;
    constexpr size_t kHistOff = (size_t)(kCap + 8) * 4, kPartOff = (size_t)kR0 * 8;
    constexpr size_t kA = kHistOff + (size_t)NB * 2 + 16, kB = kPartOff + (size_t)kPart * 8 + 16;
    const size_t lds = (kA > kB ? kA : kB) + (size_t)kCap + 16;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_floor<NT, NB, WPE, kQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) k_floor<NT, NB, WPE, kQ><<<grid, NT, lds>>>(n_rows, d_sqrt, d_exp, d_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_floor<NT, NB, WPE, kQ><<<grid, NT, lds>>>(n_rows, d_sqrt, d_exp, d_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<double> h(n_rows);
    CHECK(hipMemcpy(h.data(), d_out, sizeof(double) * n_rows, hipMemcpyDeviceToHost));
    double cs = 0.0;
    bool sane = true;
    for (double v : h) { cs += v; sane = sane && v > 0.0 && v < 1.0; }
#ifdef FLOOR_STAMPS
    {
        unsigned long long st[16], z[16] = {0};
        CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_floor_stamps), sizeof st));
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_floor_stamps), z, sizeof z));
        static const char* names[11] = {"keys + clear", "bucket atomics", "scan", "scatter", "ranking", "labels + F", "chunk counts", "event loop", "stitch", "sum", "pick-up"};
        double tot = 0.0;
        for (int i = 0; i < 11; ++i) tot += (double)st[i];
        printf("# %s: share of wavefront 0's s_memtime ticks per phase:", name);
        for (int i = 0; i < 11; ++i) printf(" %s %.1f %%;", names[i], 100.0 * (double)st[i] / tot);
        printf("\n");
    }
#endif
    const double events = (double)n_rows * kCap;
    printf("{\"shape\": \"%s\", \"threads\": %d, \"waves_per_simd_budget\": %d, \"events_per_row_pair\": %d, \"buckets\": %d, \"lds_bytes\": %zu, \"workgroups\": %d, \"row_pairs\": %d, "
           "\"ms_per_launch\": %.4f, \"events_per_s\": %.4g, \"ns_per_event_and_cu\": %.3f, \"ms_per_2e8_events\": %.3f, \"checksum\": %.12g, \"scores_in_0_1\": %s}\n",
           name, NT, WPE, kCap, NB, lds, grid, n_rows, ms, events / (ms * 1e-3), ms * 1e6 * 256.0 / events, ms * 2.0e8 / events, cs, sane ? "true" : "false");
}

}  // namespace

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    std::vector<double> hs(kPart), he(64);
    for (int k = 0; k < kPart; ++k) hs[k] = __builtin_sqrt((double)k);
    for (int k = 0; k < 64; ++k) he[k] = __builtin_exp2((double)k / 64.0);
    double *d_sqrt, *d_exp, *d_out;
    CHECK(hipMalloc(&d_sqrt, sizeof(double) * kPart));
    CHECK(hipMalloc(&d_exp, sizeof(double) * 64));
    CHECK(hipMalloc(&d_out, sizeof(double) * 20000));
    CHECK(hipMemcpy(d_sqrt, hs.data(), sizeof(double) * kPart, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_exp, he.data(), sizeof(double) * 64, hipMemcpyHostToDevice));
    // the same 2.048 x 10^8 events either way
    run<512, 8192, 4, 10>("two workgroups of 512 threads per CU (128 registers), 10 240 events per row pair", 20000, 512, d_sqrt, d_exp, d_out, reps);
    run<512, 8192, 2, 10>("one workgroup of 512 threads per CU (256 registers: what the code wants unconstrained is 210), 10 240 events per row pair", 20000, 256, d_sqrt, d_exp, d_out, reps);
    run<1024, 16384, 4, 10>("one workgroup of 1024 threads per CU (128 registers), 20 480 events per row pair", 10000, 256, d_sqrt, d_exp, d_out, reps);
    // (tried and not kept: 512 threads with 40 events per thread for the 20 480-event row pair -- 119 spilled dwords even at 256
    //  registers, 3.07 ms per 2 x 10^8 events)
    return 0;
}

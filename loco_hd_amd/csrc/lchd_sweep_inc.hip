// lchd_sweep_inc.hip -- K2 for Kullback-Leibler and Renyi divergences in O(1) per event (the common configuration).
//
// The generic sweep (k_sweep<.., MODE_GEN, ..>, lchd_sweep.hip) evaluates these distances from scratch at every breakpoint:
// one logarithm (KL) or one logarithm and one exponential (Renyi) per CATEGORY and event -- 12 and 20 ms per 10^6 C2a pairs
// against 1.5 ms for the default Hellinger distance, which is updated in O(1).  With unit category weights the PMFs are
// ratios of small integers (p_c = a_c / N_a, counts <= 512 here), and for a tiny smoothing constant eps the divergences
// separate into sums over the categories that an event changes in ONE term (reference:
// /root/reference/src/locohd/pmf/statistical_distances.rs:23-29 and :31-78):
//
//   ln(p_c + eps) = ln(a_c + eps N_a) - ln N_a = ln a_c + eps N_a / a_c - ln N_a + O((eps N_a / a_c)^2)        (a_c >= 1)
//   ln(q_c + eps) = ln eps                                                                                     (b_c == 0, exactly)
//
//   KL = sum_c p_c ln((p_c + eps) / (q_c + eps))
//      = [S_aa + eps N_a K_a - S_ab - eps N_b R_ab - A_0 ln eps] / N_a - ln N_a + (1 - A_0 / N_a) ln N_b
//        S_aa = sum a ln a,  K_a = #{a > 0},  S_ab = sum_{b>0} a ln b,  R_ab = sum_{b>0} a / b,  A_0 = sum_{b=0} a
//
//   Renyi_alpha = ln(R) / (alpha - 1),  R = sum_c p_c ((p_c + eps) / (q_c + eps))^(alpha - 1),   R N_a^alpha =
//        N_b^(alpha-1) [ S_1 + (alpha-1) eps (N_a S_2 - N_b S_3) ] + eps^(1-alpha) [ Z + (alpha-1) eps N_a Z_1 ]
//        S_1 = sum_{b>0} a^alpha b^(1-alpha),  S_2 = sum_{b>0} a^(alpha-1) b^(1-alpha),  S_3 = sum_{b>0} a^alpha b^(-alpha),
//        Z = sum_{b=0} a^alpha,  Z_1 = sum_{b=0} a^(alpha-1)
//
// The dropped terms are second order in eps N / count <= 512 eps: the host takes this path only for eps <= 1e-9 (error
// <= 3e-13, the tests hold 1e-11 against the CPU restatement's libm), unit weights, CDF-keyed environments of at most 512
// points (the tables k ln k, ln k, 1/k or k^alpha ... for k <= 512 are built in LDS by every workgroup with the library
// routines).  Sums are rebuilt from the exact integer counts at every lane chunk (<= 6 events), and -- Renyi -- whenever an
// event moves a category out of the b = 0 class (Z loses its possibly dominant term: a subtraction that must not leave
// rounding debris).  Everything else (tiles, merge path, packed 16-bit count scan, stitching) is the scheme of k_sweep.
#include "lchd_kcommon.h"

namespace lchd {

enum { INC_KL = 1, INC_RENYI = 2 };
constexpr int kIncTab = 520;              // counts 0 .. 512 (+ padding)
constexpr int kIncEPL = 6, kIncTile = 64 * kIncEPL, kIncWaves = 4;

__device__ __forceinline__ int inc_merge_path(const uint64_t* A, int nA, const uint64_t* B, int nB, int d) {
    int lo = max(0, d - nB), hi = min(d, nA);
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (A[mid] <= B[d - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint64_t inc_spread4(uint64_t x) {  // four 4-bit fields -> four 16-bit fields
    const uint32_t v = (uint32_t)x;
    const uint32_t lo = (v & 0xFu) | ((v & 0xF0u) << 12);
    const uint32_t hi = ((v >> 8) & 0xFu) | ((v & 0xF000u) << 4);
    return ((uint64_t)hi << 32) | lo;
}
static __device__ __noinline__ double inc_log(double x) { return log(x); }
static __device__ __noinline__ double inc_pow(double x, double y) { return pow(x, y); }

template <int CMAX, int KIND>
__global__ __launch_bounds__(64 * kIncWaves, 2) void k_sweep_inc(SweepArgs args) {
    constexpr int TILE = kIncTile, EPL = kIncEPL, WPB = kIncWaves;
    constexpr int NW = (CMAX + 3) / 4;    // u64 words of four 16-bit count fields per side
    constexpr int NH = (CMAX + 15) / 16;  // u64 words of sixteen 4-bit histogram fields per side
    constexpr int NTAB = KIND == INC_KL ? 3 : 5;
    __shared__ double tab[NTAB][kIncTab];
    __shared__ uint64_t sA_[WPB][TILE], sB_[WPB][TILE];
    __shared__ uint8_t cA_[WPB][TILE], cB_[WPB][TILE];
    __shared__ uint64_t lc_[WPB][2 * NW * 64];  // per-lane counts: [side][word][lane], four 16-bit fields per word
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const DevConfig* __restrict__ cfgp = args.cfg;
    const double prm0 = cfgp->sd_p0, prm1 = cfgp->sd_p1;
    const double eps = KIND == INC_KL ? (cfgp->sd_kind == SD_KL ? prm0 : prm1) : prm1;  // (Renyi with alpha = 1 is the KL form, :36-38)
    const double alpha = prm0, am1 = alpha - 1.0;
    // tables: KL  0: ln k (0 for k = 0)   1: k ln k   2: 1 / k (0 for k = 0)
    //         Renyi 0: ln k   1: k^alpha   2: k^(1-alpha)   3: k^(alpha-1)   4: k^(-alpha)     (all 0 for k = 0)
    for (int k = tid; k < kIncTab; k += 64 * WPB) {
        const double x = (double)k;
        const double lk = k ? inc_log(x) : 0.0;
        tab[0][k] = lk;
        if constexpr (KIND == INC_KL) {
            tab[1][k] = x * lk;
            tab[2][k] = k ? 1.0 / x : 0.0;
        } else {
            tab[1][k] = k ? inc_pow(x, alpha) : 0.0;
            tab[2][k] = k ? inc_pow(x, 1.0 - alpha) : 0.0;
            tab[3][k] = k ? inc_pow(x, alpha - 1.0) : 0.0;
            tab[4][k] = k ? inc_pow(x, -alpha) : 0.0;
        }
    }
    const double lneps = inc_log(eps);
    const double epow = KIND == INC_RENYI ? inc_pow(eps, 1.0 - alpha) : 0.0;
    __syncthreads();
    uint64_t* sA = sA_[wv];
    uint64_t* sB = sB_[wv];
    uint8_t* cA = cA_[wv];
    uint8_t* cB = cB_[wv];
    unsigned char* lcl = reinterpret_cast<unsigned char*>(lc_[wv]) + lane * 8;
    constexpr int kSide = NW * 512;

    const double Finf0 = cfgp->wf_finf[0];
    const int64_t pstride = (int64_t)gridDim.x * WPB, total = args.n_pairs;
    for (int64_t p = (int64_t)blockIdx.x * WPB + wv; p < total; p += pstride) {
        const int4 m = args.meta[p];
        const int mx = __builtin_amdgcn_readfirstlane(m.x), my = __builtin_amdgcn_readfirstlane(m.y);
        const int mz = __builtin_amdgcn_readfirstlane(m.z), mw = __builtin_amdgcn_readfirstlane(m.w);
        const int nA = mz & 0xFFFFFF, nB = mw & 0xFFFFFF;
        if (nA <= 0 || nB <= 0) {  // anchor out of range / overflowed or empty environment: flagged where it happened
            if (lane == 0) args.out[p] = nan("");
            continue;
        }
        const int c0a = (mz >> 24) & 255, c0b = (mw >> 24) & 255;
        const uint64_t* __restrict__ kA = args.env_a.key + (int64_t)mx * args.env_a.stride;
        const uint64_t* __restrict__ kB = args.env_b.key + (int64_t)my * args.env_b.stride;
        const uint8_t* __restrict__ tA = args.env_a.cat + (int64_t)mx * args.env_a.stride;
        const uint8_t* __restrict__ tB = args.env_b.cat + (int64_t)my * args.env_b.stride;

        uint64_t cntA[NW], cntB[NW];  // wave-uniform packed counts, seeded with the two anchors (src/locohd.rs:82-84)
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            cntA[k] = ((c0a >> 2) == k) ? (1ull << ((c0a & 3) * 16)) : 0ull;
            cntB[k] = ((c0b >> 2) == k) ? (1ull << ((c0b & 3) * 16)) : 0ull;
        }
        // ---- per-lane state ----------------------------------------------------------------------------------------
        int Na = 1, Nb = 1;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;  // KL: S_aa, S_ab, R_ab;  Renyi: S_1, S_2, S_3, Z, Z_1
        int A0 = 0, Ka = 0;
        auto add_cat = [&](int a, int b) {  // one category's contribution to the sums
            if constexpr (KIND == INC_KL) {
                const double da = (double)a;
                s0 += tab[1][a];
                s1 += da * tab[0][b];
                s2 += da * tab[2][b];
                A0 += b == 0 ? a : 0;
                Ka += a > 0 ? 1 : 0;
            } else {
                const double P = tab[1][a], P1 = tab[3][a], Q = tab[2][b], Q1 = tab[4][b];
                s0 += P * Q;
                s1 += P1 * Q;
                s2 += P * Q1;
                s3 += b == 0 ? P : 0.0;
                s4 += b == 0 ? P1 : 0.0;
            }
        };
        auto reset_sums = [&]() { s0 = s1 = s2 = s3 = s4 = 0.0; A0 = 0; Ka = 0; };
        auto load_from_words = [&](const uint64_t (&ea)[NW], const uint64_t (&eb)[NW]) {
            reset_sums();
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                add_cat((int)((ea[c >> 2] >> ((c & 3) * 16)) & 0xFFFFull), (int)((eb[c >> 2] >> ((c & 3) * 16)) & 0xFFFFull));
        };
        auto load_from_lds = [&]() {  // rebuild the sums from this lane's LDS counts (rare: see the header)
            reset_sums();
#pragma unroll 1
            for (int k = 0; k < NW; ++k) {
                const uint64_t wa = *reinterpret_cast<const uint64_t*>(lcl + k * 512), wb = *reinterpret_cast<const uint64_t*>(lcl + kSide + k * 512);
#pragma unroll
                for (int f = 0; f < 4; ++f) add_cat((int)((wa >> (f * 16)) & 0xFFFFull), (int)((wb >> (f * 16)) & 0xFFFFull));
            }
        };
        auto distance = [&]() -> double {
            const double dNa = (double)Na, dNb = (double)Nb;
            if constexpr (KIND == INC_KL) {
                const double num = ((s0 + eps * dNa * (double)Ka) - (s1 + eps * dNb * s2)) - (double)A0 * lneps;
                const double ia = tab[2][Na];
                return (num * ia - tab[0][Na]) + ((double)(Na - A0) * ia) * tab[0][Nb];
            } else {
                const double r = tab[3][Nb] * (s0 + am1 * eps * (dNa * s1 - dNb * s2)) + epow * (s3 + am1 * eps * dNa * s4);
                return (log_fast(r) - alpha * tab[0][Na]) / am1;
            }
        };

        uint64_t exA[NW], exB[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) { exA[k] = cntA[k]; exB[k] = cntB[k]; }
        load_from_words(exA, exB);
        double F_carry = u2d(kA[0]);  // F(0): both anchors sit at distance 0
        double H_carry = distance();
        double acc = 0.0;

        const int mA = nA - 1, mB = nB - 1, M = mA + mB;
        int ia = 0, ib = 0;
        for (int k0 = 0; k0 < M; k0 += TILE) {
            const int T = min(TILE, M - k0);
            const int nAt = min(TILE, mA - ia), nBt = min(TILE, mB - ib);
            wave_sync_lds();  // previous tile fully consumed
            {
                uint64_t rkA[EPL], rkB[EPL];
                uint8_t rcA[EPL], rcB[EPL];
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    const int t = lane + 64 * u;
                    rkA[u] = t < nAt ? kA[1 + ia + t] : 0ull;
                    rcA[u] = t < nAt ? tA[1 + ia + t] : (uint8_t)0;
                    rkB[u] = t < nBt ? kB[1 + ib + t] : 0ull;
                    rcB[u] = t < nBt ? tB[1 + ib + t] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    const int t = lane + 64 * u;
                    if (t < nAt) { sA[t] = rkA[u]; cA[t] = rcA[u]; }
                    if (t < nBt) { sB[t] = rkB[u]; cB[t] = rcB[u]; }
                }
            }
            wave_sync_lds();
            const int epl = (T + 63) >> 6;
            const int d0 = min(lane * epl, T), d1 = min(d0 + epl, T);
            const int i1 = inc_merge_path(sA, nAt, sB, nBt, d1);
            int i0 = __shfl_up(i1, 1);
            if (lane == 0) i0 = 0;
            const int iend = __builtin_amdgcn_readlane(i1, 63);
            const int j0 = d0 - i0, j1 = d1 - i1;
            // pass 1: 4-bit-per-category histogram of this lane's chunk, widened to 16-bit fields and scanned across the wavefront
            uint64_t hA[NH], hB[NH];
#pragma unroll
            for (int k = 0; k < NH; ++k) hA[k] = hB[k] = 0;
            for (int i = i0; i < i1; ++i) {
                const int ct = cA[i];
#pragma unroll
                for (int k = 0; k < NH; ++k) hA[k] += ((ct >> 4) == k) ? (1ull << ((ct & 15) * 4)) : 0ull;
            }
            for (int j = j0; j < j1; ++j) {
                const int ct = cB[j];
#pragma unroll
                for (int k = 0; k < NH; ++k) hB[k] += ((ct >> 4) == k) ? (1ull << ((ct & 15) * 4)) : 0ull;
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint64_t va_ = inc_spread4(hA[(k * 4) / 16] >> (((k * 4) % 16) * 4)), vb_ = inc_spread4(hB[(k * 4) / 16] >> (((k * 4) % 16) * 4));
                const uint64_t sa_ = wave_incl_scan_fields(va_), sb_ = wave_incl_scan_fields(vb_);
                exA[k] = cntA[k] + sa_ - va_;
                exB[k] = cntB[k] + sb_ - vb_;
                cntA[k] += readlane_u64(sa_, 63);
                cntB[k] += readlane_u64(sb_, 63);
                *reinterpret_cast<uint64_t*>(lcl + k * 512) = exA[k];
                *reinterpret_cast<uint64_t*>(lcl + kSide + k * 512) = exB[k];
            }
            Na = 1 + ia + i0;
            Nb = 1 + ib + j0;
            load_from_words(exA, exB);

            // pass 2: this lane's events, one after the other
            int i = i0, j = j0;
            uint64_t ka = sA[i], kb = sB[j];
            double Fp = 0.0, Hp = 0.0, firstF = 0.0, local = 0.0;
            for (int e = 0; e < epl; ++e) {
                if (d0 + e < d1) {
                    const bool takeA = (i < i1) & ((j >= j1) | (ka <= kb));  // A-first on ties; an exhausted run cannot be taken
                    const uint64_t key = takeA ? ka : kb;
                    const int ct = (takeA ? cA : cB)[takeA ? i : j];
                    i += takeA ? 1 : 0;
                    j += takeA ? 0 : 1;
                    ka = sA[i];
                    kb = sB[j];
                    const double F = u2d(key);
                    if (e == 0) firstF = F; else local += (F - Fp) * Hp;
                    // pmf.rs:47-63: one more point of category ct on one side
                    unsigned char* pf = lcl + ((ct >> 2) << 9) + ((ct & 3) << 1);
                    const int a = *reinterpret_cast<const uint16_t*>(pf), b = *reinterpret_cast<const uint16_t*>(pf + kSide);
                    *reinterpret_cast<uint16_t*>(pf + (takeA ? 0 : kSide)) = (uint16_t)((takeA ? a : b) + 1);
                    Na += takeA ? 1 : 0;
                    Nb += takeA ? 0 : 1;
                    if constexpr (KIND == INC_KL) {
                        if (takeA) {
                            s0 += tab[1][a + 1] - tab[1][a];
                            s1 += tab[0][b];
                            s2 += tab[2][b];
                            A0 += b == 0 ? 1 : 0;
                            Ka += a == 0 ? 1 : 0;
                        } else {
                            const double da = (double)a;
                            s1 += da * (tab[0][b + 1] - tab[0][b]);
                            s2 += da * (tab[2][b + 1] - tab[2][b]);
                            A0 -= b == 0 ? a : 0;
                        }
                    } else {
                        if (takeA) {
                            const double dP = tab[1][a + 1] - tab[1][a], dP1 = tab[3][a + 1] - tab[3][a];
                            const double Q = tab[2][b], Q1 = tab[4][b];
                            s0 += dP * Q;
                            s1 += dP1 * Q;
                            s2 += dP * Q1;
                            s3 += b == 0 ? dP : 0.0;
                            s4 += b == 0 ? dP1 : 0.0;
                        } else if (b == 0 && a > 0) {
                            load_from_lds();  // the category leaves the b = 0 class: Z loses a (possibly dominant) term -- rebuilt exactly
                        } else {
                            const double P = tab[1][a], P1 = tab[3][a];
                            const double dQ = tab[2][b + 1] - tab[2][b], dQ1 = tab[4][b + 1] - tab[4][b];
                            s0 += P * dQ;
                            s1 += P1 * dQ;
                            s2 += P * dQ1;
                        }
                    }
                    Hp = distance();
                    Fp = F;
                }
            }
            // stitch lane chunks: (F_first - F_last_of_previous_lane) * H_before_my_first_event
            double prevF = wave_shr1_f64(Fp), prevH = wave_shr1_f64(Hp);
            if (lane == 0) { prevF = F_carry; prevH = H_carry; }
            if (d0 < d1) local += (firstF - prevF) * prevH;
            acc += local;
            const int last = (T - 1) / epl;
            F_carry = readlane_f64(Fp, last);
            H_carry = readlane_f64(Hp, last);
            ia += iend;
            ib += T - iend;
        }
        acc = wave_sum_f64(acc);
        acc += (Finf0 - F_carry) * H_carry;  // the last interval to +inf (:165-171,204-210,212-221)
        if (lane == 0) args.out[p] = acc;
    }
}

template <int KIND>
static void launch_inc_kind(hipStream_t s, int cmax, unsigned grid, const SweepArgs& a) {
    constexpr int NTH = 64 * kIncWaves;
    if (cmax <= 8) k_sweep_inc<8, KIND><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 12) k_sweep_inc<12, KIND><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 16) k_sweep_inc<16, KIND><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 24) k_sweep_inc<24, KIND><<<grid, NTH, 0, s>>>(a);
    else k_sweep_inc<32, KIND><<<grid, NTH, 0, s>>>(a);
}
void launch_sweep_inc(hipStream_t s, int kind, int cmax, const SweepArgs& a) {
    const int64_t blocks = (a.n_pairs + kIncWaves - 1) / kIncWaves;
    const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);  // (grid-stride: the tables are built once per workgroup)
    if (kind == INC_KL) launch_inc_kind<INC_KL>(s, cmax, grid, a);
    else launch_inc_kind<INC_RENYI>(s, cmax, grid, a);
}

}  // namespace lchd

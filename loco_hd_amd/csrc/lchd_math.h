// lchd_math.h -- the numeric leaves of the LoCoHD path, written once for host and device.
//
//   CDFs                    /root/reference/src/locohd/weight_function/cdfs.rs:5-63
//   statistical distances   /root/reference/src/locohd/pmf/statistical_distances.rs:4-78
//
// Everything is f64, like the reference.  Device code gets these through the kernels' translation
// unit (ocml exp/pow/log/sqrt), host code (lchd_wf_cdf, lchd_sd_run) through libm.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LCHD_HD __host__ __device__ __forceinline__
#else
#define LCHD_HD inline
#endif

// Loop over categories: with NMAX > 0 the loop is fully unrolled to NMAX iterations and guarded by c < n
// (register-resident state on the device); with NMAX == 0 it is a plain runtime loop (host).
// (unroll count 1 = "do not unroll": a bare `unroll` on the runtime loop draws -Wpass-failed from every instantiation with NMAX == 0)
#define LCHD_PRAGMA_(x) _Pragma(#x)
#define LCHD_FOR_C(NMAX, n, c) \
    LCHD_PRAGMA_(clang loop unroll_count((NMAX) > 0 ? (NMAX) : 1)) for (int c = 0; c < ((NMAX) > 0 ? (NMAX) : (n)); ++c) if ((NMAX) == 0 || c < (n))

namespace lchd {

enum { WF_HYPER_EXP = 0, WF_DAGUM = 1, WF_UNIFORM = 2, WF_KUMARASWAMY = 3 };
enum { SD_HELLINGER = 0, SD_KS = 1, SD_KL = 2, SD_RENYI = 3 };

// cdfs.rs:5-21: 1 - sum_i a_i exp(-b_i x) / sum_i a_i, params = (a_1..a_n, b_1..b_n)
LCHD_HD double cdf_hyper_exp(const double* p, int np, double x) {
    double norm = 0.0, sum = 0.0;
    const int n = np / 2;
    for (int i = 0; i < n; ++i) {
        sum += p[i] * exp(-p[n + i] * x);
        norm += p[i];
    }
    return 1.0 - sum / norm;
}
// cdfs.rs:27-29: (1 + (x/B)^-A)^-P, params = (A, B, P)
LCHD_HD double cdf_dagum(const double* p, double x) { return pow(1.0 + pow(x / p[1], -p[0]), -p[2]); }
// cdfs.rs:39-45
LCHD_HD double cdf_uniform(const double* p, double x) {
    if (x < p[0]) return 0.0;
    if (x > p[1]) return 1.0;
    return (x - p[0]) / (p[1] - p[0]);
}
// cdfs.rs:56-63
LCHD_HD double cdf_kumaraswamy(const double* p, double x) {
    if (x < p[0]) return 0.0;
    if (x > p[1]) return 1.0;
    const double z = (x - p[0]) / (p[1] - p[0]);
    return 1.0 - pow(1.0 - pow(z, p[2]), p[3]);
}
LCHD_HD double cdf_eval(int kind, const double* p, int np, double x) {
    switch (kind) {
        case WF_HYPER_EXP: return cdf_hyper_exp(p, np, x);
        case WF_DAGUM: return cdf_dagum(p, x);
        case WF_UNIFORM: return cdf_uniform(p, x);
        default: return cdf_kumaraswamy(p, x);
    }
}

// statistical_distances.rs:4-10.  P1/P2 are callables c -> probability so that kernels can feed
// register-resident state without materialising the normalised vectors (pmf.rs:78-81 does).
template <int NMAX, class P1, class P2>
LCHD_HD double sd_hellinger(P1 p1, P2 p2, int n, double e) {
    const double inv = 1.0 / e;
    double dist = 0.0;
    LCHD_FOR_C(NMAX, n, c) dist += pow(fabs(pow(p1(c), inv) - pow(p2(c), inv)), e);
    return pow(dist / 2.0, inv);
}
// exponent 2 (the default, src/locohd.rs:365-370): pow(x, .5) == sqrt(x) and pow(|d|, 2) == d*d
template <int NMAX, class P1, class P2>
LCHD_HD double sd_hellinger2(P1 p1, P2 p2, int n) {
    double dist = 0.0;
    LCHD_FOR_C(NMAX, n, c) {
        const double d = sqrt(p1(c)) - sqrt(p2(c));
        dist += d * d;
    }
    return sqrt(dist / 2.0);
}
// :12-21
template <int NMAX, class P1, class P2>
LCHD_HD double sd_ks(P1 p1, P2 p2, int n) {
    double best = 0.0;
    LCHD_FOR_C(NMAX, n, c) {
        const double d = fabs(p1(c) - p2(c));
        best = (c == 0 || d >= best) ? d : best;
    }
    return best;
}
// :23-29
template <int NMAX, class P1, class P2>
LCHD_HD double sd_kl(P1 p1, P2 p2, int n, double eps) {
    double dist = 0.0;
    LCHD_FOR_C(NMAX, n, c) {
        const double x = p1(c);
        dist += x * log((x + eps) / (p2(c) + eps));
    }
    return dist;
}
// :31-78
template <int NMAX, class P1, class P2>
LCHD_HD double sd_renyi(P1 p1, P2 p2, int n, double alpha, double eps) {
    if (alpha == 1.0) return sd_kl<NMAX>(p1, p2, n, eps);
    if (alpha == INFINITY) {
        double best = 0.0;
        LCHD_FOR_C(NMAX, n, c) {
            const double r = (p1(c) + eps) / (p2(c) + eps);
            best = (c == 0 || r >= best) ? r : best;
        }
        return log(best);
    }
    if (alpha == 0.0) {
        double s = 0.0;
        LCHD_FOR_C(NMAX, n, c) s += (p1(c) > 0.0) ? p2(c) : 0.0;
        return -log(s);
    }
    double s = 0.0;
    LCHD_FOR_C(NMAX, n, c) {
        const double x = p1(c);
        s += x * pow((x + eps) / (p2(c) + eps), alpha - 1.0);
    }
    return log(s) / (alpha - 1.0);
}
// StatisticalDistance::run dispatch, :123-142
template <int NMAX, class P1, class P2>
LCHD_HD double sd_eval(int kind, double prm0, double prm1, P1 p1, P2 p2, int n) {
    switch (kind) {
        case SD_HELLINGER: return (prm0 == 2.0) ? sd_hellinger2<NMAX>(p1, p2, n) : sd_hellinger<NMAX>(p1, p2, n, prm0);
        case SD_KS: return sd_ks<NMAX>(p1, p2, n);
        case SD_KL: return sd_kl<NMAX>(p1, p2, n, prm0);
        default: return sd_renyi<NMAX>(p1, p2, n, prm0, prm1);
    }
}

}  // namespace lchd

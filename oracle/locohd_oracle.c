/*
 * locohd_oracle.c -- CPU restatement of the LoCoHD scoring path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker*, not the product: only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it.  The product path (loco_hd_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * It restates, function by function and in the reference's own evaluation order, the Rust core of
 * fazekaszs/loco_hd (all citations are relative to /root/reference):
 *
 *   src/locohd.rs:61-226                      -> orc_stat_dist_integral   (merge sweep + 3 tails)
 *   src/locohd.rs:410-458                     -> orc_from_dmxs
 *   src/locohd.rs:463-476                     -> orc_from_coords
 *   src/locohd.rs:479-567                     -> orc_from_primitives     (env_from_idx :514-542)
 *   src/locohd/pmf.rs:20-88                   -> pmf_* helpers
 *   src/locohd/pmf/statistical_distances.rs   -> orc_sd_validate / orc_sd_run
 *   src/locohd/weight_function.rs:22-120      -> orc_wf_validate / orc_wf_integral_point / _range
 *   src/locohd/weight_function/cdfs.rs:5-63   -> cdf_hyper_exp / cdf_dagum / cdf_uniform / cdf_kumaraswamy
 *   src/locohd/utils.rs:1-39                  -> orc_euclidean_distance / distance matrix / orc_sort_together
 *   src/locohd/tag_pairing_rule.rs:49-75      -> orc_tag_pair_accepted
 *
 * Third-party arithmetic that is NOT under /root/reference: crate `kd-tree` (Cargo.toml:17-18,
 * requirement "^0.6.0", no Cargo.lock => unpinned).  Its published algorithm is restated in
 * kd_build / kd_within below: `build_by_ordered_float` = recursive median split cycling the axis,
 * `within_radius` = axis-aligned box walk [q-r, q+r] followed by the filter  sum(diff^2) < r^2
 * (strict, on the squared distance).  No reference test puts a point exactly on the threshold
 * (tests/test_tag_pairing_rule.py:100-157 uses 1.002 with neighbours at 1 and sqrt 2), so membership
 * at distance == threshold is PARITY UNPINNED.
 *
 * PARITY STATUS: the Rust core cannot be compiled or imported in the build container (no cargo /
 * rustc / maturin, no wheel) and the reference's regression outputs
 * (tests/test_data/test_output_collection_*.pickle) are absent from the tree
 * (.MISSING_LARGE_BLOBS).  The oracle is pinned against every known-answer the reference's own tests
 * hold for this path: tests/test_locohd.py:27-73, tests/test_tag_pairing_rule.py:8-157,
 * tests/test_wfs.py:8-156 (see tests/test_oracle_kat.py).
 *
 * Strings: the reference keys categories and tags by String.  Here a category is the integer the
 * reference's HashMap would return (src/locohd.rs:312-316), -1 meaning "not in the map"
 * (src/locohd/pmf.rs:38-42 error); a tag is an interned integer (equal strings <=> equal ints).
 *
 * Floating point: Rust's f64::powf / exp / ln lower to the platform libm, i.e. the same glibc
 * pow/exp/log this file calls.  x.powf(2.) and x.powf(0.5) with literal exponents are folded by LLVM
 * to x*x and sqrt(x) (LibCallSimplifier::optimizePow); gcc folds pow(x,2.) the same way and this
 * file writes sqrt() for the 0.5 case.  Build with -ffp-contract=off (Rust never contracts).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR 1   /* reference: PyValueError */
#define ORC_PANIC 2 /* reference: Rust panic (PanicException) */

static __thread char orc_errbuf[512];
const char *orc_last_error(void) { return orc_errbuf; }
#define FAIL(code, ...) do { snprintf(orc_errbuf, sizeof orc_errbuf, __VA_ARGS__); return (code); } while (0)

/* ------------------------------------------------------------------------------------------------
 * Weight functions.  kind: 0 hyper_exp, 1 dagum, 2 uniform, 3 kumaraswamy
 * ---------------------------------------------------------------------------------------------- */
enum { WF_HYPER_EXP = 0, WF_DAGUM = 1, WF_UNIFORM = 2, WF_KUMARASWAMY = 3 };

/* src/locohd/weight_function/cdfs.rs:5-21 */
static double cdf_hyper_exp(const double *p, int np, double x) {
    double norm = 0.0, sum = 0.0;
    int border = np / 2;
    for (int i = 0; i < border; ++i) {
        sum += p[i] * exp(-p[border + i] * x);
        norm += p[i];
    }
    return 1.0 - sum / norm;
}
/* cdfs.rs:27-29 */
static double cdf_dagum(const double *p, double x) { return pow(1.0 + pow(x / p[1], -p[0]), -p[2]); }
/* cdfs.rs:39-45 */
static double cdf_uniform(const double *p, double x) {
    if (x < p[0]) return 0.0;
    if (x > p[1]) return 1.0;
    return (x - p[0]) / (p[1] - p[0]);
}
/* cdfs.rs:56-63 */
static double cdf_kumaraswamy(const double *p, double x) {
    if (x < p[0]) return 0.0;
    if (x > p[1]) return 1.0;
    double z = (x - p[0]) / (p[1] - p[0]);
    return 1.0 - pow(1.0 - pow(z, p[2]), p[3]);
}

/* src/locohd/weight_function.rs:22-93 (WeightFunction::build) */
int orc_wf_validate(int kind, const double *p, int np) {
    switch (kind) {
    case WF_HYPER_EXP:
        if (np % 2 != 0) FAIL(ORC_ERR, "For function \"hyper_exp\" there must be an even number of parameters!");
        for (int i = 0; i < np; ++i)
            if (p[i] <= 0.0) FAIL(ORC_ERR, "For function \"hyper_exp\" all parameters must be positive!");
        return ORC_OK;
    case WF_DAGUM:
        if (np != 3) FAIL(ORC_ERR, "For function \"dagum\" there must be exactly 3 parameters!");
        if (p[0] < 0.0 || p[1] < 0.0 || p[2] < 0.0) FAIL(ORC_ERR, "For function \"dagum\" all parameters must be positive!");
        return ORC_OK;
    case WF_UNIFORM:
        if (np != 2) FAIL(ORC_ERR, "For function \"uniform\" there must be exactly 2 parameters!");
        if (p[0] < 0.0) FAIL(ORC_ERR, "For function \"uniform\" the first parameter must be non-negative!");
        if (p[1] <= 0.0) FAIL(ORC_ERR, "For function \"uniform\" the second parameter must be positive!");
        if (p[0] >= p[1]) FAIL(ORC_ERR, "For function \"uniform\" the first parameter must be smaller than the second!");
        return ORC_OK;
    case WF_KUMARASWAMY:
        if (np != 4) FAIL(ORC_ERR, "For function \"kumaraswamy\" there must be exactly 4 parameters!");
        if (p[0] < 0.0) FAIL(ORC_ERR, "For function \"kumaraswamy\" the first parameter must be non-negative!");
        if (p[1] <= 0.0 || p[2] <= 0.0 || p[3] <= 0.0)
            FAIL(ORC_ERR, "For function \"kumaraswamy\" after the first parameter all parameters must be positive!");
        if (p[0] >= p[1]) FAIL(ORC_ERR, "For function \"kumaraswamy\" the first parameter must be smaller than the second!");
        return ORC_OK;
    default:
        FAIL(ORC_ERR, "No function implemented with this name!");
    }
}

static double cdf_eval(int kind, const double *p, int np, double x) {
    switch (kind) {
    case WF_HYPER_EXP: return cdf_hyper_exp(p, np, x);
    case WF_DAGUM: return cdf_dagum(p, x);
    case WF_UNIFORM: return cdf_uniform(p, x);
    default: return cdf_kumaraswamy(p, x);
    }
}

/* weight_function.rs:95-103 */
int orc_wf_integral_point(int kind, const double *p, int np, double x, double *out) {
    if (x < 0.0) FAIL(ORC_ERR, "Invalid input value: %g. All values must be non-negative!", x);
    *out = cdf_eval(kind, p, np, x);
    return ORC_OK;
}
/* weight_function.rs:118-120: CDF(to) is evaluated first, then CDF(from) */
int orc_wf_integral_range(int kind, const double *p, int np, double from, double to, double *out) {
    double hi, lo;
    int rc;
    if ((rc = orc_wf_integral_point(kind, p, np, to, &hi))) return rc;
    if ((rc = orc_wf_integral_point(kind, p, np, from, &lo))) return rc;
    *out = hi - lo;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Statistical distances.  kind: 0 Hellinger[e], 1 Kolmogorov-Smirnov[], 2 Kullback-Leibler[eps],
 * 3 Renyi[alpha, eps]
 * ---------------------------------------------------------------------------------------------- */
enum { SD_HELLINGER = 0, SD_KS = 1, SD_KL = 2, SD_RENYI = 3 };

/* statistical_distances.rs:96-121 */
int orc_sd_validate(int kind, int np) {
    static const int want[4] = {1, 0, 1, 2};
    if (kind < 0 || kind > 3) FAIL(ORC_ERR, "Invalid statistical distance name!");
    if (np != want[kind]) FAIL(ORC_ERR, "Invalid number of parameters for the statistical distance: %d", np);
    return ORC_OK;
}

/* statistical_distances.rs:4-10 */
static double sd_hellinger(const double *p1, const double *p2, int n, double e) {
    double dist = 0.0;
    for (int i = 0; i < n; ++i) dist += pow(fabs(pow(p1[i], 1.0 / e) - pow(p2[i], 1.0 / e)), e);
    return pow(dist / 2.0, 1.0 / e);
}
/* :12-21 (max_by with partial_cmp().unwrap(): NaN panics; last maximum wins, value identical) */
static int sd_ks(const double *p1, const double *p2, int n, double *out) {
    double best = 0.0;
    for (int i = 0; i < n; ++i) {
        double d = fabs(p1[i] - p2[i]);
        if (isnan(d)) return ORC_PANIC;
        if (i == 0 || d >= best) best = d;
    }
    *out = best;
    return ORC_OK;
}
/* :23-29 */
static double sd_kl(const double *p1, const double *p2, int n, double eps) {
    double dist = 0.0;
    for (int i = 0; i < n; ++i) dist += p1[i] * log((p1[i] + eps) / (p2[i] + eps));
    return dist;
}
/* :31-78 */
static int sd_renyi(const double *p1, const double *p2, int n, double alpha, double eps, double *out) {
    if (alpha == 1.0) { *out = sd_kl(p1, p2, n, eps); return ORC_OK; }
    if (alpha == INFINITY) {
        double best = 0.0;
        for (int i = 0; i < n; ++i) {
            double r = (p1[i] + eps) / (p2[i] + eps);
            if (isnan(r)) return ORC_PANIC;
            if (i == 0 || r >= best) best = r;
        }
        *out = log(best);
        return ORC_OK;
    }
    if (alpha == 0.0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) if (p1[i] > 0.0) s += p2[i];
        *out = -log(s);
        return ORC_OK;
    }
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += p1[i] * pow((p1[i] + eps) / (p2[i] + eps), alpha - 1.0);
    *out = log(s) / (alpha - 1.0);
    return ORC_OK;
}

/* statistical_distances.rs:123-142 (StatisticalDistance::run) */
int orc_sd_run(int kind, const double *prm, const double *p1, const double *p2, int n, double *out) {
    switch (kind) {
    case SD_HELLINGER: *out = sd_hellinger(p1, p2, n, prm[0]); return ORC_OK;
    case SD_KS: if (n == 0) return ORC_PANIC; return sd_ks(p1, p2, n, out);
    case SD_KL: *out = sd_kl(p1, p2, n, prm[0]); return ORC_OK;
    case SD_RENYI: return sd_renyi(p1, p2, n, prm[0], prm[1], out);
    default: FAIL(ORC_ERR, "Invalid statistical distance name!");
    }
}

/* ------------------------------------------------------------------------------------------------
 * LoCoHD instance configuration (the fields of `struct LoCoHD`, src/locohd.rs:42-55)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_categories;           /* categories.len() after HashMap de-duplication (:312-316) */
    const double *category_weights; /* [n_categories] (:319-346) */
    int32_t sd_kind;                /* statistical_distance (:365-370) */
    double sd_params[2];
    /* tag pairing rule (:357-362; tag_pairing_rule.rs:5-21) */
    int32_t tag_mode;               /* 0 = WithoutList, 1 = WithList */
    int32_t tag_accept_same;
    int32_t tag_accepted_pairs;
    int32_t tag_ordered;
    const int32_t *tag_pairs;       /* [2 * n_tag_pairs] interned (first, second) */
    int64_t n_tag_pairs;
} orc_config;

typedef struct {
    int32_t kind;
    int32_t n_params;
    const double *params;
} orc_wf;

/* src/locohd.rs:305-346 (LoCoHD::build validation) */
int orc_config_validate(int64_t n_categories_given, int64_t n_categories_map, const double *weights,
                        int64_t n_weights) {
    if (n_categories_given == 0) FAIL(ORC_ERR, "The number of possible categories (primitive types) cannot be zero!");
    if (n_weights != n_categories_map)
        FAIL(ORC_ERR, "LoCoHD parameters 'categories' and 'category_weights' must have the same lengths! "
                      "Instead, they have lengths of %lld vs. %lld!", (long long)n_categories_map, (long long)n_weights);
    int64_t bad = 0;
    for (int64_t i = 0; i < n_weights; ++i) if (weights[i] <= 0.0) ++bad;
    if (bad > 0)
        FAIL(ORC_ERR, "LoCoHD parameter 'category_weights' must only contain positive values! "
                      "Instead, it contains %lld non-positive values!", (long long)bad);
    return ORC_OK;
}

/* src/locohd/tag_pairing_rule.rs:49-75 */
int orc_tag_pair_accepted(const orc_config *c, int32_t t0, int32_t t1) {
    if (c->tag_mode == 0) {
        int accepted = (t0 == t1);
        if (!c->tag_accept_same) accepted = !accepted;
        return accepted;
    }
    int accepted = 0;
    for (int64_t i = 0; i < c->n_tag_pairs && !accepted; ++i)
        accepted = (c->tag_pairs[2 * i] == t0 && c->tag_pairs[2 * i + 1] == t1);
    if (!c->tag_ordered)
        for (int64_t i = 0; i < c->n_tag_pairs && !accepted; ++i)
            accepted = (c->tag_pairs[2 * i] == t1 && c->tag_pairs[2 * i + 1] == t0);
    if (!c->tag_accepted_pairs) accepted = !accepted;
    return accepted;
}

/* ------------------------------------------------------------------------------------------------
 * PMF system, src/locohd/pmf.rs:13-88
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_config *cfg;
    double *pmf1, *pmf2, *n1, *n2; /* n1,n2: scratch for the normalised form */
} pmf_system;

static int pmf_update(const orc_config *cfg, double *pmf, int32_t cat) { /* pmf.rs:34-63 */
    if (cat < 0 || cat >= cfg->n_categories) FAIL(ORC_ERR, "Category (with index %d) not found!", cat);
    pmf[cat] += cfg->category_weights[cat];
    return ORC_OK;
}

static int pmf_calculate_distance(pmf_system *s, double *out) { /* pmf.rs:65-88 */
    int C = s->cfg->n_categories;
    double norm1 = 0.0, norm2 = 0.0;
    for (int i = 0; i < C; ++i) norm1 += s->pmf1[i];
    for (int i = 0; i < C; ++i) norm2 += s->pmf2[i];
    if (norm1 == 0.0) FAIL(ORC_ERR, "Zero norm error for PMF1");
    if (norm2 == 0.0) FAIL(ORC_ERR, "Zero norm error for PMF2");
    for (int i = 0; i < C; ++i) s->n1[i] = s->pmf1[i] / norm1;
    for (int i = 0; i < C; ++i) s->n2[i] = s->pmf2[i] / norm2;
    return orc_sd_run(s->cfg->sd_kind, s->cfg->sd_params, s->n1, s->n2, C, out);
}

/* ------------------------------------------------------------------------------------------------
 * src/locohd.rs:61-226  LoCoHD::stat_dist_integral
 * ---------------------------------------------------------------------------------------------- */
#define TRY(x) do { int rc_ = (x); if (rc_) { free(buf); return rc_; } } while (0)

int orc_stat_dist_integral(const orc_config *cfg, const int32_t *seq_a, const double *dists_a, int64_t len_a,
                           int64_t len_dists_a, const int32_t *seq_b, const double *dists_b, int64_t len_b,
                           int64_t len_dists_b, const orc_wf *wf, double *out) {
    double *buf = NULL;
    /* :70-73 */
    if (len_a != len_dists_a || len_b != len_dists_b) FAIL(ORC_ERR, "Lists seq and dists must have equal lengths!");
    /* :74 indexes dists[0] unconditionally: an empty list is a Rust panic */
    if (len_a == 0 || len_b == 0) FAIL(ORC_PANIC, "index out of bounds: the len is 0 but the index is 0");
    /* :74-77 */
    if (dists_a[0] != 0.0 || dists_b[0] != 0.0) FAIL(ORC_ERR, "The dists list must start with a distance of 0!");

    int C = cfg->n_categories;
    buf = (double *)calloc((size_t)4 * C, sizeof(double));
    pmf_system s = {cfg, buf, buf + C, buf + 2 * C, buf + 3 * C};
    /* :82-84 */
    TRY(pmf_update(cfg, s.pmf1, seq_a[0]));
    TRY(pmf_update(cfg, s.pmf2, seq_b[0]));

    int64_t idx_a = 0, idx_b = 0; /* :89-92 */
    double sd_integral = 0.0, dist_buffer = 0.0, h, dw, new_dist;

    /* :97-130 */
    while (idx_a < len_a - 1 && idx_b < len_b - 1) {
        TRY(pmf_calculate_distance(&s, &h));
        if (dists_a[idx_a + 1] < dists_b[idx_b + 1]) {
            idx_a += 1;
            TRY(pmf_update(cfg, s.pmf1, seq_a[idx_a]));
            new_dist = dists_a[idx_a];
        } else if (dists_a[idx_a + 1] > dists_b[idx_b + 1]) {
            idx_b += 1;
            TRY(pmf_update(cfg, s.pmf2, seq_b[idx_b]));
            new_dist = dists_b[idx_b];
        } else if (dists_a[idx_a + 1] == dists_b[idx_b + 1]) {
            idx_a += 1;
            idx_b += 1;
            TRY(pmf_update(cfg, s.pmf1, seq_a[idx_a]));
            TRY(pmf_update(cfg, s.pmf2, seq_b[idx_b]));
            new_dist = dists_a[idx_a];
        } else { /* :124 unreachable!() -- reached only with NaN */
            free(buf);
            FAIL(ORC_PANIC, "internal error: entered unreachable code");
        }
        TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dist_buffer, new_dist, &dw));
        sd_integral += dw * h;
        dist_buffer = new_dist;
    }

    if (idx_b < len_b - 1) { /* :134-171 */
        TRY(pmf_calculate_distance(&s, &h));
        idx_b += 1;
        TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dists_a[len_a - 1], dists_b[idx_b], &dw));
        sd_integral += dw * h;
        TRY(pmf_update(cfg, s.pmf2, seq_b[idx_b]));
        while (idx_b < len_b - 1) {
            idx_b += 1;
            TRY(pmf_calculate_distance(&s, &h));
            TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dists_b[idx_b - 1], dists_b[idx_b], &dw));
            sd_integral += dw * h;
            TRY(pmf_update(cfg, s.pmf2, seq_b[idx_b]));
        }
        TRY(pmf_calculate_distance(&s, &h));
        TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dists_b[len_b - 1], INFINITY, &dw));
        sd_integral += dw * h;
    } else if (idx_a < len_a - 1) { /* :173-210 */
        TRY(pmf_calculate_distance(&s, &h));
        idx_a += 1;
        TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dists_b[len_b - 1], dists_a[idx_a], &dw));
        sd_integral += dw * h;
        TRY(pmf_update(cfg, s.pmf1, seq_a[idx_a]));
        while (idx_a < len_a - 1) {
            idx_a += 1;
            TRY(pmf_calculate_distance(&s, &h));
            TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dists_a[idx_a - 1], dists_a[idx_a], &dw));
            sd_integral += dw * h;
            TRY(pmf_update(cfg, s.pmf1, seq_a[idx_a]));
        }
        TRY(pmf_calculate_distance(&s, &h));
        TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dists_a[len_a - 1], INFINITY, &dw));
        sd_integral += dw * h;
    } else if (idx_a == len_a - 1 && idx_b == len_b - 1) { /* :212-221 */
        TRY(pmf_calculate_distance(&s, &h));
        TRY(orc_wf_integral_range(wf->kind, wf->params, wf->n_params, dists_a[len_a - 1], INFINITY, &dw));
        sd_integral += dw * h;
    } else {
        free(buf);
        FAIL(ORC_PANIC, "internal error: entered unreachable code");
    }
    free(buf);
    *out = sd_integral;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * src/locohd/utils.rs
 * ---------------------------------------------------------------------------------------------- */
/* utils.rs:1-8: sum of (a-b).powf(2.) in axis order, then .powf(0.5) */
double orc_euclidean_distance(const double *a, const double *b) {
    double distance = 0.0;
    for (int k = 0; k < 3; ++k) {
        double d = a[k] - b[k];
        distance += d * d;
    }
    return sqrt(distance);
}

/* utils.rs:25-39: stable argsort by distance (slice::sort_by is a stable merge sort; a NaN makes
 * partial_cmp().unwrap() panic), then gather */
static int stable_argsort(const double *d, int64_t n, int64_t *idx, int64_t *tmp) {
    for (int64_t i = 0; i < n; ++i) {
        if (isnan(d[i]) && n > 1) return ORC_PANIC;
        idx[i] = i;
    }
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (d[idx[j]] < d[idx[i]]) ? idx[j++] : idx[i++];
            while (i < mid) tmp[k++] = idx[i++];
            while (j < hi) tmp[k++] = idx[j++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
    return ORC_OK;
}

int orc_sort_together(const double *dists, const int32_t *cats, int64_t n, double *out_d, int32_t *out_c) {
    int64_t *idx = (int64_t *)malloc((size_t)(2 * n + 1) * sizeof(int64_t));
    int rc = stable_argsort(dists, n, idx, idx + n);
    if (rc == ORC_OK)
        for (int64_t i = 0; i < n; ++i) {
            out_d[i] = dists[idx[i]];
            out_c[i] = cats[idx[i]];
        }
    free(idx);
    if (rc) FAIL(rc, "called `Option::unwrap()` on a `None` value (NaN distance)");
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * A tiny parallel-for standing in for the per-instance rayon pool (src/locohd.rs:373-383,446,557).
 * Results are written by index, i.e. order-preserving like rayon's indexed collect.
 * ---------------------------------------------------------------------------------------------- */
typedef int (*row_fn)(void *ctx, int64_t i);
typedef struct {
    row_fn fn;
    void *ctx;
    int64_t n;
    int64_t next; /* claimed with an atomic add (work stealing in chunks, like rayon's splitting) */
    int err;
} pf_shared;

static void *pf_worker(void *arg) {
    pf_shared *s = (pf_shared *)arg;
    for (;;) {
        int64_t lo = __atomic_fetch_add(&s->next, 64, __ATOMIC_RELAXED);
        if (lo >= s->n) break;
        int64_t hi = lo + 64 < s->n ? lo + 64 : s->n;
        for (int64_t i = lo; i < hi; ++i) {
            int rc = s->fn(s->ctx, i);
            if (rc) {
                int cur = __atomic_load_n(&s->err, __ATOMIC_RELAXED);
                while ((cur == 0 || rc > cur) && !__atomic_compare_exchange_n(&s->err, &cur, rc, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            }
        }
    }
    return NULL;
}

static int parallel_for(row_fn fn, void *ctx, int64_t n, int n_threads) {
    if (n_threads <= 1) {
        int err = 0;
        for (int64_t i = 0; i < n; ++i) {
            int rc = fn(ctx, i);
            if (rc && (err == 0 || rc > err)) err = rc;
        }
        return err;
    }
    pf_shared sh = {fn, ctx, n, 0, 0};
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, pf_worker, &sh);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    return sh.err;
}

/* ------------------------------------------------------------------------------------------------
 * src/locohd.rs:410-458  from_dmxs.  dmx_x is row-major [n_rows][ld_x]; each row is co-sorted with
 * the whole seq (utils.rs:25-39: the mask has dists.len() entries and indexes cats => a row longer
 * than seq panics, a shorter one silently uses a prefix of seq).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_config *cfg;
    const int32_t *seq_a, *seq_b;
    int64_t len_seq_a, len_seq_b;
    const double *dmx_a, *dmx_b;
    int64_t ld_a, ld_b;
    const orc_wf *wfs;
    const int32_t *wf_idx; /* NULL => wfs[0] for every row */
    double *out;
} dmx_ctx;

static int dmx_row(void *vctx, int64_t i) {
    dmx_ctx *c = (dmx_ctx *)vctx;
    if (c->ld_a > c->len_seq_a || c->ld_b > c->len_seq_b) FAIL(ORC_PANIC, "index out of bounds (row longer than seq)");
    int64_t na = c->ld_a, nb = c->ld_b;
    double *da = (double *)malloc((size_t)(na + nb + 2) * sizeof(double));
    int32_t *ca = (int32_t *)malloc((size_t)(na + nb + 2) * sizeof(int32_t));
    double *db = da + na;
    int32_t *cb = ca + na;
    int rc = orc_sort_together(c->dmx_a + i * c->ld_a, c->seq_a, na, da, ca);
    if (!rc) rc = orc_sort_together(c->dmx_b + i * c->ld_b, c->seq_b, nb, db, cb);
    if (!rc) rc = orc_stat_dist_integral(c->cfg, ca, da, na, na, cb, db, nb, nb, &c->wfs[c->wf_idx ? c->wf_idx[i] : 0], &c->out[i]);
    free(da);
    free(ca);
    return rc;
}

int orc_from_dmxs(const orc_config *cfg, const int32_t *seq_a, int64_t len_seq_a, const int32_t *seq_b,
                  int64_t len_seq_b, const double *dmx_a, int64_t rows_a, int64_t ld_a, const double *dmx_b,
                  int64_t rows_b, int64_t ld_b, const orc_wf *wfs, const int32_t *wf_idx, int n_threads,
                  double *out) {
    if (rows_a != rows_b) /* :420-428 */
        FAIL(ORC_ERR, "Expected matrices with the same length, got lengths %lld and %lld!", (long long)rows_a, (long long)rows_b);
    dmx_ctx c = {cfg, seq_a, seq_b, len_seq_a, len_seq_b, dmx_a, dmx_b, ld_a, ld_b, wfs, wf_idx, out};
    int rc = parallel_for(dmx_row, &c, rows_a, n_threads);
    if (rc == ORC_ERR) /* :448-452: the specific message is replaced */
        FAIL(ORC_ERR, "The stat_dist_integral function returned an error during the LoCoHD calculations!");
    return rc;
}

/* src/locohd.rs:463-476  from_coords (utils.rs:10-22 builds both dense matrices first) */
int orc_from_coords(const orc_config *cfg, const int32_t *seq_a, int64_t len_seq_a, const int32_t *seq_b,
                    int64_t len_seq_b, const double *xyz_a, int64_t n_a, const double *xyz_b, int64_t n_b,
                    const orc_wf *wfs, const int32_t *wf_idx, int n_threads, double *out) {
    double *ma = (double *)calloc((size_t)(n_a * n_a + n_b * n_b + 1), sizeof(double));
    double *mb = ma + n_a * n_a;
    for (int64_t i = 0; i < n_a; ++i)
        for (int64_t j = i + 1; j < n_a; ++j) ma[i * n_a + j] = ma[j * n_a + i] = orc_euclidean_distance(xyz_a + 3 * i, xyz_a + 3 * j);
    for (int64_t i = 0; i < n_b; ++i)
        for (int64_t j = i + 1; j < n_b; ++j) mb[i * n_b + j] = mb[j * n_b + i] = orc_euclidean_distance(xyz_b + 3 * i, xyz_b + 3 * j);
    int rc = orc_from_dmxs(cfg, seq_a, len_seq_a, seq_b, len_seq_b, ma, n_a, n_a, mb, n_b, n_b, wfs, wf_idx, n_threads, out);
    free(ma);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * kd-tree crate ^0.6 restated (see header): implicit balanced tree in an index array.
 * ---------------------------------------------------------------------------------------------- */
static void kd_select(int64_t *idx, int64_t lo, int64_t hi, int64_t nth, const double *xyz, int axis) {
    /* quickselect: afterwards idx[nth] holds the element of rank nth on `axis` (select_nth_unstable_by) */
    while (hi - lo > 1) {
        double pivot = xyz[3 * idx[lo + (hi - lo) / 2] + axis];
        int64_t i = lo, j = hi - 1;
        while (i <= j) {
            while (xyz[3 * idx[i] + axis] < pivot) ++i;
            while (xyz[3 * idx[j] + axis] > pivot) --j;
            if (i <= j) { int64_t t = idx[i]; idx[i] = idx[j]; idx[j] = t; ++i; --j; }
        }
        if (nth <= j) hi = j + 1;
        else if (nth >= i) lo = i;
        else return;
    }
}
static void kd_build(int64_t *idx, int64_t lo, int64_t hi, const double *xyz, int axis) {
    if (hi - lo < 2) return;
    int64_t mid = lo + (hi - lo) / 2;
    kd_select(idx, lo, hi, mid, xyz, axis);
    kd_build(idx, lo, mid, xyz, (axis + 1) % 3);
    kd_build(idx, mid + 1, hi, xyz, (axis + 1) % 3);
}
typedef struct { int64_t *v; int64_t n, cap; } i64vec;
static void push(i64vec *r, int64_t x) {
    if (r->n == r->cap) { r->cap = r->cap ? 2 * r->cap : 64; r->v = (int64_t *)realloc(r->v, (size_t)r->cap * sizeof(int64_t)); }
    r->v[r->n++] = x;
}
static void kd_within_box(const int64_t *idx, int64_t lo, int64_t hi, const double *xyz, int axis, const double *q,
                          double r, i64vec *res) {
    if (hi <= lo) return;
    int64_t mid = lo + (hi - lo) / 2;
    const double *p = xyz + 3 * idx[mid];
    if (p[axis] < q[axis] - r) kd_within_box(idx, mid + 1, hi, xyz, (axis + 1) % 3, q, r, res);
    else if (p[axis] > q[axis] + r) kd_within_box(idx, lo, mid, xyz, (axis + 1) % 3, q, r, res);
    else {
        int inside = 1;
        for (int k = 0; k < 3; ++k) if (p[k] < q[k] - r || p[k] > q[k] + r) inside = 0;
        if (inside) push(res, idx[mid]);
        kd_within_box(idx, lo, mid, xyz, (axis + 1) % 3, q, r, res);
        kd_within_box(idx, mid + 1, hi, xyz, (axis + 1) % 3, q, r, res);
    }
}
/* within_radius: box walk, then retain(sum(diff*diff) < radius*radius) */
static void kd_within_radius(const int64_t *idx, int64_t n, const double *xyz, const double *q, double r, i64vec *res) {
    res->n = 0;
    kd_within_box(idx, 0, n, xyz, 0, q, r, res);
    int64_t m = 0;
    for (int64_t t = 0; t < res->n; ++t) {
        const double *p = xyz + 3 * res->v[t];
        double distance = 0.0;
        for (int k = 0; k < 3; ++k) { double diff = p[k] - q[k]; distance += diff * diff; }
        if (distance < r * r) res->v[m++] = res->v[t];
    }
    res->n = m;
}

/* ------------------------------------------------------------------------------------------------
 * src/locohd.rs:479-567  from_primitives.  Primitive atoms arrive as SoA: xyz [n][3], category id,
 * interned tag.  anchors = [P][2] (idx into prim_a, idx into prim_b).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_config *cfg;
    const double *xyz_a, *xyz_b;
    const int32_t *cat_a, *cat_b, *tag_a, *tag_b;
    int64_t n_a, n_b;
    const int64_t *kd_a, *kd_b;
    const int64_t *anchors;
    const orc_wf *wfs;
    const int32_t *wf_idx;
    double threshold;
    double *out;
    int64_t *env_sizes; /* optional [P][2] */
} prim_ctx;

/* the closure env_from_idx, :514-542 */
static int env_from_idx(const prim_ctx *c, int side, int64_t anchor_idx, i64vec *nb, double **out_d, int32_t **out_c, int64_t *out_n) {
    const double *xyz = side ? c->xyz_b : c->xyz_a;
    const int32_t *cat = side ? c->cat_b : c->cat_a, *tag = side ? c->tag_b : c->tag_a;
    int64_t n = side ? c->n_b : c->n_a;
    if (anchor_idx < 0 || anchor_idx >= n) FAIL(ORC_PANIC, "index out of bounds: anchor index %lld, len %lld", (long long)anchor_idx, (long long)n);
    kd_within_radius(side ? c->kd_b : c->kd_a, n, xyz, xyz + 3 * anchor_idx, c->threshold, nb); /* :521 */
    double *d = (double *)malloc((size_t)(2 * nb->n + 1) * sizeof(double));
    int32_t *ct = (int32_t *)malloc((size_t)(2 * nb->n + 1) * sizeof(int32_t));
    int64_t m = 0;
    for (int64_t t = 0; t < nb->n; ++t) {
        int64_t p = nb->v[t];
        int accepted = (p == anchor_idx);                                   /* :525 ptr::eq */
        accepted |= orc_tag_pair_accepted(c->cfg, tag[anchor_idx], tag[p]); /* :526 */
        if (!accepted) continue;
        ct[m] = cat[p];                                                        /* :536 */
        d[m] = orc_euclidean_distance(xyz + 3 * anchor_idx, xyz + 3 * p);      /* :537 */
        ++m;
    }
    int rc = orc_sort_together(d, ct, m, d + m, ct + m); /* :541 */
    *out_d = d;
    *out_c = ct;
    *out_n = m;
    return rc;
}

static int prim_pair(void *vctx, int64_t i) {
    prim_ctx *c = (prim_ctx *)vctx;
    i64vec nb = {0, 0, 0};
    double *da = NULL, *db = NULL;
    int32_t *ca = NULL, *cb = NULL;
    int64_t na = 0, nbn = 0;
    int rc = env_from_idx(c, 0, c->anchors[2 * i], &nb, &da, &ca, &na);
    if (!rc) rc = env_from_idx(c, 1, c->anchors[2 * i + 1], &nb, &db, &cb, &nbn);
    if (!rc) {
        if (c->env_sizes) { c->env_sizes[2 * i] = na; c->env_sizes[2 * i + 1] = nbn; }
        rc = orc_stat_dist_integral(c->cfg, ca + na, da + na, na, na, cb + nbn, db + nbn, nbn, nbn,
                                    &c->wfs[c->wf_idx ? c->wf_idx[i] : 0], &c->out[i]);
    }
    free(da); free(db); free(ca); free(cb); free(nb.v);
    return rc;
}

int orc_from_primitives(const orc_config *cfg, const double *xyz_a, const int32_t *cat_a, const int32_t *tag_a,
                        int64_t n_a, const double *xyz_b, const int32_t *cat_b, const int32_t *tag_b, int64_t n_b,
                        const int64_t *anchors, int64_t n_pairs, const orc_wf *wfs, const int32_t *wf_idx,
                        double threshold, int n_threads, double *out, int64_t *env_sizes) {
    int64_t *kd_a = (int64_t *)malloc((size_t)(n_a + n_b + 1) * sizeof(int64_t)), *kd_b = kd_a + n_a;
    for (int64_t i = 0; i < n_a; ++i) kd_a[i] = i;
    for (int64_t i = 0; i < n_b; ++i) kd_b[i] = i;
    kd_build(kd_a, 0, n_a, xyz_a, 0); /* :504-510 */
    kd_build(kd_b, 0, n_b, xyz_b, 0);
    prim_ctx c = {cfg, xyz_a, xyz_b, cat_a, cat_b, tag_a, tag_b, n_a, n_b, kd_a, kd_b, anchors, wfs, wf_idx, threshold, out, env_sizes};
    int rc = parallel_for(prim_pair, &c, n_pairs, n_threads); /* :545-557 */
    free(kd_a);
    if (rc == ORC_ERR) /* :559-562 */
        FAIL(ORC_ERR, "The from_anchors function returned an error during the LoCoHD calculations!");
    return rc;
}

/* A plain-C client of include/loco_hd_hip.h: the reference's small known-answer case
 * (/root/reference/tests/test_locohd.py:27-52) and a tiny from_primitives call, through the C ABI only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "loco_hd_hip.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc_ = (call);                                                    \
        if (rc_ != LCHD_OK) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lchd_last_error()); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(void) {
    lchd_ctx *ctx = NULL;
    CHECK(lchd_ctx_create(-1, &ctx));

    /* uniform[0,4], categories O A B C, Hellinger-2 */
    double wf_params[2] = {0.0, 4.0};
    lchd_weight_function wf = {LCHD_WF_UNIFORM, 2, wf_params};
    double weights[4] = {1.0, 1.0, 1.0, 1.0};
    lchd_config cfg = {0};
    cfg.n_categories = 4;
    cfg.category_weights = weights;
    cfg.n_weight_functions = 1;
    cfg.weight_functions = &wf;
    cfg.sd_kind = LCHD_SD_HELLINGER;
    cfg.sd_n_params = 1;
    cfg.sd_params[0] = 2.0;
    cfg.tag_mode = 0;
    cfg.tag_accept_same = 1;

    int32_t seq[4] = {0, 1, 2, 3};
    double da[4] = {0.0, 1.0, 2.0, 3.0}, db[4] = {0.0, 1.0, 1.0, 1.0}, score = -1.0;
    CHECK(lchd_from_anchors(ctx, &cfg, seq, 4, da, 4, seq, 4, db, 4, 0, &score));
    printf("from_anchors %.17g\n", score);
    if (fabs(score - 0.22680537598265893) > 1e-12) { fprintf(stderr, "unexpected score\n"); return 2; }

    /* error path: dists must start at 0 (src/locohd.rs:74-77) */
    double bad[4] = {0.5, 1.0, 2.0, 3.0};
    if (lchd_from_anchors(ctx, &cfg, seq, 4, bad, 4, seq, 4, db, 4, 0, &score) != LCHD_EVALUE) { fprintf(stderr, "expected LCHD_EVALUE\n"); return 3; }

    /* from_primitives: a structure against itself scores exactly 0 for every anchor pair */
    enum { N = 64 };
    double xyz[N][3];
    int32_t cat[N], tag[N];
    int64_t anchors[N][2];
    double out[N];
    for (int i = 0; i < N; ++i) {
        xyz[i][0] = (i % 4) * 1.5; xyz[i][1] = ((i / 4) % 4) * 1.5; xyz[i][2] = (i / 16) * 1.5;
        cat[i] = i % 4; tag[i] = 0;
        anchors[i][0] = i; anchors[i][1] = i;
    }
    CHECK(lchd_from_primitives(ctx, &cfg, &xyz[0][0], cat, tag, N, &xyz[0][0], cat, tag, N, &anchors[0][0], NULL, N, 3.1, out));
    for (int i = 0; i < N; ++i)
        if (out[i] != 0.0) { fprintf(stderr, "self comparison gave %g at %d\n", out[i], i); return 4; }
    anchors[0][1] = N;  /* out of range -> the reference panics */
    if (lchd_from_primitives(ctx, &cfg, &xyz[0][0], cat, tag, N, &xyz[0][0], cat, tag, N, &anchors[0][0], NULL, N, 3.1, out) != LCHD_EPANIC) {
        fprintf(stderr, "expected LCHD_EPANIC\n");
        return 5;
    }
    lchd_ctx_destroy(ctx);
    printf("cabi smoke ok (%s)\n", lchd_version());
    return 0;
}

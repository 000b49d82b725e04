/* A plain-C client of include/loco_hd_hip.h: the reference's small known-answer case
 * (/root/reference/tests/test_locohd.py:27-52), a tiny from_primitives call, the same through a device group, and a ragged
 * from_dmxs call, through the C ABI only. */
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
    anchors[0][1] = 0;

    /* the same call through a device group (two contexts on device 0): what a non-Python binding uses for several GPUs */
    {
        lchd_group *grp = NULL;
        int32_t devs[2] = {0, 0};
        int64_t counts[2] = {0, 0};
        double ref[N];
        CHECK(lchd_from_primitives(ctx, &cfg, &xyz[0][0], cat, tag, N, &xyz[0][0], cat, tag, N, &anchors[0][0], NULL, N, 3.1, ref));
        CHECK(lchd_group_create(devs, 2, &grp));
        if (lchd_group_size(grp) != 2) { fprintf(stderr, "group size\n"); return 6; }
        CHECK(lchd_group_from_primitives(grp, &cfg, &xyz[0][0], cat, tag, N, &xyz[0][0], cat, tag, N, &anchors[0][0], NULL, N, 3.1, out));
        CHECK(lchd_group_last_counts(grp, counts));
        if (counts[0] + counts[1] != N) { fprintf(stderr, "group counts %lld + %lld\n", (long long)counts[0], (long long)counts[1]); return 7; }
        for (int i = 0; i < N; ++i)
            if (out[i] != ref[i]) { fprintf(stderr, "group score %d differs: %.17g vs %.17g\n", i, out[i], ref[i]); return 8; }
        lchd_group_destroy(grp);
    }

    /* from_dmxs with rows of different lengths (Vec<Vec<f64>>; utils.rs:25-39 co-sorts a row with a prefix of seq): row r holds
     * the distances of the first len[r] atoms; a row compared with itself scores exactly 0, and what lies beyond a row's length
     * (here: NaN, and a category outside the map) is never looked at */
    {
        enum { R = 5, W = 12 };
        double dmx[R][W];
        int32_t seq12[W], len[R] = {12, 7, 9, 5, 11};
        double sc[R];
        for (int j = 0; j < W; ++j) seq12[j] = j < 11 ? j % 4 : 99;
        for (int r = 0; r < R; ++r)
            for (int j = 0; j < W; ++j) {
                const double dx = xyz[j][0] - xyz[r][0], dy = xyz[j][1] - xyz[r][1], dz = xyz[j][2] - xyz[r][2];
                dmx[r][j] = j < len[r] ? sqrt(dx * dx + dy * dy + dz * dz) : NAN;
            }
        len[0] = 11;  /* (row 0 would reach the category outside the map) */
        dmx[0][11] = NAN;
        CHECK(lchd_from_dmxs_ragged(ctx, &cfg, seq12, W, seq12, W, &dmx[0][0], R, W, len, &dmx[0][0], R, W, len, NULL, sc));
        for (int r = 0; r < R; ++r)
            if (sc[r] != 0.0) { fprintf(stderr, "ragged self comparison gave %g at row %d\n", sc[r], r); return 9; }
        len[0] = 12;  /* now row 0 reaches it: the reference raises ValueError (pmf.rs:38-42) */
        dmx[0][11] = 9.0;
        if (lchd_from_dmxs_ragged(ctx, &cfg, seq12, W, seq12, W, &dmx[0][0], R, W, len, &dmx[0][0], R, W, len, NULL, sc) != LCHD_EVALUE) {
            fprintf(stderr, "expected LCHD_EVALUE for a category outside the map\n");
            return 10;
        }
    }
    lchd_ctx_destroy(ctx);
    printf("cabi smoke ok (%s)\n", lchd_version());
    return 0;
}

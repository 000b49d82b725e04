/*
 * loco_hd_hip.h -- C ABI of the MI355X-native LoCoHD scoring core (libloco_hd_hip.so).
 *
 * This is the drop-in boundary that replaces the reference's PyO3 extension module
 * `loco_hd.loco_hd` (/root/reference/src/lib.rs:9-17, Cargo.toml:6-9 `crate-type = ["cdylib"]`) for
 * the scoring path only.  Every entry point names the reference interface it replaces.  Signatures
 * use plain pointers and sizes only (no torch / pyo3 / C++ types); all functions return 0 on success
 * and a non-zero lchd_status otherwise, with the message available from lchd_last_error() on the
 * calling thread.  Status LCHD_EVALUE corresponds to the reference's PyValueError, LCHD_EPANIC to a
 * Rust panic (pyo3 PanicException) in the reference, LCHD_EDEVICE to a HIP failure / missing GPU and
 * LCHD_EUNSUPPORTED to an input the reference accepts but this build cannot run yet.  There is NO CPU
 * fallback: every scoring entry point launches HIP kernels on the context's device.
 *
 * Strings never cross the boundary: a category is the index the reference's HashMap would give it
 * (src/locohd.rs:312-316; -1 = "not in the map", src/locohd/pmf.rs:38-42), a tag is an interned
 * integer (equal strings <=> equal integers; only equality is ever used,
 * src/locohd/tag_pairing_rule.rs:49-75).  The caller owns every buffer; the library never frees or
 * retains caller memory beyond the call (mirrors the reference's copy-in / copy-out ownership).
 */
#ifndef LOCO_HD_HIP_H
#define LOCO_HD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LCHD_OK = 0,
    LCHD_EVALUE = 1,       /* reference: ValueError */
    LCHD_EPANIC = 2,       /* reference: Rust panic */
    LCHD_EDEVICE = 3,      /* HIP error / no device */
    LCHD_EUNSUPPORTED = 4  /* valid for the reference, not (yet) for this build */
} lchd_status;

/* weight_function.rs:22-93 function_name */
typedef enum { LCHD_WF_HYPER_EXP = 0, LCHD_WF_DAGUM = 1, LCHD_WF_UNIFORM = 2, LCHD_WF_KUMARASWAMY = 3 } lchd_wf_kind;
/* pmf/statistical_distances.rs:80-85 distance_name */
typedef enum { LCHD_SD_HELLINGER = 0, LCHD_SD_KOLMOGOROV_SMIRNOV = 1, LCHD_SD_KULLBACK_LEIBLER = 2, LCHD_SD_RENYI = 3 } lchd_sd_kind;

/* One WeightFunction (weight_function.rs:6-17): kind + parameter vector. */
typedef struct {
    int32_t kind;         /* lchd_wf_kind */
    int32_t n_params;
    const double *params; /* [n_params] */
} lchd_weight_function;

/* The fields of `struct LoCoHD` (src/locohd.rs:42-55) that the scoring path reads. */
typedef struct {
    int32_t n_categories;            /* categories.len() (:312-316); this build: 1..65534 (16-bit ids on the device) */
    const double *category_weights;  /* [n_categories], all > 0 (:319-346) */
    int32_t n_weight_functions;      /* 1 for WeightFunctionOptions::Single, dict size for ::Multiple (:27-32) */
    const lchd_weight_function *weight_functions;
    int32_t sd_kind;                 /* lchd_sd_kind (:365-370) */
    int32_t sd_n_params;
    double sd_params[2];
    int32_t tag_mode;                /* 0 = WithoutList{accept_same}, 1 = WithList{...} (tag_pairing_rule.rs:5-21) */
    int32_t tag_accept_same;
    int32_t tag_accepted_pairs;
    int32_t tag_ordered;
    const int32_t *tag_pairs;        /* [n_tag_pairs][2] interned (anchor tag, neighbour tag) */
    int64_t n_tag_pairs;
} lchd_config;

typedef struct lchd_ctx lchd_ctx; /* opaque: device, stream, device workspace; replaces the rayon pool (:53,373-383) */

const char *lchd_last_error(void);
const char *lchd_version(void);

/* ---- host-side leaves (no GPU work; same arithmetic headers as the kernels) ------------------- */
/* WeightFunction::build validation, weight_function.rs:22-93 */
int lchd_wf_validate(int32_t kind, const double *params, int32_t n_params);
/* WeightFunction::integral_vec / integral_point, weight_function.rs:95-116: out[i] = CDF(x[i]); x<0 -> LCHD_EVALUE */
int lchd_wf_cdf(int32_t kind, const double *params, int32_t n_params, const double *x, int64_t n, double *out);
/* StatisticalDistance::build validation, statistical_distances.rs:96-121 */
int lchd_sd_validate(int32_t kind, int32_t n_params);
/* StatisticalDistance::run, statistical_distances.rs:123-142 */
int lchd_sd_run(int32_t kind, const double *params, const double *p1, const double *p2, int32_t n, double *out);
/* LoCoHD::build validation, src/locohd.rs:305-346 (n_given = len of the list passed, n_map = len after de-dup) */
int lchd_config_validate(int64_t n_categories_given, int64_t n_categories_map, const double *weights, int64_t n_weights);

/* ---- context ----------------------------------------------------------------------------------- */
/* device < 0 => current HIP device.  Fails with LCHD_EDEVICE when no GPU is usable. */
int lchd_ctx_create(int32_t device, lchd_ctx **out);
void lchd_ctx_destroy(lchd_ctx *ctx);
/* Launch everything on this hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream. */
int lchd_ctx_set_stream(lchd_ctx *ctx, void *hip_stream);
/* Upload a LoCoHD configuration; later *_dev calls use it. (The host-pointer drivers below do this themselves.) */
int lchd_ctx_set_config(lchd_ctx *ctx, const lchd_config *cfg);

/* ---- the four reference drivers, host pointers in / host pointers out -------------------------- */
/* LoCoHD::from_anchors, src/locohd.rs:392-406 (+ stat_dist_integral :61-226).  wf_index selects
 * cfg->weight_functions[wf_index] (the shim resolves the key, :230-283).  len_* are passed separately so
 * the "Lists seq and dists must have equal lengths!" check (:70-73) lives behind the boundary. */
int lchd_from_anchors(lchd_ctx *ctx, const lchd_config *cfg, const int32_t *seq_a, int64_t len_seq_a, const double *dists_a,
                      int64_t len_dists_a, const int32_t *seq_b, int64_t len_seq_b, const double *dists_b,
                      int64_t len_dists_b, int32_t wf_index, double *out);

/* LoCoHD::from_dmxs, src/locohd.rs:410-458.  dmx_x is row-major [rows_x][cols_x]; wf_index NULL or [rows]. */
int lchd_from_dmxs(lchd_ctx *ctx, const lchd_config *cfg, const int32_t *seq_a, int64_t len_seq_a, const int32_t *seq_b,
                   int64_t len_seq_b, const double *dmx_a, int64_t rows_a, int64_t cols_a, const double *dmx_b,
                   int64_t rows_b, int64_t cols_b, const int32_t *wf_index, double *out);

/* The same with ragged rows (the reference's Vec<Vec<f64>> may hold rows of different lengths; utils.rs:25-39 sorts each row
 * with a prefix of seq): dmx_x is padded to [rows][cols_x], row r of side x has row_len_x[r] in 1..cols_x real entries; what lies
 * beyond a row's length -- matrix entries and seq categories alike -- is never looked at. */
int lchd_from_dmxs_ragged(lchd_ctx *ctx, const lchd_config *cfg, const int32_t *seq_a, int64_t len_seq_a, const int32_t *seq_b,
                          int64_t len_seq_b, const double *dmx_a, int64_t rows_a, int64_t cols_a, const int32_t *row_len_a,
                          const double *dmx_b, int64_t rows_b, int64_t cols_b, const int32_t *row_len_b, const int32_t *wf_index,
                          double *out);

/* LoCoHD::from_coords, src/locohd.rs:463-476.  xyz_x is [n_x][3]; out is [n_a]. */
int lchd_from_coords(lchd_ctx *ctx, const lchd_config *cfg, const int32_t *seq_a, int64_t len_seq_a, const int32_t *seq_b,
                     int64_t len_seq_b, const double *xyz_a, int64_t n_a, const double *xyz_b, int64_t n_b,
                     const int32_t *wf_index, double *out);

/* LoCoHD::from_primitives, src/locohd.rs:479-567.  Primitive atoms as SoA (xyz [n][3], category, tag);
 * anchors is [n_pairs][2] = (index into a, index into b); out[i] belongs to anchors[i] (order-preserving,
 * like the reference's indexed rayon collect). */
int lchd_from_primitives(lchd_ctx *ctx, const lchd_config *cfg, const double *xyz_a, const int32_t *cat_a,
                         const int32_t *tag_a, int64_t n_a, const double *xyz_b, const int32_t *cat_b,
                         const int32_t *tag_b, int64_t n_b, const int64_t *anchors, const int32_t *wf_index,
                         int64_t n_pairs, double threshold_distance, double *out);

/* ---- device-resident path (what bench.py and multi-structure callers use) --------------------- */
/* A primitive-atom structure resident in HBM as SoA (replaces the per-call Vec<PrimitiveAtom> clone at the
 * PyO3 boundary, src/locohd/primitive_atom.rs:4-16).  xyz/cat/tag are HOST pointers here. */
typedef struct lchd_cloud lchd_cloud;
int lchd_cloud_create(lchd_ctx *ctx, const double *xyz, const int32_t *cat, const int32_t *tag, int64_t n, lchd_cloud **out);
/* A BATCH of structures in one device object (additive; replaces a Python loop of from_primitives calls such as
 * python_codes/casp14/casp14_extend_with_locohd.py:42-88 or python_codes/trajectory_analyzer.py:112-120): the atoms of
 * all structures are concatenated, sid[i] in [0, n_struct) names the structure of atom i, and an environment only ever
 * contains atoms of its anchor's own structure.  Anchor indices are positions in the concatenated arrays, so one
 * lchd_from_primitives_dev call can score anchor pairs of many structure pairs (pass the same batch as `a` and `b`
 * for all-vs-all). */
int lchd_cloud_create_batch(lchd_ctx *ctx, const double *xyz, const int32_t *cat, const int32_t *tag, const int32_t *sid,
                            int64_t n, int32_t n_struct, lchd_cloud **out);
/* Atoms a cloud / batch / frames buffer currently holds (-1 for a null handle). */
int64_t lchd_cloud_size(const lchd_cloud *cloud);
/* Replace the coordinates of an existing cloud (MD frames: same atoms, new positions). Host pointer [n][3]. */
int lchd_cloud_set_coords(lchd_ctx *ctx, lchd_cloud *cloud, const double *xyz);
void lchd_cloud_destroy(lchd_ctx *ctx, lchd_cloud *cloud);

/* from_primitives with everything already on the device: d_anchors is a DEVICE pointer [n_pairs][2] int64,
 * d_wf_index a DEVICE pointer [n_pairs] int32 or NULL, d_out a DEVICE pointer [n_pairs] double.  Work is
 * enqueued on the context's stream; the call returns after the (tiny) status word has been read back, i.e.
 * d_out is complete on return.  Uses the configuration set by lchd_ctx_set_config. */
int lchd_from_primitives_dev(lchd_ctx *ctx, lchd_cloud *a, lchd_cloud *b, const int64_t *d_anchors,
                             const int32_t *d_wf_index, int64_t n_pairs, double threshold_distance, double *d_out);

/* LoCoHD::from_coords (src/locohd.rs:463-476) with both structures already on the device: pair r = (atom r of a, atom r of
 * b), every environment is the whole structure (no threshold, no tag rule).  d_wf_index: DEVICE [n] int32 or NULL, d_out:
 * DEVICE [n] double, complete on return.  Uses the configuration set by lchd_ctx_set_config. */
int lchd_from_coords_dev(lchd_ctx *ctx, lchd_cloud *a, lchd_cloud *b, const int32_t *d_wf_index, double *d_out);

/* Split form of lchd_from_primitives_dev: _async enqueues the whole pass on the context's stream and returns without
 * waiting; lchd_ctx_finish waits, re-runs the pass with a larger environment capacity if one overflowed, and returns
 * the status.  Between the two calls the host is free (e.g. to stage the next batch of frames). */
int lchd_from_primitives_dev_async(lchd_ctx *ctx, lchd_cloud *a, lchd_cloud *b, const int64_t *d_anchors,
                                   const int32_t *d_wf_index, int64_t n_pairs, double threshold_distance, double *d_out);
int lchd_ctx_finish(lchd_ctx *ctx);

/* Trajectory frames (python_codes/trajectory_analyzer.py:97-119: same atoms, new coordinates per frame).  A frames
 * buffer is a batch cloud with room for `capacity_frames` copies of `tmpl`'s atoms (categories and tags replicated on
 * the device once).  lchd_frames_load copies a HOST block xyz[n_frames][n_atoms][3] into pinned staging, then enqueues
 * the H2D copy, the SoA unpack and the bounding-box reduction on `hip_stream` (NULL = the context's stream) and
 * returns; passes that use the buffer are ordered behind the upload with events, so uploading chunk k+1 on a second
 * stream overlaps the scoring of chunk k (use two buffers). */
int lchd_frames_create(lchd_ctx *ctx, const lchd_cloud *tmpl, int32_t capacity_frames, lchd_cloud **out);
int lchd_frames_load(lchd_ctx *ctx, lchd_cloud *frames, const double *xyz, int32_t n_frames, void *hip_stream);

/* Frames given as SOURCE atoms (the step in front of the scoring path): the reference converts every
 * frame on the host -- PrimitiveAssigner.assign_primitive_structure, loco_hd/atom_converter_utils.py:95-129, and its
 * MD variant python_codes/trajectory_analyzer.py:37-74 -- where each primitive atom is np.mean(atom_coords, axis=0) over
 * the float32 coordinates of the atoms its typing-scheme element matched.  The match (regexes on residue / atom names)
 * depends on the topology only, so the host resolves it once into a CSR map and the device evaluates the centroids
 * for every frame with np.mean's float32 arithmetic (sequential adds in member order, one division), bit for bit.
 *   src_start [n_primitive_atoms + 1], src_idx [src_start[n]] : primitive atom p <- source atoms src_idx[src_start[p]..src_start[p+1])
 *   atom_xyz  HOST float32 [n_frames][n_src_atoms][3]
 * lchd_frames_load_atoms has the stream / overlap semantics of lchd_frames_load. */
int lchd_frames_set_sources(lchd_ctx *ctx, lchd_cloud *frames, const int32_t *src_start, const int32_t *src_idx,
                            int64_t n_src_atoms);
int lchd_frames_load_atoms(lchd_ctx *ctx, lchd_cloud *frames, const float *atom_xyz, int32_t n_frames, void *hip_stream);
/* The same with the source atoms already in HBM (d_atom_xyz is a DEVICE pointer; no staging copy), and the duration of
 * the most recent conversion kernel of this buffer in ms (HIP events; <0 unless lchd_ctx_enable_timing is on). */
int lchd_frames_load_atoms_dev(lchd_ctx *ctx, lchd_cloud *frames, const float *d_atom_xyz, int32_t n_frames, void *hip_stream);
double lchd_frames_last_convert_ms(lchd_ctx *ctx, lchd_cloud *frames);
/* Coordinates of a cloud / frames buffer back on the host as [n][3] f64 (n must equal the atoms it holds). */
int lchd_cloud_get_coords(lchd_ctx *ctx, lchd_cloud *cloud, double *xyz_out, int64_t n);

/* ---- multi-GPU ------------------------------------------------------------------------------------
 * The reference's parallelism lives INSIDE the core call: a thread pool that is a field of `LoCoHD` (src/locohd.rs:53,
 * 373-383) runs the anchor pairs of one call (:545-557, order-preserving collect).  The counterpart here is a GROUP of
 * devices inside one process: lchd_group_from_primitives has the signature and the semantics of lchd_from_primitives and
 * spreads the call's anchor pairs over the group's GPUs (both structures are replicated, pairs are binned by their side-A
 * anchor so that every device builds ~1/n of side A's environments, every device scores its bin concurrently, out[i] is
 * the score of anchors[i]).  A binding in any host language gets multi-GPU scoring from this one call; no torch, no MPI. */
typedef struct lchd_group lchd_group;
/* devices: HIP device ordinals (a device may be listed twice: two contexts on it); n_devices in [1, 64]. */
int lchd_group_create(const int32_t *devices, int32_t n_devices, lchd_group **out);
void lchd_group_destroy(lchd_group *group);
int32_t lchd_group_size(const lchd_group *group);
int lchd_group_from_primitives(lchd_group *group, const lchd_config *cfg, const double *xyz_a, const int32_t *cat_a,
                               const int32_t *tag_a, int64_t n_a, const double *xyz_b, const int32_t *cat_b,
                               const int32_t *tag_b, int64_t n_b, const int64_t *anchors, const int32_t *wf_index,
                               int64_t n_pairs, double threshold_distance, double *out);
/* Pairs the most recent lchd_group_from_primitives call gave to each device: counts_out[n_devices]. */
int lchd_group_last_counts(const lchd_group *group, int64_t *counts_out);

/* One process per GPU (torch.distributed / MPI style): every rank holds the whole pair list on its device and runs the
 * SAME deterministic partition (a pure function of the list, so no communication is needed to agree on it):
 *   bin(p) = floor(anchor_a(p) * 1024 / n_atoms_a),  rank(bin) = min(world - 1, floor(#pairs in lower bins * world / n_pairs)).
 * A list whose side-A partition is unbalanced (a rank would hold more than 1.25 n_pairs / world + 1 pairs: ONE reference anchor
 * against thousands, /root/reference/python_codes/kras_scan.py:46-52) is binned by its side-B anchors instead (n_atoms_b > 0),
 * and if that partition is unbalanced too (or n_atoms_b <= 0) cut into contiguous slices, rank(p) = floor(p * world / n_pairs):
 * the reference's par_iter balances any list (src/locohd.rs:545-557), so must this.
 * lchd_shard_plan_dev computes it (one kernel on a side stream of the context + a wait for that kernel only: the pair list
 * must be complete in device memory when it is called) and returns the pair count of every rank;
 * lchd_shard_select_dev compacts THIS rank's pairs (d_sel_anchors [counts[rank]][2], d_sel_index [counts[rank]] = their
 * positions in the full list; enqueued on the context's stream, no wait);
 * lchd_unshard_scores_dev, on the gathering rank, puts score k of rank r at its pair's original position:
 * d_gathered is [world][2][stride] doubles -- rank r's scores, then its d_sel_index reinterpreted as doubles. */
int lchd_shard_plan_dev(lchd_ctx *ctx, const int64_t *d_anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                        int32_t world, int64_t *counts_out);
int lchd_shard_select_dev(lchd_ctx *ctx, const int64_t *d_anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                          int32_t rank, int64_t *d_sel_anchors, int64_t *d_sel_index);
int lchd_unshard_scores_dev(lchd_ctx *ctx, const double *d_gathered, const int64_t *counts, int32_t world, int64_t stride,
                            double *d_out, int64_t n_pairs);

/* Per-kernel timing of the most recent *_dev / driver call, measured with hipEvents on the context's stream.
 * names: "cells" (cell lists + anchor de-duplication: one phase), "anchors" (always ~0, kept for callers of the first
 * version), "env", "sweep"; returns milliseconds, <0 if unknown name / timing disabled. */
int lchd_ctx_enable_timing(lchd_ctx *ctx, int32_t on);
double lchd_ctx_last_ms(lchd_ctx *ctx, const char *phase);
/* Environment statistics of the most recent call: sum over anchor pairs of (n_A + n_B) (points incl. anchors). */
int64_t lchd_ctx_last_env_points(lchd_ctx *ctx);
/* 1 if the most recent from_coords / from_dmxs call of the context ran the fused sort + sweep kernel (one launch per
 * call, nothing but the scores written: Hellinger-2, unit category weights, at most 16 categories, rows of 1 025 .. 20 480
 * points), 0 if it ran the row sort followed by the sweep (src/locohd.rs:410-476 either way). */
int32_t lchd_ctx_last_dense_fused(lchd_ctx *ctx);
/* Determinism switch.  The reference is ONE code path (src/locohd.rs:61-226): the same anchor pair gives the same bits whatever
 * else the call holds.  By default this library picks among several sweep kernels per call -- from the call's size, from what the
 * majority of its pairs look like and from the statistics of the context's previous pass -- and they sum a pair's intervals in
 * different per-lane orders: the same pair can differ by <= 1e-13 between calls of different shape or history.
 * on != 0 pins ONE sweep family (one pair per wavefront, global-memory tables; dense rows through the row sort + that sweep) and
 * switches every history-dependent choice off: a pair's score is then a function of the pair and the configuration alone --
 * bitwise equal across batch composition, call order, sharding and second passes -- at roughly half the default throughput.
 * on == 0 returns to the default selection.  Not allowed while an asynchronous call is pending.
 * (In every mode: points of DIFFERENT categories at EXACTLY the same distance from an anchor -- lattice coordinates -- enter an
 * environment in the order the cell list's atomics produced, which may differ from run to run; they span zero-width intervals, so only
 * the rounding of the running sums differs: a few 1e-16 in a handful of pairs.  Inputs without such ties are bit-reproducible.) */
int lchd_ctx_set_deterministic(lchd_ctx *ctx, int32_t on);
int32_t lchd_ctx_get_deterministic(lchd_ctx *ctx);
/* from_primitives passes the context has enqueued since it was created.  A call is one pass in the steady state; a pass is
 * repeated when an environment overflowed the capacity tried (src/locohd.rs:514-542 has no capacity) or when the sweep launch
 * set picked from the previous call's pair statistics did not cover this call's pairs. */
int64_t lchd_ctx_pass_count(lchd_ctx *ctx);
/* Regular passes so far whose side B was NOT de-duplicated: environment slot p belongs to anchor pair p and its anchor record is
 * written straight from the pair list -- taken when the previous regular pass of the context found (almost) every side-B anchor
 * used once (trajectory frames, (i, i) lists, a rank's partners under strong scaling).  The reference builds an environment per
 * pair and side as well (src/locohd.rs:514-554). */
int64_t lchd_ctx_per_pair_pass_count(lchd_ctx *ctx);
/* Second passes run so far.  Environments live in fixed-stride slots (512 points by default); the reference's environments have no
 * capacity (src/locohd.rs:514-542: a Vec per anchor).  When a FEW environments of a call do not fit their slots, only the pairs
 * that touch them are scored again -- a pass of their own with larger slots, its scores scattered over the first pass's --
 * instead of giving every environment of the call the larger slot. */
int64_t lchd_ctx_subset_pass_count(lchd_ctx *ctx);
/* Environment-store bytes (keys + categories, both sides) carved by the passes of the most recent from_primitives call, summed. */
int64_t lchd_ctx_last_store_bytes(lchd_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* LOCO_HD_HIP_H */

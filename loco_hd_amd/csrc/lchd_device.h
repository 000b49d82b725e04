// lchd_device.h -- internal interface between the C-ABI host layer (lchd_capi.hip) and the gfx950
// kernels (lchd_*.hip, one kernel family per translation unit).  Plain-old-data argument blocks + launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lchd {

constexpr int kMaxCategories = 255;   // categories travel as u8 on the device ...
constexpr int kHugeCategories = 65534;  // ... beyond kWideCategories: the same 16-bit ids (0xFFFF = not in the map), the sweep's per-category state in global memory
// scratch of one workgroup of k_sweep_wide<.., HUGE>: [C][64] u32 count columns | [C] u32 carry row | 2 x [64][C] f64 normalised vectors
inline size_t wide_scratch_bytes_per_wave(int C) { return (((size_t)C * 65 * 4 + 15) & ~(size_t)15) + (size_t)C * 64 * 16; }
constexpr int kWideCategories = 512;  // ... or, for 256 .. 512 categories, as u16 (CloudView::cat_hi, EnvStore::cat16): from_primitives and
                                      // from_anchors only, through k_env_cells<.., uint16_t> and k_sweep_wide<.., CAT16> (its per-lane
                                      // category counts, 256 bytes per category, must fit the LDS)
#ifndef LCHD_SWEEP_EPL
#define LCHD_SWEEP_EPL 6
#endif
constexpr int kSweepEPL = LCHD_SWEEP_EPL;  // merged events per lane per tile in the sweep kernel (16-bit counts, LDS tables, <= 16 category slots; the generic distances)
constexpr int kSweepTile = 64 * kSweepEPL;
constexpr int kMetaPartials = 4096;   // most workgroups of k_pair_meta
constexpr uint64_t kPadKey = ~0ull;   // sorts after every valid (non-negative, non-NaN) f64 bit pattern

// status word written by kernels (device memory, zeroed per call)
enum : uint32_t {
    ST_BAD_ANCHOR = 1u << 0,      // anchor index outside the cloud            (reference: panic, :521)
    ST_EMPTY_ENV = 1u << 1,       // environment without any point              (reference: panic, :74)
    ST_FIRST_NOT_ZERO = 1u << 2,  // sorted dists[0] != 0                       (reference: ValueError, :74-77)
    ST_BAD_CATEGORY = 1u << 3,    // category id outside [0, C)                 (reference: ValueError, pmf.rs:38-42)
    ST_ENV_OVERFLOW = 1u << 4,    // environment larger than the kernel variant's LDS capacity (retry bigger)
    ST_ZERO_NORM = 1u << 5,       // PMF norm 0                                 (reference: ValueError, pmf.rs:70-76)
    ST_BAD_DISTANCE = 1u << 6,    // negative / NaN distance in a matrix row    (reference: ValueError / panic)
    ST_BAD_WF = 1u << 7,          // weight-function index outside the table
    ST_ROW_RETRY = 1u << 8,       // a dense row of more than 16384 points defeated the segmented sort: repeat the call with k_env_rows
};
// Device-resident status of a pass.  Invariant between passes: flags == max_env == 0 -- the last workgroup of
// k_pair_meta copies the words the host needs into the host-mapped HostStatus and resets them, so a pass needs neither a
// memset in front of it nor a device-to-host copy behind it.  n_unique / n_small are plain-stored by every pass.
struct DeviceStatus {
    uint32_t flags;
    uint32_t max_env;        // largest environment seen (for the overflow retry)
    uint32_t n_unique[2];    // unique anchors per side
    uint32_t n_overflow[2];  // environments per side that did not fit their slot (entries of EnvSide::ovf_list); reset with the flags
    unsigned long long n_small;     // pairs under the pass's first-choice small rule (k_pair_meta): who sweeps them is decided on the device
    unsigned long long n_c8;        // pairs under SweepArgs::c8_rule (the second choice of a pass without a hint)
    uint32_t max_bound;      // k_env_group: most CANDIDATES of an anchor whose candidate table overflowed (an upper bound of its environment; max_env
                             //  then only says "more than the capacity"); reset with the flags
    uint32_t n_dup_b;        // a pass without de-duplication of side B (k_pair_anchor_recs): pairs whose side-B anchor another pair of the list had
                             //  marked before (a bit set in the side's flag region, one returning atomic per pair); reset with the flags
};
// Host-mapped (pinned, device-visible) mirror: written with plain stores only -- the snapshot by one thread of k_pair_meta,
// the error words by whichever sweep wavefront meets the (rare) condition; every writer of a word stores the same value.
struct HostStatus {
    uint32_t flags;          // DeviceStatus::flags at the end of the record pass (everything the kernels before the sweep reported)
    uint32_t max_env;
    uint32_t n_unique[2];
    unsigned long long n_small;
    uint32_t sweep_flags[8]; // word k != 0 <=> a sweep kernel reported status bit k (ST_* below)
    uint32_t snapshot_seq;   // pass counter written with the snapshot (the host checks that the pass it waited for got this far)
    uint32_t pad;
    unsigned long long n_duo, n_c8;  // pairs of at most kDuoTile merged events / with both environments <= 255 points (both always counted)
    uint32_t n_overflow[2];  // DeviceStatus::n_overflow at the end of the record pass
    uint32_t max_bound;      // DeviceStatus::max_bound
    uint32_t n_dup_b;        // DeviceStatus::n_dup_b
};

// Test / tuning hooks.  Read from the environment ONCE, when a context is created (lchd_ctx_create), and handed to the
// launchers by value: nothing in the launch path calls getenv().
struct Tuning {
    bool no_struct_cells = false;   // LCHD_NO_STRUCT_CELLS: always the generic (multi-pass, global atomics) cell list
    bool no_share = false;          // LCHD_NO_SHARED_ENVS: build both sides even when they are the same device object
    bool no_key_sets = false;       // LCHD_NO_KEY_SETS: weight-function dictionaries keep distance keys (the CDF is evaluated by the sweep, per event)
    bool no_cdf_keys = false;       // LCHD_NO_CDF_KEYS: environments keep distance keys even with a single weight function
    bool no_duo = false;            // LCHD_NO_DUO: never two pairs per wavefront
    bool force_wide = false;        // LCHD_FORCE_WIDE: k_sweep_wide for any category count
    bool force_generic = false;     // LCHD_FORCE_GENERIC: MODE_GEN even for Hellinger-2
    bool force_bigenv = false;      // LCHD_FORCE_BIGENV: the !LDSTAB sweep instantiations
    bool no_sweep_hint = false;     // (deterministic mode only) always launch all three sweep kernels and let the device decide
    bool no_inline_meta = false;    // LCHD_NO_INLINE_META: small calls also run k_pair_meta + the regular sweep kernels
    bool old_rows = false;          // LCHD_OLD_ROWS: dense rows through k_env_rows (three distance passes) for every length
    bool no_dense_fused = false;    // LCHD_NO_DENSE_FUSED: dense rows always through the two-kernel path (row sort, then sweep)
    bool no_count8 = false;         // LCHD_NO_COUNT8: never the 8-bit-count sweep
    bool no_c8_team = false;        // LCHD_NO_C8_TEAM: the 8-bit-count sweep always one pair per wavefront (k_sweep<.., CNT8>)
    bool no_overflow_subset = false;  // LCHD_NO_OVERFLOW_SUBSET: an overflowing environment repeats the WHOLE pass with larger slots (never only its pairs)
    bool no_env_group = false;      // LCHD_NO_ENV_GROUP: environments of the default capacity through k_env_cells (one per wavefront) too
    bool no_sd_inc = false;         // LCHD_NO_SD_INC: Kullback-Leibler / Renyi through the generic sweep even where k_sweep_inc applies
    int env_apw = 0;                // LCHD_ENV_APW: anchors per wavefront of k_env_group (0: chosen from the number of anchors)
    int force_cmax = 0;             // LCHD_FORCE_CMAX: at least this many category slots
    int per_pair = 0;               // LCHD_PER_PAIR: -1 never a side B without de-duplication, 1 whenever it applies, 0: from the previous pass (side-B anchors (almost) all unique)
    int pre_rows = 0;               // LCHD_PRE_ROWS: -1 never prefix-count rows next to the environments (the team sweeps build their chunk-start counts per tile), 1 also for small calls, 0: by the rule of prims_enqueue
};

struct WfEntry {
    int32_t kind, n_params, offset, pad;
};

// Device-resident LoCoHD configuration (src/locohd.rs:42-55).
struct DevConfig {
    int32_t n_categories;
    int32_t n_wf;
    int32_t sd_kind;
    int32_t tag_mode, tag_accept_same, tag_accepted_pairs, tag_ordered;
    int32_t n_tag_pairs;
    double sd_p0, sd_p1;
    const double* cat_w;        // [n_categories]
    const WfEntry* wf;          // [n_wf]
    const double* wf_params;    // concatenated
    const double* wf_finf;      // [n_wf] F(+inf) per weight function (host-evaluated; 1.0 except for degenerate dagum)
    const double* wf_inv;       // [n_wf] 1 / (the CDF's constant divisor): 1 / sum a_i (hyper_exp), 1 / (x_max - x_min) (uniform, kumaraswamy)
    const uint64_t* tag_pairs;  // sorted, (anchor_tag << 32) | neighbour_tag
    const double* pow_tab;      // Hellinger with exponent e != 2: [2][65536] k^(1/e), k^(-1/e) (context-owned, filled by set_config); else null
};

// SoA structure in HBM.
struct CloudView {
    const double *x, *y, *z;
    const uint8_t* cat;       // low byte of the category id
    const uint8_t* cat_hi;    // high byte (nullptr: every id fits a byte; 255 = not in the category map, with cat_hi: 0xFFFF)
    const int32_t* tag;
    int32_t n;
    const int32_t* sid;   // structure id per atom for a batch of structures (nullptr = one structure)
    int32_t n_struct;     // >= 1
    int32_t struct_size;  // > 0: every structure is exactly this many CONSECUTIVE atoms (frames buffers, regular batches)
};

struct __attribute__((aligned(16))) CellRec {
    double x, y, z;
    uint32_t tag;   // interned tag
    uint32_t cat;   // category id (255 = not in the category map)
};
static_assert(sizeof(CellRec) == 32, "CellRec is read as two 16-byte words");

// One unique anchor, written by k_compact_anchors so that the environment kernel starts from ONE record instead of the
// chain slot -> atom -> coordinates / tag / position.
struct __attribute__((aligned(8))) AnchorRec {
    double x, y, z;
    uint32_t tag;    // interned tag of the anchor
    uint32_t apos;   // its position in cell order
    int32_t sid;     // structure of a batch (0 otherwise)
    uint32_t atom;   // index in the cloud
};

// Uniform grid over one cloud (replaces KdTree::build_by_ordered_float, src/locohd.rs:504-510).
struct GridView {
    double min[3], inv[3];   // cell index = clamp(floor((p - min) * inv), 0, dim-1)
    double cell[3];          // 1 / inv: the cell edge (only used to skip neighbour cells that lie wholly outside the radius)
    int32_t dim[3];          // per structure; cell = ((sid * dim[2] + cz) * dim[1] + cy) * dim[0] + cx
    int32_t n_cells;         // n_struct * dim[0] * dim[1] * dim[2]
    const uint32_t* cell_start;  // [n_cells + 1]
    // points permuted into cell order, one 32-byte record each (two 16-byte loads fetch everything the radius search,
    // the tag filter and the environment need: no dependent second round of loads for the survivors)
    const CellRec* rec;
    const uint32_t* pos_of;      // atom -> its position in cell order (an anchor recognises itself by position)
};

// Sorted environments: env e occupies [e*stride, e*stride + len[e]).
struct EnvStore {
    uint64_t* key;   // f64 distance bit patterns, ascending
    uint8_t* cat;
    int32_t* len;
    int64_t stride;
    int32_t cdf_keys;  // k >= 1: the store holds k SETS of keys, set w = bits of F_w(distance) for weight function w of the configuration (the
                       // sweep needs no CDF evaluation: a pair reads the set of its weight function); 0: keys are the distances
    int32_t cat16;     // 1: `cat` holds 16-bit category ids (more than 255 categories): two bytes per point
    int64_t set_stride;  // elements between two key sets (slots x stride); categories and lengths exist once
    uint64_t* pre;       // prefix-count rows, or null: row j of environment e = pre[(e * stride / kPreStep + j) * pre_words ...] = the category
                         // counts of the environment's first kPreStep * j + 1 sorted points (one row per kPreStep points) as 8-bit fields
                         // (slot c in byte c % 8 of word c / 8), written by k_env_group for configurations of at most 16 categories: the
                         // team sweeps read a chunk's start counts from them -- the row below the chunk's start plus the one-hot fields of
                         // at most kPreStep - 1 category bytes -- instead of building a histogram and a scan per tile (lchd_team_tile.h, PRE)
    int32_t pre_words;   // u64 words per row (1: up to 8 category slots, 2: up to 16, 3: up to 24, 4: up to 32)
    int32_t pad_pre;
    uint8_t* cat0;       // [slots] category of every environment's first (sorted) point, or null: what k_pair_meta puts into the pair records -- one
                         // gather into a small array instead of one into the store itself (k_env_group writes it; the other environment kernels do not)
};
#ifndef LCHD_PRE_STEP
#define LCHD_PRE_STEP 4
#endif
constexpr int kPreStep = LCHD_PRE_STEP;  // points per prefix-count row (round 6: every row of round 5's store cost the sweeps one 128-byte line per lane)
// u64 words of a prefix-count row that the team sweep of `cmax` category slots reads: TeamTile<CM>::NW of the instantiation launch_team
// picks (8, 12 / 16 slots); 0: no instantiation reads rows (k_env_group's writer handles up to four words)
inline int team_pre_words(int cmax) { return cmax <= 8 ? 1 : (cmax <= 16 ? 2 : 0); }
constexpr int kMaxKeySets = 4;  // weight-function dictionaries of up to 4 entries get one key set each (k_env_group); larger ones keep distance keys

// One structure (or batch of structures) of a from_primitives pass as the prologue sees it: the inputs, and the arrays the
// prologue fills (cell list, anchor slots, anchor records).
struct PrepSide {
    CloudView c;
    GridView g;              // geometry; g.cell_start / g.rec / g.pos_of alias the three output arrays below
    uint32_t* cell_start;    // [n_cells + 1]
    CellRec* rec;            // [n] atoms permuted into cell order
    uint32_t* pos_of;        // [n] atom -> position in cell order
    uint32_t *cell_of, *cell_count;  // scratch of the general cell-list build (cell_count [n_cells + 1] inside the zero region)
    uint32_t* scan_tmp;      // (n_cells / 4096 + 4) u32 of scratch for the multi-block scan
    uint8_t* flag8;          // [n rounded up to 32, + 32] one byte per atom: "is an anchor" (inside the zero region, 16-byte aligned)
    uint32_t* bits;          // [(n + 31) / 32 + 1] the same as a bit set (k_prep_scan / k_flags_to_bits)
    uint32_t* wpre;          // [(n + 31) / 32 + 2] anchors before each bit-set word (inside its chunk of 2^18 atoms)
    uint32_t* chunk_base;    // [(n >> 18) + 2] anchors before each chunk
    uint32_t* slot;          // [n + 1] atom -> environment slot (anchors only)
    AnchorRec* uniq;         // [max_envs]
    int32_t no_anchors;      // 1: this side's anchors are neither flagged nor de-duplicated (side B without de-duplication: k_pair_anchor_recs writes an anchor record per PAIR): cell list only
};
// Cell lists of both sides + anchor de-duplication (see lchd_prologue.hip).  [zero_base, zero_base + zero_bytes) is the
// contiguous region holding cell_count of both sides followed by flag8_a, flag8_b (in this order, flag8_b last): the prologue
// zeroes what its launch tier needs with at most one operation.  Returns the number of stream operations enqueued.
// `same`: both sides are one device object -- side B's cell list and slots are not built, the anchors of both columns are
// de-duplicated together into side A's flags / slots / records and n_unique[1] = 0.
int launch_prologue(hipStream_t s, const Tuning& t, const int64_t* anchors, int64_t n_pairs, const PrepSide& a, const PrepSide& b,
                    void* zero_base, size_t zero_bytes, DeviceStatus* st, bool same = false);
// Anchor records of a side without de-duplication (PrepSide::no_anchors), one per PAIR: record p = the side-B anchor of pair p; sets
// DeviceStatus::n_unique[1] = n_pairs.  Behind launch_prologue (reads the side's atom -> position map).
void launch_pair_anchor_recs(hipStream_t s, const int64_t* anchors, int64_t n_pairs, const PrepSide& b, DeviceStatus* st);
// Per-device function attributes (dynamic LDS above 64 KB): called by lchd_ctx_create with the context's device current.
void init_device_kernels();

// returns false if `cap` is not an available variant
// one side of an environment-build launch (both structures are built by ONE launch: workgroups [0, a.max_envs) side A, the rest B)
struct EnvSide {
    CloudView c;
    GridView g;
    const AnchorRec* uniq;
    EnvStore env;
    int64_t max_envs;   // upper bound of the side's unique anchors (the kernel stops at DeviceStatus::n_unique[side])
    double* raw_key;    // environments of more than 16384 points only: [max_envs][cap] unsorted distances ...
    uint8_t* raw_cat;   // ... and categories (k_env_collect -> k_env_rows)
    // Environments that do not fit their slot (k_env_group / k_env_cells): the slot receives the anchor alone (a valid one-point
    // environment, so the sweeps of this pass run cleanly over it) and the slot index is appended here
    // (DeviceStatus::n_overflow[side] counts); the host scores the pairs of these anchors again with larger slots.
    uint32_t* ovf_list;
};
struct EnvSides {
    EnvSide s[2];
};
bool launch_env_cells(hipStream_t s, int cap, const DevConfig* cfg, bool tag_list, const EnvSide& a, const EnvSide& b, double thr,
                      DeviceStatus* st);   // (EnvStore::cat16 of the sides picks the 16-bit-category instantiation: cap <= 16384)
// The default capacity (environments of at most kEnvGroupCap points), several environments per wavefront (lchd_env_group.hip).
// Needs a grid whose cells are at least thr / 2 wide (the search walks the 5 x 5 x 5 neighbourhood), record arrays padded by
// kEnvGroupRecPad records, fewer than 2^29 records per side and environment slots of at least kEnvGroupCap points.
// small_cap: the instantiation for environments of at most kEnvGroupCapSmall points (less LDS, one more wavefront per SIMD).
constexpr int kEnvGroupCap = 512, kEnvGroupCapSmall = 320, kEnvGroupSmallUpTo = 288, kEnvGroupRecPad = 8;
bool launch_env_group(hipStream_t s, const DevConfig* cfg, bool tag_list, bool small_cap, const EnvSide& a, const EnvSide& b, double thr,
                      int anchors_per_wave, DeviceStatus* st);

// dense rows: either distances from coordinates (dmx == nullptr) or given rows (dmx != nullptr, leading dim ld)
struct RowExtras {            // all null for from_coords / from_dmxs
    const uint8_t* row_cat;   // [n_rows][ld] categories of the row's own points (instead of the structure's)
    const int32_t* row_lens;  // [n_rows] points of each row (<= 0: skip the row)
    const uint32_t* n_unique; // rows at or beyond *n_unique do not exist
};
bool launch_env_rows(hipStream_t s, int cap, const DevConfig* cfg, const CloudView& c, const double* dmx, int64_t ld,
                     int64_t n_rows, int64_t row_len, double image_bound /* coords: >= largest squared distance, or 0 */, EnvStore env,
                     DeviceStatus* st, const RowExtras& ex = RowExtras{nullptr, nullptr, nullptr});

// "last workgroup" detection + hand-over through memory-side atomics (lchd_sweep_common.h: last_workgroup_done)
struct DoneState {
    uint32_t ctr[65 * 32];
    uint32_t acc_max[64 * 32];
    unsigned long long acc_sum[64 * 16];
};
// one side of a dense-row launch (lchd_env_rows.hip: k_env_rows2 builds both structures' rows in one launch)
struct RowSide {
    CloudView c;           // categories (and coordinates when dmx == nullptr)
    const double* dmx;     // given distance rows [n_rows][ld], or nullptr: distances from the coordinates of c
    int64_t ld, row_len;
    double image_bound;    // coordinates: >= largest squared distance (bounding-box diagonal^2); ignored for given rows
    EnvStore env;
    const int32_t* row_lens;  // ragged given rows: [n_rows] points of each row (1 .. row_len); nullptr: every row has row_len
};
struct RowSides {
    RowSide s[2];
    int64_t n_rows;
};
// rows of at most 16384 points on both sides; returns false (nothing launched) otherwise
bool launch_env_rows2(hipStream_t s, const DevConfig* cfg, const RowSide& a, const RowSide& b, int64_t n_rows, DeviceStatus* st);

// Dense rows, sort and sweep fused (lchd_dense_fused.hip): one workgroup per row pair, only the score is written.
// Applies to Hellinger-2 with unit category weights, at most 16 categories and rows of kDenseFusedMinRow < n <= kDenseFusedMaxRow points.
constexpr int kDenseFusedMinRow = 1024, kDenseFusedMaxRow = 20480;
struct DenseSide {
    CloudView c;              // categories (and coordinates when dmx == nullptr)
    const double* dmx;        // given distance rows [n_rows][ld], or nullptr
    int64_t ld;
    int32_t row_len;          // points per row
    const int32_t* row_lens;  // ragged given rows: [n_rows] points of each row (1 .. row_len), or nullptr
};
struct DenseArgs {
    const DevConfig* cfg;
    DenseSide s[2];
    int64_t n_rows;
    double image_bound;       // coordinates: >= the largest squared distance of either structure; ignored for given rows
    const int32_t* wf_index;  // [n_rows] or nullptr => 0
    double* out;
    DeviceStatus* st;
    const double* sqrt_tab;   // [65536] sqrt(k), context-owned
    // scratch of the distance pass: workgroup w keeps the (key, value) pairs of distance segment s of its CURRENT row pair at
    // [(w * scr_segs + s) * dense_fused_seg_cap() ...); dense_fused_scratch() sizes both arrays and sets scr_segs / scr_grid
    uint64_t* scr_key;
    uint8_t* scr_val;
    int32_t scr_segs, scr_grid;
    unsigned long long* ticket;  // context-owned: rows handed out beyond the first one of every workgroup (zero between launches)
};
// workgroups of a launch and segments per workgroup for rows of up to (len_a, len_b) points; returns the ENTRIES of either scratch array
size_t dense_fused_scratch(int64_t n_rows, int64_t len_a, int64_t len_b, int32_t* grid_out, int32_t* segs_out);
bool dense_fused_applies(int n_categories, int64_t len_a, int64_t len_b);
// returns false (nothing launched) when the kernel does not apply; otherwise the kernel and the status hand-over are enqueued
bool launch_dense_fused(hipStream_t s, int n_categories, const DenseArgs& a, HostStatus* hst, uint32_t seq);
void init_dense_fused_kernels();  // per device (dynamic LDS above 64 KB), called by lchd_ctx_create

struct SweepArgs {
    const DevConfig* cfg;
    EnvStore env_a, env_b;
    const int64_t* anchors;   // [P][2] or nullptr => pair p uses env (p, p)
    const uint32_t* slot_a;   // anchor index -> env slot (nullptr with anchors == nullptr)
    const uint32_t* slot_b;   // (nullptr with anchors != nullptr: side B's slot of pair p is p -- side B without de-duplication)
    int64_t n_slot_a, n_slot_b;  // atoms per side (bounds for the anchor indices)
    const int32_t* wf_index;  // [P] or nullptr => 0
    int64_t n_pairs;
    double* out;
    DeviceStatus* st;
    const double* sqrt_tab;   // [65536] sqrt(k), context-owned
    const double* rsqrt_tab;  // [65536] 1/sqrt(k)
    int4* meta;               // [P] workspace: per-pair records written by k_pair_meta, read by the sweep kernels
    DoneState* done;          // context-owned, zero between kernels: "last workgroup" counters and accumulators
    unsigned char* wide_scratch;      // more than kWideCategories categories (k_sweep_wide<.., HUGE>): context-owned global-memory block,
    int64_t wide_scratch_per_wave;    //   bytes per workgroup (wide_scratch_bytes_per_wave(C)),
    int32_t wide_scratch_waves;       //   workgroups it serves (= the grid of that launch)
    HostStatus* hst;          // host-mapped mirror (snapshot by k_pair_meta, error words by the sweep kernels)
    uint32_t seq;             // pass counter echoed into HostStatus::snapshot_seq
    int32_t duo_enabled;      // set by launch_sweep: k_sweep_duo was launched too and sweeps the small pairs when they are the majority
    int32_t small_rule;       // set by launch_sweep: which pairs count as "small" (0: <= kDuoTile merged events, k_sweep_duo; 1: both
                              // environments <= 255 points, the 8-bit-count k_sweep; 2: ... and at most 480 merged events, its
                              // two-pairs-per-wavefront form); the indirect k_sweep takes the others
    int32_t c8_rule;          // set by launch_sweep: the rule (1 or 2) HostStatus::n_c8 is counted under
    int32_t second_rule;      // set by launch_sweep: 0, or the rule of the second small-pair kernel a pass without a hint launched
    int32_t gen_tab;          // set by launch_sweep: MODE_GEN may use power tables (Hellinger with a general exponent, unit category weights)
    int32_t forced;           // set by launch_sweep: the host picked the sweep kernels (hint from the previous pass): no device-side decision
    int32_t sd_fast;          // the configuration qualifies for k_sweep_inc (lchd_sweep_inc.hip): 0 no, 1 Kullback-Leibler form, 2 Renyi form
    // Leftover list (round 6): in a pass whose sweep kernels the host picked (forced: the rule is known at launch) k_pair_meta appends
    // the pairs the team kernel's rule leaves over to left_list (wave-aggregated, a handful of atomics per launch) and the INDIRECT
    // companion walks that list instead of scanning every pair record for them (C2a: 223 of 10^6 pairs; 16.6 -> ~3 us per pass).
    // Two counter slots take turns: the pass that appends to one zeroes the other for the next pass -- no memset, no reset kernel.
    uint32_t* left_list;      // [P] workspace (nullptr: no list, the companion scans the records)
    uint32_t* left_count;     // this pass's counter slot (zero when the pass starts)
    uint32_t* left_zero;      // the other slot: zeroed by k_pair_meta of this pass
    int32_t left_listing;     // set by launch_sweep: k_pair_meta appends and the companion reads the list
    int64_t left_expected;    // pairs the previous pass of the context left over (sizes the companion's grid; any grid is correct)
};
// sweep_hint: 0 = unknown (launch every candidate kernel, the device decides from the pair records); otherwise what
// k_pair_meta counted in the previous pass of this configuration: 4 | 1 (pairs of at most 240 merged events were the
// majority: k_sweep_duo + the indirect k_sweep) | 2 (pairs with both environments <= 255 points were: the 8-bit-count k_sweep
// + the indirect one); neither: the plain k_sweep only.  Any choice is correct for any input; the hint only picks the launch set.
// ... | 8 (every pair of the previous pass had at most 240 events) | 16 (... both environments <= 255 points): the companion
// launch for the larger pairs is left out.  Returns 1 (the "small" rule of this pass was the 8-bit-count one) | 2 (the companion
// launch was left out: the caller must check this pass's counts, HostStatus::n_duo / n_c8 against the number of pairs) | 4 (k_pair_meta
// ran with the leftover-list counters of `a`: the caller swaps the two counter slots for the next pass).
int launch_sweep(hipStream_t s, const Tuning& t, int n_categories, bool hellinger2, bool unit_weights, bool wf_pow, int sweep_hint,
                 const SweepArgs& a);
// Kullback-Leibler / Renyi in O(1) per event (lchd_sweep_inc.hip): unit weights, CDF-keyed environments of at most 512 points, tiny eps;
// reads the pair records of k_pair_meta.  kind: SweepArgs::sd_fast.
void launch_sweep_inc(hipStream_t s, int kind, int cmax, const SweepArgs& a);
// trajectory frames: replicate the template's labels / unpack [frames][atoms][3] into SoA + bounding box keys
void launch_frames_labels(hipStream_t s, const uint8_t* tcat, const int32_t* ttag, int64_t n_tmpl, int32_t n_frames, uint8_t* cat,
                          int32_t* tag, int32_t* sid);
void launch_frames_unpack(hipStream_t s, const double* raw, int64_t n_atoms, double* x, double* y, double* z,
                          unsigned long long* bbox7);
// primitive atoms of every frame = float32 centroids of their source atoms (CSR src_start / src_idx), widened to f64
// tiles: [n_tiles][4] = {first primitive, end primitive, first source atom, end source atom} with a span of at most
// centroid_tile_span() source atoms, or {p0, p1, -1, -1} for a primitive range that is gathered from global memory
void launch_frames_centroids(hipStream_t s, const float* raw, int64_t n_src, const int32_t* src_start, const int32_t* src_idx,
                             const int32_t* tiles, int n_tiles, int64_t n_prim, int32_t n_frames, double* x, double* y, double* z,
                             unsigned long long* bbox7, unsigned long long* bbox_part /* [bbox_parts_capacity()][7] */);
int centroid_tile_span();
int bbox_parts_capacity();
void launch_fill_sqrt_tables(hipStream_t s, double* sqrt_tab, double* rsqrt_tab);  // 65536 entries each
void launch_fill_pow_tables(hipStream_t s, double* tab /* [2][65536] */, double exponent);  // k^(1/e), k^(-1/e)

// Multi-GPU sharding of an anchor-pair list by anchor bins (see lchd_kernels.hip).  ShardState lives in device memory,
// zero-initialised once; hist / hist_b / cursor / done are zero again after every plan.
constexpr int kShardBins = 1024, kShardMaxWorld = 64;
struct ShardState {
    uint32_t hist[kShardBins], hist_b[kShardBins];
    uint16_t rank_of_bin[kShardBins];
    int32_t mode, pad;   // key side of the last plan: 0 side-A anchor, 1 side-B anchor, 2 pair index
    int64_t counts[kShardMaxWorld];
    unsigned long long cursor;
    DoneState done;
};
struct ShardCounts {
    int64_t n[kShardMaxWorld];
};
void launch_shard_plan(hipStream_t s, const int64_t* anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b /* <= 0: unknown */,
                       int world, ShardState* st, int64_t* counts_host /* host-mapped [kShardMaxWorld + 1]: counts, then the key side */);
void launch_shard_select(hipStream_t s, const int64_t* anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int rank, int world,
                         ShardState* st, int64_t* sel_anchors, int64_t* sel_index);
// The pairs of a finished pass that touch an overflowed environment (EnvSide::ovf_list), in list order: lchd_ctx_finish scores
// them again with larger slots.  wave_count: kOverflowWaves words of scratch.
constexpr int kOverflowWaves = 4096;
struct OverflowSelect {
    const int64_t* anchors;
    int64_t n_pairs;
    const uint32_t *slot_a, *slot_b;   // atom -> environment slot of the finished pass
    const uint32_t *bits_a, *bits_b;   // bit sets over the slots (launch_mark_overflow)
    const int32_t* wf;                 // per-pair weight-function index or null
    unsigned long long* wave_count;
    int64_t *sel_index, *sel_anchors;  // [n], [n][2]
    int32_t* sel_wf;
};
void launch_env_key_sets(hipStream_t s, const DevConfig* cfg, const EnvStore& ea, const EnvStore& eb, int n_sets, int64_t max_envs, const DeviceStatus* st);
// deterministic mode: the categories of every run of equal keys in ascending order (positions >= 1), one wavefront per environment;
// st != nullptr: the environments in use are DeviceStatus::n_unique (n_a / n_b bound the launch), otherwise exactly n_a / n_b
void launch_env_canon(hipStream_t s, const EnvStore& ea, const EnvStore& eb, int64_t n_a, int64_t n_b, const DeviceStatus* st);
void launch_mark_overflow(hipStream_t s, const uint32_t* list_a, uint32_t na, const uint32_t* list_b, uint32_t nb, uint32_t* bits_a, uint32_t* bits_b);
void launch_count_overflow(hipStream_t s, const OverflowSelect& a, unsigned long long* total);
void launch_write_overflow(hipStream_t s, const OverflowSelect& a);
void launch_scatter_scores(hipStream_t s, const double* scores, const int64_t* index, int64_t n, double* out);
void launch_unshard_scores(hipStream_t s, const double* gathered, const ShardCounts& counts, int world, int64_t stride, double* out,
                           int64_t n_pairs, uint32_t* bad);
void launch_env_points(hipStream_t s, const SweepArgs& a, unsigned long long* out);
// from_anchors on lists whose distances do not ascend: the reference's loop walked literally by one lane (lchd_sweep.hip)
void launch_anchors_literal(hipStream_t s, const DevConfig* cfg, int n_categories, const EnvStore& ea, const EnvStore& eb, int nA, int nB, int wfi,
                            double* out);

}  // namespace lchd

// lchd_device.h -- internal interface between the C-ABI host layer (lchd_capi.hip) and the gfx950
// kernels (lchd_kernels.hip).  Plain-old-data argument blocks + launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lchd {

constexpr int kMaxCategories = 255;   // categories travel as u8 on the device
constexpr int kSweepEPL = 6;          // merged events per lane per tile in the sweep kernel
constexpr int kSweepTile = 64 * kSweepEPL;
constexpr int kMetaPartials = 4096;   // workgroups of k_pair_meta (one partial count each)
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
};
struct DeviceStatus {
    uint32_t flags;
    uint32_t max_env;        // largest environment seen (for the overflow retry)
    uint32_t n_unique[2];    // unique anchors per side
    unsigned long long env_points;  // sum over pairs of n_A + n_B
    unsigned long long n_small;     // pairs with at most kDuoTile merged events (k_pair_meta): who sweeps them is decided on the device
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
    const uint64_t* tag_pairs;  // sorted, (anchor_tag << 32) | neighbour_tag
};

// SoA structure in HBM.
struct CloudView {
    const double *x, *y, *z;
    const uint8_t* cat;
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
    int32_t cdf_keys;  // 1: key = bits of F(distance) for the configuration's single weight function (sweep needs no CDF evaluation)
};

// `scan_tmp` holds (n / 4096 + 2) u32 of scratch for the multi-block scan
void launch_cell_build(hipStream_t s, const CloudView& c, GridView g, uint32_t* cell_of, uint32_t* cell_count,
                       uint32_t* cell_cursor, CellRec* rec, uint32_t* pos_of, uint32_t* cell_start, uint32_t* scan_tmp);

void launch_anchor_dedupe(hipStream_t s, const int64_t* anchors, int64_t n_pairs, int side, int32_t n_points,
                          uint32_t* flag_then_slot, AnchorRec* uniq, const CloudView& c, const uint32_t* pos_of, DeviceStatus* st,
                          uint32_t* scan_tmp);

// both sides in one launch when the inputs are small; returns false (nothing launched) otherwise
bool launch_anchor_dedupe_small(hipStream_t s, const int64_t* anchors, int64_t n_pairs, const CloudView& ca, const CloudView& cb,
                                const uint32_t* pos_a, const uint32_t* pos_b, uint32_t* slot_a, uint32_t* slot_b, AnchorRec* uniq_a,
                                AnchorRec* uniq_b, DeviceStatus* st);

// returns false if `cap` is not an available variant
// one side of an environment-build launch (both structures are built by ONE launch: workgroups [0, a.max_envs) side A, the rest B)
struct EnvSide {
    CloudView c;
    GridView g;
    const AnchorRec* uniq;
    EnvStore env;
    int64_t max_envs;   // upper bound of the side's unique anchors (the kernel stops at DeviceStatus::n_unique[side])
};
struct EnvSides {
    EnvSide s[2];
};
bool launch_env_cells(hipStream_t s, int cap, const DevConfig* cfg, bool tag_list, const EnvSide& a, const EnvSide& b, double thr,
                      DeviceStatus* st);

// dense rows: either distances from coordinates (dmx == nullptr) or given rows (dmx != nullptr, leading dim ld)
bool launch_env_rows(hipStream_t s, int cap, const DevConfig* cfg, const CloudView& c, const double* dmx, int64_t ld,
                     int64_t n_rows, int64_t row_len, double image_bound /* coords: >= largest squared distance, or 0 */, EnvStore env,
                     DeviceStatus* st);

struct SweepArgs {
    const DevConfig* cfg;
    EnvStore env_a, env_b;
    const int64_t* anchors;   // [P][2] or nullptr => pair p uses env (p, p)
    const uint32_t* slot_a;   // anchor index -> env slot (nullptr with anchors == nullptr)
    const uint32_t* slot_b;
    int64_t n_slot_a, n_slot_b;  // atoms per side (bounds for the anchor indices)
    const int32_t* wf_index;  // [P] or nullptr => 0
    int64_t n_pairs;
    double* out;
    DeviceStatus* st;
    const double* sqrt_tab;   // [65536] sqrt(k), context-owned
    const double* rsqrt_tab;  // [65536] 1/sqrt(k)
    int4* meta;               // [P] workspace: per-pair records written by k_pair_meta, read by the sweep kernels
    uint32_t* partials;       // [kMetaPartials] context-owned scratch of k_pair_meta
    int32_t duo_enabled;      // set by launch_sweep: k_sweep_duo was launched too and sweeps the small pairs when they are the majority
};
void launch_sweep(hipStream_t s, int n_categories, bool hellinger2, bool unit_weights, bool wf_pow, const SweepArgs& a);
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
void launch_env_points(hipStream_t s, const SweepArgs& a, unsigned long long* out);

}  // namespace lchd

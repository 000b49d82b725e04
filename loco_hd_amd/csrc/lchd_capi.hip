// lchd_capi.hip -- host side of the C ABI declared in include/loco_hd_hip.h.
//
// Owns: validation (the reference's PyValueError sites), packing of caller buffers into SoA device
// arrays, the device workspace, kernel sequencing on one HIP stream, the status read-back and its
// translation into the reference's error classes.  No scoring arithmetic happens on the host: every
// from_* entry point launches the gfx950 kernels of lchd_*.hip and fails with LCHD_EDEVICE when
// no GPU is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/loco_hd_hip.h"
#include "lchd_device.h"
#include "lchd_math.h"

using namespace lchd;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* lchd_last_error(void) { return g_err; }
extern "C" const char* lchd_version(void) { return "loco_hd_hip 0.1 (gfx950)"; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(LCHD_EDEVICE, "HIP error %d (%s) at %s:%d: %s", (int)e_,      \
                                          hipGetErrorName(e_), __FILE__, __LINE__, #expr);              \
    } while (0)

// ------------------------------------------------------------------------------------------------
// host-side leaves
// ------------------------------------------------------------------------------------------------
static const char* wf_name(int kind) {
    static const char* names[4] = {"hyper_exp", "dagum", "uniform", "kumaraswamy"};
    return (kind >= 0 && kind < 4) ? names[kind] : "?";
}

extern "C" int lchd_wf_validate(int32_t kind, const double* p, int32_t np) {  // weight_function.rs:22-93
    const char* nm = wf_name(kind);
    auto exactly = [&](int n) { return np == n ? 0 : fail(LCHD_EVALUE, "For function \"%s\" there must be exactly %d parameters!", nm, n); };
    switch (kind) {
        case LCHD_WF_HYPER_EXP:
            if (np % 2 != 0) return fail(LCHD_EVALUE, "For function \"%s\" there must be an even number of parameters!", nm);
            for (int i = 0; i < np; ++i)
                if (p[i] <= 0.0) return fail(LCHD_EVALUE, "For function \"%s\" all parameters must be positive!", nm);
            return LCHD_OK;
        case LCHD_WF_DAGUM:
            if (int rc = exactly(3)) return rc;
            if (p[0] < 0.0 || p[1] < 0.0 || p[2] < 0.0) return fail(LCHD_EVALUE, "For function \"%s\" all parameters must be positive!", nm);
            return LCHD_OK;
        case LCHD_WF_UNIFORM:
            if (int rc = exactly(2)) return rc;
            if (p[0] < 0.0) return fail(LCHD_EVALUE, "For function \"%s\" the first parameter must be non-negative!", nm);
            if (p[1] <= 0.0) return fail(LCHD_EVALUE, "For function \"%s\" the second parameter must be positive!", nm);
            if (p[0] >= p[1]) return fail(LCHD_EVALUE, "For function \"%s\" the first parameter must be smaller than the second!", nm);
            return LCHD_OK;
        case LCHD_WF_KUMARASWAMY:
            if (int rc = exactly(4)) return rc;
            if (p[0] < 0.0) return fail(LCHD_EVALUE, "For function \"%s\" the first parameter must be non-negative!", nm);
            if (p[1] <= 0.0 || p[2] <= 0.0 || p[3] <= 0.0)
                return fail(LCHD_EVALUE, "For function \"%s\" after the first parameter all parameters must be positive!", nm);
            if (p[0] >= p[1]) return fail(LCHD_EVALUE, "For function \"%s\" the first parameter must be smaller than the second!", nm);
            return LCHD_OK;
        default: return fail(LCHD_EVALUE, "No function implemented with this name!");
    }
}

extern "C" int lchd_wf_cdf(int32_t kind, const double* p, int32_t np, const double* x, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) {  // weight_function.rs:95-116
        if (x[i] < 0.0) return fail(LCHD_EVALUE, "Invalid input value: %g. All values must be non-negative!", x[i]);
        out[i] = cdf_eval(kind, p, np, x[i]);
    }
    return LCHD_OK;
}

extern "C" int lchd_sd_validate(int32_t kind, int32_t np) {  // statistical_distances.rs:96-121
    static const int want[4] = {1, 0, 1, 2};
    static const char* names[4] = {"Hellinger", "Kolmogorov-Smirnov", "Kullback-Leibler", "Renyi"};
    if (kind < 0 || kind > 3) return fail(LCHD_EVALUE, "Invalid statistical distance name!");
    if (np != want[kind]) return fail(LCHD_EVALUE, "Invalid number of parameters for %s: %d", names[kind], np);
    return LCHD_OK;
}

extern "C" int lchd_sd_run(int32_t kind, const double* prm, const double* p1, const double* p2, int32_t n, double* out) {
    if (kind < 0 || kind > 3) return fail(LCHD_EVALUE, "Invalid statistical distance name!");
    if (n <= 0 && kind == LCHD_SD_KOLMOGOROV_SMIRNOV) return fail(LCHD_EPANIC, "called `Option::unwrap()` on a `None` value");
    const double a = (kind == LCHD_SD_KOLMOGOROV_SMIRNOV) ? 0.0 : prm[0], b = (kind == LCHD_SD_RENYI) ? prm[1] : 0.0;
    // Hellinger with a runtime exponent: the reference calls powf even for e == 2 (statistical_distances.rs:5-9)
    if (kind == LCHD_SD_HELLINGER)
        *out = sd_hellinger<0>([&](int c) { return p1[c]; }, [&](int c) { return p2[c]; }, n, a);
    else
        *out = sd_eval<0>(kind, a, b, [&](int c) { return p1[c]; }, [&](int c) { return p2[c]; }, n);
    return LCHD_OK;
}

extern "C" int lchd_config_validate(int64_t n_given, int64_t n_map, const double* w, int64_t nw) {  // src/locohd.rs:305-346
    if (n_given == 0) return fail(LCHD_EVALUE, "The number of possible categories (primitive types) cannot be zero!");
    if (nw != n_map)
        return fail(LCHD_EVALUE, "LoCoHD parameters 'categories' and 'category_weights' must have the same lengths! "
                                 "Instead, they have lengths of %lld vs. %lld!", (long long)n_map, (long long)nw);
    long long bad = 0;
    for (int64_t i = 0; i < nw; ++i) bad += (w[i] <= 0.0);
    if (bad) return fail(LCHD_EVALUE, "LoCoHD parameter 'category_weights' must only contain positive values! "
                                      "Instead, it contains %lld non-positive values!", bad);
    return LCHD_OK;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct lchd_cloud {
    double *x = nullptr, *y = nullptr, *z = nullptr;
    uint8_t* cat = nullptr;
    uint8_t* cat_hi = nullptr;  // high byte of the category ids (allocated when an id beyond 254 occurs: more than 255 categories)
    uint8_t* cat_narrow = nullptr;  // with cat_hi: the one-byte view for configurations of at most 255 categories (ids beyond 254 -> 255,
                                    // "not in the category map": the narrow kernels would otherwise score id 256 as category 0)
    int32_t* tag = nullptr;
    int32_t* sid = nullptr;  // batch of structures: structure id per atom (nullptr = one structure)
    int32_t n_struct = 1;
    int32_t struct_size = 0;  // > 0: structure k is atoms [k * struct_size, (k + 1) * struct_size)
    int64_t n = 0;
    // trajectory-frames buffer (lchd_frames_create): capacity, staging and cross-stream hand-off
    int64_t n_tmpl = 0;        // atoms per frame
    int32_t cap_frames = 0;    // frames the arrays can hold
    double* d_raw = nullptr;   // device staging of the caller's [frames][atoms][3] block
    double* h_pinned = nullptr;
    // frames given as float32 SOURCE atoms (lchd_frames_set_sources): CSR map primitive atom -> source atoms
    int32_t *d_src_start = nullptr, *d_src_idx = nullptr, *d_tiles = nullptr;
    int32_t n_tiles = 0;
    int64_t n_src = 0;
    float *d_raw32 = nullptr, *h_pinned32 = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // timing of the most recent load (only when the context's timing is on)
    bool t_valid = false;
    unsigned long long* d_bbox = nullptr;  // [7]: order-preserving keys of min xyz, max xyz, non-finite flag
    unsigned long long* d_bbox_part = nullptr;  // per-workgroup partials of the centroid kernel
    hipEvent_t ev_ready = nullptr, ev_used = nullptr;
    bool bbox_pending = false, used_valid = false;
    double bbmin[3] = {0, 0, 0}, bbmax[3] = {0, 0, 0};
    // wide: the configuration has more than 255 categories (two-byte ids where the structure carries them)
    CloudView view(bool wide) const {
        const bool two = wide && cat_hi;
        return CloudView{x, y, z, (!wide && cat_narrow) ? cat_narrow : cat, two ? cat_hi : nullptr, tag, (int32_t)n, sid, n_struct, sid ? struct_size : 0};
    }
};

enum { PH_CELLS = 0, PH_ANCHORS = 1, PH_ENV = 2, PH_SWEEP = 3, PH_N = 4 };

struct lchd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Tuning tune{};  // LCHD_* test / tuning hooks, read once in lchd_ctx_create
    Tuning tune_env{};  // ... as read from the environment (lchd_ctx_set_deterministic(0) returns to them)
    bool deterministic = false;
    // configuration
    bool cfg_set = false;
    bool hellinger2 = false, unit_weights = false, wf_pow = false;  // which sweep kernel variant applies
    bool finf_differ = false;  // the weight functions of a dictionary do not share F(+inf) (degenerate parameters): no key sets
    int sd_fast = 0;  // Kullback-Leibler (1) / Renyi (2) with parameters the O(1)-per-event sweep handles (lchd_sweep_inc.hip)
    DevConfig h_cfg{};
    DevConfig* d_cfg = nullptr;
    char* d_blob = nullptr;
    size_t blob_cap = 0;
    // workspace arena
    char* ws = nullptr;
    size_t ws_cap = 0;
    DeviceStatus* d_status = nullptr;  // clean (all zero) between passes: k_pair_meta's last workgroup resets it
    HostStatus* h_status = nullptr;    // pinned, device-visible: the kernels publish into it with plain stores (no D2H copy)
    bool status_dirty = false;         // a pass was abandoned half-way: memset d_status before the next one
    uint32_t seq = 0;                  // pass counter (HostStatus::snapshot_seq)
    int sweep_hint = 0;                // 0 unknown, else 4 | 1 (pairs of <= 240 events were the majority of the last pass) | 2 (pairs with both environments <= 255 points were): launch_sweep
    unsigned long long* d_points = nullptr;
    double* d_tabs = nullptr;  // sqrt(k) | 1/sqrt(k), 65536 entries each
    unsigned char* d_wide_scratch = nullptr;  // more than kWideCategories categories: per-workgroup state of k_sweep_wide<.., HUGE> (lchd_ctx_set_config)
    size_t wide_scratch_cap = 0;
    int wide_scratch_waves = 0;
    double* d_powtab = nullptr;  // k^(1/e) | k^(-1/e) for the configured Hellinger exponent (allocated when one is first configured)
    double powtab_e = 0.0;       // the exponent the table holds
    DoneState* d_done = nullptr;     // 'last workgroup' counters / accumulators of k_pair_meta (zero between kernels)
    uint32_t* d_left = nullptr;      // two counter slots of the leftover list (SweepArgs::left_count / left_zero), 128 B apart
    int left_slot = 0;               // the slot the next record pass appends to (zero by then: the previous one zeroed it)
    int64_t last_left = 0;           // pairs the last pass left to the INDIRECT companion
    // host-pointer calls: one grow-only device block + pinned staging block per context (no allocation in the steady state)
    char *d_io = nullptr, *h_io = nullptr;
    size_t io_cap = 0;
    std::vector<char> cfg_blob_host;  // last configuration blob uploaded (identical configurations are not uploaded again)
    int cap_hint = 512;
    bool group_small = false;  // the last pass had no environment beyond kEnvGroupSmallUpTo points: k_env_group's small instantiation
    int64_t last_biggest = 0;  // largest environment of the last pass (0: unknown): anchors per wavefront of k_env_group
    int shrink_votes = 0;  // consecutive passes whose largest environment would fit half of cap_hint
    bool last_dense_fused = false;  // the most recent dense pass ran the fused sort + sweep kernel (lchd_dense_fused.hip)
    // side B without de-duplication (one environment slot per PAIR) for side-B environments that are used once: taken when the last
    // REGULAR pass of the configuration found (almost) every side-B anchor unique; every 64th pass is a regular one again
    // (such a pass does not count side B's unique anchors, so it cannot see the anchors becoming shared)
    bool b_use_once = false;        // the last regular pass: n_unique[1] >= 0.8 n_pairs
    int64_t use_once_pairs = 0, use_once_nb = 0;  // ... its pair count and the size of its side B (the hint holds for lists like it)
    int per_pair_streak = 0;        // passes without side-B de-duplication since the last regular one
    int64_t n_per_pair_passes = 0;  // passes whose side B was not de-duplicated
    // second pass over the pairs of overflowed environments (lchd_ctx_finish): grow-only device blocks outside the arena
    char *d_ovf_bits = nullptr, *d_ovf_lists = nullptr;
    size_t ovf_bits_cap = 0, ovf_lists_cap = 0;
    int64_t n_subset_passes = 0;    // second passes run so far
    size_t last_store_bytes = 0;    // environment-store bytes (keys + categories, both sides) of the passes of the last from_primitives call
    int64_t n_passes = 0;           // from_primitives passes enqueued so far (a call is one pass unless a capacity / launch-set retry repeats it)
    // timing
    bool timing = false;
    hipEvent_t ev[PH_N + 1] = {};
    float ms[PH_N] = {-1, -1, -1, -1};
    // a from_primitives call that has been enqueued but not finished (lchd_from_primitives_dev_async)
    struct {
        bool active = false;
        lchd_cloud *a = nullptr, *b = nullptr;
        const int64_t* anchors = nullptr;
        const int32_t* wf = nullptr;
        int64_t n_pairs = 0;
        double thr = 0.0;
        double* out = nullptr;
        int cap = 0;
        bool group = false, group_small = false;  // which environment kernel the enqueued pass uses
        int sweep_info = 0;                        // launch_sweep's return value
        SweepArgs sw{};
        const uint32_t *ovf_a = nullptr, *ovf_b = nullptr;  // overflow lists of the enqueued pass (null: its kernels keep none)
        int64_t n_slots_a = 0, n_slots_b = 0;               // environment slots per side
        bool subset = false;                                // the enqueued pass IS a second pass over the pairs of overflowed environments
        bool per_pair = false;                              // the enqueued pass did not de-duplicate side B (slot p = pair p)
    } pend;
    // multi-GPU sharding helpers (lchd_shard_*): device state, host-mapped counts, the plan they belong to
    ShardState* d_shard = nullptr;
    hipStream_t shard_stream = nullptr;  // the plan kernel runs (and is waited for) here: the wait does not include the scoring passes
    hipEvent_t shard_ev = nullptr;       // ... and the selection on the context's stream is ordered behind it
    hipEvent_t shard_sel_ev = nullptr;   // the last selection (reads the plan's bin table): the next plan is ordered behind it
    bool shard_sel_pending = false;
    int64_t* h_counts = nullptr;
    uint32_t* d_bad = nullptr;
    int shard_world = 0;
    int64_t shard_pairs = 0, shard_atoms = 0, shard_atoms_b = 0;
    // most recent sweep (for lchd_ctx_last_env_points)
    SweepArgs last{};
    bool last_valid = false;
};

struct Arena {
    char* base;
    size_t off = 0, cap;
    bool dry;
    Arena(char* b, size_t c, bool d) : base(b), cap(c), dry(d) {}
    template <class T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
};

// HIP's current device is per-thread state that torch, another context or another library may change between two calls:
// every entry point that takes a context makes the context's device current for its duration and restores the caller's.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {  // device < 0: nothing to do
        if (device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define CTX_GUARD(c) DeviceGuard device_guard_((c)->device)

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static Tuning tuning_from_env() {  // the ONLY place that reads LCHD_* hooks (tests and tuning runs; unset in production)
    Tuning t;
    t.no_struct_cells = getenv("LCHD_NO_STRUCT_CELLS") != nullptr;
    t.no_share = getenv("LCHD_NO_SHARED_ENVS") != nullptr;
    t.no_cdf_keys = getenv("LCHD_NO_CDF_KEYS") != nullptr;
    t.no_key_sets = getenv("LCHD_NO_KEY_SETS") != nullptr;
    t.no_duo = getenv("LCHD_NO_DUO") != nullptr;
    t.force_wide = env_int("LCHD_FORCE_WIDE", 0) != 0;
    t.force_generic = env_int("LCHD_FORCE_GENERIC", 0) != 0;
    t.force_bigenv = env_int("LCHD_FORCE_BIGENV", 0) != 0;
    t.no_inline_meta = getenv("LCHD_NO_INLINE_META") != nullptr;
    t.no_count8 = getenv("LCHD_NO_COUNT8") != nullptr;
    t.no_c8_team = getenv("LCHD_NO_C8_TEAM") != nullptr;
    t.old_rows = getenv("LCHD_OLD_ROWS") != nullptr;
    t.no_dense_fused = getenv("LCHD_NO_DENSE_FUSED") != nullptr;
    t.no_env_group = getenv("LCHD_NO_ENV_GROUP") != nullptr;
    t.no_overflow_subset = getenv("LCHD_NO_OVERFLOW_SUBSET") != nullptr;
    t.env_apw = env_int("LCHD_ENV_APW", 0);
    t.no_sd_inc = getenv("LCHD_NO_SD_INC") != nullptr;
    t.force_cmax = env_int("LCHD_FORCE_CMAX", 0);
    t.per_pair = env_int("LCHD_PER_PAIR", 0);
    t.pre_rows = env_int("LCHD_PRE_ROWS", 0);
    return t;
}

static int ensure_ws(lchd_ctx* ctx, size_t need) {
    if (need <= ctx->ws_cap) return LCHD_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->ws) HIP_TRY(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_cap = 0;
    ctx->last_valid = false;
    size_t got = need + need / 8 + (1 << 20);
    hipError_t e = hipMalloc(&ctx->ws, got);
    if (e != hipSuccess) {  // without the growth margin
        (void)hipGetLastError();
        got = need;
        e = hipMalloc(&ctx->ws, got);
    }
    if (e != hipSuccess) {
        ctx->ws = nullptr;
        (void)hipGetLastError();
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        return fail(LCHD_EUNSUPPORTED, "this call needs a device workspace of %zu bytes (environment store included) but only %zu of %zu "
                                       "bytes are free on the device", need, free_b, total_b);
    }
    ctx->ws_cap = got;
    return LCHD_OK;
}

extern "C" int lchd_ctx_create(int32_t device, lchd_ctx** out) {
    if (!out) return fail(LCHD_EVALUE, "null output pointer");
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(LCHD_EDEVICE, "no usable HIP device (hipGetDeviceCount -> %d, %d devices): the LoCoHD scoring path has no "
                                  "CPU fallback", (int)e, n_dev);
    if (device >= n_dev) return fail(LCHD_EDEVICE, "device %d requested, %d HIP devices visible", device, n_dev);
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    lchd_ctx* c = new lchd_ctx();
    c->device = device >= 0 ? device : cur;
    c->tune = c->tune_env = tuning_from_env();
    CTX_GUARD(c);  // the caller's current device is restored on every path out of here
    auto bail = [&](hipError_t err, const char* what) {
        lchd_ctx_destroy(c);
        return fail(LCHD_EDEVICE, "HIP error %d (%s) in lchd_ctx_create: %s", (int)err, hipGetErrorName(err), what);
    };
    if ((e = hipMalloc(&c->d_cfg, sizeof(DevConfig))) != hipSuccess) return bail(e, "hipMalloc(config)");
    if ((e = hipMalloc(&c->d_status, sizeof(DeviceStatus))) != hipSuccess) return bail(e, "hipMalloc(status)");
    if ((e = hipMemset(c->d_status, 0, sizeof(DeviceStatus))) != hipSuccess) return bail(e, "hipMemset(status)");
    if ((e = hipMalloc(&c->d_points, sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(points)");
    if ((e = hipHostMalloc(&c->h_status, sizeof(HostStatus))) != hipSuccess) return bail(e, "hipHostMalloc(status)");
    memset(c->h_status, 0, sizeof(HostStatus));
    if ((e = hipMalloc(&c->d_tabs, sizeof(double) * 2 * 65536)) != hipSuccess) return bail(e, "hipMalloc(tables)");
    // (+ one cache line behind it: k_dense_fused's row counter, zero between launches -- k_dense_publish resets it)
    if ((e = hipMalloc(&c->d_done, sizeof(DoneState) + 128)) != hipSuccess) return bail(e, "hipMalloc(done)");
    if ((e = hipMemset(c->d_done, 0, sizeof(DoneState) + 128)) != hipSuccess) return bail(e, "hipMemset(done)");
    if ((e = hipMalloc(&c->d_left, 256)) != hipSuccess) return bail(e, "hipMalloc(leftover counters)");
    if ((e = hipMemset(c->d_left, 0, 256)) != hipSuccess) return bail(e, "hipMemset(leftover counters)");
    init_device_kernels();  // per device, not per process
    init_dense_fused_kernels();
    launch_fill_sqrt_tables(c->stream, c->d_tabs, c->d_tabs + 65536);
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "table fill");
    for (auto& ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
    *out = c;
    return LCHD_OK;
}

extern "C" void lchd_ctx_destroy(lchd_ctx* c) {
    if (!c) return;
    CTX_GUARD(c);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->ws);
    (void)hipFree(c->d_blob);
    (void)hipFree(c->d_cfg);
    (void)hipFree(c->d_status);
    (void)hipFree(c->d_points);
    (void)hipFree(c->d_tabs);
    (void)hipFree(c->d_powtab);
    (void)hipFree(c->d_wide_scratch);
    (void)hipFree(c->d_done);
    (void)hipFree(c->d_left);
    (void)hipFree(c->d_io);
    (void)hipFree(c->d_ovf_bits);
    (void)hipFree(c->d_ovf_lists);
    (void)hipFree(c->d_shard);
    (void)hipFree(c->d_bad);
    if (c->shard_stream) (void)hipStreamDestroy(c->shard_stream);
    if (c->shard_ev) (void)hipEventDestroy(c->shard_ev);
    if (c->shard_sel_ev) (void)hipEventDestroy(c->shard_sel_ev);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_io) (void)hipHostFree(c->h_io);
    if (c->h_status) (void)hipHostFree(c->h_status);
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    delete c;
}

extern "C" int lchd_ctx_set_stream(lchd_ctx* c, void* s) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    if (c->pend.active) return fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)");
    CTX_GUARD(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = reinterpret_cast<hipStream_t>(s);
    return LCHD_OK;
}

extern "C" int lchd_ctx_enable_timing(lchd_ctx* c, int32_t on) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    c->timing = on != 0;
    return LCHD_OK;
}
extern "C" double lchd_ctx_last_ms(lchd_ctx* c, const char* phase) {
    static const char* names[PH_N] = {"cells", "anchors", "env", "sweep"};
    if (!c || !phase) return -1.0;
    for (int i = 0; i < PH_N; ++i)
        if (!strcmp(phase, names[i])) return c->ms[i];
    return -1.0;
}
extern "C" int64_t lchd_ctx_last_env_points(lchd_ctx* c) {
    if (!c || !c->last_valid || c->pend.active) return -1;
    CTX_GUARD(c);
    launch_env_points(c->stream, c->last, c->d_points);
    unsigned long long v = 0;
    if (hipMemcpyAsync(&v, c->d_points, sizeof v, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    return (int64_t)v;
}

// One sweep family for every pair: the one-pair-per-wavefront k_sweep that reads its square-root tables from global memory (tiles
// and per-lane chunks are functions of the pair alone), behind the row-sort kernels for dense rows.  No team sweeps, no
// one-launch small-call sweep, no launch-set hints from earlier passes, no fused dense kernel, no O(1) Kullback-Leibler / Renyi
// form: a pair's score then depends on the pair and the configuration only -- not on the other pairs of the call, on what the
// context scored before, or on whether the pair was reached through a second pass.  (Environments of more than 65 535 points
// still take the 64-bit-count sweep, whatever the mode.)
extern "C" int lchd_ctx_set_deterministic(lchd_ctx* c, int32_t on) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    if (c->pend.active) return fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)");
    c->deterministic = on != 0;
    c->tune = c->tune_env;
    if (c->deterministic) {
        Tuning& t = c->tune;
        t.no_duo = t.no_count8 = t.no_c8_team = t.no_inline_meta = t.force_bigenv = t.no_dense_fused = t.no_sd_inc = t.no_sweep_hint = true;
    }
    c->sweep_hint = 0;
    return LCHD_OK;
}
extern "C" int32_t lchd_ctx_get_deterministic(lchd_ctx* c) { return (c && c->deterministic) ? 1 : 0; }

extern "C" int32_t lchd_ctx_last_dense_fused(lchd_ctx* c) { return (c && c->last_dense_fused) ? 1 : 0; }
extern "C" int64_t lchd_ctx_pass_count(lchd_ctx* c) { return c ? c->n_passes : -1; }

extern "C" int lchd_ctx_set_config(lchd_ctx* c, const lchd_config* cfg) {
    if (!c || !cfg) return fail(LCHD_EVALUE, "null context/config");
    if (c->pend.active) return fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)");
    CTX_GUARD(c);
    const int C = cfg->n_categories;
    if (C <= 0) return fail(LCHD_EVALUE, "The number of possible categories (primitive types) cannot be zero!");
    if (C > kHugeCategories)
        return fail(LCHD_EUNSUPPORTED, "at most %d categories are supported (category ids travel as 16 bits on the device; got %d)", kHugeCategories, C);
    if (C > kWideCategories) {
        // beyond what the sweep's per-lane count columns fit in LDS: a global-memory scratch block per workgroup (k_sweep_wide<.., HUGE>),
        // up to 1 GB of them (256 workgroups at most, 8 at least)
        const size_t per = wide_scratch_bytes_per_wave(C);
        const int waves = (int)std::max<size_t>(8, std::min<size_t>(256, ((size_t)1 << 30) / per));
        if (c->wide_scratch_cap < per * (size_t)waves) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->d_wide_scratch) (void)hipFree(c->d_wide_scratch);
            c->d_wide_scratch = nullptr;
            c->wide_scratch_cap = 0;
            if (hipMalloc(&c->d_wide_scratch, per * (size_t)waves) != hipSuccess) {
                (void)hipGetLastError();
                return fail(LCHD_EUNSUPPORTED, "%d categories need %zu bytes of sweep scratch on the device", C, per * (size_t)waves);
            }
            c->wide_scratch_cap = per * (size_t)waves;
        }
        c->wide_scratch_waves = waves;
    }
    if (int rc = lchd_config_validate(C, C, cfg->category_weights, C)) return rc;
    if (cfg->n_weight_functions <= 0) return fail(LCHD_EVALUE, "at least one weight function is required");
    if (int rc = lchd_sd_validate(cfg->sd_kind, cfg->sd_n_params)) return rc;
    size_t n_params = 0;
    for (int i = 0; i < cfg->n_weight_functions; ++i) {
        const lchd_weight_function& w = cfg->weight_functions[i];
        if (int rc = lchd_wf_validate(w.kind, w.params, w.n_params)) return rc;
        n_params += (size_t)w.n_params;
    }
    // blob: [cat_w][wf entries][wf params][F(+inf) per wf][reciprocal of the CDF's constant divisor per wf][tag pairs]
    const size_t o_w = 0, o_e = o_w + sizeof(double) * C, o_p = o_e + sizeof(WfEntry) * cfg->n_weight_functions;
    const size_t o_f = o_p + sizeof(double) * n_params, o_i = o_f + sizeof(double) * cfg->n_weight_functions;
    const size_t o_t = o_i + sizeof(double) * cfg->n_weight_functions;
    const size_t total = o_t + sizeof(uint64_t) * (size_t)cfg->n_tag_pairs;
    std::vector<char> blob(total + 8);
    c->finf_differ = false;
    memcpy(blob.data() + o_w, cfg->category_weights, sizeof(double) * C);
    WfEntry* ent = reinterpret_cast<WfEntry*>(blob.data() + o_e);
    double* prm = reinterpret_cast<double*>(blob.data() + o_p);
    double* finf = reinterpret_cast<double*>(blob.data() + o_f);
    double* winv = reinterpret_cast<double*>(blob.data() + o_i);
    int off = 0;
    for (int i = 0; i < cfg->n_weight_functions; ++i) {
        const lchd_weight_function& w = cfg->weight_functions[i];
        ent[i] = WfEntry{w.kind, w.n_params, off, 0};
        memcpy(prm + off, w.params, sizeof(double) * w.n_params);
        finf[i] = cdf_eval(w.kind, w.params, w.n_params, (double)INFINITY);
        if (i > 0 && !(finf[i] == finf[0])) c->finf_differ = true;  // (set to false above)
        // hyper_exp divides by sum_i a_i, uniform / kumaraswamy by (x_max - x_min) (cdfs.rs:15-20,44,61): constants of the
        // weight function, so the kernels multiply by the reciprocal (<= 1 ulp from the quotient) instead of dividing per point
        winv[i] = 1.0;
        if (w.kind == LCHD_WF_HYPER_EXP) {
            double norm = 0.0;
            for (int k = 0; k < w.n_params / 2; ++k) norm += w.params[k];
            winv[i] = 1.0 / norm;
        } else if (w.kind == LCHD_WF_UNIFORM || w.kind == LCHD_WF_KUMARASWAMY) {
            winv[i] = 1.0 / (w.params[1] - w.params[0]);
        }
        off += w.n_params;
    }
    uint64_t* tp = reinterpret_cast<uint64_t*>(blob.data() + o_t);
    for (int64_t i = 0; i < cfg->n_tag_pairs; ++i)
        tp[i] = ((uint64_t)(uint32_t)cfg->tag_pairs[2 * i] << 32) | (uint32_t)cfg->tag_pairs[2 * i + 1];
    std::sort(tp, tp + cfg->n_tag_pairs);
    // the words of DevConfig that are not in the blob, appended so that ONE comparison decides whether anything changed
    const int32_t extra[10] = {C, cfg->n_weight_functions, cfg->sd_kind, cfg->sd_n_params, cfg->tag_mode, cfg->tag_accept_same,
                               cfg->tag_accepted_pairs, cfg->tag_ordered, (int32_t)cfg->n_tag_pairs, 0};
    std::vector<char> sig(blob.begin(), blob.begin() + total);
    sig.insert(sig.end(), reinterpret_cast<const char*>(extra), reinterpret_cast<const char*>(extra) + sizeof extra);
    sig.insert(sig.end(), reinterpret_cast<const char*>(cfg->sd_params), reinterpret_cast<const char*>(cfg->sd_params) + sizeof cfg->sd_params);
    if (c->cfg_set && sig == c->cfg_blob_host) return LCHD_OK;  // same configuration as the last call: already on the device
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (total + 8 > c->blob_cap) {
        if (c->d_blob) HIP_TRY(hipFree(c->d_blob));
        c->d_blob = nullptr;
        HIP_TRY(hipMalloc(&c->d_blob, total + 4096));
        c->blob_cap = total + 4096;
    }
    HIP_TRY(hipMemcpy(c->d_blob, blob.data(), total, hipMemcpyHostToDevice));
    c->cfg_set = false;
    c->sweep_hint = 0;  // another configuration: what the last pass looked like says nothing about the next
    c->b_use_once = false;
    c->cfg_blob_host.swap(sig);
    DevConfig h{};
    h.n_categories = C;
    h.n_wf = cfg->n_weight_functions;
    h.sd_kind = cfg->sd_kind;
    h.tag_mode = cfg->tag_mode;
    h.tag_accept_same = cfg->tag_accept_same;
    h.tag_accepted_pairs = cfg->tag_accepted_pairs;
    h.tag_ordered = cfg->tag_ordered;
    h.n_tag_pairs = (int32_t)cfg->n_tag_pairs;
    h.sd_p0 = cfg->sd_n_params > 0 ? cfg->sd_params[0] : 0.0;
    h.sd_p1 = cfg->sd_n_params > 1 ? cfg->sd_params[1] : 0.0;
    h.cat_w = reinterpret_cast<const double*>(c->d_blob + o_w);
    h.wf = reinterpret_cast<const WfEntry*>(c->d_blob + o_e);
    h.wf_params = reinterpret_cast<const double*>(c->d_blob + o_p);
    h.wf_finf = reinterpret_cast<const double*>(c->d_blob + o_f);
    h.wf_inv = reinterpret_cast<const double*>(c->d_blob + o_i);
    h.tag_pairs = reinterpret_cast<const uint64_t*>(c->d_blob + o_t);
    h.pow_tab = nullptr;
    if (cfg->sd_kind == LCHD_SD_HELLINGER && cfg->sd_params[0] != 2.0) {  // power tables of the general-exponent Hellinger distance
        if (!c->d_powtab) HIP_TRY(hipMalloc(&c->d_powtab, sizeof(double) * 2 * 65536));
        if (c->powtab_e != cfg->sd_params[0]) {
            launch_fill_pow_tables(c->stream, c->d_powtab, cfg->sd_params[0]);
            HIP_TRY(hipGetLastError());
            c->powtab_e = cfg->sd_params[0];
        }
        h.pow_tab = c->d_powtab;
    }
    HIP_TRY(hipMemcpy(c->d_cfg, &h, sizeof h, hipMemcpyHostToDevice));
    c->h_cfg = h;
    c->hellinger2 = (cfg->sd_kind == LCHD_SD_HELLINGER && cfg->sd_params[0] == 2.0);
    c->unit_weights = std::all_of(cfg->category_weights, cfg->category_weights + C, [](double v) { return v == 1.0; });
    // k_sweep_inc drops terms of second order in eps * (largest count): eps <= 1e-9 keeps them below 3e-13.  Renyi: a moderate order
    // (the tables hold k^alpha for k <= 512) and a finite eps^(1 - alpha); alpha = 1 is the Kullback-Leibler form (:36-38).
    c->sd_fast = 0;
    if (c->unit_weights) {
        if (cfg->sd_kind == LCHD_SD_KULLBACK_LEIBLER && cfg->sd_params[0] > 0.0 && cfg->sd_params[0] <= 1e-9) c->sd_fast = 1;
        if (cfg->sd_kind == LCHD_SD_RENYI && cfg->sd_params[1] > 0.0 && cfg->sd_params[1] <= 1e-9) {
            const double al = cfg->sd_params[0];
            if (al == 1.0) c->sd_fast = 1;
            else if (al >= 0.01 && al <= 20.0 && std::isfinite(std::pow(cfg->sd_params[1], 1.0 - al)) && std::pow(cfg->sd_params[1], 1.0 - al) < 1e250 &&
                     cfg->sd_params[1] * 512.0 * std::max(1.0, std::fabs(al - 1.0)) <= 1e-6)
                c->sd_fast = 2;
        }
        // Kolmogorov-Smirnov on unit weights: max_c |a_c N_b - b_c N_a| / (N_a N_b) on integer counts (the KSM team sweep)
        if (cfg->sd_kind == LCHD_SD_KOLMOGOROV_SMIRNOV) c->sd_fast = 3;
    }
    c->wf_pow = false;
    for (int i = 0; i < cfg->n_weight_functions; ++i)
        c->wf_pow = c->wf_pow || cfg->weight_functions[i].kind == LCHD_WF_DAGUM || cfg->weight_functions[i].kind == LCHD_WF_KUMARASWAMY ||
                    (cfg->weight_functions[i].kind == LCHD_WF_HYPER_EXP && cfg->weight_functions[i].n_params > 8);  // not register-resident
    c->cfg_set = true;
    return LCHD_OK;
}

// ------------------------------------------------------------------------------------------------
// clouds
// ------------------------------------------------------------------------------------------------
static int upload_coords(lchd_ctx* c, lchd_cloud* cl, const double* xyz) {
    const int64_t n = cl->n;
    std::vector<double> soa((size_t)3 * n);
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const double v = xyz[3 * i + k];
            if (!std::isfinite(v)) return fail(LCHD_EVALUE, "non-finite coordinate at atom %lld", (long long)i);
            soa[(size_t)k * n + i] = v;
            mn[k] = std::min(mn[k], v);
            mx[k] = std::max(mx[k], v);
        }
    for (int k = 0; k < 3; ++k) { cl->bbmin[k] = n ? mn[k] : 0.0; cl->bbmax[k] = n ? mx[k] : 0.0; }
    if (n) {
        HIP_TRY(hipMemcpyAsync(cl->x, soa.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(cl->y, soa.data() + n, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(cl->z, soa.data() + 2 * n, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return LCHD_OK;
}

// Category ids travel as one byte (255 = not in the category map) -- or, as soon as an id beyond 254 occurs (more than 255
// categories), as two: low byte | high byte, 0xFFFF = not in the map.
static bool cats_need_hi(const int32_t* cat, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        if (cat[i] >= 255 && cat[i] < 65535) return true;
    return false;
}
static void cats_encode(const int32_t* cat, int64_t n, uint8_t* lo, uint8_t* hi /* or nullptr */) {
    for (int64_t i = 0; i < n; ++i) {
        if (hi) {
            const uint32_t v = (cat[i] >= 0 && cat[i] < 65535) ? (uint32_t)cat[i] : 0xFFFFu;
            lo[i] = (uint8_t)(v & 255u);
            hi[i] = (uint8_t)(v >> 8);
        } else {
            lo[i] = (cat[i] >= 0 && cat[i] < 255) ? (uint8_t)cat[i] : (uint8_t)255;
        }
    }
}

extern "C" int lchd_cloud_create(lchd_ctx* c, const double* xyz, const int32_t* cat, const int32_t* tag, int64_t n, lchd_cloud** out) {
    if (!c || !out) return fail(LCHD_EVALUE, "null context");
    *out = nullptr;
    if (n < 0 || n > (int64_t)1 << 30) return fail(LCHD_EUNSUPPORTED, "cloud size %lld out of range", (long long)n);
    if (n > 0 && !cat) return fail(LCHD_EVALUE, "null category array");
    CTX_GUARD(c);
    lchd_cloud* cl = new lchd_cloud();
    cl->n = n;
    const size_t m = (size_t)std::max<int64_t>(n, 1);
    auto bail = [&](hipError_t e, const char* what) {  // nothing allocated so far outlives a failed call
        lchd_cloud_destroy(c, cl);
        return fail(LCHD_EDEVICE, "HIP error %d (%s) in lchd_cloud_create: %s", (int)e, hipGetErrorName(e), what);
    };
    hipError_t e;
    if ((e = hipMalloc(&cl->x, sizeof(double) * m)) != hipSuccess) return bail(e, "hipMalloc(x)");
    if ((e = hipMalloc(&cl->y, sizeof(double) * m)) != hipSuccess) return bail(e, "hipMalloc(y)");
    if ((e = hipMalloc(&cl->z, sizeof(double) * m)) != hipSuccess) return bail(e, "hipMalloc(z)");
    if ((e = hipMalloc(&cl->cat, m)) != hipSuccess) return bail(e, "hipMalloc(cat)");
    if ((e = hipMalloc(&cl->tag, sizeof(int32_t) * m)) != hipSuccess) return bail(e, "hipMalloc(tag)");
    if (xyz) {
        if (int rc = upload_coords(c, cl, xyz)) { lchd_cloud_destroy(c, cl); return rc; }
    } else {
        if ((e = hipMemsetAsync(cl->x, 0, sizeof(double) * m, c->stream)) != hipSuccess) return bail(e, "hipMemset(x)");
        if ((e = hipMemsetAsync(cl->y, 0, sizeof(double) * m, c->stream)) != hipSuccess) return bail(e, "hipMemset(y)");
        if ((e = hipMemsetAsync(cl->z, 0, sizeof(double) * m, c->stream)) != hipSuccess) return bail(e, "hipMemset(z)");
    }
    if (n) {
        const bool wide = cats_need_hi(cat, n);
        std::vector<uint8_t> c8((size_t)n), c8h(wide ? (size_t)n : 0);
        cats_encode(cat, n, c8.data(), wide ? c8h.data() : nullptr);
        if ((e = hipMemcpy(cl->cat, c8.data(), (size_t)n, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(cat)");
        if (wide) {
            if ((e = hipMalloc(&cl->cat_hi, (size_t)n)) != hipSuccess) return bail(e, "hipMalloc(cat_hi)");
            if ((e = hipMemcpy(cl->cat_hi, c8h.data(), (size_t)n, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(cat_hi)");
            cats_encode(cat, n, c8.data(), nullptr);  // the one-byte view: ids beyond 254 are outside every map of at most 255 categories
            if ((e = hipMalloc(&cl->cat_narrow, (size_t)n)) != hipSuccess) return bail(e, "hipMalloc(cat_narrow)");
            if ((e = hipMemcpy(cl->cat_narrow, c8.data(), (size_t)n, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(cat_narrow)");
        }
        if (tag) e = hipMemcpy(cl->tag, tag, sizeof(int32_t) * n, hipMemcpyHostToDevice);
        else e = hipMemset(cl->tag, 0, sizeof(int32_t) * n);
        if (e != hipSuccess) return bail(e, "tags");
    }
    // (the copies and fills above went through the null stream; the context may launch on a non-blocking stream that is not
    // ordered behind it: creating a structure is rare, so simply wait for the device)
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(e, "hipDeviceSynchronize");
    *out = cl;
    return LCHD_OK;
}

extern "C" int lchd_cloud_create_batch(lchd_ctx* c, const double* xyz, const int32_t* cat, const int32_t* tag, const int32_t* sid,
                                       int64_t n, int32_t n_struct, lchd_cloud** out) {
    if (!c || !out) return fail(LCHD_EVALUE, "null context");
    if (n_struct < 1 || !sid) return fail(LCHD_EVALUE, "a batch needs n_struct >= 1 and a structure id per atom");
    for (int64_t i = 0; i < n; ++i)
        if (sid[i] < 0 || sid[i] >= n_struct) return fail(LCHD_EVALUE, "structure id %d of atom %lld is outside [0, %d)", sid[i], (long long)i, n_struct);
    CTX_GUARD(c);
    lchd_cloud* cl = nullptr;
    if (int rc = lchd_cloud_create(c, xyz, cat, tag, n, &cl)) return rc;
    cl->n_struct = n_struct;
    if (n && n % n_struct == 0 && n / n_struct <= INT32_MAX) {  // regular batch: equal-sized structures stored one after the other?
        const int64_t k = n / n_struct;
        bool regular = true;
        for (int64_t i = 0; i < n && regular; ++i) regular = sid[i] == (int32_t)(i / k);
        cl->struct_size = regular ? (int32_t)k : 0;
    }
    if (n) {
        hipError_t e = hipMalloc(&cl->sid, sizeof(int32_t) * n);
        if (e == hipSuccess) e = hipMemcpy(cl->sid, sid, sizeof(int32_t) * n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { lchd_cloud_destroy(c, cl); return fail(LCHD_EDEVICE, "HIP error %d while uploading structure ids", (int)e); }
    }
    *out = cl;
    return LCHD_OK;
}

extern "C" int64_t lchd_cloud_size(const lchd_cloud* cl) { return cl ? cl->n : -1; }

extern "C" int lchd_cloud_set_coords(lchd_ctx* c, lchd_cloud* cl, const double* xyz) {
    if (!c || !cl || !xyz) return fail(LCHD_EVALUE, "null argument");
    if (c->pend.active && (c->pend.a == cl || c->pend.b == cl))
        return fail(LCHD_EVALUE, "this structure is in use by an unfinished asynchronous call");
    CTX_GUARD(c);
    return upload_coords(c, cl, xyz);
}

extern "C" void lchd_cloud_destroy(lchd_ctx* c, lchd_cloud* cl) {
    if (!cl) return;
    DeviceGuard device_guard_(c ? c->device : -1);
    if (c) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(cl->x);
    (void)hipFree(cl->y);
    (void)hipFree(cl->z);
    (void)hipFree(cl->cat);
    (void)hipFree(cl->cat_hi);
    (void)hipFree(cl->cat_narrow);
    (void)hipFree(cl->tag);
    (void)hipFree(cl->sid);
    (void)hipFree(cl->d_raw);
    (void)hipFree(cl->d_bbox);
    (void)hipFree(cl->d_bbox_part);
    (void)hipFree(cl->d_src_start);
    (void)hipFree(cl->d_src_idx);
    (void)hipFree(cl->d_tiles);
    (void)hipFree(cl->d_raw32);
    if (cl->h_pinned32) (void)hipHostFree(cl->h_pinned32);
    if (cl->h_pinned) (void)hipHostFree(cl->h_pinned);
    if (cl->ev_ready) (void)hipEventDestroy(cl->ev_ready);
    if (cl->ev_used) (void)hipEventDestroy(cl->ev_used);
    if (cl->ev_t0) (void)hipEventDestroy(cl->ev_t0);
    if (cl->ev_t1) (void)hipEventDestroy(cl->ev_t1);
    delete cl;
}

// ------------------------------------------------------------------------------------------------
// status -> reference error classes
// ------------------------------------------------------------------------------------------------
enum Driver { DRV_ANCHORS, DRV_DMXS, DRV_PRIMS };

static int status_to_rc(uint32_t f, Driver drv) {
    if (f == 0) return LCHD_OK;
    if (f & ST_BAD_ANCHOR) return fail(LCHD_EPANIC, "index out of bounds: an anchor index is outside its structure (src/locohd.rs:521)");
    if (f & ST_EMPTY_ENV) return fail(LCHD_EPANIC, "index out of bounds: an environment is empty (src/locohd.rs:74)");
    if (f & ST_BAD_WF) return fail(LCHD_EVALUE, "weight-function index out of range");
    if (drv == DRV_ANCHORS) {
        if (f & ST_FIRST_NOT_ZERO) return fail(LCHD_EVALUE, "The dists list must start with a distance of 0!");
        if (f & ST_BAD_CATEGORY) return fail(LCHD_EVALUE, "Category not found!");
        if (f & ST_ZERO_NORM) return fail(LCHD_EVALUE, "Zero norm error for PMF");
    }
    if (f & (ST_FIRST_NOT_ZERO | ST_BAD_CATEGORY | ST_ZERO_NORM | ST_BAD_DISTANCE))
        return fail(LCHD_EVALUE, drv == DRV_PRIMS ? "The from_anchors function returned an error during the LoCoHD calculations!"
                                                  : "The stat_dist_integral function returned an error during the LoCoHD calculations!");
    return fail(LCHD_EDEVICE, "unexpected device status 0x%x", f);
}

// Status protocol of a pass.  begin_pass: the device status is clean unless a pass was abandoned half-way; the host-mapped
// mirror's error words and sequence number are reset by the CPU (no pass is in flight).  After the stream has been waited
// for, pass_flags() = what the kernels up to the record pass reported (snapshot by k_pair_meta) | what the sweeps reported.
static int begin_pass(lchd_ctx* c) {
    if (c->status_dirty) {
        HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(DeviceStatus), c->stream));
        HIP_TRY(hipMemsetAsync(c->d_done, 0, sizeof(DoneState), c->stream));
        HIP_TRY(hipMemsetAsync(c->d_left, 0, 256, c->stream));
        c->status_dirty = false;
    }
    HostStatus* h = c->h_status;
    for (uint32_t& w : h->sweep_flags) w = 0u;
    h->flags = 0u;
    h->max_env = 0u;
    h->snapshot_seq = 0u;
    c->seq = c->seq == 0xFFFFFFFFu ? 1u : c->seq + 1u;
    return LCHD_OK;
}
static int pass_flags(lchd_ctx* c, uint32_t* flags) {
    const HostStatus* h = c->h_status;
    if (h->snapshot_seq != c->seq) {  // the record pass never ran to its end
        c->status_dirty = true;
        return fail(LCHD_EDEVICE, "the device did not complete the pass (status snapshot %u, expected %u)", h->snapshot_seq, c->seq);
    }
    uint32_t f = h->flags;
    for (int k = 0; k < 8; ++k)
        if (h->sweep_flags[k]) f |= 1u << k;
    *flags = f;
    return LCHD_OK;
}
static int wait_pass(lchd_ctx* c, uint32_t* flags) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        c->status_dirty = true;
        return fail(LCHD_EDEVICE, "HIP error %d (%s) while waiting for the pass", (int)e, hipGetErrorName(e));
    }
    return pass_flags(c, flags);
}

static void mark(lchd_ctx* c, int i) {
    if (c->timing) (void)hipEventRecord(c->ev[i], c->stream);
}
static void collect_times(lchd_ctx* c, int first, int last) {
    for (int i = 0; i < PH_N; ++i) c->ms[i] = -1.f;
    if (!c->timing) return;
    for (int i = first; i < last; ++i) {
        float t = -1.f;
        if (hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]) == hipSuccess) c->ms[i] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// from_primitives on device-resident clouds
// ------------------------------------------------------------------------------------------------
struct GridPlan {
    double min[3], inv[3];
    int dim[3];
    int n_cells;
};

static GridPlan plan_grid(const lchd_cloud* cl, double thr, int reach) {
    GridPlan g{};
    // cells are at least (1 + 1e-9) * thr / reach wide so that |dx| < thr can never leave the (2 reach + 1)^3 neighbourhood of
    // the anchor's cell through rounding (reach 1: k_env_cells / k_env_collect; reach 2: k_env_group)
    const double cell0 = thr / reach * (1.0 + 1e-9);
    long long total = 1;
    for (int k = 0; k < 3; ++k) {
        const double ext = cl->bbmax[k] - cl->bbmin[k];
        double nd = std::floor(ext / cell0);
        if (!(nd >= 1.0)) nd = 1.0;
        if (nd > 1024.0) nd = 1024.0;
        g.dim[k] = (int)nd;
        total *= g.dim[k];
    }
    total *= cl->n_struct;
    while (total > (1ll << 23)) {  // bound the cell arrays (8M cells): coarsen the largest axis
        int k = (g.dim[0] >= g.dim[1] && g.dim[0] >= g.dim[2]) ? 0 : (g.dim[1] >= g.dim[2] ? 1 : 2);
        if (g.dim[k] == 1) break;
        total /= g.dim[k];
        g.dim[k] = (g.dim[k] + 1) / 2;
        total *= g.dim[k];
    }
    for (int k = 0; k < 3; ++k) {
        const double ext = cl->bbmax[k] - cl->bbmin[k];
        g.min[k] = cl->bbmin[k];
        const double cell = ext > 0.0 ? ext / g.dim[k] : 1.0;
        g.inv[k] = 1.0 / (cell * (1.0 + 1e-12));
    }
    g.n_cells = cl->n_struct * g.dim[0] * g.dim[1] * g.dim[2];
    return g;
}

struct SideBufs {
    uint32_t *cell_of, *cell_count, *cell_start, *pos_of, *slot, *scan_tmp, *bits, *wpre, *chunk_base;
    uint8_t* flag8;
    AnchorRec* uniq;
    CellRec* rec;
    EnvStore env;
    double* raw_key;   // environments of more than 16384 points: unsorted scratch rows (k_env_collect)
    uint8_t* raw_cat;
    uint32_t* ovf_list;  // slots of the environments that overflowed (EnvSide::ovf_list)
};

// Workspace layout of one pass.  Everything that must be zero when the prologue starts -- the general cell list's counters
// and the anchor flags of both sides -- is carved as ONE contiguous region (flags_a, flags_b last), so at most one memset
// (or none: the fused / per-structure prologue launches zero what they need themselves) precedes the kernels.
struct PassBufs {
    SideBufs a, b;
    int4* pair_meta;
    uint32_t* left_list;  // pairs the team kernel's rule leaves over (SweepArgs::left_list)
    char* zero_base;
    size_t zero_bytes;
};
static void carve_pass(Arena& ar, int64_t n_a, int cells_a, int64_t envs_a, int64_t n_b, int cells_b, int64_t envs_b, int cap, int64_t n_pairs,
                       PassBufs& pb, bool cat16 = false, int key_sets = 1, int pre_words_a = 0, int pre_words_b = 0) {
    const size_t ma = (size_t)std::max<int64_t>(n_a, 1), mb = (size_t)std::max<int64_t>(n_b, 1);
    ar.off = (ar.off + 255) & ~size_t(255);
    const size_t z0 = ar.off;
    pb.a.cell_count = ar.take<uint32_t>((size_t)cells_a + 1);
    pb.b.cell_count = ar.take<uint32_t>((size_t)cells_b + 1);
    pb.a.flag8 = reinterpret_cast<uint8_t*>(ar.take<uint4>((ma + 31) / 32 * 2 + 2));  // one byte per atom, padded to whole 32-byte words
    pb.b.flag8 = reinterpret_cast<uint8_t*>(ar.take<uint4>((mb + 31) / 32 * 2 + 2));
    pb.zero_base = ar.dry ? nullptr : ar.base + z0;
    pb.zero_bytes = ar.off - z0;
    for (int side = 0; side < 2; ++side) {
        SideBufs& b = side ? pb.b : pb.a;
        const size_t m = side ? mb : ma;
        const int n_cells = side ? cells_b : cells_a;
        const size_t ne = (size_t)std::max<int64_t>(side ? envs_b : envs_a, 1);
        b.cell_of = ar.take<uint32_t>(m);
        b.slot = ar.take<uint32_t>(m + 1);
        b.bits = ar.take<uint32_t>((m + 31) / 32 + 1);
        b.wpre = ar.take<uint32_t>((m + 31) / 32 + 2);
        b.chunk_base = ar.take<uint32_t>((m >> 18) + 2);
        b.cell_start = ar.take<uint32_t>((size_t)n_cells + 1);
        b.rec = ar.take<CellRec>(m + kEnvGroupRecPad);  // (k_env_group reads up to 7 records past a cell row's end)
        b.pos_of = ar.take<uint32_t>(m);
        b.uniq = ar.take<AnchorRec>(ne);
        b.scan_tmp = ar.take<uint32_t>(std::max<size_t>(m, (size_t)n_cells) / 4096 + 4);
        b.env.key = ar.take<uint64_t>(ne * (size_t)cap * (size_t)std::max(key_sets, 1));
        b.env.set_stride = (int64_t)(ne * (size_t)cap);
        b.env.cat = ar.take<uint8_t>(ne * (size_t)cap * (cat16 ? 2 : 1));
        b.env.len = ar.take<int32_t>(ne);
        b.env.cat0 = (cap == kEnvGroupCap && !cat16) ? ar.take<uint8_t>(ne) : nullptr;  // (written by k_env_group only: prims_enqueue drops it otherwise)
        const int pw = side ? pre_words_b : pre_words_a;  // prefix-count rows (k_env_group, configurations of at most 16 categories)
        b.env.pre = pw > 0 ? ar.take<uint64_t>(ne * (size_t)(cap / kPreStep) * (size_t)pw) : nullptr;  // (one row per kPreStep points)
        b.env.pre_words = pw;
        b.env.stride = cap;
        b.env.cdf_keys = 0;
        b.env.cat16 = cat16 ? 1 : 0;
        b.raw_key = cap > 16384 ? ar.take<double>(ne * (size_t)cap) : nullptr;
        b.raw_cat = cap > 16384 ? ar.take<uint8_t>(ne * (size_t)cap) : nullptr;
        b.ovf_list = cap <= 16384 ? ar.take<uint32_t>(ne) : nullptr;
    }
    pb.pair_meta = ar.take<int4>((size_t)n_pairs);
    pb.left_list = ar.take<uint32_t>((size_t)n_pairs);
}

static int next_pow2_host(int64_t n) {
    int p = 64;
    while (p < n) p <<= 1;
    return p;
}

// A frames buffer computes its bounding box on the device while it is being filled; fetch it before planning a grid.
static int resolve_bbox(lchd_ctx* c, lchd_cloud* cl) {
    if (!cl->bbox_pending) return LCHD_OK;
    HIP_TRY(hipEventSynchronize(cl->ev_ready));
    unsigned long long k[7];
    HIP_TRY(hipMemcpy(k, cl->d_bbox, sizeof k, hipMemcpyDeviceToHost));
    cl->bbox_pending = false;
    if (k[6]) return fail(LCHD_EVALUE, "non-finite coordinate in a trajectory frame");
    auto dec = [](unsigned long long u) { u = (u >> 63) ? (u ^ 0x8000000000000000ull) : ~u; double d; memcpy(&d, &u, 8); return d; };
    for (int q = 0; q < 3; ++q) { cl->bbmin[q] = dec(k[q]); cl->bbmax[q] = dec(k[3 + q]); }
    return LCHD_OK;
}

static void fill_sweep_args(lchd_ctx* c, SweepArgs& sw) {
    sw.cfg = c->d_cfg;
    sw.st = c->d_status;
    sw.hst = c->h_status;
    sw.seq = c->seq;
    sw.sd_fast = c->tune.no_sd_inc ? 0 : c->sd_fast;
    sw.sqrt_tab = c->d_tabs;
    sw.rsqrt_tab = c->d_tabs + 65536;
    sw.done = c->d_done;
    sw.wide_scratch = c->d_wide_scratch;
    sw.wide_scratch_per_wave = (int64_t)wide_scratch_bytes_per_wave(c->h_cfg.n_categories);
    sw.wide_scratch_waves = c->wide_scratch_waves;
}

// Everything of one from_primitives pass; no host synchronisation (the workspace only grows between passes).
static int prims_enqueue(lchd_ctx* c) {
    auto& P = c->pend;
    ++c->n_passes;
    lchd_cloud *a = P.a, *b = P.b;
    const int64_t n_pairs = P.n_pairs;
    const double thr = P.thr;
    const int cap = P.cap;
    // environments of the default capacity: several per wavefront on a grid of half-threshold cells (lchd_env_group.hip)
    // more than 255 categories: 16-bit ids in the environment store, k_env_cells<.., uint16_t> + k_sweep_wide<.., CAT16>
    const bool cat16 = c->h_cfg.n_categories > kMaxCategories;
    // Both sides the SAME device object (all-vs-all over one batch of structures, a structure against itself): an anchor's
    // environment does not depend on the side it is used on (src/locohd.rs:514-542 is one closure for both), so the cell
    // list and every environment are built once -- the anchors of both columns share side A's flags, slots and store.
    const bool same = (a == b) && !c->tune.no_share;
    const int64_t max_env_a = same ? std::min<int64_t>(a->n, 2 * n_pairs) : std::min<int64_t>(a->n, n_pairs);
    int64_t max_env_b = same ? 0 : std::min<int64_t>(b->n, n_pairs);  // (side B without de-duplication: one slot per PAIR, below)
    // (the grouped kernel addresses environment slots and records with 32-bit offsets: the limits of launch_env_group; larger
    //  calls take k_env_cells, which has none)
    const bool group = cap == kEnvGroupCap && !c->tune.no_env_group && a->n < ((int64_t)1 << 27) && b->n < ((int64_t)1 << 27) &&
                       max_env_a < ((int64_t)1 << 22) && max_env_b < ((int64_t)1 << 22) && !cat16;
    P.group = group;
    P.group_small = false;
    // Side B used once -- (almost) every side-B anchor of the last regular pass of this context was unique: the frames of a trajectory,
    // (i, i) lists, a rank's partners under strong scaling.  Such a side is not de-duplicated: environment slot p belongs to pair p
    // and its anchor record comes straight from the pair list (launch_pair_anchor_recs) -- no flags, bit set, scan and scatter over
    // the side's atoms.  Any choice is correct for any input: an anchor that occurs in several pairs is built once per pair, as the
    // reference does (src/locohd.rs:514-554).  Every 64th pass is a regular one again (this mode does not count unique anchors).
    const bool per_pair_ok = group && !same && !P.subset && !c->deterministic && c->tune.per_pair >= 0 && n_pairs < ((int64_t)1 << 22) && n_pairs > 0;
    // (history alone is not enough: the hint must have come from a list of this size on a structure of this size, and a list with more
    //  pairs than the side has atoms repeats anchors by counting -- C2a: 10^6 pairs over 10^4 atoms right after a list of (i, i) pairs)
    const bool like_hinted = n_pairs <= b->n && b->n == c->use_once_nb && 2 * n_pairs >= c->use_once_pairs && n_pairs <= 2 * c->use_once_pairs;
    const bool per_pair = per_pair_ok && (c->tune.per_pair > 0 || (c->b_use_once && like_hinted && n_pairs > 4096 && c->per_pair_streak < 64));
    P.per_pair = per_pair;
    if (per_pair) max_env_b = n_pairs;  // (one slot per PAIR)
    const GridPlan ga = plan_grid(a, thr, group ? 2 : 1), gb = plan_grid(b, thr, group ? 2 : 1);
    // Keys of the store: F(distance) whenever the sweep can use them without evaluating a CDF -- one weight function, or a
    // dictionary of up to kMaxKeySets (src/locohd.rs:230-283: every pair names its function): k_env_group writes one key set per
    // function (the sort is shared, the store's key part grows k-fold) and a pair reads the set of its function.
    const int n_wf = c->h_cfg.n_wf;
    // (dictionary: set 0 keeps the distances k_env_group writes, k_env_key_sets fills sets 1 .. n_wf; the sweeps' view starts at set 1)
    const bool dict_sets = n_wf > 1 && group && n_wf <= kMaxKeySets && P.wf && !c->tune.no_key_sets && !c->tune.no_cdf_keys && !c->finf_differ;
    const int key_sets = c->tune.no_cdf_keys ? 0 : (n_wf == 1 ? 1 : (dict_sets ? n_wf + 1 : 0));
    // Prefix-count rows next to the environments (EnvStore::pre, 8 or 16 bytes per point): the team sweeps of up to 16 category slots
    // read a chunk's start counts from them instead of building a histogram and a scan per tile.  Worth their write when environments
    // are swept more than once: not for a side without de-duplication (one pair per environment), not for small calls (one pair per
    // wavefront: the one-launch sweep), only where the team sweeps exist (Hellinger-2 / Kolmogorov-Smirnov on unit weights).
    int pre_words = 0;
    {
        const int cm = std::max(c->h_cfg.n_categories, c->tune.force_cmax);
        const bool team_cfg = (c->hellinger2 || c->sd_fast == 3) && c->unit_weights && key_sets >= 1 && !c->tune.no_duo && !c->tune.no_count8 &&
                              !c->tune.no_c8_team && !c->tune.force_generic && !c->tune.force_wide && !c->tune.force_bigenv;
        // (17 .. 28 slots -- three / four count words, the LDS-byte form of the team sweep -- were built and measured in round 6: C5's
        //  k_sweep_duo<28, 32, 480> 1.690 ms with rows against 1.679 without, k_env_group 0.93 against 0.77 ms: no rows there)
        if (group && team_cfg && cm <= 16 && !per_pair && !c->deterministic && c->tune.pre_rows >= 0 && (n_pairs > 4096 || c->tune.no_inline_meta || c->tune.pre_rows > 0))
            pre_words = team_pre_words(cm);
    }
    PassBufs pb{};
    {
        Arena dry(nullptr, 0, true);
        carve_pass(dry, a->n, ga.n_cells, max_env_a, b->n, gb.n_cells, max_env_b, cap, n_pairs, pb, cat16, key_sets, pre_words, pre_words);
        if (int rc = ensure_ws(c, dry.off + 4096)) return rc;
    }
    Arena ar(c->ws, c->ws_cap, false);
    carve_pass(ar, a->n, ga.n_cells, max_env_a, b->n, gb.n_cells, max_env_b, cap, n_pairs, pb, cat16, key_sets, pre_words, pre_words);
    SideBufs &sa = pb.a, &sb = pb.b;
    sa.env.cdf_keys = sb.env.cdf_keys = dict_sets ? 0 : key_sets;  // (what the environment kernels write into set 0)
    if (!group) sa.env.cat0 = sb.env.cat0 = nullptr;

    auto grid_view = [](const GridPlan& g, const SideBufs& s) {
        GridView v{};
        for (int k = 0; k < 3; ++k) { v.min[k] = g.min[k]; v.inv[k] = g.inv[k]; v.cell[k] = 1.0 / g.inv[k]; v.dim[k] = g.dim[k]; }
        v.n_cells = g.n_cells;
        v.cell_start = s.cell_start;
        v.rec = s.rec;
        v.pos_of = s.pos_of;
        return v;
    };
    const GridView gva = grid_view(ga, sa), gvb = grid_view(gb, sb);
    const CloudView cva = a->view(cat16), cvb = b->view(cat16);
    hipStream_t s = c->stream;
    // frames buffers are filled on another stream: order this pass behind their upload
    if (a->ev_ready && a->cap_frames) HIP_TRY(hipStreamWaitEvent(s, a->ev_ready, 0));
    if (b->ev_ready && b->cap_frames) HIP_TRY(hipStreamWaitEvent(s, b->ev_ready, 0));
    if (int rc = begin_pass(c)) return rc;
    c->status_dirty = true;  // until the whole pass has been enqueued
    mark(c, 0);
    auto prep_side = [](const CloudView& cv, const GridView& gv, const SideBufs& sbuf) {
        PrepSide ps{};
        ps.c = cv; ps.g = gv;
        ps.cell_start = sbuf.cell_start; ps.rec = sbuf.rec; ps.pos_of = sbuf.pos_of;
        ps.cell_of = sbuf.cell_of; ps.cell_count = sbuf.cell_count; ps.scan_tmp = sbuf.scan_tmp;
        ps.flag8 = sbuf.flag8; ps.bits = sbuf.bits; ps.wpre = sbuf.wpre; ps.chunk_base = sbuf.chunk_base; ps.slot = sbuf.slot; ps.uniq = sbuf.uniq;
        return ps;
    };
    {
        PrepSide psa = prep_side(cva, gva, sa), psb = prep_side(cvb, gvb, sb);
        psb.no_anchors = per_pair ? 1 : 0;  // (side B without de-duplication: no flags, no slots -- one environment per pair)
        (void)launch_prologue(s, c->tune, P.anchors, n_pairs, psa, psb, pb.zero_base, pb.zero_bytes, c->d_status, same);
        if (per_pair) launch_pair_anchor_recs(s, P.anchors, n_pairs, psb, c->d_status);
    }
    mark(c, 1);
    mark(c, 2);  // (cell lists and anchor de-duplication are one phase now; "anchors" reads 0)
    const bool tag_list = c->h_cfg.tag_mode != 0;
    const EnvSide esa{cva, gva, sa.uniq, sa.env, max_env_a, sa.raw_key, sa.raw_cat, sa.ovf_list},
                  esb{cvb, gvb, sb.uniq, sb.env, max_env_b, sb.raw_key, sb.raw_cat, sb.ovf_list};
    P.ovf_a = sa.ovf_list;
    P.ovf_b = sb.ovf_list;
    P.n_slots_a = max_env_a;
    P.n_slots_b = max_env_b;
    c->last_store_bytes += (size_t)(std::max<int64_t>(max_env_a, 1) + std::max<int64_t>(max_env_b, 1)) * (size_t)cap * ((cat16 ? 2 : 1) + 8 * (size_t)std::max(key_sets, 1));
    if (group) {
        // anchors per wavefront: as many as fit ONE group of the kernel's LDS buffer (measured, env phase in ms for 1 / 2 / 4 / 8 /
        // 16 anchors: C4, ~96-point environments 3.87 / 2.88 / 2.68 / 2.69 / 2.96; C5, ~200 points 0.81 / 0.75 / 0.75 / 0.78 / 0.81 --
        // more anchors per wavefront only lengthen the tail of the launch)
        // A call with few anchors is bound by the latency of one wavefront's chain, not by throughput: one anchor per wavefront
        // until there are enough of them to fill the chip twice (3000-atom structure pair: 19.9 -> 11.7 us).
        const int by_size = c->last_biggest > 0 && c->last_biggest <= 140 ? 4 : (c->last_biggest > kEnvGroupSmallUpTo ? 1 : 2);
        const int apw = c->tune.env_apw > 0 ? c->tune.env_apw
                                            : (int)std::max<int64_t>(1, std::min<int64_t>(by_size, (max_env_a + max_env_b) / 8192));
        P.group_small = c->group_small;
        if (!launch_env_group(s, c->d_cfg, tag_list, P.group_small, esa, esb, thr, apw, c->d_status))
            return fail(LCHD_EDEVICE, "the grouped environment kernel rejected its launch configuration");
    } else if (!launch_env_cells(s, cap, c->d_cfg, tag_list, esa, esb, thr, c->d_status))
        return fail(LCHD_EUNSUPPORTED, cat16 ? "with more than 255 categories an environment may hold at most 8192 points (capacity %d asked for)"
                                             : "no environment kernel variant with capacity %d", cap);
    if (c->deterministic)  // one order among equal keys, whatever order the cell lists' and the buckets' atomics produced (utils.rs:25-39: a stable sort)
        launch_env_canon(s, sa.env, same ? sa.env : sb.env, max_env_a, same ? 0 : max_env_b, c->d_status);
    if (dict_sets) {
        launch_env_key_sets(s, c->d_cfg, sa.env, sb.env, n_wf, max_env_a + max_env_b, c->d_status);
        for (SideBufs* sb_ : {&sa, &sb}) {  // the sweeps' view of the store: the F sets
            sb_->env.key += sb_->env.set_stride;
            sb_->env.cdf_keys = n_wf;
        }
    }
    mark(c, 3);
    SweepArgs sw{};
    fill_sweep_args(c, sw);
    sw.env_a = sa.env;
    sw.env_b = same ? sa.env : sb.env;
    sw.anchors = P.anchors;
    sw.slot_a = sa.slot;
    sw.slot_b = same ? sa.slot : (per_pair ? nullptr : sb.slot);  // (nullptr: side B's slot of pair p is p)
    sw.n_slot_a = a->n;
    sw.n_slot_b = b->n;
    sw.wf_index = P.wf;
    sw.n_pairs = n_pairs;
    sw.out = P.out;
    sw.meta = pb.pair_meta;
    if (!c->deterministic) {  // (deterministic mode: no team kernels, no companion)
        sw.left_list = pb.left_list;
        sw.left_count = c->d_left + 32 * c->left_slot;
        sw.left_zero = c->d_left + 32 * (c->left_slot ^ 1);
        sw.left_expected = P.subset ? n_pairs : c->last_left;
    }
    P.sweep_info = launch_sweep(s, c->tune, c->h_cfg.n_categories, c->hellinger2, c->unit_weights, c->wf_pow, c->sweep_hint, sw);
    if (P.sweep_info & 4) c->left_slot ^= 1;  // (the record pass ran and zeroed the other slot)
    mark(c, 4);
    if (a->ev_used) { HIP_TRY(hipEventRecord(a->ev_used, s)); a->used_valid = true; }
    if (b->ev_used) { HIP_TRY(hipEventRecord(b->ev_used, s)); b->used_valid = true; }
    HIP_TRY(hipGetLastError());
    c->status_dirty = false;  // the record pass of this sequence leaves the device status clean
    P.sw = sw;
    return LCHD_OK;
}

extern "C" int lchd_from_primitives_dev_async(lchd_ctx* c, lchd_cloud* a, lchd_cloud* b, const int64_t* d_anchors,
                                              const int32_t* d_wf_index, int64_t n_pairs, double thr, double* d_out) {
    if (!c || !a || !b) return fail(LCHD_EVALUE, "null argument");
    if (!c->cfg_set) return fail(LCHD_EVALUE, "lchd_ctx_set_config has not been called");
    if (c->pend.active) return fail(LCHD_EVALUE, "a previous asynchronous call has not been finished (lchd_ctx_finish)");
    c->last_valid = false;
    if (n_pairs <= 0) return LCHD_OK;
    if (!d_anchors || !d_out) return fail(LCHD_EVALUE, "null anchor / score pointer");
    if (n_pairs > (int64_t)0x7FFFFFFF) return fail(LCHD_EUNSUPPORTED, "more than 2^31 - 1 anchor pairs in one call (%lld): split the list", (long long)n_pairs);
    if (!(thr > 0.0))  // within_radius returns nothing => dists[0] panics (src/locohd.rs:74)
        return fail(LCHD_EPANIC, "index out of bounds: threshold_distance = %g leaves every environment empty", thr);
    if (a->n == 0 || b->n == 0) return fail(LCHD_EPANIC, "index out of bounds: anchor pairs given for an empty structure");
    if (!std::isfinite(thr)) thr = 1.7e308;
    CTX_GUARD(c);
    if (int rc = resolve_bbox(c, a)) return rc;
    if (int rc = resolve_bbox(c, b)) return rc;
    auto& P = c->pend;
    P.a = a; P.b = b; P.anchors = d_anchors; P.wf = d_wf_index; P.n_pairs = n_pairs; P.thr = thr; P.out = d_out;
    P.cap = c->cap_hint;
    P.subset = false;
    c->last_store_bytes = 0;
    if (int rc = prims_enqueue(c)) return rc;
    P.active = true;
    return LCHD_OK;
}

static int grow_block(char*& blk, size_t& cap, size_t need) {
    if (need <= cap) return LCHD_OK;
    if (blk) (void)hipFree(blk);
    blk = nullptr;
    cap = 0;
    const size_t got = need + need / 4 + 4096;
    if (hipMalloc(&blk, got) != hipSuccess) {
        (void)hipGetLastError();
        return fail(LCHD_EUNSUPPORTED, "no device memory for %zu bytes of the second pass over overflowed environments", got);
    }
    cap = got;
    return LCHD_OK;
}

static int finish_passes(lchd_ctx* c, uint32_t* flags_out);

// The pass that just finished flagged environments that did not fit their slots, and they are few: instead of repeating the
// WHOLE pass with larger slots (fixed stride: every environment of the call would pay for the largest one -- one 6 000-point
// cluster in a sparse 2 10^5-point cloud turns a 1.8 GB store into 15 GB), only the pairs that touch such an environment are
// scored again: selected on the device in list order, run as a pass of their own (its own unique anchors, slots of the size the
// overflow asked for, the regular retry loop), their scores scattered over the first pass's.  The first pass gave those
// environments the anchor alone (a valid one-point environment), so every other score and status word of it stands.
// *handled = false: not applicable (too many overflowed environments, nothing recorded), the caller repeats the whole pass.
static int rescore_overflow_pairs(lchd_ctx* c, uint32_t f1, int64_t biggest, bool* handled, uint32_t* flags_out) {
    *handled = false;
    auto& P = c->pend;
    const HostStatus* h = c->h_status;
    const uint32_t na = h->n_overflow[0], nb = h->n_overflow[1];
    const uint64_t n_uniq = (uint64_t)h->n_unique[0] + h->n_unique[1];
    if (P.subset || P.per_pair || c->tune.no_overflow_subset || !P.ovf_a || na + nb == 0 || (f1 & ST_EMPTY_ENV)) return LCHD_OK;  // (per_pair: the selection kernels read side B's slot map)
    if ((uint64_t)(na + nb) * 8 > n_uniq) return LCHD_OK;  // not a minority: larger slots for everything
    hipStream_t s = c->stream;
    const SweepArgs sw = P.sw;  // the finished pass's arrays (the arena stays as it is until the second pass is enqueued)
    const bool same = sw.slot_a == sw.slot_b && sw.env_a.key == sw.env_b.key;
    // bit sets over the slots + the selection's scratch
    const size_t wa = (size_t)(P.n_slots_a + 31) / 32 + 1, wb = same ? 0 : (size_t)(P.n_slots_b + 31) / 32 + 1;
    const size_t o_cnt = (wa + wb) * 4, o_tot = o_cnt + (size_t)kOverflowWaves * 8, bits_bytes = o_tot + 8;
    if (int rc = grow_block(c->d_ovf_bits, c->ovf_bits_cap, bits_bytes)) return rc;
    uint32_t* bits_a = reinterpret_cast<uint32_t*>(c->d_ovf_bits);
    uint32_t* bits_b = same ? bits_a : bits_a + wa;
    HIP_TRY(hipMemsetAsync(c->d_ovf_bits, 0, bits_bytes, s));
    launch_mark_overflow(s, P.ovf_a, na, P.ovf_b, nb, bits_a, bits_b);
    OverflowSelect sel{};
    sel.anchors = sw.anchors; sel.n_pairs = sw.n_pairs; sel.slot_a = sw.slot_a; sel.slot_b = sw.slot_b;
    sel.bits_a = bits_a; sel.bits_b = bits_b; sel.wf = sw.wf_index;
    sel.wave_count = reinterpret_cast<unsigned long long*>(c->d_ovf_bits + o_cnt);
    unsigned long long* d_total = reinterpret_cast<unsigned long long*>(c->d_ovf_bits + o_tot);
    launch_count_overflow(s, sel, d_total);
    unsigned long long n_sub = 0;
    HIP_TRY(hipMemcpyAsync(&n_sub, d_total, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (n_sub == 0 || n_sub * 2 > (unsigned long long)sw.n_pairs) return LCHD_OK;  // (most pairs touch one: the whole pass again)
    const size_t o_anc = (size_t)n_sub * 8, o_out = o_anc + (size_t)n_sub * 16, o_wf = o_out + (size_t)n_sub * 8,
                 list_bytes = o_wf + (sw.wf_index ? (size_t)n_sub * 4 : 0);
    if (int rc = grow_block(c->d_ovf_lists, c->ovf_lists_cap, list_bytes)) return rc;
    sel.sel_index = reinterpret_cast<int64_t*>(c->d_ovf_lists);
    sel.sel_anchors = reinterpret_cast<int64_t*>(c->d_ovf_lists + o_anc);
    sel.sel_wf = sw.wf_index ? reinterpret_cast<int32_t*>(c->d_ovf_lists + o_wf) : nullptr;
    double* sub_out = reinterpret_cast<double*>(c->d_ovf_lists + o_out);
    launch_write_overflow(s, sel);
    HIP_TRY(hipGetLastError());
    // the second pass: the context's hints describe the caller's workload, not this selection -- saved and restored around it
    const auto saved = P;
    const int cap_hint = c->cap_hint, sweep_hint = c->sweep_hint, shrink_votes = c->shrink_votes;
    const bool group_small = c->group_small;
    const int64_t last_biggest = c->last_biggest;
    P.anchors = sel.sel_anchors; P.wf = sel.sel_wf; P.n_pairs = (int64_t)n_sub; P.out = sub_out;
    // (an anchor whose candidate table overflowed reported its candidates, an upper bound: a few large slots cost little here)
    P.cap = next_pow2_host(std::max<int64_t>(std::max<int64_t>(biggest, h->max_bound), saved.cap + 1));
    P.subset = true;
    c->sweep_hint = 0;
    c->last_biggest = biggest;
    ++c->n_subset_passes;
    uint32_t f2 = 0;
    int rc = prims_enqueue(c);
    if (!rc) rc = finish_passes(c, &f2);
    const SweepArgs sub_sw = P.sw;
    P = saved;
    c->cap_hint = cap_hint; c->sweep_hint = sweep_hint; c->shrink_votes = shrink_votes; c->group_small = group_small; c->last_biggest = last_biggest;
    (void)sub_sw;
    if (rc) return rc;
    launch_scatter_scores(s, sub_out, sel.sel_index, (int64_t)n_sub, saved.out);
    HIP_TRY(hipStreamSynchronize(s));
    c->last_valid = false;  // (c->last would describe the first pass, whose arena the second one has reused)
    *flags_out = (f1 & ~ST_ENV_OVERFLOW) | f2;
    *handled = true;
    return LCHD_OK;
}

// Waits for the enqueued pass and repeats it while the device asks for it (larger slots, the full launch set); *flags_out = the
// status words of the pass that stood.
static int finish_passes(lchd_ctx* c, uint32_t* flags_out) {
    auto& P = c->pend;
    for (int attempt = 0; attempt < 6; ++attempt) {
        uint32_t f = 0;
        if (int rc = wait_pass(c, &f)) return rc;
        collect_times(c, 0, 4);
        if (f & ST_BAD_ANCHOR) { *flags_out = f; return LCHD_OK; }
        const int64_t biggest = c->h_status->max_env;  // largest environment of the pass (k_pair_meta), or what overflowed
        if (f & ST_ENV_OVERFLOW) {  // an environment did not fit its slot: its pairs again with larger slots, or the whole pass
            if (biggest > 65535 && (c->h_cfg.n_categories > kMaxCategories || biggest > (1 << 23)))
                return fail(LCHD_EUNSUPPORTED, "an environment holds %lld points; beyond 65535 per environment this build handles at most %d "
                                               "categories and 2^23 points (the 64-bit-count sweep)", (long long)biggest, kMaxCategories);
            if (P.group && P.group_small && biggest <= kEnvGroupCap) {
                c->group_small = false;  // the small instantiation of k_env_group overflowed: the same capacity with the regular one
                c->last_biggest = std::max<int64_t>(biggest, kEnvGroupCapSmall + 1);  // (also when LCHD_ENV_GROUP_SMALL=1 forces the small one)
            } else {
                // The companion sweep for the larger pairs was left out (the previous pass of this context had none): if this pass has
                // some, their scores were never written -- the whole pass again with the full launch set BEFORE the second pass over
                // the overflowed environments' pairs keeps the first pass's scores of everything else.
                if ((P.sweep_info & 2) && c->h_status->n_small != ~0ull) {
                    const unsigned long long taken = (P.sweep_info & 1) ? c->h_status->n_c8 : c->h_status->n_duo;
                    if (taken < (unsigned long long)P.n_pairs) {
                        c->sweep_hint &= ~(8 | 16);
                        if (int rc = prims_enqueue(c)) return rc;
                        continue;
                    }
                }
                // what the pairs of THIS pass looked like (the second pass overwrites the mirror with its selection's numbers)
                const HostStatus first = *c->h_status;
                bool handled = false;
                if (int rc = rescore_overflow_pairs(c, f, biggest, &handled, flags_out)) return rc;
                if (handled) {
                    if (!P.subset && first.n_small != ~0ull)  // the next call of this configuration starts from the first pass's launch set
                        c->sweep_hint = 4 | (2 * first.n_duo >= (unsigned long long)P.n_pairs ? 1 : 0) |
                                        (2 * first.n_c8 >= (unsigned long long)P.n_pairs ? 2 : 0);
                    return LCHD_OK;
                }
                // (candidate-table overflows reported an upper bound -- candidates, ~2.4 environments' worth on a uniform cloud --: the whole
                //  pass tries the slot size that would fit a typical share of them first; a subset's few slots take the bound itself)
                const int64_t bound = c->h_status->max_bound;
                P.cap = next_pow2_host(std::max<int64_t>(std::max<int64_t>(biggest, P.subset ? bound : bound / 3), P.cap + 1));
                if (!P.subset) { c->cap_hint = P.cap; c->shrink_votes = 0; }
            }
            if (int rc = prims_enqueue(c)) return rc;
            continue;
        }
        // The capacity hint decays: a single dense environment should not make every later call of this context pay for
        // its slot size (slots are fixed-stride).  Eight passes in a row that would have fitted half the capacity halve it.
        if (!P.subset) {
            if (c->cap_hint > 512 && biggest > 0 && 2 * next_pow2_host(biggest) <= c->cap_hint) {
                if (++c->shrink_votes >= 8) { c->cap_hint = std::max(512, c->cap_hint / 2); c->shrink_votes = 0; }
            } else {
                c->shrink_votes = 0;
            }
        }
        if ((P.sweep_info & 2) && c->h_status->n_small != ~0ull) {
            // the companion sweep for the larger pairs was left out because the previous pass had none: did this one?
            const unsigned long long taken = (P.sweep_info & 1) ? c->h_status->n_c8 : c->h_status->n_duo;
            if (taken < (unsigned long long)P.n_pairs) {
                c->sweep_hint &= ~(8 | 16);
                if (int rc = prims_enqueue(c)) return rc;
                continue;
            }
        }
        if (biggest > 0) { c->group_small = biggest <= kEnvGroupSmallUpTo; c->last_biggest = biggest; }
        if (!P.subset) {
            if (P.per_pair) {
                ++c->per_pair_streak;
                ++c->n_per_pair_passes;
                // (this pass did not count side B's unique anchors; its bit set counts the repeated ones: a list that shares
                //  more than a fifth of them goes back to the regular pipeline with the next pass)
                if ((unsigned long long)c->h_status->n_dup_b * 5ull > (unsigned long long)P.n_pairs) c->b_use_once = false;
            } else {  // (almost) every side-B anchor unique: the next passes of this context on such lists do not de-duplicate side B
                const bool same_obj = P.sw.slot_a == P.sw.slot_b;
                c->b_use_once = !same_obj && (unsigned long long)c->h_status->n_unique[1] * 5ull >= (unsigned long long)P.n_pairs * 4ull;
                c->use_once_pairs = P.n_pairs;
                c->use_once_nb = P.b ? P.b->n : 0;
                c->per_pair_streak = 0;
            }
        }
        if (c->h_status->n_small != ~0ull && !P.subset)  // (sizes the next pass's companion launch when it walks the leftover list)
            c->last_left = P.n_pairs - (int64_t)std::min<unsigned long long>((P.sweep_info & 1) ? c->h_status->n_c8 : c->h_status->n_duo, (unsigned long long)P.n_pairs);
        if (c->h_status->n_small != ~0ull)  // what the pairs looked like this time picks the sweep kernels of the next pass of this configuration
            c->sweep_hint = 4 | (2 * c->h_status->n_duo >= (unsigned long long)P.n_pairs ? 1 : 0) |
                            (2 * c->h_status->n_c8 >= (unsigned long long)P.n_pairs ? 2 : 0) |
                            (c->h_status->n_duo == (unsigned long long)P.n_pairs ? 8 : 0) | (c->h_status->n_c8 == (unsigned long long)P.n_pairs ? 16 : 0);
        c->last = P.sw;
        c->last_valid = true;
        *flags_out = f;
        return LCHD_OK;
    }
    return fail(LCHD_EDEVICE, "environment capacity retry did not converge");
}

extern "C" int lchd_ctx_finish(lchd_ctx* c) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    auto& P = c->pend;
    if (!P.active) return LCHD_OK;
    P.active = false;
    CTX_GUARD(c);
    uint32_t f = 0;
    if (int rc = finish_passes(c, &f)) return rc;
    return status_to_rc(f, DRV_PRIMS);
}

extern "C" int64_t lchd_ctx_subset_pass_count(lchd_ctx* c) { return c ? c->n_subset_passes : -1; }
extern "C" int64_t lchd_ctx_per_pair_pass_count(lchd_ctx* c) { return c ? c->n_per_pair_passes : -1; }
extern "C" int64_t lchd_ctx_last_store_bytes(lchd_ctx* c) { return c ? (int64_t)c->last_store_bytes : -1; }

extern "C" int lchd_from_primitives_dev(lchd_ctx* c, lchd_cloud* a, lchd_cloud* b, const int64_t* d_anchors,
                                        const int32_t* d_wf_index, int64_t n_pairs, double thr, double* d_out) {
    if (int rc = lchd_from_primitives_dev_async(c, a, b, d_anchors, d_wf_index, n_pairs, thr, d_out)) return rc;
    return lchd_ctx_finish(c);
}

// ------------------------------------------------------------------------------------------------
// trajectory frames: a batch cloud whose structures are frames of one template structure
// ------------------------------------------------------------------------------------------------
extern "C" int lchd_frames_create(lchd_ctx* c, const lchd_cloud* tmpl, int32_t capacity_frames, lchd_cloud** out) {
    if (!c || !tmpl || !out || capacity_frames < 1) return fail(LCHD_EVALUE, "bad argument");
    CTX_GUARD(c);
    if (tmpl->sid) return fail(LCHD_EVALUE, "the template of a frames buffer must be a single structure");
    const int64_t nt = tmpl->n, total = nt * capacity_frames;
    if (nt < 1 || total > ((int64_t)1 << 30)) return fail(LCHD_EUNSUPPORTED, "frames buffer of %lld atoms is out of range", (long long)total);
    lchd_cloud* cl = new lchd_cloud();
    cl->n_tmpl = nt;
    cl->struct_size = (int32_t)nt;
    cl->cap_frames = capacity_frames;
    cl->n = 0;
    cl->n_struct = 1;
    auto bail = [&](hipError_t e, const char* what) {
        lchd_cloud_destroy(c, cl);
        return fail(LCHD_EDEVICE, "HIP error %d in %s", (int)e, what);
    };
    hipError_t e;
    if ((e = hipMalloc(&cl->x, sizeof(double) * total)) != hipSuccess) return bail(e, "hipMalloc(x)");
    if ((e = hipMalloc(&cl->y, sizeof(double) * total)) != hipSuccess) return bail(e, "hipMalloc(y)");
    if ((e = hipMalloc(&cl->z, sizeof(double) * total)) != hipSuccess) return bail(e, "hipMalloc(z)");
    if ((e = hipMalloc(&cl->cat, total)) != hipSuccess) return bail(e, "hipMalloc(cat)");
    if ((e = hipMalloc(&cl->tag, sizeof(int32_t) * total)) != hipSuccess) return bail(e, "hipMalloc(tag)");
    if ((e = hipMalloc(&cl->sid, sizeof(int32_t) * total)) != hipSuccess) return bail(e, "hipMalloc(sid)");
    if ((e = hipMalloc(&cl->d_raw, sizeof(double) * 3 * total)) != hipSuccess) return bail(e, "hipMalloc(raw)");
    if ((e = hipMalloc(&cl->d_bbox, sizeof(unsigned long long) * 7)) != hipSuccess) return bail(e, "hipMalloc(bbox)");
    if ((e = hipHostMalloc(&cl->h_pinned, sizeof(double) * 3 * total)) != hipSuccess) return bail(e, "hipHostMalloc");
    if ((e = hipEventCreateWithFlags(&cl->ev_ready, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&cl->ev_used, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    launch_frames_labels(c->stream, tmpl->cat, tmpl->tag, nt, capacity_frames, cl->cat, cl->tag, cl->sid);
    if (tmpl->cat_hi) {  // more than 255 categories: the high bytes and the one-byte view travel with every frame as well
        if ((e = hipMalloc(&cl->cat_hi, total)) != hipSuccess) return bail(e, "hipMalloc(cat_hi)");
        if ((e = hipMalloc(&cl->cat_narrow, total)) != hipSuccess) return bail(e, "hipMalloc(cat_narrow)");
        launch_frames_labels(c->stream, tmpl->cat_hi, tmpl->tag, nt, capacity_frames, cl->cat_hi, cl->tag, cl->sid);
        launch_frames_labels(c->stream, tmpl->cat_narrow, tmpl->tag, nt, capacity_frames, cl->cat_narrow, cl->tag, cl->sid);
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail(e, "frames label fill");
    *out = cl;
    return LCHD_OK;
}

extern "C" int lchd_frames_load(lchd_ctx* c, lchd_cloud* fr, const double* xyz, int32_t n_frames, void* hip_stream) {
    if (!c || !fr || !xyz || !fr->cap_frames) return fail(LCHD_EVALUE, "not a frames buffer");
    if (n_frames < 1 || n_frames > fr->cap_frames) return fail(LCHD_EVALUE, "%d frames do not fit a buffer of %d", n_frames, fr->cap_frames);
    if (c->pend.active && (c->pend.a == fr || c->pend.b == fr))
        return fail(LCHD_EVALUE, "this frames buffer is in use by an unfinished asynchronous call");
    CTX_GUARD(c);
    hipStream_t s = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->stream;
    const int64_t total = fr->n_tmpl * n_frames;
    // the pinned staging block is free again once the previous upload from it has completed
    if (fr->ev_ready && fr->bbox_pending) HIP_TRY(hipEventSynchronize(fr->ev_ready));
    memcpy(fr->h_pinned, xyz, sizeof(double) * 3 * (size_t)total);
    if (fr->used_valid) HIP_TRY(hipStreamWaitEvent(s, fr->ev_used, 0));  // do not overwrite frames a running pass still reads
    HIP_TRY(hipMemcpyAsync(fr->d_raw, fr->h_pinned, sizeof(double) * 3 * (size_t)total, hipMemcpyHostToDevice, s));
    launch_frames_unpack(s, fr->d_raw, total, fr->x, fr->y, fr->z, fr->d_bbox);
    HIP_TRY(hipEventRecord(fr->ev_ready, s));
    HIP_TRY(hipGetLastError());
    fr->n = total;
    fr->n_struct = n_frames;
    fr->bbox_pending = true;
    return LCHD_OK;
}

extern "C" int lchd_frames_set_sources(lchd_ctx* c, lchd_cloud* fr, const int32_t* src_start, const int32_t* src_idx,
                                       int64_t n_src_atoms) {
    if (!c || !fr || !fr->cap_frames || !src_start || !src_idx) return fail(LCHD_EVALUE, "not a frames buffer / null map");
    if (c->pend.active && (c->pend.a == fr || c->pend.b == fr))
        return fail(LCHD_EVALUE, "this frames buffer is in use by an unfinished asynchronous call");
    CTX_GUARD(c);
    const int64_t np = fr->n_tmpl;
    if (n_src_atoms < 1 || n_src_atoms * (int64_t)fr->cap_frames > ((int64_t)1 << 31))
        return fail(LCHD_EUNSUPPORTED, "%lld source atoms x %d frames is out of range", (long long)n_src_atoms, fr->cap_frames);
    if (src_start[0] != 0) return fail(LCHD_EVALUE, "src_start[0] must be 0");
    for (int64_t p = 0; p < np; ++p)  // np.mean of an empty list is NaN in the reference (atom_converter_utils.py:126): refuse it here
        if (src_start[p + 1] <= src_start[p]) return fail(LCHD_EVALUE, "primitive atom %lld has no source atoms", (long long)p);
    const int64_t nnz = src_start[np];
    for (int64_t k = 0; k < nnz; ++k)
        if (src_idx[k] < 0 || src_idx[k] >= n_src_atoms)
            return fail(LCHD_EVALUE, "source atom index %d at position %lld is outside [0, %lld)", src_idx[k], (long long)k, (long long)n_src_atoms);
    // Tiles: consecutive primitive atoms whose members span at most `span` consecutive source atoms (staged through LDS);
    // spans are balanced so that the tiles of a frame carry similar byte counts.  A primitive atom that alone spans more
    // becomes a tile of its own that gathers from global memory.
    std::vector<int32_t> tiles;
    {
        const int cap = centroid_tile_span();
        int64_t glo = n_src_atoms, ghi = 0;
        for (int64_t k = 0; k < nnz; ++k) { glo = std::min<int64_t>(glo, src_idx[k]); ghi = std::max<int64_t>(ghi, src_idx[k] + 1); }
        const int64_t pieces = std::max<int64_t>(1, (ghi - glo + cap - 1) / cap);
        const int span = (int)std::min<int64_t>(cap, (ghi - glo + pieces - 1) / pieces + 64);
        int64_t p0 = 0;
        int lo = 0, hi = 0;
        auto close = [&](int64_t p1) { if (p1 > p0) { tiles.push_back((int32_t)p0); tiles.push_back((int32_t)p1); tiles.push_back(lo); tiles.push_back(hi); } };
        for (int64_t p = 0; p < np; ++p) {
            int plo = src_idx[src_start[p]], phi = plo + 1;
            for (int k = src_start[p]; k < src_start[p + 1]; ++k) { plo = std::min(plo, src_idx[k]); phi = std::max(phi, src_idx[k] + 1); }
            if (phi - plo > span) {  // cannot be staged: its own global-gather tile
                close(p);
                tiles.push_back((int32_t)p); tiles.push_back((int32_t)p + 1); tiles.push_back(-1); tiles.push_back(-1);
                p0 = p + 1;
                continue;
            }
            if (p == p0) { lo = plo; hi = phi; continue; }
            const int nlo = std::min(lo, plo), nhi = std::max(hi, phi);
            if (nhi - nlo > span) { close(p); p0 = p; lo = plo; hi = phi; }
            else { lo = nlo; hi = nhi; }
        }
        close(np);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (fr->ev_ready && fr->bbox_pending) HIP_TRY(hipEventSynchronize(fr->ev_ready));
    (void)hipFree(fr->d_src_start); fr->d_src_start = nullptr;
    (void)hipFree(fr->d_src_idx); fr->d_src_idx = nullptr;
    (void)hipFree(fr->d_tiles); fr->d_tiles = nullptr;
    (void)hipFree(fr->d_raw32); fr->d_raw32 = nullptr;
    if (fr->h_pinned32) { (void)hipHostFree(fr->h_pinned32); fr->h_pinned32 = nullptr; }
    fr->n_src = 0;
    const size_t raw_elems = (size_t)3 * (size_t)n_src_atoms * (size_t)fr->cap_frames;
    {   // all or nothing: a failure leaves the buffer without sources (n_src == 0) and without half-made allocations
        hipError_t e = hipMalloc(&fr->d_src_start, sizeof(int32_t) * (size_t)(np + 1));
        if (e == hipSuccess) e = hipMalloc(&fr->d_src_idx, sizeof(int32_t) * (size_t)nnz);
        if (e == hipSuccess) e = hipMalloc(&fr->d_raw32, sizeof(float) * raw_elems);
        if (e == hipSuccess) e = hipHostMalloc(&fr->h_pinned32, sizeof(float) * raw_elems);
        if (e == hipSuccess) e = hipMemcpy(fr->d_src_start, src_start, sizeof(int32_t) * (size_t)(np + 1), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(fr->d_src_idx, src_idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice);
        if (e == hipSuccess && !fr->d_bbox_part) e = hipMalloc(&fr->d_bbox_part, sizeof(unsigned long long) * 7 * (size_t)bbox_parts_capacity());
        if (e == hipSuccess) e = hipMalloc(&fr->d_tiles, sizeof(int32_t) * tiles.size());
        if (e == hipSuccess) e = hipMemcpy(fr->d_tiles, tiles.data(), sizeof(int32_t) * tiles.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(fr->d_src_start); fr->d_src_start = nullptr;
            (void)hipFree(fr->d_src_idx); fr->d_src_idx = nullptr;
            (void)hipFree(fr->d_tiles); fr->d_tiles = nullptr;
            (void)hipFree(fr->d_raw32); fr->d_raw32 = nullptr;
            if (fr->h_pinned32) { (void)hipHostFree(fr->h_pinned32); fr->h_pinned32 = nullptr; }
            return fail(LCHD_EDEVICE, "HIP error %d (%s) in lchd_frames_set_sources", (int)e, hipGetErrorName(e));
        }
    }
    fr->n_tiles = (int32_t)(tiles.size() / 4);
    fr->n_src = n_src_atoms;
    return LCHD_OK;
}

// host_src: copy through the pinned block first; otherwise `atom_xyz` is already a DEVICE pointer
static int frames_load_atoms(lchd_ctx* c, lchd_cloud* fr, const float* atom_xyz, int32_t n_frames, void* hip_stream, bool host_src) {
    if (!c || !fr || !atom_xyz || !fr->cap_frames) return fail(LCHD_EVALUE, "not a frames buffer");
    if (!fr->n_src) return fail(LCHD_EVALUE, "lchd_frames_set_sources has not been called on this frames buffer");
    if (n_frames < 1 || n_frames > fr->cap_frames) return fail(LCHD_EVALUE, "%d frames do not fit a buffer of %d", n_frames, fr->cap_frames);
    if (c->pend.active && (c->pend.a == fr || c->pend.b == fr))
        return fail(LCHD_EVALUE, "this frames buffer is in use by an unfinished asynchronous call");
    CTX_GUARD(c);
    hipStream_t s = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->stream;
    const size_t elems = (size_t)3 * (size_t)fr->n_src * (size_t)n_frames;
    const float* d_src = atom_xyz;
    if (host_src) {
        if (fr->ev_ready && fr->bbox_pending) HIP_TRY(hipEventSynchronize(fr->ev_ready));  // pinned block free again
        memcpy(fr->h_pinned32, atom_xyz, sizeof(float) * elems);
    }
    if (fr->used_valid) HIP_TRY(hipStreamWaitEvent(s, fr->ev_used, 0));
    if (host_src) {
        HIP_TRY(hipMemcpyAsync(fr->d_raw32, fr->h_pinned32, sizeof(float) * elems, hipMemcpyHostToDevice, s));
        d_src = fr->d_raw32;
    }
    fr->t_valid = false;
    if (c->timing) {
        if (!fr->ev_t0) { HIP_TRY(hipEventCreate(&fr->ev_t0)); HIP_TRY(hipEventCreate(&fr->ev_t1)); }
        HIP_TRY(hipEventRecord(fr->ev_t0, s));
    }
    launch_frames_centroids(s, d_src, fr->n_src, fr->d_src_start, fr->d_src_idx, fr->d_tiles, fr->n_tiles, fr->n_tmpl, n_frames, fr->x,
                            fr->y, fr->z, fr->d_bbox, fr->d_bbox_part);
    if (c->timing) { HIP_TRY(hipEventRecord(fr->ev_t1, s)); fr->t_valid = true; }
    HIP_TRY(hipEventRecord(fr->ev_ready, s));
    HIP_TRY(hipGetLastError());
    fr->n = fr->n_tmpl * n_frames;
    fr->n_struct = n_frames;
    fr->bbox_pending = true;
    return LCHD_OK;
}

extern "C" int lchd_frames_load_atoms(lchd_ctx* c, lchd_cloud* fr, const float* atom_xyz, int32_t n_frames, void* hip_stream) {
    return frames_load_atoms(c, fr, atom_xyz, n_frames, hip_stream, true);
}
extern "C" int lchd_frames_load_atoms_dev(lchd_ctx* c, lchd_cloud* fr, const float* d_atom_xyz, int32_t n_frames, void* hip_stream) {
    return frames_load_atoms(c, fr, d_atom_xyz, n_frames, hip_stream, false);
}
extern "C" double lchd_frames_last_convert_ms(lchd_ctx* c, lchd_cloud* fr) {
    if (!c || !fr || !fr->t_valid) return -1.0;
    CTX_GUARD(c);
    float t = -1.f;
    if (hipEventSynchronize(fr->ev_t1) != hipSuccess || hipEventElapsedTime(&t, fr->ev_t0, fr->ev_t1) != hipSuccess) return -1.0;
    return t;
}

/* Read back the primitive-atom coordinates of a frames buffer (or any cloud) as [n][3] f64: lets a caller check the device
 * centroids against its own np.mean, and feeds generate_primitive_pdb for a frame. */
extern "C" int lchd_cloud_get_coords(lchd_ctx* c, lchd_cloud* cl, double* xyz_out, int64_t n) {
    if (!c || !cl || !xyz_out) return fail(LCHD_EVALUE, "null argument");
    if (n != cl->n) return fail(LCHD_EVALUE, "the cloud holds %lld atoms, not %lld", (long long)cl->n, (long long)n);
    CTX_GUARD(c);
    if (cl->ev_ready && cl->bbox_pending) HIP_TRY(hipEventSynchronize(cl->ev_ready));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<double> soa((size_t)3 * (size_t)n);
    if (n) {
        HIP_TRY(hipMemcpy(soa.data(), cl->x, sizeof(double) * n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(soa.data() + n, cl->y, sizeof(double) * n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(soa.data() + 2 * n, cl->z, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) xyz_out[3 * i + k] = soa[(size_t)k * n + i];
    return LCHD_OK;
}

// ------------------------------------------------------------------------------------------------
// host-pointer drivers
// ------------------------------------------------------------------------------------------------
static int check_wf_index(const lchd_config* cfg, const int32_t* wf_index, int64_t n) {
    if (!wf_index) return LCHD_OK;
    for (int64_t i = 0; i < n; ++i)
        if (wf_index[i] < 0 || wf_index[i] >= cfg->n_weight_functions)
            return fail(LCHD_EVALUE, "weight-function index %d out of range at position %lld", wf_index[i], (long long)i);
    return LCHD_OK;
}

// One structure of a host-pointer call inside the context's I/O block: SoA coordinates, tags, categories.  Fills the pinned
// staging copy (AoS -> SoA, finiteness check, bounding box) and points a stack-allocated lchd_cloud at the device copy.
static int stage_cloud(const double* xyz, const int32_t* cat, const int32_t* tag, int64_t n, char* h_base, char* d_base, size_t& off,
                       lchd_cloud& cl, bool allow_wide) {
    auto take = [&](size_t bytes) { off = (off + 255) & ~size_t(255); const size_t o = off; off += bytes; return o; };
    const size_t m = (size_t)std::max<int64_t>(n, 1);
    const size_t ox = take(8 * m), oy = take(8 * m), oz = take(8 * m), ot = take(4 * m), oc = take(m), och = take(m);
    cl.x = reinterpret_cast<double*>(d_base + ox);
    cl.y = reinterpret_cast<double*>(d_base + oy);
    cl.z = reinterpret_cast<double*>(d_base + oz);
    cl.tag = reinterpret_cast<int32_t*>(d_base + ot);
    cl.cat = reinterpret_cast<uint8_t*>(d_base + oc);
    cl.cat_hi = nullptr;
    cl.n = n;
    if (!h_base) return LCHD_OK;  // sizing pass
    // (two-byte ids only under a configuration with more than 255 categories: otherwise an id beyond 254 is simply not in the map)
    const bool wide = allow_wide && cats_need_hi(cat, n);
    if (wide) cl.cat_hi = reinterpret_cast<uint8_t*>(d_base + och);
    double *hx = reinterpret_cast<double*>(h_base + ox), *hy = reinterpret_cast<double*>(h_base + oy), *hz = reinterpret_cast<double*>(h_base + oz);
    int32_t* ht = reinterpret_cast<int32_t*>(h_base + ot);
    uint8_t* hc = reinterpret_cast<uint8_t*>(h_base + oc);
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; ++i) {
        const double v[3] = {xyz ? xyz[3 * i] : 0.0, xyz ? xyz[3 * i + 1] : 0.0, xyz ? xyz[3 * i + 2] : 0.0};  // (no coordinates: given distance rows)
        if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2])) return fail(LCHD_EVALUE, "non-finite coordinate at atom %lld", (long long)i);
        hx[i] = v[0]; hy[i] = v[1]; hz[i] = v[2];
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], v[k]); mx[k] = std::max(mx[k], v[k]); }
        ht[i] = tag ? tag[i] : 0;
    }
    cats_encode(cat, n, hc, wide ? reinterpret_cast<uint8_t*>(h_base + och) : nullptr);
    for (int k = 0; k < 3; ++k) { cl.bbmin[k] = n ? mn[k] : 0.0; cl.bbmax[k] = n ? mx[k] : 0.0; }
    return LCHD_OK;
}

// Scores of a host-pointer call with at most this many pairs are stored by the sweep kernels straight into the pinned,
// device-visible staging block (8 bytes per pair over the host link): no device-to-host copy operation behind the pass.
constexpr int64_t kDirectOutPairs = 1 << 16;

static int grow_io(lchd_ctx* c, size_t total) {
    if (total <= c->io_cap) return LCHD_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_io); c->d_io = nullptr;
    if (c->h_io) { (void)hipHostFree(c->h_io); c->h_io = nullptr; }
    c->io_cap = 0;
    const size_t want = total + total / 4 + (1 << 16);
    HIP_TRY(hipMalloc(&c->d_io, want));
    hipError_t e = hipHostMalloc(&c->h_io, want);
    if (e != hipSuccess) {
        (void)hipFree(c->d_io); c->d_io = nullptr; c->h_io = nullptr;
        return fail(LCHD_EDEVICE, "HIP error %d (%s) in hipHostMalloc of the staging block", (int)e, hipGetErrorName(e));
    }
    c->io_cap = want;
    return LCHD_OK;
}

// One host-pointer from_primitives call on one context, split so that a device group can have every device's pass in flight
// at the same time.  `subset` (or nullptr = all pairs) lists the positions in anchors / wf_index / out this context scores.
// Large host-pointer calls (a million pairs: 16 MB of anchors in, 8 MB of scores out) spend more time copying between the caller's
// pageable arrays and the pinned staging block than the GPU spends scoring them.  Above kPipePairs the pair list therefore travels
// in chunks -- chunk k + 1 is copied into the staging block (by up to four threads: one core moves ~10 GB/s, the host link ~50) while
// the DMA engine sends chunk k -- and the scores come back the same way.
constexpr int64_t kPipePairs = (int64_t)1 << 18;
static void parallel_copy(void* dst, const void* src, size_t bytes) {
    const int nt = bytes >= ((size_t)6 << 20) ? 4 : (bytes >= ((size_t)2 << 20) ? 2 : 1);
    if (nt == 1) { memcpy(dst, src, bytes); return; }
    const size_t part = ((bytes / nt) + 4095) & ~(size_t)4095;
    std::thread th[3];
    int started = 0;
    for (int k = 1; k < nt; ++k) {
        const size_t o = (size_t)k * part;
        if (o >= bytes) break;
        const size_t len = std::min(part, bytes - o);
        try {
            th[started] = std::thread([=] { memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, len); });
            ++started;
        } catch (...) {  // no thread to be had: this part on the calling thread
            memcpy(static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, len);
        }
    }
    memcpy(dst, src, std::min(part, bytes));
    for (int k = 0; k < started; ++k) th[k].join();
}

struct HostCall {
    lchd_cloud a, b;  // point into the context's device I/O block; referenced by the pending pass until it is finished
    size_t o_out = 0;
    bool direct = false;
    int64_t n = 0;
    bool enqueued = false;
};
static int host_call_enqueue(lchd_ctx* c, const lchd_config* cfg, const double* xyz_a, const int32_t* cat_a, const int32_t* tag_a, int64_t n_a,
                             const double* xyz_b, const int32_t* cat_b, const int32_t* tag_b, int64_t n_b, const int64_t* anchors,
                             const int32_t* wf_index, const int64_t* subset, int64_t n, double thr, HostCall& hc) {
    hc.n = n;
    hc.enqueued = false;
    if (c->pend.active) return fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)");
    CTX_GUARD(c);
    if (int rc = lchd_ctx_set_config(c, cfg)) return rc;
    if (n == 0) return LCHD_OK;
    // Everything a call sends to the device travels as ONE block through pinned staging and ONE asynchronous copy; the block
    // and its staging twin belong to the context and only ever grow (the reference clones its arguments per call as well,
    // primitive_atom.rs:5, but a device allocation costs far more than a Vec).
    size_t o_anchors = 0, o_wf = 0, in_bytes = 0, total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        size_t off = 0;
        char* hb = pass ? c->h_io : nullptr;
        if (int rc = stage_cloud(xyz_a, cat_a, tag_a, n_a, hb, c->d_io, off, hc.a, cfg->n_categories > kMaxCategories)) return rc;
        if (int rc = stage_cloud(xyz_b, cat_b, tag_b, n_b, hb, c->d_io, off, hc.b, cfg->n_categories > kMaxCategories)) return rc;
        auto take = [&](size_t bytes) { off = (off + 255) & ~size_t(255); const size_t o = off; off += bytes; return o; };
        o_anchors = take(sizeof(int64_t) * 2 * (size_t)n);
        o_wf = take(wf_index ? sizeof(int32_t) * (size_t)n : 0);
        in_bytes = off;
        hc.o_out = take(sizeof(double) * (size_t)n);
        total = off;
        if (pass == 0)
            if (int rc = grow_io(c, total)) return rc;
    }
    int64_t* ha = reinterpret_cast<int64_t*>(c->h_io + o_anchors);
    int32_t* hw = reinterpret_cast<int32_t*>(c->h_io + o_wf);
    bool piped = false;
    if (!subset && n >= kPipePairs) {
        // the structures (and whatever lies in front of the pair list) first, then the list chunk by chunk: the copy of chunk k + 1
        // into the staging block overlaps the DMA of chunk k
        HIP_TRY(hipMemcpyAsync(c->d_io, c->h_io, o_anchors, hipMemcpyHostToDevice, c->stream));
        const int64_t chunk = std::max<int64_t>(kPipePairs, (n + 3) / 4);
        for (int64_t p0 = 0; p0 < n; p0 += chunk) {
            const size_t o = o_anchors + sizeof(int64_t) * 2 * (size_t)p0, len = sizeof(int64_t) * 2 * (size_t)std::min<int64_t>(chunk, n - p0);
            parallel_copy(c->h_io + o, reinterpret_cast<const char*>(anchors) + (o - o_anchors), len);
            HIP_TRY(hipMemcpyAsync(c->d_io + o, c->h_io + o, len, hipMemcpyHostToDevice, c->stream));
        }
        if (wf_index) {
            parallel_copy(hw, wf_index, sizeof(int32_t) * (size_t)n);
            HIP_TRY(hipMemcpyAsync(c->d_io + o_wf, c->h_io + o_wf, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        }
        piped = true;
    } else if (!subset) {
        memcpy(ha, anchors, sizeof(int64_t) * 2 * (size_t)n);
        if (wf_index) memcpy(hw, wf_index, sizeof(int32_t) * (size_t)n);
    } else {
        for (int64_t k = 0; k < n; ++k) {
            ha[2 * k] = anchors[2 * subset[k]];
            ha[2 * k + 1] = anchors[2 * subset[k] + 1];
            if (wf_index) hw[k] = wf_index[subset[k]];
        }
    }
    if (!piped) HIP_TRY(hipMemcpyAsync(c->d_io, c->h_io, in_bytes, hipMemcpyHostToDevice, c->stream));
    const int64_t* d_anchors = reinterpret_cast<const int64_t*>(c->d_io + o_anchors);
    const int32_t* d_wf = wf_index ? reinterpret_cast<const int32_t*>(c->d_io + o_wf) : nullptr;
    hc.direct = n <= kDirectOutPairs;
    double* d_out = reinterpret_cast<double*>((hc.direct ? c->h_io : c->d_io) + hc.o_out);
    if (int rc = lchd_from_primitives_dev_async(c, &hc.a, &hc.b, d_anchors, d_wf, n, thr, d_out)) return rc;
    hc.enqueued = true;
    return LCHD_OK;
}
static int host_call_finish(lchd_ctx* c, HostCall& hc, const int64_t* subset, double* out) {
    if (!hc.enqueued) return LCHD_OK;
    hc.enqueued = false;
    CTX_GUARD(c);
    int rc = lchd_ctx_finish(c);
    if (!rc && !hc.direct && !subset && hc.n >= kPipePairs) {
        // the scores come back in chunks: chunk k is copied out to the caller's array while the DMA engine fetches chunk k + 1
        const int64_t chunk = std::max<int64_t>(kPipePairs, (hc.n + 3) / 4);
        hipError_t e = hipSuccess;
        int n_ev = 0;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int64_t p0 = 0; p0 < hc.n && e == hipSuccess; p0 += chunk, ++n_ev) {
            const size_t o = hc.o_out + sizeof(double) * (size_t)p0, len = sizeof(double) * (size_t)std::min<int64_t>(chunk, hc.n - p0);
            e = hipMemcpyAsync(c->h_io + o, c->d_io + o, len, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[n_ev], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ev[n_ev], c->stream);
        }
        int k = 0;
        for (int64_t p0 = 0; p0 < hc.n && e == hipSuccess; p0 += chunk, ++k) {
            e = hipEventSynchronize(ev[k]);
            if (e == hipSuccess)
                parallel_copy(out + p0, c->h_io + hc.o_out + sizeof(double) * (size_t)p0, sizeof(double) * (size_t)std::min<int64_t>(chunk, hc.n - p0));
        }
        for (int q = 0; q < 4; ++q)
            if (ev[q]) (void)hipEventDestroy(ev[q]);
        if (e != hipSuccess) { (void)hipStreamSynchronize(c->stream); rc = fail(LCHD_EDEVICE, "HIP error %d in D2H scores", (int)e); }
        c->last_valid = false;
        c->pend.a = c->pend.b = nullptr;
        return rc;
    }
    if (!rc && !hc.direct) {
        hipError_t e = hipMemcpyAsync(c->h_io + hc.o_out, c->d_io + hc.o_out, sizeof(double) * (size_t)hc.n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(LCHD_EDEVICE, "HIP error %d in D2H scores", (int)e);
    }
    if (!rc) {
        const double* h = reinterpret_cast<const double*>(c->h_io + hc.o_out);
        if (!subset) memcpy(out, h, sizeof(double) * (size_t)hc.n);
        else for (int64_t k = 0; k < hc.n; ++k) out[subset[k]] = h[k];
    }
    c->last_valid = false;  // the anchors of this call live in the I/O block, which the next call overwrites
    c->pend.a = c->pend.b = nullptr;  // the clouds of the call are gone
    return rc;
}

extern "C" int lchd_from_primitives(lchd_ctx* c, const lchd_config* cfg, const double* xyz_a, const int32_t* cat_a,
                                    const int32_t* tag_a, int64_t n_a, const double* xyz_b, const int32_t* cat_b,
                                    const int32_t* tag_b, int64_t n_b, const int64_t* anchors, const int32_t* wf_index,
                                    int64_t n_pairs, double thr, double* out) {
    if (!c || !cfg) return fail(LCHD_EVALUE, "null context / configuration");
    if (int rc = check_wf_index(cfg, wf_index, n_pairs)) return rc;
    if (n_a < 0 || n_b < 0 || n_a > ((int64_t)1 << 30) || n_b > ((int64_t)1 << 30)) return fail(LCHD_EUNSUPPORTED, "structure size out of range");
    HostCall hc;
    if (int rc = host_call_enqueue(c, cfg, xyz_a, cat_a, tag_a, n_a, xyz_b, cat_b, tag_b, n_b, anchors, wf_index, nullptr, n_pairs, thr, hc)) return rc;
    return host_call_finish(c, hc, nullptr, out);
}

// ------------------------------------------------------------------------------------------------
// Sharding of an anchor-pair list by side-A anchor: the rule of lchd_kernels.hip (k_shard_plan), on the host
// ------------------------------------------------------------------------------------------------
// key side 0: bins of the side-A anchor; 1: of the side-B anchor (side A's partition is unbalanced: a rank would hold more than
// 1.25 P / world + 1 pairs -- one reference anchor against thousands, python_codes/kras_scan.py:46-52); 2: contiguous slices
static int shard_rule_host(const int64_t* anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int world, std::vector<uint16_t>& rank_of_bin) {
    auto bin_of = [](int64_t a, int64_t n) { a = a < 0 ? 0 : (a >= n ? n - 1 : a); return (int)((a * kShardBins) / n); };
    std::vector<uint64_t> ha(kShardBins, 0), hb(kShardBins, 0);
    for (int64_t p = 0; p < n_pairs; ++p) {
        ++ha[bin_of(anchors[2 * p], n_atoms_a)];
        if (n_atoms_b > 0) ++hb[bin_of(anchors[2 * p + 1], n_atoms_b)];
    }
    auto plan_side = [&](const std::vector<uint64_t>& hist) {  // fills rank_of_bin; true: balanced
        std::vector<uint64_t> cnt((size_t)world, 0);
        rank_of_bin.assign(kShardBins, 0);
        uint64_t pre = 0;
        for (int b = 0; b < kShardBins; ++b) {
            const uint64_t r = std::min<uint64_t>(n_pairs > 0 ? (pre * (uint64_t)world) / (uint64_t)n_pairs : 0, (uint64_t)world - 1);
            rank_of_bin[b] = (uint16_t)r;
            cnt[r] += hist[b];
            pre += hist[b];
        }
        return *std::max_element(cnt.begin(), cnt.end()) * 4ull * (uint64_t)world <= 5ull * (uint64_t)n_pairs + 4ull * (uint64_t)world;
    };
    if (plan_side(ha)) return 0;
    if (n_atoms_b > 0 && plan_side(hb)) return 1;
    return 2;
}

static int shard_plan_enqueue(lchd_ctx* c, const int64_t* d_anchors, const int64_t* h_src, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                              int32_t world);
static int shard_plan_wait(lchd_ctx* c, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int32_t world, int64_t* counts_out);

struct lchd_group {
    std::vector<lchd_ctx*> ctx;
    std::vector<int64_t> last_counts;
    // the caller's pair list and the scores in the caller's order: pinned, visible to every device of the group (grow-only)
    int64_t* h_list = nullptr;
    double* h_out = nullptr;
    char* h_stage = nullptr;  // both structures, staged ONCE per call (SoA, categories, tags): every device copies from here
    size_t list_cap = 0, out_cap = 0, stage_cap = 0;
};

extern "C" int lchd_group_create(const int32_t* devices, int32_t n_devices, lchd_group** out) {
    if (!devices || !out || n_devices < 1 || n_devices > kShardMaxWorld) return fail(LCHD_EVALUE, "a device group needs 1..%d devices", kShardMaxWorld);
    *out = nullptr;
    lchd_group* g = new lchd_group();
    for (int k = 0; k < n_devices; ++k) {
        lchd_ctx* c = nullptr;
        if (int rc = lchd_ctx_create(devices[k], &c)) { lchd_group_destroy(g); return rc; }
        g->ctx.push_back(c);
    }
    g->last_counts.assign((size_t)n_devices, 0);
    *out = g;
    return LCHD_OK;
}
extern "C" void lchd_group_destroy(lchd_group* g) {
    if (!g) return;
    for (lchd_ctx* c : g->ctx) lchd_ctx_destroy(c);
    if (g->h_list) (void)hipHostFree(g->h_list);
    if (g->h_out) (void)hipHostFree(g->h_out);
    if (g->h_stage) (void)hipHostFree(g->h_stage);
    delete g;
}

// Device-side partition of a group call (no per-pair work on the calling thread): the pair list goes ONCE into a pinned block that
// every device copies from; every device plans the partition itself (k_shard_plan: the same pure function of the list on all of
// them), selects its share (k_shard_select), scores it, and writes its scores straight to their positions in ONE pinned, host-mapped
// score block (k_scatter_scores); the caller gets one bulk copy of that block.  The calling thread touches the list twice (copy in,
// copy out) whatever the number of devices; the host-side partition it replaces walked the list once per step of: histogram, W index
// lists, gather into W staging blocks, scatter of the scores -- ~5 ns per pair that did not shrink with W.
struct GroupDev {
    lchd_cloud a, b;
    size_t o_list = 0, o_sel = 0, o_idx = 0, o_sc = 0;
    int64_t n_mine = 0;
    bool planned = false, enqueued = false;
};
static int group_call_device_partition(lchd_group* g, const lchd_config* cfg, const double* xyz_a, const int32_t* cat_a, const int32_t* tag_a,
                                       int64_t n_a, const double* xyz_b, const int32_t* cat_b, const int32_t* tag_b, int64_t n_b,
                                       const int64_t* anchors, int64_t n_pairs, double thr, double* out) {
    const int world = (int)g->ctx.size();
    const size_t list_bytes = sizeof(int64_t) * 2 * (size_t)n_pairs, out_bytes = sizeof(double) * (size_t)n_pairs;
    if (g->list_cap < list_bytes) {
        if (g->h_list) (void)hipHostFree(g->h_list);
        g->h_list = nullptr; g->list_cap = 0;
        HIP_TRY(hipHostMalloc(&g->h_list, list_bytes + list_bytes / 8, hipHostMallocPortable | hipHostMallocMapped));
        g->list_cap = list_bytes + list_bytes / 8;
    }
    if (g->out_cap < out_bytes) {
        if (g->h_out) (void)hipHostFree(g->h_out);
        g->h_out = nullptr; g->out_cap = 0;
        HIP_TRY(hipHostMalloc(&g->h_out, out_bytes + out_bytes / 8, hipHostMallocPortable | hipHostMallocMapped));
        g->out_cap = out_bytes + out_bytes / 8;
    }
    memcpy(g->h_list, anchors, list_bytes);  // (the caller's memory is pageable: one copy, every device reads the pinned block)
    std::vector<GroupDev> dev((size_t)world);
    int first_rc = LCHD_OK;
    char first_msg[sizeof g_err] = "";
    auto note = [&](int rc) {
        if (rc && !first_rc) { first_rc = rc; memcpy(first_msg, g_err, sizeof g_err); }
        return rc;
    };
    const int64_t slot = n_pairs / world + n_pairs / (2 * world) + 4096;  // most pairs of a share under the balance rule (1.25 P / W + 1), with headroom
    // 1. both structures are staged ONCE (AoS -> SoA, finiteness, bounding box: host work that must not grow with the number of
    //    devices) into the group's pinned block; every device copies them and the pair list from there and plans the partition
    lchd_cloud sa, sb;  // device pointers relative to a null base: offsets
    size_t in_bytes = 0;
    {
        size_t off = 0;
        if (int rc = stage_cloud(xyz_a, cat_a, tag_a, n_a, nullptr, nullptr, off, sa, cfg->n_categories > kMaxCategories)) return rc;
        if (int rc = stage_cloud(xyz_b, cat_b, tag_b, n_b, nullptr, nullptr, off, sb, cfg->n_categories > kMaxCategories)) return rc;
        in_bytes = off;
        if (g->stage_cap < in_bytes) {
            if (g->h_stage) (void)hipHostFree(g->h_stage);
            g->h_stage = nullptr; g->stage_cap = 0;
            HIP_TRY(hipHostMalloc(&g->h_stage, in_bytes + in_bytes / 8, hipHostMallocPortable));
            g->stage_cap = in_bytes + in_bytes / 8;
        }
        off = 0;
        if (int rc = stage_cloud(xyz_a, cat_a, tag_a, n_a, g->h_stage, nullptr, off, sa, cfg->n_categories > kMaxCategories)) return rc;
        if (int rc = stage_cloud(xyz_b, cat_b, tag_b, n_b, g->h_stage, nullptr, off, sb, cfg->n_categories > kMaxCategories)) return rc;
    }
    auto rebase = [](const lchd_cloud& src, char* d_base) {  // the staged structure as device `d_base` sees it
        lchd_cloud cl = src;
        auto mv = [&](auto*& p) { p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(d_base + reinterpret_cast<uintptr_t>(p)); };
        mv(cl.x); mv(cl.y); mv(cl.z); mv(cl.tag); mv(cl.cat);  // (offsets from a null base; x sits at offset 0)
        if (cl.cat_hi) mv(cl.cat_hi);                            // (absent unless the configuration has more than 255 categories)
        return cl;
    };
    for (int r = 0; r < world; ++r) {
        lchd_ctx* c = g->ctx[(size_t)r];
        GroupDev& D = dev[(size_t)r];
        if (c->pend.active) { note(fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)")); continue; }
        CTX_GUARD(c);
        if (note(lchd_ctx_set_config(c, cfg))) continue;
        size_t off = in_bytes;
        auto take = [&](size_t bytes) { off = (off + 255) & ~size_t(255); const size_t o = off; off += bytes; return o; };
        D.o_list = take(list_bytes);
        D.o_sel = take(sizeof(int64_t) * 2 * (size_t)slot);
        D.o_idx = take(sizeof(int64_t) * (size_t)slot);
        D.o_sc = take(sizeof(double) * (size_t)slot);
        if (note(grow_io(c, off))) continue;
        D.a = rebase(sa, c->d_io);
        D.b = rebase(sb, c->d_io);
        if (hipMemcpyAsync(c->d_io, g->h_stage, in_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) { note(fail(LCHD_EDEVICE, "H2D copy of the structures failed")); continue; }
        if (note(shard_plan_enqueue(c, reinterpret_cast<const int64_t*>(c->d_io + D.o_list), g->h_list, n_pairs, n_a, n_b, world))) continue;
        D.planned = true;
    }
    // 2. every device: its share's size, the selection, the pass
    for (int r = 0; r < world; ++r) {
        lchd_ctx* c = g->ctx[(size_t)r];
        GroupDev& D = dev[(size_t)r];
        if (!D.planned) continue;
        CTX_GUARD(c);
        int64_t counts[kShardMaxWorld];
        if (note(shard_plan_wait(c, n_pairs, n_a, n_b, world, counts))) continue;
        D.n_mine = counts[r];
        g->last_counts[(size_t)r] = D.n_mine;
        if (D.n_mine > slot) { note(fail(LCHD_EDEVICE, "a share of %lld pairs exceeds the slot of %lld", (long long)D.n_mine, (long long)slot)); continue; }
        if (D.n_mine == 0) continue;
        int64_t* d_sel = reinterpret_cast<int64_t*>(c->d_io + D.o_sel);
        int64_t* d_idx = reinterpret_cast<int64_t*>(c->d_io + D.o_idx);
        if (note(lchd_shard_select_dev(c, reinterpret_cast<const int64_t*>(c->d_io + D.o_list), n_pairs, n_a, n_b, r, d_sel, d_idx))) continue;
        if (note(lchd_from_primitives_dev_async(c, &D.a, &D.b, d_sel, nullptr, D.n_mine, thr, reinterpret_cast<double*>(c->d_io + D.o_sc)))) continue;
        D.enqueued = true;
    }
    // 3. finish (capacity retries happen here), then the scores to their positions in the pinned block
    for (int r = 0; r < world; ++r) {
        lchd_ctx* c = g->ctx[(size_t)r];
        GroupDev& D = dev[(size_t)r];
        if (!D.enqueued) continue;
        CTX_GUARD(c);
        if (note(lchd_ctx_finish(c))) { D.enqueued = false; continue; }
        launch_scatter_scores(c->stream, reinterpret_cast<const double*>(c->d_io + D.o_sc), reinterpret_cast<const int64_t*>(c->d_io + D.o_idx),
                              D.n_mine, g->h_out);
    }
    for (int r = 0; r < world; ++r) {
        lchd_ctx* c = g->ctx[(size_t)r];
        GroupDev& D = dev[(size_t)r];
        c->last_valid = false;
        c->pend.a = c->pend.b = nullptr;  // the structures of the call live in the I/O block, which the next call overwrites
        if (!D.enqueued) continue;
        CTX_GUARD(c);
        if (hipStreamSynchronize(c->stream) != hipSuccess) note(fail(LCHD_EDEVICE, "HIP error while waiting for device %d of the group", r));
    }
    if (first_rc) { memcpy(g_err, first_msg, sizeof g_err); return first_rc; }
    memcpy(out, g->h_out, out_bytes);
    return LCHD_OK;
}
extern "C" int32_t lchd_group_size(const lchd_group* g) { return g ? (int32_t)g->ctx.size() : 0; }
extern "C" int lchd_group_last_counts(const lchd_group* g, int64_t* counts_out) {
    if (!g || !counts_out) return fail(LCHD_EVALUE, "null argument");
    for (size_t k = 0; k < g->ctx.size(); ++k) counts_out[k] = g->last_counts[k];
    return LCHD_OK;
}

extern "C" int lchd_group_from_primitives(lchd_group* g, const lchd_config* cfg, const double* xyz_a, const int32_t* cat_a,
                                          const int32_t* tag_a, int64_t n_a, const double* xyz_b, const int32_t* cat_b,
                                          const int32_t* tag_b, int64_t n_b, const int64_t* anchors, const int32_t* wf_index,
                                          int64_t n_pairs, double thr, double* out) {
    if (!g || !cfg || g->ctx.empty()) return fail(LCHD_EVALUE, "null group / configuration");
    if (int rc = check_wf_index(cfg, wf_index, n_pairs)) return rc;
    if (n_a < 0 || n_b < 0 || n_a > ((int64_t)1 << 30) || n_b > ((int64_t)1 << 30)) return fail(LCHD_EUNSUPPORTED, "structure size out of range");
    const int world = (int)g->ctx.size();
    std::fill(g->last_counts.begin(), g->last_counts.end(), 0);
    if (n_pairs <= 0) {
        for (lchd_ctx* c : g->ctx)
            if (int rc = lchd_ctx_set_config(c, cfg)) return rc;  // the reference validates its arguments even for an empty list
        return LCHD_OK;
    }
    // several devices, one weight function: the partition runs on the devices (above)
    if (world > 1 && n_a > 0 && !wf_index)
        return group_call_device_partition(g, cfg, xyz_a, cat_a, tag_a, n_a, xyz_b, cat_b, tag_b, n_b, anchors, n_pairs, thr, out);
    // the pair list, binned by anchor (the rule of k_shard_plan): device r gets the positions subset[r]
    std::vector<std::vector<int64_t>> subset((size_t)world);
    if (world == 1 || n_a <= 0) {
        subset[0].resize((size_t)n_pairs);
        for (int64_t p = 0; p < n_pairs; ++p) subset[0][(size_t)p] = p;
    } else {
        std::vector<uint16_t> rob;
        const int mode = shard_rule_host(anchors, n_pairs, n_a, n_b, world, rob);
        for (int r = 0; r < world; ++r) subset[(size_t)r].reserve((size_t)(n_pairs / world + n_pairs / (4 * world) + 16));
        for (int64_t p = 0; p < n_pairs; ++p) {
            if (mode == 2) { subset[(size_t)((p * world) / n_pairs)].push_back(p); continue; }
            const int64_t n = mode == 1 ? n_b : n_a;
            int64_t a = anchors[2 * p + (mode == 1 ? 1 : 0)];
            a = a < 0 ? 0 : (a >= n ? n - 1 : a);
            subset[rob[(size_t)((a * kShardBins) / n)]].push_back(p);
        }
    }
    // enqueue every device's pass (asynchronous), then collect: the devices work concurrently
    std::vector<HostCall> calls((size_t)world);
    int first_rc = LCHD_OK;
    char first_msg[sizeof g_err] = "";
    auto note = [&](int rc) {
        if (rc && !first_rc) { first_rc = rc; memcpy(first_msg, g_err, sizeof g_err); }
    };
    for (int r = 0; r < world; ++r) {
        g->last_counts[(size_t)r] = (int64_t)subset[(size_t)r].size();
        note(host_call_enqueue(g->ctx[(size_t)r], cfg, xyz_a, cat_a, tag_a, n_a, xyz_b, cat_b, tag_b, n_b, anchors, wf_index,
                               subset[(size_t)r].data(), (int64_t)subset[(size_t)r].size(), thr, calls[(size_t)r]));
    }
    for (int r = 0; r < world; ++r) note(host_call_finish(g->ctx[(size_t)r], calls[(size_t)r], subset[(size_t)r].data(), out));
    if (first_rc) memcpy(g_err, first_msg, sizeof g_err);
    return first_rc;
}

// ---- one process per GPU: the same partition on the device ------------------------------------------------------------
static int ensure_shard_state(lchd_ctx* c) {
    if (c->d_shard) return LCHD_OK;
    // The plan kernel runs on a NON-BLOCKING side stream, which is not ordered behind work of the null stream: the state is
    // zeroed ON that stream (a plain hipMemset may still be pending when the first plan starts and would then wipe the
    // plan's bin table under the selection kernel), and the call waits for it.
    hipError_t e = hipStreamCreateWithFlags(&c->shard_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->shard_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->shard_sel_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&c->d_shard, sizeof(ShardState));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_shard, 0, sizeof(ShardState), c->shard_stream);
    if (e == hipSuccess) e = hipHostMalloc(&c->h_counts, sizeof(int64_t) * (kShardMaxWorld + 1));
    if (e == hipSuccess) e = hipMalloc(&c->d_bad, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_bad, 0, sizeof(uint32_t), c->shard_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->shard_stream);
    if (e != hipSuccess) {
        if (c->shard_stream) { (void)hipStreamDestroy(c->shard_stream); c->shard_stream = nullptr; }
        if (c->shard_ev) { (void)hipEventDestroy(c->shard_ev); c->shard_ev = nullptr; }
        if (c->shard_sel_ev) { (void)hipEventDestroy(c->shard_sel_ev); c->shard_sel_ev = nullptr; }
        (void)hipFree(c->d_shard); c->d_shard = nullptr;
        if (c->h_counts) { (void)hipHostFree(c->h_counts); c->h_counts = nullptr; }
        (void)hipFree(c->d_bad); c->d_bad = nullptr;
        return fail(LCHD_EDEVICE, "HIP error %d (%s) while allocating the sharding state", (int)e, hipGetErrorName(e));
    }
    return LCHD_OK;
}
// The plan in two halves: enqueue (asynchronous, on the context's side stream) and wait (the per-rank counts come back through
// pinned memory).  A caller that drives several devices enqueues all plans before it waits for the first.
// h_src: when not null the pair list is first copied from this (pinned) host block to d_anchors, on the same side stream.
static int shard_plan_enqueue(lchd_ctx* c, const int64_t* d_anchors, const int64_t* h_src, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                              int32_t world) {
    c->shard_world = 0;
    if (int rc = ensure_shard_state(c)) return rc;
    // On the context's side stream: the pair list is an INPUT (the caller has it ready), so the plan neither waits for the
    // scoring passes queued on the context's stream nor makes the host wait for them.
    if (c->shard_sel_pending) { HIP_TRY(hipStreamWaitEvent(c->shard_stream, c->shard_sel_ev, 0)); c->shard_sel_pending = false; }
    if (h_src) HIP_TRY(hipMemcpyAsync(const_cast<int64_t*>(d_anchors), h_src, sizeof(int64_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, c->shard_stream));
    launch_shard_plan(c->shard_stream, d_anchors, n_pairs, n_atoms_a, n_atoms_b, world, c->d_shard, c->h_counts);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->shard_ev, c->shard_stream));
    return LCHD_OK;
}
static int shard_plan_wait(lchd_ctx* c, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int32_t world, int64_t* counts_out) {
    HIP_TRY(hipStreamSynchronize(c->shard_stream));
    int64_t total = 0;
    for (int r = 0; r < world; ++r) { counts_out[r] = c->h_counts[r]; total += counts_out[r]; }
    if (total != n_pairs) return fail(LCHD_EDEVICE, "the shard plan accounts for %lld of %lld pairs", (long long)total, (long long)n_pairs);
    c->shard_world = world; c->shard_pairs = n_pairs; c->shard_atoms = n_atoms_a; c->shard_atoms_b = n_atoms_b;
    return LCHD_OK;
}
extern "C" int lchd_shard_plan_dev(lchd_ctx* c, const int64_t* d_anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int32_t world,
                                   int64_t* counts_out) {
    if (!c || !counts_out) return fail(LCHD_EVALUE, "null argument");
    if (world < 1 || world > kShardMaxWorld) return fail(LCHD_EVALUE, "world size %d outside [1, %d]", world, kShardMaxWorld);
    if (n_pairs < 0 || n_atoms_a < 1) return fail(LCHD_EVALUE, "bad pair / atom count");
    for (int r = 0; r < world; ++r) counts_out[r] = 0;
    c->shard_world = 0;
    if (n_pairs == 0) { c->shard_world = world; c->shard_pairs = 0; c->shard_atoms = n_atoms_a; c->shard_atoms_b = n_atoms_b; return LCHD_OK; }
    if (!d_anchors) return fail(LCHD_EVALUE, "null anchor pointer");
    CTX_GUARD(c);
    if (int rc = shard_plan_enqueue(c, d_anchors, nullptr, n_pairs, n_atoms_a, n_atoms_b, world)) return rc;
    return shard_plan_wait(c, n_pairs, n_atoms_a, n_atoms_b, world, counts_out);
}
extern "C" int lchd_shard_select_dev(lchd_ctx* c, const int64_t* d_anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int32_t rank,
                                     int64_t* d_sel_anchors, int64_t* d_sel_index) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    if (c->shard_world < 1 || n_pairs != c->shard_pairs || n_atoms_a != c->shard_atoms || n_atoms_b != c->shard_atoms_b)
        return fail(LCHD_EVALUE, "lchd_shard_plan_dev has not been called for this pair list");
    if (rank < 0 || rank >= c->shard_world) return fail(LCHD_EVALUE, "rank %d outside the planned world of %d", rank, c->shard_world);
    if (n_pairs == 0 || c->h_counts[rank] == 0) return LCHD_OK;  // nothing for this rank: the outputs may be null
    if (!d_anchors || !d_sel_anchors || !d_sel_index) return fail(LCHD_EVALUE, "null pointer");
    CTX_GUARD(c);
    HIP_TRY(hipStreamWaitEvent(c->stream, c->shard_ev, 0));
    launch_shard_select(c->stream, d_anchors, n_pairs, n_atoms_a, n_atoms_b, rank, c->shard_world, c->d_shard, d_sel_anchors, d_sel_index);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->shard_sel_ev, c->stream));
    c->shard_sel_pending = true;
    return LCHD_OK;
}
extern "C" int lchd_unshard_scores_dev(lchd_ctx* c, const double* d_gathered, const int64_t* counts, int32_t world, int64_t stride,
                                       double* d_out, int64_t n_pairs) {
    if (!c || !counts) return fail(LCHD_EVALUE, "null argument");
    if (world < 1 || world > kShardMaxWorld) return fail(LCHD_EVALUE, "world size %d outside [1, %d]", world, kShardMaxWorld);
    ShardCounts sc{};
    int64_t total = 0;
    for (int r = 0; r < world; ++r) {
        if (counts[r] < 0 || counts[r] > stride) return fail(LCHD_EVALUE, "rank %d holds %lld scores in a slot of %lld", r, (long long)counts[r], (long long)stride);
        sc.n[r] = counts[r];
        total += counts[r];
    }
    if (total != n_pairs) return fail(LCHD_EVALUE, "the ranks hold %lld scores for %lld pairs", (long long)total, (long long)n_pairs);
    if (n_pairs == 0) return LCHD_OK;
    if (!d_gathered || !d_out) return fail(LCHD_EVALUE, "null pointer");
    CTX_GUARD(c);
    if (int rc = ensure_shard_state(c)) return rc;
    launch_unshard_scores(c->stream, d_gathered, sc, world, stride, d_out, n_pairs, c->d_bad);
    HIP_TRY(hipGetLastError());
    uint32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, c->d_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (bad) {
        (void)hipMemsetAsync(c->d_bad, 0, sizeof(uint32_t), c->stream);
        return fail(LCHD_EVALUE, "a gathered pair position lies outside [0, %lld)", (long long)n_pairs);
    }
    return LCHD_OK;
}

// Shared tail of from_anchors / from_dmxs / from_coords: environments are already sorted in `ea`/`eb`
// (one per row), pair p = (row p, row p).
static int sweep_rows(lchd_ctx* c, const EnvStore& ea, const EnvStore& eb, const int32_t* d_wf, int64_t rows, double* d_out,
                      int4* d_meta, Driver drv, uint32_t* flags_out = nullptr) {
    SweepArgs sw{};
    fill_sweep_args(c, sw);
    sw.env_a = ea;
    sw.env_b = eb;
    sw.wf_index = d_wf;
    sw.n_pairs = rows;
    sw.out = d_out;
    sw.meta = d_meta;
    (void)launch_sweep(c->stream, c->tune, c->h_cfg.n_categories, c->hellinger2, c->unit_weights, c->wf_pow, 0, sw);
    mark(c, 4);
    HIP_TRY(hipGetLastError());
    c->status_dirty = false;  // the record pass of this sequence resets the device status
    uint32_t f = 0;
    if (int rc = wait_pass(c, &f)) return rc;
    collect_times(c, 2, 4);
    if (flags_out) *flags_out = f;
    if (f & ST_ROW_RETRY) return LCHD_OK;  // the caller repeats the pass with the other row kernel
    return status_to_rc(f, drv);
}

// One dense pass (from_coords / from_dmxs semantics: pair r = row r of A against row r of B, the environment is the whole
// structure): row sort of both structures, sweep.  a / b: device-resident structures (coordinates used unless d_ma / d_mb,
// DEVICE pointers to given distance rows, are set); d_wf: device weight-function indices or nullptr; d_out: device or
// host-mapped scores.  `retry` reports that a long row defeated the segmented in-LDS sort (repeat with old_rows).
static bool unit_weights_for_dense(const lchd_ctx* c) { return c->unit_weights; }

static int dense_pass(lchd_ctx* c, const lchd_cloud& a, const lchd_cloud& b, const double* d_ma, const double* d_mb, int64_t rows,
                      int64_t cols_a, int64_t cols_b, const int32_t* d_wf, double* d_out, bool old_rows, bool& retry,
                      const double* h_ma = nullptr, const double* h_mb = nullptr, const int32_t* d_len_a = nullptr,
                      const int32_t* d_len_b = nullptr) {
    retry = false;
    const int cap_a = next_pow2_host(cols_a), cap_b = next_pow2_host(cols_b);
    // more than 255 categories: 16-bit ids in the environment store, k_env_rows<.., uint16_t> + k_sweep_wide<.., CAT16>
    const bool cat16 = c->h_cfg.n_categories > kMaxCategories;
    const size_t cat_bytes = cat16 ? 2 : 1;
    auto diag2 = [](const lchd_cloud& cl) {  // squared diagonal of the bounding box, with a little headroom
        double s2 = 0.0;
        for (int k = 0; k < 3; ++k) { const double e = cl.bbmax[k] - cl.bbmin[k]; s2 += e * e; }
        return s2 * (1.0 + 1e-9) + 1e-300;
    };
    c->last_dense_fused = false;
    // The common configuration (Hellinger-2, unit category weights, <= 16 categories, rows of 1025 .. 20480 points = kDenseFusedMaxRow): sort and
    // sweep in ONE kernel per row pair, nothing but the score is written (lchd_dense_fused.hip).  No environment store.
    if (!old_rows && !c->tune.no_dense_fused && !cat16 && c->hellinger2 && unit_weights_for_dense(c) &&
        dense_fused_applies(c->h_cfg.n_categories, cols_a, cols_b)) {
        double* w_ma2 = nullptr;
        double* w_mb2 = nullptr;
        uint64_t* scr_key = nullptr;
        uint8_t* scr_val = nullptr;
        int32_t scr_grid = 0, scr_segs = 0;
        const size_t scr_n = dense_fused_scratch(rows, cols_a, cols_b, &scr_grid, &scr_segs);
        for (int dry = 1; dry >= 0; --dry) {
            Arena ar(dry ? nullptr : c->ws, dry ? 0 : c->ws_cap, dry != 0);
            scr_key = ar.take<uint64_t>(scr_n);  // the distance pass's (key, value) pairs, one region per workgroup
            scr_val = ar.take<uint8_t>(scr_n);
            if (h_ma) {
                w_ma2 = ar.take<double>((size_t)rows * cols_a);
                w_mb2 = ar.take<double>((size_t)rows * cols_b);
            }
            if (dry) if (int rc2 = ensure_ws(c, ar.off + 4096)) return rc2;
        }
        if (h_ma) {
            HIP_TRY(hipMemcpyAsync(w_ma2, h_ma, sizeof(double) * rows * cols_a, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(w_mb2, h_mb, sizeof(double) * rows * cols_b, hipMemcpyHostToDevice, c->stream));
            d_ma = w_ma2;
            d_mb = w_mb2;
        }
        DenseArgs da{};
        da.cfg = c->d_cfg;
        da.s[0] = DenseSide{a.view(cat16), d_ma, cols_a, (int32_t)cols_a, d_len_a};
        da.s[1] = DenseSide{b.view(cat16), d_mb, cols_b, (int32_t)cols_b, d_len_b};
        da.n_rows = rows;
        da.image_bound = std::max(diag2(a), diag2(b));
        da.wf_index = d_wf;
        da.out = d_out;
        da.st = c->d_status;
        da.sqrt_tab = c->d_tabs;
        da.scr_key = scr_key;
        da.scr_val = scr_val;
        da.scr_grid = scr_grid;
        da.scr_segs = scr_segs;
        da.ticket = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->d_done) + sizeof(DoneState));
        mark(c, 2);
        if (launch_dense_fused(c->stream, c->h_cfg.n_categories, da, c->h_status, c->seq)) {
            mark(c, 3);
            mark(c, 4);
            HIP_TRY(hipGetLastError());
            c->status_dirty = false;  // the kernel's companion launch hands the status over and resets it
            uint32_t f = 0;
            if (int rc2 = wait_pass(c, &f)) return rc2;
            collect_times(c, 2, 4);
            c->last_dense_fused = true;
            if (f & ST_ROW_RETRY) { retry = true; return LCHD_OK; }  // a row the fused kernel gives up on: the caller repeats with the two-kernel path
            return status_to_rc(f, DRV_DMXS);
        }
    }
    EnvStore ea{}, eb{};
    double *w_ma = nullptr, *w_mb = nullptr;
    int4* d_meta = nullptr;
    for (int dry = 1; dry >= 0; --dry) {
        Arena ar(dry ? nullptr : c->ws, dry ? 0 : c->ws_cap, dry != 0);
        ea.key = ar.take<uint64_t>((size_t)rows * cap_a);
        ea.cat = ar.take<uint8_t>((size_t)rows * cap_a * cat_bytes);
        ea.len = ar.take<int32_t>((size_t)rows);
        ea.stride = cap_a;
        ea.cat16 = eb.cat16 = cat16 ? 1 : 0;
        eb.key = ar.take<uint64_t>((size_t)rows * cap_b);
        eb.cat = ar.take<uint8_t>((size_t)rows * cap_b * cat_bytes);
        eb.len = ar.take<int32_t>((size_t)rows);
        eb.stride = cap_b;
        ea.cdf_keys = eb.cdf_keys = (c->h_cfg.n_wf == 1 && !c->tune.no_cdf_keys) ? 1 : 0;
        d_meta = ar.take<int4>((size_t)rows);
        if (h_ma) {  // given distance matrices in host memory: straight from the caller's buffers into the workspace
            w_ma = ar.take<double>((size_t)rows * cols_a);
            w_mb = ar.take<double>((size_t)rows * cols_b);
        }
        if (dry) if (int rc2 = ensure_ws(c, ar.off + 4096)) return rc2;
    }
    hipStream_t s = c->stream;
    if (h_ma) {
        HIP_TRY(hipMemcpyAsync(w_ma, h_ma, sizeof(double) * rows * cols_a, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(w_mb, h_mb, sizeof(double) * rows * cols_b, hipMemcpyHostToDevice, s));
        d_ma = w_ma;
        d_mb = w_mb;
    }
    mark(c, 2);
    const RowSide rsa{a.view(cat16), d_ma, cols_a, cols_a, diag2(a), ea, d_len_a}, rsb{b.view(cat16), d_mb, cols_b, cols_b, diag2(b), eb, d_len_b};
    if (old_rows || cat16 || !launch_env_rows2(s, c->d_cfg, rsa, rsb, rows, c->d_status)) {  // (rows beyond 20480 points: keys sorted in global memory)
        const RowExtras exa{nullptr, d_len_a, nullptr}, exb{nullptr, d_len_b, nullptr};
        if (!launch_env_rows(s, cap_a, c->d_cfg, a.view(cat16), d_ma, cols_a, rows, cols_a, diag2(a), ea, c->d_status, exa) ||
            !launch_env_rows(s, cap_b, c->d_cfg, b.view(cat16), d_mb, cols_b, rows, cols_b, diag2(b), eb, c->d_status, exb))
            return fail(LCHD_EUNSUPPORTED, "no dense environment kernel for this row length");
    }
    if (c->deterministic) launch_env_canon(s, ea, eb, rows, rows, nullptr);  // one order among equal keys (a row's sort places ties by LDS-atomic order)
    mark(c, 3);
    uint32_t f = 0;
    if (int rc2 = sweep_rows(c, ea, eb, d_wf, rows, d_out, d_meta, DRV_DMXS, &f)) return rc2;
    if (f & ST_ROW_RETRY) retry = true;
    return LCHD_OK;
}

static int dense_driver(lchd_ctx* c, const lchd_config* cfg, const int32_t* seq_a, int64_t len_seq_a, const int32_t* seq_b,
                        int64_t len_seq_b, const double* xyz_a, const double* xyz_b, const double* dmx_a, const double* dmx_b,
                        int64_t rows, int64_t cols_a, int64_t cols_b, const int32_t* wf_index, double* out,
                        const int32_t* row_len_a = nullptr, const int32_t* row_len_b = nullptr) {
    if (c->pend.active) return fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)");
    CTX_GUARD(c);
    if (int rc = lchd_ctx_set_config(c, cfg)) return rc;
    if (int rc = check_wf_index(cfg, wf_index, rows)) return rc;
    c->last_valid = false;
    if (rows == 0) return LCHD_OK;
    // utils.rs:25-39: the sort mask has row-length entries and indexes seq => a row longer than seq panics
    if (cols_a > len_seq_a || cols_b > len_seq_b) return fail(LCHD_EPANIC, "index out of bounds: a distance row is longer than its seq");
    if (cols_a == 0 || cols_b == 0) return fail(LCHD_EPANIC, "index out of bounds: empty distance row (src/locohd.rs:74)");
    if ((cols_a > 65535 || cols_b > 65535) && (cfg->n_categories > kMaxCategories || cols_a > (1 << 23) || cols_b > (1 << 23)))
        return fail(LCHD_EUNSUPPORTED, "dense rows of more than 65535 points are supported with at most %d categories and 2^23 points (got %lld / %lld)",
                    kMaxCategories, (long long)cols_a, (long long)cols_b);
    // The two structures (SoA coordinates + categories), the weight-function indices and -- for calls of up to kDirectOutPairs
    // rows -- the scores travel through the context's pinned staging block (one asynchronous copy in, none out); nothing is
    // allocated per call.
    for (int side = 0; side < 2 && (row_len_a || row_len_b); ++side) {  // ragged rows (utils.rs:25-39: a row is sorted with a prefix of seq)
        const int32_t* rl = side ? row_len_b : row_len_a;
        const int64_t cols = side ? cols_b : cols_a;
        if (!rl) return fail(LCHD_EVALUE, "row lengths must be given for both matrices or for neither");
        for (int64_t r = 0; r < rows; ++r) {
            if (rl[r] < 1) return fail(LCHD_EPANIC, "index out of bounds: empty distance row (src/locohd.rs:74)");
            if (rl[r] > cols) return fail(LCHD_EVALUE, "row %lld is longer (%d) than the padded matrix (%lld columns)", (long long)r, rl[r], (long long)cols);
        }
    }
    lchd_cloud a, b;
    size_t o_wf = 0, o_out = 0, in_bytes = 0, o_la = 0, o_lb = 0;
    for (int pass = 0; pass < 2; ++pass) {
        size_t off = 0;
        char* hb = pass ? c->h_io : nullptr;
        if (int rc = stage_cloud(xyz_a, seq_a, nullptr, cols_a, hb, c->d_io, off, a, cfg->n_categories > kMaxCategories)) return rc;
        if (int rc = stage_cloud(xyz_b, seq_b, nullptr, cols_b, hb, c->d_io, off, b, cfg->n_categories > kMaxCategories)) return rc;
        auto take = [&](size_t bytes) { off = (off + 255) & ~size_t(255); const size_t o = off; off += bytes; return o; };
        o_wf = take(wf_index ? sizeof(int32_t) * (size_t)rows : 0);
        o_la = take(row_len_a ? sizeof(int32_t) * (size_t)rows : 0);
        o_lb = take(row_len_b ? sizeof(int32_t) * (size_t)rows : 0);
        in_bytes = off;
        o_out = take(sizeof(double) * (size_t)rows);
        if (pass == 0)
            if (int rc = grow_io(c, off)) return rc;
    }
    if (wf_index) memcpy(c->h_io + o_wf, wf_index, sizeof(int32_t) * (size_t)rows);
    if (row_len_a) memcpy(c->h_io + o_la, row_len_a, sizeof(int32_t) * (size_t)rows);
    if (row_len_b) memcpy(c->h_io + o_lb, row_len_b, sizeof(int32_t) * (size_t)rows);
    const int32_t* d_la = row_len_a ? reinterpret_cast<const int32_t*>(c->d_io + o_la) : nullptr;
    const int32_t* d_lb = row_len_b ? reinterpret_cast<const int32_t*>(c->d_io + o_lb) : nullptr;
    const bool direct = rows <= kDirectOutPairs;
    const int32_t* d_wf = wf_index ? reinterpret_cast<const int32_t*>(c->d_io + o_wf) : nullptr;
    double* d_out = reinterpret_cast<double*>((direct ? c->h_io : c->d_io) + o_out);
    int rc = LCHD_OK;
    bool retry = false;
    for (int attempt = 0; attempt < 2; ++attempt) {  // (second attempt: a long row defeated the segmented in-LDS sort)
        if (int rc2 = begin_pass(c)) return rc2;
        c->status_dirty = true;  // until the record pass has been enqueued (sweep_rows)
        HIP_TRY(hipMemcpyAsync(c->d_io, c->h_io, in_bytes, hipMemcpyHostToDevice, c->stream));
        rc = dense_pass(c, a, b, nullptr, nullptr, rows, cols_a, cols_b, d_wf, d_out, c->tune.old_rows || attempt > 0, retry, dmx_a, dmx_b, d_la, d_lb);
        if (rc || !retry) break;
    }
    if (rc) return rc;
    if (!direct) {
        HIP_TRY(hipMemcpyAsync(c->h_io + o_out, d_out, sizeof(double) * rows, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    memcpy(out, c->h_io + o_out, sizeof(double) * (size_t)rows);
    return LCHD_OK;
}

/* from_coords with both structures already on the device (what bench.py --workload c2b measures): pair r = (atom r of a,
 * atom r of b), the environments are the whole structures.  d_wf_index / d_out are device pointers ([n] int32 or NULL,
 * [n] double).  Uses the configuration of lchd_ctx_set_config; d_out is complete on return. */
extern "C" int lchd_from_coords_dev(lchd_ctx* c, lchd_cloud* a, lchd_cloud* b, const int32_t* d_wf_index, double* d_out) {
    if (!c || !a || !b) return fail(LCHD_EVALUE, "null argument");
    if (!c->cfg_set) return fail(LCHD_EVALUE, "lchd_ctx_set_config has not been called");
    if (c->pend.active) return fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)");
    if (a->sid || b->sid) return fail(LCHD_EVALUE, "from_coords takes single structures, not batches");
    if (a->n != b->n)  // src/locohd.rs:420-428 via :472-475
        return fail(LCHD_EVALUE, "Expected matrices with the same length, got lengths %lld and %lld!", (long long)a->n, (long long)b->n);
    c->last_valid = false;
    if (a->n == 0) return LCHD_OK;
    if (!d_out) return fail(LCHD_EVALUE, "null score pointer");
    if (a->n > 65535 && (c->h_cfg.n_categories > kMaxCategories || a->n > (1 << 23)))
        return fail(LCHD_EUNSUPPORTED, "dense rows of more than 65535 points are supported with at most %d categories and 2^23 points (got %lld)",
                    kMaxCategories, (long long)a->n);
    CTX_GUARD(c);
    bool retry = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (int rc = begin_pass(c)) return rc;
        c->status_dirty = true;
        if (int rc = dense_pass(c, *a, *b, nullptr, nullptr, a->n, a->n, b->n, d_wf_index, d_out, c->tune.old_rows || attempt > 0, retry)) return rc;
        if (!retry) break;
    }
    return LCHD_OK;
}

extern "C" int lchd_from_coords(lchd_ctx* c, const lchd_config* cfg, const int32_t* seq_a, int64_t len_seq_a, const int32_t* seq_b,
                                int64_t len_seq_b, const double* xyz_a, int64_t n_a, const double* xyz_b, int64_t n_b,
                                const int32_t* wf_index, double* out) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    if (n_a != n_b)  // src/locohd.rs:420-428 via :472-475
        return fail(LCHD_EVALUE, "Expected matrices with the same length, got lengths %lld and %lld!", (long long)n_a, (long long)n_b);
    return dense_driver(c, cfg, seq_a, len_seq_a, seq_b, len_seq_b, xyz_a, xyz_b, nullptr, nullptr, n_a, n_a, n_b, wf_index, out);
}

extern "C" int lchd_from_dmxs(lchd_ctx* c, const lchd_config* cfg, const int32_t* seq_a, int64_t len_seq_a, const int32_t* seq_b,
                              int64_t len_seq_b, const double* dmx_a, int64_t rows_a, int64_t cols_a, const double* dmx_b,
                              int64_t rows_b, int64_t cols_b, const int32_t* wf_index, double* out) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    if (rows_a != rows_b)  // src/locohd.rs:420-428
        return fail(LCHD_EVALUE, "Expected matrices with the same length, got lengths %lld and %lld!", (long long)rows_a, (long long)rows_b);
    return dense_driver(c, cfg, seq_a, len_seq_a, seq_b, len_seq_b, nullptr, nullptr, dmx_a, dmx_b, rows_a, cols_a, cols_b, wf_index, out);
}

/* from_dmxs with ragged rows: the reference takes Vec<Vec<f64>> and sorts each row with a PREFIX of seq (utils.rs:25-39), so rows
 * may be shorter than seq and differ in length.  dmx_x is the row-major [rows][cols_x] matrix padded with anything; row r of
 * side x has row_len_x[r] (1 .. cols_x) real entries.  Entries and categories beyond a row's length are never looked at. */
extern "C" int lchd_from_dmxs_ragged(lchd_ctx* c, const lchd_config* cfg, const int32_t* seq_a, int64_t len_seq_a, const int32_t* seq_b,
                                     int64_t len_seq_b, const double* dmx_a, int64_t rows_a, int64_t cols_a, const int32_t* row_len_a,
                                     const double* dmx_b, int64_t rows_b, int64_t cols_b, const int32_t* row_len_b, const int32_t* wf_index,
                                     double* out) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    if (rows_a != rows_b)  // src/locohd.rs:420-428
        return fail(LCHD_EVALUE, "Expected matrices with the same length, got lengths %lld and %lld!", (long long)rows_a, (long long)rows_b);
    if (!row_len_a || !row_len_b) return fail(LCHD_EVALUE, "null row-length array");
    return dense_driver(c, cfg, seq_a, len_seq_a, seq_b, len_seq_b, nullptr, nullptr, dmx_a, dmx_b, rows_a, cols_a, cols_b, wf_index, out, row_len_a,
                        row_len_b);
}

extern "C" int lchd_from_anchors(lchd_ctx* c, const lchd_config* cfg, const int32_t* seq_a, int64_t len_seq_a, const double* dists_a,
                                 int64_t len_dists_a, const int32_t* seq_b, int64_t len_seq_b, const double* dists_b,
                                 int64_t len_dists_b, int32_t wf_index, double* out) {
    if (!c) return fail(LCHD_EVALUE, "null context");
    if (c->pend.active) return fail(LCHD_EVALUE, "an asynchronous call has not been finished (lchd_ctx_finish)");
    CTX_GUARD(c);
    if (int rc = lchd_ctx_set_config(c, cfg)) return rc;
    if (wf_index < 0 || wf_index >= cfg->n_weight_functions) return fail(LCHD_EVALUE, "weight-function index out of range");
    c->last_valid = false;
    // src/locohd.rs:70-77
    if (len_seq_a != len_dists_a || len_seq_b != len_dists_b) return fail(LCHD_EVALUE, "Lists seq and dists must have equal lengths!");
    if (len_seq_a == 0 || len_seq_b == 0) return fail(LCHD_EPANIC, "index out of bounds: the len is 0 but the index is 0");
    if (dists_a[0] != 0.0 || dists_b[0] != 0.0) return fail(LCHD_EVALUE, "The dists list must start with a distance of 0!");
    // environments of more than 65 535 points: the 64-bit-count form of the wide sweep (8-bit category ids, 24-bit lengths)
    if ((len_seq_a > 65535 || len_seq_b > 65535) && (cfg->n_categories > kMaxCategories || len_seq_a >= (1 << 24) || len_seq_b >= (1 << 24)))
        return fail(LCHD_EUNSUPPORTED, "environments of more than 65535 points are supported with at most %d categories and fewer than 2^24 points", kMaxCategories);
    bool ascending = true;  // (the reference never checks: lists that do not ascend take the literal walk of its loop, below)
    for (int side = 0; side < 2; ++side) {
        const double* d = side ? dists_b : dists_a;
        const int64_t n = side ? len_dists_b : len_dists_a;
        for (int64_t i = 0; i < n; ++i) {
            if (std::isnan(d[i])) return fail(LCHD_EPANIC, "internal error: entered unreachable code (NaN distance)");
            if (d[i] < 0.0) return fail(LCHD_EVALUE, "Invalid input value: %g. All values must be non-negative!", d[i]);
            if (i && d[i] < d[i - 1]) ascending = false;
        }
    }
    if (!ascending && (len_seq_a > (1 << 24) || len_seq_b > (1 << 24)))
        return fail(LCHD_EUNSUPPORTED, "lists whose distances do not ascend are walked by one lane: at most 2^24 points each");
    // pmf.rs:38-42: every point of both lists enters a PMF.  (The environment kernels make this check for the other entry
    // points; here the lists go to the sweep kernel as they are, and that kernel does not test categories.)
    for (int side = 0; side < 2; ++side) {
        const int32_t* q = side ? seq_b : seq_a;
        const int64_t n = side ? len_seq_b : len_seq_a;
        for (int64_t i = 0; i < n; ++i)
            if (q[i] < 0 || q[i] >= cfg->n_categories) return fail(LCHD_EVALUE, "Category not found!");
    }
    // The two lists, the weight-function index and the score travel through the context's pinned staging block: ONE
    // asynchronous copy on the context's stream in (ordered with everything else on that stream), the score is stored by the
    // sweep kernel straight into the pinned block.
    const size_t na = (size_t)len_seq_a, nb = (size_t)len_seq_b;
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) & ~size_t(255); const size_t o = off; off += bytes; return o; };
    const bool cat16 = cfg->n_categories > kMaxCategories;  // 16-bit category ids in the two lists
    const size_t cb_ = cat16 ? 2 : 1;
    const size_t o_ka = take(8 * na), o_kb = take(8 * nb), o_ca = take(cb_ * na), o_cb = take(cb_ * nb), o_la = take(4), o_lb = take(4), o_wf = take(4);
    const size_t in_bytes = off;
    const size_t o_meta = take(sizeof(int4)), o_out = take(sizeof(double));
    if (int rc = grow_io(c, off)) return rc;
    auto stage = [&](size_t o_k, size_t o_c, size_t o_l, const int32_t* seq, const double* d, size_t n) {
        uint64_t* k = reinterpret_cast<uint64_t*>(c->h_io + o_k);
        uint8_t* c8 = reinterpret_cast<uint8_t*>(c->h_io + o_c);
        uint16_t* c16 = reinterpret_cast<uint16_t*>(c->h_io + o_c);
        for (size_t i = 0; i < n; ++i) {
            const double v = d[i] + 0.0;  // -0.0 -> +0.0
            memcpy(&k[i], &v, 8);
            if (cat16) c16[i] = (uint16_t)seq[i]; else c8[i] = (uint8_t)seq[i];  // inside [0, n_categories): checked above
        }
        const int32_t len = (int32_t)n;
        memcpy(c->h_io + o_l, &len, 4);
    };
    stage(o_ka, o_ca, o_la, seq_a, dists_a, na);
    stage(o_kb, o_cb, o_lb, seq_b, dists_b, nb);
    memcpy(c->h_io + o_wf, &wf_index, 4);
    EnvStore ea{}, eb{};
    ea.key = reinterpret_cast<uint64_t*>(c->d_io + o_ka); ea.cat = reinterpret_cast<uint8_t*>(c->d_io + o_ca);
    ea.len = reinterpret_cast<int32_t*>(c->d_io + o_la); ea.stride = len_seq_a;
    eb.key = reinterpret_cast<uint64_t*>(c->d_io + o_kb); eb.cat = reinterpret_cast<uint8_t*>(c->d_io + o_cb);
    eb.len = reinterpret_cast<int32_t*>(c->d_io + o_lb); eb.stride = len_seq_b;
    ea.cat16 = eb.cat16 = cat16 ? 1 : 0;
    hipStream_t s = c->stream;
    if (int rc = begin_pass(c)) return rc;
    c->status_dirty = true;
    HIP_TRY(hipMemcpyAsync(c->d_io, c->h_io, in_bytes, hipMemcpyHostToDevice, s));
    mark(c, 2);
    mark(c, 3);
    double* h_out = reinterpret_cast<double*>(c->h_io + o_out);
    if (!ascending) {
        // src/locohd.rs:97-221 on lists that do not ascend: no sort-based kernel computes what that loop computes; one lane walks it
        c->status_dirty = false;
        if (cfg->n_categories > 2000)
            return fail(LCHD_EUNSUPPORTED, "lists whose distances do not ascend are walked with the count vectors in LDS: at most 2000 categories (got %d)",
                        cfg->n_categories);
        launch_anchors_literal(s, c->d_cfg, cfg->n_categories, ea, eb, (int)len_seq_a, (int)len_seq_b, wf_index, h_out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        *out = *h_out;
        return LCHD_OK;
    }
    // a single pair: its weight-function index travels as a 1-element device array only when it is not 0
    int rc = sweep_rows(c, ea, eb, wf_index != 0 ? reinterpret_cast<const int32_t*>(c->d_io + o_wf) : nullptr, 1, h_out,
                        reinterpret_cast<int4*>(c->d_io + o_meta), DRV_ANCHORS);
    if (!rc) *out = *h_out;
    return rc;
}

// lchd_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the LoCoHD scoring path.
//
// Pipeline for one from_primitives call (reference: /root/reference/src/locohd.rs:479-567):
//
//   K0  cell list           regular batches / small structures: one workgroup per structure (histogram + scan + scatter in
//                           LDS: k_prologue_fused, k_cells_struct2); otherwise k_prep_count / k_prep_scan / k_prep_scatter.
//                           Atoms are permuted into cell order as 32-byte records {x, y, z, tag, cat}
//                           (replaces KdTree::build_by_ordered_float, :504-510); batches of structures carry the structure id
//                           as the slowest grid dimension
//   K0' anchor de-dup       in the same launches (anchor bit sets, popcount scan, compaction): an anchor that occurs in
//                           many pairs gets its environment built once; every
//                           unique anchor gets a 40-byte record (coordinates, tag, position in cell order, structure)
//   K1  environment build   k_env_cells<NT,TAGLIST>, both structures in one launch: radius search over the concatenated
//                           neighbour-cell runs (full wavefronts of candidates, two 16-byte loads per candidate, no dependent
//                           loads) + tag filter + distances, then an O(n) bucket sort (d^3 buckets, LDS histogram + scan +
//                           scatter + per-lane insertion sort) for environments of <= 512 points, LDS bitonic network
//                           otherwise; optional CDF keying (replaces env_from_idx :514-542, utils::sort_together utils.rs:25-39)
//       dense variant       k_env_rows<NT,GLOBALKV>: whole cloud / given distance-matrix row (from_coords, from_dmxs),
//                           bucket sort on the row's empirical distance CDF
//   K2' pair records        k_pair_meta: one 16-byte record per pair {slot A, slot B, n_A | cat, n_B | cat}
//                           and the number of pairs small enough for k_sweep_duo
//   K2  sweep               k_sweep<CMAX,MODE,FMODE,LDSTAB,INDIRECT>: merge-path partition of the two sorted environments,
//                           per-lane sequential sweep with a wavefront DPP prefix scan of packed category counts,
//                           statistical distance per breakpoint, CDF differences, DPP reduction
//                           (replaces stat_dist_integral :61-226, pmf.rs, statistical_distances.rs, cdfs.rs);
//                           k_sweep_duo<CMAX,TL,TILE>: the pairs that fit one tile, four (<= 240 events) or two (8-bit counts,
//                           <= 480 events) per wavefront in teams of TL lanes -- the device or the host's hint decides which
//                           rule is in force; k_sweep_wide for 33..512 categories and for environments beyond 65535 points
//       trajectory frames   k_frames_labels / k_frames_unpack: SoA unpack + bounding box of a block of frames;
//                           k_frames_centroids (+ k_bbox_finish): primitive atoms of every frame from its float32 source atoms
//
// One wavefront owns one environment (K1: a 64-thread workgroup) or one anchor pair (K2: four pairs per
// 256-thread workgroup; k_sweep_duo: eight or sixteen); a launch has thousands of independent wavefronts, so all 256 CUs / 8 XCDs are
// filled without any inter-workgroup communication.  All arithmetic is f64 like the reference; no MFMA
// (there is no contraction in this path).
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "lchd_device.h"
#include "lchd_math.h"
#include "lchd_kcommon.h"
#include "lchd_team_tile.h"

#ifndef LCHD_ENV_FLAT
#define LCHD_ENV_FLAT 4   // steps of 64 candidates whose record loads are issued together in the radius search
#endif

namespace lchd {

// ------------------------------------------------------------------------------------------------
// K0: uniform grid.  Points keep their f64 coordinates; only the bucketing uses the grid.
// ------------------------------------------------------------------------------------------------
// Exclusive scan of n u32 by ONE 1024-thread workgroup (n is a cell or atom count: small). out[n] = total.
// In-place (out == in) is allowed.
__global__ __launch_bounds__(1024) void k_exclusive_scan(const uint32_t* in, uint32_t* out, int n, uint32_t* total_out) {
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const uint32_t v = (i < n) ? in[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint32_t wpre = 0;
        for (int w = 0; w < wave; ++w) wpre += wave_sum[w];
        const uint32_t carry = carry_s;
        if (i < n) out[i] = carry + wpre + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + wpre + incl;
        __syncthreads();
    }
    if (tid == 0) {
        out[n] = carry_s;
        if (total_out) *total_out = carry_s;
    }
}

// Multi-block exclusive scan for large inputs (batches of structures: millions of atoms / cells):
// per-block sums -> single-workgroup scan of the sums -> per-block scan with the block's offset.  4096 items per block.
constexpr int kScanItems = 4096;
__global__ __launch_bounds__(1024) void k_scan_block_sums(const uint32_t* in, int n, uint32_t* bsum) {
    __shared__ uint32_t ws[16];
    const int tid = threadIdx.x, base = blockIdx.x * kScanItems;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + k * 1024 + tid; v += i < n ? in[i] : 0u; }
    for (int m = 32; m > 0; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
    if ((tid & 63) == 0) ws[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += ws[w]; bsum[blockIdx.x] = t; }
}
__global__ __launch_bounds__(1024) void k_scan_apply(const uint32_t* in, uint32_t* out, int n, const uint32_t* bpre, int n_blocks,
                                                     uint32_t* total_out) {
    __shared__ uint32_t ws[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, base = blockIdx.x * kScanItems;
    // thread t owns items base + 4t .. base + 4t + 3 (blocked), so one wave scan + a 16-entry LDS pass suffice
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + 4 * tid + k; v[k] = i < n ? in[i] : 0u; sum += v[k]; }
    const uint32_t incl = wave_incl_scan_u32(sum);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t pre = bpre[blockIdx.x] + incl - sum;
    for (int w = 0; w < wave; ++w) pre += ws[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + 4 * tid + k; if (i < n) out[i] = pre; pre += v[k]; }
    if (blockIdx.x == n_blocks - 1 && tid == 1023) {
        out[n] = pre;
        if (total_out) *total_out = pre;
    }
}
static void launch_exclusive_scan(hipStream_t s, const uint32_t* in, uint32_t* out, int n, uint32_t* total_out, uint32_t* tmp) {
    const int nb = (n + kScanItems - 1) / kScanItems;
    k_scan_block_sums<<<nb, 1024, 0, s>>>(in, n, tmp);
    k_exclusive_scan<<<1, 1024, 0, s>>>(tmp, tmp, nb, nullptr);
    k_scan_apply<<<nb, 1024, 0, s>>>(in, out, n, tmp, nb, total_out);
}

// ------------------------------------------------------------------------------------------------
// Prologue of a from_primitives pass: cell lists of both structures + anchor de-duplication, in as few launches as the sizes
// allow (a structure pair of a few thousand atoms spends more time between kernels than inside them):
//   fused     both sides single structures of <= kStructAtomsMax atoms, <= kFusedPairsMax pairs: ONE launch, workgroup 0 =
//             side A, workgroup 1 = side B: cell list in LDS (histogram with returned ranks, scan, scatter), anchor flags as
//             an LDS bit set, scan, anchor records
//   struct    equal-sized structures that fit LDS (trajectory frames, regular batches, one medium structure): one workgroup
//             per structure, both sides in one launch, which also zeroes the anchor flags
//   general   three launches, each parallel over the atoms / pairs of both sides: k_prep_count (cell + rank inside it through
//             the returning atomic on the cell counter; anchors into two bit sets), k_prep_scan (one workgroup per side),
//             k_prep_scatter (records into cell order, anchor slots + records); the caller zeroes counters and bit sets with
//             ONE memset.  After a struct launch the same three kernels only do the anchor half of their work.
// ------------------------------------------------------------------------------------------------
constexpr int kStructCellsMax = 4096, kStructAtomsMax = 12000;  // 16 KB + 48 KB of dynamic LDS stay under the 64 KB launch limit
constexpr int kFusedPairsMax = 1 << 16;                         // one workgroup per side reads the whole pair list
constexpr int kBitWordsMax = (kStructAtomsMax + 31) / 32;       // anchor flags of one side as a bit set

// One workgroup of NT threads builds the cell list of ONE structure of `size` atoms starting at atom `base`, entirely in
// LDS -- histogram with returned ranks, scan, scatter -- instead of the five global passes (two of them with one global
// atomic per atom) of the generic path.  The structure owns cells [cell_base, cell_base + cps) and records
// [base, base + size).  smem: hist[cps] u32 | cid[size] u16 | rank[size] u16; on return hist[] holds the first slot of
// every cell (relative to `base`) and cid / rank are intact, so position(atom a) = base + hist[cid[a]] + rank[a].
// category id of atom i: low byte | high byte, or -- a structure without high bytes -- the byte itself, its "not in the map" value
// 255 widened to 0xFFFF (a configuration with more than 255 categories has a category 255)
__device__ __forceinline__ uint32_t cat_of_atom(const CloudView& c, int64_t i) {
    const uint32_t lo = c.cat[i];
    return c.cat_hi ? (lo | ((uint32_t)c.cat_hi[i] << 8)) : (lo == 255u ? 0xFFFFu : lo);
}
template <int NT>
__device__ __forceinline__ void cell_build_wg(const CloudView& c, const GridView& g, int cps, int64_t base, int size, int64_t cell_base,
                                              bool write_end, CellRec* __restrict__ rec, uint32_t* __restrict__ pos_of,
                                              uint32_t* __restrict__ cell_start, unsigned char* smem, uint32_t* wsum /* [NT / 64] */) {
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    uint16_t* cid = reinterpret_cast<uint16_t*>(smem + (size_t)cps * 4);
    uint16_t* rank_ = cid + size;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < cps; k += NT) hist[k] = 0u;
    __syncthreads();
    // (four atoms per thread and step: their coordinate loads are in flight together -- a single workgroup walking a
    // structure of ten thousand atoms is bound by memory latency, not by bandwidth)
    constexpr int U = 4;
    for (int a0 = tid; a0 < size; a0 += U * NT) {
        double X[U], Y[U], Z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + min(a0 + u * NT, size - 1);
            X[u] = c.x[i]; Y[u] = c.y[i]; Z[u] = c.z[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int a = a0 + u * NT;
            if (a < size) {
                const int cx = cell_coord(X[u], g.min[0], g.inv[0], g.dim[0]);
                const int cy = cell_coord(Y[u], g.min[1], g.inv[1], g.dim[1]);
                const int cz = cell_coord(Z[u], g.min[2], g.inv[2], g.dim[2]);
                const int cell = (cz * g.dim[1] + cy) * g.dim[0] + cx;
                cid[a] = (uint16_t)cell;
                rank_[a] = (uint16_t)atomicAdd(&hist[cell], 1u);
            }
        }
    }
    __syncthreads();
    // exclusive scan of hist[0 .. cps): thread t owns the consecutive entries [t * per, (t + 1) * per)
    const int per = (cps + NT - 1) / NT, lo = min(tid * per, cps), hi = min(lo + per, cps);
    uint32_t sum = 0;
    for (int k = lo; k < hi; ++k) sum += hist[k];
    const uint32_t incl = wave_incl_scan_u32(sum);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - sum;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    for (int k = lo; k < hi; ++k) {
        const uint32_t h = hist[k];
        hist[k] = pre;
        cell_start[cell_base + k] = (uint32_t)base + pre;
        pre += h;
    }
    if (write_end && tid == NT - 1) cell_start[cell_base + cps] = (uint32_t)(base + size);
    __syncthreads();
    for (int a0 = tid; a0 < size; a0 += U * NT) {
        CellRec r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + min(a0 + u * NT, size - 1);
            r[u].x = c.x[i];
            r[u].y = c.y[i];
            r[u].z = c.z[i];
            r[u].tag = (uint32_t)c.tag[i];
            r[u].cat = cat_of_atom(c, i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int a = a0 + u * NT;
            if (a < size) {
                const uint32_t pos = (uint32_t)base + hist[cid[a]] + rank_[a];
                rec[pos] = r[u];
                pos_of[base + a] = pos;
            }
        }
    }
}

// Anchor flags of one side as an LDS bit set (bits[w] bit k <=> atom 32 w + k is an anchor) -> environment slots and anchor
// records.  wpre [nw + 1] receives the exclusive prefix of the per-word counts.  `apos_of(i)` = position of atom i in cell
// order.  Only the slots of anchors are written (nothing reads the others).  nw <= NT.
template <int NT, class F>
__device__ __forceinline__ void dedupe_finish_wg(const uint32_t* bits, uint32_t* wpre, int nw, const CloudView& c, uint32_t* __restrict__ slot,
                                                 AnchorRec* __restrict__ uniq, uint32_t* n_unique_out, uint32_t* wsum /* [NT / 64] */, F apos_of) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t v = tid < nw ? (uint32_t)__popc(bits[tid]) : 0u;
    const uint32_t incl = wave_incl_scan_u32(v);
    __syncthreads();  // wsum may still be read by the caller's previous phase
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - v;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (tid < nw) wpre[tid] = pre;
    if (tid == nw - 1) { wpre[nw] = pre + v; *n_unique_out = pre + v; slot[c.n] = pre + v; }
    __syncthreads();
    // four atoms per thread and step, their loads in flight together (one workgroup, latency-bound: see cell_build_wg)
    constexpr int U = 4;
    for (int i0 = tid; i0 < c.n; i0 += U * NT) {
        AnchorRec r[U];
        uint32_t sl[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * NT, c.n - 1);
            const uint32_t w = bits[i >> 5];
            on[u] = (i0 + u * NT < c.n) && ((w >> (i & 31)) & 1u);
            sl[u] = wpre[i >> 5] + (uint32_t)__popc(w & ((1u << (i & 31)) - 1u));
            r[u].x = c.x[i]; r[u].y = c.y[i]; r[u].z = c.z[i];
            r[u].tag = (uint32_t)c.tag[i];
            r[u].apos = apos_of(i);
            r[u].sid = c.sid ? c.sid[i] : 0;
            r[u].atom = (uint32_t)i;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (on[u]) {
                slot[i0 + u * NT] = sl[u];
                uniq[sl[u]] = r[u];
            }
    }
}

// fused: see the section header.  Dynamic LDS: max over the sides of (cells * 4 + atoms * 4) bytes.
__global__ __launch_bounds__(1024) void k_prologue_fused(const int64_t* __restrict__ anchors, int64_t n_pairs, PrepSide pa, PrepSide pb,
                                                          DeviceStatus* st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pf[];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t bits[kBitWordsMax + 1], wpre[kBitWordsMax + 2];
    const int side = blockIdx.x, tid = threadIdx.x;
    const PrepSide& P = side ? pb : pa;
    const CloudView c = P.c;
    const int n = c.n, cps = P.g.dim[0] * P.g.dim[1] * P.g.dim[2], nw = (n + 31) >> 5;
    for (int w = tid; w <= kBitWordsMax; w += 1024) bits[w] = 0u;
    cell_build_wg<1024>(c, P.g, cps, 0, n, 0, true, P.rec, P.pos_of, P.cell_start, smem_pf, wsum);  // (its barriers order the clear above)
    if (P.no_anchors) {  // (k_env_sweep validates this side's anchor indices itself)
        if (tid == 0) st->n_unique[side] = 0u;
        return;
    }
    bool bad = false;
    for (int64_t p0 = tid; p0 < n_pairs; p0 += 4 * 1024) {
        int64_t av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = anchors[2 * min(p0 + u * 1024, n_pairs - 1) + side];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (p0 + u * 1024 < n_pairs) {
                const int64_t a = av[u];
                if (a < 0 || a >= n) bad = true;
                else atomicOr(&bits[a >> 5], 1u << (a & 31));
            }
    }
    if (__ballot(bad) && (tid & 63) == 0) atomicOr(&st->flags, ST_BAD_ANCHOR);
    __syncthreads();
    const uint32_t* hist = reinterpret_cast<const uint32_t*>(smem_pf);
    const uint16_t* cid = reinterpret_cast<const uint16_t*>(smem_pf + (size_t)cps * 4);
    const uint16_t* rank_ = cid + n;
    dedupe_finish_wg<1024>(bits, wpre, nw, c, P.slot, P.uniq, &st->n_unique[side], wsum,
                           [&](int i) { return hist[cid[i]] + (uint32_t)rank_[i]; });
}

// struct: one workgroup per structure, both sides; workgroups past the structures zero `zero_words` u32 at zero_base
// (the anchor flags of the de-duplication that follows).
template <int NT>
__global__ __launch_bounds__(NT) void k_cells_struct2(PrepSide pa, PrepSide pb, int nsa, int nsb, uint32_t* __restrict__ zero_base,
                                                      int64_t zero_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cb[];
    __shared__ uint32_t wsum[NT / 64];
    const int b = blockIdx.x;
    if (b >= nsa + nsb) {
        const int64_t nz = (int64_t)gridDim.x - nsa - nsb;
        for (int64_t i = (int64_t)(b - nsa - nsb) * NT + threadIdx.x; i < zero_words; i += nz * NT) zero_base[i] = 0u;
        return;
    }
    const int side = b >= nsa ? 1 : 0, k = side ? b - nsa : b;
    const PrepSide& P = side ? pb : pa;
    const int size = P.c.struct_size, cps = P.g.dim[0] * P.g.dim[1] * P.g.dim[2];
    const int ns = side ? nsb : nsa;
    cell_build_wg<NT>(P.c, P.g, cps, (int64_t)k * size, size, (int64_t)k * cps, k == ns - 1, P.rec, P.pos_of, P.cell_start, smem_cb, wsum);
}

// ---- the general prologue: three launches, each parallel over atoms / pairs of BOTH sides ------------------------------
//   k_prep_count    atoms: cell of every atom + its rank inside the cell (the returning atomic on the cell's counter);
//                   pairs: one byte flag per anchor, PLAIN stores (a million pairs over ten thousand atoms hammer a
//                   handful of cache lines: as atomics on a bit set they serialise at the memory side -- 0.9 ms --, as
//                   plain stores every XCD's L2 absorbs its share)
//   k_prep_scan     one workgroup per side: exclusive scan of the cell counters (a few thousand cells; batches with more
//                   than kPrepScanCells cells take the multi-block scan), byte flags -> bit set, scan of the words'
//                   popcounts (sides of more than kPrepScanAtoms atoms: k_prep_bits first, one workgroup per chunk)
//   k_prep_scatter  atoms: record into cell order, atom -> position; anchors: environment slot + anchor record
// (the struct path builds the cell lists in k_cells_struct2 and skips the atom halves of k_prep_count / k_prep_scan)
constexpr int kPrepScanCells = 1 << 16;  // cells one workgroup scans (64 per thread)
constexpr int kPrepScanAtoms = 1 << 18;  // atoms whose byte flags one workgroup turns into the bit set (256 KB through one CU: ~10 us)
__global__ void k_prep_count(const int64_t* __restrict__ anchors, int64_t n_pairs, PrepSide pa, PrepSide pb, int cells_a, int cells_b,
                             DeviceStatus* st) {
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x, g0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // atoms of side A, then of side B (a side whose cell list the struct path builds is skipped: cells_x == 0)
    const int64_t na = cells_a ? pa.c.n : 0, nb = cells_b ? pb.c.n : 0;
    for (int64_t t = g0; t < na + nb; t += gsz) {
        const bool sb_ = t >= na;
        const PrepSide& P = sb_ ? pb : pa;
        const int64_t i = sb_ ? t - na : t;
        const GridView& g = P.g;
        const int cx = cell_coord(P.c.x[i], g.min[0], g.inv[0], g.dim[0]);
        const int cy = cell_coord(P.c.y[i], g.min[1], g.inv[1], g.dim[1]);
        const int cz = cell_coord(P.c.z[i], g.min[2], g.inv[2], g.dim[2]);
        const int sid = P.c.sid ? P.c.sid[i] : 0;
        const uint32_t cell = (uint32_t)((((int64_t)sid * g.dim[2] + cz) * g.dim[1] + cy) * g.dim[0] + cx);
        P.cell_of[i] = cell;
        P.pos_of[i] = atomicAdd(&P.cell_count[cell], 1u);  // rank inside the cell, replaced by the position in k_prep_scatter
    }
    bool bad = false;
    const int32_t n_a = pa.c.n, n_b = pb.c.n;
    for (int64_t p = g0; p < n_pairs; p += gsz) {
        const longlong2 ab = reinterpret_cast<const longlong2*>(anchors)[p];
        if (ab.x < 0 || ab.x >= n_a) bad = true; else pa.flag8[ab.x] = 1;
        if (ab.y < 0 || ab.y >= n_b) bad = true; else if (!pb.no_anchors) pb.flag8[ab.y] = 1;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(&st->flags, ST_BAD_ANCHOR);
}

// exclusive scan of n u32 by the calling 1024-thread workgroup; out[n] = total.  Chunks of 4096 items are staged through LDS:
// coalesced loads, every thread scans four consecutive LDS entries, coalesced stores.  (Each thread walking its own run of
// consecutive items in global memory -- the first version -- is a chain of dependent, uncoalesced loads: 50 us for the 3 x 10^4
// cells of a 2 x 10^5-atom structure, more than the two streaming kernels around it together.)  In place (out == in) is allowed.
__device__ __forceinline__ uint32_t scan_wg_1024(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n, bool popcount,
                                                 uint32_t* wsum /* [16] */) {
    __shared__ uint32_t stage[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t carry = 0;
    for (int64_t c0 = 0; c0 < n; c0 += 4096) {
        __syncthreads();  // (stage / wsum of the previous chunk have been read)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = c0 + tid + 1024 * k;
            const uint32_t v = i < n ? in[i] : 0u;
            stage[tid + 1024 * k] = popcount ? (uint32_t)__popc(v) : v;
        }
        __syncthreads();
        const uint4 v4 = reinterpret_cast<const uint4*>(stage)[tid];
        const uint32_t sum = v4.x + v4.y + v4.z + v4.w;
        const uint32_t incl = wave_incl_scan_u32(sum);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t pre = carry + incl - sum, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) pre += wsum[w]; total += wsum[w]; }
        reinterpret_cast<uint4*>(stage)[tid] = make_uint4(pre, pre + v4.x, pre + v4.x + v4.y, pre + v4.x + v4.y + v4.z);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = c0 + tid + 1024 * k;
            if (i < n) out[i] = stage[tid + 1024 * k];
        }
        carry += total;
    }
    __syncthreads();
    if (tid == 1023) out[n] = carry;
    return carry;
}
// 32 byte flags -> one word of the bit set (the flag array is padded to a multiple of 32 bytes, 16-byte aligned)
__device__ __forceinline__ uint32_t flags_word(const uint8_t* __restrict__ flag8, int64_t w) {
    const uint4 lo = reinterpret_cast<const uint4*>(flag8)[2 * w], hi = reinterpret_cast<const uint4*>(flag8)[2 * w + 1];
    auto nib = [](uint32_t v) -> uint32_t {  // four byte flags (0 / 1) -> four bits
        return (v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u);
    };
    return nib(lo.x) | (nib(lo.y) << 4) | (nib(lo.z) << 8) | (nib(lo.w) << 12) | (nib(hi.x) << 16) | (nib(hi.y) << 20) | (nib(hi.z) << 24) |
           (nib(hi.w) << 28);
}
// Sides of more than kPrepScanAtoms atoms (trajectory batches: millions of atoms): one workgroup per chunk of
// kPrepScanAtoms atoms turns the chunk's byte flags into bit-set words, scans the words' popcounts inside the chunk (wpre =
// anchors before the word WITHIN its chunk) and leaves the chunk's total in chunk_base[chunk]; k_prep_scan then only scans the
// chunk totals.  (One workgroup walking 80 000 words took longer than the cell lists of the whole batch.)
constexpr int kChunkWords = kPrepScanAtoms / 32;  // 8192 words, 8 consecutive ones per thread
__global__ __launch_bounds__(1024) void k_prep_bits(PrepSide pa, PrepSide pb, int chunks_a) {
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool sb_ = (int)blockIdx.x >= chunks_a;
    const PrepSide& P = sb_ ? pb : pa;
    const int chunk = sb_ ? blockIdx.x - chunks_a : blockIdx.x;
    const int64_t nw = ((int64_t)P.c.n + 31) >> 5, w0 = (int64_t)chunk * kChunkWords + 8 * tid;
    uint32_t bw[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bw[k] = w0 + k < nw ? flags_word(P.flag8, w0 + k) : 0u;
        sum += (uint32_t)__popc(bw[k]);
    }
    const uint32_t incl = wave_incl_scan_u32(sum);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - sum, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) pre += wsum[w]; total += wsum[w]; }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (w0 + k < nw) {
            P.bits[w0 + k] = bw[k];
            P.wpre[w0 + k] = pre;
            pre += (uint32_t)__popc(bw[k]);
        }
    if (tid == 0) P.chunk_base[chunk] = total;
}
__global__ __launch_bounds__(1024) void k_prep_scan(PrepSide pa, PrepSide pb, int cells_a, int cells_b, int bits_ready, DeviceStatus* st) {
    __shared__ uint32_t wsum[16];
    const int side = blockIdx.x;
    const PrepSide& P = side ? pb : pa;
    const int cells = side ? cells_b : cells_a;
    if (cells > 0 && cells <= kPrepScanCells) scan_wg_1024(P.cell_count, P.cell_start, cells, false, wsum);
    if (P.no_anchors) {
        if (threadIdx.x == 0) st->n_unique[side] = 0u;
        return;
    }
    const int64_t nw = ((int64_t)P.c.n + 31) >> 5;
    if (bits_ready) {  // k_prep_bits has done the words and the scans inside the chunks: only the chunk totals are left
        const int64_t n_chunks = (nw + kChunkWords - 1) / kChunkWords;
        const uint32_t total = scan_wg_1024(P.chunk_base, P.chunk_base, n_chunks, false, wsum);
        if (threadIdx.x == 0) st->n_unique[side] = total;
        return;
    }
    for (int64_t w = threadIdx.x; w < nw; w += 1024) P.bits[w] = flags_word(P.flag8, w);
    __syncthreads();  // (the scan below reads words other threads of this workgroup wrote)
    if (threadIdx.x == 0) P.chunk_base[0] = 0u;
    const uint32_t total = scan_wg_1024(P.bits, P.wpre, nw, true, wsum);
    if (threadIdx.x == 0) st->n_unique[side] = total;
}
__global__ void k_prep_scatter(PrepSide pa, PrepSide pb, int cells_a, int cells_b) {
    const int64_t gsz = (int64_t)gridDim.x * blockDim.x, g0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t na = pa.c.n, nb = pb.c.n;
    for (int64_t t = g0; t < na + nb; t += gsz) {
        const bool sb_ = t >= na;
        const PrepSide& P = sb_ ? pb : pa;
        const int64_t i = sb_ ? t - na : t;
        const CloudView& c = P.c;
        const bool general = (sb_ ? cells_b : cells_a) != 0;
        const uint32_t w = P.no_anchors ? 0u : P.bits[i >> 5];
        const bool anchor = (w >> (i & 31)) & 1u;
        if (!general && !anchor) continue;  // (struct path, not an anchor: nothing to do -- most atoms of a trajectory batch)
        const double x = c.x[i], y = c.y[i], z = c.z[i];
        const uint32_t tag = (uint32_t)c.tag[i];
        uint32_t pos = P.pos_of[i];
        if (general) {  // general cell list: rank inside the cell -> position, record into cell order
            pos += P.cell_start[P.cell_of[i]];
            CellRec r;
            r.x = x; r.y = y; r.z = z;
            r.tag = tag;
            r.cat = cat_of_atom(c, i);
            P.rec[pos] = r;
            P.pos_of[i] = pos;
        }
        if (anchor) {  // its environment slot and its record
            const uint32_t sl = P.chunk_base[i >> 18] + P.wpre[i >> 5] + (uint32_t)__popc(w & ((1u << (i & 31)) - 1u));
            P.slot[i] = sl;
            AnchorRec r;
            r.x = x; r.y = y; r.z = z;
            r.tag = tag;
            r.apos = pos;
            r.sid = c.sid ? c.sid[i] : 0;
            r.atom = (uint32_t)i;
            P.uniq[sl] = r;
        }
    }
}

// Side B without de-duplication (PrepSide::no_anchors: (almost) every anchor of the side occurs in ONE pair -- the frames of a
// trajectory, (i, i) lists, a rank's partners under strong scaling): environment slot p belongs to pair p, and its anchor record is
// written straight from the pair list -- no byte flags, no bit set, no scan, no scatter over the side's atoms (C4: 54 -> ~10 us per
// pass).  An anchor that does occur in several pairs is built once per pair, as the reference does (src/locohd.rs:514-554).
__global__ void k_pair_anchor_recs(const int64_t* __restrict__ anchors, int64_t n_pairs, PrepSide pb, DeviceStatus* st) {
    const CloudView& c = pb.c;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = anchors[2 * p + 1];
        if (i < 0 || i >= c.n) i = 0;  // (flagged by k_prep_count; the pair record marks the pair unusable)
        AnchorRec r;
        r.x = c.x[i]; r.y = c.y[i]; r.z = c.z[i];
        r.tag = (uint32_t)c.tag[i];
        r.apos = pb.pos_of[i];
        r.sid = c.sid ? c.sid[i] : 0;
        r.atom = (uint32_t)i;
        pb.uniq[p] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->n_unique[1] = (uint32_t)n_pairs;
}
void launch_pair_anchor_recs(hipStream_t s, const int64_t* anchors, int64_t n_pairs, const PrepSide& b, DeviceStatus* st) {
    if (n_pairs <= 0) return;
    const int64_t nb = (n_pairs + 255) / 256;
    k_pair_anchor_recs<<<(unsigned)std::min<int64_t>(nb, 4096), 256, 0, s>>>(anchors, n_pairs, b, st);
}

static bool fits_struct_path(const PrepSide& P, const Tuning& t, CloudView& cs) {
    cs = P.c;
    if (!P.c.sid) { cs.struct_size = P.c.n; cs.n_struct = 1; }
    const int cps = P.g.dim[0] * P.g.dim[1] * P.g.dim[2];
    // (a single structure of more than 4096 atoms is faster through the three parallel launches than through one
    // workgroup, whose memory pipe moves ~25 GB/s: ~50 us per 10^4 atoms)
    return !t.no_struct_cells && cs.struct_size > 0 && cs.struct_size <= kStructAtomsMax && cps <= kStructCellsMax &&
           (int64_t)cs.n_struct * cs.struct_size == P.c.n && (cs.n_struct >= 8 || cs.struct_size <= 4096);
}

#ifndef LCHD_STRUCT_NT
#define LCHD_STRUCT_NT 512   // (measured, C4 cell lists: 128 0.506, 256 0.466, 512 0.445 ms per step) threads of the per-structure cell-list workgroups of a batch of more than 16 structures
#endif
int launch_prologue(hipStream_t s, const Tuning& t, const int64_t* anchors, int64_t n_pairs, const PrepSide& a_in, const PrepSide& b_in,
                    void* zero_base, size_t zero_bytes, DeviceStatus* st, bool same) {
    PrepSide a = a_in, b = b_in;
    CloudView csa, csb;
    bool fa = fits_struct_path(a, t, csa), fb = fits_struct_path(b, t, csb);
    if (same) {
        // one object on both sides: column 1's anchors are flagged in side A's byte flags (k_prep_count validates them against
        // the same atom count), side B gets no cell list (fb: "already built") and, seen as a structure of 0 atoms by the
        // scan and scatter kernels, no slots and no records
        b.flag8 = a.flag8;
        fb = true;
    }
    const int cps_a = a.g.dim[0] * a.g.dim[1] * a.g.dim[2], cps_b = b.g.dim[0] * b.g.dim[1] * b.g.dim[2];
    int ops = 0;
    if (!same && fa && fb && csa.n_struct == 1 && csb.n_struct == 1 && n_pairs <= kFusedPairsMax && !t.no_small_dedupe && a.c.n > 0 && b.c.n > 0) {
        const size_t lds = std::max((size_t)cps_a * 4 + (size_t)a.c.n * 4, (size_t)cps_b * 4 + (size_t)b.c.n * 4);
        k_prologue_fused<<<2, 1024, lds, s>>>(anchors, n_pairs, a, b, st);
        return 1;
    }
    // the anchor flags (and, for the general cell list, its counters) must be zero: folded into the struct launch when both
    // sides take it, otherwise ONE memset over the contiguous region the caller laid out
    const bool fold_zero = fa && fb;  // (same: fb is true by definition, so side A decides)
    if (!fold_zero) { (void)hipMemsetAsync(zero_base, 0, zero_bytes, s); ++ops; }
    if (fa || (fb && !same)) {
        PrepSide sa_ = a, sb_ = b;
        sa_.c = csa; sb_.c = csb;
        const int nsa = fa ? csa.n_struct : 0, nsb = (fb && !same) ? csb.n_struct : 0;
        const size_t lds = std::max(fa ? (size_t)cps_a * 4 + (size_t)csa.struct_size * 4 : 0, (fb && !same) ? (size_t)cps_b * 4 + (size_t)csb.struct_size * 4 : 0);
        // the anchor flags sit at the END of the zero region: [.. counters ..][flags_a][flags_b]
        uint32_t* zb = fold_zero ? reinterpret_cast<uint32_t*>(a.flag8) : nullptr;
        const int64_t zw = fold_zero ? (int64_t)((reinterpret_cast<char*>(zero_base) + zero_bytes - reinterpret_cast<char*>(a.flag8)) / 4) : 0;
        if (nsa + nsb <= 16) {
            const int nz = fold_zero ? (int)std::min<int64_t>(64, (zw + 4095) / 4096) : 0;
            k_cells_struct2<1024><<<nsa + nsb + nz, 1024, lds, s>>>(sa_, sb_, nsa, nsb, zb, zw);
        } else {
            const int nz = fold_zero ? (int)std::min<int64_t>(1024, (zw + 1023) / 1024) : 0;
            k_cells_struct2<LCHD_STRUCT_NT><<<nsa + nsb + nz, LCHD_STRUCT_NT, lds, s>>>(sa_, sb_, nsa, nsb, zb, zw);
        }
        ++ops;
    }
    const int cells_a = fa ? 0 : a.g.n_cells, cells_b = fb ? 0 : b.g.n_cells;  // 0: the struct path has built that side's cell list
    const int64_t work = std::max<int64_t>((cells_a ? a.c.n : 0) + (int64_t)(cells_b ? b.c.n : 0), n_pairs);
    const int64_t nbk = (work + 255) / 256;
    k_prep_count<<<(unsigned)std::max<int64_t>(1, std::min<int64_t>(nbk, 8192)), 256, 0, s>>>(anchors, n_pairs, a, b, cells_a, cells_b, st);
    ++ops;
    if (same) b.c.n = 0;  // (for the scan / scatter kernels below: nothing to do on side B, n_unique[1] = 0)
    for (int side = 0; side < 2; ++side) {  // batches with more cells than one workgroup scans
        const PrepSide& P = side ? b : a;
        const int cells = side ? cells_b : cells_a;
        if (cells > kPrepScanCells) { launch_exclusive_scan(s, P.cell_count, P.cell_start, cells, nullptr, P.scan_tmp); ops += 3; }
    }
    const bool big = a.c.n > kPrepScanAtoms || (!b.no_anchors && b.c.n > kPrepScanAtoms);
    if (big) {
        const int ca = (int)((((int64_t)a.c.n + 31) / 32 + kChunkWords - 1) / kChunkWords),
                  cb = b.no_anchors ? 0 : (int)((((int64_t)b.c.n + 31) / 32 + kChunkWords - 1) / kChunkWords);
        k_prep_bits<<<ca + cb, 1024, 0, s>>>(a, b, ca);
        ++ops;
    }
    k_prep_scan<<<2, 1024, 0, s>>>(a, b, cells_a, cells_b, big ? 1 : 0, st);
    if (b.no_anchors && cells_b == 0) b.c.n = 0;  // (side B: no anchors, and the struct path has built its cell list: nothing left to scatter)
    const int64_t nba = ((int64_t)a.c.n + b.c.n + 255) / 256;
    k_prep_scatter<<<(unsigned)std::max<int64_t>(1, std::min<int64_t>(nba, 8192)), 256, 0, s>>>(a, b, cells_a, cells_b);
    return ops + 2;
}

// ------------------------------------------------------------------------------------------------
// Bitonic sort of (u64 key, u8 value) pairs resident in LDS by a workgroup of NT threads.
// Keys are f64 bit patterns of non-negative distances: unsigned integer order == numeric order.
// Equal keys may come out in any order: ties only ever produce zero-width intervals in the sweep
// (SURVEY.md section 0), so the score does not depend on it.
// ------------------------------------------------------------------------------------------------
template <int NT, class VT = uint8_t>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* key, VT* val, int n2, int tid) {
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += NT) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool up = ((i & k) == 0);
                const uint64_t a = key[i], b = key[l];
                if (up ? (a > b) : (a < b)) {
                    key[i] = b;
                    key[l] = a;
                    const VT va = val[i];
                    val[i] = val[l];
                    val[l] = va;
                }
            }
            __syncthreads();
        }
    }
}


// Single-weight-function configurations: replace the sorted distances by F(distance) so that the sweep kernel never
// evaluates a CDF (every pair that re-uses this environment would recompute the same values).  F is non-decreasing,
// so the order is unchanged; a running maximum removes last-bit inversions of the floating-point CDF (the merge in
// the sweep kernel needs sorted keys; equal F values are zero-width intervals and contribute exactly 0).
template <int NT>
__device__ __forceinline__ void keys_to_cdf_lds(uint64_t* key, int n, int tid, const DevConfig* __restrict__ cfg) {
    const WfEntry wf = cfg->wf[0];
    const double* __restrict__ prm = cfg->wf_params + wf.offset;
    const double winv = cfg->wf_inv[0];
    const int chunk = (n + NT - 1) / NT, lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    uint64_t m = 0;
    for (int i = lo; i < hi; ++i) {
        const uint64_t f = d2u(cdf_lean(wf.kind, prm, wf.n_params, winv, u2d(key[i])) + 0.0);
        m = f > m ? f : m;
        key[i] = m;
    }
    // exclusive prefix maximum of the per-thread maxima
    uint64_t incl = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = shfl_up_u64(incl, d);
        if ((tid & 63) >= d) incl = t > incl ? t : incl;
    }
    uint64_t excl = shfl_up_u64(incl, 1);
    if ((tid & 63) == 0) excl = 0;
    if constexpr (NT > 64) {
        __shared__ uint64_t wave_max[NT / 64];
        if ((tid & 63) == 63) wave_max[tid >> 6] = incl;
        __syncthreads();
        for (int w = 0; w < (tid >> 6); ++w) excl = wave_max[w] > excl ? wave_max[w] : excl;
    }
    for (int i = lo; i < hi; ++i) key[i] = key[i] > excl ? key[i] : excl;
    __syncthreads();
}

// The same for one wavefront: every lane converts its (strided) keys, then the wave checks that the result is still
// non-decreasing; the running maximum is only needed when the floating-point CDF produced a last-bit inversion, which a
// single lane then repairs in place (rare enough not to matter).
__device__ __forceinline__ void keys_to_cdf_wave(uint64_t* key, int n, int lane, const DevConfig* __restrict__ cfg) {
    const WfEntry wf = cfg->wf[0];
    const double* __restrict__ prm = cfg->wf_params + wf.offset;
    const double winv = cfg->wf_inv[0];
    for (int i = lane; i < n; i += 64) key[i] = d2u(cdf_lean(wf.kind, prm, wf.n_params, winv, u2d(key[i])) + 0.0);
    __syncthreads();
    bool inv = false;
    for (int i = lane; i < n; i += 64) inv = inv || (i > 0 && key[i] < key[i - 1]);
    if (__ballot(inv)) {
        if (lane == 0) {
            uint64_t m = 0;
            for (int i = 0; i < n; ++i) { m = key[i] > m ? key[i] : m; key[i] = m; }
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
// K1 (thresholded): one wavefront builds the sorted environment of one unique anchor.
//   radius search   kd-tree crate within_radius semantics: keep p iff sum(diff^2) < thr^2   (:521)
//   tag filter      p is the anchor itself, or pair_accepted(anchor.tag, p.tag)             (:524-528)
//   distance        sqrt(sum(diff^2)), same summation order as utils.rs:1-8                 (:537)
//   sort            ascending distance                                                       (:541)
// ------------------------------------------------------------------------------------------------
#ifdef LCHD_SWEEP_STAMPS
__device__ unsigned long long g_env_stamps[8];
#define ESTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && (blockIdx.x & 127) == 0) atomicAdd(&g_env_stamps[i], t_ - estamp_last); estamp_last = t_; } while (0)
#else
#define ESTAMP(i) do { } while (0)
#endif
// VT: the category type of the LDS buffer and of the store (uint16_t: more than 255 categories, EnvStore::cat16; no O(n) bucket sort)
template <int NT, bool TAGLIST, class VT = uint8_t>  // TAGLIST: the tag rule is a pair list (binary searches); otherwise one comparison, no branch
#ifdef ENV_W8
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 64 ? 8 : 1, NT == 64 ? 8 : 8))) void k_env_cells(
#else
__global__ __launch_bounds__(NT) void k_env_cells(
#endif
const DevConfig* __restrict__ cfgp, EnvSides sides, double thr, int cap, DeviceStatus* st) {
    // both structures in one launch: workgroups [0, sides.s[0].max_envs) build side A, the rest side B; the side's block of
    // kernel arguments is read with a wave-uniform index (scalar loads from the kernarg segment, no per-field selects)
    const int side = (int64_t)blockIdx.x >= sides.s[0].max_envs ? 1 : 0;
    const EnvSide& S = sides.s[side];
    const GridView g = S.g;
    const AnchorRec* __restrict__ uniq = S.uniq;
    const EnvStore env = S.env;
    // dynamic LDS: cap * (8 + sizeof(VT)) bytes (u64 keys, then the categories)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* key = reinterpret_cast<uint64_t*>(smem);
    VT* val = reinterpret_cast<VT*>(smem + (size_t)cap * 8);
    constexpr bool NARROW = sizeof(VT) == 1;
    __shared__ int count_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t e = (int64_t)blockIdx.x - (side ? sides.s[0].max_envs : 0);
#ifdef LCHD_SWEEP_STAMPS
    unsigned long long estamp_last = __builtin_amdgcn_s_memtime();
#endif
    if (e >= (int64_t)st->n_unique[side]) return;
    const DevConfig cfg = *cfgp;
    const AnchorRec arec = uniq[e];
    const double ax = arec.x, ay = arec.y, az = arec.z;
    const int32_t atag = (int32_t)arec.tag;
    const uint32_t apos = arec.apos;  // the anchor's own record in cell order
    const int asid = arec.sid;
    const double thr2 = thr * thr;
    const bool accept_same = cfg.tag_accept_same != 0;
    auto tag_ok = [&](int32_t t_other) -> bool {  // tag_pairing_rule.rs:49-75
        if constexpr (TAGLIST) return tag_pair_accepted(cfg, atag, t_other);
        else return (atag == t_other) == accept_same;
    };
    const int cx = cell_coord(ax, g.min[0], g.inv[0], g.dim[0]);
    const int cy = cell_coord(ay, g.min[1], g.inv[1], g.dim[1]);
    const int cz = cell_coord(az, g.min[2], g.inv[2], g.dim[2]);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
    if (NT > 64) {
        if (tid == 0) count_s = 0;
        __syncthreads();
    }

    int count = 0;  // NT == 64: the wave's running count; NT > 64: unused (count_s is the shared cursor)
    if constexpr (NT == 64) {
        // One wavefront.  The (up to) nine (y,z) rows of neighbour cells are contiguous runs of the cell-ordered records; their
        // bounds are fetched first (one dependent-load latency), then the runs are walked as ONE concatenated candidate list,
        // 64 candidates per step (every step is a full wavefront, however short the individual runs are); the record loads
        // of U steps are issued together.
        // The row bounds are worked out by lanes 0..8 (one row each: vector address arithmetic and two vector loads), turned
        // into offsets of the concatenated list by a wave scan and handed to every lane through v_readlane.  Done row by row
        // in scalar code the same thing took ~300 scalar instructions per environment, and the scalar unit (one per CU,
        // shared by all resident waves) was what bounded this kernel.
        const int kk = lane < 9 ? lane : 8;
        const int kz = (kk * 11) >> 5, ky = kk - 3 * kz;  // kk / 3, kk % 3 for kk < 9
        const int zz = cz - 1 + kz, yy = cy - 1 + ky;
        // Neighbour cells that lie wholly outside the radius are skipped: with the anchor at fractional position f in its
        // cell, a neighbour row / cell is at least (f or 1 - f) * edge away along every axis in which it differs.  A sphere
        // of radius thr meets on average 17 of the 27 cells (edge = 1.1 thr), so a third of the candidates never get loaded.
        // The test carries a relative margin of 1e-6 on thr^2 (rounding of the cell assignment is ~1e-16).
        const double fx = (ax - g.min[0]) * g.inv[0] - (double)cx, fy = (ay - g.min[1]) * g.inv[1] - (double)cy,
                     fz = (az - g.min[2]) * g.inv[2] - (double)cz;
        const double gy = fmax((ky == 0 ? fy : (ky == 2 ? 1.0 - fy : 0.0)) * g.cell[1], 0.0);
        const double gz = fmax((kz == 0 ? fz : (kz == 2 ? 1.0 - fz : 0.0)) * g.cell[2], 0.0);
        const double gxl = fmax(fx * g.cell[0], 0.0), gxh = fmax((1.0 - fx) * g.cell[0], 0.0);
        const double r2 = gy * gy + gz * gz, thr2m = thr2 * (1.0 + 1e-6);
        const int xl = (r2 + gxl * gxl < thr2m) ? x0 : cx, xh = (r2 + gxh * gxh < thr2m) ? x1 : cx;  // this row's x range
        const bool in = lane < 9 && zz >= 0 && zz < g.dim[2] && yy >= 0 && yy < g.dim[1] && r2 < thr2m;
        const int row = in ? (int)((((int64_t)asid * g.dim[2] + zz) * g.dim[1] + yy) * g.dim[0]) : 0;
        const int b_ = (int)g.cell_start[row + xl], e_ = (int)g.cell_start[row + xh + 1];
        const uint32_t len = in ? (uint32_t)(e_ - b_) : 0u;
        const uint32_t incl = wave_incl_scan_u32(len);
        const int roff_v = (int)(incl - len), dl_v = b_ - roff_v;
        // (each value passes through an empty asm: a select between two readlanes of one register is otherwise folded into
        // ONE readlane with a per-lane lane index, which the backend can only implement through a table in scratch memory)
#define LCHD_ROW(k)                                                                               \
    int dl##k = __builtin_amdgcn_readlane(dl_v, k), ro##k = __builtin_amdgcn_readlane(roff_v, k); \
    asm("" : "+s"(dl##k), "+s"(ro##k));
        LCHD_ROW(0) LCHD_ROW(1) LCHD_ROW(2) LCHD_ROW(3) LCHD_ROW(4) LCHD_ROW(5) LCHD_ROW(6) LCHD_ROW(7) LCHD_ROW(8)
#undef LCHD_ROW
        (void)ro0;
        const int total = __builtin_amdgcn_readlane((int)incl, 8);
        ESTAMP(0);
        const double2* __restrict__ rec2 = reinterpret_cast<const double2*>(g.rec);
        constexpr int U = LCHD_ENV_FLAT;
        for (int c0 = 0; c0 < total; c0 += 64 * U) {
            int idx[U];
            double2 R0[U], R1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = min(c0 + 64 * u + lane, total - 1);  // lanes past the end re-read the last candidate (masked below)
                int d = dl0;  // candidate t of the concatenated list -> record index t + dl[row of t]
                d = (t >= ro1) ? dl1 : d;
                d = (t >= ro2) ? dl2 : d;
                d = (t >= ro3) ? dl3 : d;
                d = (t >= ro4) ? dl4 : d;
                d = (t >= ro5) ? dl5 : d;
                d = (t >= ro6) ? dl6 : d;
                d = (t >= ro7) ? dl7 : d;
                d = (t >= ro8) ? dl8 : d;
                idx[u] = t + d;
                R0[u] = rec2[2 * (uint64_t)(uint32_t)idx[u]];  // (record indices are non-negative: zero extension is cheaper)
                R1[u] = rec2[2 * (uint64_t)(uint32_t)idx[u] + 1];
            }
            __builtin_amdgcn_sched_barrier(0);  // all 2U loads are issued before the first distance is computed
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c0 + 64 * u < total) {  // wave-uniform
                    const bool v = c0 + 64 * u + lane < total;
                    const double dx = R0[u].x - ax, dy = R0[u].y - ay, dz = R1[u].x - az;
                    double d2 = dx * dx;   // TU is built with -ffp-contract=off: same roundings as the
                    d2 = d2 + dy * dy;     // reference's `distance += diff * diff`
                    d2 = d2 + dz * dz;
                    const uint64_t tc = d2u(R1[u].y);  // tag | cat << 32
                    bool ok = false;
                    if constexpr (TAGLIST) {
                        if (v && d2 < thr2) ok = ((uint32_t)idx[u] == apos) || tag_ok((int32_t)(uint32_t)tc);
                    } else {  // four compares and scalar mask logic, no branch
                        ok = (v & (d2 < thr2)) & (((uint32_t)idx[u] == apos) | tag_ok((int32_t)(uint32_t)tc));
                    }
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
                    if (ok) {
                        const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (pos < cap) {
                            key[pos] = d2u(d2);  // the square root is taken after compaction (a sixth of the candidates survive)
                            val[pos] = (VT)(tc >> 32);
                        }
                    }
                    count += __popcll(m);
                }
            }
        }
    } else {
    for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.dim[2] - 1); ++zz) {
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.dim[1] - 1); ++yy) {
            // the (up to) three x-neighbour cells of one (y,z) row are contiguous in the cell-ordered arrays
            const int row = (int)((((int64_t)asid * g.dim[2] + zz) * g.dim[1] + yy) * g.dim[0]);  // neighbours of the anchor's own structure only
            const int beg = (int)g.cell_start[row + x0], end = (int)g.cell_start[row + x1 + 1];
            for (int base = beg + wave * 64; base < end; base += NT) {
                const int idx = base + lane;
                bool ok = false;
                double d2 = 0.0;
                uint32_t ccat = 0;
                if (idx < end) {
                    const CellRec r = g.rec[idx];
                    const double dx = r.x - ax, dy = r.y - ay, dz = r.z - az;
                    d2 = dx * dx;
                    d2 = d2 + dy * dy;
                    d2 = d2 + dz * dz;
                    ccat = r.cat;
                    if (d2 < thr2) ok = ((uint32_t)idx == apos) || tag_ok((int32_t)r.tag);
                }
                const unsigned long long m = __ballot(ok);
                int wbase = 0;
                // several waves append concurrently: reserve a slice of the list per wave-iteration
                if (lane == 0 && m) wbase = atomicAdd(&count_s, __popcll(m));
                wbase = __shfl(wbase, 0);
                if (ok) {
                    const int pos = wbase + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < cap) {
                        key[pos] = d2u(sqrt(d2));
                        val[pos] = (VT)ccat;
                    }
                }
            }
        }
    }
    }
    if (NT > 64) {
        __syncthreads();
        count = count_s;
    }
    ESTAMP(1);
    if (count > cap) {
        // The environment does not fit its slot: flagged, its size reported, its slot index appended to the side's overflow list.  The
        // slot receives the anchor alone -- a valid one-point environment, so the sweeps of this pass run cleanly over the pairs of
        // this anchor; the host scores those pairs again with larger slots (lchd_ctx_finish).
        if (tid == 0) {
            atomicOr(&st->flags, ST_ENV_OVERFLOW);
            atomicMax(&st->max_env, (uint32_t)count);
            const uint32_t k = atomicAdd(&st->n_overflow[side], 1u);
            if (S.ovf_list) S.ovf_list[k] = (uint32_t)e;
            const uint32_t acat = g.rec[arec.apos].cat;
            env.len[e] = 1;
            env.key[e * env.stride] = 0ull;
            reinterpret_cast<VT*>(env.cat)[e * env.stride] = (int)acat < cfg.n_categories ? (VT)acat : (VT)0;
        }
        return;
    }
    if (count == 0) {
        if (tid == 0) { atomicOr(&st->flags, ST_EMPTY_ENV); env.len[e] = 0; }
        return;
    }
    bool sorted = false;
    if constexpr (NT == 64 && NARROW) {
        // Typical environments (<= 512 points) are sorted in O(n) by one wavefront: inside a sphere the number of points
        // grows like d^3, so bucket = floor(256 * (d / thr)^3) spreads them almost evenly over 256 buckets (any
        // monotone map is correct; it only has to be balanced to be fast).  LDS histogram with returned slots -> wave
        // scan of the bucket sizes -> scatter from registers, grouped by bucket -> every element ranks itself among the
        // members of its own bucket on the exact f64 key (lane-parallel, a few LDS reads each) -> final placement.
        // Clustered inputs (a bucket with > 16 points) use the bitonic network.
        constexpr int B = 256, EPT = 8;
        if (count <= 64 * EPT) {
            uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (size_t)cap * 9 + ((16 - (((size_t)cap * 9) & 15)) & 15));  // [B + 1]
            for (int b = lane; b <= B; b += 64) hist[b] = 0u;
            __syncthreads();
            const double qs = (double)B / (thr2 * thr);  // B / thr^3
            uint64_t rk[EPT];
            uint32_t rp[EPT];  // category | bucket << 8 | slot inside the bucket << 16 (one register per element)
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int i = lane + 64 * q;
                rk[q] = 0; rp[q] = 0;
                if (i < count) {
                    const double d2 = u2d(key[i]);
                    const double d = sqrt(d2);  // utils.rs:1-8
                    rk[q] = d2u(d);
                    const double t = d2 * d * qs;
                    const int b = t < (double)B ? (int)t : B - 1;
                    rp[q] = (uint32_t)val[i] | ((uint32_t)b << 8) | (atomicAdd(&hist[b], 1u) << 16);
                }
            }
            __syncthreads();
            // exclusive scan of the 256 bucket sizes: lane l owns buckets 4l .. 4l+3
            uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            const uint32_t incl = wave_incl_scan_u32(mine);
            const uint32_t seg_lo = incl - mine;
            const unsigned long long too_big = __ballot(max(max(h0, h1), max(h2, h3)) > 16u);
            __syncthreads();
            hist[4 * lane] = seg_lo;
            hist[4 * lane + 1] = seg_lo + h0;
            hist[4 * lane + 2] = seg_lo + h0 + h1;
            hist[4 * lane + 3] = seg_lo + h0 + h1 + h2;
            if (lane == 63) hist[B] = incl;  // = count
            __syncthreads();
            if (!too_big) {
                // group by bucket (arrival order inside a bucket) ...
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = lane + 64 * q;
                    if (i < count) {
                        const uint32_t pos = hist[(rp[q] >> 8) & 0xFFu] + (rp[q] >> 16);
                        key[pos] = rk[q];
                        rp[q] = (rp[q] & 0xFFFFu) | (pos << 16);
                    }
                }
                __syncthreads();
                // ... then every element ranks itself among the (one to a few) members of its bucket on the exact f64 key
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = lane + 64 * q;
                    if (i < count) {
                        const uint32_t b = (rp[q] >> 8) & 0xFFu, pos = rp[q] >> 16;
                        const uint32_t s0 = hist[b], s1 = hist[b + 1];
                        uint32_t rank = s0;
                        for (uint32_t j = s0; j < s1; ++j) {
                            const uint64_t kj = key[j];
                            rank += (kj < rk[q]) | ((kj == rk[q]) & (j < pos));
                        }
                        rp[q] = (rp[q] & 0xFFFFu) | (rank << 16);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = lane + 64 * q;
                    if (i < count) {
                        key[rp[q] >> 16] = rk[q];
                        val[rp[q] >> 16] = (uint8_t)rp[q];
                    }
                }
                __syncthreads();
                sorted = true;
            }
        }
    }
    if (!sorted) {
        if constexpr (NT == 64) {  // the keys still hold d^2
            for (int i = tid; i < count; i += NT) key[i] = d2u(sqrt(u2d(key[i])));
            __syncthreads();
        }
        const int n2 = next_pow2(count);
        for (int i = count + tid; i < n2; i += NT) { key[i] = kPadKey; val[i] = (VT)0; }
        __syncthreads();
        bitonic_sort_lds<NT, VT>(key, val, n2, tid);
    }
    ESTAMP(2);
    uint64_t* ok_ = env.key + e * env.stride;
    VT* oc_ = reinterpret_cast<VT*>(env.cat) + e * env.stride;
    // categories outside the map are reported HERE (pmf.rs:38-42 raises for a point of a used environment, which is exactly
    // what gets written below) and stored as 0: the sweep kernels do not test categories again
    bool bad = false;
    bool written = false;
    if constexpr (NT == 64) {
        if (env.cdf_keys) {
            // One pass: sorted distance -> F(distance) -> global memory, the monotonicity of the converted keys checked on the
            // way (F is monotone; its floating-point evaluation may produce a last-bit inversion between neighbours, which the
            // running maximum of the separate path below repairs -- rare enough to pay a second pass then).  The separate
            // conversion pass over LDS, its barrier and the second read of the keys were a tenth of this kernel.
            const WfEntry wf = cfg.wf[0];
            const double* __restrict__ prm = cfg.wf_params + wf.offset;
            const double winv = cfg.wf_inv[0];
            bool inv = false;
            double carry = 0.0;  // F of the previous round's last key (F >= 0)
            for (int i0 = 0; i0 < count; i0 += 64) {
                const int i = i0 + lane;
                const bool act = i < count;
                const double f = act ? cdf_lean(wf.kind, prm, wf.n_params, winv, u2d(key[act ? i : 0])) + 0.0 : INFINITY;
                double prev = wave_shr1_f64(f);
                if (lane == 0) prev = carry;
                inv |= act && f < prev;
                carry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(f), 63), __builtin_amdgcn_readlane(__double2loint(f), 63));
                if (act) {
                    const VT v = val[i];
                    bad |= (int)v >= cfg.n_categories;
                    ok_[i] = d2u(f);
                    oc_[i] = (int)v < cfg.n_categories ? v : (VT)0;
                }
            }
            written = !__ballot(inv);
        }
    }
    if (!written) {
        if (env.cdf_keys) {
            if constexpr (NT == 64) keys_to_cdf_wave(key, count, lane, cfgp);
            else keys_to_cdf_lds<NT>(key, count, tid, cfgp);
        }
        for (int i = tid; i < count; i += NT) {
            const VT v = val[i];
            bad |= (int)v >= cfg.n_categories;
            ok_[i] = key[i];
            oc_[i] = (int)v < cfg.n_categories ? v : (VT)0;
        }
    }
    ESTAMP(3);
    if (__ballot(bad) && (tid & 63) == 0) atomicOr(&st->flags, ST_BAD_CATEGORY);
    if (tid == 0) env.len[e] = count;
    ESTAMP(4);
}

template <int NT>
static void launch_env_cells_nt(hipStream_t s, dim3 grid, size_t lds, bool tag_list, const DevConfig* cfg, const EnvSide& a, const EnvSide& b,
                                double thr, int cap, DeviceStatus* st) {
    EnvSides sides;
    sides.s[0] = a;
    sides.s[1] = b;
    if (a.env.cat16) {
        if (tag_list) k_env_cells<NT, true, uint16_t><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
        else k_env_cells<NT, false, uint16_t><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
        return;
    }
    if (tag_list) k_env_cells<NT, true><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
    else k_env_cells<NT, false><<<grid, NT, lds, s>>>(cfg, sides, thr, cap, st);
}

// Thresholded environments of more than 16384 points (a threshold that swallows most of a large structure): too many keys for
// LDS.  One 1024-thread workgroup per unique anchor walks the neighbour cells like k_env_cells and appends the survivors --
// distance and category -- UNSORTED to the environment's scratch row in global memory; k_env_rows then sorts each scratch row
// into the environment store in global memory, exactly as it does for given distance rows.
template <bool TAGLIST>
__global__ __launch_bounds__(1024) void k_env_collect(const DevConfig* __restrict__ cfgp, EnvSides sides, double thr, int cap, DeviceStatus* st) {
    const int side = (int64_t)blockIdx.x >= sides.s[0].max_envs ? 1 : 0;
    const EnvSide& S = sides.s[side];
    const GridView g = S.g;
    const int64_t e = (int64_t)blockIdx.x - (side ? sides.s[0].max_envs : 0);
    __shared__ int count_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (e >= (int64_t)st->n_unique[side]) return;
    const DevConfig cfg = *cfgp;
    const AnchorRec arec = S.uniq[e];
    const double ax = arec.x, ay = arec.y, az = arec.z, thr2 = thr * thr;
    const int32_t atag = (int32_t)arec.tag;
    const bool accept_same = cfg.tag_accept_same != 0;
    const int cx = cell_coord(ax, g.min[0], g.inv[0], g.dim[0]);
    const int cy = cell_coord(ay, g.min[1], g.inv[1], g.dim[1]);
    const int cz = cell_coord(az, g.min[2], g.inv[2], g.dim[2]);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
    double* __restrict__ rk = S.raw_key + e * (int64_t)cap;
    uint8_t* __restrict__ rc = S.raw_cat + e * (int64_t)cap;
    if (tid == 0) count_s = 0;
    __syncthreads();
    for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.dim[2] - 1); ++zz)
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.dim[1] - 1); ++yy) {
            const int row = (int)((((int64_t)arec.sid * g.dim[2] + zz) * g.dim[1] + yy) * g.dim[0]);
            const int beg = (int)g.cell_start[row + x0], end = (int)g.cell_start[row + x1 + 1];
            for (int base = beg + wave * 64; base < end; base += 1024) {
                const int idx = base + lane;
                bool ok = false;
                double d2 = 0.0;
                uint32_t ccat = 0;
                if (idx < end) {
                    const CellRec r = g.rec[idx];
                    const double dx = r.x - ax, dy = r.y - ay, dz = r.z - az;
                    d2 = dx * dx;  // utils.rs:1-8 order, uncontracted
                    d2 = d2 + dy * dy;
                    d2 = d2 + dz * dz;
                    ccat = r.cat;
                    if (d2 < thr2) {
                        if constexpr (TAGLIST) ok = ((uint32_t)idx == arec.apos) || tag_pair_accepted(cfg, atag, (int32_t)r.tag);
                        else ok = ((uint32_t)idx == arec.apos) || ((atag == (int32_t)r.tag) == accept_same);
                    }
                }
                const unsigned long long m = __ballot(ok);
                int wbase = 0;
                if (lane == 0 && m) wbase = atomicAdd(&count_s, __popcll(m));
                wbase = __shfl(wbase, 0);
                if (ok) {
                    const int pos = wbase + __popcll(m & ((1ull << lane) - 1ull));
                    if (pos < cap) { rk[pos] = sqrt(d2); rc[pos] = (uint8_t)ccat; }
                }
            }
        }
    __syncthreads();
    if (tid == 0) {
        const int count = count_s;
        if (count > cap) { atomicOr(&st->flags, ST_ENV_OVERFLOW); atomicMax(&st->max_env, (uint32_t)count); S.env.len[e] = 0; }
        else if (count == 0) { atomicOr(&st->flags, ST_EMPTY_ENV); S.env.len[e] = 0; }
        else S.env.len[e] = count;  // (k_env_rows sorts the row into the store and keeps this length)
    }
}

bool launch_env_cells(hipStream_t s, int cap, const DevConfig* cfg, bool tag_list, const EnvSide& a, const EnvSide& b, double thr,
                      DeviceStatus* st) {
    if (a.max_envs + b.max_envs <= 0) return true;
    const bool cat16 = a.env.cat16 != 0;
    if (cap > 16384 && cap <= (1 << 23) && !(cap & (cap - 1))) {  // collect unsorted, then the global-memory row sort (beyond 65536: swept by k_sweep_wide<.., BIG>)
        if (!a.raw_key || !b.raw_key || cat16) return false;
        EnvSides sides;
        sides.s[0] = a;
        sides.s[1] = b;
        const dim3 grid((unsigned)(a.max_envs + b.max_envs));
        if (tag_list) k_env_collect<true><<<grid, 1024, 0, s>>>(cfg, sides, thr, cap, st);
        else k_env_collect<false><<<grid, 1024, 0, s>>>(cfg, sides, thr, cap, st);
        for (int side = 0; side < 2; ++side) {
            const EnvSide& S = side ? b : a;
            if (S.max_envs <= 0) continue;
            RowExtras ex{S.raw_cat, S.env.len, &st->n_unique[side]};
            if (!launch_env_rows(s, cap, cfg, S.c, S.raw_key, cap, S.max_envs, cap, 0.0, S.env, st, ex)) return false;
        }
        return true;
    }
    if (cap < 64 || cap > 16384 || (cap & (cap - 1))) return false;
    if (cat16 && cap > 8192) return false;  // (10 bytes per point: 16384 points would need the CU's whole LDS)
    const dim3 grid((unsigned)(a.max_envs + b.max_envs));
    const size_t lds = (size_t)cap * (cat16 ? 10 : 9);
    if (cap <= 2048) {
        launch_env_cells_nt<64>(s, grid, lds + 16 + 257 * sizeof(uint32_t), tag_list, cfg, a, b, thr, cap, st);
    } else if (cap <= 4096) {
        launch_env_cells_nt<256>(s, grid, lds, tag_list, cfg, a, b, thr, cap, st);
    } else {
        launch_env_cells_nt<1024>(s, grid, lds, tag_list, cfg, a, b, thr, cap, st);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// K1 (dense): one workgroup sorts one full row -- from_coords (distances from anchor `row` to every atom,
// utils.rs:10-22 + :25-39) or from_dmxs (a caller-supplied distance-matrix row, src/locohd.rs:439-440).
// Dynamic LDS: n2 * 9 bytes.
// ------------------------------------------------------------------------------------------------
constexpr int kRowBucketsSmall = 2048;   // distance buckets of the dense-row sort (rows <= 16384 points)
constexpr int kRowBucketsMax = 8192;     // ... of k_env_rows2 when the row leaves room for them
constexpr int kRowBucketsBig = 16384;    // ... for rows of up to 65535 points (keys stay in global memory)
constexpr int kRowBucketsHuge = 32768;   // ... for longer rows
constexpr int kRowCoarse = 256;       // uniform bins of the row's empirical distance CDF
constexpr int kRowBucketLimit = 64;   // a fuller bucket sends the row to the bitonic network instead

template <int NT, bool GLOBALKV, class VT = uint8_t>  // VT uint16_t: more than 255 categories (EnvStore::cat16, CloudView::cat_hi)
__global__ __launch_bounds__(NT) void k_env_rows(const DevConfig* __restrict__ cfgp, CloudView c, const double* __restrict__ dmx,
                                                 int64_t ld, int64_t row_len, int n2, int n_buckets, double image_bound, EnvStore env,
                                                 DeviceStatus* st, RowExtras ex) {
    // ex (thresholded environments of more than 16384 points, collected unsorted by k_env_collect): the row's own categories
    // and length instead of the structure's, rows beyond the side's unique anchors are not there
    if (ex.n_unique && (uint32_t)blockIdx.x >= *ex.n_unique) return;
    // Sorting one row of n <= 16384 distances in O(n): an empirical CDF of the row on kRowCoarse uniform bins of
    // [0, max] of a monotone image of the distance (d^2 for coordinates) gives every point an interpolated rank; rank * kRowBuckets / n is its bucket, so buckets hold
    // ~n / kRowBuckets points whatever the shape of the cloud.  One LDS histogram + scan + scatter puts the points
    // into bucket order, then one thread finishes each bucket with an insertion sort on the exact f64 keys.  The map
    // distance -> bucket is monotone, which is all correctness needs; a pathological row (a bucket with more than
    // kRowBucketLimit points, e.g. thousands of identical distances) takes the bitonic network instead.
    // Distances are recomputed in every phase (3 L2-resident loads + a sqrt) instead of being kept in registers.
    // GLOBALKV (rows of 16 385 .. 65 535 points): the keys are sorted in place in the environment store (global memory,
    // L2-resident per row) and only the bucket histogram lives in LDS; otherwise keys and categories are in LDS too.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t r_ = blockIdx.x;
    uint64_t* key = GLOBALKV ? env.key + r_ * env.stride : reinterpret_cast<uint64_t*>(smem);
    VT* val = GLOBALKV ? reinterpret_cast<VT*>(env.cat) + r_ * env.stride : reinterpret_cast<VT*>(smem + (size_t)n2 * 8);
    constexpr size_t kPer = 8 + sizeof(VT);
    uint32_t* hist = GLOBALKV ? reinterpret_cast<uint32_t*>(smem)
                              : reinterpret_cast<uint32_t*>(smem + (size_t)n2 * kPer + ((16 - (((size_t)n2 * kPer) & 15)) & 15));  // [n_buckets + 1]
    const int kRowBuckets = n_buckets;
    __shared__ double red_max[16];
    __shared__ uint32_t red_cnt[16];
    __shared__ uint32_t scan_carry;
    __shared__ uint32_t coarse[kRowCoarse + 1], cum[kRowCoarse + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = blockIdx.x;
    const int n = ex.row_lens ? ex.row_lens[r] : (int)row_len;
    if (n <= 0) return;  // (an environment its collector flagged as empty or too large)
    const double* __restrict__ row = dmx ? dmx + r * ld : nullptr;
    const uint8_t* __restrict__ cats = ex.row_cat ? ex.row_cat + r * ld : c.cat;
    auto cat_at = [&](int i) -> VT {  // (two-byte ids: the structure's own planes; the collected rows of k_env_collect are one-byte only)
        if constexpr (sizeof(VT) == 2) return (VT)cat_of_atom(c, i);
        else return cats[i];
    };
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (!dmx) { ax = c.x[r]; ay = c.y[r]; az = c.z[r]; }
    // The bucketing phases work on a MONOTONE image of the distance -- the squared distance for coordinates (no square root
    // until the key is written), the distance itself for a given row -- and only the scatter takes the root of the survivors'
    // d^2; `dist_of` of the same image is what the reference computes (utils.rs:1-8).
    auto image_of = [&](int i, bool& bad) -> double {
        if (dmx) {
            double v = row[i];
            if (!(v >= 0.0)) { bad = true; v = 0.0; }  // negative or NaN
            return v + 0.0;                              // -0.0 -> +0.0
        }
        const double dx = ax - c.x[i], dy = ay - c.y[i], dz = az - c.z[i];
        double d2 = dx * dx;  // utils.rs:1-8 order, uncontracted
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        return d2;
    };
    auto dist_from_image = [&](double m) -> double { return dmx ? m : sqrt(m); };
    auto dist_of = [&](int i, bool& bad) -> double { return dist_from_image(image_of(i, bad)); };

    // 1. largest finite distance image -- or, for coordinates, the caller's bound (squared diagonal of the bounding box): any
    //    upper bound will do, the empirical CDF below adapts the buckets to wherever the points really are
    bool bad = false;
    double dmax = image_bound > 0.0 ? image_bound : 0.0;
    if (!(image_bound > 0.0))
        for (int i = tid; i < n; i += NT) {
            const double v = image_of(i, bad);
            if (v < 1.0e300 && v > dmax) dmax = v;
        }
    if (bad) atomicOr(&st->flags, ST_BAD_DISTANCE);
    for (int m = 32; m > 0; m >>= 1) dmax = fmax(dmax, shfl_xor_f64(dmax, m));
    if (lane == 0) red_max[wave] = dmax;
    for (int b = tid; b <= kRowBuckets; b += NT) hist[b] = 0u;
    for (int b = tid; b <= kRowCoarse; b += NT) coarse[b] = 0u;
    if (tid == 0) scan_carry = 0;
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) dmax = fmax(dmax, red_max[w]);
    // 2. empirical CDF on the coarse bins
    const double inv_w = dmax > 0.0 ? (double)kRowCoarse / dmax : 0.0;
    for (int i = tid; i < n; i += NT) {
        const double v = image_of(i, bad);
        if (v <= dmax) atomicAdd(&coarse[min((int)(v * inv_w), kRowCoarse - 1)], 1u);
    }
    __syncthreads();
    if (wave == 0) {  // cum[b] = points below bin b
        uint32_t carry = 0;
        for (int base = 0; base < kRowCoarse; base += 64) {
            const uint32_t v = coarse[base + lane];
            const uint32_t incl = wave_incl_scan_u32(v);
            cum[base + lane] = carry + incl - v;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) cum[kRowCoarse] = carry;
    }
    __syncthreads();
    const double rank_scale = n > 0 ? (double)kRowBuckets / (double)n : 0.0;
    auto bucket_of = [&](double v) -> int {
        if (!(v <= dmax)) return kRowBuckets - 1;  // +inf entries of a distance matrix
        const double t = v * inv_w;
        const int bin = min((int)t, kRowCoarse - 1);
        const double frac = fmin(t - (double)bin, 1.0);
        const double q = ((double)cum[bin] + frac * (double)coarse[bin]) * rank_scale;
        return q < (double)kRowBuckets ? (int)q : kRowBuckets - 1;
    };
    // 3. bucket histogram
    uint32_t biggest = 0;
    for (int i = tid; i < n; i += NT) biggest = max(biggest, atomicAdd(&hist[bucket_of(image_of(i, bad))], 1u) + 1u);
    for (int m = 32; m > 0; m >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, m));
    if (lane == 0) red_cnt[wave] = biggest;
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) biggest = max(biggest, red_cnt[w]);

    if (biggest > (uint32_t)kRowBucketLimit) {
        for (int i = tid; i < n2; i += NT) {
            key[i] = i < n ? d2u(dist_of(i, bad)) : kPadKey;
            val[i] = i < n ? cat_at(i) : (VT)0;
        }
        __syncthreads();
        bitonic_sort_lds<NT, VT>(key, val, n2, tid);
    } else {
        // 4. exclusive scan: hist[b] = first slot of bucket b
        for (int base = 0; base < kRowBuckets; base += NT) {
            const int b = base + tid;
            const uint32_t v = b < kRowBuckets ? hist[b] : 0u;
            const uint32_t incl = wave_incl_scan_u32(v);
            if (lane == 63) red_cnt[wave] = incl;
            __syncthreads();
            uint32_t wpre = 0;
            for (int w = 0; w < wave; ++w) wpre += red_cnt[w];
            const uint32_t carry = scan_carry;
            if (b < kRowBuckets) hist[b] = carry + wpre + incl - v;
            __syncthreads();
            if (tid == NT - 1) scan_carry = carry + wpre + incl;
            __syncthreads();
        }
        // 5. scatter; the bucket cursor advances in place, so afterwards hist[b] = END of bucket b
        for (int i = tid; i < n; i += NT) {
            const double m = image_of(i, bad);
            const uint32_t pos = atomicAdd(&hist[bucket_of(m)], 1u);
            key[pos] = d2u(dist_from_image(m));
            val[pos] = cat_at(i);
        }
        __syncthreads();
        // 6. finish every bucket with an insertion sort on the exact keys
        for (int b = tid; b < kRowBuckets; b += NT) {
            const int lo = b ? (int)hist[b - 1] : 0, hi = (int)hist[b];
            for (int i = lo + 1; i < hi; ++i) {
                const uint64_t k = key[i];
                const VT v = val[i];
                int j = i - 1;
                while (j >= lo && key[j] > k) { key[j + 1] = key[j]; val[j + 1] = val[j]; --j; }
                key[j + 1] = k;
                val[j + 1] = v;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        env.len[r] = n;
        if (n > 0 && key[0] != 0ull) atomicOr(&st->flags, ST_FIRST_NOT_ZERO);  // src/locohd.rs:74-77, on the distance
    }
    __syncthreads();
    if (env.cdf_keys) keys_to_cdf_lds<NT>(key, n, tid, cfgp);
    {   // categories outside the map: reported here, stored as 0 (see k_env_cells)
        const int C = cfgp->n_categories;
        bool bad_c = false;
        if constexpr (!GLOBALKV) {
            uint64_t* ok_ = env.key + r * env.stride;
            VT* oc_ = reinterpret_cast<VT*>(env.cat) + r * env.stride;
            for (int i = tid; i < n; i += NT) {
                const VT v = val[i];
                bad_c |= (int)v >= C;
                ok_[i] = key[i];
                oc_[i] = (int)v < C ? v : (VT)0;
            }
        } else {
            for (int i = tid; i < n; i += NT) {
                const VT v = val[i];
                if ((int)v >= C) { bad_c = true; val[i] = 0; }
            }
        }
        if (__ballot(bad_c) && (tid & 63) == 0) atomicOr(&st->flags, ST_BAD_CATEGORY);
    }
}

// ------------------------------------------------------------------------------------------------
// K1 (dense), rows of at most 16 384 points: the same bucket sort with every point's distance image computed ONCE and held
// in registers (EPT points per thread), one LDS atomic per histogram, a scatter without atomics (bucket start + the slot
// the histogram atomic returned), and the last step done by ALL threads: every point ranks itself among the handful of
// members of its bucket on the exact f64 key, then writes its (CDF-converted) key to its final place.  (k_env_rows
// recomputes the distances in three passes and finishes the buckets with one thread each -- 3.6 ms for the 2 x 10^4 rows
// of two 10^4-atom structures; this kernel builds both structures' rows in one launch.)
// Dynamic LDS: n2 * 9 bytes (keys, categories) + the bucket histogram.
// ------------------------------------------------------------------------------------------------
constexpr int kRowSegCap = 11776;  // rows of more than 16384 points: most points of one distance segment (keys in LDS)
constexpr int kRowLongEpt = 20;    // ... and the points per thread of such a row (<= 20480 points, 1024 threads)
constexpr size_t kRowSegLds = (size_t)kRowSegCap * 9 + (size_t)(kRowBucketsMax + 1) * 4;  // keys, categories, histogram
static_assert(((size_t)kRowSegCap * 9) % 16 == 0, "histogram alignment");
// EPT: points per thread of ONE sort -- the whole row, or one distance segment of a long row (NSEG = 2).
// Long rows (16385 .. 20480 points: more keys than the LDS holds) are sorted segment by segment: the coarse empirical CDF
// says which half of the buckets -- the nearer or the farther half of the row, ~n/2 points each -- a point falls into;
// for each segment the block compacts its points into the key array (ballot prefix inside a wave, wave totals through LDS:
// a deterministic order), every thread takes EPT of them back into registers, and from there the sort is the one of a
// 10^4-point row.  (A first version kept all 20 points of a thread in registers through both segments and tested
// "is it in this segment?" per point and phase: half the lanes idle in every instruction, 136 bytes of spill: 22.8 ms
// for the 4 x 10^4 rows of two 2 x 10^4-atom structures.)
template <int NT, int EPT, int NSEG>
__global__ __launch_bounds__(NT, (NT == 512 ? 2 : 4)) void k_env_rows2(const DevConfig* __restrict__ cfgp, RowSides sides, int n2, DeviceStatus* st) {
    static_assert(NSEG == 1 || NSEG == 2, "a point's segment is one bit");
    constexpr int n_seg = NSEG;
    constexpr int PEPT = NSEG > 1 ? kRowLongEpt : EPT;  // points per thread of the whole row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* key = reinterpret_cast<uint64_t*>(smem);
    uint8_t* val = smem + (NSEG > 1 ? (size_t)kRowSegCap : (size_t)n2) * 8;
    __shared__ double red_max[NT / 64];
    __shared__ uint32_t red_cnt[NT / 64], far_cnt[NT / 64];
    __shared__ uint32_t seg_tot[kRowBucketsMax / 64 + 1];
    __shared__ uint32_t coarse[kRowCoarse + 1], cum[kRowCoarse + 1];
    __shared__ uint32_t seg_n_s;
    __shared__ uint64_t carry_key_s;
    __shared__ double split_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int side = (int64_t)blockIdx.x >= sides.n_rows ? 1 : 0;
    const RowSide& S = sides.s[side];
    const int64_t r = (int64_t)blockIdx.x - (side ? sides.n_rows : 0);
    const CloudView c = S.c;
    const EnvStore env = S.env;
    const int n = S.row_lens ? S.row_lens[r] : (int)S.row_len;  // (ragged distance matrices: every row its own length)
    // Buckets: as many as fit (up to kRowBucketsMax, ~1 point per bucket: the ranking step reads a bucket's members once
    // per member).  The histogram lives in the part of the key array the row does not need -- the array is sized for the
    // bitonic fallback, a power of two --, or behind the categories when the row fills it.
    int NB = kRowBucketsSmall;
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (size_t)n2 * 9 + ((16 - (((size_t)n2 * 9) & 15)) & 15));  // [NB + 1]
    if (n_seg > 1) {
        NB = kRowBucketsMax;
        hist = reinterpret_cast<uint32_t*>(smem + (size_t)kRowSegCap * 9);
    } else {
        const int nk = (n + 63) & ~63;
        while (NB > 64 && NB >= 4 * nk) NB >>= 1;  // short rows: no more than ~2 buckets per point
        for (int cand = kRowBucketsMax; cand > kRowBucketsSmall; cand >>= 1)
            if (cand <= 2 * nk && (size_t)(n2 - nk) * 8 >= (size_t)(cand + 1) * 4) {
                NB = cand;
                hist = reinterpret_cast<uint32_t*>(key + nk);
                break;
            }
    }
    const double* __restrict__ row = S.dmx ? S.dmx + r * S.ld : nullptr;
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (!row) { ax = c.x[r]; ay = c.y[r]; az = c.z[r]; }
#ifdef LCHD_SWEEP_STAMPS
    unsigned long long estamp_last = __builtin_amdgcn_s_memtime();
#endif

    // 1. the distance image of this thread's points (d^2 for coordinates, utils.rs:1-8 order, uncontracted; the distance
    //    itself for a given row) and their categories (four to a register).  Point i = tid + q * NT: coalesced.
    //    (The "given row or coordinates?" test stays OUTSIDE the loops over a thread's points: inside, it was a branch per
    //    point -- wave-uniform, but the loads behind it were issued one point after the other.)
    bool bad = false;
    auto image_row = [&](int tid, int q) -> double {
        const int i = tid + q * NT;
        double v = row[i < n ? i : 0];
        if (i < n && !(v >= 0.0)) { bad = true; v = 0.0; }  // negative or NaN
        return v + 0.0;                                      // -0.0 -> +0.0
    };
    auto image_xyz = [&](int tid, int q) -> double {
        const int i = tid + q * NT;
        const int ii = i < n ? i : 0;
        const double dx = ax - c.x[ii], dy = ay - c.y[ii], dz = az - c.z[ii];
        double d2 = dx * dx;
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        return d2;
    };
    constexpr int IB = PEPT <= 10 ? PEPT : (PEPT % 10 == 0 ? 10 : 8);
    double m[EPT];                   // the images of the points being sorted (the row, or the current segment)
    double mp[NSEG > 1 ? PEPT : 1];  // long rows: the images of all the thread's points while they are dealt to the segments
    uint32_t ct4[(EPT + 3) / 4];
    uint32_t cp4[NSEG > 1 ? (PEPT + 3) / 4 : 1];  // ... and their categories
#pragma unroll
    for (int q = 0; q < (EPT + 3) / 4; ++q) ct4[q] = 0u;
#pragma unroll
    for (int q = 0; q < (NSEG > 1 ? (PEPT + 3) / 4 : 1); ++q) cp4[q] = 0u;
    if constexpr (NSEG == 1) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * NT;
            ct4[q >> 2] |= (uint32_t)c.cat[i < n ? i : 0] << ((q & 3) * 8);
        }
        if (row) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) m[q] = image_row(tid, q);
        } else {
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                m[q] = image_xyz(tid, q);
                if ((q + 1) % IB == 0) __builtin_amdgcn_sched_barrier(0);  // (at most 3 * IB coordinate loads in flight)
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < PEPT; ++q) {
            const int i = tid + q * NT;
            cp4[q >> 2] |= (uint32_t)c.cat[i < n ? i : 0] << ((q & 3) * 8);
        }
        if (row) {
#pragma unroll
            for (int q = 0; q < PEPT; ++q) mp[q] = image_row(tid, q);
        } else {
#pragma unroll
            for (int q = 0; q < PEPT; ++q) {
                mp[q] = image_xyz(tid, q);
                if ((q + 1) % IB == 0) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    auto row_img = [&](int q) -> double {  // q static
        if constexpr (NSEG > 1) return mp[q];
        else return m[q];
    };
    auto cat_of = [&](int q) -> uint8_t { return (uint8_t)(ct4[q >> 2] >> ((q & 3) * 8)); };  // q static
    if (__ballot(bad) && lane == 0) atomicOr(&st->flags, ST_BAD_DISTANCE);
    ESTAMP(0);
    // largest finite image -- or, for coordinates, the caller's bound (squared diagonal of the bounding box): any upper
    // bound will do, the empirical CDF below adapts the buckets to wherever the points really are
    double dmax = S.image_bound > 0.0 ? S.image_bound : 0.0;
    if (!(S.image_bound > 0.0)) {
#pragma unroll
        for (int q = 0; q < PEPT; ++q)
            if (tid + q * NT < n && row_img(q) < 1.0e300 && row_img(q) > dmax) dmax = row_img(q);
        for (int k = 32; k > 0; k >>= 1) dmax = fmax(dmax, shfl_xor_f64(dmax, k));
        if (lane == 0) red_max[wave] = dmax;
    }
    for (int b = tid; b <= kRowCoarse; b += NT) coarse[b] = 0u;
    if (tid == 0) carry_key_s = 0ull;
    __syncthreads();
    if (!(S.image_bound > 0.0))
        for (int w = 0; w < NT / 64; ++w) dmax = fmax(dmax, red_max[w]);
    // 2. empirical CDF of the row on kRowCoarse uniform bins of [0, dmax]
    const double inv_w = dmax > 0.0 ? (double)kRowCoarse / dmax : 0.0;
#pragma unroll
    for (int q = 0; q < PEPT; ++q)
        if (tid + q * NT < n && row_img(q) <= dmax) atomicAdd(&coarse[min((int)(row_img(q) * inv_w), kRowCoarse - 1)], 1u);
    __syncthreads();
    if (wave == 0) {  // cum[b] = points below bin b
        uint32_t carry = 0;
        for (int base = 0; base < kRowCoarse; base += 64) {
            const uint32_t v = coarse[base + lane];
            const uint32_t incl = wave_incl_scan_u32(v);
            cum[base + lane] = carry + incl - v;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) cum[kRowCoarse] = carry;
    }
    __syncthreads();
    ESTAMP(1);
    const DevConfig cfg = *cfgp;
    const WfEntry wf = cfg.wf[0];
    const double* __restrict__ prm = cfg.wf_params + wf.offset;
    const double winv = cfg.wf_inv[0];
    const int NBT = NB * n_seg;  // buckets of the whole row; distance segment sg owns buckets [sg * NB, (sg + 1) * NB)
    const double rank_scale = n > 0 ? (double)NBT / (double)n : 0.0;
    // interpolated rank of an image in the row -> one of NBT balanced buckets.  Single precision: any map that never
    // decreases with the image sorts correctly (rounding to float, the product with a positive constant, the truncation and
    // the interpolation inside a bin -- which never exceeds the next bin's start -- all are), it only has to balance the
    // buckets, and the double-precision conversions were a third of this phase's instructions.
    const float inv_wf = (float)inv_w, rank_scale_f = (float)rank_scale;
    auto bucket_of = [&](double v) -> int {
        int gb = NBT - 1;  // +inf entries of a distance matrix
        if (v <= dmax) {
            const float t = (float)v * inv_wf;
            const int bin = min((int)t, kRowCoarse - 1);
            const float frac = fminf(t - (float)bin, 1.0f);
            const float qq = ((float)cum[bin] + frac * (float)coarse[bin]) * rank_scale_f;
            gb = qq < (float)NBT ? (int)qq : NBT - 1;
        }
        return gb;
    };
    uint64_t* ok_ = env.key + r * env.stride;
    uint8_t* oc_ = env.cat + r * env.stride;
    uint32_t seg_total[2] = {0u, 0u};
    if constexpr (NSEG > 1) {
        // Long rows: deal the points to the two segments, ONCE and from the registers (every further pass over the row's
        // coordinates costs ~8 000 cycles of this CU's 64-byte-per-clock L1 path: 480 KB).  The nearer segment's points go
        // into the key array, the farther segment's into the row's own slot of the environment store (which its sorted
        // keys overwrite at the end); position = points of the lower waves + of this wave's earlier q + of the lower lanes:
        // a deterministic order.
        // Which segment?  One comparison with the image at which the empirical CDF reaches n / 2 (any threshold keeps the
        // two segments ordered; this one balances them).
        if (wave == 0) {
            const uint32_t half = (uint32_t)n / 2u;
            int bin = 0;  // the last bin that starts at or below the median rank
            for (int b = lane; b < kRowCoarse; b += 64) bin = cum[b] <= half ? b : bin;
            for (int k = 32; k > 0; k >>= 1) bin = max(bin, __shfl_xor(bin, k));
            if (lane == 0) {
                const double inside = coarse[bin] ? (double)(half - cum[bin]) / (double)coarse[bin] : 0.0;
                split_s = inv_w > 0.0 ? ((double)bin + fmin(inside, 1.0)) / inv_w : 0.0;
            }
        }
        __syncthreads();
        const double split = split_s;
        uint32_t seg_bits = 0u;  // bit q = the segment of point q
#pragma unroll
        for (int q = 0; q < PEPT; ++q)
            if (tid + q * NT < n && mp[q] >= split) seg_bits |= 1u << q;
        ESTAMP(0);  // (diagnostic builds: the segment bits are booked on the image phase, the dealing on the coarse-CDF phase)
        uint32_t wn0 = 0, wn1 = 0;
#pragma unroll
        for (int q = 0; q < PEPT; ++q) {
            const bool in = tid + q * NT < n, far = (seg_bits >> q) & 1u;
            wn0 += (uint32_t)__popcll(__ballot(in && !far));
            wn1 += (uint32_t)__popcll(__ballot(in && far));
        }
        if (lane == 0) { red_cnt[wave] = wn0; far_cnt[wave] = wn1; }
        __syncthreads();
        uint32_t at0 = 0, at1 = 0;
        for (int w = 0; w < NT / 64; ++w) {
            const uint32_t v0 = red_cnt[w], v1 = far_cnt[w];
            at0 += w < wave ? v0 : 0u;
            at1 += w < wave ? v1 : 0u;
            seg_total[0] += v0;
            seg_total[1] += v1;
        }
        const uint32_t seg_max = max(seg_total[0], seg_total[1]);
        if (seg_max > (uint32_t)kRowSegCap || seg_max > (uint32_t)(EPT * NT)) {
            // the empirical CDF balanced the segments badly (no in-LDS fallback for these rows: the host repeats the call
            // with k_env_rows)
            if (tid == 0) { atomicOr(&st->flags, ST_ROW_RETRY); env.len[r] = 0; }
            return;
        }
#pragma unroll
        for (int q = 0; q < PEPT; ++q) {
            const int i = tid + q * NT;
            const bool in = i < n, far = (seg_bits >> q) & 1u;
            const unsigned long long m0 = __ballot(in && !far), m1 = __ballot(in && far);
            const uint8_t cv = (uint8_t)(cp4[q >> 2] >> ((q & 3) * 8));
            if (in && !far) {
                const uint32_t pos = at0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
                key[pos] = d2u(mp[q]);
                val[pos] = cv;
            }
            if (in && far) {
                const uint32_t pos = seg_total[0] + at1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                ok_[pos] = d2u(mp[q]);
                oc_[pos] = cv;
            }
            at0 += (uint32_t)__popcll(m0);
            at1 += (uint32_t)__popcll(m1);
        }
        __syncthreads();
        ESTAMP(1);
    }
    bool bad_c = false;
    int seg_base = 0;
#pragma unroll 1
    for (int sg = 0; sg < NSEG; ++sg) {
        // (an opaque copy of the thread index: addresses derived from it -- 60 coordinate pointers -- are otherwise hoisted out of
        //  the segment loop and kept alive through it: 280 spilled registers)
        int tl = tid;
        if constexpr (NSEG > 1) asm volatile("" : "+v"(tl));
        int n_pts = n;  // points of this sort
        if constexpr (NSEG > 1) {
            // every thread takes EPT of the segment's points back into registers: from the key array, or from the row's slot
            // of the environment store
            const uint32_t total = seg_total[sg];
            n_pts = (int)total;
#pragma unroll
            for (int q = 0; q < (EPT + 3) / 4; ++q) ct4[q] = 0u;
            if (sg == 0) {
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = tl + q * NT, ii = i < n_pts ? i : 0;
                    m[q] = u2d(key[ii]);
                    ct4[q >> 2] |= (uint32_t)val[ii] << ((q & 3) * 8);
                }
            } else {  // (clamped, unconditional loads: all in flight together; .glc -- written by other waves of this block)
                const uint64_t* src_k = ok_ + seg_total[0];
                const uint8_t* src_c = oc_ + seg_total[0];
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int i = tl + q * NT, ii = i < n_pts ? i : 0;
                    m[q] = u2d(__hip_atomic_load(&src_k[ii], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    ct4[q >> 2] |= (uint32_t)__hip_atomic_load(&src_c[ii], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << ((q & 3) * 8);
                }
            }
            __syncthreads();  // (the sort below reuses the key array)
        }
        for (int b = tl; b <= NB; b += NT) hist[b] = 0u;
        __syncthreads();
        // 3. the point's bucket; the histogram atomic returns its slot inside the bucket
        uint32_t bs[EPT];  // bucket | slot << 13
        uint32_t biggest = 0;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            bs[q] = ~0u;
            if (tl + q * NT < n_pts) {
                const int b = min(max(bucket_of(m[q]) - sg * NB, 0), NB - 1);  // (in range by the choice of the segment)
                const uint32_t slot = atomicAdd(&hist[b], 1u);
                bs[q] = (uint32_t)b | (slot << 13);
                biggest = max(biggest, slot + 1u);
            }
        }
        for (int k = 32; k > 0; k >>= 1) biggest = max(biggest, (uint32_t)__shfl_xor((int)biggest, k));
        if (lane == 0) red_cnt[wave] = biggest;
        __syncthreads();
        for (int w = 0; w < NT / 64; ++w) biggest = max(biggest, red_cnt[w]);
        __syncthreads();
        ESTAMP(2);
        if (biggest > (uint32_t)kRowBucketLimit) {
            if (n_seg > 1) {  // (rows of more than 16384 points have no in-LDS fallback: the host repeats the call with k_env_rows)
                if (tl == 0) { atomicOr(&st->flags, ST_ROW_RETRY); env.len[r] = 0; }
                return;
            }
            // a pathological row (thousands of identical distances): the bitonic network on the exact keys
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int i = tl + q * NT;
                if (i < n) { key[i] = d2u(row ? m[q] : sqrt(m[q])); val[i] = cat_of(q); }
            }
            for (int i = n + tl; i < n2; i += NT) { key[i] = kPadKey; val[i] = 0; }
            __syncthreads();
            bitonic_sort_lds<NT>(key, val, n2, tl);
            if (tl == 0 && n > 0 && key[0] != 0ull) atomicOr(&st->flags, ST_FIRST_NOT_ZERO);  // src/locohd.rs:74-77
            __syncthreads();
            if (env.cdf_keys) keys_to_cdf_lds<NT>(key, n, tl, cfgp);
            if (tl == 0) seg_n_s = (uint32_t)n;
            __syncthreads();
        } else {
            // 4. exclusive scan in groups of 64 buckets (one wavefront scan each), then of the group totals, then one
            //    coalesced pass adds the group offsets: hist[b] = first slot of bucket b, hist[NB] = points of the segment
            const int n_grp = NB >> 6;
            for (int gq = wave; gq < n_grp; gq += NT / 64) {
                const uint32_t v = hist[gq * 64 + lane];
                const uint32_t incl = wave_incl_scan_u32(v);
                hist[gq * 64 + lane] = incl - v;
                if (lane == 63) seg_tot[gq] = incl;
            }
            __syncthreads();
            if (wave == 0) {
                uint32_t carry = 0;
                for (int base = 0; base < n_grp; base += 64) {
                    const uint32_t v = base + lane < n_grp ? seg_tot[base + lane] : 0u;
                    const uint32_t incl = wave_incl_scan_u32(v);
                    if (base + lane < n_grp) seg_tot[base + lane] = carry + incl - v;
                    carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                }
                if (lane == 0) seg_n_s = carry;
            }
            __syncthreads();
            for (int b = tl; b < NB; b += NT) hist[b] += seg_tot[b >> 6];
            if (tl == 0) hist[NB] = seg_n_s;
            __syncthreads();
            ESTAMP(3);
            // 5. scatter (no atomics): position = bucket start + slot; the exact distance replaces the image in the register
#pragma unroll
            for (int q = 0; q < EPT; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t b = bs[q] & 8191u, pos = hist[b] + (bs[q] >> 13);
                    if (!row) m[q] = sqrt(m[q]);  // utils.rs:1-8
                    key[pos] = d2u(m[q]);
                    bs[q] = b | (pos << 13);
                }
            __syncthreads();
            ESTAMP(4);
            // 6. every point ranks itself among the members of its bucket on the exact key (ties: by position); four members
            //    per step, their LDS reads in flight together (a bucket holds one or two points on average)
#pragma unroll
            for (int q = 0; q < EPT; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t b = bs[q] & 8191u, pos = bs[q] >> 13;
                    const uint32_t lo = hist[b], hi = hist[b + 1];
                    const uint64_t mine = d2u(m[q]);
                    uint32_t rank = lo;
                    for (uint32_t j = lo; j < hi; j += 4) {
                        uint64_t kj[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) kj[u] = key[min(j + u, hi - 1)];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            rank += (j + u < hi) & ((kj[u] < mine) | ((kj[u] == mine) & (j + u < pos)));
                    }
                    bs[q] = rank;
                }
            __syncthreads();
            ESTAMP(5);
            // 7. final placement, keys converted to F(distance) for single-weight-function configurations
            bool nz = false;
#pragma unroll
            for (int q = 0; q < EPT; ++q)
                if (bs[q] != ~0u) {
                    const uint32_t rank = bs[q];
                    nz |= (sg == 0 && rank == 0u && m[q] != 0.0);  // src/locohd.rs:74-77, on the distance
                    key[rank] = env.cdf_keys ? d2u(cdf_lean(wf.kind, prm, wf.n_params, winv, m[q]) + 0.0) : d2u(m[q]);
                    val[rank] = cat_of(q);
                }
            if (__ballot(nz) && lane == 0) atomicOr(&st->flags, ST_FIRST_NOT_ZERO);
            __syncthreads();
        }
        ESTAMP(6);
        {   // 8. write-out; categories outside the map: reported here, stored as 0 (see k_env_cells).  F is monotone, but its
            //    floating-point evaluation may produce a last-bit inversion between neighbours: looked for on the way out
            //    (the keys are being read anyway) and, in the rare case, repaired by a running maximum and written again.
            const int seg_n = (int)seg_n_s, C = cfg.n_categories;
            bool inv = false;
            for (int i = tl; i < seg_n; i += NT) {
                const uint8_t v = val[i];
                const uint64_t k = key[i];
                bad_c |= (int)v >= C;
                if (env.cdf_keys) inv |= k < (i ? key[i - 1] : carry_key_s);
                ok_[seg_base + i] = k;
                oc_[seg_base + i] = (int)v < C ? v : (uint8_t)0;
            }
            if (__syncthreads_or(inv ? 1 : 0)) {
                if (tl == 0) {
                    uint64_t mx = carry_key_s;
                    for (int i = 0; i < seg_n; ++i) { mx = key[i] > mx ? key[i] : mx; key[i] = mx; }
                }
                __syncthreads();
                for (int i = tl; i < seg_n; i += NT) ok_[seg_base + i] = key[i];
                __syncthreads();
            }
            if (tl == 0 && seg_n > 0) carry_key_s = key[seg_n - 1];
            seg_base += seg_n;
            __syncthreads();
        }
        ESTAMP(7);
    }
    if (tid == 0) env.len[r] = n;
    if (__ballot(bad_c) && lane == 0) atomicOr(&st->flags, ST_BAD_CATEGORY);
}

bool launch_env_rows2(hipStream_t s, const DevConfig* cfg, const RowSide& a, const RowSide& b, int64_t n_rows, DeviceStatus* st) {
    if (n_rows <= 0) return true;
    const int64_t longest = std::max(a.row_len, b.row_len);
    if (longest > 20480 || a.row_len < 1 || b.row_len < 1) return false;
    int n2 = 64;
    while (n2 < longest && n2 < 16384) n2 <<= 1;
    // rows of 16385 .. 20480 points: sorted in distance segments of ~10^4 points each (the segment's keys in LDS)
    const int n_seg = longest > 16384 ? 2 : 1;
    if (n_seg > 1 && (std::min(a.row_len, b.row_len) <= 16384)) return false;  // (one launch, one segment count: both sides must be long)
    if (n_seg > 1 && (a.row_lens || b.row_lens)) return false;  // (ragged rows may be short: same reason)
    RowSides sides;
    sides.s[0] = a; sides.s[1] = b;
    if (a.dmx) sides.s[0].image_bound = 0.0;  // given rows: the kernel finds the largest finite entry itself
    if (b.dmx) sides.s[1].image_bound = 0.0;
    sides.n_rows = n_rows;
    const dim3 grid((unsigned)(2 * n_rows));
    const size_t lds = (size_t)n2 * 9 + 16 + (size_t)(kRowBucketsSmall + 1) * sizeof(uint32_t);
    if (n_seg > 1) k_env_rows2<1024, 12, 2><<<grid, 1024, kRowSegLds, s>>>(cfg, sides, n2, st);
    else if (n2 <= 1024) k_env_rows2<64, 16, 1><<<grid, 64, lds, s>>>(cfg, sides, n2, st);
    else if (n2 <= 4096) k_env_rows2<256, 16, 1><<<grid, 256, lds, s>>>(cfg, sides, n2, st);
    else if (n2 <= 8192) k_env_rows2<1024, 8, 1><<<grid, 1024, lds, s>>>(cfg, sides, n2, st);
#ifdef LCHD_ROWS_NT512
    else if (longest <= 10240) k_env_rows2<512, 20, 1><<<grid, 512, lds, s>>>(cfg, sides, n2, st);
#endif
    else if (longest <= 10240) k_env_rows2<1024, 10, 1><<<grid, 1024, lds, s>>>(cfg, sides, n2, st);
    else k_env_rows2<1024, 16, 1><<<grid, 1024, lds, s>>>(cfg, sides, n2, st);
    return true;
}

bool launch_env_rows(hipStream_t s, int cap, const DevConfig* cfg, const CloudView& c, const double* dmx, int64_t ld,
                     int64_t n_rows, int64_t row_len, double image_bound, EnvStore env, DeviceStatus* st, const RowExtras& ex) {
    if (dmx) image_bound = 0.0;  // given rows: the kernel finds the largest finite entry itself
    if (n_rows <= 0) return true;
    if (row_len > cap || cap > (1 << 23)) return false;
    const dim3 grid((unsigned)n_rows);
    if (cap > 65536) {  // rows of more than 65 535 points (swept by k_sweep_wide<.., BIG>): keys in the store, 32768 buckets (128 KB of LDS)
        if (env.cat16) return false;
        const size_t lds = (size_t)(kRowBucketsHuge + 1) * sizeof(uint32_t) + 16;
        k_env_rows<1024, true><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsHuge, image_bound, env, st, ex);
        return true;
    }
    if (env.cat16) {  // more than 255 categories: two bytes per point (rows of up to 8192 points in LDS, longer ones in the store)
        if (ex.row_cat) return false;
        if (cap > 8192) {
            const size_t lds = (size_t)(kRowBucketsBig + 1) * sizeof(uint32_t) + 16;
            k_env_rows<1024, true, uint16_t><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsBig, image_bound, env, st, ex);
        } else {
            const size_t lds = (size_t)cap * 10 + 16 + (size_t)(kRowBucketsSmall + 1) * sizeof(uint32_t);
            if (cap <= 1024) k_env_rows<64, false, uint16_t><<<grid, 64, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
            else k_env_rows<1024, false, uint16_t><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
        }
        return true;
    }
    if (cap > 16384) {  // keys in global memory, 64 KB histogram in LDS
        const size_t lds = (size_t)(kRowBucketsBig + 1) * sizeof(uint32_t) + 16;
        k_env_rows<1024, true><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsBig, image_bound, env, st, ex);
        return true;
    }
    const size_t lds = (size_t)cap * 9 + 16 + (size_t)(kRowBucketsSmall + 1) * sizeof(uint32_t);
    if (cap <= 1024) {
        k_env_rows<64, false><<<grid, 64, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
    } else if (cap <= 4096) {
        k_env_rows<256, false><<<grid, 256, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
    } else {
        k_env_rows<1024, false><<<grid, 1024, lds, s>>>(cfg, c, dmx, ld, row_len, cap, kRowBucketsSmall, image_bound, env, st, ex);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// K2: the sweep.  One wavefront per anchor pair, four pairs per 256-thread workgroup.
//
// S = sum_k [F(t_{k+1}) - F(t_k)] * H(state after k events), t_0 = 0, t_{M+1} = inf, where the events are
// the merged non-anchor points of both environments (the reference's two-pointer
// loop src/locohd.rs:97-223 evaluates exactly this sum; cross-list ties collapse because a zero-width
// interval contributes exactly 0).
//
// Events are processed in tiles of 384: lane l owns ceil(T/64) <= 6 consecutive merged events of the tile, found with a
// merge-path binary search in LDS.  A packed (16-bit fields) wavefront prefix scan of the per-lane
// category histograms gives every lane the exact integer category counts at its first event; it then
// walks its events sequentially with the per-category state in registers.
//
// MODE_H2U / MODE_H2W: Hellinger distance with exponent 2 (the default, src/locohd.rs:365-370), unit /
//   arbitrary category weights.  The per-lane state is just the packed integer category counts plus the
//   running Bhattacharyya numerator D = sum_c sqrt(a_c b_c); an event touches one category, so D is updated
//   in O(1) from an LDS table of sqrt(k) and H^2 = 1 - D / sqrt(N_a N_b).  Where that cancellation form would
//   lose accuracy (H^2 < kExactH2Below = 1e-6) the literal sum_c (sqrt(a_c/N_a) - sqrt(b_c/N_b))^2 / 2 is evaluated instead,
//   which also gives exactly 0 for identical environments.
// MODE_GEN: every other StatisticalDistance (statistical_distances.rs:4-78): weighted counts in registers,
//   normalised like pmf.rs:65-83, distance through one out-of-line call.
// ------------------------------------------------------------------------------------------------
enum { MODE_H2U = 0, MODE_H2W = 1, MODE_GEN = 2 };
// where F(t) comes from: the environment keys already are F values / inline CDFs only / any CDF
enum { F_KEY = 0, F_FAST = 1, F_ANY = 2 };
#ifndef LCHD_PASS1_FUSED
#define LCHD_PASS1_FUSED 1  // k_sweep: the chunk histogram is one fixed-trip loop over the lane's points
#endif
#ifndef LCHD_LDS_COUNTS
#define LCHD_LDS_COUNTS 1   // k_sweep (Hellinger-2, LDS tables, > 12 category slots): per-lane category counts live in LDS during the event loop
#endif
#ifndef LCHD_HEADS_REREAD
#define LCHD_HEADS_REREAD 1   // k_sweep: both list heads are re-read from LDS after every event
#endif
#ifndef LCHD_CAT_HEADS
#define LCHD_CAT_HEADS 1      // k_sweep / k_sweep_duo: the categories of both list heads are read together with their keys
#endif
#ifndef LCHD_BRANCHFREE_HEADS
#define LCHD_BRANCHFREE_HEADS 1
#endif
#ifndef LCHD_SWEEP_WAVES
#define LCHD_SWEEP_WAVES 4
#endif
#ifndef LCHD_BIG_SQRT_COMPUTE
#define LCHD_BIG_SQRT_COMPUTE 1
#endif
#ifndef LCHD_SWEEP_W3MAX
#define LCHD_SWEEP_W3MAX 16   // largest category-slot count that is compiled for 3 waves per SIMD (above: 2)
#endif
#ifndef LCHD_SWEEP_MINW
#define LCHD_SWEEP_MINW 2
#endif
#ifndef LCHD_GEN_W3MAX
#define LCHD_GEN_W3MAX 0   // generic-distance sweeps (MODE_GEN) with at most this many category slots are compiled for 3 waves/SIMD
#endif
#ifndef LCHD_EPL_WGEN
#define LCHD_EPL_WGEN 7  // ... of the sweeps with category weights and of the generic distances, CDF-keyed environments (measured on C2a: weights 2.86 -> 2.54 ms, KS 4.62 -> 4.29 ms; the plain 16-bit Hellinger sweep and the sweeps that evaluate the CDF themselves are faster with 6: their LDS tables + tiles of 448 leave 3 workgroups per CU)
#endif
#ifndef LCHD_EPL_C8S
#define LCHD_EPL_C8S 8   // ... of the 8-bit-count sweep with at most 16 category slots: see LCHD_EPL_C8 (C2a: 343 events per pair on average; tiles of 384: 1.77 ms, 448: 1.61 ms, 512 with whole-list staging: 1.585 ms)
#endif
#ifndef LCHD_EPL_C8
#define LCHD_EPL_C8 8    // ... of the 8-bit-count sweep: tiles of 512 -- two environments of <= 255 points never merge to more, so every pair is ONE tile (a list is staged whole: 256 entries; C5: 448-event tiles + tile-sized staging 3.08 ms, whole-list staging 2.88 ms, 512-event tiles 2.80 ms)
#endif
#ifndef LCHD_C8_WAVES
#define LCHD_C8_WAVES 3  // waves per SIMD the 8-bit-count sweep with more than 16 category slots is compiled for
#endif
#ifndef LCHD_EPL_DENSE
#define LCHD_EPL_DENSE 9   // ... of the sweeps without LDS tables (environments beyond 512 points: dense rows, thousands of events per pair)
#endif
#ifndef LCHD_EPL_BIG
#define LCHD_EPL_BIG 8   // merged events per lane per tile of the many-slot Hellinger-2 sweep (k_sweep<20..32>): tiles of 512
#endif
constexpr int kDuoTileFwd = kDuoTile;  // (lchd_team_tile.h)
// The small rule in force in this pass, or -1 (the plain sweep takes every pair).  With a hint the host launched exactly the
// kernels that have to run (forced); without one every candidate kernel is launched and all of them decide here, from the
// counts of k_pair_meta: the first-choice rule if its pairs are the majority, else the second-choice rule if ITS pairs are --
// the same function of the pair list the host evaluates for the next pass's hint.
__device__ __forceinline__ int rule_in_force(const SweepArgs& args) {
    if (args.forced) return args.small_rule;
    const unsigned long long P = (unsigned long long)args.n_pairs;
    if (2 * args.st->n_small >= P) return args.small_rule;
    if (args.second_rule && 2 * args.st->n_c8 >= P) return args.second_rule;
    return -1;
}
#ifndef LCHD_INLINE_META_PAIRS
#define LCHD_INLINE_META_PAIRS 4096
#endif
constexpr int64_t kInlineMetaPairs = LCHD_INLINE_META_PAIRS;   // calls of at most this many pairs: the sweep works out the pair records itself (one launch)
constexpr int kSqrtTab = 512;  // LDSTAB kernels: environments of at most 512 points, sqrt tables entirely in LDS
constexpr int kSweepWaves = LCHD_SWEEP_WAVES;  // anchor pairs (wavefronts) per workgroup

// A sweep kernel reports a (rare) condition: plain store of 1 into the condition's word of the host-mapped mirror (every
// writer stores the same value; no atomics on host memory, no device-to-host copy afterwards).
__device__ __forceinline__ void sweep_report(HostStatus* h, uint32_t bit) { h->sweep_flags[__builtin_ctz(bit)] = 1u; }

// "Which workgroup finishes last, and what did all of them add up to?" -- without a fence.  An agent-scope release fence on
// this part writes the XCD's whole L2 back (the L2s of the eight XCDs are not coherent with each other), and a kernel that
// has just written 16 MB of pair records pays that per workgroup: 3 900 fences turned an 18 us kernel into a 137 us one.
// Device-scope atomics are performed at the memory side and are coherent by themselves, so everything the workgroups
// hand over travels IN atomics: up to 64 accumulators / counters on separate cache lines (thousands of atomics on one
// word would cost ~11 ns each), a workgroup's counter increment carries a data dependency on the values its accumulator
// atomics RETURNED (so they have been performed), and the workgroup that completes its counter bumps the top-level one.
// Called by ONE thread per workgroup; returns true in exactly one workgroup, which then collects the accumulators with
// atomic exchanges (resetting them).  Everything is left at zero.
constexpr int kDoneStride = 32;  // u32 per slot: 128 bytes apart
static_assert(kPrepScanAtoms == 1 << 18, "k_prep_scatter: chunk of atom i = i >> 18");
static_assert(sizeof(DoneState) == (65 + 64 + 64) * kDoneStride * 4, "DoneState layout (lchd_device.h)");
__device__ __forceinline__ bool last_workgroup_done(DoneState* d, unsigned long long add_sum, uint32_t add_max) {
    const uint32_t n = gridDim.x, G = n < 64u ? n : 64u, g = blockIdx.x % G;
    const uint32_t gs = n / G + (g < n % G ? 1u : 0u);
    const unsigned long long r0 = atomicAdd(&d->acc_sum[g * (kDoneStride / 2)], add_sum);
    const uint32_t r1 = atomicMax(&d->acc_max[g * kDoneStride], add_max);
    uint32_t dep = (uint32_t)r0 | r1;
    asm volatile("v_and_b32 %0, 0, %0" : "+v"(dep));  // 0, but only known once both atomics have returned
    const uint32_t c = atomicAdd(&d->ctr[g * kDoneStride], 1u + dep);
    if (c != gs - 1u) return false;
    uint32_t dep2 = atomicExch(&d->ctr[g * kDoneStride], 0u);  // (= gs: every workgroup of the group is through)
    asm volatile("v_and_b32 %0, 0, %0" : "+v"(dep2));
    if (atomicAdd(&d->ctr[64 * kDoneStride], 1u + dep2) != G - 1u) return false;
    atomicExch(&d->ctr[64 * kDoneStride], 0u);
    return true;
}
// by the threads of the LAST workgroup (slot k handled by thread k < 64): the totals, accumulators reset
__device__ __forceinline__ void collect_done(DoneState* d, int k, unsigned long long& sum, uint32_t& mx) {
    sum = atomicExch(&d->acc_sum[k * (kDoneStride / 2)], 0ull);
    mx = atomicExch(&d->acc_max[k * kDoneStride], 0u);
}

// The end of a pass's record phase, by ONE thread of the last workgroup: what the host wants to know goes into the
// host-mapped mirror (plain stores), the device status is reset for the next pass.
__device__ __forceinline__ void publish_status(const SweepArgs& args, unsigned long long n_small, uint32_t biggest_env) {
    DeviceStatus* st = args.st;
    HostStatus* h = args.hst;
    st->n_small = n_small;  // read by the sweep kernels of this pass when the host did not pick them itself
    const uint32_t over = __hip_atomic_load(&st->max_env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // set by an overflowing environment
    h->flags = __hip_atomic_load(&st->flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->max_env = over > biggest_env ? over : biggest_env;
    h->n_unique[0] = st->n_unique[0];
    h->n_unique[1] = st->n_unique[1];
    h->n_small = n_small;
    h->n_overflow[0] = st->n_overflow[0];
    h->n_overflow[1] = st->n_overflow[1];
    h->max_bound = __hip_atomic_load(&st->max_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h->snapshot_seq = args.seq;
    st->flags = 0u;
    st->max_env = 0u;
    st->n_overflow[0] = 0u;
    st->n_overflow[1] = 0u;
    st->max_bound = 0u;
}


// spread the four 4-bit fields of the low 16 bits of x into four 16-bit fields
__device__ __forceinline__ uint64_t spread4(uint64_t x) {
    // two 32-bit halves, three operations each (and, and / bfe, shift-or); the 64-bit shift-or-mask form is compiled to
    // quarter-rate 32x32 multiplies
    const uint32_t v = (uint32_t)x;
    const uint32_t lo = (v & 0xFu) | ((v & 0xF0u) << 12);
    const uint32_t hi = ((v >> 8) & 0xFu) | ((v & 0xF000u) << 4);
    return ((uint64_t)hi << 32) | lo;
}

// StatisticalDistance::run for Hellinger with a general exponent (statistical_distances.rs:4-10) and Renyi (:31-78) on the
// weighted category counts va / vb with sums sa / sb (pmf.rs:65-83 normalises by the sums).  Inlined (a call from a kernel
// with ~200 live registers costs more in saves and restores than the arithmetic), but with RUNTIME loops over the categories
// on a scratch copy of the counts: one copy of the per-category code, not one per unrolled slot.
//   Hellinger: p^(1/e) = va^(1/e) * sa^(-1/e).  With unit category weights va and sa are integers (< 65536: the count fields
//   are 16 bits), so both factors come from the configuration's tables pow_tab[k] = k^(1/e), pow_tab[65536 + k] = k^(-1/e)
//   (library pow, filled when the configuration is set): one pow per category -- |x - y|^e -- instead of three, none when
//   e is 1, 2, 3 or 4.  Weighted categories take pow_fast for all three.
//   Renyi: ratio^(alpha - 1) = exp((alpha - 1) ln ratio) through the fast log / exp.
__device__ __forceinline__ double sd_generic_fast(int kind, double p0, double p1, const double* va, const double* vb, double sa, double sb, int C,
                                               const double* __restrict__ pow_tab, int tab_half = 65536) {
    const double ia = 1.0 / sa, ib = 1.0 / sb;  // (one reciprocal per side: <= 1 ulp from pmf.rs:78-81's per-category divisions)
    if (kind == SD_HELLINGER) {
        const double e = p0, einv = 1.0 / e;
        const int ie = (e == 1.0 || e == 2.0 || e == 3.0 || e == 4.0) ? (int)e : 0;  // |d|^e by multiplication
        double na1 = 0.0, nb1 = 0.0;
        if (pow_tab) { na1 = pow_tab[tab_half + (int)sa]; nb1 = pow_tab[tab_half + (int)sb]; }
        double dist = 0.0;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            double x, y;
            if (pow_tab) { x = pow_tab[(int)va[c]] * na1; y = pow_tab[(int)vb[c]] * nb1; }
            else { x = pow_fast(va[c] * ia, einv); y = pow_fast(vb[c] * ib, einv); }
            const double d = fabs(x - y);
            dist += ie == 1 ? d : (ie == 2 ? d * d : (ie == 3 ? d * d * d : (ie == 4 ? (d * d) * (d * d) : pow_fast(d, e))));
        }
        return pow_fast(dist / 2.0, einv);
    }
    const double alpha = p0, eps = p1;
    if (alpha == (double)INFINITY) {
        double best = 0.0;
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            const double r = (va[c] * ia + eps) / (vb[c] * ib + eps);
            best = (c == 0 || r >= best) ? r : best;
        }
        return log_fast(best);
    }
    if (alpha == 0.0) {
        double sm = 0.0;
#pragma unroll 1
        for (int c = 0; c < C; ++c) sm += (va[c] > 0.0) ? vb[c] * ib : 0.0;
        return -log_fast(sm);
    }
    double sm = 0.0;
#pragma unroll 1
    for (int c = 0; c < C; ++c) {
        const double x = va[c] * ia;
        sm += x * pow_fast((x + eps) / (vb[c] * ib + eps), alpha - 1.0);
    }
    return log_fast(sm) / (alpha - 1.0);
}

// One pair's weight function.  hyper_exp with <= 4 terms and uniform keep their parameters in (scalar)
// registers; everything else goes through the out-of-line evaluator with the parameter pointer.
struct WfRegs {
    int kind, np, nterm;
    bool fast;
    double a[4], b[4];
    double inv;  // DevConfig::wf_inv
    const double* p;
};
__device__ __forceinline__ WfRegs wf_load(const WfEntry& e, const double* p, double inv) {
    WfRegs w;
    w.inv = inv;
    w.kind = e.kind;
    w.np = e.n_params;
    w.nterm = e.n_params / 2;
    w.p = p;
    w.fast = (e.kind == WF_UNIFORM) || (e.kind == WF_HYPER_EXP && w.nterm <= 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { w.a[i] = 0.0; w.b[i] = 0.0; }
    if (e.kind == WF_UNIFORM) { w.a[0] = p[0]; w.a[1] = p[1]; }
    else if (w.fast) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < w.nterm) { w.a[i] = p[i]; w.b[i] = p[w.nterm + i]; }
    }
    return w;
}
template <bool WFANY>
__device__ __forceinline__ double cdf_dev(const WfRegs& w, double x) {
    if (w.kind == WF_UNIFORM) {  // cdfs.rs:39-45
        if (x < w.a[0]) return 0.0;
        if (x > w.a[1]) return 1.0;
        return (x - w.a[0]) * w.inv;
    }
    if (w.fast) {  // cdfs.rs:5-21, same accumulation order
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < w.nterm) sum += w.a[i] * exp_nonpos(-w.b[i] * x);
        return 1.0 - sum * w.inv;
    }
    if constexpr (WFANY) return cdf_pow_based(w.kind, w.p, w.np, x);
    else return 0.0;  // unreachable: the host routes tables with other weight functions to the WFANY build
}

// StatisticalDistance::run for the non-default distances; out of line so that the sweep kernel stays small.
__device__ __noinline__ double sd_generic(int kind, double p0, double p1, const double* p, const double* q, int C) {
    return sd_eval<0>(kind, p0, p1, [&](int c) { return p[c]; }, [&](int c) { return q[c]; }, C);
}

// from_anchors on lists whose distances do NOT ascend.  The reference never checks (src/locohd.rs:70-77 only looks at dists[0]) and
// its two-pointer loop then still computes a well-defined number: the heads are compared as they come, the tail of the list that
// is left over is walked in list order, and the first tail interval starts at the LAST element of the finished list (:134-221).
// None of the sort-based kernels can reproduce that, so this one walks the loop itself: one lane, the two weighted count vectors
// (pmf.rs:47-63) in LDS, the statistical distance through the generic evaluator on the normalised vectors (pmf.rs:65-88) at
// every step, F(to) - F(from) per interval (weight_function.rs:118-120).  O((n_A + n_B) C) on one lane: an edge path, not a fast one.
__global__ __launch_bounds__(64) void k_anchors_literal(const DevConfig* __restrict__ cfgp, EnvStore ea, EnvStore eb, int nA, int nB, int wfi,
                                                        double* __restrict__ out) {
    extern __shared__ double lit_s[];  // [4][C]: weighted counts of A, of B, the two normalised vectors
    const int C = cfgp->n_categories;
    double *pa = lit_s, *pb = lit_s + C, *qa = lit_s + 2 * C, *qb = lit_s + 3 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 64) lit_s[c] = 0.0;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const DevConfig cfg = *cfgp;
    const WfEntry wf = cfg.wf[wfi];
    const double* prm = cfg.wf_params + wf.offset;
    auto cat_of = [&](const EnvStore& e, int i) -> int { return e.cat16 ? (int)reinterpret_cast<const uint16_t*>(e.cat)[i] : (int)e.cat[i]; };
    auto dist_of = [&](const EnvStore& e, int i) -> double { return u2d(e.key[i]); };
    auto F = [&](double x) -> double { return x == INFINITY ? cfg.wf_finf[wfi] : cdf_eval(wf.kind, prm, wf.n_params, x); };
    auto range = [&](double from, double to) -> double { const double hi = F(to); return hi - F(from); };
    auto H = [&]() -> double {  // pmf.rs:65-88: fresh sums, normalised copies, the configured distance
        double sa = 0.0, sb = 0.0;
        for (int c = 0; c < C; ++c) { sa += pa[c]; sb += pb[c]; }
        for (int c = 0; c < C; ++c) { qa[c] = pa[c] / sa; qb[c] = pb[c] / sb; }
        return sd_generic(cfg.sd_kind, cfg.sd_p0, cfg.sd_p1, qa, qb, C);
    };
    auto add_a = [&](int i) { const int c = cat_of(ea, i); pa[c] += cfg.cat_w[c]; };
    auto add_b = [&](int j) { const int c = cat_of(eb, j); pb[c] += cfg.cat_w[c]; };
    add_a(0);
    add_b(0);
    int i = 0, j = 0;
    double acc = 0.0, prev = 0.0;
    while (i < nA - 1 && j < nB - 1) {
        const double h = H();
        const double a = dist_of(ea, i + 1), b = dist_of(eb, j + 1);
        double nd;
        if (a < b) { ++i; add_a(i); nd = a; }
        else if (a > b) { ++j; add_b(j); nd = b; }
        else { ++i; ++j; add_a(i); add_b(j); nd = a; }  // (equal: NaN distances were refused on the host)
        acc += range(prev, nd) * h;
        prev = nd;
    }
    const double last_a = dist_of(ea, nA - 1), last_b = dist_of(eb, nB - 1);
    if (j < nB - 1) {  // list A is finished
        double h = H();
        ++j;
        acc += range(last_a, dist_of(eb, j)) * h;
        add_b(j);
        while (j < nB - 1) {
            ++j;
            h = H();
            acc += range(dist_of(eb, j - 1), dist_of(eb, j)) * h;
            add_b(j);
        }
        acc += range(last_b, INFINITY) * H();
    } else if (i < nA - 1) {  // list B is finished
        double h = H();
        ++i;
        acc += range(last_b, dist_of(ea, i)) * h;
        add_a(i);
        while (i < nA - 1) {
            ++i;
            h = H();
            acc += range(dist_of(ea, i - 1), dist_of(ea, i)) * h;
            add_a(i);
        }
        acc += range(last_a, INFINITY) * H();
    } else {
        acc += range(last_a, INFINITY) * H();
    }
    *out = acc;
}
void launch_anchors_literal(hipStream_t s, const DevConfig* cfg, int n_categories, const EnvStore& ea, const EnvStore& eb, int nA, int nB, int wfi,
                            double* out) {
    k_anchors_literal<<<1, 64, sizeof(double) * 4 * (size_t)n_categories, s>>>(cfg, ea, eb, nA, nB, wfi, out);
}

// Diagnostic build only (-DLCHD_SWEEP_STAMPS, never the shipped library): per-phase s_memtime deltas summed over all
// wavefronts, read back with lchd_debug_sweep_stamps().
#ifdef LCHD_SWEEP_STAMPS
__device__ unsigned long long g_sweep_stamps[8];
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// register budget: 4 waves/SIMD (<= 128 VGPRs) up to 12 category slots, 3 (<= 168) up to 16, 2 beyond
// INLINE_META (small calls: a few thousand pairs, where launches cost more than arithmetic): the kernel works out every
// pair's record itself instead of reading what k_pair_meta wrote, and its last workgroup publishes the status snapshot --
// ONE launch does the whole sweep phase.
// CNT8 (environments of at most 255 points on both sides, i.e. most pairs at protein-like densities): the packed category
// counts are 8-bit fields, eight per word instead of four -- half the words to scan across the wavefront, to unpack at every
// tile and to keep per lane, and 3 KB less LDS per wavefront, which lets the many-slot variants run at 3 waves per SIMD
// instead of 2.  Pairs with a larger environment are left to the INDIRECT instantiation of the 16-bit kernel.
template <int CMAX, int MODE, int FMODE, bool LDSTAB, bool INDIRECT = false, bool INLINE_META = false, bool CNT8 = false>
#ifndef LCHD_DENSE_PARTTAB
#define LCHD_DENSE_PARTTAB 1024   // entries of the partial sqrt table of the sweeps without full LDS tables (0: none)
#endif
#ifndef LCHD_EXACT_H2_LOOP
#define LCHD_EXACT_H2_LOOP 1
#endif
#ifndef LCHD_C8S_WAVES
#define LCHD_C8S_WAVES 4   // waves per SIMD the 8-bit-count sweep with at most 12 category slots is compiled for
#endif
__global__ __launch_bounds__(64 * kSweepWaves, (MODE == MODE_GEN ? (CMAX <= LCHD_GEN_W3MAX ? 3 : 2) : (CMAX <= 12 ? (CNT8 ? LCHD_C8S_WAVES : 4) : (CMAX <= LCHD_SWEEP_W3MAX ? 3 : (CNT8 ? LCHD_C8_WAVES : 2))))) void k_sweep(SweepArgs args) {
    static_assert(!(INDIRECT && INLINE_META), "the indirect instantiation reads the records of k_pair_meta");
    static_assert(!CNT8 || (MODE == MODE_H2U && FMODE == F_KEY && LDSTAB && !INDIRECT && !INLINE_META), "8-bit counts: default configuration only");
    // Merged events per lane per tile.  The per-tile prologue (staging, merge path, scan of the packed counts, state reload)
    // costs about as many instructions as the events of a 384-event tile themselves, and it grows with the category slots:
    // the variants with many slots (25 categories at 0.05 atoms/A^3: ~416 events per pair) take tiles of 64 x LCHD_EPL_BIG so
    // that such a pair is ONE tile instead of a full one plus a nearly empty one.
    constexpr bool H2_ = (MODE != MODE_GEN);
    constexpr int EPL = (CNT8 && CMAX > 16) ? LCHD_EPL_C8 : (CNT8 && CMAX <= 16) ? LCHD_EPL_C8S : ((H2_ && LDSTAB && CMAX > 16) ? LCHD_EPL_BIG : ((H2_ && !LDSTAB) ? LCHD_EPL_DENSE : ((MODE != MODE_H2U && FMODE == F_KEY) ? LCHD_EPL_WGEN : kSweepEPL))),
                  TILE = 64 * EPL, WPB = kSweepWaves;
    // entries staged per list and tile: a tile's worth -- but the pairs of the 8-bit-count sweep have at most 254 non-anchor
    // points per environment, so 256 entries hold a whole list (4 KB of keys per wave instead of 7) and a tile of 512 events
    // holds a whole pair
    constexpr int LT = CNT8 ? 256 : TILE, LU = LT / 64;
    static_assert(!CNT8 || (kCount8MaxEnv <= LT && 2 * (kCount8MaxEnv - 1) <= TILE), "a whole list per staging buffer, a whole pair per tile");
    static_assert(EPL <= 15, "4-bit chunk-local counters");
    constexpr int FB = CNT8 ? 8 : 16;     // bits per count field
    constexpr int FPW = 64 / FB;          // count fields per u64 word
    constexpr uint64_t FMASK = CNT8 ? 0xFFull : 0xFFFFull;
    constexpr int NW = (CMAX + FPW - 1) / FPW;  // u64 words of count fields per side
    constexpr int NH = (CMAX + 15) / 16;  // u64 words of 4-bit histogram fields per side
    constexpr bool H2 = (MODE != MODE_GEN);
    constexpr int NV = H2 ? 1 : CMAX;     // only the generic path keeps per-category values in registers
    constexpr int NT = LDSTAB ? (CNT8 ? 256 + 8 : kSqrtTab + 8) : 1;  // sqrt(k), 1/sqrt(k) for k <= 512 (255) in LDS; otherwise read from the global tables
    __shared__ double t_sqrt[NT], t_rsqrt[NT];
    constexpr bool PARTTAB = !LDSTAB && H2_ && (LCHD_DENSE_PARTTAB != 0);
    constexpr int kPartTab = LCHD_DENSE_PARTTAB > 0 ? LCHD_DENSE_PARTTAB : 1;
    __shared__ double t_part[PARTTAB ? kPartTab : 1];
    __shared__ double w_s[32], sw_s[32];
    // MODE_GEN, Hellinger with a general exponent, environments of at most kSqrtTab points: k^(1/e) and k^(-1/e) for k <= 512 in
    // LDS (the two look-ups per category and event went to the 1 MB tables in global memory: latency-bound at 2 waves per SIMD)
    constexpr int kGenTab = kSqrtTab + 8;
    __shared__ double t_pow[(MODE == MODE_GEN) ? 2 * kGenTab : 1];
    __shared__ uint64_t sA_[WPB][LT], sB_[WPB][LT];
    __shared__ uint8_t cA_[WPB][LT], cB_[WPB][LT];
    // per-lane category counts of the event loop: [side][word][lane] u64 of four 16-bit fields (a lane only ever touches its own)
    // (13 and more category slots only: up to 12 the register form runs at 4 waves/SIMD, which the extra 3 KB of LDS per wave
    // would cut to 3 -- measured 2-6 % slower -- while from 13 on the LDS form is 4-13 % faster at unchanged occupancy)
#ifndef LCHD_C8_REGCNT
#define LCHD_C8_REGCNT 0
#endif
#ifndef LCHD_C8_LDSCNT_ALL
#define LCHD_C8_LDSCNT_ALL 1   // the 8-bit-count sweeps keep their per-lane counts in LDS for every slot count (<= 12 slots: their 2 KB per wave do not cost a wave of occupancy, and the byte read-modify-write replaces the word select + 4-bit counter chains: C2a sweep 1.577 -> 1.523 ms)
#endif
    constexpr bool LDSCNT = H2 && LDSTAB && (NW > 3 || (CNT8 && LCHD_C8_LDSCNT_ALL)) && (LCHD_LDS_COUNTS != 0) && !(CNT8 && LCHD_C8_REGCNT);  // (16-bit fields: from 13 category slots on)
    __shared__ uint64_t lc_[LDSCNT ? WPB : 1][LDSCNT ? 2 * NW * 64 : 1];
    // When pairs with at most kDuoTile merged events are the majority of a launch, k_sweep_duo sweeps them two per wavefront
    // and the INDIRECT instantiation of this kernel picks the remaining ones out of the pair records; otherwise the plain
    // instantiation sweeps everything.  All three decide from the same word (k_pair_meta: DeviceStatus::n_small).
    const int small_rule = (INDIRECT || CNT8 || !args.forced) ? rule_in_force(args) : -1;
    if (!args.forced) {  // (forced: the host launched exactly the kernels that have to run)
        if constexpr (CNT8) { if (small_rule != 1) return; }          // (the small-pair kernels and their companion: only when
        else if constexpr (INDIRECT) { if (small_rule < 0) return; }  //  the pairs of their rule are the majority)
        else { if (args.duo_enabled && small_rule >= 0) return; }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform => everything derived from it stays scalar
    const DevConfig* __restrict__ cfgp = args.cfg;
    const int C = cfgp->n_categories;
    const double* __restrict__ g_sqrt = args.sqrt_tab;    // [65536] sqrt(k)
    const double* __restrict__ g_rsqrt = args.rsqrt_tab;  // [65536] 1/sqrt(k)
    if constexpr (LDSTAB)
        for (int k = tid; k < NT; k += 64 * WPB) {
            t_sqrt[k] = g_sqrt[k];
            t_rsqrt[k] = g_rsqrt[k];
        }
    if constexpr (PARTTAB)
        for (int k = tid; k < kPartTab; k += 64 * WPB) t_part[k] = g_sqrt[k];
    if (tid < 32) {
        const double wv_ = tid < C ? cfgp->cat_w[tid] : 0.0;
        w_s[tid] = wv_;
        sw_s[tid] = sqrt(wv_);
    }
    bool gen_lds = false;
    if constexpr (MODE == MODE_GEN) {
        gen_lds = args.gen_tab && cfgp->pow_tab && args.env_a.stride <= kSqrtTab && args.env_b.stride <= kSqrtTab;  // (wave-uniform)
        if (gen_lds)
            for (int k = tid; k < kGenTab; k += 64 * WPB) {
                t_pow[k] = cfgp->pow_tab[k];
                t_pow[kGenTab + k] = cfgp->pow_tab[65536 + k];
            }
    }
    __syncthreads();
    uint64_t* sA = sA_[wv];
    uint64_t* sB = sB_[wv];
    uint8_t* cA = cA_[wv];
    uint8_t* cB = cB_[wv];
    unsigned char* lcl = reinterpret_cast<unsigned char*>(lc_[LDSCNT ? wv : 0]) + lane * 8;  // this lane's slot of word 0, side A
    constexpr int kLcSide = NW * 512;  // bytes from a side-A field to the same field of side B

#if LCHD_BIG_SQRT_COMPUTE
    // environments beyond the LDS tables: sqrt(count) is computed (rsq seed + Goldschmidt, <= 1 ulp from the table value)
    // instead of being fetched from the 65536-entry global tables -- four dependent L2 round trips per event otherwise
    // (dense rows: counts below kPartTab -- per-category counts of a 10^4-point row with ten categories stay there until the row's
    //  last tiles -- come from a partial LDS table, larger ones are computed; a per-lane branch, both arms only near a row's end)
    auto sqrt_cnt = [&](int cnt) -> double {
        if constexpr (LDSTAB) return t_sqrt[cnt];
        else if constexpr (PARTTAB) { if (cnt < kPartTab) return t_part[cnt]; else return sqrt_unit((double)cnt); }
        else return sqrt_unit((double)cnt);
    };
    auto rsqrt_cnt = [&](int cnt) -> double {
        if constexpr (LDSTAB) return t_rsqrt[cnt];
        else {
            const double x = (double)cnt;
            double y = __builtin_amdgcn_rsq(x);
            y = y * fma(-0.5 * x, y * y, 1.5);
            y = y * fma(-0.5 * x, y * y, 1.5);
            return y;
        }
    };
#else
    auto sqrt_cnt = [&](int cnt) -> double { if constexpr (LDSTAB) return t_sqrt[cnt]; else return g_sqrt[cnt]; };
    auto rsqrt_cnt = [&](int cnt) -> double { if constexpr (LDSTAB) return t_rsqrt[cnt]; else return g_rsqrt[cnt]; };
#endif
    // sqrt of the weighted count of category c (c may be dynamic)
    auto root_of = [&](int c, int cnt) -> double {
        if constexpr (MODE == MODE_H2W) return sqrt_cnt(cnt) * sw_s[c & 31];
        else return sqrt_cnt(cnt);
    };

#ifdef LCHD_SWEEP_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
    // One 16-byte record per pair (k_pair_meta) replaces the dependent chain anchors -> slot -> len -> first category; the
    // record of the wave's NEXT pair is requested before the current pair is processed.
    // configuration words the loop needs: read once (the compiler must assume the status atomics may alias *cfgp)
    const int n_wf = cfgp->n_wf;
    const double* __restrict__ finf_tab = cfgp->wf_finf;
    const double Finf0 = finf_tab[0];
    const int64_t pstride = (int64_t)gridDim.x * WPB;
    const int64_t total = args.n_pairs;
    int64_t q = (int64_t)blockIdx.x * WPB + wv;
    int biggest_env = 0;  // INLINE_META: largest environment this wave has met
    auto record_of = [&](int64_t pp) -> int4 {  // pp wave-uniform
        if constexpr (INLINE_META) {  // the arithmetic of k_pair_meta
            int64_t ea = pp, eb = pp;
            bool ok = true;
            if (args.anchors) {
                const int64_t ia_ = args.anchors[2 * pp], ib_ = args.anchors[2 * pp + 1];
                ok = !(ia_ < 0 || ib_ < 0 || ia_ >= args.n_slot_a || ib_ >= args.n_slot_b);
                if (ok) { ea = args.slot_a[ia_]; eb = args.slot_b ? args.slot_b[ib_] : pp; }
            }
            int nA_ = 0, nB_ = 0, c0a_ = 0, c0b_ = 0;
            if (ok) {
                nA_ = args.env_a.len[ea];
                nB_ = args.env_b.len[eb];
                if (nA_ > 0 && nB_ > 0) {
                    c0a_ = args.env_a.cat[ea * args.env_a.stride];
                    c0b_ = args.env_b.cat[eb * args.env_b.stride];
                } else {
                    nA_ = nB_ = 0;
                }
            }
            return make_int4((int)ea, (int)eb, nA_ | (c0a_ << 24), nB_ | (c0b_ << 24));
        } else {
            return args.meta[pp];
        }
    };
    // the record lives in four scalar registers; the next one is moved there as soon as its (early) load has returned, so the
    // loop's back edge never waits on vector memory (in particular not on the score store of the pair just finished)
    int mx, my, mz, mw;
    {
        const int4 m0 = record_of(q < total ? q : 0);
        mx = __builtin_amdgcn_readfirstlane(m0.x); my = __builtin_amdgcn_readfirstlane(m0.y);
        mz = __builtin_amdgcn_readfirstlane(m0.z); mw = __builtin_amdgcn_readfirstlane(m0.w);
    }
    int nx = mx, ny = my, nz = mz, nw = mw;
    // INDIRECT: the wave walks blocks of 64 consecutive pairs, every lane holding one record; the pairs that are too large for
    // k_sweep_duo are picked out of a block with a ballot and swept one after the other (the plain instantiation folds all of
    // this away and keeps its one-record-ahead loop)
    int64_t blk = (int64_t)blockIdx.x * WPB + wv, p_cur = 0;
    unsigned long long todo = 0;
    int4 mm = make_int4(0, 0, 0, 0);
    bool ok = true;
    auto advance = [&]() -> bool {
        while (todo == 0) {
            if (blk * 64 >= total) return false;
            const int64_t pp = blk * 64 + lane;
            mm = pp < total ? args.meta[pp] : make_int4(0, 0, 0, 0);
            // the pairs the small-pair kernel of this launch leaves over: more than kDuoTile merged events (k_sweep_duo), or an
            // environment of more than 255 points (the 8-bit-count k_sweep)
            const int za = mm.z & 0xFFFFFF, zb = mm.w & 0xFFFFFF;
            todo = __ballot(za > 0 && !pair_is_small(small_rule, za, zb));
            p_cur = blk * 64;
            blk += pstride;
        }
        const int b = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        p_cur = (p_cur & ~(int64_t)63) + b;
        mx = __builtin_amdgcn_readlane(mm.x, b); my = __builtin_amdgcn_readlane(mm.y, b);
        mz = __builtin_amdgcn_readlane(mm.z, b); mw = __builtin_amdgcn_readlane(mm.w, b);
        return true;
    };
    if constexpr (INDIRECT) ok = advance();
    for (; INDIRECT ? ok : (q < total);
         INDIRECT ? (void)(ok = advance()) : (void)(q += pstride, mx = nx, my = ny, mz = nz, mw = nw)) {
        const int64_t p = INDIRECT ? p_cur : q;
        const int4 mn = INDIRECT ? make_int4(0, 0, 0, 0) : record_of(q + pstride < total ? q + pstride : q);
        auto take_next = [&]() {
            if constexpr (!INDIRECT) {
                nx = __builtin_amdgcn_readfirstlane(mn.x); ny = __builtin_amdgcn_readfirstlane(mn.y);
                nz = __builtin_amdgcn_readfirstlane(mn.z); nw = __builtin_amdgcn_readfirstlane(mn.w);
            }
        };
        const int nA = mz & 0xFFFFFF, nB = mw & 0xFFFFFF;
        if constexpr (INLINE_META) biggest_env = max(biggest_env, max(nA, nB));
        if (nA <= 0 || nB <= 0) {  // anchor out of range (flagged by k_mark_anchors) or overflow / empty environment (flagged by K1)
            if (lane == 0) args.out[p] = nan("");
            take_next();
            continue;
        }
        if constexpr (CNT8) {
            if (max(nA, nB) > kCount8MaxEnv) {  // a count could leave its 8-bit field: the indirect 16-bit kernel takes this pair
                take_next();
                continue;
            }
        }
        const int64_t ea = mx, eb = my;
        const int c0a = (mz >> 24) & 255, c0b = (mw >> 24) & 255;  // categories of the two anchors
        // (a dictionary's key sets, EnvStore::cdf_keys > 1: the set of this pair's weight function)
        const int kset = (FMODE == F_KEY && args.wf_index) ? args.wf_index[p] : 0;
        const int64_t kset_ok = (kset >= 0 && kset < n_wf) ? kset : 0;
        const uint64_t* __restrict__ kA = args.env_a.key + ea * args.env_a.stride + kset_ok * args.env_a.set_stride;
        const uint64_t* __restrict__ kB = args.env_b.key + eb * args.env_b.stride + kset_ok * args.env_b.set_stride;
        const uint8_t* __restrict__ tA = args.env_a.cat + ea * args.env_a.stride;
        const uint8_t* __restrict__ tB = args.env_b.cat + eb * args.env_b.stride;
        const int wfi = args.wf_index ? args.wf_index[p] : 0;
        if (args.wf_index && (wfi < 0 || wfi >= n_wf)) {
            if (lane == 0) { sweep_report(args.hst, ST_BAD_WF); args.out[p] = nan(""); }
            take_next();
            continue;
        }
        constexpr bool WFANY = (FMODE == F_ANY);
        WfRegs wf{};
        if constexpr (FMODE != F_KEY) {
            const WfEntry wfe = cfgp->wf[wfi];
            wf = wf_load(wfe, cfgp->wf_params + wfe.offset, cfgp->wf_inv[wfi]);
            if (kA[0] != 0ull || kB[0] != 0ull) {  // src/locohd.rs:74-77 (F_KEY: checked by the environment kernels)
                if (lane == 0) { sweep_report(args.hst, ST_FIRST_NOT_ZERO); args.out[p] = nan(""); }
                take_next();
                continue;
            }
        }
        auto cdf_of_key = [&](uint64_t k) -> double {
            if constexpr (FMODE == F_KEY) return u2d(k);
            else return cdf_dev<WFANY>(wf, u2d(k));
        };

        bool zero_norm = false;
        // wave-uniform packed integer category counts (16-bit fields), seeded with the two anchors (:82-84)
        uint64_t cntA[NW], cntB[NW];
        {
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                cntA[k] = ((c0a / FPW) == k) ? (1ull << ((c0a % FPW) * FB)) : 0ull;
                cntB[k] = ((c0b / FPW) == k) ? (1ull << ((c0b % FPW) * FB)) : 0ull;
            }
        }

        // ---- per-lane state -------------------------------------------------------------------------
        uint64_t exA[NW], exB[NW];   // packed category counts at the start of this lane's chunk
        uint64_t dA[NH], dB[NH];     // what the chunk has added so far, 4 bits per category
#pragma unroll
        for (int k = 0; k < NH; ++k) dA[k] = dB[k] = 0;
        int totA = 1, totB = 1;      // points seen per side (incl. anchor)
        double ra = 0.0, rb = 0.0;   // H2: 1/sqrt(total weight)
        double na = 0.0, nb = 0.0;   // H2W: total weights
        double D = 0.0;              // H2: sum_c sqrt(a_c * b_c)  (Bhattacharyya numerator)
        double va[NV], vb[NV];       // GEN: weighted category counts (pmf.rs:16-17)

        auto field = [&](const uint64_t (&ex)[NW], int c) -> int {  // static c
            return (int)((ex[c / FPW] >> ((c % FPW) * FB)) & FMASK);
        };
        auto load_state = [&]() {  // registers <- packed counts exA/exB and totals totA/totB
            if constexpr (H2) {
                D = 0.0;
                if constexpr (MODE == MODE_H2W) na = nb = 0.0;
#pragma unroll
                for (int k = 0; k < NW; ++k) {  // padded categories have count 0 on both sides: contribute 0
#pragma unroll
                    for (int f = 0; f < FPW; ++f) {
                        const int c = FPW * k + f;
                        if (c >= CMAX) continue;
                        const int ca = field(exA, c), cb = field(exB, c);
                        if constexpr (MODE == MODE_H2W) {
                            D += w_s[c] * (sqrt_cnt(ca) * sqrt_cnt(cb));
                            na += w_s[c] * (double)ca;
                            nb += w_s[c] * (double)cb;
                        } else {
                            D += sqrt_cnt(ca) * sqrt_cnt(cb);
                        }
                    }
                    // <= 16 slots run at 3-4 waves/SIMD on a tight register budget: one word's table look-ups in flight at a time;
                    // the larger variants (2 waves/SIMD, 256 registers) profit from every second word's being in flight together
                    if constexpr (CMAX <= 16 || CNT8) __builtin_amdgcn_sched_barrier(0);
                    else if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (MODE == MODE_H2W) { ra = 1.0 / sqrt(na); rb = 1.0 / sqrt(nb); }
                else { ra = rsqrt_cnt(totA); rb = rsqrt_cnt(totB); }
            } else {
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    va[c] = w_s[c] * (double)field(exA, c);
                    vb[c] = w_s[c] * (double)field(exB, c);
                }
            }
        };
        // exact squared Hellinger distance in the literal difference-of-roots form (statistical_distances.rs:4-10)
        auto exact_h2 = [&]() -> double {
            double acc2 = 0.0;
            if constexpr (LDSCNT && CMAX > 16 && (LCHD_EXACT_H2_LOOP != 0)) {
                // many slots, counts in LDS: a runtime loop over the count words (one copy of the eight-field body): the rarely
                // taken path no longer sizes the kernel's register allocation
#pragma unroll 1
                for (int k = 0; k < NW; ++k) {
                    const uint64_t wa = *reinterpret_cast<const uint64_t*>(lcl + k * 512), wb = *reinterpret_cast<const uint64_t*>(lcl + kLcSide + k * 512);
#pragma unroll
                    for (int f = 0; f < FPW; ++f) {
                        const int ca = (int)((wa >> (f * FB)) & FMASK), cb = (int)((wb >> (f * FB)) & FMASK);
                        const double d = root_of(FPW * k + f, ca) * ra - root_of(FPW * k + f, cb) * rb;  // (padded slots: 0 - 0)
                        acc2 = fma(d, d, acc2);
                    }
                }
                return 0.5 * acc2;
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                // the chunk-start words go through an empty volatile asm: they do not change during the event loop, and the
                // optimiser otherwise hoists all 2 * CMAX table addresses of this rarely taken path out of the loop, where
                // they occupy registers the common path has to spill for
                uint64_t ea = exA[k], eb = exB[k];
                if constexpr (!LDSCNT) asm volatile("" : "+v"(ea), "+v"(eb));
#pragma unroll
                for (int f = 0; f < FPW; ++f) {
                    const int c = FPW * k + f;
                    if (c >= CMAX) continue;
                    int ca, cb;
                    if constexpr (LDSCNT) {
                        if constexpr (CNT8) {
                            ca = *reinterpret_cast<const uint8_t*>(lcl + k * 512 + f);
                            cb = *reinterpret_cast<const uint8_t*>(lcl + kLcSide + k * 512 + f);
                        } else {
                            ca = *reinterpret_cast<const uint16_t*>(lcl + k * 512 + f * 2);
                            cb = *reinterpret_cast<const uint16_t*>(lcl + kLcSide + k * 512 + f * 2);
                        }
                    } else {
                        ca = (int)((ea >> (f * FB)) & FMASK) + (int)((dA[c >> 4] >> ((c & 15) * 4)) & 15ull);
                        cb = (int)((eb >> (f * FB)) & FMASK) + (int)((dB[c >> 4] >> ((c & 15) * 4)) & 15ull);
                    }
                    const double d = root_of(c, ca) * ra - root_of(c, cb) * rb;  // equal inputs cancel exactly
                    acc2 = fma(d, d, acc2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            return 0.5 * acc2;
        };
        auto distance = [&]() -> double {  // pmf.rs:85-88
            if constexpr (H2) {
                // H^2 = 1 - sum_c sqrt(p_c q_c): O(1) per event from the running D.  Its rounding error (~1e-16
                // absolute) only matters when H^2 itself is tiny, so small values are recomputed in the exact form
                // (which also returns exactly 0 for identical environments).
                double h2 = 1.0 - (ra * rb) * D;
                if (h2 < kExactH2Below) h2 = exact_h2();
                return sqrt_unit(h2);
            } else {
                double sa_ = 0.0, sb_ = 0.0;  // pmf.rs:67-68: fresh sums
#pragma unroll
                for (int c = 0; c < CMAX; ++c) sa_ += va[c];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) sb_ += vb[c];
                if (sa_ == 0.0 || sb_ == 0.0) zero_norm = true;
                const double ia_ = 1.0 / sa_, ib_ = 1.0 / sb_;  // one reciprocal per side (<= 1 ulp from pmf.rs:78-81's per-category divisions)
                const int kind = cfgp->sd_kind;
                if (kind == SD_KS) {  // statistical_distances.rs:12-21, straight from the registers
                    double best = 0.0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) best = fmax(best, fabs(va[c] * ia_ - vb[c] * ib_));  // padded slots give |0 - 0|
                    return best;
                }
                const double prm0 = cfgp->sd_p0, prm1 = cfgp->sd_p1;
                if (kind == SD_KL || (kind == SD_RENYI && prm0 == 1.0)) {  // :23-29 (Renyi with alpha = 1: :36-38)
                    const double eps = kind == SD_KL ? prm0 : prm1;
                    double dist = 0.0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if (c < C) {
                            const double x = va[c] * ia_;
                            dist += x * log_fast((x + eps) / (vb[c] * ib_ + eps));
                        }
                    }
                    return dist;
                }
                if constexpr (MODE == MODE_GEN) {
                    // Hellinger with exponent 1, 2, 3 or 4, unit weights, environments inside the LDS power tables: k^(1/e) from
                    // the tables, |x - y|^e by multiplication -- no transcendental per category, so the per-category code is a dozen
                    // instructions and can be unrolled over the slots straight from the registers (the runtime-loop form below
                    // goes through a scratch copy of the counts)
                    if (gen_lds && kind == SD_HELLINGER && (prm0 == 1.0 || prm0 == 2.0 || prm0 == 3.0 || prm0 == 4.0)) {
                        const int ie = (int)prm0;
                        const double na1 = t_pow[kGenTab + (int)sa_], nb1 = t_pow[kGenTab + (int)sb_];
                        double dist = 0.0;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) {
                            const double d = fabs(t_pow[(int)va[c]] * na1 - t_pow[(int)vb[c]] * nb1);  // (padded slots: |0 - 0|)
                            dist += ie == 1 ? d : (ie == 2 ? d * d : (ie == 3 ? d * d * d : (d * d) * (d * d)));
                        }
                        return pow_fast(dist / 2.0, 1.0 / prm0);
                    }
                }
                // Hellinger with a general exponent, Renyi: runtime loops over a scratch copy of the weighted counts (unrolled per
                // category slot these branches tripled the kernel's size; as an out-of-line call the register saves cost more
                // than the arithmetic)
                double ca_[CMAX], cb_[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) { ca_[c] = va[c]; cb_[c] = vb[c]; }
                if constexpr (MODE == MODE_GEN) {
                    if (gen_lds) return sd_generic_fast(kind, prm0, prm1, ca_, cb_, sa_, sb_, C, t_pow, kGenTab);
                }
                return sd_generic_fast(kind, prm0, prm1, ca_, cb_, sa_, sb_, C, args.gen_tab ? cfgp->pow_tab : nullptr);
            }
        };

#pragma unroll
        for (int k = 0; k < NW; ++k) { exA[k] = cntA[k]; exB[k] = cntB[k]; }
        double F_carry = cdf_of_key(kA[0]);  // F(0): both anchors sit at distance 0
        double H_carry;
        if constexpr (H2) {
            // only the two anchors: both PMFs are point masses => H = 0 if they share the category, else 1 (exactly)
            H_carry = (c0a == c0b) ? 0.0 : 1.0;
        } else {
            load_state();
            H_carry = distance();
        }
        double acc = 0.0;

        const int mA = nA - 1, mB = nB - 1, M = mA + mB;  // non-anchor events
        int ia = 0, ib = 0;
        for (int k0 = 0; k0 < M; k0 += TILE) {
            const int T = min(TILE, M - k0);
            const int nAt = min(LT, mA - ia), nBt = min(LT, mB - ib);
            STAMP(0);
            wave_sync_lds();  // previous tile fully consumed
            // stage the tile: the global loads of BOTH lists are issued before the first LDS write (one memory latency per tile)
            {
                // (wave-uniform base + 32-bit lane offset + immediate: one address register pair serves all loads of a list)
                uint64_t rkA[LU], rkB[LU];
                uint8_t rcA[LU], rcB[LU];
                const char* pkA = reinterpret_cast<const char*>(kA + (1 + ia));
                const char* pkB = reinterpret_cast<const char*>(kB + (1 + ib));
                const uint8_t* pcA = tA + (1 + ia);
                const uint8_t* pcB = tB + (1 + ib);
                const uint32_t lane8 = (uint32_t)lane * 8u, lane1 = (uint32_t)lane;
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const bool in = lane + 64 * u < nAt;
                    rkA[u] = in ? *reinterpret_cast<const uint64_t*>(pkA + lane8 + 512u * u) : 0ull;
                    rcA[u] = in ? pcA[lane1 + 64u * u] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const bool in = lane + 64 * u < nBt;
                    rkB[u] = in ? *reinterpret_cast<const uint64_t*>(pkB + lane8 + 512u * u) : 0ull;
                    rcB[u] = in ? pcB[lane1 + 64u * u] : (uint8_t)0;
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int t = lane + 64 * u;
                    if (t < nAt) { sA[t] = rkA[u]; cA[t] = rcA[u]; }
                }
#pragma unroll
                for (int u = 0; u < LU; ++u) {
                    const int t = lane + 64 * u;
                    if (t < nBt) { sB[t] = rkB[u]; cB[t] = rcB[u]; }
                }
            }
            wave_sync_lds();
            STAMP(1);
            // lane l owns merged events [d0, d1); each lane searches the END of its chunk
            const int epl = (T + 63) >> 6;  // <= EPL (<= 15: the 4-bit histogram fields)
            const int d0 = min(lane * epl, T), d1 = min(d0 + epl, T);
            const int i1 = merge_path(sA, nAt, sB, nBt, d1);
            int i0 = __shfl_up(i1, 1);
            if (lane == 0) i0 = 0;
            const int iend = __builtin_amdgcn_readlane(i1, 63);
            const int j0 = d0 - i0, j1 = d1 - i1;
            STAMP(2);

            // pass 1: 4-bit-per-category histogram of this lane's chunk (at most 8 points per side)
            uint64_t hA[NH], hB[NH];
#pragma unroll
            for (int k = 0; k < NH; ++k) hA[k] = hB[k] = 0;
#if LCHD_PASS1_FUSED
            if constexpr (NH == 1) {
                // one fixed-trip loop over the chunk's (at most EPL) points, A's run first, then B's: the two data-dependent
                // loops it replaces each ran for the longest run of any lane.  hT counts every point, hA only A's.
                const int nAl = i1 - i0, nl = d1 - d0;
                const uint8_t* pa_ = cA + i0;
                const uint8_t* pb_ = cB + (j0 - nAl);
                uint64_t hT = 0;
#pragma unroll
                for (int m = 0; m < EPL; ++m) {
                    if (m < epl) {  // wave-uniform
                        const bool isA = m < nAl;
                        const int ct = (isA ? pa_ : pb_)[m < nl ? m : 0];
                        const uint64_t inc = (m < nl) ? (1ull << ((ct & 15) * 4)) : 0ull;
                        hT += inc;
                        hA[0] += isA ? inc : 0ull;
                    }
                }
                hB[0] = hT - hA[0];
            } else
#endif
            {
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
            }
            STAMP(3);
            // widen to 16-bit fields and exclusive-scan across the wavefront
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const uint64_t va_ = CNT8 ? spread8(hA[(k * 8) / 16] >> (((k * 8) % 16) * 4)) : spread4(hA[(k * 4) / 16] >> (((k * 4) % 16) * 4));
                const uint64_t vb_ = CNT8 ? spread8(hB[(k * 8) / 16] >> (((k * 8) % 16) * 4)) : spread4(hB[(k * 4) / 16] >> (((k * 4) % 16) * 4));
                const uint64_t sa_ = wave_incl_scan_fields(va_), sb_ = wave_incl_scan_fields(vb_);
                exA[k] = cntA[k] + sa_ - va_;
                exB[k] = cntB[k] + sb_ - vb_;
                cntA[k] += readlane_u64(sa_, 63);  // carry for the next tile (scalar)
                cntB[k] += readlane_u64(sb_, 63);
            }
            totA = 1 + ia + i0;
            totB = 1 + ib + j0;
            STAMP(4);
            load_state();
            if constexpr (LDSCNT) {
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    *reinterpret_cast<uint64_t*>(lcl + k * 512) = exA[k];
                    *reinterpret_cast<uint64_t*>(lcl + kLcSide + k * 512) = exB[k];
                }
            }
            STAMP(5);

            // pass 2: sequential sweep of this lane's events.  Branch-free: both list heads stay in registers and the one
            // that was consumed is refilled with a single (address-selected) LDS read.  The packed counts exA/exB stay
            // fixed at their chunk-start values; what the chunk itself adds (<= 6 per category) is kept in 4-bit fields.
            int i = i0, j = j0;
#if LCHD_HEADS_REREAD
            uint64_t ka = sA[i], kb = sB[j];  // both heads are re-read after every event; run ends are tested on the indices
#if LCHD_CAT_HEADS
            int cta = cA[i], ctb = cB[j];     // ... and so are their categories: the event's category is a select, not an LDS round trip behind takeA
#endif
#else
            uint64_t ka = (i < i1) ? sA[i] : kPadKey, kb = (j < j1) ? sB[j] : kPadKey;
#endif
#pragma unroll
            for (int k = 0; k < NH; ++k) dA[k] = dB[k] = 0;
            double Fp = 0.0, Hp = 0.0, firstF = 0.0, local = 0.0;
            for (int e = 0; e < epl; ++e) {
                if (d0 + e < d1) {
#if LCHD_HEADS_REREAD
                    // A-first on ties; an exhausted run cannot be taken.  Two LDS reads per event instead of one, but none of
                    // the selects that steer a single refill into the right head register (the kernel is VALU-issue bound).
                    const bool takeA = (i < i1) & ((j >= j1) | (ka <= kb));
                    const uint64_t key = takeA ? ka : kb;
#if LCHD_CAT_HEADS
                    const int ct = takeA ? cta : ctb;
#else
                    const int ct = (takeA ? cA : cB)[takeA ? i : j];
#endif
                    i += takeA ? 1 : 0;
                    j += takeA ? 0 : 1;
                    ka = sA[i];  // (one past the run's end at most: inside the tile buffers, never used)
                    kb = sB[j];
#if LCHD_CAT_HEADS
                    cta = cA[i];
                    ctb = cB[j];
#endif
#else
                    const bool takeA = (ka <= kb);  // an exhausted list shows the pad key (> every real key)
                    const uint64_t key = takeA ? ka : kb;
#if LCHD_BRANCHFREE_HEADS
                    const int ct = (takeA ? cA : cB)[takeA ? i : j];
                    i += takeA ? 1 : 0;
                    j += takeA ? 0 : 1;
                    {
                        const int nidx = takeA ? i : j, nend = takeA ? i1 : j1;
                        const uint64_t nk = (takeA ? sA : sB)[min(nidx, LT - 1)];
                        const uint64_t nh = nidx < nend ? nk : kPadKey;
                        ka = takeA ? nh : ka;
                        kb = takeA ? kb : nh;
                    }
#else
                    const int ct = takeA ? cA[i] : cB[j];
                    if (takeA) { ++i; ka = (i < i1) ? sA[i] : kPadKey; } else { ++j; kb = (j < j1) ? sB[j] : kPadKey; }
#endif
#endif
                    const double F = cdf_of_key(key);
                    if (e == 0) firstF = F; else local += (F - Fp) * Hp;
                    totA += takeA ? 1 : 0;
                    totB += takeA ? 0 : 1;
                    if constexpr (H2) {
                        // pmf.rs:47-63: one more point of category ct on one side
                        int cntA_, cntB_;
                        if constexpr (LDSCNT) {
                            // counts of category ct on both sides: two 16-bit LDS reads at one address (+ an immediate for side
                            // B); the side that took the event writes its count back incremented.  LDS serves a wave's requests
                            // in order, so the next event of this lane sees the update.
                            if constexpr (CNT8) {
                                unsigned char* pf = lcl + ((ct >> 3) << 9) + (ct & 7);
                                cntA_ = *pf;
                                cntB_ = *(pf + kLcSide);
                                *(pf + (takeA ? 0 : kLcSide)) = (unsigned char)((takeA ? cntA_ : cntB_) + 1);
                            } else {
                            unsigned char* pf = lcl + ((ct >> 2) << 9) + ((ct & 3) << 1);
                            cntA_ = *reinterpret_cast<const uint16_t*>(pf);
                            cntB_ = *reinterpret_cast<const uint16_t*>(pf + kLcSide);
                            *reinterpret_cast<uint16_t*>(pf + (takeA ? 0 : kLcSide)) = (uint16_t)((takeA ? cntA_ : cntB_) + 1);
                            }
                        } else {
                        const int sh = (ct % FPW) * FB, sh4 = (ct & 15) * 4;
                        uint64_t wA = exA[0], wB = exB[0];  // (every category is inside the map: checked at the environment build)
#pragma unroll
                        for (int k = 1; k < NW; ++k) {
                            const bool hit = ((ct / FPW) == k);
                            wA = hit ? exA[k] : wA;
                            wB = hit ? exB[k] : wB;
                        }
                        uint64_t qA = dA[0], qB = dB[0];
                        if constexpr (NH == 2) { qA = (ct & 16) ? dA[1] : qA; qB = (ct & 16) ? dB[1] : qB; }
                        cntA_ = (int)((wA >> sh) & FMASK) + (int)((qA >> sh4) & 15ull);  // before the update
                        cntB_ = (int)((wB >> sh) & FMASK) + (int)((qB >> sh4) & 15ull);
                        const uint64_t inc4 = 1ull << sh4;
                        if constexpr (NH == 2) {
                            dA[0] += (takeA && !(ct & 16)) ? inc4 : 0ull;
                            dA[1] += (takeA && (ct & 16)) ? inc4 : 0ull;
                            dB[0] += (!takeA && !(ct & 16)) ? inc4 : 0ull;
                            dB[1] += (!takeA && (ct & 16)) ? inc4 : 0ull;
                        } else {
                            dA[0] += takeA ? inc4 : 0ull;
                            dB[0] += takeA ? 0ull : inc4;
                        }
                        }
                        const int mine = takeA ? cntA_ : cntB_, other = takeA ? cntB_ : cntA_;
                        double delta = (sqrt_cnt(mine + 1) - sqrt_cnt(mine)) * sqrt_cnt(other);
                        if constexpr (MODE == MODE_H2W) {
                            const double wv_ = w_s[ct & 31];
                            delta *= wv_;
                            na += takeA ? wv_ : 0.0;
                            nb += takeA ? 0.0 : wv_;
                            const double r = 1.0 / sqrt(takeA ? na : nb);
                            ra = takeA ? r : ra;
                            rb = takeA ? rb : r;
                        } else if constexpr (LDSTAB && !LDSCNT) {
                            ra = rsqrt_cnt(totA);  // two table reads instead of one read and five selects (the LDS-count variants already
                                                   // queue five LDS operations per event: there the select form is the faster one)
                            rb = rsqrt_cnt(totB);
                        } else {
                            const double r = rsqrt_cnt(takeA ? totA : totB);
                            ra = takeA ? r : ra;
                            rb = takeA ? rb : r;
                        }
                        D += delta;
                    } else {
                        const double wv_ = w_s[ct & 31];
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) {
                            const bool hit = (c == ct);
                            va[c] += (hit && takeA) ? wv_ : 0.0;
                            vb[c] += (hit && !takeA) ? wv_ : 0.0;
                        }
                    }
                    Hp = distance();
                    Fp = F;
                }
            }
            STAMP(6);
            // stitch lane chunks: (F_first - F_last_of_previous_lane) * H_before_my_first_event
            double prevF = wave_shr1_f64(Fp), prevH = wave_shr1_f64(Hp);  // DPP, no LDS round trip
            if (lane == 0) { prevF = F_carry; prevH = H_carry; }
            if (d0 < d1) local += (firstF - prevF) * prevH;
            acc += local;
            const int last = (T - 1) / epl;  // wave-uniform
            F_carry = readlane_f64(Fp, last);
            H_carry = readlane_f64(Hp, last);
            ia += iend;
            ib += T - iend;
        }
        take_next();  // its load was issued before this pair's tile loads, which have all been waited for
        // wave64 reduction + the last interval to +inf (:165-171,204-210,212-221)
        acc = wave_sum_f64(acc);
        const double Finf = args.wf_index ? finf_tab[wfi] : Finf0;
        acc += (Finf - F_carry) * H_carry;
        const unsigned long long anyzero = __ballot(zero_norm);  // (categories were checked when the environments were built)
        if (lane == 0) {
            if (anyzero) sweep_report(args.hst, ST_ZERO_NORM);
            args.out[p] = acc;
        }
        STAMP(7);
    }
#ifdef LCHD_SWEEP_STAMPS
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_sweep_stamps[k], stamp_acc[k]);
#endif
    if constexpr (INLINE_META) {
        // what k_pair_meta's last workgroup does for the other sweeps: largest environment, status snapshot for the host,
        // device status reset for the next pass (n_small is not counted here: the host keeps its previous hint)
        __shared__ int big_s[WPB];
        __shared__ bool last_s;
        if (lane == 0) big_s[wv] = biggest_env;
        __syncthreads();
        if (tid == 0) {
            int b = big_s[0];
#pragma unroll
            for (int k = 1; k < WPB; ++k) b = max(b, big_s[k]);
            last_s = last_workgroup_done(args.done, 0ull, (uint32_t)b);
        }
        __syncthreads();
        if (last_s && tid < 64) {
            unsigned long long v_;
            uint32_t mx_;
            collect_done(args.done, tid, v_, mx_);
            for (int m = 32; m > 0; m >>= 1) mx_ = max(mx_, (uint32_t)__shfl_xor((int)mx_, m));
            if (tid == 0) publish_status(args, ~0ull, mx_);  // n_small is not counted here: the host keeps its previous hint
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2 for small environments: TWO anchor pairs per wavefront, 32 lanes each.
//
// With environments of ~70-100 points per side (coarse-grained typing, the reference's main use) a pair has ~150 merged
// events: one wavefront per pair spends most of its instructions on the per-tile prologue (staging, merge path, scan, state
// reload, reduction), all of them executed for 64 lanes of which a third idle.  Here every wave-wide instruction serves two
// pairs.  A pair qualifies if it has at most kDuoTile merged events (exactly one tile, no carries between tiles); the
// configuration must be Hellinger-2 with unit category weights, CDF-keyed environments, at most 16 category slots.  The
// host launches this kernel AND k_sweep; k_pair_meta counts the qualifying pairs (DeviceStatus::n_small): when they are
// the majority this kernel sweeps them and k_sweep only the rest, otherwise this kernel returns at once.
// ------------------------------------------------------------------------------------------------
#ifndef LCHD_TEAM_BIG_WAVES
#define LCHD_TEAM_BIG_WAVES 3   // waves per SIMD k_sweep_duo is compiled for with more than 16 category slots
#endif
#define LCHD_DUO_TL 16   // lanes per pair of k_sweep_duo's <= 240-event form: four pairs per wavefront (round 1 / 2: 32 lanes, two pairs, 224 events)
// (the name is historic: round 1 swept TWO pairs per wavefront; with TL = 16 a wavefront sweeps FOUR -- the per-tile prologue, which
// is two thirds of this kernel's instructions at ~150 events per pair, is shared by twice as many pairs, the event loop costs the
// same per pair: C3 459 -> see DESIGN section 4)
// TILE_ = 240: pairs of at most 240 merged events (small_rule 0); TILE_ = 480 (TL = 32): pairs whose environments both have at most
// 255 points and that have at most 480 merged events (small_rule 2) -- the 8-bit-count k_sweep's pairs, two per wavefront (C2a: ~343
// events per pair)
// WGT: category weights other than 1 (pmf.rs:47-63 adds weight[c] per point): H^2 = 1 - sum_c w_c sqrt(a_c b_c) / sqrt(W_a W_b) with the
// weighted totals W = sum_c w_c count_c -- the same integer count fields and tables, one multiplier per category from LDS, two
// running totals and one reciprocal square root per event instead of the two table look-ups of the unit-weight form.
// KSM: the Kolmogorov-Smirnov distance max_c |a_c / N_a - b_c / N_b| (statistical_distances.rs:12-21) with unit weights instead of
// Hellinger-2: every event needs all categories, but as INTEGERS -- max_c |a_c N_b - b_c N_a| over the 8-bit count fields (two 24-bit
// multiplies, one v_sad_u32, one max per category), scaled once by 1 / (N_a N_b) from the reciprocal-root table; no square root.
#ifndef LCHD_COMPANION_GRID
#define LCHD_COMPANION_GRID 2048u   // (measured: 1024 -> 2048: C2a 19.4 -> 16.6 us, C4 47.3 -> 37.3 us per pass; 4096: no further gain) workgroups of the INDIRECT companion sweep (it walks every pair record and sweeps the few the team kernel left)
#endif
#ifndef LCHD_STAGE_PAIRS
#define LCHD_STAGE_PAIRS 1   // the team sweeps stage two buffer entries per lane and round (0: one)
#endif
#ifndef LCHD_WGT_W3
#define LCHD_WGT_W3 0       // 1: ... are compiled for 3 waves per SIMD (170 registers: no spills)
#endif
template <int CMAX, int TL = LCHD_DUO_TL, int TILE_ = kDuoTile, bool WGT = false, bool KSM = false>
__global__ __launch_bounds__(64 * kSweepWaves, ((CMAX <= 16 && !(WGT && CMAX > 8 && (LCHD_WGT_LDSCNT || LCHD_WGT_W3))) ? 4 : LCHD_TEAM_BIG_WAVES)) void k_sweep_duo(SweepArgs args) {
    // (the tile itself -- merge path, chunk histogram, count scans, event loop, stitching -- is lchd_team_tile.h: shared with k_env_sweep)
    using TT = TeamTile<CMAX, TL, TILE_, WGT, KSM>;
    constexpr int TEAMS = TT::TEAMS, EPL = TT::EPL, TILE = TT::TILE, WPB = kSweepWaves, NT = TT::NT, LW = TT::LW;
    constexpr bool LCNT = TT::LCNT;
    constexpr int RULE = TILE_ == kDuoTile ? 0 : 2;
    __shared__ double t_sqrt[NT], t_rsqrt[NT];
    // one buffer per team: list A's points, then list B's (at most TILE together; + the spare entries the head re-reads may touch)
    __shared__ uint64_t s_[WPB][TEAMS][TILE + 2];
    __shared__ uint8_t c_[WPB][TEAMS][TILE + 8];
    __shared__ uint64_t lc_[LCNT ? WPB : 1][LCNT ? LW * 64 : 1];  // (TeamTile::LCNT: per-lane count rows of the event loop)
    __shared__ double w_s[WGT ? 32 : 1];
    if (!args.forced && rule_in_force(args) != RULE) return;  // another rule's pairs are the majority, or none's: k_sweep sweeps everything
    const int tid = threadIdx.x, lane = tid & 63, tl = lane & (TL - 1), team = lane / TL;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const DevConfig* __restrict__ cfgp = args.cfg;
    const double Finf0 = cfgp->wf_finf[0];
    for (int k = tid; k < NT; k += 64 * WPB) {
        t_sqrt[k] = args.sqrt_tab[k];
        t_rsqrt[k] = args.rsqrt_tab[k];
    }
    if constexpr (WGT) {
        if (tid < 32) w_s[tid] = tid < cfgp->n_categories ? cfgp->cat_w[tid] : 0.0;
    }
    __syncthreads();
    uint64_t* sA = s_[wv][team];
    uint8_t* cA = c_[wv][team];
    unsigned char* lcl = reinterpret_cast<unsigned char*>(lc_[LCNT ? wv : 0]) + lane * 8;  // this lane's eight bytes of word 0

    const int64_t pstride = (int64_t)gridDim.x * WPB * TEAMS;
    for (int64_t pb = ((int64_t)blockIdx.x * WPB + wv) * TEAMS; pb < args.n_pairs; pb += pstride) {
        const int64_t p = pb + team;
        const bool live = p < args.n_pairs;
        const int4 m = args.meta[live ? p : pb];
        const bool usable = live && (m.z & 0xFFFFFF) > 0 && (m.w & 0xFFFFFF) > 0;
        const bool mine = !usable || pair_is_small(RULE, m.z & 0xFFFFFF, m.w & 0xFFFFFF);  // larger pairs belong to k_sweep
        const bool valid = usable && mine;
        const int mA = valid ? (m.z & 0xFFFFFF) - 1 : 0, mB = valid ? (m.w & 0xFFFFFF) - 1 : 0, T = mA + mB;  // non-anchor events
        const int c0a = (m.z >> 24) & 255, c0b = (m.w >> 24) & 255;
        // (a dictionary's key sets: the set of this pair's weight function -- k_pair_meta has checked the index of every usable pair)
        int64_t ksetA = 0, ksetB = 0;
        if (args.wf_index) {  // (wave-uniform: configurations with one weight function never multiply)
            const int64_t kset = valid ? args.wf_index[p] : 0;
            ksetA = kset * args.env_a.set_stride;
            ksetB = kset * args.env_b.set_stride;
        }
        // (slot x stride as ONE 32 x 32 -> 64-bit multiply: slots and strides are below 2^31)
        const uint64_t offA = (uint64_t)(uint32_t)m.x * (uint32_t)args.env_a.stride, offB = (uint64_t)(uint32_t)m.y * (uint32_t)args.env_b.stride;
        const uint64_t* __restrict__ kA = args.env_a.key + offA + ksetA;
        const uint64_t* __restrict__ kB = args.env_b.key + offB + ksetB;
        const uint8_t* __restrict__ tA = args.env_a.cat + offA;
        const uint8_t* __restrict__ tB = args.env_b.cat + offB;
        const double F0 = valid ? u2d(kA[0]) : 0.0;            // F(0): both anchors sit at distance 0
#if LCHD_STAGE_PAIRS
        // list B starts at an EVEN entry of the buffer (one unused entry behind an odd list A): the staging below moves two entries per
        // lane and round -- one 16-byte key load, one 16-byte LDS write -- and no pair of entries straddles the two lists
        const int mAe = (mA + 1) & ~1, Tb = mAe + mB;  // <= TILE + 1: the buffers hold TILE + 2 entries
        uint64_t* sB = sA + mAe;
        uint8_t* cB = cA + mAe;
#else
        uint64_t* sB = sA + mA;
        uint8_t* cB = cA + mA;
#endif

        // lane tl of a team owns merged events [d0, d1) of its pair
        const int epl = (T + TL - 1) / TL;  // <= EPL
        int epl_w = __builtin_amdgcn_readlane(epl, 0);  // wave-uniform trip count: the longest of the teams' chunks
#pragma unroll
        for (int k = 1; k < TEAMS; ++k) epl_w = max(epl_w, __builtin_amdgcn_readlane(epl, k * TL));

        wave_sync_lds();  // the previous pairs' tiles are fully consumed
#if LCHD_STAGE_PAIRS
        {   // stage [A's points | pad | B's points]: entries 2 q and 2 q + 1 of the buffer by lane q % TL in round q / TL; all loads before
            // the first LDS write.  A pair's second entry may lie one past its list's last point (still inside the environment's slot or,
            // for the last slot, the workspace's slack): it lands in the pad entry or behind the buffer's used part and is never read.
            constexpr int EPL2 = (EPL + 1) / 2;
            static_assert(2 * TL * EPL2 >= TILE_ + 1, "the rounds cover the buffer's used part (pad entry included)");
            typedef unsigned long long __attribute__((ext_vector_type(2), aligned(8))) key2_t;
            const int epl2 = (Tb + 2 * TL - 1) / (2 * TL);
            int epl2_w = __builtin_amdgcn_readlane(epl2, 0);
#pragma unroll
            for (int k = 1; k < TEAMS; ++k) epl2_w = max(epl2_w, __builtin_amdgcn_readlane(epl2, k * TL));
            key2_t rk[EPL2];
            uint32_t rc[EPL2];
            const uint64_t* kBs = kB - mAe;
            const uint8_t* tBs = tB - mAe;
#pragma unroll
            for (int u = 0; u < EPL2; ++u) { rk[u] = key2_t{0ull, 0ull}; rc[u] = 0u; }
            if (valid) {
#pragma unroll
                for (int u = 0; u < EPL2; ++u) {
                    if (u < epl2_w) {  // (wave-uniform: rounds no team of this wavefront needs are skipped)
                        const int t0 = 2 * (tl + TL * u);
                        const int tt = t0 < Tb ? t0 : 0;  // (beyond the used part: re-read the row's first pair, nothing is written)
                        const bool isA = tt < mAe;
                        const uint64_t* src = (isA ? kA : kBs) + 1 + tt;
                        const uint8_t* csrc = (isA ? tA : tBs) + 1 + tt;
                        rk[u] = *reinterpret_cast<const key2_t*>(src);
                        rc[u] = (uint32_t)csrc[0] | ((uint32_t)csrc[1] << 8);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EPL2; ++u) {
                if (u < epl2_w) {
                    const int t0 = 2 * (tl + TL * u);
                    if (t0 < Tb) {
                        *reinterpret_cast<ulonglong2*>(sA + t0) = ulonglong2{rk[u].x, rk[u].y};
                        *reinterpret_cast<uint16_t*>(cA + t0) = (uint16_t)rc[u];
                    }
                }
            }
        }
#else
        {   // stage [A's points | B's points]: entry t of the buffer is A[1 + t] or B[1 + t - mA]; all loads before the first LDS write.
            // One predicate for the whole team (the pair is swept here), none per entry: an entry beyond T re-reads the pair's last
            // point (index clamped: inside the row) and lands in the buffer's unused tail (t < TILE).
            uint64_t rk[EPL];
            uint8_t rc[EPL];
            const uint64_t* kBs = kB - mA;
            const uint8_t* tBs = tB - mA;
#pragma unroll
            for (int u = 0; u < EPL; ++u) { rk[u] = 0ull; rc[u] = 0; }
            if (valid) {
#pragma unroll
                for (int u = 0; u < EPL; ++u) {
                    if (u < epl_w) {  // (wave-uniform: rounds no team of this wavefront needs are skipped)
                        const int t = min(tl + TL * u, T - 1);  // (T = 0: entry 0 of list A's row, the anchor)
                        const bool isA = t < mA;
                        rk[u] = (isA ? kA : kBs)[1 + t];
                        rc[u] = (isA ? tA : tBs)[1 + t];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
                if (u < epl_w) {
                    const int t = tl + TL * u;
                    sA[t] = rk[u];
                    cA[t] = rc[u];
                }
            }
        }
#endif
        wave_sync_lds();

        const double acc = TT::run(sA, cA, sB, cB, mA, mB, T, epl, epl_w, c0a, c0b, F0, Finf0, t_sqrt, t_rsqrt, w_s, lcl, tl);
        if (tl == TL - 1 && live && mine) args.out[p] = valid ? acc : nan("");  // (categories were checked when the environments were built)
    }
}

constexpr int kWideMaxCat = kWideCategories;  // (512: the per-lane count columns, 256 bytes per category, must fit the LDS)

// Many-categories variant (32 < C <= 255): the per-lane category counts live in LDS columns instead of registers, all
// category loops are runtime loops, and the sqrt tables are read from global memory.  Slower per pair than k_sweep,
// but independent of the category count in registers.  WPB = anchor pairs (wavefronts) per workgroup.
// BIG: environments of more than 65 535 points (the reference sorts and sweeps any length, utils.rs:25-39): the two counts of a
// category are the halves of a 64-bit word instead of a 32-bit one, square roots beyond the 65 536-entry tables are computed.
// HUGE (more than kWideCategories categories, up to kHugeCategories): the per-lane count columns, the carry row and the generic
// distances' normalised vectors live in a global-memory scratch block per workgroup (SweepArgs::wide_scratch) instead of LDS /
// registers, the category weights are read from the configuration.  Nothing here is fast; it exists so that the reference's
// arbitrary category map (src/locohd.rs:312-316) has no upper size short of the 16-bit ids of the store.
template <int MODE, int FMODE, int WPB, bool CAT16 = false, bool BIG = false, bool HUGE = false>  // CAT16: 16-bit category ids in the environment store (EnvStore::cat16)
__global__ __launch_bounds__(64 * WPB) void k_sweep_wide(SweepArgs args) {
    static_assert(!(BIG && CAT16), "the pair record holds a 24-bit length next to an 8-bit category");
    static_assert(!HUGE || (CAT16 && !BIG && WPB == 1), "the global-memory form: 16-bit ids, one wavefront per workgroup");
    constexpr int kLdsCat = HUGE ? 1 : kWideMaxCat;
    using CT = typename std::conditional<CAT16, uint16_t, uint8_t>::type;
    using W = typename std::conditional<BIG, uint64_t, uint32_t>::type;  // count of side A | count of side B << SH
    constexpr int SH = BIG ? 32 : 16;
    constexpr W kOneA = (W)1, kOneB = (W)1 << SH, kMaskA = kOneB - 1;
    constexpr int TILE = kSweepTile;
    constexpr bool LDSTAB = false;
    constexpr bool H2 = (MODE != MODE_GEN);
    constexpr int NT = LDSTAB ? kSqrtTab + 8 : 1;  // sqrt(k), 1/sqrt(k) for k <= 512 in LDS; otherwise read from the global tables
    // Dynamic LDS: per-lane category counts, cnt[wave][category][lane] = count_A | count_B << 16.  A lane only
    // ever touches its own column, and column-major placement makes every access conflict-free.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    __shared__ double t_sqrt[NT], t_rsqrt[NT];
    __shared__ double w_s[kLdsCat], sw_s[kLdsCat];
    __shared__ W carry_[WPB][BIG ? 256 : kLdsCat];  // per category: counts before the current tile (A | B << SH)
    __shared__ uint64_t sA_[WPB][TILE], sB_[WPB][TILE];
    __shared__ CT cA_[WPB][TILE], cB_[WPB][TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform => everything derived from it stays scalar
    const DevConfig* __restrict__ cfgp = args.cfg;
    const int C = cfgp->n_categories;
    const double* __restrict__ g_sqrt = args.sqrt_tab;    // [65536] sqrt(k)
    const double* __restrict__ g_rsqrt = args.rsqrt_tab;  // [65536] 1/sqrt(k)
    if constexpr (LDSTAB)
        for (int k = tid; k < NT; k += 64 * WPB) {
            t_sqrt[k] = g_sqrt[k];
            t_rsqrt[k] = g_rsqrt[k];
        }
    for (int c = tid; c < kLdsCat; c += 64 * WPB) {
        const double wv_ = (!HUGE && c < C) ? cfgp->cat_w[c] : 0.0;
        w_s[c] = wv_;
        sw_s[c] = sqrt(wv_);
    }
    __syncthreads();
    uint64_t* sA = sA_[wv];
    uint64_t* sB = sB_[wv];
    CT* cA = cA_[wv];
    CT* cB = cB_[wv];
    W* carry = carry_[wv];
    W* cnt = reinterpret_cast<W*>(smem_dyn) + (size_t)wv * C * 64 + lane;  // this lane's column: cnt[c * 64]
    // HUGE: the same objects in this workgroup's scratch block: [C][64] count columns | [C] carry row | 2 x [64][C] doubles
    unsigned char* const hbase = HUGE ? args.wide_scratch + (size_t)blockIdx.x * (size_t)args.wide_scratch_per_wave : nullptr;
    W* const cnt_g = reinterpret_cast<W*>(hbase) + lane;
    W* const carry_g = reinterpret_cast<W*>(hbase + (size_t)C * 64 * sizeof(W));
    double* const pn_g = reinterpret_cast<double*>(hbase + (((size_t)C * 65 * sizeof(W) + 15) & ~(size_t)15)) + (size_t)lane * C;
    double* const qn_g = pn_g + (size_t)64 * C;
    auto cnt_at = [&](int c) -> W& { if constexpr (HUGE) return cnt_g[(size_t)c * 64]; else return cnt[c * 64]; };
    auto carry_at = [&](int c) -> W& { if constexpr (HUGE) return carry_g[c]; else return carry[c]; };
    auto wgt = [&](int c) -> double { if constexpr (HUGE) return cfgp->cat_w[c]; else return w_s[c]; };
    auto swgt = [&](int c) -> double { if constexpr (HUGE) return sqrt(cfgp->cat_w[c]); else return sw_s[c]; };
    // (HUGE: the carry row goes from lane 63 to lane 0 through global memory)
    auto wsync = [&]() { if constexpr (HUGE) { __threadfence(); __builtin_amdgcn_wave_barrier(); } wave_sync_lds(); };

    auto sqrt_cnt = [&](int k) -> double {
        if constexpr (LDSTAB) return t_sqrt[k];
        else if constexpr (BIG) return k < 65536 ? g_sqrt[k] : sqrt((double)k);  // (k_fill_sqrt_tables: the same expressions)
        else return g_sqrt[k];
    };
    auto rsqrt_cnt = [&](int k) -> double {
        if constexpr (LDSTAB) return t_rsqrt[k];
        else if constexpr (BIG) return k < 65536 ? g_rsqrt[k] : 1.0 / sqrt((double)k);
        else return g_rsqrt[k];
    };
    auto scan_counts = [&](W x) -> W {  // inclusive wave scan of both halves at once (no half overflows: counts stay below 2^SH)
        if constexpr (BIG) return (W)wave_incl_scan_u32((uint32_t)x) | ((W)wave_incl_scan_u32((uint32_t)(x >> 32)) << 32);
        else return wave_incl_scan_u32(x);
    };

    // One 16-byte record per pair (k_pair_meta) replaces the dependent chain anchors -> slot -> len -> first category; the
    // record of the wave's NEXT pair is requested before the current pair is processed.
    // configuration words the loop needs: read once (the compiler must assume the status atomics may alias *cfgp)
    const int n_wf = cfgp->n_wf;
    const double* __restrict__ finf_tab = cfgp->wf_finf;
    const double Finf0 = finf_tab[0];
    const int64_t pstride = (int64_t)gridDim.x * WPB;
    int64_t p = (int64_t)blockIdx.x * WPB + wv;
    // the record lives in four scalar registers; the next one is moved there as soon as its (early) load has returned, so the
    // loop's back edge never waits on vector memory (in particular not on the score store of the pair just finished)
    int mx, my, mz, mw;
    {
        const int4 m0 = args.meta[p < args.n_pairs ? p : 0];
        mx = __builtin_amdgcn_readfirstlane(m0.x); my = __builtin_amdgcn_readfirstlane(m0.y);
        mz = __builtin_amdgcn_readfirstlane(m0.z); mw = __builtin_amdgcn_readfirstlane(m0.w);
    }
    int nx = mx, ny = my, nz = mz, nw = mw;
    for (; p < args.n_pairs; p += pstride, mx = nx, my = ny, mz = nz, mw = nw) {
        const int4 mn = args.meta[p + pstride < args.n_pairs ? p + pstride : p];
        auto take_next = [&]() {
            nx = __builtin_amdgcn_readfirstlane(mn.x); ny = __builtin_amdgcn_readfirstlane(mn.y);
            nz = __builtin_amdgcn_readfirstlane(mn.z); nw = __builtin_amdgcn_readfirstlane(mn.w);
        };
        const int nA = CAT16 ? (mz & 0xFFFF) : (mz & 0xFFFFFF), nB = CAT16 ? (mw & 0xFFFF) : (mw & 0xFFFFFF);
        if (nA <= 0 || nB <= 0) {  // anchor out of range (flagged by k_mark_anchors) or overflow / empty environment (flagged by K1)
            if (lane == 0) args.out[p] = nan("");
            take_next();
            continue;
        }
        const int64_t ea = mx, eb = my;
        const int c0a = CAT16 ? ((mz >> 16) & 0xFFFF) : ((mz >> 24) & 255), c0b = CAT16 ? ((mw >> 16) & 0xFFFF) : ((mw >> 24) & 255);  // categories of the two anchors
        // (a dictionary's key sets, EnvStore::cdf_keys > 1: the set of this pair's weight function)
        const int kset = (FMODE == F_KEY && args.wf_index) ? args.wf_index[p] : 0;
        const int64_t kset_ok = (kset >= 0 && kset < n_wf) ? kset : 0;
        const uint64_t* __restrict__ kA = args.env_a.key + ea * args.env_a.stride + kset_ok * args.env_a.set_stride;
        const uint64_t* __restrict__ kB = args.env_b.key + eb * args.env_b.stride + kset_ok * args.env_b.set_stride;
        const CT* __restrict__ tA = reinterpret_cast<const CT*>(args.env_a.cat) + ea * args.env_a.stride;
        const CT* __restrict__ tB = reinterpret_cast<const CT*>(args.env_b.cat) + eb * args.env_b.stride;
        const int wfi = args.wf_index ? args.wf_index[p] : 0;
        if (args.wf_index && (wfi < 0 || wfi >= n_wf)) {
            if (lane == 0) { sweep_report(args.hst, ST_BAD_WF); args.out[p] = nan(""); }
            take_next();
            continue;
        }
        constexpr bool WFANY = (FMODE == F_ANY);
        WfRegs wf{};
        if constexpr (FMODE != F_KEY) {
            const WfEntry wfe = cfgp->wf[wfi];
            wf = wf_load(wfe, cfgp->wf_params + wfe.offset, cfgp->wf_inv[wfi]);
            if (kA[0] != 0ull || kB[0] != 0ull) {  // src/locohd.rs:74-77 (F_KEY: checked by the environment kernels)
                if (lane == 0) { sweep_report(args.hst, ST_FIRST_NOT_ZERO); args.out[p] = nan(""); }
                take_next();
                continue;
            }
        }
        auto cdf_of_key = [&](uint64_t k) -> double {
            if constexpr (FMODE == F_KEY) return u2d(k);
            else return cdf_dev<WFANY>(wf, u2d(k));
        };

        bool bad_cat = false, zero_norm = false;
        // ---- per-lane state (category counts live in LDS) ------------------------------------------------
        int totA = 1, totB = 1;      // points seen per side (incl. anchor)
        double ra = 1.0, rb = 1.0;   // H2: 1/sqrt(total weight)
        double na = 0.0, nb = 0.0;   // H2W: total weights
        double D = 0.0;              // H2: sum_c sqrt(a_c * b_c)  (Bhattacharyya numerator)

        // exact squared Hellinger distance in the literal difference-of-roots form (statistical_distances.rs:4-10)
        auto exact_h2 = [&]() -> double {
            double acc2 = 0.0;
            for (int c = 0; c < C; ++c) {
                const W v = cnt_at(c);
                double xa = sqrt_cnt((int)(v & kMaskA)), xb = sqrt_cnt((int)(v >> SH));
                if constexpr (MODE == MODE_H2W) { xa *= swgt(c); xb *= swgt(c); }
                const double d = xa * ra - xb * rb;  // equal inputs cancel exactly
                acc2 = fma(d, d, acc2);
            }
            return 0.5 * acc2;
        };
        auto distance = [&]() -> double {  // pmf.rs:85-88
            if constexpr (H2) {
                // H^2 = 1 - sum_c sqrt(p_c q_c): O(1) per event from the running D.  Its rounding error (~1e-16
                // absolute) only matters when H^2 itself is tiny, so small values are recomputed in the exact form
                // (which also returns exactly 0 for identical environments).
                double h2 = 1.0 - (ra * rb) * D;
                if (h2 < kExactH2Below) h2 = exact_h2();
                return sqrt_unit(h2);
            } else {
                double pn_l[kLdsCat], qn_l[kLdsCat];
                double* const pn = HUGE ? pn_g : pn_l;
                double* const qn = HUGE ? qn_g : qn_l;
                double sa_ = 0.0, sb_ = 0.0;  // pmf.rs:67-68: fresh sums
                for (int c = 0; c < C; ++c) {
                    const W v = cnt_at(c);
                    pn[c] = wgt(c) * (double)(v & kMaskA);
                    qn[c] = wgt(c) * (double)(v >> SH);
                    sa_ += pn[c];
                    sb_ += qn[c];
                }
                if (sa_ == 0.0 || sb_ == 0.0) zero_norm = true;
                const double ia_ = 1.0 / sa_, ib_ = 1.0 / sb_;
                for (int c = 0; c < C; ++c) { pn[c] *= ia_; qn[c] *= ib_; }
                return sd_generic(cfgp->sd_kind, cfgp->sd_p0, cfgp->sd_p1, pn, qn, C);
            }
        };

        // seed with the two anchors (:82-84): carry row and every lane's column
        if (c0a >= C || c0b >= C) bad_cat = true;
        wsync();
        for (int c = lane; c < C; c += 64) carry_at(c) = (c == c0a ? kOneA : (W)0) | (c == c0b ? kOneB : (W)0);
        for (int c = 0; c < C; ++c) cnt_at(c) = (c == c0a ? kOneA : (W)0) | (c == c0b ? kOneB : (W)0);
        if constexpr (H2) {
            // (a category outside the map: bad_cat, the score is NaN whatever is computed here)
            const int w0a = HUGE ? (c0a < C ? c0a : 0) : (c0a & (kWideMaxCat - 1)), w0b = HUGE ? (c0b < C ? c0b : 0) : (c0b & (kWideMaxCat - 1));
            if (c0a == c0b && !bad_cat) D = (MODE == MODE_H2W) ? wgt(w0a) : 1.0;
            if constexpr (MODE == MODE_H2W) {
                na = wgt(w0a);
                nb = wgt(w0b);
                ra = 1.0 / sqrt(na);
                rb = 1.0 / sqrt(nb);
            }
        }
        wsync();
        double F_carry = cdf_of_key(kA[0]);  // F(0): both anchors sit at distance 0
        double H_carry = bad_cat ? 0.0 : distance();
        double acc = 0.0;

        const int mA = nA - 1, mB = nB - 1, M = mA + mB;  // non-anchor events
        int ia = 0, ib = 0;
        for (int k0 = 0; k0 < M; k0 += TILE) {
            const int T = min(TILE, M - k0);
            const int nAt = min(TILE, mA - ia), nBt = min(TILE, mB - ib);
            wsync();  // previous tile fully consumed
            for (int t = lane; t < nAt; t += 64) { sA[t] = kA[1 + ia + t]; cA[t] = tA[1 + ia + t]; }
            for (int t = lane; t < nBt; t += 64) { sB[t] = kB[1 + ib + t]; cB[t] = tB[1 + ib + t]; }
            wave_sync_lds();
            // lane l owns merged events [d0, d1); each lane searches the END of its chunk
            const int epl = (T + 63) >> 6;
            const int d0 = min(lane * epl, T), d1 = min(d0 + epl, T);
            const int i1 = merge_path(sA, nAt, sB, nBt, d1);
            int i0 = __shfl_up(i1, 1);
            if (lane == 0) i0 = 0;
            const int iend = __builtin_amdgcn_readlane(i1, 63);
            const int j0 = d0 - i0, j1 = d1 - i1;

            // pass 1: histogram of this lane's chunk into its LDS column
            for (int c = 0; c < C; ++c) cnt_at(c) = (W)0;
            for (int i = i0; i < i1; ++i) {
                const int ct = cA[i];
                if (ct >= C) bad_cat = true; else cnt_at(ct) += kOneA;
            }
            for (int j = j0; j < j1; ++j) {
                const int ct = cB[j];
                if (ct >= C) bad_cat = true; else cnt_at(ct) += kOneB;
            }
            // per category: wave64 inclusive scan (the carry of earlier tiles enters through lane 0); the exclusive
            // prefix = counts at this lane's first event.  Both 16-bit halves scan at once (every count < 65536).
            totA = 1 + ia + i0;
            totB = 1 + ib + j0;
            if constexpr (H2) D = 0.0;
            if constexpr (MODE == MODE_H2W) na = nb = 0.0;
            for (int c = 0; c < C; ++c) {
                const W own = cnt_at(c);
                const W incl = scan_counts(own + (lane == 0 ? carry_at(c) : (W)0));
                const W excl = incl - own;
                cnt_at(c) = excl;
                if (lane == 63) carry_at(c) = incl;
                if constexpr (H2) {
                    const int ca = (int)(excl & kMaskA), cb = (int)(excl >> SH);
                    if constexpr (MODE == MODE_H2W) {
                        D += wgt(c) * (sqrt_cnt(ca) * sqrt_cnt(cb));
                        na += wgt(c) * (double)ca;
                        nb += wgt(c) * (double)cb;
                    } else {
                        D += sqrt_cnt(ca) * sqrt_cnt(cb);
                    }
                }
            }
            if constexpr (MODE == MODE_H2W) { ra = 1.0 / sqrt(na); rb = 1.0 / sqrt(nb); }
            else if constexpr (MODE == MODE_H2U) { ra = rsqrt_cnt(totA); rb = rsqrt_cnt(totB); }

            // pass 2: sequential sweep of this lane's events (the two list heads stay in registers)
            int i = i0, j = j0;
            uint64_t ka = (i < i1) ? sA[i] : kPadKey, kb = (j < j1) ? sB[j] : kPadKey;
            double Fp = 0.0, Hp = 0.0, firstF = 0.0, local = 0.0;
            for (int e = d0; e < d1; ++e) {
                const bool takeA = (ka <= kb);  // an exhausted list shows the pad key (> every real key)
                const uint64_t key = takeA ? ka : kb;
                const int ct = takeA ? cA[i] : cB[j];
                if (takeA) { ++i; ka = (i < i1) ? sA[i] : kPadKey; } else { ++j; kb = (j < j1) ? sB[j] : kPadKey; }
                const double F = cdf_of_key(key);
                if (e == d0) firstF = F; else local += (F - Fp) * Hp;
                // pmf.rs:47-63: one more point of category ct on one side
                const bool okc = ct < C;
                const int cs = okc ? ct : 0;
                const W old = cnt_at(cs);
                cnt_at(cs) = old + (okc ? (takeA ? kOneA : kOneB) : (W)0);
                totA += takeA ? 1 : 0;
                totB += takeA ? 0 : 1;
                if constexpr (H2) {
                    const int cntA_ = (int)(old & kMaskA), cntB_ = (int)(old >> SH);
                    const int mine = takeA ? cntA_ : cntB_, other = takeA ? cntB_ : cntA_;
                    double delta = (sqrt_cnt(mine + 1) - sqrt_cnt(mine)) * sqrt_cnt(other);
                    if constexpr (MODE == MODE_H2W) {
                        const double wv_ = wgt(cs);
                        delta *= wv_;
                        na += takeA ? wv_ : 0.0;
                        nb += takeA ? 0.0 : wv_;
                        const double r = 1.0 / sqrt(takeA ? na : nb);
                        ra = takeA ? r : ra;
                        rb = takeA ? rb : r;
                    } else {
                        const double r = rsqrt_cnt(takeA ? totA : totB);
                        ra = takeA ? r : ra;
                        rb = takeA ? rb : r;
                    }
                    D += okc ? delta : 0.0;
                }
                Hp = distance();
                Fp = F;
            }
            // stitch lane chunks: (F_first - F_last_of_previous_lane) * H_before_my_first_event
            double prevF = wave_shr1_f64(Fp), prevH = wave_shr1_f64(Hp);  // DPP, no LDS round trip
            if (lane == 0) { prevF = F_carry; prevH = H_carry; }
            if (d0 < d1) local += (firstF - prevF) * prevH;
            acc += local;
            const int last = (T - 1) / epl;  // wave-uniform
            F_carry = readlane_f64(Fp, last);
            H_carry = readlane_f64(Hp, last);
            ia += iend;
            ib += T - iend;
        }
        take_next();  // its load was issued before this pair's tile loads, which have all been waited for
        // wave64 reduction + the last interval to +inf (:165-171,204-210,212-221)
        acc = wave_sum_f64(acc);
        const double Finf = args.wf_index ? finf_tab[wfi] : Finf0;
        acc += (Finf - F_carry) * H_carry;
        const unsigned long long anybad = __ballot(bad_cat), anyzero = __ballot(zero_norm);
        if (lane == 0) {
            if (anybad) { sweep_report(args.hst, ST_BAD_CATEGORY); acc = nan(""); }
            if (anyzero) sweep_report(args.hst, ST_ZERO_NORM);
            args.out[p] = acc;
        }
    }
}


template <int MODE, int FMODE, bool LDSTAB>
static void launch_sweep_mode(hipStream_t s, int cmax, unsigned grid, const SweepArgs& a) {
    constexpr int NTH = 64 * kSweepWaves;
    if (cmax <= 8) k_sweep<8, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 12) k_sweep<12, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 16) k_sweep<16, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 20) k_sweep<20, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 24) k_sweep<24, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else if (cmax <= 28) k_sweep<28, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
    else k_sweep<32, MODE, FMODE, LDSTAB><<<grid, NTH, 0, s>>>(a);
}
template <int MODE, bool LDSTAB>
static void launch_sweep_f(hipStream_t s, int cmax, unsigned grid, int fmode, const SweepArgs& a) {
    if (fmode == F_KEY) launch_sweep_mode<MODE, F_KEY, LDSTAB>(s, cmax, grid, a);
    else if (fmode == F_FAST && MODE == MODE_H2U) launch_sweep_mode<MODE_H2U, F_FAST, LDSTAB>(s, cmax, grid, a);
    else launch_sweep_mode<MODE, F_ANY, LDSTAB>(s, cmax, grid, a);
}

template <int MODE>
static void launch_sweep_wide(hipStream_t s, int n_cat, int64_t n_pairs, int fmode, const SweepArgs& a) {
    // dynamic LDS = WPB * C * 64 * 4 bytes of per-lane count columns
    if (a.env_a.stride > 65535 || a.env_b.stride > 65535) {  // environments of more than 65 535 points: 64-bit count words (<= 255 categories: the host checks)
        const unsigned grid = (unsigned)(n_pairs < 8192 ? n_pairs : 8192);
        const size_t dyn = (size_t)n_cat * 512;
        if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1, false, true><<<grid, 64, dyn, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 1, false, true><<<grid, 64, dyn, s>>>(a);
    } else if (n_cat <= 64) {
        const int64_t blocks = (n_pairs + 3) / 4;
        const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
        const size_t dyn = (size_t)4 * n_cat * 256;
        if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 4><<<grid, 256, dyn, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 4><<<grid, 256, dyn, s>>>(a);
    } else if (n_cat > kWideCategories) {  // the global-memory form (the caller has checked that the scratch block exists)
        const unsigned grid = (unsigned)std::min<int64_t>(n_pairs, a.wide_scratch_waves);
        if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1, true, false, true><<<grid, 64, 0, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 1, true, false, true><<<grid, 64, 0, s>>>(a);
    } else {
        const unsigned grid = (unsigned)(n_pairs < 8192 ? n_pairs : 8192);
        const size_t dyn = (size_t)n_cat * 256;  // (> 64 KB from 257 categories' worth on: init_device_kernels raised the limit)
        if (a.env_a.cat16) {  // more than 255 categories: 16-bit ids in the store
            if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1, true><<<grid, 64, dyn, s>>>(a);
            else k_sweep_wide<MODE, F_ANY, 1, true><<<grid, 64, dyn, s>>>(a);
        } else if (fmode == F_KEY) k_sweep_wide<MODE, F_KEY, 1><<<grid, 64, dyn, s>>>(a);
        else k_sweep_wide<MODE, F_ANY, 1><<<grid, 64, dyn, s>>>(a);
    }
}

// One record per anchor pair for the sweep kernels: {environment slot A, slot B, n_A | category of anchor A << 24,
// n_B | category of anchor B << 24}; n = 0 marks a pair the sweep must answer with NaN (anchor index out of range -- already
// flagged by k_mark_anchors -- or an environment that overflowed / is empty -- flagged by K1).
__global__ void k_pair_meta(SweepArgs args) {
    int n_duo = 0, n_c8 = 0, biggest = 0;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < args.n_pairs; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t ea = p, eb = p;
        bool ok = true;
        if (args.anchors) {
            const int64_t ia_ = args.anchors[2 * p], ib_ = args.anchors[2 * p + 1];
            ok = !(ia_ < 0 || ib_ < 0 || ia_ >= args.n_slot_a || ib_ >= args.n_slot_b);
            if (ok) { ea = args.slot_a[ia_]; eb = args.slot_b ? args.slot_b[ib_] : p; }
        }
        if (args.wf_index && args.env_a.cdf_keys > 1) {
            // key sets: the pair's weight-function index picks the set its sweep reads -- an index outside the dictionary is reported
            // here and the pair marked unusable (the sweep answers NaN)
            const int wfi = args.wf_index[p];
            if (wfi < 0 || wfi >= args.env_a.cdf_keys) { ok = false; atomicOr(&args.st->flags, ST_BAD_WF); }
        }
        int nA = 0, nB = 0, c0a = 0, c0b = 0;
        if (ok) {
            nA = args.env_a.len[ea];
            nB = args.env_b.len[eb];
            if (nA > 0 && nB > 0) {
                if (args.env_a.cat16) {
                    c0a = reinterpret_cast<const uint16_t*>(args.env_a.cat)[ea * args.env_a.stride];
                    c0b = reinterpret_cast<const uint16_t*>(args.env_b.cat)[eb * args.env_b.stride];
                } else if (args.env_a.cat0 && args.env_b.cat0) {  // (the grouped environment kernel's header array: L2-resident, unlike the store)
                    c0a = args.env_a.cat0[ea];
                    c0b = args.env_b.cat0[eb];
                } else {
                    c0a = args.env_a.cat[ea * args.env_a.stride];
                    c0b = args.env_b.cat[eb * args.env_b.stride];
                }
            } else {
                nA = nB = 0;
            }
        }
        // (16-bit categories, k_sweep_wide only: the environment length -- at most 65535 -- in the low half, the category above it)
        if (args.env_a.cat16) args.meta[p] = make_int4((int)ea, (int)eb, nA | (c0a << 16), nB | (c0b << 16));
        else args.meta[p] = make_int4((int)ea, (int)eb, nA | (c0a << 24), nB | (c0b << 24));
        biggest = max(biggest, max(nA, nB));
        // pairs k_sweep_duo takes: everything that fits its tile, and the unusable ones (it writes their NaN); pairs the
        // 8-bit-count sweep takes: both environments of at most 255 points.  Both are counted whichever rule this pass uses:
        // the host picks the next pass's kernels from them.
        n_duo += (nA + nB - 2 <= kDuoTileFwd) ? 1 : 0;
        n_c8 += pair_is_small(args.c8_rule, nA, nB) ? 1 : 0;
    }
    // pairs that fit one 32-lane tile: if they are the majority, k_sweep_duo sweeps them and k_sweep only the rest.  One
    // partial count per workgroup; the workgroup that finishes LAST folds them, publishes what the host wants to know into
    // the host-mapped mirror and resets the device status for the next pass -- no separate summing kernel, no memset before
    // a pass, no copy after it.
    __shared__ int big_s[4];
    __shared__ bool last_s;
    unsigned long long n_both = (unsigned long long)n_duo | ((unsigned long long)n_c8 << 32);  // (a launch has < 2^32 pairs)
    for (int m = 32; m > 0; m >>= 1) { n_both += shfl_u64(n_both, (threadIdx.x & 63) ^ m); biggest = max(biggest, __shfl_xor(biggest, m)); }
    __shared__ unsigned long long both_s[4];
    if ((threadIdx.x & 63) == 0) { both_s[threadIdx.x >> 6] = n_both; big_s[threadIdx.x >> 6] = biggest; }
    __syncthreads();
    if (threadIdx.x == 0)  // (largest environment of the pass: the host lets its capacity hint decay with it)
        last_s = last_workgroup_done(args.done, both_s[0] + both_s[1] + both_s[2] + both_s[3],
                                     (uint32_t)max(max(big_s[0], big_s[1]), max(big_s[2], big_s[3])));
    __syncthreads();
    if (!last_s || threadIdx.x >= 64) return;
    unsigned long long v;
    uint32_t mx;
    collect_done(args.done, threadIdx.x, v, mx);
    for (int m = 32; m > 0; m >>= 1) { v += shfl_u64(v, threadIdx.x ^ m); mx = max(mx, (uint32_t)__shfl_xor((int)mx, m)); }
    if (threadIdx.x == 0) {
        const unsigned long long duo = v & 0xFFFFFFFFull, c8 = v >> 32;
        args.hst->n_duo = duo;
        args.hst->n_c8 = c8;
        args.st->n_c8 = c8;
        publish_status(args, args.small_rule ? c8 : duo, mx);  // (small_rule 1 or 2 == c8_rule whenever it is not 0)
    }
}

// A team kernel (tile240: four pairs of <= 240 events per wavefront, else two 8-bit-count pairs of <= 480), the INDIRECT companion for
// the pairs its rule leaves over, and -- a pass without a hint -- the second team rule's kernel (tgrid != 0).
// TM: 0 Hellinger-2 with unit weights, 1 with category weights, 2 Kolmogorov-Smirnov with unit weights
template <int CM, int TM>
static void launch_team(hipStream_t s, bool tile240, unsigned dgrid, unsigned bgrid, bool others, unsigned tgrid, const SweepArgs& a) {
    constexpr int NTH = 64 * kSweepWaves;
    constexpr bool WGT = TM == 1, KSM = TM == 2;
    if (tile240) k_sweep_duo<CM, LCHD_DUO_TL, kDuoTile, WGT, KSM><<<dgrid, NTH, 0, s>>>(a);
    else k_sweep_duo<CM, 32, kTeam8Tile, WGT, KSM><<<dgrid, NTH, 0, s>>>(a);
    if (others) {
        if constexpr (KSM) k_sweep<CM, MODE_GEN, F_KEY, false, true><<<bgrid, NTH, 0, s>>>(a);
        else k_sweep<CM, WGT ? MODE_H2W : MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
    }
    if (tgrid) k_sweep_duo<CM, 32, kTeam8Tile, WGT, KSM><<<tgrid, NTH, 0, s>>>(a);
}

int launch_sweep(hipStream_t s, const Tuning& t, int n_categories, bool hellinger2, bool unit_weights, bool wf_pow, int sweep_hint,
                 const SweepArgs& a_in) {
    if (a_in.n_pairs <= 0) return 0;
    SweepArgs a = a_in;
    a.duo_enabled = 0;
    a.forced = 0;
    a.gen_tab = (unit_weights && !t.no_tables) ? 1 : 0;  // (MODE_GEN, Hellinger with a general exponent: the configuration's power tables apply)
    if (t.force_generic) hellinger2 = false;  // test hook
    if (a.n_pairs <= kInlineMetaPairs && !t.no_inline_meta && hellinger2 && unit_weights && n_categories <= 32 && !t.force_wide &&
        a.env_a.cdf_keys && a.env_b.cdf_keys && a.env_a.stride <= kSqrtTab && a.env_b.stride <= kSqrtTab && !t.force_bigenv) {
        // small call, default configuration: one launch (records worked out by the sweep itself, one pair per wavefront)
        const int cm = std::max(n_categories, t.force_cmax);
        const unsigned g = (unsigned)((a.n_pairs + kSweepWaves - 1) / kSweepWaves);
        constexpr int NTH = 64 * kSweepWaves;
        if (cm <= 8) k_sweep<8, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
        else if (cm <= 12) k_sweep<12, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
        else if (cm <= 16) k_sweep<16, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
        else if (cm <= 20) k_sweep<20, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
        else if (cm <= 24) k_sweep<24, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
        else if (cm <= 28) k_sweep<28, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
        else k_sweep<32, MODE_H2U, F_KEY, true, false, true><<<g, NTH, 0, s>>>(a);
        return 0;
    }
    const bool wide = n_categories > 32 || t.force_wide || a.env_a.stride > 65535 || a.env_b.stride > 65535;  // (long environments: the 64-bit-count form of the wide sweep)
    const int64_t blocks = (a.n_pairs + kSweepWaves - 1) / kSweepWaves;
    // grid-stride: LDS tables are built once per block.  8192 workgroups = 8 rounds of the 1024 that are resident at a time: finer
    // than that the table loads show, coarser the last round's imbalance does (measured on C2a: 4096 +2.8 %, 16384 +0.5 %)
    const int64_t gcap = t.sweep_grid > 0 ? t.sweep_grid : 8192;
    // ... the team sweeps (shorter iterations, a smaller table load per workgroup): 16384 (C2a 1.3648 -> 1.358 ms, C3 0.7835 -> 0.7769; 32768: no further gain)
    const int64_t tcap = t.sweep_grid > 0 ? t.sweep_grid : 16384;
    const unsigned grid = (unsigned)(blocks < gcap ? blocks : gcap);
    const int cmax = std::max(n_categories, t.force_cmax);  // (force_cmax: test hook)
    const bool small = a.env_a.stride <= kSqrtTab && a.env_b.stride <= kSqrtTab && !t.force_bigenv;  // every count fits the LDS tables
    const int fmode = (a.env_a.cdf_keys && a.env_b.cdf_keys) ? F_KEY : (wf_pow ? F_ANY : F_FAST);
    // Two kernels for "small" pairs exist for the default configuration (Hellinger-2, unit weights, CDF-keyed environments):
    // k_sweep_duo (two pairs of <= 240 merged events per wavefront, <= 16 category slots) and the 8-bit-count k_sweep (both
    // environments <= 255 points, more than 16 slots); the INDIRECT 16-bit k_sweep takes what they leave over.
    // ... and, up to 16 slots, for category weights other than 1 (the WGT instantiations of the team kernels; the one-pair-per-wavefront
    // 8-bit-count sweep has no weighted form, so both team rules must be available)
    const bool weighted_team = !unit_weights && cmax <= 16 && !t.no_duo && !t.no_c8_team && !t.no_count8 && (t.c8_team_max == 0 || t.c8_team_max >= cmax);
    // ... and for the Kolmogorov-Smirnov distance with unit weights (SweepArgs::sd_fast == 3: the KSM instantiations)
    const bool ks_team = !hellinger2 && a.sd_fast == 3 && unit_weights && cmax <= 16 && !t.no_duo && !t.no_c8_team && !t.no_count8 && !t.force_generic &&
                         (t.c8_team_max == 0 || t.c8_team_max >= cmax);
    const bool fast_cfg = !wide && ((hellinger2 && (unit_weights || weighted_team)) || ks_team) && small && fmode == F_KEY;  // (F_KEY with a weight-function dictionary: the store holds one key set per function)
    // sweep_hint (what k_pair_meta counted in the previous pass of this configuration): 0 = nothing known, else
    // 4 | (pairs of <= 240 events were the majority ? 1 : 0) | (pairs with both environments <= 255 points were ? 2 : 0).
    // Up to 16 slots k_sweep_duo is the first choice and the 8-bit-count sweep the second (C2a: environments of ~170 points,
    // pairs of ~340 events -- too long for a 32-lane tile, but their counts fit 8 bits: 2 count words per side instead of 3);
    // above 16 slots only the 8-bit-count sweep exists.
    const int hint_bits = t.no_sweep_hint ? 0 : sweep_hint;
    const bool known = (hint_bits & 4) != 0, duo_major = (hint_bits & 1) != 0, c8_major = (hint_bits & 2) != 0;
    const bool c8_small_slots = fast_cfg && cmax <= 16 && !t.no_count8 && known && !(duo_major && !t.no_duo) && c8_major;
    const bool use_duo = fast_cfg && cmax <= 16 && !t.no_duo && !c8_small_slots;
    const bool use_c8 = fast_cfg && !t.no_count8 && (cmax > 16 || c8_small_slots);
    // up to 16 slots the 8-bit-count pairs are swept two per wavefront (rule 2: and at most 480 merged events)
    const bool team_ok = !t.no_c8_team && cmax <= (t.c8_team_max > 0 ? t.c8_team_max : 32);
    const bool c8_team = use_c8 && team_ok;
    a.c8_rule = team_ok ? 2 : 1;  // (what k_pair_meta counts as n_c8 -- whichever small-pair kernel this pass uses)
    a.small_rule = use_c8 ? a.c8_rule : 0;
    // no hint and up to 16 slots: k_sweep_duo's rule first, the two-pairs-per-wavefront 8-bit-count rule second
    a.second_rule = (!known && use_duo && fast_cfg && !t.no_count8 && !t.no_c8_team) ? 2 : 0;
    const int hint = !known ? 0 : ((use_c8 ? c8_major : duo_major) ? 1 : 2);
    // ... | 8 (EVERY pair of the previous pass had at most 240 events) | 16 (... both environments <= 255 points): the companion
    // launch for the larger pairs would find nothing to do and is left out; the host checks the counts of THIS pass afterwards
    // and repeats it with the full launch set if a larger pair turned up after all (the returned bit 2 says the launch was left out)
    const bool no_others = hint == 1 && (hint_bits & (use_c8 ? 16 : 8)) != 0;
    const int info = (use_c8 ? 1 : 0) | (no_others ? 2 : 0);
    {
        const int64_t nb = (a.n_pairs + 255) / 256;
        const int mgrid = (int)(nb < kMetaPartials ? nb : kMetaPartials);
        k_pair_meta<<<mgrid, 256, 0, s>>>(a);
    }
    if (wide) {
        const int fm = (a.env_a.cdf_keys && a.env_b.cdf_keys) ? F_KEY : F_ANY;
        if (!hellinger2) launch_sweep_wide<MODE_GEN>(s, n_categories, a.n_pairs, fm, a);
        else if (unit_weights) launch_sweep_wide<MODE_H2U>(s, n_categories, a.n_pairs, fm, a);
        else launch_sweep_wide<MODE_H2W>(s, n_categories, a.n_pairs, fm, a);
        return 0;
    }
    if (use_duo || use_c8) {
        // Without a hint the small-pair kernel, its companion and the plain sweep are all launched and the number of small
        // pairs (k_pair_meta) decides on the device which of them do the work; with the hint of the previous pass only the
        // kernels that will work are launched.
        a.forced = hint != 0;
        if (hint != 2) {
            a.duo_enabled = 1;
            const unsigned bgrid = grid < LCHD_COMPANION_GRID ? grid : LCHD_COMPANION_GRID;  // the listed (larger) pairs are a minority whenever this launch does anything
            constexpr int NTH = 64 * kSweepWaves;
            if (use_duo) {
                constexpr int kTeamPairs = (64 / LCHD_DUO_TL) * kSweepWaves;  // pairs per workgroup and round
                const int64_t dblocks = (a.n_pairs + kTeamPairs - 1) / kTeamPairs;
                const unsigned dgrid = (unsigned)(dblocks < tcap ? dblocks : tcap);
                const int64_t tblocks = (a.n_pairs + 2 * kSweepWaves - 1) / (2 * kSweepWaves);
                const unsigned tgrid = (unsigned)(tblocks < tcap ? tblocks : tcap);
                if (ks_team) {
                    if (cmax <= 8) launch_team<8, 2>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                    else if (cmax <= 12) launch_team<12, 2>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                    else launch_team<16, 2>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                } else if (unit_weights) {
                    if (cmax <= 8) launch_team<8, 0>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                    else if (cmax <= 12) launch_team<12, 0>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                    else launch_team<16, 0>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                } else {
                    if (cmax <= 8) launch_team<8, 1>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                    else if (cmax <= 12) launch_team<12, 1>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                    else launch_team<16, 1>(s, true, dgrid, bgrid, !no_others, a.second_rule ? tgrid : 0u, a);
                }
            } else if (c8_team) {
                constexpr int kTeamPairs = 2 * kSweepWaves;
                const int64_t dblocks = (a.n_pairs + kTeamPairs - 1) / kTeamPairs;
                const unsigned dgrid = (unsigned)(dblocks < tcap ? dblocks : tcap);
                if (ks_team) {
                    if (cmax <= 8) launch_team<8, 2>(s, false, dgrid, bgrid, !no_others, 0u, a);
                    else if (cmax <= 12) launch_team<12, 2>(s, false, dgrid, bgrid, !no_others, 0u, a);
                    else launch_team<16, 2>(s, false, dgrid, bgrid, !no_others, 0u, a);
                } else if (!unit_weights) {
                    if (cmax <= 8) launch_team<8, 1>(s, false, dgrid, bgrid, !no_others, 0u, a);
                    else if (cmax <= 12) launch_team<12, 1>(s, false, dgrid, bgrid, !no_others, 0u, a);
                    else launch_team<16, 1>(s, false, dgrid, bgrid, !no_others, 0u, a);
                }
                else if (cmax <= 8) launch_team<8, 0>(s, false, dgrid, bgrid, !no_others, 0u, a);
                else if (cmax <= 12) launch_team<12, 0>(s, false, dgrid, bgrid, !no_others, 0u, a);
                else if (cmax <= 16) launch_team<16, 0>(s, false, dgrid, bgrid, !no_others, 0u, a);
                else if (cmax <= 20) launch_team<20, 0>(s, false, dgrid, bgrid, !no_others, 0u, a);
                else if (cmax <= 24) launch_team<24, 0>(s, false, dgrid, bgrid, !no_others, 0u, a);
                else if (cmax <= 28) launch_team<28, 0>(s, false, dgrid, bgrid, !no_others, 0u, a);
                else launch_team<32, 0>(s, false, dgrid, bgrid, !no_others, 0u, a);
            } else {
                if (cmax <= 8) { k_sweep<8, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a); if (!no_others) k_sweep<8, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a); }
                else if (cmax <= 12) { k_sweep<12, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a); if (!no_others) k_sweep<12, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a); }
                else if (cmax <= 16) { k_sweep<16, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a); if (!no_others) k_sweep<16, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a); }
                else if (cmax <= 20) { k_sweep<20, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a); if (!no_others) k_sweep<20, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a); }
                else if (cmax <= 24) { k_sweep<24, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a); if (!no_others) k_sweep<24, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a); }
                else if (cmax <= 28) { k_sweep<28, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a); if (!no_others) k_sweep<28, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a); }
                else { k_sweep<32, MODE_H2U, F_KEY, true, false, false, true><<<grid, NTH, 0, s>>>(a); if (!no_others) k_sweep<32, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a); }
            }
            if (hint == 1) return info;
        }
    }
    if (!hellinger2) {
        if ((a.sd_fast == 1 || a.sd_fast == 2) && unit_weights && small && fmode == F_KEY && !a.wf_index && cmax <= 32) launch_sweep_inc(s, a.sd_fast, cmax, a);
        else launch_sweep_f<MODE_GEN, false>(s, cmax, grid, fmode, a);
    } else if (unit_weights) {
        if (small) launch_sweep_f<MODE_H2U, true>(s, cmax, grid, fmode, a);
        else launch_sweep_f<MODE_H2U, false>(s, cmax, grid, fmode, a);
    } else {
        if (small) launch_sweep_f<MODE_H2W, true>(s, cmax, grid, fmode, a);
        else launch_sweep_f<MODE_H2W, false>(s, cmax, grid, fmode, a);
    }
    return info & 1;
}

void launch_sweep_companion(hipStream_t s, const Tuning& t, int n_categories, int rule, const SweepArgs& a_in) {
    if (a_in.n_pairs <= 0) return;
    SweepArgs a = a_in;
    a.duo_enabled = 1;
    a.forced = 1;
    a.small_rule = rule;
    a.c8_rule = 2;
    a.second_rule = 0;
    a.gen_tab = 0;
    const int64_t blocks = (a.n_pairs + kSweepWaves - 1) / kSweepWaves;
    const unsigned bgrid = (unsigned)std::min<int64_t>(blocks, LCHD_COMPANION_GRID);
    constexpr int NTH = 64 * kSweepWaves;
    const int cmax = std::max(n_categories, t.force_cmax);
    if (cmax <= 8) k_sweep<8, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
    else if (cmax <= 12) k_sweep<12, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
    else if (cmax <= 16) k_sweep<16, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
    else if (cmax <= 20) k_sweep<20, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
    else if (cmax <= 24) k_sweep<24, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
    else if (cmax <= 28) k_sweep<28, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
    else k_sweep<32, MODE_H2U, F_KEY, true, true><<<bgrid, NTH, 0, s>>>(a);
}

// Kernels that may be launched with more than 64 KB of dynamic LDS need the limit raised per DEVICE: lchd_ctx_create calls this
// with the context's device current (a process-wide "done once" flag would leave a second device without the attribute).
void init_device_kernels() {
    auto raise = [](const void* fn, int bytes) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, false>), 16384 * 9);
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, true>), 16384 * 9);
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, false, uint16_t>), 8192 * 10);
    raise(reinterpret_cast<const void*>(&k_env_cells<1024, true, uint16_t>), 8192 * 10);
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, true>), (int)((kRowBucketsHuge + 1) * sizeof(uint32_t) + 16));
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, false>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, true, uint16_t>), (int)((kRowBucketsBig + 1) * sizeof(uint32_t) + 16));
    raise(reinterpret_cast<const void*>(&k_env_rows<1024, false, uint16_t>), 8192 * 10 + 16 + (kRowBucketsSmall + 1) * 4);
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 16, 1>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 10, 1>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
#ifdef LCHD_ROWS_NT512
    raise(reinterpret_cast<const void*>(&k_env_rows2<512, 20, 1>), 16384 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
#endif
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 12, 2>), (int)kRowSegLds);
    raise(reinterpret_cast<const void*>(&k_env_rows2<1024, 8, 1>), 8192 * 9 + 16 + (kRowBucketsSmall + 1) * 4);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_KEY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_ANY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_KEY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_ANY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_KEY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_ANY, 1>), 256 * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_KEY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_ANY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_KEY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_ANY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_KEY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_ANY, 1, false, true>), 256 * 512);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_KEY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_GEN, F_ANY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_KEY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2U, F_ANY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_KEY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_sweep_wide<MODE_H2W, F_ANY, 1, true>), kWideCategories * 256);
    raise(reinterpret_cast<const void*>(&k_prologue_fused), kStructCellsMax * 4 + kStructAtomsMax * 4);  // + ~3 KB static: above 64 KB in total
    raise(reinterpret_cast<const void*>(&k_cells_struct2<1024>), kStructCellsMax * 4 + kStructAtomsMax * 4);
    raise(reinterpret_cast<const void*>(&k_cells_struct2<LCHD_STRUCT_NT>), kStructCellsMax * 4 + kStructAtomsMax * 4);
    (void)hipGetLastError();
}

// sum over pairs of n_A + n_B (algorithmic-bytes accounting for bench.py; not part of the scoring path)
__global__ void k_env_points(SweepArgs args, unsigned long long* out) {
    unsigned long long local = 0;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < args.n_pairs; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t ea = p, eb = p;
        if (args.anchors) {
            const int64_t ia_ = args.anchors[2 * p], ib_ = args.anchors[2 * p + 1];
            if (ia_ < 0 || ib_ < 0 || ia_ >= args.n_slot_a || ib_ >= args.n_slot_b) continue;
            ea = args.slot_a[ia_];
            eb = args.slot_b ? args.slot_b[ib_] : p;
        }
        local += (unsigned long long)max(args.env_a.len[ea], 0) + (unsigned long long)max(args.env_b.len[eb], 0);
    }
    for (int m = 32; m > 0; m >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)local, m), hi = __shfl_xor((uint32_t)(local >> 32), m);
        local += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(out, local);
}
void launch_env_points(hipStream_t s, const SweepArgs& a, unsigned long long* out) {
    (void)hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
    if (a.n_pairs > 0) k_env_points<<<1024, 256, 0, s>>>(a, out);
}

}  // namespace lchd

namespace lchd {
// ------------------------------------------------------------------------------------------------
// Multi-GPU sharding of an anchor-pair list (one process per GPU, every rank holds the whole list and both structures).
// A rank that scores a contiguous slice of RANDOM pairs builds almost every environment of both structures itself; pairs
// binned by their side-A anchor make every rank build ~1/world of side A's environments.  The rule is a pure function of
// the list, so every rank computes the same partition without talking to the others:
//   bin(p)  = floor(a_p * kShardBins / n_atoms_a)                      (a_p = side-A anchor index, clamped into range)
//   rank(b) = min(world - 1, floor(#pairs in bins < b * world / P))    (the rank in which the bin's first pair falls)
// k_shard_plan: histogram of the bins (LDS-private per workgroup), the last workgroup turns it into rank(b) and the
// per-rank pair counts (also stored into host-mapped memory) and zeroes the histogram and the selection cursor again;
// k_shard_select: this rank's pairs, compacted (order = workgroup arrival, the original positions travel with them);
// k_unshard_scores: on the gathering rank, score k of rank r goes to its pair's original position.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int shard_bin(int64_t a, int64_t n_atoms) {
    a = a < 0 ? 0 : (a >= n_atoms ? n_atoms - 1 : a);
    return (int)((a * kShardBins) / n_atoms);
}
// The rule (loco_hd_amd/dist.py: shard_rule; lchd_capi.hip: shard_rule_host -- the same arithmetic in all three places):
//   key side   0: bins of the side-A anchor; if that partition is unbalanced (a rank would hold more than 1.25 P / world + 1
//              pairs: a list with one reference anchor against thousands, python_codes/kras_scan.py:46-52) 1: bins of the side-B
//              anchor; if that one is unbalanced too (or n_atoms_b is not given) 2: contiguous slices of the pair list
//   rank(bin) = min(world - 1, floor(#pairs in lower bins * world / P));   key side 2: rank(pair p) = floor(p * world / P)
__global__ __launch_bounds__(1024) void k_shard_plan(const int64_t* __restrict__ anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                                                      int world, ShardState* st, int64_t* counts_host) {
    __shared__ uint32_t ha[kShardBins], hb[kShardBins];
    __shared__ bool last_s;
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long cnt_s[kShardMaxWorld];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ha[tid] = 0u;  // kShardBins == blockDim.x == 1024
    hb[tid] = 0u;
    __syncthreads();
    const bool with_b = n_atoms_b > 0;
    for (int64_t p = blockIdx.x * 1024ll + tid; p < n_pairs; p += (int64_t)gridDim.x * 1024) {
        const longlong2 ab = reinterpret_cast<const longlong2*>(anchors)[p];
        atomicAdd(&ha[shard_bin(ab.x, n_atoms_a)], 1u);
        if (with_b) atomicAdd(&hb[shard_bin(ab.y, n_atoms_b)], 1u);
    }
    __syncthreads();
    {   // returning form, and the value is consumed: the atomics have been PERFORMED when the wave passes this point
        uint32_t r = 0;
        if (ha[tid]) r = atomicAdd(&st->hist[tid], ha[tid]);
        if (hb[tid]) r += atomicAdd(&st->hist_b[tid], hb[tid]);
        asm volatile("" ::"v"(r));
    }
    __syncthreads();
    if (tid == 0) last_s = last_workgroup_done(&st->done, 0ull, 0u);  // (the workgroup's histogram atomics completed before the barrier)
    __syncthreads();
    if (!last_s) return;
    // the last workgroup: for a key side, the exclusive scan of its 1024 bins (one per thread), rank(b) and the per-rank counts;
    // side A first, side B if A's partition is unbalanced, contiguous slices if B's is too
    const uint32_t va = __hip_atomic_load(&st->hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t vb = __hip_atomic_load(&st->hist_b[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto plan_side = [&](uint32_t v) -> bool {  // true: balanced (no rank holds more than 1.25 P / world + 1 pairs)
        const uint32_t incl = wave_incl_scan_u32(v);
        __syncthreads();
        if (lane == 63) wsum[wave] = incl;
        if (tid < kShardMaxWorld) cnt_s[tid] = 0ull;
        __syncthreads();
        unsigned long long pre = incl - v;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        int r = (int)((pre * (unsigned long long)world) / (unsigned long long)n_pairs);
        r = r < world - 1 ? r : world - 1;
        st->rank_of_bin[tid] = (uint16_t)r;
        if (v) atomicAdd(&cnt_s[r], (unsigned long long)v);
        __syncthreads();
        unsigned long long mx = 0;
        for (int w = 0; w < world; ++w) mx = cnt_s[w] > mx ? cnt_s[w] : mx;
        return mx * 4ull * (unsigned long long)world <= 5ull * (unsigned long long)n_pairs + 4ull * (unsigned long long)world;
    };
    int mode = 0;
    if (!plan_side(va)) mode = (with_b && plan_side(vb)) ? 1 : 2;
    st->hist[tid] = 0u;
    st->hist_b[tid] = 0u;
    __syncthreads();
    if (tid < world) {
        long long c = (long long)cnt_s[tid];
        if (mode == 2) {  // pairs p with floor(p * world / P) == tid: [ceil(tid P / world), ceil((tid + 1) P / world))
            const long long lo = ((long long)tid * n_pairs + world - 1) / world, hi = ((long long)(tid + 1) * n_pairs + world - 1) / world;
            c = hi - lo;
        }
        st->counts[tid] = c;
        counts_host[tid] = c;
    }
    if (tid == 0) { st->cursor = 0ull; st->mode = mode; counts_host[kShardMaxWorld] = mode; }
}
__global__ __launch_bounds__(256) void k_shard_select(const int64_t* __restrict__ anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b,
                                                      int rank, int world, ShardState* st, int64_t* __restrict__ sel_anchors,
                                                      int64_t* __restrict__ sel_index) {
    __shared__ uint32_t wcnt[4];
    __shared__ unsigned long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = 8;  // pairs per thread and round: one cursor atomic per 2048 pairs
    const uint16_t* __restrict__ rob = st->rank_of_bin;
    const int mode = st->mode;
    for (int64_t p0 = (int64_t)blockIdx.x * 256 * PER; p0 < n_pairs; p0 += (int64_t)gridDim.x * 256 * PER) {
        longlong2 ab[PER];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t p = p0 + (int64_t)tid * PER + u;  // a thread owns PER consecutive pairs: original order inside a round
            if (p < n_pairs) {
                ab[u] = reinterpret_cast<const longlong2*>(anchors)[p];
                int r;
                if (mode == 2) r = (int)((p * world) / n_pairs);
                else r = rob[mode == 1 ? shard_bin(ab[u].y, n_atoms_b) : shard_bin(ab[u].x, n_atoms_a)];
                if (r == rank) mine |= 1u << u;
            }
        }
        const uint32_t c = (uint32_t)__popc(mine);
        const uint32_t incl = wave_incl_scan_u32(c);
        if (lane == 63) wcnt[wave] = incl;
        __syncthreads();
        if (tid == 0) base_s = atomicAdd(&st->cursor, (unsigned long long)(wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]));
        __syncthreads();
        unsigned long long o = base_s + incl - c;
        for (int w = 0; w < wave; ++w) o += wcnt[w];
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if ((mine >> u) & 1u) {
                reinterpret_cast<longlong2*>(sel_anchors)[o] = ab[u];
                sel_index[o] = p0 + (int64_t)tid * PER + u;
                ++o;
            }
        __syncthreads();
    }
}
__global__ void k_unshard_scores(const double* __restrict__ gathered, ShardCounts counts, int world, int64_t stride, double* __restrict__ out,
                                 int64_t n_pairs, uint32_t* bad) {
    // gathered: [world][2][stride] -- scores, then the original pair positions as int64 bit patterns
    for (int r = 0; r < world; ++r) {
        const double* sc = gathered + (int64_t)r * 2 * stride;
        const int64_t* ix = reinterpret_cast<const int64_t*>(sc + stride);
        for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < counts.n[r]; k += (int64_t)gridDim.x * blockDim.x) {
            const int64_t p = ix[k];
            if (p < 0 || p >= n_pairs) { *bad = 1u; continue; }
            out[p] = sc[k];
        }
    }
}
void launch_shard_plan(hipStream_t s, const int64_t* anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int world, ShardState* st,
                       int64_t* counts_host) {
    const int64_t nb = (n_pairs + 8191) / 8192;
    k_shard_plan<<<(unsigned)(nb < 256 ? (nb > 0 ? nb : 1) : 256), 1024, 0, s>>>(anchors, n_pairs, n_atoms_a, n_atoms_b, world, st, counts_host);
}
void launch_shard_select(hipStream_t s, const int64_t* anchors, int64_t n_pairs, int64_t n_atoms_a, int64_t n_atoms_b, int rank, int world,
                         ShardState* st, int64_t* sel_anchors, int64_t* sel_index) {
    const int64_t nb = (n_pairs + 2047) / 2048;
    k_shard_select<<<(unsigned)(nb < 1024 ? (nb > 0 ? nb : 1) : 1024), 256, 0, s>>>(anchors, n_pairs, n_atoms_a, n_atoms_b, rank, world, st, sel_anchors,
                                                                                    sel_index);
}
// ---- weight-function dictionaries: one set of F keys per function ------------------------------------------------------------
// The grouped environment kernel left DISTANCE keys in set 0 of the store (EnvStore::set_stride elements per set); one wavefront per
// environment writes F_w(distance) into set 1 + w for every function w of the dictionary (src/locohd.rs:230-283: each pair names its
// function; the sweep then reads the set of the pair's function and never evaluates a CDF).  A running maximum keeps every set sorted
// whatever the last bits of the floating-point CDF do (keys_to_cdf_lds does the same; equal F values are zero-width intervals).
__global__ __launch_bounds__(256) void k_env_key_sets(const DevConfig* __restrict__ cfgp, EnvStore ea, EnvStore eb, int n_sets, const DeviceStatus* st) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t nu_a = st->n_unique[0], nu_b = st->n_unique[1];
    for (int64_t e = (int64_t)blockIdx.x * 4 + wave; e < nu_a + nu_b; e += (int64_t)gridDim.x * 4) {
        const bool on_b = e >= nu_a;
        const int64_t slot = on_b ? e - nu_a : e;
        uint64_t* const base = on_b ? eb.key : ea.key;
        const int64_t stride = on_b ? eb.stride : ea.stride, set_stride = on_b ? eb.set_stride : ea.set_stride;
        const int n = (on_b ? eb.len : ea.len)[slot];
        const uint64_t* __restrict__ src = base + slot * stride;
        for (int w = 0; w < n_sets; ++w) {
            const WfEntry wf = cfgp->wf[w];
            const double* __restrict__ prm = cfgp->wf_params + wf.offset;
            const double winv = cfgp->wf_inv[w];
            uint64_t* __restrict__ dst = base + (int64_t)(1 + w) * set_stride + slot * stride;
            uint64_t carry = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                uint64_t f = i < n ? d2u(cdf_lean(wf.kind, prm, wf.n_params, winv, u2d(src[i])) + 0.0) : 0ull;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint64_t t = shfl_up_u64(f, d);
                    if (lane >= d) f = t > f ? t : f;
                }
                f = carry > f ? carry : f;
                if (i < n) dst[i] = f;
                carry = shfl_u64(f, 63);
            }
        }
    }
}
void launch_env_key_sets(hipStream_t s, const DevConfig* cfg, const EnvStore& ea, const EnvStore& eb, int n_sets, int64_t max_envs, const DeviceStatus* st) {
    if (n_sets <= 0 || max_envs <= 0) return;
    const int64_t nb = (max_envs + 3) / 4;
    k_env_key_sets<<<(unsigned)(nb < 8192 ? nb : 8192), 256, 0, s>>>(cfg, ea, eb, n_sets, st);
}

// ---- the pairs of a finished pass that touch an overflowed environment (EnvSide::ovf_list) ----------------------------------
// k_mark_overflow: overflow lists -> bit sets over the sides' slots (zeroed by the host).  k_select_overflow<false>: every
// wavefront counts the marked pairs of its contiguous share of the list; k_scan_overflow: exclusive scan of the (at most
// kOverflowWaves) counts; k_select_overflow<true>: the same walk again, now writing pair index, anchor pair and weight-function
// index in list order (ordered compaction: the second pass sees the pairs in the caller's order).
__global__ void k_mark_overflow(const uint32_t* __restrict__ list_a, uint32_t na, const uint32_t* __restrict__ list_b, uint32_t nb,
                                uint32_t* bits_a, uint32_t* bits_b) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < na + nb; k += gridDim.x * blockDim.x) {
        const uint32_t e = k < na ? list_a[k] : list_b[k - na];
        atomicOr(&(k < na ? bits_a : bits_b)[e >> 5], 1u << (e & 31));
    }
}
template <bool WRITE>
__global__ __launch_bounds__(64) void k_select_overflow(OverflowSelect a) {
    const int lane = threadIdx.x;
    const int64_t share = (a.n_pairs + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * share, p1 = p0 + share < a.n_pairs ? p0 + share : a.n_pairs;
    unsigned long long at = WRITE ? a.wave_count[blockIdx.x] : 0ull;  // (after the scan: the share's first position in the selection)
    for (int64_t q = p0; q < p1; q += 64) {
        const int64_t p = q + lane;
        bool big = false;
        int64_t ia = 0, ib = 0;
        if (p < p1) {
            ia = a.anchors[2 * p];
            ib = a.anchors[2 * p + 1];
            const uint32_t sa = a.slot_a[ia], sb = a.slot_b[ib];
            big = ((a.bits_a[sa >> 5] >> (sa & 31)) & 1u) | ((a.bits_b[sb >> 5] >> (sb & 31)) & 1u);
        }
        const unsigned long long m = __ballot(big);
        if constexpr (WRITE) {
            if (big) {
                const unsigned long long k = at + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                a.sel_index[k] = p;
                a.sel_anchors[2 * k] = ia;
                a.sel_anchors[2 * k + 1] = ib;
                if (a.wf) a.sel_wf[k] = a.wf[p];
            }
        }
        at += (unsigned long long)__popcll(m);
    }
    if constexpr (!WRITE)
        if (lane == 0) a.wave_count[blockIdx.x] = at;
}
__global__ __launch_bounds__(1024) void k_scan_overflow(unsigned long long* wave_count, int n, unsigned long long* total) {
    __shared__ unsigned long long part[1024];
    const int tid = threadIdx.x;
    constexpr int PER = kOverflowWaves / 1024;
    unsigned long long v[PER], sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) { v[u] = tid * PER + u < n ? wave_count[tid * PER + u] : 0ull; sum += v[u]; }
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long add = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    unsigned long long pre = part[tid] - sum;
#pragma unroll
    for (int u = 0; u < PER; ++u) { if (tid * PER + u < n) wave_count[tid * PER + u] = pre; pre += v[u]; }
    if (tid == 1023) *total = part[1023];
}
void launch_mark_overflow(hipStream_t s, const uint32_t* list_a, uint32_t na, const uint32_t* list_b, uint32_t nb, uint32_t* bits_a, uint32_t* bits_b) {
    if (na + nb == 0) return;
    const uint32_t nbk = (na + nb + 255) / 256;
    k_mark_overflow<<<nbk < 1024 ? nbk : 1024, 256, 0, s>>>(list_a, na, list_b, nb, bits_a, bits_b);
}
int overflow_select_waves(int64_t n_pairs) {
    const int64_t w = (n_pairs + 255) / 256;
    return (int)(w < kOverflowWaves ? (w > 0 ? w : 1) : kOverflowWaves);
}
void launch_count_overflow(hipStream_t s, const OverflowSelect& a, unsigned long long* total) {
    const int w = overflow_select_waves(a.n_pairs);
    k_select_overflow<false><<<w, 64, 0, s>>>(a);
    k_scan_overflow<<<1, 1024, 0, s>>>(a.wave_count, w, total);
}
void launch_write_overflow(hipStream_t s, const OverflowSelect& a) {
    k_select_overflow<true><<<overflow_select_waves(a.n_pairs), 64, 0, s>>>(a);
}

// one share's scores back to their positions in the caller's order (out: any device-visible memory, e.g. a host-mapped block)
__global__ void k_scatter_scores(const double* __restrict__ scores, const int64_t* __restrict__ index, int64_t n, double* __restrict__ out) {
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) out[index[k]] = scores[k];
}
void launch_scatter_scores(hipStream_t s, const double* scores, const int64_t* index, int64_t n, double* out) {
    if (n <= 0) return;
    const int64_t nb = (n + 255) / 256;
    k_scatter_scores<<<(unsigned)(nb < 4096 ? nb : 4096), 256, 0, s>>>(scores, index, n, out);
}
void launch_unshard_scores(hipStream_t s, const double* gathered, const ShardCounts& counts, int world, int64_t stride, double* out,
                           int64_t n_pairs, uint32_t* bad) {
    k_unshard_scores<<<1024, 256, 0, s>>>(gathered, counts, world, stride, out, n_pairs, bad);
}
}  // namespace lchd

namespace lchd {
__global__ void k_fill_sqrt_tables(double* sqrt_tab, double* rsqrt_tab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 65536) {
        const double r = sqrt((double)k);
        sqrt_tab[k] = r;
        rsqrt_tab[k] = 1.0 / r;
    }
}
#ifdef LCHD_SWEEP_STAMPS
}  // namespace lchd
extern "C" int lchd_debug_env_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lchd::g_env_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lchd::g_env_stamps), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int lchd_debug_sweep_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lchd::g_sweep_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lchd::g_sweep_stamps), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
namespace lchd {
#endif
__global__ void k_fill_pow_tables(double* tab, double einv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 65536) {
        tab[k] = pow((double)k, einv);
        tab[65536 + k] = pow((double)k, -einv);
    }
}
void launch_fill_pow_tables(hipStream_t s, double* tab, double exponent) { k_fill_pow_tables<<<256, 256, 0, s>>>(tab, 1.0 / exponent); }
void launch_fill_sqrt_tables(hipStream_t s, double* sqrt_tab, double* rsqrt_tab) {
    k_fill_sqrt_tables<<<256, 256, 0, s>>>(sqrt_tab, rsqrt_tab);
}
}  // namespace lchd

namespace lchd {
__global__ void k_frames_labels(const uint8_t* tcat, const int32_t* ttag, int64_t n_tmpl, int64_t total, uint8_t* cat, int32_t* tag,
                                int32_t* sid) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / n_tmpl, k = i - f * n_tmpl;
        cat[i] = tcat[k];
        tag[i] = ttag[k];
        sid[i] = (int32_t)f;
    }
}
void launch_frames_labels(hipStream_t s, const uint8_t* tcat, const int32_t* ttag, int64_t n_tmpl, int32_t n_frames, uint8_t* cat,
                          int32_t* tag, int32_t* sid) {
    k_frames_labels<<<2048, 256, 0, s>>>(tcat, ttag, n_tmpl, n_tmpl * n_frames, cat, tag, sid);
}

// order-preserving map double -> u64 (so that atomicMin / atomicMax on integers order like the doubles)
__device__ __forceinline__ unsigned long long ordered_key(double d) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__global__ void k_frames_unpack(const double* __restrict__ raw, int64_t n, double* __restrict__ x, double* __restrict__ y,
                                double* __restrict__ z, unsigned long long* bbox7) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double vx = raw[3 * i], vy = raw[3 * i + 1], vz = raw[3 * i + 2];
        x[i] = vx; y[i] = vy; z[i] = vz;
        bad = bad || !(fabs(vx) < INFINITY) || !(fabs(vy) < INFINITY) || !(fabs(vz) < INFINITY);
        mn[0] = fmin(mn[0], vx); mn[1] = fmin(mn[1], vy); mn[2] = fmin(mn[2], vz);
        mx[0] = fmax(mx[0], vx); mx[1] = fmax(mx[1], vy); mx[2] = fmax(mx[2], vz);
    }
    for (int m = 32; m > 0; m >>= 1)
        for (int k = 0; k < 3; ++k) { mn[k] = fmin(mn[k], shfl_xor_f64(mn[k], m)); mx[k] = fmax(mx[k], shfl_xor_f64(mx[k], m)); }
    const unsigned long long anybad = __ballot(bad);
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; ++k) { atomicMin(&bbox7[k], ordered_key(mn[k])); atomicMax(&bbox7[3 + k], ordered_key(mx[k])); }
        if (anybad) atomicOr(&bbox7[6], 1ull);
    }
}
void launch_frames_unpack(hipStream_t s, const double* raw, int64_t n_atoms, double* x, double* y, double* z, unsigned long long* bbox7) {
    static const unsigned long long init[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    (void)hipMemcpyAsync(bbox7, init, sizeof init, hipMemcpyHostToDevice, s);
    const int64_t nb = (n_atoms + 255) / 256;
    k_frames_unpack<<<(unsigned)(nb < 1024 ? nb : 1024), 256, 0, s>>>(raw, n_atoms, x, y, z, bbox7);
}

// Frames given as SOURCE atoms (float32, the precision of Bio.PDB Atom.coord / MDAnalysis Timestep.positions): primitive
// atom p of every frame is the centroid of source atoms src_idx[src_start[p] .. src_start[p+1]) -- what
// PrimitiveAssigner.assign_primitive_structure computes per frame on the host with np.mean(atom_coords, axis=0)
// (/root/reference/loco_hd/atom_converter_utils.py:106-126, python_codes/trajectory_analyzer.py:55-72).  Same arithmetic
// as that call: float32 accumulator starting from +0, members added in list order, one IEEE float32 division
// by the member count; the result is widened to f64 exactly like PrimitiveAtom.coordinates.
//
// HBM-bound by construction: the host cuts the primitive atoms into tiles whose members span at most kCentroidSpan
// consecutive source atoms (typing schemes walk residues in order, so the CSR map is local); a workgroup streams one
// (frame, tile) slice of the source coordinates into LDS with 16-byte coalesced loads, gathers the members from LDS
// (stride 3 floats: conflict-free) and writes x/y/z coalesced.  Every source atom is read once per frame, every
// primitive atom written once: 12 B x n_src + 24 B x n_prim per frame.  A tile whose span does not fit (lo == hi == -1)
// gathers straight from global memory.  The bounding box is reduced per workgroup before it touches the 7 global words.
constexpr int kCentroidSpan = 4096;  // source atoms per tile: 48 KB of LDS
constexpr int kBboxParts = 4096;     // capacity of a frames buffer's per-workgroup bounding-box partials
__global__ __launch_bounds__(256) void k_frames_centroids(const float* __restrict__ raw, int64_t n_src, const int32_t* __restrict__ src_start,
                                                          const int32_t* __restrict__ src_idx, const int4* __restrict__ tiles, int n_tiles,
                                                          int64_t n_prim, int64_t n_items, double* __restrict__ x, double* __restrict__ y,
                                                          double* __restrict__ z, unsigned long long* __restrict__ bbox_part) {
    __shared__ __attribute__((aligned(16))) float s_xyz[kCentroidSpan * 3 + 8];
    __shared__ double s_red[4][6];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int64_t w = blockIdx.x; w < n_items; w += gridDim.x) {
        const int64_t f = w / n_tiles;
        const int4 t = tiles[(int)(w - f * n_tiles)];  // {first primitive, end primitive, first source atom, end source atom}
        const float* fr = raw + 3 * f * n_src;
        const bool staged = t.z >= 0;
        int head = 0;
        if (staged) {
            // floats [g0, g1) of the buffer = source atoms [lo, hi) of this frame.  LDS origin `base` = the 16-byte aligned
            // ADDRESS at or below g0 (the caller's pointer need not be 16-byte aligned; base can be < 0 only for the very
            // first floats of the buffer, which are then copied one by one).
            const int64_t g0 = 3 * f * n_src + 3 * (int64_t)t.z, g1 = 3 * f * n_src + 3 * (int64_t)t.w;
            const int m = (int)((reinterpret_cast<uintptr_t>(raw) >> 2) & 3);
            const int64_t base = ((g0 + m) & ~(int64_t)3) - m;
            const int64_t v0 = base < 0 ? base + 4 : base;  // first aligned float4 inside the buffer
            head = (int)(g0 - base);
            const int n4 = g1 > v0 ? (int)((g1 - v0) >> 2) : 0;
            const float4* src4 = reinterpret_cast<const float4*>(raw + v0);
            float4* dst4 = reinterpret_cast<float4*>(s_xyz) + ((v0 - base) >> 2);
            for (int k = tid; k < n4; k += 256) dst4[k] = src4[k];
            for (int64_t k = g0 + tid; k < v0; k += 256) s_xyz[k - base] = raw[k];
            for (int64_t k = v0 + 4 * (int64_t)n4 + tid; k < g1; k += 256) s_xyz[k - base] = raw[k];
            __syncthreads();
        }
        for (int p = t.x + tid; p < t.y; p += 256) {
            const int b = src_start[p], e = src_start[p + 1];
            float sx = 0.0f, sy = 0.0f, sz = 0.0f;  // NumPy's add.reduce starts from +0: a lone -0.0 member comes out as +0.0
            if (staged) {
                for (int k = b; k < e; ++k) {
                    const float* a = s_xyz + head + 3 * (src_idx[k] - t.z);
                    sx += a[0]; sy += a[1]; sz += a[2];
                }
            } else {
                for (int k = b; k < e; ++k) {
                    const float* a = fr + 3 * (int64_t)src_idx[k];
                    sx += a[0]; sy += a[1]; sz += a[2];
                }
            }
            const float cnt = (float)(e - b);
            const double vx = (double)__fdiv_rn(sx, cnt), vy = (double)__fdiv_rn(sy, cnt), vz = (double)__fdiv_rn(sz, cnt);
            const int64_t o = f * n_prim + p;
            x[o] = vx; y[o] = vy; z[o] = vz;
            bad = bad || !(fabs(vx) < INFINITY) || !(fabs(vy) < INFINITY) || !(fabs(vz) < INFINITY);
            mn[0] = fmin(mn[0], vx); mn[1] = fmin(mn[1], vy); mn[2] = fmin(mn[2], vz);
            mx[0] = fmax(mx[0], vx); mx[1] = fmax(mx[1], vy); mx[2] = fmax(mx[2], vz);
        }
        if (staged) __syncthreads();  // the tile is consumed before the next one overwrites it
    }
    for (int m = 32; m > 0; m >>= 1)
        for (int k = 0; k < 3; ++k) { mn[k] = fmin(mn[k], shfl_xor_f64(mn[k], m)); mx[k] = fmax(mx[k], shfl_xor_f64(mx[k], m)); }
    const unsigned long long anybad = __ballot(bad);
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if ((tid & 63) == 0) {
        for (int k = 0; k < 3; ++k) { s_red[tid >> 6][k] = mn[k]; s_red[tid >> 6][3 + k] = mx[k]; }
        if (anybad) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    // per-workgroup partial result, no atomics: k_bbox_finish folds the partials into the 7 words the host reads
    if (tid < 6) {
        double v = s_red[0][tid];
        for (int wv = 1; wv < 4; ++wv) v = tid < 3 ? fmin(v, s_red[wv][tid]) : fmax(v, s_red[wv][tid]);
        bbox_part[(size_t)blockIdx.x * 7 + tid] = ordered_key(v);
    }
    if (tid == 6) bbox_part[(size_t)blockIdx.x * 7 + 6] = s_bad ? 1ull : 0ull;
}
__global__ __launch_bounds__(256) void k_bbox_finish(const unsigned long long* __restrict__ part, int n_parts, unsigned long long* bbox7) {
    __shared__ unsigned long long red[4][7];
    const int tid = threadIdx.x;
    unsigned long long v[7] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull, 0ull};
    for (int b = tid; b < n_parts; b += 256) {
        for (int k = 0; k < 3; ++k) { v[k] = min(v[k], part[(size_t)b * 7 + k]); v[3 + k] = max(v[3 + k], part[(size_t)b * 7 + 3 + k]); }
        v[6] |= part[(size_t)b * 7 + 6];
    }
    for (int m = 32; m > 0; m >>= 1)
        for (int k = 0; k < 7; ++k) {
            const unsigned long long o = shfl_u64(v[k], (tid & 63) ^ m);
            v[k] = k < 3 ? min(v[k], o) : (k < 6 ? max(v[k], o) : (v[k] | o));
        }
    if ((tid & 63) == 0) for (int k = 0; k < 7; ++k) red[tid >> 6][k] = v[k];
    __syncthreads();
    if (tid < 7) {
        unsigned long long r = red[0][tid];
        for (int w = 1; w < 4; ++w) r = tid < 3 ? min(r, red[w][tid]) : (tid < 6 ? max(r, red[w][tid]) : (r | red[w][tid]));
        bbox7[tid] = r;
    }
}
void launch_frames_centroids(hipStream_t s, const float* raw, int64_t n_src, const int32_t* src_start, const int32_t* src_idx,
                             const int32_t* tiles, int n_tiles, int64_t n_prim, int32_t n_frames, double* x, double* y, double* z,
                             unsigned long long* bbox7, unsigned long long* bbox_part) {
    const int64_t items = (int64_t)n_tiles * n_frames;
    const int grid = (int)(items < kBboxParts ? items : kBboxParts);  // more workgroups than fit at once: the dispatcher balances them
    k_frames_centroids<<<grid, 256, 0, s>>>(raw, n_src, src_start, src_idx, reinterpret_cast<const int4*>(tiles), n_tiles, n_prim, items, x, y,
                                            z, bbox_part);
    k_bbox_finish<<<1, 256, 0, s>>>(bbox_part, grid, bbox7);
}
int centroid_tile_span() { return kCentroidSpan; }
int bbox_parts_capacity() { return kBboxParts; }
}  // namespace lchd

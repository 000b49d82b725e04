// lchd_prologue.hip -- K0 of a from_primitives pass: the cell lists of both structures (replaces KdTree::build_by_ordered_float,
// /root/reference/src/locohd.rs:504-510) and the de-duplication of the anchors, in as few launches as the sizes allow.
#include <algorithm>

#include "lchd_sweep_common.h"

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
    if (P.no_anchors) {  // (side B without de-duplication: k_pair_anchor_recs validates this side's anchor indices)
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
constexpr int64_t kDupSampleAbove = (int64_t)1 << 17;
__global__ void k_pair_anchor_recs(const int64_t* __restrict__ anchors, int64_t n_pairs, PrepSide pb, DeviceStatus* st) {
    const CloudView& c = pb.c;
    uint32_t dup = 0;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = anchors[2 * p + 1];
        if (i < 0 || i >= c.n) {  // src/locohd.rs:521: the reference's index panic (the pair record marks the pair unusable)
            atomicOr(&st->flags, ST_BAD_ANCHOR);  // (the one-workgroup prologue does not look at this side's anchors in this mode)
            i = 0;
        }
        // how many pairs share their side-B anchor with an earlier one?  The side's flag region (zeroed by the prologue, otherwise unused
        // in this mode) as a bit set, one returning atomic per pair: the count tells the host when this side has stopped being "used
        // once" (a pass per pair is then a waste).  (Plain loads and stores do not work: a small list's threads all load before any stores.)
        // Large lists: every 16th pair only (1.7 10^6 returning atomics cost 0.19 ms per C4 pass; the host scales the count).
        if (n_pairs <= kDupSampleAbove || (p & 15) == 0) {
            const uint32_t bit = 1u << (i & 31);
            if (atomicOr(reinterpret_cast<uint32_t*>(pb.flag8) + (i >> 5), bit) & bit) ++dup;
        }
        AnchorRec r;
        r.x = c.x[i]; r.y = c.y[i]; r.z = c.z[i];
        r.tag = (uint32_t)c.tag[i];
        r.apos = pb.pos_of[i];
        r.sid = c.sid ? c.sid[i] : 0;
        r.atom = (uint32_t)i;
        pb.uniq[p] = r;
    }
    for (int m = 32; m > 0; m >>= 1) dup += (uint32_t)__shfl_xor((int)dup, m);
    if ((threadIdx.x & 63) == 0 && dup) atomicAdd(&st->n_dup_b, n_pairs <= kDupSampleAbove ? dup : 16u * dup);  // (sampled: scaled to the list)
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
    if (!same && fa && fb && csa.n_struct == 1 && csb.n_struct == 1 && n_pairs <= kFusedPairsMax && a.c.n > 0 && b.c.n > 0) {
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

void init_prologue_kernels() {
    auto raise = [](const void* fn, int bytes) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    raise(reinterpret_cast<const void*>(&k_prologue_fused), kStructCellsMax * 4 + kStructAtomsMax * 4);  // + ~3 KB static: above 64 KB in total
    raise(reinterpret_cast<const void*>(&k_cells_struct2<1024>), kStructCellsMax * 4 + kStructAtomsMax * 4);
    raise(reinterpret_cast<const void*>(&k_cells_struct2<LCHD_STRUCT_NT>), kStructCellsMax * 4 + kStructAtomsMax * 4);
    (void)hipGetLastError();
}
static_assert(kPrepScanAtoms == 1 << 18, "k_prep_scatter: chunk of atom i = i >> 18");

}  // namespace lchd

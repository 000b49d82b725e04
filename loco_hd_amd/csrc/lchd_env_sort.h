// lchd_env_sort.h -- LDS sorts and the distance -> F(distance) key conversion shared by the environment kernels that own a whole
// workgroup (lchd_env_cells.hip, lchd_env_rows.hip).  Reference: utils::sort_together (/root/reference/src/locohd/utils.rs:25-39).
#pragma once
#include "lchd_kcommon.h"

namespace lchd {

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


}  // namespace lchd

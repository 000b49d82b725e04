// issue_rates.hip -- wall-clock cost of one wave-instruction per SIMD on gfx950, by instruction class, at 1 / 2 / 4 / 8 resident
// wavefronts per SIMD.  Every class is a group of FOUR independent streams (different destination registers), repeated 16 times
// per loop iteration, so neither a dependency chain nor the loop overhead (two scalar instructions per 64) shows.
//
//   hipcc -O3 --offload-arch=gfx950 profiles/ubench/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates > profiles/r04/issue_rates.json
//
// Output (JSON): per class and occupancy, ns per instruction per SIMD = kernel wall time / (instructions per wave x waves per SIMD),
// the same in shader cycles (s_memtime deltas / instructions per wave / waves per SIMD) and the clock the chip held
// (s_memtime / s_memrealtime x 100 MHz).  Used by profiles/isa_histogram.py to turn a kernel's instruction mix into a
// class-weighted issue ceiling (bench.py: roofline.valu_ceiling_frac).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define G4(a, b, c, d) asm volatile(a "\n" b "\n" c "\n" d : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), \
                                    "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(la) : "vcc", "s10", "s11", "s12", "s13", "memory")

// operand numbers: %0-%3 u32 a0..a3, %4-%7 u64 u0..u3, %8-%11 f64 d0..d3, %12-%15 f32 f0..f3, %16 LDS byte address (per lane)
template <int K>
__device__ __forceinline__ void body(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, unsigned long long& u0, unsigned long long& u1,
                                     unsigned long long& u2, unsigned long long& u3, double& d0, double& d1, double& d2, double& d3, float& f0,
                                     float& f1, float& f2, float& f3, unsigned la) {
    if constexpr (K == 0) { REP16(G4("v_add_u32 %0, %0, %1", "v_add_u32 %1, %1, %2", "v_add_u32 %2, %2, %3", "v_add_u32 %3, %3, %0");) }
    if constexpr (K == 1) { REP16(G4("v_and_b32 %0, %0, %1", "v_lshlrev_b32 %1, 3, %1", "v_xor_b32 %2, %2, %3", "v_lshrrev_b32 %3, 1, %3");) }
    if constexpr (K == 2) { REP16(G4("v_lshl_add_u32 %0, %0, 2, %1", "v_and_or_b32 %1, %1, %2, %3", "v_bfe_u32 %2, %2, 4, 8", "v_lshl_or_b32 %3, %3, 3, %0");) }
    if constexpr (K == 3) { REP16(G4("v_cndmask_b32_e32 %0, %0, %1, vcc", "v_cndmask_b32_e32 %1, %1, %2, vcc", "v_cndmask_b32_e32 %2, %2, %3, vcc", "v_cndmask_b32_e32 %3, %3, %0, vcc");) }
    if constexpr (K == 4) { REP16(G4("v_cndmask_b32_e64 %0, %0, %1, s[10:11]", "v_cndmask_b32_e64 %1, %1, %2, s[12:13]", "v_cndmask_b32_e64 %2, %2, %3, s[10:11]", "v_cndmask_b32_e64 %3, %3, %0, s[12:13]");) }
    if constexpr (K == 5) { REP16(G4("v_cmp_lt_u32 vcc, %0, %1", "v_cmp_lt_u32 s[10:11], %1, %2", "v_cmp_eq_u32 vcc, %2, %3", "v_cmp_gt_u32 s[12:13], %3, %0");) }
    if constexpr (K == 6) { REP16(G4("v_lshrrev_b64 %4, 4, %4", "v_lshlrev_b64 %5, 1, %5", "v_lshrrev_b64 %6, 3, %6", "v_lshlrev_b64 %7, 2, %7");) }
    if constexpr (K == 7) { REP16(G4("v_cmp_lt_u64 vcc, %4, %5", "v_cmp_lt_u64 s[10:11], %5, %6", "v_cmp_eq_u64 vcc, %6, %7", "v_cmp_le_u64 s[12:13], %7, %4");) }
    if constexpr (K == 8) { REP16(G4("v_lshl_add_u64 %4, %4, 0, %5", "v_lshl_add_u64 %5, %5, 0, %6", "v_lshl_add_u64 %6, %6, 0, %7", "v_lshl_add_u64 %7, %7, 0, %4");) }
    if constexpr (K == 9) { REP16(G4("v_add_co_u32 %0, vcc, %0, %1", "v_addc_co_u32 %1, vcc, %1, %2, vcc", "v_add_co_u32 %2, vcc, %2, %3", "v_addc_co_u32 %3, vcc, %3, %0, vcc");) }
    if constexpr (K == 10) { REP16(G4("v_add_f64 %8, %8, %9", "v_add_f64 %9, %9, %10", "v_add_f64 %10, %10, %11", "v_add_f64 %11, %11, %8");) }
    if constexpr (K == 11) { REP16(G4("v_mul_f64 %8, %8, %9", "v_mul_f64 %9, %9, %10", "v_mul_f64 %10, %10, %11", "v_mul_f64 %11, %11, %8");) }
    if constexpr (K == 12) { REP16(G4("v_fma_f64 %8, %8, %9, %10", "v_fma_f64 %9, %9, %10, %11", "v_fma_f64 %10, %10, %11, %8", "v_fma_f64 %11, %11, %8, %9");) }
    if constexpr (K == 13) { REP16(G4("v_rsq_f64 %8, %8", "v_rsq_f64 %9, %9", "v_rsq_f64 %10, %10", "v_rsq_f64 %11, %11");) }
    if constexpr (K == 14) { REP16(G4("v_sqrt_f64 %8, %8", "v_rcp_f64 %9, %9", "v_sqrt_f64 %10, %10", "v_rcp_f64 %11, %11");) }
    if constexpr (K == 15) { REP16(G4("v_add_f32 %12, %12, %13", "v_fma_f32 %13, %13, %14, %15", "v_mul_f32 %14, %14, %15", "v_fma_f32 %15, %15, %12, %13");) }
    if constexpr (K == 16) { REP16(G4("v_cvt_f64_u32 %8, %0", "v_cvt_f64_i32 %9, %1", "v_cvt_f64_u32 %10, %2", "v_cvt_f64_i32 %11, %3");) }
    if constexpr (K == 17) { REP16(G4("v_cvt_f32_f64 %12, %8", "v_cvt_i32_f64 %0, %9", "v_cvt_f32_f64 %14, %10", "v_cvt_u32_f64 %2, %11");) }
    if constexpr (K == 18) { REP16(G4("v_mov_b32 %0, %1", "v_mov_b32 %1, %2", "v_mov_b32 %2, %3", "v_mov_b32 %3, %0");) }
    if constexpr (K == 19) { REP16(G4("v_mov_b64 %4, %5", "v_mov_b64 %5, %6", "v_mov_b64 %6, %7", "v_mov_b64 %7, %4");) }
    if constexpr (K == 20) { REP16(G4("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %1, %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %2, %2, %3 row_shr:4 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %3, %3, %0 row_shr:8 row_mask:0xf bank_mask:0xf");) }
    if constexpr (K == 21) { REP16(G4("v_readlane_b32 s10, %0, 5", "v_readlane_b32 s11, %1, 9", "v_readfirstlane_b32 s12, %2", "v_readlane_b32 s13, %3, 63");) }
    if constexpr (K == 22) { REP16(G4("v_mul_lo_u32 %0, %0, %1", "v_mul_hi_u32 %1, %1, %2", "v_mul_lo_u32 %2, %2, %3", "v_mul_hi_u32 %3, %3, %0");) }
    if constexpr (K == 23) { REP16(G4("v_mad_u64_u32 %4, vcc, %0, %1, %4", "v_mad_u64_u32 %5, vcc, %1, %2, %5", "v_mad_u64_u32 %6, vcc, %2, %3, %6", "v_mad_u64_u32 %7, vcc, %3, %0, %7");) }
    if constexpr (K == 24) { REP16(G4("ds_read_b64 %4, %16", "ds_read_b64 %5, %16 offset:8", "ds_read_b64 %6, %16 offset:16", "ds_read_b64 %7, %16 offset:24\n s_waitcnt lgkmcnt(0)");) }
    if constexpr (K == 25) { REP16(G4("ds_read_u8 %0, %16", "ds_read_u8 %1, %16 offset:1", "ds_read_u8 %2, %16 offset:2", "ds_read_u8 %3, %16 offset:3\n s_waitcnt lgkmcnt(0)");) }
    if constexpr (K == 26) { REP16(G4("ds_read_b32 %0, %16", "ds_read_b32 %1, %16 offset:4", "ds_read_b32 %2, %16 offset:8", "ds_read_b32 %3, %16 offset:12\n s_waitcnt lgkmcnt(0)");) }
    if constexpr (K == 27) { REP16(G4("ds_write_b32 %16, %0", "ds_write_b32 %16, %1 offset:4", "ds_write_b8 %16, %2 offset:8", "ds_write_b64 %16, %4 offset:16\n s_waitcnt lgkmcnt(0)");) }
    if constexpr (K == 28) { REP16(G4("v_max_f64 %8, %8, %9", "v_min_f64 %9, %9, %10", "v_max_f64 %10, %10, %11", "v_min_f64 %11, %11, %8");) }
    if constexpr (K == 29) { REP16(G4("v_cmp_lt_f64 vcc, %8, %9", "v_cmp_lt_f64 s[10:11], %9, %10", "v_cmp_ge_f64 vcc, %10, %11", "v_cmp_eq_f64 s[12:13], %11, %8");) }
    if constexpr (K == 30) { REP16(G4("v_ldexp_f64 %8, %8, %0", "v_rndne_f64 %9, %9", "v_ldexp_f64 %10, %10, %2", "v_rndne_f64 %11, %11");) }
    if constexpr (K == 31) { REP16(G4("s_nop 0", "s_nop 0", "s_nop 0", "s_nop 0");) }
    // mixed patterns around v_cndmask_b32_e32 (a 64-bit select is two of them back to back behind a compare)
    if constexpr (K == 32) { REP16(G4("v_cndmask_b32_e32 %0, %0, %1, vcc", "v_cndmask_b32_e32 %1, %1, %2, vcc", "v_add_u32 %2, %2, %3", "v_add_u32 %3, %3, %0");) }
    if constexpr (K == 33) { REP16(G4("v_cmp_lt_u32 vcc, %0, %1", "v_cndmask_b32_e32 %1, %1, %2, vcc", "v_cndmask_b32_e32 %2, %2, %3, vcc", "v_add_u32 %3, %3, %0");) }
    if constexpr (K == 34) { REP16(G4("v_cndmask_b32_e32 %0, %0, %1, vcc", "v_add_u32 %1, %1, %2", "v_cndmask_b32_e32 %2, %2, %3, vcc", "v_add_u32 %3, %3, %0");) }
    if constexpr (K == 35) { REP16(G4("v_cmp_lt_u64 vcc, %4, %5", "v_cndmask_b32_e32 %0, %0, %1, vcc", "v_cndmask_b32_e32 %2, %2, %3, vcc", "v_lshrrev_b64 %6, 3, %6");) }
    if constexpr (K == 36) { REP16(G4("v_add_u32_e64 %0, %0, %1", "v_sub_u32_e64 %1, %1, %2", "v_add_u32_e64 %2, %2, %3", "v_sub_u32_e64 %3, %3, %0");) }
    if constexpr (K == 37) { REP16(G4("v_cmp_lt_u32_e32 vcc, %0, %1", "v_cmp_gt_u32_e32 vcc, %1, %2", "v_cmp_eq_u32_e32 vcc, %2, %3", "v_cmp_ne_u32_e32 vcc, %3, %0");) }
    if constexpr (K == 38) { REP16(G4("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0", "v_add_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1", "v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2", "v_add_u32_sdwa %3, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1");) }
    if constexpr (K == 39) { REP16(G4("v_cmp_lt_u32 s[10:11], %0, %1", "v_cndmask_b32_e64 %1, %1, %2, s[10:11]", "v_cndmask_b32_e64 %2, %2, %3, s[10:11]", "v_add_u32 %3, %3, %0");) }
}
constexpr int kClasses = 40;
static const char* kNames[kClasses] = {
    "b32_add", "b32_logic_shift", "b32_three_operand", "cndmask_e32_vcc", "cndmask_e64_sgpr", "cmp_u32", "b64_shift", "cmp_u64", "lshl_add_u64",
    "add_co_addc_pair_half", "f64_add", "f64_mul", "f64_fma", "f64_rsq", "f64_sqrt_rcp", "f32_alu", "cvt_to_f64", "cvt_from_f64", "mov_b32", "mov_b64",
    "dpp_add_u32", "readlane", "mul_u32", "mad_u64_u32", "ds_read_b64", "ds_read_u8", "ds_read_b32", "ds_write_mixed", "f64_minmax", "cmp_f64",
    "f64_ldexp_rndne", "s_nop", "mix_2cndmask_e32_2add", "mix_cmp_2cndmask_e32_add", "mix_cndmask_e32_add_alternating",
    "mix_cmp64_2cndmask_e32_shift64", "b32_add_vop3_encoding", "cmp_u32_e32_vcc", "b32_add_sdwa", "mix_cmp_2cndmask_e64_add"};

template <int K>
__global__ __launch_bounds__(256) void bench(unsigned long long* out, int iters, double* sink, const double* src) {
    __shared__ double lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) lds[i] = src[i] * 0;
    __syncthreads();
    double d0 = src[tid] + 1.0, d1 = src[tid + 1] + 1.0, d2 = src[tid + 2] + 1.0, d3 = src[tid + 3] + 1.0;
    unsigned long long u0 = (unsigned long long)tid * 0x9E3779B97F4A7C15ull, u1 = u0 ^ 0x1234567, u2 = u0 + 77, u3 = u1 * 3;
    unsigned a0 = tid, a1 = tid * 3 + 1, a2 = tid ^ 0x55, a3 = tid + 9;
    float f0 = tid * 0.5f + 1.0f, f1 = tid + 2.0f, f2 = 1.0001f, f3 = 0.9999f;
    const unsigned la = (unsigned)(size_t)lds + (unsigned)(tid & 63) * 32u;  // LDS byte address (low 32 bits of the generic pointer's offset)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) body<K>(a0, a1, a2, a3, u0, u1, u2, u3, d0, d1, d2, d3, f0, f1, f2, f3, la);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((tid & 63) == 0) { out[(blockIdx.x * 4 + (tid >> 6)) * 2] = t1 - t0; out[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = r1 - r0; }
    sink[blockIdx.x * 256 + tid] = d0 + d1 + d2 + d3 + (double)(u0 + u1 + u2 + u3) + (double)(a0 + a1 + a2 + a3) + (double)(f0 + f1 + f2 + f3) + lds[tid];
}

struct Row { std::string name; int w; double ns, cyc, ghz; };
static std::vector<Row> rows;

template <int K>
void run(int blocks_per_cu, unsigned long long* d_out, double* d_sink, double* d_src) {
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    bench<K><<<blocks, 256>>>(d_out, 10, d_sink, d_src);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    bench<K><<<blocks, 256>>>(d_out, iters, d_sink, d_src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 8);
    hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    double tick = 0, real = 0;
    for (size_t i = 0; i < h.size(); i += 2) { tick += (double)h[i]; real += (double)h[i + 1]; }
    tick /= h.size() / 2; real /= h.size() / 2;
    const double instr = (double)iters * 64;  // per wave
    rows.push_back({kNames[K], blocks_per_cu, ms * 1e6 / (instr * blocks_per_cu), tick / (instr * blocks_per_cu), tick / real * 0.1});
    hipEventDestroy(e0); hipEventDestroy(e1);
}
template <int K> void run_all(unsigned long long* o, double* s, double* src) {
    for (int w : {1, 2, 4, 8}) run<K>(w, o, s, src);
    if constexpr (K + 1 < kClasses) run_all<K + 1>(o, s, src);
}
int main() {
    unsigned long long* d_out; double *d_sink, *d_src;
    hipMalloc(&d_out, sizeof(unsigned long long) * 256 * 8 * 8);
    hipMalloc(&d_sink, sizeof(double) * 256 * 8 * 256);
    hipMalloc(&d_src, sizeof(double) * 4096);
    std::vector<double> h(4096); for (int i = 0; i < 4096; ++i) h[i] = 0.5 + i * 1e-3;
    hipMemcpy(d_src, h.data(), sizeof(double) * 4096, hipMemcpyHostToDevice);
    run_all<0>(d_out, d_sink, d_src);
    printf("{\n \"device\": \"gfx950 (MI355X)\", \"method\": \"4 independent streams x 16 per loop iteration, 2000 iterations, 256 x W workgroups of 256 threads (one wavefront per SIMD and workgroup); ns = kernel wall time / (instructions per wave x W)\",\n \"classes\": {\n");
    for (size_t i = 0; i < rows.size(); i += 4) {
        printf("  \"%s\": {", rows[i].name.c_str());
        for (int k = 0; k < 4; ++k)
            printf("\"w%d\": {\"ns_per_instr_per_simd\": %.4f, \"cycles_per_instr_per_simd\": %.3f, \"clock_ghz\": %.3f}%s", rows[i + k].w, rows[i + k].ns,
                   rows[i + k].cyc, rows[i + k].ghz, k < 3 ? ", " : "");
        printf("}%s\n", i + 4 < rows.size() ? "," : "");
    }
    printf(" }\n}\n");
    return 0;
}

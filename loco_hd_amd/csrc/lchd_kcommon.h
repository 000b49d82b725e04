// lchd_kcommon.h -- device helpers shared by every kernel translation unit (lchd_prologue / _env_* / _sweep* / _dense_fused / _kernels .hip):
// lane shuffles / DPP scans, the table-driven exp / log / pow, the lean CDF evaluation and the tag pairing rule.
#pragma once
#include "lchd_device.h"
#include "lchd_math.h"

namespace lchd {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_coord(double p, double mn, double inv, int dim) {
    int c = (int)floor((p - mn) * inv);
    return min(max(c, 0), dim - 1);
}
__device__ __forceinline__ uint64_t d2u(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double u2d(uint64_t u) { return __longlong_as_double((long long)u); }

__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    int lo = __shfl_up(__double2loint(v), d), hi = __shfl_up(__double2hiint(v), d);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __shfl_xor(__double2loint(v), m), hi = __shfl_xor(__double2hiint(v), m);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
    uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// Inclusive prefix sum across the 64 lanes with DPP adds (row_shr 1/2/4/8 inside each row of 16 lanes, then the
// two row broadcasts): 6 VALU instructions, no LDS crossbar traffic.  Lanes without a source add the identity 0.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)v;
}
// 16-bit count fields never carry into each other (every count < 65536), so a u64 of four fields scans as two u32
__device__ __forceinline__ uint64_t wave_incl_scan_fields(uint64_t x) {
    const uint32_t lo = wave_incl_scan_u32((uint32_t)x), hi = wave_incl_scan_u32((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// f64 across lanes without the LDS crossbar: DPP moves of the two halves (gfx9 DPP has whole-wave shifts)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64_or_zero(double v) {  // lanes without a source (or masked rows) read +0.0
    // all rows enabled: bound_ctrl supplies the zero itself (no v_mov of the old value); masked rows need the explicit 0
    constexpr bool BC = (ROW_MASK == 0xf);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BC);
    return __hiloint2double(hi, lo);
}
// value of lane - 1 (lane 0 keeps its own value): wave_shr:1
__device__ __forceinline__ double wave_shr1_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, returned wave-uniform (same shape as wave_incl_scan_u32; lane 63 ends up with the total)
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov_f64_or_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_mov_f64_or_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_mov_f64_or_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_mov_f64_or_zero<0x118, 0xf>(v);  // row_shr:8
    v += dpp_mov_f64_or_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_mov_f64_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {  // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// hyper_exp and uniform are the common weight functions and stay inline; the pow-based CDFs are called.
static __device__ __noinline__ double cdf_pow_based(int kind, const double* p, int np, double x) { return cdf_eval(kind, p, np, x); }
// exp(x) for x <= 0 (the weight-function CDFs only ever take exp(-b * distance)): x = n * ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^(n >> 6) * 2^((n & 63) / 64) * exp(r) with a 64-entry table and a degree-5 polynomial (r^6/720 < 4e-17).
// Within ~1.5 ulp of the correctly rounded value (libm / ocml: < 1 ulp) at a third of ocml's instruction count; exp(-inf) = 0.
static __device__ const double kExp2Tab[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
__device__ __forceinline__ double exp_fast(double x);
__device__ __forceinline__ double exp_nonpos(double x) {
    const double nd = rint(x * 0x1.71547652b82fep+6);  // 64 / ln 2
    double r = fma(-nd, 0x1.62e42fefa39efp-7, x);       // ln 2 / 64, high part
    r = fma(-nd, 0x1.abc9e3b39803fp-62, r);              // ... low part
    const int n = (int)nd;
    const double t = kExp2Tab[n & 63];
    double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = q * r;  // exp(r) - 1
    const double v = ldexp(fma(t, q, t), n >> 6);
    return x < -746.0 ? 0.0 : v;  // (below the smallest subnormal; also x = -inf, where n is meaningless)
}
// any x (the generic statistical distances): overflow -> +inf, NaN -> NaN
__device__ __forceinline__ double exp_fast(double x) {
    if (!(x > -746.0)) return x < 0.0 ? 0.0 : x;  // very negative or -inf: 0; NaN: NaN
    if (x > 709.79) return INFINITY;
    return exp_nonpos(x);  // (the reduction is the same for either sign)
}

// ln(x) for finite x > 0 (0 -> -inf; subnormals through the library): x = 2^k * m, m in [1, 2), 128 table intervals with
// c_j = 1 + (j + 1/2)/128: ln m = ln c_j + log1p(m / c_j - 1), |m / c_j - 1| < 1/256, degree-5 polynomial.  The table holds
// the rounded 1 / c_j and -ln of exactly that rounded value, so the argument reduction itself is exact in the fma.  Absolute
// error ~2e-16 (relative ~1 ulp away from x = 1) at about a third of the library routine's instructions.
static __device__ const double kLogInvTab[128] = {
    0x1.fe01fe01fe020p-1, 0x1.fa11caa01fa12p-1, 0x1.f6310aca0dbb5p-1, 0x1.f25f644230ab5p-1,
    0x1.ee9c7f8458e02p-1, 0x1.eae807aba01ebp-1, 0x1.e741aa59750e4p-1, 0x1.e3a9179dc1a73p-1,
    0x1.e01e01e01e01ep-1, 0x1.dca01dca01dcap-1, 0x1.d92f2231e7f8ap-1, 0x1.d5cac807572b2p-1,
    0x1.d272ca3fc5b1ap-1, 0x1.cf26e5c44bfc6p-1, 0x1.cbe6d9601cbe7p-1, 0x1.c8b265afb8a42p-1,
    0x1.c5894d10d4986p-1, 0x1.c26b5392ea01cp-1, 0x1.bf583ee868d8bp-1, 0x1.bc4fd65883e7bp-1,
    0x1.b951e2b18ff23p-1, 0x1.b65e2e3beee05p-1, 0x1.b37484ad806cep-1, 0x1.b094b31d922a4p-1,
    0x1.adbe87f94905ep-1, 0x1.aaf1d2f87ebfdp-1, 0x1.a82e65130e159p-1, 0x1.a574107688a4ap-1,
    0x1.a2c2a87c51ca0p-1, 0x1.a01a01a01a01ap-1, 0x1.9d79f176b682dp-1, 0x1.9ae24ea5510dap-1,
    0x1.9852f0d8ec0ffp-1, 0x1.95cbb0be377aep-1, 0x1.934c67f9b2ce6p-1, 0x1.90d4f120190d5p-1,
    0x1.8e6527af1373fp-1, 0x1.8bfce8062ff3ap-1, 0x1.899c0f601899cp-1, 0x1.87427bcc092b9p-1,
    0x1.84f00c2780614p-1, 0x1.82a4a0182a4a0p-1, 0x1.8060180601806p-1, 0x1.7e225515a4f1dp-1,
    0x1.7beb3922e017cp-1, 0x1.79baa6bb6398bp-1, 0x1.77908119ac60dp-1, 0x1.756cac201756dp-1,
    0x1.734f0c541fe8dp-1, 0x1.713786d9c7c09p-1, 0x1.6f26016f26017p-1, 0x1.6d1a62681c861p-1,
    0x1.6b1490aa31a3dp-1, 0x1.691473a88d0c0p-1, 0x1.6719f3601671ap-1, 0x1.6524f853b4aa3p-1,
    0x1.63356b88ac0dep-1, 0x1.614b36831ae94p-1, 0x1.5f66434292dfcp-1, 0x1.5d867c3ece2a5p-1,
    0x1.5babcc647fa91p-1, 0x1.59d61f123ccaap-1, 0x1.5805601580560p-1, 0x1.56397ba7c52e2p-1,
    0x1.54725e6bb82fep-1, 0x1.52aff56a8054bp-1, 0x1.50f22e111c4c5p-1, 0x1.4f38f62dd4c9bp-1,
    0x1.4d843bedc2c4cp-1, 0x1.4bd3edda68fe1p-1, 0x1.4a27fad76014ap-1, 0x1.4880522014880p-1,
    0x1.46dce34596066p-1, 0x1.453d9e2c776cap-1, 0x1.43a2730abee4dp-1, 0x1.420b5265e5951p-1,
    0x1.40782d10e6566p-1, 0x1.3ee8f42a5af07p-1, 0x1.3d5d991aa75c6p-1, 0x1.3bd60d9232955p-1,
    0x1.3a524387ac822p-1, 0x1.38d22d366088ep-1, 0x1.3755bd1c945eep-1, 0x1.35dce5f9f2af8p-1,
    0x1.34679ace01346p-1, 0x1.32f5ced6a1dfap-1, 0x1.3187758e9ebb6p-1, 0x1.301c82ac40260p-1,
    0x1.2eb4ea1fed14bp-1, 0x1.2d50a012d50a0p-1, 0x1.2bef98e5a3711p-1, 0x1.2a91c92f3c105p-1,
    0x1.293725bb804a5p-1, 0x1.27dfa38a1ce4dp-1, 0x1.268b37cd60127p-1, 0x1.2539d7e9177b2p-1,
    0x1.23eb79717605bp-1, 0x1.22a0122a0122ap-1, 0x1.21579804855e6p-1, 0x1.2012012012012p-1,
    0x1.1ecf43c7fb84cp-1, 0x1.1d8f5672e4abdp-1, 0x1.1c522fc1ce059p-1, 0x1.1b17c67f2bae3p-1,
    0x1.19e0119e0119ep-1, 0x1.18ab083902bdbp-1, 0x1.1778a191bd684p-1, 0x1.1648d50fc3201p-1,
    0x1.151b9a3fdd5c9p-1, 0x1.13f0e8d344724p-1, 0x1.12c8b89edc0acp-1, 0x1.11a3019a74826p-1,
    0x1.107fbbe011080p-1, 0x1.0f5edfab325a2p-1, 0x1.0e40655826011p-1, 0x1.0d24456359e3ap-1,
    0x1.0c0a7868b4171p-1, 0x1.0af2f722eecb5p-1, 0x1.09ddba6af8360p-1, 0x1.08cabb37565e2p-1,
    0x1.07b9f29b8eae2p-1, 0x1.06ab59c7912fbp-1, 0x1.059eea0727586p-1, 0x1.04949cc1664c5p-1,
    0x1.038c6b78247fcp-1, 0x1.02864fc7729e9p-1, 0x1.0182436517a37p-1, 0x1.0080402010080p-1,
};
static __device__ const double kLogCTab[128] = {
    0x1.ff00aa2b10ba0p-9, 0x1.7dc475f810a69p-7, 0x1.3cea44346a584p-6, 0x1.b9fc027af919ap-6,
    0x1.1b0d98923d97fp-5, 0x1.58a5bafc8e4d3p-5, 0x1.95c830ec8e3f2p-5, 0x1.d276b8adb0b56p-5,
    0x1.075983598e471p-4, 0x1.253f62f0a1417p-4, 0x1.42edcbea646eep-4, 0x1.60658a93750c4p-4,
    0x1.7da766d7b12d0p-4, 0x1.9ab42462033aep-4, 0x1.b78c82bb0eda0p-4, 0x1.d4313d66cb35dp-4,
    0x1.f0a30c01162a4p-4, 0x1.0671512ca596fp-3, 0x1.14785846742acp-3, 0x1.2266f190a5acdp-3,
    0x1.303d718e47fd5p-3, 0x1.3dfc2b0ecc62ap-3, 0x1.4ba36f39a55e5p-3, 0x1.59338d9982085p-3,
    0x1.66acd4272ad51p-3, 0x1.740f8f54037a3p-3, 0x1.815c0a14357e9p-3, 0x1.8e928de886d41p-3,
    0x1.9bb362e7dfb85p-3, 0x1.a8becfc882f19p-3, 0x1.b5b519e8fb5a6p-3, 0x1.c2968558c18c2p-3,
    0x1.cf6354e09c5ddp-3, 0x1.dc1bca0abec7bp-3, 0x1.e8c0252aa5a60p-3, 0x1.f550a564b7b37p-3,
    0x1.00e6c45ad501dp-2, 0x1.071b85fcd590dp-2, 0x1.0d46b579ab74bp-2, 0x1.136870293a8b0p-2,
    0x1.1980d2dd4236fp-2, 0x1.1f8ff9e48a2f3p-2, 0x1.2596010df763ap-2, 0x1.2b9303ab89d25p-2,
    0x1.31871c9544185p-2, 0x1.3772662bfd85cp-2, 0x1.3d54fa5c1f710p-2, 0x1.432ef2a04e813p-2,
    0x1.49006804009d0p-2, 0x1.4ec9732600269p-2, 0x1.548a2c3add263p-2, 0x1.5a42ab0f4cfe2p-2,
    0x1.5ff3070a793d4p-2, 0x1.659b57303e1f2p-2, 0x1.6b3bb2235943dp-2, 0x1.70d42e2789236p-2,
    0x1.7664e1239dbcfp-2, 0x1.7bede0a37afbfp-2, 0x1.816f41da0d495p-2, 0x1.86e919a330ba1p-2,
    0x1.8c5b7c858b48bp-2, 0x1.91c67eb45a83ep-2, 0x1.972a341135159p-2, 0x1.9c86b02dc0862p-2,
    0x1.a1dc064d5b995p-2, 0x1.a72a4966bd9e9p-2, 0x1.ac718c258b0e5p-2, 0x1.b1b1e0ebdfc5ap-2,
    0x1.b6eb59d3cf35cp-2, 0x1.bc1e08b0dad0ap-2, 0x1.c149ff115f027p-2, 0x1.c66f4e3ff6ff9p-2,
    0x1.cb8e0744d7acap-2, 0x1.d0a63ae721e64p-2, 0x1.d5b7f9ae2c684p-2, 0x1.dac353e2c5955p-2,
    0x1.dfc859906d5b5p-2, 0x1.e4c71a8687704p-2, 0x1.e9bfa659861f5p-2, 0x1.eeb20c640ddf3p-2,
    0x1.f39e5bc811e5dp-2, 0x1.f884a36fe9ec1p-2, 0x1.fd64f20f61571p-2, 0x1.011fab125ff8ap-1,
    0x1.0389eefce633cp-1, 0x1.05f14bd26459cp-1, 0x1.0855c884b450ep-1, 0x1.0ab76bece14d2p-1,
    0x1.0d163ccb9d6b8p-1, 0x1.0f7241c9b497dp-1, 0x1.11cb81787ccf8p-1, 0x1.1422025243d45p-1,
    0x1.1675cababa60ep-1, 0x1.18c6e0ff5cf07p-1, 0x1.1b154b57da29ep-1, 0x1.1d610fe677003p-1,
    0x1.1faa34b87094cp-1, 0x1.21f0bfc65beecp-1, 0x1.2434b6f483934p-1, 0x1.26762013430e0p-1,
    0x1.28b500df60783p-1, 0x1.2af15f02640acp-1, 0x1.2d2b4012edc9dp-1, 0x1.2f62a99509546p-1,
    0x1.3197a0fa7fe6ap-1, 0x1.33ca2ba328994p-1, 0x1.35fa4edd36ea0p-1, 0x1.38280fe58797fp-1,
    0x1.3a5373e7ebdf9p-1, 0x1.3c7c7fff73206p-1, 0x1.3ea33936b2f5bp-1, 0x1.40c7a4880dceap-1,
    0x1.42e9c6ddf80bfp-1, 0x1.4509a5133bb0ap-1, 0x1.472743f33aaadp-1, 0x1.4942a83a2fc07p-1,
    0x1.4b5bd6956e273p-1, 0x1.4d72d3a39fd01p-1, 0x1.4f87a3f5026e9p-1, 0x1.519a4c0ba3446p-1,
    0x1.53aad05b99b7cp-1, 0x1.55b9354b40bcep-1, 0x1.57c57f336f191p-1, 0x1.59cfb25fae87fp-1,
    0x1.5bd7d30e71c73p-1, 0x1.5ddde57149923p-1, 0x1.5fe1edad18919p-1, 0x1.61e3efda46467p-1,
};
// (the library routines stay out of line: inlined at every call site of the unrolled per-category loops they made the generic
// sweep kernels several hundred KB of code, beyond the reach of a conditional branch)
static __device__ __noinline__ double log_library(double x) { return log(x); }
static __device__ __noinline__ double pow_library(double x, double y) { return pow(x, y); }
__device__ __forceinline__ double log_fast(double x) {
    if (!(x >= 0x1p-1022) || !(x < INFINITY)) return log_library(x);  // zero, subnormal, negative, inf, NaN: the library's semantics
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const int k = (int)(u >> 52) - 1023;
    const int j = (int)((u >> 45) & 127ull);
    const double m = __longlong_as_double((long long)((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    const double r = fma(m, kLogInvTab[j], -1.0);
    double p = fma(r, 1.0 / 5.0, -0.25);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    p = p * r;
    p = fma(p, r, r);
    const double kd = (double)k;
    const double hi = fma(kd, 0x1.62e42fefa39efp-1, kLogCTab[j]);
    return fma(kd, 0x1.abc9e3b39803fp-56, hi + p);
}

// x^y for x >= 0 (probabilities, ratios of probabilities, their differences): exp(y ln x).  Relative error ~(|y ln x| + 2) ulp --
// the library's pow is correctly rounded to < 1 ulp and costs ~10x as many instructions; the statistical distances that call
// this (Hellinger with a general exponent, Renyi) sum a handful of such terms, so 1e-15 of relative error per term is far
// inside the path's 1e-6 (and the tests' 1e-11).
__device__ __forceinline__ double pow_fast(double x, double y) {
    if (x == 0.0) return y > 0.0 ? 0.0 : (y == 0.0 ? 1.0 : INFINITY);
    if (!(x > 0.0) || !(x < INFINITY)) return pow_library(x, y);  // negative, inf, NaN: the library's semantics
    return exp_fast(y * log_fast(x));
}

// `inv` = DevConfig::wf_inv of the weight function: the reciprocal of the CDF's constant divisor (<= 1 ulp from the
// reference's quotient; an f64 division costs about as much as half the exponential)
__device__ __forceinline__ double cdf_lean(int kind, const double* __restrict__ p, int np, double inv, double x) {
    if (kind == WF_HYPER_EXP) {  // cdfs.rs:5-21, same accumulation order
        double sum = 0.0;
        const int n = np / 2;
        for (int i = 0; i < n; ++i) sum += p[i] * exp_nonpos(-p[n + i] * x);  // (b_i > 0, x >= 0: weight_function.rs:31-40,97-100)
        return 1.0 - sum * inv;
    }
    if (kind == WF_UNIFORM) {  // cdfs.rs:39-45
        if (x < p[0]) return 0.0;
        if (x > p[1]) return 1.0;
        return (x - p[0]) * inv;
    }
    return cdf_pow_based(kind, p, np, x);
}

// category id of atom i: low byte | high byte, or -- a structure without high bytes -- the byte itself, its "not in the map" value
// 255 widened to 0xFFFF (a configuration with more than 255 categories has a category 255)
__device__ __forceinline__ uint32_t cat_of_atom(const CloudView& c, int64_t i) {
    const uint32_t lo = c.cat[i];
    return c.cat_hi ? (lo | ((uint32_t)c.cat_hi[i] << 8)) : (lo == 255u ? 0xFFFFu : lo);
}

__device__ __forceinline__ void wave_sync_lds() {
    // LDS operations of one wavefront execute in issue order; this only stops the compiler from moving
    // LDS accesses across the point and drains outstanding LDS traffic.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// tag_pairing_rule.rs:49-75 on interned tags
__device__ __forceinline__ bool tag_pair_accepted(const DevConfig& cfg, int32_t t_anchor, int32_t t_other) {
    if (cfg.tag_mode == 0) return (t_anchor == t_other) == (cfg.tag_accept_same != 0);
    auto contains = [&](uint64_t k) {
        int lo = 0, hi = cfg.n_tag_pairs;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const uint64_t v = cfg.tag_pairs[mid];
            if (v == k) return true;
            if (v < k) lo = mid + 1; else hi = mid;
        }
        return false;
    };
    bool acc = contains(((uint64_t)(uint32_t)t_anchor << 32) | (uint32_t)t_other);
    if (!cfg.tag_ordered) acc = acc || contains(((uint64_t)(uint32_t)t_other << 32) | (uint32_t)t_anchor);
    return cfg.tag_accepted_pairs ? acc : !acc;
}

}  // namespace lchd

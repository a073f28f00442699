// Development micro-benchmark: VALU issue cost on gfx950 in SHADER CYCLES per wave-instruction and SIMD,
// at 1, 2, 4 and 8 waves per SIMD, for the instruction kinds the path tracer's kernels are made of.
//
// Every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around a loop of 128 inline-asm
// instructions per iteration (the loop's own scalar instructions are under 3 % of the stream), and records
// which SIMD it ran on (HW_ID / XCC_ID).  The host groups waves by SIMD and reports
//   cycles per instruction and SIMD = (latest end - earliest start on that SIMD) / (instructions all its waves issued),
// median over SIMDs, together with the number of waves each SIMD really held and the clock the stamps imply.
// Occupancy is pinned with dynamic LDS: a workgroup of 4 waves (one per SIMD) asks for 160 KiB / k, so exactly
// k workgroups fit a CU.
//
// Output: one JSON object per line (profiles/r03/issue_cycles.jsonl is this program's stdout).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum Op {
  FMA_VGPR,      // v_fma_f32 v, v, v, v         (VOP3, 8 bytes)
  FMAC_VGPR,     // v_fmac_f32 v, v, v           (VOP2, 4 bytes)
  FMA_LITERAL,   // v_fmaak_f32 v, v, v, literal (VOP2 + 32-bit literal)
  FMA_SGPR,      // v_fma_f32 v, v, s, v         (one operand through the constant bus)
  MUL_VGPR,      // v_mul_f32
  ADD_VGPR,      // v_add_f32
  MUL_ADD,       // v_mul_f32 then a dependent v_add_f32: a*b+c without contraction, two instructions
  MAX_MIN,       // v_max_f32 / v_min_f32 alternating
  CMP_SEL,       // v_cmp_lt_f32 vcc + v_cndmask_b32: compare-and-select pairs
  CMP_SEL_SGPR,  // v_cmp_lt_f32 s[..] + v_cndmask_b32 ..., s[..] (VOP3 forms)
  PK_FMA,        // v_pk_fma_f32 (two fp32 per lane)
  PK_MUL,        // v_pk_mul_f32
  RCP,           // v_rcp_f32
  RSQ,           // v_rsq_f32
  FMA_F64,       // v_fma_f64
  MUL_F64,       // v_mul_f64
  CVT_F64_F32,   // v_cvt_f64_f32
  CVT_F32_F64,   // v_cvt_f32_f64
  MUL_HI_U32,    // v_mul_hi_u32
  MUL_LO_U32,    // v_mul_lo_u32
  MAD_U64_U32,   // v_mad_u64_u32
  XOR_B32,       // v_xor_b32
  DIV_SEQ,       // the compiler's IEEE fp32 division (a / b), counted as one operation
  SQRT_SEQ,      // the compiler's correctly rounded sqrtf, counted as one operation
  MOV,           // v_mov_b32
  MUL_SGPR,      // v_mul_f32 v, s, v
  MUL_INLINE,    // v_mul_f32 v, 1.0, v       (inline constant)
  MUL_LITERAL,   // v_mul_f32 v, literal, v
  ADD_INLINE,    // v_add_f32 v, 1.0, v
  SUB_VGPR,      // v_sub_f32
  FMA_INLINE,    // v_fma_f32 v, v, 1.0, 0.5  (two inline constants)
  FMA_NEG,       // v_fma_f32 v, -v, v, v     (source modifier)
  CMP_ONLY,      // v_cmp_lt_f32 vcc, v, v
  CMP_SGPR_ONLY, // v_cmp_lt_f32 s[a:b], v, v
  CNDMASK_ONLY,  // v_cndmask_b32 v, v, v, vcc
  MAX_ONLY,      // v_max_f32
  MIN_ONLY,      // v_min_f32
  MAX3,          // v_max3_f32
  MED3,          // v_med3_f32
  AND_B32,       // v_and_b32
  LSHL_B32,      // v_lshlrev_b32
  ADD_U32,       // v_add_u32
  MOV_SGPR,      // v_mov_b32 v, s
  READFIRSTLANE, // v_readfirstlane_b32 s, v
  VALU_SALU,     // v_mul_f32 and an independent s_add_u32 alternating (counted: the v_mul only)
  VALU_LDS,      // 8 v_mul_f32 per same-address ds_read_b128 (counted: the v_mul only)
  LDS_B128_SAME, // ds_read_b128, all lanes one address (counted per ds_read)
  LDS_B128_LANE, // ds_read_b128, consecutive 16-byte slots per lane
  PK_MUL_SGPR,   // v_pk_mul_f32 v, v, s[a:b] op_sel_hi:[1,0]  (scalar broadcast to both halves)
  PK_FMA_SGPR,   // v_pk_fma_f32 v, v, s[a:b], v op_sel_hi:[1,0,1]
  CNDMASK_SMASK, // v_cndmask_b32_e64 v, v, v, s[a:b]  (mask written once, before the loop)
  SALU_ONLY,     // s_and_b64 alone (counted per s_and)
  VALU2_SALU1,   // 2 v_mul_f32 per s_and_b64 (counted: v_mul)
  VALU4_SALU1,   // 4 v_mul_f32 per s_and_b64
  VALU8_SALU1,   // 8 v_mul_f32 per s_and_b64
  VALU4_SNOP1,   // 4 v_mul_f32 per s_nop 0
  VALU4_SMOV1,   // 4 v_mul_f32 per s_mov_b32
  CMP_SAND_SEL,  // v_cmp_lt_f32 vcc; s_and_b64 s, vcc, s; v_cndmask_b32_e64 (counted: 2 VALU)
  ADD3_U32,      // v_add3_u32
  LSHL_ADD_U32,  // v_lshl_add_u32
  MIN_U32,       // v_min_u32
  MIN3_U32,      // v_min3_u32
  AND_OR_B32,    // v_and_or_b32
  BFE_U32,       // v_bfe_u32
  LSHR_B32,      // v_lshrrev_b32
  CVT_F32_U32,   // v_cvt_f32_u32
  CVT_I32_F32,   // v_cvt_i32_f32
  FLOOR_F32,     // v_floor_f32
  FREXP_EXP,     // v_frexp_exp_i32_f32
  LDEXP_F32,     // v_ldexp_f32
  CMP_LT_U32,    // v_cmp_lt_u32 vcc
  ADD_CLAMP_U32, // v_add_u32 ... clamp (VOP3)
  SUB_U32,       // v_sub_u32
  MOV_DPP,       // v_mov_b32 row_shr:1 (DPP)
  BPERMUTE,      // ds_bpermute_b32
  RCP_F64,       // v_rcp_f64
  N_OPS
};

static const char *op_name[N_OPS] = {
    "v_fma_f32 (vgpr operands)", "v_fmac_f32 (VOP2)", "v_fmaak_f32 (literal)", "v_fma_f32 (one sgpr operand)", "v_mul_f32", "v_add_f32",
    "v_mul_f32 + dependent v_add_f32 (pair = 2)", "v_max_f32 / v_min_f32", "v_cmp_lt_f32 vcc + v_cndmask_b32 (pair = 2)",
    "v_cmp_lt_f32 sgpr-pair + v_cndmask_b32 (pair = 2)", "v_pk_fma_f32", "v_pk_mul_f32", "v_rcp_f32", "v_rsq_f32", "v_fma_f64", "v_mul_f64",
    "v_cvt_f64_f32", "v_cvt_f32_f64", "v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_xor_b32", "IEEE a/b sequence (1 op)",
    "IEEE sqrtf sequence (1 op)", "v_mov_b32", "v_mul_f32 (one sgpr operand)", "v_mul_f32 (inline constant 1.0)", "v_mul_f32 (literal)",
    "v_add_f32 (inline constant 1.0)", "v_sub_f32", "v_fma_f32 (inline constants 1.0, 0.5)", "v_fma_f32 (neg modifier)", "v_cmp_lt_f32 vcc alone",
    "v_cmp_lt_f32 sgpr-pair alone", "v_cndmask_b32 vcc alone", "v_max_f32 alone", "v_min_f32 alone", "v_max3_f32", "v_med3_f32", "v_and_b32",
    "v_lshlrev_b32", "v_add_u32", "v_mov_b32 v, s", "v_readfirstlane_b32", "v_mul_f32 + independent s_add_u32 (counted: v_mul)",
    "8 v_mul_f32 per same-address ds_read_b128 (counted: v_mul)", "ds_read_b128 same address", "ds_read_b128 per-lane address", "v_pk_mul_f32 (sgpr pair, op_sel broadcast)",
    "v_pk_fma_f32 (sgpr pair, op_sel broadcast)", "v_cndmask_b32_e64 sgpr mask alone", "s_and_b64 alone", "2 v_mul_f32 per s_and_b64 (counted: v_mul)",
    "4 v_mul_f32 per s_and_b64 (counted: v_mul)", "8 v_mul_f32 per s_and_b64 (counted: v_mul)", "4 v_mul_f32 per s_nop 0 (counted: v_mul)",
    "4 v_mul_f32 per s_mov_b32 (counted: v_mul)", "v_cmp_lt_f32 vcc + s_and_b64 + v_cndmask_b32_e64 (counted: 2 VALU)",
    "v_add3_u32", "v_lshl_add_u32", "v_min_u32", "v_min3_u32", "v_and_or_b32", "v_bfe_u32", "v_lshrrev_b32", "v_cvt_f32_u32", "v_cvt_i32_f32",
    "v_floor_f32", "v_frexp_exp_i32_f32", "v_ldexp_f32", "v_cmp_lt_u32 vcc alone", "v_add_u32 clamp (VOP3)", "v_sub_u32", "v_mov_b32 dpp row_shr:1",
    "ds_bpermute_b32", "v_rcp_f64"};

// instructions counted per inner step (one step = one "operation" on one chain)
static const int op_insts[N_OPS] = {1, 1, 1, 1, 1, 1, 2, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 2,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

struct Stamp {
  unsigned long long t0, t1, r0, r1;
  unsigned hw_id, xcc_id;
};

typedef float f2 __attribute__((ext_vector_type(2)));

// CHAINS independent dependency chains per wave; 128 operations per loop iteration.
template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k_issue(Stamp *stamps, float *sink, int iters, float sc0) {
  extern __shared__ float lds_pad[];
  float f[CHAINS], g[CHAINS];
  double d[CHAINS];
  f2 p[CHAINS];
  unsigned u[CHAINS];
  unsigned long long w[CHAINS];
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 q[4];
  unsigned lds_addr_same = 64 * (threadIdx.x >> 6), lds_addr_lane = 16 * threadIdx.x;
  asm volatile("" : "+v"(lds_addr_same), "+v"(lds_addr_lane));
  unsigned long long smask = 0x5555555555555555ull, spair = 0x3f8003473f800347ull;
  asm volatile("" : "+s"(smask), "+s"(spair));
  float c0 = 1.0001f, c1 = 0.5f, c2 = 0.99991f;
  asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));
  double dc0 = 1.0001, dc1 = 0.5;
  asm volatile("" : "+v"(dc0), "+v"(dc1));
  f2 pc0 = {1.0001f, 0.9999f}, pc1 = {0.5f, 0.25f};
  asm volatile("" : "+v"(pc0), "+v"(pc1));
#pragma unroll
  for (int i = 0; i < CHAINS; i++) {
    f[i] = 1.0f + 0.001f * (float)(threadIdx.x + i);
    g[i] = 2.0f + 0.002f * (float)(threadIdx.x + i);
    d[i] = (double)f[i];
    p[i] = f2{f[i], g[i]};
    u[i] = 0x9E3779B9u * (threadIdx.x + 1 + i);
    w[i] = u[i];
    asm volatile("" : "+v"(f[i]), "+v"(g[i]), "+v"(d[i]), "+v"(p[i]), "+v"(u[i]), "+v"(w[i]));
  }
  if (threadIdx.x == 100000) lds_pad[0] = 1.0f;  // keeps the dynamic LDS request alive
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 128 / CHAINS; s++) {
#pragma unroll
      for (int i = 0; i < CHAINS; i++) {
        if (OP == FMA_VGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c0), "v"(c1));
        if (OP == FMAC_VGPR) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[i]) : "v"(c2), "v"(c1));
        if (OP == FMA_LITERAL) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(f[i]) : "v"(c0));
        if (OP == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "s"(sc0), "v"(c1));
        if (OP == MUL_VGPR) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c0));
        if (OP == ADD_VGPR) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c1));
        if (OP == MUL_ADD) asm volatile("v_mul_f32 %1, %0, %2\n\tv_add_f32 %0, %1, %3" : "+v"(f[i]), "+v"(g[i]) : "v"(c2), "v"(c1));
        if (OP == MAX_MIN) {
          if (s & 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c0));
          else asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c1));
        }
        if (OP == CMP_SEL) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(f[i]) : "v"(g[i]), "v"(c0) : "vcc");
        if (OP == CMP_SEL_SGPR) {
          unsigned long long m;
          asm volatile("v_cmp_lt_f32 %1, %0, %2\n\tv_cndmask_b32 %0, %2, %3, %1" : "+v"(f[i]), "=&s"(m) : "v"(g[i]), "v"(c0));
        }
        if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));
        if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc0));
        if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        if (OP == RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
        if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dc0), "v"(dc1));
        if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc0));
        if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
        if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
        if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(c0) : "vcc");
        if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == DIV_SEQ) { f[i] = g[i] / f[i]; asm volatile("" : "+v"(f[i])); }
        if (OP == SQRT_SEQ) { f[i] = __builtin_sqrtf(f[i]) + c1; asm volatile("" : "+v"(f[i])); }
        if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(g[i]));
        if (OP == MUL_SGPR) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "s"(sc0));
        if (OP == MUL_INLINE) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(f[i]));
        if (OP == MUL_LITERAL) asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(f[i]));
        if (OP == ADD_INLINE) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(f[i]));
        if (OP == SUB_VGPR) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c1));
        if (OP == FMA_INLINE) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(f[i]));
        if (OP == FMA_NEG) asm volatile("v_fma_f32 %0, -%0, %1, %2" : "+v"(f[i]) : "v"(c0), "v"(c1));
        if (OP == CMP_ONLY) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(g[i]) : "vcc");
        if (OP == CMP_SGPR_ONLY) { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(f[i]), "v"(g[i])); }
        if (OP == CNDMASK_ONLY) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(g[i]));
        if (OP == MAX_ONLY) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c0));
        if (OP == MIN_ONLY) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c0));
        if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c0), "v"(c1));
        if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c0), "v"(c1));
        if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == LSHL_B32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
        if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == MOV_SGPR) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "s"(sc0));
        if (OP == READFIRSTLANE) { unsigned r; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(r) : "v"(f[i])); }
        if (OP == VALU_SALU) { unsigned r; asm volatile("v_mul_f32 %0, %0, %2\n\ts_add_u32 %1, %3, 1" : "+v"(f[i]), "=s"(r) : "v"(c0), "s"(iters) : "scc"); }
        if (OP == VALU_LDS) {
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c0));
          if (i == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(q[s & 3]) : "v"(lds_addr_same));
        }
        if (OP == LDS_B128_SAME) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(lds_addr_same));
        if (OP == LDS_B128_LANE) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(lds_addr_lane));
        if (OP == PK_MUL_SGPR) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "s"(spair));
        if (OP == PK_FMA_SGPR) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(spair), "v"(pc1));
        if (OP == CNDMASK_SMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g[i]), "s"(smask));
        if (OP == SALU_ONLY) { unsigned long long m; asm volatile("s_and_b64 %0, %1, %1" : "=s"(m) : "s"(smask) : "scc"); }
        if (OP == VALU2_SALU1 || OP == VALU4_SALU1 || OP == VALU8_SALU1 || OP == VALU4_SNOP1 || OP == VALU4_SMOV1) {
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c0));
          const int every = OP == VALU2_SALU1 ? 2 : OP == VALU8_SALU1 ? 8 : 4;
          if ((s * CHAINS + i) % every == 0) {
            unsigned long long m; unsigned r;
            if (OP == VALU4_SNOP1) asm volatile("s_nop 0");
            else if (OP == VALU4_SMOV1) asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "s"(iters));
            else asm volatile("s_and_b64 %0, %1, %1" : "=s"(m) : "s"(smask) : "scc");
          }
        }
        if (OP == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c0), "v"(c1));
        if (OP == LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == MIN3_U32) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c0), "v"(c1));
        if (OP == AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(c0), "v"(c1));
        if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));
        if (OP == LSHR_B32) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u[i]));
        if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
        if (OP == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(f[i]));
        if (OP == FLOOR_F32) asm volatile("v_floor_f32 %0, %0" : "+v"(f[i]));
        if (OP == FREXP_EXP) asm volatile("v_frexp_exp_i32_f32 %0, %1" : "=v"(u[i]) : "v"(f[i]));
        if (OP == LDEXP_F32) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f[i]) : "v"(lds_addr_same));
        if (OP == CMP_LT_U32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(c0) : "vcc");
        if (OP == ADD_CLAMP_U32) asm volatile("v_add_u32_e64 %0, %0, %1 clamp" : "+v"(u[i]) : "v"(c0));
        if (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(c0));
        if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]) : "v"(g[i]));
        if (OP == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(q[i & 3].x) : "v"(lds_addr_lane), "v"(g[i]));
        if (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        if (OP == CMP_SAND_SEL) {
          unsigned long long m;
          asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\ts_and_b64 %1, vcc, %4\n\ts_nop 0\n\tv_cndmask_b32_e64 %0, %2, %3, %1"
                       : "+v"(f[i]), "=&s"(m) : "v"(g[i]), "v"(c0), "s"(smask) : "vcc", "scc");
        }
      }
    }
    if (OP == VALU_LDS || OP == LDS_B128_SAME || OP == LDS_B128_LANE || OP == BPERMUTE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) acc += f[i] + g[i] + (float)d[i] + p[i].x + p[i].y + (float)u[i] + (float)w[i];
  if (OP == VALU_LDS || OP == LDS_B128_SAME || OP == LDS_B128_LANE) {
    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    acc += q[0].x + q[1].y + q[2].z + q[3].w;
  }
  if (acc == 123.456f) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Stamp st = {t0, t1, r0, r1, hw, xcc};
    stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = st;
  }
}

struct Result {
  double cyc_per_inst_simd, cyc_per_inst_wave, mhz, ms;
  double waves_per_simd_mean;
  int waves_per_simd_min, waves_per_simd_max, simds;
};

template <int OP, int CHAINS>
int run(int k, int iters, Stamp *d_stamps, float *d_sink, Result *res) {
  const int blocks = 256 * k;
  // k workgroups per CU and no more: 160 KiB of LDS per CU
  size_t lds = k == 1 ? 96 * 1024 : (160 * 1024) / k;
  CHECK(hipFuncSetAttribute((const void *)k_issue<OP, CHAINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((k_issue<OP, CHAINS>), dim3(blocks), dim3(256), lds, 0, d_stamps, d_sink, 4, 1.0001f);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((k_issue<OP, CHAINS>), dim3(blocks), dim3(256), lds, 0, d_stamps, d_sink, iters, 1.0001f);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  std::vector<Stamp> h(blocks * 4);
  CHECK(hipMemcpy(h.data(), d_stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
  struct Simd { unsigned long long lo = ~0ull, hi = 0; int waves = 0; };
  std::map<unsigned long long, Simd> simds;
  std::vector<double> per_wave, mhz;
  const double insts = (double)iters * 128 * op_insts[OP];
  for (const Stamp &s : h) {
    // HW_ID: wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]
    unsigned long long key = ((unsigned long long)(s.xcc_id & 0xf) << 32) | (s.hw_id & 0xfff0u & ~0xc0u);
    Simd &x = simds[key];
    x.lo = std::min(x.lo, s.t0);
    x.hi = std::max(x.hi, s.t1);
    x.waves++;
    per_wave.push_back((double)(s.t1 - s.t0) / insts);
    mhz.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 100.0);
  }
  std::vector<double> per_simd;
  int wmin = 1 << 30, wmax = 0;
  double wsum = 0;
  for (auto &kv : simds) {
    per_simd.push_back((double)(kv.second.hi - kv.second.lo) / (insts * kv.second.waves));
    wmin = std::min(wmin, kv.second.waves);
    wmax = std::max(wmax, kv.second.waves);
    wsum += kv.second.waves;
  }
  auto median = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  res->cyc_per_inst_simd = median(per_simd);
  res->cyc_per_inst_wave = median(per_wave);
  res->mhz = median(mhz);
  res->ms = ms;
  res->simds = (int)simds.size();
  res->waves_per_simd_min = wmin;
  res->waves_per_simd_max = wmax;
  res->waves_per_simd_mean = wsum / simds.size();
  return 0;
}

template <int OP, int CHAINS>
int sweep(Stamp *d_stamps, float *d_sink, int iters) {
  for (int k : {1, 2, 4, 8}) {
    Result r;
    if (run<OP, CHAINS>(k, iters, d_stamps, d_sink, &r)) return 1;
    // ns per instruction and SIMD from the host-side event time: every SIMD issued k * insts instructions in ms
    double ns_evt = r.ms * 1e6 / ((double)iters * 128 * op_insts[OP] * k);
    printf("{\"op\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"cycles_per_inst_per_simd\": %.3f, "
           "\"cycles_per_inst_one_wave_sees\": %.3f, \"shader_mhz\": %.0f, \"ns_per_inst_per_simd_by_events\": %.3f, "
           "\"simds_seen\": %d, \"waves_on_a_simd_min_mean_max\": [%d, %.2f, %d], \"kernel_ms\": %.3f}\n",
           op_name[OP], CHAINS, k, r.cyc_per_inst_simd, r.cyc_per_inst_wave, r.mhz, ns_evt, r.simds, r.waves_per_simd_min,
           r.waves_per_simd_mean, r.waves_per_simd_max, r.ms);
    fflush(stdout);
  }
  return 0;
}

int main(int argc, char **argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2000;
  Stamp *d_stamps;
  float *d_sink;
  CHECK(hipMalloc(&d_stamps, 256 * 8 * 4 * sizeof(Stamp)));
  CHECK(hipMalloc(&d_sink, 64));
  // warm the clocks: half a second of work before the first measurement
  for (int i = 0; i < 40; i++) { Result r; if (run<FMA_VGPR, 8>(4, iters, d_stamps, d_sink, &r)) return 1; }
  int rc = 0;
  if (argc > 2 && atoi(argv[2]) == 2) {
    rc |= sweep<MUL_SGPR, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MUL_INLINE, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MUL_LITERAL, 8>(d_stamps, d_sink, iters);
    rc |= sweep<ADD_INLINE, 8>(d_stamps, d_sink, iters);
    rc |= sweep<SUB_VGPR, 8>(d_stamps, d_sink, iters);
    rc |= sweep<FMA_INLINE, 8>(d_stamps, d_sink, iters);
    rc |= sweep<FMA_NEG, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CMP_ONLY, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CMP_SGPR_ONLY, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CNDMASK_ONLY, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MAX_ONLY, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MIN_ONLY, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MAX3, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MED3, 8>(d_stamps, d_sink, iters);
    rc |= sweep<AND_B32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<LSHL_B32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<ADD_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MOV_SGPR, 8>(d_stamps, d_sink, iters);
    rc |= sweep<READFIRSTLANE, 8>(d_stamps, d_sink, iters);
    rc |= sweep<VALU_SALU, 8>(d_stamps, d_sink, iters);
    rc |= sweep<VALU_LDS, 8>(d_stamps, d_sink, iters);
    rc |= sweep<LDS_B128_SAME, 8>(d_stamps, d_sink, iters / 4);
    rc |= sweep<LDS_B128_LANE, 8>(d_stamps, d_sink, iters / 4);
    return rc;
  }
  if (argc > 2 && atoi(argv[2]) == 3) {
    rc |= sweep<PK_MUL_SGPR, 8>(d_stamps, d_sink, iters);
    rc |= sweep<PK_FMA_SGPR, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CNDMASK_SMASK, 8>(d_stamps, d_sink, iters);
    rc |= sweep<SALU_ONLY, 8>(d_stamps, d_sink, iters);
    rc |= sweep<VALU2_SALU1, 8>(d_stamps, d_sink, iters);
    rc |= sweep<VALU4_SALU1, 8>(d_stamps, d_sink, iters);
    rc |= sweep<VALU8_SALU1, 8>(d_stamps, d_sink, iters);
    rc |= sweep<VALU4_SNOP1, 8>(d_stamps, d_sink, iters);
    rc |= sweep<VALU4_SMOV1, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CMP_SAND_SEL, 8>(d_stamps, d_sink, iters);
    return rc;
  }
  if (argc > 2 && atoi(argv[2]) == 4) {
    rc |= sweep<ADD3_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<LSHL_ADD_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MIN_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MIN3_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<AND_OR_B32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<BFE_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<LSHR_B32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CVT_F32_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CVT_I32_F32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<FLOOR_F32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<FREXP_EXP, 8>(d_stamps, d_sink, iters);
    rc |= sweep<LDEXP_F32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<CMP_LT_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<ADD_CLAMP_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<SUB_U32, 8>(d_stamps, d_sink, iters);
    rc |= sweep<MOV_DPP, 8>(d_stamps, d_sink, iters);
    rc |= sweep<BPERMUTE, 8>(d_stamps, d_sink, iters / 4);
    rc |= sweep<RCP_F64, 8>(d_stamps, d_sink, iters);
    return rc;
  }
  rc |= sweep<FMA_VGPR, 8>(d_stamps, d_sink, iters);
  rc |= sweep<FMA_VGPR, 1>(d_stamps, d_sink, iters);   // one dependent chain: latency
  rc |= sweep<FMA_VGPR, 2>(d_stamps, d_sink, iters);
  rc |= sweep<FMA_VGPR, 4>(d_stamps, d_sink, iters);
  rc |= sweep<FMAC_VGPR, 8>(d_stamps, d_sink, iters);
  rc |= sweep<FMA_LITERAL, 8>(d_stamps, d_sink, iters);
  rc |= sweep<FMA_SGPR, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MUL_VGPR, 8>(d_stamps, d_sink, iters);
  rc |= sweep<ADD_VGPR, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MUL_ADD, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MUL_ADD, 2>(d_stamps, d_sink, iters);
  rc |= sweep<MAX_MIN, 8>(d_stamps, d_sink, iters);
  rc |= sweep<CMP_SEL, 8>(d_stamps, d_sink, iters);
  rc |= sweep<CMP_SEL_SGPR, 8>(d_stamps, d_sink, iters);
  rc |= sweep<PK_FMA, 8>(d_stamps, d_sink, iters);
  rc |= sweep<PK_MUL, 8>(d_stamps, d_sink, iters);
  rc |= sweep<RCP, 8>(d_stamps, d_sink, iters);
  rc |= sweep<RSQ, 8>(d_stamps, d_sink, iters);
  rc |= sweep<FMA_F64, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MUL_F64, 8>(d_stamps, d_sink, iters);
  rc |= sweep<CVT_F64_F32, 8>(d_stamps, d_sink, iters);
  rc |= sweep<CVT_F32_F64, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MUL_HI_U32, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MUL_LO_U32, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MAD_U64_U32, 8>(d_stamps, d_sink, iters);
  rc |= sweep<XOR_B32, 8>(d_stamps, d_sink, iters);
  rc |= sweep<MOV, 8>(d_stamps, d_sink, iters);
  rc |= sweep<DIV_SEQ, 8>(d_stamps, d_sink, iters / 8);
  rc |= sweep<SQRT_SEQ, 8>(d_stamps, d_sink, iters / 8);
  return rc;
}

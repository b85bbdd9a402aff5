// valu_rate_bench.hip — issue rate of the VALU instructions the query kernel leans on (gfx950), relative to v_add_u32.
// Each kernel runs 8 waves/SIMD of an unrolled, 8-chain loop of one instruction; reported: cycles per wave-instruction
// per SIMD (a full-rate wave64 op is 4 cycles on the 16-lane SIMD... measured, not assumed).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define ITER 2048
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define DEFK(NAME, DECL, OP, FOLD)                                                        \
  __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {             \
    DECL                                                                                  \
    for (int i = 0; i < ITER; ++i) { BODY8(OP) BODY8(OP) }                                \
    uint32_t acc = 0; FOLD                                                                \
    if (acc == 0x12345678u) out[threadIdx.x] = acc;                                       \
  }
#define D32 uint32_t x[8]; for (int j = 0; j < 8; ++j) x[j] = seed + threadIdx.x * (j + 1);
#define F32 for (int j = 0; j < 8; ++j) acc ^= x[j];
#define D64 uint64_t x[8]; for (int j = 0; j < 8; ++j) x[j] = ((uint64_t)seed << 20) + threadIdx.x * (j + 1);
#define F64 for (int j = 0; j < 8; ++j) acc ^= (uint32_t)x[j] ^ (uint32_t)(x[j] >> 32);

#define OP_ADD(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_MULLO(j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_MUL24(j) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_MAD24(j) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[j]) : "v"(seed));
#define OP_MULHI(j) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_XOR(j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_LSHLOR(j) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[j]) : "v"(seed));
#define OP_ALIGNBIT(j) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[j]) : "v"(seed));
#define OP_BFI(j) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_BFREV(j) asm volatile("v_bfrev_b32 %0, %0" : "+v"(x[j]));
#define OP_CNDMASK(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(seed) : "vcc");
#define OP_CNDMASK_S(j) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[j]) : "v"(seed), "s"(m64));
#define OP_CMPCND(j) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(seed) : "vcc");
#define OP_AND(j) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_OR(j) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_SHL32(j) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x[j]));
#define OP_SHR32(j) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x[j]));
#define OP_SUB(j) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_MIN(j) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
#define OP_ANDOR(j) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(seed));
#define OP_ADD3(j) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(seed));
#define OP_PERM(j) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[j]) : "v"(seed));
#define OP_MOV(j) asm volatile("v_mov_b32 %0, %1" : "=v"(x[j]) : "v"(x[(j + 1) & 7]));
#define OP_ADDLIT(j) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x[j]));
#define OP_ADDC(j) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[j]) : "v"(seed) : "vcc");
#define OP_DPP(j) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[j]));
#define OP_SHL64(j) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x[j]));
#define OP_SHR64V(j) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(x[j]) : "v"(seed));
#define OP_CMP64(j) asm volatile("v_cmp_gt_u64 vcc, %0, %1" ::"v"(x[j]), "v"(x[(j + 1) & 7]) : "vcc");
#define OP_CMP32(j) asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(x[j]), "v"(x[(j + 1) & 7]) : "vcc");
#define OP_ADD64(j) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[j]) : "v"(x[(j + 1) & 7]));
#define OP_MAD64(j) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x[j]) : "v"(seed) : "vcc");
#define OP_BPERM(j) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(x[j]) : "v"(seed));
#define OP_READLANE(j) { uint32_t s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(x[j])); asm volatile("" ::"s"(s)); }
#define OP_MBCNT(j) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x[j]) : "v"(seed));
#define OP_BCNT(j) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x[j]) : "v"(seed));

DEFK(k_add, D32, OP_ADD, F32)
DEFK(k_mullo, D32, OP_MULLO, F32)
DEFK(k_mul24, D32, OP_MUL24, F32)
DEFK(k_mad24, D32, OP_MAD24, F32)
DEFK(k_mulhi, D32, OP_MULHI, F32)
DEFK(k_xor, D32, OP_XOR, F32)
DEFK(k_lshlor, D32, OP_LSHLOR, F32)
DEFK(k_alignbit, D32, OP_ALIGNBIT, F32)
DEFK(k_bfi, D32, OP_BFI, F32)
DEFK(k_bfrev, D32, OP_BFREV, F32)
DEFK(k_cndmask, D32, OP_CNDMASK, F32)
DEFK(k_dpp, D32, OP_DPP, F32)
#define D32M D32 uint64_t m64 = __ballot(threadIdx.x & 1);
DEFK(k_cndmask_s, D32M, OP_CNDMASK_S, F32)
DEFK(k_cmpcnd, D32, OP_CMPCND, F32)
DEFK(k_and, D32, OP_AND, F32)
DEFK(k_or, D32, OP_OR, F32)
DEFK(k_shl32, D32, OP_SHL32, F32)
DEFK(k_shr32, D32, OP_SHR32, F32)
DEFK(k_sub, D32, OP_SUB, F32)
DEFK(k_min, D32, OP_MIN, F32)
DEFK(k_andor, D32, OP_ANDOR, F32)
DEFK(k_add3, D32, OP_ADD3, F32)
DEFK(k_perm, D32, OP_PERM, F32)
DEFK(k_mov, D32, OP_MOV, F32)
DEFK(k_addlit, D32, OP_ADDLIT, F32)
DEFK(k_addc, D32, OP_ADDC, F32)
DEFK(k_shl64, D64, OP_SHL64, F64)
DEFK(k_shr64v, D64, OP_SHR64V, F64)
DEFK(k_cmp64, D64, OP_CMP64, F64)
DEFK(k_cmp32, D32, OP_CMP32, F32)
DEFK(k_add64, D64, OP_ADD64, F64)
DEFK(k_mad64, D64, OP_MAD64, F64)
DEFK(k_bperm, D32, OP_BPERM, F32)
DEFK(k_readlane, D32, OP_READLANE, F32)
DEFK(k_mbcnt, D32, OP_MBCNT, F32)
DEFK(k_bcnt, D32, OP_BCNT, F32)

int main() {
  uint32_t* out; CK(hipMalloc(&out, 4096));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1000.0;
  printf("instr,cycles_per_wave_instr_per_simd,relative_to_add   (%d CUs, %.0f MHz nominal)\n", cus, mhz);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double base = 0;
#define RUN(K, NAME) { const int blocks = cus * 8; K<<<blocks, 256>>>(out, 12345); CK(hipDeviceSynchronize());       \
    CK(hipEventRecord(e0)); K<<<blocks, 256>>>(out, 12345); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));  \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                                  \
    const double instr_per_simd = 8.0 /*waves per SIMD*/ * ITER * 16.0;                                              \
    const double cyc = ms * 1e-3 * mhz * 1e6 / instr_per_simd; if (base == 0) base = cyc;                            \
    printf("%s,%.2f,%.2f\n", NAME, cyc, cyc / base); }
  RUN(k_add, "v_add_u32") RUN(k_xor, "v_xor_b32") RUN(k_mullo, "v_mul_lo_u32") RUN(k_mulhi, "v_mul_hi_u32")
  RUN(k_mul24, "v_mul_u32_u24") RUN(k_mad24, "v_mad_u32_u24") RUN(k_mad64, "v_mad_u64_u32") RUN(k_lshlor, "v_lshl_or_b32")
  RUN(k_alignbit, "v_alignbit_b32") RUN(k_bfi, "v_bfi_b32") RUN(k_bfrev, "v_bfrev_b32") RUN(k_cndmask, "v_cndmask_b32")
  RUN(k_dpp, "v_mov_b32_dpp") RUN(k_shl64, "v_lshlrev_b64(imm)") RUN(k_shr64v, "v_lshrrev_b64(vgpr)") RUN(k_cmp64, "v_cmp_gt_u64")
  RUN(k_cmp32, "v_cmp_gt_u32") RUN(k_add64, "v_lshl_add_u64") RUN(k_bperm, "ds_bpermute_b32+wait") RUN(k_readlane, "v_readlane_b32")
  RUN(k_cndmask_s, "v_cndmask_b32_e64(sgpr mask)") RUN(k_cmpcnd, "v_cmp_gt_u32+v_cndmask(vcc) pair") RUN(k_and, "v_and_b32") RUN(k_or, "v_or_b32")
  RUN(k_shl32, "v_lshlrev_b32") RUN(k_shr32, "v_lshrrev_b32") RUN(k_sub, "v_sub_u32") RUN(k_min, "v_min_u32") RUN(k_andor, "v_and_or_b32")
  RUN(k_add, "v_add_u32 (again)")
  RUN(k_mbcnt, "v_mbcnt_lo") RUN(k_bcnt, "v_bcnt_u32_b32")
  return 0;
}

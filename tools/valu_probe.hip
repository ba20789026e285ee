// valu_probe.hip -- issue cost of the vector instructions the count kernel is made of, on one CU: 16 waves (4 per SIMD), each
// running a long unrolled run of ONE instruction on independent registers; prints clocks per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o build/valu_probe && build/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM, CONSTR_T)                                                                                    \
  __global__ __launch_bounds__(1024) void NAME(u64* out, int iters, u64 seed) {                                      \
    CONSTR_T a0 = (CONSTR_T)(seed + threadIdx.x), a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = (CONSTR_T)(seed | 3);   \
    for (int i = 0; i < iters; ++i) {                                                                                \
      REP8(asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b));                              \
           asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b));)                             \
    }                                                                                                                \
    out[threadIdx.x] = (u64)(a0 ^ a1 ^ a2 ^ a3);                                                                     \
  }

BODY(k_add_u32, "v_add_u32 %0, %0, %1", unsigned)
BODY(k_xor_b32, "v_xor_b32 %0, %0, %1", unsigned)
BODY(k_mul_u24, "v_mul_u32_u24 %0, %0, %1", unsigned)
BODY(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", unsigned)
BODY(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7", unsigned)
BODY(k_bfe, "v_bfe_u32 %0, %0, 3, 20", unsigned)
BODY(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1", unsigned)
BODY(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1", unsigned)
BODY(k_lshl_b64, "v_lshlrev_b64 %0, 2, %0", u64)
BODY(k_lshr_b64, "v_lshrrev_b64 %0, 3, %0", u64)
BODY(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %1", u64)
BODY(k_mov_b64, "v_mov_b64 %0, %1", u64)
BODY(k_cmp_eq_u64, "v_cmp_eq_u64 vcc, %0, %1", u64)
BODY(k_cmp_eq_u32, "v_cmp_eq_u32 vcc, %0, %1", unsigned)
BODY(k_cmp_lt_i32_s, "v_cmp_lt_i32 s[20:21], %0, %1", unsigned)
BODY(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", unsigned)
BODY(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", unsigned)
BODY(k_cndmask_01, "v_cndmask_b32_e64 %0, 0, 1, s[20:21]", unsigned)
BODY(k_and_or, "v_and_or_b32 %0, %0, %1, %1", unsigned)
BODY(k_add3, "v_add3_u32 %0, %0, %1, %1", unsigned)
BODY(k_perm, "v_perm_b32 %0, %0, %1, %1", unsigned)
BODY(k_readlane, "v_readlane_b32 s22, %0, 3", unsigned)
BODY(k_writelane, "v_writelane_b32 %0, s22, 3", unsigned)
BODY(k_cmp_cnd_vcc, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc", unsigned)
BODY(k_cmp_cnd_sgpr, "v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]", unsigned)
BODY(k_cnd_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc", unsigned)
BODY(k_addc_vcc, "v_addc_co_u32 %0, vcc, %0, %1, vcc", unsigned)
BODY(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", unsigned)
BODY(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf", unsigned)
BODY(k_and_b32, "v_and_b32 %0, %0, %1", unsigned)
BODY(k_lshl_b32, "v_lshlrev_b32 %0, 3, %0", unsigned)
BODY(k_min_u32, "v_min_u32 %0, %0, %1", unsigned)
BODY(k_mov_b32, "v_mov_b32 %0, %1", unsigned)
BODY(k_sub_u32, "v_sub_u32 %0, %0, %1", unsigned)
BODY(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0", unsigned)
BODY(k_sdwa_and, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", unsigned)
BODY(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1", unsigned)

__global__ __launch_bounds__(1024) void k_cndmask_only1(u64* out, int iters, u64 seed) {
  unsigned a[16];
  for (int j = 0; j < 16; ++j) a[j] = (unsigned)seed + threadIdx.x * j;
  unsigned b = (unsigned)seed | 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b));
  }
  unsigned x = 0;
  for (int j = 0; j < 16; ++j) x ^= a[j];
  out[threadIdx.x] = x;
}

template <class K> static void run(const char* name, K k, u64* d, double mhz) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, 100, 12345ull);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, iters, 12345ull);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)iters * 32 * 4;  // 4 waves per SIMD
  printf("%-16s %7.3f ms  %.2f clocks per wave-instruction per SIMD (at %.0f MHz)\n", name, ms, ms * 1e-3 * mhz * 1e6 / insts_per_simd, mhz);
}

int main() {
  u64* d; (void)hipMalloc(&d, 1024 * 8);
  int khz = 0; (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  const double mhz = khz / 1000.0;
#define R(k) run(#k, k, d, mhz)
  R(k_add_u32); R(k_xor_b32); R(k_mul_u24); R(k_mul_lo); R(k_alignbit); R(k_bfe); R(k_lshl_add); R(k_mad_u24);
  R(k_cndmask_s); R(k_cndmask_01); R(k_cndmask_only1); R(k_and_or); R(k_add3); R(k_perm); R(k_readlane); R(k_writelane);
  R(k_cmp_cnd_vcc); R(k_cmp_cnd_sgpr); R(k_cnd_e64_vcc); R(k_addc_vcc); R(k_mov_dpp); R(k_add_dpp); R(k_and_b32); R(k_lshl_b32); R(k_min_u32); R(k_mov_b32); R(k_sub_u32);
  R(k_lshl_b64); R(k_lshr_b64); R(k_lshl_add_u64); R(k_mov_b64); R(k_cmp_eq_u64); R(k_cmp_eq_u32); R(k_cmp_lt_i32_s); R(k_cndmask);
  R(k_mbcnt); R(k_sdwa_and); R(k_pk_add_u16);
  return 0;
}

// Issue-cost microbenchmark for the instructions the thrower's inner loop is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_cost scripts/ubench/issue_cost.hip && /tmp/issue_cost
// Each kernel runs ITER iterations of 8 independent chains of one instruction in every lane of a full chip
// (1024 workgroups x 256 threads); cost = time relative to v_add_u32 (4 cycles per wave64 on a SIMD16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096

#define BODY8(STMT) \
  STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8];
  float f[8];
  uint64_t w[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 9u + i; f[i] = 1.0f + (float)(a[i] & 1023) * 1e-3f; w[i] = ((uint64_t)a[i] << 32) | (a[i] * 3u); }
  for (int it = 0; it < ITER; ++it) {
    if (OP == 0) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 1) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 2) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 3) {
      // MWC step: x = mul * lo(x) + hi(x)
#define S(i) { uint32_t lo = (uint32_t)w[i], hi = (uint32_t)(w[i] >> 32); uint64_t r; \
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(seed), "v"((uint64_t)hi) : "vcc"); w[i] = r; }
      BODY8(S)
#undef S
    } else if (OP == 4) {
#define S(i) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 5) {
#define S(i) asm volatile("v_sin_f32 %0, %0" : "+v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 6) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
      BODY8(S)
#undef S
    } else if (OP == 7) {
#define S(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i])); asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 8) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %0, 11" : "+v"(a[i]));
      BODY8(S)
#undef S
    } else if (OP == 9) {
#define S(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 10) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 11) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %0, 9, %1" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 12) {
#define S(i) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 13) {
#define S(i) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 14) {
#define S(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 15) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      BODY8(S)
#undef S
    } else if (OP == 16) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
      BODY8(S)
#undef S
    } else if (OP == 17) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 18) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(w[i]));
      BODY8(S)
#undef S
    } else if (OP == 19) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
      BODY8(S)
#undef S
    } else if (OP == 20) {
#define S(i) asm volatile("v_trunc_f32 %0, %0" : "+v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 21) {
#define S(i) asm volatile("v_floor_f32 %0, %0" : "+v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 22) {
#define S(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
      BODY8(S)
#undef S
    } else if (OP == 23) {
#define S(i) asm volatile("v_and_or_b32 %0, %0, %1, 1.0" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 24) {
#define S(i) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f[i]) : "v"(a[i])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 25) {
#define S(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i])); asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(a[i]));
      BODY8(S)
#undef S
    } else if (OP == 26) {
#define S(i) asm volatile("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(seed), "v"(a[(i + 1) & 7]) : "vcc");
      BODY8(S)
#undef S
    } else if (OP == 27) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
      BODY8(S)
#undef S
    } else if (OP == 28) {
#define S(i) asm volatile("v_cos_f32 %0, %0" : "+v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 29) {
      // the mix of k_lane: 4 transcendentals among 12 full-rate instructions
#define S(i) asm volatile("v_log_f32 %0, %0\n\tv_xor_b32 %1, %1, %2\n\tv_add_u32 %1, %1, %2\n\tv_xor_b32 %1, %1, %2" : "+v"(f[i]), "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(f[i]) ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  if (r == 0x12345678u) out[0] = r;
}

template <int OP>
double run(uint32_t* d) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<OP><<<1024, 256>>>(d, 12345u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 3; ++i) k<OP><<<1024, 256>>>(d, 12345u + i);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / 3;
}

int main() {
  uint32_t* d; (void)hipMalloc(&d, 64);
  const char* names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32 (MWC step)", "v_log_f32", "v_sin_f32",
                         "v_fma_f32", "v_cvt_f32_u32 + v_cvt_u32_f32 (2 instr)", "v_alignbit_b32", "v_mul_u32_u24", "v_mad_u32_u24",
                         "v_lshl_add_u32", "v_xad_u32", "v_cvt_flr_i32_f32", "v_sqrt_f32", "v_xor_b32", "v_pk_fma_f32",
                         "v_rcp_f32", "v_mul_f64", "v_fma_f64", "v_trunc_f32", "v_floor_f32", "v_lshlrev_b32", "v_and_or_b32",
                         "v_cvt_f32_u32_sdwa + v_add_u32 (2 instr)", "v_cvt_i32_f32 + v_add_f32 (2 instr)",
                         "v_cmp_eq_u32 + v_addc_co_u32 (2 instr)", "v_add_f32", "v_cos_f32",
                         "v_log_f32 + 3 full-rate (4 instr)"};
  double t[30];
  t[0] = run<0>(d); t[1] = run<1>(d); t[2] = run<2>(d); t[3] = run<3>(d); t[4] = run<4>(d); t[5] = run<5>(d);
  t[6] = run<6>(d); t[7] = run<7>(d); t[8] = run<8>(d); t[9] = run<9>(d); t[10] = run<10>(d); t[11] = run<11>(d);
  t[12] = run<12>(d); t[13] = run<13>(d); t[14] = run<14>(d); t[15] = run<15>(d); t[16] = run<16>(d); t[17] = run<17>(d);
  t[18] = run<18>(d); t[19] = run<19>(d);
  t[20] = run<20>(d); t[21] = run<21>(d); t[22] = run<22>(d); t[23] = run<23>(d); t[24] = run<24>(d); t[25] = run<25>(d);
  t[26] = run<26>(d); t[27] = run<27>(d); t[28] = run<28>(d); t[29] = run<29>(d);
  // absolute price too: ITER x 8 statements per wave, 4 waves per SIMD, at the clock the chip actually ran
  printf("(absolute: v_add_u32 = %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz)\n", t[0] * 1e6 / (4096.0 * 8 * 4),
         t[0] * 1e6 / (4096.0 * 8 * 4) * 2.4);
  for (int i = 0; i < 30; ++i) printf("%-44s %8.3f ms  = %5.2f x v_add_u32 (~%4.1f cycles per wave64)\n", names[i], t[i], t[i] / t[0], 4.0 * t[i] / t[0]);
  return 0;
}

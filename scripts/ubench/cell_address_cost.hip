// Issue cost of the instructions that turn an electron's position into its tile address (gfx950), relative to
// v_add_u32.  Written for the round-6 change of k_lane to bin-local coordinates: the new sequence
// (v_cvt_flr_i32_f32 x 2, v_lshl_add_u32 with a VGPR addend, v_mad_i32_i24) against the old one (v_cvt_i32_f32 x 2,
// v_lshl_add_u32 with an SGPR addend, v_mad_u32_u24).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cell_address_cost scripts/ubench/cell_address_cost.hip && /tmp/cell_address_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 4096
#define BODY8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int sgpr) {
  uint32_t a[8];
  float f[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 9u + i; f[i] = 1.0f + (float)(a[i] & 1023) * 1e-3f; }
  const uint32_t vo = threadIdx.x * 4u + seed;
  for (int it = 0; it < ITER; ++it) {
    if (OP == 0) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
      BODY8(S)
#undef S
    } else if (OP == 1) {
#define S(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 2) {
#define S(i) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      BODY8(S)
#undef S
    } else if (OP == 3) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sgpr), "v"(vo));
      BODY8(S)
#undef S
    } else if (OP == 4) {
#define S(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sgpr), "v"(vo));
      BODY8(S)
#undef S
    } else if (OP == 5) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "s"(sgpr));
      BODY8(S)
#undef S
    } else if (OP == 6) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(vo));
      BODY8(S)
#undef S
    } else if (OP == 7) {
      // old sequence, per chain
#define S(i) asm volatile("v_cvt_i32_f32 %0, %1\n\tv_lshl_add_u32 %0, %0, 2, %2\n\tv_mad_u32_u24 %0, %0, %2, %0" : "=&v"(a[i]) : "v"(f[i]), "s"(sgpr));
      BODY8(S)
#undef S
    } else if (OP == 8) {
      // new sequence, per chain
#define S(i) asm volatile("v_cvt_flr_i32_f32 %0, %1\n\tv_lshl_add_u32 %0, %0, 2, %3\n\tv_mad_i32_i24 %0, %0, %2, %0" : "=&v"(a[i]) : "v"(f[i]), "s"(sgpr), "v"(vo));
      BODY8(S)
#undef S
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(f[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
float run(uint32_t* d, const char* name, float base, int per_chain) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 12345u, 77);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 12345u, 77);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("%-44s %8.3f ms  %6.2f x v_add_u32 per instruction\n", name, ms, base > 0 ? ms / base / per_chain : 1.0f);
  return ms;
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 1024 * 256 * 4);
  const float b = run<0>(d, "v_add_u32", 0.f, 1);
  run<1>(d, "v_cvt_i32_f32", b, 1);
  run<2>(d, "v_cvt_flr_i32_f32", b, 1);
  run<3>(d, "v_mad_u32_u24 v, v, s, v", b, 1);
  run<4>(d, "v_mad_i32_i24 v, v, s, v", b, 1);
  run<5>(d, "v_lshl_add_u32 v, v, 2, s", b, 1);
  run<6>(d, "v_lshl_add_u32 v, v, 2, v", b, 1);
  run<7>(d, "old: cvt_i32, lshl_add s, mad_u32_u24", b, 3);
  run<8>(d, "new: cvt_flr, lshl_add v, mad_i32_i24", b, 3);
  return 0;
}

// How far is the float32 fast path of the replay thrower (k_throw.h, RNG_MODE 0) from the fp64 Box-Muller it stands in
// for?  Exhaustive over the 2^31 values a rand_r call can return (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/replay_fast_error scripts/ubench/replay_fast_error.hip && /tmp/replay_fast_error
//   angle : | v_cos_f32(k 2^-31) - cos(2 pi k / 2147483647) |, the same for sin                 (k = 0 .. 2^31 - 1)
//   radius: | R32(k) - sqrt(-2 log(k / 2147483647)) |, R32 = sqrt(-2 ln2 ((e - 31) + log2 m)), (float)k = m 2^e, by size of R
// The fp64 side is the code of the slow path (ocml sin / cos / log / sqrt), whose results equal glibc's on every
// electron of the reference's golden frames.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ void fmax_atomic(float* p, float v) { atomicMax((int*)p, __float_as_int(v)); }   // v >= 0

__device__ __forceinline__ float radius32(uint32_t k) {
  const float kf = (float)k;
  const float m = __builtin_amdgcn_frexp_mantf(kf);            // [0.5, 1)
  const int e = __builtin_amdgcn_frexp_expf(kf);
  const float t = (float)(e - 31) + __builtin_amdgcn_logf(m);  // log2(k 2^-31) <= 0
  return __builtin_amdgcn_sqrtf(t * -1.3862943611198906f);
}

__global__ __launch_bounds__(256) void k_angle(float* out) {
  const double kPi = 3.14159265358979323846;
  float ec = 0.f, es = 0.f;
  const uint64_t n = 1ull << 31, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
    const double theta = 2. * kPi * (double)(int)k / ((double)2147483647);
    const float rev = (float)(uint32_t)k * 4.656612873077393e-10f;
    ec = fmaxf(ec, (float)fabs((double)__builtin_amdgcn_cosf(rev) - cos(theta)));
    es = fmaxf(es, (float)fabs((double)__builtin_amdgcn_sinf(rev) - sin(theta)));
  }
  fmax_atomic(&out[0], ec);
  fmax_atomic(&out[1], es);
}

// out[2 + b]: max |R32 - R64| for R64 in bucket b: [0, 1e-3), [1e-3, 1e-2), [1e-2, 0.1), [0.1, 1), [1, 3), [3, 7)
__global__ __launch_bounds__(256) void k_radius(float* out) {
  float er[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint64_t n = 1ull << 31, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; k < n; k += stride) {
    const double R = sqrt(-2. * log((double)(int)k / ((double)2147483647)));
    const float d = (float)fabs((double)radius32((uint32_t)k) - R);
    const int b = R < 1e-3 ? 0 : R < 1e-2 ? 1 : R < 0.1 ? 2 : R < 1. ? 3 : R < 3. ? 4 : 5;
    er[b] = fmaxf(er[b], d);
  }
  for (int b = 0; b < 6; ++b) fmax_atomic(&out[2 + b], er[b]);
}

int main() {
  float* d;
  float h[8] = {0};
  hipMalloc(&d, sizeof h);
  hipMemset(d, 0, sizeof h);
  hipLaunchKernelGGL(k_angle, dim3(4096), dim3(256), 0, 0, d);
  hipLaunchKernelGGL(k_radius, dim3(4096), dim3(256), 0, 0, d);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("angle : max |cos32 - cos64| = %.3e   max |sin32 - sin64| = %.3e   (all 2^31 values of k)\n", h[0], h[1]);
  const char* name[6] = {"R < 1e-3", "1e-3 <= R < 1e-2", "1e-2 <= R < 0.1", "0.1 <= R < 1", "1 <= R < 3", "3 <= R"};
  for (int b = 0; b < 6; ++b) printf("radius: max |R32 - R64| = %.3e   for %s\n", h[2 + b], name[b]);
  return 0;
}

// Micro-benchmark: how vector, scalar and LDS instructions of different wavefronts share a SIMD's issue.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/issue_rate tools/micro/issue_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define V8(a) \
  asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x), "v"(y))
#define S8(s) \
  asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n" \
               "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n" \
               : "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]) : : "scc")
#define VS8(a, s) \
  asm volatile("v_fma_f32 %0, %0, %12, %13\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %12, %13\n s_add_u32 %9, %9, 1\n" \
               "v_fma_f32 %2, %2, %12, %13\n s_add_u32 %10, %10, 1\n v_fma_f32 %3, %3, %12, %13\n s_add_u32 %11, %11, 1\n" \
               "v_fma_f32 %4, %4, %12, %13\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %12, %13\n s_add_u32 %9, %9, 1\n" \
               "v_fma_f32 %6, %6, %12, %13\n s_add_u32 %10, %10, 1\n v_fma_f32 %7, %7, %12, %13\n s_add_u32 %11, %11, 1\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                 "+s"(s[0]), "+s"(s[1]), "+s"(s[2]), "+s"(s[3]) : "v"(x), "v"(y) : "scc")

typedef float v2f_t __attribute__((ext_vector_type(2)));
#define P8(b) \
  asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n" \
               "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n" \
               : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : "v"(x2), "v"(y2))

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float x, float y) {
  __shared__ float lds[1024];
  float a[8];
  unsigned s[4] = {1, 2, 3, 4};
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i;
  lds[threadIdx.x] = x; lds[threadIdx.x + 256] = y;
  __syncthreads();
  float acc = 0.f;
  v2f_t b[8], x2 = {x, x}, y2 = {y, y};
  for (int i = 0; i < 8; i++) b[i] = v2f_t{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < iters; it++) {
    if (MODE == 5) { P8(b); P8(b); P8(b); P8(b); }                     // 32 packed VALU (64 FMAs per lane)
    if (MODE == 0) { V8(a); V8(a); V8(a); V8(a); }                     // 32 VALU
    if (MODE == 1) { VS8(a, s); VS8(a, s); VS8(a, s); VS8(a, s); }     // 32 VALU + 32 SALU interleaved
    if (MODE == 2) { S8(s); S8(s); S8(s); S8(s); }                     // 32 SALU
    if (MODE == 3) {                                                   // 32 VALU + 8 broadcast ds_read_b128
      V8(a); V8(a); V8(a); V8(a);
      const float4 *p = reinterpret_cast<const float4 *>(lds) + (it & 31);
      float4 q0 = p[0], q1 = p[32], q2 = p[64], q3 = p[96], q4 = p[128], q5 = p[160], q6 = p[192], q7 = p[224];
      acc += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
    }
    if (MODE == 4) {                                                   // 8 broadcast ds_read_b128 only
      const float4 *p = reinterpret_cast<const float4 *>(lds) + (it & 31);
      float4 q0 = p[0], q1 = p[32], q2 = p[64], q3 = p[96], q4 = p[128], q5 = p[160], q6 = p[192], q7 = p[224];
      acc += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
    }
  }
  float r = acc;
  for (int i = 0; i < 8; i++) r += a[i] + b[i].x + b[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + s[0] + s[1] + s[2] + s[3];
}

template <int MODE>
void run(const char *name, int blocks_per_cu, float *d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  k<MODE><<<grid, 256>>>(d, 10, 1.0001f, 0.5f);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // each SIMD runs blocks_per_cu wavefronts; cycles per loop iteration per SIMD at 2.4 GHz nominal
  double cyc = ms * 1e-3 * 2.4e9 / iters;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.1f cyc/iter/SIMD  %.1f cyc/iter/wave\n", name, blocks_per_cu, ms, cyc, cyc / blocks_per_cu);
}

int main() {
  float *d;
  hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
  const int iters = 20000;
  for (int w : {1, 2, 4, 8}) {
    run<0>("32 VALU", w, d, iters);
    run<1>("32 VALU + 32 SALU", w, d, iters);
    run<2>("32 SALU", w, d, iters);
    run<3>("32 VALU + 8 ds_read_b128", w, d, iters);
    run<4>("8 ds_read_b128", w, d, iters);
    run<5>("32 v_pk_fma_f32", w, d, iters);
  }
  return 0;
}

// Register-only MFMA issue-rate micro-benchmarks: the measured ceilings the roofline fractions are
// quoted against (MI355X_MICROARCH.md lists no fp64 MFMA peak; the i8 figure there is a floor).
#include "rg_internal.h"

template <int KIND>
__global__ __launch_bounds__(256) void k_mfma_peak(int iters, double* sink) {
  if (KIND == 0) {  // v_mfma_f64_16x16x4_f64, 4 independent accumulators
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
  } else {  // v_mfma_i32_32x32x32_i8, 4 independent accumulators
    v16i acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (int)threadIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123456789) sink[0] = s;
  }
}

extern "C" int rg_k_mfma_peak(int kind, int iters, double* tera_ops_out) {
  double* sink = nullptr;
  if (hipMalloc((void**)&sink, 8) != hipSuccess) return RG_ERR_HIP;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * 8;  // 8 workgroups of 4 waves per CU
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    if (kind == 0) hipLaunchKernelGGL(k_mfma_peak<0>, dim3(grid), dim3(256), 0, 0, iters, sink);
    else hipLaunchKernelGGL(k_mfma_peak<1>, dim3(grid), dim3(256), 0, 0, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per = (kind == 0) ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 32;
  const double ops = per * 32.0 * iters * 4.0 /*waves*/ * grid;
  *tera_ops_out = ops / (ms * 1e-3) / 1e12;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(sink);
  return RG_OK;
}

// What do the scatter's reservations cost?  1610 workgroups x 1024 threads, every thread 8 atomic adds with return, all
// workgroups on the same 8192 cursors (the product's pattern), against: 32-bit adds (same 8-byte stride / packed),
// adds without return, cursors private to an XCD (8 copies), and 4 adds per thread (half the reservations).
//   hipcc --offload-arch=gfx950 -O3 tools/cursor_atomic_probe.hip -o build/cursor_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

// MODE 0: u64 rtn   1: u32 rtn stride 8   2: u32 rtn packed   3: u64 no return   4: u64 rtn, per-XCD copies   5: u64 rtn, 4 per thread
template <int MODE>
__global__ __launch_bounds__(1024) void probe(u64* __restrict__ cur, u64* __restrict__ sink) {
  const unsigned t = threadIdx.x;
  u64 acc = 0;
  unsigned x = 0;
  if (MODE == 4) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 7u;
  }
  constexpr int N = MODE == 5 ? 4 : 8;
  u64 r[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const unsigned b = t + i * 1024;
    if (MODE == 0 || MODE == 5) r[i] = atomicAdd(&cur[b], (u64)1);
    else if (MODE == 1) r[i] = atomicAdd((unsigned*)&cur[b], 1u);
    else if (MODE == 2) r[i] = atomicAdd((unsigned*)cur + b, 1u);
    else if (MODE == 3) { __hip_atomic_fetch_add(&cur[b], (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); r[i] = 0; }
    else r[i] = atomicAdd(&cur[(size_t)x * 8192 + b], (u64)1);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) acc += r[i];
  if (acc == 0x123456789ull) sink[0] = acc;
}

int main() {
  const int G = 1610;
  u64 *cur, *sink;
  hipMalloc(&cur, 8 * 8192 * 8);
  hipMalloc(&sink, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[6] = {"u64 with return (product)", "u32 with return, 8-byte stride", "u32 with return, packed", "u64 without return",
                          "u64 with return, cursors private to an XCD", "u64 with return, 4 per thread"};
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
      hipMemset(cur, 0, 8 * 8192 * 8);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(G), dim3(1024), 0, 0, cur, sink); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(G), dim3(1024), 0, 0, cur, sink); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(G), dim3(1024), 0, 0, cur, sink); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(G), dim3(1024), 0, 0, cur, sink); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(G), dim3(1024), 0, 0, cur, sink); break;
        default: hipLaunchKernelGGL(probe<5>, dim3(G), dim3(1024), 0, 0, cur, sink); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)G * 1024 * (mode == 5 ? 4 : 8);
      printf("rep %d  %-46s %7.1f us  %6.1f G atomics/s\n", rep, names[mode], ms * 1e3, n / (ms * 1e-3) / 1e9);
    }
  return 0;
}

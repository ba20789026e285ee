// Does the L2 combine 16-byte record stores into full lines when all writers of a line sit on one XCD?
// The scatter's record stores are written back as partial lines (553 MB for 243 MB of records): the 8 records of a
// 128-byte line come from 8 tiles, i.e. 8 workgroups on 8 different XCDs with 8 different L2s.  This probe writes the
// same amount of records with the 8 writers of every line (a) on 8 XCDs, (b) on one XCD, and times both
// (rocprofv3 --pmc WRITE_SIZE on it gives the traffic).   hipcc --offload-arch=gfx950 -O3 tools/xcd_write_probe.hip -o build/xcd_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(ulonglong2* __restrict__ out, int iters, unsigned* __restrict__ xcc_seen) {
  const unsigned G = gridDim.x, w = blockIdx.x, t = threadIdx.x;
  if (t == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcc_seen[w] = x & 15u;
  }
  for (int i = 0; i < iters; ++i) {
    // a "bucket" per thread and iteration: G/8 lines apart, 1024 buckets per iteration
    const u64 bucket = (u64)i * 1024 + t;
    u64 line;
    unsigned slot;
    if (MODE == 0) {  // the 8 writers of a line: workgroups 8q .. 8q+7 (8 XCDs)
      line = bucket * (G / 8) + w / 8;
      slot = w % 8;
    } else {          // the 8 writers of a line: workgroups of one XCD (same w % 8)
      const unsigned g = w % 8, m = w / 8;  // m = 0 .. G/8-1
      line = bucket * (G / 8) + g * (G / 64) + m / 8;
      slot = m % 8;
    }
    out[line * 8 + slot] = make_ulonglong2(bucket, (u64)w << 32 | (unsigned)i);
  }
}

int main(int argc, char** argv) {
  const int G = 256, iters = argc > 1 ? atoi(argv[1]) : 60;  // 256 x 1024 x 60 records = 252 MB
  const size_t lines = (size_t)iters * 1024 * (G / 8);
  ulonglong2* out;
  unsigned* seen;
  hipMalloc(&out, lines * 128);
  hipMalloc(&seen, G * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      hipMemset(out, 0, lines * 128);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(1024), 0, 0, out, iters, seen);
      else hipLaunchKernelGGL(probe<1>, dim3(G), dim3(1024), 0, 0, out, iters, seen);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): %.1f us for %.0f MB of records\n", mode, mode ? "one XCD per line" : "8 XCDs per line", ms * 1e3, lines * 128 / 1e6);
    }
  unsigned h[256];
  hipMemcpy(h, seen, sizeof h, hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
  printf("\n");
  return 0;
}

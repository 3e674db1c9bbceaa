// tools/barrier_microbench.hip -- what a workgroup barrier costs on gfx950, per workgroup size, and what a hand-made
// LDS-counter barrier costs beside it.  One workgroup per CU (150 KB of dynamic LDS), N barriers in a loop, nothing else.
//
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_microbench.hip -o tools/barrier_microbench && tools/barrier_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                         \
  do                                                                                     \
  {                                                                                      \
    hipError_t e_ = (x);                                                                 \
    if(e_ != hipSuccess)                                                                 \
    {                                                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));          \
      exit(1);                                                                           \
    }                                                                                    \
  } while(0)

extern __shared__ unsigned lds[];

// MODE 0: __syncthreads() (s_waitcnt + s_barrier); 1: s_barrier alone; 2: LDS counter (one atomic per wave, spin on a read);
// 3: __syncthreads() with a dependent chain of 64 additions per stage in every wave (a stage that does something)
template <int MODE> __global__ void bar_loop(const int n, long long *cycles, float *sink)
{
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  if(threadIdx.x == 0) lds[0] = 0;
  __syncthreads();
  float v = (float)threadIdx.x;
  const long long t0 = clock64();
  for(int i = 0; i < n; i++)
  {
    if(MODE == 0) __syncthreads();
    if(MODE == 1) __builtin_amdgcn_s_barrier();
    if(MODE == 2)
    {
      if(lane == 0) atomicAdd(&lds[0], 1u);
      const unsigned want = (unsigned)(i + 1) * nw;
      while(__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - (int)want < 0) __builtin_amdgcn_s_sleep(1);
    }
    if(MODE == 3)
    {
#pragma unroll
      for(int k = 0; k < 64; k++) v = v + 1.25f;
      __syncthreads();
    }
  }
  const long long t1 = clock64();
  if(threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if(v == -1.0f) sink[0] = v;
}

template <int MODE> static void run(const char *name, const int threads, const int n)
{
  long long *d_cycles;
  float *d_sink;
  CHECK(hipMalloc(&d_cycles, 256 * sizeof(long long)));
  CHECK(hipMalloc(&d_sink, 4));
  const size_t ldsb = 150 * 1024;
  CHECK(hipFuncSetAttribute((const void *)bar_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  bar_loop<MODE><<<256, threads, ldsb>>>(100, d_cycles, d_sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  bar_loop<MODE><<<256, threads, ldsb>>>(n, d_cycles, d_sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long c[256];
  CHECK(hipMemcpy(c, d_cycles, sizeof(c), hipMemcpyDeviceToHost));
  double avg = 0;
  for(int i = 0; i < 256; i++) avg += (double)c[i];
  avg /= 256;
  printf("{\"what\": \"%s\", \"threads\": %d, \"barriers\": %d, \"kernel_ms\": %.4f, \"ns_per_barrier\": %.1f, \"clock64_ticks_per_barrier\": %.1f}\n",
         name, threads, n, ms, ms * 1e6 / n, avg / n);
  CHECK(hipFree(d_cycles));
  CHECK(hipFree(d_sink));
}

int main()
{
  const int n = 20000;
  for(int threads : { 1024, 512, 256, 128, 64 })
  {
    run<0>("__syncthreads", threads, n);
    run<1>("s_barrier", threads, n);
    run<2>("lds counter", threads, n);
    run<3>("64 dependent adds + __syncthreads", threads, n);
  }
  return 0;
}

// tools/vgpr_bank_microbench.hip -- do VGPR bank conflicts slow a full-rate VALU stream on gfx950?
//
//   hipcc --offload-arch=gfx950 -O2 tools/vgpr_bank_microbench.hip -o tools/vgpr_bank_microbench && tools/vgpr_bank_microbench
//
// Round 5: after clock, occupancy, fetch latency and instruction fetch were measured and ruled out as what holds the diffusion PDE and
// the non-local-means kernel at ~3.8 cycles per VALU instruction (profiles/r05_*), the remaining suspect of a compiler-scheduled
// stream is the register file: a VALU instruction reads up to three source registers, the file has four banks (register number
// mod 4), and hipcc does not allocate registers by bank on gfx9.  This benchmark issues v_fma_f32 / v_add_f32 / v_mul_f32 with
// hand-picked register numbers -- sources in three different banks, two in one bank, all three in one bank -- four independent
// chains per wave, W = 4 waves per SIMD on every CU, and reports cycles per instruction per SIMD with the clock read beside it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                \
  do                                                                            \
  {                                                                             \
    hipError_t e_ = (x);                                                        \
    if(e_ != hipSuccess)                                                        \
    {                                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                 \
    }                                                                           \
  } while(0)

struct rec
{
  long long ticks, real;
};

// destinations v40..v43 (one per chain); sources named per variant.  The chains are dependent through the destination only where
// the variant says so (fma: d = d * s1 + s2 reads d, s1, s2).
#define INIT "v_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\t" \
             "v_mov_b32 v44, 0.5\n\tv_mov_b32 v45, 0.5\n\tv_mov_b32 v46, 0.5\n\tv_mov_b32 v47, 0.5\n\t" \
             "v_mov_b32 v48, 0.5\n\tv_mov_b32 v49, 0.5\n\tv_mov_b32 v50, 0.5\n\tv_mov_b32 v51, 0.5\n\t" \
             "v_mov_b32 v52, 0.5\n\tv_mov_b32 v53, 0.5\n\tv_mov_b32 v54, 0.5\n\tv_mov_b32 v55, 0.5\n\t"
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s70", "s71", "s72", "scc"
#define LOOP(B0, B1, B2, B3)                                                                                                        \
  asm volatile(INIT "s_mov_b32 s72, %1\n\ts_getpc_b64 s[70:71]\n\t.rept 64\n\t" B0 "\n\t" B1 "\n\t" B2 "\n\t" B3 "\n\t.endr\n\t"      \
               "s_sub_u32 s72, s72, 1\n\ts_cmp_eq_u32 s72, 0\n\ts_cbranch_scc1 1f\n\ts_setpc_b64 s[70:71]\n1:\n\tv_mov_b32 %0, v40" \
               : "=v"(res)                                                                                                          \
               : "s"(trips)                                                                                                         \
               : CLOB)

template <int V> __global__ __launch_bounds__(1024) void bench(float *out, rec *recs, const int trips)
{
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if(threadIdx.x == 0) lds[0] = 1.0f;
  __syncthreads();
  float res = 0.f;
  const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  // register number mod 4 = bank.  d = v40..43 (banks 0..3)
  if constexpr(V == 0) // fma, sources d(bank b), b+1, b+2: three different banks
    LOOP("v_fma_f32 v40, v40, v45, v46", "v_fma_f32 v41, v41, v46, v47", "v_fma_f32 v42, v42, v47, v44", "v_fma_f32 v43, v43, v44, v45");
  else if constexpr(V == 1) // fma, the two other sources in ONE bank (different from d's)
    LOOP("v_fma_f32 v40, v40, v45, v49", "v_fma_f32 v41, v41, v46, v50", "v_fma_f32 v42, v42, v47, v51", "v_fma_f32 v43, v43, v44, v48");
  else if constexpr(V == 2) // fma, all three sources in d's bank
    LOOP("v_fma_f32 v40, v40, v44, v48", "v_fma_f32 v41, v41, v45, v49", "v_fma_f32 v42, v42, v46, v50", "v_fma_f32 v43, v43, v47, v51");
  else if constexpr(V == 3) // add, two sources in different banks
    LOOP("v_add_f32 v40, v40, v45", "v_add_f32 v41, v41, v46", "v_add_f32 v42, v42, v47", "v_add_f32 v43, v43, v44");
  else if constexpr(V == 4) // add, both sources in one bank
    LOOP("v_add_f32 v40, v40, v44", "v_add_f32 v41, v41, v45", "v_add_f32 v42, v42, v46", "v_add_f32 v43, v43, v47");
  else if constexpr(V == 5) // mul then dependent add on another register pair (the convolution's pattern), different banks
    LOOP("v_mul_f32 v52, v44, v45", "v_add_f32 v40, v52, v41", "v_mul_f32 v53, v46, v47", "v_add_f32 v42, v53, v43");
  else // the same, every pair in one bank
    LOOP("v_mul_f32 v52, v44, v48", "v_add_f32 v40, v52, v48", "v_mul_f32 v53, v45, v49", "v_add_f32 v41, v53, v49");
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = res + lds[0];
  if(lane == 0)
  {
    recs[(size_t)blockIdx.x * 16 + wave].ticks = t1 - t0;
    recs[(size_t)blockIdx.x * 16 + wave].real = r1 - r0;
  }
}

int main()
{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  rec *recs;
  CHECK(hipMalloc(&out, (size_t)cus * 1024 * sizeof(float)));
  CHECK(hipMalloc(&recs, (size_t)cus * 16 * sizeof(rec)));
  std::vector<rec> h((size_t)cus * 16);
  typedef void (*kern_t)(float *, rec *, int);
  const kern_t ks[7] = { bench<0>, bench<1>, bench<2>, bench<3>, bench<4>, bench<5>, bench<6> };
  const char *const names[7] = { "v_fma_f32, sources in three banks", "v_fma_f32, two sources in one bank", "v_fma_f32, three sources in one bank",
                                 "v_add_f32, sources in two banks", "v_add_f32, both sources in one bank", "v_mul + dependent v_add, different banks",
                                 "v_mul + dependent v_add, one bank" };
  const int trips = 20000;
  printf("{\"device\": \"%s\", \"note\": \"256 instructions per trip, four chains, W waves per SIMD on all %d CUs; cycles per instruction per SIMD (s_memtime, median), clock from s_memrealtime\",\n \"runs\": [\n",
         prop.gcnArchName, cus);
  bool first = true;
  for(int v = 0; v < 7; v++)
    for(int W = 2; W <= 4; W += 2)
    {
      const size_t lds = 96 * 1024;
      CHECK(hipFuncSetAttribute((const void *)ks[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(ks[v], dim3(cus), dim3(256 * W), lds, 0, out, recs, 10);
      CHECK(hipDeviceSynchronize());
      hipLaunchKernelGGL(ks[v], dim3(cus), dim3(256 * W), lds, 0, out, recs, trips);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h.data(), recs, (size_t)cus * 16 * sizeof(rec), hipMemcpyDeviceToHost));
      std::vector<double> cyc, clk;
      for(int i = 0; i < cus * 16; i++)
      {
        if((i & 15) >= 4 * W) continue;
        cyc.push_back((double)h[i].ticks / ((double)trips * 256 * W));
        clk.push_back(h[i].real > 0 ? 100.0 * (double)h[i].ticks / (double)h[i].real : 0.0);
      }
      std::sort(cyc.begin(), cyc.end());
      std::sort(clk.begin(), clk.end());
      printf("%s  {\"stream\": \"%s\", \"W\": %d, \"cycles\": %.3f, \"sclk_mhz\": %.0f}", first ? "" : ",\n", names[v], W, cyc[cyc.size() / 2], clk[clk.size() / 2]);
      first = false;
    }
  printf("\n ]\n}\n");
  return 0;
}

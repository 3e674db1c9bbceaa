// tools/valu_clock_microbench.hip -- VALU issue cost per instruction class WITH THE SHADER CLOCK MEASURED, not assumed.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_clock_microbench.hip -o tools/valu_clock_microbench
//   tools/valu_clock_microbench > profiles/r05_valu_issue_cycles.json
//
// Round 4's review, "weak" item 9: tools/valu_microbench.hip priced the kernels' bounds with "measured issue rates" that
// were event time x an ASSUMED 2.4 GHz (v_add 2.5, v_fma 3.0 cycles against the guide's 2.0), which cannot tell "more
// cycles per instruction" from "2 cycles at a lower sustained clock".  Here every wave reads BOTH counters around its loop:
//   s_memtime      the shader clock (MI355X_MICROARCH.md: tick = shader cycle)
//   s_memrealtime  the constant 100 MHz reference
// so that   sclk = 100 MHz x d(memtime) / d(memrealtime)   and   cycles per instruction per SIMD = d(memtime) / (W x N)
// are both measured in the same loop, on every SIMD of the chip at once (the chip clocks to its power budget: one CU alone
// runs faster).  ONE workgroup per CU (it owns the CU through its LDS request), 256 x W threads = W waves per SIMD, W = 2 and
// 4, every wave released by the same barrier (a first version ran W = 8 as two workgroups per CU: their waves did not
// overlap for the whole loop and the per-wave tick count read low).  Streams: ILP independent chains per wave (1 = a dependent chain, 4, 8), 64 instructions of
// ONE class per loop trip, plus two mixes shaped like the kernels (the PDE's multiply / add / fma / compare / select mix
// and the non-local-means weight: v_mul, v_max, v_cvt, v_add_u32, v_cmp, v_cndmask, 4 x v_fma-less multiply-adds).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                                          \
  do                                                                                                      \
  {                                                                                                       \
    hipError_t e_ = (x);                                                                                  \
    if(e_ != hipSuccess)                                                                                  \
    {                                                                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                           \
      return 1;                                                                                           \
    }                                                                                                     \
  } while(0)

enum
{
  OP_FMA,
  OP_MUL,
  OP_ADD,
  OP_ADD_U32,
  OP_MAX3,
  OP_CMP,
  OP_CNDMASK,
  OP_CMP_CNDMASK, // the pair as compiled code has it: compare into VCC, select on VCC (with the hazard's wait states)
  OP_RCP,
  OP_SQRT,
  OP_DIV_SCALE,
  OP_DIV_FMAS,
  OP_DIV_FIXUP,
  OP_CVT_I32,
  OP_MIX_PDE,
  OP_MIX_NLM,
  OP_COUNT
};
static const char *const k_names[OP_COUNT]
    = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_add_u32", "v_max3_f32", "v_cmp_lt_f32", "v_cndmask_b32", "v_cmp+v_cndmask (per pair)",
        "v_rcp_f32", "v_sqrt_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_cvt_i32_f32", "mix_pde (per instruction)",
        "mix_nlm_weight (per instruction)" };
// instructions per "slot" of the stream (a slot is what the loop repeats 64 / ILP times per chain)
static const int k_per_slot[OP_COUNT] = { 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 8, 10 };

struct rec
{
  long long ticks, real;
};

template <int OP, int ILP> __global__ __launch_bounds__(1024) void bench(float *out, rec *recs, const int trips)
{
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float f[8], g[8];
  int n[8];
#pragma unroll
  for(int k = 0; k < 8; k++)
  {
    f[k] = 1.0f + 0.001f * (lane + k);
    g[k] = 0.5f + 0.002f * (lane + 2 * k);
    n[k] = lane * 3 + k;
  }
  const float c1 = 0.9999f, c2 = 1e-7f, c3 = 1.0001f;
  if(threadIdx.x == 0) lds[0] = c1;
  __syncthreads();
  const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for(int it = 0; it < trips; it++)
  {
#pragma unroll
    for(int r = 0; r < 64 / ILP; r++)
    {
#pragma unroll
      for(int k = 0; k < ILP; k++)
      {
        if constexpr(OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
        else if constexpr(OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
        else if constexpr(OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c2));
        else if constexpr(OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[k]) : "v"(lane));
        else if constexpr(OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
        else if constexpr(OP == OP_CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[k]), "v"(c1) : "vcc");
        else if constexpr(OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[k]) : "v"(c1) : "vcc");
        else if constexpr(OP == OP_CMP_CNDMASK)
          asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(f[k]) : "v"(c1), "v"(c3) : "vcc");
        else if constexpr(OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
        else if constexpr(OP == OP_SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
        else if constexpr(OP == OP_DIV_SCALE) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(f[k]) : "v"(c1) : "vcc");
        else if constexpr(OP == OP_DIV_FMAS) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2) : "vcc");
        else if constexpr(OP == OP_DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
        else if constexpr(OP == OP_CVT_I32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[k]) : "v"(f[k]));
        else if constexpr(OP == OP_MIX_PDE)
          // two taps of a convolution (mul, add, mul, sub), one fma, one max, one compare + select: the PDE's classes in its proportions
          asm volatile("v_mul_f32 %1, %0, %2\n\tv_add_f32 %0, %1, %0\n\tv_mul_f32 %1, %0, %3\n\tv_sub_f32 %0, %0, %1\n\t"
                       "v_fma_f32 %0, %0, %2, %3\n\tv_max_f32 %1, %0, %3\n\tv_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc"
                       : "+v"(f[k]), "+v"(g[k])
                       : "v"(c1), "v"(c2)
                       : "vcc");
        else if constexpr(OP == OP_MIX_NLM)
          // one weight and its four accumulations (nlm2_body.h mexp2_scaled() + accumulate): mul, max, cvt, add_u32, cmp, cndmask, 4 x (mul, add) folded as 2 + 2
          asm volatile("v_mul_f32 %1, %0, %3\n\tv_max_f32 %1, %1, %4\n\tv_cvt_i32_f32 %2, %1\n\tv_add_u32 %2, 0x3f800000, %2\n\t"
                       "v_cmp_lt_i32 vcc, %2, %5\n\tv_cndmask_b32 %1, %2, %1, vcc\n\tv_mul_f32 %2, %0, %1\n\tv_add_f32 %0, %0, %2\n\t"
                       "v_mul_f32 %2, %3, %1\n\tv_add_f32 %0, %0, %2"
                       : "+v"(f[k]), "+v"(g[k]), "+v"(n[k])
                       : "v"(c1), "v"(c2), "v"(lane)
                       : "vcc");
      }
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
  float acc = lds[0];
#pragma unroll
  for(int k = 0; k < 8; k++) acc += f[k] + g[k] + (float)n[k];
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc;
  if(lane == 0)
  {
    recs[(size_t)blockIdx.x * 16 + wave].ticks = t1 - t0;
    recs[(size_t)blockIdx.x * 16 + wave].real = r1 - r0;
  }
}

typedef void (*kern_t)(float *, rec *, int);
template <int ILP> static kern_t kernel_of(const int op)
{
  switch(op)
  {
#define CASE(O) \
  case O: return bench<O, ILP>;
    CASE(OP_FMA)
    CASE(OP_MUL) CASE(OP_ADD) CASE(OP_ADD_U32) CASE(OP_MAX3) CASE(OP_CMP) CASE(OP_CNDMASK) CASE(OP_CMP_CNDMASK) CASE(OP_RCP) CASE(OP_SQRT)
        CASE(OP_DIV_SCALE) CASE(OP_DIV_FMAS) CASE(OP_DIV_FIXUP) CASE(OP_CVT_I32) CASE(OP_MIX_PDE) CASE(OP_MIX_NLM)
#undef CASE
  }
  return nullptr;
}

int main(int argc, char **argv)
{
  const int trips = argc > 1 ? atoi(argv[1]) : 3000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float *out;
  rec *recs;
  const int maxblocks = cus * 2;
  CHECK(hipMalloc(&out, (size_t)maxblocks * 1024 * sizeof(float)));
  CHECK(hipMalloc(&recs, (size_t)maxblocks * 16 * sizeof(rec)));
  std::vector<rec> h((size_t)maxblocks * 16);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_api\": %d, \"trips\": %d, \"slots_per_wave\": %d,\n"
         " \"note\": \"per class, ILP (independent chains per wave) and W (waves per SIMD, every SIMD of the chip busy): cycles = shader cycles "
         "(s_memtime) per wave64 instruction per SIMD = d(memtime) / (W x instructions per wave), median over the waves; sclk_mhz = 100 x "
         "d(memtime) / d(memrealtime), median; ns = wall time (HIP events) per instruction per SIMD; cycles_at_2400 = what round 2 - 4 "
         "called 'measured cycles' (wall x 2.4 GHz)\",\n \"classes\": {\n",
         prop.gcnArchName, cus, prop.clockRate / 1000, trips, trips * 64);
  for(int op = 0; op < OP_COUNT; op++)
  {
    printf("  \"%s\": {", k_names[op]);
    bool first = true;
    const int ilps[3] = { 1, 4, 8 };
    for(int ii = 0; ii < 3; ii++)
      for(int W = 2; W <= 4; W += 2)
      {
        const int ILP = ilps[ii];
        if((op == OP_MIX_PDE || op == OP_MIX_NLM) && ILP == 8) continue;
        const kern_t k = ILP == 1 ? kernel_of<1>(op) : (ILP == 4 ? kernel_of<4>(op) : kernel_of<8>(op));
        // one workgroup a CU: the LDS request keeps the dispatcher from stacking two
        const size_t lds = 96 * 1024;
        CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int blocks = cus;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256 * W), lds, 0, out, recs, 20); // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256 * W), lds, 0, out, recs, trips);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        CHECK(hipMemcpy(h.data(), recs, (size_t)blocks * 16 * sizeof(rec), hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk;
        const double ninstr = (double)trips * 64.0 * k_per_slot[op]; // per wave
        for(int i = 0; i < blocks * 16; i++)
        {
          if((i & 15) >= 4 * W) continue; // waves the workgroup does not have
          cyc.push_back((double)h[i].ticks / (ninstr * W));
          clk.push_back(h[i].real > 0 ? 100.0 * (double)h[i].ticks / (double)h[i].real : 0.0);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        const double ns = (double)ms * 1e6 / (ninstr * W);
        printf("%s\n    \"ILP%d_W%d\": {\"cycles\": %.3f, \"cycles_p10\": %.3f, \"cycles_p90\": %.3f, \"sclk_mhz\": %.0f, \"sclk_mhz_min\": %.0f, "
               "\"ns\": %.4f, \"cycles_at_2400\": %.3f}",
               first ? "" : ",", ILP, W, cyc[cyc.size() / 2], cyc[cyc.size() / 10], cyc[cyc.size() * 9 / 10], clk[clk.size() / 2], clk[0], ns,
               ns * 2.4);
        first = false;
      }
    printf("}%s\n", op + 1 < OP_COUNT ? "," : "");
  }
  printf(" }\n}\n");
  return 0;
}

// tools/ifetch_microbench.hip -- does INSTRUCTION FETCH bound a large straight-line kernel on gfx950?
//
//   hipcc --offload-arch=gfx950 -O2 tools/ifetch_microbench.hip -o tools/ifetch_microbench && tools/ifetch_microbench > out.json
//
// Round 5: the diffusion PDE and the non-local-means kernel issue one VALU instruction per ~3.8 cycles and SIMD at 2.2 - 2.3 GHz,
// whatever their occupancy (the PDE at three waves per SIMD with its fetches a whole row step ahead runs as fast as at four
// waves without: gpurun A/B), where tools/valu_clock_microbench.hip's 256-byte loops issue one per 2.0.  Those
// loops live in a wave's instruction buffer; the kernels' bodies are 30 - 130 KB of straight-line code that every wave streams
// through the instruction cache (64 KB, shared by two CUs).  This benchmark runs the same full-rate instruction (v_add_f32, four
// independent chains per wave, four waves per SIMD, every CU busy) as loop bodies of 2 KB ... 256 KB, in the 4-byte encoding
// (VOP2) and in 8-byte encodings (VOP3; VOP2 + a 32-bit literal), and reports cycles per instruction per SIMD and the
// instruction bytes per cycle and CU the front end delivered.
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

#define STR_(x) #x
#define STR(x) STR_(x)
// ENC 0: v_add_f32_e32 (4 bytes); 1: v_add_f32_e64 (VOP3, 8 bytes); 2: v_add_f32_e32 with a literal operand (8 bytes)
template <int ENC, int GROUPS> __global__ __launch_bounds__(1024) void bench(float *out, rec *recs, const int trips)
{
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float f0 = 1.0f + 0.001f * lane, f1 = 0.5f + 0.002f * lane, f2 = 0.25f + 0.003f * lane, f3 = 2.0f + 0.004f * lane;
  const float c = 1e-7f;
  if(threadIdx.x == 0) lds[0] = c;
  __syncthreads();
  const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  // the loop is written in assembly: the compiler cannot size the body, and a body beyond 128 KB is out of reach of a
  // conditional branch's 16-bit offset -- the back edge is s_setpc_b64 to the address s_getpc_b64 took at the loop's head
#define BODY(INSTR)                                                                                                                   \
  asm volatile("s_mov_b32 s72, %5\n\ts_getpc_b64 s[70:71]\n\t.rept %6\n\t" INSTR " %0, %4, %0\n\t" INSTR " %1, %4, %1\n\t" INSTR " %2, %4, %2\n\t" INSTR \
               " %3, %4, %3\n\t.endr\n\ts_sub_u32 s72, s72, 1\n\ts_cmp_eq_u32 s72, 0\n\ts_cbranch_scc1 1f\n\ts_setpc_b64 s[70:71]\n1:"                    \
               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)                                                                                \
               : "v"(c), "s"(trips), "n"(GROUPS)                                                                                      \
               : "s70", "s71", "s72", "scc")
  if constexpr(ENC == 0) { BODY("v_add_f32_e32"); }
  else if constexpr(ENC == 1) { BODY("v_add_f32_e64"); }
  else
  {
    asm volatile("s_mov_b32 s72, %5\n\ts_getpc_b64 s[70:71]\n\t.rept %6\n\tv_add_f32_e32 %0, 0x33d6bf95, %0\n\tv_add_f32_e32 %1, 0x33d6bf95, %1\n\t"
                 "v_add_f32_e32 %2, 0x33d6bf95, %2\n\tv_add_f32_e32 %3, 0x33d6bf95, %3\n\t.endr\n\ts_sub_u32 s72, s72, 1\n\ts_cmp_eq_u32 s72, 0\n\t"
                 "s_cbranch_scc1 1f\n\ts_setpc_b64 s[70:71]\n1:"
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
                 : "v"(c), "s"(trips), "n"(GROUPS)
                 : "s70", "s71", "s72", "scc");
  }
#undef BODY
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = f0 + f1 + f2 + f3 + lds[0];
  if(lane == 0)
  {
    recs[(size_t)blockIdx.x * 16 + wave].ticks = t1 - t0;
    recs[(size_t)blockIdx.x * 16 + wave].real = r1 - r0;
  }
}

typedef void (*kern_t)(float *, rec *, int);
template <int ENC> static kern_t kernel_of(const int groups)
{
  switch(groups)
  {
    case 128: return bench<ENC, 128>;
    case 512: return bench<ENC, 512>;
    case 2048: return bench<ENC, 2048>;
    case 4096: return bench<ENC, 4096>;
    case 8192: return bench<ENC, 8192>;
    case 16384: return bench<ENC, 16384>;
  }
  return nullptr;
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
  printf("{\"device\": \"%s\", \"cus\": %d,\n \"note\": \"v_add_f32, four independent chains per wave, W waves per SIMD (one 256 W-thread workgroup per CU, all CUs), loop bodies of "
         "`body_bytes` of straight-line code; cycles = s_memtime ticks per instruction per SIMD (median over the waves), sclk from s_memrealtime, "
         "fetch_bytes_per_cycle_per_cu = instruction bytes the four SIMDs of a CU consumed per shader cycle\",\n \"runs\": [\n",
         prop.gcnArchName, cus);
  const int groups_list[6] = { 128, 512, 2048, 4096, 8192, 16384 };
  const char *const enc_names[3] = { "VOP2 (4 bytes)", "VOP3 (8 bytes)", "VOP2 + literal (8 bytes)" };
  bool first = true;
  for(int enc = 0; enc < 3; enc++)
    for(int gi = 0; gi < 6; gi++)
      for(int W = 2; W <= 4; W += 2)
      {
        const int groups = groups_list[gi];
        const kern_t k = enc == 0 ? kernel_of<0>(groups) : (enc == 1 ? kernel_of<1>(groups) : kernel_of<2>(groups));
        const int ninstr = groups * 4, bytes = ninstr * (enc == 0 ? 4 : 8);
        const int trips = std::max(2, (int)(6000000LL / ninstr));
        const size_t lds = 96 * 1024;
        CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3(cus), dim3(256 * W), lds, 0, out, recs, 2);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k, dim3(cus), dim3(256 * W), lds, 0, out, recs, trips);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), recs, (size_t)cus * 16 * sizeof(rec), hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk;
        for(int i = 0; i < cus * 16; i++)
        {
          if((i & 15) >= 4 * W) continue;
          cyc.push_back((double)h[i].ticks / ((double)trips * ninstr * W));
          clk.push_back(h[i].real > 0 ? 100.0 * (double)h[i].ticks / (double)h[i].real : 0.0);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        const double c = cyc[cyc.size() / 2];
        printf("%s  {\"encoding\": \"%s\", \"body_bytes\": %d, \"W\": %d, \"cycles\": %.3f, \"sclk_mhz\": %.0f, \"fetch_bytes_per_cycle_per_cu\": %.2f}",
               first ? "" : ",\n", enc_names[enc], bytes, W, c, clk[clk.size() / 2], 4.0 * (enc == 0 ? 4 : 8) / c);
        first = false;
      }
  printf("\n ]\n}\n");
  return 0;
}

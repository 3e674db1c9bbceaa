// tools/valu_microbench.hip -- issue cost of the instruction classes the pixel kernels are made of, on this chip.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_microbench.hip -o tools/valu_microbench && tools/valu_microbench > out.json
//
// Why: DESIGN.md prices the arithmetic-bound kernels (rgb_chain, rcd_tiles, nlm_chunks, diffuse_pde) against a VALU
// issue floor = sum over instruction classes of (count x cycles per wave64 instruction per SIMD).  The guide gives
// 2 cycles for full-rate binary32 on CDNA4's SIMD-32; binary64, packed and the transcendental unit are measured
// here rather than assumed.  Method: every SIMD of every CU runs W waves (W = 1, 2, 4, 8), each wave a loop of
// 8 independent chains x 16 instructions of ONE class; cycles per instruction per SIMD =
//   (s_memtime ticks of the loop) / (instructions per wave x W)   [the tick is the shader clock, MI355X_MICROARCH.md]
// and, as a cross-check that does not rely on the tick, wall time x 2.4 GHz / the same count.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>

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
  OP_FMA_F32,
  OP_MUL_F32,
  OP_ADD_F32,
  OP_PK_FMA_F32,
  OP_PK_MUL_F32,
  OP_PK_ADD_F32,
  OP_FMA_F64,
  OP_MUL_F64,
  OP_ADD_F64,
  OP_RCP_F32,
  OP_SQRT_F32,
  OP_RSQ_F32,
  OP_LOG_F32,
  OP_EXP_F32,
  OP_RCP_F64,
  OP_DIV_SCALE_F32,
  OP_DIV_FMAS_F32,
  OP_DIV_FIXUP_F32,
  OP_CVT_F64_F32,
  OP_CVT_F32_F64,
  OP_CVT_I32_F32,
  OP_CNDMASK,
  OP_CMP_F32,
  OP_MAX_F32,
  OP_AND_B32,
  OP_ADD_U32,
  OP_LSHL_ADD_U32,
  OP_MUL_LO_U32,
  OP_MAD_U32_U24,
  OP_LDEXP_F64,
  OP_FREXP_MANT_F32,
  OP_MOV_B32,
  OP_DS_READ_B32,
  OP_DS_READ_B64,
  OP_DS_READ_B128,
  OP_DS_WRITE_B32,
  OP_COUNT
};

static const char *const k_names[OP_COUNT]
    = { "v_fma_f32",        "v_mul_f32",       "v_add_f32",        "v_pk_fma_f32",  "v_pk_mul_f32",   "v_pk_add_f32",
        "v_fma_f64",        "v_mul_f64",       "v_add_f64",        "v_rcp_f32",     "v_sqrt_f32",     "v_rsq_f32",
        "v_log_f32",        "v_exp_f32",       "v_rcp_f64",        "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32",
        "v_cvt_f64_f32",    "v_cvt_f32_f64",   "v_cvt_i32_f32",    "v_cndmask_b32", "v_cmp_lt_f32",   "v_max_f32",
        "v_and_b32",        "v_add_u32",       "v_lshl_add_u32",   "v_mul_lo_u32",  "v_mad_u32_u24",  "v_ldexp_f64",
        "v_frexp_mant_f32", "v_mov_b32",       "ds_read_b32",      "ds_read_b64",   "ds_read_b128",   "ds_write_b32" };

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// one "round" = 8 instructions, one per chain; 16 rounds per loop trip = 128 instructions per trip
template <int OP> __global__ __launch_bounds__(64) void bench(float *out, long long *ticks, const int trips)
{
  __shared__ float lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  float f[8];
  double d[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  f2 p[8];
  f4 q[8];
  int n[8];
  for(int k = 0; k < 8; k++)
  {
    f[k] = 1.0f + 0.001f * (lane + k);
    d[k] = 1.0 + 0.001 * (lane + k);
    p[k] = f2{ f[k], f[k] + 0.5f };
    q[k] = f4{ f[k], f[k], f[k], f[k] };
    n[k] = lane * 3 + k;
  }
  lds[lane] = f[0];
  lds[lane + 64] = f[1];
  const float c1 = 0.9999f, c2 = 1e-7f;
  const double e1 = 0.9999, e2 = 1e-9;
  const f2 pc1 = { c1, c1 }, pc2 = { c2, c2 };
  const int laddr = lane * 4;      // byte address for 4-byte LDS accesses
  const int laddr8 = lane * 8, laddr16 = lane * 16 % 1024;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for(int it = 0; it < trips; it++)
  {
#pragma unroll
    for(int r = 0; r < 16; r++)
    {
      if constexpr(OP == OP_FMA_F32)
      {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_MUL_F32)
      {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_ADD_F32)
      {
#define X(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c2));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_PK_FMA_F32)
      {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pc1), "v"(pc2));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_PK_MUL_F32)
      {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc1));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_PK_ADD_F32)
      {
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc2));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_FMA_F64)
      {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(e1), "v"(e2));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_MUL_F64)
      {
#define X(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(e1));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_ADD_F64)
      {
#define X(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(e2));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_RCP_F32)
      {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_SQRT_F32)
      {
#define X(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_RSQ_F32)
      {
#define X(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_LOG_F32)
      {
#define X(k) asm volatile("v_log_f32 %0, %0" : "+v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_EXP_F32)
      {
#define X(k) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_RCP_F64)
      {
#define X(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_DIV_SCALE_F32)
      {
#define X(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(f[k]) : "v"(c1) : "vcc");
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_DIV_FMAS_F32)
      {
#define X(k) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2) : "vcc");
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_DIV_FIXUP_F32)
      {
#define X(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_CVT_F64_F32)
      {
#define X(k) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_CVT_F32_F64)
      {
#define X(k) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(d[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_CVT_I32_F32)
      {
#define X(k) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[k]) : "v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_CNDMASK)
      {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[k]) : "v"(c1) : "vcc");
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_CMP_F32)
      {
#define X(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[k]), "v"(c1) : "vcc");
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_MAX_F32)
      {
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_AND_B32)
      {
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[k]) : "v"(n[(k + 1) & 7]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_ADD_U32)
      {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[k]) : "v"(lane));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_LSHL_ADD_U32)
      {
#define X(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(n[k]) : "v"(lane));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_MUL_LO_U32)
      {
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[k]) : "v"(lane));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_MAD_U32_U24)
      {
#define X(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(n[k]) : "v"(lane));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_LDEXP_F64)
      {
#define X(k) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[k]) : "v"(0));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_FREXP_MANT_F32)
      {
#define X(k) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(f[k]));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_MOV_B32)
      {
#define X(k) asm volatile("v_mov_b32 %0, %1" : "=v"(f[k]) : "v"(c1));
        REP8(X)
#undef X
      }
      else if constexpr(OP == OP_DS_READ_B32)
      {
#define X(k) asm volatile("ds_read_b32 %0, %1" : "=v"(f[k]) : "v"(laddr));
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
      else if constexpr(OP == OP_DS_READ_B64)
      {
#define X(k) asm volatile("ds_read_b64 %0, %1" : "=v"(p[k]) : "v"(laddr8));
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
      else if constexpr(OP == OP_DS_READ_B128)
      {
#define X(k) asm volatile("ds_read_b128 %0, %1" : "=v"(q[k]) : "v"(laddr16));
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
      else if constexpr(OP == OP_DS_WRITE_B32)
      {
#define X(k) asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(f[k]) : "memory");
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float acc = 0.f;
  for(int k = 0; k < 8; k++) acc += f[k] + (float)d[k] + p[k].x + p[k].y + q[k].x + q[k].w + (float)n[k];
  out[(size_t)blockIdx.x * 64 + lane] = acc;
  if(lane == 0) ticks[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(float *, long long *, int);
template <int OP> struct table
{
  static void fill(kern_t *t)
  {
    t[OP] = bench<OP>;
    table<OP + 1>::fill(t);
  }
};
template <> struct table<OP_COUNT>
{
  static void fill(kern_t *) {}
};

int main(int argc, char **argv)
{
  const int trips = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  kern_t kern[OP_COUNT];
  table<0>::fill(kern);
  const int maxblocks = cus * 4 * 8;
  float *out;
  long long *ticks;
  CHECK(hipMalloc(&out, (size_t)maxblocks * 64 * sizeof(float)));
  CHECK(hipMalloc(&ticks, (size_t)maxblocks * sizeof(long long)));
  std::vector<long long> h(maxblocks);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_api\": %d, \"trips\": %d, \"instr_per_wave\": %d,\n \"note\": "
         "\"cycles per wave64 instruction per SIMD; ticks = s_memtime over the loop of the slowest wave; wall = event time x "
         "2.4 GHz; W = waves per SIMD (one 64-thread workgroup each)\",\n \"classes\": {\n",
         prop.name, cus, prop.clockRate / 1000, trips, trips * 128);
  for(int op = 0; op < OP_COUNT; op++)
  {
    printf("  \"%s\": {", k_names[op]);
    const int ws[4] = { 1, 2, 4, 8 };
    for(int wi = 0; wi < 4; wi++)
    {
      const int W = ws[wi];
      const int blocks = cus * 4 * W;
      hipLaunchKernelGGL(kern[op], dim3(blocks), dim3(64), 0, 0, out, ticks, 10); // warm-up
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(kern[op], dim3(blocks), dim3(64), 0, 0, out, ticks, trips);
      CHECK(hipEventRecord(b, 0));
      CHECK(hipEventSynchronize(b));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, a, b));
      CHECK(hipMemcpy(h.data(), ticks, (size_t)blocks * sizeof(long long), hipMemcpyDeviceToHost));
      long long mx = 0;
      double mean = 0;
      for(int i = 0; i < blocks; i++)
      {
        if(h[i] > mx) mx = h[i];
        mean += (double)h[i];
      }
      mean /= blocks;
      const double ninstr = (double)trips * 128.0 * W;
      printf("%s\"W%d\": {\"ticks_mean\": %.3f, \"ticks_max\": %.3f, \"wall\": %.3f}", wi ? ", " : "", W, mean / ninstr,
             (double)mx / ninstr, (double)ms * 1e-3 * 2.4e9 / ninstr);
    }
    printf("}%s\n", op + 1 < OP_COUNT ? "," : "");
  }
  printf(" }\n}\n");
  return 0;
}

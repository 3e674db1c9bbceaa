// tools/nan_propagation.hip -- what v_add / v_mul / v_sub make of NaN operands on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/nan_propagation.hip -o /tmp/nanp && /tmp/nanp
// Measured: of two NaN operands the result is the FIRST one (src0), quieted, as on x86; but x - NaN flips the NaN's sign
// (v_sub_f32 negates src1) where x86's subss keeps it.  Matters where a NaN's bits become a number again -- the exponent
// arithmetic of AMaZE's xdivf() -- and nowhere else (AMaZE row).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t *in, uint32_t *out)
{
  float a = __uint_as_float(in[0]), b = __uint_as_float(in[1]), r1, r2, r3, r4;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(a), "v"(b));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r2) : "v"(b), "v"(a));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r3) : "v"(a), "v"(b));
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r4) : "v"(a), "v"(b));
  out[0] = __float_as_uint(r1);
  out[1] = __float_as_uint(r2);
  out[2] = __float_as_uint(r3);
  out[3] = __float_as_uint(r4);
  out[4] = __float_as_uint(a + 1.0f);
  out[5] = __float_as_uint(0.5f - a);
}
int main()
{
  uint32_t cases[][2] = { { 0x7FC00001u, 0xFFC00002u }, { 0xFFC00002u, 0x7FC00001u }, { 0x7F800001u, 0xFFC00002u },
                          { 0xFF923456u, 0x7FC12345u }, { 0x7FC00001u, 0x3F800000u } };
  uint32_t *din, *dout, h[6];
  if(hipMalloc(&din, 8) != hipSuccess || hipMalloc(&dout, 24) != hipSuccess) return 1;
  for(auto &c : cases)
  {
    hipMemcpy(din, c, 8, hipMemcpyHostToDevice);
    k<<<1, 1>>>(din, dout);
    hipMemcpy(h, dout, 24, hipMemcpyDeviceToHost);
    printf("a %08x b %08x : a+b %08x  b+a %08x  a*b %08x  a-b %08x  a+1 %08x  0.5-a %08x\n", c[0], c[1], h[0], h[1], h[2], h[3], h[4], h[5]);
  }
  return 0;
}

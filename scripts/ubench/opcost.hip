// Per-opcode issue cost of the integer VALU ops used by the SHA-256 round and the Buzhash scan (one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// 4 independent chains so that latency never limits: a,b,c,d
#define BODY(OP) REP64(OP(a) OP(b) OP(c) OP(d))
#define K(name, OP) __global__ void name(unsigned* out, int iters, unsigned s){ unsigned a=s+threadIdx.x,b=a*3,c=a*5,d=a*7,k=s|1; \
  for(int i=0;i<iters;i++){ BODY(OP) } out[threadIdx.x]=a^b^c^d^k; }
#define OP_ADD(x)   asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_XOR(x)   asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_ADD3(x)  asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x));
#define OP_BITOP(x) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(x) : "v"(k));
#define OP_BFI(x)   asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_PERM(x)  asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_FMA(x)   asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_LSHL(x)  asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(x));
#define OP_MAX3(x)  asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_XAD(x)   asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(k));
K(k_add, OP_ADD) K(k_xor, OP_XOR) K(k_add3, OP_ADD3) K(k_align, OP_ALIGN) K(k_bitop, OP_BITOP) K(k_bfi, OP_BFI)
K(k_perm, OP_PERM) K(k_fma, OP_FMA) K(k_lshl, OP_LSHL) K(k_max3, OP_MAX3) K(k_xad, OP_XAD) K(k_lshladd, OP_LSHLADD)
typedef void (*kern)(unsigned*, int, unsigned);
int main(){
  unsigned* out; CK(hipMalloc(&out, 4096));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* n; kern k; } ks[] = {{"v_add_u32 (VOP2)",k_add},{"v_xor_b32 (VOP2)",k_xor},{"v_lshlrev_b32 (VOP2)",k_lshl},{"v_add3_u32",k_add3},
    {"v_alignbit_b32",k_align},{"v_bitop3_b32",k_bitop},{"v_bfi_b32",k_bfi},{"v_perm_b32",k_perm},{"v_max3_u32",k_max3},{"v_xad_u32",k_xad},{"v_lshl_add_u32",k_lshladd},{"v_fma_f32",k_fma}};
  const int iters=4000; const double n = (double)iters*256;
  for (int waves : {1, 1024, 2048}) for (auto &k : ks) { float ms=0;
    for(int r=0;r<2;r++){ CK(hipEventRecord(e0)); hipLaunchKernelGGL(k.k, dim3(waves), dim3(64), 0, 0, out, iters, 12345u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms,e0,e1)); }
    printf("waves=%4d %-22s %.3f ns/instr/wave  (%.2f cycles @2.4GHz)\n", waves, k.n, ms*1e6/n, ms*1e6/n*2.4); }
  return 0;
}

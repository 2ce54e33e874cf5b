// Does a wave with fewer active lanes issue integer VALU faster on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint32_t rotr(uint32_t x,int n){return __builtin_rotateright32(x,n);}
__device__ __forceinline__ uint32_t xor3(uint32_t a,uint32_t b,uint32_t c){return __builtin_amdgcn_bitop3_b32(a,b,c,0x96);}
__device__ __forceinline__ uint32_t maj3(uint32_t a,uint32_t b,uint32_t c){return __builtin_amdgcn_bitop3_b32(a,b,c,0xE8);}
#define RND(a,b,c,d,e,f,g,h,wk) do{ uint32_t t1=(h)+xor3(rotr(e,6),rotr(e,11),rotr(e,25))+((g)^((e)&((f)^(g))))+(wk); \
  uint32_t t2=xor3(rotr(a,2),rotr(a,13),rotr(a,22))+maj3(a,b,c); (d)+=t1; (h)=t1+t2; }while(0)
__global__ void k(uint32_t* out, int iters, uint32_t seed, int active) {
  if ((int)threadIdx.x >= active) return;
  uint32_t a=seed+threadIdx.x,b=a*3,c=a*5,d=a*7,e=a*11,f=a*13,g=a*17,h=a*19, w=seed;
  for(int it=0; it<iters*8; ++it){
    RND(a,b,c,d,e,f,g,h,w); RND(h,a,b,c,d,e,f,g,w); RND(g,h,a,b,c,d,e,f,w); RND(f,g,h,a,b,c,d,e,w);
    RND(e,f,g,h,a,b,c,d,w); RND(d,e,f,g,h,a,b,c,w); RND(c,d,e,f,g,h,a,b,w); RND(b,c,d,e,f,g,h,a,w);
  }
  out[blockIdx.x*64+threadIdx.x]=a^b^c^d^e^f^g^h;
}
int main(){
  uint32_t* out; CK(hipMalloc(&out, 1<<24));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters=2000;
  for (int bd : {64, 32, 16}) for (int active : {64, 32, 16, 1}) { if (active > bd) continue;
    for (int g : {1, 1024}) { float ms=0;
      for(int rep=0;rep<2;rep++){ CK(hipEventRecord(e0)); hipLaunchKernelGGL(k,dim3(g),dim3(bd),0,0,out,iters,123u,active);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms,e0,e1)); }
      printf("blockDim=%2d active=%2d waves=%4d ms=%7.3f ns/instr=%6.3f\n", bd, active, g, ms, ms*1e6/((double)iters*64*14));
    } }
  return 0;
}

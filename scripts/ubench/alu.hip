// Pure-VALU SHA-256 round loop (no memory in the loop) at controlled occupancy:
// how does per-wave instruction rate scale with waves/SIMD and active CUs on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
__device__ __forceinline__ uint32_t rotr(uint32_t x,int n){return __builtin_rotateright32(x,n);}
__device__ __forceinline__ uint32_t xor3(uint32_t a,uint32_t b,uint32_t c){return __builtin_amdgcn_bitop3_b32(a,b,c,0x96);}
__device__ __forceinline__ uint32_t maj3(uint32_t a,uint32_t b,uint32_t c){return __builtin_amdgcn_bitop3_b32(a,b,c,0xE8);}
#define RND(a,b,c,d,e,f,g,h,wk) do{ uint32_t t1=(h)+xor3(rotr(e,6),rotr(e,11),rotr(e,25))+((g)^((e)&((f)^(g))))+(wk); \
  uint32_t t2=xor3(rotr(a,2),rotr(a,13),rotr(a,22))+maj3(a,b,c); (d)+=t1; (h)=t1+t2; }while(0)

template<int UNROLL_BLOCKS>
__global__ __launch_bounds__(64) void k_rounds(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a=seed+threadIdx.x,b=a*3,c=a*5,d=a*7,e=a*11,f=a*13,g=a*17,h=a*19;
  uint32_t w[16];
  #pragma unroll
  for(int i=0;i<16;i++) w[i]=seed*(i+1)+threadIdx.x;
  for(int it=0; it<iters; ++it){
    #pragma unroll
    for(int u=0;u<UNROLL_BLOCKS;u++){
      #pragma unroll
      for(int q=0;q<64;q+=8){
        RND(a,b,c,d,e,f,g,h,w[(q+0)&15]); RND(h,a,b,c,d,e,f,g,w[(q+1)&15]); RND(g,h,a,b,c,d,e,f,w[(q+2)&15]); RND(f,g,h,a,b,c,d,e,w[(q+3)&15]);
        RND(e,f,g,h,a,b,c,d,w[(q+4)&15]); RND(d,e,f,g,h,a,b,c,w[(q+5)&15]); RND(c,d,e,f,g,h,a,b,w[(q+6)&15]); RND(b,c,d,e,f,g,h,a,w[(q+7)&15]);
      }
    }
  }
  out[blockIdx.x*64+threadIdx.x]=a^b^c^d^e^f^g^h;
}
// tiny loop body (fits any I-cache): one round per iteration, rotating names via moves avoided by 8-round body
__global__ __launch_bounds__(64) void k_small(uint32_t* out, int iters, uint32_t seed) {
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
  const int iters=2000;   // blocks (64 rounds each) per wave
  int grids[]={1,64,256,512,1024,2048,4096,8192};
  for(int variant=0;variant<3;variant++){
    for(int g: grids){
      for(int rep=0;rep<2;rep++){
        CK(hipEventRecord(e0));
        if(variant==0) hipLaunchKernelGGL(k_rounds<1>,dim3(g),dim3(64),0,0,out,iters,123u);
        else if(variant==1) hipLaunchKernelGGL(k_rounds<4>,dim3(g),dim3(64),0,0,out,iters/4,123u);
        else hipLaunchKernelGGL(k_small,dim3(g),dim3(64),0,0,out,iters,123u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms,e0,e1));
        if(rep==1){
          double instr = (double)iters*64*14;  // VALU instr per wave
          printf("variant=%d (%s) waves=%5d ms=%8.3f us/block=%6.3f ns/instr/wave=%6.3f  chip Ginstr/s=%8.2f\n", variant,
             variant==0?"64-round body 7KB":variant==1?"256-round body 28KB":"8-round body", g, ms, ms*1e3/iters, ms*1e6/instr, g*instr/ms/1e6);
        }
      }
    }
  }
  return 0;
}

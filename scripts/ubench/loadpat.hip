// Ceiling of the scan kernel's HBM access pattern without any hashing: every lane streams its own strip in
// full 128-byte lines (8 dwordx4 per line), strips 4352 B apart (A), versus a quad-cooperative pattern where the
// 4 lanes of a quad read 64 contiguous bytes per instruction (B), versus fully coalesced streaming (C).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int LINES = 34; constexpr unsigned SL = LINES*128; constexpr unsigned long long TILE = 64ull*SL;
__global__ __launch_bounds__(512, 2) void k_lane(const uint8_t* d, unsigned long long ntiles, unsigned long long* q, unsigned* out) {
  const int lane = threadIdx.x & 63; unsigned acc = 0;
  for (;;) { unsigned long long t = 0; if (lane == 0) t = atomicAdd(q, 1ull); t = __shfl(t, 0, 64); if (t >= ntiles) break;
    const uint4* p = reinterpret_cast<const uint4*>(d + t*TILE + (unsigned long long)lane*SL);
    #pragma unroll 2
    for (int line = 0; line < LINES; ++line) {
      #pragma unroll
      for (int g = 0; g < 8; ++g) { uint4 v = p[line*8+g]; acc ^= v.x ^ v.y ^ v.z ^ v.w; } } }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(512, 2) void k_quad(const uint8_t* d, unsigned long long ntiles, unsigned long long* q, unsigned* out) {
  const int lane = threadIdx.x & 63; unsigned acc = 0; const int ql = lane & 3, qb = lane & ~3;
  for (;;) { unsigned long long t = 0; if (lane == 0) t = atomicAdd(q, 1ull); t = __shfl(t, 0, 64); if (t >= ntiles) break;
    const uint8_t* tb = d + t*TILE;
    #pragma unroll 2
    for (int line = 0; line < LINES; ++line) {
      #pragma unroll
      for (int g = 0; g < 8; ++g) {  // instruction g: quad lanes read 64 contiguous bytes of strip (qb + g/2)'s line, half (g&1)
        const uint4* p = reinterpret_cast<const uint4*>(tb + (unsigned long long)(qb + (g >> 1))*SL + line*128 + (g & 1)*64 + ql*16);
        uint4 v = *p; acc ^= v.x ^ v.y ^ v.z ^ v.w; } } }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(512, 2) void k_coal(const uint8_t* d, unsigned long long nbytes, unsigned* out) {
  unsigned acc = 0; const unsigned long long n16 = nbytes/16, stride = (unsigned long long)gridDim.x*blockDim.x;
  const uint4* p = reinterpret_cast<const uint4*>(d);
  for (unsigned long long i = (unsigned long long)blockIdx.x*blockDim.x + threadIdx.x; i < n16; i += stride) { uint4 v = p[i]; acc ^= v.x^v.y^v.z^v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
int main(){
  const unsigned long long nbytes = 32ull<<30; uint8_t* d; CK(hipMalloc(&d, nbytes)); CK(hipMemset(d, 1, nbytes));
  unsigned long long* q; CK(hipMalloc(&q, 8)); unsigned* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned long long ntiles = nbytes / TILE;
  for (int v = 0; v < 3; ++v) for (int rep = 0; rep < 2; ++rep) { CK(hipMemset(q, 0, 8)); CK(hipEventRecord(e0));
    if (v == 0) hipLaunchKernelGGL(k_lane, dim3(256), dim3(512), 0, 0, d, ntiles, q, out);
    else if (v == 1) hipLaunchKernelGGL(k_quad, dim3(256), dim3(512), 0, 0, d, ntiles, q, out);
    else hipLaunchKernelGGL(k_coal, dim3(2048), dim3(512), 0, 0, d, nbytes, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    if (rep) printf("%s: %.2f ms for 32 GiB = %.0f GB/s\n", v==0?"per-lane 128B lines (scan pattern)":v==1?"quad-cooperative 64B pieces":"fully coalesced", ms, nbytes/ms/1e6); }
  return 0;
}

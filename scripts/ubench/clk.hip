// What do clock64() (s_memtime) and wall_clock64() (s_memrealtime) count on gfx950, and how long does a dependent VALU chain
// take in each? One wave spins for ~20 ms; a second launch fills the chip with the same loop (power -> clock).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *out, int iters) {
    unsigned x = threadIdx.x;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) x = __builtin_amdgcn_alignbit(x, x, 7) + 0x9e3779b9u;  // 2 dependent VALU ops
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = x; }
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 64);
    for (int blocks : {1, 1024, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 200000);
            hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            const double wall_s = h[1] / 100e6, instr = 200000.0 * 128;
            printf("blocks %4d: clock64 delta %llu wall ticks %llu -> clock64 rate %.1f MHz; %.3f ns per dependent VALU op = %.2f clock64 ticks\n",
                   blocks, h[0], h[1], h[0] / wall_s / 1e6, wall_s * 1e9 / instr, (double)h[0] / instr);
        }
    }
    return 0;
}

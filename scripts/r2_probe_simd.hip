// Probe (measurement aid, not product code): (1) where do the waves of a 512-thread / 137 KB-LDS workgroup land
// (SIMD id per wave index), (2) sustained shader clock and VALU issue rate of a dependent integer chain with one
// and two waves per SIMD on every CU. Build: hipcc --offload-arch=gfx950 -O3 r2_probe_simd.hip -o _build/probe_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_place(uint32_t *out) {
    __shared__ uint32_t big[137000 / 4];
    if (threadIdx.x == 0) big[blockIdx.x & 1023] = 1;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    // keep the workgroup resident for a while so that the next ones go to other CUs
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
    if (big[5] == 77) out[0] = 0;
}

// dependent chain shaped like SHA rounds: rotate (alignbit), xor3, add3 — `iters` x 8 VALU ops per lane
__global__ __launch_bounds__(64) void k_chain(uint32_t *out, uint64_t *clk, uint32_t iters, uint32_t seed) {
    extern __shared__ uint32_t pad[];
    uint32_t a = seed + threadIdx.x, b = a * 3u + 1u, c = a ^ 0x9e3779b9u;
    const uint64_t w0 = wall_clock64();
    const uint64_t c0 = clock64();
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a = __builtin_amdgcn_alignbit(a, a, 7) ^ b ^ c;   // v_alignbit + v_xor3
            b = b + a + c;                                     // v_add3
            c = __builtin_amdgcn_alignbit(c, c, 13) ^ a ^ b;  // v_alignbit + v_xor3
            a = a + b + c;                                     // v_add3
        }
    }
    const uint64_t c1 = clock64();
    const uint64_t w1 = wall_clock64();
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = c1 - c0;
        clk[blockIdx.x * 2 + 1] = w1 - w0;
    }
    if (a == 0x12345678u) out[0] = b + c;
    if (pad && a == 0x87654321u) out[1] = pad[0];
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clockRate %d kHz wall_clock_rate? (assumed 100 MHz)\n", prop.gcnArchName, cus, prop.clockRate);
    uint32_t *d_out;
    const int wgs = cus * 2;
    CK(hipMalloc(&d_out, (size_t)wgs * 8 * 2 * 4));
    CK(hipMemset(d_out, 0xff, (size_t)wgs * 8 * 2 * 4));
    hipLaunchKernelGGL(k_place, dim3(wgs), dim3(512), 0, 0, d_out);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h((size_t)wgs * 16);
    CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    int hist[8][4] = {};
    int balanced = 0, pairmode_distinct = 0;
    for (int g = 0; g < wgs; ++g) {
        int per_simd[4] = {}, first4[4] = {};
        for (int w = 0; w < 8; ++w) {
            const uint32_t hw = h[(size_t)(g * 8 + w) * 2];
            const int simd = (hw >> 4) & 3;
            hist[w][simd]++;
            per_simd[simd]++;
            if (w < 4) first4[simd]++;
        }
        if (per_simd[0] == 2 && per_simd[1] == 2 && per_simd[2] == 2 && per_simd[3] == 2) balanced++;
        if (first4[0] == 1 && first4[1] == 1 && first4[2] == 1 && first4[3] == 1) pairmode_distinct++;
    }
    printf("placement over %d workgroups of 8 waves: wave -> SIMD histogram\n", wgs);
    for (int w = 0; w < 8; ++w) printf("  wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("workgroups with 2 waves on every SIMD: %d / %d; with waves 0-3 on four distinct SIMDs: %d / %d\n", balanced, wgs, pairmode_distinct, wgs);
    // does (w&3) == simd hold, and do the dense roles (consumer 0,1,6,7 / producer 2,3,4,5) mix on every SIMD?
    int mixed = 0;
    for (int g = 0; g < wgs; ++g) {
        int cons[4] = {}, prod[4] = {};
        for (int w = 0; w < 8; ++w) {
            const int simd = (h[(size_t)(g * 8 + w) * 2] >> 4) & 3;
            if (w < 2 || w >= 6) cons[simd]++; else prod[simd]++;
        }
        bool ok = true;
        for (int s = 0; s < 4; ++s) ok = ok && cons[s] == 1 && prod[s] == 1;
        mixed += ok;
    }
    printf("workgroups where every SIMD hosts one consumer + one producer (dense roles): %d / %d\n", mixed, wgs);
    for (int w = 0; w < 8; ++w) printf("  wg0 wave %d hw_id %08x xcc %08x\n", w, h[w * 2], h[w * 2 + 1]);

    // issue rate + clock
    uint64_t *d_clk;
    CK(hipMalloc(&d_clk, (size_t)cus * 16 * 16));
    const uint32_t iters = 400000;  // x16 VALU ops
    for (int per_simd = 1; per_simd <= 4; per_simd *= 2) {
        for (int frac = 0; frac < 2; ++frac) {  // all CUs / a quarter of the CUs
            const int use_cus = frac ? cus / 4 : cus;
            const int n = use_cus * 4 * per_simd;
            const size_t lds = (160u << 10) / (4 * per_simd) - 1024;  // caps residency at 4*per_simd single-wave workgroups per CU
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(k_chain, dim3(n), dim3(64), lds, 0, d_out, d_clk, 1000u, 1u);  // warm
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_chain, dim3(n), dim3(64), lds, 0, d_out, d_clk, iters, 1u);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> c((size_t)n * 2);
            CK(hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost));
            double sc = 0, sw = 0;
            for (int i = 0; i < n; ++i) { sc += (double)c[i * 2]; sw += (double)c[i * 2 + 1]; }
            const double ops = (double)iters * 16.0;
            const double wall_s = sw / n / 100e6;
            printf("chain: %d wave(s)/SIMD on %d CUs: %.2f ms; per wave: clock64 ticks/op %.3f, wall ns/op %.4f, clock64 rate %.1f MHz; chip %.2f T lane-ops/s\n",
                   per_simd, use_cus, ms, sc / n / ops, wall_s * 1e9 / ops, sc / sw * 100.0, (double)n * 64.0 * ops / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

// VALU issue rate on this GPU: waves x instructions / time, for 32-bit integer ops at full occupancy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OCC>
__global__ __launch_bounds__(256) void k_valu(unsigned* out, int iters) {
    unsigned a = threadIdx.x, b = blockIdx.x, c = a ^ 0x55u, d = b + 7u, e = a * 3u, f = b ^ 0x1234u, g = a + b, h = a - b;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            a = (a ^ b) + 0x9E3779B9u; b = (b ^ c) + 0x7F4A7C15u; c = (c ^ d) + 0x85EBCA6Bu; d = (d ^ e) + 0xC2B2AE35u;
            e = (e ^ f) + 0x27D4EB2Fu; f = (f ^ g) + 0x165667B1u; g = (g ^ h) + 0xD3A2646Cu; h = (h ^ a) + 0xFD7046C5u;
        }
    }
    unsigned r = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
    if (r == 0x12345678u) out[0] = r;
}
int main() {
    unsigned* out; (void)hipMalloc(&out, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 200;
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        dim3 grid(256 * wg_per_cu * 4);
        float best = 1e9;
        for (int r = 0; r < 4; r++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_valu<0>, grid, dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        double instr = (double)grid.x * 4 /*waves*/ * iters * 16 * 16 /*2 ops x 8*/;
        printf("grid %6u: %.3f ms  %.0f G wave-instr/s\n", grid.x, best, instr / best / 1e6);
    }
    return 0;
}

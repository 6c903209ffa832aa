// Issue cost of individual gfx950 VALU instructions (inline asm, 8 independent chains, full occupancy),
// reported relative to v_and_b32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned u32;
typedef unsigned long long u64;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM32)                                                                              \
    __global__ __launch_bounds__(256) void NAME(u32* out, int iters) {                                  \
        u32 r[8];                                                                                       \
        for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 2654435761u + i;                               \
        u32 s = threadIdx.x | 1;                                                                        \
        for (int it = 0; it < iters; it++) {                                                            \
            _Pragma("unroll") for (int k = 0; k < 4; k++) {                                              \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM32 : "+v"(r[i]) : "v"(s)); } \
            }                                                                                           \
        }                                                                                               \
        u32 x = 0;                                                                                      \
        for (int i = 0; i < 8; i++) x ^= r[i];                                                          \
        if (x == 0x12345678u) out[0] = x;                                                               \
    }
#define KERNEL64(NAME, ASM64)                                                                            \
    __global__ __launch_bounds__(256) void NAME(u32* out, int iters) {                                  \
        u64 r[8];                                                                                       \
        for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 2654435761ull + i;                             \
        u32 s = (threadIdx.x & 31) | 1;                                                                 \
        for (int it = 0; it < iters; it++) {                                                            \
            _Pragma("unroll") for (int k = 0; k < 4; k++) {                                              \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM64 : "+v"(r[i]) : "v"(s)); } \
            }                                                                                           \
        }                                                                                               \
        u64 x = 0;                                                                                      \
        for (int i = 0; i < 8; i++) x ^= r[i];                                                          \
        if (x == 0x12345678ull) out[0] = (u32)x;                                                        \
    }

KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_lshr, "v_lshrrev_b32 %0, %1, %0")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 31")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(k_ffbl, "v_ffbl_b32 %0, %0")
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %1")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %1")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %1")
KERNEL(k_min, "v_min_u32 %0, %0, %1")
KERNEL(k_mul, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_cmp_cnd, "v_cmp_ne_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp_s, "v_cmp_ne_u32 s[20:21], %0, %1")
KERNEL(k_movdpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_adddpp, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 5")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %1, %0")
KERNEL64(k_lshl64, "v_lshlrev_b64 %0, 1, %0")
KERNEL64(k_lshl64v, "v_lshlrev_b64 %0, %1, %0")
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, %1, %0")
KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 1, %0")
KERNEL64(k_mov64, "v_mov_b64 %0, %0")

typedef void (*kern_t)(u32*, int);
struct Entry { const char* name; kern_t k; int n; };

int main() {
    u32* out; (void)hipMalloc(&out, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    Entry es[] = {
        {"v_and_b32", k_and, 1}, {"v_xor_b32", k_xor, 1}, {"v_add_u32", k_add, 1}, {"v_lshlrev_b32 const", k_lshl, 1},
        {"v_lshrrev_b32 var", k_lshr, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_perm_b32", k_perm, 1},
        {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_ffbl_b32", k_ffbl, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_and_or_b32", k_and_or, 1}, {"v_or3_b32", k_or3, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_lshl_add_u32", k_lshl_add, 1},
        {"v_xad_u32", k_xad, 1}, {"v_add3_u32", k_add3, 1}, {"v_min_u32", k_min, 1}, {"v_mul_lo_u32", k_mul, 1},
        {"v_cmp + v_cndmask (pair)", k_cmp_cnd, 1}, {"v_cmp -> sgpr", k_cmp_s, 1}, {"v_mov_b32_dpp", k_movdpp, 1},
        {"v_add_u32_dpp", k_adddpp, 1}, {"ds_bpermute + wait", k_bpermute, 1}, {"v_readlane_b32", k_readlane, 1},
        {"v_mbcnt_lo", k_mbcnt, 1}, {"v_lshlrev_b64 const", k_lshl64, 1}, {"v_lshlrev_b64 var", k_lshl64v, 1},
        {"v_lshrrev_b64 var", k_lshr64, 1}, {"v_lshl_add_u64", k_lshladd64, 1}, {"v_mov_b64", k_mov64, 1}};
    float base = 0;
    const int iters = 100;
    for (auto& e : es) {
        float best = 1e9;
        for (int r = 0; r < 3; r++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(256 * 8 * 2), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        if (base == 0) base = best;
        double instr = 256.0 * 8 * 2 * 4 * iters * 32;
        printf("%-28s %.3f ms  %6.0f G wave-instr/s  x%.2f\n", e.name, best, instr / best / 1e6, best / base);
    }
    return 0;
}

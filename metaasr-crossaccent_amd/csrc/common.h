// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libmasr.
// Wave = 64 lanes.  One MFMA primitive is used everywhere:
//   v_mfma_f32_16x16x32_bf16   D[16x16] += A[16x32] * B[32x16]
//   A fragment: lane l holds A[row = l&15][k = 8*(l>>4) + j], j = 0..7
//   B fragment: lane l holds B[k = 8*(l>>4) + j][col = l&15]
//   C/D       : lane l, reg r  ->  row = 4*(l>>4) + r, col = l&15
// so every operand is staged in LDS as [row-or-col][k contiguous] and read
// with one 16-byte ds_read per fragment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MASR_WAVE 64

__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}

__device__ __forceinline__ bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }

// one LDS-DMA instruction (global_load_lds_dwordx4: lane l's 16 bytes land at lds_dst + 16 l; lds_dst = LDS byte address, wave-uniform),
// written in asm so that the compiler does not know an LDS write is pending: behind a __builtin_amdgcn_global_load_lds it puts
// `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 (not in front of plain ds_read_b128), which drains a DMA ring at the first
// transposing fragment read of each step.  The kernel's own counted s_waitcnt vmcnt(N) + barrier must then order every LDS read behind
// the DMA that feeds it; the compiler's counted waits for its own loads can only over-wait beside these (vmcnt retires in order).
__device__ __forceinline__ void glds16(const void* g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {            // LDS byte address of a pointer into __shared__ (wave-uniform p)
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const void*)p);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Stateless counter-based RNG for dropout: the same (seed, site, index) gives the same
// bit in forward and backward, so no mask is ever stored.
__device__ __forceinline__ uint32_t hash_u32(uint32_t seed, uint32_t site, uint32_t idx) {
    uint32_t x = idx * 0x9E3779B1u ^ (seed + site * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// keep-scale: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float dropout_scale(uint32_t seed, uint32_t site, uint32_t idx, float p, float inv_keep) {
    // 24-bit uniform u = (h >> 8) / 2^24 in [0,1); dropped where u < p.  Compared as integers: u and p 2^24 are exact in fp32, so
    // u < p  <=>  (h >> 8) < ceil(p 2^24) -- the threshold is loop-invariant, which leaves a shift and a compare per element
    const uint32_t thr = (uint32_t)ceilf(p * 16777216.0f);
    return (hash_u32(seed, site, idx) >> 8) < thr ? 0.0f : inv_keep;
}

#define HIP_CHECK_RET(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) { mk_set_error(#expr, hipGetErrorString(_e)); return -1; } \
    } while (0)

void mk_set_error(const char* what, const char* detail);

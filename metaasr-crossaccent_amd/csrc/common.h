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

// Stateless counter-based RNG for dropout: the same (seed, site, index) gives the same bit in forward and backward, so no mask is ever
// stored (and a resumed run regenerates the masks of the step it continues from).  Round 6: ONE 32-bit hash word decides the element PAIR
// (2k, 2k + 1) -- its low / high 16 bits against a 16-bit threshold -- and the (seed, site) part is mixed once per kernel into a key that
// is XORed into the pair index: two 32-bit multiplies per pair where round 5 paid three per element (v_mul_lo_u32 is a quarter-rate
// instruction; the decision was a third of the encoder attention forward and ~6 us of FFN1's epilogue).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {                  // "lowbias32" integer finaliser (a bijection of 32-bit words)
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t dropout_key(uint32_t seed, uint32_t site) { return mix32(seed * 0x9E3779B1u + site * 0x85EBCA77u + 0x632BE5ABu); }   // loop-invariant
// dropped where u16 < ceil(p 2^16): the drop probability is p rounded UP to a multiple of 2^-16 (p = 0.1: 0.100006)
__device__ __forceinline__ uint32_t dropout_thr(float p) { return (uint32_t)ceilf(p * 65536.0f); }
__device__ __forceinline__ uint32_t dropout_word(uint32_t key, uint32_t pair) { return mix32(pair ^ key); }
// keep-scale of ONE element: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float dropout_scale(uint32_t seed, uint32_t site, uint32_t idx, float p, float inv_keep) {
    const uint32_t w = dropout_word(dropout_key(seed, site), idx >> 1);
    return ((idx & 1u) ? (w >> 16) : (w & 0xFFFFu)) < dropout_thr(p) ? 0.0f : inv_keep;
}
// ... of the FOUR consecutive elements idx0 .. idx0 + 3: two hash words when idx0 is even, a third one when it is odd (the elements then
// straddle three pairs; the halves are lined up with two funnel shifts)
__device__ __forceinline__ void dropout_scale4(uint32_t seed, uint32_t site, uint32_t idx0, float p, float inv_keep, float (&s)[4]) {
    const uint32_t key = dropout_key(seed, site), thr = dropout_thr(p), p0 = idx0 >> 1;
    uint32_t a = dropout_word(key, p0), b = dropout_word(key, p0 + 1);
    if (idx0 & 1u) {
        const uint32_t c = dropout_word(key, p0 + 2);
        a = __builtin_amdgcn_alignbit(b, a, 16);                         // [a.hi, b.lo]
        b = __builtin_amdgcn_alignbit(c, b, 16);                         // [b.hi, c.lo]
    }
    s[0] = (a & 0xFFFFu) < thr ? 0.0f : inv_keep; s[1] = (a >> 16) < thr ? 0.0f : inv_keep;
    s[2] = (b & 0xFFFFu) < thr ? 0.0f : inv_keep; s[3] = (b >> 16) < thr ? 0.0f : inv_keep;
}

#define HIP_CHECK_RET(expr)                                                        \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) { mk_set_error(#expr, hipGetErrorString(_e)); return -1; } \
    } while (0)

void mk_set_error(const char* what, const char* detail);

// VGG front-end of MyTransformer.extract_feat
// (reference: src/model/transformer_pytorch/mono_transformer_torch.py:49-60,113-116 --
//  Conv2d(1,64,3,p1)+ReLU, Conv2d(64,64)+ReLU, MaxPool2d(2,2), Conv2d(64,128)+ReLU,
//  Conv2d(128,128)+ReLU, MaxPool2d(2,2); ~71 % of the path's FLOPs).
//
// MI355X layout: activations are NHWC bf16 ([B][T][D][C], channels innermost) so that the
// 3x3 convolutions become implicit GEMMs whose K index (tap, ci) is contiguous in memory:
//   out[p][co] = sum_{tap,ci} in[p + off(tap)][ci] * wk[co][tap*CIN + ci]
// with M = B*T*D pixels, N = COUT, K = 9*CIN on v_mfma_f32_16x16x32_bf16.
// The same kernel computes dgrad (input = dY, weights = 180-degree-rotated/transposed shadow).
// wgrad is the reduction-major product dW[co][tap,ci] = sum_p dY[p][co] * in[p+off][ci]
// (transposing LDS reads), split over pixel ranges with a deterministic slab reduce.
#include "common.h"
#include "kernels.h"
#include "folds.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 40;

// ConvArgs::pool_idx codes of one 2 x 2 window, two channels per register: v00 v01 / v10 v11 are bit patterns of NON-NEGATIVE
// bf16 values (they order like 15-bit integers, differences cannot overflow).  Position of the first maximum in row-major scan
// order (strict >, as torch's max_pool2d backward), 4 where the maximum x is 0.  There is no packed 16-bit compare: a > b is the
// sign of b - a (v_pk_sub_i16 + a shift), the selection is a v_bfi_b32.  The code of each channel lands in the low byte of
// its 16-bit half; pool_pack4 gathers four of them into a word.  This runs on the MFMA waves' own time: every instruction counts.
typedef __attribute__((ext_vector_type(2))) short s16x2;
__device__ __forceinline__ unsigned pool_code2(s16x2 v00, s16x2 v01, s16x2 v10, s16x2 v11, s16x2 m01, s16x2 m23, s16x2 x) {
    // (written out: from vector C the compiler goes back to 16-bit scalar compares + v_cndmask through SDWA, twice the instructions)
    unsigned s01, s23, lower, zr, code;
    asm("v_pk_sub_i16 %0, %5, %6\n\t"                       // < 0 where v01 > v00
        "v_pk_sub_i16 %1, %7, %8\n\t"                       // < 0 where v11 > v10
        "v_pk_sub_i16 %2, %9, %10\n\t"                      // < 0 where the lower row holds the larger maximum
        "v_pk_add_u16 %3, %11, -1 op_sel_hi:[1,0]\n\t"      // 0xffff where x == 0
        "v_pk_lshrrev_b16 %0, 15, %0 op_sel_hi:[0,1]\n\t"   // 0 / 1: winner of the upper row
        "v_pk_lshrrev_b16 %1, 15, %1 op_sel_hi:[0,1]\n\t"
        "v_pk_ashrrev_i16 %2, 15, %2 op_sel_hi:[0,1]\n\t"   // -1 / 0
        "v_pk_lshrrev_b16 %3, 15, %3 op_sel_hi:[0,1]\n\t"   // 1 where x == 0 (then every difference above is 0: sel = 0)
        "v_or_b32 %1, 0x20002, %1\n\t"                      // 2 / 3: winner of the lower row
        "v_bfi_b32 %4, %2, %1, %0\n\t"
        "v_lshl_or_b32 %4, %3, 2, %4"
        : "=&v"(s01), "=&v"(s23), "=&v"(lower), "=&v"(zr), "=&v"(code)
        : "v"(v00), "v"(v01), "v"(v10), "v"(v11), "v"(m01), "v"(m23), "v"(x));
    return code;
}
__device__ __forceinline__ unsigned pool_pack4(unsigned c01, unsigned c23) { return __builtin_amdgcn_perm(c23, c01, 0x06040200u); }
// lane ^ 1 and lane ^ 8 as DPP moves (quad_perm [1,0,3,2]; rotation by 8 within the row of 16) instead of LDS permutes
__device__ __forceinline__ s16x2 lane_xor1(s16x2 v) { return __builtin_bit_cast(s16x2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)); }
__device__ __forceinline__ s16x2 lane_xor8(s16x2 v) { return __builtin_bit_cast(s16x2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false)); }

// ------------------------------------------------------------------ conv1 (CIN = 1), direct fp32
// bits (optional): the ReLU mask of this map as one 64-bit word per pixel [B][H][W], bit c = (out[..][c] > 0) -- all the masked
// dgrad of the NEXT conv needs of the map (conv3x3_resw_w1x_kernel, ConvArgs::mask_bits): 8 bytes per pixel instead of 128, and half
// the vector-memory instructions that launch is bound by.  Here it costs ONE more store instruction per 16-pixel segment: a lane owns the
// 16 channels 16 q .. 16 q + 15 of its pixel, i.e. one 16-bit piece of the pixel's word.
// (Tried instead: the patch wave of conv2's forward launch derives the words from the patches it streams -- 32 LDS reads + ~500
// VALU per tile on a wave that shares its SIMD with an MFMA wave: conv2 forward 93 -> 133 us.)
// Round 6: the CIN = 1 layer on the fp32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fmaf chain -- the precision policy keeps
// this layer in fp32).  A 16-pixel row segment is ONE product  D^T[cout][pixel] = W[cout][tap] X^T[tap][pixel]  per 16 output channels: four
// channel blocks x three tap groups (taps 0-3, 4-7, 8 + zeros) = 12 MFMAs, the bias as the C operand of the first -- the summation order
// (bias, tap 0 .. 8) is that of the vector-ALU kernel this replaces, so the bits are too.  The rows of block j are ASSIGNED to channels
// 16 (i >> 2) + 4 j + (i & 3): a lane (pixel p = lane & 15, q = lane >> 4) then ends with the 16 consecutive channels 16 q .. 16 q + 15 of its
// pixel = two 16-byte stores and one 16-bit piece of the pixel's ReLU sign word.  The window (3 x 18 floats) is one load per segment, the taps
// come by lane permutes.
// What bounded the launch (55-60 us for 164 MB of output, with the vector ALU or the MFMA alike) was not arithmetic: vector-memory operations
// retire in order, so a segment's window load waited for the write acknowledgements of the previous segment's stores -- one store round trip +
// one load round trip per segment and wave.  Here every access is a BUFFER access whose out-of-image lanes carry an offset past the
// descriptor's end (loads return 0 = the zero padding, stores are dropped): no branch around a memory instruction, so the compiler counts its
// waits, and the window of the NEXT segment is requested before the current one's stores.
__global__ __launch_bounds__(256) void conv1_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, bf16* __restrict__ out,
                                                             int B, int H, int W, int cstride, unsigned short* __restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int p = lane & 15, q = lane >> 4;
    const unsigned npix = (unsigned)B * H * W, OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, npix * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out, 0, npix * (unsigned)cstride * 2u, 0x00020000);     // (< 4 GB: checked by the launcher)
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(bits, 0, bits ? npix * 8u : 0u, 0x00020000);
    float wa[4][3];
    f32x4 bi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = 16 * (p >> 2) + 4 * j + (p & 3);           // A operand: row i = lane & 15, k = lane >> 4
#pragma unroll
        for (int g = 0; g < 3; ++g) { const int tap = 4 * g + q; wa[j][g] = tap < 9 ? w[co * 9 + tap] : 0.f; }
#pragma unroll
        for (int r = 0; r < 4; ++r) bi[j][r] = bias[16 * q + 4 * j + r];      // C / D: rows 4 q + r of block j
    }
    int src[3];                                                   // window element of tap 4 g + q for pixel p (tap >= 9: the centre, times a zero weight)
#pragma unroll
    for (int g = 0; g < 3; ++g) { const int tap = 4 * g + q < 9 ? 4 * g + q : 4; src[g] = (tap / 3) * 18 + p + tap % 3; }
    const int nseg = (W + 15) / 16, total = B * H * nseg;
    const int wrow = lane / 18, wcol = lane % 18;
    // every wave takes ONE contiguous run of segments and steps (row of the [B * H] row list, frame t, segment sx of the row) by addition:
    // a strided walk paid two integer divisions per segment on the scalar unit the CU's four SIMDs share
    const int nwaves = gridDim.x * 4, per = (total + nwaves - 1) / nwaves;
    const int first = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * per);
    const int last = first + per < total ? first + per : total;
    struct Seg { int sg, row, t, sx; };
    auto advance = [&](Seg& g) {
        ++g.sg;
        if (++g.sx == nseg) { g.sx = 0; ++g.row; if (++g.t == H) g.t = 0; }
    };
    const int woff = (wrow - 1) * W + wcol - 1;                   // this lane's window element relative to (row, d0)
    const unsigned lane_oo = ((unsigned)p * cstride + 16u * q) * 2u, lane_bo = (unsigned)p * 8u + 2u * q;
    auto window = [&](const Seg& g) -> float {                    // 0 outside the image / the run (buffer load past the descriptor's end)
        const int tt = g.t + wrow - 1, dd = g.sx * 16 + wcol - 1;
        const bool ok = g.sg < last && lane < 54 && (unsigned)tt < (unsigned)H && (unsigned)dd < (unsigned)W;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? (unsigned)(g.row * W + g.sx * 16 + woff) * 4u : OOB, 0, 0));
    };
    Seg cur;
    cur.sg = first; cur.row = first / nseg; cur.sx = first - cur.row * nseg; cur.t = cur.row % H;
    float wv = window(cur);
    // (the first window is waited for HERE: merged with the loop's back edge -- window complete, three stores pending -- a pending load at
    // the loop's entry makes the compiler wait vmcnt(0) at the top of every iteration)
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
    while (cur.sg < last) {
        Seg nxt = cur;
        advance(nxt);
        const float wnext = window(nxt);                          // in flight under this segment's MFMAs and in FRONT of its stores
        float xb[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) xb[g] = __shfl(wv, src[g]);
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][0], xb[0], bi[j], 0, 0, 0);
#pragma unroll
        for (int g = 1; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][g], xb[g], acc[j], 0, 0, 0);
        // bf16 first (v_cvt_pk_bf16_f32), ReLU on the packed halves (a negative bf16 is a negative 16-bit integer: v_pk_max_i16 with 0), sign
        // flags as v_pk_min_u16(half, 1): 8 + 8 + 8 + 8 instructions (asm: hipcc scalarises the packed forms of this chain)
        typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
        u4_t o[2];
        unsigned e = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned dw, fl;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dw) : "v"(acc[j][2 * h]), "v"(acc[j][2 * h + 1]));
                asm("v_pk_max_i16 %0, %1, %2" : "=v"(dw) : "v"(dw), "s"(0u));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(fl) : "v"(dw), "s"(0x00010001u));
                o[j >> 1][2 * (j & 1) + h] = dw;                  // channels 16 q + 4 j + 2 h, + 1
                e |= fl << (4 * j + 2 * h);                       // flags at bits 4 j + 2 h and 16 + 4 j + 2 h
            }
        e = (e | e >> 15) & 0xFFFFu;                              // bit c = channel 16 q + c passed the ReLU
        // addresses without a vector multiply (the compiler branches around one): segment base on the scalar unit + this lane's fixed offset
        const unsigned segpix = (unsigned)(cur.row * W + cur.sx * 16);
        const bool in = cur.sx * 16 + p < W;
        const unsigned oo = in ? segpix * (unsigned)cstride * 2u + lane_oo : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(o[0], ro, oo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o[1], ro, oo, 16, 0);
        // (a null `bits` has an empty descriptor: the store is dropped)
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)e, rb, in ? segpix * 8u + lane_bo : OOB, 0, 0);
        wv = wnext;
        cur = nxt;
    }
}

constexpr int C1_PIX = 512;    // pixels per block in conv1 wgrad (2048 until round 5: the BLSTM bench shape then ran on 130 workgroups, half the CUs with one each -- 77 us per launch)
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const bf16* __restrict__ dy,
                                                          float* __restrict__ slab, int B, int H, int W, int cstride) {
    const long P = (long)B * H * W;
    const long p0 = (long)blockIdx.x * C1_PIX;
    const int pl = threadIdx.x >> 3, cg = threadIdx.x & 7;
    float acc[8][10];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[j][k] = 0.f;
    for (int it = 0; it < C1_PIX / 32; ++it) {
        const long p = p0 + it * 32 + pl;
        if (p >= P) break;
        const int d = (int)(p % W);
        const int t = (int)((p / W) % H);
        float xv[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tt = t + tap / 3 - 1, dd = d + tap % 3 - 1;
            xv[tap] = (tt >= 0 && tt < H && dd >= 0 && dd < W) ? x[p + (long)(tap / 3 - 1) * W + (tap % 3 - 1)] : 0.f;
        }
        const bf16x8 g = ld8(dy + p * cstride + cg * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gj = (float)g[j];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[j][tap] = fmaf(gj, xv[tap], acc[j][tap]);
            acc[j][9] += gj;
        }
    }
    // reduce over the 8 pixel-lanes of a wave that share cg (lane bits 3..5), then over the 4 waves
    __shared__ float red[4][8][80];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            float v = acc[j][k];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 8) red[wave][lane][j * 10 + k] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 640; i += 256) {
        const int c = i / 80, e = i % 80;
        const float v = red[0][c][e] + red[1][c][e] + red[2][c][e] + red[3][c][e];
        // slab[block][co = c*8 + e/10][k = e%10]
        slab[(long)blockIdx.x * 640 + (c * 8 + e / 10) * 10 + e % 10] = v;
    }
}
// workgroup = 16 consecutive outputs x 16 strided walks over a range of the per-block partials (64-byte row segments
// instead of one float per cache line), then a fixed-order fold in LDS -> deterministic.  Two passes: C1_RSPLIT row
// ranges in parallel into `part` (a single pass of 40 workgroups spent 31 us on 13 MB: latency-bound), then those rows.
constexpr int C1_RSPLIT = 16;
__global__ __launch_bounds__(256) void conv1_wgrad_reduce(const float* __restrict__ slab, int nblocks, float* __restrict__ dw,
                                                          float* __restrict__ db, float* __restrict__ part_out) {
    conv1_wgrad_reduce_body(slab, nblocks, dw, db, part_out, blockIdx.x, blockIdx.y, gridDim.y);
}
// slab rows [nblocks, nblocks + C1_RSPLIT) hold the first pass's partials
static void launch_conv1_reduce(float* slab, int nb, float* dw, float* db, hipStream_t s) {
    float* part = slab + (long)nb * 640;
    hipLaunchKernelGGL(conv1_wgrad_reduce, dim3(40, C1_RSPLIT), dim3(256), 0, s, slab, nb, nullptr, nullptr, part);
    hipLaunchKernelGGL(conv1_wgrad_reduce, dim3(40, 1), dim3(256), 0, s, part, C1_RSPLIT, dw, db, nullptr);
}

// fragment of a reduction-major LDS tile [k][m] (row stride LDT): 32 k-rows x columns r0..r0+15 through two transposing reads
template <int LDT>
__device__ __forceinline__ bf16x8 frag_rm(const bf16* tile, int r0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16* a0 = tile + (4 * g + q) * LDT + r0 + 4 * p;
    const bf16* a1 = a0 + 16 * LDT;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// ------------------------------------------------------------------ streaming 3x3 (fwd and dgrad), v3
// Phase timing of the patch-per-workgroup kernel this replaced (one TH x 16 tile per workgroup, patch in LDS once, weights through registers;
// deleted in round 6 with its last caller) showed a workgroup spending a quarter of its life waiting for its input patch, a quarter pushing
// its output tile out and a third in the MFMA loop, with the one other workgroup on the CU rarely in the complementary phase.  This kernel
// keeps that tile geometry and MFMA loop and takes the memory phases off the MFMA waves' timeline:
//   * persistent workgroups (one per CU) walk the tiles; a stage = (tile, 64-channel input slab)
//   * 6 waves: waves 0-3 only read LDS, issue MFMAs and store their accumulators; wave 4 streams the per-tap weight slices
//     and wave 5 the input patches, both by LDS-DMA (global_load_lds, no VGPR staging).  vmcnt is per wave and counts
//     in issue order, so giving each stream its own wave is what lets the next PATCH stay in flight for a whole stage
//     (9 taps) while the weight slices are waited on tap by tap, and keeps the MFMA waves' own stores out of both waits.
//   * one s_barrier per tap is the only synchronisation: passing barrier u means "tap u's weights (and, at tap 0, the
//     stage's patch) have landed" and "everybody is done with tap u-1's buffers".
//   * LDS images are unpadded 128-byte rows (a DMA wave-instruction writes 1 KiB linearly); the bank-conflict fix is an
//     XOR of the 16-byte chunk index with (patch column & 7) / (weight row & 7), applied to the DMA source address and to
//     the fragment reads.  Halo pixels outside the image are fetched from a zero line.
//   * weight rows are permuted on their way into LDS so that a lane's accumulators of two adjacent 16-channel blocks are 8
//     CONSECUTIVE output channels: the epilogue is bias/ReLU in registers and one 16-byte global store per (pixel
//     tile, 32-channel group) -- no LDS round trip, no barrier; the 2x2 max-pool is taken across registers and lanes.
__device__ __attribute__((aligned(128))) bf16 g_zero_line[64];

// MaxPool2d(2,2) + ReLU backward of ONE pooled cell x 8 channels, as the patch producers of the UNPOOL dgrad flavours need it: g = the
// cell's gradient (4 dwords of 2 bf16), code = its 8 pool codes (ConvArgs::pool_idx: window position 0..3 of the first maximum, 4 = nothing
// passed the ReLU); o[k] = the 8 channels of window position k (row-major), the gradient where code == k, else 0.  0x80 - (code ^ k) has bit
// 7 of a byte set exactly where the codes match (codes are 0..4: no borrow crosses a byte); v_perm moves the two bytes of a channel pair
// into the high bytes of the 16-bit halves and an arithmetic shift makes them 0xffff / 0 masks: 16 vector instructions per position.
__device__ __forceinline__ void unpool_expand(__attribute__((ext_vector_type(4))) unsigned g, uint2 code, __attribute__((ext_vector_type(4))) unsigned (&o)[4]) {
    typedef __attribute__((ext_vector_type(2))) short sh2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned kx = 0x01010101u * (unsigned)k;
        const unsigned e0 = 0x80808080u - (code.x ^ kx), e1 = 0x80808080u - (code.y ^ kx);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const unsigned e = kk < 2 ? e0 : e1;
            const unsigned m = __builtin_amdgcn_perm(e, e, (kk & 1) ? 0x030C020Cu : 0x010C000Cu);
            const sh2 mm = __builtin_bit_cast(sh2, m) >> 15;
            o[k][kk] = g[kk] & __builtin_bit_cast(unsigned, mm);
        }
    }
}
__device__ unsigned g_conv_sched[2];                      // tile counter of launches that bring none (single-stream tools and tests)

// MASK = 0: forward flavour (bias, ReLU flag, optional fused pool); MASK = 1: dgrad flavour (outputs zeroed where a.mask <= 0;
// no bias / ReLU / pool); MASK = 2: the same with the mask read as SIGN BITS (a.mask_bits: four dwords per pixel, dword q = the
// bytes of channel groups 8q.., 32+8q.., 64+8q.., 96+8q.. -- exactly what lane (rr, q) masks, so ONE 4-byte load per pixel where the
// bf16 map costs four 16-byte ones, and 4 mask registers instead of 64: room for the 32-row tiles of the unmasked kernels).
// The forward flavour with 128 output channels writes those words for ITS output when a.out_sign_bits is set.
// UNPOOL (dgrad behind a max-pool): the input map is never materialised -- three producer waves (5, 6, 7; one per SIMD beside an MFMA
// wave, the fourth SIMD hosts the weight wave) build each stage's patch from the POOLED gradient a.in_pooled [B][H/2][W/2][CIN] and the
// one-byte pool codes a.in_idx (ConvArgs::pool_idx of the forward launch), see unpool_expand.  512 threads.
template <int CIN, int COUT, int TH, int TW, int MASK, bool UNPOOL = false>
__global__ __launch_bounds__(UNPOOL ? 512 : 384) void conv3x3_stream_kernel(ConvArgs a, int ntiles, int tiles_x, int tiles_y) {
    constexpr int PW = TW + 2, PH = TH + 2, RPT = 16 / TW;
    constexpr int MF = TH * TW / 64, NF = COUT / 16, NH = NF / 2;
    constexpr int KTOT = 9 * CIN, NSLAB = CIN / 64;
    constexpr int D = 4;                                   // weight ring: slices u+1 (being read ahead), u, and two in flight
    constexpr int PPIECES = (PH * PW + 7) / 8;             // 1 KiB DMA pieces per patch (8 pixels x 128 B)
    constexpr int PCHUNK = (PPIECES + 5) / 6;              // the next patch is issued over taps 0..5, PCHUNK pieces after each barrier
    constexpr int PBYTES = PPIECES * 1024, WPIECES = COUT / 8, WBYTES = COUT * 128;
    __shared__ __attribute__((aligned(1024))) char lds[2 * PBYTES + D * WBYTES];
    __shared__ int tileq[4];                               // tile ids of this workgroup's k-th, k+1-th, ... tile (-1 = none)
    __shared__ __attribute__((aligned(16))) float sbias[COUT];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    typedef __attribute__((ext_vector_type(2))) short short2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    char* const pbuf = lds;
    char* const wbuf = lds + 2 * PBYTES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, W = a.W;
    // Tiles are handed out by an atomic counter (a.sched[0]), not by a static stride: under concurrent streams some of the
    // "resident" workgroups start late, and with a static split the launch would last until those had walked their full
    // share.  The patch wave fetches two tiles ahead and publishes the ids through tileq; every wave derives the same
    // sequence of stages = (tile, 64-channel slab) from it.  The last workgroup to run dry re-arms the counter.
    // Barrier protocol (every wave executes 1 + 9 x stages barriers): passing the opening barrier means slice 0, patch 0 and
    // tileq[0..1] are in; passing barrier u (in the MIDDLE of tap u) means slice u+1 -- and at the last tap of a stage the next
    // stage's patch -- have landed, and that every MFMA wave is done with tap u-1's slice (and, at the first tap of a
    // stage, with the previous stage's patch).  The MFMA waves read the fragments of the next half-tap before issuing the
    // MFMAs of the current one, so the barrier wait and the LDS latency sit under 16 MFMAs already in the pipe.
    // (an LDS-qualified volatile read: through a generic `volatile int*` hipcc emits flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0), i.e.
    // the read queues at the texture addresser behind the patch wave's DMA pieces and waits for all of this wave's stores)
    auto read_tileq = [&](int k) { return __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)&tileq[k & 3]); };

    if (wave == 4) {
        // ---------------- weight stream: slice (slab, tap) = [COUT rows][64 ci], LDS row r holds channel perm(r).
        // Slices do not depend on the tile, so the stream simply runs three ahead and is drained at the end.
        const int sub = lane >> 3, sl = lane & 7;
        const bf16* wsrc[WPIECES];
#pragma unroll
        for (int i = 0; i < WPIECES; ++i) {
            const int r = i * 8 + sub, j = r >> 4, m = r & 15;
            const int co = (j >> 1) * 32 + (m >> 2) * 8 + (j & 1) * 4 + (m & 3);
            wsrc[i] = a.wk + (long)co * KTOT + ((sl ^ (r & 7)) * 8);
        }
        int is = 0, it = 0, islot = 0;
        auto issue_next = [&]() {
            const int off = it * CIN + is * 64;
#pragma unroll
            for (int i = 0; i < WPIECES; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(wsrc[i] + off), (lptr_t*)(wbuf + islot * WBYTES + i * 1024), 16, 0, 0);
            islot = (islot + 1) & (D - 1);
            if (++it == 9) { it = 0; if (++is == NSLAB) is = 0; }
        };
        for (int k = 0; k < D - 1; ++k) issue_next();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WPIECES) : "memory");
        asm volatile("s_barrier" ::: "memory");           // opening barrier: slice 0 is in
        for (int k = 0; read_tileq(k) >= 0; ++k) {
            for (int u = 0; u < 9 * NSLAB; ++u) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPIECES) : "memory");   // slice u+1 is in, u+2 may be in flight
                asm volatile("s_barrier" ::: "memory");
                issue_next();                              // slice u+3 -> the slot of tap u-1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the read-ahead past the last tile lands before the LDS is released
        return;
    }
    if constexpr (UNPOOL) {
    if (wave >= 5) {
        // ---------------- patch producers.  Item = (pooled cell, 16-byte channel chunk): 16 B of gradient + 8 B of codes in, the cell's
        // (up to) four patch pixels out.  The patch origin (t0 - 1, d0 - 1) is odd, so the patch covers PH / 2 + 1 pooled rows and
        // PW / 2 + 1 pooled columns, the first and the last of them with one row / column of their window.  Items are dealt to the
        // producers round-robin in groups of 64; everything that does not depend on the tile is computed once.
        constexpr int NPROD = 3, NPR = PH / 2 + 1, NPC = PW / 2 + 1, NITEM = NPR * NPC * 8, NIT = (NITEM + 64 * NPROD - 1) / (64 * NPROD);
        static_assert(NIT <= 7, "the items of a patch are expanded during taps 1 .. 7");
        const int pw = wave - 5, H2 = H / 2, W2 = W / 2;
        int rel[NIT], offA[NIT], offB[NIT], rcf[NIT];      // rcf = R << 16 | Cc << 8 | flags (1 item, 2 upper row, 4 lower row, 8 left, 16 right column in the patch)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = (it * NPROD + pw) * 64 + lane, ee = e < NITEM ? e : 0;
            const int c = ee & 7, pp = ee >> 3, R = pp / NPC, Cc = pp % NPC, pi0 = 2 * R - 1, pj0 = 2 * Cc - 1;
            rel[it] = ((R - 1) * W2 + (Cc - 1)) * CIN + c * 8;
            offA[it] = (pi0 * PW + pj0) * 128 + ((c ^ (pj0 & 7)) << 4);
            offB[it] = (pi0 * PW + pj0 + 1) * 128 + ((c ^ ((pj0 + 1) & 7)) << 4);
            rcf[it] = R << 16 | Cc << 8 | (e < NITEM ? 1 : 0) | (pi0 >= 0 ? 2 : 0) | (pi0 + 1 < PH ? 4 : 0) | (pj0 >= 0 ? 8 : 0) | (pj0 + 1 < PW ? 16 : 0);
        }
        u32x4 gr[NIT]; uint2 cd[NIT]; unsigned okm = 0;
        auto issue_loads = [&](int tile, int slab) {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
            const int t2o = ty * (TH / 2), d2o = tx * (TW / 2);
            const long base = (((long)b * H2 + t2o) * W2 + d2o) * CIN + slab * 64;
            okm = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int t2 = t2o + (rcf[it] >> 16) - 1, d2 = d2o + ((rcf[it] >> 8) & 255) - 1;
                const bool ok = (rcf[it] & 1) && (unsigned)t2 < (unsigned)H2 && (unsigned)d2 < (unsigned)W2;   // (floor mode: a cropped last row / column gets no gradient)
                const long o = ok ? base + rel[it] : (long)(lane & 7) * 8;
                gr[it] = *reinterpret_cast<const u32x4*>(a.in_pooled + o);
                cd[it] = *reinterpret_cast<const uint2*>(a.in_idx + o);
                okm |= (ok ? 1u : 0u) << it;
            }
        };
        auto expand_store = [&](int it, char* dst) {
            const int f = rcf[it];
            if (!(f & 1)) return;
            const u32x4 z = {0u, 0u, 0u, 0u};
            u32x4 o[4];
            unpool_expand((okm >> it) & 1 ? gr[it] : z, cd[it], o);
            if (f & 2) {
                if (f & 8) *reinterpret_cast<u32x4*>(dst + offA[it]) = o[0];
                if (f & 16) *reinterpret_cast<u32x4*>(dst + offB[it]) = o[1];
            }
            if (f & 4) {
                if (f & 8) *reinterpret_cast<u32x4*>(dst + offA[it] + PW * 128) = o[2];
                if (f & 16) *reinterpret_cast<u32x4*>(dst + offB[it] + PW * 128) = o[3];
            }
        };
        // Tiles: the FIRST tile of a workgroup is its index (every producer knows it without a hand-over), the later ones come from the
        // counter as in the DMA flavour (wave 5 fetches, two tiles ahead, and publishes through tileq): tile = gridDim.x + counter value.
        bool dry = false;
        const unsigned lim = (unsigned)ntiles, G = gridDim.x;
        auto issue_fetch = [&]() -> unsigned { unsigned t = 0; if (!dry && lane == 0) t = atomicAdd(&a.sched[0], 1u); return t; };
        auto went_dry = [&]() {
            dry = true;
            if (lane == 0 && atomicAdd(&a.sched[1], 1u) == gridDim.x - 1) { atomicExch(&a.sched[0], 0u); atomicExch(&a.sched[1], 0u); }
        };
        auto resolve = [&](unsigned raw) -> int {
            if (dry) return -1;
            const unsigned t = G + __builtin_amdgcn_readfirstlane(raw);
            if (t >= lim) { went_dry(); return -1; }
            return (int)t;
        };
        int cur = (int)blockIdx.x, nxt = -1;
        if (pw == 0) {
            nxt = resolve(issue_fetch());
            if (lane == 0) { tileq[0] = cur; tileq[1] = nxt; }
        }
        issue_loads(cur, 0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) expand_store(it, pbuf);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");           // opening barrier: patch 0 and tileq[0..1] are in
        if (pw != 0) nxt = read_tileq(1);
        int g = 0;
        for (int k = 0; cur >= 0; ++k) {
            int nn = -1;
            unsigned raw = 0;
#pragma unroll 1
            for (int slab = 0; slab < NSLAB; ++slab, ++g) {
                const bool last_slab = slab == NSLAB - 1;
                const bool more = !last_slab || nxt >= 0;
                char* const dst = pbuf + ((g + 1) & 1) * PBYTES;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (tap == 8) {
                        if (pw == 0 && slab == 0) {
                            nn = resolve(raw);
                            if (lane == 0) tileq[(k + 2) & 3] = nn;
                        }
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next stage's patch (and tile id) is in
                    }
                    asm volatile("s_barrier" ::: "memory");
                    if (pw == 0 && tap == 0 && slab == 0) raw = issue_fetch();         // the tile after next, two tiles ahead
                    if (more) {
                        if (tap == 0) issue_loads(last_slab ? nxt : cur, last_slab ? 0 : slab + 1);
                        else if (tap - 1 < NIT) expand_store(tap - 1 < NIT ? tap - 1 : 0, dst);
                    }
                }
            }
            if (pw != 0) nn = read_tileq(k + 2);          // (published before the barrier of slab 0's last tap)
            cur = nxt; nxt = nn;
        }
        return;
    }
    }
    if (!UNPOOL && wave == 5) {
        // ---------------- patch stream.  Per-piece source offsets relative to the tile origin are fixed for the launch.
        const int sub = lane >> 3, sl = lane & 7;
        int rel[PPIECES], pij[PPIECES];
#pragma unroll
        for (int i = 0; i < PPIECES; ++i) {
            int p = i * 8 + sub;
            if (p >= PH * PW) p = PH * PW - 1;             // tail lanes of the last piece: any valid pixel (never read back)
            const int pi = p / PW, pj = p % PW;
            rel[i] = ((pi - 1) * W + (pj - 1)) * CIN + (sl ^ (pj & 7)) * 8;
            pij[i] = pi << 8 | pj;
        }
        const bf16* zsrc = g_zero_line + sl * 8;
        const bf16* org = nullptr;                         // in + image b + tile origin + slab, of the patch being issued
        int t0 = 0, d0 = 0;
        bool interior = false;
        auto begin_patch = [&](int tile, int slab) {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
            d0 = tx * TW; t0 = ty * TH;
            org = a.in + ((long)b * H * W + (long)t0 * W + d0) * CIN + slab * 64;
            interior = t0 >= 1 && t0 + TH + 1 <= H && d0 >= 1 && d0 + TW + 1 <= W;
        };
        auto issue_pieces = [&](int g, int first, int count) {
            char* dst = pbuf + (g & 1) * PBYTES;
            if (interior) {
#pragma unroll
                for (int i = 0; i < PPIECES; ++i)
                    if (i >= first && i < first + count)
                        __builtin_amdgcn_global_load_lds((gptr_t*)(org + rel[i]), (lptr_t*)(dst + i * 1024), 16, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < PPIECES; ++i)
                    if (i >= first && i < first + count) {
                        const int t = t0 + (pij[i] >> 8) - 1, d = d0 + (pij[i] & 255) - 1;
                        const bf16* src = (t >= 0 && t < H && d >= 0 && d < W) ? org + rel[i] : zsrc;
                        __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + i * 1024), 16, 0, 0);
                    }
            }
        };
        // tile fetch: the counter value comes back through vmcnt like any load, so a fetch is issued at the first tap of a
        // tile and only looked at (resolve) at its last one.  `dry`: a workgroup reports running dry exactly once.
        bool dry = false;
        const unsigned lim = (unsigned)ntiles;
        auto issue_fetch = [&](unsigned n) -> unsigned {
            unsigned t = 0;
            if (!dry && lane == 0) t = atomicAdd(&a.sched[0], n);
            return t;
        };
        auto went_dry = [&]() {
            dry = true;
            if (lane == 0 && atomicAdd(&a.sched[1], 1u) == gridDim.x - 1) { atomicExch(&a.sched[0], 0u); atomicExch(&a.sched[1], 0u); }
        };
        auto resolve = [&](unsigned raw) -> int {
            if (dry) return -1;
            const unsigned t = __builtin_amdgcn_readfirstlane(raw);
            if (t >= lim) { went_dry(); return -1; }
            return (int)t;
        };
        int cur, nxt;
        {   // the first two tiles come from one fetch of 2
            const unsigned t = __builtin_amdgcn_readfirstlane(issue_fetch(2u));
            cur = t < lim ? (int)t : -1;
            nxt = t + 1 < lim ? (int)t + 1 : -1;
            if (nxt < 0) went_dry();
        }
        if (lane == 0) { tileq[0] = cur; tileq[1] = nxt; }
        if (cur >= 0) { begin_patch(cur, 0); issue_pieces(0, 0, PPIECES); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");           // opening barrier: patch 0 and tileq[0..1] are in
        int g = 0;
        for (int k = 0; cur >= 0; ++k) {
            int nn = -1;
            unsigned raw = 0;
#pragma unroll 1
            for (int slab = 0; slab < NSLAB; ++slab, ++g) {
                const bool last_slab = slab == NSLAB - 1;
                const bool more = !last_slab || nxt >= 0;
                if (more) begin_patch(last_slab ? nxt : cur, last_slab ? 0 : slab + 1);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (tap == 8) {
                        if (slab == 0) {
                            nn = resolve(raw);
                            if (lane == 0) tileq[(k + 2) & 3] = nn;
                        }
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next stage's patch (and tile id) is in
                    }
                    asm volatile("s_barrier" ::: "memory");
                    if (tap == 0 && slab == 0) raw = issue_fetch(1u);                  // the tile after next, two tiles ahead
                    if (tap < 6 && more) issue_pieces(g + 1, tap * PCHUNK, PCHUNK);
                }
            }
            cur = nxt; nxt = nn;
        }
        return;
    }

    // ---------------- MFMA waves
    const int rr = lane & 15, q = lane >> 4;
    const int pcol0 = rr % TW, prow0 = wave * MF * RPT + rr / TW;
    int poff[3][2], woff[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        woff[kc] = rr * 128 + (((kc * 4 + q) ^ (rr & 7)) * 16);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) poff[dx][kc] = (prow0 * PW + pcol0 + dx) * 128 + (((kc * 4 + q) ^ ((pcol0 + dx) & 7)) * 16);
    }
    // lane (rr, q) ends up with channels h*32 + q*8 .. +7 of pixel rr of each pixel tile.  The bias waits in LDS for the
    // epilogue (32 registers per lane at 128 output channels, which the 32-row tiles do not have)
    for (int c = threadIdx.x; c < COUT; c += 256) sbias[c] = (!MASK && a.bias) ? a.bias[c] : 0.f;

    // One half-tap = MF*NF MFMAs on one fragment set while the reads of the NEXT set are issued in their shadow, one read
    // after each of the first MF+NF MFMAs (an MFMA holds the pipe for 16 cycles, the wave is free to issue in between; with
    // all the reads issued up front the pipe idles for their issue time every half-tap).  sched_barrier(0) after every
    // instruction pins exactly this order.  The reads are inline asm waited on by hand: the compiler's own bookkeeping would
    // put a full lgkmcnt(0) between a set's reads and the MFMAs of the previous set, an exposed LDS round trip per half-tap.
    // Pixel fragments (A) are double-buffered; weight fragments (B) live in ONE set: the MFMAs run weight-block-major
    // (all pixel tiles of block j, then block j+1), so block j's registers are free after its MF MFMAs and its reload for
    // the next half-tap is issued right there -- it has the rest of this half-tap to land and is needed in the same order.
    // (A second B set costs 32 registers per lane at 128 output channels, which 32-row tiles do not have.)
    struct AFrags { bf16x8 a[MF]; };
    bf16x8 bfr[NF];
    typedef __attribute__((address_space(3))) const char lds_cchar;
    f32x4 acc[MF][NF];
    auto half_tap = [&](const AFrags& use, AFrags& ld, const char* pl, const char* wl, int tap, int kc) {
        const int dy = tap / 3, dx = tap % 3;
        const unsigned pa = (unsigned)(size_t)((lds_cchar*)pl) + poff[dx][kc];
        const unsigned wa = (unsigned)(size_t)((lds_cchar*)wl) + woff[kc];
        // LDS reads return in issue order: ..., a[0..MF) , b[0], ..., b[NF-1] of the previous half-tap, then this half-tap's.
        // Block 0 needs a[*] and b[0]: at most the NF-1 later weight reads may be outstanding.  Block j > 0 needs b[j]:
        // behind it are b[j+1..NF) and this half-tap's MF + j reads, MF + NF - 1 in all.
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NF - 1) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MF * NF; ++m) {
            const int j = m / MF, i = m % MF;
            if (i == 0 && j > 0) {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MF + NF - 1) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[i][j] = mma16(bfr[j], use.a[i], acc[i][j]);     // weights as A: a lane ends up with 4 consecutive channels
            __builtin_amdgcn_sched_barrier(0);
            if (m < MF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld.a[m < MF ? m : 0]) : "v"(pa), "n"(((m < MF ? m : 0) * RPT + dy) * PW * 128));
            if (i == MF - 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[j]) : "v"(wa), "n"(j * 2048));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto load_frags = [&](AFrags& f, const char* pcur, const char* wcur, int tap, int kc) {
        const int dy = tap / 3, dx = tap % 3;
        const unsigned pa = (unsigned)(size_t)((lds_cchar*)pcur) + poff[dx][kc];
        const unsigned wa = (unsigned)(size_t)((lds_cchar*)wcur) + woff[kc];
#pragma unroll
        for (int i = 0; i < MF; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a[i]) : "v"(pa), "n"((i * RPT + dy) * PW * 128));
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[j]) : "v"(wa), "n"(j * 2048));
    };

    AFrags f0, f1;
    asm volatile("s_barrier" ::: "memory");               // opening barrier
    int tile = read_tileq(0);
    if (tile >= 0) load_frags(f0, pbuf, wbuf, 0, 0);
    int g = 0;
    for (int k = 0; tile >= 0; ++k) {
        // this lane's output addressing for the tile
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int d0 = tx * TW, t0 = ty * TH;
        const int tl = t0 + prow0, dl = d0 + pcol0;        // this lane's pixel of pixel tile 0
        const int CS = a.out_cstride ? a.out_cstride : COUT;      // channels per pixel of the output map (this launch: out_coff .. + COUT)
        const unsigned voff = (unsigned)((tl * W + dl) * CS + a.out_coff + q * 8) * 2u;
        char* out_b = reinterpret_cast<char*>(a.out) + (long)b * H * W * CS * 2;
        auto row_ok = [&](int i) { return tl + i * RPT < H && dl < W; };
        auto row_off = [&](int i) { return voff + (unsigned)(i * RPT) * (unsigned)(W * CS * 2); };
        const char* mask_b = MASK == 1 ? reinterpret_cast<const char*>(a.mask) + (long)b * H * W * CS * 2 : nullptr;
        u32x4 mk[MASK == 1 ? MF : 1][MASK == 1 ? NH : 1]; // ReLU mask of this lane's outputs, requested three taps before the epilogue
        unsigned mb[MASK == 2 ? MF : 1];                  // ... as sign bits: this lane's dword of its pixel's four
        const unsigned* bits_b = MASK == 2 ? reinterpret_cast<const unsigned*>(a.mask_bits) + (((long)b * H + tl) * W + dl) * 4 + q : nullptr;
        int next_tile = -1;
#pragma unroll 1
        for (int slab = 0; slab < NSLAB; ++slab, ++g) {
            if (slab == 0) {
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const char* pcur = pbuf + (g & 1) * PBYTES;
            const char* pnext = pbuf + ((g + 1) & 1) * PBYTES;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if constexpr (MASK == 1) {
                    if (tap == 6 && slab == NSLAB - 1) {
#pragma unroll
                        for (int i = 0; i < MF; ++i)
#pragma unroll
                            for (int h = 0; h < NH; ++h) {
                                const u32x4 z = {0u, 0u, 0u, 0u};
                                mk[i][h] = row_ok(i) ? *reinterpret_cast<const u32x4*>(mask_b + row_off(i) + h * 64) : z;
                            }
                    }
                }
                if constexpr (MASK == 2) {
                    if (tap == 0 && slab == NSLAB - 1) {
#pragma unroll
                        for (int i = 0; i < MF; ++i) mb[i] = row_ok(i) ? bits_b[(long)i * RPT * W * 4] : 0u;
                    }
                }
                const char* wcur = wbuf + ((g + tap) & (D - 1)) * WBYTES;          // 9 g + tap = g + tap (mod 4)
                const char* wnext = wbuf + ((g + tap + 1) & (D - 1)) * WBYTES;
                // first half: MFMAs on f0 (read during the previous half-tap), reads of this tap's second k-half into f1
                half_tap(f0, f1, pcur, wcur, tap, 1);
                // barrier u: slice u+1 (and at tap 8 the next patch) is in.  No LDS wait is needed here: the reads this wave
                // still has in flight are of slice u and of the current patch, not of the buffers the producers may now
                // overwrite (slice u-1, the previous stage's patch: their last reads were consumed a half-tap ago)
                asm volatile("s_barrier" ::: "memory");
                // second half: MFMAs on f1, reads of the next tap's first k-half into f0 (after the very last tap of the
                // workgroup these read stale LDS and are never used)
                if (tap < 8) half_tap(f1, f0, pcur, wnext, tap + 1, 0);
                else half_tap(f1, f0, pnext, wnext, 0, 0);
            }
            if (slab == NSLAB - 1) next_tile = read_tileq(k + 1);
        }
        tile = next_tile;

        // epilogue of the tile, straight from the accumulators: bias in fp32, round to bf16, then ReLU / pool as
        // packed 16-bit integer ops (for bf16 bit patterns max(x, 0) is max_i16(x, 0), and non-negative values order like integers)
        const int H2 = H / 2, W2 = W / 2;
        constexpr int IP = (TW == 16) ? 2 : 1;             // pixel tiles per pooling group (TW = 16: two rows = two tiles)
        const short fl = (!MASK && (a.relu & 255)) ? (short)0 : (short)-32768;
        const short2_t floor2 = {fl, fl};
        const bool store_dense = MASK || !(a.out_optional && a.pool_out), want_idx = a.pool_idx != nullptr;
        short2_t pm[NH][4];
        unsigned sgn[(MASK == 0 && COUT == 128) ? MF : 1];
#pragma unroll
        for (int i = 0; i < ((MASK == 0 && COUT == 128) ? MF : 1); ++i) sgn[i] = 0u;
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const bool ok = row_ok(i);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const unsigned o = row_off(i) + h * 64;
                short2_t pk[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4& c = acc[i][2 * h + (kk >> 1)];
                    f32x2 v = {c[2 * (kk & 1)], c[2 * (kk & 1) + 1]};
                    v += *reinterpret_cast<const f32x2*>(&sbias[h * 32 + q * 8 + 2 * kk]);
                    unsigned r;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[0]), "v"(v[1]));
                    pk[kk] = __builtin_elementwise_max(__builtin_bit_cast(short2_t, r), floor2);   // ReLU (or a no-op)
                }
                if constexpr (MASK == 1) {
                    const short2_t zero2 = {0, 0}, one2 = {1, 1};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const unsigned mw = mk[i][h][kk];      // (bit_cast straight from a vector element reads element 0: clang 22)
                        short2_t m = __builtin_bit_cast(short2_t, mw);
                        m = __builtin_elementwise_min(__builtin_elementwise_max(m, zero2), one2);   // 1 where mask > 0
                        pk[kk] = pk[kk] & (zero2 - m);
                    }
                }
                if constexpr (MASK == 2) {
                    const int byte = (int)(mb[i] >> (8 * h));   // bits 2kk, 2kk+1 -> the two halves of packed pair kk
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(byte, 2 * kk, 1), hi = (unsigned)__builtin_amdgcn_sbfe(byte, 2 * kk + 1, 1);
                        pk[kk] = __builtin_bit_cast(short2_t, __builtin_bit_cast(unsigned, pk[kk]) & ((lo & 0xFFFFu) | (hi & 0xFFFF0000u)));
                    }
                }
                if constexpr (MASK == 0 && COUT == 128) {
                    if (a.out_sign_bits) {                     // (uniform) sign byte of this lane's 8 ReLU'd channels: per half min(v, 1) = 1 for v > 0
                        unsigned e = 0;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) e |= __builtin_bit_cast(unsigned, __builtin_elementwise_min(pk[kk], short2_t{1, 1})) << (2 * kk);
                        sgn[i] |= ((e | e >> 15) & 0xFFu) << (8 * h);
                    }
                }
                if (ok && store_dense) {
                    u32x4 ov;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) ov[kk] = __builtin_bit_cast(unsigned, pk[kk]);
                    *reinterpret_cast<u32x4*>(out_b + o) = ov;
                }
                if (!MASK && a.pool_out) {
                    // 2 x 2 window (ReLU'd values only): the row pair is (tile i, tile i+1) for TW = 16 and (lane, lane ^ 8)
                    // for TW = 8; the column pair is (lane, lane ^ 1).  The owner lane holds the window's top-left element.
                    if (IP == 2 && (i & 1) == 0) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) pm[h][kk] = pk[kk];
                    } else {
                        u32x4 po;
                        unsigned code[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const short2_t v00 = IP == 2 ? pm[h][kk] : pk[kk];
                            const short2_t v01 = lane_xor1(v00);
                            const short2_t v10 = TW == 8 ? lane_xor8(v00) : pk[kk];
                            const short2_t v11 = lane_xor1(v10);
                            const short2_t m01 = __builtin_elementwise_max(v00, v01), m23 = __builtin_elementwise_max(v10, v11);
                            const short2_t x = __builtin_elementwise_max(m01, m23);
                            po[kk] = __builtin_bit_cast(unsigned, x);
                            code[kk] = want_idx ? pool_code2(v00, v01, v10, v11, m01, m23, x) : 0u;
                        }
                        const int t2 = (tl + (i - (IP - 1)) * RPT) >> 1, d2 = dl >> 1;
                        const bool owner = (rr & 1) == 0 && (TW == 16 || rr < 8);
                        if (owner && t2 < H2 && d2 < W2) {
                            const long pe = (((long)b * H2 + t2) * W2 + d2) * COUT + h * 32 + q * 8;
                            *reinterpret_cast<u32x4*>(a.pool_out + pe) = po;
                            if (want_idx) *reinterpret_cast<uint2*>(a.pool_idx + pe) = uint2{pool_pack4(code[0], code[1]), pool_pack4(code[2], code[3])};
                        }
                    }
                }
            }
            if constexpr (MASK == 0 && COUT == 128) {
                // this lane's dword of the pixel's sign words: one 4-byte store per pixel tile
                if (a.out_sign_bits && ok) reinterpret_cast<unsigned*>(a.out_sign_bits)[(((long)b * H + tl + i * RPT) * W + dl) * 4 + q] = sgn[i];
            }
        }
    }
}

// ------------------------------------------------------------------ streaming 3x3, 64 -> 64 channels: weights resident in LDS
// The streaming kernel above is bound by the CU's vector-memory issue rate, not by bytes: every 1 KiB wave-instruction
// (LDS-DMA piece or 16-byte-per-lane store) costs the texture addresser ~65 cycles, and a 16 x 16 tile needs 41 (patch)
// + 72 (nine 8 KiB weight slices) + 32 (stores) of them against 4.6k cycles of MFMA.  With 64 input and 64 output
// channels the whole filter bank is 72 KiB: a persistent workgroup loads it ONCE, which halves the instruction count per
// tile, removes the weight wave and leaves one barrier per tile (patch hand-over) instead of nine.
// LDS: 9 x [64 rows][128 B] weights (rows permuted, chunks XOR-swizzled as above) + two patches = 154 KiB.
template <int TH, int TW>
__global__ __launch_bounds__(320) void conv3x3_resw_kernel(ConvArgs a, int ntiles, int tiles_x, int tiles_y) {
    constexpr int CIN = 64, COUT = 64;
    constexpr int PW = TW + 2, PH = TH + 2, RPT = 16 / TW;
    constexpr int MF = TH * TW / 64, NF = COUT / 16, NH = NF / 2;
    constexpr int KTOT = 9 * CIN;
    constexpr int PPIECES = (PH * PW + 7) / 8;             // 1 KiB DMA pieces per patch (8 pixels x 128 B)
    constexpr int PBYTES = PPIECES * 1024, WBYTES = COUT * 128;
    __shared__ __attribute__((aligned(1024))) char lds[2 * PBYTES + 9 * WBYTES];
    __shared__ int tileq[4];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    typedef __attribute__((ext_vector_type(2))) short short2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    char* const pbuf = lds;
    char* const wres = lds + 2 * PBYTES;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, W = a.W;
    // (an LDS-qualified volatile read: through a generic `volatile int*` hipcc emits flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0), i.e.
    // the read queues at the texture addresser behind the patch wave's DMA pieces and waits for all of this wave's stores)
    auto read_tileq = [&](int k) { return __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)&tileq[k & 3]); };

    {   // the filter bank: 72 pieces of 8 rows, dealt round-robin to the 5 waves
        const int sub = lane >> 3, sl = lane & 7;
#pragma unroll
        for (int n = 0; n < (72 + 4) / 5; ++n) {
            const int pc = n * 5 + wave;
            if (pc < 72) {
                const int tap = pc >> 3, i = pc & 7;
                const int r = i * 8 + sub, j = r >> 4, m = r & 15;
                const int co = (j >> 1) * 32 + (m >> 2) * 8 + (j & 1) * 4 + (m & 3);
                const bf16* src = a.wk + (long)co * KTOT + tap * CIN + ((sl ^ (r & 7)) * 8);
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(wres + pc * 1024), 16, 0, 0);
            }
        }
    }

    if (wave == 4) {
        // ---------------- patch stream (see the streaming kernel); barrier g hands patch g over and frees patch g-1's buffer
        const int sub = lane >> 3, sl = lane & 7;
        int rel[PPIECES], pij[PPIECES];
#pragma unroll
        for (int i = 0; i < PPIECES; ++i) {
            int p = i * 8 + sub;
            if (p >= PH * PW) p = PH * PW - 1;
            const int pi = p / PW, pj = p % PW;
            rel[i] = ((pi - 1) * W + (pj - 1)) * CIN + (sl ^ (pj & 7)) * 8;
            pij[i] = pi << 8 | pj;
        }
        const bf16* zsrc = g_zero_line + sl * 8;
        auto issue_patch = [&](int tile, int g) {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
            const int d0 = tx * TW, t0 = ty * TH;
            const bf16* org = a.in + ((long)b * H * W + (long)t0 * W + d0) * CIN;
            const bool interior = t0 >= 1 && t0 + TH + 1 <= H && d0 >= 1 && d0 + TW + 1 <= W;
            char* dst = pbuf + (g & 1) * PBYTES;
            if (interior) {
#pragma unroll
                for (int i = 0; i < PPIECES; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t*)(org + rel[i]), (lptr_t*)(dst + i * 1024), 16, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < PPIECES; ++i) {
                    const int t = t0 + (pij[i] >> 8) - 1, d = d0 + (pij[i] & 255) - 1;
                    const bf16* src = (t >= 0 && t < H && d >= 0 && d < W) ? org + rel[i] : zsrc;
                    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + i * 1024), 16, 0, 0);
                }
            }
        };
        bool dry = false;
        const unsigned lim = (unsigned)ntiles;
        auto issue_fetch = [&](unsigned n) -> unsigned {
            unsigned t = 0;
            if (!dry && lane == 0) t = atomicAdd(&a.sched[0], n);
            return t;
        };
        auto went_dry = [&]() {
            dry = true;
            if (lane == 0 && atomicAdd(&a.sched[1], 1u) == gridDim.x - 1) { atomicExch(&a.sched[0], 0u); atomicExch(&a.sched[1], 0u); }
        };
        int cur, nxt;
        {
            const unsigned t = __builtin_amdgcn_readfirstlane(issue_fetch(2u));
            cur = t < lim ? (int)t : -1;
            nxt = t + 1 < lim ? (int)t + 1 : -1;
            if (nxt < 0) went_dry();
        }
        if (lane == 0) { tileq[0] = cur; tileq[1] = nxt; }
        if (cur >= 0) issue_patch(cur, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");           // opening barrier: weights (every wave's share), patch 0, tileq[0..1]
        for (int k = 0; cur >= 0; ++k) {
            // tile k is being computed: fetch the id of tile k+2, stream tile k+1's patch, publish, hand over
            const unsigned raw = issue_fetch(1u);
            if (nxt >= 0) issue_patch(nxt, k + 1);
            int nn = -1;
            if (!dry) {
                const unsigned t = __builtin_amdgcn_readfirstlane(raw);
                if (t >= lim) went_dry(); else nn = (int)t;
            }
            if (lane == 0) tileq[(k + 2) & 3] = nn;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");       // barrier k (middle of tile k's last tap)
            cur = nxt; nxt = nn;
        }
        return;
    }

    // ---------------- MFMA waves
    const int rr = lane & 15, q = lane >> 4;
    const int pcol0 = rr % TW, prow0 = wave * MF * RPT + rr / TW;
    int poff[3][2], woff[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        woff[kc] = rr * 128 + (((kc * 4 + q) ^ (rr & 7)) * 16);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) poff[dx][kc] = (prow0 * PW + pcol0 + dx) * 128 + (((kc * 4 + q) ^ ((pcol0 + dx) & 7)) * 16);
    }
    f32x2 bv[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bv[h][k][0] = a.bias ? a.bias[h * 32 + q * 8 + 2 * k] : 0.f;
            bv[h][k][1] = a.bias ? a.bias[h * 32 + q * 8 + 2 * k + 1] : 0.f;
        }
    struct Frags { bf16x8 a[MF], b[NF]; };
    typedef __attribute__((address_space(3))) const char lds_cchar;
    const unsigned wbase = (unsigned)(size_t)((lds_cchar*)wres);
    f32x4 acc[MF][NF];
    // half_tap: MF*NF MFMAs on `use`, the reads of `ld` issued in their shadow (see the streaming kernel)
    auto half_tap = [&](const Frags& use, Frags& ld, const char* pl, int tap, int kc) {
        const int dy = tap / 3, dx = tap % 3;
        const unsigned pa = (unsigned)(size_t)((lds_cchar*)pl) + poff[dx][kc];
        const unsigned wa = wbase + woff[kc] + tap * WBYTES;   // (the DS offset field is 16 bits: the tap's 8 KiB stride goes here)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MF * NF; ++m) {
            const int i = m / NF, j = m % NF;
            acc[i][j] = mma16(use.b[j], use.a[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (m < MF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld.a[m < MF ? m : 0]) : "v"(pa), "n"(((m < MF ? m : 0) * RPT + dy) * PW * 128));
            else if (m < MF + NF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld.b[m < MF + NF ? m - MF : 0]) : "v"(wa), "n"((m < MF + NF ? m - MF : 0) * 2048));
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    Frags f0, f1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the weights
    asm volatile("s_barrier" ::: "memory");               // opening barrier
    int tile = read_tileq(0);
    {   // fragments of (tap 0, first k-half)
        const unsigned pa = (unsigned)(size_t)((lds_cchar*)pbuf) + poff[0][0];
        const unsigned wa = wbase + woff[0];
#pragma unroll
        for (int i = 0; i < MF; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f0.a[i]) : "v"(pa), "n"(i * RPT * PW * 128));
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f0.b[j]) : "v"(wa), "n"(j * 2048));
    }
    for (int k = 0; tile >= 0; ++k) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int d0 = tx * TW, t0 = ty * TH;
        const int tl = t0 + prow0, dl = d0 + pcol0;
        const unsigned voff = (unsigned)((tl * W + dl) * COUT + q * 8) * 2u;
        char* out_b = reinterpret_cast<char*>(a.out) + (long)b * H * W * COUT * 2;
        auto row_ok = [&](int i) { return tl + i * RPT < H && dl < W; };
        auto row_off = [&](int i) { return voff + (unsigned)(i * RPT) * (unsigned)(W * COUT * 2); };
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* pcur = pbuf + (k & 1) * PBYTES;
        const char* pnext = pbuf + ((k + 1) & 1) * PBYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            half_tap(f0, f1, pcur, tap, 1);
            if (tap < 8) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                half_tap(f1, f0, pcur, tap + 1, 0);
            } else {
                // barrier k: the next tile's patch and tileq[k+1..k+2] are in; this wave will not read patch k again
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                half_tap(f1, f0, pnext, 0, 0);            // (after the last tile: stale LDS, never used)
            }
        }
        const int next_tile = read_tileq(k + 1);
        tile = next_tile;

        // epilogue as in the streaming kernel
        const int H2 = H / 2, W2 = W / 2;
        constexpr int IP = (TW == 16) ? 2 : 1;
        const short fl = (a.relu & 255) ? (short)0 : (short)-32768;
        const short2_t floor2 = {fl, fl};
        const bool store_dense = !(a.out_optional && a.pool_out), want_idx = a.pool_idx != nullptr;
        short2_t pm[NH][4];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
            const bool ok = row_ok(i);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const unsigned o = row_off(i) + h * 64;
                short2_t pk[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4& c = acc[i][2 * h + (kk >> 1)];
                    f32x2 v = {c[2 * (kk & 1)], c[2 * (kk & 1) + 1]};
                    v += bv[h][kk];
                    unsigned r;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[0]), "v"(v[1]));
                    pk[kk] = __builtin_elementwise_max(__builtin_bit_cast(short2_t, r), floor2);
                }
                if (ok && store_dense) {
                    u32x4 ov;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) ov[kk] = __builtin_bit_cast(unsigned, pk[kk]);
                    *reinterpret_cast<u32x4*>(out_b + o) = ov;
                }
                if (a.pool_out) {
                    if (IP == 2 && (i & 1) == 0) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) pm[h][kk] = pk[kk];
                    } else {
                        u32x4 po;
                        unsigned code[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            const short2_t v00 = IP == 2 ? pm[h][kk] : pk[kk];
                            const short2_t v01 = lane_xor1(v00);
                            const short2_t v10 = TW == 8 ? lane_xor8(v00) : pk[kk];
                            const short2_t v11 = lane_xor1(v10);
                            const short2_t m01 = __builtin_elementwise_max(v00, v01), m23 = __builtin_elementwise_max(v10, v11);
                            const short2_t x = __builtin_elementwise_max(m01, m23);
                            po[kk] = __builtin_bit_cast(unsigned, x);
                            code[kk] = want_idx ? pool_code2(v00, v01, v10, v11, m01, m23, x) : 0u;
                        }
                        const int t2 = (tl + (i - (IP - 1)) * RPT) >> 1, d2 = dl >> 1;
                        const bool owner = (rr & 1) == 0 && (TW == 16 || rr < 8);
                        if (owner && t2 < H2 && d2 < W2) {
                            const long pe = (((long)b * H2 + t2) * W2 + d2) * COUT + h * 32 + q * 8;
                            *reinterpret_cast<u32x4*>(a.pool_out + pe) = po;
                            if (want_idx) *reinterpret_cast<uint2*>(a.pool_idx + pe) = uint2{pool_pack4(code[0], code[1]), pool_pack4(code[2], code[3])};
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ 64 -> 64 dgrad of the second conv with the conv1 weight gradient fused
// Same skeleton as the resident-weight kernel (persistent workgroups, filter bank in LDS, patch stream by LDS-DMA) on 16 x 16-pixel
// tiles (0.5 LDS fragment reads per MFMA, like the forward kernel): the accumulators are masked with the ReLU mask of conv1's output,
// written to LDS as a bf16 tile and contracted on the MFMA with the 3 x 3 neighbourhoods of the 1-channel network input (streamed in
// with the patch) -- d(conv1 output) is never stored; each wave keeps its 16 channels x 16 taps partial in registers over ALL tiles of
// the workgroup (one row of 640 sums per workgroup).  Two 41 KiB patches and the 72 KiB filter bank leave 6 KiB of the CU's 160 KiB, so:
//   * the masked bf16 tile [256 pixels][64 + 8 channels] (36 KiB) is staged in the patch buffer the tile has just CONSUMED (no wave
//     reads patch k after barrier k); a third barrier per tile (E3, in place of the old "tap image built" one) keeps the patch
//     wave from refilling that buffer before the contraction has read it;
//   * the tap image of the 1-channel network input ([9 taps][256 pixels] bf16, 4.6 KiB, kappa-ordered for frag_rm) is built by the
//     PATCH wave -- idle between its DMA issue and the hand-over barrier -- from the fp32 neighbourhood that travels with the patch
//     (one 4-pixel group per lane: three 8-byte reads and three 8-byte writes per tap row), while the MFMA waves run the tap loop:
//     the MFMA waves' epilogue, every instruction of which is on the critical path, loses the 16 + 16 LDS accesses per lane and one
//     barrier; the row of ones (bias gradient) is a register constant.
// UNPOOL: the input map (the gradient behind the first max-pool) is never materialised -- FOUR producer waves (4 .. 7, one per SIMD;
// wave 4 keeps the tap image and the neighbourhood DMA) build each patch from the pooled gradient a.in_pooled + the pool codes
// a.in_idx (unpool_expand), 512 threads.
template <bool UNPOOL = false>
__global__ __launch_bounds__(UNPOOL ? 512 : 320) void conv3x3_resw_w1x_kernel(ConvArgs a, int ntiles, int tiles_x, int tiles_y) {
    constexpr int MASK_TAP = 0;
    constexpr int CIN = 64, COUT = 64, TH = 16, TW = 16;
    constexpr int PW = TW + 2, PH = TH + 2, RPT = 16 / TW;
    constexpr int MF = TH * TW / 64, NF = COUT / 16, NH = NF / 2;
    constexpr int KTOT = 9 * CIN;
    constexpr int PPIECES = (PH * PW + 7) / 8;             // 1 KiB DMA pieces per patch (8 pixels x 128 B)
    constexpr int PBYTES = PPIECES * 1024, WBYTES = COUT * 128;
    constexpr int OS = COUT + 8;                           // row stride of the masked bf16 tile [pixel][channel] (frag_rm reads)
    constexpr int NPIX = TH * TW, XS = NPIX + 8;           // tap-major input image [9 taps][XS]
    constexpr int XT_BYTES = 9 * XS * 2, XP_FLOATS = (PH * PW + 7) / 8 * 8;
    static_assert(NPIX * OS * 2 <= PBYTES, "the masked tile must fit into a consumed patch buffer");
    static_assert(2 * PBYTES + 9 * WBYTES + XT_BYTES + XP_FLOATS * 4 + 16 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) char lds[2 * PBYTES + 9 * WBYTES + XT_BYTES + XP_FLOATS * 4 + 16];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    typedef __attribute__((ext_vector_type(2))) short short2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    char* const pbuf = lds;
    char* const wres = lds + 2 * PBYTES;
    bf16* const sxt = reinterpret_cast<bf16*>(wres + 9 * WBYTES);
    float* const sxp = reinterpret_cast<float*>(wres + 9 * WBYTES + XT_BYTES);
    int* const tileq = reinterpret_cast<int*>(wres + 9 * WBYTES + XT_BYTES + XP_FLOATS * 4);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, W = a.W;
    // (an LDS-qualified volatile read: through a generic `volatile int*` hipcc emits flat_load_dword sc0 sc1 + s_waitcnt vmcnt(0), i.e.
    // the read queues at the texture addresser behind the patch wave's DMA pieces and waits for all of this wave's stores)
    auto read_tileq = [&](int k) { return __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) int*)&tileq[k & 3]); };

    {   // the filter bank: 72 pieces of 8 rows, dealt round-robin to the waves
        constexpr int NWV = UNPOOL ? 8 : 5;
        const int sub = lane >> 3, sl = lane & 7;
#pragma unroll
        for (int n = 0; n < (72 + NWV - 1) / NWV; ++n) {
            const int pc = n * NWV + wave;
            if (pc < 72) {
                const int tap = pc >> 3, i = pc & 7;
                const int r = i * 8 + sub, j = r >> 4, m = r & 15;
                const int co = (j >> 1) * 32 + (m >> 2) * 8 + (j & 1) * 4 + (m & 3);
                const bf16* src = a.wk + (long)co * KTOT + tap * CIN + ((sl ^ (r & 7)) * 8);
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(wres + pc * 1024), 16, 0, 0);
            }
        }
    }

    if constexpr (UNPOOL) {
    if (wave >= 4) {
        // ---------------- patch producers (see the streaming kernel's UNPOOL flavour: item = pooled cell x 16-byte channel chunk); wave 4
        // also keeps the tap image, the neighbourhood DMA and tileq.  Tiles are dealt by a static stride, so every producer knows them.
        constexpr int NPROD = 4, NPR = PH / 2 + 1, NPC = PW / 2 + 1, NITEM = NPR * NPC * 8, NIT = (NITEM + 64 * NPROD - 1) / (64 * NPROD);
        const int pw = wave - 4, H2 = H / 2, W2 = W / 2;
        int rel[NIT], offA[NIT], offB[NIT], rcf[NIT];      // rcf = R << 16 | Cc << 8 | flags (1 item, 2 upper row, 4 lower row, 8 left, 16 right column in the patch)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = (it * NPROD + pw) * 64 + lane, ee = e < NITEM ? e : 0;
            const int c = ee & 7, pp = ee >> 3, R = pp / NPC, Cc = pp % NPC, pi0 = 2 * R - 1, pj0 = 2 * Cc - 1;
            rel[it] = ((R - 1) * W2 + (Cc - 1)) * CIN + c * 8;
            offA[it] = (pi0 * PW + pj0) * 128 + ((c ^ (pj0 & 7)) << 4);
            offB[it] = (pi0 * PW + pj0 + 1) * 128 + ((c ^ ((pj0 + 1) & 7)) << 4);
            rcf[it] = R << 16 | Cc << 8 | (e < NITEM ? 1 : 0) | (pi0 >= 0 ? 2 : 0) | (pi0 + 1 < PH ? 4 : 0) | (pj0 >= 0 ? 8 : 0) | (pj0 + 1 < PW ? 16 : 0);
        }
        u32x4 gr[NIT]; uint2 cd[NIT]; unsigned okm = 0;
        auto issue_loads = [&](int tile) {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
            const int t2o = ty * (TH / 2), d2o = tx * (TW / 2);
            const long base = (((long)b * H2 + t2o) * W2 + d2o) * CIN;
            okm = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int t2 = t2o + (rcf[it] >> 16) - 1, d2 = d2o + ((rcf[it] >> 8) & 255) - 1;
                const bool ok = (rcf[it] & 1) && (unsigned)t2 < (unsigned)H2 && (unsigned)d2 < (unsigned)W2;   // (floor mode: a cropped last row / column gets no gradient)
                const long o = ok ? base + rel[it] : (long)(lane & 7) * 8;
                gr[it] = *reinterpret_cast<const u32x4*>(a.in_pooled + o);
                cd[it] = *reinterpret_cast<const uint2*>(a.in_idx + o);
                okm |= (ok ? 1u : 0u) << it;
            }
        };
        auto expand_store = [&](int it, char* dst) {
            const int f = rcf[it];
            if (!(f & 1)) return;
            const u32x4 z = {0u, 0u, 0u, 0u};
            u32x4 o[4];
            unpool_expand((okm >> it) & 1 ? gr[it] : z, cd[it], o);
            if (f & 2) {
                if (f & 8) *reinterpret_cast<u32x4*>(dst + offA[it]) = o[0];
                if (f & 16) *reinterpret_cast<u32x4*>(dst + offB[it]) = o[1];
            }
            if (f & 4) {
                if (f & 8) *reinterpret_cast<u32x4*>(dst + offA[it] + PW * 128) = o[2];
                if (f & 16) *reinterpret_cast<u32x4*>(dst + offB[it] + PW * 128) = o[3];
            }
        };
        const float* zx = reinterpret_cast<const float*>(g_zero_line);
        auto issue_x1 = [&](int tile) {                    // the (TH+2) x 18 neighbourhood of the 1-channel network input, ONE buffer (see below)
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
            const int d0 = tx * TW, t0 = ty * TH;
            const float* x_b = a.x1 + (long)b * H * W;
#pragma unroll
            for (int i = 0; i < (PH * PW + 63) / 64; ++i) {
                const int e = i * 64 + lane, pi = e / PW, pj = e % PW;
                const int t = t0 + pi - 1, d = d0 + pj - 1;
                const float* src = (t >= 0 && t < H && d >= 0 && d < W) ? x_b + (long)t * W + d : zx;
                if (e < PH * PW) __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(sxp + i * 64), 4, 0, 0);
            }
        };
        const int gpr = lane >> 2, gc = lane & 3;
        const int xt_off = (gpr >> 1) * 32 + 8 * gc + 4 * (gpr & 1);
        auto build_tap_image = [&]() {                     // (as the DMA flavour's, below)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const f32x2* row = reinterpret_cast<const f32x2*>(sxp + (gpr + dy) * PW + 4 * gc);
                const f32x2 v0 = row[0], v1 = row[1], v2 = row[2];
                const float w[6] = {v0[0], v0[1], v1[0], v1[1], v2[0], v2[1]};
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    bf16x4 o;
                    o[0] = (bf16)w[dx]; o[1] = (bf16)w[dx + 1]; o[2] = (bf16)w[dx + 2]; o[3] = (bf16)w[dx + 3];
                    *reinterpret_cast<bf16x4*>(sxt + (dy * 3 + dx) * XS + xt_off) = o;
                }
            }
        };
        const int G = (int)gridDim.x;
        int cur = (int)blockIdx.x;                          // (the grid never exceeds the tile count)
        issue_loads(cur);
        if (pw == 0) {
            issue_x1(cur);
            if (lane == 0) { tileq[0] = cur; tileq[1] = cur + G < ntiles ? cur + G : -1; }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) expand_store(it, pbuf);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");           // opening barrier: weights (every wave's share), patch 0, tileq[0..1]
        for (int k = 0; cur >= 0; ++k) {
            const int nxt = cur + G < ntiles ? cur + G : -1;
            if (nxt >= 0) issue_loads(nxt);
            if (pw == 0) {
                // tile k is being computed: its tap image first (frees sxp), then tile k+1's neighbourhood
                build_tap_image();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (nxt >= 0) issue_x1(nxt);
                if (lane == 0) tileq[(k + 2) & 3] = (nxt >= 0 && nxt + G < ntiles) ? nxt + G : -1;
            }
            if (nxt >= 0) {
                char* const dst = pbuf + ((k + 1) & 1) * PBYTES;       // (patch k-1's buffer: free since E3 of tile k-1)
#pragma unroll
                for (int it = 0; it < NIT; ++it) expand_store(it, dst);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");       // barrier k (middle of tile k's last tap)
            asm volatile("s_barrier" ::: "memory");       // E1: the masked tile is staged (in patch k's buffer)
            asm volatile("s_barrier" ::: "memory");       // E3: contraction done -- patch k's buffer and the tap image are free again
            cur = nxt;
        }
        return;
    }
    }
    if (!UNPOOL && wave == 4) {
        // ---------------- patch stream + tap image; barrier k hands patch k+1 and tap image k over, E3 frees patch k's buffer
        const int sub = lane >> 3, sl = lane & 7;
        int rel[PPIECES], pij[PPIECES];
#pragma unroll
        for (int i = 0; i < PPIECES; ++i) {
            int p = i * 8 + sub;
            if (p >= PH * PW) p = PH * PW - 1;
            const int pi = p / PW, pj = p % PW;
            rel[i] = ((pi - 1) * W + (pj - 1)) * CIN + (sl ^ (pj & 7)) * 8;
            pij[i] = pi << 8 | pj;
        }
        const bf16* zsrc = g_zero_line + sl * 8;
        const float* zx = reinterpret_cast<const float*>(g_zero_line);
        auto issue_patch = [&](int tile, int g) {
            const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
            const int d0 = tx * TW, t0 = ty * TH;
            {   // the (TH+2) x 18 neighbourhood of the 1-channel network input (4-byte LDS-DMA, zeros outside the image), ONE buffer:
                // issued after the tap image of the running tile has been built from the previous content
                const float* x_b = a.x1 + (long)b * H * W;
#pragma unroll
                for (int i = 0; i < (PH * PW + 63) / 64; ++i) {
                    const int e = i * 64 + lane, pi = e / PW, pj = e % PW;
                    const int t = t0 + pi - 1, d = d0 + pj - 1;
                    const float* src = (t >= 0 && t < H && d >= 0 && d < W) ? x_b + (long)t * W + d : zx;
                    if (e < PH * PW) __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(sxp + i * 64), 4, 0, 0);
                }
            }
            const bf16* org = a.in + ((long)b * H * W + (long)t0 * W + d0) * CIN;
            const bool interior = t0 >= 1 && t0 + TH + 1 <= H && d0 >= 1 && d0 + TW + 1 <= W;
            char* dst = pbuf + (g & 1) * PBYTES;
            if (interior) {
#pragma unroll
                for (int i = 0; i < PPIECES; ++i)
                    __builtin_amdgcn_global_load_lds((gptr_t*)(org + rel[i]), (lptr_t*)(dst + i * 1024), 16, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < PPIECES; ++i) {
                    const int t = t0 + (pij[i] >> 8) - 1, d = d0 + (pij[i] & 255) - 1;
                    const bf16* src = (t >= 0 && t < H && d >= 0 && d < W) ? org + rel[i] : zsrc;
                    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + i * 1024), 16, 0, 0);
                }
            }
        };
        // tap image of the tile whose neighbourhood sits in sxp: lane = pixel row pr, columns 4c .. 4c+3.  Column kappa(pixel) of the
        // image undoes the k-permutation of frag_rm (MFMA k index 8g+j <-> slab row 4g+j for j < 4, 16+4g+(j-4) for j >= 4): the four
        // pixels of a group are four CONSECUTIVE columns.  Pixels beyond the image edge need no masking: their dY rows are zero.
        const int gpr = lane >> 2, gc = lane & 3;
        const int xt_off = (gpr >> 1) * 32 + 8 * gc + 4 * (gpr & 1);
        auto build_tap_image = [&]() {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const f32x2* row = reinterpret_cast<const f32x2*>(sxp + (gpr + dy) * PW + 4 * gc);
                const f32x2 v0 = row[0], v1 = row[1], v2 = row[2];
                const float w[6] = {v0[0], v0[1], v1[0], v1[1], v2[0], v2[1]};
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    bf16x4 o;
                    o[0] = (bf16)w[dx]; o[1] = (bf16)w[dx + 1]; o[2] = (bf16)w[dx + 2]; o[3] = (bf16)w[dx + 3];
                    *reinterpret_cast<bf16x4*>(sxt + (dy * 3 + dx) * XS + xt_off) = o;
                }
            }
        };
        bool dry = false;
        // Tiles are dealt by a STATIC stride (workgroup w: tiles w, w + G, ...), not by the tile counter: the wave-level partial sums
        // run over all tiles of a workgroup, so the assignment fixes the summation order -- the gradient is bit-reproducible.
        unsigned own = blockIdx.x;
        const unsigned lim = (unsigned)ntiles;
        auto issue_fetch = [&]() -> unsigned { const unsigned t = own; own += gridDim.x; return t; };
        int cur, nxt;
        {
            const unsigned t0f = issue_fetch(), t1f = issue_fetch();
            cur = t0f < lim ? (int)t0f : -1;
            nxt = t1f < lim ? (int)t1f : -1;
            if (nxt < 0) dry = true;
        }
        if (lane == 0) { tileq[0] = cur; tileq[1] = nxt; }
        if (cur >= 0) issue_patch(cur, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");           // opening barrier: weights (every wave's share), patch 0, tileq[0..1]
        for (int k = 0; cur >= 0; ++k) {
            // tile k is being computed: its tap image first (frees sxp), then tile k+1's patch + neighbourhood, publish, hand over
            const unsigned raw = issue_fetch();
            build_tap_image();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (nxt >= 0) issue_patch(nxt, k + 1);
            int nn = -1;
            if (!dry) {
                if (raw >= lim) dry = true; else nn = (int)raw;
            }
            if (lane == 0) tileq[(k + 2) & 3] = nn;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");       // barrier k (middle of tile k's last tap)
            asm volatile("s_barrier" ::: "memory");       // E1: the masked tile is staged (in patch k's buffer)
            asm volatile("s_barrier" ::: "memory");       // E3: contraction done -- patch k's buffer and the tap image are free again
            cur = nxt; nxt = nn;
        }
        return;
    }

    // ---------------- MFMA waves
    const int rr = lane & 15, q = lane >> 4;
    const int pcol0 = rr % TW, prow0 = wave * MF * RPT + rr / TW;
    int poff[3][2], woff[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        woff[kc] = rr * 128 + (((kc * 4 + q) ^ (rr & 7)) * 16);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) poff[dx][kc] = (prow0 * PW + pcol0 + dx) * 128 + (((kc * 4 + q) ^ ((pcol0 + dx) & 7)) * 16);
    }
    struct Frags { bf16x8 a[MF], b[NF]; };
    typedef __attribute__((address_space(3))) const char lds_cchar;
    const unsigned wbase = (unsigned)(size_t)((lds_cchar*)wres);
    f32x4 acc[MF][NF];
    // half_tap: MF*NF MFMAs on `use`, the reads of `ld` issued in their shadow (see the streaming kernel)
    auto half_tap = [&](const Frags& use, Frags& ld, const char* pl, int tap, int kc) {
        const int dy = tap / 3, dx = tap % 3;
        const unsigned pa = (unsigned)(size_t)((lds_cchar*)pl) + poff[dx][kc];
        const unsigned wa = wbase + woff[kc] + tap * WBYTES;   // (the DS offset field is 16 bits: the tap's 8 KiB stride goes here)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MF * NF; ++m) {
            const int i = m / NF, j = m % NF;
            acc[i][j] = mma16(use.b[j], use.a[i], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (m < MF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld.a[m < MF ? m : 0]) : "v"(pa), "n"(((m < MF ? m : 0) * RPT + dy) * PW * 128));
            else if (m < MF + NF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld.b[m < MF + NF ? m - MF : 0]) : "v"(wa), "n"((m < MF + NF ? m - MF : 0) * 2048));
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    Frags f0, f1;
    f32x4 cw = {0.f, 0.f, 0.f, 0.f};                    // conv1 weight-gradient partial of this wave over all its tiles
    const int wtap = lane & 15;
    bf16x8 cfrag;                                         // B fragment of the rows that are not in LDS: tap 9 = ones (bias gradient), 10.. = zeros
#pragma unroll
    for (int e = 0; e < 8; ++e) cfrag[e] = (bf16)(wtap == 9 ? 1.f : 0.f);
    const bf16* const xt_row = sxt + (wtap < 9 ? wtap : 8) * XS + q * 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the weights
    asm volatile("s_barrier" ::: "memory");               // opening barrier
    int tile = read_tileq(0);
    {   // fragments of (tap 0, first k-half)
        const unsigned pa = (unsigned)(size_t)((lds_cchar*)pbuf) + poff[0][0];
        const unsigned wa = wbase + woff[0];
#pragma unroll
        for (int i = 0; i < MF; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f0.a[i]) : "v"(pa), "n"(i * RPT * PW * 128));
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f0.b[j]) : "v"(wa), "n"(j * 2048));
    }
    for (int k = 0; tile >= 0; ++k) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int d0 = tx * TW, t0 = ty * TH;
        const int tl = t0 + prow0, dl = d0 + pcol0;
        // ReLU mask of conv1's output as one 64-bit word per pixel (bit c = channel c passed; written by conv1's forward
        // launch, mk_conv1_fwd): this lane's pixel of row i, requested at the first tap -- four 8-byte loads per wave
        // and tile where the bf16 map itself cost eight 16-byte loads, in a launch that is bound by vector-memory instructions at the
        // CU's one texture addresser (measured: 122 us with the map, 77 us with no mask at all, 89 us with the words)
        const uint2* bits_b = reinterpret_cast<const uint2*>(a.mask_bits) + (long)b * H * W + (long)tl * W + dl;
        auto row_ok = [&](int i) { return tl + i * RPT < H && dl < W; };
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        char* const pcur = pbuf + (k & 1) * PBYTES;
        const char* pnext = pbuf + ((k + 1) & 1) * PBYTES;
        uint2 mb[MF];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap == MASK_TAP) {
#pragma unroll
                for (int i = 0; i < MF; ++i) mb[i] = row_ok(i) ? bits_b[(long)i * RPT * W] : uint2{0u, 0u};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            half_tap(f0, f1, pcur, tap, 1);
            if (tap < 8) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                half_tap(f1, f0, pcur, tap + 1, 0);
            } else {
                // barrier k: the next tile's patch, this tile's tap image and tileq[k+1..k+2] are in; no wave reads patch k again
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                half_tap(f1, f0, pnext, 0, 0);            // (after the last tile: stale LDS, never used)
            }
        }
        tile = read_tileq(k + 1);

        // epilogue: the masked bf16 tile (zeros outside the image) goes into the consumed patch buffer and is contracted there with
        // the tap image: [64 channels] x [9 taps + ones] over the tile's 256 pixels, on the MFMA
        bf16* const otile = reinterpret_cast<bf16*>(pcur);
        {
#pragma unroll
        for (int i = 0; i < MF; ++i) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                u32x4 ov;
                // byte q of the word's half h = this lane's 8 channels; bits 2kk, 2kk+1 -> the two halves of a packed pair
                const int byte = (int)((h ? mb[i].y : mb[i].x) >> (8 * q));
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4& c = acc[i][2 * h + (kk >> 1)];
                    unsigned r;
                    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(c[2 * (kk & 1)]), "v"(c[2 * (kk & 1) + 1]));
                    const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(byte, 2 * kk, 1), hi = (unsigned)__builtin_amdgcn_sbfe(byte, 2 * kk + 1, 1);
                    ov[kk] = r & ((lo & 0xFFFFu) | (hi & 0xFFFF0000u));      // (rows outside the image: their word is 0)
                }
                *reinterpret_cast<u32x4*>(otile + ((wave * MF + i) * 16 + rr) * OS + h * 32 + q * 8) = ov;
            }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // E1
        // wave w: output channels 16w .. 16w+15 x 16 taps, contraction over the tile's pixels (slabs of 32)
        {
#pragma unroll
        for (int sl = 0; sl < NPIX / 32; ++sl) {
            const bf16x8 af = frag_rm<OS>(otile + sl * 32 * OS, wave * 16, lane);
            const bf16x8 xb = ld8(xt_row + sl * 32);
            cw = mma16(af, wtap < 9 ? xb : cfrag, cw);
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // E3
    }
    if ((lane & 15) < 10) {                                // one row of 640 partial sums per workgroup (mk_conv1_wgrad_fused_reduce)
#pragma unroll
        for (int r = 0; r < 4; ++r) a.w1_slab[(long)blockIdx.x * 640 + (wave * 16 + 4 * (lane >> 4) + r) * 10 + (lane & 15)] = cw[r];
    }
}

// ------------------------------------------------------------------ wgrad (reduction over pixels)

// ------------------------------------------------------------------ wgrad v2: patch-tiled, persistent, all 9 taps per workgroup
// dW[co][tap][ci] = sum_pixels dy[pix][co] * x[pix + off(tap)][ci].  A workgroup owns one (64-ci slab, 64-co half) and walks
// over 8x16-pixel tiles of the batch: the x patch (10x18 pixels, zero outside the image) and the dy tile go to LDS once
// and feed ALL 9 taps (both operands via ds_read_b64_tr_b16, contraction over the pixel index), so L2/HBM traffic per
// FLOP is 9x lower than one-tap-per-workgroup.  The 9 x 64 x 64 fp32 partial stays in registers (144 VGPRs/lane) across
// all tiles of the workgroup and is written once; a fixed-order slab reduce makes the result deterministic.
constexpr int W2_PS = 80, W2_LDY = 80;   // 160-B rows: tr-reads conflict-free
constexpr int W2_TOTAL_WG = 256;    // one persistent workgroup per CU (see mk_conv3x3_wgrad)

// W2_TW: tile width in pixels (16, or 8 for maps whose width pads badly to 16); tile = 128 pixels
// POOLED: the dy tile comes from ConvWgradArgs::dy_pooled + pool_idx: a thread fetches ONE pooled cell x 8 channels (16 B + 8 B of codes) and
// stages the four positions of its window (4 + 0 instead of 4 x 16-byte loads per thread and tile)
template <int CIN, int COUT, int W2_TW, bool POOLED = false>
__global__ __launch_bounds__(512) void conv3x3_wgrad2_kernel(ConvWgradArgs a, int nwg, int ntiles, int tiles_x, int tiles_y) {
    constexpr int W2_TH = 128 / W2_TW, W2_PH = W2_TH + 2, W2_PW = W2_TW + 2, RPS = 32 / W2_TW;      // RPS = pixel rows per 32-pixel slab
    constexpr int KTOT = 9 * CIN;
    constexpr int NPCH = (W2_PH * W2_PW * 8 + 255) / 256;   // patch chunks per thread (6)
    constexpr int NDCH = W2_TH * W2_TW * 8 / 256;           // dy chunks per thread (4)
    // one workgroup per CU: TWO tile buffers -- the next tile is written to LDS at the start of a tile's MFMA phase instead of behind
    // it (staging between two barriers with the MFMA pipes idle cost a third of the launch: 135 us against 90 for 128 -> 128 channels)
    constexpr int NB = 2, PATCH_EL = W2_PH * W2_PW * W2_PS, DYT_EL = W2_TH * W2_TW * W2_LDY;
    __shared__ __attribute__((aligned(16))) bf16 patch_[NB * PATCH_EL];
    __shared__ __attribute__((aligned(16))) bf16 dyt_[NB * DYT_EL];

    // 512 threads in two roles -- waves 0-3 multiply (wave w: input channels 16 w ..), waves 4-7 fetch and stage the next tile; each SIMD
    // holds one wave of each role, so a tile's ~220 vector instructions of staging (pooled-dy expansion, zero fill, LDS stores, address
    // arithmetic) and its load waits run BESIDE the MFMAs instead of in front of them in the same wave.  tid = index inside the role.
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const bool producer = threadIdx.x >= 256;
    const int cs = blockIdx.y % (CIN / 64), ch = blockIdx.y / (CIN / 64);
    const int H = a.H, W = a.W;

    // one tile's fetch, in registers between request and staging (TWO of them are kept in flight)
    struct Stage { bf16x8 pr[NPCH], dr[NDCH]; uint2 pcode; unsigned okp, okd; };
    // Per-thread constants of the tile fetch: chunk i of the patch is pixel (ppi, ppj) relative to the tile origin (t0, d0) and sits prel
    // elements behind that pixel's address; per tile only the origin (wave-uniform: scalar ALU) and two range tests per chunk remain.  (With
    // the divisions, 64-bit multiplies and selects of the first version the fetch cost ~200 vector instructions per tile, which a lone
    // workgroup per CU has nobody to hide behind: 135 us against 90 with two per CU on the 128 -> 128 layer.)
    int ppi[NPCH], ppj[NPCH], prel[NPCH];
#pragma unroll
    for (int i = 0; i < NPCH; ++i) {
        const int c = tid + i * 256, pix = c >> 3;
        ppi[i] = pix / W2_PW - 1; ppj[i] = pix % W2_PW - 1;
        prel[i] = (ppi[i] * W + ppj[i]) * CIN + (c & 7) * 8;
        if (c >= W2_PH * W2_PW * 8) { ppi[i] = -(1 << 20); prel[i] = 0; }    // (past the patch: never in range)
    }
    int dpi[NDCH], dpj[NDCH], drel[NDCH];
#pragma unroll
    for (int i = 0; i < NDCH; ++i) {
        const int c = tid + i * 256, pix = c >> 3;
        dpi[i] = pix / W2_TW; dpj[i] = pix % W2_TW;
        drel[i] = (dpi[i] * W + dpj[i]) * COUT + (c & 7) * 8;
    }
    const int pp_r = (tid >> 3) / (W2_TW / 2), pp_c = (tid >> 3) % (W2_TW / 2), H2 = H / 2, W2p = W / 2;
    const int pprel = (pp_r * W2p + pp_c) * COUT + (tid & 7) * 8;
    // the tiles of a workgroup are fetched in walk order (first, first + nwg, ...): their (x, y, image) coordinates advance by a fixed
    // step with two carries instead of three integer divisions per tile (~90 instructions of the staging waves' ~370 per tile)
    int wt = blockIdx.x, wx = wt % tiles_x, wy = (wt / tiles_x) % tiles_y, wb = wt / (tiles_x * tiles_y);
    const int sx = nwg % tiles_x, sy = (nwg / tiles_x) % tiles_y, sb = nwg / (tiles_x * tiles_y);
    const int lx = (ntiles - 1) % tiles_x, ly = ((ntiles - 1) / tiles_x) % tiles_y, lb = (ntiles - 1) / (tiles_x * tiles_y);
    auto load_tile = [&](Stage& S, int tile) {
        (void)tile;                                                     // (== wt: callers walk first, first + nwg, ... in order)
        const bool past = wt >= ntiles;
        const int tx = past ? lx : wx, ty = past ? ly : wy, b = past ? lb : wb;
        wt += nwg; wx += sx;
        { const int c = wx >= tiles_x; wx -= c ? tiles_x : 0; wy += sy + c; }
        { const int c = wy >= tiles_y; wy -= c ? tiles_y : 0; wb += sb + c; }
        const int t0 = ty * W2_TH, d0 = tx * W2_TW;
        const bf16* in_o = a.in + (((long)b * H + t0) * W + d0) * CIN + cs * 64;             // the tile's origin pixel (always inside the map)
        S.okp = 0; S.okd = 0;
#pragma unroll
        for (int i = 0; i < NPCH; ++i) {
            const bool ok = (unsigned)(t0 + ppi[i]) < (unsigned)H && (unsigned)(d0 + ppj[i]) < (unsigned)W;
            if (ok) S.okp |= 1u << i;
            S.pr[i] = ld8(in_o + (ok ? prel[i] : (tid & 7) * 8));
        }
        if constexpr (POOLED) {
            const int t2 = t0 / 2 + pp_r, d2 = d0 / 2 + pp_c;
            const bool ok = t2 < H2 && d2 < W2p;                            // (floor mode: the cropped last row / column gets no gradient)
            if (ok) S.okd = 1u;
            // (origin cell of the tile: inside the pooled map unless the map's last row / column is the cropped one -- then ok is false for
            // every cell of the tile and the clamped origin is read)
            const int t2o = t0 / 2 < H2 ? t0 / 2 : H2 - 1, d2o = d0 / 2 < W2p ? d0 / 2 : W2p - 1;
            const long po = (((long)b * H2 + t2o) * W2p + d2o) * COUT + ch * 64;
            const int rel = ok ? pprel : (tid & 7) * 8;
            S.dr[0] = ld8(a.dy_pooled + po + rel);
            S.pcode = *reinterpret_cast<const uint2*>(a.pool_idx + po + rel);
        } else {
            const bf16* dy_o = a.dy + (((long)b * H + t0) * W + d0) * COUT + ch * 64;
#pragma unroll
            for (int i = 0; i < NDCH; ++i) {
                const bool ok = t0 + dpi[i] < H && d0 + dpj[i] < W;
                if (ok) S.okd |= 1u << i;
                S.dr[i] = ld8(dy_o + (ok ? drel[i] : (tid & 7) * 8));
            }
        }
    };
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool do_db = a.db != nullptr && cs == 0;
    // dbm: 1.0 when this tile's dy counts towards the bias gradient (branch-free: the fetch / staging / MFMA code of a tile is ONE basic block, so
    // that the scheduler can spread the vector work of the staging over the MFMA phase)
    auto store_tile = [&](const Stage& S, bf16* patch, bf16* dyt, float dbm) {
#pragma unroll
        for (int i = 0; i < NPCH; ++i) {
            const int c = tid + i * 256;
            if (c < W2_PH * W2_PW * 8) st8(patch + (c >> 3) * W2_PS + (c & 7) * 8, (S.okp >> i) & 1 ? S.pr[i] : zero8());
        }
        if constexpr (POOLED) {
            const int pp = tid >> 3, pr2 = pp / (W2_TW / 2), pc2 = pp % (W2_TW / 2);
            bf16x8 o[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned cj = ((j < 4 ? S.pcode.x : S.pcode.y) >> (8 * (j & 3))) & 0xffu;
                const bf16 gj = S.okd ? S.dr[0][j] : (bf16)0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k][j] = cj == (unsigned)k ? gj : (bf16)0.f;
                csum[j] = fmaf(dbm, cj < 4u ? (float)gj : 0.f, csum[j]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) st8(dyt + ((2 * pr2 + (k >> 1)) * W2_TW + 2 * pc2 + (k & 1)) * W2_LDY + (tid & 7) * 8, o[k]);
        } else {
#pragma unroll
        for (int i = 0; i < NDCH; ++i) {
            const int c = tid + i * 256;
            const bf16x8 v = (S.okd >> i) & 1 ? S.dr[i] : zero8();
            st8(dyt + (c >> 3) * W2_LDY + (c & 7) * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) csum[j] = fmaf(dbm, (float)v[j], csum[j]);
        }
        }
    };

    // wave w owns input channels 16w .. 16w+15 x all 64 output channels x 9 taps: the four dy fragments of a pixel slab are
    // tap-independent (read once per slab), only ONE x fragment is read per tap: 26 transposing reads per 36 MFMAs
    // (the 2 x 2 wave grid needed 40)
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const int g = lane >> 4, q = (lane & 15) >> 2, p4 = (lane & 3) * 4;
    const float dbm1 = do_db ? 1.f : 0.f;
    float* red = reinterpret_cast<float*>(patch_);                        // bias-gradient partials [256][8] floats = 8 KB (after the last tile)

    auto read_a = [&](const bf16* dyt, int kc, bf16x8 (&af)[4]) {
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const bf16* a0 = dyt + (kc * 32 + 4 * g + q) * W2_LDY + fm * 16 + p4;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a0);
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a0 + 16 * W2_LDY));
            af[fm][0] = lo[0]; af[fm][1] = lo[1]; af[fm][2] = lo[2]; af[fm][3] = lo[3];
            af[fm][4] = hi[0]; af[fm][5] = hi[1]; af[fm][6] = hi[2]; af[fm][7] = hi[3];
        }
    };
    auto read_b = [&](const bf16* patch, int kc, int tap) {
        const int dyi = tap / 3, dxj = tap % 3;
        const int pi = 4 * g + q;                               // pixel of the slab's first half this lane addresses
        const bf16* b0 = patch + ((RPS * kc + pi / W2_TW + dyi) * W2_PW + pi % W2_TW + dxj) * W2_PS + wave * 16 + p4;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)b0);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(b0 + (16 / W2_TW) * W2_PW * W2_PS));
        bf16x8 bfr;
        bfr[0] = lo[0]; bfr[1] = lo[1]; bfr[2] = lo[2]; bfr[3] = lo[3];
        bfr[4] = hi[0]; bfr[5] = hi[1]; bfr[6] = hi[2]; bfr[7] = hi[3];
        return bfr;
    };
    // wave w owns input channels 16w .. 16w+15 x all 64 output channels x 9 taps: the four dy fragments of a pixel slab are
    // tap-independent (read once per slab), only ONE x fragment is read per tap: 26 transposing reads per 36 MFMAs
    // (the 2 x 2 wave grid needed 40)
    // The partial slab is written in REGISTER order -- element ((tap * 4 + fm) * 256 + thread) * 4 + r of this (64-ci, 64-co) block's 36 864 --
    // 36 sixteen-byte stores per lane, a kilobyte per wave-instruction (in [co][tap][ci] order the same 147 KB left as 144 dword stores per
    // lane in 64-byte pieces, at the end of the launch with every CU storing at once); conv3x3_wgrad_reduce_body un-permutes (folds.h)
    auto write_slab = [&](f32x4 (&acc)[9][4]) {
        float* out = a.slab + (long)blockIdx.x * COUT * KTOT + (long)blockIdx.y * (64 * 9 * 64) + tid * 4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm) *reinterpret_cast<f32x4*>(out + (tap * 4 + fm) * 1024) = acc[tap][fm];
    };
    auto fold_db = [&]() {                                       // threads with equal tid % 8 own the same 8 output channels
        const int rc = tid / 8, j = tid % 8;
        float sum = 0.f;
        for (int t = rc; t < 256; t += 8) sum += red[t * 8 + j];
        a.slab[(long)nwg * COUT * KTOT + (long)blockIdx.x * COUT + ch * 64 + tid] = sum;
    };

    {
        // The two roles are two separate loops with the SAME sequence of barriers (one in front of the first tile, one behind every tile,
        // one behind the bias partials): the accumulators live only in the consumers' branch and the staging registers only in the
        // producers', so one register budget (256 per thread at two waves per SIMD) holds either.
        const int first = blockIdx.x, niter = first < ntiles ? (ntiles - first + nwg - 1) / nwg : 0;
        if (producer) {
            // TWO tiles in flight: a tile's loads are requested two tile periods before they are staged (with one, the producers alone ran
            // at a memory round trip per tile, ~2 us, and set the pace of the launch).  Stage A carries the odd tiles of the walk, B the
            // even ones; behind the last tile the registers hold clamped re-reads, staged into the buffer nobody reads any more, not counted.
            Stage A, B;
            A.pcode = B.pcode = uint2{0u, 0u};
            if (niter) { load_tile(B, first); load_tile(A, first + nwg); store_tile(B, patch_, dyt_, dbm1); load_tile(B, first + 2 * nwg); }
            __syncthreads();
            for (int it = 0; it < niter; it += 2) {
                // consumers read buffer it & 1 = 0: tile it + 1 (A) goes into buffer 1, then A requests tile it + 3
                const int tile = first + it * nwg;
                store_tile(A, patch_ + PATCH_EL, dyt_ + DYT_EL, tile + nwg < ntiles ? dbm1 : 0.f);
                load_tile(A, tile + 3 * nwg);
                __syncthreads();
                if (it + 1 < niter) {
                    // consumers read buffer 1: tile it + 2 (B) goes into buffer 0, then B requests tile it + 4
                    store_tile(B, patch_, dyt_, tile + 2 * nwg < ntiles ? dbm1 : 0.f);
                    load_tile(B, tile + 4 * nwg);
                    __syncthreads();
                }
            }
            if (do_db) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[tid * 8 + j] = csum[j];
            }
            __syncthreads();
        } else {
            f32x4 acc[9][4];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
            for (int it = 0; it < niter; ++it) {
                const bf16* patch = patch_ + (it & 1) * PATCH_EL;
                const bf16* dyt = dyt_ + (it & 1) * DYT_EL;
                // 36 steps (4 pixel slabs x 9 taps) of one x fragment x four dy fragments.  One wave per SIMD multiplies, so nobody else
                // covers the LDS latency: the x fragments run THREE steps ahead through a ring of four (one step = 4 MFMAs = 64 cycles; at
                // one step ahead every tap waited ~a step for its fragment), the dy fragments of the next slab arrive during tap 4.
                // The order is pinned (left alone the scheduler folds the ring into read-and-use-at-once).
                bf16x8 af[2][4], br[4];
                read_a(dyt, 0, af[0]);
#pragma unroll
                for (int t = 0; t < 3; ++t) br[t] = read_b(patch, 0, t);
                __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
#define CW_STEP(s_)                                                                                                                  \
                {                                                                                                                    \
                    constexpr int kc_ = (s_) / 9, tap_ = (s_) % 9;                                                                   \
                    if ((s_) + 3 < 36) br[((s_) + 3) & 3] = read_b(patch, ((s_) + 3) / 9, ((s_) + 3) % 9);                              \
                    if (tap_ == 4 && kc_ < 3) read_a(dyt, kc_ + 1, af[(kc_ + 1) & 1]);                                                \
                    _Pragma("unroll") for (int fm = 0; fm < 4; ++fm) acc[tap_][fm] = mma16(af[kc_ & 1][fm], br[(s_) & 3], acc[tap_][fm]); \
                    __builtin_amdgcn_sched_group_barrier(0x100, ((s_) + 3 < 36 ? 2 : 0) + (tap_ == 4 && kc_ < 3 ? 8 : 0), 0);         \
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                               \
                }
                CW_STEP(0) CW_STEP(1) CW_STEP(2) CW_STEP(3) CW_STEP(4) CW_STEP(5) CW_STEP(6) CW_STEP(7) CW_STEP(8)
                CW_STEP(9) CW_STEP(10) CW_STEP(11) CW_STEP(12) CW_STEP(13) CW_STEP(14) CW_STEP(15) CW_STEP(16) CW_STEP(17)
                CW_STEP(18) CW_STEP(19) CW_STEP(20) CW_STEP(21) CW_STEP(22) CW_STEP(23) CW_STEP(24) CW_STEP(25) CW_STEP(26)
                CW_STEP(27) CW_STEP(28) CW_STEP(29) CW_STEP(30) CW_STEP(31) CW_STEP(32) CW_STEP(33) CW_STEP(34) CW_STEP(35)
#undef CW_STEP
                __syncthreads();
            }
            write_slab(acc);
            __syncthreads();
            if (do_db && tid < 64) fold_db();
        }
    }
}

// dw[co][ci][3][3] = sum over the per-workgroup partial slabs; 64 outputs x 4 slab-lanes per workgroup, four independent
// partial sums per thread (fixed order -> deterministic); the trailing COUT entries are the bias gradient
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce(const float* __restrict__ slab, int nsplit, float* __restrict__ dw,
                                                            float* __restrict__ db, int CIN, int COUT) {
    conv3x3_wgrad_reduce_body(slab, nsplit, dw, db, CIN, COUT, blockIdx.x);
}

// ------------------------------------------------------------------ max-pool 2x2 (floor) NHWC
// ceil_mode: output (H+1)/2 x (W+1)/2, windows clipped at the map edge (MaxPool2d(2, 2, ceil_mode=True) of the BLSTM front-end)
__global__ void maxpool_fwd_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int B, int H, int W, int C, int ceil_mode) {
    const int H2 = ceil_mode ? (H + 1) / 2 : H / 2, W2 = ceil_mode ? (W + 1) / 2 : W / 2, C8 = C / 8;
    const long n = (long)B * H2 * W2 * C8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i % C8);
    const int d2 = (int)((i / C8) % W2);
    const int t2 = (int)((i / ((long)C8 * W2)) % H2);
    const int b = (int)(i / ((long)C8 * W2 * H2));
    const bf16* base = in + (((long)b * H + 2 * t2) * W + 2 * d2) * C + c8 * 8;
    const bool r1 = 2 * t2 + 1 < H, c1 = 2 * d2 + 1 < W;     // second row / column of the window exists
    const bf16x8 v00 = ld8(base), v01 = c1 ? ld8(base + C) : v00, v10 = r1 ? ld8(base + (long)W * C) : v00,
                 v11 = (r1 && c1) ? ld8(base + (long)W * C + C) : v00;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = (bf16)fmaxf(fmaxf((float)v00[j], (float)v01[j]), fmaxf((float)v10[j], (float)v11[j]));
    st8(out + i * 8, o);
}
__global__ void maxpool_relu_bwd_kernel(const bf16* __restrict__ in, const bf16* __restrict__ dout, bf16* __restrict__ din,
                                        int B, int H, int W, int C, int ceil_mode) {
    const int H2 = ceil_mode ? (H + 1) / 2 : H / 2, W2 = ceil_mode ? (W + 1) / 2 : W / 2, C8 = C / 8;
    const int H2c = (H + 1) / 2, W2c = (W + 1) / 2;           // cells incl. the cropped edge
    const long n = (long)B * H2c * W2c * C8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i % C8);
    const int d2 = (int)((i / C8) % W2c);
    const int t2 = (int)((i / ((long)C8 * W2c)) % H2c);
    const int b = (int)(i / ((long)C8 * W2c * H2c));
    const long base = (((long)b * H + 2 * t2) * W + 2 * d2) * C + c8 * 8;
    if (t2 < H2 && d2 < W2 && !(2 * t2 + 1 < H && 2 * d2 + 1 < W)) {
        // ceil_mode window clipped by the map edge: arg-max over the elements that exist
        const bool ex[4] = {true, 2 * d2 + 1 < W, 2 * t2 + 1 < H, 2 * t2 + 1 < H && 2 * d2 + 1 < W};
        const long offs[4] = {0, (long)C, (long)W * C, (long)W * C + C};
        bf16x8 v[4];
        for (int k = 0; k < 4; ++k) v[k] = ex[k] ? ld8(in + base + offs[k]) : zero8();
        const bf16x8 g = ld8(dout + ((((long)b * H2 + t2) * W2 + d2) * C + c8 * 8));
        bf16x8 o[4];
        for (int j = 0; j < 8; ++j) {
            int arg = 0; float mx = (float)v[0][j];
            for (int k = 1; k < 4; ++k) if (ex[k] && (float)v[k][j] > mx) { mx = (float)v[k][j]; arg = k; }
            for (int k = 0; k < 4; ++k) o[k][j] = (k == arg && mx > 0.f) ? g[j] : (bf16)0.f;
        }
        for (int k = 0; k < 4; ++k) if (ex[k]) st8(din + base + offs[k], o[k]);
    } else if (t2 < H2 && d2 < W2) {
        const long offs[4] = {0, (long)C, (long)W * C, (long)W * C + C};
        bf16x8 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ld8(in + base + offs[k]);
        const bf16x8 g = ld8(dout + ((((long)b * H2 + t2) * W2 + d2) * C + c8 * 8));
        bf16x8 o[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int arg = 0; float mx = (float)v[0][j];            // first max in scan order (strict >), as torch
#pragma unroll
            for (int k = 1; k < 4; ++k) if ((float)v[k][j] > mx) { mx = (float)v[k][j]; arg = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][j] = (k == arg && mx > 0.f) ? g[j] : (bf16)0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) st8(din + base + offs[k], o[k]);
    } else {
        // cells outside every pooling window (odd H or W): gradient is zero
        for (int dt = 0; dt < 2; ++dt)
            for (int dd = 0; dd < 2; ++dd)
                if (2 * t2 + dt < H && 2 * d2 + dd < W) st8(din + base + ((long)dt * W + dd) * C, zero8());
    }
}

// the codes from a stored (ReLU'd) map: for the conv kernels that do not emit them in their epilogue
__global__ void maxpool_idx_kernel(const bf16* __restrict__ in, uint8_t* __restrict__ idx, int B, int H, int W, int C) {
    const int H2 = H / 2, W2 = W / 2, C8 = C / 8;
    const long n = (long)B * H2 * W2 * C8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i % C8);
    const int d2 = (int)((i / C8) % W2);
    const int t2 = (int)((i / ((long)C8 * W2)) % H2);
    const int b = (int)(i / ((long)C8 * W2 * H2));
    const bf16* base = in + (((long)b * H + 2 * t2) * W + 2 * d2) * C + c8 * 8;
    const bf16x8 v[4] = {ld8(base), ld8(base + C), ld8(base + (long)W * C), ld8(base + (long)W * C + C)};
    unsigned w[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int arg = 0; float mx = (float)v[0][j];
#pragma unroll
        for (int k = 1; k < 4; ++k) if ((float)v[k][j] > mx) { mx = (float)v[k][j]; arg = k; }
        w[j >> 2] |= (unsigned)(mx > 0.f ? arg : 4) << (8 * (j & 3));
    }
    *reinterpret_cast<uint2*>(idx + i * 8) = uint2{w[0], w[1]};
}

}  // namespace

static int conv_ncu();
// conv1 forward: exactly the workgroups that are resident together, every wave taking one contiguous run of the 16-pixel row segments
static long conv1_mfma_grid(int B, int H, int W) {
    static const int resident = [] {
        int per_cu = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1_fwd_mfma_kernel, 256, 0);
        return conv_ncu() * (per_cu > 0 ? per_cu : 4);
    }();
    const long blocks = ((long)B * H * ((W + 15) / 16) + 3) / 4;
    return blocks < resident ? blocks : resident;
}
int mk_conv1_fwd(const float* x, const float* w, const float* bias, bf16* out, int B, int H, int W, hipStream_t s, unsigned long long* relu_bits) {
    const long P = (long)B * H * W;
    if (P * 64 * 2 >= (1L << 32) - 65536) { mk_set_error("mk_conv1_fwd", "map too large (the output is addressed through one 4 GB buffer descriptor)"); return -1; }
    hipLaunchKernelGGL(conv1_fwd_mfma_kernel, dim3((unsigned)conv1_mfma_grid(B, H, W)), dim3(256), 0, s, x, w, bias, out, B, H, W, 64, reinterpret_cast<unsigned short*>(relu_bits));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
long mk_conv1_wgrad_slab_floats(int B, int H, int W) { return (((long)B * H * W + C1_PIX - 1) / C1_PIX + C1_RSPLIT) * 640; }

// CIN = 1 conv with COUT = 64 * n channels (the BLSTM front-end has 128): n launches of the 64-channel kernels
int mk_conv1_fwd_n(const float* x, const float* w, const float* bias, bf16* out, int B, int H, int W, int COUT, hipStream_t s) {
    if (COUT % 64) { mk_set_error("mk_conv1_fwd_n", "COUT must be a multiple of 64"); return -1; }
    const long P = (long)B * H * W;
    if (P * COUT * 2 >= (1L << 32) - 65536) { mk_set_error("mk_conv1_fwd_n", "map too large (the output is addressed through one 4 GB buffer descriptor)"); return -1; }
    for (int c0 = 0; c0 < COUT; c0 += 64)
        hipLaunchKernelGGL(conv1_fwd_mfma_kernel, dim3((unsigned)conv1_mfma_grid(B, H, W)), dim3(256), 0, s, x, w + c0 * 9, bias + c0, out + c0, B, H, W, COUT, (unsigned short*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int mk_conv1_wgrad_n(const float* x, const bf16* dy, float* dw, float* db, float* slab, int B, int H, int W, int COUT, hipStream_t s) {
    if (COUT % 64) { mk_set_error("mk_conv1_wgrad_n", "COUT must be a multiple of 64"); return -1; }
    const long P = (long)B * H * W;
    const int nb = (int)((P + C1_PIX - 1) / C1_PIX);
    for (int c0 = 0; c0 < COUT; c0 += 64) {
        hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(nb), dim3(256), 0, s, x, dy + c0, slab, B, H, W, COUT);
        launch_conv1_reduce(slab, nb, dw + c0 * 9, db + c0, s);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

static int conv_ncu() {                                         // (task-slot threads call this concurrently: a thread-safe one-time initialisation)
    static const int ncu = [] {
        int dev = 0, n = 0;
        hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return ncu;
}
// workgroups (= rows of 640 partial sums) of the fused conv1-wgrad dgrad: one persistent workgroup per CU
static int resw_w1_rows(int B, int H, int W) {
    const long ntiles = (long)B * ((H + 15) / 16) * ((W + 15) / 16);
    return (int)(ntiles < conv_ncu() ? ntiles : conv_ncu());
}
long mk_conv1_wgrad_fused_slab_floats(int B, int H, int W) { return ((long)resw_w1_rows(B, H, W) + C1_RSPLIT) * 640; }
int mk_conv1_wgrad_fused_reduce(float* slab, int B, int H, int W, float* dw, float* db, hipStream_t s) {
    launch_conv1_reduce(slab, resw_w1_rows(B, H, W), dw, db, s);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// tile counters of launches that bring none (single-stream tools and tests)
static unsigned* sched_or_fallback(unsigned* sched) {
    if (sched) return sched;
    static unsigned* const fallback = [] { unsigned* p = nullptr; hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_conv_sched)); return p; }();
    return fallback;
}
static int launch_resw_w1(const ConvArgs& a, hipStream_t s) {
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16, ntiles = tiles_x * tiles_y * a.B;
    if (!a.mask_bits) { mk_set_error("mk_conv3x3", "fused conv1 wgrad needs ConvArgs::mask_bits"); return -1; }
    const dim3 g((unsigned)resw_w1_rows(a.B, a.H, a.W));
    if (a.in_pooled) hipLaunchKernelGGL((conv3x3_resw_w1x_kernel<true>), g, dim3(512), 0, s, a, ntiles, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv3x3_resw_w1x_kernel<false>), g, dim3(320), 0, s, a, ntiles, tiles_x, tiles_y);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// persistent grids: one workgroup per CU for the resident-weight kernel, as many as fit (1 or 2 per CU) for the streaming kernels
static void launch_resw(const ConvArgs& a, hipStream_t s) {
    const int resident = conv_ncu();
    ConvArgs b = a;
    b.sched = sched_or_fallback(a.sched);
    // 16 x 16 tiles at every width.  Round 6, measured and not kept: (i) 32 x 8 tiles where the width pads better to 8 -- the shipped idim 83:
    // 88 columns instead of 96 -- instantiate and pass every test, but run the W = 83 forward in 121 us against 112: two patch rows per MFMA
    // row block and a 34 x 10 patch per 256 pixels cost more than the 8 % of padded pixels return; (ii) the 64 -> 128 forward (conv3) as two
    // 64-channel passes of this kernel (output channel stride / offset, the 128-channel sign words written a half-word per pass): 72 us
    // against 61 for the weight-ring kernel -- the 40-column map pads to 48 under 16-wide tiles and each pass reloads its 72 KB bank.
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16, ntiles = tiles_x * tiles_y * a.B;
    hipLaunchKernelGGL((conv3x3_resw_kernel<16, 16>), dim3((unsigned)(ntiles < resident ? ntiles : resident)), dim3(320), 0, s, b, ntiles, tiles_x, tiles_y);
}
template <int CI, int CO, int TWV, int MASK, int THV = 16, bool UNPOOL = false>
static void launch_stream_t(const ConvArgs& a, int tiles_x, int tiles_y, hipStream_t s) {
    constexpr int NT = UNPOOL ? 512 : 384;
    static const int resident = [] {                             // (task-slot threads launch concurrently: thread-safe one-time initialisation)
        int per_cu = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_stream_kernel<CI, CO, THV, TWV, MASK, UNPOOL>, NT, 0);
        if (per_cu <= 0) per_cu = 1;
        return conv_ncu() * (per_cu < 2 ? per_cu : 2);
    }();
    const int ntiles = tiles_x * tiles_y * a.B;
    ConvArgs b = a;
    b.sched = sched_or_fallback(a.sched);
    hipLaunchKernelGGL((conv3x3_stream_kernel<CI, CO, THV, TWV, MASK, UNPOOL>), dim3((unsigned)(ntiles < resident ? ntiles : resident)), dim3(NT), 0, s, b, ntiles,
                       tiles_x, tiles_y);
}
template <int CI, int CO, int TWV, int THV = 16>
static void launch_stream(const ConvArgs& a, int tiles_x, int tiles_y, hipStream_t s) {
    if (a.mask) launch_stream_t<CI, CO, TWV, 1, THV>(a, tiles_x, tiles_y, s);
    else launch_stream_t<CI, CO, TWV, 0, THV>(a, tiles_x, tiles_y, s);
}
// ConvArgs::out_sign_bits for launches whose epilogue does not write them: from the stored 128-channel map, thread = (pixel, dword q)
__global__ void sign_bits128_kernel(const bf16* __restrict__ map, unsigned* __restrict__ bits, long npix) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 4) return;
    const long pix = i >> 2; const int q = (int)(i & 3);
    unsigned w = 0;
    for (int h = 0; h < 4; ++h) {
        const bf16x8 v = ld8(map + pix * 128 + h * 32 + q * 8);
        for (int j = 0; j < 8; ++j) w |= ((float)v[j] > 0.f ? 1u : 0u) << (8 * h + j);
    }
    bits[i] = w;
}
static int conv3x3_dispatch(const ConvArgs& a, hipStream_t s);
int mk_conv3x3(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    uint8_t* idx_after = nullptr;
    unsigned long long* sign_after = nullptr;
    const bool small = a.CIN <= 128 && a.COUT <= 128;        // the streaming / resident-weight kernels (8-wide tiles at every map width; 64 -> 64 forward: 16-wide)
    if ((a.in != nullptr) == (a.in_pooled != nullptr)) { mk_set_error("mk_conv3x3", "exactly one of in / in_pooled"); return -1; }
    if (a.in_pooled && (!a.in_idx || !(a.x1 || (a.CIN == 128 && a.COUT == 128 && a.mask_bits)))) {
        mk_set_error("mk_conv3x3", "in_pooled: needs in_idx; only the two dgrads behind a max-pool read it (64 <- 64 with the fused conv1 weight gradient, 128 <- 128 with mask_bits)");
        return -1;
    }
    {   // pool_idx / out_optional / out_sign_bits are honoured by the epilogues of the forward flavours of those kernels only: elsewhere
        // the launch stores the map and the codes / sign words are computed from it afterwards
        const bool in_epilogue = small && !a.x1 && !a.mask;
        if (!in_epilogue) { idx_after = a.pool_idx; a.pool_idx = nullptr; a.out_optional = 0; }
        if (a.out_sign_bits && (a.COUT != 128 || !a.out)) { mk_set_error("mk_conv3x3", "out_sign_bits: 128 output channels, stored map"); return -1; }
        if (a.out_sign_bits && !in_epilogue) { sign_after = a.out_sign_bits; a.out_sign_bits = nullptr; }
        if ((a.pool_idx && !a.pool_out) || (!a.out && !(a.out_optional && a.pool_out) && !a.x1)) { mk_set_error("mk_conv3x3", "pool_idx needs pool_out, and out may only be dropped when the launch pools"); return -1; }
    }
    const int rc = conv3x3_dispatch(a, s);
    if (rc == 0 && sign_after) {
        const long npix = (long)a.B * a.H * a.W;
        hipLaunchKernelGGL(sign_bits128_kernel, dim3((unsigned)((npix * 4 + 255) / 256)), dim3(256), 0, s, a.out, reinterpret_cast<unsigned*>(sign_after), npix);
    }
    if (rc == 0 && idx_after) return mk_maxpool_idx(a.out, idx_after, a.B, a.H, a.W, a.COUT, s);
    return rc;
}
static int conv3x3_dispatch(const ConvArgs& a, hipStream_t s) {
    if (a.x1) {
        if (!(a.CIN == 64 && a.COUT == 64) || !a.w1_slab) { mk_set_error("mk_conv3x3", "fused conv1 wgrad needs the 64->64 dgrad"); return -1; }
        return launch_resw_w1(a, s);                         // (-1 without launching when mask_bits is missing: w1_slab stays unwritten)
    }
    if (a.mask && (a.bias || a.relu || a.pool_out)) { mk_set_error("mk_conv3x3", "a launch is a forward flavour (bias / ReLU / pool) or a dgrad flavour (mask), not both"); return -1; }
    if (a.CIN <= 128 && a.COUT <= 128) {
        // a launch that stores only the pooled map (floor mode) has no use for the last row / column of an odd map: the shipped idim 83
        // pools to 41 columns, and the 128 -> 128 forward on them needs 40 = five 8-wide tiles, not six
        const bool pooled_only = !a.out && a.pool_out && !a.mask && a.W > 1 && a.H > 1;
        const int We = pooled_only ? (a.W & ~1) : a.W, He = pooled_only ? (a.H & ~1) : a.H;
        const int tiles_x = (We + 7) / 8, tiles_y = (He + 15) / 16, ty32 = (He + 31) / 32;
        // 64 -> 64 forward takes the resident-weight kernel with 16-wide tiles, also on widths that pad badly to 16 (W = 83: 0.100 ms
        // against 0.172 ms for the weight-ring kernel on 8-wide tiles).  32-row tiles where the registers allow: the per-tap weight slices
        // (the bulk of these kernels' vector-memory instructions) are amortised over twice the pixels -- not for 64 -> 128 (spills), nor
        // for a masked 128 <- 128 dgrad that reads the bf16 map as its mask (64 mask registers; with sign bits: 4)
        if (a.CIN == 64 && a.COUT == 64 && !a.mask) launch_resw(a, s);
        else if (a.CIN == 64 && a.COUT == 64) launch_stream<64, 64, 8>(a, tiles_x, tiles_y, s);
        else if (a.CIN == 64 && a.COUT == 128) launch_stream<64, 128, 8>(a, tiles_x, tiles_y, s);
        else if (a.CIN == 128 && a.COUT == 64) launch_stream<128, 64, 8, 32>(a, tiles_x, ty32, s);
        else if (a.CIN == 128 && a.COUT == 128) {
            if (!a.mask) launch_stream_t<128, 128, 8, 0, 32>(a, tiles_x, ty32, s);
            else if (a.mask_bits && a.in_pooled) launch_stream_t<128, 128, 8, 2, 32, true>(a, tiles_x, ty32, s);
            else if (a.mask_bits) launch_stream_t<128, 128, 8, 2, 32>(a, tiles_x, ty32, s);
            else launch_stream<128, 128, 8>(a, tiles_x, tiles_y, s);
        } else { mk_set_error("mk_conv3x3", "unsupported channel counts"); return -1; }
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    // BLSTM front-end (256 channels), round 6: the streaming kernel in passes of 128 output channels over the same patches (its weight ring
    // holds four [COUT][64] slices: 256 output channels at once would need 128 KB of it).  The patch kernel these launches ran on until
    // round 5 took 163 us for 256 -> 256 on the 8 x 200 x 42 map of tools/bench_blstm.py.
    if (a.pool_out || a.pool_idx || a.out_sign_bits || a.mask_bits || a.in_pooled) { mk_set_error("mk_conv3x3", "256-channel launches: plain forward / masked dgrad only"); return -1; }
    if (!((a.CIN == 128 && a.COUT == 256) || (a.CIN == 256 && a.COUT == 256) || (a.CIN == 256 && a.COUT == 128))) { mk_set_error("mk_conv3x3", "unsupported channel counts"); return -1; }
    const int tiles_x = (a.W + 7) / 8, tiles_y = (a.H + 15) / 16, ty32 = (a.H + 31) / 32;
    for (int c0 = 0; c0 < a.COUT; c0 += 128) {
        ConvArgs b = a;
        b.COUT = 128; b.out_cstride = a.COUT; b.out_coff = c0;
        b.wk = a.wk + (long)c0 * 9 * a.CIN;
        if (a.bias) b.bias = a.bias + c0;
        if (a.CIN == 128) { if (b.mask) launch_stream_t<128, 128, 8, 1, 16>(b, tiles_x, tiles_y, s); else launch_stream_t<128, 128, 8, 0, 32>(b, tiles_x, ty32, s); }
        else { if (b.mask) launch_stream_t<256, 128, 8, 1, 16>(b, tiles_x, tiles_y, s); else launch_stream_t<256, 128, 8, 0, 32>(b, tiles_x, ty32, s); }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ONE persistent workgroup per CU with the next tiles prefetched in registers (256 partial slabs: 37 MB written and read again per layer).
// One configuration for every mode: the partition of the pixels into partial sums is part of the result's bits (slots = sequential run).
static int wgrad2_nwg(int CIN, int COUT) { return W2_TOTAL_WG / ((CIN / 64) * (COUT / 64)); }
// tile = 16 x 8 pixels (wide x tall), or 8 x 16 where the map's width pads better to eights (40 -> 40 instead of 48, 83 -> 88 instead of 96:
// the 40-wide layers spent 20 % of their MFMAs on columns outside the map)
static int wgrad2_we(int W, bool pooled) { return pooled && W > 1 ? (W & ~1) : W; }
static int wgrad2_tw(int W) { return ((W + 7) / 8) * 8 < ((W + 15) / 16) * 16 ? 8 : 16; }
long mk_conv3x3_wgrad_slab_floats(int B, int H, int W, int CIN, int COUT) {
    (void)B; (void)H; (void)W;
    return (long)wgrad2_nwg(CIN, COUT) * (COUT * 9 * CIN + COUT);
}
// tiles and workgroups of a launch (one place: the fold launch that sums the partial slabs must count what the kernel wrote).  A pooled gradient
// is zero on the row / column the floor-mode pool cropped: tiles cover the even part of the map only.
struct Wgrad2Grid { int tw, tiles_x, tiles_y, ntiles, nwg; };
static Wgrad2Grid wgrad2_grid(const ConvWgradArgs& a) {
    const bool pooled = a.dy_pooled && a.pool_idx;
    const int We = wgrad2_we(a.W, pooled), He = wgrad2_we(a.H, pooled);
    Wgrad2Grid g;
    g.tw = wgrad2_tw(We);
    const int th = 128 / g.tw;
    g.tiles_x = (We + g.tw - 1) / g.tw; g.tiles_y = (He + th - 1) / th; g.ntiles = g.tiles_x * g.tiles_y * a.B;
    g.nwg = wgrad2_nwg(a.CIN, a.COUT);
    if (g.nwg > g.ntiles) g.nwg = g.ntiles;
    return g;
}
int mk_conv3x3_wgrad_nsplit(const ConvWgradArgs& a) { return wgrad2_grid(a).nwg; }
int mk_conv1_wgrad_fused_rows(int B, int H, int W) { return resw_w1_rows(B, H, W); }
int mk_conv3x3_wgrad(const ConvWgradArgs& a, hipStream_t s, int phase) {
    // phase 0: both launches; 1: the partial-slab kernel only; 2: the slab reduce only (the engine times them in separate slots)
    const Wgrad2Grid wg = wgrad2_grid(a);
    const int tw = wg.tw, tiles_x = wg.tiles_x, tiles_y = wg.tiles_y, ntiles = wg.ntiles, nwg = wg.nwg;
    const dim3 grid(nwg, (a.CIN / 64) * (a.COUT / 64));
#define W2(CI, CO, PL)                                                                                                                  \
    do {                                                                                                                                \
        if (tw == 8) hipLaunchKernelGGL((conv3x3_wgrad2_kernel<CI, CO, 8, PL>), grid, dim3(512), 0, s, a, nwg, ntiles, tiles_x, tiles_y);  \
        else hipLaunchKernelGGL((conv3x3_wgrad2_kernel<CI, CO, 16, PL>), grid, dim3(512), 0, s, a, nwg, ntiles, tiles_x, tiles_y);         \
    } while (0)
    const bool pooled = a.dy_pooled && a.pool_idx;
    if (!pooled && !a.dy) { mk_set_error("mk_conv3x3_wgrad", "dy missing"); return -1; }
    if (pooled && !((a.CIN == 64 && a.COUT == 64) || (a.CIN == 128 && a.COUT == 128))) { mk_set_error("mk_conv3x3_wgrad", "pooled dy: 64->64 and 128->128 only"); return -1; }
    if (phase == 2) {}
    else if (pooled && a.CIN == 64) W2(64, 64, true);
    else if (pooled) W2(128, 128, true);
    else if (a.CIN == 64 && a.COUT == 64) W2(64, 64, false);
    else if (a.CIN == 64 && a.COUT == 128) W2(64, 128, false);
    else if (a.CIN == 128 && a.COUT == 128) W2(128, 128, false);
    else if (a.CIN == 128 && a.COUT == 256) W2(128, 256, false);
    else if (a.CIN == 256 && a.COUT == 256) W2(256, 256, false);
    else { mk_set_error("mk_conv3x3_wgrad", "unsupported channel counts"); return -1; }
#undef W2
    const int n = a.COUT * 9 * a.CIN + a.COUT;
    if (phase != 1) hipLaunchKernelGGL(conv3x3_wgrad_reduce, dim3((n + 63) / 64), dim3(256), 0, s, a.slab, nwg, a.dw, a.db, a.CIN, a.COUT);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int mk_maxpool_fwd(const bf16* in, bf16* out, int B, int H, int W, int C, hipStream_t s, int ceil_mode) {
    const long n = ceil_mode ? (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8) : (long)B * (H / 2) * (W / 2) * (C / 8);
    if (n == 0) return 0;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, B, H, W, C, ceil_mode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int mk_maxpool_relu_bwd(const bf16* in, const bf16* dout, bf16* din, int B, int H, int W, int C, hipStream_t s, int ceil_mode) {
    const long n = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool_relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, dout, din, B, H, W, C, ceil_mode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int mk_maxpool_idx(const bf16* in, uint8_t* idx, int B, int H, int W, int C, hipStream_t s) {
    const long n = (long)B * (H / 2) * (W / 2) * (C / 8);
    if (n == 0) return 0;
    hipLaunchKernelGGL(maxpool_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, idx, B, H, W, C);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

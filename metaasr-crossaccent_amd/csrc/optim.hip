// Flat ("multi-tensor") optimiser kernels over the single fp32 parameter / gradient buffers.
// All parameters live in ONE contiguous buffer, so clip / SGD / Adam / accumulate / all-reduce
// are each one streaming pass at HBM rate instead of ~113 small launches
// (reference call sites: src/fo_meta_interface.py:148-149,180-221,223-250; optimizer.py:19-28).
#include "common.h"
#include "kernels.h"
#include "folds.h"
#include <type_traits>

namespace {

constexpr int SS_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ slab) {
    double acc = 0.0;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        } else {
            for (long j = i; j < n; ++j) acc += (double)x[j] * x[j];
        }
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) reinterpret_cast<double*>(slab)[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sumsq_final(const float* __restrict__ slab, int nblocks, float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += reinterpret_cast<const double*>(slab)[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(red[0]);
}

// torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to <= 1; NaN stays NaN
__device__ __forceinline__ float clip_coef(float norm, float max_norm) {
    const float c = max_norm / (norm + 1e-6f);
    return (c != c) ? c : fminf(c, 1.0f);
}

// inner step of run_task (fo_meta_interface.py:242-248): clip, then (unless the norm is NaN)
// torch.optim.SGD(momentum, nesterov, dampening 0): buf = first ? g : m*buf + g; p -= lr*(nesterov ? g + m*buf : buf)
struct SgdUpd {
    float coef, lr, momentum; int nesterov, first_step;
    // (explicit fused multiply-adds: left to -ffp-contract the scalar and the 16-byte forms of a pass were contracted differently, one
    // ulp apart on ~1 element in 10 000)
    __device__ __forceinline__ float operator()(float pi, float gr, float mi, float& mo) const {
        const float gi = __fmul_rn(gr, coef);
        float step = gi;
        if (momentum != 0.f) {
            const float b = first_step ? gi : __fmaf_rn(momentum, mi, gi);
            mo = b;
            step = nesterov ? __fmaf_rn(momentum, b, gi) : b;
        }
        return __fmaf_rn(-lr, step, pi);
    }
};
__global__ void clip_sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, long n,
                                const float* __restrict__ norm, float max_norm, float lr, float momentum, int nesterov,
                                int step_flags, int vec) {
    // step_flags: bit 0 = first step of this optimiser (no buffer yet: buf = g); bit 1 = its LAST step (the optimiser is dropped
    // afterwards, as run_task drops its SGD after k steps: the buffer update would be a dead 4 B/param store and is skipped)
    const int first_step = step_flags & 1;
    float coef = 1.f;
    if (norm) {
        const float nv = norm[0];
        if (nv != nv) return;                               // math.isnan(grad_norm) -> skip the step
        coef = clip_coef(nv, max_norm);
    }
    const SgdUpd upd{coef, lr, momentum, nesterov, first_step};
    // 16 bytes per lane and stream (the flat buffers are 16-byte aligned allocations): a 4-byte-per-lane pass needs four times
    // the vector-memory instructions for the same bytes
    const long n4 = vec ? n >> 2 : 0, stride = (long)gridDim.x * blockDim.x;
    const bool use_m = momentum != 0.f, rd_m = use_m && !first_step, wr_m = use_m && !(step_flags & 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 pv = reinterpret_cast<const float4*>(p)[i], gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = rd_m ? reinterpret_cast<const float4*>(mom)[i] : float4{0.f, 0.f, 0.f, 0.f};
        float4 o;
        o.x = upd(pv.x, gv.x, mv.x, mv.x); o.y = upd(pv.y, gv.y, mv.y, mv.y); o.z = upd(pv.z, gv.z, mv.z, mv.z); o.w = upd(pv.w, gv.w, mv.w, mv.w);
        reinterpret_cast<float4*>(p)[i] = o;
        if (wr_m) reinterpret_cast<float4*>(mom)[i] = mv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {      // tail (n % 4 elements)
        float mo = rd_m ? mom[i] : 0.f;
        p[i] = upd(p[i], g[i], mo, mo);
        if (wr_m) mom[i] = mo;
    }
}
// element-wise passes over flat fp32 buffers, 16 bytes per lane and stream when every pointer is 16-byte aligned (vec), with a
// scalar tail / fallback: body(i) handles element i, body4(i4) elements 4*i4 .. 4*i4+3
template <class F1, class F4>
__device__ __forceinline__ void flat_pass(long n, int vec, F1 body, F4 body4) {
    const long n4 = vec ? n >> 2 : 0, stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = t0; i < n4; i += stride) body4(i);
    for (long i = (n4 << 2) + t0; i < n; i += stride) body(i);
}
#define F4P(ptr) reinterpret_cast<float4*>(ptr)
#define F4C(ptr) reinterpret_cast<const float4*>(ptr)
// nan_zero (masr_set_drop_nan_grads, the opt-out of quirk Q5): a gradient whose norm is NaN becomes / contributes zeros instead of NaNs
__global__ void clip_scale_kernel(float* __restrict__ g, long n, const float* __restrict__ norm, float max_norm, int vec, int nan_zero) {
    const float coef = clip_coef(norm[0], max_norm);
    if (nan_zero && coef != coef) {
        flat_pass(n, vec, [&](long i) { g[i] = 0.f; }, [&](long i) { F4P(g)[i] = float4{0.f, 0.f, 0.f, 0.f}; });
        return;
    }
    flat_pass(n, vec, [&](long i) { g[i] *= coef; },
              [&](long i) { float4 v = F4C(g)[i]; v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef; F4P(g)[i] = v; });
}
__global__ void clip_axpy_kernel(float* __restrict__ acc, const float* __restrict__ g, long n, const float* __restrict__ norm,
                                 float max_norm, int vec, int nan_zero) {
    const float coef = clip_coef(norm[0], max_norm);
    if (nan_zero && coef != coef) return;
    flat_pass(n, vec, [&](long i) { acc[i] += coef * g[i]; },
              [&](long i) { float4 a = F4C(acc)[i]; const float4 v = F4C(g)[i];
                            a.x += coef * v.x; a.y += coef * v.y; a.z += coef * v.z; a.w += coef * v.w; F4P(acc)[i] = a; });
}
__global__ void scale_kernel(float* __restrict__ x, long n, float a, int vec) {
    flat_pass(n, vec, [&](long i) { x[i] *= a; },
              [&](long i) { float4 v = F4C(x)[i]; v.x *= a; v.y *= a; v.z *= a; v.w *= a; F4P(x)[i] = v; });
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float a, int vec) {
    flat_pass(n, vec, [&](long i) { y[i] += a * x[i]; },
              [&](long i) { float4 o = F4C(y)[i]; const float4 v = F4C(x)[i];
                            o.x += a * v.x; o.y += a * v.y; o.z += a * v.z; o.w += a * v.w; F4P(y)[i] = o; });
}
// torch.optim.Adam (no amsgrad, no weight decay): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= lr/(1-b1^t) * m/denom
// decay_mul = 1 - lr * weight_decay (AdamW: decoupled, applied to the weight first, as torch.optim.AdamW) or 1;
// l2 = weight_decay of torch.optim.Adam (added to the gradient) or 0
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float step_size, float b1, float b2, float eps, float inv_sqrt_bc2, float decay_mul, float l2, int vec) {
    auto one = [&](float& pp, float gg, float& mm, float& vv) {
        const float pi = pp * decay_mul;
        const float gi = gg + l2 * pi;
        mm = mm + (gi - mm) * (1.f - b1);                        // lerp, as torch
        vv = b2 * vv + (1.f - b2) * gi * gi;
        pp = pi - step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    };
    flat_pass(n, vec, [&](long i) { one(p[i], g[i], m[i], v[i]); },
              [&](long i) { float4 pv = F4C(p)[i], mv = F4C(m)[i], vv = F4C(v)[i]; const float4 gv = F4C(g)[i];
                            one(pv.x, gv.x, mv.x, vv.x); one(pv.y, gv.y, mv.y, vv.y); one(pv.z, gv.z, mv.z, vv.z); one(pv.w, gv.w, mv.w, vv.w);
                            F4P(p)[i] = pv; F4P(m)[i] = mv; F4P(v)[i] = vv; });
}
// Adam step that the DEVICE skips when the gradient norm is NaN (`if math.isnan(grad_norm): warn else: step()` of the training loops,
// mono_interface.py:141-148, multi_interface.py:108-114) so that the host need not wait for the norm.  The host is at most ONE step
// ahead: when it queues this step it does not know yet whether the previous one was skipped, i.e. whether this is the optimiser's
// step number n+1 or n.  It passes the scalars for both (A: the previous step was applied, B: it was skipped); the previous launch left
// its verdict in flags[slot ^ 1], this one leaves its own in flags[slot].
struct AdamCand { float step_size, inv_sqrt_bc2, decay_mul, l2; };
__global__ void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                    float b1, float b2, float eps, AdamCand A, AdamCand B, const float* __restrict__ norm,
                                    int* __restrict__ flags, int slot, int vec) {
    const float nv = norm[0];
    const bool skip = nv != nv;
    const bool prev_skipped = flags[slot ^ 1] != 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) flags[slot] = skip ? 1 : 0;
    if (skip) return;
    const AdamCand c = prev_skipped ? B : A;
    auto one = [&](float& pp, float gg, float& mm, float& vv) {
        const float pi = pp * c.decay_mul;
        const float gi = gg + c.l2 * pi;
        mm = mm + (gi - mm) * (1.f - b1);
        vv = b2 * vv + (1.f - b2) * gi * gi;
        pp = pi - c.step_size * mm / (sqrtf(vv) * c.inv_sqrt_bc2 + eps);
    };
    flat_pass(n, vec, [&](long i) { one(p[i], g[i], m[i], v[i]); },
              [&](long i) { float4 pv = F4C(p)[i], mv = F4C(m)[i], vv = F4C(v)[i]; const float4 gv = F4C(g)[i];
                            one(pv.x, gv.x, mv.x, vv.x); one(pv.y, gv.y, mv.y, vv.y); one(pv.z, gv.z, mv.z, vv.z); one(pv.w, gv.w, mv.w, vv.w);
                            F4P(p)[i] = pv; F4P(m)[i] = mv; F4P(v)[i] = vv; });
}
// the same step on g = (((g0 + g1) + g2) + ...) * scale, the sum the meta loop used to build with one axpy pass per task and a
// scale pass (same additions in the same order: bit-identical), read straight from the task slots' gradient buffers
struct GradList { const float* g[8]; int n; float scale; };
__global__ void adam_sum_kernel(float* __restrict__ p, GradList gl, float* __restrict__ m, float* __restrict__ v,
                                long n, float step_size, float b1, float b2, float eps, float inv_sqrt_bc2, int vec) {
    auto one = [&](float& pp, float gg, float& mm, float& vv) {
        mm = mm + (gg - mm) * (1.f - b1);
        vv = b2 * vv + (1.f - b2) * gg * gg;
        pp = pp - step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
    };
    flat_pass(n, vec, [&](long i) { float gg = 0.f + gl.g[0][i]; for (int k = 1; k < gl.n; ++k) gg += gl.g[k][i]; one(p[i], gg * gl.scale, m[i], v[i]); },
              [&](long i) { float4 pv = F4C(p)[i], mv = F4C(m)[i], vv = F4C(v)[i]; float4 gv = F4C(gl.g[0])[i];
                            gv.x = 0.f + gv.x; gv.y = 0.f + gv.y; gv.z = 0.f + gv.z; gv.w = 0.f + gv.w;
                            for (int k = 1; k < gl.n; ++k) { const float4 t = F4C(gl.g[k])[i]; gv.x += t.x; gv.y += t.y; gv.z += t.z; gv.w += t.w; }
                            one(pv.x, gv.x * gl.scale, mv.x, vv.x); one(pv.y, gv.y * gl.scale, mv.y, vv.y);
                            one(pv.z, gv.z * gl.scale, mv.z, vv.z); one(pv.w, gv.w * gl.scale, mv.w, vv.w);
                            F4P(p)[i] = pv; F4P(m)[i] = mv; F4P(v)[i] = vv; });
}
// out = ((g0 + g1) + ...) + g_{n-1}: the rank-local sum of a wave's task gradients, the payload of the wave's ONE all-reduce
// (same additions in the same order as zero + n axpy passes)
__global__ void sum_n_kernel(float* __restrict__ out, GradList gl, long n, int vec) {
    flat_pass(n, vec, [&](long i) { float gg = 0.f + gl.g[0][i]; for (int k = 1; k < gl.n; ++k) gg += gl.g[k][i]; out[i] = gg * gl.scale; },
              [&](long i) { float4 gv = F4C(gl.g[0])[i];
                            gv.x = 0.f + gv.x; gv.y = 0.f + gv.y; gv.z = 0.f + gv.z; gv.w = 0.f + gv.w;
                            for (int k = 1; k < gl.n; ++k) { const float4 t = F4C(gl.g[k])[i]; gv.x += t.x; gv.y += t.y; gv.z += t.z; gv.w += t.w; }
                            gv.x *= gl.scale; gv.y *= gl.scale; gv.z *= gl.scale; gv.w *= gl.scale;
                            F4P(out)[i] = gv; });
}
__global__ void cast_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = (bf16)x[i];
}
// y[c][r] = bf16(x[r][c]) via a 32x32 LDS tile
__global__ void transpose_cast_kernel(const float* __restrict__ x, bf16* __restrict__ y, int R, int C, long ldy) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 256 threads: 8 rows per pass
    for (int k = ty; k < 32; k += 8)
        t[k][tx] = (r0 + k < R && c0 + tx < C) ? x[(long)(r0 + k) * C + c0 + tx] : 0.f;
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) y[(long)(c0 + k) * ldy + r0 + tx] = (bf16)t[tx][k];
}
__global__ void conv_shadow_kernel(const float* __restrict__ w, bf16* __restrict__ wk, bf16* __restrict__ wd, int CO, int CI) {
    const int n = CO * CI * 9;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int tap = i % 9, ci = (i / 9) % CI, co = i / (9 * CI);
    const bf16 v = (bf16)w[i];
    wk[(long)co * 9 * CI + tap * CI + ci] = v;                        // forward: out[co] += in[p+off(tap)][ci] * w
    if (wd) wd[(long)ci * 9 * CO + (8 - tap) * CO + co] = v;          // dgrad: din[ci] += dy[p-off(tap)][co] * w
}
__global__ __launch_bounds__(256) void vgg2enc_unpermute_kernel(const float* __restrict__ g, float* __restrict__ dw, int E, int C, int Dp) {
    vgg2enc_unpermute_body(g, dw, E, C, Dp, blockIdx.x);
}

// every operand shadow of the model, one job list, one launch (see kernels.h)
constexpr int SH_TILE = 64;

// One full-width 64 x 64 tile of a Linear weight whose rows are a multiple of four floats: fp32 rows in, bf16 out in BOTH layouts, 16 bytes per
// lane and access on the global side although the tensor sits at an arbitrary dword offset of the flat buffer (P itself is 16-byte aligned).
// Every row of the tile starts `mis` floats past a 16-byte boundary (the same for all rows) and spans 17 ALIGNED float4s; SEVENTEEN lanes per row
// take one vector each (15 rows per pass, five passes: thread group rr takes rows 5 rr .. 5 rr + 4), round it to bf16 and drop it into two LDS images of the tile -- A[row][col] and
// At[col][row] -- at its tile columns.  Both images carry a margin (A: 8 columns in front, 8 behind, rows 64..74; At: 4 rows in front and behind,
// columns 64..77) that takes what the first / 17th vector hold of the neighbouring tile and the rows the fifth pass reads past the tile, so that
// no lane tests what it owns; the margins are never read.  The read side is 16 bytes from LDS and one 16-byte store per lane, pass and layout.
// (Round 6: the fp32 tile with scalar LDS reads and a rotation against its bank conflicts executed 780 instructions per wave and tile and ran the
// launch at 3.0 TB/s; this form ~170.)
// row strides (elements): A 160 B (16-byte reads); At 156 B = 39 dwords: the seventeen lanes of a row write column elements 4 x 39 dwords apart
// = 28 banks apart mod 64, sixteen different banks (at 152 B they met in eight: SQ_LDS_BANK_CONFLICT 0.31 of the wave cycles), read back as dwords
constexpr int SH_SA = 80, SH_ST = 78;
constexpr int SH_LDS_BYTES = (75 * SH_SA + 72 * SH_ST) * 2;       // 23 232 B: both images (the generic tile's fp32 [64][65] = 16 640 B shares them)
__device__ __forceinline__ void shadow_tile_aligned(const float* __restrict__ P, const ShadowDesc& d, bf16* __restrict__ p0, bf16* __restrict__ p1,
                                                    int r0, int c0, bf16* lds) {
    constexpr int SA = SH_SA, ST = SH_ST;
    bf16* const A = lds;                                          // [row k][8 + col]
    bf16* const At = lds + 75 * SA;                               // [4 + col][row k]
    const int mis = (int)((d.src + (long)r0 * d.K + c0) & 3);
    const int v = threadIdx.x % 17, rr = threadIdx.x / 17;
    if (rr < 15) {                                                // (thread 255 sits out)
        // mis == 0: the 17th vector lies behind the tile (possibly behind the tensor): the 16th is read again and lands in the margin
        const int vec = (v == 16 && mis == 0) ? 15 : v;
        const float* base = P + d.src + (long)r0 * d.K + c0 - mis + 4 * vec;
        // thread (rr, v) takes rows 5 rr + pass: the four rows a wave writes at a time are five apart (adjacent rows would be the two halves of
        // one dword of an At line)
        bf16* a = A + 5 * rr * SA + 8 + 4 * v - mis;
        bf16* at = At + (4 + 4 * v - mis) * ST + 5 * rr;
        // all five loads first: a workgroup that waited for each pass's vector before asking for the next spent five memory round trips per tile
        // (the launch ran at the latency, 54 us for 200 MB, whatever its instruction count)
        float4 qs[5];
#pragma unroll
        for (int pass = 0; pass < 5; ++pass) {
            const int k = 5 * rr + pass;
            const int kr = r0 + k < d.N ? k : 0;                   // rows behind the tensor: the tile's first row again (never stored)
            qs[pass] = *reinterpret_cast<const float4*>(base + (long)kr * d.K);
        }
#pragma unroll
        for (int pass = 0; pass < 5; ++pass) {
            const float4 q = qs[pass];
            const bf16 b0 = (bf16)q.x, b1 = (bf16)q.y, b2 = (bf16)q.z, b3 = (bf16)q.w;
            bf16* ak = a + pass * SA;
            if (mis == 0) {
                *reinterpret_cast<bf16x4*>(ak) = bf16x4{b0, b1, b2, b3};
            } else {
                ak[0] = b0; ak[1] = b1; ak[2] = b2; ak[3] = b3;
            }
            bf16* atk = at + pass;
            atk[0] = b0; atk[ST] = b1; atk[2 * ST] = b2; atk[3 * ST] = b3;
        }
    }
    __syncthreads();
    const int sub = threadIdx.x & 7, line = threadIdx.x >> 3;        // 8 lanes x 8 elements = one 64-element line
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int l = line + 32 * pass;
        if (r0 + l < d.N) st8(p0 + (long)(r0 + l) * d.K + c0 + sub * 8, *reinterpret_cast<const bf16x8*>(A + l * SA + 8 + sub * 8));   // k16 [N][K]
        const int r = r0 + sub * 8;                                  // t16 [K][ldt]: line = tile column, elements run over the tile's rows
        if (r < d.N) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            const unsigned* w = reinterpret_cast<const unsigned*>(At + (4 + l) * ST + sub * 8);          // (4-byte aligned lines)
            const bf16x8 o = __builtin_bit_cast(bf16x8, u32x4{w[0], w[1], w[2], w[3]});
            bf16* dst = p1 + (long)(c0 + l) * d.ldt + r;
            if (r + 8 <= d.N) st8(dst, o);
            else for (int e = 0; r + e < d.N; ++e) dst[e] = o[e];   // (pads of a padded row stay untouched)
        }
    }
}
__global__ __launch_bounds__(256) void all_shadows_kernel(const float* __restrict__ P, const ShadowJobs jobs) {
    auto elem = [&](long i) -> float { return P[i]; };
    int lo = 0, hi = jobs.n - 1;                                  // last job whose tile_start <= blockIdx.x (uniform: scalar ALU)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= jobs.d[mid].tile_start) lo = mid; else hi = mid - 1;
    }
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[SH_LDS_BYTES];
    static_assert(SH_LDS_BYTES >= SH_TILE * (SH_TILE + 1) * 4, "the generic tile fits");
    const int e = lo;
    const ShadowDesc& d = jobs.d[e];
    const int blk = blockIdx.x - d.tile_start;
    const float* x = P + d.src;
    bf16* p0 = jobs.p[2 * e]; bf16* p1 = jobs.p[2 * e + 1];
    if (d.type == SH_LINEAR && (d.K & 3) == 0) {
        const int tc = (d.K + SH_TILE - 1) / SH_TILE;
        const int r0 = (blk / tc) * SH_TILE, c0 = (blk % tc) * SH_TILE;
        if (c0 + SH_TILE <= d.K) { shadow_tile_aligned(P, d, p0, p1, r0, c0, reinterpret_cast<bf16*>(lds_raw)); return; }
    }
    if (d.type == SH_LINEAR || d.type == SH_VGG2ENC) {
        // the remaining tiles (a ragged last tile column, rows that are no multiple of four floats, the vgg2enc gather):
        // 64 x 64 tile: fp32 rows in, through LDS, bf16 out as 16 bytes per lane in BOTH layouts -- the pass is bound by
        // vector-memory instructions: 4-byte loads and 2-byte stores made 48 of them per wave and tile where 9 suffice.
        // SH_VGG2ENC is the same tile pass with a column gather on the way in: output column fn = dd * C + c comes from source column
        // c * Dp + dd (one element per thread and 2-byte scattered stores made this 5 % of the bytes 16 % of the launch's time and 43 % of
        // its workgroups)
        const bool vgg = d.type == SH_VGG2ENC;
        const int dK = vgg ? d.a0 * d.a1 : d.K, dldt = vgg ? d.N : d.ldt;
        const int tc = (dK + SH_TILE - 1) / SH_TILE;
        const int r0 = (blk / tc) * SH_TILE, c0 = (blk % tc) * SH_TILE;
        float (*t)[SH_TILE + 1] = reinterpret_cast<float (*)[SH_TILE + 1]>(lds_raw);
        if (vgg) {
            const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 4 rows per pass, 64 consecutive OUTPUT columns per row
            const int fn = c0 + tx, C = d.a0, Dp = d.a1;
            const int f = (fn % C) * Dp + fn / C;                        // source column
            for (int k = ty; k < SH_TILE; k += 4) {
                const bool ok = r0 + k < d.N && fn < dK;
                t[k][tx] = ok ? elem(d.src + (long)(r0 + k) * dK + f) : 0.f;
            }
        } else {
            const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 4 rows per pass, 64 consecutive columns per row
            for (int k = ty; k < SH_TILE; k += 4) {
                const bool ok = r0 + k < d.N && c0 + tx < d.K;
                t[k][tx] = ok ? elem(d.src + (long)(r0 + k) * d.K + c0 + tx) : 0.f;
            }
        }
        __syncthreads();
        const int sub = threadIdx.x & 7, line = threadIdx.x >> 3;        // 8 lanes x 8 elements = one 64-element line
        // LDS bank of element (line l, column 8 sub + e) is (l + 8 sub + e) mod 32: lanes sub and sub + 4 of a line would meet in every read
        // (2-way conflict, SQ_LDS_BANK_CONFLICT 0.49 of the LDS cycles).  Lanes with sub >= 4 take their eight elements rotated by four.
        const int rot = (sub >> 2) * 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int l = line + 32 * pass;
            {   // k16 [N][K]: line = tile row
                const int r = r0 + l, c = c0 + sub * 8;
                if (r < d.N && c < dK) {
                    bf16x8 o;
                    float tv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) tv[e] = t[l][sub * 8 + ((e + rot) & 7)];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16)(rot ? tv[(e + 4) & 7] : tv[e]);
                    if (c + 8 <= dK) st8(p0 + (long)r * dK + c, o);
                    else for (int e = 0; c + e < dK; ++e) p0[(long)r * dK + c + e] = o[e];
                }
            }
            {   // t16 [K][ldt]: line = tile column, elements run over the tile's rows
                const int c = c0 + l, r = r0 + sub * 8;
                if (c < dK && r < d.N) {
                    bf16x8 o;
                    float tv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) tv[e] = t[sub * 8 + ((e + rot) & 7)][l];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16)(rot ? tv[(e + 4) & 7] : tv[e]);
                    if (r + 8 <= d.N) st8(p1 + (long)c * dldt + r, o);
                    else for (int e = 0; r + e < d.N; ++e) p1[(long)c * dldt + r + e] = o[e];       // (pads of a padded row stay untouched)
                }
            }
        }
    } else if (d.type == SH_CONV) {
        const int CO = d.N, CI = d.K, n = CO * CI * 9;
        const int i = blk * 256 + threadIdx.x;
        if (i >= n) return;
        const int tap = i % 9, ci = (i / 9) % CI, co = i / (9 * CI);
        const bf16 v = (bf16)elem(d.src + i);
        p0[(long)co * 9 * CI + tap * CI + ci] = v;                        // forward: out[co] += in[p+off(tap)][ci] * w
        p1[(long)ci * 9 * CO + (8 - tap) * CO + co] = v;                  // dgrad: din[ci] += dy[p-off(tap)][co] * w
    } else {
        const int i = blk * 256 + threadIdx.x;
        if (i < d.N) reinterpret_cast<float*>(p0)[i] = elem(d.src + i);
    }
    (void)x;
}

inline int aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return !(((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15);
}
inline unsigned flat_blocks(long n) {
    long b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : (mk_set_error(__func__, "launch failed"), -1))

long mk_sumsq_slab_floats(long) { return SS_BLOCKS * 2; }
int mk_sumsq(const float* x, long n, float* slab, float* out_norm, hipStream_t s) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(SS_BLOCKS), dim3(256), 0, s, x, n, slab);
    hipLaunchKernelGGL(sumsq_final, dim3(1), dim3(256), 0, s, slab, SS_BLOCKS, out_norm);
    return LAUNCH_OK();
}
int mk_clip_sgd(float* p, const float* g, float* mom, long n, const float* norm, float max_norm, float lr, float momentum,
                  int nesterov, int first_step, hipStream_t s) {
    const int vec = aligned16(p, g, mom);                                           // (mom may be null: momentum 0)
    hipLaunchKernelGGL(clip_sgd_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, p, g, mom, n, norm, max_norm, lr, momentum, nesterov, first_step, vec);
    return LAUNCH_OK();
}
int mk_clip_scale(float* g, long n, const float* norm, float max_norm, hipStream_t s, int nan_zero) {
    hipLaunchKernelGGL(clip_scale_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, g, n, norm, max_norm, aligned16(g), nan_zero);
    return LAUNCH_OK();
}
int mk_clip_axpy(float* acc, const float* g, long n, const float* norm, float max_norm, hipStream_t s, int nan_zero) {
    hipLaunchKernelGGL(clip_axpy_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, acc, g, n, norm, max_norm, aligned16(acc, g), nan_zero);
    return LAUNCH_OK();
}
int mk_scale(float* x, long n, float a, hipStream_t s) {
    hipLaunchKernelGGL(scale_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, x, n, a, aligned16(x));
    return LAUNCH_OK();
}
int mk_axpy(float* y, const float* x, long n, float a, hipStream_t s) {
    hipLaunchKernelGGL(axpy_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, y, x, n, a, aligned16(y, x));
    return LAUNCH_OK();
}
int mk_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int t, float weight_decay, int decoupled, hipStream_t s) {
    const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
    hipLaunchKernelGGL(adam_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, p, g, m, v, n, (float)(lr / bc1), b1, b2, eps,
                       (float)(1.0 / sqrt(bc2)), decoupled ? 1.f - lr * weight_decay : 1.f, decoupled ? 0.f : weight_decay, aligned16(p, g, m, v));
    return LAUNCH_OK();
}
int mk_adam_guarded(float* p, const float* g, float* m, float* v, long n, float lr_a, int t_a, float lr_b, int t_b, float b1, float b2, float eps,
                    float weight_decay, int decoupled, const float* norm, int* flags, int slot, hipStream_t s) {
    auto cand = [&](float lr, int t) {
        const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
        return AdamCand{(float)(lr / bc1), (float)(1.0 / sqrt(bc2)), decoupled ? 1.f - lr * weight_decay : 1.f, decoupled ? 0.f : weight_decay};
    };
    hipLaunchKernelGGL(adam_guarded_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, p, g, m, v, n, b1, b2, eps, cand(lr_a, t_a), cand(lr_b, t_b),
                       norm, flags, slot & 1, aligned16(p, g, m, v));
    return LAUNCH_OK();
}
int mk_adam_sum(float* p, const float* const* grads, int n_grads, float gscale, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                int t, hipStream_t s) {
    if (n_grads < 1 || n_grads > 8) { mk_set_error("mk_adam_sum", "1 to 8 gradient buffers"); return -1; }
    GradList gl{}; gl.n = n_grads; gl.scale = gscale;
    bool al = aligned16(p, m, v);
    for (int k = 0; k < n_grads; ++k) { gl.g[k] = grads[k]; al = al && !((uintptr_t)grads[k] & 15); }
    const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
    hipLaunchKernelGGL(adam_sum_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, p, gl, m, v, n, (float)(lr / bc1), b1, b2, eps,
                       (float)(1.0 / sqrt(bc2)), al ? 1 : 0);
    return LAUNCH_OK();
}
int mk_sum_n(float* out, const float* const* grads, int n_grads, float scale, long n, hipStream_t s) {
    if (n_grads < 1 || n_grads > 8) { mk_set_error("mk_sum_n", "1 to 8 buffers"); return -1; }
    GradList gl{}; gl.n = n_grads; gl.scale = scale;
    bool al = aligned16(out);
    for (int k = 0; k < n_grads; ++k) { gl.g[k] = grads[k]; al = al && !((uintptr_t)grads[k] & 15); }
    hipLaunchKernelGGL(sum_n_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, out, gl, n, al ? 1 : 0);
    return LAUNCH_OK();
}
// RAdam (Liu et al. 2020): Adam moments; while the variance estimate is unreliable the step is the bias-corrected momentum alone,
// afterwards Adam's step times the rectification r_t.  All of that is scalar work on the host: the update itself is adam_kernel
// with a folded step size and, in the unrectified phase, a denominator of 1 (inv_sqrt_bc2 = 0, eps = 1).  Two sets of conventions:
//   variant 1 = `torch_optimizer.RAdam`, the package the reference imports (src/transformer_torch_trainer.py:36-41; the authors'
//     published implementation): rectified once N_sma >= 5; sqrt(1 - b2^t) folded into the step size, i.e. denom = sqrt(v) + eps
//     (eps NOT scaled by the bias correction); weight decay applied to the WEIGHT (p -= lr * wd * p) before the update;
//   variant 0 = torch.optim.RAdam: rectified once rho_t > 5; denom = sqrt(v) / sqrt(1 - b2^t) + eps; weight decay = L2 term on the gradient.
int mk_radam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int t, float weight_decay, int variant, hipStream_t s) {
    const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
    const double rho_inf = 2.0 / (1.0 - b2) - 1.0, rho_t = rho_inf - 2.0 * t * pow((double)b2, t) / bc2;
    const bool rect = variant ? rho_t >= 5.0 : rho_t > 5.0;
    double step = lr / bc1; float inv_sqrt_bc2 = 0.f, e = 1.f;
    if (rect) {
        step *= sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
        if (variant) { step *= sqrt(bc2); inv_sqrt_bc2 = 1.f; } else inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        e = eps;
    }
    const float decay_mul = variant ? 1.f - lr * weight_decay : 1.f, l2 = variant ? 0.f : weight_decay;
    hipLaunchKernelGGL(adam_kernel, dim3(flat_blocks((n + 3) / 4)), dim3(256), 0, s, p, g, m, v, n, (float)step, b1, b2, e, inv_sqrt_bc2, decay_mul, l2, aligned16(p, g, m, v));
    return LAUNCH_OK();
}
int mk_cast_bf16(const float* x, bf16* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(flat_blocks(n)), dim3(256), 0, s, x, y, n);
    return LAUNCH_OK();
}
int mk_transpose_cast_bf16(const float* x, bf16* y, int R, int C, long ldy, hipStream_t s) {
    hipLaunchKernelGGL(transpose_cast_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, x, y, R, C, ldy);
    return LAUNCH_OK();
}
int mk_shadow_blocks(const ShadowDesc& d) {
    switch (d.type) {
        case SH_LINEAR: return ((d.N + SH_TILE - 1) / SH_TILE) * ((d.K + SH_TILE - 1) / SH_TILE);
        case SH_CONV: return (d.N * d.K * 9 + 255) / 256;
        case SH_VGG2ENC: return ((d.N + SH_TILE - 1) / SH_TILE) * ((d.a0 * d.a1 + SH_TILE - 1) / SH_TILE);
        default: return (d.N + 255) / 256;
    }
}
int mk_all_shadows(const float* P, const ShadowJobs& jobs, hipStream_t s) {
    if (jobs.n <= 0) return 0;
    hipLaunchKernelGGL(all_shadows_kernel, dim3(jobs.blocks), dim3(256), 0, s, P, jobs);
    return LAUNCH_OK();
}
int mk_conv_weight_shadows(const float* w, bf16* wk, bf16* wd, int CO, int CI, hipStream_t s) {
    const int n = CO * CI * 9;
    hipLaunchKernelGGL(conv_shadow_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wk, wd, CO, CI);
    return LAUNCH_OK();
}
int mk_vgg2enc_grad_unpermute(const float* g, float* dw, int E, int C, int Dp, hipStream_t s) {
    const long n = (long)E * C * Dp;
    hipLaunchKernelGGL(vgg2enc_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, dw, E, C, Dp);
    return LAUNCH_OK();
}

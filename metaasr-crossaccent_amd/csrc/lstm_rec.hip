// The BLSTM recurrence as ONE launch per layer and pass (forward through time / backward through time), both directions in it
// (src/modules/encoder.py:100-125 RNNP.forward: nn.LSTM(bidirectional) on the packed batch).
//
// lstm.hip runs the recurrence as one launch per timestep: ~5.7 us each for ~0.3 us of arithmetic, 600 launches in a training
// step of config/blstm (3 layers x 100 frames x 2 passes), and every launch streams W_hh (1.1 MB per direction) through its
// CUs again.  Here a direction is NW workgroups (12 for H = 360) that live for the whole sequence:
//   * a workgroup owns a slice of <= 32 hidden units and keeps its slice of W_hh (forward: rows of the unit-major gate axis;
//     backward: rows of W_hh^T) in REGISTERS as MFMA B-operand fragments for all T steps -- 96 KB per workgroup, 48 VGPRs
//     per lane: the weights are read once per layer and pass;
//   * per step a workgroup needs the WHOLE h_{t-1} (forward; dz_{t+1} backward) of its direction, i.e. what every workgroup
//     of the direction produced one step earlier.  That exchange is the only serial hop of a step and carries no flag and no
//     fence: the data are their own flags (cdna_hip_programming.md Guideline 16, form R2).  A value pair travels as ONE
//     aligned 8-byte granule {tag = step + 1, two bf16}, stored write-through with an agent-scope relaxed atomic store; a
//     consumer re-reads its granules with agent-scope relaxed loads until every tag is the step it waits for.  Two granule
//     buffers alternate by step parity: nobody can be two steps ahead of anybody (a step needs everyone's previous one), so
//     a buffer is never overwritten before its last reader is done.  Tags are zeroed before every launch (hipMemsetAsync).
//   * the gathered operand goes through LDS once (padded rows) and from there into the MFMA A fragments of the eight waves;
//     cell state (forward) and the dL/dc carry (backward) never leave their lane's registers.
// Every spin is bounded: a workgroup that waits ~seconds sets *err and leaves (its peers then run into their own bound);
// the engine reports it with the step's stats.  All workgroups of a launch must be resident together -- 24 of 512 threads
// on a 256-CU chip; under other streams' load a late one starts when a CU frees up, the early ones only spin meanwhile.
// Arithmetic: the same fp32 formulas as lstm.hip's step kernels; the recurrent product is accumulated in ONE MFMA chain per
// output tile where the step kernels add four partial chains, so results agree to fp32 rounding, not bit for bit.
#include <atomic>
#include "kernels.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int REC_THREADS = 512;
constexpr int REC_UNITS = 32;                                   // hidden units per workgroup at most (8 waves x one 4-unit gate tile / two 16-unit tiles)
constexpr unsigned SPIN_LIMIT = 1u << 22;                       // polls of ~1 us each

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const bf16 a = (bf16)lo, b = (bf16)hi;
    return (unsigned)*reinterpret_cast<const unsigned short*>(&a) | ((unsigned)*reinterpret_cast<const unsigned short*>(&b) << 16);
}
// value of lane G4 of this lane's quad (DPP quad_perm [G4, G4, G4, G4])
template <int G4>
__device__ __forceinline__ float quad_bcast(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), G4 * 0x55, 0xF, 0xF, true));
}
__device__ __forceinline__ void put_granule(unsigned long long* p, unsigned epoch, unsigned value) {
    __hip_atomic_store((gu64*)p, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The gather of a step's operand.  Granules are read two at a time (16-byte agent-scope loads through a buffer descriptor: aux 16 = sc1);
// rows hold an even number of granules.  pk[e] = pair index | LDS word << 16 (the map does not change with the step: gather_plan, once per
// launch), bit e of `mine` set where slot e is used.  A pair whose two tags match goes to LDS at once; only the others are read again.
template <int NPT>
__device__ __forceinline__ void gather_plan(int pairs, int pairs_per_row, int row_words, unsigned (&pk)[NPT], unsigned& mine) {
    static_assert(NPT <= 32, "one bit per slot");
    mine = 0u;
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
        const int i = threadIdx.x + e * REC_THREADS;
        pk[e] = 0u;
        if (i < pairs) { pk[e] = (unsigned)i | ((unsigned)((i / pairs_per_row) * row_words + 2 * (i % pairs_per_row)) << 16); mine |= 1u << e; }
    }
}
template <int NPT>
__device__ __forceinline__ bool gather_granules(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_base, const unsigned (&pk)[NPT], unsigned mine, unsigned epoch,
                                                unsigned* __restrict__ dst, int* err) {
    unsigned pending = mine;
    for (unsigned spin = 0;; ++spin) {
        u32x4 x[NPT];
#pragma unroll
        for (int e = 0; e < NPT; ++e)
            if (pending >> e & 1u) x[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_base + (pk[e] & 0xffffu) * 16u, 0, 16);
#pragma unroll
        for (int e = 0; e < NPT; ++e)
            if ((pending >> e & 1u) && x[e][1] == epoch && x[e][3] == epoch) {
                *reinterpret_cast<uint2*>(dst + (pk[e] >> 16)) = make_uint2(x[e][0], x[e][2]);
                pending &= ~(1u << e);
            }
        if (__syncthreads_and(pending == 0u)) break;
        if (spin > SPIN_LIMIT) { if (threadIdx.x == 0) atomicExch(err, 1); return false; }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

struct RecArgs { unsigned long long* gr; int* err; int NW, UPW; unsigned gr_bytes; int stall; };

// ---- forward through time.  grid = 2 * NW workgroups of 512 threads: direction = blockIdx / NW, slice = blockIdx % NW.
// Wave w multiplies the gate columns of units [u0 + 4w, u0 + 4w + 4) (one 16-column tile of the unit-major gate axis).
// KS = 32-wide k-steps held (KP <= 32 KS), MT = 16-row batch tiles.
template <int KS, int MT>
__global__ __launch_bounds__(REC_THREADS) void lstm_fwd_rec_kernel(LstmStepArgs a, RecArgs ra) {
    constexpr int HS = KS * 32 + 8;                              // LDS row of h in bf16 (+16 bytes: rows fall on different banks)
    __shared__ __attribute__((aligned(16))) bf16 hs[MT * 16 * HS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (Tried: all workgroups of a direction on ONE XCD -- a grid of 8 NW under round-robin placement, the others leaving at once -- so that a
    // direction's exchange stays behind one L2: forward 290 -> 273 us per layer, backward 394 -> 415.  Not kept.)
    const int dir = blockIdx.x / ra.NW, slice = blockIdx.x % ra.NW;
    if (*ra.err != 0) return;                                   // an earlier launch of this step timed out: the step is lost already, do not spin again
    if (ra.stall && blockIdx.x == 0) return;                    // fault injection (masr_test_blstm_stall): a workgroup that never publishes
    const int H = a.H, G = 4 * H, KP = a.KP, T = a.T, B = a.B;
    const int u0 = slice * ra.UPW, uend = u0 + ra.UPW < H ? u0 + ra.UPW : H;
    const int GRR = (H + 1) / 2, GR = (GRR + 1) / 2 * 2;         // granules per batch row: real ones / with the pad that makes the row even
    unsigned long long* grd = ra.gr + (long)dir * 2 * B * GR;   // [parity][B][GR]
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ra.gr, 0, ra.gr_bytes, 0x00020000);
    for (int i = tid; i < MT * 16 * HS / 2; i += REC_THREADS) reinterpret_cast<unsigned*>(hs)[i] = 0u;
    // this wave's slice of W_hh as MFMA B fragments: lane (c = lane & 15, q = lane >> 4) holds W[row c of the tile][32 kk + 8 q .. + 8]
    bf16x8 wf[KS];
    const int q = lane >> 4, c16 = lane & 15;
    const int ut = u0 + 4 * wave;                                // first unit of this wave's tile
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int k = kk * 32 + q * 8;
        wf[kk] = (ut + (c16 >> 2) < uend && k < KP) ? ld8(a.whh16[dir] + ((long)4 * ut + c16) * KP + k) : zero8();
    }
    // the (batch row, unit) pair this lane does the gate arithmetic for, per batch tile mt
    const int p = lane & 3, uu = c16 >> 2, rg = lane >> 4, unit = ut + uu;
    float cst[MT], hst[MT];
    int len[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { cst[mt] = 0.f; hst[mt] = 0.f; const int row = mt * 16 + rg * 4 + p; len[mt] = row < B ? a.lens[row] : 0; }
    constexpr int NPT = (MT * 16 * KS * 8 + REC_THREADS - 1) / REC_THREADS;
    unsigned pk[NPT], mine;
    gather_plan<NPT>(B * GR / 2, GR / 2, HS / 2, pk, mine);
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? s : T - 1 - s;
        // input contribution of the accumulator elements (row rg * 4 + i, gate column c16): issued before the wait
        float gxv[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = mt * 16 + rg * 4 + i;
                gxv[mt][i] = (row < B && unit < uend) ? a.gx[dir][((long)row * T + t) * G + 4 * unit + p] : 0.f;
            }
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            const unsigned base = (unsigned)(((long)dir * 2 + ((s - 1) & 1)) * B * GR * 8);
            if (!gather_granules<NPT>(rsrc, base, pk, mine, (unsigned)s, reinterpret_cast<unsigned*>(hs), ra.err)) return;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mma16(ld8(hs + (mt * 16 + c16) * HS + kk * 32 + q * 8), wf[kk], acc[mt]);
            __syncthreads();                                     // every wave has read h before the next step's gather writes it
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // 4 x 4 transpose inside the lane quad: lane p takes accumulator element p (batch row rg * 4 + p) of the quad's four gate lanes
            float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float zi = acc[mt][i] + gxv[mt][i];
                const float v0 = quad_bcast<0>(zi), v1 = quad_bcast<1>(zi), v2 = quad_bcast<2>(zi), v3 = quad_bcast<3>(zi);
                if (i == p) { z[0] = v0; z[1] = v1; z[2] = v2; z[3] = v3; }
            }
            const int row = mt * 16 + rg * 4 + p;
            const bool valid = row < B && unit < uend, active = valid && t < len[mt];
            float hnew = valid ? hst[mt] : 0.f;                 // (a sequence past its end keeps its state: packed-sequence semantics)
            const float ig = sigm(z[0]), fg = sigm(z[1]), gg = tanhf(z[2]), og = sigm(z[3]);              // torch gate order i, f, g, o
            if (active) {
                const float cn = fg * cst[mt] + ig * gg;
                hnew = og * tanhf(cn);
                cst[mt] = cn;
            }
            hst[mt] = hnew;
            // the exchange first: units (even, odd) of a row share a granule; the odd unit's lane sits 4 lanes up
            const float hodd = __shfl_down(hnew, 4);
            if (valid && !(uu & 1) && s + 1 < T) {
                unsigned long long* gp = grd + (long)(s & 1) * B * GR + (long)row * GR + (unit >> 1);
                put_granule(gp, (unsigned)(s + 1), pack2(hnew, unit + 1 < uend ? hodd : 0.f));
                if ((unit >> 1) == GRR - 1 && GR > GRR) put_granule(gp + 1, (unsigned)(s + 1), 0u);     // the pad granule of an odd row
            }
            if (valid) {
                const long r = (long)row * T + t;
                a.y16[r * (2 * H) + dir * H + unit] = (bf16)(active ? hnew : 0.f);
                if (active) {
                    *reinterpret_cast<f32x4*>(a.act[dir] + r * G + 4 * unit) = f32x4{ig, fg, gg, og};
                    a.c[dir][r * H + unit] = cst[mt];
                }
            }
        }
    }
}

// ---- backward through time.  Same grid.  Wave w: output tile w & 1 (16 hidden units of the slice), quarter w >> 1 of the reduction over
// the 4H gate axis; the four quarters meet in LDS.  KQ = k-steps of 32 held per wave (ceil(4H / 32 / 4) <= KQ).
template <int KQ, int MT>
__global__ __launch_bounds__(REC_THREADS) void lstm_bwd_rec_kernel(LstmStepArgs a, RecArgs ra) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dir = blockIdx.x / ra.NW, slice = blockIdx.x % ra.NW;
    if (*ra.err != 0) return;                                   // (see lstm_fwd_rec_kernel)
    const int H = a.H, G = 4 * H, T = a.T, B = a.B;
    const int DS = G + 8;                                        // LDS row of dz in bf16 (+16 bytes)
    bf16* dzs = reinterpret_cast<bf16*>(smem_raw);               // [MT * 16][DS]
    float* red = reinterpret_cast<float*>(smem_raw + (size_t)MT * 16 * DS * sizeof(bf16));   // [4 quarters][2 tiles][MT][16][16]
    const int u0 = slice * ra.UPW, uend = u0 + ra.UPW < H ? u0 + ra.UPW : H;
    const int GR = 2 * H;                                        // granules per batch row (4H / 2; even)
    unsigned long long* grd = ra.gr + (long)dir * 2 * B * GR;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ra.gr, 0, ra.gr_bytes, 0x00020000);
    for (int i = tid; i < MT * 16 * DS / 2; i += REC_THREADS) reinterpret_cast<unsigned*>(dzs)[i] = 0u;
    const int NKS = G / 32, KQN = (NKS + 3) / 4;                 // k-steps in all / per quarter
    const int tj = wave & 1, kp = wave >> 1, q = lane >> 4, c16 = lane & 15;
    const int k_lo = kp * KQN, k_hi = k_lo + KQN < NKS ? k_lo + KQN : NKS;
    bf16x8 wt[KQ];
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
        const int n = u0 + 16 * tj + c16, ks = k_lo + kk;
        wt[kk] = (n < uend && ks < k_hi) ? ld8(a.whhT16[dir] + (long)n * G + ks * 32 + q * 8) : zero8();
    }
    // pairs (batch row, unit) of this thread: index tid + 512 e -> row = index / 32, unit u0 + index % 32
    constexpr int NP = MT;
    float dcst[NP];
    int plen[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) { dcst[e] = 0.f; const int row = (tid + e * REC_THREADS) >> 5; plen[e] = row < B ? a.lens[row] : 0; }
    constexpr int NPT = (MT * 16 * KQ * 4 * 8 + REC_THREADS - 1) / REC_THREADS;
    unsigned pk[NPT], mine;
    gather_plan<NPT>(B * GR / 2, GR / 2, DS / 2, pk, mine);
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = dir == 0 ? T - 1 - s : s;
        const int tprev = dir == 0 ? t - 1 : t + 1;              // the step whose cell state entered this one in the forward pass
        // saved forward values of this thread's pairs: issued before the wait
        float dyv[NP], cn[NP], cp[NP];
        f32x4 gate[NP];
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const int idx = tid + e * REC_THREADS, row = idx >> 5, unit = u0 + (idx & 31);
            const bool on = row < B && unit < uend && t < plen[e];
            const long r = (long)row * T + t;
            dyv[e] = on ? a.dy[r * (2 * H) + dir * H + unit] : 0.f;
            gate[e] = on ? *reinterpret_cast<const f32x4*>(a.act[dir] + r * G + 4 * unit) : f32x4{0.f, 0.f, 0.f, 0.f};
            cn[e] = on ? a.c[dir][r * H + unit] : 0.f;
            const bool first = dir == 0 ? t == 0 : t == plen[e] - 1;   // first forward step of the sequence: c_prev = 0
            cp[e] = (on && !first) ? a.c[dir][((long)row * T + tprev) * H + unit] : 0.f;
        }
        if (s > 0) {
            const unsigned base = (unsigned)(((long)dir * 2 + ((s - 1) & 1)) * B * GR * 8);
            if (!gather_granules<NPT>(rsrc, base, pk, mine, (unsigned)s, reinterpret_cast<unsigned*>(dzs), ra.err)) return;
            __syncthreads();
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                const int ks = k_lo + kk < NKS ? k_lo + kk : NKS - 1;      // (past the quarter's end the weight fragment is zero)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mma16(ld8(dzs + (mt * 16 + c16) * DS + ks * 32 + q * 8), wt[kk], acc[mt]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) red[(((kp * 2 + tj) * MT + mt) * 16 + q * 4 + i) * 16 + c16] = acc[mt][i];
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            const int idx = tid + e * REC_THREADS, row = idx >> 5, ul = idx & 31, unit = u0 + ul;
            if (row >= B || unit >= uend) continue;
            const int mt = row >> 4, rr = row & 15;
            float dh = dyv[e];
            if (s > 0) {
                const int o = (((ul >> 4) * MT + mt) * 16 + rr) * 16 + (ul & 15), qs = 2 * MT * 256;
                dh += (red[o] + red[o + qs]) + (red[o + 2 * qs] + red[o + 3 * qs]);
            }
            const long r = (long)row * T + t;
            float dz0 = 0.f, dz1 = 0.f, dz2 = 0.f, dz3 = 0.f;
            if (t < plen[e]) {
                const float ig = gate[e][0], fg = gate[e][1], gg = gate[e][2], og = gate[e][3];
                const float tc = tanhf(cn[e]);
                const bool last = dir == 0 ? t == plen[e] - 1 : t == 0;   // last forward step of the sequence: nothing flows in
                const float dc = dh * og * (1.f - tc * tc) + (last ? 0.f : dcst[e]);
                dcst[e] = dc * fg;
                dz0 = dc * gg * ig * (1.f - ig);
                dz1 = dc * cp[e] * fg * (1.f - fg);
                dz2 = dc * ig * (1.f - gg * gg);
                dz3 = dh * tc * og * (1.f - og);
            }
            const unsigned lo = pack2(dz0, dz1), hi = pack2(dz2, dz3);
            if (s + 1 < T) {
                unsigned long long* gp = grd + (long)(s & 1) * B * GR + (long)row * GR + 2 * unit;
                put_granule(gp, (unsigned)(s + 1), lo);
                put_granule(gp + 1, (unsigned)(s + 1), hi);
            }
            *reinterpret_cast<uint2*>(a.dz16[dir] + r * G + 4 * unit) = make_uint2(lo, hi);
        }
    }
}

int rec_slices(int H) { return (H + REC_UNITS - 1) / REC_UNITS; }
int rec_units(int H) { const int nw = rec_slices(H); return ((H + nw - 1) / nw + 3) / 4 * 4; }
int fwd_row_granules(int H) { return ((H + 1) / 2 + 1) / 2 * 2; }

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : (mk_set_error(__func__, "launch failed"), -1))

// shapes the resident kernels hold: the weight slice of a workgroup in registers (KP <= 384), two batch tiles, whole k-steps
bool mk_lstm_rec_ok(int B, int H, int KP) { return B >= 1 && B <= 32 && H >= 4 && KP <= 384 && KP % 32 == 0 && KP >= H && (4 * H) % 32 == 0; }
int64_t mk_lstm_rec_words(int B, int H) { return (int64_t)2 * 2 * B * 2 * H + 2; }      // 64-bit words: the granules of the wider (backward) exchange

template <int MT>
static void fwd_rec_launch(const LstmStepArgs& a, const RecArgs& ra, hipStream_t s) {
    const dim3 grid(2 * ra.NW);
    const int ks = a.KP / 32;
    if (ks <= 4) hipLaunchKernelGGL((lstm_fwd_rec_kernel<4, MT>), grid, dim3(REC_THREADS), 0, s, a, ra);
    else if (ks <= 8) hipLaunchKernelGGL((lstm_fwd_rec_kernel<8, MT>), grid, dim3(REC_THREADS), 0, s, a, ra);
    else hipLaunchKernelGGL((lstm_fwd_rec_kernel<12, MT>), grid, dim3(REC_THREADS), 0, s, a, ra);
}
int mk_lstm_fwd_rec(const LstmStepArgs& a, unsigned long long* words, int* err, hipStream_t s, int test_stall) {
    if (!mk_lstm_rec_ok(a.B, a.H, a.KP)) { mk_set_error("mk_lstm_fwd_rec", "shape not covered by the resident recurrence"); return -1; }
    // tags of the forward exchange: [2 dirs][2 parities][B][granules per row] (the block starts its allocation; zeroed in whole 16 bytes)
    const size_t bytes = sizeof(unsigned long long) * (size_t)2 * 2 * a.B * fwd_row_granules(a.H);
    RecArgs ra{words, err, rec_slices(a.H), rec_units(a.H), (unsigned)bytes, test_stall};
    if (hipMemsetAsync(words, 0, bytes, s) != hipSuccess) { mk_set_error("mk_lstm_fwd_rec", "memset failed"); return -1; }
    if (a.B <= 16) fwd_rec_launch<1>(a, ra, s); else fwd_rec_launch<2>(a, ra, s);
    return LAUNCH_OK();
}
template <int KQ, int MT>
static int bwd_rec_launch1(const LstmStepArgs& a, const RecArgs& ra, hipStream_t s) {
    const size_t lds = (size_t)MT * 16 * (4 * a.H + 8) * sizeof(bf16) + sizeof(float) * 4 * 2 * MT * 16 * 16;
    // once per instantiation AND device (a set bit = raised on that device; two threads racing here both raise it: harmless)
    static std::atomic<unsigned long long> raised{0};
    int dev = 0; hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_rec_kernel<KQ, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256) != hipSuccess) return -1;
        raised.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((lstm_bwd_rec_kernel<KQ, MT>), dim3(2 * ra.NW), dim3(REC_THREADS), lds, s, a, ra);
    return 0;
}
int mk_lstm_bwd_rec(const LstmStepArgs& a, unsigned long long* words, int* err, hipStream_t s) {
    if (!mk_lstm_rec_ok(a.B, a.H, a.KP)) { mk_set_error("mk_lstm_bwd_rec", "shape not covered by the resident recurrence"); return -1; }
    const size_t bytes = sizeof(unsigned long long) * (size_t)2 * 2 * a.B * 2 * a.H;
    RecArgs ra{words, err, rec_slices(a.H), rec_units(a.H), (unsigned)bytes, 0};
    if (hipMemsetAsync(words, 0, bytes, s) != hipSuccess) { mk_set_error("mk_lstm_bwd_rec", "memset failed"); return -1; }
    const int kq = (4 * a.H / 32 + 3) / 4;
    int rc;
    if (a.B <= 16) rc = kq <= 4 ? bwd_rec_launch1<4, 1>(a, ra, s) : kq <= 8 ? bwd_rec_launch1<8, 1>(a, ra, s) : bwd_rec_launch1<12, 1>(a, ra, s);
    else rc = kq <= 4 ? bwd_rec_launch1<4, 2>(a, ra, s) : kq <= 8 ? bwd_rec_launch1<8, 2>(a, ra, s) : bwd_rec_launch1<12, 2>(a, ra, s);
    if (rc) { mk_set_error("mk_lstm_bwd_rec", "could not raise the dynamic LDS limit"); return -1; }
    return LAUNCH_OK();
}

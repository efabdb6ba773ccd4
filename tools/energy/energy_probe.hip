// Energy per instruction on gfx950, for the costing of DESIGN 6.2 (Winograd trades MFMAs for vector instructions under a board power cap):
// back-to-back loops of (0) v_mfma_f32_16x16x32_bf16, (1) v_fma_f32, (2) one MFMA + 9 FMAs, (3) ds_read_b128 (conflict-free, 1 KB per
// wave-instruction), (4) one MFMA + one ds_read_b128 whose result is the MFMA's next A operand, (5) v_mfma_f32_32x32x16_bf16
// (twice the FLOPs of (0) per instruction from the same operand registers) -- on random contents, one wave per SIMD on every CU, each run for a few seconds while tools/energy/run.sh samples hwmon power1_average.  Not part of the product; nothing links it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/energy_probe tools/energy/energy_probe.hip && /tmp/energy_probe <mode> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256) void burn(const float* __restrict__ seed, float* __restrict__ sink, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x4 acc[8];
    float v[16];
    bf16x8 a[2], b[2];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = seed[(tid * 37 + i * 4 + j) & 65535];
    for (int i = 0; i < 16; ++i) v[i] = seed[(tid * 53 + i) & 65535];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)seed[(tid * 11 + i * 8 + j) & 65535]; b[i][j] = (__bf16)seed[(tid * 7 + i * 8 + j + 99) & 65535]; }
    const float c0 = seed[tid & 1023] * 1e-3f + 0.999f, c1 = seed[(tid + 1) & 1023] * 1e-3f;
    __shared__ __attribute__((aligned(16))) float lds[MODE >= 3 ? 16384 : 4];      // 64 KB of random bits: 16 x 1 KB rows per wave
    if (MODE >= 3) { for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = seed[(i * 3 + blockIdx.x) & 65535]; __syncthreads(); }
    const unsigned lbase = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds) + (threadIdx.x >> 6) * 16384 + (threadIdx.x & 63) * 16;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 rd[8];
    for (int i = 0; i < 8; ++i) rd[i] = u4{0u, 0u, 0u, 0u};
    f32x16 big[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = MODE == 5 ? seed[(tid * 29 + i * 16 + j) & 65535] : 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 1], b[(i >> 1) & 1], big[i], 0, 0, 0);
            if ((it & 63) == 63) for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] *= 1e-3f;
        }
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[i]) : "v"(lbase), "n"(i * 1024));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[i]) : "v"(lbase), "n"(i * 1024));
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, rd[(i + 4) & 7]), b[i & 1], acc[i], 0, 0, 0);
                if (i == 3 || i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if ((it & 63) == 63) for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
        }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
                if (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) v[(i * 9 + k) & 15] = __builtin_fmaf(v[(i * 9 + k) & 15], c0, c1);
                }
            }
            // keep the accumulators bounded without leaving the MFMA pipe idle for long
            if ((it & 63) == 63) for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = __builtin_fmaf(v[k], c0, c1);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += big[i][j];
    for (int i = 0; i < 8; ++i) s += __builtin_bit_cast(float, rd[i][0] & 0x3f800000u);
    if (s == 123.456f) sink[tid] = s;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<float> h(65536);
    srand(1); for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *seed, *sink; hipMalloc(&seed, 65536 * 4); hipMalloc(&sink, (size_t)ncu * 256 * 4); hipMemcpy(seed, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(burn<0>, dim3(ncu), dim3(256), 0, 0, seed, sink, iters);
        else if (mode == 1) hipLaunchKernelGGL(burn<1>, dim3(ncu), dim3(256), 0, 0, seed, sink, iters);
        else if (mode == 2) hipLaunchKernelGGL(burn<2>, dim3(ncu), dim3(256), 0, 0, seed, sink, iters);
        else if (mode == 3) hipLaunchKernelGGL(burn<3>, dim3(ncu), dim3(256), 0, 0, seed, sink, iters);
        else if (mode == 5) hipLaunchKernelGGL(burn<5>, dim3(ncu), dim3(256), 0, 0, seed, sink, iters);
        else hipLaunchKernelGGL(burn<4>, dim3(ncu), dim3(256), 0, 0, seed, sink, iters);
    };
    launch(); hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) { for (int k = 0; k < 4; ++k) launch(); hipDeviceSynchronize(); n += 4; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double waves = (double)ncu * 4, per_wave_iter = (double)n * iters;
    const double mfma = (mode == 1 || mode == 3 ? 0.0 : 8.0) * per_wave_iter * waves, valu = (mode == 1 ? 128.0 : mode == 2 ? 72.0 : 0.0) * per_wave_iter * waves;
    const double lds = (mode == 3 ? 16.0 : mode == 4 ? 8.0 : 0.0) * per_wave_iter * waves;
    if (mode == 5) { printf("mode 5: %.2f s, %.3e MFMA/s (32x32x16 bf16 = 32768 FLOP each)\n", dt, 8.0 * per_wave_iter * waves / dt); return 0; }
    printf("mode %d: %.2f s, %.3e MFMA/s (16x16x32 bf16), %.3e vector instructions/s (wave64 v_fma_f32), %.3e ds_read_b128/s (1 KB each)\n", mode, dt, mfma / dt, valu / dt, lds / dt);
    return 0;
}

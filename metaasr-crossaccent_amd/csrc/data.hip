// Ragged gather + zero pad of the numpy-memmap fbank shard rows (reference: CommonVoiceDataset.__getitem__
// and collate_fn, src/io/dataset.py:21-33,147-153) as one coalesced HBM pass: the shard [sum T_i][D] is
// resident in HBM, each utterance is rows [row_start, row_start+len).
#include "common.h"
#include "kernels.h"

namespace {
__global__ void gather_pad_kernel(const float* __restrict__ feat, const long* __restrict__ row_start,
                                  const int* __restrict__ lens, float* __restrict__ xs, int B, int Tmax, int D) {
    const long n = (long)B * Tmax * D;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int d = (int)(i % D);
        const int t = (int)((i / D) % Tmax);
        const int b = (int)(i / ((long)D * Tmax));
        xs[i] = t < lens[b] ? feat[(row_start[b] + t) * D + d] : 0.f;
    }
}
}  // namespace

int mk_gather_pad(const float* feat, const long* row_start, const int* lens, float* xs, int B, int Tmax, int D, hipStream_t s) {
    const long n = (long)B * Tmax * D;
    if (n == 0) return 0;
    long nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(gather_pad_kernel, dim3((unsigned)nb), dim3(256), 0, s, feat, row_start, lens, xs, B, Tmax, D);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

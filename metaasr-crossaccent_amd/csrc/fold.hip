// The tail of a training step's backward pass as ONE launch (engine.hip backward): the slab reduces of the three 3x3-conv weight
// gradients and of conv1's (fused into conv2's dgrad), the dgamma / dbeta folds of every LayerNorm, the un-permutation of vgg2enc's weight
// gradient into the reference's feature order, and the embedding backward (mono_transformer_torch.py:49-62,129-133: the autograd of those
// modules).  They are independent of each other and were eight launches of 5-25 us, mostly launch latency, at the end of every step.
// Job bodies: folds.h (the same device functions their own kernels call); the job list travels by value in the kernel arguments.
#include "kernels.h"
#include "folds.h"

namespace {

__global__ __launch_bounds__(256) void backward_folds_kernel(const FoldJobs j) {
    int b = blockIdx.x;
    // longest jobs first: the embedding rows (tiny workgroups, a dependent chain start -> positions -> rows), then the big slab reduces
    const int n_embed = j.embed.dtable ? j.embed.V * (j.embed.E / 64) : 0;
    if (b < n_embed) {
        embed_bwd_body(j.embed.order, j.embed.start, j.embed.dy, j.embed.dtable, j.embed.E, j.embed.accumulate, j.embed.drop_p, j.embed.seed, j.embed.site,
                       j.embed.seed_ptr, b % j.embed.V, b / j.embed.V);
        return;
    }
    b -= n_embed;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int nb = k < j.nconv ? (j.conv[k].COUT * 9 * j.conv[k].CIN + j.conv[k].COUT + 63) / 64 : 0;
        if (b < nb) { conv3x3_wgrad_reduce_body(j.conv[k].slab, j.conv[k].nsplit, j.conv[k].dw, j.conv[k].db, j.conv[k].CIN, j.conv[k].COUT, b); return; }
        b -= nb;
    }
    const int n_unp = j.unperm.dw ? (int)(((long)j.unperm.E * j.unperm.C * j.unperm.Dp + 255) / 256) : 0;
    if (b < n_unp) { vgg2enc_unpermute_body(j.unperm.g, j.unperm.dw, j.unperm.E, j.unperm.C, j.unperm.Dp, b); return; }
    b -= n_unp;
    const int ln_x = (2 * j.E + 31) / 32, n_ln = ln_x * j.ln.n;
    if (b < n_ln) { const LnReduceDesc& d = j.ln.p[b / ln_x]; ln_bwd_reduce_body(d.slab, d.nblocks, d.dgamma, d.dbeta, j.E, b % ln_x); return; }
    b -= n_ln;
    if (j.conv1.dw && b < 40) conv1_wgrad_reduce256_body(j.conv1.slab, j.conv1.nblocks, j.conv1.dw, j.conv1.db, b);
}

}  // namespace

int mk_backward_folds(const FoldJobs& j, hipStream_t s) {
    if (j.conv1.dw && j.conv1.nblocks > 256) { mk_set_error("mk_backward_folds", "conv1 slab: at most 256 rows"); return -1; }
    if (j.nconv < 0 || j.nconv > 3 || j.ln.n < 0 || j.ln.n > LN_GROUP_MAX || (j.embed.dtable && j.embed.E % 64)) { mk_set_error("mk_backward_folds", "bad job list"); return -1; }
    long blocks = j.embed.dtable ? (long)j.embed.V * (j.embed.E / 64) : 0;
    for (int k = 0; k < j.nconv; ++k) blocks += (j.conv[k].COUT * 9 * j.conv[k].CIN + j.conv[k].COUT + 63) / 64;
    if (j.unperm.dw) blocks += ((long)j.unperm.E * j.unperm.C * j.unperm.Dp + 255) / 256;
    blocks += (long)((2 * j.E + 31) / 32) * j.ln.n;
    if (j.conv1.dw) blocks += 40;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(backward_folds_kernel, dim3((unsigned)blocks), dim3(256), 0, s, j);
    if (hipGetLastError() != hipSuccess) { mk_set_error("mk_backward_folds", "launch failed"); return -1; }
    return 0;
}

import ctypes as C, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
import masr_amd
from masr_amd import _cabi
L = _cabi.lib()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(tag, M, N, K, rm=False, iters=30, rot=8, c16=True):
    sets=[]
    for _ in range(rot):
        if rm:   # A = dY [k=M][i=N_out], B = X [k=M][j=K_out]; test_gemm(A, lda, B, ldb, M(out rows), N(out cols), K(reduction))
            sets.append(dict(A=torch.randn(K, M, device="cuda").bfloat16(), B=torch.randn(K, N, device="cuda").bfloat16(), C=torch.zeros(M, N, device="cuda")))
        else:
            sets.append(dict(A=torch.randn(M, K, device="cuda").bfloat16(), B=torch.randn(N, K, device="cuda").bfloat16(), C=torch.zeros(M, N, device="cuda")))
    def call(d):
        if rm: _cabi.check(L.masr_test_gemm(P(d["A"]), M, P(d["B"]), N, M, N, K, 1, None, 0, P(d["C"]), N, S()), "g")
        else: _cabi.check(L.masr_test_gemm(P(d["A"]), K, P(d["B"]), K, M, N, K, 0, None, 0, P(d["C"]), N, S()), "g")
    with torch.cuda.stream(torch.cuda.Stream()):
        for i in range(3): call(sets[i % rot])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): call(sets[i % rot])
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{tag:28s} M={M:5d} N={N:5d} K={K:5d} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")
print("NT (fwd / dgrad), M = 4000 rows")
for tag,N,K in (("qkv fwd",1536,512),("out fwd / dgrad",512,512),("ffn1 fwd",2048,512),("ffn2 fwd",512,2048),("vgg2enc fwd",512,2560),("kv_all fwd",4096,512),
                ("qkv dgrad",512,1536),("ffn1 dgrad",512,2048),("ffn2 dgrad",2048,512),("kv dgrad",512,4096),("vgg2enc dgrad",2560,512)):
    run(tag, 4000, N, K)
print("RM (wgrad): out [N_w][K_w], reduction over 4000 rows (no split-K in this entry)")
for tag,Mo,No in (("qkv wgrad",1536,512),("out wgrad",512,512),("ffn1 wgrad",2048,512),("ffn2 wgrad",512,2048),("vgg2enc wgrad",512,2560),("kv_all wgrad",4096,512)):
    run(tag, Mo, No, 4000, rm=True)

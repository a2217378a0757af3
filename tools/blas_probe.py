"""Development aid: fp32 GEMM time of the BERT-base linears under rocBLAS and hipBLASLt (torch's two BLAS back ends)."""
import time, torch
dev = torch.device("cuda:0")
shapes = [("qkv/out 4096x768x768", 4096, 768, 768), ("ffn up 4096x3072x768", 4096, 3072, 768), ("ffn down 4096x768x3072", 4096, 768, 3072),
          ("squad 12288x768x768", 12288, 768, 768), ("squad 12288x3072x768", 12288, 3072, 768), ("squad 12288x768x3072", 12288, 768, 3072)]
for lib in ("cublas", "cublaslt"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev)
        for _ in range(5):
            y = torch.nn.functional.linear(x, w, b)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = torch.nn.functional.linear(x, w, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"{lib:9s} {name:26s} {us:8.2f} us  {2 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
    # batched attention matmuls
    q = torch.randn(32, 12, 128, 64, device=dev); k = torch.randn(32, 12, 128, 64, device=dev)
    for _ in range(5):
        s = torch.matmul(q, k.transpose(-1, -2))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        s = torch.matmul(q, k.transpose(-1, -2))
    e1.record(); torch.cuda.synchronize()
    print(f"{lib:9s} QK^T [32,12,128,64]x[.,64,128] {e0.elapsed_time(e1) * 1e3 / 50:8.2f} us", flush=True)

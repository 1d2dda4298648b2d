"""What the vendor library reaches on the engine's GEMM shapes (bf16, random data): a yardstick for gemm2, not a product path."""
import torch, time
torch.manual_seed(0)
dev = "cuda"
M = 73728      # the first encode slice of the bench hour: 144 chunks x 512 frames
for (N, K) in ((4096, 1024), (1024, 4096), (1024, 1024), (3072, 1024), (2048, 1024), (1024, 19456)):
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        C = A @ W.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        C = A @ W.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s", flush=True)
    del A, W, C

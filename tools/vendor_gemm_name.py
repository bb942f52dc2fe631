"""Which hipBLASLt kernel serves the four GritLM-7B GEMM shapes?  Run under `rocprofv3 --kernel-trace --stats`: the kernel NAME carries the
vendor kernel's macro tile, MFMA instruction, workgroup shape and staging options (context for DESIGN.md §4)."""
import torch
dev = torch.device("cuda:0")
M = 131072
for N, K in ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    del a, w, c

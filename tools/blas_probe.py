import torch
M,N,K=131072,6144,4096
a=torch.randn((M,K),device="cuda").to(torch.bfloat16); w=(torch.randn((N,K),device="cuda")*0.02).to(torch.bfloat16)
for _ in range(3): torch.matmul(a,w.t())
torch.cuda.synchronize()
a2=torch.randn((M,14336),device="cuda").to(torch.bfloat16); w2=(torch.randn((4096,14336),device="cuda")*0.02).to(torch.bfloat16)
for _ in range(3): torch.matmul(a2,w2.t())
torch.cuda.synchronize()

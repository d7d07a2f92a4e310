import torch
M=31162
for (N,K) in [(768,3072),(3072,768),(2304,768),(768,768)]:
    x=torch.randn(M,K,device="cuda").bfloat16(); w=torch.randn(N,K,device="cuda").bfloat16()
    for _ in range(5): y=torch.matmul(x,w.t())
    torch.cuda.synchronize()
# TN shapes: dw[N,K] = dy[M,N]^T x[M,K]
for (N,K) in [(3072,768),(768,768)]:
    dy=torch.randn(M,N,device="cuda").bfloat16(); x=torch.randn(M,K,device="cuda").bfloat16()
    for _ in range(5): y=torch.matmul(dy.t(),x)
    torch.cuda.synchronize()

import torch
dev="cuda"
for M,N,K in [(8192,8192,8192),(43840,4096,1024),(43840,1024,4096),(43840,3072,1024),(43840,1024,1024)]:
    A=torch.randn(M,K,device=dev).half(); W=torch.randn(N,K,device=dev).half()
    for _ in range(3): torch.matmul(A,W.t())
torch.cuda.synchronize()

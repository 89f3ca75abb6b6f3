import torch
for m, n, k in [(8192, 8192, 8192), (201728, 3072, 768), (201728, 768, 3072)]:
    a = torch.randn(m, k, device="cuda").half(); w = torch.randn(n, k, device="cuda").half()
    for _ in range(3): torch.matmul(a, w.t())
    torch.cuda.synchronize()

import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from proto_clip_amd.model import Adapter
torch.manual_seed(0)
ad = Adapter(512, "conv-3x", dtype=torch.half).cuda()
x = torch.nn.functional.normalize(torch.randn(4096, 512, device="cuda"), dim=-1).half()
with torch.no_grad():
  for _ in range(3):
      y = ad(x, l2norm_out=True)
torch.cuda.synchronize()

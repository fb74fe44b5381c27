import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/experiment-yolo_amd", "/root/repo/tests"]
import torch, torch.nn.functional as F
from ultralytics.hip.engine import ConvSpec, Engine, Storage
eng = Engine("cuda:0")
cin, cout, ks, H, W, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), 2
torch.manual_seed(0)
x = torch.randn(N, cin, H, W).half().float()
w = (torch.randn(cout, cin, ks, ks) / (cin*ks*ks) ** 0.5).half().float()
sp = ConvSpec("b", w.cuda(), None, None, ks, 1, 0); sp.gweight = torch.zeros_like(sp.weight)
eng.prepare_conv(sp); eng.pack(sp)
xs = Storage(eng, N, H, W, cin); xs.buf.copy_(x.permute(0, 2, 3, 1).half().cuda())
y = torch.zeros(N, H, W, cout, dtype=torch.float16, device="cuda")
eng._conv_raw(sp, xs.act(), y.data_ptr(), cout, 0)
torch.cuda.synchronize()
ref = F.conv2d(x, w, None, 1, ks // 2).permute(0, 2, 3, 1)
d = (y.float().cpu() - ref).abs()
bad = (d > 0.02).nonzero()
print("max err", d.max().item(), "bad count", len(bad))
print(bad[:20])
import collections
print("bad rows", sorted(collections.Counter(bad[:, 1].tolist()).items()))
print("bad cols", sorted(collections.Counter(bad[:, 2].tolist()).items()))
print("bad ch", sorted(collections.Counter(bad[:, 3].tolist()).items()))

"""Stand-alone duration of the fused recurrent policy step (apx_lstm_step) at the rollout's shape, back to back (weights hot in L2) and with a cache-flushing
write in between (weights cold, as behind a 2.3 ms env step):  python tools/t_lstm_step.py   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apex_amd import engine
dev = torch.device("cuda:0")
for B in (2048, 4096):
    net = engine.Lstm(49, 128, 2, 10, dev); net.params.normal_(0, 0.05); net.pack_step()
    x = torch.randn(B, 49, device=dev); hc = torch.zeros(2, 2, B, 128, device=dev); noise = torch.randn(B, 10, device=dev); act = torch.empty(B, 10, device=dev)
    mean = torch.zeros(49, device=dev); std = torch.ones(49, device=dev)
    junk = torch.empty(64 << 20, device=dev)      # 256 MB: larger than L2 + MALL
    for cold in (False, True):
        for _ in range(5): net.step(x, hc, mean, std, noise=noise, sigma=0.1, act_out=act)
        tot = 0.0; n = 50
        for _ in range(n):
            if cold: junk.fill_(1.0)
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); net.step(x, hc, mean, std, noise=noise, sigma=0.1, act_out=act); b.record(); torch.cuda.synchronize()
            tot += a.elapsed_time(b)
        print("B = %d %s: %.1f us per call (event pair around one launch)" % (B, "cold" if cold else "hot", tot / n * 1e3))

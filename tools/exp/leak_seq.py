#!/usr/bin/env python3
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from test_gpu_leaks import _cycle
f0, _ = torch.cuda.mem_get_info(0)
seq = []
for i in range(10):
    _cycle(); torch.cuda.synchronize()
    f, _ = torch.cuda.mem_get_info(0)
    seq.append((f0 - f) / 2**20)
print("MiB in use after each cycle (relative to the start):", ["%.0f" % v for v in seq])

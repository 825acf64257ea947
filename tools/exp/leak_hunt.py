#!/usr/bin/env python3
"""which stage of tests/test_gpu_leaks.py::_cycle keeps device memory?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from test_gpu_leaks import _cycle
for st in (sys.argv[1:] or ["", "a", "b", "c", "d", "e", "f", "g", "h"]):
    _cycle(stages=st); torch.cuda.synchronize()
    f1, _ = torch.cuda.mem_get_info(0)
    for _ in range(3): _cycle(stages=st)
    torch.cuda.synchronize()
    f4, _ = torch.cuda.mem_get_info(0)
    print("stages %-3r: %.1f MiB kept over three cycles" % (st, (f1 - f4) / 2**20), flush=True)

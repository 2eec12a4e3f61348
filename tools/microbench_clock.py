import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xuance_amd import _lib
from xuance_amd._lib import stream_ptr
from probe_lib import call
out = torch.zeros(2, dtype=torch.int64, device="cuda"); sink = torch.zeros(64 * 1024, device="cuda")
for blocks in (1, 16, 256, 1024):
    for iters in (1000, 20000):
        call("xrl_probe_mfma_chain", iters, blocks, out.data_ptr(), sink.data_ptr(), stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call("xrl_probe_mfma_chain", iters, blocks, out.data_ptr(), sink.data_ptr(), stream_ptr()); e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        c, w = out.tolist()
        print(f"blocks={blocks:5d} iters={iters:6d}: {us:9.1f} us, {c/iters:6.1f} shader-cycles/MFMA, wall ticks {w} "
              f"=> shader clock ~ {c / (w / 100e6) / 1e9:.2f} GHz (if wall clock is 100 MHz), us/MFMA = {us/iters*1e3:.1f} ns")

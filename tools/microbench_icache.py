"""Cold vs warm instruction fetch on the GPU box: 2048 dependent VALU ops of straight-line code (16 KB)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xuance_amd._lib import stream_ptr
from probe_lib import call
out = torch.zeros(8, dtype=torch.int64, device="cuda"); sink = torch.zeros(512 * 1024, device="cuda")
for blocks, threads in ((1, 64), (1, 512), (24, 512), (256, 512)):
    for rep in range(3):
        out.zero_()
        call("xrl_probe_icache", 4, blocks, threads, out.data_ptr(), sink.data_ptr(), stream_ptr())
        torch.cuda.synchronize()
    print(f"blocks={blocks:4d} threads={threads:4d}: cycles per pass (2048 dependent v_fma, 16 KB code):", out.tolist()[:4])

for blocks, threads in ((1, 64), (1, 512), (24, 512)):
    for rep in range(3):
        out.zero_()
        call("xrl_probe_ijump", 4, blocks, threads, out.data_ptr(), sink.data_ptr(), stream_ptr())
        torch.cuda.synchronize()
    print(f"blocks={blocks:4d} threads={threads:4d}: cycles per pass of 16 far branches (2 KB apart):", out.tolist()[:4])

"""Cost and correctness of a counter barrier + data exchange between workgroups pinned to one XCD (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xuance_amd._lib import stream_ptr
from probe_lib import call
for variant in (2, 3):
  for n_wg in (2, 8, 24):
    counter = torch.zeros(64, dtype=torch.int32, device="cuda")
    slots = torch.zeros(32 * 64, device="cuda")
    out = torch.zeros(40, dtype=torch.int64, device="cuda")
    call("xrl_probe_xcd_barrier", 2000 | (variant << 24), n_wg, counter.data_ptr(), slots.data_ptr(), out.data_ptr(), stream_ptr())
    torch.cuda.synchronize()
    o = out.tolist()
    print(f"variant {variant} n_wg={n_wg:3d}: {o[0]} cycles per iteration (2 barriers + exchange), wrong values {o[1]}, timeouts {o[2]}, XCC ids {sorted(set(o[3:3 + n_wg]))}")

import sys
sys.path.insert(0, "/root/repo")
from tools import bench_secondary as bs
for fused in (True, False):
    orig = bs._qmix_cfg
    def cfg(n, rnn, _o=orig, _f=fused):
        c = _o(n, rnn); c.use_fused_acting = _f; return c
    bs._qmix_cfg = cfg
    r = bs.qmix_3m(True)
    bs._qmix_cfg = orig
    print("fused acting", fused, r["value"], r["update_us"], flush=True)

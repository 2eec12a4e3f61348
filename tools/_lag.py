import sys
sys.path.insert(0, "/root/repo")
from tools import bench_secondary as bs
for K in (2, 4, 8):
    orig = bs._qmix_cfg
    def cfg(n, rnn, _o=orig, _k=K):
        c = _o(n, rnn); c.episode_loop_unroll = _k; return c
    bs._qmix_cfg = cfg
    r = bs.qmix_3m(True)
    bs._qmix_cfg = orig
    print("unroll", K, r["value"], r["update_us"], flush=True)

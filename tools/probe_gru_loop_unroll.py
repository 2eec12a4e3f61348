"""Recurrent QMIX episode loop (bench_secondary.qmix_3m(True)) at several (episode_loop_unroll, episode_loop_lag) settings."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_secondary as bs
base = bs._qmix_cfg
out = []
for unroll, lag in ((4, 1), (8, 1), (2, 2), (6, 1), (4, 2), (8, 2), (4, 1)):
    def cfg(n, rnn, _u=unroll, _l=lag):
        c = base(n, rnn)
        c.episode_loop_unroll, c.episode_loop_lag = _u, _l
        return c
    bs._qmix_cfg = cfg
    r = bs.qmix_3m(True)
    rec = dict(unroll=unroll, lag=lag, value=r["value"], update_us=r["update_us"])
    print(json.dumps(rec)); out.append(rec)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

import os, sys, threading, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden"))
import cases
from papteam_opticalflow_amd import Papof
a, b = cases.load_pair("1920")
for n_sor in (9, 33, 7):
    hs = [Papof(0) for _ in range(3)]
    run = lambda h: h.coarse2fine_flow_sched(a, b, 1, 0.012, 0.75, 2, 0, 1, n_sor, 0)[0]
    want = run(hs[0])
    assert np.array_equal(want, run(hs[0]))
    bad = [0, 0, 0]
    def work(i):
        for _ in range(12):
            bad[i] += not np.array_equal(run(hs[i]), want)
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in th]; [t.join() for t in th]
    print("n_sor", n_sor, "three handles in flight, default order: wrong results per handle", bad, "of 12", flush=True)
    [h.close() for h in hs]

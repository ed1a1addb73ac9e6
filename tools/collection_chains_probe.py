import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden"))
import cases
from papteam_opticalflow_amd import flow_collection
for res, n_pairs, B in (("240", 192, 32), ("240", 192, 16), ("480", 96, 16)):
    a, b = cases.load_frame_u8(res, 1), cases.load_frame_u8(res, 2)
    video = ([a, b] * (n_pairs // 2 + 1))[:n_pairs + 1]
    for k in (1, 2, 3, 4, 6):
        flow_collection(video, 5, in_flight=k, batch=B, on_pair=lambda *r: None)
        t0 = time.perf_counter()
        flow_collection(video, 5, in_flight=k, batch=B, on_pair=lambda *r: None)
        dt = time.perf_counter() - t0
        print("%s batches of %d, %d chains in flight: %.3f ms per pair, %.0f pairs/s" % (res, B, k, dt / n_pairs * 1e3, n_pairs / dt), flush=True)

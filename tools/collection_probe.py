#!/usr/bin/env python3
"""Throughput of a collection (the reference's TestSuite workload: consecutive pairs of a frame list, reference
schedule, host uint8 frames in, float64 results out) as a function of the number of sequences in flight.
usage: collection_probe.py [res] [pairs] [k,k,...]      env: PAPOF_OVERLAP=0 (one stream per handle), GPU_MAX_HW_QUEUES"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import cases
from papteam_opticalflow_amd import flow_collection
res = sys.argv[1] if len(sys.argv) > 1 else "240"
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8, 16, 32]
a, b = cases.load_frame_u8(res, 1), cases.load_frame_u8(res, 2)
video = ([a, b] * (n_pairs // 2 + 1))[:n_pairs + 1]
h, w, _ = a.shape
tag = "overlap=%s queues=%s" % (os.environ.get("PAPOF_OVERLAP", "1"), os.environ.get("GPU_MAX_HW_QUEUES", "default"))
for k in ks:
    flow_collection(video[:2 * k + 1], 5, in_flight=k, batch=0, on_pair=lambda *r: None)  # handles, arenas, warm-up
    t0 = time.perf_counter()
    flow_collection(video, 5, in_flight=k, batch=0, on_pair=lambda *r: None)
    dt = time.perf_counter() - t0
    print("%sx%s  %3d pairs  %s  in flight %2d : %7.2f ms per pair  %7.2f Mpix/s  %6.1f pairs/s"
          % (w, h, n_pairs, tag, k, dt / n_pairs * 1e3, n_pairs * h * w / 1e6 / dt, n_pairs / dt), flush=True)
# the same collection in batches (csrc/batch.hip): B pairs per launch chain, 1 / 2 chains in flight
for B in [int(x) for x in os.environ.get("PROBE_BATCHES", "8,16,32").split(",")]:
    for k in (1, 2):
        flow_collection(video, 5, in_flight=k, batch=B, on_pair=lambda *r: None)
        t0 = time.perf_counter()
        flow_collection(video, 5, in_flight=k, batch=B, on_pair=lambda *r: None)
        dt = time.perf_counter() - t0
        print("%sx%s  %3d pairs  batches of %2d, %d chains in flight : %7.2f ms per pair  %7.2f Mpix/s  %6.1f pairs/s"
              % (w, h, n_pairs, B, k, dt / n_pairs * 1e3, n_pairs * h * w / 1e6 / dt, n_pairs / dt), flush=True)

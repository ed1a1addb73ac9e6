#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into profiles/pmc_traffic.json.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE reads
exactly for 16-byte-per-lane streaming stores.  Infinity-Cache hits are counted too (these are L2 fabric-side
requests), so "traffic" is an upper bound of the HBM bytes.

usage: pmc_traffic.py <fetch_counter_collection.csv | run_results.db> <write ... .csv | .db> <key> [kernel substring]
"""
import csv
import json
import os
import sys


def per_dispatch(path, counter, kernel):
    vals = []
    if path.endswith(".db"):  # rocpd database (the default output of ROCm 7.2's rocprofv3)
        import sqlite3
        q = "select value from counters_collection where counter_name = ? and kernel_name like ?"
        return [float(v[0]) for v in sqlite3.connect(path).execute(q, (counter, "%" + kernel + "%"))]
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and kernel in r.get("Kernel_Name", ""):
            vals.append(float(r["Counter_Value"]))
    return vals


def main():
    fetch_csv, write_csv, key = sys.argv[1:4]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "k_sor_"  # k_sor_exact and k_sor_fused: every solve of the call
    f = per_dispatch(fetch_csv, "FETCH_SIZE", kernel)
    w = per_dispatch(write_csv, "WRITE_SIZE", kernel)
    out = {"kernel": kernel, "launches_sampled": [len(f), len(w)],
           "fetch_kib_raw_avg": sum(f) / len(f), "write_kib_avg": sum(w) / len(w),
           "bytes_per_launch": int((2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024),
           "note": "FETCH_SIZE doubled (gfx950 half-count of wide coalesced reads); includes Infinity-Cache hits"}
    # per LEVEL: the solver launches of a call differ in grid size by pyramid level (one launch per solve at these sizes), so
    # the averages per (kernel, grid), largest grid first, are the levels finest first -- bench.py's roofline.by_level[*].traffic
    if fetch_csv.endswith(".db") and write_csv.endswith(".db"):
        import sqlite3
        q = ("select kernel_name, grid_size, avg(value), count(*) from counters_collection where counter_name = ? and "
             "kernel_name like ? group by kernel_name, grid_size")
        fg = {(r[0], r[1]): (r[2], r[3]) for r in sqlite3.connect(fetch_csv).execute(q, ("FETCH_SIZE", "%" + kernel + "%"))}
        wg = {(r[0], r[1]): (r[2], r[3]) for r in sqlite3.connect(write_csv).execute(q, ("WRITE_SIZE", "%" + kernel + "%"))}
        short = lambda n: n.replace("papof::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out["by_grid_finest_first"] = [
            {"kernel": short(k[0]), "grid_threads": k[1], "launches_sampled": fg[k][1],
             "bytes_per_launch": int((2.0 * fg[k][0] + wg[k][0]) * 1024)}
            for k in sorted(fg, key=lambda k: -k[1]) if k in wg]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = out["bytes_per_launch"]
    data[key + "_detail"] = out
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

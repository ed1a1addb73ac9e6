#!/bin/bash
# same-box A/B of one environment switch on the default bench line: tools/ab_env.sh <tag> <VAR=value> [rounds] [bench args...]
# alternates "switch unset" / "switch set" `rounds` times (default 3) and prints ms_per_step of every run
tag=$1; sw=$2; rounds=${3:-3}; shift 3 || shift $#
cd "$(dirname "$0")/.."
for r in $(seq 1 $rounds); do
  python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-collection "$@" > gpurun_out/${tag}_base_$r.json 2> gpurun_out/${tag}_base_$r.err || exit 1
  env $sw python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-collection "$@" > gpurun_out/${tag}_sw_$r.json 2> gpurun_out/${tag}_sw_$r.err || exit 1
done
python3 - <<PY
import json, glob
for kind in ("base", "sw"):
    v = [json.load(open(f))["ms_per_step"] for f in sorted(glob.glob("gpurun_out/${tag}_%s_*.json" % kind))]
    print(kind, "${sw}" if kind == "sw" else "(unset)", v, "mean %.4f" % (sum(v) / len(v)))
PY

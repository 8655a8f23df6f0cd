#!/bin/bash
# One GPU-box call: a list of test files, then a same-box A/B of bench.py with one environment switch.
#   tools/gpu_ab.sh <tag> "<pytest files>" <ENVVAR=value for the B run> [bench args]
# Results under gpurun_out/<tag>/ (tests.log, bench_a.json = default, bench_b.json = with the switch).
tag=$1; files=$2; sw=$3; shift 3
out=gpurun_out/$tag
mkdir -p $out
if [ -n "$files" ]; then
  python -m pytest $files -x -q -m gpu > $out/tests.log 2>&1; echo rc=$? >> $out/tests.log; tail -30 $out/tests.log
fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager-steps 0 "$@" > $out/bench_a.json 2> $out/bench_a.err
if [ -n "$sw" ]; then env $sw python bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager-steps 0 "$@" > $out/bench_b.json 2> $out/bench_b.err; fi
python - <<PY
import json, os
for f in ("bench_a", "bench_b"):
    p = "$out/" + f + ".json"
    if os.path.exists(p) and os.path.getsize(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(f, "$sw" if f == "bench_b" else "default", round(d["ms_per_step"], 3), "ms/step  loss", d["loss"])
PY

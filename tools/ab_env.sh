#!/bin/bash
# A/B timing of ONE library under different settings of the engine's measurement knobs, alternating on one box.
# usage: tools/ab_env.sh "<bench.py args>" ROUNDS "name:VAR=val VAR2=val" ["name2:..."]
args="$1"; rounds="$2"; shift 2
for r in $(seq 1 "$rounds"); do
  for nv in "$@"; do
    name="${nv%%:*}"; envs="${nv#*:}"
    out=$(env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-mode $args 2>/dev/null | tail -1)
    echo "$name round $r: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("kernel_ms=%.3f ms_per_step=%.3f plan_ms=%.3f value=%.4g subst=%.6f phred=%.4f" % (d["roofline"]["kernel_ms"], d["ms_per_step"], d["plan_ms_per_step"], d["value"], d.get("substitution_rate", 0), d.get("mean_phred", 0)))' 2>&1)"
  done
done

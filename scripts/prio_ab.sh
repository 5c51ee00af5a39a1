#!/bin/bash
# A/B of the pair kernel without / with progress-based wave priority (EVPK_PRIO = 0, 1, 2), same box, alternating
out=gpurun_out/${1:-prio}
mkdir -p $out
run() {
  local label=$1; shift
  env "$@" python3 bench.py --steps 5 --warmup 2 --cpu-subcycles 0 --no-variants $EXTRA > $out/$label.json 2> $out/$label.err
  python3 - "$out/$label.json" "$label" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:24s} ms_per_step {d['ms_per_step']:.3f} loop {r['loop_ms_per_step']:.3f} x{r['subcycles_per_launch']} avg_launch_ms {r['avg_launch_ms']:.4f} frac {r['frac']:.3f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2 3; do
  for p in ${PRIOS:-0 1 2}; do run prio${p}_$rep EVPK_PRIO=$p; done
done
EXTRA="--ns open"
for p in ${PRIOS:-0 1 2}; do run open_prio$p EVPK_PRIO=$p; done
for p in ${PRIOS:-0 1 2}; do
EVPK_PRIO=$p EVPK_DEBUG_CLOCKS=$out/tl_prio$p.txt python3 bench.py --ns open --steps 2 --warmup 2 --cpu-subcycles 0 --no-variants > /dev/null 2>&1
python3 scripts/k_timeline.py $out/tl_prio$p.txt | head -5
done

#!/bin/bash
# A/B of the three-subcycle pipeline kernel (k_subcycle3w) against the pair kernel on the bench workload, open N-S
export EVPK_LIB=${EVPK_LIB:-cice5_amd/libevpk_exp.so}      # k_subcycle3w lives in the experimental build only (make -C cice5_amd/csrc exp)
out=gpurun_out/${1:-k3}
mkdir -p $out
run() { # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --ns open --steps 4 --warmup 2 --cpu-subcycles 0 --no-variants > $out/$label.json 2> $out/$label.err
  python3 - "$out/$label.json" "$label" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"{sys.argv[2]:28s} ms_per_step {d['ms_per_step']:.3f} loop {r['loop_ms_per_step']:.3f} kernel {r['kernel'][:20]} x{r['subcycles_per_launch']} "
          f"avg_launch_ms {r['avg_launch_ms']:.4f} launches {r['launches']} frac {r['frac']:.3f} others {[(o['subcycles_per_launch'], o['launches'], round(o['avg_launch_ms'],4)) for o in r['other_kernels']]}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run pair EVPK_TRIPLE=0
for R in ${K3_ROWS:-16 24 32 48}; do run triple_R$R EVPK_TRIPLE=1 EVPK_STRIP_ROWS3=$R; done

#!/bin/bash
# Round 5, last GPU pass: smoke(), the bench line as the driver runs it, more seeded draws on the final (scratch-free) library
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_last}; out=gpurun_out/$tag; mkdir -p "$out"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i "smoke" | tee "$out/smoke.txt"
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python3 -c "
import json
o=json.loads([l for l in open('$out/bench_default.json') if l.startswith('{')][-1]); r=o['roofline']
print('bench: value %.4e  ms/evp %.3f  frac %.3f  frac_survey_8d %.3f  launch %.4f ms (rocprof %s)  traffic %s B (%s GB/s, %s of peak, x%s of compulsory)  verify %s  cpu %.3e on %d cores' % (o['value'], o['ms_per_step'], r['frac'], r['frac_survey_8d'], r['avg_launch_ms'], r['rocprof_avg_launch_ms'], r['traffic'], r['traffic_GBps'], r['traffic_frac_of_peak'], r['traffic_over_alg'], o['verify']['bit_identical'], o['cpu_baseline']['value'], o['cpu_baseline']['cores']))" | tee "$out/bench_default.txt"
{
echo "more seeded draws on the final library (no kernel on the default evp path uses scratch memory), pytest summary lines:"
EVPK_FUZZ_N=4000 EVPK_FUZZ_BASE=610000 EVPK_FUZZ_R_N=0 EVPK_FUZZ_E_N=0 timeout 1500 python3 -m pytest tests/test_fuzz_gpu.py -q -k random_configuration 2>&1 | tee "$out/fuzz_full.txt" | grep -E "passed|failed|FAILED" | tail -4
} > "$out/fuzz.txt" 2>&1
cat "$out/fuzz.txt"

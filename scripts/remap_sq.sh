#!/bin/bash
# where k_remap_fluxupd / k_remap_construct spend their time: the call with 12, 4 and 2 tracers (geometry + mass share), and the SQ
# counters of the kernels at the bench size.  usage: scripts/remap_sq.sh [out-dir under gpurun_out]
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-remap_sq}
mkdir -p $out
export TMPDIR=/tmp
for trcr in "0,1,1,1,1,2,1,1,1,1" "0,1" ""; do
  python3 scripts/remap_bench.py --cpu-grid 0 --reps 3 --trcr "$trcr" 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.readlines() if x.startswith('{')]
o=json.loads(l[-1]); print('trcr=[$trcr]', {k:o[k] for k in o if k in ('ms_per_call','value','ntrace','fields')})" >> $out/ntrace.txt
done
cat $out/ntrace.txt
B="python3 scripts/remap_bench.py --cpu-grid 0 --reps 1"
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"
n=1
for P in "$P1" "$P2"; do
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d $out/pmc_$n -o sq -- $B > $out/pmc_$n.log 2>&1
  python3 scripts/pmc_sq.py $out/pmc_$n k_remap >> $out/sq_remap.txt
  rm -rf $out/pmc_$n
  n=$((n+1))
done
cat $out/sq_remap.txt

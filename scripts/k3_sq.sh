#!/bin/bash
# SQ counters of the pair kernel and of the three-subcycle pipeline kernel on the bench workload (open N-S), two passes of <= 8 counters
export EVPK_LIB=${EVPK_LIB:-cice5_amd/libevpk_exp.so}      # k_subcycle3w lives in the experimental build only (make -C cice5_amd/csrc exp)
out=gpurun_out/${1:-k3_sq}
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --ns open --steps 2 --warmup 1 --cpu-subcycles 0 --no-variants"
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"
for mode in pair triple; do
  if [ $mode = pair ]; then export EVPK_TRIPLE=0; else export EVPK_TRIPLE=1; export EVPK_STRIP_ROWS3=${K3_R:-24}; fi
  n=1
  for P in "$P1" "$P2"; do
    timeout 300 rocprofv3 --pmc $P --kernel-trace -d $out/pmc_${mode}_$n -o sq -- $B > $out/pmc_${mode}_$n.log 2>&1
    python3 scripts/pmc_sq.py $out/pmc_${mode}_$n k_subcycle >> $out/sq_${mode}.txt
    rm -rf $out/pmc_${mode}_$n
    n=$((n+1))
  done
done
tail -n 60 $out/sq_pair.txt $out/sq_triple.txt

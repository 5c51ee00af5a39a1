#!/bin/bash
# single-GPU cost of the x-slab machinery (pack / self-copy / unpack, split launches) at several ghost-zone depths
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# EVPK_FORCE_EXCHANGE=1: device copies stand in for the messages; =2: a one-rank RCCL communicator (ncclSend/ncclRecv to itself)
for fe in 1 2; do
for m in 0 1 2 4; do
  [ $m = 0 ] && [ $fe = 2 ] && continue
  if [ $m = 0 ]; then env="EVPK_FORCE_EXCHANGE=0"; else env="EVPK_FORCE_EXCHANGE=$fe EVPK_ZONE_M=$m"; fi
  for ov in 1 0; do
    [ $m = 0 ] && [ $ov = 0 ] && continue
    echo "== transport=$([ $fe = 2 ] && echo rccl-self || echo copies) m=$m overlap=$ov"
    env $env EVPK_OVERLAP=$ov python bench.py --ns open --steps 5 --warmup 2 --cpu-subcycles 0 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); c=j['config']
print(round(j['ms_per_step'],3),'ms  loop',round(j['roofline']['loop_ms_per_step'],3),'k2',round(j['roofline']['avg_launch_ms'],4),'zone',c['ghost_zone_cols'],c['zone_exchanges_per_evp'],c['zone_bytes_sent_rank0'])"
  done
done
done

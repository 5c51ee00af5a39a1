#!/bin/bash
# the multi-rank path with 2 / 4 rank processes sharing ONE GPU, peer-mapped transport against the host-staged relay:
# a functional check and the per-exchange cost of the transport, not a scaling number (the ranks share the chip)
cd "$(dirname "$0")/.." || exit 1
export EVPK_FORCE_DEVICE=0
for ns in open tripole; do
for n in 2 3 4; do
for xp in ipc shm; do
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus $n --steps 3 --warmup 2 --cpu-subcycles 0 --transport $xp --ns $ns 2>/dev/null | python3 -c "
import json,sys
l=[x for x in sys.stdin.readlines() if x.startswith('{')]
if not l: print('$ns $n ranks $xp: no result'); sys.exit(0)
o=json.loads(l[-1]); c=o['config']
r=o['roofline']
print('%-8s %d ranks %-4s ms/evp=%.3f loop=%.3f (kernels %.2f + exchanges %.2f of %d) transport=%s zone_cols=%d exchanges/evp=%d+%d value=%.3e'%('$ns',$n,'$xp',o['ms_per_step'],r['loop_ms_per_step'],r['kernel_ms_per_step_all_kinds'],r.get('bound_ms_per_step',0),r.get('bound_updates_per_step',0),c['transport'],c['ghost_zone_cols'],c['zone_exchanges_per_evp'],c.get('band_row_exchanges_per_evp',0),o['value'] or 0))"
done; done; done

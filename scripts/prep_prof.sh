#!/bin/bash
# per-kernel time of the once-per-evp kernels (prep / finish) of the bench workload: rocprofv3 --kernel-trace --stats, steady-state evps
# only (the first, `fresh`, evp of a context touches every cell and is dropped via min/avg), then scripts/prep_roofline.py
out=gpurun_out/${1:-prep}
mkdir -p $out
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 6 --warmup 2 --cpu-subcycles 0 --no-variants ${PREP_FLAGS} > $out/trace.log 2>&1
cp "$(ls $out/trace/*/*kernel_trace.csv $out/trace/*kernel_trace.csv 2>/dev/null | head -1)" $out/kernel_trace.csv
rm -rf $out/trace
python3 scripts/prep_roofline.py $out/kernel_trace.csv ${PREP_FLAGS} | tee $out/prep_roofline.txt

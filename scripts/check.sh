#!/bin/bash
# Everything a change has to survive.  On a box without a GPU: build + CPU suite.  With one: + GPU suite, smoke, a short bench.
cd "$(dirname "$0")/.." || exit 1
set -e
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null
python3 -m pytest tests -x -q -m "not gpu"
if python3 -c "import ctypes,sys; h=ctypes.CDLL('libamdhip64.so'); n=ctypes.c_int(0); sys.exit(0 if h.hipGetDeviceCount(ctypes.byref(n))==0 and n.value>0 else 1)" 2>/dev/null; then
  python3 -m pytest tests -x -q -m gpu
  python3 -c "import __graft_entry__ as g; g.smoke()"
  python3 bench.py --steps 3 --warmup 1 --cpu-subcycles 0 | tail -1 | cut -c1-300
fi

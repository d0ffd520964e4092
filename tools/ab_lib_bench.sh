#!/bin/bash
# same-box A/B in the bench loop of the in-tree library against a build of ONE source with other -D flags (linked with the in-tree objects of the rest):
# usage: tools/ab_lib_bench.sh <source.hip> "<-D flags of build B>" "<extra env of B>" [rounds]
#   e.g. tools/ab_lib_bench.sh pv_attn.hip "-DPV_ATTN8_MAX3=0 -DPV_ATTN8_LOOP_PAD=3" "PV_GEMM_BIG128=0" 3
cd "$(dirname "$0")/.."
python - "$1" $2 <<'PY'
import os, subprocess, sys
import photoverse_amd.build as b
src, flags = sys.argv[1], sys.argv[2:]
o = "/tmp/ab_lib_alt.o"
subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *flags, "-c", os.path.join(b.CSRC, src), "-o", o])
objs = [o if f == src else os.path.join(b.LIBDIR, f.replace(".hip", ".o")) for f in b.SOURCES]
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libpv_alt.so", *objs])
PY
for i in $(seq 1 ${4:-3}); do
  for v in "" "PV_HIP_LIB=/tmp/libpv_alt.so $3"; do
    r=$(env $v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-train-forward 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i  [${v:-in-tree library}]  steps/s, ms/step: $r"
  done
done

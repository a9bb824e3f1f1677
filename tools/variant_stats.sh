#!/bin/bash
# Static screening of a tuning variant without a GPU: registers / scratch / LDS / occupancy of the dominant kernel and its
# static VALU issue cost (tools/isa_cost.py).   bash tools/variant_stats.sh NAME "-DSCSFM_X=1 ..." [NAME2 "flags2" ...]
cd "$(dirname "$0")/.."
mkdir -p /tmp/isa
K=${KERNEL:-pair_fwd_spec_kernelIfLb1ELj7ELb0ELb0E}
while [ $# -ge 2 ]; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize -Wno-unused-function \
      -Iinclude $2 -S --cuda-device-only -o /tmp/isa/$1.s sc-sfmlearner-release_amd/csrc/scsfm_pair.hip 2>/dev/null
    python - "$1" "$K" <<'PY'
import re, sys
sys.path.insert(0, "tools")
from isa_cost import weight
name, k = sys.argv[1], sys.argv[2]
lines = open(f"/tmp/isa/{name}.s").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN5scsfm20" + k) or (k in l and re.match(r"^_Z\w+:", l)))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
meta = {}
for l in lines[end:end + 400]:
    m = re.match(r";\s*(NumVgprs|ScratchSize|LDSByteSize|Occupancy|NumSgprs):\s*(\d+)", l)
    if m and m.group(1) not in meta:
        meta[m.group(1)] = int(m.group(2))
body = [l.strip().split()[0] for l in lines[start:end] if l.startswith("\t") and not l.startswith("\t.") and not l.strip().startswith(";")]
valu = [o for o in body if o.startswith("v_")]
cost = sum(weight(o) for o in valu)
print(f"{name:14s} vgpr={meta.get('NumVgprs')} sgpr={meta.get('NumSgprs')} scratch={meta.get('ScratchSize')} lds={meta.get('LDSByteSize')} occ={meta.get('Occupancy')} "
      f"valu={len(valu)} cost={cost:.0f} ds={sum(o.startswith('ds_') for o in body)} vmem={sum(o.startswith(('buffer_','global_','flat_')) for o in body)} barrier={sum(o=='s_barrier' for o in body)}")
PY
  ) &
  shift 2
done
wait

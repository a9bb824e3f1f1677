#!/bin/bash
# GPU session: a list of variants/*.so through tools/variant_check.py (kernel-only time + result checksums), in the
# order given (repeat names to alternate).   gpurun -- 'bash tools/gpu_variants.sh TAG a b a b'
set +e
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
TAG=${1:-v}; shift
: > $O/variants_$TAG.jsonl
for v in "$@"; do
  if [ "${v%time}" != "$v" ]; then   # a -DPROBE_TIMING build: the stage timeline
    echo "=== $v (timeline)"; SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/march_timing.py 2>&1 | tail -n 1 | tee -a $O/variants_$TAG.jsonl
  else
    echo "=== $v"; SCSFM_HIP_LIB=$R/variants/$v.so timeout 300 python tools/variant_check.py 2>&1 | tail -n 1 | tee -a $O/variants_$TAG.jsonl
  fi
done

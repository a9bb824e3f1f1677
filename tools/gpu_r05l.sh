set +e
export TMPDIR=/tmp
for v in ${VARIANTS:-fbase fb4g1 fb3g4 fb4g4 fbase fb4g1 fb3g4 fb4g4}; do
  echo "=== $v"; VARIANT_EXTRA=1 SCSFM_HIP_LIB=$PWD/variants/$v.so timeout 300 python tools/variant_check.py --depths smooth 2>&1 | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], 'plain fwd', d['us_plain_fwd'], 'spec', d['us'])"
done

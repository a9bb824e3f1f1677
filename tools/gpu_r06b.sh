# round 6, session b: the texel variant with one / two tail pixels in flight against the base, three depth laws; the smooth
# forward with and without the texel pack riding along
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/variants_r06b.jsonl; : > $O
for v in r6base r6rgbd1 r6rgbd1t2 r6base r6rgbd1 r6rgbd1t2; do
  echo "=== $v"; SCSFM_HIP_LIB=$PWD/variants/$v.so timeout 400 python tools/variant_check.py --depths smooth,scene,iid 2>&1 | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); d.pop('check',None); print(json.dumps(d))" | tee -a $O
done

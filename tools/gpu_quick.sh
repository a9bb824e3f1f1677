set +e
export TMPDIR=/tmp
MIOPEN_FIND_MODE=FAST timeout 900 python tools/e2e_probe.py 2>&1 | grep -v Warning | tail -4

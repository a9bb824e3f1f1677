set +e
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40

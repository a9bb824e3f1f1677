set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trainer.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -40 | cut -c1-400

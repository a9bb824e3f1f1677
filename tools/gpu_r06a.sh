# round 6, session a: RGBD-texel variants of the dominant kernel A/B (kernel alone, alternating processes), then the GPU
# suite and the default bench line of the tree as it stands
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_variants.sh r06a r6base r6rgbd1 r6rgbd2 r6base r6rgbd1 r6rgbd2
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 > gpurun_out/r06a_pytest.txt
tail -n 3 gpurun_out/r06a_pytest.txt
timeout 600 python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err
tail -c 600 gpurun_out/r06a_bench.json

# round 6, session h: the full record of the tree as it stands (tools/gpu_round.sh r06: smoke, GPU suite, bench on the four
# configurations and the three depth laws, rocprofv3 kernel trace, FETCH_SIZE / WRITE_SIZE and SQ counter passes), then the
# counters' calibration on known-bytes kernels
set +e
export TMPDIR=/tmp
R=$PWD
bash tools/gpu_round.sh r06
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof_calib -o calib_$C -- $R/tools/ubench/fetch_calib > $R/gpurun_out/calib_$C.log 2>&1; echo "calib $C rc=$?"
done
cd $R
python tools/pmc_calib_summary.py gpurun_out/prof_calib 1073741824 --json gpurun_out/r06_fetch_calibration.json | tail -8

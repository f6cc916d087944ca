# The open item of DESIGN.md section 13: the whole GPU suite in ONE process with the sub-network forks ON (tests/conftest.py keeps
# them off otherwise), to reproduce the hipGraphLaunch segfault met ~300 tests into the process and get its faulting frame.
# pytest's faulthandler prints the Python stack of the faulting thread; AMD_LOG_LEVEL=1 adds the runtime's own error lines.
# Writes gpurun_out/suite_forks_on.log (tail it: a progress line per test file keeps the call alive).  ~10-15 GPU-minutes.
mkdir -p gpurun_out
export TD_TEST_FORKS=1 AMD_LOG_LEVEL=${AMD_LOG_LEVEL:-1} PYTHONFAULTHANDLER=1
timeout -k 10 ${SUITE_TIMEOUT:-1100} python -u -X faulthandler -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/suite_forks_on.log 2>&1
rc=$?
echo "rc=$rc" >> gpurun_out/suite_forks_on.log
grep -n "Fatal Python error\|Segmentation\|File \"" gpurun_out/suite_forks_on.log | head -40
tail -5 gpurun_out/suite_forks_on.log
exit 0

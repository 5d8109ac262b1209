#!/bin/bash
# Late hipGraph capture reproducer under several hypotheses; each variant is a fresh process.  A variant that is killed by its
# timeout stops the script (no further GPU step after a hang); an ordinary failure / segfault is recorded and the next variant runs.
mkdir -p gpurun_out
FILES="tests/test_fullsize_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_zz_late_graph_capture.py"
run() {
  name=$1; shift
  echo "=== $name" | tee -a gpurun_out/graph_repro.log
  env FA_ATT=pool "$@" timeout -k 10 420 python -X faulthandler -m pytest $FILES -m gpu -x -q -k "not fullsize_backward and not trained_weights_train_step" > gpurun_out/graph_repro_$name.log 2>&1
  rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/graph_repro.log
  tail -5 gpurun_out/graph_repro_$name.log >> gpurun_out/graph_repro.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping" | tee -a gpurun_out/graph_repro.log; exit 1; fi
}
run fresh_event TCCT_WGRAD_FRESH_EVENT=1
run thread_local TCCT_GRAPH_ERRMODE=thread_local
run shared_pool TCCT_GRAPH_SHARED_POOL=1
run baseline TCCT_NOP=1
exit 0

#!/bin/bash
# run on the GPU box: the world-1 RCCL step-graph test in a loop under rocgdb until it aborts; native backtraces of every thread
# -> gpurun_out/<tag>/bt_<i>.txt   (one run in ~7 died with SIGABRT from a thread without a Python frame and no message)
TAG=${1:-hunt}; N=${2:-10}
mkdir -p gpurun_out/$TAG
export NCCL_DEBUG=WARN TORCH_CPP_LOG_LEVEL=INFO TORCH_SHOW_CPP_STACKTRACES=1 AMD_LOG_LEVEL=1
cat /proc/sys/kernel/core_pattern > gpurun_out/$TAG/core_pattern.txt
for i in $(seq 1 $N); do
  timeout 240 /opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "handle SIGABRT stop print" -ex "handle SIG32 SIG33 SIG34 SIG35 nostop noprint pass" \
    -ex run -ex "echo \n==== after run ====\n" -ex "info threads" -ex "thread apply all bt 40" \
    --args python -m pytest tests/test_gpu_trainer.py -q -x -k data_parallel > gpurun_out/$TAG/gdb_$i.log 2>&1
  rc=$?
  if grep -q "SIGABRT" gpurun_out/$TAG/gdb_$i.log; then
    echo "run $i: SIGABRT caught (rc=$rc)"; cp gpurun_out/$TAG/gdb_$i.log gpurun_out/$TAG/bt_$i.txt; break
  fi
  echo "run $i: rc=$rc $(grep -c passed gpurun_out/$TAG/gdb_$i.log) passed-lines"
  tail -c 3000 gpurun_out/$TAG/gdb_$i.log > gpurun_out/$TAG/gdb_$i.tail; rm gpurun_out/$TAG/gdb_$i.log
done

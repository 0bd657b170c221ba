#!/bin/bash
# runs tools/lab/ipc_probe.bin with 2 and 3 processes on the one GPU; logs under gpurun_out/ipc_probe/
set -u
out=gpurun_out/ipc_probe
mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
rc=0
for cfg in "2 0" "2 1" "2 2" "3 0"; do
  set -- $cfg
  n=$1; kind=$2
  d=$(mktemp -d /tmp/ipcprobe.XXXXXX)
  pids=()
  for ((r=0; r<n; r++)); do
    timeout -k 5 60 tools/lab/ipc_probe.bin $d $r $n $kind > $out/n${n}_k${kind}_r$r.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p || rc=1; done
  echo "== n=$n kind=$kind"; cat $out/n${n}_k${kind}_r*.log
  rm -rf $d
done
exit $rc

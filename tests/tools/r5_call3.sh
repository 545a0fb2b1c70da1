#!/bin/bash
# round-5 GPU call 3: v_sqrt_f32 accuracy, MALL write-then-read probe, device-vs-oracle transform accuracy, and the
# reproduction of the memory fault filed in profiles/r04_stft8192_ablation.txt (round-4 ablation builds a64 / a60)
R=$PWD; O=$R/gpurun_out/r5c3; rm -rf $O; mkdir -p $O
timeout 120 tests/tools/probes/sqrt_probe > $O/sqrt_probe.txt 2>&1; cat $O/sqrt_probe.txt
timeout 300 tests/tools/probes/mall_probe > $O/mall_probe.txt 2>&1; tail -70 $O/mall_probe.txt
timeout 600 python tests/tools/fft_accuracy.py > $O/fft_accuracy.json 2> $O/fft_accuracy.err; echo "fft_accuracy rc=$?"; python -c "
import json; d=json.load(open('$O/fft_accuracy.json')); print({k:v for k,v in d.items() if k.endswith('_summary')})"; tail -3 $O/fft_accuracy.err
for tag in r4a64 r4a60 r4a64 r4a60 r4a64 r4a60; do
  KBENCH_SERIAL=1 timeout 90 tests/tools/kbench bliss-rs_amd/libblissgpu_$tag.so 512 180 2 > $O/fault_$tag.out 2> $O/fault_$tag.err; rc=$?
  echo "$tag rc=$rc $(cut -c1-200 $O/fault_$tag.out | head -1)"; head -3 $O/fault_$tag.err
  if [ $rc -ne 0 ]; then
    AMD_LOG_LEVEL=3 HIP_LAUNCH_BLOCKING=1 KBENCH_SERIAL=1 timeout 120 tests/tools/kbench bliss-rs_amd/libblissgpu_$tag.so 512 180 2 > $O/faultlog_$tag.out 2> $O/faultlog_$tag.err
    echo "  with blocking launches: rc=$?"; grep -a "ShaderName" $O/faultlog_$tag.err | tail -4 | cut -c1-300; grep -a "fault" $O/faultlog_$tag.err | head -2
    rm -f $O/faultlog_$tag.err.full
    tail -c 20000 $O/faultlog_$tag.err > $O/faultlog_$tag.tail; rm $O/faultlog_$tag.err
  fi
done

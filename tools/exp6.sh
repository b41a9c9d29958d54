#!/bin/bash
set -u
for i in 1 2; do
NEEDLE_HIP_LIBRARY=$PWD/needle_amd/lib/variants/libneedle_hip_stamp.so NEEDLE_HIP_DEBUG=64 timeout -k 10 200 python3 bench.py --precision bf16 --steps 20 --warmup 5 --cpu-seconds 0 --secondary-steps 0 2>gpurun_out/exp6.err | cut -c1-10; grep "in-kernel" gpurun_out/exp6.err | tail -3
NEEDLE_HIP_LIBRARY=$PWD/needle_amd/lib/variants/libneedle_hip_stamp.so NEEDLE_HIP_DEBUG=65 timeout -k 10 200 python3 bench.py --precision bf16 --steps 20 --warmup 5 --cpu-seconds 0 --secondary-steps 0 2>gpurun_out/exp6.err | cut -c1-10; grep "in-kernel" gpurun_out/exp6.err | tail -2
done

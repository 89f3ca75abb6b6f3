#!/bin/bash
# round 6, second switch matrix: the GPU suite under the ResNet routing switches added in the round's second half
mkdir -p gpurun_out
{
for sw in "PCLIP_CONV_STRIP=0" "PCLIP_CONV_STEM=0" "PCLIP_CONV_POOL=0"; do
  echo "== $sw"; env $sw timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED" | tail -3
done
} > gpurun_out/r06_switch_matrix_d.txt 2>&1
cat gpurun_out/r06_switch_matrix_d.txt

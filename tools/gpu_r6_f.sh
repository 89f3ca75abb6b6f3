#!/bin/bash
# round 6: the two switch legs that failed the first matrix, after their fixes; the RN chunk probe
mkdir -p gpurun_out
{
echo "== default: fused / ties / full size"; python -m pytest tests/test_gpu_parity.py -q -k "fused or exact_ties or full_size" 2>&1 | tail -3
echo "== PCLIP_CLASSIFY_PANEL_EXACT=1"; PCLIP_CLASSIFY_PANEL_EXACT=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_e2e.py -q 2>&1 | tail -3
echo "== PCLIP_CLASSIFY_PANEL_PASSES=1"; PCLIP_CLASSIFY_PANEL_PASSES=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_e2e.py -q 2>&1 | tail -3

} > gpurun_out/r06_switch_matrix_b.txt 2>&1
python tools/rn_chunk_probe.py > gpurun_out/r06_rn_chunk_probe.txt 2>&1
tail -30 gpurun_out/r06_switch_matrix_b.txt; cat gpurun_out/r06_rn_chunk_probe.txt | grep -v amdgpu

# C2 latency evidence: the one-launch tests (normal + stress library, both hand-over forms) and the hipGraph-replay probe
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "proto_classify" 2>&1 | tail -3 > gpurun_out/c2_tests.txt
PCLIP_PROTO_CLASSIFY_WT=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -x -q -m gpu -k "proto_classify" 2>&1 | tail -3 >> gpurun_out/c2_tests.txt
cat gpurun_out/c2_tests.txt
python tools/c2_probe.py > gpurun_out/c2_probe.txt 2>&1
cat gpurun_out/c2_probe.txt

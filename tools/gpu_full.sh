cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/full_tests.txt
cat gpurun_out/full_tests.txt
bash tools/c2_probe.sh
timeout 900 python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
tail -c 1500 gpurun_out/bench_c2.json

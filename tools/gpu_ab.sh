# A/B of libpclip.so vs libpclip_old.so on the bench's GEMM shapes + rocprof kernel stats of the bench; argument: tag
TAG=${1:-ab}
mkdir -p gpurun_out
timeout 900 python tools/ab_lib.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_$TAG.log
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
head -12 $R/gpurun_out/prof_$TAG/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
tail -c 600 $R/gpurun_out/prof_$TAG.log

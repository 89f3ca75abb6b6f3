# bench with libpclip.so and with a variant library swapped in (argument: variant tag), alternating
TAG=$1
cp proto-clip_amd/libpclip.so /tmp/base.so
for r in 1 2; do
  for v in base $TAG; do
    if [ $v = base ]; then cp /tmp/base.so proto-clip_amd/libpclip.so; else cp proto-clip_amd/libpclip_$TAG.so proto-clip_amd/libpclip.so; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['gemm_ms_per_step'],2), d['sclk_mhz_under_load'])"
  done
done
cp /tmp/base.so proto-clip_amd/libpclip.so

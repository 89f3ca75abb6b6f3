for r in 1 2; do for m in 0 1; do PCLIP_ATT_PIPE=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipe $m', round(d['value']), round(d['ms_per_step'],3), d['sclk_mhz_under_load'])"; done; done
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from proto_clip_amd import ops, _lib
sys.path.insert(0, 'tools')
from kernel_bench import timeit
lib = _lib.load()
B, L, H = 1024, 197, 12
qkv = torch.randn(B * L, 3 * H * 64, device="cuda").half(); out = torch.empty(B * L, H * 64, device="cuda", dtype=torch.float16)
for mode in (0, 1, 0, 1):
    lib.pclip_attention_config(mode, 0)
    t = timeit(lambda: ops.attention(qkv, B, L, H, False, out), iters=20)
    print("attention mode", mode, round(t * 1e6, 1), "us")
PY

# round-3 GPU pass A: clock / power measurements + baseline tests + bench with telemetry
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/smi_discovery.txt 2>&1
import sys, time
sys.path.insert(0, ".")
from proto_clip_amd import telemetry
src = telemetry.open_source()
print("telemetry source:", src.name if src else None)
if src:
    t0 = time.perf_counter(); r = [src.read() for _ in range(20)]; print("20 reads in %.3f s" % (time.perf_counter() - t0), r[:3])
try:
    import amdsmi
    amdsmi.amdsmi_init(); h = amdsmi.amdsmi_get_processor_handles()[0]
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print({k: m[k] for k in m if "clk" in k or "power" in k or "throttle" in k or "temperature_hotspot" in k})
    print(amdsmi.amdsmi_get_power_info(h)); print(amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
except Exception as e:
    print("amdsmi failed:", repr(e))
PY
head -c 3000 gpurun_out/smi_discovery.txt; echo
timeout 300 python tools/clock_probe.py > gpurun_out/clock_probe.txt 2>&1; cat gpurun_out/clock_probe.txt | tail -12
timeout 600 python tools/power_trace.py 200 > gpurun_out/power_trace.txt 2> gpurun_out/power_trace.err; cat gpurun_out/power_trace.txt; tail -3 gpurun_out/power_trace.err
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 --tb=short 2>&1 | grep -vE "^E   +(\+|where)" > gpurun_out/pytest_gpu_r3a.log; grep -v "of the bound" gpurun_out/pytest_gpu_r3a.log | tail -15
cp gpurun_out/observed_tolerances.json gpurun_out/observed_tolerances_r3a.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err; cat gpurun_out/bench_r3a.json; tail -3 gpurun_out/bench_r3a.err

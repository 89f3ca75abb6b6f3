#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): query images/sec on the ImageNet 16-shot ViT-B/16 configuration
(configs[2]: full encoder + conv-3x adapter + dual-bank classification), synthetic data, random-init
weights of the named architecture.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of B=1024 pre-processed query images per GPU
(the reference's loader batch, main.py:505): prototype reduction from the (rank-sharded) 16 000-row
support bank [+ RCCL all-gather of the fp32 class sums when N>1], fp32->fp16 image cast, ViT-B/16
encode_image, row normalise, conv-3x adapter + normalise, distance GEMM against both 1000-class banks,
softmax fusion + argmax.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

N_CLASS, SHOTS, DIM, BATCH = 1000, 16, 512, 1024
ALPHA, BETA = 0.5, 12.0                       # configs/imagenet.yml:14-15
GFLOP_PER_IMG_ENCODER = 35.13                 # SURVEY §6 (torch flop counter on the REFERENCE module: every token through all 12 blocks)
GFLOP_PER_IMG_TAIL = 0.00452                  # conv-3x adapter 2.47 MFLOP + both similarity contractions 2.05 MFLOP
# attention contractions the HIP tower executes per image: 11 blocks x 4 H L^2 dh + the last block's one-query attention 4 H L dh (H = 12, L = 197, dh = 64)
GFLOP_PER_IMG_ATTENTION = (11 * 4 * 12 * 197 * 197 * 64 + 4 * 12 * 197 * 64) / 1e9
MFMA_PEAK_TFLOPS = 2500.0                     # fp16 dense, MI355X_MICROARCH.md


def gemm_flops(M, N, K):
    return 2.0 * M * N * K


def build_state(device, rank, world):
    from proto_clip_amd import ops, synth
    from proto_clip_amd.clip.model import BACKBONES, build_model, random_state_dict
    from proto_clip_amd.dist import shard_bounds
    from proto_clip_amd.model import Adapter
    model = build_model(random_state_dict(seed=1, **BACKBONES["ViT-B/16"])).to(device)
    torch.manual_seed(1)
    adapter = Adapter(DIM, "conv-3x", dtype=torch.half)
    with torch.no_grad():
        for n_, p_ in adapter.named_parameters():
            if "bn" in n_:
                p_.add_((torch.randn(p_.shape) * 0.1).half())
    adapter = adapter.to(device)
    split = synth.make_split(N_CLASS, SHOTS, DIM, 8, 8, seed=1)
    bank_rows = split.visual_memory_keys.t().contiguous()                 # [16000, 512] fp16, sorted by class
    labels = torch.arange(N_CLASS).repeat_interleave(SHOTS).int()
    lo, hi = shard_bounds(N_CLASS * SHOTS, rank, world)                   # this rank's slab of the support set
    st = dict(model=model, adapter=adapter, bank=bank_rows[lo:hi].to(device), bank_labels=labels[lo:hi].to(device),
              text=ops.l2norm_rows(split.textual_memory_bank.t().contiguous().to(device)))
    # synthetic pre-processed images, fp32 like the reference's loader output: BATCH DISTINCT images (the GEMMs run at a
    # data-dependent power cap, so a tiled batch is not a neutral stand-in), distinct seed per rank; generated in chunks on a
    # few host threads (numpy releases the GIL), each chunk from its own PRNG stream
    from concurrent.futures import ThreadPoolExecutor
    chunk = 64
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        parts = list(pool.map(lambda c: synth.make_images(chunk, 224, seed=100 + rank, stream=50 + 2 * c, n_class=N_CLASS), range(BATCH // chunk)))
    st["images"] = torch.cat(parts).to(device).contiguous()
    return st


def step(st):
    """proto_clip_amd.dist.hot_path_step: the prototype reduction (main.py:399-402) + all-gather run on a side stream under the
    encoder (clip/model.py:338 -> utils.py:352 -> model.py:49-78 + main.py:408-409); the classification (utils.py:225) joins them."""
    from proto_clip_amd.dist import HipPath, PrototypeExchange, hot_path_step
    if "path" not in st:
        st["path"], st["exchange"] = HipPath(st["model"], st["adapter"]), PrototypeExchange()
    with torch.no_grad():
        return hot_path_step(st["path"], st["exchange"], st["bank"], st["bank_labels"], N_CLASS, st["images"], st["text"], ALPHA, BETA)


def measure_gemm(st):
    """The MFMA-GEMM launches of one step, timed IN the step by the kernels themselves.  Anything put between consecutive GEMMs — a HIP event pair
    (round 5), a one-lane stamp kernel — costs ~33 us per GEMM (31.4 ms of "GEMM time" where rocprofv3 summed the same kernels to 28.9), and replaying a GEMM back
    to back outside the step runs it at ANOTHER clock (the chip is at its power cap: a train of identical GEMMs holds a lower clock than the same GEMMs between
    LayerNorm / attention launches: +10 % on one box, -1 % on another).  So the GEMM kernels carry a measurement hook (`pclip_gemm_timing`, csrc pgemm::time_begin /
    time_end): with a slot pointer, every workgroup folds the device's constant 100 MHz counter into slot[0] (minimum = the launch's first instruction) and slot[1]
    (maximum = its last) — the span `rocprofv3 --kernel-trace` reports as the kernel's duration, with nothing between the kernels.  An instrumented step (outside
    the timed region, behind twelve plain steps so that the clock has settled) runs with the hook on; a call's time = the sum over its launches (a call whose
    last round of tiles is split = two).
    tools/roofline_check.py compares the line with the rocprofv3 summary of the same command (profiles/r06_roofline_check_*.txt)."""
    from proto_clip_amd import ops, _lib
    lib = _lib.load()
    real = ops.gemm
    rec = []
    dev = st["images"].device
    NS = 1024
    buf = torch.zeros(NS, 2, dtype=torch.int64, device=dev)
    buf[:, 0] = torch.iinfo(torch.int64).max                    # begin words: minimum of non-negative 63-bit stamps
    torch.cuda.synchronize()

    def timed(a, w, bias=None, act=0, residual=None, out=None):
        i0 = lib.pclip_gemm_timing_count()
        y = real(a, w, bias, act, residual, out)
        rec.append(((a.shape[0], w.shape[0], a.shape[1], act, bias is not None, residual is not None), i0, lib.pclip_gemm_timing_count()))
        return y

    # The chip's DVFS loop needs ~0.2 s of load to settle (a first step after a pause runs ~10 % slow: tools/gemm4w_stamps.py met the same): twelve plain steps
    # go ahead of the instrumented one in the same train of launches, no synchronisation in between (the hook is a host-side switch)
    for _ in range(12):
        step(st)
    ops.gemm = timed
    n0 = lib.pclip_gemm_kernel_launches()
    _lib.check(lib.pclip_gemm_timing(buf.data_ptr(), NS), "pclip_gemm_timing")
    try:
        step(st)
        torch.cuda.synchronize()
    finally:
        lib.pclip_gemm_timing(None, 0)
        ops.gemm = real
    launches = lib.pclip_gemm_kernel_launches() - n0           # a call whose last round is split = two kernel launches
    t = buf.cpu().numpy()
    W = getattr(st["model"].visual, "width", 0)

    def name_of(M, N, K, act, has_bias, has_res):
        if W and M > 4096:
            if (N, K, act) == (3 * W, W, 0) and not has_res: return "in_proj"
            if (N, K) == (W, W) and has_res: return "out_proj"
            if (N, K, act) == (4 * W, W, 1): return "c_fc"
            if (N, K) == (W, 4 * W) and has_res: return "c_proj"
        return "other"

    per, total_ms, total_fl, timed_launches = {}, 0.0, 0.0, 0
    for key, i0, i1 in rec:
        ms = sum(max(int(t[i][1] - t[i][0]), 0) for i in range(i0, i1)) * 1e-5           # 10 ns ticks -> ms
        timed_launches += i1 - i0
        fl = gemm_flops(*key[:3])
        d = per.setdefault(name_of(*key), dict(calls=0, ms_per_step=0.0, flops_per_step=0.0))
        d["calls"] += 1
        d["ms_per_step"] += ms
        d["flops_per_step"] += fl
        total_ms += ms
        total_fl += fl
    for nm, d in per.items():
        d["avg_call_us"] = 1e3 * d["ms_per_step"] / d["calls"]
        if nm != "other":
            d["tflops"] = d["flops_per_step"] / (d["ms_per_step"] * 1e-3) / 1e12
            d["frac"] = d["tflops"] / MFMA_PEAK_TFLOPS
        d.pop("flops_per_step")
    if "other" in per:
        per["other (patch embedding, class-row tail, projection)"] = per.pop("other")
    return dict(launches=launches, calls=len(rec), total_ms=total_ms, avg_us=1e3 * total_ms / max(launches, 1),
                tflops=total_fl / (total_ms * 1e-3) / 1e12, flops=total_fl, per_variant=per, timed_launches=timed_launches)


def self_check(st, n=64):
    """After the timed loop: the labels the step produces for `n` of its images must be the labels of a step over those images ALONE
    (a row's bits do not depend on the batch around it — the contract the parity tests state at small sizes, checked here on the
    configuration that was just timed).  Every rank runs it (the step contains the all-gather)."""
    idx = torch.arange(0, BATCH, BATCH // n, device=st["images"].device)
    full = step(st)
    sub = step(dict(st, images=st["images"][idx].contiguous()))
    torch.cuda.synchronize()
    ok = full.shape == (BATCH,) and bool(torch.equal(full[idx], sub)) and int(torch.unique(full).numel()) >= 2
    return "ok" if ok else f"MISMATCH: {int((full[idx] != sub).sum())} of {n} labels differ from the sub-batch step"


def c2_kernels(device):
    """BASELINE configs[1] (EuroSAT 16-shot ViT-B/32: prototype build + classification kernels only; SURVEY 8d C2 = 8.35 MB of algorithmic traffic):
    device time per call by hipGraph replay of 20 calls (tools/small_bench.py), for the `extra` block."""
    from proto_clip_amd import ops
    N, K, D, Q = 10, 16, 512, 8100
    g = torch.Generator(device=device).manual_seed(3)
    nrm = torch.nn.functional.normalize
    mem = nrm(torch.randn(N * K, D, device=device, generator=g), dim=-1).half()
    q = nrm(torch.randn(Q, D, device=device, generator=g), dim=-1).half()
    zt = nrm(torch.randn(N, D, device=device, generator=g), dim=-1).half()
    zi = ops.proto_build(mem, N, K)

    def gpu_time(fn, reps=20, iters=20):
        fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (iters * reps)

    t_pb = gpu_time(lambda: ops.proto_build(mem, N, K))
    t_cl = gpu_time(lambda: ops.classify(q, zi, zt, 1.0, 0.7, want_p=False, want_argmax=True))
    byts = Q * D * 2 + 2 * N * D * 2 + Q * 4
    return {"workload": "C2 EuroSAT 16-shot ViT-B/32: N = 10, K = 16, D = 512, Q = 8100 (kernels only, hipGraph replay)",
            "proto_build_us": t_pb * 1e6, "classify_argmax_us": t_cl * 1e6, "classify_algorithmic_bytes": byts,
            "classify_gb_per_s": byts / t_cl / 1e9, "classify_frac_of_hbm_8tb": byts / t_cl / 8e12}


def c3_classify(device):
    """The classification stage at BASELINE configs[2]'s FULL size (Q = 50 000 test queries, N = 1000, D = 512: the fused row-panel kernel of the default routing, which
    the 1024-query step does not reach), device time by hipGraph replay, on class-structured features (SURVEY 8d's generator) and on structureless ones."""
    from proto_clip_amd import ops
    Q, N, K, D = 50000, 1000, 16, 512
    g = torch.Generator(device=device).manual_seed(1)
    nrm = torch.nn.functional.normalize
    cen = torch.randn(N, D, device=device, generator=g)
    y = torch.randint(0, N, (Q,), device=device, generator=g)
    q_s = nrm(cen[y] + 0.8 * torch.randn(Q, D, device=device, generator=g), dim=-1).half()
    zi = ops.proto_build(nrm(cen.repeat_interleave(K, 0) + 0.8 * torch.randn(N * K, D, device=device, generator=g), dim=-1).half(), N, K)
    zt = nrm(cen + 0.5 * torch.randn(N, D, device=device, generator=g), dim=-1).half()
    q_r = nrm(torch.randn(Q, D, device=device, generator=g), dim=-1).half()

    def gpu_time(fn, reps=5, iters=5):
        fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / (iters * reps)

    out = {"workload": "C3 classification stage alone: Q = 50000, N = 1000, D = 512, argmax (default routing = fused row panels)"}
    for tag, q in (("structured", q_s), ("structureless", q_r)):
        ops.classify_panel_stats(reset=True)
        t = gpu_time(lambda: ops.classify(q, zi, zt, ALPHA, BETA, want_p=False, want_argmax=True))
        npan, nsec = ops.classify_panel_stats()
        with ops.classify_two_stage():
            t2 = gpu_time(lambda: ops.classify(q, zi, zt, ALPHA, BETA, want_p=False, want_argmax=True))
        out[tag] = {"default_routing_us": t * 1e6, "two_stage_us": t2 * 1e6, "panels_second_pass_fraction": (nsec / npan) if npan else None,
                    "fused_kernel_ran": bool(npan)}
    return out


PMC_TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")     # newest committed summary first


def pmc_traffic():
    """HBM bytes per GEMM launch from the rocprofv3 PMC passes of this same command (FETCH_SIZE x2 + WRITE_SIZE,
    separate passes, gfx950 correction) — counters cannot be read from inside the process, so the committed summary
    profiles/rNN_pmc_traffic.json (tools/gpu_r5.sh runs the passes LAST, on the tree that is timed) is reported — but only if it is a summary OF THIS
    LIBRARY: every GEMM kernel symbol it lists must be a symbol of the libpclip.so that was just timed (round 4 reported counters of an earlier
    tree).  Returns (bytes or None, file name, note)."""
    try:
        from proto_clip_amd._lib import LIB_PATH
        with open(LIB_PATH, "rb") as f:
            blob = f.read()
    except Exception:
        blob = b""
    for name in PMC_TRAFFIC_FILES:
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                d = json.load(f)
            syms = [k.split(" grid=")[0] for k in d["kernels"] if "linear" in k]
            stale = [k for k in syms if k.encode() not in blob]
            # ... and it must cover the kernel family that does the work in this library (a round-4 summary knows nothing of linear4w_kernel)
            if b"linear4w_kernel" in blob and not any("linear4w_kernel" in k for k in syms):
                stale = stale or ["(no linear4w_kernel entry)"]
            if not syms or stale:
                return None, name, f"stale: {len(stale)} of {len(syms)} GEMM kernel symbols of the summary are not in this libpclip.so (e.g. {stale[0][:60] if stale else '-'}...)"
            return float(d["linear_kernel_hbm_bytes_per_launch"]), name, "symbols match this libpclip.so"
        except Exception:
            continue
    return None, None, "no committed summary"


def measure_clock(st, min_steps=40):
    """Shader clock and socket power while the step loop runs (proto_clip_amd.telemetry, 50 Hz), in a separate UN-TIMED loop so
    that the sampling thread cannot touch the timed region.  The GEMMs run at a data-dependent power cap (MI355X_MICROARCH.md,
    DVFS give-back): the roofline fraction is also stated against the dense peak at the clock the chip actually sustained."""
    from proto_clip_amd.telemetry import Sampler
    with Sampler(period=0.02, skip_s=0.3) as s:
        for _ in range(min_steps):
            step(st)
        torch.cuda.synchronize()
    return s.summary()


def _cpu_workload(batch):
    """(run, images): the oracle's path for one batch — fp32 ViT-B/16 encode_image (as clip.load(device='cpu') yields) + conv-3x adapter + P."""
    from oracle import clip_oracle, proto_oracle as po
    from proto_clip_amd import synth
    from proto_clip_amd.clip.model import BACKBONES, random_state_dict
    import math
    sd = random_state_dict(seed=1, **BACKBONES["ViT-B/16"])
    torch.manual_seed(1)
    s = int(math.ceil(math.sqrt(DIM)))
    g = torch.Generator().manual_seed(2)
    ad = {"conv1.weight": torch.randn(16, 1, 1, 1, generator=g).half(), "conv2.weight": (torch.randn(16, 16, 3, 3, generator=g) / 12).half(),
          "conv3.weight": (torch.randn(1, 16, 1, 1, generator=g) / 4).half()}
    for i, c in ((1, 16), (2, 16), (3, 1)):
        ad[f"bn{i}.weight"] = torch.ones(c, s, s).half()
        ad[f"bn{i}.bias"] = torch.zeros(c, s, s).half()
    split = synth.make_split(N_CLASS, SHOTS, DIM, 8, 8, seed=1)
    zi = po.proto_build(split.visual_memory_keys.t().contiguous(), N_CLASS, SHOTS)
    zt = po.l2norm_rows(split.textual_memory_bank.t().contiguous())
    imgs = synth.make_images(16, 224, seed=100, n_class=N_CLASS).repeat(batch // 16, 1, 1, 1)

    def run(x):
        f = clip_oracle.encode_image(sd, x, half=False).half()
        a = po.l2norm_rows(po.adapter_conv(po.l2norm_rows(f), ad, "conv-3x"))
        return po.P(a, zi, zt, ALPHA, BETA).max(1)[1]

    return run, imgs


def cpu_worker(args):
    """One process of the host-level CPU baseline (`bench.py --cpu-worker threads,batch,first_cpu,dir,index`): pins itself to `threads`
    logical CPUs from `first_cpu`, builds the oracle workload, warms up, waits for the parent's go file and times ONE batch."""
    threads, batch, first, d, idx = args.cpu_worker.split(",")
    threads, batch, first, idx = int(threads), int(batch), int(first), int(idx)
    try:
        os.sched_setaffinity(0, set(range(first, first + threads)))
    except OSError:
        pass
    torch.set_num_threads(threads)
    run, imgs = _cpu_workload(batch)
    run(imgs[:4])
    open(os.path.join(d, f"ready{idx}"), "w").close()
    t_wait = time.perf_counter()
    while not os.path.exists(os.path.join(d, "go")):
        if time.perf_counter() - t_wait > 300:
            raise SystemExit(3)
        time.sleep(0.005)
    t0 = time.time()
    run(imgs)
    t1 = time.time()
    with open(os.path.join(d, f"done{idx}"), "w") as f:
        f.write(f"{t0} {t1}")


def cpu_baseline(batch=64, budget_s=150.0):
    """The ORACLE (CPU restatement of the reference path) timed on the HOST's cores (north_star: the same box's host cores, core count stated)
    on a bounded sample of the same workload (~256 images per configuration): configurations processes x threads with DISJOINT CPU
    ranges — 1 x 16 in this process (one batch of 64), then k x 16 in k worker processes started together (k = 4, 8, 16 while k * 16 logical
    cores and 4 GB per process exist and the rate still rises; 256 / k images each, at least 16) — host-level rate = images of all processes /
    (last finish - first start); the best configuration is reported, `cores` = the logical cores it used."""
    import subprocess
    import tempfile
    ncpu = os.cpu_count() or 1
    prev = torch.get_num_threads()
    t_start = time.perf_counter()
    rates = {}
    run, imgs = _cpu_workload(batch)
    thr = min(16, ncpu)
    try:
        torch.set_num_threads(thr)
        run(imgs[:4])                                    # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        run(imgs)
        rates[(1, thr)] = batch / (time.perf_counter() - t0)
    finally:
        torch.set_num_threads(prev)
    del run, imgs
    try:
        with open("/proc/meminfo") as f:
            avail_gb = next(int(l.split()[1]) for l in f if l.startswith("MemAvailable")) / 2 ** 20
    except Exception:
        avail_gb = 0.0
    prev_rate = rates[(1, thr)]
    for k in (4, 8, 16):
        if k * thr > ncpu or k * 4.0 > 0.5 * avail_gb or time.perf_counter() - t_start > budget_s * 0.6:
            break
        pb = max(16, 4 * batch // k)                       # images per process: the sample stays ~256 images per configuration (bounded CPU time)
        d = tempfile.mkdtemp(prefix="pclip_cpu_")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{thr},{pb},{i * thr},{d},{i}"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(thr)))
                 for i in range(k)]
        try:
            t_w = time.perf_counter()
            while sum(os.path.exists(os.path.join(d, f"ready{i}")) for i in range(k)) < k and time.perf_counter() - t_w < 120 and all(p.poll() is None for p in procs):
                time.sleep(0.05)
            open(os.path.join(d, "go"), "w").close()
            for p_ in procs:
                p_.wait(timeout=180)
            spans = [tuple(map(float, open(os.path.join(d, f"done{i}")).read().split())) for i in range(k)]
            rates[(k, thr)] = k * pb / (max(t1 for _, t1 in spans) - min(t0 for t0, _ in spans))
            if rates[(k, thr)] < prev_rate:                # past the host's optimum (memory-bound): more processes only get slower
                break
            prev_rate = rates[(k, thr)]
        except Exception:
            for p_ in procs:
                if p_.poll() is None:
                    p_.kill()
            break
        finally:
            import shutil
            shutil.rmtree(d, ignore_errors=True)
    best = max(rates, key=rates.get)

    def numa_note():
        # which NUMA nodes the best configuration's CPU ranges fall on: worker i is pinned to logical CPUs [16 i, 16 i + 16)
        try:
            import glob
            nodes = {}
            for path in sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist")):
                nid = int(path.split("node")[-1].split("/")[0])
                cpus = set()
                for part in open(path).read().strip().split(","):
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
                nodes[nid] = cpus
            if not nodes:
                return "NUMA layout not exposed"
            where = [sorted({n for n, c in nodes.items() if c & set(range(i * best[1], (i + 1) * best[1]))}) for i in range(best[0])]
            return (f"{len(nodes)} NUMA node(s) (" + "; ".join(f"node {n}: {len(c)} logical CPUs" for n, c in sorted(nodes.items())) + "); the best configuration's "
                    f"{best[0]} x {best[1]} CPU ranges sit on node(s) " + ", ".join("+".join(map(str, w)) for w in where))
        except Exception as e:
            return f"NUMA layout unreadable ({type(e).__name__})"

    return dict(value=rates[best], unit="query images/sec", cores=best[0] * best[1], kind="port", processes=best[0], threads_per_process=best[1],
                sample=f"one batch per process (64 images in-process, 256 / k per worker, >= 16) through the oracle (fp32 ViT-B/16 encode_image + conv-3x adapter + P), processes x torch threads on "
                       f"disjoint logical-CPU ranges, host-level img/s: {', '.join(f'{k}x{t}: {v:.1f}' for (k, t), v in rates.items())}; host has {ncpu} logical cores; "
                       f"{numa_note()}; the reference module itself measured 14.2 img/s on 8 threads at survey time (SURVEY.md section 6)")


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: re-exec under it, one rank per GPU on this node."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


class HipHooks:
    """What `run` drives: the HIP path on cuda:LOCAL_RANK over RCCL.  tests/test_dist_cpu.py substitutes a CPU / gloo / oracle set to execute
    this file's own plumbing (rank / world environment, support-set sharding, barrier + max-over-ranks timing, the self-check collective, one
    JSON line from rank 0 only) without a GPU."""
    backend = "nccl"
    instrument = True                       # GEMM events, clock sampling, CPU baseline, extras: GPU-box only
    batch = BATCH
    build_state = staticmethod(build_state)
    step = staticmethod(step)
    self_check = staticmethod(self_check)

    @staticmethod
    def device(local):
        torch.cuda.set_device(local)
        return torch.device("cuda", local)

    @staticmethod
    def sync():
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the un-timed side measurements of the `extra` block")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)
    return run(args, HipHooks)


def run(args, hooks, out=None):
    out = out or sys.stdout
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # under torch.distributed.run (any N)
    if not launched and args.gpus > 1:
        self_launch(args)                                                 # never returns
    device = hooks.device(local)
    if launched:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on STDOUT when its first communicator is created; stdout carries exactly one JSON line (the contract), so file descriptor 1
        # points at stderr while the process group and its communicator come up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(hooks.backend, **({"device_id": device} if device.type == "cuda" else {}))
            dist.barrier()
            hooks.sync()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world} under torch.distributed.run: pass the same N to both")
    step, BATCH = hooks.step, hooks.batch

    st = hooks.build_state(device, rank, world)
    for _ in range(args.warmup):
        step(st)

    def barrier():
        if launched:
            dist.barrier()
        hooks.sync()

    # per-step device times beside the wall clock of the contract: one event per step boundary on the launch stream (recording is asynchronous and costs
    # nothing inside the timed region) -> median / p10 / p90 (hipEvent median; a box's clock wander is visible as the spread)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if device.type == "cuda" else None
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if evs: evs[i].record()
        step(st)
    if evs: evs[args.steps].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)) if evs else []
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(round(q * (len(step_ms) - 1))))] if step_ms else None
    if launched:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    check = hooks.self_check(st)   # every rank: the step contains the all-gather
    if launched:
        flag = torch.tensor([0 if check == "ok" else 1], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() and check == "ok":
            check = "MISMATCH on another rank"
    nogm = dict(launches=0, calls=0, total_ms=0.0, avg_us=0.0, tflops=0.0, flops=0.0)
    gm = measure_gemm(st) if hooks.instrument else nogm          # every rank runs it: the instrumented step contains the all-gather
    clk = measure_clock(st) if world == 1 and hooks.instrument else {"source": None}
    if rank == 0:
        traffic, traffic_file, traffic_note = pmc_traffic()
        sclk = (clk.get("sclk_mhz") or {}).get("median")
        power = (clk.get("power_w") or {}).get("median")
        imgs_per_s = args.steps * BATCH * world / dt
        ref_gflop = GFLOP_PER_IMG_ENCODER + GFLOP_PER_IMG_TAIL
        exe_gflop = gm["flops"] / 1e9 / BATCH + GFLOP_PER_IMG_ATTENTION + GFLOP_PER_IMG_TAIL     # what the HIP path executes: the last block runs on the class rows only
        line = {
            "metric": "query images/sec, ImageNet 16-shot ViT-B/16 (few-shot top-1 parity: tests/)",
            "value": imgs_per_s, "unit": "query images/sec", "n_gpus": world,
            "rccl_world_size": dist.get_world_size() if launched else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "median_ms": pct(0.5), "p10_ms": pct(0.1), "p90_ms": pct(0.9),
            "step_time_note": "ms_per_step = wall clock over the K steps between the two barriers (the contract); median / p10 / p90 = per-step HIP-event times on rank 0's launch stream inside the same loop",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "sclk_mhz_under_load": sclk, "power_w": power,
            "clock_source": {k: clk.get(k) for k in ("source", "samples", "sclk_mhz", "power_w")},
            "config": {"workload": "C3 ImageNet 16-shot ViT-B/16 conv-3x: prototype reduce + encode_image + adapter + dual-bank classify",
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "classes": N_CLASS, "shots": SHOTS, "embed_dim": DIM,
                       "alpha": ALPHA, "beta": BETA, "parallelism": f"dp{world} (support rows and queries sharded; all-gather of class sums)"},
            "self_check": check,
            "roofline": {"bound": "mfma", "kernel": "linear4w_kernel (four-wave 256 x 256 tiles, hand-scheduled asm K-loop) + linear_fast_kernel (row-split tails) + linear_small_kernel (fp16 MFMA GEMM: every encoder linear incl. patch embedding and projection, with bias / QuickGELU / residual add in its epilogue"
                                                    + "; the LayerNorms are separate passes (the reference's rounding points); the class-row tail runs the small-M variant)",
                         "achieved": gm["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": gm["tflops"] / MFMA_PEAK_TFLOPS,
                         "frac_at_sustained_clock": (gm["tflops"] / (MFMA_PEAK_TFLOPS * sclk / 2400.0)) if sclk else None,
                         # context, not the contract's peak: what the matrix pipe sustains on this class of operands under the 1.4 kW socket cap (register-only loops,
                         # tools/probe/mfma_power.hip, measured on other boxes of the pool; on zero operands the same loops reach the nominal 2.5 PFLOP/s)
                         "power_limited_reference": {"mfma_16x16x32_registers_only_tflops": 1940.0, "four_wave_k_loop_alone_tflops": 1435.0,
                                                     "frac_of_registers_only": gm["tflops"] / 1940.0, "source": "profiles/r05_mfma_power_probe.txt, r05_gemm4w_epilogue.txt"},
                         "traffic": traffic, "traffic_unit": f"HBM bytes per launch (profiles/{traffic_file}: {traffic_note})",
                         "launches_per_step": gm["launches"], "avg_launch_us": gm["avg_us"],
                         "gemm_ms_per_step": gm["total_ms"], "algorithmic_gflop_per_step": gm["flops"] / 1e9,
                         "per_variant": gm.get("per_variant"),
                         "timed_launches": gm.get("timed_launches"),
                         "method": "the GEMM kernels' own measurement hook (pclip_gemm_timing): every workgroup of every GEMM launch of one instrumented step folds the device's 100 MHz counter into (min begin, max end) — the kernel's span as rocprofv3 --kernel-trace reports it, with nothing between the kernels"},
            "whole_path": {"note": "reference-equivalent = the FLOPs the REFERENCE module spends per image (every token through all 12 blocks); executed = what the HIP path runs "
                                   "(GEMM launches of the instrumented step + attention contractions + adapter + similarity: the last block works on the class rows only)",
                           "reference_equivalent_gflop_per_image": ref_gflop, "reference_equivalent_tflops": imgs_per_s / world * ref_gflop / 1e3,
                           "reference_equivalent_frac_of_mfma_peak": imgs_per_s / world * ref_gflop / 1e3 / MFMA_PEAK_TFLOPS,
                           "executed_gflop_per_image": exe_gflop, "executed_tflops": imgs_per_s / world * exe_gflop / 1e3,
                           "executed_frac": imgs_per_s / world * exe_gflop / 1e3 / MFMA_PEAK_TFLOPS},
        }
        if world == 1 and not args.no_extra and hooks.instrument:
            try:
                line["extra"] = {"c2_kernels": c2_kernels(device), "c3_classify": c3_classify(device)}
            except Exception as e:                       # side measurements never take the headline down
                line["extra"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and hooks.instrument:
            line["cpu_baseline"] = cpu_baseline()
        if check != "ok":
            # a step whose rows depend on the batch around them is not the workload: no headline number, and a failing exit status behind the line
            line["invalid_value"], line["value"] = line["value"], None
        print(json.dumps(line), file=out, flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()
    if check != "ok":
        raise SystemExit(3)


if __name__ == "__main__":
    main()

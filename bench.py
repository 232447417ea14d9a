#!/usr/bin/env python3
"""bench.py — attention forward throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (flash_attn_func / flash_attn_varlen_func -> C-ABI -> HIP kernel)
over one batch of synthetic tensors already resident in HBM.  Default workload = BASELINE configs[1]
(C2: batch 4, 16 heads, head dim 128, seq 8192, bf16, non-causal) per GPU; with N GPUs every rank runs
its own batch shard (no collective on the data path: attention shards over batch) => weak scaling.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--scaling weak|strong]
  N>1: either under torchrun (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
  bench.py --gpus N: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the env), or plain `python bench.py --gpus N`:
  the parent then starts the N ranks itself (fresh child processes, before it has touched the GPU) and relays rank 0's line.
  --scaling strong (c5 only: BASELINE config 5 / SURVEY.md 8(d)): the global batch of 32 is split 32/N per GPU.
  --dry-run: CPU rehearsal of the launch / rendezvous / timing / reporting logic (gloo, no kernel; "dry_run": true).

Prints ONE JSON line on rank 0 with metric/value/unit (whole-job TFLOP/s), `roofline` (MFMA bound: the
kernel's algorithmic FLOPs / its HIP-event duration vs the 2.5 PFLOP/s dense bf16 peak) and `cpu_baseline`
(PyTorch-eager SDPA on the host cores, bounded sample; reported baseline, not a target).
FLOP convention = the reference's: 4*b*h*sq*sk*d, halved when causal
(benchmarks/benchmark_flash_attention.py:27-30, hopper/benchmark_attn.py:62-74); softmax not counted.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md (256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz)
C5_GLOBAL_BATCH = 32       # BASELINE config 5: global batch 32, strong-scaled over 1/2/4/8 GPUs
C5_DTYPE = "fp8_e4m3 (both products on v_mfma_scale_f32_32x32x64_f8f6f4, unit scales; fp32 accumulate, bf16 out)"
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 peak of the block-scaled MFMA (2x bf16 per clock), MI355X_MICROARCH.md

WORKLOADS = {
    # name: (description, batch, heads_q, heads_kv, seqlen, head_dim, causal, varlen lens or None)
    "c2": ("C2 b4 h16 d128 s8192 bf16 non-causal", 4, 16, 16, 8192, 128, False, None),
    "c3": ("C3 b4 h16 d128 s16384 bf16 causal", 4, 16, 16, 16384, 128, True, None),
    "c4": ("C4 varlen GQA hq32 hkv8 d128 lens 8192..1024 bf16 non-causal", 8, 32, 8, 8192, 128, False,
           [8192, 7168, 6144, 5120, 4096, 3072, 2048, 1024]),
    # C5: fp8 e4m3 inputs (bf16 output), descales = 1; global batch 32 strong-scaled in BASELINE.md -> here 4 per GPU
    # backward of C2 / C3: dq, dk, dv from (dout, q, k, v, out, lse); FLOPs = 2.5 x forward
    # (benchmarks/benchmark_flash_attention.py:27-30 mode="bwd")
    "c2_bwd": ("C2 backward b4 h16 d128 s8192 bf16 non-causal", 4, 16, 16, 8192, 128, False, None),
    "c3_bwd": ("C3 backward b4 h16 d128 s16384 bf16 causal", 4, 16, 16, 16384, 128, True, None),
    # decode step over a KV cache: sq = 1, GQA 32/8, cache of 8192 rows per sequence, one new row
    # appended in place; HBM-bound: the figure of merit is cache bytes read per second
    "decode": ("decode b32 hq32 hkv8 d128 cache 8192 bf16 (flash_attn_with_kvcache, 1 new row appended)", 32, 32, 8,
               8192, 128, False, None),
    "c5": ("C5 fp8 e4m3 b4 h16 d128 s8192 non-causal (native e4m3 MFMA, no expansion pass)", 4, 16, 16,
           8192, 128, False, None),
}


def flops_of(w):
    _, b, h, hk, s, d, causal, lens = w
    if lens is not None:
        f = sum(4 * h * d * L * L for L in lens)
    else:
        f = 4 * b * h * s * s * d
    if w[0].split()[1] == "backward":
        f = f * 5 // 2
    return f // 2 if causal else f


def algorithmic_bytes(w):
    _, b, h, hk, s, d, causal, lens = w
    rows = sum(lens) if lens is not None else b * s
    if w[0].split()[1] == "backward":  # read Q,K,V,O,dO + LSE; write dQ,dK,dV (+ D written and read once)
        return 2 * (4 * rows * h * d + 4 * rows * hk * d) + 4 * rows * h * 3
    if w[0].startswith("C5"):  # e4m3 Q, K, V (1 B/element), bf16 O, fp32 LSE
        return rows * h * d + 2 * rows * hk * d + 2 * rows * h * d + 4 * rows * h
    return 2 * (2 * rows * h * d + 2 * rows * hk * d) + 4 * rows * h  # Q+O, K+V (bf16) + LSE (fp32)


def measured_traffic(workload_key):
    """(bytes, source): HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/rNN_traffic.json), newest
    round that measured this workload -- a counter pass of the same command, NOT a measurement of this run (PMC passes
    need the profiler); (None, None) when no counter pass has been committed for it."""
    import glob
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload_key:
            best, src = d.get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(path)
    return best, src


def make_inputs(w, device, seed):
    import torch
    _, b, h, hk, s, d, causal, lens = w
    if w[0].startswith("C5"):
        g = torch.Generator(device=device).manual_seed(seed)
        mk = lambda hh: torch.randn(b, s, hh, d, device=device, dtype=torch.bfloat16, generator=g).to(torch.float8_e4m3fn)
        q, k, v = mk(h), mk(hk), mk(hk)
        ones = torch.ones(b, hk, device=device, dtype=torch.float32)
        return (q, k, v), {"fp8": True, "descale": ones}
    g = torch.Generator(device=device).manual_seed(seed)
    if lens is None:
        q = torch.randn(b, s, h, d, device=device, dtype=torch.bfloat16, generator=g)
        k = torch.randn(b, s, hk, d, device=device, dtype=torch.bfloat16, generator=g)
        v = torch.randn(b, s, hk, d, device=device, dtype=torch.bfloat16, generator=g)
        return (q, k, v), {}
    total = sum(lens)
    q = torch.randn(total, h, d, device=device, dtype=torch.bfloat16, generator=g)
    k = torch.randn(total, hk, d, device=device, dtype=torch.bfloat16, generator=g)
    v = torch.randn(total, hk, d, device=device, dtype=torch.bfloat16, generator=g)
    cu = torch.tensor([0] + list(__import__("itertools").accumulate(lens)), dtype=torch.int32, device=device)
    return (q, k, v), {"cu": cu, "max": max(lens)}


def cpu_baseline(w, budget_s=20.0):
    """PyTorch-eager SDPA (fused CPU kernel) on the host cores, on a head-slice of the same workload sized
    for ~budget_s of CPU work.  Checker/baseline only: imports oracle/, never the other way round."""
    import torch
    from oracle.attention_ref import sdpa_cpu
    name, b, h, hk, s, d, causal, lens = w
    s_eff = s if lens is None else lens[0]
    cores = torch.get_num_threads()
    torch.manual_seed(0)

    def run(nheads):
        q = torch.randn(1, s_eff, nheads, d, dtype=torch.bfloat16)
        k = torch.randn(1, s_eff, nheads, d, dtype=torch.bfloat16)
        v = torch.randn(1, s_eff, nheads, d, dtype=torch.bfloat16)
        sdpa_cpu(q, k, v, causal)  # warm-up
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            sdpa_cpu(q, k, v, causal)
        dt = (time.perf_counter() - t0) / reps
        fl = 4 * nheads * s_eff * s_eff * d / (2 if causal else 1)
        return dt, fl
    dt, fl = run(1)
    nheads = int(max(1, min(h, budget_s / 4.0 / max(dt, 1e-3))))
    if nheads > 1:
        dt, fl = run(nheads)
    return {
        "value": round(fl / dt / 1e12, 4), "unit": "TFLOP/s", "cores": cores, "kind": "port",
        "sample": f"torch SDPA (PyTorch-eager, CPU, bf16) on a (1,{s_eff},{nheads},{d}) slice of the workload, "
                  f"causal={causal}, 1 warm-up + 3 timed, {dt:.3f} s each, torch.get_num_threads()={cores}",
    }


def check_rows(w, q, k, v, extra, result, n_rows=64):
    """After the timed region (never inside it): 64 sampled rows of the tensor the last step produced, all heads, against
    the oracle -- a kernel that exits early or computes something else must not print a number unnoticed.  Bound = the
    tests' (|out - ref|max <= 2 |pt - ref|max + atol, tests/test_flash_attn.py:1121; fp8: hopper/test_flash_attn.py:193-194).
    Checker only: imports oracle/, never the other way round.  Returns (ok, max error, bound)."""
    import torch
    from oracle import attention_ref as oracle
    _, b, h, hk, s, d, causal, lens = w
    out = (result[0] if isinstance(result, (tuple, list)) else result).float().cpu()
    g = torch.Generator().manual_seed(1)
    fp8 = "fp8" in extra
    worst, worst_bound = None, None   # the sample that came closest to its bound
    seqs = [(i, 0, s) for i in range(b)] if lens is None else []
    if lens is not None:
        cu = extra["cu"].cpu().tolist()
        seqs = [(i, cu[i], cu[i + 1] - cu[i]) for i in range(len(lens))]
    per = max(1, n_rows // len(seqs))
    for i, start, length in seqs:
        rows = torch.randperm(length, generator=g)[:per].sort().values
        if lens is None:
            qi, ki, vi, oi = q[i, rows].cpu()[None], k[i].cpu()[None], v[i].cpu()[None], out[i, rows][None]
        else:
            sl = slice(start, start + length)
            qi, ki, vi, oi = q[sl][rows].cpu()[None], k[sl].cpu()[None], v[sl].cpu()[None], out[sl][rows][None]
        bias = None
        if causal:  # the row subset's causal mask as an additive bias (tests/test_oracle.py pins this form)
            jj = torch.arange(length).view(1, -1)
            bias = torch.where(jj <= rows.view(-1, 1), 0.0, float("-inf")).view(1, 1, len(rows), length)
        kw = dict(attn_bias=bias)
        if fp8:
            qi, ki, vi = qi.float(), ki.float(), vi.float()
            ref = oracle.attention_ref(qi, ki, vi, **kw)[0]
            pt = oracle.attention_ref(qi.to(torch.bfloat16), ki.to(torch.bfloat16), vi.to(torch.bfloat16), upcast=False,
                                      reorder_ops=True, intermediate_dtype=torch.float8_e4m3fn, **kw)[0]
            atol = 2 * (ref.float() + 0.3 - 0.3 - ref.float()).abs().max().item()
        else:
            ref = oracle.attention_ref(qi, ki, vi, **kw)[0]
            pt = oracle.attention_ref(qi, ki, vi, upcast=False, reorder_ops=True, **kw)[0]
            atol = 1e-5
        err = (oi - ref.float()).abs().max().item()
        bound = 2 * (pt.float() - ref.float()).abs().max().item() + atol
        if worst is None or not (err <= bound) or err - bound > worst - worst_bound:
            worst, worst_bound = err, bound
        if not (err <= bound):
            return False, err, bound
    return True, worst, worst_bound


def dry_run(args, w, rank, world, dist, timing_group):
    """CPU rehearsal (tests/test_bench_harness.py): same rendezvous, barriers, max-over-ranks timing and JSON line as the
    real run; the step is a fixed host-side wait instead of the kernel (there is no CPU fallback of the product path)."""
    import torch
    for _ in range(args.warmup):
        pass
    if dist is not None:
        dist.barrier(group=timing_group)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    if dist is not None:
        dist.barrier(group=timing_group)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=timing_group)
        elapsed = float(t.item())
    flops = flops_of(w)
    if rank == 0:
        print(json.dumps({
            "metric": "attn fwd TFLOPS (aggregate over GPUs; per-GPU in per_gpu) + %MFMA-peak, bf16 hdim128 seq8192",
            "value": round(flops * world * args.steps / elapsed / 1e12, 2), "unit": "TFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "none (dry run)",
            "data": "none", "dry_run": True,
            "config": {"workload": w[0], "batch_per_gpu": w[1], "sharding": f"batch shard x{world}, no collective"},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this parent has made no
    GPU call: it never imports torch), relay rank 0's stdout (the one JSON line), fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while every child is polled: a rank that dies before the rendezvous (bad device,
    # import error) must not leave its siblings waiting in init_process_group until the store times out
    import threading
    import time
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * len(procs)
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):  # first failure: end the siblings (exact PIDs this function started)
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if any(rcs):
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the clocks to settle (30 steps after 5 warm-ups read ~2 % low: the first launches of a
    # process run at a lower clock), still well under a second of GPU time for every workload
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))  # c2 = BASELINE metric config
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="strong: c5 only, global batch 32 split over the GPUs (BASELINE config 5)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant override (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run comparison of 64 sampled output rows with the oracle")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of launch/rendezvous/reporting (gloo, no kernel)")
    args = ap.parse_args()
    if args.scaling == "strong" and args.workload != "c5":
        ap.error("--scaling strong is defined for --workload c5 (global batch 32, BASELINE config 5)")
    if args.scaling == "strong" and C5_GLOBAL_BATCH % args.gpus:
        ap.error(f"--scaling strong: {C5_GLOBAL_BATCH} batches do not split over {args.gpus} GPUs")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run and os.environ.get("FA_BENCH_DRY_FAIL_RANK") == str(rank):  # rehearsal of a rank lost before the rendezvous
        sys.exit(3)
    if args.gpus != world:
        print(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    import torch
    if not args.dry_run and not torch.cuda.is_available():
        print("bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("FA_BENCH_FORCE_DIST"):  # (the env switch rehearses this path on one GPU)
        import datetime
        import torch.distributed as dist
        RENDEZVOUS_TIMEOUT = datetime.timedelta(seconds=int(os.environ.get("FA_BENCH_RENDEZVOUS_S", "600")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # single node; the container hostname may not resolve
        # RCCL prints a version banner on stdout when it initialises: keep stdout for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.dry_run:
                dist.init_process_group(backend="gloo", timeout=RENDEZVOUS_TIMEOUT)
                timing_group = dist.group.WORLD
            else:
                dist.init_process_group(backend="nccl", device_id=device, timeout=RENDEZVOUS_TIMEOUT)
                dist.barrier()  # one collective over RCCL/xGMI: brings the communicator up outside the timed region
                torch.cuda.synchronize()
                # the timing bracket is a host-side barrier (gloo): attention shards over batch, there is no data-path
                # collective, and a device-side barrier would add its own kernel + launch latency to the measured time
                timing_group = dist.new_group(backend="gloo")
            dist.barrier(group=timing_group)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    w = WORKLOADS[args.workload]
    if args.scaling == "strong":  # this rank's share of the global batch
        w = (w[0].replace("b4", f"b{C5_GLOBAL_BATCH // world} (global b{C5_GLOBAL_BATCH} / {world})"),
             C5_GLOBAL_BATCH // world) + w[2:]
    if args.dry_run:
        return dry_run(args, w, rank, world, dist, timing_group if dist is not None else None)

    import flash_attention_annotated_amd as fa
    from flash_attention_annotated_amd import _lib

    lib = _lib.load()
    if args.variant:
        lib.fa_set_default_variant(args.variant)

    if args.workload == "decode":
        (q, k, v), extra = (None, None, None), {}
    else:
        (q, k, v), extra = make_inputs(w, device, seed=rank)  # each rank: its own batch shard, already in HBM

    decode_state = None
    if args.workload == "decode":
        _, b_, h_, hk_, s_, d_, _, _ = w
        g = torch.Generator(device=device).manual_seed(rank)
        mk = lambda *shape: torch.randn(*shape, device=device, dtype=torch.bfloat16, generator=g)
        decode_state = (mk(b_, 1, h_, d_), mk(b_, s_, hk_, d_), mk(b_, s_, hk_, d_), mk(b_, 1, hk_, d_), mk(b_, 1, hk_, d_),
                        torch.full((b_,), s_ - 1, dtype=torch.int32, device=device))
    bwd_state = None
    if args.workload.endswith("_bwd"):
        from flash_attention_annotated_amd import flash_attn_2_cuda as ext
        scale = w[5] ** -0.5
        out, lse, _, _ = ext.fwd(q, k, v, None, None, 0.0, scale, w[6], -1, -1, 0.0, False, None)
        g = torch.randn(q.shape, device=device, dtype=q.dtype, generator=torch.Generator(device=device).manual_seed(100 + rank))
        bwd_state = (ext, g, out, lse, torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), scale)

    def step():
        if decode_state is not None:
            qd, kc, vc, kn, vn, cs = decode_state
            return fa.flash_attn_with_kvcache(qd, kc, vc, kn, vn, cache_seqlens=cs)
        if bwd_state is not None:
            ext, g, out, lse, dq, dk, dv, scale = bwd_state
            return ext.bwd(g, q, k, v, out, lse, dq, dk, dv, None, 0.0, scale, w[6], -1, -1, 0.0, False, None, None)
        if "fp8" in extra:
            from flash_attention_annotated_amd import hopper_interface as fa3
            d1 = extra["descale"]
            return fa3.flash_attn_func(q, k, v, causal=w[6], q_descale=d1, k_descale=d1, v_descale=d1)
        if "cu" in extra:
            return fa.flash_attn_varlen_func(q, k, v, extra["cu"], extra["cu"], extra["max"], extra["max"],
                                             causal=w[6])
        return fa.flash_attn_func(q, k, v, causal=w[6])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # per-launch kernel durations from HIP events on the launch stream (torch's current stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(group=timing_group)
    t0 = time.perf_counter()
    result = None
    for a, b_ in ev:
        a.record()
        result = step()
        b_.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(group=timing_group)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=timing_group)
        elapsed = float(t.item())
    kernel_ms = sorted(a.elapsed_time(b_) for a, b_ in ev)
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    flops = flops_of(w)
    total_flops = flops * world * args.steps
    value = total_flops / elapsed / 1e12
    achieved = flops / (avg_kernel_ms * 1e-3) / 1e12

    if rank == 0 and args.workload == "decode":
        _, b_, h_, hk_, s_, d_, _, _ = w
        cache_bytes = 2 * b_ * s_ * hk_ * d_ * 2            # K and V rows read once
        other = 2 * b_ * h_ * d_ * 2 + 4 * b_ * hk_ * d_ * 2  # q, out, appended rows (read + write)
        gbs = (cache_bytes + other) * world * args.steps / elapsed / 1e9
        ach = (cache_bytes + other) / (avg_kernel_ms * 1e-3) / 1e9
        print(json.dumps({
            "metric": "decode attention: KV-cache bytes streamed per second (aggregate over GPUs)", "value": round(gbs, 1),
            "unit": "GB/s", "per_gpu": round(gbs / world, 1), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (torch.randn N(0,1), seed = rank)",
            "config": {"workload": w[0], "batch_per_gpu": b_, "heads_q": h_, "heads_kv": hk_, "cache_len": s_,
                       "head_dim": d_, "sharding": f"batch shard x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(ach / 8000.0, 4), "traffic": None, "step_ms_avg": round(avg_kernel_ms, 4),
                         "step_ms_min": round(kernel_ms[0], 4), "algorithmic_bytes_per_step": cache_bytes + other,
                         "note": "one step = append launch + attention launch (HIP events around both)"},
            "cpu_baseline": None}), flush=True)
    elif rank == 0:
        traffic, traffic_src = measured_traffic(args.workload)
        peak = PEAK_FP8_TFLOPS if args.workload == "c5" else PEAK_BF16_TFLOPS
        out = {
            "metric": ("attn bwd TFLOPS (reference convention: 2.5 x forward FLOPs)" if args.workload.endswith("_bwd") else
                       "attn fwd TFLOPS (aggregate over GPUs; per-GPU in per_gpu) + %MFMA-peak, bf16 hdim128 seq8192"),
            "value": round(value, 2),
            "unit": "TFLOP/s",
            "per_gpu": round(value / world, 2),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": C5_DTYPE if args.workload == "c5" else "bf16",
            "data": "synthetic (torch.randn N(0,1), seed = rank)",
            "config": {"workload": w[0], "batch_per_gpu": w[1], "heads_q": w[2], "heads_kv": w[3], "seqlen": w[4],
                       "head_dim": w[5], "causal": w[6], "sharding": f"batch shard x{world}, no collective"},
            "roofline": {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "frac_of_bf16_peak": round(achieved / PEAK_BF16_TFLOPS, 4),
                "instruction": ("v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3, unit block scales)" if args.workload == "c5" else
                                "v_mfma_f32_32x32x16_bf16"),
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms_avg": round(avg_kernel_ms, 4), "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 4),
                "kernel_ms_min": round(kernel_ms[0], 4), "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": algorithmic_bytes(w),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        else:
            out["cpu_baseline"] = None
        # outside the timed region: 64 sampled rows of what the last step wrote, against the oracle
        if args.workload in ("c2", "c3", "c4", "c5") and not args.no_check:
            ok, err, bound = check_rows(w, q, k, v, extra, result)
            out["checked"] = bool(ok)
            out["check"] = {"rows": 64, "max_err": round(err, 6), "bound": round(bound, 6),
                            "against": "oracle/attention_ref.py, all heads, bound 2 |pt - ref| + atol"}
            if not ok:
                print(json.dumps(out), flush=True)
                print(f"bench.py: sampled output rows differ from the oracle (err {err:.3e} > bound {bound:.3e})", file=sys.stderr)
                sys.exit(3)
        else:
            out["checked"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Developer aid: A/B of library builds in ONE process, interleaved rounds (cdna guide rule 24).
Usage: python tools/ab_interleaved.py [--rounds R] [--shapes c2,c3,s2048,...] lib_a.so lib_b.so ...   ("tree" = the in-tree library)
Swaps the ctypes handle behind flash_attention_annotated_amd._lib between timings; prints median / best TFLOP/s per build."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib

SHAPES = {  # name: (b, s, h, d, causal, dtype)
    "c2": (4, 8192, 16, 128, False), "c3": (4, 16384, 16, 128, True), "s512": (32, 512, 16, 128, False),
    "s1024": (16, 1024, 16, 128, False), "s2048": (8, 2048, 16, 128, False), "s2048c": (8, 2048, 16, 128, True),
    "s4096": (4, 4096, 16, 128, False), "d64": (2, 8192, 32, 64, False), "d64c": (2, 8192, 32, 64, True),
    "d96": (2, 8192, 21, 96, False), "d256": (2, 8192, 8, 256, False), "d256c": (2, 8192, 8, 256, True),
    "d192": (2, 8192, 10, 192, False), "d160": (2, 8192, 12, 160, False),
    "c5": (4, 8192, 16, 128, False), "c5c": (4, 8192, 16, 128, True),   # fp8 e4m3 inputs (FA3 surface)
}
args = sys.argv[1:]
rounds, shapes = 5, ["c2"]
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
if "--shapes" in args:
    i = args.index("--shapes"); shapes = args[i + 1].split(","); del args[i:i + 2]
libs = args or ["tree"]
handles = {}
for name in libs:
    _lib._lib = None
    if name == "tree":
        os.environ.pop("FA_FWD_LIB", None)
    else:
        os.environ["FA_FWD_LIB"] = os.path.abspath(name)
    handles[name] = _lib.load()

def t(f, n=20):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, e in ev:
        a.record(); f(); e.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(e) for a, e in ev)[n // 2]

for sh in shapes:
    b, s, h, d, causal = SHAPES[sh]
    q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    fl = 4 * b * h * s * s * d / (2 if causal else 1)
    if sh.startswith("c5"):
        from flash_attention_annotated_amd import hopper_interface as fa3
        q, k, v = (x.to(torch.float8_e4m3fn) for x in (q, k, v))
        run = lambda: fa3.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
    else:
        run = lambda: fa.flash_attn_func(q, k, v, causal=causal, return_attn_probs=True)
    res = {n: [] for n in libs}
    for n in libs:   # warm up (clock, caches, lazy module load)
        _lib._lib = handles[n]
        for _ in range(10): run()
    torch.cuda.synchronize()
    for r in range(rounds):
        for n in libs:
            _lib._lib = handles[n]
            res[n].append(t(run))
    outs = {}
    for n in libs:   # the builds must agree (schedule variants are bit-identical by construction)
        _lib._lib = handles[n]
        r = run()
        outs[n] = (r[0].float(), r[1].float())
    for n in libs[1:]:
        do = (outs[n][0] - outs[libs[0]][0]).abs().max().item()
        dl = (outs[n][1] - outs[libs[0]][1]).abs().max().item()
        if do != 0 or dl != 0:
            print(f"   !! {os.path.basename(n)} differs from {os.path.basename(libs[0])}: out {do:.3e} lse {dl:.3e}", flush=True)
    for n in libs:
        ms = sorted(res[n])
        print(f"{sh:7s} {os.path.basename(n):28s} median {fl / ms[len(ms) // 2] / 1e9:7.1f}  best {fl / ms[0] / 1e9:7.1f} TFLOP/s   ({ms[len(ms) // 2]:.4f} ms)", flush=True)

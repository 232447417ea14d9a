"""Developer aid: shader cycles per fast half-step (and per phase) of fwd_kernel_w64 and the clock the chip holds while the
kernel runs.  Needs the instrumented build (stamps perturb the loop by ~100 cycles each: they drain lgkmcnt)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFA_CYCLES -I include -I flash_attention_annotated_amd/csrc \
        flash_attention_annotated_amd/csrc/fa_fwd_api.hip flash_attention_annotated_amd/csrc/fa_bwd_api.hip -o tools/bin/libfa_cycles.so
and FA_FWD_LIB=tools/bin/libfa_cycles.so.  Stamp 2i = entry of a fast half-step (phase 1: QK^T MFMAs || softmax of B),
stamp 2i+1 = between its phases (phase 2: PV MFMAs || softmax of A); half-steps alternate KB = 0 (carries the 8 LDS-DMA
pieces of the next tiles) and KB = 1 (ends in the tile barrier)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib

lib = _lib.load()
lib.fa_debug_read_cycles.argtypes = [ctypes.c_void_p]
b, s, h, d = 4, 8192, 16, 128
q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for _ in range(200):  # let the clock settle
    fa.flash_attn_func(q, k, v)
torch.cuda.synchronize()
buf = np.zeros((256, 4, 64), dtype=np.uint64)
assert lib.fa_debug_read_cycles(buf.ctypes.data) == 0
if buf[0, 0, 44] > 0:  # generated asm loop (head-dim tile 128): one stamp pair around the whole block
    cyc = (buf[:, :, 41] - buf[:, :, 40]).astype(np.float64)
    wall = (buf[:, :, 43] - buf[:, :, 42]).astype(np.float64) * 10.0
    half = buf[:, :, 45].astype(np.float64)
    print(f"asm loop: tiles requested {int(buf[0, 0, 44])}, half-steps done median {np.median(half):.0f}; cycles per 64-key tile median "
          f"{np.median(cyc / (half / 2)):.0f} (p10 {np.percentile(cyc / (half / 2), 10):.0f}, p90 {np.percentile(cyc / (half / 2), 90):.0f}; ideal 2048); "
          f"in-kernel clock {np.median(cyc / wall):.3f} GHz; block wall time {np.median(wall) / 1e3:.1f} us")
    t0, t1, t2, t3 = (buf[:, :, k].astype(np.float64) * 0.01 for k in (46, 42, 43, 47))  # us
    print(f"workgroup timeline (wave medians, us): entry -> block {np.median(t1 - t0):.2f} | asm block {np.median(t2 - t1):.2f} | "
          f"block -> O stores issued {np.median(t3 - t2):.2f} | total {np.median(t3 - t0):.2f}")
    if buf[0, 0, 48] > 0:  # finer stamps (100 MHz wall clock)
        e = {k: buf[:, :, k].astype(np.float64) * 0.01 for k in (46, 48, 49, 50, 42, 43, 51, 52, 47)}
        seg = [("entry -> Q loads + first K/V tiles requested", 46, 48), ("-> Q landed (vmcnt 0)", 48, 49), ("-> first tile barrier", 49, 50),
               ("-> asm block entry (first scores, softmax A(0))", 50, 42), ("asm block", 42, 43), ("-> key sweep done", 43, 51),
               ("-> O normalised, in LDS", 51, 52), ("-> O stores issued", 52, 47)]
        print("  " + " | ".join(f"{n} {np.median(e[b_] - e[a_]):.2f}" for n, a_, b_ in seg))
n = int(buf[0, 0, 61])
if n < 8:
    sys.exit(0)
st = buf[:, :, :n].astype(np.float64)
dt = np.diff(st, axis=2)                      # (wg, wave, n-1) cycles between consecutive stamps
wall = (buf[:, :, 63].astype(np.float64) - buf[:, :, 62].astype(np.float64)) * 10.0  # ns
clk = (st[:, :, -1] - st[:, :, 0]) / wall     # GHz
print(f"C2 b{b} s{s} h{h} d{d}: {n} stamps per wave; in-kernel clock median {np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f})")
# stamps alternate: entry(KB0) mid(KB0) entry(KB1) mid(KB1) ...
names = ["KB0 phase 1 (QK + DMA K)", "KB0 phase 2 (PV + DMA V)", "KB1 phase 1 (QK)", "KB1 phase 2 (PV) + barrier + driver"]
for i in range(4):
    x = dt[:, :, i::4]
    if x.size:
        print(f"   {names[i]:40s} median {np.median(x):7.0f} cycles   p10 {np.percentile(x, 10):7.0f}   p90 {np.percentile(x, 90):7.0f}   ideal 512 (16 MFMA x 32)")
tile = st[:, :, 4::4][:, :, 1:] - st[:, :, 4::4][:, :, :-1]
print(f"   per 64-key tile: median {np.median(tile):.0f} cycles (ideal 2048) = {np.median(tile) / np.median(clk) / 1e3:.3f} us")
for w in range(4):
    print(f"   wave {w}: first stamps", (st[0, w, :9] - st[0, 0, 0]).astype(np.int64).tolist())

"""Developer aid: where a forward workgroup's time goes outside the key sweep, and how long a CU waits between two
workgroups.  Needs a stamped build (100 MHz wall-clock stamps; the window of 256 workgroups kept is a compile-time constant):
    bash tools/build_variant.sh cyc1024 -DFA_CYCLES -DFA_CYCLES_WG0=1024
    FA_FWD_LIB=tools/bin/libfa_cyc1024.so python tools/wg_timeline.py [b s h causal]
Stamps (fa_fwd_kernel_w64.h): 53 kernel entry, 46 item decoded, 48 Q + first K/V tiles requested, 49 landed, 50 first tile
barrier, 42 / 43 around the generated block, 51 key sweep done, 52 O normalised and staged, 47 O stores issued (wave end);
54 = XCC_ID << 32 | HW_ID.  Two workgroups that ran back to back on one CU give the dispatch gap (entry - predecessor's end)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib

lib = _lib.load()
lib.fa_debug_read_cycles.argtypes = [ctypes.c_void_p]
b, s, h = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 8192, 16)
causal = len(sys.argv) > 4 and sys.argv[4] == "1"
d = 128
q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for _ in range(100):
    fa.flash_attn_func(q, k, v, causal=causal)
torch.cuda.synchronize()
buf = np.zeros((256, 4, 64), dtype=np.uint64)
assert lib.fa_debug_read_cycles(buf.ctypes.data) == 0
ok = buf[:, 0, 47] > 0
print(f"b{b} s{s} h{h} d{d} causal={causal}: {int(ok.sum())} stamped workgroups of {b * h * ((s + 255) // 256)}")
us = lambda kk: buf[ok][:, :, kk].astype(np.float64) * 0.01
seg = [("entry -> item decoded", 53, 46), ("-> Q + first K/V tiles requested", 46, 48), ("-> landed (vmcnt 0)", 48, 49),
       ("-> first tile barrier", 49, 50), ("-> generated block entry (first scores, softmax A(0))", 50, 42), ("block", 42, 43),
       ("-> key sweep done", 43, 51), ("-> O normalised + staged", 51, 52), ("-> O stores issued / wave end", 52, 47)]
have_block = buf[ok][:, :, 42].min() > 0
tot = 0.0
for n, a_, b_ in seg:
    if not have_block and (a_ in (42, 43) or b_ in (42, 43)):
        continue
    x = us(b_) - us(a_)
    tot += np.median(x)
    print(f"  {n:58s} median {np.median(x):7.2f} us   p10 {np.percentile(x, 10):7.2f}   p90 {np.percentile(x, 90):7.2f}")
whole = us(47) - us(53)
print(f"  entry -> wave end: median {np.median(whole):.2f} us; outside the block: {np.median(whole) - np.median(us(43) - us(42)) if have_block else float('nan'):.2f} us")
# dispatch gap: per CU, sort the workgroups by entry time, gap = entry(next) - max over waves of end(previous)
hw = buf[ok][:, 0, 54]
cu_key = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(65536) + (hw & np.uint64(0xff00))  # xcc, se/sh/cu
start = us(53).min(axis=1)
end = us(47).max(axis=1)
gaps = []
for key in np.unique(cu_key):
    idx = np.where(cu_key == key)[0]
    idx = idx[np.argsort(start[idx])]
    for a_, b_ in zip(idx[:-1], idx[1:]):
        g = start[b_] - end[a_]
        if -5 < g < 50:
            gaps.append(g)
if gaps:
    gaps = np.array(gaps)
    print(f"  CU hand-over (end of a workgroup -> entry of the next on the same CU): n {len(gaps)}, median {np.median(gaps):.2f} us, "
          f"p10 {np.percentile(gaps, 10):.2f}, p90 {np.percentile(gaps, 90):.2f}")
print(f"  distinct CUs seen: {len(np.unique(cu_key))}")

"""Developer aid: item timeline of the PERSISTENT forward kernel (stamped build: bash tools/build_variant.sh cycp -DFA_CYCLES;
FA_FWD_LIB=tools/bin/libfa_cycp.so FA_FWD_PERSIST=1 python tools/wg_timeline_persist.py [b s h causal]).
Stamps of a workgroup's items 1 and 2 (fa_fwd_kernel_w64.h, FA_PSTAMP): 55 item switch done, 56 Q fragments read + next item
known + its Q requested, 42 / 43 around the generated block (44 / 45: masked block), 51 sweep closed (last barrier), 57 O stores issued."""
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
q, k, v = (torch.randn(b, s, h, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for _ in range(100):
    fa.flash_attn_func(q, k, v, causal=causal)
torch.cuda.synchronize()
buf = np.zeros((128, 2, 4, 64), dtype=np.uint64)   # (workgroup, item 1 / 2, wave, stamp)
assert lib.fa_debug_read_cycles(buf.ctypes.data) == 0
it1 = buf[:, 0].astype(np.float64) * 0.01          # us
it2 = buf[:, 1].astype(np.float64) * 0.01
ok = (buf[:, 0, 0, 57] > 0) & (buf[:, 1, 0, 55] > 0)
print(f"b{b} s{s} h{h} causal={causal}: {int(ok.sum())} workgroups with stamped items 1 and 2")
a = it1[ok]
nx = it2[ok]
def med(x): return f"median {np.median(x):7.2f} us  p10 {np.percentile(x, 10):7.2f}  p90 {np.percentile(x, 90):7.2f}"
print("  item switch done -> Q fragments, next item decoded, its Q requested   ", med(a[:, :, 56] - a[:, :, 55]))
blk0 = np.where(a[:, :, 42] > 0, a[:, :, 42], a[:, :, 44])
blk1 = np.where(a[:, :, 43] > 0, a[:, :, 43], a[:, :, 45])
print("  -> generated block entry (first scores, softmax A(0))                 ", med(blk0 - a[:, :, 56]))
print("  first block                                                           ", med(blk1 - blk0))
print("  last block exit -> sweep closed (incl. masked block / generic tail)    ", med(a[:, :, 51] - np.maximum(a[:, :, 43], a[:, :, 45])))
print("  -> O normalised, staged, stores issued                                ", med(a[:, :, 57] - a[:, :, 51]))
if a[:, :, 58].min() > 0:
    print("       sweep closed -> loop top ", med(a[:, :, 59] - a[:, :, 51]), "\n       -> lane tables      ", med(a[:, :, 60] - a[:, :, 59]),
          "\n       -> LSE written      ", med(a[:, :, 58] - a[:, :, 60]), "\n       -> O stores issued  ", med(a[:, :, 57] - a[:, :, 58]))
print("  -> item switch done (O zeroed, state reset)                           ", med(nx[:, :, 55] - a[:, :, 57]))
print("  whole item                                                            ", med(nx[:, :, 55] - a[:, :, 55]))
print("  last block exit of item i -> generated block entry of item i+1        ", med(np.where(nx[:, :, 42] > 0, nx[:, :, 42], nx[:, :, 44]) - np.maximum(a[:, :, 43], a[:, :, 45])))

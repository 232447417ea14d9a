"""Developer aid: where a fwd_kernel_w64 workgroup spends its time outside the main loop.  Needs the instrumented build
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFA_TIMING -I include -I flash_attention_annotated_amd/csrc \
        flash_attention_annotated_amd/csrc/fa_fwd_api.hip flash_attention_annotated_amd/csrc/fa_bwd_api.hip -o tools/_timing/libfa_timing.so
and FA_FWD_LIB=tools/_timing/libfa_timing.so.  Timestamps (100 MHz wall clock, per wave): 0 after argument decode,
1 prologue loads issued, 2 first tiles landed (barrier), 3 first scores + softmax done, 4 main loop done, 5 last loads
landed (barrier), 6 O stores issued, 7 O stores retired.  Prints mean phase lengths and the gap between consecutive
workgroups on the same CU (dispatch + argument decode)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import flash_attention_annotated_amd as fa
from flash_attention_annotated_amd import _lib

lib = _lib.load()
lib.fa_debug_read_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (b, s, h, d, causal) in ((4, 8192, 16, 128, False), (8, 2048, 16, 128, False), (32, 512, 16, 128, False), (32, 512, 16, 128, True),
                            (8, 2048, 32, 64, False)):
    q, k, v = (torch.randn(b, s, h, d, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    lib.fa_set_default_variant(3)  # the pipelined kernel regardless of the shape policy
    for _ in range(3):
        fa.flash_attn_func(q, k, v, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fa.flash_attn_func(q, k, v, causal=causal); e1.record()
    torch.cuda.synchronize()
    lib.fa_set_default_variant(0)
    n_wg = min(4096, b * h * ((s + 255) // 256))
    buf = np.zeros((n_wg, 4, 8), dtype=np.uint64)
    assert lib.fa_debug_read_timing(buf.ctypes.data, n_wg) == 0
    t = buf.astype(np.float64) * 0.01  # us
    w0 = t[:, 0, :]  # wave 0
    names = ["decode->loads issued", "loads issued->tiles landed", "first scores+softmax", "main loop", "final DMA barrier",
             "epilogue to stores issued", "stores retire"]
    ph = np.diff(w0, axis=1)
    print(f"b{b} s{s} h{h} d{d} causal={causal}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {n_wg} workgroups")
    for i, nm in enumerate(names):
        print(f"   {nm:32s} mean {ph[:, i].mean():7.2f} us   p10 {np.percentile(ph[:, i], 10):7.2f}   p90 {np.percentile(ph[:, i], 90):7.2f}")
    total = w0[:, 7] - w0[:, 0]
    print(f"   {'T0..T7 per workgroup':32s} mean {total.mean():7.2f} us")
    # gap between consecutive workgroups on a CU: sort all (start, end) and greedily chain: the next WG on a CU starts
    # right after one ends; estimate = median over WGs of (start - latest end before it among 256 slots)
    order = np.argsort(w0[:, 0])
    ends = np.sort(w0[:, 7])
    starts = w0[order, 0]
    first_wave = starts[:256].max() - starts[0]
    gaps = []
    for i, st in enumerate(starts[256:], 256):
        # the (i-255)-th end frees the slot this workgroup takes
        gaps.append(st - ends[i - 256])
    if gaps:
        g = np.array(gaps)
        print(f"   {'end of a WG -> T0 of its successor':32s} mean {g.mean():7.2f} us   p10 {np.percentile(g, 10):7.2f}   p90 {np.percentile(g, 90):7.2f}")
    print(f"   first 256 workgroups reach T0 within {first_wave:.2f} us of each other; waves of a WG: T4 spread mean "
          f"{(t[:, :, 4].max(1) - t[:, :, 4].min(1)).mean():.2f} us")

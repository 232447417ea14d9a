"""ORACLE tooling — run ONLY in the build container (needs /root/reference, read-only).

1. Imports the reference's own pure-torch oracle `attention_ref`
   (/root/reference/tests/test_util.py:185-274 and /root/reference/hopper/test_util.py:226-348)
   and asserts that this repo's restatement (oracle/attention_ref.py) and the C fp64 restatement
   (oracle/attention_ref_c.c) reproduce it on a sweep of cases.
2. Freezes the reference's outputs into tests/golden/attention_ref_golden.pt so the GPU box
   (which never sees /root/reference) can replay them.  Inputs are stored as a seed recipe plus a
   checksum; outputs are stored in full.  A fixture is data only: no reference source text.

Usage:  python oracle/make_golden.py
"""
import ctypes
import importlib.util
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import attention_ref as mine  # noqa: E402
from oracle.cases import CASES, FA3_CASES, GRAD_CASES, make_grad_output, make_alibi_slopes, make_descales, make_inputs, padding_masks, checksum  # noqa: E402


def import_reference():
    """tests/test_util.py pulls in the `flash_attn` package, whose __init__ imports the compiled
    extension; an empty stand-in module object lets the pure-Python parts import."""
    sys.modules.setdefault("flash_attn_2_cuda", types.ModuleType("flash_attn_2_cuda"))
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_test_util_fa2", os.path.join(REF, "tests/test_util.py"))
    fa2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fa2)
    sys.path.insert(0, os.path.join(REF, "hopper"))
    spec = importlib.util.spec_from_file_location("ref_test_util_fa3", os.path.join(REF, "hopper/test_util.py"))
    fa3 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fa3)
    return fa2, fa3


def import_reference_alibi():
    """attn_bias_from_alibi_slopes lives in tests/test_flash_attn.py (:29-56), a module that queries the GPU at import.
    Only that one function is compiled out of the file's syntax tree and run here."""
    import ast
    from einops import rearrange, repeat
    path = os.path.join(REF, "tests/test_flash_attn.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "attn_bias_from_alibi_slopes"]
    ns = {"torch": torch, "rearrange": rearrange, "repeat": repeat}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns["attn_bias_from_alibi_slopes"]


def c_oracle():
    path = os.path.join(ROOT, "oracle/_ref/liboracle_attn.so")
    lib = ctypes.CDLL(path)
    f = lib.fa_oracle_attention_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_float] + [ctypes.c_int] * 3 + [ctypes.c_float]

    def run(q, k, v, scale, causal, window, softcap):
        q, k, v = (t.float().contiguous() for t in (q, k, v))
        b, sq, h, d = q.shape
        sk, hk = k.shape[1], k.shape[2]
        out = torch.empty_like(q)
        lse = torch.empty(b, h, sq, dtype=torch.float32)
        st = f(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), b, sq, sk, h, hk, d,
               scale, int(causal), window[0], window[1], softcap)
        assert st == 0
        return out, lse
    return run


def main():
    fa2, fa3 = import_reference()
    ref_alibi = import_reference_alibi()
    crun = c_oracle()
    golden = {}
    worst = 0.0
    for name, c in CASES.items():
        q, k, v = make_inputs(c)
        qm, km = padding_masks(c)
        kw = dict(causal=c["causal"], window_size=c["window"], softcap=c["softcap"])
        if c.get("fp8"):
            # fp8 cases are pinned to the FA3 oracle (the only one with descales): hopper/test_util.py:226-348,
            # called the way hopper/test_flash_attn.py:163-181 does
            qd, kd, vd = make_descales(c)
            dkw = dict(q_descale=qd, k_descale=kd, v_descale=vd)
            ref_out, _ = fa3.attention_ref(q, k, v, None, None, **kw, **dkw)
            ref_pt, _ = fa3.attention_ref(q, k, v, None, None, **kw, **dkw, upcast=False, reorder_ops=True,
                                          intermediate_dtype=torch.float8_e4m3fn)
            ref_out32, _ = fa3.attention_ref(q.float(), k.float(), v.float(), None, None, **kw, **dkw)
            my_out, _, my_lse = mine.attention_ref(q, k, v, **kw, **dkw, return_lse=True)
            my_pt, _ = mine.attention_ref(q, k, v, **kw, **dkw, upcast=False, reorder_ops=True,
                                          intermediate_dtype=torch.float8_e4m3fn)
            my_out32, _ = mine.attention_ref(q.float(), k.float(), v.float(), **kw, **dkw)
            e1 = (my_out32 - ref_out32).abs().max().item()
            e2 = (my_pt.float() - ref_pt.float()).abs().max().item()
            # (the FA3 oracle multiplies q by softmax_scale where the FA2 one, which the restatement follows, divides by
            #  sqrt(d): fp32 rounding noise on outputs of magnitude ~4)
            assert e1 <= 5e-6 and e2 <= 1.6e-2 and (my_out.float() - ref_out.float()).abs().max().item() <= 1.6e-2, (name, e1, e2)
            stride = c.get("store_row_stride", 1)
            golden[name] = {
                "case": {k2: (list(v2) if isinstance(v2, tuple) else v2) for k2, v2 in c.items()},
                "input_checksum": torch.tensor([checksum(q), checksum(k), checksum(v)], dtype=torch.float64),
                "out_ref_fp32": ref_out32[:, ::stride].contiguous(),
                "out_pt": ref_pt[:, ::stride].contiguous(),
                "lse": my_lse[:, :, ::stride].contiguous(),
            }
            print(f"{name:34s} fp32 err {e1:.2e}  pt err {e2:.1e}  (FA3 oracle, descales, e4m3 P)")
            continue
        slopes = make_alibi_slopes(c)
        if slopes is not None:
            bias = ref_alibi(slopes, c["sq"], c["sk"], qm, km, causal=c["causal"])
            assert torch.equal(bias, mine.attn_bias_from_alibi_slopes(slopes, c["sq"], c["sk"], qm, km, causal=c["causal"])), name
            kw["attn_bias"] = bias
        # the reference, three ways: fp32 ("out_ref"), low-precision reordered ("out_pt"), FA3 flavour
        ref_out, ref_attn = fa2.attention_ref(q, k, v, qm, km, **kw)
        ref_pt, _ = fa2.attention_ref(q, k, v, qm, km, **kw, upcast=False, reorder_ops=True)
        ref3_out, _ = fa3.attention_ref(q, k, v, qm, km, **kw)
        ref_out32, _ = fa2.attention_ref(q.float(), k.float(), v.float(), qm, km, **kw)
        # this repo's restatement
        my_out, my_attn, my_lse = mine.attention_ref(q, k, v, qm, km, **kw, return_lse=True)
        my_pt, _ = mine.attention_ref(q, k, v, qm, km, **kw, upcast=False, reorder_ops=True)
        my_out32, _ = mine.attention_ref(q.float(), k.float(), v.float(), qm, km, **kw)
        e1 = (my_out32 - ref_out32).abs().max().item()
        e2 = (my_pt.float() - ref_pt.float()).abs().max().item()
        e3 = (my_out.float() - ref3_out.float()).abs().max().item()
        e4 = (my_attn.float() - ref_attn.float()).abs().max().item()
        assert e1 <= 1e-6, (name, "fp32 restatement vs reference", e1)
        assert e2 == 0.0, (name, "low-precision reordered restatement vs reference", e2)
        assert torch.equal(my_out, ref_out), (name, "cast output differs")
        assert e3 <= (1e-6 if q.dtype == torch.float32 else 8e-3), (name, "vs FA3-flavour oracle", e3)
        assert e4 <= 1e-6 if q.dtype == torch.float32 else e4 <= 4e-3, (name, "attention probs", e4)
        worst = max(worst, e1)
        # C fp64 restatement (dense cases only: no padding masks)
        if qm is None and km is None and slopes is None:
            scale = q.shape[-1] ** -0.5
            c_out, c_lse = crun(q, k, v, scale, c["causal"], c["window"], c["softcap"])
            e5 = (c_out - ref_out32).abs().max().item()
            assert e5 <= 1e-5, (name, "C oracle (fp64) vs reference (fp32)", e5)
            fin = torch.isfinite(my_lse)
            assert torch.equal(fin, torch.isfinite(c_lse)), (name, "lse inf pattern")
            e6 = (c_lse[fin] - my_lse[fin]).abs().max().item() if fin.any() else 0.0
            assert e6 <= 2e-5, (name, "lse torch vs C", e6)
        stride = c.get("store_row_stride", 1)
        golden[name] = {
            "case": {k2: (list(v2) if isinstance(v2, tuple) else v2) for k2, v2 in c.items()},
            "input_checksum": torch.tensor([checksum(q), checksum(k), checksum(v)], dtype=torch.float64),
            "out_ref_fp32": ref_out32[:, ::stride].contiguous(),   # reference, fp32 math on upcast inputs
            "out_pt": ref_pt[:, ::stride].contiguous(),            # reference in low precision, reordered
            "lse": my_lse[:, :, ::stride].contiguous(),            # restatement (checked against the C oracle)
        }
        print(f"{name:34s} fp32 err {e1:.2e}  pt err {e2:.1e}  fa3 err {e3:.2e}")
    out_path = os.path.join(ROOT, "tests/golden/attention_ref_golden.pt")
    torch.save(golden, out_path)
    print(f"wrote {out_path} ({os.path.getsize(out_path) / 1e6:.2f} MB), worst fp32 deviation {worst:.2e}")

    # ---- FA3-only forward arguments: attention_chunk (construct_chunk_mask hopper/test_util.py:193-223) and a V head dim of
    #      its own (:245-246, :284-285), pinned to the FA3 oracle called the way hopper/test_flash_attn.py:163-181 does ----
    fa3_golden = {}
    for name, c in FA3_CASES.items():
        q, k, v = make_inputs(c)
        qm, km = padding_masks(c)
        kw = dict(causal=c["causal"], window_size=c["window"], softcap=c["softcap"], attention_chunk=c["chunk"])
        ref_out, ref_attn = fa3.attention_ref(q, k, v, qm, km, **kw)
        ref_pt, _ = fa3.attention_ref(q, k, v, qm, km, **kw, upcast=False, reorder_ops=True)
        ref_out32, _ = fa3.attention_ref(q.float(), k.float(), v.float(), qm, km, **kw)
        my_out, my_attn, my_lse = mine.attention_ref(q, k, v, qm, km, **kw, return_lse=True)
        my_pt, _ = mine.attention_ref(q, k, v, qm, km, **kw, upcast=False, reorder_ops=True)
        my_out32, _ = mine.attention_ref(q.float(), k.float(), v.float(), qm, km, **kw)
        assert tuple(ref_out.shape) == (c["b"], c["sq"], c["h"], c["dv"]), name
        e1 = (my_out32 - ref_out32).abs().max().item()
        e2 = (my_pt.float() - ref_pt.float()).abs().max().item()
        e4 = (my_attn.float() - ref_attn.float()).abs().max().item()
        # (q * scale there, q / sqrt(d) here: fp32 rounding noise; in 16 bits the two roundings of q differ by an ulp)
        assert e1 <= 5e-6 and e2 <= 1.6e-2 and e4 <= 4e-3, (name, e1, e2, e4)
        if c["chunk"] > 0:  # the mask itself, bit for bit
            want = fa3.construct_chunk_mask(c["sq"], c["sk"], c["chunk"], qm, km)
            assert torch.equal(want, mine.chunk_mask(c["sq"], c["sk"], c["chunk"], qm, km)), (name, "chunk mask")
        stride = c.get("store_row_stride", 1)
        fa3_golden[name] = {
            "case": {k2: (list(v2) if isinstance(v2, tuple) else v2) for k2, v2 in c.items()},
            "input_checksum": torch.tensor([checksum(q), checksum(k), checksum(v)], dtype=torch.float64),
            "out_ref_fp32": ref_out32[:, ::stride].contiguous(),
            "out_pt": ref_pt[:, ::stride].contiguous(),
            "lse": my_lse[:, :, ::stride].contiguous(),
        }
        print(f"{name:34s} fp32 err {e1:.2e}  pt err {e2:.1e}  probs {e4:.1e}  (FA3 oracle: chunk {c['chunk']}, dv {c['dv']})")
    out_path = os.path.join(ROOT, "tests/golden/attention_fa3_golden.pt")
    torch.save(fa3_golden, out_path)
    print(f"wrote {out_path} ({os.path.getsize(out_path) / 1e6:.2f} MB)")

    # ---- left-padded keys (key_leftpad of tests/test_util.py:150-182 and tests/test_flash_attn.py:29-56): restatement
    #      == reference, mask and ALiBi bias
    gen = torch.Generator().manual_seed(99)
    ql = torch.randn(2, 5, 2, 32, generator=gen).to(torch.bfloat16)
    kl = torch.randn(2, 40, 2, 32, generator=gen).to(torch.bfloat16)
    vl = torch.randn(2, 40, 2, 32, generator=gen).to(torch.bfloat16)
    lens, lpad = torch.tensor([33, 40]), torch.tensor([7, 0], dtype=torch.int32)
    kmask = (torch.arange(40).view(1, -1) < lens.view(-1, 1)) & (torch.arange(40).view(1, -1) >= lpad.view(-1, 1))
    sl = torch.rand(2, 2, generator=gen) * 0.3
    for causal, window in ((True, (-1, -1)), (False, (6, 2)), (False, (-1, -1))):
        b_ref = ref_alibi(sl, 5, 40, None, kmask, causal=causal, key_leftpad=lpad)
        b_my = mine.attn_bias_from_alibi_slopes(sl, 5, 40, None, kmask, causal=causal, key_leftpad=lpad)
        assert torch.equal(b_ref, b_my), ("alibi leftpad", causal)
        for bias in (None, b_ref):
            want = fa2.attention_ref(ql, kl, vl, None, kmask, bias, causal=causal, window_size=window, key_leftpad=lpad)[0]
            got = mine.attention_ref(ql, kl, vl, None, kmask, attn_bias=bias, causal=causal, window_size=window,
                                     key_leftpad=lpad)[0]
            assert torch.equal(want, got), ("leftpad", causal, window, bias is not None)
    print("key_leftpad restatement == reference (mask + ALiBi, 6 variants)")

    # ---- rotary embedding: the restatement equals the reference's pure-torch apply_rotary_emb_torch
    #      (flash_attn/layers/rotary.py:22-36) on fp32 inputs, positions = per-batch offsets (+ row) ------------------
    from flash_attn.layers.rotary import apply_rotary_emb_torch  # noqa: E402  (reference package, stub extension)
    gen = torch.Generator().manual_seed(77)
    for interleaved in (False, True):
        for per_row in (True, False):
            x = torch.randn(3, 5, 2, 64, generator=gen)
            ang = torch.rand(40, 16, generator=gen) * 6.28
            cos, sin = torch.cos(ang), torch.sin(ang)
            offs = torch.tensor([0, 7, 30], dtype=torch.int32)
            pos = offs.long().view(3, 1) + (torch.arange(5).view(1, 5) if per_row else 0)
            want = apply_rotary_emb_torch(x, cos[pos.expand(3, 5)], sin[pos.expand(3, 5)], interleaved=interleaved)
            got = mine.apply_rotary_emb_ref(x, cos, sin, offs, interleaved=interleaved, per_row_positions=per_row)
            assert torch.equal(want, got), ("rotary restatement", interleaved, per_row)
    print("rotary restatement == reference apply_rotary_emb_torch (fp32, 4 variants)")

    # ---- backward fixtures: autograd through the reference oracle (tests/test_flash_attn.py:1071-1105) ----------
    grads = {}
    for name, c in GRAD_CASES.items():
        q, k, v = make_inputs(c)
        g = make_grad_output(c)
        qm, km = padding_masks(c)
        kw = dict(causal=c["causal"], window_size=c["window"], softcap=c["softcap"])
        slopes = make_alibi_slopes(c)
        if slopes is not None:
            kw["attn_bias"] = ref_alibi(slopes, c["sq"], c["sk"], qm, km, causal=c["causal"])

        def run(fn, **extra):
            ql, kl, vl = (t.clone().requires_grad_(True) for t in (q, k, v))
            out = fn(ql, kl, vl, qm, km, **kw, **extra)[0]
            return (out.detach(),) + torch.autograd.grad(out, (ql, kl, vl), g)
        # The FA2 oracle soft-caps in place (tests/test_util.py:235-238), which autograd rejects; the FA3 oracle
        # (hopper/test_util.py:294-295, what hopper/test_flash_attn.py differentiates) does not.
        ref_fn = fa3.attention_ref if c["softcap"] > 0 else fa2.attention_ref
        ref = run(ref_fn)
        pt = run(ref_fn, upcast=False, reorder_ops=True)
        my = run(mine.attention_ref)
        my_pt = run(mine.attention_ref, upcast=False, reorder_ops=True)
        for a, b_, what in zip(ref + pt, my + my_pt, ("out", "dq", "dk", "dv") * 2):
            if c["softcap"] > 0:  # FA3 oracle: q * scale instead of q / sqrt(d) -> 16-bit rounding flips
                err = (a.float() - b_.float()).abs().max().item()
                assert err <= 2.0 ** -6 * max(1.0, a.float().abs().max().item()), (name, what, err)
            else:
                assert torch.equal(a, b_), (name, what, "restatement autograd differs from the reference's")
        grads[name] = {
            "case": {k2: (list(v2) if isinstance(v2, tuple) else v2) for k2, v2 in c.items()},
            "input_checksum": torch.tensor([checksum(q), checksum(k), checksum(v), checksum(g)], dtype=torch.float64),
            "out_ref": ref[0], "dq_ref": ref[1], "dk_ref": ref[2], "dv_ref": ref[3],
            "out_pt": pt[0], "dq_pt": pt[1], "dk_pt": pt[2], "dv_pt": pt[3],
        }
        errs = [(pt[i].float() - ref[i].float()).abs().max().item() for i in (1, 2, 3)]
        print(f"{name:34s} |d*_pt - d*_ref| = {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}")
    # ---- split-KV merge (hopper/test_flash_attn.py:1105-1114, a module that needs the GPU extension at import: the
    #      one function is compiled out of the syntax tree like attn_bias_from_alibi_slopes) ------------------------
    import ast
    path = os.path.join(REF, "hopper/test_flash_attn.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "attention_combine_ref"]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    gen = torch.Generator().manual_seed(31)
    op = torch.randn(5, 3, 17, 4, 40, generator=gen)
    lp = torch.randn(5, 3, 17, 4, generator=gen)
    lp[2:, :1] = -float("inf")
    lp[:, 2, 3] = -float("inf")  # a row no split saw
    want_o, want_l = ns["attention_combine_ref"](op, lp)
    got_o, got_l = mine.attention_combine_ref(op, lp)
    assert torch.equal(want_o, got_o) and torch.equal(want_l, got_l)
    grads["combine_pin"] = {"out_partial": op, "lse_partial": lp, "out": want_o, "lse": want_l}
    print("attention_combine_ref restatement == reference")

    # ---- S_dmask decoders (tests/test_flash_attn.py:411-463 convert_flash_attn_S_to_softmax, :466-526 normalize_flash_attn_S;
    #      the module needs the GPU extension at import, the two functions are compiled out of its syntax tree).  The block
    #      width comes from _get_block_size_n, which asks the CUDA device: it is handed in here ---------------------------
    import math as _math
    import torch.nn.functional as _F
    from einops import rearrange as _rearrange
    path = os.path.join(REF, "tests/test_flash_attn.py")
    tree = ast.parse(open(path).read(), filename=path)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef)
           and n.name in ("convert_flash_attn_S_to_softmax", "normalize_flash_attn_S")]
    assert len(fns) == 2
    for bn in (32, 64, 128):
        ns = {"torch": torch, "math": _math, "F": _F, "rearrange": _rearrange,
              "construct_local_mask": fa2.construct_local_mask, "_get_block_size_n": lambda *a, _bn=bn: _bn}
        exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
        gen = torch.Generator().manual_seed(900 + bn)
        bq, hq, sq_, sk_, dq_ = 2, 3, 150, 201, 32
        qx = torch.randn(bq, sq_, hq, dq_, generator=gen).to(torch.bfloat16)
        kx = torch.randn(bq, sk_, hq, dq_, generator=gen).to(torch.bfloat16)
        qmx = torch.arange(sq_)[None, :] < torch.tensor([sq_, 97])[:, None]
        kmx = torch.arange(sk_)[None, :] < torch.tensor([130, sk_])[:, None]
        S = torch.randn(bq, hq, 256, 256, generator=gen).to(torch.bfloat16)   # signs and magnitudes: any values do
        for causal, window in ((False, (-1, -1)), (True, (-1, -1)), (False, (40, 17))):
            want = ns["convert_flash_attn_S_to_softmax"](S, sq_, sk_, qmx, kmx, dq_, True, causal=causal, window_size=window)
            got = mine.convert_flash_attn_S_to_softmax(S, sq_, sk_, qmx, kmx, causal=causal, window_size=window)
            assert torch.equal(want, got), ("convert_flash_attn_S_to_softmax", bn, causal, window)
            want_n = ns["normalize_flash_attn_S"](want.abs(), qx, kx, kx, qmx, kmx, None, True, causal=causal, window_size=window)
            got_n = mine.normalize_flash_attn_S(got.abs(), qx, kx, kx, qmx, kmx, None, True, causal=causal, window_size=window,
                                                block_size_n=bn)
            assert torch.equal(want_n, got_n), ("normalize_flash_attn_S", bn, causal, window)
    # the block-width table itself (flash_attn/flash_attn_interface.py:23-46), for a device that is neither sm8x nor sm90
    spec_if = ast.parse(open(os.path.join(REF, "flash_attn/flash_attn_interface.py")).read())
    fn_bs = [n for n in spec_if.body if isinstance(n, ast.FunctionDef) and n.name == "_get_block_size_n"]
    class _Cuda:  # (what torch.cuda.get_device_capability answers on gfx950: major 9, minor != 0)
        @staticmethod
        def get_device_capability(device=None):
            return (9, 5)
    class _Torch:
        cuda = _Cuda
    ns_bs = {"torch": _Torch}
    exec(compile(ast.Module(body=fn_bs, type_ignores=[]), "flash_attn_interface.py", "exec"), ns_bs)
    for hd in (16, 32, 40, 64, 80, 96, 128, 160, 192, 224, 256):
        for drop in (False, True):
            for caus in (False, True):
                assert ns_bs["_get_block_size_n"]("cuda", hd, drop, caus) == mine.sdmask_block_size_n(hd, drop, caus), (hd, drop, caus)
    print("S_dmask decoders and block-width table == reference")

    # ---- dropout (tests/test_util.py:262-269): with the SAME keep-mask the restatement equals the reference, output
    #      and gradients; the case is kept as a fixture (mask included) for tests/test_oracle.py --------------------
    gen = torch.Generator().manual_seed(4242)
    qd = torch.randn(2, 64, 4, 32, generator=gen).to(torch.bfloat16)
    kd = torch.randn(2, 96, 2, 32, generator=gen).to(torch.bfloat16)
    vd = torch.randn(2, 96, 2, 32, generator=gen).to(torch.bfloat16)
    gd = torch.randn(2, 64, 4, 32, generator=gen).to(torch.bfloat16)
    p_drop = 0.17
    keep = torch.rand(2, 4, 64, 96, generator=gen) > p_drop
    for causal in (False, True):
        def run_d(fn, **extra):
            ql, kl, vl = (t.clone().requires_grad_(True) for t in (qd, kd, vd))
            out = fn(ql, kl, vl, None, None, dropout_p=p_drop, dropout_mask=keep, causal=causal, **extra)[0]
            return (out.detach(),) + torch.autograd.grad(out, (ql, kl, vl), gd)
        ref = run_d(fa2.attention_ref)
        pt = run_d(fa2.attention_ref, upcast=False, reorder_ops=True)
        my = run_d(mine.attention_ref)
        my_pt = run_d(mine.attention_ref, upcast=False, reorder_ops=True)
        for a, b_, what in zip(ref + pt, my + my_pt, ("out", "dq", "dk", "dv") * 2):
            assert torch.equal(a, b_), ("dropout", causal, what)
        grads[f"dropout_pin_causal{int(causal)}"] = {
            "q": qd, "k": kd, "v": vd, "g": gd, "keep": keep, "p_dropout": p_drop, "causal": causal,
            "out_ref": ref[0], "dq_ref": ref[1], "dk_ref": ref[2], "dv_ref": ref[3],
            "out_pt": pt[0], "dq_pt": pt[1], "dk_pt": pt[2], "dv_pt": pt[3],
        }
    print("dropout restatement == reference (out + grads, fp32 and 16-bit orders, 2 variants)")

    out_path = os.path.join(ROOT, "tests/golden/attention_grad_golden.pt")
    torch.save(grads, out_path)
    print(f"wrote {out_path} ({os.path.getsize(out_path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()

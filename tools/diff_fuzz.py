#!/usr/bin/env python3
"""Differential check of the generated asm loops against the C++ tile paths: the same random problems (head dims 64 / 96 / 128,
causal / sliding windows / none, GQA, odd lengths) run through the shipped library and through a build with the generated
blocks switched off (-DFA_ABLATE=32 -DFA_BWD_ABLATE=3), outputs compared.  Usage:
    python tools/diff_fuzz.py <ablated .so> [n_cases]"""
import os
import subprocess
import sys

import torch


def cases(n):
    g = torch.Generator().manual_seed(1234)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    out = []
    for i in range(n):
        d = [64, 96, 128, 128][ri(0, 3)]
        hk = ri(1, 3)
        h = hk * [1, 1, 2, 4][ri(0, 3)]
        sq, sk = ri(65, 2300), ri(65, 2300)
        mode = ri(0, 3)
        if mode == 0:
            causal, window = False, (-1, -1)
        elif mode == 1:
            causal, window = True, (-1, -1)
        elif mode == 2:
            causal, window = False, (ri(16, 600), ri(0, 200))
        else:
            causal, window = False, (ri(16, 600), 0)
        out.append(dict(b=ri(1, 3), sq=sq, sk=sk if not causal else max(sk, sq), h=h, hk=hk, d=d, causal=causal, window=window,
                        dtype=[torch.bfloat16, torch.float16][ri(0, 1)], seed=i))
    return out


def run(path, n):
    import flash_attention_annotated_amd as fa
    res = []
    for c in cases(n):
        torch.manual_seed(c["seed"])
        q = torch.randn(c["b"], c["sq"], c["h"], c["d"], dtype=c["dtype"], device="cuda", requires_grad=True)
        k = torch.randn(c["b"], c["sk"], c["hk"], c["d"], dtype=c["dtype"], device="cuda", requires_grad=True)
        v = torch.randn(c["b"], c["sk"], c["hk"], c["d"], dtype=c["dtype"], device="cuda", requires_grad=True)
        out, lse, _ = fa.flash_attn_func(q, k, v, causal=c["causal"], window_size=c["window"], return_attn_probs=True)
        g = torch.randn_like(out)
        dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
        res.append([t.float().cpu() for t in (out, lse, dq, dk, dv)])
    torch.save(res, path)


if __name__ == "__main__":
    if sys.argv[1] == "--run":
        run(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    ablated, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
    env = dict(os.environ)
    subprocess.run([sys.executable, __file__, "--run", "/tmp/fuzz_a.pt", str(n)], check=True, env=env)
    env["FA_FWD_LIB"] = ablated
    subprocess.run([sys.executable, __file__, "--run", "/tmp/fuzz_b.pt", str(n)], check=True, env=env)
    a, b = torch.load("/tmp/fuzz_a.pt"), torch.load("/tmp/fuzz_b.pt")
    bad = 0
    for c, ra, rb in zip(cases(n), a, b):
        errs = []
        for name, x, y in zip(("out", "lse", "dq", "dk", "dv"), ra, rb):
            fin = torch.isfinite(y)
            assert torch.equal(torch.isfinite(x), fin), (c, name, "inf pattern")
            e = (x[fin] - y[fin]).abs().max().item() if fin.any() else 0.0
            tol = 2e-3 if name == "lse" else (2e-2 if c["dtype"] == torch.bfloat16 else 4e-3) * max(1.0, y[fin].abs().max().item() if fin.any() else 1.0)
            if not e <= tol:
                errs.append(f"{name} {e:.3e} > {tol:.1e}")
        if errs:
            bad += 1
            print("MISMATCH", {k_: v_ for k_, v_ in c.items()}, errs)
    print(f"{n} cases, {bad} mismatching")
    sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Generator of flash_attention_annotated_amd/csrc/fa_fwd_loop_d256_gen.h: the steady-state loop of fwd_kernel_d256 (head dims
129 .. 256, head-dim tile 256) as one inline-asm block per element type -- the sibling of tools/gen_fwd_loop.py for the shape
hopper/tile_size.h:20-45 gives its own tiles.

Shape (fa_fwd_kernel_d256.h): workgroup = 4 waves (one per SIMD), BLOCK_M = 128: a wave owns ONE 32-row q-block -- O alone is
8 x 16 = 128 accumulator registers, Q 16 x 4 = 64 -- and the 64-key K/V tile (32 KiB each at 512-byte rows) is consumed as two
32-key half-steps.  With a single q-block there is no second block to overlap with, so the two halves of the pipeline are

    phase 1   MFMA: S(j+1) = K.Q^T            16 k-steps, no VALU (LDS fragment reads and LDS-DMA in the MFMA shadows)
    phase 2   MFMA: O += V(j)^T.P(j)          8 head-dim blocks x 2 k-steps  ||  VALU: exp/sum/pack of S(j+1) -> P(j+1)

Per 32 x 32 scores there are 32 MFMAs here against 16 in the 128-wide loop: the softmax VALU (one pair-slice per two PV MFMAs,
the slice order measured there: profiles/r3_sched_sweep*.txt) leaves this loop matrix-bound.
Running max: stale up to THR as in the 128-wide loop, guarded by the partial row sums of the fresh P (a sum above 2^THR =
some score outgrew the stale max: the block is left at the half-step boundary and the C++ half-step redoes P(j+1) from the
scores, which stay in v[0:15]).  K/V tiles by `buffer_load_dwordx4 ... lds` (rows past the end of the sequence land as
zeros), 2-deep K and V rings: K tile t+2 / V tile t+1 are requested during tile t and have to have landed at its end (one
barrier per tile, `s_waitcnt vmcnt(0)` in front of it); K tiles are staged shifted by 32 keys like everywhere else.

Run:  python tools/gen_fwd_loop_d256.py   (tests/test_gen_loop.py checks the committed header is current)
"""
import os
import sys

D = 256
ROWB = D * 2                # bytes per LDS row
TILE = 64 * ROWB            # 32 KiB
DEFF = 256                  # head dims actually contracted / produced: 256, 192 or 160 (the zero padding of the 512-byte rows skipped)
KSTEPS = DEFF // 16         # k-steps of the score product (16 / 12 / 10)
NSTEP = 2 * (DEFF // 32)    # (db, st) steps of the PV product (16 / 12 / 10)
NPAIRS = 8                  # score pairs per 32 x 32 block and lane
LD = 8                      # LDS-DMA pieces per wave, tile and matrix
FD = 4                      # LDS fragments are fetched FD MFMAs ahead (every MFMA has a fragment of its own here)
RING = FD + 1
MASKED = False              # loop body that masks the fresh scores (key > RA -> -inf: causal / right window / sequence end) before their softmax
LBL = "fd"                  # label prefix of the loop body being emitted (the unmasked and the masked body share one asm statement)
ALIBI = False               # variant that adds the ALiBi bias -slope |row + sk - sq - key| to the fresh scores in place (RREL, aslope)
SOFTCAP = False             # variant that soft-caps the fresh scores in place (s <- tanh(s * pre)) in front of their softmax
ABLATE = 0                  # developer-only timing ablations: 1 no LDS-DMA, 2 no guard, 8 no softmax VALU, 16 no barrier

# ---- register map (arch VGPRs) ----
S = 0                       # 16: S(j+1), raw scores
P0, P1 = 16, 24             # P(j) of even / odd half-steps (8 packed registers each)
KF, VF = 32, 52             # fragment rings, RING x 4 each
KA, VA = 72, 88             # 16 + 16 LDS address registers
KOFF, VOFF = 104, 112       # 8 + 8 LDS-DMA lane offsets
MC, LA, LAS, T0, T1, PS0, PS1, TMP, KBASE, VBASE = 120, 121, 122, 123, 124, 125, 126, 127, 128, 129
RA, NINF = 131, 132         # masked body: this lane's last visible key minus the key base of the next scores (+ 4 per lane half), -32 per half-step; -inf
RREL = 130                  # ALIBI: (row + sk - sq) - (key base of the next scores + 4 (lane >> 5)) of this lane, -32 per half-step
LAST = 129


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


class Emitter:
    def __init__(self, mfma, cvt):
        self.lines = []
        self.mfma = mfma
        self.cvt = cvt
        self.lds_q = []

    def e(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f"{name}:")

    def ds_k(self, dst, ks, off, tag):
        self.e(f"ds_read_b128 {v(dst, 4)}, {v(KA + ks)} offset:{off}")
        self.lds_q.append(tag)

    def ds_v(self, dst, db, st, off, tag):
        for j2 in range(2):
            self.e(f"ds_read_b64_tr_b16 {v(dst + 2 * j2, 2)}, {v(VA + 2 * db + j2)} offset:{off + (16 * st + 8 * j2) * ROWB}")
            self.lds_q.append(tag)

    def wait_for(self, tag):
        idx = [i for i, t in enumerate(self.lds_q) if t == tag]
        if not idx:
            return
        last = idx[-1]
        self.e(f"s_waitcnt lgkmcnt({len(self.lds_q) - 1 - last})")
        self.lds_q = self.lds_q[last + 1:]


def cvt_bf16(dst, t0, t1):
    return [f"v_cvt_pk_bf16_f32 {v(dst)}, {v(t0)}, {v(t1)}"]


def cvt_f16(dst, t0, t1):
    return [f"v_cvt_f16_f32 {v(dst)}, {v(t0)}",
            f"v_cvt_f16_f32_sdwa {v(dst)}, {v(t1)} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"]


def keyoff(i):
    return (i & 3) + 8 * (i >> 2)   # key of accumulator register i inside its 32-key block (+ 4 per lane half)


def pairs_of(sl, nslices):
    """score pairs whose softmax runs in slice `sl` of phase 2 (8 pairs over 8 / 6 / 5 two-MFMA slices)"""
    return [p for p in range(NPAIRS) if p * nslices // NPAIRS == sl]


def gen_half(E, slot, KB, uid):
    """Half-step j of the tile in ring slot `slot` (t = j / 2, KB = j & 1).  Fragment stream: K fragments 0..15 of the scores of
    half-step j+1 (K tile t+1: ring slot slot^1, its half KB), then V^T fragments 0..15 of half-step j (V tile t: slot, half KB),
    then the next half-step's K fragments."""
    p_cur, p_nxt = (P0, P1) if KB == 0 else (P1, P0)
    kb_off = (slot ^ 1) * TILE + KB * 32 * ROWB
    vb_off = slot * TILE + KB * 32 * ROWB
    nk_off = (slot ^ 1) * TILE + 32 * ROWB if KB == 0 else slot * TILE   # K rows of the NEXT half-step's scores
    kdst = slot * TILE              # K tile t+2 over K tile t (read during tile t-1)
    vdst = 2 * TILE + (slot ^ 1) * TILE   # V tile t+1 over V tile t-1
    kf = lambda i: KF + 4 * (i % RING)
    vf = lambda i: VF + 4 * (i % RING)
    mf = E.mfma

    def tag_of(i):
        if i < KSTEPS:
            return ("k", uid, i)
        if i < KSTEPS + NSTEP:
            return ("v", uid, i - KSTEPS)
        return ("k", uid + 1, i - KSTEPS - NSTEP)

    def fetch(i):
        if i < KSTEPS:
            E.ds_k(kf(i), i, kb_off, tag_of(i))
        elif i < KSTEPS + NSTEP:
            t = i - KSTEPS
            E.ds_v(vf(t), t >> 1, t & 1, vb_off, tag_of(i))
        else:
            nks = i - KSTEPS - NSTEP
            E.ds_k(kf(nks), nks, nk_off, tag_of(i))

    E.e(f"; ---- slot {slot} half-step KB={KB}: phase 1 (scores of the next half-step)")
    def pieces_of(i, n):
        """LDS-DMA pieces issued in slice i of n: one per slice while n >= LD, otherwise the LD pieces dealt in order over the n
        slices (DEFF < 128: fewer MFMA slices than pieces); a slice's pieces never straddle an M0 group of four"""
        ps = [i] if (n >= LD and i < LD) else ([] if n >= LD else [p for p in range(LD) if p * n // LD == i])
        assert not ps or ps[0] // 4 == ps[-1] // 4
        return ps
    for ks in range(KSTEPS):
        kp_ = pieces_of(ks, KSTEPS) if KB == 0 else []
        if kp_ and kp_[0] % 4 == 0:
            E.e(f"s_add_u32 m0, %[lds_wave], {kdst + 1024 * kp_[0]}")
        E.wait_for(tag_of(ks))            # (a no-op but for the half-step's first fragment: the previous MFMA's shadow waited)
        c = "0" if ks == 0 else v(S, 16)
        E.e(f"{mf} {v(S, 16)}, {v(kf(ks), 4)}, %[q{ks}], {c}")
        if not (KB == 1 and ks + FD >= KSTEPS + NSTEP):
            fetch(ks + FD)
        if not (ABLATE & 1):
            for pc in kp_:
                E.e(f"buffer_load_dwordx4 {v(KOFF + pc)}, %[kdesc], %[ktile] offen offset:{1024 * (pc % 4)} lds")
        E.wait_for(tag_of(ks + 1))
    if KB == 0:
        E.e("s_add_u32 %[ktile], %[ktile], %[kstep]")
    E.e(f"; ---- slot {slot} half-step KB={KB}: phase 2 (P.V of this half-step, softmax of the fresh scores)")
    valu = not (ABLATE & 8)
    for t in range(NSTEP):
        db, st = t >> 1, t & 1
        prs = pairs_of(t >> 1, NSTEP // 2)   # score pairs of this slice (two MFMAs): the first around them, further ones behind
        pr = prs[0]
        s0, s1 = S + 2 * pr, S + 2 * pr + 1
        direct = (pr == 0)             # the first pair exponentiates straight into the row-sum registers
        t0, t1 = (PS0, PS1) if direct else (T0, T1)
        if KB == 1 and t == NSTEP - FD:
            # tile barrier, FD MFMAs early: every K/V read of this tile has been issued; everything requested during this tile
            # (K tile t+2, V tile t+1) has landed.  Behind it the next tile's first K fragments are fetched under the last MFMAs.
            E.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
            E.lds_q = []
            if not (ABLATE & 16):
                E.e("s_barrier")
        vp_ = pieces_of(t, NSTEP) if KB == 0 else []
        if vp_ and vp_[0] % 4 == 0:
            E.e(f"s_add_u32 m0, %[lds_wave], {vdst + 1024 * vp_[0]}")
        E.wait_for(tag_of(KSTEPS + t))
        E.e(f"{mf} %[oa{db}], {v(vf(t), 4)}, {v(p_cur + 4 * st, 4)}, %[oa{db}]")
        fetch(KSTEPS + t + FD)
        if not (ABLATE & 1):
            for pc in vp_:
                E.e(f"buffer_load_dwordx4 {v(VOFF + pc)}, %[vdesc], %[vtile] offen offset:{1024 * (pc % 4)} lds")
        if not (KB == 1 and t == NSTEP - 1):
            E.wait_for(tag_of(KSTEPS + t + 1))
        if valu:
            if t == 0:
                E.e("s_nop 7")          # the last score MFMA's result -> its first VALU reader (in the shadow of this MFMA)
            # SOFTCAP: tanh(x) = 1 - 2 / (exp(2 x) + 1) on the pair's two scores, in place (cap2 = 2 log2(e) softmax_scale / softcap;
            # fast_tanh of fa_fwd_kernel.h; hopper/utils.h:635-641).  Two interleaved chains: a transcendental's result is never
            # read by the very next instruction.  The first half (mul, exp, +1, rcp) behind the slice's first MFMA, the rest and the
            # pair's softmax behind the second.
            def tanh_a(r0, r1):
                return [f"v_mul_f32 {v(T0)}, %[cap2], {v(r0)}", f"v_mul_f32 {v(T1)}, %[cap2], {v(r1)}",
                        f"v_exp_f32 {v(T0)}, {v(T0)}", f"v_exp_f32 {v(T1)}, {v(T1)}",
                        f"v_add_f32 {v(T0)}, 1.0, {v(T0)}", f"v_add_f32 {v(T1)}, 1.0, {v(T1)}",
                        f"v_rcp_f32 {v(T0)}, {v(T0)}", f"v_rcp_f32 {v(T1)}, {v(T1)}"]
            def tanh_b(r0, r1):
                return [f"v_fma_f32 {v(r0)}, {v(T0)}, -2.0, 1.0", f"v_fma_f32 {v(r1)}, {v(T1)}, -2.0, 1.0"]
            def alibi_ops(r0, r1):  # s -= slope |rel|, rel = RREL - keyoff(register): sub, cvt, fma with |.| per score
                i0 = r0 - S
                return [f"v_subrev_u32 {v(T0)}, {keyoff(i0)}, {v(RREL)}", f"v_subrev_u32 {v(T1)}, {keyoff(i0 + 1)}, {v(RREL)}",
                        f"v_cvt_f32_i32 {v(T0)}, {v(T0)}", f"v_cvt_f32_i32 {v(T1)}, {v(T1)}",
                        f"v_fma_f32 {v(r0)}, |{v(T0)}|, -%[aslope], {v(r0)}", f"v_fma_f32 {v(r1)}, |{v(T1)}|, -%[aslope], {v(r1)}"]
            def mask_ops(r0, r1):   # behind cap and bias, in front of the exponentials: register i is key base + keyoff(i)
                if not MASKED:
                    return []
                out = []
                for r_ in (r0, r1):
                    out += [f"v_cmp_gt_i32 vcc, {keyoff(r_ - S)}, {v(RA)}", f"v_cndmask_b32 {v(r_)}, {v(r_)}, {v(NINF)}, vcc"]
                return out
            if st == 0:
                if ALIBI:
                    for ins in alibi_ops(s0, s1):
                        E.e(ins)
                if SOFTCAP:
                    for ins in tanh_a(s0, s1):
                        E.e(ins)
                else:
                    for ins in mask_ops(s0, s1):
                        E.e(ins)
                    E.e(f"v_fma_f32 {v(t0)}, {v(s0)}, %[csc], -{v(MC)}")
                    E.e(f"v_fma_f32 {v(t1)}, {v(s1)}, %[csc], -{v(MC)}")
                    E.e(f"v_exp_f32 {v(t0)}, {v(t0)}")
            else:
                if SOFTCAP:
                    for ins in tanh_b(s0, s1) + mask_ops(s0, s1):
                        E.e(ins)
                    E.e(f"v_fma_f32 {v(t0)}, {v(s0)}, %[csc], -{v(MC)}")
                    E.e(f"v_fma_f32 {v(t1)}, {v(s1)}, %[csc], -{v(MC)}")
                    E.e(f"v_exp_f32 {v(t0)}, {v(t0)}")
                E.e(f"v_exp_f32 {v(t1)}, {v(t1)}")
                if direct:
                    E.e("s_nop 0")      # (one instruction between the second v_exp and the pack that reads it)
                else:
                    E.e(f"v_add_f32 {v(PS0)}, {v(PS0)}, {v(T0)}")
                    E.e(f"v_add_f32 {v(PS1)}, {v(PS1)}, {v(T1)}")
                for ins in E.cvt(p_nxt + pr, t0, t1):
                    E.e(ins)
                for pr2 in prs[1:]:     # (DEFF 192 / 160: 8 pairs over 6 / 5 slices)
                    if ALIBI:
                        for ins in alibi_ops(S + 2 * pr2, S + 2 * pr2 + 1):
                            E.e(ins)
                    if SOFTCAP:
                        for ins in tanh_a(S + 2 * pr2, S + 2 * pr2 + 1) + tanh_b(S + 2 * pr2, S + 2 * pr2 + 1):
                            E.e(ins)
                    for ins in mask_ops(S + 2 * pr2, S + 2 * pr2 + 1):
                        E.e(ins)
                    E.e(f"v_fma_f32 {v(T0)}, {v(S + 2 * pr2)}, %[csc], -{v(MC)}")
                    E.e(f"v_fma_f32 {v(T1)}, {v(S + 2 * pr2 + 1)}, %[csc], -{v(MC)}")
                    E.e(f"v_exp_f32 {v(T0)}, {v(T0)}")
                    E.e(f"v_exp_f32 {v(T1)}, {v(T1)}")
                    E.e(f"v_add_f32 {v(PS0)}, {v(PS0)}, {v(T0)}")
                    E.e(f"v_add_f32 {v(PS1)}, {v(PS1)}, {v(T1)}")
                    for ins in E.cvt(p_nxt + pr2, T0, T1):
                        E.e(ins)
    if KB == 0:
        E.e("s_add_u32 %[vtile], %[vtile], %[vstep]")
    if ALIBI:  # the next half-step's scores start 32 keys further on
        E.e(f"v_subrev_u32 {v(RREL)}, 32, {v(RREL)}")
    if MASKED:
        E.e(f"v_subrev_u32 {v(RA)}, 32, {v(RA)}")
    E.e("s_add_u32 %[done], %[done], 1")
    if ABLATE & 2:
        return
    # guard: the partial row sums of P(j+1) -- !(ps <= LIM): some score outgrew the stale max (or inf / NaN)
    E.e(f"v_add_f32 {v(TMP)}, {v(PS0)}, {v(PS1)}")
    E.e(f"v_mov_b32 {v(LAS)}, {v(LA)}")
    E.e(f"v_cmp_nge_f32 vcc, %[lim], {v(TMP)}")
    E.e(f"v_add_f32 {v(LA)}, {v(LA)}, {v(TMP)}")
    E.e("s_mov_b64 %[redo], vcc")
    E.e(f"s_cbranch_vccnz .Lfd_exit_{'%='}")


def gen_block(mfma, cvt):
    E = Emitter(mfma, cvt)
    u = "%="
    E.e("s_mov_b32 %[m0save], m0")
    E.e("s_mov_b64 %[redo], 0")
    # LDS address registers: K fragment ks = lds0 + (kbase ^ 32 ks); V^T fragment (db, j2) = lds0 + 2 TILE + (vbase ^ (64 db + 32 j2))
    for ks in range(16):
        E.e(f"v_xor_b32 {v(KA + ks)}, {32 * ks}, {v(KBASE)}")
    for db in range(D // 32):
        for j2 in range(2):
            E.e(f"v_xor_b32 {v(VA + 2 * db + j2)}, {64 * db + 32 * j2}, {v(VBASE)}")
    for i in range(16):
        E.e(f"v_add_u32 {v(KA + i)}, %[lds0], {v(KA + i)}")
        E.e(f"v_add_u32 {v(VA + i)}, %[lds0v], {v(VA + i)}")
    # Two loop bodies in ONE asm statement (one register map for hipcc: a second statement around the same O / Q operands made it
    # spill at the block boundaries, measured -6 .. -25 %): the unmasked body and -- entered when %[masked] != 0 -- the body that
    # masks the fresh scores (the diagonal / tail tiles of a wave).  The ALiBi form has the unmasked body only.
    global MASKED, LBL
    bodies = [(False, "fd")] if ALIBI else [(False, "fd"), (True, "fdm")]
    if len(bodies) > 1:
        E.e(f"v_mov_b32 {v(NINF)}, 0xff800000")
        E.e("s_cmp_lg_u32 %[masked], 0")
        E.e(f"s_cbranch_scc1 .Lfdm_entry_{u}")
    for masked, lbl in bodies:
        MASKED, LBL = masked, lbl
        if masked:
            E.label(f".L{lbl}_entry_{u}")
        # entry: the first FD K fragments of the first half-step (KB = 0 of ring slot slot0: K tile t+1 sits in slot slot0 ^ 1)
        E.e("s_cmp_eq_u32 %[slot0], 1")
        E.e(f"s_cbranch_scc1 .L{lbl}_in1_{u}")
        for s_ in range(2):
            if s_:
                E.label(f".L{lbl}_in{s_}_{u}")
            E.lds_q = []
            for i in range(FD):
                E.ds_k(KF + 4 * i, i, (s_ ^ 1) * TILE, ("k", 1000 + 2 * s_, i))
            E.e(f"s_branch .L{lbl}_t{s_}_{u}")
        for s_ in range(2):
            E.label(f".L{lbl}_t{s_}_{u}")
            E.lds_q = [("k", 1000 + 2 * s_, i) for i in range(FD)]
            gen_half(E, s_, 0, 1000 + 2 * s_)
            gen_half(E, s_, 1, 1000 + 2 * s_ + 1)
            E.e("s_sub_u32 %[count], %[count], 1")
            E.e("s_cmp_eq_u32 %[count], 0")
            E.e(f"s_cbranch_scc1 .Lfd_exit_{u}")
            if s_ == 1:
                E.e(f"s_branch .L{lbl}_t0_{u}")
    MASKED, LBL = False, "fd"
    E.label(f".Lfd_exit_{u}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e("s_nop 15")
    E.e("s_nop 7")   # asm MFMA results -> compiler-visible readers
    E.e("s_mov_b32 m0, %[m0save]")
    return E.lines


HEADER = '''// GENERATED by tools/gen_fwd_loop_d256.py -- do not edit; regenerate with `python tools/gen_fwd_loop_d256.py`.
//
// fa::FastLoop256<T>::run: the steady-state loop of fwd_kernel_d256 (head-dim tile 256, one 32-row q-block per wave) as one
// inline-asm block (see the generator's docstring).  Register map (arch VGPRs):
//   v[0:15] S (raw scores of the next half-step)  v[16:23] P of even half-steps  v[24:31] P of odd half-steps
//   v[32:51] K fragment ring  v[52:71] V^T fragment ring  v[72:103] LDS address registers  v[104:119] LDS-DMA lane offsets
//   v120 m c  v121 l  v122 l before the last half-step  v[123:127] temporaries  v128 / v129 lane parts of the fragment addresses
// O (8 x 16) and the Q fragments (16 x 4): AGPR tuples wherever hipcc keeps them.
// `count` tiles are run unless the guard fires: `done` half-steps were completed (odd: the fresh P is in podd), `redo` != 0
// means P of the next half-step must be redone from s with a fresh max (l = l_saved first).
#pragma once

namespace fa {

// DEFF: head dims contracted / produced (256, or 192 / 160: the k-steps and O blocks of the zero padding are skipped)
// SOFTCAP: the fresh scores are soft-capped in place before their softmax (cap2 = 2 log2(e) softmax_scale / softcap); on a guard
// trip the scores the caller redoes P from are already capped.
// ALIBI: the fresh scores get the ALiBi bias in place (aslope = slope / softmax_scale, rrel = this lane's row + sk - sq minus the key
// base of the first half-step's NEXT scores in its lane half; stepped by the block).
// masked != 0 (all but the ALiBi form): the body that gives the fresh scores the causal / right-window / end-of-sequence mask is run
// (key > limit -> -inf; ra = the lane's last visible key minus the key base of the first half-step's NEXT scores in its lane half):
// the diagonal and tail tiles of a wave.  Both bodies live in one asm statement.
template <typename T, int DEFF, bool SOFTCAP = false, bool ALIBI = false> struct FastLoop256;
'''

FUNC = '''template <> struct FastLoop256<%(T)s, %(DEFF)d, %(SC)s, %(AL)s> {
    static __device__ __forceinline__ void run(f32x16 (&oa)[8], u32x4 (&q)[16], f32x16 &s, u32x4 (&peven)[2], u32x4 (&podd)[2],
                                               float &l, float &l_saved, float mc, uint32_t kbase, uint32_t vbase,
                                               const uint32_t (&koff)[8], const uint32_t (&voff)[8], float csc, float lim,
                                               u32x4 kdesc, u32x4 vdesc, uint32_t ktile, uint32_t vtile, uint32_t kstep,
                                               uint32_t vstep, uint32_t lds0, uint32_t lds_wave, int slot0, int &count,
                                               int &done, uint64_t &redo%(caparg)s) {
        uint32_t m0save;
        const uint32_t lds0v = lds0 + %(vregion)d;
        asm volatile(
%(body)s
            : %(accs)s,
              "+{v[0:15]}"(s), "+{v[16:19]}"(peven[0]), "+{v[20:23]}"(peven[1]), "+{v[24:27]}"(podd[0]), "+{v[28:31]}"(podd[1]),
              "+{v%(LA)d}"(l), "+{v%(LAS)d}"(l_saved)%(alout)s,
              [ktile] "+s"(ktile), [vtile] "+s"(vtile), [count] "+s"(count), [done] "+s"(done), [redo] "=&s"(redo),
              [m0save] "=&s"(m0save)
            : "{v%(MC)d}"(mc), "{v%(KBASE)d}"(kbase), "{v%(VBASE)d}"(vbase),
              %(offs)s,
              [csc] "s"(csc), [lim] "s"(lim), [kstep] "s"(kstep), [vstep] "s"(vstep), [kdesc] "s"(kdesc), [vdesc] "s"(vdesc),
              [lds0] "s"(lds0), [lds0v] "s"(lds0v), [lds_wave] "s"(lds_wave), [slot0] "s"(slot0)%(capin)s
            : "memory", "vcc", "scc"%(clobbers)s);
    }
};
'''


def render(lines):
    out = []
    for l in lines:
        out.append(f'            "{l}\\n"' if l.endswith(":") else f'            "{l}\\n\\t"')
    return "\n".join(out)


def main():
    global ABLATE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "flash_attention_annotated_amd", "csrc", "fa_fwd_loop_d256_gen.h")
    if "--ablate" in sys.argv:
        ABLATE = int(sys.argv[sys.argv.index("--ablate") + 1])
    text = HEADER
    join = lambda xs: (",\n              ".join(", ".join(xs[i:i + 4]) for i in range(0, len(xs), 4)))
    accs = join([f'[oa{i}] "+a"(oa[{i}])' for i in range(8)] + [f'[q{i}] "+a"(q[{i}])' for i in range(16)])
    offs = join([f'"{{v{KOFF + i}}}"(koff[{i}])' for i in range(8)] + [f'"{{v{VOFF + i}}}"(voff[{i}])' for i in range(8)])
    bound = set(range(0, 32)) | set(range(KOFF, KOFF + 16)) | {MC, LA, LAS, KBASE, VBASE}
    clob = "".join(f', "v{i}"' for i in range(LAST + 1) if i not in bound)
    global DEFF, KSTEPS, NSTEP, SOFTCAP, ALIBI
    for softcap, alibi in ((False, False), (True, False), (False, True)):
        SOFTCAP, ALIBI = softcap, alibi
        # (DEFF <= 128 exists for softcap / ALiBi only: head dims <= 128 with one of them run this kernel shape, fa_fwd_api.hip variant 4)
        for deff in ((256, 192, 160, 128, 96, 64) if (softcap or alibi) else (256, 192, 160)):
            DEFF, KSTEPS, NSTEP = deff, deff // 16, 2 * (deff // 32)
            for T, mf, cvt in (("__bf16", "v_mfma_f32_32x32x16_bf16", cvt_bf16), ("_Float16", "v_mfma_f32_32x32x16_f16", cvt_f16)):
                capin = (', [cap2] "s"(cap2)' if softcap else "") + (', [aslope] "s"(aslope)' if alibi else ', [masked] "s"(masked)')
                caparg = (", float cap2" if softcap else "") + (", float aslope, int rrel" if alibi else ", int masked, int ra")
                text += "\n" + FUNC % {"T": T, "DEFF": deff, "body": render(gen_block(mf, cvt)), "accs": accs, "offs": offs,
                                       "clobbers": clob + ("" if alibi else f', "v{NINF}"'), "vregion": 2 * TILE, "LA": LA, "LAS": LAS, "MC": MC, "KBASE": KBASE,
                                       "VBASE": VBASE, "SC": "true" if softcap else "false", "AL": "true" if alibi else "false",
                                       "caparg": caparg, "capin": capin,
                                       "alout": f', "+{{v{RREL}}}"(rrel)' if alibi else f', "+{{v{RA}}}"(ra)'}
    text += "\n}  // namespace fa\n"
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    if "--out" in sys.argv:
        path = sys.argv[sys.argv.index("--out") + 1]
    open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generator of flash_attention_annotated_amd/csrc/fa_fwd_loop_gen.h: the steady-state tile loop of fwd_kernel_w64 at
head-dim tile 128 as ONE inline-asm block per element type, every register and every issue slot assigned here.

Why generated asm: the wave is alone on its SIMD and issues in order, so whatever sits between two MFMAs is the schedule.
The C++ form of the same loop (fast_half in fa_fwd_kernel_w64.h) spends ~4000 cycles per 64-key tile where the 64 MFMAs
need 2048: hipcc's glue between the hand-written slices (state copies, wave-uniform branches, address arithmetic, M0
save/restore around every LDS-DMA piece, asm-boundary pads) is issued serially with the matrix pipe idle
(profiles/r2_loop_cycles.txt: the loop stripped of ALL VALU / LDS / DMA work still takes ~3000 cycles per tile).  Here a tile is:
64 MFMAs, 32 softmax pair-slices, 48 LDS fragment reads with counted waits, 8 LDS-DMA pieces (2 instructions each), two guard
checks, one barrier, 6 scalar instructions of loop control.

Dataflow = fast_half's (see fa_fwd_kernel_w64.h): per 32-key half-step two phases
    phase 1   MFMA: S_A(j+1), S_B(j+1) = K.Q^T             VALU: exp/sum/pack of S_B(j)       -> P_B(j)
    phase 2   MFMA: O_A += V^T P_A(j), O_B += V^T P_B(j)    VALU: exp/sum/pack of S_A(j+1)     -> P_A(j+1)
with the running maxima kept stale (guards: A by its partial row sums, B by a max look-ahead); any guard trip leaves the
block with the pipeline state the generic C++ half-step expects.  K/V rings, tile shift, DMA distance and the
one-barrier-per-tile protocol are unchanged, so waves may leave the block at different tiles (causal) and keep
rendezvousing with the others from the C++ paths.

Run:  python tools/gen_fwd_loop.py  (writes the header in place; tests/test_gen_loop.py checks the committed file is current)
"""
import os
import sys

D = 128                     # head-dim TILE (LDS rows of 256 B)
ROWB = D * 2
TILE = 64 * ROWB            # bytes of one K or V tile image
DEFF = 128                  # head dims actually contracted / produced: 128, or 96 (k-steps and O blocks of the zero padding skipped)
KSTEPS = DEFF // 16         # k-steps of the QK^T product (8 or 6)
NSTEP = 2 * (DEFF // 32)    # (db, st) steps of the PV product (8 or 6)
NPAIRS = 8                  # score pairs per 32x32 block and lane
MASKED = False              # variant with the causal / right-window / end-of-sequence mask applied to the fresh scores
# PERSIST (FastLoop128P, the persistent kernel): the K / V look-ahead stream runs on into the NEXT work item's first tiles.
# One raw descriptor spans the whole K (V) tensor, so an item is just a byte offset: at the tile step where `kswc` (`vswc`)
# reaches zero the source offset `ktile` (`vtile`) is replaced by `ktile_nx` (`vtile_nx`); 3 scalar instructions per tile each.
PERSIST = False
FIXED_AGPR = False   # (tried: fixed AGPR homes for O / Q in the persistent loop made hipcc's allocation worse, not better)


def keyoff(i):
    return (i & 3) + 8 * (i >> 2)   # key of accumulator register i inside its 32-key block (+ 4 per lane half)


def pairs_of(slice_idx, nslices):
    """score pairs whose softmax runs in slice `slice_idx` of a phase (8 pairs over 8 or 6 slices)"""
    return [p for p in range(NPAIRS) if p * nslices // NPAIRS == slice_idx]
LD = 4                      # LDS-DMA pieces per wave and tile
# developer-only timing ablations (results are WRONG when non-zero; `--ablate N --out path`, never committed):
# 1 no LDS-DMA, 2 no guard checks, 4 no max look-ahead, 8 no softmax VALU, 16 no barrier
ABLATE = 0
# further developer-only ablation bits: 32 no phase-1 softmax VALU, 64 no phase-2 softmax VALU, 128 v_exp -> v_mov,
# 256 no cvt, 512 no row-sum adds, 1024 no fma (the exp reads the raw score)
# SLICE: the order of one slice's instructions (both phases), tokens:
#   w  wait for this slice's LDS fragment   wn  ... for the NEXT slice's   A / B  the two MFMAs (q-blocks A and B)
#   ds fetch of the fragment FD slices ahead   f0 f1 e0 e1 a0 a1 c  the score pair's fma / exp / row-sum adds / pack   la look-ahead max
# Round 2 was "ds w A f0 f1 e0 e1 B a0 a1 c" (2638 cycles per tile at C2).  Round 3 (profiles/r3_sched_sweep*.txt): LDS and wait
# instructions issue for free right behind an MFMA (the MFMA holds the VALU port for 8 cycles, not the LDS / scalar ports),
# anywhere else they push the gap over its 24 cycles of VALU; and one transcendental per gap: 2420 - 2450 cycles, C2 +3.0 - 3.5 %.
SLICE = "A ds f0 f1 e0 B wn e1 a0 a1 c"
FD = 2          # LDS fragments are fetched FD slices ahead of their MFMAs (fragment rings of FD + 1 tuples)
FIRSTPAIR = 1   # 1: the first pair of a phase-2 row sum exponentiates straight into the sum registers (no zeroing, no adds)

# ---- register map (arch VGPRs) ----
SA, SBX, SBY = 0, 16, 32
PAX, PAY, PB = 48, 56, 64
KF, VF = 72, 84             # 3 x 4 each
KA, VA = 96, 104            # 8 + 8 LDS address registers
KOFF, VOFF = 112, 116
MCA, MCB, LA, LB0, LAS, MB = 120, 121, 122, 123, 124, 125
T0, T1, PSA0, PSA1, NXA, NXB, TMP, LB1 = 126, 127, 128, 129, 130, 131, 132, 133
KBASE, VBASE, MBT = 134, 135, 136
RA, RB, NINF = 137, 138, 139   # MASKED variant: per-lane key limits relative to the next half-step's key base, -inf
# ---- AGPRs ----
OA, OB, QA, QB = 0, 64, 128, 160


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


class Emitter:
    def __init__(self, mfma, cvt):
        self.lines = []
        self.mfma = mfma
        self.cvt = cvt          # function (dst, t0, t1) -> list of instructions
        self.lds_q = []         # outstanding LDS reads (tags), oldest first

    def e(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f"{name}:")

    # --- LDS reads with counted waits ---
    def ds_k(self, dst, ks, off, tag):
        self.e(f"ds_read_b128 {v(dst, 4)}, {v(KA + ks)} offset:{off}")
        self.lds_q.append(tag)

    def ds_v(self, dst, db, st, off, tag):
        for j2 in range(2):
            self.e(f"ds_read_b64_tr_b16 {v(dst + 2 * j2, 2)}, {v(VA + 2 * db + j2)} offset:{off + (16 * st + 8 * j2) * ROWB}")
            self.lds_q.append(tag)

    def wait_for(self, tag):
        """s_waitcnt lgkmcnt(N) such that every read tagged `tag` has returned (LDS returns in order)."""
        idx = [i for i, t in enumerate(self.lds_q) if t == tag]
        if not idx:
            return
        last = idx[-1]
        n = len(self.lds_q) - 1 - last
        self.e(f"s_waitcnt lgkmcnt({n})")
        self.lds_q = self.lds_q[last + 1:]

    def wait_all(self):
        self.lds_q = []


def cvt_bf16(dst, t0, t1):
    return [f"v_cvt_pk_bf16_f32 {v(dst)}, {v(t0)}, {v(t1)}"]


def cvt_f16(dst, t0, t1):
    return [f"v_cvt_f16_f32 {v(dst)}, {v(t0)}",
            f"v_cvt_f16_f32_sdwa {v(dst)}, {v(t1)} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"]


def pair_ops(E, phase, s0, s1, mc, acc0, acc1, dst, direct=False, tmp=None):
    """The VALU of one score pair by token: t = exp2(s * c - mc); acc += t; dst = pack(t0, t1).
    direct: the exponentials land in the accumulators themselves (first pair of a fresh sum: no adds)."""
    off = (ABLATE & 8) or (ABLATE & (32 if phase == 1 else 64))
    if off:
        return {k: [] for k in ("f0", "f1", "e0", "e1", "a0", "a1", "c")}
    TT0, TT1 = tmp if tmp else (T0, T1)
    t0, t1 = (acc0, acc1) if direct else (TT0, TT1)
    ex = "v_mov_b32" if ABLATE & 128 else "v_exp_f32"
    ops = {"f0": [f"v_fma_f32 {v(t0)}, {v(s0)}, %[csc], -{v(mc)}"], "f1": [f"v_fma_f32 {v(t1)}, {v(s1)}, %[csc], -{v(mc)}"],
           "e0": [f"{ex} {v(t0)}, {v(t0)}"], "e1": [f"{ex} {v(t1)}, {v(t1)}"],
           "a0": [f"v_add_f32 {v(acc0)}, {v(acc0)}, {v(TT0)}"], "a1": [f"v_add_f32 {v(acc1)}, {v(acc1)}, {v(TT1)}"],
           "c": E.cvt(dst, t0, t1)}
    if ABLATE & 1024:
        ops["f0"], ops["f1"] = [], []
        ops["e0"], ops["e1"] = [f"{ex} {v(t0)}, {v(s0)}"], [f"{ex} {v(t1)}, {v(s1)}"]
    if (ABLATE & 512) or direct:
        ops["a0"], ops["a1"] = [], []
    if direct:
        ops["a0"] = ["s_nop 0"]   # (keeps one instruction between the second v_exp and the pack that reads it)
    if ABLATE & 256:
        ops["c"] = []
    return ops


PAIR_TOKENS = ("f0", "f1", "e0", "e1", "a0", "a1", "c")


def gen_half(E, slot, KB, uid):
    """One 32-key half-step of the tile in ring slot `slot`."""
    sb_cur, sb_nxt = (SBX, SBY) if KB == 0 else (SBY, SBX)
    pa_cur, pa_nxt = (PAX, PAY) if KB == 0 else (PAY, PAX)
    kb_off = ((slot + 1) % 3) * TILE + KB * 32 * ROWB     # K rows of the scores of half-step j+1
    vb_off = slot * TILE + KB * 32 * ROWB                  # V rows of half-step j (VA carries the V region base)
    kdst = slot * TILE                                     # LDS-DMA targets of this tile: K tile n+3 -> K ring slot `slot`,
    vdst = (3 + (slot + 2) % 3) * TILE                     #                               V tile n+2 -> V ring slot (slot+2)%3
    RING = FD + 1
    kf = lambda i: KF + 4 * (i % RING)
    vf = lambda i: VF + 4 * (i % RING)
    mf = E.mfma
    tokens = SLICE.split()
    E.e(f"; ---- slot {slot} half-step KB={KB}: phase 1")
    # fragment stream of a half-step: K fragments 0 .. KSTEPS-1 (phase 1), V^T fragments 0 .. NSTEP-1 (phase 2), then the next
    # half-step's K fragments; slice i of the stream fetches element i + FD
    def fetch(i):
        if i < KSTEPS:
            E.ds_k(kf(i), i, kb_off, ("k", uid, i))
        elif i < KSTEPS + NSTEP:
            t = i - KSTEPS
            E.ds_v(vf(t), t >> 1, t & 1, vb_off, ("v", uid, t))
        else:
            nks = i - KSTEPS - NSTEP  # first FD K fragments of the next half-step
            off = kb_off + 32 * ROWB if KB == 0 else ((slot + 2) % 3) * TILE
            E.ds_k(kf(nks), nks, off, ("k", uid + 1, nks))
    def tag_of(i):
        if i < KSTEPS:
            return ("k", uid, i)
        if i < KSTEPS + NSTEP:
            return ("v", uid, i - KSTEPS)
        return ("k", uid + 1, i - KSTEPS - NSTEP)
    for ks in range(KSTEPS):
        c_a = "0" if ks == 0 else v(SA, 16)
        c_b = "0" if ks == 0 else v(sb_nxt, 16)
        prs = pairs_of(ks, KSTEPS)
        ops = pair_ops(E, 1, sb_cur + 2 * prs[0], sb_cur + 2 * prs[0] + 1, MCB, LB0, LB1, PB + prs[0])
        for tok in tokens:
            if tok == "ds":
                fetch(ks + FD)
            elif tok == "w":
                if KB == 0 and ks == 0:
                    E.e(f"s_add_u32 m0, %[lds_wave], {kdst}")
                E.wait_for(tag_of(ks))
            elif tok == "wn":
                E.wait_for(tag_of(ks + 1))
            elif tok == "A":
                if "w" not in tokens:
                    if KB == 0 and ks == 0:
                        E.e(f"s_add_u32 m0, %[lds_wave], {kdst}")
                    E.wait_for(tag_of(ks))   # (no-op when an earlier `wn` covered it)
                E.e(f"{mf} {v(SA, 16)}, {v(kf(ks), 4)}, %[qa{ks}], {c_a}")
                if PERSIST and KB == 0 and ks == 0:
                    E.e("s_cmp_eq_u32 %[kswc], 0")
                    E.e("s_cselect_b32 %[ktile], %[ktile_nx], %[ktile]")
                    E.e("s_sub_u32 %[kswc], %[kswc], 1")
                if KB == 0 and ks < LD and not (ABLATE & 1):
                    # piece ks of K tile n+3: 1 KiB at M0 + 1024 ks (the instruction offset moves the LDS target AND the source:
                    # the lane offsets carry -1024 ks); rows past the end of the sequence read as zeros (raw buffer, num_records)
                    E.e(f"buffer_load_dwordx4 {v(KOFF + ks)}, %[kdesc], %[ktile] offen offset:{1024 * ks} lds")
            elif tok == "B":
                E.e(f"{mf} {v(sb_nxt, 16)}, {v(kf(ks), 4)}, %[qb{ks}], {c_b}")
            elif tok in PAIR_TOKENS:
                for ins in ops[tok]:
                    E.e(ins)
            elif tok == "la":
                pass
            else:
                raise ValueError(tok)
        for pr in prs[1:]:  # (DEFF = 96: 8 pairs over 6 slices)
            more = pair_ops(E, 1, sb_cur + 2 * pr, sb_cur + 2 * pr + 1, MCB, LB0, LB1, PB + pr)
            for tok in PAIR_TOKENS:
                for ins in more[tok]:
                    E.e(ins)
    if KB == 0:  # K source of the next tile's DMA
        E.e("s_add_u32 %[ktile], %[ktile], %[kstep]")
    E.e(f"; ---- slot {slot} half-step KB={KB}: phase 2")
    valu = not ((ABLATE & 8) or (ABLATE & 64))
    direct = FIRSTPAIR and valu and not (ABLATE & 512)
    if not direct:
        E.e(f"v_mov_b32 {v(PSA0)}, 0")
        E.e(f"v_mov_b32 {v(PSA1)}, 0")
    for t in range(NSTEP):
        db, st = t >> 1, t & 1
        if KB == 1 and t == NSTEP - FD:
            # tile barrier, FD slices early: every K/V read of this tile has been issued; the DMA pieces issued one tile ago
            # (all but this tile's 2 LD youngest) have landed.  Behind it the next tile's first K fragments are fetched
            # under the last slices' MFMAs.
            E.e(f"s_waitcnt vmcnt({2 * LD}) lgkmcnt(0)")
            E.wait_all()
            if not (ABLATE & 16):
                E.e("s_barrier")
        prs = pairs_of(t, NSTEP)
        last = (t == NSTEP - 1) and not (ABLATE & 2)
        ops = pair_ops(E, 2, SA + 2 * prs[0], SA + 2 * prs[0] + 1, MCA, PSA0, PSA1, pa_nxt + prs[0], direct=direct and t == 0)
        # look-ahead max of S_B(j+1) (complete since the end of phase 1), two v_max3 per slice over four slices that end
        # two slices before the guard
        nmax = min(4, NSTEP - 2)          # slices that carry the look-ahead max (16 registers: 4 or 8 per slice)
        m0_ = NSTEP - 2 - nmax
        if MASKED:
            m0_ = max(m0_, 1)  # (registers 4k..4k+3 are masked in slice k: read them from slice k+1 on)
        def lookahead():
            if m0_ <= t < m0_ + nmax and not (ABLATE & 4):
                per = 16 // nmax
                for i in range((t - m0_) * per, (t - m0_ + 1) * per, 4):
                    first = (i == 0)
                    E.e(f"v_max3_f32 {v(NXA)}, {v(sb_nxt + i)}, {v(sb_nxt + i + 1)}, {v(MB) if first else v(NXA)}")
                    E.e(f"v_max3_f32 {v(NXB)}, {v(sb_nxt + i + 2)}, {v(sb_nxt + i + 3)}, {v(MB) if first else v(NXB)}")
        la_done = False
        for tok in tokens:
            if tok == "ds":
                fetch(KSTEPS + t + FD)
            elif tok == "w":
                if KB == 0 and t == 0:
                    E.e(f"s_add_u32 m0, %[lds_wave], {vdst}")
                E.wait_for(tag_of(KSTEPS + t))
            elif tok == "wn":
                if not (KB == 1 and t == NSTEP - 1):   # (the next half-step's entry waits itself: other tile, other tags)
                    E.wait_for(tag_of(KSTEPS + t + 1))
            elif tok == "A":
                if "w" not in tokens:
                    if KB == 0 and t == 0:
                        E.e(f"s_add_u32 m0, %[lds_wave], {vdst}")
                    E.wait_for(tag_of(KSTEPS + t))
                E.e(f"{mf} %[oa{db}], {v(vf(t), 4)}, {v(pa_cur + 4 * st, 4)}, %[oa{db}]")
                if PERSIST and KB == 0 and t == 0:
                    E.e("s_cmp_eq_u32 %[vswc], 0")
                    E.e("s_cselect_b32 %[vtile], %[vtile_nx], %[vtile]")
                    E.e("s_sub_u32 %[vswc], %[vswc], 1")
                if KB == 0 and t < LD and not (ABLATE & 1):
                    E.e(f"buffer_load_dwordx4 {v(VOFF + t)}, %[vdesc], %[vtile] offen offset:{1024 * t} lds")
                if MASKED and t == 0:
                    # mask of S_A(j+1) (complete since the last MFMA A of phase 1), before its first score is exponentiated:
                    # register i is key base + keyoff(i); masked iff keyoff(i) > RA = (last visible key of this lane's row A) - key base
                    for i in range(16):
                        E.e(f"v_cmp_gt_i32 vcc, {keyoff(i)}, {v(RA)}")
                        E.e(f"v_cndmask_b32 {v(SA + i)}, {v(SA + i)}, {v(NINF)}, vcc")
                nmask = 4 if NSTEP >= 6 else 2   # slices that carry the mask of S_B(j+1): 4 (or 8) registers each
                if MASKED and t < nmask:
                    # mask of S_B(j+1), ahead of the look-ahead max that reads each register at least one slice later
                    for i in range((16 // nmask) * t, (16 // nmask) * (t + 1)):
                        E.e(f"v_cmp_gt_i32 vcc, {keyoff(i)}, {v(RB)}")
                        E.e(f"v_cndmask_b32 {v(sb_nxt + i)}, {v(sb_nxt + i)}, {v(NINF)}, vcc")
            elif tok == "B":
                if last:
                    # guard B (look-ahead): some score of S_B(j+1) exceeds m_b + THR / c in ANY lane (each lane half holds its
                    # own 16 keys of the row: no cross-half max needed for a wave-wide "any")
                    E.e(f"v_max_f32 {v(NXA)}, {v(NXA)}, {v(NXB)}")
                    E.e(f"v_cmp_nge_f32 vcc, {v(MBT)}, {v(NXA)}")      # !(m_b + THR / c >= max)
                E.e(f"{mf} %[ob{db}], {v(vf(t), 4)}, {v(PB + 4 * st, 4)}, %[ob{db}]")
                if last:
                    E.e("s_mov_b64 %[bflag], vcc")
            elif tok in PAIR_TOKENS:
                for ins in ops[tok]:
                    E.e(ins)
            elif tok == "la":
                lookahead()
                la_done = True
            else:
                raise ValueError(tok)
        for pr in prs[1:]:
            more = pair_ops(E, 2, SA + 2 * pr, SA + 2 * pr + 1, MCA, PSA0, PSA1, pa_nxt + pr)
            for tok in PAIR_TOKENS:
                for ins in more[tok]:
                    E.e(ins)
        if last:
            E.e(f"v_add_f32 {v(TMP)}, {v(PSA0)}, {v(PSA1)}")
        if not la_done:
            lookahead()
    if KB == 0:
        E.e("s_add_u32 %[vtile], %[vtile], %[vstep]")
    if MASKED:  # the next half-step's scores start 32 keys further on
        E.e(f"v_subrev_u32 {v(RA)}, 32, {v(RA)}")
        E.e(f"v_subrev_u32 {v(RB)}, 32, {v(RB)}")
    # ---- guards (rare exits): guard A = the partial row sums of P_A(j+1) ----
    E.e("s_add_u32 %[done], %[done], 1")
    if ABLATE & 2:
        return
    E.e(f"v_cmp_nge_f32 vcc, %[lim], {v(TMP)}")          # !(ps <= LIM): P_A(j+1) outgrew the stale max (or inf / NaN)
    E.e(f"v_mov_b32 {v(LAS)}, {v(LA)}")
    E.e(f"v_add_f32 {v(LA)}, {v(LA)}, {v(TMP)}")
    E.e("s_mov_b64 %[redo], vcc")
    E.e("s_or_b64 vcc, vcc, %[bflag]")
    E.e(f"s_cbranch_vccnz .Lfa_exit_{'%='}")


def gen_block(mfma, cvt):
    E = Emitter(mfma, cvt)
    u = "%="
    E.e("s_mov_b32 %[m0save], m0")
    # LDS address registers: K fragment ks = lds0 + (kbase ^ 32 ks); V^T fragment (db, j2) = lds0 + 3 TILE + (vbase ^ (64 db + 32 j2))
    for ks in range(KSTEPS):
        E.e(f"v_xor_b32 {v(KA + ks)}, {32 * ks}, {v(KBASE)}")
    for db in range(D // 32):
        for j2 in range(2):
            E.e(f"v_xor_b32 {v(VA + 2 * db + j2)}, {64 * db + 32 * j2}, {v(VBASE)}")
    for i in range(8):
        E.e(f"v_add_u32 {v(KA + i)}, %[lds0], {v(KA + i)}")
        E.e(f"v_add_u32 {v(VA + i)}, %[lds0v], {v(VA + i)}")
    E.e(f"v_mov_b32 {v(LB1)}, 0")
    if MASKED:
        E.e(f"v_mov_b32 {v(NINF)}, 0xff800000")
    # entry: first two K fragments of the first half-step, from K ring slot (slot0 + 1) % 3
    E.e("s_cmp_eq_u32 %[slot0], 1")
    E.e(f"s_cbranch_scc1 .Lfa_in1_{u}")
    E.e("s_cmp_eq_u32 %[slot0], 2")
    E.e(f"s_cbranch_scc1 .Lfa_in2_{u}")
    for s in range(3):
        if s:
            E.label(f".Lfa_in{s}_{u}")
        E.lds_q = []
        for i in range(FD):
            E.ds_k(KF + 4 * i, i, ((s + 1) % 3) * TILE, ("k", 1000 + 2 * s, i))
        E.e(f"s_branch .Lfa_t{s}_{u}")
    uid = 1000
    for s in range(3):
        E.label(f".Lfa_t{s}_{u}")
        E.lds_q = [("k", 1000 + 2 * s, i) for i in range(FD)]
        gen_half(E, s, 0, 1000 + 2 * s)
        gen_half(E, s, 1, 1000 + 2 * s + 1)
        # (the K fragments fetched at the end of KB = 1 carry uid 1000 + 2 s + 2 = the next slot's KB = 0 tags; for slot 2 they
        #  wrap to slot 0's)
        E.e("s_sub_u32 %[count], %[count], 1")
        E.e("s_cmp_eq_u32 %[count], 0")
        E.e(f"s_cbranch_scc1 .Lfa_exit_{u}")
        if s == 2:
            E.e(f"s_branch .Lfa_t0_{u}")
    E.label(f".Lfa_exit_{u}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e(f"v_add_f32 {v(LB0)}, {v(LB0)}, {v(LB1)}")
    E.e("s_nop 15")
    E.e("s_nop 7")   # asm MFMA results -> compiler-visible readers
    E.e("s_mov_b32 m0, %[m0save]")
    return E.lines


HEADER = '''// GENERATED by tools/gen_fwd_loop.py -- do not edit; regenerate with `python tools/gen_fwd_loop.py`.
//
// fa::FastLoop128<T>::run: the steady-state tile loop of fwd_kernel_w64 (head-dim tile 128) as one inline-asm block with
// every register assigned (see the generator's docstring for the why and the dataflow).  Register map:
//   O_A / O_B (8 x 16) and the Q fragments (16 x 4): AGPR tuples wherever hipcc keeps them (asm operands)
//   v[0:15] S_A  v[16:31] S_B (even half-steps)  v[32:47] S_B (odd)  v[48:55] P_A (even)  v[56:63] P_A (odd)  v[64:71] P_B
//   v[72:83] K fragment ring  v[84:95] V^T fragment ring  v[96:111] LDS address registers  v[112:119] LDS-DMA lane offsets
//   v120 m_a c  v121 m_b c  v122 l_a  v123 l_b  v124 l_a before the last half-step  v125 m_b  v136 m_b + THR / c
//   v[126:133] temporaries
// On return: `done` half-steps were completed (an odd count leaves the next scores / P_A in sby / pay: the caller's
// to_canonical_after_odd()), `redo` != 0 means P_A of the next half-step must be redone from sa with a fresh max, and the
// LDS-DMA pieces of the last tile started are in flight exactly as after the C++ fast loop.
// K/V tiles arrive by `buffer_load_dwordx4 ... offen lds` through raw buffer descriptors (kdesc / vdesc: base = first row
// of this (batch, kv head), num_records = bytes up to the end of the last valid row): rows past the end of the sequence
// land as zeros, so look-ahead tiles need no clamping; ktile / vtile are the byte offsets of the next tiles to fetch
// (soffset), koff / voff the lane offsets minus 1024 per piece (the instruction offset that steps the LDS target also
// enters the source address; tools/probe_bufload.hip, profiles/r2_probe_bufload.txt).
#pragma once

namespace fa {

template <typename T, int DEFF, bool MASKED = false> struct FastLoop128;
template <typename T, bool MASKED = false> struct FastLoop128P;  // DEFF = 128 with the item switch of the persistent kernel (PERSIST in the generator)
template <typename T, bool MASKED = false> struct FastLoop64;  // head-dim tile 64 (LDS rows of 128 B, 4 k-steps, 2 O blocks, 2 LDS-DMA pieces per wave)
// DEFF: head dims contracted / produced (128, or 96: zero padding skipped).  MASKED: the fresh scores S_A(j+1) / S_B(j+1) get
// the causal / right-window / end-of-sequence mask (key > limit -> -inf; ra / rb = the lane's last visible key minus the key
// base 64 n_min + 32 (j+1) + 4 (lane >> 5) of the first half-step's NEXT scores): the diagonal and tail tiles of a wave.
'''

FUNC = '''template <> struct %(STRUCT)s {
    static __device__ __forceinline__ void run(f32x16 (&oa)[%(NDB)d], f32x16 (&ob)[%(NDB)d], u32x4 (&qa)[%(NKS)d], u32x4 (&qb)[%(NKS)d], f32x16 &sa,
                                               f32x16 &sbx, f32x16 &sby, u32x4 (&pax)[2], u32x4 (&pay)[2], float &l_a, float &l_b,
                                               float &l_a_saved, float mca, float mcb, float m_b, uint32_t kbase, uint32_t vbase,
                                               const uint32_t (&koff)[%(LD)d], const uint32_t (&voff)[%(LD)d], float csc, float thr_c,
                                               float lim, u32x4 kdesc, u32x4 vdesc, uint32_t ktile, uint32_t vtile,
                                               uint32_t kstep, uint32_t vstep, uint32_t lds0, uint32_t lds_wave, int slot0,
                                               int &count, int &done, uint64_t &redo, int ra = 0, int rb = 0%(pargs)s) {
        uint32_t m0save;
        uint64_t bflag;
        const uint32_t lds0v = lds0 + %(vregion)d;
        asm volatile(
%(body)s
            : %(accs)s,
              "+{v[0:15]}"(sa), "+{v[16:31]}"(sbx), "+{v[32:47]}"(sby),
              "+{v[48:51]}"(pax[0]), "+{v[52:55]}"(pax[1]), "+{v[56:59]}"(pay[0]), "+{v[60:63]}"(pay[1]),
              "+{v122}"(l_a), "+{v123}"(l_b), "+{v124}"(l_a_saved),
              [ktile] "+s"(ktile), [vtile] "+s"(vtile),
              [count] "+s"(count), [done] "+s"(done), [redo] "=&s"(redo), [bflag] "=&s"(bflag), [m0save] "=&s"(m0save)%(maskout)s%(pout)s
            : "{v120}"(mca), "{v121}"(mcb), "{v125}"(m_b), "{v136}"(m_b + thr_c), "{v134}"(kbase), "{v135}"(vbase),
              %(offs)s,
              [csc] "s"(csc), [lim] "s"(lim), [kstep] "s"(kstep), [vstep] "s"(vstep),
              [kdesc] "s"(kdesc), [vdesc] "s"(vdesc),
              [lds0] "s"(lds0), [lds0v] "s"(lds0v), [lds_wave] "s"(lds_wave), [slot0] "s"(slot0)%(pin)s
            : "memory", "vcc", "scc"%(clobbers)s);
    }
};
'''


def operands(ndb, nks, ld):
    if PERSIST and FIXED_AGPR:
        # fixed AGPR homes: inside the item loop of the persistent kernel hipcc otherwise moves the accumulators and the Q
        # fragments between AGPR tuples from one asm statement to the next (copies through VGPRs, spills to scratch)
        fx = lambda base, i, n: f'"+{{a[{base + n * i}:{base + n * i + n - 1}]}}"'
        accs = [f'[oa{i}] {fx(OA, i, 16)}(oa[{i}])' for i in range(ndb)] + [f'[ob{i}] {fx(OB, i, 16)}(ob[{i}])' for i in range(ndb)]
        accs += [f'[qa{i}] {fx(QA, i, 4)}(qa[{i}])' for i in range(nks)] + [f'[qb{i}] {fx(QB, i, 4)}(qb[{i}])' for i in range(nks)]
    else:
        accs = [f'[oa{i}] "+a"(oa[{i}])' for i in range(ndb)] + [f'[ob{i}] "+a"(ob[{i}])' for i in range(ndb)]
        accs += [f'[qa{i}] "+a"(qa[{i}])' for i in range(nks)] + [f'[qb{i}] "+a"(qb[{i}])' for i in range(nks)]
    offs = [f'"{{v{KOFF + i}}}"(koff[{i}])' for i in range(ld)] + [f'"{{v{VOFF + i}}}"(voff[{i}])' for i in range(ld)]
    join = lambda xs: (",\n              ".join(", ".join(xs[i:i + 4]) for i in range(0, len(xs), 4)))
    return join(accs), join(offs)


def render(lines):
    out = []
    for l in lines:
        if l.endswith(":"):
            out.append(f'            "{l}\\n"')
        else:
            out.append(f'            "{l}\\n\\t"')
    return "\n".join(out)


def main():
    global ABLATE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "flash_attention_annotated_amd", "csrc", "fa_fwd_loop_gen.h")
    if "--ablate" in sys.argv:
        ABLATE = int(sys.argv[sys.argv.index("--ablate") + 1])
    for name in ("fd", "firstpair"):   # developer-only schedule variants (with --out)
        if f"--{name}" in sys.argv:
            globals()[name.upper()] = int(sys.argv[sys.argv.index(f"--{name}") + 1])
    if "--slice" in sys.argv:
        globals()["SLICE"] = sys.argv[sys.argv.index("--slice") + 1].replace("_", " ")
    if "--out" in sys.argv:
        path = sys.argv[sys.argv.index("--out") + 1]
    global D, ROWB, TILE, LD, DEFF, KSTEPS, NSTEP, MASKED, KF, VF, PERSIST
    if FD == 3:
        KF, VF = 156, 172   # 4 x 4 each
    text = HEADER
    # (head-dim tile, dims contracted, masked variant?)
    configs = [(128, 128, False, False), (128, 96, False, False), (128, 128, True, False), (128, 96, True, False), (64, 64, False, False),
               (64, 64, True, False), (128, 128, False, True), (128, 128, True, True)]
    for d, deff, masked, persist in configs:
        PERSIST = persist
        D, ROWB, TILE, LD = d, d * 2, 64 * d * 2, d // 32
        DEFF, KSTEPS, NSTEP, MASKED = deff, deff // 16, 2 * (deff // 32), masked
        unused = [KOFF + i for i in range(LD, 4)] + [VOFF + i for i in range(LD, 4)]  # (no inputs there at LD = 2)
        clob = "".join(f', "v{i}"' for i in list(range(64, 112)) + unused + list(range(126, 134)) + (list(range(156, 188)) if FD == 3 else []))
        for T, mf, cvt in (("__bf16", "v_mfma_f32_32x32x16_bf16", cvt_bf16), ("_Float16", "v_mfma_f32_32x32x16_f16", cvt_f16)):
            lines = gen_block(mf, cvt)
            accs, offs = operands(d // 32, d // 16, LD)
            struct = (f"FastLoop128<{T}, {deff}, {'true' if masked else 'false'}>" if d == 128 else
                      f"FastLoop64<{T}, {'true' if masked else 'false'}>")
            if persist:
                struct = f"FastLoop128P<{T}, {'true' if masked else 'false'}>"
            text += "\n" + FUNC % {"STRUCT": struct, "NDB": d // 32, "NKS": d // 16, "LD": LD, "accs": accs, "offs": offs,
                                   "body": render(lines), "clobbers": clob + (', "v139"' if masked else ""),
                                   "vregion": 3 * TILE, "maskout": ', "+{v137}"(ra), "+{v138}"(rb)' if masked else "",
                                   "pargs": ", uint32_t ktile_nx = 0, uint32_t vtile_nx = 0, uint32_t kswc = ~0u, uint32_t vswc = ~0u" if persist else "",
                                   "pout": ', [kswc] "+s"(kswc), [vswc] "+s"(vswc)' if persist else "",
                                   "pin": ', [ktile_nx] "s"(ktile_nx), [vtile_nx] "s"(vtile_nx)' if persist else ""}
    text += "\n}  // namespace fa\n"
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generator of flash_attention_annotated_amd/csrc/fa_fwd_loop_fp8_gen.h: the steady-state tile loop of the native fp8
(e4m3) forward kernel fwd_kernel_fp8 (fa_fwd_kernel_fp8.h) as one inline-asm block, the fp8 sibling of
tools/gen_fwd_loop.py (same reasons: one wave per SIMD issues in order; every issue slot is assigned here).

Both products run on v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit block scales (2x the bf16 rate per
clock, tools/mfma_ceiling.hip; operand maps profiles/r2_probe_layouts.txt):
  * S^T = K.Q^T: A = 32 key rows x 64 head-dim bytes (lane (m, h): row m, bytes 32 h..), B = Q^T the same way, two k-steps
    for d = 128.  The 32 rows of score block beta are the keys 32 hm + 16 beta + 4 a + b for MFMA row m = 8 a + 4 hm + b, so
    that lane half h of the accumulator holds keys 32 h + 16 beta + i in register i: the e4m3-rounded probabilities of a
    64-key tile, packed four to a dword in register order, ARE the B operand of the PV product (k slot 32 h + j = key
    32 h + j) -- the role of the fp8 register permutation of hopper/mainloop_fwd_sm90_tma_gmma_ws.hpp:1157-1160.
  * O^T += V^T.P^T over a whole 64-key tile: A = V^T fragment (lane (d, h): V[32 h + j][d], j = 0..31) from the row-major
    V tile through four ds_read_b64_tr_b8 (8 keys x 16 head-dim bytes per 16-lane group each) -- the role of the
    in-smem V transpose of mainloop...hpp:702-739, done by the LDS read itself.
  * P' = exp2(s c - (m c - OFF)) carries the factor 2^OFF (hopper/softmax.h:67-69 uses 8) so that small probabilities stay
    in e4m3's normal range; l carries it too and it cancels in O / l (LSE subtracts OFF ln 2).
Per 64-key tile and wave: 16 MFMAs of 64 cycles (1024 cycles) against 32 softmax pair-slices (~1150 issue cycles): the
loop is VALU-bound, which is why the fp8 path keeps the exact  s c - m c  fused multiply-add form and spends nothing else.

Pipeline (same two phases as the bf16 loop, with the 64-key tile as the step):
    phase 1   MFMA: S_A(n+1), S_B(n+1) = K(n+1).Q^T            VALU: softmax of S_B(n)    -> P_B(n)
    phase 2   MFMA: O_A += V(n)^T P_A(n), O_B += V(n)^T P_B(n)   VALU: softmax of S_A(n+1)  -> P_A(n+1)
Running maxima are kept stale up to THR (P' <= 2^(OFF + THR) <= 2^8 < 448): both q-blocks get a max look-ahead before
their scores are exponentiated.  A trip of A is taken inside the loop (new max for this tile's exponentials, l_a rescaled, the
O_A rescale handed to the caller: `pend`/`alpha_a`); a trip of B leaves the loop at the tile boundary with S_B(n+1)
untouched (`tripb`).  The caller rescales and re-enters.
"""
import os
import sys

D = 128
ROWB = 128                  # bytes per LDS row (1 byte / element)
TILE = 64 * ROWB            # 8 KiB
LD = 2                      # LDS-DMA pieces per wave, tile and matrix

# ---- register map (arch VGPRs) ----
SA, SBX, SBY = 0, 32, 64            # 2 blocks x 16 each
PAX, PAY, PB = 96, 104, 112         # 8 dwords each (32 e4m3 probabilities per lane)
KF, VF = 120, 136                   # double buffers of 8
KA, VA = 168, 172                   # 4 + 8 LDS address registers
KOFF, VOFF = 180, 182
MCA, MCB, LA, LB0, MA, MB = 184, 185, 186, 187, 188, 189
T0, T1, T2, T3, NXA0, NXA1, NXB0, NXB1 = 190, 191, 192, 193, 194, 195, 196, 197
LB1, ONE, MAT, MBT, ALA, KBASE, VBASE, PSA0, PSA1, LAS, STB = 198, 199, 200, 201, 202, 203, 204, 205, 206, 207, 208
MFMA = "v_mfma_scale_f32_32x32x64_f8f6f4"
# developer-only timing ablations (results are WRONG when non-zero; `--ablate N --out path`, never committed):
# 1 no LDS-DMA, 2 no guard exits, 4 no max look-ahead, 8 no softmax VALU, 16 no barrier, 32 no LDS fragment reads
ABLATE = 0
# ORDER: where a slice's LDS fragment fetch and its wait sit (2 = the fetch as in 1, the wait as in 0).  0 (round 2): fetch in front of the slice, wait right in front of the
# MFMA; 1 (round 3, what the bf16 loop measured, profiles/r3_sched_sweep*.txt): the fetch right behind MFMA A and the wait for
# the NEXT slice's fragment right behind MFMA B -- LDS / wait instructions issue for free in the shadow of an MFMA, anywhere
# else they take VALU issue slots of a loop that is VALU-bound.
ORDER = 1
# EXPMIX: 0 = a quad's 4 fma, then its 4 exp, then its 2 packs; 1 = the two pairs of a quad staggered (fma fma exp exp of pair 0 run
# beside fma fma of pair 1, ...) so that transcendental and plain VALU instructions alternate
EXPMIX = 1


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


class Emitter:
    def __init__(self):
        self.lines = []
        self.lds_q = []

    def e(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f"{name}:")

    def ds_k(self, dst, st, off, tag):
        if ABLATE & 32:
            return
        for e_ in range(2):
            self.e(f"ds_read_b128 {v(dst + 4 * e_, 4)}, {v(KA + 2 * st + e_)} offset:{off}")
            self.lds_q.append(tag)

    def ds_v(self, dst, db, off, tag):
        if ABLATE & 32:
            return
        for t in range(4):
            self.e(f"ds_read_b64_tr_b8 {v(dst + 2 * t, 2)}, {v(VA + 2 * db + (t & 1))} offset:{off + 1024 * t}")
            self.lds_q.append(tag)

    def wait_for(self, tag):
        idx = [i for i, t in enumerate(self.lds_q) if t == tag]
        if not idx:
            return
        last = idx[-1]
        self.e(f"s_waitcnt lgkmcnt({len(self.lds_q) - 1 - last})")
        self.lds_q = self.lds_q[last + 1:]


def pair(E, s0, mc, pdst, pidx, sum0, sum1):
    """softmax of scores s0, s0+1 -> two e4m3 bytes of P dword pdst + pidx // 2 (low half for even pidx, high for odd)"""
    E.e(f"v_fma_f32 {v(T0)}, {v(s0)}, %[csc], -{v(mc)}")
    E.e(f"v_fma_f32 {v(T1)}, {v(s0 + 1)}, %[csc], -{v(mc)}")
    E.e(f"v_exp_f32 {v(T0)}, {v(T0)}")
    E.e(f"v_exp_f32 {v(T1)}, {v(T1)}")
    return [f"v_add_f32 {v(sum0)}, {v(sum0)}, {v(T0)}",
            f"v_add_f32 {v(sum1)}, {v(sum1)}, {v(T1)}",
            f"v_cvt_pk_fp8_f32 {v(pdst + pidx // 2)}, {v(T0)}, {v(T1)}" + (" op_sel:[0,0,1]" if pidx & 1 else "")]


def pair2(E, s0, mc, pdst, pidx, sum0, sum1):
    """two pairs interleaved (exp results are consumed two instructions later: transcendental forwarding rule)"""
    if ABLATE & 8:
        return
    T = [(T0, T1), (T2, T3)]
    fma = lambda k: [f"v_fma_f32 {v(T[k][0])}, {v(s0 + 2 * k)}, %[csc], -{v(mc)}", f"v_fma_f32 {v(T[k][1])}, {v(s0 + 2 * k + 1)}, %[csc], -{v(mc)}"]
    exp = lambda k: [f"v_exp_f32 {v(T[k][0])}, {v(T[k][0])}", f"v_exp_f32 {v(T[k][1])}, {v(T[k][1])}"]
    def tail(k):
        p = pidx + k
        return [f"v_add_f32 {v(sum0)}, {v(sum0)}, {v(T[k][0])}", f"v_add_f32 {v(sum1)}, {v(sum1)}, {v(T[k][1])}",
                f"v_cvt_pk_fp8_f32 {v(pdst + p // 2)}, {v(T[k][0])}, {v(T[k][1])}" + (" op_sel:[0,0,1]" if p & 1 else "")]
    if EXPMIX == 0:
        seq = fma(0) + fma(1) + exp(0) + exp(1) + tail(0) + tail(1)
    else:  # transcendental and plain instructions alternate
        f0, f1, e0, e1, t0, t1 = fma(0), fma(1), exp(0), exp(1), tail(0), tail(1)
        seq = [f0[0], f0[1], e0[0], f1[0], e0[1], f1[1], e1[0], t0[0], e1[1], t0[1], t0[2], t1[0], t1[1], t1[2]]
    for ins in seq:
        E.e(ins)


def gen_tile(E, slot, uid):
    """one 64-key tile in ring slot `slot`: state in (SBX, PAX), out (SBX, PAX) again after the copy-free role swap of TWO
    tiles -- so the generated body is two tiles per ring slot pair; here one tile with explicit cur/nxt given by `uid` parity"""
    odd = uid & 1
    sb_cur, sb_nxt = (SBX, SBY) if not odd else (SBY, SBX)
    pa_cur, pa_nxt = (PAX, PAY) if not odd else (PAY, PAX)
    k_off = ((slot + 1) % 3) * TILE            # K tile n+1 (scores of the next tile)
    v_off = slot * TILE                         # V tile n (VA carries the V region base)
    kdst = slot * TILE                          # K tile n+3 -> K ring slot `slot`
    vdst = (3 + (slot + 2) % 3) * TILE          # V tile n+2 -> V ring slot (slot+2)%3
    kf = lambda i: KF + 8 * (i & 1)
    vf = lambda i: VF + 8 * (i & 1)
    E.e(f"; ---- ring slot {slot}, tile parity {odd}: phase 1")
    for q in range(4):
        beta, st = q >> 1, q & 1
        # fragments one slice ahead (a slice is ~300 cycles of softmax VALU; the first K fragment of a tile is fetched by its
        # predecessor / the entry)
        def fetch1():
            if q + 1 < 4:
                E.ds_k(kf(q + 1), (q + 1) & 1, k_off + 2048 * ((q + 1) >> 1), ("k", uid, q + 1))
            else:
                E.ds_v(vf(0), 0, v_off, ("v", uid, 0))
        if ORDER == 0:
            fetch1()
        if q == 0:
            E.e(f"s_add_u32 m0, %[lds_wave], {kdst}")
        E.wait_for(("k", uid, q))     # (ORDER 1: a no-op, the previous slice waited behind its MFMA B; kept for the tile's first slice)
        c_a = "0" if st == 0 else v(SA + 16 * beta, 16)
        c_b = "0" if st == 0 else v(sb_nxt + 16 * beta, 16)
        E.e(f"{MFMA} {v(SA + 16 * beta, 16)}, {v(kf(q), 8)}, %[qa{st}], {c_a}, {v(ONE)}, {v(ONE)} op_sel_hi:[0,0,0]")
        if ORDER >= 1:
            fetch1()
        if q < LD and not (ABLATE & 1):
            E.e(f"buffer_load_dwordx4 {v(KOFF + q)}, %[kdesc], %[ktile] offen offset:{1024 * q} lds")
        pair2(E, sb_cur + 8 * q, MCB, PB, 4 * q, LB0, LB1)
        E.e(f"{MFMA} {v(sb_nxt + 16 * beta, 16)}, {v(kf(q), 8)}, %[qb{st}], {c_b}, {v(ONE)}, {v(ONE)} op_sel_hi:[0,0,0]")
        if ORDER == 1:
            E.wait_for(("k", uid, q + 1) if q + 1 < 4 else ("v", uid, 0))
        pair2(E, sb_cur + 8 * q + 4, MCB, PB, 4 * q + 2, LB0, LB1)
        # look-ahead max of S_A(n+1), block 0 (complete since slice 1: >= 30 instructions ago when read in slices 2, 3)
        if q >= 2 and not (ABLATE & 4):
            for g in range(2):
                i = (q - 2) * 8 + 4 * g
                seed = v(MA) if (q == 2) else v(NXA0 + g)
                E.e(f"v_max3_f32 {v(NXA0 + g)}, {v(SA + i)}, {v(SA + i + 1)}, {seed}")
                E.e(f"v_max3_f32 {v(NXA0 + g)}, {v(SA + i + 2)}, {v(SA + i + 3)}, {v(NXA0 + g)}")
    E.e("s_add_u32 %[ktile], %[ktile], %[kstep]")
    E.e(f"; ---- ring slot {slot}, tile parity {odd}: phase 2")
    E.e(f"v_mov_b32 {v(PSA0)}, 0")
    E.e(f"v_mov_b32 {v(PSA1)}, 0")
    for db in range(4):
        def fetch2():
            if db + 1 < 4:
                E.ds_v(vf(db + 1), db + 1, v_off, ("v", uid, db + 1))
        def tile_barrier():
            # tile barrier in front of the last PV pair: every K/V read of this tile has been issued and is waited for here;
            # the DMA pieces issued one tile ago (all but this tile's 2 LD youngest) have landed.  Behind it the first K
            # fragment of the next tile is fetched under the last PV MFMAs.
            E.e(f"s_waitcnt vmcnt({2 * LD}) lgkmcnt(0)")
            E.lds_q = []
            if not (ABLATE & 16):
                E.e("s_barrier")
        if ORDER == 0:
            if db + 1 < 4:
                fetch2()
            else:
                tile_barrier()
                E.ds_k(kf(0), 0, ((slot + 2) % 3) * TILE, ("k", uid + 1, 0))
        elif db == 3:
            tile_barrier()
        if db == 0:
            E.e(f"s_add_u32 m0, %[lds_wave], {vdst}")
        E.wait_for(("v", uid, db))
        E.e(f"{MFMA} %[oa{db}], {v(vf(db), 8)}, {v(pa_cur, 8)}, %[oa{db}], {v(ONE)}, {v(ONE)} op_sel_hi:[0,0,0]")
        if ORDER >= 1:
            if db + 1 < 4:
                fetch2()
            else:
                E.ds_k(kf(0), 0, ((slot + 2) % 3) * TILE, ("k", uid + 1, 0))
        if db < LD and not (ABLATE & 1):
            E.e(f"buffer_load_dwordx4 {v(VOFF + db)}, %[vdesc], %[vtile] offen offset:{1024 * db} lds")
        if db == 0 and not (ABLATE & 6):
            # look-ahead max of S_A(n+1), block 1 (its last MFMA: slice 3 of phase 1, > 40 instructions ago), then the decision
            for g in range(2):
                i = 16 + 8 * g
                E.e(f"v_max3_f32 {v(NXA0 + g)}, {v(SA + i)}, {v(SA + i + 1)}, {v(NXA0 + g)}")
                E.e(f"v_max3_f32 {v(NXA0 + g)}, {v(SA + i + 2)}, {v(SA + i + 3)}, {v(NXA0 + g)}")
                E.e(f"v_max3_f32 {v(NXA0 + g)}, {v(SA + i + 4)}, {v(SA + i + 5)}, {v(NXA0 + g)}")
                E.e(f"v_max3_f32 {v(NXA0 + g)}, {v(SA + i + 6)}, {v(SA + i + 7)}, {v(NXA0 + g)}")
            E.e(f"v_max_f32 {v(NXA0)}, {v(NXA0)}, {v(NXA1)}")
            E.e(f"v_cmp_nge_f32 vcc, {v(MAT)}, {v(NXA0)}")   # !(m_a + THR / c >= max): some row of A outgrew its stale max
            E.e(f"v_mov_b32 {v(NXA1)}, {v(NXA0)}")
            E.e(f"s_cbranch_vccnz .Lf8_rare_a{uid % 6}_%=")
            E.label(f".Lf8_back_a{uid % 6}_%=")
        pair2(E, SA + 8 * db, MCA, pa_nxt, 4 * db, PSA0, PSA1)
        E.e(f"{MFMA} %[ob{db}], {v(vf(db), 8)}, {v(PB, 8)}, %[ob{db}], {v(ONE)}, {v(ONE)} op_sel_hi:[0,0,0]")
        if ORDER == 1 and db + 1 < 4:
            E.wait_for(("v", uid, db + 1))
        pair2(E, SA + 8 * db + 4, MCA, pa_nxt, 4 * db + 2, PSA0, PSA1)
        # look-ahead max of S_B(n+1) (complete since the end of phase 1)
        if not (ABLATE & 4):
            for g in range(2):
                i = 8 * db + 4 * g
                seed = v(MB) if db == 0 else v(NXB0 + g)
                E.e(f"v_max3_f32 {v(NXB0 + g)}, {v(sb_nxt + i)}, {v(sb_nxt + i + 1)}, {seed}")
                E.e(f"v_max3_f32 {v(NXB0 + g)}, {v(sb_nxt + i + 2)}, {v(sb_nxt + i + 3)}, {v(NXB0 + g)}")
    E.e("s_add_u32 %[vtile], %[vtile], %[vstep]")
    # ---- tile end: l_a, B's decision, exits ----
    E.e(f"v_max_f32 {v(NXB0)}, {v(NXB0)}, {v(NXB1)}")
    E.e(f"v_add_f32 {v(PSA0)}, {v(PSA0)}, {v(PSA1)}")
    E.e(f"v_cmp_nge_f32 vcc, {v(MBT)}, {v(NXB0)}")          # !(m_b + THR / c >= max)
    E.e(f"v_mov_b32 {v(LAS)}, {v(LA)}")                       # l_a before this tile's P_A(n+1) (the caller's phantom last tile)
    E.e(f"v_add_f32 {v(LA)}, {v(LA)}, {v(PSA0)}")
    E.e("s_add_u32 %[done], %[done], 1")
    E.e("s_or_b64 vcc, vcc, %[pend]")
    E.e("s_sub_u32 %[count], %[count], 1")
    if not (ABLATE & 6):
        E.e(f"s_cbranch_vccnz .Lf8_tripb_%=")
    E.e("s_cmp_eq_u32 %[count], 0")
    E.e(f"s_cbranch_scc1 .Lf8_exit_%=")


def gen_block():
    E = Emitter()
    u = "%="
    E.e("s_mov_b32 %[m0save], m0")
    E.e("s_mov_b64 %[pend], 0")
    E.e("s_mov_b64 %[tripb], 0")
    # LDS address registers.  K fragment (st, e) of block 0: lds0 + (kbase ^ (64 st + 16 e)) (+ 2048 per block, + 8192 per
    # ring slot as immediates); V^T fragment (db, t): lds0 + 3 TILE + (vbase ^ 16 (2 db | (t & 1))) (+ 1024 t, + 8192 per slot)
    for st in range(2):
        for e_ in range(2):
            E.e(f"v_xor_b32 {v(KA + 2 * st + e_)}, {64 * st + 16 * e_}, {v(KBASE)}")
    for db in range(4):
        for t1 in range(2):
            E.e(f"v_xor_b32 {v(VA + 2 * db + t1)}, {16 * (2 * db + t1)}, {v(VBASE)}")
    for i in range(4):
        E.e(f"v_add_u32 {v(KA + i)}, %[lds0], {v(KA + i)}")
    for i in range(8):
        E.e(f"v_add_u32 {v(VA + i)}, %[lds0v], {v(VA + i)}")
    E.e(f"v_mov_b32 {v(LB1)}, 0")
    E.e(f"v_mov_b32 {v(ONE)}, 0x7f7f7f7f")
    # pipeline state in: S_B(n) (32 scores) and P_A(n) (8 dwords) come through a per-wave LDS hand-off area (10 x 16 B per
    # lane, chunk c of lane l at STB + 1024 c) instead of 40 fixed-register operands: hipcc then keeps none of it live in
    # registers across the block (with them bound to v[32:63] / v[96:103] it spilled ~70 registers per lane to scratch around
    # every entry)
    for c in range(8):
        E.e(f"ds_read_b128 {v(SBX + 4 * c, 4)}, {v(STB)} offset:{1024 * c}")
    for c in range(2):
        E.e(f"ds_read_b128 {v(PAX + 4 * c, 4)}, {v(STB)} offset:{1024 * (8 + c)}")
    E.e("s_waitcnt lgkmcnt(0)")
    # entry: first K fragment pair of the first tile, from K ring slot (slot0 + 1) % 3.  The body below is unrolled over the
    # three ring slots x the two (SBX/SBY, PAX/PAY) role parities = 6 tiles; the caller enters with parity 0 state, so the
    # entry slot picks one of the three even positions: position p = slot0 handles slot p % 3 with parity p & 1 -> enter at
    # the position with p % 3 == slot0 and p even: p = slot0 if slot0 even else slot0 + 3.
    E.e("s_cmp_eq_u32 %[slot0], 1")
    E.e(f"s_cbranch_scc1 .Lf8_in1_{u}")
    E.e("s_cmp_eq_u32 %[slot0], 2")
    E.e(f"s_cbranch_scc1 .Lf8_in2_{u}")
    entry_pos = {0: 0, 1: 4, 2: 2}
    for s in range(3):
        if s:
            E.label(f".Lf8_in{s}_{u}")
        E.lds_q = []
        E.ds_k(KF, 0, ((s + 1) % 3) * TILE, ("k", 2000 + entry_pos[s], 0))
        E.e(f"s_branch .Lf8_t{entry_pos[s]}_{u}")
    for pos in range(6):
        E.label(f".Lf8_t{pos}_{u}")
        E.lds_q = [("k", 2000 + pos, 0), ("k", 2000 + pos, 0)]
        gen_tile(E, pos % 3, 2000 + pos)
        if pos == 5:
            E.e(f"s_branch .Lf8_t0_{u}")
    # ---- rare: a row of q-block A outgrew its stale max (taken right after the look-ahead, before any score of S_A(n+1) is
    # exponentiated).  New max for every row, alpha for l_a now and for O_A at the caller (pend), then back. ----
    for pos in range(6 if not (ABLATE & 6) else 0):
        E.label(f".Lf8_rare_a{pos}_{u}")
        E.e("s_nop 1")
        E.e(f"v_permlane32_swap_b32 {v(NXA0)}, {v(NXA1)}")
        E.e(f"v_max_f32 {v(NXA0)}, {v(NXA0)}, {v(NXA1)}")      # row max over both lane halves (>= m_a: the chains were seeded with it)
        E.e(f"v_sub_f32 {v(T0)}, {v(MA)}, {v(NXA0)}")
        E.e(f"v_mul_f32 {v(T0)}, %[csc], {v(T0)}")
        E.e(f"v_exp_f32 {v(T0)}, {v(T0)}")                      # alpha = 2^((m_old - m_new) c) <= 1
        E.e(f"v_mov_b32 {v(MA)}, {v(NXA0)}")
        E.e(f"v_mul_f32 {v(ALA)}, {v(ALA)}, {v(T0)}")
        E.e(f"v_mul_f32 {v(LA)}, {v(LA)}, {v(T0)}")
        E.e(f"v_mul_f32 {v(T1)}, %[csc], {v(NXA0)}")
        E.e(f"v_subrev_f32 {v(MCA)}, %[off], {v(T1)}")          # m c - OFF (one scalar source per VALU instruction)
        E.e(f"v_add_f32 {v(MAT)}, %[thr_c], {v(NXA0)}")
        E.e("s_mov_b64 %[pend], -1")
        E.e(f"s_branch .Lf8_back_a{pos}_{u}")
    E.label(f".Lf8_tripb_{u}")
    E.e(f"v_cmp_nge_f32 vcc, {v(MBT)}, {v(NXB0)}")
    E.e("s_mov_b64 %[tripb], vcc")
    E.label(f".Lf8_exit_{u}")
    E.e("s_waitcnt lgkmcnt(0)")
    # pipeline state out, from the buffers of the parity the block stopped at
    E.e("s_bitcmp1_b32 %[done], 0")
    E.e(f"s_cbranch_scc1 .Lf8_out_odd_{u}")
    for c in range(8):
        E.e(f"ds_write_b128 {v(STB)}, {v(SBX + 4 * c, 4)} offset:{1024 * c}")
    for c in range(2):
        E.e(f"ds_write_b128 {v(STB)}, {v(PAX + 4 * c, 4)} offset:{1024 * (8 + c)}")
    E.e(f"s_branch .Lf8_out_done_{u}")
    E.label(f".Lf8_out_odd_{u}")
    for c in range(8):
        E.e(f"ds_write_b128 {v(STB)}, {v(SBY + 4 * c, 4)} offset:{1024 * c}")
    for c in range(2):
        E.e(f"ds_write_b128 {v(STB)}, {v(PAY + 4 * c, 4)} offset:{1024 * (8 + c)}")
    E.label(f".Lf8_out_done_{u}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e(f"v_add_f32 {v(LB0)}, {v(LB0)}, {v(LB1)}")
    E.e("s_nop 15")
    E.e("s_nop 15")
    E.e("s_nop 7")   # asm MFMA results (16 passes) -> compiler-visible readers
    E.e("s_mov_b32 m0, %[m0save]")
    return E.lines


HEADER = '''// GENERATED by tools/gen_fwd_loop_fp8.py -- do not edit; regenerate with `python tools/gen_fwd_loop_fp8.py`.
//
// fa::FastLoopFp8::run: the steady-state tile loop of fwd_kernel_fp8 (e4m3 inputs, head dim 128) as one inline-asm block
// (see the generator's docstring).  Register map (arch VGPRs):
//   v[0:31] S_A (2 blocks x 16)  v[32:63] S_B (even tiles)  v[64:95] S_B (odd)  v[96:103] P_A (even)  v[104:111] P_A (odd)
//   v[112:119] P_B  v[120:135] K fragment double buffer  v[136:151] V^T fragment double buffer  v[168:179] LDS address registers
//   v[180:183] LDS-DMA lane offsets  v184 m_a c - OFF  v185 m_b c - OFF  v186 l_a  v187 l_b  v188 m_a  v189 m_b
//   v[190:199] temporaries / constants  v200 m_a + THR / c  v201 m_b + THR / c  v202 alpha_a handed to the caller
//   v207 l_a before the last tile's P_A (l_a_saved)
// O_A / O_B (8 x 16): AGPR tuples wherever hipcc keeps them; the Q fragments (4 x 8): VGPR tuples of hipcc's choice (the
// block leaves v[152:167] and v[209:255] alone) -- pinned in AGPRs they were spilled to scratch around the boundary code.
// The pipeline state -- S_B(n) (32 scores per lane) and P_A(n) (8 dwords) -- enters and leaves through a per-wave LDS
// hand-off area: chunk c (16 B) of lane l at state_lds + 1024 c, state_lds = area base + 16 l, chunks 0..7 = S_B, 8..9 = P_A.
// The block runs `count` tiles unless a guard fires: it returns at a tile boundary with `done` tiles completed; pend != 0: O_A must be multiplied by alpha_a (l_a and
// m_a are already updated); tripb != 0: q-block B needs a fresh max before S_B of the next tile is exponentiated.
#pragma once

namespace fa {

struct FastLoopFp8 {
    static __device__ __forceinline__ void run(f32x16 (&oa)[4], f32x16 (&ob)[4], const u32x8 (&qa)[2], const u32x8 (&qb)[2],
                                               uint32_t state_lds, float &l_a, float &l_b,
                                               float &m_a, float &alpha_a, float &l_a_saved, float mca, float mcb, float m_b, uint32_t kbase,
                                               uint32_t vbase, const uint32_t (&koff)[2], const uint32_t (&voff)[2], float csc,
                                               float thr_c, float off, u32x4 kdesc, u32x4 vdesc, uint32_t ktile, uint32_t vtile,
                                               uint32_t kstep, uint32_t vstep, uint32_t lds0, uint32_t lds_wave, int slot0,
                                               int &count, int &done, uint64_t &pend, uint64_t &tripb) {
        uint32_t m0save;
        const uint32_t lds0v = lds0 + %(vregion)d;
        asm volatile(
%(body)s
            : [oa0] "+a"(oa[0]), [oa1] "+a"(oa[1]), [oa2] "+a"(oa[2]), [oa3] "+a"(oa[3]),
              [ob0] "+a"(ob[0]), [ob1] "+a"(ob[1]), [ob2] "+a"(ob[2]), [ob3] "+a"(ob[3]),
              "+{v186}"(l_a), "+{v187}"(l_b), "+{v188}"(m_a), "+{v202}"(alpha_a), "=&{v207}"(l_a_saved),
              [ktile] "+s"(ktile), [vtile] "+s"(vtile), [count] "+s"(count), [done] "+s"(done),
              [pend] "=&s"(pend), [tripb] "=&s"(tripb), [m0save] "=&s"(m0save)
            : "{v184}"(mca), "{v185}"(mcb), "{v189}"(m_b), "{v200}"(m_a + thr_c), "{v201}"(m_b + thr_c),
              "{v203}"(kbase), "{v204}"(vbase), "{v208}"(state_lds), "{v180}"(koff[0]), "{v181}"(koff[1]), "{v182}"(voff[0]), "{v183}"(voff[1]),
              [csc] "s"(csc), [thr_c] "s"(thr_c), [off] "s"(off), [kstep] "s"(kstep), [vstep] "s"(vstep),
              [kdesc] "s"(kdesc), [vdesc] "s"(vdesc), [lds0] "s"(lds0), [lds0v] "s"(lds0v), [lds_wave] "s"(lds_wave),
              [slot0] "s"(slot0), [qa0] "v"(qa[0]), [qa1] "v"(qa[1]), [qb0] "v"(qb[0]), [qb1] "v"(qb[1])
            : "memory", "vcc", "scc"%(clobbers)s);
    }
};

}  // namespace fa
'''


def render(lines):
    out = []
    for l in lines:
        if l.endswith(":"):
            out.append(f'            "{l}\\n"')
        else:
            out.append(f'            "{l}\\n\\t"')
    return "\n".join(out)


def main():
    global ABLATE, ORDER, EXPMIX
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "flash_attention_annotated_amd", "csrc", "fa_fwd_loop_fp8_gen.h")
    for name in ("ablate", "order", "expmix"):   # developer-only variants (with --out)
        if f"--{name}" in sys.argv:
            globals()[name.upper()] = int(sys.argv[sys.argv.index(f"--{name}") + 1])
    clob = "".join(f', "v{i}"' for i in list(range(0, 152)) + list(range(168, 180)) + list(range(190, 200)) + [205, 206])
    text = HEADER % {"body": render(gen_block()), "clobbers": clob, "vregion": 3 * TILE}
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    if "--out" in sys.argv:
        path = sys.argv[sys.argv.index("--out") + 1]
    open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")


if __name__ == "__main__":
    main()

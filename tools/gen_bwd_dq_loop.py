#!/usr/bin/env python3
"""Generator of flash_attention_annotated_amd/csrc/fa_bwd_dq_loop_gen.h: the interior of bwd_dq_kernel (head dim 128, two
32-row query blocks per wave, no mask / softcap / alibi / dropout) as ONE inline-asm block per element type: a run of 64-key
K / V tiles, every register and issue slot assigned here (the reasons are tools/gen_fwd_loop.py's).

A tile is 96 MFMAs: per 32-key half kb the scores S^T = K Q^T and dP^T = V dO^T of both row blocks (32 MFMAs), then
dQ^T += K^T dS^T (16 MFMAs).  The pointwise work of a half (144 VALU instructions) runs in the issue slots of the NEXT score
group, which for the second half is the first score group of the next tile -- the loop is software-pipelined across tiles:

    prologue   S/dP(kb0, t0)
    L(t):      LDS-DMA of tile t+1 | S/dP(kb1, t) + pointwise(kb0, t) | dQ(kb0, t) [barrier t two steps before its end]
               last tile? -> tail:  pointwise(kb1, t) | dQ(kb1, t)
               S/dP(kb0, t+1) + pointwise(kb1, t) | dQ(kb1, t) -> L(t+1)

dQ(kb1, t) reads tile t's K after barrier t, so the K / V tiles rotate through THREE LDS slots (tile t+3 replaces tile t and
is requested after barrier t+1); the barrier / LDS-DMA protocol itself is the C++ loop's (tile t+1 requested after barrier
t-1, landed by barrier t), so every wave may run the block or the C++ tile path independently.

Run:  python tools/gen_bwd_dq_loop.py  (tests/test_gen_loop.py checks the committed header is current)
"""
import os
import sys

D = 128
ROWB = D * 2
TILE = 64 * ROWB
NSLOT = 3
VREG = NSLOT * TILE          # V tiles behind the K tiles
KSTEPS = D // 16
NSTEP = 2 * (D // 32)
LD = 4
# FETCH_AFTER: 1 = a step's LDS fragment fetches sit right behind its first MFMA (free in the MFMA's shadow:
# profiles/r3_sched_sweep*.txt), 0 = in front of the step (round 2)
FETCH_AFTER = 1

# ---- arch VGPRs ----
SET = (0, 64)                # per half kb: S nb0, S nb1, dP nb0, dP nb1 (16 each)
DSF = (128, 144)             # per half kb: packed dS of nb0 (8), nb1 (8)
KR, VR, KT = 160, 172, 184   # fragment rings (3 x 4 each)
RA, TA = 196, 204
KOFF, VOFF = 212, 216
LSE2, DSUM = 220, 222        # + nb
KBASE, VBASE = 224, 225
RV = 226                     # row-fragment addresses of the V slots (RA + 3 tiles: the 16-bit DS offset ends at 64 KiB)
NVGPR = 234


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


class Emitter:
    def __init__(self, mfma, cvt):
        self.lines, self.mfma, self.cvt, self.lds_q = [], mfma, cvt, []

    def e(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append(f"{name}:")

    def row_frag(self, ring, idx, ks, off, tag):
        base = RA
        if off >= VREG:
            base, off = RV, off - VREG
        self.e(f"ds_read_b128 {v(ring + 4 * (idx % 3), 4)}, {v(base + ks)} offset:{off}")
        self.lds_q.append(tag)

    def tr_frag(self, idx, t, off, tag):
        db, st = t >> 1, t & 1
        for j2 in range(2):
            self.e(f"ds_read_b64_tr_b16 {v(KT + 4 * (idx % 3) + 2 * j2, 2)}, {v(TA + 2 * db + j2)} "
                   f"offset:{off + (16 * st + 8 * j2) * ROWB}")
            self.lds_q.append(tag)

    def wait_for(self, tag):
        idx = [i for i, t in enumerate(self.lds_q) if t == tag]
        if not idx:
            return
        last = idx[-1]
        self.e(f"s_waitcnt lgkmcnt({min(15, len(self.lds_q) - 1 - last)})")
        self.lds_q = self.lds_q[last + 1:]


def cvt_bf16(dst, t0, t1):
    return [f"v_cvt_pk_bf16_f32 {v(dst)}, {v(t0)}, {v(t1)}"]


def cvt_f16(dst, t0, t1):
    return [f"v_cvt_f16_f32 {v(dst)}, {v(t0)}",
            f"v_cvt_f16_f32_sdwa {v(dst)}, {v(t1)} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"]


def pointwise(E, kb):
    """P = exp2(S c - lse), dS = P (dP - D), packed: both row blocks of half kb (query = lane: LSE / D are one register each).
    (The packed fp32 forms v_pk_fma/add/mul_f32 -- 6 instead of 9 instructions per score pair -- were tried here: in the shadow
    of the MFMAs they are slower, dQ kernel 2.56 -> 3.05 ms at C2, and the results came out wrong although the same sequence is
    exact on its own, tools/probe_pk_hazard.hip: not used.)"""
    items = []
    for nb in range(2):
        s, dp = SET[kb] + 16 * nb, SET[kb] + 32 + 16 * nb
        for pr in range(8):
            i0, i1 = 2 * pr, 2 * pr + 1
            items += [f"v_fma_f32 {v(s + i0)}, {v(s + i0)}, %[csc], -{v(LSE2 + nb)}",
                      f"v_fma_f32 {v(s + i1)}, {v(s + i1)}, %[csc], -{v(LSE2 + nb)}",
                      f"v_exp_f32 {v(s + i0)}, {v(s + i0)}",
                      f"v_exp_f32 {v(s + i1)}, {v(s + i1)}",
                      f"v_sub_f32 {v(dp + i0)}, {v(dp + i0)}, {v(DSUM + nb)}",
                      f"v_sub_f32 {v(dp + i1)}, {v(dp + i1)}, {v(DSUM + nb)}",
                      f"v_mul_f32 {v(dp + i0)}, {v(s + i0)}, {v(dp + i0)}",
                      f"v_mul_f32 {v(dp + i1)}, {v(s + i1)}, {v(dp + i1)}"]
            items += E.cvt(DSF[kb] + 8 * nb + pr, dp + i0, dp + i1)
    return items


class Slots:
    def __init__(self, E, items, first, nslots):
        self.E, self.items, self.first, self.nslots = E, list(items), first, nslots
        self.per = -(-len(self.items) // max(1, nslots - first))

    def slot(self, k):
        if k < self.first:
            return
        n = 0
        while self.items and (n < self.per or k == self.nslots - 1):
            self.E.e(self.items.pop(0))
            n += 1


def group_sdp(E, slot, kb, uid, valu_kb, dma_slot, nxt):
    """S^T / dP^T of half kb of the tile in LDS slot `slot` into register set kb, with pointwise(valu_kb) in its issue slots.
    nxt = ("tr", slot, kb, uid) | ("row", slot, kb, uid): the group that follows (its first two fragments are fetched here)."""
    kb_off, vb_off = slot * TILE + kb * 32 * ROWB, VREG + slot * TILE + kb * 32 * ROWB
    s0, s1, d0, d1 = (SET[kb] + 16 * i for i in range(4))
    mf = E.mfma
    E.e(f"; ---- slot {slot} half {kb}: S^T, dP^T")
    sl = Slots(E, pointwise(E, valu_kb) if valu_kb is not None else [], 2, 4 * KSTEPS)
    if dma_slot is not None:
        E.e(f"s_add_u32 m0, %[lds_wave], {dma_slot * TILE}")
    for ks in range(KSTEPS):
        def fetch():
            if ks + 2 < KSTEPS:
                E.row_frag(KR, ks + 2, ks + 2, kb_off, ("k", uid, ks + 2))
                E.row_frag(VR, ks + 2, ks + 2, vb_off, ("v", uid, ks + 2))
            elif nxt[0] != "row":
                fetch_first(E, nxt, ks + 2 - KSTEPS)
        if not FETCH_AFTER:
            fetch()
        if dma_slot is not None and ks == LD:
            E.e(f"s_add_u32 m0, %[lds_wave], {VREG + dma_slot * TILE}")
        E.wait_for(("k", uid, ks))
        c = (lambda r: "0" if ks == 0 else v(r, 16))
        # LDS-DMA pieces of the next tile: K pieces behind the first MFMA of the first LD k-steps, V pieces behind the following
        # k-steps (one per k-step when there are 2 LD of them, otherwise also behind the third MFMA)
        first, third = [], []
        if dma_slot is not None:
            if ks < LD:
                first = [("k", ks)]
            elif KSTEPS >= 2 * LD:
                first = [("v", ks - LD)]
            else:
                per = -(-LD // (KSTEPS - LD))
                vp = list(range((ks - LD) * per, min(LD, (ks - LD + 1) * per)))
                first, third = [("v", i) for i in vp[:1]], [("v", i) for i in vp[1:]]

        def pieces(lst):
            for kind, i in lst:
                if kind == "k":
                    E.e(f"buffer_load_dwordx4 {v(KOFF + i)}, %[kdesc], %[ktile] offen offset:{1024 * i} lds")
                else:
                    E.e(f"buffer_load_dwordx4 {v(VOFF + i)}, %[vdesc], %[vtile] offen offset:{1024 * i} lds")
        E.e(f"{mf} {v(s0, 16)}, {v(KR + 4 * (ks % 3), 4)}, %[qa{ks}], {c(s0)}")
        if FETCH_AFTER:
            fetch()
        pieces(first)
        sl.slot(4 * ks)
        E.e(f"{mf} {v(s1, 16)}, {v(KR + 4 * (ks % 3), 4)}, %[qb{ks}], {c(s1)}")
        sl.slot(4 * ks + 1)
        E.wait_for(("v", uid, ks))
        E.e(f"{mf} {v(d0, 16)}, {v(VR + 4 * (ks % 3), 4)}, %[ga{ks}], {c(d0)}")
        pieces(third)
        sl.slot(4 * ks + 2)
        E.e(f"{mf} {v(d1, 16)}, {v(VR + 4 * (ks % 3), 4)}, %[gb{ks}], {c(d1)}")
        sl.slot(4 * ks + 3)
        if ks == KSTEPS - 1 and nxt[0] == "row":
            # (prologue only: the next group reads the same rings, whose positions 0 / 1 hold fragments of this group's last
            #  steps until the MFMAs above have issued)
            fetch_first(E, nxt, 0)
            fetch_first(E, nxt, 1)
    if dma_slot is not None:
        E.e("s_add_u32 %[ktile], %[ktile], %[kstep]")
        E.e("s_add_u32 %[vtile], %[vtile], %[vstep]")


def fetch_first(E, nxt, i):
    kind, slot, kb, uid = nxt
    if kind == "row":
        E.row_frag(KR, i, i, slot * TILE + kb * 32 * ROWB, ("k", uid, i))
        E.row_frag(VR, i, i, VREG + slot * TILE + kb * 32 * ROWB, ("v", uid, i))
    elif kind == "tr":
        E.tr_frag(i, i, slot * TILE + kb * 32 * ROWB, ("t", uid, i))


def group_dq(E, slot, kb, uid, barrier, nxt):
    """dQ^T += K^T dS^T from half kb of the tile in LDS slot `slot`."""
    off = slot * TILE + kb * 32 * ROWB
    mf = E.mfma
    E.e(f"; ---- slot {slot} half {kb}: dQ^T")
    E.e("s_nop 1")   # VALU-packed dS -> MFMA operand
    for t in range(NSTEP):
        db, st = t >> 1, t & 1
        if barrier and t == NSTEP - 2:
            # barrier t, two steps early: this wave's LDS-DMA pieces of tile t+1 (requested a tile ago) have landed; behind it
            # every wave's have, and the first fragments of tile t+1 may be fetched
            E.e("s_waitcnt vmcnt(0)")
            E.e("s_barrier")
        def fetch():
            if t + 2 < NSTEP:
                E.tr_frag(t + 2, t + 2, off, ("t", uid, t + 2))
            elif nxt is not None:
                fetch_first(E, nxt, t + 2 - NSTEP)
        if not FETCH_AFTER:
            fetch()
        E.wait_for(("t", uid, t))
        E.e(f"{mf} %[dq{db}], {v(KT + 4 * (t % 3), 4)}, {v(DSF[kb] + 4 * st, 4)}, %[dq{db}]")
        if FETCH_AFTER:
            fetch()
        E.e(f"{mf} %[dq{D // 32 + db}], {v(KT + 4 * (t % 3), 4)}, {v(DSF[kb] + 8 + 4 * st, 4)}, %[dq{D // 32 + db}]")


def gen_block(mfma, cvt):
    E = Emitter(mfma, cvt)
    u = "%="
    E.e("s_mov_b32 %[m0save], m0")
    for ks in range(KSTEPS):
        E.e(f"v_xor_b32 {v(RA + ks)}, {32 * ks}, {v(KBASE)}")
    for db in range(D // 32):
        for j2 in range(2):
            E.e(f"v_xor_b32 {v(TA + 2 * db + j2)}, {64 * db + 32 * j2}, {v(VBASE)}")
    for i in range(8):
        E.e(f"v_add_u32 {v(RA + i)}, %[lds0], {v(RA + i)}")
        E.e(f"v_add_u32 {v(TA + i)}, %[lds0], {v(TA + i)}")
    for i in range(8):
        E.e(f"v_add_u32 {v(RV + i)}, {VREG}, {v(RA + i)}")
    E.e("s_cmp_eq_u32 %[slot0], 1")
    E.e(f"s_cbranch_scc1 .Ldq_p1_{u}")
    E.e("s_cmp_eq_u32 %[slot0], 2")
    E.e(f"s_cbranch_scc1 .Ldq_p2_{u}")
    uid = lambda s, what: 100 * s + what   # what: 0 S/dP(kb0), 1 S/dP(kb1), 2 dQ(kb0), 3 dQ(kb1)
    # ---- prologues: S/dP(kb0) of the first tile, unpipelined ----
    for s in range(NSLOT):
        if s:
            E.label(f".Ldq_p{s}_{u}")
        E.lds_q = []
        for i in range(2):
            fetch_first(E, ("row", s, 0, uid(s, 0)), i)
        group_sdp(E, s, 0, uid(s, 0), None, None, ("row", s, 1, uid(s, 1)))
        E.e(f"s_branch .Ldq_L{s}_{u}")
    # ---- the loop, one copy per LDS slot ----
    tails = []
    for s in range(NSLOT):
        n = (s + 1) % NSLOT
        E.label(f".Ldq_L{s}_{u}")
        E.lds_q = [("k", uid(s, 1), 0), ("v", uid(s, 1), 0), ("k", uid(s, 1), 1), ("v", uid(s, 1), 1)]
        group_sdp(E, s, 1, uid(s, 1), 0, n, ("tr", s, 0, uid(s, 2)))
        group_dq(E, s, 0, uid(s, 2), True, ("row", n, 0, uid(n, 0)))
        E.e("s_sub_u32 %[count], %[count], 1")
        E.e("s_cmp_eq_u32 %[count], 0")
        E.e(f"s_cbranch_scc1 .Ldq_T{s}_{u}")
        tails.append(list(E.lds_q))
        group_sdp(E, n, 0, uid(n, 0), 1, None, ("tr", s, 1, uid(s, 3)))
        group_dq(E, s, 1, uid(s, 3), False, ("row", n, 1, uid(n, 1)))
        if s == NSLOT - 1:
            E.e(f"s_branch .Ldq_L0_{u}")
    # ---- tails: the last tile's second half, nothing to overlap with ----
    for s in range(NSLOT):
        E.label(f".Ldq_T{s}_{u}")
        E.e("s_waitcnt lgkmcnt(0)")   # (the look-ahead fragments of the tile behind the run are dropped)
        E.lds_q = []
        for i in range(2):
            fetch_first(E, ("tr", s, 1, uid(s, 3)), i)
        for ins in pointwise(E, 1):
            E.e(ins)
        group_dq(E, s, 1, uid(s, 3), False, None)
        if s != NSLOT - 1:
            E.e(f"s_branch .Ldq_exit_{u}")
    E.label(f".Ldq_exit_{u}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e("s_nop 15")
    E.e("s_nop 7")
    E.e("s_mov_b32 m0, %[m0save]")
    return E.lines


HEADER = '''// GENERATED by tools/gen_bwd_dq_loop.py -- do not edit; regenerate with `python tools/gen_bwd_dq_loop.py`.
//
// fa::BwdDqLoop128<T>::run: `count` consecutive 64-key K / V tiles through the dQ update of bwd_dq_kernel (head dim 128, two
// 32-row query blocks per wave, nothing to mask) as one inline-asm block, software-pipelined across tiles; see the
// generator's docstring.
//   dQ^T accumulators (8 x 16) and the Q / dO fragments (32 x 4): AGPR tuples wherever hipcc keeps them (asm operands)
//   v[0:127] S^T / dP^T of the two halves   v[128:159] packed dS   v[160:195] fragment rings   v[196:211] LDS address registers
//   v[212:219] LDS-DMA lane offsets   v[220:223] LSE (log2 units) and D of this lane's two rows   v[226:233] V-slot addresses
// Protocol (= the C++ loop's, with three LDS slots): on entry tile t0 is in slot `slot0` and the barrier behind it has been
// passed; on return tile t0 + count is in its slot, its barrier passed (raw buffer descriptors: a tile past the end of the
// sequence lands as zeros and is never read).
#pragma once

namespace fa {

template <typename T> struct BwdDqLoop128;
template <typename T> struct BwdDqLoop96;   // head dims 65..96 on the 128-wide tiles: the zero padding is skipped (6 k-steps, 3 dQ blocks per row block)
template <typename T> struct BwdDqLoop64;   // head dim 64: LDS rows of 128 B, 4 k-steps, 2 dQ blocks per row block, 2 LDS-DMA pieces per wave
'''

FUNC = '''template <> struct BwdDqLoop%(D)d<%(T)s> {
    static __device__ __forceinline__ void run(f32x16 (&dq)[%(NACC)d], const u32x4 (&qa)[%(NKS)d], const u32x4 (&qb)[%(NKS)d], const u32x4 (&ga)[%(NKS)d],
                                               const u32x4 (&gb)[%(NKS)d], float lse_a, float lse_b, float dsum_a, float dsum_b,
                                               uint32_t kbase, uint32_t vbase, const uint32_t (&koff)[%(LD)d],
                                               const uint32_t (&voff)[%(LD)d], float csc, u32x4 kdesc, u32x4 vdesc, uint32_t ktile,
                                               uint32_t vtile, uint32_t kstep, uint32_t vstep, uint32_t lds0, uint32_t lds_wave,
                                               int slot0, int count) {
        uint32_t m0save;
        asm volatile(
%(body)s
            : %(accs)s,
              [ktile] "+s"(ktile), [vtile] "+s"(vtile), [count] "+s"(count), [m0save] "=&s"(m0save)
            : %(frags)s,
              "{v220}"(lse_a), "{v221}"(lse_b), "{v222}"(dsum_a), "{v223}"(dsum_b), "{v224}"(kbase), "{v225}"(vbase),
              %(offs)s,
              [csc] "s"(csc), [kdesc] "s"(kdesc), [vdesc] "s"(vdesc), [kstep] "s"(kstep), [vstep] "s"(vstep),
              [lds0] "s"(lds0), [lds_wave] "s"(lds_wave), [slot0] "s"(slot0)
            : "memory", "scc"%(clobbers)s);
    }
};
'''


def operands(ndb_tile, ndb, nks, ld):
    join = lambda xs: (",\n              ".join(", ".join(xs[i:i + 4]) for i in range(0, len(xs), 4)))
    accs = [f'[dq{nb * ndb_tile + db}] "+a"(dq[{nb * ndb_tile + db}])' for nb in range(2) for db in range(ndb)]
    frags = [f'[{n}{i}] "a"({n}[{i}])' for n in ("qa", "qb", "ga", "gb") for i in range(nks)]
    offs = [f'"{{v{KOFF + i}}}"(koff[{i}])' for i in range(ld)] + [f'"{{v{VOFF + i}}}"(voff[{i}])' for i in range(ld)]
    return join(accs), join(frags), join(offs)


def render(lines):
    out = []
    for l in lines:
        out.append(f'            "{l}\\n"' if l.endswith(":") else f'            "{l}\\n\\t"')
    return "\n".join(out)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "flash_attention_annotated_amd", "csrc", "fa_bwd_dq_loop_gen.h")
    global D, ROWB, TILE, VREG, KSTEPS, NSTEP, LD
    text = HEADER
    for d, deff in ((128, 128), (128, 96), (64, 64)):   # (head-dim tile, head dims contracted / produced)
        D, ROWB, TILE, KSTEPS, NSTEP, LD = d, d * 2, 64 * d * 2, deff // 16, 2 * (deff // 32), d // 32
        VREG = NSLOT * TILE
        unused = [KOFF + i for i in range(LD, 4)] + [VOFF + i for i in range(LD, 4)]  # (no inputs there at LD = 2)
        clob = "".join(f', "v{i}"' for i in list(range(212)) + unused + list(range(RV, NVGPR)))
        accs, frags, offs = operands(d // 32, deff // 32, KSTEPS, LD)
        for T, mf, cvt in (("__bf16", "v_mfma_f32_32x32x16_bf16", cvt_bf16), ("_Float16", "v_mfma_f32_32x32x16_f16", cvt_f16)):
            text += "\n" + FUNC % {"T": T, "D": deff, "NACC": 2 * (d // 32), "NKS": d // 16, "LD": LD, "accs": accs, "frags": frags,
                                   "offs": offs, "body": render(gen_block(mf, cvt)), "clobbers": clob}
    text += "\n}  // namespace fa\n"
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")


if __name__ == "__main__":
    main()

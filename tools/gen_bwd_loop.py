#!/usr/bin/env python3
"""Generator of flash_attention_annotated_amd/csrc/fa_bwd_loop_gen.h: the interior of bwd_dkdv_kernel (head dim 128, one
32-key block per wave, no mask / softcap / alibi / dropout) as ONE inline-asm block per element type -- a run of 64-row query
tiles of one head, every register and every issue slot assigned here (same reasons as tools/gen_fwd_loop.py: the wave is
alone on its SIMD and issues in order; the C++ form serialises MFMA groups, pointwise and LDS latencies: ~5500 cycles per tile
where the 64 MFMAs need 2048).

A tile (64 query rows x this wave's 32 keys) is four groups of 16 MFMAs; the pointwise work of one 32-row block runs in
the issue slots of the NEXT group:
    A   S0 = Q0 K^T, dP0 = dO0 V^T                     + LDS-DMA of the next tile (8 pieces) + its LSE / D row
    B   S1 = Q1 K^T, dP1 = dO1 V^T                     VALU: P0 = exp2(S0 c - lse), dS0 = P0 (dP0 - D)   -> packed
    C   dV^T += dO0^T P0,  dK^T += Q0^T dS0            VALU: P1, dS1
    D   dV^T += dO1^T P1,  dK^T += Q1^T dS1            tile barrier two steps before the end, first fragments of the next tile
LDS layout, swizzle, fragment addressing and the one-barrier-per-tile protocol are bwd_dkdv_kernel's (fa_bwd_kernel.h), so
each wave may enter and leave the block at its own tiles and keep rendezvousing with the others from the C++ path.

Run:  python tools/gen_bwd_loop.py  (writes the header in place; tests/test_gen_loop.py checks the committed file is current)
"""
import os
import sys

D = 128
ROWB = D * 2
TILE = 64 * ROWB            # bytes of one Q or dO tile image (64 rows)
STAT = 4 * TILE             # lse_s[2][64] then dsum_s[2][64] (fp32)
KSTEPS = D // 16
NSTEP = 2 * (D // 32)
LD = 4
# developer-only timing ablations (results are WRONG when non-zero; `--ablate N --out path`, never committed): 8 no pointwise VALU
ABLATE = 0
# FETCH_AFTER: 1 = a step's LDS fragment fetches sit right behind its first MFMA (they issue for free in the MFMA's shadow:
# profiles/r3_sched_sweep*.txt), 0 = in front of the step (round 2)
FETCH_AFTER = 1

# ---- arch VGPRs ----
S0, DP0, S1, DP1 = 0, 16, 32, 48
PF0, DSF0, PF1, DSF1 = 64, 72, 80, 88
QR, GR = 96, 108            # row-fragment rings (3 x 4 each)
GT, QT = 120, 132           # transposed-fragment rings
LSE, DSM = 144, 160
RA, TA = 176, 184           # LDS address registers: row fragments (8 k-steps), transposed fragments (4 db x 2 j2)
QOFF, GOFF = 192, 196
SADDR, STATR, SOFF, SLDS = 200, 201, 202, 203
KBASE, VBASE = 208, 209
NVGPR = 210


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

    def row_frag(self, ring, idx, ks, off, tag):
        self.e(f"ds_read_b128 {v(ring + 4 * (idx % 3), 4)}, {v(RA + ks)} offset:{off}")
        self.lds_q.append(tag)

    def tr_frag(self, ring, idx, t, off, tag):
        db, st = t >> 1, t & 1
        for j2 in range(2):
            self.e(f"ds_read_b64_tr_b16 {v(ring + 4 * (idx % 3) + 2 * j2, 2)}, {v(TA + 2 * db + j2)} "
                   f"offset:{off + (16 * st + 8 * j2) * ROWB}")
            self.lds_q.append(tag)

    def stat_reads(self, c, rb, uid, g4s):
        for g4 in g4s:
            off = c * 256 + (32 * rb + 8 * g4) * 4
            self.e(f"ds_read_b128 {v(LSE + 4 * g4, 4)}, {v(SADDR)} offset:{off}")
            self.lds_q.append(("st", uid, g4, 0))
            self.e(f"ds_read_b128 {v(DSM + 4 * g4, 4)}, {v(SADDR)} offset:{off + 512}")
            self.lds_q.append(("st", uid, g4))

    def wait_for(self, tag):
        idx = [i for i, t in enumerate(self.lds_q) if t == tag]
        if not idx:
            return
        last = idx[-1]
        self.e(f"s_waitcnt lgkmcnt({min(15, len(self.lds_q) - 1 - last)})")
        self.lds_q = self.lds_q[last + 1:]

    def wait_all(self):
        self.lds_q = []


def cvt_bf16(dst, t0, t1):
    return [f"v_cvt_pk_bf16_f32 {v(dst)}, {v(t0)}, {v(t1)}"]


def cvt_f16(dst, t0, t1):
    return [f"v_cvt_f16_f32 {v(dst)}, {v(t0)}",
            f"v_cvt_f16_f32_sdwa {v(dst)}, {v(t1)} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD"]


def pointwise(E, rb, uid, c):
    """VALU work of one 32-row block as a list of items: ("wait", tag) | ("read", ...) | instruction string.  The LSE / D rows
    of the block's first 16 rows are fetched at the group's start (by the caller), the rest from here."""
    s, dp = (S0, DP0) if rb == 0 else (S1, DP1)
    pf, dsf = (PF0, DSF0) if rb == 0 else (PF1, DSF1)
    items = []
    for pr in range(8):
        i0, i1 = 2 * pr, 2 * pr + 1
        if pr % 2 == 0:
            items.append(("wait", ("st", uid, pr >> 1)))
        items += [f"v_fma_f32 {v(s + i0)}, {v(s + i0)}, %[csc], -{v(LSE + i0)}",
                  f"v_fma_f32 {v(s + i1)}, {v(s + i1)}, %[csc], -{v(LSE + i1)}",
                  f"v_exp_f32 {v(s + i0)}, {v(s + i0)}",
                  f"v_exp_f32 {v(s + i1)}, {v(s + i1)}",
                  f"v_sub_f32 {v(dp + i0)}, {v(dp + i0)}, {v(DSM + i0)}",
                  f"v_sub_f32 {v(dp + i1)}, {v(dp + i1)}, {v(DSM + i1)}",
                  f"v_mul_f32 {v(dp + i0)}, {v(s + i0)}, {v(dp + i0)}",
                  f"v_mul_f32 {v(dp + i1)}, {v(s + i1)}, {v(dp + i1)}"]
        items += E.cvt(pf + pr, s + i0, s + i1)
        items += E.cvt(dsf + pr, dp + i0, dp + i1)
        if pr == 0:
            items.append(("read", (c, rb, uid, (2, 3))))
    return items


class Slots:
    """Hands the pointwise items out over the issue slots of a 16-MFMA group (slots `first`..15)."""
    def __init__(self, E, items, first=2, nslots=None):
        nslots = nslots or 2 * KSTEPS
        self.E, self.items, self.first, self.nslots = E, list(items), first, nslots
        n = sum(1 for it in self.items if not isinstance(it, tuple))
        self.per = -(-n // max(1, nslots - first))

    def slot(self, k):
        if k < self.first:
            return
        n = 0
        last = (k == self.nslots - 1)
        while self.items and (n < self.per or last):
            it = self.items.pop(0)
            if isinstance(it, tuple) and it[0] == "read":
                self.E.stat_reads(*it[1])
            elif isinstance(it, tuple):
                self.E.wait_for(it[1])
            else:
                if not (ABLATE & 8):
                    self.E.e(it)
                n += 1


def group_sdp(E, c, rb, uid, valu, dma):
    """S / dP of 32-row block rb of the tile in buffer c.  Ring positions: 8 rb + ks."""
    n = c ^ 1
    qb, gb = c * TILE + rb * 32 * ROWB, (2 + c) * TILE + rb * 32 * ROWB
    s, dp = (S0, DP0) if rb == 0 else (S1, DP1)
    mf = E.mfma
    E.e(f"; ---- buffer {c}: S{rb}, dP{rb}")
    if valu is not None:
        E.stat_reads(c, 0, uid, (0, 1))  # LSE / D rows of block 0 (whose pointwise runs in this group)
        sl = Slots(E, pointwise(E, 0, uid, c))
    else:
        sl = Slots(E, [])
    if dma:
        E.e(f"s_add_u32 m0, %[lds_wave], {n * TILE}")
    for ks in range(KSTEPS):
        pos = KSTEPS * rb + ks

        def fetch():
            if ks + 2 < KSTEPS:
                E.row_frag(QR, pos + 2, ks + 2, qb, ("q", uid, rb, ks + 2))
                E.row_frag(GR, pos + 2, ks + 2, gb, ("g", uid, rb, ks + 2))
            elif rb == 0:   # first fragments of block 1
                nk = ks + 2 - KSTEPS
                E.row_frag(QR, pos + 2, nk, c * TILE + 32 * ROWB, ("q", uid, 1, nk))
                E.row_frag(GR, pos + 2, nk, (2 + c) * TILE + 32 * ROWB, ("g", uid, 1, nk))
            else:           # first transposed fragments of group C (block 0)
                nt = ks + 2 - KSTEPS
                E.tr_frag(GT, nt, nt, (2 + c) * TILE, ("gt", uid, 0, nt))
                E.tr_frag(QT, nt, nt, c * TILE, ("qt", uid, 0, nt))
        if not FETCH_AFTER:
            fetch()
        if dma and ks == LD:
            E.e(f"s_add_u32 m0, %[lds_wave], {(2 + n) * TILE}")
        E.wait_for(("q", uid, rb, ks))
        cs = "0" if ks == 0 else v(s, 16)
        cd = "0" if ks == 0 else v(dp, 16)
        # LDS-DMA pieces of the next tile: Q pieces behind the S MFMAs of the first LD k-steps, dO pieces behind the following
        # MFMAs (one per k-step when there are 2 LD k-steps, otherwise also behind the dP MFMAs)
        after_s, after_dp = [], []
        if dma:
            if ks < LD:
                after_s = [("q", ks)]
            elif KSTEPS >= 2 * LD:
                after_s = [("g", ks - LD)]
            else:
                per = -(-LD // (KSTEPS - LD))      # dO pieces per remaining k-step
                gp = list(range((ks - LD) * per, min(LD, (ks - LD + 1) * per)))
                after_s, after_dp = [("g", i) for i in gp[:1]], [("g", i) for i in gp[1:]]

        def pieces(lst):
            for kind, i in lst:
                if kind == "q":
                    E.e(f"buffer_load_dwordx4 {v(QOFF + i)}, %[qdesc], %[qtile] offen offset:{1024 * i} lds")
                else:
                    E.e(f"buffer_load_dwordx4 {v(GOFF + i)}, %[gdesc], %[gtile] offen offset:{1024 * i} lds")
        E.e(f"{mf} {v(s, 16)}, {v(QR + 4 * (pos % 3), 4)}, %[kf{ks}], {cs}")
        if FETCH_AFTER:
            fetch()
        if dma and ks == 0:
            E.e(f"buffer_load_dword {v(STATR)}, {v(SOFF)}, %[sdesc], %[stile] offen")
        pieces(after_s)
        sl.slot(2 * ks)
        E.wait_for(("g", uid, rb, ks))
        E.e(f"{mf} {v(dp, 16)}, {v(GR + 4 * (pos % 3), 4)}, %[vf{ks}], {cd}")
        pieces(after_dp)
        sl.slot(2 * ks + 1)
    if dma:
        E.e("s_add_u32 %[qtile], %[qtile], %[qstep]")
        E.e("s_add_u32 %[gtile], %[gtile], %[gstep]")
        E.e("s_add_u32 %[stile], %[stile], 256")


def group_dvdk(E, c, rb, uid, valu, last, next_uid=None):
    """dV^T / dK^T updates from 32-row block rb.  Ring positions: 8 rb + t."""
    n = c ^ 1
    gb, qb = (2 + c) * TILE + rb * 32 * ROWB, c * TILE + rb * 32 * ROWB
    pf, dsf = (PF0, DSF0) if rb == 0 else (PF1, DSF1)
    mf = E.mfma
    E.e(f"; ---- buffer {c}: dV, dK from block {rb}")
    if valu:
        E.stat_reads(c, 1, uid + 500, (0, 1))
        sl = Slots(E, pointwise(E, 1, uid + 500, c))
    else:
        sl = Slots(E, [])
    E.e("s_nop 1")   # VALU-packed P / dS -> MFMA operand
    for t in range(NSTEP):
        db, st = t >> 1, t & 1
        pos = NSTEP * rb + t
        if last and t == NSTEP - 2:
            # tile barrier, two steps early: every read of this tile's buffers has been issued; the next tile's LDS-DMA pieces
            # (issued in group A) have long landed; its LSE / D row goes to LDS here
            E.e("s_waitcnt vmcnt(0)")
            E.e(f"v_mul_f32 {v(STATR)}, %[sfac], {v(STATR)}")
            E.e(f"ds_write_b32 {v(SLDS)}, {v(STATR)} offset:{n * 256}")
            E.e("s_waitcnt lgkmcnt(0)")
            E.wait_all()
            E.e("s_barrier")
        def fetch():
            if t + 2 < NSTEP:
                E.tr_frag(GT, pos + 2, t + 2, gb, ("gt", uid, rb, t + 2))
                E.tr_frag(QT, pos + 2, t + 2, qb, ("qt", uid, rb, t + 2))
            elif rb == 0:
                nt = t + 2 - NSTEP
                E.tr_frag(GT, pos + 2, nt, (2 + c) * TILE + 32 * ROWB, ("gt", uid, 1, nt))
                E.tr_frag(QT, pos + 2, nt, c * TILE + 32 * ROWB, ("qt", uid, 1, nt))
            else:           # first row fragments of the next tile (buffer n, block 0): ring positions 0, 1
                nk = t + 2 - NSTEP
                E.row_frag(QR, nk, nk, n * TILE, ("q", next_uid, 0, nk))
                E.row_frag(GR, nk, nk, (2 + n) * TILE, ("g", next_uid, 0, nk))
        if not FETCH_AFTER:
            fetch()
        E.wait_for(("gt", uid, rb, t))
        E.e(f"{mf} %[dv{db}], {v(GT + 4 * (pos % 3), 4)}, {v(pf + 4 * st, 4)}, %[dv{db}]")
        if FETCH_AFTER:
            fetch()
        sl.slot(2 * t)
        E.wait_for(("qt", uid, rb, t))
        E.e(f"{mf} %[dk{db}], {v(QT + 4 * (pos % 3), 4)}, {v(dsf + 4 * st, 4)}, %[dk{db}]")
        sl.slot(2 * t + 1)


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
    E.e("s_cmp_eq_u32 %[buf0], 1")
    E.e(f"s_cbranch_scc1 .Lfb_in1_{u}")
    for c in range(2):
        if c:
            E.label(f".Lfb_in1_{u}")
        E.lds_q = []
        for ks in range(2):
            E.row_frag(QR, ks, ks, c * TILE, ("q", 1000 + c, 0, ks))
            E.row_frag(GR, ks, ks, (2 + c) * TILE, ("g", 1000 + c, 0, ks))
        if c == 0:
            E.e(f"s_branch .Lfb_t0_{u}")
    # (buffer 1's entry falls through to its tile code: emit tile 1 first, then tile 0)
    for c in (1, 0):
        E.label(f".Lfb_t{c}_{u}")
        uid = 1000 + c
        E.lds_q = [("q", uid, 0, 0), ("g", uid, 0, 0), ("q", uid, 0, 1), ("g", uid, 0, 1)]
        group_sdp(E, c, 0, uid, None, True)
        group_sdp(E, c, 1, uid, True, False)
        group_dvdk(E, c, 0, uid, True, False)
        group_dvdk(E, c, 1, uid, False, True, next_uid=1000 + (c ^ 1))
        E.e("s_sub_u32 %[count], %[count], 1")
        E.e("s_cmp_eq_u32 %[count], 0")
        E.e(f"s_cbranch_scc1 .Lfb_exit_{u}")
        if c == 0:
            E.e(f"s_branch .Lfb_t1_{u}")
    E.label(f".Lfb_exit_{u}")
    E.e("s_waitcnt lgkmcnt(0)")
    E.e("s_nop 15")
    E.e("s_nop 7")
    E.e("s_mov_b32 m0, %[m0save]")
    return E.lines


HEADER = '''// GENERATED by tools/gen_bwd_loop.py -- do not edit; regenerate with `python tools/gen_bwd_loop.py`.
//
// fa::BwdLoop128<T>::run: `count` consecutive 64-row query tiles of ONE head through the dK / dV update of bwd_dkdv_kernel
// (head dim 128, one 32-key block per wave, nothing to mask) as one inline-asm block; see the generator's docstring.
//   dK^T / dV^T accumulators (8 x 16) and the K / V fragments (16 x 4): AGPR tuples wherever hipcc keeps them (asm operands)
//   v[0:63] S0 dP0 S1 dP1   v[64:95] packed P0 dS0 P1 dS1   v[96:119] row-fragment rings   v[120:143] transposed-fragment rings
//   v[144:175] LSE / D rows   v[176:191] LDS address registers   v[192:199] LDS-DMA lane offsets   v[200:203] LSE / D staging
// Protocol (= the C++ loop's): on entry the first tile's Q / dO / LSE / D are in LDS buffer `buf0` and the tile barrier behind
// them has been passed; every tile issues this wave's LDS-DMA pieces of the NEXT tile (raw buffer descriptors: rows past the end
// of the sequence land as zeros) and its LSE / D row (waves 0, 2: LSE x log2 e, waves 1, 3: D), and ends in one barrier.  On
// return the tile after the last one computed is in LDS, its barrier passed: the caller continues there.
#pragma once

namespace fa {

template <typename T> struct BwdLoop128;
template <typename T> struct BwdLoop96;   // head dims 65..96 on the 128-wide tiles: the zero padding is skipped (6 k-steps, 3 blocks per accumulator)
template <typename T> struct BwdLoop64;   // head dim 64: LDS rows of 128 B, 4 k-steps, 2 blocks per accumulator, 2 LDS-DMA pieces per wave
'''

FUNC = '''template <> struct BwdLoop%(D)d<%(T)s> {
    static __device__ __forceinline__ void run(f32x16 (&dk)[%(NDB)d], f32x16 (&dv)[%(NDB)d], const u32x4 (&kf)[%(NKS)d], const u32x4 (&vf)[%(NKS)d],
                                               uint32_t kbase, uint32_t vbase, const uint32_t (&qoff)[%(LD)d],
                                               const uint32_t (&goff)[%(LD)d], uint32_t saddr, uint32_t soff, uint32_t slds, float csc,
                                               float sfac, u32x4 qdesc, u32x4 gdesc, u32x4 sdesc, uint32_t qtile, uint32_t gtile,
                                               uint32_t stile, uint32_t qstep, uint32_t gstep, uint32_t lds0, uint32_t lds_wave,
                                               int buf0, int count) {
        uint32_t m0save;
        asm volatile(
%(body)s
            : %(accs)s,
              [qtile] "+s"(qtile), [gtile] "+s"(gtile), [stile] "+s"(stile), [count] "+s"(count), [m0save] "=&s"(m0save)
            : %(frags)s,
              "{v208}"(kbase), "{v209}"(vbase),
              %(offs)s,
              "{v200}"(saddr), "{v202}"(soff), "{v203}"(slds),
              [csc] "s"(csc), [sfac] "s"(sfac), [qdesc] "s"(qdesc), [gdesc] "s"(gdesc), [sdesc] "s"(sdesc),
              [qstep] "s"(qstep), [gstep] "s"(gstep), [lds0] "s"(lds0), [lds_wave] "s"(lds_wave), [buf0] "s"(buf0)
            : "memory", "scc"%(clobbers)s);
    }
};
'''


def operands(ndb, nks, ld):
    join = lambda xs: (",\n              ".join(", ".join(xs[i:i + 4]) for i in range(0, len(xs), 4)))
    accs = [f'[dk{i}] "+a"(dk[{i}])' for i in range(ndb)] + [f'[dv{i}] "+a"(dv[{i}])' for i in range(ndb)]
    frags = [f'[{n}{i}] "a"({n}[{i}])' for n in ("kf", "vf") for i in range(nks)]
    offs = [f'"{{v{QOFF + i}}}"(qoff[{i}])' for i in range(ld)] + [f'"{{v{GOFF + i}}}"(goff[{i}])' for i in range(ld)]
    return join(accs), join(frags), join(offs)


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
    path = os.path.join(root, "flash_attention_annotated_amd", "csrc", "fa_bwd_loop_gen.h")
    if "--ablate" in sys.argv:
        ABLATE = int(sys.argv[sys.argv.index("--ablate") + 1])
        path = sys.argv[sys.argv.index("--out") + 1]
    global D, ROWB, TILE, STAT, KSTEPS, NSTEP, LD
    text = HEADER
    for d, deff in ((128, 128), (128, 96), (64, 64)):   # (head-dim tile, head dims contracted / produced)
        D, ROWB, TILE, KSTEPS, NSTEP, LD = d, d * 2, 64 * d * 2, deff // 16, 2 * (deff // 32), d // 32
        STAT = 4 * TILE
        inputs = set(range(QOFF, QOFF + LD)) | set(range(GOFF, GOFF + LD)) | {200, 202, 203, 208, 209}
        clob = "".join(f', "v{i}"' for i in range(NVGPR) if i not in inputs)
        accs, frags, offs = operands(deff // 32, KSTEPS, LD)
        for T, mf, cvt in (("__bf16", "v_mfma_f32_32x32x16_bf16", cvt_bf16), ("_Float16", "v_mfma_f32_32x32x16_f16", cvt_f16)):
            text += "\n" + FUNC % {"T": T, "D": deff, "NDB": d // 32, "NKS": d // 16, "LD": LD, "accs": accs, "frags": frags, "offs": offs,
                                   "body": render(gen_block(mf, cvt)), "clobbers": clob}
    text += "\n}  // namespace fa\n"
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(path) and open(path).read() == text else 1)
    open(path, "w").write(text)
    print(f"wrote {path}: {text.count(chr(10))} lines")


if __name__ == "__main__":
    main()

"""Scan a hipcc -S dump for MFMA operand hazards the compiler does not pad around inline-asm MFMAs.

Rule checked (cdna guide §5.7 item 2): a VGPR/AGPR written by a VALU instruction (v_cvt*, v_mov, v_accvgpr_*, ...)
must not be read by a v_mfma as SrcA/SrcB/SrcC within the next 2 wait states.  Every instruction in between
counts one wait state, `s_nop N` counts N+1.  MFMA->MFMA accumulate chains are exempt (hardware interlocked).
Usage: python tools/isa_hazards.py file.s  -> prints violations, exit code 1 if any.
"""
import re
import sys

MFMA_RESULT_STATES = 11  # 8-pass XDL result -> any non-MFMA reader (what hipcc pads for its own MFMAs: s_nop 10)
MFMA_RESULT_STATES_16PASS = 19  # the 64-cycle forms (v_mfma*_32x32x64_f8f6f4): 16 passes
REQUIRED = 2  # the guide's figure; the scan is run with a margin (see tests/test_isa_hazards.py)


def regs(tok):
    tok = tok.strip()
    m = re.match(r'^([va])\[(\d+):(\d+)\]$', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'^([va])(\d+)$', tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


TRANS = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')


def scan(path, required=REQUIRED):
    violations = []
    kernel = None
    window = []  # (states_so_far_after, dst_regs, text, lineno) of recent VALU writes
    last_trans = None  # (dst_regs, text, lineno) when the previous instruction was a transcendental
    mfma_out = []      # (states since issue, dst regs, text, lineno) of recent MFMAs
    in_asm = False     # inside an inline-asm block (;;#ASMSTART .. ;;#ASMEND)
    for lineno, raw in enumerate(open(path), 1):
        l = raw.strip()
        if l.startswith(';;#ASMSTART'):
            in_asm = True
        elif l.startswith(';;#ASMEND'):
            in_asm = False
        m = re.match(r'^(_Z\w+):', l)
        if m:
            kernel = m.group(1)
            window = []
            continue
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
            continue
        op = l.split()[0]
        if op == 's_branch':   # unconditional: the next line is not this one's successor
            window, mfma_out, last_trans = [], [], None
            continue
        # gfx940+ trans forwarding: a non-transcendental VALU may not read a transcendental's result in the very
        # next instruction (1 wait state); hipcc pads this only for instructions it scheduled itself
        if last_trans is not None and op.startswith('v_') and not op.startswith(TRANS):
            srcs_t = set()
            parts = l.split(None, 1)[1].split(',') if ' ' in l else []
            for t in parts[1:]:
                srcs_t |= regs(t.strip().lstrip('-').lstrip('|').rstrip('|'))
            if op.startswith('v_mfma') or op.startswith('v_cmp'):
                for t in parts[:1]:
                    pass
            # (both instructions compiler-scheduled with an s_waitcnt between them: LLVM's hazard recognizer counts that
            #  instruction as the wait state -- its own code is its own responsibility; anything touching inline asm is ours)
            if srcs_t & last_trans[0] and (in_asm or last_trans[3] or lineno == last_trans[2] + 1):
                violations.append((kernel, last_trans[2], last_trans[1], lineno, l, 0))
        if op.startswith(TRANS):
            last_trans = (regs(l.split(None, 1)[1].split(',')[0]), l, lineno, in_asm)
        elif op not in ('s_waitcnt',):
            last_trans = None
        # MFMA result read (or overwritten) by anything but a chained MFMA before the matrix pipe has written it
        if ' ' in l and (op.startswith('v_') or op.startswith('ds_') or op.startswith('global_') or op.startswith('buffer_')
                         or op.startswith('scratch_')):
            toks = [t.strip().lstrip('-') for t in l.split(None, 1)[1].split(',')]
            touched = set()
            for t in toks:
                touched |= regs(t)
            for dist, dst, text, ln in mfma_out:
                if dist < (MFMA_RESULT_STATES_16PASS if '32x32x64' in text else MFMA_RESULT_STATES) and (dst & touched):
                    if op.startswith('v_mfma'):
                        # allowed: accumulate chain (same tuple as SrcC and vDst); anything else is flagged
                        srcc = toks[3].split()[0] if op.startswith('v_mfma_scale') else toks[-1].split()[0]
                        if regs(toks[0]) == dst and regs(srcc) == dst and not (
                                (regs(toks[1]) | regs(toks[2])) & dst):
                            continue
                    violations.append((kernel, ln, text, lineno, l, dist))
        if op.startswith('v_mfma'):
            ops = l.split(None, 1)[1].split(',')
            srcs = set()
            for t in ops[1:]:
                srcs |= regs(t)
            for dist, dst, text, ln in window:
                # (hipcc's hazard recognizer covers the MFMAs it emits itself; only asm ones are unprotected)
                if in_asm and dist < required and (dst & srcs):
                    violations.append((kernel, ln, text, lineno, l, dist))
            states = 1
            new_w = []
        else:
            states = 1
            if op == 's_nop':
                states = int(l.split()[1]) + 1
            elif op in ('s_waitcnt', 's_barrier') or op.startswith(';'):
                states = 0  # may retire without spending an issue cycle: do not count on it
        # age the windows
        mfma_out = [(d + states, dst, t, ln) for d, dst, t, ln in mfma_out if d + states < MFMA_RESULT_STATES_16PASS + 1]
        if op.startswith('v_mfma'):
            mfma_out.append((0, regs(l.split(None, 1)[1].split(',')[0]), l, lineno))
        window = [(d + states, dst, t, ln) for d, dst, t, ln in window if d + states < required + 1]
        if op.startswith('v_') and not op.startswith('v_mfma') and not op.startswith('v_cmp'):
            dst = regs(l.split(None, 1)[1].split(',')[0]) if ' ' in l else set()
            window.append((0, dst, l, lineno))
    return violations


if __name__ == '__main__':
    v = scan(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else REQUIRED)
    for k, ln, text, ln2, mf, dist in v:
        print(f"{k}: line {ln}: `{text}` -> line {ln2}: `{mf}` ({dist} wait states between)")
    print(f"{len(v)} violation(s)")
    sys.exit(1 if v else 0)

#!/usr/bin/env python3
"""Generates ofps_amd/csrc/lk_rows9.inc: the nine window rows of one Gauss-Newton step of lk_level_lds_kernel<4, .> as ONE
hand-scheduled gfx950 inline-asm body (spec revision 2), for tiles whose window columns AND rows sample consecutive texels
(interior tiles: 97 % of a 1080p frame).

Why one block: a row's LDS reads (10 texels of its lower sample row + 9 tile records) used to be issued as a burst at the row's
start and drained at its first tap; with 24 waves per CU sharing one in-order LDS queue that drain waits behind every other
wave's burst -- rows ran at ~1,800 cycles per wave for 81 VALU instructions, with neither the VALU (66 %) nor the LDS pipe
(79 %) saturated, and halving the reads bought 7 % (profiles/r04/lk_lds_experiments.txt).  Here the reads are software-pipelined
ACROSS rows with no extra registers: as soon as tap k of row r has consumed the vertical interpolation t[k], that register
receives texel k of row r + 1's lower sample row (it is row r + 1's l[k]: the two register sets swap roles from row to row),
and the two record quads freed by taps 7 and 8 receive row r + 1's first two records.  A row therefore starts with everything
it needs already requested a whole row earlier; each tap issues two reads and waits only for a record requested two taps
before.  In-flight loads across rows are only safe inside one asm statement (the compiler must never touch a register with a
load in flight), hence one block per step with immediate offsets for every address:
    texel k of the sample row r rows below the first:  ja + r * JSB + 4 k      (JSB = row pitch of jl[][] in bytes)
    record of window row r, tap k:                      ta + r * TRB + 16 k     (TRB = row pitch of tile[][] in bytes)
The arithmetic is the per-row kernel's (lk_row9_asm): same operations on the same operands in the same order -> same bits.

Register sets: r0..r18.  Row r has parity P = r & 1.  P = 0: l[k] = r[k] (k = 0..9), t[k] = r[10 + k]; P = 1: l[k] = r[10 + k]
(k < 9), l[9] = r[9], t[k] = r[k].  Record quads: LK_QE (even taps), LK_QO (odd taps) -- string macros lk.hip defines: the top eight
registers of the kernel's register budget (v[72:79] at 6 waves per SIMD).
Every s_waitcnt count is derived below from the queue of reads in flight (LDS returns in issue order).
"""
import os
DUMMY = int(os.environ.get("LK_GEN_DUMMY", "0"))      # timing experiments only: that many extra full-rate VALU instructions per tap
NQ = 2                                                # record quads in flight.  (3, 4, 6 were measured in round 4 with fixed registers
#                                                       v[72:95]: 228 / 225 / 229 us against 222 -- profiles/r04/lk_lds_experiments.txt)


def QR(slot):                                         # the bodies are function-like macros: lk.hip passes the quads' register names
    return f'" Q{slot} "'                             # (the top eight registers of the kernel's budget, which depends on the build)


def QC(slot, c):
    return f'" Q{slot}_{c} "'


QARGS = ", ".join(f"Q{i}, Q{i}_0, Q{i}_1, Q{i}_2" for i in range(NQ))

JSB, TRB, N = 256, 640, 9          # JSB is overridden per generated variant (main)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def L(P, k):
    return f"%[r{k}]" if P == 0 else (f"%[r{10 + k}]" if k < 9 else "%[r9]")


def T(P, k):
    return f"%[r{10 + k}]" if P == 0 else f"%[r{k}]"


class Emit:
    def __init__(self):
        self.lines = []
        self.queue = []          # ids of LDS reads in issue order (only the relative order after the last full drain matters)

    def op(self, s):
        self.lines.append(s)

    def read(self, ident, s):
        self.queue.append(ident)
        self.lines.append(s)

    def wait_for(self, needed):
        """s_waitcnt lgkmcnt(n): n = reads issued after the youngest needed one."""
        idx = [self.queue.index(i) for i in needed if i in self.queue]
        if not idx:
            return
        n = len(self.queue) - 1 - max(idx)
        n = min(n, 15)                                   # (the counter has four bits: waiting for more than needed is safe)
        self.lines.append(f"s_waitcnt lgkmcnt({n})")
        self.queue = self.queue[max(idx) + 1:]          # everything up to the youngest needed read has landed


def body(with_g: bool, jsb: int = 256) -> str:
    global JSB
    JSB = jsb
    e = Emit()
    e.op("s_waitcnt lgkmcnt(0)")                          # scalar loads the compiler may have in flight return out of order: none past here
    # ---- prologue: upper sample row of window row 0 -> its nine horizontal interpolations in row 0's t-set (P = 0: r10..r18),
    # the tenth texel in r9; row 0's first two records; row 0's lower texels 0..8 (the tenth, r9, once r9 is free)
    up = [T(0, k) for k in range(9)] + ["%[r9]"]
    for k in range(10):
        e.read(("U", k), f"ds_read_b32 {up[k]}, %[ja] offset:{4 * k}")
    for g in range(NQ):                                   # the first NQ records (taps are numbered 0..80 across the rows; tap g uses quad g % NQ)
        e.read(("Q", g), f"ds_read_b128 {QR(g % NQ)}, %[ta] offset:{(g // N) * TRB + 16 * (g % N)}")
    for k in range(9):                                    # (the lower texels are requested between the interpolations: lgkmcnt counts to 15)
        e.wait_for([("U", k), ("U", k + 1)])
        e.op(f"v_sub_f32 %[tmp], {up[k + 1]}, {up[k]}")
        e.op(f"v_fmac_f32 {up[k]}, %[a{k}], %[tmp]")
        e.read(("X", 0, k), f"ds_read_b32 {L(0, k)}, %[ja] offset:{JSB + 4 * k}")
    e.read(("X", 0, 9), f"ds_read_b32 {L(0, 9)}, %[ja] offset:{JSB + 36}")
    for r in range(N):
        P, last = r & 1, r == N - 1
        # the row's vertical fraction: v_fract_f32 of the oracle's sum ((float)(y + r - R)) + v
        if r == 0:
            e.op("v_add_f32 %[ay], %[yf0], %[fy]")
        else:
            e.op(f"v_add_f32 %[ay], {float(r)}, %[yf0]")
            e.op("v_add_f32 %[ay], %[ay], %[fy]")
        e.op("v_fract_f32 %[ay], %[ay]")
        for k in range(N):
            g = r * N + k
            q = (QC(g % NQ, 0), QC(g % NQ, 1), QC(g % NQ, 2))
            e.wait_for([("X", r, k), ("X", r, k + 1)])
            e.op(f"v_sub_f32 %[tmp], {L(P, k + 1)}, {L(P, k)}")
            if k == 8 and not last:                      # l[9] (r9) has been read for the last time: row r + 1's tenth texel
                e.read(("X", r + 1, 9), f"ds_read_b32 %[r9], %[ja] offset:{(r + 2) * JSB + 36}")
            e.op(f"v_fmac_f32 {L(P, k)}, %[a{k}], %[tmp]")
            e.op(f"v_sub_f32 %[tmp], {L(P, k)}, {T(P, k)}")
            e.op(f"v_fmac_f32 {T(P, k)}, %[ay], %[tmp]")
            for _ in range(DUMMY):
                e.op("v_mul_f32 %[ay2], %[fy], %[fy]")
            e.wait_for([("Q", g)])
            e.op(f"v_sub_f32 %[tmp], {q[0]}, {T(P, k)}")
            if with_g:
                e.op(f"v_fmac_f32 %[gxx], {q[1]}, {q[1]}")
                e.op(f"v_fmac_f32 %[bx], {q[1]}, %[tmp]")
                e.op(f"v_fmac_f32 %[gxy], {q[1]}, {q[2]}")
                e.op(f"v_fmac_f32 %[by], {q[2]}, %[tmp]")
                e.op(f"v_fmac_f32 %[gyy], {q[2]}, {q[2]}")
            else:
                e.op(f"v_fmac_f32 %[bx], {q[1]}, %[tmp]")
                e.op(f"v_fmac_f32 %[by], {q[2]}, %[tmp]")
            if g + NQ < N * N:                           # the quad this tap used: the record of tap g + NQ (the next row's first ones at a row's end)
                e.read(("Q", g + NQ), f"ds_read_b128 {QR(g % NQ)}, %[ta] offset:{((g + NQ) // N) * TRB + 16 * ((g + NQ) % N)}")
            if not last:                                 # t[k] is dead: it is row r + 1's l[k]
                e.read(("X", r + 1, k), f"ds_read_b32 {T(P, k)}, %[ja] offset:{(r + 2) * JSB + 4 * k}")
    e.op("s_waitcnt lgkmcnt(0)")
    return "\n".join(f'    "{ln}\\n\\t"' for ln in e.lines)


class Stream:
    """One pixel's operands inside the two-pixel block: register sets, fractions, sums, addresses -- named per stream."""

    def __init__(self, tag, reg, a, tmp, ay, fy, ja, acc, yoff):
        self.tag, self.reg, self.a, self.tmp, self.ay, self.fy, self.ja, self.acc, self.yoff = tag, reg, a, tmp, ay, fy, ja, acc, yoff

    def L(self, P, k):
        return f"%[{self.reg}{k}]" if P == 0 else (f"%[{self.reg}{10 + k}]" if k < 9 else f"%[{self.reg}9]")

    def T(self, P, k):
        return f"%[{self.reg}{10 + k}]" if P == 0 else f"%[{self.reg}{k}]"


def body_pair(with_g: bool, jsb: int = 256) -> str:
    """Two vertically adjacent pixels per lane (A above B): B's window row r reads the records of A's window row r + 1, so the ten
    record rows R = 0..9 are read ONCE -- record (R, k) serves A's tap (R, k) and B's tap (R - 1, k): 90 ds_read_b128 for two
    pixels instead of 162.  Each pixel keeps its own texel registers, fractions and sums (the current frame is sampled at ITS
    flow); per pixel the operations and their order are the single-pixel block's.  B's first-row set-up (ten texels, nine
    interpolations) rides in record row 0, where only A has taps."""
    A = Stream("A", "r", "a", "%[tmp]", "%[ay]", "%[fy]", "%[ja]", ("%[bx]", "%[by]", "%[gxx]", "%[gxy]", "%[gyy]"), 0)
    B = Stream("B", "s", "b", "%[tmq]", "%[az]", "%[fz]", "%[jb]", ("%[cx]", "%[cy]", "%[hxx]", "%[hxy]", "%[hyy]"), 1)
    e = Emit()
    e.op("s_waitcnt lgkmcnt(0)")

    def upper(S, k):                                       # where the set-up keeps the upper sample row of window row 0
        return S.T(0, k) if k < 9 else f"%[{S.reg}9]"

    def issue_upper(S):
        for k in range(10):
            e.read(("U", S.tag, k), f"ds_read_b32 {upper(S, k)}, {S.ja} offset:{4 * k}")

    def setup_piece(S, k):                                 # interpolation k of the first upper row; its register's successor read
        e.wait_for([("U", S.tag, k), ("U", S.tag, k + 1)])
        e.op(f"v_sub_f32 {S.tmp}, {upper(S, k + 1)}, {upper(S, k)}")
        e.op(f"v_fmac_f32 {upper(S, k)}, %[{S.a}{k}], {S.tmp}")
        e.read(("X", S.tag, 0, k), f"ds_read_b32 {S.L(0, k)}, {S.ja} offset:{jsb + 4 * k}")
        if k == 8:
            e.read(("X", S.tag, 0, 9), f"ds_read_b32 {S.L(0, 9)}, {S.ja} offset:{jsb + 36}")

    def row_fraction(S, r):                                # v_fract_f32 of the oracle's sum ((float)(y + r - R)) + v; B's y is A's + 1
        c = r + S.yoff
        if c == 0:
            e.op(f"v_add_f32 {S.ay}, %[yf0], {S.fy}")
        else:
            e.op(f"v_add_f32 {S.ay}, {float(c)}, %[yf0]")
            e.op(f"v_add_f32 {S.ay}, {S.ay}, {S.fy}")
        e.op(f"v_fract_f32 {S.ay}, {S.ay}")

    def tap(S, r, k):
        P = r & 1
        e.wait_for([("X", S.tag, r, k), ("X", S.tag, r, k + 1)])
        e.op(f"v_sub_f32 {S.tmp}, {S.L(P, k + 1)}, {S.L(P, k)}")
        if k == 8 and r < N - 1:
            e.read(("X", S.tag, r + 1, 9), f"ds_read_b32 %[{S.reg}9], {S.ja} offset:{(r + 2) * jsb + 36}")
        e.op(f"v_fmac_f32 {S.L(P, k)}, %[{S.a}{k}], {S.tmp}")
        e.op(f"v_sub_f32 {S.tmp}, {S.L(P, k)}, {S.T(P, k)}")
        e.op(f"v_fmac_f32 {S.T(P, k)}, {S.ay}, {S.tmp}")

    def use(S, r, k, q):
        P = r & 1
        bx, by, gxx, gxy, gyy = S.acc
        e.op(f"v_sub_f32 {S.tmp}, {q[0]}, {S.T(P, k)}")
        if with_g:
            e.op(f"v_fmac_f32 {gxx}, {q[1]}, {q[1]}")
            e.op(f"v_fmac_f32 {bx}, {q[1]}, {S.tmp}")
            e.op(f"v_fmac_f32 {gxy}, {q[1]}, {q[2]}")
            e.op(f"v_fmac_f32 {by}, {q[2]}, {S.tmp}")
            e.op(f"v_fmac_f32 {gyy}, {q[2]}, {q[2]}")
        else:
            e.op(f"v_fmac_f32 {bx}, {q[1]}, {S.tmp}")
            e.op(f"v_fmac_f32 {by}, {q[2]}, {S.tmp}")

    def prefetch(S, r, k):                                 # t[k] of window row r is dead: it becomes row r + 1's l[k]
        if r < N - 1:
            e.read(("X", S.tag, r + 1, k), f"ds_read_b32 {S.T(r & 1, k)}, {S.ja} offset:{(r + 2) * jsb + 4 * k}")

    NR = N + 1                                             # record rows
    issue_upper(A)
    for g in range(NQ):
        e.read(("Q", g), f"ds_read_b128 {QR(g % NQ)}, %[ta] offset:{(g // N) * TRB + 16 * (g % N)}")
    for k in range(9):
        setup_piece(A, k)
    issue_upper(B)
    for R in range(NR):
        if R < N:
            row_fraction(A, R)
        if R >= 1:
            row_fraction(B, R - 1)
        for k in range(N):
            g = R * N + k
            q = (QC(g % NQ, 0), QC(g % NQ, 1), QC(g % NQ, 2))
            if R < N:
                tap(A, R, k)
            if R == 0:
                setup_piece(B, k)
            else:
                tap(B, R - 1, k)
            e.wait_for([("Q", g)])
            if R < N:
                use(A, R, k, q)
            if R >= 1:
                use(B, R - 1, k, q)
            if g + NQ < NR * N:
                e.read(("Q", g + NQ), f"ds_read_b128 {QR(g % NQ)}, %[ta] offset:{((g + NQ) // N) * TRB + 16 * ((g + NQ) % N)}")
            if R < N:
                prefetch(A, R, k)
            if R >= 1:
                prefetch(B, R - 1, k)
    e.op("s_waitcnt lgkmcnt(0)")
    return "\n".join(f'    "{ln}\\n\\t"' for ln in e.lines)


def main():
    out = ['// GENERATED by tools/gen_lk_rows9.py -- do not edit.  See that file for the schedule and the derivation of every wait count.',
           f'// TRB = {TRB} (tile[][] row pitch in bytes); one pair of bodies per jl[][] row pitch (OFPS_LK_JS floats): lk.hip static_asserts both.',
           '// Every body is a function-like macro of the record quads: (Q0, Q0_0, Q0_1, Q0_2, Q1, Q1_0, Q1_1, Q1_2) = the quad as a register',
           '// range and its first three registers, as string literals.',
           '#define LK_ROWS9_TRB ' + str(TRB)]
    for k, js in enumerate((64,)):          # (68 was measured in round 4: more bank conflicts, 238 vs 221 us; profiles/r04/lk_lds_experiments.txt)
        out += [('#if' if k == 0 else '#elif') + f' OFPS_LK_JS == {js}', f'#define LK_ROWS9_JSB {4 * js}',
                '#define LK_ROWS9_BODY(' + QARGS + ') \\', body(False, 4 * js).replace("\n", " \\\n"), '',
                '#define LK_ROWS9_BODY_G(' + QARGS + ') \\', body(True, 4 * js).replace("\n", " \\\n"), '',
                '#define LK_ROWS9_PAIR_BODY(' + QARGS + ') \\', body_pair(False, 4 * js).replace("\n", " \\\n"), '',
                '#define LK_ROWS9_PAIR_BODY_G(' + QARGS + ') \\', body_pair(True, 4 * js).replace("\n", " \\\n"), '']
    out += ['#else', '#error "lk_rows9.inc has no body for this OFPS_LK_JS"', '#endif', '']
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ofps_amd", "csrc", "lk_rows9.inc")   # a path: tests diff against the committed file
    with open(path, "w") as f:
        f.write("\n".join(out))
    print(path, len(out), "blocks")


if __name__ == "__main__":
    main()

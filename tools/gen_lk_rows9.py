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
DUMMYQ = int(os.environ.get("LK_GEN_DUMMYQ", "0"))    # timing experiments only: extra ds_read_b128 per row into v[80:83] (96-register build)
NQ = int(os.environ.get("LK_GEN_NQ", "2"))            # record quads in flight (2: v[72:79] at the 80-register budget)
QB = int(os.environ.get("LK_GEN_QBASE", "72"))        # first register of the quads


def QR(slot):
    return f"v[{QB + 4 * slot}:{QB + 4 * slot + 3}]"


def QC(slot, c):
    return f"v{QB + 4 * slot + c}"

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
            if k < DUMMYQ:
                e.read(("D", g), f"ds_read_b128 v[80:83], %[ta] offset:{r * TRB + 16 * k}")
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


def main():
    out = ['// GENERATED by tools/gen_lk_rows9.py -- do not edit.  See that file for the schedule and the derivation of every wait count.',
           f'// TRB = {TRB} (tile[][] row pitch in bytes); one pair of bodies per jl[][] row pitch (OFPS_LK_JS floats): lk.hip static_asserts both.',
           '#define LK_ROWS9_TRB ' + str(TRB),
           '#define LK_ROWS9_CLOBBERS ' + ", ".join(f'"v{QB + i}"' for i in range(4 * NQ)) + (', "v80", "v81", "v82", "v83"' if DUMMYQ else ""),
           f'#define LK_ROWS9_TOP_REG {QB + 4 * NQ - 1}']
    for k, js in enumerate((64,)):          # (68 was measured in round 4: more bank conflicts, 238 vs 221 us; profiles/r04/lk_lds_experiments.txt)
        out += [('#if' if k == 0 else '#elif') + f' OFPS_LK_JS == {js}', f'#define LK_ROWS9_JSB {4 * js}',
                '#define LK_ROWS9_BODY \\', body(False, 4 * js).replace("\n", " \\\n"), '',
                '#define LK_ROWS9_BODY_G \\', body(True, 4 * js).replace("\n", " \\\n"), '']
    out += ['#else', '#error "lk_rows9.inc has no body for this OFPS_LK_JS"', '#endif', '']
    path = os.path.join(ROOT, "ofps_amd", "csrc", "lk_rows9.inc")
    with open(path, "w") as f:
        f.write("\n".join(out))
    print(path, len(out), "blocks")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Instruction mix of sad_strip_kernel<B,R> in a hipcc -save-temps .s file: counts by mnemonic class, registers, LDS, scratch.
  python tools/sad_isa_stats.py <file.s> [B R]      (default 8 32: BASELINE configs[3])"""
import collections
import re
import sys

path = sys.argv[1]
B, R = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 32)
name = f"_ZN12_GLOBAL__N_116sad_strip_kernelILi{B}ELi{R}EEEvNS_9SadParamsEii"
txt = open(path).read()
start = txt.index(name + ":")
end = txt.index(".Lfunc_end", start)
body = txt[start:end]
meta = txt[end:end + 6000]
ops = collections.Counter()
for ln in body.splitlines():
    ln = ln.strip()
    if not ln or ln.startswith((";", ".", "_Z", "BB", "s_nop")) or ln.endswith(":"):
        continue
    ops[ln.split()[0]] += 1
def tot(pred):
    return sum(v for k, v in ops.items() if pred(k))
valu = tot(lambda k: k.startswith("v_"))
print(f"{path}: sad_strip_kernel<{B},{R}>")
print("  VALU", valu, " of which qsad", ops["v_qsad_pk_u16_u8"], " sad_u8", ops["v_sad_u8"], " other", valu - ops["v_qsad_pk_u16_u8"] - ops["v_sad_u8"])
print("  SALU", tot(lambda k: k.startswith("s_") and not k.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_barrier"))),
      " branches", tot(lambda k: k.startswith(("s_cbranch", "s_branch"))), " s_waitcnt", ops["s_waitcnt"])
print("  LDS", tot(lambda k: k.startswith("ds_")), {k: v for k, v in ops.items() if k.startswith("ds_")})
print("  VMEM", tot(lambda k: k.startswith(("global_", "buffer_", "scratch_", "flat_"))))
keys = ["v_min3_u32", "v_min_u32", "v_lshl_or_b32", "v_and_or_b32", "v_add_u32", "v_mov_b32", "v_cndmask_b32", "v_or_b32", "v_and_b32",
        "v_lshlrev_b32", "v_lshrrev_b32", "v_bfi_b32", "v_perm_b32", "v_or3_b32", "v_readlane_b32", "v_writelane_b32", "s_mov_b32", "s_cselect_b32",
        "s_movk_i32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"]
print("  ", {k: ops[k] for k in keys if ops[k]})
other = {k: v for k, v in ops.most_common() if k.startswith("v_") and k not in keys and k not in ("v_qsad_pk_u16_u8", "v_sad_u8")}
print("   other VALU:", dict(list(other.items())[:14]))
for key in (".vgpr_count", ".sgpr_count", ".agpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count"):
    m = re.search(re.escape(name) + r".*?" + re.escape(key) + r":\s*(\d+)", txt[end:], flags=re.S)
    # metadata block is per kernel in the amdhsa.kernels list; search near the symbol name there
meta_i = txt.index(".name:           " + name) if (".name:           " + name) in txt else -1
if meta_i >= 0:
    blk = txt[max(0, meta_i - 1500):meta_i + 1500]
    print("  ", {k: int(v) for k, v in re.findall(r"\.(vgpr_count|sgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s*(\d+)", blk)})

#!/usr/bin/env python3
"""isa_mix.py <file.s> <kernel-substring> — instruction mix of the basic blocks of one kernel,
priced with the issue costs measured by tools/microbench/valu_issue.hip on MI355X:

  fast   2.3 cycles/SIMD  v_fma/mul/add/sub/fmac/fmaak/fmamk_f32 with VGPR / inline / literal operands,
                          v_mov_b32, v_add_u32, v_and_b32 (two waves of a SIMD issue these side by side)
  slow   4.2              every other VALU op: v_pk_*, DPP, SGPR operand, v_max/min/med3, cvt, shifts, cmp ...
  trans  8.2              v_exp/log/rcp/rsq/sqrt_f32
A packed op does two lanes' worth of work, so v_pk at 4.2 is as good as two paired fast ops.

  hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s ; tools/isa_mix.py x.s 'atrous_stream_kernelILi4ELi2E'
"""
import re
import sys
from collections import Counter

FAST = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32",
        "v_mov_b32", "v_add_u32", "v_and_b32", "v_mul_f32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_fmac_f32_e32",
        "v_mov_b32_e32", "v_add_u32_e32", "v_and_b32_e32", "v_subrev_f32_e32", "v_mul_f32_e64", "v_add_f32_e64",
        "v_sub_f32_e64", "v_fmac_f32_e64", "v_subrev_f32_e64", "v_or_b32_e32", "v_xor_b32_e32", "v_sub_u32_e32"}
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
COST = {"fast": 2.3, "slow": 4.2, "trans": 8.2}


def classify(line):
    op = line.split()[0]
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "trans"
        args = line[len(op):]
        uses_sgpr = re.search(r"(?<![a-z_\[])s\d+|s\[\d+:\d+\]|vcc|exec", args.split(";")[0]) is not None
        if op in FAST and not uses_sgpr and "dpp" not in line and "sdwa" not in line:
            return "fast"
        return "slow"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_0-9$]+:", l) and key in l.split(":")[0])
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        s = l.strip()
        if re.match(r"^\.?[A-Za-z_0-9$]+:", s):
            blocks.append((name, cur))
            name, cur = s.split(":")[0], []
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        cur.append(s)
    blocks.append((name, cur))
    tot = Counter()
    for name, ins in blocks:
        c = Counter(classify(i) for i in ins)
        tot.update(c)
        valu = c["fast"] + c["slow"] + c["trans"]
        if valu < min_valu:
            continue
        cyc = sum(c[k] * COST[k] for k in COST)
        ops = Counter(i.split()[0] for i in ins if i.startswith("v_") and classify(i) == "slow")
        print(f"{name:12s} insts {len(ins):5d}  valu {valu:4d} = fast {c['fast']:4d} slow {c['slow']:4d} trans {c['trans']:3d}"
              f"  lds {c['lds']:3d} vmem {c['vmem']:3d} salu {c['salu']:3d} wait {c['wait']:3d}  ~{cyc:6.0f} cyc")
        print("             slow ops:", ", ".join(f"{k} {v}" for k, v in ops.most_common(12)))
    print("kernel total:", dict(tot))


if __name__ == "__main__":
    main()

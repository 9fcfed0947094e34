#!/usr/bin/env python3
"""Splits the instructions of the hot loops of extz2_pair_kernel<3,false> by encoding class, from the device assembly
of the built sources (hipcc -S --cuda-device-only of sedef_amd/csrc/sdf_unity.hip).  The steady row of the headline
batch is the shortest innermost loop with all three window registers active, no v_readlane (the scalar-H rows read
lanes) and no N handling.
usage: isa_split.py unity.s > profiles/pair_kernel_isa.json"""
import json
import re
import sys

VOP3P = re.compile(r"^v_pk_|^v_mad_mix|^v_dot")
VOP2 = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_cndmask_b32", "v_lshlrev_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_add_co_u32", "v_addc_co_u32",
        "v_mul_u32_u24", "v_add_u16", "v_sub_u16", "v_max_u16", "v_min_u16", "v_max_i16", "v_min_i16", "v_lshlrev_b16"}
VOP1 = {"v_mov_b32", "v_not_b32", "v_bfrev_b32", "v_readfirstlane_b32"}
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_add_u16", "v_sub_u16", "v_subrev_u16", "v_max_i16", "v_max_u16", "v_min_i16", "v_min_u16", "v_not_b32"}


def klass(ins, ops):
    if ins.startswith("s_"):
        return "SALU/branch/wait"
    if ins.startswith("ds_"):
        return "LDS"
    if ins.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if "dpp" in ops or ins.endswith("_dpp") or "row_" in ops or "wave_" in ops or "quad_perm" in ops:
        return "VALU DPP (8-byte)"
    if "sdwa" in ins or "src0_sel" in ops or "dst_sel" in ops:
        return "VALU SDWA (8-byte)"
    if VOP3P.match(ins):
        return "VALU VOP3P packed (8-byte)"
    if ins.startswith("v_cmp"):
        return "VALU VOPC / VOP3 compare"
    base = ins.replace("_e32", "").replace("_e64", "")
    if ins.endswith("_e64") or base not in VOP2 | VOP1:
        return "VALU VOP3 (8-byte)"
    # a VOP2 / VOP1 instruction with a 32-bit literal is 8 bytes too, but issues like the 4-byte form
    return "VALU VOP2/VOP1 (4-byte)"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3sdf17extz2_pair_kernelILi3ELb0ELb0EE") and l.rstrip().endswith("sdf_result"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    labels = {}
    ins = []
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        parts = t.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else ""))
    loops = []
    for k, (op, ops) in enumerate(ins):
        if (op.startswith("s_cbranch") or op == "s_branch") and ops.strip() in labels and labels[ops.strip()] <= k:
            loops.append((labels[ops.strip()], k))
    inner = [(a, b) for (a, b) in loops if not any(a <= c and d <= b and (c, d) != (a, b) for (c, d) in loops)]
    out = {"kernel": "extz2_pair_kernel<3,false,false>", "source": "hipcc -O3 --offload-arch=gfx950 -S of sedef_amd/csrc/sdf_unity.hip",
           "whole_kernel": {}, "innermost_loops": []}
    for op, ops in ins:
        c = klass(op, ops)
        out["whole_kernel"][c] = out["whole_kernel"].get(c, 0) + 1
    for a, b in inner:
        cnt = {}
        names = [op for op, _ in ins[a:b + 1]]
        for op, ops in ins[a:b + 1]:
            c = klass(op, ops)
            cnt[c] = cnt.get(c, 0) + 1
        valu = sum(v for k_, v in cnt.items() if k_.startswith("VALU"))
        if valu < 40:
            continue
        # issue class by the measured cost (profiles/r05_ubench_valu_ops.txt): ~2.3 cycles for 32-bit add / sub / logic / right
        # shifts / moves and the 16-bit VOP2 forms without a scalar operand, ~4.2 for everything else
        fast = 0
        for op, ops in ins[a:b + 1]:
            base = op.replace("_e32", "").replace("_e64", "")
            if base in FAST and "dpp" not in op and "sdwa" not in op and "row_" not in ops and "_sel" not in ops and \
                    not re.search(r"\bs\d+|\bs\[|vcc", ops):
                fast += 1
        out["innermost_loops"].append({"instructions": b - a + 1, "valu": valu, "by_encoding": cnt,
                                       "issue_class": {"2.3_cycles": fast, "4.2_cycles": valu - fast,
                                                       "model_cycles_per_row": round(fast * 2.3 + (valu - fast) * 4.2, 1)},
                                       "v_readlane": sum(1 for x in names if x.startswith("v_readlane")),
                                       "v_pk": sum(1 for x in names if x.startswith("v_pk_"))})
    # the steady regime (every register of the window active: >= 50 packed instructions -- 66 until three differences per
    # register became 32-bit subtracts and the code comparison an xor, round 5), without the N handling the headline batch
    # does not need: the shortest such loop
    steady = [l for l in out["innermost_loops"] if l["v_readlane"] == 0 and l["v_pk"] >= 45 and
              l["by_encoding"].get("VALU DPP (8-byte)", 0) >= 10]
    steady.sort(key=lambda l: l["instructions"])
    if steady:
        out["steady_row"] = steady[0]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

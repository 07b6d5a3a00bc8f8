#!/usr/bin/env python3
"""Per-kernel summary of the gfx950 ISA of libsbm_hip (the .s itself is a build product and git-ignored):
registers, LDS, occupancy and the static instruction mix.  usage: make -C shape_based_matching_amd/csrc asm &&
python tools/isa_summary.py shape_based_matching_amd/csrc/sbm_capi.gfx950.s > profiles/rNN_isa_summary.txt"""
import collections
import re
import subprocess
import sys

path = sys.argv[1]
text = open(path).read()
bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.M | re.S)}
metas = {m.group(1): m.group(2) for m in re.finditer(r"^\s*\.amdhsa_kernel (_Z\w+)\n(.*?)\.end_amdhsa_kernel", text, re.M | re.S)}
names = [n for n in bodies if n in metas]
try:
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
    demangle = dict(zip(names, out)) if len(out) == len(names) else {}
except Exception:
    demangle = {}
print(f"{'kernel':64s} {'VGPR':>5s} {'SGPR':>5s} {'LDS':>6s} {'occ':>3s} {'valu':>6s} {'salu':>6s} {'vmem':>5s} {'lds':>4s} {'s_nop':>5s} {'v_mov':>5s} {'dpp':>4s} {'dot2':>5s} {'bitop3':>6s} {'perm':>5s}")
for name in names:
    body, meta = bodies[name], metas[name]
    ops = collections.Counter(l.split()[0] for l in body.split("\n") if re.match(r"\s+[a-z]", l))
    g = lambda pat: sum(v for k, v in ops.items() if re.match(pat, k))

    def meta_val(key):
        r = re.search(key + r"\s+(\d+)", meta)
        return int(r.group(1)) if r else -1

    vg = meta_val(r"\.amdhsa_next_free_vgpr")
    sg = meta_val(r"\.amdhsa_next_free_sgpr")
    lds = meta_val(r"\.amdhsa_group_segment_fixed_size")
    occ = min(8, 512 // max(8, (vg + 7) // 8 * 8)) if vg > 0 else 0
    dn = demangle.get(name, name)
    dn = re.sub(r"\(.*", "", dn).replace("void ", "").replace("sbm::", "")
    print(f"{dn[:64]:64s} {vg:5d} {sg:5d} {lds:6d} {occ:3d} {g(r'v_'):6d} {g(r's_'):6d} {g(r'(global_|buffer_|flat_)'):5d} {g(r'ds_'):4d} "
          f"{ops.get('s_nop', 0):5d} {g(r'v_mov_b32(_e32)?$'):5d} {g(r'v_.*_dpp$'):4d} {g(r'v_dot2'):5d} {g(r'v_bitop3'):6d} {g(r'v_perm_b32'):5d}")

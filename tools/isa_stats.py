"""Instruction histogram per kernel from the -save-temps gfx950 assembly (csrc/Makefile target `isa`)."""
import collections, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/fls_reg-hip-amdgcn-amd-amdhsa-gfx950.s"
want = sys.argv[2:]
lines = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for idx, (i, name) in enumerate(starts):
    if want and not any(w in name for w in want):
        continue
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(lines)
    ops = []
    for l in lines[i + 1:end]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        ops.append(re.sub(r"_e(32|64)$|_dpp$", "", t.split()[0]))
    c = collections.Counter(ops)
    f64 = sum(v for k, v in c.items() if "f64" in k)
    print(f"{name[:70]:70s} instr {len(ops):6d} f64 {f64:5d} bperm {c.get('ds_bpermute_b32',0):4d} dpp {sum(v for k,v in c.items() if 'dpp' in k):4d} "
          f"div_scale {c.get('v_div_scale_f64',0):3d} sqrt/rsq {c.get('v_rsq_f64',0)+c.get('v_sqrt_f64',0):3d} waitcnt {c.get('s_waitcnt',0):4d}")
    if "-v" in sys.argv:
        print("    ", c.most_common(25))

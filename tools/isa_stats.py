"""instruction mix of one kernel from `hipcc -S --cuda-device-only` output: tools/isa_stats.py file.s <mangled-substring> [lo hi]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and l.split(":")[0].endswith(pat) or (l.startswith("_Z") and pat in l and ":" in l and not l.startswith("\t")))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
body = lines[start:end]
code_end = next((i for i, l in enumerate(body) if l.strip().startswith(".section")), len(body))
ins = [l.strip() for l in body[:code_end] if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().split(";")[0].strip().endswith(":")]
meta = "\n".join(body[code_end:])
for k in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size", "group_segment_fixed_size"):
    m = re.search(r"\.amdhsa_%s (\d+)" % k, meta)
    print(k, m.group(1) if m else None)
c = collections.Counter(i.split()[0] for i in ins)
print("instructions", len(ins))
groups = collections.Counter()
for k, v in c.items():
    g = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
    groups[g] += v
print(dict(groups))
print(sorted(c.items(), key=lambda kv: -kv[1])[:40])

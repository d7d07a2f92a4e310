"""Print VGPR / spill / scratch of every kernel in a hipcc -S listing:  python tools/kernel_regs.py file.s [substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
recs = re.findall(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', txt, re.S)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in recs), capture_output=True, text=True).stdout.splitlines()
for (n, priv, v, sp), d in zip(recs, names):
    d = d.replace("(anonymous namespace)::", "")
    if flt in d:
        print(f"vgpr {v:>3} spill {sp:>3} scratch {priv:>4}  {d[:120]}")

"""VGPR / spill / scratch / LDS of every kernel in a built object or library (no recompilation: the device ELF is cut out of
the offload bundle and its notes are read):  python tools/obj_regs.py unimm_amd/csrc/_obj/gemm.o [substring]"""
import re, subprocess, sys, tempfile, os
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
data = open(path, "rb").read()
b = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
starts = [m.start() for m in re.finditer(b"\x7fELF", data) if b < 0 or m.start() > b]
readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
for s in starts:
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[s:])
        name = f.name
    txt = subprocess.run([readelf, "--notes", name], capture_output=True, text=True).stdout
    os.unlink(name)
    recs = re.findall(r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S)
    names = subprocess.run(["c++filt"], input="\n".join(r[1] for r in recs), capture_output=True, text=True).stdout.splitlines()
    for (lds, n, priv, sg, v, sp), d in zip(recs, names):
        d = d.replace("(anonymous namespace)::", "")
        if flt in d:
            print(f"vgpr {v:>3} sgpr {sg:>3} spill {sp:>3} scratch {priv:>4} lds {lds:>6}  {d[:130]}")

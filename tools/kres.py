#!/usr/bin/env python3
"""Per-kernel resources (VGPRs, SGPRs, scratch/spills, LDS) of the built device code object.
usage: tools/kres.py [object-or-so] [name-filter]"""
import re
import subprocess
import sys
import tempfile
import os

ROCM_LLVM = "/opt/rocm/lib/llvm/bin"


def main(path, flt=""):
    with tempfile.TemporaryDirectory() as td:
        # the device code object sits in the host object's .hip_fatbin section as a clang offload bundle
        fat = os.path.join(td, "fat.bin")
        out = os.path.join(td, "dev.co")
        # objcopy rewrites its input when no output is named (and with it the mtime make goes by): always write to a scratch copy
        subprocess.run([f"{ROCM_LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(td, "copy.o")], capture_output=True)
        r = subprocess.run([f"{ROCM_LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
            out = path
        notes = subprocess.run([f"{ROCM_LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for line in notes.splitlines():
        m = re.match(r"\s+\.(\w+):\s+(.*)", line)
        if not m:
            m2 = re.match(r"\s+- \.(\w+):\s+(.*)", line)
            if m2 and m2.group(1) in ("agpr_count", "args"):
                if cur.get("name"):
                    rows.append(cur)
                cur = {}
                m = m2
            elif not m2:
                continue
            else:
                m = m2
        cur[m.group(1)] = m.group(2).strip()
    if cur.get("name"):
        rows.append(cur)
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'spillV':>6} {'spillS':>6} {'scratch':>7} {'lds':>7}  kernel")
    for k in rows:
        name = k.get("name", "")
        if flt and not re.search(flt, name):
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").replace("he355::", "").replace("void ", "")
        dem = re.sub(r"\((he355|unsigned|const|int|K\d|Floor|Poly|Behz).*", "", dem)[:100]
        print(f"{k.get('vgpr_count','?'):>5} {k.get('agpr_count','?'):>5} {k.get('sgpr_count','?'):>5} {k.get('vgpr_spill_count','?'):>6} "
              f"{k.get('sgpr_spill_count','?'):>6} {k.get('private_segment_fixed_size','?'):>7} {k.get('group_segment_fixed_size','?'):>7}  {dem}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "reference-seal-backend_amd/csrc/_obj/he355_kernels.o", sys.argv[2] if len(sys.argv) > 2 else "")

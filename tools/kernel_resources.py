"""Per-kernel register / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py vidsitu_amd/csrc/conv_igemm.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Iinclude",
       "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                     capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>7} {'occ':>4} {'LDS':>7}  kernel")
for r, d in zip(rows, dem):
    if flt and flt not in d:
        continue
    print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('SGPRs','?'):>5} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('Occupancy [waves/SIMD]','?'):>4} "
          f"{r.get('LDS Size [bytes/block]','?'):>7}  {d.replace('void ','')[:110]}")

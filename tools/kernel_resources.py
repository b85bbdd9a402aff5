#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/kernel_resources.py cuclark_amd/csrc/mic_kernels.hip [filter] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = args[0]
    flt = args[1] if len(args) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include",
           f"-I{ROOT}/cuclark_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.stderr.write(r.stderr[-4000:])
        sys.exit(1)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass", line) or re.search(r"remark:\s+(.*) \[-Rpass", line)
        body = re.sub(r"^.*?:\d+:\d+:\s*(remark:\s*)?", "", line).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:") or body.startswith("Name:"):
            name = body.split(":", 1)[1].strip()
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": d}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"{'kernel':90s} {'SGPR':>5s} {'VGPR':>5s} {'AGPR':>5s} {'scr':>5s} {'occ':>4s} {'LDS':>7s}")
    for c in rows:
        if flt and flt not in c["name"]:
            continue
        n = c["name"].replace("(anonymous namespace)::", "").replace("(MicQueryArgs)", "")
        print(f"{n[:90]:90s} {c.get('TotalSGPRs', c.get('SGPRs', '?')):>5s} {c.get('VGPRs', '?'):>5s} {c.get('AGPRs', '?'):>5s} "
              f"{c.get('ScratchSize [bytes/lane]', '?'):>5s} {c.get('Occupancy [waves/SIMD]', '?'):>4s} {c.get('LDS Size [bytes/block]', '?'):>7s}")


if __name__ == "__main__":
    main()

"""Register / LDS / occupancy table of every kernel in a HIP source (device-only compile with the library's flags).
usage: python tools/kernel_regs.py raytracer_amd/csrc/rt_trace.hip [name-filter]"""
import re, subprocess, sys
FLAGS = "--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero".split()


def main():
    src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    FLAGS.extend(sys.argv[3:])          # extra compiler flags, e.g. -DRT_SORT_KPT=20
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                       capture_output=True, text=True)
    cur, rows = None, []
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for k in rows:
        if flt in k["name"]:
            n = re.sub(r"\(anonymous namespace\)::", "", k["name"]).split("(")[0]
            print(f"{n:70s} vgpr {k.get('VGPRs', -1):4d} agpr {k.get('AGPRs', 0):3d} sgpr-spill {k.get('SGPRs Spill', 0):3d} vgpr-spill {k.get('VGPRs Spill', 0):3d} "
                  f"occ {k.get('Occupancy [waves/SIMD]', -1)} lds {k.get('LDS Size [bytes/block]', 0)}")


if __name__ == "__main__":
    main()

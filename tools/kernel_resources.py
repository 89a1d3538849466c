#!/usr/bin/env python
"""profiles/<tag>_resources.csv: registers, LDS, scratch and the compiler's occupancy figure of every gfx950 kernel in
libpbwtgpu.so, from the compiler's own resource remarks (`-Rpass-analysis=kernel-resource-usage` on the same source and flags
as pbwt_amd/build.py; no GPU needed).  DESIGN.md section 2's co-residency rule (a chain kernel must fit the hole one retiring
consumer workgroup leaves: <= 56 VGPRs, <= 26 KB LDS, 4 waves) is checked against this table by tests/test_abi.py.
Usage: python tools/kernel_resources.py [tag=r03]"""
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def collect():
    src = os.path.join(ROOT, "pbwt_amd", "csrc", "pbwt_engine.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", ln)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"mangled": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["mangled"] for r in rows])
    out = []
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*\)$", "", n).replace("void ", "").replace("pbwtk::", "")
        lds = int(r.get("LDS Size [bytes/block]", 0))
        vg = int(r.get("VGPRs", 0))
        # waves per SIMD the register file allows (512 VGPRs per lane and SIMD on gfx950, allocation granule 8) and workgroups per CU the 160 KB of LDS allow
        occ_v = min(8, 512 // max(8, (vg + 7) // 8 * 8))
        out.append({"kernel": n, "vgprs": vg, "agprs": int(r.get("AGPRs", 0)), "sgprs": int(r.get("TotalSGPRs", 0)),
                    "scratch_bytes_per_lane": int(r.get("ScratchSize [bytes/lane]", 0)), "lds_bytes_per_workgroup": lds,
                    "occupancy_waves_per_simd": int(r.get("Occupancy [waves/SIMD]", 0)), "waves_per_simd_by_vgprs": occ_v,
                    "workgroups_per_cu_by_lds": (160 * 1024 // lds) if lds else 0})
    return sorted(out, key=lambda r: r["kernel"])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    rows = collect()
    path = os.path.join(ROOT, "profiles", tag + "_resources.csv")
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    print("wrote %s: %d kernels" % (path, len(rows)))
    for r in rows:
        if r["kernel"].startswith("skel_") or r["kernel"].startswith("sweep_hist"):
            print("%-52s vgpr %3d lds %6d occ %d scratch %d" % (r["kernel"][:52], r["vgprs"], r["lds_bytes_per_workgroup"], r["occupancy_waves_per_simd"], r["scratch_bytes_per_lane"]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy table of the gfx950 build (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py            # prints the table
Used by tests/test_kernel_resources.py: a by-value argument block that lands in scratch (2 KB per lane in the
one-block decision kernel) once cost the sequential mode half its throughput without failing any parity test."""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cluster-capacity_amd", "csrc", "ccsim_engine.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"]


def demangle_kernel(sym: str) -> str:
    """_ZN5ccsim6k_scanILi0ELb0ELb1ELi0EEEvNS_8ScanArgsE -> k_scan<0,0,1,0>; _ZN5ccsim7k_finalENS_8ScanArgsE -> k_final"""
    m = re.match(r"_ZN5ccsim(\d+)", sym)
    if not m:
        return sym
    n = int(m.group(1))
    start = m.end()
    name, rest = sym[start:start + n], sym[start + n:]
    if rest.startswith("I"):
        args = re.findall(r"L([ib])(\d+)E", rest.split("EEv")[0] + "E")
        name += "<" + ",".join(v for _, v in args) + ">"
    return name


def collect(hipcc: str = "/opt/rocm/bin/hipcc") -> dict:
    with tempfile.TemporaryDirectory() as td:
        p = subprocess.run([hipcc, *FLAGS, "-Rpass-analysis=kernel-resource-usage", SRC, "-o", os.path.join(td, "x.so")],
                           capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(p.stderr[-2000:])
    rows, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = demangle_kernel(m.group(1))
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    return rows


LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def collect_from_library(lib_path: str) -> dict:
    """The same table read off the BUILT library (no second compile of the engine: the first form of this check took 70 s of the CPU
    suite): .hip_fatbin -> the gfx950 code object -> its AMDGPU metadata note (per kernel: .vgpr_count, .private_segment_fixed_size =
    the scratch frame, .vgpr_spill_count).  Occupancy = waves per SIMD the unified 512-entry register file allows (allocation granule 8)."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.run([f"{LLVM_BIN}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat], check=True, capture_output=True)
        subprocess.run([f"{LLVM_BIN}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                       check=True, capture_output=True)
        notes = subprocess.run([f"{LLVM_BIN}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    rows, cur = {}, None
    for line in notes.splitlines():
        if line.startswith("  - "):  # the next kernel of amdhsa.kernels
            cur = {}
            line = "    " + line[4:]
        m = re.match(r"    \.(name|vgpr_count|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s+(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
            if len(cur) == 5:
                v = int(cur["vgpr_count"])
                rows[demangle_kernel(cur["name"])] = {"VGPRs": cur["vgpr_count"], "ScratchSize": cur["private_segment_fixed_size"], "VGPRs Spill": cur["vgpr_spill_count"],
                                                     "SGPRs Spill": cur["sgpr_spill_count"], "Occupancy": str(min(8, 512 // max(8, -(-v // 8) * 8)))}
                cur = None
    return rows


if __name__ == "__main__":
    for k, v in collect().items():
        print(k.ljust(34), "VGPRs", v.get("VGPRs", "?").rjust(4), " scratch", v.get("ScratchSize", "?").rjust(5),
              " spill", v.get("VGPRs Spill", "?").rjust(3), " occupancy", v.get("Occupancy", "?"))

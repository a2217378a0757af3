#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of the shipped libosq_hip.so, read from the code-object metadata
(the AMDGPU msgpack note) of every gfx950 code object bundled in the library.

    python tools/kernel_resources.py [--lib path] [--filter substring] [--json]

Used by tests/test_abi_and_host.py::test_resident_kernels_do_not_spill (CPU: no GPU needed)."""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "outlier_suppression_amd", "libosq_hip.so")


def _demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.split("\n")[:len(names)]
    except Exception:
        return names


def kernel_resources(lib=DEFAULT_LIB):
    """[{name, vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size,
    group_segment_fixed_size, max_flat_workgroup_size}] for every kernel of every gfx950 code object in ``lib``."""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        # a library linked from several objects carries several bundles back to back, each starting with the magic
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(tmp, f"bundle{i}.bin")
            open(part, "wb").write(blob[s:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            co = os.path.join(tmp, f"co{i}.o")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                if line.lstrip().startswith("- ") and key in ("agpr_count", "args"):      # first key of a kernel entry
                    cur = {}
                    rows.append(cur)
                if cur is None:
                    continue
                if key in ("name", "symbol"):
                    cur[key] = val
                elif key in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                             "group_segment_fixed_size", "max_flat_workgroup_size", "wavefront_size"):
                    try:
                        cur[key] = int(val)
                    except ValueError:
                        pass
    rows = [r for r in rows if "name" in r and "vgpr_count" in r]
    for r, d in zip(rows, _demangle([r["name"] for r in rows])):
        r["demangled"] = d
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--filter", default="")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows = [r for r in kernel_resources(a.lib) if a.filter in r["demangled"]]
    if a.json:
        print(json.dumps(rows, indent=1))
        return
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>7} {'wg':>5}  kernel")
    for r in sorted(rows, key=lambda r: r["demangled"]):
        print(f"{r.get('vgpr_count', 0):>5} {r.get('agpr_count', 0):>5} {r.get('sgpr_count', 0):>5} {r.get('vgpr_spill_count', 0):>6} "
              f"{r.get('sgpr_spill_count', 0):>6} {r.get('private_segment_fixed_size', 0):>7} {r.get('group_segment_fixed_size', 0):>7} "
              f"{r.get('max_flat_workgroup_size', 0):>5}  {r['demangled'][:150]}")


if __name__ == "__main__":
    sys.exit(main())

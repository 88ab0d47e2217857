#!/usr/bin/env python3
"""Which shipped kernel instantiations did a run launch?

Compares the kernel names of one or more rocprofv3 ``*_kernel_stats.csv`` / ``*_kernel_trace.csv`` files with the device
stubs of ``liblq_hip.so`` (``nm -C ... | grep __device_stub__``: one stub per __global__ instantiation the library ships)
and prints the instantiations no launch was recorded for.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cov -- python3 -m pytest tests -q -m gpu
    python3 tools/kernel_coverage.py gpurun_out/cov > profiles/r03/kernel_coverage.txt

Exit status 1 when an ``lq::`` instantiation was never launched and is not listed (with a reason) in
``tools/kernel_coverage_allow.txt``.
"""
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "learned_quantization_amd", "csrc", "liblq_hip.so")
ALLOW = os.path.join(ROOT, "tools", "kernel_coverage_allow.txt")


def norm(name: str) -> str:
    """Canonical form of a demangled kernel name: no return type, no argument list, no spaces."""
    name = name.strip().strip('"')
    if name.endswith(".kd"):
        name = name[:-3]
    name = re.sub(r"^void\s+", "", name)
    # cut the argument list: the last top-level '(' outside template brackets
    depth = 0
    cut = None
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    if cut is not None:
        name = name[:cut]
    name = name.replace("__device_stub__", "")
    name = re.sub(r"\((?:int|bool|lq::OpKind)\)", "", name)          # "(int)3" -> "3"
    name = name.replace("true", "1").replace("false", "0")
    return re.sub(r"\s+", "", name)


def shipped():
    out = subprocess.check_output(["nm", "-C", LIB], text=True)
    names = set()
    for line in out.splitlines():
        if "__device_stub__" not in line:
            continue
        sym = line.split(None, 2)[2]
        names.add(norm(sym))
    return names


def launched(paths):
    names = {}
    files = []
    for p in paths:
        if os.path.isdir(p):
            for d, _, fs in os.walk(p):
                files += [os.path.join(d, f) for f in fs if f.endswith("kernel_stats.csv") or f.endswith("kernel_trace.csv")]
        else:
            files.append(p)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                n = r.get("Name") or r.get("Kernel_Name")
                if not n:
                    continue
                calls = int(r["Calls"]) if r.get("Calls") else 1
                k = norm(n)
                names[k] = names.get(k, 0) + calls
    return names, files


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    ship = shipped()
    run, files = launched(sys.argv[1:])
    allow = {}
    if os.path.exists(ALLOW):
        for line in open(ALLOW):
            line = line.rstrip("\n")
            if not line.strip() or line.startswith("#"):
                continue
            k, _, why = line.partition("\t")
            allow[norm(k)] = why.strip()
    lq_run = {k: v for k, v in run.items() if k.startswith("lq::")}
    missing = sorted(ship - set(lq_run))
    unknown = sorted(set(lq_run) - ship)
    print(f"# kernel coverage: {len(ship)} shipped instantiations (nm -C liblq_hip.so | grep __device_stub__), "
          f"{len(ship) - len(missing)} launched, {len(missing)} never launched")
    print(f"# trace files: {len(files)}")
    fam = {}
    for k in ship:
        f = k.split("<")[0]
        a = fam.setdefault(f, [0, 0])
        a[0] += 1
        a[1] += k in lq_run
    print("# family                       shipped launched")
    for f in sorted(fam):
        print(f"# {f:28s} {fam[f][0]:7d} {fam[f][1]:8d}")
    bad = 0
    for k in missing:
        why = allow.get(k)
        if why is None:
            bad += 1
        print(f"UNLAUNCHED {k}" + (f"\t# allowed: {why}" if why else ""))
    for k in unknown:
        print(f"# launched but not among the stubs (name form?): {k} x{lq_run[k]}")
    print(f"# unlaunched without a reason: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

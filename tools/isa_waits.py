#!/usr/bin/env python3
"""Reads the gfx950 assembly of the tile kernel and shows where the compiler waits on the (single, in-order)
vector-memory counter.  CPU only.

  tools/isa_waits.py seq                 the kernel's loads, waits, asm deliveries, barriers and stores in
                                         program order, runs compressed (ldx6@1823 W0x1@3073 asmx4@3078 stx10@3507 ...)
  tools/isa_waits.py reach LINE LABEL N  is LABEL (e.g. .LBB0_5, the loop latch) reachable from assembly line LINE
                                         of /tmp/k_recon_tiles.s without passing an `s_waitcnt vmcnt(<= N)`?
                                         Prints the path: this is how the never-taken bypass edges that make the
                                         compiler believe a prefetch undelivered were found.
The assembly is produced with the Makefile's flags into /tmp/k_recon_tiles.s (kernel k_recon_tiles<false> only).
"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/k_recon_tiles.s"


def build():
    full = "/tmp/vpcc_tiles_full.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-I" + REPO + "/include", "-I" + REPO + "/tmc2-rs_amd/csrc",
           *os.environ.get("VPCC_ISA_EXTRA", "").split(), "-S", "--cuda-device-only", "-o", full, REPO + "/tmc2-rs_amd/csrc/vpcc_tiles.hip"]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    on, lines = False, []
    for l in open(full):
        if l.startswith("_ZN4vpcc13k_recon_tilesILb0"):
            on = True
        if on:
            lines.append(l.rstrip("\n"))
            if "s_endpgm" in l:
                break
    open(OUT, "w").write("\n".join(lines))
    return lines


def seq(L):
    ev = []
    for i, l in enumerate(L):
        t = l.strip()
        if re.match(r"s_waitcnt.*vmcnt", t):
            ev.append((i + 1, "W" + re.search(r"vmcnt\((\d+)\)", t).group(1)))
        elif t.startswith("global_store"):
            ev.append((i + 1, "st"))
        elif t.startswith("scratch_"):
            ev.append((i + 1, "SPILL"))
        elif t.startswith("global_load") or t.startswith("global_atomic"):
            ev.append((i + 1, "ld"))
        elif t.startswith("s_barrier"):
            ev.append((i + 1, "BAR"))
        elif "ASMSTART" in t:
            ev.append((i + 1, "asm"))
        elif re.match(r"^\.LBB0_\d+:.*Loop Header: Depth=1", l):
            ev.append((i + 1, "LOOPHDR"))
    out, prev, cnt, start = [], None, 0, 0
    for ln, k in ev + [(0, None)]:
        if k == prev:
            cnt += 1
            continue
        if prev:
            out.append("%sx%d@%d" % (prev, cnt, start))
        prev, cnt, start = k, 1, ln
    print(" ".join(out))


def reach(L, start, target, maxn):
    labels = {}
    for i, l in enumerate(L):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    seen, stack = set(), [(start, [start])]
    while stack:
        i, path = stack.pop()
        while i < len(L) and i not in seen:
            seen.add(i)
            t = L[i].strip()
            m = re.match(r"^(\.LBB\d+_\d+):", L[i])
            if m and m.group(1) == target:
                print("reachable, via lines", [x + 1 for x in path + [i]])
                return
            w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if w and int(w.group(1)) <= maxn:
                break
            b = re.match(r"s_branch (\S+)", t)
            if b:
                i = labels[b.group(1)]
                path = path + [i]
                continue
            c = re.match(r"s_cbranch_\w+ (\S+)", t)
            if c:
                stack.append((labels[c.group(1)], path + [labels[c.group(1)]]))
            if t.startswith("s_endpgm"):
                break
            i += 1
    print("not reachable")


if __name__ == "__main__":
    L = build()
    if len(sys.argv) >= 5 and sys.argv[1] == "reach":
        reach(L, int(sys.argv[2]), sys.argv[3], int(sys.argv[4]))
    else:
        seq(L)

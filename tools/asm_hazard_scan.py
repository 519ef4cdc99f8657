#!/usr/bin/env python3
"""Scan the built library's gfx950 code for two hazards that the compiler keeps clear of for its own instructions but
cannot see when one side sits inside an `asm` statement (round 4 found both in the resident path's polling code):

  A. a VMEM store of more than 8 bytes followed, less than two wait states later, by a VALU write to one of its data
     registers (the store may send the new value: a granule went out with a clobbered tag, its reader timed out);
  B. a VALU write of an SGPR (v_readlane / v_readfirstlane -- the restore of a spilt SGPR --, a compare's mask) followed,
     less than five wait states later, by a VMEM instruction that reads that SGPR as its base.

The scan is linear (straight-line distance in the disassembly, `s_nop N` = N + 1 wait states, any other instruction = 1,
a conditional branch taken as falling through; an unconditional one ends the look-back), so it can miss a hazard along a
taken branch.

    python tools/asm_hazard_scan.py [path/to/libstorm_hip.so]      exit code 1 if anything is found
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
STORE = re.compile(r"^(global|flat|scratch|buffer)_store_(dwordx[34]|b96|b128)\b")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
VMEM = re.compile(r"^(global|flat|scratch|buffer)_(load|store|atomic)")


def regs(pattern, text):
    out = set()
    for m in pattern.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def operands(ins):
    body = ins.split("//")[0].strip()
    parts = body.split(None, 1)
    return parts[0], ([p.strip() for p in parts[1].split(",")] if len(parts) > 1 else [])


def scan(dis_path):
    found = []
    func = "?"
    window = []  # (mnemonic, operand list, wait states this instruction takes)
    for raw in open(dis_path, errors="replace"):
        line = raw.rstrip()
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            func, window = m.group(1), []
            continue
        if not line.startswith("\t"):
            continue
        mnem, ops = operands(line)
        if not mnem:
            continue
        states = 1
        if mnem == "s_nop":
            states = int(ops[0], 0) + 1
        # B: this VMEM instruction reads SGPRs that a VALU wrote less than five wait states ago
        if VMEM.match(mnem):
            used = set()
            for o in ops:
                used |= regs(SREG, o)
            dist = 0
            for pm, pops, pstates in reversed(window):
                if dist >= 5:
                    break
                if pm.startswith("v_") and pops and (regs(SREG, pops[0]) & used) and not pops[0].startswith("v"):
                    found.append(("B", func, f"{pm} {', '.join(pops)}  ->  {mnem} {', '.join(ops)}  ({dist} wait states)"))
                dist += pstates
        # A: this VALU instruction writes data registers of a wide store less than two wait states back
        if mnem.startswith("v_") and ops and not mnem.startswith("v_cmp") and not mnem.startswith("v_readlane") and not mnem.startswith("v_readfirstlane"):
            written = regs(VREG, ops[0])
            dist = 0
            for pm, pops, pstates in reversed(window):
                if dist >= 2:
                    break
                if STORE.match(pm):
                    data = regs(VREG, pops[0] if pm.startswith("buffer") else pops[1])
                    if data & written:
                        found.append(("A", func, f"{pm} {', '.join(pops)}  ->  {mnem} {', '.join(ops)}  ({dist} wait states)"))
                dist += pstates
        if mnem in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_endpgm"):  # (a conditional branch falls through: one wait state)
            window = []
            continue
        window.append((mnem, ops, states))
        if len(window) > 12:
            window.pop(0)
    return found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stormruler_amd", "libstorm_hip.so")
    tmp = tempfile.mkdtemp(prefix="hazard_scan_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950")))
        if not objs:
            print("no gfx950 code objects found in", lib)
            return 2
        total, stores = [], 0
        for o in objs:
            dis = o + ".dis"
            with open(dis, "w") as f:
                subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", o], check=True, stdout=f, stderr=subprocess.DEVNULL)
            stores += sum(1 for line in open(dis, errors="replace") if STORE.match(line.strip()))
            total += scan(dis)
        for kind, func, what in total:
            print(f"hazard {kind} in {func}: {what}")
        print(f"{len(objs)} code objects, {stores} wide stores scanned: {len(total)} hazards")
        return 1 if total else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())

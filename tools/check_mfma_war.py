#!/usr/bin/env python3
"""Static check of the generated gfx950 assembly: no VALU / LDS / VMEM instruction may WRITE a
register that an inline-asm MFMA issued within the last WINDOW instructions READS as SrcA/SrcB
(hipcc knows nothing about the MFMA inside an asm statement, frees its operand registers at once
and may re-use them for the very next instruction; the hardware is still reading them).
Usage: check_mfma_war.py file.s [kernel-name-substring ...]"""
import re, sys

WINDOW = 2   # instructions after the MFMA that must not write its sources (measured: the next one corrupts)

def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return {(tok[0], r) for r in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()

def check(path, names):
    text = open(path).read()
    bad = []
    for k in re.split(r"\n(?=_ZN9range_hip\w+:)", text):
        name = k.split(":", 1)[0]
        if names and not any(n in name for n in names):
            continue
        lines = [l.split(";")[0].strip() for l in k.splitlines()]
        lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]
        for i, l in enumerate(lines):
            if not l.startswith("v_mfma"):
                continue
            ops = l.split(None, 1)[1].split(", ")
            srcs = regs(ops[1]) | regs(ops[2])
            seen = 0
            for m in lines[i + 1:]:
                if m.startswith(("s_nop", "v_mfma")):
                    break               # a wait state (or the next MFMA) ends the hazard window
                if m.startswith("s_") and not m.startswith("s_waitcnt"):
                    continue            # other SALU does not touch VGPRs (and takes no VGPR time)
                seen += 1
                if seen > WINDOW:
                    break
                if m.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load")):
                    dst = regs(m.split(None, 1)[1].split(", ")[0]) if " " in m else set()
                    if m.startswith("v_") and dst & srcs:
                        bad.append((name, l, m))
    return bad

if __name__ == "__main__":
    bad = check(sys.argv[1], sys.argv[2:])
    for n, a, b in bad[:40]:
        print(n[:60], "|", a, "<=", b)
    print("violations:", len(bad))
    sys.exit(1 if bad else 0)

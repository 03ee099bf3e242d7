#!/usr/bin/env python3
"""Static check of a kernel's ISA for reads of registers that an inline-asm global load has in flight (csrc/patch_gemm.hip keeps
loaded registers in flight across K-steps; the compiler believes an asm's outputs are valid at once and may copy them).
For every `global_load_dword*` (not the LDS-DMA form) the instructions up to the first `s_waitcnt vmcnt(N)` that covers it (N <= VMEM
instructions issued after the load, counted along the fall-through order) are scanned for the destination registers as SOURCE operands.
usage: hipcc ... -S -o k.s file.hip;  python3 tools/check_inflight.py k.s <kernel-name-substring>"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main(path, name):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(name) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = [l.split(";")[0].strip() for l in lines[start:end]]
    body = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
    bad = 0
    for i, l in enumerate(body):
        m = re.match(r"global_load_dword(x\d)?\s+(\S+?),", l)
        if not m or "lds" in l:
            continue
        dest = regs(m.group(2))
        later = 0
        for j in range(i + 1, min(len(body), i + 4000)):
            x = body[j]
            w = re.match(r"s_waitcnt vmcnt\((\d+)\)", x)
            if w and int(w.group(1)) <= later:
                break
            if re.match(r"(global_load|global_store|buffer_|scratch_)", x):
                later += 1
            ops = re.findall(r"v\[\d+:\d+\]|v\d+", x)
            if not ops:
                continue
            is_store = x.startswith(("global_store", "ds_write", "scratch_store", "v_cmp", "s_"))
            srcs = ops if is_store else ops[1:]
            if any(regs(o) & dest for o in srcs):
                print(f"  READ of in-flight {m.group(2)} (load at +{i}) by: {x}   (+{j})")
                bad += 1
            if not is_store and regs(ops[0]) & dest and not x.startswith("global_load"):
                dest = dest - regs(ops[0])                     # overwritten by something else: no longer the load's
                if not dest:
                    break
    # transposed LDS reads (ds_read_b64_tr_b16 is always inline asm here): on every path from the read to the next s_waitcnt
    # lgkmcnt(0) nobody may read OR write the destination -- a write (the compiler reusing the register of a result it believes
    # dead) is clobbered when the read lands.  Paths follow the branches (the loops are rotated: the wait often sits at a label).
    full = [l.split(";")[0].strip() for l in lines[start:end]]
    full = [l for l in full if l and not (l.startswith(".") and not l.endswith(":"))]
    label_at = {l[:-1]: i for i, l in enumerate(full) if l.endswith(":")}
    for i, l in enumerate(full):
        m = re.match(r"ds_read_b64_tr_b16\s+(\S+?),", l)
        if not m:
            continue
        dest = regs(m.group(1))
        seen, todo, hit = set(), [i + 1], None
        while todo and hit is None:
            j = todo.pop()
            while j < len(full) and j not in seen:
                seen.add(j)
                x = full[j]
                if x.endswith(":"):
                    j += 1
                    continue
                if re.match(r"s_waitcnt.*lgkmcnt\(0\)", x):
                    break
                ops = re.findall(r"v\[\d+:\d+\]|v\d+", x)
                if any(regs(o) & dest for o in ops):
                    hit = x
                    break
                br = re.match(r"(s_branch|s_cbranch_\w+)\s+(\S+)", x)
                if br:
                    if br.group(2) in label_at:
                        todo.append(label_at[br.group(2)])
                    if br.group(1) == "s_branch":
                        break
                j += 1
        if hit:
            print(f"  in-flight {m.group(1)} (ds_read_b64_tr_b16 at +{i}) touched by: {hit}")
            bad += 1
    print(f"{name}: {bad} read(s) of in-flight registers")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))

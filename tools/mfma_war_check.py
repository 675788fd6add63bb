#!/usr/bin/env python3
"""Static check of the gfx950 MFMA operand hazard of docs/history/DESIGN_r1-r4.md 3.5 on compiler output: for every v_mfma in a kernel, list the instructions
that WRITE one of its A / B source registers before the matrix instruction can have finished reading them -- i.e. before the next
v_mfma issues (it waits for the pipe) or WAIT wait states of s_nop have passed.  usage: mfma_war_check.py file.s kernel_symbol [WAIT]"""
import re, sys

src, sym = sys.argv[1], sys.argv[2]
WAIT = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = []
for i in range(start + 1, end):
    l = lines[i].split(";")[0].strip()
    if l and not l.startswith(".") and not l.endswith(":"):
        body.append((i + 1, l))


def regs(op):
    op = op.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", op)
    return {int(m.group(1))} if m else set()


def dst_regs(ins):
    mn, _, rest = ins.partition(" ")
    ops = [o for o in rest.split(",")]
    if not ops or mn.startswith(("s_", "buffer_store", "global_store", "ds_write", "flat_store", "global_atomic", "v_cmp", "v_nop", "scratch_store")):
        return set()
    return regs(ops[0])          # first operand = destination for VALU / loads


hits = 0
nm = 0
for k, (ln, ins) in enumerate(body):
    if not ins.startswith("v_mfma"):
        continue
    nm += 1
    ops = ins.split(" ", 1)[1].split(",")
    ab = regs(ops[1]) | regs(ops[2])
    waited = 0
    for ln2, ins2 in body[k + 1:k + 200]:
        if ins2.startswith("v_mfma"):
            break
        m = re.match(r"s_nop (\d+)", ins2)
        if m:
            waited += int(m.group(1)) + 1
            if waited >= WAIT:
                break
            continue
        if ins2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")):
            break
        w = dst_regs(ins2) & ab
        if w:
            hits += 1
            print("line %d: %s\n   line %d writes %s after %d wait states: %s" % (ln, ins, ln2, sorted(w), waited, ins2))
print("%d v_mfma instructions, %d early writes of an A/B operand" % (nm, hits))

# ---- second check: the matrix instruction's RESULT (vDst = SrcC) touched by a non-matrix instruction before NEED wait states have
# passed (1 per issued instruction, n + 1 per s_nop n; an intervening v_mfma counts 8: it occupies the pipe for its passes)
NEED = int(sys.argv[4]) if len(sys.argv) > 4 else 11


def all_regs(ins):
    return set().union(*[regs(o) for o in re.split(r"[ ,]", ins.split(" ", 1)[1])]) if " " in ins else set()


hits2 = 0
for k, (ln, ins) in enumerate(body):
    if not ins.startswith("v_mfma"):
        continue
    d = regs(ins.split(" ", 1)[1].split(",")[0])
    waited = 0
    for ln2, ins2 in body[k + 1:k + 60]:
        if waited >= NEED:
            break
        if ins2.startswith("v_mfma"):
            waited += 8
            continue
        m = re.match(r"s_nop (\d+)", ins2)
        if m:
            waited += int(m.group(1)) + 1
            continue
        if ins2.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            break
        if ins2.startswith("v_") or ins2.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")):
            if all_regs(ins2) & d:
                hits2 += 1
                print("line %d: %s\n   line %d touches the result after %d wait states: %s" % (ln, ins, ln2, waited, ins2))
        waited += 1
print("%d touches of a matrix result inside %d wait states" % (hits2, NEED))

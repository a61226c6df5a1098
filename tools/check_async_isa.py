#!/usr/bin/env python3
"""Build-time check of the kernels that track their own loads (csrc/smst_async.h): in the generated ISA no instruction may read or
overwrite the destination registers of a global load while that load can still be in flight.  The compiler does not know that an
inline-assembly load is pending, so a register copy it inserts on its own (a loop-carried value resolved at the top of the loop
body, a live-range split) would read garbage -- seen once, in the first version of vocoderProduceAligned.

Method: per function, basic blocks and their successors from the labels and branches; forward data flow of the ORDERED list of
outstanding vector-memory operations (inline-assembly loads with their destination registers; the compiler's own loads, which it
waits for itself, and stores as place holders: vmcnt counts them all and retires in order); `s_waitcnt vmcnt(N)` keeps the N youngest.  At a join the longer list wins if the shorter one is its suffix,
otherwise the two are concatenated (conservative).  Any VGPR operand that belongs to an outstanding load is reported.

Which functions: every function of the file whose body contains an inline-assembly vector-memory LOAD (`#ASMSTART` ... `global_load` /
`buffer_load`): picked by content, not by name, so a new user of smst_async.h -- another template instantiation, another translation
unit -- cannot escape the gate.  The build runs the check on EVERY kernel translation unit (csrc/Makefile); a unit without such a
function passes with "0 kernel(s)" unless --expect-some is given (the units known to contain users: a refactoring that loses them all fails).

usage: tools/check_async_isa.py [--expect-some] <file.s> [substring of the mangled kernel names to check instead ...]
exit status 1 if a hazard is found."""
import re
import sys

CAP = 24


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


def has_async_loads(body):
    """does the function contain a vector-memory load inside an inline-assembly section?"""
    in_asm = False
    for _, line in body:
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif in_asm and re.match(r"\s*(global_load|buffer_load|flat_load)", line):
            return True
    return False


def functions(path, wanted=None):
    """(name, body) of the functions to check: those whose name contains one of `wanted`, or -- wanted empty / None -- every function
    with an inline-assembly load."""
    def selected(name, body):
        return any(w in name for w in wanted) if wanted else has_async_loads(body)
    name, body = None, []
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name and selected(name, body):
                yield name, body
            name, body = m.group(1), []
            continue
        if line.startswith(".Lfunc_end"):
            if name and selected(name, body):
                yield name, body
            name, body = None, []
            continue
        if name:
            body.append((ln, line))


def join(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if len(a) < len(b):
        a, b = b, a
    if a[len(a) - len(b):] == b:
        return a
    out = list(a)
    for e in b:
        if e not in out:
            out.append(e)
    return tuple(out[-CAP:])


def check_function(name, body, path):
    blocks, labels, cur = [], {}, []
    in_asm = False
    for ln, line in body:
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        text = line.split(";")[0].strip()
        if not text:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", text)
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            labels[m.group(1)] = len(blocks)
            continue
        if text.startswith("."):
            continue
        cur.append((ln, text, in_asm))
        if text.split()[0] in ("s_branch", "s_endpgm") or text.startswith("s_cbranch"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        last = b[-1][1] if b else ""
        op = last.split()[0] if last else ""
        s = []
        if op == "s_endpgm":
            pass
        elif op == "s_branch":
            s.append(labels.get(last.split()[1]))
        elif op.startswith("s_cbranch"):
            s.append(labels.get(last.split()[1]))
            s.append(i + 1)
        else:
            s.append(i + 1)
        succ.append([x for x in s if x is not None and x < len(blocks)])
    state_in = [None]*len(blocks)
    state_in[0] = ()
    hazards = {}
    work = [0]
    rounds = 0
    while work and rounds < 20000:
        rounds += 1
        i = work.pop()
        st = list(state_in[i])
        for ln, text, from_asm in blocks[i]:
            parts = text.replace(",", " ").split()
            op, args = parts[0], parts[1:]
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", text)
                if m:
                    n = int(m.group(1))
                    st = st[len(st) - n:] if n else []
                continue
            touched = frozenset().union(*[regs(a) for a in args]) if args else frozenset()
            flying = frozenset().union(*st) if st else frozenset()
            hit = touched & flying
            if hit:
                hazards[ln] = "%s:%d: %s: %s touches registers %s of a load that may still be in flight" % (path, ln, name[:60], text, sorted(hit))
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                # only the kernel's own (inline-assembly) loads can be touched too early: the compiler waits for the ones it tracks,
                # which still take their place in the order vmcnt retires in
                st.append(regs(args[0]) if from_asm else frozenset())
            elif op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "scratch_store")):
                st.append(frozenset())
            st = st[-CAP:]
        out = tuple(st)
        for j in succ[i]:
            merged = join(state_in[j], out)
            if merged != state_in[j]:
                state_in[j] = merged
                work.append(j)
    return [hazards[k] for k in sorted(hazards)]


def scratch_users(path, wanted):
    """Kernels among `wanted` (exact names) that spill to scratch: a spill is a vector-memory operation the count-based waits do not know about
    (they only get stricter, never wrong -- the data-flow check above covers scratch operations -- but the kernel then runs with
    waits for everything in flight again: 6.75 -> 9.3 ms per step when it happened)."""
    users, name = [], None
    for line in open(path):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", line)
        if m and name and name in wanted and int(m.group(1)) > 0:
            users.append((name, int(m.group(1))))
    return users


if __name__ == "__main__":
    args = sys.argv[1:]
    expect_some = "--expect-some" in args
    args = [a for a in args if a != "--expect-some"]
    path, wanted = args[0], args[1:]
    total = 0
    chosen = list(functions(path, wanted))
    for name, size in scratch_users(path, [name for name, _ in chosen]):
        print("%s: %s spills %d bytes of scratch per lane (register budget exceeded)" % (path, name[:70], size))
        total += 1
    for name, body in chosen:
        found = check_function(name, body, path)
        for h in found[:12]:
            print(h)
        total += len(found)
    print("async-load ISA check: %s: %d kernel(s) with loads of their own, %d hazard(s)" % (path.rsplit("/", 1)[-1], len(chosen), total))
    sys.exit(1 if total or (expect_some and not chosen) else 0)

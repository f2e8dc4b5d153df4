"""Check the hand-counted vector-memory waits of tools/k6_pc/propagate_pc.hip in the ISA hipcc generated.

The producer waves of the producer / consumer K6 kernel issue their global loads from inline asm and retire them with
hand-counted ``s_waitcnt vmcnt(N)``.  Between such a load and its wait the compiler believes the destination registers
already hold the value, so the kernel is only correct if the compiler neither copies nor touches them in that window.
The source keeps to rules that make this the normal outcome (one asm site per load and period, no initialisation of the
register sets); this tool VERIFIES the outcome on the generated code:

  * every ``; pc-wait <registers>`` names register groups that are the destination of exactly one earlier ``; pc-load``
    in the same loop (walking backwards through the loop body, around the back edge if necessary);
  * no instruction between that load and the wait mentions any of those registers;
  * every ``pc-load`` is retired by a ``pc-wait`` (an orphan means the value was copied and the wait names the copy);
  * no scratch (spill) instruction appears in the kernel: spills are vector-memory operations the counts do not know.

    python tools/k6_pc/verify_pc_asm.py            # compiles tools/k6_pc/propagate_pc.hip (device ISA only) and checks every instance
    python tools/k6_pc/verify_pc_asm.py file.s     # checks an existing listing

Exit code 0 = every instance passes.  tests/test_pc_kernel_asm.py runs it on the library's sources (CPU: hipcc
cross-compiles without a GPU).
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "propagate_pc.hip")

REG_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
REG_ONE = re.compile(r"\bv(\d+)\b")


def regs_of(text):
    """Set of VGPR numbers mentioned in an instruction's operand text."""
    out = set()
    for a, b in REG_RANGE.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in REG_ONE.findall(REG_RANGE.sub(" ", text)):
        out.add(int(a))
    return out


def group_key(tok):
    m = REG_RANGE.fullmatch(tok)
    if m:
        return tuple(range(int(m.group(1)), int(m.group(2)) + 1))
    m = REG_ONE.fullmatch(tok)
    if m:
        return (int(m.group(1)),)
    raise ValueError("not a register operand: %r" % tok)


def compile_listing(defines=()):
    hipcc = "/opt/rocm/bin/hipcc"
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-S",
           "--cuda-device-only", SRC, "-o", out] + ["-D" + d for d in defines]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def kernels(lines):
    """(name, first line, last line) of every propagate_pc_kernel instance in the listing."""
    res, name, start = [], None, None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_ZN\S*propagate_pc_kernel\S*):", ln)
        if m:
            name, start = m.group(1), i
        if name and ln.strip().startswith("s_endpgm"):
            res.append((name, start, i))
            name = None
    return res


def loops(lines, lo, hi):
    """Depth-1 loops of a kernel as (header label, first line, last line), from LLVM's loop comments."""
    res = []
    hdr = None
    for i in range(lo, hi + 1):
        m = re.match(r"^(\.LBB\d+_\d+):.*=>This Loop Header: Depth=1", lines[i])
        if m:
            hdr = (m.group(1)[1:], i)           # "LBB4_18"
            res.append([hdr[0], i, i])
            continue
        if res and re.search(r"(in Loop: Header=|Parent Loop )%s\b" % res[-1][0][1:], lines[i]) or \
                (res and re.search(r"Header=%s\b" % res[-1][0][1:], lines[i])):
            res[-1][2] = i
    # a block's instructions follow its label: extend each loop to the line before the next label after its last block
    out = []
    for name, a, b in res:
        e = b + 1
        while e <= hi and not re.match(r"^\.LBB\d+_\d+:", lines[e]):
            e += 1
        out.append((name, a, e - 1))
    return out


def check_kernel(lines, name, lo, hi):
    errs = []
    for i in range(lo, hi + 1):
        if re.search(r"\b(scratch_|buffer_store|buffer_load)", lines[i]):
            errs.append("line %d: scratch / buffer access in the kernel (spill?): %s" % (i + 1, lines[i].strip()))
    n_wait = n_load = 0
    for lname, a, b in loops(lines, lo, hi):
        body = list(range(a, b + 1))
        loads = {}                                   # line -> register group
        waits = []                                   # (line, [groups])
        for i in body:
            ln = lines[i]
            if "; pc-load" in ln:
                loads[i] = group_key(ln.split()[1].rstrip(","))
            elif "; pc-wait" in ln:
                toks = ln.split("pc-wait", 1)[1].split()
                waits.append((i, [group_key(t) for t in toks]))
        if not loads and not waits:
            continue
        n_load += len(loads)
        retired = set()
        pos = {ln_: k for k, ln_ in enumerate(body)}
        for wl, groups in waits:
            n_wait += 1
            for g in groups:
                # nearest load of exactly this group, walking backwards around the loop
                k = pos[wl]
                found = None
                for step in range(1, len(body)):
                    j = body[(k - step) % len(body)]
                    if j in loads and loads[j] == g:
                        found = (j, step)
                        break
                    if j in loads and set(loads[j]) & set(g):
                        errs.append("%s: wait at line %d names v%s but line %d loads the overlapping v%s"
                                    % (lname, wl + 1, list(g), j + 1, list(loads[j])))
                        break
                if not found:
                    errs.append("%s: wait at line %d names v[%d:%d] which no asm load in the loop writes (copied?)"
                                % (lname, wl + 1, g[0], g[-1]))
                    continue
                j, step = found
                retired.add(j)
                gs = set(g)
                for s in range(1, step):
                    mid = body[(k - s) % len(body)]
                    txt = lines[mid].split(";")[0]
                    if not txt.strip() or txt.strip().startswith((".", "#")) or txt.strip().endswith(":"):
                        continue
                    if regs_of(txt) & gs:
                        errs.append("%s: line %d touches v%s between its load (line %d) and its wait (line %d): %s"
                                    % (lname, mid + 1, sorted(regs_of(txt) & gs), j + 1, wl + 1, txt.strip()))
        for j in loads:
            if j not in retired:
                errs.append("%s: asm load at line %d (v%s) is never named by a wait" % (lname, j + 1, list(loads[j])))
    if n_load == 0 or n_wait == 0:
        errs.append("no pc-load / pc-wait markers found inside a loop (listing without the markers?)")
    return errs, n_load, n_wait


def check_listing(path, verbose=True):
    lines = open(path).read().splitlines()
    ks = kernels(lines)
    if not ks:
        raise SystemExit("no propagate_pc_kernel in %s" % path)
    bad = 0
    for name, lo, hi in ks:
        errs, nl, nw = check_kernel(lines, name, lo, hi)
        if verbose:
            print("%s: %d asm loads, %d waits: %s" % (name, nl, nw, "ok" if not errs else "%d PROBLEMS" % len(errs)))
            for e in errs[:20]:
                print("   ", e)
        bad += len(errs)
    return bad


def main():
    if len(sys.argv) > 1:
        return 1 if check_listing(sys.argv[1]) else 0
    bad = 0
    for defines in ((), ("MMDFN_TUNING",)):
        path = compile_listing(defines)
        print("== build %s" % (" ".join("-D" + d for d in defines) or "(production)"))
        bad += check_listing(path)
        os.unlink(path)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

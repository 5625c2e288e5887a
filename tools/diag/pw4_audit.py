"""Static audit of the hand-placed attention kernel's ISA (developer tool + CPU test): hipcc knows nothing of the instructions inside
an asm statement, so it neither pads hazards in front of them nor keeps its own spill traffic out of the accumulator registers the
statements own. Compiles tools/diag/attn_pw4.hip to assembly and checks
  1. no compiler-generated access (v_accvgpr_*, scratch reload) to the asm-owned a[0:215] (O, Q and BOTH fragment rings: K a192-203,
     V^T a204-215 - fragments are issued in one statement and used in a later one, so their registers must survive in between; the
     bound is read from the kernel's own clobber list UV_ACL_ALL) outside ;;#ASMSTART / ;;#ASMEND,
  2. no vector-ALU write of a register within two wait states in front of an asm MFMA that reads it (VALU write -> MFMA source
     operand needs wait states; an s_waitcnt or any other instruction counts one, s_nop N counts N + 1).
Prints the findings; exit status 1 if there are any."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def compile_to_asm(out):
    sys.path.insert(0, ROOT)
    from univid_amd import build
    src = os.path.join(ROOT, "tools", "diag", "attn_pw4.hip")
    cmd = [build._hipcc(), *build.FLAGS, *build.DIAG_PW4_FLAGS, "-I", build.CSRC, "-S", "--cuda-device-only", "-o", out, src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])


def owned_agprs(src):
    """Number of accumulator registers the kernel's asm statements own = entries of its UV_ACL_ALL clobber list (a0 .. a215)."""
    m = re.search(r"#define UV_ACL_ALL (.*)", open(src).read())
    return 1 + max(int(x) for x in re.findall(r'"a(\d+)"', m.group(1)))


OWNED = owned_agprs(os.path.join(ROOT, "tools", "diag", "attn_pw4.hip"))      # 216: a0 .. a215


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(path):
    lines = open(path).read().split("\n")
    findings = []
    inasm = False
    real = []          # (line number, text, inside asm) of real instructions in order
    for i, l in enumerate(lines):
        t = l.strip()
        if "#ASMSTART" in t:
            inasm = True
            continue
        if "#ASMEND" in t:
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        real.append((i + 1, t, inasm))
        if not inasm:
            m = re.search(r"v_accvgpr_(write|read)_b32 (\S+), (\S+)", t)
            if m:
                a = (m.group(2) if m.group(1) == "write" else m.group(3)).strip(",")
                if a.startswith("a") and int(a[1:]) < OWNED:
                    findings.append(f"line {i + 1}: compiler touches asm-owned {a}: {t}")
            m = re.search(r"scratch_load_dword\w* a(\d+)", t)
            if m and int(m.group(1)) < OWNED:
                findings.append(f"line {i + 1}: compiler reloads into asm-owned a{m.group(1)}: {t}")
    for k, (ln, t, ia) in enumerate(real):
        if not t.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        rd = set()
        for o in ops[1:]:
            rd |= regs(o)
        states, j = 0, k - 1
        while j >= 0 and states < 2:
            pl, pt, pia = real[j]
            m = re.match(r"s_nop (\d+)", pt)
            if pt.startswith("v_") and not pt.startswith("v_mfma"):
                dst = pt.split(None, 1)[1].split(",")[0].strip()
                if regs(dst) & rd:
                    findings.append(f"line {ln}: MFMA reads {sorted(regs(dst) & rd)} written {states} wait state(s) earlier by `{pt}` (line {pl})")
            states += int(m.group(1)) + 1 if m else 1
            j -= 1
    return findings


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(d, "attn_pw4.s")
        if len(sys.argv) <= 1:
            compile_to_asm(out)
        f = audit(out)
    print("\n".join(f[:40]) if f else "pw4 audit: clean")
    sys.exit(1 if f else 0)

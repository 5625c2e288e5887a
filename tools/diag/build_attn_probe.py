"""Builds tools/diag/libunivid_probe{1,2}.so: the product library with flash_attn_fwd12_kernel compiled as a TIMING-ONLY probe (WRONG results):
  probe 1: every second K / V^T fragment read from LDS is skipped, the skipped fragment is a copy of the previous one (half the LDS fragment
           bytes per MFMA - what a 64-queries-per-wave kernel would save);
  probe 2: every read is still issued and waited for, then the same fragments are replaced by the same copies (identical operand DATA to
           probe 1, full LDS traffic).
time(2) - time(1) = what the fragment reads themselves cost; tree - time(2) = the operand-data (power) effect of repeated operands.
Run with tools/attn_so_ab.py <probe .so>.   python tools/diag/build_attn_probe.py"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from univid_amd import build as b

b.build(verbose=False)
objs = [os.path.join(b.OBJDIR, os.path.basename(s)[:-4] + ".o") for s in b.sources() if not s.endswith("attention.hip")]
for n in (1, 2):
    o = os.path.join(HERE, f"attention_probe{n}.o")
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.FILE_FLAGS.get("attention.hip", []), f"-DUV_ATTN_PROBE={n}", "-c", os.path.join(b.CSRC, "attention.hip"), "-o", o])
    lib = os.path.join(HERE, f"libunivid_probe{n}.so")
    subprocess.check_call([b._hipcc(), "-shared", "-fPIC", f"--offload-arch={b.ARCH}", *objs, o, "-o", lib])
    os.remove(o)
    print(lib)

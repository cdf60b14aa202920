"""Static instruction counts of a BP kernel's iteration loop (the innermost loop: Depth=2 blocks of the ISA listing).
usage: python tools/isa_loop_count.py '<template instantiation>' [header]
   e.g. python tools/isa_loop_count.py 'bp_kernel<6,3,2,4,512,6,true,0,1024>' bp_kernel.hip.h
Compiles the one instantiation for gfx950 with the library's flags and prints wave-instructions per thread-iteration by
class.  The BP loops are fully unrolled inside an iteration and their branches are wave-uniform, so the static count of the
loop body is what a wave executes per iteration on the common path (rare-path blocks -- decision flips, LLR stores -- are
included: an upper bound)."""
import os, re, subprocess, sys, tempfile

inst = sys.argv[1]
hdr = sys.argv[2] if len(sys.argv) > 2 else ("bp_class_kernel.hip.h" if "class" in inst else "bp_local_kernel.hip.h" if "local" in inst else "bp_kernel.hip.h")
params = {"bp_kernel": "BpParams", "bp_local_kernel": "BpLocalParams", "bp_class_kernel": "BpClassParams"}[inst.split("<")[0]]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    src = os.path.join(d, "k.hip")
    open(src, "w").write(f'#include "{root}/bp_osd_amd/csrc/{hdr}"\ntemplate __global__ void bposd::{inst}(const bposd::{params});\n')
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", src, "-o", os.path.join(d, "k.o"),
                           "-save-temps"], cwd=d, stderr=subprocess.DEVNULL)
    asm = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]).read()
depth = 0
cnt = {}
for line in asm.split("\n"):
    if line.startswith(".LBB") or line.startswith("; %bb."):
        depth = 0
    mdepth = re.search(r"(?:in Loop: Header=\S+|This (?:Inner )?Loop Header:) Depth=(\d+)", line)
    if mdepth:
        depth = int(mdepth.group(1))
    t = line.strip()
    if not t or t.startswith(";") or t.startswith(".") or depth < 2:
        continue
    op = t.split()[0]
    if op.startswith("v_"):
        cls = "valu_fp64" if ("_f64" in op) else "valu_other"
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
        cls = "wait/nop"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        cls = "branch"
    elif op.startswith("s_barrier"):
        cls = "barrier"
    elif op.startswith("s_"):
        cls = "salu"
    elif op.startswith("ds_"):
        cls = "lds"
    elif op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_") or op.startswith("flat_"):
        cls = "vmem"
    else:
        cls = "other"
    cnt[cls] = cnt.get(cls, 0) + 1
print(inst, " ".join(f"{k}={v}" for k, v in sorted(cnt.items())), "valu_total=%d" % (cnt.get("valu_fp64", 0) + cnt.get("valu_other", 0)))

#!/usr/bin/env python3
"""
Build guard: scans the gfx950 ISA of every non-kernel device function of mcq_kernels.hip for callee-saved SGPRs (s34-39, s48-55, s64-71,
s80-87, s96-103) that the function WRITES without having spilled them in its prologue.  hipcc 7.2 does this when a function grows past the
range of a 16-bit branch offset: branch relaxation runs after prologue / epilogue insertion and scavenges s[98:99] for the
s_getpc_b64 / s_setpc_b64 long branches (round 3: factor() with three factor_t bodies inlined -- active_set()'s loop stride lived in
s[98:99]; found with rocgdb).  Exit code 1 if anything is found.     scripts/check_csr.py [kernels.s | extra hipcc flags]
"""
import re, subprocess, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "global_racetrajectory_optimization_amd", "csrc", "mcq_kernels.hip")
# The ISA comes from csrc/build.sh itself (ASM_OUT=<file>: the same flags as the library, no second recipe to keep in step); given a
# file as first argument this script only scans it, otherwise it compiles the device side once with $HIPCC.
if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
    asm = sys.argv[1]
else:
    asm = os.path.join(tempfile.gettempdir(), "mcq_check_csr.s")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-enable-ipra=0", "-mllvm", "-amdgpu-schedule-metric-bias=0", "--gpu-max-threads-per-block=512",
                    "-S", "--cuda-device-only", "-o", asm, src] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
funcs, cur = {}, None
for ln in open(asm):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1); funcs[cur] = []; continue
    if ln.startswith(".Lfunc_end"):
        cur = None; continue
    if cur:
        funcs[cur].append(ln)
csr = set(range(34, 40)) | set(range(48, 56)) | set(range(64, 72)) | set(range(80, 88)) | set(range(96, 104))
SKIP = ("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_setpc", "s_endpgm", "s_sleep", "s_cmp", "s_bitcmp")
SECOND = ("v_div_scale", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64", "v_mad_i64")
found, nlong = 0, 0
for name, lines in funcs.items():
    saved, written = set(), set()
    kernel = any(".amdhsa_kernel" in l for l in lines) or "kernel" in name
    for ln in lines:
        ln = ln.split(";")[0]
        if "Lpost_getpc" in ln and "s_add_u32" in ln:
            nlong += 1
        m = re.search(r"v_writelane_b32 v\d+, s(\d+),", ln)
        if m:
            saved.add(int(m.group(1))); continue
        toks = ln.strip().split(None, 1)
        if len(toks) < 2 or toks[0].startswith(SKIP) or toks[0] == "v_readlane_b32" and False:
            continue
        op, args = toks
        parts = [a.strip() for a in args.split(",")]
        if op.startswith(SECOND):
            dsts = parts[1:2]
        elif op.startswith("v_") and not op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
            continue
        else:
            dsts = parts[:1]
        for d in dsts:
            m = re.match(r"s\[(\d+):(\d+)\]$", d)
            if m:
                written.update(range(int(m.group(1)), int(m.group(2)) + 1))
            m = re.match(r"s(\d+)$", d)
            if m:
                written.add(int(m.group(1)))
    # v_readlane restores count as writes too: only registers never saved are reported
    bad = sorted((written & csr) - saved)
    if bad and not kernel:
        found += 1
        print(f"{name}: writes callee-saved SGPRs it never saved: {bad}")
print(f"check_csr: {len(funcs)} functions, {nlong} long branches, {found} offender(s)")
sys.exit(1 if found else 0)

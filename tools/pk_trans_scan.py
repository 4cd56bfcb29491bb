"""Scan the gfx950 code objects of the library for a packed-FP32 instruction (v_pk_*_f32, v_pk_mov_b32) that reads the result of a
transcendental instruction (v_rsq / v_rcp / v_sqrt / v_exp / v_log / v_sin / v_cos) within a few instructions of it.

Why: round 5 measured (tools/lab/fpfh_race.py) that on MI355X such a pair can read a STALE register in one half of the
packed operation when other wavefronts keep the SIMD's transcendental pipe busy (the compiler separates the two by one wait state,
s_nop 0; alone on the chip that is enough, next to kernels issuing float64 transcendentals it is not).  pcr_fgr.hip is therefore built
with -fno-slp-vectorize (no packed FP32 at all); this scan is the check for the other translation units.
usage: python tools/pk_trans_scan.py [unit ...]      (needs the .o files of csrc/build.sh)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "point-cloud-registration-with-global-refinement_amd", "csrc")
llvm = "/opt/rocm/lib/llvm/bin"
units = sys.argv[1:] or ["pcr_sort", "pcr_cloud", "pcr_gicp", "pcr_featnn", "pcr_fgr", "pcr_api"]
TRANS = re.compile(r"^v_(rsq|rcp|sqrt|exp|log|sin|cos|rcp_iflag)_(f32|f16|f64|legacy_f32)")
WINDOW = 64
def regs(tok):
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^-?\|?v(\d+)\|?$", tok)
    return {int(m.group(1))} if m else set()
total = 0
for u in units:
    obj = os.path.join(csrc, u + ".o")
    if not os.path.exists(obj): continue
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, f"/tmp/{u}.fat.bin"])
    subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input=/tmp/{u}.fat.bin", f"--output=/tmp/{u}.gfx950.co"])
    dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", f"/tmp/{u}.gfx950.co"], capture_output=True, text=True).stdout
    kernel, hits, pk_total = None, {}, {}
    live = []          # (dest regs, age, mnemonic)
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m: kernel = m.group(1); live = []; continue
        t = line.strip().split("//")[0].strip()
        if not t or kernel is None: continue
        parts = t.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        if not op.startswith(("v_", "s_", "ds_", "global_", "buffer_", "scratch_", "flat_")): continue
        if op.startswith("v_pk_") and ("f32" in op or "mov_b32" in op):
            pk_total[kernel] = pk_total.get(kernel, 0) + 1
            src = set()
            for o in ops[1:]: src |= regs(o)
            for d, age, name in live:
                if d & src: hits.setdefault(kernel, []).append((name, op, age))
        live = [(d, age + 1, name) for d, age, name in live if age + 1 <= WINDOW]
        if op.startswith("v_") and ops:
            dst = regs(ops[0])
            live = [(d - dst, age, name) for d, age, name in live]
            live = [x for x in live if x[0]]
            if TRANS.match(op): live.append((dst, 0, op))
    names = subprocess.run(["c++filt"], input="\n".join(sorted(set(list(hits) + list(pk_total)))), capture_output=True, text=True).stdout.splitlines()
    dem = dict(zip(sorted(set(list(hits) + list(pk_total))), names))
    print(f"{u}: {sum(pk_total.values())} packed-FP32 instructions in {len(pk_total)} kernels; transcendental result -> packed read within {WINDOW} instructions: {sum(len(v) for v in hits.values())}")
    for k, v in hits.items():
        total += len(v)
        print("   ", dem[k][:90], [(a, b, f"{c} apart") for a, b, c in v][:6])
print("TOTAL", total)
sys.exit(1 if total else 0)

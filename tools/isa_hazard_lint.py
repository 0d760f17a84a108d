"""ISA lint for a gfx950 hazard hipcc (ROCm 7.2) does not guard (found in round 5, measured by tools/hwtests/pkmul_mfma.hip):
a packed-fp32 VALU op (v_pk_mul/add/fma_f32) with op_sel:[0,1] - the LOW result reads source 0's low register and source 1's HIGH
register - returns a wrong low result in lanes 48-63 when f16/bf16 matrix instructions (v_mfma_*_f16 / _bf16) are in flight around it:
41 % of the time with an MFMA issued right behind it, ~1e-6 with MFMAs merely nearby.  op_sel:[1,0], [1,1], op_sel_hi:[1,0], the fma forms
[0,0,1] [1,0,0] [0,1,1] [1,1,0] [1,0,1] and v_pk_mov_b32 measured clean, fp32 MFMAs (16x16x4) do not trigger it.  (The lint also flags
fma [0,1,1], which measured clean: it matches on the first two selectors.)

The lint disassembles every code object of the built library and lists the op_sel:[0,1] packed ops of EVERY kernel (round 6: a kernel
without MFMAs of its own shares its SIMD with other kernels' waves - profiles/r06_lanes_48_63.md), tagged "own-mfma" when the kernel
itself contains f16/bf16 MFMAs and "foreign-mfma" otherwise.  Exit status 1 when there is one.
usage: isa_hazard_lint.py [build dir or .so/.o files]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"\bv_pk_(mul|add|fma)_f32\b.*op_sel:\[0,1[\],]")
MFMA16 = re.compile(r"\bv_mfma_\w+_(f16|bf16)\b|\bv_smfmac_")
KERNEL = re.compile(r"^[0-9a-f]+ <(.+)>:")


def code_objects(path):
    """Device ELF files of every offload bundle inside an object / shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return
        data = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), data)]
        for i, s in enumerate(starts):
            e = starts[i + 1] if i + 1 < len(starts) else len(data)
            part = os.path.join(tmp, f"bundle{i}.bin")
            open(part, "wb").write(data[s:e])
            out = os.path.join(tmp, f"dev{i}.co")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True)
            if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
                yield subprocess.run([f"{LLVM}/llvm-objdump", "-d", out], capture_output=True, text=True).stdout


def lint_text(text):
    findings, kernel, lines = [], None, []

    def flush():
        if kernel is None:
            return
        # every kernel counts: on a shared SIMD the f16 / bf16 MFMA in flight can belong to ANOTHER kernel's wave (a second stream),
        # so a VALU-only kernel is exposed too.  dist = distance to the kernel's own nearest f16/bf16 MFMA, -1 when it has none
        # (severity tag only: "own-mfma" sites fail on their own, "foreign-mfma" sites need a co-resident MFMA kernel).
        mf = [i for i, l in enumerate(lines) if MFMA16.search(l)]
        for i, l in enumerate(lines):
            if PK.search(l):
                dist = min(abs(i - j) for j in mf) if mf else -1
                findings.append((kernel, l.split("//")[0].strip(), dist))

    for line in text.splitlines():
        m = KERNEL.match(line)
        if m:
            flush()
            kernel, lines = m.group(1), []
        elif kernel is not None and line.startswith("\t"):
            lines.append(line)
    flush()
    return findings


def lint_paths(paths):
    out = []
    for p in paths:
        for text in code_objects(p):
            for k, ins, dist in lint_text(text):
                out.append((os.path.basename(p), k, ins, dist))
    return out


def main():
    args = sys.argv[1:] or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffab-pytorch_amd", "lib", "libdiffab_hip.so")]
    paths = []
    for a in args:
        if os.path.isdir(a):
            paths += sorted(os.path.join(a, f) for f in os.listdir(a) if f.endswith((".o", ".so")))
        else:
            paths.append(a)
    found = lint_paths(paths)
    for f, k, ins, dist in found:
        tag = f"own-mfma, nearest f16/bf16 MFMA {dist} instructions away" if dist >= 0 else "foreign-mfma: the kernel has no f16/bf16 MFMA of its own"
        print(f"{f}: {k[:90]}\n    {ins}    ({tag})")
    print(f"{len(found)} op_sel:[0,1] packed-fp32 ops in {len(set((f, k) for f, k, _, _ in found))} kernels")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())

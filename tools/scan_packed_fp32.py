"""Count the packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in the gfx950 code of the built kernel library: there must
be none (round 6: their results go wrong beside another kernel's f16 MFMAs -- __graft_entry__.py, profiles/r06_packed_fp32_beside_mfma.txt).
    python tools/scan_packed_fp32.py [liblgd_hip.so]      -> prints the count per kernel object, exits 1 if any"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def scan(lib):
    """-> {kernel symbol: packed fp32 instruction count} over the gfx950 code objects bundled in `lib` (kernels without any are left out)"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]   # one bundle per translation unit
        dis = ""
        for i, st in enumerate(starts):
            part, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "b%d.co" % i)
            open(part, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   "--input=" + part, "--output=" + co], stderr=subprocess.DEVNULL)
            dis += subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    cur = None
    n_instr = 0
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            continue
        if cur and re.search(r"\bv_[a-z0-9_]+", line):
            n_instr += 1
            if re.search(r"\bv_pk_(add|mul|fma)_f32\b", line):
                out[cur] = out.get(cur, 0) + 1
    return out, n_instr


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "lgd_amd", "_lib", "liblgd_hip.so")
    found, n = scan(lib)
    print("%s: %d vector instructions scanned, %d packed fp32 among them" % (lib, n, sum(found.values())))
    for k, v in sorted(found.items(), key=lambda kv: -kv[1])[:20]:
        print("  %5d  %s" % (v, k))
    sys.exit(1 if found else 0)

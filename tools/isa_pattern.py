"""One character per instruction class for a kernel of the built library -- M = MFMA, r / W = LDS read / write, g / S =
global load / store, w = s_waitcnt with lgkmcnt, v = any other s_waitcnt, x = transcendental / division, / = branch,
|B| = block barrier -- so that serialised "read, wait, use" chains ("rwMrwMrwM", "gvgvgv") stand out from block-form
code ("rrrrrrrr w MMMMMMMM"). This view found the round-3 stalls of the 64-wide PPO body (every MFMA behind its own LDS
reads; row loads behind a branch, one memory round trip per trip) and the serial bias-gradient column sums of the 32-wide
chain. CPU only (llvm-objdump on the built .so).
Usage: python tools/isa_pattern.py <kernel-name substring> [max lines]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(HERE), "imitation_amd", "libimitation_hip.so")


def disassemble(sub: str):
    tmp = tempfile.mkdtemp()
    so = os.path.join(tmp, "lib.so")
    shutil.copy(LIB, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
    for f in sorted(os.listdir(tmp)):
        if "hipv4" not in f:
            continue
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--demangle", os.path.join(tmp, f)],
                             capture_output=True, text=True, check=True).stdout
        out, name = [], None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                if name is not None:
                    yield name, out
                name, out = (m.group(1), []) if sub in m.group(1) else (None, [])
            elif name is not None:
                out.append(line)
        if name is not None:
            yield name, out
    shutil.rmtree(tmp, ignore_errors=True)


def classify(line: str) -> str:
    if "v_mfma" in line:
        return "M"
    if "s_waitcnt" in line:
        return "w" if "lgkmcnt" in line else "v"
    if "ds_read" in line or "ds_load" in line:
        return "r"
    if "ds_write" in line or "ds_store" in line:
        return "W"
    if "global_load" in line or "buffer_load" in line or "flat_load" in line:
        return "g"
    if "global_store" in line or "buffer_store" in line or "flat_store" in line or "global_atomic" in line:
        return "S"
    if "s_barrier" in line:
        return "\n|B|\n"
    if re.search(r"v_(exp|rcp|sqrt|rsq|log|div)", line):
        return "x"
    if "s_cbranch" in line:
        return "/"
    return ""


if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 else "ppo_epoch_persistent_kernel<64, true>"
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    for name, lines in disassemble(sub):
        print(f"== {name}: {len(lines)} instructions")
        pat = "".join(classify(ln) for ln in lines)
        rows = [seg[i:i + 200] for seg in pat.split("\n") for i in range(0, max(len(seg), 1), 200)]
        print("\n".join(rows[:limit]))

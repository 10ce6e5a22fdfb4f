"""Register / scratch / LDS use of every kernel in the built libimitation_hip.so, from the code objects' notes
(`llvm-objdump --offloading` + `llvm-readelf --notes`). Usage: python tools/kernel_resources.py [substring]
`spills(path)` is what tests/test_host_logic.py::test_production_kernels_do_not_spill reads."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "imitation_amd", "libimitation_hip.so")


def kernel_notes(lib: str = LIB):
    """[{name (demangled), vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] for every kernel of `lib`."""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)],
                                 capture_output=True, text=True, check=True).stdout
            for blk in txt.split("- .agpr_count")[1:]:
                g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                out.append(dict(mangled=name, vgpr=g("vgpr_count"), agpr=int(re.match(r":\s+(\d+)", blk).group(1)),
                                sgpr=g("sgpr_count"), vgpr_spill=g("vgpr_spill_count"),
                                sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                                lds=g("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(k["mangled"] for k in out), capture_output=True,
                           text=True).stdout.splitlines()
    for k, n in zip(out, names):
        k["name"] = n
    return out


if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 else ""
    for k in kernel_notes():
        if sub in k["name"]:
            short = re.sub(r"\(anonymous namespace\)::", "", k["name"]).split("(")[0]
            print(f"{short[:70]:70s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} "
                  f"spill v{k['vgpr_spill']} s{k['sgpr_spill']} scratch {k['scratch']} lds {k['lds']}")

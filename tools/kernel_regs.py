"""Register / scratch use of the built kernels (from the code object's notes): python tools/kernel_regs.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
llvm = "/opt/rocm/lib/llvm/bin"
obj = os.path.join(ROOT, "diaglib_amd", "_obj", "hip_engine.hip.o")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run([llvm + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([llvm + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                    "--output=" + co, "--unbundle"], check=True)
    notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    names = subprocess.run(["c++filt"], input="\n".join(re.findall(r"\.name:\s+(\S+)", notes)), capture_output=True, text=True).stdout.split("\n")
ks = notes.split("- .agpr_count:")[1:]
for k, dn in zip(ks, names):
    if sys.argv[1:] and not all(a in dn for a in sys.argv[1:]):
        continue
    g = lambda f: re.search(r"\." + f + r":\s+(\d+)", k).group(1)
    print(f"{dn[:110]:110s} agpr {k.split()[0]:>3s} vgpr {g('vgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")

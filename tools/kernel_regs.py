"""VGPR / SGPR / LDS / scratch of every kernel of the product build (or --probe: the instrumented one),
read from the ISA the compiler emits: python tools/kernel_regs.py [--probe] [name-filter]"""
import glob, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
extra = ["-DFFQ_PROBES=1"] if "--probe" in sys.argv else []
flt = [a for a in sys.argv[1:] if not a.startswith("--")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", '-DFFQ_BUILD_ID="x"',
                "--save-temps", "-o", d + "/lib.so", R + "/fastq-and-furious_amd/csrc/ffq_hip.hip", "-lz"] + extra,
               cwd=d, check=True, stderr=subprocess.DEVNULL)
s = open(glob.glob(d + "/*gfx950*.s")[0]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, flags=re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1)
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and not any(f in dn for f in flt):
        continue
    print("%-52s vgpr %3s sgpr %3s lds %6s scratch %s" % (dn[:52], g("next_free_vgpr"), g("next_free_sgpr"),
                                                           g("group_segment_fixed_size"), g("private_segment_fixed_size")))

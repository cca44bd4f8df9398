"""Which kernels of a source file changed between two builds?  No GPU needed.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idiffuman4d_amd/csrc -S --cuda-device-only <old>.hip -o old.s   (same for new.s)
    python tools/dev/isa_diff.py old.s new.s
Prints SAME / DIFF per kernel (labels, comments and blank space normalised) and NEW for kernels only the second listing has.  Used in
round 3 to show that adding the 160-wide tiles left the ISA of all 42 shipped GEMM / convolution kernels untouched."""
import re
import sys


def kernels(path):
    text = open(path).read()
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', text, re.S | re.M):
        body = re.sub(r';.*', '', m.group(2))
        body = re.sub(r'\.LBB\d+_', '.LBB_', body)
        out[m.group(1)] = '\n'.join(line.strip() for line in re.sub(r'[ \t]+', ' ', body).splitlines())
    return out


def norm(name):
    """A trailing `false` template argument added since (round 4: PAR = false on the GEMM / convolution kernels) does not make a
    different kernel: gemm_kernel_glds<128,128,2,2,false> of the old listing is <128,128,2,2,false,false> of the new one."""
    name = re.sub(r'Li0E(EEvNS_10GemmParamsE)$', r'\1', name)   # round 5: PAR became an int (0 = fast) ...
    name = re.sub(r'Lb0E(EEvNS_10AttnParamsE)$', r'\1', name)   # ... and attn_kernel got H16 = false
    return re.sub(r'Lb0E(EEvNS_10GemmParamsE)$', r'\1', name)


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
bn = {}
for n in b:
    bn.setdefault(n, n)
    bn.setdefault(norm(n), n)
seen = set()
for n in sorted(a):
    m = bn.get(n)
    seen.add(m)
    print(("SAME " if a[n] == b[m] else "DIFF ") if m else "GONE ", n[:120])
for n in sorted(set(b) - seen):
    print("NEW  ", n[:120])

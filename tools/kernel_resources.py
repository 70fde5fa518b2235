"""Per-kernel register / spill / LDS table of a compiled HIP object (no GPU needed):
    python tools/kernel_resources.py simclr_amd/csrc/build/conv.o [filter-substring ...]
Reads the AMDGPU metadata note of the bundled gfx950 code object."""
import re
import subprocess
import sys
import tempfile
import os

LLVM = '/opt/rocm/lib/llvm/bin'


def resources(obj):
    with tempfile.TemporaryDirectory() as d:
        co = os.path.join(d, 'dev.co')
        fat = os.path.join(d, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj])
        subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--unbundle',
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--input=' + fat, '--output=' + co],
                              stderr=subprocess.DEVNULL)
        txt = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], text=True)
    out = []
    for blk in txt.split('- .agpr_count:')[1:]:
        g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
        name = g('name')
        try:
            name = subprocess.check_output([os.path.join(LLVM, 'llvm-cxxfilt')], input=name, text=True).strip()
            name = name.replace('(anonymous namespace)::', '').replace('unsigned short', 'bf16')
        except Exception:
            pass
        out.append(dict(name=name, agpr=blk.split()[0], vgpr=g('vgpr_count'), sgpr=g('sgpr_count'),
                        spill=g('vgpr_spill_count'), lds=g('group_segment_fixed_size'), scratch=g('private_segment_fixed_size')))
    return out


if __name__ == '__main__':
    flt = sys.argv[2:]
    for r in resources(sys.argv[1]):
        if flt and not all(f in r['name'] for f in flt):
            continue
        print('vgpr %4s agpr %4s sgpr %4s spill %4s scratch %5s lds %6s  %s' % (r['vgpr'], r['agpr'], r['sgpr'], r['spill'], r['scratch'], r['lds'], r['name'][:150]))

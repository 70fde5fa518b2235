"""Where do a kernel's scratch (spill) accesses and waits sit relative to its loops?
    llvm-objdump -d dev.co > conv.s;  python tools/isa_loops.py conv.s <kernel-name-substring>
Finds every backward branch of the kernel (= a loop [target, branch]) and prints per loop: instructions, #mfma, #scratch
loads / stores, #s_waitcnt with vmcnt, #ds_read, #LDS-DMA, #s_barrier; innermost loops first.  Then the whole kernel."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    ins = []        # (offset, text)
    name = None
    base = None
    for l in open(path):
        m = re.match(r'^([0-9a-f]+) <(.+)>:$', l)
        if m:
            if name is not None:
                break
            if key in m.group(2):
                name, base = m.group(2), int(m.group(1), 16)
            continue
        if name is None:
            continue
        m = re.match(r'^\s+(.*?)\s*// ([0-9A-F]+):', l)
        if m:
            ins.append((int(m.group(2), 16) - base, m.group(1), l))
    print(name, len(ins), 'instructions')

    def stats(sub):
        t = '\n'.join(x[1] for x in sub)
        return dict(n=len(sub), mfma=len(re.findall(r'v_mfma', t)), sld=len(re.findall(r'scratch_load', t)),
                    sst=len(re.findall(r'scratch_store', t)), vm=len(re.findall(r's_waitcnt[^\n]*vmcnt', t)),
                    vm0=len(re.findall(r's_waitcnt[^\n]*vmcnt\(0\)', t)),
                    ds=len(re.findall(r'ds_read', t)), glds=len(re.findall(r'global_load_lds|buffer_load\S* .* lds', t)),
                    bar=len(re.findall(r's_barrier', t)), valu=len(re.findall(r'^v_(?!mfma)', t, flags=re.M)))
    loops = []
    for off, text, raw in ins:
        m = re.match(r's_c?branch\S*\s', text)
        if m:
            t = re.search(r'\+0x([0-9a-f]+)>', raw)
            if t and int(t.group(1), 16) <= off:
                loops.append((int(t.group(1), 16), off))
    loops.sort(key=lambda ab: ab[1] - ab[0])
    for a, b in loops:
        sub = [x for x in ins if a <= x[0] <= b]
        s = stats(sub)
        if s['mfma'] or s['sld'] or s['sst']:
            print('loop +0x%05x..+0x%05x  n=%5d mfma=%4d valu=%4d scratch ld/st=%3d/%3d vmcnt=%2d (vmcnt(0)=%2d) ds_read=%3d glds=%2d barrier=%2d' % (
                a, b, s['n'], s['mfma'], s['valu'], s['sld'], s['sst'], s['vm'], s['vm0'], s['ds'], s['glds'], s['bar']))
    s = stats(ins)
    print('kernel                      n=%5d mfma=%4d valu=%4d scratch ld/st=%3d/%3d vmcnt=%2d (vmcnt(0)=%2d) ds_read=%3d glds=%2d barrier=%2d' % (
        s['n'], s['mfma'], s['valu'], s['sld'], s['sst'], s['vm'], s['vm0'], s['ds'], s['glds'], s['bar']))


if __name__ == '__main__':
    main()

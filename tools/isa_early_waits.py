"""Static check of a field kernel's ISA for EXPOSED memory round trips: per inter-barrier segment (= one weight chunk), the vector
loads issued in the segment that are waited for (s_waitcnt vmcnt(N) with N small enough to cover them) before the segment's MFMAs
have run.  Found the forward-direction sweep's early multiply in round 4.
Usage: python tools/isa_early_waits.py ho-nerf_amd/csrc/build/hn_field2_hand_adj.o [kernel-name-filter]"""
import os, re, subprocess, sys, tempfile
LLVM = '/opt/rocm/lib/llvm/bin'


def scan(obj, flt=''):
    """-> {kernel name: [(segment, mfma before the wait, mfma of the segment, loads, vmcnt), ...]}"""
    obj = os.path.abspath(obj)
    with tempfile.TemporaryDirectory() as T:
        subprocess.run(['cp', obj, os.path.join(T, 'x.o')], check=True)
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', 'x.o'], cwd=T, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(T) if 'gfx950' in f]
        dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', os.path.join(T, dev[0])], capture_output=True, text=True).stdout
    out = {}
    for fn in re.split(r'\n(?=[0-9a-f]{16} <)', dis):
        m = re.match(r'[0-9a-f]{16} <(\S+)>:', fn)
        if not m:
            continue
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().split('(')[0]
        if flt not in name:
            continue
        lines = [l.split('//')[0].strip() for l in fn.split('\n')[1:]]
        segs, cur = [], []
        for l in lines:
            if 's_barrier' in l:
                segs.append(cur)
                cur = []
            cur.append(l)
        segs.append(cur)
        bad = []
        for si, sg in enumerate(segs):
            n_mfma_total = sum('v_mfma' in l for l in sg)
            if n_mfma_total < 16:
                continue
            loads_before = 0     # non-LDS vector loads issued so far in this segment
            later = 0            # VMEM ops issued after the last such load (any kind)
            mf = 0
            for l in sg:
                if 'v_mfma' in l:
                    mf += 1
                is_vmem = l.startswith('buffer_') or l.startswith('global_') or l.startswith('scratch_')
                if is_vmem:
                    if 'load' in l and 'lds' not in l and not l.startswith('scratch_'):
                        loads_before += 1
                        later = 0
                    else:
                        later += 1
                w = re.search(r'vmcnt\((\d+)\)', l)
                if w and loads_before and int(w.group(1)) <= later and mf * 4 < n_mfma_total:
                    bad.append((si, mf, n_mfma_total, loads_before, int(w.group(1))))
                    break
        out[name] = (len(segs), bad)
    return out


if __name__ == '__main__':
    for name, (nseg, bad) in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '').items():
        print('%-44s %4d segments, %3d with a load of the segment waited for in its first quarter' % (name[:44], nseg, len(bad)))
        if bad:
            print('   segments (index: mfma before the wait / of, loads, vmcnt):', ' '.join('%d:%d/%d,%d,%d' % b for b in bad[:40]))

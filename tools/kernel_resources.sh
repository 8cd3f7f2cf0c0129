#!/bin/bash
# Register / scratch use of the kernels in an object file: name, vgprs, agprs, spilled vgprs, scratch bytes.
# Usage: tools/kernel_resources.sh ho-nerf_amd/csrc/build/hn_field2_hand.o [filter]
F=$(readlink -f $1)
P=${2:-.}
T=$(mktemp -d)
cp $F $T/x.o
(cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null 2>&1)
D=$(ls $T/*gfx950* 2>/dev/null | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $D 2>/dev/null | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)',blk) or [0,'?'])[1]
    name=subprocess.run(['c++filt',g('name')],capture_output=True,text=True).stdout.strip()
    print('%-56s vgpr %s agpr %s spill %s scratch %s B' % (name.split('(')[0][:56], g('vgpr_count'), blk.split()[0], g('vgpr_spill_count'), g('private_segment_fixed_size')))
" | grep -E "$P"
rm -rf $T

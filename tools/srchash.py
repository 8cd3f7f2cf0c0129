"""Hash of the sources a profiled kernel is built from: stamped into profiles/rNN/pmc_*.json by tools/pmc_summary.py and
compared by bench.py, which quotes a committed PMC figure only while the kernel's sources are the ones it was collected on."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ho-nerf_amd', 'csrc')
# what decides the traffic of the f16x3 hand evaluation kernel: the kernel, its building blocks, the weight-stream layout, the flags
HAND_EVAL_SOURCES = ('hn_field2_hand.hip', 'hn_field2_hand_adj.inl', 'hn_mlp2.h', 'hn_common.h', 'hn_pack2.hip', 'Makefile')


def source_hash(names=HAND_EVAL_SOURCES):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), 'rb') as f:
            h.update(n.encode() + b'\0' + f.read() + b'\0')
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(source_hash())

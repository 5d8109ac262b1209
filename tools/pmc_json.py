"""profiles/TAG_pmc.json: HBM traffic per launch of the roofline kernels from the rocprofv3 PMC passes of `bench.py --roofline-only`
(tools/collect_profiles.sh: separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs), together with a hash of the kernel sources the
passes were collected on.  bench.py reads the newest file whose hash matches the sources it runs, and reports `traffic: null` otherwise.

    python tools/pmc_json.py TAG [dtype bs height width]      (reads gpurun_out/TAG_fetch, gpurun_out/TAG_write)

HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE, both counters in KiB: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for 16-B-per-lane streaming stores and for float atomics."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ('tcct_amd/csrc/conv_mfma.hip', 'tcct_amd/csrc/common.h')     # what the dominant kernel is compiled from


def source_hash(root=ROOT):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def newest_matching(dtype, bs, height, width, root=ROOT):
    """-> (dict kernel-name -> hbm bytes per launch, file name) of the newest profiles/*_pmc.json collected on the current kernel
    sources at this shape, or (None, reason)"""
    pdir = os.path.join(root, 'profiles')
    cands = sorted((f for f in os.listdir(pdir) if f.endswith('_pmc.json')), key=lambda f: os.path.getmtime(os.path.join(pdir, f)), reverse=True)
    if not cands:
        return None, 'no profiles/*_pmc.json'
    cur = source_hash(root)
    for f in cands:
        d = json.load(open(os.path.join(pdir, f)))
        if d.get('source_sha') == cur and d.get('shape') == [dtype, bs, height, width]:
            return {k: v['hbm_bytes'] for k, v in d['kernels'].items()}, f
    return None, f'kernel sources changed since {cands[0]} was collected (or other shape): re-run tools/collect_profiles.sh'


def main():
    tag = sys.argv[1]
    shape = ['bf16', 8, 800, 1100] if len(sys.argv) < 6 else [sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])]
    sha = sys.argv[6] if len(sys.argv) > 6 else source_hash()
    G = os.path.join(ROOT, 'gpurun_out')
    pm = {}
    for kind in ('fetch', 'write'):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f'{G}/{tag}_{kind}/{kind}_counter_collection.csv')):
            d[r['Kernel_Name']].append(float(r['Counter_Value']))
        pm[kind] = {k: (sum(v) / len(v), len(v)) for k, v in d.items()}
    kernels = {}
    for name, (f, n) in pm['fetch'].items():
        if name in pm['write'] and name.startswith(('void k_', 'k_')):
            w = pm['write'][name][0]
            kernels[name.split('(')[0]] = {'fetch_kib_raw': round(f, 1), 'write_kib': round(w, 1), 'launches': n,
                                           'hbm_bytes': int(round((2 * f + w) * 1024))}
    out = {'tag': tag, 'source_sha': sha, 'sources': list(KERNEL_SOURCES), 'shape': shape, 'command': 'rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python3 bench.py --roofline-only',
           'formula': 'hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 wide-read correction, MI355X_MICROARCH.md)', 'kernels': kernels}
    path = os.path.join(ROOT, 'profiles', f'{tag}_pmc.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(path, len(kernels), 'kernels')


if __name__ == '__main__':
    main()

"""Scan the loops of every kernel of the library for instruction classes that usually mean wasted issue slots (round 6): IEEE division sequences
(v_div_scale / v_div_fmas / v_div_fixup), fp64 arithmetic, integer multiplies (64-bit address arithmetic), register spills (scratch_*).  A hit is a question, not a
verdict: a run-time `switch` over activation kinds keeps every branch in the loop body although one runs, so read the loop before acting (and tools/step_valu.sh tells which
kernels spend their cycles issuing VALU instructions at all).

    python tools/isa_scan.py [file.hip ...]      (default: every tcct_amd/csrc/*.hip; compiles with --save-temps into a temporary directory, ~1-2 min per file)"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'tcct_amd', 'csrc')
SLOW = ('v_div_scale_f32', 'v_div_fmas_f32', 'v_div_fixup_f32', 'v_sqrt_f32', 'v_div_scale_f64', 'v_rcp_f64', 'v_rsq_f64', 'v_sqrt_f64', 'v_fma_f64', 'v_mul_f64', 'v_add_f64',
        'v_cvt_f64_f32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_u64_u32', 'v_log_f32', 'scratch_')


def scan(asm):
    rows = []
    s = open(asm).read()
    for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', s):
        name, i0 = m.group(1), m.end()
        body = s[i0:s.find('.Lfunc_end', i0)]
        tot, loop, nloop = collections.Counter(), False, 0
        for line in body.split('\n'):
            t = line.strip()
            if t.startswith('.LBB'):
                loop = False
                continue
            if t.startswith(';') and 'Loop' in t:
                loop = True
            if not loop or not t or t.startswith(('.', ';')):
                continue
            nloop += 1
            op = t.split()[0]
            for sl in SLOW:
                if op.startswith(sl):
                    tot[sl] += 1
        if tot:
            rows.append((sum(tot.values()), name, nloop, dict(tot)))
    return rows


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            base = os.path.splitext(os.path.basename(f))[0]
            r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-c', os.path.abspath(f), '-o', base + '.o', '--save-temps'],
                               cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            asm = os.path.join(tmp, base + '-hip-amdgcn-amd-amdhsa-gfx950.s')
            if r.returncode != 0 or not os.path.exists(asm):
                print(f'# {base}: compile failed\n' + r.stdout[-800:])
                continue
            rows += [(n, base, name, nl, d) for n, name, nl, d in scan(asm)]
    rows.sort(reverse=True)
    names = '\n'.join(r[2] for r in rows)
    dem = subprocess.run(['c++filt'], input=names, stdout=subprocess.PIPE, text=True).stdout.split('\n')
    print('| hits | file | kernel | loop instructions | classes |\n|---|---|---|---|---|')
    for (n, base, _, nl, d), dn in zip(rows, dem):
        print(f'| {n} | {base} | `{dn[:100]}` | {nl} | {d} |')


if __name__ == '__main__':
    main()

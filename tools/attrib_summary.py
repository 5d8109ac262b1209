"""Per-level and per-branch tables of the training step from tools/attrib_trace.sh: time, launches, real HBM bytes (PMC) and SURVEY 8(d) model bytes.

    python tools/attrib_summary.py TAG  ->  gpurun_out/TAG_attrib_summary.md (+ TAG_attrib_calls.csv: one row per C-ABI call with its dispatches)

The dispatch sequence of each pass (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE: three runs of the same deterministic program) is cut at the `k_marker`
dispatches, whose grid size carries the id of the C-ABI call that follows (tools/attrib_trace.py logs what that call was).  Dispatches behind a marker
that are not this library's (torch's autograd accumulation adds, memsets) are counted with the call they follow.

Model bytes: SURVEY 8(d)'s layer-granular traffic model (every convolution / linear layer reads its input once and writes its output once, everything else
fused; backward = 2 x forward), split by level here -- elements per level-0 pixel, forward: CNN 419 / 96 / 24 / 6 / 1.5 (first layer 35 + six 32->32
convolutions per level), ViT 347 (stem 35 + stage 0) / 134 / 44.5 / 13.75, fusion + decoder + heads 165 / 105.25 / 28.31 / 7.58 / 1.5 by the resolution of
a layer's output; sum 1393.4 = SURVEY's figure."""
import collections, csv, glob, json, math, os, re, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
log = json.load(open(os.path.join(root, f'{tag}_attrib_calls.json')))
calls = {c['id']: c for c in log['calls']}
NS, BS = log['steps'], log['bs']
P0 = BS * 800 * 1104

MODEL = {('CNN', 0): 419.0, ('CNN', 1): 96.0, ('CNN', 2): 24.0, ('CNN', 3): 6.0, ('CNN', 4): 1.5,
         ('ViT', 1): 347.0, ('ViT', 2): 134.0, ('ViT', 3): 44.5, ('ViT', 4): 13.75,
         ('fusion+decoder+heads', 0): 165.0, ('fusion+decoder+heads', 1): 105.25, ('fusion+decoder+heads', 2): 28.3125, ('fusion+decoder+heads', 3): 7.578, ('fusion+decoder+heads', 4): 1.5}
GB_PER_ELEM = 883200 * BS * 2 * 3 / 1e9          # bf16, forward + backward (the two input-layer dgrads that do not exist are 0.3 % of the total)


def pixels(c):
    i = c['ints']
    n = i.get('N') if 'H' in i else None
    b = i.get('B', n if n is not None else 1)
    cand = []
    if 'M' in i:
        cand.append(i['M'])
    if 'H' in i and 'W' in i:
        cand.append(b * i['H'] * i['W'])
    if 'Ho' in i and 'Wo' in i:
        cand.append(b * i['Ho'] * i['Wo'])
    if 'B' in i and 'N' in i and 'H' not in i:
        cand.append(i['B'] * i['N'])
    return max(cand) if cand else None


def level_of_pixels(p):
    if not p:
        return None
    return min(4, max(0, round(math.log(P0 / p, 4))))


def classify(c):
    s, sym = c['scope'], c['sym']
    lv = level_of_pixels(pixels(c))
    m = re.search(r'base_cnn\.path_estan\.(\d)', s)
    if m:
        return 'CNN', int(m.group(1))
    if s.startswith('base.base_cnn'):
        return 'CNN', 0 if lv is None else lv
    m = re.search(r'base_vit\.(?:patch_embed_stages|mhca_stages)\.(\d)', s)
    if m:
        return 'ViT', int(m.group(1)) + 1
    if s.startswith('base.base_vit'):
        return 'ViT', 1
    if s.startswith('base'):
        return 'fusion+decoder+heads', lv
    if sym in ('tcct_grad_sumsq', 'tcct_clip_adamw', 'tcct_conv32_pack_weights_multi', 'tcct_image_to_nhwc4'):
        return 'optimizer / step prologue', None
    return 'loss', lv


def dispatches(kind):
    """-> list of (call id, kernel name, value) in dispatch order for the traced steps; value = duration in us or the counter"""
    if kind == 'trace':
        f = glob.glob(os.path.join(root, f'{tag}_attrib_trace', '**', '*kernel_trace.csv'), recursive=True)
        if not f:
            return None
        rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
        seq = [(r['Kernel_Name'], int(r['Grid_Size_X']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
    else:
        f = glob.glob(os.path.join(root, f'{tag}_attrib_{kind}', '**', '*counter_collection.csv'), recursive=True)
        if not f:
            return None
        rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Dispatch_Id']))
        seq = [(r['Kernel_Name'], int(r['Grid_Size']), float(r['Counter_Value'])) for r in rows]
    out, cur = [], None
    for name, grid, v in seq:
        if name.startswith('k_marker'):
            cur = grid // 64
            continue
        if cur is not None:
            out.append((cur, name, v))
    return out


tr, fe, wr = dispatches('trace'), dispatches('fetch'), dispatches('write')
per = collections.defaultdict(lambda: {'us': 0.0, 'n': 0, 'short_us': 0.0, 'short_n': 0, 'fetch': 0.0, 'write': 0.0, 'kernels': collections.Counter()})
for cid, name, us in tr or []:
    p = per[cid]
    p['us'] += us; p['n'] += 1
    p['kernels'][name.split('(')[0]] += 1
    if us < 20:
        p['short_us'] += us; p['short_n'] += 1
for seq, key in ((fe, 'fetch'), (wr, 'write')):
    for cid, name, v in seq or []:
        per[cid][key] += v
have_pmc = fe is not None and wr is not None

acc = collections.defaultdict(lambda: collections.defaultdict(float))
with open(os.path.join(root, f'{tag}_attrib_calls.csv'), 'w') as fcsv:
    fcsv.write('id,step,dir,branch,level,symbol,scope,dispatches,us,hbm_MB,kernels\n')
    for cid, c in sorted(calls.items()):
        p = per.get(cid)
        if p is None:
            continue
        br, lv = classify(c)
        mb = (2 * p['fetch'] + p['write']) * 1024 / 1e6        # counters in KiB; FETCH_SIZE doubled (gfx950 wide reads, MI355X_MICROARCH.md)
        fcsv.write('%d,%d,%s,%s,%s,%s,%s,%d,%.1f,%.2f,"%s"\n' % (cid, c['step'], c['dir'], br, 'n/a' if lv is None else lv, c['sym'], c['scope'], p['n'], p['us'], mb,
                                                              ' + '.join(f'{k} x{v}' if v > 1 else k for k, v in p['kernels'].items())))
        for key in ((br, lv), (br, 'all'), ('all', lv), ('all', 'all'), ('dir:' + c['dir'], 'all')):
            a = acc[key]
            a['us'] += p['us']; a['n'] += p['n']; a['short_us'] += p['short_us']; a['short_n'] += p['short_n']; a['mb'] += mb; a['calls'] += 1

L = [f'# {tag}: the training step by level and branch (`tools/attrib_trace.sh`; `--los={log["los"]}`, bs {BS}, bf16, single stream, {NS} traced steps; per step)\n',
     'Every C-ABI call of the traced steps is preceded by a marker launch whose grid size is its id; the dispatches of the kernel trace and of the two PMC passes '
     '(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE KiB) are cut at the markers and joined with the call log (symbol, geometry, '
     'module scope; a backward node counts with the module its forward ran in).  Model bytes: SURVEY 8(d) by level (docstring of `tools/attrib_summary.py`).  '
     'The markers themselves (one empty launch per call) are not counted; they add ~2 us of stream time per call to the traced run.\n']


def table(title, keys, label):
    L.append(f'## {title}\n')
    L.append('| ' + label + ' | ms | launches | of them < 20 us | ms in < 20 us | real GB (PMC) | model GB | real / model | TB/s on real bytes |\n|---|---|---|---|---|---|---|---|---|')
    for key, name, model in keys:
        a = acc.get(key)
        if not a:
            continue
        ms, gb = a['us'] / NS / 1e3, a['mb'] / NS / 1e3
        L.append(f'| {name} | {ms:.3f} | {a["n"] / NS:.0f} | {a["short_n"] / NS:.0f} | {a["short_us"] / NS / 1e3:.3f} | {gb:.2f} | '
                 + (f'{model:.2f} | {gb / model:.2f}' if model else 'n/a | n/a') + f' | {gb / ms if ms else 0:.2f} |')
    L.append('')


branches = ['CNN', 'ViT', 'fusion+decoder+heads', 'loss', 'optimizer / step prologue']
bm = {b: sum(v for (bb, _), v in MODEL.items() if bb == b) * GB_PER_ELEM for b in branches}
table('by branch', [((b, 'all'), b, bm.get(b) or None) for b in branches] + [(('all', 'all'), '**whole step**', sum(MODEL.values()) * GB_PER_ELEM)], 'branch')
lm = {lv: sum(v for (_, l_), v in MODEL.items() if l_ == lv) * GB_PER_ELEM for lv in range(5)}
table('by level (all branches; level = resolution 800x1104 / 2^level)', [(('all', lv), f'L{lv}', lm[lv]) for lv in range(5)] + [(('all', None), 'no geometry (finalize, compose, optimizer)', None)], 'level')
table('by branch and level', [((b, lv), f'{b} L{lv}', MODEL.get((b, lv), 0) * GB_PER_ELEM or None) for b in branches[:4] for lv in range(5)], 'branch, level')
table('by direction', [(('dir:fwd', 'all'), 'forward (+ loss forward, step prologue, optimizer)', None), (('dir:bwd', 'all'), 'backward', None)], 'direction')
if not have_pmc:
    L.append('(no PMC passes found: the byte columns are zero)\n')
# the small levels in detail: per call, levels 2-4 of the two encoders
L.append('## levels 3-4 of the CNN encoder, call by call (step 0 of the traced steps)\n')
L.append('| dir | level | C-ABI call | dispatches | us | HBM MB | kernels |\n|---|---|---|---|---|---|---|')
for cid, c in sorted(calls.items()):
    p = per.get(cid)
    if p is None or c['step'] != 0:
        continue
    br, lv = classify(c)
    if br == 'CNN' and lv in (3, 4):
        mb = (2 * p['fetch'] + p['write']) * 1024 / 1e6
        L.append(f'| {c["dir"]} | L{lv} | `{c["sym"]}` | {p["n"]} | {p["us"]:.1f} | {mb:.2f} | ' + ', '.join(f'`{k[:60]}`' for k in p['kernels']) + ' |')
out = os.path.join(root, f'{tag}_attrib_summary.md')
open(out, 'w').write('\n'.join(L) + '\n')
print('\n'.join(L[:60]))

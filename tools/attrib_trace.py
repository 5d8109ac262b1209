"""Per-level / per-branch attribution of the training step: which encoder level and which branch does every launch belong to?

rocprofv3's kernel trace and PMC tables record a dispatch's name and GRID but no arguments.  This script runs the single-stream training step with a marker
launch (`tcct_marker`: an empty kernel whose grid size IS an id) in front of every C-ABI call and writes, per id, what the call was: symbol, tensor geometry,
direction and the module scope it was issued from (forward: module hooks; backward: the scope its autograd node was created in).  `tools/attrib_summary.py`
then cuts the dispatch sequence of the trace / PMC passes of THIS script at the markers and sums time, launches and HBM bytes per level and branch.

    TCCT_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d OUT -o at -- python3 tools/attrib_trace.py --log OUT/attrib_calls.json [--los di]

Run under the profiler or alone (alone it only writes the call log).  3 warm-up + 3 traced steps; only the traced steps carry markers."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['TCCT_STREAMS'] = '0'         # one stream: durations add up and the dispatch order is the call order

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--log', required=True)
    p.add_argument('--los', default='di')
    p.add_argument('--bs', type=int, default=8)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--infer', action='store_true', help='trace KiteSeg.predict (eval mode, no_grad) instead of the training step')
    a = p.parse_args()
    ba = argparse.Namespace(los=a.los, bs=a.bs, height=800, width=1100, dtype='bf16', att='pool')
    from tcct_amd import _lib, ops
    from tcct_amd._lib import lib
    k, ds, _ = bench.build_trainer(ba, 1)
    batch = ds.make_batch(a.bs, seed=2023)
    img, lab, _, _ = ds.parse(batch)
    img, lab = img.contiguous(), lab.contiguous()

    # ---- scope tracking: module path while a module's forward runs; an autograd node's backward runs in the scope its forward ran in
    scope = ['']
    names = {m: n for n, m in k.model.named_modules()}

    def pre(m, *_):
        scope.append(names.get(m, '?'))

    def post(m, *_):
        scope.pop()
    for m in k.model.modules():
        m.register_forward_pre_hook(pre)
        m.register_forward_hook(post, always_call=True)

    def wrap_method(cls, name, gen=False):
        orig = getattr(cls, name)
        if gen:
            def w(self, *aa, **kk):
                g = orig(self, *aa, **kk)
                while True:
                    scope.append(names.get(self, '?'))
                    try:
                        v = next(g)
                    except StopIteration:
                        return
                    finally:
                        scope.pop()
                    yield v
        else:
            def w(self, *aa, **kk):
                scope.append(names.get(self, '?'))
                try:
                    return orig(self, *aa, **kk)
                finally:
                    scope.pop()
        setattr(cls, name, w)
    import importlib
    T = importlib.import_module('tcct_amd.nets.tcct')      # (`tcct_amd.nets.tcct` the attribute is the factory alias of stc_tt)
    wrap_method(T.CrossResNet, 'iter_levels', gen=True)
    wrap_method(T.MPViT, 'iter_stages', gen=True)
    wrap_method(T.Conv2d_BN, 'forward_deferred')
    wrap_method(T.MPUpBlock, 'forward_through')
    wrap_method(T.ResBlock, 'tail')
    direction = ['fwd']
    import tcct_amd.nets.reg as R
    import tcct_amd.kite.loop_seg as LS
    seen = set()
    for mod in (ops, R, LS, T):
        for nm, F in list(vars(mod).items()):
            if isinstance(F, type) and issubclass(F, torch.autograd.Function) and F is not torch.autograd.Function and F not in seen:
                seen.add(F)

                def mk(F):
                    of, ob = F.forward, F.backward

                    def fwd(ctx, *aa, **kk):
                        ctx._attr_scope = scope[-1]
                        return of(ctx, *aa, **kk)

                    def bwd(ctx, *aa, **kk):
                        scope.append(getattr(ctx, '_attr_scope', ''))
                        direction.append('bwd')
                        try:
                            return ob(ctx, *aa, **kk)
                        finally:
                            scope.pop()
                            direction.pop()
                    F.forward, F.backward = staticmethod(fwd), staticmethod(bwd)
                mk(F)

    calls, state = [], {'on': False, 'id': 0, 'step': -1, 'busy': False}
    marker = lib.load().tcct_marker

    def trace(sym, sig, args):
        if not state['on'] or state['busy']:
            return
        state['id'] += 1
        marker(state['id'], args[-1])
        ints = {nm: int(v) for (ct, nm), v in zip(sig[:-1], args) if isinstance(v, int) and not isinstance(v, bool) and abs(v) < (1 << 40)}
        big = max((t.numel() * t.element_size() for t in args if isinstance(t, torch.Tensor)), default=0)
        tb = sum(t.numel() * t.element_size() for t in args if isinstance(t, torch.Tensor))
        calls.append({'id': state['id'], 'step': state['step'], 'sym': sym, 'scope': scope[-1], 'dir': direction[-1], 'ints': ints, 'max_tensor_bytes': big, 'tensor_bytes': tb})
    _lib._TRACE[0] = trace

    if a.infer:
        k.model.eval()
        run = lambda: k.predict(img)                  # noqa: E731
    else:
        k.model.train()
        run = lambda: k.train_step(img, lab)          # noqa: E731
    for _ in range(a.warmup):
        run()
    torch.cuda.synchronize()
    state['on'] = True
    for s in range(a.steps):
        state['step'] = s
        run()
    torch.cuda.synchronize()
    state['on'] = False
    json.dump({'los': a.los, 'bs': a.bs, 'steps': a.steps, 'infer': a.infer, 'calls': calls}, open(a.log, 'w'))
    print(f'{len(calls)} C-ABI calls logged over {a.steps} steps -> {a.log}', file=sys.stderr)


if __name__ == '__main__':
    main()

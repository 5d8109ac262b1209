"""CLI — mirror of reference kite/main.py:18-72: the same 23 flags, `--net` resolved to a factory in `tcct_amd.nets`,
`RegNet(net, con=args.type_udh, out_channels=...)`, `KiteSeg(...).fit(epochs)`.

Additions: `--los=di+reg+fpl` shorthand (BASELINE.json) == `--los=di --reg=true --udh=true`; `--db=synth` synthetic
GOALS-shaped generator; `--pl=true` = one process per GPU under torchrun; `--dtype=bf16|fp32` compute precision.

    python -m tcct_amd.kite.main --bs=8 --net=stc_tt --los=di --db=synth --epochs=1
"""
import argparse

import torch


def str2bool(v):
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Unsupported value encountered.')


def build_parser():
    p = argparse.ArgumentParser(description='KiteOCT Argument')
    p.add_argument('--db', type=str, default='synth', help='dataset')
    p.add_argument('--lr', type=float, default=1e-2, help='learning rate')
    p.add_argument('--wd', type=float, default=5e-4, help='weight decay (ignored by the reference too: fixed 2e-4)')
    p.add_argument('--inc', type=str, default='', help='instruction')
    p.add_argument('--gpu', type=str, default='0', help='cuda number')
    p.add_argument('--los', type=str, default='dice', help='loss function')
    p.add_argument('--net', type=str, default='stc_tt', help='network')
    p.add_argument('--pth', type=str2bool, default=True)
    p.add_argument('--bs', type=int, default=2, help='batch size')
    p.add_argument('--epochs', type=int, default=100)
    p.add_argument('--root', type=str, default='', help='folder to train or test again')
    p.add_argument('--resume', type=str2bool, default=False)
    p.add_argument('--reg', type=str2bool, default=False, help='reg loss!')
    p.add_argument('--coff_reg', type=float, default=.1)
    p.add_argument('--epl', type=str2bool, default=False)
    p.add_argument('--coff_epl', type=float, default=.1)
    p.add_argument('--udh', type=str2bool, default=False, help='udh loss!')
    p.add_argument('--coff_udh', type=float, default=1)
    p.add_argument('--type_udh', type=str, default='cos', choices=['cos', 'mse'])
    p.add_argument('--ds', type=str2bool, default=False, help='Deep Supervision (always on, as in the reference)')
    p.add_argument('--coff_ds', type=float, default=1)
    p.add_argument('--pl', type=str2bool, default=False, help='Parallel: one process per GPU (torchrun)')
    p.add_argument('--bug', type=str2bool, default=False, help='Debug Mode!')
    p.add_argument('--dtype', type=str, default='bf16', choices=['bf16', 'fp32'], help='compute dtype of activations')
    p.add_argument('--att', type=str, default='pool', choices=['pool', 'factor'],
                   help="token mixer of the ViT blocks: 'pool' = MetaPool (reference nets/tcct.py:449); 'factor' = the factorised attention the "
                        "reference keeps commented out (nets/tcct.py:443-448; --net=stc_tt / tcct only)")
    p.add_argument('--graph', type=str2bool, default=False,
                   help='EXPERIMENTAL: replay the training step from a hipGraph (launch-bound crop sizes such as the 256x256 of the reference '
                        'recipe; single process, fixed batch shape).  A capture late in a long process has crashed inside hipGraphLaunch '
                        '(DESIGN 5b, cause open): use it from a fresh process only')
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    if '+' in args.los:                      # 'di+reg+fpl' shorthand
        parts = args.los.split('+')
        args.los = parts[0]
        for q in parts[1:]:
            if q == 'reg':
                args.reg = True
            elif q in ('fpl', 'udh'):
                args.udh = True
            else:
                raise SystemExit(f'unknown loss component {q!r} in --los')
    return args


def main(argv=None):
    args = parse_args(argv)
    from .. import nets
    from ..data import EyeSetGenerator
    from .loop_seg import KiteSeg
    dataset = EyeSetGenerator(dbname=args.db)
    factory = getattr(nets, args.net, None)
    if factory is None:
        raise SystemExit(f'--net={args.net}: unknown network (available: stc_tt / tcct, stc_tb, gtc_tt, gtc_tb, cnnu, pnnu, vitu)')
    kw = dict(compute_dtype=torch.bfloat16 if args.dtype == 'bf16' else torch.float32)
    if args.att != 'pool':
        if args.net not in ('stc_tt', 'tcct'):
            raise SystemExit(f'--att={args.att} is only offered for --net=stc_tt')
        kw['att'] = args.att
    net = factory(dataset.out_channels, **kw)
    net = nets.RegNet(net, con=args.type_udh, out_channels=dataset.out_channels)
    keras = KiteSeg(model=net, dataset=dataset, root=args.root, args=args)
    if args.resume:
        path = args.root + '/val_top.pt'
        keras.model.load_state_dict(torch.load(path, map_location=keras.device, weights_only=True), strict=False)
        print('loaded model:', path)
    keras.fit(epochs=1 if args.bug else args.epochs)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""bench.py — TCCT `stc_tt` training hot path on MI355X: OCT B-scans/s, forward + losses + backward + clip + AdamW.

    python bench.py --gpus N --steps K --warmup W

N>1: one rank per GPU.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
WORLD_SIZE/RANK/LOCAL_RANK are then in the environment) or, when `--gpus N` arrives WITHOUT that environment, this process becomes a
launcher: it starts exactly that torch.distributed.run command as a CHILD process before anything here has touched the GPU (never an
exec), relays the child's one JSON line and exit code.  A line whose `n_gpus` differs from `--gpus` is an error, never printed.

A "step" = one pass of the hot path over one synthetic minibatch of `--bs` B-scans (1x800x1100 each, already resident in
HBM; loader-side prep = 1->3 channel replicate inside the NHWC conversion kernel + W zero-pad 1100->1104).  Workload at
N=1 = BASELINE.json configs[1]: `stc_tt --los=di bs=8 1x800x1100 bf16`; other configs via --los=di+reg / di+reg+fpl.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the oracle = CPU port of the reference path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL between the ranks of one node (before HIP initialises)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
F32_MFMA_PEAK_TFLOPS = 157.3    # dense fp32 matrix peak (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU x 256 CUs x 2.4 GHz)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=10)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--bs', type=int, default=8, help='per-GPU minibatch (weak scaling: global = bs * gpus)')
    p.add_argument('--height', type=int, default=800)
    p.add_argument('--width', type=int, default=1100)
    p.add_argument('--los', type=str, default='di', help="di | di+reg | di+reg+fpl")
    p.add_argument('--dtype', type=str, default='bf16', choices=['bf16', 'fp32'])
    p.add_argument('--att', type=str, default='pool', choices=['pool', 'factor'],
                   help="token mixer: 'pool' = the reference's stc_tt (headline); 'factor' = its commented-out factorised attention (not the headline)")
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--no-other-workloads', action='store_true', help="skip the --los=di+reg / di+reg+fpl lines (`config.other_workloads`) that follow the headline loop at N=1")
    p.add_argument('--roofline-only', action='store_true', help='only run the dominant-kernel timing loop (for rocprofv3)')
    p.add_argument('--wgrad-mode', type=int, default=0, help='A/B arms of the weight-gradient kernels (tcct_conv32_wgrad_mode: 0 default, 4 = shifted lines for 1xK / Kx1)')
    p.add_argument('--conv-mode', type=int, default=0, help='A/B arms of the 3x3 forward kernels (tcct_conv32_fwd_mode: 0 default, 1 = tiled)')
    p.add_argument('--set', type=str, default='', help='A/B: comma-separated NAME=value assignments of tcct_amd.ops module constants (switches 0|1, e.g. FUSED_CONV_BWD=1; integer constants, e.g. STAGE_FORK_MAX_PIXELS=0)')
    return p.parse_args()


def build_trainer(a, world):
    from tcct_amd.kite.main import parse_args
    from tcct_amd.data import SynthOCT
    from tcct_amd import nets
    from tcct_amd.kite.loop_seg import KiteSeg
    args = parse_args([f'--los={a.los}', f'--bs={a.bs}', '--db=synth', f'--pl={"true" if (world > 1 or os.environ.get("TCCT_FORCE_DIST") == "1") else "false"}',
                       f'--dtype={a.dtype}', '--root=/tmp/tcct_bench_root'])
    ds = SynthOCT(height=a.height, width=a.width, device='cuda')
    net = nets.stc_tt(ds.out_channels, compute_dtype=torch.bfloat16 if a.dtype == 'bf16' else torch.float32, att=a.att)
    net = nets.RegNet(net, con=args.type_udh, out_channels=ds.out_channels)
    k = KiteSeg(model=net, dataset=ds, root=args.root, args=args)
    return k, ds, args


class _StubTrainer:
    """TCCT_BENCH_STUB=1 -- a REHEARSAL of the multi-rank plumbing on a machine without GPUs (tests/test_host_cpu.py runs `bench.py --gpus 8`
    through the launcher here): every rank joins a gloo group and a "step" is the data-parallel part of the real one, i.e. ONE all-reduce of a flat
    fp32 buffer of the real gradient size over tcct_amd.dist (802 298 elements) followed by the 1/world mean, and nothing else.  The printed line is
    labelled as a stub in `metric` and `data`; it measures nothing about the product."""

    def __init__(self, world, rank):
        from tcct_amd import dist as tdist
        self.tdist, self.world = tdist, world
        self.flat = torch.full((802298,), float(rank + 1))
        self.want = sum(range(1, world + 1)) / world

        class _O:
            allreduce_mode = 'single blocking all-reduce after backward (stub: gloo, CPU tensor)'
            _flat = None
        self.optimG = _O()

    def train_step(self, img, lab):
        self.flat.fill_(float(self.tdist.world_rank()[1] + 1))
        self.tdist.allreduce_sum_(self.flat)
        mean = self.flat / self.world
        if abs(float(mean[0]) - self.want) > 1e-6 or abs(float(mean[-1]) - self.want) > 1e-6:
            raise SystemExit(f'stub all-reduce: mean {float(mean[0])} != {self.want}')
        return mean[:1].sum()


def pmc_traffic(a, kernel_match):
    """`roofline.traffic`: HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of `bench.py --roofline-only`
    (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, FETCH_SIZE doubled per MI355X_MICROARCH.md), as recorded by
    tools/pmc_json.py in profiles/TAG_pmc.json TOGETHER WITH a hash of the kernel sources.  Only a file collected on exactly the
    sources this run was built from (and at this shape) is used; otherwise traffic is null and the note says why."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import pmc_json
        table, src = pmc_json.newest_matching(a.dtype, a.bs, a.height, a.width, ROOT)
    except Exception as e:        # a missing profiles/ directory must not take the bench down
        return None, f'profiles unreadable: {e}'
    if table is None:
        return None, src
    for name, b in table.items():
        if kernel_match in name:
            return int(b), f'profiles/{src}: {name}'
    return None, f'profiles/{src} has no kernel matching {kernel_match!r}'


def dominant_kernel_roofline(a, iters=20):
    """HIP-event timing of the dominant kernel of the step at the bench shapes, on the stream it is launched on (torch's
    current stream == the stream every tcct_* call receives).  Dominant kernel by total time (profiles/r01_*): the MFMA
    implicit-GEMM convolution (k_conv32_fwd33_stream / k_conv32_mfma: forward and input-gradient of the 3x3 / 1xk 32->32 convolutions);
    timed on its most frequent instance, the 3x3 at level 0 ([bs,800,1104,32]).  Algorithmic bytes per launch = read x once +
    write y once (SURVEY §8(d) layer-granular model) = 2 * bs*H*W*32 * sizeof(dtype); the 18 KB of packed weights are noise.
    The weight-gradient kernel (second by total time) is reported next to it as `second`."""
    from tcct_amd._lib import lib
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    Wp = (a.width + 15) // 16 * 16
    x = torch.randn((a.bs, a.height, Wp, 32), device='cuda', dtype=torch.float32).to(dt)
    dy = torch.randn((a.bs, a.height, Wp, 32), device='cuda', dtype=torch.float32).to(dt)
    y = torch.empty_like(x)
    w = torch.randn((32, 32, 3, 3), device='cuda') * 0.06
    b = torch.zeros(32, device='cuda')
    dw = torch.empty((32, 32, 3, 3), device='cuda')
    db = torch.empty(32, device='cuda')
    if a.dtype == 'bf16':
        wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
        lib.conv32_pack_weights(w, wp, 3, 3, 0)
        name, name2 = 'k_conv32_fwd33_stream<0> (3x3 32->32 fwd/dgrad @L0, row streams)', 'k_conv32_wgrad33_stream (3x3 32->32 weight gradient @L0, row streams)'
        match = 'k_conv32_fwd33_stream<0>'                      # the symbol as rocprofv3 prints it
        fn = lambda: lib.conv32_fwd(x, wp, b, y, a.bs, a.height, Wp, 3, 3, 1, 1)                              # noqa: E731
        fn2 = lambda: lib.conv32_wgrad(x, dy, dw, db, a.bs, a.height, Wp, 3, 3, 1, 1)                         # noqa: E731
    else:       # parity mode: the fp32 MFMA convolution (v_mfma_f32_32x32x2_f32), bound by the fp32 matrix rate, not by HBM
        wpf = torch.empty(9 * 1024, device='cuda', dtype=torch.float32)
        lib.conv32f_pack_weights(w, wpf, 3, 3, 0)
        name, name2 = 'k_conv32f_mfma<false> (3x3 32->32 fwd/dgrad @L0, fp32 MFMA)', 'k_conv32f_wgrad<false,9> (3x3 32->32 @L0, fp32 MFMA)'
        match = 'k_conv32f_mfma<false>'
        fn = lambda: lib.conv32f_fwd(x, wpf, b, None, y, a.bs, a.height, Wp, 3, 3, 1, 1)                      # noqa: E731
        fn2 = lambda: lib.conv32f_wgrad(x, dy, dw, db, a.bs, a.height, Wp, 3, 3, 1, 1)                       # noqa: E731

    # Spin-up: the first ~10 ms of GPU activity after an idle period are a clock / power-management transient (round 2's launch-by-launch timing: the same
    # launch reads 0.202 ms, climbs to 0.24 around launches 10-40 and settles at 0.197 from launch ~60 on, and stays there across short syncs).
    # A training step is 25 ms of back-to-back kernels, so the settled state is the representative one; a neutral kernel (a device copy) does the
    # spin-up so that EVERY launch of the timed kernels -- also the ones rocprofv3 averages over in `--roofline-only` -- is in that state.
    for _ in range(120):
        y.copy_(x)

    def timed(f):
        for _ in range(2):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    ms, ms2 = timed(fn), timed(fn2)
    # the two other kernels at the top of the per-symbol ranking (profiles/r01_q_summary.md), same live timing: the BatchNorm
    # backward reduction (reads x and dy of a level-0 tensor) and the pointwise GEMM at its heaviest shape (64->64 at level 1)
    others = []
    if a.dtype == 'bf16':
        M0, C0 = a.bs * a.height * Wp, 32
        mr = torch.zeros(2 * C0, device='cuda'); mr[C0:] = 1
        ab = torch.ones(2 * C0, device='cuda')
        sums = torch.zeros(2 * C0, device='cuda', dtype=torch.float64)
        t3 = timed(lambda: lib.bn_bwd_reduce(x, dy, M0, C0, mr, ab, 1, 0, sums, 1))
        b3 = 2.0 * x.numel() * 2
        others.append({'kernel': 'k_bn_bwd_reduce<bf16,4> (32 ch @L0)', 'ms_per_launch': round(t3, 4), 'algorithmic_bytes': int(b3),
                       'achieved': round(b3 / (t3 * 1e-3) / 1e9, 1), 'frac': round(b3 / (t3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        M1 = a.bs * (a.height // 2) * (Wp // 2)
        x1 = torch.randn((M1, 64), device='cuda').to(dt); y1 = torch.empty_like(x1)
        w1 = torch.randn((64, 64), device='cuda') * 0.1; b1 = torch.zeros(64, device='cuda')
        t4 = timed(lambda: lib.pw_fwd(x1, w1, b1, y1, M1, 64, 64, 0, 1))
        b4 = 2.0 * x1.numel() * 2
        others.append({'kernel': 'k_pw_fwd2<2,4> (64->64 @L1; tile-staged pointwise forward)', 'ms_per_launch': round(t4, 4), 'algorithmic_bytes': int(b4),
                       'achieved': round(b4 / (t4 * 1e-3) / 1e9, 1), 'frac': round(b4 / (t4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        # its whole backward in one launch (reads x and dy, writes dx; dW / db accumulate in fp32)
        dy1 = torch.randn_like(x1); dx1 = torch.empty_like(x1)
        dw1 = torch.zeros((64, 64), device='cuda'); db1 = torch.zeros(64, device='cuda')
        t5 = timed(lambda: lib.pw_bwd(x1, dy1, w1, None, dx1, dw1, db1, M1, 64, 64))
        b5 = 3.0 * x1.numel() * 2
        others.append({'kernel': 'k_pw_bwd<2,4> (64->64 @L1; dx + dW + db in one pass)', 'ms_per_launch': round(t5, 4), 'algorithmic_bytes': int(b5),
                       'achieved': round(b5 / (t5 * 1e-3) / 1e9, 1), 'frac': round(b5 / (t5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        # round 5: conv3x3 -> conv3x3 of CrossCNNBlock.block12 as one launch (reads x, writes the intermediate and y: 3 tensors instead of 4) and the 13-tap
        # weight gradients as one-wave-per-SIMD row streams (read x and dy)
        midc = torch.empty_like(x)
        t6 = timed(lambda: lib.conv32_chain33(x, wp, b, midc, wp, b, y, None, a.bs, a.height, Wp, None))
        b6 = 3.0 * x.numel() * 2
        others.append({'kernel': 'k_conv32_chain33<0> (3x3 -> 3x3 @L0 in one launch; algorithmic bytes = x + intermediate + y)', 'ms_per_launch': round(t6, 4),
                       'algorithmic_bytes': int(b6), 'achieved': round(b6 / (t6 * 1e-3) / 1e9, 1), 'frac': round(b6 / (t6 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        del midc
        for kh, kw, nm in ((1, 13, 'k_conv32_wgradk_stream<13,false> (1x13 weight gradient @L0)'), (13, 1, 'k_conv32_wgradk_stream<13,true> (13x1 weight gradient @L0)')):
            dwk = torch.empty(32, 32, kh, kw, device='cuda')
            t7 = timed(lambda: lib.conv32_wgrad(x, dy, dwk, db, a.bs, a.height, Wp, kh, kw, kh // 2, kw // 2))
            b7 = 2.0 * x.numel() * 2
            others.append({'kernel': nm, 'ms_per_launch': round(t7, 4), 'algorithmic_bytes': int(b7), 'achieved': round(b7 / (t7 * 1e-3) / 1e9, 1),
                           'frac': round(b7 / (t7 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    if a.dtype == 'bf16' and ('fpl' in a.los or 'udh' in a.los or 'reg' in a.los):
        # the loss-side kernels of BASELINE configs[2..3] (algorithmic bytes = inputs read once + outputs written once)
        def entry(kernel, t, nbytes):
            others.append({'kernel': kernel, 'ms_per_launch': round(t, 4), 'algorithmic_bytes': int(nbytes), 'achieved': round(nbytes / (t * 1e-3) / 1e9, 1),
                           'frac': round(nbytes / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        B_, H_ = a.bs, a.height
        if 'reg' in a.los:
            xg = torch.randn((B_, H_, Wp, 4), device='cuda'); eg = torch.rand((B_, H_, Wp, 4), device='cuda')
            og = torch.empty((B_, H_, Wp), device='cuda'); sg = torch.empty((B_ * Wp * 4, 3), device='cuda')
            tg = timed(lambda: lib.gumbel_colsoftmax_fwd(xg, eg, og, sg, B_, H_, Wp, 4))
            entry('k_gumbel_fwd<4> (sampling softmax over H + channel sum, fp32 [8,800,1104,4]; reads x and eps, three passes with recomputation)', tg,
                  4.0 * (2 * xg.numel() + og.numel()))
            del xg, eg, og, sg
        if 'fpl' in a.los or 'udh' in a.los:
            g1 = torch.randn((B_, H_ // 2, Wp // 2, 32), device='cuda').to(dt); g2 = torch.randn((B_, H_ // 4, Wp // 4, 32), device='cuda').to(dt)
            i1 = torch.empty(g1.numel() // 32, device='cuda'); i2 = torch.empty(g2.numel() // 32, device='cuda')
            tn = timed(lambda: lib.normadd_fwd(x, g1, g2, i1, i2, y, B_, H_, Wp, 32, H_ // 2, Wp // 2, H_ // 4, Wp // 4, 1e-12, 1))
            entry('k_normadd_fwd<bf16> (+ 2 x k_invnorm): feats = mean of the three L2-normalised, resized decoder maps', tn,
                  2.0 * (2 * x.numel() + g1.numel() + g2.numel()))
            Mf = B_ * H_ * Wp
            labf = torch.randint(0, 5, (Mf,), device='cuda', dtype=torch.uint8)
            labf = torch.sort(labf.view(B_, H_, Wp), dim=1).values.contiguous().view(-1)          # layered labels, as B-scans have
            probf = torch.rand(Mf, device='cuda') * 0.1 + 0.15                                     # ~1/C everywhere: the freshly initialised network of the bench
            ws = torch.empty(int(lib.fpl_select_workspace_bytes()), device='cuda', dtype=torch.uint8)
            cnt = torch.empty(16, device='cuda', dtype=torch.int32); bm = torch.empty(Mf, device='cuda', dtype=torch.uint8)
            ps = torch.empty((5, 32, 32), device='cuda')
            tf_ = timed(lambda: lib.fpl_select(x, labf, probf, Mf, 5, ws, cnt, bm, ps, 1))
            entry('tcct_fpl_select: radix multi-select of the bin boundaries (k_fs_hist x 7: one launch per radix level, the block with the last ticket resolves; k_fs_assign) + bin sums on MFMA (k_fs_binsum_mfma), 9 launches; '
                  'replaces rocPRIM radix_sort_pairs + the sorted gather', tf_, 2.0 * x.numel() + Mf * (1 + 4 + 1))
            # the same on a TRAINED network's distribution: saturated, tie-heavy probabilities (most pixels at exactly 1.0, the rest spread): more radix
            # levels have to be resolved before the boundaries separate
            probt = torch.sigmoid(8.0 * torch.randn(Mf, device='cuda'))
            tf2_ = timed(lambda: lib.fpl_select(x, labf, probt, Mf, 5, ws, cnt, bm, ps, 1))
            entry('tcct_fpl_select on a trained-network distribution (prob = sigmoid(8 N(0,1)): saturated, tie-heavy)', tf2_, 2.0 * x.numel() + Mf * (1 + 4 + 1))
            del g1, g2, i1, i2, labf, probf, probt, ws, cnt, bm, ps
    bytes_alg = 2.0 * x.numel() * x.element_size()
    ach, ach2 = bytes_alg / (ms * 1e-3) / 1e9, bytes_alg / (ms2 * 1e-3) / 1e9
    flops = 2.0 * 9 * 32 * 32 * a.bs * a.height * Wp
    traffic, traffic_src = pmc_traffic(a, match)
    if a.dtype != 'bf16':
        tf, tf2 = flops / (ms * 1e-3) / 1e12, flops / (ms2 * 1e-3) / 1e12
        return {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                'traffic': None, 'traffic_source': 'not collected for the parity mode', 'kernel': name, 'ms_per_launch': round(ms, 4), 'launches_timed': iters,
                'algorithmic_flops': int(flops), 'hbm_GBs': round(ach, 1),
                'second': {'kernel': name2, 'achieved': round(tf2, 2), 'frac': round(tf2 / F32_MFMA_PEAK_TFLOPS, 4), 'ms_per_launch': round(ms2, 4)}, 'others': others}
    return {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
            'traffic': traffic, 'traffic_source': traffic_src, 'kernel': name, 'ms_per_launch': round(ms, 4),
            'launches_timed': iters, 'algorithmic_bytes': int(bytes_alg), 'tflops': round(flops / (ms * 1e-3) / 1e12, 2),
            'second': {'kernel': name2, 'achieved': round(ach2, 1), 'frac': round(ach2 / HBM_PEAK_GBS, 4), 'ms_per_launch': round(ms2, 4),
                       'algorithmic_bytes': int(bytes_alg)}, 'others': others}


def copy_ceiling(a, iters=20):
    """achievable HBM bandwidth on THIS box for a level-0 tensor pair (read 452 MB + write 452 MB at the bench shape): the library's streaming copy
    (`tcct_stream_copy`: 16 B per lane, one 8 KB chunk per block -- the shape tools/probe/stream_probe.hip found fastest; the hardware guide quotes
    6.29 TB/s for a float4 copy) is the yardstick; torch's `copy_` (the round-3 yardstick, ~10-15 % slower) is reported beside it"""
    from tcct_amd._lib import lib
    Wp = (a.width + 15) // 16 * 16
    n = a.bs * a.height * Wp * 32
    x = torch.empty(n, device='cuda', dtype=torch.bfloat16).normal_()
    y = torch.empty_like(x)

    def timed(fn):
        for _ in range(60):         # spin-up, see dominant_kernel_roofline
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    ms = timed(lambda: lib.stream_copy(x, y, 2 * n))
    ms_t = timed(lambda: y.copy_(x))
    return {'kernel': 'tcct_stream_copy (16 B/lane, one 8 KB chunk per block) of a level-0 tensor', 'bytes': int(2 * n * 2), 'ms': round(ms, 4),
            'GBs': round(2 * n * 2 / (ms * 1e-3) / 1e9, 1), 'torch_copy_ms': round(ms_t, 4), 'torch_copy_GBs': round(2 * n * 2 / (ms_t * 1e-3) / 1e9, 1),
            'guide_float4_copy_GBs': 6290.0}


def optimizer_ms(k, iters=20):
    """clip_grad_norm_(12) + AdamW alone (k_sumsq + k_clip_adamw on the flat buffers; lr 0 and no weight decay so the weights stay put),
    HIP-event timed: SURVEY 8(d) asks for it next to the fused total"""
    from tcct_amd._lib import lib
    f = k.optimG._flat
    if f is None:
        return None
    g0 = k.optimG.param_groups[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m, v = f['m'].clone(), f['v'].clone()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        lib.grad_sumsq(f['g'], f['n'], f['sumsq'])
        lib.clip_adamw(f['p'], f['g'], m, v, f['n'], f['sumsq'], 12.0, 1.0, 0.0, float(g0['betas'][0]), float(g0['betas'][1]), float(g0['eps']), 0.0, 1, f['norm'])
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters, 4)


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.lower().startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(a):
    """oracle (CPU port of the reference path, pinned to the reference by tests/golden) on the host cores: full steps
    (fwd + Dice deep supervision [+reg+fpl] + bwd + clip + AdamW) on ONE full-size B-scan (3x800x1104 fp32).  The thread
    count is calibrated first (torch's CPU backend gets SLOWER beyond 16-32 threads on the 2x64-core EPYC GPU boxes).
    BASELINE.md 3 also asks for BASELINE.json configs[0] exactly (bs 2, 64x64, --los=di: 5 warm-up + 20 timed steps) and for the full-loss
    GOALS-shape line: both ride along as `cfg1` / `full_loss` (the headline `value` stays the workload of the GPU line)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import tcct_oracle as O
    keys = [(k, tuple(s)) for k, s in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_keys.json')))]
    Wp = (a.width + 15) // 16 * 16

    def one_step(H, W, state, udh, reg, bs=1):
        if state is None:
            sd = O.formula_state_dict(keys)
            names = [k for k, v in sd.items() if v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))
                     and not k.startswith('fcp.')]                 # unused parameters simply end up with grad None
            for n in names:
                sd[n].requires_grad_(True)
            img, lab = O.synth_batch(bs, H, W, seed=2023)
            oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
            noise = (torch.rand(bs, 4, H, W), torch.rand(bs, 4, H, W), torch.rand(1, 1, H, 1), torch.rand(1, 1, H, 1)) if reg else None
            state = dict(sd=sd, names=names, img=img, oh=oh, noise=noise, M=None, V=None, step=0)
        sd, names = state['sd'], state['names']
        for n in names:
            sd[n].grad = None
        t0 = time.time()
        tot, _, _, _ = O.total_loss(sd, state['img'], state['oh'], udh=udh, reg=reg, noise=state['noise'])
        tot.backward()
        P = [sd[n] for n in names if sd[n].grad is not None]
        if state['M'] is None:
            state['M'] = [torch.zeros_like(p) for p in P]
            state['V'] = [torch.zeros_like(p) for p in P]
        state['step'] += 1
        O.clip_adamw_step(P, [p.grad for p in P], state['M'], state['V'], state['step'], 1e-6)
        return time.time() - t0, state

    def timed(H, W, udh, reg, bs, warm, n):
        st = None
        for _ in range(warm):
            _, st = one_step(H, W, st, udh, reg, bs)
        ts = []
        for _ in range(n):
            t, st = one_step(H, W, st, udh, reg, bs)
            ts.append(t)
        return sum(ts) / len(ts)

    udh, reg = 'fpl' in a.los or 'udh' in a.los, 'reg' in a.los
    ncpu = os.cpu_count() or 1
    model = _cpu_model()
    best, best_t = None, 1e30
    for th in (8, 16, 32, 64):
        if th > ncpu:
            break
        torch.set_num_threads(th)
        _, st = one_step(208, 288, None, udh, reg)
        t, _ = one_step(208, 288, st, udh, reg)
        if t < best_t:
            best, best_t = th, t
    torch.set_num_threads(best)
    dt = timed(a.height, Wp, udh, reg, 1, 1, 3)          # 1 warm-up (oneDNN primitive creation) + 3 timed
    out = {'value': round(1.0 / dt, 4), 'unit': 'B-scans/s', 'cores': best, 'kind': 'port', 'cpu_model': model,
           'sample': f'3 timed steps (after 1 warm-up), bs=1, 3x{a.height}x{Wp} fp32, --los={a.los}; oracle/tcct_oracle.py on torch CPU '
                     f'{torch.__version__}; {best} threads = fastest of 8/16/32/64 on this {ncpu}-CPU host ({model}); {dt:.2f}s/step'}
    # BASELINE.json configs[0] exactly: stc_tt --los=di, bs 2, 64 x 64 crops, fp32; 5 warm-up + 20 timed steps.  Small maps want few threads.
    th1 = min(best, 8)
    torch.set_num_threads(th1)
    d1 = timed(64, 64, False, False, 2, 5, 20)
    out['cfg1'] = {'value': round(2.0 / d1, 3), 'unit': 'B-scan crops/s', 'cores': th1, 'kind': 'port',
                   'sample': f'BASELINE.json configs[0]: stc_tt --los=di bs=2 3x64x64 fp32, 5 warm-up + 20 timed steps, {d1 * 1e3:.1f} ms/step, {th1} threads'}
    torch.set_num_threads(best)
    if not (udh and reg):
        d2 = timed(a.height, Wp, True, True, 1, 1, 2)
        out['full_loss'] = {'value': round(1.0 / d2, 4), 'unit': 'B-scans/s', 'cores': best, 'kind': 'port',
                            'sample': f'--los=di+reg+fpl (BASELINE.json configs[3] loss set), bs=1, 3x{a.height}x{Wp} fp32, 1 warm-up + 2 timed steps, {d2:.2f}s/step'}
    return out


_RESULT_OUT = None


def _reserve_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner on fd 1 when its communicator is created (seen under
    torchrun on the GPU boxes: "RCCL version : ..." ahead of the result), and any library may do the like: keep the real stdout for
    the result line and route everything else written to fd 1 during the run to stderr."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    return _RESULT_OUT


def launch_ranks(a):
    """`--gpus N` (N > 1) without a torchrun environment: be the launcher.  The parent makes NO GPU call (no torch.cuda.* that
    initialises HIP, no tcct_amd import) — it starts `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a child
    process, lets its stderr through, and relays exactly one JSON result line from the child's stdout.  Exit code = the child's."""
    import socket
    import subprocess
    port = os.environ.get('TCCT_BENCH_PORT')
    if port is None:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', TCCT_BENCH_CHILD='1')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or a.gpus) // a.gpus)))
    print('bench.py: launching', ' '.join(cmd), file=sys.stderr, flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in r.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        elif ln:
            print(ln, file=sys.stderr)
    if r.returncode != 0:
        raise SystemExit(f'bench.py: the {a.gpus}-rank child exited with code {r.returncode}' if r.returncode > 0 else
                         f'bench.py: the {a.gpus}-rank child was killed by signal {-r.returncode}')
    if line is None:
        raise SystemExit('bench.py: the child ranks printed no result line')
    got = json.loads(line)
    if got.get('n_gpus') != a.gpus or got.get('config', {}).get('ranks') != a.gpus:
        raise SystemExit(f"bench.py: asked for --gpus={a.gpus} but the result line says n_gpus={got.get('n_gpus')} "
                         f"ranks={got.get('config', {}).get('ranks')}")
    print(line, flush=True)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(a)              # before ANY GPU call in this process
    out_stream = _reserve_stdout()
    from tcct_amd import dist as tdist
    world, rank, local = tdist.env_world()
    if world != a.gpus:
        raise SystemExit(f'--gpus={a.gpus} but WORLD_SIZE={world}: one rank per GPU, launch with --nproc-per-node={a.gpus}')
    stub = os.environ.get('TCCT_BENCH_STUB') == '1'         # launcher / process-group rehearsal without GPUs (_StubTrainer)
    if not stub and not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X')
    sync = (lambda: None) if stub else torch.cuda.synchronize
    dev = torch.device('cpu') if stub else None
    if not stub:
        if os.environ.get('TCCT_DIST_BACKEND') == 'gloo':      # test mode: several ranks share the GPUs that exist (RCCL needs one GPU per rank)
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    torch.manual_seed(2023 + rank)          # per-rank noise stream (DropPath masks, Gumbel / jitter draws): seed = base + rank, SURVEY 8(e)
    if not stub:
        torch.cuda.manual_seed_all(2023 + rank)
    if not stub and a.set:
        from tcct_amd import ops as _ops
        for kv in a.set.split(','):
            k_, v_ = kv.split('=')
            if not hasattr(_ops, k_):
                raise SystemExit(f'--set: tcct_amd.ops has no constant {k_}')
            cur_ = getattr(_ops, k_)
            setattr(_ops, k_, int(v_) if isinstance(cur_, int) and not isinstance(cur_, bool) else bool(int(v_)))
    if not stub and (a.wgrad_mode or a.conv_mode):
        from tcct_amd._lib import lib as _lib
        _lib.conv32_wgrad_mode(a.wgrad_mode)
        _lib.conv32_fwd_mode(a.conv_mode)
    if a.roofline_only:
        print(json.dumps({'roofline': dominant_kernel_roofline(a)}), file=out_stream, flush=True)
        return
    # The dominant-kernel timing runs BEFORE the training loop, on a quiet allocator: measured after it (27 GB pool live, two side streams
    # warm) the same kernel read 0.263 ms against 0.216-0.245 ms cold (round-2 review) -- ONE place, the cold one, is what the line reports.
    roof = copyc = None
    if rank == 0 and not a.no_roofline and not stub:
        roof = dominant_kernel_roofline(a)
        copyc = copy_ceiling(a)
        torch.cuda.empty_cache()
    if stub:
        tdist.init('gloo')
        k, img, lab = _StubTrainer(world, rank), None, None
    else:
        k, ds, args = build_trainer(a, world)
        k.model.train()
        batch = ds.make_batch(a.bs, seed=2023 + rank)
        img, lab, _, _ = ds.parse(batch)
        img, lab = img.contiguous(), lab.contiguous()
    sync()
    for _ in range(a.warmup):
        k.train_step(img, lab)
    sync()
    tdist.barrier()
    sync()
    t0 = time.perf_counter()
    loss = None
    marks = [] if stub else [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]     # per-step GPU timeline (diagnostic; no host sync)
    host = [time.perf_counter()]
    if marks:
        marks[0].record()
    for i in range(a.steps):
        loss = k.train_step(img, lab)
        if marks:
            marks[i + 1].record()
        host.append(time.perf_counter())
    sync()
    tdist.barrier()
    sync()
    dt = time.perf_counter() - t0
    dt = tdist.max_over_ranks(dt, dev)
    lossv = float(loss.item())
    devices = tdist.gather_strings('cpu (stub)' if stub else f'cuda:{local} ({torch.cuda.get_device_name(local)})')
    ranks_seen = None
    import torch.distributed as tdd0
    if tdd0.is_initialized():       # evidence that a collective over ALL ranks really ran (every rank takes part): ones all-reduced == world size
        one = torch.ones(1, device=dev)
        tdd0.all_reduce(one)
        ranks_seen = int(one.item())
    tdist.barrier()
    if rank != 0:
        tdist.barrier()         # leave together with rank 0 (which still times the roofline kernels): no rank tears the group down early
        return
    import torch.distributed as tdd
    ranks = tdd.get_world_size() if tdd.is_initialized() else 1
    if ranks != a.gpus:
        raise SystemExit(f'--gpus={a.gpus} but the process group has {ranks} ranks')
    value = a.bs * world * a.steps / dt
    out = {
        'metric': ('STUB (launcher rehearsal, no GPU work): ' if stub else '') + 'OCT B-scans/sec fwd+bwd(+clip+AdamW), stc_tt bs=8 1x800x1100',
        'value': round(value, 3), 'unit': 'B-scans/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype,
        'data': 'stub: one gloo all-reduce of the flat gradient size per step, CPU tensors' if stub else 'synthetic',
        'config': {'workload': f'stc_tt{"" if a.att == "pool" else "(att=" + a.att + ")"} --los={a.los} bs={a.bs}/GPU 1x{a.height}x{a.width} (net tensors 3x{a.height}x{(a.width + 15) // 16 * 16})',
                   'global_batch': a.bs * world, 'parallelism': f'dp{world}', 'ranks': ranks,
                   'backend': (tdd.get_backend() if tdd.is_initialized() else None), 'rank_devices': devices,
                   'grad_allreduce': getattr(k.optimG, 'allreduce_mode', 'none'), 'loss_last': round(lossv, 4),
                   'step_ms_gpu_min_med_max': _min_med_max([marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]) if marks else None,
                   'step_ms_host_enqueue_min_med_max': _min_med_max([1e3 * (host[i + 1] - host[i]) for i in range(a.steps)]),
                   'slowest_step': max(range(a.steps), key=lambda i: host[i + 1] - host[i]),
                   'peak_mem_GB': None if stub else round(torch.cuda.max_memory_allocated() / 2**30, 2)},
    }
    import torch.distributed as tdd2
    if tdd2.is_initialized():
        out['config']['allreduce_ranks_seen'] = ranks_seen
        try:
            out['config']['nccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version()) if tdd2.get_backend() == 'nccl' else None
        except Exception as e:                                  # noqa: BLE001
            out['config']['nccl_version'] = f'unavailable: {e}'
    out['config']['clip_adamw_ms'] = None if stub else optimizer_ms(k)
    if world == 1 and not stub and not a.no_other_workloads and not a.no_cpu_baseline and a.los == 'di':      # (the profiling / A-B tools all pass --no-cpu-baseline)
        # BASELINE.json configs[2] / [3] on the same box, in the same line (the headline `value` stays configs[1]): a fresh trainer per loss set -- the optimizer's
        # flat layout and the model's head composition depend on the loss flags -- 3 warm-up + 10 timed steps each, the same barrier-free single-rank timing.
        del k
        torch.cuda.empty_cache()
        others = []
        for los in ('di+reg', 'di+reg+fpl'):
            a2 = argparse.Namespace(**dict(vars(a), los=los))
            k2, ds2, _ = build_trainer(a2, world)
            k2.model.train()
            b2 = ds2.make_batch(a.bs, seed=2023 + rank)
            i2, l2, _, _ = ds2.parse(b2)
            i2, l2 = i2.contiguous(), l2.contiguous()
            for _ in range(3):
                k2.train_step(i2, l2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                lz = k2.train_step(i2, l2)
            torch.cuda.synchronize()
            d2 = time.perf_counter() - t1
            others.append({'workload': f'stc_tt --los={los} bs={a.bs}/GPU 1x{a.height}x{a.width}', 'ms_per_step': round(d2 / 10 * 1e3, 3),
                           'value': round(a.bs * 10 / d2, 3), 'unit': 'B-scans/s', 'steps': 10, 'warmup': 3, 'loss_last': round(float(lz.item()), 4)})
            del k2, ds2, i2, l2, b2
            torch.cuda.empty_cache()
        out['config']['other_workloads'] = others
    if roof is not None:
        out['roofline'] = roof
        out['roofline']['measured'] = 'before the training loop (quiet allocator), after a 120-copy spin-up (clock transient of the first ~10 ms of GPU activity, see dominant_kernel_roofline)'
        out['roofline']['copy_ceiling'] = copyc
        # whole-step view against the layer-granular traffic model of SURVEY §8(d): 7.38 GB (bf16) / 14.8 GB (fp32) per B-scan
        per_img = 7.38e9 if a.dtype == 'bf16' else 14.8e9
        out['roofline']['step_model_GBs'] = round(per_img * value / world / 1e9, 1)
    if world == 1 and not a.no_cpu_baseline and not stub:
        out['cpu_baseline'] = cpu_baseline(a)
    print(json.dumps(out), file=out_stream, flush=True)
    tdist.barrier()


def _min_med_max(v):
    v = sorted(v)
    return [round(v[0], 2), round(v[len(v) // 2], 2), round(v[-1], 2)]


def _shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    try:
        main()
    finally:
        _shutdown()
